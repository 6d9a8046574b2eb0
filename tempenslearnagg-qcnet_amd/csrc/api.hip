#include "common.hip.h"

extern "C" int qcn_version(void) { return 200; }

extern "C" const char *qcn_status_string(int status)
{
    switch (status) {
    case QCN_OK: return "ok";
    case QCN_ERR_INVALID_ARGUMENT: return "invalid argument";
    case QCN_ERR_UNSUPPORTED: return "unsupported shape or size";
    case QCN_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
    case QCN_ERR_LAUNCH_FAILED: return "HIP launch failed";
    case QCN_ERR_TOO_LARGE: return "size exceeds the fixed capacity of this allocation-free entry point";
    default: return "unknown status";
    }
}

// Measurement aid (bench.py: roofline.achievable_peak): a plain streaming copy, one 16-B element per thread, one workgroup
// per 4 KiB in launch order — the guide's float4-copy microbenchmark (MI355X_MICROARCH.md: 6.29 TB/s read + write on
// HBM3E), as the honest denominator for "fraction of achievable" instead of a library memcpy.  (scripts/dbg/copy_bench.hip,
// profiles/r3_copy_bench.txt: this form 6.23 TB/s; persistent grids with grid-stride loops 4.3-5.3, hipMemcpyDtoD 5.1.)
namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n_vec)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_vec) dst[i] = src[i];
}
}  // namespace

extern "C" int qcn_stream_copy(const void *src, void *dst, size_t n_bytes, void *stream)
{
    if (n_bytes == 0) return QCN_OK;
    if (!src || !dst || (n_bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return QCN_ERR_INVALID_ARGUMENT;
    const size_t n_vec = n_bytes / 16;
    const size_t blocks = (n_vec + 255) / 256;
    if (blocks >= ((size_t)1 << 31)) return QCN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, qcn_stream(stream),
                       reinterpret_cast<const f32x4 *>(src), reinterpret_cast<f32x4 *>(dst), n_vec);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}
