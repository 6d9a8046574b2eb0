// Exclusive scan and destination-sorted CSR construction (see include/qcnet_hip.h).
// Sorting / scanning are rocPRIM device primitives (stable LSD radix sort, single-pass scan): they are the
// right tool for byte/integer reshuffles; the hand-written kernels around them only convert and search.
#include <cstring>
#include <mutex>

#include <rocprim/rocprim.hpp>

#include "common.hip.h"

namespace {

inline size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

struct CastI32ToI64 {
    __device__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

__global__ __launch_bounds__(256) void csr_prepare_kernel(const int64_t *__restrict__ dst, int64_t E,
                                                          int32_t *__restrict__ key, int32_t *__restrict__ val)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    key[e] = (int32_t)dst[e];
    val[e] = (int32_t)e;
}

// rowptr[i] = first CSR slot whose destination is >= i (lower bound on the sorted keys); col gathers src.
__global__ __launch_bounds__(256) void csr_finish_kernel(const int32_t *__restrict__ key_sorted,
                                                         const int32_t *__restrict__ eid,
                                                         const int64_t *__restrict__ src, int64_t E, int64_t n_dst,
                                                         int32_t *__restrict__ rowptr, int32_t *__restrict__ col)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_dst) {
        int64_t lo = 0, hi = E;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (key_sorted[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
        }
        rowptr[i] = (int32_t)lo;
    }
    if (i < E) col[i] = (int32_t)src[eid[i]];
}

}  // namespace

extern "C" size_t qcn_scan_workspace_bytes(int64_t n)
{
    if (n <= 0) return 256;
    size_t bytes = 0;
    auto in = rocprim::make_transform_iterator((const int32_t *)nullptr, CastI32ToI64());
    rocprim::exclusive_scan(nullptr, bytes, in, (int64_t *)nullptr, (int64_t)0, (size_t)(n + 1),
                            rocprim::plus<int64_t>(), (hipStream_t)0);
    return align256(bytes) + 256;
}

extern "C" int qcn_exclusive_scan_i32(const int32_t *count, int64_t n, int64_t *offset, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    if (n < 0 || !offset) return QCN_ERR_INVALID_ARGUMENT;
    if (n == 0) return hipMemsetAsync(offset, 0, sizeof(int64_t), qcn_stream(stream)) == hipSuccess
                           ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
    if (!count || !workspace) return QCN_ERR_INVALID_ARGUMENT;
    // Scan n + 1 elements so that offset[n] receives the total: `count` must have one readable slot past
    // the end (the Python wrapper allocates n + 1 and zeroes the tail).
    size_t need = 0;
    auto in = rocprim::make_transform_iterator(count, CastI32ToI64());
    if (rocprim::exclusive_scan(nullptr, need, in, offset, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(),
                                qcn_stream(stream)) != hipSuccess)
        return QCN_ERR_LAUNCH_FAILED;
    if (need > workspace_bytes) return QCN_ERR_WORKSPACE_TOO_SMALL;
    if (rocprim::exclusive_scan(workspace, need, in, offset, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(),
                                qcn_stream(stream)) != hipSuccess)
        return QCN_ERR_LAUNCH_FAILED;
    return QCN_OK;
}

static unsigned csr_end_bit(int64_t n_dst)
{
    unsigned end_bit = 1;
    while (end_bit < 32 && ((int64_t)1 << end_bit) < n_dst) ++end_bit;
    return end_bit;
}

static size_t csr_sort_temp_bytes(int64_t E, unsigned end_bit)
{
    size_t bytes = 0;
    rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr,
                              (int32_t *)nullptr, (size_t)E, 0u, end_bit, (hipStream_t)0);
    return bytes;
}

extern "C" size_t qcn_csr_workspace_bytes(int64_t num_edges, int64_t n_dst)
{
    if (num_edges <= 0) return 256;
    const size_t arr = align256((size_t)num_edges * sizeof(int32_t));
    return 3 * arr + align256(csr_sort_temp_bytes(num_edges, csr_end_bit(n_dst))) + 256;
}

// shared body of qcn_csr_build / qcn_csr_build_static: `ws` holds qcn_csr_workspace_bytes(num_edges, n_dst) bytes
static int csr_build_impl(const int64_t *src, const int64_t *dst, int64_t num_edges, int64_t n_dst, int32_t *rowptr,
                          int32_t *col, int32_t *eid, char *ws, hipStream_t s)
{
    const size_t arr = align256((size_t)num_edges * sizeof(int32_t));
    int32_t *key_in = (int32_t *)ws;
    int32_t *val_in = (int32_t *)(ws + arr);
    int32_t *key_out = (int32_t *)(ws + 2 * arr);
    void *temp = ws + 3 * arr;
    const unsigned end_bit = csr_end_bit(n_dst);
    size_t temp_bytes = csr_sort_temp_bytes(num_edges, end_bit);
    const unsigned grid_e = (unsigned)((num_edges + 255) / 256);
    hipLaunchKernelGGL(csr_prepare_kernel, dim3(grid_e), dim3(256), 0, s, dst, num_edges, key_in, val_in);
    QCN_CHECK_LAUNCH();
    if (rocprim::radix_sort_pairs(temp, temp_bytes, key_in, key_out, val_in, eid, (size_t)num_edges, 0u, end_bit, s) !=
        hipSuccess)
        return QCN_ERR_LAUNCH_FAILED;
    const int64_t n = num_edges > n_dst + 1 ? num_edges : n_dst + 1;
    hipLaunchKernelGGL(csr_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, key_out, eid, src,
                       num_edges, n_dst, rowptr, col);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

static int csr_check_args(const int64_t *src, const int64_t *dst, int64_t num_edges, int64_t n_dst, int32_t *rowptr,
                          int32_t *col, int32_t *eid)
{
    if (num_edges < 0 || n_dst < 0 || !rowptr) return QCN_ERR_INVALID_ARGUMENT;
    if (num_edges >= (int64_t)1 << 31 || n_dst >= ((int64_t)1 << 31) - 1) return QCN_ERR_UNSUPPORTED;
    if (num_edges > 0 && (!src || !dst || !col || !eid)) return QCN_ERR_INVALID_ARGUMENT;
    return QCN_OK;
}

extern "C" int qcn_csr_build(const int64_t *src, const int64_t *dst, int64_t num_edges, int64_t n_dst,
                             int32_t *rowptr, int32_t *col, int32_t *eid, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    const int st = csr_check_args(src, dst, num_edges, n_dst, rowptr, col, eid);
    if (st != QCN_OK) return st;
    hipStream_t s = qcn_stream(stream);
    if (num_edges == 0)
        return hipMemsetAsync(rowptr, 0, (size_t)(n_dst + 1) * sizeof(int32_t), s) == hipSuccess
                   ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
    if (!workspace) return QCN_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < qcn_csr_workspace_bytes(num_edges, n_dst)) return QCN_ERR_WORKSPACE_TOO_SMALL;
    return csr_build_impl(src, dst, num_edges, n_dst, rowptr, col, eid, (char *)workspace, s);
}

// ------------------------------------------------------------------------------------------------
// qcn_csr_build_static — the same CSR build for callers that cannot allocate (HIP graph capture, serving loops):
// the sort's workspace (keys, edge ids, sorted keys, radix-sort temporaries) is this file-scope device array instead
// of a caller buffer.  Every GPU the library runs on holds its own copy until the process ends; the array's name is a
// host-side stand-in, so its per-device address is looked up once per device (hipGetSymbolAddress with that device
// current) and only that pointer is ever handed to kernels.
// ------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[QCN_CSR_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {};
}  // namespace

// device pointer of `scratch` on the current device (nullptr on failure)
static unsigned char *csr_static_scratch()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || p == nullptr) return nullptr;
        g_scratch_ptr[dev] = static_cast<unsigned char *>(p);
    }
    return g_scratch_ptr[dev];
}

// Also the set-up call: resolves the current device's array address (hipGetSymbolAddress + a mutex, neither of which
// belongs inside a graph capture), so that qcn_csr_build_static itself only reads a cached pointer afterwards.
extern "C" int64_t qcn_csr_static_max_edges(void)
{
    (void)csr_static_scratch();
    return QCN_CSR_STATIC_MAX_EDGES;
}

extern "C" int qcn_csr_build_static(const int64_t *src, const int64_t *dst, int64_t num_edges, int64_t n_dst,
                                    int32_t *rowptr, int32_t *col, int32_t *eid, void *stream)
{
    const int st = csr_check_args(src, dst, num_edges, n_dst, rowptr, col, eid);
    if (st != QCN_OK) return st;
    if (num_edges > QCN_CSR_STATIC_MAX_EDGES) return QCN_ERR_TOO_LARGE;
    hipStream_t s = qcn_stream(stream);
    if (num_edges == 0)
        return hipMemsetAsync(rowptr, 0, (size_t)(n_dst + 1) * sizeof(int32_t), s) == hipSuccess
                   ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
    if (qcn_csr_workspace_bytes(num_edges, n_dst) > (size_t)QCN_CSR_STATIC_SCRATCH_BYTES) return QCN_ERR_TOO_LARGE;
    unsigned char *ws = csr_static_scratch();
    if (ws == nullptr) return QCN_ERR_LAUNCH_FAILED;
    // nothing in the workspace has to start at zero: the prepare kernel writes every key / id slot the sort reads and
    // rocPRIM initialises its own histograms inside `temp`; whatever the previous call left behind is overwritten
    return csr_build_impl(src, dst, num_edges, n_dst, rowptr, col, eid, (char *)ws, s);
}
