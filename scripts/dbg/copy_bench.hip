// HBM streaming microbenchmark: float4 copy / read-only variants (grid shape, temporal hint, unroll) on a 1 GiB buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(const f32x4 *__restrict__ s, f32x4 *__restrict__ d, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) d[i] = s[i];
}
// block-contiguous: each block copies a contiguous chunk
template <int U>
__global__ __launch_bounds__(256) void copy_chunk_k(const f32x4 *__restrict__ s, f32x4 *__restrict__ d, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    size_t i = b + threadIdx.x;
    for (; i + (U - 1) * 256 < e; i += U * 256) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) d[i + u * 256] = v[u];
    }
    for (; i < e; i += 256) d[i] = s[i];
}
template <int U>
__global__ __launch_bounds__(256) void read_k(const f32x4 *__restrict__ s, float *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
#define T(name, bytes, launch) { launch; hipDeviceSynchronize(); hipEventRecord(a); for (int r = 0; r < 10; ++r) { launch; } hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); printf("%-44s %7.1f GB/s\n", name, (double)(bytes) * 10 / (ms * 1e-3) / 1e9); }
int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    f32x4 *s, *d; float *o;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMalloc(&o, 64);
    hipMemset(s, 0, bytes); hipMemset(d, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned full = (unsigned)(n / 256);
    T("copy plain   full grid (1 vec/thread)", 2 * bytes, (copy_k<1, false><<<full, 256>>>(s, d, n)));
    T("copy plain   full grid / 4, U=4", 2 * bytes, (copy_k<4, false><<<full / 4, 256>>>(s, d, n)));
    T("copy nt      full grid / 4, U=4", 2 * bytes, (copy_k<4, true><<<full / 4, 256>>>(s, d, n)));
    T("copy plain   2048 WG, U=8", 2 * bytes, (copy_k<8, false><<<2048, 256>>>(s, d, n)));
    T("copy nt      2048 WG, U=8", 2 * bytes, (copy_k<8, true><<<2048, 256>>>(s, d, n)));
    T("copy plain   4096 WG, U=4", 2 * bytes, (copy_k<4, false><<<4096, 256>>>(s, d, n)));
    T("copy plain   8192 WG, U=4", 2 * bytes, (copy_k<4, false><<<8192, 256>>>(s, d, n)));
    T("copy chunked 2048 WG, U=4", 2 * bytes, (copy_chunk_k<4><<<2048, 256>>>(s, d, n)));
    T("copy chunked 8192 WG, U=4", 2 * bytes, (copy_chunk_k<4><<<8192, 256>>>(s, d, n)));
    T("read only    2048 WG, U=8", bytes, (read_k<8><<<2048, 256>>>(s, o, n)));
    T("read only    4096 WG, U=4", bytes, (read_k<4><<<4096, 256>>>(s, o, n)));
    T("read only    full grid / 4, U=4", bytes, (read_k<4><<<full / 4, 256>>>(s, o, n)));
    T("hipMemcpyDtoD", 2 * bytes, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0));
    return 0;
}
