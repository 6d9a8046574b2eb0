// How fast can one CU move gathered 512-B rows global -> LDS?  The data path of attn_modes_kernel in isolation:
// 8 waves per workgroup, per tile every wave issues N_DMA LDS-DMA pieces (1 KiB = two gathered 512-B rows), ring of 4
// slots, counted vmcnt + raw barrier, no compute.  Prints GB/s for tables of several sizes (L2 / Infinity Cache / HBM)
// and 1 or 2 workgroups per CU.   hipcc --offload-arch=gfx950 -O3 dma_gather.hip -o dma_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kWaves = 8;
template <int N_DMA>
__global__ __launch_bounds__(512) void gather_kernel(const char *table, const int *rows, int tiles_per_block, unsigned *sink)
{
    __shared__ __attribute__((aligned(16))) unsigned char s0[N_DMA * 8192], s1[N_DMA * 8192], s2[N_DMA * 8192], s3[N_DMA * 8192];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int *my = rows + (size_t)blockIdx.x * tiles_per_block * 16 * N_DMA;
    unsigned acc = 0;
    auto issue = [&](int t, unsigned char *slot) {
#pragma unroll
        for (int d = 0; d < N_DMA; ++d) {
            const int r = my[(t * N_DMA + d) * 16 + 2 * wave + (lane >> 5)];
            const char *src = table + (size_t)(unsigned)r * 512u + 16u * ((lane & 31) ^ (2 * wave + (lane >> 5)));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + d * 8192 + wave * 1024), 16, 0, 0);
        }
    };
    issue(0, s0); issue(1, s1); issue(2, s2);
    // tile t has landed when at most 3 younger tiles are in flight; every ring slot is its own __shared__ variable and a
    // compile-time phase (otherwise the compiler's LDS-DMA alias tracking drains the queue before each LDS read)
#define STEP(T, CUR, NXT)                                                                            \
    {                                                                                                \
        issue(min((T) + 3, tiles_per_block - 1), NXT);                                               \
        if (N_DMA == 3) __builtin_amdgcn_s_waitcnt(0x0070 | 9); else __builtin_amdgcn_s_waitcnt(0x0070 | 3); \
        __builtin_amdgcn_s_barrier();                                                                \
        acc += *reinterpret_cast<const unsigned *>(CUR + lane * 16 + wave * 1024);                   \
        __builtin_amdgcn_s_barrier();                                                                \
    }
    for (int t = 0; t < tiles_per_block; t += 4) {
        STEP(t, s0, s3) STEP(t + 1, s1, s0) STEP(t + 2, s2, s1) STEP(t + 3, s3, s2)
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const int tiles_per_block = 256;
    for (int n_dma : {1, 3})
        for (size_t table_rows : {(size_t)4096, (size_t)65536, (size_t)(1 << 22)})
            for (int per_cu : {1, 2}) {
                const int blocks = 256 * per_cu;
                if (n_dma == 3 && per_cu == 2) continue;     // 2 x 96 KiB of LDS do not fit
                char *table; int *rows; unsigned *sink;
                hipMalloc(&table, table_rows * 512); hipMemset(table, 1, table_rows * 512);
                const size_t n_idx = (size_t)blocks * tiles_per_block * 16 * n_dma;
                std::vector<int> h(n_idx);
                for (size_t k = 0; k < n_idx; ++k) h[k] = (int)(((k * 2654435761ull) >> 7) % table_rows);
                hipMalloc(&rows, n_idx * 4); hipMemcpy(rows, h.data(), n_idx * 4, hipMemcpyHostToDevice);
                hipMalloc(&sink, 4);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (n_dma == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(512), 0, 0, table, rows, tiles_per_block, sink);
                    else hipLaunchKernelGGL(gather_kernel<3>, dim3(blocks), dim3(512), 0, 0, table, rows, tiles_per_block, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)n_idx * 512.0;
                printf("pieces/wave/tile %d  table %7.1f MB  %d WG/CU: %8.1f us  %7.2f TB/s  %5.1f B/clk/CU (2.4 GHz)  err %d\n", n_dma,
                       table_rows * 512 / 1e6, per_cu, ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, (int)hipGetLastError());
                hipFree(table); hipFree(rows); hipFree(sink);
            }
    return 0;
}
