// Does v_mfma_f32_32x32x16_f16 issue every 32 cycles whatever register file its operands come from?  One wave per SIMD,
// three accumulators in rotation, operand classes forced through inline-asm constraints ("a" = AGPR, "v" = VGPR).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_agpr_rate.hip -o mfma_agpr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ unsigned long long g_cycles[4];
#define MF(CA, CB, CC)                                                                                                       \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+" CC(acc[k % 3]) : CA(wf[(k * 7) % 16]), CB((k & 1) ? b0 : b1))
template <int KIND>
__global__ __launch_bounds__(256) void rate(const f32x4 *w, float *out, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x4 wf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wf[i] = w[i * 64 + lane];
    f32x4 b0 = w[lane], b1 = w[64 + lane];
    f32x16 acc[3];
    for (int t = 0; t < 3; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 72; ++k) {
            if (KIND == 0) MF("v", "v", "a");
            if (KIND == 1) MF("a", "v", "a");
            if (KIND == 2) MF("a", "v", "v");
            if (KIND == 3) MF("v", "v", "v");
            if (KIND == 4) MF("a", "a", "a");
            if (KIND == 5) MF("v", "a", "a");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) g_cycles[0] = t1 - t0;
    float s = 0.f;
    for (int t = 0; t < 3; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}
template <int KIND>
void run(const char *name)
{
    f32x4 *w; float *out;
    hipMalloc(&w, 192 * 64 * 16); hipMemset(w, 0, 192 * 64 * 16); hipMalloc(&out, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<KIND>), dim3(256), dim3(256), 0, 0, w, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<KIND>), dim3(256), dim3(256), 0, 0, w, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc; hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cycles), 8);
    printf("%-28s %6.1f cycles / MFMA, %6.2f ns / MFMA\n", name, (double)cyc / (72.0 * iters), ms * 1e6 / (72.0 * iters));
    hipFree(w); hipFree(out);
}
int main()
{
    run<0>("A vgpr, B vgpr, C/D agpr");
    run<1>("A agpr, B vgpr, C/D agpr");
    run<2>("A agpr, B vgpr, C/D vgpr");
    run<3>("A vgpr, B vgpr, C/D vgpr");
    run<4>("A agpr, B agpr, C/D agpr");
    run<5>("A vgpr, B agpr, C/D agpr");
    return 0;
}
