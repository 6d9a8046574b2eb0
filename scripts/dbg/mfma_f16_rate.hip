// Issue rate of v_mfma_f32_32x32x16_f16 in the recurrence's shape: one wave per SIMD, 72 MFMAs a round over three accumulators
// in rotation, the A operands from 192 resident registers (half of them end up in AGPRs).  Variants: A from 192 registers or
// from 8; three accumulators or six; with or without four vector instructions per gap.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f16_rate.hip -o mfma_f16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ unsigned long long g_cycles[4];
template <int NA, int NACC, int NV, int KIND = 1>
__global__ __launch_bounds__(256) void rate(const f32x4 *w, float *out, int iters)
{
    const int lane = threadIdx.x & 63;
    half8 wf[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) wf[i] = __builtin_bit_cast(half8, w[i * 64 + lane]);
    half8 b0 = __builtin_bit_cast(half8, w[lane]), b1 = __builtin_bit_cast(half8, w[64 + lane]);
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float v[4] = {1.f, 2.f, 3.f, 4.f};
    f32x16 other[3];
    for (int t = 0; t < 3; ++t) { other[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t % NA], b0, acc[0], 0, 0, 0); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 72; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(k * 7) % NA], (k & 1) ? b0 : b1, acc[k % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (KIND == 1) v[i & 3] = fmaf(v[i & 3], 1.0001f, 0.5f);
                if (KIND == 2) v[i & 3] = __builtin_amdgcn_exp2f(v[i & 3]);
                if (KIND == 3) v[i & 3] = __builtin_amdgcn_rcpf(v[i & 3]);
                if (KIND == 4) v[i & 3] = fmaf(acc[(k + 1) % NACC][(k + i) & 15], 1.0001f, v[i & 3]);   // reads an accumulator two MFMAs old
                if (KIND == 5) v[i & 3] = fmaf(other[(k / 4) % 3][(k * 4 + i) & 15], 1.0001f, v[i & 3]);   // reads finished accumulators
                asm volatile("" : "+v"(v[i & 3]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) g_cycles[0] = t1 - t0;
    float s = v[0] + v[1] + v[2] + v[3] + other[0][3] + other[1][5] + other[2][7];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}
template <int NA, int NACC, int NV, int KIND = 1>
void run(const char *name, int blocks)
{
    f32x4 *w; float *out;
    hipMalloc(&w, 192 * 64 * 16); hipMemset(w, 0, 192 * 64 * 16); hipMalloc(&out, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<NA, NACC, NV, KIND>), dim3(blocks), dim3(256), 0, 0, w, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<NA, NACC, NV, KIND>), dim3(blocks), dim3(256), 0, 0, w, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc; hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cycles), 8);
    printf("%-34s blocks %4d: %6.1f s_memtime ticks / MFMA, %6.2f ns / MFMA  (%.0f TF/s f16 over the chip)\n", name, blocks,
           (double)cyc / (72.0 * iters), ms * 1e6 / (72.0 * iters), blocks * 4 * 72.0 * iters * 32768.0 / ms * 1e-9);
    hipFree(w); hipFree(out);
}
int main()
{
    run<48, 3, 0>("A 192 regs, 3 acc", 256);
    run<48, 3, 0>("A 192 regs, 3 acc", 128);
    run<2, 3, 0>("A 8 regs, 3 acc", 256);
    run<2, 6, 0>("A 8 regs, 6 acc", 256);
    run<48, 3, 4>("A 192 regs, 3 acc, 4 fma / gap", 256);
    run<48, 3, 6>("A 192 regs, 3 acc, 6 fma / gap", 256);
    run<2, 3, 4>("A 8 regs, 3 acc, 4 fma / gap", 256);
    run<48, 3, 2, 2>("A 192, 3 acc, 2 exp / gap", 256);
    run<48, 3, 4, 2>("A 192, 3 acc, 4 exp / gap", 256);
    run<48, 3, 4, 3>("A 192, 3 acc, 4 rcp / gap", 256);
    run<48, 3, 4, 5>("A 192, 3 acc, 4 fma(finished acc) / gap", 256);
    run<48, 3, 2, 5>("A 192, 3 acc, 2 fma(finished acc) / gap", 256);
    return 0;
}
