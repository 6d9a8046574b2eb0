// Practical fp32-MFMA ceiling on this box: every wave runs a dependent-free stream of v_mfma_f32_32x32x2_f32
// (4 independent accumulators, operands in registers), 8 or 4 waves per CU.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void peak(float *out, int iters)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}
template <int WAVES>
void run(const char *name, int blocks, int iters)
{
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(peak<WAVES>, dim3(blocks), dim3(64 * WAVES), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak<WAVES>, dim3(blocks), dim3(64 * WAVES), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * WAVES * iters * 64 * 4096.0;
    printf("%s: blocks %d waves %d iters %d  %.3f ms  %.1f TF/s\n", name, blocks, WAVES, iters, ms, flops / ms * 1e-9);
    hipFree(out);
}
int main()
{
    run<4>("1 wave/SIMD", 256, 20000);
    run<8>("2 waves/SIMD", 256, 10000);
    run<4>("1 wave/SIMD x4 rounds", 1024, 5000);
    run<16>("4 waves/SIMD", 256, 5000);
    return 0;
}
