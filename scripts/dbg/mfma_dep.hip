// Issue rate of v_mfma_f32_32x32x16_f16: one dependent chain on a single accumulator against 2 / 4 independent accumulators
// (one wave per SIMD and two waves per SIMD).  Prints cycles per MFMA (s_memtime ticks are 100 MHz: we use wall clock instead).
#include <hip/hip_runtime.h>
#include <cstdio>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 12 / NACC; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int threads, const char *what)
{
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 12 * (threads / 256);   // waves per SIMD = threads / 256
    const double tflops = 256.0 * 4 * mfma_per_simd * 32768 / (ms * 1e-3) / 1e12;
    printf("%-28s acc %d  %7.2f ms  %6.1f TF/s  %5.1f ns per MFMA per SIMD\n", what, NACC, ms, tflops, ms * 1e6 / mfma_per_simd);
    hipFree(out);
}

int main()
{
    run<1>(256, "1 wave/SIMD"); run<2>(256, "1 wave/SIMD"); run<4>(256, "1 wave/SIMD");
    run<1>(512, "2 waves/SIMD"); run<2>(512, "2 waves/SIMD"); run<4>(512, "2 waves/SIMD");
    return 0;
}
