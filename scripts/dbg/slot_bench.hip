// What one wave per SIMD sustains in a "slot" of the two-tile chain kernels (chain_pair.hip.h): three 32x32x16 f16 MFMAs
// on one accumulator plus NV vector instructions in NCH independent dependency chains (and optionally four ds_read_b128).
// Prints ns per slot (wall clock; 96 MFMA cycles are ~44 ns at 2.2 GHz).  Measured (MI355X): 3 MFMA 50 ns; 24 VALU 36 ns; 3 MFMA + 24
// VALU 72 ns — a single wave's vector instructions hide only half under its own MFMAs.  The `lds` variants use one element of each
// ds_read_b128, so the compiler overlaps their destination registers and waits between them: they time the LDS LATENCY
// (~30 ns per read), not its throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NM, int NV, int NCH, int NLDS, int DIST = 0>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ f32x4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f32x4{1.f * i, 2.f, 3.f, 4.f};
    __syncthreads();
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[8];
    for (int c = 0; c < 8; ++c) v[c] = threadIdx.x * 0.01f + c;
    f32x4 l[4] = {}, lp[4] = {}, lq[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // four slots per iteration
            if (NM > 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NLDS; ++i) l[i] = lds[(threadIdx.x + 64 * i + 256 * u + it) & 1023];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 1; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i % NCH] = __builtin_fmaf(v[i % NCH], 1.0001f, 0.5f);
            if (NM > 1) {
#pragma unroll
                for (int m = 1; m < NM; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NLDS; ++i) { v[i % NCH] += DIST == 2 ? lq[i][0] : DIST ? lp[i][0] : l[i][0]; lq[i] = lp[i]; lp[i] = l[i]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int c = 0; c < 8; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int NCH, int NLDS, int DIST = 0>
void run()
{
    float *out;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, NCH, NLDS, DIST>), dim3(256), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, NCH, NLDS, DIST>), dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("mfma %d  valu %2d in %d chains  lds %d dist %d : %6.1f ns per slot\n", NM, NV, NCH, NLDS, DIST, ms * 1e6 / (iters * 4.0));
    (void)hipFree(out);
}

int main()
{
    run<3, 0, 1, 0>(); run<0, 24, 8, 0>(); run<0, 24, 4, 0>(); run<0, 24, 2, 0>(); run<0, 24, 1, 0>();
    run<3, 12, 4, 0>(); run<3, 24, 8, 0>(); run<3, 24, 4, 0>(); run<3, 24, 2, 0>(); run<3, 24, 1, 0>();
    run<3, 0, 1, 2>(); run<3, 0, 1, 4>(); run<3, 24, 4, 4>(); run<3, 16, 4, 4>();
    run<0, 0, 1, 2>(); run<0, 0, 1, 4>(); run<0, 24, 4, 4>(); run<3, 0, 1, 2, 1>(); run<3, 0, 1, 4, 1>(); run<3, 24, 4, 4, 1>(); run<3, 24, 4, 2, 1>(); run<3, 24, 4, 1, 1>();
    run<3, 0, 1, 2, 2>(); run<3, 0, 1, 4, 2>(); run<3, 24, 4, 4, 2>(); run<3, 24, 4, 2, 2>(); run<0, 0, 1, 4, 2>();
    return 0;
}
