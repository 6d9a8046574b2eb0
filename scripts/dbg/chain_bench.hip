// Where do the chain kernels lose MFMA time?  11 chained GEMMs per block like attn_post, with knobs.
#include "../../tempenslearnagg-qcnet_amd/csrc/chain_tile.hip.h"
#include <cstdio>
#include <vector>
using namespace qcn_chain;

template <int PF>
__device__ __forceinline__ void gemm_pf(f32x16 (&acc)[4], const f32x16 (&in)[4], const f32x4 *w, int lane)
{
    f32x4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = w[i * 64 + lane];
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const f32x4 a = ring[s % PF];
        if (s + PF < 64) ring[s % PF] = w[(s + PF) * 64 + lane];
        const float b = in[s >> 4][s & 15];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b, acc[3], 0, 0, 0);
    }
}

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;

// split-f16 GEMM: operands carried as hi + 2^-11 lo' (two f16 planes, 22 bits), three products, fp32 accumulate;
// the row is scaled by a power of two first so that its largest element is in [1/2, 1).
__device__ __forceinline__ void gemm_f16x3(f32x16 (&acc)[4], const f32x16 (&in)[4], const f32x4 *w, int lane)
{
    // row scale
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(in[t][r]));
    mx = fmaxf(mx, xhalf(mx));
    int e;
    (void)frexpf(mx, &e);
    const float sc = ldexpf(1.0f, -e), usc = ldexpf(1.0f, e);
    f32x16 lo_acc[4];
    fill_zero(lo_acc);
    f32x16 hi_acc[4];
    fill_zero(hi_acc);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        half8 bh, bl;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float x0 = in[j >> 1][8 * (j & 1) + 2 * p] * sc, x1 = in[j >> 1][8 * (j & 1) + 2 * p + 1] * sc;
            const half2v hh = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
            const float r0 = (x0 - (float)hh[0]) * 2048.f, r1 = (x1 - (float)hh[1]) * 2048.f;
            const half2v ll = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(r0, r1));
            bh[2 * p] = hh[0]; bh[2 * p + 1] = hh[1];
            bl[2 * p] = ll[0]; bl[2 * p + 1] = ll[1];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const half8 ah = __builtin_bit_cast(half8, w[((j * 4 + t) * 2 + 0) * 64 + lane]);
            const half8 al = __builtin_bit_cast(half8, w[((j * 4 + t) * 2 + 1) * 64 + lane]);
            hi_acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, hi_acc[t], 0, 0, 0);
            lo_acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, lo_acc[t], 0, 0, 0);
            lo_acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, lo_acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += (hi_acc[t][r] + lo_acc[t][r] * (1.0f / 2048.f)) * usc;
}

// MODE bit0: stage each matrix; bit1: barrier each GEMM; PF: 0 = library gemm(), else explicit prefetch distance
// IO bit0: store the GEMM result row-per-lane each GEMM; bit1: load a row vector each GEMM and add it
template <int WAVES, int MODE, int PF, int IO = 0>
__global__ __launch_bounds__(64 * WAVES) void bench(const float *__restrict__ x, const float *__restrict__ mats,
                                                    float *__restrict__ y, int64_t n_rows, int n_gemm)
{
    __shared__ __attribute__((aligned(16))) f32x4 wlds[2 * kMatVec4];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32 + (lane & 31);
    const int64_t rowc = row < n_rows ? row : n_rows - 1;
    stage_matrix<WAVES>(mats, wlds, wave, lane);
    stage_matrix<WAVES>(mats + kMatFloats, wlds + kMatVec4, wave, lane);
    f32x16 v[4], acc[4];
    load_row(v, x, rowc, h);
    sync_for_stage();
#pragma unroll 1
    for (int g = 0; g < n_gemm; ++g) {
        fill_zero(acc);
        if (MODE & 2) sync_for_stage();
        if (MODE & 1) stage_matrix<WAVES>(mats + (size_t)((g + 1) % 11) * kMatFloats, wlds + ((g + 1) & 1) * kMatVec4, wave, lane);
        if (PF == 99) gemm_f16x3(acc, v, wlds + (g & 1) * kMatVec4, lane);
        else if (PF == 0) gemm<WAVES>(acc, v, wlds + (g & 1) * kMatVec4, lane);
        else gemm_pf<(PF > 0 && PF < 9) ? PF : 1>(acc, v, wlds + (g & 1) * kMatVec4, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = acc[t] * 0.05f;
        if (IO & 1) { if (row < n_rows) store_row(v, y, row, h); }
        if (IO & 2) {
            f32x16 e[4];
            load_row(e, x, rowc, h);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] += e[t];
        }
    }
    if (row < n_rows) store_row(v, y, row, h);
}

template <int WAVES, int MODE, int PF, int IO = 0>
void run(const float *x, const float *mats, float *y, int64_t n_rows)
{
    const unsigned grid = (unsigned)((n_rows + 32 * WAVES - 1) / (32 * WAVES));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<WAVES, MODE, PF, IO>), dim3(grid), dim3(64 * WAVES), 0, 0, x, mats, y, n_rows, 11);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((bench<WAVES, MODE, PF, IO>), dim3(grid), dim3(64 * WAVES), 0, 0, x, mats, y, n_rows, 11);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    printf("waves %d stage %d barrier %d prefetch %d io %d: %8.1f us  %6.1f TF/s\n", WAVES, MODE & 1, (MODE >> 1) & 1, PF, IO, ms * 1e3,
           (double)n_rows * 11 * 32768.0 / ms * 1e-9);
}

int main(int argc, char **argv)
{
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 800000;
    float *x, *mats, *y;
    hipMalloc(&x, n_rows * 512); hipMalloc(&y, n_rows * 512); hipMalloc(&mats, 11 * kMatFloats * 4);
    std::vector<float> hx(n_rows * 128), hm(11 * kMatFloats);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    for (size_t i = 0; i < hm.size(); ++i) hm[i] = (float)((i * 40503u) >> 4 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(mats, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    run<8, 3, 0>(x, mats, y, n_rows);
    run<8, 3, 99>(x, mats, y, n_rows);
    run<8, 3, 99, 1>(x, mats, y, n_rows);
    run<4, 3, 99>(x, mats, y, n_rows);
    run<8, 0, 99>(x, mats, y, n_rows);
    return 0;
}
