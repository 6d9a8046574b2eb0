// Feature-major fp32-MFMA chain machinery (gfx950) for the kernels whose rows are independent 128-wide vectors and
// whose work is a CHAIN of 128x128 linear layers with per-row epilogues between them: FourierEmbedding (fourier.hip)
// and the AttentionLayer node-side blocks (nodeops.hip).
//
// Every GEMM is computed transposed, Y^T = W X^T, with v_mfma_f32_32x32x2_f32 (exact fp32 fma chains):
//   * the WEIGHTS are the MFMA A operand (M = output features, four 32-feature tiles),
//   * the wave's 32 rows are the N dimension: lane (n = lane & 31, h = lane >> 5) only ever holds data of row n.
// A row vector lives in 64 registers, f32x16 v[4]:  v[t][r] = feature  32 t + (r & 3) + 8 (r >> 2) + 4 h  of row n.
// That is the accumulator layout of the MFMA *and* a valid B-operand layout for the next one (the k order of a
// contraction is free as long as both operands agree, and the weights are packed on the host in exactly the order the
// k-steps consume them): the output of one layer feeds the next layer straight from registers.  There is no
// activation tile in LDS, no transposition between layers, and LayerNorm / ReLU / sigmoid / residuals are in-lane
// loops plus ONE cross-half exchange per row reduction.
//
// LDS holds nothing but weights: two 64 KiB buffers, each one packed 128x128 matrix.  Matrix i+1 is copied
// global -> LDS by the LDS-DMA path (global_load_lds_dwordx4, no staging registers) while the MFMAs of matrix i run;
// one barrier per matrix.  Packed layout (host: ops.pack_chain_matrix):
//   P[s][lane][t] = W[out = 32 t + (lane & 31)][in = kfeat(s, lane >> 5)],   s = 0..63,  kfeat(s, h) = feature of v[s>>4][s&15]
// so k-step s is one conflict-free ds_read_b128 per lane feeding four MFMAs (one per output tile).
#pragma once
#include "common.hip.h"

namespace qcn_chain {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kC = 128;
constexpr int kRowsPerWave = 32;
constexpr int kMatFloats = kC * kC;            // one packed matrix
constexpr int kMatVec4 = kMatFloats / 4;       // in float4 units (= 64 k-steps x 64 lanes)
constexpr int kLdsBytes = 2 * kMatFloats * 4;  // 128 KiB

// feature index of element (t, r) for lane half h
__device__ __forceinline__ int feat(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }

// Block barrier taken right before stage_matrix: first retire every ordinary load issued so far (bias / operand rows
// headed for the accumulators), so that the DMA queued behind the barrier is never in front of something the first
// MFMA has to wait for (the VM counter retires in order; a wait placed after the DMA would drain it).
__device__ __forceinline__ void sync_for_stage()
{
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();
}

// Start the LDS-DMA copy of one packed matrix into LDS buffer `dst`; every wave of the block copies WAVES-th of it.
template <int WAVES>
__device__ __forceinline__ void stage_matrix(const float *__restrict__ packed, f32x4 *dst, int wave, int lane)
{
    constexpr int kPerWave = (kMatVec4 / 64) / WAVES;   // 1 KiB pieces per wave
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
        const int piece = wave * kPerWave + i;
        const float *g = packed + ((size_t)piece * 64 + lane) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(dst + piece * 64), 16, 0, 0);
    }
}

// acc (+)= W in  for the matrix staged in `w` (LDS): 64 k-steps, 256 MFMAs, fully unrolled (register indices static).
// The weight fragment of k-step s + kPrefetch is requested before the MFMAs of k-step s are issued, so the LDS
// latency never sits between a wave's MFMAs (measured: 117 -> 130 TF/s on an 11-GEMM chain, scripts/dbg/chain_bench).
constexpr int kPrefetch = 1;
// `next` != nullptr: the LDS-DMA pieces of the NEXT matrix (into `next_dst`, the other buffer) are issued between the
// k-steps, one every 64 / pieces steps, instead of in one burst after the barrier: a piece costs its wave ~60-180
// issue cycles, which hide behind an MFMA already in the pipe but add up when eight are issued back to back.
template <int WAVES>
__device__ __forceinline__ void gemm(f32x16 (&acc)[4], const f32x16 (&in)[4], const f32x4 *w, int lane,
                                     const float *__restrict__ next = nullptr, f32x4 *next_dst = nullptr, int wave = 0)
{
    constexpr int kPieces = (kMatVec4 / 64) / WAVES;   // 1 KiB pieces per wave
    constexpr int kEvery = 64 / kPieces;
    f32x4 ring[kPrefetch];
#pragma unroll
    for (int i = 0; i < kPrefetch; ++i) ring[i] = w[i * 64 + lane];
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const f32x4 a = ring[s % kPrefetch];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], in[s >> 4][s & 15], acc[0], 0, 0, 0);
        if (s + kPrefetch < 64) ring[s % kPrefetch] = w[(s + kPrefetch) * 64 + lane];
        const float b = in[s >> 4][s & 15];
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b, acc[1], 0, 0, 0);
        if (s % kEvery == 0 && next != nullptr) {
            const int piece = wave * kPieces + s / kEvery;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(next + ((size_t)piece * 64 + lane) * 4),
                                             (__attribute__((address_space(3))) void *)(next_dst + piece * 64), 16, 0, 0);
        }
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b, acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep the request-ahead order: hipcc otherwise sinks the reads to their use
    }
}

// per-feature vector (bias, LayerNorm affine, ...) in register layout: 16 broadcast float4 loads
__device__ __forceinline__ void load_feature_vec(f32x16 (&v)[4], const float *__restrict__ vec, int h)
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(vec + 32 * t + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[t][4 * q + j] = x[j];
        }
}

__device__ __forceinline__ void fill_zero(f32x16 (&v)[4])
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[t][r] = 0.f;
}

// one row of a row-major [n_rows, 128] tensor (row already clamped into range)
__device__ __forceinline__ void load_row(f32x16 (&v)[4], const float *__restrict__ src, int64_t row, int h)
{
    load_feature_vec(v, src + row * kC, h);
}

__device__ __forceinline__ void store_row(const f32x16 (&v)[4], float *__restrict__ dst, int64_t row, int h)
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 x;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = v[t][4 * q + j];
            *reinterpret_cast<f32x4 *>(dst + row * kC + 32 * t + 8 * q + 4 * h) = x;
        }
}

// LayerNorm over the 128 features of the lane's row (two-pass mean / variance), optional affine and ReLU
template <bool AFFINE, bool RELU>
__device__ __forceinline__ void layer_norm(f32x16 (&v)[4], const float *__restrict__ gamma,
                                           const float *__restrict__ beta, int h, float eps)
{
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) s[t] += v[t][r];
    float sum = (s[0] + s[1]) + (s[2] + s[3]);
    sum += xhalf(sum);
    const float mean = sum * (1.0f / kC);
    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            v[t][r] -= mean;
            q[t] = fmaf(v[t][r], v[t][r], q[t]);
        }
    float sq = (q[0] + q[1]) + (q[2] + q[3]);
    sq += xhalf(sq);
    const float rstd = rsqrtf(sq * (1.0f / kC) + eps);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (AFFINE) {
                g = *reinterpret_cast<const f32x4 *>(gamma + 32 * t + 8 * qd + 4 * h);
                b = *reinterpret_cast<const f32x4 *>(beta + 32 * t + 8 * qd + 4 * h);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = v[t][4 * qd + j] * rstd;
                if (AFFINE) y = fmaf(y, g[j], b[j]);
                v[t][4 * qd + j] = RELU ? fmaxf(y, 0.f) : y;
            }
        }
}

// rows-per-block choice: WAVES waves x 32 rows.  A block always stages whole matrices, so fewer waves per block only
// pays when the launch would otherwise leave CUs idle or the last round of blocks mostly empty.
__host__ inline int pick_waves(int64_t n_rows, int n_cus = 256)
{
    int best = 8;
    double best_cost = 1e300;
    for (int w = 8; w >= 2; w >>= 1) {
        const int64_t blocks = (n_rows + 32 * w - 1) / (32 * w);
        const int64_t rounds = (blocks + n_cus - 1) / n_cus;
        const double cost = (double)rounds * (w > 4 ? 2.0 : 1.08);   // 2 waves / SIMD hide the epilogues
        if (cost < best_cost - 1e-9) { best_cost = cost; best = w; }
    }
    return best;
}

}  // namespace qcn_chain
