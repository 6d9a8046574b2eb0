// Shared fp32-MFMA row-tile machinery (gfx950) for the kernels whose rows are independent 128-wide vectors:
// FourierEmbedding (fourier.hip), the AttentionLayer node-side blocks (nodeops.hip), the GRU (gru.hip).
//
// A workgroup = 4 waves x 32 rows.  A wave owns 32 COMPLETE rows: 32 x 128 outputs = four 32x32 accumulator tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chains), so LayerNorm / activations / residuals are in-register epilogues and
// the only cross-wave object is the weight stream: k-major pre-transposed [128 k][128 n] matrices, 32 k per chunk,
// double-buffered in LDS, one barrier per chunk, the next chunk's global loads in flight under 64 MFMAs per wave.
#pragma once
#include "common.hip.h"

namespace qcn_tile {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kC = 128;
constexpr int kRowsPerWave = 32;
constexpr int kRowsPerBlock = 128;
constexpr int kChunkK = 32;
constexpr int kPad = 33;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// sum over the 32 lanes of a wave half
__device__ __forceinline__ float half_sum(float v)
{
    v += dpp_mov<0xB1>(v);    // xor 1
    v += dpp_mov<0x4E>(v);    // xor 2
    v += dpp_mov<0x141>(v);   // row_half_mirror (xor 7)
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += __shfl_xor(v, 16);
    return v;
}

struct Smem {
    float a[4][kC][kPad];          // per-wave A tile, k-major
    float w[2][kChunkK][kC];       // weight chunk double buffer
    float x[4][4][kRowsPerWave];   // per-wave scratch (raw Fourier inputs)
};

// accumulator element (tile t, reg r) of a lane is  row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5),  col = 32 t + (lane & 31)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// LayerNorm (+ optional affine, + optional ReLU) over complete rows held in the accumulators
template <bool AFFINE, bool RELU>
__device__ __forceinline__ void row_layer_norm(f32x16 (&acc)[4], const float *__restrict__ gamma,
                                               const float *__restrict__ beta, int lane, float eps)
{
    float g[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        g[t] = AFFINE ? gamma[32 * t + (lane & 31)] : 1.f;
        b[t] = AFFINE ? beta[32 * t + (lane & 31)] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float mean = half_sum(acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r]) * (1.0f / kC);
        float d[4], sq = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) { d[t] = acc[t][r] - mean; sq = fmaf(d[t], d[t], sq); }
        const float rstd = rsqrtf(half_sum(sq) * (1.0f / kC) + eps);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = AFFINE ? fmaf(d[t] * rstd, g[t], b[t]) : d[t] * rstd;
            acc[t][r] = RELU ? fmaxf(v, 0.f) : v;
        }
    }
}

__device__ __forceinline__ void acc_to_tile(const f32x16 (&acc)[4], float (&tile)[kC][kPad], int lane)
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[32 * t + (lane & 31)][acc_row(r, lane)] = acc[t][r];
}

struct Stream {            // weight chunk stream state (all threads of the workgroup)
    float4 pre[4];         // prefetched chunk: 32 k x 128 n floats = 16 KiB / 256 threads = 4 float4 each
    int chunk;             // index of the chunk held in `pre`
};

__device__ __forceinline__ void prefetch_chunk(Stream &s, const float *__restrict__ wt, int tid)
{
    const float4 *src = reinterpret_cast<const float4 *>(wt);
#pragma unroll
    for (int i = 0; i < 4; ++i) s.pre[i] = src[i * 256 + tid];
}

// acc[0..3] += A(32 x 128, wave tile) * Wt(128 x 128) with Wt streamed in four 32-k chunks.  `next` = the first chunk of
// the following GEMM (or nullptr): its loads are issued before the last chunk's MFMAs.
__device__ __forceinline__ void gemm_pass(f32x16 (&acc)[4], Smem &sm, Stream &s, const float *__restrict__ wt,
                                          const float *__restrict__ next, int wave, int lane, int tid)
{
#pragma unroll 1
    for (int c = 0; c < kC / kChunkK; ++c) {
        float (&buf)[kChunkK][kC] = sm.w[s.chunk & 1];
        float4 *dst = reinterpret_cast<float4 *>(&buf[0][0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i * 256 + tid] = s.pre[i];
        __syncthreads();
        const float *nxt = (c + 1 < kC / kChunkK) ? wt + (c + 1) * kChunkK * kC : next;
        if (nxt != nullptr) prefetch_chunk(s, nxt, tid);
        s.chunk++;
        const int kbase = c * kChunkK;
#pragma unroll
        for (int ks = 0; ks < kChunkK / 2; ++ks) {
            const float av = sm.a[wave][kbase + 2 * ks + (lane >> 5)][lane & 31];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float bv = buf[2 * ks + (lane >> 5)][32 * t + (lane & 31)];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
}


// rows of a [n_rows, 128] tensor in accumulator layout (each (r, t) access: 32 lanes x 4 B = one 128-B segment)
__device__ __forceinline__ void load_rows(f32x16 (&acc)[4], const float *__restrict__ src, int64_t row0, int64_t n_rows,
                                          int lane)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int64_t row = row0 + acc_row(r, lane);
        row = row < n_rows ? row : n_rows - 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][r] = src[row * kC + 32 * t + (lane & 31)];
    }
}

__device__ __forceinline__ void store_rows(const f32x16 (&acc)[4], float *__restrict__ dst, int64_t row0, int64_t n_rows,
                                           int lane)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + acc_row(r, lane);
        if (row < n_rows) {
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[row * kC + 32 * t + (lane & 31)] = acc[t][r];
        }
    }
}

__device__ __forceinline__ void fill_cols(f32x16 (&acc)[4], const float *__restrict__ vec, int lane)
{
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float b = vec ? vec[32 * t + (lane & 31)] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b;
    }
}

}  // namespace qcn_tile
