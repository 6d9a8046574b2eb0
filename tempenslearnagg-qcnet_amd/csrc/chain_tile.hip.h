// Feature-major fp32-MFMA chain machinery (gfx950) for the kernels whose rows are independent 128-wide vectors and
// whose work is a CHAIN of 128x128 linear layers with per-row epilogues between them: FourierEmbedding (fourier.hip)
// and the AttentionLayer node-side blocks (nodeops.hip).
//
// Every GEMM is computed transposed, Y^T = W X^T, on the matrix cores, in one of three arithmetic modes (the
// `matmul_mode` argument of the entry points, include/qcnet_hip.h QCN_MATMUL_*):
//   exact : v_mfma_f32_32x32x2_f32 — exact fp32 fma chains (bitwise an fmaf loop), 64 cycles per 32x32x2 tile step;
//   split : every fp32 operand is carried as the unevaluated sum hi + lo of two f16 numbers (22 significant bits) after
//           an exact power-of-two scaling that puts the largest |element| of the row (resp. of the weight matrix) into
//           [2^14, 2^15), and the product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation
//           (the dropped lo*lo term is below 2^-22 of the product): three 32-cycle MFMAs per 16 k instead of eight
//           64-cycle ones.  Element-wise operand error <= max(2^-22 |x|, 2^-29 max|x|) — the result is within a few
//           fp32 ulps of the exact mode (tests/test_gpu_attention.py::test_split_matmul_error), at ~3x the rate.
//   bf16  : every operand rounded to bf16 (round to nearest even; weights on the host, activations by v_cvt_pk_bf16_f32),
//           ONE v_mfma_f32_32x32x16_bf16 per 16 k, fp32 accumulation — BASELINE.json configs[3] "bf16 projections".
//           Biases, rank-1 terms, LayerNorm / ReLU / sigmoid / residuals stay fp32.  A packed matrix is half the size
//           (32 KiB: it occupies the first half of its 64 KiB slot in global memory and in LDS).
// In all modes:
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
#include <cstdlib>

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

// Phase skew.  One barrier per matrix keeps the eight waves of a block in lockstep: all of them run a GEMM (two waves per
// SIMD queue on one MFMA pipe while the vector ALU idles), then all of them run the per-row epilogue (LayerNorm, operand
// split, sincos: the MFMA pipe idles).  With the skew the block runs as two groups half a phase apart — every wave
// takes a barrier before AND after each GEMM, group B takes one extra barrier at the start and group A one at the end —
// so that at any time one wave of a SIMD is in its GEMM while the other one is in its epilogue:
//   slot      0    1    2    3    4         A: V0 | G0 | V1 | G1 | ...
//   group A   V0   G0   V1   G1   V2        B:  - | V0 | G0 | V1 | G1 ...
// Buffer hazards hold at slot granularity: matrix i is read in slots 2i+1 (A) and 2i+2 (B); the pieces of matrix i+1
// are issued inside those two GEMMs into the other buffer (last read in slot 2i by B) and have landed (vmcnt(0) before
// every barrier) when A starts G(i+1) in slot 2i+3.  Only blocks of eight waves are skewed (waves w and w + 4 share a SIMD).
// skew: 0 off, 1 group B = waves 4..7, 2 group B = odd waves (QCN_CHAIN_SKEW, for A/B runs).
// Measured (same box, 800 k rows): 3-4 % SLOWER for Fourier / attn_post in mode 1, up to 15 % slower in mode 2 — the lockstep
// is not what holds these kernels back (neither are the LDS weight reads: a build without them gains 6 %), so the default
// is off; the knob stays for A/B runs.
struct Skew {
    bool on, grp_b;
};

template <int WAVES>
__device__ __forceinline__ Skew make_skew(int mode, int wave)
{
    Skew s;
    s.on = WAVES == 8 && mode != 0;
    s.grp_b = mode == 2 ? (wave & 1) != 0 : wave >= 4;
    return s;
}

// the skew's extra barriers (every wave of a block takes the same number: 2 per GEMM + 1)
__device__ __forceinline__ void skew_begin(const Skew &s) { if (s.on && s.grp_b) sync_for_stage(); }
__device__ __forceinline__ void skew_after_gemm(const Skew &s) { if (s.on) sync_for_stage(); }
__device__ __forceinline__ void skew_end(const Skew &s) { if (s.on && !s.grp_b) sync_for_stage(); }

__host__ inline int chain_skew_mode(int dflt)
{
    if (const char *env = qcn_knob("QCN_CHAIN_SKEW")) { const int m = atoi(env); if (m >= 0 && m <= 2) return m; }
    return dflt;
}

constexpr int kModeExact = QCN_MATMUL_EXACT_F32, kModeSplit = QCN_MATMUL_SPLIT_F16, kModeBf16 = QCN_MATMUL_BF16;

// 1 KiB LDS-DMA pieces of one packed matrix (bf16: half a slot)
template <int MODE> constexpr int mat_pieces() { return MODE == kModeBf16 ? kMatVec4 / 64 / 2 : kMatVec4 / 64; }

// Start the LDS-DMA copy of one packed matrix into LDS buffer `dst`; every wave of the block copies WAVES-th of it.
template <int WAVES, int MODE>
__device__ __forceinline__ void stage_matrix(const float *__restrict__ packed, f32x4 *dst, int wave, int lane)
{
    constexpr int kPerWave = mat_pieces<MODE>() / WAVES;   // 1 KiB pieces per wave
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
        const int piece = wave * kPerWave + i;
        const float *g = packed + ((size_t)piece * 64 + lane) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(dst + piece * 64), 16, 0, 0);
    }
}

// Per-feature vectors (biases, LayerNorm affines, frequencies, scale pairs) live in LDS for the block's lifetime: every
// epilogue reads them as two-address broadcast ds_read_b128 (~100 cycles) instead of 16 global loads per vector whose
// L2 latency sat exposed between the GEMMs (registers are full, nothing can be requested ahead): a build without these
// loads ran 8-13 % (Fourier) / 4-6 % (node kernels) faster.  n_floats is a multiple of 4, both pointers 16-byte aligned.
template <int WAVES>
__device__ __forceinline__ void stage_vectors(const float *__restrict__ src, float *dst, int n_floats, int tid)
{
    for (int i = 4 * tid; i < n_floats; i += 4 * 64 * WAVES)
        *reinterpret_cast<f32x4 *>(dst + i) = *reinterpret_cast<const f32x4 *>(src + i);
}

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// MFMA B operand of one row vector.  exact: the 64 fp32 registers themselves.  split: 2 x 32 registers of packed f16
// (hi / lo planes) + the row's power-of-two scale and its inverse.  bf16: 32 registers of packed bf16.
template <int MODE> struct Operand;
template <> struct Operand<kModeExact> { f32x16 v[4]; };
template <> struct Operand<kModeSplit> { half8 hi[8], lo[8]; float sc, usc; };
template <> struct Operand<kModeBf16> { bf16x8 v[8]; };

// x - (float)hh[k] in ONE instruction: v_fma_mix_f32 widens the f16 half on the fly (the compiler emits v_cvt_f32_f16 +
// v_sub_f32 for the plain expression); exact either way (the difference is representable), so the bits do not change
__device__ __forceinline__ float lo_of(float x, half2v hh, int k)
{
#ifdef QCN_NO_MIX
    return x - (float)hh[k];
#endif
    float r;
    if (k == 0)
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(x));
    else
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(x));
    return r;
}

__device__ __forceinline__ void make_operand(Operand<kModeExact> &o, const f32x16 (&in)[4])
{
#pragma unroll
    for (int t = 0; t < 4; ++t) o.v[t] = in[t];
}

__device__ __forceinline__ void make_operand(Operand<kModeSplit> &o, const f32x16 (&in)[4])
{
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) mx[t] = fmaxf(mx[t], fabsf(in[t][r]));
    float m = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
    m = fmaxf(m, xhalf(m));
    // m = f * 2^(eb - 126), f in [1/2, 1): scale by 2^(15 - (eb - 126)) so that the row maximum lands in [2^14, 2^15).
    // eb is clamped to [64, 190]: rows below 2^-62 keep a fixed scale, rows of 2^65 and above overflow f16 to inf.
    int eb = (__builtin_bit_cast(int, m) >> 23) & 0xff;
    eb = eb < 64 ? 64 : (eb > 190 ? 190 : eb);
    o.sc = __builtin_bit_cast(float, (268 - eb) << 23);
    o.usc = __builtin_bit_cast(float, (eb - 14) << 23);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = 8 * (j & 1) + 2 * p;
            const float x0 = in[j >> 1][r] * o.sc, x1 = in[j >> 1][r + 1] * o.sc;
            const half2v hh = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
            const float r0 = lo_of(x0, hh, 0), r1 = lo_of(x1, hh, 1);        // exact in fp32
            const half2v ll = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(r0, r1));
            o.hi[j][2 * p] = hh[0]; o.hi[j][2 * p + 1] = hh[1];
            o.lo[j][2 * p] = ll[0]; o.lo[j][2 * p + 1] = ll[1];
        }
}

// bf16: round to nearest even (v_cvt_pk_bf16_f32); register 8 (j & 1) + i of v[j >> 1] -> element i of fragment j, the k
// order of the split mode's hi plane
__device__ __forceinline__ void make_operand(Operand<kModeBf16> &o, const f32x16 (&in)[4])
{
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[j][i] = (__bf16)in[j >> 1][8 * (j & 1) + i];
}

// Operand of a row vector whose magnitude has a STATIC bound (LayerNorm outputs: |(x - mean) rstd| <= sqrt(127), so
// |y| <= sqrt(127) max|gamma| + max|beta|; sines / cosines; a hidden layer bounded through the L1 norms of its weight rows).
// The host folds the power-of-two scale `sc` that puts that bound at 2^15 into the parameters that produce the vector
// (gamma * sc, beta * sc, ...), so `in` arrives already scaled: no row-maximum scan, no scaling multiply — the split is
// cvt_pkrtz + one subtraction (against the widened hi) + cvt_pkrtz per pair, under half the vector-ALU work of the dynamic form.
// Precision: the worst-case bound sits a few binades above the typical row maximum, which only lowers the absolute floor
// of the lo plane (f16 subnormals: 2^-25 in scaled units, i.e. <= 2^-35 of the row maximum).  `sc` / `usc` are carried for
// the generic gemm() (S = sc * matrix scale).
__device__ __forceinline__ void make_operand_static(Operand<kModeExact> &o, const f32x16 (&in)[4], float, float)
{
    make_operand(o, in);
}

__device__ __forceinline__ void make_operand_static(Operand<kModeBf16> &o, const f32x16 (&in)[4], float, float)
{
    make_operand(o, in);
}

__device__ __forceinline__ void make_operand_static(Operand<kModeSplit> &o, const f32x16 (&in)[4], float sc, float usc)
{
    o.sc = sc;
    o.usc = usc;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = 8 * (j & 1) + 2 * p;
            const float x0 = in[j >> 1][r], x1 = in[j >> 1][r + 1];
            const half2v hh = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
            const float r0 = lo_of(x0, hh, 0), r1 = lo_of(x1, hh, 1);        // exact in fp32
            const half2v ll = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(r0, r1));
            o.hi[j][2 * p] = hh[0]; o.hi[j][2 * p + 1] = hh[1];
            o.lo[j][2 * p] = ll[0]; o.lo[j][2 * p + 1] = ll[1];
        }
}

template <int WAVES, int MODE>
__device__ __forceinline__ void stage_piece(const float *__restrict__ next, f32x4 *next_dst, int wave, int lane, int i)
{
    constexpr int kPieces = mat_pieces<MODE>() / WAVES;   // 1 KiB pieces per wave
    const int piece = wave * kPieces + i;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(next + ((size_t)piece * 64 + lane) * 4),
                                     (__attribute__((address_space(3))) void *)(next_dst + piece * 64), 16, 0, 0);
}

// acc += W in  for the matrix staged in `w` (LDS), fully unrolled (register indices static).  `mscale` = {sw, uw}: the
// matrix' power-of-two scale and its inverse (split mode; ignored in exact mode).
// `next` != nullptr: the LDS-DMA pieces of the NEXT matrix (into `next_dst`, the other buffer) are issued between the
// k-steps instead of in one burst after the barrier: a piece costs its wave ~60-180 issue cycles, which hide behind an
// MFMA already in the pipe but add up when eight are issued back to back.
// exact: 64 k-steps x 4 tiles; the weight fragment of k-step s + 1 is requested before the MFMAs of k-step s are all
// issued, so the LDS latency never sits between a wave's MFMAs.
template <int WAVES>
__device__ __forceinline__ void gemm(f32x16 (&acc)[4], const Operand<kModeExact> &in, const f32x4 *w,
                                     const float *__restrict__, int lane, const float *__restrict__ next = nullptr,
                                     f32x4 *next_dst = nullptr, int wave = 0)
{
    constexpr int kPieces = (kMatVec4 / 64) / WAVES;
    constexpr int kEvery = 64 / kPieces;
    f32x4 nxt = w[lane];
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const f32x4 a = nxt;
        const float b = in.v[s >> 4][s & 15];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b, acc[0], 0, 0, 0);
        if (s + 1 < 64) nxt = w[(s + 1) * 64 + lane];
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b, acc[1], 0, 0, 0);
        if (s % kEvery == 0 && next != nullptr) stage_piece<WAVES, kModeExact>(next, next_dst, wave, lane, s / kEvery);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b, acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep the request-ahead order: hipcc otherwise sinks the reads to their use
    }
}

// split: 8 k16-steps x 4 tiles x (hi*hi + hi*lo + lo*hi).  LDS image of a matrix: [k16-step j][tile t][plane p][lane]
// 16-byte fragments (8 f16: out = 32 t + (lane & 31), k = the 8 features of registers 8 (j & 1) .. + 7 of v[j >> 1]).
// The accumulators are moved to the operands' scale and back by two exact power-of-two multiplies.
// SCALE = false: the accumulators are taken and left in the operands' scale (gemm_add below starts them at zero and folds the
// way back into the fma that adds the bias).
// ZERO: the accumulators start at zero — the first MFMA of each takes the constant, nothing is zeroed beforehand.
template <int WAVES, bool SCALE, bool ZERO = false>
__device__ __forceinline__ void gemm_split_core(f32x16 (&acc)[4], const Operand<kModeSplit> &in, const f32x4 *w,
                                                const float *__restrict__ mscale, int lane, const float *__restrict__ next,
                                                f32x4 *next_dst, int wave)
{
    constexpr int kPieces = (kMatVec4 / 64) / WAVES;
    constexpr int kEvery = 32 / kPieces > 0 ? 32 / kPieces : 1;
    constexpr int kPerSlot = kPieces > 32 ? kPieces / 32 : 1;
    if constexpr (SCALE) {
        const float S = in.sc * mscale[0];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] *= S;
    }
    f32x4 nh = w[lane], nl = w[64 + lane];
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#ifdef QCN_GEMM_PRIO
    __builtin_amdgcn_s_setprio(QCN_GEMM_PRIO);
#endif
#pragma unroll
    for (int g = 0; g < 32; ++g) {            // g = 4 j + t
        const int j = g >> 2, t = g & 3;
        const half8 ah = __builtin_bit_cast(half8, nh), al = __builtin_bit_cast(half8, nl);
#ifndef QCN_EXP_NO_MFMA
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in.hi[j], ZERO && j == 0 ? zero : acc[t], 0, 0, 0);
#else
        acc[t][g & 15] += (float)ah[0] * (float)in.hi[j][0] + (float)al[1] * (float)in.lo[j][1];
#endif
#ifndef QCN_EXP_NO_LDS
        if (g + 1 < 32) { nh = w[(2 * (g + 1)) * 64 + lane]; nl = w[(2 * (g + 1) + 1) * 64 + lane]; }
#else
        if (g + 1 < 32) { nh[g & 3] += 1.0f; nl[g & 3] += 1.0f; }
#endif
#ifndef QCN_EXP_NO_MFMA
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in.lo[j], acc[t], 0, 0, 0);
#endif
#ifndef QCN_EXP_NO_STAGE
        if (g % kEvery == 0 && next != nullptr) {
#pragma unroll
            for (int i = 0; i < kPerSlot; ++i) stage_piece<WAVES, kModeSplit>(next, next_dst, wave, lane, (g / kEvery) * kPerSlot + i);
        }
#endif
#ifndef QCN_EXP_NO_MFMA
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, in.hi[j], acc[t], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef QCN_GEMM_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (SCALE) {
        const float U = in.usc * mscale[1];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] *= U;
    }
}

template <int WAVES>
__device__ __forceinline__ void gemm(f32x16 (&acc)[4], const Operand<kModeSplit> &in, const f32x4 *w,
                                     const float *__restrict__ mscale, int lane, const float *__restrict__ next = nullptr,
                                     f32x4 *next_dst = nullptr, int wave = 0)
{
    gemm_split_core<WAVES, true>(acc, in, w, mscale, lane, next, next_dst, wave);
}

// bf16: 8 k16-steps x 4 tiles, one MFMA each.  LDS image of a matrix: [k16-step j][tile t][lane] 16-byte fragments (8 bf16:
// out = 32 t + (lane & 31), k = the 8 features of registers 8 (j & 1) .. + 7 of v[j >> 1]) = 32 KiB.  The four fragments of
// k16-step j + 1 are requested before the MFMAs of step j issue (registers are plentiful in this mode: the operand is 32).
template <int WAVES>
__device__ __forceinline__ void gemm(f32x16 (&acc)[4], const Operand<kModeBf16> &in, const f32x4 *w,
                                     const float *__restrict__, int lane, const float *__restrict__ next = nullptr,
                                     f32x4 *next_dst = nullptr, int wave = 0)
{
    constexpr int kPieces = mat_pieces<kModeBf16>() / WAVES;     // 4 / 8 / 16
    constexpr int kPerStep = kPieces / 8 > 0 ? kPieces / 8 : 1;  // pieces issued per k16-step
    constexpr int kEvery = kPieces >= 8 ? 1 : 8 / kPieces;       // ... every kEvery-th step
    f32x4 cur[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) cur[t] = w[t * 64 + lane];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f32x4 nxt[4];
        if (j + 1 < 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) nxt[t] = w[(4 * (j + 1) + t) * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[t]), in.v[j], acc[t], 0, 0, 0);
        if (j % kEvery == 0 && next != nullptr) {
#pragma unroll
            for (int i = 0; i < kPerStep; ++i) stage_piece<WAVES, kModeBf16>(next, next_dst, wave, lane, (j / kEvery) * kPerStep + i);
        }
        if (j + 1 < 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc = W in + add, where add(t, q) gives the additive term (bias, rank-1 term ...) of registers 4 q .. 4 q + 3 of acc[t].
// exact / bf16: the accumulators start at `add` (exact mode: the fp32 fma chain of an fmaf loop that starts at the bias).
// split: they start at ZERO in the operands' scale and the term is added by the one fma per element that also scales the
// sum back — 64 fma instead of 64 moves + 2 x 64 multiplies per GEMM, a sixth of the vector-ALU work of a GEMM + epilogue
// phase (the rescaling factors are powers of two, so this is the same sum with the bias added last instead of first).
template <int WAVES, int MODE, typename Add>
__device__ __forceinline__ void gemm_add(f32x16 (&acc)[4], const Operand<MODE> &in, const f32x4 *w,
                                         const float *__restrict__ mscale, int lane, Add add,
                                         const float *__restrict__ next = nullptr, f32x4 *next_dst = nullptr, int wave = 0)
{
    if constexpr (MODE == kModeSplit) {
        gemm_split_core<WAVES, false, true>(acc, in, w, mscale, lane, next, next_dst, wave);
        const float U = in.usc * mscale[1];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = add(t, q);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][4 * q + j] = fmaf(acc[t][4 * q + j], U, a[j]);
            }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = add(t, q);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][4 * q + j] = a[j];
            }
        gemm<WAVES>(acc, in, w, mscale, lane, next, next_dst, wave);
    }
}

// acc += W in with the accumulators LEFT in the operands' scale (split mode; the other modes have no scale): a sum over
// several GEMMs whose (operand scale x matrix scale) products the host made equal needs no vector-ALU work between them —
// the caller applies `acc * U + bias` once at the end.
template <int WAVES, int MODE, bool ZERO = false>
__device__ __forceinline__ void gemm_raw(f32x16 (&acc)[4], const Operand<MODE> &in, const f32x4 *w, int lane,
                                         const float *__restrict__ next = nullptr, f32x4 *next_dst = nullptr, int wave = 0)
{
    if constexpr (MODE == kModeSplit) {
        gemm_split_core<WAVES, false, ZERO>(acc, in, w, nullptr, lane, next, next_dst, wave);
    } else {
        if constexpr (ZERO) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
        gemm<WAVES>(acc, in, w, nullptr, lane, next, next_dst, wave);
    }
}

// acc = acc * u + add(t, q): the way back from the operands' scale fused with the bias / rank-1 term (one fma per element)
template <typename Add>
__device__ __forceinline__ void scale_add(f32x16 (&acc)[4], float u, Add add)
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = add(t, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][4 * q + j] = fmaf(acc[t][4 * q + j], u, a[j]);
        }
}

// the additive term of gemm_add for a per-feature vector (bias) in LDS / global memory; nullptr: zero
struct BiasTerm {
    const float *vec;
    int h;
    __device__ __forceinline__ f32x4 operator()(int t, int q) const
    {
        if (vec == nullptr) return f32x4{0.f, 0.f, 0.f, 0.f};
        return *reinterpret_cast<const f32x4 *>(vec + 32 * t + 8 * q + 4 * h);
    }
};

// per-feature vector (bias, LayerNorm affine, ...) in register layout: 16 broadcast float4 loads
__device__ __forceinline__ void load_feature_vec(f32x16 (&v)[4], const float *__restrict__ vec, int h)
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#ifdef QCN_EXP_NO_VEC
            const f32x4 x = {0.25f * t, 0.5f * q, 1.0f * h, 2.0f};
#else
            const f32x4 x = *reinterpret_cast<const f32x4 *>(vec + 32 * t + 8 * q + 4 * h);
#endif
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
#ifdef QCN_EXP_NO_ROW
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[t][r] = 0.125f * (float)(r + 16 * t) + (float)(row & 31) + (float)h;
#else
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(src + row * kC + 32 * t + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[t][4 * q + j] = x[j];
        }
#endif
}

// Coalesced row I/O through a per-wave LDS transposition block (4 KiB = the wave's 32 rows x one 32-feature block).
// load_row (and the per-lane stores it used to be paired with) lets every lane touch its own 512-B row: one instruction =
// 32 lines of 32 B.  Here a global
// instruction covers 8 rows x 128 B (full lines: lane L -> row 8 i + (L >> 3), 16-B chunk L & 7) and the lane-owns-row
// register layout is produced by the LDS: chunk c of row n sits at chunk position c ^ ((n >> 1) & 7) of the row's
// 128 bytes, which makes both the row-major writes and the (row n, chunks 2 q + h) reads conflict-free.  DS operations
// of one wave execute in order, so no barrier is needed between the two halves.
// `row0` = first row of the wave's tile; rows >= n_rows are clamped on load and masked on store.
__device__ __forceinline__ void load_tile(f32x16 (&v)[4], const float *__restrict__ src, int64_t row0, int64_t n_rows,
                                          f32x4 *xpose, int lane)
{
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3), c = lane & 7;
            int64_t row = row0 + r;
            row = row < n_rows ? row : n_rows - 1;
            g[i] = *reinterpret_cast<const f32x4 *>(src + row * kC + 32 * t + 4 * c);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3), c = lane & 7;
            xpose[r * 8 + (c ^ ((r >> 1) & 7))] = g[i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 x = xpose[n * 8 + ((2 * q + h) ^ ((n >> 1) & 7))];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[t][4 * q + j] = x[j];
        }
    }
}

__device__ __forceinline__ void store_tile(const f32x16 (&v)[4], float *__restrict__ dst, int64_t row0, int64_t n_rows,
                                           f32x4 *xpose, int lane)
{
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 x;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = v[t][4 * q + j];
            xpose[n * 8 + ((2 * q + h) ^ ((n >> 1) & 7))] = x;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3), c = lane & 7;
            const f32x4 x = xpose[r * 8 + (c ^ ((r >> 1) & 7))];
            if (row0 + r < n_rows) *reinterpret_cast<f32x4 *>(dst + (row0 + r) * kC + 32 * t + 4 * c) = x;
        }
    }
}

// The same with a 2 KiB block per wave (32 rows x 16 features: half a feature block per pass) for the kernels whose
// per-feature vectors leave less than 4 KiB per wave: a global instruction covers 16 rows x 64 B (lane L -> row
// 16 i + (L >> 2), chunk L & 3), chunk c of row n sits at chunk position c ^ ((n >> 2) & 3) of the row's 64 bytes.
__device__ __forceinline__ void load_tile2(f32x16 (&v)[4], const float *__restrict__ src, int64_t row0, int64_t n_rows,
                                           f32x4 *xpose, int lane)
{
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tp = 0; tp < 8; ++tp) {           // pass = (feature block t, half)
        const int t = tp >> 1, half = tp & 1;
        f32x4 g[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 16 * i + (lane >> 2), c = lane & 3;
            int64_t row = row0 + r;
            row = row < n_rows ? row : n_rows - 1;
            g[i] = *reinterpret_cast<const f32x4 *>(src + row * kC + 32 * t + 16 * half + 4 * c);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 16 * i + (lane >> 2), c = lane & 3;
            xpose[r * 4 + (c ^ ((r >> 2) & 3))] = g[i];
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const f32x4 x = xpose[n * 4 + ((2 * qq + h) ^ ((n >> 2) & 3))];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[t][4 * (2 * half + qq) + j] = x[j];
        }
    }
}

__device__ __forceinline__ void store_tile2(const f32x16 (&v)[4], float *__restrict__ dst, int64_t row0, int64_t n_rows,
                                            f32x4 *xpose, int lane)
{
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tp = 0; tp < 8; ++tp) {
        const int t = tp >> 1, half = tp & 1;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            f32x4 x;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = v[t][4 * (2 * half + qq) + j];
            xpose[n * 4 + ((2 * qq + h) ^ ((n >> 2) & 3))] = x;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 16 * i + (lane >> 2), c = lane & 3;
            const f32x4 x = xpose[r * 4 + (c ^ ((r >> 2) & 3))];
            if (row0 + r < n_rows) *reinterpret_cast<f32x4 *>(dst + (row0 + r) * kC + 32 * t + 16 * half + 4 * c) = x;
        }
    }
}

// The row as split f16 (the operand format of qcn_attn_modes_f32, see qcn_split_rows_f16): dst row = [hi 128 | lo 128] with
// hi + lo = v * scale, through the 2 KiB transposition block: per pass the 16 features of a row are 32 B of hi and 32 B of lo
// halves (chunks 0, 1 | 2, 3 of the row's 64 bytes, XOR-swizzled as in store_tile2); a global instruction then writes
// 16 rows x (32 B hi + 32 B lo) instead of 8-byte pieces.
__device__ __forceinline__ void store_tile2_split(const f32x16 (&v)[4], _Float16 *__restrict__ dst, int64_t row0, int64_t n_rows,
                                                  f32x4 *xpose, int lane, float scale)
{
    using half4v = __attribute__((ext_vector_type(4))) _Float16;
    const int n = lane & 31, h = lane >> 5;
    half4v *const xh = reinterpret_cast<half4v *>(xpose);          // 8-byte units: row n = units 8 n .. 8 n + 7
#pragma unroll
    for (int tp = 0; tp < 8; ++tp) {
        const int t = tp >> 1, half = tp & 1;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            half4v hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = v[t][4 * (2 * half + qq) + j] * scale;
                hi[j] = (_Float16)a;
                lo[j] = (_Float16)(a - (float)hi[j]);
            }
            xh[n * 8 + 2 * (qq ^ ((n >> 2) & 3)) + h] = hi;
            xh[n * 8 + 2 * ((2 + qq) ^ ((n >> 2) & 3)) + h] = lo;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 16 * i + (lane >> 2), c = lane & 3;
            const f32x4 x = xpose[r * 4 + (c ^ ((r >> 2) & 3))];
            if (row0 + r < n_rows)
                *reinterpret_cast<f32x4 *>(dst + (row0 + r) * (2 * kC) + (c >> 1) * kC + 32 * t + 16 * half + 8 * (c & 1)) = x;
        }
    }
}

// The row as bf16 (round to nearest even; the operand format of qcn_attn_modes_bf16): dst row = 128 bf16 = 256 B, through the
// 2 KiB transposition block, one 32-feature block (64 B per row) per pass: lane (n, h) writes its four 8-byte groups
// (features 32 t + 8 q + 4 h ..) at 16-B chunk q ^ ((n >> 2) & 3) of row n, a global instruction then writes 16 rows x 64 B.
__device__ __forceinline__ void store_tile2_bf16(const f32x16 (&v)[4], __bf16 *__restrict__ dst, int64_t row0, int64_t n_rows,
                                                 f32x4 *xpose, int lane)
{
    using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
    const int n = lane & 31, h = lane >> 5;
    bf16x4 *const xh = reinterpret_cast<bf16x4 *>(xpose);          // 8-byte units: row n = units 8 n .. 8 n + 7
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x4 b;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = (__bf16)v[t][4 * q + j];
            xh[n * 8 + 2 * (q ^ ((n >> 2) & 3)) + h] = b;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 16 * i + (lane >> 2), c = lane & 3;
            const f32x4 x = xpose[r * 4 + (c ^ ((r >> 2) & 3))];
            if (row0 + r < n_rows) *reinterpret_cast<f32x4 *>(dst + (row0 + r) * kC + 32 * t + 8 * c) = x;
        }
    }
}

// LayerNorm over the 128 features of the lane's row (two-pass mean / variance), optional affine and ReLU
template <bool AFFINE, bool RELU>
__device__ __forceinline__ void layer_norm(f32x16 (&v)[4], const float *__restrict__ gamma,
                                           const float *__restrict__ beta, int h, float eps)
{
    // partial sums over the registers r = k mod 4, in the order (t, r): the order of the stepped form in chain_pair.hip.h
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r & 3] += v[t][r];
    float sum = (s[0] + s[1]) + (s[2] + s[3]);
    sum += xhalf(sum);
    const float mean = sum * (1.0f / kC);
    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[t][r] -= mean;
            q[r & 3] = fmaf(v[t][r], v[t][r], q[r & 3]);
        }
    float sq = (q[0] + q[1]) + (q[2] + q[3]);
    sq += xhalf(sq);
    const float rstd = rsqrtf(sq * (1.0f / kC) + eps);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
#ifdef QCN_EXP_NO_VEC
            if (AFFINE) { g = f32x4{1.5f, 0.5f * t, 1.0f + qd, 2.0f}; b = f32x4{0.5f, 0.25f * h, 0.125f, 1.0f}; }
#else
            if (AFFINE) {
                g = *reinterpret_cast<const f32x4 *>(gamma + 32 * t + 8 * qd + 4 * h);
                b = *reinterpret_cast<const f32x4 *>(beta + 32 * t + 8 * qd + 4 * h);
            }
#endif
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = v[t][4 * qd + j] * rstd;
                if (AFFINE) y = fmaf(y, g[j], b[j]);
                v[t][4 * qd + j] = RELU ? fmaxf(y, 0.f) : y;
            }
        }
}

// LayerNorm of a row whose mean is ZERO by construction: the host centred the Linear layer that produced it (rows of
// (I - 11^T / 128) W, centred bias and rank-1 column — the mean of W f + b is linear in f), so the mean pass and the
// subtraction disappear.  gamma / beta arrive pre-multiplied by the static operand scale of the consumer.
template <bool RELU>
__device__ __forceinline__ void layer_norm_centered(f32x16 (&v)[4], const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, int h, float eps)
{
    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r & 3] = fmaf(v[t][r], v[t][r], q[r & 3]);
    float sq = (q[0] + q[1]) + (q[2] + q[3]);
    sq += xhalf(sq);
    const float rstd = rsqrtf(sq * (1.0f / kC) + eps);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + 32 * t + 8 * qd + 4 * h);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + 32 * t + 8 * qd + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = fmaf(v[t][4 * qd + j] * rstd, g[j], b[j]);
                v[t][4 * qd + j] = RELU ? fmaxf(y, 0.f) : y;
            }
        }
}

// rows-per-block choice: WAVES waves x 32 rows.  A block always stages whole matrices, so fewer waves per block only
// pays when the launch would otherwise leave CUs idle or the last round of blocks mostly empty.  `small_cost` = time of
// a 4- (or 2-) wave block relative to an 8-wave block's 2.0, measured per kernel (MI355X, split mode, microseconds per
// round of blocks: attn_post 50 / 100 — its four-wave blocks keep their rows in registers —, Fourier in = 3 43 / 57,
// ln_linear x4 33 / 50).
__host__ inline int pick_waves(int64_t n_rows, double small_cost = 1.08, int n_cus = 0)
{
    if (n_cus <= 0) n_cus = qcn_num_cus();
    if (const char *env = qcn_knob("QCN_CHAIN_WAVES")) { const int w = atoi(env); if (w == 2 || w == 4 || w == 8) return w; }
    int best = 8;
    double best_cost = 1e300;
    for (int w = 8; w >= 2; w >>= 1) {
        const int64_t blocks = (n_rows + 32 * w - 1) / (32 * w);
        const int64_t rounds = (blocks + n_cus - 1) / n_cus;
        const double cost = (double)rounds * (w > 4 ? 2.0 : small_cost);   // 2 waves / SIMD hide the epilogues
        if (cost < best_cost - 1e-9) { best_cost = cost; best = w; }
    }
    return best;
}

// launch one of the <MODE, WAVES> instantiations through a caller-defined QCN_LAUNCH(M, W)
#define QCN_CHAIN_DISPATCH_W(M, waves)                           \
    do {                                                         \
        if ((waves) == 8) QCN_LAUNCH(M, 8);                      \
        else if ((waves) == 4) QCN_LAUNCH(M, 4);                 \
        else QCN_LAUNCH(M, 2);                                   \
    } while (0)
#define QCN_CHAIN_DISPATCH(mode, waves)                                                    \
    do {                                                                                   \
        if ((mode) == QCN_MATMUL_SPLIT_F16) QCN_CHAIN_DISPATCH_W(QCN_MATMUL_SPLIT_F16, waves); \
        else if ((mode) == QCN_MATMUL_BF16) QCN_CHAIN_DISPATCH_W(QCN_MATMUL_BF16, waves);  \
        else QCN_CHAIN_DISPATCH_W(QCN_MATMUL_EXACT_F32, waves);                            \
    } while (0)

__host__ inline bool valid_mode(int mode)
{
    return mode == QCN_MATMUL_EXACT_F32 || mode == QCN_MATMUL_SPLIT_F16 || mode == QCN_MATMUL_BF16;
}

}  // namespace qcn_chain
