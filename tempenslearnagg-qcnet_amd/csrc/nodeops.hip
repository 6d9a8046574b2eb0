// A2 node side — everything of AttentionLayer.forward (layers/attention_layer.py:70-85,103-122) that is per NODE
// (the per-edge part is attention*.hip), as two launches per layer instead of ~15 library calls:
//
//   qcn_ln_linear_f32  : xn = LN(x);  y_j = W_j xn + b_j  for up to four 128-wide projections (q / k / v / s), xn kept
//                        for the gate (reference :74-79, :115-117, to_s of :108).
//   qcn_attn_post_f32  : g = sigmoid(W_g [agg, xn] + b_g);  u = agg + g * (s - agg)              (update, :106-108)
//                        x1 = x + LN_post(W_out u + b_out)                                        (:119, :83)
//                        out = x1 + LN_ffpost(W_2 relu(W_1 LN_ffpre(x1) + b_1) + b_2)             (:84, :121-122)
//
// Both run on the feature-major chain machinery of chain_tile.hip.h: a lane owns one row, a layer's output registers
// are the next layer's MFMA operand, LDS holds only the double-buffered packed weights (LDS-DMA staged), every
// LayerNorm / sigmoid / residual is an in-lane epilogue, the 512-wide FFN hidden layer is produced and consumed in
// four 128-wide blocks and never leaves the registers.  HBM traffic per row: reads of agg, xn, s, x and one write
// (2.5 KiB) against 11 x 32 KiB of flops: MFMA-bound (fp32 matrix cores, exact fp32 fma chains).
#include "chain_pair.hip.h"
#include <cstdio>

namespace {

using namespace qcn_chain;

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void ln_linear_kernel(const float *__restrict__ x, int64_t n_rows,
                                                               const float *__restrict__ ln_g,
                                                               const float *__restrict__ ln_b,
                                                               const float *__restrict__ wp,
                                                               const float *__restrict__ bias, int n_out,
                                                               float *__restrict__ y0, float *__restrict__ y1,
                                                               float *__restrict__ y2, float *__restrict__ y3,
                                                               float *__restrict__ xn_out)
{
    // two matrix buffers + one 4 KiB transposition block per wave (load_tile / store_tile), in ONE __shared__ array
    // (+ the biases, 4 x 128 floats: gemm_add reads them right behind a GEMM, where a global load would sit exposed)
    // (eight-wave blocks have no LDS left for them — 160 KiB — and keep the accumulators starting at the bias)
    constexpr bool kLdsBias = WAVES <= 4;
    __shared__ __attribute__((aligned(16))) f32x4 wlds[2 * kMatVec4 + WAVES * 256 + (kLdsBias ? kC : 0)];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * (WAVES * kRowsPerWave) + wave * kRowsPerWave;
    f32x4 *const xpose = wlds + 2 * kMatVec4 + wave * 256;
    float *const sbias = reinterpret_cast<float *>(wlds + 2 * kMatVec4 + WAVES * 256);

    stage_matrix<WAVES, MODE>(wp, wlds, wave, lane);
    if (kLdsBias && bias != nullptr) stage_vectors<WAVES>(bias, sbias, n_out * kC, tid);   // visible after the first sync_for_stage
    f32x16 xv[4];
    load_tile(xv, x, row0, n_rows, xpose, lane);
    if (ln_g != nullptr) layer_norm<true, false>(xv, ln_g, ln_b, h, 1e-5f);
    if (xn_out != nullptr) store_tile(xv, xn_out, row0, n_rows, xpose, lane);
    const float *mscale = wp + (size_t)n_out * kMatFloats;   // {scale, 1 / scale} per matrix
    Operand<MODE> xo;
    make_operand(xo, xv);
#pragma unroll 1
    for (int j = 0; j < n_out; ++j) {
        f32x16 acc[4];
        if constexpr (kLdsBias) {
            sync_for_stage();   // matrix j has landed (every wave drained its DMA), buffer (j+1)&1 is no longer read
            gemm_add<WAVES, MODE>(acc, xo, wlds + (j & 1) * kMatVec4, mscale + 2 * j, lane,
                                  BiasTerm{bias != nullptr ? sbias + j * kC : nullptr, h},
                                  j + 1 < n_out ? wp + (size_t)(j + 1) * kMatFloats : nullptr, wlds + ((j + 1) & 1) * kMatVec4, wave);
        } else {
            if (bias != nullptr) load_feature_vec(acc, bias + j * kC, h);
            else fill_zero(acc);
            sync_for_stage();
            gemm<WAVES>(acc, xo, wlds + (j & 1) * kMatVec4, mscale + 2 * j, lane,
                        j + 1 < n_out ? wp + (size_t)(j + 1) * kMatFloats : nullptr, wlds + ((j + 1) & 1) * kMatVec4, wave);
        }
        float *y = j == 0 ? y0 : j == 1 ? y1 : j == 2 ? y2 : y3;
        store_tile(acc, y, row0, n_rows, xpose, lane);
    }
}

// pack: 11 chain-packed matrices in consumption order [Wg_agg, Wg_x, Wout, W1_0, W2_0, W1_1, W2_1, W1_2, W2_2, W1_3, W2_3]
// then plain vectors [bg, bout, lnpost_g, lnpost_b, lnffpre_g, lnffpre_b, b1 (512), b2, lnffpost_g, lnffpost_b,
// lnpre_dst_g, lnpre_dst_b] then {scale, 1 / scale} per matrix (32 floats reserved)
constexpr int kPostMats = 11;
constexpr int kVBg = 0, kVBout = kC, kVLpG = 2 * kC, kVLpB = 3 * kC, kVLfG = 4 * kC, kVLfB = 5 * kC, kVB1 = 6 * kC,
              kVB2 = 10 * kC, kVLqG = 11 * kC, kVLqB = 12 * kC, kVPreG = 13 * kC, kVPreB = 14 * kC,
              kPostVecs = 15 * kC, kPostScales = 32;

// Chained form (qcn_attn_post_chain_f32): to_s is computed here from the LN(x_dst) the gate needs anyway (the s projection
// never goes through HBM), and the projections the NEXT layer applies to this layer's output (its LayerNorm + to_q, or
// to_q / to_k / to_v) are applied to the output row while it is still in registers.
// pack: 12 matrices [Wg_agg, Wg_x, Ws, Wout, W1_0, W2_0, ...], the vectors above, then bs, then the scale table.
constexpr int kChainMats = 12, kVBs = kPostVecs, kChainVecs = kPostVecs + kC, kNextVecs = 5 * kC;   // next: ln_g, ln_b, 3 biases

struct PostNext {
    const float *wt;        // n chain-packed matrices + their {scale, 1 / scale} pairs (the qcn_ln_linear_f32 layout)
    const float *bias;      // [n, 128] or nullptr
    const float *ln_g, *ln_b;
    float *y[3];
    int n;                  // 0: no next-layer projections
    float sc;               // split mode: static operand scale of LN_next's output (a power of two; <= 0: row maxima instead)
};

#ifdef QCN_CHAIN_STAMPS
// diagnostic build only (scripts/dbg/build_variant.sh stamps -DQCN_CHAIN_STAMPS): cycles of block 0 / waves 0 and 4 per
// phase class {epilogue + loads, barrier wait, GEMM}, read back and printed by the entry point
__device__ unsigned long long g_post_stamps[2][4];
#define PSTAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_t[k] += now_ - last_; last_ = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define PSTAMP(k)
#endif


// ---- feed-forward block with the hidden blocks software-pipelined INSIDE a tile (blocks of up to four waves: one wave per
// SIMD, 512 registers).  The four first-layer GEMMs W1_hb read the same operand and do not depend on each other, so the
// epilogue of hidden block hb (bias, ReLU, row maximum, operand split: vector ALU) is cut into steps and issued between
// the MFMAs of the NEXT block's first-layer GEMM instead of behind its own:
//     W1_0 | W1_1 + E_0 | W2_0 | W1_2 + E_1 | W2_1 | W1_3 + E_2 | W2_2 | E_3 | W2_3
// Same arithmetic, element for element, as gemm_add -> ReLU -> make_operand of the sequential form (bit-identical rows).
struct HidSteps {
    float mx[4];
    float red, red2, sc;
    f32x4 b0, b1;
};

// LDS reads of step g (issued one slot ahead): the bias groups 2 g, 2 g + 1
__device__ __forceinline__ void hid_load(HidSteps &S, const float *bias, int h, int g)
{
    if (g < 8) {
        S.b0 = lds_vec4(bias, (2 * g) >> 2, (2 * g) & 3, h);
        S.b1 = lds_vec4(bias, (2 * g + 1) >> 2, (2 * g + 1) & 3, h);
    }
}

// steps 0..7: y = relu(a * U + b) (two groups per step) and four running maxima; 8, 9: row maximum -> power-of-two scale
// (make_operand's rule); 10..25: one group per step scaled and split into the operand
// STATIC (the chained form): U and the bias carry the hidden block's static operand scale (host: ops.pack_attn_post_chain),
// so y arrives scaled — no row maximum, no scaling multiply: steps 0..7 y = relu(a * U + b), 10..25 the split.
template <bool STATIC = false>
__device__ __forceinline__ void hid_step(f32x16 (&a)[4], float U, Operand<kModeSplit> &hid, HidSteps &S, int g, const f32x4 c0,
                                         const f32x4 c1)
{
    if constexpr (STATIC) {
        if (g < 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = 2 * g + i, t = m >> 2, q = m & 3;
                const f32x4 c = i ? c1 : c0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float y = fmaxf(fmaf(a[t][4 * q + k], U, c[k]), 0.f);
                    pin(y);
                    a[t][4 * q + k] = y;
                }
            }
        } else if (g >= 10 && g < 26) {
            const int m = g - 10, t = m >> 2, q = m & 3;
            put4(hid, t, q, grp(a, t, q));
        }
        return;
    }
    if (g < 8) {
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) S.mx[k] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = 2 * g + i, t = m >> 2, q = m & 3;
            const f32x4 c = i ? c1 : c0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float y = fmaxf(fmaf(a[t][4 * q + k], U, c[k]), 0.f);
                pin(y);
                a[t][4 * q + k] = y;
                S.mx[k] = fmaxf(S.mx[k], y);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) pin(S.mx[k]);
    } else if (g == 8) {
        S.red = fmaxf(fmaxf(S.mx[0], S.mx[1]), fmaxf(S.mx[2], S.mx[3]));
        S.red2 = xhalf(S.red);
    } else if (g == 9) {
        const float m = fmaxf(S.red, S.red2);
        int eb = (__builtin_bit_cast(int, m) >> 23) & 0xff;
        eb = eb < 64 ? 64 : (eb > 190 ? 190 : eb);
        hid.sc = __builtin_bit_cast(float, (268 - eb) << 23);
        hid.usc = __builtin_bit_cast(float, (eb - 14) << 23);
        S.sc = hid.sc;
    } else if (g < 26) {
        const int m = g - 10, t = m >> 2, q = m & 3;
        put4(hid, t, q, grp(a, t, q) * S.sc);
    }
}

// acc += W in with the accumulators left in the operands' scale (gemm_raw), `load(g + 1)` and `step(g, b0, b1)` issued in
// slot g = 0..31 (chain_pair.hip.h's slot layout: first MFMA, the fragment reads two slots ahead, the LDS-DMA pieces of the
// next matrix and the step's LDS reads in one scheduling region; the step's vector work around the other two MFMAs).
template <int WAVES, bool ZERO = false, typename Load, typename Step>
__device__ __forceinline__ void gemm_steps(f32x16 (&acc)[4], const Operand<kModeSplit> &in, const f32x4 *w, int lane,
                                           const float *__restrict__ next, f32x4 *next_dst, int wave, HidSteps &S, Load load,
                                           Step step)
{
    constexpr int kPieces = (kMatVec4 / 64) / WAVES;       // 16 (four waves) / 32 (two)
    constexpr int kPerSlot = kPieces > 32 ? kPieces / 32 : 1;
    constexpr int kEvery = kPieces >= 32 ? 1 : 32 / kPieces;
    f32x4 nh = w[lane], nl = w[64 + lane], nh2 = w[2 * 64 + lane], nl2 = w[3 * 64 + lane];
    f32x16 zero;   // (ZERO: the accumulators start at zero, chain_tile.hip.h gemm_split_core)
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) {            // g = 4 j + t
        const int j = g >> 2, t = g & 3;
        const half8 ah = __builtin_bit_cast(half8, nh), al = __builtin_bit_cast(half8, nl);
        nh = nh2;
        nl = nl2;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in.hi[j], ZERO && j == 0 ? zero : acc[t], 0, 0, 0);
        if (g + 2 < 32) { nh2 = w[(2 * (g + 2)) * 64 + lane]; nl2 = w[(2 * (g + 2) + 1) * 64 + lane]; }
        if (g % kEvery == 0) {
#pragma unroll
            for (int i = 0; i < kPerSlot; ++i) stage_piece<WAVES, kModeSplit>(next, next_dst, wave, lane, (g / kEvery) * kPerSlot + i);
        }
        const f32x4 c0 = S.b0, c1 = S.b1;
        load(g + 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in.lo[j], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, in.hi[j], acc[t], 0, 0, 0);
        step(g, c0, c1);
        slot_pattern<2, 8>();
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MODE, int WAVES, bool CHAIN = false>
__global__ __launch_bounds__(64 * WAVES) void attn_post_kernel(const float *__restrict__ agg,
                                                               const float *__restrict__ xn,
                                                               const float *__restrict__ s,
                                                               const float *__restrict__ xres,
                                                               const float *__restrict__ pack, int64_t n_rows,
                                                               float *__restrict__ out, int skew_mode, PostNext next)
{
    static_assert(!CHAIN || WAVES <= 4, "the chained form holds the s projection in registers: one wave per SIMD");
    constexpr int kVecFloats = (CHAIN ? kChainVecs : kPostVecs) + kPostScales + (CHAIN ? kNextVecs : 0);
    constexpr bool kStaticNext = CHAIN && MODE == kModeSplit;
    constexpr int kMats = CHAIN ? kChainMats : kPostMats;
    constexpr int kM = CHAIN ? 1 : 0;            // matrices behind Wg_x move up by one (Ws)
    __shared__ __attribute__((aligned(16))) f32x4 wlds[2 * kMatVec4 + kVecFloats / 4 + WAVES * 128];
    f32x4 *const buf0 = wlds, *const buf1 = wlds + kMatVec4;
    // (the vectors sit behind the two matrix buffers in ONE __shared__ array: with an array of their own hipcc's LDS-DMA
    // alias tracking put a vmcnt(0) in front of every LDS read, the weight fragments of the GEMM loop included)
    float *const svec = reinterpret_cast<float *>(wlds + 2 * kMatVec4);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * (WAVES * kRowsPerWave) + wave * kRowsPerWave;
    f32x4 *const xpose = wlds + 2 * kMatVec4 + kVecFloats / 4 + wave * 128;   // load_tile2 / store_tile2
    const float *mats = pack;
    const float *vec = svec;
    const float *msc = vec + (CHAIN ? kChainVecs : kPostVecs);
    const float *nvec = msc + kPostScales;     // CHAIN: ln_g, ln_b, biases of the next layer's projections
    const float eps = 1e-5f;
    Operand<MODE> op;
    // Chained split form: LayerNorm outputs and the hidden blocks are STATIC-scale operands — the host folded their power-of-two
    // scales into gamma / beta / b1 and into the scale table (ops.pack_attn_post_chain; table slots 24.. = {s_pre, 1 / s_pre,
    // s_ff, 1 / s_ff, 1 / P}), and the four second-layer GEMMs accumulate raw in ONE scale P.
    constexpr bool kStatic = CHAIN && MODE == kModeSplit;
    // the matrices alternate between the two buffers; behind Ws the parity flips
    f32x4 *const bA = CHAIN ? wlds : wlds + kMatVec4, *const bB = CHAIN ? wlds + kMatVec4 : wlds;

    const Skew skew = make_skew<WAVES>(skew_mode, wave);
#ifdef QCN_CHAIN_STAMPS
    unsigned long long acc_t[4] = {0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
    stage_matrix<WAVES, MODE>(mats, buf0, wave, lane);
    stage_vectors<WAVES>(pack + (size_t)kMats * kMatFloats, svec, (CHAIN ? kChainVecs : kPostVecs) + kPostScales, tid);
    if constexpr (CHAIN) {
        if (next.n > 0) {
            float *nv = svec + kChainVecs + kPostScales;
            const float nsc_ln = kStaticNext && next.sc > 0.f ? next.sc : 1.0f;     // folded into gamma / beta
            for (int i = tid; i < kC; i += 64 * WAVES) { nv[i] = next.ln_g[i] * nsc_ln; nv[kC + i] = next.ln_b[i] * nsc_ln; }
            for (int i = tid; i < next.n * kC; i += 64 * WAVES) nv[2 * kC + i] = next.bias != nullptr ? next.bias[i] : 0.f;
        }
    }
    __syncthreads();
    skew_begin(skew);
    // ---- gate: g = sigmoid(Wg [agg, xn] + bg)
    f32x16 va[4], acc[4];
    load_tile2(va, agg, row0, n_rows, xpose, lane);
    make_operand(op, va);
    PSTAMP(0) sync_for_stage(); PSTAMP(1)
    gemm_add<WAVES, MODE>(acc, op, buf0, msc + 0, lane, BiasTerm{vec + kVBg, h}, mats + kMatFloats, buf1, wave);
    skew_after_gemm(skew); PSTAMP(2)
    // Blocks of up to four waves (one wave per SIMD, 512 registers): x_res is read ONCE and x1 is never parked in `out` —
    // a row the lane reads back is a scattered load (32 lines of 32 B per instruction) with its latency exposed (−5 % at
    // 96 000 rows).  Eight-wave blocks (256 registers) keep the read-back: held in registers there, the two row vectors
    // become 250 spilled registers and cost more than the loads they save (+3-6 %).
    constexpr bool kKeepRows = WAVES <= 4;
    f32x16 xr[4];
    {
        f32x16 vx[4];
        if (xn != nullptr) {
            load_tile2(vx, xn, row0, n_rows, xpose, lane);
        } else {   // xn = attn_prenorm_x_dst(x): recomputed (same arithmetic as qcn_ln_linear_f32) instead of a round trip
            load_tile2(vx, xres, row0, n_rows, xpose, lane);
            if constexpr (kKeepRows) {
#pragma unroll
                for (int t = 0; t < 4; ++t) xr[t] = vx[t];
            }
            layer_norm<true, false>(vx, vec + kVPreG, vec + kVPreB, h, eps);
        }
        if constexpr (kStatic) make_operand_static(op, vx, msc[24], msc[25]);
        else make_operand(op, vx);
        PSTAMP(0) sync_for_stage(); PSTAMP(1)
        gemm<WAVES>(acc, op, buf1, msc + 2, lane, mats + 2 * kMatFloats, buf0, wave);
        skew_after_gemm(skew); PSTAMP(2)
    }
    // ---- u = agg + g * (s - agg)
    {
        f32x16 vs[4];
        if constexpr (CHAIN) {   // s = Ws LN(x_dst) + bs: the gate's operand again (reference :108)
            PSTAMP(0) sync_for_stage(); PSTAMP(1)
            gemm_add<WAVES, MODE>(vs, op, buf0, msc + 4, lane, BiasTerm{vec + kVBs, h}, mats + 3 * kMatFloats, buf1, wave);
            PSTAMP(2)
        } else {
            load_tile2(vs, s, row0, n_rows, xpose, lane);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // sigmoid on the hardware exp2 / rcp (v_exp_f32, v_rcp_f32: 1 ulp each; the gate is a weight in [0, 1], its
                // absolute error of ~1e-7 is below the rounding of the update it feeds): four instructions instead of
                // expf + IEEE division.  exp2 may overflow to +inf: 1 + inf = inf, rcp(inf) = 0, the gate's limit.
                const float den = 1.0f + __builtin_amdgcn_exp2f(acc[t][r] * -1.4426950408889634f);
                const float g = __builtin_amdgcn_rcpf(den);
                va[t][r] = fmaf(g, vs[t][r] - va[t][r], va[t][r]);
            }
    }
    // ---- x1 = x + LN_post(Wout u + bout)
    make_operand(op, va);
    PSTAMP(0) sync_for_stage(); PSTAMP(1)
    gemm_add<WAVES, MODE>(acc, op, bB, msc + 2 * (2 + kM), lane, BiasTerm{vec + kVBout, h}, mats + (size_t)(3 + kM) * kMatFloats, bA, wave);
    skew_after_gemm(skew); PSTAMP(2)
    layer_norm<true, false>(acc, vec + kVLpG, vec + kVLpB, h, eps);
    if (!kKeepRows || xn != nullptr) load_tile2(xr, xres, row0, n_rows, xpose, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) xr[t] += acc[t];           // x1
    if constexpr (!kKeepRows) store_tile2(xr, out, row0, n_rows, xpose, lane);   // parked in `out` until the last residual
    // ---- feed-forward block, hidden layer in four 128-wide blocks
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = xr[t];
    layer_norm<true, false>(acc, vec + kVLfG, vec + kVLfB, h, eps);
    if constexpr (kStatic) make_operand_static(op, acc, msc[26], msc[27]);
    else make_operand(op, acc);
    f32x16 acc_o[4];
    if constexpr (kStatic) fill_zero(acc_o);                  // raw sum of the four second-layer GEMMs; bias and scale at the end
    else load_feature_vec(acc_o, vec + kVB2, h);
    constexpr bool kPipelined = MODE == kModeSplit && WAVES <= 4;
    if constexpr (kPipelined) {
        auto W1 = [&](int hb) { return mats + (size_t)(3 + kM + 2 * hb) * kMatFloats; };
        auto W2 = [&](int hb) { return mats + (size_t)(4 + kM + 2 * hb) * kMatFloats; };
        auto U1 = [&](int hb) { return op.usc * msc[2 * (3 + kM + 2 * hb) + 1]; };
        const float *after = CHAIN && next.n > 0 ? next.wt : nullptr;
        f32x16 a0[4], a1[4];
        Operand<MODE> hid;
        HidSteps S;
        // first-layer GEMM of block hb into `an` (from buffer `wb`), the epilogue of block hb - 1 (accumulators `ap`) between
        // its MFMAs; stages `nx` into the other buffer
        auto g1_with_e = [&](f32x16 (&an)[4], f32x16 (&ap)[4], int hb, f32x4 *wb, const float *nx, f32x4 *nb) {
            const float *bias = vec + kVB1 + (hb - 1) * kC;
            const float U = U1(hb - 1);
            hid_load(S, bias, h, 0);
            PSTAMP(0) sync_for_stage(); PSTAMP(1)
            gemm_steps<WAVES, true>(an, op, wb, lane, nx, nb, wave, S, [&](int g) { hid_load(S, bias, h, g); },
                              [&](int g, const f32x4 c0, const f32x4 c1) { hid_step<kStatic>(ap, U, hid, S, g, c0, c1); });
            PSTAMP(2)
        };
        auto g2 = [&](int hb, f32x4 *wb, const float *nx, f32x4 *nb) {
            PSTAMP(0) sync_for_stage(); PSTAMP(1)
            if constexpr (kStatic) gemm_raw<WAVES, MODE>(acc_o, hid, wb, lane, nx, nb, wave);
            else gemm<WAVES>(acc_o, hid, wb, msc + 2 * (4 + kM + 2 * hb), lane, nx, nb, wave);
            PSTAMP(2)
        };
        PSTAMP(0) sync_for_stage(); PSTAMP(1)
        gemm_raw<WAVES, MODE, true>(a0, op, bA, lane, W1(1), bB, wave);               // W1_0
        PSTAMP(2)
        g1_with_e(a1, a0, 1, bB, W2(0), bA);                                          // W1_1 + E_0
        g2(0, bA, W1(2), bB);                                                         // W2_0
        g1_with_e(a0, a1, 2, bB, W2(1), bA);                                          // W1_2 + E_1
        g2(1, bA, W1(3), bB);                                                         // W2_1
        g1_with_e(a1, a0, 3, bB, W2(2), bA);                                          // W1_3 + E_2
        g2(2, bA, W2(3), bB);                                                         // W2_2
        {                                                                             // E_3 on its own
            const float *bias = vec + kVB1 + 3 * kC;
            const float U = U1(3);
            hid_load(S, bias, h, 0);
#pragma unroll
            for (int g = 0; g < 26; ++g) {
                const f32x4 c0 = S.b0, c1 = S.b1;
                hid_load(S, bias, h, g + 1);
                hid_step<kStatic>(a1, U, hid, S, g, c0, c1);
            }
        }
        g2(3, bB, after, bA);                                                         // W2_3
        if constexpr (kStatic) scale_add(acc_o, msc[28], BiasTerm{vec + kVB2, h});
    } else {
    const float *after = CHAIN && next.n > 0 ? next.wt : nullptr;
#pragma unroll 1
    for (int hb = 0; hb < 4; ++hb) {
        const float *w1 = mats + (size_t)(3 + kM + 2 * hb) * kMatFloats;
        PSTAMP(0) sync_for_stage(); PSTAMP(1)
        gemm_add<WAVES, MODE>(acc, op, bA, msc + 2 * (3 + kM + 2 * hb), lane, BiasTerm{vec + kVB1 + hb * kC, h}, w1 + kMatFloats,
                              bB, wave);
        skew_after_gemm(skew); PSTAMP(2)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
        Operand<MODE> hid;
        make_operand(hid, acc);
        PSTAMP(0) sync_for_stage(); PSTAMP(1)
        gemm<WAVES>(acc_o, hid, bB, msc + 2 * (4 + kM + 2 * hb), lane, hb < 3 ? w1 + 2 * kMatFloats : after, bA, wave);
        skew_after_gemm(skew); PSTAMP(2)
    }
    }
    layer_norm<true, false>(acc_o, vec + kVLqG, vec + kVLqB, h, eps);
    if constexpr (!kKeepRows) load_tile2(xr, out, row0, n_rows, xpose, lane);   // the wave's own earlier store
#pragma unroll
    for (int t = 0; t < 4; ++t) acc_o[t] += xr[t];
    store_tile2(acc_o, out, row0, n_rows, xpose, lane);
    if constexpr (CHAIN) {
        // ---- the next layer's LayerNorm + projections of this row (its attn_prenorm_x_dst + to_q, or x_src + to_q / to_k / to_v
        // for a layer over one node set): qcn_ln_linear_f32's arithmetic on the row it would have read back
        if (next.n > 0) {
            layer_norm<true, false>(acc_o, nvec, nvec + kC, h, eps);
            if (kStaticNext && next.sc > 0.f) make_operand_static(op, acc_o, next.sc, 1.0f / next.sc);
            else make_operand(op, acc_o);
            const float *nsc = next.wt + (size_t)next.n * kMatFloats;
#pragma unroll 1
            for (int j = 0; j < next.n; ++j) {
                sync_for_stage();
                gemm_add<WAVES, MODE>(acc, op, (j & 1) ? bB : bA, nsc + 2 * j, lane, BiasTerm{nvec + (2 + j) * kC, h},
                                      j + 1 < next.n ? next.wt + (size_t)(j + 1) * kMatFloats : nullptr, (j & 1) ? bA : bB, wave);
                store_tile2(acc, next.y[j], row0, n_rows, xpose, lane);
            }
        }
    }
    skew_end(skew);
#ifdef QCN_CHAIN_STAMPS
    PSTAMP(0)
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4))
        for (int c = 0; c < 4; ++c) g_post_stamps[wave >> 2][c] = acc_t[c];
#endif
}

// A5 — MLPLayer heads (layers/mlp_layer.py:20-36: Linear(128,128) -> LayerNorm -> ReLU -> Linear(128,out)) of the decoder
// (modules/qcnet_decoder.py:117-138,229-233,270-280): up to three heads that read the same row in one launch.
// pack: 2 chain-packed matrices per head [W1_h, W2_h (out rows zero-padded to 128)], then per head the plain vectors
// [b1, ln_g, ln_b, b2 (zero-padded)] (4 * 128), then the scale table (32 floats).  Outputs are dense [n_rows, out_h].
constexpr int kMlpMaxHeads = 3, kMlpVecs = 4 * kC, kMlpScales = 32;

struct MlpHeads {
    float *y[kMlpMaxHeads];
    int out_dim[kMlpMaxHeads];
    int n_heads;
};

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void mlp_heads_kernel(const float *__restrict__ x, int64_t n_rows,
                                                               const float *__restrict__ pack, MlpHeads hd, int skew_mode)
{
    __shared__ __attribute__((aligned(16))) f32x4 wlds[2 * kMatVec4 + (kMlpMaxHeads * kMlpVecs + kMlpScales) / 4];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row = (int64_t)blockIdx.x * (WAVES * kRowsPerWave) + wave * kRowsPerWave + (lane & 31);
    const bool valid = row < n_rows;
    const int64_t rowc = valid ? row : n_rows - 1;
    const int n_mats = 2 * hd.n_heads;
    float *const svec = reinterpret_cast<float *>(wlds + 2 * kMatVec4);
    const float *vecs = svec;
    const float *msc = vecs + hd.n_heads * kMlpVecs;

    const Skew skew = make_skew<WAVES>(skew_mode, wave);
    stage_matrix<WAVES, MODE>(pack, wlds, wave, lane);
    stage_vectors<WAVES>(pack + (size_t)n_mats * kMatFloats, svec, hd.n_heads * kMlpVecs + kMlpScales, tid);
    __syncthreads();
    skew_begin(skew);
    f32x16 xv[4];
    load_row(xv, x, rowc, h);
    Operand<MODE> xo;
    make_operand(xo, xv);
#pragma unroll 1
    for (int i = 0; i < hd.n_heads; ++i) {
        const float *vec = vecs + i * kMlpVecs;
        f32x16 acc[4];
        sync_for_stage();
        gemm_add<WAVES, MODE>(acc, xo, wlds + ((2 * i) & 1) * kMatVec4, msc + 4 * i, lane, BiasTerm{vec, h},
                              pack + (size_t)(2 * i + 1) * kMatFloats, wlds + ((2 * i + 1) & 1) * kMatVec4, wave);
        skew_after_gemm(skew);
        layer_norm<true, true>(acc, vec + kC, vec + 2 * kC, h, 1e-5f);
        Operand<MODE> hid;
        make_operand(hid, acc);
        sync_for_stage();
        gemm_add<WAVES, MODE>(acc, hid, wlds + ((2 * i + 1) & 1) * kMatVec4, msc + 4 * i + 2, lane, BiasTerm{vec + 3 * kC, h},
                              2 * i + 2 < n_mats ? pack + (size_t)(2 * i + 2) * kMatFloats : nullptr,
                              wlds + ((2 * i + 2) & 1) * kMatVec4, wave);
        skew_after_gemm(skew);
        if (valid) {
            const int od = hd.out_dim[i];
            float *y = hd.y[i] + row * od;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 32 * t + 8 * q + 4 * h;
                    if ((od & 3) == 0) {
                        if (f0 < od) {
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = acc[t][4 * q + j];
                            *reinterpret_cast<f32x4 *>(y + f0) = v;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (f0 + j < od) y[f0 + j] = acc[t][4 * q + j];
                    }
                }
        }
    }
    skew_end(skew);
}

}  // namespace

extern "C" size_t qcn_mlp_heads_pack_floats(int n_heads)
{
    return n_heads < 1 || n_heads > kMlpMaxHeads ? 0 : (size_t)2 * n_heads * kMatFloats + (size_t)n_heads * kMlpVecs + kMlpScales;
}

extern "C" int qcn_mlp_heads_f32(const float *x, int64_t n_rows, const float *pack, int n_heads, const int *out_dims,
                                 float *y0, float *y1, float *y2, int matmul_mode, void *stream)
{
    if (n_rows < 0 || n_heads < 1 || n_heads > kMlpMaxHeads || !out_dims || !valid_mode(matmul_mode)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows == 0) return QCN_OK;
    if (!x || !pack || !y0 || (n_heads > 1 && !y1) || (n_heads > 2 && !y2)) return QCN_ERR_INVALID_ARGUMENT;
    MlpHeads hd{};
    float *ys[kMlpMaxHeads] = {y0, y1, y2};
    for (int i = 0; i < n_heads; ++i) {
        if (out_dims[i] < 1 || out_dims[i] > kC) return QCN_ERR_UNSUPPORTED;
        hd.y[i] = ys[i];
        hd.out_dim[i] = out_dims[i];
    }
    hd.n_heads = n_heads;
    hipStream_t s = qcn_stream(stream);
    const int waves = pick_waves(n_rows);
    const unsigned grid = (unsigned)((n_rows + 32 * waves - 1) / (32 * waves));
    const int skew_mode = chain_skew_mode(0);
#define QCN_LAUNCH(S, W) hipLaunchKernelGGL((mlp_heads_kernel<S, W>), dim3(grid), dim3(64 * W), 0, s, x, n_rows, pack, hd, skew_mode)
    QCN_CHAIN_DISPATCH(matmul_mode, waves);
#undef QCN_LAUNCH
    const bool ok = hipGetLastError() == hipSuccess;
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}

extern "C" size_t qcn_attn_post_pack_floats(void) { return (size_t)kPostMats * kMatFloats + kPostVecs + kPostScales; }

extern "C" size_t qcn_attn_post_chain_pack_floats(void) { return (size_t)kChainMats * kMatFloats + kChainVecs + kPostScales; }

extern "C" int qcn_attn_post_chain_f32(const float *agg, const float *x_res, const float *pack, int64_t n_rows, float *out,
                                       const float *next_wt, const float *next_bias, const float *next_ln_weight,
                                       const float *next_ln_bias, float next_operand_scale, int n_next, float *y0, float *y1,
                                       float *y2, int matmul_mode, void *stream)
{
    if (n_rows < 0 || n_next < 0 || n_next > 3 || !valid_mode(matmul_mode)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows == 0) return QCN_OK;
    if (!agg || !x_res || !pack || !out || out == agg || out == x_res) return QCN_ERR_INVALID_ARGUMENT;
    if (n_next > 0 && (!next_wt || !next_ln_weight || !next_ln_bias || !y0 || (n_next > 1 && !y1) || (n_next > 2 && !y2)))
        return QCN_ERR_INVALID_ARGUMENT;
    for (float *y : {y0, y1, y2})
        if (y != nullptr && n_next > 0 && (y == out || y == agg || y == x_res)) return QCN_ERR_INVALID_ARGUMENT;
    hipStream_t s = qcn_stream(stream);
    // one wave per SIMD: the s projection stays in registers beside the gate's accumulators (eight-wave blocks have 256)
    int waves = pick_waves(n_rows, 0.9);
    if (waves > 4) waves = 4;
    const unsigned grid = (unsigned)((n_rows + 32 * waves - 1) / (32 * waves));
    const PostNext next{next_wt, next_bias, next_ln_weight, next_ln_bias, {y0, y1, y2}, n_next, next_operand_scale};
#define QCN_LAUNCH_C(M)                                                                                                     \
    do {                                                                                                                    \
        if (waves == 4)                                                                                                     \
            hipLaunchKernelGGL((attn_post_kernel<M, 4, true>), dim3(grid), dim3(256), 0, s, agg, nullptr, nullptr, x_res, pack, \
                               n_rows, out, 0, next);                                                                       \
        else                                                                                                                \
            hipLaunchKernelGGL((attn_post_kernel<M, 2, true>), dim3(grid), dim3(128), 0, s, agg, nullptr, nullptr, x_res, pack, \
                               n_rows, out, 0, next);                                                                       \
    } while (0)
    if (matmul_mode == QCN_MATMUL_SPLIT_F16) QCN_LAUNCH_C(QCN_MATMUL_SPLIT_F16);
    else if (matmul_mode == QCN_MATMUL_BF16) QCN_LAUNCH_C(QCN_MATMUL_BF16);
    else QCN_LAUNCH_C(QCN_MATMUL_EXACT_F32);
#undef QCN_LAUNCH_C
    return hipGetLastError() == hipSuccess ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}

extern "C" int qcn_ln_linear_f32(const float *x, int64_t n_rows, const float *ln_weight, const float *ln_bias,
                                 const float *wt, const float *bias, int n_out, float *y0, float *y1, float *y2,
                                 float *y3, float *xn_out, int matmul_mode, void *stream)
{
    if (n_rows < 0 || n_out < 1 || n_out > 4 || !valid_mode(matmul_mode)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows == 0) return QCN_OK;
    if (!x || !wt || !y0 || (n_out > 1 && !y1) || (n_out > 2 && !y2) || (n_out > 3 && !y3)) return QCN_ERR_INVALID_ARGUMENT;
    if ((ln_weight == nullptr) != (ln_bias == nullptr)) return QCN_ERR_INVALID_ARGUMENT;
    hipStream_t s = qcn_stream(stream);
    const int waves = pick_waves(n_rows, 1.32);
    const unsigned grid = (unsigned)((n_rows + 32 * waves - 1) / (32 * waves));
#define QCN_LAUNCH(S, W)                                                                                           \
    hipLaunchKernelGGL((ln_linear_kernel<S, W>), dim3(grid), dim3(64 * W), 0, s, x, n_rows, ln_weight, ln_bias, wt, bias, \
                       n_out, y0, y1, y2, y3, xn_out)
    QCN_CHAIN_DISPATCH(matmul_mode, waves);
#undef QCN_LAUNCH
    const bool ok = hipGetLastError() == hipSuccess;
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}

extern "C" int qcn_attn_post_f32(const float *agg, const float *xn_dst, const float *s_proj, const float *x_res,
                                 const float *pack, int64_t n_rows, float *out, int matmul_mode, void *stream)
{
    if (n_rows < 0 || !valid_mode(matmul_mode)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows == 0) return QCN_OK;
    if (!agg || !s_proj || !x_res || !pack || !out) return QCN_ERR_INVALID_ARGUMENT;   // xn_dst may be NULL
    if (out == agg || out == xn_dst || out == s_proj || out == x_res) return QCN_ERR_INVALID_ARGUMENT;
    hipStream_t s = qcn_stream(stream);
    // (split mode: the four-wave blocks run the feed-forward block software-pipelined — 0.90 of the eight-wave blocks' time
    // per row at 960 k rows, same box)
    const int waves = pick_waves(n_rows, matmul_mode == QCN_MATMUL_SPLIT_F16 ? 0.9 : 1.0);
    const unsigned grid = (unsigned)((n_rows + 32 * waves - 1) / (32 * waves));
    const int skew_mode = chain_skew_mode(0);
#define QCN_LAUNCH(S, W)                                                                                          \
    hipLaunchKernelGGL((attn_post_kernel<S, W>), dim3(grid), dim3(64 * W), 0, s, agg, xn_dst, s_proj, x_res, pack, n_rows, \
                       out, skew_mode, PostNext{})
    QCN_CHAIN_DISPATCH(matmul_mode, waves);
#undef QCN_LAUNCH
    const bool ok = hipGetLastError() == hipSuccess;
#ifdef QCN_CHAIN_STAMPS
    {
        unsigned long long h[2][4];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_post_stamps), sizeof(h));
        for (int w = 0; w < 2; ++w)
            printf("attn_post stamps wave %d (waves %d, rows %lld): epilogue+loads %llu, barrier %llu, gemm %llu cycles\n", 4 * w,
                   waves, (long long)n_rows, h[w][0], h[w][1], h[w][2]);
    }
#endif
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}
