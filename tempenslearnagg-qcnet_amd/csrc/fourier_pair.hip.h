// FourierEmbedding (fourier.hip) in the two-tiles-per-wave form of chain_pair.hip.h: blocks of four waves x 64 rows, the
// matrix pipe runs the GEMM of one tile while the vector ALU runs the other tile's features / LayerNorm / operand split.
// Included by fourier.hip (same packed parameters); split and bf16 modes, no categorical embeddings, with the embedding
// stored (NORM = normalize_out) or with the folded tail of qcn_fourier_folded_linear_f32 (TAIL).
#pragma once
#include "chain_pair.hip.h"

namespace {

using namespace qcn_chain;

#define AI __attribute__((always_inline))

template <int MODE> struct PairTile {
    f32x16 y[4];        // running sum over input dims (operands' scale until the last dim is in)
    f32x16 hh[4];       // hidden vector of the current MLP / output of the current Linear
    Operand<MODE> op;   // operand of the tile's next GEMM
    int64_t row0;
    bool cold;
};

// State of the epilogue in flight, shared by the wave's two tiles (their epilogues never overlap).
struct PairScratch {
    RowStats st;
    f32x4 c4, s4;       // sines / cosines of the frequency group being computed
    f32x4 cp, sp;       // ... of the previous group, converted to operand fragments one pair per step
    float red, red2;
};

// The rows' inputs live OUTSIDE the tile structs, as scalars of their own: selecting among struct members by the (runtime)
// input dim becomes a dynamically indexed access, which pins the whole tile in scratch memory.
struct RowIn {
    float v0, v1, v2, v3;
};

template <int IN> __device__ __forceinline__ float pick_x(const RowIn &r, int d)
{
    float v = r.v0;
    if (IN > 1) v = d == 1 ? r.v1 : v;
    if (IN > 2) v = d == 2 ? r.v2 : v;
    if (IN > 3) v = d == 3 ? r.v3 : v;
    return v;
}

#ifdef QCN_PAIR_STAMPS
// diagnostic build only (scripts/dbg/build_variant.sh stamps -DQCN_PAIR_STAMPS): s_memtime of block 0's waves at every
// phase boundary, read back and printed by the entry point
__device__ unsigned long long g_pair_stamps[kPairWaves][64];
#define PST() { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && lane == 0 && pst_i < 64) g_pair_stamps[wave][pst_i] = __builtin_amdgcn_s_memtime(); ++pst_i; __builtin_amdgcn_sched_barrier(0); }
#else
#define PST()
#endif

#ifdef QCN_SLOT_STAMPS   // -DQCN_SLOT_STAMPS=1: the e1 GEMM of input dim 1, =2: its feature GEMM
#define QCN_SLOT_STAMP_SEL(COND, WHICH) ((COND) && QCN_SLOT_STAMPS == (WHICH))
#else
#define QCN_SLOT_STAMP_SEL(COND, WHICH) 0
#endif

constexpr int kTailVecs = 3 * kC + 8;   // tail biases + {scale, 1 / scale} pairs, staged behind the pack's vectors

template <int IN, int MODE, bool TAIL, int NORM>
__global__ __launch_bounds__(64 * kPairWaves) void fourier_pair_kernel(const float *__restrict__ x, const float *__restrict__ pack,
                                                                       float *__restrict__ y, int64_t n_rows, FourierTail tail)
{
    static_assert(MODE != kModeExact && IN >= 1 && (!TAIL || NORM == 0), "unsupported instantiation");
    constexpr bool kCentred = MODE != kModeBf16;   // Linear layers in front of a LayerNorm centred on the host (ops.pack_fourier)
    constexpr int kVecBlock = IN * kDimVecs + kOutVecs + kScales;
    __shared__ __attribute__((aligned(16))) f32x4 wlds[2 * kMatVec4 + (kVecBlock + kTailVecs) / 4 + kPairWaves * 128];
    f32x4 *const buf0 = wlds, *const buf1 = wlds + kMatVec4;
    float *const svec = reinterpret_cast<float *>(wlds + 2 * kMatVec4);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x4 *const xpose = wlds + 2 * kMatVec4 + (kVecBlock + kTailVecs) / 4 + wave * 128;
    const float eps = 1e-5f;
    const float *mats = pack;
    const float *vecs = svec;
    const float *out_vec = vecs + IN * kDimVecs;
    const float *msc = out_vec + kOutVecs;
    const float *tvec = msc + kScales;             // tail: biases [3][128], then scale pairs

    PairTile<MODE> A, B;
    RowIn inA, inB;
    const int64_t wave_row0 = (int64_t)blockIdx.x * kPairBlockRows + wave * kPairRows;
    auto init_tile = [&](PairTile<MODE> &T, RowIn &in, int64_t row0) AI {
        T.row0 = row0;
        const int64_t row = row0 + n;
        const int64_t rowc = row < n_rows ? row : n_rows - 1;
        in.v0 = x[rowc * IN];
        in.v1 = IN > 1 ? x[rowc * IN + 1] : 0.f;
        in.v2 = IN > 2 ? x[rowc * IN + 2] : 0.f;
        in.v3 = IN > 3 ? x[rowc * IN + 3] : 0.f;
        T.cold = false;
        fill_zero(T.y);
    };
    init_tile(A, inA, wave_row0);
    init_tile(B, inB, wave_row0 + kRowsPerWave);

    stage_matrix<kPairWaves, MODE>(mats, buf0, wave, lane);
    stage_vectors<kPairWaves>(pack + (size_t)(2 * IN + 1) * kMatFloats, svec, kVecBlock, tid);
    if (TAIL) {
        float *tv = svec + kVecBlock;
        for (int i = tid; i < tail.n_out * kC; i += 64 * kPairWaves) tv[i] = tail.bias != nullptr ? tail.bias[i] : 0.f;
        if (tid < 2 * tail.n_out) tv[3 * kC + tid] = tail.wt[(size_t)tail.n_out * kMatFloats + tid];
    }
    __syncthreads();

    // uniform scalars of the scale table (scalar registers)
    auto uni = [&](float v) AI { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float u2 = uni(msc[kTblU2]), uo = uni(msc[2 * IN]), uso = uni(msc[kTblSo + 1]);
    Staged P;        // LDS parameters of the epilogue step in flight, read one slot ahead (chain_pair.hip.h)
    const lds_cfptr lbase = opaque_lds_ptr(vecs + 4 * h);            // per-lane base of the vector block (lds_vec4_at)
    constexpr int kOutOff = IN * kDimVecs, kTailOff = kOutOff + kOutVecs + kScales;
    PairScratch S;

    // ---- features of input dim d: one frequency per step; a finished group of four (cos -> group (m >> 2, m & 3) of the
    // operand, sin two feature blocks on) is converted one pair per step while the next group is computed, so every step
    // carries two independent dependency chains (one wave per SIMD hides nothing else)
    auto feat_load = [&](int d, int g) AI {
        const float *dv = vecs + d * kDimVecs;
        if (g < kSlots && (g & 3) == 0) P.a = *reinterpret_cast<const f32x4 *>(dv + kVFreq + 32 * h + g);
        if (g == 0) P.b[0] = dv[kVFmx + h];
    };
    auto feat_convert = [&](PairTile<MODE> &T, int m, int k) AI {   // pair k = 0..3 of finished group m
        const f32x4 &v = k < 2 ? S.cp : S.sp;
        put2(T.op, (k < 2 ? 0 : 2) + (m >> 2), m & 3, k & 1, v[2 * (k & 1)], v[2 * (k & 1) + 1]);
    };
    auto feat_step = [&](PairTile<MODE> &T, float xv, int g, const Staged &cur) AI {
        if (g == 0) T.cold = !(fabsf(xv) * cur.b[0] <= 5215.0f);   // see fourier_kernel: validity of the fast reduction
        float sn, cs;
        fourier_sincos(xv, cur.a[g & 3], &sn, &cs);
        pin(cs);
        pin(sn);
        S.c4[g & 3] = cs;
        S.s4[g & 3] = sn;
        if (g >= 4) feat_convert(T, (g >> 2) - 1, g & 3);
        if ((g & 3) == 3) { S.cp = S.c4; S.sp = S.s4; }
    };
    auto feat_finish = [&](PairTile<MODE> &T) AI {   // the last group, behind the GEMM
#pragma unroll
        for (int k = 0; k < 4; ++k) feat_convert(T, 7, k);
    };
    // lanes outside the fast reduction's range redo their dim with ocml's sincosf (cold; every lane keeps ITS path's values,
    // so a row's result does not depend on its neighbours)
    auto feat_cold = [&](PairTile<MODE> &T, int d, float xv) AI {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(T.cold) == 0, 1)) return;
        const float *dv = vecs + d * kDimVecs;
#pragma unroll 1
        for (int m = 0; m < 8; ++m) {
            f32x4 c, s;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f = dv[kVFreq + 32 * h + 4 * m + k];
                float sn, cs, sn2, cs2, p;
                fourier_sincos(xv, f, &sn, &cs);
                sincosf(fourier_angle(xv, f, &p), &sn2, &cs2);
                c[k] = T.cold ? cs2 : cs;
                s[k] = T.cold ? sn2 : sn;
            }
            switch (m) {
#define QCN_CASE(M) case M: put4(T.op, (M) >> 2, (M) & 3, c); put4(T.op, 2 + ((M) >> 2), (M) & 3, s); break;
                QCN_CASE(0) QCN_CASE(1) QCN_CASE(2) QCN_CASE(3) QCN_CASE(4) QCN_CASE(5) QCN_CASE(6) default: QCN_CASE(7)
#undef QCN_CASE
            }
        }
    };

    // ---- LayerNorm + ReLU -> operand behind a Linear layer whose accumulators are in v: v = u v + a first (steps 0..7, two
    // groups per step; the additive terms come from `terms`), then the statistics (centred rows: the variance only) and one
    // group per step normalised, scaled into the consumer's static operand scale and converted
    constexpr int kLnNorm = kCentred ? 10 : 16;      // first normalise step
    auto ln_load = [&](lds_cfptr base, int gamma_off, int beta_off, int g) AI {
        if (g >= kLnNorm && g < kLnNorm + 16) {
            P.a = lds_vec4_at(base, gamma_off, (g - kLnNorm) >> 2, (g - kLnNorm) & 3);
            P.b = lds_vec4_at(base, beta_off, (g - kLnNorm) >> 2, (g - kLnNorm) & 3);
        }
    };
    auto ln_steps = [&](f32x16 (&v)[4], PairTile<MODE> &T, float u, const f32x4 a0, const f32x4 a1, int g, const Staged &cur) AI {
        if (g < 8) {
            ln_step_scale_add<kCentred>(v, S.st, g, u, a0, a1);
        } else if (g == 8) {
            S.red = ln_reduce_begin(S.st);
            S.red2 = xhalf(S.red);
        } else if (kCentred) {
            if (g == 9) S.st.rstd = rsqrtf((S.red + S.red2) * (1.0f / kC) + eps);
            else if (g < 26) ln_step_normalize_put<MODE, true>(v, S.st, T.op, g - 10, cur.a, cur.b);
        } else {
            if (g == 9) S.st.mean = (S.red + S.red2) * (1.0f / kC);
            else if (g < 14) ln_step_centre(v, S.st, g - 10);
            else if (g == 14) { S.red = ln_reduce_begin(S.st); S.red2 = xhalf(S.red); }
            else if (g == 15) S.st.rstd = rsqrtf((S.red + S.red2) * (1.0f / kC) + eps);
            else ln_step_normalize_put<MODE, true>(v, S.st, T.op, g - 16, cur.a, cur.b);
        }
    };
    // behind a first layer: h = U acc + x w_last + b1
    auto e1_load = [&](lds_cfptr dbase, int g) AI {       // dbase = lbase + d * kDimVecs
        if (g < 8) {
            P.a = lds_vec4_at(dbase, kVB1, (2 * g) >> 2, (2 * g) & 3);
            P.b = lds_vec4_at(dbase, kVW1Last, (2 * g) >> 2, (2 * g) & 3);
            P.c = lds_vec4_at(dbase, kVB1, (2 * g + 1) >> 2, (2 * g + 1) & 3);
            P.d = lds_vec4_at(dbase, kVW1Last, (2 * g + 1) >> 2, (2 * g + 1) & 3);
        }
        ln_load(dbase, kVG1, kVBe1, g);
    };
    auto e1_step = [&](PairTile<MODE> &T, float xv, float u1, int g, const Staged &cur) AI {
        f32x4 a0, a1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a0[k] = fmaf(xv, cur.b[k], cur.a[k]); a1[k] = fmaf(xv, cur.d[k], cur.c[k]); }
        ln_steps(T.hh, T, u1, a0, a1, g, cur);
    };
    // behind the last second layer: y = U2 acc + sum b2, to_out's LayerNorm
    auto eo_load = [&](int g) AI {
        if (g < 8) {
            P.a = lds_vec4_at(lbase, kOutOff + kVB2, (2 * g) >> 2, (2 * g) & 3);
            P.b = lds_vec4_at(lbase, kOutOff + kVB2, (2 * g + 1) >> 2, (2 * g + 1) & 3);
        }
        ln_load(lbase, kOutOff + kVOutG, kOutOff + kVOutBe, g);
    };
    auto eo_step = [&](PairTile<MODE> &T, int g, const Staged &cur) AI { ln_steps(T.y, T, u2, cur.a, cur.b, g, cur); };

    // ---- fp32 rows to global memory over steps g0 .. g0 + 8: pass k is transposed in step g0 + k, stored in step g0 + k + 1
    auto store_steps = [&](const f32x16 (&v)[4], float *dst, int64_t row0, int g, int g0, const Staged &cur) AI {
        if (g > g0 && g <= g0 + 8) store_pass(dst, row0, n_rows, lane, g - g0 - 1, cur.c, cur.d);
        if (g >= g0 && g < g0 + 8) xpose_pass(v, xpose, lane, g - g0, P.c, P.d);
    };

    // ---- behind to_out's Linear: bias, optional row normalisation, store
    auto ef_load = [&](int g) AI {
        if (g < 8) {
            P.a = lds_vec4_at(lbase, kOutOff + kVOutB, (2 * g) >> 2, (2 * g) & 3);
            P.b = lds_vec4_at(lbase, kOutOff + kVOutB, (2 * g + 1) >> 2, (2 * g + 1) & 3);
        }
    };
    auto ef_step = [&](PairTile<MODE> &T, int g, const Staged &cur) AI {
        if (g < 8) {
            ln_step_scale_add<false>(T.hh, S.st, g, uo, cur.a, cur.b);
        } else if (NORM == 0) {
            store_steps(T.hh, y, T.row0, g, 8, cur);
        } else if (g < 16) {
            if (g == 8) { S.red = ln_reduce_begin(S.st); S.red2 = xhalf(S.red); }
            else if (g == 9) S.st.mean = (S.red + S.red2) * (1.0f / kC);
            else if (g < 14) ln_step_centre(T.hh, S.st, g - 10);
            else if (g == 14) { S.red = ln_reduce_begin(S.st); S.red2 = xhalf(S.red); }
            else S.st.rstd = rsqrtf((S.red + S.red2) * (1.0f / kC) + eps);
        } else if (NORM == 1) {        // steps 16 .. 31: scale a pass' two groups and transpose it, store it one step later
            const int k = (g - 16) >> 1, t = k >> 1, half = k & 1;
            if (!(g & 1)) {
#pragma unroll
                for (int r = 0; r < 8; ++r) { float yv = T.hh[t][8 * half + r] * S.st.rstd; pin(yv); T.hh[t][8 * half + r] = yv; }
                xpose_pass(T.hh, xpose, lane, k, P.c, P.d);
            } else {
                store_pass(y, T.row0, n_rows, lane, k, cur.c, cur.d);
            }
        } else if (NORM == 3) {
            const int p = (g - 16) >> 1;        // passes 0..3 in steps 16..23
            if (g < 24) {
                if (!(g & 1)) T.hh[p] *= S.st.rstd;
                else store_tile2_bf16_pass(T.hh, reinterpret_cast<__bf16 *>(y), T.row0, n_rows, xpose, lane, p);
            }
        } else {
            const int p = (g - 16) >> 1, t = p >> 1, half = p & 1;
            if (!(g & 1)) {
#pragma unroll
                for (int r = 0; r < 8; ++r) { float yv = T.hh[t][8 * half + r] * S.st.rstd; pin(yv); T.hh[t][8 * half + r] = yv; }
            } else {
                store_tile2_split_pass(T.hh, reinterpret_cast<_Float16 *>(y), T.row0, n_rows, xpose, lane, 2048.0f, p);
            }
        }
    };

    // ---- behind tail matrix j: y_j = U_j acc + b_j, store
    auto et_load = [&](int j, int g) AI {
        if (g < 8) {
            P.a = lds_vec4_at(lbase + j * kC, kTailOff, (2 * g) >> 2, (2 * g) & 3);
            P.b = lds_vec4_at(lbase + j * kC, kTailOff, (2 * g + 1) >> 2, (2 * g + 1) & 3);
        }
    };
    auto et_step = [&](PairTile<MODE> &T, int j, float uj, int g, const Staged &cur) AI {
        if (g < 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = 2 * g + i, t = m >> 2, q = m & 3;
                const f32x4 b = i ? cur.b : cur.a;
#pragma unroll
                for (int k = 0; k < 4; ++k) { float yv = fmaf(T.hh[t][4 * q + k], uj, b[k]); pin(yv); T.hh[t][4 * q + k] = yv; }
            }
        } else {
            float *yj = j == 0 ? tail.y[0] : j == 1 ? tail.y[1] : tail.y[2];
            store_steps(T.hh, yj, T.row0, g, 8, cur);
        }
    };
    auto tail_u = [&](int j) AI { return uso * uni(tvec[3 * kC + 2 * j + 1]); };

// a GEMM's callbacks: the staged parameters of step g are read during slot g - 1 (LOAD(0) before the GEMM)
#define QCN_PIPE(LOAD, STEP) P, LOAD, STEP

#ifdef QCN_PAIR_STAMPS
    int pst_i = 0;
#endif
    // ---- the pipeline
    PST()
    {
        const float xa0 = inA.v0;
        feat_load(0, 0);
        run_steps(QCN_PIPE([&](int g) AI { feat_load(0, g); }, [&](int g, const Staged &cur) AI { feat_step(A, xa0, g, cur); }));
        feat_finish(A);
        PST()
        feat_cold(A, 0, xa0);
    }
    PST() sync_for_stage(); PST()
#pragma unroll 1
    for (int d = 0; d < IN; ++d) {
        const float *w2 = mats + (size_t)(2 * d + 1) * kMatFloats;
        const float xa = pick_x<IN>(inA, d), xb = pick_x<IN>(inB, d), xa_next = pick_x<IN>(inA, d + 1);
        const float u1 = uni(msc[2 * d]);
        auto load_feat_b = [&](int g) AI { feat_load(d, g); };
        auto load_feat_a = [&](int g) AI { feat_load(d + 1, g); };
        lds_cfptr dbase = lbase + d * kDimVecs;
        asm volatile("" : "+v"(dbase));
        auto load_e1 = [&](int g) AI { e1_load(dbase, g); };
        // W1_d is in buffer 0; W2_d goes to buffer 1
        load_feat_b(0);
        gemm_bt<true>(A.hh, A.op, buf0, lane, w2, buf1, wave, 0,
                QCN_PIPE(load_feat_b, [&](int g, const Staged &cur) AI { feat_step(B, xb, g, cur); }), QCN_SLOT_STAMP_SEL(d == 1, 2));
        feat_finish(B);
        PST()
        feat_cold(B, d, xb);
        load_e1(0);
        gemm_bt<true>(B.hh, B.op, buf0, lane, w2, buf1, wave, 1,
                QCN_PIPE(load_e1, [&](int g, const Staged &cur) AI { e1_step(A, xa, u1, g, cur); }), QCN_SLOT_STAMP_SEL(d == 1, 1));
        PST()
        PST() sync_for_stage(); PST()
        // W2_d is in buffer 1; the next matrix (first layer of dim d + 1, to_out's Linear, or the first tail matrix) goes to buffer 0
        const float *nxt = TAIL && d == IN - 1 ? tail.wt : mats + (size_t)(2 * d + 2) * kMatFloats;
        load_e1(0);
        gemm_bt(A.y, A.op, buf1, lane, nxt, buf0, wave, 0,
                QCN_PIPE(load_e1, [&](int g, const Staged &cur) AI { e1_step(B, xb, u1, g, cur); }));
        PST()
        if (d + 1 < IN) {
            load_feat_a(0);
            gemm_bt(B.y, B.op, buf1, lane, nxt, buf0, wave, 1,
                    QCN_PIPE(load_feat_a, [&](int g, const Staged &cur) AI { feat_step(A, xa_next, g, cur); }));
            feat_finish(A);
            PST()
            feat_cold(A, d + 1, xa_next);
        } else {
            eo_load(0);
            gemm_bt(B.y, B.op, buf1, lane, nxt, buf0, wave, 1, QCN_PIPE(eo_load, [&](int g, const Staged &cur) AI { eo_step(A, g, cur); }));
            PST()
        }
        PST() sync_for_stage(); PST()
    }
    if (!TAIL) {
        eo_load(0);
        gemm_bt<true>(A.hh, A.op, buf0, lane, pack, buf1, wave, 0, QCN_PIPE(eo_load, [&](int g, const Staged &cur) AI { eo_step(B, g, cur); }));
        PST()
        ef_load(0);
        gemm_bt<true>(B.hh, B.op, buf0, lane, pack, buf1, wave, 1, QCN_PIPE(ef_load, [&](int g, const Staged &cur) AI { ef_step(A, g, cur); }));
        PST()
        ef_load(0);
        run_steps(QCN_PIPE(ef_load, [&](int g, const Staged &cur) AI { ef_step(B, g, cur); }));
        PST()
    } else {
        const float *t1 = tail.n_out > 1 ? tail.wt + kMatFloats : pack;   // (no next matrix: any source, idle buffer)
        const float ut0 = tail_u(0);
        auto load_t0 = [&](int g) AI { et_load(0, g); };
        eo_load(0);
        gemm_bt<true>(A.hh, A.op, buf0, lane, t1, buf1, wave, 0, QCN_PIPE(eo_load, [&](int g, const Staged &cur) AI { eo_step(B, g, cur); }));
        PST()
        load_t0(0);
        gemm_bt<true>(B.hh, B.op, buf0, lane, t1, buf1, wave, 1, QCN_PIPE(load_t0, [&](int g, const Staged &cur) AI { et_step(A, 0, ut0, g, cur); }));
        PST()
#pragma unroll 1
        for (int j = 1; j < tail.n_out; ++j) {
            sync_for_stage();
            const float *tn = j + 1 < tail.n_out ? tail.wt + (size_t)(j + 1) * kMatFloats : pack;
            f32x4 *const bj = wlds + (j & 1) * kMatVec4, *const bn = wlds + ((j + 1) & 1) * kMatVec4;
            const float up = tail_u(j - 1), uj = tail_u(j);
            auto load_prev = [&](int g) AI { et_load(j - 1, g); };
            auto load_cur = [&](int g) AI { et_load(j, g); };
            // (tile B's previous result is still being stored while tile A's GEMM runs: A.hh and B.hh are distinct)
            load_prev(0);
            gemm_bt<true>(A.hh, A.op, bj, lane, tn, bn, wave, 0, QCN_PIPE(load_prev, [&](int g, const Staged &cur) AI { et_step(B, j - 1, up, g, cur); }));
            load_cur(0);
            gemm_bt<true>(B.hh, B.op, bj, lane, tn, bn, wave, 1, QCN_PIPE(load_cur, [&](int g, const Staged &cur) AI { et_step(A, j, uj, g, cur); }));
        }
        const int jl = tail.n_out - 1;
        const float ul = tail_u(jl);
        auto load_last = [&](int g) AI { et_load(jl, g); };
        load_last(0);
        run_steps(QCN_PIPE(load_last, [&](int g, const Staged &cur) AI { et_step(B, jl, ul, g, cur); }));
    }
#undef QCN_PIPE
}

#undef AI

}  // namespace
