// A4 — the whole FourierEmbedding forward (layers/fourier_embedding.py:49-67) as one kernel:
//   x[N,in] -> per input dim i: [cos(2 pi x_i f_ij), sin(2 pi x_i f_ij), x_i] (129) -> Linear(129,128) -> LN -> ReLU ->
//   Linear(128,128); sum over i (+ categorical embeddings) -> LN -> ReLU -> Linear(128,128) [-> row normalise].
// Nothing but x and the [N,128] result touches HBM (the reference materialises [N,in,129] features and five [N,128]
// intermediates per input dim).  The contractions run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// fma chains, so the result stays within fp32 rounding of the reference — there is no reduced-precision step).
//
// Feature-major chain (chain_tile.hip.h): a lane owns one row; the 128 Fourier features of an input dim are computed
// straight into the MFMA operand registers (lane half h evaluates sincos for frequencies 32 h .. 32 h + 31: k-steps
// 0..31 carry the cosines, 32..63 the sines, and W1 is packed on the host in that k order), the hidden vector after
// LN + ReLU is the operand of the second layer, whose accumulator is the running sum over input dims; the 129th
// input column (x itself) is a rank-1 term folded into the accumulator init.  LDS holds only the double-buffered
// packed weights.  MFMA-bound: 2*128*(129+128) flops per row and input dim + 2*128*128 for to_out.
#include "chain_tile.hip.h"
#include <cstdio>

namespace {

using namespace qcn_chain;

// packed parameter layout (floats); host side: ops.pack_fourier (include/qcnet_hip.h, qcn_fourier_embed_f32)
//   matrices (chain-packed, 16384 each): [W1c_0, W2_0, W1c_1, W2_1, ..., Wout]   (W1c: output-centred first layers)
//   then per input dim: freq[64], w1c_last[128], b1c, ln_g * s1, ln_b * s1, {max|freq[0:32]|, max|freq[32:64]|, 0, 0}  (kDimVecs)
//   then to_out: sum_d b2_d, ln_g * so, ln_b * so, b                            (kOutVecs)
//   then the scale table (32 floats): U_m = 1 / (operand scale x matrix scale) for the first layers m = 2 d and
//   to_out m = 2 IN, [24] = so, [25] = 1 / so, [26] = U of the W2_d (one value)
// s1 / so: static power-of-two operand scales of the LayerNorm outputs (chain_tile.hip.h make_operand_static); the W2_d
// are packed so that s1_d x (matrix scale of W2_d) is the same for every d: their products accumulate in ONE scale.
constexpr int kVFreq = 0, kVW1Last = 64, kVB1 = 64 + kC, kVG1 = 64 + 2 * kC, kVBe1 = 64 + 3 * kC, kVFmx = 64 + 4 * kC,
              kDimVecs = 64 + 4 * kC + 4;
constexpr int kVB2 = 0, kVOutG = kC, kVOutBe = 2 * kC, kVOutB = 3 * kC, kOutVecs = 4 * kC, kScales = 32, kTblSo = 24, kTblU2 = 26;

// The reference's angle ((x * f) * 2) * pi, evaluated in fp32 in that order (layers/fourier_embedding.py:58): the doubling is
// exact, so the two last products are ONE rounding of p * (2 pi_f32) with p = fl(x * f).
__device__ __forceinline__ float fourier_angle(float xv, float f, float *p_out)
{
    const float p = __fmul_rn(xv, f);
    *p_out = p;
    return __fmul_rn(p, 6.2831854820251465f);    // 2 * 3.1415927410125732 (pi rounded to fp32), exactly representable
}

// sin and cos of that fp32 angle on the revolution-based hardware v_sin_f32 / v_cos_f32: rev = ang / (2 pi) - k with
// 1 / (2 pi) carried as hi + lo (two fmas; the first one is exact up to its final rounding because ang * hi and k cancel to
// a value below one) and k = rint(p) — p is the angle in revolutions up to 1e-7 relative, so |rev| <= 0.5 + 5e-6 — which
// keeps the rounding of k off the dependency chain p -> angle -> rev -> sin (one wave per SIMD runs this chain unhidden).
// Max abs error of rev against the exact value 3.3e-8, i.e. 2.1e-7 in the sine of the same fp32 angle for |angle| < 2^15
// (the three-term Cody-Waite form in radians it replaces: 3.3e-7 at two instructions more per angle; ocml sincosf: 7e-8,
// scripts/dbg/sincos_acc.hip).  Valid for |ang| <= 2^15 (the caller takes the ocml path for a lane's whole input dim
// otherwise, see fourier_kernel).
__device__ __forceinline__ void fourier_sincos(float xv, float f, float *sn, float *cs)
{
    float p;
    const float ang = fourier_angle(xv, f, &p);
    const float k = rintf(p);
#ifdef QCN_SINCOS_CW     // the three-term Cody-Waite form (A/B builds)
    float r = fmaf(-k, 6.28125f, ang);
    r = fmaf(-k, 1.9350051879882812e-3f, r);
    r = fmaf(-k, 3.0199159795074593e-7f, r);
    const float rev = r * 0.15915494309189535f;
#else
    float rev = fmaf(ang, 0.15915493667125702f, -k);         // 1 / (2 pi) = 0.15915493667125702 + 6.4206382432985265e-09
    rev = fmaf(ang, 6.4206382432985265e-09f, rev);
#endif
    *sn = __builtin_amdgcn_sinf(rev);
    *cs = __builtin_amdgcn_cosf(rev);
}

// Optional linear layers applied to the embedding before it leaves the registers (qcn_fourier_linear_f32): y_j = W_j emb + b_j.
struct FourierTail {
    const float *wt;      // n_out chain-packed matrices + their scale pairs (the qcn_ln_linear_f32 layout)
    const float *bias;    // [n_out, 128] or nullptr
    float *y[3];
    int n_out;            // 0: store the embedding itself
    int fold;             // 1: the caller folded to_out's Linear into wt / bias (y_j = W'_j relu(LN(sum)) + b'_j): one GEMM less
    // GRU-input form (qcn_fourier_gru_input_f32): rows are (sequence, step) pairs, row = seq * gru_steps + step, and a wave's
    // tile is ONE step of 32 consecutive sequences (tile = seq_block * gru_steps + step) instead of 32 consecutive rows; the
    // outputs y[j] are written in the register layout itself — [tile][feature block 4][q 4][lane 64][4 floats], which is
    // exactly what lane (n, h) of wave `feature block` of qcn_gru_blocked_f32 consumes for that step: every store (and every
    // load of the recurrence) is a contiguous 1 KiB per wave, and the three LDS transpositions per tile disappear.
    int gru_steps;        // 0: plain row-major outputs
    int64_t n_seq;
};

template <int IN, int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fourier_kernel(const float *__restrict__ x,
                                                             const float *__restrict__ extra, int extra_div,
                                                             const float *__restrict__ pack, float *__restrict__ y,
                                                             int64_t n_rows, int normalize_out, FourierTail tail, int skew_mode)
{
    // two matrix buffers + the per-feature vectors, in ONE __shared__ array (see attn_post_kernel)
    constexpr int kVecBlock = IN * kDimVecs + kOutVecs + kScales;
    constexpr int kTailBias = 3 * kC;                // biases of the tail's layers, staged behind the pack's vectors
    __shared__ __attribute__((aligned(16))) f32x4 wlds[2 * kMatVec4 + (kVecBlock + kTailBias) / 4 + WAVES * 128];
    f32x4 *const buf0 = wlds, *const buf1 = wlds + kMatVec4;
    float *const svec = reinterpret_cast<float *>(wlds + 2 * kMatVec4);
    f32x4 *const xpose = wlds + 2 * kMatVec4 + (kVecBlock + kTailBias) / 4 + (threadIdx.x >> 6) * 128;   // store_tile2: coalesced row stores
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int64_t row = (int64_t)blockIdx.x * (WAVES * kRowsPerWave) + wave * kRowsPerWave + (lane & 31);
    bool valid = row < n_rows;
    int64_t rowc = valid ? row : n_rows - 1;   // tail lanes recompute the last row; their store is masked
    int64_t gru_tile = -1;                     // GRU-input form: this wave's (sequence block, step) tile; -1: none / beyond the end
    if (tail.gru_steps > 0) {
        const int64_t n_tiles = ((tail.n_seq + kRowsPerWave - 1) / kRowsPerWave) * tail.gru_steps;
        const int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
        const int64_t tc = tile < n_tiles ? tile : n_tiles - 1;
        const int64_t seq = (tc / tail.gru_steps) * kRowsPerWave + (lane & 31);
        rowc = (seq < tail.n_seq ? seq : tail.n_seq - 1) * tail.gru_steps + tc % tail.gru_steps;   // sequences past the end: the last one again
        row = rowc;
        valid = true;
        gru_tile = tile < n_tiles ? tile : -1;
    }
    const float eps = 1e-5f;
    const float *mats = pack;
    const float *vecs = svec;
    const float *out_vec = vecs + IN * kDimVecs;
    const float *msc = out_vec + kOutVecs;
    Operand<MODE> op;

    const Skew skew = make_skew<WAVES>(skew_mode, wave);
    stage_matrix<WAVES, MODE>(mats, buf0, wave, lane);
    stage_vectors<WAVES>(pack + (size_t)(2 * IN + 1) * kMatFloats, svec, kVecBlock, tid);
    if (tail.n_out > 0)
        for (int i = tid; i < tail.n_out * kC; i += 64 * WAVES) svec[kVecBlock + i] = tail.bias != nullptr ? tail.bias[i] : 0.f;
    __syncthreads();
    skew_begin(skew);

    f32x16 acc_y[4];
    fill_zero(acc_y);

#pragma unroll 1
    for (int d = 0; d < IN; ++d) {
        const float *dv = vecs + d * kDimVecs;
        const float xv = x[rowc * IN + d];
        // ---- Fourier features of this lane's 32 frequencies: cos in k-steps 0..31, sin in 32..63
        // One range test per lane and input dim instead of one per angle: |x| max|f| <= 5215 bounds every |angle| of the
        // lane below 2^15 (three roundings included), where the fast reduction is valid.  Lanes beyond it (positions of
        // 1e5 m and more, or a huge learned frequency) redo their 32 angles with ocml's sincosf in a cold block of its own:
        // inlined per angle that code was two thirds of this loop's instructions.
        f32x16 ft[4];
        float fmx = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 f4 = *reinterpret_cast<const f32x4 *>(dv + kVFreq + 32 * h + 4 * q);
            fmx = fmaxf(fmaxf(fmx, fmaxf(fabsf(f4[0]), fabsf(f4[1]))), fmaxf(fabsf(f4[2]), fabsf(f4[3])));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = 4 * q + j;
                float sn, cs;
                fourier_sincos(xv, f4[j], &sn, &cs);
                ft[s >> 4][s & 15] = cs;
                ft[2 + (s >> 4)][s & 15] = sn;
            }
        }
        if (__builtin_expect(!(fabsf(xv) * fmx <= 5215.0f), 0)) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 f4 = *reinterpret_cast<const f32x4 *>(dv + kVFreq + 32 * h + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int s = 4 * q + j;
                    float sn, cs, p;
                    sincosf(fourier_angle(xv, f4[j], &p), &sn, &cs);
                    ft[s >> 4][s & 15] = cs;
                    ft[2 + (s >> 4)][s & 15] = sn;
                }
            }
        }
        // ---- h = W1 [features, x] + b1, centred (the host centred W1 / b1 over the output features): sines and cosines
        // are a static-scale operand (|.| <= 1), the 129th input column (x itself) is a rank-1 term that joins the bias in
        // the one fma that also brings the accumulators back from the operands' scale
        make_operand_static(op, ft, 1.0f, 1.0f);
        f32x16 acc_h[4];
        sync_for_stage();
        gemm_raw<WAVES, MODE, true>(acc_h, op, buf0, lane, mats + (size_t)(2 * d + 1) * kMatFloats, buf1, wave);
        skew_after_gemm(skew);
        scale_add(acc_h, msc[2 * d], [&](int t, int q) {
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(dv + kVB1 + 32 * t + 8 * q + 4 * h);
            const f32x4 wl = *reinterpret_cast<const f32x4 *>(dv + kVW1Last + 32 * t + 8 * q + 4 * h);
            return f32x4{fmaf(xv, wl[0], b1[0]), fmaf(xv, wl[1], b1[1]), fmaf(xv, wl[2], b1[2]), fmaf(xv, wl[3], b1[3])};
        });
        if constexpr (MODE == kModeBf16) layer_norm<true, true>(acc_h, dv + kVG1, dv + kVBe1, h, eps);   // bf16: W1 as it is
        else layer_norm_centered<true>(acc_h, dv + kVG1, dv + kVBe1, h, eps);
        // ---- y += W2 h: every input dim accumulates in the same scale, nothing happens between these GEMMs
        make_operand_static(op, acc_h, 1.0f, 1.0f);
        sync_for_stage();
        gemm_raw<WAVES, MODE>(acc_y, op, buf1, lane,
                              tail.fold && d == IN - 1 ? tail.wt : mats + (size_t)(2 * d + 2) * kMatFloats, buf0, wave);
        skew_after_gemm(skew);
    }
    // y = U2 acc + sum_d b2_d (+ the summed categorical embeddings, :65-66)
    scale_add(acc_y, msc[kTblU2], BiasTerm{out_vec + kVB2, h});
    if (extra != nullptr) {
        f32x16 e[4];
        load_row(e, extra, rowc / extra_div, h);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc_y[t] += e[t];
    }

    // ---- to_out: LN -> ReLU -> Linear (:67)
    layer_norm<true, true>(acc_y, out_vec + kVOutG, out_vec + kVOutBe, h, eps);
    make_operand_static(op, acc_y, msc[kTblSo], msc[kTblSo + 1]);
    const int64_t row0 = row - (lane & 31);
    if (!tail.fold) {
        f32x16 acc_o[4];
        sync_for_stage();
        gemm_raw<WAVES, MODE, true>(acc_o, op, buf0, lane, tail.n_out > 0 ? tail.wt : nullptr, buf1, wave);
        skew_after_gemm(skew);
        scale_add(acc_o, msc[2 * IN], BiasTerm{out_vec + kVOutB, h});
        if (normalize_out) layer_norm<false, false>(acc_o, nullptr, nullptr, h, eps);
        if (tail.n_out == 0) {
            if (normalize_out == 2) store_tile2_split(acc_o, reinterpret_cast<_Float16 *>(y), row0, n_rows, xpose, lane, 2048.0f);
            else if (normalize_out == 3) store_tile2_bf16(acc_o, reinterpret_cast<__bf16 *>(y), row0, n_rows, xpose, lane);
            else store_tile2(acc_o, y, row0, n_rows, xpose, lane);
            skew_end(skew);
            return;
        }
        make_operand(op, acc_o);
    }
    // ---- tail: the embedding (folded: its last hidden vector) is the operand of n_out more layers; matrix j sits in
    // buffer (j + 1) & 1 behind to_out's matrix, in buffer j & 1 when that GEMM is folded away
    const int first_buf = tail.fold ? 0 : 1;
    const float *tsc = tail.wt + (size_t)tail.n_out * kMatFloats;
    // The stores of layer j are issued AFTER the stage barrier of layer j + 1: that barrier begins with vmcnt(0) (the LDS-DMA of
    // the layer's matrix must have landed), and vmcnt counts stores too — issued in front of it, every wave of the block sat
    // out the round trip of its sixteen stores to L2 three times per tile.  Behind it they have the whole GEMM to drain: 1 % of
    // the launch at 5.76 M rows (an ablation without the stores says they cost 1.0 of its 3.9 ms: 8.8 GB, mostly HBM time that
    // does not hide under the GEMMs) (the finished accumulators wait in a second register set).
    auto tail_gemm = [&](int j, f32x16 (&acc)[4]) {
        // (split mode: the bias joins in the one fma that brings the accumulators back from the operands' scale)
        gemm_add<WAVES, MODE>(acc, op, wlds + ((j + first_buf) & 1) * kMatVec4, tsc + 2 * j, lane,
                              BiasTerm{svec + kVecBlock + j * kC, h},
                              j + 1 < tail.n_out ? tail.wt + (size_t)(j + 1) * kMatFloats : nullptr,
                              wlds + ((j + first_buf + 1) & 1) * kMatVec4, wave);
        skew_after_gemm(skew);
    };
    auto tail_store = [&](int j, f32x16 (&acc)[4]) {
        if (tail.gru_steps > 0) {
            if (gru_tile >= 0) {            // (uniform per wave) the padding sequences of the last block are written too
                float *dst = tail.y[j] + gru_tile * (int64_t)(kRowsPerWave * kC) + lane * 4;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = acc[t][4 * q + i];
                        // (non-temporal: 8.8 GB per launch at cfg3 that nothing reads before the recurrence streams it back)
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst + (4 * t + q) * 256));
                    }
            }
        } else {
            store_tile2(acc, tail.y[j], row0, n_rows, xpose, lane);
        }
    };
    f32x16 acc_prev[4];
#pragma unroll 1
    for (int j = 0; j < tail.n_out; ++j) {
        sync_for_stage();
        if (j > 0) tail_store(j - 1, acc_prev);
        f32x16 acc[4];
        tail_gemm(j, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc_prev[t] = acc[t];
    }
    tail_store(tail.n_out - 1, acc_prev);
    skew_end(skew);
}

}  // namespace

#include "fourier_pair.hip.h"

extern "C" size_t qcn_fourier_pack_floats(int input_dim)
{
    return (size_t)(2 * input_dim + 1) * kMatFloats + (size_t)input_dim * kDimVecs + kOutVecs + kScales;
}

// launches of at least this many rows take the two-tile kernel (QCN_PAIR_MIN_ROWS overrides, for A/B runs)
static int64_t pair_min_rows()
{
    static const int64_t v = [] {
        const char *env = qcn_knob("QCN_PAIR_MIN_ROWS");
        return env ? (int64_t)atoll(env) : (int64_t)192 * kPairBlockRows;
    }();
    return v;
}

static int launch_fourier(const float *x, int input_dim, const float *extra, int64_t extra_div, const float *pack,
                          int64_t n_rows, int normalize_out, float *y, const FourierTail &tail, int mode, void *stream)
{
    if (!valid_mode(mode)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows < 0 || input_dim < 0 || input_dim > 4) return n_rows < 0 ? QCN_ERR_INVALID_ARGUMENT : QCN_ERR_UNSUPPORTED;
    if (n_rows == 0) return QCN_OK;
    if (!pack || (tail.n_out == 0 && !y) || (input_dim > 0 && !x) || (input_dim == 0 && !extra)) return QCN_ERR_INVALID_ARGUMENT;
    if (extra_div < 1 || extra_div > 0x7fffffff || normalize_out < 0 || normalize_out > 3) return QCN_ERR_INVALID_ARGUMENT;
    hipStream_t s = qcn_stream(stream);
    // Two tiles per wave (fourier_pair.hip.h) for the launches that fill the chip and whose rows carry three or four input
    // dims; the one-tile kernel below keeps the small launches (its 2- and 4-wave blocks spread few rows over more CUs), the
    // exact mode, input_dim < 3, categorical embeddings and the tail forms (measured, 800 k rows, split mode, same box:
    // input_dim 3 / 4 take 0.90-0.93 of the one-tile kernel's time, input_dim 2 takes 1.1, the folded tail the same time —
    // its three outputs are bound by their stores).  QCN_PAIR_TAIL=1 builds the tail / input_dim 2 instantiations for A/B runs.
#ifdef QCN_PAIR_TAIL
    const bool pair_ok = input_dim >= 2 && (tail.n_out == 0 || tail.fold);
#else
    const bool pair_ok = input_dim >= 3 && tail.n_out == 0;
#endif
    if (mode != QCN_MATMUL_EXACT_F32 && extra == nullptr && pair_ok && n_rows >= pair_min_rows()) {
        const unsigned pgrid = (unsigned)((n_rows + kPairBlockRows - 1) / kPairBlockRows);
#define QCN_PAIR(IN_, M_, T_, N_)                                                                                         \
    hipLaunchKernelGGL((fourier_pair_kernel<IN_, M_, T_, N_>), dim3(pgrid), dim3(64 * kPairWaves), 0, s, x, pack, y, n_rows, tail)
#ifdef QCN_PAIR_TAIL
#define QCN_PAIR_T(IN_, M_) if (tail.n_out > 0) QCN_PAIR(IN_, M_, true, 0); else
#else
#define QCN_PAIR_T(IN_, M_)
#endif
#define QCN_PAIR_N(IN_, M_)                                                          \
    do {                                                                             \
        QCN_PAIR_T(IN_, M_)                                                          \
        if (normalize_out == 0) QCN_PAIR(IN_, M_, false, 0);                         \
        else if (normalize_out == 1) QCN_PAIR(IN_, M_, false, 1);                    \
        else if (normalize_out == 2) QCN_PAIR(IN_, M_, false, 2);                    \
        else QCN_PAIR(IN_, M_, false, 3);                                            \
    } while (0)
#define QCN_PAIR_M(IN_)                                                              \
    do {                                                                             \
        if (mode == QCN_MATMUL_SPLIT_F16) QCN_PAIR_N(IN_, QCN_MATMUL_SPLIT_F16);     \
        else QCN_PAIR_N(IN_, QCN_MATMUL_BF16);                                       \
    } while (0)
#ifdef QCN_EXP_ONLY
        QCN_PAIR(3, QCN_MATMUL_SPLIT_F16, false, 1);
#else
#ifdef QCN_PAIR_TAIL
        if (input_dim == 2) QCN_PAIR_M(2);
        else
#endif
        if (input_dim == 3) QCN_PAIR_M(3);
        else QCN_PAIR_M(4);
#endif
#undef QCN_PAIR_M
#undef QCN_PAIR_N
#undef QCN_PAIR_T
#undef QCN_PAIR
#ifdef QCN_SLOT_STAMPS
        {
            unsigned long long hst[kPairWaves][32];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(hst, HIP_SYMBOL(g_slot_stamps), sizeof(hst));
            for (int w = 0; w < kPairWaves; ++w) {
                printf("slot stamps in=%d wave %d:", input_dim, w);
                for (int i = 1; i < 32; ++i) printf(" %llu", hst[w][i] - hst[w][i - 1]);
                printf("\n");
            }
        }
#endif
#ifdef QCN_PAIR_STAMPS
        {
            unsigned long long hst[kPairWaves][64];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(hst, HIP_SYMBOL(g_pair_stamps), sizeof(hst));
            for (int w = 0; w < kPairWaves; ++w) {
                printf("pair stamps in=%d wave %d:", input_dim, w);
                for (int i = 1; i < 64 && hst[w][i] > hst[w][i - 1]; ++i) printf(" %llu", hst[w][i] - hst[w][i - 1]);
                printf("\n");
            }
        }
#endif
        return hipGetLastError() == hipSuccess ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
    }
    // (GRU-input form: one tile per (sequence block, step), the last sequence block padded to 32)
    const int64_t tile_rows = tail.gru_steps > 0 ? ((tail.n_seq + 31) / 32) * 32 * tail.gru_steps : n_rows;
    const int waves = pick_waves(tile_rows, 1.5);
    const unsigned grid = (unsigned)((tile_rows + 32 * waves - 1) / (32 * waves));
    const int skew_mode = chain_skew_mode(0);
#define QCN_LAUNCH(S, W)                                                                                           \
    hipLaunchKernelGGL((fourier_kernel<QCN_IN, S, W>), dim3(grid), dim3(64 * W), 0, s, x, extra, (int)extra_div, pack, y, \
                       n_rows, normalize_out, tail, skew_mode)
#ifdef QCN_EXP_ONLY   // compile-time experiments: one instantiation per mode
    hipLaunchKernelGGL((fourier_kernel<3, QCN_MATMUL_SPLIT_F16, 8>), dim3(grid), dim3(512), 0, s, x, extra, (int)extra_div, pack, y,
                       n_rows, normalize_out, tail, skew_mode);
    hipLaunchKernelGGL((fourier_kernel<3, QCN_MATMUL_BF16, 8>), dim3(grid), dim3(512), 0, s, x, extra, (int)extra_div, pack, y,
                       n_rows, normalize_out, tail, skew_mode);
    if (false)
#endif
#ifndef QCN_EXP_ONLY
    switch (input_dim) {
#define QCN_IN 0
    case 0: QCN_CHAIN_DISPATCH(mode, waves); break;
#undef QCN_IN
#define QCN_IN 1
    case 1: QCN_CHAIN_DISPATCH(mode, waves); break;
#undef QCN_IN
#define QCN_IN 2
    case 2: QCN_CHAIN_DISPATCH(mode, waves); break;
#undef QCN_IN
#define QCN_IN 3
    case 3: QCN_CHAIN_DISPATCH(mode, waves); break;
#undef QCN_IN
#define QCN_IN 4
    default: QCN_CHAIN_DISPATCH(mode, waves); break;
#undef QCN_IN
    }
#else
    {}
#endif
#undef QCN_LAUNCH
    const bool ok = hipGetLastError() == hipSuccess;
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}

extern "C" int qcn_fourier_embed_f32(const float *x, int input_dim, const float *extra, int64_t extra_div,
                                     const float *pack, int64_t n_rows, int normalize_out, float *y, int matmul_mode,
                                     void *stream)
{
    FourierTail tail{};
    return launch_fourier(x, input_dim, extra, extra_div, pack, n_rows, normalize_out, y, tail, matmul_mode, stream);
}

extern "C" int qcn_fourier_linear_f32(const float *x, int input_dim, const float *extra, int64_t extra_div,
                                      const float *pack, int64_t n_rows, const float *wt, const float *bias, int n_out,
                                      float *y0, float *y1, float *y2, int matmul_mode, void *stream)
{
    if (n_out < 1 || n_out > 3 || !wt || !y0 || (n_out > 1 && !y1) || (n_out > 2 && !y2)) return QCN_ERR_INVALID_ARGUMENT;
    FourierTail tail{wt, bias, {y0, y1, y2}, n_out, 0, 0, 0};
    return launch_fourier(x, input_dim, extra, extra_div, pack, n_rows, 0, nullptr, tail, matmul_mode, stream);
}

extern "C" int qcn_fourier_folded_linear_f32(const float *x, int input_dim, const float *extra, int64_t extra_div,
                                             const float *pack, int64_t n_rows, const float *wt, const float *bias, int n_out,
                                             float *y0, float *y1, float *y2, int matmul_mode, void *stream)
{
    if (n_out < 1 || n_out > 3 || !wt || !y0 || (n_out > 1 && !y1) || (n_out > 2 && !y2)) return QCN_ERR_INVALID_ARGUMENT;
    if (input_dim < 1) return QCN_ERR_UNSUPPORTED;   // the last in-loop GEMM stages the first folded matrix
    FourierTail tail{wt, bias, {y0, y1, y2}, n_out, 1, 0, 0};
    return launch_fourier(x, input_dim, extra, extra_div, pack, n_rows, 0, nullptr, tail, matmul_mode, stream);
}

extern "C" int64_t qcn_gru_blocked_rows(int64_t n_seq, int n_steps)
{
    return n_seq < 0 || n_steps < 1 ? -1 : ((n_seq + 31) / 32) * 32 * (int64_t)n_steps;
}

extern "C" int qcn_fourier_gru_input_f32(const float *x, int input_dim, const float *pack, int64_t n_seq, int n_steps,
                                         const float *wt, const float *bias, int folded, float *gi_r, float *gi_z, float *gi_n,
                                         int matmul_mode, void *stream)
{
    if (n_seq < 0 || n_steps < 1 || !wt || !gi_r || !gi_z || !gi_n) return QCN_ERR_INVALID_ARGUMENT;
    if (input_dim < 1) return QCN_ERR_UNSUPPORTED;
    if (n_seq == 0) return QCN_OK;
    FourierTail tail{wt, bias, {gi_r, gi_z, gi_n}, 3, folded ? 1 : 0, n_steps, n_seq};
    // (the two-tile kernel is never taken: a tail with outputs keeps the one-tile form, see launch_fourier)
    return launch_fourier(x, input_dim, nullptr, 1, pack, n_seq * n_steps, 0, nullptr, tail, matmul_mode, stream);
}
