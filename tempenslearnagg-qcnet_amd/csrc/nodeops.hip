// A2 node side — everything of AttentionLayer.forward (layers/attention_layer.py:70-85,103-122) that is per NODE
// (the per-edge part is attention*.hip), as two launches per layer instead of ~15 library calls:
//
//   qcn_ln_linear_f32  : xn = LN(x);  y_j = W_j xn + b_j  for up to four 128-wide projections (q / k / v / s), xn kept
//                        for the gate (reference :74-79, :115-117, to_s of :108).
//   qcn_attn_post_f32  : g = sigmoid(W_g [agg, xn] + b_g);  u = agg + g * (s - agg)              (update, :106-108)
//                        x1 = x + LN_post(W_out u + b_out)                                        (:119, :83)
//                        out = x1 + LN_ffpost(W_2 relu(W_1 LN_ffpre(x1) + b_1) + b_2)             (:84, :121-122)
//
// Both use the row-tile machinery of mfma_tile.hip.h: a wave owns 32 complete rows, every LayerNorm / sigmoid / residual
// is an in-register epilogue, the 512-wide FFN hidden layer is produced and consumed in four 128-wide blocks and never
// leaves the CU.  HBM traffic per row: reads of agg, xn, s, x and one write (2.5 KiB) against 11 x 32 KiB of flops:
// MFMA-bound (fp32 matrix cores, exact fp32 fma chains).
#include "mfma_tile.hip.h"

namespace {

using namespace qcn_tile;

constexpr int kMat = kC * kC;

__global__ __launch_bounds__(256) void ln_linear_kernel(const float *__restrict__ x, int64_t n_rows,
                                                        const float *__restrict__ ln_g, const float *__restrict__ ln_b,
                                                        const float *__restrict__ wt, const float *__restrict__ bias,
                                                        int n_out, float *__restrict__ y0, float *__restrict__ y1,
                                                        float *__restrict__ y2, float *__restrict__ y3,
                                                        float *__restrict__ xn_out)
{
    __shared__ __attribute__((aligned(16))) Smem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * kRowsPerBlock + wave * kRowsPerWave;
    Stream st;
    st.chunk = 0;
    prefetch_chunk(st, wt, tid);

    f32x16 xv[4];
    load_rows(xv, x, row0, n_rows, lane);
    if (ln_g != nullptr) row_layer_norm<true, false>(xv, ln_g, ln_b, lane, 1e-5f);
    if (xn_out != nullptr) store_rows(xv, xn_out, row0, n_rows, lane);
    acc_to_tile(xv, sm.a[wave], lane);
#pragma unroll 1
    for (int j = 0; j < n_out; ++j) {
        f32x16 acc[4];
        fill_cols(acc, bias ? bias + j * kC : nullptr, lane);
        gemm_pass(acc, sm, st, wt + (size_t)j * kMat, j + 1 < n_out ? wt + (size_t)(j + 1) * kMat : nullptr, wave, lane,
                  tid);
        float *y = j == 0 ? y0 : j == 1 ? y1 : j == 2 ? y2 : y3;
        store_rows(acc, y, row0, n_rows, lane);
    }
}

// pack: 11 k-major matrices in consumption order [Wg_agg, Wg_x, Wout, W1_0, W2_0, W1_1, W2_1, W1_2, W2_2, W1_3, W2_3]
// then vectors [bg, bout, lnpost_g, lnpost_b, lnffpre_g, lnffpre_b, b1 (512), b2, lnffpost_g, lnffpost_b]
constexpr int kPostMats = 11;
constexpr int kVBg = 0, kVBout = kC, kVLpG = 2 * kC, kVLpB = 3 * kC, kVLfG = 4 * kC, kVLfB = 5 * kC, kVB1 = 6 * kC,
              kVB2 = 10 * kC, kVLqG = 11 * kC, kVLqB = 12 * kC, kPostVecs = 13 * kC;

__global__ __launch_bounds__(256) void attn_post_kernel(const float *__restrict__ agg, const float *__restrict__ xn,
                                                        const float *__restrict__ s, const float *__restrict__ xres,
                                                        const float *__restrict__ pack, int64_t n_rows,
                                                        float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) Smem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * kRowsPerBlock + wave * kRowsPerWave;
    const float *mats = pack;
    const float *vec = pack + (size_t)kPostMats * kMat;
    const float eps = 1e-5f;
    Stream st;
    st.chunk = 0;
    prefetch_chunk(st, mats, tid);

    // ---- gate: g = sigmoid(Wg [agg, xn] + bg)
    f32x16 va[4], acc[4];
    load_rows(va, agg, row0, n_rows, lane);
    acc_to_tile(va, sm.a[wave], lane);
    fill_cols(acc, vec + kVBg, lane);
    gemm_pass(acc, sm, st, mats, mats + kMat, wave, lane, tid);
    {
        f32x16 vx[4];
        load_rows(vx, xn, row0, n_rows, lane);
        acc_to_tile(vx, sm.a[wave], lane);
    }
    gemm_pass(acc, sm, st, mats + kMat, mats + 2 * kMat, wave, lane, tid);
    // ---- u = agg + g * (s - agg)
    {
        f32x16 vs[4];
        load_rows(vs, s, row0, n_rows, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float g = 1.0f / (1.0f + expf(-acc[t][r]));
                va[t][r] = fmaf(g, vs[t][r] - va[t][r], va[t][r]);
            }
    }
    acc_to_tile(va, sm.a[wave], lane);
    // ---- x1 = x + LN_post(Wout u + bout)
    fill_cols(acc, vec + kVBout, lane);
    gemm_pass(acc, sm, st, mats + 2 * kMat, mats + 3 * kMat, wave, lane, tid);
    row_layer_norm<true, false>(acc, vec + kVLpG, vec + kVLpB, lane, eps);
    f32x16 x1[4];
    load_rows(x1, xres, row0, n_rows, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) x1[t] += acc[t];
    // ---- feed-forward block, hidden layer in four 128-wide blocks
    f32x16 xn1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xn1[t] = x1[t];
    row_layer_norm<true, false>(xn1, vec + kVLfG, vec + kVLfB, lane, eps);
    f32x16 acc_o[4];
    fill_cols(acc_o, vec + kVB2, lane);
#pragma unroll 1
    for (int hb = 0; hb < 4; ++hb) {
        const float *w1 = mats + (size_t)(3 + 2 * hb) * kMat;
        acc_to_tile(xn1, sm.a[wave], lane);
        fill_cols(acc, vec + kVB1 + hb * kC, lane);
        gemm_pass(acc, sm, st, w1, w1 + kMat, wave, lane, tid);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
        acc_to_tile(acc, sm.a[wave], lane);
        gemm_pass(acc_o, sm, st, w1 + kMat, hb < 3 ? w1 + 2 * kMat : nullptr, wave, lane, tid);
    }
    row_layer_norm<true, false>(acc_o, vec + kVLqG, vec + kVLqB, lane, eps);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc_o[t] += x1[t];
    store_rows(acc_o, out, row0, n_rows, lane);
}

}  // namespace

extern "C" size_t qcn_attn_post_pack_floats(void) { return (size_t)kPostMats * kMat + kPostVecs; }

extern "C" int qcn_ln_linear_f32(const float *x, int64_t n_rows, const float *ln_weight, const float *ln_bias,
                                 const float *wt, const float *bias, int n_out, float *y0, float *y1, float *y2,
                                 float *y3, float *xn_out, void *stream)
{
    if (n_rows < 0 || n_out < 1 || n_out > 4) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows == 0) return QCN_OK;
    if (!x || !wt || !y0 || (n_out > 1 && !y1) || (n_out > 2 && !y2) || (n_out > 3 && !y3)) return QCN_ERR_INVALID_ARGUMENT;
    if ((ln_weight == nullptr) != (ln_bias == nullptr)) return QCN_ERR_INVALID_ARGUMENT;
    hipStream_t s = qcn_stream(stream);
    const unsigned grid = (unsigned)((n_rows + kRowsPerBlock - 1) / kRowsPerBlock);
    void *token = qcn_prof_begin(QCN_PROF_GEMM, (double)n_rows * n_out * 2.0 * kMat, s);
    hipLaunchKernelGGL(ln_linear_kernel, dim3(grid), dim3(256), 0, s, x, n_rows, ln_weight, ln_bias, wt, bias, n_out, y0,
                       y1, y2, y3, xn_out);
    const bool ok = hipGetLastError() == hipSuccess;
    qcn_prof_end(token, s);
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}

extern "C" int qcn_attn_post_f32(const float *agg, const float *xn_dst, const float *s_proj, const float *x_res,
                                 const float *pack, int64_t n_rows, float *out, void *stream)
{
    if (n_rows < 0) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows == 0) return QCN_OK;
    if (!agg || !xn_dst || !s_proj || !x_res || !pack || !out) return QCN_ERR_INVALID_ARGUMENT;
    hipStream_t s = qcn_stream(stream);
    const unsigned grid = (unsigned)((n_rows + kRowsPerBlock - 1) / kRowsPerBlock);
    void *token = qcn_prof_begin(QCN_PROF_GEMM, (double)n_rows * kPostMats * 2.0 * kMat, s);
    hipLaunchKernelGGL(attn_post_kernel, dim3(grid), dim3(256), 0, s, agg, xn_dst, s_proj, x_res, pack, n_rows, out);
    const bool ok = hipGetLastError() == hipSuccess;
    qcn_prof_end(token, s);
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}
