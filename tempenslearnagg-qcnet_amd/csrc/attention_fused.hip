// A3 (fused positional form) — per-destination attention that never materialises the per-edge projections.
//
// Reference math (layers/attention_layer.py:94-101): k_e = k[j] + W_kr LN_r(r_e), v_e = v[j] + W_vr LN_r(r_e) + b_vr.
// With rhat_e = (r_e - mean)/std the normalised edge feature (layer independent) and gamma/beta the layer's LN_r affine:
//   q_i^h . (W_kr^h LN_r(r_e))  = utilde_i^h . rhat_e + const_i^h      utilde_i^h = gamma * (W_kr^h)^T q_i^h   [128]
//   sum_e a_e^h (W_vr^h LN_r(r_e) + b^h) = W_vr^h (gamma * s_i^h) + A_i^h (W_vr^h beta + b^h)
//                                           s_i^h = sum_e a_e^h rhat_e [128],  A_i^h = sum_e a_e^h
// const_i^h is the same for every edge of a destination and head, so it cancels in the softmax.  Both per-edge
// [E,128]x[128,128] GEMMs of the reference become per-destination 16x128 products that run in the kernel's
// prologue / epilogue out of LDS-resident weights; the edge loop reads rhat_e (512 B), k[j], v[j] and nothing else.
// Algorithmic HBM bytes: 512*E_base (rhat) + 8*E_base (col, eid) + gathers of k/v (cache-served) + 1 KiB per
// destination (q in, out) — SURVEY.md §8(d) "r-projection fused" row: 1552*E + 1024*n_dst.
//
// Mapping: one wave64 per destination, lane l owns channels 2l, 2l+1 (head = l >> 3).  16 waves (1024 threads) per
// workgroup, one workgroup per CU: W_kr*gamma and W_vr*gamma (64 KiB each) live in LDS for the workgroup's lifetime,
// XOR-swizzled at 8-byte granularity so that both the row read of the prologue and the 16-row column-slice read of the
// epilogue are bank-conflict free.  Mode replication of the decoder (qcnet_decoder.py:190-192,208-211: the K mode
// queries of an agent share one edge list and one r) is expressed by `rep`/`layout`: consecutive waves take the K modes
// of one CSR row, so rhat_e is fetched from HBM once and served to the other K-1 waves from L1/L2.
#include "common.hip.h"

namespace {

constexpr int kC = QCN_CHANNELS;       // 128
constexpr int kH = QCN_HEADS;          // 8
constexpr int kWaves = 16;
constexpr int kUnroll = 4;
constexpr float kLog2e = 1.4426950408889634f;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ float head_sum(float v)
{
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    return v;
}

__device__ __forceinline__ float bcast_lane(float v, int lane)   // lane is a compile-time constant at every call
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Pairwise exchange-and-add across the two wave halves: afterwards lanes 0-31 hold (x summed over the lane pair
// l, l+32) and lanes 32-63 hold (y summed over the pair).
// (Inline asm, not __builtin_amdgcn_permlane32_swap: hipcc 7.2 folds the builtin's two results into one register
// when both are consumed by one add — verified on the box, scripts/dbg/butterfly.hip.  The s_nop covers the
// VALU-write -> permlane-read hazard, which nothing pads inside an asm string.)
__device__ __forceinline__ float fold32(float x, float y)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
// Same across the 16-lane rows: even rows end with x (pair l, l+16), odd rows with y.
__device__ __forceinline__ float fold16(float x, float y)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 ld2(const float *__restrict__ base, int64_t row, int lane)
{
    return *reinterpret_cast<const float2 *>(base + row * kC + 2 * lane);
}

struct FusedArgs {
    const float *q, *k, *v, *rhat;
    const int32_t *rowptr, *col, *eid;
    const float *wk, *wv, *bv;     // wk[o][c] = W_kr[o][c]*gamma[c]; wv likewise; bv = W_vr beta + b_vr
    float *out;
    int64_t n_dst;
    int n_rows, rep, layout;       // layout 0: node = row; 1: node = mode*n_rows + row; 2: node = row*rep + mode
    int col_shift;                 // source row = col[slot] - col_shift (window views of an absolute-time graph)
    // frame groups (all sliding frames in one launch): destination idx -> group g = idx / group_rows, CSR row =
    // idx % group_rows + g * group_row_shift, source row = col - col_shift - g * group_col_shift + g * group_src_stride
    int group_rows, group_row_shift, group_col_shift, group_src_stride;
};

template <bool HAS_EID>
__global__ __launch_bounds__(1024) void attn_fused_kernel(FusedArgs a)
{
    // [o][slot ^ (o & 15)] with slot = channel pair; 64 KiB each
    __shared__ __attribute__((aligned(16))) float2 s_wk[kC][64];
    __shared__ __attribute__((aligned(16))) float2 s_wv[kC][64];
    __shared__ __attribute__((aligned(16))) float2 s_vec[kWaves][64];    // per-wave s^h staging
    __shared__ __attribute__((aligned(16))) float s_out[kWaves][kC];     // per-wave projected output

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < kC * 64; i += 1024) {
        const int o = i >> 6, slot = i & 63;
        s_wk[o][slot ^ (o & 15)] = *reinterpret_cast<const float2 *>(a.wk + o * kC + 2 * slot);
        s_wv[o][slot ^ (o & 15)] = *reinterpret_cast<const float2 *>(a.wv + o * kC + 2 * slot);
    }
    __syncthreads();

    // contiguous chunk of destinations per workgroup (XCD-contiguous through the block remap)
    const int64_t nb = gridDim.x;
    const int64_t lb = qcn_xcd_remap(blockIdx.x, gridDim.x);
    const int64_t per = (a.n_dst + nb - 1) / nb;
    const int64_t begin = lb * per;
    const int64_t end_idx = begin + per < a.n_dst ? begin + per : a.n_dst;
    const float2 bias = ld2(a.bv, 0, lane);
    const int d16 = lane & 15, quarter = lane >> 4;

    for (int64_t idx = begin + wave; idx < end_idx; idx += kWaves) {
        int64_t row = idx, node = idx;
        if (a.layout == 1) { row = idx / a.rep; node = (idx % a.rep) * (int64_t)a.n_rows + row; }
        else if (a.layout == 2) { row = idx / a.rep; }
        int src_off = -a.col_shift;
        if (a.group_rows > 0) {
            const int g = (int)(idx / a.group_rows);
            row = idx - (int64_t)g * a.group_rows + (int64_t)g * a.group_row_shift;
            src_off += g * (a.group_src_stride - a.group_col_shift);
        }
        const int beg = a.rowptr[row], end = a.rowptr[row + 1];
        float2 qv = ld2(a.q, node, lane);
        qv.x *= 0.25f * kLog2e;      // head_dim ** -0.5 (layers/attention_layer.py:38), logits kept in base 2
        qv.y *= 0.25f * kLog2e;

        float m = -INFINITY, l = 0.f;
        float2 accv = make_float2(0.f, 0.f);
        float2 s[kH];
#pragma unroll
        for (int h = 0; h < kH; ++h) s[h] = make_float2(0.f, 0.f);

        if (end > beg) {
            // ---- prologue: utilde^h[c] = sum_d wk[16h + d][c] * q'^h[d] for this lane's two channels
            float2 u[kH];
#pragma unroll
            for (int h = 0; h < kH; ++h) {
                float ux = 0.f, uy = 0.f;
#pragma unroll
                for (int dp = 0; dp < 8; ++dp) {
                    const float qa = bcast_lane(qv.x, 8 * h + dp), qb = bcast_lane(qv.y, 8 * h + dp);
                    const int o0 = 16 * h + 2 * dp, o1 = o0 + 1;
                    const float2 wa = s_wk[o0][lane ^ (o0 & 15)], wb = s_wk[o1][lane ^ (o1 & 15)];
                    ux = fmaf(wa.x, qa, ux); uy = fmaf(wa.y, qa, uy);
                    ux = fmaf(wb.x, qb, ux); uy = fmaf(wb.y, qb, uy);
                }
                u[h] = make_float2(ux, uy);
            }

            // ---- edge loop
            for (int j = beg; j < end; j += kUnroll) {
                float2 kk[kUnroll], vv[kUnroll], rr[kUnroll];
#pragma unroll
                for (int t = 0; t < kUnroll; ++t) {
                    const int jj = min(j + t, end - 1);           // clamp, mask below (no per-load branches)
                    const int src = a.col[jj] + src_off;
                    const int e = HAS_EID ? a.eid[jj] : jj;
                    kk[t] = ld2(a.k, src, lane);
                    vv[t] = ld2(a.v, src, lane);
                    rr[t] = ld2(a.rhat, e, lane);
                }
                float sim[kUnroll];
                float smax = -INFINITY;
#pragma unroll
                for (int t = 0; t < kUnroll; ++t) {
                    float p[kH];
#pragma unroll
                    for (int h = 0; h < kH; ++h) p[h] = fmaf(u[h].y, rr[t].y, u[h].x * rr[t].x);
                    // transposed butterfly: 8 per-head partials over 64 lanes -> head (lane >> 3) on every lane
                    const float t0 = fold32(p[0], p[4]), t1 = fold32(p[1], p[5]);
                    const float t2 = fold32(p[2], p[6]), t3 = fold32(p[3], p[7]);
                    const float w0 = fold16(t0, t2), w1 = fold16(t1, t3);
                    const bool hi = (lane & 8) != 0;
                    const float keep = hi ? w1 : w0, give = hi ? w0 : w1;
                    float part = keep + dpp_mov<0x128>(give);      // row_ror:8
                    part += fmaf(qv.y, kk[t].y, qv.x * kk[t].x);
                    part = head_sum(part);
                    sim[t] = (j + t < end) ? part : -INFINITY;
                    smax = fmaxf(smax, sim[t]);
                }
                const float mn = fmaxf(m, smax);
                if (!__all(mn == m)) {                             // running max moved for some head: rescale
                    const float alpha = __builtin_amdgcn_exp2f(m - mn);
                    l *= alpha; accv.x *= alpha; accv.y *= alpha;
#pragma unroll
                    for (int h = 0; h < kH; ++h) {
                        const float ah = bcast_lane(alpha, 8 * h);
                        s[h].x *= ah; s[h].y *= ah;
                    }
                    m = mn;
                }
#pragma unroll
                for (int t = 0; t < kUnroll; ++t) {
                    const float pe = __builtin_amdgcn_exp2f(sim[t] - m);
                    l += pe;
                    accv.x = fmaf(pe, vv[t].x, accv.x);
                    accv.y = fmaf(pe, vv[t].y, accv.y);
#pragma unroll
                    for (int h = 0; h < kH; ++h) {
                        const float ph = bcast_lane(pe, 8 * h);
                        s[h].x = fmaf(ph, rr[t].x, s[h].x);
                        s[h].y = fmaf(ph, rr[t].y, s[h].y);
                    }
                }
            }

            // ---- epilogue: out^h[d] = sum_c wv[16h + d][c] * s^h[c]; lane (d16, quarter) sums 32 channels
#pragma unroll
            for (int h = 0; h < kH; ++h) {
                s_vec[wave][lane] = s[h];
                wave_lds_fence();
                float acc = 0.f;
                const int o = 16 * h + d16;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int slot = 16 * quarter + jj;
                    const float2 w = s_wv[o][slot ^ d16];
                    const float2 sv = s_vec[wave][slot];
                    acc = fmaf(w.x, sv.x, acc);
                    acc = fmaf(w.y, sv.y, acc);
                }
                acc += __shfl_xor(acc, 16);
                acc += __shfl_xor(acc, 32);
                if (quarter == 0) s_out[wave][o] = acc;
                wave_lds_fence();
            }
        }
        float2 proj = make_float2(0.f, 0.f);
        if (end > beg) proj = *reinterpret_cast<const float2 *>(&s_out[wave][2 * lane]);
        const float inv = 1.0f / (l + 1e-16f);        // PyG softmax denominator
        const float asum = l * inv;                   // A_i^h: 1 for non-empty segments, 0 for empty ones
        float2 o2;
        o2.x = fmaf(inv, proj.x + accv.x, asum * bias.x);
        o2.y = fmaf(inv, proj.y + accv.y, asum * bias.y);
        *reinterpret_cast<float2 *>(a.out + node * kC + 2 * lane) = o2;
        wave_lds_fence();
    }
}

// Row normalisation rhat = (r - mean) / sqrt(var + eps): the layer-independent half of every LN_r
// (layers/attention_layer.py:63,82).  One wave per row, two values per lane.
__global__ __launch_bounds__(256) void row_normalize_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                            int64_t n_rows, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float2 v = ld2(x, row, lane);
    float sum = v.x + v.y;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum * (1.0f / kC);
    const float dx = v.x - mean, dy = v.y - mean;
    float sq = dx * dx + dy * dy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq * (1.0f / kC) + eps);
    *reinterpret_cast<float2 *>(y + row * kC + 2 * lane) = make_float2(dx * rstd, dy * rstd);
}

}  // namespace

extern "C" int qcn_row_normalize_f32(const float *x, float *y, int64_t n_rows, float eps, void *stream)
{
    if (n_rows < 0) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows == 0) return QCN_OK;
    if (!x || !y) return QCN_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(row_normalize_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, qcn_stream(stream), x, y,
                       n_rows, eps);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

extern "C" int qcn_attn_fused_f32(const float *q, const float *k, const float *v, const float *rhat,
                                  const int32_t *rowptr, const int32_t *col, const int32_t *eid, const float *wk,
                                  const float *wv, const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows,
                                  int64_t num_edges, int rep, int layout, int64_t col_shift, const int64_t *group,
                                  float *out, void *stream)
{
    if (n_dst < 0 || n_src < 0 || n_rows < 0 || num_edges < 0 || rep < 1) return QCN_ERR_INVALID_ARGUMENT;
    if (layout < 0 || layout > 2) return QCN_ERR_INVALID_ARGUMENT;
    if (group != nullptr) {
        if (layout != 0 || rep != 1 || group[0] < 1 || n_dst % group[0] != 0) return QCN_ERR_INVALID_ARGUMENT;
        const int64_t n_groups = n_dst / group[0];
        if (group[0] + (n_groups - 1) * group[1] > n_rows) return QCN_ERR_INVALID_ARGUMENT;
        for (int i = 0; i < 4; ++i)
            if (group[i] < 0 || group[i] >= ((int64_t)1 << 31)) return QCN_ERR_INVALID_ARGUMENT;
    } else if (layout == 0 ? (rep != 1 || n_rows != n_dst) : n_rows * rep != n_dst) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst == 0) return QCN_OK;
    if (!q || !out || !rowptr || !wk || !wv || !bv) return QCN_ERR_INVALID_ARGUMENT;
    if (num_edges > 0 && (!k || !v || !col || !rhat)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst >= ((int64_t)1 << 31) || n_src >= ((int64_t)1 << 31) || num_edges >= ((int64_t)1 << 31))
        return QCN_ERR_UNSUPPORTED;
    if (col_shift < 0 || col_shift >= ((int64_t)1 << 31)) return QCN_ERR_INVALID_ARGUMENT;
    FusedArgs args{q, k, v, rhat, rowptr, col, eid, wk, wv, bv, out, n_dst, (int)n_rows, rep, layout, (int)col_shift,
                   group ? (int)group[0] : 0, group ? (int)group[1] : 0, group ? (int)group[2] : 0, group ? (int)group[3] : 0};
    // one workgroup per CU (144 KiB of LDS); every workgroup gets a contiguous chunk of >= 16 destinations
    int64_t grid = (n_dst + kWaves - 1) / kWaves;
    if (grid > 256) grid = 256;
    hipStream_t s = qcn_stream(stream);
    void *token = qcn_prof_begin(QCN_PROF_ATTN_FUSED, 1552.0 * (double)num_edges * (double)rep + 1024.0 * (double)n_dst, s);
    if (eid)
        hipLaunchKernelGGL((attn_fused_kernel<true>), dim3((unsigned)grid), dim3(1024), 0, s, args);
    else
        hipLaunchKernelGGL((attn_fused_kernel<false>), dim3((unsigned)grid), dim3(1024), 0, s, args);
    const bool ok = hipGetLastError() == hipSuccess;
    qcn_prof_end(token, s);
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}
