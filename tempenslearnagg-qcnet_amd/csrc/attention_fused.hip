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
// Mapping: one wave64 per destination.  Lane = 8 h + sub: head h, and
//   * for k / v / q / out the lane owns channels 2 lane, 2 lane + 1 (they belong to head h: coalesced 512-B row reads),
//   * for the 8 x 128 positional state (utilde^h, s^h) the lane owns the 16-channel slice 16 sub .. 16 sub + 15 of ITS
//     head: the per-edge logit is 16 in-lane FMAs + a 3-step DPP sum over the 8 lanes of the head, the weighted sum of
//     rhat 16 in-lane FMAs with the lane's own softmax weight — no cross-head traffic at all.  The rhat row is read
//     from HBM once, coalesced, and handed to the slice owners through a per-wave 512-B LDS row (one ds_write_b64 +
//     four broadcast ds_read_b128 per lane and edge).
// 16 waves (1024 threads) per workgroup, one workgroup per CU: W_kr*gamma and W_vr*gamma (64 KiB each) live in LDS for
// the workgroup's lifetime.  All LDS rows are stored in 16-byte chunks with chunk c of row o at c ^ (2 ((o >> 4) & 1) +
// (c >> 4)): the slice reads of 16 consecutive lanes (two heads x eight slices) then fall on 16 distinct bank groups.
// Rows and indices are requested one resp. two iterations ahead of their use.  Mode replication of the decoder
// (qcnet_decoder.py:190-192,208-211: the K mode queries of an agent share one edge list and one r) is expressed by
// `rep`/`layout`: consecutive waves take the K modes of one CSR row, so rhat_e is fetched from HBM once and served
// to the other K-1 waves from L1/L2.
#include "common.hip.h"
#include <cstdlib>

namespace {

constexpr int kC = QCN_CHANNELS;       // 128

constexpr int kWaves = 16;
constexpr int kSub = 2;                // edges whose positional slices are live at a time
constexpr float kLog2e = 1.4426950408889634f;

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;   // v_pk_fma_f32 operands

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppMirror8 = 0x141;     // row_half_mirror: lane i <-> 7 - i inside every group of 8

// sum over the 8 lanes of a head
__device__ __forceinline__ float head_sum(float v)
{
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppMirror8>(v);
    return v;
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

// the positional rows are read once per launch (0.5-1 GB per relation): non-temporal, so that they do not push the gathered
// k / v rows and the weights out of the caches (table attention: 4.36 -> 4.80 TB/s with the same policy, attention_modes.hip)
__device__ __forceinline__ float2 ld2_stream(const float *__restrict__ base, int64_t row, int lane)
{
#ifdef QCN_NO_NT
    return ld2(base, row, lane);
#else
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const f32x2 x = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(base + row * kC + 2 * lane));
    return make_float2(x[0], x[1]);
#endif
}

struct FusedArgs {
    const float *q, *k, *v, *rhat;
    const int32_t *rowptr, *col, *eid;
    const float *wk, *wv, *bv;     // wk[o][c] = W_kr[o][c]*gamma[c]; wv likewise; bv = W_vr beta + b_vr
    float *out;
    int64_t n_dst;
    int n_rows, rep, layout;       // layout 0: node = row; 1: node = mode*n_rows + row; 2: node = row*rep + mode
    int col_shift;                 // source row = col[slot] - col_shift (window views of an absolute-time graph)
    // optional indirections (layout 0): destination idx uses CSR row row_index[idx] and reads source rows
    // col - col_shift + src_offset[idx] — many destination sets (sliding frames, per-frame special steps) over one CSR
    const int32_t *row_index, *src_offset;
};

template <int KU> struct EdgeRows {
    float2 k[KU], v[KU], r[KU];
};

template <bool HAS_EID, int KU>
__device__ __forceinline__ void load_indices(const FusedArgs &a, int j, int end, int src_off, int (&src)[KU], int (&e)[KU])
{
#pragma unroll
    for (int t = 0; t < KU; ++t) {
        const int jj = min(j + t, end - 1);           // clamp, masked at the use (no per-load branches)
        src[t] = a.col[jj] + src_off;
        e[t] = HAS_EID ? a.eid[jj] : jj;
    }
}

template <int KU>
__device__ __forceinline__ void load_rows(const FusedArgs &a, const int (&src)[KU], const int (&e)[KU], int lane, EdgeRows<KU> &r)
{
#pragma unroll
    for (int t = 0; t < KU; ++t) {
        r.k[t] = ld2(a.k, src[t], lane);
        r.v[t] = ld2(a.v, src[t], lane);
        r.r[t] = ld2_stream(a.rhat, e[t], lane);
    }
}

// R = number of mode copies of one CSR row that a wave carries together (1, or a divisor of `rep`): the edge rows are
// loaded and handed over through LDS once per edge and serve R softmax streams — the vector-memory, L1 and LDS traffic
// per (edge, mode) visit drops R-fold, only the FMAs remain per visit.  R > 1 runs 8 waves of up to 256 registers.
template <bool HAS_EID, int R>
__global__ __launch_bounds__(R == 1 ? 1024 : (R <= 3 ? 512 : 256)) void attn_fused_kernel(FusedArgs a)
{
    constexpr int kW = R == 1 ? kWaves : (R <= 3 ? kWaves / 2 : kWaves / 4);
    // edges per iteration = rows requested per prefetch stage (4 with R = 3 spills 48 registers and is 1.6x slower)
    constexpr int kU = 2;
    // one array (see the chunk swizzle above): [0, 128) rows of wk, [128, 256) rows of wv, then kU staging rows per wave
    __shared__ __attribute__((aligned(16))) float lds[(2 * kC + kW * kU) * kC];
    float *const s_wk = lds, *const s_wv = lds + kC * kC;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 3, sub = lane & 7;
    float *const s_stage = lds + (2 * kC + wave * kU) * kC;

    for (int i = tid; i < kC * 32; i += 64 * kW) {
        const int o = i >> 5, c = i & 31;
        const int pc = c ^ (2 * ((o >> 4) & 1) + (c >> 4));
        *reinterpret_cast<f32x4 *>(s_wk + o * kC + 4 * pc) = *reinterpret_cast<const f32x4 *>(a.wk + o * kC + 4 * c);
        *reinterpret_cast<f32x4 *>(s_wv + o * kC + 4 * pc) = *reinterpret_cast<const f32x4 *>(a.wv + o * kC + 4 * c);
    }
    __syncthreads();

    // per-lane LDS offsets (floats).  Slice chunk i of a weight row of this lane's head / of a staging row:
    int w_off[4], r_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w_off[i] = 16 * h * kC + 4 * ((4 * sub + i) ^ (2 * (h & 1) + (sub >> 2)));
        r_off[i] = 4 * ((4 * sub + i) ^ (sub >> 2));
    }
    // this lane's own channel pair inside a staging row (writer side), and the head's 16 q values (reader side)
    const int st_off = 4 * ((lane >> 1) ^ (lane >> 5)) + 2 * (lane & 1);
    int q_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q_off[i] = 4 * ((4 * h + i) ^ (h >> 2));

    // contiguous chunk of work items per workgroup (XCD-contiguous through the block remap); a work item = one
    // destination (R == 1) or R consecutive modes of one CSR row
    const int64_t n_items = a.n_dst / R;
    const int groups = a.rep / R;                 // mode groups per CSR row (R > 1)
    const int64_t nb = gridDim.x;
    const int64_t lb = qcn_xcd_remap(blockIdx.x, gridDim.x);
    const int64_t per = (n_items + nb - 1) / nb;
    const int64_t begin = lb * per;
    const int64_t end_idx = begin + per < n_items ? begin + per : n_items;
    const float2 bias = ld2(a.bv, 0, lane);
    const bool b0 = (sub & 1) != 0, b1 = (sub & 2) != 0, b2 = (sub & 4) != 0;

    for (int64_t idx = begin + wave; idx < end_idx; idx += kW) {
        int64_t row = idx, node[R];
        if (R == 1) {
            node[0] = idx;
            if (a.layout == 1) { row = idx / a.rep; node[0] = (idx % a.rep) * (int64_t)a.n_rows + row; }
            else if (a.layout == 2) { row = idx / a.rep; }
        } else {
            row = idx / groups;
            const int m0 = (int)(idx - row * groups) * R;
#pragma unroll
            for (int r = 0; r < R; ++r)
                node[r] = a.layout == 1 ? (int64_t)(m0 + r) * a.n_rows + row : row * a.rep + m0 + r;
        }
        int src_off = -a.col_shift;
        if (a.row_index != nullptr) row = a.row_index[idx];
        if (a.src_offset != nullptr) src_off += a.src_offset[idx];
        const int beg = a.rowptr[row], end = a.rowptr[row + 1];
        float2 qv[R];
        float m[R], l[R];
        float2 accv[R], proj[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            qv[r] = ld2(a.q, node[r], lane);
            qv[r].x *= 0.25f * kLog2e;      // head_dim ** -0.5 (layers/attention_layer.py:38), logits kept in base 2
            qv[r].y *= 0.25f * kLog2e;
            m[r] = -INFINITY; l[r] = 0.f;
            accv[r] = make_float2(0.f, 0.f);
            proj[r] = make_float2(0.f, 0.f);
        }

        if (end > beg) {
            int src[kU], e[kU];
            EdgeRows<kU> cur, nxt;
            load_indices<HAS_EID, kU>(a, beg, end, src_off, src, e);
            load_rows(a, src, e, lane, cur);
            load_indices<HAS_EID, kU>(a, beg + kU, end, src_off, src, e);

            // ---- prologue: utilde^h[c] = sum_d wk[16h + d][c] * q'^h[d] for the lane's 16-channel slice
            f32x2 u[R][8], s[R][8];   // the slice as channel pairs: packed fp32 FMAs (two channels per instruction)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                *reinterpret_cast<float2 *>(s_stage + st_off) = qv[r];
                wave_lds_fence();
                float qh[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4 *>(s_stage + q_off[i]);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) qh[4 * i + jj] = t4[jj];
                }
                wave_lds_fence();
#pragma unroll
                for (int c = 0; c < 8; ++c) { u[r][c] = f32x2{0.f, 0.f}; s[r][c] = f32x2{0.f, 0.f}; }
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const f32x2 qd = {qh[d], qh[d]};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(s_wk + w_off[i] + d * kC);
                        u[r][2 * i] = __builtin_elementwise_fma(f32x2{w4[0], w4[1]}, qd, u[r][2 * i]);
                        u[r][2 * i + 1] = __builtin_elementwise_fma(f32x2{w4[2], w4[3]}, qd, u[r][2 * i + 1]);
                    }
                }
            }

            // ---- edge loop
            for (int j = beg; j < end; j += kU) {
#pragma unroll
                for (int t = 0; t < kU; ++t) *reinterpret_cast<float2 *>(s_stage + t * kC + st_off) = cur.r[t];
                load_rows(a, src, e, lane, nxt);                                    // rows of iteration j + kU
                load_indices<HAS_EID, kU>(a, j + 2 * kU, end, src_off, src, e);     // indices of iteration j + 2 kU
                wave_lds_fence();
#pragma unroll
                for (int sb = 0; sb < kU; sb += kSub) {
                    f32x2 r16[kSub][8];
#pragma unroll
                    for (int t = 0; t < kSub; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 t4 = *reinterpret_cast<const f32x4 *>(s_stage + (sb + t) * kC + r_off[i]);
                            r16[t][2 * i] = f32x2{t4[0], t4[1]};
                            r16[t][2 * i + 1] = f32x2{t4[2], t4[3]};
                        }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float sim[kSub];
                        float smax = -INFINITY;
#pragma unroll
                        for (int t = 0; t < kSub; ++t) {
                            f32x2 pa = f32x2{qv[r].x, qv[r].y} * f32x2{cur.k[sb + t].x, cur.k[sb + t].y}, pb = f32x2{0.f, 0.f};
#pragma unroll
                            for (int c = 0; c < 8; c += 2) {
                                pa = __builtin_elementwise_fma(u[r][c], r16[t][c], pa);
                                pb = __builtin_elementwise_fma(u[r][c + 1], r16[t][c + 1], pb);
                            }
                            pa += pb;
                            const float part = head_sum(pa[0] + pa[1]);
                            sim[t] = (j + sb + t < end) ? part : -INFINITY;
                            smax = fmaxf(smax, sim[t]);
                        }
                        const float mn = fmaxf(m[r], smax);
                        if (!__all(mn == m[r])) {                      // running max moved for some head: rescale
                            const float alpha = __builtin_amdgcn_exp2f(m[r] - mn);
                            l[r] *= alpha; accv[r].x *= alpha; accv[r].y *= alpha;
#pragma unroll
                            for (int c = 0; c < 8; ++c) s[r][c] *= alpha;
                            m[r] = mn;
                        }
#pragma unroll
                        for (int t = 0; t < kSub; ++t) {
                            const float pe = __builtin_amdgcn_exp2f(sim[t] - m[r]);
                            l[r] += pe;
                            accv[r].x = fmaf(pe, cur.v[sb + t].x, accv[r].x);
                            accv[r].y = fmaf(pe, cur.v[sb + t].y, accv[r].y);
                            const f32x2 pe2 = {pe, pe};
#pragma unroll
                            for (int c = 0; c < 8; ++c) s[r][c] = __builtin_elementwise_fma(pe2, r16[t][c], s[r][c]);
                        }
                    }
                }
                wave_lds_fence();                                  // staging rows are free again
                cur = nxt;
            }

            // ---- epilogue: out^h[d] = sum_c wv[16h + d][c] * s^h[c]: in-lane partial over the slice, then a
            // reduce-scatter over the 8 lanes of the head that leaves d = 2 sub, 2 sub + 1 on lane sub
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float part[16];
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(s_wv + w_off[i] + d * kC);
                        p0 = __builtin_elementwise_fma(f32x2{w4[0], w4[1]}, s[r][2 * i], p0);
                        p1 = __builtin_elementwise_fma(f32x2{w4[2], w4[3]}, s[r][2 * i + 1], p1);
                    }
                    p0 += p1;
                    part[d] = p0[0] + p0[1];
                }
                float r1[8], r2[4];
#pragma unroll
                for (int d = 0; d < 8; ++d) {       // partner 7 - sub (opposite bit 2): keep d with bit 3 == own bit 2
                    const float keep = b2 ? part[d + 8] : part[d], give = b2 ? part[d] : part[d + 8];
                    r1[d] = keep + dpp_mov<kDppMirror8>(give);
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {       // partner sub ^ 2: keep d with bit 2 == own bit 1
                    const float keep = b1 ? r1[d + 4] : r1[d], give = b1 ? r1[d] : r1[d + 4];
                    r2[d] = keep + dpp_mov<kDppXor2>(give);
                }
                {                                   // partner sub ^ 1: keep d with bit 1 == own bit 0
                    const float k0 = b0 ? r2[2] : r2[0], g0 = b0 ? r2[0] : r2[2];
                    const float k1 = b0 ? r2[3] : r2[1], g1 = b0 ? r2[1] : r2[3];
                    proj[r].x = k0 + dpp_mov<kDppXor1>(g0);
                    proj[r].y = k1 + dpp_mov<kDppXor1>(g1);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float inv = 1.0f / (l[r] + 1e-16f);     // PyG softmax denominator
            const float asum = l[r] * inv;                // A_i^h: 1 for non-empty segments, 0 for empty ones
            float2 o2;
            o2.x = fmaf(inv, proj[r].x + accv[r].x, asum * bias.x);
            o2.y = fmaf(inv, proj[r].y + accv[r].y, asum * bias.y);
            *reinterpret_cast<float2 *>(a.out + node[r] * kC + 2 * lane) = o2;
        }
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
                                  int64_t num_edges, int rep, int layout, int64_t col_shift,
                                  const int32_t *row_index, const int32_t *src_offset, float *out, void *stream)
{
    if (n_dst < 0 || n_src < 0 || n_rows < 0 || num_edges < 0 || rep < 1) return QCN_ERR_INVALID_ARGUMENT;
    if (layout < 0 || layout > 2) return QCN_ERR_INVALID_ARGUMENT;
    if (row_index != nullptr || src_offset != nullptr) {
        if (layout != 0 || rep != 1) return QCN_ERR_INVALID_ARGUMENT;
        if (row_index == nullptr && n_rows != n_dst) return QCN_ERR_INVALID_ARGUMENT;
    } else if (layout == 0 ? (rep != 1 || n_rows != n_dst) : n_rows * rep != n_dst) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst == 0) return QCN_OK;
    if (!q || !out || !rowptr || !wk || !wv || !bv) return QCN_ERR_INVALID_ARGUMENT;
    if (num_edges > 0 && (!k || !v || !col || !rhat)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst >= ((int64_t)1 << 31) || n_src >= ((int64_t)1 << 31) || num_edges >= ((int64_t)1 << 31))
        return QCN_ERR_UNSUPPORTED;
    if (col_shift < 0 || col_shift >= ((int64_t)1 << 31)) return QCN_ERR_INVALID_ARGUMENT;
    FusedArgs args{q, k, v, rhat, rowptr, col, eid, wk, wv, bv, out, n_dst, (int)n_rows, rep, layout, (int)col_shift,
                   row_index, src_offset};
    // one workgroup per CU (136-144 KiB of LDS); every workgroup gets a contiguous chunk of work items.  Mode-replicated
    // graphs carry 3 (or 2) modes of a row per wave.
    int R = rep % 3 == 0 ? 3 : (rep % 2 == 0 ? 2 : 1);
    if (const char *env = qcn_knob("QCN_ATTN_R")) { const int want = atoi(env); if ((want == 1 || want == 2 || want == 3 || want == 6) && rep % want == 0) R = want; }
    const int waves = R == 1 ? kWaves : (R <= 3 ? kWaves / 2 : kWaves / 4);
    int64_t grid = (n_dst / R + waves - 1) / waves;
    if (grid > qcn_num_cus()) grid = qcn_num_cus();
    hipStream_t s = qcn_stream(stream);
#define QCN_LAUNCH(E, RR) hipLaunchKernelGGL((attn_fused_kernel<E, RR>), dim3((unsigned)grid), dim3(64 * (RR == 1 ? kWaves : (RR <= 3 ? kWaves / 2 : kWaves / 4))), 0, s, args)
    if (eid) { if (R == 6) QCN_LAUNCH(true, 6); else if (R == 3) QCN_LAUNCH(true, 3); else if (R == 2) QCN_LAUNCH(true, 2); else QCN_LAUNCH(true, 1); }
    else     { if (R == 6) QCN_LAUNCH(false, 6); else if (R == 3) QCN_LAUNCH(false, 3); else if (R == 2) QCN_LAUNCH(false, 2); else QCN_LAUNCH(false, 1); }
#undef QCN_LAUNCH
    const bool ok = hipGetLastError() == hipSuccess;
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}
