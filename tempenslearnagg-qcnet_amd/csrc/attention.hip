// A3 — fused per-destination attention: gather k/v rows of the sources (+ per-edge positional terms),
// wavefront segment softmax, weighted accumulation of V, one coalesced store per destination row.
//
// Mapping (C = 128 = 8 heads x 16): one wave64 per destination node, lane l owns channels 2l, 2l+1, so a
// head is 8 consecutive lanes and every row access is one 512-B coalesced wave instruction
// (global_load_dwordx2).  Edges of a destination are contiguous in the dst-sorted CSR; they are consumed four
// at a time so up to 16 row loads are in flight per wave, and the softmax is the online (running max / running
// sum) form, so every input byte is read exactly once and nothing but the [n_dst,128] result is written.
// HBM-bound: 2064 B/edge + 1024 B/destination with positional terms.
#include "common.hip.h"

namespace {

constexpr int kC = QCN_CHANNELS;
constexpr int kUnroll = 4;

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
    return v + __builtin_bit_cast(float, moved);
}

// Sum over the 8 lanes of a head: xor-1 and xor-2 inside a quad, then mirror the two quads of the half-row.
__device__ __forceinline__ float head_sum(float v)
{
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    return v;
}

__device__ __forceinline__ float2 ld2(const float *__restrict__ base, int64_t row, int lane)
{
    return *reinterpret_cast<const float2 *>(base + row * kC + 2 * lane);
}

template <bool HAS_R, bool HAS_EID>
__global__ __launch_bounds__(256) void attn_fwd_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ kr, const float *__restrict__ vr, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, const int32_t *__restrict__ eid, int64_t n_dst, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t node = (int64_t)qcn_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (node >= n_dst) return;
    const int beg = rowptr[node], end = rowptr[node + 1];
    const float2 qv = ld2(q, node, lane);
    const float scale = 0.25f;  // head_dim ** -0.5, layers/attention_layer.py:38

    float m = -INFINITY, l = 0.f;
    float2 acc = make_float2(0.f, 0.f);
    for (int j = beg; j < end; j += kUnroll) {
        float2 kk[kUnroll], vv[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            // Clamp instead of branching: every load is unconditional (no per-load waits), the padded
            // slots re-read the last edge's rows from L1 and are masked out of the softmax below.
            const int jj = min(j + u, end - 1);
            const int s = col[jj];
            kk[u] = ld2(k, s, lane);
            vv[u] = ld2(v, s, lane);
            if (HAS_R) {
                const int e = HAS_EID ? eid[jj] : jj;
                const float2 a = ld2(kr, e, lane), b = ld2(vr, e, lane);
                kk[u].x += a.x; kk[u].y += a.y;
                vv[u].x += b.x; vv[u].y += b.y;
            }
        }
        float sim[kUnroll];
        float smax = -INFINITY;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const float part = head_sum(fmaf(qv.y, kk[u].y, qv.x * kk[u].x));
            sim[u] = (j + u < end) ? part * scale : -INFINITY;
            smax = fmaxf(smax, sim[u]);
        }
        const float mn = fmaxf(m, smax);
        const float alpha = expf(m - mn);
        float psum = 0.f;
        acc.x *= alpha; acc.y *= alpha;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const float p = expf(sim[u] - mn);
            psum += p;
            acc.x = fmaf(p, vv[u].x, acc.x);
            acc.y = fmaf(p, vv[u].y, acc.y);
        }
        l = fmaf(l, alpha, psum);
        m = mn;
    }
    const float inv = 1.0f / (l + 1e-16f);  // PyG softmax denominator, torch_geometric.utils.softmax
    *reinterpret_cast<float2 *>(out + node * kC + 2 * lane) = make_float2(acc.x * inv, acc.y * inv);
}

}  // namespace

extern "C" int qcn_attn_fwd_f32(const float *q, const float *k, const float *v, const float *kr, const float *vr,
                                const int32_t *rowptr, const int32_t *col, const int32_t *eid, int64_t n_dst,
                                int64_t n_src, int64_t num_edges, float *out, void *stream)
{
    if (n_dst < 0 || n_src < 0 || num_edges < 0) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst == 0) return QCN_OK;
    if (!q || !out || !rowptr) return QCN_ERR_INVALID_ARGUMENT;
    if (num_edges > 0 && (!k || !v || !col)) return QCN_ERR_INVALID_ARGUMENT;
    if ((kr == nullptr) != (vr == nullptr)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst >= ((int64_t)1 << 31) || n_src >= ((int64_t)1 << 31) || num_edges >= ((int64_t)1 << 31))
        return QCN_ERR_UNSUPPORTED;
    hipStream_t s = qcn_stream(stream);
    const unsigned grid = (unsigned)((n_dst + 3) / 4);
    const bool has_r = kr != nullptr && num_edges > 0;

    if (has_r) {
        if (eid)
            hipLaunchKernelGGL((attn_fwd_kernel<true, true>), dim3(grid), dim3(256), 0, s, q, k, v, kr, vr, rowptr, col,
                               eid, n_dst, out);
        else
            hipLaunchKernelGGL((attn_fwd_kernel<true, false>), dim3(grid), dim3(256), 0, s, q, k, v, kr, vr, rowptr,
                               col, eid, n_dst, out);
    } else {
        hipLaunchKernelGGL((attn_fwd_kernel<false, false>), dim3(grid), dim3(256), 0, s, q, k, v, kr, vr, rowptr, col,
                           eid, n_dst, out);
    }
    const bool ok = hipGetLastError() == hipSuccess;
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}
