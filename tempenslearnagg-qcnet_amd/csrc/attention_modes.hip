// A3 (mode-replicated relations on the matrix cores) — the decoder's time->mode, polygon->mode and agent->mode attention
// (modules/qcnet_decoder.py:190-226,259-263): the K mode queries of an agent share ONE edge list, one positional
// embedding r_e per edge and the same source rows, so everything that depends on the edge alone is computed once per
// (agent, edge) and serves all K modes:
//   kk_e = k[j] + W_kr LN_r(r_e)            vv_e = v[j] + W_vr LN_r(r_e) (+ b_vr)      (layers/attention_layer.py:94-101)
// are 128x128 products per edge — dense MFMA work — and the per-(mode, head) part left over is tiny:
//   logit[mode, e] = q[mode, head] . kk_e[head]   (16-dim),    out[mode, head] = sum_e softmax_e(logit) vv_e[head].
//
// Mapping: one workgroup (8 waves) per agent at a time, wave w = head w.  Edges are taken 16 at a time.  Per tile:
//   * the 16 rhat rows (split f16: value * 2^11 = hi + lo, 512 B per row, see qcn_split_rows_f16) are copied
//     global -> LDS once per workgroup by LDS-DMA (wave w copies 1 KiB = rows 2w, 2w+1) into a 16-B-chunk image with chunk c
//     of row e at c ^ e: the operand read of a k-step (lane (e, g) takes chunk 4 ks + g) is then conflict-free;
//   * kk^T[d, e] = W_kr[16w + d, :] . rhat_e   and   vv[e, d] = rhat_e . W_vr[16w + d, :]   are 4 k-steps of
//     v_mfma_f32_16x16x32_f16, three per k-step and product (hi hi + hi lo + lo hi, fp32 accumulate: the split-f16
//     arithmetic of the chain kernels).  The head's weight fragments (hi and lo, 64 VGPRs) stay in registers for the whole
//     launch; the rhat fragment of a lane is the B operand of the first product and the A operand of the second;
//   * k[j] / v[j] are added straight from global memory in the accumulator layouts (16 B resp. 4 x 4 B per lane);
//   * logits P[mode, e] = Q_w[mode, :] . kk^T[:, e] and O[mode, d] += alpha[mode, :] . vv[:, d] are four exact-fp32
//     v_mfma_f32_16x16x4_f32 each, fed by the accumulator registers of the projections (k index = the lane group, so
//     no data movement); the softmax runs in the accumulator layout (row max = a DPP reduction over the 16 lanes of
//     a row group), and alpha goes through one 16x16 LDS transposition per tile to become an A operand.
// Tiles are software-pipelined across agents: indices two tiles ahead, rows + DMA one tile ahead, one barrier per tile.
// Algorithmic HBM bytes per launch: 512 E_base (rhat) + 8 E_base (col, eid) + gathers of k / v (cache served) + 1 KiB
// per destination (q in, out) — the same figure as qcn_attn_fused_f32.
#include "common.hip.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int kC = QCN_CHANNELS;       // 128
constexpr int kHeads = 8;              // waves per workgroup
constexpr int kTile = 16;              // edges per tile
constexpr int kRowBytes = 512;         // one split-f16 row: 128 hi + 128 lo
constexpr int kTStride = 20;           // floats per row of the per-wave transposition tile
constexpr float kLog2e = 1.4426950408889634f;

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// reductions over the 16 lanes of a DPP row (all 16 lanes get the result)
__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));   // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));   // row_mirror
    return v;
}

__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct ModesArgs {
    const float *q, *k, *v;
    const _Float16 *rs;            // [rows, 256] split rhat
    const int32_t *rowptr, *col, *eid;
    const half8 *wfrag;            // [head][k|v][ks][hi|lo][lane] fragments, then {usc_k, usc_v}
    const float *bv;
    float *out;
    int n_rows, rep, layout, col_shift;
};

// one 16-edge tile of one CSR row: edges [j0, j0 + 16) clipped to `end`
struct Tile {
    int row, j0, end;              // row < 0: no tile
    bool first;                    // first tile of its row
};

struct TileIdx {                   // per-lane indices of a tile (clamped to valid edges)
    int rid;                       // rhat row of the edge this lane copies by DMA
    int cole;                      // source row of edge (lane & 15)
};

template <bool HAS_EID>
__device__ __forceinline__ TileIdx load_tile_idx(const ModesArgs &a, const Tile &t, int wave, int lane)
{
    TileIdx x;
    const int last = t.end - 1;
    const int jd = min(t.j0 + 2 * wave + (lane >> 5), last);
    x.rid = HAS_EID ? a.eid[jd] : jd;
    x.cole = a.col[min(t.j0 + (lane & 15), last)] - a.col_shift;
    return x;
}

struct TileRows {
    f32x4 kf;                      // k[col_e][16 w + 4 g .. + 3]
    float vf[4];                   // v[col_{4 g + r}][16 w + i]
    f32x4 qf;                      // q[node(row, mode = i)][16 w + 4 g .. + 3] (lanes with i < rep)
};

__device__ __forceinline__ int64_t node_of(const ModesArgs &a, int row, int mode)
{
    return a.layout == 1 ? (int64_t)mode * a.n_rows + row : (int64_t)row * a.rep + mode;
}

__device__ __forceinline__ TileRows load_tile_rows(const ModesArgs &a, const TileIdx &x, int row, int wave, int lane)
{
    TileRows r;
    const int i = lane & 15, g = lane >> 4;
    r.kf = *reinterpret_cast<const f32x4 *>(a.k + (int64_t)x.cole * kC + 16 * wave + 4 * g);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int cv = __shfl(x.cole, 4 * g + t);       // source row of edge 4 g + t (held by lane 4 g + t)
        r.vf[t] = a.v[(int64_t)cv * kC + 16 * wave + i];
    }
    r.qf = *reinterpret_cast<const f32x4 *>(a.q + node_of(a, row, min(i, a.rep - 1)) * kC + 16 * wave + 4 * g);
    return r;
}

// wave w copies LDS chunks [64 w, 64 w + 64) of the tile image: row e = 2 w + (lane >> 5), physical chunk lane & 31
__device__ __forceinline__ void issue_tile_dma(const ModesArgs &a, const TileIdx &x, unsigned char *buf, int wave, int lane)
{
    const int e = 2 * wave + (lane >> 5), pc = lane & 31;
    const int c = (pc & 15) ^ e;                       // logical 16-B chunk inside the hi (pc < 16) or lo half
    const _Float16 *src = a.rs + (int64_t)x.rid * 256 + (pc >> 4) * 128 + c * 8;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(buf + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// rowptr is read through the scalar unit (constant address space: it is not written during the launch, which the
// compiler cannot see through the struct argument) — a vector load here would put a vmcnt(0) into the tile pipeline
__device__ __forceinline__ int rowptr_at(const ModesArgs &a, int row)
{
    using cint = const __attribute__((address_space(4))) int32_t;
    return *((cint *)(uintptr_t)a.rowptr + row);
}

// first tile of the first non-empty row in [row, row_end)
__device__ __forceinline__ Tile first_tile(const ModesArgs &a, int row, int row_end)
{
    while (row < row_end) {
        const int b = rowptr_at(a, row), e = rowptr_at(a, row + 1);
        if (e > b) return Tile{row, b, e, true};
        ++row;
    }
    return Tile{-1, 0, 0, false};
}

__device__ __forceinline__ Tile next_tile(const ModesArgs &a, Tile t, int row_end)
{
    if (t.row < 0) return t;
    if (t.j0 + kTile < t.end) { t.j0 += kTile; t.first = false; return t; }
    return first_tile(a, t.row + 1, row_end);
}

// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt = imm[15:14]:imm[3:0]; expcnt / lgkmcnt fields left at "no wait")
constexpr int vmcnt_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }

template <int S> struct Phase {};

// kDepth tiles are in flight behind the one being computed: the tile computed in iteration n had its rhat DMA and its
// k / v / q row loads issued in iteration n - kDepth, with indices loaded in iteration n - 2 kDepth.  Ring slots
// (tile n -> slot n % kBufs) are compile-time constants in the kBufs-times unrolled loop, and every LDS image is a
// __shared__ variable of its own, so that neither register copies nor the compiler's LDS-DMA alias waits drain the
// queue early.  Every iteration issues the same kOps vector-memory loads (clamped duplicates past the last tile); the
// one manual wait per iteration, vmcnt((kDepth - 1) kOps), retires the DMA of the NEXT tile and leaves the younger
// iterations' loads in flight.
constexpr int kDepth = 2;
constexpr int kBufs = kDepth + 1;

template <bool HAS_EID>
__global__ __launch_bounds__(64 * kHeads) void attn_modes_kernel(ModesArgs a)
{
    static_assert(kBufs == 3, "one __shared__ image per ring slot below");
    __shared__ __attribute__((aligned(16))) unsigned char s_r0[kTile * kRowBytes];       // rhat tile images
    __shared__ __attribute__((aligned(16))) unsigned char s_r1[kTile * kRowBytes];
    __shared__ __attribute__((aligned(16))) unsigned char s_r2[kTile * kRowBytes];
    __shared__ __attribute__((aligned(16))) float s_t[kHeads][16 * kTStride];            // alpha transposition
    constexpr int kOps = (HAS_EID ? 2 : 1) + 1 + 6;      // index loads + DMA + k, 4 v, q row loads per iteration

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;

    // contiguous chunk of CSR rows per workgroup (XCD-contiguous through the block remap)
    const int nb = gridDim.x;
    const int lb = qcn_xcd_remap(blockIdx.x, gridDim.x);
    const int per = (a.n_rows + nb - 1) / nb;
    const int row_begin = min(lb * per, a.n_rows), row_end = min(row_begin + per, a.n_rows);

    // the head's weight fragments: frag[lane][j] = W[16 w + (lane & 15)][32 ks + 8 (lane >> 4) + j]
    half8 wk_hi[4], wk_lo[4], wv_hi[4], wv_lo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const half8 *base = a.wfrag + (size_t)((wave * 2) * 4 + ks) * 2 * 64 + lane;
        wk_hi[ks] = base[0];
        wk_lo[ks] = base[64];
        wv_hi[ks] = base[4 * 2 * 64];
        wv_lo[ks] = base[4 * 2 * 64 + 64];
    }
    const float *scales = reinterpret_cast<const float *>(a.wfrag + (size_t)kHeads * 2 * 4 * 2 * 64);
    const float usc_k = scales[0], usc_v = scales[1];
    const float bias = a.bv[16 * wave + i];
    float *const s_tw = s_t[wave];

    // rows without edges produce zeros (PyG softmax over an empty segment, attention_layer.py:118)
    for (int row = row_begin; row < row_end; ++row) {
        if (rowptr_at(a, row + 1) > rowptr_at(a, row)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mode = 4 * g + r;
            if (mode < a.rep) a.out[node_of(a, row, mode) * kC + 16 * wave + i] = 0.f;
        }
    }

    Tile tl[2 * kDepth + 1];
    tl[0] = first_tile(a, row_begin, row_end);
    if (tl[0].row < 0) return;
#pragma unroll
    for (int j = 1; j <= 2 * kDepth; ++j) tl[j] = next_tile(a, tl[j - 1], row_end);

    TileRows rows[kBufs];
    TileIdx idx[kBufs];
    auto image = [&](auto ph) -> unsigned char * {
        constexpr int S = decltype(ph)::value;
        return S == 0 ? s_r0 : (S == 1 ? s_r1 : s_r2);
    };
    auto safe = [&](const Tile &t) -> Tile { return t.row >= 0 ? t : tl[0]; };
    // prologue: tiles 0 .. kDepth - 1 completely, indices of tiles kDepth .. 2 kDepth - 1
    {
        const Tile ta = safe(tl[0]), tb = safe(tl[1]);
        const TileIdx xa = load_tile_idx<HAS_EID>(a, ta, wave, lane), xb = load_tile_idx<HAS_EID>(a, tb, wave, lane);
        issue_tile_dma(a, xa, s_r0, wave, lane);
        rows[0] = load_tile_rows(a, xa, ta.row, wave, lane);
        issue_tile_dma(a, xb, s_r1, wave, lane);
        rows[1] = load_tile_rows(a, xb, tb.row, wave, lane);
        idx[2] = load_tile_idx<HAS_EID>(a, safe(tl[2]), wave, lane);
        idx[0] = load_tile_idx<HAS_EID>(a, safe(tl[3]), wave, lane);
        rows[2] = rows[1];
        idx[1] = idx[0];
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    __syncthreads();

    f32x4 qa = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    float m[4], l[4];

    // one iteration with the current tile in ring slot S; returns true after the last tile
    auto step = [&](auto ph) -> bool {
        constexpr int S = decltype(ph)::value;
        constexpr int SB = (S + kDepth) % kBufs;          // slot of the tile kDepth ahead (DMA + rows now)
        constexpr int SC = (S + 2 * kDepth) % kBufs;      // slot of the indices loaded now
        const Tile t0 = tl[0], tb = safe(tl[kDepth]), tc = safe(tl[2 * kDepth]);
        const Tile tn = next_tile(a, tl[2 * kDepth], row_end);
        issue_tile_dma(a, idx[SB], image(std::integral_constant<int, SB>{}), wave, lane);
        rows[SB] = load_tile_rows(a, idx[SB], tb.row, wave, lane);
        idx[SC] = load_tile_idx<HAS_EID>(a, tc, wave, lane);

        // ---- the current tile
        if (t0.first) {
            qa = rows[S].qf * (0.25f * kLog2e);   // head_dim ** -0.5 (layers/attention_layer.py:38), logits in base 2
            if (i >= a.rep) qa = f32x4{0.f, 0.f, 0.f, 0.f};
            o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
        }
        f32x4 kk = {0.f, 0.f, 0.f, 0.f}, kx = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f}, vx = {0.f, 0.f, 0.f, 0.f};
        const unsigned char *img = image(ph) + i * kRowBytes;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = 16 * ((4 * ks + g) ^ i);
            const half8 rh = *reinterpret_cast<const half8 *>(img + off);
            const half8 rl = *reinterpret_cast<const half8 *>(img + 256 + off);
            kk = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk_hi[ks], rh, kk, 0, 0, 0);
            vv = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh, wv_hi[ks], vv, 0, 0, 0);
            kx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk_hi[ks], rl, kx, 0, 0, 0);
            vx = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl, wv_hi[ks], vx, 0, 0, 0);
            kx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk_lo[ks], rh, kx, 0, 0, 0);
            vx = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh, wv_lo[ks], vx, 0, 0, 0);
        }
        // kk^T[d = 4 g + r][e = i], vv[e = 4 g + r][d = i]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            kk[r] = fmaf(kk[r] + kx[r], usc_k, rows[S].kf[r]);
            vv[r] = fmaf(vv[r] + vx[r], usc_v, rows[S].vf[r]);
        }
        f32x4 p = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
        p = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0], kk[0], p, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[1], kk[1], p2, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[2], kk[2], p, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[3], kk[3], p2, 0, 0, 0);
        p += p2;
        // p[r] = logit[mode = 4 g + r][e = i]
        const bool valid = t0.j0 + i < t0.end;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = valid ? p[r] : -INFINITY;
            const float mn = fmaxf(m[r], row16_max(pv));
            const float corr = __builtin_amdgcn_exp2f(m[r] - mn);
            m[r] = mn;
            p[r] = __builtin_amdgcn_exp2f(pv - mn);
            l[r] = fmaf(l[r], corr, p[r]);
            o[r] *= corr;
            s_tw[(4 * g + r) * kTStride + i] = p[r];
        }
        wave_lds_fence();
        const f32x4 pa = *reinterpret_cast<const f32x4 *>(s_tw + i * kTStride + 4 * g);   // alpha[mode = i][e = 4 g + r]
        wave_lds_fence();
        f32x4 o2 = {0.f, 0.f, 0.f, 0.f};
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[0], vv[0], o, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[1], vv[1], o2, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[2], vv[2], o, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[3], vv[3], o2, 0, 0, 0);
        o += o2;

        if (tl[1].row != t0.row) {                              // last tile of its row
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ls = row16_sum(l[r]);
                const float inv = 1.0f / (ls + 1e-16f);         // PyG softmax denominator
                const int mode = 4 * g + r;
                if (mode < a.rep) a.out[node_of(a, t0.row, mode) * kC + 16 * wave + i] = fmaf(o[r], inv, ls * inv * bias);
            }
        }
        if (tl[1].row < 0) return true;
        // the next tile's DMA has landed (vmcnt) and this wave's LDS reads are done (lgkmcnt 0); a raw barrier:
        // __syncthreads() would add a release fence, i.e. vmcnt(0)
        __builtin_amdgcn_s_waitcnt(vmcnt_imm((kDepth - 1) * kOps) & ~0x0F00);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < 2 * kDepth; ++j) tl[j] = tl[j + 1];
        tl[2 * kDepth] = tn;
        return false;
    };
    while (true) {
        if (step(std::integral_constant<int, 0>{})) break;
        if (step(std::integral_constant<int, 1>{})) break;
        if (step(std::integral_constant<int, 2>{})) break;
    }
}

// fp32 rows -> split-f16 rows: y[row] = [hi(x * 2^scale_log2) (128) | lo (128)], hi + lo = x * 2^scale_log2 to 22 bits
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ x, _Float16 *__restrict__ y,
                                                         int64_t n_rows, float scale)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float2 v = *reinterpret_cast<const float2 *>(x + row * kC + 2 * lane);
    const float a = v.x * scale, b = v.y * scale;
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    using half2v = __attribute__((ext_vector_type(2))) _Float16;
    *reinterpret_cast<half2v *>(y + row * 256 + 2 * lane) = half2v{ha, hb};
    *reinterpret_cast<half2v *>(y + row * 256 + 128 + 2 * lane) = half2v{la, lb};
}

}  // namespace

extern "C" int qcn_split_rows_f16(const float *x, int64_t n_rows, int scale_log2, void *y, void *stream)
{
    if (n_rows < 0 || scale_log2 < -64 || scale_log2 > 64) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows == 0) return QCN_OK;
    if (!x || !y) return QCN_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, qcn_stream(stream), x,
                       reinterpret_cast<_Float16 *>(y), n_rows, ldexpf(1.0f, scale_log2));
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

extern "C" size_t qcn_attn_modes_pack_bytes(void) { return (size_t)kHeads * 2 * 4 * 2 * 64 * 16 + 2 * sizeof(float); }

extern "C" int qcn_attn_modes_f32(const float *q, const float *k, const float *v, const void *rhat_split,
                                  const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                                  const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                                  int rep, int layout, int64_t col_shift, float *out, void *stream)
{
    if (n_dst < 0 || n_src < 0 || n_rows < 0 || num_edges < 0 || rep < 1 || rep > 6) return QCN_ERR_INVALID_ARGUMENT;
    if (layout != 1 && layout != 2) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows * rep != n_dst) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst == 0) return QCN_OK;
    if (!q || !out || !rowptr || !wpack || !bv) return QCN_ERR_INVALID_ARGUMENT;
    if (num_edges > 0 && (!k || !v || !col || !rhat_split)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst >= ((int64_t)1 << 31) || n_src >= ((int64_t)1 << 31) || num_edges >= ((int64_t)1 << 31))
        return QCN_ERR_UNSUPPORTED;
    if (col_shift < 0 || col_shift >= ((int64_t)1 << 31)) return QCN_ERR_INVALID_ARGUMENT;
    ModesArgs args{q, k, v, reinterpret_cast<const _Float16 *>(rhat_split), rowptr, col, eid,
                   reinterpret_cast<const half8 *>(wpack), bv, out, (int)n_rows, rep, layout, (int)col_shift};
    int per_cu = 2;
    if (const char *env = getenv("QCN_MODES_WG_PER_CU")) { const int want = atoi(env); if (want >= 1 && want <= 8) per_cu = want; }
    int64_t grid = n_rows < 256 * per_cu ? n_rows : 256 * per_cu;
    hipStream_t s = qcn_stream(stream);
    void *token = qcn_prof_begin(QCN_PROF_ATTN_FUSED, 1552.0 * (double)num_edges * (double)rep + 1024.0 * (double)n_dst, s);
    if (eid) hipLaunchKernelGGL((attn_modes_kernel<true>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
    else     hipLaunchKernelGGL((attn_modes_kernel<false>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
    const bool ok = hipGetLastError() == hipSuccess;
    qcn_prof_end(token, s);
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}
