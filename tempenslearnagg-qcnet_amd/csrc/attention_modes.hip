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
#include <cstdio>
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
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct ModesArgs {
    const float *q, *k, *v;
    const void *rs;                // [rows, 256] split-f16 rhat (512 B per row) or [rows, 128] bf16 rhat (256 B per row)
    const int32_t *rowptr, *col, *eid;
    const half8 *wfrag;            // [head][k|v][ks][hi|lo][lane] fragments, then {usc_k, usc_v}
    const float *bv;
    float *out;
    int n_rows, rep, layout, col_shift;
    unsigned long long *stamps;    // QCN_MODES_STAMPS builds only
    // per-edge projections kk_e / vv_e, written by the STORE instantiations of attn_modes_kernel, read by attn_kv_kernel
    // (qcn_attn_modes_store_* / qcn_attn_modes_cached_f32).  HEAD-MAJOR tables of `table_slots` slots per head plane
    // (include/qcnet_hip.h): edge j of CSR row r sits in slot s = ((rowptr[r] + 3 r + 3) & ~3) + (j - rowptr[r]) — every
    // row starts on a multiple of four slots, in closed form, no scan —
    //   kk[head][s][16]        : a wave's 16-edge tile is ONE contiguous 1 KiB (lane (e, g): 16 B at 64 s + 16 g)
    //   vv[head][s / 4][16][4] : groups of four slots transposed, so that the B operand of the weighted sum (lane (g, d):
    //                            channel d of edges 4 g .. 4 g + 3) is one 16-B load and the tile again 1 KiB contiguous.
    // (that is the QCN_KV_PLANES form; the shipped form interleaves the heads per group of four slots, see kk_slot_off.
    // Round 2 kept [slot][128] rows: 64-B fragments per (slot, head), six loads per tile and wave; 4.0 TB/s of tables.)
    float *kk, *vv;
    long long table_slots;
};

// one 16-edge tile of one CSR row: edges [j0, j0 + 16) clipped to `end`
// Table addressing (byte offsets inside a table; see ModesArgs::kk).  QCN_KV_PLANES=1 selects the head-plane form
// kk[head][s][16] / vv[head][s / 4][16][4] (a wave's tile = 1 KiB contiguous, but 16 distant streams per workgroup: A/B
// builds only); the default interleaves the heads inside every group of four slots,
//   kk[s / 4][head][s % 4][16]     vv[s / 4][head][16][s % 4],
// so that a workgroup's eight waves together read ONE contiguous 8 KiB per tile and table (row-buffer locality in HBM)
// while every wave instruction still covers whole 256-B pieces (two full 128-B lines per piece).
#ifndef QCN_KV_PLANES
#define QCN_KV_PLANES 0
#endif
__device__ __forceinline__ uint32_t kk_slot_off(uint32_t slot, int head, int g)
{
#if QCN_KV_PLANES
    (void)head;
    return slot * 64u + 16u * g;
#else
    return (slot >> 2) * 2048u + (uint32_t)head * 256u + (slot & 3u) * 64u + 16u * g;
#endif
}
__device__ __forceinline__ uint32_t vv_group_off(uint32_t group, int head, int d)
{
#if QCN_KV_PLANES
    (void)head;
    return group * 256u + 16u * d;
#else
    return group * 2048u + (uint32_t)head * 256u + 16u * d;
#endif
}
__device__ __forceinline__ size_t kv_plane_base(int head, long long table_slots)
{
#if QCN_KV_PLANES
    return (size_t)head * (size_t)table_slots * 64u;
#else
    (void)head; (void)table_slots;
    return 0;
#endif
}

struct Tile {
    int row, j0, end;              // row < 0: no tile
    bool first;                    // first tile of its row
    int s0;                        // table slot of edge j0 (see ModesArgs::kk): a multiple of four
};

struct TileIdx {                   // indices of the edge whose rows this lane copies by DMA (clamped to valid edges)
    int rid;                       // its rhat row
    int src;                       // its source row (k / v)
};

// all global addresses are base (uniform) + 32-bit byte offset: the launcher refuses tables of 4 GiB and more
template <typename T>
__device__ __forceinline__ T ld_off(const void *base, uint32_t byte_off)
{
    return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}

template <bool HAS_EID>
__device__ __forceinline__ TileIdx load_tile_idx(const ModesArgs &a, const Tile &t, int wave, int lane)
{
    TileIdx x;
    const int jd = min(t.j0 + 2 * wave + (lane >> 5), t.end - 1);
    x.rid = HAS_EID ? ld_off<int>(a.eid, 4u * (uint32_t)jd) : jd;
    x.src = ld_off<int>(a.col, 4u * (uint32_t)jd) - a.col_shift;
    return x;
}

// Byte offsets of destination rows: node(row, mode) * 512 = row_term(row) (uniform, scalar unit) + a per-lane constant of
// the mode, so the per-tile address work is one add.
struct NodeAddr {
    uint32_t q_lane;          // q: mode = min(lane & 15, rep - 1), + 64 w + 16 g
    uint32_t o_lane[4];       // out: mode = 4 g + r, + 64 w + 4 i
    uint32_t row_mul;         // row_term = row * row_mul
};

__device__ __forceinline__ NodeAddr make_node_addr(const ModesArgs &a, int wave, int lane)
{
    const int i = lane & 15, g = lane >> 4;
    const uint32_t mode_mul = a.layout == 1 ? (uint32_t)a.n_rows * 512u : 512u;
    NodeAddr n;
    n.row_mul = a.layout == 1 ? 512u : (uint32_t)a.rep * 512u;
    n.q_lane = (uint32_t)min(i, a.rep - 1) * mode_mul + 64u * wave + 16u * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) n.o_lane[r] = (uint32_t)(4 * g + r) * mode_mul + 64u * wave + 4u * i;
    return n;
}

// q[node(row, mode = i)][16 w + 4 g .. + 3] (lanes with i >= rep read mode rep - 1 and are zeroed at the use)
__device__ __forceinline__ f32x4 load_q(const ModesArgs &a, const NodeAddr &n, int row)
{
    return ld_off<f32x4>(a.q, (uint32_t)row * n.row_mul + n.q_lane);
}

// sum over the 16 lanes of every DPP row, four independent values at once (see row16_max4)
__device__ __forceinline__ void row16_sum4(float (&v)[4])
{
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// Row epilogue: out[node(row, mode = 4 g + r)][16 w + i] = o / (l + 1e-16) + (l / (l + 1e-16)) bias   (PyG softmax
// denominator; l, o the running sum / weighted sum of the row's softmax)
__device__ __forceinline__ void store_row_out(const ModesArgs &a, const NodeAddr &n, int row, int g, const float (&l)[4],
                                              const f32x4 &o, float bias)
{
    float ls[4] = {l[0], l[1], l[2], l[3]};
    row16_sum4(ls);
    const uint32_t row_term = (uint32_t)row * n.row_mul;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float den = ls[r] + 1e-16f;
        float inv = __builtin_amdgcn_rcpf(den);
        inv = fmaf(fmaf(-den, inv, 1.0f), inv, inv);    // one Newton step on the hardware reciprocal
        if (4 * g + r < a.rep)
            *reinterpret_cast<float *>(reinterpret_cast<char *>(a.out) + (row_term + n.o_lane[r])) = fmaf(o[r], inv, ls[r] * inv * bias);
    }
}

// One ring slot: the tile's 16 rhat rows (split f16), 16 k rows and 16 v rows (fp32), 512 B each, as 16-B chunks with
// chunk c of row e stored at c ^ e (rhat: inside its hi / lo half).  Wave w copies rows 2 w, 2 w + 1 of each image.
struct Slot {
    unsigned char r[kTile * kRowBytes], k[kTile * kRowBytes], v[kTile * kRowBytes];
};

// Cache policy of everything that streams through once per launch — the positional rows (LDS-DMA aux bit 1 = nt on gfx950),
// the per-edge tables written by the STORE launch and read by attn_kv_kernel: non-temporal, so that 0.5-2 GB per launch do
// not push the gathered k / v rows out of L2 / MALL.  Same box, pl2m relation of cfg3 (2.03 M edges): table attention 478 ->
// 433 us (4.36 -> 4.80 TB/s of tables), recomputing launch 858 -> 839 us, STORE launch 936 -> 923 us.  QCN_NO_NT: A/B builds.
#ifdef QCN_NO_NT
#define QCN_RS_AUX 0
#else
#define QCN_RS_AUX 2
#endif
template <bool BF16>
__device__ __forceinline__ void issue_tile_dma(const ModesArgs &a, const TileIdx &x, Slot *slot, int wave, int lane)
{
    const int e = 2 * wave + (lane >> 5), pc = lane & 31;
    const uint32_t ck = (uint32_t)(pc ^ e);                          // k / v: logical chunk
    const char *sk = reinterpret_cast<const char *>(a.k) + ((uint32_t)x.src * 512u + ck * 16u);
    const char *sv = reinterpret_cast<const char *>(a.v) + ((uint32_t)x.src * 512u + ck * 16u);
    if constexpr (BF16) {
        // a bf16 rhat row is 16 chunks: lanes 0..31 copy rows 2 w (lanes 0..15) and 2 w + 1 (lanes 16..31), 512 B per wave,
        // chunk c of row e at c ^ e; the upper half of the wave sits the instruction out (every wave still issues it once,
        // so the vmcnt bookkeeping of the pipeline is the same as in the f16 form)
        const int rid0 = __builtin_amdgcn_readlane(x.rid, 0), rid1 = __builtin_amdgcn_readlane(x.rid, 32);
        const int er = 2 * wave + ((lane >> 4) & 1);
        const uint32_t cr = (uint32_t)((lane & 15) ^ er);
        const char *sr = reinterpret_cast<const char *>(a.rs) + ((uint32_t)((lane & 16) ? rid1 : rid0) * 256u + cr * 16u);
        __builtin_amdgcn_sched_barrier(0);
        if (lane < 32)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sr,
                                             (__attribute__((address_space(3))) void *)(slot->r + wave * 512), 16, 0, QCN_RS_AUX);
    } else {
        const uint32_t cr = (uint32_t)(((pc & 15) ^ e) | (pc & 16));     // rhat: logical chunk (hi half 0..15, lo half 16..31)
        const char *sr = reinterpret_cast<const char *>(a.rs) + ((uint32_t)x.rid * 512u + cr * 16u);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sr,
                                         (__attribute__((address_space(3))) void *)(slot->r + wave * 1024), 16, 0, QCN_RS_AUX);
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sk,
                                     (__attribute__((address_space(3))) void *)(slot->k + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sv,
                                     (__attribute__((address_space(3))) void *)(slot->v + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// rowptr is read through the scalar unit (constant address space: it is not written during the launch, which the
// compiler cannot see through the struct argument) — a vector load here would put a vmcnt(0) into the tile pipeline
__device__ __forceinline__ int rowptr_at(const ModesArgs &a, int row)
{
    using cint = const __attribute__((address_space(4))) int32_t;
    return *((cint *)(uintptr_t)a.rowptr + row);
}

// smallest r in [0, n_rows] with rowptr[r] >= target (rowptr non-decreasing, rowptr[n_rows] >= target): a 64-ary search,
// every lane probes one row per round (three rounds for 16 000 rows)
__device__ __forceinline__ int rowptr_lower_bound(const int32_t *rowptr, int n_rows, int target, int lane)
{
    int lo = 0, hi = n_rows;                                   // the answer lies in [lo, hi]
    while (lo < hi) {
        const int step = (hi - lo + 63) / 64;
        const int p = min(lo + lane * step, hi);
        const unsigned long long m = __ballot(rowptr[p] >= target);    // monotone in the lane index
        if (m == 0) {
            lo = min(lo + 63 * step, hi) + 1;                  // beyond the last probe (which is < hi: rowptr[hi] >= target)
        } else {
            const int first = __ffsll((long long)m) - 1;
            const int new_lo = first == 0 ? lo : min(lo + (first - 1) * step, hi) + 1;
            hi = min(lo + first * step, hi);
            lo = new_lo;
        }
    }
    return __builtin_amdgcn_readfirstlane(lo);
}

// Contiguous chunk of CSR rows of logical block lb out of nb: chunks of equal EDGE count (block lb takes the rows whose
// first edge lies in [lb E / nb, (lb + 1) E / nb)) — with equal row counts the fullest of 256 chunks of the cfg3 decoder
// relations carries 9-11 % more tiles than the average and the launch waits for it; balanced, 1-2 %.
__device__ __forceinline__ void row_chunk(const int32_t *rowptr, int n_rows, int lb, int nb, int lane, int *row_begin, int *row_end)
{
    using cint = const __attribute__((address_space(4))) int32_t;
    const int first_edge = *((cint *)(uintptr_t)rowptr);              // a rowptr that is a slice of a larger CSR starts above 0
    const int num_edges = *((cint *)(uintptr_t)rowptr + n_rows) - first_edge;
    if (num_edges <= 0) {
        const int per = (n_rows + nb - 1) / nb;
        *row_begin = min(lb * per, n_rows);
        *row_end = min(*row_begin + per, n_rows);
        return;
    }
    *row_begin = lb == 0 ? 0 : rowptr_lower_bound(rowptr, n_rows, first_edge + (int)((long long)lb * num_edges / nb), lane);
    *row_end = lb + 1 >= nb ? n_rows
                            : rowptr_lower_bound(rowptr, n_rows, first_edge + (int)((long long)(lb + 1) * num_edges / nb), lane);
}

// first tile of the first non-empty row in [row, row_end)
__device__ __forceinline__ Tile first_tile(const ModesArgs &a, int row, int row_end)
{
    while (row < row_end) {
        const int b = rowptr_at(a, row), e = rowptr_at(a, row + 1);
        if (e > b) return Tile{row, b, e, true, (b + 3 * row + 3) & ~3};
        ++row;
    }
    return Tile{-1, 0, 0, false, 0};
}

__device__ __forceinline__ Tile next_tile(const ModesArgs &a, Tile t, int row_end)
{
    if (t.row < 0) return t;
    if (t.j0 + kTile < t.end) { t.j0 += kTile; t.s0 += kTile; t.first = false; return t; }
    return first_tile(a, t.row + 1, row_end);
}

// s_waitcnt vmcnt(n) lgkmcnt(0) (gfx9 encoding: vmcnt = imm[15:14]:imm[3:0], lgkmcnt = imm[11:8], expcnt untouched)
constexpr int wait_imm(int n) { return 0x0070 | (n & 15) | ((n >> 4) << 14); }

// max over the 16 lanes of every DPP row, four independent values at once (one v_max_f32_dpp per value and step; the
// other three values between two steps of one value cover the VALU -> DPP wait states)
__device__ __forceinline__ void row16_max4(float (&v)[4])
{
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// Software pipeline.  In iteration n a wave
//   * issues the three DMAs (rhat, k, v rows) and the q load of tile n + kDepth, and the index loads of tile
//     n + kDepth + kIdxAhead,
//   * runs the softmax / weighted-sum half of tile n out of registers (kk, vv of the previous iteration), interleaved
//     with the projection MFMAs of tile n + 1 (ring slot (n + 1) % kBufs): the MFMA pipe works on the next tile while
//     the vector ALU does the exp / max / rescale chain of the current one,
//   * waits until the DMAs of tile n + 2 have landed (vmcnt((kDepth - 2) kOps): the younger iterations' loads stay in
//     flight) and meets the other seven waves at one raw s_barrier.
// Ring slots (tile t -> slot t % kBufs) are compile-time constants in the kBufs-times unrolled loop and every slot is a
// __shared__ variable of its own, so that neither register copies nor the compiler's LDS-DMA alias waits drain the
// queue.  Every iteration issues the same kOps vector-memory loads (clamped duplicates past the last tile).
#ifndef QCN_MODES_DEPTH
#define QCN_MODES_DEPTH 3
#endif
constexpr int kDepth = QCN_MODES_DEPTH;   // 3 .. 5
constexpr int kBufs = kDepth + 1;
constexpr int kIdxAhead = 2;

template <bool HAS_EID, bool BF16, bool STORE = false>
__global__ __launch_bounds__(64 * kHeads) void attn_modes_kernel(ModesArgs a)
{
    static_assert(kBufs >= 4 && kBufs <= 6, "one __shared__ slot per ring position below");
    __shared__ __attribute__((aligned(16))) Slot s_0;
    __shared__ __attribute__((aligned(16))) Slot s_1;
    __shared__ __attribute__((aligned(16))) Slot s_2;
    __shared__ __attribute__((aligned(16))) Slot s_3;
    struct Unused { unsigned char pad[16]; };
    __shared__ __attribute__((aligned(16))) std::conditional_t<(kBufs > 4), Slot, Unused> s_4;
    __shared__ __attribute__((aligned(16))) std::conditional_t<(kBufs > 5), Slot, Unused> s_5;
    __shared__ __attribute__((aligned(16))) float s_t[kHeads][16 * kTStride];            // alpha transposition
    constexpr int kOps = (HAS_EID ? 2 : 1) + 3 + 1;      // index loads + 3 DMAs + q per iteration
    constexpr int kTiles = kDepth + kIdxAhead + 1;       // tile descriptors carried: n .. n + kDepth + kIdxAhead

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;

    // contiguous, edge-balanced chunk of CSR rows per workgroup (XCD-contiguous through the block remap)
    const int nb = gridDim.x;
    const int lb = qcn_xcd_remap(blockIdx.x, gridDim.x);
    int row_begin, row_end;
    row_chunk(a.rowptr, a.n_rows, lb, nb, lane, &row_begin, &row_end);

    // the head's weight fragments: frag[lane][j] = W[16 w + (lane & 15)][32 ks + 8 (lane >> 4) + j]
    half8 wk_hi[4], wk_lo[4], wv_hi[4], wv_lo[4];
    bf16x8 wk_b[4], wv_b[4];          // bf16 form: [head][wk | wv][ks][lane] fragments, no scaling
    float usc_k = 1.f, usc_v = 1.f;
    if constexpr (BF16) {
        const bf16x8 *wb = reinterpret_cast<const bf16x8 *>(a.wfrag);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wk_b[ks] = wb[(size_t)((wave * 2) * 4 + ks) * 64 + lane];
            wv_b[ks] = wb[(size_t)((wave * 2 + 1) * 4 + ks) * 64 + lane];
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const half8 *base = a.wfrag + (size_t)((wave * 2) * 4 + ks) * 2 * 64 + lane;
            wk_hi[ks] = base[0];
            wk_lo[ks] = base[64];
            wv_hi[ks] = base[4 * 2 * 64];
            wv_lo[ks] = base[4 * 2 * 64 + 64];
        }
        const float *scales = reinterpret_cast<const float *>(a.wfrag + (size_t)kHeads * 2 * 4 * 2 * 64);
        usc_k = scales[0];
        usc_v = scales[1];
    }
    const float bias = a.bv[16 * wave + i];
    float *const s_tw = s_t[wave];
    const NodeAddr na = make_node_addr(a, wave, lane);

    // per-lane LDS byte offsets inside a slot (tile independent)
    constexpr int kRhatRow = BF16 ? 256 : kRowBytes;
    int r_off[4], v_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) r_off[ks] = i * kRhatRow + 16 * ((4 * ks + g) ^ i);               // rhat: row i, chunk 4 ks + g
    const int k_off = i * kRowBytes + 16 * ((4 * wave + g) ^ i);                                      // k: row i, floats 16 w + 4 g ..
#pragma unroll
    for (int t = 0; t < 4; ++t)                                                                       // v: row 4 g + t, float 16 w + i
        v_off[t] = (4 * g + t) * kRowBytes + 16 * ((4 * wave + (i >> 2)) ^ (4 * g + t)) + 4 * (i & 3);

    // rows without edges produce zeros (PyG softmax over an empty segment, attention_layer.py:118)
    for (int row = row_begin; row < row_end; ++row) {
        if (rowptr_at(a, row + 1) > rowptr_at(a, row)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < a.rep) *reinterpret_cast<float *>(reinterpret_cast<char *>(a.out) + ((uint32_t)row * na.row_mul + na.o_lane[r])) = 0.f;
    }

    Tile tl[kTiles];
    tl[0] = first_tile(a, row_begin, row_end);
    if (tl[0].row < 0) return;
#pragma unroll
    for (int j = 1; j < kTiles; ++j) tl[j] = next_tile(a, tl[j - 1], row_end);

    f32x4 qring[kBufs];
    TileIdx idx[kBufs];
    auto slot_of = [&](auto ph) -> Slot * {
        constexpr int S = decltype(ph)::value;
        if constexpr (S == 0) return &s_0;
        else if constexpr (S == 1) return &s_1;
        else if constexpr (S == 2) return &s_2;
        else if constexpr (S == 3) return &s_3;
        else if constexpr (S == 4) return reinterpret_cast<Slot *>(&s_4);
        else return reinterpret_cast<Slot *>(&s_5);
    };
    auto safe = [&](const Tile &t) -> Tile { return t.row >= 0 ? t : tl[0]; };

    // kk^T[d = 4 g + r][e = i] = usc_k W_kr[16 w + d, :] . rhat_e + k[src_e][16 w + d]
    // vv[e = 4 g + r][d = i]   = usc_v rhat_e . W_vr[16 w + d, :] + v[src_e][16 w + d]      of the tile in `slot`;
    // `between(ks)` is called after the MFMAs of k-step ks: the caller's vector-ALU work that overlaps them
    auto project = [&](const Slot *slot, f32x4 &kk, f32x4 &vv, auto between) {
        f32x4 kx = {0.f, 0.f, 0.f, 0.f}, vx = {0.f, 0.f, 0.f, 0.f};
        kk = f32x4{0.f, 0.f, 0.f, 0.f};
        vv = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16) {
            bf16x8 rb[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) rb[ks] = *reinterpret_cast<const bf16x8 *>(slot->r + r_off[ks]);
            const f32x4 kf = *reinterpret_cast<const f32x4 *>(slot->k + k_off);
            float vf[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) vf[r] = *reinterpret_cast<const float *>(slot->v + v_off[r]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk_b[ks], rb[ks], kk, 0, 0, 0);
                vv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[ks], wv_b[ks], vv, 0, 0, 0);
                between(ks);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                kk[r] += kf[r];
                vv[r] += vf[r];
            }
            return;
        }
        half8 rh[4], rl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            rh[ks] = *reinterpret_cast<const half8 *>(slot->r + r_off[ks]);
            rl[ks] = *reinterpret_cast<const half8 *>(slot->r + r_off[ks] + 256);
        }
        const f32x4 kf = *reinterpret_cast<const f32x4 *>(slot->k + k_off);
        float vf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vf[r] = *reinterpret_cast<const float *>(slot->v + v_off[r]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kk = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk_hi[ks], rh[ks], kk, 0, 0, 0);
            vv = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh[ks], wv_hi[ks], vv, 0, 0, 0);
            kx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk_hi[ks], rl[ks], kx, 0, 0, 0);
            vx = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl[ks], wv_hi[ks], vx, 0, 0, 0);
            kx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk_lo[ks], rh[ks], kx, 0, 0, 0);
            vx = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh[ks], wv_lo[ks], vx, 0, 0, 0);
            between(ks);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            kk[r] = fmaf(kk[r] + kx[r], usc_k, kf[r]);
            vv[r] = fmaf(vv[r] + vx[r], usc_v, vf[r]);
        }
    };

    // STORE: the tile's projections go to the head's planes of the tables (ModesArgs::kk): kk as one 16-B store per lane
    // (slot of edge e = i, floats 4 g ..), vv as one 16-B store per lane (group of edges 4 g .. 4 g + 3, channel i) — each a
    // contiguous 1 KiB per wave; kStoreOps vector-memory stores per tile, which the counted waits of the pipeline allow for
    constexpr int kStoreOps = STORE ? 2 : 0;
    char *const kk_plane = STORE ? reinterpret_cast<char *>(a.kk) + kv_plane_base(wave, a.table_slots) : nullptr;
    char *const vv_plane = STORE ? reinterpret_cast<char *>(a.vv) + kv_plane_base(wave, a.table_slots) : nullptr;
    auto store_kv = [&](const Tile &t, const f32x4 &kk_t, const f32x4 &vv_t) {
        if constexpr (STORE) {
            // Both instructions are issued by every wave, always (the waits count instructions; group 0 of a tile is never
            // empty).  kk: positions past the row's end hold the projection of the row's last edge — their DMAs were clamped
            // to it — and are written to ITS slot, the same bits again.  vv: a group that begins past the row's end is
            // masked off; the tail of the row's last group (its padding slots) receives the last edge's values.
            if (t.row >= 0) {         // (uniform; without a next tile the iteration returns before its wait)
                const int last = t.end - 1 - t.j0;
#ifndef QCN_NO_NT
                __builtin_nontemporal_store(kk_t, reinterpret_cast<f32x4 *>(kk_plane + kk_slot_off((uint32_t)(t.s0 + min(i, last)), wave, g)));
                if (4 * g <= last)
                    __builtin_nontemporal_store(vv_t, reinterpret_cast<f32x4 *>(vv_plane + vv_group_off((uint32_t)((t.s0 >> 2) + g), wave, i)));
#else
                *reinterpret_cast<f32x4 *>(kk_plane + kk_slot_off((uint32_t)(t.s0 + min(i, last)), wave, g)) = kk_t;
                if (4 * g <= last)
                    *reinterpret_cast<f32x4 *>(vv_plane + vv_group_off((uint32_t)((t.s0 >> 2) + g), wave, i)) = vv_t;
#endif
            }
        }
    };

    // prologue: DMAs + q of tiles 0 .. kDepth - 1, indices of tiles kDepth .. kDepth + kIdxAhead - 1, projection of tile 0
    {
        auto fill = [&](auto ph) {
            constexpr int J = decltype(ph)::value;
            if constexpr (J < kDepth) {
                const Tile t = safe(tl[J]);
                const TileIdx x = load_tile_idx<HAS_EID>(a, t, wave, lane);
                issue_tile_dma<BF16>(a, x, slot_of(ph), wave, lane);
                qring[J] = load_q(a, na, t.row);
                idx[J] = x;
            }
        };
        fill(std::integral_constant<int, 0>{}); fill(std::integral_constant<int, 1>{}); fill(std::integral_constant<int, 2>{});
        fill(std::integral_constant<int, 3>{}); fill(std::integral_constant<int, 4>{});
        qring[kDepth] = qring[0];
#pragma unroll
        for (int j = 0; j < kIdxAhead; ++j) idx[(kDepth + j) % kBufs] = load_tile_idx<HAS_EID>(a, safe(tl[kDepth + j]), wave, lane);
    }
    __builtin_amdgcn_s_waitcnt(wait_imm(0));
    __builtin_amdgcn_s_barrier();

    f32x4 qa = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f}, kk, vv;
    float m[4], l[4];
#ifdef QCN_MODES_STAMPS
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_it = 0;
#define STAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_t[k] += now_ - last_; last_ = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP(k)
#endif
    project(&s_0, kk, vv, [](int) {});
    store_kv(tl[0], kk, vv);

    // one iteration with the current tile in ring slot S; returns true after the last tile
    auto step = [&](auto ph) -> bool {
        constexpr int S = decltype(ph)::value;
        constexpr int SN = (S + 1) % kBufs;                    // slot of the next tile (projected now)
        constexpr int SB = (S + kDepth) % kBufs;               // slot of the tile kDepth ahead (DMAs + q now)
        constexpr int SC = (S + kDepth + kIdxAhead) % kBufs;   // slot of the indices loaded now
#ifdef QCN_MODES_STAMPS
        unsigned long long last_ = __builtin_amdgcn_s_memtime();
        ++n_it;
#endif
        const Tile t0 = tl[0], tb = safe(tl[kDepth]), tc = safe(tl[kTiles - 1]);
        const Tile tn = next_tile(a, tl[kTiles - 1], row_end);
        STAMP(0)
        issue_tile_dma<BF16>(a, idx[SB], slot_of(std::integral_constant<int, SB>{}), wave, lane);
        STAMP(1)
        qring[SB] = load_q(a, na, tb.row);
        idx[SC] = load_tile_idx<HAS_EID>(a, tc, wave, lane);
        STAMP(2)

        if (t0.first) {
            qa = qring[S] * (0.25f * kLog2e);     // head_dim ** -0.5 (layers/attention_layer.py:38), logits in base 2
            if (i >= a.rep) qa = f32x4{0.f, 0.f, 0.f, 0.f};
            o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
        }
        // ---- softmax / weighted sum of tile n (kk, vv in registers) around the projection MFMAs of tile n + 1
        const f32x4 vcur = vv;
        f32x4 p = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
        p = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0], kk[0], p, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[1], kk[1], p2, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[2], kk[2], p, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[3], kk[3], p2, 0, 0, 0);
        const bool valid = t0.j0 + i < t0.end;
        f32x4 pa;
        f32x4 kk_n, vv_n;
        project(slot_of(std::integral_constant<int, SN>{}), kk_n, vv_n, [&](int ks) {
            if (ks == 0) {
                // p[r] = logit[mode = 4 g + r][e = i]
                p += p2;
                float pv[4], mx[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { pv[r] = valid ? p[r] : -INFINITY; mx[r] = pv[r]; }
                row16_max4(mx);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float mn = fmaxf(m[r], mx[r]);
                    const float corr = __builtin_amdgcn_exp2f(m[r] - mn);
                    m[r] = mn;
                    p[r] = __builtin_amdgcn_exp2f(pv[r] - mn);
                    l[r] = fmaf(l[r], corr, p[r]);
                    o[r] *= corr;
                    s_tw[(4 * g + r) * kTStride + i] = p[r];
                }
            } else if (ks == 1) {
                wave_lds_fence();
                pa = *reinterpret_cast<const f32x4 *>(s_tw + i * kTStride + 4 * g);   // alpha[mode = i][e = 4 g + r]
            } else if (ks == 2) {
                wave_lds_fence();
            }
        });
        f32x4 o2 = {0.f, 0.f, 0.f, 0.f};
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[0], vcur[0], o, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[1], vcur[1], o2, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[2], vcur[2], o, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[3], vcur[3], o2, 0, 0, 0);
        o += o2;
        kk = kk_n;
        vv = vv_n;
        store_kv(tl[1], kk, vv);
        STAMP(3)

        const bool last = tl[1].row != t0.row;                  // last tile of its row
        if (last) {
            store_row_out(a, na, t0.row, g, l, o, bias);
        }
        STAMP(4)
        if (tl[1].row < 0) return true;
        // the DMAs of tile n + 2 have landed (vmcnt) and this wave's LDS reads are done (lgkmcnt 0); a raw barrier:
        // __syncthreads() would add a release fence, i.e. vmcnt(0)
        // (the four output stores of a row's last tile are younger than this iteration's loads and count as well)
        // (STORE: this iteration's kStoreOps projection stores are younger than its loads as well)
        if (last && a.rep >= 4) __builtin_amdgcn_s_waitcnt(wait_imm((kDepth - 2) * kOps + kStoreOps + 4));
        else __builtin_amdgcn_s_waitcnt(wait_imm((kDepth - 2) * kOps + kStoreOps));
        STAMP(5)
        __builtin_amdgcn_s_barrier();
        STAMP(6)
#pragma unroll
        for (int j = 0; j + 1 < kTiles; ++j) tl[j] = tl[j + 1];
        tl[kTiles - 1] = tn;
        return false;
    };
    while (true) {
        if (step(std::integral_constant<int, 0>{})) break;
        if (step(std::integral_constant<int, 1>{})) break;
        if (step(std::integral_constant<int, 2>{})) break;
        if (step(std::integral_constant<int, 3>{})) break;
        if constexpr (kBufs > 4) if (step(std::integral_constant<int, 4 % kBufs>{})) break;
        if constexpr (kBufs > 5) if (step(std::integral_constant<int, 5 % kBufs>{})) break;
    }
#ifdef QCN_MODES_STAMPS
    if (blockIdx.x == 0 && lane == 0 && a.stamps) {
        for (int k = 0; k < 7; ++k) a.stamps[wave * 8 + k] = acc_t[k];
        a.stamps[wave * 8 + 7] = n_it;
    }
#endif
}

// The same relation with the per-edge projections ALREADY in memory (kk, vv in CSR-slot order, written by a STORE launch
// of the kernel above for the same graph, layer and sources): the decoder's recurrent proposal steps apply one layer to
// the same sources three times (modules/qcnet_decoder.py:219-227), so steps two and three only need
//   logit[mode, e] = q[mode, head] . kk_e[head],   out[mode, head] = sum_e softmax_e(logit) vv_e[head]
// — per tile and wave five loads (kk: 16 B per lane, vv: 4 x 4 B, q), the eight fp32 MFMAs and the softmax of the kernel
// above, in the same order on the same bits (the result is bit-identical to recomputing the projections).  No LDS ring,
// no weights, no barriers: the eight waves (heads) of a workgroup walk the same row chunk independently; loads run
// kKvDepth tiles ahead in compile-time register slots (the compiler's own waitcnt bookkeeping suffices for plain loads).
constexpr int kKvDepth = 3;
constexpr int kKvBufs = kKvDepth + 1;
#ifndef QCN_KV_MIN_WAVES
#define QCN_KV_MIN_WAVES 4       // waves per SIMD the register allocation must allow (2 per workgroup)
#endif
constexpr int kKvWgPerCu = QCN_KV_MIN_WAVES / 2;

__global__ __launch_bounds__(64 * kHeads, QCN_KV_MIN_WAVES) void attn_kv_kernel(ModesArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_t[kHeads][16 * kTStride];            // alpha transposition
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int nb = gridDim.x;
    const int lb = qcn_xcd_remap(blockIdx.x, gridDim.x);
    int row_begin, row_end;
    row_chunk(a.rowptr, a.n_rows, lb, nb, lane, &row_begin, &row_end);
    const float bias = a.bv[16 * wave + i];
    float *const s_tw = s_t[wave];
    const NodeAddr na = make_node_addr(a, wave, lane);

    for (int row = row_begin; row < row_end; ++row) {          // rows without edges produce zeros
        if (rowptr_at(a, row + 1) > rowptr_at(a, row)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < a.rep) *reinterpret_cast<float *>(reinterpret_cast<char *>(a.out) + ((uint32_t)row * na.row_mul + na.o_lane[r])) = 0.f;
    }

    Tile tl[kKvBufs];                                            // tiles n .. n + kKvDepth
    tl[0] = first_tile(a, row_begin, row_end);
    if (tl[0].row < 0) return;
#pragma unroll
    for (int j = 1; j < kKvBufs; ++j) tl[j] = next_tile(a, tl[j - 1], row_end);

    f32x4 kkr[kKvBufs], vvr[kKvBufs], qr[kKvBufs];
    const char *const kk_plane = reinterpret_cast<const char *>(a.kk) + kv_plane_base(wave, a.table_slots);
    const char *const vv_plane = reinterpret_cast<const char *>(a.vv) + kv_plane_base(wave, a.table_slots);
    auto request = [&](auto ph, const Tile &t_in) {
        constexpr int S = decltype(ph)::value;
        const Tile t = t_in.row >= 0 ? t_in : tl[0];             // past the last tile: a harmless duplicate
        const int last = t.end - 1 - t.j0;
        // positions / groups past the row's end re-read its last edge / last group (finite values; their weights are 0)
#ifndef QCN_NO_NT
        kkr[S] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(kk_plane + kk_slot_off((uint32_t)(t.s0 + min(i, last)), wave, g)));
        vvr[S] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(vv_plane + vv_group_off((uint32_t)((t.s0 >> 2) + min(g, last >> 2)), wave, i)));
#else
        kkr[S] = ld_off<f32x4>(kk_plane, kk_slot_off((uint32_t)(t.s0 + min(i, last)), wave, g));
        vvr[S] = ld_off<f32x4>(vv_plane, vv_group_off((uint32_t)((t.s0 >> 2) + min(g, last >> 2)), wave, i));
#endif
        qr[S] = load_q(a, na, t.row);
    };
    request(std::integral_constant<int, 0>{}, tl[0]);
    request(std::integral_constant<int, 1>{}, tl[1]);
    request(std::integral_constant<int, 2>{}, tl[2]);

    f32x4 qa = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    float m[4], l[4];
    auto step = [&](auto ph) -> bool {
        constexpr int S = decltype(ph)::value;
        constexpr int SB = (S + kKvDepth) % kKvBufs;
        const Tile t0 = tl[0];
        request(std::integral_constant<int, SB>{}, tl[kKvDepth]);
        const Tile tn = next_tile(a, tl[kKvDepth], row_end);
        if (t0.first) {
            qa = qr[S] * (0.25f * kLog2e);
            if (i >= a.rep) qa = f32x4{0.f, 0.f, 0.f, 0.f};
            o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
        }
        const f32x4 kk = kkr[S];
        f32x4 p = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
        p = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0], kk[0], p, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[1], kk[1], p2, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[2], kk[2], p, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[3], kk[3], p2, 0, 0, 0);
        const bool valid = t0.j0 + i < t0.end;
        p += p2;
        float pv[4], mx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pv[r] = valid ? p[r] : -INFINITY; mx[r] = pv[r]; }
        row16_max4(mx);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mn = fmaxf(m[r], mx[r]);
            const float corr = __builtin_amdgcn_exp2f(m[r] - mn);
            m[r] = mn;
            p[r] = __builtin_amdgcn_exp2f(pv[r] - mn);
            l[r] = fmaf(l[r], corr, p[r]);
            o[r] *= corr;
            s_tw[(4 * g + r) * kTStride + i] = p[r];
        }
        wave_lds_fence();
        const f32x4 pa = *reinterpret_cast<const f32x4 *>(s_tw + i * kTStride + 4 * g);   // alpha[mode = i][e = 4 g + r]
        wave_lds_fence();
        f32x4 o2 = {0.f, 0.f, 0.f, 0.f};
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[0], vvr[S][0], o, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[1], vvr[S][1], o2, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[2], vvr[S][2], o, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[3], vvr[S][3], o2, 0, 0, 0);
        o += o2;
        if (tl[1].row != t0.row) store_row_out(a, na, t0.row, g, l, o, bias);     // last tile of its row
        if (tl[1].row < 0) return true;
#pragma unroll
        for (int j = 0; j + 1 < kKvBufs; ++j) tl[j] = tl[j + 1];
        tl[kKvBufs - 1] = tn;
        return false;
    };
    while (true) {
        if (step(std::integral_constant<int, 0>{})) break;
        if (step(std::integral_constant<int, 1>{})) break;
        if (step(std::integral_constant<int, 2>{})) break;
        if (step(std::integral_constant<int, 3>{})) break;
    }
}

// Lean variant: two ring slots, no projection-ahead, at most 128 registers — two workgroups per CU, so that one
// workgroup's DMA wait / barrier / row epilogue overlaps the other one's MFMA phase (the pipelined kernel above keeps
// one workgroup per CU and hides latency inside the workgroup instead).
template <bool HAS_EID>
__global__ __launch_bounds__(64 * kHeads, 4) void attn_modes_lean_kernel(ModesArgs a)
{
    __shared__ __attribute__((aligned(16))) Slot s_0;
    __shared__ __attribute__((aligned(16))) Slot s_1;
    __shared__ __attribute__((aligned(16))) float s_t[kHeads][16 * kTStride];
    constexpr int kLate = HAS_EID ? 2 : 1;                // loads issued after the DMAs in an iteration: the indices (+ q)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;

    const int nb = gridDim.x;
    const int lb = qcn_xcd_remap(blockIdx.x, gridDim.x);
    int row_begin, row_end;
    row_chunk(a.rowptr, a.n_rows, lb, nb, lane, &row_begin, &row_end);

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
    const NodeAddr na = make_node_addr(a, wave, lane);
    const int r_base = i * kRowBytes, r_x = 16 * i;       // rhat chunk of k-step ks: r_base + ((64 ks + 16 g) ^ r_x)
    const int k_off = i * kRowBytes + 16 * ((4 * wave + g) ^ i);
    const int v_base = 4 * g * kRowBytes + 4 * (i & 3), v_c = 4 * wave + (i >> 2);

    for (int row = row_begin; row < row_end; ++row) {
        if (rowptr_at(a, row + 1) > rowptr_at(a, row)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < a.rep) *reinterpret_cast<float *>(reinterpret_cast<char *>(a.out) + ((uint32_t)row * na.row_mul + na.o_lane[r])) = 0.f;
    }

    Tile t0 = first_tile(a, row_begin, row_end);
    if (t0.row < 0) return;
    Tile t1 = next_tile(a, t0, row_end), t2 = next_tile(a, t1, row_end);
    auto safe = [&](const Tile &t) -> Tile { return t.row >= 0 ? t : t0; };
    TileIdx x1;
    {
        const TileIdx x0 = load_tile_idx<HAS_EID>(a, t0, wave, lane);
        issue_tile_dma<false>(a, x0, &s_0, wave, lane);
        x1 = load_tile_idx<HAS_EID>(a, safe(t1), wave, lane);
    }
    __builtin_amdgcn_s_waitcnt(wait_imm(0));
    __builtin_amdgcn_s_barrier();

    f32x4 qa = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    float m[4], l[4];

    auto step = [&](auto ph) -> bool {
        constexpr int S = decltype(ph)::value;
        Slot *const cur = S == 0 ? &s_0 : &s_1;
        Slot *const nxt = S == 0 ? &s_1 : &s_0;
        const Tile tn = next_tile(a, t2, row_end);
        // DMAs of the next tile first, then the loads that may stay in flight over the barrier
        issue_tile_dma<false>(a, x1, nxt, wave, lane);
        x1 = load_tile_idx<HAS_EID>(a, safe(t2), wave, lane);

        const bool first = t0.first;
        if (first) {                    // q is used after the projection MFMAs: its latency hides behind them
            qa = load_q(a, na, t0.row) * (0.25f * kLog2e);
            if (i >= a.rep) qa = f32x4{0.f, 0.f, 0.f, 0.f};
            o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
        }
        // the small products first: the accumulator then grows the way a sorted sum does
        f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = r_base + ((64 * ks + 16 * g) ^ r_x);
            const half8 rh = *reinterpret_cast<const half8 *>(cur->r + off);
            const half8 rl = *reinterpret_cast<const half8 *>(cur->r + off + 256);
            kk = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk_hi[ks], rl, kk, 0, 0, 0);
            vv = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl, wv_hi[ks], vv, 0, 0, 0);
            kk = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk_lo[ks], rh, kk, 0, 0, 0);
            vv = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh, wv_lo[ks], vv, 0, 0, 0);
            kk = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk_hi[ks], rh, kk, 0, 0, 0);
            vv = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh, wv_hi[ks], vv, 0, 0, 0);
        }
        const f32x4 kf = *reinterpret_cast<const f32x4 *>(cur->k + k_off);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float vf = *reinterpret_cast<const float *>(cur->v + v_base + r * kRowBytes + 16 * (v_c ^ (4 * g + r)));
            kk[r] = fmaf(kk[r], usc_k, kf[r]);
            vv[r] = fmaf(vv[r], usc_v, vf);
        }
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) p = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[r], kk[r], p, 0, 0, 0);
        const bool valid = t0.j0 + i < t0.end;
        float pv[4], mx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pv[r] = valid ? p[r] : -INFINITY; mx[r] = pv[r]; }
        row16_max4(mx);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mn = fmaxf(m[r], mx[r]);
            const float corr = __builtin_amdgcn_exp2f(m[r] - mn);
            m[r] = mn;
            const float pe = __builtin_amdgcn_exp2f(pv[r] - mn);
            l[r] = fmaf(l[r], corr, pe);
            o[r] *= corr;
            s_tw[(4 * g + r) * kTStride + i] = pe;
        }
        wave_lds_fence();
        const f32x4 pa = *reinterpret_cast<const f32x4 *>(s_tw + i * kTStride + 4 * g);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], vv[r], o, 0, 0, 0);

        const bool last = t1.row != t0.row;
        if (last) {
            store_row_out(a, na, t0.row, g, l, o, bias);
        }
        if (t1.row < 0) return true;
        // everything older than this iteration's late loads (indices, q of a first tile, output stores of a last tile)
        // has landed: the next tile's DMAs
        const int late = (first ? 1 : 0) + (last && a.rep >= 4 ? 4 : 0);
        if (late == 0) __builtin_amdgcn_s_waitcnt(wait_imm(kLate));
        else if (late == 1) __builtin_amdgcn_s_waitcnt(wait_imm(kLate + 1));
        else if (late == 4) __builtin_amdgcn_s_waitcnt(wait_imm(kLate + 4));
        else __builtin_amdgcn_s_waitcnt(wait_imm(kLate + 5));
        __builtin_amdgcn_s_barrier();
        t0 = t1; t1 = t2; t2 = tn;
        return false;
    };
    while (true) {
        if (step(std::integral_constant<int, 0>{})) break;
        if (step(std::integral_constant<int, 1>{})) break;
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

extern "C" int64_t qcn_attn_modes_table_slots(int64_t num_edges, int64_t n_rows)
{
    if (num_edges < 0 || n_rows < 0) return -1;
    return (num_edges + 3 * n_rows + 7) & ~(int64_t)3;     // >= ((rowptr[r] + 3 r + 3) & ~3) + degree(r), rounded to a group
}

// slots are addressed by 32-bit byte offsets inside a head plane (64 B per slot)
static bool table_slots_ok(int64_t table_slots, int64_t num_edges, int64_t n_rows)
{
    return table_slots >= qcn_attn_modes_table_slots(num_edges, n_rows) && (table_slots & 3) == 0 &&
           table_slots < ((int64_t)1 << (QCN_KV_PLANES ? 26 : 23));
}

extern "C" size_t qcn_attn_modes_pack_bytes(void) { return (size_t)kHeads * 2 * 4 * 2 * 64 * 16 + 2 * sizeof(float); }

static int launch_attn_modes(bool bf16, const float *q, const float *k, const float *v, const void *rhat_op,
                             const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                             const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                             int64_t n_feat_rows, int rep, int layout, int64_t col_shift, float *out, void *stream,
                             bool store = false, float *kk_out = nullptr, float *vv_out = nullptr, int64_t table_slots = 0)
{
    if (store && num_edges > 0 && (!kk_out || !vv_out)) return QCN_ERR_INVALID_ARGUMENT;
    if (store && !table_slots_ok(table_slots, num_edges, n_rows)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst < 0 || n_src < 0 || n_rows < 0 || num_edges < 0 || rep < 1 || rep > 6) return QCN_ERR_INVALID_ARGUMENT;
    if (layout != 1 && layout != 2) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows * rep != n_dst) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst == 0) return QCN_OK;
    if (!q || !out || !rowptr || !wpack || !bv) return QCN_ERR_INVALID_ARGUMENT;
    if (num_edges > 0 && (!k || !v || !col || !rhat_op)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst >= ((int64_t)1 << 31) || n_src >= ((int64_t)1 << 31) || num_edges >= ((int64_t)1 << 31))
        return QCN_ERR_UNSUPPORTED;
    if (col_shift < 0 || col_shift >= ((int64_t)1 << 31)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_feat_rows < 0 || (!eid && n_feat_rows < num_edges)) return QCN_ERR_INVALID_ARGUMENT;
    // the kernel addresses q / out, k / v and rhat rows by 32-bit byte offsets (at most 512 B per row)
    const int64_t max_rows = (int64_t)1 << 23;
    if (n_dst >= max_rows || n_src >= max_rows || n_feat_rows >= max_rows) return QCN_ERR_UNSUPPORTED;
    ModesArgs args{q, k, v, rhat_op, rowptr, col, eid, reinterpret_cast<const half8 *>(wpack), bv, out, (int)n_rows, rep,
                   layout, (int)col_shift, nullptr, kk_out, vv_out, (long long)table_slots};
#ifdef QCN_MODES_STAMPS
    static unsigned long long *d_stamps = nullptr;
    if (!d_stamps) (void)hipMalloc(&d_stamps, 64 * sizeof(unsigned long long));
    (void)hipMemsetAsync(d_stamps, 0, 64 * sizeof(unsigned long long), qcn_stream(stream));
    args.stamps = d_stamps;
#endif
    // one workgroup per CU (the LDS ring admits only one at a time anyway): with two contiguous row chunks per CU the second
    // prologue / pipeline fill costs more than the finer partition returns (isolated attention roofline 0.497 -> 0.523)
    int per_cu = 1;
    if (const char *env = qcn_knob("QCN_MODES_WG_PER_CU")) { const int want = atoi(env); if (want >= 1 && want <= 8) per_cu = want; }
    const int64_t n_cus = qcn_num_cus();
    int64_t grid = n_rows < n_cus * per_cu ? n_rows : n_cus * per_cu;
    hipStream_t s = qcn_stream(stream);
    // QCN_MODES_LEAN=1: the two-workgroups-per-CU variant (same results; measured equal on cfg3, kept for A/B runs)
    const char *lean_env = qcn_knob("QCN_MODES_LEAN");
    const bool lean = !store && !bf16 && lean_env && atoi(lean_env) != 0;
    if (store) {
        if (bf16) {
            if (eid) hipLaunchKernelGGL((attn_modes_kernel<true, true, true>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
            else     hipLaunchKernelGGL((attn_modes_kernel<false, true, true>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
        } else {
            if (eid) hipLaunchKernelGGL((attn_modes_kernel<true, false, true>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
            else     hipLaunchKernelGGL((attn_modes_kernel<false, false, true>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
        }
    } else if (lean) {
        if (eid) hipLaunchKernelGGL((attn_modes_lean_kernel<true>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
        else     hipLaunchKernelGGL((attn_modes_lean_kernel<false>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
    } else if (bf16) {
        if (eid) hipLaunchKernelGGL((attn_modes_kernel<true, true>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
        else     hipLaunchKernelGGL((attn_modes_kernel<false, true>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
    } else {
        if (eid) hipLaunchKernelGGL((attn_modes_kernel<true, false>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
        else     hipLaunchKernelGGL((attn_modes_kernel<false, false>), dim3((unsigned)grid), dim3(64 * kHeads), 0, s, args);
    }
    const bool ok = hipGetLastError() == hipSuccess;
#ifdef QCN_MODES_STAMPS
    {
        unsigned long long h[64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; w += 7) {
            const double n = (double)h[w * 8 + 7];
            if (n > 0) printf("stamps wave %d iters %.0f: next_tile %.0f dma %.0f loads %.0f compute %.0f store %.0f wait %.0f barrier %.0f cycles per iteration\n",
                              w, n, h[w * 8] / n, h[w * 8 + 1] / n, h[w * 8 + 2] / n, h[w * 8 + 3] / n, h[w * 8 + 4] / n, h[w * 8 + 5] / n, h[w * 8 + 6] / n);
        }
    }
#endif
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}

extern "C" int qcn_attn_modes_f32(const float *q, const float *k, const float *v, const void *rhat_split,
                                  const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                                  const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                                  int64_t n_feat_rows, int rep, int layout, int64_t col_shift, float *out, void *stream)
{
    return launch_attn_modes(false, q, k, v, rhat_split, rowptr, col, eid, wpack, bv, n_dst, n_src, n_rows, num_edges,
                             n_feat_rows, rep, layout, col_shift, out, stream);
}

extern "C" int qcn_attn_modes_store_f32(const float *q, const float *k, const float *v, const void *rhat_split,
                                        const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                                        const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                                        int64_t n_feat_rows, int rep, int layout, int64_t col_shift, float *out,
                                        float *kk_out, float *vv_out, int64_t table_slots, void *stream)
{
    return launch_attn_modes(false, q, k, v, rhat_split, rowptr, col, eid, wpack, bv, n_dst, n_src, n_rows, num_edges,
                             n_feat_rows, rep, layout, col_shift, out, stream, true, kk_out, vv_out, table_slots);
}

extern "C" int qcn_attn_modes_store_bf16(const float *q, const float *k, const float *v, const void *rhat_bf16,
                                         const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                                         const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                                         int64_t n_feat_rows, int rep, int layout, int64_t col_shift, float *out,
                                         float *kk_out, float *vv_out, int64_t table_slots, void *stream)
{
    return launch_attn_modes(true, q, k, v, rhat_bf16, rowptr, col, eid, wpack, bv, n_dst, n_src, n_rows, num_edges,
                             n_feat_rows, rep, layout, col_shift, out, stream, true, kk_out, vv_out, table_slots);
}

extern "C" int qcn_attn_modes_cached_f32(const float *q, const float *kk, const float *vv, const int32_t *rowptr,
                                         const float *bv, int64_t n_dst, int64_t n_rows, int64_t num_edges,
                                         int64_t table_slots, int rep, int layout, float *out, void *stream)
{
    if (n_dst < 0 || n_rows < 0 || num_edges < 0 || rep < 1 || rep > 6) return QCN_ERR_INVALID_ARGUMENT;
    if (layout != 1 && layout != 2) return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows * rep != n_dst) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst == 0) return QCN_OK;
    if (!q || !out || !rowptr || !bv) return QCN_ERR_INVALID_ARGUMENT;
    if (num_edges > 0 && (!kk || !vv)) return QCN_ERR_INVALID_ARGUMENT;
    if (n_dst >= ((int64_t)1 << 23) || num_edges >= ((int64_t)1 << 31)) return QCN_ERR_UNSUPPORTED;   // 32-bit byte offsets
    if (!table_slots_ok(table_slots, num_edges, n_rows)) return QCN_ERR_INVALID_ARGUMENT;
    ModesArgs args{q, nullptr, nullptr, nullptr, rowptr, nullptr, nullptr, nullptr, bv, out, (int)n_rows, rep, layout, 0,
                   nullptr, const_cast<float *>(kk), const_cast<float *>(vv), (long long)table_slots};
    // two workgroups per CU: the kernel holds no LDS ring and few registers, and its waves only wait for loads
    const int64_t wgs = kKvWgPerCu * (int64_t)qcn_num_cus();
    const int64_t grid = n_rows < wgs ? n_rows : wgs;
    hipLaunchKernelGGL(attn_kv_kernel, dim3((unsigned)grid), dim3(64 * kHeads), 0, qcn_stream(stream), args);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

extern "C" size_t qcn_attn_modes_bf16_pack_bytes(void) { return (size_t)kHeads * 2 * 4 * 64 * 16; }

extern "C" int qcn_attn_modes_bf16(const float *q, const float *k, const float *v, const void *rhat_bf16,
                                   const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                                   const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                                   int64_t n_feat_rows, int rep, int layout, int64_t col_shift, float *out, void *stream)
{
    return launch_attn_modes(true, q, k, v, rhat_bf16, rowptr, col, eid, wpack, bv, n_dst, n_src, n_rows, num_edges,
                             n_feat_rows, rep, layout, col_shift, out, stream);
}
