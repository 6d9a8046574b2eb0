// Two row tiles per wave: the software-pipelined form of the feature-major MFMA chain (chain_tile.hip.h).
//
// Blocks of four waves, ONE wave per SIMD with the whole 512-entry register file, each wave owning two 32-row tiles
// A and B that run half a phase apart in a single instruction stream:
//
//     barrier k | GEMM(A, W_k) with the epilogue of B behind W_(k-1) issued between its MFMAs
//               | GEMM(B, W_k) with the epilogue of A behind W_k     issued between its MFMAs | barrier k+1
//
// With two waves per SIMD (eight-wave blocks, one tile per wave) the per-row epilogues (LayerNorm, operand split, sincos:
// vector ALU) and the GEMMs (matrix pipe) of the two waves add up instead of overlapping: one barrier per matrix keeps
// the waves in lockstep, the measured overlap is < 10 % and the matrix pipe is busy 30-37 % of the time
// (profiles/r2_sq_counters.json).  Here the overlap is in program order: every GEMM is 32 slots of MFMAs and every
// epilogue is cut into 32 steps of <= ~24 vector instructions, step g is issued in slot g, and a scheduling barrier after
// every slot keeps the compiler from regrouping them.  An MFMA holds the issue port for 8 of its 32 cycles, so a slot of
// three MFMAs has room for ~18-24 independent vector instructions (MI355X_MICROARCH.md, "instructions HIDDEN per MFMA gap").
// LDS holds what it held before (two 64 KiB weight buffers, per-feature vectors, a 2 KiB transposition block per wave);
// a matrix is staged once per 256 rows and read by both tiles.
#pragma once
#include "chain_tile.hip.h"

namespace qcn_chain {

constexpr int kPairWaves = 4;                       // waves per block
constexpr int kPairRows = 2 * kRowsPerWave;         // rows per wave
constexpr int kPairBlockRows = kPairWaves * kPairRows;
constexpr int kSlots = 32;                          // epilogue steps per GEMM

// ---- GEMM with a callback between the MFMA groups ------------------------------------------------------------------
// acc += W in (accumulators left in the operands' scale, see gemm_raw), `between(g)` issued in slot g = 0..31.
// `half` = 0 / 1: which half of the wave's LDS-DMA pieces of the NEXT matrix this GEMM issues (the two GEMMs that read a
// matrix stage its successor together).
template <int MODE> struct PairPieces {
    static constexpr int per_wave = mat_pieces<MODE>() / kPairWaves;   // 16 (8 in the bf16 mode)
    static constexpr int per_gemm = per_wave / 2;
    static constexpr int every = kSlots / per_gemm;                    // a piece every 4th (8th) slot
};

template <int MODE>
__device__ __forceinline__ void stage_piece_pair(const float *__restrict__ next, f32x4 *next_dst, int wave, int lane, int half,
                                                 int i)
{
    const int piece = wave * PairPieces<MODE>::per_wave + half * PairPieces<MODE>::per_gemm + i;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(next + ((size_t)piece * 64 + lane) * 4),
                                     (__attribute__((address_space(3))) void *)(next_dst + piece * 64), 16, 0, 0);
}

// Per-feature parameters of the epilogue step in flight.  A step's LDS reads are issued ONE SLOT AHEAD (with a single wave
// per SIMD nothing else covers an LDS round trip inside a slot): slot g issues the reads of step g + 1 right behind its
// first MFMA, then computes step g from the registers its predecessor filled.  Shared by the wave's two tiles (their
// epilogues never overlap).
struct Staged {
    f32x4 a, b, c, d;
};

// A step's results are pinned to the slot that computes them: the optimiser is free to SINK side-effect-free arithmetic down
// to its first use (the row reduction eight slots later, the next GEMM), across any number of scheduling barriers — it did,
// leaving empty slots and one slot with all the work.  An empty volatile asm that takes the value in and out of a vector
// register cannot be moved across the slot's (side-effecting) scheduling barrier and costs no instruction.
__device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(unsigned &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(f32x4 &x)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) { float e = x[k]; pin(e); x[k] = e; }
}

// Issue order inside a slot.  Left alone the compiler issues a slot's MFMAs back to back (the wave then sits on the issue
// port until the pipe takes the third one), the step's vector instructions behind them and the LDS reads of the next slot
// last, where their latency is exposed at the next slot's first MFMA.  So a slot is two scheduling regions: (1) first MFMA,
// fragment reads of slot g + 1, LDS-DMA piece, parameter reads of step g + 1; (2) step g interleaved with the other MFMAs:
// up to NV vector instructions in front of each of them.
template <int NM, int NV>
__device__ __forceinline__ void slot_pattern()
{
#ifdef QCN_EXP_NOPATTERN
    return;
#endif
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);   // VALU
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
    }
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
}

// `next` is ALWAYS staged into `next_dst` (no branch inside the slots: control flow there lets the compiler sink the steps'
// vector work out of their slots); a caller with no next matrix passes any valid matrix-sized source and an idle buffer.
// acc += W in (accumulators left in the operands' scale, see gemm_raw); slot g = 0..31 runs `load(g + 1)` (the LDS reads
// of the next step into P) and `step(g, cur)` (cur = P as step g - 1 left it; the caller runs load(0) before the GEMM).
#ifdef QCN_SLOT_STAMPS
// diagnostic build only: s_memtime after every slot of the GEMM the kernel selects through qcn_slot_stamp_on
__device__ unsigned long long g_slot_stamps[kPairWaves][32];
#endif

// ZERO: the accumulators start at zero — the first MFMA of each takes the constant instead of 64 registers zeroed beforehand
// (32 v_mov_b64 per GEMM and tile in kernels whose vector issue is the bound)
template <bool ZERO = false, typename Load, typename Step>
__device__ __forceinline__ void gemm_bt(f32x16 (&acc)[4], const Operand<kModeSplit> &in, const f32x4 *w, int lane,
                                        const float *__restrict__ next, f32x4 *next_dst, int wave, int half, Staged &P, Load load,
                                        Step step, int stamp = 0)
{
    using PP = PairPieces<kModeSplit>;
    // weight fragments are requested TWO slots ahead: a slot that carries little vector work is three MFMAs long, less than an
    // LDS round trip at this occupancy
    f32x4 nh = w[lane], nl = w[64 + lane], nh2 = w[2 * 64 + lane], nl2 = w[3 * 64 + lane];
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) {            // g = 4 j + t
        const int j = g >> 2, t = g & 3;
        const half8 ah = __builtin_bit_cast(half8, nh), al = __builtin_bit_cast(half8, nl);
        nh = nh2;
        nl = nl2;
#ifndef QCN_EXP_NO_MFMA
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in.hi[j], ZERO && j == 0 ? zero : acc[t], 0, 0, 0);
#else
        acc[t][g & 15] += (float)ah[0] * (float)in.hi[j][0] + (float)al[1] * (float)in.lo[j][1];
#endif
#ifndef QCN_EXP_NO_LDS
        if (g + 2 < 32) { nh2 = w[(2 * (g + 2)) * 64 + lane]; nl2 = w[(2 * (g + 2) + 1) * 64 + lane]; }
#else
        if (g + 2 < 32) { nh2[g & 3] += 1.0f; nl2[g & 3] += 1.0f; }
#endif
#ifndef QCN_EXP_NO_STAGE
        if (g % PP::every == 0) stage_piece_pair<kModeSplit>(next, next_dst, wave, lane, half, g / PP::every);
#endif
        const Staged cur = P;
        load(g + 1);
        __builtin_amdgcn_sched_barrier(0);
#ifndef QCN_EXP_NO_MFMA
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in.lo[j], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, in.hi[j], acc[t], 0, 0, 0);
#endif
#ifndef QCN_EXP_NOSTEP
        step(g, cur);
#endif
        slot_pattern<2, 8>();
        __builtin_amdgcn_sched_barrier(0);
#ifdef QCN_SLOT_STAMPS
        if (stamp && blockIdx.x == 0 && lane == 0) g_slot_stamps[wave][g] = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

template <bool ZERO = false, typename Load, typename Step>
__device__ __forceinline__ void gemm_bt(f32x16 (&acc)[4], const Operand<kModeBf16> &in, const f32x4 *w, int lane,
                                        const float *__restrict__ next, f32x4 *next_dst, int wave, int half, Staged &P, Load load,
                                        Step step, int = 0)
{
    using PP = PairPieces<kModeBf16>;
    if constexpr (ZERO) fill_zero(acc);
    f32x4 cur_w = w[lane];
#pragma unroll
    for (int g = 0; g < 32; ++g) {            // g = 4 j + t
        const int j = g >> 2, t = g & 3;
        const bf16x8 a = __builtin_bit_cast(bf16x8, cur_w);
        if (g + 1 < 32) cur_w = w[(g + 1) * 64 + lane];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, in.v[j], acc[t], 0, 0, 0);
        if (g % PP::every == 0) stage_piece_pair<kModeBf16>(next, next_dst, wave, lane, half, g / PP::every);
        const Staged cur = P;
        load(g + 1);
        __builtin_amdgcn_sched_barrier(0);
        step(g, cur);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool ZERO = false, typename Load, typename Step>
__device__ __forceinline__ void gemm_bt(f32x16 (&acc)[4], const Operand<kModeExact> &in, const f32x4 *w, int lane,
                                        const float *__restrict__ next, f32x4 *next_dst, int wave, int half, Staged &P, Load load,
                                        Step step, int = 0)
{
    using PP = PairPieces<kModeExact>;
    if constexpr (ZERO) fill_zero(acc);
    f32x4 nxt = w[lane];
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const f32x4 a = nxt;
        const float b = in.v[s >> 4][s & 15];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b, acc[0], 0, 0, 0);
        if (s + 1 < 64) nxt = w[(s + 1) * 64 + lane];
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b, acc[1], 0, 0, 0);
        if (s % (2 * PP::every) == 0)
            stage_piece_pair<kModeExact>(next, next_dst, wave, lane, half, s / (2 * PP::every));
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b, acc[3], 0, 0, 0);
        if (s & 1) {
            const Staged cur = P;
            load((s >> 1) + 1);
            step(s >> 1, cur);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// an epilogue on its own (pipeline fill / drain): the same 32 steps without a GEMM around them
template <typename Load, typename Step>
__device__ __forceinline__ void run_steps(Staged &P, Load load, Step step)
{
#pragma unroll
    for (int g = 0; g < kSlots; ++g) {
        const Staged cur = P;
        load(g + 1);
        __builtin_amdgcn_sched_barrier(0);
        step(g, cur);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- operand construction, four elements at a time -------------------------------------------------------------------
// Registers 4 q .. 4 q + 3 of v[t] (a "group": features 32 t + 8 q + 4 h ..+3 of the lane's row) are half of operand
// fragment 2 t + (q >> 1).  `x` is already in the operand's static scale (make_operand_static).
// (lo_of: chain_tile.hip.h)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// elements 2 p, 2 p + 1 of group (t, q): one packed word of the hi plane and one of the lo plane
__device__ __forceinline__ void put2(Operand<kModeSplit> &o, int t, int q, int p, float x0, float x1)
{
    const int j = 2 * t + (q >> 1), w = 2 * (q & 1) + p;
    const half2v hh = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    const half2v ll = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(lo_of(x0, hh, 0), lo_of(x1, hh, 1)));   // exact in fp32
    unsigned hw = __builtin_bit_cast(unsigned, hh), lw = __builtin_bit_cast(unsigned, ll);
    pin(hw);
    pin(lw);
    u32x4 hv = __builtin_bit_cast(u32x4, o.hi[j]), lv = __builtin_bit_cast(u32x4, o.lo[j]);
    hv[w] = hw;
    lv[w] = lw;
    o.hi[j] = __builtin_bit_cast(half8, hv);
    o.lo[j] = __builtin_bit_cast(half8, lv);
}

__device__ __forceinline__ void put4(Operand<kModeSplit> &o, int t, int q, const f32x4 x)
{
    put2(o, t, q, 0, x[0], x[1]);
    put2(o, t, q, 1, x[2], x[3]);
}

__device__ __forceinline__ void put2(Operand<kModeBf16> &o, int t, int q, int p, float x0, float x1);
__device__ __forceinline__ void put4(Operand<kModeBf16> &o, int t, int q, const f32x4 x)
{
    put2(o, t, q, 0, x[0], x[1]);
    put2(o, t, q, 1, x[2], x[3]);
}

__device__ __forceinline__ void put4(Operand<kModeExact> &o, int t, int q, const f32x4 x)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) o.v[t][4 * q + k] = x[k];
}

// one pair (p = 0 / 1) of a group: elements 2 p, 2 p + 1
__device__ __forceinline__ void put2(Operand<kModeBf16> &o, int t, int q, int p, float x0, float x1)
{
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    const int j = 2 * t + (q >> 1), w = 2 * (q & 1) + p;
    unsigned pw = __builtin_bit_cast(unsigned, bf16x2{(__bf16)x0, (__bf16)x1});
    pin(pw);
    u32x4 v = __builtin_bit_cast(u32x4, o.v[j]);
    v[w] = pw;
    o.v[j] = __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void put2(Operand<kModeExact> &o, int t, int q, int p, float x0, float x1)
{
    o.v[t][4 * q + 2 * p] = x0;
    o.v[t][4 * q + 2 * p + 1] = x1;
}

__device__ __forceinline__ f32x4 grp(const f32x16 (&v)[4], int t, int q)
{
    return f32x4{v[t][4 * q], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]};
}

__device__ __forceinline__ void set_grp(f32x16 (&v)[4], int t, int q, const f32x4 x)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) v[t][4 * q + k] = x[k];
}

__device__ __forceinline__ f32x4 lds_vec4(const float *vec, int t, int q, int h)
{
    return *reinterpret_cast<const f32x4 *>(vec + 32 * t + 8 * q + 4 * h);
}

// The same off a per-lane base that already holds the lane half (`lane_base` = vector block + 4 h, made opaque to the
// optimiser by the caller): every parameter read is then ONE ds_read_b128 with an immediate offset.  Left to itself the
// compiler hoists the ~50 loop-invariant addresses of the post-loop epilogues out of the input-dim loop, spills them, and
// reloads each one from scratch memory in front of its read (a vmcnt(0) wait behind the LDS-DMA pieces in flight).
using lds_cfptr = const __attribute__((address_space(3))) float *;

__device__ __forceinline__ f32x4 lds_vec4_at(lds_cfptr lane_base, int float_offset, int t, int q)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(lane_base + float_offset + 32 * t + 8 * q);
}

// (an explicit LDS pointer: through a generic pointer the asm would hide the address space and the reads become flat loads)
__device__ __forceinline__ lds_cfptr opaque_lds_ptr(const float *p)
{
    lds_cfptr q = (lds_cfptr)p;
    asm volatile("" : "+v"(q));
    return q;
}

// ---- LayerNorm in steps ------------------------------------------------------------------------------------------------
// Row statistics of a tile: four partial sums per pass (elements r = k mod 4 of the lane's registers, in register order: the
// order of chain_tile.hip.h's layer_norm, so both kernel forms give bit-identical rows) and the finished mean / rstd.
struct RowStats {
    float part[4];
    float mean, rstd;
};

// steps 0 .. 7: v = v * u + a (groups 2 g and 2 g + 1, additive terms a0 / a1), partial sums of v (generic) or of v^2
// (centred rows)
template <bool CENTRED>
__device__ __forceinline__ void ln_step_scale_add(f32x16 (&v)[4], RowStats &st, int g, float u, const f32x4 a0, const f32x4 a1)
{
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) st.part[t] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = 2 * g + i, t = m >> 2, q = m & 3;
        const f32x4 a = i ? a1 : a0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float y = fmaf(v[t][4 * q + k], u, a[k]);
            pin(y);
            v[t][4 * q + k] = y;
            st.part[k] = CENTRED ? fmaf(y, y, st.part[k]) : st.part[k] + y;   // four chains: a single one runs at the fma latency
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) pin(st.part[k]);
}

// first half of a row reduction: the lane's sum, its partner half's request in flight
__device__ __forceinline__ float ln_reduce_begin(const RowStats &st)
{
    return (st.part[0] + st.part[1]) + (st.part[2] + st.part[3]);
}

// steps `0 .. 3`: centre four groups per step and accumulate the squares
__device__ __forceinline__ void ln_step_centre(f32x16 (&v)[4], RowStats &st, int g)
{
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) st.part[t] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float c = v[g][4 * q + k] - st.mean;
            pin(c);
            v[g][4 * q + k] = c;
            st.part[k] = fmaf(c, c, st.part[k]);
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) pin(st.part[k]);
}

// one group: y = v * rstd * gamma + beta (ReLU), already in the consumer's static operand scale -> operand fragment
template <int MODE, bool RELU>
__device__ __forceinline__ void ln_step_normalize_put(const f32x16 (&v)[4], const RowStats &st, Operand<MODE> &op, int m,
                                                      const f32x4 gm, const f32x4 bt)
{
    const int t = m >> 2, q = m & 3;
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float n = fmaf(v[t][4 * q + k] * st.rstd, gm[k], bt[k]);
        y[k] = RELU ? fmaxf(n, 0.f) : n;
    }
    put4(op, t, q, y);
}

// ---- coalesced row stores in passes (store_tile2 and its split-f16 / bf16 forms, chain_tile.hip.h) ---------------------
// The fp32 pass in two halves for the stepped epilogues: `xpose_pass` writes the lane's two groups and reads the two
// row-major pieces back (LDS round trip in flight), `store_pass` writes them to global memory one step later.
__device__ __forceinline__ void xpose_pass(const f32x16 (&v)[4], f32x4 *xpose, int lane, int tp, f32x4 &r0, f32x4 &r1)
{
    const int n = lane & 31, h = lane >> 5;
    const int t = tp >> 1, half = tp & 1;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) xpose[n * 4 + ((2 * qq + h) ^ ((n >> 2) & 3))] = grp(v, t, 2 * half + qq);
    const int ra = lane >> 2, rb = 16 + (lane >> 2), c = lane & 3;
    r0 = xpose[ra * 4 + (c ^ ((ra >> 2) & 3))];
    r1 = xpose[rb * 4 + (c ^ ((rb >> 2) & 3))];
}

__device__ __forceinline__ void store_pass(float *__restrict__ dst, int64_t row0, int64_t n_rows, int lane, int tp, const f32x4 r0,
                                           const f32x4 r1)
{
    const int t = tp >> 1, half = tp & 1, c = lane & 3;
    const int64_t ra = row0 + (lane >> 2), rb = ra + 16;
    if (ra < n_rows) *reinterpret_cast<f32x4 *>(dst + ra * kC + 32 * t + 16 * half + 4 * c) = r0;
    if (rb < n_rows) *reinterpret_cast<f32x4 *>(dst + rb * kC + 32 * t + 16 * half + 4 * c) = r1;
}

// pass tp = 0..7: feature block tp >> 1, half tp & 1 (two groups of the lane's row)
__device__ __forceinline__ void store_tile2_pass(const f32x16 (&v)[4], float *__restrict__ dst, int64_t row0, int64_t n_rows,
                                                 f32x4 *xpose, int lane, int tp)
{
    const int n = lane & 31, h = lane >> 5;
    const int t = tp >> 1, half = tp & 1;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) xpose[n * 4 + ((2 * qq + h) ^ ((n >> 2) & 3))] = grp(v, t, 2 * half + qq);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * i + (lane >> 2), c = lane & 3;
        const f32x4 x = xpose[r * 4 + (c ^ ((r >> 2) & 3))];
        if (row0 + r < n_rows) *reinterpret_cast<f32x4 *>(dst + (row0 + r) * kC + 32 * t + 16 * half + 4 * c) = x;
    }
}

__device__ __forceinline__ void store_tile2_split_pass(const f32x16 (&v)[4], _Float16 *__restrict__ dst, int64_t row0,
                                                       int64_t n_rows, f32x4 *xpose, int lane, float scale, int tp)
{
    using half4v = __attribute__((ext_vector_type(4))) _Float16;
    const int n = lane & 31, h = lane >> 5;
    half4v *const xh = reinterpret_cast<half4v *>(xpose);          // 8-byte units: row n = units 8 n .. 8 n + 7
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

// pass t = 0..3: one 32-feature block (64 B of bf16 per row)
__device__ __forceinline__ void store_tile2_bf16_pass(const f32x16 (&v)[4], __bf16 *__restrict__ dst, int64_t row0, int64_t n_rows,
                                                      f32x4 *xpose, int lane, int t)
{
    using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
    const int n = lane & 31, h = lane >> 5;
    bf16x4 *const xh = reinterpret_cast<bf16x4 *>(xpose);
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

}  // namespace qcn_chain
