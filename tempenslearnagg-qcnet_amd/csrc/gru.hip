// K10 — the recurrent half of nn.GRU(128 -> 128, 1 layer) as used by the decoder's trajectory embedder
// (modules/qcnet_decoder.py:86-88,257: 60 steps, batch = agents x modes, only the final hidden state is used).
// The input half  gi_t = W_ih x_t + b_ih  has no time dependence and is one large GEMM over all (sequence, step) rows
// (qcn_ln_linear_f32 without LayerNorm); this kernel runs the 60 dependent steps
//     r = sigmoid(gi_r + W_hr h + b_hr);  z = sigmoid(gi_z + W_hz h + b_hz);
//     n = tanh(gi_n + r * (W_hn h + b_hn));  h = (1 - z) * n + z * h
// for a block of 32 sequences in ONE launch (the reference path is ~15 library launches per step).
//
// Mapping: a workgroup = 32 sequences x 4 waves; wave w owns hidden columns [32w, 32w+32) of all three gates, i.e. three
// 32x32 accumulator tiles with K = 128.  Its slice of W_hh (3 x 128 x 32 floats) is loaded ONCE into 192 VGPRs in
// MFMA B-fragment order and reused for all 60 steps, so a step costs 192 fp32 MFMAs per wave, 64 LDS fragment reads of the
// shared hidden state and one barrier: latency-bound by the 60-step dependency chain, not by memory.
#include <type_traits>
#include "mfma_tile.hip.h"

namespace {

using namespace qcn_tile;

__global__ __launch_bounds__(256) void gru_kernel(const float *__restrict__ gi_r, const float *__restrict__ gi_z,
                                                  const float *__restrict__ gi_n, const float *__restrict__ whh_t,
                                                  const float *__restrict__ b_hh, const float *__restrict__ h0,
                                                  int64_t n_seq, int n_steps, float *__restrict__ h_out)
{
    __shared__ __attribute__((aligned(16))) float s_h[2][kC][kPad];   // hidden state, k-major, double-buffered
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * kRowsPerWave;
    const int col = 32 * wave + (lane & 31);

    // W_hh slice in B-fragment order: for k-step ks, gate g: B[k = 2 ks + (lane >> 5)][n = col]; whh_t is [128 k][384 n]
    float wfrag[3][64];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < 64; ++ks) wfrag[g][ks] = whh_t[(size_t)(2 * ks + (lane >> 5)) * (3 * kC) + g * kC + col];
    const float bhr = b_hh[col], bhz = b_hh[kC + col], bhn = b_hh[2 * kC + col];

    // h_0: the same learned vector for every sequence (traj_emb_h0, qcnet_decoder.py:88,257)
    float hreg[16];
    {
        const float hv = h0[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) { hreg[r] = hv; s_h[0][col][acc_row(r, lane)] = hv; }
    }
    int64_t grow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + acc_row(r, lane);
        grow[r] = (row < n_seq ? row : n_seq - 1) * n_steps;
    }
    __syncthreads();

#pragma unroll 1
    for (int t = 0; t < n_steps; ++t) {
        const float (&hcur)[kC][kPad] = s_h[t & 1];
        float (&hnext)[kC][kPad] = s_h[(t + 1) & 1];
        float gr[16], gz[16], gn[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t off = (size_t)(grow[r] + t) * kC + col;
            gr[r] = gi_r[off]; gz[r] = gi_z[off]; gn[r] = gi_n[off];
        }
        f32x16 ar, az, an;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ar[r] = bhr; az[r] = bhz; an[r] = bhn; }
#pragma unroll
        for (int ks = 0; ks < 64; ++ks) {
            const float av = hcur[2 * ks + (lane >> 5)][lane & 31];
            ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wfrag[0][ks], ar, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wfrag[1][ks], az, 0, 0, 0);
            an = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wfrag[2][ks], an, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float rg = 1.0f / (1.0f + expf(-(gr[r] + ar[r])));
            const float zg = 1.0f / (1.0f + expf(-(gz[r] + az[r])));
            const float ng = tanhf(fmaf(rg, an[r], gn[r]));
            hreg[r] = fmaf(zg, hreg[r] - ng, ng);          // (1 - z) * n + z * h
            hnext[col][acc_row(r, lane)] = hreg[r];
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + acc_row(r, lane);
        if (row < n_seq) h_out[row * kC + col] = hreg[r];
    }
}

// ---- split-f16 mode (include/qcnet_hip.h: QCN_MATMUL_SPLIT_F16) --------------------------------------------------------
// Same recurrence, transposed like the chain kernels (W_hh is the MFMA A operand, the 32 sequences are N): wave w owns
// output features [32w, 32w+32) of the three gates.  Its W_hh slice lives in 192 VGPRs as hi / lo f16 fragments of
// v_mfma_f32_32x32x16_f16 (3 gates x 8 k16-steps x 2 planes x 4 registers), a step is 72 MFMAs (hi*hi + hi*lo + lo*hi)
// instead of 192 fp32 ones.  The hidden state is bounded by construction, |h_t| <= max(1, max|h_0|) (a convex
// combination of tanh values and h_{t-1}), so its hi / lo split uses ONE fixed power-of-two scale and no row reduction;
// the fp32 state itself stays in the owner's registers (the recurrence h = (1 - z) n + z h is evaluated in fp32), only
// the matmul operand is split.  Waves exchange the split state through a padded, double-buffered LDS image
// [plane][sequence][128 f16 + 8 pad] (row stride 272 B: the 16-byte fragment reads of 16 consecutive lanes fall on 16
// distinct bank groups).  gi rows of step t + 1 are requested before the MFMAs of step t.
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4v = __attribute__((ext_vector_type(4))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kHRow = kC + 8;               // halfs per LDS row
constexpr int kGruPackFloats = 3 * 4 * 8 * 2 * 64 * 4;   // [gate][wave][k16-step][plane][lane] x 16 B
constexpr int kGruScaleFloats = 8;          // {s_r, 1/s_r, s_z, 1/s_z, s_n, 1/s_n, s_h, 1/s_h}

// x - (float)hh[k] in one v_fma_mix_f32 (exact either way; v_cvt_f32_f16 + v_sub_f32 otherwise: see chain_tile.hip.h)
__device__ __forceinline__ float lo_of(float x, half2v hh, int k)
{
    float r;
    if (k == 0)
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(x));
    else
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(x));
    return r;
}

// Gate non-linearities on the hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each): sigmoid(x) = 1 / (1 + 2^(-x log2 e)),
// tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)); absolute error ~1e-7 on values in [-1, 1], a fraction of the instructions of
// the ocml expf / tanhf calls that dominated the step's dependent chain.
__device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ float fast_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x)
{
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// BLOCKED: the input halves come in the layout qcn_fourier_gru_input_f32 writes — [sequence block][step][wave 4][q 4][lane 64]
// [4 floats] per gate — so that the twelve loads of a step are twelve contiguous 1 KiB pieces per wave instead of 32-byte
// pieces of 32 different rows each.
template <bool BLOCKED>
__global__ __launch_bounds__(256) void gru_split_kernel(const float *__restrict__ gi_r, const float *__restrict__ gi_z,
                                                        const float *__restrict__ gi_n, const float *__restrict__ wpack,
                                                        const float *__restrict__ b_hh, const float *__restrict__ h0,
                                                        int64_t n_seq, int n_steps, float *__restrict__ h_out)
{
    __shared__ __attribute__((aligned(16))) _Float16 s_h[2][2][kRowsPerWave][kHRow];   // [buffer][plane][seq][feature]
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t seq = (int64_t)blockIdx.x * kRowsPerWave + n;
    const bool valid = seq < n_seq;
    const int64_t grow = (valid ? seq : n_seq - 1) * n_steps;
    const float *scales = wpack + kGruPackFloats;
    const float s_hid = scales[6];
    const float u_r = scales[1] * scales[7], u_z = scales[3] * scales[7], u_n = scales[5] * scales[7];

    half8 wf[3][8][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                wf[g][j][p] = __builtin_bit_cast(half8, *reinterpret_cast<const f32x4 *>(
                    wpack + ((((size_t)(g * 4 + wave) * 8 + j) * 2 + p) * 64 + lane) * 4));

    // this lane's 16 output features: 32 wave + 8 q + 4 hh + i  (register 4 q + i)
    float bhr[16], bhz[16], bhn[16], hreg[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * wave + 8 * q + 4 * hh;
        const f32x4 br = *reinterpret_cast<const f32x4 *>(b_hh + f0);
        const f32x4 bz = *reinterpret_cast<const f32x4 *>(b_hh + kC + f0);
        const f32x4 bn = *reinterpret_cast<const f32x4 *>(b_hh + 2 * kC + f0);
        const f32x4 hv = *reinterpret_cast<const f32x4 *>(h0 + f0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { bhr[4 * q + i] = br[i]; bhz[4 * q + i] = bz[i]; bhn[4 * q + i] = bn[i]; hreg[4 * q + i] = hv[i]; }
    }
    auto publish = [&](int buf) {   // split the lane's 16 state values and store them as 4-feature groups
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half4v hi4, lo4;
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const float x0 = hreg[4 * q + i] * s_hid, x1 = hreg[4 * q + i + 1] * s_hid;
                const half2v a = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
                const half2v b = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(lo_of(x0, a, 0), lo_of(x1, a, 1)));
                hi4[i] = a[0]; hi4[i + 1] = a[1]; lo4[i] = b[0]; lo4[i + 1] = b[1];
            }
            const int f0 = 32 * wave + 8 * q + 4 * hh;
            *reinterpret_cast<half4v *>(&s_h[buf][0][n][f0]) = hi4;
            *reinterpret_cast<half4v *>(&s_h[buf][1][n][f0]) = lo4;
        }
    };
    publish(0);

    f32x4 gnx[3][4];   // gi of the next step
    auto request = [&](int t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t off = BLOCKED ? ((size_t)blockIdx.x * n_steps + t) * (kRowsPerWave * kC) + ((4 * wave + q) * 64 + lane) * 4
                                       : (size_t)(grow + t) * kC + 32 * wave + 8 * q + 4 * hh;
            gnx[0][q] = *reinterpret_cast<const f32x4 *>(gi_r + off);
            gnx[1][q] = *reinterpret_cast<const f32x4 *>(gi_z + off);
            gnx[2][q] = *reinterpret_cast<const f32x4 *>(gi_n + off);
        }
    };
    request(0);
    __syncthreads();

#pragma unroll 1
    for (int t = 0; t < n_steps; ++t) {
        f32x4 gcur[3][4];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) gcur[g][q] = gnx[g][q];
        request(t + 1 < n_steps ? t + 1 : t);
        f32x16 ar, az, an;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ar[r] = 0.f; az[r] = 0.f; an[r] = 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const half8 bh = *reinterpret_cast<const half8 *>(&s_h[t & 1][0][n][16 * j + 8 * hh]);
            const half8 bl = *reinterpret_cast<const half8 *>(&s_h[t & 1][1][n][16 * j + 8 * hh]);
            ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][j][0], bh, ar, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][j][0], bh, az, 0, 0, 0);
            an = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[2][j][0], bh, an, 0, 0, 0);
            ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][j][0], bl, ar, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][j][0], bl, az, 0, 0, 0);
            an = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[2][j][0], bl, an, 0, 0, 0);
            ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][j][1], bh, ar, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][j][1], bh, az, 0, 0, 0);
            an = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[2][j][1], bh, an, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = fmaf(ar[r], u_r, bhr[r]), pz = fmaf(az[r], u_z, bhz[r]), pn = fmaf(an[r], u_n, bhn[r]);
            const float rg = fast_sigmoid(gcur[0][r >> 2][r & 3] + pr);
            const float zg = fast_sigmoid(gcur[1][r >> 2][r & 3] + pz);
            const float ng = fast_tanh(fmaf(rg, pn, gcur[2][r >> 2][r & 3]));
            hreg[r] = fmaf(zg, hreg[r] - ng, ng);          // (1 - z) * n + z * h
        }
        publish((t + 1) & 1);
        __syncthreads();
    }
    if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = hreg[4 * q + i];
            *reinterpret_cast<f32x4 *>(h_out + seq * kC + 32 * wave + 8 * q + 4 * hh) = o;
        }
    }
}


// ---- two groups of 32 sequences per workgroup, one group's MFMAs under the other group's gates --------------------------
// A step of gru_split_kernel is a serial chain per wave — 72 MFMAs (~2.3k cycles), then the gates (sixteen elements of two
// sigmoids and a tanh: ~2.5k cycles of vector issue), publish, barrier — and with the wave's 192 weight registers there is
// one wave per SIMD, so nothing fills either unit while the other works.  Here a workgroup carries TWO independent groups of
// 32 sequences, A and B, through the same weights, half a step apart:
//     phase Y(t):   MFMAs of B, step t      under  gates of A, step t   (publishes h_A(t+1));  barrier
//     phase X(t+1): MFMAs of A, step t+1    under  gates of B, step t   (publishes h_B(t+1));  barrier
// The LDS image of a group's state is single-buffered: it is read in one phase and rewritten in the next, a barrier apart.
// Register file: 192 weights + 2 x 48 accumulators + 2 x 16 state, so the input halves do NOT wait in registers: each wave
// copies its own 3 x 4 KiB of a step global -> LDS by LDS-DMA (the blocked layout is already lane-major, a lane reads back
// the 16 bytes its own copy wrote, no barrier involved), quarter by quarter into the place it has just consumed — two
// phases ahead of the use — and the gates read a quarter when they need it.  In-order completion makes the wait a constant:
// when quarter q is read, the copies issued after the one it waits for are those of six quarters (18 loads).  The phase
// barrier waits for LDS only: __syncthreads() would drain the load queue every phase.  Biases live in LDS, pre-multiplied.
__global__ __launch_bounds__(256) void gru_pair_kernel(const float *__restrict__ gi_r, const float *__restrict__ gi_z,
                                                       const float *__restrict__ gi_n, const float *__restrict__ wpack,
                                                       const float *__restrict__ b_hh, const float *__restrict__ h0,
                                                       int64_t n_seq, int n_steps, float *__restrict__ h_out)
{
    __shared__ __attribute__((aligned(16))) f32x4 s_gi[2][3][4][4][64];                // [group][gate][wave][quarter][lane]
    __shared__ __attribute__((aligned(16))) _Float16 s_h[2][2][kRowsPerWave][kHRow];   // [group][plane][seq][feature]
    __shared__ __attribute__((aligned(16))) float s_b[3][kC];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n_blocks = (n_seq + kRowsPerWave - 1) / kRowsPerWave;
    // group 1 of the last workgroup of an odd number of blocks repeats the last block and stores nothing
    const int64_t blk1 = 2 * (int64_t)blockIdx.x + 1 < n_blocks ? 2 * (int64_t)blockIdx.x + 1 : n_blocks - 1;
    const int64_t blk_of[2] = {2 * (int64_t)blockIdx.x, blk1};
    const float *scales = wpack + kGruPackFloats;
    const float s_hid = scales[6];
    // folded into the gates' constants: -log2e for r and z, 2 log2e for n (sigmoid and tanh on exp2 / rcp, see fast_sigmoid)
    constexpr float kL = 1.4426950408889634f;
    const float v_r = -kL * scales[1] * scales[7], v_z = -kL * scales[3] * scales[7], v_n = 2.0f * kL * scales[5] * scales[7];
    for (int i = tid; i < 3 * kC; i += 256) s_b[i / kC][i % kC] = b_hh[i] * (i < 2 * kC ? -kL : 2.0f * kL);

    auto request_q = [&](int grp, int t, int q) {
        t = t < n_steps ? t : n_steps - 1;                      // past the end: a harmless repeat (keeps the load count)
        const size_t off = ((size_t)blk_of[grp] * n_steps + t) * (kRowsPerWave * kC) + ((4 * wave + q) * 64 + lane) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gi_r + off),
                                         (__attribute__((address_space(3))) void *)&s_gi[grp][0][wave][q][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gi_z + off),
                                         (__attribute__((address_space(3))) void *)&s_gi[grp][1][wave][q][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gi_n + off),
                                         (__attribute__((address_space(3))) void *)&s_gi[grp][2][wave][q][0], 16, 0, 0);
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) request_q(0, 0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) request_q(1, 0, q);

    half8 wf[3][8][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                wf[g][j][p] = __builtin_bit_cast(half8, *reinterpret_cast<const f32x4 *>(
                    wpack + ((((size_t)(g * 4 + wave) * 8 + j) * 2 + p) * 64 + lane) * 4));

    // this lane's 16 output features of either group: 32 wave + 8 q + 4 hh + i  (register 4 q + i), carried times s_hid (a
    // power of two: exact), the form the split wants
    float hA[16], hB[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 hv = *reinterpret_cast<const f32x4 *>(h0 + 32 * wave + 8 * q + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) { hA[4 * q + i] = hv[i] * s_hid; hB[4 * q + i] = hv[i] * s_hid; }
    }
    auto publish_q = [&](int grp, const float (&h)[16], int q) {   // four features of the lane's state, split, to LDS
        half4v hi4, lo4;
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const float x0 = h[4 * q + i], x1 = h[4 * q + i + 1];
            const half2v a = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
            const half2v b = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(lo_of(x0, a, 0), lo_of(x1, a, 1)));
            hi4[i] = a[0]; hi4[i + 1] = a[1]; lo4[i] = b[0]; lo4[i + 1] = b[1];
        }
        const int f0 = 32 * wave + 8 * q + 4 * hh;
        *reinterpret_cast<half4v *>(&s_h[grp][0][n][f0]) = hi4;
        *reinterpret_cast<half4v *>(&s_h[grp][1][n][f0]) = lo4;
    };
    auto phase_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // One phase = four slots.  Slot q: the eighteen MFMAs of k-steps 2 q, 2 q + 1 of group `gm`, each between two scheduling
    // barriers, and in front of each a chunk of quarter q of the gates of group `gg` sized to the 24-32 cycles of issue that
    // an MFMA leaves (it takes the pipe for 32 cycles and the issue port for 8): four instructions of the transcendental unit
    // or up to eight others.  The gates are written stage by stage ACROSS the quarter's four elements: one element's gate
    // arithmetic is a chain of fourteen dependent instructions, and issued element by element (what the compiler makes of the
    // plain loop) the wave waits out every latency.  LDS reads (state fragments of the next k-step, biases and input halves of
    // the next quarter) are issued a few chunks ahead of their use.
    //     r = 1 / (1 + 2^xr), xr = -log2e (gi_r + u_r acc_r + b_r); z alike; n = 1 - 2 / (1 + 2^xn),
    //     xn = 2 log2e (gi_n + r (u_n acc_n + b_n))
    auto phase = [&](auto with_matmul, auto with_gates, int gm, f32x16 &mr, f32x16 &mz, f32x16 &mn, int gg, int t,
                     const f32x16 &gr, const f32x16 &gz, const f32x16 &gn, float (&h)[16]) {
        constexpr bool kM = decltype(with_matmul)::value, kG = decltype(with_gates)::value;
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        half8 fh[8], fl[8];
        f32x4 bb[4][3], gq[4][3];
        auto frag = [&](int j) {
            if (kM && j < 8) {
                fh[j] = *reinterpret_cast<const half8 *>(&s_h[gm][0][n][16 * j + 8 * hh]);
                fl[j] = *reinterpret_cast<const half8 *>(&s_h[gm][1][n][16 * j + 8 * hh]);
            }
        };
        auto quarter_inputs = [&](int q) {     // biases and input halves of quarter q
            if (kG && q < 4) {
                asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                const int f0 = 32 * wave + 8 * q + 4 * hh;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    bb[q][g] = *reinterpret_cast<const f32x4 *>(&s_b[g][f0]);
                    gq[q][g] = s_gi[gg][g][wave][q][lane];
                }
            }
        };
        frag(0);
        quarter_inputs(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int k = 0;
            auto mf = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                if (kM) {
                    const int j = 2 * q + k / 9, v = (k % 9) / 3, g = k % 3;
                    const half8 a = wf[g][j][v == 2 ? 1 : 0], b = v == 1 ? fl[j] : fh[j];
                    const bool first = j == 0 && v == 0;
                    if (g == 0) mr = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, first ? zero : mr, 0, 0, 0);
                    if (g == 1) mz = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, first ? zero : mz, 0, 0, 0);
                    if (g == 2) mn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, first ? zero : mn, 0, 0, 0);
                    ++k;
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (kG) {
                float xr[4], xz[4], pn[4], xn[4], gs[4];
// (pinned: side-effect-free arithmetic otherwise sinks across the scheduling barriers to its first use)
#define QCN_STAGE(dst, expr) _Pragma("unroll") for (int i = 0; i < 4; ++i) { dst = expr; pin(dst); }
                frag(2 * q + 1);
                QCN_STAGE(xr[i], fmaf(gr[4 * q + i], v_r, bb[q][0][i]))
                mf();
                QCN_STAGE(xz[i], fmaf(gz[4 * q + i], v_z, bb[q][1][i]))
                mf();
                QCN_STAGE(pn[i], fmaf(gn[4 * q + i], v_n, bb[q][2][i]))
                mf();
                QCN_STAGE(xr[i], fmaf(gq[q][0][i], -kL, xr[i]))
                mf();
                QCN_STAGE(xz[i], fmaf(gq[q][1][i], -kL, xz[i]))
                mf();
                QCN_STAGE(xr[i], __builtin_amdgcn_exp2f(xr[i]))
                mf();
                QCN_STAGE(xz[i], __builtin_amdgcn_exp2f(xz[i]))
                mf();
                QCN_STAGE(xr[i], xr[i] + 1.0f)
                mf();
                QCN_STAGE(gs[i], gq[q][2][i] * (2.0f * kL))
                QCN_STAGE(xz[i], xz[i] + 1.0f)
                mf();
                frag(2 * q + 2);
                QCN_STAGE(xr[i], __builtin_amdgcn_rcpf(xr[i]))
                mf();
                request_q(gg, t + 1, q);       // into the place just consumed
                QCN_STAGE(xn[i], fmaf(xr[i], pn[i], gs[i]))
                mf();
                QCN_STAGE(xn[i], __builtin_amdgcn_exp2f(xn[i]))
                mf();
                QCN_STAGE(xz[i], __builtin_amdgcn_rcpf(xz[i]))
                mf();
                quarter_inputs(q + 1);
                QCN_STAGE(xn[i], xn[i] + 1.0f)
                mf();
                QCN_STAGE(xn[i], __builtin_amdgcn_rcpf(xn[i]))
                mf();
                QCN_STAGE(xn[i], fmaf(xn[i], -2.0f * s_hid, s_hid))
                QCN_STAGE(xr[i], h[4 * q + i] - xn[i])
                mf();
                QCN_STAGE(h[4 * q + i], fmaf(xz[i], xr[i], xn[i]))     // (1 - z) * n + z * h, times s_hid
                mf();
                publish_q(gg, h, q);
                mf();
#undef QCN_STAGE
            } else {
                frag(2 * q + 1);
#pragma unroll
                for (int i = 0; i < 9; ++i) mf();
                frag(2 * q + 2);
#pragma unroll
                for (int i = 0; i < 9; ++i) mf();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const std::true_type on;
    const std::false_type off;

#pragma unroll
    for (int q = 0; q < 4; ++q) { publish_q(0, hA, q); publish_q(1, hB, q); }
    __syncthreads();

    f32x16 aAr, aAz, aAn, aBr = {}, aBz = {}, aBn = {};   // (B's are first written by Y(0); X(0) only names them)
    phase(on, off, 0, aAr, aAz, aAn, 1, 0, aBr, aBz, aBn, hB);            // X(0): nothing to put under it
    phase_barrier();                                                      // Y(0) rewrites the image X(0) has read
#pragma unroll 1
    for (int t = 0; t < n_steps - 1; ++t) {
        phase(on, on, 1, aBr, aBz, aBn, 0, t, aAr, aAz, aAn, hA);         // Y(t)
        phase_barrier();
        phase(on, on, 0, aAr, aAz, aAn, 1, t, aBr, aBz, aBn, hB);         // X(t + 1)
        phase_barrier();
    }
    phase(on, on, 1, aBr, aBz, aBn, 0, n_steps - 1, aAr, aAz, aAn, hA);   // Y(n_steps - 1)
    phase_barrier();
    phase(off, on, 0, aAr, aAz, aAn, 1, n_steps - 1, aBr, aBz, aBn, hB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // no copy may land after the workgroup has gone

#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        const int64_t sq = (2 * (int64_t)blockIdx.x + grp) * kRowsPerWave + n;
        if (sq < n_seq) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (grp ? hB[4 * q + i] : hA[4 * q + i]) * scales[7];
                *reinterpret_cast<f32x4 *>(h_out + sq * kC + 32 * wave + 8 * q + 4 * hh) = o;
            }
        }
    }
}

}  // namespace

extern "C" size_t qcn_gru_pack_floats(void) { return (size_t)kGruPackFloats + kGruScaleFloats; }

extern "C" int qcn_gru_f32(const float *gi_r, const float *gi_z, const float *gi_n, const float *whh_t,
                           const float *b_hh, const float *h0, int64_t n_seq, int n_steps, float *h_out, int matmul_mode,
                           void *stream)
{
    if (n_seq < 0 || n_steps < 1) return QCN_ERR_INVALID_ARGUMENT;
    if (matmul_mode != QCN_MATMUL_EXACT_F32 && matmul_mode != QCN_MATMUL_SPLIT_F16) return QCN_ERR_INVALID_ARGUMENT;
    if (n_seq == 0) return QCN_OK;
    if (!gi_r || !gi_z || !gi_n || !whh_t || !b_hh || !h0 || !h_out) return QCN_ERR_INVALID_ARGUMENT;
    hipStream_t s = qcn_stream(stream);
    const unsigned grid = (unsigned)((n_seq + kRowsPerWave - 1) / kRowsPerWave);
    if (matmul_mode == QCN_MATMUL_SPLIT_F16)
        hipLaunchKernelGGL(gru_split_kernel<false>, dim3(grid), dim3(256), 0, s, gi_r, gi_z, gi_n, whh_t, b_hh, h0, n_seq, n_steps,
                           h_out);
    else
        hipLaunchKernelGGL(gru_kernel, dim3(grid), dim3(256), 0, s, gi_r, gi_z, gi_n, whh_t, b_hh, h0, n_seq, n_steps, h_out);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

extern "C" int qcn_gru_blocked_f32(const float *gi_r, const float *gi_z, const float *gi_n, const float *whh_pack,
                                   const float *b_hh, const float *h0, int64_t n_seq, int n_steps, float *h_out, void *stream)
{
    if (n_seq < 0 || n_steps < 1) return QCN_ERR_INVALID_ARGUMENT;
    if (n_seq == 0) return QCN_OK;
    if (!gi_r || !gi_z || !gi_n || !whh_pack || !b_hh || !h0 || !h_out) return QCN_ERR_INVALID_ARGUMENT;
    const unsigned grid = (unsigned)((n_seq + kRowsPerWave - 1) / kRowsPerWave);
    hipLaunchKernelGGL(gru_pair_kernel, dim3((grid + 1) / 2), dim3(256), 0, qcn_stream(stream), gi_r, gi_z, gi_n, whh_pack,
                       b_hh, h0, n_seq, n_steps, h_out);

    QCN_CHECK_LAUNCH();
    return QCN_OK;
}
