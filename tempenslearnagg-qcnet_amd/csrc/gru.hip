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

}  // namespace

extern "C" int qcn_gru_f32(const float *gi_r, const float *gi_z, const float *gi_n, const float *whh_t,
                           const float *b_hh, const float *h0, int64_t n_seq, int n_steps, float *h_out, void *stream)
{
    if (n_seq < 0 || n_steps < 1) return QCN_ERR_INVALID_ARGUMENT;
    if (n_seq == 0) return QCN_OK;
    if (!gi_r || !gi_z || !gi_n || !whh_t || !b_hh || !h0 || !h_out) return QCN_ERR_INVALID_ARGUMENT;
    hipStream_t s = qcn_stream(stream);
    const unsigned grid = (unsigned)((n_seq + kRowsPerWave - 1) / kRowsPerWave);
    hipLaunchKernelGGL(gru_kernel, dim3(grid), dim3(256), 0, s, gi_r, gi_z, gi_n, whh_t, b_hh, h0, n_seq, n_steps, h_out);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}
