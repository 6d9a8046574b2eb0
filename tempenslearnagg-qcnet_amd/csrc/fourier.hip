// A4 — the whole FourierEmbedding forward (layers/fourier_embedding.py:49-67) as one kernel:
//   x[N,in] -> per input dim i: [cos(2 pi x_i f_ij), sin(2 pi x_i f_ij), x_i] (129) -> Linear(129,128) -> LN -> ReLU ->
//   Linear(128,128); sum over i (+ categorical embeddings) -> LN -> ReLU -> Linear(128,128) [-> row normalise].
// Nothing but x and the [N,128] result touches HBM (the reference materialises [N,in,129] features and five [N,128]
// intermediates per input dim).  The contractions run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// fma chains, so the result stays within fp32 rounding of the reference — there is no reduced-precision step).
//
// Tiling: 128 rows per workgroup, 4 waves, each wave owns 32 complete rows (32 x 128 outputs = four 32x32 accumulator
// tiles), so both LayerNorms are in-register reductions over the 32 lanes of a wave half — no cross-wave traffic.
// A operand: the wave's private k-major LDS tile [128 k][32 rows] (+1 padding: conflict-free for the column-wise
// stores of the LN output and the row-wise fragment reads).  B operand: the pre-transposed weights (k-major, packed
// once on the host) streamed from L2 through a double-buffered 32-k LDS chunk shared by the 4 waves, one barrier per
// chunk, the next chunk's global loads in flight under the current chunk's 64 MFMAs per wave.
// MFMA-bound: 2*128*(129+128) flops per row and input dim + 2*128*128 for to_out.
#include "mfma_tile.hip.h"

namespace {

using namespace qcn_tile;

constexpr int kF = 64;
// packed parameter layout (floats); see ops.pack_fourier_weights
constexpr int kOffFreq = 0;                       // [64]
constexpr int kOffW1t = 64;                       // [128 k][128 n]
constexpr int kOffW1Last = kOffW1t + kC * kC;     // [128]
constexpr int kOffB1 = kOffW1Last + kC;
constexpr int kOffG1 = kOffB1 + kC;
constexpr int kOffBe1 = kOffG1 + kC;
constexpr int kOffW2t = kOffBe1 + kC;             // [128 k][128 n]
constexpr int kOffB2 = kOffW2t + kC * kC;
constexpr int kDimStride = kOffB2 + kC;
constexpr int kOutG = 0, kOutBe = kC, kOutWt = 2 * kC, kOutB = 2 * kC + kC * kC, kOutStride = 3 * kC + kC * kC;

template <int IN>
__global__ __launch_bounds__(256) void fourier_kernel(const float *__restrict__ x, const float *__restrict__ extra,
                                                      int extra_div, const float *__restrict__ pack,
                                                      float *__restrict__ y, int64_t n_rows, int normalize_out)
{
    __shared__ __attribute__((aligned(16))) Smem sm;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * kRowsPerBlock + wave * kRowsPerWave;
    const float eps = 1e-5f;
    const float *out_pack = pack + IN * kDimStride;

    Stream st;
    st.chunk = 0;
    prefetch_chunk(st, IN > 0 ? pack + kOffW1t : out_pack + kOutWt, tid);

    if (IN > 0) {
        // raw inputs of the wave's 32 rows (tail rows clamp to the last valid row; their stores are masked)
        for (int i = lane; i < IN * kRowsPerWave; i += 64) {
            const int r = i % kRowsPerWave, d = i / kRowsPerWave;
            int64_t row = row0 + r;
            row = row < n_rows ? row : n_rows - 1;
            sm.x[wave][d][r] = x[row * IN + d];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    f32x16 acc_y[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_y[t][r] = 0.f;

#pragma unroll 1
    for (int d = 0; d < IN; ++d) {
        const float *dp = pack + d * kDimStride;
        // ---- Fourier features into the A tile: k < 64 cos, 64 <= k < 128 sin (reference cat order, :60)
        {
            const float xv = sm.x[wave][d][lane & 31];
            const int fbase = (lane >> 5) * 32;
#pragma unroll 4
            for (int j = 0; j < 32; ++j) {
                const float f = dp[kOffFreq + fbase + j];
                // ((x * f) * 2) * pi in fp32, the reference's evaluation order (:58)
                const float ang = __fmul_rn(__fmul_rn(__fmul_rn(xv, f), 2.0f), 3.14159265358979323846f);
                float sn, cs;
                sincosf(ang, &sn, &cs);
                sm.a[wave][fbase + j][lane & 31] = cs;
                sm.a[wave][kF + fbase + j][lane & 31] = sn;
            }
        }
        // ---- h = W1 [features, x] + b1 : the 129th input column is a rank-1 term folded into the accumulator init
        f32x16 acc_h[4];
        {
            float xr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xr[r] = sm.x[wave][d][acc_row(r, lane)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = 32 * t + (lane & 31);
                const float b1 = dp[kOffB1 + col], wl = dp[kOffW1Last + col];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_h[t][r] = fmaf(xr[r], wl, b1);
            }
        }
        gemm_pass(acc_h, sm, st, dp + kOffW1t, dp + kOffW2t, wave, lane, tid);
        row_layer_norm<true, true>(acc_h, dp + kOffG1, dp + kOffBe1, lane, eps);
        acc_to_tile(acc_h, sm.a[wave], lane);
        // ---- y += W2 h + b2
        const float *next = (d + 1 < IN) ? pack + (d + 1) * kDimStride + kOffW1t : out_pack + kOutWt;
        gemm_pass(acc_y, sm, st, dp + kOffW2t, next, wave, lane, tid);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float b2 = dp[kOffB2 + 32 * t + (lane & 31)];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_y[t][r] += b2;
        }
    }

    if (extra != nullptr) {   // summed categorical embeddings (:65-66)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int64_t row = row0 + acc_row(r, lane);
            row = (row < n_rows ? row : n_rows - 1) / extra_div;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc_y[t][r] += extra[row * kC + 32 * t + (lane & 31)];
        }
    }

    // ---- to_out: LN -> ReLU -> Linear (:67)
    row_layer_norm<true, true>(acc_y, out_pack + kOutG, out_pack + kOutBe, lane, eps);
    acc_to_tile(acc_y, sm.a[wave], lane);
    f32x16 acc_o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float bo = out_pack[kOutB + 32 * t + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[t][r] = bo;
    }
    gemm_pass(acc_o, sm, st, out_pack + kOutWt, nullptr, wave, lane, tid);
    if (normalize_out) row_layer_norm<false, false>(acc_o, nullptr, nullptr, lane, eps);

#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + acc_row(r, lane);
        if (row < n_rows) {
#pragma unroll
            for (int t = 0; t < 4; ++t) y[row * kC + 32 * t + (lane & 31)] = acc_o[t][r];
        }
    }
}

}  // namespace

extern "C" size_t qcn_fourier_pack_floats(int input_dim)
{
    return (size_t)input_dim * kDimStride + kOutStride;
}

extern "C" int qcn_fourier_embed_f32(const float *x, int input_dim, const float *extra, int64_t extra_div,
                                     const float *pack, int64_t n_rows, int normalize_out, float *y, void *stream)
{
    if (n_rows < 0 || input_dim < 0 || input_dim > 4) return n_rows < 0 ? QCN_ERR_INVALID_ARGUMENT : QCN_ERR_UNSUPPORTED;
    if (n_rows == 0) return QCN_OK;
    if (!pack || !y || (input_dim > 0 && !x) || (input_dim == 0 && !extra)) return QCN_ERR_INVALID_ARGUMENT;
    if (extra_div < 1 || extra_div > 0x7fffffff) return QCN_ERR_INVALID_ARGUMENT;
    hipStream_t s = qcn_stream(stream);
    const unsigned grid = (unsigned)((n_rows + kRowsPerBlock - 1) / kRowsPerBlock);
    const double flops = (double)n_rows * (input_dim * 2.0 * kC * (129.0 + kC) + 2.0 * kC * kC);
    void *token = qcn_prof_begin(QCN_PROF_FOURIER, flops, s);
#define QCN_LAUNCH_FOURIER(IN)                                                                                   \
    hipLaunchKernelGGL((fourier_kernel<IN>), dim3(grid), dim3(256), 0, s, x, extra, (int)extra_div, pack, y, n_rows, \
                       normalize_out)
    switch (input_dim) {
    case 0: QCN_LAUNCH_FOURIER(0); break;
    case 1: QCN_LAUNCH_FOURIER(1); break;
    case 2: QCN_LAUNCH_FOURIER(2); break;
    case 3: QCN_LAUNCH_FOURIER(3); break;
    default: QCN_LAUNCH_FOURIER(4); break;
    }
#undef QCN_LAUNCH_FOURIER
    const bool ok = hipGetLastError() == hipSuccess;
    qcn_prof_end(token, s);
    return ok ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
}
