// G0 / K7 — the relative-geometry vector every relation feeds its FourierEmbedding
// (utils/geometry.py:19-23,47-51; modules/qcnet_map_encoder.py:121-128, modules/qcnet_agent_encoder.py:133-139,161-176,
// modules/qcnet_decoder.py:161-167,183-188,201-206), one launch per relation instead of ~15 elementwise / gather ops:
//   dp = pos_src[src] - pos_dst[dst]
//   r  = [ |dp|_2,  atan2(c x dp, c . dp) with c = (cos, sin)(head_dst[dst]),  wrap(head_src[src] - head_dst[dst]) (, extra) ]
// One thread per edge; HBM-bound on the index and output streams (16 B of indices + 12-16 B out per edge; the gathered
// positions / headings are cache-served).  Every operation is the individually rounded fp32 operation of the reference's
// elementwise form (no fma contraction), and the dot product keeps the reference's reduction form 0 + a + b: for a zero
// dp the products can be -0 and (-0) + (-0) = -0 would turn atan2(0, .) from 0 into pi.
#include "common.hip.h"

namespace {

__global__ __launch_bounds__(256) void rel_geometry_kernel(const float *__restrict__ pos_src, const float *__restrict__ pos_dst,
                                                           const float *__restrict__ head_src,
                                                           const float *__restrict__ head_dst,
                                                           const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                           const int64_t *__restrict__ extra, int64_t n_edges, int pos_stride_src,
                                                           int pos_stride_dst, float *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int64_t j = src[e], i = dst[e];
    const float dx = __fsub_rn(pos_src[j * pos_stride_src], pos_dst[i * pos_stride_dst]);
    const float dy = __fsub_rn(pos_src[j * pos_stride_src + 1], pos_dst[i * pos_stride_dst + 1]);
    const float hd = head_dst[i];
    const float cx = cosf(hd), cy = sinf(hd);
    const float dist = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    const float cross = __fsub_rn(__fmul_rn(cx, dy), __fmul_rn(cy, dx));
    const float dot = __fadd_rn(__fadd_rn(0.0f, __fmul_rn(cx, dx)), __fmul_rn(cy, dy));
    const float ang = atan2f(cross, dot);
    // wrap_angle: remainder(a + pi, 2 pi) - pi with the floor-mod of torch.remainder
    const float kPi = 3.14159265358979323846f, kTwoPi = 6.28318530717958647692f;
    const float a = __fadd_rn(__fsub_rn(head_src[j], hd), kPi);
    float m = fmodf(a, kTwoPi);
    if (m != 0.0f && m < 0.0f) m = __fadd_rn(m, kTwoPi);
    const float wrapped = __fadd_rn(m, -kPi);
    if (extra != nullptr) {
        float4 o = make_float4(dist, ang, wrapped, (float)extra[e]);
        *reinterpret_cast<float4 *>(out + 4 * e) = o;
    } else {
        out[3 * e] = dist; out[3 * e + 1] = ang; out[3 * e + 2] = wrapped;
    }
}

}  // namespace

extern "C" int qcn_rel_geometry_f32(const float *pos_src, int pos_stride_src, const float *pos_dst, int pos_stride_dst,
                                    const float *head_src, const float *head_dst, const int64_t *src, const int64_t *dst,
                                    const int64_t *extra, int64_t n_edges, float *out, void *stream)
{
    if (n_edges < 0 || pos_stride_src < 2 || pos_stride_dst < 2) return QCN_ERR_INVALID_ARGUMENT;
    if (n_edges == 0) return QCN_OK;
    if (!pos_src || !pos_dst || !head_src || !head_dst || !src || !dst || !out) return QCN_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(rel_geometry_kernel, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, qcn_stream(stream), pos_src,
                       pos_dst, head_src, head_dst, src, dst, extra, n_edges, pos_stride_src, pos_stride_dst, out);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

// ---- proposals of the recurrent steps -> cumulative way-points (modules/qcnet_decoder.py:234-250) ----------------------------
// The decoder's recurrent proposal steps each emit a chunk of per-step increments [n_rows, steps_per_chunk * dim]; the
// reference concatenates the chunks and runs  cumsum(x)  (positions),  cumsum(elu(x) + 1) + 0.1  (scales),
// cumsum(tanh(x) * pi)  (headings),  1 / (cumsum(elu(x) + 1) + 0.02)  (concentrations)  over the step axis: two cats, an
// outer-dim scan (190 us at 96 000 x 60 x 2 in torch) and three elementwise launches per pair of outputs.  Here: one thread
// per (row, channel) walks the steps of all chunks — sequential fp32 sums, the order of a CPU cumsum.
namespace {

struct CumsumChunks {
    const float *c[4];
};

__global__ __launch_bounds__(256) void proposal_cumsum_kernel(CumsumChunks ch, int n_chunks, int64_t n_rows, int steps_per_chunk,
                                                              int dim, int transform, float offset, int reciprocal,
                                                              float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * dim) return;
    const int64_t row = i / dim;
    const int d = (int)(i - row * dim);
    float acc = 0.0f;
    float *o = out + row * (int64_t)(n_chunks * steps_per_chunk * dim) + d;
    for (int c = 0; c < n_chunks; ++c) {
        const float *src = ch.c[c] + row * (int64_t)(steps_per_chunk * dim) + d;
        for (int t = 0; t < steps_per_chunk; ++t) {
            float x = src[t * dim];
            if (transform == 1) x = __fadd_rn(x > 0.0f ? x : __fsub_rn(expf(x), 1.0f), 1.0f);     // elu(x) + 1
            else if (transform == 2) x = __fmul_rn(tanhf(x), 3.14159265358979323846f);          // tanh(x) * pi
            acc = __fadd_rn(acc, x);
            const float y = __fadd_rn(acc, offset);
            *o = reciprocal ? __fdiv_rn(1.0f, y) : y;
            o += dim;
        }
    }
}

}  // namespace

extern "C" int qcn_proposal_cumsum_f32(const float *const *chunks, int n_chunks, int64_t n_rows, int steps_per_chunk, int dim,
                                       int transform, float offset, int reciprocal, float *out, void *stream)
{
    if (n_chunks < 1 || n_chunks > 4 || n_rows < 0 || steps_per_chunk < 1 || dim < 1 || transform < 0 || transform > 2 || !chunks)
        return QCN_ERR_INVALID_ARGUMENT;
    if (n_rows == 0) return QCN_OK;
    CumsumChunks ch{};
    for (int c = 0; c < n_chunks; ++c) {
        if (!chunks[c]) return QCN_ERR_INVALID_ARGUMENT;
        ch.c[c] = chunks[c];
    }
    if (!out) return QCN_ERR_INVALID_ARGUMENT;
    const int64_t n = n_rows * dim;
    hipLaunchKernelGGL(proposal_cumsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qcn_stream(stream), ch, n_chunks,
                       n_rows, steps_per_chunk, dim, transform, offset, reciprocal, out);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}
