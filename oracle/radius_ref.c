/* CPU ORACLE — test infrastructure, never the product.
 *
 * Plain-C restatement of the radius search the reference calls through torch_cluster 1.6.1
 * (call sites: modules/qcnet_agent_encoder.py:158,168; modules/qcnet_map_encoder.py:141;
 * modules/qcnet_decoder.py:174,194).  torch_cluster is a third-party wheel that is not in
 * /root/reference and not installed here (environment.yml:178), so this follows its published
 * device algorithm: one scan per y row over the x rows of the same
 * example in ascending index order, squared fp32 distance accumulated as fma(d_k, d_k, acc),
 * strict `< r*r`, stop after max_num_neighbors hits.  PARITY UNPINNED (no reference fixture exists).
 *
 * Also the bit-level authority for the numpy/torch restatement in qcnet_oracle.py::radius.
 */
#include <math.h>
#include <stdint.h>

/* Returns the number of pairs written (<= cap); -1 if cap is too small.
 * row[e] = y index, col[e] = x index (global indices), grouped by ascending y. */
int64_t qcn_oracle_radius(const float *x, const float *y, const int64_t *ptr_x, const int64_t *ptr_y,
                          int64_t num_examples, int64_t dim, float r2, int64_t max_num_neighbors,
                          int64_t *row, int64_t *col, int64_t cap)
{
    int64_t e = 0;
    for (int64_t b = 0; b < num_examples; ++b) {
        for (int64_t ny = ptr_y[b]; ny < ptr_y[b + 1]; ++ny) {
            int64_t count = 0;
            for (int64_t nx = ptr_x[b]; nx < ptr_x[b + 1] && count < max_num_neighbors; ++nx) {
                float d0 = x[nx * dim] - y[ny * dim];
                float dist = d0 * d0;
                for (int64_t k = 1; k < dim; ++k) {
                    float dk = x[nx * dim + k] - y[ny * dim + k];
                    dist = fmaf(dk, dk, dist);
                }
                if (dist < r2) {
                    if (e >= cap) return -1;
                    row[e] = ny;
                    col[e] = nx;
                    ++e;
                    ++count;
                }
            }
        }
    }
    return e;
}

/* Segment softmax + weighted aggregation for one attention propagate call, fp64 accumulation:
 * an independent high-precision check of the HIP attention kernel (layers/attention_layer.py:94-101
 * + PyG softmax/scatter-add).  q [nd,H*D]; k, v [ns,H*D]; kr, vr [E,H*D] or NULL;
 * src, dst [E]; out [nd,H*D] fp64 (zeroed by the caller). */
void qcn_oracle_attn(const float *q, const float *k, const float *v, const float *kr, const float *vr,
                     const int64_t *src, const int64_t *dst, int64_t E, int64_t nd, int H, int D,
                     double *seg_max, double *seg_sum, double *out)
{
    const int C = H * D;
    const double scale = 1.0 / sqrt((double)D);
    for (int64_t i = 0; i < nd * H; ++i) { seg_max[i] = -INFINITY; seg_sum[i] = 0.0; }
    for (int pass = 0; pass < 3; ++pass) {
        for (int64_t e = 0; e < E; ++e) {
            const int64_t s = src[e], d = dst[e];
            for (int h = 0; h < H; ++h) {
                double sim = 0.0;
                for (int c = h * D; c < (h + 1) * D; ++c) {
                    float kj = k[s * C + c];
                    if (kr) kj = kj + kr[e * C + c];
                    sim += (double)q[d * C + c] * (double)kj;
                }
                sim *= scale;
                if (pass == 0) {
                    if (sim > seg_max[d * H + h]) seg_max[d * H + h] = sim;
                } else if (pass == 1) {
                    seg_sum[d * H + h] += exp(sim - seg_max[d * H + h]);
                } else {
                    const double a = exp(sim - seg_max[d * H + h]) / (seg_sum[d * H + h] + 1e-16);
                    for (int c = h * D; c < (h + 1) * D; ++c) {
                        float vj = v[s * C + c];
                        if (vr) vj = vj + vr[e * C + c];
                        out[d * C + c] += a * (double)vj;
                    }
                }
            }
        }
    }
}
