/*
 * qcnet_hip.h — C ABI of libqcnet_hip.so, the MI355X (gfx950) implementation of the QCNet /
 * TempEnsLearnAgg graph-attention hot path.
 *
 * The reference (KaiYin77/TempEnSLearnAgg-QCNet) has no FFI of its own: its native work is delegated to
 * torch_cluster / torch_geometric / ATen.  Each entry point below therefore cites the
 * reference call site whose native work it replaces.
 *
 * Conventions
 *   - every function returns an int status: QCN_OK (0) or a negative QCN_ERR_* code; nothing throws;
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns every buffer
 *     (outputs and workspaces), nothing is allocated or freed behind the ABI;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it and no call synchronises;
 *   - feature width is fixed: C = 128 channels = H (8 heads) x D (16), the reference's AV2 configuration
 *     (predictors/qcnet_sliding.py:422,433-434); other widths return QCN_ERR_UNSUPPORTED;
 *   - index tensors at the API boundary are int64 (PyTorch / PyG convention); CSR arrays are int32.
 */
#ifndef QCNET_HIP_H
#define QCNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QCN_OK 0
#define QCN_ERR_INVALID_ARGUMENT (-1)
#define QCN_ERR_UNSUPPORTED (-2)
#define QCN_ERR_WORKSPACE_TOO_SMALL (-3)
#define QCN_ERR_LAUNCH_FAILED (-4)
#define QCN_ERR_TOO_LARGE (-5) /* size beyond the fixed capacity of an allocation-free entry point */

#define QCN_CHANNELS 128
#define QCN_HEADS 8
#define QCN_HEAD_DIM 16

/* Library version (major*10000 + minor*100 + patch) and status strings. */
int qcn_version(void);
const char *qcn_status_string(int status);

/* ------------------------------------------------------------------------------------------------
 * G1 / G2 — radius search.  Replaces torch_cluster.radius / radius_graph
 *   (modules/qcnet_agent_encoder.py:158,168; modules/qcnet_map_encoder.py:141;
 *    modules/qcnet_decoder.py:174,194; modules/temporal_aggregate.py:175,195).
 *
 * For every y row (ascending) the x rows of the same example are scanned in ascending index order; a
 * pair is a hit when the fp32 chain  d = dx*dx; d = fma(dy, dy, d) [; d = fma(dz, dz, d)]  is < r2.
 * The scan of a y row stops after `max_num_neighbors` hits.  A hit is EMITTED unless
 *   drop_self != 0 and x index == y index            (radius_graph(loop=False) self-pair removal), or
 *   mask_x != NULL and mask_x[x index] == 0, or mask_y != NULL and mask_y[y index] == 0
 *                                                    (the boolean column filters the modules apply right
 *                                                     after the search, e.g. qcnet_agent_encoder.py:160,170).
 * Filters never change which hits count towards the cap, exactly like filtering after the search.
 * Two passes: qcn_radius_count writes the emitted count per y row; the caller turns it into an exclusive
 * scan `offset` (qcn_exclusive_scan_i32) and allocates E = offset[n_y]; qcn_radius_fill writes
 * row[e] = y index, col[e] = x index (int64), grouped by ascending y, x ascending inside a group.
 * ptr_x / ptr_y: int64 [num_examples + 1] example boundaries (sorted batch vectors, as bucketize gives).
 * dim is 2 or 3.
 */
int qcn_radius_count(const float *x, const float *y, const int64_t *ptr_x, const int64_t *ptr_y,
                     int64_t num_examples, int64_t n_y, int dim, float r2, int max_num_neighbors,
                     int drop_self, const uint8_t *mask_x, const uint8_t *mask_y,
                     int32_t *count, void *stream);
int qcn_radius_fill(const float *x, const float *y, const int64_t *ptr_x, const int64_t *ptr_y,
                    int64_t num_examples, int64_t n_y, int dim, float r2, int max_num_neighbors,
                    int drop_self, const uint8_t *mask_x, const uint8_t *mask_y,
                    const int64_t *offset, int64_t *row, int64_t *col, void *stream);

/* Exclusive prefix sum of int32 counts into int64 offsets [n + 1] (offset[n] = total).
 * Replaces the masked_select compaction inside torch_cluster's radius op.  Workspace: query with
 * qcn_scan_workspace_bytes(n). */
size_t qcn_scan_workspace_bytes(int64_t n);
int qcn_exclusive_scan_i32(const int32_t *count, int64_t n, int64_t *offset, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K3 — destination-sorted CSR of an edge list.  Replaces the per-call index_select / scatter indexing PyG's
 * MessagePassing.propagate does on the raw COO edge_index (layers/attention_layer.py:118).
 *   src, dst : int64 [E] (edge_index rows 0 / 1);  0 <= dst < n_dst
 *   rowptr   : int32 [n_dst + 1];  col : int32 [E] source node per CSR slot;
 *   eid      : int32 [E] original edge id per CSR slot (stable: ascending inside a destination).
 */
size_t qcn_csr_workspace_bytes(int64_t num_edges, int64_t n_dst);
int qcn_csr_build(const int64_t *src, const int64_t *dst, int64_t num_edges, int64_t n_dst,
                  int32_t *rowptr, int32_t *col, int32_t *eid, void *workspace, size_t workspace_bytes,
                  void *stream);

/* qcn_csr_build for callers that cannot allocate — code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It has no counterpart in the reference (there the indexing happens inside PyG's propagate with
 * torch's caching allocator behind it).  Same arguments and bit-identical rowptr / col / eid as qcn_csr_build, but the
 * sort's workspace is a device array that lives inside the library (QCN_CSR_STATIC_SCRATCH_BYTES per GPU the library
 * runs on, held until the process ends) and the call makes no hipMalloc / hipFree or allocating library call.
 *   - num_edges <= QCN_CSR_STATIC_MAX_EDGES (qcn_csr_static_max_edges()), else QCN_ERR_TOO_LARGE and nothing is launched;
 *     QCN_ERR_TOO_LARGE also when rocPRIM's sort temporaries for this (num_edges, n_dst) exceed the array
 *     (qcn_csr_workspace_bytes(num_edges, n_dst) > QCN_CSR_STATIC_SCRATCH_BYTES: very large n_dst);
 *   - set-up: the array's address is looked up once per device (hipGetSymbolAddress under a mutex).  Call
 *     qcn_csr_static_max_edges() once per device, with that device current, BEFORE a graph capture begins (it performs
 *     the lookup); without it the first qcn_csr_build_static call on a device does, which must then be outside a capture;
 *   - every call on one GPU shares that GPU's array: calls that overlap on one GPU (two streams, two host threads)
 *     race — serialise them (one stream, or events between them); calls on different GPUs are independent;
 *   - the device current at the call is the one whose array is used: `stream` must belong to it. */
#define QCN_CSR_STATIC_MAX_EDGES (1 << 20)
#define QCN_CSR_STATIC_SCRATCH_BYTES (32 << 20)
int64_t qcn_csr_static_max_edges(void);
int qcn_csr_build_static(const int64_t *src, const int64_t *dst, int64_t num_edges, int64_t n_dst,
                         int32_t *rowptr, int32_t *col, int32_t *eid, void *stream);

/* ------------------------------------------------------------------------------------------------
 * A3 — fused gather + segment softmax + weighted scatter-add of AttentionLayer.message / PyG propagate /
 * PyG softmax (layers/attention_layer.py:87-101,118):
 *   for dst i, edges e=(j->i):  k_e = k[j] + kr[e],  v_e = v[j] + vr[e],
 *   s_e,h = (q[i,h,:] . k_e[h,:]) * D^-0.5,  a_e,h = exp(s - max_i) / (sum_i exp(s - max_i) + 1e-16),
 *   out[i,h,:] = sum_e a_e,h * v_e[h,:]            (0 for destinations without edges)
 *   q [n_dst,128]; k, v [n_src,128]; kr, vr [E,128] indexed by ORIGINAL edge id (both NULL when the layer
 *   has no positional embedding, layers/attention_layer.py:94); rowptr/col/eid from qcn_csr_build
 *   (eid == NULL means kr/vr rows are already in CSR order); out [n_dst,128].
 */
int qcn_attn_fwd_f32(const float *q, const float *k, const float *v, const float *kr, const float *vr,
                     const int32_t *rowptr, const int32_t *col, const int32_t *eid,
                     int64_t n_dst, int64_t n_src, int64_t num_edges, float *out, void *stream);

/* A3, fused positional form — same result as qcn_attn_fwd_f32 fed with kr = to_k_r(LN_r(r)), vr = to_v_r(LN_r(r))
 * (layers/attention_layer.py:82,94-96), but the two per-edge projections are never materialised (algebra in
 * csrc/attention_fused.hip).  rhat [E_base,128] = row-normalised r (qcn_row_normalize_f32: LN_r without its affine);
 * wk[o][c] = to_k_r.weight[o][c] * gamma[c], wv[o][c] = to_v_r.weight[o][c] * gamma[c],
 * bv = to_v_r.weight @ beta + to_v_r.bias  (gamma, beta = attn_prenorm_r affine), all fp32.
 * Mode replication (modules/qcnet_decoder.py:170,190-192,208-211: K mode queries share one edge list and one r):
 *   layout 0: n_rows == n_dst, rep == 1, destination node = CSR row;
 *   layout 1: destination node = mode * n_rows + row   (mode-major, pl2m / a2m);
 *   layout 2: destination node = row * rep + mode      (agent-major, t2m);   n_dst == n_rows * rep.
 * rowptr [n_rows+1], col / eid [E_base] as in qcn_csr_build (eid NULL: rhat rows are in CSR order).
 * rowptr entries are offsets into col / eid and need not start at 0: a window of consecutive rows of a larger graph is
 * passed as a pointer into that graph's rowptr (the sliding frames of predictors/temp_ens_learn_agg.py:197 are such
 * windows of one absolute-time graph); col_shift is subtracted from every col entry to re-base source ids.
 * row_index / src_offset (device int32 [n_dst], each may be NULL; layout 0, rep 1 only): destination idx uses CSR row
 * row_index[idx] (default idx) of the n_rows-row graph and reads source rows col - col_shift + src_offset[idx] of k / v.
 * Several destination sets over one CSR without copying it: the sliding frames of an absolute-time relation, the
 * per-frame special steps of the streaming encoder.  The caller guarantees 0 <= row_index[idx] < n_rows and
 * 0 <= source row < n_src (CSRGraph.validate checks both on request). */
int qcn_attn_fused_f32(const float *q, const float *k, const float *v, const float *rhat,
                       const int32_t *rowptr, const int32_t *col, const int32_t *eid, const float *wk,
                       const float *wv, const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows,
                       int64_t num_edges, int rep, int layout, int64_t col_shift, const int32_t *row_index,
                       const int32_t *src_offset, float *out, void *stream);

/* Measurement aid, no counterpart in the reference: dst[0 .. n_bytes) = src[0 .. n_bytes) as a float4 streaming copy
 * (16-byte aligned pointers, n_bytes a multiple of 16).  bench.py times it on 1 GiB to report the box's ACHIEVABLE HBM
 * bandwidth (read + write) next to the 8 TB/s peak of SURVEY.md 8(d). */
int qcn_stream_copy(const void *src, void *dst, size_t n_bytes, void *stream);

/* A3 for mode-replicated relations, on the matrix cores — same result as qcn_attn_fused_f32 with layout 1 or 2 and
 * rep <= 6 (the decoder's t2m / pl2m / a2m relations, modules/qcnet_decoder.py:190-226,259-263 with num_modes = 6):
 * per (CSR row, edge) the projected key / value  k[j] + to_k_r(LN_r(r_e)),  v[j] + to_v_r(LN_r(r_e))  are computed once
 * as split-f16 MFMA products and shared by the rep mode queries of the row (csrc/attention_modes.hip).
 *   rhat_split [E_base rows, 256] f16: row e = [hi (128) | lo (128)] with hi + lo = rhat_e * 2^11, made by
 *     qcn_split_rows_f16(rhat, n, 11, out)  (rhat = qcn_row_normalize_f32 output / a normalising qcn_fourier_embed_f32);
 *   wpack: qcn_attn_modes_pack_bytes() bytes = [head 8][wk | wv][k-step 4][hi | lo][lane 64][8 f16] with
 *     frag[lane][j] = W[16 head + (lane & 15)][32 kstep + 8 (lane >> 4) + j] * 2^sw  (wk, wv as in qcn_attn_fused_f32,
 *     sw = the power of two that puts max |W| into [2^14, 2^15)), then two fp32: 2^-(sw_k + 11), 2^-(sw_v + 11)
 *     (host: ops.pack_attn_modes);
 *   n_feat_rows = rows of rhat_split (>= num_edges without eid; every eid entry < n_feat_rows);
 *   the other arguments as in qcn_attn_fused_f32 (no row_index / src_offset).  Rows are addressed by 32-bit byte
 *   offsets: n_dst, n_src, n_feat_rows < 2^23, else QCN_ERR_UNSUPPORTED (use qcn_attn_fused_f32).
 * Logits and the weighted sums are exact-fp32 MFMAs; |k + to_k_r(.)| is not limited. */
int qcn_split_rows_f16(const float *x, int64_t n_rows, int scale_log2, void *y, void *stream);
size_t qcn_attn_modes_pack_bytes(void);
int qcn_attn_modes_f32(const float *q, const float *k, const float *v, const void *rhat_split,
                       const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                       const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                       int64_t n_feat_rows, int rep, int layout, int64_t col_shift, float *out, void *stream);

/* The same relation with bf16 positional projections (QCN_MATMUL_BF16 below; BASELINE.json configs[3]): the operands of the two
 * per-edge products W_kr LN_r(r_e), W_vr LN_r(r_e) are rounded to bf16 (round to nearest even), one v_mfma_f32_16x16x32_bf16
 * per product, fp32 accumulation; k[j] / v[j], the logits, the softmax and the weighted sums stay fp32.
 *   rhat_bf16 [E_base rows, 128] bf16 (a normalising qcn_fourier_embed_f32 with normalize_out = 3, or any RNE cast of rhat);
 *   wpack: qcn_attn_modes_bf16_pack_bytes() bytes = [head 8][wk | wv][k-step 4][lane 64][8 bf16], frag as above without
 *   scaling (host: ops.pack_attn_modes(mode=QCN_MATMUL_BF16)).  Everything else as qcn_attn_modes_f32. */
size_t qcn_attn_modes_bf16_pack_bytes(void);
int qcn_attn_modes_bf16(const float *q, const float *k, const float *v, const void *rhat_bf16,
                        const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                        const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                        int64_t n_feat_rows, int rep, int layout, int64_t col_shift, float *out, void *stream);

/* The same relation applied SEVERAL TIMES to the same sources by the same layer — the decoder's recurrent proposal steps
 * (modules/qcnet_decoder.py:219-227 run t2m / pl2m attention layer i three times on unchanged x_t / x_pl and unchanged
 * weights; only the mode queries change): the per-edge terms kk_e = k[j] + W_kr LN_r(r_e), vv_e = v[j] + W_vr LN_r(r_e) are
 * the same in every application.
 *   qcn_attn_modes_store_f32 / _bf16: qcn_attn_modes_f32 / _bf16, and additionally writes kk_e, vv_e of every edge to
 *     the tables kk_out, vv_out: table_slots * 128 fp32 each (caller-allocated, contents need no initialisation),
 *     table_slots >= qcn_attn_modes_table_slots(num_edges, n_rows) = (num_edges + 3 n_rows + 7) & ~3, a multiple of 4 and
 *     < 2^23, else QCN_ERR_INVALID_ARGUMENT.  Layout: the edge at CSR position j of row r (r counted from the first row
 *     of THIS rowptr) has slot
 *       s = ((rowptr[r] + 3 r + 3) & ~3) + (j - rowptr[r])            — every row starts on a multiple of four slots —
 *     and inside every group of four slots the heads are interleaved,
 *       kk_out[s / 4][head][s % 4][c] = kk_e[16 head + c]     vv_out[s / 4][head][c][s % 4] = vv_e[16 head + c]   (c < 16),
 *     so that a wave (= head) moves whole 256-byte pieces and the eight waves of a workgroup one contiguous 8 KiB per
 *     16-edge tile and table; slots between rows are padding (ops.attn_modes_table_rows turns the tables back into
 *     [num_edges, 128] rows).
 *   qcn_attn_modes_cached_f32: the propagate from those two tables (q [n_dst,128], the SAME rowptr [n_rows+1] and
 *     table_slots, bv as above; no k / v / rhat / col / eid / weights): bit-identical to what qcn_attn_modes_* returns for
 *     the same q, at 1 KiB read per edge — two 16-byte loads per lane, 16-edge tile and head — and without the
 *     projection MFMAs.  The caller keeps kk / vv alive and unchanged between the calls and
 *     guarantees that sources, graph and layer weights are those of the storing call.  The reference recomputes the
 *     projections in every step (layers/attention_layer.py:94-101 inside each propagate). */
int64_t qcn_attn_modes_table_slots(int64_t num_edges, int64_t n_rows);
int qcn_attn_modes_store_f32(const float *q, const float *k, const float *v, const void *rhat_split,
                             const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                             const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                             int64_t n_feat_rows, int rep, int layout, int64_t col_shift, float *out,
                             float *kk_out, float *vv_out, int64_t table_slots, void *stream);
int qcn_attn_modes_store_bf16(const float *q, const float *k, const float *v, const void *rhat_bf16,
                              const int32_t *rowptr, const int32_t *col, const int32_t *eid, const void *wpack,
                              const float *bv, int64_t n_dst, int64_t n_src, int64_t n_rows, int64_t num_edges,
                              int64_t n_feat_rows, int rep, int layout, int64_t col_shift, float *out,
                              float *kk_out, float *vv_out, int64_t table_slots, void *stream);
int qcn_attn_modes_cached_f32(const float *q, const float *kk, const float *vv, const int32_t *rowptr, const float *bv,
                              int64_t n_dst, int64_t n_rows, int64_t num_edges, int64_t table_slots, int rep, int layout,
                              float *out, void *stream);

/* Arithmetic of the MFMA chain kernels below (qcn_fourier_*_f32, qcn_ln_linear_f32, qcn_attn_post_f32, qcn_mlp_heads_f32,
 * qcn_gru_f32): their `matmul_mode` argument.  The library keeps no mode of its own (no global mutable state: every call
 * names its arithmetic, calls are re-entrant); weight packs are mode specific — pass the pack made for the same mode.
 *   QCN_MATMUL_EXACT_F32: v_mfma_f32_32x32x2_f32, exact fp32 fma chains.
 *   QCN_MATMUL_SPLIT_F16: every fp32 operand carried as hi + lo of two f16 numbers (22 significant bits) after an exact
 *     power-of-two scaling of each activation row / weight matrix to [2^14, 2^15); products hi*hi + hi*lo + lo*hi on
 *     v_mfma_f32_32x32x16_f16, fp32 accumulation.  Results agree with the exact mode to a few fp32 ulps (operand error
 *     <= max(2^-22 |x|, 2^-29 max|x|)); finite for |x| < 2^64.  The fp32-equivalent default of the Python layer.
 *   QCN_MATMUL_BF16: every matrix operand (weights and activations) rounded to bf16, round-to-nearest-even, one
 *     v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation; biases, rank-1 terms, LayerNorm, activations and residuals
 *     stay fp32 — BASELINE.json configs[3] ("bf16 projections"; the reference's counterpart would be autocast around its
 *     nn.Linear layers, layers/attention_layer.py:40-55, layers/fourier_embedding.py:34-46).  qcn_gru_f32 has no bf16
 *     form: the 60-step recurrence stays in the split mode (the GRU's INPUT projections follow the mode). */
#define QCN_MATMUL_EXACT_F32 0
#define QCN_MATMUL_SPLIT_F16 1
#define QCN_MATMUL_BF16 2

/* Chain-packed 128x128 matrix — the weight layout of the three MFMA chain kernels (csrc/chain_tile.hip.h): 16384 floats
 * plus a {scale, 1 / scale} pair that travels in the scale table of the pack.  For W = nn.Linear(128,128).weight
 * ([out][in]), with kin(s, h) = 32 (s >> 4) + (s & 3) + 8 ((s & 15) >> 2) + 4 h the input feature that position s of
 * lane half h holds in the register layout of a row vector:
 *   exact: P[s][lane][t] = W[32 t + (lane & 31)][kin(s, lane >> 5)],  s = 0..63, lane = 0..63, t = 0..3 (fp32);
 *          scale pair {1, 1}.
 *   split: W' = W * 2^(15 - e), 2^(e-1) <= max|W| < 2^e; hi = f16(W'), lo = f16(W' - hi);
 *          P[j][t][p][lane][i] = plane_p(W'[32 t + (lane & 31)][kin(8 j + i, lane >> 5)]),  j = 0..7, p = 0 (hi), 1 (lo),
 *          i = 0..7 (f16, 32768 of them); scale pair {2^(15 - e), 2^(e - 15)}.
 *   bf16:  P[j][t][lane][i] = bf16(W[32 t + (lane & 31)][kin(8 j + i, lane >> 5)]) (16384 bf16 = the first 8192 floats of
 *          the 16384-float slot, the rest is ignored); scale pair {1, 1}.
 * The first layer of a FourierEmbedding MLP uses kin(s, h) = 32 h + s for s < 32 (cosines) and 64 + 32 h + (s - 32)
 * otherwise (sines).  Host packer: ops.pack_chain_matrix. */

/* A4 — the whole FourierEmbedding.forward (layers/fourier_embedding.py:49-67) in one launch:
 *   x [n_rows, input_dim] (input_dim 0..4; NULL for 0), extra [ceil(n_rows / extra_div), 128] or NULL = the summed
 *   categorical embeddings, row n of the output uses extra row n / extra_div (modules/qcnet_agent_encoder.py:109
 *   repeats one type embedding per agent over T steps: extra_div = T); y [n_rows, 128].
 *   normalize_out != 0 additionally applies (y - mean) / sqrt(var + 1e-5) per row, the layer-independent half of the
 *   consuming AttentionLayers' attn_prenorm_r (layers/attention_layer.py:82); normalize_out == 2 writes that result
 *   as split-f16 rows instead (y = [n_rows, 256] f16, scale 2^11: the rhat_split operand of qcn_attn_modes_f32),
 *   normalize_out == 3 as bf16 rows (y = [n_rows, 128] bf16, round to nearest even: the operand of qcn_attn_modes_bf16).
 *   pack: qcn_fourier_pack_floats(input_dim) floats (host packer: ops.pack_fourier), the 2 input_dim + 1 chain-packed matrices
 *     [W1c_0 (cos | sin order above), W2c_0, W1c_1, ..., to_out.2.weight]
 *   where W1c_i = mlps.i.0.weight[:, :128] and W2c_i = mlps.i.3.weight with the mean over the OUTPUT features removed from
 *   every column (rows of (I - 11^T / 128) W; not in the bf16 mode): the mean of W f + b is linear in f, so the LayerNorm
 *   behind such a layer starts at the variance.  Then per input dim i
 *     [freqs.weight[i] (64) | centred mlps.i.0.weight[:, 128] | centred mlps.i.0.bias | mlps.i.1.weight * s1_i |
 *      mlps.i.1.bias * s1_i | {max |freqs[i][0:32]|, max |freqs[i][32:64]|, 0, 0}]
 *   followed by [centred sum_i mlps.i.3.bias | to_out.0.weight * so | to_out.0.bias * so | to_out.2.bias], then the scale
 *   table (32 floats): [2 i] = 1 / (matrix scale of W1c_i), [2 input_dim] = 1 / (so x matrix scale of to_out.2),
 *   [24] = so, [25] = 1 / so, [26] = 1 / (s1_i x matrix scale of W2c_i) — ONE value: the W2c_i are packed with matrix
 *   scales that make it equal for every i, so their products accumulate without any work between the GEMMs.
 *   s1_i / so (split mode; 1 otherwise) are the static power-of-two operand scales of the LayerNorm outputs: the largest
 *   2^k with (sqrt(127) max|gamma| + max|beta|) 2^k <= 2^15 — a LayerNorm output is bounded by that whatever its input, so the
 *   kernels split it into f16 hi + lo without scanning the row for its maximum.
 *   hidden_dim = 128, num_freq_bands = 64 only. */
size_t qcn_fourier_pack_floats(int input_dim);
int qcn_fourier_embed_f32(const float *x, int input_dim, const float *extra, int64_t extra_div,
                          const float *pack, int64_t n_rows, int normalize_out, float *y, int matmul_mode, void *stream);

/* A4 followed by up to three Linear(128,128) layers on the embedding before it leaves the registers:
 * y_j = W_j FourierEmbedding(x) + b_j, j < n_out <= 3 (the decoder's trajectory re-encoding, modules/qcnet_decoder.py:255-257:
 * y_emb followed by the GRU's input half W_i* y + b_i*).  wt / bias as for qcn_ln_linear_f32; the embedding is not stored. */
int qcn_fourier_linear_f32(const float *x, int input_dim, const float *extra, int64_t extra_div,
                           const float *pack, int64_t n_rows, const float *wt, const float *bias, int n_out,
                           float *y0, float *y1, float *y2, int matmul_mode, void *stream);

/* The same chain with the embedding's last Linear (to_out.2, layers/fourier_embedding.py:42-46,67) folded into the layers
 * that follow it: there is no non-linearity between to_out.2 and W_j, so the caller passes W'_j = W_j W_out and
 * b'_j = W_j b_out + b_j (products formed in fp64) and the kernel applies them to relu(LN(sum)) directly — one GEMM per
 * row less.  input_dim >= 1.  No counterpart in the reference (it runs the two Linear layers one after the other,
 * modules/qcnet_decoder.py:255-257); results differ from qcn_fourier_linear_f32 by fp32 rounding only. */
int qcn_fourier_folded_linear_f32(const float *x, int input_dim, const float *extra, int64_t extra_div,
                                  const float *pack, int64_t n_rows, const float *wt, const float *bias, int n_out,
                                  float *y0, float *y1, float *y2, int matmul_mode, void *stream);

/* A2 node side (K5/K6 of SURVEY.md §2b fused) — xn = LayerNorm(x) (skipped when ln_weight == NULL);
 * y_j = W_j xn + b_j for j < n_out <= 4 (the to_q / to_k / to_v / to_s projections of one AttentionLayer call,
 * layers/attention_layer.py:74-79,115-117,108).  x, y_j, xn_out: [n_rows,128]; wt: n_out chain-packed matrices
 * (16384 floats each) followed by their n_out {scale, 1 / scale} pairs; bias: [n_out,128] or NULL; xn_out may be NULL. */
int qcn_ln_linear_f32(const float *x, int64_t n_rows, const float *ln_weight, const float *ln_bias,
                      const float *wt, const float *bias, int n_out, float *y0, float *y1, float *y2,
                      float *y3, float *xn_out, int matmul_mode, void *stream);

/* A2 node side, after the propagate — gate/update, to_out, post-norm residual and the whole feed-forward block
 * (layers/attention_layer.py:83-84,106-108,119,121-122):
 *   g = sigmoid(to_g [agg, xn_dst]); u = agg + g (s_proj - agg); x1 = x_res + LN_post(to_out u);
 *   out = x1 + LN_ffpost(ff_mlp.3 relu(ff_mlp.0 LN_ffpre(x1))).
 * pack (qcn_attn_post_pack_floats() floats): 11 chain-packed 128x128 matrices in consumption order
 *   [to_g.weight[:, :128], to_g.weight[:, 128:], to_out.weight, then for hb = 0..3:
 *    ff_mlp.0.weight[128hb:128hb+128, :], ff_mlp.3.weight[:, 128hb:128hb+128]]
 * followed by [to_g.bias, to_out.bias, attn_postnorm.{weight,bias}, ff_prenorm.{weight,bias}, ff_mlp.0.bias (512),
 *   ff_mlp.3.bias, ff_postnorm.{weight,bias}, attn_prenorm_x_dst.{weight,bias}], then the scale table (32 floats, {scale, 1 / scale} per matrix).  xn_dst = attn_prenorm_x_dst(x_res) as
 * written by qcn_ln_linear_f32, or NULL: it is then recomputed from x_res (same arithmetic, no round trip).
 * All tensors [n_rows,128] fp32; out must not alias an input (it parks
 * x1 between the two residuals). */
size_t qcn_attn_post_pack_floats(void);
int qcn_attn_post_f32(const float *agg, const float *xn_dst, const float *s_proj, const float *x_res,
                      const float *pack, int64_t n_rows, float *out, int matmul_mode, void *stream);

/* The same block for CHAINED layers (a stack of AttentionLayer calls in which one layer's output is the next one's
 * destination input: modules/qcnet_decoder.py:219-227,259-263, qcnet_agent_encoder.py:181-187, qcnet_map_encoder.py:170-172).
 * Two projections move into this kernel, where their input row already sits in registers:
 *   - s_proj = to_s(attn_prenorm_x_dst(x_res)) (layers/attention_layer.py:108) is computed from the LayerNorm output the
 *     gate needs anyway — it is neither written by qcn_ln_linear_f32 nor read back here;
 *   - n_next > 0: y_j = W_j LN_next(out) + b_j, j < n_next <= 3 — the NEXT layer's attn_prenorm_x_dst + to_q (n_next = 1),
 *     or attn_prenorm_x_src + to_q / to_k / to_v of a layer over one node set (n_next = 3), layers/attention_layer.py:74-79 of
 *     the following call — applied to the output row before it leaves the registers: that layer then needs no
 *     qcn_ln_linear_f32 launch for its destination rows.  next_wt / next_bias: the qcn_ln_linear_f32 layout (n_next
 *     chain-packed matrices + their scale pairs; [n_next,128] or NULL); next_ln_weight / next_ln_bias: [128], plain;
 *     next_operand_scale (split mode): the power of two that puts the bound sqrt(127) max|gamma| + max|beta| of LN_next's
 *     output at 2^15 (its static operand scale; the kernel folds it into gamma / beta), or <= 0: per-row maxima.
 * pack (qcn_attn_post_chain_pack_floats() floats): 12 chain-packed matrices [to_g.weight[:, :128], to_g.weight[:, 128:],
 *   to_s.weight, to_out.weight, then the eight feed-forward blocks as above], the vectors of qcn_attn_post_f32's pack,
 *   to_s.bias, the scale table (32 floats: 12 {scale, 1 / scale} pairs, then {s_pre, 1 / s_pre, s_ff, 1 / s_ff, 1 / P}).
 *   Split mode: attn_prenorm_x_dst's and ff_prenorm's gamma / beta arrive multiplied by the static operand scales s_pre /
 *   s_ff of their outputs, ff_mlp.0.bias block hb by the hidden block's static scale s_h[hb], the 1 / scale slot of
 *   ff_mlp.0's block hb by s_h[hb] as well, and the four ff_mlp.3 blocks are packed with matrix scales P / s_h[hb]
 *   (host: ops.pack_attn_post_chain; bounds: |LN(x)| <= sqrt(127) max|gamma| + max|beta|, hidden through the L1 norms of the
 *   weight rows).  Other modes: plain parameters, table padded with ones.
 *   Same arithmetic per element as qcn_ln_linear_f32 + qcn_attn_post_f32 up to the operand scales (22-bit operands either way).
 *   out and y_j must not alias an input or each other.  Runs in blocks of at most four waves (one wave per SIMD). */
size_t qcn_attn_post_chain_pack_floats(void);
int qcn_attn_post_chain_f32(const float *agg, const float *x_res, const float *pack, int64_t n_rows, float *out,
                            const float *next_wt, const float *next_bias, const float *next_ln_weight,
                            const float *next_ln_bias, float next_operand_scale, int n_next, float *y0, float *y1, float *y2,
                            int matmul_mode, void *stream);

/* M4 glue — the cumulative way-points of the decoder's recurrent proposal steps (modules/qcnet_decoder.py:234-250): chunk c
 * (c < n_chunks <= 4, host array of device pointers) holds the increments of steps [c * steps_per_chunk, (c + 1) *
 * steps_per_chunk) as [n_rows, steps_per_chunk * dim]; out [n_rows, n_chunks * steps_per_chunk, dim] =
 *   post(cumsum over steps of f(x)),  f = identity (transform 0) | elu(x) + 1 (1) | tanh(x) * pi (2),
 *   post(s) = s + offset, or 1 / (s + offset) with reciprocal != 0.
 * Replaces torch.cat + torch.cumsum(dim=-2) + the elementwise ops around them; sums are sequential fp32 adds. */
int qcn_proposal_cumsum_f32(const float *const *chunks, int n_chunks, int64_t n_rows, int steps_per_chunk, int dim,
                            int transform, float offset, int reciprocal, float *out, void *stream);

/* G0 / K7 — relative geometry of an edge list (utils/geometry.py:19-23 angle_between_2d_vectors, :47-51 wrap_angle, and
 * their call sites modules/qcnet_map_encoder.py:121-128, qcnet_agent_encoder.py:133-139,161-176, qcnet_decoder.py:161-206):
 *   dp = pos_src[src[e]] - pos_dst[dst[e]];  out[e] = [|dp|, angle(heading_dst, dp), wrap(head_src - head_dst) (, extra[e])]
 * pos_* [n, pos_stride_*] fp32 (x, y first), head_* [n] fp32, src / dst / extra int64 [n_edges] (extra may be NULL: 3
 * columns instead of 4).  The caller guarantees the indices are in range (they come from the graph builders). */
int qcn_rel_geometry_f32(const float *pos_src, int pos_stride_src, const float *pos_dst, int pos_stride_dst,
                         const float *head_src, const float *head_dst, const int64_t *src, const int64_t *dst,
                         const int64_t *extra, int64_t n_edges, float *out, void *stream);

/* A5 — up to three MLPLayer heads (layers/mlp_layer.py:20-36: Linear(128,128) -> LayerNorm -> ReLU -> Linear(128,out_h),
 * out_h <= 128) over the same rows in one launch (the decoder's to_loc_* / to_scale_* / to_pi heads,
 * modules/qcnet_decoder.py:229-233,270-280).  x [n_rows,128]; y_h [n_rows, out_dims[h]] dense; out_dims: HOST array.
 * pack (qcn_mlp_heads_pack_floats(n_heads) floats): per head the two chain-packed matrices [mlp.0.weight, mlp.3.weight
 * with its rows zero-padded to 128], then per head [mlp.0.bias, mlp.1.weight, mlp.1.bias, mlp.3.bias zero-padded], then
 * the scale table (32 floats). */
size_t qcn_mlp_heads_pack_floats(int n_heads);
int qcn_mlp_heads_f32(const float *x, int64_t n_rows, const float *pack, int n_heads, const int *out_dims,
                      float *y0, float *y1, float *y2, int matmul_mode, void *stream);

/* K10 — recurrent half of the decoder's nn.GRU(128,128) trajectory embedder (modules/qcnet_decoder.py:86-88,257),
 * final hidden state only.  gi_r / gi_z / gi_n [n_seq * n_steps, 128]: the input half W_i* x_t + b_i* for row
 * seq * n_steps + t (one qcn_ln_linear_f32 call without LayerNorm); b_hh [384]; h0 [128] (broadcast to every sequence);
 * h_out [n_seq, 128].  whh, by matmul mode:
 *   exact: [128, 384] = weight_hh_l0^T (gate order r, z, n), fp32;
 *   split: qcn_gru_pack_floats() floats = f16 hi / lo fragments [gate][wave w][k16-step j][plane][lane][8] of
 *          W'_g[32 w + (lane & 31)][16 j + 8 (lane >> 5) + i] (W'_g = gate matrix scaled by a power of two to
 *          [2^14, 2^15)), followed by {s_r, 1/s_r, s_z, 1/s_z, s_n, 1/s_n, s_h, 1/s_h}; s_h = power of two with
 *          max(1, max|h0|) * s_h <= 2^14 (the hidden state never leaves [-max(1, max|h0|), +...]).
 *          Host packer: ops.pack_gru_whh. */
size_t qcn_gru_pack_floats(void);
int qcn_gru_f32(const float *gi_r, const float *gi_z, const float *gi_n, const float *whh,
                const float *b_hh, const float *h0, int64_t n_seq, int n_steps, float *h_out, int matmul_mode,
                void *stream);

/* The trajectory re-encoding (modules/qcnet_decoder.py:255-257) with the GRU's input halves handed over in the RECURRENCE'S
 * own layout — no counterpart in the reference (there y_emb's output, a transpose and cuDNN's GRU).
 *   qcn_fourier_gru_input_f32 = qcn_fourier_linear_f32 / qcn_fourier_folded_linear_f32 (folded != 0) with n_out = 3 on rows
 *     x[seq * n_steps + t] (x: [n_seq * n_steps, input_dim]), but a wave's tile is ONE step of 32 consecutive sequences and
 *     the outputs are blocked:  gi_g[((seq / 32) * n_steps + t) * 4096 + ((4 w + q) * 64 + lane) * 4 + i]  holds feature
 *     32 w + 8 q + 4 (lane / 32) + i of sequence 32 (seq / 32) + lane % 32 at step t — each gi_g has
 *     qcn_gru_blocked_rows(n_seq, n_steps) * 128 floats (the last sequence block is padded to 32; padding is written);
 *   qcn_gru_blocked_f32 = qcn_gru_f32 in the split-f16 arithmetic (whh = its split pack) reading that layout: the twelve
 *     loads of a step are contiguous 1 KiB pieces per wave (row-major gi: 32-byte pieces of 32 rows each), and the producer
 *     stores straight from its accumulator registers.  The input halves are the same bits as the row-major call's; the
 *     recurrence itself is the two-group kernel (two blocks of 32 sequences per workgroup half a step apart, input halves by
 *     LDS-DMA, 132 KiB of LDS per workgroup; csrc/gru.hip), whose gates round in another order than qcn_gru_f32's: the
 *     final states agree to a few 1e-7 on values in [-1, 1], not bit for bit.  Deterministic run to run. */
int64_t qcn_gru_blocked_rows(int64_t n_seq, int n_steps);
int qcn_fourier_gru_input_f32(const float *x, int input_dim, const float *pack, int64_t n_seq, int n_steps,
                              const float *wt, const float *bias, int folded, float *gi_r, float *gi_z, float *gi_n,
                              int matmul_mode, void *stream);
int qcn_gru_blocked_f32(const float *gi_r, const float *gi_z, const float *gi_n, const float *whh_pack,
                        const float *b_hh, const float *h0, int64_t n_seq, int n_steps, float *h_out, void *stream);

/* K6 — y = (x - mean) / sqrt(var + eps) per 128-wide row (LayerNorm without affine; the layer-independent half
 * of attn_prenorm_r, layers/attention_layer.py:63,82). */
int qcn_row_normalize_f32(const float *x, float *y, int64_t n_rows, float eps, void *stream);

/* G3 — edge lists that are outer products of two per-example masks: torch_geometric.utils.dense_to_sparse /
 * utils/graph.py:88-94 bipartite_dense_to_sparse of  adj[b, i, j] = mask_src[b, i] & mask_dst[b, j]  restricted to the band
 * delta_lo <= j - i <= delta_hi (pass INT32_MIN / INT32_MAX for no band).  Call sites: the temporal graph
 * (modules/qcnet_agent_encoder.py:129-132: band [1, time_span]), time -> mode and mode -> mode
 * (modules/qcnet_decoder.py:160-169,213; modules/temporal_aggregate.py:161-170,214).  Output in the reference's order
 * (row-major nonzero of the dense adjacency): src[e] = b * n + i, dst[e] = b * m + j.
 * mask_src [n_batch, n], mask_dst [n_batch, m] uint8.  Two passes like the radius search: count [n_batch * n] per source
 * node, the caller scans it (qcn_exclusive_scan_i32) and allocates E = offset[n_batch * n], fill writes src / dst [E]. */
int qcn_mask_pairs_count(const uint8_t *mask_src, const uint8_t *mask_dst, int64_t n_batch, int n, int m, int delta_lo,
                         int delta_hi, int32_t *count, void *stream);
int qcn_mask_pairs_fill(const uint8_t *mask_src, const uint8_t *mask_dst, int64_t n_batch, int n, int m, int delta_lo,
                        int delta_hi, const int64_t *offset, int64_t *src, int64_t *dst, void *stream);

/* G4 — torch_geometric.utils.coalesce(edge_index, edge_attr, reduce='max') as utils/graph.py:45-54 merge_edges calls it
 * (modules/qcnet_map_encoder.py:146): edges sorted by (src, dst), duplicates merged, the largest uint8 attribute kept.
 * 0 <= src, dst < num_nodes <= 2^27.  Two passes: qcn_coalesce_count sorts inside `workspace`
 * (qcn_coalesce_workspace_bytes(E) bytes, must stay untouched until the fill) and writes offset [E + 1] (int64; offset[E] =
 * number of distinct edges, which the caller reads to allocate); qcn_coalesce_fill writes out_src / out_dst [n_out] and
 * out_attr [n_out] (attr / out_attr may be NULL). */
size_t qcn_coalesce_workspace_bytes(int64_t num_edges);
int qcn_coalesce_count(const int64_t *src, const int64_t *dst, const uint8_t *attr, int64_t num_edges, int64_t num_nodes,
                       void *workspace, size_t workspace_bytes, int64_t *offset, void *stream);
int qcn_coalesce_fill(const void *workspace, const int64_t *offset, int64_t num_edges, int64_t num_nodes, int64_t *out_src,
                      int64_t *out_dst, uint8_t *out_attr, void *stream);

/* N3 — the state updates of every metric class of the reference in one launch (metrics/brier.py:37-52,
 * min_ade.py:36-54, min_ahe.py:37-53, min_fde.py:36-45, min_fhe.py:36-48, mr.py:38-54, prob_mr.py:38-59) after
 * valid_filter (metrics/utils.py:59-82) and the per-sample topk (metrics/utils.py:22-56).
 *   pred [n, n_modes, n_steps, pred_dim], prob [n, n_modes] or NULL (uniform), target [n, n_steps, target_dim],
 *   valid [n, n_steps] uint8 or NULL (all valid), select [n] uint8 or NULL (all; the eval mask of
 *   predictors/temp_ens_learn_agg.py:250-254 without the boolean-index copies).
 *   pos_dim: leading components compared by the displacement norms.  pred_head: component of pred holding the heading,
 *   -1 = no heading (the AHE / FHE columns are 0), -2 = the direction of the predicted motion vector with the origin as
 *   step -1 (temp_ens_learn_agg.py:255-260, fused).  target_head: component of target holding the heading.
 *   max_guesses as the classes' ctor argument (n_modes <= 8); keep_invalid_final_step as valid_filter's.
 * record [n, QCN_METRIC_RECORD] fp32: what each class adds to its `sum` for that sample, column per (class, criterion);
 * QCN_METRIC_COUNTED is 1 for a sample valid_filter keeps and the whole record 0 otherwise; columns 13-15 hold the
 * FDE-best mode, the ADE-best mode and the last valid step (diagnostics). */
#define QCN_METRIC_RECORD 16
enum {
    QCN_METRIC_BRIER_FDE = 0, QCN_METRIC_BRIER_ADE = 1, QCN_METRIC_ADE_FDE = 2, QCN_METRIC_ADE_ADE = 3,
    QCN_METRIC_AHE_FDE = 4, QCN_METRIC_AHE_ADE = 5, QCN_METRIC_FDE = 6, QCN_METRIC_FHE = 7, QCN_METRIC_MISS_FDE = 8,
    QCN_METRIC_MISS_MAXDE = 9, QCN_METRIC_PROBMISS_FDE = 10, QCN_METRIC_PROBMISS_MAXDE = 11, QCN_METRIC_COUNTED = 12
};
int qcn_motion_metrics_f32(const float *pred, int pred_dim, const float *prob, const float *target, int target_dim,
                           const uint8_t *valid, const uint8_t *select, int64_t n, int n_modes, int n_steps, int pos_dim,
                           int pred_head, int target_head, int max_guesses, int keep_invalid_final_step,
                           float miss_threshold, float *record, void *stream);

/* N3 — regression targets in the agent frame (predictors/temp_ens_learn_agg.py:104-121 get_ground_truth):
 * out[a, t - first_step] = [R(-theta_a) (xy[a, t] - xy[a, origin_step]), z[a, t] - z[a, origin_step] (0 for pos_dim 2),
 * wrap_angle(heading[a, t] - theta_a)], theta_a = heading[a, origin_step], for t in [first_step, first_step + n_steps).
 * position [n_agents, n_total_steps, pos_dim] (pos_dim 2 or 3), heading [n_agents, n_total_steps], out [n_agents, n_steps, 4]
 * (16-byte aligned). */
int qcn_to_agent_frame_f32(const float *position, const float *heading, int64_t n_agents, int n_total_steps, int pos_dim,
                           int origin_step, int first_step, int n_steps, float *out, void *stream);

/* N3 — predicted trajectories back in the world frame (predictors/qcnet_sliding.py:179-192, visualize.py:17-30 postprocess):
 * out[i, j] = traj[rows[i], j, :2] [[cos, sin], [-sin, cos]] + xy[rows[i], origin_step] for the per_agent = modes * steps
 * points j of an agent.  traj [n_agents, per_agent, traj_dim]; rows [n_rows] int64 agent ids (the eval mask's indices) or
 * NULL (rows = 0..n_rows-1); out [n_rows, per_agent, 2] (8-byte aligned). */
int qcn_to_world_frame_f32(const float *traj, int traj_dim, const float *position, const float *heading,
                           const int64_t *rows, int64_t n_rows, int per_agent, int n_total_steps, int pos_dim,
                           int origin_step, float *out, void *stream);

/* N3 — the NMS ensembling baseline of the sliding-window evaluation (utils/nms.py:11-38 nms, scored as
 * predictors/qcnet_sliding.py:48-54 calculate_metrics inside the loop :294-317), every (agent, window size) at once.
 *   traj [n_agents, n_cand_total, n_steps, 2] world-frame candidates (window-major: the reference stacks the frames
 *   time_shift = 10 .. 1, each with cand_per_window modes), conf [n_agents, n_cand_total], gt [n_agents, n_steps, 2].
 *   Window w (0-based) selects among the first (w + 1) * cand_per_window candidates; n_cand_total <= 64,
 *   n_select <= cand_per_window (the reference's top-up loop needs as many candidates as selections).
 * selected [n_agents, n_windows, n_select] int32: candidate indices in selection order; score [n_agents, n_windows, 3]:
 * min FDE, min ADE, miss (min FDE > miss_threshold).  Equal confidences are taken higher index first. */
int qcn_nms_select_f32(const float *traj, const float *conf, const float *gt, int64_t n_agents, int n_cand_total,
                       int n_steps, int n_windows, int cand_per_window, int n_select, float radius, float miss_threshold,
                       int *selected, float *score, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QCNET_HIP_H */
