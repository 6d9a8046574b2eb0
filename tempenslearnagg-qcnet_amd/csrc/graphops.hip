// G3 / G4 — the mask enumerations and the edge merge of the graph builders (see include/qcnet_hip.h).
//
// mask_pairs: torch_geometric.utils.dense_to_sparse / utils/graph.py:88-94 bipartite_dense_to_sparse applied to an outer
// product of two per-example masks, optionally restricted to a band dst - src in [delta_lo, delta_hi] — the three graphs the
// modules build that way (temporal: modules/qcnet_agent_encoder.py:129-132; time->mode and mode->mode:
// modules/qcnet_decoder.py:160-169,213) without the dense [B, n, m] adjacency, its nonzero() and the two filter passes.
// One thread per source node counts / writes its run of destinations, so the output is in the reference's row-major order.
//
// coalesce_max: torch_geometric.utils.coalesce(reduce='max') as utils/graph.py:45-54 merge_edges uses it
// (modules/qcnet_map_encoder.py:146): sort by (src, dst), merge duplicates, keep the largest attribute.  The attribute
// rides in the low byte of the sort key, so the last element of every run of equal (src, dst) carries the maximum.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hip.h"

namespace {

inline size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

template <bool kFill>
__global__ __launch_bounds__(256) void mask_pairs_kernel(const uint8_t *__restrict__ mask_src, const uint8_t *__restrict__ mask_dst,
                                                         int64_t n_rows, int n, int m, int delta_lo, int delta_hi,
                                                         int32_t *__restrict__ count, const int64_t *__restrict__ offset,
                                                         int64_t *__restrict__ src, int64_t *__restrict__ dst)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // row = b * n + i
    if (row >= n_rows) return;
    const int64_t b = row / n;
    const int i = (int)(row - b * n);
    int c = 0;
    if (mask_src[row]) {
        // j - i in [delta_lo, delta_hi], clamped to the destination range
        const int64_t lo64 = (int64_t)i + delta_lo, hi64 = (int64_t)i + delta_hi;
        const int lo = lo64 < 0 ? 0 : (lo64 > m ? m : (int)lo64);
        const int hi = hi64 >= m ? m - 1 : (int)hi64;
        const uint8_t *md = mask_dst + b * m;
        int64_t at = kFill ? offset[row] : 0;
        for (int j = lo; j <= hi; ++j) {
            if (md[j]) {
                if (kFill) {
                    src[at] = row;
                    dst[at] = b * m + j;
                    ++at;
                }
                ++c;
            }
        }
    }
    if (!kFill) count[row] = c;
}

__global__ __launch_bounds__(256) void coalesce_keys_kernel(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                            const uint8_t *__restrict__ attr, int64_t E, int64_t num_nodes,
                                                            uint64_t *__restrict__ key)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    key[e] = ((uint64_t)(src[e] * num_nodes + dst[e]) << 8) | (attr ? attr[e] : 0);
}

// flag[e] = 1 when sorted element e is the LAST of its (src, dst) run (it carries the run's largest attribute)
__global__ __launch_bounds__(256) void coalesce_flag_kernel(const uint64_t *__restrict__ key, int64_t E, int32_t *__restrict__ flag)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > E) return;
    flag[e] = e < E && (e == E - 1 || (key[e] >> 8) != (key[e + 1] >> 8)) ? 1 : 0;
}

__global__ __launch_bounds__(256) void coalesce_fill_kernel(const uint64_t *__restrict__ key, const int64_t *__restrict__ offset,
                                                            int64_t E, int64_t num_nodes, int64_t *__restrict__ out_src,
                                                            int64_t *__restrict__ out_dst, uint8_t *__restrict__ out_attr)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E || offset[e + 1] == offset[e]) return;
    const int64_t at = offset[e];
    const uint64_t k = key[e] >> 8;
    out_src[at] = (int64_t)(k / (uint64_t)num_nodes);
    out_dst[at] = (int64_t)(k % (uint64_t)num_nodes);
    if (out_attr) out_attr[at] = (uint8_t)(key[e] & 0xff);
}

struct CastI32ToI64 {
    __device__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

unsigned coalesce_end_bit(int64_t num_nodes)
{
    unsigned bits = 1;
    while (bits < 27 && ((int64_t)1 << bits) < num_nodes) ++bits;
    return 8 + 2 * bits;
}

size_t coalesce_sort_bytes(int64_t E, unsigned end_bit)
{
    size_t bytes = 0;
    rocprim::radix_sort_keys(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)E, 0u, end_bit, (hipStream_t)0);
    return bytes;
}

size_t coalesce_scan_bytes(int64_t E)
{
    size_t bytes = 0;
    auto in = rocprim::make_transform_iterator((const int32_t *)nullptr, CastI32ToI64());
    rocprim::exclusive_scan(nullptr, bytes, in, (int64_t *)nullptr, (int64_t)0, (size_t)(E + 1), rocprim::plus<int64_t>(),
                            (hipStream_t)0);
    return bytes;
}

}  // namespace

static int mask_pairs_check(const uint8_t *mask_src, const uint8_t *mask_dst, int64_t n_batch, int n, int m, int delta_lo,
                            int delta_hi)
{
    if (n_batch < 0 || n < 0 || m < 0 || delta_lo > delta_hi) return QCN_ERR_INVALID_ARGUMENT;
    if (n_batch * (int64_t)n > 0 && (!mask_src || (m > 0 && !mask_dst))) return QCN_ERR_INVALID_ARGUMENT;
    return QCN_OK;
}

extern "C" int qcn_mask_pairs_count(const uint8_t *mask_src, const uint8_t *mask_dst, int64_t n_batch, int n, int m, int delta_lo,
                                    int delta_hi, int32_t *count, void *stream)
{
    const int st = mask_pairs_check(mask_src, mask_dst, n_batch, n, m, delta_lo, delta_hi);
    if (st != QCN_OK) return st;
    const int64_t n_rows = n_batch * n;
    if (n_rows == 0) return QCN_OK;
    if (!count) return QCN_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(mask_pairs_kernel<false>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, qcn_stream(stream), mask_src,
                       mask_dst, n_rows, n, m, delta_lo, delta_hi, count, (const int64_t *)nullptr, (int64_t *)nullptr,
                       (int64_t *)nullptr);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

extern "C" int qcn_mask_pairs_fill(const uint8_t *mask_src, const uint8_t *mask_dst, int64_t n_batch, int n, int m, int delta_lo,
                                   int delta_hi, const int64_t *offset, int64_t *src, int64_t *dst, void *stream)
{
    const int st = mask_pairs_check(mask_src, mask_dst, n_batch, n, m, delta_lo, delta_hi);
    if (st != QCN_OK) return st;
    const int64_t n_rows = n_batch * n;
    if (n_rows == 0) return QCN_OK;
    if (!offset || !src || !dst) return QCN_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(mask_pairs_kernel<true>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, qcn_stream(stream), mask_src,
                       mask_dst, n_rows, n, m, delta_lo, delta_hi, (int32_t *)nullptr, offset, src, dst);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

extern "C" size_t qcn_coalesce_workspace_bytes(int64_t num_edges)
{
    if (num_edges <= 0) return 256;
    const size_t keys = align256((size_t)num_edges * sizeof(uint64_t));
    const size_t flags = align256((size_t)(num_edges + 1) * sizeof(int32_t));
    const size_t sort = coalesce_sort_bytes(num_edges, 62), scan = coalesce_scan_bytes(num_edges);
    return 2 * keys + flags + align256(sort > scan ? sort : scan) + 256;
}

extern "C" int qcn_coalesce_count(const int64_t *src, const int64_t *dst, const uint8_t *attr, int64_t num_edges, int64_t num_nodes,
                                  void *workspace, size_t workspace_bytes, int64_t *offset, void *stream)
{
    if (num_edges < 0 || num_nodes < 1 || !offset) return QCN_ERR_INVALID_ARGUMENT;
    if (num_nodes > ((int64_t)1 << 27) || num_edges >= ((int64_t)1 << 31)) return QCN_ERR_UNSUPPORTED;
    hipStream_t s = qcn_stream(stream);
    if (num_edges == 0) return hipMemsetAsync(offset, 0, sizeof(int64_t), s) == hipSuccess ? QCN_OK : QCN_ERR_LAUNCH_FAILED;
    if (!src || !dst || !workspace) return QCN_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < qcn_coalesce_workspace_bytes(num_edges)) return QCN_ERR_WORKSPACE_TOO_SMALL;
    const size_t keys = align256((size_t)num_edges * sizeof(uint64_t));
    const size_t flags = align256((size_t)(num_edges + 1) * sizeof(int32_t));
    char *ws = (char *)workspace;
    uint64_t *key_sorted = (uint64_t *)ws;                       // qcn_coalesce_fill reads this slot
    uint64_t *key_in = (uint64_t *)(ws + keys);
    int32_t *flag = (int32_t *)(ws + 2 * keys);
    void *temp = ws + 2 * keys + flags;
    const unsigned grid = (unsigned)((num_edges + 1 + 255) / 256);
    hipLaunchKernelGGL(coalesce_keys_kernel, dim3(grid), dim3(256), 0, s, src, dst, attr, num_edges, num_nodes, key_in);
    QCN_CHECK_LAUNCH();
    const unsigned end_bit = coalesce_end_bit(num_nodes);
    size_t sort_bytes = coalesce_sort_bytes(num_edges, end_bit);
    if (rocprim::radix_sort_keys(temp, sort_bytes, key_in, key_sorted, (size_t)num_edges, 0u, end_bit, s) != hipSuccess)
        return QCN_ERR_LAUNCH_FAILED;
    hipLaunchKernelGGL(coalesce_flag_kernel, dim3(grid), dim3(256), 0, s, key_sorted, num_edges, flag);
    QCN_CHECK_LAUNCH();
    size_t scan_bytes = coalesce_scan_bytes(num_edges);
    auto in = rocprim::make_transform_iterator((const int32_t *)flag, CastI32ToI64());
    if (rocprim::exclusive_scan(temp, scan_bytes, in, offset, (int64_t)0, (size_t)(num_edges + 1), rocprim::plus<int64_t>(), s) !=
        hipSuccess)
        return QCN_ERR_LAUNCH_FAILED;
    return QCN_OK;
}

extern "C" int qcn_coalesce_fill(const void *workspace, const int64_t *offset, int64_t num_edges, int64_t num_nodes, int64_t *out_src,
                                 int64_t *out_dst, uint8_t *out_attr, void *stream)
{
    if (num_edges < 0 || num_nodes < 1) return QCN_ERR_INVALID_ARGUMENT;
    if (num_edges == 0) return QCN_OK;
    if (!workspace || !offset || !out_src || !out_dst) return QCN_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(coalesce_fill_kernel, dim3((unsigned)((num_edges + 255) / 256)), dim3(256), 0, qcn_stream(stream),
                       (const uint64_t *)workspace, offset, num_edges, num_nodes, out_src, out_dst, out_attr);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}
