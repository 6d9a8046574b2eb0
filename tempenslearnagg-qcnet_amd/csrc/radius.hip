// G1/G2 radius search (see include/qcnet_hip.h).  One thread per y row; all threads of a wave usually sit
// in the same example, so the x reads are wave-uniform broadcasts served from L1/L2.  Integer/compare work,
// HBM-light: the output write (16 B per pair) is the only traffic that scales.
#include "common.hip.h"

namespace {

__device__ __forceinline__ int64_t find_example(const int64_t *ptr, int64_t num_examples, int64_t n)
{
    // largest b with ptr[b] <= n  (ptr is non-decreasing, ptr[num_examples] > n)
    int64_t lo = 0, hi = num_examples;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (ptr[mid] <= n) lo = mid; else hi = mid;
    }
    return lo;
}

template <int DIM, bool FILL>
__global__ __launch_bounds__(256) void radius_kernel(
    const float *__restrict__ x, const float *__restrict__ y, const int64_t *__restrict__ ptr_x,
    const int64_t *__restrict__ ptr_y, int64_t num_examples, int64_t n_y, float r2, int max_nbr, int drop_self,
    const uint8_t *__restrict__ mask_x, const uint8_t *__restrict__ mask_y, const int64_t *__restrict__ offset,
    int32_t *__restrict__ count, int64_t *__restrict__ row, int64_t *__restrict__ col)
{
    const int64_t ny = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ny >= n_y) return;
    const int64_t b = find_example(ptr_y, num_examples, ny);
    const int64_t x_beg = ptr_x[b], x_end = ptr_x[b + 1];
    const bool keep_y = mask_y == nullptr || mask_y[ny] != 0;
    float yv[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) yv[d] = y[ny * DIM + d];
    int hits = 0, kept = 0;
    int64_t out = FILL ? offset[ny] : 0;
    for (int64_t nx = x_beg; nx < x_end; ++nx) {
        const float d0 = x[nx * DIM] - yv[0];
        float dist = __fmul_rn(d0, d0);
#pragma unroll
        for (int d = 1; d < DIM; ++d) {
            const float dd = x[nx * DIM + d] - yv[d];
            dist = __fmaf_rn(dd, dd, dist);
        }
        if (dist < r2) {
            const bool emit = keep_y && !(drop_self && nx == ny) && (mask_x == nullptr || mask_x[nx] != 0);
            if (emit) {
                if (FILL) { row[out] = ny; col[out] = nx; ++out; }
                ++kept;
            }
            if (++hits >= max_nbr) break;
        }
    }
    if (!FILL) count[ny] = kept;
}

template <bool FILL>
int launch(const float *x, const float *y, const int64_t *ptr_x, const int64_t *ptr_y, int64_t num_examples,
           int64_t n_y, int dim, float r2, int max_nbr, int drop_self, const uint8_t *mask_x,
           const uint8_t *mask_y, const int64_t *offset, int32_t *count, int64_t *row, int64_t *col, void *stream)
{
    if (n_y < 0 || num_examples < 1 || max_nbr < 1) return QCN_ERR_INVALID_ARGUMENT;
    if (dim != 2 && dim != 3) return QCN_ERR_UNSUPPORTED;
    if (n_y == 0) return QCN_OK;
    if (!x || !y || !ptr_x || !ptr_y) return QCN_ERR_INVALID_ARGUMENT;
    if (FILL ? (!offset || !row || !col) : !count) return QCN_ERR_INVALID_ARGUMENT;
    const unsigned grid = (unsigned)((n_y + 255) / 256);
    if (dim == 2)
        hipLaunchKernelGGL((radius_kernel<2, FILL>), dim3(grid), dim3(256), 0, qcn_stream(stream), x, y, ptr_x, ptr_y,
                           num_examples, n_y, r2, max_nbr, drop_self, mask_x, mask_y, offset, count, row, col);
    else
        hipLaunchKernelGGL((radius_kernel<3, FILL>), dim3(grid), dim3(256), 0, qcn_stream(stream), x, y, ptr_x, ptr_y,
                           num_examples, n_y, r2, max_nbr, drop_self, mask_x, mask_y, offset, count, row, col);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

}  // namespace

extern "C" int qcn_radius_count(const float *x, const float *y, const int64_t *ptr_x, const int64_t *ptr_y,
                                int64_t num_examples, int64_t n_y, int dim, float r2, int max_num_neighbors,
                                int drop_self, const uint8_t *mask_x, const uint8_t *mask_y, int32_t *count,
                                void *stream)
{
    return launch<false>(x, y, ptr_x, ptr_y, num_examples, n_y, dim, r2, max_num_neighbors, drop_self, mask_x,
                         mask_y, nullptr, count, nullptr, nullptr, stream);
}

extern "C" int qcn_radius_fill(const float *x, const float *y, const int64_t *ptr_x, const int64_t *ptr_y,
                               int64_t num_examples, int64_t n_y, int dim, float r2, int max_num_neighbors,
                               int drop_self, const uint8_t *mask_x, const uint8_t *mask_y, const int64_t *offset,
                               int64_t *row, int64_t *col, void *stream)
{
    return launch<true>(x, y, ptr_x, ptr_y, num_examples, n_y, dim, r2, max_num_neighbors, drop_self, mask_x,
                        mask_y, offset, nullptr, row, col, stream);
}
