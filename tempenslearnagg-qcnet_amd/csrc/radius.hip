// G1/G2 radius search (see include/qcnet_hip.h).  Integer/compare work, HBM-light: the output write (16 B per pair) is the
// only traffic that scales.  (Round 1 ran one THREAD per y row with a linear scan and per-lane scattered stores — the
// upstream device kernel's shape; SURVEY.md 2b K1 asks for the LDS-tiled / ballot-compacted form below.)
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

// One WAVE per y row: the 64 lanes test 64 consecutive x rows per round (one coalesced 512-B read of x instead of 64
// wave-uniform scalar-like reads), the hits of a round are a ballot, their order is the lane order (= ascending x, the
// canonical order), the neighbour cap is a popcount on the ballot and the emitted pairs are compacted by a prefix popcount,
// so a round's stores are contiguous.  A block (4 waves, 4 rows each = 16 consecutive y rows) whose rows all lie in one
// example — all blocks but the ones straddling an example boundary — stages the example's x rows through LDS in tiles of
// kTile points, shared by its 16 rows; a straddling block reads x from global memory, row by row.
constexpr int kRadiusWaves = 4, kRadiusRowsPerWave = 4, kRadiusRowsPerBlock = kRadiusWaves * kRadiusRowsPerWave;
constexpr int kRadiusTile = 1024;

template <int DIM>
__device__ __forceinline__ float dist2(const float (&xv)[DIM], const float (&yv)[DIM])
{
    const float d0 = xv[0] - yv[0];
    float dist = __fmul_rn(d0, d0);
#pragma unroll
    for (int d = 1; d < DIM; ++d) {
        const float dd = xv[d] - yv[d];
        dist = __fmaf_rn(dd, dd, dist);
    }
    return dist;
}

// per-row scan state (wave-uniform)
struct RowScan {
    int hits, kept;
    bool done;
};

// one round: lane tests x row `nx` (valid lanes only); returns updated state, writes pairs when FILL
template <bool FILL>
__device__ __forceinline__ void radius_round(RowScan &st, bool hit, bool emit_ok, int64_t ny, int64_t nx, int max_nbr,
                                             int64_t out0, int64_t *__restrict__ row, int64_t *__restrict__ col, int lane)
{
    const unsigned long long hit_mask = __ballot(hit);
    if (hit_mask == 0) return;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int room = max_nbr - st.hits;                           // hits still admitted (>= 1 here)
    const int rank_hit = __popcll(hit_mask & below);              // this lane's position among the round's hits
    const bool counted = hit && rank_hit < room;
    const unsigned long long emit_mask = __ballot(counted && emit_ok);
    if (FILL && counted && emit_ok) {
        const int64_t o = out0 + st.kept + __popcll(emit_mask & below);
        row[o] = ny;
        col[o] = nx;
    }
    const int n_hits = __popcll(hit_mask);
    st.kept += __popcll(emit_mask);
    st.hits += n_hits < room ? n_hits : room;
    if (st.hits >= max_nbr) st.done = true;
}

template <int DIM, bool FILL>
__global__ __launch_bounds__(64 * kRadiusWaves) void radius_kernel(
    const float *__restrict__ x, const float *__restrict__ y, const int64_t *__restrict__ ptr_x,
    const int64_t *__restrict__ ptr_y, int64_t num_examples, int64_t n_y, float r2, int max_nbr, int drop_self,
    const uint8_t *__restrict__ mask_x, const uint8_t *__restrict__ mask_y, const int64_t *__restrict__ offset,
    int32_t *__restrict__ count, int64_t *__restrict__ row, int64_t *__restrict__ col)
{
    __shared__ float xs[kRadiusTile * DIM];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * kRadiusRowsPerBlock;
    const int64_t row_last = (row0 + kRadiusRowsPerBlock < n_y ? row0 + kRadiusRowsPerBlock : n_y) - 1;
    const int64_t b_first = find_example(ptr_y, num_examples, row0);
    const bool one_example = ptr_y[b_first + 1] > row_last;       // block-uniform

    // this wave's rows
    int64_t ny[kRadiusRowsPerWave], out0[kRadiusRowsPerWave];
    float yv[kRadiusRowsPerWave][DIM];
    bool keep_y[kRadiusRowsPerWave];
    RowScan st[kRadiusRowsPerWave];
#pragma unroll
    for (int i = 0; i < kRadiusRowsPerWave; ++i) {
        ny[i] = row0 + wave * kRadiusRowsPerWave + i;
        const bool live = ny[i] < n_y;
        const int64_t nyc = live ? ny[i] : n_y - 1;
#pragma unroll
        for (int d = 0; d < DIM; ++d) yv[i][d] = y[nyc * DIM + d];
        keep_y[i] = mask_y == nullptr || mask_y[nyc] != 0;
        out0[i] = FILL ? offset[nyc] : 0;
        st[i] = RowScan{0, 0, !live};
    }

    if (one_example) {
        const int64_t x_beg = ptr_x[b_first], x_end = ptr_x[b_first + 1];
        for (int64_t t0 = x_beg; t0 < x_end; t0 += kRadiusTile) {
            const int n_tile = (int)(x_end - t0 < kRadiusTile ? x_end - t0 : kRadiusTile);
            __syncthreads();                                       // the previous tile is no longer read
            for (int j = threadIdx.x; j < n_tile * DIM; j += 64 * kRadiusWaves) xs[j] = x[t0 * DIM + j];
            __syncthreads();
            bool all_done = true;
#pragma unroll
            for (int i = 0; i < kRadiusRowsPerWave; ++i) {
                for (int j0 = 0; j0 < n_tile && !st[i].done; j0 += 64) {
                    const int j = j0 + lane;
                    const bool valid = j < n_tile;
                    float xv[DIM];
#pragma unroll
                    for (int d = 0; d < DIM; ++d) xv[d] = xs[(valid ? j : 0) * DIM + d];
                    const int64_t nx = t0 + j;
                    const bool hit = valid && dist2<DIM>(xv, yv[i]) < r2;
                    const bool emit_ok = keep_y[i] && !(drop_self && nx == ny[i]) && (mask_x == nullptr || !hit || mask_x[nx] != 0);
                    radius_round<FILL>(st[i], hit, emit_ok, ny[i], nx, max_nbr, out0[i], row, col, lane);
                }
                all_done = all_done && st[i].done;
            }
            if (__syncthreads_and(all_done)) break;                // every row of the block has reached its cap
        }
    } else {
#pragma unroll
        for (int i = 0; i < kRadiusRowsPerWave; ++i) {
            if (st[i].done) continue;
            const int64_t b = find_example(ptr_y, num_examples, ny[i]);
            const int64_t x_beg = ptr_x[b], x_end = ptr_x[b + 1];
            for (int64_t j0 = x_beg; j0 < x_end && !st[i].done; j0 += 64) {
                const int64_t nx = j0 + lane;
                const bool valid = nx < x_end;
                float xv[DIM];
#pragma unroll
                for (int d = 0; d < DIM; ++d) xv[d] = x[(valid ? nx : x_beg) * DIM + d];
                const bool hit = valid && dist2<DIM>(xv, yv[i]) < r2;
                const bool emit_ok = keep_y[i] && !(drop_self && nx == ny[i]) && (mask_x == nullptr || !hit || mask_x[nx] != 0);
                radius_round<FILL>(st[i], hit, emit_ok, ny[i], nx, max_nbr, out0[i], row, col, lane);
            }
        }
    }
    if (!FILL && lane == 0) {
#pragma unroll
        for (int i = 0; i < kRadiusRowsPerWave; ++i)
            if (ny[i] < n_y) count[ny[i]] = st[i].kept;
    }
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
    const unsigned grid = (unsigned)((n_y + kRadiusRowsPerBlock - 1) / kRadiusRowsPerBlock);
    if (dim == 2)
        hipLaunchKernelGGL((radius_kernel<2, FILL>), dim3(grid), dim3(64 * kRadiusWaves), 0, qcn_stream(stream), x, y, ptr_x, ptr_y,
                           num_examples, n_y, r2, max_nbr, drop_self, mask_x, mask_y, offset, count, row, col);
    else
        hipLaunchKernelGGL((radius_kernel<3, FILL>), dim3(grid), dim3(64 * kRadiusWaves), 0, qcn_stream(stream), x, y, ptr_x, ptr_y,
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
