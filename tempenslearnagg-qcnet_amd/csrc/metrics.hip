// N3 — evaluation metrics and frame transforms of the validation / post-processing step.
//
// motion_metrics_kernel: every torchmetrics state update of the reference's metrics package in one launch — Brier
// (metrics/brier.py:37-52), minADE (metrics/min_ade.py:36-54), minAHE (metrics/min_ahe.py:37-53), minFDE
// (metrics/min_fde.py:36-45), minFHE (metrics/min_fhe.py:36-48), MR (metrics/mr.py:38-54) and ProbMR
// (metrics/prob_mr.py:38-59), each with both of its criteria (FDE / ADE, FDE / MAXDE), after the reference's
// valid_filter (metrics/utils.py:59-82) and topk (metrics/utils.py:22-56, the per-sample branch).  One wave per sample;
// lanes run over the future steps.  The per-sample record is what the classes add to their `sum` state; a record with
// counted == 0 is a sample valid_filter drops.
//
// agent_frame_kernel: the two rigid transforms around the predictor — world -> agent frame of the regression targets
// (predictors/temp_ens_learn_agg.py:104-121 get_ground_truth) and agent -> world frame of the predicted trajectories
// (predictors/qcnet_sliding.py:179-197, visualize.py:17-35 postprocess).
#include "common.hip.h"

namespace {

constexpr int kMaxModes = 8;
constexpr float kPi = 3.14159265358979323846f;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

// utils/geometry.py:47-51: -pi + (a + pi) mod 2 pi with torch's floor-mod (result takes the divisor's sign)
__device__ __forceinline__ float wrap_angle(float a)
{
    const float two_pi = 2.0f * kPi;
    float m = fmodf(a + kPi, two_pi);
    if (m < 0.0f) m += two_pi;
    return -kPi + m;
}

struct MetricsArgs {
    const float *pred;        // [n, n_modes, n_steps, pred_dim]
    const float *prob;        // [n, n_modes] or null (uniform over the kept modes)
    const float *target;      // [n, n_steps, target_dim]
    const uint8_t *valid;     // [n, n_steps] or null (all valid)
    const uint8_t *select;    // [n] or null
    float *record;            // [n, QCN_METRIC_RECORD]
    int64_t n;
    int n_modes, n_steps, pred_dim, target_dim, pos_dim, pred_head, target_head, max_guesses, keep_invalid_final_step;
    float miss_threshold;
};

__device__ __forceinline__ float point_distance(const float *a, const float *b, int pos_dim)
{
    float s = 0.f;
    for (int c = 0; c < pos_dim; ++c) {
        const float d = a[c] - b[c];
        s += d * d;
    }
    return sqrtf(s);
}

__global__ __launch_bounds__(256) void motion_metrics_kernel(MetricsArgs g)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= g.n) return;
    float *rec = g.record + i * QCN_METRIC_RECORD;
    const int T = g.n_steps;
    const uint8_t *vm = g.valid ? g.valid + i * T : nullptr;
    // last valid step (argmax of valid * (t + 1), metrics/min_fde.py:41) and the number of valid steps
    int last = -1;
    float n_valid = 0.f;
    for (int t = lane; t < T; t += 64)
        if (!vm || vm[t]) { last = max(last, t); n_valid += 1.f; }
    last = wave_max_i(last);
    n_valid = wave_sum(n_valid);
    const bool kept = g.keep_invalid_final_step ? last >= 0 : last == T - 1;      // metrics/utils.py:69-72
    if ((g.select && !g.select[i]) || !kept) {
        if (lane < QCN_METRIC_RECORD) rec[lane] = 0.f;
        return;
    }
    // topk (metrics/utils.py:28-56): the kept modes in the order the reference's tensors hold them — all modes in place,
    // the max_guesses most probable in descending probability (ties: lower index first), or the first max_guesses.
    const int G = min(g.max_guesses, g.n_modes);
    int order[kMaxModes];
    float p[kMaxModes];
    float psum = 0.f;
    {
        unsigned taken = 0;
        for (int r = 0; r < G; ++r) {
            int pick = r;
            if (g.prob && G < g.n_modes) {
                pick = -1;
                float best = 0.f;
                for (int k = 0; k < g.n_modes; ++k) {
                    const float v = g.prob[i * g.n_modes + k];
                    if (!((taken >> k) & 1u) && (pick < 0 || v > best)) { pick = k; best = v; }
                }
                taken |= 1u << pick;
            }
            order[r] = pick;
            p[r] = g.prob ? g.prob[i * g.n_modes + pick] : 1.0f;
            psum += p[r];
        }
    }
    const float *tg = g.target + i * T * g.target_dim;
    float sum_d[kMaxModes], max_d[kMaxModes], fde[kMaxModes];
    for (int r = 0; r < G; ++r) {
        const float *pm = g.pred + ((i * g.n_modes + order[r]) * T) * g.pred_dim;
        float s = 0.f, m = 0.f;                          // masked distances are 0, so the running max starts there too
        for (int t = lane; t < T; t += 64) {
            const float d = (!vm || vm[t]) ? point_distance(pm + t * g.pred_dim, tg + t * g.target_dim, g.pos_dim) : 0.f;
            s += d;
            m = fmaxf(m, d);
        }
        sum_d[r] = wave_sum(s);
        max_d[r] = wave_max(m);
        fde[r] = point_distance(pm + last * g.pred_dim, tg + last * g.target_dim, g.pos_dim);
    }
    int b_fde = 0, b_ade = 0, b_max = 0;                 // first minimum, as torch.argmin / min
    for (int r = 1; r < G; ++r) {
        if (fde[r] < fde[b_fde]) b_fde = r;
        if (sum_d[r] < sum_d[b_ade]) b_ade = r;
        if (max_d[r] < max_d[b_max]) b_max = r;
    }
    // heading errors of the best mode under either criterion (metrics/min_ahe.py:50-52, min_fhe.py:45-47); the predicted
    // heading is a component of pred, or (pred_head == -2) the direction of the predicted motion vector with the origin
    // as step -1 (predictors/temp_ens_learn_agg.py:255-260)
    float ahe[2] = {0.f, 0.f}, fhe = 0.f;
    if (g.pred_head != -1) {
        for (int c = 0; c < 2; ++c) {
            const int r = c == 0 ? b_fde : b_ade;
            if (c == 1 && b_ade == b_fde) { ahe[1] = ahe[0]; break; }
            const float *pm = g.pred + ((i * g.n_modes + order[r]) * T) * g.pred_dim;
            float a = 0.f, f = 0.f;
            for (int t = lane; t < T; t += 64) {
                float head;
                if (g.pred_head >= 0) {
                    head = pm[t * g.pred_dim + g.pred_head];
                } else {
                    const float px = t > 0 ? pm[(t - 1) * g.pred_dim] : 0.f, py = t > 0 ? pm[(t - 1) * g.pred_dim + 1] : 0.f;
                    head = atan2f(pm[t * g.pred_dim + 1] - py, pm[t * g.pred_dim] - px);
                }
                const float err = fabsf(wrap_angle(head - tg[t * g.target_dim + g.target_head]));
                a += (!vm || vm[t]) ? err : 0.f;
                if (t == last) f = err;
            }
            ahe[c] = wave_sum(a);
            if (c == 0) fhe = wave_sum(f);
        }
    }
    if (lane == 0) {
        const float p_fde = p[b_fde] / psum, p_ade = p[b_ade] / psum, p_max = p[b_max] / psum;
        const bool miss_fde = fde[b_fde] > g.miss_threshold, miss_max = max_d[b_max] > g.miss_threshold;
        rec[QCN_METRIC_BRIER_FDE] = (1.0f - p_fde) * (1.0f - p_fde);
        rec[QCN_METRIC_BRIER_ADE] = (1.0f - p_ade) * (1.0f - p_ade);
        rec[QCN_METRIC_ADE_FDE] = sum_d[b_fde] / n_valid;
        rec[QCN_METRIC_ADE_ADE] = sum_d[b_ade] / n_valid;
        rec[QCN_METRIC_AHE_FDE] = ahe[0] / n_valid;
        rec[QCN_METRIC_AHE_ADE] = ahe[1] / n_valid;
        rec[QCN_METRIC_FDE] = fde[b_fde];
        rec[QCN_METRIC_FHE] = fhe;
        rec[QCN_METRIC_MISS_FDE] = miss_fde ? 1.f : 0.f;
        rec[QCN_METRIC_MISS_MAXDE] = miss_max ? 1.f : 0.f;
        rec[QCN_METRIC_PROBMISS_FDE] = miss_fde ? 1.f : 1.0f - p_fde;
        rec[QCN_METRIC_PROBMISS_MAXDE] = miss_max ? 1.f : 1.0f - p_max;
        rec[QCN_METRIC_COUNTED] = 1.f;
        rec[13] = (float)order[b_fde];
        rec[14] = (float)order[b_ade];
        rec[15] = (float)last;
    }
}

// world -> agent frame (direction 0): out[a, t] = [R(-theta_a) (xy[a, t] - origin_a), (z[a, t] - z_a), wrap(head[a, t] - theta_a)]
// as predictors/temp_ens_learn_agg.py:104-121 (bmm with [[cos, -sin], [sin, cos]] on row vectors).
__global__ __launch_bounds__(256) void to_agent_frame_kernel(const float *__restrict__ position, const float *__restrict__ heading,
                                                             int64_t n_agents, int n_total_steps, int pos_dim, int origin_step,
                                                             int first_step, int n_steps, float *__restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_agents * n_steps) return;
    const int64_t a = idx / n_steps;
    const int t = (int)(idx - a * n_steps) + first_step;
    const float *po = position + (a * n_total_steps + origin_step) * pos_dim;
    const float *pt = position + (a * n_total_steps + t) * pos_dim;
    const float theta = heading[a * n_total_steps + origin_step];
    const float c = cosf(theta), s = sinf(theta);
    const float dx = pt[0] - po[0], dy = pt[1] - po[1];
    float4 r;
    r.x = dx * c + dy * s;                               // row vector times [[c, -s], [s, c]]
    r.y = dx * -s + dy * c;
    r.z = pos_dim == 3 ? pt[2] - po[2] : 0.f;
    r.w = wrap_angle(heading[a * n_total_steps + t] - theta);
    reinterpret_cast<float4 *>(out)[idx] = r;
}

// agent -> world frame (direction 1): out[i, k, t] = traj[i, k, t, :2] [[cos, sin], [-sin, cos]] + origin_i
// (predictors/qcnet_sliding.py:183-192); `rows` selects the agents (the eval mask's indices) or is null.
__global__ __launch_bounds__(256) void to_world_frame_kernel(const float *__restrict__ traj, int traj_dim, const float *__restrict__ position,
                                                             const float *__restrict__ heading, const int64_t *__restrict__ rows,
                                                             int64_t n_rows, int per_agent, int n_total_steps, int pos_dim,
                                                             int origin_step, float *__restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_rows * per_agent) return;
    const int64_t i = idx / per_agent;
    const int64_t a = rows ? rows[i] : i;
    const float *po = position + (a * n_total_steps + origin_step) * pos_dim;
    const float theta = heading[a * n_total_steps + origin_step];
    const float c = cosf(theta), s = sinf(theta);
    const float *p = traj + (a * per_agent + (idx - i * per_agent)) * traj_dim;
    float2 r;
    r.x = p[0] * c + p[1] * -s + po[0];
    r.y = p[0] * s + p[1] * c + po[1];
    reinterpret_cast<float2 *>(out)[idx] = r;
}

// NMS ensembling baseline (utils/nms.py:11-38, scored as predictors/qcnet_sliding.py:48-54,294-317): one wave per (agent,
// window size); lane = candidate.  Candidates are taken in descending confidence (ties: higher index first, the order a
// stable ascending argsort popped from its end gives); a taken candidate suppresses every other whose END POINT is closer
// than `radius`; a short selection is topped up by confidence from the suppressed ones.
__global__ __launch_bounds__(256) void nms_select_kernel(const float *__restrict__ traj, const float *__restrict__ conf,
                                                         const float *__restrict__ gt, int64_t n_agents, int n_cand_total,
                                                         int n_steps, int n_windows, int cand_per_window, int n_select,
                                                         float radius, float miss_threshold, int *__restrict__ selected,
                                                         float *__restrict__ score)
{
    const int lane = threadIdx.x & 63;
    const int64_t prob_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (prob_id >= n_agents * n_windows) return;
    const int64_t a = prob_id / n_windows;
    const int w = (int)(prob_id - a * n_windows);
    const int n_cand = (w + 1) * cand_per_window;
    const float *tr = traj + a * n_cand_total * n_steps * 2;
    const bool mine = lane < n_cand;
    const float c = mine ? conf[a * n_cand_total + lane] : 0.f;
    const float ex = mine ? tr[((int64_t)lane * n_steps + n_steps - 1) * 2] : 0.f;
    const float ey = mine ? tr[((int64_t)lane * n_steps + n_steps - 1) * 2 + 1] : 0.f;
    bool alive = mine, taken = false;
    int n_sel = 0;
    int my_slot = -1;
    for (int phase = 0; phase < 2; ++phase) {            // 0: greedy suppression, 1: top-up from everything not taken
        while (n_sel < n_select) {
            const bool cand = phase == 0 ? alive : (mine && !taken);
            const unsigned long long live = __ballot(cand);
            if (!live) break;
            float best = cand ? c : -INFINITY;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
            const unsigned long long at_best = __ballot(cand && c == best);
            const int pick = 63 - __builtin_clzll(at_best);      // highest index among equals
            const float sx = __shfl(ex, pick), sy = __shfl(ey, pick);
            if (lane == pick) { taken = true; alive = false; my_slot = n_sel; }
            const float dx = sx - ex, dy = sy - ey;
            if (phase == 0 && sqrtf(dx * dx + dy * dy) < radius) alive = false;
            ++n_sel;
        }
    }
    int *sel = selected + prob_id * n_select;
    if (my_slot >= 0) sel[my_slot] = lane;
    // min FDE / min ADE / miss of the selection against the ground truth (all steps count: no validity mask here)
    const float *g = gt + a * n_steps * 2;
    float min_fde = INFINITY, min_ade = INFINITY;
    const unsigned long long chosen = __ballot(taken);
    for (unsigned long long rest = chosen; rest; rest &= rest - 1) {
        const int k = __builtin_ctzll(rest);
        const float *p = tr + (int64_t)k * n_steps * 2;
        float s = 0.f;
        for (int t = lane; t < n_steps; t += 64) {
            const float dx = p[2 * t] - g[2 * t], dy = p[2 * t + 1] - g[2 * t + 1];
            s += sqrtf(dx * dx + dy * dy);
        }
        s = wave_sum(s) / (float)n_steps;
        const float dx = p[2 * (n_steps - 1)] - g[2 * (n_steps - 1)], dy = p[2 * (n_steps - 1) + 1] - g[2 * (n_steps - 1) + 1];
        min_fde = fminf(min_fde, sqrtf(dx * dx + dy * dy));
        min_ade = fminf(min_ade, s);
    }
    if (lane == 0) {
        float *sc = score + prob_id * 3;
        sc[0] = min_fde;
        sc[1] = min_ade;
        sc[2] = min_fde > miss_threshold ? 1.f : 0.f;
    }
}

}  // namespace

extern "C" int qcn_motion_metrics_f32(const float *pred, int pred_dim, const float *prob, const float *target, int target_dim,
                                      const uint8_t *valid, const uint8_t *select, int64_t n, int n_modes, int n_steps, int pos_dim,
                                      int pred_head, int target_head, int max_guesses, int keep_invalid_final_step,
                                      float miss_threshold, float *record, void *stream)
{
    if (n < 0 || n_modes < 1 || n_steps < 1 || max_guesses < 1 || pos_dim < 1 || pos_dim > pred_dim || pos_dim > target_dim)
        return QCN_ERR_INVALID_ARGUMENT;
    if (pred_head < -2 || pred_head >= pred_dim) return QCN_ERR_INVALID_ARGUMENT;
    if (pred_head != -1 && (target_head < 0 || target_head >= target_dim)) return QCN_ERR_INVALID_ARGUMENT;
    if (pred_head == -2 && pos_dim < 2) return QCN_ERR_INVALID_ARGUMENT;
    if (n_modes > kMaxModes) return QCN_ERR_UNSUPPORTED;
    if (n == 0) return QCN_OK;
    if (!pred || !target || !record) return QCN_ERR_INVALID_ARGUMENT;
    MetricsArgs g;
    g.pred = pred; g.prob = prob; g.target = target; g.valid = valid; g.select = select; g.record = record;
    g.n = n; g.n_modes = n_modes; g.n_steps = n_steps; g.pred_dim = pred_dim; g.target_dim = target_dim; g.pos_dim = pos_dim;
    g.pred_head = pred_head; g.target_head = target_head; g.max_guesses = max_guesses;
    g.keep_invalid_final_step = keep_invalid_final_step; g.miss_threshold = miss_threshold;
    hipLaunchKernelGGL(motion_metrics_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, qcn_stream(stream), g);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

extern "C" int qcn_to_agent_frame_f32(const float *position, const float *heading, int64_t n_agents, int n_total_steps, int pos_dim,
                                      int origin_step, int first_step, int n_steps, float *out, void *stream)
{
    if (n_agents < 0 || n_total_steps < 1 || (pos_dim != 2 && pos_dim != 3) || origin_step < 0 || origin_step >= n_total_steps ||
        first_step < 0 || n_steps < 0 || first_step + n_steps > n_total_steps)
        return QCN_ERR_INVALID_ARGUMENT;
    const int64_t total = n_agents * n_steps;
    if (total == 0) return QCN_OK;
    if (!position || !heading || !out) return QCN_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(to_agent_frame_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, qcn_stream(stream), position,
                       heading, n_agents, n_total_steps, pos_dim, origin_step, first_step, n_steps, out);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

extern "C" int qcn_to_world_frame_f32(const float *traj, int traj_dim, const float *position, const float *heading,
                                      const int64_t *rows, int64_t n_rows, int per_agent, int n_total_steps, int pos_dim,
                                      int origin_step, float *out, void *stream)
{
    if (n_rows < 0 || per_agent < 0 || traj_dim < 2 || n_total_steps < 1 || pos_dim < 2 || origin_step < 0 ||
        origin_step >= n_total_steps)
        return QCN_ERR_INVALID_ARGUMENT;
    const int64_t total = n_rows * per_agent;
    if (total == 0) return QCN_OK;
    if (!traj || !position || !heading || !out) return QCN_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(to_world_frame_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, qcn_stream(stream), traj, traj_dim,
                       position, heading, rows, n_rows, per_agent, n_total_steps, pos_dim, origin_step, out);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}

extern "C" int qcn_nms_select_f32(const float *traj, const float *conf, const float *gt, int64_t n_agents, int n_cand_total,
                                  int n_steps, int n_windows, int cand_per_window, int n_select, float radius,
                                  float miss_threshold, int *selected, float *score, void *stream)
{
    if (n_agents < 0 || n_steps < 1 || n_windows < 1 || cand_per_window < 1 || n_select < 1 ||
        (int64_t)n_windows * cand_per_window > n_cand_total || n_select > cand_per_window)
        return QCN_ERR_INVALID_ARGUMENT;
    if (n_cand_total > 64) return QCN_ERR_UNSUPPORTED;
    if (n_agents == 0) return QCN_OK;
    if (!traj || !conf || !gt || !selected || !score) return QCN_ERR_INVALID_ARGUMENT;
    const int64_t n_prob = n_agents * n_windows;
    hipLaunchKernelGGL(nms_select_kernel, dim3((unsigned)((n_prob + 3) / 4)), dim3(256), 0, qcn_stream(stream), traj, conf, gt,
                       n_agents, n_cand_total, n_steps, n_windows, cand_per_window, n_select, radius, miss_threshold, selected,
                       score);
    QCN_CHECK_LAUNCH();
    return QCN_OK;
}
