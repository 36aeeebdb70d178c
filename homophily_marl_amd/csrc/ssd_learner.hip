// ssd_learner.hip -- the two integer/elementwise pieces of the homophily learner that sit on the data path.
//   k_build_inputs        HomophilyMAC._build_inputs tail   (controllers/homophily_controller.py:137-184)
//   k_incentive_transfer  incentive reward transfer         (learners/homophily_learner.py:94-115)
//   k_td_sim_loss         double-Q TD losses of both heads + similarity loss, forward AND the gradient w.r.t. the Q-values in one
//                         launch (learners/homophily_learner.py:94-217)
// All are elementwise / row kernels over [B(*T)*n] rows; one thread per output element / row.
#include "ssd_device.h"

namespace ssd {

__global__ void k_build_inputs(int32_t rows, int32_t n, int32_t A, int32_t t0, const int64_t* __restrict__ last_actions,
                               const float* __restrict__ last_reward, const int64_t* __restrict__ last_actions_inc,
                               const float* __restrict__ pos, float pos_scale, float* __restrict__ out,
                               int32_t out_stride, int32_t out_offset) {
    const int width = A + n + 4;
    const size_t total = (size_t)rows * width;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / width), k = (int)(idx - (size_t)row * width);
        const int b = row / n, i = row - b * n;
        const int agent_major = t0 & 2;
        // t0 >> 8 = T > 0: the rows are [batch, T] and the three history tensors hold every step's OWN values -- the previous step is
        // row b - 1, none at the first step of an episode (the learner's time-batched assembly, no shifted copies)
        const int hist_T = t0 >> 8;
        const int t_zero = (t0 & 1) || (hist_T && b % hist_T == 0);
        const size_t pb = hist_T ? (size_t)b - 1 : (size_t)b, prow = pb * n + i;
        float v;
        if (k < A) v = (!t_zero && last_actions[prow] == k) ? 1.f : 0.f;                 // one-hot of the last action (:137-141)
        else if (k < A + n) v = (k - A == i) ? 1.f : 0.f;                            // agent id (:142-143)
        else if (k == A + n) {                                                       // sign(last reward) (:145-150)
            const float r = t_zero ? 0.f : last_reward[prow];
            v = (float)((r > 0.f) - (r < 0.f));
        } else if (k == A + n + 1) {                                                 // sign(#recv+ - #recv-) (:152-164)
            int recv = 0;
            if (!t_zero)
                for (int g = 0; g < n; ++g) {
                    if (g == i) continue;                                            // inc_mask_actions: no self incentive
                    const int64_t x = last_actions_inc[(pb * n + g) * n + i];
                    recv += (x == 1) - (x == 2);
                }
            v = (float)((recv > 0) - (recv < 0));
        } else v = pos[(size_t)row * 2 + (k - A - n - 2)] / pos_scale;               // pos / ||(H, W)|| (:179-181)
        const size_t orow = agent_major ? (size_t)i * (rows / n) + b : (size_t)row;
        out[orow * out_stride + out_offset + k] = v;
    }
}

// The same rows for any _build_inputs flag set: every block at the column the layout gives it (negative = absent).
__global__ void k_build_inputs_flags(int32_t rows, int32_t n, int32_t A, int32_t t0, BuildInputsLayout L,
                                     const int64_t* __restrict__ last_actions, const float* __restrict__ last_reward,
                                     const int64_t* __restrict__ last_actions_inc, const float* __restrict__ pos, float pos_scale,
                                     float* __restrict__ out, int32_t out_stride, int32_t out_offset) {
    const int width = L.width;
    const size_t total = (size_t)rows * width;
    const int agent_major = t0 & 2, hist_T = t0 >> 8;                    // hist_T: see k_build_inputs
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / width), k = (int)(idx - (size_t)row * width);
        const int b = row / n, i = row - b * n;
        const int t_zero = (t0 & 1) || (hist_T && b % hist_T == 0);
        const size_t pb = hist_T ? (size_t)b - 1 : (size_t)b, prow = pb * n + i;
        float v = 0.f;
        if (L.o_act >= 0 && k >= L.o_act && k < L.o_act + A) v = (!t_zero && last_actions[prow] == k - L.o_act) ? 1.f : 0.f;  // (:137-141)
        else if (L.o_id >= 0 && k >= L.o_id && k < L.o_id + n) v = (k - L.o_id == i) ? 1.f : 0.f;                              // (:142-143)
        else if (k == L.o_r) {                                                                                                   // (:145-150)
            const float r = t_zero ? 0.f : last_reward[prow];
            v = (float)((r > 0.f) - (r < 0.f));
        } else if (k == L.o_i) {                                                                                                 // (:152-164)
            int recv = 0;
            if (!t_zero)
                for (int g = 0; g < n; ++g) {
                    if (g == i) continue;
                    const int64_t x = last_actions_inc[(pb * n + g) * n + i];
                    recv += (x == 1) - (x == 2);
                }
            v = (float)((recv > 0) - (recv < 0));
        } else if (L.o_oth >= 0 && k >= L.o_oth && k < L.o_oth + n * A) {          // everybody's last action, agent order (:166-173)
            const int g = (k - L.o_oth) / A, a = (k - L.o_oth) - g * A;
            v = (!t_zero && last_actions[pb * n + g] == a) ? 1.f : 0.f;
        } else if (L.o_dist >= 0 && k >= L.o_dist && k < L.o_dist + n) {           // 1 - |pos_i - pos_g| / ||(H, W)|| (:174-178)
            const int g = k - L.o_dist;
            const float dx = pos[(size_t)row * 2] - pos[((size_t)b * n + g) * 2], dy = pos[(size_t)row * 2 + 1] - pos[((size_t)b * n + g) * 2 + 1];
            v = 1.f - sqrtf(dx * dx + dy * dy) / pos_scale;
        } else if (L.o_pos >= 0 && k >= L.o_pos && k < L.o_pos + 2) v = pos[(size_t)row * 2 + (k - L.o_pos)] / pos_scale;       // (:179-181)
        const size_t orow = agent_major ? (size_t)i * (rows / n) + b : (size_t)row;
        out[orow * out_stride + out_offset + k] = v;
    }
}

BuildInputsLayout build_inputs_layout(int n, int A, uint32_t input_flags) {
    const uint32_t fl = input_flags ? (input_flags & ~SSD_INPUT_EXPLICIT) : (uint32_t)SSD_INPUT_FLAGS_SHIPPED;
    BuildInputsLayout L;
    int col = 0;
    auto take = [&](uint32_t bit, int w) { const int o = (fl & bit) ? col : -1; if (fl & bit) col += w; return o; };
    L.o_act = take(SSD_INPUT_LAST_ACTION, A);                            // the reference's order (homophily_controller.py:137-184)
    L.o_id = take(SSD_INPUT_AGENT_ID, n);
    L.o_r = take(SSD_INPUT_REWARD, 1);
    L.o_i = take(SSD_INPUT_INC_REWARD, 1);
    L.o_oth = take(SSD_INPUT_OTHERS_LAST_ACTION, n * A);
    L.o_dist = take(SSD_INPUT_DISTANCE, n);
    L.o_pos = take(SSD_INPUT_AGENT_POS, 2);
    L.width = col;
    return L;
}

__global__ void k_incentive_transfer(int32_t B, int32_t T, int32_t n, const int64_t* __restrict__ a_inc,
                                     const float* __restrict__ rewards, float effect_ratio, float cost_ratio,
                                     float incentive, float seq_len, float* __restrict__ give, float* __restrict__ recv_pos,
                                     float* __restrict__ recv_neg, float* __restrict__ recv_zero, float* __restrict__ r_env,
                                     float* __restrict__ r_inc) {
    const size_t total = (size_t)B * T * n;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
        const size_t bt = it / n;
        const int i = (int)(it - bt * n);
        const int b = (int)(bt / T), t = (int)(bt - (size_t)b * T);
        const int64_t* m = a_inc + bt * n * n;
        int g = 0, rp = 0, rn = 0;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            g += m[(size_t)i * n + j] != 0;            // give: i -> j  (:101)
            const int64_t x = m[(size_t)j * n + i];    // receive: j -> i (:102-103)
            rp += x == 1; rn += x == 2;
        }
        recv_pos[it] = (float)rp; recv_neg[it] = (float)rn; recv_zero[it] = (float)(n - 1 - rp - rn);
        if (t < T - 1) {
            const size_t i1 = ((size_t)b * (T - 1) + t) * n + i;
            give[i1] = (float)g;
            const float r = rewards[i1];
            r_env[i1] = (r + (float)(rp - rn) * effect_ratio * incentive) / seq_len;   // :113
            r_inc[i1] = (r - (float)g * cost_ratio * incentive) / seq_len;             // :114
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_td_sim_loss: what homophily_learner.py:94-217 builds out of ~100 tensor ops (and autograd differentiates with ~200 more), per
// (episode b, transition t, agent i) in one thread:
//   incentive transfer (:94-115)   give_i, recv+-_i -> rewards_for_env / rewards_for_inc = (r +- ...) / max_seq_length
//   env head TD (:118-177)         double-Q: a* = argmax of the LIVE q_env[t+1] over the available actions, value from the TARGET net;
//                                  td = q_env[t, a_t] - (r_env + gamma_env (1 - terminated) tq_env[t+1, a*])
//   inc head TD                    per receiver j != i: a*_j = argmax_c q_inc[t+1, i, j, c];
//                                  td = sum_j q_inc[t, i, j, a_inc_ij] - (r_inc + gamma_inc (1 - terminated) sum_j tq_inc[t+1, i, j, a*_j])
//   similarity loss (:184-217)     window (sim_horizon) activity flags -> cluster = 2 rw + cn (the exact-value rule standing in for x-means,
//                                  SURVEY.md 8c), idle = cn + rw; for every ordered triple i != k != j != i with equal clusters:
//                                  clamp_min(-log softmax(q_inc[t, i, j])[a_inc_kj], threshold) * idle_i * idle_k
//   L = (sum (td_env mask)^2 + sum (td_inc mask)^2) / den0 + w_sim * sum_sim / (1 + den1),   den = the GLOBAL denominators (data parallel)
// Outputs: per-thread partial sums (the caller adds them in a fixed order) and, MODE 1, dL/dq_env and dL/dq_inc -- each thread owns
// the gradient rows of its (b, t, i), so nothing is accumulated across threads (deterministic).  MODE 0: the two denominators only.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr float kNeg = -9999999.f;            // the reference's masked_fill value (homophily_learner.py:137,150)
template <int MODE>
__global__ __launch_bounds__(128) void k_td_sim_loss(ssd_td_loss_args a) {
    const int B = a.batch, T1 = a.t_slots, T = T1 - 1, n = a.n_agents, A = a.n_actions;
    const int total = B * T1 * n;
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= total) return;
    const int i = it % n, bt = it / n, t = bt % T1, b = bt / T1;
    float* dqe = MODE ? a.dq_env + (size_t)it * A : nullptr;
    float* dqi = MODE ? a.dq_inc + (size_t)it * n * 3 : nullptr;
    if (t == T) {                              // the bootstrap slot: its Q-values enter the targets only (no gradient)
        if (MODE) { for (int k = 0; k < A; ++k) dqe[k] = 0.f; for (int k = 0; k < n * 3; ++k) dqi[k] = 0.f; }
        return;
    }
    float* out = a.partials + ((size_t)(b * T + t) * n + i) * SSD_TD_LOSS_PARTIALS;
    const size_t row = (size_t)bt * n;                                          // (b, t, agent 0)
    const float mask = (float)a.filled[bt] * (t ? 1.f - (float)a.terminated[bt - 1] : 1.f);   // :62-64
    // ---- similarity mask: window flags of every agent (:184-191), cluster / idle (:194-206) ------------------------------------
    uint32_t cn_bits = 0, rw_bits = 0;
    const int t_lo = t - a.sim_horizon + 1 > 0 ? t - a.sim_horizon + 1 : 0;
    for (int k = 0; k < n; ++k) {
        float cn = 0.f, rw = 0.f;
        for (int tau = t_lo; tau <= t; ++tau) {
            const size_t e = ((size_t)b * T1 + tau) * n + k;
            cn += a.clean_num[e] > 0.f ? 1.f : 0.f;
            rw += a.reward[e] / a.reward_scale;
        }
        cn_bits |= (cn > 0.f ? 1u : 0u) << k; rw_bits |= (rw > 0.f ? 1u : 0u) << k;
    }
    const int cn_i = (cn_bits >> i) & 1, rw_i = (rw_bits >> i) & 1, cl_i = 2 * rw_i + cn_i, idle_i = cn_i + rw_i;
    // sim(i, k) = [cluster_i == cluster_k] idle_i idle_k  (0, 1, 2 or 4)
    auto sim_ik = [&](int k) -> float {
        const int cn_k = (cn_bits >> k) & 1, rw_k = (rw_bits >> k) & 1;
        return (2 * rw_k + cn_k == cl_i) ? (float)(idle_i * (cn_k + rw_k)) : 0.f;
    };
    float sim_sum = 0.f;
    for (int k = 0; k < n; ++k) { if (k != i) sim_sum += sim_ik(k) * (float)(n - 2); }   // receivers j != i, j != k
    out[0] = mask; out[1] = sim_sum;
    if (!MODE) return;
    // ---- incentive transfer (:94-115) ----------------------------------------------------------------------------------------------
    const int64_t* ainc = a.actions_inc + row * n;                              // [giver][receiver] at (b, t)
    int give = 0, rp = 0, rn = 0;
    for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        give += ainc[(size_t)i * n + j] != 0;
        const int64_t x = ainc[(size_t)j * n + i];
        rp += x == 1; rn += x == 2;
    }
    const float r = a.reward[row + i] / a.reward_scale;
    const float clean = a.clean_num[row + i] > 0.f ? 1.f : 0.f;
    const float rv = (float)(rp - rn);
    const float r_env = (r + rv * a.incentive_ratio * a.incentive) / a.seq_len;
    const float r_inc = (r - (float)give * a.incentive_cost * a.incentive) / a.seq_len;
    const float live = 1.f - (float)a.terminated[bt];
    const float den0 = a.dens[0], den1 = 1.f + a.dens[1];
    // ---- env head (:118-177) ---------------------------------------------------------------------------------------------------
    const size_t qe = (size_t)it * A, qe1 = qe + (size_t)n * A;                 // (b, t, i) and (b, t + 1, i)
    const int a_t = (int)a.actions[row + i];
    const float chosen_env = a.q_env[qe + a_t];
    float tmax_env;
    {
        int best = 0; float bq = -INFINITY, btq = kNeg;
        for (int k = 0; k < A; ++k) {
            const bool ok = a.avail[qe1 + k] != 0;
            const float q = a.double_q ? (ok ? a.q_env[qe1 + k] : kNeg) : (ok ? a.tq_env[qe1 + k] : kNeg);
            if (q > bq) { bq = q; best = k; btq = ok ? a.tq_env[qe1 + k] : kNeg; }   // first maximum
        }
        (void)best;
        tmax_env = btq;
    }
    const float td_env = chosen_env - (r_env + a.gamma_env * live * tmax_env);
    const float g_env = 2.f * td_env * mask * mask / den0;
    for (int k = 0; k < A; ++k) dqe[k] = k == a_t ? g_env : 0.f;
    // ---- inc head: TD over the receivers j != i ---------------------------------------------------------------------------------
    const size_t qi = (size_t)it * n * 3, qi1 = qi + (size_t)n * n * 3;
    float sum_chosen = 0.f, sum_tmax = 0.f, q_inc_taken = 0.f;
    for (int j = 0; j < n; ++j) {
        const int c = (int)ainc[(size_t)i * n + j];
        const float qc = a.q_inc[qi + j * 3 + c];
        q_inc_taken += qc;
        if (j == i) continue;
        sum_chosen += qc;
        const float* sel = (a.double_q ? a.q_inc : a.tq_inc) + qi1 + j * 3;
        const int best = sel[1] > sel[0] ? (sel[2] > sel[1] ? 2 : 1) : (sel[2] > sel[0] ? 2 : 0);   // first maximum
        sum_tmax += a.tq_inc[qi1 + j * 3 + best];
    }
    const float td_inc = sum_chosen - (r_inc + a.gamma_inc * live * sum_tmax);
    const float g_inc = 2.f * td_inc * mask * mask / den0;
    // ---- similarity loss (:208-217) and the inc gradient rows ------------------------------------------------------------------------
    float sim_num = 0.f;
    const float wsim = a.sim_loss_weight / den1;
    for (int j = 0; j < n; ++j) {
        const float q0 = a.q_inc[qi + j * 3], q1 = a.q_inc[qi + j * 3 + 1], q2 = a.q_inc[qi + j * 3 + 2];
        const float mx = fmaxf(q0, fmaxf(q1, q2));
        const float e0 = expf(q0 - mx), e1 = expf(q1 - mx), e2 = expf(q2 - mx), es = e0 + e1 + e2;
        const float p[3] = {e0 / es, e1 / es, e2 / es};
        float g[3] = {0.f, 0.f, 0.f};
        if (j != i) {
            for (int k = 0; k < n; ++k) {
                if (k == i || k == j) continue;
                const float sm = sim_ik(k);
                if (sm == 0.f) continue;
                const int c = (int)ainc[(size_t)k * n + j];                     // the incentive action k actually gave j
                const float nl = -logf(c == 0 ? p[0] : c == 1 ? p[1] : p[2]);
                sim_num += fmaxf(nl, a.sim_threshold) * sm;
                if (nl >= a.sim_threshold) {                                    // clamp_min passes the gradient where input >= min
                    g[0] += sm * (p[0] - (c == 0 ? 1.f : 0.f)); g[1] += sm * (p[1] - (c == 1 ? 1.f : 0.f)); g[2] += sm * (p[2] - (c == 2 ? 1.f : 0.f));
                }
            }
        }
        const int cij = (int)ainc[(size_t)i * n + j];
        for (int x = 0; x < 3; ++x) dqi[j * 3 + x] = wsim * g[x] + ((j != i && x == cij) ? g_inc : 0.f);
    }
    out[2] = (td_env * mask) * (td_env * mask); out[3] = (td_inc * mask) * (td_inc * mask); out[4] = sim_num;
    out[5] = chosen_env; out[6] = q_inc_taken; out[7] = (float)give; out[8] = rv;
    out[9] = clean * rv; out[10] = clean; out[11] = r * rv; out[12] = r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_column_sums: out[g, c] = sum_r x[g, r, c] (bias gradients: the row sums of dL/dY).  One workgroup per (g, 64 columns): 4 waves
// each sum every 4th row of their column (coalesced across the 64 columns), then the four partial sums are added in a fixed
// order through LDS -- deterministic, no cross-workgroup hand-off (ATen's multi-block reductions keep arrival semaphores that a
// memset node clears; inside replayed hipGraphs they were observed to return other reductions' partial sums on this stack).
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_column_sums(const float* __restrict__ x, float* __restrict__ out, int R, int C, int chunk) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + lane, g = blockIdx.y;
    const int r0 = blockIdx.z * chunk, r1 = r0 + chunk < R ? r0 + chunk : R;       // this workgroup's rows
    float s0 = 0.f, s1 = 0.f;                                          // two chains: independent loads in flight
    if (c < C) {
        const float* p = x + (size_t)g * R * C + c;
        int r = r0 + rg;
        for (; r + 4 < r1; r += 8) { s0 += p[(size_t)r * C]; s1 += p[(size_t)(r + 4) * C]; }
        if (r < r1) s0 += p[(size_t)r * C];
    }
    part[rg][lane] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < C) out[((size_t)g * gridDim.z + blockIdx.z) * C + c] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_copy_blocks: up to SSD_COPY_BLOCKS_MAX strided 2-D block copies in one launch (blockIdx.y = block, table by value in the
// kernel arguments): the side-by-side r | z | n images of the GRU parameters and the split of their gradients.
// ---------------------------------------------------------------------------------------------------------------------------
struct CopyTable { ssd_block_copy e[SSD_COPY_BLOCKS_MAX]; };
__global__ __launch_bounds__(256) void k_copy_blocks(CopyTable t) {
    const ssd_block_copy b = t.e[blockIdx.y];
    const int total = b.rows * b.cols;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / b.cols, c = i - r * b.cols;
        b.dst[(size_t)r * b.dst_stride + c] = b.src[(size_t)r * b.src_stride + c];
    }
}

// k_fill_blocks: up to SSD_FILL_BLOCKS_MAX 32-bit pattern fills in one launch (blockIdx.y = block): the runner state an episode opens
// with (previous actions -1, previous reward / incentive actions / returns / hidden states / time index 0) was seven fill launches.
struct FillTable { ssd_block_fill e[SSD_FILL_BLOCKS_MAX]; };
__global__ __launch_bounds__(256) void k_fill_blocks(FillTable t) {
    const ssd_block_fill b = t.e[blockIdx.y];
    uint32_t* d = static_cast<uint32_t*>(b.dst);
    const size_t words = (size_t)b.bytes >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) d[i] = b.value;
}
void launch_fill_blocks(const ssd_block_fill* blocks, int count, hipStream_t stream) {
    FillTable t;
    size_t most = 4;
    for (int i = 0; i < SSD_FILL_BLOCKS_MAX; ++i) {
        t.e[i] = blocks[i < count ? i : 0];
        if (i < count && (size_t)blocks[i].bytes > most) most = (size_t)blocks[i].bytes;
    }
    int gx = (int)((most / 4 + 1023) / 1024);                          // 4 words per thread for the largest block
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(k_fill_blocks, dim3(gx, count), dim3(256), 0, stream, t);
}

// k_runner_stats: the device side of EpisodeRunner's episode statistics (episode_runner.py:121-152) for one rollout of all envs:
// acc[0..3] += sum collective_return, sum equality, sum of the agents' episode returns, sum of their squares, in f64, one workgroup,
// a fixed-order tree (deterministic) -- ten ATen launches (casts, four reductions, a product, a concatenation, an addition) before.
__global__ __launch_bounds__(1024) void k_runner_stats(const float* __restrict__ coll, const float* __restrict__ eq, const float* __restrict__ ret,
                                                       int n_env, int n_ret, double* __restrict__ acc) {
    __shared__ double red[4][1024];
    const int t = threadIdx.x;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = t; i < n_env; i += 1024) { s[0] += (double)coll[i]; s[1] += (double)eq[i]; }
    for (int i = t; i < n_ret; i += 1024) { const double r = (double)ret[i]; s[2] += r; s[3] += r * r; }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][t] = s[k];
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (t < w) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][t] += red[k][t + w];
        }
        __syncthreads();
    }
    if (t < 4) acc[t] += red[t][0];
}
void launch_runner_stats(const float* coll, const float* eq, const float* ret, int n_env, int n_ret, double* acc, hipStream_t stream) {
    hipLaunchKernelGGL(k_runner_stats, dim3(1), dim3(1024), 0, stream, coll, eq, ret, n_env, n_ret, acc);
}

static int grid_for(size_t total);
// k_dueling_q: q = v + a - mean_k a per (agent, row) of the learner's time-batched heads, output in the batch layout [B, T, n, inner, K];
// BWD: da = dq - mean_k dq, dv = sum_k dq.  One thread per (agent, row); K <= 16.  ld: floats per row of a (and da); v / dv rows have the
// same stride when `merged` (advantages and value are columns 0..K-1 and K of ONE layer output [n, rows, K + 1]), else 1.
// gs (BWD, merged, nullable): the row-group sums of the gradient, gs[i, tb, 0..K] = sum over the `inner` rows of (i, tb) -- what the part
// of the layer input that is shared by the group (the incentive head's h_i under all receivers j) receives.
template <bool BWD>
__global__ __launch_bounds__(256) void k_dueling_q(const float* __restrict__ a, const float* __restrict__ v, float* __restrict__ q,
                                                   const float* __restrict__ dq, float* __restrict__ da, float* __restrict__ dv, int n, int T, int B,
                                                   int inner, int K, int ld, int merged, float* __restrict__ gs) {
    const long rows = (long)T * B * inner, total = rows * n;
    const int vld = merged ? ld : 1;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx / rows);
        const long r = idx - (long)i * rows;
        const long tb = r / inner;
        const int j = (int)(r - tb * inner), t = (int)(tb / B), b = (int)(tb - (long)t * B);
        const size_t qo = ((((size_t)b * T + t) * n + i) * inner + j) * K, ao = ((size_t)i * rows + r) * ld;
        float x[16], sum = 0.f;
        for (int k = 0; k < K; ++k) { x[k] = BWD ? dq[qo + k] : a[ao + k]; sum += x[k]; }
        if (BWD) {
            const float mean = sum / (float)K;
            for (int k = 0; k < K; ++k) da[ao + k] = x[k] - mean;
            dv[((size_t)i * rows + r) * vld] = sum;
            if (gs && j == 0) {                                        // this thread also adds the group's rows in order (deterministic)
                float g[17];
                for (int k = 0; k <= K; ++k) g[k] = 0.f;
                for (int jj = 0; jj < inner; ++jj) {
                    const size_t qj = qo + (size_t)jj * K;
                    float y[16], s2 = 0.f;
                    for (int k = 0; k < K; ++k) { y[k] = dq[qj + k]; s2 += y[k]; }
                    const float m2 = s2 / (float)K;
                    for (int k = 0; k < K; ++k) g[k] += y[k] - m2;
                    g[K] += s2;
                }
                float* go = gs + ((size_t)i * (rows / inner) + tb) * (K + 1);
                for (int k = 0; k <= K; ++k) go[k] = g[k];
            }
        } else {
            const float vv = v[((size_t)i * rows + r) * vld], mean = sum / (float)K;
            for (int k = 0; k < K; ++k) q[qo + k] = (vv + x[k]) - mean;               // v + a - mean(a), in the reference's order
        }
    }
}

void launch_dueling_q(const float* a, const float* v, float* q, const float* dq, float* da, float* dv, int n, int T, int B, int inner, int K,
                      hipStream_t stream, int ld, int merged, float* gs) {
    const size_t total = (size_t)n * T * B * inner;
    if (ld <= 0) ld = K;
    if (dq) hipLaunchKernelGGL(k_dueling_q<true>, dim3(grid_for(total)), dim3(256), 0, stream, a, v, q, dq, da, dv, n, T, B, inner, K, ld, merged, gs);
    else hipLaunchKernelGGL(k_dueling_q<false>, dim3(grid_for(total)), dim3(256), 0, stream, a, v, q, dq, da, dv, n, T, B, inner, K, ld, merged, gs);
}

// k_gather_rows: rows ids[e] of up to SSD_COPY_BLOCKS_MAX fields into consecutive rows of their destinations (blockIdx.y = field,
// blockIdx.z = e); 4-byte accesses at any alignment for the bulk of a row, bytes for its tail.
struct GatherTable { ssd_row_gather f[SSD_COPY_BLOCKS_MAX]; };
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
__global__ __launch_bounds__(256) void k_gather_rows(GatherTable t, const int64_t* __restrict__ ids) {
    const ssd_row_gather f = t.f[blockIdx.y];
    const uint8_t* src = static_cast<const uint8_t*>(f.src) + (size_t)ids[blockIdx.z] * f.row_bytes;
    uint8_t* dst = static_cast<uint8_t*>(f.dst) + (size_t)blockIdx.z * f.row_bytes;
    const long words = f.row_bytes >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long)gridDim.x * 256)
        *reinterpret_cast<u32_unaligned*>(dst + 4 * i) = *reinterpret_cast<const u32_unaligned*>(src + 4 * i);
    if (blockIdx.x == 0 && (long)threadIdx.x < (f.row_bytes & 3)) dst[4 * words + threadIdx.x] = src[4 * words + threadIdx.x];
}

void launch_gather_rows(const ssd_row_gather* fields, int count, const int64_t* ids, int n_ids, hipStream_t stream) {
    GatherTable t;
    int64_t most = 4;
    for (int i = 0; i < SSD_COPY_BLOCKS_MAX; ++i) {
        t.f[i] = fields[i < count ? i : 0];
        if (i < count && fields[i].row_bytes > most) most = fields[i].row_bytes;
    }
    int gx = (int)((most / 4 + 1023) / 1024);                             // 4 words per thread for the longest row
    if (gx < 1) gx = 1;
    if (gx > 32) gx = 32;
    hipLaunchKernelGGL(k_gather_rows, dim3(gx, count, n_ids), dim3(256), 0, stream, t, ids);
}

// k_sample_ids: `count` distinct episode ids out of [0, population), uniformly, from a counter generator keyed by (seed, call) --
// ReplayBuffer.sample's np.random.choice(replace = False) (episode_buffer.py:240-244) without a host draw or an H2D copy.  Floyd's
// subset sampling (a uniform count-subset with count draws, no population-sized permutation), then a Fisher-Yates pass over the
// count picks.  One thread: count is a batch size (16 in the shipped configuration; the ABI bounds it at SSD_SAMPLE_IDS_MAX).
__device__ __forceinline__ uint32_t sample_draw(uint32_t s0, uint32_t s1, uint32_t call, uint32_t i) {
    uint32_t x = s0 ^ (call * 0x9E3779B9u);
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    x ^= s1 + i * 0x85EBCA6Bu;
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    return x;
}
__global__ __launch_bounds__(64) void k_sample_ids(uint32_t s0, uint32_t s1, uint32_t call, int population, int count, int64_t* __restrict__ out) {
    // the picks live in LDS while they are compared and swapped (through `out` every pick waited for the stores before it: 15 us for 16 ids)
    __shared__ int32_t pick[SSD_SAMPLE_IDS_MAX];
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0) {
        for (int i = 0; i < count; ++i) {
            const int j = population - count + i;
            const int t = (int)(((uint64_t)sample_draw(s0, s1, call, (uint32_t)i) * (uint32_t)(j + 1)) >> 32);
            bool taken = false;
            for (int k = 0; k < i; ++k) taken |= pick[k] == t;
            pick[i] = taken ? j : t;
        }
        for (int i = count - 1; i > 0; --i) {
            const int k = (int)(((uint64_t)sample_draw(s0, s1, call, (uint32_t)(count + i)) * (uint32_t)(i + 1)) >> 32);
            const int32_t a = pick[i]; pick[i] = pick[k]; pick[k] = a;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += 64) out[i] = (int64_t)pick[i];
}
void launch_sample_ids(uint64_t seed, uint32_t call, int population, int count, int64_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_sample_ids, dim3(1), dim3(64), 0, stream, (uint32_t)seed, (uint32_t)(seed >> 32), call, population, count, out);
}

void launch_copy_blocks(const ssd_block_copy* blocks, int count, hipStream_t stream) {
    CopyTable t;
    int most = 1;
    for (int i = 0; i < SSD_COPY_BLOCKS_MAX; ++i) {
        t.e[i] = blocks[i < count ? i : 0];
        if (i < count && blocks[i].rows * blocks[i].cols > most) most = blocks[i].rows * blocks[i].cols;
    }
    int gx = (most + 1023) / 1024;                                     // 4 elements per thread for the largest block
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_copy_blocks, dim3(gx, count), dim3(256), 0, stream, t);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_gradsq_partials / k_clip_adam: both gradient clips and both Adam steps of the learner over the flat gradient buffer
// (include/ssd_hip.h: ssd_clip_adam_step; homophily_learner.py:223-226).  1024 elements per workgroup, 4 per thread.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int ADAM_CHUNK = 1024;
__device__ __forceinline__ float block_sum_256(float v, float* red) {      // fixed-order tree over 256 threads
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// job (parameter tensor) of flat element `base`, given the ascending first elements j_off[0 .. n_jobs]
__device__ __forceinline__ int adam_job_of(const long* j_off, int n_jobs, long base) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (j_off[mid] <= base) lo = mid; else hi = mid - 1; }
    return lo;
}

__global__ __launch_bounds__(256) void k_gradsq_partials(ssd_clip_adam_args a) {
    __shared__ float red[256];
    __shared__ long j_off[SSD_ADAM_MAX_JOBS + 1];
    __shared__ int j_seg[SSD_ADAM_MAX_JOBS];
    if ((int)threadIdx.x < a.n_jobs) {
        const ssd_adam_job j = a.jobs[threadIdx.x];
        j_off[threadIdx.x] = j.offset; j_seg[threadIdx.x] = j.segment;
        if ((int)threadIdx.x == a.n_jobs - 1) j_off[a.n_jobs] = j.offset + j.numel;
    }
    __syncthreads();
    const long base = (long)blockIdx.x * ADAM_CHUNK + threadIdx.x * 4;
    float s[3] = {0.f, 0.f, 0.f};
    if (base < a.total) {
        int job = adam_job_of(j_off, a.n_jobs, base);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long i = base + r;
            if (i < a.total) {
                while (i >= j_off[job + 1]) ++job;
                const float g = a.flat_grad[i];
                const float gg = g * g;
                const int seg = j_seg[job];
                s[0] += seg == 0 ? gg : 0.f; s[1] += seg == 1 ? gg : 0.f; s[2] += seg == 2 ? gg : 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = block_sum_256(s[k], red);
        if (threadIdx.x == 0) a.partials[(size_t)blockIdx.x * 3 + k] = v;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < a.n_jobs) {                   // step counters (read by the second launch only)
        const ssd_adam_job j = a.jobs[threadIdx.x];
        if (j.step[0]) *j.step[0] += 1.f;
        if (j.step[1]) *j.step[1] += 1.f;
    }
}

__global__ __launch_bounds__(256) void k_clip_adam(ssd_clip_adam_args a, int chunks) {
    __shared__ float red[256];
    __shared__ long j_off[SSD_ADAM_MAX_JOBS + 1];
    __shared__ float j_size[SSD_ADAM_MAX_JOBS][2], j_rsq[SSD_ADAM_MAX_JOBS][2];   // lr / (1 - b1^t), 1 / sqrt(1 - b2^t) per optimiser
    const int t = threadIdx.x;
    float S[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                          // the chunk sums in a fixed order
        float v = 0.f;
        for (int c = t; c < chunks; c += 256) v += a.partials[(size_t)c * 3 + k];
        S[k] = block_sum_256(v, red);
    }
    // torch.clamp(clip_coef, max = 1) keeps a NaN coefficient (clip_grad_norm_ then spreads it over every gradient of the group);
    // fminf alone would return 1 and hide a non-finite norm
    const float r_inc = a.clip / (sqrtf(S[0] + S[2]) + 1e-6f);
    const float c_inc = r_inc != r_inc ? r_inc : fminf(1.f, r_inc);
    const float r_env = a.clip / (sqrtf(c_inc * c_inc * S[0] + S[1]) + 1e-6f);
    const float c_env = r_env != r_env ? r_env : fminf(1.f, r_env);
    if (t < a.n_jobs) {
        const ssd_adam_job j = a.jobs[t];
        j_off[t] = j.offset;
        if (t == a.n_jobs - 1) j_off[a.n_jobs] = j.offset + j.numel;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float st = j.step[o] ? *j.step[o] : 1.f;
            const float bc1 = (float)(1.0 - pow((double)a.beta1, (double)st)), bc2 = (float)(1.0 - pow((double)a.beta2, (double)st));
            j_size[t][o] = (o == 0 ? a.lr_inc : a.lr_env) / bc1;
            j_rsq[t][o] = sqrtf(bc2);
        }
    }
    __syncthreads();
    const long base = (long)blockIdx.x * ADAM_CHUNK + t * 4;
    if (base >= a.total) return;
    int job = adam_job_of(j_off, a.n_jobs, base);
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2;
    for (int r = 0; r < 4; ++r) {
        const long i = base + r;
        if (i >= a.total) break;
        while (i >= j_off[job + 1]) ++job;
        const ssd_adam_job j = a.jobs[job];
        const long e = i - j.offset;
        float g = a.flat_grad[i];
        g = j.segment == 0 ? (g * c_inc) * c_env : (j.segment == 1 ? g * c_env : g * c_inc);   // clip inc first, then env
        a.flat_grad[i] = g;
        float p = j.param[e];
#pragma unroll
        for (int o = 0; o < 2; ++o) {                                      // optimiser_inc.step(), then optimiser_env.step()
            if (!j.exp_avg[o]) continue;
            float m = j.exp_avg[o][e], v = j.exp_avg_sq[o][e];
            m = m + w1 * (g - m);                                          // lerp(m, g, 1 - beta1)
            v = a.beta2 * v + (w2 * g) * g;
            j.exp_avg[o][e] = m; j.exp_avg_sq[o][e] = v;
            const float denom = sqrtf(v) / j_rsq[job][o] + a.eps;
            p -= (j_size[job][o] * m) / denom;
        }
        j.param[e] = p;
    }
}

void launch_clip_adam(const ssd_clip_adam_args* a, hipStream_t stream) {
    const int chunks = (int)((a->total + ADAM_CHUNK - 1) / ADAM_CHUNK);
    hipLaunchKernelGGL(k_gradsq_partials, dim3(chunks), dim3(256), 0, stream, *a);
    hipLaunchKernelGGL(k_clip_adam, dim3(chunks), dim3(256), 0, stream, *a, chunks);
}

static int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : b > 4096 ? 4096 : b);
}

void launch_build_inputs(int32_t batch, int32_t n, int32_t A, int32_t t0, const int64_t* last_actions,
                         const float* last_reward, const int64_t* last_actions_inc, const float* pos, float pos_scale,
                         float* out, int32_t out_stride, int32_t out_offset, hipStream_t stream) {
    const size_t total = (size_t)batch * n * (A + n + 4);
    hipLaunchKernelGGL(k_build_inputs, dim3(grid_for(total)), dim3(256), 0, stream, batch * n, n, A, t0, last_actions,
                       last_reward, last_actions_inc, pos, pos_scale, out, out_stride, out_offset);
}

void launch_build_inputs_flags(int32_t batch, int32_t n, int32_t A, int32_t t0, uint32_t input_flags, const int64_t* last_actions,
                               const float* last_reward, const int64_t* last_actions_inc, const float* pos, float pos_scale,
                               float* out, int32_t out_stride, int32_t out_offset, hipStream_t stream) {
    const BuildInputsLayout L = build_inputs_layout(n, A, input_flags);
    if (L.width < 1) return;                                            // no tail blocks at all
    const size_t total = (size_t)batch * n * L.width;
    hipLaunchKernelGGL(k_build_inputs_flags, dim3(grid_for(total)), dim3(256), 0, stream, batch * n, n, A, t0, L, last_actions,
                       last_reward, last_actions_inc, pos, pos_scale, out, out_stride, out_offset);
}

// other_j = [one-hot(a_j), pos_j / scale, orient_j, r_j, clean_j, apple_den_j] (homophily_agent.py:194-201) of every (episode b, step t,
// agent j) as other [T * B, n, A + 7] (rows t * B + b: the layout the time-batched incentive head consumes), and the one-hot alone a
// second time as act_tm [n, T * B, A] (agent-major: the extra input columns of fc1_inc).  One thread per (b, t, j): the reference
// builds this from F.one_hot, a division, a concatenation and two permuted copies -- six launches here, plus two per network for act_tm.
__global__ __launch_bounds__(256) void k_unroll_other(const int64_t* __restrict__ actions, const float* __restrict__ pos, const float* __restrict__ orient,
                                                      const float* __restrict__ reward, const float* __restrict__ clean, const float* __restrict__ den,
                                                      float pos_scale, int B, int T, int n, int A, float* __restrict__ other, float* __restrict__ act_tm) {
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= B * T * n) return;
    const int j = it % n, bt = it / n, t = bt % T, b = bt / T;
    const int a = (int)actions[it];
    const int E = A + 7;
    float* o = other + ((size_t)(t * B + b) * n + j) * E;
    float* at = act_tm + ((size_t)j * T * B + (size_t)t * B + b) * A;
    for (int k = 0; k < A; ++k) { const float v = k == a ? 1.f : 0.f; o[k] = v; at[k] = v; }
    o[A] = pos[(size_t)it * 2] / pos_scale; o[A + 1] = pos[(size_t)it * 2 + 1] / pos_scale;
    o[A + 2] = orient[(size_t)it * 2]; o[A + 3] = orient[(size_t)it * 2 + 1];
    o[A + 4] = reward[it]; o[A + 5] = clean[it]; o[A + 6] = den[it];
}
void launch_unroll_other(const int64_t* actions, const float* pos, const float* orient, const float* reward, const float* clean, const float* den,
                         float pos_scale, int B, int T, int n, int A, float* other, float* act_tm, hipStream_t stream) {
    const int total = B * T * n;
    hipLaunchKernelGGL(k_unroll_other, dim3((total + 255) / 256), dim3(256), 0, stream, actions, pos, orient, reward, clean, den, pos_scale, B, T, n, A, other, act_tm);
}

void launch_incentive_transfer(int32_t B, int32_t T, int32_t n, const int64_t* a_inc, const float* rewards,
                               float effect_ratio, float cost_ratio, float incentive, float seq_len, float* give,
                               float* recv_pos, float* recv_neg, float* recv_zero, float* r_env, float* r_inc,
                               hipStream_t stream) {
    const size_t total = (size_t)B * T * n;
    hipLaunchKernelGGL(k_incentive_transfer, dim3(grid_for(total)), dim3(256), 0, stream, B, T, n, a_inc, rewards,
                       effect_ratio, cost_ratio, incentive, seq_len, give, recv_pos, recv_neg, recv_zero, r_env, r_inc);
}

// Two deterministic stages when there are many rows: chunks of COLSUM_CHUNK rows -> workspace [G, chunks, C] -> out [G, C].
void launch_column_sums(const float* x, float* out, int G, int R, int C, float* workspace, hipStream_t stream) {
    const int chunks = (R + SSD_COLSUM_CHUNK - 1) / SSD_COLSUM_CHUNK;
    if (chunks <= 1 || !workspace) {
        hipLaunchKernelGGL(k_column_sums, dim3((C + 63) / 64, G, 1), dim3(256), 0, stream, x, out, R, C, R);
        return;
    }
    hipLaunchKernelGGL(k_column_sums, dim3((C + 63) / 64, G, chunks), dim3(256), 0, stream, x, workspace, R, C, SSD_COLSUM_CHUNK);
    hipLaunchKernelGGL(k_column_sums, dim3((C + 63) / 64, G, 1), dim3(256), 0, stream, workspace, out, chunks, C, chunks);
}

void launch_td_sim_loss(const ssd_td_loss_args* a, int mode, hipStream_t stream) {
    const int total = a->batch * a->t_slots * a->n_agents;
    const dim3 grid((total + 127) / 128), block(128);
    if (mode) hipLaunchKernelGGL(k_td_sim_loss<1>, grid, block, 0, stream, *a);
    else hipLaunchKernelGGL(k_td_sim_loss<0>, grid, block, 0, stream, *a);
}

}  // namespace ssd
