// ssd_policy.hip -- the homophily controller one layer at a time: the pieces that are not GEMMs.
//   k_encoder       conv(3->C, k3, s1) + LeakyReLU + Linear(C*(V-2)^2 -> F) + LeakyReLU  (homophily_agent.py:20-27,213-214)
//   k_conv_leaky    the conv + LeakyReLU half of it (any window size)
//   k_gru_gates     r, z, n gates and the state update of the hand-written GRU cell       (homophily_agent.py:162-165,188-191)
//   k_gru_fwd_train / k_gru_bwd   the same with saved gates / its backward (per-step form; the learner uses ssd_gru_seq.hip)
//   k_dueling_pick  q = v + a - mean(a) and the epsilon-greedy choice                    (homophily_agent.py:168-170,204-206;
//                                                                                         action_selectors.py:44-68)
//   k_store_step    the small per-step fields of the episode storage in one launch       (episode_runner.py:59-93)
// In this per-layer composition the per-agent matrix products stay in hipBLASLt.  It is the second implementation of the
// rollout-time controller (window sizes / palettes without a fused encoder, FastPolicy(fused=False)) and the cross-check of the
// fused matrix-core kernels in ssd_policy_mfma.hip, which replace it on the hot path.
#include "ssd_policy_common.h"

namespace ssd {

// ---------------------------------------------------------------------------------------------------------------
// Encoder.  One wave per observation row (env, agent).  The obs [3, V, V] f32 is staged in LDS; lanes own conv output
// positions (p = lane, lane + 64, ...) and keep the C conv channels of their positions in registers; the linear layer
// is accumulated per lane over its positions for all F outputs (W read coalesced: W_t[f][c * P + p], lanes = p) and
// reduced across the wave with a butterfly.  Conv weights are wave-uniform (scalar loads).
// out row stride / offset let the features land directly inside the agent-input matrix.
// ---------------------------------------------------------------------------------------------------------------
template <int C, int F, int RPW>
__global__ __launch_bounds__(256) void k_encoder(const float* __restrict__ obs, int rows, int V, const float* __restrict__ cw /*[C,3,3,3]*/,
                                                 const float* __restrict__ cb /*[C]*/, const float* __restrict__ lw /*[F, C*P]*/,
                                                 const float* __restrict__ lb /*[F]*/, float* __restrict__ out, int out_stride,
                                                 int n_agents, int agent_major, float* __restrict__ store, long store_env_stride,
                                                 const int64_t* __restrict__ store_t) {
    // RPW rows per wave: every linear weight fetched from L2 is used for RPW rows (the weight matrix is 130 KB; one row
    // per wave re-read it 20 480 times per timestep).
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = (blockIdx.x * 4 + wave) * RPW;
    if (row0 >= rows) return;
    const int VV = V * V, O = V - 2, P = O * O, K = C * P, L = 3 * VV;
    float* in = sm + wave * RPW * (L + 4);
    // stage RPW observations in LDS; optionally copy them into the episode storage obs[env, t] on the way
    // (rows of one env are contiguous: obs element (row, e) of env b = row / n sits at b * env_stride + t * n * L + (row % n) * L + e)
    const long t_off = store ? (long)(*store_t) * n_agents * L : 0;
    for (int q = 0; q < RPW; ++q) {
        const int row = row0 + q;
        if (row < rows) {
            const float* src = obs + (size_t)row * L;
            float* dst = nullptr;
            if (store) { const int b = row / n_agents, i = row - b * n_agents; dst = store + (long)b * store_env_stride + t_off + (long)i * L; }
            for (int e = lane; e < L; e += 64) { const float v = src[e]; in[q * (L + 4) + e] = v; if (dst) dst[e] = v; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float acc[RPW][F];
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int f = 0; f < F; ++f) acc[q][f] = 0.f;
    for (int p = lane; p < P; p += 64) {
        const int y = p / O, x = p - y * O;
        float c[RPW][C];
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
            for (int o = 0; o < C; ++o) c[q][o] = cb[o];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float v[RPW];
#pragma unroll
                    for (int q = 0; q < RPW; ++q) v[q] = in[q * (L + 4) + ch * VV + (y + dy) * V + x + dx];
#pragma unroll
                    for (int o = 0; o < C; ++o) {
                        const float w = cw[((o * 3 + ch) * 3 + dy) * 3 + dx];
#pragma unroll
                        for (int q = 0; q < RPW; ++q) c[q][o] = fmaf(w, v[q], c[q][o]);
                    }
                }
#pragma unroll
        for (int o = 0; o < C; ++o) {
            float a[RPW];
#pragma unroll
            for (int q = 0; q < RPW; ++q) a[q] = leaky(c[q][o]);
            const float* w = lw + o * P + p;              // Flatten order of [C, O, O]: k = o * P + p
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float wf = w[(size_t)f * K];
#pragma unroll
                for (int q = 0; q < RPW; ++q) acc[q][f] = fmaf(wf, a[q], acc[q][f]);
            }
        }
    }
    // wave reduction of the F partial sums per row; afterwards lane f holds output f
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        float mine = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float v = acc[q][f];
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);
            if (lane == f) mine = v;
        }
        const int row = row0 + q;
        if (lane < F && row < rows) {
            size_t orow = row;
            if (agent_major) { const int b = row / n_agents, i = row - b * n_agents; orow = (size_t)i * (rows / n_agents) + b; }
            out[orow * out_stride + lane] = leaky(mine + lb[lane]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv(3 -> C, k3, s1) + LeakyReLU only: one workgroup per observation row, the row staged in LDS, one thread per conv
// output k = o * P + p (Flatten order).  HBM-bound (2.7 KB in, 4 KB out per row).  The Linear layer behind it is a plain
// [rows, C*P] x [C*P, F] GEMM and goes to hipBLASLt.  Rows can be written agent-major, and the observation can be
// copied into the episode storage obs[env, t] on the way (see k_encoder).
// ---------------------------------------------------------------------------------------------------------------
template <int C, int VT>   // VT: compile-time view edge (0 = runtime), so the index divisions become multiplies
__global__ __launch_bounds__(256) void k_conv_leaky(const float* __restrict__ obs, int rows, int Vrt, const float* __restrict__ cw,
                                                    const float* __restrict__ cb, float* __restrict__ out, int n_agents, int agent_major,
                                                    float* __restrict__ store, long store_env_stride, const int64_t* __restrict__ store_t) {
    extern __shared__ float sm[];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int V = VT ? VT : Vrt;
    const int VV = V * V, O = V - 2, P = O * O, K = C * P, L = 3 * VV;
    float* in = sm;               // [L]
    float* w = sm + L;            // [C * 27] + [C]
    const float* src = obs + (size_t)row * L;
    const int b = row / n_agents, i = row - b * n_agents;
    float* dst = store ? store + (long)b * store_env_stride + (long)(*store_t) * n_agents * L + (long)i * L : nullptr;
    for (int e = tid; e < L; e += 256) { const float v = src[e]; in[e] = v; if (dst) dst[e] = v; }
    for (int e = tid; e < C * 27; e += 256) w[e] = cw[e];
    if (tid < C) w[C * 27 + tid] = cb[tid];
    __syncthreads();
    const size_t orow = agent_major ? (size_t)i * (rows / n_agents) + b : (size_t)row;
    for (int k = tid; k < K; k += 256) {
        const int o = k / P, p = k - o * P, y = p / O, x = p - y * O;
        float acc = w[C * 27 + o];
        const float* wo = w + o * 27;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc = fmaf(wo[(ch * 3 + dy) * 3 + dx], in[ch * VV + (y + dy) * V + x + dx], acc);
        out[orow * K + k] = leaky(acc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Store step: the small per-timestep fields of the episode storage [N, T+1, ...] in one launch (the reference issues one
// EpisodeBatch.update per field, episode_runner.py:67,80-93).  Destination element = base[(b * slots + t) * width + k].
// ---------------------------------------------------------------------------------------------------------------
struct StoreStep {
    const int64_t* t;
    int N, n, A, slots;
    const float *pos, *orient, *reward, *clean, *den;        // [N, n, 2] x2, [N, n] x3 (nullable: skipped)
    const uint8_t* term;                                      // [N]
    const int64_t *actions, *actions_inc;                     // [N, n], [N, n, n]
    float *d_pos, *d_orient, *d_reward, *d_clean, *d_den, *d_onehot;
    uint8_t* d_term;
    int64_t *d_actions, *d_actions_inc;
    int64_t *p_act, *p_inc;
    float *p_rew, *ep_ret;
    int64_t *next_t, *ctr_inc;
};
__global__ void k_store_step(StoreStep s) {
    const long t = *s.t;
    const int n = s.n, A = s.A;
    const int per_env = n * (2 + 2 + 1 + 1 + 1 + 1 + A + n) + 1;
    const long total = (long)s.N * per_env;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / per_env;
        int k = (int)(idx - b * per_env);
        const long bt = b * s.slots + t;
        if (k < 2 * n) { if (s.pos) s.d_pos[bt * 2 * n + k] = s.pos[b * 2 * n + k]; continue; } k -= 2 * n;
        if (k < 2 * n) { if (s.orient) s.d_orient[bt * 2 * n + k] = s.orient[b * 2 * n + k]; continue; } k -= 2 * n;
        if (k < n) {
            if (s.reward) {
                const float r = s.reward[b * n + k];
                s.d_reward[bt * n + k] = r;
                if (s.p_rew) s.p_rew[b * n + k] = r;
                if (s.ep_ret) s.ep_ret[b * n + k] += r;
            }
            continue;
        } k -= n;
        if (k < n) { if (s.clean) s.d_clean[bt * n + k] = s.clean[b * n + k]; continue; } k -= n;
        if (k < n) { if (s.den) s.d_den[bt * n + k] = s.den[b * n + k]; continue; } k -= n;
        if (k < n) { if (s.actions) { const int64_t v = s.actions[b * n + k]; s.d_actions[bt * n + k] = v; if (s.p_act) s.p_act[b * n + k] = v; } continue; } k -= n;
        if (k < n * A) { if (s.actions) { const int i = k / A, a = k - i * A; s.d_onehot[(bt * n + i) * A + a] = s.actions[b * n + i] == a ? 1.f : 0.f; } continue; } k -= n * A;
        if (k < n * n) { if (s.actions_inc) { const int64_t v = s.actions_inc[b * n * n + k]; s.d_actions_inc[bt * n * n + k] = v; if (s.p_inc) s.p_inc[b * n * n + k] = v; } continue; } k -= n * n;
        if (s.term) s.d_term[bt] = s.term[b];
    }
    // counters this kernel does not read: no ordering against the other blocks is needed
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (s.next_t) *s.next_t = t + 1;
        if (s.ctr_inc) *s.ctr_inc += 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GRU gates: gi, gh [R, 3H] (input-side and hidden-side projections incl. biases), h [R, H] updated in place.
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_gru_gates(const float* __restrict__ gi, const float* __restrict__ gh, float* __restrict__ h, int R, int H) {
    const size_t total = (size_t)R * H;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / H;
        const int k = (int)(idx - r * H);
        const float* a = gi + r * 3 * H;
        const float* b = gh + r * 3 * H;
        const float rg = 1.f / (1.f + __expf(-(a[k] + b[k])));
        const float zg = 1.f / (1.f + __expf(-(a[H + k] + b[H + k])));
        const float ng = tanhf(a[2 * H + k] + rg * b[2 * H + k]);
        h[idx] = (1.f - zg) * ng + zg * h[idx];
    }
}

// Training variants: forward keeps (r, z, n) for the backward pass; backward maps dL/dh_new to dL/dgi, dL/dgh, dL/dh_prev.
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z h
__global__ void k_gru_fwd_train(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h,
                                float* __restrict__ h_new, float* __restrict__ rzn, int R, int H) {
    const size_t total = (size_t)R * H;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / H;
        const int k = (int)(idx - r * H);
        const float* a = gi + r * 3 * H;
        const float* b = gh + r * 3 * H;
        const float rg = 1.f / (1.f + expf(-(a[k] + b[k])));
        const float zg = 1.f / (1.f + expf(-(a[H + k] + b[H + k])));
        const float ng = tanhf(a[2 * H + k] + rg * b[2 * H + k]);
        float* s = rzn + r * 3 * H;
        s[k] = rg; s[H + k] = zg; s[2 * H + k] = ng;
        h_new[idx] = (1.f - zg) * ng + zg * h[idx];
    }
}
__global__ void k_gru_bwd(const float* __restrict__ dh, const float* __restrict__ rzn, const float* __restrict__ gh,
                          const float* __restrict__ h, float* __restrict__ d_gi, float* __restrict__ d_gh, float* __restrict__ dh_prev,
                          int R, int H) {
    const size_t total = (size_t)R * H;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / H;
        const int k = (int)(idx - r * H);
        const float* s = rzn + r * 3 * H;
        const float rg = s[k], zg = s[H + k], ng = s[2 * H + k];
        const float g = dh[idx];
        const float ghn = gh[r * 3 * H + 2 * H + k];
        const float d_n = g * (1.f - zg) * (1.f - ng * ng);       // through tanh
        const float d_z = g * (h[idx] - ng) * zg * (1.f - zg);    // through sigmoid
        const float d_r = d_n * ghn * rg * (1.f - rg);
        float* a = d_gi + r * 3 * H;
        float* b = d_gh + r * 3 * H;
        a[k] = d_r; b[k] = d_r;
        a[H + k] = d_z; b[H + k] = d_z;
        a[2 * H + k] = d_n; b[2 * H + k] = d_n * rg;
        dh_prev[idx] = g * zg;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Dueling head + epsilon-greedy.  av [R, A + 1]: A advantages then the state value.  avail [A] (u8, same for all rows,
// nullptr = all available).  Random numbers: the counter generator of include/ssd_hip.h keyed by (seed, *step, row).
// Row r of the [n(i), B, ...] agent-major input is written to the env-major position given by (out_b_stride, out_i_stride).
// inc head: rows are (i, b, j); diagonal (i == j) forced to 0 (no self-incentive, homophily_controller.py:44-46).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_dueling_pick(const float* __restrict__ av, int R, int A, const uint8_t* __restrict__ avail,
                               const float* __restrict__ eps_p, const int64_t* __restrict__ step_p, uint32_t seed, int n_agents, int B,
                               int pairs /*0: rows (i,b); 1: rows (i,b,j)*/, int64_t* __restrict__ actions, float* __restrict__ q_out,
                               uint32_t env_id_base) {
    const float eps = *eps_p;
    const uint32_t step = (uint32_t)*step_p;
    const uint32_t n = (uint32_t)n_agents;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        const float* a = av + (size_t)r * (A + 1);
        // the exploration key is the GLOBAL env id (env_id_base + b), agent (and receiver): shards draw what the whole job draws
        size_t o;
        uint32_t key;
        bool diag = false;
        if (!pairs) { const int i = r / B, b = r - i * B; o = (size_t)b * n_agents + i; key = (env_id_base + (uint32_t)b) * n + (uint32_t)i; }
        else { const int i = r / (B * n_agents), rem = r - i * B * n_agents, b = rem / n_agents, j = rem - b * n_agents;
               o = ((size_t)b * n_agents + i) * n_agents + j; diag = i == j;
               key = ((env_id_base + (uint32_t)b) * n + (uint32_t)i) * n + (uint32_t)j; }
        int act = dueling_pick_row(a, a[A], A, avail, eps, step, seed, key, q_out ? q_out + (size_t)r * A : nullptr);
        if (diag) act = 0;
        actions[o] = act;
    }
}

void launch_encoder(const float* obs, int rows, int V, const float* cw, const float* cb, const float* lw, const float* lb, float* out,
                    int out_stride, int n_agents, int agent_major, float* store, long store_env_stride, const int64_t* store_t,
                    hipStream_t s) {
    constexpr int RPW = 1;
    const size_t lds = 4 * RPW * (size_t)(3 * V * V + 4) * sizeof(float);
    const int rows_per_block = 4 * RPW;
    hipLaunchKernelGGL((k_encoder<6, 32, RPW>), dim3((rows + rows_per_block - 1) / rows_per_block), dim3(256), lds, s, obs, rows, V, cw, cb,
                       lw, lb, out, out_stride, n_agents, agent_major, store, store_env_stride, store_t);
}
void launch_conv_leaky(const float* obs, int rows, int V, const float* cw, const float* cb, float* out, int n_agents, int agent_major,
                       float* store, long store_env_stride, const int64_t* store_t, hipStream_t s) {
    const size_t lds = (size_t)(3 * V * V + 6 * 27 + 6) * sizeof(float);
    if (V == 15) hipLaunchKernelGGL((k_conv_leaky<6, 15>), dim3(rows), dim3(256), lds, s, obs, rows, V, cw, cb, out, n_agents, agent_major, store, store_env_stride, store_t);
    else if (V == 31) hipLaunchKernelGGL((k_conv_leaky<6, 31>), dim3(rows), dim3(256), lds, s, obs, rows, V, cw, cb, out, n_agents, agent_major, store, store_env_stride, store_t);
    else hipLaunchKernelGGL((k_conv_leaky<6, 0>), dim3(rows), dim3(256), lds, s, obs, rows, V, cw, cb, out, n_agents, agent_major, store, store_env_stride, store_t);
}
void launch_store_step(const ssd_store_step* a, hipStream_t s) {
    StoreStep k;
    k.t = a->t_index; k.N = a->n_env; k.n = a->n_agents; k.A = a->n_actions; k.slots = a->t_slots;
    k.pos = a->pos; k.orient = a->orient; k.reward = a->reward; k.clean = a->clean_num; k.den = a->apple_den; k.term = a->terminated;
    k.actions = a->actions; k.actions_inc = a->actions_inc;
    k.d_pos = a->dst_pos; k.d_orient = a->dst_orient; k.d_reward = a->dst_reward; k.d_clean = a->dst_clean_num; k.d_den = a->dst_apple_den;
    k.d_onehot = a->dst_actions_onehot; k.d_term = a->dst_terminated; k.d_actions = a->dst_actions; k.d_actions_inc = a->dst_actions_inc;
    k.p_act = a->prev_actions; k.p_inc = a->prev_actions_inc; k.p_rew = a->prev_reward; k.ep_ret = a->ep_return;
    k.next_t = a->next_t_out; k.ctr_inc = a->counter_inc;
    const long total = (long)k.N * (k.n * (8 + k.A + k.n) + 1);
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_store_step, dim3(blocks), dim3(256), 0, s, k);
}
void launch_gru_gates(const float* gi, const float* gh, float* h, int R, int H, hipStream_t s) {
    size_t total = (size_t)R * H; int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gru_gates, dim3(blocks), dim3(256), 0, s, gi, gh, h, R, H);
}
void launch_gru_fwd_train(const float* gi, const float* gh, const float* h, float* h_new, float* rzn, int R, int H, hipStream_t s) {
    size_t total = (size_t)R * H; int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gru_fwd_train, dim3(blocks), dim3(256), 0, s, gi, gh, h, h_new, rzn, R, H);
}
void launch_gru_bwd(const float* dh, const float* rzn, const float* gh, const float* h, float* d_gi, float* d_gh, float* dh_prev, int R,
                    int H, hipStream_t s) {
    size_t total = (size_t)R * H; int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gru_bwd, dim3(blocks), dim3(256), 0, s, dh, rzn, gh, h, d_gi, d_gh, dh_prev, R, H);
}
void launch_dueling_pick(const float* av, int R, int A, const uint8_t* avail, const float* eps, const int64_t* step, uint32_t seed,
                         int n_agents, int B, int pairs, int64_t* actions, float* q_out, uint32_t env_id_base, hipStream_t s) {
    int blocks = (R + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_dueling_pick, dim3(blocks), dim3(256), 0, s, av, R, A, avail, eps, step, seed, n_agents, B, pairs, actions, q_out, env_id_base);
}

}  // namespace ssd
