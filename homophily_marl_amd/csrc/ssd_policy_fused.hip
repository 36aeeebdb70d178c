// ssd_policy_fused.hip -- the rollout-time controller step as ONE launch per head (include/ssd_hip.h: ssd_policy_head_*).
//
// What is evaluated (reference): HomophilyMAC._build_inputs tail (homophily_controller.py:137-184), HomophilyAgent.forward_env /
// forward_inc (homophily_agent.py:154-208: fc1 -> LeakyReLU -> hand-written GRU cell -> dueling Q) and the epsilon-greedy
// selector (action_selectors.py:44-68).  Every agent has its own weights, so the work is n independent small-matrix chains
// over n_env rows each: [N, 64] x [64, 64] -> [N, 64] x [64, 192] (+ [N, 64] x [64, 192]) -> [N, 64] x [64, 16].
//
// Mapping to CDNA4:
//   * one workgroup (8 waves) per (agent, slice of envs); the agent's whole weight image (126 KB incl. padding) is staged in
//     LDS once per workgroup and every wave then walks 16-row tiles of that agent's envs -- weights cross L2 -> CU once per
//     workgroup instead of once per GEMM and tile;
//   * all products are computed TRANSPOSED, D^T = W^T x X^T, with v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf chain):
//     the weight is the A operand (lane l supplies W^T[out = l & 15][k]), the activations are the B operand (lane l supplies
//     X[row = l & 15][k]), and the result tile puts the activation ROW on the lane (l & 15) and 4 consecutive OUTPUT features
//     (4 (l >> 4) + reg) in the lane's registers.  With the k index of MFMA step (c, r) chosen as 16 c + 4 (l >> 4) + r this is
//     exactly the B-operand layout of the next product, so fc1 -> GRU -> fc2 chain in registers with no LDS round trip, no
//     shuffles, and the GRU gate arithmetic is lane-local (r, z, n of one feature sit in the same lane and register index);
//   * the same k order makes every operand fetch a 16-byte access: ds_read_b128 of W^T rows (row stride 68 floats: conflict
//     free) feeds 4 MFMA steps, the activations are read / written as float4 per lane.
// Launch: grid = n_agents * blocks_per_agent, 512 threads; every wave runs a bounded tile loop, no cross-wave dependencies
// after the weight staging barrier.
#include "ssd_policy_common.h"

namespace ssd {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Diagnostic build (-DSSD_STAMPS, tools/pstamps.py): per-wave s_memtime stamps of the kernel phases, [wave][16] u64.
#ifdef SSD_STAMPS
static unsigned long long* g_policy_stamps = nullptr;
#define PSTAMP_DECL unsigned long long* stamps;
#define PSTAMP_SET(k) (k).stamps = g_policy_stamps
#define PSTAMP(i) do { if (a.stamps && lane == 0) a.stamps[(size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define PNOW() __builtin_amdgcn_s_memtime()
#define PSTAMP_VAL(i, v) do { if (a.stamps && lane == 0) a.stamps[(size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * 16 + (i)] = (v); } while (0)
#else
#define PSTAMP_DECL
#define PSTAMP_SET(k)
#define PSTAMP(i)
#define PNOW() 0ull
#define PSTAMP_VAL(i, v)
#endif

constexpr int WS = 68;                         // LDS row stride of the transposed weights (floats)
constexpr int ROW_FC1 = 0, ROW_WI = 64, ROW_WH = 256, ROW_FC2 = 448, W_ROWS = 464;
constexpr int OFF_BIAS = W_ROWS * WS;          // fc1[64] | gru_i[192] | gru_h[192] | fc2[16]
constexpr int OFF_B1 = OFF_BIAS, OFF_BI = OFF_BIAS + 64, OFF_BH = OFF_BIAS + 256, OFF_B2 = OFF_BIAS + 448;
constexpr int OFF_W2O = OFF_BIAS + 464;        // inc: pair part of fc2, [16][4]
constexpr int IMAGE = OFF_W2O + 64;
static_assert(IMAGE == SSD_POLICY_IMAGE_FLOATS, "image layout out of sync with include/ssd_hip.h");
constexpr int HEAD_WAVES = 8;
constexpr int SCRATCH = 16 * 16;               // per wave: fc2 output tile [row 16][out 16]

struct HeadK {
    int N, n, A, inp, bpa;
    float pos_scale;
    uint32_t seed;
    float* inputs;
    float* h;
    const float* weights;
    const uint8_t* avail;
    const float* eps;
    const int64_t* step;
    const int64_t* prev_actions;
    const float* prev_reward;
    const int64_t* prev_inc;
    const float* pos;
    const int64_t* actions;
    const float *pos_pre, *orient_pre, *reward, *clean, *den;
    int64_t* out_actions;
    float* q_out;
    const float* orient;
    int32_t* out_actions_i32;
    float *pos_copy, *orient_copy;
    // filing into the episode storage (slot *t_index) and the runner state carried to the next step
    const int64_t* t_index; int slots;
    float *d_pos, *d_orient, *d_onehot, *d_reward, *d_clean, *d_den;
    uint8_t* d_term; const uint8_t* term;
    int64_t *d_actions, *d_actions_inc, *p_act, *p_inc;
    float *p_rew, *ep_ret;
    int64_t* next_t;
    PSTAMP_DECL
};

// acc[ot] += W^T[16 ot .. 16 ot + 15][:] x B   for OT output tiles; wt points at the first weight row of the block in LDS.
template <int OT>
__device__ __forceinline__ void gemm_t(const float* wt, const f32x4 (&B)[4], f32x4* acc, int m, int q) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        f32x4 a[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) a[ot] = *reinterpret_cast<const f32x4*>(wt + (16 * ot + m) * WS + 16 * ct + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ot][r], B[ct][r], acc[ot], 0, 0, 0);
    }
}

// sigmoid / tanh on v_exp_f32 + v_rcp_f32 (1 ulp each): ~2e-7 from the libm forms used by the per-layer path
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fminf(fabsf(x), 15.f);                            // 1 - 2 / (e^{2|x|} + 1), saturated
    const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * ax) + 1.f);
    return copysignf(t, x);
}

template <int INC>
__global__ __launch_bounds__(HEAD_WAVES * 64) void k_head(HeadK a) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int agent = blockIdx.x / a.bpa, bia = blockIdx.x - agent * a.bpa;
    const int m = lane & 15, q = lane >> 4;
    const int N = a.N, n = a.n, A = a.A;
    PSTAMP(0);
    {   // stage this agent's weight image (already in LDS layout): every load in flight before the first LDS write
        const f32x4* src = reinterpret_cast<const f32x4*>(a.weights + (size_t)agent * IMAGE);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        constexpr int PER = (IMAGE / 4 + HEAD_WAVES * 64 - 1) / (HEAD_WAVES * 64);
        f32x4 tmp[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int e = tid + j * HEAD_WAVES * 64; if (e < IMAGE / 4) tmp[j] = src[e]; }
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int e = tid + j * HEAD_WAVES * 64; if (e < IMAGE / 4) dst[e] = tmp[j]; }
    }
    __syncthreads();
    PSTAMP(1);
    bool first = true;
    float* scratch = lds + IMAGE + wave * SCRATCH;
    const float eps = *a.eps;
    const uint32_t step = (uint32_t)*a.step;
    const long slot_t = a.t_index ? (long)*a.t_index : 0;
    if (INC && a.next_t && blockIdx.x == 0 && tid == 0) *a.next_t = slot_t + 1;   // not read by this kernel (t_index is a copy)
    const int tiles = (N + 15) >> 4;
    for (int tile = wave * a.bpa + bia; tile < tiles; tile += a.bpa * HEAD_WAVES) {   // consecutive tiles go to different CUs
        const int b = tile * 16 + m;
        const bool valid = b < N;
        const int bc = valid ? b : N - 1;
        const size_t arow = (size_t)agent * N + bc;                    // agent-major row
        float* in_row = a.inputs + arow * 64;
        // ---- B operand of fc1: the 64 (zero padded) input features of row m, 4 per (ct, lane) -------------------------
        f32x4 x[4];
        if (!INC) {
            x[0] = *reinterpret_cast<const f32x4*>(in_row + 4 * q);
            x[1] = *reinterpret_cast<const f32x4*>(in_row + 16 + 4 * q);
            const size_t er = (size_t)bc * n + agent;                  // env-major row
            const int pa = (int)a.prev_actions[er];
            const float pr = a.prev_reward[er];
            int recv = 0;
            for (int g = 0; g < n; ++g) {
                if (g == agent) continue;                              // inc_mask_actions: no self incentive
                const int64_t v = a.prev_inc[((size_t)bc * n + g) * n + agent];
                recv += (v == 1) - (v == 2);
            }
            const float px = a.pos[er * 2] / a.pos_scale, py = a.pos[er * 2 + 1] / a.pos_scale;
            if (valid && q == 0 && (a.pos_copy || a.d_pos)) {          // the pose BEFORE the env step (inc head input, storage slot t)
                const float p0 = a.pos[er * 2], p1 = a.pos[er * 2 + 1], o0 = a.orient[er * 2], o1 = a.orient[er * 2 + 1];
                if (a.pos_copy) { a.pos_copy[er * 2] = p0; a.pos_copy[er * 2 + 1] = p1; a.orient_copy[er * 2] = o0; a.orient_copy[er * 2 + 1] = o1; }
                if (a.d_pos) {
                    const size_t sr = (((size_t)bc * a.slots + slot_t) * n + agent) * 2;
                    a.d_pos[sr] = p0; a.d_pos[sr + 1] = p1; a.d_orient[sr] = o0; a.d_orient[sr + 1] = o1;
                }
            }
#pragma unroll
            for (int ct = 2; ct < 4; ++ct) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * ct + 4 * q + r - 32;            // tail column (controller :137-184)
                    float v = 0.f;
                    if (j < A) v = pa == j ? 1.f : 0.f;
                    else if (j < A + n) v = j - A == agent ? 1.f : 0.f;
                    else if (j == A + n) v = (float)((pr > 0.f) - (pr < 0.f));
                    else if (j == A + n + 1) v = (float)((recv > 0) - (recv < 0));
                    else if (j == A + n + 2) v = px;
                    else if (j == A + n + 3) v = py;
                    x[ct][r] = v;
                }
                if (valid) *reinterpret_cast<f32x4*>(in_row + 16 * ct + 4 * q) = x[ct];   // the inc head reads the full row
            }
        } else {
            const int act = (int)a.actions[(size_t)bc * n + agent];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                x[ct] = *reinterpret_cast<const f32x4*>(in_row + 16 * ct + 4 * q);
                if (ct >= 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 16 * ct + 4 * q + r - a.inp;     // [inputs | one-hot(action)] (homophily_agent.py:181)
                        if (k >= 0 && k < A) x[ct][r] = act == k ? 1.f : 0.f;
                    }
                }
            }
        }
        if (first) PSTAMP(2);
        // ---- fc1 + LeakyReLU -----------------------------------------------------------------------------------------------
        f32x4 x1[4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) x1[ot] = *reinterpret_cast<const f32x4*>(lds + OFF_B1 + 16 * ot + 4 * q);
        gemm_t<4>(lds + ROW_FC1 * WS, x, x1, m, q);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) x1[ot][r] = leaky(x1[ot][r]);
        if (first) PSTAMP(3);
        // ---- GRU cell: r, z share one accumulator for the input and the hidden side; n needs both separately -------------------
        float* h_row = a.h + arow * 64;
        f32x4 hp[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) hp[ct] = *reinterpret_cast<const f32x4*>(h_row + 16 * ct + 4 * q);
        f32x4 g[16];                                                   // 0-3 r, 4-7 z, 8-11 i_n, 12-15 h_n
#pragma unroll
        for (int ot = 0; ot < 8; ++ot)
            g[ot] = *reinterpret_cast<const f32x4*>(lds + OFF_BI + 16 * ot + 4 * q) + *reinterpret_cast<const f32x4*>(lds + OFF_BH + 16 * ot + 4 * q);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            g[8 + ot] = *reinterpret_cast<const f32x4*>(lds + OFF_BI + 128 + 16 * ot + 4 * q);
            g[12 + ot] = *reinterpret_cast<const f32x4*>(lds + OFF_BH + 128 + 16 * ot + 4 * q);
        }
        gemm_t<8>(lds + ROW_WI * WS, x1, g, m, q);
        gemm_t<4>(lds + (ROW_WI + 128) * WS, x1, g + 8, m, q);
        if (first) PSTAMP(4);
        gemm_t<8>(lds + ROW_WH * WS, hp, g, m, q);
        gemm_t<4>(lds + (ROW_WH + 128) * WS, hp, g + 12, m, q);
        if (first) PSTAMP(5);
        f32x4 hn[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rg = sigmoid_fast(g[ft][r]);
                const float zg = sigmoid_fast(g[4 + ft][r]);
                const float ng = tanh_fast(g[8 + ft][r] + rg * g[12 + ft][r]);
                hn[ft][r] = (1.f - zg) * ng + zg * hp[ft][r];
            }
            if (valid) *reinterpret_cast<f32x4*>(h_row + 16 * ft + 4 * q) = hn[ft];
        }
        if (first) PSTAMP(6);
        // ---- fc2 (advantages + value, padded to 16 outputs) ---------------------------------------------------------------
        f32x4 o2 = *reinterpret_cast<const f32x4*>(lds + OFF_B2 + 4 * q);
        gemm_t<1>(lds + ROW_FC2 * WS, hn, &o2, m, q);
        *reinterpret_cast<f32x4*>(scratch + m * 16 + 4 * q) = o2;      // scratch[row][out]
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!INC) {
            const int bb = tile * 16 + lane;
            if (lane < 16 && bb < N) {
                float av[16];
                for (int k = 0; k <= A; ++k) av[k] = scratch[lane * 16 + k];
                const uint32_t r = (uint32_t)(agent * N + bb);
                const int act = dueling_pick_row(av, av[A], A, a.avail, eps, step, a.seed, r,
                                                 a.q_out ? a.q_out + (size_t)r * A : nullptr);
                a.out_actions[(size_t)bb * n + agent] = act;
                if (a.out_actions_i32) a.out_actions_i32[(size_t)bb * n + agent] = act;
                if (a.p_act) a.p_act[(size_t)bb * n + agent] = act;
                if (a.d_actions) {
                    const size_t sr = ((size_t)bb * a.slots + slot_t) * n + agent;
                    a.d_actions[sr] = act;
                    for (int k = 0; k < A; ++k) a.d_onehot[sr * A + k] = k == act ? 1.f : 0.f;
                }
            }
        } else {
            const float* w2o = lds + OFF_W2O;                          // [E][4]: 3 advantages + value per extra feature
            for (int it = lane; it < 16 * n; it += 64) {
                const int row = it / n, j = it - row * n, bb = tile * 16 + row;
                if (bb >= N) continue;
                const size_t ej = (size_t)bb * n + j;
                // other_j = [one-hot(a_j), pos_j / scale, orient_j, r_j, clean_j, apple_den_j] (homophily_agent.py:194-201)
                const int aj = (int)a.actions[ej];
                float f[7];
                f[0] = a.pos_pre[ej * 2] / a.pos_scale; f[1] = a.pos_pre[ej * 2 + 1] / a.pos_scale;
                f[2] = a.orient_pre[ej * 2]; f[3] = a.orient_pre[ej * 2 + 1];
                f[4] = a.reward[ej]; f[5] = a.clean[ej]; f[6] = a.den[ej];
                float av[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float s = scratch[row * 16 + o] + w2o[aj * 4 + o];
#pragma unroll
                    for (int e = 0; e < 7; ++e) s = fmaf(f[e], w2o[(A + e) * 4 + o], s);
                    av[o] = s;
                }
                const uint32_t r = (uint32_t)((agent * N + bb) * n + j);
                int act = dueling_pick_row(av, av[3], 3, nullptr, eps, step, a.seed, r, a.q_out ? a.q_out + (size_t)r * 3 : nullptr);
                if (j == agent) act = 0;                               // no self incentive (homophily_controller.py:44-46)
                a.out_actions[((size_t)bb * n + agent) * n + j] = act;
                if (a.p_inc) a.p_inc[((size_t)bb * n + agent) * n + j] = act;
                const size_t sr = ((size_t)bb * a.slots + slot_t) * n + agent;
                if (a.d_actions_inc) a.d_actions_inc[sr * n + j] = act;
                if (j == 0 && a.d_reward) {                            // once per (env, agent): this step's outcome
                    const size_t ea = (size_t)bb * n + agent;
                    const float rw = a.reward[ea];
                    a.d_reward[sr] = rw; a.d_clean[sr] = a.clean[ea]; a.d_den[sr] = a.den[ea];
                    if (a.p_rew) a.p_rew[ea] = rw;
                    if (a.ep_ret) a.ep_ret[ea] += rw;
                    if (agent == 0 && a.d_term) a.d_term[(size_t)bb * a.slots + slot_t] = a.term[bb];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                               // scratch is reused by the next tile
        if (first) PSTAMP(7);
        first = false;
    }
    PSTAMP(8);
}

static int launch_head(const ssd_policy_head* p, int inc, hipStream_t s) {
    HeadK k;
    k.N = p->n_env; k.n = p->n_agents; k.A = p->n_actions; k.inp = p->input_shape;
    k.pos_scale = p->pos_scale; k.seed = p->seed;
    k.inputs = p->inputs; k.h = p->h; k.weights = p->weights; k.avail = p->avail; k.eps = p->epsilon; k.step = p->step;
    k.prev_actions = p->prev_actions; k.prev_reward = p->prev_reward; k.prev_inc = p->prev_actions_inc; k.pos = p->pos;
    k.actions = p->actions; k.pos_pre = p->pos_pre; k.orient_pre = p->orient_pre; k.reward = p->reward; k.clean = p->clean_num;
    k.den = p->apple_den; k.out_actions = p->out_actions; k.q_out = p->q_out;
    PSTAMP_SET(k);
    k.t_index = p->t_index; k.slots = p->t_slots;
    k.d_pos = p->dst_pos; k.d_orient = p->dst_orient; k.d_onehot = p->dst_actions_onehot; k.d_reward = p->dst_reward;
    k.d_clean = p->dst_clean_num; k.d_den = p->dst_apple_den; k.d_term = p->dst_terminated; k.term = p->terminated;
    k.d_actions = p->dst_actions; k.d_actions_inc = p->dst_actions_inc; k.p_act = p->prev_actions_out; k.p_inc = p->prev_actions_inc_out;
    k.p_rew = p->prev_reward_out; k.ep_ret = p->ep_return; k.next_t = p->next_t_out;
    k.orient = p->orient; k.out_actions_i32 = p->out_actions_i32; k.pos_copy = p->pos_copy; k.orient_copy = p->orient_copy;
    const int tiles = (k.N + 15) / 16;
    int bpa = 256 / k.n;                                               // one workgroup per CU (the image fills most of the LDS)
    if (bpa > tiles) bpa = tiles;
    if (bpa < 1) bpa = 1;
    k.bpa = bpa;
    const size_t lds = (size_t)(IMAGE + HEAD_WAVES * SCRATCH) * sizeof(float);
    static bool attr_done_dev[64] = {};                               // the attribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    bool& attr_done = attr_done_dev[dev];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (inc) hipLaunchKernelGGL(k_head<1>, dim3(k.n * bpa), dim3(HEAD_WAVES * 64), lds, s, k);
    else hipLaunchKernelGGL(k_head<0>, dim3(k.n * bpa), dim3(HEAD_WAVES * 64), lds, s, k);
    return 0;
}

int launch_policy_head(const ssd_policy_head* p, int inc, hipStream_t s) { return launch_head(p, inc, s); }
#ifdef SSD_STAMPS
void set_policy_stamps(unsigned long long* buf) { g_policy_stamps = buf; }
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// Encoder: Conv2d(3, 6, 3, 1) + LeakyReLU + Flatten + Linear(6 (V-2)^2, 32) + LeakyReLU (homophily_agent.py:20-27,213-214) in one
// launch; the 4 KB/row conv activations never leave the registers (the unfused form writes and re-reads 83 MB per timestep).
//   * workgroup = 4 waves = 16 observation rows, staged in LDS; the rows are read where the env kernel wrote them -- a dense
//     [rows, 3, V, V] buffer or directly time slot t of the episode storage [n_env, t_slots, n, 3, V, V] (no copy of obs);
//   * after the staging barrier the waves are independent.  The K = 6 * 169 reduction of the Linear layer is cut into groups of
//     16 conv positions per channel; wave w takes the groups c = w, w + 4, w + 8 of every channel.  For a group, lane
//     (row = l & 15, q = l >> 4) computes the conv outputs of ITS row at the 4 positions 16 c + 4 q + r for all 6 channels on the
//     VALU (27 taps from LDS, weights through the scalar cache) -- which is exactly the B-operand layout of
//     v_mfma_f32_16x16x4_f32 (transposed product as in k_head), so the activations go from the FMA result registers straight
//     into the matrix core; the A operand (Linear weight slice, zero-padded [6][32][176] repack) streams from L2 as float4 and is
//     requested before the group's FMAs;
//   * the four partial [32, 16] results are added in a fixed order through LDS (deterministic).
// ---------------------------------------------------------------------------------------------------------------------------
struct EncK {
    const float* obs; int rows;
    float* out; int out_stride, n, agent_major;
    long env_stride, slot_stride; const int64_t* slot_t;   // row (b, i) = obs + b * env_stride + *slot_t * slot_stride + i * 3VV
    int64_t* slot_t_copy;
    int64_t* counter_inc;
    int code;                           // obs holds u8 class codes [.., V, V] (SSD_OBS_CODE) instead of f32 [.., 3, V, V]
    PSTAMP_DECL
};

template <int V>
__global__ __launch_bounds__(256, 3) void k_encode(EncK a, const float* __restrict__ cw, const float* __restrict__ cb,
                                                    const float* __restrict__ lwp, const float* __restrict__ lb) {
    // the weights are separate __restrict__ kernel arguments so that the wave-uniform conv weights come through the scalar cache
    constexpr int O = V - 2, P = O * O, VV = V * V, L = 3 * VV, PG = (P + 15) / 16, PP = PG * 16;
    extern __shared__ float lds[];
    float* tile = lds;                  // [16][L] observation rows; reused for the cross-wave reduction at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const int nrows = a.rows - row0 < 16 ? a.rows - row0 : 16;
    PSTAMP(0);
    {   // stage the 16 rows: dword loads (rows are 4-byte aligned only), all in flight before the first LDS write
        const long t_off = a.slot_t ? (long)(*a.slot_t) * a.slot_stride : 0;
        if (blockIdx.x == 0 && tid == 0) {
            if (a.slot_t_copy) *a.slot_t_copy = *a.slot_t;
            if (a.counter_inc) *a.counter_inc += 1;
        }
        if (a.code) {
            // compact storage: one class code per cell (0 nothing, 1 apple, 2 waste, 3 wall-or-agent; simplified palette) is expanded
            // to the three colour planes on the way into LDS -- waste = R, apple = G, wall / agent = B at 255/256 (map_env.py:945)
            const uint8_t* codes = reinterpret_cast<const uint8_t*>(a.obs);
            constexpr int CPT = (16 * VV + 255) / 256;
            uint8_t cv[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int idx = tid + 256 * j, r = idx / VV, cell = idx - r * VV;
                cv[j] = 0;
                if (idx < 16 * VV && r < nrows) {
                    const int row = row0 + r, b = row / a.n, i = row - b * a.n;
                    cv[j] = codes[(long)b * a.env_stride + t_off + (long)i * VV + cell];
                }
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int idx = tid + 256 * j, r = idx / VV, cell = idx - r * VV;
                if (idx < 16 * VV) {
                    const float on = 255.0f / 256.0f;
                    tile[r * L + cell] = cv[j] == 2 ? on : 0.f;
                    tile[r * L + VV + cell] = cv[j] == 1 ? on : 0.f;
                    tile[r * L + 2 * VV + cell] = cv[j] == 3 ? on : 0.f;
                }
            }
        } else {
        // row r of the tile = (env b, agent i): walk (b, i) incrementally, one address per row (wave-uniform arithmetic)
        constexpr int PR = (L + 255) / 256, FULLJ = L / 256;          // loads per thread and row; the first FULLJ need no bound test
        float tmp[16][PR];
        const float* src[16];
        {
            int b = row0 / a.n, i = row0 - b * a.n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                src[r] = a.obs + (long)b * a.env_stride + t_off + (long)i * L;
                if (r + 1 < nrows) { if (++i == a.n) { i = 0; ++b; } }      // rows past the end re-read the last valid one (discarded)
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < FULLJ; ++j) tmp[r][j] = src[r][tid + 256 * j];
        if (PR > FULLJ) {                                              // the ragged last load of every row under ONE predicate
            const bool in = tid + 256 * FULLJ < L;
#pragma unroll
            for (int r = 0; r < 16; ++r) tmp[r][FULLJ] = in ? src[r][tid + 256 * FULLJ] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < PR; ++j) { const int e = tid + 256 * j; if (j < FULLJ || e < L) tile[r * L + e] = r < nrows ? tmp[r][j] : 0.f; }
        }
    }
    __syncthreads();
    PSTAMP(1);
    f32x4 mac[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
    for (int c = wave; c < PG; c += 4) {
        // Linear weight slices of this group, all 6 channels: requested now, consumed after the group's FMAs
        f32x4 af[6][2];
#pragma unroll
        for (int oc = 0; oc < 6; ++oc)
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
                af[oc][ft] = *reinterpret_cast<const f32x4*>(lwp + ((size_t)(oc * 32 + 16 * ft + m) * PP + 16 * c + 4 * q));
        int base[4];
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = 16 * c + 4 * q + r;
            ok[r] = p < P;
            const int pc = ok[r] ? p : 0, y = pc / O, x = pc - y * O;
            base[r] = m * L + y * V + x;
        }
        float acc[6][4];
#pragma unroll
        for (int oc = 0; oc < 6; ++oc)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[oc][r] = cb[oc];
#pragma unroll 1
        for (int ch = 0; ch < 3; ++ch) {                               // not unrolled: 54 scalar weights live at a time
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = tile[base[r] + ch * VV + dy * V + dx];
#pragma unroll
                    for (int oc = 0; oc < 6; ++oc) {
                        const float w = cw[((oc * 3 + ch) * 3 + dy) * 3 + dx];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[oc][r] = fmaf(w, v[r], acc[oc][r]);
                    }
                }
        }
#pragma unroll
        for (int oc = 0; oc < 6; ++oc) {
            f32x4 bop;
#pragma unroll
            for (int r = 0; r < 4; ++r) bop[r] = ok[r] ? leaky(acc[oc][r]) : 0.f;   // K padding: activation 0 (x weight pad 0)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) mac[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[oc][ft][r], bop[r], mac[ft], 0, 0, 0);
        }
    }
    PSTAMP(2);
    __syncthreads();                                                   // every wave is done with the tile: reuse it for the reduction
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    red[(wave * 2 + 0) * 64 + lane] = mac[0];
    red[(wave * 2 + 1) * 64 + lane] = mac[1];
    __syncthreads();
    if (wave == 0 && m < nrows) {
        const int row = row0 + m, b = row / a.n, i = row - b * a.n;
        const size_t orow = a.agent_major ? (size_t)i * (a.rows / a.n) + b : (size_t)row;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            f32x4 s = red[(0 * 2 + ft) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) s += red[(w * 2 + ft) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) a.out[orow * a.out_stride + 16 * ft + 4 * q + r] = leaky(s[r] + lb[16 * ft + 4 * q + r]);
        }
    }
    PSTAMP(3);
}

int launch_policy_encode(const float* obs, int rows, int V, const float* cw, const float* cb, const float* lwp, const float* lb, float* out,
                         int out_stride, int n_agents, int agent_major, long env_stride, long slot_stride, const int64_t* slot_t,
                         int64_t* slot_t_copy, int64_t* counter_inc, int code, hipStream_t s) {
    if (V != 15) return -2;
    constexpr int L = 3 * 15 * 15;
    const size_t lds = (size_t)(16 * L) * sizeof(float);
    static bool attr_done_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (!attr_done_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode<15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        attr_done_dev[dev] = true;
    }
    EncK k{obs, rows, out, out_stride, n_agents, agent_major, env_stride ? env_stride : (long)n_agents * (code ? L / 3 : L), slot_stride, slot_t, slot_t_copy, counter_inc, code};
    PSTAMP_SET(k);
    hipLaunchKernelGGL(k_encode<15>, dim3((rows + 15) / 16), dim3(256), lds, s, k, cw, cb, lwp, lb);
    return 0;
}

}  // namespace ssd
