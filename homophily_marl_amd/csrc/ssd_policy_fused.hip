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

__device__ __forceinline__ float sigmoid_fast(float x) { return 1.f / (1.f + __expf(-x)); }   // as k_gru_gates

template <int INC>
__global__ __launch_bounds__(HEAD_WAVES * 64) void k_head(HeadK a) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int agent = blockIdx.x / a.bpa, bia = blockIdx.x - agent * a.bpa;
    const int m = lane & 15, q = lane >> 4;
    const int N = a.N, n = a.n, A = a.A;
    {   // stage this agent's weight image (already in LDS layout) -- flat 16-byte copy
        const f32x4* src = reinterpret_cast<const f32x4*>(a.weights + (size_t)agent * IMAGE);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int e = tid; e < IMAGE / 4; e += HEAD_WAVES * 64) dst[e] = src[e];
    }
    __syncthreads();
    float* scratch = lds + IMAGE + wave * SCRATCH;
    const float eps = *a.eps;
    const uint32_t step = (uint32_t)*a.step;
    const int tiles = (N + 15) >> 4;
    for (int tile = bia * HEAD_WAVES + wave; tile < tiles; tile += a.bpa * HEAD_WAVES) {
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
        // ---- fc1 + LeakyReLU -----------------------------------------------------------------------------------------------
        f32x4 x1[4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) x1[ot] = *reinterpret_cast<const f32x4*>(lds + OFF_B1 + 16 * ot + 4 * q);
        gemm_t<4>(lds + ROW_FC1 * WS, x, x1, m, q);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) x1[ot][r] = leaky(x1[ot][r]);
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
        gemm_t<8>(lds + ROW_WH * WS, hp, g, m, q);
        gemm_t<4>(lds + (ROW_WH + 128) * WS, hp, g + 12, m, q);
        f32x4 hn[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rg = sigmoid_fast(g[ft][r]);
                const float zg = sigmoid_fast(g[4 + ft][r]);
                const float ng = tanhf(g[8 + ft][r] + rg * g[12 + ft][r]);
                hn[ft][r] = (1.f - zg) * ng + zg * hp[ft][r];
            }
            if (valid) *reinterpret_cast<f32x4*>(h_row + 16 * ft + 4 * q) = hn[ft];
        }
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
            }
        }
        __builtin_amdgcn_wave_barrier();                               // scratch is reused by the next tile
    }
}

static int launch_head(const ssd_policy_head* p, int inc, hipStream_t s) {
    HeadK k;
    k.N = p->n_env; k.n = p->n_agents; k.A = p->n_actions; k.inp = p->input_shape;
    k.pos_scale = p->pos_scale; k.seed = p->seed;
    k.inputs = p->inputs; k.h = p->h; k.weights = p->weights; k.avail = p->avail; k.eps = p->epsilon; k.step = p->step;
    k.prev_actions = p->prev_actions; k.prev_reward = p->prev_reward; k.prev_inc = p->prev_actions_inc; k.pos = p->pos;
    k.actions = p->actions; k.pos_pre = p->pos_pre; k.orient_pre = p->orient_pre; k.reward = p->reward; k.clean = p->clean_num;
    k.den = p->apple_den; k.out_actions = p->out_actions; k.q_out = p->q_out;
    const int tiles = (k.N + 15) / 16;
    int bpa = 256 / k.n;                                               // one workgroup per CU (the image fills most of the LDS)
    const int need = (tiles + HEAD_WAVES - 1) / HEAD_WAVES;
    if (bpa > need) bpa = need;
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

// ---------------------------------------------------------------------------------------------------------------------------
// Encoder: Conv2d(3, 6, 3, 1) + LeakyReLU + Flatten + Linear(6 (V-2)^2, 32) + LeakyReLU (homophily_agent.py:20-27,213-214) in one
// launch, the 4 KB/row conv activations never leaving the CU (the unfused form writes and re-reads 83 MB per timestep).
//   * workgroup = 4 waves = 16 observation rows; the rows (contiguous in HBM) are staged in LDS with 16-byte loads and, on the
//     way, copied into the episode storage obs[b, t];
//   * conv on the VALU: thread = (row, output line y) keeps its 3 x 3 x V input window in registers and produces the V-2 outputs
//     of the line for one output channel at a time (27 scalar-operand FMAs per output, weights through the scalar cache);
//   * per output channel the [16 rows, (V-2)^2] activations go to a double-buffered LDS chunk and are contracted with the
//     matching slice of the Linear weight by v_mfma_f32_16x16x4_f32 (transposed product as in k_head: weight = A operand,
//     streamed from L2 as float4 from a zero-padded [6][32][176] repack; activations = B operand, ds_read_b128); the four waves
//     split the K groups and their partial sums are added in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------------------------------
struct EncK {
    const float* obs; int rows;
    const float *cw, *cb, *lwp, *lb;
    float* out; int out_stride, n, agent_major;
    float* store; long store_env_stride; const int64_t* store_t;
};

template <int V>
__global__ __launch_bounds__(256, 2) void k_encode(EncK a) {
    constexpr int O = V - 2, P = O * O, L = 3 * V * V, PG = (P + 15) / 16, PP = PG * 16, CS = PP + 4, ITEMS = 16 * O;
    static_assert(ITEMS <= 256 && (16 * L) % 4 == 0 && CS % 4 == 0 && (CS % 32) != 0, "tile shape");
    extern __shared__ float lds[];
    float* tile = lds;                  // [16][L], contiguous like the source rows
    float* cbuf = lds + 16 * L;         // [2][16][CS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const int nrows = a.rows - row0 < 16 ? a.rows - row0 : 16;
    {
        const float* src = a.obs + (size_t)row0 * L;
        const int total = nrows * L;
        for (int e4 = tid; e4 < 16 * L / 4; e4 += 256) {
            f32x4 v;
            if (4 * e4 + 3 < total) v = *reinterpret_cast<const f32x4*>(src + 4 * e4);
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 4 * e4 + r < total ? src[4 * e4 + r] : 0.f;
            *reinterpret_cast<f32x4*>(tile + 4 * e4) = v;
        }
        for (int e = tid; e < 2 * 16 * (CS - P); e += 256) {          // K padding of the chunks: zero (0 x weight pad 0)
            const int rr = e / (CS - P), j = e - rr * (CS - P);
            cbuf[rr * CS + P + j] = 0.f;
        }
    }
    __syncthreads();
    if (a.store) {
        const long t_off = (long)(*a.store_t) * a.n * L;
        for (int r = 0; r < nrows; ++r) {
            const int row = row0 + r, b = row / a.n, i = row - b * a.n;
            float* dst = a.store + (long)b * a.store_env_stride + t_off + (long)i * L;
            for (int e = tid; e < L; e += 256) dst[e] = tile[r * L + e];
        }
    }
    const bool active = tid < ITEMS;
    const int row_l = active ? tid / O : 0, y = active ? tid - row_l * O : 0;
    float in[3][3][V];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int x = 0; x < V; ++x) in[ch][dy][x] = tile[row_l * L + ch * V * V + (y + dy) * V + x];
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
    for (int oc = 0; oc < 6; ++oc) {
        // this wave's slices of the Linear weight for channel oc (prefetched: consumed after the barrier)
        f32x4 af[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = wave + 4 * j;
            if (c < PG)
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
                    af[j][ft] = *reinterpret_cast<const f32x4*>(a.lwp + ((size_t)(oc * 32 + 16 * ft + m) * PP + 16 * c + 4 * q));
        }
        float* cb_w = cbuf + (oc & 1) * 16 * CS;
        if (active) {
            float o[O];
            const float bias = a.cb[oc];
#pragma unroll
            for (int x = 0; x < O; ++x) o[x] = bias;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float w = a.cw[((oc * 3 + ch) * 3 + dy) * 3 + dx];
#pragma unroll
                        for (int x = 0; x < O; ++x) o[x] = fmaf(w, in[ch][dy][x + dx], o[x]);
                    }
#pragma unroll
            for (int x = 0; x < O; ++x) cb_w[row_l * CS + y * O + x] = leaky(o[x]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = wave + 4 * j;
            if (c < PG) {
                const f32x4 bf = *reinterpret_cast<const f32x4*>(cb_w + m * CS + 16 * c + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][ft][r], bf[r], acc[ft], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                                   // the observation tile is dead: reuse it for the reduction
    f32x4* red = reinterpret_cast<f32x4*>(tile);
    red[(wave * 2 + 0) * 64 + lane] = acc[0];
    red[(wave * 2 + 1) * 64 + lane] = acc[1];
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
            for (int r = 0; r < 4; ++r) a.out[orow * a.out_stride + 16 * ft + 4 * q + r] = leaky(s[r] + a.lb[16 * ft + 4 * q + r]);
        }
    }
}

int launch_policy_encode(const float* obs, int rows, int V, const float* cw, const float* cb, const float* lwp, const float* lb, float* out,
                         int out_stride, int n_agents, int agent_major, float* store, long store_env_stride, const int64_t* store_t,
                         hipStream_t s) {
    if (V != 15) return -2;
    constexpr int L = 3 * 15 * 15, CS = 176 + 4;
    const size_t lds = (size_t)(16 * L + 2 * 16 * CS) * sizeof(float);
    static bool attr_done_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (!attr_done_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode<15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        attr_done_dev[dev] = true;
    }
    EncK k{obs, rows, cw, cb, lwp, lb, out, out_stride, n_agents, agent_major, store, store_env_stride, store_t};
    hipLaunchKernelGGL(k_encode<15>, dim3((rows + 15) / 16), dim3(256), lds, s, k);
    return 0;
}

}  // namespace ssd
