// ssd_device.h -- device-side structures shared by the kernels (ssd_env.hip) and the C ABI (ssd_abi.hip).
// gfx950 only: one 64-lane wavefront owns one env; 4 waves (4 envs) per workgroup; no workgroup barriers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ssd_hip.h"

namespace ssd {

constexpr int kWave = 64;
#ifndef SSD_WAVES_PER_BLOCK
#define SSD_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = SSD_WAVES_PER_BLOCK;
constexpr int kBlock = kWave * kWavesPerBlock;

enum : int { C_EMPTY = 0, C_WALL = 1, C_APPLE = 2, C_WASTE = 3, C_RIVER = 4, C_STREAM = 5 };
enum : int { O_LEFT = 0, O_RIGHT = 1, O_UP = 2, O_DOWN = 3 };
enum : int { MODE_RESET = 0, MODE_STEP = 1, MODE_STEP_OBS = 2, MODE_OBS = 3 };

// Scalars of the world description.  Passed to the kernels BY VALUE (kernel-argument segment), so a wave has them after
// one scalar load instead of kernarg -> pointer -> fields.
struct DevHead {
    int32_t kind, H, W, HW, GS;          // GS = grid row stride in bytes (HW rounded up to 16)
    int32_t n, N, V, v, VV, VVp;         // VVp = V*V rounded up to 4
    int32_t Wp, PMS;                     // padded class map: row stride W + 2v, bytes (H + 2v) * Wp rounded up to 16
    int32_t vshift;                      // log2 of the power of two >= V: window rows are dealt to lanes 2^vshift at a time
    int32_t episode_limit, spawn_rotation, obs_color, rng_mode, n_actions;
    int32_t n_apple, n_waste;
    int32_t random_spawn, n_spawn, spawn_len;   // extra_args.random_spawn_point; 'P' cells; length of the reference's spawn list
    uint32_t env_id_base, seed_lo, seed_hi;
    uint32_t magic_W, magic_V, magic_VV, magic_3VV, magic_HW;  // floor(2^32/d)+1: q = umulhi(x, magic), exact for x < 2^16
};

// Static description of the world, one per handle, resident in HBM (tables and site lists; read through the caches).
struct DevSpec : DevHead {
    double thr_dep, thr_res, p_waste, p_apple;
    double harvest_p[4];
    // compute_probabilities (cleanup.py:189-204) tabulated on the host by the number of waste cells on the map: the
    // same fp64 expression evaluated once per possible count (-ffp-contract=off), so the kernel needs no fp64 division
    alignas(16) double tab_p[SSD_MAX_SITES + 1][2];   // {p_apple, p_waste}: one 16-byte read (TAPE mode compares recorded doubles)
    // COUNTER mode compares 24-bit integers (Rng::threshold): {threshold(p_apple), threshold(p_waste), flags, 0} with
    // flags bit 0 = p_apple > 0, bit 1 = not np.isclose(p_waste, 0) -- the kernel then needs no fp64 instruction at all
    alignas(16) uint32_t tab_thr[SSD_MAX_SITES + 1][4];
    uint32_t harvest_thr[4];
    float tab_den[SSD_MAX_CELLS + 1];    // apple_den = (float)(apples / (H * W)) in fp64 (map_env.py:291-292) by the apple count (0..H*W)
    uint16_t apple[SSD_MAX_SITES];       // cell index of each apple site, row-major scan order
    uint16_t waste[SSD_MAX_SITES];
    // the same lists as the step kernel's lanes hold them: record l = {apple[l], apple[64 + l], .., waste[l], waste[64 + l], ..}
    // (0 past the end of a list), so that a wave fetches its eight site cells per lane with ONE 16-byte request instead of eight
    alignas(16) uint16_t site_t[64][8];
    uint16_t spawn_cell[SSD_MAX_AGENTS]; // spawn cell of agent a under random_spawn_point = False
    uint16_t spawn_all[SSD_MAX_SPAWN];   // every spawn cell in row-major order (random_spawn_point = True)
    alignas(16) uint8_t reset_grid[SSD_MAX_CELLS];   // world after reset_map + custom_reset
    uint8_t lut[16 * 3];                 // full-colour LUT by class (0..5 cell codes, 5 + agent char)
};

// Mutable per-env state (device pointers).  arec packs one agent into 32 bits: row | col << 8 | orient << 16.
// One env's counters, 32 bytes: a wave reads them with ONE request (lanes 0 and 1, 16 bytes each) instead of four.
struct EnvHdr {
    uint32_t rng_base[4];  // Philox base words of the current episode (COUNTER mode), written by reset / import
    uint32_t epoch;        // resets so far
    int32_t ep_step;       // steps since the last reset
    uint32_t counts;       // waste cells << 16 | apple cells of the grid, 0xFFFFFFFF = unknown (recount)
    uint32_t pad;
};
struct DevState {
    uint8_t* grid;       // [N, GS]
    uint2* agents;       // [N, n] {arec, episode reward}
    EnvHdr* hdr;         // [N]
    int32_t* err;        // [1] sticky error bits
    unsigned long long* stamps;  // diagnostic builds only (-DSSD_STAMPS): [N, 32] s_memtime per phase; else null
};

struct DevTape {
    const uint8_t* move_order;
    const double* uniforms;
    int32_t ustride;
    const uint8_t* waste_order;
    const uint8_t* spawn_rot;
    const uint8_t* spawn_order;
};

struct DevStepOut {
    float *reward, *clean_num, *apple_den;
    uint8_t* terminated;
    float *collective, *equality;
    int32_t* n_draws;
};

struct DevObsOut {
    void* obs;
    int32_t fmt;
    float *state, *pos, *orient;
    long env_stride, slot_stride;  // placement of obs inside an episode storage (elements); 0, 0 = dense
    int32_t t_slots;               // slots of that storage (0 = unchecked): no observation is written at slot >= t_slots
    uint8_t* code;                 // optional side output: u8 cell classes [N, n, code_agent_stride]
    int32_t code_agent_stride;     // V * V rounded up to 16
    unsigned long long* stamps;  // diagnostic builds only
};

enum : int { ERR_BAD_ACTION = 1, ERR_BAD_TAPE = 2, ERR_KEYERROR = 4, ERR_TAPE_OVERRUN = 8, ERR_SLOT_OVERRUN = 16, ERR_F16_RANGE = 32 };
// The sticky numeric-status word of the current device (ssd_numeric_status): bit ERR_F16_RANGE is set by the pack kernels, the
// rollout heads and the learner's recurrence when a SCALED value of a two-term f16 split product leaves f16's range (65 504) -- the
// product would silently carry inf / NaN.  Allocated on first use (ssd_create and the pack entry points call it outside any capture).
int32_t* numeric_err_word();
constexpr float F16_MAX = 65504.f;

// LDS bytes one wave needs: grid | agent overlay | padded class map | output planes (+16 alignment slack) | colour lut.
// The class map + planes region doubles as scratch for the tape-mode waste ranks (2 * 256 bytes) during the step.
__host__ __device__ inline int lds_planes_bytes(const DevHead& s) { return ((s.n * 3 * s.VV + 16) + 15) & ~15; }
// (+ one agent's class-code window, V * V rounded up to 16, + 16 for the dump byte of idle lanes: the obs_code side output)
__host__ __device__ inline bool lds_code_all(const DevHead& s) { return s.n * SSD_CODE_AGENT_STRIDE(s.V) <= 2560; }   // all agents' windows fit
__host__ __device__ inline int lds_code_bytes(const DevHead& s) { return (lds_code_all(s) ? s.n : 1) * SSD_CODE_AGENT_STRIDE(s.V) + 16; }
#if defined(SSD_STAMPS) && SSD_STAMPS == 2
constexpr int kStampLds = 256;     // diagnostic build: the wave's stamps, at the end of its slice
#else
constexpr int kStampLds = 0;
#endif
__host__ __device__ inline int lds_per_wave(const DevHead& s) {
    int obs = s.PMS + lds_planes_bytes(s);
    if (obs < 512) obs = 512;
    return ((2 * s.GS + obs + 64 + lds_code_bytes(s) + 255) & ~255) + kStampLds;   // a multiple of the 256-byte LDS bank row
}

void launch_env(int mode, const DevSpec* spec, const DevSpec& host_spec, DevState st, const int32_t* actions,
                const uint8_t* env_mask, DevTape tape, DevStepOut so, DevObsOut oo, hipStream_t stream);

void launch_export(const DevSpec* spec, const DevSpec& hs, DevState st, ssd_state dst, hipStream_t stream);
void launch_import(const DevSpec* spec, const DevSpec& hs, DevState st, ssd_state src, hipStream_t stream);

void launch_build_inputs(int32_t batch, int32_t n, int32_t A, int32_t t0, const int64_t* last_actions,
                         const float* last_reward, const int64_t* last_actions_inc, const float* pos, float pos_scale,
                         float* out, int32_t out_stride, int32_t out_offset, hipStream_t stream);
struct BuildInputsLayout { int o_act, o_id, o_r, o_i, o_oth, o_dist, o_pos, width; };   // first column of each block, < 0: absent
BuildInputsLayout build_inputs_layout(int n, int A, uint32_t input_flags);
void launch_build_inputs_flags(int32_t batch, int32_t n, int32_t A, int32_t t0, uint32_t input_flags, const int64_t* last_actions,
                               const float* last_reward, const int64_t* last_actions_inc, const float* pos, float pos_scale,
                               float* out, int32_t out_stride, int32_t out_offset, hipStream_t stream);
void launch_incentive_transfer(int32_t B, int32_t T, int32_t n, const int64_t* a_inc, const float* rewards,
                               float effect_ratio, float cost_ratio, float incentive, float seq_len, float* give,
                               float* recv_pos, float* recv_neg, float* recv_zero, float* r_env, float* r_inc,
                               hipStream_t stream);

void launch_column_sums(const float* x, float* out, int G, int R, int C, float* workspace, hipStream_t stream);
void launch_copy_blocks(const ssd_block_copy* blocks, int count, hipStream_t stream);
void launch_clip_adam(const ssd_clip_adam_args* a, hipStream_t stream);
void launch_dueling_q(const float* a, const float* v, float* q, const float* dq, float* da, float* dv, int n, int T, int B, int inner, int K, hipStream_t stream,
                      int ld = 0, int merged = 0, float* gs = nullptr);
void launch_sample_ids(uint64_t seed, uint32_t call, int population, int count, int64_t* out, hipStream_t stream);
void launch_gather_rows(const ssd_row_gather* fields, int count, const int64_t* ids, int n_ids, hipStream_t stream);
void launch_td_sim_loss(const ssd_td_loss_args* a, int mode, hipStream_t stream);
void launch_encoder(const float* obs, int rows, int V, const float* cw, const float* cb, const float* lw, const float* lb, float* out,
                    int out_stride, int n_agents, int agent_major, float* store, long store_env_stride, const int64_t* store_t,
                    hipStream_t s);
void launch_conv_leaky(const float* obs, int rows, int V, const float* cw, const float* cb, float* out, int n_agents, int agent_major,
                       float* store, long store_env_stride, const int64_t* store_t, hipStream_t s);
void launch_store_step(const ssd_store_step* a, hipStream_t s);
void launch_gru_gates(const float* gi, const float* gh, float* h, int R, int H, hipStream_t s);
void launch_gru_fwd_train(const float* gi, const float* gh, const float* h, float* h_new, float* rzn, int R, int H, hipStream_t s);
void launch_gru_bwd(const float* dh, const float* rzn, const float* gh, const float* h, float* d_gi, float* d_gh, float* dh_prev, int R,
                    int H, hipStream_t s);
int launch_policy_encode(const ssd_policy_encode_args* p, hipStream_t s);
int launch_pack_encoder_lut(const float* cw, const float* cb, const float* lw, int V, int prec, void* table, void* lin_frags, hipStream_t s);
int launch_pack_encoder(const float* cw, const float* cb, const float* lw, int V, int prec, void* conv_frags, void* lin_frags, hipStream_t s);
void launch_pack_head(const ssd_policy_head_params* p, int prec, void* image, hipStream_t s);
void launch_gru_seq_fwd(const float* const* gi_parts, int n_parts, const float* const* wh_parts, const float* const* bh_parts, int n_wparts, float* hs,
                        float* rzn, float* ghn, int T, int G, int B, hipStream_t s);
void launch_gru_seq_bwd(const float* const* dhs_parts, const float* hs, const float* rzn, const float* ghn, const float* const* wh_parts, int n_wparts,
                        float* const* d_gi_parts, int n_parts, float* dgh, float* d_wh, float* d_bh_part, int T, int G, int Gn, int B, hipStream_t s);
int launch_policy_head(const ssd_policy_head* p, int inc, hipStream_t s);
void policy_head_plan(int n_env, int n_agents, int fused, int* wg_per_agent, int* waves_out, int* tiles_per_wave);
int launch_policy_inc_encode(const ssd_policy_head* ph, const ssd_policy_encode_args* pe, hipStream_t s);
int conv_wgrad_partial_rows(int R);
int launch_conv_wgrad(const uint8_t* codes, const float* d_conv, float* partial, int R, int V, hipStream_t s);
void launch_unroll_other(const int64_t* actions, const float* pos, const float* orient, const float* reward, const float* clean, const float* den,
                         float pos_scale, int B, int T, int n, int A, float* other, float* act_tm, hipStream_t stream);
float* bmm_scratch(hipStream_t stream);
int learner_precision();                  // 2: f32-equivalent (default), 1: single bf16 products (ssd_set_learner_precision)
void set_learner_precision(int p);
void launch_fill_blocks(const ssd_block_fill* blocks, int count, hipStream_t stream);
void launch_runner_stats(const float* coll, const float* eq, const float* ret, int n_env, int n_ret, double* acc, hipStream_t stream);
int launch_bias_bmm_fwd(const float* x, const float* w, const float* b, float* y, int n, int R, int I, int O, hipStream_t s, int leaky = 0);
int launch_bias_bmm_bwd(const float* g, const float* x, const float* w, float* dx, float* dw, float* db, const float* slope_of, int n, int R,
                        int I, int O, hipStream_t s, long x_set = 0, long g_set = 0, const float* act_y = nullptr, const float* x2 = nullptr, int I1 = 0,
                        int x1_div = 1, int x2_shared = 0, long w_set = 0);
int launch_bias_bmm2_fwd(const float* x1, const float* x2, const float* w, const float* b, float* y, int n, int R, int I1, int I2, int O, int x1_div,
                         int x2_shared, hipStream_t s);
#ifdef SSD_STAMPS
void set_policy_stamps(unsigned long long* buf);
#endif
void launch_dueling_pick(const float* av, int R, int A, const uint8_t* avail, const float* eps, const int64_t* step, uint32_t seed,
                         int n_agents, int B, int pairs, int64_t* actions, float* q_out, uint32_t env_id_base, hipStream_t s);

}  // namespace ssd
