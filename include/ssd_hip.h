/*
 * ssd_hip.h -- C ABI of the MI355X-native vectorised SSD (Cleanup / Harvest) grid world.
 *
 * This is the drop-in boundary for the hot path of drdh/Homophily-MARL.  The reference has no FFI at all:
 * its env is reached through Python duck-typing (src/envs/__init__.py:6-11, src/envs/multiagentenv.py:8-75).
 * Every entry point below therefore cites the reference *Python* interface it replaces; the host-side
 * mirror of that interface lives in homophily_marl_amd/envs/ and calls these symbols through ctypes.
 *
 * Conventions
 *   - return 0 on success, a negative ssd_status otherwise; the message is in ssd_last_error() (thread local).
 *   - no C++ exception crosses this boundary, nothing aborts.
 *   - the caller owns every in/out array (device pointers for ssd_*, host pointers for the oracle's
 *     ssd_cpu_* mirror); the library owns only the opaque handle.
 *   - all GPU entry points are asynchronous on the hipStream_t passed as `void* stream` (0 = null stream).
 *   - one handle = one device; calls on one handle are not re-entrant.
 *
 * Cell alphabet of the world grid (u8 codes): 0 ' ', 1 '@', 2 'A', 3 'H', 4 'R', 5 'S'
 * (src/envs/ssd/map_env.py:132,817-820; src/envs/ssd/cleanup.py:117-124).
 * Orientation codes: 0 LEFT, 1 RIGHT, 2 UP, 3 DOWN = list(ORIENTATIONS) order (map_env.py:28-31,791-793).
 * Positions are [row, col] (map_env.py:20-31).
 */
#ifndef SSD_HIP_H
#define SSD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSD_ABI_VERSION 7

#define SSD_MAX_AGENTS 10   /* maps hold at most 10 spawn points; agent ids >= 10 break the reference (map_env.py:370) */
#define SSD_MAX_CELLS 1024  /* H*W upper bound (largest reference map is 48x18 = 864) */
#define SSD_MAX_SPAWN 16    /* spawn points ('P' cells) under random_spawn_point (the reference maps hold 3 / 5 / 10) */
#define SSD_MAX_SITES 256   /* apple / waste site list upper bound (largest: 206 apple sites) */

typedef enum ssd_status {
    SSD_OK = 0,
    SSD_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
    SSD_ERR_DEVICE = -2,      /* HIP runtime error */
    SSD_ERR_NOMEM = -3,
    SSD_ERR_UNSUPPORTED = -4
} ssd_status;

enum { SSD_ENV_CLEANUP = 0, SSD_ENV_HARVEST = 1 };
enum { SSD_RNG_TAPE = 0, SSD_RNG_COUNTER = 1 };
enum { SSD_OBS_F32 = 0, SSD_OBS_BF16 = 1, SSD_OBS_U8 = 2, SSD_OBS_CODE = 3 };
enum { SSD_COLOR_SIMPLIFIED = 0, SSD_COLOR_FULL = 1 };

/* Constructor arguments.  Replaces CleanupEnv.__init__ / HarvestEnv.__init__ kwargs
 * (cleanup.py:29-61, harvest.py:18-29, config/envs/cleanup.yaml:3-15). */
typedef struct ssd_config {
    int32_t env_kind;            /* SSD_ENV_* */
    int32_t height, width;       /* map rows, cols */
    const char* ascii_map;       /* height*width chars, row-major, alphabet of constants.py:1-10 */
    int32_t n_agents;            /* <= SSD_MAX_AGENTS */
    int32_t n_env;               /* envs in this handle (this rank's shard) */
    int32_t view_size;           /* v; window edge V = 2v+1 */
    int32_t episode_limit;       /* terminated iff steps >= episode_limit (map_env.py:890-894) */
    int32_t random_spawn_point;  /* extra_args.random_spawn_point: 1 = every agent's spawn_point() shuffles the spawn list first */
    int32_t spawn_rotation;      /* extra_args.random_spawn_rotation: 0..3, or -1 = None = random */
    int32_t obs_color;           /* SSD_COLOR_* (extra_args.obs_color) */
    int32_t rng_mode;            /* SSD_RNG_* */
    int32_t device;              /* HIP device ordinal (ignored by the ssd_cpu_* mirror) */
    uint32_t env_id_base;        /* global id of env 0 of this shard (COUNTER streams are keyed by global id) */
    uint64_t seed;               /* COUNTER mode key */
    /* Cleanup thresholds (cleanup.py:31-54) */
    double threshold_depletion, threshold_restoration, waste_spawn_prob, apple_respawn_prob;
    /* Harvest SPAWN_PROB[min(k,3)] (harvest.py:20-22,118) */
    double harvest_spawn_prob[4];
} ssd_config;

/* One call's worth of recorded random draws (TAPE mode; appendix A.6 of SURVEY.md).  All arrays are
 * per env, row-major, and may be NULL in COUNTER mode.
 *   move_order  : result of np.random.shuffle over the MOVE/STAY list (map_env.py:540-542): the agent ids in
 *                 shuffled order, 0xFF padded.
 *   uniforms    : every np.random.rand(1)[0] of the call in consumption order (cleanup.py:172,183, harvest.py:119).
 *   waste_order : self.waste_points after random.shuffle (cleanup.py:178) as indices into the row-major waste
 *                 site list; only read when the call shuffles.
 *   spawn_rot   : np.random.randint(4) per agent at reset (map_env.py:789); only read when spawn_rotation = -1.
 *   spawn_order : self.spawn_points after the random.shuffle of agent a's spawn_point() call (map_env.py:776-777) as indices into
 *                 the row-major list of distinct spawn points (map_env.py:140-146); the agent takes the LAST free point of that
 *                 order.  spawn_list_len = n_spawn_points for Harvest and 2 * n_spawn_points for Cleanup, whose constructor appends
 *                 every point a second time (cleanup.py:79-80): each point id then appears twice. */
typedef struct ssd_tape {
    const uint8_t* move_order;   /* [n_env, n_agents] */
    const double* uniforms;      /* [n_env, uniforms_stride] */
    int32_t uniforms_stride;
    const uint8_t* waste_order;  /* [n_env, n_waste_sites] */
    const uint8_t* spawn_rot;    /* [n_env, n_agents] */
    const uint8_t* spawn_order;  /* [n_env, n_agents, spawn_list_len]; only read by resets with random_spawn_point = 1 */
} ssd_tape;

/* Outputs of one transition.  Replaces the return of MapEnv.step (map_env.py:874-915):
 * reward f64[n] -> f32, terminated, info{clean_num, apple_den, collective_return, equality_metric}. */
typedef struct ssd_step_out {
    float* reward;              /* [n_env, n] */
    float* clean_num;           /* [n_env, n] */
    float* apple_den;           /* [n_env, n] (same value for every agent, map_env.py:291-292) */
    uint8_t* terminated;        /* [n_env] */
    float* collective_return;   /* [n_env], written at termination (map_env.py:901-912), nullable */
    float* equality;            /* [n_env], written at termination, nullable */
    int32_t* n_draws;           /* [n_env] uniforms consumed by the call (parity check of the draw count), nullable */
} ssd_step_out;

/* Outputs of an observation pass.  Replaces get_obs / get_state / get_agent_pos / get_agent_orientation
 * (map_env.py:917-957).  Every pointer is nullable: only non-NULL outputs are produced.
 *   obs   : [n_env, n, 3, V, V] as f32 / bf16 (values k/256) or u8 (values k), or with SSD_OBS_CODE
 *           [n_env, n, V, V] u8 cell classes (0 nothing, 1 apple, 2 waste, 3 wall-or-agent; simplified colours only).
 *   state : [n_env, 3, H, W] f32. */
typedef struct ssd_obs_out {
    void* obs;
    int32_t obs_format;         /* SSD_OBS_* */
    float* state;
    float* pos;                 /* [n_env, n, 2] row, col as float */
    float* orient;              /* [n_env, n, 2] ORIENTATIONS vector as float */
    /* Optional placement of `obs` inside an episode storage [n_env, t_slots, n, ...] (EpisodeBatch.update of the "obs" key,
     * episode_runner.py:59-67, without the copy): env b's block is written at element offset
     * b * obs_env_stride + ep_step * obs_slot_stride, ep_step being the env's step counter AFTER the call (0 after a reset).
     * Both 0 (default): dense [n_env, n, ...].  GPU library only. */
    int64_t obs_env_stride, obs_slot_stride;
    /* obs_t_slots (0 = unchecked): number of time slots of that storage.  A call whose slot would be >= obs_t_slots (stepping an
     * env past its episode storage) never touches another env's block: its observation lands in the env's own LAST slot and the
     * sticky bit 16 of ssd_poll_error is raised. */
    int32_t obs_t_slots;
    /* obs_code (nullable; simplified colours; together with a f32 / bf16 / u8 `obs`): additionally emit the window as one
     * CHANNEL-MASK byte per cell (bit 0 = R = waste, bit 1 = G = apple, bit 2 = B = wall or agent; 0 = nothing: the planes of the
     * simplified palette are one-hot, so a cell is 0, 1, 2 or 4 -- SSD_CODE_CHANNEL_MASK) into a dense side buffer
     * u8 [n_env, n, obs_code_agent_stride], obs_code_agent_stride = V * V rounded up to 16, 16-byte aligned.  This is what the
     * rollout-time encoder (ssd_policy_encode) consumes: 225 B instead of 2 700 B per agent-step at V = 15. */
    uint8_t* obs_code;
} ssd_obs_out;
#define SSD_CODE_AGENT_STRIDE(V) ((((V) * (V)) + 15) & ~15)
/* byte alphabets of per-cell windows: the SSD_OBS_CODE observation format (0 nothing, 1 apple, 2 waste, 3 wall-or-agent) and the
 * channel masks of ssd_obs_out.obs_code */
enum { SSD_CODE_CLASS = 0, SSD_CODE_CHANNEL_MASK = 1 };

/* Raw env state for parity tests and KATs (Agent.set_pos etc. in the reference). */
typedef struct ssd_state {
    uint8_t* grid;              /* [n_env, H*W] cell codes */
    int16_t* pos;               /* [n_env, n, 2] */
    uint8_t* orient;            /* [n_env, n] */
    int32_t* ep_reward;         /* [n_env, n] cumulative episode reward (self.rewards, map_env.py:885-888) */
    int32_t* ep_step;           /* [n_env] self._episode_steps */
    uint32_t* epoch;            /* [n_env] COUNTER-mode episode counter (resets so far) */
} ssd_state;

typedef struct ssd_env ssd_env; /* opaque */

int ssd_abi_version(void);
const char* ssd_last_error(void);

/* REGISTRY[env](**env_args)  (envs/__init__.py:6-11, runners/episode_runner.py:15) */
int ssd_create(const ssd_config* cfg, ssd_env** out);
int ssd_destroy(ssd_env* env);

/* MapEnv.reset (map_env.py:986-993 -> _reset :297-326).  env_mask: u8[n_env] device pointer, nullable = all. */
int ssd_reset(ssd_env* env, const uint8_t* env_mask, const ssd_tape* tape, ssd_step_out* out, void* stream);
/* MapEnv.step (map_env.py:874-915).  actions: int32[n_env, n] device pointer. */
int ssd_step(ssd_env* env, const int32_t* actions, const ssd_tape* tape, ssd_step_out* out, void* stream);
/* get_obs / get_state / get_agent_pos / get_agent_orientation (map_env.py:917-957) */
int ssd_observe(ssd_env* env, ssd_obs_out* out, void* stream);
/* step immediately followed by observe of the new state, in one launch (the rollout inner loop,
 * runners/episode_runner.py:57-97: env.step then next iteration's get_obs). */
int ssd_step_observe(ssd_env* env, const int32_t* actions, const ssd_tape* tape, ssd_step_out* out,
                     ssd_obs_out* obs, void* stream);

/* Sticky device-side error bits (1 action out of range = KeyError in action_map, agent.py:174-176,235-237;
 * 2 malformed tape; 4 agent_by_pos KeyError; 8 tape overrun; 16 observation slot >= obs_t_slots).  Synchronises the device;
 * clears the bits. */
int ssd_poll_error(ssd_env* env, int32_t* bits);

int ssd_export_state(ssd_env* env, ssd_state* dst, void* stream);
int ssd_import_state(ssd_env* env, const ssd_state* src, void* stream);

/* Static facts derived from the map (get_env_info, map_env.py:1008-1019 and the site lists of
 * cleanup.py:72-90 / harvest.py:31-35). */
typedef struct ssd_info {
    int32_t n_actions;          /* 9 Cleanup, 8 Harvest (agent.py:153-154,207-209) */
    int32_t n_apple_sites, n_waste_sites, n_spawn_points;
    int32_t max_uniforms;       /* upper bound of uniforms consumed by one call = apple sites + waste sites */
    int32_t obs_edge;           /* V */
} ssd_info;
int ssd_get_info(const ssd_env* env, ssd_info* out);

/* HomophilyMAC._build_inputs tail (controllers/homophily_controller.py:137-184): everything except the conv
 * encoder.  Writes [B*n, A + n + 1 + 1 + 2] = onehot(last action) | onehot(id) | sign(last reward) |
 * sign(#recv+ - #recv-) | pos/||(H,W)||.  t0 is a flag word: bit 0 selects the t == 0 branch (zeros for the three history
 * terms; a last action of -1 has the same effect per row), bit 1 writes agent-major rows (i * batch + b) instead of (b * n + i);
 * bits 8.. = T > 0: `batch` counts [episodes, T] rows and the three history tensors hold every step's OWN action / reward / incentives
 * -- the kernel reads the row of the previous timestep and takes the t == 0 branch at the first step of each episode (the learner's
 * time-batched assembly without shifted copies of the tensors). */
int ssd_build_inputs(int32_t batch, int32_t n_agents, int32_t n_actions, int32_t t0,
                     const int64_t* last_actions /*[B,n]*/, const float* last_reward /*[B,n]*/,
                     const int64_t* last_actions_inc /*[B,n,n]*/, const float* pos /*[B,n,2]*/,
                     float pos_scale, float* out, int32_t out_stride, int32_t out_offset, void* stream);

/* Incentive reward transfer (learners/homophily_learner.py:94-115).  actions_inc int64[B,T,n,n] (giver dim 2,
 * receiver dim 3), rewards f32[B,T-1,n].  Outputs f32: give[B,T-1,n], recv_pos/neg/zero[B,T,n],
 * rewards_for_env/inc[B,T-1,n] = (r +/- ...) / seq_len with seq_len = batch.max_seq_length (true division). */
/* The time-batched incentive head's per-receiver features (homophily_agent.py:194-201) of a sampled batch, one launch:
 * actions i64 [B, T, n], pos / orient f32 [B, T, n, 2], reward / clean_num / apple_den f32 [B, T, n] (contiguous) ->
 * other f32 [T * B, n, n_actions + 7] = [one-hot(a_j), pos_j / pos_scale, orient_j, r_j, clean_j, apple_den_j] at row t * B + b, and
 * act_tm f32 [n, T * B, n_actions] = the one-hot alone, agent-major (the extra input columns of fc1_inc). */
int ssd_unroll_other(const int64_t* actions, const float* pos, const float* orient, const float* reward, const float* clean_num, const float* apple_den,
                     float pos_scale, int32_t batch, int32_t T, int32_t n_agents, int32_t n_actions, float* other, float* act_tm, void* stream);
int ssd_incentive_transfer(int32_t batch, int32_t T, int32_t n_agents, const int64_t* actions_inc,
                           const float* rewards, float effect_ratio, float cost_ratio, float incentive,
                           float seq_len, float* give, float* recv_pos, float* recv_neg, float* recv_zero,
                           float* rewards_for_env, float* rewards_for_inc, void* stream);

/* ssd_column_sums: out f32 [groups, cols] = sum over rows of x f32 [groups, rows, cols] (contiguous): the bias gradients of the
 * learner's affine layers (the row sums of dL/dY).  Deterministic; with workspace f32 [groups, ceil(rows / SSD_COLSUM_CHUNK), cols]
 * (nullable) two launches: chunk sums, then their sum -- no cross-workgroup hand-off inside a launch. */
#define SSD_COLSUM_CHUNK 64
int ssd_column_sums(const float* x, float* out, int32_t groups, int32_t rows, int32_t cols, float* workspace, void* stream);

/* ssd_copy_blocks: `count` (1..SSD_COPY_BLOCKS_MAX) strided 2-D f32 block copies as ONE launch,
 * dst[r * dst_stride + c] = src[r * src_stride + c] for r < rows, c < cols (strides in floats; blocks must not overlap).  `blocks` is a
 * HOST array read during the call.  The learner packs the r / z / n blocks of the GRU parameters of both heads side by side with it
 * (homophily_agent.py:83-112 keeps them as 24 separate tensors) and splits the gradients again: 2 launches instead of 8 concatenations
 * and 24 strided copies per train step. */
/* ssd_dueling_q: the dueling combination of the learner's time-batched heads (homophily_agent.py:168-170, 203-207: q = v + a - mean_k a)
 * written straight in the batch layout the loss reads -- a f32 [n, rows, K], v f32 [n, rows, 1] with rows = T * B * inner in time-major
 * order r = (t * B + b) * inner + j (what ssd_bias_bmm_fwd leaves) -> q f32 [B, T, n, inner, K] (inner = 1: q_env; inner = n: q_inc) --
 * and its backward dq -> (da, dv) in the layers' layout: one launch each instead of mean / sub / add / permuted copy and their autograd. */
int ssd_dueling_q_fwd(const float* a, const float* v, float* q, int32_t n, int32_t T, int32_t B, int32_t inner, int32_t K, void* stream);
int ssd_dueling_q_bwd(const float* dq, float* da, float* dv, int32_t n, int32_t T, int32_t B, int32_t inner, int32_t K, void* stream);

/* ssd_gather_rows: dst_f[e] = src_f[ids[e]] for e < n_ids and every field f < count (1..SSD_COPY_BLOCKS_MAX) as ONE launch; a field is
 * a dense array of rows of row_bytes bytes (any alignment).  ReplayBuffer.sample (episode_buffer.py:240-244) draws 16 episodes out of
 * thirteen storage fields with it: one launch instead of one indexing kernel per field.  `fields` is a HOST array read during the call,
 * ids a DEVICE array of n_ids int64 (each within the fields' rows: the caller checks). */
typedef struct ssd_row_gather {
    const void* src;
    void* dst;
    int64_t row_bytes;
} ssd_row_gather;
int ssd_gather_rows(const ssd_row_gather* fields, int32_t count, const int64_t* ids, int32_t n_ids, void* stream);

/* ssd_sample_ids: ids[0 .. count) = `count` DISTINCT episode indices drawn uniformly from [0, population) on the device --
 * ReplayBuffer.sample's `np.random.choice(episodes_in_buffer, batch_size, replace=False)` (episode_buffer.py:240-244) without a host
 * draw or an H2D copy of the indices.  Counter generator keyed by (seed, call): the caller advances `call` per sample; the same
 * (seed, call, population, count) gives the same ids on every device (a restatement for host tensors: ops.sample_ids).  Floyd's
 * subset sampling + a Fisher-Yates pass over the picks; a bounded draw is floor(u32 * m / 2^32) (bias < m / 2^32).  ids: DEVICE int64. */
#define SSD_SAMPLE_IDS_MAX 1024
int ssd_sample_ids(uint64_t seed, uint32_t call, int32_t population, int32_t count, int64_t* ids, void* stream);

#define SSD_COPY_BLOCKS_MAX 32
typedef struct ssd_block_copy {
    const float* src;
    float* dst;
    int32_t rows, cols, src_stride, dst_stride;
} ssd_block_copy;
int ssd_copy_blocks(const ssd_block_copy* blocks, int32_t count, void* stream);

/* ssd_fill_blocks: `count` (1..SSD_FILL_BLOCKS_MAX) fills with a 32-bit pattern as ONE launch (dst 4-byte aligned, bytes a multiple of
 * 4; `blocks` is a HOST array read during the call).  The vectorised runner opens an episode with it: previous actions -1 (i64: the
 * pattern 0xFFFFFFFF), previous reward / incentive actions / episode returns / hidden states / time index 0 (episode_runner.py:48-60,
 * homophily_controller.py:95-99 init_hidden) -- seven fill launches before. */
#define SSD_FILL_BLOCKS_MAX 16
typedef struct ssd_block_fill {
    void* dst;
    int64_t bytes;
    uint32_t value;
    uint32_t reserved;
} ssd_block_fill;
int ssd_fill_blocks(const ssd_block_fill* blocks, int32_t count, void* stream);

/* ssd_runner_stats: one rollout's share of EpisodeRunner's statistics (episode_runner.py:121-152) added to a device accumulator,
 * acc f64 [4] += { sum collective_return [n_env], sum equality [n_env], sum episode_return [n_returns], sum episode_return^2 }:
 * one launch, f64, fixed summation order; the host reads acc at its log interval. */
int ssd_runner_stats(const float* collective_return, const float* equality, const float* episode_return, int32_t n_env, int32_t n_returns,
                     double* acc, void* stream);

/* ssd_clip_adam_step: the optimiser tail of HomophilyLearner.cal_loss_and_step (homophily_learner.py:223-226) --
 *   clip_grad_norm_(params_inc, clip); clip_grad_norm_(params_env, clip); optimiser_inc.step(); optimiser_env.step()
 * with the conv encoder a member of BOTH parameter groups (homophily_agent.py:127-146: its gradient is scaled by both clips, the second
 * norm sees the first scaling, and both Adams move it, each with its own moments) -- as TWO launches over the flat gradient buffer
 * instead of ~20 (2 x multi-tensor norm + scalar arithmetic + multi-tensor scale, 2 x fused Adam + step counters):
 *   1. per 1024-element chunk, the sums of squares of the three segments (encoder / env head / inc head, per job); every step counter += 1
 *   2. every workgroup adds the chunk sums in a fixed order (deterministic; no atomics, no cross-workgroup hand-off), forms
 *      c_inc = min(1, clip / (sqrt(S_enc + S_inc) + 1e-6)), c_env = min(1, clip / (sqrt(c_inc^2 S_enc + S_env) + 1e-6)), writes the
 *      scaled gradients back and applies torch.optim.Adam's update (no weight decay / amsgrad; lerp form of the first moment,
 *      bias corrections from the per-parameter step counters) -- inc optimiser first, then env, as the reference steps them.
 * jobs: DEVICE array, one per parameter tensor in flat-buffer order; state index 0 = the inc optimiser's (exp_avg, exp_avg_sq, step),
 * 1 = the env optimiser's; a parameter outside an optimiser has NULLs there.  partials: f32 [ceil(total / 1024)][3] workspace. */
#define SSD_ADAM_MAX_JOBS 64
typedef struct ssd_adam_job {
    float* param;
    int64_t offset;                /* first element in the flat gradient buffer */
    int32_t numel, segment;        /* segment: 0 encoder (both optimisers), 1 env head, 2 inc head */
    float* exp_avg[2];
    float* exp_avg_sq[2];
    float* step[2];                /* f32 device scalars (torch's capturable / fused Adam keeps them so) */
} ssd_adam_job;
typedef struct ssd_clip_adam_args {
    float* flat_grad;
    int64_t total;                 /* elements of flat_grad = sum of the jobs' numel (the jobs tile it in order) */
    const ssd_adam_job* jobs;
    int32_t n_jobs;
    float* partials;
    float lr_inc, lr_env, beta1, beta2, eps, clip;
} ssd_clip_adam_args;
int ssd_clip_adam_step(const ssd_clip_adam_args* args, void* stream);

/* ssd_td_sim_loss: the loss of HomophilyLearner.cal_loss_and_step (learners/homophily_learner.py:94-217) -- incentive reward
 * transfer, double-Q TD losses of the env head and the incentive head, the similarity loss with the exact-value clustering rule --
 * AND its gradient w.r.t. the live Q-values in one launch (the reference builds it from ~100 tensor ops that autograd differentiates).
 * All tensors are over the batch's T+1 time slots (t_slots = batch.max_seq_length), contiguous:
 *   q_env, tq_env f32 [B, T+1, n, A] (live / target net, masked by avail as the reference does); q_inc, tq_inc f32 [B, T+1, n, n, 3];
 *   actions i64 [B, T+1, n]; actions_inc i64 [B, T+1, n(giver), n(receiver)]; avail i32 [B, T+1, n, A]; reward, clean_num f32 [B, T+1, n];
 *   terminated u8 [B, T+1]; filled i64 [B, T+1].
 * mode 0: only the per-row denominators: partials[:, 0] = mask, partials[:, 1] = sum of sim_loss_mask (their sums are
 *         mask.sum() and sim_loss_mask.sum(); data parallel: all-reduce them before mode 1).
 * mode 1: dens f32 [2] = the GLOBAL (mask.sum(), sim_loss_mask.sum()); writes dq_env / dq_inc (shapes of q_env / q_inc, every element)
 *         = d loss / d q with loss = (sum (td_env mask)^2 + sum (td_inc mask)^2) / dens[0] + sim_loss_weight sum_sim / (1 + dens[1]),
 *         and partials f32 [B * T * n, SSD_TD_LOSS_PARTIALS] per (b, t, i): 0 mask, 1 sim mask sum, 2 (td_env mask)^2,
 *         3 (td_inc mask)^2, 4 similarity terms, 5 q_env taken, 6 sum_j q_inc taken, 7 give, 8 recv+ - recv-, 9 clean (recv+ - recv-),
 *         10 clean flag, 11 r (recv+ - recv-), 12 r  -- the caller adds the rows in a fixed order (losses and the learner's log values).
 * consider_others_inc (config/algs/homophily.yaml, default False) is not covered: the caller keeps its tensor-op path for it. */
#define SSD_TD_LOSS_PARTIALS 16
typedef struct ssd_td_loss_args {
    int32_t batch, t_slots, n_agents, n_actions, sim_horizon, double_q;
    float gamma_env, gamma_inc, reward_scale, incentive_ratio, incentive_cost, incentive, seq_len, sim_threshold, sim_loss_weight;
    const float *q_env, *q_inc, *tq_env, *tq_inc;
    const int64_t *actions, *actions_inc;
    const int32_t* avail;
    const float *reward, *clean_num;
    const uint8_t* terminated;
    const int64_t* filled;
    const float* dens;
    float *dq_env, *dq_inc, *partials;
} ssd_td_loss_args;
int ssd_td_sim_loss(const ssd_td_loss_args* args, int32_t mode, void* stream);

/* ---- rollout-time (inference) controller pieces that are not GEMMs (csrc/ssd_policy.hip) -----------------------------
 * ssd_encoder: HomophilyAgent.rgb_preprocess (homophily_agent.py:20-27,213-214) = Conv2d(3, conv_out, 3, 1) + LeakyReLU +
 *   Flatten + Linear(conv_out * (V-2)^2, feat_out) + LeakyReLU on obs f32 [rows, 3, V, V] -> out[row * out_stride + 0..feat_out).
 *   agent_major: rows are (env b, agent i); write row (i * (rows / n_agents) + b) instead (per-agent batched GEMM layout).
 * ssd_gru_gates: the gate arithmetic of the hand-written GRU cell (homophily_agent.py:162-165,188-191) on gi = x W_i + b_i,
 *   gh = h W_h + b_h, both [rows, 3 * hidden] in (r, z, n) order; h [rows, hidden] is updated in place.
 * ssd_dueling_pick: q = v + a - mean(a) (homophily_agent.py:168-170,204-206) on av [rows, n_actions + 1] (advantages, then the
 *   value) followed by the epsilon-greedy choice of EpsilonGreedyActionSelector (action_selectors.py:44-68) with the available
 *   mask avail u8[n_actions] (NULL = all).  epsilon f32 and step i64 are device scalars; rows are agent-major (i, b) or, with
 *   pairs = 1, (i, b, j) with the diagonal i == j forced to 0 (homophily_controller.py:44-46); actions are written env-major
 *   [batch, n] / [batch, n, n].  q_out (nullable) receives q [rows, n_actions].  The exploration draw of a row is keyed by
 *   (seed, *step, (env_id_base + b) * n + i [, * n + j]): the GLOBAL env id, so env shards draw what the unsharded job draws. */
int ssd_encoder(const float* obs, int32_t rows, int32_t view_edge, int32_t conv_out, int32_t feat_out, const float* conv_w,
                const float* conv_b, const float* lin_w, const float* lin_b, float* out, int32_t out_stride, int32_t n_agents,
                int32_t agent_major, float* store_obs, int64_t store_env_stride, const int64_t* store_t, void* stream);
/* store_obs (nullable): additionally copy the observation of row (env b, agent i) to
 * store_obs[b * store_env_stride + (*store_t) * n_agents * 3VV + i * 3VV ...], i.e. into obs[b, t] of an episode storage
 * f32 [n_env, T+1, n, 3, V, V] (EpisodeBatch.update of the "obs" key, episode_runner.py:59-67) with store_env_stride =
 * (T+1) * n * 3VV and the time index read from device memory. */

/* ssd_conv_leaky: only the Conv2d(3, conv_out, 3, 1) + LeakyReLU + Flatten part of rgb_preprocess: out f32 [rows, conv_out*(V-2)^2]
 * (row order as agent_major says); the Linear layer behind it is a plain GEMM.  store_* as in ssd_encoder. */
int ssd_conv_leaky(const float* obs, int32_t rows, int32_t view_edge, int32_t conv_out, const float* conv_w, const float* conv_b,
                   float* out, int32_t n_agents, int32_t agent_major, float* store_obs, int64_t store_env_stride,
                   const int64_t* store_t, void* stream);

/* One launch for the small per-timestep fields of an episode storage [n_env, t_slots, ...] (the EpisodeBatch.update calls of
 * episode_runner.py:59-93): every non-NULL source is written to dst[(b * t_slots + *t_index), ...]; actions also as one-hot. */
typedef struct ssd_store_step {
    const int64_t* t_index;     /* device scalar */
    int32_t n_env, n_agents, n_actions, t_slots;
    const float *pos, *orient;                       /* [n_env, n, 2] */
    const float *reward, *clean_num, *apple_den;     /* [n_env, n] */
    const uint8_t* terminated;                       /* [n_env] */
    const int64_t *actions, *actions_inc;            /* [n_env, n], [n_env, n, n] */
    float *dst_pos, *dst_orient, *dst_reward, *dst_clean_num, *dst_apple_den, *dst_actions_onehot;
    uint8_t* dst_terminated;
    int64_t *dst_actions, *dst_actions_inc;
    /* optional runner state carried to the next timestep by the same launch (all nullable): the "previous" action / reward /
     * incentive inputs of the controller, the running episode return (+= reward), and two device counters that this launch
     * does not read itself (so one thread can update them without a grid-wide barrier): *next_t_out = *t_index + 1 and
     * *counter_inc += 1.  next_t_out must not alias t_index (ssd_policy_encode's slot_t_copy provides the second scalar). */
    int64_t* prev_actions;          /* [n_env, n]    <- actions */
    float* prev_reward;             /* [n_env, n]    <- reward */
    int64_t* prev_actions_inc;      /* [n_env, n, n] <- actions_inc */
    float* ep_return;               /* [n_env, n]    += reward */
    int64_t* next_t_out;
    int64_t* counter_inc;
} ssd_store_step;
int ssd_store_step_launch(const ssd_store_step* args, void* stream);
int ssd_gru_gates(const float* gi, const float* gh, float* h, int32_t rows, int32_t hidden, void* stream);
/* Training forms of the GRU gate arithmetic: forward also stores (r, z, n) [rows, 3 * hidden]; backward turns dL/dh_new into
 * dL/dgi, dL/dgh [rows, 3 * hidden] and the direct part of dL/dh_prev [rows, hidden]. */
int ssd_gru_gates_fwd(const float* gi, const float* gh, const float* h, float* h_new, float* rzn, int32_t rows, int32_t hidden, void* stream);
int ssd_gru_gates_bwd(const float* dh_new, const float* rzn, const float* gh, const float* h, float* d_gi, float* d_gh, float* dh_prev,
                      int32_t rows, int32_t hidden, void* stream);
int ssd_dueling_pick(const float* av, int32_t rows, int32_t n_actions, const uint8_t* avail, const float* epsilon, const int64_t* step,
                     uint32_t seed, int32_t n_agents, int32_t batch, int32_t pairs, int64_t* actions, float* q_out, uint32_t env_id_base,
                     void* stream);

/* ---- the learner's recurrence over all T timesteps in one launch per direction (csrc/ssd_gru_seq.hip) -----------------------
 * The GRU cell of HomophilyAgent (homophily_agent.py:162-165,188-191) unrolled from a zero state as the learner does
 * (homophily_learner.py:68-91), hidden = 64.  gi f32 [T, G, B, 192] = x_t W_i + b_i in (r, z, n) order for G independent weight
 * sets (agents x heads) and B sequences (B a multiple of 16: pad); wh [G, 64, 192], bh [G, 192]; hs f32 [G, T, B, 64] receives h_1..h_T.
 * Training: rzn [T, G, B, 192] and ghn [T, G, B, 64] (both or neither) receive the gates and the hidden-side n pre-activation.
 * Backward: dhs = dL/dhs -> d_gi [T, G, B, 192], d_wh [G, 64, 192], d_bh_part [G, ceil(B/16), 192] (partial sums per 16-row tile;
 * the caller adds them over the tile axis); dgh [G, T, B, 192] is workspace (dL/d(h W_h + b_h) per step, the operand of d_wh). */
int ssd_gru_seq_fwd(const float* gi, const float* wh, const float* bh, float* hs, float* rzn, float* ghn, int32_t T, int32_t G, int32_t B,
                    void* stream);
int ssd_gru_seq_bwd(const float* dhs, const float* hs, const float* rzn, const float* ghn, const float* wh, float* d_gi, float* dgh,
                    float* d_wh, float* d_bh_part, int32_t T, int32_t G, int32_t B, void* stream);
/* The same two launches with the input-side projections given as n_parts (1..4) separately allocated SET-MAJOR tensors
 * gi_parts[k] f32 [G / n_parts, T, B, 192] (what ssd_bias_bmm_fwd leaves for a group of weight sets; the learner passes live / target net
 * x env / inc head as they are -- no concatenation, no transpose to time-major), and their gradients written into d_gi_parts[k] of the
 * same shape (every part must be valid memory; a part that needs no gradient is scratch).  hs, rzn, ghn, dgh, d_wh, d_bh_part as above. */
int ssd_gru_seq_fwd_parts(const float* const* gi_parts, int32_t n_parts, const float* const* wh_parts, const float* const* bh_parts, int32_t n_wparts,
                          float* hs, float* rzn, float* ghn, int32_t T, int32_t G, int32_t B, void* stream);
int ssd_gru_seq_bwd_parts(const float* const* dhs_parts, const float* hs, const float* rzn, const float* ghn, const float* const* wh_parts,
                          int32_t n_wparts, float* const* d_gi_parts, int32_t n_parts, float* dgh, float* d_wh, float* d_bh_part, int32_t T, int32_t G,
                          int32_t G_grad, int32_t B, void* stream);
/* ABI 7: the recurrence weights of the _parts launches come as n_wparts (1..4) separately allocated tensors wh_parts[k] f32 [G / n_wparts, 64, 192]
 * and bh_parts[k] f32 [G / n_wparts, 192] over the sets in order (the learner: live net [2n], target net [2n] -- no concatenation per step);
 * d_wh and d_bh_part stay single outputs.  The backward walks only the first G_grad sets (a whole number of gi parts: the sets whose
 * states carry a gradient -- the live net's, the target net's follow): dhs_parts[k] f32 [G / n_parts, T, B, 64] are the gradients of the
 * states per gi part (parts past G_grad are not read), d_gi_parts past G_grad are not written, dgh [G_grad, T, B, 192],
 * d_wh [G_grad, 64, 192], d_bh_part [G_grad, tiles, 192]; hs / rzn / ghn keep the forward's shapes over all G sets. */

/* ---- the learner's per-agent affine layers (csrc/ssd_bmm.hip) ---------------------------------------------------------------
 * th.baddbmm(b, x, w) over the agent axis (homophily_agent.py:154-208: fc1, GRU input projections, dueling heads) and its backward,
 * f32 (exact-f32 MFMAs), contiguous tensors: x [n, rows, in], w [n, in, out], b [n, out], y / g [n, rows, out].
 *   fwd: y = b + x w
 *   bwd: dx = g w^T, dw = x^T g, db = column sums of g; each output nullable (dx needs w, dw / db need x); one launch --
 *        two when a long row axis (>= 4096 rows) meets few dw tiles: the rows are then cut into chunks whose partial tiles a
 *        second small launch adds in order through a 1 MiB scratch that belongs to (device, stream): launches on different streams
 *        never share one --, deterministic.  The scratch of a stream is allocated at its first launch; a stream that is going to CAPTURE
 *        such launches reserves its scratch before the capture starts (ssd_bmm_reserve_scratch; inside a capture nothing can be
 *        allocated and the launch stays unchunked).  Operand sets of 2^30 elements or more: SSD_ERR_UNSUPPORTED.  slope_of (nullable, [n, rows, in]): dx is multiplied elementwise by LeakyReLU'(.) taken from
 *        the sign of slope_of (the backward through a LeakyReLU whose OUTPUT is slope_of). */
int ssd_bias_bmm_fwd(const float* x, const float* w, const float* b, float* y, int32_t n, int32_t rows, int32_t in, int32_t out, void* stream);
/* The same layer followed by nn.LeakyReLU() (slope 0.01), as HomophilyAgent applies it to fc1 (homophily_agent.py:158,182):
 *   leaky_fwd: y = LeakyReLU(b + x w);   leaky_bwd: g is the gradient w.r.t. y, y the forward's output -- g is multiplied by
 *   LeakyReLU'(.) (taken from the sign of y) where it is loaded, then dx, dw, db as ssd_bias_bmm_bwd.  Two elementwise launches per
 *   layer and direction less. */
int ssd_bias_bmm_leaky_fwd(const float* x, const float* w, const float* b, float* y, int32_t n, int32_t rows, int32_t in, int32_t out, void* stream);
int ssd_bias_bmm_leaky_bwd(const float* g, const float* y, const float* x, const float* w, float* dx, float* dw, float* db, const float* slope_of,
                           int32_t n, int32_t rows, int32_t in, int32_t out, void* stream);
int ssd_bias_bmm_bwd(const float* g, const float* x, const float* w, float* dx, float* dw, float* db, const float* slope_of, int32_t n,
                     int32_t rows, int32_t in, int32_t out, void* stream);
/* ---- ABI 7: the dueling heads of the learner as ONE layer per head + one dueling launch ---------------------------------------------
 * The reference evaluates the advantage and the value layer of a head separately (homophily_agent.py:166-170,202-207) and, for the
 * incentive head, on materialised rows [h_i | other_j] of every ordered pair (:194-201).  Here the two layers are the columns 0..K-1 and K
 * of one weight [n, in, K + 1] (the caller concatenates the parameters once per step), and the pair rows are never built:
 *   ssd_bias_bmm2_fwd    y[g][r] = b[g] + [x1[g][r / x1_div] | x2[(g)][r]] w[g]: TWO-SOURCE rows -- columns [0, in1) from x1 f32
 *                        [n, rows / x1_div, in1] (h_i, the same for the x1_div receivers j), columns [in1, in1 + in2) from x2 f32
 *                        [rows, in2] shared by all weight sets (x2_shared) or [n, rows, in2]; in1 a multiple of 16.
 *   ssd_bias_bmm2_bwd_w  dw [n, in1 + in2, out] = rows^T g and db = column sums of g for the same two-source rows (either nullable).
 *   ssd_bias_bmm_bwd_x   dx [n, rows, in] = g w^T where w[g] are the LEADING `in` rows of a wider layer (w_set floats between two weight sets):
 *                        the gradient of x1 from the row-group sums of g (ssd_dueling_head_bwd's gs).
 *   ssd_dueling_head_fwd q [B, T, n, inner, K] = v + a - mean_k a from y f32 [n, T * B * inner, K + 1] (a = columns 0..K-1, v = column K).
 *   ssd_dueling_head_bwd dq -> dy (same shape as y: da = dq - mean_k dq, dv = sum_k dq) and, when gs is given, gs f32 [n, T * B, K + 1] =
 *                        the sum of dy over the `inner` rows of every (agent, t, b) in row order (deterministic). */
int ssd_bias_bmm2_fwd(const float* x1, const float* x2, const float* w, const float* b, float* y, int32_t n, int32_t rows, int32_t in1, int32_t in2,
                      int32_t out, int32_t x1_div, int32_t x2_shared, void* stream);
int ssd_bias_bmm2_bwd_w(const float* g, const float* x1, const float* x2, float* dw, float* db, int32_t n, int32_t rows, int32_t in1, int32_t in2,
                        int32_t out, int32_t x1_div, int32_t x2_shared, void* stream);
int ssd_bias_bmm_bwd_x(const float* g, const float* w, float* dx, int32_t n, int32_t rows, int32_t in, int32_t out, int64_t w_set, void* stream);
int ssd_dueling_head_fwd(const float* y, float* q, int32_t n, int32_t T, int32_t B, int32_t inner, int32_t K, void* stream);
int ssd_dueling_head_bwd(const float* dq, float* dy, float* gs, int32_t n, int32_t T, int32_t B, int32_t inner, int32_t K, void* stream);
/* Make the row-chunk scratch of `stream` on the current device exist (SSD_ERR_DEVICE when it cannot be allocated). */
int ssd_bmm_reserve_scratch(void* stream);
/* Arithmetic of the LEARNER's matrix products from now on (process-wide; a launch takes the value at launch -- or capture -- time):
 *   2 (default)  f32: ssd_bias_bmm_* on exact-f32 MFMAs, ssd_gru_seq_* on two- / three-term split products (f32-equivalent);
 *   1            the labelled reduced-precision variant (config key learner_dtype: bf16): operands rounded to bf16, ONE
 *                v_mfma_f32_16x16x32_bf16 per 32 reduction indices in ssd_bias_bmm_fwd / _bwd (dx, dw), ssd_gru_seq_fwd / _bwd and
 *                the weight gradient of W_h; f32 accumulation, f32 parameters / optimiser state / loss.  The learner's encoder
 *                forward (ssd_policy_encode with `act`) takes precision 1 fragment images from ssd_policy_pack_encoder.
 * SSD_ERR_INVALID for any other value. */
int ssd_set_learner_precision(int32_t precision);
int ssd_learner_precision(void);

/* Weight (and bias) gradient of the encoder's Conv2d(3, 6, 3) on windows given as SSD_OBS_CODE class codes u8 [rows, V, V]
 * (V = 15 / 31): d_conv = dL/d(conv output) f32 [rows, 6, V-2, V-2] -> partial f32 [ssd_conv_wgrad_partial_rows(rows), 168]: per
 * wave, 162 weight-gradient sums in conv_w order [oc][ch][dy][dx] (the planes' 255/256 included) followed by the 6 bias sums; the
 * caller adds the rows (ssd_column_sums).  Rows past the last wave's windows are written as zeros. */
#define SSD_CONV_WGRAD_COLS 168
int ssd_conv_wgrad_partial_rows(int32_t rows);
int ssd_conv_wgrad_codes(const uint8_t* codes, const float* d_conv, float* partial, int32_t rows, int32_t view_edge, void* stream);

/* ---- fused rollout-time controller step (csrc/ssd_policy_mfma.hip) -------------------------------------------------------
 * What HomophilyMAC.select_actions_env / select_actions_inc evaluate (homophily_controller.py:30-65, 127-184 on top of
 * homophily_agent.py:20-27,154-214 and action_selectors.py:44-68) as THREE launches per timestep:
 *   ssd_policy_encode   rgb_preprocess: Conv2d(3, 6, 3) + LeakyReLU + Flatten + Linear(6 (V-2)^2, 32) [+ LeakyReLU]
 *   ssd_policy_head_env input tail (one-hot last action, agent id, sign of last reward, sign of received incentives, pos / scale)
 *                       -> fc1_env + LeakyReLU -> GRU cell -> dueling Q -> epsilon-greedy over the available actions
 *   ssd_policy_head_inc [inputs | one-hot(action)] -> fc1_inc + LeakyReLU -> GRU cell -> per ordered pair (i -> j)
 *                       [h_i | one-hot(a_j), pos_j / scale, orient_j, r_j, clean_j, apple_den_j] -> dueling Q -> epsilon-greedy, diagonal 0
 * All matrix products run on the 16-bit matrix cores (v_mfma_f32_16x16x32_{f16,bf16}, f32 accumulate).  `precision`:
 *   2  f32-equivalent (the default, the reference's dtype): every f32 operand is split into two f16 terms x = hi + lo (22
 *      significand bits) and a product is evaluated as hi*hi + hi*lo + lo*hi -- three MFMAs; the one-hot observation planes are exact
 *      in f16, so the conv needs two.  Weights and activations carry exact power-of-two scales that keep the lo terms in the normal
 *      f16 range; measured deviation from the reference's f32 Q-values < 1e-5 (tests/test_policy_mfma.py).
 *   1  bf16: single bf16 terms, one MFMA per product (the "bf16 Q-net" of BASELINE.json configs[1]; rollout inference only).
 * hidden = 64, n_feat = 32, conv_out = 6 are fixed (config/default.yaml:44,59-63).
 *
 * Weights are passed as kernel-ready FRAGMENT IMAGES built on the device by ssd_policy_pack_* from the reference-shaped f32
 * parameters (names and shapes of homophily_agent.py:37-125, [1, n, in, out] contiguous).  A fragment is the A operand of one
 * MFMA: 64 lanes x 8 values, lane l = (q = l >> 4, m = l & 15) holding row m of a 16-row output tile at the 8 reduction indices of
 * its quarter q -- 1 KiB, read by one ds_read_b128 / global_load_dwordx4 per lane.  All products are computed TRANSPOSED
 * (D^T = W^T X^T: the weight is the A operand, 16 activation rows are the B operand), and the reduction index of step s, quarter q,
 * element j is k = 32 s + 16 (j >> 2) + 4 q + (j & 3), which makes an MFMA result tile the B operand of the next product without
 * any lane movement (fc1 -> GRU -> dueling chain in registers).
 *
 * Head image (per agent, SSD_POLICY_IMAGE_BYTES(precision) bytes, 16-byte aligned), stored IN THE ORDER THE HEAD KERNEL CONSUMES IT in
 * pieces of 1 KiB (one LDS-DMA wave instruction each), 8 pieces to a chunk: the kernel STREAMS the image through a small ring in LDS
 * behind its own arithmetic (a head workgroup holds 48 KiB of it at a time, not 118 KB):
 *   per K-STEP c < 14 of the chain, [term t < precision][output tile ot < 4] fragments at piece 4 precision c + 4 t + ot:
 *     c = s: fc1 (s = the 32-deep half of the reduction); c = 2 + 2 g + s: GRU input side, gate g = r, z, n; c = 8 + 2 g + s: hidden side;
 *   then the resident chunk (8 pieces from piece 56 precision): fc2's [term t][s < 2] fragments (advantage rows, then the value row;
 *     inc: the h part) at + 2 t + s; the f32 tail at + 2 precision -- biases fc1[64] gru_i[192] gru_h[192] fc2[16], then (inc) the pair
 *     part of fc2 f32 [16 extra features][4] -- zero-padded to SSD_POLICY_TAIL_PIECES pieces; zero padding to the end of the chunk.
 * Activations are agent-major: inputs f32 [n, n_env, 64] (columns 0..31 = encoder output; the env head fills 32..63: tail then
 * zeros), h f32 [n, n_env, 64] updated in place. q_out (nullable): env f32 [n, n_env, n_actions]; inc f32 [n, n_env, n, 3].
 * Exploration draws: the package's counter generator keyed by (seed, *step, GLOBAL env id = env_id_base + env, agent[, j]) --
 * a shard of a larger job draws what the unsharded job draws for the same envs (ssd_dueling_pick uses the same key). */
#define SSD_POLICY_HEAD_FRAGS 58
#define SSD_POLICY_HEAD_TAIL_FLOATS (464 + 64)
#define SSD_POLICY_TAIL_PIECES 3
#define SSD_POLICY_IMAGE_PIECES(precision) (56 * (precision) + 8)              /* 14 K-steps x 4 precision pieces + the resident chunk */
#define SSD_POLICY_IMAGE_BYTES(precision) (SSD_POLICY_IMAGE_PIECES(precision) * 1024)
typedef struct ssd_policy_head {
    int32_t n_env, n_agents, n_actions, input_shape;
    float pos_scale;
    uint32_t seed;
    float* inputs;                 /* [n, n_env, 64] */
    float* h;                      /* [n, n_env, 64] */
    const void* weights;           /* [n, SSD_POLICY_IMAGE_BYTES(precision)] from ssd_policy_pack_head */
    const uint8_t* avail;          /* env: u8 [n_actions] or NULL */
    const float* epsilon;          /* device scalar */
    const int64_t* step;           /* device scalar */
    /* env head: previous timestep (last action -1 = none) and the current position */
    const int64_t* prev_actions;   /* [n_env, n] */
    const float* prev_reward;      /* [n_env, n] */
    const int64_t* prev_actions_inc; /* [n_env, n, n] */
    const float* pos;              /* [n_env, n, 2] */
    /* inc head: the env actions just taken, the PRE-step pose and this step's outcome */
    const int64_t* actions;        /* [n_env, n] */
    const float *pos_pre, *orient_pre;   /* [n_env, n, 2] */
    const float *reward, *clean_num, *apple_den;   /* [n_env, n] */
    int64_t* out_actions;          /* env: [n_env, n]; inc: [n_env, n, n] */
    float* q_out;                  /* nullable */
    /* env head, optional by-products for the runner (all nullable): the actions again as int32 (the env's action type), and
     * copies of the current pose (pos, orient [n_env, n, 2]) taken before the env step overwrites it */
    const float* orient;
    int32_t* out_actions_i32;
    float *pos_copy, *orient_copy;
    /* Optional: the head also files its results in the episode storage [n_env, t_slots, ...] at time slot *t_index (what
     * ssd_store_step_launch does as a separate launch; every pointer nullable = skipped; nothing is filed when *t_index >= t_slots),
     * and carries the runner state:
     *   env head: dst_actions, dst_actions_onehot, dst_pos, dst_orient (the current pose), prev_actions_out <- actions
     *   inc head: dst_actions_inc, prev_actions_inc_out <- actions_inc; dst_reward / dst_clean_num / dst_apple_den /
     *             dst_terminated <- this step's reward, clean_num, apple_den, terminated; prev_reward_out <- reward;
     *             ep_return += reward; *next_t_out = *t_index + 1 (next_t_out must not alias t_index). */
    const int64_t* t_index;
    int32_t t_slots;
    float *dst_pos, *dst_orient, *dst_actions_onehot, *dst_reward, *dst_clean_num, *dst_apple_den;
    uint8_t* dst_terminated;
    const uint8_t* terminated;     /* [n_env] */
    int64_t *dst_actions, *dst_actions_inc;
    int64_t *prev_actions_out, *prev_actions_inc_out;
    float *prev_reward_out, *ep_return;
    int64_t* next_t_out;
    /* ---- ABI 2 ---- */
    int32_t precision;             /* 2 (f32-equivalent, default when 0) or 1 (bf16) -- must match the image */
    uint32_t env_id_base;          /* global id of env 0 of this shard (exploration key) */
    /* env head: the encoder's unfinished output (ssd_policy_encode with bands > 1): features = LeakyReLU(lin_b + sum over bands
     * of feat_part[band][agent-major row][32]); written to inputs[:, 0..31] by this launch.  NULL: inputs[:, 0..31] are final. */
    const float* feat_part;
    int32_t feat_bands;
    const float* lin_b;            /* [32] */
    /* ---- ABI 3 ---- */
    /* env head: which blocks _build_inputs appends after the 32 encoder features (homophily_controller.py:137-184, in that order;
     * SSD_INPUT_* bits, or-ed with SSD_INPUT_EXPLICIT so that the empty set can be told from 0).  0 = SSD_INPUT_FLAGS_SHIPPED
     * (config/default.yaml).  input_shape must equal 32 + the blocks' widths and
     * input_shape + n_actions <= 64; obs_others_last_action (n * n_actions more columns) does not fit the 64-column image and is
     * rejected with SSD_ERR_UNSUPPORTED. */
    uint32_t input_flags;
    /* Counter hand-over of the pipelined rollout (all nullable).  inc head: *next_step_out = *step + 1 (like next_t_out for the time
     * index).  env head: *t_copy_out = *t_index, *step_copy_out = *step.  A launch never writes a scalar that it reads: the env head
     * reads the masters and writes the copies, the inc head reads the copies and writes the masters. */
    int64_t *next_step_out, *t_copy_out, *step_copy_out;
    /* ---- ABI 7 ---- */
    /* Compact runner state for the env head's input phase (all optional).  recv_inc: the previous step's incentive actions as
     * RECEIVER-major bytes u8 [n(receiver), n_env, 16] (byte g = what giver g sent this receiver: 0 / 1 / 2; the receiver's own byte
     * and bytes >= n are 0): when given, the env head reads it instead of prev_actions_inc [n_env, n(giver), n(receiver)] i64 -- one
     * 4-byte load per lane over 256 contiguous bytes per 16-row tile instead of n 8-byte loads over n x 16 cache lines.
     * recv_inc_out: the inc head writes byte `agent` of every receiver's record next to prev_actions_inc_out (the caller zeroes the
     * array when an episode opens, like prev_actions_inc).  avail_bits: bit 31 set = bits 0 .. n_actions - 1 ARE the availability
     * mask of `avail` (a constant of the env class): the env head then issues no loads for it. */
    const uint8_t* recv_inc;
    uint8_t* recv_inc_out;
    uint32_t avail_bits;
} ssd_policy_head;
#define SSD_INPUT_LAST_ACTION 1u   /* obs_last_action: one-hot of the previous env action, n_actions columns */
#define SSD_INPUT_AGENT_ID    2u   /* obs_agent_id: one-hot of the agent, n columns */
#define SSD_INPUT_REWARD      4u   /* obs_reward: sign of the previous reward */
#define SSD_INPUT_INC_REWARD  8u   /* obs_inc_reward: sign(#rewards - #punishments received at the previous step) */
#define SSD_INPUT_DISTANCE    16u  /* obs_distance: 1 - |pos - pos_g| / pos_scale for every agent g, n columns */
#define SSD_INPUT_AGENT_POS   32u  /* obs_agent_pos: pos / pos_scale, 2 columns */
#define SSD_INPUT_OTHERS_LAST_ACTION 64u   /* obs_others_last_action: every agent's last-action one-hot, n * n_actions columns, between
                                             the received-incentive sign and the distances -- ssd_build_inputs_flags only */
#define SSD_INPUT_EXPLICIT    0x80000000u   /* marks a given flag word (the empty set is SSD_INPUT_EXPLICIT alone) */
#define SSD_INPUT_FLAGS_SHIPPED (SSD_INPUT_LAST_ACTION | SSD_INPUT_AGENT_ID | SSD_INPUT_REWARD | SSD_INPUT_INC_REWARD | SSD_INPUT_AGENT_POS)
/* ssd_build_inputs for ANY _build_inputs flag set (homophily_controller.py:137-184; the learner's time-batched input assembly):
 * input_flags as in ssd_policy_head (0 = the shipped set) plus SSD_INPUT_OTHERS_LAST_ACTION.  Writes the blocks, in the reference's
 * order, into out[:, out_offset : out_offset + width]; width = ssd_build_inputs_width(...). */
int ssd_build_inputs_width(int32_t n_agents, int32_t n_actions, uint32_t input_flags);
int ssd_build_inputs_flags(int32_t batch, int32_t n_agents, int32_t n_actions, int32_t t0, uint32_t input_flags,
                           const int64_t* last_actions, const float* last_reward, const int64_t* last_actions_inc, const float* pos,
                           float pos_scale, float* out, int32_t out_stride, int32_t out_offset, void* stream);
int ssd_policy_head_env(const ssd_policy_head* args, void* stream);
int ssd_policy_head_inc(const ssd_policy_head* args, void* stream);
/* How a head launch of (n_env, n_agents) is cut on the current device: workgroups per agent, compute waves per workgroup (a 16-row
 * tile each) and the number of tiles the busiest wave walks -- 1 while the grid fits the chip (the kernels without a back edge),
 * more beyond (Cleanup-10 x 8192: the looped instantiations).  fused_with_encoder: the inc head inside ssd_policy_head_inc_encode. */
int ssd_policy_head_plan(int32_t n_env, int32_t n_agents, int32_t fused_with_encoder, int32_t* workgroups_per_agent, int32_t* compute_waves,
                         int32_t* tiles_per_wave);
/* The inc head of timestep t and ssd_policy_encode of timestep t + 1 as ONE launch (declared below the encoder's arguments):
 * ssd_policy_head_inc_encode. */

/* The reference-shaped parameters of one head (homophily_agent.py:37-125; every tensor [1, n, in, out] / [1, n, 1, out] f32,
 * contiguous, device memory).  w_i / w_h / b_i / b_h in (r, z, n) order.  env: fc1_in = input_shape, fc2_in = 64, fc2_out = n_actions;
 * inc: fc1_in = input_shape + n_actions, fc2_in = 64 + n_actions + 7, fc2_out = 3. */
typedef struct ssd_policy_head_params {
    const float *fc1_w, *fc1_b;
    const float *w_i[3], *w_h[3], *b_i[3], *b_h[3];
    const float *fc2_w, *fc2_b, *fc2_v_w, *fc2_v_b;
    int32_t n_agents, fc1_in, fc2_in, fc2_out;
} ssd_policy_head_params;
/* image: [n, SSD_POLICY_IMAGE_BYTES(precision)] */
int ssd_policy_pack_head(const ssd_policy_head_params* params, int32_t precision, void* image, void* stream);

/* ssd_policy_encode: rgb_preprocess for 15 x 15 (view_size 7) and 31 x 31 (view_size 15) windows of the SIMPLIFIED palette.
 * Input: one byte per window cell in either alphabet (`alphabet`: SSD_CODE_CLASS = the SSD_OBS_CODE classes of an episode storage,
 * SSD_CODE_CHANNEL_MASK = the side buffer of ssd_obs_out.obs_code); the three colour planes are rebuilt in LDS as one-hot bytes, so the
 * conv is a banded (Toeplitz) GEMM over 16 batch rows per MFMA and the whole encoder runs on the matrix cores:
 *   conv   D[(2 output channels) x (8 positions of an output row)][16 batch rows] += A[.][K] x B[K][batch rows] per input row dy:
 *          K = 3 planes x the 10-cell window of the 8 positions, laid out as three quarters of 8 cells + a "tail" quarter that holds
 *          cells 8, 9 of the three planes (30 of 32 K entries carry data; 9 of them are taps of a given output).
 *          A = Toeplitz image of the 3 x 3 taps (ssd_policy_pack_encoder), B = plane bytes read from LDS (one ds_read_b64 + 4 v_perm)
 *   Linear two neighbouring conv result tiles (16 positions x 2 channels) ARE the B operand of a Linear K-step (no transposition,
 *          no LDS)
 * The class of row (env b, agent i) cell c is read at codes[b * env_stride + (*slot_t) * slot_stride + i * agent_stride + c]
 * (slot_t NULL: slot 0): the dense side buffer of ssd_obs_out.obs_code (env_stride = n * agent_stride, slot_stride = 0) or an
 * episode storage of class codes u8 [n_env, t_slots, n, V, V] (agent_stride = V * V).  code_bytes = readable bytes behind `codes`;
 * the kernel reads whole aligned 4-byte words, never one that holds no byte of [codes, codes + code_bytes).
 * Output rows are agent-major (i * n_env + b) when agent_major, else (b * n + i).
 *   V = 15: one launch computes the whole Linear sum; out[row * out_stride + 0..31] = LeakyReLU(. + lin_b)   (part = NULL)
 *   V = 31: the 29 output rows are cut into SSD_ENCODE_BANDS(31) = 3 bands evaluated by different workgroups; band k writes its
 *           partial Linear sum to part[k][row][32] and the consumer (ssd_policy_head_env: feat_part / feat_bands / lin_b) adds
 *           them in band order, adds the bias and applies the LeakyReLU (out = NULL).
 * slot_t_copy (nullable) receives *slot_t: a second copy of the time index for the kernels that file results and advance slot_t
 * (ssd_policy_head.t_index / next_t_out, ssd_store_step_launch).  counter_inc (nullable): *counter_inc += 1 (the exploration-draw
 * counter read by the heads that follow; this kernel does not read it).
 * Fragment images (ssd_policy_pack_encoder from conv_w f32 [6, 3, 3, 3] and lin_w f32 [32, 6 (V-2)^2]):
 *   conv_frags [term][s 3][dy 3][lane][8]: element (q, m, j) = tap weight w[oc][ch][dy][d] * 255/256, oc = 2 s + (m >> 3),
 *              position p = m & 7; q < 3: ch = q, cell = j; q = 3: ch = j >> 1, cell = 8 + (j & 1) (j < 6); d = cell - p in 0..2, else 0
 *   lin_frags  [unit = ((y * XTP + xtp) * 3 + s)][output tile 2][term][lane][8]: element (q, m, j) = lin_w[16 Mt + m][oc * P + y * O + x],
 *              r = 4 q + (j & 3), oc = 2 s + (r >> 3), x = 8 (2 xtp + (j >> 2)) + (r & 7), 0 where x >= O   (XTP = 1 / 2 pairs of
 *              8-position tiles per output row) */
/* SSD_ENCODE_LAYOUT_LUT (ssd_policy_encode_args.layout; launches without `act`): the CONVOLUTION AS A TABLE SUM.  A window cell is one
 * of four classes and lights at most one plane at 255/256, so the three taps of input row dy contribute to all six channels a value
 * that depends only on the three classes under them: conv[c](y, x) = sum_dy T[dy][cls(y+dy, x) + 4 cls(y+dy, x+1) + 16 cls(y+dy, x+2)][c]
 * (T[0] carries the bias) -- exact f32 sums of entries built by ssd_policy_pack_encoder_lut, no matrix-core work for the conv.  Images:
 *   conv_frags = the table f32 [3 dy][64 idx][6 channels] (SSD_ENCODE_LUT_TABLE_BYTES; precision 2: scaled by 256, the split scale of
 *                the conv activations), classes as in SSD_OBS_CODE (2 waste -> R, 1 apple -> G, 3 wall / agent -> B);
 *   lin_frags  = [K-step gs][output tile 2][term][lane][8] with K-steps numbered through the bands (band k: ceil(rows_k * O / 4) of
 *                them, SSD_ENCODE_LUT_KSTEPS(V) in all): element (q, m, j) = lin_w[16 Mt + m][j * P + y * O + x] for j < 6, the position
 *                p = 4 s + q of the band (s = gs - the band's first K-step; y = band * R + p / O, x = p mod O), 0 for j >= 6 and past the band.
 * The kernel keeps the window rows in LDS as packed 2-bit classes; lane (q, m) evaluates position 4 s + q of batch row m and its six
 * channel values ARE its B operand of the Linear's K-step s: 258 MFMAs per 16-row tile (15 x 15) instead of 702. */
#define SSD_ENCODE_LAYOUT_TOEPLITZ 0
#define SSD_ENCODE_LAYOUT_LUT 1
#define SSD_ENCODE_LUT_TABLE_BYTES (3 * 64 * 6 * 4)
#define SSD_ENCODE_LUT_KSTEPS(V) ((V) == 31 ? 73 + 73 + 66 : 43)
#define SSD_ENCODE_LUT_LIN_BYTES(V, precision) (SSD_ENCODE_LUT_KSTEPS(V) * 2 * (precision) * 1024)
#define SSD_ENCODE_BANDS(V) ((V) == 31 ? 3 : 1)
#define SSD_ENCODE_UNITS(V) ((V) == 31 ? 29 * 2 * 3 : 13 * 1 * 3)
#define SSD_ENCODE_CONV_FRAG_BYTES(V, precision) ((precision) * 9 * 1024)
#define SSD_ENCODE_LIN_FRAG_BYTES(V, precision) (SSD_ENCODE_UNITS(V) * 2 * (precision) * 1024)
typedef struct ssd_policy_encode_args {
    const uint8_t* codes;
    int64_t code_bytes, env_stride, slot_stride, agent_stride;
    const int64_t* slot_t;
    int32_t rows, view_edge, n_agents, agent_major, precision;
    const void *conv_frags, *lin_frags;
    const float *conv_b, *lin_b;
    float* out; int32_t out_stride;
    float* part;
    int64_t* slot_t_copy;
    int64_t* counter_inc;
    int32_t alphabet;              /* SSD_CODE_CLASS / SSD_CODE_CHANNEL_MASK */
    float* act;                    /* nullable (precision 2 only): LeakyReLU(conv) f32 [rows, 6, V-2, V-2], row = b * n + i: the
                                      activations the learner's backward needs (the training forward of the encoder) */
    int32_t slot_add;              /* the time slot read is *slot_t + slot_add (the pipelined rollout encodes slot t + 1 while the
                                      device time index still says t); the caller keeps it inside the storage */
    int32_t layout;                /* ABI 7: SSD_ENCODE_LAYOUT_TOEPLITZ (0: images of ssd_policy_pack_encoder) or SSD_ENCODE_LAYOUT_LUT (images
                                      of ssd_policy_pack_encoder_lut; not with `act`) */
} ssd_policy_encode_args;
int ssd_policy_encode(const ssd_policy_encode_args* args, void* stream);
/* ssd_policy_head_inc(inc_args) and ssd_policy_encode(enc_args) as ONE launch -- the pipelined rollout's third launch of a timestep:
 * both follow the env step of t and share no data (the inc head reads the input rows of t, the encoder reads the observation of slot
 * t + 1 and must write a DIFFERENT `inputs` buffer / `part`), so one launch-to-launch gap of the timestep disappears.  enc_args: no
 * act, no slot_t_copy / counter_inc (the heads hand the counters over, see ssd_policy_head); same precision as inc_args. */
int ssd_policy_head_inc_encode(const ssd_policy_head* inc_args, const ssd_policy_encode_args* enc_args, void* stream);
/* conv_b (f32 [6]): read only for the range bound of the conv activations (see SSD_ERRBIT_F16_RANGE). */
/* conv_w f32 [6, 3, 3, 3], conv_b [6], lin_w [32, 6 (V-2)^2] -> the images of SSD_ENCODE_LAYOUT_LUT: table (SSD_ENCODE_LUT_TABLE_BYTES,
 * 16-byte aligned) and lin_frags (SSD_ENCODE_LUT_LIN_BYTES(V, precision)). */
int ssd_policy_pack_encoder_lut(const float* conv_w, const float* conv_b, const float* lin_w, int32_t view_edge, int32_t precision, void* table,
                                void* lin_frags, void* stream);
int ssd_policy_pack_encoder(const float* conv_w, const float* conv_b, const float* lin_w, int32_t view_edge, int32_t precision, void* conv_frags,
                            void* lin_frags, void* stream);

/* Range guard of the two-term f16 split products (precision 2; homophily_agent.py:154-208 computed in f32 by the reference).  The
 * splits use fixed power-of-two scales: head weights x 64, head activations x 16, conv weights / conv activations / Linear weights
 * x 256, the learner's recurrence W_h x 64.  A scaled value beyond f16's 65 504 would turn the product into inf / NaN silently, so
 * the kernels raise the sticky bit SSD_ERRBIT_F16_RANGE in the device's numeric-status word instead: the pack kernels for every
 * weight term and -- from the weights alone -- for the worst-case conv activation; the rollout heads for the activations they scale
 * (the fc1 operand and fc1's output; hidden states are < 1); the recurrence kernel for W_h.  ssd_numeric_status returns and
 * clears the word (it synchronises the device); ssd_poll_error ORs it into an env's bits. */
#define SSD_ERRBIT_F16_RANGE 32
int ssd_numeric_status(int32_t* bits);

/* ---- COUNTER-mode generator (shared definition) ---------------------------------------------
 * Two levels, so that the expensive part is computed once per EPISODE (by the reset call) and kept in the env state:
 *   episode = number of resets of this env including the one that opened the current episode (1, 2, ...)
 *   b[0..3] = philox4x32_10(counter = {0, 0, env_global_id, episode}, key = {seed_lo, seed_hi})
 *   c       = call index inside the episode: 0 for the reset call, (steps completed so far) + 1 for a step call
 *   x(stream, k) = mix32(b[stream] ^ (c << 16) ^ k),  stream = SSD_STREAM_*  (one Philox word per stream; k, c < 65536)
 *   mix32(x): x ^= x >> 17; x *= 0xed5ad4bb; x ^= x >> 11; x *= 0xac4c1b51; x ^= x >> 15; x *= 0x31848bab; x ^= x >> 14
 *             (the "triple32" integer hash, a bijection on 32 bits)
 *   uniform number k of the call: u = (x(UNIFORM, k) >> 8) * 2^-24 as double, compared `u < p` in fp64
 *   shuffles: stable sort of the items by (x(stream, item index) >> 8, item index)   (24-bit key: key | index fits 32 bits)
 *   spawn rotation of agent a: x(SPAWN_ROT, a) >> 30
 *   spawn list shuffle of agent a (random_spawn_point): sort of the list elements e = copy * n_spawn_points + point id
 *             (copy = 0, and 1 for Cleanup's duplicated list) by (x(SPAWN_ROT, 256 + 32 a + e) >> 8, e); at most SSD_MAX_SPAWN = 16 points
 * ssd_state.epoch carries `episode`; importing it re-derives b.                                                        */
enum { SSD_STREAM_UNIFORM = 0, SSD_STREAM_MOVE = 1, SSD_STREAM_WASTE = 2, SSD_STREAM_SPAWN_ROT = 3 };

#ifdef __cplusplus
}
#endif
#endif /* SSD_HIP_H */
