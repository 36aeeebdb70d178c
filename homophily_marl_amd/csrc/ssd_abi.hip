// ssd_abi.hip -- the extern "C" boundary of libssd_hip.so (declared in include/ssd_hip.h).
// Host code only: argument checking, map parsing, device allocation, kernel launches.  No torch types, no
// exceptions across the boundary, nothing aborts; all launches are asynchronous on the caller's stream.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "ssd_device.h"

using namespace ssd;

struct ssd_env {
    DevSpec hs;        // host copy
    DevSpec* dspec;    // device copy
    DevState st;
    int device;
    int n_spawn;
};

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, const char* a = "") {
    std::snprintf(g_err, sizeof g_err, fmt, a);
    return code;
}
#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return fail(SSD_ERR_DEVICE, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

static uint32_t magic(uint32_t d) { return d <= 1 ? 0xFFFFFFFFu : (uint32_t)(0x100000000ull / d) + 1u; }

extern "C" {

int ssd_abi_version(void) { return SSD_ABI_VERSION; }
const char* ssd_last_error(void) { return g_err; }

// `u < p` for the COUNTER-mode draw u = m * 2^-24 (24-bit integer m): p * 2^24 is an exact fp64 scaling, so u < p  <=>  m < ceil(p * 2^24)
static uint32_t draw_threshold(double p) { return p >= 1.0 ? (1u << 24) : !(p > 0.0) ? 0u : (uint32_t)std::ceil(p * 16777216.0); }

int ssd_create(const ssd_config* cfg, ssd_env** out) {
    if (!cfg || !out || !cfg->ascii_map) return fail(SSD_ERR_INVALID, "null argument");
    if (cfg->n_agents < 1 || cfg->n_agents > SSD_MAX_AGENTS) return fail(SSD_ERR_INVALID, "n_agents out of range (1..10)");
    if (cfg->height < 3 || cfg->width < 3 || cfg->height > 255 || cfg->width > 255 || cfg->height * cfg->width > SSD_MAX_CELLS)
        return fail(SSD_ERR_INVALID, "map size out of range");
    if (cfg->n_env < 1) return fail(SSD_ERR_INVALID, "n_env must be >= 1");
    if (cfg->view_size < 0 || cfg->view_size > 31) return fail(SSD_ERR_INVALID, "view_size out of range (0..31)");
    {   // LDS budget: 4 waves (envs) per workgroup must fit the 160 KiB of a CU
        const long V = 2L * cfg->view_size + 1, pm = (cfg->height + 2L * cfg->view_size) * (cfg->width + 2L * cfg->view_size);
        const long per_wave = 2L * ((cfg->height * cfg->width + 15) & ~15) + pm + 16 + cfg->n_agents * 3L * V * V + 32 + 64 + (cfg->n_agents * SSD_CODE_AGENT_STRIDE(V) <= 2560 ? cfg->n_agents : 1) * SSD_CODE_AGENT_STRIDE(V) + 16;
        if (per_wave * 4 > 160 * 1024) return fail(SSD_ERR_INVALID, "map / view_size / n_agents exceed the LDS budget of one workgroup");
    }
    if (cfg->spawn_rotation > 3) return fail(SSD_ERR_INVALID, "spawn_rotation must be -1..3");
    if (cfg->env_kind != SSD_ENV_CLEANUP && cfg->env_kind != SSD_ENV_HARVEST) return fail(SSD_ERR_INVALID, "env_kind");
    if (cfg->rng_mode != SSD_RNG_TAPE && cfg->rng_mode != SSD_RNG_COUNTER) return fail(SSD_ERR_INVALID, "rng_mode");

    ssd_env* E = new (std::nothrow) ssd_env();
    if (!E) return fail(SSD_ERR_NOMEM, "host allocation failed");
    std::memset(E, 0, sizeof *E);
    DevSpec& S = E->hs;
    S.kind = cfg->env_kind; S.H = cfg->height; S.W = cfg->width; S.HW = S.H * S.W; S.GS = (S.HW + 15) & ~15;
    S.n = cfg->n_agents; S.N = cfg->n_env; S.v = cfg->view_size; S.V = 2 * S.v + 1; S.VV = S.V * S.V; S.VVp = (S.VV + 3) & ~3;
    S.Wp = S.W + 2 * S.v; S.PMS = ((S.H + 2 * S.v) * S.Wp + 15) & ~15;
    S.vshift = 0; while ((1 << S.vshift) < S.V) ++S.vshift;
    S.episode_limit = cfg->episode_limit; S.spawn_rotation = cfg->spawn_rotation < 0 ? -1 : cfg->spawn_rotation;
    S.obs_color = cfg->obs_color == SSD_COLOR_FULL ? SSD_COLOR_FULL : SSD_COLOR_SIMPLIFIED;
    S.rng_mode = cfg->rng_mode; S.n_actions = S.kind == SSD_ENV_CLEANUP ? 9 : 8;
    S.env_id_base = cfg->env_id_base; S.seed_lo = (uint32_t)cfg->seed; S.seed_hi = (uint32_t)(cfg->seed >> 32);
    S.magic_W = magic(S.W); S.magic_V = magic(S.V); S.magic_VV = magic(S.VV); S.magic_3VV = magic(3 * S.VV); S.magic_HW = magic(S.HW);
    S.thr_dep = cfg->threshold_depletion; S.thr_res = cfg->threshold_restoration;
    S.p_waste = cfg->waste_spawn_prob; S.p_apple = cfg->apple_respawn_prob;
    for (int i = 0; i < 4; ++i) { S.harvest_p[i] = cfg->harvest_spawn_prob[i]; S.harvest_thr[i] = draw_threshold(S.harvest_p[i]); }
    if ((long)S.n * 3 * S.VV >= 65536) { delete E; return fail(SSD_ERR_INVALID, "n_agents * 3 * V * V must stay below 65536"); }

    // Row-major scan of the layout (map_env.py:143-148, cleanup.py:77-90, harvest.py:31-35).
    std::vector<int> spawn;
    for (int i = 0; i < S.HW; ++i) {
        const char ch = cfg->ascii_map[i];
        uint8_t code = C_EMPTY;
        bool overflow = false;
        if (ch == '@') code = C_WALL;
        else if (ch == 'P') spawn.push_back(i);
        else if (S.kind == SSD_ENV_CLEANUP) {
            if (ch == 'B') { if (S.n_apple < SSD_MAX_SITES) S.apple[S.n_apple++] = (uint16_t)i; else overflow = true; }
            else if (ch == 'H') { code = C_WASTE; if (S.n_waste < SSD_MAX_SITES) S.waste[S.n_waste++] = (uint16_t)i; else overflow = true; }
            else if (ch == 'R') code = C_RIVER;
            else if (ch == 'S') code = C_STREAM;
        } else if (ch == 'A') { code = C_APPLE; if (S.n_apple < SSD_MAX_SITES) S.apple[S.n_apple++] = (uint16_t)i; else overflow = true; }
        if (overflow) { delete E; return fail(SSD_ERR_INVALID, "too many apple / waste sites (max 256)"); }
        S.reset_grid[i] = code;
    }
    // compute_probabilities (cleanup.py:189-204) for every possible waste count, reference operation order, plain fp64
    for (int current = 0; current <= SSD_MAX_SITES; ++current) {
        volatile double wd = 0, pa = 0, pw = 0;
        const int potential = S.n_waste;
        if (potential > 0) {
            const int free_area = potential - current;
            volatile double q = (double)free_area / (double)potential;
            wd = 1 - q;
        }
        if (!(wd >= S.thr_dep)) {
            pw = S.p_waste;
            if (wd <= S.thr_res) pa = S.p_apple;
            else {
                volatile double num = wd - S.thr_res, den = S.thr_dep - S.thr_res;
                volatile double frac = num / den;
                volatile double one_minus = 1 - frac;
                pa = one_minus * S.p_apple;
            }
        }
        S.tab_p[current][0] = pa; S.tab_p[current][1] = pw;
        const double pa_ = pa, pw_ = pw;
        S.tab_thr[current][0] = draw_threshold(pa_); S.tab_thr[current][1] = draw_threshold(pw_);
        S.tab_thr[current][2] = (pa_ > 0 ? 1u : 0u) | (!(std::fabs(pw_) <= 1e-8) ? 2u : 0u);
        S.tab_thr[current][3] = 0;
    }
    for (int a = 0; a <= SSD_MAX_CELLS; ++a) S.tab_den[a] = (float)((double)a / (double)S.HW);
    for (int l = 0; l < 64; ++l)
        for (int ch = 0; ch < 4; ++ch) {
            S.site_t[l][ch] = ch * 64 + l < S.n_apple ? S.apple[ch * 64 + l] : (uint16_t)0;
            S.site_t[l][4 + ch] = ch * 64 + l < S.n_waste ? S.waste[ch * 64 + l] : (uint16_t)0;
        }
    E->n_spawn = (int)spawn.size();
    S.random_spawn = cfg->random_spawn_point ? 1 : 0; S.n_spawn = (int)spawn.size();
    S.spawn_len = (S.kind == SSD_ENV_CLEANUP ? 2 : 1) * S.n_spawn;   // Cleanup's constructor appends every point again (cleanup.py:79-80)
    if (S.random_spawn && spawn.size() > SSD_MAX_SPAWN) { delete E; return fail(SSD_ERR_UNSUPPORTED, "random_spawn_point supports at most SSD_MAX_SPAWN spawn points"); }
    for (size_t i = 0; i < spawn.size() && i < SSD_MAX_SPAWN; ++i) S.spawn_all[i] = (uint16_t)spawn[i];
    if ((int)spawn.size() < S.n) { delete E; return fail(SSD_ERR_INVALID, "There are not enough spawn points! Check your map?"); }
    // spawn_point() returns the LAST free spawn point (map_env.py:779-784): agent a gets the a-th from the end.
    for (int a = 0; a < S.n; ++a) S.spawn_cell[a] = (uint16_t)spawn[spawn.size() - 1 - a];
    // agents must never stand on the border (the kernels index t = P +- W, P +- 1 without a bounds test on rows)
    for (int a = 0; a < S.n; ++a) {
        const int r = S.spawn_cell[a] / S.W, c = S.spawn_cell[a] % S.W;
        if (r == 0 || c == 0 || r == S.H - 1 || c == S.W - 1) { delete E; return fail(SSD_ERR_INVALID, "spawn point on the map border"); }
    }
    for (int i = 0; i < S.HW; ++i) {
        const int r = i / S.W, c = i % S.W;
        if ((r == 0 || c == 0 || r == S.H - 1 || c == S.W - 1) && S.reset_grid[i] != C_WALL) {
            delete E; return fail(SSD_ERR_INVALID, "the map border must be wall ('@')");
        }
    }
    // Full-colour LUT (map_env.py:33-62 DEFAULT_COLOURS; cleanup.py:14-17 CLEANUP_COLORS), rows: cell codes 0..5, then 5 + agent char.
    static const uint8_t agent_rgb[10][3] = {{0, 0, 0}, {159, 67, 255}, {2, 81, 154}, {204, 0, 204}, {216, 30, 54},
                                             {254, 151, 0}, {205, 155, 155}, {99, 99, 255}, {250, 204, 255}, {238, 223, 16}};
    uint8_t (*lut)[3] = (uint8_t (*)[3])S.lut;
    lut[C_WALL][0] = lut[C_WALL][1] = lut[C_WALL][2] = 180;
    lut[C_APPLE][1] = 255;
    if (S.kind == SSD_ENV_CLEANUP) {
        lut[C_WASTE][0] = 99; lut[C_WASTE][1] = 156; lut[C_WASTE][2] = 194;
        lut[C_RIVER][0] = lut[C_STREAM][0] = 113; lut[C_RIVER][1] = lut[C_STREAM][1] = 75; lut[C_RIVER][2] = lut[C_STREAM][2] = 24;
    }
    for (int ch = 1; ch <= 9; ++ch) std::memcpy(lut[5 + ch], agent_rgb[ch], 3);

    // ---- device side ----
    int prev = 0;
    E->device = cfg->device;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess) e = hipSetDevice(cfg->device);
    const size_t N = (size_t)S.N, n = (size_t)S.n;
    if (e == hipSuccess) e = hipMalloc((void**)&E->dspec, sizeof(DevSpec));
    if (e == hipSuccess) e = hipMalloc((void**)&E->st.grid, N * S.GS);
    if (e == hipSuccess) e = hipMalloc((void**)&E->st.agents, N * n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&E->st.hdr, N * sizeof(ssd::EnvHdr));
    if (e == hipSuccess) e = hipMalloc((void**)&E->st.err, 4);
    if (e == hipSuccess) e = hipMemcpy(E->dspec, &S, sizeof(DevSpec), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(E->st.grid, 0, N * S.GS);
    if (e == hipSuccess) e = hipMemset(E->st.agents, 0, N * n * 8);
    if (e == hipSuccess) {      // counts unknown (recount at the first step), everything else 0
        std::vector<ssd::EnvHdr> hz(N);
        memset(hz.data(), 0, N * sizeof(ssd::EnvHdr));
        for (size_t i = 0; i < N; ++i) hz[i].counts = 0xFFFFFFFFu;
        e = hipMemcpy(E->st.hdr, hz.data(), N * sizeof(ssd::EnvHdr), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemset(E->st.err, 0, 4);
    if (e == hipSuccess && !numeric_err_word()) e = hipErrorOutOfMemory;      // exists before anything captures a graph
    if (e == hipSuccess && !bmm_scratch(nullptr)) e = hipErrorOutOfMemory;           // (the affine layers' backward scratch: same reason)
    if (e == hipSuccess) {
        // a fresh handle holds the reset image with agents on their spawn cells (the reference constructor
        // also builds agents before the first reset, map_env.py:149)
        std::vector<uint8_t> g((size_t)S.GS, 0);
        std::memcpy(g.data(), S.reset_grid, (size_t)S.HW);
        std::vector<uint32_t> rec(n);
        for (size_t a = 0; a < n; ++a) {
            const uint32_t r = S.spawn_cell[a] / S.W, c = S.spawn_cell[a] % S.W;
            rec[a] = r | (c << 8) | ((uint32_t)(S.spawn_rotation < 0 ? 0 : S.spawn_rotation) << 16);
        }
        std::vector<uint8_t> gall(N * S.GS);
        std::vector<uint32_t> rall(N * n * 2, 0u);      // {arec, episode reward = 0}
        for (size_t i = 0; i < N; ++i) {
            std::memcpy(&gall[i * S.GS], g.data(), (size_t)S.GS);
            for (size_t a = 0; a < n; ++a) rall[(i * n + a) * 2] = rec[a];
        }
        e = hipMemcpy(E->st.grid, gall.data(), gall.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(E->st.agents, rall.data(), rall.size() * 4, hipMemcpyHostToDevice);
    }
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        fail(SSD_ERR_DEVICE, "device setup failed: %s", hipGetErrorString(e));
        ssd_destroy(E);
        return SSD_ERR_DEVICE;
    }
    *out = E;
    return SSD_OK;
}

int ssd_destroy(ssd_env* E) {
    if (!E) return SSD_OK;
    (void)hipFree(E->dspec); (void)hipFree(E->st.grid); (void)hipFree(E->st.agents); (void)hipFree(E->st.hdr); (void)hipFree(E->st.err);
    delete E;
    return SSD_OK;
}

int ssd_get_info(const ssd_env* E, ssd_info* o) {
    if (!E || !o) return fail(SSD_ERR_INVALID, "null argument");
    o->n_actions = E->hs.n_actions; o->n_apple_sites = E->hs.n_apple; o->n_waste_sites = E->hs.n_waste;
    o->n_spawn_points = E->n_spawn; o->max_uniforms = E->hs.n_apple + E->hs.n_waste; o->obs_edge = E->hs.V;
    return SSD_OK;
}

static int make_tape(const ssd_env* E, const ssd_tape* t, DevTape* d) {
    std::memset(d, 0, sizeof *d);
    if (E->hs.rng_mode == SSD_RNG_TAPE) {
        if (!t || !t->uniforms || !t->move_order) return fail(SSD_ERR_INVALID, "TAPE mode needs tape.uniforms and tape.move_order");
        if (E->hs.kind == SSD_ENV_CLEANUP && E->hs.n_waste > 0 && !t->waste_order) return fail(SSD_ERR_INVALID, "TAPE mode needs tape.waste_order for Cleanup");
        if (E->hs.spawn_rotation < 0 && !t->spawn_rot) return fail(SSD_ERR_INVALID, "TAPE mode with random rotation needs tape.spawn_rot");
        if (t->uniforms_stride < 1) return fail(SSD_ERR_INVALID, "tape.uniforms_stride");
    }
    if (t) { d->move_order = t->move_order; d->uniforms = t->uniforms; d->ustride = t->uniforms_stride; d->waste_order = t->waste_order; d->spawn_rot = t->spawn_rot;
             d->spawn_order = t->spawn_order; }
    return SSD_OK;
}
static DevStepOut make_so(const ssd_step_out* o) {
    DevStepOut d; std::memset(&d, 0, sizeof d);
    if (o) { d.reward = o->reward; d.clean_num = o->clean_num; d.apple_den = o->apple_den; d.terminated = o->terminated;
             d.collective = o->collective_return; d.equality = o->equality; d.n_draws = o->n_draws; }
    return d;
}
static int make_oo(const ssd_env* E, const ssd_obs_out* o, DevObsOut* d) {
    std::memset(d, 0, sizeof *d);
    if (!o) return fail(SSD_ERR_INVALID, "null ssd_obs_out");
    if (o->obs) {
        if (o->obs_format < SSD_OBS_F32 || o->obs_format > SSD_OBS_CODE) return fail(SSD_ERR_INVALID, "obs_format");
        if (o->obs_format == SSD_OBS_CODE && E->hs.obs_color != SSD_COLOR_SIMPLIFIED) return fail(SSD_ERR_INVALID, "SSD_OBS_CODE needs simplified colours");
        if (((uintptr_t)o->obs & 15) != 0) return fail(SSD_ERR_INVALID, "obs must be 16-byte aligned");
    }
    if (o->state && ((uintptr_t)o->state & 15) != 0) return fail(SSD_ERR_INVALID, "state must be 16-byte aligned");
    if (o->obs_env_stride < 0 || o->obs_slot_stride < 0 || (!o->obs_env_stride && o->obs_slot_stride))
        return fail(SSD_ERR_INVALID, "obs_env_stride / obs_slot_stride");
    if (o->obs_t_slots < 0) return fail(SSD_ERR_INVALID, "obs_t_slots");
    if (o->obs_code) {
        if (!o->obs || o->obs_format == SSD_OBS_CODE) return fail(SSD_ERR_INVALID, "obs_code accompanies a f32 / bf16 / u8 obs output");
        if (E->hs.obs_color != SSD_COLOR_SIMPLIFIED) return fail(SSD_ERR_INVALID, "obs_code needs simplified colours");
        if (((uintptr_t)o->obs_code & 15) != 0) return fail(SSD_ERR_INVALID, "obs_code must be 16-byte aligned");
    }
    d->obs = o->obs; d->fmt = o->obs_format; d->state = o->state; d->pos = o->pos; d->orient = o->orient;
    d->env_stride = (long)o->obs_env_stride; d->slot_stride = (long)o->obs_slot_stride;
    d->t_slots = o->obs_t_slots; d->code = o->obs_code; d->code_agent_stride = SSD_CODE_AGENT_STRIDE(E->hs.V);
    d->stamps = E->st.stamps;
    return SSD_OK;
}
}   // extern "C" (reopened below): the numeric-status word is a C++ helper of namespace ssd
namespace ssd {
int32_t* numeric_err_word() {
    static int32_t* word[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!word[dev]) {
        int32_t* w = nullptr;
        if (hipMalloc((void**)&w, 64) != hipSuccess || hipMemset(w, 0, 64) != hipSuccess) return nullptr;
        word[dev] = w;
    }
    return word[dev];
}
}  // namespace ssd
extern "C" {

static int launched(void) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SSD_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    return SSD_OK;
}

int ssd_reset(ssd_env* E, const uint8_t* env_mask, const ssd_tape* tape, ssd_step_out* out, void* stream) {
    if (!E) return fail(SSD_ERR_INVALID, "null env");
    DevTape t; if (int rc = make_tape(E, tape, &t)) return rc;
    if (E->hs.rng_mode == SSD_RNG_TAPE && E->hs.random_spawn && !t.spawn_order) return fail(SSD_ERR_INVALID, "TAPE mode with random_spawn_point needs tape.spawn_order");
    DevObsOut oo; std::memset(&oo, 0, sizeof oo);
    launch_env(MODE_RESET, E->dspec, E->hs, E->st, nullptr, env_mask, t, make_so(out), oo, (hipStream_t)stream);
    return launched();
}

int ssd_step(ssd_env* E, const int32_t* actions, const ssd_tape* tape, ssd_step_out* out, void* stream) {
    if (!E || !actions) return fail(SSD_ERR_INVALID, "null argument");
    DevTape t; if (int rc = make_tape(E, tape, &t)) return rc;
    DevObsOut oo; std::memset(&oo, 0, sizeof oo);
    launch_env(MODE_STEP, E->dspec, E->hs, E->st, actions, nullptr, t, make_so(out), oo, (hipStream_t)stream);
    return launched();
}

int ssd_observe(ssd_env* E, ssd_obs_out* out, void* stream) {
    if (!E) return fail(SSD_ERR_INVALID, "null env");
    DevObsOut oo; if (int rc = make_oo(E, out, &oo)) return rc;
    DevTape t; std::memset(&t, 0, sizeof t);
    launch_env(MODE_OBS, E->dspec, E->hs, E->st, nullptr, nullptr, t, make_so(nullptr), oo, (hipStream_t)stream);
    return launched();
}

int ssd_step_observe(ssd_env* E, const int32_t* actions, const ssd_tape* tape, ssd_step_out* out, ssd_obs_out* obs, void* stream) {
    if (!E || !actions) return fail(SSD_ERR_INVALID, "null argument");
    DevTape t; if (int rc = make_tape(E, tape, &t)) return rc;
    DevObsOut oo; if (int rc = make_oo(E, obs, &oo)) return rc;
    launch_env(MODE_STEP_OBS, E->dspec, E->hs, E->st, actions, nullptr, t, make_so(out), oo, (hipStream_t)stream);
    return launched();
}

#ifdef SSD_STAMPS
// diagnostic build only: device buffer [n_env, 32] of u64 receiving the phase stamps
int ssd_debug_set_stamps(ssd_env* E, unsigned long long* buf) { E->st.stamps = buf; return SSD_OK; }
int ssd_debug_set_policy_stamps(unsigned long long* buf) { ssd::set_policy_stamps(buf); return SSD_OK; }
#endif

int ssd_poll_error(ssd_env* E, int32_t* bits) {
    if (!E || !bits) return fail(SSD_ERR_INVALID, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(bits, E->st.err, 4, hipMemcpyDeviceToHost));
    if (*bits) HIP_TRY(hipMemset(E->st.err, 0, 4));
    int32_t num = 0;                                                   // + the device's numeric status (SSD_ERRBIT_F16_RANGE)
    if (int rc = ssd_numeric_status(&num)) return rc;
    *bits |= num;
    return SSD_OK;
}

int ssd_numeric_status(int32_t* bits) {
    if (!bits) return fail(SSD_ERR_INVALID, "null argument");
    int32_t* w = numeric_err_word();
    if (!w) return fail(SSD_ERR_DEVICE, "numeric status word: allocation failed");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(bits, w, 4, hipMemcpyDeviceToHost));
    if (*bits) HIP_TRY(hipMemset(w, 0, 4));
    return SSD_OK;
}

int ssd_export_state(ssd_env* E, ssd_state* dst, void* stream) {
    if (!E || !dst) return fail(SSD_ERR_INVALID, "null argument");
    launch_export(E->dspec, E->hs, E->st, *dst, (hipStream_t)stream);
    return launched();
}

int ssd_import_state(ssd_env* E, const ssd_state* src, void* stream) {
    if (!E || !src) return fail(SSD_ERR_INVALID, "null argument");
    launch_import(E->dspec, E->hs, E->st, *src, (hipStream_t)stream);
    return launched();
}

int ssd_build_inputs(int32_t batch, int32_t n_agents, int32_t n_actions, int32_t t0, const int64_t* last_actions,
                     const float* last_reward, const int64_t* last_actions_inc, const float* pos, float pos_scale,
                     float* out, int32_t out_stride, int32_t out_offset, void* stream) {
    if (batch < 1 || n_agents < 1 || n_actions < 1 || !out || !pos) return fail(SSD_ERR_INVALID, "bad argument");
    if (!(t0 & 1) && (!last_actions || !last_reward || !last_actions_inc)) return fail(SSD_ERR_INVALID, "t > 0 needs the t-1 tensors");
    if (out_stride < out_offset + n_actions + n_agents + 4) return fail(SSD_ERR_INVALID, "out_stride too small");
    launch_build_inputs(batch, n_agents, n_actions, t0, last_actions, last_reward, last_actions_inc, pos, pos_scale, out,
                        out_stride, out_offset, (hipStream_t)stream);
    return launched();
}

int ssd_unroll_other(const int64_t* actions, const float* pos, const float* orient, const float* reward, const float* clean_num, const float* apple_den,
                     float pos_scale, int32_t batch, int32_t T, int32_t n_agents, int32_t n_actions, float* other, float* act_tm, void* stream) {
    if (!actions || !pos || !orient || !reward || !clean_num || !apple_den || !other || !act_tm || batch < 1 || T < 1 || n_agents < 1 || n_actions < 1 ||
        !(pos_scale > 0.f)) return fail(SSD_ERR_INVALID, "bad argument");
    launch_unroll_other(actions, pos, orient, reward, clean_num, apple_den, pos_scale, batch, T, n_agents, n_actions, other, act_tm, (hipStream_t)stream);
    return launched();
}

int ssd_incentive_transfer(int32_t batch, int32_t T, int32_t n_agents, const int64_t* actions_inc, const float* rewards,
                           float effect_ratio, float cost_ratio, float incentive, float seq_len, float* give,
                           float* recv_pos, float* recv_neg, float* recv_zero, float* rewards_for_env,
                           float* rewards_for_inc, void* stream) {
    if (batch < 1 || T < 2 || n_agents < 1 || !actions_inc || !rewards || !give || !recv_pos || !recv_neg || !recv_zero ||
        !rewards_for_env || !rewards_for_inc)
        return fail(SSD_ERR_INVALID, "bad argument");
    launch_incentive_transfer(batch, T, n_agents, actions_inc, rewards, effect_ratio, cost_ratio, incentive, seq_len, give,
                              recv_pos, recv_neg, recv_zero, rewards_for_env, rewards_for_inc, (hipStream_t)stream);
    return launched();
}

int ssd_column_sums(const float* x, float* out, int32_t groups, int32_t rows, int32_t cols, float* workspace, void* stream) {
    if (!x || !out || groups < 1 || rows < 1 || cols < 1 || groups > 65535 || (rows + SSD_COLSUM_CHUNK - 1) / SSD_COLSUM_CHUNK > 65535)
        return fail(SSD_ERR_INVALID, "bad argument");
    launch_column_sums(x, out, groups, rows, cols, workspace, (hipStream_t)stream);
    return launched();
}

int ssd_build_inputs_width(int32_t n_agents, int32_t n_actions, uint32_t input_flags) {
    if (n_agents < 1 || n_actions < 1 || ((input_flags & ~SSD_INPUT_EXPLICIT) & ~127u)) return -1;
    return build_inputs_layout(n_agents, n_actions, input_flags).width;
}

int ssd_build_inputs_flags(int32_t batch, int32_t n_agents, int32_t n_actions, int32_t t0, uint32_t input_flags, const int64_t* last_actions,
                           const float* last_reward, const int64_t* last_actions_inc, const float* pos, float pos_scale, float* out,
                           int32_t out_stride, int32_t out_offset, void* stream) {
    if (batch < 1 || n_agents < 1 || n_actions < 1 || !out || !pos) return fail(SSD_ERR_INVALID, "bad argument");
    if ((input_flags & ~SSD_INPUT_EXPLICIT) & ~127u) return fail(SSD_ERR_UNSUPPORTED, "ssd_build_inputs_flags: unknown input_flags bit");
    if (!(t0 & 1) && (!last_actions || !last_reward || !last_actions_inc)) return fail(SSD_ERR_INVALID, "t > 0 needs the t-1 tensors");
    if (out_stride < out_offset + build_inputs_layout(n_agents, n_actions, input_flags).width) return fail(SSD_ERR_INVALID, "out_stride too small");
    launch_build_inputs_flags(batch, n_agents, n_actions, t0, input_flags, last_actions, last_reward, last_actions_inc, pos, pos_scale, out,
                              out_stride, out_offset, (hipStream_t)stream);
    return launched();
}

int ssd_clip_adam_step(const ssd_clip_adam_args* a, void* stream) {
    if (!a || !a->flat_grad || !a->jobs || !a->partials || a->total < 1 || a->n_jobs < 1 || a->n_jobs > SSD_ADAM_MAX_JOBS)
        return fail(SSD_ERR_INVALID, "ssd_clip_adam_step: bad argument");
    if (a->total > (int64_t)INT32_MAX) return fail(SSD_ERR_INVALID, "ssd_clip_adam_step: total");
    if (!(a->beta1 >= 0.f && a->beta1 < 1.f) || !(a->beta2 >= 0.f && a->beta2 < 1.f) || !(a->eps > 0.f) || !(a->clip > 0.f))
        return fail(SSD_ERR_INVALID, "ssd_clip_adam_step: betas / eps / clip");
    launch_clip_adam(a, (hipStream_t)stream);
    return launched();
}

int ssd_dueling_q_fwd(const float* a, const float* v, float* q, int32_t n, int32_t T, int32_t B, int32_t inner, int32_t K, void* stream) {
    if (!a || !v || !q || n < 1 || T < 1 || B < 1 || inner < 1 || K < 1 || K > 16) return fail(SSD_ERR_INVALID, "ssd_dueling_q_fwd: bad argument");
    launch_dueling_q(a, v, q, nullptr, nullptr, nullptr, n, T, B, inner, K, (hipStream_t)stream);
    return launched();
}
int ssd_dueling_q_bwd(const float* dq, float* da, float* dv, int32_t n, int32_t T, int32_t B, int32_t inner, int32_t K, void* stream) {
    if (!dq || !da || !dv || n < 1 || T < 1 || B < 1 || inner < 1 || K < 1 || K > 16) return fail(SSD_ERR_INVALID, "ssd_dueling_q_bwd: bad argument");
    launch_dueling_q(nullptr, nullptr, nullptr, dq, da, dv, n, T, B, inner, K, (hipStream_t)stream);
    return launched();
}

int ssd_gather_rows(const ssd_row_gather* fields, int32_t count, const int64_t* ids, int32_t n_ids, void* stream) {
    if (!fields || count < 1 || count > SSD_COPY_BLOCKS_MAX || !ids || n_ids < 1 || n_ids > 65535) return fail(SSD_ERR_INVALID, "ssd_gather_rows: bad argument");
    for (int i = 0; i < count; ++i)
        if (!fields[i].src || !fields[i].dst || fields[i].row_bytes < 1) return fail(SSD_ERR_INVALID, "ssd_gather_rows: bad field");
    launch_gather_rows(fields, count, ids, n_ids, (hipStream_t)stream);
    return launched();
}

int ssd_sample_ids(uint64_t seed, uint32_t call, int32_t population, int32_t count, int64_t* ids, void* stream) {
    if (!ids || count < 1 || count > SSD_SAMPLE_IDS_MAX || population < count) return fail(SSD_ERR_INVALID, "ssd_sample_ids: 1 <= count <= min(population, SSD_SAMPLE_IDS_MAX)");
    launch_sample_ids(seed, call, population, count, ids, (hipStream_t)stream);
    return launched();
}

int ssd_copy_blocks(const ssd_block_copy* blocks, int32_t count, void* stream) {
    if (!blocks || count < 1 || count > SSD_COPY_BLOCKS_MAX) return fail(SSD_ERR_INVALID, "ssd_copy_blocks: 1..SSD_COPY_BLOCKS_MAX blocks");
    for (int i = 0; i < count; ++i) {
        const ssd_block_copy& b = blocks[i];
        if (!b.src || !b.dst || b.rows < 1 || b.cols < 1 || b.src_stride < b.cols || b.dst_stride < b.cols ||
            (int64_t)b.rows * b.cols > INT32_MAX)
            return fail(SSD_ERR_INVALID, "ssd_copy_blocks: bad block");
    }
    launch_copy_blocks(blocks, count, (hipStream_t)stream);
    return launched();
}

int ssd_fill_blocks(const ssd_block_fill* blocks, int32_t count, void* stream) {
    if (!blocks || count < 1 || count > SSD_FILL_BLOCKS_MAX) return fail(SSD_ERR_INVALID, "ssd_fill_blocks: 1..SSD_FILL_BLOCKS_MAX blocks");
    for (int i = 0; i < count; ++i)
        if (!blocks[i].dst || blocks[i].bytes < 4 || (blocks[i].bytes & 3) || ((uintptr_t)blocks[i].dst & 3)) return fail(SSD_ERR_INVALID, "ssd_fill_blocks: a block is a 4-byte aligned multiple of 4 bytes");
    launch_fill_blocks(blocks, count, (hipStream_t)stream);
    return launched();
}
int ssd_runner_stats(const float* collective_return, const float* equality, const float* episode_return, int32_t n_env, int32_t n_returns,
                     double* acc, void* stream) {
    if (!collective_return || !equality || !episode_return || !acc || n_env < 1 || n_returns < 1) return fail(SSD_ERR_INVALID, "bad argument");
    launch_runner_stats(collective_return, equality, episode_return, n_env, n_returns, acc, (hipStream_t)stream);
    return launched();
}

int ssd_td_sim_loss(const ssd_td_loss_args* a, int32_t mode, void* stream) {
    if (!a || a->batch < 1 || a->t_slots < 2 || a->n_agents < 2 || a->n_agents > SSD_MAX_AGENTS || a->n_actions < 1 || a->sim_horizon < 1)
        return fail(SSD_ERR_INVALID, "bad argument");
    if (!a->reward || !a->clean_num || !a->terminated || !a->filled || !a->partials) return fail(SSD_ERR_INVALID, "null argument");
    if (mode && (!a->q_env || !a->q_inc || !a->tq_env || !a->tq_inc || !a->actions || !a->actions_inc || !a->avail || !a->dens || !a->dq_env || !a->dq_inc))
        return fail(SSD_ERR_INVALID, "null argument");
    if (!(a->seq_len > 0.f) || !(a->reward_scale != 0.f)) return fail(SSD_ERR_INVALID, "seq_len / reward_scale");
    launch_td_sim_loss(a, mode ? 1 : 0, (hipStream_t)stream);
    return launched();
}

int ssd_encoder(const float* obs, int32_t rows, int32_t view_edge, int32_t conv_out, int32_t feat_out, const float* conv_w,
                const float* conv_b, const float* lin_w, const float* lin_b, float* out, int32_t out_stride, int32_t n_agents,
                int32_t agent_major, float* store_obs, int64_t store_env_stride, const int64_t* store_t, void* stream) {
    if (!obs || !conv_w || !conv_b || !lin_w || !lin_b || !out || rows < 1 || view_edge < 3) return fail(SSD_ERR_INVALID, "bad argument");
    if (conv_out != 6 || feat_out != 32) return fail(SSD_ERR_UNSUPPORTED, "ssd_encoder is instantiated for conv_out 6, obs_dim_net 32 (config/default.yaml:59-63)");
    if ((agent_major || store_obs) && (n_agents < 1 || rows % n_agents)) return fail(SSD_ERR_INVALID, "rows must be a multiple of n_agents");
    if (store_obs && !store_t) return fail(SSD_ERR_INVALID, "store_obs needs store_t");
    if (4 * 4 * (3 * view_edge * view_edge + 4) * 4 > 160 * 1024) return fail(SSD_ERR_INVALID, "view too large for the encoder's LDS staging");
    launch_encoder(obs, rows, view_edge, conv_w, conv_b, lin_w, lin_b, out, out_stride, n_agents, agent_major, store_obs,
                   (long)store_env_stride, store_t, (hipStream_t)stream);
    return launched();
}

int ssd_conv_leaky(const float* obs, int32_t rows, int32_t view_edge, int32_t conv_out, const float* conv_w, const float* conv_b,
                   float* out, int32_t n_agents, int32_t agent_major, float* store_obs, int64_t store_env_stride,
                   const int64_t* store_t, void* stream) {
    if (!obs || !conv_w || !conv_b || !out || rows < 1 || view_edge < 3 || n_agents < 1 || rows % n_agents) return fail(SSD_ERR_INVALID, "bad argument");
    if (conv_out != 6) return fail(SSD_ERR_UNSUPPORTED, "ssd_conv_leaky is instantiated for conv_out 6 (config/default.yaml:59)");
    if (store_obs && !store_t) return fail(SSD_ERR_INVALID, "store_obs needs store_t");
    launch_conv_leaky(obs, rows, view_edge, conv_w, conv_b, out, n_agents, agent_major, store_obs, (long)store_env_stride, store_t,
                      (hipStream_t)stream);
    return launched();
}

int ssd_store_step_launch(const ssd_store_step* a, void* stream) {
    if (!a || !a->t_index || a->n_env < 1 || a->n_agents < 1 || a->t_slots < 1) return fail(SSD_ERR_INVALID, "bad argument");
    if (a->next_t_out && a->next_t_out == a->t_index) return fail(SSD_ERR_INVALID, "next_t_out must not alias t_index");
    if ((a->prev_reward || a->ep_return) && !a->reward) return fail(SSD_ERR_INVALID, "prev_reward / ep_return need reward");
    launch_store_step(a, (hipStream_t)stream);
    return launched();
}

int ssd_gru_gates(const float* gi, const float* gh, float* h, int32_t rows, int32_t hidden, void* stream) {
    if (!gi || !gh || !h || rows < 1 || hidden < 1) return fail(SSD_ERR_INVALID, "bad argument");
    launch_gru_gates(gi, gh, h, rows, hidden, (hipStream_t)stream);
    return launched();
}

int ssd_gru_gates_fwd(const float* gi, const float* gh, const float* h, float* h_new, float* rzn, int32_t rows, int32_t hidden, void* stream) {
    if (!gi || !gh || !h || !h_new || !rzn || rows < 1 || hidden < 1) return fail(SSD_ERR_INVALID, "bad argument");
    launch_gru_fwd_train(gi, gh, h, h_new, rzn, rows, hidden, (hipStream_t)stream);
    return launched();
}

int ssd_gru_gates_bwd(const float* dh_new, const float* rzn, const float* gh, const float* h, float* d_gi, float* d_gh, float* dh_prev,
                      int32_t rows, int32_t hidden, void* stream) {
    if (!dh_new || !rzn || !gh || !h || !d_gi || !d_gh || !dh_prev || rows < 1 || hidden < 1) return fail(SSD_ERR_INVALID, "bad argument");
    launch_gru_bwd(dh_new, rzn, gh, h, d_gi, d_gh, dh_prev, rows, hidden, (hipStream_t)stream);
    return launched();
}

int ssd_dueling_pick(const float* av, int32_t rows, int32_t n_actions, const uint8_t* avail, const float* epsilon, const int64_t* step,
                     uint32_t seed, int32_t n_agents, int32_t batch, int32_t pairs, int64_t* actions, float* q_out, uint32_t env_id_base,
                     void* stream) {
    if (!av || !epsilon || !step || !actions || rows < 1 || n_actions < 1 || n_agents < 1 || batch < 1) return fail(SSD_ERR_INVALID, "bad argument");
    if (rows != (pairs ? n_agents * batch * n_agents : n_agents * batch)) return fail(SSD_ERR_INVALID, "rows must be n*B (or n*B*n for pairs)");
    launch_dueling_pick(av, rows, n_actions, avail, epsilon, step, seed, n_agents, batch, pairs, actions, q_out, env_id_base, (hipStream_t)stream);
    return launched();
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int ssd_gru_seq_fwd(const float* gi, const float* wh, const float* bh, float* hs, float* rzn, float* ghn, int32_t T, int32_t G, int32_t B,
                    void* stream) {
    if (!gi || !wh || !bh || !hs || T < 1 || G < 1 || B < 1 || (!rzn) != (!ghn)) return fail(SSD_ERR_INVALID, "bad argument");
    if (B % 16) return fail(SSD_ERR_INVALID, "B must be a multiple of 16 (pad the sequences)");
    if (!al16(gi) || !al16(wh) || !al16(bh) || !al16(hs) || !al16(rzn) || !al16(ghn)) return fail(SSD_ERR_INVALID, "tensors must be 16-byte aligned");
    launch_gru_seq_fwd(&gi, 0, &wh, &bh, 1, hs, rzn, ghn, T, G, B, (hipStream_t)stream);
    return launched();
}
int ssd_gru_seq_fwd_parts(const float* const* gi_parts, int32_t n_parts, const float* const* wh_parts, const float* const* bh_parts, int32_t n_wparts,
                          float* hs, float* rzn, float* ghn, int32_t T, int32_t G, int32_t B, void* stream) {
    if (!gi_parts || n_parts < 1 || n_parts > 4 || !wh_parts || !bh_parts || n_wparts < 1 || n_wparts > 4 || !hs || T < 1 || G < 1 || G % n_parts ||
        G % n_wparts || B < 1 || (!rzn) != (!ghn))
        return fail(SSD_ERR_INVALID, "bad argument");
    for (int k = 0; k < n_wparts; ++k)
        if (!wh_parts[k] || !bh_parts[k] || !al16(wh_parts[k]) || !al16(bh_parts[k])) return fail(SSD_ERR_INVALID, "weight parts must be non-null and 16-byte aligned");
    if (B % 16) return fail(SSD_ERR_INVALID, "B must be a multiple of 16 (pad the sequences)");
    for (int k = 0; k < n_parts; ++k)
        if (!gi_parts[k] || !al16(gi_parts[k])) return fail(SSD_ERR_INVALID, "gi parts must be non-null and 16-byte aligned");
    if (!al16(hs) || !al16(rzn) || !al16(ghn)) return fail(SSD_ERR_INVALID, "tensors must be 16-byte aligned");
    launch_gru_seq_fwd(gi_parts, n_parts, wh_parts, bh_parts, n_wparts, hs, rzn, ghn, T, G, B, (hipStream_t)stream);
    return launched();
}
int ssd_gru_seq_bwd(const float* dhs, const float* hs, const float* rzn, const float* ghn, const float* wh, float* d_gi, float* dgh,
                    float* d_wh, float* d_bh_part, int32_t T, int32_t G, int32_t B, void* stream) {
    if (!dhs || !hs || !rzn || !ghn || !wh || !d_gi || !dgh || !d_wh || !d_bh_part || T < 1 || G < 1 || B < 1) return fail(SSD_ERR_INVALID, "bad argument");
    if (B % 16) return fail(SSD_ERR_INVALID, "B must be a multiple of 16 (pad the sequences)");
    if (!al16(dhs) || !al16(hs) || !al16(rzn) || !al16(ghn) || !al16(wh) || !al16(d_gi) || !al16(dgh)) return fail(SSD_ERR_INVALID, "tensors must be 16-byte aligned");
    launch_gru_seq_bwd(&dhs, hs, rzn, ghn, &wh, 1, &d_gi, 0, dgh, d_wh, d_bh_part, T, G, G, B, (hipStream_t)stream);
    return launched();
}
int ssd_gru_seq_bwd_parts(const float* const* dhs_parts, const float* hs, const float* rzn, const float* ghn, const float* const* wh_parts, int32_t n_wparts,
                          float* const* d_gi_parts, int32_t n_parts, float* dgh, float* d_wh, float* d_bh_part, int32_t T, int32_t G, int32_t G_grad, int32_t B,
                          void* stream) {
    if (!dhs_parts || !hs || !rzn || !ghn || !wh_parts || n_wparts < 1 || n_wparts > 4 || !d_gi_parts || n_parts < 1 || n_parts > 4 || !dgh || !d_wh ||
        !d_bh_part || T < 1 || G < 1 || G % n_parts || G % n_wparts || B < 1 || G_grad < 1 || G_grad > G || G_grad % (G / n_parts))
        return fail(SSD_ERR_INVALID, "bad argument");
    for (int k = 0; k * (G / n_parts) < G_grad; ++k)
        if (!dhs_parts[k] || !al16(dhs_parts[k])) return fail(SSD_ERR_INVALID, "dhs parts of the sets with a gradient must be non-null and 16-byte aligned");
    for (int k = 0; k < n_wparts; ++k)
        if (!wh_parts[k] || !al16(wh_parts[k])) return fail(SSD_ERR_INVALID, "weight parts must be non-null and 16-byte aligned");
    if (B % 16) return fail(SSD_ERR_INVALID, "B must be a multiple of 16 (pad the sequences)");
    for (int k = 0; k * (G / n_parts) < G_grad; ++k)
        if (!d_gi_parts[k] || !al16(d_gi_parts[k])) return fail(SSD_ERR_INVALID, "d_gi parts of the sets with a gradient must be non-null and 16-byte aligned");
    if (!al16(hs) || !al16(rzn) || !al16(ghn) || !al16(dgh)) return fail(SSD_ERR_INVALID, "tensors must be 16-byte aligned");
    launch_gru_seq_bwd(dhs_parts, hs, rzn, ghn, wh_parts, n_wparts, d_gi_parts, n_parts, dgh, d_wh, d_bh_part, T, G, G_grad, B, (hipStream_t)stream);
    return launched();
}

int ssd_bias_bmm_fwd(const float* x, const float* w, const float* b, float* y, int32_t n, int32_t rows, int32_t in, int32_t out, void* stream) {
    if (!x || !w || !b || !y || n < 1 || rows < 1 || in < 1 || out < 1) return fail(SSD_ERR_INVALID, "bad argument");
    if (launch_bias_bmm_fwd(x, w, b, y, n, rows, in, out, (hipStream_t)stream)) return fail(SSD_ERR_UNSUPPORTED, "ssd_bias_bmm_fwd: a weight / operand set of 2^30 elements or more");
    return launched();
}
int ssd_bias_bmm_leaky_fwd(const float* x, const float* w, const float* b, float* y, int32_t n, int32_t rows, int32_t in, int32_t out, void* stream) {
    if (!x || !w || !b || !y || n < 1 || rows < 1 || in < 1 || out < 1) return fail(SSD_ERR_INVALID, "bad argument");
    if (launch_bias_bmm_fwd(x, w, b, y, n, rows, in, out, (hipStream_t)stream, 1)) return fail(SSD_ERR_UNSUPPORTED, "ssd_bias_bmm_leaky_fwd: a weight / operand set of 2^30 elements or more");
    return launched();
}
int ssd_bias_bmm_leaky_bwd(const float* g, const float* y, const float* x, const float* w, float* dx, float* dw, float* db, const float* slope_of,
                           int32_t n, int32_t rows, int32_t in, int32_t out, void* stream) {
    if (!g || !y || n < 1 || rows < 1 || in < 1 || out < 1) return fail(SSD_ERR_INVALID, "bad argument");
    if ((dx && !w) || ((dw || db) && !x) || (slope_of && !dx)) return fail(SSD_ERR_INVALID, "dx needs w, dw / db need x, slope_of needs dx");
    if (launch_bias_bmm_bwd(g, x, w, dx, dw, db, slope_of, n, rows, in, out, (hipStream_t)stream, 0, 0, y)) return fail(SSD_ERR_UNSUPPORTED, "ssd_bias_bmm_leaky_bwd: an operand set of 2^30 elements or more");
    return launched();
}
int ssd_bias_bmm_bwd(const float* g, const float* x, const float* w, float* dx, float* dw, float* db, const float* slope_of, int32_t n,
                     int32_t rows, int32_t in, int32_t out, void* stream) {
    if (!g || n < 1 || rows < 1 || in < 1 || out < 1) return fail(SSD_ERR_INVALID, "bad argument");
    if ((dx && !w) || ((dw || db) && !x) || (slope_of && !dx)) return fail(SSD_ERR_INVALID, "dx needs w, dw / db need x, slope_of needs dx");
    if (launch_bias_bmm_bwd(g, x, w, dx, dw, db, slope_of, n, rows, in, out, (hipStream_t)stream)) return fail(SSD_ERR_UNSUPPORTED, "ssd_bias_bmm_bwd: an operand set of 2^30 elements or more");
    return launched();
}

int ssd_dueling_head_fwd(const float* y, float* q, int32_t n, int32_t T, int32_t B, int32_t inner, int32_t K, void* stream) {
    if (!y || !q || n < 1 || T < 1 || B < 1 || inner < 1 || K < 1 || K > 15) return fail(SSD_ERR_INVALID, "bad argument");
    launch_dueling_q(y, y + K, q, nullptr, nullptr, nullptr, n, T, B, inner, K, (hipStream_t)stream, K + 1, 1, nullptr);
    return launched();
}
int ssd_dueling_head_bwd(const float* dq, float* dy, float* gs, int32_t n, int32_t T, int32_t B, int32_t inner, int32_t K, void* stream) {
    if (!dq || !dy || n < 1 || T < 1 || B < 1 || inner < 1 || K < 1 || K > 15) return fail(SSD_ERR_INVALID, "bad argument");
    launch_dueling_q(nullptr, nullptr, nullptr, dq, dy, dy + K, n, T, B, inner, K, (hipStream_t)stream, K + 1, 1, gs);
    return launched();
}
int ssd_bias_bmm2_fwd(const float* x1, const float* x2, const float* w, const float* b, float* y, int32_t n, int32_t rows, int32_t in1, int32_t in2,
                      int32_t out, int32_t x1_div, int32_t x2_shared, void* stream) {
    if (!x1 || !x2 || !w || !b || !y || n < 1 || rows < 1 || in1 < 1 || in2 < 1 || out < 1) return fail(SSD_ERR_INVALID, "bad argument");
    const int rc = launch_bias_bmm2_fwd(x1, x2, w, b, y, n, rows, in1, in2, out, x1_div, x2_shared, (hipStream_t)stream);
    if (rc == -3) return fail(SSD_ERR_INVALID, "ssd_bias_bmm2_fwd: in1 a multiple of 16, rows a multiple of x1_div >= 1");
    if (rc) return fail(SSD_ERR_UNSUPPORTED, "ssd_bias_bmm2_fwd: a weight / operand set of 2^30 elements or more");
    return launched();
}
int ssd_bias_bmm2_bwd_w(const float* g, const float* x1, const float* x2, float* dw, float* db, int32_t n, int32_t rows, int32_t in1, int32_t in2,
                        int32_t out, int32_t x1_div, int32_t x2_shared, void* stream) {
    if (!g || !x1 || !x2 || (!dw && !db) || n < 1 || rows < 1 || in1 < 1 || in2 < 1 || out < 1) return fail(SSD_ERR_INVALID, "bad argument");
    const int rc = launch_bias_bmm_bwd(g, x1, nullptr, nullptr, dw, db, nullptr, n, rows, in1 + in2, out, (hipStream_t)stream, 0, 0, nullptr, x2, in1, x1_div,
                                       x2_shared, 0);
    if (rc == -3) return fail(SSD_ERR_INVALID, "ssd_bias_bmm2_bwd_w: in1 a multiple of 16, rows a multiple of x1_div >= 1");
    if (rc) return fail(SSD_ERR_UNSUPPORTED, "ssd_bias_bmm2_bwd_w: an operand set of 2^30 elements or more");
    return launched();
}
int ssd_bias_bmm_bwd_x(const float* g, const float* w, float* dx, int32_t n, int32_t rows, int32_t in, int32_t out, int64_t w_set, void* stream) {
    if (!g || !w || !dx || n < 1 || rows < 1 || in < 1 || out < 1 || w_set < (int64_t)in * out) return fail(SSD_ERR_INVALID, "bad argument");
    if (launch_bias_bmm_bwd(g, nullptr, w, dx, nullptr, nullptr, nullptr, n, rows, in, out, (hipStream_t)stream, 0, 0, nullptr, nullptr, 0, 1, 0, (long)w_set))
        return fail(SSD_ERR_UNSUPPORTED, "ssd_bias_bmm_bwd_x: an operand set of 2^30 elements or more");
    return launched();
}
int ssd_bmm_reserve_scratch(void* stream) {
    if (!bmm_scratch((hipStream_t)stream)) return fail(SSD_ERR_DEVICE, "ssd_bmm_reserve_scratch: allocation failed (inside a stream capture?)");
    return SSD_OK;
}
int ssd_set_learner_precision(int32_t precision) {
    if (precision != 1 && precision != 2) return fail(SSD_ERR_INVALID, "ssd_set_learner_precision: 1 (bf16) or 2 (f32)");
    set_learner_precision(precision);
    return SSD_OK;
}
int ssd_learner_precision(void) { return learner_precision(); }
int ssd_conv_wgrad_partial_rows(int32_t rows) { return rows < 1 ? 0 : conv_wgrad_partial_rows(rows); }
int ssd_conv_wgrad_codes(const uint8_t* codes, const float* d_conv, float* partial, int32_t rows, int32_t view_edge, void* stream) {
    if (!codes || !d_conv || !partial || rows < 1) return fail(SSD_ERR_INVALID, "bad argument");
    if (launch_conv_wgrad(codes, d_conv, partial, rows, view_edge, (hipStream_t)stream)) return fail(SSD_ERR_UNSUPPORTED, "ssd_conv_wgrad_codes is instantiated for 15 x 15 and 31 x 31 windows");
    return launched();
}

static int check_encode(const ssd_policy_encode_args* a) {
    if (!a || !a->codes || !a->conv_frags || !a->lin_frags || !a->conv_b || !a->lin_b || a->rows < 1 || a->n_agents < 1 || a->rows % a->n_agents)
        return fail(SSD_ERR_INVALID, "bad argument");
    if (a->view_edge != 15 && a->view_edge != 31)
        return fail(SSD_ERR_UNSUPPORTED, "ssd_policy_encode is instantiated for view_size 7 and 15 (15 x 15 / 31 x 31 windows); use ssd_conv_leaky + GEMM");
    if (a->precision != 0 && a->precision != 1 && a->precision != 2) return fail(SSD_ERR_INVALID, "precision must be 1 (bf16) or 2 (f32-equivalent)");
    if (a->alphabet != SSD_CODE_CLASS && a->alphabet != SSD_CODE_CHANNEL_MASK) return fail(SSD_ERR_INVALID, "alphabet");
    if (a->layout != SSD_ENCODE_LAYOUT_TOEPLITZ && a->layout != SSD_ENCODE_LAYOUT_LUT) return fail(SSD_ERR_INVALID, "layout");
    if (a->act && a->layout != SSD_ENCODE_LAYOUT_TOEPLITZ) return fail(SSD_ERR_INVALID, "the activation output (training forward) takes the Toeplitz images");
    const int bands = SSD_ENCODE_BANDS(a->view_edge);
    if (bands == 1 ? (!a->out || a->part || a->out_stride < 32) : (!a->part || a->out)) return fail(SSD_ERR_INVALID, "one band writes `out`, several bands write `part`");
    if ((reinterpret_cast<uintptr_t>(a->conv_frags) | reinterpret_cast<uintptr_t>(a->lin_frags) | reinterpret_cast<uintptr_t>(a->part)) & 15)
        return fail(SSD_ERR_INVALID, "conv_frags / lin_frags / part must be 16-byte aligned");
    const long VV = (long)a->view_edge * a->view_edge;
    if (a->agent_stride < VV || a->env_stride < (long)a->n_agents * a->agent_stride || a->slot_stride < 0 || (a->slot_stride && !a->slot_t))
        return fail(SSD_ERR_INVALID, "env_stride / slot_stride / agent_stride / slot_t");
    if (a->code_bytes < (long)(a->rows / a->n_agents - 1) * a->env_stride + (long)(a->n_agents - 1) * a->agent_stride + VV)
        return fail(SSD_ERR_INVALID, "code_bytes does not cover the rows");
    if (a->slot_t_copy && (!a->slot_t || a->slot_t_copy == a->slot_t)) return fail(SSD_ERR_INVALID, "slot_t_copy needs a distinct slot_t");
    if (a->slot_add < 0 || (a->slot_add && !a->slot_t)) return fail(SSD_ERR_INVALID, "slot_add needs slot_t");
    return SSD_OK;
}

int ssd_policy_encode(const ssd_policy_encode_args* a, void* stream) {
    if (const int bad = check_encode(a)) return bad;
    const int rc = launch_policy_encode(a, (hipStream_t)stream);
    if (rc) return fail(SSD_ERR_DEVICE, "hipFuncSetAttribute(max dynamic LDS) failed");
    return launched();
}

int ssd_policy_pack_encoder_lut(const float* conv_w, const float* conv_b, const float* lin_w, int32_t view_edge, int32_t precision, void* table,
                                void* lin_frags, void* stream) {
    if (!conv_w || !conv_b || !lin_w || !table || !lin_frags) return fail(SSD_ERR_INVALID, "null argument");
    if (precision != 1 && precision != 2) return fail(SSD_ERR_INVALID, "precision must be 1 or 2");
    if ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(lin_frags)) & 15) return fail(SSD_ERR_INVALID, "images must be 16-byte aligned");
    if (launch_pack_encoder_lut(conv_w, conv_b, lin_w, view_edge, precision, table, lin_frags, (hipStream_t)stream))
        return fail(SSD_ERR_UNSUPPORTED, "ssd_policy_pack_encoder_lut: view_edge must be 15 or 31");
    return launched();
}

int ssd_policy_pack_encoder(const float* conv_w, const float* conv_b, const float* lin_w, int32_t view_edge, int32_t precision, void* conv_frags,
                            void* lin_frags, void* stream) {
    if (!conv_w || !conv_b || !lin_w || !conv_frags || !lin_frags) return fail(SSD_ERR_INVALID, "null argument");
    if (precision != 1 && precision != 2) return fail(SSD_ERR_INVALID, "precision must be 1 or 2");
    if ((reinterpret_cast<uintptr_t>(conv_frags) | reinterpret_cast<uintptr_t>(lin_frags)) & 15) return fail(SSD_ERR_INVALID, "fragment images must be 16-byte aligned");
    if (launch_pack_encoder(conv_w, conv_b, lin_w, view_edge, precision, conv_frags, lin_frags, (hipStream_t)stream))
        return fail(SSD_ERR_UNSUPPORTED, "ssd_policy_pack_encoder: view_edge must be 15 or 31");
    return launched();
}

int ssd_policy_pack_head(const ssd_policy_head_params* p, int32_t precision, void* image, void* stream) {
    if (!p || !image || !p->fc1_w || !p->fc1_b || !p->fc2_w || !p->fc2_b || !p->fc2_v_w || !p->fc2_v_b) return fail(SSD_ERR_INVALID, "null argument");
    for (int g = 0; g < 3; ++g) if (!p->w_i[g] || !p->w_h[g] || !p->b_i[g] || !p->b_h[g]) return fail(SSD_ERR_INVALID, "null GRU parameter");
    if (precision != 1 && precision != 2) return fail(SSD_ERR_INVALID, "precision must be 1 or 2");
    if (p->n_agents < 1 || p->fc1_in < 1 || p->fc1_in > 64 || p->fc2_in < 64 || p->fc2_in > 80 || p->fc2_out < 1 || p->fc2_out > 15)
        return fail(SSD_ERR_UNSUPPORTED, "ssd_policy_pack_head: fc1_in <= 64, 64 <= fc2_in <= 80, fc2_out <= 15");
    if (reinterpret_cast<uintptr_t>(image) & 15) return fail(SSD_ERR_INVALID, "image must be 16-byte aligned");
    launch_pack_head(p, precision, image, (hipStream_t)stream);
    return launched();
}

static int check_head(const ssd_policy_head* a, int inc) {
    if (!a || !a->inputs || !a->h || !a->weights || !a->epsilon || !a->step || !a->out_actions) return fail(SSD_ERR_INVALID, "bad argument");
    if (a->n_env < 1 || a->n_agents < 1 || a->n_actions < 1) return fail(SSD_ERR_INVALID, "bad argument");
    if (a->precision != 0 && a->precision != 1 && a->precision != 2) return fail(SSD_ERR_INVALID, "precision must be 1 (bf16) or 2 (f32-equivalent)");
    if (a->feat_part && (inc || a->feat_bands < 1 || !a->lin_b || (reinterpret_cast<uintptr_t>(a->feat_part) & 15)))
        return fail(SSD_ERR_INVALID, "feat_part (env head only) needs feat_bands >= 1, lin_b and 16-byte alignment");
    if (!inc && (a->pos_copy || a->dst_pos) && (!a->orient || (a->pos_copy && !a->orient_copy) || (a->dst_pos && !a->dst_orient)))
        return fail(SSD_ERR_INVALID, "pos_copy / dst_pos need orient and their orient twin");
    const bool files = a->dst_pos || a->dst_actions || a->dst_actions_inc || a->dst_reward || a->next_t_out;
    if (files && (!a->t_index || a->t_slots < 1)) return fail(SSD_ERR_INVALID, "filing into the episode storage needs t_index and t_slots");
    if (a->next_t_out && a->next_t_out == a->t_index) return fail(SSD_ERR_INVALID, "next_t_out must not alias t_index");
    if (a->dst_actions && !a->dst_actions_onehot) return fail(SSD_ERR_INVALID, "dst_actions needs dst_actions_onehot");
    if (a->dst_reward && (!a->dst_clean_num || !a->dst_apple_den)) return fail(SSD_ERR_INVALID, "dst_reward needs dst_clean_num and dst_apple_den");
    if (a->dst_terminated && !a->terminated) return fail(SSD_ERR_INVALID, "dst_terminated needs terminated");
    if (inc ? (!a->actions || !a->pos_pre || !a->orient_pre || !a->reward || !a->clean_num || !a->apple_den)
            : (!a->prev_actions || !a->prev_reward || (!a->prev_actions_inc && !a->recv_inc) || !a->pos)) return fail(SSD_ERR_INVALID, "missing head input");
    if ((a->recv_inc || a->recv_inc_out) && (a->n_agents > 16 || ((reinterpret_cast<uintptr_t>(a->recv_inc) | reinterpret_cast<uintptr_t>(a->recv_inc_out)) & 15)))
        return fail(SSD_ERR_INVALID, "recv_inc / recv_inc_out: 16-byte records (n_agents <= 16), 16-byte aligned");
    if (inc ? (a->recv_inc != nullptr) : (a->recv_inc_out != nullptr)) return fail(SSD_ERR_INVALID, "recv_inc is the env head's input, recv_inc_out the inc head's output");
    // layout limits of the fused kernel: 32 encoder features + tail (+ one-hot action for inc) within 64 columns, 16 fc2 rows
    {
        const uint32_t fl = a->input_flags ? (a->input_flags & ~SSD_INPUT_EXPLICIT) : (uint32_t)SSD_INPUT_FLAGS_SHIPPED;
        if (fl & ~63u) return fail(SSD_ERR_UNSUPPORTED, "ssd_policy_head: input_flags holds a block the fused head does not build");
        const int tail = ((fl & SSD_INPUT_LAST_ACTION) ? a->n_actions : 0) + ((fl & SSD_INPUT_AGENT_ID) ? a->n_agents : 0) +
                         ((fl & SSD_INPUT_REWARD) ? 1 : 0) + ((fl & SSD_INPUT_INC_REWARD) ? 1 : 0) +
                         ((fl & SSD_INPUT_DISTANCE) ? a->n_agents : 0) + ((fl & SSD_INPUT_AGENT_POS) ? 2 : 0);
        if (a->input_shape != 32 + tail || a->input_shape + a->n_actions > 64 || a->n_actions + 7 > 16)
            return fail(SSD_ERR_UNSUPPORTED, "ssd_policy_head: input_shape must be 32 + the width of the input_flags blocks and fit 64 columns");
    }
    if ((reinterpret_cast<uintptr_t>(a->inputs) | reinterpret_cast<uintptr_t>(a->h) | reinterpret_cast<uintptr_t>(a->weights)) & 15)
        return fail(SSD_ERR_INVALID, "inputs / h / weights must be 16-byte aligned");
    if (inc ? (a->t_copy_out || a->step_copy_out) : (a->next_step_out != nullptr))
        return fail(SSD_ERR_INVALID, "counter hand-over: the env head writes the copies, the inc head the next values");
    if ((a->next_step_out && a->next_step_out == a->step) || (a->t_copy_out && (!a->t_index || a->t_copy_out == a->t_index)) ||
        (a->step_copy_out && a->step_copy_out == a->step))
        return fail(SSD_ERR_INVALID, "counter hand-over: a launch must not write a scalar it reads");
    return SSD_OK;
}

static int policy_head(const ssd_policy_head* a, int inc, void* stream) {
    if (const int bad = check_head(a, inc)) return bad;
    const int rc = launch_policy_head(a, inc, (hipStream_t)stream);
    if (rc == -3) return fail(SSD_ERR_UNSUPPORTED, "ssd_policy_head is instantiated for n_actions 9 (Cleanup) and 8 (Harvest)");
    if (rc) return fail(SSD_ERR_DEVICE, "ssd_policy_head: launch / hipFuncSetAttribute(max dynamic LDS) failed");
    return launched();
}
int ssd_policy_head_inc_encode(const ssd_policy_head* h, const ssd_policy_encode_args* e, void* stream) {
    if (const int bad = check_head(h, 1)) return bad;
    if (const int bad = check_encode(e)) return bad;
    if (e->act || e->slot_t_copy || e->counter_inc) return fail(SSD_ERR_INVALID, "ssd_policy_head_inc_encode: no act / slot_t_copy / counter_inc");
    if ((h->precision == 1) != (e->precision == 1)) return fail(SSD_ERR_INVALID, "ssd_policy_head_inc_encode: one precision for both halves");
    if (e->out == h->inputs) return fail(SSD_ERR_INVALID, "ssd_policy_head_inc_encode: the encoder must write the other inputs buffer");
    if (e->slot_t && (e->slot_t == h->next_t_out)) return fail(SSD_ERR_INVALID, "ssd_policy_head_inc_encode: the encoder must not read the scalar the inc head writes");
    const int rc = launch_policy_inc_encode(h, e, (hipStream_t)stream);
    if (rc == -3) return fail(SSD_ERR_UNSUPPORTED, "ssd_policy_head is instantiated for n_actions 9 (Cleanup) and 8 (Harvest)");
    if (rc == -2) return fail(SSD_ERR_UNSUPPORTED, "ssd_policy_encode is instantiated for 15 x 15 / 31 x 31 windows");
    if (rc) return fail(SSD_ERR_DEVICE, "ssd_policy_head_inc_encode: launch / hipFuncSetAttribute(max dynamic LDS) failed");
    return launched();
}
int ssd_policy_head_plan(int32_t n_env, int32_t n_agents, int32_t fused_with_encoder, int32_t* workgroups_per_agent, int32_t* compute_waves,
                         int32_t* tiles_per_wave) {
    if (!workgroups_per_agent || !compute_waves || !tiles_per_wave) return fail(SSD_ERR_INVALID, "null argument");
    if (n_env < 1 || n_agents < 1) return fail(SSD_ERR_INVALID, "ssd_policy_head_plan: n_env, n_agents >= 1");
    int a = 0, b = 0, c = 0;
    policy_head_plan(n_env, n_agents, fused_with_encoder, &a, &b, &c);
    *workgroups_per_agent = a; *compute_waves = b; *tiles_per_wave = c;
    return SSD_OK;
}
int ssd_policy_head_env(const ssd_policy_head* a, void* stream) { return policy_head(a, 0, stream); }
int ssd_policy_head_inc(const ssd_policy_head* a, void* stream) { return policy_head(a, 1, stream); }

}  // extern "C"
