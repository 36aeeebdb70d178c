// ssd_env.hip -- gfx950 kernels of the vectorised SSD grid world (Cleanup / Harvest).
//
// Execution model: ONE 64-lane wavefront owns ONE env for the whole transition.  Its grid lives in LDS, its agents
// live one per lane (lanes 0..n-1) in registers, and every "dict lookup" of the reference's Python becomes a wave
// ballot / readlane.  There are no workgroup barriers: the 4 waves of a workgroup are independent envs, so a wave
// that needs the serial conflict resolution does not hold up its neighbours.
//
// The kernel is VALU-issue bound, not HBM bound (profiles/): the design below minimises instructions per env:
//   - the wave index is forced scalar, so every per-env address and the Philox base are computed on the scalar unit;
//   - the observation is produced in three linear passes through LDS: a zero-padded class map (no bounds tests in the
//     window gather), the windows gathered into LDS in exactly the output order [n][3][V][V], and a byte -> float
//     expansion with aligned ds_read + 16-byte global stores (no per-element index decode).
//
// Reference behaviour restated here (cited per function): src/envs/ssd/map_env.py, cleanup.py, harvest.py, agent.py,
// src/utils/utility_funcs.py.  The CPU oracle under oracle/ is a separate, serial restatement used only by tests.
#include <type_traits>

#include "ssd_device.h"

namespace ssd {

// Diagnostic build (tools/stamps.py, -DSSD_STAMPS): lane 0 records s_memtime at phase boundaries after draining
// the wave's outstanding memory operations.  In the product build the macros are empty and no stamp executes.
#if defined(SSD_STAMPS) && SSD_STAMPS == 2
// -DSSD_STAMPS=2: stamps go to 256 bytes of LDS per wave and leave in one store at the end of the kernel: no wait for anything in
// flight, no store in flight that a later `s_waitcnt vmcnt(0)` of the kernel would have to wait for -- the product's timeline
#define STAMP_TO(buf, i)                                                                               \
    do {                                                                                               \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                                          \
        if (lane == 0) E.stamp_lds[i] = t_;                                                            \
    } while (0)
#define STAMP_REAL(buf, i)                                                                             \
    do {                                                                                               \
        unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                      \
        if (lane == 0) E.stamp_lds[i] = t_;                                                            \
    } while (0)
#define STAMP_HWID(buf, j)                                                                             \
    do {                                                                                               \
        unsigned x_ = __builtin_amdgcn_s_getreg((3 << 11) | 20), h_ = __builtin_amdgcn_s_getreg((31 << 11) | 4); \
        if (lane == 0) E.stamp_lds[j] = (unsigned long long)x_ | ((unsigned long long)h_ << 8);        \
    } while (0)
#define STAMP_FLUSH(buf)                                                                               \
    do {                                                                                               \
        wsync();                                                                                       \
        if ((buf) && lane < 32) (buf)[(size_t)env * 32 + lane] = E.stamp_lds[lane];                     \
    } while (0)
#elif defined(SSD_STAMPS)
#define STAMP_TO(buf, i)                                                                               \
    do {                                                                                               \
        __builtin_amdgcn_s_waitcnt(0);                                                                 \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                                          \
        if ((buf) && lane == 0) (buf)[(size_t)env * 32 + (i)] = t_;                                    \
    } while (0)
// slot i <- the chip-wide 100 MHz clock (s_memrealtime: comparable across XCDs, s_memtime is not), slot j <- XCC_ID | HW_ID << 8
#define STAMP_REAL(buf, i)                                                                             \
    do {                                                                                               \
        unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                      \
        if ((buf) && lane == 0) (buf)[(size_t)env * 32 + (i)] = t_;                                    \
    } while (0)
#define STAMP_HWID(buf, j)                                                                             \
    do {                                                                                               \
        unsigned x_ = __builtin_amdgcn_s_getreg((3 << 11) | 20), h_ = __builtin_amdgcn_s_getreg((31 << 11) | 4); \
        if ((buf) && lane == 0) (buf)[(size_t)env * 32 + (j)] = (unsigned long long)x_ | ((unsigned long long)h_ << 8); \
    } while (0)
#define STAMP_FLUSH(buf) do {} while (0)
#else
#define STAMP_TO(buf, i) do {} while (0)
#define STAMP_REAL(buf, i) do {} while (0)
#define STAMP_HWID(buf, j) do {} while (0)
#define STAMP_FLUSH(buf) do {} while (0)
#endif
#define STAMP(i) STAMP_TO(st.stamps, i)
#define STAMP_OBS(i) STAMP_TO(oo.stamps, i)

// ---------------------------------------------------------------------------------------------------------------
// wave primitives
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int rl(int v, int idx) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(idx)); }
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {  // set bits of m in lanes below mine
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ int popc64(uint64_t m) { return __builtin_popcountll(m); }
__device__ __forceinline__ int first_lane(uint64_t m) { return __builtin_ctzll(m); }
__device__ __forceinline__ int last_lane(uint64_t m) { return 63 - __builtin_clzll(m); }
// Order the LDS traffic of the lanes of this wave.  LDS executes one wave's instructions in issue order and no other
// wave touches this wave's slice, so a wavefront-scope fence (a compiler ordering constraint, no s_waitcnt) suffices.
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t udiv(uint32_t x, uint32_t magic) { return __umulhi(x, magic); }

// COUNTER-mode generator (include/ssd_hip.h): b = Philox4x32-10({0, 0, env, epoch}, seed) once per call (wave-uniform,
// scalar unit), then x(stream, k) = mix32(b[stream] ^ k) per draw.
__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* o) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // triple32
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    return x;
}

struct Rng {
    bool tape;
    const double* tape_u;  // this env's row
    int ustride;
    uint32_t b[4];         // Philox base words of the episode, one per stream (wave-uniform), pre-xored with (call << 16)
    int32_t* err;
    __device__ __forceinline__ uint32_t u32(int stream, uint32_t k) const { return mix32(b[stream] ^ k); }
    // np.random.rand(1)[0] number k of this call (cleanup.py:172,183; harvest.py:119)
    __device__ __forceinline__ double uniform(int k) const {
        if (tape) return k < ustride ? tape_u[k] : 2.0;   // lanes past the first success read speculatively
        return (double)(u32(SSD_STREAM_UNIFORM, (uint32_t)k) >> 8) * (1.0 / 16777216.0);
    }
    // `uniform(k) < p` without fp64 in the COUNTER path: the draw is m * 2^-24 with the 24-bit integer m, and p * 2^24 is an
    // exact fp64 scaling, so m * 2^-24 < p  <=>  m < ceil(p * 2^24) = threshold(p)  (exactly, for every double p).
    static __device__ __forceinline__ uint32_t threshold(double p) {
        return p >= 1.0 ? (1u << 24) : !(p > 0.0) ? 0u : (uint32_t)ceil(p * 16777216.0);
    }
    __device__ __forceinline__ bool below(int k, double p, uint32_t thr) const {
        if (tape) return (k < ustride ? tape_u[k] : 2.0) < p;
        return (u32(SSD_STREAM_UNIFORM, (uint32_t)k) >> 8) < thr;
    }
};

// Everything one wave knows about its env.
struct Env {
    const DevSpec* S;   // tables (HBM)
    const DevHead* h;   // scalars (kernel arguments)
    uint8_t* g;     // LDS grid [GS]
    uint8_t* occ;   // LDS agent overlay [GS]: 0 or the agent char (1..9) of the highest id standing there
    uint8_t* pm;    // LDS padded class map [PMS] (observe) / scratch (step)
    uint8_t* pl;    // LDS output planes
    uint8_t* cbuf;  // LDS class-code window of one agent (obs_code side output)
    int lane, n, W, HW, GS;
    int obs_slot;   // the env's step counter after this call (time slot of the observation in an episode storage)
    int32_t* err;   // sticky error word
    bool ag;        // lane < n
    int P;          // my agent's cell (r * W + c); unique negative for non-agent lanes
    int O;          // orientation
    int ap[4], ws[4];  // my lanes' apple / waste site cells (site index = chunk * 64 + lane), preloaded
    // Table windows requested with the state loads, so that the step's two table look-ups are lane reads instead of dependent
    // trips to L2 in the middle of the wave's chain: lane l holds tab_thr[w0 - l] (COUNTER mode; beams only remove waste) and
    // tab_den[a0 - 8 + l] (at most n apples eaten; more than 55 grown in one step: computed).  w0 / a0 < 0: none.
    int w0, a0;
    uint32_t pf_ta, pf_tw, pf_fl;
    float pf_den;
    unsigned long long* stamps; unsigned long long* stamp_lds; int env;   // diagnostic builds
    bool pm_zeroed;     // the padded class map was cleared with the state loads and nothing wrote to it since
};

__device__ __forceinline__ int agent_char(int a) { int v = (a % 10) + 1; return v >= 10 ? 1 : v; }  // '<U1' truncation, map_env.py:370,377

// count cells with g == code (and, if free_only, no agent on them)
__device__ __forceinline__ int count_cells(const Env& E, uint32_t code, bool free_only) {
    int cnt = 0;
    const uint32_t pat = code * 0x01010101u;
    for (int base = 0; base < E.GS; base += 4 * kWave) {
        int i = base + 4 * E.lane;
        uint32_t t = 0x7F7F7F7Fu;  // "no match" for lanes past the end
        if (i < E.GS) {
            t = *(const uint32_t*)(E.g + i) ^ pat;          // zero byte <=> match (padding bytes hold 0 = C_EMPTY)
            if (free_only) t |= *(const uint32_t*)(E.occ + i);
        }
        uint32_t nz = (t + 0x7F7F7F7Fu) & 0x80808080u;      // bytes are < 0x80: bit 7 set <=> byte != 0
#pragma unroll
        for (int j = 0; j < 4; ++j) cnt += popc64(ballot(!((nz >> (8 * j + 7)) & 1u)));
    }
    return cnt;
}

// ---------------------------------------------------------------------------------------------------------------
// update_moves (map_env.py:477-661)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void resolve_moves(Env& E, int T, uint64_t mover_mask, const Rng& R, const uint8_t* tape_order) {
    const int n = E.n, lane = E.lane;
    const bool mover = (mover_mask >> lane) & 1;
    const int nm = popc64(mover_mask);
    // O = shuffled mover list (map_env.py:540-542); lane i holds O[i]
    int Oi = 0xFF;
    if (R.tape) {
        if (lane < nm) Oi = tape_order[lane];
        bool bad = lane < nm && (Oi >= n || !((mover_mask >> (Oi & 63)) & 1));
        if (ballot(bad)) { if (lane == 0) atomicOr(R.err, ERR_BAD_TAPE); Oi = 0; }
    } else {
        uint32_t key = R.u32(SSD_STREAM_MOVE, (uint32_t)lane) >> 8;
        int ork = 0;
        for (int b = 0; b < n; ++b) {
            uint32_t kb = (uint32_t)rl((int)key, b);
            bool mb = (mover_mask >> b) & 1;
            ork += mb && (kb < key || (kb == key && b < lane));
        }
        for (int b = 0; b < n; ++b) { int ob = rl(ork, b); if (((mover_mask >> b) & 1) && ob == lane) Oi = b; }
    }
    int M = T;  // agent_moves[a]
    // contested cells in lexicographic (= cell index) order (np.unique, :543-544, :553-609)
    int cntT = 0;
    for (int b = 0; b < n; ++b) { int tb = rl(T, b); cntT += ((mover_mask >> b) & 1) && tb == T; }
    int last = -1;
    for (;;) {
        uint64_t cand = ballot(mover && cntT > 1 && T > last);
        if (!cand) break;
        int u = 0x7FFFFFFF;
        for (uint64_t m = cand; m; m &= m - 1) { int tb = rl(T, first_lane(m)); u = tb < u ? tb : u; }
        last = u;
        const uint64_t cont = ballot(mover && T == u);
        const uint64_t live = ballot(E.ag && E.P == u);                      // move in self.agent_pos (:567)
        bool cell_free = true;
        if (live) {
            const int occ = last_lane(live);                                  // agent_by_pos: highest id wins (:570)
            const int Pocc = rl(E.P, occ), Mocc = rl(M, occ);
            const bool hm_occ = (mover_mask >> occ) & 1;
            const int cm = hm_occ ? Mocc : Pocc;                              // agent_moves.get(occ, pos) (:574)
            const bool blocked = lane == occ || !hm_occ || Pocc == cm || (Mocc == E.P && u == Pocc);  // (:578-594)
            cell_free = ballot(((cont >> lane) & 1) && blocked) == 0;
        }
        if (cell_free) {                                                      // :598-601 first contender in O order moves in
            int winner = -1;
            for (int i = 0; i < nm; ++i) { int c = rl(Oi, i); if (winner < 0 && ((cont >> (c & 63)) & 1)) winner = c; }
            if (lane == winner) E.P = u;
        }
        if ((cont >> lane) & 1) M = E.P;                                      // :604-609
    }
    // settle loop (:612-661)
    uint64_t hm = mover_mask;
    while (hm) {
        const int SN = E.P;                   // agent_by_pos snapshot of this pass (:613)
        const uint64_t incopy = hm;           // moves_copy keys (:616)
        uint64_t del = 0;
        const int before = popc64(hm);
        for (int a = 0; a < n; ++a) {
            const uint64_t bit_a = 1ull << a;
            if (!(incopy & bit_a) || (del & bit_a)) continue;
            const int m = rl(M, a);
            const uint64_t live = ballot(E.ag && E.P == m);                   // move in self.agent_pos (:621)
            if (live) {
                const uint64_t snap = ballot(E.ag && SN == m);
                if (!snap) {                                                  // KeyError in the reference; unreachable
                    if (lane == 0) atomicOr(R.err, ERR_KEYERROR);
                    hm &= ~bit_a; del |= bit_a; continue;
                }
                const int occ = last_lane(snap);                              // :624
                const uint64_t bit_o = 1ull << occ;
                const int Pocc = rl(E.P, occ), Mocc = rl(M, occ), Pa = rl(E.P, a);
                const int cm = (hm & bit_o) ? Mocc : Pocc;                    // live agent_moves (:627)
                if (a == occ) { hm &= ~bit_a; del |= bit_a; }                 // :630
                else if (!(incopy & bit_o) || Pocc == cm) { hm &= ~bit_a; del |= bit_a; }            // :636
                else if (Mocc == Pa && m == Pocc) { hm &= ~(bit_a | bit_o); del |= bit_a | bit_o; }  // :642-648
            } else {                                                          // :650-653
                if (lane == a) E.P = m;
                hm &= ~bit_a; del |= bit_a;
            }
        }
        if (popc64(hm) == before) {                                           // :658-661 cycles: move them all
            if ((hm >> lane) & 1) E.P = M;
            break;
        }
    }
}

// turns, wall-checked targets (map_env.py:498-511), then conflict resolution
__device__ __forceinline__ void move_phase(Env& E, int act, const Rng& R, const uint8_t* tape_order) {
    const int n = E.n, lane = E.lane;
    const bool mover = E.ag && (unsigned)act <= 4u;
    if (E.ag && act == 5) E.O = (0x0132 >> (4 * E.O)) & 3;   // TURN_CLOCKWISE: LEFT->UP->RIGHT->DOWN (map_env.py:853-861)
    if (E.ag && act == 6) E.O = (0x1023 >> (4 * E.O)) & 3;   // TURN_COUNTERCLOCKWISE: LEFT->DOWN->RIGHT->UP (:844-852)
    int T = E.P;
    if (mover && act != 4) {
        // ACTIONS vectors [row, col] (map_env.py:20-24) rotated by orientation (rotate_action :826-841)
        int v0 = act == 0 ? -1 : act == 1 ? 1 : 0;
        int v1 = act == 2 ? -1 : act == 3 ? 1 : 0;
        int d0, d1;
        if (E.O == O_UP) { d0 = v0; d1 = v1; }
        else if (E.O == O_LEFT) { d0 = v1; d1 = -v0; }        // np.dot([[0,1],[-1,0]], v)
        else if (E.O == O_RIGHT) { d0 = -v1; d1 = v0; }       // np.dot([[0,-1],[1,0]], v)
        else { d0 = -v0; d1 = -v1; }
        int t = E.P + d0 * E.W + d1;                          // the border is all wall, so t stays inside the map
        if ((unsigned)t < (unsigned)E.HW && E.g[t] != C_WALL) T = t;   // return_valid_pos (agent.py:111-119)
    }
    const uint64_t mover_mask = ballot(mover);
    if (!mover_mask) return;                                  // :534
    // fast path: nobody targets another agent's cell and no two movers share a target => every move applies
    bool clash = false;
    for (int b = 0; b < n; ++b) {
        int pb = rl(E.P, b), tb = rl(T, b);
        bool mb = (mover_mask >> b) & 1;
        if (mover && b != lane) clash |= (T == pb) | (mb && T == tb);
    }
    if (ballot(clash) == 0) { if (mover) E.P = T; return; }
    resolve_moves(E, T, mover_mask, R, tape_order);
}

// ---------------------------------------------------------------------------------------------------------------
// CLEAN beams (map_env.py:687-769 with cleanup.py:135-143).  FIRE beams change nothing but the shooter's reward
// (hit() is a no-op, agent.py:184-186,246-248; blocking_cells='P' never occurs), so they are not traced.
// Returns the number of cells cleaned by agent f; 15 lanes = 3 beams x 5 cells.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clean_beams(Env& E, int f) {
    const int W = E.W, H = E.h->H;
    const int pf = rl(E.P, f), of = rl(E.O, f);
    const int pr = (int)udiv((uint32_t)pf, E.h->magic_W), pc = pf - pr * W;
    const int dr = of == O_LEFT ? -1 : of == O_RIGHT ? 1 : 0;
    const int dc = of == O_UP ? -1 : of == O_DOWN ? 1 : 0;
    const int rr = -dc, rc = dr;                      // rotate_right(dir) = np.dot([[0,-1],[1,0]], dir)
    const int b = E.lane / 5, k = E.lane - 5 * b;     // beam, step
    const int sr = b == 0 ? pr : b == 1 ? pr + rr - dr : pr - rr - dr;
    const int sc = b == 0 ? pc : b == 1 ? pc + rc - dc : pc - rc - dc;
    const int r = sr + (k + 1) * dr, c = sc + (k + 1) * dc;
    const bool act15 = E.lane < 15;
    const bool inb = act15 && (unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W;
    const int cell = inb ? r * W + c : 0;
    const int gc = inb ? E.g[cell] : C_WALL;
    const int oc = inb ? E.occ[cell] : 0;
    const bool stop = act15 && (!inb || gc == C_WALL || oc != 0 || gc == C_WASTE);
    const uint32_t m = (uint32_t)ballot(stop);
    const uint32_t field = b < 3 ? (m >> (5 * b)) & 31u : 0u;
    const bool first = stop && (field & ((1u << k) - 1u)) == 0;
    const bool hit = first && inb && gc == C_WASTE;   // :745-754: 'H' -> 'R', with or without an agent on it
    const int cnt = popc64(ballot(hit));
    wsync();
    if (hit) E.g[cell] = C_RIVER;
    wsync();
    return cnt;
}

// ---------------------------------------------------------------------------------------------------------------
// custom_map_update: Cleanup (cleanup.py:146-204) / Harvest (harvest.py:86-122).  Returns the uniforms consumed.
// ---------------------------------------------------------------------------------------------------------------
// wave-wide unsigned minimum with DPP (no LDS crossbar): quad swaps, row rotations, then row broadcasts; lane 63 ends with
// the minimum of all 64 lanes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_min(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
    return o < v ? o : v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_min<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v = dpp_min<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v = dpp_min<0x124, 0xF>(v);    // row_ror:4
    v = dpp_min<0x128, 0xF>(v);    // row_ror:8
    v = dpp_min<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
    v = dpp_min<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <bool TAPE>
__device__ __forceinline__ int spawn_cleanup(Env& E, const Rng& R, const uint8_t* tape_waste, int& n_waste_cells, int& n_apple_cells) {
    const DevSpec* S = E.S;
    const DevHead* h = E.h;
    const int lane = E.lane;
    // compute_probabilities (cleanup.py:189-204): looked up by the waste count in the host-built tables -- the fp64 probabilities
    // where recorded doubles are compared (TAPE), the 24-bit thresholds of Rng::threshold and two flags where the counter
    // generator's integers are (no fp64 instruction in that instantiation)
    const int current = h->n_waste > 0 ? n_waste_cells : 0;    // kept incrementally in the env state
    double p_apple = 0.0, p_waste = 0.0;
    uint32_t t_apple = 0, t_waste = 0;
    bool apple_on, waste_on;
    if (TAPE) {
        p_apple = S->tab_p[current][0]; p_waste = S->tab_p[current][1];
        apple_on = p_apple > 0; waste_on = !(fabs(p_waste) <= 1e-8);                 // np.isclose(p, 0)
    } else {
        const int back = E.w0 - current;                       // (wave-uniform)
        uint32_t fl;
        if (E.w0 >= 0 && (unsigned)back < (unsigned)kWave) {
            t_apple = (uint32_t)rl((int)E.pf_ta, back); t_waste = (uint32_t)rl((int)E.pf_tw, back); fl = (uint32_t)rl((int)E.pf_fl, back);
        } else {
            const uint4 tv = *(const uint4*)S->tab_thr[current];
            t_apple = tv.x; t_waste = tv.y; fl = tv.z;
        }
        apple_on = fl & 1u; waste_on = fl & 2u;
    }
    int k = 0;
    // Every LDS read of the phase first -- the apple sites' overlay and grid bytes and the waste sites' grid bytes, <= 12 reads in ONE
    // round trip: written per chunk (`occ == 0 && g != 'A'`, then the store of the grown apple) each chunk costs two dependent
    // trips, and the waste sites' reads cannot move above the apple stores by themselves (byte buffers alias for the compiler).
    // Apple sites and waste sites are disjoint cells, every site is its own cell and the beams are done, so nothing written below
    // changes what another read of this phase would have seen.
    const int nw = h->n_waste;
    int a_oc[4], a_gc[4], w_gc[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        a_oc[ch] = 1; a_gc[ch] = C_APPLE; w_gc[ch] = C_WASTE;
        if (ch * kWave < h->n_apple) { a_oc[ch] = E.occ[E.ap[ch]]; a_gc[ch] = E.g[E.ap[ch]]; }
        if (ch * kWave < nw) w_gc[ch] = E.g[E.ws[ch]];
    }
    // apples: one draw per site that holds neither an agent nor an apple, in site order (cleanup.py:168-174).
    // Apple sites and waste sites are disjoint, so writing 'A' at once is equivalent to the deferred update_map.
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        if (ch * kWave < h->n_apple) {
            const bool in = ch * kWave + lane < h->n_apple;
            const int cell = E.ap[ch];
            const bool elig = in & (a_oc[ch] == 0) & (a_gc[ch] != C_APPLE);
            const uint64_t bal = ballot(elig);
            bool grow = false;
            if (elig && apple_on) {
                grow = R.below(k + (int)lanes_below(bal), p_apple, t_apple);
                if (grow) E.g[cell] = C_APPLE;
            }
            if (apple_on) n_apple_cells += popc64(ballot(grow));
            k += popc64(bal);
        }
    }
    { const int env = E.env; (void)env; STAMP_TO(E.stamps, 14); }
    // waste: at most one spawn, first free site in shuffled order whose draw succeeds (cleanup.py:177-186)
    if (waste_on) {
        uint16_t* scratch = (uint16_t*)E.pm;                  // tape mode: rank of each site in the shuffled list
        if (R.tape) {
            E.pm_zeroed = false;
            for (int base = 0; base < nw; base += kWave) {
                const int p = base + lane;
                if (p < nw) {
                    int s = tape_waste[p];
                    if (s >= nw) { atomicOr(R.err, ERR_BAD_TAPE); s = 0; }
                    scratch[s] = (uint16_t)p;
                }
            }
            wsync();
        }
        // free sites and their sort keys (<= 4 chunks of 64 sites): composite (24-bit key << 8) | site, unique per site
        int nfree = 0;
        uint32_t best[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            best[ch] = ~0u;
            const int s = ch * kWave + lane;
            if (ch * kWave < nw) {
                const bool fr = s < nw && w_gc[ch] != C_WASTE;
                nfree += popc64(ballot(fr));
                if (fr) {
                    const uint32_t key = R.tape ? (uint32_t)scratch[s] : (R.u32(SSD_STREAM_WASTE, (uint32_t)s) >> 8);
                    best[ch] = (key << 8) | (uint32_t)s;
                }
            }
        }
        // J = index (in shuffled free-site order) of the first successful draw
        int J = -1;
        for (int base = 0; base < nfree && J < 0; base += kWave) {
            const int j = base + lane;
            const bool ok = j < nfree && R.below(k + j, p_waste, t_waste);
            const uint64_t b = ballot(ok);
            if (b) J = base + first_lane(b);
        }
        k += J >= 0 ? J + 1 : nfree;
        { const int env = E.env; (void)env; STAMP_TO(E.stamps, 15); }
        if (J >= 0) {
            // the (J+1)-th smallest (key, site) among the free sites
            uint32_t sel = ~0u;
            for (int it = 0; it <= J; ++it) {
                uint32_t mn = best[0];
#pragma unroll
                for (int ch = 1; ch < 4; ++ch) mn = best[ch] < mn ? best[ch] : mn;
                sel = wave_min_u32(mn);
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) if (best[ch] == sel) best[ch] = ~0u;
            }
            const int s = (int)(sel & 0xFFu), sc = s >> 6;           // site s = chunk sc, lane s & 63: that lane holds its cell
            if (lane == (s & 63)) E.g[sc == 0 ? E.ws[0] : sc == 1 ? E.ws[1] : sc == 2 ? E.ws[2] : E.ws[3]] = C_WASTE;
            n_waste_cells += 1;
        }
    }
    wsync();
    return k;
}

__device__ __forceinline__ int spawn_harvest(Env& E, const Rng& R, int& n_apple_cells, const double (&harvest_p)[4], const uint32_t (&t_harvest)[4]) {
    const DevHead* h = E.h;
    const int lane = E.lane, W = E.W, H = h->H;
    int k = 0;
    uint32_t spawn_bits = 0;  // decisions are applied after ALL sites were examined (synchronous update, harvest.py:86-90)
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        if (ch * kWave < h->n_apple) {
            const bool in = ch * kWave + lane < h->n_apple;
            const int cell = E.ap[ch];
            const bool elig = in && E.occ[cell] == 0 && E.g[cell] != C_APPLE;
            const uint64_t bal = ballot(elig);
            if (elig) {
                const int r = (int)udiv((uint32_t)cell, h->magic_W), c = cell - r * W;
                int num = 0;  // offsets with j*j + k*k <= APPLE_RADIUS (= 2): the 3x3 block (harvest.py:108-116)
#pragma unroll
                for (int j = -1; j <= 1; ++j)
#pragma unroll
                    for (int q = -1; q <= 1; ++q) {
                        const int x = r + j, y = c + q;
                        if ((unsigned)x < (unsigned)H && (unsigned)y < (unsigned)W) num += E.g[x * W + y] == C_APPLE;
                    }
                const int pi = num < 3 ? num : 3;
                if (R.below(k + (int)lanes_below(bal), pi == 0 ? harvest_p[0] : pi == 1 ? harvest_p[1] : pi == 2 ? harvest_p[2] : harvest_p[3], pi == 0 ? t_harvest[0] : pi == 1 ? t_harvest[1] : pi == 2 ? t_harvest[2] : t_harvest[3])) spawn_bits |= 1u << ch;
            }
            n_apple_cells += popc64(ballot((spawn_bits >> ch) & 1));
            k += popc64(bal);
        }
    }
    wsync();
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
        if ((spawn_bits >> ch) & 1) E.g[E.ap[ch]] = C_APPLE;
    wsync();
    return k;
}

// ---------------------------------------------------------------------------------------------------------------
// observation (map_env.py:360-379,418-446,795-815,923-957; utility_funcs.py:58-116)
// ---------------------------------------------------------------------------------------------------------------
// class of one map cell.  Simplified colours: a channel bit mask, 1 = R (waste), 2 = G (apple), 4 = B (wall or agent).
// Full colours: LUT row (cell code, or 5 + agent char).
template <bool FULL>
__device__ __forceinline__ int class_of(int gc, int oc, int kind) {
    if (FULL) return oc ? 5 + oc : gc;
    if (oc || gc == C_WALL) return 4;
    if (gc == C_APPLE) return 2;
    return (gc == C_WASTE && kind == SSD_ENV_CLEANUP) ? 1 : 0;
}
// byte value (0..255) of channel ch for a class
template <bool FULL>
__device__ __forceinline__ uint32_t class_value(const uint8_t* lut, int cls, int ch) {
    if (FULL) return lut[cls * 3 + ch];
    return ((cls >> ch) & 1) ? 255u : 0u;
}

template <typename T> struct Cvt;
template <> struct Cvt<float> { static __device__ __forceinline__ float f(uint32_t v) { return (float)v * (1.0f / 256.0f); } };
template <> struct Cvt<uint16_t> { static __device__ __forceinline__ uint16_t f(uint32_t v) { return (uint16_t)(__float_as_uint((float)v * (1.0f / 256.0f)) >> 16); } };
template <> struct Cvt<uint8_t> { static __device__ __forceinline__ uint8_t f(uint32_t v) { return (uint8_t)v; } };

// Generic element-wise emitter (used for the rarely requested `state` output): L elements val(0..L-1) to dst[0..L)
// with 16-byte stores wherever the address allows.  elem_off = index of dst[0] in the 16-byte aligned tensor.
template <typename T, typename F>
__device__ __forceinline__ void emit(T* dst, size_t elem_off, int L, int lane, F val) {
    constexpr int EPV = 16 / (int)sizeof(T);
    int head = (int)((EPV - (elem_off % EPV)) % EPV);
    if (head > L) head = L;
    for (int i = lane; i < head; i += kWave) dst[i] = Cvt<T>::f(val(i));
    const int nvec = (L - head) / EPV;
    for (int q = lane; q < nvec; q += kWave) {
        const int f0 = head + q * EPV;
        union { T e[EPV]; uint4 v; } u;
#pragma unroll
        for (int j = 0; j < EPV; ++j) u.e[j] = Cvt<T>::f(val(f0 + j));
        *(uint4*)(dst + f0) = u.v;
    }
    const int t0 = head + nvec * EPV;
    for (int i = t0 + lane; i < L; i += kWave) dst[i] = Cvt<T>::f(val(i));
}

// Expansion of LDS bytes to the output dtype (value / 256), 16 bytes per lane per store, packed explicitly so the
// compiler cannot split the store.  Byte f of the env's block sits at pl[f + delta]; delta makes every vector's source
// ONE naturally aligned LDS word.
__device__ __forceinline__ float b2f(uint32_t w, int j) { return (float)((w >> (8 * j)) & 0xFFu) * (1.0f / 256.0f); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {   // k/256 is exact in bf16: truncation == rounding
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// ONE: the plane bytes are 0 / 1 flags of the simplified palette (value 255 / 256 where set): the f32 bit pattern is one integer
// multiply per element (byte * 0x3F7F0000) instead of a convert and a multiply
template <typename T, bool ONE = false> struct Expand;
template <> struct Expand<float, false> {
    static __device__ __forceinline__ uint4 vec(const uint8_t* src, int q) {
        const uint32_t w = *(const uint32_t*)(src + 4 * q);
        return make_uint4(__float_as_uint(b2f(w, 0)), __float_as_uint(b2f(w, 1)), __float_as_uint(b2f(w, 2)), __float_as_uint(b2f(w, 3)));
    }
};
template <> struct Expand<float, true> {
    static __device__ __forceinline__ uint4 vec(const uint8_t* src, int q) {
        const uint32_t w = *(const uint32_t*)(src + 4 * q);
        constexpr uint32_t K = 0x3F7F0000u;                       // 255 / 256 as f32
        return make_uint4((w & 0xFFu) * K, ((w >> 8) & 0xFFu) * K, ((w >> 16) & 0xFFu) * K, (w >> 24) * K);
    }
};
template <> struct Expand<uint16_t, false> {
    static __device__ __forceinline__ uint4 vec(const uint8_t* src, int q) {
        const uint2 w = *(const uint2*)(src + 8 * q);
        return make_uint4(pack_bf16(b2f(w.x, 0), b2f(w.x, 1)), pack_bf16(b2f(w.x, 2), b2f(w.x, 3)),
                          pack_bf16(b2f(w.y, 0), b2f(w.y, 1)), pack_bf16(b2f(w.y, 2), b2f(w.y, 3)));
    }
};
template <> struct Expand<uint8_t, false> {
    static __device__ __forceinline__ uint4 vec(const uint8_t* src, int q) { return *(const uint4*)(src + 16 * q); }
};

// vectors [q0, q1) of the env's block: out = dst + head, src = pl + head + delta.  U vectors per lane per batch so that U LDS
// reads are in flight before the first store issues; U is sized to one agent's share of the block at V = 15 (169 / 85 / 43
// vectors for f32 / bf16 / u8), so that the common call is ONE batch without dead slots.
template <typename T, bool ONE>
__device__ __forceinline__ void expand_range(const uint8_t* src, T* out, int q0, int q1, int lane) {
    constexpr int EPV = 16 / (int)sizeof(T);
    constexpr int U = sizeof(T) == 4 ? 3 : sizeof(T) == 2 ? 2 : 1;
    for (int q = q0 + lane; q < q1; q += U * kWave) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (q + u * kWave < q1) ? Expand<T, ONE>::vec(src, q + u * kWave) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#ifdef SSD_NOSTORE   // diagnostic only: keep the compute, drop (almost) every store
            if (v[u].x == 0x12345678u)
#endif
            if (q + u * kWave < q1) *(uint4*)(out + (size_t)(q + u * kWave) * EPV) = v[u];
        }
    }
}

// FMT: SSD_OBS_F32 / BF16 / U8 (three byte planes per agent) or SSD_OBS_CODE (one class plane per agent)
// WC: also emit the window as one channel-mask byte per cell (bit 0 R, 1 G, 2 B; simplified palette) into the dense side buffer
// oo.code -- what the rollout-time encoder reads: one extra LDS byte per cell in the gather, one 16-byte store per lane and agent.
// PMC: the class map already holds SSD_OBS_CODE values (observe_phase wrote them: nothing else reads the map in this call)
template <bool FULL, int FMT, bool WC, bool PMC = false, int NT = 0>
__device__ __forceinline__ void observe_windows(Env& E, int env, const DevObsOut& oo, const uint8_t* lut) {
    typedef typename std::conditional<FMT == SSD_OBS_F32, float, typename std::conditional<FMT == SSD_OBS_BF16, uint16_t, uint8_t>::type>::type T;
    constexpr bool CODE = FMT == SSD_OBS_CODE;
    constexpr bool ONE = false;                                   // (0 / 1 plane flags + integer-multiply expansion: measured no faster)
    constexpr int EPV = 16 / (int)sizeof(T);
    const DevHead* h = E.h;
    const int lane = E.lane, n = E.n, W = E.W, V = h->V, VV = h->VV, Wp = h->Wp;
    const int A = CODE ? VV : 3 * VV;                             // elements per agent
    const int L = n * A;
    // element offset of this env's block: dense, or slot ep_step of an episode storage [n_env, t_slots, n, ...]
    const size_t off = oo.env_stride ? (size_t)env * (size_t)oo.env_stride + (size_t)E.obs_slot * (size_t)oo.slot_stride : (size_t)env * L;
    int head = (int)((EPV - (off % EPV)) % EPV);
    if (head > L) head = L;
    const int delta = (EPV - head % EPV) % EPV;
    T* dst = (T*)oo.obs + off;
    const uint8_t* src = E.pl + head + delta;                     // EPV-byte aligned
    const int nvec = (L - head) / EPV;
    // Rows of a window are dealt to lanes 2^vshift at a time: lane = (row in group, column).
    const int sh = h->vshift, j = lane & ((1 << sh) - 1), il = lane >> sh, rpi = kWave >> sh;
    const bool jv = j < V;
    if (!CODE && !FULL) {   // simplified palette: at most one of the three channel bytes of a cell is non-zero -> zero all, set one
        for (int i = lane * 16; i < lds_planes_bytes(*h); i += kWave * 16) *(uint4*)(E.pl + i) = make_uint4(0, 0, 0, 0);
        wsync();
    }
    if (CODE && !WC) {
        // Class-code windows (the episode storage's format).  The wave-uniform control per agent of the general path (orientation
        // selects, loop bounds, one exec mask per predicated store, an expand call per agent: ~120 scalar instructions per agent)
        // was the kernel's tightest issue slot.  Here lane a < n computes agent a's window origin and rot90 coefficients as vector
        // code, the per-agent loop fetches them with three v_readlane, every pass is address arithmetic + one class read + one code
        // write (idle lanes and cells outside the window write to a dump byte: no exec masks), and the codes leave in ONE batch.
        const int pr0 = (int)udiv((uint32_t)E.P, h->magic_W), pc0 = E.P - pr0 * W;
        int ci0, cj0, c00;
        if (E.O == O_UP) { ci0 = Wp; cj0 = 1; c00 = 0; }
        else if (E.O == O_LEFT) { ci0 = -1; cj0 = Wp; c00 = V - 1; }
        else if (E.O == O_DOWN) { ci0 = -Wp; cj0 = -1; c00 = (V - 1) * Wp + (V - 1); }
        else { ci0 = 1; cj0 = -Wp; c00 = (V - 1) * Wp; }
        const int base0 = pr0 * Wp + pc0 + c00;
        const int dumpc = delta + L;                                  // one byte behind the last window (inside the planes' slack)
        // Agents in groups of AG: every class read of the group is in flight before the first code is written (one LDS round
        // trip per group and trip, not one per agent -- the reads and writes are byte accesses to buffers the compiler must
        // assume to alias, so it keeps the source order).  The kernel is bound by the NUMBER of vector instructions (4 waves per
        // SIMD keep the vector pipe ~ 3/4 busy), so nothing here is predicated per agent: lanes outside the window (column V of
        // the 2^vshift dealt, rows past V) read whatever byte their affine index lands on -- within a row and a column of the
        // window, inside this wave's LDS -- and write it to a dump zone behind the last window: the destination offsets are
        // computed once per trip, an agent adds its a * VV.
        constexpr int AG = (NT > 0 && NT < 5) ? NT : 5;
        const int dstep = rpi * V;
        for (int a0 = 0; a0 < n; a0 += AG) {
            int sidx[AG], sstep[AG];
#pragma unroll
            for (int g = 0; g < AG; ++g) {
                const int a = a0 + g < n ? a0 + g : n - 1;
                const int sb = rl(base0, a), sci = rl(ci0, a), scj = rl(cj0, a);
                sstep[g] = rpi * sci;
                sidx[g] = sb + il * sci + j * scj;
            }
            int d = delta + a0 * VV + il * V + j;
            for (int i = il; i < V; i += 4 * rpi, d += 4 * dstep) {
                uint8_t cls[AG][4];
                int wr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) wr[u] = (jv && i + u * rpi < V) ? d + u * dstep : dumpc;
#pragma unroll
                for (int g = 0; g < AG; ++g) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) cls[g][u] = E.pm[sidx[g] + u * sstep[g]];
                    sidx[g] += 4 * sstep[g];
                }
#pragma unroll
                for (int g = 0; g < AG; ++g) {
                    const bool live = (NT > 0 && NT % AG == 0) || a0 + g < n;         // (wave-uniform; a constant for the shipped team sizes)
#pragma unroll
                    for (int u = 0; u < 4; ++u)   // class bit 1 / 2 / 4 -> code 2 / 1 / 3
                        E.pl[(live ? wr[u] : dumpc) + g * VV] = PMC ? cls[g][u] : (uint8_t)((0x30120u >> (4 * cls[g][u])) & 0xFu);
                }
            }
        }
        wsync();
        expand_range<T, ONE>(src, dst + head, 0, nvec, lane);
        if (lane < head) dst[lane] = Cvt<T>::f(E.pl[lane + delta]);
        const int t0 = head + nvec * EPV;
        if (t0 + lane < L) dst[t0 + lane] = Cvt<T>::f(E.pl[t0 + lane + delta]);
        return;
    }
    int q_done = 0;
    const int dump = (int)(lut - E.pl);
    // code windows: every agent's in LDS when they fit (stored in ONE batch behind the last agent, off the per-agent critical
    // path), else one agent's at a time (31 x 31 windows: LDS is what limits the workgroups per CU there)
    const int cstride = oo.code_agent_stride;
    const bool call = WC && lds_code_all(*h);
    uint8_t* cbuf = E.cbuf;
    const int cdump = call ? n * cstride : cstride;
    for (int a = 0; a < n; ++a) {
        const int pa = rl(E.P, a), oa = rl(E.O, a);
        const int pr = (int)udiv((uint32_t)pa, h->magic_W), pc = pa - pr * W;
        // rotate_view = np.rot90(view, k), k = 0 UP, 1 LEFT, 2 DOWN, 3 RIGHT (map_env.py:795-815): output (i, j) reads
        // view (x, y) = UP (i, j); LEFT (j, V-1-i); DOWN (V-1-i, V-1-j); RIGHT (V-1-j, i).  In the padded class map the
        // window's top-left is (pr, pc), so the source index is the affine form base + i * ci + j * cj.
        int ci, cj, c0;
        if (oa == O_UP) { ci = Wp; cj = 1; c0 = 0; }
        else if (oa == O_LEFT) { ci = -1; cj = Wp; c0 = V - 1; }
        else if (oa == O_DOWN) { ci = -Wp; cj = -1; c0 = (V - 1) * Wp + (V - 1); }
        else { ci = 1; cj = -Wp; c0 = (V - 1) * Wp; }
        const int sstep = rpi * ci, dstep = rpi * V;
        int sidx = pr * Wp + pc + c0 + il * ci + j * cj;
        int d = a * A + delta + il * V + j;
        int cd = (call ? a * cstride : 0) + il * V + j;
        // 4 row groups per batch: the 4 class reads are in flight together, then the plane bytes are written
        for (int i = il; i < V; i += 4 * rpi, sidx += 4 * sstep, d += 4 * dstep, cd += 4 * dstep) {
            int cls[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ok[u] = jv && i + u * rpi < V; cls[u] = E.pm[ok[u] ? sidx + u * sstep : 0]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int du = d + u * dstep, c = cls[u];
                if (CODE) {
                    if (ok[u]) E.pl[du] = (uint8_t)(c == 2 ? 1 : c == 1 ? 2 : c == 4 ? 3 : 0);
                } else if (FULL) {
                    if (ok[u]) { E.pl[du] = lut[c * 3]; E.pl[du + VV] = lut[c * 3 + 1]; E.pl[du + 2 * VV] = lut[c * 3 + 2]; }
                } else {
                    // class bit 1 / 2 / 4 = channel R / G / B = plane 0 / 1 / 2.  Branch-free: empty cells and idle lanes write
                    // their 255 into a dump byte behind the planes (the lut slot, unused with the simplified palette)
                    E.pl[(ok[u] && c) ? du + __mul24(c >> 1, VV) : dump] = ONE ? 1 : 255;
                }
                if (WC) cbuf[ok[u] ? cd + u * dstep : cdump] = (uint8_t)c;   // channel mask: 1 R (waste), 2 G (apple), 4 B (wall / agent)
            }
        }
        wsync();
        if (WC && !call) {   // this agent's code window: 16-byte stores (the side buffer's agent stride is a multiple of 16)
            uint8_t* cdst = oo.code + ((size_t)env * n + a) * cstride;
            for (int v16 = lane * 16; v16 < cstride; v16 += kWave * 16) *(uint4*)(cdst + v16) = *(const uint4*)(cbuf + v16);
        }
        // expand and store every vector that is complete by now, so the store stream overlaps the next agent's gather
        const int q_end = a == n - 1 ? nvec : ((a + 1) * A - head) / EPV;
        if (q_end > q_done) { expand_range<T, ONE>(src, dst + head, q_done, q_end, lane); q_done = q_end; }
    }
    if (call) {          // every agent's code window in one batch of 16-byte stores
        uint8_t* cdst = oo.code + (size_t)env * n * cstride;
        for (int v16 = lane * 16; v16 < n * cstride; v16 += kWave * 16) *(uint4*)(cdst + v16) = *(const uint4*)(cbuf + v16);
    }
    // ragged ends (fewer than EPV elements each)
    if (lane < head) dst[lane] = Cvt<T>::f(E.pl[lane + delta] * (ONE ? 255u : 1u));
    const int t0 = head + nvec * EPV;
    if (t0 + lane < L) dst[t0 + lane] = Cvt<T>::f(E.pl[t0 + lane + delta] * (ONE ? 255u : 1u));
}

// OV: the observation variant this instantiation serves.  OV_ANY dispatches on the request at run time (every format, both
// palettes, the side outputs); the other two are what the rollout (class-code windows into the episode storage) and the
// format-R workload (f32 planes, nothing else) ask for, compiled without the dispatch: the step kernel is bound by the number of
// instructions a wave executes and by its footprint in the instruction cache (the all-formats kernel is ~ 12 K instructions,
// a wave executes ~ 1.5 K of them scattered over the whole image).
enum { OV_ANY = 0, OV_CODE = 1, OV_F32 = 2 };
template <bool FULL, int NT, int OV>
__device__ __forceinline__ void observe_phase(Env& E, int env, const DevObsOut& oo) {
    const DevSpec* S = E.S;
    const DevHead* h = E.h;
    const int lane = E.lane, n = E.n, W = E.W, v = h->v, Wp = h->Wp;
    uint8_t* lut = E.pl + lds_planes_bytes(*h);  // 48 bytes behind the planes
    if (FULL) { if (lane < 48) lut[lane] = S->lut[lane]; }
    // pass 0: zero-padded class map, pm[(r + v) * Wp + (c + v)] = class of map cell (r, c); the padding IS
    // return_view's zero padding (utility_funcs.py:93-116), so the window gather needs no bounds test
    if (!E.pm_zeroed)
        for (int i = lane * 16; i < h->PMS; i += kWave * 16) *(uint4*)(E.pm + i) = make_uint4(0, 0, 0, 0);
    wsync();
    const bool codes_in_map = OV == OV_CODE || (OV == OV_ANY && !FULL && oo.obs && oo.fmt == SSD_OBS_CODE && !oo.state);     // (wave-uniform)
    if (!FULL) {
        // simplified palette: 4 cells per lane and trip, classes by byte-SWAR on the packed cell codes (all codes < 0x80):
        // wall or agent -> 4, apple -> 2, waste (Cleanup) -> 1, else 0
        const uint32_t waste_on = h->kind == SSD_ENV_CLEANUP ? 0x80808080u : 0u;
        for (int i4 = lane * 4; i4 < E.HW; i4 += 4 * kWave) {
            const uint32_t g4 = *(const uint32_t*)(E.g + i4), o4 = *(const uint32_t*)(E.occ + i4);
            const uint32_t hi = 0x80808080u, lo = 0x7F7F7F7Fu;
            const uint32_t wo = (((o4 + lo) | ~((g4 ^ 0x01010101u) + lo)) & hi);          // byte != 0 in occ, or code == '@'
            const uint32_t ap = ~((g4 ^ 0x02020202u) + lo) & hi & ~wo;                    // code == 'A', nobody on it
            const uint32_t wa = ~((g4 ^ 0x03030303u) + lo) & waste_on & ~wo;              // code == 'H'
            // class bits 4 / 2 / 1, or -- when only the class-code windows read the map -- the codes themselves (3 / 1 / 2): the gather
            // then copies bytes instead of translating each one
            const uint32_t cls4 = codes_in_map ? ((wo >> 7) | (wo >> 6) | (ap >> 7) | (wa >> 6)) : ((wo >> 5) | (ap >> 6) | (wa >> 7));
            const int r = (int)udiv((uint32_t)i4, h->magic_W);
            int c = i4 - r * W, d = (r + v) * Wp + c + v;
            // cells past H * W (the grid is padded to 16 bytes with code 0, nobody stands there) have class 0 and land in the
            // map's zero padding below the last row: no bound test per cell
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                E.pm[d] = (uint8_t)(cls4 >> (8 * j));
                ++c; ++d;
                if (c == W) { c = 0; d += Wp - W; }
            }
        }
    } else
    for (int cell0 = lane; cell0 < E.HW; cell0 += 4 * kWave) {
        int gc[4], oc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cell = cell0 + u * kWave;
            gc[u] = cell < E.HW ? E.g[cell] : 0; oc[u] = cell < E.HW ? E.occ[cell] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cell = cell0 + u * kWave;
            if (cell < E.HW) {
                const int r = (int)udiv((uint32_t)cell, h->magic_W), c = cell - r * W;
                E.pm[(r + v) * Wp + c + v] = (uint8_t)class_of<FULL>(gc[u], oc[u], h->kind);
            }
        }
    }
    wsync();
    STAMP_OBS(8);
    // an env stepped past its episode storage writes no observation (it would land in the next env's block)
    if (OV == OV_CODE) observe_windows<false, SSD_OBS_CODE, false, true, NT>(E, env, oo, lut);
    else if (OV == OV_F32) observe_windows<false, SSD_OBS_F32, false>(E, env, oo, lut);
    else if (oo.obs) {
        // gather windows into LDS in output order, expand to the output dtype (value / 256, CHW; map_env.py:945)
        if (oo.code && !FULL) {
            if (oo.fmt == SSD_OBS_F32) observe_windows<FULL, SSD_OBS_F32, true>(E, env, oo, lut);
            else if (oo.fmt == SSD_OBS_BF16) observe_windows<FULL, SSD_OBS_BF16, true>(E, env, oo, lut);
            else if (oo.fmt == SSD_OBS_U8) observe_windows<FULL, SSD_OBS_U8, true>(E, env, oo, lut);
            else if (codes_in_map) observe_windows<false, SSD_OBS_CODE, false, true, NT>(E, env, oo, lut);
            else observe_windows<false, SSD_OBS_CODE, false, false, NT>(E, env, oo, lut);
        } else if (oo.fmt == SSD_OBS_F32) observe_windows<FULL, SSD_OBS_F32, false>(E, env, oo, lut);
        else if (oo.fmt == SSD_OBS_BF16) observe_windows<FULL, SSD_OBS_BF16, false>(E, env, oo, lut);
        else if (oo.fmt == SSD_OBS_U8) observe_windows<FULL, SSD_OBS_U8, false>(E, env, oo, lut);
        else if (codes_in_map) observe_windows<false, SSD_OBS_CODE, false, true, NT>(E, env, oo, lut);
        else observe_windows<false, SSD_OBS_CODE, false, false, NT>(E, env, oo, lut);
    }
    STAMP_OBS(9);
    if (OV == OV_ANY && oo.state) {  // get_state (map_env.py:950-957): [3, H, W] / 256
        const int L = 3 * E.HW;
        const size_t off = (size_t)env * L;
        emit<float>(oo.state + off, off, L, lane, [&](int f) -> uint32_t {
            const int ch = (int)udiv((uint32_t)f, h->magic_HW);
            const int cell = f - ch * E.HW;
            const int r = (int)udiv((uint32_t)cell, h->magic_W), c = cell - r * W;
            return class_value<FULL>(lut, E.pm[(r + v) * Wp + c + v], ch);
        });
    }
    if (E.ag) {
        const int pr = (int)udiv((uint32_t)E.P, h->magic_W), pc = E.P - pr * W;
        const size_t o2 = ((size_t)env * n + lane) * 2;
        if (oo.pos) { oo.pos[o2] = (float)pr; oo.pos[o2 + 1] = (float)pc; }                 // get_agent_pos (:917-918)
        if (oo.orient) {                                                                    // ORIENTATIONS vector (:920-921)
            oo.orient[o2] = E.O == O_LEFT ? -1.f : E.O == O_RIGHT ? 1.f : 0.f;
            oo.orient[o2 + 1] = E.O == O_UP ? -1.f : E.O == O_DOWN ? 1.f : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the kernel: MODE_RESET (_reset, map_env.py:297-326), MODE_STEP / MODE_STEP_OBS (_step + step, :227-295,:874-915),
// MODE_OBS (get_obs & co, :917-957)
// ---------------------------------------------------------------------------------------------------------------
// NT: compile-time number of agents (0 = runtime): the per-agent readlane loops unroll completely
// Kernel arguments as ONE struct.  hipcc loads every by-value argument into SGPRs at kernel entry; the two output descriptors are
// only needed at the end of the step / in the observation phase, so they are read from the kernarg segment THERE (cold_kernarg):
// ~40 fewer live SGPRs through the move / beam / spawn phases, where the kernel otherwise spills them to VGPR lanes.
struct EnvArgs {
    DevHead hd; const DevSpec* S; DevState st; const int32_t* actions; const uint8_t* env_mask; DevTape tape; int lds_stride;
    DevStepOut so; DevObsOut oo;
};
// k_env's kernarg segment: five pointers and four ints (the preloadable head, see k_env), then the EnvArgs struct
constexpr int kEnvArgsOffset = (5 * 8 + 4 * 4 + (int)alignof(EnvArgs) - 1) / (int)alignof(EnvArgs) * (int)alignof(EnvArgs);
template <typename T>
__device__ __forceinline__ T cold_kernarg(int offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    // A typed read of the constant address space through a pointer the optimizer cannot see through: scalar-memory loads (s_load),
    // placed here, and pointer members stay pointers (the compiler then knows they are global, not flat).  Read through a CHAR
    // pointer the copy is an under-aligned access, compiled to VECTOR-memory loads followed by s_waitcnt vmcnt(0): a trip to memory
    // that also waits for every store the wave has in flight.
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                                     // loads through p cannot be hoisted above this point
    typedef const T __attribute__((address_space(4))) kt;
    T out;
    __builtin_memcpy(&out, (kt*)((const char __attribute__((address_space(4)))*)p + offset), sizeof(T));
    return out;
#else
    return T{};                                                     // host pass of the single-source compile: never called
#endif
}

// TAPE: recorded random draws (the reference-parity path; instantiated for NT = 0 only) or the counter generator
// The first nine arguments repeat what the wave needs for its first requests (the env's records, the actions, the spec, three sizes)
// as SCALAR kernel arguments in front of the struct: built with -mllvm -amdgpu-kernarg-preload-count=14 they arrive in SGPRs with
// the wave (gfx950 kernarg preload; struct arguments are not preloaded), so the state loads are requested before the first
// scalar-memory trip to the kernarg segment has returned.  Without the flag they are ordinary arguments.
#ifdef SSD_ENV_WPE      // experiment knob: tell the register allocator / scheduler how many waves per SIMD the launch really has
#define SSD_ENV_OCC __attribute__((amdgpu_waves_per_eu(SSD_ENV_WPE, SSD_ENV_WPE)))
#else
#define SSD_ENV_OCC
#endif
template <int MODE, int NT, bool TAPE, int OV = OV_ANY>
__global__ __launch_bounds__(kBlock) SSD_ENV_OCC void k_env(EnvHdr* p_hdr, uint2* p_agents, uint8_t* p_grid, const int32_t* p_actions, const DevSpec* p_spec,
                                                int p_N, int p_GS, int p_PMS, int p_lds_stride, const EnvArgs A) {
    const DevHead& hd = A.hd;
    const DevSpec* __restrict__ S = p_spec;
    DevState st;
    st.grid = p_grid; st.agents = p_agents; st.hdr = p_hdr; st.err = A.st.err; st.stamps = A.st.stamps;
    const int32_t* __restrict__ actions = p_actions;
    const uint8_t* __restrict__ env_mask = A.env_mask;
    const DevTape tape = A.tape;
    const int lds_stride = p_lds_stride;
    extern __shared__ uint4 smem[];
    const int lane = threadIdx.x & 63;
    const DevHead* h = &hd;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: per-env addresses live in SGPRs
    const int env = blockIdx.x * kWavesPerBlock + wave;
    if (env >= p_N) return;
    if (MODE == MODE_RESET && env_mask && !env_mask[env]) return;

    // Speed only (never correctness): the 4 waves that share a SIMD get distinct static priorities from their hardware
    // wave slot, so they drift apart and their 16-byte store bursts are spread over the kernel instead of hitting the
    // per-CU store path all at once at the end (profiles/: stores and compute otherwise do not overlap at all).
#ifndef SSD_NO_SETPRIO
    if (MODE == MODE_STEP_OBS || MODE == MODE_OBS) {
        const uint32_t slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 3u;   // HW_REG_HW_ID.WAVE_ID[1:0]
        if (slot == 0) __builtin_amdgcn_s_setprio(3);
        else if (slot == 1) __builtin_amdgcn_s_setprio(2);
        else if (slot == 2) __builtin_amdgcn_s_setprio(1);
    }
#endif

    Env E;
    E.S = S; E.h = h; E.lane = lane; E.n = NT ? NT : h->n; E.W = h->W; E.HW = h->HW; E.GS = p_GS;
    E.g = (uint8_t*)smem + (size_t)wave * lds_stride;
    E.occ = E.g + E.GS;
    E.pm = E.occ + E.GS;
    E.pl = E.pm + p_PMS;
    E.cbuf = E.g + lds_stride - kStampLds - lds_code_bytes(*h);
    E.stamp_lds = (unsigned long long*)(E.g + lds_stride - 256);
    E.ag = lane < E.n;
    E.err = st.err;
    E.stamps = st.stamps; E.env = env;
    const int n = E.n, GS = E.GS;
    STAMP(0);
    STAMP_REAL(st.stamps, 12);
    STAMP_HWID(st.stamps, 11);

    STAMP(16);
    uint8_t* ggrid = st.grid + (size_t)env * GS;
    // ---- issue every global load up front: counters + generator base, agents, actions, grid (or reset image), site lists ----
    // The loads land in registers and are consumed only after the last one was requested: written as `LDS[i] = global[i]` (or
    // with the counters decoded where they are loaded) the compiler waits for each group before it requests the next one --
    // four dependent trips to memory at the head of every wave's chain instead of one.  And FEW requests: the sixteen waves of a
    // CU arrive here together, and the CU's address unit takes >= 4 cycles per wave and request (16 for 16-byte ones): the env's
    // counters are one 32-byte record (lanes 0 / 1), an agent's two words one 8-byte record, a lane's eight site cells one
    // 16-byte record (DevSpec::site_t) -- 7 requests per wave where the separate arrays took 20.
    constexpr bool tape_mode = TAPE;
    uint4 hv = make_uint4(0, 0, 0, 0);
    if (lane < 2) hv = ((const uint4*)(st.hdr + env))[lane];
    uint2 ar = make_uint2(0u, 0u);
    int act = 4;
    if (E.ag && MODE != MODE_RESET) ar = st.agents[(size_t)env * n + lane];
    if (E.ag && (MODE == MODE_STEP || MODE == MODE_STEP_OBS)) act = actions[(size_t)env * n + lane];
    uint4 gv = make_uint4(0, 0, 0, 0);                           // GS <= SSD_MAX_CELLS = 64 lanes x 16 bytes: one vector per lane
    if (lane * 16 < GS) gv = MODE == MODE_RESET ? *(const uint4*)(S->reset_grid + lane * 16) : *(const uint4*)(ggrid + lane * 16);
    if (MODE != MODE_OBS) {
        const uint4 sv = *(const uint4*)S->site_t[lane];
        E.ap[0] = (int)(sv.x & 0xFFFFu); E.ap[1] = (int)(sv.x >> 16); E.ap[2] = (int)(sv.y & 0xFFFFu); E.ap[3] = (int)(sv.y >> 16);
        E.ws[0] = (int)(sv.z & 0xFFFFu); E.ws[1] = (int)(sv.z >> 16); E.ws[2] = (int)(sv.w & 0xFFFFu); E.ws[3] = (int)(sv.w >> 16);
    } else {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) E.ap[ch] = E.ws[ch] = 0;
    }
    // Every other kernel argument the load phase reads, requested in ONE batch (the compiler fetches by-value arguments from the
    // kernarg segment where they are first used: seven dependent scalar-memory trips, ~ 0.4-1 K cycles each while the segment is
    // cold, between the wave's start and its last state load otherwise).
    asm volatile("" :: "s"(h->rng_mode), "s"(h->kind), "s"(h->HW), "s"(h->W), "s"(h->n), "s"(h->n_actions), "s"(h->n_waste), "s"(h->n_apple), "s"(st.err));
    double harvest_p[4] = {0.0, 0.0, 0.0, 0.0};
    uint32_t harvest_t[4] = {0u, 0u, 0u, 0u};
    if (MODE != MODE_OBS && h->kind != SSD_ENV_CLEANUP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (TAPE) harvest_p[i] = S->harvest_p[i]; else harvest_t[i] = S->harvest_thr[i]; }
    }
    // LDS clears under the loads: the agent overlay, and the observation's padded class map (tape-mode spawns use that one as
    // scratch and say so)
    for (int i = lane * 16; i < GS; i += kWave * 16) *(uint4*)(E.occ + i) = make_uint4(0, 0, 0, 0);
    E.pm_zeroed = false;
    if (MODE == MODE_STEP_OBS || MODE == MODE_OBS) {
        for (int i = lane * 16; i < p_PMS; i += kWave * 16) *(uint4*)(E.pm + i) = make_uint4(0, 0, 0, 0);
        E.pm_zeroed = true;
    }
    STAMP(17);
    // ---- second level: table windows around the counts the env state carries (Env::w0), read by lane index after the beams / spawn ----
    const uint32_t epoch = (uint32_t)__builtin_amdgcn_readlane((int)hv.x, 1);
    const int ep_step0 = MODE == MODE_RESET ? 0 : __builtin_amdgcn_readlane((int)hv.y, 1);
    const uint32_t counts0 = (MODE == MODE_STEP || MODE == MODE_STEP_OBS) ? (uint32_t)__builtin_amdgcn_readlane((int)hv.z, 1) : 0u;
    int ep_r = (int)ar.y;
    const uint32_t rec_v = ar.x;
    E.obs_slot = MODE == MODE_STEP_OBS ? ep_step0 + 1 : ep_step0;
    E.w0 = E.a0 = -1; E.pf_ta = E.pf_tw = E.pf_fl = 0u; E.pf_den = 0.f;
    if ((MODE == MODE_STEP || MODE == MODE_STEP_OBS) && counts0 != 0xFFFFFFFFu) {
        if (!TAPE && h->kind == SSD_ENV_CLEANUP) {
            E.w0 = h->n_waste > 0 ? (int)(counts0 >> 16) : 0;
            const int wi = E.w0 - lane;
            if (wi >= 0 && wi <= SSD_MAX_SITES) { const uint4 tv = *(const uint4*)S->tab_thr[wi]; E.pf_ta = tv.x; E.pf_tw = tv.y; E.pf_fl = tv.z; }
        }
        E.a0 = (int)(counts0 & 0xFFFFu);
        const int ai = E.a0 - 8 + lane;
        if (ai >= 0 && ai <= h->HW) E.pf_den = S->tab_den[ai];
    }
    // ---- consume ----
    if (lane * 16 < GS) *(uint4*)(E.g + lane * 16) = gv;
    STAMP(18);

    Rng R;
    R.tape = tape_mode;
    R.ustride = tape.ustride;
    R.tape_u = R.tape ? tape.uniforms + (size_t)env * tape.ustride : nullptr;
    R.err = st.err;
    R.b[0] = R.b[1] = R.b[2] = R.b[3] = 0;
    if (MODE == MODE_RESET && !R.tape) {
        // a new episode: derive its Philox base (once per episode, scalar unit) and keep it in the env state
        philox4(0u, 0u, h->env_id_base + (uint32_t)env, epoch + 1u, h->seed_lo, h->seed_hi, R.b);
        if (lane == 0) *(uint4*)st.hdr[env].rng_base = make_uint4(R.b[0], R.b[1], R.b[2], R.b[3]);
    } else if ((MODE == MODE_STEP || MODE == MODE_STEP_OBS) && !R.tape) {
        const uint32_t call = ((uint32_t)(ep_step0 + 1)) << 16;      // call index inside the episode (reset = 0)
        R.b[0] = __builtin_amdgcn_readfirstlane(hv.x) ^ call; R.b[1] = __builtin_amdgcn_readfirstlane(hv.y) ^ call;
        R.b[2] = __builtin_amdgcn_readfirstlane(hv.z) ^ call; R.b[3] = __builtin_amdgcn_readfirstlane(hv.w) ^ call;
    }
    const uint8_t* tape_order = tape.move_order ? tape.move_order + (size_t)env * n : nullptr;
    const uint8_t* tape_waste = tape.waste_order ? tape.waste_order + (size_t)env * h->n_waste : nullptr;

    int spawn_p = 0;
    if (MODE == MODE_RESET && h->random_spawn) {
        // random_spawn_point: every agent's spawn_point() first shuffles the spawn list (map_env.py:776-777), then takes the LAST
        // free point of that order = the free point with the largest (position in the order, id).  Lanes = spawn point ids; the
        // agents are served one after the other (whole wave, reset only).
        // The list holds every point once (Harvest) or twice (Cleanup, cleanup.py:79-80).  Lanes = list elements.
        const int ns = h->n_spawn, len = h->spawn_len;
        const bool sl = lane < len;
        for (int a = 0; a < n; ++a) {
            int pid;                                             // the point this lane's element stands for
            uint32_t key;                                        // its rank in the shuffled list (any order-preserving key)
            if (R.tape) {                                        // lane = position in the recorded list
                pid = sl ? (int)tape.spawn_order[((size_t)env * n + a) * len + lane] : 0;
                if (pid >= ns) { atomicOr(R.err, ERR_BAD_TAPE); pid = 0; }
                key = (uint32_t)lane;
            } else {                                             // lane = element e = copy * ns + id, sorted by (x >> 8, e)
                pid = lane >= ns ? lane - ns : lane;
                key = ((R.u32(SSD_STREAM_SPAWN_ROT, 256u + 32u * (uint32_t)a + (uint32_t)lane) >> 8) << 8) | (uint32_t)lane;
            }
            // occupancy is per POINT: a lane is a candidate while no earlier agent took its point
            bool occupied = false;
            for (int b = 0; b < a; ++b) occupied |= __builtin_amdgcn_readlane(spawn_p, b) == (sl ? (int)S->spawn_all[pid] : -1);
            const uint32_t cand = (sl && !occupied) ? key + 1u : 0u;                               // 0 = not a candidate
            const uint32_t best = ~wave_min_u32(~cand);                                            // wave max
            const bool me = cand != 0u && cand == best;
            const uint64_t who = ballot(me);
            const int sel = who ? first_lane(who) : 0;
            const int cell = __builtin_amdgcn_readlane(sl ? (int)S->spawn_all[pid] : 0, sel);
            if (lane == a) spawn_p = cell;
        }
    }
    if (E.ag) {
        if (MODE == MODE_RESET) {
            // setup_agents: agent a takes the last spawn point still free (map_env.py:771-784) = spawn_cell[a] without the shuffle;
            // spawn_rotation (:786-793)
            E.P = h->random_spawn ? spawn_p : (int)S->spawn_cell[lane];
            if (h->spawn_rotation >= 0) E.O = h->spawn_rotation;
            else if (R.tape) E.O = tape.spawn_rot[(size_t)env * n + lane] & 3;
            else E.O = (int)(R.u32(SSD_STREAM_SPAWN_ROT, (uint32_t)lane) >> 30);
        } else {
            E.P = (int)(rec_v & 0xFF) * E.W + (int)((rec_v >> 8) & 0xFF);
            E.O = (int)((rec_v >> 16) & 3);
        }
        if (MODE == MODE_STEP || MODE == MODE_STEP_OBS) {
            if ((unsigned)act >= (unsigned)h->n_actions) { atomicOr(st.err, ERR_BAD_ACTION); act = 4; }  // KeyError in action_map
        }
    } else {
        E.P = -1 - lane; E.O = 0;
    }
    wsync();
    STAMP(1);

    int n_draws = 0;
    if (MODE == MODE_RESET) {
        if (E.ag) E.occ[E.P] = (uint8_t)agent_char(lane);   // spawn cells are distinct
        wsync();
        int n_waste_cells = h->kind == SSD_ENV_CLEANUP ? h->n_waste : 0;       // custom_reset: all waste present / all apples grown
        int n_apple_cells = h->kind == SSD_ENV_CLEANUP ? 0 : h->n_apple;
        n_draws = h->kind == SSD_ENV_CLEANUP ? spawn_cleanup<TAPE>(E, R, tape_waste, n_waste_cells, n_apple_cells)
                                             : spawn_harvest(E, R, n_apple_cells, harvest_p, harvest_t);             // map_env.py:313
        ep_r = 0;
        if (lane == 0) {
            *(uint4*)&st.hdr[env].epoch = make_uint4(epoch + 1, 0u, ((uint32_t)n_waste_cells << 16) | (uint32_t)n_apple_cells, 0u);
            int32_t* nd = cold_kernarg<int32_t*>((int)(kEnvArgsOffset + offsetof(EnvArgs, so) + offsetof(DevStepOut, n_draws)));
            if (nd) nd[env] = n_draws;
            if (R.tape && n_draws > R.ustride) atomicOr(st.err, ERR_TAPE_OVERRUN);
        }
    }
    if (MODE == MODE_STEP || MODE == MODE_STEP_OBS) {
        int reward = 0, cleaned = 0;
        // waste / apple cell counts of the grid: carried in the env state and updated by what this step changes
        // (recounted when the state was imported)
        int n_waste_cells, n_apple_cells;
        if (counts0 == 0xFFFFFFFFu) { n_waste_cells = count_cells(E, C_WASTE, false); n_apple_cells = count_cells(E, C_APPLE, false); }
        else { n_waste_cells = (int)(counts0 >> 16); n_apple_cells = (int)(counts0 & 0xFFFFu); }
        move_phase(E, act, R, tape_order);                                       // map_env.py:251
        STAMP(2);
        // consume (map_env.py:253-256, agent.py:195-201,250-256) runs in id order, so only the lowest id on a cell eats;
        // the overlay of get_map_with_agents (map_env.py:360-379) keeps the highest id on a cell
        {
            bool lower = false, higher = false;
            for (int b = 0; b < n; ++b) { const int pb = rl(E.P, b); lower |= (b < lane && pb == E.P); higher |= (b > lane && pb == E.P); }
            const bool eats = E.ag && !lower && E.g[E.P] == C_APPLE;
            if (eats) { reward += 1; E.g[E.P] = C_EMPTY; }
            n_apple_cells -= popc64(ballot(eats));
            if (E.ag && !higher) E.occ[E.P] = (uint8_t)agent_char(lane);
        }
        wsync();
        STAMP(3);
        // update_custom_moves (map_env.py:663-673): sequential over agents, the map is updated after each one
        {
            const uint64_t fire = ballot(E.ag && act >= 7);
            for (uint64_t m = fire; m; m &= m - 1) {
                const int f = first_lane(m);
                const int af = rl(act, f);
                if (h->kind == SSD_ENV_CLEANUP && af == 8) {                      // CLEAN (cleanup.py:135-143)
                    const int c = clean_beams(E, f);
                    if (lane == f) cleaned = c;
                    n_waste_cells -= c;
                } else if (lane == f) reward -= 1;                                // fire_beam('F') (agent.py:188-190,239-241)
            }
        }
        STAMP(4);
        n_draws = h->kind == SSD_ENV_CLEANUP ? spawn_cleanup<TAPE>(E, R, tape_waste, n_waste_cells, n_apple_cells)
                                             : spawn_harvest(E, R, n_apple_cells, harvest_p, harvest_t);             // map_env.py:263
        STAMP(5);
        // scalars (map_env.py:291-292, 883-914).  After the consume loop no agent stands on an apple and nothing spawns
        // under an agent, so the apples visible in map_with_agents are all apples of the grid.
        const int apples = n_apple_cells;
        // host-tabulated fp64 quotient for every count 0..H*W (imported grids too): from the window requested with the state loads
        const int dslot = apples - E.a0 + 8;                      // (wave-uniform)
        // (outside the window: the same correctly rounded fp64 quotient computed here -- NOT a load: a load that may be pending makes
        // every later use of its register wait for vmcnt(0), i.e. for all the stores of this phase)
        const float den = (E.a0 >= 0 && (unsigned)dslot < (unsigned)kWave) ? __int_as_float(rl(__float_as_int(E.pf_den), dslot))
                                                                          : (float)((double)apples / (double)E.HW);
        ep_r += reward;
        const int step = ep_step0 + 1;
        const bool term = step >= h->episode_limit;
        STAMP(19);
        const DevStepOut so = cold_kernarg<DevStepOut>((int)(kEnvArgsOffset + offsetof(EnvArgs, so)));
        STAMP(20);
        if (E.ag) {
            const size_t o = (size_t)env * n + lane;
            if (so.reward) so.reward[o] = (float)reward;
            if (so.clean_num) so.clean_num[o] = (float)cleaned;
            if (so.apple_den) so.apple_den[o] = den;
        }
        if (term && (so.collective || so.equality)) {
            double sum = 0, asum = 0, diff = 0;
            for (int b = 0; b < n; ++b) {
                const double rb = (double)rl(ep_r, b);
                sum += rb; asum += fabs(rb);
                for (int c = 0; c < n; ++c) diff += fabs(rb - (double)rl(ep_r, c));
            }
            const double eq = sum != 0 ? 1 - diff / (2 * n * asum) : 1.0;
            if (lane == 0) { if (so.collective) so.collective[env] = (float)sum; if (so.equality) so.equality[env] = (float)eq; }
        }
        if (lane == 0) {
            *(uint4*)&st.hdr[env].epoch = make_uint4(epoch, (uint32_t)step, ((uint32_t)n_waste_cells << 16) | (uint32_t)n_apple_cells, 0u);
            if (so.terminated) so.terminated[env] = term ? 1 : 0;
            if (so.n_draws) so.n_draws[env] = n_draws;
            if (R.tape && n_draws > R.ustride) atomicOr(st.err, ERR_TAPE_OVERRUN);
        }
    }
    if (MODE == MODE_OBS) {
        bool higher = false;
        for (int b = 0; b < n; ++b) { const int pb = rl(E.P, b); higher |= (b > lane && pb == E.P); }
        if (E.ag && !higher) E.occ[E.P] = (uint8_t)agent_char(lane);
        wsync();
    }
    STAMP(6);

    if (MODE != MODE_OBS) {
        // ---- write the state back ----
        for (int i = lane * 16; i < GS; i += kWave * 16) *(uint4*)(ggrid + i) = *(const uint4*)(E.g + i);
        if (E.ag) {
            const int pr = (int)udiv((uint32_t)E.P, h->magic_W), pc = E.P - pr * E.W;
            st.agents[(size_t)env * n + lane] = make_uint2((uint32_t)pr | ((uint32_t)pc << 8) | ((uint32_t)E.O << 16), (uint32_t)ep_r);
        }
    }
    STAMP(7);
    if (MODE == MODE_STEP_OBS || MODE == MODE_OBS) {
        const DevObsOut oo = cold_kernarg<DevObsOut>((int)(kEnvArgsOffset + offsetof(EnvArgs, oo)));
        if (oo.t_slots > 0 && E.obs_slot >= oo.t_slots) {
            // stepped past the episode storage: never write into the next env's block -- the observation lands in this env's LAST
            // slot and the sticky error bit tells the caller (ssd_poll_error)
            E.obs_slot = oo.t_slots - 1;
            if (lane == 0) atomicOr(st.err, ERR_SLOT_OVERRUN);
        }
        if (OV != OV_ANY) observe_phase<false, NT, OV>(E, env, oo);
        else if (h->obs_color == SSD_COLOR_FULL) observe_phase<true, NT, OV_ANY>(E, env, oo);
        else observe_phase<false, NT, OV_ANY>(E, env, oo);
    }
    STAMP(10);
    STAMP_REAL(st.stamps, 13);
    STAMP_FLUSH(st.stamps);
}

void launch_env(int mode, const DevSpec* spec, const DevSpec& hs, DevState st, const int32_t* actions,
                const uint8_t* env_mask, DevTape tape, DevStepOut so, DevObsOut oo, hipStream_t stream) {
    const int blocks = (hs.N + kWavesPerBlock - 1) / kWavesPerBlock;
    const int stride = lds_per_wave(hs);
    const size_t lds = (size_t)stride * kWavesPerBlock;
    EnvArgs A;
    A.hd = (const DevHead&)hs; A.S = spec; A.st = st; A.actions = actions; A.env_mask = env_mask; A.tape = tape; A.lds_stride = stride;
    A.so = so; A.oo = oo;
#define SSD_ENV_ARGS st.hdr, st.agents, st.grid, actions, spec, (int)hs.N, (int)hs.GS, (int)hs.PMS, stride, A
#define SSD_LAUNCH(M, NT_, TP_) hipLaunchKernelGGL((k_env<M, NT_, TP_>), dim3(blocks), dim3(kBlock), lds, stream, SSD_ENV_ARGS)
#define SSD_LAUNCH_N(M)                                                                                   \
    do {                                                                                                  \
        if (tape_mode) SSD_LAUNCH(M, 0, true);                                                            \
        else if (hs.n == 5) SSD_LAUNCH(M, 5, false); else if (hs.n == 10) SSD_LAUNCH(M, 10, false);       \
        else if (hs.n == 3) SSD_LAUNCH(M, 3, false); else SSD_LAUNCH(M, 0, false);                        \
    } while (0)
    const bool tape_mode = hs.rng_mode == SSD_RNG_TAPE;
    switch (mode) {
        case MODE_RESET: if (tape_mode) SSD_LAUNCH(MODE_RESET, 0, true); else SSD_LAUNCH(MODE_RESET, 0, false); break;
        case MODE_STEP: SSD_LAUNCH_N(MODE_STEP); break;
        case MODE_STEP_OBS: {
            // the two requests with an instantiation of their own (observe_phase): simplified palette, windows only
            const bool plain = !tape_mode && hs.obs_color != SSD_COLOR_FULL && oo.obs && !oo.state && (hs.n == 5 || hs.n == 10);
            const int ov = !plain ? OV_ANY : oo.fmt == SSD_OBS_CODE ? OV_CODE : (oo.fmt == SSD_OBS_F32 && !oo.code) ? OV_F32 : OV_ANY;
            if (ov == OV_CODE) { if (hs.n == 5) hipLaunchKernelGGL((k_env<MODE_STEP_OBS, 5, false, OV_CODE>), dim3(blocks), dim3(kBlock), lds, stream, SSD_ENV_ARGS);
                                 else hipLaunchKernelGGL((k_env<MODE_STEP_OBS, 10, false, OV_CODE>), dim3(blocks), dim3(kBlock), lds, stream, SSD_ENV_ARGS); }
            else if (ov == OV_F32) { if (hs.n == 5) hipLaunchKernelGGL((k_env<MODE_STEP_OBS, 5, false, OV_F32>), dim3(blocks), dim3(kBlock), lds, stream, SSD_ENV_ARGS);
                                     else hipLaunchKernelGGL((k_env<MODE_STEP_OBS, 10, false, OV_F32>), dim3(blocks), dim3(kBlock), lds, stream, SSD_ENV_ARGS); }
            else SSD_LAUNCH_N(MODE_STEP_OBS);
            break;
        }
        default:                                                      // observations draw nothing
            if (hs.n == 5) SSD_LAUNCH(MODE_OBS, 5, false); else if (hs.n == 10) SSD_LAUNCH(MODE_OBS, 10, false);
            else if (hs.n == 3) SSD_LAUNCH(MODE_OBS, 3, false); else SSD_LAUNCH(MODE_OBS, 0, false);
            break;
    }
#undef SSD_LAUNCH_N
#undef SSD_LAUNCH
#undef SSD_ENV_ARGS
}

// ---------------------------------------------------------------------------------------------------------------
// state export / import (parity tests, KATs, warm starts)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_export(DevSpec const* S, DevState st, ssd_state d) {
    const DevHead* h = S;
    const int N = h->N, n = h->n, HW = h->HW, GS = h->GS;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    if (d.grid) for (size_t i = tid; i < (size_t)N * HW; i += nt) { size_t e = i / HW; d.grid[i] = st.grid[e * GS + (i - e * HW)]; }
    for (size_t i = tid; i < (size_t)N * n; i += nt) {
        const uint2 ar = st.agents[i];
        const uint32_t rec = ar.x;
        if (d.pos) { d.pos[2 * i] = (int16_t)(rec & 0xFF); d.pos[2 * i + 1] = (int16_t)((rec >> 8) & 0xFF); }
        if (d.orient) d.orient[i] = (uint8_t)((rec >> 16) & 3);
        if (d.ep_reward) d.ep_reward[i] = (int32_t)ar.y;
    }
    for (size_t i = tid; i < (size_t)N; i += nt) {
        if (d.ep_step) d.ep_step[i] = st.hdr[i].ep_step;
        if (d.epoch) d.epoch[i] = st.hdr[i].epoch;
    }
}

__global__ void k_import(DevSpec const* S, DevState st, ssd_state s) {
    const DevHead* h = S;
    const int N = h->N, n = h->n, HW = h->HW, GS = h->GS;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    if (s.grid) for (size_t i = tid; i < (size_t)N * HW; i += nt) {
        size_t e = i / HW;
        uint8_t c = s.grid[i];
        st.grid[e * GS + (i - e * HW)] = c <= C_STREAM ? c : (uint8_t)C_EMPTY;
    }
    for (size_t i = tid; i < (size_t)N * n; i += nt) {
        uint2 ar = st.agents[i];
        uint32_t rec = ar.x;
        if (s.pos) {
            int r = s.pos[2 * i], c = s.pos[2 * i + 1];
            r = r < 0 ? 0 : r >= h->H ? h->H - 1 : r; c = c < 0 ? 0 : c >= h->W ? h->W - 1 : c;
            rec = (rec & ~0xFFFFu) | (uint32_t)r | ((uint32_t)c << 8);
        }
        if (s.orient) rec = (rec & ~0x30000u) | ((uint32_t)(s.orient[i] & 3) << 16);
        ar.x = rec;
        if (s.ep_reward) ar.y = (uint32_t)s.ep_reward[i];
        st.agents[i] = ar;
    }
    for (size_t i = tid; i < (size_t)N; i += nt) {
        if (s.ep_step) st.hdr[i].ep_step = s.ep_step[i];
        if (s.epoch) {
            st.hdr[i].epoch = s.epoch[i];
            philox4(0u, 0u, h->env_id_base + (uint32_t)i, s.epoch[i], h->seed_lo, h->seed_hi, st.hdr[i].rng_base);
        }
        if (s.grid || s.pos) st.hdr[i].counts = 0xFFFFFFFFu;   // imported grid / agents: recount at the next step
    }
}

void launch_export(const DevSpec* spec, const DevSpec& hs, DevState st, ssd_state dst, hipStream_t stream) {
    size_t work = (size_t)hs.N * hs.HW;
    int blocks = (int)((work + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_export, dim3(blocks), dim3(256), 0, stream, spec, st, dst);
}
void launch_import(const DevSpec* spec, const DevSpec& hs, DevState st, ssd_state src, hipStream_t stream) {
    size_t work = (size_t)hs.N * hs.HW;
    int blocks = (int)((work + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_import, dim3(blocks), dim3(256), 0, stream, spec, st, src);
}

}  // namespace ssd
