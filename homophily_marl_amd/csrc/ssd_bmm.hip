// ssd_bmm.hip -- the learner's per-agent affine layers y[g] = b[g] + x[g] w[g] and their backward (include/ssd_hip.h:
// ssd_bias_bmm_fwd / _bwd).
//
// Reference: HomophilyAgent keeps one weight set per agent and applies fc1 / the GRU input projections / the dueling heads with
// th.baddbmm over the agent axis (homophily_agent.py:154-208); the learner evaluates them for every (episode, timestep) row of the
// sampled batch (homophily_learner.py:68-91).  At the learner's sizes (n = 5 weight sets, R = 16 x 101 rows, 64..80 inputs,
// 1..192 outputs) these are 0.01-0.2 GFLOP products: a library batched GEMM spends 15-20 us on each (24 of them per train step,
// forward and backward) where the arithmetic is ~1 us.  Here every product is one launch of exact-f32 MFMAs
// (v_mfma_f32_16x16x4_f32) with the whole chip busy:
//   forward   one wave per (weight set, 16-row tile, group of output tiles): x rows come in as 16-byte loads that feed four K-steps,
//             the weights straight from L2 (each element is used once per wave: no staging), bias as the accumulator's initial value;
//   backward  ONE launch for all three gradients: dx = g w^T in the forward's form (both operands as 16-byte loads along the
//             output axis), dw = x^T g with the row axis as K split over the 16 waves of a workgroup and added in LDS in a fixed
//             order, db = column sums of g taken from the operands the dw waves load anyway.  Deterministic: no atomics.
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>

namespace ssd {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte load at any dword address

struct BmmK {
    const float *x, *w, *b, *g;
    float *y, *dx, *dw, *db;
    const float* slope_of;     // backward, nullable: dx is multiplied by LeakyReLU'(.) taken from the sign of slope_of[g][row][i]
    int n, R, I, O;
    long x_set, g_set;         // elements between two weight sets of x / g (default R * I, R * O; larger when the rows are a slice)
    const float* act_y;        // backward of the LEAKY forward: its output y [n, R, O]; g is multiplied by LeakyReLU'(.) from y's sign where it is loaded
    int leaky;                 // forward: y = LeakyReLU(b + x w) (nn.LeakyReLU default slope 0.01)
    // two-source rows (the incentive head's [h_i | other_j] rows without materialising them): columns [0, I1) of row r come from
    // x [n, R / x1_div, I1] at row r / x1_div, columns [I1, I) from x2 at row r (x2_set elements between weight sets: 0 = shared)
    const float* x2; int I1, x1_div; long x2_set;
    uint32_t x1_magic;         // ceil(2^32 / x1_div) (0 when x1_div == 1): r / x1_div == __umulhi(r, x1_magic) for r < 2^28
    long w_set;                // elements between two weight sets of w (default I * O; larger when w is the leading rows of a wider layer)
    float* part;               // backward with row_chunks > 1: per (set, tile, chunk) partial dw tile [64 lanes x 4] + db [16] (BMM_PART floats)
    int row_chunks;            // the rows (K of dw) are cut into this many chunks, one workgroup each; k_bmm_dw_reduce adds them in order
};
constexpr int BMM_PART = 272;                    // floats per partial record: 256 (dw tile) + 16 (db)

// base[idx] with the BYTE offset formed in 32 bits: the load takes the uniform base from scalar registers and one VGPR of offset
// (no 64-bit multiply-add per lane, and no load destination doubling as the dead half of a 64-bit address temporary)
__device__ __forceinline__ float ld32(const float* base, uint32_t idx) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(base) + (size_t)(idx * 4u));
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// The labelled reduced-precision variant of the learner's GEMMs (learner_dtype: bf16; ssd_set_learner_precision(1)): operands rounded
// to bf16 (round to nearest even), ONE v_mfma_f32_16x16x32_bf16 per 32 reduction indices, f32 accumulation.  A lane's eight
// k-values are the 2 x 4 (dx / forward) or 8 x 1 (dw) values the f32 kernels feed to eight consecutive 16x16x4 steps -- the loads, the
// masks and the order of the chunks are those of the f32 path; the parameters, the optimiser and the loss stay f32.
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ f32x4 mfma32b(const float (&a)[8], const float (&b)[8], f32x4 c) {
    bf8 x, y;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)a[j]; y[j] = (__bf16)b[j]; }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
}
static int g_learner_precision = 2;            // 2: f32 (exact-f32 MFMAs; two- / three-term splits in the recurrence), 1: single bf16 products
int learner_precision() { return g_learner_precision; }
void set_learner_precision(int p) { g_learner_precision = p == 1 ? 1 : 2; }

// 4 consecutive floats p[k0 .. k0 + 3] of a row of `len` floats; elements at or past len read as 0
__device__ __forceinline__ f32x4 load4(const float* p, int k0, int len) {
    if (k0 + 4 <= len) return *reinterpret_cast<const f32x4_u*>(p + k0);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (k0 + r < len) v[r] = p[k0 + r];
    return v;
}

// y[g][row][o] = b[g][o] + sum_i x[g][row][i] w[g][i][o].  A wave owns one unit = (16-row tile, group of up to BMM_TPW output tiles);
// units are dealt to the waves of the grid in order (grid (ceil(units / 4), n), 256 threads).  D[row 4 q + r][col m]; K-step
// (chunk c, r) uses k = 16 c + 4 q + r for lane quarter q.
// XV: I is a multiple of 4, a lane's four x values of a chunk come as one 16-byte load (no chunk straddles the end of a row).
constexpr int BMM_TPW = 3;                      // output tiles per wave
template <bool XV, bool BF = false>
__global__ __launch_bounds__(256) void k_bias_bmm_fwd(BmmK a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int R = a.R, I = a.I, O = a.O;
    const int otiles = (O + 15) >> 4, groups = (otiles + BMM_TPW - 1) / BMM_TPW, rtiles = (R + 15) >> 4;
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= rtiles * groups) return;
    const int g = blockIdx.y, row0 = (unit / groups) * 16, t0 = (unit % groups) * BMM_TPW;
    const int rowc = row0 + m < R ? row0 + m : R - 1;
    const bool two = a.x2 != nullptr;                                  // (uniform) two-source rows, see BmmK
    const int I1 = two ? a.I1 : I;
    const float* xs = a.x + (two ? (size_t)g * (size_t)(R / a.x1_div) * I1 : (size_t)g * R * I);      // the weight set's rows (uniform)
    const uint32_t xrow = two ? (uint32_t)(rowc / a.x1_div) * (uint32_t)I1 : (uint32_t)rowc * (uint32_t)I;   // this lane's row starts at xrow
    const float* x2s = two ? a.x2 + (size_t)g * a.x2_set : a.x;
    const uint32_t x2row = (uint32_t)rowc * (uint32_t)(I - I1);
    const float* wg = a.w + (size_t)g * (a.w_set ? a.w_set : (long)I * O);
    f32x4 acc[BMM_TPW];
#pragma unroll
    for (int t = 0; t < BMM_TPW; ++t) {
        const int o = 16 * (t0 + t) + m;
        const float bv = (t0 + t < otiles && o < O) ? a.b[(size_t)g * O + o] : 0.f;
        acc[t] = f32x4{bv, bv, bv, bv};
    }
    // Loads never branch (a predicated load makes hipcc wait for everything in flight where the branch joins): addresses are clamped
    // into the operand and the value is masked when it is used; chunk c + 1 is requested before chunk c is multiplied.
    const int chunks = (I + 15) >> 4;
    int oc[BMM_TPW]; bool on[BMM_TPW];
#pragma unroll
    for (int t = 0; t < BMM_TPW; ++t) {
        const int o = 16 * (t0 + t) + m;
        on[t] = t0 + t < otiles && o < O; oc[t] = on[t] ? o : 0;
    }
    struct Chunk { float x[4]; float w[BMM_TPW][4]; };
    auto fetch = [&](int c, Chunk& d) {
        const int k0 = 16 * c + 4 * q;
        if constexpr (XV) {
            const int kc = k0 < I ? k0 : I - 4;
            const bool second = 16 * c >= I1;                          // (uniform: I1 is a multiple of 16) this chunk lies in the second source
            const float* base = second ? x2s : xs;
            const uint32_t off = second ? x2row + (uint32_t)(kc - I1) : xrow + (uint32_t)kc;
            const f32x4 v = *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const uint8_t*>(base) + (size_t)(off * 4u));
#pragma unroll
            for (int r = 0; r < 4; ++r) d.x[r] = v[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + r < I ? k0 + r : I - 1;
            if constexpr (!XV) {
                const bool sec = k >= I1;                              // (per lane: any width of the second source)
                d.x[r] = ld32(sec ? x2s : xs, sec ? x2row + (uint32_t)(k - I1) : xrow + (uint32_t)k);
            }
#pragma unroll
            for (int t = 0; t < BMM_TPW; ++t) d.w[t][r] = ld32(wg, (uint32_t)k * (uint32_t)O + (uint32_t)oc[t]);
        }
    };
    auto use = [&](int c, const Chunk& d) {
        const int k0 = 16 * c + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool kon = k0 + r < I;
            const float xv = kon ? d.x[r] : 0.f;
#pragma unroll
            for (int t = 0; t < BMM_TPW; ++t)
                if (t0 + t < otiles) acc[t] = mfma4(xv, (kon && on[t]) ? d.w[t][r] : 0.f, acc[t]);
        }
    };
    // bf16: chunks c and c + 1 (this lane's k = 16 c + 4 q + r and 16 (c + 1) + 4 q + r) are the 8 reduction indices of ONE K = 32 MFMA
    auto use_pair = [&](int c, const Chunk& d0, const Chunk& d1, bool two) {
        const int k0 = 16 * c + 4 * q;
        float xv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { xv[r] = k0 + r < I ? d0.x[r] : 0.f; xv[4 + r] = (two && k0 + 16 + r < I) ? d1.x[r] : 0.f; }
#pragma unroll
        for (int t = 0; t < BMM_TPW; ++t) {
            if (t0 + t >= otiles) continue;
            float wv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { wv[r] = (k0 + r < I && on[t]) ? d0.w[t][r] : 0.f; wv[4 + r] = (two && k0 + 16 + r < I && on[t]) ? d1.w[t][r] : 0.f; }
            acc[t] = mfma32b(xv, wv, acc[t]);
        }
    };
    Chunk ca, cb;
    fetch(0, ca);
    for (int c = 0; c < chunks; c += 2) {
        if (c + 1 < chunks) fetch(c + 1, cb);
        if constexpr (BF) {
            Chunk cur = ca;
            if (c + 2 < chunks) fetch(c + 2, ca);
            use_pair(c, cur, cb, c + 1 < chunks);
        } else {
            use(c, ca);
            if (c + 1 < chunks) {
                if (c + 2 < chunks) fetch(c + 2, ca);
                use(c + 1, cb);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < BMM_TPW; ++t) {
        const int o = 16 * (t0 + t) + m;
        if (t0 + t >= otiles || o >= O) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * q + r;
            const float v = acc[t][r];
            if (row < R) a.y[((size_t)g * R + row) * O + o] = (a.leaky && !(v > 0.f)) ? 0.01f * v : v;
        }
    }
}

// One launch, grid (DW + DX, n): blocks [0, DW) compute one 16 x 16 tile of dw = x^T g (rows of the batch = K, a sixteenth per wave,
// partials added in LDS in wave order) and, for input tile 0, the matching 16 entries of db = column sums of g -- the long chains
// of the launch (up to 127 K-steps of 4 rows per wave), so they are dispatched first; blocks [DW, DW + DX) compute dx = g w^T, one
// unit = (16-row tile, 16-input tile) per wave, units dealt to the waves in order.
// A dw tile streams its two 64-byte column slabs of ALL rows through one CU's L1 (the cache-line rate of that L1 is the bound: 22 us
// for 8 080 rows), and jobs with few tiles (the inc head's [80, 3] layer: 5 tiles x 5 sets) leave most CUs idle.  With row_chunks > 1
// the rows are cut into chunks, DW = tiles x chunks workgroups write partial tiles, and a second, tiny launch (k_bmm_dw_reduce) adds
// the chunks in order: deterministic, no hand-off between workgroups of one launch.
__device__ __forceinline__ int gridDim_tiles(const BmmK& a) { return ((a.I + 15) >> 4) * ((a.O + 15) >> 4); }
// adds the row chunks of every dw tile in chunk order: grid (tiles, n), one wave
__global__ __launch_bounds__(64) void k_bmm_dw_reduce(BmmK a) {
    const int lane = threadIdx.x, m = lane & 15, q = lane >> 4, tile = blockIdx.x, g = blockIdx.y, C = a.row_chunks;
    const int otiles = (a.O + 15) >> 4, it = tile / otiles, ot = tile - it * otiles, o = 16 * ot + m;
    const float* P = a.part + ((size_t)((size_t)g * gridDim.x + tile) * C) * BMM_PART;
    f32x4 sum = *reinterpret_cast<const f32x4*>(P + 4 * lane);
    float sb = q == 0 ? P[256 + m] : 0.f;
    for (int c = 1; c < C; ++c) {
        sum += *reinterpret_cast<const f32x4*>(P + (size_t)c * BMM_PART + 4 * lane);
        if (q == 0) sb += P[(size_t)c * BMM_PART + 256 + m];
    }
    if (o >= a.O) return;
    if (a.dw) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ii = 16 * it + 4 * q + r;
            if (ii < a.I) a.dw[((size_t)g * a.I + ii) * a.O + o] = sum[r];
        }
    }
    if (a.db && it == 0 && q == 0) a.db[(size_t)g * a.O + o] = sb;
}
constexpr int BMM_BWD_WAVES = 16;               // the row axis (K of dw: up to T B n = 8080 rows) is split 16 ways
// OV: O is a multiple of 4: 16-byte operand loads along the output axis in the dx part.  ACT: the layer's forward applied LeakyReLU
// (a.act_y = its output): every g value is multiplied by the slope at its element as it is loaded (dx, dw and db all see g slope).
template <bool OV, bool ACT, bool BF = false>
__global__ __launch_bounds__(BMM_BWD_WAVES * 64) void k_bias_bmm_bwd(BmmK a, int dw_blocks) {
    __shared__ f32x4 red[BMM_BWD_WAVES][64];
    __shared__ float redb[BMM_BWD_WAVES][16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int g = blockIdx.y;
    const int R = a.R, I = a.I, O = a.O;
    const int itiles = (I + 15) >> 4, otiles = (O + 15) >> 4;
    const float* gg = a.g + (size_t)g * a.g_set;
    if ((int)blockIdx.x >= dw_blocks) {
        const int unit = (blockIdx.x - dw_blocks) * BMM_BWD_WAVES + wave, rtiles = (R + 15) >> 4;
        if (unit >= rtiles * itiles) return;
        const int row0 = (unit / itiles) * 16, i = 16 * (unit % itiles) + m;
        const int rowc = row0 + m < R ? row0 + m : R - 1;
        const float* ws = a.w + (size_t)g * (a.w_set ? a.w_set : (long)I * O);
        const uint32_t grow = (uint32_t)rowc * (uint32_t)O, wrow = (uint32_t)(i < I ? i : I - 1) * (uint32_t)O;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int chunks = (O + 15) >> 4;
        // reduction index = output feature; branch-free loads (clamped, masked at use), chunk c + 1 requested before chunk c is used
        struct Pair { f32x4 g, w, y; };
        const float* yy = ACT ? a.act_y + (size_t)g * a.g_set : nullptr;
        auto fetch = [&](int c, Pair& d) {
            const int k0 = 16 * c + 4 * q;
            if constexpr (OV) {
                const uint32_t kb = (uint32_t)(k0 < O ? k0 : O - 4);
                d.g = *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const uint8_t*>(gg) + (size_t)((grow + kb) * 4u));
                d.w = *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const uint8_t*>(ws) + (size_t)((wrow + kb) * 4u));
                if constexpr (ACT) d.y = *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const uint8_t*>(yy) + (size_t)((grow + kb) * 4u));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t k = (uint32_t)(k0 + r < O ? k0 + r : O - 1);
                    d.g[r] = ld32(gg, grow + k); d.w[r] = ld32(ws, wrow + k);
                    if constexpr (ACT) d.y[r] = ld32(yy, grow + k);
                }
            }
        };
        auto use = [&](int c, const Pair& d) {
            const int k0 = 16 * c + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool on = k0 + r < O;
                float gv = on ? d.g[r] : 0.f;
                if constexpr (ACT) gv = d.y[r] > 0.f ? gv : 0.01f * gv;
                acc = mfma4(gv, on ? d.w[r] : 0.f, acc);
            }
        };
        auto use_pair = [&](int c, const Pair& d0, const Pair& d1, bool two) {      // bf16: two chunks = the 8 reduction indices of one K = 32 MFMA
            const int k0 = 16 * c + 4 * q;
            float gv[8], wv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool on0 = k0 + r < O, on1 = two && k0 + 16 + r < O;
                float g0 = on0 ? d0.g[r] : 0.f, g1 = on1 ? d1.g[r] : 0.f;
                if constexpr (ACT) { g0 = d0.y[r] > 0.f ? g0 : 0.01f * g0; g1 = (on1 && !(d1.y[r] > 0.f)) ? 0.01f * g1 : g1; }
                gv[r] = g0; gv[4 + r] = g1; wv[r] = on0 ? d0.w[r] : 0.f; wv[4 + r] = on1 ? d1.w[r] : 0.f;
            }
            acc = mfma32b(gv, wv, acc);
        };
        Pair pa, pb;
        fetch(0, pa);
        for (int c = 0; c < chunks; c += 2) {
            if (c + 1 < chunks) fetch(c + 1, pb);
            if constexpr (BF) {
                const Pair cur = pa;
                if (c + 2 < chunks) fetch(c + 2, pa);
                use_pair(c, cur, pb, c + 1 < chunks);
            } else {
                use(c, pa);
                if (c + 1 < chunks) {
                    if (c + 2 < chunks) fetch(c + 2, pa);
                    use(c + 1, pb);
                }
            }
        }
        if (i < I) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * q + r;
                if (row >= R) continue;
                const size_t e = ((size_t)g * R + row) * I + i;
                float v = acc[r];
                if (a.slope_of) v *= a.slope_of[e] > 0.f ? 1.f : 0.01f;   // nn.LeakyReLU default slope
                a.dx[e] = v;
            }
        }
        return;
    }
    // ---- dw / db: tile (it, ot); this wave's rows: part `wave` of the steps of 4 rows ----------------------------------------
    const int C = a.row_chunks, tile = (int)blockIdx.x / C, chunk = (int)blockIdx.x - tile * C, it = tile / otiles, ot = tile - it * otiles;
    const int i = 16 * it + m, o = 16 * ot + m;
    const bool ion = i < I, oon = o < O;
    const float* xg = a.x + (size_t)g * a.x_set;
    const int steps = (R + 3) >> 2, spc = (steps + C - 1) / C;        // steps of 4 rows; per chunk
    const int c0 = chunk * spc < steps ? chunk * spc : steps, c1 = c0 + spc < steps ? c0 + spc : steps;
    const int per = (c1 - c0 + BMM_BWD_WAVES - 1) / BMM_BWD_WAVES;
    const int s0 = c0 + wave * per < c1 ? c0 + wave * per : c1, s1 = s0 + per < c1 ? s0 + per : c1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // 2 x UN steps of operands in flight: the chain is load latency, not arithmetic.  Clamped addresses, values masked at use (no
    // branch around a load); the order of the additions is the step order, as before.
    constexpr int UN = 8;
    static_assert(UN == 8, "the bf16 path packs UN steps into one K = 32 MFMA");
    const uint32_t icol = ion ? i : I - 1, ocol = oon ? o : O - 1;
    const float* yg = ACT ? a.act_y + (size_t)g * a.g_set : nullptr;
    // two-source rows: this tile's 16 input columns lie in x (row r / x1_div, I1 columns) or in x2 (row r, I - I1 columns) -- uniform
    const bool two = a.x2 != nullptr, second = two && 16 * it >= a.I1;
    const uint32_t xmagic = (two && !second) ? a.x1_magic : 0u;        // 0: the row itself
    const uint32_t xld = two ? (second ? (uint32_t)(I - a.I1) : (uint32_t)a.I1) : (uint32_t)I;
    const float* xsrc = two ? (second ? a.x2 + (size_t)g * a.x2_set : a.x + (size_t)g * (size_t)(R / a.x1_div) * a.I1) : xg;
    const uint32_t xcol = second ? icol - (uint32_t)a.I1 : icol;
    auto fetch = [&](int s, float (&av)[UN], float (&bv)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int su = s + u < s1 ? s + u : s1 - 1;                 // past the wave's part: its last step again (the same cache lines)
            const int row = 4 * su + q;
            const uint32_t rc = row < R ? row : R - 1;
            av[u] = ld32(xsrc, (xmagic ? __umulhi(rc, xmagic) : rc) * xld + xcol);
            bv[u] = ld32(gg, rc * (uint32_t)O + ocol);
            if constexpr (ACT) { const float yv = ld32(yg, rc * (uint32_t)O + ocol); bv[u] = yv > 0.f ? bv[u] : 0.01f * bv[u]; }
        }
    };
    auto use = [&](int s, const float (&av)[UN], const float (&bv)[UN]) {
        float xs[UN], gs[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool rv = 4 * (s + u) + q < R && s + u < s1;
            xs[u] = (rv && ion) ? av[u] : 0.f; gs[u] = (rv && oon) ? bv[u] : 0.f;
            if constexpr (!BF) acc = mfma4(xs[u], gs[u], acc);
            bsum += gs[u];
        }
        if constexpr (BF) acc = mfma32b(xs, gs, acc);                   // the UN = 8 steps of 4 rows are the 32 reduction indices of one MFMA
    };
    float a0[UN], b0[UN], a1[UN], b1[UN];
    if (s0 < s1) fetch(s0, a0, b0);
    for (int s = s0; s < s1; s += 2 * UN) {
        const bool second = s + UN < s1;
        if (second) fetch(s + UN, a1, b1);
        use(s, a0, b0);
        if (second) {
            if (s + 2 * UN < s1) fetch(s + 2 * UN, a0, b0);
            use(s + UN, a1, b1);
        }
    }
    bsum += __shfl_xor(bsum, 16); bsum += __shfl_xor(bsum, 32);        // the four row quarters of a step
    red[wave][lane] = acc;
    if (q == 0) redb[wave][m] = bsum;
    __syncthreads();
    if (wave == 0) {
        f32x4 sum = red[0][lane];
#pragma unroll
        for (int v = 1; v < BMM_BWD_WAVES; ++v) sum += red[v][lane];
        if (C > 1) {                                                   // this chunk's partial tile + bias sums; k_bmm_dw_reduce finishes
            float* P = a.part + ((size_t)((size_t)g * gridDim_tiles(a) + tile) * C + chunk) * BMM_PART;
            *reinterpret_cast<f32x4*>(P + 4 * lane) = sum;
            if (q == 0) {
                float sb = redb[0][m];
#pragma unroll
                for (int v = 1; v < BMM_BWD_WAVES; ++v) sb += redb[v][m];
                P[256 + m] = sb;
            }
            return;
        }
        if (a.dw && oon) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ii = 16 * it + 4 * q + r;
                if (ii < I) a.dw[((size_t)g * I + ii) * O + o] = sum[r];
            }
        }
        if (a.db && it == 0 && q == 0 && oon) {
            float sb = redb[0][m];
#pragma unroll
            for (int v = 1; v < BMM_BWD_WAVES; ++v) sb += redb[v][m];
            a.db[(size_t)g * O + o] = sb;
        }
    }
}

// Scratch of the row-chunked backward (partial dw tiles): one allocation per (device, stream) -- two streams that run chunked
// backward launches concurrently each add their partial tiles in their own buffer.  Made on the first call outside a stream capture
// (ssd_create makes the default stream's; ssd_bmm_reserve_scratch makes a capture stream's BEFORE the capture starts -- the learner
// captures its step on a stream of its own); inside a capture an allocation fails and the launch stays unchunked.
constexpr size_t BMM_SCRATCH_BYTES = 1 << 20;
constexpr int BMM_SCRATCH_SLOTS = 64;                 // streams per device with a scratch of their own; later ones run unchunked
float* bmm_scratch(hipStream_t stream) {
    struct Slot { hipStream_t stream; float* buf; };
    static Slot slots[64][BMM_SCRATCH_SLOTS] = {};
    static int used[64] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < used[dev]; ++i)
        if (slots[dev][i].stream == stream) return slots[dev][i].buf;
    if (used[dev] == BMM_SCRATCH_SLOTS) return nullptr;
    float* p = nullptr;
    if (hipMalloc((void**)&p, BMM_SCRATCH_BYTES) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // (e.g. inside a capture: unchunked)
    slots[dev][used[dev]++] = Slot{stream, p};
    return p;
}

int launch_bias_bmm_fwd(const float* x, const float* w, const float* b, float* y, int n, int R, int I, int O, hipStream_t s, int leaky) {
    BmmK k = {};
    k.x = x; k.w = w; k.b = b; k.y = y; k.n = n; k.R = R; k.I = I; k.O = O; k.leaky = leaky;
    const int otiles = (O + 15) / 16, units = ((R + 15) / 16) * ((otiles + BMM_TPW - 1) / BMM_TPW);
    if ((long)I * O >= (1L << 30) || (long)R * I >= (1L << 30)) return -2;        // 32-bit byte offsets inside a weight set
    const dim3 grid((units + 3) / 4, n);
    if (g_learner_precision == 1) {
        if ((I & 3) == 0) hipLaunchKernelGGL((k_bias_bmm_fwd<true, true>), grid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((k_bias_bmm_fwd<false, true>), grid, dim3(256), 0, s, k);
    } else {
        if ((I & 3) == 0) hipLaunchKernelGGL((k_bias_bmm_fwd<true, false>), grid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((k_bias_bmm_fwd<false, false>), grid, dim3(256), 0, s, k);
    }
    return 0;
}

// y = b + [x1 (row r / x1_div) | x2 (row r)] w: the two-source forward (see BmmK); in1 a multiple of 16, in2 a multiple of 4
int launch_bias_bmm2_fwd(const float* x1, const float* x2, const float* w, const float* b, float* y, int n, int R, int I1, int I2, int O, int x1_div,
                         int x2_shared, hipStream_t s) {
    BmmK k = {};
    const int I = I1 + I2;
    k.x = x1; k.x2 = x2; k.I1 = I1; k.x1_div = x1_div; k.x2_set = x2_shared ? 0 : (long)R * I2;
    k.w = w; k.b = b; k.y = y; k.n = n; k.R = R; k.I = I; k.O = O;
    if ((I1 & 15) || x1_div < 1 || R % x1_div) return -3;
    if ((long)I * O >= (1L << 30) || (long)R * I >= (1L << 30)) return -2;
    const int otiles = (O + 15) / 16, units = ((R + 15) / 16) * ((otiles + BMM_TPW - 1) / BMM_TPW);
    const dim3 grid((units + 3) / 4, n);
    const bool xv = (I2 & 3) == 0;      // 16-byte row loads need rows of whole quads in both sources (Harvest: 15 extra features -> scalar loads)
    if (g_learner_precision == 1) {
        if (xv) hipLaunchKernelGGL((k_bias_bmm_fwd<true, true>), grid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((k_bias_bmm_fwd<false, true>), grid, dim3(256), 0, s, k);
    } else {
        if (xv) hipLaunchKernelGGL((k_bias_bmm_fwd<true, false>), grid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((k_bias_bmm_fwd<false, false>), grid, dim3(256), 0, s, k);
    }
    return 0;
}

int launch_bias_bmm_bwd(const float* g, const float* x, const float* w, float* dx, float* dw, float* db, const float* slope_of, int n, int R,
                        int I, int O, hipStream_t s, long x_set, long g_set, const float* act_y, const float* x2, int I1, int x1_div, int x2_shared,
                        long w_set) {
    BmmK k = {};
    k.act_y = act_y;
    if (x2) {                      // two-source rows: dw / db only (dx of the first source is a row-group sum: the caller forms it from the summed g)
        if (dx || (I1 & 15) || x1_div < 1 || R % x1_div) return -3;
        k.x2 = x2; k.I1 = I1; k.x1_div = x1_div; k.x2_set = x2_shared ? 0 : (long)R * (I - I1);
        k.x1_magic = x1_div == 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)x1_div - 1) / (uint64_t)x1_div);
        if (R >= (1 << 28)) return -2;
    }
    k.w_set = w_set;
    k.g = g; k.x = x; k.w = w; k.dx = dx; k.dw = dw; k.db = db; k.slope_of = slope_of; k.n = n; k.R = R; k.I = I; k.O = O;
    k.x_set = x_set ? x_set : (long)R * I; k.g_set = g_set ? g_set : (long)R * O;
    const int dxb = dx ? (((R + 15) / 16) * ((I + 15) / 16) + BMM_BWD_WAVES - 1) / BMM_BWD_WAVES : 0;
    int dwb = (dw || db) ? ((I + 15) / 16) * ((O + 15) / 16) : 0;
    if (dxb + dwb == 0) return 0;
    // row chunks: only for long row axes (>= 64 steps of 4 rows per wave) whose tiles would leave the chip mostly idle
    k.row_chunks = 1;
    const int tiles = dwb;
    if (dwb && (R + 3) / 4 >= 64 * BMM_BWD_WAVES && dwb * n <= 128) {
        int c = 256 / (dwb * n);
        if (c > 8) c = 8;
        float* part = bmm_scratch(s);
        if (c > 1 && part && (size_t)n * dwb * c * BMM_PART * sizeof(float) <= BMM_SCRATCH_BYTES) { k.row_chunks = c; k.part = part; dwb *= c; }
    }
    if ((long)R * I >= (1L << 30) || (long)R * O >= (1L << 30) || (long)I * O >= (1L << 30)) return -2;       // 32-bit byte offsets inside a weight / operand set
    const dim3 grid(dwb + dxb, n), block(BMM_BWD_WAVES * 64);
    const bool ov = (O & 3) == 0, bf = g_learner_precision == 1;
#define SSD_BWD(OV_, ACT_, BF_) hipLaunchKernelGGL((k_bias_bmm_bwd<OV_, ACT_, BF_>), grid, block, 0, s, k, dwb)
    if (act_y) {
        if (ov) { if (bf) SSD_BWD(true, true, true); else SSD_BWD(true, true, false); }
        else { if (bf) SSD_BWD(false, true, true); else SSD_BWD(false, true, false); }
    } else {
        if (ov) { if (bf) SSD_BWD(true, false, true); else SSD_BWD(true, false, false); }
        else { if (bf) SSD_BWD(false, false, true); else SSD_BWD(false, false, false); }
    }
#undef SSD_BWD
    if (k.row_chunks > 1) hipLaunchKernelGGL(k_bmm_dw_reduce, dim3(tiles, n), dim3(64), 0, s, k);
    return 0;
}


// ---- weight gradient of the encoder's convolution on class-code windows ----------------------------------------------------------
// Reference: the Conv2d(3, 6, 3) of HomophilyAgent.conv_to_fc (homophily_agent.py:20-27) applied to the simplified-palette
// observation: every window cell is one of four classes and lights at most one colour plane at 255/256 (cleanup.py:93-105).  So
//   d_w[oc][ch][dy][dx] = 255/256 * sum over (row, y, x) of d_conv[row][oc][y][x] * [class(row, y + dy, x + dx) lights plane ch]
// needs no im2col and no f32 planes: a lane owns an output position, reads the 9 class bytes of its 3 x 3 patch from the wave's LDS
// copy of the window and adds the six channel gradients of its position into 6 x 27 (+ 6 bias) per-lane accumulators selected by
// 0/1 masks.  A wave walks CW_ROWS windows, then the accumulators are summed over its lanes and written as one row of
// `partial` [waves, 168]; the caller adds the rows (ops.column_sums: deterministic).  HBM: d_conv is read once (the term that
// matters: 4 * 6 * O * O bytes per window), codes V * V bytes per window.
constexpr int CW_ROWS = 4, CW_OUT = 6 * 27 + 6;     // windows per wave (measured at 8080 windows: 8 -> 46 us, 4 -> 31.5 us, 2 -> 42 us)
// sum over the 64 lanes with DPP (no LDS crossbar): quad swaps, row rotations, row broadcasts; lane 63 ends with the total
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v = dpp_add<0x124, 0xF>(v);    // row_ror:4
    v = dpp_add<0x128, 0xF>(v);    // row_ror:8
    v = dpp_add<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3
    return v;
}
template <int V>
__global__ __launch_bounds__(256) void k_conv_wgrad(const uint8_t* __restrict__ codes, const float* __restrict__ d_conv, float* __restrict__ partial, int R) {
    constexpr int O = V - 2, P = O * O, VV = V * V, VVP = (VV + 15) & ~15;
    __shared__ uint8_t win[4][VVP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x * 4 + wave;                              // global wave = row of `partial`
    float acc[6][27], accb[6];
#pragma unroll
    for (int oc = 0; oc < 6; ++oc) {
        accb[oc] = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[oc][k] = 0.f;
    }
    // items = (window, pass of 64 positions); the six gradient values of the NEXT item are requested before the current one is
    // accumulated (a wave has nothing else to hide the load latency behind)
    constexpr int PASSES = (P + 63) / 64;
    const int row_begin = wg * CW_ROWS, rows_here = row_begin >= R ? 0 : (R - row_begin < CW_ROWS ? R - row_begin : CW_ROWS);
    const int items = rows_here * PASSES;
    auto fetch = [&](int it, float (&v)[6]) {
        const int row = row_begin + it / PASSES, p = (it % PASSES) * 64 + lane;
        const float* dr = d_conv + (size_t)row * 6 * P;
#pragma unroll
        for (int oc = 0; oc < 6; ++oc) v[oc] = p < P ? dr[oc * P + p] : 0.f;
    };
    float cur[6], nxt[6];
    if (items > 0) fetch(0, nxt);
    for (int it = 0; it < items; ++it) {
#pragma unroll
        for (int oc = 0; oc < 6; ++oc) cur[oc] = nxt[oc];
        if (it + 1 < items) fetch(it + 1, nxt);
        const int pass = it % PASSES;
        if (pass == 0) {                                               // a new window: its class codes into the wave's LDS buffer
            const int row = row_begin + it / PASSES;
            __builtin_amdgcn_wave_barrier();
            for (int e = lane; e < VV; e += 64) win[wave][e] = codes[(size_t)row * VV + e];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const int p = pass * 64 + lane;
        const int pc = p < P ? p : 0, y = pc / O, x = pc - y * O;
        float mk[27];                                                  // [tap = dy * 3 + dx][plane]: 1 where the patch cell lights the plane
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int c = win[wave][(y + t / 3) * V + x + t % 3];
            mk[3 * t + 0] = c == 2 ? 1.f : 0.f;                        // waste -> R
            mk[3 * t + 1] = c == 1 ? 1.f : 0.f;                        // apple -> G
            mk[3 * t + 2] = c == 3 ? 1.f : 0.f;                        // wall / agent -> B
        }
#pragma unroll
        for (int oc = 0; oc < 6; ++oc) {
            const float v = cur[oc];                                   // 0 for the lanes past the last position
            accb[oc] += v;
#pragma unroll
            for (int k = 0; k < 27; ++k) acc[oc][k] = fmaf(v, mk[k], acc[oc][k]);
        }
    }
    // sum over the 64 lanes (fixed order), lane 63 writes; layout: [oc][ch][dy][dx] then the 6 bias sums
    float* out = partial + (size_t)wg * CW_OUT;
#pragma unroll
    for (int oc = 0; oc < 6; ++oc) {
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const float v = wave_sum_to_lane63(acc[oc][k]);
            const int t = k / 3, ch = k - 3 * t;
            if (lane == 63) out[(oc * 3 + ch) * 9 + t] = v * (255.f / 256.f);
        }
        const float b = wave_sum_to_lane63(accb[oc]);
        if (lane == 63) out[6 * 27 + oc] = b;
    }
}

int conv_wgrad_partial_rows(int R) { return ((R + CW_ROWS - 1) / CW_ROWS + 3) / 4 * 4; }

int launch_conv_wgrad(const uint8_t* codes, const float* d_conv, float* partial, int R, int V, hipStream_t s) {
    const int waves = conv_wgrad_partial_rows(R);
    if (V == 15) hipLaunchKernelGGL(k_conv_wgrad<15>, dim3(waves / 4), dim3(256), 0, s, codes, d_conv, partial, R);
    else if (V == 31) hipLaunchKernelGGL(k_conv_wgrad<31>, dim3(waves / 4), dim3(256), 0, s, codes, d_conv, partial, R);
    else return -2;
    return 0;
}

}  // namespace ssd
