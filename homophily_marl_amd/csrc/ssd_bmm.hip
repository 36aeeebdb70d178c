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
#include <stdint.h>

namespace ssd {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte load at any dword address

struct BmmK {
    const float *x, *w, *b, *g;
    float *y, *dx, *dw, *db;
    int n, R, I, O;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// 4 consecutive floats p[k0 .. k0 + 3] of a row of `len` floats; elements at or past len read as 0
__device__ __forceinline__ f32x4 load4(const float* p, int k0, int len) {
    if (k0 + 4 <= len) return *reinterpret_cast<const f32x4_u*>(p + k0);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (k0 + r < len) v[r] = p[k0 + r];
    return v;
}

// y[g][row][o] = b[g][o] + sum_i x[g][row][i] w[g][i][o].  grid (ceil(R / 16), n), 256 threads: wave v takes the output tiles
// v, v + 4, ...  D[row 4 q + r][col m]; K-step (chunk c, r) uses k = 16 c + 4 q + r for lane quarter q.
constexpr int BMM_MAX_TPW = 3;                  // output tiles per wave: O <= 192
__global__ __launch_bounds__(256) void k_bias_bmm_fwd(BmmK a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int g = blockIdx.y, row0 = blockIdx.x * 16;
    const int otiles = (a.O + 15) >> 4;
    if (wave >= otiles) return;
    const int R = a.R, I = a.I, O = a.O;
    const int rowc = row0 + m < R ? row0 + m : R - 1;
    const float* xr = a.x + ((size_t)g * R + rowc) * I;
    const float* wg = a.w + (size_t)g * I * O;
    f32x4 acc[BMM_MAX_TPW];
#pragma unroll
    for (int t = 0; t < BMM_MAX_TPW; ++t) {
        const int o = 16 * (wave + 4 * t) + m;
        const float bv = (wave + 4 * t < otiles && o < O) ? a.b[(size_t)g * O + o] : 0.f;
        acc[t] = f32x4{bv, bv, bv, bv};
    }
    const int chunks = (I + 15) >> 4;
    for (int c = 0; c < chunks; ++c) {
        const int k0 = 16 * c + 4 * q;
        const f32x4 xa = load4(xr, k0, I);
        float bs[BMM_MAX_TPW][4];
#pragma unroll
        for (int t = 0; t < BMM_MAX_TPW; ++t) {
            const int o = 16 * (wave + 4 * t) + m;
            const bool on = wave + 4 * t < otiles && o < O;
#pragma unroll
            for (int r = 0; r < 4; ++r) bs[t][r] = (on && k0 + r < I) ? wg[(size_t)(k0 + r) * O + o] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < BMM_MAX_TPW; ++t)
                if (wave + 4 * t < otiles) acc[t] = mfma4(xa[r], bs[t][r], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < BMM_MAX_TPW; ++t) {
        const int o = 16 * (wave + 4 * t) + m;
        if (wave + 4 * t >= otiles || o >= O) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * q + r;
            if (row < R) a.y[((size_t)g * R + row) * O + o] = acc[t][r];
        }
    }
}

// One launch, grid (DX + DW, n): blocks [0, DX) compute dx = g w^T for a 16-row tile (wave v: input tiles v, v + 4, ...); blocks
// [DX, DX + DW) compute one 16 x 16 tile of dw = x^T g (rows of the batch = K, a sixteenth per wave, partials added in LDS in wave
// order) and, for input tile 0, the matching 16 entries of db = column sums of g.
constexpr int BMM_BWD_WAVES = 16;               // the row axis (K of dw: up to T B n = 8080 rows) is split 16 ways
constexpr int BMM_MAX_ITPW = 1;                 // input tiles per wave in the dx role: I <= 256
__global__ __launch_bounds__(BMM_BWD_WAVES * 64) void k_bias_bmm_bwd(BmmK a, int dx_blocks) {
    __shared__ f32x4 red[BMM_BWD_WAVES][64];
    __shared__ float redb[BMM_BWD_WAVES][16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int g = blockIdx.y;
    const int R = a.R, I = a.I, O = a.O;
    const int itiles = (I + 15) >> 4, otiles = (O + 15) >> 4;
    const float* gg = a.g + (size_t)g * R * O;
    if ((int)blockIdx.x < dx_blocks) {
        if (!a.dx || wave >= itiles) return;
        const int row0 = blockIdx.x * 16;
        const int rowc = row0 + m < R ? row0 + m : R - 1;
        const float* gr = gg + (size_t)rowc * O;
        const float* wg = a.w + (size_t)g * I * O;
        f32x4 acc[BMM_MAX_ITPW];
#pragma unroll
        for (int t = 0; t < BMM_MAX_ITPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int chunks = (O + 15) >> 4;
        for (int c = 0; c < chunks; ++c) {
            const int k0 = 16 * c + 4 * q;                             // reduction index = output feature
            const f32x4 ga = load4(gr, k0, O);
            f32x4 wb[BMM_MAX_ITPW];
#pragma unroll
            for (int t = 0; t < BMM_MAX_ITPW; ++t) {
                const int i = 16 * (wave + BMM_BWD_WAVES * t) + m;
                wb[t] = (wave + BMM_BWD_WAVES * t < itiles && i < I) ? load4(wg + (size_t)i * O, k0, O) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < BMM_MAX_ITPW; ++t)
                    if (wave + BMM_BWD_WAVES * t < itiles) acc[t] = mfma4(ga[r], wb[t][r], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < BMM_MAX_ITPW; ++t) {
            const int i = 16 * (wave + BMM_BWD_WAVES * t) + m;
            if (wave + BMM_BWD_WAVES * t >= itiles || i >= I) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * q + r;
                if (row < R) a.dx[((size_t)g * R + row) * I + i] = acc[t][r];
            }
        }
        return;
    }
    // ---- dw / db: tile (it, ot); this wave's rows: part `wave` of the steps of 4 rows ----------------------------------------
    const int tile = blockIdx.x - dx_blocks, it = tile / otiles, ot = tile - it * otiles;
    const int i = 16 * it + m, o = 16 * ot + m;
    const bool ion = i < I, oon = o < O;
    const float* xg = a.x + (size_t)g * R * I;
    const int steps = (R + 3) >> 2, per = (steps + BMM_BWD_WAVES - 1) / BMM_BWD_WAVES;
    const int s0 = wave * per, s1 = s0 + per < steps ? s0 + per : steps;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    constexpr int UN = 4;
    int s = s0;
    for (; s + UN <= s1; s += UN) {
        float av[UN], bv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int row = 4 * (s + u) + q;
            const bool rv = row < R;
            av[u] = (rv && ion) ? xg[(size_t)row * I + i] : 0.f;
            bv[u] = (rv && oon) ? gg[(size_t)row * O + o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) { acc = mfma4(av[u], bv[u], acc); bsum += bv[u]; }
    }
    for (; s < s1; ++s) {
        const int row = 4 * s + q;
        const bool rv = row < R;
        const float av = (rv && ion) ? xg[(size_t)row * I + i] : 0.f, bv = (rv && oon) ? gg[(size_t)row * O + o] : 0.f;
        acc = mfma4(av, bv, acc); bsum += bv;
    }
    bsum += __shfl_xor(bsum, 16); bsum += __shfl_xor(bsum, 32);        // the four row quarters of a step
    red[wave][lane] = acc;
    if (q == 0) redb[wave][m] = bsum;
    __syncthreads();
    if (wave == 0) {
        f32x4 sum = red[0][lane];
#pragma unroll
        for (int v = 1; v < BMM_BWD_WAVES; ++v) sum += red[v][lane];
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

int launch_bias_bmm_fwd(const float* x, const float* w, const float* b, float* y, int n, int R, int I, int O, hipStream_t s) {
    if (O > 16 * 4 * BMM_MAX_TPW) return -3;
    BmmK k = {};
    k.x = x; k.w = w; k.b = b; k.y = y; k.n = n; k.R = R; k.I = I; k.O = O;
    hipLaunchKernelGGL(k_bias_bmm_fwd, dim3((R + 15) / 16, n), dim3(256), 0, s, k);
    return 0;
}

int launch_bias_bmm_bwd(const float* g, const float* x, const float* w, float* dx, float* dw, float* db, int n, int R, int I, int O, hipStream_t s) {
    if (I > 16 * BMM_BWD_WAVES * BMM_MAX_ITPW) return -3;
    BmmK k = {};
    k.g = g; k.x = x; k.w = w; k.dx = dx; k.dw = dw; k.db = db; k.n = n; k.R = R; k.I = I; k.O = O;
    const int dxb = dx ? (R + 15) / 16 : 0;
    const int dwb = (dw || db) ? ((I + 15) / 16) * ((O + 15) / 16) : 0;
    if (dxb + dwb == 0) return 0;
    hipLaunchKernelGGL(k_bias_bmm_bwd, dim3(dxb + dwb, n), dim3(BMM_BWD_WAVES * 64), 0, s, k, dxb);
    return 0;
}

}  // namespace ssd
