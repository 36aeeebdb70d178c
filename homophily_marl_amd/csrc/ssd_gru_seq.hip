// ssd_gru_seq.hip -- the learner's recurrence as ONE forward and ONE backward launch over all T timesteps.
//
// Reference: HomophilyLearner.train evaluates the controller for t = 0..T-1 from a zero hidden state (homophily_learner.py:68-91),
// each step running the hand-written GRU cell of HomophilyAgent (homophily_agent.py:162-165,188-191):
//     r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z h,   gh = h W_h + b_h
// The input-side projections gi = x W_i + b_i do not depend on the recurrence and are computed for all T at once by the caller
// (one batched GEMM); what is left per step is a [B, 64] x [64, 192] product and the gate arithmetic -- at the learner's batch
// (B = 16 episodes) a latency chain of ~10 tiny launches per step and direction.  Here one workgroup owns one (weight set g,
// 16-row tile) and walks the whole sequence:
//   * 4 compute waves (+ a writer wave that takes each step's results from an LDS staging image to HBM, so that no compute wave
//     ever waits for a store); wave w owns hidden features 16 w .. 16 w + 15 (all three gates); its slice of W_h lives in registers for all T as
//     MFMA A operands, products are computed in the transposed form of ssd_policy_mfma.hip (activation row on the lane, 4
//     consecutive features in the lane's registers), so the gate arithmetic is lane-local;
//   * forward: v_mfma_f32_16x16x32_f16 on two-term f16 splits (f32-equivalent; |h| < 1), the new state crosses the waves through a
//     double-buffered LDS image of the split terms: one barrier per step;
//   * backward walks t = T-1 .. 0 with the carried dL/dh in registers (exact f32 MFMAs); dL/dW_h is one product over all
//     (t, row) pairs computed after the walk (the x^T g role of csrc/ssd_bmm.hip).
// H = 64 is fixed.  Rows are independent sequences: any B (tiles of 16, the last one masked).
#include "ssd_policy_common.h"

namespace ssd {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr int GH = 64, G3 = 192;                        // hidden, 3 * hidden
constexpr int HSH = 72;                                  // LDS row stride of the f16 state image (halves): 144 B = 16 * odd -> conflict-free b128 reads
// forward: f32-equivalent products on the f16 matrix cores from two-term splits (ssd_policy_mfma.hip); exact power-of-two scales
// keep the low terms in f16's normal range (|h| < 1, |W_h| ~ 0.1)
constexpr float GRU_WS = 64.f, GRU_XS = 16.f, GRU_INV = 1.f / (GRU_WS * GRU_XS);

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }   // backward-side helpers keep libm accuracy
// sigmoid / tanh on v_exp_f32 + v_rcp_f32 (1 ulp each), as the rollout heads
__device__ __forceinline__ float sigm_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast_(float x) {
    const float ax = fminf(fabsf(x), 15.f);
    const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * ax) + 1.f);
    return copysignf(t, x);
}
// The workgroup barrier of a recurrence step orders LDS traffic only: wait for this wave's LDS operations and join.
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
// Results leave through a WRITER wave.  gfx950 retires a wave's loads and stores through one in-order counter: a compute wave that
// stores its step's results waits, at its next counted wait for a prefetched input, for those stores to be acknowledged by memory as
// well (measured: 5 result stores per step cost the forward walk 33 us of 90, the backward walk 56 us of 165).  So the compute waves
// put a step's results into an LDS staging image next to the state image (same barrier), and wave 4 -- which never waits for
// memory -- copies the image of step t to HBM in 1 KiB pieces while the compute waves are in step t + 1.  Rows of the staging image
// are padded to 16 * odd bytes (mod 256): b128 accesses of 16 lanes with different rows fall on different banks.
constexpr int GRU_COMPUTE_WAVES = 4, GRU_THREADS = (GRU_COMPUTE_WAVES + 1) * 64;
// piece i (1 KiB) of a [16, ROWF] f32 tile, lane l: the lane's 16 bytes lie in row (1024 i + 16 l) / (4 ROWF) at byte column (..) % (4 ROWF)
template <int ROWF>
__device__ __forceinline__ void piece_rc(int i, int lane, int& row, int& colf) {
    const int off = 1024 * i + 16 * lane;
    row = off / (4 * ROWF); colf = (off - row * 4 * ROWF) >> 2;
}
__device__ __forceinline__ f32x4 mma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
using b8 = __attribute__((ext_vector_type(8))) __bf16;
using b4 = __attribute__((ext_vector_type(4))) __bf16;
__device__ __forceinline__ f32x4 mmab(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}

// gi [T, G, B, 192], wh [G, 64, 192], bh [G, 192] -> hs [G, T, B, 64]; optional (training) rzn [T, G, B, 192], ghn [T, G, B, 64]
// One workgroup per (weight set g, 16-row tile), 4 compute waves + the writer; wave ft owns hidden features 16 ft .. 16 ft + 15 of the three gates.
// Per step: gh^T = W_h^T h^T as 18 v_mfma_f32_16x16x32_f16 (3 gates x 2 K-steps x 3 split products; the W_h slice stays in
// registers as hi / lo fragments for all T), lane-local gate arithmetic, and the new state goes to the other waves through a
// double-buffered LDS image that already holds the hi / lo f16 terms (one barrier per step; each lane splits only its own 4 values).
// B is a multiple of 16 (the host pads): no row predicate and no branch inside the steps, so that the compiler's vmcnt accounting
// stays exact.
// Where the input-side projections (forward) / their gradients (backward) live: up to 4 separately allocated parts of G / n_parts
// weight sets each.  Time-major [T, G, B, 192] (one part) is t_stride = G B 192, set_stride = B 192; the per-agent affine layers'
// own output [sets, T, B, 192] (set-major) is set_stride = T B 192, t_stride = B 192 -- the learner hands its four projection outputs
// (live / target net x env / inc head) over as they are, without concatenating or transposing 12-25 MB copies.
struct GruParts {
    float* p[4];
    int spp;                       // weight sets per part
    long set_stride, t_stride;     // elements
};
__device__ __forceinline__ float* gru_part_base(const GruParts& P, int g) {
    const int part = g / P.spp;
    float* b = part == 0 ? P.p[0] : (part == 1 ? P.p[1] : (part == 2 ? P.p[2] : P.p[3]));
    return b + (size_t)(g - part * P.spp) * P.set_stride;
}

// The recurrence weights as up to 4 separately allocated parts of G / n_parts weight sets each (wh [sets, 64, 192], bh [sets, 192]): the
// learner hands over the live net's and the target net's images as they are (no concatenation per step).
struct GruW {
    const float* wh[4];
    const float* bh[4];
    int spp;                       // weight sets per part
};
__device__ __forceinline__ const float* gru_wh(const GruW& W, int g) {
    const int part = g / W.spp;
    const float* b = part == 0 ? W.wh[0] : (part == 1 ? W.wh[1] : (part == 2 ? W.wh[2] : W.wh[3]));
    return b + (size_t)(g - part * W.spp) * GH * G3;
}
__device__ __forceinline__ const float* gru_bh(const GruW& W, int g) {
    const int part = g / W.spp;
    const float* b = part == 0 ? W.bh[0] : (part == 1 ? W.bh[1] : (part == 2 ? W.bh[2] : W.bh[3]));
    return b + (size_t)(g - part * W.spp) * G3;
}

constexpr int FST = 324;            // staging row stride (floats): [hs 64 | r 64 | z 64 | n 64 | gh_n 64] + pad, 1296 B = 16 * 81
// BF: the labelled reduced-precision variant (learner_dtype: bf16): W_h and the state as single bf16 terms, ONE v_mfma_f32_16x16x32_bf16
// per product (6 per step instead of 18), no scales (bf16 has f32's exponent range), f32 accumulation and gate arithmetic.
template <bool TRAIN, bool BF = false>
__global__ __launch_bounds__(GRU_THREADS) void k_gru_seq_fwd(const GruParts gi, const GruW W,
                                                             float* __restrict__ hs, float* __restrict__ rzn, float* __restrict__ ghn, int T, int G,
                                                             int B, int tiles, int32_t* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) _Float16 hx[2][2][16][HSH];   // [buffer][term][row][feature]
    __shared__ __attribute__((aligned(16))) float st[2][16][FST];         // [buffer][row][staged results of the step]
    const int tid = threadIdx.x, lane = tid & 63, ft = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x / tiles, tile = blockIdx.x - g * tiles;
    if (ft == GRU_COMPUTE_WAVES) {                                     // the writer wave
        float* hs_t = hs + (((size_t)g * T) * B + tile * 16) * GH + 4 * lane;          // + t * B * GH
        float* rzn_t = TRAIN ? rzn + ((size_t)g * B + tile * 16) * G3 + 4 * lane : nullptr;   // + t * G * B * G3
        float* ghn_t = TRAIN ? ghn + ((size_t)g * B + tile * 16) * GH + 4 * lane : nullptr;   // + t * G * B * GH
        for (int t = 0; t < T; ++t) {
            lds_barrier();                                             // the step's image is complete
            const float* im = &st[t & 1][0][0];
            f32x4 vh[4], vr[TRAIN ? 12 : 1], vn[TRAIN ? 4 : 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) { int r, c; piece_rc<GH>(i, lane, r, c); vh[i] = *reinterpret_cast<const f32x4*>(im + r * FST + c); }
            if constexpr (TRAIN) {
#pragma unroll
                for (int i = 0; i < 12; ++i) { int r, c; piece_rc<G3>(i, lane, r, c); vr[i] = *reinterpret_cast<const f32x4*>(im + r * FST + GH + c); }
#pragma unroll
                for (int i = 0; i < 4; ++i) { int r, c; piece_rc<GH>(i, lane, r, c); vn[i] = *reinterpret_cast<const f32x4*>(im + r * FST + 4 * GH + c); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(hs_t + (size_t)t * B * GH + 256 * i) = vh[i];
            if constexpr (TRAIN) {
#pragma unroll
                for (int i = 0; i < 12; ++i) *reinterpret_cast<f32x4*>(rzn_t + (size_t)t * G * B * G3 + 256 * i) = vr[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(ghn_t + (size_t)t * G * B * GH + 256 * i) = vn[i];
            }
        }
        return;
    }
    const int m = lane & 15, q = lane >> 4;
    const int row = tile * 16 + m, rc = row;
    const float* __restrict__ wh = gru_wh(W, g);                       // this weight set's [64, 192] / [192]
    const float* __restrict__ bh = gru_bh(W, g);
    // resident A fragments: lane (q, m) = output feature gate * 64 + 16 ft + m, reduction indices k = 32 s + 8 q + j
    u32x4 ah[3][2], al[3][2];
    bool out_of_range = false;                                        // ONE test after the 48 loads: a branch per value made each load wait for the one before (14 us)
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if constexpr (BF) {
                b8 h;
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = (__bf16)wh[(size_t)(32 * s + 8 * q + j) * G3 + gate * GH + 16 * ft + m];
                ah[gate][s] = __builtin_bit_cast(u32x4, h); al[gate][s] = ah[gate][s];
            } else {
                h8 h, l;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = wh[(size_t)(32 * s + 8 * q + j) * G3 + gate * GH + 16 * ft + m] * GRU_WS;
                    out_of_range |= !(fabsf(w) <= F16_MAX);            // |W_h| >= 1 023.5 (or NaN): outside the split's range; |h| < 1 needs no check
                    h[j] = (_Float16)w; l[j] = (_Float16)(w - (float)h[j]);
                }
                ah[gate][s] = __builtin_bit_cast(u32x4, h); al[gate][s] = __builtin_bit_cast(u32x4, l);
            }
        }
    if (out_of_range && err) atomicOr(err, ERR_F16_RANGE);
    f32x4 bias[3];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) bias[gate] = *reinterpret_cast<const f32x4*>(bh + gate * GH + 16 * ft + 4 * q);
    const int fo = 16 * ft + 4 * q;                                   // this lane's 4 features
    f32x4 hown = {0.f, 0.f, 0.f, 0.f};                                // h_{t-1}[row m][fo .. fo + 3]
    u32x4 xh[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, xl[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};   // B operand: h_0 = 0
    // The step's input-side projections are requested PF steps ahead; the compute waves issue no stores, so a counted wait for a
    // prefetched projection waits for nothing younger.
    constexpr int PF = 4;
    f32x4 gbuf[PF][3];
    const float* gbase = gru_part_base(gi, g) + (size_t)rc * G3 + fo;
    auto load_gi = [&](int t, f32x4 (&d)[3]) {
        const float* gir = gbase + (size_t)t * gi.t_stride;
        d[0] = *reinterpret_cast<const f32x4*>(gir); d[1] = *reinterpret_cast<const f32x4*>(gir + GH); d[2] = *reinterpret_cast<const f32x4*>(gir + 2 * GH);
    };
#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (d < T) load_gi(d, gbuf[d]);
    auto step = [&](int t, f32x4 (&gb)[3]) {
        const f32x4 gr = gb[0], gz = gb[1], gn = gb[2];
        if (t + PF < T) load_gi(t + PF, gb);
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (t > 0) {                                                   // small terms first; the three gates' chains interleave
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if constexpr (BF) {
#pragma unroll
                    for (int gate = 0; gate < 3; ++gate) acc[gate] = mmab(ah[gate][s], xh[s], acc[gate]);
                } else {
#pragma unroll
                    for (int gate = 0; gate < 3; ++gate) acc[gate] = mma16(al[gate][s], xh[s], acc[gate]);
#pragma unroll
                    for (int gate = 0; gate < 3; ++gate) acc[gate] = mma16(ah[gate][s], xl[s], acc[gate]);
#pragma unroll
                    for (int gate = 0; gate < 3; ++gate) acc[gate] = mma16(ah[gate][s], xh[s], acc[gate]);
                }
            }
        }
        f32x4 hn, rg, zg, ng, an;
        constexpr float INVS = BF ? 1.f : GRU_INV;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rg[r] = sigm_fast(gr[r] + fmaf(acc[0][r], INVS, bias[0][r]));
            zg[r] = sigm_fast(gz[r] + fmaf(acc[1][r], INVS, bias[1][r]));
            an[r] = fmaf(acc[2][r], INVS, bias[2][r]);                 // gh_n
            ng[r] = tanh_fast_(gn[r] + rg[r] * an[r]);
            hn[r] = (1.f - zg[r]) * ng[r] + zg[r] * hown[r];
        }
        hown = hn;
        if constexpr (BF) {   // this lane's 4 new values as bf16 -> the LDS image the other waves read their B operand from
            b4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (__bf16)hn[r];
            *reinterpret_cast<u32x2*>(&hx[t & 1][0][m][fo]) = __builtin_bit_cast(u32x2, h);
        } else {   // ... as scaled hi / lo f16 terms
            h4 h, l;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float x = hn[r] * GRU_XS; h[r] = (_Float16)x; l[r] = (_Float16)(x - (float)h[r]); }
            *reinterpret_cast<u32x2*>(&hx[t & 1][0][m][fo]) = __builtin_bit_cast(u32x2, h);
            *reinterpret_cast<u32x2*>(&hx[t & 1][1][m][fo]) = __builtin_bit_cast(u32x2, l);
        }
        {   // results of the step -> staging image (the writer wave takes them to hs / rzn / ghn)
            float* o = &st[t & 1][m][fo];
            *reinterpret_cast<f32x4*>(o) = hn;
            if constexpr (TRAIN) {
                *reinterpret_cast<f32x4*>(o + GH) = rg; *reinterpret_cast<f32x4*>(o + 2 * GH) = zg; *reinterpret_cast<f32x4*>(o + 3 * GH) = ng;
                *reinterpret_cast<f32x4*>(o + 4 * GH) = an;
            }
        }
        lds_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            xh[s] = *reinterpret_cast<const u32x4*>(&hx[t & 1][0][m][32 * s + 8 * q]);
            if constexpr (!BF) xl[s] = *reinterpret_cast<const u32x4*>(&hx[t & 1][1][m][32 * s + 8 * q]);
        }
    };
    int t0 = 0;
    for (; t0 + PF <= T; t0 += PF) {                                   // whole groups: the PF register sets rotate by unrolling
#pragma unroll
        for (int d = 0; d < PF; ++d) step(t0 + d, gbuf[d]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (t0 + d < T) step(t0 + d, gbuf[d]);
}

// dhs [G, T, B, 64] (dL/d hs), hs, rzn, ghn, wh as above -> d_gi [T, G, B, 192], dgh [G, T, B, 192] (dL/dgh_t, the B operand of the
// weight gradient), d_bh_part [G, tiles, 192].  The recurrence carries dL/dh only: dL/dh_{t-1} = direct + dL/dgh_t W_h^T.  Gradients
// span too many decades for fixed-scale f16 splits, and gfx950's f32 MFMA runs at the VALU rate (48 of them per step were the
// kernel's critical path), so the product is evaluated on v_mfma_f32_16x16x32_bf16 from THREE-term bf16 splits x = b1 + b2 + b3
// (bf16 has f32's exponent range; 3 x 8 significand bits) keeping the six partial products of order <= 2^-16: f32-equivalent,
// 36 MFMAs of 16 cycles per step.  dL/dW_h = sum_t h_{t-1}^T dL/dgh_t does not feed the recurrence: it is one product per weight
// set over all (t, row) pairs, computed afterwards (launch_gru_seq_bwd) instead of inside this kernel's 100-step latency chain.
constexpr int DSB = 200;               // LDS row stride of the bf16 dL/dgh image (halves): 400 B = 16 * odd (mod 256)
__device__ __forceinline__ void split3(float x, __bf16& t1, __bf16& t2, __bf16& t3) {
    t1 = (__bf16)x; const float r1 = x - (float)t1;                   // both subtractions are exact
    t2 = (__bf16)r1; const float r2 = r1 - (float)t2;
    t3 = (__bf16)r2;
}

constexpr int BST = 260;            // staging row stride (floats): [d_r 64 | d_z 64 | d_n 64 | d_hn 64] + pad, 1040 B = 16 * 65
constexpr int GRU_BWD_LDS = 2 * 3 * 16 * DSB * 2 + 2 * 16 * BST * 4;   // bf16 operand image + f32 staging image: 71 680 B (dynamic)
// BF: single bf16 terms (learner_dtype: bf16): 6 MFMAs per step instead of 36.
template <bool BF>
__global__ __launch_bounds__(GRU_THREADS) void k_gru_seq_bwd(const GruParts dhs, const float* __restrict__ hs, const float* __restrict__ rzn,
                                                             const float* __restrict__ ghn, const GruW W, const GruParts d_gi,
                                                             float* __restrict__ dgh, float* __restrict__ d_bh_part, int T, int G, int B, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gru_lds[];
    __bf16 (*dgb)[3][16][DSB] = reinterpret_cast<__bf16 (*)[3][16][DSB]>(gru_lds);                    // dL/dgh of the step: [buffer][term][row][192]
    float (*st)[16][BST] = reinterpret_cast<float (*)[16][BST]>(gru_lds + 2 * 3 * 16 * DSB * 2);         // [buffer][row][staged results of the step]
    const int tid = threadIdx.x, lane = tid & 63, ft = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x / tiles, tile = blockIdx.x - g * tiles;
    if (ft == GRU_COMPUTE_WAVES) {                                     // the writer wave (see k_gru_seq_fwd)
        float* dgi_t = gru_part_base(d_gi, g) + (size_t)tile * 16 * G3 + 4 * lane;        // + t * t_stride
        float* dgh_t = dgh + (((size_t)g * T) * B + tile * 16) * G3 + 4 * lane;           // + t * B * G3
        for (int t = T - 1; t >= 0; --t) {
            lds_barrier();
            const float* im = &st[t & 1][0][0];
            f32x4 vg[12], vh[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                int r, c; piece_rc<G3>(i, lane, r, c);
                vg[i] = *reinterpret_cast<const f32x4*>(im + r * BST + c);                              // d_r | d_z | d_n
                vh[i] = *reinterpret_cast<const f32x4*>(im + r * BST + c + (c >= 2 * GH ? GH : 0));     // d_r | d_z | d_hn
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) *reinterpret_cast<f32x4*>(dgi_t + (size_t)t * d_gi.t_stride + 256 * i) = vg[i];
#pragma unroll
            for (int i = 0; i < 12; ++i) *reinterpret_cast<f32x4*>(dgh_t + (size_t)t * B * G3 + 256 * i) = vh[i];
        }
        return;
    }
    const int m = lane & 15, q = lane >> 4;
    const int row = tile * 16 + m, rc = row;                          // B is a multiple of 16 (the host pads): see k_gru_seq_fwd
    const int fo = 16 * ft + 4 * q;
    const float* __restrict__ wh = gru_wh(W, g);
    // resident A fragments of W_h: lane (q, m) = hidden feature 16 ft + m, reduction indices (gate outputs) k = 32 s + 8 q + j
    u32x4 wa[6][3];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        b8 t1, t2, t3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 x1, x2, x3;
            split3(wh[(size_t)(16 * ft + m) * G3 + 32 * s + 8 * q + j], x1, x2, x3);
            t1[j] = x1; t2[j] = x2; t3[j] = x3;
        }
        wa[s][0] = __builtin_bit_cast(u32x4, t1); wa[s][1] = __builtin_bit_cast(u32x4, t2); wa[s][2] = __builtin_bit_cast(u32x4, t3);
    }
    f32x4 dbh[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 carry = {0.f, 0.f, 0.f, 0.f};                                // dL/dh_t arriving from step t + 1
    // everything a step reads from global memory is requested PF steps ahead; the compute waves issue no stores (see k_gru_seq_fwd)
    constexpr int PF = 2;                                              // 2 steps ~ 2 us ahead; 4 would need 48 more registers than the 256 a wave has when 5 waves share 4 SIMDs
    struct StepIn { f32x4 dout, rg, zg, ng, gn, hprev; };
    const float* dbase = gru_part_base(dhs, g) + (size_t)rc * GH + fo;      // dL/dhs of this set: parts [sets, T, B, 64], + t * t_stride
    auto load_step = [&](int t, StepIn& in) {
        const size_t tr = ((size_t)t * G + g) * B + rc;
        in.dout = *reinterpret_cast<const f32x4*>(dbase + (size_t)t * dhs.t_stride);
        const float* s = rzn + tr * G3 + fo;
        in.rg = *reinterpret_cast<const f32x4*>(s); in.zg = *reinterpret_cast<const f32x4*>(s + GH); in.ng = *reinterpret_cast<const f32x4*>(s + 2 * GH);
        in.gn = *reinterpret_cast<const f32x4*>(ghn + tr * GH + fo);
        in.hprev = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) in.hprev = *reinterpret_cast<const f32x4*>(hs + (((size_t)g * T + (t - 1)) * B + rc) * GH + fo);
    };
    StepIn buf[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (T - 1 - d >= 0) load_step(T - 1 - d, buf[d]);
    auto step = [&](int t, StepIn& in) {
        const int pb = t & 1;
        const f32x4 dout = in.dout, rg = in.rg, zg = in.zg, ng = in.ng, gn = in.gn, hprev = in.hprev;
        if (t - PF >= 0) load_step(t - PF, in);
        f32x4 d_r, d_z, d_n, d_hn, direct;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dh = dout[r] + carry[r];
            d_n[r] = dh * (1.f - zg[r]) * (1.f - ng[r] * ng[r]);       // through tanh
            d_z[r] = dh * (hprev[r] - ng[r]) * zg[r] * (1.f - zg[r]);  // through sigmoid
            d_r[r] = d_n[r] * gn[r] * rg[r] * (1.f - rg[r]);
            d_hn[r] = d_n[r] * rg[r];                                  // dL/dgh_n
            direct[r] = dh * zg[r];
        }
        {   // this lane's 12 values as three bf16 terms (BF: one) -> the LDS image the other waves read their B operand from
            const f32x4 v[3] = {d_r, d_z, d_hn};
#pragma unroll
            for (int gate = 0; gate < 3; ++gate) {
                b4 t1, t2, t3;
#pragma unroll
                for (int r = 0; r < 4; ++r) { __bf16 x1, x2, x3; split3(v[gate][r], x1, x2, x3); t1[r] = x1; t2[r] = x2; t3[r] = x3; }
                *reinterpret_cast<u32x2*>(&dgb[pb][0][m][gate * GH + fo]) = __builtin_bit_cast(u32x2, t1);
                if constexpr (!BF) {
                    *reinterpret_cast<u32x2*>(&dgb[pb][1][m][gate * GH + fo]) = __builtin_bit_cast(u32x2, t2);
                    *reinterpret_cast<u32x2*>(&dgb[pb][2][m][gate * GH + fo]) = __builtin_bit_cast(u32x2, t3);
                }
            }
        }
        {   // the step's results -> staging image (the writer wave takes them to d_gi and dgh)
            float* o = &st[pb][m][fo];
            *reinterpret_cast<f32x4*>(o) = d_r; *reinterpret_cast<f32x4*>(o + GH) = d_z; *reinterpret_cast<f32x4*>(o + 2 * GH) = d_n;
            *reinterpret_cast<f32x4*>(o + 3 * GH) = d_hn;
        }
        dbh[0] += d_r; dbh[1] += d_z; dbh[2] += d_hn;
        lds_barrier();
        // dL/dh_{t-1}, matrix part: D[feature 16 ft + 4 q + reg][row m] = sum_k W_h[feature][k] dL/dgh[row][k]; partial products by
        // order of magnitude into three accumulators (smallest first when they are added)
        f32x4 a3 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            u32x4 bt[3];
#pragma unroll
            for (int term = 0; term < (BF ? 1 : 3); ++term) bt[term] = *reinterpret_cast<const u32x4*>(&dgb[pb][term][m][32 * s + 8 * q]);
            if constexpr (BF) {
                a1 = mmab(wa[s][0], bt[0], a1);
            } else {
                a3 = mmab(wa[s][0], bt[2], a3); a2 = mmab(wa[s][0], bt[1], a2); a1 = mmab(wa[s][0], bt[0], a1);
                a3 = mmab(wa[s][2], bt[0], a3); a2 = mmab(wa[s][1], bt[0], a2);
                a3 = mmab(wa[s][1], bt[1], a3);
            }
        }
        carry = direct + ((a3 + a2) + a1);
        // the next step writes the other LDS buffer; the one after next is ordered behind the next barrier
    };
    int t0 = T - 1;
    for (; t0 - (PF - 1) >= 0; t0 -= PF) {                             // whole groups: the PF register sets rotate by unrolling
#pragma unroll
        for (int d = 0; d < PF; ++d) step(t0 - d, buf[d]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (t0 - d >= 0) step(t0 - d, buf[d]);
    // dL/db_h: column sums over the tile's rows (the 16 lanes that share q)
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = dbh[gate][r];
#pragma unroll
            for (int sh = 8; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);
            if (m == 0) d_bh_part[((size_t)g * tiles + tile) * G3 + gate * GH + fo + r] = v;
        }
}

static GruParts gru_parts(float* const* parts, int n_parts, int T, int G, int B) {
    GruParts P;
    for (int k = 0; k < 4; ++k) P.p[k] = parts[k < n_parts ? k : 0];
    if (n_parts <= 0) {            // one time-major tensor [T, G, B, 192]
        P.spp = G; P.set_stride = (long)B * G3; P.t_stride = (long)G * B * G3;
    } else {                       // n_parts set-major tensors [G / n_parts, T, B, 192]
        P.spp = G / n_parts; P.set_stride = (long)T * B * G3; P.t_stride = (long)B * G3;
    }
    return P;
}

// n_parts 0: gi_parts[0] is one time-major tensor; else set-major parts (see GruParts)
static GruW gru_w(const float* const* wh_parts, const float* const* bh_parts, int n_wparts, int G) {
    GruW W;
    const int np = n_wparts < 1 ? 1 : n_wparts;
    for (int k = 0; k < 4; ++k) { W.wh[k] = wh_parts[k < np ? k : 0]; W.bh[k] = bh_parts ? bh_parts[k < np ? k : 0] : nullptr; }
    W.spp = G / np;
    return W;
}
void launch_gru_seq_fwd(const float* const* gi_parts, int n_parts, const float* const* wh_parts, const float* const* bh_parts, int n_wparts, float* hs,
                        float* rzn, float* ghn, int T, int G, int B, hipStream_t s) {
    const int tiles = (B + 15) / 16;
    const GruW W = gru_w(wh_parts, bh_parts, n_wparts, G);
    const GruParts P = gru_parts(const_cast<float* const*>(reinterpret_cast<const float* const*>(gi_parts)), n_parts, T, G, B);
    int32_t* err = numeric_err_word();
    const bool bf = learner_precision() == 1;
#define SSD_GF(TR_, BF_) hipLaunchKernelGGL((k_gru_seq_fwd<TR_, BF_>), dim3(G * tiles), dim3(GRU_THREADS), 0, s, P, W, hs, rzn, ghn, T, G, B, tiles, err)
    if (rzn) { if (bf) SSD_GF(true, true); else SSD_GF(true, false); }
    else { if (bf) SSD_GF(false, true); else SSD_GF(false, false); }
#undef SSD_GF
}
// Gn <= G: only the first Gn weight sets are walked (the sets whose outputs carry a gradient: the live net's; the target net's follow);
// rzn / ghn keep their [T, G, B, .] strides, dgh / d_wh / d_bh_part are [Gn, ...].  dhs_parts: n_parts (or one, n_parts 0) tensors
// [sets, T, B, 64]; parts past Gn are not read.
void launch_gru_seq_bwd(const float* const* dhs_parts, const float* hs, const float* rzn, const float* ghn, const float* const* wh_parts, int n_wparts,
                        float* const* d_gi_parts, int n_parts, float* dgh, float* d_wh, float* d_bh_part, int T, int G, int Gn, int B, hipStream_t s) {
    const int tiles = (B + 15) / 16;
    const GruW W = gru_w(wh_parts, nullptr, n_wparts, G);
    GruParts D;
    {
        const int np = n_parts < 1 ? 1 : n_parts;
        for (int k = 0; k < 4; ++k) D.p[k] = const_cast<float*>(dhs_parts[(k < np && k * (G / np) < Gn) ? k : 0]);
        D.spp = G / np; D.set_stride = (long)T * B * GH; D.t_stride = (long)B * GH;
    }
    const GruParts P = gru_parts(d_gi_parts, n_parts, T, G, B);
    static bool attr_done_dev[64] = {};                                // the attribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_done_dev[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_seq_bwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, GRU_BWD_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_seq_bwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, GRU_BWD_LDS);
        attr_done_dev[dev] = true;                                     // a refused attribute shows as a launch error (ssd_poll_error / hipGetLastError)
    }
    if (learner_precision() == 1) hipLaunchKernelGGL(k_gru_seq_bwd<true>, dim3(Gn * tiles), dim3(GRU_THREADS), GRU_BWD_LDS, s, D, hs, rzn, ghn, W, P, dgh, d_bh_part, T, G, B, tiles);
    else hipLaunchKernelGGL(k_gru_seq_bwd<false>, dim3(Gn * tiles), dim3(GRU_THREADS), GRU_BWD_LDS, s, D, hs, rzn, ghn, W, P, dgh, d_bh_part, T, G, B, tiles);
    // dL/dW_h[g] = sum_{t >= 1, b} h_{t-1}[b]^T dL/dgh_t[b]: rows (t, b) of hs [G, T, B, 64] against rows (t + 1, b) of dgh [G, T, B, 192] --
    // the x^T g role of the per-agent-layer kernel (csrc/ssd_bmm.hip: K = (T - 1) B rows split over 16 waves per tile, exact f32)
    if (T > 1) launch_bias_bmm_bwd(dgh + (size_t)B * G3, hs, nullptr, nullptr, d_wh, nullptr, nullptr, Gn, (T - 1) * B, GH, G3, s, (long)T * B * GH, (long)T * B * G3);
    else (void)hipMemsetAsync(d_wh, 0, (size_t)Gn * GH * G3 * sizeof(float), s);
}

}  // namespace ssd
