// ssd_gru_seq.hip -- the learner's recurrence as ONE forward and ONE backward launch over all T timesteps.
//
// Reference: HomophilyLearner.train evaluates the controller for t = 0..T-1 from a zero hidden state (homophily_learner.py:68-91),
// each step running the hand-written GRU cell of HomophilyAgent (homophily_agent.py:162-165,188-191):
//     r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z h,   gh = h W_h + b_h
// The input-side projections gi = x W_i + b_i do not depend on the recurrence and are computed for all T at once by the caller
// (one batched GEMM); what is left per step is a [B, 64] x [64, 192] product and the gate arithmetic -- at the learner's batch
// (B = 16 episodes) a latency chain of ~10 tiny launches per step and direction.  Here one workgroup owns one (weight set g,
// 16-row tile) and walks the whole sequence:
//   * 4 waves, wave w owns hidden features 16 w .. 16 w + 15 (all three gates); its slice of W_h lives in registers for all T
//     (48 VGPRs as MFMA A operand), products are v_mfma_f32_16x16x4_f32 in the transposed form of ssd_policy_fused.hip
//     (activation row on the lane, 4 consecutive features in the lane's registers), so the gate arithmetic is lane-local;
//   * the new state is exchanged through a double-buffered LDS tile: one barrier per step;
//   * backward walks t = T-1 .. 0 with the carried dL/dh in registers, dL/dW_h accumulates in MFMA accumulators over all T
//     (A = h_{t-1}^T, B = dL/dgh_t read k-major from LDS) and is written once.
// H = 64 is fixed.  Rows are independent sequences: any B (tiles of 16, the last one masked).
#include "ssd_policy_common.h"

namespace ssd {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int GH = 64, G3 = 192, HS = 68, DS = 196;     // hidden, 3 * hidden, LDS row strides (floats)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }   // as k_gru_fwd_train

// gi [T, G, B, 192], wh [G, 64, 192], bh [G, 192] -> hs [G, T, B, 64]; optional (training) rzn [T, G, B, 192], ghn [T, G, B, 64]
__global__ __launch_bounds__(256) void k_gru_seq_fwd(const float* __restrict__ gi, const float* __restrict__ wh, const float* __restrict__ bh,
                                                     float* __restrict__ hs, float* __restrict__ rzn, float* __restrict__ ghn, int T, int G,
                                                     int B, int tiles) {
    __shared__ float hbuf[2][16][HS];
    const int tid = threadIdx.x, lane = tid & 63, ft = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int g = blockIdx.x / tiles, tile = blockIdx.x - g * tiles;
    const int row = tile * 16 + m;
    const bool valid = row < B;
    const int rc = valid ? row : B - 1;
    // resident slice of W_h^T: wa[gate][ct][r] = W_h[k = 16 ct + 4 q + r][gate * 64 + 16 ft + m]
    float wa[3][4][4];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) wa[gate][ct][r] = wh[((size_t)g * GH + 16 * ct + 4 * q + r) * G3 + gate * GH + 16 * ft + m];
    f32x4 bias[3];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) bias[gate] = *reinterpret_cast<const f32x4*>(bh + (size_t)g * G3 + gate * GH + 16 * ft + 4 * q);
    f32x4 hp[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) hp[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fo = 16 * ft + 4 * q;                                   // this lane's 4 features
    for (int t = 0; t < T; ++t) {
        const size_t tr = ((size_t)t * G + g) * B + rc;                // row of the [T, G, B, .] tensors
        const float* gir = gi + tr * G3 + fo;
        const f32x4 gr = *reinterpret_cast<const f32x4*>(gir), gz = *reinterpret_cast<const f32x4*>(gir + GH),
                    gn = *reinterpret_cast<const f32x4*>(gir + 2 * GH);
        f32x4 acc[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int gate = 0; gate < 3; ++gate)
                    acc[gate] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[gate][ct][r], hp[ct][r], acc[gate], 0, 0, 0);
        // own previous state: hp[ft] holds features 16 ft + 4 q + r -- ft is wave-uniform, select without dynamic indexing
        const f32x4 hown = ft == 0 ? hp[0] : ft == 1 ? hp[1] : ft == 2 ? hp[2] : hp[3];
        f32x4 hn, rg, zg, ng;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rg[r] = sigm(gr[r] + acc[0][r]);
            zg[r] = sigm(gz[r] + acc[1][r]);
            ng[r] = tanhf(gn[r] + rg[r] * acc[2][r]);
            hn[r] = (1.f - zg[r]) * ng[r] + zg[r] * hown[r];
        }
        if (valid) {
            *reinterpret_cast<f32x4*>(hs + (((size_t)g * T + t) * B + row) * GH + fo) = hn;
            if (rzn) {
                float* s = rzn + tr * G3 + fo;
                *reinterpret_cast<f32x4*>(s) = rg; *reinterpret_cast<f32x4*>(s + GH) = zg; *reinterpret_cast<f32x4*>(s + 2 * GH) = ng;
                *reinterpret_cast<f32x4*>(ghn + tr * GH + fo) = acc[2];
            }
        }
        *reinterpret_cast<f32x4*>(&hbuf[t & 1][m][fo]) = hn;
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) hp[ct] = *reinterpret_cast<const f32x4*>(&hbuf[t & 1][m][16 * ct + 4 * q]);
    }
}

// dhs [G, T, B, 64] (dL/d hs), hs, rzn, ghn, wh as above -> d_gi [T, G, B, 192], d_wh_part [G, tiles, 64, 192], d_bh_part [G, tiles, 192]
__global__ __launch_bounds__(256) void k_gru_seq_bwd(const float* __restrict__ dhs, const float* __restrict__ hs, const float* __restrict__ rzn,
                                                     const float* __restrict__ ghn, const float* __restrict__ wh, float* __restrict__ d_gi,
                                                     float* __restrict__ d_wh_part, float* __restrict__ d_bh_part, int T, int G, int B, int tiles) {
    __shared__ float dg[2][16][DS];     // dL/dgh of the step, [row][192]
    __shared__ float hb[2][16][HS];     // h_{t-1}, [row][64]
    const int tid = threadIdx.x, lane = tid & 63, ft = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int g = blockIdx.x / tiles, tile = blockIdx.x - g * tiles;
    const int row = tile * 16 + m;
    const bool valid = row < B;
    const int rc = valid ? row : B - 1;
    const int fo = 16 * ft + 4 * q;
    // resident slice of W_h for dL/dh_{t-1} = dL/dgh W_h^T: A[m = hidden feature 16 ft + m][k = gate output 16 c + 4 q + r]
    f32x4 wd[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) wd[c] = *reinterpret_cast<const f32x4*>(wh + ((size_t)g * GH + 16 * ft + m) * G3 + 16 * c + 4 * q);
    f32x4 aw[3][4];                                                    // dL/dW_h[16 ht + 4 q + reg][16 (3 ft + o) + m]
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) aw[o][ht] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dbh[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 carry = {0.f, 0.f, 0.f, 0.f};                                // dL/dh_t arriving from step t + 1
    for (int t = T - 1; t >= 0; --t) {
        const int pb = t & 1;
        const size_t tr = ((size_t)t * G + g) * B + rc;
        const f32x4 dout = *reinterpret_cast<const f32x4*>(dhs + (((size_t)g * T + t) * B + rc) * GH + fo);
        const float* s = rzn + tr * G3 + fo;
        const f32x4 rg = *reinterpret_cast<const f32x4*>(s), zg = *reinterpret_cast<const f32x4*>(s + GH), ng = *reinterpret_cast<const f32x4*>(s + 2 * GH);
        const f32x4 gn = *reinterpret_cast<const f32x4*>(ghn + tr * GH + fo);
        f32x4 hprev = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) hprev = *reinterpret_cast<const f32x4*>(hs + (((size_t)g * T + (t - 1)) * B + rc) * GH + fo);
        f32x4 d_r, d_z, d_n, d_hn, direct;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dh = valid ? dout[r] + carry[r] : 0.f;
            d_n[r] = dh * (1.f - zg[r]) * (1.f - ng[r] * ng[r]);       // through tanh
            d_z[r] = dh * (hprev[r] - ng[r]) * zg[r] * (1.f - zg[r]);  // through sigmoid
            d_r[r] = d_n[r] * gn[r] * rg[r] * (1.f - rg[r]);
            d_hn[r] = d_n[r] * rg[r];                                  // dL/dgh_n
            direct[r] = dh * zg[r];
        }
        if (valid) {
            float* o = d_gi + tr * G3 + fo;
            *reinterpret_cast<f32x4*>(o) = d_r; *reinterpret_cast<f32x4*>(o + GH) = d_z; *reinterpret_cast<f32x4*>(o + 2 * GH) = d_n;
        }
        dbh[0] += d_r; dbh[1] += d_z; dbh[2] += d_hn;
        *reinterpret_cast<f32x4*>(&dg[pb][m][fo]) = d_r;
        *reinterpret_cast<f32x4*>(&dg[pb][m][GH + fo]) = d_z;
        *reinterpret_cast<f32x4*>(&dg[pb][m][2 * GH + fo]) = d_hn;
        *reinterpret_cast<f32x4*>(&hb[pb][m][fo]) = hprev;
        __syncthreads();
        // dL/dh_{t-1}, matrix part: D[feature 16 ft + 4 q + reg][row m] = sum_k W_h[feature][k] dL/dgh[row][k]  (4 chains of 12)
        f32x4 ah[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(&dg[pb][m][16 * c + 4 * q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) ah[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[c][r], bq[r], ah[r], 0, 0, 0);
        }
        carry = direct + (ah[0] + ah[1]) + (ah[2] + ah[3]);
        // dL/dW_h += h_{t-1}^T dL/dgh: A[m = hidden feature][k = row], B[k = row][n = gate output]; this wave: outputs 48 ft .. 48 ft + 47
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) {
            float a_h[4], b_d[3];
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) a_h[ht] = hb[pb][4 * sk + q][16 * ht + m];
#pragma unroll
            for (int o = 0; o < 3; ++o) b_d[o] = dg[pb][4 * sk + q][16 * (3 * ft + o) + m];
#pragma unroll
            for (int o = 0; o < 3; ++o)
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) aw[o][ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_h[ht], b_d[o], aw[o][ht], 0, 0, 0);
        }
        // the next step writes the other LDS buffers; the one after next is ordered behind the next barrier
    }
    float* wp = d_wh_part + ((size_t)g * tiles + tile) * GH * G3;
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int ht = 0; ht < 4; ++ht)
#pragma unroll
            for (int r = 0; r < 4; ++r) wp[(size_t)(16 * ht + 4 * q + r) * G3 + 16 * (3 * ft + o) + m] = aw[o][ht][r];
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

void launch_gru_seq_fwd(const float* gi, const float* wh, const float* bh, float* hs, float* rzn, float* ghn, int T, int G, int B, hipStream_t s) {
    const int tiles = (B + 15) / 16;
    hipLaunchKernelGGL(k_gru_seq_fwd, dim3(G * tiles), dim3(256), 0, s, gi, wh, bh, hs, rzn, ghn, T, G, B, tiles);
}
void launch_gru_seq_bwd(const float* dhs, const float* hs, const float* rzn, const float* ghn, const float* wh, float* d_gi, float* d_wh_part,
                        float* d_bh_part, int T, int G, int B, hipStream_t s) {
    const int tiles = (B + 15) / 16;
    hipLaunchKernelGGL(k_gru_seq_bwd, dim3(G * tiles), dim3(256), 0, s, dhs, hs, rzn, ghn, wh, d_gi, d_wh_part, d_bh_part, T, G, B, tiles);
}

}  // namespace ssd
