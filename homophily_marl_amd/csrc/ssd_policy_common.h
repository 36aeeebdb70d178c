// ssd_policy_common.h -- device helpers shared by the rollout-time controller kernels (ssd_policy.hip, ssd_policy_fused.hip).
#pragma once
#include "ssd_device.h"

namespace ssd {

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : 0.01f * x; }   // nn.LeakyReLU default slope

__device__ __forceinline__ uint32_t mix32p(uint32_t x) {
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    return x;
}

// Dueling head + epsilon-greedy on one row: a[0..A) advantages, v the state value (homophily_agent.py:168-170;
// action_selectors.py:44-68).  The exploration draws are keyed by (seed, step, r).  Returns the action; q (nullable) gets Q.
__device__ __forceinline__ int dueling_pick_row(const float* a, float v, int A, const uint8_t* avail, float eps, uint32_t step,
                                                uint32_t seed, uint32_t r, float* q_out) {
    float mean = 0.f;
    for (int k = 0; k < A; ++k) mean += a[k];
    mean /= (float)A;
    int best = 0, navail = 0;
    float bq = -INFINITY;
    for (int k = 0; k < A; ++k) {
        const float q = v + a[k] - mean;
        if (q_out) q_out[k] = q;
        const bool ok = !avail || avail[k];
        navail += ok;
        if (ok && q > bq) { bq = q; best = k; }          // first maximum, like torch.max / argmax
    }
    const uint32_t x0 = mix32p(seed ^ mix32p(step * 0x9E3779B9u + r));
    const uint32_t x1 = mix32p(x0 ^ 0x85EBCA6Bu);
    int act = best;
    if ((float)(x0 >> 8) * (1.0f / 16777216.0f) < eps) {
        int pick = (int)(((uint64_t)x1 * (uint32_t)navail) >> 32);   // uniform over the available actions
        for (int k = 0; k < A; ++k) { const bool ok = !avail || avail[k]; if (ok) { if (pick == 0) { act = k; break; } --pick; } }
    }
    return act;
}

}  // namespace ssd
