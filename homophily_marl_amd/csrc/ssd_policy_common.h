// ssd_policy_common.h -- device helpers shared by the rollout-time controller kernels (ssd_policy.hip, ssd_policy_mfma.hip, ssd_gru_seq.hip).
#pragma once
#include "ssd_device.h"

namespace ssd {

// nn.LeakyReLU, default slope: x > 0 ? x : 0.01 x == max(x, 0.01 x) for every finite x and both zeros (one v_max instead of a
// compare + select per value; the compiler may not make this substitution itself under IEEE NaN rules; values here are never NaN)
__device__ __forceinline__ float leaky(float x) {
    const float y = 0.01f * x;
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));          // fmaxf() adds a canonicalising v_max per operand
    return r;
#else
    return x > y ? x : y;
#endif
}

__device__ __forceinline__ uint32_t mix32p(uint32_t x) {
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    return x;
}

// Dueling head + epsilon-greedy on one row: a[0..A) advantages, v the state value (homophily_agent.py:168-170;
// action_selectors.py:44-68).  The exploration draws are keyed by (seed, step, r).  Returns the action; q (nullable) gets Q.
// avail_bits: bit k set = action k available (all ones: no mask).  AT > 0: compile-time action count (the loops unroll and `a`
// stays in registers); AT = 0: runtime A.
// QP: the type of the q pointer (a plain float*, or a pointer that carries its address space)
template <int AT = 0, typename QP = float*>
__device__ __forceinline__ int dueling_pick_bits(const float* a, float v, int A_rt, uint32_t avail_bits, float eps, uint32_t step,
                                                 uint32_t seed, uint32_t r, QP q_out) {
    const int A = AT ? AT : A_rt;
    float mean = 0.f;
#pragma unroll
    for (int k = 0; k < A; ++k) mean += a[k];
    mean /= (float)A;
    int best = 0;
    float bq = -INFINITY;
#pragma unroll
    for (int k = 0; k < A; ++k) {
        const float q = v + a[k] - mean;
        if (q_out) q_out[k] = q;
        const bool ok = (avail_bits >> k) & 1u;
        if (ok && q > bq) { bq = q; best = k; }          // first maximum, like torch.max / argmax
    }
    const uint32_t live = avail_bits & (A >= 32 ? 0xFFFFFFFFu : (1u << A) - 1u);
    const uint32_t x0 = mix32p(seed ^ mix32p(step * 0x9E3779B9u + r));
    const uint32_t x1 = mix32p(x0 ^ 0x85EBCA6Bu);
    int act = best;
    if ((float)(x0 >> 8) * (1.0f / 16777216.0f) < eps) {
        int pick = (int)(((uint64_t)x1 * (uint32_t)__builtin_popcount(live)) >> 32);   // uniform over the available actions
        uint32_t mset = live;
        for (; pick > 0; --pick) mset &= mset - 1u;      // drop the `pick` lowest available actions
        act = mset ? __builtin_ctz(mset) : best;
    }
    return act;
}
__device__ __forceinline__ uint32_t avail_to_bits(const uint8_t* avail, int A) {
    uint32_t bits = 0xFFFFFFFFu;
    if (avail) {   // 16 independent byte loads (clamped index: no branch, nothing read past the array), one wait
        uint8_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = avail[k < A ? k : A - 1];
        bits = 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) bits |= ((k < A && v[k]) ? 1u : 0u) << k;
    }
    return bits;
}
__device__ __forceinline__ int dueling_pick_row(const float* a, float v, int A, const uint8_t* avail, float eps, uint32_t step,
                                                uint32_t seed, uint32_t r, float* q_out) {
    return dueling_pick_bits<0>(a, v, A, avail_to_bits(avail, A), eps, step, seed, r, q_out);
}

}  // namespace ssd
