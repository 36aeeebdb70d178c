// ssd_learner.hip -- the two integer/elementwise pieces of the homophily learner that sit on the data path.
//   k_build_inputs        HomophilyMAC._build_inputs tail   (controllers/homophily_controller.py:137-184)
//   k_incentive_transfer  incentive reward transfer         (learners/homophily_learner.py:94-115)
// Both are HBM-bound elementwise kernels over [B(*T)*n] rows; one thread per output element / row.
#include "ssd_device.h"

namespace ssd {

__global__ void k_build_inputs(int32_t rows, int32_t n, int32_t A, int32_t t0, const int64_t* __restrict__ last_actions,
                               const float* __restrict__ last_reward, const int64_t* __restrict__ last_actions_inc,
                               const float* __restrict__ pos, float pos_scale, float* __restrict__ out,
                               int32_t out_stride, int32_t out_offset) {
    const int width = A + n + 4;
    const size_t total = (size_t)rows * width;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / width), k = (int)(idx - (size_t)row * width);
        const int b = row / n, i = row - b * n;
        const int agent_major = t0 & 2;
        const int t_zero = t0 & 1;
        float v;
        if (k < A) v = (!t_zero && last_actions[row] == k) ? 1.f : 0.f;                  // one-hot of the last action (:137-141)
        else if (k < A + n) v = (k - A == i) ? 1.f : 0.f;                            // agent id (:142-143)
        else if (k == A + n) {                                                       // sign(last reward) (:145-150)
            const float r = t_zero ? 0.f : last_reward[row];
            v = (float)((r > 0.f) - (r < 0.f));
        } else if (k == A + n + 1) {                                                 // sign(#recv+ - #recv-) (:152-164)
            int recv = 0;
            if (!t_zero)
                for (int g = 0; g < n; ++g) {
                    if (g == i) continue;                                            // inc_mask_actions: no self incentive
                    const int64_t x = last_actions_inc[((size_t)b * n + g) * n + i];
                    recv += (x == 1) - (x == 2);
                }
            v = (float)((recv > 0) - (recv < 0));
        } else v = pos[(size_t)row * 2 + (k - A - n - 2)] / pos_scale;               // pos / ||(H, W)|| (:179-181)
        const size_t orow = agent_major ? (size_t)i * (rows / n) + b : (size_t)row;
        out[orow * out_stride + out_offset + k] = v;
    }
}

__global__ void k_incentive_transfer(int32_t B, int32_t T, int32_t n, const int64_t* __restrict__ a_inc,
                                     const float* __restrict__ rewards, float effect_ratio, float cost_ratio,
                                     float incentive, float seq_len, float* __restrict__ give, float* __restrict__ recv_pos,
                                     float* __restrict__ recv_neg, float* __restrict__ recv_zero, float* __restrict__ r_env,
                                     float* __restrict__ r_inc) {
    const size_t total = (size_t)B * T * n;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
        const size_t bt = it / n;
        const int i = (int)(it - bt * n);
        const int b = (int)(bt / T), t = (int)(bt - (size_t)b * T);
        const int64_t* m = a_inc + bt * n * n;
        int g = 0, rp = 0, rn = 0;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            g += m[(size_t)i * n + j] != 0;            // give: i -> j  (:101)
            const int64_t x = m[(size_t)j * n + i];    // receive: j -> i (:102-103)
            rp += x == 1; rn += x == 2;
        }
        recv_pos[it] = (float)rp; recv_neg[it] = (float)rn; recv_zero[it] = (float)(n - 1 - rp - rn);
        if (t < T - 1) {
            const size_t i1 = ((size_t)b * (T - 1) + t) * n + i;
            give[i1] = (float)g;
            const float r = rewards[i1];
            r_env[i1] = (r + (float)(rp - rn) * effect_ratio * incentive) / seq_len;   // :113
            r_inc[i1] = (r - (float)g * cost_ratio * incentive) / seq_len;             // :114
        }
    }
}

static int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : b > 4096 ? 4096 : b);
}

void launch_build_inputs(int32_t batch, int32_t n, int32_t A, int32_t t0, const int64_t* last_actions,
                         const float* last_reward, const int64_t* last_actions_inc, const float* pos, float pos_scale,
                         float* out, int32_t out_stride, int32_t out_offset, hipStream_t stream) {
    const size_t total = (size_t)batch * n * (A + n + 4);
    hipLaunchKernelGGL(k_build_inputs, dim3(grid_for(total)), dim3(256), 0, stream, batch * n, n, A, t0, last_actions,
                       last_reward, last_actions_inc, pos, pos_scale, out, out_stride, out_offset);
}

void launch_incentive_transfer(int32_t B, int32_t T, int32_t n, const int64_t* a_inc, const float* rewards,
                               float effect_ratio, float cost_ratio, float incentive, float seq_len, float* give,
                               float* recv_pos, float* recv_neg, float* recv_zero, float* r_env, float* r_inc,
                               hipStream_t stream) {
    const size_t total = (size_t)B * T * n;
    hipLaunchKernelGGL(k_incentive_transfer, dim3(grid_for(total)), dim3(256), 0, stream, B, T, n, a_inc, rewards,
                       effect_ratio, cost_ratio, incentive, seq_len, give, recv_pos, recv_neg, recv_zero, r_env, r_inc);
}

}  // namespace ssd
