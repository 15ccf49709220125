// pcc_policy.hip -- the rollout half of the on-device PPO caller: the
// reference's policy (src/gym/stable_solve.py:39-45: separate pi / vf MLPs, hidden sizes --arch = 32,16,
// tanh, a state-independent log-std Gaussian head) evaluated for a whole env batch in ONE launch --
// action mean, sampled action, its log-probability and the value estimate -- so that the env half is
// not starved by a dozen small framework launches per step.  One lane per env; the few thousand
// parameters sit in LDS, every lane walks them in the same order (broadcast reads, no bank conflicts);
// fp32 like the framework path it replaces.  No MFMA: 65 536 x ~3 kFLOP is microseconds of plain FMAs.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pcc_policy.h"

namespace {

constexpr int kMaxParams = 8192;   // floats of both networks
constexpr int kMaxHidden = 64;

// parameter block, floats: pi {W1[h1][D], b1[h1], W2[h2][h1], b2[h2], W3[h2], b3, log_std}, then vf {same without log_std}
__device__ __forceinline__ float mlp_forward(const float *p, const float *x, int D, int h1, int h2, float *z1, float *z2) {
    const float *W1 = p, *b1 = W1 + h1 * D, *W2 = b1 + h1, *b2 = W2 + h2 * h1, *W3 = b2 + h2, *b3 = W3 + h2;
    for (int j = 0; j < h1; j++) {
        float s = b1[j];
        for (int k = 0; k < D; k++) s = fmaf(W1[j * D + k], x[k], s);
        z1[j] = tanhf(s);
    }
    for (int j = 0; j < h2; j++) {
        float s = b2[j];
        for (int k = 0; k < h1; k++) s = fmaf(W2[j * h1 + k], z1[k], s);
        z2[j] = tanhf(s);
    }
    float out = b3[0];
    for (int k = 0; k < h2; k++) out = fmaf(W3[k], z2[k], out);
    return out;
}

template <int D>
__global__ __launch_bounds__(256) void policy_act_kernel(const float *obs, int64_t n, const float *params, int n_params,
                                                         int h1, int h2, const float *noise, float *mean_out,
                                                         float *act_out, float *logp_out, float *value_out) {
    __shared__ float sp[kMaxParams];
    for (int k = threadIdx.x; k < n_params; k += blockDim.x) sp[k] = params[k];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[D];
#pragma unroll
    for (int k = 0; k < D; k++) x[k] = obs[i * D + k];
    float z1[kMaxHidden], z2[kMaxHidden];
    const int n_pi = h1 * D + h1 + h2 * h1 + h2 + h2 + 1;   // without log_std
    const float mu = mlp_forward(sp, x, D, h1, h2, z1, z2);
    const float log_std = sp[n_pi];
    const float v = mlp_forward(sp + n_pi + 1, x, D, h1, h2, z1, z2);
    const float eps = noise ? noise[i] : 0.0f;
    const float a = mu + expf(log_std) * eps;
    // log N(a; mu, sigma) = -eps^2 / 2 - log_std - log(2 pi) / 2
    if (mean_out) mean_out[i] = mu;
    if (act_out) act_out[i] = a;
    if (logp_out) logp_out[i] = -0.5f * eps * eps - log_std - 0.918938533204672742f;
    if (value_out) value_out[i] = v;
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) by the hardware's exp2 and reciprocal: absolute error ~1e-7, saturates cleanly -- the same
// function the gradient kernel evaluates (pcc_ppo.hip: the rollout's and the update's forward agree), a fifth of libm's tanhf
// in instructions (48 of them per network and env: half of the fixed kernel's time went into them)
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// The reference's own sizes (--arch 32,16) with everything a compile-time constant: the hidden activations stay in
// registers (the generic kernel above indexes z1[j] with a run-time j: scratch memory), the loops unroll.
template <int D, int H1, int H2>
__device__ __forceinline__ float mlp_forward_fixed(const float *p, const float (&x)[D]) {
    const float *W1 = p, *b1 = W1 + H1 * D, *W2 = b1 + H1, *b2 = W2 + H2 * H1, *W3 = b2 + H2, *b3 = W3 + H2;
    float z1[H1], z2[H2];
#pragma unroll
    for (int j = 0; j < H1; j++) {
        float s = b1[j];
#pragma unroll
        for (int k = 0; k < D; k++) s = fmaf(W1[j * D + k], x[k], s);
        z1[j] = tanh_fast(s);
    }
#pragma unroll
    for (int j = 0; j < H2; j++) {
        float s = b2[j];
#pragma unroll
        for (int k = 0; k < H1; k++) s = fmaf(W2[j * H1 + k], z1[k], s);
        z2[j] = tanh_fast(s);
    }
    float out = b3[0];
#pragma unroll
    for (int k = 0; k < H2; k++) out = fmaf(W3[k], z2[k], out);
    return out;
}

// Weights straight from the parameter block with compile-time offsets: every lane reads the same address, so the loads are
// scalar (s_load into SGPRs, which the FMAs take as operands) -- no LDS copy, no LDS read per FMA (the first form of this
// kernel staged the parameters in LDS and read one weight per FMA from there: 32 us for 65 536 envs, LDS-issue-bound with one
// wavefront per SIMD).  The two networks of an env run in two lanes of different workgroups (blockIdx.y = 0: pi -> mean,
// action, log-probability; 1: vf -> value): twice the wavefronts, half the chain.
template <int D, int H1, int H2>
__global__ __launch_bounds__(256) void policy_act_fixed_kernel(const float *__restrict__ obs, int64_t n,
                                                               const float *__restrict__ params, int n_params,
                                                               const float *__restrict__ noise, float *__restrict__ mean_out,
                                                               float *__restrict__ act_out, float *__restrict__ logp_out,
                                                               float *__restrict__ value_out) {
    (void)n_params;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[D];
#pragma unroll
    for (int k = 0; k < D; k++) x[k] = obs[i * D + k];
    constexpr int n_pi = H1 * D + H1 + H2 * H1 + H2 + H2 + 1;   // without log_std
    if (blockIdx.y == 0) {
        const float mu = mlp_forward_fixed<D, H1, H2>(params, x);
        const float log_std = params[n_pi];
        const float eps = noise ? noise[i] : 0.0f;
        if (mean_out) mean_out[i] = mu;
        if (act_out) act_out[i] = mu + expf(log_std) * eps;
        if (logp_out) logp_out[i] = -0.5f * eps * eps - log_std - 0.918938533204672742f;
    } else {
        const float v = mlp_forward_fixed<D, H1, H2>(params + n_pi + 1, x);
        if (value_out) value_out[i] = v;
    }
}

}  // namespace

extern "C" int pcc_policy_act(const float *obs, int64_t n_envs, int obs_dim, const float *params, int h1, int h2,
                              const float *noise, float *mean_out, float *act_out, float *logp_out, float *value_out,
                              void *stream) {
    if (!obs || !params || n_envs < 1) return -1;
    if (h1 < 1 || h2 < 1 || h1 > kMaxHidden || h2 > kMaxHidden) return -1;
    const int n_net = h1 * obs_dim + h1 + h2 * h1 + h2 + h2 + 1;
    const int n_params = 2 * n_net + 1;
    if (n_params > kMaxParams) return -1;
    const dim3 grid((unsigned)((n_envs + 255) / 256)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (h1 == 32 && h2 == 16) {   // the reference's --arch: the fully unrolled build
        switch (obs_dim) {
#define PCC_POLICY_FIXED(DD)                                                                                             \
    case DD:                                                                                                             \
        hipLaunchKernelGGL((policy_act_fixed_kernel<DD, 32, 16>), dim3(grid.x, 2), block, 0, st, obs, n_envs, params, n_params, noise, \
                           mean_out, act_out, logp_out, value_out);                                                     \
        return hipGetLastError() == hipSuccess ? 0 : -3;
            PCC_POLICY_FIXED(30)
            PCC_POLICY_FIXED(36)
            PCC_POLICY_FIXED(3)
            PCC_POLICY_FIXED(6)
            PCC_POLICY_FIXED(12)
            PCC_POLICY_FIXED(60)
#undef PCC_POLICY_FIXED
            default: break;
        }
    }
    switch (obs_dim) {   // the observation length is a compile-time constant of the unrolled loads
#define PCC_POLICY_CASE(DD)                                                                                              \
    case DD:                                                                                                             \
        hipLaunchKernelGGL(policy_act_kernel<DD>, grid, block, 0, st, obs, n_envs, params, n_params, h1, h2, noise, mean_out, \
                           act_out, logp_out, value_out);                                                               \
        break;
        PCC_POLICY_CASE(30)   // history 10 x 3 features: the reference's default observation (ns:382-388)
        PCC_POLICY_CASE(36)   // history 3 x all 12 features
        PCC_POLICY_CASE(3)
        PCC_POLICY_CASE(6)
        PCC_POLICY_CASE(12)
        PCC_POLICY_CASE(60)
#undef PCC_POLICY_CASE
        default: return -2;   // observation length without an instantiation: the caller falls back to the framework path
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
