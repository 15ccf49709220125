// pcc_ppo.hip -- the update half of the on-device PPO caller: one optimiser step of
// PPO1's objective (clipped surrogate + 0.5 * value error - ent_coef * entropy; stable-baselines' defaults as
// src/gym/stable_solve.py:52 uses them) on one minibatch of the rollout, for the reference's policy shape
// (src/gym/stable_solve.py:39-45: pi and vf MLPs, obs -> 32 -> 16 -> 1, tanh, state-independent log-std), as TWO
// launches instead of the ~150 small framework launches of forward + autograd + Adam:
//
//   ppo_grad_kernel   lane = sample.  Forward and the activation gradients of both networks are plain fp32 FMAs with the
//                     weights as scalar operands (uniform loads).  The weight gradients are contractions over the
//                     samples, dW1 = dZ1^T X, dW2 = dZ2^T H1: the 64 samples of a wavefront's tile are the K dimension of
//                     fp32 MFMAs (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact fp32, the vector rate -- what
//                     they buy here is the cross-lane sum), their operands transposed through LDS (row stride 33 / 17
//                     floats: conflict-free both ways).  The accumulators stay in registers over all tiles of a
//                     wavefront; every workgroup writes ONE partial gradient.
//   ppo_adam_kernel   sums the partial gradients in a fixed order (deterministic), adds the entropy
//                     term, applies Adam (torch.optim.Adam's arithmetic), writes the parameters in place.
//
// Plus the GAE recursion as one launch (thread = env, T steps backwards).  fp32 like the framework path; checked against
// torch autograd in tests/test_ppo.py.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pcc_policy.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 2;
constexpr int kMaxBlocks = 1024;  // 2 wavefronts each: two per SIMD of the chip (<= 256 registers, 17 KB of LDS per wavefront)

template <int D, int H1, int H2>
struct Net {  // offsets inside one network's block of the parameter vector (include/pcc_policy.h)
    static constexpr int W1 = 0, B1 = H1 * D, W2 = B1 + H1, B2 = W2 + H2 * H1, W3 = B2 + H2, B3 = W3 + H2, N = B3 + 1;
};

// tanh(x) = 1 - 2 / (exp(2x) + 1): absolute error ~1e-7 (one rounding of the quotient against 1), saturates cleanly
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// The weights are scalar operands of the FMAs (uniform loads straight from the parameter vector: no vector register, no
// LDS bandwidth -- broadcast reads of an LDS copy cost 1 KB of LDS return path per 4 weights and made the kernel LDS-bound).
// Left alone the compiler hoists these ~3 000 loop-invariant scalar loads out of the tile loop -- or, kept inside, to the
// top of the iteration -- and spills the scalar registers into vector-register lanes (1 500 - 7 000 v_readlane per tile):
// p, through an offset the compiler cannot see through (a scalar 0 made by an asm statement) that does not exist before
// the value `after` does: the loads through the result stay behind whatever produced `after`, and scalar
__device__ __forceinline__ const float *not_before(const float *p, float after) {
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero) : "v"(after));
    return p + zero;
}

template <int D, int H1, int H2>
__device__ __forceinline__ float net_forward(const float *__restrict__ p0, const float (&x)[32], float (&h1)[H1],
                                             float (&h2)[H2]) {
    using L = Net<D, H1, H2>;
    // the loads of unit j may start once unit j - 2 is done: one unit's weights run ahead of the FMAs
#pragma unroll
    for (int j = 0; j < H1; j++) {
        const float *p = not_before(p0, j >= 2 ? h1[j - 2] : x[0]);
        float s = p[L::B1 + j];
#pragma unroll
        for (int k = 0; k < D; k++) s = fmaf(p[L::W1 + j * D + k], x[k], s);
        h1[j] = tanh_fast(s);
    }
#pragma unroll
    for (int j = 0; j < H2; j++) {
        const float *p = not_before(p0, j >= 2 ? h2[j - 2] : h1[H1 - 2 + j]);
        float s = p[L::B2 + j];
#pragma unroll
        for (int k = 0; k < H1; k++) s = fmaf(p[L::W2 + j * H1 + k], h1[k], s);
        h2[j] = tanh_fast(s);
    }
    const float *p = not_before(p0, h2[H2 - 2]);
    float out = p[L::B3];
#pragma unroll
    for (int k = 0; k < H2; k++) out = fmaf(p[L::W3 + k], h2[k], out);
    return out;
}

// gradients of the loss with respect to the pre-activations, given d loss / d output
template <int D, int H1, int H2>
__device__ __forceinline__ void net_backward(const float *__restrict__ p0, const float (&h1)[H1], const float (&h2)[H2],
                                             float dout, float (&dz1)[H1], float (&dz2)[H2]) {
    using L = Net<D, H1, H2>;
    {
        const float *p = not_before(p0, dout);
#pragma unroll
        for (int j = 0; j < H2; j++) dz2[j] = p[L::W3 + j] * dout * (1.0f - h2[j] * h2[j]);
    }
#pragma unroll
    for (int k = 0; k < H1; k++) dz1[k] = 0.0f;
#pragma unroll
    for (int j = 0; j < H2; j++) {   // row j of W2 at a time (contiguous loads), 32 independent sums
        const float *p = not_before(p0, j >= 2 ? dz1[H1 - 1] : dz2[H2 - 1]);   // (dz1[H1 - 1] as of row j - 2)
#pragma unroll
        for (int k = 0; k < H1; k++) dz1[k] = fmaf(p[L::W2 + j * H1 + k], dz2[j], dz1[k]);
    }
#pragma unroll
    for (int k = 0; k < H1; k++) dz1[k] *= 1.0f - h1[k] * h1[k];
}

struct NetAcc {   // one network's gradient sums of a wavefront
    f32x16 w1;    // dW1 (and db1 in column D): row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), column = lane & 31
    f32x4 w2a, w2b;  // dW2: row = 4 * (lane >> 4) + reg, column = (lane & 15) (+ 16 for w2b)
    float b2, w3;    // partial sums over the samples 16 q .. 16 q + 15 (q = lane >> 4) for column lane & 15
    float b3;        // this lane's samples
};

__device__ __forceinline__ void acc_zero(NetAcc &a) {
#pragma unroll
    for (int r = 0; r < 16; r++) a.w1[r] = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) { a.w2a[r] = 0.0f; a.w2b[r] = 0.0f; }
    a.b2 = a.w3 = a.b3 = 0.0f;
}

constexpr int kS33 = 33, kS17 = 17;
constexpr int kBufFloats = kWave * kS33;          // one [64][33] operand buffer
constexpr int kWaveLds = 2 * kBufFloats;          // A operand (dz1, then dz2) | B operand (x, then h1, then h2 and dz3)

// One network's weight-gradient sums over the 64 samples of the tile: the lane's activations go to LDS sample-major and
// come back as MFMA operands (A[i = unit][k = sample], B[k = sample][j = input]).
template <int D, int H1, int H2>
__device__ __forceinline__ void accumulate_tile(NetAcc &acc, float *bufA, float *bufH, const uint32_t lane,
                                                const float (&x)[32], const float (&h1)[H1], const float (&h2)[H2], const float (&dz1)[H1],
                                                const float (&dz2)[H2], const float dout) {
    static_assert(H1 == 32 && H2 == 16, "the MFMA tiles below are the reference's --arch 32,16");
    // ---- layer 1: dW1[i][j] += sum_s dz1[s][i] * x[s][j]   (x[s][D] = 1: db1)
#pragma unroll
    for (int i = 0; i < H1; i++) bufA[lane * kS33 + i] = dz1[i];
#pragma unroll
    for (int j = 0; j < 32; j++) bufH[lane * kS33 + j] = x[j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll 8
    for (int t = 0; t < kWave / 2; t++) {
        const uint32_t s = 2u * t + (lane >> 5);
        acc.w1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bufA[s * kS33 + (lane & 31u)], bufH[s * kS33 + (lane & 31u)], acc.w1, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- layer 2: dW2[i][j] += sum_s dz2[s][i] * h1[s][j]
#pragma unroll
    for (int i = 0; i < H2; i++) bufA[lane * kS17 + i] = dz2[i];
#pragma unroll
    for (int j = 0; j < H1; j++) bufH[lane * kS33 + j] = h1[j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int t = 0; t < kWave / 4; t++) {
        const uint32_t s = 4u * t + (lane >> 4);
        const float a = bufA[s * kS17 + (lane & 15u)];
        acc.w2a = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bufH[s * kS33 + (lane & 15u)], acc.w2a, 0, 0, 0);
        acc.w2b = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bufH[s * kS33 + 16u + (lane & 15u)], acc.w2b, 0, 0, 0);
    }
    // db2[c] += sum_s dz2[s][c]: lane (q, c) takes the samples 16 q .. 16 q + 15
    {
        const uint32_t c = lane & 15u, q = lane >> 4;
        float s2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; e++) s2 += bufA[(16u * q + e) * kS17 + c];
        acc.b2 += s2;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- layer 3: dW3[c] += sum_s dout[s] * h2[s][c];  db3 += dout
#pragma unroll
    for (int j = 0; j < H2; j++) bufH[lane * kS17 + j] = h2[j];
    bufH[kWave * kS17 + lane] = dout;
    __builtin_amdgcn_wave_barrier();
    {
        const uint32_t c = lane & 15u, q = lane >> 4;
        float s3 = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; e++) s3 = fmaf(bufH[kWave * kS17 + 16u * q + e], bufH[(16u * q + e) * kS17 + c], s3);
        acc.w3 += s3;
    }
    acc.b3 += dout;
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// a wavefront's sums of one network -> its slice g[0 .. Net::N) of a gradient vector in LDS (added to what is there)
template <int D, int H1, int H2>
__device__ __forceinline__ void acc_store(const NetAcc &acc, float *g, const uint32_t lane, const bool add) {
    using L = Net<D, H1, H2>;
    auto put = [&](int idx, float v) { g[idx] = add ? g[idx] + v : v; };
    const uint32_t col = lane & 31u;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t row = (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * (lane >> 5);
        if (col < (uint32_t)D) put(L::W1 + (int)row * D + (int)col, acc.w1[r]);
        else if (col == (uint32_t)D) put(L::B1 + (int)row, acc.w1[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t row = 4u * (lane >> 4) + (uint32_t)r, c = lane & 15u;
        put(L::W2 + (int)row * H1 + (int)c, acc.w2a[r]);
        put(L::W2 + (int)row * H1 + 16 + (int)c, acc.w2b[r]);
    }
    // the four q-partials of db2 / dW3, the 64 lane sums of db3
    float b2 = acc.b2, w3 = acc.w3;
    b2 += __shfl_xor(b2, 16, kWave); b2 += __shfl_xor(b2, 32, kWave);
    w3 += __shfl_xor(w3, 16, kWave); w3 += __shfl_xor(w3, 32, kWave);
    const float b3 = wave_sum(acc.b3);
    if (lane < 16u) { put(L::B2 + (int)lane, b2); put(L::W3 + (int)lane, w3); }
    if (lane == 0u) put(L::B3, b3);
}

// Gradient of one minibatch: samples perm[start .. start + count) (perm NULL: start .. start + count) of the rollout.
// partial[block][n_params + 4]: the block's gradient sums, then its sums of {surrogate, squared value error, |ratio - 1| >
// clip, 1}.
template <int D, int H1, int H2>
__global__ __launch_bounds__(kWavesPerBlock *kWave, 2) void ppo_grad_kernel(
    const float *__restrict__ obs, const float *__restrict__ act, const float *__restrict__ logp_old,
    const float *__restrict__ adv, const float *__restrict__ ret, const int64_t *__restrict__ perm, int64_t start,
    int64_t count, const float *__restrict__ params, float clip, float *__restrict__ partial) {
    using L = Net<D, H1, H2>;
    constexpr int kPi = 0, kLogStd = L::N, kVf = L::N + 1, kParams = 2 * L::N + 1;
    static_assert(D < 32, "the observation and the constant 1 of the bias share one 32-column operand");
    static_assert(kParams + 4 <= kWaveLds, "the block's gradient is reduced in a wavefront's LDS buffers");
    __shared__ float lds[kWavesPerBlock * kWaveLds];
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    float *bufA = lds + wv * kWaveLds, *bufH = bufA + kBufFloats;
    const float log_std = params[kLogStd];
    const float inv_std = __expf(-log_std);
    const float inv_n = 1.0f / (float)count;
    NetAcc pi, vf;
    acc_zero(pi);
    acc_zero(vf);
    float g_logstd = 0.0f, s_pg = 0.0f, s_vf = 0.0f, s_clip = 0.0f;
    const int64_t n_tiles = (count + kWave - 1) / kWave;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wv; tile < n_tiles; tile += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t k = tile * kWave + lane;
        const bool valid = k < count;
        const int64_t idx = valid ? (perm ? perm[start + k] : start + k) : 0;
        float x[32];
        {
            const float2 *row = reinterpret_cast<const float2 *>(obs + idx * D);   // D even: 8-byte aligned rows
            if constexpr (D % 2 == 0) {
#pragma unroll
                for (int j = 0; j < D / 2; j++) { const float2 v = row[j]; x[2 * j] = v.x; x[2 * j + 1] = v.y; }
            } else {
#pragma unroll
                for (int j = 0; j < D; j++) x[j] = obs[idx * D + j];
            }
        }
#pragma unroll
        for (int j = D; j < 32; j++) x[j] = j == D ? 1.0f : 0.0f;
        if (!valid) {
#pragma unroll
            for (int j = 0; j < 32; j++) x[j] = 0.0f;   // (and every d loss / d output below is 0)
        }
        float h1[H1], h2[H2], dz1[H1], dz2[H2];
        // ---- policy network: log-probability of the taken action, clipped surrogate
        {
            const float mu = net_forward<D, H1, H2>(params + kPi, x, h1, h2);
            const float a = valid ? act[idx] : 0.0f, lp_old = valid ? logp_old[idx] : 0.0f, ad = valid ? adv[idx] : 0.0f;
            const float z = (a - mu) * inv_std;
            const float lp = -0.5f * z * z - log_std - 0.918938533204672742f;
            const float ratio = __expf(lp - lp_old);
            const float lo = 1.0f - clip, hi = 1.0f + clip;
            const float rc = fminf(fmaxf(ratio, lo), hi);
            const float surr1 = ratio * ad, surr2 = rc * ad;
            const bool through = surr1 <= surr2;   // min picks the unclipped term (inside the range both are the same)
            // loss = -mean(min(surr1, surr2)): d loss / d logp = -adv * ratio / n where the unclipped term is the minimum
            const float dlp = (valid && through) ? -ad * ratio * inv_n : 0.0f;
            const float dmu = dlp * z * inv_std;
            g_logstd += dlp * (z * z - 1.0f);
            if (valid) {
                s_pg += fminf(surr1, surr2);
                s_clip += (ratio < lo || ratio > hi) ? 1.0f : 0.0f;
            }
            net_backward<D, H1, H2>(params + kPi, h1, h2, dmu, dz1, dz2);
            accumulate_tile<D, H1, H2>(pi, bufA, bufH, lane, x, h1, h2, dz1, dz2, dmu);
        }
        // ---- value network: 0.5 * mean((v - ret)^2)
        {
            const float v = net_forward<D, H1, H2>(params + kVf, x, h1, h2);
            const float err = valid ? v - ret[idx] : 0.0f;
            s_vf += err * err;
            const float dv = err * inv_n;
            net_backward<D, H1, H2>(params + kVf, h1, h2, dv, dz1, dz2);
            accumulate_tile<D, H1, H2>(vf, bufA, bufH, lane, x, h1, h2, dz1, dz2, dv);
        }
    }
    // ---- the block's partial gradient: every wavefront's sums -> LDS (wavefront 0's buffers), one after the other
    __syncthreads();
    float *g = lds;
    for (uint32_t w = 0; w < (uint32_t)kWavesPerBlock; w++) {
        if (wv == w) {
            acc_store<D, H1, H2>(pi, g + kPi, lane, w != 0);
            acc_store<D, H1, H2>(vf, g + kVf, lane, w != 0);
            const float gl = wave_sum(g_logstd), a = wave_sum(s_pg), b = wave_sum(s_vf), c = wave_sum(s_clip);
            if (lane == 0) {
                g[kLogStd] = (w ? g[kLogStd] : 0.0f) + gl;
                g[kParams + 0] = (w ? g[kParams + 0] : 0.0f) + a;
                g[kParams + 1] = (w ? g[kParams + 1] : 0.0f) + b;
                g[kParams + 2] = (w ? g[kParams + 2] : 0.0f) + c;
                g[kParams + 3] = 0.0f;
            }
        }
        __syncthreads();
    }
    float *out = partial + (int64_t)blockIdx.x * (kParams + 4);
    for (int k = threadIdx.x; k < kParams + 4; k += blockDim.x) out[k] = g[k];
}

// 16 parameters per workgroup: thread (r, c) sums parameter c's partial gradients of the blocks r, r + 16, ..., the 16 row sums
// are added in order (deterministic: no atomics anywhere); then the entropy term (entropy of the diagonal Gaussian = const +
// log_std: d/d log_std of -ent_coef * entropy is -ent_coef) and Adam as torch.optim.Adam computes it.
__global__ __launch_bounds__(256) void ppo_adam_kernel(const float *__restrict__ partial, int n_blocks, int n_params,
                                                       int logstd_index, float ent_coef, float *__restrict__ params,
                                                       float *__restrict__ m, float *__restrict__ v, float lr, float beta1,
                                                       float beta2, float eps, float bias1, float bias2_sqrt, float inv_count,
                                                       float *__restrict__ grad_out, float *__restrict__ stats_out) {
    __shared__ float rows[16][17];
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int p = blockIdx.x * 16 + c;
    const int stride = n_params + 4;
    float g = 0.0f;
    if (p < stride)
        for (int b = r; b < n_blocks; b += 16) g += partial[(int64_t)b * stride + p];
    rows[r][c] = g;
    __syncthreads();
    if (r != 0 || p >= stride) return;
    g = rows[0][c];
#pragma unroll
    for (int k = 1; k < 16; k++) g += rows[k][c];
    if (p >= n_params) {   // {-policy loss, 2 * value loss, clipped fraction, -} as means over the minibatch
        if (stats_out) stats_out[p - n_params] = g * inv_count;
        return;
    }
    if (p == logstd_index) g -= ent_coef;
    if (grad_out) grad_out[p] = g;
    if (lr == 0.0f) return;   // gradient only
    const float mm = beta1 * m[p] + (1.0f - beta1) * g;
    const float vv = beta2 * v[p] + (1.0f - beta2) * g * g;
    m[p] = mm;
    v[p] = vv;
    const float denom = sqrtf(vv) / bias2_sqrt + eps;
    params[p] -= (lr / bias1) * (mm / denom);
}

// Generalised advantage estimation over [T][N] rollout rows (thread = env, backwards in time): dones[t] marks that the
// env was reset after step t.
__global__ void gae_kernel(const float *__restrict__ rew, const float *__restrict__ val, const uint8_t *__restrict__ done,
                           const float *__restrict__ last_val, int T, int64_t n, float gamma, float lam,
                           float *__restrict__ adv, float *__restrict__ ret) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float next_v = last_val[i], run = 0.0f;
    for (int t = T - 1; t >= 0; t--) {
        const int64_t k = (int64_t)t * n + i;
        const float alive = done[k] ? 0.0f : 1.0f;
        const float v = val[k];
        const float delta = rew[k] + gamma * next_v * alive - v;
        run = delta + gamma * lam * alive * run;
        adv[k] = run;
        ret[k] = run + v;
        next_v = v;
    }
}

}  // namespace

extern "C" int pcc_ppo_scratch_floats(int obs_dim, int h1, int h2) {
    const int n_net = h1 * obs_dim + h1 + h2 * h1 + h2 + h2 + 1;
    return kMaxBlocks * (2 * n_net + 1 + 4);
}

extern "C" int pcc_ppo_minibatch_step(const float *obs, const float *act, const float *logp_old, const float *adv,
                                      const float *ret, const int64_t *perm, int64_t start, int64_t count, int obs_dim,
                                      int h1, int h2, float *params, float *adam_m, float *adam_v, int adam_step, float lr,
                                      float beta1, float beta2, float eps, float clip, float ent_coef, float *scratch,
                                      float *grad_out, float *stats_out, void *stream) {
    if (!obs || !act || !logp_old || !adv || !ret || !params || !scratch || count < 1 || start < 0) return -1;
    if (lr != 0.0f && (!adam_m || !adam_v || adam_step < 1)) return -1;
    if (h1 != 32 || h2 != 16) return -2;
    const int n_net = h1 * obs_dim + h1 + h2 * h1 + h2 + h2 + 1;
    const int n_params = 2 * n_net + 1;
    const int64_t tiles = (count + kWave - 1) / kWave;
    int64_t blocks = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)blocks), block(kWavesPerBlock * kWave);
    switch (obs_dim) {
#define PCC_PPO_CASE(DD)                                                                                                   \
    case DD:                                                                                                               \
        hipLaunchKernelGGL((ppo_grad_kernel<DD, 32, 16>), grid, block, 0, st, obs, act, logp_old, adv, ret, perm, start, count, \
                           params, clip, scratch);                                                                        \
        break;
        PCC_PPO_CASE(30)   // history 10 x 3 features: the reference's default observation (ns:382-388)
        PCC_PPO_CASE(3)
        PCC_PPO_CASE(6)
        PCC_PPO_CASE(12)
#undef PCC_PPO_CASE
        default: return -2;   // the caller falls back to its framework path
    }
    if (hipGetLastError() != hipSuccess) return -3;
    const float bias1 = lr != 0.0f ? 1.0f - powf(beta1, (float)adam_step) : 1.0f;
    const float bias2 = lr != 0.0f ? sqrtf(1.0f - powf(beta2, (float)adam_step)) : 1.0f;
    hipLaunchKernelGGL(ppo_adam_kernel, dim3((unsigned)((n_params + 4 + 15) / 16)), dim3(256), 0, st, scratch, (int)blocks,
                       n_params, n_net, ent_coef, params, adam_m, adam_v, lr, beta1, beta2, eps, bias1, bias2,
                       1.0f / (float)count, grad_out, stats_out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int pcc_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_value, int T,
                       int64_t n_envs, float gamma, float lam, float *adv_out, float *ret_out, void *stream) {
    if (!rewards || !values || !dones || !last_value || !adv_out || !ret_out || T < 1 || n_envs < 1) return -1;
    hipLaunchKernelGGL(gae_kernel, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rewards, values, dones, last_value, T, n_envs, gamma, lam, adv_out, ret_out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
