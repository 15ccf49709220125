// pcc_ppo.hip -- the update half of the on-device PPO caller: one optimiser step of
// PPO1's objective (clipped surrogate + 0.5 * value error - ent_coef * entropy; stable-baselines' defaults as
// src/gym/stable_solve.py:52 uses them) on one minibatch of the rollout, for the reference's policy shape
// (src/gym/stable_solve.py:39-45: pi and vf MLPs, obs -> 32 -> 16 -> 1, tanh, state-independent log-std), as TWO
// launches instead of the ~150 small framework launches of forward + autograd + Adam:
//
//   ppo_grad_mfma_kernel  every contraction of forward and backward on the matrix cores, fp32 (v_mfma_f32_32x32x2_f32 /
//                     v_mfma_f32_16x16x4_f32: exact fp32): a layer is a small GEMM per tile of 64 samples, the weights are
//                     B operands held in registers, the activations A operands read sample-major from LDS, tanh and
//                     its derivative elementwise in the MFMA's C layout; the weight gradients dW = dZ^T X are contractions
//                     over the samples whose accumulators stay in registers over all tiles of a wavefront.  Every
//                     workgroup writes ONE partial gradient.  (The first version of this kernel evaluated the networks
//                     lane = sample with plain FMAs and only the weight gradients on MFMAs: 392 us per million samples
//                     with the weights as scalar operands -- every output unit waits for its scalar loads -- 579 with
//                     the weights broadcast from LDS; this one 328.  profiles/r03_experiments.json.)
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
constexpr int kMaxBlocks = 512;   // 2 wavefronts each: one per SIMD of the chip, which is what ~480 registers allow

template <int D, int H1, int H2>
struct Net {  // offsets inside one network's block of the parameter vector (include/pcc_policy.h)
    static constexpr int W1 = 0, B1 = H1 * D, W2 = B1 + H1, B2 = W2 + H2 * H1, W3 = B2 + H2, B3 = W3 + H2, N = B3 + 1;
};

// tanh(x) = 1 - 2 / (exp(2x) + 1): absolute error ~1e-7 (one rounding of the quotient against 1), saturates cleanly
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

constexpr int kS33 = 33, kS17 = 17;               // row strides of the sample-major LDS buffers: conflict-free both ways
constexpr int kBufFloats = kWave * kS33;          // one [64][33] operand buffer

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// ======================================================================================
// ppo_grad_mfma_kernel: a layer is a small GEMM per tile of 64 samples: the weights are the B
// operands and stay in registers for the whole kernel (forward: W1^T, W2^T; backward: W2), the activations are the A
// operands, read sample-major from LDS, the results land in the MFMA's C layout (lane = output unit, registers = samples),
// where tanh, its derivative and the bias gradients are elementwise / per-lane sums; each layer's output goes back to LDS
// sample-major for the next contraction.  Per network and tile 144 MFMAs (fp32: v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32), ~7 k cycles of the matrix pipe.
//   layouts (cdna_hip_programming.md): 32x32x2: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31], C: col = l & 31,
//   row = (r & 3) + 8 (r >> 2) + 4 (l >> 5); 16x16x4: A[l & 15][k = l >> 4], B[k = l >> 4][l & 15], C: col = l & 15,
//   row = 4 (l >> 4) + r.
// ======================================================================================
constexpr int kMfmaWaveLds = 3 * kBufFloats + 4 * kWave;   // x | h1 | dz (dz2, then dz1) | act, logp_old, adv, ret

__device__ __forceinline__ float row16_sum(float x) {   // the sum over the 16 lanes of a DPP row, in every lane of it
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));    // quad_perm:[1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, false));    // quad_perm:[2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, false));   // row_half_mirror
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, false));   // row_mirror
    return x;
}

struct MfmaWeights {   // one network's weights as MFMA B operands (and per-lane constants of the C layout)
    float w1b[16];     // layer 1 forward:  W1[n = l & 31][k = 2 t + (l >> 5)]
    float w2b[8];      // layer 2 forward:  W2[n = l & 15][k = 4 t + (l >> 4)]
    float w2c[8];      // layer 2 backward: W2[j = 2 t + (l >> 5)][n = l & 31]
    float b1c, b2c, w3c, b3;
};
struct MfmaAcc {       // one network's gradient sums of a wavefront
    f32x16 w1;         // dW1: row = unit (C layout of 32x32), col = feature
    f32x4 w2a, w2b;    // dW2: row = unit j (C layout of 16x16), col = k (+ 16)
    float b1, b2, w3, b3;   // per-lane partial sums: b1 for unit l & 31, b2 / w3 for unit l & 15, b3 (lanes with l & 15 == 0 count)
};

template <int D, int H1, int H2>
__device__ __forceinline__ void mfma_load_weights(MfmaWeights &w, const float *__restrict__ p, uint32_t lane) {
    using L = Net<D, H1, H2>;
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const uint32_t k = 2u * t + (lane >> 5);
        w.w1b[t] = k < (uint32_t)D ? p[L::W1 + (lane & 31u) * D + k] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        w.w2b[t] = p[L::W2 + (lane & 15u) * H1 + 4u * t + (lane >> 4)];
        w.w2c[t] = p[L::W2 + (2u * t + (lane >> 5)) * H1 + (lane & 31u)];
    }
    w.b1c = p[L::B1 + (lane & 31u)];
    w.b2c = p[L::B2 + (lane & 15u)];
    w.w3c = p[L::W3 + (lane & 15u)];
    w.b3 = p[L::B3];
}

// forward of one network over the tile: on return h1 (C layout of two 32x32 tiles) and h2 (C layout of four 16x16 tiles) hold
// the activations, H1s holds h1 sample-major, out[T4][r] the network's output for sample 16 T4 + 4 (l >> 4) + r
template <int D, int H1, int H2>
__device__ __forceinline__ void mfma_forward(const MfmaWeights &w, const float *Xs, float *H1s, uint32_t lane, float (&h1)[2][16],
                                             float (&h2)[4][4], float (&out)[4][4]) {
#pragma unroll
    for (int T = 0; T < 2; T++) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = w.b1c;
#pragma unroll
        for (int t = 0; t < 16; t++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[(32u * T + (lane & 31u)) * kS33 + 2u * t + (lane >> 5)], w.w1b[t], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            h1[T][r] = tanh_fast(acc[r]);
            const uint32_t m = 32u * T + (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * (lane >> 5);
            H1s[m * kS33 + (lane & 31u)] = h1[T][r];
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int T4 = 0; T4 < 4; T4++) {
        f32x4 acc;
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = w.b2c;
#pragma unroll
        for (int t = 0; t < 8; t++)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(H1s[(16u * T4 + (lane & 15u)) * kS33 + 4u * t + (lane >> 4)], w.w2b[t], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            h2[T4][r] = tanh_fast(acc[r]);
            out[T4][r] = row16_sum(h2[T4][r] * w.w3c) + w.b3;   // the 16 lanes of a DPP row = the 16 units of one sample
        }
    }
}

// backward of one network over the tile, given d loss / d output per sample (dout[T4][r], like `out` above)
template <int D, int H1, int H2>
__device__ __forceinline__ void mfma_backward(const MfmaWeights &w, MfmaAcc &g, const float *Xs, const float *H1s, float *Zs,
                                              uint32_t lane, const float (&h1)[2][16], const float (&h2)[4][4],
                                              const float (&dout)[4][4]) {
    // ---- layer 3 and the pre-activation gradient of layer 2 (C layout of the 16x16 tiles), sample-major to LDS
#pragma unroll
    for (int T4 = 0; T4 < 4; T4++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float d = dout[T4][r];
            const float dz2 = d * w.w3c * (1.0f - h2[T4][r] * h2[T4][r]);
            g.w3 = fmaf(d, h2[T4][r], g.w3);
            g.b2 += dz2;
            if ((lane & 15u) == 0u) g.b3 += d;
            const uint32_t m = 16u * T4 + 4u * (lane >> 4) + (uint32_t)r;
            Zs[m * kS17 + (lane & 15u)] = dz2;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- dW2[j][k] += sum_s dz2[s][j] h1[s][k]
#pragma unroll 4
    for (int t = 0; t < kWave / 4; t++) {
        const uint32_t s = 4u * t + (lane >> 4);
        const float a = Zs[s * kS17 + (lane & 15u)];
        g.w2a = __builtin_amdgcn_mfma_f32_16x16x4f32(a, H1s[s * kS33 + (lane & 15u)], g.w2a, 0, 0, 0);
        g.w2b = __builtin_amdgcn_mfma_f32_16x16x4f32(a, H1s[s * kS33 + 16u + (lane & 15u)], g.w2b, 0, 0, 0);
    }
    // ---- dh1[s][k] = sum_j dz2[s][j] W2[j][k]; dz1 = dh1 (1 - h1^2) (C layout of the 32x32 tiles)
    float dz1[2][16];
#pragma unroll
    for (int T = 0; T < 2; T++) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; t++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Zs[(32u * T + (lane & 31u)) * kS17 + 2u * t + (lane >> 5)], w.w2c[t], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            dz1[T][r] = acc[r] * (1.0f - h1[T][r] * h1[T][r]);
            g.b1 += dz1[T][r];
        }
    }
    __builtin_amdgcn_wave_barrier();   // (every read of dz2 is issued: dz1 takes the buffer over)
#pragma unroll
    for (int T = 0; T < 2; T++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t m = 32u * T + (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * (lane >> 5);
            Zs[m * kS33 + (lane & 31u)] = dz1[T][r];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- dW1[i][k] += sum_s dz1[s][i] x[s][k]
#pragma unroll 8
    for (int t = 0; t < kWave / 2; t++) {
        const uint32_t s = 2u * t + (lane >> 5);
        g.w1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Zs[s * kS33 + (lane & 31u)], Xs[s * kS33 + (lane & 31u)], g.w1, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
}

template <int D, int H1, int H2>
__device__ __forceinline__ void mfma_acc_store(const MfmaAcc &acc, float *g, const uint32_t lane, const bool add) {
    using L = Net<D, H1, H2>;
    auto put = [&](int idx, float v) { g[idx] = add ? g[idx] + v : v; };
    const uint32_t col = lane & 31u;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t row = (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * (lane >> 5);
        if (col < (uint32_t)D) put(L::W1 + (int)row * D + (int)col, acc.w1[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t row = 4u * (lane >> 4) + (uint32_t)r, c = lane & 15u;
        put(L::W2 + (int)row * H1 + (int)c, acc.w2a[r]);
        put(L::W2 + (int)row * H1 + 16 + (int)c, acc.w2b[r]);
    }
    float b1 = acc.b1, b2 = acc.b2, w3 = acc.w3;
    b1 += __shfl_xor(b1, 32, kWave);
    b2 += __shfl_xor(b2, 16, kWave); b2 += __shfl_xor(b2, 32, kWave);
    w3 += __shfl_xor(w3, 16, kWave); w3 += __shfl_xor(w3, 32, kWave);
    const float b3 = wave_sum(acc.b3);
    if (lane < 32u) put(L::B1 + (int)lane, b1);
    if (lane < 16u) { put(L::B2 + (int)lane, b2); put(L::W3 + (int)lane, w3); }
    if (lane == 0u) put(L::B3, b3);
}

template <int D, int H1, int H2>
__global__ __launch_bounds__(kWavesPerBlock *kWave, 1) void ppo_grad_mfma_kernel(
    const float *__restrict__ obs, const float *__restrict__ act, const float *__restrict__ logp_old,
    const float *__restrict__ adv, const float *__restrict__ ret, const int64_t *__restrict__ perm, int64_t start,
    int64_t count, const float *__restrict__ params, float clip, float *__restrict__ partial) {
    using L = Net<D, H1, H2>;
    static_assert(H1 == 32 && H2 == 16 && D <= 32, "the MFMA tiles are the reference's --arch 32,16 on at most 32 features");
    constexpr int kPi = 0, kLogStd = L::N, kVf = L::N + 1, kParams = 2 * L::N + 1;
    static_assert(kParams + 4 <= kMfmaWaveLds, "the block's gradient is reduced in a wavefront's LDS buffers");
    __shared__ float lds[kWavesPerBlock * kMfmaWaveLds];
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    float *Xs = lds + wv * kMfmaWaveLds, *H1s = Xs + kBufFloats, *Zs = H1s + kBufFloats, *Sc = Zs + kBufFloats;
    const float log_std = params[kLogStd];
    const float inv_std = __expf(-log_std);
    const float inv_n = 1.0f / (float)count;
    // both networks' weights stay in registers (~480 with the activations of a tile: one wavefront per SIMD.  One network
    // at a time -- two passes over the samples, two wavefronts per SIMD at 256 registers -- was measured: 506 us against
    // 328 per million samples, spills and a second gather of the observation rows)
    MfmaWeights wpi, wvf;
    mfma_load_weights<D, H1, H2>(wpi, params + kPi, lane);
    mfma_load_weights<D, H1, H2>(wvf, params + kVf, lane);
    MfmaAcc gpi, gvf;
#pragma unroll
    for (int r = 0; r < 16; r++) { gpi.w1[r] = 0.0f; gvf.w1[r] = 0.0f; }
#pragma unroll
    for (int r = 0; r < 4; r++) { gpi.w2a[r] = gpi.w2b[r] = 0.0f; gvf.w2a[r] = gvf.w2b[r] = 0.0f; }
    gpi.b1 = gpi.b2 = gpi.w3 = gpi.b3 = 0.0f;
    gvf.b1 = gvf.b2 = gvf.w3 = gvf.b3 = 0.0f;
    float g_logstd = 0.0f, s_pg = 0.0f, s_vf = 0.0f, s_clip = 0.0f;
    const int64_t n_tiles = (count + kWave - 1) / kWave;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wv; tile < n_tiles; tile += (int64_t)gridDim.x * kWavesPerBlock) {
        // ---- the tile's samples, sample-major in LDS (lane = sample for the loads only)
        {
            const int64_t k = tile * kWave + lane;
            const bool valid = k < count;
            const int64_t idx = valid ? (perm ? perm[start + k] : start + k) : 0;
            if constexpr (D % 2 == 0) {
                const float2 *row = reinterpret_cast<const float2 *>(obs + idx * D);   // D even: 8-byte aligned rows
#pragma unroll
                for (int j = 0; j < D / 2; j++) {
                    const float2 v = row[j];
                    Xs[lane * kS33 + 2 * j] = valid ? v.x : 0.0f;
                    Xs[lane * kS33 + 2 * j + 1] = valid ? v.y : 0.0f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < D; j++) Xs[lane * kS33 + j] = valid ? obs[idx * D + j] : 0.0f;
            }
#pragma unroll
            for (int j = D; j < 32; j++) Xs[lane * kS33 + j] = 0.0f;
            Sc[lane] = valid ? act[idx] : 0.0f;
            Sc[kWave + lane] = valid ? logp_old[idx] : 0.0f;
            Sc[2 * kWave + lane] = valid ? adv[idx] : 0.0f;
            Sc[3 * kWave + lane] = valid ? ret[idx] : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
        const int64_t left = count - tile * kWave;   // samples of this tile that exist
        float h1[2][16], h2[4][4], out[4][4], dout[4][4];
        // ---- policy network: log-probability of the taken action, clipped surrogate
        mfma_forward<D, H1, H2>(wpi, Xs, H1s, lane, h1, h2, out);
#pragma unroll
        for (int T4 = 0; T4 < 4; T4++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t m = 16u * T4 + 4u * (lane >> 4) + (uint32_t)r;
                const bool valid = (int64_t)m < left;
                const float a = Sc[m], lp_old = Sc[kWave + m], ad = Sc[2 * kWave + m];
                const float z = (a - out[T4][r]) * inv_std;
                const float lp = -0.5f * z * z - log_std - 0.918938533204672742f;
                const float ratio = __expf(lp - lp_old);
                const float lo = 1.0f - clip, hi = 1.0f + clip;
                const float rc = fminf(fmaxf(ratio, lo), hi);
                const float surr1 = ratio * ad, surr2 = rc * ad;
                const bool through = surr1 <= surr2;   // min picks the unclipped term (inside the range both are the same)
                const float dlp = (valid && through) ? -ad * ratio * inv_n : 0.0f;
                dout[T4][r] = dlp * z * inv_std;
                if (valid && (lane & 15u) == 0u) {   // (the 16 lanes of a row hold the same sample: one of them counts)
                    g_logstd += dlp * (z * z - 1.0f);
                    s_pg += fminf(surr1, surr2);
                    s_clip += (ratio < lo || ratio > hi) ? 1.0f : 0.0f;
                }
            }
        }
        mfma_backward<D, H1, H2>(wpi, gpi, Xs, H1s, Zs, lane, h1, h2, dout);
        // ---- value network: 0.5 * mean((v - ret)^2)
        mfma_forward<D, H1, H2>(wvf, Xs, H1s, lane, h1, h2, out);
#pragma unroll
        for (int T4 = 0; T4 < 4; T4++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t m = 16u * T4 + 4u * (lane >> 4) + (uint32_t)r;
                const bool valid = (int64_t)m < left;
                const float err = valid ? out[T4][r] - Sc[3 * kWave + m] : 0.0f;
                dout[T4][r] = err * inv_n;
                if ((lane & 15u) == 0u) s_vf += err * err;
            }
        }
        mfma_backward<D, H1, H2>(wvf, gvf, Xs, H1s, Zs, lane, h1, h2, dout);
    }
    // ---- the block's partial gradient: every wavefront's sums -> LDS (wavefront 0's buffers), one after the other
    __syncthreads();
    float *g = lds;
    for (uint32_t w = 0; w < (uint32_t)kWavesPerBlock; w++) {
        if (wv == w) {
            mfma_acc_store<D, H1, H2>(gpi, g + kPi, lane, w != 0);
            mfma_acc_store<D, H1, H2>(gvf, g + kVf, lane, w != 0);
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
    float *out_p = partial + (int64_t)blockIdx.x * (kParams + 4);
    for (int k = threadIdx.x; k < kParams + 4; k += blockDim.x) out_p[k] = g[k];
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
        hipLaunchKernelGGL((ppo_grad_mfma_kernel<DD, 32, 16>), grid, block, 0, st, obs, act, logp_old, adv, ret, perm, start, \
                           count, params, clip, scratch);                                                                 \
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
