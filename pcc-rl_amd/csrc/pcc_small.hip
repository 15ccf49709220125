// pcc_small.hip -- step_small_kernel (both halves of a small batch's step in one launch), reset_init_kernel (new links and
// fresh state at a reset) and the pool bookkeeping kernel of pcc_set_ring_pools.
#include "pcc_send_item.h"
#include "pcc_retire_env.h"
#include "pcc_kernels.h"

#ifndef PCC_SMALL_OCC
#define PCC_SMALL_OCC 2  // wavefronts per SIMD the register budget is cut for: the kernel is a chain of dependent round trips of
                         // one workgroup per 64 envs (at most 128 workgroups on 256 compute units), not an occupancy problem
#endif

namespace {

// envs of a workgroup of the small-batch kernel: 32 = one round of the four wavefronts at 8 lanes per env in the retire part
// (64 envs per workgroup were two rounds, i.e. the retire half's chain of dependent round trips twice: 19.0 us per step of
// config 2 against 14-15; a batch below 8 192 envs is at most 256 workgroups either way: one per compute unit)
constexpr int kSmallEnvs = 32;

// ======================================================================================
// step_small_kernel: both halves of a step in ONE launch, for batches too small for work lists (pcc_step of fewer than
// list_min_envs envs).  A workgroup owns kSmallEnvs envs: its first wavefront sends them, a lane each (send_light_item, the tail by the
// wave path), then the four wavefronts retire them, 8 lanes per env.  No cross-workgroup dependency: an env's retire
// half needs only its own send half.  At 4 096 envs of two packets a step is launch overhead and dependent loads, and
// one launch instead of two is a third of it (config 2: 34 -> about 24 us per step).
// ======================================================================================
// n_steps > 1 (pcc_step_many): the workgroup runs that many steps of its 64 envs back to back inside the launch -- step t
// reads actions + t * act_stride bytes and writes the t-th row of every output -- with a workgroup barrier between the
// halves of a step and between steps: nothing of an env is ever touched by another workgroup, so no launch boundary is
// needed, and at 4 096 envs of two packets a launch boundary (~8 us) is a third of a step.
template <int NS, bool TRACE>
__global__ __launch_bounds__(4 * kWave, PCC_SMALL_OCC) void step_small_kernel(Dev D, const void *actions, int actions_f64, float *obs_out,
                                                                           float *reward_out, uint8_t *done_out, double *steps_out,
                                                                           int n_steps, int64_t act_stride) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int64_t base = (int64_t)blockIdx.x * kSmallEnvs;
    __shared__ EnvSlot<NS> s_slots[kSlots];
    const int64_t row = D.n * NS;
#pragma unroll 1
    for (int t = 0; t < n_steps; t++) {
        const void *act_t = static_cast<const char *>(actions) + (int64_t)t * act_stride;
        if (wv == 0) {
            const int64_t i = base + lane;
            const bool has = lane < (uint32_t)kSmallEnvs && i < D.n;
            uint32_t pk = 0;
            const uint64_t left = send_light_item<NS, TRACE>(D, lane, has ? i : 0, has, blockIdx.x, 0, 0, act_t, actions_f64, pk);
            if (left) {  // the last lanes of the rounds go on by the wave path, from the state the item stored (pcc_send_item.h)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                (void)send_wave_item<NS, TRACE, 1>(D, lane, has ? i : 0, ((left >> lane) & 1ull) != 0ull, false, 0xFFFFFFFFu, 0, 0, act_t, actions_f64, s_slots);
            }
        }
        __syncthreads();  // the records and the state the first wavefront wrote are read by all four (same CU: workgroup scope)
        float *obs_t = obs_out ? obs_out + (int64_t)t * row * D.HF : nullptr;
        float *rew_t = reward_out ? reward_out + (int64_t)t * row : nullptr;
        uint8_t *done_t = done_out ? done_out + (int64_t)t * D.n : nullptr;
        double *steps_t = steps_out ? steps_out + (int64_t)t * row * PCC_STEP_COLS : nullptr;
        {   // 4 wavefronts x 8 envs at 8 lanes each: the workgroup's 32 envs in one round
            const int64_t i = base + (int64_t)(wv * 8u + lane / 8u);
            Group g;
            g.lane = lane & 7u;
            g.shift = lane & ~7u;
            if (i < base + kSmallEnvs && i < D.n)
                (void)retire_env<NS, false, 8>(D, i, g, 0, 0, 0, 0, obs_t, rew_t, done_t, steps_t, nullptr, 0);
        }
        if (t + 1 < n_steps) __syncthreads();  // the next step's send half reads what this retire half wrote
    }
}

// ======================================================================================
// reset_init_kernel: ns:454-477 -- parameters, fresh link/sender/history state.  The two warm-up
// MIs (ns:478-479) are run by send_kernel / retire_kernel in warm mode on the marked envs.
// ======================================================================================
// all_envs: the host knows that EVERY env is reset by this launch (a full reset, or the episode boundary of a batch in
// lockstep).  Then nobody keeps a pool slot and the free stacks are simply rebuilt in order (slot 0 on top) instead of
// being pushed slot by slot in whatever order the atomics land: a batch whose pool rings sit in the order they were
// handed out runs its send half 15-35 % faster than one whose rings are scattered over the pools (the second episode of
// a handle took 0.163 ms per send launch against 0.118 for the first; profiles/r03_experiments.json).
template <int NS>
__global__ __launch_bounds__(kWave) void reset_init_kernel(Dev D, const uint8_t *mask, int use_done, int gate, int all_envs,
                                                           float *obs_out) {
    if (gate && __hip_atomic_load(D.any_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != D.step_seq) return;
    const int64_t i = (int64_t)blockIdx.x * kWave + threadIdx.x;
    if (all_envs) {
        for (int c = 1; c < D.n_tiers; c++) {
            const int64_t share = (int64_t)D.pool_share[c], slots = share * D.parts;   // (every partition's stack: its lowest slot on top)
            for (int64_t j = i; j < slots; j += (int64_t)gridDim.x * kWave)
                D.pool_free[(size_t)c * D.pool_free_stride + j] = (uint32_t)(j / share * share + (share - 1 - j % share));
            if (i < (int64_t)D.parts) D.tier_top[(size_t)(c * kParts + i) * kTopStride] = (int32_t)share;
        }
    }
    if (i >= D.n) return;
    // use_done 1: the envs that finished their episode; 2: the envs a retire launch marked for a restart
    const bool sel = use_done == 2 ? D.env[i].resetting == 2 : (!mask || mask[i]) && (!use_done || D.env[i].done);
    D.env[i].resetting = sel ? 1 : 0;
    if (sel && D.shadows && !all_envs && shadow_list(&D.env[D.n + i])) {
        // a reset that is not the env's own episode end overtakes its shadow (prepared for the episode index this reset now
        // takes): have the shadow prepared again, for the episode after this one (listed once, whoever else lists it in this step)
        const uint32_t row = D.step_seq & 3u;
        D.refill_list[(size_t)row * (size_t)D.n + atomicAdd(&D.refill_count[row * kCntStride], 1u)] = (uint32_t)i;
    }
    if (sel) {
        release_ring_slots<NS>(D, i, !all_envs);
        reset_env<NS>(D, i, obs_out);
    }
}

// pcc_set_ring_pools: every sender back in its own tier-0 rings, holding no pool slot (the pools are being replaced)
__global__ void forget_ring_slots_kernel(Dev D) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D.n * D.ns) return;
    const int64_t s = j / D.n, i = j % D.n;
    const int64_t k = sidx(D, (int)s, i);   // (sender blocks are [S][2 N]: every env's block is followed, N blocks on, by its shadow's)
    for (int c = 0; c < kMaxTiers; c++) D.snd[k].ring_held[c] = 0;
    D.snd[k].ring_tier = 0;
    D.snd[k].ring_base = D.tier_base[0] + (size_t)(i * D.ns + s) * tier_slot_bytes(D, 0);
    // (the shadow's blocks hold no pool slot -- a refill never touches a pool -- and keep their private rings)
}

}  // namespace

namespace pcc {

void launch_step_small(const Dev &d, bool trace, hipStream_t st, const void *actions, int actions_f64, float *obs_out,
                       float *reward_out, uint8_t *done_out, double *steps_out, int n_steps, int64_t act_stride) {
    const dim3 grid((unsigned)((d.n + kSmallEnvs - 1) / kSmallEnvs)), block(4 * kWave);
#define PCC_S(NS_, TR_) \
    hipLaunchKernelGGL((step_small_kernel<NS_, TR_>), grid, block, 0, st, d, actions, actions_f64, obs_out, reward_out, done_out, steps_out, n_steps, act_stride)
    if (d.ns == 1) { if (trace) PCC_S(1, true); else PCC_S(1, false); }
    else { if (trace) PCC_S(2, true); else PCC_S(2, false); }
#undef PCC_S
}

void launch_reset_init(const Dev &d, hipStream_t st, const uint8_t *mask, int use_done, int gate, int all_envs, float *obs_out) {
    const dim3 grid((unsigned)((d.n + kWave - 1) / kWave));
    if (d.ns == 1) hipLaunchKernelGGL(reset_init_kernel<1>, grid, dim3(kWave), 0, st, d, mask, use_done, gate, all_envs, obs_out);
    else hipLaunchKernelGGL(reset_init_kernel<2>, grid, dim3(kWave), 0, st, d, mask, use_done, gate, all_envs, obs_out);
}

void launch_forget_ring_slots(const Dev &d, hipStream_t st) {
    const int64_t senders = d.n * d.ns;
    hipLaunchKernelGGL(forget_ring_slots_kernel, dim3((unsigned)((senders + 255) / 256)), dim3(256), 0, st, d);
}

}  // namespace pcc
