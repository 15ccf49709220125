// pcc_send_restart.hip -- send_restart_kernel: the restart items of the send half.  An env that finished its episode while
// the batch is out of lockstep (the host cannot know which step ends whose episode) was marked by the retire launch
// (resetting = 2) and filed in the restart list; here it gets its new links and fresh state (ns:469-477), the two
// unrecorded warm-up intervals (ns:478-479: send + retire each, by one wavefront) and then the send half of its first
// interval, like every other env of the step.  One env per item; runs beside the light and the wave kernel on a third
// stream.  Cut for two workgroups per compute unit (256 registers): the warm-up retire is inlined, and there are only as
// many items as envs finished in the last step.
#include "pcc_send_item.h"
#include "pcc_retire_env.h"
#include "pcc_kernels.h"

#ifndef PCC_RESTART_OCC
#define PCC_RESTART_OCC 2  // (a variant build cuts it for 4 -- 128 registers, heavy spilling -- to keep round 3's failing case under test)
#endif

namespace {

template <int NS, bool TRACE>
__global__ __launch_bounds__(4 * kWave, PCC_RESTART_OCC) void send_restart_kernel(Dev D, int read_buf, const void *actions, int actions_f64) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const uint32_t wave = blockIdx.x * 4u + wv, n_waves = gridDim.x * 4u;
    __shared__ EnvSlot<NS> s_slots[4][kSlots];
    // the restart lists of the partitions, one after the other, as one ranking
    uint32_t total = 0;
    for (uint32_t part = 0; part < D.parts; part++) total += *cls_count_of(D, list_view(D, read_buf, part), (uint32_t)kRestart);
    for (uint32_t g = wave; g < total; g += n_waves) {
        uint32_t part = 0, t = g;
        for (;;) {
            const uint32_t n_p = *cls_count_of(D, list_view(D, read_buf, part), (uint32_t)kRestart);
            if (t < n_p) break;
            t -= n_p;
            part++;
        }
        const bool has = lane == 0;
        const int64_t i = has ? (int64_t)cls_list_of(D, list_view(D, read_buf, part), (uint32_t)kRestart)[t] : 0;
        // new links and fresh state (ns:469-477) unless a flush already did all of it (pcc_get_state, a masked reset)
        if (has && D.env[i].resetting == 2) reset_env<NS>(D, i, nullptr);
        if (has && D.shadows && shadow_list(&D.env[D.n + i])) {   // its shadow (if any) was not usable: have it prepared for the episode after this one
            const uint32_t row = D.step_seq & 3u;
            D.refill_list[(size_t)row * (size_t)D.n + atomicAdd(&D.refill_count[row * kCntStride], 1u)] = (uint32_t)i;
        }
        // what one lane wrote is read by the others of this wavefront: a workgroup-scope fence is enough, and an
        // agent-scope one (__threadfence) writes back and invalidates the XCD's whole L2 under everybody's feet
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const int64_t i0 = (int64_t)__builtin_amdgcn_readfirstlane((int)i) | ((int64_t)__builtin_amdgcn_readfirstlane((int)(i >> 32)) << 32);
        for (int pass = 0; pass < 2; pass++) {  // the warm-up intervals: send, then retire by 8 lanes
            (void)send_wave_item<NS, TRACE, 1>(D, lane, i, has, true, 0xFFFFFFFFu, 1, (uint32_t)pass, actions, actions_f64, s_slots[wv]);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // records and state just written are read by other lanes
            if (lane < 8u) {
                Group g;
                g.lane = lane; g.shift = 0;
                (void)retire_env<NS, false, 8>(D, i0, g, 1, (uint32_t)pass, pass == 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        (void)send_wave_item<NS, TRACE, 1>(D, lane, i, has, true, 0xFFFFFFFFu, 0, 0, actions, actions_f64, s_slots[wv]);
    }
}

// refill_kernel: the NEXT episode of an env, prepared ahead of time in the env's shadow (block N + i, pcc_dev.h) -- new links
// (ns:469-477) and the two warm-up intervals (ns:478-479) -- so that the retire half can swap it in the moment the env
// finishes (retire_env) and the restart costs the step nothing.  With Philox uniforms an episode's links and draws are keyed
// by (env id, episode index): nothing of the running episode enters the next one's start.  Launched on a side stream of the
// handle two steps after the swap that emptied the shadow (by then the env has moved off the shadow's rings), never joined
// into a step that could wait for it: the envs it serves need their shadows an episode later.  The shadow's rings are private
// and of fixed size (tier 1): warm-up intervals that could overflow them are not sent -- the shadow stays unusable
// (resetting = 3) and the env restarts through the restart list as before.  No ring pool is touched here.
template <int NS>
__global__ __launch_bounds__(4 * kWave, PCC_RESTART_OCC) void refill_kernel(Dev D, uint32_t row, uint32_t fill_seq) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const uint32_t wave = blockIdx.x * 4u + wv, n_waves = gridDim.x * 4u;
    const uint32_t n_refill = D.refill_count[row * kCntStride];
    __shared__ EnvSlot<NS> s_slots[4][kSlots];
    for (uint32_t t = wave; t < n_refill; t += n_waves) {
        const bool has = lane == 0;
        const int64_t ie = (int64_t)D.refill_list[(size_t)row * (size_t)D.n + t];
        const int64_t i = D.n + ie;   // the shadow's block
        bool go = false;
        if (has && shadow_claim(&D.env[i])) {   // (listed -> being refilled: one wavefront, whoever else lists or holds the id)
            D.env[i].episode = D.env[ie].episode;   // the episode after the one the env is running
            D.env[i].fill_seq = fill_seq;
            D.env[i].params_gen = D.params_gen;     // (what its links are drawn from: the swap asks for the same generation)
            D.env[i].total_sent = 0;
            D.env[i].flags = 0;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int64_t k = sidx(D, s, i);
                D.snd[k].ring_base = D.shadow_rings + (size_t)(ie * NS + s) * tier_slot_bytes(D, kShadowTier);
                D.snd[k].ring_tier = (uint8_t)kShadowTier;
                for (int c = 0; c < kMaxTiers; c++) D.snd[k].ring_held[c] = 0;
            }
            reset_env<NS>(D, i, nullptr);
            go = true;
        }
        if (!__ballot(go)) continue;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        bool failed = false;
        for (int pass = 0; pass < 2 && !failed; pass++) {
            bool refused = false;
            (void)send_wave_item<NS, false, 1>(D, lane, i, has, true, 0xFFFFFFFFu, 1, (uint32_t)pass, nullptr, 0, s_slots[wv], 0u, nullptr, true, &refused);
            failed = refused;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (!failed && lane < 8u) {
                Group g;
                g.lane = lane; g.shift = 0;
                (void)retire_env<NS, false, 8>(D, i, g, 1, (uint32_t)pass, pass == 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        if (failed && has) D.env[i].resetting = 3;   // (these links' warm-up intervals do not fit a shadow's rings)
    }
    // (the host clears the row's count behind this launch: the retire launch four steps on files into it again)
}

}  // namespace

namespace pcc {

void launch_refill(const Dev &d, unsigned grid, hipStream_t st, uint32_t row, uint32_t fill_seq) {
    if (d.ns == 1) hipLaunchKernelGGL(refill_kernel<1>, dim3(grid), dim3(4 * kWave), 0, st, d, row, fill_seq);
    else hipLaunchKernelGGL(refill_kernel<2>, dim3(grid), dim3(4 * kWave), 0, st, d, row, fill_seq);
}


void launch_send_restart(const Dev &d, bool trace, unsigned grid, hipStream_t st, int read_buf, const void *actions, int actions_f64) {
#define PCC_R(NS_, TR_) hipLaunchKernelGGL((send_restart_kernel<NS_, TR_>), dim3(grid), dim3(4 * kWave), 0, st, d, read_buf, actions, actions_f64)
    if (d.ns == 1) { if (trace) PCC_R(1, true); else PCC_R(1, false); }
    else { if (trace) PCC_R(2, true); else PCC_R(2, false); }
#undef PCC_R
}

}  // namespace pcc
