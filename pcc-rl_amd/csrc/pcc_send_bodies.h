// pcc_send_bodies.h -- what a workgroup of the send half does, by kind:
//   light_body  the light items (pcc_send_item.h: send_light_item): lane rounds, a lane per env;
//   wave_body   the wave-path items and the team items (send_wave_item; pcc_wave_pass.h: heavy_mi / heavy_mi2).
// Both fit one register budget without spills (they share no register state: a light item hands its last lanes to the wave
// path through memory, and the wave path parks the envs of an item in LDS), so ONE launch can hold both kinds of workgroup --
// send_kernel (pcc_send.hip), the wave-path workgroups dispatched first, the light ones last: a compute unit's memory
// pipeline serves its oldest wavefronts first, and lane-round wavefronts as the oldest starve everybody else.  (As two
// kernels on two streams they were measured slower: the two launches start in the wrong order and the join costs ~15 us a
// step, profiles/r04_experiments.json.)
// Every workgroup works for ONE partition of the batch (pcc_dev.h "partitions": the one its XCD keeps working for).
#pragma once
#include "pcc_send_item.h"

namespace {

// positions per lane of a wave-path item's closed-form passes (heavy_mi's NP; team passes and the light items' stragglers: 4).
// 8 (passes of 512 positions: the per-pass work -- header, regime constants, token division, scans -- over twice the packets) was
// built in round 6, is exact (the parity suite ran through it) and is NOT faster: 0.0908 / 0.0935 ms per send launch at thresholds
// 480 / 384 against 0.0907 / 0.0921 with 4 (tools/ab_block.py, profiles/r06_knob_sweeps.json) -- the launch's tail is its longest
// lane-round items, and an env of 300-500 packets fills neither pass length.  -DPCC_WAVE_POS=8 builds it (40 KB of LDS per workgroup).
#ifndef PCC_WAVE_POS
#define PCC_WAVE_POS 4
#endif
constexpr int kWaveItemPos = PCC_WAVE_POS;

// POS: positions per lane of the wave-path items' passes in this kernel (the staging buffer holds a pass of one wavefront)
template <int NS, int POS = (NS == 1 ? kWaveItemPos : 4)>
struct SendLds {
    static constexpr int kPos = POS;
    EnvSlot<NS> slots[4][kSlots];      // send_wave_item's parked envs, per wavefront
    uint32_t tab[4][6][kClasses];      // wave_body's class table, per wavefront
    TeamX team;                        // what the wavefronts of a team pass tell each other
    double2 stage[4][POS * kWave + 1];   // heavy_mi / heavy_mi2 <.., STAGE>: the records of a 256-position pass on their way out, per wavefront
};
#ifndef PCC_STAGE_RECORDS
#define PCC_STAGE_RECORDS 1
#endif
constexpr bool kStageRecords = PCC_STAGE_RECORDS != 0;

// The light items of one workgroup (b of Q light workgroups): one item per wavefront and round, dealt statically -- the grid
// covers the worst case (every env light), a wavefront without an item leaves at once.  With work lists (view >= 0: the list set of the workgroup's
// partition; b and Q count within the partition) the envs of every class below the wave-path threshold, send_envs_per_wave (64) of a class at a time, longest class first;
// without (after a reset, the warm-up intervals, small batches) the envs in index order.  The four items of a workgroup
// share a compute unit, one per SIMD; dealt in snake order (workgroup b: ranks b, 2Q-1-b, 2Q+b, 4Q-1-b) a workgroup's items
// add up to about the same number of packets -- the lane rounds' scattered 16-byte record stores go through the compute
// unit's one address path.
template <int NS, bool TRACE, class LDS = SendLds<NS>>
__device__ __forceinline__ void light_body(const Dev &D, LDS &lds, const uint32_t lane, const uint32_t wv, const uint32_t b_wg,
                                           const uint32_t Q, const int view, const uint32_t tl_base, const int warm, const uint32_t warm_mi,
                                           const void *actions, const int actions_f64) {
    const uint32_t n_waves = Q * 4u;
    const uint32_t E = D.send_envs_per_wave;
    const bool listed = view >= 0;
    const int cls_heavy = D.use_cwnd ? kClasses : (D.heavy_predict >= 1e9 ? kClasses : class_of((float)D.heavy_predict));
    // lane l looks after light class cls_heavy - 1 - l (longest first); inclusive prefix of the items per class.
    // The longest classes (from light_half_predict packets up) go 32 envs to an item instead of 64: an iteration of the lane
    // rounds is one scattered store instruction, which costs the wavefront ~3.3 ns per line it touches (64 lanes: ~215 ns;
    // 32: ~100 ns, below the ~160 ns of arithmetic) -- and the longest light items are the launch's critical path, while
    // two thirds of the launch's wavefront slots idle (tools/microbench/store_bench5, tools/send_timeline.py)
    const int cls_mine = cls_heavy - 1 - (int)lane;
    const int cls_half = D.light_half_predict >= 1e9f ? kClasses : class_of(D.light_half_predict);
    const uint32_t E_mine = (cls_mine >= cls_half && E >= 2u) ? E / 2u : E;
    uint32_t n_mine = 0, items_mine = 0;
    if (listed && cls_mine >= 0) {
        n_mine = *cls_count_of(D, (uint32_t)view, (uint32_t)cls_mine);
        items_mine = (n_mine + E_mine - 1) / E_mine;
    }
    uint32_t incl = items_mine;
    for (int o = 1; o < kClasses; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += up;
    }
    const uint32_t n_items = listed ? rl_u32(incl, kClasses - 1) : (uint32_t)((D.n + E - 1) / E);
    for (uint32_t r0 = 0; r0 < n_items; r0 += n_waves) {
        const uint32_t b = (D.light_snake && (wv & 1u)) ? Q - 1u - b_wg : b_wg;
        // (more items than wavefronts -- the grid holds the workgroups that stay resident, pcc_sim.hip: the second round's items,
        // the shortest classes, go out in the opposite order: the wavefront that had the longest item gets the shortest)
        const uint32_t w_rank = wv * Q + b;
        const uint32_t t = r0 + (((r0 / n_waves) & 1u) ? n_waves - 1u - w_rank : w_rank);
        if (t >= n_items) continue;
        int64_t i;
        bool has;
        if (listed) {
            const uint64_t above = __ballot(incl > t);  // (lanes past the last class repeat the total: harmless, never first)
            const uint32_t L = (uint32_t)__ffsll((unsigned long long)above) - 1u;
            const uint32_t off = t - (rl_u32(incl, L) - rl_u32(items_mine, L));
            const uint32_t n_cls = rl_u32(n_mine, L);
            const uint32_t *list = cls_list_of(D, (uint32_t)view, (uint32_t)(cls_heavy - 1 - (int)L));
            const uint32_t E_cls = rl_u32(E_mine, L);
            const uint32_t idx = off * E_cls + lane;
            has = lane < E_cls && idx < n_cls;
            i = has ? (int64_t)list[idx] : 0;
        } else {
            i = (int64_t)t * E + lane;
            has = lane < E && i < D.n;
        }
        const bool prio = t < D.prio_light_items;
        if (prio) set_prio(D.prio_level);
        uint32_t pk = 0;
        const uint64_t left = send_light_item<NS, TRACE>(D, lane, i, has, tl_base + t, warm, warm_mi, actions, actions_f64, pk);
        if (left) {
            // the item's last lanes (see send_light_item) go on by the wave path, from the state the item just stored: what
            // one lane wrote is read by others of this wavefront -- a workgroup-scope fence orders that (same L1)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            (void)send_wave_item<NS, TRACE, 1, kStageRecords>(D, lane, i, ((left >> lane) & 1ull) != 0ull, false, 0xFFFFFFFFu, warm, warm_mi,
                                                                         actions, actions_f64, lds.slots[wv], 0u, nullptr, false, nullptr, lds.stage[wv]);
        }
        if (prio) set_prio(0u);
    }
}

// The wave-path and team items of one workgroup (b of the G wave workgroups of its partition).  Persistent: the wavefronts
// take work items until none are left:
//   * a wave-path item: envs of a class from heavy_predict packets up.  An item holds as many envs as make up about
//     heavy_item_packets packets (1-8): lanes 0..e-1 load an env each, so that the claim, the list entry and the state --
//     three dependent round trips -- are paid once per item; the wavefront then sends them one after the other;
//   * a team item: one env predicted above team_predict packets, sent by the four wavefronts of a workgroup together
//     (heavy_mi<.., 4>, 1 024 positions per pass).  The first workgroups take them -- workgroup b items b, b + n_tw, ... --
//     and then claim like everybody.
// Items are ranked by class, largest first.  The first item of a wavefront is dealt statically (no atomic), the rest comes
// off 16 sharded cursors (one returning atomic on one word saturates near 90 claims/us).  (A wavefront that finds its
// partition's cursors empty stops: going on with the next partition's items was built -- a loop around all of this -- and
// cost the kernel its spill-free register budget whichever way the hop count was kept.)
// FUSED (pcc_fused.hip): every env that is sent goes into the ready queue of the wave-path classes of this wavefront's XCD
// (fused_push), after the wavefronts that stored for it have drained their stores.
template <int NS, bool TRACE, bool FUSED = false, class LDS = SendLds<NS>>
__device__ __forceinline__ void wave_body(const Dev &D, LDS &lds, const uint32_t lane, const uint32_t wv, const uint32_t wave_wgs,
                                          const int read_buf, const void *actions, const int actions_f64, const uint32_t xcc = 0u,
                                          const uint32_t blk = blockIdx.x) {
    // workgroup blockIdx.x of the wave_wgs wave-path workgroups: number b_wg of the G of its partition.  (Computed from
    // the block index here: nothing of the launch's bookkeeping need stay in registers across
    // the items: the kernel sits at its register budget.)
    const uint32_t b_wg = blk >> D.parts_shift, G = wave_wgs >> D.parts_shift;   // (blk: this workgroup's number among the wave-path workgroups)
    const uint32_t wave = b_wg * 4u + wv;
    constexpr bool kTeams = NS == 1;
    const int cls_heavy = D.heavy_predict >= 1e9 ? kClasses : class_of((float)D.heavy_predict);
    const uint32_t team_wgs_max = G / 4u > 0u ? G / 4u : 1u;  // workgroups that may start with team items
    const bool teams_on = kTeams && D.team_predict < 1e9;
    const int cls_team = teams_on ? (class_of((float)D.team_predict) > cls_heavy ? class_of((float)D.team_predict) : cls_heavy) : kClasses;
    uint32_t (*tab)[kClasses] = lds.tab[wv];
    {
        constexpr bool home = true;
        const uint32_t pv = blk & (D.parts - 1u);
        const uint32_t view = list_view(D, read_buf, pv);
        // lane l < kClasses looks after class kClasses - 1 - l: largest class first
        const int cls_mine = kClasses - 1 - (int)lane;
        uint32_t n_mine = 0, items_mine = 0, n_team_mine = 0, e_mine = 1;
        if (lane < (uint32_t)kClasses && cls_mine >= cls_heavy) {
            n_mine = *cls_count_of(D, view, (uint32_t)cls_mine);
            const float pk = 8.0f * __expf(0.22314355f * ((float)cls_mine - 0.5f));  // 8 * 1.25^(c - 1/2)
            e_mine = (uint32_t)fminf(fmaxf(D.heavy_item_packets / pk, 1.0f), 8.0f);
            if (cls_mine >= cls_team) n_team_mine = n_mine;
            else items_mine = (n_mine + e_mine - 1) / e_mine;
        }
        uint32_t incl = items_mine, incl_team = n_team_mine;
        for (int o = 1; o < kClasses; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o), upt = (uint32_t)__shfl_up((int)incl_team, o);
            if (lane >= (uint32_t)o) { incl += up; incl_team += upt; }
        }
        const uint32_t n_items = rl_u32(incl, kClasses - 1), n_team = rl_u32(incl_team, kClasses - 1);
        // the class table lives in LDS (one copy per wavefront: no barrier needed), not in registers that would stay live across
        // every item: rows = inclusive item prefix, items, envs, envs per item, inclusive team prefix, team envs
        if (lane < (uint32_t)kClasses) {
            tab[0][lane] = incl; tab[1][lane] = items_mine; tab[2][lane] = n_mine; tab[3][lane] = e_mine;
            tab[4][lane] = incl_team; tab[5][lane] = n_team_mine;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // profile build: timeline slots of this partition's items (the light items' come first, pcc_send.hip; the item slots
        // end at 2 n: the last partition's range is cut there when n is not a multiple of 64 * parts)
        const uint32_t tl_base = (uint32_t)D.n + pv * D.part_envs, tl_end = 2u * (uint32_t)D.n;
        // ---- team items
        const uint32_t n_tw = n_team < team_wgs_max ? n_team : team_wgs_max;  // workgroups that have team items
        if constexpr (kTeams) {
            if (home && b_wg < n_tw) {
                if (D.prio_team) set_prio(D.prio_level);
                for (uint32_t tt = b_wg; tt < n_team; tt += n_tw) {
                    const uint64_t above = __ballot(lane < (uint32_t)kClasses && tab[4][lane & (kClasses - 1)] > tt);
                    const uint32_t L = (uint32_t)__ffsll((unsigned long long)above) - 1u;
                    const uint32_t off = tt - (uni_u32(tab[4][L]) - uni_u32(tab[5][L]));
                    const uint32_t *list = cls_list_of(D, view, kClasses - 1u - L);
                    const int64_t i = lane == 0 ? (int64_t)list[off] : 0;
                    (void)send_wave_item<NS, TRACE, kTeams ? kTeamMax : 1, kStageRecords>(D, lane, i, lane == 0, true, tl_base + n_items + tt < tl_end ? tl_base + n_items + tt : 0xFFFFFFFFu, 0, 0, actions,
                                                                           actions_f64, lds.slots[wv], wv, &lds.team, false, nullptr, lds.stage[wv]);
                    if constexpr (FUSED) {   // all four wavefronts stored records: each drains, then wavefront 0 publishes
                        fused_drain();
                        __syncthreads();
                        if (wv == 0u) fused_push(D, read_buf, xcc, 1u, 1ull, lane, i);
                    }
                }
                if (D.prio_team) set_prio(0u);
            }
        }
        // ---- wave-path items: the first one dealt statically to the workgroups without team items, the rest claimed
        const uint32_t n_static = (G - n_tw) * 4u;
        uint32_t t = n_items;   // (nothing dealt: claim)
        if (home && b_wg >= n_tw) {
            // workgroup b' of the Gs without team items, wavefront wv: rank wv * Gs + b' -- a workgroup's four items (and the
            // items of the workgroups that share its compute unit) are spread over the ranking, so every compute unit gets
            // about the same number of packets; wave_oldest_first = 0 deals a workgroup four consecutive ranks instead (the
            // largest items of every third of the ranking then meet on the same compute units: measured slower)
            const uint32_t Gs = G - n_tw, bs = b_wg - n_tw;
            t = D.wave_oldest_first ? wv * Gs + bs : bs * 4u + wv;
        }
        uint32_t *cursors = cursors_of(D, view);
        const uint32_t s_mine = wave % kShards;
        for (;;) {
            if (t >= n_items) {
                t = 0xFFFFFFFFu;
                if constexpr (FUSED) {
                    // every shard's cursor looked at side by side (lane k: shard s_mine + k), then ONE claim -- the serial look
                    // below is 16 dependent round trips for a wavefront that finds everything empty, and here that wavefront
                    // has retire work waiting
                    for (uint32_t tries = 0; tries < 4u && t == 0xFFFFFFFFu; tries++) {
                        const uint32_t sh = (s_mine + lane) % kShards;
                        const uint32_t seen = lane < kShards ? __hip_atomic_load(cursors + sh * kCursorStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
                        const uint64_t open = __ballot(lane < kShards && (uint64_t)seen * kShards + sh + n_static < (uint64_t)n_items);
                        if (!open) break;
                        const uint32_t k = (uint32_t)__ffsll((unsigned long long)open) - 1u;
                        uint32_t got = 0xFFFFFFFFu;
                        if (lane == k) {
                            const uint64_t cand = (uint64_t)atomicAdd(cursors + sh * kCursorStride, 1u) * kShards + sh + n_static;
                            if (cand < (uint64_t)n_items) got = (uint32_t)cand;
                        }
                        t = rl_u32(got, k);
                    }
                } else
                if (lane == 0) {
                    for (uint32_t k = 0; k < kShards && t == 0xFFFFFFFFu; k++) {
                        const uint32_t sh = (s_mine + k) % kShards;
                        uint32_t *cur = cursors + sh * kCursorStride;
                        const uint32_t seen = __hip_atomic_load(cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((uint64_t)seen * kShards + sh + n_static >= (uint64_t)n_items) continue;  // looks empty: no atomic
                        const uint64_t cand = (uint64_t)atomicAdd(cur, 1u) * kShards + sh + n_static;
                        if (cand < (uint64_t)n_items) t = (uint32_t)cand;
                    }
                }
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                if (t == 0xFFFFFFFFu) break;
            }
            const uint64_t above = __ballot(lane < (uint32_t)kClasses && tab[0][lane & (kClasses - 1)] > t);
            const uint32_t L = (uint32_t)__ffsll((unsigned long long)above) - 1u;
            const uint32_t e_cls = uni_u32(tab[3][L]), n_cls = uni_u32(tab[2][L]);
            const uint32_t off = t - (uni_u32(tab[0][L]) - uni_u32(tab[1][L]));
            const uint32_t *list = cls_list_of(D, view, kClasses - 1u - L);
            const uint32_t idx = off * e_cls + lane;
            const bool has = lane < e_cls && idx < n_cls;
            const int64_t i = has ? (int64_t)list[idx] : 0;
            const bool prio = t < D.prio_wave_items;
            if (prio) set_prio(D.prio_level);
            (void)send_wave_item<NS, TRACE, 1, kStageRecords, LDS::kPos>(D, lane, i, has, true, tl_base + t < tl_end ? tl_base + t : 0xFFFFFFFFu, 0, 0, actions, actions_f64, lds.slots[wv],
                                                                                       0u, nullptr, false, nullptr, lds.stage[wv]);
            if constexpr (FUSED) {
                fused_drain();
                fused_push(D, read_buf, xcc, 1u, __ballot(has), lane, i);
            }
            if (prio) set_prio(0u);
            t = n_items;  // forces a claim
        }
    }
}

}  // namespace
