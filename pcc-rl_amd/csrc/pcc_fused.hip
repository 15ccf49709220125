// pcc_fused.hip -- step_fused_kernel: BOTH halves of a full-size step in ONE launch, an env's retire half running as soon
// as ITS send half is done (no barrier between the halves).
//
// Why: as two launches (pcc_send.hip, pcc_retire.hip) each half waits for its own longest work -- the send launch is as
// long as its longest light item (a chain of ~380 lane-round iterations) on a machine that is a third busy, the retire
// launch as long as the workgroups of its largest envs -- so ~100 us of machine work take ~180 us.  Nothing couples the
// halves across envs (an env's retire half needs only its own send half: step_small_kernel has always relied on that), so
// here every wavefront of a grid that is about what the device holds at once
//   1. sends: the wave-path workgroups their team and wave-path items (wave_body), then everybody the light items off a
//      cursor (fused_light_loop) -- every item claimed dynamically, so nobody ever waits for a workgroup that is not running;
//   2. publishes every env whose send half is complete in one of the two READY QUEUES of the XCD it runs on (fused_push:
//      the envs of the light classes, and those of the wave-path classes);
//   3. retires: once no send item is left to claim, the wavefront takes envs off the ready queues OF ITS OWN XCD -- up to 8
//      at 8 lanes per env from the light queue, up to 4 at 16 lanes from the other, the same retire_env as the retire
//      launch -- and files them in the work lists of the next step, until every env of the batch has been published and
//      its XCD's queues are empty.
// Visibility inside the launch (pcc_dev.h "ready queues"): per-XCD L2s are not coherent with each other, and every way of
// making a send half visible ACROSS XCDs was measured to cost more than the launch gains.  The queues are therefore per
// physical XCD (XCC_ID from the hardware): producer and consumer share an L2 by construction -- not by the dispatcher's
// placement, which HIP does not promise -- so plain stores + s_waitcnt vmcnt(0) + an 8-byte granule are a complete hand-off.
// With the placement the hardware shows (block b on XCD b % 8) an XCD's queues hold exactly its partition's envs.
// Results do not depend on placement, dispatch order or timing: which wavefront sends or retires an env never changes a value.
// Every wait is bounded (fused_spin_ticks): a wait that gives up flags PCC_FLAG_INTERNAL.
//
// Replaces Network.run_for_dur (ns:123-205) for all envs of a step, as send_kernel + retire_kernel do.
#include "pcc_send_bodies.h"
#include "pcc_retire_env.h"
#include "pcc_kernels.h"

#ifndef PCC_FUSED_OCC
#define PCC_FUSED_OCC 4   // workgroups (4 wavefronts) per compute unit the register budget is cut for
#endif
#ifndef PCC_FUSED_OCC2
#define PCC_FUSED_OCC2 3  // ... of the two-sender builds (retire_env<2> needs ~156 registers)
#endif

namespace {

__device__ __forceinline__ uint32_t ld_u32_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The light items of the partition whose list set is `view`, off ONE cursor, longest class first (light_body deals the same
// items statically: here an item must belong to a wavefront that is running).  After an item: its envs into ready queue 0.
template <int NS, bool TRACE>
__device__ __forceinline__ void fused_light_loop(const Dev &D, SendLds<NS, 4> &lds, const uint32_t lane, const uint32_t wv, const int read_buf,
                                                 const uint32_t part, const void *actions, const int actions_f64, const uint32_t xcc) {
    const uint32_t view = list_view(D, read_buf, part);
    const uint32_t E = D.send_envs_per_wave;
    const int cls_heavy = D.heavy_predict >= 1e9 ? kClasses : class_of((float)D.heavy_predict);
    // lane l looks after light class cls_heavy - 1 - l (longest first); inclusive prefix of the items per class
    const int cls_mine = cls_heavy - 1 - (int)lane;
    const int cls_half = D.light_half_predict >= 1e9f ? kClasses : class_of(D.light_half_predict);
    const uint32_t E_mine = (cls_mine >= cls_half && E >= 2u) ? E / 2u : E;
    uint32_t n_mine = 0, items_mine = 0;
    if (cls_mine >= 0) {
        n_mine = *cls_count_of(D, view, (uint32_t)cls_mine);
        items_mine = (n_mine + E_mine - 1) / E_mine;
    }
    uint32_t incl = items_mine;
    for (int o = 1; o < kClasses; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += up;
    }
    const uint32_t n_items = rl_u32(incl, kClasses - 1);
    uint32_t *cur = fq_word(D, read_buf, part, kFLight);
    const uint32_t tl_base = part * (D.part_envs / 32u + (uint32_t)kClasses + 1u);   // profile build: the partition's timeline slots
    for (;;) {
        uint32_t t = 0xFFFFFFFFu;
        if (lane == 0 && ld_u32_agent(cur) < n_items) {   // (a plain look first: no atomic on an empty cursor)
            const uint32_t c = atomicAdd(cur, 1u);
            if (c < n_items) t = c;
        }
        t = uni_u32(t);
        if (t == 0xFFFFFFFFu) break;
        const uint64_t above = __ballot(incl > t);  // (lanes past the last class repeat the total: harmless, never first)
        const uint32_t L = (uint32_t)__ffsll((unsigned long long)above) - 1u;
        const uint32_t off = t - (rl_u32(incl, L) - rl_u32(items_mine, L));
        const uint32_t n_cls = rl_u32(n_mine, L);
        const uint32_t *list = cls_list_of(D, view, (uint32_t)(cls_heavy - 1 - (int)L));
        const uint32_t E_cls = rl_u32(E_mine, L);
        const uint32_t idx = off * E_cls + lane;
        const bool has = lane < E_cls && idx < n_cls;
        const int64_t i = has ? (int64_t)list[idx] : 0;
        uint32_t pk = 0;
        const bool prio = t < D.prio_light_items;   // (the longest light items are the launch's critical path)
        if (prio) set_prio(D.prio_level);
        const uint64_t left = send_light_item<NS, TRACE>(D, lane, i, has, tl_base + t, 0, 0u, actions, actions_f64, pk);
        const uint64_t live = __ballot(has);
        fused_drain();
        fused_push(D, read_buf, xcc, 0u, live & ~left, lane, i);
        if (left) {
            // the item's last lanes go on by the wave path, from the state the item just stored (pcc_send_bodies.h)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            (void)send_wave_item<NS, TRACE, 1>(D, lane, i, ((left >> lane) & 1ull) != 0ull, false, 0xFFFFFFFFu, 0, 0u, actions, actions_f64,
                                               lds.slots[wv]);
            fused_drain();
            fused_push(D, read_buf, xcc, 0u, left, lane, i);
        }
        if (prio) set_prio(0u);
    }
}

// One unit of retire work: the n_unit <= 64 / G envs named by entries [c, c + n_unit) of a ready queue, G lanes each; filed in
// the lists of buffer fill_buf (by the env's partition; one atomic per partition and class present in the wavefront).
// Returns false when a wait gave up.
template <int NS, int G>
__device__ __forceinline__ bool fused_retire_unit(const Dev &D, const uint32_t lane, const int read_buf, const int fill_buf, const uint32_t total,
                                                  const unsigned long long *ent,
                                                  const uint32_t c, const uint32_t n_unit, const uint32_t q, const uint32_t xcc,
                                                  const uint64_t t_claim, float *obs_out, float *reward_out, uint8_t *done_out,
                                                  double *steps_out) {
    constexpr uint32_t kPer = kWave / G;
    const uint32_t slot = lane / (uint32_t)G;
    // (a claim can run past the queue's tail -- fused_retire_loop -- and, in a small batch with hundreds of idle wavefronts, past
    // the queue's storage: such an entry can never be filled)
    // The unit's granules, ONE load instruction per look for the whole wavefront: lane k < n_unit reads entry c + k (a look
    // that every lane of every group executes for itself costs the compute unit's address path 16 cycles per instruction, and
    // with a few thousand wavefronts waiting that was most of its capacity: the lane rounds next to them ran 3x slower).
    // An entry below the queue's tail is reserved, its granule is on its way; one past it is filled by a later push -- or
    // never, when the step is complete: then it is dropped.  (A claim can also run past the queue's storage in a small batch
    // with hundreds of idle wavefronts: such an entry can never be filled.)
    const bool mine = lane < n_unit && c + lane < D.q_cap;
    unsigned long long e = 0ull;
    uint64_t ready = 0ull, valid = __ballot(mine);
    bool gave_up = false;
    {
        const uint64_t t0 = wall_clock64();
        for (uint32_t spins = 0;; spins++) {
            if (mine && !((ready >> lane) & 1ull)) e = __hip_atomic_load(ent + c + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ready = __ballot(mine && (uint32_t)(e >> 32) == D.step_seq);
            if ((ready & valid) == valid) break;
            if (spins < 4u) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(64);   // (~0.2 us, then ~1.7 us)
            if ((spins & 7u) == 7u) {   // now and then: is the step complete, and does the unit reach past the final tail?
                uint32_t pushed = lane < kXcds ? ld_u32_agent(fq_word(D, read_buf, lane, kFPushed)) : 0u;
                for (int o = 4; o; o >>= 1) pushed += (uint32_t)__shfl_xor((int)pushed, o);
                pushed = uni_u32(pushed);
                if (pushed >= total) {
                    const uint32_t tail = uni_u32(ld_u32_agent(fq_word(D, read_buf, xcc, kFTail + q)));
                    valid &= tail > c ? (tail - c >= 64u ? ~0ull : (1ull << (tail - c)) - 1ull) : 0ull;
                }
                if (wall_clock64() - t0 > (uint64_t)D.fused_spin_ticks) { gave_up = true; break; }
            }
        }
    }
    // lane l of group `slot` gets the group's env (entry c + slot) from the lane that read it
    const bool has = ((valid >> slot) & 1ull) != 0ull;
    e = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(e >> 32), (int)slot) << 32) | (uint32_t)__shfl((int)(uint32_t)e, (int)slot);
    if (gave_up) return false;
    // Nothing to acquire across XCDs (the queue is this XCD's: pcc_dev.h).  The workgroup-scope fence keeps the compiler from
    // moving the env's loads above the poll; fused_acquire = 2 (debug) invalidates this compute unit's L1 as well.
    if (D.fused_acquire >= 2u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint64_t t_ready = prof_on(D) ? wall_clock64() : 0;
    const int64_t i = has ? (int64_t)(uint32_t)e : D.n;
    Group g;
    g.lane = lane & (uint32_t)(G - 1);
    g.shift = lane & ~(uint32_t)(G - 1);
    float pred = -1.0f;
    if (i < D.n) pred = retire_env<NS, false, G>(D, i, g, 0, 0u, 0, 0, obs_out, reward_out, done_out, steps_out, nullptr, 0);
    // ---- file the wavefront's envs in the class lists of the next send (pcc_dev.h "work lists"), each in its partition's set
    const uint32_t env_id = pred != -1.0f ? (uint32_t)i : 0xFFFFFFFFu;
    const bool files = env_id != 0xFFFFFFFFu;
    const uint32_t cls = (uint32_t)class_of(pred);
    const uint32_t key = files ? (min(part_of(D, i), D.parts - 1u) << 8) | cls : 0xFFFFFFFFu;   // (partition, class)
    uint32_t rank = 0, same = 0, leader = slot;
#pragma unroll
    for (uint32_t l = 0; l < kPer; l++) {
        const uint32_t ok = (uint32_t)__shfl((int)key, (int)(l * G));
        const bool match = ok == key && ok != 0xFFFFFFFFu;
        same += match ? 1u : 0u;
        rank += (match && l < slot) ? 1u : 0u;
        if (match && l < leader) leader = l;
    }
    const uint32_t fview = list_view(D, fill_buf, files ? key >> 8 : 0u);
    uint32_t base = 0u;
    if (files && leader == slot && g.lane == 0u) base = atomicAdd(cls_count_of(D, fview, cls), same);
    base = (uint32_t)__shfl((int)base, (int)(leader * G));
    if (files && g.lane == 0u) cls_list_of(D, fview, cls)[base + rank] = env_id;
    if (prof_on(D) && lane == 0 && D.n >= 1024) {   // profile build: when the unit was claimed / ready / done (tools/fused_timeline.py)
        uint64_t *cnt = D.timeline + (int64_t)19 * D.n;   // (a running counter: the region holds the last few launches' units)
        const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long *>(cnt), 1ull) % (unsigned long long)(D.n / 2);
        uint64_t *w = cnt + 32 + at * 4;   // (words 16..31 of the region: the XCDs of the two launches' first blocks, pcc_send.hip / pcc_retire.hip)
        w[0] = ((uint64_t)D.step_seq << 32) | (q << 31) | n_unit;
        w[1] = t_claim; w[2] = t_ready; w[3] = wall_clock64();
    }
    return true;
}

// The retire half of the fused step: units off the two ready queues of XCD `xcc` until every env of the batch (`total`: what
// the lists of the step hold) has been published and both queues are empty.  A unit is what is ready, up to 64 / G envs; a
// wavefront that finds fewer waits a little for a full one (an 8-lane unit costs the wavefront about the same with 3 envs as
// with 8) unless everything has been published.
template <int NS>
__device__ __forceinline__ void fused_retire_loop(const Dev &D, const uint32_t lane, const int read_buf, const int fill_buf, const uint32_t xcc,
                                                  float *obs_out, float *reward_out, uint8_t *done_out, double *steps_out) {
    // envs in the lists of this step: every partition's class counts (the restart list is empty: pcc_sim.hip)
    uint32_t total = 0;
    for (uint32_t w = lane; w < D.parts * (uint32_t)kClasses; w += kWave) total += *cls_count_of(D, list_view(D, read_buf, w / (uint32_t)kClasses), w % (uint32_t)kClasses);
    for (int o = 32; o; o >>= 1) total += (uint32_t)__shfl_xor((int)total, o);
    total = uni_u32(total);
    uint32_t *head0 = fq_word(D, read_buf, xcc, kFHead), *head1 = fq_word(D, read_buf, xcc, kFHead + 1u);
    // what a look reads, ONE load instruction for the wavefront: lanes 0..3 the heads and tails of this XCD's queues, lanes
    // 8..15 every XCD's count of published envs
    const uint32_t *look = lane < 4u ? fq_word(D, read_buf, xcc, lane < 2u ? kFHead + lane : kFTail + (lane - 2u))
                                     : fq_word(D, read_buf, (lane - 8u) & (kXcds - 1u), kFPushed);
    const bool looks = lane < 4u || (lane >= 8u && lane < 8u + kXcds);
    const uint64_t t_start = wall_clock64();
    bool all = false;   // every env of the step has been published (then every tail is final)
    for (;;) {
        uint32_t q = 3u, c = 0u, n_unit = 0u;
        {
            uint32_t naps = 1u, waited = 0u;   // (a nap: 32 x 64 cycles, ~0.9 us)
            for (;;) {
                // (the counts are read by the same instruction as the tails; found complete they are read once more, tails AFTER
                // counts, so that the tails are final)
                uint32_t v = looks ? ld_u32_agent(look) : 0u;
                uint32_t pushed = (lane >= 8u && lane < 8u + kXcds) ? v : 0u;
                for (int o = 4; o; o >>= 1) pushed += (uint32_t)__shfl_xor((int)pushed, o);
                pushed = rl_u32(pushed, 8u);
                if (!all && pushed >= total) {
                    all = true;
                    v = looks ? ld_u32_agent(look) : 0u;
                }
                const uint32_t h0 = rl_u32(v, 0u), h1 = rl_u32(v, 1u), t0 = rl_u32(v, 2u), t1 = rl_u32(v, 3u);
                const bool eager = all || waited >= D.fused_partial_naps;
                if ((D.fused_debug & 4u) && !all) {   // (experiment: no retire work before every env is sent)
                    for (uint32_t k = 0; k < naps; k++) __builtin_amdgcn_s_sleep(32);
                    if (naps < D.fused_max_naps) naps *= 2u;
                    if (wall_clock64() - t_start > (uint64_t)D.fused_spin_ticks) break;
                    continue;
                }
                // A claim is one fetch-add of what was seen to be there; two wavefronts that saw the same envs both add, and the
                // later one's share runs past the tail it saw: entries that are not reserved yet.  It keeps them -- whoever
                // publishes next fills entries that are already spoken for, and the wavefront that holds them starts at once
                // (fused_retire_unit waits for the granules, and drops what lies past a final tail).
                // (the wave-path classes first: their envs are the long ones)
                if (h1 < t1 && (t1 - h1 >= (uint32_t)(kWave / 16) || eager)) {
                    n_unit = min(t1 - h1, (uint32_t)(kWave / 16));
                    if (lane == 0) c = atomicAdd(head1, n_unit);
                    q = 1u;
                    break;
                }
                if (h0 < t0 && (t0 - h0 >= (uint32_t)(kWave / 8) || eager)) {
                    n_unit = min(t0 - h0, (uint32_t)(kWave / 8));
                    if (lane == 0) c = atomicAdd(head0, n_unit);
                    q = 0u;
                    break;
                }
                if (all && h1 >= t1 && h0 >= t0) { q = 2u; break; }
                // nothing (or too little) ready: back off
                for (uint32_t k = 0; k < naps; k++) __builtin_amdgcn_s_sleep(32);
                waited += naps;
                if (naps < D.fused_max_naps) naps *= 2u;
                if (wall_clock64() - t_start > (uint64_t)D.fused_spin_ticks) break;   // (q = 3: gave up)
            }
        }
        q = uni_u32(q);
        c = uni_u32(c);
        n_unit = uni_u32(n_unit);
        if (q == 2u) break;
        bool ok = q < 2u;
        const uint64_t t_claim = prof_on(D) ? wall_clock64() : 0;
        // (the lane index through an opaque move, every time round: otherwise everything retire_env derives from it -- a few dozen
        // values per lane width -- is hoisted out of this loop and lives in registers across the whole body: 61 spilled registers)
        uint32_t lane_it = lane;
        asm volatile("" : "+v"(lane_it));
        if (q == 1u) ok = fused_retire_unit<NS, 16>(D, lane_it, read_buf, fill_buf, total, fused_entries(D, 1u, xcc), c, n_unit, 1u, xcc, t_claim, obs_out, reward_out, done_out, steps_out);
        else if (q == 0u) ok = fused_retire_unit<NS, 8>(D, lane_it, read_buf, fill_buf, total, fused_entries(D, 0u, xcc), c, n_unit, 0u, xcc, t_claim, obs_out, reward_out, done_out, steps_out);
        if (!ok) {   // never silent: a wait that gave up means envs were not stepped
            if (lane == 0) atomicOr(&D.env[0].flags, (uint32_t)PCC_FLAG_INTERNAL);
            break;
        }
    }
}

// a list buffer's counters, cursors and the fused step's words to zero
__device__ __forceinline__ void clear_list_buffer(const Dev &D, const int buf) {
    for (uint32_t w = threadIdx.x; w < D.parts * (uint32_t)(kClasses + 1); w += blockDim.x)
        *cls_count_of(D, list_view(D, buf, w / (uint32_t)(kClasses + 1)), w % (uint32_t)(kClasses + 1)) = 0u;
    for (uint32_t w = threadIdx.x; w < D.parts * kShards; w += blockDim.x)
        cursors_of(D, list_view(D, buf, w / kShards))[(w % kShards) * kCursorStride] = 0u;
    for (uint32_t w = threadIdx.x; w < kXcds * kFctlWords; w += blockDim.x)
        *fq_word(D, buf, w / kFctlWords, w % kFctlWords) = 0u;
}

// Grid (workgroups of 4 wavefronts): [0, wave_wgs) start with the wave-path work (dispatched first: pcc_send.hip explains why
// the order matters), the rest with the light items; both counts are multiples of the partitions, workgroup b SENDS for
// partition b % parts (pcc_dev.h "partitions") and RETIRES for the XCD it finds itself on.  read_buf: the lists this step
// reads; fill_buf: the lists it files into (clean when the launch starts); zero_buf: the third buffer, cleared here for the
// step after.
template <int NS, bool TRACE>
__global__ __launch_bounds__(4 * kWave, NS == 2 ? PCC_FUSED_OCC2 : PCC_FUSED_OCC) void step_fused_kernel(Dev D, int read_buf, int fill_buf, int zero_buf,
                                                                                              uint32_t wave_wgs, uint32_t light_front, int retire_on, const void *actions, int actions_f64,
                                                                                              float *obs_out, float *reward_out, uint8_t *done_out,
                                                                                              double *steps_out) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    if (blockIdx.x == 0) clear_list_buffer(D, zero_buf);
    if (prof_on(D) && blockIdx.x == 0 && threadIdx.x == 0 && D.n >= 1024) {   // profile build: when this launch started, and which it is
        D.timeline[(int64_t)19 * D.n + 1] = wall_clock64();
        D.timeline[(int64_t)19 * D.n + 2] = D.step_seq;
    }
    __shared__ SendLds<NS, 4> lds;   // (the frozen one-launch step keeps 4 positions per lane: its register and LDS budget are what they were)
    const uint32_t part = blockIdx.x & (D.parts - 1u);
    const uint32_t xcc = xcc_id();
    // (light_front: the workgroups in FRONT of the wave-path ones start with the light items -- the rest of the light-first
    // workgroups follow behind them)
    if (blockIdx.x >= light_front && blockIdx.x - light_front < wave_wgs)
        wave_body<NS, TRACE, true, SendLds<NS, 4>>(D, lds, lane, wv, wave_wgs, read_buf, actions, actions_f64, xcc, blockIdx.x - light_front);
    fused_light_loop<NS, TRACE>(D, lds, lane, wv, read_buf, part, actions, actions_f64, xcc);
    // (retire_on = 0, PCC_TUNE_FUSED = 2: an experiment -- this launch is the send half only, a retire launch follows)
    if (retire_on) fused_retire_loop<NS>(D, lane, read_buf, fill_buf, xcc, obs_out, reward_out, done_out, steps_out);

}

// (the fused step needs the buffer it files into clean BEFORE it starts; it clears the next one itself, so this runs only
// behind a step of the other kind)
__global__ void clear_list_buffer_kernel(Dev D, int buf) { clear_list_buffer(D, buf); }

template <int NS, bool TRACE>
int resident_blocks() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, step_fused_kernel<NS, TRACE>, 4 * kWave, 0) != hipSuccess) return 0;
    return nb;
}

}  // namespace

namespace pcc {

void launch_step_fused(const Dev &d, bool trace, unsigned grid, unsigned wave_wgs, unsigned light_front, hipStream_t st, int read_buf, int fill_buf, int zero_buf,
                       int retire_on, const void *actions, int actions_f64, float *obs_out, float *reward_out, uint8_t *done_out, double *steps_out) {
#define PCC_F(NS_, TR_)                                                                                                              \
    hipLaunchKernelGGL((step_fused_kernel<NS_, TR_>), dim3(grid), dim3(4 * kWave), 0, st, d, read_buf, fill_buf, zero_buf, wave_wgs, light_front, retire_on, actions, \
                       actions_f64, obs_out, reward_out, done_out, steps_out)
    if (d.ns == 1) { if (trace) PCC_F(1, true); else PCC_F(1, false); }
    else { if (trace) PCC_F(2, true); else PCC_F(2, false); }
#undef PCC_F
}

void launch_clear_list_buffer(const Dev &d, hipStream_t st, int buf) {
    hipLaunchKernelGGL(clear_list_buffer_kernel, dim3(1), dim3(256), 0, st, d, buf);
}

int fused_resident_blocks(int ns, bool trace) {
    if (ns == 1) return trace ? resident_blocks<1, true>() : resident_blocks<1, false>();
    return trace ? resident_blocks<2, true>() : resident_blocks<2, false>();
}

}  // namespace pcc
