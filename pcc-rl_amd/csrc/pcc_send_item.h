// pcc_send_item.h -- one work item of the send half: the SEND events of the coming monitor interval (ns:155-178, with
// apply_rate_delta ns:235-241, 275-281 in front) for the envs the lanes of a wavefront were given.
//   * load_env / store_env: an env's state into the lane that owns it (action applied, ring tier checked) and back;
//   * send_light_item: up to 64 envs of about the same predicted packet count, a lane each, in rounds;
//   * send_wave_item:  the envs held by the lanes, one after the other by all 64 lanes (heavy_mi / heavy_mi2), or ONE env
//                      by the W wavefronts of a workgroup (a team item).
// Which path sends an env is a performance choice only: every path is exact.  A light item hands the lanes that are still
// sending when (nearly) all others are done to the wave path THROUGH MEMORY -- it stores every env's state, and
// send_wave_item loads the stragglers afresh (fresh = false: the action is applied, part of the interval is sent) -- so
// that no lane state stays in registers across the wave passes: the two paths share a kernel without sharing a register
// budget (round 3's single send_item kept ~45 registers per lane live across heavy_mi: 56-160 bytes of scratch per lane
// and 100-240 spilled scalar registers in every build of the send kernel).
#pragma once
#include "pcc_wave_pass.h"

namespace {

template <int NS>
struct LaneEnv {
    double dl, lr, maxq, ebw, q, tu, end;
    uint32_t episode, mi, gid, flags;
    double gap[NS], nsend[NS];
    uint32_t ta[NS], td[NS], ha[NS], hd[NS], sent[NS];
    RingRef rings[NS];
    const double *trace;
    uint32_t thr;        // u32_to_unit(x) < lr  <=>  x < thr (ceil(lr * 2^32): the scaling is exact) ...
    bool always;         // ... unless lr >= 1
    bool live, run;      // the lane has an env to send for / its interval is not empty (ns:128)
};

// fresh: the interval starts here (the action is applied, nothing of it is sent); otherwise the lane rounds of a light item
// have sent part of it and stored the state (send_light_item's stragglers).
// W > 1: every wavefront of the team loads the same env and computes alike; wavefront 0 stores.
// no_promote (the refill of a shadow, pcc_send_restart.hip): the rings are what they are -- an interval that could overflow
// them is not sent at all (E.run = false) and flagged PCC_FLAG_INTERNAL in E.flags for the caller to see (never stored).
template <int NS, bool TRACE, int W>
__device__ __forceinline__ LaneEnv<NS> load_env(const Dev &D, const uint32_t lane, const int64_t i, const bool in_range,
                                                const bool fresh, const int warm, const uint32_t warm_mi, const void *actions,
                                                const int actions_f64, const uint32_t wv, const bool no_promote = false) {
    LaneEnv<NS> E;
    const bool writer = W == 1 || wv == 0u;
    E.live = in_range && !(warm && !D.env[in_range ? i : D.n].resetting);
    // A lane without an env reads block N (env 0's shadow: always there, never sent by this launch) -- not env 0's: in the
    // fused step (pcc_fused.hip) a compute unit must not load lines of an env it does not send, or its L1 holds a copy that is
    // stale by the time it retires that env (observed: env 0 retired from the L1 copy a neighbour's idle lanes had loaded)
    const int64_t ii = E.live ? i : D.n;
    E.dl = D.env[ii].dl; E.lr = D.env[ii].lr; E.maxq = D.env[ii].maxq; E.ebw = D.env[ii].ebw;
    E.q = D.env[ii].q; E.tu = D.env[ii].tu;
    const double now = D.env[ii].now;
    E.end = now + D.env[ii].run_dur;  // ns:124
    E.episode = D.env[ii].episode - 1;
    E.mi = warm ? warm_mi : D.env[ii].steps + 2;
    E.gid = D.gid_base + (uint32_t)env_of(D, ii);
    E.flags = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = sidx(D, s, ii);
        double rate = D.snd[k].rate;
        const bool act = fresh && !warm && E.live;
        if (act) {
            // (the action's address is computed HERE, from the env index behind a barrier the optimizer cannot see through: hoisted
            // out of the wave path's loop over an item's envs, the index times 2 or NS or the address itself stayed live across
            // heavy_mi and was the one value the send kernel spilled to scratch)
            uint32_t i_here = (uint32_t)ii;
            asm volatile("" : "+v"(i_here));
            const int64_t a = D.use_cwnd ? (int64_t)i_here * 2 : (int64_t)i_here * NS + s;  // USE_CWND: [rate action, cwnd action] per env
            const char *ap = (const char *)actions + (a << (actions_f64 ? 3 : 2));
            double delta = actions_f64 ? *(const double *)ap : (double)*(const float *)ap;
            if (delta != delta) { delta = 0.0; E.flags |= PCC_FLAG_BAD_ACTION; }  // NaN: never silent, never in the clock
            delta *= D.delta_scale;
            rate = delta >= 0.0 ? rate * (1.0 + delta) : rate / (1.0 - delta);
            if (rate > kMaxRate) rate = kMaxRate;
            if (rate < kMinRate) rate = kMinRate;
            if constexpr (W == 1) D.snd[k].rate = rate;
        }
        if constexpr (W > 1) {  // the new rate is stored once every wavefront of the team has read the old one
            __syncthreads();
            if (writer && act) D.snd[k].rate = rate;
        }
        E.gap[s] = 1.0 / rate;  // ns:161
        E.nsend[s] = D.snd[k].next_send;
        E.ta[s] = D.snd[k].ta; E.td[s] = D.snd[k].td;
        E.ha[s] = D.snd[k].ha; E.hd[s] = D.snd[k].hd;
        E.sent[s] = fresh ? 0u : D.snd[k].mi_sent;
    }
    E.trace = TRACE ? D.trace + ii * D.trace_stride : nullptr;
    E.run = E.live && now < E.end;
    // ---- ring tier: an upper bound of this MI's packets per sender is known up front (the send
    // times advance by gap up to rounding; one more SEND can follow in the retire half), so rings
    // that could overflow are moved to a bigger tier now, by the whole wavefront
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = sidx(D, s, ii);
        if (fresh) {
            uint32_t want = 0;
            const uint32_t tier_now = (uint32_t)D.snd[k].ring_tier;
            // (a sender whose episode was just swapped in still sits on its former shadow's rings: it moves into storage of its
            // own now, whatever the interval needs, so that the shadow can be refilled)
            const bool borrowed = E.live && tier_now == kTierBorrowed;
            if (E.run || borrowed) {
                const double ahead = E.nsend[s] < E.end ? (E.end - E.nsend[s]) / E.gap[s] + 4.0 : 1.0;
                const uint32_t n_max = ahead < 1e9 ? (uint32_t)ahead : 1000000000u;
                want = tier_for(D, E.ta[s] - E.ha[s] + n_max, E.td[s] - E.hd[s] + n_max);
                if (borrowed && want >= (uint32_t)D.n_tiers) want = (uint32_t)D.n_tiers - 1u;  // (it has to move; an overflow is flagged later)
            }
            if (no_promote) {
                if (E.run && want > tier_now) { E.run = false; E.flags |= PCC_FLAG_INTERNAL; }
            } else {
                uint64_t pm = __ballot((borrowed && want < (uint32_t)D.n_tiers) ||
                                       (E.run && tier_now != kTierBorrowed && want > tier_now && want < (uint32_t)D.n_tiers));
                char *own0 = D.tier_base[0] + (size_t)(env_of(D, ii) * NS + s) * tier_slot_bytes(D, 0);
                while (pm && writer) {
                    const uint32_t l = (uint32_t)__ffsll((unsigned long long)pm) - 1u;
                    pm &= pm - 1ull;
                    if (!promote_rings(D, lane, l, k, want, E.ha[s], E.ta[s], E.hd[s], E.td[s], own0) && lane == l) E.flags |= PCC_FLAG_POOL_EXHAUSTED;
                }
                if constexpr (W > 1) __syncthreads();  // the other wavefronts of a team read the address wavefront 0 just stored
            }
        }
        E.rings[s] = ring_ref(D, k);
    }
    const double thr_d = ceil(E.lr * 4294967296.0);
    E.always = thr_d >= 4294967296.0;
    E.thr = E.always ? 0xFFFFFFFFu : (thr_d > 0.0 ? (uint32_t)thr_d : 0u);
    return E;
}

template <int NS>
__device__ __forceinline__ void store_env(const Dev &D, const int64_t i, LaneEnv<NS> &E) {
    D.env[i].q = E.q; D.env[i].tu = E.tu;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        // never silent: more packets in flight than a ring holds means records were overwritten
        if (E.ta[s] - E.ha[s] > E.rings[s].cap || E.td[s] - E.hd[s] > 2u * E.rings[s].cap) E.flags |= PCC_FLAG_RING_OVERFLOW;
        const int64_t k = sidx(D, s, i);
        D.snd[k].next_send = E.nsend[s];
        D.snd[k].ta = E.ta[s]; D.snd[k].td = E.td[s];
        D.snd[k].mi_sent = E.sent[s];
    }
    if (E.flags) D.env[i].flags |= E.flags;
}

// profile build: the 8 words of a work item's timeline record (tools/send_timeline.py): start, end of the lane rounds /
// of the state loads, end (100 MHz ticks), envs sent by the wave path | the last of them << 16, packets of the wavefront,
// packets of its largest env, packets sent by the wave path, live lanes | closed-form passes << 8 | chain + serial passes << 24
template <int NS>
__device__ __forceinline__ void timeline_record(const Dev &D, const uint32_t lane, const uint32_t tl_slot, const LaneEnv<NS> &E,
                                                uint64_t tl0, uint64_t tl1, uint64_t heavy_envs, uint64_t heavy_pk,
                                                uint64_t closed, uint64_t other, uint64_t last_env, uint64_t cycles = 0) {
    uint64_t sum = E.live ? E.sent[0] : 0, mx = sum, hp = heavy_pk;
    for (int o = 32; o; o >>= 1) {
        sum += __shfl_xor(sum, o);
        hp += __shfl_xor(hp, o);
        const uint64_t other_mx = __shfl_xor(mx, o);
        mx = other_mx > mx ? other_mx : mx;
    }
    const uint64_t live_n = (uint64_t)__popcll(__ballot(E.live));
    if (lane == 0 && tl_slot != 0xFFFFFFFFu) {
        uint64_t *w = D.timeline + (int64_t)tl_slot * 8;
        w[0] = tl0; w[1] = tl1; w[2] = wall_clock64(); w[3] = heavy_envs | (last_env << 16); w[4] = sum; w[5] = mx; w[6] = hp;
        w[7] = live_n | (closed << 8) | (other << 24) | (cycles << 32);   // (cycles: light items only, whose pass counts are 0)
    }
}

// ---- the lane rounds' packet, branch-free (round 6).  The same operations of ns:66-84 in the reference's order as
// link_send (pcc_dev.h), other selects -- one basic block, no exec-mask regions, a shorter chain from q to q':
//   qcur    = max(0, q - (t - tu))                            ns:66-67
//   grown   = qcur + 1/bw   (= 1/bw + qcur: the operands of ns:79 and ns:82 commute, the sums are the same double)
//   dropped = lost at random || grown > maxq                  ns:73, 79 as ONE compare, against (lost ? a negative number : maxq)
//   q'      = dropped ? (lost ? q : qcur) : grown             ns:74, 80-82: the inner select does not wait for the compare
//   tu'     = lost ? tu : t                                   ns:76
// a4 / d4 are the ring indices times 16 (byte offsets before masking; they wrap like the indices do).
struct LightState {
    double q, tu, t;
    uint32_t a4, d4;
};
__device__ __forceinline__ void light_packet(LightState &S, const bool lost, const double dl, const double maxq, const double ebw,
                                             const double gap, char *base, const uint32_t mask_b, const uint32_t dmask_b,
                                             const uint32_t cap_b, const bool store) {
    const double t = S.t;
    const double qcur = max0(S.q - (t - S.tu));
    const double grown = qcur + ebw;
    const double lat0 = dl + qcur;            // ns:170
    // (maxq > 0 and grown >= 1/bw > 0: any negative limit makes the compare true)
    const double lim = __hiloint2double(lost ? (int)0xBFF00000u : __double2hiint(maxq), __double2loint(maxq));
    const bool dropped = grown > lim;
    const double keep = lost ? S.q : qcur;
    S.q = dropped ? keep : grown;
    S.tu = lost ? S.tu : t;
    double2 rec;
    rec.x = t + lat0;                         // ns:174
    rec.y = lat0;                             // ns:173
    const uint32_t off_a = S.a4 & mask_b, off_d = cap_b + (S.d4 & dmask_b);
    if (store) st_rec(reinterpret_cast<double2 *>(base + (dropped ? off_d : off_a)), rec);
    const uint32_t inc = dropped ? 16u : 0u;
    S.d4 += inc;
    S.a4 += 16u - inc;
    S.t = t + gap;                            // ns:161
}
// ... and for two senders on the shared link: the packet of the (time, sender id) merge, the sender a select (equal times: sender
// 0 first, the heap's order; ns:42-43).  Ring indices times 16, per sender.
struct Light2State {
    double q, tu, nsend[2];
    uint32_t a4[2], d4[2];   // ring positions in bytes; the packets a sender sent in the rounds are their growth / 16
};
__device__ __forceinline__ void light_packet2(Light2State &S, const bool lost, const double dl, const double maxq, const double ebw,
                                              const double gap0, const double gap1, char *base0, char *base1, const uint32_t mask_b0,
                                              const uint32_t mask_b1, const uint32_t dmask_b0, const uint32_t dmask_b1, const uint32_t cap_b0,
                                              const uint32_t cap_b1) {
    const bool s1 = S.nsend[1] < S.nsend[0];
    const double t = s1 ? S.nsend[1] : S.nsend[0];
    const double n0 = S.nsend[0] + gap0, n1 = S.nsend[1] + gap1;   // ns:161 for whichever of the two sends
    const double qcur = max0(S.q - (t - S.tu));
    const double grown = qcur + ebw;
    const double lat0 = dl + qcur;            // ns:170
    const double lim = __hiloint2double(lost ? (int)0xBFF00000u : __double2hiint(maxq), __double2loint(maxq));
    const bool dropped = grown > lim;
    const double keep = lost ? S.q : qcur;
    S.q = dropped ? keep : grown;
    S.tu = lost ? S.tu : t;
    double2 rec;
    rec.x = t + lat0;                         // ns:174
    rec.y = lat0;                             // ns:173
    const uint32_t a4 = s1 ? S.a4[1] : S.a4[0], d4 = s1 ? S.d4[1] : S.d4[0];
    const uint32_t off_a = a4 & (s1 ? mask_b1 : mask_b0), off_d = (s1 ? cap_b1 : cap_b0) + (d4 & (s1 ? dmask_b1 : dmask_b0));
    st_rec(reinterpret_cast<double2 *>((s1 ? base1 : base0) + (dropped ? off_d : off_a)), rec);
    // the sender's ring position by multiply-adds with 0 / 1 (v_mad_u32_u24), not by four more selects
    const uint32_t is1 = s1 ? 1u : 0u, is0 = 1u - is1;
    const uint32_t inc_d = dropped ? 16u : 0u, inc_a = 16u - inc_d;
    S.a4[0] = __umul24(inc_a, is0) + S.a4[0]; S.a4[1] = __umul24(inc_a, is1) + S.a4[1];
    S.d4[0] = __umul24(inc_d, is0) + S.d4[0]; S.d4[1] = __umul24(inc_d, is1) + S.d4[1];   // ns:260-262
    S.nsend[0] = s1 ? S.nsend[0] : n0;
    S.nsend[1] = s1 ? n1 : S.nsend[1];
}
#ifndef PCC_LIGHT_BLOCKS
#define PCC_LIGHT_BLOCKS 1
#endif
constexpr int kLightBlocks = PCC_LIGHT_BLOCKS;   // Philox blocks (4 packets each) per loop body of the lane rounds
__device__ __forceinline__ void light_philox(const Dev &D, uint32_t blk, uint32_t mi, uint32_t episode, uint32_t gid, uint32_t (&w)[4]) {
    if (prof_skip(D, 8)) { w[0] = blk * 2654435761u; w[1] = w[0] ^ gid; w[2] = w[1] * 40503u; w[3] = w[2] ^ mi; }
    else philox4x32_10_sched(blk, mi, episode, gid, D.key0, D.key1, w);
}

// A LIGHT item: lane l sends for env i (or for none).  Lane-serial rounds of round_packets packets per env; after a
// round, if at most takeover_lanes (default: one) lanes still have packets to send, they are the tail everybody else
// would wait for: the item stores every env and returns their mask -- the caller sends them by the wave path
// (send_wave_item, fresh = false), which sends ONE env's packets much faster than a lone lane does.  (The envs of a light
// item were filed together because they are about the same length, so the lanes normally finish within a round of each
// other and the mask is empty.)
// packets_out: the packets the item's lanes sent (wave-uniform; launch statistics).
template <int NS, bool TRACE>
__device__ __forceinline__ uint64_t send_light_item(const Dev &D, const uint32_t lane, const int64_t i, const bool in_range,
                                                    const uint32_t tl_slot, const int warm, const uint32_t warm_mi,
                                                    const void *actions, const int actions_f64, uint32_t &packets_out) {
    packets_out = 0u;
    LaneEnv<NS> E = load_env<NS, TRACE, 1>(D, lane, i, in_range, true, warm, warm_mi, actions, actions_f64, 0u);
    if (!__ballot(E.live)) return 0ull;
    const uint64_t tl0 = prof_on(D) ? wall_clock64() : 0;
    const uint64_t cy0 = prof_on(D) ? __builtin_readcyclecounter() : 0;   // (shader clock: the item's cycles go into the record's last word)
    const int64_t ii = E.live ? i : D.n;   // (see load_env)
    bool active = false;
    uint64_t tl_r1 = 0;
    uint32_t tl_trip = 0;   // (profile build: trips of the unchecked-block loop so far, all rounds)
    if (NS == 1) {
        const double dl = E.dl, lr = E.lr, maxq = E.maxq, ebw = E.ebw, end = E.end;
        const uint32_t thr = E.thr, episode = E.episode, mi = E.mi, gid = E.gid;
        const bool always = E.always;
        const double *trace = E.trace;
        char *base = E.rings[0].base;
        const uint32_t mask_b = (E.rings[0].cap - 1u) << 4, dmask_b = (2u * E.rings[0].cap - 1u) << 4, cap_b = E.rings[0].cap << 4;
        double q = E.q, tu = E.tu;
        if (D.use_cwnd) {
            // ---- USE_CWND (ns:54, 251-255, 158-160): a SEND goes out only while fewer than cwnd
            // packets are unacknowledged.  That couples the SEND stream to the notifications, so this
            // path is lane-serial with two cursors over the lane's own rings: everything acknowledged
            // or reported lost at or before the SEND time (ACK events sort before a SEND of the same
            // time, ns:42-43) is no longer in flight.  A blocked SEND still passes through the link's
            // queue and takes its loss draw (ns:170-175 are outside the `if`): it updates (q, tu) and
            // the RNG position, but leaves no record and is not counted as sent.
            uint32_t cw = D.snd[ii].cwnd;
            if (!warm && E.live) {  // apply_cwnd_delta + set_cwnd: ns:243-249, 283-289
                const int64_t ai = ii * 2 + 1;
                double delta = actions_f64 ? ((const double *)actions)[ai] : (double)((const float *)actions)[ai];
                if (delta != delta) { delta = 0.0; E.flags |= PCC_FLAG_BAD_ACTION; }
                delta *= D.delta_scale;
                const double c = delta >= 0.0 ? (double)cw * (1.0 + delta) : (double)cw / (1.0 - delta);
                cw = c >= 5000.0 ? 5000u : (c < 4.0 ? 4u : (uint32_t)c);  // int(), then [MIN_CWND, MAX_CWND] (ns:33-34)
                D.snd[ii].cwnd = cw;
            }
            const double2 *acc = E.rings[0].accepted(), *drp = E.rings[0].dropped();
            const uint32_t amask_r = E.rings[0].mask(), dmask_r = E.rings[0].dmask();
            double t = E.nsend[0];
            uint32_t a = E.ta[0], d = E.td[0], pa = E.ha[0], pd = E.hd[0], draws = 0, nsent = 0;
            const uint32_t ep0 = D.env[ii].ep_draws;
            while (E.run && t < end) {
                while (pa != a && ld_t1(acc + (pa & amask_r)) + dl <= t) pa++;
                while (pd != d && ld_t1(drp + (pd & dmask_r)) + dl <= t) pd++;
                uint32_t extra = 0;  // later members of a near group of drops that are due although record pd is not
                if (pd != d) {
                    double tp = ld_t1(drp + (pd & dmask_r));
                    if (near_time(tp + dl, t)) {
                        for (uint32_t k = pd + 1; k != d; k++) {
                            const double tk = ld_t1(drp + (k & dmask_r));
                            if (!near_time(tk, tp)) break;
                            if (tk + dl <= t) extra++;
                            tp = tk;
                        }
                    }
                }
                const bool can_send = (a - pa) + (d - pd) - extra < cw;
                double u;
                if (TRACE) {
                    const uint64_t pos = (uint64_t)ep0 + draws;
                    if ((int64_t)pos >= D.trace_stride) { E.flags |= PCC_FLAG_TRACE_OVERRUN; u = 1.0; }
                    else u = trace[pos];
                } else {
                    u = philox_packet_uniform(D, gid, episode, mi, draws);
                }
                draws++;
                bool dropped;
                const double2 rec = link_send(t, u < lr, dl, maxq, ebw, q, tu, dropped);
                if (can_send) {
                    const uint32_t off = dropped ? cap_b + ((d << 4) & dmask_b) : ((a << 4) & mask_b);
                    st_rec(reinterpret_cast<double2 *>(base + off), rec);
                    a += dropped ? 0u : 1u;
                    d += dropped ? 1u : 0u;
                    nsent++;
                }
                t += E.gap[0];  // ns:161: the next SEND is scheduled either way
            }
            if (E.live) {
                D.env[ii].mi_draws = draws;
                D.env[ii].ep_draws = ep0 + draws;
            }
            E.nsend[0] = t;
            E.sent[0] = nsent;
            E.ta[0] = a; E.td[0] = d;
        } else {
            const double gap = E.gap[0];
            double t = E.nsend[0];
            uint32_t a = E.ta[0], d = E.td[0];
            active = E.run;
            if constexpr (!TRACE) {
                // Philox uniforms: the loop body is ONE basic block per kLightBlocks Philox blocks (4 packets each) -- the
                // branch-free packet (light_packet) and the Philox rounds of the NEXT blocks, which depend on nothing the packets
                // compute: the compiler interleaves the two instruction streams, so the multiplies fill the latency of the
                // recurrence's dependent fp64 operations instead of running in front of them (round 6: 250 -> see
                // profiles/r06_lane_round_microbench.txt ns per iteration; the longest light item is the launch's critical path).
                // The first `safe` packets are certainly before `end` (t advances by gap up to rounding; two packets of margin),
                // so whole blocks run without the fp64 exit test.
                LightState S;
                S.q = q; S.tu = tu; S.t = t; S.a4 = a << 4; S.d4 = d << 4;
                const uint32_t a4_0 = S.a4, d4_0 = S.d4;
                uint32_t blk = 0;  // Philox block = packets sent in this MI / 4
                for (;;) {
                    if (active) {
                        const double ahead = (end - S.t) / gap - 2.0;
                        uint32_t safe4 = ahead >= 4.0 ? (uint32_t)fmin(ahead, (double)D.round_packets) >> 2 : 0u;
                        uint32_t budget4 = D.round_packets / 4 - safe4;
                        uint32_t w[kLightBlocks][4];
#pragma unroll
                        for (int b = 0; b < kLightBlocks; b++) light_philox(D, blk + b, mi, episode, gid, w[b]);
                        for (; safe4 >= (uint32_t)kLightBlocks; safe4 -= kLightBlocks) {
                            if (prof_on(D)) {   // profile build: a stamp every 8 trips (32 packets) of this wavefront, 16 per item (tools/send_timeline.py)
                                if ((tl_trip & 7u) == 0u && (tl_trip >> 3) < 16u && tl_slot < 4096u && D.n >= 32768 &&
                                    lane == (uint32_t)__ffsll((unsigned long long)__ballot(true)) - 1u)
                                    D.timeline[((int64_t)8192 + 2 * (int64_t)tl_slot) * 8 + (tl_trip >> 3)] = wall_clock64();
                                tl_trip++;
                            }
                            uint32_t wn[kLightBlocks][4];
#pragma unroll
                            for (int b = 0; b < kLightBlocks; b++) light_philox(D, blk + kLightBlocks + b, mi, episode, gid, wn[b]);
#pragma unroll
                            for (int b = 0; b < kLightBlocks; b++)
#pragma unroll
                                for (int k = 0; k < 4; k++)
                                    light_packet(S, always || w[b][k] < thr, dl, maxq, ebw, gap, base, mask_b, dmask_b, cap_b, !prof_skip(D, 4));
                            blk += kLightBlocks;
#pragma unroll
                            for (int b = 0; b < kLightBlocks; b++)
#pragma unroll
                                for (int k = 0; k < 4; k++) w[b][k] = wn[b][k];
                        }
                        budget4 += safe4;   // (a block left over when two go to a body)
                        for (; budget4 && S.t < end; budget4--) {
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if (k > 0 && !(S.t < end)) break;
                                light_packet(S, always || w[0][k] < thr, dl, maxq, ebw, gap, base, mask_b, dmask_b, cap_b, !prof_skip(D, 4));
                            }
                            blk++;
                            light_philox(D, blk, mi, episode, gid, w[0]);
                        }
                        active = S.t < end;
                    }
                    const uint64_t am = __ballot(active);
                    if (prof_on(D) && tl_r1 == 0) tl_r1 = wall_clock64();   // (profile build: when the first round of round_packets ended)
                    if (!am || (uint32_t)__popcll(am) <= D.takeover_lanes) break;
                }
                q = S.q; tu = S.tu; t = S.t;
                a += (S.a4 - a4_0) >> 4; d += (S.d4 - d4_0) >> 4;   // (the packets of one interval: far fewer than 2^28)
            } else {
                for (;;) {
                    if (active) {
                        for (uint32_t budget = D.round_packets; budget && t < end; budget--) {
                            const uint64_t pos = (uint64_t)a + d;
                            double u = 1.0;
                            if ((int64_t)pos >= D.trace_stride) E.flags |= PCC_FLAG_TRACE_OVERRUN;
                            else u = trace[pos];
                            bool dropped;
                            const double2 rec = link_send(t, u < lr, dl, maxq, ebw, q, tu, dropped);
                            const uint32_t off = dropped ? cap_b + ((d << 4) & dmask_b) : ((a << 4) & mask_b);
                            st_rec(reinterpret_cast<double2 *>(base + off), rec);
                            a += dropped ? 0u : 1u;
                            d += dropped ? 1u : 0u;
                            t += gap;
                        }
                        active = t < end;
                    }
                    const uint64_t am = __ballot(active);
                    if (!am || (uint32_t)__popcll(am) <= D.takeover_lanes) break;
                }
            }
            E.nsend[0] = t;
            E.sent[0] = (a - E.ta[0]) + (d - E.td[0]);
            E.ta[0] = a; E.td[0] = d;
        }
        E.q = q; E.tu = tu;
    } else {
        // two senders merged in (time, sender id) order: lane-serial rounds, the tail of the wavefront goes to the
        // two-sender wave path
        const double dl = E.dl, lr = E.lr, maxq = E.maxq, ebw = E.ebw, end = E.end;
        const uint32_t thr = E.thr, episode = E.episode, mi = E.mi, gid = E.gid;
        const bool always = E.always;
        const double *trace = E.trace;
        double q = E.q, tu = E.tu;
        double gap[NS], nsend[NS];
        uint32_t ta[NS], td[NS], sent[NS];
        char *bases[NS];
        uint32_t cap_bs[NS], mask_bs[NS], dmask_bs[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            gap[s] = E.gap[s]; nsend[s] = E.nsend[s]; ta[s] = E.ta[s]; td[s] = E.td[s]; sent[s] = 0;
            bases[s] = E.rings[s].base;
            cap_bs[s] = E.rings[s].cap << 4; mask_bs[s] = (E.rings[s].cap - 1u) << 4; dmask_bs[s] = (2u * E.rings[s].cap - 1u) << 4;
        }
        active = E.run;
        uint32_t blk = 0;  // Philox block = packets of this MI sent on the link / 4
        for (;;) {
            if (active) {
                if (!TRACE) {
                    // lockstep blocks of four packets of the merged stream; the sender of a packet is a select, not a branch, so
                    // lanes with different interleavings stay together.  Like the one-sender rounds (round 6): the packet is
                    // branch-free (light_packet2), the Philox rounds of the NEXT block are in the same basic block, and the blocks
                    // whose packets are certainly before `end` run without the exit test -- sender s has at least
                    // (end - nsend[s]) / gap[s] - 2 more packets before `end`, and the first c0 + c1 packets of the merge are then
                    // all before `end` (they are the c0 + c1 earliest of at least that many).
                    Light2State S;
                    S.q = q; S.tu = tu;
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) {
                        const int s = s2 < NS ? s2 : 0;
                        S.nsend[s2] = nsend[s]; S.a4[s2] = ta[s] << 4; S.d4[s2] = td[s] << 4;
                    }
                    const uint32_t a4_0[2] = {S.a4[0], S.a4[1]}, d4_0[2] = {S.d4[0], S.d4[1]};
                    const double ah0 = (end - S.nsend[0]) / gap[0] - 2.0, ah1 = (end - S.nsend[1]) / gap[NS - 1] - 2.0;
                    const double ahead = (ah0 > 0.0 ? floor(ah0) : 0.0) + (ah1 > 0.0 ? floor(ah1) : 0.0);
                    uint32_t safe4 = ahead >= 4.0 ? (uint32_t)fmin(ahead, (double)D.round_packets) >> 2 : 0u;
                    uint32_t budget4 = D.round_packets / 4 - safe4;
                    uint32_t w[4];
                    light_philox(D, blk, mi, episode, gid, w);
                    for (; safe4; safe4--) {
                        uint32_t wn[4];
                        light_philox(D, blk + 1u, mi, episode, gid, wn);
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            light_packet2(S, always || w[k] < thr, dl, maxq, ebw, gap[0], gap[NS - 1], bases[0], bases[NS - 1], mask_bs[0], mask_bs[NS - 1],
                                          dmask_bs[0], dmask_bs[NS - 1], cap_bs[0], cap_bs[NS - 1]);
                        blk++;
#pragma unroll
                        for (int k = 0; k < 4; k++) w[k] = wn[k];
                    }
                    for (; budget4 && (S.nsend[1] < S.nsend[0] ? S.nsend[1] : S.nsend[0]) < end; budget4--) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (k > 0 && !((S.nsend[1] < S.nsend[0] ? S.nsend[1] : S.nsend[0]) < end)) break;
                            light_packet2(S, always || w[k] < thr, dl, maxq, ebw, gap[0], gap[NS - 1], bases[0], bases[NS - 1], mask_bs[0], mask_bs[NS - 1],
                                          dmask_bs[0], dmask_bs[NS - 1], cap_bs[0], cap_bs[NS - 1]);
                        }
                        blk++;
                        light_philox(D, blk, mi, episode, gid, w);
                    }
                    q = S.q; tu = S.tu;
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) {
                        if (s2 < NS) {
                            nsend[s2 < NS ? s2 : 0] = S.nsend[s2];
                            sent[s2 < NS ? s2 : 0] += ((S.a4[s2] - a4_0[s2]) + (S.d4[s2] - d4_0[s2])) >> 4;   // (far fewer than 2^27 an interval)
                            ta[s2 < NS ? s2 : 0] += (S.a4[s2] - a4_0[s2]) >> 4; td[s2 < NS ? s2 : 0] += (S.d4[s2] - d4_0[s2]) >> 4;
                        }
                    }
                } else {
                    for (uint32_t budget = D.round_packets; budget; budget--) {
                        const bool s1 = nsend[NS - 1] < nsend[0];
                        const double t = s1 ? nsend[NS - 1] : nsend[0];
                        if (!(t < end)) break;
                        uint64_t pos = 0;
#pragma unroll
                        for (int x = 0; x < NS; x++) pos += (uint64_t)ta[x] + td[x];
                        double u = 1.0;
                        if ((int64_t)pos >= D.trace_stride) E.flags |= PCC_FLAG_TRACE_OVERRUN;
                        else u = trace[pos];
                        bool dropped;
                        const double2 rec = link_send(t, u < lr, dl, maxq, ebw, q, tu, dropped);
                        const uint32_t a_s = s1 ? ta[NS - 1] : ta[0], d_s = s1 ? td[NS - 1] : td[0];
                        const uint32_t off = dropped ? (s1 ? cap_bs[NS - 1] : cap_bs[0]) + ((d_s << 4) & (s1 ? dmask_bs[NS - 1] : dmask_bs[0]))
                                                     : ((a_s << 4) & (s1 ? mask_bs[NS - 1] : mask_bs[0]));
                        st_rec(reinterpret_cast<double2 *>((s1 ? bases[NS - 1] : bases[0]) + off), rec);
                        const uint32_t acc = dropped ? 0u : 1u, drp = dropped ? 1u : 0u;
                        if (s1) {
                            ta[NS - 1] += acc; td[NS - 1] += drp; sent[NS - 1]++;
                            nsend[NS - 1] = t + gap[NS - 1];
                        } else {
                            ta[0] += acc; td[0] += drp; sent[0]++;
                            nsend[0] = t + gap[0];
                        }
                    }
                }
                active = (nsend[NS - 1] < nsend[0] ? nsend[NS - 1] : nsend[0]) < end;
            }
            const uint64_t am = __ballot(active);
            if (!am || (uint32_t)__popcll(am) <= D.takeover_lanes) break;
        }
#pragma unroll
        for (int s = 0; s < NS; s++) { E.nsend[s] = nsend[s]; E.ta[s] = ta[s]; E.td[s] = td[s]; E.sent[s] = sent[s]; }
        E.q = q; E.tu = tu;
    }
    // (profile build: word 6 of a light item's record = ticks from its start to the end of its first round)
    if (prof_on(D)) timeline_record<NS>(D, lane, tl_slot, E, tl0, wall_clock64(), 0, lane == 0 && tl_r1 ? tl_r1 - tl0 : 0, 0, 0, 0,
                                        __builtin_readcyclecounter() - cy0);
    if (E.live) store_env<NS>(D, i, E);
    uint32_t pk = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) pk += E.live ? E.sent[s] : 0u;
    for (int o = 32; o; o >>= 1) pk += (uint32_t)__shfl_xor((int)pk, o);
    packets_out = pk;
    return __ballot(active);
}

// An env's state as the wave path needs it, parked in LDS while the wavefront sends the envs of an item one after the other:
// the lanes that loaded the envs (load_env: up to kSlots at a time, their dependent round trips side by side) put them here
// and keep nothing -- round 3 kept the ~35 registers of every lane's env live across heavy_mi, which is what pushed the send
// kernels over their register budget (56-160 bytes of scratch per lane).
constexpr int kSlots = 8;
template <int NS>
struct EnvSlot {
    double dl, lr, maxq, ebw, q, tu, end;
    double gap[NS], nsend[NS];
    const double *trace;
    char *base[NS];
    int64_t i;
    uint32_t cap[NS], ta[NS], td[NS], ha[NS], hd[NS], sent[NS];
    uint32_t thr, episode, mi, gid, flags, bits;  // bits: 1 = always lost, 2 = the interval is not empty, 4 = live
    uint32_t cost0, pad;                          // low word of the clock when the wave path took the env (SndBlk::send_cost): parked here, not in a register
};

__device__ __forceinline__ double uni_f64(double v) {  // a value every lane holds alike -> scalar registers
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t uni_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni_u64(uint64_t v) { return ((uint64_t)uni_u32((uint32_t)(v >> 32)) << 32) | uni_u32((uint32_t)v); }

// A WAVE-PATH item: the envs the lanes hold (lane l: env i, or none), one after the other by all 64 lanes.  Lanes without
// an env stay in: the wave path needs all 64 lanes as workers.  `slots`: this wavefront's kSlots LDS slots.
// W > 1: a TEAM item -- one env (lane 0 of every wavefront names it) sent by the W wavefronts of the workgroup together
// (heavy_mi<.., W>); every wavefront loads the env's state and computes everything alike, wavefront 0 writes.
// Returns the packets the item sent (wave-uniform; the launch statistics of pcc_get_send_split).
// STAGE / stage: the 256-position passes' records leave through 256 LDS slots of this wavefront (heavy_mi, heavy_mi2).
// P: positions per lane of the closed-form passes (heavy_mi; one sender).
template <int NS, bool TRACE, int W, bool STAGE = false, int P = 4>
__device__ __forceinline__ uint32_t send_wave_item(const Dev &D, const uint32_t lane, const int64_t i, const bool in_range,
                                                   const bool fresh, const uint32_t tl_slot, const int warm, const uint32_t warm_mi,
                                                   const void *actions, const int actions_f64, EnvSlot<NS> *slots,
                                                   const uint32_t wv = 0, TeamX *X = nullptr, const bool no_promote = false,
                                                   bool *refused = nullptr, double2 *stage = nullptr) {
    static_assert(W == 1 || NS == 1, "team items are built for one sender");
    const bool writer = W == 1 || wv == 0u;
    const uint64_t tl0 = prof_on(D) ? wall_clock64() : 0;
    uint64_t tl1 = 0, tl_closed = 0, tl_other = 0, tl_env = 0, tl_envs = 0, tl_max = 0, tl_why = 0;
    uint32_t total = 0;
    uint64_t todo = __ballot(in_range);
    while (todo) {
        // ---- the next (up to) kSlots envs: their lanes load them side by side and park them
        const uint32_t rank = (uint32_t)count_below(todo);
        const bool mine = ((todo >> lane) & 1ull) != 0ull && rank < (uint32_t)kSlots;
        const uint64_t chunk = __ballot(mine);
        todo &= ~chunk;
        const uint32_t n_chunk = (uint32_t)__popcll(chunk);
        {
            LaneEnv<NS> E = load_env<NS, TRACE, W>(D, lane, i, mine, fresh, warm, warm_mi, actions, actions_f64, wv, no_promote);
            if (refused) {  // (no_promote: an interval the rings cannot hold was not sent; the flag is the caller's, not the env's)
                *refused = __ballot(mine && (E.flags & PCC_FLAG_INTERNAL)) != 0ull;
                E.flags &= ~(uint32_t)PCC_FLAG_INTERNAL;
            }
            if (mine) {
                EnvSlot<NS> &S = slots[rank];
                S.dl = E.dl; S.lr = E.lr; S.maxq = E.maxq; S.ebw = E.ebw; S.q = E.q; S.tu = E.tu; S.end = E.end;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    S.gap[s] = E.gap[s]; S.nsend[s] = E.nsend[s]; S.base[s] = E.rings[s].base; S.cap[s] = E.rings[s].cap;
                    S.ta[s] = E.ta[s]; S.td[s] = E.td[s]; S.ha[s] = E.ha[s]; S.hd[s] = E.hd[s]; S.sent[s] = E.sent[s];
                }
                S.trace = E.trace; S.i = i;
                S.thr = E.thr; S.episode = E.episode; S.mi = E.mi; S.gid = E.gid; S.flags = E.flags;
                S.bits = (E.always ? 1u : 0u) | (E.run ? 2u : 0u) | (E.live ? 4u : 0u);
            }
        }
        // (what a lane parked is read by every lane of this wavefront: LDS accesses of one wavefront execute in order; the
        // barrier keeps the compiler from moving the reads up)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (prof_on(D) && tl1 == 0) tl1 = wall_clock64();
        for (uint32_t k = 0; k < n_chunk; k++) {
            const EnvSlot<NS> &S = slots[k];
            const uint32_t bits = uni_u32(S.bits);
            if (!(bits & 4u)) continue;  // (a warm-up launch: not an env that is being reset)
            const int64_t ie = (int64_t)uni_u64((uint64_t)S.i);
            uint32_t flags = uni_u32(S.flags), sent_new[NS], ta_new[NS], td_new[NS];
            double q_new = uni_f64(S.q), tu_new = uni_f64(S.tu), nsend_new[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) { sent_new[s] = uni_u32(S.sent[s]); ta_new[s] = uni_u32(S.ta[s]); td_new[s] = uni_u32(S.td[s]); nsend_new[s] = uni_f64(S.nsend[s]); }
            uint32_t before = 0;
#pragma unroll
            for (int s = 0; s < NS; s++) before += sent_new[s];
            if (lane == 0) slots[k].cost0 = (uint32_t)wall_clock64();   // (SndBlk::send_cost; const_cast-free: the slot is this wavefront's)
            if (bits & 2u) {
                if constexpr (NS == 1) {
                    SendState st;
                    if (prof_counters(D) && lane == 0 && writer) atomicAdd(&D.pass_stats[fresh ? 11 : 12], 1ull);
                    st.q = q_new; st.tu = tu_new; st.t = nsend_new[0];
                    st.a = ta_new[0]; st.d = td_new[0]; st.flags = 0;
                    st.prof_closed = 0; st.prof_other = 0;
                    st.sent = sent_new[0];  // packets of this MI the lane rounds already sent
                    heavy_mi<TRACE, W, STAGE, P>(D, lane, wv, X, uni_f64(S.dl), uni_f64(S.lr), uni_u32(S.thr), (bits & 1u) != 0u, uni_f64(S.maxq),
                                              uni_f64(S.ebw), uni_f64(S.gap[0]), uni_f64(S.end), uni_u32(S.episode), uni_u32(S.mi), uni_u32(S.gid),
                                              reinterpret_cast<const double *>(uni_u64(reinterpret_cast<uint64_t>(S.trace))),
                                              reinterpret_cast<char *>(uni_u64(reinterpret_cast<uint64_t>(S.base[0]))), uni_u32(S.cap[0]), st, stage);
                    sent_new[0] += (st.a - ta_new[0]) + (st.d - td_new[0]);
                    q_new = st.q; tu_new = st.tu; nsend_new[0] = st.t; ta_new[0] = st.a; td_new[0] = st.d; flags |= st.flags;
                    if (kProfile) { tl_closed += st.prof_closed; tl_other += st.prof_other; tl_env = (uint64_t)ie; }
                } else {
                    SendState2 st;
                    st.q = q_new; st.tu = tu_new; st.flags = 0;
                    st.prof_closed = 0; st.prof_other = 0; st.prof_why = 0;
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        st.t[s] = nsend_new[s < NS ? s : 0]; st.a[s] = ta_new[s < NS ? s : 0]; st.d[s] = td_new[s < NS ? s : 0];
                        st.sent[s] = sent_new[s < NS ? s : 0];
                    }
                    heavy_mi2<TRACE, STAGE>(D, lane, uni_f64(S.dl), uni_f64(S.lr), uni_u32(S.thr), (bits & 1u) != 0u, uni_f64(S.maxq), uni_f64(S.ebw),
                                     uni_f64(S.gap[0]), uni_f64(S.gap[NS - 1]), uni_f64(S.end), uni_u32(S.episode), uni_u32(S.mi), uni_u32(S.gid),
                                     reinterpret_cast<const double *>(uni_u64(reinterpret_cast<uint64_t>(S.trace))),
                                     reinterpret_cast<char *>(uni_u64(reinterpret_cast<uint64_t>(S.base[0]))),
                                     reinterpret_cast<char *>(uni_u64(reinterpret_cast<uint64_t>(S.base[NS - 1]))),
                                     uni_u32(S.cap[0]), uni_u32(S.cap[NS - 1]), st, stage);
                    q_new = st.q; tu_new = st.tu; flags |= st.flags;
                    if (kProfile) { tl_closed += st.prof_closed; tl_other += st.prof_other; tl_env = (uint64_t)ie; tl_why |= st.prof_why; }
#pragma unroll
                    for (int s = 0; s < NS; s++) { nsend_new[s] = st.t[s]; ta_new[s] = st.a[s]; td_new[s] = st.d[s]; sent_new[s] = st.sent[s]; }
                }
            }
            uint32_t after = 0;
#pragma unroll
            for (int s = 0; s < NS; s++) after += sent_new[s];
            total += after - before;
            if (kProfile) { tl_envs++; tl_max = (after - before) > tl_max ? (after - before) : tl_max; }
            // ---- the env's state back to memory (one lane: every value is wave-uniform)
            if (lane == 0 && writer) {
                D.env[ie].q = q_new; D.env[ie].tu = tu_new;
                if (fresh) {   // (a whole interval by the wave path: what it cost, for the retire half's filing)
                    const uint32_t ticks = (uint32_t)wall_clock64() - S.cost0;   // (mod 2^32: 43 s)
                    D.snd[sidx(D, 0, ie)].send_cost = (ticks < 0xFFFFFFu ? ticks : 0xFFFFFFu) | (S.mi << 24);
                }
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    // never silent: more packets in flight than a ring holds means records were overwritten
                    const uint32_t cap = S.cap[s];
                    if (ta_new[s] - S.ha[s] > cap || td_new[s] - S.hd[s] > 2u * cap) flags |= PCC_FLAG_RING_OVERFLOW;
                    const int64_t ks = sidx(D, s, ie);
                    D.snd[ks].next_send = nsend_new[s];
                    D.snd[ks].ta = ta_new[s]; D.snd[ks].td = td_new[s];
                    D.snd[ks].mi_sent = sent_new[s];
                }
                if (flags) D.env[ie].flags |= flags;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the slots are reused by the next chunk
    }
    if (prof_on(D) && lane == 0 && writer && tl_slot != 0xFFFFFFFFu) {
        uint64_t *w = D.timeline + (int64_t)tl_slot * 8;
        w[0] = tl0; w[1] = tl1; w[2] = wall_clock64(); w[3] = tl_envs | (tl_env << 16); w[4] = total; w[5] = tl_max | (tl_why << 32); w[6] = total;
        w[7] = tl_envs | (tl_closed << 8) | (tl_other << 24);
    }
    return total;
}

}  // namespace
