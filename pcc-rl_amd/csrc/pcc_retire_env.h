// pcc_retire_env.h -- the retire half of one env (retire_env: boundary searches, the MI-ending event, numpy-exact RTT
// means, metrics, history, observation, reward), the event-loop engine of the dormant options, and reset_env.
// Included by the retire kernel, the restart kernel (warm-up intervals of a restart item) and the small-batch kernel.
#pragma once
#include "pcc_dev.h"

namespace {

// ======================================================================================
// retire_kernel: G lanes per env -- 8 for most envs, 16 for the few with long RTT lists.  The half is bound by
// instruction issue, not by memory (1 000 extra VALU instructions per wavefront cost it 18 us of 113,
// profiles/r03_experiments.json): nearly all of retire_env is per-env control flow that a wavefront executes once for
// all its groups, so twice the envs per wavefront is nearly half the instructions per env.  What 16 lanes buy -- the
// whole-list sum and the half sums of an env side by side -- only pays for the envs whose sums are many leaves.
// ======================================================================================
struct Group {
    uint32_t lane;   // 0..G-1 inside the env's group
    uint32_t shift;  // bit position of the group's lane 0 in a wave ballot
};

template <int G>
__device__ __forceinline__ uint32_t gballot(const Group &g, bool p) {
    return (uint32_t)(__ballot(p) >> g.shift) & ((1u << G) - 1u);
}

template <int G>
__device__ __forceinline__ double gbcast(double v, uint32_t src) { return __shfl(v, (int)src, G); }
template <int G>
__device__ __forceinline__ uint32_t gbcast(uint32_t v, uint32_t src) { return (uint32_t)__shfl((int)v, (int)src, G); }
// ... the same with the group's position taken from `g` instead of from the lane id __shfl reads: in the fused step
// (pcc_fused.hip) retire_env sits in a loop, and everything derived from the lane id -- the shuffle addresses of every
// __shfl with a width -- is otherwise hoisted out of it and kept in registers across the whole body (the caller passes a
// lane index the compiler cannot see through)
template <int G>
__device__ __forceinline__ uint32_t gbcast(const Group &g, uint32_t v, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((g.shift + (src & (uint32_t)(G - 1))) << 2), (int)v);
}
template <int G>
__device__ __forceinline__ double gbcast(const Group &g, double v, uint32_t src) {
    const int idx = (int)((g.shift + (src & (uint32_t)(G - 1))) << 2);
    const int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// First index k in [lo, hi) whose record fails `t1 + add < end` (hi if none), by G-ary search:
// every round the G lanes sample the ends of G equal sub-ranges.  Exact for a monotone
// predicate; on the dropped ring the answer can be off inside one near group, which the caller
// repairs (fix_drop_boundary).
template <int G>
__device__ __forceinline__ uint32_t search_boundary(const Group &g, const double2 *ring, uint32_t mask, uint32_t lo,
                                                    uint32_t hi, double add, double end) {
    while (hi - lo > (uint32_t)G) {
        const uint32_t stride = (hi - lo + G - 1) / G;
        uint32_t sidx = lo + (g.lane + 1) * stride;
        if (sidx > hi) sidx = hi;
        sidx -= 1;
        const bool pass = ld_t1(ring + (sidx & mask)) + add < end;
        const uint32_t mfail = ~gballot<G>(g, pass) & ((1u << G) - 1u);
        if (!mfail) return hi;  // the last sample is record hi-1
        const uint32_t f = (uint32_t)__ffs((int)mfail) - 1u;
        const uint32_t s_f = gbcast<G>(g, sidx, f);
        if (f) lo = gbcast<G>(g, sidx, f - 1) + 1;
        hi = s_f;
        if (hi < lo) hi = lo;
    }
    const uint32_t k = lo + g.lane;
    const bool fail = k < hi && !(ld_t1(ring + (k & mask)) + add < end);
    const uint32_t m = gballot<G>(g, fail);
    return m ? lo + (uint32_t)__ffs((int)m) - 1u : hi;
}

// K boundary searches advanced together, so their dependent loads overlap: per round every search
// still running samples its 16 sub-range ends; the last step loads the 16 records
// [lo - 2, lo + 14) around each transition, which also tells whether the records next to the
// transition are "near" (within rounding distance) -- if not, the transition is exact as found and
// ring[b] is already in a register.
struct Bound {
    uint32_t b;      // first index failing `t1 + add < end` (== hi if none)
    bool clean;      // no near-equal neighbours around b-1, b: no event-order repair needed
    double t, lat;   // ring[b] (valid when b < hi0)
};

template <int K, int G>
__device__ __forceinline__ void search_many(const Group &g, const double2 *const (&ring)[K], const uint32_t (&mask)[K],
                                            const uint32_t (&lo0)[K], const uint32_t (&hi0)[K], const double (&add)[K],
                                            double end, const uint32_t (&hint)[K], Bound (&out)[K],
                                            unsigned long long *stat = nullptr /* profile build: hit counters */) {
    static_assert(K == 4 && (G == 16 || G == 8), "four searches per group of 16 or 8 lanes");
    constexpr int R = 16 / G;  // records of a 16-record window per lane
    // The window step (first and last): the group looks at the 16 records base .. base + 15 around [lo, hi], hi - lo <= 12,
    // base = lo - 2 (clamped to the ring's start); lane l holds records base + l (+ 8 with 8 lanes).  It finds the
    // transition inside [lo, hi] and tells whether the records next to it are "near" -- and, from records lo - 1 and hi,
    // whether the transition IS inside: with a good prediction of the boundary (hint: where it would be if this interval
    // retired what the last one did) the whole search is this one round trip, 2-3 lines per ring instead of the 12-16 of
    // a descent from the ring's ends.
    uint32_t lo[K], hi[K];
    bool inside[K], below[K];   // below: the predicted window missed and the transition lies under it
    bool all_inside = true;
    auto window = [&](const bool (&need)[K], const bool last) {
        double2 r[K][R];
        uint32_t base[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            base[k] = lo[k] - lo0[k] >= 2u ? lo[k] - 2u : lo0[k];
#pragma unroll
            for (int h = 0; h < R; h++) {
                const uint32_t idx = base[k] + (uint32_t)(h * G) + g.lane;
                r[k][h].x = 0.0; r[k][h].y = 0.0;
                if (need[k] && idx < hi0[k]) r[k][h] = ld_rec(ring[k] + (idx & mask[k]));
            }
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (!need[k]) continue;  // (the same for all lanes of the group)
            uint32_t mpass = 0, m = 0, mnear = 0;
#pragma unroll
            for (int h = 0; h < R; h++) {
                const uint32_t idx = base[k] + (uint32_t)(h * G) + g.lane;
                const bool in = idx < hi0[k];
                const bool passes = in && (r[k][h].x + add[k] < end);
                mpass |= gballot<G>(g, passes) << (h * G);
                m |= gballot<G>(g, in && idx >= lo[k] && idx < hi[k] && !passes) << (h * G);
                // near flag of record idx: records idx and idx + 1 both exist (in the window) and are within rounding distance
                double tn = gbcast<G>(g, r[k][h].x, (g.lane + 1u) & (uint32_t)(G - 1));
                if (h + 1 < R) {
                    const double tw = gbcast<G>(g, r[k][h + 1 < R ? h + 1 : h].x, 0u);  // the first record of the next row
                    if (g.lane == (uint32_t)G - 1u) tn = tw;
                }
                const bool has_next = (h + 1 < R) || g.lane + 1 < (uint32_t)G;
                const bool nr = in && (idx + 1 < hi0[k]) && has_next && near_time(r[k][h].x, tn);
                mnear |= gballot<G>(g, nr) << (h * G);
            }
            const uint32_t b = m ? base[k] + (uint32_t)__ffs((int)m) - 1u : hi[k];
            // the transition lies in [lo, hi] iff record lo - 1 passes and record hi fails (where they exist)
            const bool lo_ok = lo[k] == lo0[k] || ((mpass >> (lo[k] - 1u - base[k])) & 1u);
            const bool hi_ok = hi[k] == hi0[k] || !((mpass >> (hi[k] - base[k])) & 1u);
            inside[k] = last || (lo_ok && hi_ok);  // (after the descent the window holds the transition by construction)
            below[k] = !lo_ok;
            if (!inside[k]) {  // the descent goes on in the part of the ring the window points to
                if (!lo_ok) { hi[k] = lo[k] - 1u; lo[k] = lo0[k]; }
                else { lo[k] = hi[k] + 1u; hi[k] = hi0[k]; }
                continue;
            }
            // pairs that matter: (b-2,b-1), (b-1,b), (b,b+1) -> window positions (b-2-base), (b-1-base), (b-base)
            uint32_t want = 0;
            for (int d = 0; d < 3; d++) {
                const int l = (int)(b - base[k]) - 2 + d;
                if (l >= 0 && l < 16) want |= 1u << l;
            }
            out[k].b = b;
            out[k].clean = (mnear & want) == 0u;
            const uint32_t lb = b - base[k] < 16u ? b - base[k] : 0u;  // (the group's own value)
            double bx = r[k][0].x, by = r[k][0].y;
            if (R > 1 && lb >= (uint32_t)G) { bx = r[k][R - 1].x; by = r[k][R - 1].y; }
            out[k].t = gbcast<G>(g, bx, lb & (G - 1));
            out[k].lat = gbcast<G>(g, by, lb & (G - 1));
        }
    };
    // ---- 1. the predicted windows
    bool need[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        uint32_t h = hint[k] < lo0[k] ? lo0[k] : (hint[k] > hi0[k] ? hi0[k] : hint[k]);
        lo[k] = h - lo0[k] >= 5u ? h - 5u : lo0[k];
        hi[k] = hi0[k] - lo[k] > 12u ? lo[k] + 12u : hi0[k];
        need[k] = true;
    }
    window(need, false);
#pragma unroll
    for (int k = 0; k < K; k++) all_inside = all_inside && inside[k];
    if (kProfile && stat) {  // searches, searches whose predicted window held the boundary; wavefronts, wavefronts without a descent
        if (g.lane == 0) {
            atomicAdd(&stat[0], (unsigned long long)K);
            atomicAdd(&stat[1], (unsigned long long)((inside[0] ? 1 : 0) + (inside[1] ? 1 : 0) + (inside[2] ? 1 : 0) + (inside[3] ? 1 : 0)));
        }
        const uint64_t act = __ballot(true), hit = __ballot(all_inside);
        if ((threadIdx.x & (kWave - 1)) == (uint32_t)__ffsll((unsigned long long)act) - 1u) {
            atomicAdd(&stat[2], 1ull);
            if (act == hit) atomicAdd(&stat[3], 1ull);
        }
    }
    if (all_inside) return;
    // ---- 2. narrowing rounds for the searches whose window missed: search q belongs to G/4 lanes, which sample the
    // ends of 4 equal sub-ranges (two each with 8 lanes) -- all four searches in the same instructions, 16 scattered
    // lines per round instead of 64, at the price of one or two more rounds than a 16-way split would need.
    constexpr int LQ = G / 4;       // lanes per search
    constexpr int PL = 4 / LQ;      // probes per lane
    const uint32_t q = g.lane / LQ, j = g.lane % LQ;
    uint32_t lo_m = q == 0 ? lo[0] : q == 1 ? lo[1] : q == 2 ? lo[2] : lo[3];
    uint32_t hi_m = q == 0 ? hi[0] : q == 1 ? hi[1] : q == 2 ? hi[2] : hi[3];
    const bool done_m = q == 0 ? inside[0] : q == 1 ? inside[1] : q == 2 ? inside[2] : inside[3];
    const double2 *ring_m = q == 0 ? ring[0] : q == 1 ? ring[1] : q == 2 ? ring[2] : ring[3];
    const uint32_t mask_m = q == 0 ? mask[0] : q == 1 ? mask[1] : q == 2 ? mask[2] : mask[3];
    const double add_m = q == 0 ? add[0] : q == 1 ? add[1] : q == 2 ? add[2] : add[3];
    const bool below_m = q == 0 ? below[0] : q == 1 ? below[1] : q == 2 ? below[2] : below[3];
    // The first round gallops away from the predicted window instead of cutting the whole remaining range in four: a
    // prediction that missed has mostly missed by little, and the transition then sits within a few dozen records of the
    // window's edge -- probes at distances 12, 48, 192, 768 from that edge bracket it in one round (range <= 12, 36, 144, ...)
    // where four equal parts of a ring holding hundreds of records need two or three.  Any ascending probes narrow a
    // bracket of a monotone predicate exactly.
    bool first = true;
    for (;;) {
        const bool active = !done_m && hi_m - lo_m > 12u;
        if (!gballot<G>(g, active)) break;
        const uint32_t stride = (hi_m - lo_m + 3u) / 4u;
        uint32_t sidx[PL];
        uint32_t passbits = 0;  // bit p: probe p of my search passes (probes j * PL + e of lane j)
#pragma unroll
        for (int e = 0; e < PL; e++) {
            const uint32_t pr = j * PL + (uint32_t)e;   // probe 0..3, ascending record indices
            uint32_t x = lo_m + (pr + 1u) * stride;
            if (x > hi_m) x = hi_m;
            if (first) {
                if (below_m) {   // from the top of the range down: records hi - 768, hi - 192, hi - 48, hi - 12
                    const uint32_t dist = 12u << (2u * (3u - pr));
                    x = hi_m - lo_m > dist ? hi_m - dist + 1u : lo_m + 1u;
                } else {         // from the bottom up: records lo + 11, lo + 47, lo + 191, lo + 767
                    const uint32_t dist = 12u << (2u * pr);
                    x = hi_m - lo_m > dist ? lo_m + dist : hi_m;
                }
            }
            sidx[e] = x - 1u;
        }
        first = false;
        double tsamp[PL];
#pragma unroll
        for (int e = 0; e < PL; e++) {
            tsamp[e] = 0.0;
            if (active) tsamp[e] = ld_t1(ring_m + (sidx[e] & mask_m));
        }
#pragma unroll
        for (int e = 0; e < PL; e++) {
            const uint32_t bm = gballot<G>(g, tsamp[e] + add_m < end) >> (LQ * q);  // my search's lanes
#pragma unroll
            for (int l = 0; l < LQ; l++) passbits |= ((bm >> l) & 1u) << (l * PL + e);
        }
        const uint32_t mfail = ~passbits & 0xFu;
        // samples of my search's failing probe f and of the probe before it (every lane shuffles)
        const uint32_t f = mfail ? (uint32_t)__ffs((int)mfail) - 1u : 0u;
        const uint32_t fp = f ? f - 1u : 0u;
        uint32_t mine_f = sidx[0], mine_p = sidx[0];
        if (PL > 1) { mine_f = (f % PL) ? sidx[PL - 1] : sidx[0]; mine_p = (fp % PL) ? sidx[PL - 1] : sidx[0]; }
        const uint32_t s_f = gbcast<G>(g, mine_f, LQ * q + f / PL);
        const uint32_t s_p = gbcast<G>(g, mine_p, LQ * q + fp / PL);
        const uint32_t s_last = gbcast<G>(g, sidx[PL - 1], LQ * q + LQ - 1);   // the highest probe of my search
        if (active) {
            if (!mfail) {
                lo_m = s_last + 1u;  // every probe passes: the transition is above the last one (a cut in four ends at record hi - 1)
            } else {
                if (f) lo_m = s_p + 1u;
                hi_m = s_f < lo_m ? lo_m : s_f;
            }
        }
    }
    // ---- 3. the window around each of those transitions
#pragma unroll
    for (int k = 0; k < K; k++) {
        need[k] = !inside[k];
        if (need[k]) { lo[k] = gbcast<G>(g, lo_m, LQ * k); hi[k] = gbcast<G>(g, hi_m, LQ * k); }
    }
    window(need, true);
}

// ---- serial paths on the dropped ring (one lane) -----------------------------------------

// move ring[k] in front of ring[p] (p <= k), keeping the order of the records in between
__device__ __forceinline__ void rotate_to_front(double2 *ring, uint32_t mask, uint32_t p, uint32_t k) {
    const double2 r = ld_rec(ring + (k & mask));
    for (uint32_t m = k; m > p; m--) st_rec(ring + (m & mask), ld_rec(ring + ((m - 1) & mask)));
    st_rec(ring + (p & mask), r);
}

// Repairs around a search transition b on the dropped ring, where records may be out of event order: b-1 and b themselves
// plus everything chained to them by near-equal times -- the near window [g0, g1), h <= g0 <= b <= g1 <= tail.
//   fix_drop_boundary: the exact retire boundary.  On return records [h, p) are exactly those with t1 + dl < end (members of
//     the window that pass are moved in front, the rest keep their order); also the best hop-2 candidate (smallest
//     (t2, lat2) key) among the window's unretired records that are already past the forward hop (t1 < end).
//   drop_hop1_candidate: smallest (t1, lat) among the window's records still on the forward hop (t1 >= end); c = the search
//     transition for `t1 < end`.
// (results by value: reference out-parameters of an out-of-line function live in scratch memory)
struct DropFix { uint32_t p, cand_idx; double cand_t, cand_lat; };
struct Cand { double t, lat; };

// ---- by the whole group: G records per memory round trip -------------------------------------------------------------
// Rounds 1-3 walked a near group record by record on the lead lane, a dependent load each; the envs that overdrive their links --
// the largest of a launch, the ones it waits for -- drop packets in runs, and their near groups are 5-20 records: 9-22 us of
// a wide workgroup's 85-90 (tools/retire_timeline.py, profiles/r04_experiments.json).  Here lane l of the group holds record
// l of the window: the chain of near pairs is a ballot, the stable partition of fix_drop_boundary a pair of prefix counts,
// the candidates a min-reduction.  Windows of more than G records (rare) are walked by the lead lane, record by record.
// Every lane returns the same values.
template <int G>
__device__ __forceinline__ void near_window_g(const Group &g, const double2 *ring, uint32_t mask, uint32_t h, uint32_t tail,
                                              uint32_t b, uint32_t &g0, uint32_t &g1) {
    g0 = b;
    g1 = b;
    if (b > h) {
        g0 = b - 1;
        for (;;) {
            // lane l: record g0 - l; pair l = (record g0 - l - 1, record g0 - l), tested as near_time(earlier, later)
            const bool has = g0 - h >= g.lane;
            const double t = has ? ld_t1(ring + ((g0 - g.lane) & mask)) : 0.0;
            const double tp = gbcast<G>(g, t, (g.lane + 1u) & (uint32_t)(G - 1));
            const bool nr = g.lane + 1u < (uint32_t)G && g0 - h >= g.lane + 1u && near_time(tp, t);
            const uint32_t run = (uint32_t)__ffs((int)~gballot<G>(g, nr)) - 1u;  // near pairs from pair 0 on (pair G-1 never is)
            g0 -= run;
            if (run < (uint32_t)G - 1u) break;
        }
    }
    if (b < tail) {
        g1 = b + 1;
        for (;;) {
            // lane l: record g1 - 1 + l; pair l = (record g1 - 1 + l, record g1 + l), tested as near_time(later, earlier)
            const bool has = g1 - 1u + g.lane < tail;
            const double t = has ? ld_t1(ring + ((g1 - 1u + g.lane) & mask)) : 0.0;
            const double tn = gbcast<G>(g, t, (g.lane + 1u) & (uint32_t)(G - 1));
            const bool nr = g.lane + 1u < (uint32_t)G && g1 + g.lane < tail && near_time(tn, t);
            const uint32_t run = (uint32_t)__ffs((int)~gballot<G>(g, nr)) - 1u;
            g1 += run;
            if (run < (uint32_t)G - 1u) break;
        }
    }
}

// the serial tail of fix_drop_boundary over a window that is already known
__device__ __forceinline__ DropFix fix_drop_window(double2 *ring, uint32_t mask, uint32_t g0, uint32_t g1, double dl, double end) {
    uint32_t p = g0;
    for (uint32_t k = g0; k < g1; k++) {
        const double2 r = ld_rec(ring + (k & mask));
        if (r.x + dl < end) {
            if (k != p) rotate_to_front(ring, mask, p, k);
            p++;
        }
    }
    DropFix out;
    out.p = p; out.cand_idx = 0xFFFFFFFFu; out.cand_t = INFINITY; out.cand_lat = 0.0;
    for (uint32_t k = p; k < g1; k++) {
        const double2 r = ld_rec(ring + (k & mask));
        if (r.x < end) {
            const double t2 = r.x + dl, l2 = r.y + dl;
            if (out.cand_idx == 0xFFFFFFFFu || t2 < out.cand_t || (t2 == out.cand_t && l2 < out.cand_lat)) {
                out.cand_idx = k; out.cand_t = t2; out.cand_lat = l2;
            }
        }
    }
    return out;
}

// smallest (t, lat, idx) over the group's lanes that have one (idx 0xFFFFFFFF: none); every lane gets the result
template <int G>
__device__ __forceinline__ void gmin_key(double &t, double &lat, uint32_t &idx) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) {
        const double ot = __shfl_xor(t, o, G), ol = __shfl_xor(lat, o, G);
        const uint32_t oi = (uint32_t)__shfl_xor((int)idx, o, G);
        const bool take = oi != 0xFFFFFFFFu && (idx == 0xFFFFFFFFu || ot < t || (ot == t && (ol < lat || (ol == lat && oi < idx))));
        if (take) { t = ot; lat = ol; idx = oi; }
    }
}

// fix_drop_boundary by the group (all its lanes call; the caller fences before anybody reads the ring again)
template <int G>
__device__ __noinline__ DropFix fix_drop_boundary_g(uint32_t g_lane, uint32_t g_shift, double2 *ring, uint32_t mask, uint32_t h,
                                                    uint32_t tail, uint32_t b, double dl, double end) {
    Group g;   // (scalars in, not the struct: a struct argument of an out-of-line function goes through scratch memory)
    g.lane = g_lane; g.shift = g_shift;
    uint32_t g0, g1;
    near_window_g<G>(g, ring, mask, h, tail, b, g0, g1);
    const uint32_t n = g1 - g0;
    DropFix out;
    if (n > (uint32_t)G) {
        if (g.lane == 0) out = fix_drop_window(ring, mask, g0, g1, dl, end);
        out.p = gbcast<G>(out.p, 0); out.cand_idx = gbcast<G>(out.cand_idx, 0);
        out.cand_t = gbcast<G>(out.cand_t, 0); out.cand_lat = gbcast<G>(out.cand_lat, 0);
        return out;
    }
    const bool in = g.lane < n;
    double2 r;
    r.x = 0.0; r.y = 0.0;
    if (in) r = ld_rec(ring + ((g0 + g.lane) & mask));
    const bool pass = in && (r.x + dl < end);
    const uint32_t below = (1u << g.lane) - 1u;
    const uint32_t mp = gballot<G>(g, pass), mf = gballot<G>(g, in && !pass);
    const uint32_t npass = (uint32_t)__popc(mp);
    // stable partition: the passing records to the front in their order, the others behind them in theirs (what moving each
    // passing record in front of the first one that does not pass, one by one, comes to)
    const uint32_t newpos = pass ? (uint32_t)__popc(mp & below) : npass + (uint32_t)__popc(mf & below);
    if (in && newpos != g.lane) st_rec(ring + ((g0 + newpos) & mask), r);
    out.p = g0 + npass;
    // best hop-2 candidate among the unretired records of the window that are past the forward hop; ties: the first in ring order
    const bool elig = in && !pass && r.x < end;
    double ct = elig ? r.x + dl : INFINITY, cl = elig ? r.y + dl : 0.0;
    uint32_t ci = elig ? g0 + newpos : 0xFFFFFFFFu;
    gmin_key<G>(ct, cl, ci);
    out.cand_idx = ci;
    out.cand_t = ci != 0xFFFFFFFFu ? ct : INFINITY;
    out.cand_lat = ci != 0xFFFFFFFFu ? cl : 0.0;
    return out;
}

// drop_hop1_candidate by the group
template <int G>
__device__ __noinline__ Cand drop_hop1_candidate_g(uint32_t g_lane, uint32_t g_shift, const double2 *ring, uint32_t mask, uint32_t h,
                                                   uint32_t tail, uint32_t c, double end) {
    Group g;
    g.lane = g_lane; g.shift = g_shift;
    uint32_t g0, g1;
    near_window_g<G>(g, ring, mask, h, tail, c, g0, g1);
    const uint32_t n = g1 - g0;
    Cand out;
    if (n > (uint32_t)G) {
        out.t = INFINITY; out.lat = 0.0;
        if (g.lane == 0) {
            for (uint32_t k = g0; k < g1; k++) {
                const double2 r = ld_rec(ring + (k & mask));
                if (!(r.x < end) && (r.x < out.t || (r.x == out.t && r.y < out.lat))) { out.t = r.x; out.lat = r.y; }
            }
        }
        out.t = gbcast<G>(out.t, 0); out.lat = gbcast<G>(out.lat, 0);
        return out;
    }
    const bool in = g.lane < n;
    double2 r;
    r.x = 0.0; r.y = 0.0;
    if (in) r = ld_rec(ring + ((g0 + g.lane) & mask));
    const bool elig = in && !(r.x < end);
    double ct = elig ? r.x : INFINITY, cl = elig ? r.y : 0.0;
    uint32_t ci = elig ? g.lane : 0xFFFFFFFFu;
    gmin_key<G>(ct, cl, ci);
    out.t = ci != 0xFFFFFFFFu ? ct : INFINITY;
    out.lat = ci != 0xFFFFFFFFu ? cl : 0.0;
    return out;
}

// --------------------------------------------------------------------------------------
// numpy-exact np.mean pieces.  np.add.reduce splits the samples into 8192-element chunks summed
// left to right; each chunk is DOUBLE_pairwise_sum: split n -> (n/2 rounded down to a multiple of
// 8, rest) until <= 128; a leaf keeps 8 strided accumulators r[j] += a[8b + j], folds them
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and adds the < 8 leftover samples one by one.
// The RTT samples of an MI are a contiguous slice of the accepted ring (sample = lat0 + dl), so a
// leaf is random access: an 8-lane subgroup loads its <= 16 strided samples per lane in one
// round trip (lane j owns r[j]) and folds with __shfl_xor.
// --------------------------------------------------------------------------------------
struct LeafPair { double a, b; };

// x + (x of lane ^ 1), x + (x of lane ^ 2), x + (x of the mirrored lane of the 8-lane half row) as
// DPP moves: one VALU-class operation each instead of a trip through the LDS crossbar.  Adds are
// commutative, so the mirrored partner (lane 7 - j, which holds the other quad's sum) gives the
// same ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) in every lane.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(phi, plo);
}
__device__ __forceinline__ double fold8(double x) {
    x = x + dpp_f64<0xB1>(x);   // quad_perm:[1,0,3,2]
    x = x + dpp_f64<0x4E>(x);   // quad_perm:[2,3,0,1]
    x = x + dpp_f64<0x141>(x);  // row_half_mirror
    return x;
}

// Two leaves per call, one memory round trip: leaf A = [begA, begA + lenA) with lenA <= 128, and,
// when lenA < 72 (at most 8 full blocks), leaf B = [begB, begB + lenB) with lenB < 72 in the
// upper eight load slots that a long leaf A would use itself.  (lenB must be 0 when lenA >= 72.)
// Lane j of the 8-lane subgroup owns accumulator r[j] and loads only the samples it adds (the L1
// request rate, not the bytes, is what the sums are bound by).  The < 8 leftover samples sit one
// per lane and are added, in order, in the subgroup's lane 0: ONLY LANE 0 of each subgroup returns
// the leaf sums, the other lanes return garbage.
__device__ __noinline__ LeafPair leaf_sum2(const double2 *ring, uint32_t mask, uint32_t begA, uint32_t lenA,
                                              uint32_t begB, uint32_t lenB, double dl, uint32_t sl) {
    const uint32_t nblkA = lenA >> 3, nblkB = lenB >> 3;  // full blocks of 8 ...
    const uint32_t ntA = lenA & 7u, ntB = lenB & 7u;      // ... and < 8 leftover samples, added one by one at the end
    const char *base = reinterpret_cast<const char *>(ring) + 8;  // .y of record 0
    const uint32_t bmask = mask << 4;
    const uint32_t oA = ((begA + sl) << 4) & bmask;
    const bool wideA = nblkA > 8;
    const uint32_t o2 = wideA ? oA + 1024u : ((begB + sl) << 4);  // upper bank: blocks 8.. of A, or B
    const uint32_t n2 = wideA ? nblkA - 8u : nblkB;
    // Slots without a sample hold -dl: (-dl) + dl is exactly +0.0 and x + 0.0 == x, so every add
    // below is unconditional -- no compares, no selects -- and still numpy's value bit for bit.
    const double none = -dl;
    double v[16], tvA = none, tvB = none;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        v[b] = none;
        if ((uint32_t)b < nblkA) v[b] = ld_f64(base + ((oA + 128u * b) & bmask));
    }
    if (sl < ntA) tvA = ld_f64(base + ((((begA + 8u * nblkA + sl) << 4)) & bmask));
#pragma unroll
    for (int b = 0; b < 8; b++) {
        v[8 + b] = none;
        if ((uint32_t)b < n2) v[8 + b] = ld_f64(base + ((o2 + 128u * b) & bmask));
    }
    if (sl < ntB) tvB = ld_f64(base + ((((begB + 8u * nblkB + sl) << 4)) & bmask));
    double ra = v[0] + dl, rb = 0.;
#pragma unroll
    for (int b = 1; b < 8; b++) ra += v[b] + dl;
    if (wideA) {
#pragma unroll
        for (int b = 8; b < 16; b++) ra += v[b] + dl;
    } else {
        rb = v[8] + dl;
#pragma unroll
        for (int b = 1; b < 8; b++) rb += v[8 + b] + dl;
    }
    ra = fold8(ra);
    rb = fold8(rb);
    // leftover sample e comes to lane 0 (and 8) of the row by a DPP shift; the moves are independent
    double ta[7], tb[7];
    ta[0] = tvA; tb[0] = tvB;
    ta[1] = dpp_f64<0x101>(tvA); tb[1] = dpp_f64<0x101>(tvB);
    ta[2] = dpp_f64<0x102>(tvA); tb[2] = dpp_f64<0x102>(tvB);
    ta[3] = dpp_f64<0x103>(tvA); tb[3] = dpp_f64<0x103>(tvB);
    ta[4] = dpp_f64<0x104>(tvA); tb[4] = dpp_f64<0x104>(tvB);
    ta[5] = dpp_f64<0x105>(tvA); tb[5] = dpp_f64<0x105>(tvB);
    ta[6] = dpp_f64<0x106>(tvA); tb[6] = dpp_f64<0x106>(tvB);
#pragma unroll
    for (int e = 0; e < 7; e++) {
        ra += ta[e] + dl;
        rb += tb[e] + dl;
    }
    LeafPair out;
    out.a = ra;
    out.b = rb;
    return out;
}

// np.add.reduce over ring[beg, beg + n) as a resumable walk: next() names the next leaf, feed()
// takes its sum.  8192-sample chunks left to right; inside a chunk DOUBLE_pairwise_sum's
// recursion (split n -> n/2 rounded down to a multiple of 8 | rest, until <= 128) walked left to
// right with an explicit stack (depth <= 6).
// The stack lives in LDS, a row per 8-lane subgroup (the lanes of a subgroup walk the same list: they write the same sizes;
// the sums are lane 0's -- the leaf code returns them there -- and only that lane writes them).  In registers every access was
// a chain of selects over the 7 levels (a dynamically indexed array goes to scratch memory): ~300 of the ~650 instructions
// between two leaves of a long list, and the retire launch is bound by what its wavefronts ISSUE -- its longest workgroups
// are 16 such leaves one after the other (profiles/r05_latency_ring.json, r05_rtt_prefetch.json).
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
constexpr int kWalkDepth = 7;          // 8192 -> 4096 -> ... -> 128
constexpr int kWalkRowWords = 24;      // words 0..6: the right parts' sizes; words 8..21: the left parts' sums
constexpr int kWalkMaxBlock = 4 * kWave;
struct NpSumWalk {
    lds_u32 *right_n;
    lds_f64 *left_sum;
    uint32_t have_left;
    int sp;
    uint32_t cur, pos, left_in_job;
    double tot;
    bool done, writer;

    __device__ __forceinline__ void bind(lds_u32 *row, bool lane0) {
        right_n = row;
        left_sum = (lds_f64 *)(row + 8);
        writer = lane0;
    }
    __device__ __forceinline__ void start(uint32_t beg, uint32_t n) {
        pos = beg; left_in_job = n; tot = 0.; sp = 0; have_left = 0; done = n == 0;
        cur = n < kNpBufsize ? n : kNpBufsize;
    }
    __device__ __forceinline__ bool single_leaf() const { return sp == 0 && cur == left_in_job && cur <= 128; }
    // the next leaf: [leaf_beg, leaf_beg + leaf_len)
    __device__ __forceinline__ void next(uint32_t &leaf_beg, uint32_t &leaf_len) {
        while (cur > 128) {
            uint32_t n2 = cur / 2;
            n2 -= n2 % 8;
            right_n[sp] = cur - n2;
            have_left &= ~(1u << sp);
            sp++;
            cur = n2;
        }
        leaf_beg = pos;
        leaf_len = cur;
    }
    __device__ __forceinline__ void feed(double val) {
        pos += cur;
        left_in_job -= cur;
        while (sp > 0) {
            const int top = sp - 1;
            if (!(have_left & (1u << top))) {
                if (writer) left_sum[top] = val;
                cur = right_n[top];
                have_left |= 1u << top;
                return;  // descend into the right part
            }
            val = left_sum[top] + val;
            sp--;
        }
        tot += val;  // one chunk finished (0.0 + x == x for the first)
        cur = left_in_job < kNpBufsize ? left_in_job : kNpBufsize;
        done = left_in_job == 0;
    }
};

// Means over the RTTs (= forward latency + dl) of the n > 0 acknowledged packets
// ring[from, from + n) of the accepted ring: the whole list (so:119-122) and, when asked,
// mean(second half) - mean(first half) (so:138-142).  With 16 lanes per env lanes 0-7 walk the whole list while lanes
// 8-15 walk the two halves; with 8 lanes the three lists are walked one after the other.  Every 8-lane subgroup of the
// wavefront walks its own list(s) but all of them call the leaf code together, one memory round trip per call: with 16
// lanes n <= 128 -- the usual case -- is a single call (whole list | both halves), with 8 lanes two.
template <int G>
__device__ __forceinline__ void rtt_means(const Group &g, const double2 *ring, uint32_t mask, uint32_t from,
                                          uint32_t n, double dl, bool need_halves, double &mean_all,
                                          double &lat_inc) {
    static_assert(G == 16 || G == 8, "one or two 8-lane subgroups per env");
    const uint32_t sub = g.lane >> 3, sl = g.lane & 7u;
    const uint32_t half = n / 2;
    const bool halves = need_halves && half >= 1;
    // jobs of this subgroup, in order; 16 lanes: {whole} | {first half, second half}; 8 lanes: {whole, first half, second half}
    constexpr int kJobs = G == 16 ? 2 : 3;
    uint32_t jb0, jn0, jb1, jn1, jb2, jn2;
    if (G == 16) {
        jb0 = from; jn0 = sub == 0 ? n : (halves ? half : 0u);
        jb1 = from + half; jn1 = sub == 0 ? 0u : (halves ? n - half : 0u);
        jb2 = from; jn2 = 0u;
    } else {
        jb0 = from; jn0 = n;
        jb1 = from; jn1 = halves ? half : 0u;
        jb2 = from + half; jn2 = halves ? n - half : 0u;
    }
    // (the job that may take its successor along in one call: both halves, when they are single short leaves)
    constexpr int kPairJob = G == 16 ? 0 : 1;
    const uint32_t pair_beg = kPairJob == 0 ? jb1 : jb2, pair_n = kPairJob == 0 ? jn1 : jn2;
    double res0 = 0.0, res1 = 0.0, res2 = 0.0;
    // (rows are read and written as 8-byte words from word 8 on: the array is aligned for that by declaration, not by luck)
    static_assert(kWalkRowWords % 2 == 0, "a row's sums start on an 8-byte boundary");
    __shared__ __attribute__((aligned(16))) uint32_t s_walk[kWalkMaxBlock / 8 * kWalkRowWords];
    NpSumWalk w;
    w.bind((lds_u32 *)s_walk + (threadIdx.x >> 3) * kWalkRowWords, sl == 0);
    int job = 0;
    // the first job at or after `j` that has samples (kJobs: none); starts the walk over it
    auto start_from = [&](int j) {
        if (j == 0 && jn0 == 0u) j = 1;
        if (j == 1 && jn1 == 0u) j = 2;
        if (j == 2 && (kJobs < 3 || jn2 == 0u)) j = kJobs;
        job = j;
        if (j == 0) w.start(jb0, jn0);
        else if (j == 1) w.start(jb1, jn1);
        else if (j == 2 && kJobs == 3) w.start(jb2, jn2);
    };
    start_from(0);
    for (;;) {
        const bool active = job < kJobs;
        if (!__ballot(active)) break;
        uint32_t begA = from, lenA = 0, begB = from, lenB = 0;
        bool pair = false;
        if (active) {
            w.next(begA, lenA);
            pair = job == kPairJob && w.single_leaf() && lenA < 72 && pair_n != 0 && pair_n < 72;
            if (pair) { begB = pair_beg; lenB = pair_n; }
        }
        const LeafPair p = leaf_sum2(ring, mask, begA, lenA, begB, lenB, dl, sl);
        if (active) {
            if (pair) {
                if (kPairJob == 0) { res0 = p.a; res1 = p.b; } else { res1 = p.a; res2 = p.b; }
                job = kJobs;
            } else {
                w.feed(p.a);
                if (w.done) {
                    if (job == 0) res0 = w.tot; else if (job == 1) res1 = w.tot; else res2 = w.tot;
                    start_from(job + 1);
                }
            }
        }
    }
    if (G == 16) {
        mean_all = gbcast<G>(g, res0, 0u) / (double)n;
        lat_inc = halves ? gbcast<G>(g, res1, 8u) / (double)(n - half) - gbcast<G>(g, res0, 8u) / (double)half : 0.0;
    } else {
        mean_all = gbcast<G>(g, res0, 0u) / (double)n;
        lat_inc = halves ? gbcast<G>(g, res2, 0u) / (double)(n - half) - gbcast<G>(g, res1, 0u) / (double)half : 0.0;
    }
}

// the 12 metrics of one MI (so:110-191) from its counts and RTT means
__device__ __forceinline__ void mi_metrics(uint32_t sent, uint32_t acked, uint32_t lost, double dur, double lat,
                                           double inc, double &min_lat, double (&m)[PCC_N_METRICS]) {
    const int64_t bs = (int64_t)sent * kBytesPerPacket, ba = (int64_t)acked * kBytesPerPacket,
                  bl = (int64_t)lost * kBytesPerPacket;
    m[PCC_M_RECV_DUR] = dur;
    m[PCC_M_SEND_DUR] = dur;
    m[PCC_M_SEND_RATE] = dur > 0.0 ? 8.0 * (double)bs / dur : 0.0;
    m[PCC_M_RECV_RATE] = dur > 0.0 ? 8.0 * (double)(ba - kBytesPerPacket) / dur : 0.0;
    m[PCC_M_AVG_LATENCY] = lat;
    m[PCC_M_LOSS_RATIO] = (bl + ba > 0) ? (double)bl / (double)(bl + ba) : 0.0;
    m[PCC_M_LATENCY_INCREASE] = inc;
    m[PCC_M_ACK_LATENCY_INFLATION] = dur > 0.0 ? inc / dur : 0.0;
    m[PCC_M_SENT_LATENCY_INFLATION] = dur > 0.0 ? inc / dur : 0.0;
    double cm;  // so:158-176; min_lat == 0.0 <=> no entry for this sender yet
    if (min_lat > 0.0) {
        if (lat == 0.0) cm = min_lat;
        else if (lat < min_lat) { cm = lat; min_lat = lat; }
        else cm = min_lat;
    } else {
        if (lat > 0.0) { cm = lat; min_lat = lat; }
        else cm = 0.0;
    }
    m[PCC_M_CONN_MIN_LATENCY] = cm;
    m[PCC_M_SEND_RATIO] = (m[PCC_M_RECV_RATE] > 0.0 && m[PCC_M_SEND_RATE] < 1000.0 * m[PCC_M_RECV_RATE])
                              ? m[PCC_M_SEND_RATE] / m[PCC_M_RECV_RATE] : 1.0;
    m[PCC_M_LATENCY_RATIO] = cm > 0.0 ? lat / cm : 1.0;
}

// m[id] for a per-lane id without an indexed (= scratch memory) array: OR of masked bit patterns
__device__ __forceinline__ double select_metric(const double (&m)[PCC_N_METRICS], int id) {
    unsigned long long bits = 0ull;
#pragma unroll
    for (int k = 0; k < PCC_N_METRICS; k++) bits |= (id == k) ? (unsigned long long)__double_as_longlong(m[k]) : 0ull;
    return __longlong_as_double((long long)bits);
}

// --------------------------------------------------------------------------------------
// USE_LATENCY_NOISE (ns:51-52, 150-151, 171-172): every link latency is multiplied by
// random.uniform(1.0, MAX_LATENCY_NOISE), one more draw of the stream per hop.  Packets overtake each
// other on both hops, so the two monotone rings cannot hold the in-flight set: with this option an
// env keeps the reference's own structure, a heap of its events, in global memory, and ONE
// lane runs the reference's event loop (ns:127-178) over it -- exactness, not speed, is the point of
// a dormant option.  Only acknowledgement events live in the heap (hop 1: arrives at the return link,
// hop 2: arrives at the sender); the sender's one pending SEND is next_send as everywhere else.  The
// reference orders events as tuples (time, sender, type, hop, latency, dropped): 'A' < 'S' puts an
// ACK before a SEND at equal times, the rest is heap_less.  Any priority queue pops the same order.
// --------------------------------------------------------------------------------------
__device__ __forceinline__ bool sign_of(double x) { return __double_as_longlong(x) < 0; }
__device__ __forceinline__ bool heap_less(const double2 a, const double2 b) {
    const double ta = fabs(a.x), tb = fabs(b.x);
    if (ta != tb) return ta < tb;
    const bool ha = sign_of(a.x), hb = sign_of(b.x);  // hop 2
    if (ha != hb) return hb;
    const double la = fabs(a.y), lb = fabs(b.y);
    if (la != lb) return la < lb;
    return !sign_of(a.y) && sign_of(b.y);  // dropped: False < True
}
// The heap is 8-ary: node p's children are nodes 8p + 1 .. 8p + 8, and node p lives in slot p + 7 of the sender's array, so
// the eight children of a node are one aligned 128-byte line -- a level of the sift-down is one trip to memory for the lane,
// and 4 096 events in flight are 4 levels, not 12 (the lane's trips, one after the other, are what an interval of this
// build lasts).  kHeapPad more slots per array than events.
constexpr uint32_t kHeapPad = 8;
__device__ __forceinline__ double2 *heap_node(double2 *H, uint32_t p) { return H + 7u + p; }
__device__ __forceinline__ void heap_push(double2 *H, uint32_t &n, const double2 v) {
    uint32_t pos = n++;
    while (pos > 0) {
        const uint32_t parent = (pos - 1u) >> 3;
        const double2 pv = ld_rec(heap_node(H, parent));
        if (!heap_less(v, pv)) break;
        st_rec(heap_node(H, pos), pv);
        pos = parent;
    }
    st_rec(heap_node(H, pos), v);
}
__device__ __forceinline__ double2 heap_pop(double2 *H, uint32_t &n) {
    const double2 top = ld_rec(heap_node(H, 0));
    const double2 last = ld_rec(heap_node(H, --n));
    uint32_t pos = 0;
    for (;;) {
        const uint32_t c0 = 8u * pos + 1u;
        if (c0 >= n) break;
        double2 cv[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) {   // (one line: the loads leave together)
            cv[j].x = INFINITY; cv[j].y = INFINITY;
            if (c0 + j < n) cv[j] = ld_rec(heap_node(H, c0 + j));
        }
        double2 best = cv[0];
        uint32_t c = c0;
#pragma unroll
        for (uint32_t j = 1; j < 8u; j++)
            if (c0 + j < n && heap_less(cv[j], best)) { best = cv[j]; c = c0 + j; }
        if (!heap_less(best, last)) break;
        st_rec(heap_node(H, pos), best);
        pos = c;
    }
    if (n) st_rec(heap_node(H, pos), last);
    return top;
}

// pcc_noise_sorted.hip writes the events in flight back in no particular order and says so in bit 31 of SndBlk::heap_n:
// Floyd's bottom-up construction makes a heap of them (once; the event loop clears the bit).
__device__ __forceinline__ void heap_sift_down(double2 *H, uint32_t n, uint32_t pos, const double2 v) {
    for (;;) {
        const uint32_t c0 = 8u * pos + 1u;
        if (c0 >= n) break;
        double2 best = ld_rec(heap_node(H, c0));
        uint32_t c = c0;
        for (uint32_t j = 1; j < 8u && c0 + j < n; j++) {
            const double2 cv = ld_rec(heap_node(H, c0 + j));
            if (heap_less(cv, best)) { best = cv; c = c0 + j; }
        }
        if (!heap_less(best, v)) break;
        st_rec(heap_node(H, pos), best);
        pos = c;
    }
    st_rec(heap_node(H, pos), v);
}
__device__ __forceinline__ uint32_t heapify_if_loose(double2 *H, uint32_t raw_n) {
    const uint32_t n = raw_n & 0x7FFFFFFFu;
    if ((raw_n >> 31) && n > 1u)
        for (uint32_t p = (n - 2u) >> 3;; p--) {
            heap_sift_down(H, n, p, ld_rec(heap_node(H, p)));
            if (p == 0u) break;
        }
    return n;
}

template <int NS>
struct EngineOut {
    double now, q, tu;
    double nsend[NS];
    uint32_t sent[NS], acked[NS], lost[NS];
    uint32_t flags;
};

// One monitor interval of env i, by one lane: the reference's event loop (ns:123-178) with its dormant options --
// USE_LATENCY_NOISE (D.use_noise: one more draw of the stream per hop) and/or USE_CWND (cwnd[s]; 0xFFFFFFFF without) --
// for NS senders.  Every sender has its own heap of acknowledgement events (= its packets in flight, which is what its
// window counts); the event order (time, sender id, 'A' < 'S', hop, latency, dropped) (ns:42-43, 111) across the senders
// is the scan below: lower sender first at equal times, a sender's ACK before its SEND.
template <int NS>
__device__ __noinline__ EngineOut<NS> event_engine(const Dev &D, int64_t i, double start, double end, const double (&rate)[NS],
                                                   const double (&nsend0)[NS], uint32_t mi, const uint32_t (&cwnd)[NS]) {
    const double dl = D.env[i].dl, lr = D.env[i].lr, maxq = D.env[i].maxq, ebw = D.env[i].ebw;
    double q = D.env[i].q, tu = D.env[i].tu;
    const uint32_t episode = D.env[i].episode - 1;
    uint32_t mi_draws = 0, ep_draws = D.env[i].ep_draws;
    uint32_t hn[NS];
    double2 *H[NS], *R[NS];
    double nsend[NS];
    EngineOut<NS> o;
    o.flags = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = (int64_t)s * D.n + i;   // (the event heaps are [S][N]; the sender blocks [S][2N]: sidx)
        H[s] = D.noise_heap + (size_t)k * (D.noise_cap + kHeapPad);
        hn[s] = heapify_if_loose(H[s], D.snd[sidx(D, s, i)].heap_n);
        R[s] = D.noise_rtt + (size_t)k * D.noise_cap;
        nsend[s] = nsend0[s];
        o.sent[s] = o.acked[s] = o.lost[s] = 0;
    }
    double now = start;
    // What the loop reads of D, once: D is a reference here (this function is not inlined), and after every store to a heap
    // the compiler has to load a field of it again -- a dozen trips to memory per event, one after the other, in a loop
    // that is nothing but one lane's trips.
    const bool from_trace = D.rng_mode == PCC_RNG_TRACE, use_noise = D.use_noise != 0;
    const double noise_span = D.noise_span;
    const double *const trace_row = from_trace ? D.trace + i * D.trace_stride : nullptr;
    const int64_t trace_stride = D.trace_stride;
    const uint32_t key0 = D.key0, key1 = D.key1, gid = D.gid_base + (uint32_t)env_of(D, i), noise_cap = D.noise_cap;
    // (draws j, j+1, j+2, j+3 of an interval are the four words of one Philox block: computed once)
    uint32_t blk_of = 0xFFFFFFFFu, blk_w[4] = {0u, 0u, 0u, 0u};
    auto draw = [&]() -> double {
        if (from_trace) {
            const uint32_t pos = ep_draws++;
            if ((int64_t)pos >= trace_stride) { o.flags |= PCC_FLAG_TRACE_OVERRUN; return 1.0; }
            return trace_row[pos];
        }
        ep_draws++;
        const uint32_t j = mi_draws++;
        if ((j >> 2) != blk_of) { blk_of = j >> 2; philox4x32_10(blk_of, mi, episode, gid, key0, key1, blk_w); }
        const uint32_t x = j & 3u;
        return u32_to_unit(x == 0 ? blk_w[0] : x == 1 ? blk_w[1] : x == 2 ? blk_w[2] : blk_w[3]);   // = philox_packet_uniform
    };
    auto noisy = [&](double ll) -> double {  // ns:150-151, 171-172
        if (use_noise) ll *= 1.0 + noise_span * draw();
        return ll;
    };
    while (now < end) {  // ns:128
        int bs = 0;
        bool from_heap = false;
        double bt = INFINITY;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (hn[s] > 0) {
                const double t = fabs(ld_rec(heap_node(H[s], 0)).x);
                if (t < bt) { bt = t; bs = s; from_heap = true; }
            }
            if (nsend[s] < bt) { bt = nsend[s]; bs = s; from_heap = false; }
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (s != bs) continue;
            if (from_heap) {
                const double2 ev = heap_pop(H[s], hn[s]);
                now = fabs(ev.x);
                const double lat = fabs(ev.y);
                if (sign_of(ev.x)) {  // hop 2 == len(path): the sender hears of it (ns:139-146)
                    if (sign_of(ev.y)) o.lost[s]++;
                    else {
                        if (o.acked[s] < noise_cap) { double2 r; r.x = 0.0; r.y = lat; st_rec(R[s] + o.acked[s], r); }
                        else o.flags |= PCC_FLAG_RING_OVERFLOW;
                        o.acked[s]++;
                    }
                } else {  // hop 1: over the return link, which never queues (ns:147-153)
                    const double ll = noisy(dl + max0(0.0 - (now - 0.0)));
                    double2 nv;
                    nv.x = -(now + ll);
                    nv.y = sign_of(ev.y) ? -(lat + ll) : lat + ll;
                    if (hn[s] < noise_cap) heap_push(H[s], hn[s], nv);
                    else o.flags |= PCC_FLAG_RING_OVERFLOW;
                }
            } else {  // SEND (ns:155-175)
                now = nsend[s];
                // USE_CWND (ns:158-160): the packet leaves only while fewer than cwnd of the sender's are unacknowledged
                // -- every packet in flight is exactly one event of its heap -- but a blocked SEND still takes its noise
                // draw and passes through the link's queue and loss draw (ns:170-175 are outside the `if`)
                const bool can_send = hn[s] < cwnd[s];
                o.sent[s] += can_send ? 1u : 0u;
                nsend[s] = now + 1.0 / rate[s];  // ns:161
                const double qd = max0(q - (now - tu));
                const double ll = noisy(dl + qd);  // drawn before the loss decision (ns:171-175)
                const double lat = 0.0 + ll;
                bool dropped;
                if (draw() < lr) dropped = true;  // ns:73-74
                else {
                    q = qd; tu = now;            // ns:75-76
                    if (ebw + q > maxq) dropped = true;  // ns:78-79
                    else { q += ebw; dropped = false; }
                }
                double2 nv;
                nv.x = now + ll;
                nv.y = dropped ? -lat : lat;
                if (can_send) {
                    if (hn[s] < noise_cap) heap_push(H[s], hn[s], nv);
                    else o.flags |= PCC_FLAG_RING_OVERFLOW;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        D.snd[sidx(D, s, i)].heap_n = hn[s];
        o.nsend[s] = nsend[s];
    }
    D.env[i].ep_draws = ep_draws;
    D.env[i].mi_draws = mi_draws;
    o.now = now; o.q = q; o.tu = tu;
    return o;
}

// ns:454-477 for one env, by one lane: parameters, fresh link/sender/history state (the two warm-up MIs,
// ns:478-479, are run by the send and retire halves in warm mode).  The caller sets D.env[i].resetting.
// The senders' ring-pool slots go back to their free stacks here (nothing is in flight any more) -- pushes
// happen only in reset and retire launches, pops only in send launches: no stack races.
template <int NS>
__device__ __forceinline__ void release_ring_slots(const Dev &D, const int64_t i, const bool push = true) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = sidx(D, s, i);
        for (int c = 1; c < D.n_tiers; c++) {
            const uint32_t held = D.snd[k].ring_held[c];
            if (held) {
                if (push) pool_push(D, (uint32_t)c, held - 1u);
                D.snd[k].ring_held[c] = 0;
            }
        }
        D.snd[k].ring_tier = 0;  // the sender starts over in its own tier-0 rings
        D.snd[k].ring_base = D.tier_base[0] + (size_t)((int64_t)i * NS + s) * tier_slot_bytes(D, 0);
    }
}

// (the caller has released the ring-pool slots)
template <int NS>
__device__ __forceinline__ void reset_env(const Dev &D, const int64_t i, float *obs_out) {
    const uint32_t episode = D.env[i].episode;
    D.env[i].episode = episode + 1;

    double bw, lat, queue, loss, rate0[NS];
    const int64_t ie = env_of(D, i);   // (block i may be the shadow of env ie: the same links, keyed by the env's id)
    if (D.p_bw) {
        bw = D.p_bw[ie]; lat = D.p_dl[ie]; queue = D.p_queue[ie]; loss = D.p_loss[ie];
#pragma unroll
        for (int s = 0; s < NS; s++) rate0[s] = D.p_rate0[(int64_t)s * D.n + ie];
    } else {  // ns:455-466
        uint32_t w0[4], w1[4];
        const uint32_t gid = D.gid_base + (uint32_t)ie;
        philox4x32_10(0u, kParamTag, episode, gid, D.key0, D.key1, w0);
        philox4x32_10(1u, kParamTag, episode, gid, D.key0, D.key1, w1);
        bw = D.lo[0] + (D.hi[0] - D.lo[0]) * u32_to_unit(w0[0]);
        lat = D.lo[1] + (D.hi[1] - D.lo[1]) * u32_to_unit(w0[1]);
        queue = (double)(1 + (long long)exp(D.lo[2] + (D.hi[2] - D.lo[2]) * u32_to_unit(w0[2])));
        loss = D.lo[3] + (D.hi[3] - D.lo[3]) * u32_to_unit(w0[3]);
#pragma unroll
        for (int s = 0; s < NS; s++) rate0[s] = (D.lo[4] + (D.hi[4] - D.lo[4]) * u32_to_unit(w1[s])) * bw;
    }
    // caller-supplied parameters cannot be checked on the host (device arrays): never silent.  A link outside what
    // the formulation covers is flagged and replaced by a harmless stand-in -- with rate0 <= 0 or NaN the SEND times
    // would not advance and the send loops would never end
    bool bad = !(bw > 0.0) || !(bw <= 1e8) || !(lat > 0.0) || !(lat <= 1e6) || !(queue >= 1.0) || !(queue <= 1e9) ||
               !(loss >= 0.0) || !(loss <= 1.0);
#pragma unroll
    for (int s = 0; s < NS; s++) bad = bad || !(rate0[s] > 0.0) || !(rate0[s] <= 1e9);
    if (bad) {
        D.env[i].flags |= PCC_FLAG_BAD_PARAMS;
        bw = 100.0; lat = 0.1; queue = 2.0; loss = 0.0;
#pragma unroll
        for (int s = 0; s < NS; s++) rate0[s] = 100.0;
    }
    D.env[i].bw = bw; D.env[i].dl = lat; D.env[i].lr = loss;
    D.env[i].maxq = queue / bw;   // ns:64
    D.env[i].ebw = 1.0 / bw;      // ns:77
    D.env[i].q = 0.0; D.env[i].tu = 0.0; D.env[i].now = 0.0;
    D.env[i].run_dur = 3 * lat;   // ns:467
    D.env[i].steps = 0;
    D.env[i].done = 0;
    D.env[i].mi_draws = 0; D.env[i].ep_draws = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = sidx(D, s, i);
        D.snd[k].rate = rate0[s];
        D.snd[k].rate0 = rate0[s];
        D.snd[k].next_send = 1.0 / rate0[s];  // ns:111
        D.snd[k].ha = 0; D.snd[k].hd = 0; D.snd[k].ta = 0; D.snd[k].td = 0; D.snd[k].mi_sent = 0;
        D.snd[k].cwnd = 25;     // ns:209, 227
        D.snd[k].heap_n = 0;    // event-loop build: nothing in flight (the first SEND is next_send)
        D.snd[k].min_lat = 0.0;   // fresh sender id => no connection minimum yet (ns:229-233, so:158)
        D.snd[k].ack_rate = 0.f; D.snd[k].loss_rate = 0.f; D.snd[k].on_return_a = 0; D.snd[k].on_return_d = 0;
        D.snd[k].ep_return = 0.0;
        // all-empty history (so:57-62): every metric of an empty MI is 0 except the two ratios
        // (a shadow has no history row: the retire half writes the empty history when it swaps the shadow in)
        float *hist = D.hist + ((int64_t)ie * NS + s) * D.HF;
        float *obs = obs_out ? obs_out + ((int64_t)ie * NS + s) * D.HF : nullptr;
        for (int h = 0; i < D.n && h < D.H; h++)
            for (int f = 0; f < D.F; f++) {
                const int id = D.fid[f];
                const double v = (id == PCC_M_SEND_RATIO || id == PCC_M_LATENCY_RATIO) ? 1.0 : 0.0;
                const float x = (float)(v / c_metric_scale[id]);
                hist[h * D.F + f] = x;
                if (obs) obs[h * D.F + f] = x;
            }
    }
}

// Returns the env's predicted packet count for the next monitor interval (-1: nothing to report; -2: the env finished its
// episode and was reset here -- restart = 1 -- its warm-up intervals are due in the next send launch).
template <int NS, bool NOISE, int G>
__device__ __forceinline__ float retire_env(const Dev &D, const int64_t i, const Group g, int warm, uint32_t warm_mi,
                                            int last_warm, int restart, float *obs_out, float *reward_out, uint8_t *done_out,
                                            double *steps_out, const void *actions, int actions_f64) {
    if (warm && !D.env[i].resetting) return -1.0f;
    const bool lead = g.lane == 0;
    // profiling only: where a wavefront's retire time goes (lane 0's view), summed per workgroup
    const bool tl = prof_on(D) && (threadIdx.x & (kWave - 1)) == 0;
    uint64_t *tlw = prof_on(D) ? D.timeline + (int64_t)2 * D.n * 8 + (int64_t)blockIdx.x * 16 : nullptr;
    uint64_t tl_t = tl ? wall_clock64() : 0;
    if (tl && threadIdx.x == 0) tlw[0] = tl_t;  // this launch's start of the workgroup (slot 1: its end)
#define PCC_TL_STAMP(slot)                                                                               \
    if (tl) {                                                                                            \
        const uint64_t t_now = wall_clock64();                                                           \
        atomicAdd(reinterpret_cast<unsigned long long *>(&tlw[slot]), (unsigned long long)(t_now - tl_t)); \
        tl_t = t_now;                                                                                    \
    }

    const double dl = D.env[i].dl;
    const double start = D.env[i].now;
    const double run_dur = D.env[i].run_dur;
    const double end = start + run_dur;  // ns:124
    const uint32_t steps = D.env[i].steps;
    const unsigned long long total_before = D.env[i].total_sent;
    double now = start;

    double nsend[NS];
    uint32_t ha[NS], hd[NS], ta[NS], td[NS], sent[NS], acked[NS], lost[NS], from[NS];
    double2 *ra[NS], *rd[NS];
    uint32_t amask[NS], dmasks[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = sidx(D, s, i);
        nsend[s] = D.snd[k].next_send;
        ha[s] = D.snd[k].ha; hd[s] = D.snd[k].hd; ta[s] = D.snd[k].ta; td[s] = D.snd[k].td;
        sent[s] = D.snd[k].mi_sent;
        acked[s] = lost[s] = 0;
        from[s] = ha[s];
        const RingRef rr = ring_ref(D, k);
        ra[s] = rr.accepted(); rd[s] = rr.dropped();
        amask[s] = rr.mask(); dmasks[s] = rr.dmask();
    }
    uint32_t flags = 0;
    double noise_rate[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) noise_rate[s] = 0.0;

    if constexpr (NOISE) {
        // the event-loop build: the whole interval in the lead lane -- rate (and window) actions (ns:235-249; there is no
        // send half in this build), then the reference's event loop
        double rate[NS];
        uint32_t cw[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int64_t k = sidx(D, s, i);
            rate[s] = D.snd[k].rate;
            cw[s] = D.use_cwnd ? D.snd[k].cwnd : 0xFFFFFFFFu;
            if (!warm) {
                const int64_t ar = D.use_cwnd ? 2 * (i * NS + s) : i * NS + s;  // USE_CWND: [rate action, cwnd action] per sender
                double delta = actions_f64 ? ((const double *)actions)[ar] : (double)((const float *)actions)[ar];
                if (delta != delta) { delta = 0.0; flags |= PCC_FLAG_BAD_ACTION; }
                delta *= D.delta_scale;
                rate[s] = delta >= 0.0 ? rate[s] * (1.0 + delta) : rate[s] / (1.0 - delta);
                if (rate[s] > kMaxRate) rate[s] = kMaxRate;
                if (rate[s] < kMinRate) rate[s] = kMinRate;
                if (D.use_cwnd) {  // apply_cwnd_delta + set_cwnd: ns:243-249, 283-289
                    double dc = actions_f64 ? ((const double *)actions)[ar + 1] : (double)((const float *)actions)[ar + 1];
                    if (dc != dc) { dc = 0.0; flags |= PCC_FLAG_BAD_ACTION; }
                    dc *= D.delta_scale;
                    const double c = dc >= 0.0 ? (double)cw[s] * (1.0 + dc) : (double)cw[s] / (1.0 - dc);
                    cw[s] = c >= 5000.0 ? 5000u : (c < 4.0 ? 4u : (uint32_t)c);  // int(), then [MIN_CWND, MAX_CWND] (ns:33-34)
                    if (lead) D.snd[k].cwnd = cw[s];
                }
            }
            noise_rate[s] = rate[s];
        }
        EngineOut<NS> o;
        o.now = start; o.q = 0.0; o.tu = 0.0; o.flags = 0;
#pragma unroll
        for (int s = 0; s < NS; s++) { o.nsend[s] = nsend[s]; o.sent[s] = o.acked[s] = o.lost[s] = 0; }
        if (lead) {
            bool ran = false;
            if (D.noise_out != nullptr && D.noise_out[i].seq == D.noise_seq) {   // pcc_noise_sorted.hip has run this interval
                const NoiseOut r = D.noise_out[i];
                o.now = r.now; o.q = r.q; o.tu = r.tu; o.nsend[0] = r.nsend;
                o.sent[0] = r.sent; o.acked[0] = r.acked; o.lost[0] = r.lost; o.flags = r.flags;
                if constexpr (NS == 2) {   // (entry N + i: sender 1's)
                    const NoiseOut r1 = D.noise_out[D.n + i];
                    o.nsend[NS - 1] = r1.nsend; o.sent[NS - 1] = r1.sent; o.acked[NS - 1] = r1.acked; o.lost[NS - 1] = r1.lost;
                }
                ran = true;
            }
            if (!ran) o = event_engine<NS>(D, i, start, end, rate, nsend, warm ? warm_mi : steps + 2, cw);
#pragma unroll
            for (int s = 0; s < NS; s++) D.snd[sidx(D, s, i)].rate = rate[s];
            D.env[i].q = o.q; D.env[i].tu = o.tu;
        }
        // the RTT samples the lead lane stored are read by all lanes of the group below: same wavefront, same L1 -- a
        // workgroup-scope fence orders them (an agent-scope one writes back and invalidates the XCD's whole L2)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        now = gbcast<G>(o.now, 0);
        flags |= gbcast<G>(o.flags, 0);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            nsend[s] = gbcast<G>(o.nsend[s], 0);
            sent[s] = gbcast<G>(o.sent[s], 0); acked[s] = gbcast<G>(o.acked[s], 0); lost[s] = gbcast<G>(o.lost[s], 0);
            ra[s] = D.noise_rtt + ((size_t)s * D.n + i) * D.noise_cap;
            amask[s] = D.noise_cap - 1u;
            from[s] = 0;
        }
    } else if (start < end) {  // ns:128: otherwise the loop body never runs
        // candidates for the MI-ending event per sender: hop-1, hop-2 (with the ring it sits in)
        double t_h1[NS], t_h2[NS], l_h2[NS];
        uint32_t k_h2[NS];
        bool h2_is_drop[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            // ---- all four boundaries of this sender in one joint search (3-4 dependent loads)
            const double2 *const rings[4] = {ra[s], ra[s], rd[s], rd[s]};
            const uint32_t masks[4] = {amask[s], amask[s], dmasks[s], dmasks[s]};
            const uint32_t los[4] = {ha[s], ha[s], hd[s], hd[s]};
            const uint32_t his[4] = {ta[s], ta[s], td[s], td[s]};
            const double adds[4] = {dl, 0.0, dl, 0.0};
            Bound bnd[4];
            // where the boundaries would be if this interval retired what the last one did (per second of simulated time)
            const int64_t k_s = sidx(D, s, i);
            const float span = (float)run_dur;
            const uint32_t h_pa = ha[s] + (uint32_t)(D.snd[k_s].ack_rate * span), h_pd = hd[s] + (uint32_t)(D.snd[k_s].loss_rate * span);
            const uint32_t hints[4] = {h_pa, h_pa + D.snd[k_s].on_return_a, h_pd, h_pd + D.snd[k_s].on_return_d};
            PCC_TL_STAMP(3)  // state loads
            search_many<4, G>(g, rings, masks, los, his, adds, end, hints, bnd,
                           prof_on(D) ? reinterpret_cast<unsigned long long *>(tlw + 12) : nullptr);
            if (lead && run_dur > 0.0) {  // one 16-byte store (the ending event may move a boundary by one more: no matter)
                const float inv = 1.0f / span;
                uint4 pr;
                pr.x = __float_as_uint((float)(bnd[0].b - ha[s]) * inv);
                pr.y = __float_as_uint((float)(bnd[2].b - hd[s]) * inv);
                pr.z = bnd[1].b - bnd[0].b;
                pr.w = bnd[3].b >= bnd[2].b ? bnd[3].b - bnd[2].b : 0u;
                *reinterpret_cast<uint4 *>(&D.snd[k_s].ack_rate) = pr;
            }
            PCC_TL_STAMP(4)  // the joint boundary search
            // ---- accepted ring: send order is event order, the transitions are exact
            const uint32_t pa = bnd[0].b, ca = bnd[1].b;            // hop-2 / hop-1 events < end
            acked[s] = pa - ha[s];                                   // ns:144-146
            ha[s] = pa;
            double a2_t = INFINITY, a2_l = 0.0, a1_t = INFINITY, a1_l = 0.0;
            if (pa < ca) { a2_t = bnd[0].t + dl; a2_l = bnd[0].lat + dl; }  // first unretired is past the forward hop
            if (ca < ta[s]) { a1_t = bnd[1].t; a1_l = bnd[1].lat; }
            // ---- dropped ring: exact as found unless near-equal times surround the transition
            uint32_t pd = bnd[2].b, dk = 0xFFFFFFFFu;
            double d2_t = INFINITY, d2_l = 0.0;
            bool rotated = false;
            if (bnd[2].clean) {
                if (pd < td[s] && bnd[2].t < end) { dk = pd; d2_t = bnd[2].t + dl; d2_l = bnd[2].lat + dl; }
            } else {
                {   // (by the whole group: a near group is one round trip, not one per record)
                    const DropFix fx = fix_drop_boundary_g<G>(g.lane, g.shift, rd[s], dmasks[s], hd[s], td[s], bnd[2].b, dl, end);
                    pd = fx.p; dk = fx.cand_idx; d2_t = fx.cand_t; d2_l = fx.cand_lat;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                rotated = true;  // records may have moved inside the window
            }
            lost[s] = pd - hd[s];                                    // ns:141-143
            hd[s] = pd;
            double d1_t = INFINITY, d1_l = 0.0;
            if (!rotated && bnd[3].clean) {
                if (bnd[3].b < td[s]) { d1_t = bnd[3].t; d1_l = bnd[3].lat; }
            } else if (td[s] != pd) {
                const uint32_t cd = rotated ? search_boundary<G>(g, rd[s], dmasks[s], pd, td[s], 0.0, end)
                                            : (bnd[3].b < pd ? pd : bnd[3].b);
                const Cand c1 = drop_hop1_candidate_g<G>(g.lane, g.shift, rd[s], dmasks[s], pd, td[s], cd, end);
                d1_t = c1.t; d1_l = c1.lat;
            }
            // ---- best of each kind by the heap key (time, latency, dropped): ns:111,161,178
            t_h1[s] = (d1_t < a1_t || (d1_t == a1_t && d1_l < a1_l)) ? d1_t : a1_t;
            h2_is_drop[s] = (d2_t < a2_t || (d2_t == a2_t && d2_l < a2_l));
            t_h2[s] = h2_is_drop[s] ? d2_t : a2_t;
            l_h2[s] = h2_is_drop[s] ? d2_l : a2_l;
            k_h2[s] = h2_is_drop[s] ? dk : pa;
        }
        PCC_TL_STAMP(5)  // candidates, near-group repairs
        // ---- the event that ends the MI: smallest (time, sender, 'A' < 'S', hop) among the stream
        // heads, all >= end here; the reference still processes it (ns:128-131)
        int best = 0;
        double tb = t_h1[0];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (s > 0 && t_h1[s] < tb) { tb = t_h1[s]; best = 3 * s; }
            if (t_h2[s] < tb) { tb = t_h2[s]; best = 3 * s + 1; }
            if (nsend[s] < tb) { tb = nsend[s]; best = 3 * s + 2; }
        }
        now = tb;  // ns:131
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (best == 3 * s + 1) {  // hop-2: acknowledge / lose one more packet
                if (h2_is_drop[s]) {
                    if (k_h2[s] != hd[s]) {
                        if (lead) rotate_to_front(rd[s], dmasks[s], hd[s], k_h2[s]);
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    }
                    lost[s]++;
                    hd[s]++;
                } else {
                    acked[s]++;
                    ha[s]++;
                }
            } else if (best == 3 * s + 2) {  // SEND: one more packet leaves (ns:155-178)
                const double t = nsend[s];
                double q = D.env[i].q, tu = D.env[i].tu;
                double u;
                if (D.rng_mode == PCC_RNG_TRACE) {
                    uint64_t pos = 0;
#pragma unroll
                    for (int x = 0; x < NS; x++) pos += (uint64_t)ta[x] + td[x];
                    if ((int64_t)pos >= D.trace_stride) { flags |= PCC_FLAG_TRACE_OVERRUN; u = 1.0; }
                    else u = D.trace[i * D.trace_stride + pos];
                } else {
                    uint32_t j = 0;
#pragma unroll
                    for (int x = 0; x < NS; x++) j += sent[x];
                    if (D.use_cwnd) j = D.env[i].mi_draws;  // draws, not packets: blocked SENDs drew too
                    u = philox_packet_uniform(D, D.gid_base + (uint32_t)env_of(D, i), D.env[i].episode - 1,
                                              warm ? warm_mi : steps + 2, j);
                }
                if (D.use_cwnd && D.rng_mode == PCC_RNG_TRACE) {
                    const uint32_t pos = D.env[i].ep_draws;
                    if ((int64_t)pos >= D.trace_stride) { flags |= PCC_FLAG_TRACE_OVERRUN; u = 1.0; }
                    else u = D.trace[i * D.trace_stride + pos];
                }
                // USE_CWND (ns:251-255): everything due before this event is retired, so what is in
                // flight is exactly what the rings still hold
                const bool can_send = !D.use_cwnd || (ta[s] - ha[s]) + (td[s] - hd[s]) < D.snd[sidx(D, s, i)].cwnd;
                if (D.use_cwnd && lead) D.env[i].ep_draws += 1u;
                const double rate = D.snd[sidx(D, s, i)].rate;
                sent[s] += can_send ? 1u : 0u;
                nsend[s] = t + 1.0 / rate;
                bool dropped;
                const double2 rec = link_send(t, u < D.env[i].lr, dl, D.env[i].maxq, D.env[i].ebw, q, tu, dropped);
                if (!can_send) {
                    // blocked by the window: the link saw it (queue, draw), nothing is in flight
                } else if (dropped) {
                    if (lead) st_rec(rd[s] + (td[s] & dmasks[s]), rec);
                    td[s]++;
                } else {
                    if (lead) st_rec(ra[s] + (ta[s] & amask[s]), rec);
                    ta[s]++;
                }
                if (ta[s] - ha[s] > amask[s] + 1u || td[s] - hd[s] > dmasks[s] + 1u) flags |= PCC_FLAG_RING_OVERFLOW;
                if (lead) { D.env[i].q = q; D.env[i].tu = tu; }
            }
        }
        (void)l_h2;
    }

    PCC_TL_STAMP(6)  // the MI-ending event
    // ---- state
    unsigned long long sent_total = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) sent_total += sent[s];
    // the ordering of dropped packets rests on kNearTol * now << 1/bw (near groups never span two packet times)
    if (now * (64.0 * kNearTol) > D.env[i].ebw) flags |= PCC_FLAG_TIME_RANGE;
    if (lead) {
        if (flags) D.env[i].flags |= flags;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int64_t k = sidx(D, s, i);
            D.snd[k].ha = ha[s]; D.snd[k].hd = hd[s]; D.snd[k].ta = ta[s]; D.snd[k].td = td[s];  // one 16-byte store
        }
    }
    if (warm) {  // reset(): the two warm-up MIs are not recorded (ns:478-479)
        if (lead) {
            D.env[i].now = now;
            D.env[i].total_sent = total_before + sent_total;
#pragma unroll
            for (int s = 0; s < NS; s++) D.snd[sidx(D, s, i)].next_send = nsend[s];
            if (last_warm) D.env[i].resetting = 0;
        }
        return -1.0f;
    }
    // (the rest of the env's state is written at the end, next to its neighbours in the block)

    // ---- metrics, history, observation, reward: ns:416-438 with so:44-73
    bool need_halves = steps_out != nullptr;
    for (int f = 0; f < D.F; f++) {
        const int id = D.fid[f];
        need_halves |= (id == PCC_M_LATENCY_INCREASE || id == PCC_M_ACK_LATENCY_INFLATION ||
                        id == PCC_M_SENT_LATENCY_INFLATION);
    }
    const double dur = now - start;  // ns:311-314
    double new_run_dur = run_dur, rate_sum = 0.0;
    uint32_t cost_word = 0, sent_all = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) sent_all += sent[s];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = sidx(D, s, i);
        double lat = 0.0, inc = 0.0;
        PCC_TL_STAMP(7)  // state write-back
        if (acked[s] > 0 && !prof_skip(D, 1))
            rtt_means<G>(g, ra[s], amask[s], from[s], acked[s], NOISE ? 0.0 : dl, need_halves, lat, inc);  // noise: the samples are whole RTTs
        PCC_TL_STAMP(8)  // RTT means
        // everything the rest of the MI reads, in one batch of loads (one round trip, not five)
        float *hist = D.hist + ((int64_t)i * NS + s) * D.HF;
        const int keep = D.HF - D.F;
        constexpr int kRows = 32 / G;                  // passes of G lanes over the usual 10 x 3 history
        const bool small_hist = D.HF <= kRows * G;
        float old_row[kRows];                          // the history entries this lane rolls down
#pragma unroll
        for (int b = 0; b < kRows; b++) old_row[b] = 0.f;
        if (small_hist) {
#pragma unroll
            for (int b = 0; b < kRows; b++) {
                const int x = b * G + (int)g.lane;
                if (x < keep) old_row[b] = hist[x + D.F];
            }
        }
        double min_lat = D.snd[k].min_lat;
        if (s == 0) cost_word = D.snd[k].send_cost;
        const double ep_before = D.snd[k].ep_return;
        const double rate_now = NOISE ? noise_rate[s] : D.snd[k].rate;
        rate_sum += rate_now;
        double m[PCC_N_METRICS];
        mi_metrics(sent[s], acked[s], lost[s], dur, lat, inc, min_lat, m);
        PCC_TL_STAMP(9)  // metrics
        const double reward =  // ns:194,205
            (10.0 * m[PCC_M_RECV_RATE] / (double)(8 * kBytesPerPacket) - 1e3 * m[PCC_M_AVG_LATENCY] -
             2e3 * m[PCC_M_LOSS_RATIO]) * kRewardScale;
        if (s == 0 && m[PCC_M_AVG_LATENCY] > 0.0) new_run_dur = 0.5 * m[PCC_M_AVG_LATENCY];  // ns:437-438

        // history roll (so:64-66) + observation (ns:400-404, so:68-73), G lanes wide
        float *obs = obs_out ? obs_out + ((int64_t)i * NS + s) * D.HF : nullptr;
        // the new MI's features: feature f lands in entry keep + f, i.e. in lane (keep + f) % G of one pass
        // (features 0..G-1 in nf0, G..2G-1 in nf1: at most 16 features).  The ids are wave-uniform (kernel arguments):
        // a scalar switch picks each value, divided only when its scale is not 1 (so:193-206: the two rates)
        float nf0 = 0.f, nf1 = 0.f;
        for (int f = 0; f < D.F; f++) {
            const int id = D.fid[f];
            double val;
            switch (id) {
                case 0: val = m[0] / 1e7; break;
                case 1: val = m[1] / 1e7; break;
                case 2: val = m[2]; break;
                case 3: val = m[3]; break;
                case 4: val = m[4]; break;
                case 5: val = m[5]; break;
                case 6: val = m[6]; break;
                case 7: val = m[7]; break;
                case 8: val = m[8]; break;
                case 9: val = m[9]; break;
                case 10: val = m[10]; break;
                default: val = m[11]; break;
            }
            static_assert(PCC_M_SEND_RATE == 0 && PCC_M_RECV_RATE == 1 && PCC_N_METRICS == 12, "the switch above");
            if (((keep + f) & (G - 1)) == (int)g.lane) { if (f < G) nf0 = (float)val; else nf1 = (float)val; }
        }
        // an env that finishes its episode here and restarts (see the end of this function) shows the first observation
        // of its next episode: the all-empty history (so:57-62; every metric of an empty MI is 0 but the two ratios)
        const bool restarts = restart && steps + 1 >= D.max_steps;
        if (small_hist) {
#pragma unroll
            for (int b = 0; b < kRows; b++) {
                const int x = b * G + (int)g.lane;
                float v = (x - keep) < G ? nf0 : nf1;
                if (x < keep) v = old_row[b];
                if (x < D.HF && !prof_skip(D, 2)) {
                    if (restarts) {  // (the next episode's history starts empty -- whether its state is swapped in below or set up later)
                        const int id = D.fid[x % D.F];
                        v = (float)(((id == PCC_M_SEND_RATIO || id == PCC_M_LATENCY_RATIO) ? 1.0 : 0.0) / c_metric_scale[id]);
                    }
                    hist[x] = v;
                    if (obs) obs[x] = v;
                }
            }
        } else {
            for (int base = 0; base < D.HF && !prof_skip(D, 2); base += G) {
                const int x = base + (int)g.lane;
                float v = (x - keep) < G ? nf0 : nf1;
                if (x < keep) v = hist[x + D.F];
                if (x < D.HF) {
                    if (restarts) {
                        const int id = D.fid[x % D.F];
                        v = (float)(((id == PCC_M_SEND_RATIO || id == PCC_M_LATENCY_RATIO) ? 1.0 : 0.0) / c_metric_scale[id]);
                    }
                    hist[x] = v;
                    if (obs) obs[x] = v;
                }
            }
        }
        PCC_TL_STAMP(10)  // history + observation
        if (lead) {
            const double ret = ep_before + reward;
            D.snd[k].next_send = nsend[s]; D.snd[k].min_lat = min_lat;  // 16 bytes
            D.snd[k].ep_return = ret;
            if (steps + 1 >= D.max_steps) D.snd[k].last_return = ret;
            if (reward_out) reward_out[i * NS + s] = (float)reward;
        }
        if (steps_out) {
#pragma unroll
            for (int mb = 0; mb < PCC_N_METRICS; mb += G)
                if (mb + (int)g.lane < PCC_N_METRICS)
                    steps_out[(i * NS + s) * PCC_STEP_COLS + PCC_COL_METRIC0 + mb + g.lane] = select_metric(m, mb + (int)g.lane);
        }
        if (steps_out && lead) {
            double *row = steps_out + (i * NS + s) * PCC_STEP_COLS;
            row[PCC_COL_SENT] = (double)sent[s];
            row[PCC_COL_ACKED] = (double)acked[s];
            row[PCC_COL_LOST] = (double)lost[s];
            row[PCC_COL_RATE] = rate_now;
            row[PCC_COL_CUR_TIME] = now;
            row[PCC_COL_REWARD] = reward;
        }
    }
    if (lead) {
        if (steps_out)
            for (int s = 0; s < NS; s++) steps_out[(i * NS + s) * PCC_STEP_COLS + PCC_COL_RUN_DUR] = new_run_dur;
        const uint8_t done = (steps + 1 >= D.max_steps) ? 1 : 0;  // ns:444
        D.env[i].now = now; D.env[i].run_dur = new_run_dur;  // 16 bytes
        D.env[i].total_sent = total_before + sent_total;     // 16 bytes with the two below
        D.env[i].steps = steps + 1;
        D.env[i].done = done;
        if (done) *D.any_done = D.step_seq;  // somebody needs the auto-reset launches of this step (every writer writes the same word)
        if (done_out) done_out[i] = done;
    }
    if (restart && steps + 1 >= D.max_steps) {
        // auto-reset of envs that are not in lockstep, without extra launches.
        // (1) The env's SHADOW (block N + i) holds its next episode, prepared ahead of time by the refill kernel -- new links
        // (ns:469-477) and the two warm-up intervals (ns:478-479) already run: it is swapped in here, the env is filed like
        // every other by the packets its first interval will send, and the next send launch treats it like every other
        // (its rings are still the shadow's: load_env moves the records in flight into the env's own storage).
        // (2) No valid shadow (restart without bit 1; a masked reset overtook the refill; the warm-up intervals did not fit
        // the shadow's rings): the env is marked and filed in the restart list, and the next send half gives it new links
        // and runs the warm-up intervals right before its first interval (pcc_send_restart.hip) -- nothing in between reads
        // any of that (the first observation of an episode is the empty history, written above).
        float filed = -2.0f;
        if (lead) {
            release_ring_slots<NS>(D, i);  // here, not in the send launch: see release_ring_slots
            const int64_t sh = D.n + i;
            // (a shadow is trusted three steps after the launch that prepared it was queued: the caller's stream has waited
            // for that launch by then -- pcc_sim.hip -- so every line of it is what the refill wrote)
            const bool ok = (restart & 2) != 0 && D.env[sh].resetting == 0 && D.env[sh].episode == D.env[i].episode + 1u &&
                            D.step_seq - D.env[sh].fill_seq >= 3u && D.env[sh].params_gen == D.params_gen;
            if (ok) {
                D.env[i].bw = D.env[sh].bw; D.env[i].dl = D.env[sh].dl; D.env[i].lr = D.env[sh].lr; D.env[i].maxq = D.env[sh].maxq;
                D.env[i].ebw = D.env[sh].ebw; D.env[i].episode = D.env[sh].episode;
                D.env[i].q = D.env[sh].q; D.env[i].tu = D.env[sh].tu; D.env[i].now = D.env[sh].now; D.env[i].run_dur = D.env[sh].run_dur;
                D.env[i].total_sent = total_before + sent_total + D.env[sh].total_sent;
                D.env[i].steps = 0; D.env[i].done = 0; D.env[i].resetting = 0;
                D.env[i].mi_draws = D.env[sh].mi_draws; D.env[i].ep_draws = D.env[sh].ep_draws;
                if (D.env[sh].flags) D.env[i].flags |= D.env[sh].flags;
                double rates = 0.0;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const int64_t kp = sidx(D, s, i), ks = sidx(D, s, sh);
                    D.snd[kp].rate = D.snd[ks].rate; D.snd[kp].rate0 = D.snd[ks].rate0;
                    D.snd[kp].next_send = D.snd[ks].next_send; D.snd[kp].min_lat = D.snd[ks].min_lat;
                    D.snd[kp].ep_return = 0.0;
                    D.snd[kp].ring_base = D.snd[ks].ring_base; D.snd[kp].ring_tier = (uint8_t)kTierBorrowed;
                    D.snd[kp].ha = D.snd[ks].ha; D.snd[kp].hd = D.snd[ks].hd; D.snd[kp].ta = D.snd[ks].ta; D.snd[kp].td = D.snd[ks].td;
                    D.snd[kp].mi_sent = D.snd[ks].mi_sent; D.snd[kp].cwnd = D.snd[ks].cwnd; D.snd[kp].heap_n = D.snd[ks].heap_n;
                    D.snd[kp].ack_rate = D.snd[ks].ack_rate; D.snd[kp].loss_rate = D.snd[ks].loss_rate;
                    D.snd[kp].on_return_a = D.snd[ks].on_return_a; D.snd[kp].on_return_d = D.snd[ks].on_return_d;
                    rates += D.snd[ks].rate;
                }
                filed = (float)(D.env[sh].run_dur * rates);   // (the class of the new episode's first interval)
                if (shadow_list(&D.env[sh])) {                 // consumed: the refill kernel prepares the episode after this one
                    const uint32_t row = D.step_seq & 3u;
                    const uint32_t at = atomicAdd(&D.refill_count[row * kCntStride], 1u);
                    D.refill_list[(size_t)row * (size_t)D.n + at] = (uint32_t)i;
                }
                atomicAdd(&D.restart_stats[0], 1ull);
            } else {
                D.env[i].resetting = 2;
                atomicAdd(&D.restart_stats[1], 1ull);
            }
        }
        return __shfl(filed, 0, G);
    }
    PCC_TL_STAMP(11)  // outputs
    if (tl) atomicMax(reinterpret_cast<unsigned long long *>(&tlw[1]), (unsigned long long)tl_t);
#undef PCC_TL_STAMP
    // prediction for the next MI's send half: packets ~ MI length x current rate (the next action
    // moves the rate by at most a few percent)
    float pred = (float)(new_run_dur * rate_sum);
    // ... and for an env of the wave-path classes, what those packets COST the last time (SndBlk::send_cost): the send launch
    // hands its items out largest class first, and an env whose passes stop every few packets (the accept chain: 90-150 ns per
    // packet against ~25) is an item of 100-200 us -- filed by its packet count it starts 40-60 us into the launch and is the
    // last thing running (config 5, profiles/r06_config5_timeline.json).  Filed as if it had as many times the packets as
    // its cost per packet exceeds the usual, it starts first.  Speed only: every path is exact whatever the class.
    if (!NOISE && pred >= (float)D.heavy_predict && sent_all != 0u && (cost_word >> 24) == ((steps + 2u) & 0xFFu)) {
        const float ns_per_packet = (float)(cost_word & 0xFFFFFFu) * 10.0f / (float)sent_all;
        const float factor = fminf(fmaxf(ns_per_packet * (1.0f / 32.0f), 1.0f), 8.0f);
        // (one sender: not across the team threshold -- four wavefronts per env are for envs that ARE that large)
        const float cap = (NS == 1 && D.team_predict < 1e9) ? fmaxf(pred, 0.98f * (float)D.team_predict) : 3.0e38f;
        pred = fminf(pred * factor, cap);
    }
    return pred;
}

}  // namespace
