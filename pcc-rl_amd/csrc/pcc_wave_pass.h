// pcc_wave_pass.h -- the wave path of the send half: one env's SEND events of a monitor interval by all 64 lanes of a
// wavefront (heavy_mi, one sender; heavy_mi2, two senders), or by the four wavefronts of a workgroup (a team pass).
// Included by every kernel file that sends envs this way: the wave kernel (its items), the light kernel (the tail of a
// light item), the restart kernel and the small-batch kernel.
#pragma once
#include "pcc_dev.h"

namespace {

__device__ __forceinline__ uint32_t exponent_bits(double x) { return ((uint32_t)__double2hiint(x) >> 20) & 0x7FFu; }

struct SendState {  // wave-uniform while an env is processed by the whole wave
    double q, tu, t;
    uint32_t a, d, sent, flags;
    uint32_t prof_closed, prof_other;  // profile build: committed closed-form passes / chain + serial passes of the env
};

// ---- pieces of the wave pass ---------------------------------------------------------------
// Lindley map b -> max(b + s, c) of the token bucket (see heavy_mi); maps compose as
// (s2, c2) after (s1, c1) = (s1 + s2, max(c1 + s2, c2)), so the tokens every lane starts with come
// from one prefix scan of the lanes' composites: six DPP steps, no LDS.
constexpr int kLindNone = -(1 << 28);  // "-inf" with room for every shift a pass can add
#ifndef PCC_RELAX_SWEEPS
#define PCC_RELAX_SWEEPS 1
#endif
// 1: the accept chain of heavy_mi2 by sweeps over all its packets side by side; 0: from accepted packet to accepted packet
constexpr int kRelaxSweeps = PCC_RELAX_SWEEPS;
constexpr uint32_t kMaxPasses = 1u << 22;  // passes of one env and interval before the wave path gives up (PCC_FLAG_INTERNAL)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void lind_step(int &s, int &c) {
    // lanes without a source (start of a row / rows the mask leaves out) read the identity map
    const int ps = __builtin_amdgcn_update_dpp(0, s, CTRL, ROW_MASK, 0xF, false);
    const int pc = __builtin_amdgcn_update_dpp(kLindNone, c, CTRL, ROW_MASK, 0xF, false);
    const int nc = pc + s > c ? pc + s : c;  // this lane's map after the source's
    s = ps + s;
    c = nc;
}

// exclusive prefix over the 64 lanes: on return (s, c) is the composite of all lower lanes' maps, (tot_s, tot_c) the
// composite of all 64 (what the next wavefront of a team starts from)
__device__ __forceinline__ void lind_exclusive_scan(int &s, int &c, int &tot_s, int &tot_c) {
    lind_step<0x111, 0xF>(s, c);  // row_shr:1
    lind_step<0x112, 0xF>(s, c);  // row_shr:2
    lind_step<0x114, 0xF>(s, c);  // row_shr:4
    lind_step<0x118, 0xF>(s, c);  // row_shr:8  -> inclusive inside each row of 16
    lind_step<0x142, 0xA>(s, c);  // row_bcast:15 into rows 1 and 3
    lind_step<0x143, 0xC>(s, c);  // row_bcast:31 into rows 2 and 3 -> inclusive over the wave
    tot_s = __builtin_amdgcn_readlane(s, kWave - 1);
    tot_c = __builtin_amdgcn_readlane(c, kWave - 1);
    s = __builtin_amdgcn_update_dpp(0, s, 0x138, 0xF, 0xF, false);          // wave_shr:1
    c = __builtin_amdgcn_update_dpp(kLindNone, c, 0x138, 0xF, 0xF, false);
}

// What the W wavefronts of a TEAM pass tell each other through LDS (heavy_mi<.., W> with W > 1: one env sent by a whole
// workgroup, 256 W positions per pass).  Three exchanges per pass, each followed by one workgroup barrier.
constexpr int kTeamMax = 4;
struct TeamX {
    int cnt[kTeamMax];           // 1: packets the wavefront accepts (regimes without the token scan)
    int ls[kTeamMax], lc[kTeamMax];  // 1: the wavefront's composite Lindley map (token scan)
    int b0;                      // 1: tokens in front of the pass's first packet
    uint32_t pstop[kTeamMax];    // 2: first position of the wavefront that ends the pass (256 = none)
    uint32_t jstop[kTeamMax];    // 2: packets accepted before it (by the whole team)
    uint32_t sflag[kTeamMax];    // 2: ... and whether a broken precondition ended it
    uint32_t has_last[kTeamMax]; // 3: the wavefront committed a packet that reached the queue,
    double last_q[kTeamMax], last_t[kTeamMax];  // 3: and the link state behind its last one
};

__device__ __forceinline__ double pow2_f64(int e_unbiased) {  // 2^e for a normal result
    return __hiloint2double((e_unbiased + 1023) << 20, 0);
}

// One monitor interval of SENDs for ONE env by all 64 lanes (NS = 1), up to 256 packets per pass:
// lane l owns pass positions 4l..4l+3 = one Philox block.  Exact, not approximate -- every pass
// reproduces the per-packet recurrence of Link.packet_enters_link (ns:66-84) bit for bit; the
// argument is spelled out (and machine-checked against the plain recurrence, on fuzzed states and on
// the MI start states of whole episodes) in tests/models/send_pass_model.c, which mirrors this
// function operation by operation.
//   * Send times: inside one binade t_{k+1} = fl(t_k + gap) advances by a constant G = t_1 - t_0,
//     an exact multiple of ulp(t), so position k is sent at t_0 + k G -- checked per pass (two
//     equal increments, t_0 >= 512 gap so that k G is exact, no binade crossing inside the pass).
//   * Regime A, "always empty": gap >= 1/bw and the first packet already finds the queue drained.
//     Then every packet does: latency dl, accepted unless lost at random, queue = 1/bw behind it.
//   * Regime B, "backlogged in one binade": with t, tu >= maxq the drain q - (t - tu) is exact, and
//     while the queue never empties and q stays inside one binade [2^e, 2^(e+1)) every quantity is a
//     multiple of u = ulp(q) and fl(1/bw + x) = x + R, R = 1/bw rounded to a multiple of u.  After j
//     accepted packets the queue seen at t is exactly x = q0 + j R - (t - tu0), and "accepted" is a
//     token bucket: packet k is accepted iff it is not a random loss and j(k) < N_k =
//     floor((maxq - R - q0 + (t_k - tu0)) / R) + 1.  b_k = N_k - j(k) follows Lindley's recursion
//     b' = max(b - m, 0) + a (m: not lost, a: token arrivals), a (max,+)-linear map: one prefix scan
//     gives all 256 decisions.  When the queue has room for >= 300 packets, or the sender is slower
//     than the link, tokens never run out and the scan is skipped.  A packet that breaks a
//     precondition (queue empties, q leaves the binade) is detected per packet; the pass commits the
//     prefix before the first such packet.
//   * Otherwise (episode start, binade changes, ties of the rounding of 1/bw): a few packets with the
//     plain recurrence, wave-uniform.
// Records leave in send order as dense runs per ring -> coalesced stores.
// W > 1: a TEAM pass -- the W wavefronts of a workgroup send one env together, wavefront wv owning positions
// 256 wv .. 256 wv + 255 of a pass of 256 W.  Every wavefront carries the same SendState and takes the same decisions
// (what one wavefront needs of the others -- accepted packets / Lindley composite of the wavefronts before it, the first
// position that ends the pass, the link state behind the last packet -- goes through X in LDS, one barrier each); the
// serial and accept-chain fallbacks are computed by all of them alike and stored by wavefront 0.  The same pass with 256
// lanes is what tests/models/send_pass_model.c checks (pcc_model_set_lanes).
// When regime C is tried.  Product: the static test `maxq - 64/bw < B` (random losses let the queue dip well below
// maxq - 1/bw).  -DPCC_ADAPTIVE_C=1 builds the variant that tries it only once regime B was stopped by a packet that left its
// binade: every pass is exact, so results are the same bit for bit -- round 3 built this variant, found light items of batches
// >= 1 024 envs skipping envs, and left it unexplained (the send kernel spilled registers then; it does not now): the variant
// library is built and run through the parity tests (tests/test_variants.py) so that the finding stays checked.
#ifndef PCC_ADAPTIVE_C
#define PCC_ADAPTIVE_C 0
#endif
#if PCC_ADAPTIVE_C
#define PCC_TRY_REGIME_C (c_armed)
#else
#define PCC_TRY_REGIME_C (maxq - 64.0 * ebw < B)
#endif

// STAGE (round 6; the send kernel's wavefronts): the records of a closed-form pass leave through `stage`, 256 slots of LDS per
// wavefront.  A lane owns four CONSECUTIVE positions, so the store instruction of position i had neighbouring lanes ~4 ring
// slots (64 bytes) apart: every lane a request of its own to the compute unit's address path -- one per record, 9.85 M write
// requests per send launch (TCP_TCC_WRITE_REQ, profiles/r06_pmc_attribution.json), next to the lane rounds' scattered stores
// that wait behind them (a lane-round iteration costs 85-115 ns of arithmetic alone and 240-260 next to the wave path:
// profiles/r06_lane_round_microbench.txt, r06_send_timeline_*.json).  Staged, the accepted records of the pass sit in LDS in
// ring order from slot 0 up, the dropped ones from slot 255 down, and lane l stores slot 64 k + l: 1 KB of consecutive ring
// bytes per instruction, a quarter of the requests.  Same records, same ring slots.
// NP: positions per lane of a closed-form pass (a multiple of 4: whole Philox blocks; 4 in the product).  Round 6 built 8 for the
// wave path's items -- the pass header, the regime constants, the token division and the scans are per PASS, ~345 of a
// 4-position pass's ~765 vector instructions -- and measured no gain (pcc_send_bodies.h: PCC_WAVE_POS).  Same decisions, same
// records at either length: the passes are exact whatever their length (tests/models/send_pass_model.c checks 256 and 1 024
// positions; the GPU parity suite ran through 512).
template <bool TRACE, int W, bool STAGE = false, int NP = 4>
__device__ __forceinline__ void heavy_mi(const Dev &D, uint32_t lane, const uint32_t wv, TeamX *X, double dl, double lr,
                                         uint32_t thr, bool always, double maxq, double ebw, double gap, double end,
                                         uint32_t episode, uint32_t mi, uint32_t gid, const double *trace, char *base,
                                         uint32_t cap, SendState &st, double2 *stage = nullptr) {
    static_assert(W >= 1 && W <= kTeamMax, "team size");
    const uint32_t mask_b = (cap - 1u) << 4, dmask_b = (2u * cap - 1u) << 4, cap_b = cap << 4;
    static_assert(NP % 4 == 0 && NP >= 4 && NP <= 16, "whole Philox blocks per lane; the per-position bits fit a word");
    constexpr uint32_t kLanePos = (uint32_t)NP, kWavePos = kLanePos * kWave, kPass = kWavePos * W, kPosMask = (1u << NP) - 1u;
    const uint32_t glane = (W > 1 ? wv * kWave : 0u) + lane;  // lane of the team
    const bool first_lane = glane == 0u;
    const bool writer = W == 1 || wv == 0u;  // who stores what every wavefront of the team computes alike
    uint32_t serial_len = 8;
    uint32_t chain_left = 0;  // passes to send by the accept chain before the closed forms are tried again
#if PCC_ADAPTIVE_C
    bool c_armed = false;     // (variant build) regime B was stopped by a packet that broke its preconditions: regime C gets its chance
#endif
    uint32_t guard = 0;  // every pass commits at least one packet; a loop that does not end is a bug, not a reason to hang the GPU
    while (st.t < end) {
        if (++guard > kMaxPasses) { st.flags |= PCC_FLAG_INTERNAL; break; }
        const uint64_t dbg_c0 = prof_counters(D) ? __builtin_readcyclecounter() : 0;
        const double t0 = st.t;
        const double t1s = t0 + gap;
        const double G = t1s - t0;
        const double t2s = t1s + gap;
        const double tend = t0 + (double)kPass * G;
        // positions whose send time leaves the binade of t0 are not part of the pass (t0 + k G would not be
        // exact there): the pass ends at lim = min(end, top of the binade)
        const double ttop = pow2_f64((int)exponent_bits(t0) - 1022);
        const double lim = end < ttop ? end : ttop;
        const bool ok_t = (t2s - t1s == G) && (G > 0.0) && (t0 >= ((double)kPass + 4.0) * gap) &&
                          (exponent_bits(t0) == exponent_bits(t2s));
        const uint32_t skip = st.sent & 3u;  // positions of lane 0's Philox block that were sent before this pass
        const double D0 = t0 - st.tu;
        const double x0 = st.q - D0;         // the queue the first packet sees (before max0), ns:66-67
        // ---- regime (wave-uniform)
        int regime = 0;  // 0 serial, 1 = A, 2 = B, 3 = C
        uint32_t e = 0;
        double u = 0.0, R = 0.0;
        int64_t Q0i = 0, D0i = 0, Gi = 0, Ri = 0, Ci = 0;
        int64_t Mi3 = 0, Bi3 = 0, Ii3 = 0;  // regime C: maxq, the straddled power of two, floor(1/bw) in units of v
        int cl3 = 0;                        // regime C: [frac(1/bw in units of v) > 1/2]
        bool maxq_above = false, free_mode = false;
        if (ok_t && chain_left == 0u) {
            if (G >= ebw && !(x0 > 0.0)) {
                regime = 1;
            } else if (W == 1 && [&]() {
                // ---- regime C, "full queue straddling a power of two": maxq sits just above B = 2^E (maxq - 1/bw < B <=
                // maxq), so the full queue lives in two binades -- values below B are multiples of v = ulp(B) / 2, values
                // from B up multiples of 2 v, and fl(qcur + 1/bw) rounds to the grid its result lands on.  Regime B would
                // stop every few packets (q leaves its binade) and the accept chain take over at ~95 ns per packet: a few
                // such envs of 1-2 k packets were the critical path of whole launches (0.150 instead of 0.113 ms; random losses let
                // the queue dip well below maxq - 1/bw, so the regime is tried up to 64 packets above B).  In
                // units of v with 1/bw = (I + f) v, 0 < f < 1, f != 1/2: a result below B is n + I + cl (cl = [f > 1/2]),
                // a result from B up is n + I rounded up to even.  The pass takes decisions and landing sides from the
                // base trajectory (the constant increment R0 = I + cl: the token scan of regime B in units of v), which
                // is off the true one by at most j units after j accepts -- a packet whose decision or landing side is
                // closer than that to its threshold ends the pass -- and then runs the two-state automaton (parity of
                // the queue) over the accepted packets to get every correction.  tests/models/send_pass_model.c, regime C.
                const uint32_t eM = exponent_bits(maxq), eq = exponent_bits(st.q), eb = exponent_bits(ebw);
                const double B = pow2_f64((int)eM - 1023);
                // (tried when the full queue's band reaches down to B -- losses widen it -- or once regime B was stopped)
                if (!((st.q > 0.0) && eM > 66u && eM < 1100u && (eq == eM || eq + 1u == eM) && PCC_TRY_REGIME_C && (x0 > 0.0) &&
                      (st.tu + st.tu >= tend) && exponent_bits(st.tu) >= eM && eb + 2u <= eM))
                    return false;
                const double v = pow2_f64((int)eM - 1 - 1023 - 52), inv_v = pow2_f64(-((int)eM - 1 - 1023 - 52));
                const double probe = pow2_f64((int)eM - 1 - 1023);
                const double R0 = (probe + ebw) - probe;  // 1/bw on the grid of v
                const double errv = ebw - R0;
                const double span = (D0 + (double)kPass * G) * inv_v;
                if (!(span < 4.0e18 && R0 > 0.0 && errv != 0.0 && fabs(errv) != 0.5 * v)) return false;
                Q0i = (int64_t)(st.q * inv_v);
                D0i = (int64_t)(D0 * inv_v);
                Gi = (int64_t)(G * inv_v);
                Ri = (int64_t)(R0 * inv_v);
                if (!(Gi < Ri)) return false;  // the sender is not faster than the link: not this regime
                Mi3 = (int64_t)(maxq * inv_v);
                Bi3 = (int64_t)(B * inv_v);
                cl3 = errv < 0.0 ? 1 : 0;
                Ii3 = Ri - cl3;
                Ci = (Mi3 - Ri) - Q0i + D0i;
                u = v; R = R0; e = eM - 1u;
                return true;
            }()) {
                regime = 3;
            } else {
                e = exponent_bits(st.q);
                const uint32_t eb = exponent_bits(ebw);
                bool ok = (st.q > 0.0) && e > 64u && e < 1100u && (st.tu + st.tu >= tend) && (x0 > 0.0) &&
                          (eb <= e) && exponent_bits(st.tu) >= e && exponent_bits(maxq) >= e;
                if (ok) {
                    u = pow2_f64((int)e - 1023 - 52);
                    const double inv_u = pow2_f64(-((int)e - 1023 - 52));
                    const double probe = pow2_f64((int)e - 1023);
                    R = (eb == e) ? ebw : (probe + ebw) - probe;
                    const double err = ebw - R;
                    const bool tie = fabs(err) == 0.5 * u;
                    const double span = (D0 + (double)kPass * G) * inv_u;  // everything in units of u must fit an int64
                    ok = span < 4.0e18 && R > 0.0;
                    if (ok) {
                        Q0i = (int64_t)(st.q * inv_u);
                        D0i = (int64_t)(D0 * inv_u);
                        Gi = (int64_t)(G * inv_u);
                        Ri = (int64_t)(R * inv_u);
                        // room in the queue in packets (estimate): with >= kPass + 44 no packet of this pass can be
                        // tail-dropped and the token arithmetic is not needed (maxq / u may not fit an int64)
                        const double room = ((maxq - R) - x0) / R;
                        free_mode = room >= (double)kPass + 44.0;
                        const int64_t Mi = free_mode ? 0 : (int64_t)(maxq * inv_u);
                        Ci = (Mi - Ri) - Q0i + D0i;  // tokens before packet k: floor((Ci + Ri + k Gi) / Ri) >= 0
                        // a tie rounds to even: x + R holds only while every x is an even multiple of u
                        if (tie && ((Q0i | D0i | Gi) & 1)) ok = false;
                        maxq_above = exponent_bits(maxq) > e;
                        // the first packet would already take q out of the binade: no point in trying
                        const uint32_t es0 = exponent_bits(x0 + R);
                        if (es0 < e || (es0 > e && maxq_above)) ok = false;
                    }
                }
                if (ok) regime = 2;
            }
        }

        if (regime != 0) {
            // ---- loss decisions of the lane's four positions (bit i: lost at random, ns:73)
            const int kbase = NP * (int)glane - (int)skip;  // packet index (within the pass) of position 0 of this lane
            uint32_t rnd4 = 0;
            if (TRACE) {
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    const int k = kbase + i;
                    const int64_t pos = (int64_t)((uint64_t)st.a + st.d) + k;
                    double uu = 1.0;
                    if (k >= 0 && pos < D.trace_stride) uu = trace[pos];
                    rnd4 |= (uu < lr ? 1u : 0u) << i;
                }
            } else {
#pragma unroll
                for (int blk = 0; blk < NP / 4; blk++) {   // (position p of the pass is packet 4 (sent >> 2) + p of the interval)
                    uint32_t w[4];
                    philox4x32_10((st.sent >> 2) + (uint32_t)(NP / 4) * glane + (uint32_t)blk, mi, episode, gid, D.key0, D.key1, w);
#pragma unroll
                    for (int i = 0; i < 4; i++) rnd4 |= ((always || w[i] < thr) ? 1u : 0u) << (4 * blk + i);
                }
            }
            // ---- which positions hold a packet of this MI, and which of those reach the queue
            uint32_t ex4 = 0, m4 = 0;
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const int k = kbase + i;
                const double tki = t0 + (double)(k < 0 ? 0 : k) * G;  // exact
                const bool ex = k >= 0 && tki < lim;
                ex4 |= (ex ? 1u : 0u) << i;
                m4 |= ((ex && !((rnd4 >> i) & 1u)) ? 1u : 0u) << i;
            }
            // what the lane keeps of its four positions: accepted / flagged bits, and the queue (before
            // max0) and the accepted count at its first position -- the rest is replayed when needed
            uint32_t acc4 = 0, flag4 = 0;
            uint32_t up4 = 0, cp4 = 0;   // regime C: accepted packets that land from B up; corrections (2 bits each, +1)
            int c_before = 0;            // regime C: corrections accumulated in front of this lane
            int64_t xi_base = 0;
            double x_base = 0.0;
            int j_base = 0;
            if (regime == 1) {
                acc4 = m4;
                int in_wave = 0;
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    const uint64_t bm = __ballot((acc4 >> i) & 1u);
                    j_base += (int)count_below(bm);
                    in_wave += (int)__popcll(bm);
                }
                if constexpr (W > 1) {  // exchange 1: packets accepted by the wavefronts before this one
                    if (lane == 0) X->cnt[wv] = in_wave;
                    __syncthreads();
                    for (uint32_t w2 = 0; w2 < wv; w2++) j_base += X->cnt[w2];
                }
            } else {
                const bool over = regime == 3 || (!free_mode && Gi < Ri);  // overdriven and close to full: the token scan decides
                const int k0 = kbase < 0 ? 0 : kbase;
                int b = 0, N = 0;
                uint32_t a4 = 0;
                if (over) {
                    // tokens at the lane's first packet: one division, double estimate + exact correction
                    const int64_t num = Ci + Ri + (int64_t)k0 * Gi;  // >= 0
                    N = (int)((double)num * (1.0 / (double)Ri));
                    int64_t rem = num - (int64_t)N * Ri;
                    if (rem < 0) { N--; rem += Ri; }
                    if (rem >= Ri) { N++; rem -= Ri; }
                    int ssum = 0, cmax = kLindNone;
#pragma unroll
                    for (int i = 0; i < NP; i++) {
                        int a = 0;
                        if (kbase + i >= 0) {  // arrivals run on past the MI end: harmless
                            rem += Gi;
                            if (rem >= Ri) { rem -= Ri; a = 1; }
                        }
                        a4 |= (uint32_t)a << i;
                        const int sft = a - (int)((m4 >> i) & 1u);
                        cmax = cmax + sft > a ? cmax + sft : a;  // this packet's map after the earlier ones
                        ssum += sft;
                    }
                    int b0 = __builtin_amdgcn_readfirstlane(N);  // lane 0's first packet is packet 0
                    int tot_s, tot_c;
                    lind_exclusive_scan(ssum, cmax, tot_s, tot_c);
                    if constexpr (W > 1) {  // exchange 1: the composite of the wavefronts before this one goes first
                        if (lane == 0) { X->ls[wv] = tot_s; X->lc[wv] = tot_c; if (wv == 0u) X->b0 = b0; }
                        __syncthreads();
                        int ps = 0, pc = kLindNone;
                        for (uint32_t w2 = 0; w2 < wv; w2++) {
                            const int s2 = X->ls[w2], c2 = X->lc[w2];
                            pc = pc + s2 > c2 ? pc + s2 : c2;
                            ps += s2;
                        }
                        cmax = pc + ssum > cmax ? pc + ssum : cmax;
                        ssum += ps;
                        b0 = X->b0;
                    }
                    b = b0 + ssum > cmax ? b0 + ssum : cmax;
                    j_base = N - b;
                } else {
                    // the first packet of the pass meets the threshold test like any other; after it, with the
                    // sender slower than the link (or >= 300 packets of room), a token is always there:
                    // accepted = not lost, accepted before the lane = a prefix popcount
                    acc4 = m4;
                    if (first_lane && !(free_mode || Ci >= 0)) acc4 &= ~(1u << skip);
                    int in_wave = 0;
#pragma unroll
                    for (int i = 0; i < NP; i++) {
                        const uint64_t bm = __ballot((acc4 >> i) & 1u);
                        j_base += (int)count_below(bm);
                        in_wave += (int)__popcll(bm);
                    }
                    if constexpr (W > 1) {  // exchange 1
                        if (lane == 0) X->cnt[wv] = in_wave;
                        __syncthreads();
                        for (uint32_t w2 = 0; w2 < wv; w2++) j_base += X->cnt[w2];
                    }
                }
                // exact base: x = (Q0 + j R - D0 - k0 G) u in integers, one exact conversion
                const int64_t xi = Q0i + (int64_t)j_base * Ri - D0i - (int64_t)k0 * Gi;
                xi_base = xi;
                x_base = (double)xi * u;
                if (regime == 3) {
                    // ---- regime C: decisions and landing sides of the base trajectory, in integers, with their margins
                    int64_t xk = xi;
                    int jr = j_base;
#pragma unroll
                    for (int i = 0; i < NP; i++) {
                        const bool m = (m4 >> i) & 1u;
                        const bool a = m && b > 0;
                        acc4 |= (a ? 1u : 0u) << i;
                        const int64_t slack = (xk + Ri) - Mi3, land = (xk + Ii3) - Bi3, mar = (int64_t)jr + 2;
                        const bool f = m && ((slack >= -mar && slack <= mar) || (a && land >= -mar && land <= mar) ||
                                             (xk - mar <= 0) || (xk + Ri - mar < Bi3 / 2 + 2));
                        flag4 |= (f ? 1u : 0u) << i;
                        up4 |= ((a && (xk + Ii3 >= Bi3)) ? 1u : 0u) << i;
                        if (kbase + i >= 0) {
                            b = (b - (m ? 1 : 0) > 0 ? b - (m ? 1 : 0) : 0) + (int)((a4 >> i) & 1u);
                            xk = (a ? xk + Ri : xk) - Gi;
                            jr += a ? 1 : 0;
                        }
                    }
                    // ---- the parity automaton over the accepted packets: an accept that lands from B up leaves an even
                    // queue (parity 0), one that lands below flips the parity by kappa = (I + cl) mod 2.  A lane's four
                    // positions compose to one map on {0, 1} (bit 0: constant, bit 1: the constant / the flip), the
                    // lanes' maps to an exclusive prefix (six DPP steps), and lane 0 starts from the parity of q.
                    const uint32_t kap = (uint32_t)((Ii3 + cl3) & 1);
                    uint32_t fn = 0;  // identity
#pragma unroll
                    for (int i = 0; i < NP; i++) {
                        if ((acc4 >> i) & 1u) {
                            if ((up4 >> i) & 1u) fn = 1u;          // constant 0
                            else fn ^= kap << 1;                   // flip (of the constant, or of the flip)
                        }
                    }
                    uint32_t pre = fn;
                    auto compose = [](uint32_t first, uint32_t then) -> uint32_t {  // `then` after `first`
                        return (then & 1u) ? then : ((first & 1u) | ((first ^ then) & 2u));
                    };
#pragma unroll
                    for (int o = 1; o < kWave; o <<= 1) {
                        const uint32_t prev = (uint32_t)__shfl_up((int)pre, o);
                        if (lane >= (uint32_t)o) pre = compose(prev, pre);
                    }
                    uint32_t excl = (uint32_t)__shfl_up((int)pre, 1);
                    if (lane == 0) excl = 0;  // identity
                    uint32_t P = (uint32_t)(Q0i & 1);
                    P = (excl & 1u) ? ((excl >> 1) & 1u) : (P ^ ((excl >> 1) & 1u));  // parity of the queue in front of this lane
                    // corrections c' = c - cl of the lane's accepted packets, and their sum
                    int csum = 0;
#pragma unroll
                    for (int i = 0; i < NP; i++) {
                        if ((acc4 >> i) & 1u) {
                            int c;
                            if ((up4 >> i) & 1u) { c = (int)((P + (uint32_t)Ii3) & 1u); P = 0; }
                            else { c = cl3; P ^= kap; }
                            cp4 |= (uint32_t)((c - cl3) + 1) << (2 * i);  // 0, 1, 2 = -1, 0, +1
                            csum += c - cl3;
                        } else {
                            cp4 |= 1u << (2 * i);
                        }
                    }
                    int cincl = csum;
#pragma unroll
                    for (int o = 1; o < kWave; o <<= 1) {
                        const int prev = __shfl_up(cincl, o);
                        if (lane >= (uint32_t)o) cincl += prev;
                    }
                    c_before = cincl - csum;  // corrections accumulated in front of this lane
                } else {
                double x = x_base;
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    const bool m = (m4 >> i) & 1u;
                    bool a;
                    if (over) {
                        a = m && b > 0;
                        acc4 |= (a ? 1u : 0u) << i;
                    } else {
                        a = (acc4 >> i) & 1u;
                    }
                    const double sx = x + R;  // the queue behind this packet if it is accepted (ns:82)
                    const uint32_t es = exponent_bits(sx);
                    const bool f = m && (!(x > 0.0) || es < e || (es > e && maxq_above));
                    flag4 |= (f ? 1u : 0u) << i;
                    if (kbase + i >= 0) {
                        if (over) b = (b - (m ? 1 : 0) > 0 ? b - (m ? 1 : 0) : 0) + (int)((a4 >> i) & 1u);
                        x = (a ? sx : x) - G;  // exact: multiples of u below 2^(e+1)
                    }
                }
                }
            }
            // ---- the pass stops at the first position that is past the MI end or breaks a precondition
            uint32_t stop4 = (~ex4 | flag4) & kPosMask;
            if (first_lane) stop4 &= ~((1u << skip) - 1u);  // positions before `skip` are not part of the pass
            const uint64_t stop_lanes = __ballot(stop4 != 0u);
            uint32_t p_stop = kWavePos, j_stop;  // (in this wavefront's positions)
            bool stopped_by_flag = false;
            if (stop_lanes) {
                const uint32_t ls = (uint32_t)__ffsll((unsigned long long)stop_lanes) - 1u;
                const uint32_t is = ((uint32_t)__ffs((int)stop4) - 1u) & (kLanePos - 1u);
                p_stop = kLanePos * ls + rl_u32(is, ls);
                j_stop = rl_u32((uint32_t)j_base + (uint32_t)__popc(acc4 & ((1u << is) - 1u)), ls);
                stopped_by_flag = rl_u32((flag4 >> is) & 1u, ls) != 0u;
            } else {
                j_stop = rl_u32((uint32_t)j_base + (uint32_t)__popc(acc4), kWave - 1u);
            }
            if constexpr (W > 1) {  // exchange 2: the first wavefront with a stop ends the team's pass
                if (lane == 0) { X->pstop[wv] = p_stop; X->jstop[wv] = j_stop; X->sflag[wv] = stopped_by_flag ? 1u : 0u; }
                __syncthreads();
                uint32_t w2 = 0;
                while (w2 + 1u < (uint32_t)W && X->pstop[w2] == kWavePos) w2++;
                p_stop = w2 * kWavePos + X->pstop[w2];
                j_stop = X->jstop[w2];
                stopped_by_flag = X->sflag[w2] != 0u;
            }
            const uint32_t ncommit = p_stop - skip;
            // q hovering around a power of two (or a queue that keeps running empty) breaks a pass after a
            // few packets every time: send the next stretch by the accept chain, which has no such
            // precondition, then try again
            // ... unless regime C has not had its chance yet (regime B stopped at the edge of its binade: the next pass
            // tries the two-binade form)
            if (stopped_by_flag && ncommit < 32u) chain_left = 4u;
#if PCC_ADAPTIVE_C
            if (stopped_by_flag && regime == 2) { c_armed = true; chain_left = 0u; }
#endif
            if (ncommit) {
                if (TRACE && (int64_t)((uint64_t)st.a + st.d + ncommit) > D.trace_stride) st.flags |= PCC_FLAG_TRACE_OVERRUN;
                // ---- records, in send order per ring (the lane replays its positions); the link state
                // behind the last packet that reached the queue
                double last_q = 0.0, last_t = 0.0;
                bool have_last = false;
                double x = x_base;
                int64_t xt = xi_base + c_before;  // regime C: the true queue in units of v
                uint32_t j = (uint32_t)j_base;
                // STAGE: what this wavefront's first committed position has in front of it in the pass: accepted, dropped
                uint32_t acc0 = 0, drp0 = 0, n_acc_w = 0, n_com_w = 0;
                if constexpr (STAGE) {
                    acc0 = rl_u32((uint32_t)j_base, 0);
                    const uint32_t p0 = W > 1 ? wv * kWavePos : 0u, pv = p0 > skip ? p0 : skip;   // (this wavefront's first position)
                    drp0 = (pv - skip) - acc0;
                }
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    const uint32_t p = kLanePos * glane + (uint32_t)i;
                    const bool a = (acc4 >> i) & 1u;
                    const int cpr = (int)((cp4 >> (2 * i)) & 3u) - 1;  // regime C: this packet's correction c' (0 elsewhere)
                    if (regime == 3) x = (double)xt * u;  // exact: even from B up
                    if constexpr (STAGE) {
                        const bool com = p >= skip && p < p_stop;
                        n_com_w += (uint32_t)__popcll(__ballot(com));
                        n_acc_w += (uint32_t)__popcll(__ballot(com && a));
                    }
                    if (p >= skip && p < p_stop) {
                        const uint32_t kk = p - skip;             // packets of the pass before this one
                        const double tki = t0 + (double)kk * G;   // exact
                        const double qc = regime == 1 ? 0.0 : max0(x);  // ns:66-67
                        double2 rec;
                        rec.y = dl + qc;                          // ns:170
                        rec.x = tki + rec.y;                      // ns:174
                        if constexpr (STAGE) {
                            stage[a ? j - acc0 : (kWavePos - 1u) - ((kk - j) - drp0)] = rec;
                        } else {
                            const uint32_t off = a ? (((st.a + j) << 4) & mask_b) : cap_b + (((st.d + (kk - j)) << 4) & dmask_b);
                            st_rec_pass(reinterpret_cast<double2 *>(base + off), rec);
                        }
                        if ((m4 >> i) & 1u) {
                            have_last = true;
                            last_t = tki;
                            last_q = regime == 1 ? ebw + 0.0 : (a ? x + R : x);  // ns:75-82
                            if (regime == 3) last_q = (double)(a ? xt + Ri + cpr : xt) * u;  // = fl(qcur + 1/bw), exactly
                        }
                    }
                    if (kbase + i >= 0) {
                        x = (a ? x + R : x) - G;
                        xt = (a ? xt + Ri + cpr : xt) - Gi;
                        j += a ? 1u : 0u;
                    }
                }
                if constexpr (STAGE) {
                    // (what the lanes staged is read by other lanes of this wavefront: LDS accesses of one wavefront execute in
                    // order; the barriers keep the compiler from moving them across)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const uint32_t a0 = st.a + acc0, d0 = st.d + drp0, n_drp_w = n_com_w - n_acc_w;
                    for (uint32_t sl = lane; sl < n_acc_w; sl += kWave)
                        st_rec_run(reinterpret_cast<double2 *>(base + (((a0 + sl) << 4) & mask_b)), stage[sl]);
                    for (uint32_t sl = lane; sl < n_drp_w; sl += kWave)
                        st_rec_run(reinterpret_cast<double2 *>(base + cap_b + (((d0 + sl) << 4) & dmask_b)), stage[(kWavePos - 1u) - sl]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();   // (the next pass stages into the same slots)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                const uint64_t lm = __ballot(have_last);
                if constexpr (W == 1) {
                    if (lm) {
                        const uint32_t ll = 63u - (uint32_t)__clzll((long long)lm);
                        st.q = rl_f64(last_q, ll);
                        st.tu = rl_f64(last_t, ll);
                    }
                } else {  // exchange 3: the last wavefront that committed a packet which reached the queue
                    const uint32_t ll = lm ? 63u - (uint32_t)__clzll((long long)lm) : 0u;
                    const double wq = rl_f64(last_q, ll), wt = rl_f64(last_t, ll);
                    if (lane == 0) { X->has_last[wv] = lm ? 1u : 0u; X->last_q[wv] = wq; X->last_t[wv] = wt; }
                    __syncthreads();
                    for (int w2 = W - 1; w2 >= 0; w2--)
                        if (X->has_last[w2]) { st.q = X->last_q[w2]; st.tu = X->last_t[w2]; break; }
                }
                if (prof_counters(D) && lane == 0 && writer) {
                    const int c = regime == 1 ? 0 : (regime == 3 || (!free_mode && Gi < Ri)) ? 1 : 2;
                    atomicAdd(&D.pass_stats[c], 1ull);
                    atomicAdd(&D.pass_stats[4 + c], (unsigned long long)ncommit);
                    atomicAdd(&D.pass_stats[13], (unsigned long long)(__builtin_readcyclecounter() - dbg_c0));
                }
                st.t = (t0 + (double)(ncommit - 1u) * G) + gap;  // ns:161 on the last packet's (exact) send time
                st.a += j_stop;
                st.d += ncommit - j_stop;
                st.sent += ncommit;
                if (kProfile) st.prof_closed++;
                serial_len = 8;
                continue;
            }
            if (prof_counters(D) && lane == 0 && writer) atomicAdd(&D.pass_stats[8], 1ull);  // nothing to commit: first packet flagged
        } else if (prof_counters(D) && lane == 0 && writer) {
            atomicAdd(&D.pass_stats[ok_t ? 10 : 9], 1ull);
        }
        // ---- serial pass: up to serial_len packets with the plain recurrence, wave-uniform (every lane
        // computes the same values; lane k keeps packet k's record), exact with no precondition
        {
            bool rnd;
            if (TRACE) {
                const uint64_t pos = (uint64_t)st.a + st.d + lane;
                double uu = 1.0;
                if ((int64_t)pos < D.trace_stride) uu = trace[pos];
                rnd = uu < lr;
            } else {
                const uint32_t jp = st.sent + lane;
                uint32_t w[4];
                philox4x32_10(jp >> 2, mi, episode, gid, D.key0, D.key1, w);
                const uint32_t xw = (jp & 3u) == 0 ? w[0] : (jp & 3u) == 1 ? w[1] : (jp & 3u) == 2 ? w[2] : w[3];
                rnd = always || xw < thr;
            }
            const uint64_t rmask = __ballot(rnd);
            double my_t = 0.0, my_lat = 0.0;
            bool my_drop = true;
            uint32_t nv;
            if (chain_left) chain_left--;
            const double tend64 = t0 + 64.0 * G;
            const bool ok_chain = (t2s - t1s == G) && (G > 0.0) && (t0 >= 128.0 * gap) &&
                                  (exponent_bits(t0) == exponent_bits(tend64)) && (st.tu >= maxq) &&
                                  (st.tu + st.tu >= tend64);
            if (ok_chain) {
                // ---- accept-to-accept pass over 64 packets, one per lane.  With t, tu >= maxq the drain
                // q - (t - tu) is exact, so between two ACCEPTED packets the queue seen by packet k is
                // max(0, q_m - (t_k - t_m)) whatever tail drops and random losses lie in between, and the
                // tail-drop test is monotone in k.  Phase 1 is the chain from one accepted packet to the
                // next (ballot of "not lost, not full", first set lane, readlanes), every floating-point
                // step the reference's own; phase 2 lets every lane finish its packet from the state its
                // segment started with.
                const double tk = t0 + (double)lane * G;
                const bool vk = tk < end;
                const uint64_t vmask = __ballot(vk);
                nv = (uint32_t)__popcll(vmask);  // valid lanes are a prefix (tk increases)
                double qm = st.q, tm = st.tu;
                uint64_t open = vmask & ~rmask;          // lanes that can still be the next accepted packet
                uint64_t amask = 0;                      // accepted lanes
                uint32_t na = 0;
                double seg_q = 0.0, seg_t = 0.0;         // lane j: link state after the j-th accepted packet
                while (open) {
                    const double qc = max0(qm - (tk - tm));  // queue seen by packet k if nothing was accepted since tm
                    const bool full = ebw + qc > maxq;       // monotone non-increasing in k
                    const uint64_t cm = open & ~__ballot(full);
                    if (!cm) break;
                    const uint32_t ks = (uint32_t)__ffsll((unsigned long long)cm) - 1u;
                    qm = rl_f64(ebw + qc, ks);               // ns:82
                    tm = rl_f64(tk, ks);                     // ns:76
                    if (lane == na) { seg_q = qm; seg_t = tm; }
                    na++;
                    amask |= 1ull << ks;
                    open &= ~((2ull << ks) - 1ull);          // lanes after ks
                }
                const uint32_t seg = (uint32_t)count_below(amask);
                const int src = seg ? (int)seg - 1 : 0;
                double q_seg = __shfl(seg_q, src);
                double t_seg = __shfl(seg_t, src);
                if (!seg) { q_seg = st.q; t_seg = st.tu; }
                const double qc = max0(q_seg - (tk - t_seg));
                my_lat = dl + qc;                            // ns:170
                my_drop = !((amask >> lane) & 1ull);
                const double my_q_after = my_drop ? qc : ebw + qc;  // link state this packet leaves unless a random loss
                my_t = tk + my_lat;                          // ns:174
                const uint64_t touch = vmask & ~rmask;
                if (touch) {
                    const uint32_t kl = 63u - (uint32_t)__clzll((long long)touch);
                    st.q = rl_f64(my_q_after, kl);
                    st.tu = rl_f64(tk, kl);
                }
                if (nv) st.t = (t0 + (double)(nv - 1u) * G) + gap;  // ns:161 on the last packet's (exact) send time
            } else {
                double t = t0;
                // packets certainly before `end` (two of margin for the rounding of t += gap) run under a
                // scalar loop counter; the rest with the exit test, kept scalar through readfirstlane
                const double ahead = (end - t0) / gap - 2.0;
                uint32_t nsafe = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)(ahead >= 64.0 ? 64u : (ahead > 0.0 ? (uint32_t)ahead : 0u)));
                if (nsafe > serial_len) nsafe = serial_len;
                uint32_t k = 0;
                for (; k < nsafe; k++) {
                    bool dropped;
                    const double2 rec = link_send(t, (rmask >> k) & 1ull, dl, maxq, ebw, st.q, st.tu, dropped);
                    if (lane == k) { my_t = rec.x; my_lat = rec.y; my_drop = dropped; }
                    t += gap;  // ns:161
                }
                for (; k < serial_len && __builtin_amdgcn_readfirstlane((int)(t < end)); k++) {
                    bool dropped;
                    const double2 rec = link_send(t, (rmask >> k) & 1ull, dl, maxq, ebw, st.q, st.tu, dropped);
                    if (lane == k) { my_t = rec.x; my_lat = rec.y; my_drop = dropped; }
                    t += gap;  // ns:161
                }
                nv = k;
                st.t = t;
                if (serial_len < 64u) serial_len *= 2u;
            }
            const bool valid = lane < nv;
            if (TRACE && (int64_t)((uint64_t)st.a + st.d + nv) > D.trace_stride) st.flags |= PCC_FLAG_TRACE_OVERRUN;
            const uint64_t dm = __ballot(valid && my_drop), am = __ballot(valid && !my_drop);
            if (valid && writer) {
                double2 rec;
                rec.x = my_t;
                rec.y = my_lat;
                const uint32_t off = my_drop ? cap_b + (((st.d + (uint32_t)count_below(dm)) << 4) & dmask_b)
                                             : (((st.a + (uint32_t)count_below(am)) << 4) & mask_b);
                st_rec_pass(reinterpret_cast<double2 *>(base + off), rec);
            }
            st.a += (uint32_t)__popcll(am);
            st.d += (uint32_t)__popcll(dm);
            st.sent += nv;
            if (kProfile) st.prof_other += ok_chain ? 1u : 0x10000u;  // (chain passes low, serial passes high)
            if (prof_counters(D) && lane == 0 && writer) {
                atomicAdd(&D.pass_stats[3], 1ull);
                atomicAdd(&D.pass_stats[7], (unsigned long long)nv);
                atomicAdd(&D.pass_stats[14], (unsigned long long)(__builtin_readcyclecounter() - dbg_c0));
            }
        }
    }
}

// Two senders on the shared link, one env, all 64 lanes.  Same exactness argument as heavy_mi; the
// 64 packets of a pass are the first 64 of the (time, sender id) merge of the two senders'
// arithmetic send sequences, found per lane by a merge-path search.
struct SendState2 {
    double q, tu, t[2];
    uint32_t a[2], d[2], sent[2], flags;
    uint32_t prof_closed, prof_other;  // profile build: token passes / chain passes (low half) and plain-recurrence passes (high half)
    uint32_t prof_why;                 // profile build: which preconditions refused a token pass (bits 0-10) or the chain (16-20)
};

// STAGE (round 6): the records of a 256-position pass leave through `stage` (256 LDS slots of this wavefront) as four runs
// in ring order -- sender 0 accepted, sender 0 dropped, sender 1 accepted, sender 1 dropped -- and lane l stores slot 64 k + l of
// a run: whole ring lines per instruction, nontemporal (st_rec_run).
template <bool TRACE, bool STAGE = false>
__device__ __forceinline__ void heavy_mi2(const Dev &D, uint32_t lane, double dl, double lr, uint32_t thr, bool always,
                                          double maxq, double ebw, double gap0, double gap1, double end,
                                          uint32_t episode, uint32_t mi, uint32_t gid, const double *trace, char *base0,
                                          char *base1, uint32_t cap0, uint32_t cap1, SendState2 &st, double2 *stage = nullptr) {
    const uint32_t caps[2] = {cap0, cap1};
    const double gap[2] = {gap0, gap1};
    uint32_t chain_left = 0;  // passes to send by the accept chain before the token pass is tried again
    uint32_t guard = 0;
    while ((st.t[0] < st.t[1] ? st.t[0] : st.t[1]) < end) {
        if (++guard > kMaxPasses) { st.flags |= PCC_FLAG_INTERNAL; break; }
        const uint64_t dbg_c0 = prof_counters(D) ? __builtin_readcyclecounter() : 0;
        uint32_t dbg_sweeps256 = 0;
        // ---- token pass, up to 256 packets: the queue stays backlogged in one binade (heavy_mi's regime B, here for the
        // merged stream; lane l owns the positions 4 l .. 4 l + 3 = one Philox block).  Every quantity is a multiple of
        // u = ulp(q): the queue in front of merged position p is x_p = Q0 + j_p R - D_p (j_p packets accepted before it,
        // D_p = t_p - tu); it is accepted iff it is not lost at random and x_p + R <= maxq, i.e. iff tokens are left:
        // b_p = N_p - j_p > 0 with N_p = floor((M - Q0 + D_p) / R).  b_{p+1} = max(b_p - m_p, 0) + (N_{p+1} - N_p) is a Lindley
        // map per position -- uneven token arrivals, because the two senders' send times interleave unevenly -- and the maps
        // compose by one prefix scan (lind_exclusive_scan).  A position whose queue runs empty or leaves the binade ends the
        // pass in front of it; the accept chain below (no such precondition, 64 packets) takes over from there.
        {
            constexpr uint32_t kPass = 4u * kWave;
            double G2[2];
            bool okb = true;
            double tend_max = 0.0, lim = end;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const double t0 = st.t[s], t1s = t0 + gap[s];
                G2[s] = t1s - t0;
                const double t2s = t1s + gap[s], tend = t0 + (double)kPass * G2[s];
                // (t0 + c G is exact for c <= 256 while it stays in t0's binade: positions whose send time leaves the binade of
                // either sender are not part of the pass, which ends at lim = min(end, top of the lower binade) -- like heavy_mi)
                const double ttop = pow2_f64((int)exponent_bits(t0) - 1022);
                lim = ttop < lim ? ttop : lim;
                okb = okb && (t2s - t1s == G2[s]) && (G2[s] > 0.0) && (t0 >= ((double)kPass + 4.0) * gap[s]) &&
                      (exponent_bits(t0) == exponent_bits(t2s));
                tend_max = tend > tend_max ? tend : tend_max;
            }
            const uint32_t e = exponent_bits(st.q), eb = exponent_bits(ebw);
            const double T0 = st.t[0] <= st.t[1] ? st.t[0] : st.t[1];
            const double x0 = st.q - (T0 - st.tu);
            // ---- sweep pass (below): the same 256 positions decided by the plain recurrence, lane after lane.  It evaluates the
            // reference's expressions on the reference's operands (ns:66-82), so all it needs is the positions' send times
            // (the conditions above) -- not exact drains, not the queue in one binade.  Taken for the passes after a token
            // pass that stopped early, and whenever the token pass is refused.
            const bool oks = kRelaxSweeps != 0 && okb;
            okb = okb && (st.tu + st.tu >= tend_max) && chain_left == 0u && (st.q > 0.0) && e > 64u && e < 1100u && (x0 > 0.0) && eb <= e &&
                  exponent_bits(st.tu) >= e && exponent_bits(maxq) >= e;
            if (chain_left) chain_left--;
            if (kProfile && !okb) {
                uint32_t why = 0;
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const double t0 = st.t[s], t1s = t0 + gap[s], g = t1s - t0, t2s = t1s + gap[s];
                    why |= (t2s - t1s == g ? 0u : 1u) | (t0 >= ((double)kPass + 4.0) * gap[s] ? 0u : 2u) |
                           (exponent_bits(t0) == exponent_bits(t2s) ? 0u : 4u);
                }
                why |= (st.tu + st.tu >= tend_max ? 0u : 8u) | (st.q > 0.0 ? 0u : 16u) | ((e > 64u && e < 1100u) ? 0u : 32u) |
                       (x0 > 0.0 ? 0u : 64u) | (eb <= e ? 0u : 128u) | (exponent_bits(st.tu) >= e ? 0u : 256u) |
                       (exponent_bits(maxq) >= e ? 0u : 512u);
                st.prof_why |= why;
            }
            double u = 0.0, R = 0.0;
            int64_t Q0i = 0, Ri = 1, Mi = 0, Dsi[2] = {0, 0}, Gsi[2] = {0, 0};
            bool free_mode = false, maxq_above = false;
            if (okb) {
                u = pow2_f64((int)e - 1023 - 52);
                const double inv_u = pow2_f64(-((int)e - 1023 - 52));
                const double probe = pow2_f64((int)e - 1023);
                R = (eb == e) ? ebw : (probe + ebw) - probe;   // 1/bw on the grid of u (ns:82 rounds x + 1/bw to it)
                const double err = ebw - R;
                const bool tie = fabs(err) == 0.5 * u;
                okb = (tend_max - st.tu) * inv_u < 4.0e18 && R > 0.0;
                if (okb) {
                    Q0i = (int64_t)(st.q * inv_u);
                    Ri = (int64_t)(R * inv_u);
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        Dsi[s] = (int64_t)((st.t[s] - st.tu) * inv_u);   // exact: tu <= t <= 2 tu, multiples of u
                        Gsi[s] = (int64_t)(G2[s] * inv_u);
                    }
                    const double room = ((maxq - R) - x0) / R;   // packets of room in the queue (estimate)
                    free_mode = room >= (double)kPass + 44.0;     // nothing of this pass can be tail-dropped
                    Mi = free_mode ? 0 : (int64_t)(maxq * inv_u);
                    // a tie rounds to even: x + R holds only while every x is an even multiple of u
                    if (tie && ((Q0i | Dsi[0] | Dsi[1] | Gsi[0] | Gsi[1]) & 1)) okb = false;
                    maxq_above = exponent_bits(maxq) > e;
                }
            }
            if (kProfile && !okb && !(st.prof_why & 0x3FFu)) st.prof_why |= 1024u;   // (the grid of u: a tie on odd multiples, a span too wide)
            if (okb || oks) {
                const uint32_t sent_all = st.sent[0] + st.sent[1];
                const uint32_t skip = sent_all & 3u;   // positions of lane 0's Philox block that were sent before this pass
                const int kbase = 4 * (int)lane - (int)skip;   // packet index (within the pass) of this lane's position 0
                // ---- loss decisions of the lane's four positions
                uint32_t rnd4 = 0;
                if (TRACE) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int k = kbase + i;
                        const int64_t pos = (int64_t)((uint64_t)st.a[0] + st.d[0] + st.a[1] + st.d[1]) + k;
                        double uu = 1.0;
                        if (k >= 0 && pos < D.trace_stride) uu = trace[pos];
                        rnd4 |= (uu < lr ? 1u : 0u) << i;
                    }
                } else {
                    uint32_t w[4];
                    philox4x32_10((sent_all >> 2) + lane, mi, episode, gid, D.key0, D.key1, w);
#pragma unroll
                    for (int i = 0; i < 4; i++) rnd4 |= ((always || w[i] < thr) ? 1u : 0u) << i;
                }
                // ---- merge path: c0 = how many of sender 0's packets precede the lane's first packet (sender 0 first on
                // equal times): smallest c with B[kf - c - 1] < A[c]; then the lane's packets one by one
                const uint32_t kf = kbase < 0 ? 0u : (uint32_t)kbase;
                uint32_t lo = 0, hi = kf;
                while (__ballot(lo < hi)) {
                    if (lo < hi) {
                        const uint32_t c = (lo + hi) >> 1;
                        const double Ac = st.t[0] + (double)c * G2[0];
                        const double Bp = st.t[1] + (double)(kf - c - 1u) * G2[1];
                        if (Bp < Ac) hi = c;
                        else lo = c + 1u;
                    }
                }
                uint32_t c0 = lo, c1 = kf - lo;
                uint32_t s4 = 0, ex4 = 0, m4 = 0;   // bit i: sender of position i; it holds a packet of this MI; ... that reaches the queue
                double tk[4];
                int64_t Dp[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double A = st.t[0] + (double)c0 * G2[0], B = st.t[1] + (double)c1 * G2[1];
                    const bool is1 = !(A <= B);
                    tk[i] = is1 ? B : A;
                    Dp[i] = is1 ? Dsi[1] + (int64_t)c1 * Gsi[1] : Dsi[0] + (int64_t)c0 * Gsi[0];
                    const bool there = kbase + i >= 0;
                    const bool ex = there && tk[i] < lim;
                    s4 |= (is1 ? 1u : 0u) << i;
                    ex4 |= (ex ? 1u : 0u) << i;
                    m4 |= ((ex && !((rnd4 >> i) & 1u)) ? 1u : 0u) << i;
                    if (there) { c0 += is1 ? 0u : 1u; c1 += is1 ? 1u : 0u; }   // (positions before `skip` all stand for the first packet)
                }
                // ---- accept decisions
                uint32_t acc4 = 0;
                int jb = 0;   // packets accepted before the lane's first position
                double xq[4] = {0.0, 0.0, 0.0, 0.0};   // the queue in front of every position (ns:66-67 before max0)
                double sw_q = 0.0, sw_t = 0.0;         // sweep pass: the link state behind the lane's four positions
                uint32_t flag4 = 0;
                if (!okb) {
                    // ---- sweep pass.  Every lane holds the link state behind its four positions; one sweep = the lane takes
                    // the state behind the lane below it (lane 0: the state the pass starts from), sends its positions one
                    // after the other (ns:66-82: a packet lost at random or not of this pass hands the state on, one refused by
                    // the full queue leaves it drained to its send time, an accepted one adds 1/bw) and keeps what is behind them.  After sweep i the lanes 0 .. i-1 are final, so 64 sweeps are
                    // the serial recurrence bit for bit; a sweep that moves no lane's state has reached that fixed point early.
                    // The first guess is "the lane below left the queue drained", which is right wherever a busy period ends
                    // inside a lane: sweep i then settles the busy periods that span i lanes, all of them at once -- a link
                    // that keeps running empty (what stops the token pass) takes two or three sweeps for its 256 packets.
                    double iq = 0.0, it = st.tu;
                    for (uint32_t sweep = 0; sweep <= kWave; sweep++) {
                        if (sweep) { iq = wave_shr1_f64(sw_q, st.q); it = wave_shr1_f64(sw_t, st.tu); }
                        double q = iq, t = it;
                        uint32_t a4n = 0;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const double qc = max0(q - (tk[i] - t));
                            const bool m = (m4 >> i) & 1u;
                            const bool a = m && !(ebw + qc > maxq);
                            xq[i] = qc;
                            a4n |= (a ? 1u : 0u) << i;
                            q = m ? (a ? ebw + qc : qc) : q;   // ns:75-82
                            t = m ? tk[i] : t;                 // ns:76
                        }
                        const bool moved = __double_as_longlong(q) != __double_as_longlong(sw_q) ||
                                           __double_as_longlong(t) != __double_as_longlong(sw_t) || sweep == 0u;
                        sw_q = q;
                        sw_t = t;
                        acc4 = a4n;
                        if (kProfile) dbg_sweeps256++;
                        if (sweep && !__ballot(moved)) break;
                    }
                } else if (free_mode) {
                    acc4 = m4;
#pragma unroll
                    for (int i = 0; i < 4; i++) jb += (int)count_below(__ballot((acc4 >> i) & 1u));
                } else {
                    int N[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {   // tokens up to each position: a division, double estimate + exact correction
                        const int64_t num = (Mi - Q0i) + Dp[i];   // >= 0
                        int n = (int)((double)num * (1.0 / (double)Ri));
                        int64_t rem = num - (int64_t)n * Ri;
                        if (rem < 0) { n--; rem += Ri; }
                        if (rem >= Ri) { n++; }
                        N[i] = n;
                    }
                    int Nnext = __shfl_down(N[0], 1);
                    if (lane == kWave - 1u) Nnext = N[3];
                    int ssum = 0, cmax = kLindNone;
                    int a_[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        a_[i] = (i < 3 ? N[i + 1 < 4 ? i + 1 : 3] : Nnext) - N[i];   // tokens that arrive before the next position
                        const int sft = a_[i] - (int)((m4 >> i) & 1u);
                        cmax = cmax + sft > a_[i] ? cmax + sft : a_[i];
                        ssum += sft;
                    }
                    const int b0 = __builtin_amdgcn_readfirstlane(N[0]);
                    int tot_s, tot_c;
                    lind_exclusive_scan(ssum, cmax, tot_s, tot_c);
                    int b = b0 + ssum > cmax ? b0 + ssum : cmax;
                    jb = N[0] - b;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool m = (m4 >> i) & 1u;
                        acc4 |= ((m && b > 0) ? 1u : 0u) << i;
                        b = (b - (m ? 1 : 0) > 0 ? b - (m ? 1 : 0) : 0) + a_[i];
                    }
                }
                // ---- the queue in front of every position, exactly; positions that break a precondition
                if (okb) {
                    int j = jb;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int64_t xi = Q0i + (int64_t)j * Ri - Dp[i];
                        const double x = (double)xi * u;   // exact
                        xq[i] = x;
                        const double sx = x + R;           // the queue behind this packet if it is accepted (ns:82)
                        const uint32_t es = exponent_bits(sx);
                        const bool m = (m4 >> i) & 1u;
                        flag4 |= ((m && (!(x > 0.0) || es < e || (es > e && maxq_above))) ? 1u : 0u) << i;
                        j += (int)((acc4 >> i) & 1u);
                    }
                }
                // ---- the pass stops at the first position that is past the MI end or breaks a precondition
                uint32_t stop4 = (~ex4 | flag4) & 0xFu;
                if (lane == 0) stop4 &= ~((1u << skip) - 1u);   // positions before `skip` are not part of the pass
                const uint64_t stop_lanes = __ballot(stop4 != 0u);
                uint32_t p_stop = kPass;
                if (stop_lanes) {
                    const uint32_t ls = (uint32_t)__ffsll((unsigned long long)stop_lanes) - 1u;
                    p_stop = 4u * ls + rl_u32(((uint32_t)__ffs((int)stop4) - 1u) & 3u, ls);
                }
                const uint32_t ncommit = p_stop - skip;
                // q hovers around a binade edge or keeps running empty: the next passes by sweeps
                if (okb && p_stop < kPass && ncommit < 32u) chain_left = oks ? 3u : 2u;
                if (kProfile) st.prof_closed += okb ? 1u : 0x100u;   // (token passes in the low byte, sweep passes above)
                if (prof_counters(D) && lane == 0 && !okb) {   // (profile build: sweep passes / their packets / sweeps / cycles)
                    atomicAdd(&D.pass_stats[2], 1ull);
                    atomicAdd(&D.pass_stats[6], (unsigned long long)ncommit);
                    atomicAdd(&D.pass_stats[8], (unsigned long long)dbg_sweeps256);
                    atomicAdd(&D.pass_stats[14], (unsigned long long)(__builtin_readcyclecounter() - dbg_c0));
                }
                if (prof_counters(D) && lane == 0 && okb) {   // (profile build: token passes / their packets / their cycles)
                    atomicAdd(&D.pass_stats[1], 1ull);
                    atomicAdd(&D.pass_stats[5], (unsigned long long)ncommit);
                    atomicAdd(&D.pass_stats[13], (unsigned long long)(__builtin_readcyclecounter() - dbg_c0));
                    if (ncommit < 32u) {   // ... and the short ones among them
                        atomicAdd(&D.pass_stats[10], 1ull);
                        atomicAdd(&D.pass_stats[11], (unsigned long long)ncommit);
                        atomicAdd(&D.pass_stats[12], (unsigned long long)(__builtin_readcyclecounter() - dbg_c0));
                    }
                }
                if (ncommit) {
                    if (TRACE && (int64_t)((uint64_t)st.a[0] + st.d[0] + st.a[1] + st.d[1] + ncommit) > D.trace_stride)
                        st.flags |= PCC_FLAG_TRACE_OVERRUN;
                    // ---- records: four dense runs (sender x accepted / dropped).  The lane's counts of each kind, 16 bits each
                    // in one 64-bit word, and their exclusive prefix over the lanes
                    uint32_t in4 = 0;
                    unsigned long long cnt = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t pp = 4u * lane + (uint32_t)i;
                        if (pp >= skip && pp < p_stop) {
                            in4 |= 1u << i;
                            cnt += 1ull << (16u * (2u * ((s4 >> i) & 1u) + (((acc4 >> i) & 1u) ? 0u : 1u)));
                        }
                    }
                    unsigned long long incl = cnt;
#pragma unroll
                    for (int o = 1; o < kWave; o <<= 1) {
                        const unsigned long long up = (unsigned long long)__shfl_up((long long)incl, o);
                        if (lane >= (uint32_t)o) incl += up;
                    }
                    const unsigned long long total = rl_u64(incl, kWave - 1u);
                    unsigned long long before = incl - cnt;
                    const uint32_t a0n = (uint32_t)(total & 0xFFFFu), d0n = (uint32_t)((total >> 16) & 0xFFFFu);
                    const uint32_t a1n = (uint32_t)((total >> 32) & 0xFFFFu), d1n = (uint32_t)((total >> 48) & 0xFFFFu);
                    double last_q = 0.0, last_t = 0.0;
                    bool have_last = false;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool a = (acc4 >> i) & 1u;
                        if ((in4 >> i) & 1u) {
                            const bool sdr = (s4 >> i) & 1u;
                            const uint32_t kind = 2u * (sdr ? 1u : 0u) + (a ? 0u : 1u);
                            const uint32_t idx = (uint32_t)(before >> (16u * kind)) & 0xFFFFu;
                            before += 1ull << (16u * kind);
                            const double x = xq[i];
                            double2 rec;
                            rec.y = dl + max0(x);         // ns:66-67, 170
                            rec.x = tk[i] + rec.y;        // ns:174
                            if constexpr (STAGE) {
                                // the run's first slot: the runs follow each other in the order of `kind`
                                const uint32_t run0 = sdr ? (a ? a0n + d0n : a0n + d0n + a1n) : (a ? 0u : a0n);
                                stage[run0 + idx] = rec;
                            } else {
                                const uint32_t cp = sdr ? cap1 : cap0;
                                const uint32_t off = a ? ((((sdr ? st.a[1] : st.a[0]) + idx) << 4) & ((cp - 1u) << 4))
                                                       : (cp << 4) + ((((sdr ? st.d[1] : st.d[0]) + idx) << 4) & ((2u * cp - 1u) << 4));
                                st_rec_pass(reinterpret_cast<double2 *>((sdr ? base1 : base0) + off), rec);
                            }
                            if ((m4 >> i) & 1u) { have_last = true; last_t = tk[i]; last_q = a ? x + (okb ? R : ebw) : x; }   // ns:75-82
                        }
                    }
                    const uint64_t lm = __ballot(have_last);
                    if (lm) {   // the link state behind the last committed packet that reached the queue
                        const uint32_t ll = 63u - (uint32_t)__clzll((long long)lm);
                        st.q = rl_f64(last_q, ll);
                        st.tu = rl_f64(last_t, ll);
                    }
                    if constexpr (STAGE) {
                        // (what the lanes staged is read by other lanes of this wavefront: LDS accesses of one wavefront execute in
                        // order; the barriers keep the compiler from moving them across)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                        for (int s = 0; s < 2; s++) {
                            char *const bs = s ? base1 : base0;
                            const uint32_t cp = s ? cap1 : cap0, an = s ? a1n : a0n, dn = s ? d1n : d0n;
                            const uint32_t ra = s ? a0n + d0n : 0u, rd = ra + an;   // the two runs' first slots
                            const uint32_t mask_b = (cp - 1u) << 4, dmask_b = (2u * cp - 1u) << 4, cap_b = cp << 4;
                            for (uint32_t sl = lane; sl < an; sl += kWave)
                                st_rec_run(reinterpret_cast<double2 *>(bs + (((st.a[s] + sl) << 4) & mask_b)), stage[ra + sl]);
                            for (uint32_t sl = lane; sl < dn; sl += kWave)
                                st_rec_run(reinterpret_cast<double2 *>(bs + cap_b + (((st.d[s] + sl) << 4) & dmask_b)), stage[rd + sl]);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();   // (the next pass stages into the same slots)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                    // ns:161 on each sender's last packet's (exact) send time: the next one may be the first of a new binade
                    if (a0n + d0n) st.t[0] = (st.t[0] + (double)(a0n + d0n - 1u) * G2[0]) + gap[0];
                    if (a1n + d1n) st.t[1] = (st.t[1] + (double)(a1n + d1n - 1u) * G2[1]) + gap[1];
                    st.a[0] += a0n; st.d[0] += d0n; st.sent[0] += a0n + d0n;
                    st.a[1] += a1n; st.d[1] += d1n; st.sent[1] += a1n + d1n;
                    continue;
                }
            }
        }
        // ---- loss decisions of the next 64 packets of the merged stream
        uint64_t rm;
        if (TRACE) {
            const uint64_t pos = (uint64_t)st.a[0] + st.d[0] + st.a[1] + st.d[1] + lane;
            double u = 1.0;
            if ((int64_t)pos < D.trace_stride) u = trace[pos];
            rm = __ballot(u < lr);
        } else {
            const uint32_t j = st.sent[0] + st.sent[1] + lane;
            uint32_t w[4];
            philox4x32_10(j >> 2, mi, episode, gid, D.key0, D.key1, w);
            const uint32_t x = (j & 3u) == 0 ? w[0] : (j & 3u) == 1 ? w[1] : (j & 3u) == 2 ? w[2] : w[3];
            rm = __ballot(always || x < thr);
        }
        // ---- per-sender send sequences: t0 + k*G, exact while the preconditions hold
        double G[2];
        bool ok = kRelaxSweeps != 0 || (st.tu >= maxq);   // (the sweeps below are the plain recurrence: exact drains are not needed)
        double tend_max = 0.0, lim = end;   // (lim: the interval's end or the top of the lower binade of the two send times)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const double t0 = st.t[s], t1s = t0 + gap[s];
            G[s] = t1s - t0;
            const double t2s = t1s + gap[s], tend = t0 + 64.0 * G[s];
            const double ttop = pow2_f64((int)exponent_bits(t0) - 1022);
            lim = ttop < lim ? ttop : lim;
            ok = ok && (t2s - t1s == G[s]) && (t0 >= 128.0 * gap[s]) && (exponent_bits(t0) == exponent_bits(t2s)) &&
                 (G[s] > 0.0);
            tend_max = tend > tend_max ? tend : tend_max;
        }
        ok = ok && (kRelaxSweeps != 0 || st.tu + st.tu >= tend_max);
        if (kProfile && !ok) {
            uint32_t why = st.tu >= maxq ? 0u : 0x10000u;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const double t0 = st.t[s], t1s = t0 + gap[s], g = t1s - t0, t2s = t1s + gap[s];
                why |= (t2s - t1s == g ? 0u : 0x20000u) | (t0 >= 128.0 * gap[s] ? 0u : 0x40000u) |
                       (exponent_bits(t0) == exponent_bits(t2s) ? 0u : 0x80000u);
            }
            why |= st.tu + st.tu >= tend_max ? 0u : 0x100000u;
            st.prof_why |= why;
        }
        double my_t = 0.0, my_lat = 0.0;
        bool my_drop = true;
        uint32_t my_s = 0, nv;
        bool dbg_settled = false;
        int dbg_sweeps = 0;
        const uint64_t dbg_c1 = prof_counters(D) ? __builtin_readcyclecounter() : 0;
        if (!ok) {
            // ---- serial pass: the plain merged recurrence, wave-uniform, lane k keeps packet k
            uint32_t k = 0;
            for (; k < 64u; k++) {
                const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(st.t[1] < st.t[0] ? 1 : 0));
                const double t = s ? st.t[1] : st.t[0];
                if (!__builtin_amdgcn_readfirstlane((int)(t < end))) break;
                const bool rnd = (rm >> k) & 1ull;
                bool dropped;
                const double2 rec = link_send(t, rnd, dl, maxq, ebw, st.q, st.tu, dropped);
                if (lane == k) { my_t = rec.x; my_lat = rec.y; my_drop = dropped; my_s = s; }
                if (s) st.t[1] = t + gap[1];
                else st.t[0] = t + gap[0];
            }
            nv = k;
        } else {
            // ---- merge path: c = how many of sender 0's packets precede merged position `lane`
            // (sender 0 first on equal times); smallest c with B[lane-c-1] < A[c]
            uint32_t lo = 0, hi = lane;
            while (lo < hi) {
                const uint32_t c = (lo + hi) >> 1;
                const double Ac = st.t[0] + (double)c * G[0];
                const double Bp = st.t[1] + (double)(lane - c - 1) * G[1];
                if (Bp < Ac) hi = c;
                else lo = c + 1;
            }
            const uint32_t c0 = lo, c1 = lane - lo;
            const double A = st.t[0] + (double)c0 * G[0], B = st.t[1] + (double)c1 * G[1];
            my_s = (A <= B) ? 0u : 1u;
            const double tk = my_s ? B : A;
            const bool valid = tk < lim;
            const uint64_t vmask = __ballot(valid);
            nv = (uint32_t)__popcll(vmask);  // merged times increase: valid lanes are a prefix
            const uint64_t rmask = rm;
            const uint64_t openm = vmask & ~rmask;  // packets that reach the queue
            bool settled = false;
            double qc = 0.0;
            if (kRelaxSweeps != 0) {
                // ---- phase 1, side by side.  Every lane holds the link state (queue, time of the last update) BEHIND its
                // packet (ns:72-84): an accepted packet leaves (1/bw + the queue it found, its send time), one refused by the
                // full queue leaves the queue drained to its send time, one lost at random or past the interval's end hands
                // on what it was given.  One sweep = every lane takes the
                // state of the lane below it (lane 0: the state the pass started from) and redoes its own packet; after
                // sweep i the lanes 0 .. i-1 are final, so nv sweeps are the plain recurrence bit for bit (the same
                // expressions on the same operands), and a sweep that moves nothing has reached that fixed point early:
                // a packet that finds the queue drained leaves (1/bw, t_k) whatever came before it, so with the guess
                // "everybody found it drained" sweep i settles the i-th packet of EVERY busy period of the pass at once.
                const uint64_t upto = openm & ((2ull << lane) - 1ull);   // packets that reach the queue, up to this lane
                const bool mine = (openm >> lane) & 1ull;
                const int gl = upto ? 63 - (int)__clzll((long long)upto) : 0;
                double oq = ebw, ot = __shfl(tk, gl);
                if (!upto) { oq = st.q; ot = st.tu; }
                bool full = false, moved = false;
                for (uint32_t sweep = 0; sweep < nv; sweep += 4u) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const double iq = wave_shr1_f64(oq, st.q), it = wave_shr1_f64(ot, st.tu);
                        qc = max0(iq - (tk - it));
                        full = ebw + qc > maxq;
                        const double nq = mine ? (full ? qc : ebw + qc) : iq, nt = mine ? tk : it;   // ns:75-82
                        if (r == 3)
                            moved = valid && (__double_as_longlong(nq) != __double_as_longlong(oq) ||
                                              __double_as_longlong(nt) != __double_as_longlong(ot));
                        oq = nq;
                        ot = nt;
                    }
                    if (kProfile) dbg_sweeps += 4;
                    if (!__ballot(moved)) break;
                }
                settled = true;
                if (kProfile) dbg_settled = true;
                my_drop = !(mine && !full);
            }
            if (!settled) {
                // phase 1: accepted packet to accepted packet
                double qm = st.q, tm = st.tu;
                uint64_t open = openm, amask = 0;
                uint32_t na = 0;
                double seg_q = 0.0, seg_t = 0.0;
                while (open) {
                    const double qk = max0(qm - (tk - tm));
                    const bool full = ebw + qk > maxq;
                    const uint64_t cm = open & ~__ballot(full);
                    if (!cm) break;
                    const uint32_t ks = (uint32_t)__ffsll((unsigned long long)cm) - 1u;
                    qm = rl_f64(ebw + qk, ks);
                    tm = rl_f64(tk, ks);
                    if (lane == na) { seg_q = qm; seg_t = tm; }
                    na++;
                    amask |= 1ull << ks;
                    open &= ~((2ull << ks) - 1ull);
                }
                // phase 2: every lane finishes its own packet
                const uint32_t seg = (uint32_t)count_below(amask);
                const int src = seg ? (int)seg - 1 : 0;
                double q_seg = __shfl(seg_q, src), t_seg = __shfl(seg_t, src);
                if (!seg) { q_seg = st.q; t_seg = st.tu; }
                qc = max0(q_seg - (tk - t_seg));
                my_drop = !((amask >> lane) & 1ull);
            }
            my_lat = dl + qc;
            const double my_q_after = my_drop ? qc : ebw + qc;
            my_t = tk + my_lat;
            const uint64_t touch = vmask & ~rmask;
            if (touch) {
                const uint32_t kl = 63u - (uint32_t)__clzll((long long)touch);
                st.q = rl_f64(my_q_after, kl);
                st.tu = rl_f64(tk, kl);
            }
            const uint32_t n1 = (uint32_t)__popcll(__ballot(valid && my_s == 1u)), n0 = nv - n1;
            if (n0) st.t[0] = (st.t[0] + (double)(n0 - 1u) * G[0]) + gap[0];   // ns:161 on the last packet's (exact) send time
            if (n1) st.t[1] = (st.t[1] + (double)(n1 - 1u) * G[1]) + gap[1];
        }
        // ---- records: four dense runs (sender x accepted/dropped)
        const bool valid = lane < nv;
        if (TRACE && (int64_t)((uint64_t)st.a[0] + st.d[0] + st.a[1] + st.d[1] + nv) > D.trace_stride)
            st.flags |= PCC_FLAG_TRACE_OVERRUN;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const uint64_t dm = __ballot(valid && my_s == (uint32_t)s && my_drop);
            const uint64_t am = __ballot(valid && my_s == (uint32_t)s && !my_drop);
            if (valid && my_s == (uint32_t)s) {
                double2 rec;
                rec.x = my_t;
                rec.y = my_lat;
                const uint32_t cap_b = caps[s] << 4, mask_b = (caps[s] - 1u) << 4, dmask_b = (2u * caps[s] - 1u) << 4;
                const uint32_t off = my_drop ? cap_b + (((st.d[s] + (uint32_t)count_below(dm)) << 4) & dmask_b)
                                             : (((st.a[s] + (uint32_t)count_below(am)) << 4) & mask_b);
                st_rec_pass(reinterpret_cast<double2 *>((s ? base1 : base0) + off), rec);
            }
            st.a[s] += (uint32_t)__popcll(am);
            st.d[s] += (uint32_t)__popcll(dm);
            st.sent[s] += (uint32_t)__popcll(am) + (uint32_t)__popcll(dm);
        }
        if (kProfile) st.prof_other += ok ? 1u : 0x10000u;
        if (prof_counters(D) && lane == 0) {   // (profile build: 0 = settled side by side, 2 = serial chain, 3 = plain recurrence)
            const int c = !ok ? 3 : dbg_settled ? 0 : 2;
            atomicAdd(&D.pass_stats[c], 1ull);
            atomicAdd(&D.pass_stats[4 + c], (unsigned long long)nv);
            atomicAdd(&D.pass_stats[ok ? 14 : 9], (unsigned long long)(__builtin_readcyclecounter() - dbg_c1));
            if (dbg_settled) atomicAdd(&D.pass_stats[8], (unsigned long long)dbg_sweeps);
        }
    }
}

}  // namespace
