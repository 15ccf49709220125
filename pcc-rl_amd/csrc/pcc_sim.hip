// pcc_sim.hip -- MI355X (gfx950) batched congestion-control simulator: kernels + C ABI.
//
// What this replaces (reference = PCCproject/PCC-RL; "ns" = src/gym/network_sim.py,
// "so" = src/common/sender_obs.py): the per-env heap-driven discrete-event loop
// Network.run_for_dur (ns:123-205) with its Link queue model (ns:56-96) and Sender
// accounting (ns:207-342), the monitor-interval metrics + history (so:20-206) and the env
// protocol around them (ns:344-496) -- for N independent envs advanced one monitor interval
// (MI) per launch.
//
// Formulation (NOT the reference's heap):
//   * For one sender the heap only ever holds three kinds of events: the single pending SEND,
//     packets on the forward hop ("hop-1" events) and packets on the return hop ("hop-2"
//     events).  Heap pop order == global order of each kind by the tuple key
//     (time, ..., latency, dropped), so each kind is a key-sorted FIFO.
//   * Link state and the loss RNG are touched only by SEND events, and the rate is constant
//     inside an MI, so an MI can be processed stream by stream: all SENDs before the MI end,
//     then all hop-1 events, then all hop-2 events, then the single event that ends the MI
//     (the first event with time >= end, which the reference still processes, ns:128-131).
//   * In-flight packets live in one HBM ring per env per sender: 16-byte records
//     (fp64 event time, fp64 accumulated latency with the drop flag in its sign bit).  The
//     ring is [head, mid) = hop-2 region, [mid, tail) = hop-1 region; a hop-1 event is
//     converted in place.  Records are kept key-sorted by insertion from the back (float
//     rounding makes a dropped packet and its successor arrive "at the same time", and the
//     tuple tie-break decides who ends an MI).
//   * Every floating-point operation on the timeline is IEEE binary64 in the reference's
//     order (compile with -ffp-contract=off); the per-MI RTT means replicate numpy's
//     pairwise summation, because run_dur = 0.5 * mean feeds back into the event boundaries.
//
// Mapping: one lane per env, one 64-lane wavefront per workgroup; SoA state so that lane i
// of a wave touches element i of every array (coalesced); no LDS, no MFMA (there is no
// contraction anywhere on this path).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "pcc_sim.h"

namespace {

constexpr int kMaxFeatures = 16;
constexpr int kMaxSenders = 2;
constexpr int kWave = 64;
constexpr double kMaxRate = 1000.0;      // ns:36
constexpr double kMinRate = 40.0;        // ns:37
constexpr double kRewardScale = 0.001;   // ns:39
constexpr int64_t kBytesPerPacket = 1500;  // ns:46
constexpr uint32_t kNpBufsize = 8192;    // numpy add.reduce inner-loop chunk
constexpr uint32_t kParamTag = 0xFFFFFFFFu;

// metric registry so:193-206
__constant__ double c_metric_scale[PCC_N_METRICS] = {1e7, 1e7, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
const double h_metric_min[PCC_N_METRICS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0, -1.0, 0.0, 0.0, 1.0, 0.0};
const double h_metric_max[PCC_N_METRICS] = {1e9, 1e9, 100.0, 100.0, 100.0, 1.0, 10.0, 10.0, 100.0, 100.0, 10000.0, 1000.0};
const double h_metric_scale[PCC_N_METRICS] = {1e7, 1e7, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// Everything a kernel needs, passed by value.
struct Dev {
    int64_t n;
    int ns, H, F, HF;
    int32_t fid[kMaxFeatures];
    uint32_t cap, cap_mask;
    uint32_t key0, key1, gid_base;
    double delta_scale;
    uint32_t max_steps;
    double lo[5], hi[5];
    int rng_mode;
    const double *trace;
    int64_t trace_stride;
    const double *p_bw, *p_dl, *p_queue, *p_loss, *p_rate0;
    // link + env state, [N]
    double *bw, *dl, *lr, *maxq, *ebw, *q, *tu, *now, *run_dur;
    uint32_t *steps, *episode, *flags;
    uint8_t *done;
    unsigned long long *total_sent;
    // per sender, [S][N]
    double *rate, *rate0, *next_send, *min_lat, *ep_return, *last_return;
    uint32_t *h2, *h1, *tail;
    float *hist;    // [N][S][HF]
    double2 *ring;  // [N][S][cap]
    unsigned long long *prof;  // optional [N][8] cycle stamps per step (diagnostics), else null
};

// --------------------------------------------------------------------------------------
// small helpers
// --------------------------------------------------------------------------------------
__device__ __forceinline__ double max0(double x) { return x > 0.0 ? x : 0.0; }  // max(0.0, x)

__device__ __forceinline__ bool rec_dropped(double2 r) { return __double_as_longlong(r.y) < 0; }

// Python tuple order on (time, latency, dropped) -- the fields that can differ inside one
// stream of one sender (ns:111,161,178)
__device__ __forceinline__ bool key_less(double2 a, double2 b) {
    if (a.x != b.x) return a.x < b.x;
    const double la = fabs(a.y), lb = fabs(b.y);
    if (la != lb) return la < lb;
    return !rec_dropped(a) && rec_dropped(b);
}

__device__ __forceinline__ void insert_sorted(double2 *ring, uint32_t mask, uint32_t lo, uint32_t pos,
                                              double2 rec) {
    while (pos > lo) {
        const double2 prev = ring[(pos - 1) & mask];
        if (!key_less(rec, prev)) break;
        ring[pos & mask] = prev;
        pos--;
    }
    ring[pos & mask] = rec;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u32_to_unit(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }

// --------------------------------------------------------------------------------------
// one env in registers
// --------------------------------------------------------------------------------------
template <int NS>
struct Env {
    double dl, lr, maxq, ebw;  // link parameters (ns:59-64); ebw = 1.0 / bw (ns:77)
    double q, tu;              // link-0 queue delay and its update time (ns:62-63)
    double now;                // network clock (ns:102)
    double rate[NS], nsend[NS];
    uint32_t h2[NS], h1[NS], tail[NS];
    double2 *ring[NS];
    uint32_t flags;
};

template <int NS>
struct Rng {
    int mode;
    const double *trace;  // this env's row
    int64_t trace_n;
    uint32_t k0, k1, gid, episode, mi;
    uint32_t j[NS];
    uint32_t w[NS][4];
};

template <int NS>
struct MiCounts {
    uint32_t sent[NS], acked[NS], lost[NS], from[NS];
    double start;
};

// hop-1 region tail bookkeeping for the sorted push
struct Last {
    double2 rec;
    bool have;
};

template <int NS, int s>
__device__ __forceinline__ double packet_uniform(Env<NS> &e, Rng<NS> &r) {
    if (r.mode == PCC_RNG_TRACE) {
        uint64_t pos = e.tail[0];
        if (NS > 1) pos += e.tail[NS - 1];
        if ((int64_t)pos >= r.trace_n) { e.flags |= PCC_FLAG_TRACE_OVERRUN; return 1.0; }
        return r.trace[pos];
    }
    const uint32_t jj = r.j[s]++;
    if ((jj & 3u) == 0u) philox4x32_10(jj >> 2, r.mi + ((uint32_t)s << 24), r.episode, r.gid, r.k0, r.k1, r.w[s]);
    const uint32_t i = jj & 3u;
    const uint32_t x = i == 0 ? r.w[s][0] : i == 1 ? r.w[s][1] : i == 2 ? r.w[s][2] : r.w[s][3];
    return u32_to_unit(x);
}

// SEND event at time t for sender s: ns:155-178 with Link.packet_enters_link ns:72-84
template <int NS, int s>
__device__ __forceinline__ void send_packet(Env<NS> &e, Rng<NS> &rng, const double (&gap)[NS], uint32_t mask,
                                            uint32_t cap, Last &last, MiCounts<NS> &c) {
    const double t = e.nsend[s];
    c.sent[s]++;                                   // ns:260-262
    e.nsend[s] = t + gap[s];                       // ns:161
    const double qcur = max0(e.q - (t - e.tu));    // ns:66-67
    const double lat0 = e.dl + qcur;               // ns:170 (latency before this packet queues)
    const double u = packet_uniform<NS, s>(e, rng);
    bool dropped;
    if (u < e.lr) {                                // ns:73-74: random loss leaves the queue untouched
        dropped = true;
    } else {
        e.q = qcur;                                // ns:75-76
        e.tu = t;
        if (e.ebw + e.q > e.maxq) {                // ns:79-81 tail drop
            dropped = true;
        } else {
            e.q += e.ebw;                          // ns:82
            dropped = false;
        }
    }
    double2 rec;
    rec.x = t + lat0;                              // ns:174
    rec.y = dropped ? -lat0 : lat0;                // ns:173 (0.0 + lat0), ns:175
    if (e.tail[s] - e.h2[s] >= cap) {              // ring full: never silent
        e.flags |= PCC_FLAG_RING_OVERFLOW;
        return;
    }
    double2 *ring = e.ring[s];
    if (!last.have || !key_less(rec, last.rec)) {
        ring[e.tail[s] & mask] = rec;
        last.rec = rec;
    } else {
        insert_sorted(ring, mask, e.h1[s], e.tail[s], rec);  // last stays the region maximum
    }
    last.have = true;
    e.tail[s]++;
}

// hop-1 event of sender s (ns:147-154): link 1 is never entered by a packet, so its latency is
// dl + max(0, 0 - (t - 0)) = dl exactly; the record becomes a hop-2 event in place.
template <int NS, int s>
__device__ __forceinline__ void hop1_event(Env<NS> &e, uint32_t mask, double2 rec, Last &last2) {
    double2 conv;
    conv.x = rec.x + e.dl;
    const double l2 = fabs(rec.y) + e.dl;
    conv.y = rec_dropped(rec) ? -l2 : l2;
    double2 *ring = e.ring[s];
    if (!last2.have || !key_less(conv, last2.rec)) {
        ring[e.h1[s] & mask] = conv;
        last2.rec = conv;
    } else {
        insert_sorted(ring, mask, e.h2[s], e.h1[s], conv);
    }
    last2.have = true;
    e.h1[s]++;
}

// One monitor interval for one env: Network.run_for_dur, ns:123-178 (reward is computed by the
// caller from the counts).  On return e.h2[s] has advanced past the hop-2 events consumed;
// c.from[s] is where they started.
template <int NS>
__device__ __forceinline__ void run_mi(Env<NS> &e, Rng<NS> &rng, double dur, uint32_t mask, uint32_t cap,
                                       MiCounts<NS> &c, unsigned long long *ts = nullptr) {
    const double end = e.now + dur;  // ns:124
    c.start = e.now;                 // ns:319-324 reset_obs
    double gap[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        c.sent[s] = c.acked[s] = c.lost[s] = 0;
        c.from[s] = e.h2[s];
        gap[s] = 1.0 / e.rate[s];
        rng.j[s] = 0;
    }
    if (ts) ts[0] = clock64();
    if (!(e.now < end)) return;  // ns:128 loop never entered

    // Two passes keep ring occupancy at the true in-flight count: pass 0 retires what was already in
    // flight (hop-1 then hop-2 events before `end`), pass 1 runs the SEND stream and then the events
    // those sends produced inside this MI.  Order between streams is free (see file header).
    double t1[NS], t2[NS];
    double2 r1[NS], r2[NS];
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            if (ts) ts[1] = clock64();
            // ---- SEND stream: every SEND with time < end
            Last last[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                last[s].have = e.tail[s] > e.h1[s];
                if (last[s].have) last[s].rec = e.ring[s][(e.tail[s] - 1) & mask];
            }
            if (NS == 1) {
                while (e.nsend[0] < end) send_packet<NS, 0>(e, rng, gap, mask, cap, last[0], c);
            } else {
                for (;;) {
                    // (time, sender id): lower id first on equal times
                    const bool pick1 = e.nsend[NS - 1] < e.nsend[0];
                    const double t = pick1 ? e.nsend[NS - 1] : e.nsend[0];
                    if (!(t < end)) break;
                    if (pick1) send_packet<NS, NS - 1>(e, rng, gap, mask, cap, last[NS - 1], c);
                    else send_packet<NS, 0>(e, rng, gap, mask, cap, last[0], c);
                }
            }
            if (ts) ts[2] = clock64();
        }
        // ---- hop-1 stream, then hop-2 stream, per sender
#pragma unroll
        for (int s = 0; s < NS; s++) {
            Last last2;
            last2.have = e.h1[s] > e.h2[s];
            if (last2.have) last2.rec = e.ring[s][(e.h1[s] - 1) & mask];
            t1[s] = INFINITY;
            while (e.h1[s] < e.tail[s]) {
                const double2 rec = e.ring[s][e.h1[s] & mask];
                if (!(rec.x < end)) { t1[s] = rec.x; r1[s] = rec; break; }
                if (s == 0) hop1_event<NS, 0>(e, mask, rec, last2);
                else hop1_event<NS, NS - 1>(e, mask, rec, last2);
            }
            t2[s] = INFINITY;
            while (e.h2[s] < e.h1[s]) {
                const double2 rec = e.ring[s][e.h2[s] & mask];
                if (!(rec.x < end)) { t2[s] = rec.x; r2[s] = rec; break; }
                if (rec_dropped(rec)) c.lost[s]++;   // ns:141-143
                else c.acked[s]++;                   // ns:144-146
                e.h2[s]++;
            }
        }
    }
    if (ts) ts[3] = clock64();

    // ---- the event that ends the MI: smallest (time, sender, type 'A'<'S', hop) among the
    // stream heads; all of them are >= end here
    int best = 0;  // 3*s + {0: hop-1, 1: hop-2, 2: SEND}
    double tb = t1[0];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (s > 0 && t1[s] < tb) { tb = t1[s]; best = 3 * s; }
        if (t2[s] < tb) { tb = t2[s]; best = 3 * s + 1; }
        if (e.nsend[s] < tb) { tb = e.nsend[s]; best = 3 * s + 2; }
    }
    e.now = tb;  // ns:131
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (best == 3 * s) {
            Last l2;
            l2.have = e.h1[s] > e.h2[s];
            if (l2.have) l2.rec = e.ring[s][(e.h1[s] - 1) & mask];
            if (s == 0) hop1_event<NS, 0>(e, mask, r1[s], l2);
            else hop1_event<NS, NS - 1>(e, mask, r1[s], l2);
        } else if (best == 3 * s + 1) {
            if (rec_dropped(r2[s])) c.lost[s]++;
            else c.acked[s]++;
            e.h2[s]++;
        } else if (best == 3 * s + 2) {
            Last l1;
            l1.have = e.tail[s] > e.h1[s];
            if (l1.have) l1.rec = e.ring[s][(e.tail[s] - 1) & mask];
            if (s == 0) send_packet<NS, 0>(e, rng, gap, mask, cap, l1, c);
            else send_packet<NS, NS - 1>(e, rng, gap, mask, cap, l1, c);
        }
    }
}

// --------------------------------------------------------------------------------------
// np.mean over the RTTs acknowledged in this MI (so:119-122, 138-142), numpy-exact.
// The samples are the non-dropped records of ring[from, to) in order.
// --------------------------------------------------------------------------------------
struct RttStream {
    const double2 *ring;
    uint32_t mask, pos;
    __device__ __forceinline__ double next() {
        double2 r;
        do { r = ring[pos++ & mask]; } while (rec_dropped(r));
        return r.y;
    }
};

// numpy DOUBLE_pairwise_sum for n <= 128 (one leaf of the recursion)
__device__ __forceinline__ double pw_leaf(RttStream &st, uint32_t n) {
    if (n < 8) {
        double res = 0.;
        for (uint32_t i = 0; i < n; i++) res += st.next();
        return res;
    }
    double r0 = st.next(), r1 = st.next(), r2 = st.next(), r3 = st.next();
    double r4 = st.next(), r5 = st.next(), r6 = st.next(), r7 = st.next();
    const uint32_t lim = n - (n % 8);
    uint32_t i = 8;
    for (; i < lim; i += 8) {
        r0 += st.next(); r1 += st.next(); r2 += st.next(); r3 += st.next();
        r4 += st.next(); r5 += st.next(); r6 += st.next(); r7 += st.next();
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += st.next();
    return res;
}

// numpy DOUBLE_pairwise_sum for n <= 8192: the recursion n -> (n2 = n/2 - (n/2)%8, n - n2)
// unrolled into a left-to-right walk over its leaves with an explicit stack (depth <= 6)
__device__ __forceinline__ double pw_sum(RttStream &st, uint32_t n) {
    uint32_t right_n[8];
    double left_sum[8];
    uint32_t have_left = 0;
    int sp = 0;
    uint32_t cur = n;
    double val;
    for (;;) {
        while (cur > 128) {
            uint32_t n2 = cur / 2;
            n2 -= n2 % 8;
            right_n[sp] = cur - n2;
            have_left &= ~(1u << sp);
            sp++;
            cur = n2;
        }
        val = pw_leaf(st, cur);
        bool descend = false;
        while (sp > 0) {
            const int top = sp - 1;
            if (!(have_left & (1u << top))) {
                left_sum[top] = val;
                have_left |= 1u << top;
                cur = right_n[top];
                descend = true;
                break;
            }
            val = left_sum[top] + val;
            sp--;
        }
        if (!descend) return val;
    }
}

// np.add.reduce: per-8192 chunk pairwise sums accumulated left to right
__device__ __forceinline__ double np_sum(RttStream &st, uint32_t n) {
    double tot = 0.;
    for (uint32_t i = 0; i < n; i += kNpBufsize) {
        const uint32_t m = n - i < kNpBufsize ? n - i : kNpBufsize;
        tot += pw_sum(st, m);
    }
    return tot;
}

// the 12 metrics of one MI (so:110-191).  min_lat: connection minimum, 0.0 = no entry yet.
__device__ __forceinline__ void mi_metrics(uint32_t sent, uint32_t acked, uint32_t lost, double dur,
                                           const double2 *ring, uint32_t mask, uint32_t from, double &min_lat,
                                           bool update_min, bool need_halves, double (&m)[PCC_N_METRICS]) {
    const int64_t bs = (int64_t)sent * kBytesPerPacket, ba = (int64_t)acked * kBytesPerPacket,
                  bl = (int64_t)lost * kBytesPerPacket;
    m[PCC_M_RECV_DUR] = dur;
    m[PCC_M_SEND_DUR] = dur;
    m[PCC_M_SEND_RATE] = dur > 0.0 ? 8.0 * (double)bs / dur : 0.0;
    m[PCC_M_RECV_RATE] = dur > 0.0 ? 8.0 * (double)(ba - kBytesPerPacket) / dur : 0.0;
    double lat = 0.0, inc = 0.0;
    if (acked > 0) {
        RttStream st{ring, mask, from};
        lat = np_sum(st, acked) / (double)acked;
        const uint32_t half = acked / 2;
        if (need_halves && half >= 1) {
            RttStream sh{ring, mask, from};
            const double first = np_sum(sh, half) / (double)half;
            const double second = np_sum(sh, acked - half) / (double)(acked - half);
            inc = second - first;
        }
    }
    m[PCC_M_AVG_LATENCY] = lat;
    m[PCC_M_LOSS_RATIO] = (bl + ba > 0) ? (double)bl / (double)(bl + ba) : 0.0;
    m[PCC_M_LATENCY_INCREASE] = inc;
    m[PCC_M_ACK_LATENCY_INFLATION] = dur > 0.0 ? inc / dur : 0.0;
    m[PCC_M_SENT_LATENCY_INFLATION] = dur > 0.0 ? inc / dur : 0.0;
    double cm;
    if (min_lat > 0.0) {
        if (lat == 0.0) cm = min_lat;
        else if (lat < min_lat) { cm = lat; if (update_min) min_lat = lat; }
        else cm = min_lat;
    } else {
        if (lat > 0.0) { cm = lat; if (update_min) min_lat = lat; }
        else cm = 0.0;
    }
    m[PCC_M_CONN_MIN_LATENCY] = cm;
    m[PCC_M_SEND_RATIO] = (m[PCC_M_RECV_RATE] > 0.0 && m[PCC_M_SEND_RATE] < 1000.0 * m[PCC_M_RECV_RATE])
                              ? m[PCC_M_SEND_RATE] / m[PCC_M_RECV_RATE] : 1.0;
    m[PCC_M_LATENCY_RATIO] = cm > 0.0 ? lat / cm : 1.0;
}

__device__ __forceinline__ double select_metric(const double (&m)[PCC_N_METRICS], int id) {
    double v = m[0];
#pragma unroll
    for (int k = 1; k < PCC_N_METRICS; k++) v = (id == k) ? m[k] : v;
    return v;
}

// --------------------------------------------------------------------------------------
// state load/store
// --------------------------------------------------------------------------------------
template <int NS>
__device__ __forceinline__ void load_env(const Dev &D, int64_t i, Env<NS> &e) {
    e.dl = D.dl[i]; e.lr = D.lr[i]; e.maxq = D.maxq[i]; e.ebw = D.ebw[i];
    e.q = D.q[i]; e.tu = D.tu[i]; e.now = D.now[i];
    e.flags = D.flags[i];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = (int64_t)s * D.n + i;
        e.rate[s] = D.rate[k]; e.nsend[s] = D.next_send[k];
        e.h2[s] = D.h2[k]; e.h1[s] = D.h1[k]; e.tail[s] = D.tail[k];
        e.ring[s] = D.ring + ((int64_t)i * NS + s) * D.cap;
    }
}

template <int NS>
__device__ __forceinline__ void store_env(const Dev &D, int64_t i, const Env<NS> &e) {
    D.q[i] = e.q; D.tu[i] = e.tu; D.now[i] = e.now;
    D.flags[i] = e.flags;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = (int64_t)s * D.n + i;
        D.rate[k] = e.rate[s]; D.next_send[k] = e.nsend[s];
        D.h2[k] = e.h2[s]; D.h1[k] = e.h1[s]; D.tail[k] = e.tail[s];
    }
}

template <int NS>
__device__ __forceinline__ void init_rng(const Dev &D, int64_t i, uint32_t episode, uint32_t mi, Rng<NS> &r) {
    r.mode = D.rng_mode;
    r.trace = D.trace ? D.trace + i * D.trace_stride : nullptr;
    r.trace_n = D.trace_stride;
    r.k0 = D.key0; r.k1 = D.key1;
    r.gid = D.gid_base + (uint32_t)i;
    r.episode = episode;
    r.mi = mi;
#pragma unroll
    for (int s = 0; s < NS; s++) r.j[s] = 0;
}

// --------------------------------------------------------------------------------------
// reset kernel: ns:454-484
// --------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(kWave) void reset_kernel(Dev D, const uint8_t *mask, int use_done, float *obs_out) {
    const int64_t i = (int64_t)blockIdx.x * kWave + threadIdx.x;
    if (i >= D.n) return;
    if (mask && !mask[i]) return;
    if (use_done && !D.done[i]) return;

    const uint32_t episode = D.episode[i];
    D.episode[i] = episode + 1;

    // ---- parameters: ns:455-466
    double bw, lat, queue, loss, rate0[NS];
    if (D.p_bw) {
        bw = D.p_bw[i]; lat = D.p_dl[i]; queue = D.p_queue[i]; loss = D.p_loss[i];
#pragma unroll
        for (int s = 0; s < NS; s++) rate0[s] = D.p_rate0[(int64_t)s * D.n + i];
    } else {
        uint32_t w0[4], w1[4];
        const uint32_t gid = D.gid_base + (uint32_t)i;
        philox4x32_10(0u, kParamTag, episode, gid, D.key0, D.key1, w0);
        philox4x32_10(1u, kParamTag, episode, gid, D.key0, D.key1, w1);
        bw = D.lo[0] + (D.hi[0] - D.lo[0]) * u32_to_unit(w0[0]);
        lat = D.lo[1] + (D.hi[1] - D.lo[1]) * u32_to_unit(w0[1]);
        queue = (double)(1 + (long long)exp(D.lo[2] + (D.hi[2] - D.lo[2]) * u32_to_unit(w0[2])));
        loss = D.lo[3] + (D.hi[3] - D.lo[3]) * u32_to_unit(w0[3]);
#pragma unroll
        for (int s = 0; s < NS; s++) rate0[s] = (D.lo[4] + (D.hi[4] - D.lo[4]) * u32_to_unit(w1[s])) * bw;
    }

    Env<NS> e;
    e.dl = lat; e.lr = loss; e.maxq = queue / bw; e.ebw = 1.0 / bw;  // ns:58-64,77
    e.q = 0.0; e.tu = 0.0; e.now = 0.0;
    e.flags = D.flags[i];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        e.rate[s] = rate0[s];
        e.nsend[s] = 1.0 / rate0[s];  // ns:111
        e.h2[s] = e.h1[s] = e.tail[s] = 0;
        e.ring[s] = D.ring + ((int64_t)i * NS + s) * D.cap;
    }
    const double run_dur = 3 * lat;  // ns:467

    // ---- two unrecorded warm-up MIs: ns:478-479
    Rng<NS> rng;
    init_rng<NS>(D, i, episode, 0, rng);
    unsigned long long sent_total = 0;
    for (int w = 0; w < 2; w++) {
        rng.mi = (uint32_t)w;
        MiCounts<NS> c;
        run_mi<NS>(e, rng, run_dur, D.cap_mask, D.cap, c);
#pragma unroll
        for (int s = 0; s < NS; s++) sent_total += c.sent[s];
    }

    D.bw[i] = bw; D.dl[i] = e.dl; D.lr[i] = e.lr; D.maxq[i] = e.maxq; D.ebw[i] = e.ebw;
    D.run_dur[i] = run_dur;
    D.steps[i] = 0;
    D.done[i] = 0;
    D.total_sent[i] += sent_total;
    store_env<NS>(D, i, e);

    // ---- fresh sender: empty history (so:57-62) and no connection minimum (so:158)
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = (int64_t)s * D.n + i;
        D.rate0[k] = rate0[s];
        D.min_lat[k] = 0.0;
        D.ep_return[k] = 0.0;
        float *hist = D.hist + ((int64_t)i * NS + s) * D.HF;
        float *obs = obs_out ? obs_out + ((int64_t)i * NS + s) * D.HF : nullptr;
        for (int h = 0; h < D.H; h++)
            for (int f = 0; f < D.F; f++) {
                const int id = D.fid[f];
                const double v = (id == PCC_M_SEND_RATIO || id == PCC_M_LATENCY_RATIO) ? 1.0 : 0.0;
                const float x = (float)(v / c_metric_scale[id]);
                hist[h * D.F + f] = x;
                if (obs) obs[h * D.F + f] = x;
            }
    }
}

// --------------------------------------------------------------------------------------
// step kernel: ns:406-444
// --------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(kWave) void step_kernel(Dev D, const void *actions, int actions_f64, float *obs_out,
                                                     float *reward_out, uint8_t *done_out, double *steps_out) {
    const int64_t i = (int64_t)blockIdx.x * kWave + threadIdx.x;
    if (i >= D.n) return;

    Env<NS> e;
    load_env<NS>(D, i, e);
    const uint32_t steps = D.steps[i];
    const double run_dur = D.run_dur[i];

    // ---- apply_rate_delta + set_rate: ns:235-241, 275-281
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t a = i * NS + s;
        double delta = actions_f64 ? ((const double *)actions)[a] : (double)((const float *)actions)[a];
        delta *= D.delta_scale;
        double r = delta >= 0.0 ? e.rate[s] * (1.0 + delta) : e.rate[s] / (1.0 - delta);
        if (r > kMaxRate) r = kMaxRate;
        if (r < kMinRate) r = kMinRate;
        e.rate[s] = r;
    }

    // ---- one monitor interval
    Rng<NS> rng;
    init_rng<NS>(D, i, D.episode[i] - 1, steps + 2, rng);
    MiCounts<NS> c;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    run_mi<NS>(e, rng, run_dur, D.cap_mask, D.cap, c, D.prof ? ts : nullptr);
    if (D.prof) ts[4] = clock64();
    const double dur = e.now - c.start;  // ns:311-314

    // ---- metrics, history, reward
    bool need_halves = steps_out != nullptr;
    for (int f = 0; f < D.F; f++) {
        const int id = D.fid[f];
        need_halves |= (id == PCC_M_LATENCY_INCREASE || id == PCC_M_ACK_LATENCY_INFLATION ||
                        id == PCC_M_SENT_LATENCY_INFLATION);
    }
    double new_run_dur = run_dur;
    unsigned long long sent_total = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = (int64_t)s * D.n + i;
        double min_lat = D.min_lat[k];
        double m[PCC_N_METRICS];
        mi_metrics(c.sent[s], c.acked[s], c.lost[s], dur, e.ring[s], D.cap_mask, c.from[s], min_lat, true,
                   need_halves, m);
        D.min_lat[k] = min_lat;
        sent_total += c.sent[s];
        // ns:194,205
        const double reward =
            (10.0 * m[PCC_M_RECV_RATE] / (double)(8 * kBytesPerPacket) - 1e3 * m[PCC_M_AVG_LATENCY] -
             2e3 * m[PCC_M_LOSS_RATIO]) * kRewardScale;
        if (s == 0 && m[PCC_M_AVG_LATENCY] > 0.0) new_run_dur = 0.5 * m[PCC_M_AVG_LATENCY];  // ns:437-438

        // history roll (so:64-66) + observation (ns:400-404, so:68-73)
        float *hist = D.hist + ((int64_t)i * NS + s) * D.HF;
        float *obs = obs_out ? obs_out + ((int64_t)i * NS + s) * D.HF : nullptr;
        const int keep = D.HF - D.F;
        for (int x = 0; x < keep; x++) {
            const float v = hist[x + D.F];
            hist[x] = v;
            if (obs) obs[x] = v;
        }
        for (int f = 0; f < D.F; f++) {
            const int id = D.fid[f];
            const float v = (float)(select_metric(m, id) / c_metric_scale[id]);
            hist[keep + f] = v;
            if (obs) obs[keep + f] = v;
        }
        if (reward_out) reward_out[i * NS + s] = (float)reward;
        const double ret = D.ep_return[k] + reward;
        D.ep_return[k] = ret;
        if (steps + 1 >= D.max_steps) D.last_return[k] = ret;
        if (steps_out) {
            double *row = steps_out + (i * NS + s) * PCC_STEP_COLS;
            row[PCC_COL_SENT] = (double)c.sent[s];
            row[PCC_COL_ACKED] = (double)c.acked[s];
            row[PCC_COL_LOST] = (double)c.lost[s];
            row[PCC_COL_RATE] = e.rate[s];
            row[PCC_COL_CUR_TIME] = e.now;
            row[PCC_COL_RUN_DUR] = 0.0;  // patched below once sender 0 is known
            row[PCC_COL_REWARD] = reward;
#pragma unroll
            for (int k2 = 0; k2 < PCC_N_METRICS; k2++) row[PCC_COL_METRIC0 + k2] = m[k2];
        }
    }
    if (steps_out)
        for (int s = 0; s < NS; s++) steps_out[(i * NS + s) * PCC_STEP_COLS + PCC_COL_RUN_DUR] = new_run_dur;

    store_env<NS>(D, i, e);
    D.run_dur[i] = new_run_dur;
    D.steps[i] = steps + 1;
    D.total_sent[i] += sent_total;
    const uint8_t done = (steps + 1 >= D.max_steps) ? 1 : 0;  // ns:444
    D.done[i] = done;
    if (done_out) done_out[i] = done;
    if (D.prof) {
        ts[5] = clock64();
        ts[6] = c.sent[0];
        ts[7] = c.acked[0] + c.lost[0];
#pragma unroll
        for (int k = 0; k < 8; k++) D.prof[i * 8 + k] = ts[k];
    }
}

}  // namespace

// ======================================================================================
// host side: the C ABI
// ======================================================================================
struct pcc_sim {
    Dev d;
    int device;
    void *state_blob;
    size_t state_bytes;
    void *ring_blob;
    size_t ring_bytes;
    bool ever_reset;
    bool lockstep;      // every env was last reset by the same full reset (host knows when `done` fires)
    uint32_t host_steps;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(char *b) : base(b) {}
    template <typename T>
    T *take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

size_t carve_state(Dev &d, char *base) {
    Carver c(base);
    const size_t n = (size_t)d.n, sn = n * d.ns;
    d.bw = c.take<double>(n); d.dl = c.take<double>(n); d.lr = c.take<double>(n);
    d.maxq = c.take<double>(n); d.ebw = c.take<double>(n); d.q = c.take<double>(n);
    d.tu = c.take<double>(n); d.now = c.take<double>(n); d.run_dur = c.take<double>(n);
    d.steps = c.take<uint32_t>(n); d.episode = c.take<uint32_t>(n); d.flags = c.take<uint32_t>(n);
    d.done = c.take<uint8_t>(n);
    d.total_sent = c.take<unsigned long long>(n);
    d.rate = c.take<double>(sn); d.rate0 = c.take<double>(sn); d.next_send = c.take<double>(sn);
    d.min_lat = c.take<double>(sn); d.ep_return = c.take<double>(sn); d.last_return = c.take<double>(sn);
    d.h2 = c.take<uint32_t>(sn); d.h1 = c.take<uint32_t>(sn); d.tail = c.take<uint32_t>(sn);
    d.hist = c.take<float>(sn * d.HF);
    return (c.off + 255) & ~(size_t)255;
}

int check_hip(hipError_t err, const char *what) {
    if (err == hipSuccess) return PCC_OK;
    return fail(PCC_EHIP, "%s: %s", what, hipGetErrorString(err));
}

}  // namespace

extern "C" {

const char *pcc_last_error(void) { return g_err; }

int pcc_metric_info(int id, double *min_val, double *max_val, double *scale) {
    if (id < 0 || id >= PCC_N_METRICS) return fail(PCC_EINVAL, "metric id %d out of range", id);
    if (min_val) *min_val = h_metric_min[id];
    if (max_val) *max_val = h_metric_max[id];
    if (scale) *scale = h_metric_scale[id];
    return PCC_OK;
}

int pcc_create(int64_t n_envs, int n_senders, int history_len, const int32_t *feature_ids, int n_features,
               uint64_t seed, uint32_t env_gid_base, uint32_t ring_capacity, int device_id, pcc_sim_t **out) {
    if (!out) return fail(PCC_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_envs < 1 || n_envs > (int64_t)1 << 31) return fail(PCC_EINVAL, "n_envs=%lld out of range", (long long)n_envs);
    if (n_senders < 1 || n_senders > kMaxSenders) return fail(PCC_EINVAL, "n_senders must be 1 or 2");
    if (history_len < 1 || history_len > 4096) return fail(PCC_EINVAL, "history_len=%d out of range", history_len);
    if (!feature_ids || n_features < 1 || n_features > kMaxFeatures)
        return fail(PCC_EINVAL, "n_features must be 1..%d", kMaxFeatures);
    for (int f = 0; f < n_features; f++)
        if (feature_ids[f] < 0 || feature_ids[f] >= PCC_N_METRICS)
            return fail(PCC_EINVAL, "feature id %d out of range", feature_ids[f]);
    if (ring_capacity == 0) ring_capacity = 32768;
    if (ring_capacity < 16 || (ring_capacity & (ring_capacity - 1)))
        return fail(PCC_EINVAL, "ring_capacity must be a power of two >= 16");

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
        return fail(PCC_ENODEV, "no HIP device visible (this library is gfx950-only and has no CPU path)");
    int device = device_id;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return fail(PCC_ENODEV, "hipGetDevice failed");
    if (device >= count) return fail(PCC_ENODEV, "device %d does not exist (%d visible)", device, count);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(PCC_ENODEV, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PCC_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    DeviceGuard guard(device);

    pcc_sim *sim = new (std::nothrow) pcc_sim();
    if (!sim) return fail(PCC_ENOMEM, "host allocation failed");
    memset(sim, 0, sizeof *sim);
    Dev &d = sim->d;
    d.n = n_envs; d.ns = n_senders; d.H = history_len; d.F = n_features; d.HF = history_len * n_features;
    for (int f = 0; f < n_features; f++) d.fid[f] = feature_ids[f];
    d.cap = ring_capacity; d.cap_mask = ring_capacity - 1;
    d.key0 = (uint32_t)seed; d.key1 = (uint32_t)(seed >> 32); d.gid_base = env_gid_base;
    d.delta_scale = 0.025;  // src/common/config.py:17
    d.max_steps = 400;      // ns:41
    const double lo[5] = {100, 0.05, 0, 0.0, 0.3}, hi[5] = {500, 0.5, 8, 0.05, 1.5};  // ns:355-358,466
    memcpy(d.lo, lo, sizeof lo); memcpy(d.hi, hi, sizeof hi);
    d.rng_mode = PCC_RNG_PHILOX;
    sim->device = device;
    sim->state_bytes = carve_state(d, nullptr);
    sim->ring_bytes = (size_t)n_envs * n_senders * ring_capacity * sizeof(double2);
    if (hipMalloc(&sim->state_blob, sim->state_bytes) != hipSuccess) {
        delete sim;
        return fail(PCC_ENOMEM, "hipMalloc(%zu) for env state failed", sim->state_bytes);
    }
    if (hipMalloc(&sim->ring_blob, sim->ring_bytes) != hipSuccess) {
        (void)hipFree(sim->state_blob);
        const size_t want = sim->ring_bytes;
        delete sim;
        return fail(PCC_ENOMEM, "hipMalloc(%zu) for the in-flight rings failed (n_envs*n_senders*ring_capacity*16 B)", want);
    }
    carve_state(d, static_cast<char *>(sim->state_blob));
    d.ring = static_cast<double2 *>(sim->ring_blob);
    if (hipMemset(sim->state_blob, 0, sim->state_bytes) != hipSuccess) {
        pcc_destroy(sim);
        return fail(PCC_EHIP, "hipMemset of env state failed");
    }
    *out = sim;
    return PCC_OK;
}

void pcc_destroy(pcc_sim_t *sim) {
    if (!sim) return;
    DeviceGuard guard(sim->device);
    if (sim->state_blob) (void)hipFree(sim->state_blob);
    if (sim->ring_blob) (void)hipFree(sim->ring_blob);
    delete sim;
}

int64_t pcc_device_bytes(const pcc_sim_t *sim) { return sim ? (int64_t)(sim->state_bytes + sim->ring_bytes) : 0; }

int pcc_set_link_params(pcc_sim_t *sim, const double *bw, const double *dl, const double *queue, const double *loss,
                        const double *rate0) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    const int given = (bw != nullptr) + (dl != nullptr) + (queue != nullptr) + (loss != nullptr) + (rate0 != nullptr);
    if (given != 0 && given != 5) return fail(PCC_EINVAL, "pass all five parameter arrays or none");
    sim->d.p_bw = bw; sim->d.p_dl = dl; sim->d.p_queue = queue; sim->d.p_loss = loss; sim->d.p_rate0 = rate0;
    return PCC_OK;
}

int pcc_set_param_ranges(pcc_sim_t *sim, const double *lo, const double *hi) {
    if (!sim || !lo || !hi) return fail(PCC_EINVAL, "NULL argument");
    for (int k = 0; k < 5; k++) {
        if (!(lo[k] <= hi[k])) return fail(PCC_EINVAL, "range %d is empty", k);
        sim->d.lo[k] = lo[k]; sim->d.hi[k] = hi[k];
    }
    return PCC_OK;
}

int pcc_set_rng(pcc_sim_t *sim, int mode, const double *trace, int64_t trace_stride) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    if (mode == PCC_RNG_PHILOX) {
        sim->d.rng_mode = mode; sim->d.trace = nullptr; sim->d.trace_stride = 0;
        return PCC_OK;
    }
    if (mode == PCC_RNG_TRACE) {
        if (!trace || trace_stride < 1) return fail(PCC_EINVAL, "PCC_RNG_TRACE needs a trace buffer");
        sim->d.rng_mode = mode; sim->d.trace = trace; sim->d.trace_stride = trace_stride;
        return PCC_OK;
    }
    return fail(PCC_EINVAL, "unknown rng mode %d", mode);
}

int pcc_set_seed(pcc_sim_t *sim, uint64_t seed) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    sim->d.key0 = (uint32_t)seed;
    sim->d.key1 = (uint32_t)(seed >> 32);
    return PCC_OK;
}

int pcc_set_profile_buffer(pcc_sim_t *sim, uint64_t *buf) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    sim->d.prof = reinterpret_cast<unsigned long long *>(buf);
    return PCC_OK;
}

int pcc_set_delta_scale(pcc_sim_t *sim, double delta_scale) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    sim->d.delta_scale = delta_scale;
    return PCC_OK;
}

int pcc_set_max_steps(pcc_sim_t *sim, int max_steps) {
    if (!sim || max_steps < 1) return fail(PCC_EINVAL, "max_steps must be >= 1");
    sim->d.max_steps = (uint32_t)max_steps;
    return PCC_OK;
}

static int launch_reset(pcc_sim_t *sim, const uint8_t *mask, int use_done, float *obs_out, hipStream_t st) {
    const Dev &d = sim->d;
    const dim3 grid((unsigned)((d.n + kWave - 1) / kWave)), block(kWave);
    if (d.ns == 1) hipLaunchKernelGGL(reset_kernel<1>, grid, block, 0, st, d, mask, use_done, obs_out);
    else hipLaunchKernelGGL(reset_kernel<2>, grid, block, 0, st, d, mask, use_done, obs_out);
    return check_hip(hipGetLastError(), "reset kernel launch");
}

int pcc_reset(pcc_sim_t *sim, const uint8_t *mask, float *obs_out, void *stream) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    DeviceGuard guard(sim->device);
    const int rc = launch_reset(sim, mask, 0, obs_out, static_cast<hipStream_t>(stream));
    if (rc != PCC_OK) return rc;
    if (!mask) {
        sim->ever_reset = true;
        sim->lockstep = true;
        sim->host_steps = 0;
    } else {
        sim->lockstep = false;  // some envs are now at a different step count
    }
    return PCC_OK;
}

int pcc_step(pcc_sim_t *sim, const void *actions, int actions_f64, float *obs_out, float *reward_out,
             uint8_t *done_out, double *steps_out, int auto_reset, void *stream) {
    if (!sim || !actions) return fail(PCC_EINVAL, "NULL argument");
    if (!sim->ever_reset) return fail(PCC_ESTATE, "pcc_step before pcc_reset (the reference raises TypeError: run_dur is None)");
    DeviceGuard guard(sim->device);
    const Dev &d = sim->d;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((d.n + kWave - 1) / kWave)), block(kWave);
    if (d.ns == 1)
        hipLaunchKernelGGL(step_kernel<1>, grid, block, 0, st, d, actions, actions_f64, obs_out, reward_out, done_out, steps_out);
    else
        hipLaunchKernelGGL(step_kernel<2>, grid, block, 0, st, d, actions, actions_f64, obs_out, reward_out, done_out, steps_out);
    int rc = check_hip(hipGetLastError(), "step kernel launch");
    if (rc != PCC_OK) return rc;
    sim->host_steps++;
    if (auto_reset) {
        // when every env is in lockstep the host knows which step finishes the episode and
        // skips the (otherwise no-op) masked reset launch
        const bool may_be_done = !sim->lockstep || sim->host_steps >= d.max_steps;
        if (may_be_done) {
            rc = launch_reset(sim, nullptr, 1, obs_out, st);
            if (rc != PCC_OK) return rc;
            if (sim->lockstep) sim->host_steps = 0;
        }
    } else if (sim->lockstep && sim->host_steps >= d.max_steps) {
        sim->lockstep = false;  // caller resets on its own schedule from here on
    }
    return PCC_OK;
}

int pcc_get_state(pcc_sim_t *sim, int field, void *out, void *stream) {
    if (!sim || !out) return fail(PCC_EINVAL, "NULL argument");
    const Dev &d = sim->d;
    const size_t n = (size_t)d.n, sn = n * d.ns;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (field) {
        case PCC_F_BW: src = d.bw; bytes = n * 8; break;
        case PCC_F_DL: src = d.dl; bytes = n * 8; break;
        case PCC_F_LR: src = d.lr; bytes = n * 8; break;
        case PCC_F_MAXQ: src = d.maxq; bytes = n * 8; break;
        case PCC_F_QDELAY: src = d.q; bytes = n * 8; break;
        case PCC_F_QTIME: src = d.tu; bytes = n * 8; break;
        case PCC_F_NOW: src = d.now; bytes = n * 8; break;
        case PCC_F_RUN_DUR: src = d.run_dur; bytes = n * 8; break;
        case PCC_F_STEPS: src = d.steps; bytes = n * 4; break;
        case PCC_F_EPISODE: src = d.episode; bytes = n * 4; break;
        case PCC_F_FLAGS: src = d.flags; bytes = n * 4; break;
        case PCC_F_RATE: src = d.rate; bytes = sn * 8; break;
        case PCC_F_RATE0: src = d.rate0; bytes = sn * 8; break;
        case PCC_F_NEXT_SEND: src = d.next_send; bytes = sn * 8; break;
        case PCC_F_MIN_LAT: src = d.min_lat; bytes = sn * 8; break;
        case PCC_F_RING_HEAD: src = d.h2; bytes = sn * 4; break;
        case PCC_F_RING_MID: src = d.h1; bytes = sn * 4; break;
        case PCC_F_RING_TAIL: src = d.tail; bytes = sn * 4; break;
        case PCC_F_EP_RETURN: src = d.ep_return; bytes = sn * 8; break;
        case PCC_F_LAST_RETURN: src = d.last_return; bytes = sn * 8; break;
        case PCC_F_TOTAL_SENT: src = d.total_sent; bytes = n * 8; break;
        default: return fail(PCC_EINVAL, "unknown field %d", field);
    }
    DeviceGuard guard(sim->device);
    return check_hip(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)),
                     "pcc_get_state copy");
}

}  // extern "C"
