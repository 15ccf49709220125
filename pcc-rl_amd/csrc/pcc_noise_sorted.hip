// pcc_noise_sorted.hip -- USE_LATENCY_NOISE (ns:51-52, 150-151, 171-172) for one sender without a window, WITHOUT the event
// loop: one workgroup per env runs the env's monitor interval as counts, two sorts and a scan.  The event-loop build
// (pcc_retire_env.h: event_engine) pops the reference's heap on ONE lane per env, and the longest env's ~5 000 events, ~6 us of
// dependent trips each, are the launch (33-37 ms per step at 16 384 envs).
//
// What looks sequential: every link latency is multiplied by one more draw of the env's stream, and the stream is consumed
// in EVENT order -- a SEND takes two draws (noise, loss), an arrival at the return link (hop 1) one, an arrival at the sender
// (hop 2) none -- so the draw a packet gets depends on how the events of all packets interleave, and they overtake each other.
// What is parallel all the same (restated and checked bit for bit against the oracle's event loop on the CPU:
// tests/models/noise_sorting_model.py, tests/test_noise_formulation.py):
//   * SEND times depend on nothing drawn: t_0 = the pending SEND, t_{k+1} = t_k + 1/rate.
//   * The draw index of an event inside the interval = 2 x (SENDs before it) + (hop-1 arrivals before it).
//   * A hop-1 arrival is at least dl behind its SEND (noise factor >= 1, queue delay >= 0): the arrivals before SEND k belong
//     to packets sent more than dl ago, so SENDs go in BLOCKS of ~dl/gap packets whose counts depend on earlier blocks only (a
//     binary search per packet, all lanes at once); the link's queue recurrence (ns:66-84) is a scan by one lane.
//   * The interval's hop-1 arrivals, old and new, are one sort by the reference's key (time, latency, dropped); an arrival's
//     draw index is 2 x (SENDs strictly before it: 'A' < 'S' at equal times) + its rank.  Its hop-2 time follows.
//   * The hop-2 arrivals are one more sort; those before `end` are the interval's acknowledgements and loss reports, in the
//     order the reference appends their RTTs.
//   * The event that ends the interval (the first at or after `end`, still processed: ns:128-131) is the smallest of the next
//     SEND, the first hop-1 arrival and the first hop-2 arrival not yet due.
// In-flight events live where the event loop keeps them (Dev::noise_heap, same encoding), in no particular order, with bit 31 of
// SndBlk::heap_n set (the event loop makes a heap of them first: heapify_if_loose).  An interval only looks at the events that
// are due by its last possible moment -- the next SEND at or after `end` -- so a wavefront streams through the array (once to
// count, once to take), takes those into LDS and closes the gaps in place; what is left in LDS at the end, and the interval's
// new events, are appended.  An env with thousands of packets in a deep queue costs its interval the stream, not LDS.  An
// interval whose DUE events or SENDs do not fit the arrays runs as sub-intervals in the second instance (see the kernel);
// the first instance leaves such an env alone, and an env whose array in memory could fill up is the event loop's, in the
// retire launch that follows (NoiseOut::seq says who ran an interval).  Results do not depend on who runs it.
// Speed (16 384 envs, steps 20..120 of an episode): 3.7 ms per step against the event loop's 37 (profiles/r05_v4_engine_throughput.json).
#include "pcc_retire_env.h"
#include "pcc_kernels.h"

namespace {

// events are (time, +-latency): sign of the latency = dropped (the heap's encoding with the hop bit taken off the time)
__device__ __forceinline__ bool ev_less(double ax, double ay, double bx, double by) {
    if (ax != bx) return ax < bx;
    const double la = fabs(ay), lb = fabs(by);
    if (la != lb) return la < lb;
    return !sign_of(ay) && sign_of(by);   // dropped: False < True
}

// x[0, n) (and y along with it) into the order of ev_less, by the workgroup's T threads: bitonic over the next power of two
// (the tail padded with +inf), unless the list is in order already.  The arrays hold that power of two.
template <int T>
__device__ void sort_events(double *x, double *y, uint32_t n, uint32_t tid) {
    bool bad = false;
    for (uint32_t j = tid; j + 1u < n; j += T) bad |= ev_less(x[j + 1u], y[j + 1u], x[j], y[j]);
    if (!__syncthreads_or(bad ? 1 : 0)) return;
    uint32_t P = 2u;
    while (P < n) P <<= 1;
    for (uint32_t j = n + tid; j < P; j += T) { x[j] = INFINITY; y[j] = 0.0; }
    __syncthreads();
    for (uint32_t k = 2u; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
            for (uint32_t idx = tid; idx < (P >> 1); idx += T) {
                const uint32_t lo = ((idx & ~(j - 1u)) << 1) | (idx & (j - 1u)), hi = lo | j;
                const bool up = (lo & k) == 0u;
                const double ax = x[lo], ay = y[lo], bx = x[hi], by = y[hi];
                if (ev_less(bx, by, ax, ay) == up) { x[lo] = bx; y[lo] = by; x[hi] = ax; y[hi] = ay; }
            }
            __syncthreads();
        }
    }
}
template <int T>
__device__ void sort_times(double *x, uint32_t n, uint32_t tid) {
    uint32_t P = 2u;
    while (P < n) P <<= 1;
    for (uint32_t j = n + tid; j < P; j += T) x[j] = INFINITY;
    __syncthreads();
    for (uint32_t k = 2u; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
            for (uint32_t idx = tid; idx < (P >> 1); idx += T) {
                const uint32_t lo = ((idx & ~(j - 1u)) << 1) | (idx & (j - 1u)), hi = lo | j;
                const bool up = (lo & k) == 0u;
                const double a = x[lo], b = x[hi];
                if ((b < a) == up) { x[lo] = b; x[hi] = a; }
            }
            __syncthreads();
        }
    }
}
// how many of the sorted x[0, n) are < v (lower) / <= v (upper)
__device__ __forceinline__ uint32_t count_lt(const double *x, uint32_t n, double v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (x[m] < v) lo = m + 1u; else hi = m; }
    return lo;
}
__device__ __forceinline__ uint32_t count_le(const double *x, uint32_t n, double v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (x[m] <= v) lo = m + 1u; else hi = m; }
    return lo;
}

struct DrawCtx {
    const double *trace_row;   // nullptr: Philox
    int64_t trace_stride;
    uint32_t ep0, key0, key1, gid, episode, mi;
};
// the idx-th draw of the interval (event_engine: draw())
__device__ __forceinline__ double draw_at(const DrawCtx &c, uint32_t idx, uint32_t &flags) {
    if (c.trace_row) {
        const uint32_t pos = c.ep0 + idx;
        if ((int64_t)pos >= c.trace_stride) { flags |= PCC_FLAG_TRACE_OVERRUN; return 1.0; }
        return c.trace_row[pos];
    }
    uint32_t w[4];
    philox4x32_10(idx >> 2, c.mi, c.episode, c.gid, c.key0, c.key1, w);
    const uint32_t x = idx & 3u;
    return u32_to_unit(x == 0 ? w[0] : x == 1 ? w[1] : x == 2 ? w[2] : w[3]);
}

struct LinkState { double q, tu; };
// the link's part of one SEND at time t with its two draws (event_engine's SEND branch, ns:155-175): the hop-1 event
__device__ __forceinline__ void send_one(double t, double u_noise, double u_loss, double dl, double lr, double maxq, double ebw,
                                         double span, LinkState &L, double &ax, double &ay) {
    const double qd = max0(L.q - (t - L.tu));
    double ll = dl + qd;
    ll *= 1.0 + span * u_noise;              // drawn before the loss decision (ns:171-175)
    const double lat = 0.0 + ll;
    bool dropped;
    if (u_loss < lr) dropped = true;         // ns:73-74
    else {
        L.q = qd; L.tu = t;                  // ns:75-76
        if (ebw + L.q > maxq) dropped = true;   // ns:78-79
        else { L.q += ebw; dropped = false; }
    }
    ax = t + ll;
    ay = dropped ? -lat : lat;
}

// By ONE wavefront.  The events that are due by `bound` go from the env's array H[0, n) to LDS -- hop 1 to (x1, y1)[0, m1),
// hop 2 to (x2, y2)[0, m2), times positive -- and the others move up in place: returns how many stay in H.  Batches of 64,
// four in flight: they are in registers before anything is written, and what is written lies below what is still to be read.
__device__ __forceinline__ uint32_t take_due(double2 *H, uint32_t n, double bound, double *x1, double *y1, uint32_t &m1,
                                             double *x2, double *y2, uint32_t &m2, uint32_t lane) {
    uint32_t keep = 0;
    m1 = m2 = 0;
    constexpr int kDeep = 4;
    for (uint32_t base = 0; base < n; base += kDeep * kWave) {
        double2 evs[kDeep];
#pragma unroll
        for (int b = 0; b < kDeep; b++) {
            const uint32_t j = base + (uint32_t)b * kWave + lane;
            evs[b].x = 0.0; evs[b].y = 0.0;
            if (j < n) evs[b] = ld_rec(heap_node(H, j));
        }
#pragma unroll
        for (int b = 0; b < kDeep; b++) {
            const uint32_t j = base + (uint32_t)b * kWave + lane;
            const bool in = j < n;
            const double2 ev = evs[b];
            const bool due = in && fabs(ev.x) <= bound;
            const bool t1 = due && !sign_of(ev.x), t2 = due && sign_of(ev.x), stay = in && !due;
            const uint64_t b1 = __ballot(t1), b2 = __ballot(t2), bs = __ballot(stay);
            if (t1) { const uint32_t p = m1 + count_below(b1); x1[p] = ev.x; y1[p] = ev.y; }
            if (t2) { const uint32_t p = m2 + count_below(b2); x2[p] = -ev.x; y2[p] = ev.y; }
            if (stay) st_rec(heap_node(H, keep + count_below(bs)), ev);
            m1 += (uint32_t)__popcll(b1);
            m2 += (uint32_t)__popcll(b2);
            keep += (uint32_t)__popcll(bs);
        }
    }
    return keep;
}

// T threads per env (the sorts, the draws and the hop-1 arrivals are theirs; the streams over the env's array and the
// acknowledgement list are the first wavefront's, the two scans the first lane's).
// SPLIT: an interval whose due events or SENDs do not fit the arrays is run as several sub-intervals, one after the other
// -- all events before a SEND time, then on from there: the event order is a time order, the draw index carries over --
// each with at most CAPK SENDs and as few of them as makes its due events fit.  !SPLIT: such an env is left alone.
template <int CAP, int CAPK, int CHUNK, bool SPLIT, int T>
__global__ __launch_bounds__(T) void noise_sorted_kernel(Dev D, int warm, uint32_t warm_mi, int gate, const void *actions, int actions_f64) {
    if (gate && __hip_atomic_load(D.any_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != D.step_seq) return;
    const int64_t i = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1u);
    const bool wave0 = tid < (uint32_t)kWave;
    if (warm && !D.env[i].resetting) return;
    NoiseOut *const out = D.noise_out + i;
    if (out->seq == D.noise_seq) return;      // a smaller instance has run this env's interval already

    __shared__ double e1x[CAP], e1y[CAP], e2x[CAP], e2y[CAP], ts[CAPK + 1], news[CAPK], un[CHUNK], ul[CHUNK];
    __shared__ uint32_t s_k, s_a, s_b, s_c;
    __shared__ double s_q, s_tu, s_now, s_nsend, s_xx, s_xy;
    __shared__ uint32_t s_kind, s_flags;

    const int64_t k0s = sidx(D, 0, i);
    double rate = D.snd[k0s].rate;
    if (!warm) {   // apply_rate_delta (ns:235-241, 275-281): as the retire launch does it for the same env (it stores the rate)
        double delta = actions_f64 ? ((const double *)actions)[i] : (double)((const float *)actions)[i];
        if (delta != delta) delta = 0.0;
        delta *= D.delta_scale;
        rate = delta >= 0.0 ? rate * (1.0 + delta) : rate / (1.0 - delta);
        if (rate > kMaxRate) rate = kMaxRate;
        if (rate < kMinRate) rate = kMinRate;
    }
    const double dl = D.env[i].dl, lr = D.env[i].lr, maxq = D.env[i].maxq, ebw = D.env[i].ebw, span = D.noise_span;
    const double start = D.env[i].now, end = start + D.env[i].run_dur;
    const double gap = 1.0 / rate;
    // A latency below the clock's resolution (t + dl == t: pcc_set_param_ranges only asks for dl > 0) would let a packet's arrival
    // at the return link tie with its own SEND, which the event loop runs first -- and the count of "SENDs strictly before the
    // arrival" below would miss it (two draws).  Such an env is the event loop's (nothing is written yet).
    if (!(end + 0.5 * dl > end)) return;
    const double nsend0 = D.snd[k0s].next_send;
    const uint32_t n_old = D.snd[k0s].heap_n & 0x7FFFFFFFu, noise_cap = D.noise_cap;
    double2 *const H = D.noise_heap + (size_t)i * (noise_cap + kHeapPad);
    double2 *const R = D.noise_rtt + (size_t)i * noise_cap;
    DrawCtx dc;
    dc.trace_row = D.rng_mode == PCC_RNG_TRACE ? D.trace + i * D.trace_stride : nullptr;
    dc.trace_stride = D.trace_stride;
    dc.ep0 = D.env[i].ep_draws; dc.key0 = D.key0; dc.key1 = D.key1;
    dc.gid = D.gid_base + (uint32_t)env_of(D, i);
    dc.episode = D.env[i].episode - 1;
    dc.mi = warm ? warm_mi : D.env[i].steps + 2;
    uint32_t flags = 0;

    if (!(start < end)) {   // ns:128: the loop body never runs
        if (tid == 0) {
            D.env[i].mi_draws = 0;
            out->now = start; out->q = D.env[i].q; out->tu = D.env[i].tu; out->nsend = nsend0;
            out->sent = out->acked = out->lost = 0; out->flags = 0;
            out->seq = D.noise_seq;
        }
        return;
    }

    // what the sub-intervals hand on
    uint32_t n_heap = n_old, draws = 0, sent = 0, acked = 0, lost = 0, kind = 0;
    double cur_send = nsend0;
    bool first = true, no_sends_before = false, give_up = false;
    if (tid == 0) { s_q = D.env[i].q; s_tu = D.env[i].tu; s_now = start; s_nsend = nsend0; }
    const double ratio = dl / gap;
    const uint32_t W = ratio >= 3.0 ? (ratio < 1e9 ? (uint32_t)ratio - 1u : 0x3FFFFFFFu) : 1u;   // SENDs per block, see below

    for (;;) {
        // ---- the SEND times of this sub-interval (the reference's own additions): up to CAPK of them before `end`
        if (tid == 0) {
            uint32_t K = 0;
            double t = cur_send;
            while (t < end && K < (uint32_t)CAPK) { ts[K++] = t; t = t + gap; }
            ts[K] = t;
            s_k = K;
        }
        __syncthreads();
        uint32_t K = s_k;
        double t_after = ts[K];            // the first SEND that is not this sub-interval's
        bool final = !(t_after < end);     // ... is the interval's last possible moment: the sub-interval ends the interval
        if (!SPLIT && !final) return;      // (nothing written yet)
        // ---- does what is due by then fit?  (hop 1: the due ones and the sub-interval's own; hop 2: the due ones and one per
        // hop-1 event.)  If not, fewer SENDs: a shorter sub-interval.
        for (;;) {
            __syncthreads();
            if (wave0) {
                uint32_t due1 = 0, due2 = 0;
                for (uint32_t base = 0; base < n_heap; base += 4u * kWave) {
                    double x[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t j = base + (uint32_t)b * kWave + lane;
                        x[b] = j < n_heap ? ld_t1(heap_node(H, j)) : INFINITY;
                    }
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const bool due = fabs(x[b]) <= t_after;
                        due1 += (uint32_t)__popcll(__ballot(due && !sign_of(x[b])));
                        due2 += (uint32_t)__popcll(__ballot(due && sign_of(x[b])));
                    }
                }
                if (lane == 0) { s_a = due1; s_b = due2; }
            }
            __syncthreads();
            const uint32_t due1 = s_a, due2 = s_b;
            // (and the env's array in memory, which has the reference's size: every SEND adds an event, nothing else does;
            // an interval that could fill it is the event loop's, which raises the flag at the push that does)
            const bool room = n_heap + K + 2u < noise_cap;
            if (room && due1 + K + 1u <= (uint32_t)CAP && due2 + due1 + K + 1u <= (uint32_t)CAP) break;
            if (!SPLIT || K == 0u) {
                if (first) return;                      // (nothing written yet: a larger instance, or the event loop)
                flags |= room ? PCC_FLAG_INTERNAL       // more events due inside one gap between SENDs than the arrays hold
                              : PCC_FLAG_RING_OVERFLOW; // the env's array is full
                give_up = true;
                break;
            }
            K >>= 1;
            t_after = ts[K];
            final = false;
        }
        if (!final && K == 0u) {   // (a sub-interval without a SEND must not follow one: nothing would move)
            if (no_sends_before) flags |= PCC_FLAG_INTERNAL;
            no_sends_before = true;
        } else no_sends_before = false;
        if (give_up || (flags & PCC_FLAG_INTERNAL)) break;
        const double lim = final ? end : t_after;       // events before `lim` are this sub-interval's

        // ---- the events that are due, sorted by hop
        __syncthreads();
        if (wave0) {
            uint32_t m1, m2;
            const uint32_t kp = take_due(H, n_heap, t_after, e1x, e1y, m1, e2x, e2y, m2, lane);
            if (lane == 0) { s_a = m1; s_b = m2; s_c = kp; }
        }
        __syncthreads();
        uint32_t n1 = s_a, n2 = s_b;
        const uint32_t n_keep = s_c;
        sort_events<T>(e1x, e1y, n1, tid);
        sort_events<T>(e2x, e2y, n2, tid);
        __syncthreads();
        const uint32_t n1_old = n1, n2_old = n2;

        // ---- SENDs in blocks of W: the hop-1 arrivals before SEND k = old ones <= t_k + new ones of EARLIER blocks <= t_k
        // (a block's own arrivals lie behind its last SEND: a_j >= t_j + dl and (W - 1) gaps < dl with two gaps to spare)
        uint32_t n_news = 0;   // arrivals of earlier blocks, sorted, in news[]
        for (uint32_t b0 = 0; b0 < K; b0 += W) {
            const uint32_t b1 = (K - b0 > W) ? b0 + W : K;
            for (uint32_t c0 = b0; c0 < b1; c0 += (uint32_t)CHUNK) {
                const uint32_t c1 = (b1 - c0 > (uint32_t)CHUNK) ? c0 + (uint32_t)CHUNK : b1;
                for (uint32_t k = c0 + tid; k < c1; k += T) {
                    const double tk = ts[k];
                    const uint32_t idx = draws + 2u * k + count_le(e1x, n1_old, tk) + count_le(news, n_news, tk);
                    un[k - c0] = draw_at(dc, idx, flags);
                    ul[k - c0] = draw_at(dc, idx + 1u, flags);
                }
                __syncthreads();
                if (tid == 0) {   // the queue recurrence: a scan
                    LinkState L; L.q = s_q; L.tu = s_tu;
                    for (uint32_t k = c0; k < c1; k++) {
                        double ax, ay;
                        send_one(ts[k], un[k - c0], ul[k - c0], dl, lr, maxq, ebw, span, L, ax, ay);
                        e1x[n1_old + k] = ax; e1y[n1_old + k] = ay;
                    }
                    s_q = L.q; s_tu = L.tu;
                }
                __syncthreads();
            }
            if (b1 < K) {   // the next block searches these arrivals too
                for (uint32_t k = b0 + tid; k < b1; k += T) news[k] = e1x[n1_old + k];
                n_news = b1;
                __syncthreads();
                sort_times<T>(news, n_news, tid);
                __syncthreads();
            }
        }
        n1 = n1_old + K;
        __syncthreads();
        sort_events<T>(e1x, e1y, n1, tid);
        __syncthreads();

        // ---- hop-1 arrivals before `lim`: a draw each, in their order; their hop-2 events
        const uint32_t n1p = count_lt(e1x, n1, lim);
        for (uint32_t j = tid; j < n1p; j += T) {
            const double a = e1x[j], y = e1y[j];
            const uint32_t idx = draws + 2u * count_lt(ts, K, a) + j;   // SENDs strictly before it ('A' < 'S'), arrivals before it
            double ll = dl + max0(0.0 - (a - 0.0));                     // the return link never queues (ns:147-153)
            ll *= 1.0 + span * draw_at(dc, idx, flags);
            const double lat = fabs(y) + ll;
            e2x[n2_old + j] = a + ll;
            e2y[n2_old + j] = sign_of(y) ? -lat : lat;
        }
        n2 = n2_old + n1p;
        __syncthreads();
        sort_events<T>(e2x, e2y, n2, tid);
        __syncthreads();

        // ---- hop-2 arrivals before `lim`: acknowledgements (their RTTs in this order) and loss reports
        const uint32_t n2p = count_lt(e2x, n2, lim);
        if (wave0) {
            uint32_t ak = acked, ls = lost, fl = 0;
            for (uint32_t base = 0; base < n2p; base += kWave) {
                const uint32_t j = base + lane;
                const bool in = j < n2p;
                const double y = in ? e2y[j] : 0.0;
                const bool ok = in && !sign_of(y), bad = in && sign_of(y);
                const uint64_t mo = __ballot(ok), mb = __ballot(bad);
                if (ok) {
                    const uint32_t p = ak + count_below(mo);
                    if (p < noise_cap) { double2 r; r.x = 0.0; r.y = y; st_rec(R + p, r); }
                    else fl |= PCC_FLAG_RING_OVERFLOW;
                }
                ak += (uint32_t)__popcll(mo);
                ls += (uint32_t)__popcll(mb);
            }
            flags |= fl;
            if (lane == 0) { s_a = ak; s_b = ls; }
        }
        __syncthreads();
        acked = s_a; lost = s_b;

        // ---- the last sub-interval: the event that ends the interval, (time, 'A' < 'S', hop, latency, dropped)
        kind = 3;   // 0: the next SEND, 1: a hop-1 arrival, 2: a hop-2 arrival, 3: none (the interval goes on)
        if (final) {
            if (tid == 0) {
                uint32_t kd = 0;
                double tb = t_after;
                if (n1p < n1 && e1x[n1p] <= tb) { kd = 1; tb = e1x[n1p]; }
                if (n2p < n2) {
                    const double b = e2x[n2p];
                    if (kd == 0 ? b <= tb : b < tb) { kd = 2; tb = b; }
                }
                LinkState L; L.q = s_q; L.tu = s_tu;
                double nsend = t_after, xx = 0.0, xy = 0.0;
                uint32_t fl = 0;
                if (kd == 0) {
                    const uint32_t idx = draws + 2u * K + n1p;
                    const double u0 = draw_at(dc, idx, fl), u1 = draw_at(dc, idx + 1u, fl);
                    send_one(t_after, u0, u1, dl, lr, maxq, ebw, span, L, xx, xy);   // a hop-1 event, not due
                    nsend = t_after + gap;   // ns:161
                } else if (kd == 1) {
                    const double a = e1x[n1p], y = e1y[n1p];
                    double ll = dl + max0(0.0 - (a - 0.0));
                    ll *= 1.0 + span * draw_at(dc, draws + 2u * K + n1p, fl);
                    const double lat = fabs(y) + ll;
                    xx = a + ll;            // a hop-2 event, not due
                    xy = sign_of(y) ? -lat : lat;
                } else {
                    const double y = e2y[n2p];
                    xx = sign_of(y) ? 1.0 : 0.0;   // (a loss report, or an acknowledgement: its RTT is the interval's last)
                    if (!sign_of(y)) {
                        if (acked < noise_cap) { double2 r; r.x = 0.0; r.y = y; st_rec(R + acked, r); }
                        else fl |= PCC_FLAG_RING_OVERFLOW;
                    }
                }
                s_xx = xx; s_xy = xy;
                s_kind = kd; s_now = tb; s_nsend = nsend; s_q = L.q; s_tu = L.tu; s_flags = fl;
            }
            __syncthreads();
            kind = s_kind;
            flags |= s_flags;
            if (kind == 2) { if (s_xx != 0.0) lost++; else acked++; }
        }

        // ---- what was due and is still in flight, and the new events, go back behind what stayed (in no particular order)
        const uint32_t f1 = n1p + (kind == 1 ? 1u : 0u), f2 = n2p + (kind == 2 ? 1u : 0u);
        uint32_t r1 = n1 - f1, r2 = n2 - f2;
        if (n_keep + r1 + r2 + 2u > noise_cap) {   // (cannot happen after the test above; never past the env's array)
            flags |= PCC_FLAG_RING_OVERFLOW;
            r1 = 0; r2 = 0;
        }
        for (uint32_t j = tid; j < r1; j += T) { double2 r; r.x = e1x[f1 + j]; r.y = e1y[f1 + j]; st_rec(heap_node(H, n_keep + j), r); }
        for (uint32_t j = tid; j < r2; j += T) { double2 r; r.x = -e2x[f2 + j]; r.y = e2y[f2 + j]; st_rec(heap_node(H, n_keep + r1 + j), r); }
        n_heap = n_keep + r1 + r2;
        if (kind == 0 || kind == 1) {
            if (tid == 0) { double2 r; r.x = kind == 0 ? s_xx : -s_xx; r.y = s_xy; st_rec(heap_node(H, n_heap), r); }
            n_heap++;
        }
        draws += 2u * K + n1p + (kind == 0 ? 2u : kind == 1 ? 1u : 0u);
        sent += K + (kind == 0 ? 1u : 0u);
        cur_send = t_after;
        first = false;
        __threadfence_block();
        __syncthreads();   // (the arrays, and the env's array in memory, are the next sub-interval's)
        if (final) break;
    }

    // every thread's flags
    for (uint32_t bit = 1u; bit <= PCC_FLAG_INTERNAL; bit <<= 1)
        if (__syncthreads_or((flags & bit) != 0u ? 1 : 0)) flags |= bit;
    if (tid == 0) {
        D.snd[k0s].heap_n = n_heap | 0x80000000u;
        D.env[i].ep_draws = dc.ep0 + draws;
        D.env[i].mi_draws = dc.trace_row ? 0u : draws;
        out->now = s_now; out->q = s_q; out->tu = s_tu; out->nsend = s_nsend;
        out->sent = sent; out->acked = acked; out->lost = lost; out->flags = flags;
        out->seq = D.noise_seq;
    }
}


// =====================================================================================================================
// Two senders on the link (round 6).  The reference orders events as (time, sender, 'A' < 'S', hop, latency, dropped) -- the
// sender id comes BEFORE the kind of event -- and both senders' events consume ONE draw stream, so (restated in numpy and
// checked bit for bit against the oracle's event loop: tests/models/noise_sorting2_model.py, tests/test_noise_formulation.py):
//   * the SENDs of both senders are one sequence merged by (time, sender); the link's queue recurrence scans it in that order;
//   * a hop-1 arrival (a, s') comes before the SEND (t, s) iff (a, s') <= (t, s) lexicographically; a SEND (t, s) before the
//     arrival (a, s') iff (t, s) < (a, s');
//   * the draw index of an event = 2 x (SENDs of either sender before it) + (hop-1 arrivals of either sender before it);
//   * hop-1 and hop-2 arrivals sort by (time, sender, latency, dropped); acknowledgements and RTT lists are per sender, in
//     that order; blocks of SENDs are cut by TIME: everything sent less than 0.9 dl after the block's first SEND (an arrival
//     is at least dl behind its SEND: none of the block's own can come before a SEND of the block).
// One instance, no sub-intervals: an env whose due events or SENDs do not fit the arrays, or whose arrays in memory could
// fill up, is left to the event loop of the retire launch (NoiseOut::seq says who ran the interval); results do not depend on it.
// Events in flight stay in the senders' own arrays (Dev::noise_heap [S][N]), loose (bit 31 of SndBlk::heap_n).
// NoiseOut: entry i = sender 0's counts and the env's clock / link state / flags / seq; entry N + i = sender 1's counts.
// =====================================================================================================================
__device__ __forceinline__ bool ev_less2(double ax, uint32_t as, double ay, double bx, uint32_t bs, double by) {
    if (ax != bx) return ax < bx;
    if (as != bs) return as < bs;
    const double la = fabs(ay), lb = fabs(by);
    if (la != lb) return la < lb;
    return !sign_of(ay) && sign_of(by);   // dropped: False < True
}
// (x, s, y)[0, n) into the order of ev_less2 by the workgroup's T threads (bitonic over the next power of two, +inf padding)
template <int T>
__device__ void sort_events2(double *x, uint32_t *sd, double *y, uint32_t n, uint32_t tid) {
    bool bad = false;
    for (uint32_t j = tid; j + 1u < n; j += T) bad |= ev_less2(x[j + 1u], sd[j + 1u], y[j + 1u], x[j], sd[j], y[j]);
    if (!__syncthreads_or(bad ? 1 : 0)) return;
    uint32_t P = 2u;
    while (P < n) P <<= 1;
    for (uint32_t j = n + tid; j < P; j += T) { x[j] = INFINITY; sd[j] = 0u; y[j] = 0.0; }
    __syncthreads();
    for (uint32_t k = 2u; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
            for (uint32_t idx = tid; idx < (P >> 1); idx += T) {
                const uint32_t lo = ((idx & ~(j - 1u)) << 1) | (idx & (j - 1u)), hi = lo | j;
                const bool up = (lo & k) == 0u;
                const double ax = x[lo], ay = y[lo], bx = x[hi], by = y[hi];
                const uint32_t as = sd[lo], bs = sd[hi];
                if (ev_less2(bx, bs, by, ax, as, ay) == up) { x[lo] = bx; y[lo] = by; sd[lo] = bs; x[hi] = ax; y[hi] = ay; sd[hi] = as; }
            }
            __syncthreads();
        }
    }
}
// (time, sender) pairs by (time, sender)
template <int T>
__device__ void sort_pairs(double *x, uint32_t *sd, uint32_t n, uint32_t tid) {
    uint32_t P = 2u;
    while (P < n) P <<= 1;
    for (uint32_t j = n + tid; j < P; j += T) { x[j] = INFINITY; sd[j] = 0u; }
    __syncthreads();
    for (uint32_t k = 2u; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
            for (uint32_t idx = tid; idx < (P >> 1); idx += T) {
                const uint32_t lo = ((idx & ~(j - 1u)) << 1) | (idx & (j - 1u)), hi = lo | j;
                const bool up = (lo & k) == 0u;
                const double a = x[lo], b = x[hi];
                const uint32_t as = sd[lo], bs = sd[hi];
                const bool b_first = b < a || (b == a && bs < as);
                if (b_first == up) { x[lo] = b; sd[lo] = bs; x[hi] = a; sd[hi] = as; }
            }
            __syncthreads();
        }
    }
}
// how many of the (time, sender)-sorted pairs are < (v, vs) / <= (v, vs), lexicographically
__device__ __forceinline__ uint32_t count_pairs_lt(const double *x, const uint32_t *sd, uint32_t n, double v, uint32_t vs) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (x[m] < v || (x[m] == v && sd[m] < vs)) lo = m + 1u; else hi = m; }
    return lo;
}
__device__ __forceinline__ uint32_t count_pairs_le(const double *x, const uint32_t *sd, uint32_t n, double v, uint32_t vs) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (x[m] < v || (x[m] == v && sd[m] <= vs)) lo = m + 1u; else hi = m; }
    return lo;
}

// sender `sdr`'s events that are due by `bound` from its array H[0, n) to LDS, tagged; the others move up in place (one wavefront)
__device__ __forceinline__ uint32_t take_due2(double2 *H, uint32_t n, double bound, uint32_t sdr, double *x1, uint32_t *s1, double *y1, uint32_t &m1,
                                              double *x2, uint32_t *s2, double *y2, uint32_t &m2, uint32_t lane) {
    uint32_t keep = 0;
    constexpr int kDeep = 4;
    for (uint32_t base = 0; base < n; base += kDeep * kWave) {
        double2 evs[kDeep];
#pragma unroll
        for (int b = 0; b < kDeep; b++) {
            const uint32_t j = base + (uint32_t)b * kWave + lane;
            evs[b].x = 0.0; evs[b].y = 0.0;
            if (j < n) evs[b] = ld_rec(heap_node(H, j));
        }
#pragma unroll
        for (int b = 0; b < kDeep; b++) {
            const uint32_t j = base + (uint32_t)b * kWave + lane;
            const bool in = j < n;
            const double2 ev = evs[b];
            const bool due = in && fabs(ev.x) <= bound;
            const bool t1 = due && !sign_of(ev.x), t2 = due && sign_of(ev.x), stay = in && !due;
            const uint64_t b1 = __ballot(t1), b2 = __ballot(t2), bs = __ballot(stay);
            if (t1) { const uint32_t p = m1 + count_below(b1); x1[p] = ev.x; s1[p] = sdr; y1[p] = ev.y; }
            if (t2) { const uint32_t p = m2 + count_below(b2); x2[p] = -ev.x; s2[p] = sdr; y2[p] = ev.y; }
            if (stay) st_rec(heap_node(H, keep + count_below(bs)), ev);
            m1 += (uint32_t)__popcll(b1);
            m2 += (uint32_t)__popcll(b2);
            keep += (uint32_t)__popcll(bs);
        }
    }
    return keep;
}

// SPLIT as in the one-sender kernel: the first instance (one wavefront per env) takes the envs whose interval fits its arrays in one go
// and leaves the others alone; the second runs those as sub-intervals.
template <int CAP, int CAPK, int CHUNK, bool SPLIT, int T>
__global__ __launch_bounds__(T) void noise_sorted2_kernel(Dev D, int warm, uint32_t warm_mi, int gate, const void *actions, int actions_f64) {
    if (gate && __hip_atomic_load(D.any_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != D.step_seq) return;
    const int64_t i = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1u);
    const bool wave0 = tid < (uint32_t)kWave;
    if (warm && !D.env[i].resetting) return;
    NoiseOut *const out0 = D.noise_out + i, *const out1 = D.noise_out + D.n + i;
    if (out0->seq == D.noise_seq) return;

    constexpr uint32_t kShare = (uint32_t)CAPK / 2u;   // SENDs of one sender per sub-interval
    __shared__ double e1x[CAP], e1y[CAP], e2x[CAP], e2y[CAP], mt[CAPK + 2], nx[CAPK], un[CHUNK], ul[CHUNK], tsd[2][kShare + 1];
    __shared__ uint32_t e1s[CAP], e2s[CAP], ms[CAPK + 2], nsd[CAPK];
    __shared__ uint32_t s_a, s_b, s_c, s_d, s_cnt[2], s_gen[2], s_kind, s_flags, s_ks, s_ak[2], s_ls[2];
    __shared__ double s_q, s_tu, s_now, s_xx, s_xy, s_nsend[2];

    double rate[2], gap[2], cur[2];
    uint32_t n_heap[2];
    double2 *H[2], *R[2];
    const uint32_t noise_cap = D.noise_cap;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int64_t ks = sidx(D, s, i);
        rate[s] = D.snd[ks].rate;
        if (!warm) {   // apply_rate_delta (ns:235-241, 275-281): as the retire launch does it for the same env (it stores the rate)
            const int64_t a = i * 2 + s;
            double delta = actions_f64 ? ((const double *)actions)[a] : (double)((const float *)actions)[a];
            if (delta != delta) delta = 0.0;
            delta *= D.delta_scale;
            rate[s] = delta >= 0.0 ? rate[s] * (1.0 + delta) : rate[s] / (1.0 - delta);
            if (rate[s] > kMaxRate) rate[s] = kMaxRate;
            if (rate[s] < kMinRate) rate[s] = kMinRate;
        }
        gap[s] = 1.0 / rate[s];
        cur[s] = D.snd[ks].next_send;
        n_heap[s] = D.snd[ks].heap_n & 0x7FFFFFFFu;
        const int64_t kh = (int64_t)s * D.n + i;   // (the event arrays are [S][N])
        H[s] = D.noise_heap + (size_t)kh * (noise_cap + kHeapPad);
        R[s] = D.noise_rtt + (size_t)kh * noise_cap;
    }
    const double nsend_in[2] = {cur[0], cur[1]};
    const double dl = D.env[i].dl, lr = D.env[i].lr, maxq = D.env[i].maxq, ebw = D.env[i].ebw, span = D.noise_span;
    const double start = D.env[i].now, end = start + D.env[i].run_dur;
    if (!(end + 0.5 * dl > end)) return;   // (a latency below the clock's resolution: the event loop's, see the one-sender kernel)
    DrawCtx dc;
    dc.trace_row = D.rng_mode == PCC_RNG_TRACE ? D.trace + i * D.trace_stride : nullptr;
    dc.trace_stride = D.trace_stride;
    dc.ep0 = D.env[i].ep_draws; dc.key0 = D.key0; dc.key1 = D.key1;
    dc.gid = D.gid_base + (uint32_t)env_of(D, i);
    dc.episode = D.env[i].episode - 1;
    dc.mi = warm ? warm_mi : D.env[i].steps + 2;
    uint32_t flags = 0;

    if (!(start < end)) {   // ns:128: the loop body never runs
        if (tid == 0) {
            D.env[i].mi_draws = 0;
            out0->now = start; out0->q = D.env[i].q; out0->tu = D.env[i].tu; out0->nsend = nsend_in[0]; out1->nsend = nsend_in[1];
            out0->sent = out0->acked = out0->lost = 0; out1->sent = out1->acked = out1->lost = 0; out0->flags = 0;
            out0->seq = D.noise_seq;
        }
        return;
    }

    // what the sub-intervals hand on
    uint32_t draws = 0, sent[2] = {0u, 0u}, acked[2] = {0u, 0u}, lost[2] = {0u, 0u}, kind = 3u, ksd = 0u;
    bool first = true, no_sends_before = false, give_up = false;
    if (tid == 0) { s_q = D.env[i].q; s_tu = D.env[i].tu; s_now = start; s_nsend[0] = cur[0]; s_nsend[1] = cur[1]; }

    for (;;) {
        // ---- the SEND times of this sub-interval, per sender (the reference's own additions): up to kShare of each before `end`
        __syncthreads();
        if (tid < 2u) {
            const uint32_t s = tid;
            uint32_t k = 0;
            double t = cur[s];
            while (t < end && k < kShare) { tsd[s][k++] = t; t = t + gap[s]; }
            tsd[s][k] = t;       // the sender's next SEND: at or after `end`, or the first that did not fit
            s_gen[s] = k;
        }
        __syncthreads();
        const uint32_t g0 = s_gen[0], g1 = s_gen[1];
        // the sub-interval's boundary in event order: `end`, or the first SEND that did not fit -- (time, sender), the earlier one
        bool final = !(tsd[0][g0] < end) && !(tsd[1][g1] < end);
        double lim_t = end;
        uint32_t lim_s = 0u;
        if (!final) {
            const bool o0 = tsd[0][g0] < end, o1 = tsd[1][g1] < end;
            if (o0 && (!o1 || tsd[0][g0] <= tsd[1][g1])) { lim_t = tsd[0][g0]; lim_s = 0u; }
            else { lim_t = tsd[1][g1]; lim_s = 1u; }
        }
        // the senders' SENDs before the boundary: (t, s) < (lim_t, lim_s) (every one of them when the boundary is `end`)
        uint32_t Ks[2];
        Ks[0] = final ? g0 : (lim_s == 0u ? count_lt(tsd[0], g0, lim_t) : count_le(tsd[0], g0, lim_t));
        Ks[1] = final ? g1 : count_lt(tsd[1], g1, lim_t);
        // (sender 0's SEND at exactly lim_t comes before (lim_t, 1) and is not before (lim_t, 0); sender 1's at lim_t never is)
        for (uint32_t k = tid; k < Ks[0]; k += T) { mt[k] = tsd[0][k]; ms[k] = 0u; }
        for (uint32_t k = tid; k < Ks[1]; k += T) { mt[Ks[0] + k] = tsd[1][k]; ms[Ks[0] + k] = 1u; }
        uint32_t K = Ks[0] + Ks[1];
        if (!SPLIT && !final) return;      // (nothing written yet: the larger instance's)
        __syncthreads();
        sort_pairs<T>(mt, ms, K, tid);
        __syncthreads();
        // ---- does what is due by the boundary fit the arrays, and have the senders' arrays in memory room?  If not, fewer SENDs.
        double t_bound = final ? (tsd[0][g0] <= tsd[1][g1] ? tsd[0][g0] : tsd[1][g1]) : lim_t;
        for (;;) {
            __syncthreads();
            if (wave0) {
                uint32_t due1 = 0, due2 = 0;
#pragma unroll
                for (int s = 0; s < 2; s++)
                    for (uint32_t base = 0; base < n_heap[s]; base += kWave) {
                        const uint32_t j = base + lane;
                        const double x = j < n_heap[s] ? ld_t1(heap_node(H[s], j)) : INFINITY;
                        const bool due = fabs(x) <= t_bound;
                        due1 += (uint32_t)__popcll(__ballot(due && !sign_of(x)));
                        due2 += (uint32_t)__popcll(__ballot(due && sign_of(x)));
                    }
                if (lane == 0) { s_a = due1; s_b = due2; }
            }
            __syncthreads();
            const uint32_t due1 = s_a, due2 = s_b;
            const bool room = n_heap[0] + Ks[0] + 2u < noise_cap && n_heap[1] + Ks[1] + 2u < noise_cap;
            if (room && due1 + K + 1u <= (uint32_t)CAP && due2 + due1 + K + 1u <= (uint32_t)CAP) break;
            if (!SPLIT || K == 0u) {
                if (first) return;                      // (nothing written yet: the larger instance's, or the event loop's)
                flags |= room ? PCC_FLAG_INTERNAL : PCC_FLAG_RING_OVERFLOW;
                give_up = true;
                break;
            }
            // half the merged SENDs: the boundary is the first one left out
            K >>= 1;
            lim_t = mt[K]; lim_s = ms[K];
            final = false;
            t_bound = lim_t;
            if (wave0) {   // the senders' shares of the shorter prefix
                uint32_t c1 = 0;
                for (uint32_t base = 0; base < K; base += kWave) c1 += (uint32_t)__popcll(__ballot(base + lane < K && ms[base + lane] == 1u));
                if (lane == 0) s_c = c1;
            }
            __syncthreads();
            Ks[1] = s_c; Ks[0] = K - Ks[1];
        }
        if (!final && K == 0u) {   // (a sub-interval without a SEND must not follow one: nothing would move)
            if (no_sends_before) flags |= PCC_FLAG_INTERNAL;
            no_sends_before = true;
        } else no_sends_before = false;
        if (give_up || (flags & PCC_FLAG_INTERNAL)) break;

        // ---- the events that are due, by hop, tagged with their sender; the others close up in place
        __syncthreads();
        if (wave0) {
            uint32_t m1 = 0, m2 = 0;
            const uint32_t kp0 = take_due2(H[0], n_heap[0], t_bound, 0u, e1x, e1s, e1y, m1, e2x, e2s, e2y, m2, lane);
            const uint32_t kp1 = take_due2(H[1], n_heap[1], t_bound, 1u, e1x, e1s, e1y, m1, e2x, e2s, e2y, m2, lane);
            if (lane == 0) { s_a = m1; s_b = m2; s_c = kp0; s_d = kp1; }
        }
        __syncthreads();
        uint32_t n1 = s_a, n2 = s_b;
        const uint32_t keep0 = s_c, keep1 = s_d;
        sort_events2<T>(e1x, e1s, e1y, n1, tid);
        sort_events2<T>(e2x, e2s, e2y, n2, tid);
        __syncthreads();
        const uint32_t n1_old = n1, n2_old = n2;

        // ---- SENDs in blocks cut by time: the hop-1 arrivals before SEND (t, s) = old ones <= (t, s) + new ones of EARLIER blocks <= (t, s)
        uint32_t n_news = 0;
        for (uint32_t b0 = 0; b0 < K;) {
            const double t_cut = mt[b0] + 0.9 * dl;
            uint32_t b1 = count_lt(mt, K, t_cut);   // (mt is sorted by time first)
            if (b1 <= b0) b1 = b0 + 1u;
            for (uint32_t c0 = b0; c0 < b1; c0 += (uint32_t)CHUNK) {
                const uint32_t c1 = (b1 - c0 > (uint32_t)CHUNK) ? c0 + (uint32_t)CHUNK : b1;
                for (uint32_t k = c0 + tid; k < c1; k += T) {
                    const double tk = mt[k];
                    const uint32_t sk = ms[k];
                    const uint32_t idx = draws + 2u * k + count_pairs_le(e1x, e1s, n1_old, tk, sk) + count_pairs_le(nx, nsd, n_news, tk, sk);
                    un[k - c0] = draw_at(dc, idx, flags);
                    ul[k - c0] = draw_at(dc, idx + 1u, flags);
                }
                __syncthreads();
                if (tid == 0) {   // the queue recurrence over the merged SENDs: a scan
                    LinkState L; L.q = s_q; L.tu = s_tu;
                    for (uint32_t k = c0; k < c1; k++) {
                        double ax, ay;
                        send_one(mt[k], un[k - c0], ul[k - c0], dl, lr, maxq, ebw, span, L, ax, ay);
                        e1x[n1_old + k] = ax; e1y[n1_old + k] = ay; e1s[n1_old + k] = ms[k];
                    }
                    s_q = L.q; s_tu = L.tu;
                }
                __syncthreads();
            }
            if (b1 < K) {   // the next block searches these arrivals too
                for (uint32_t k = b0 + tid; k < b1; k += T) { nx[k] = e1x[n1_old + k]; nsd[k] = e1s[n1_old + k]; }
                n_news = b1;
                __syncthreads();
                sort_pairs<T>(nx, nsd, n_news, tid);
                __syncthreads();
            }
            b0 = b1;
        }
        n1 = n1_old + K;
        __syncthreads();
        sort_events2<T>(e1x, e1s, e1y, n1, tid);
        __syncthreads();

        // ---- hop-1 arrivals before the boundary: a draw each, in their order; their hop-2 events
        const uint32_t n1p = final ? count_lt(e1x, n1, end) : count_pairs_le(e1x, e1s, n1, lim_t, lim_s);
        for (uint32_t j = tid; j < n1p; j += T) {
            const double a = e1x[j], y = e1y[j];
            const uint32_t sp = e1s[j];
            const uint32_t idx = draws + 2u * count_pairs_lt(mt, ms, K, a, sp) + j;   // SENDs (t, s) < (a, sp), arrivals before it
            double ll = dl + max0(0.0 - (a - 0.0));                                    // the return link never queues (ns:147-153)
            ll *= 1.0 + span * draw_at(dc, idx, flags);
            const double lat = fabs(y) + ll;
            e2x[n2_old + j] = a + ll;
            e2y[n2_old + j] = sign_of(y) ? -lat : lat;
            e2s[n2_old + j] = sp;
        }
        n2 = n2_old + n1p;
        __syncthreads();
        sort_events2<T>(e2x, e2s, e2y, n2, tid);
        __syncthreads();

        // ---- hop-2 arrivals before the boundary: acknowledgements (their RTTs in this order, per sender) and loss reports
        const uint32_t n2p = final ? count_lt(e2x, n2, end) : count_pairs_le(e2x, e2s, n2, lim_t, lim_s);
        if (wave0) {
            uint32_t ak[2] = {acked[0], acked[1]}, ls[2] = {lost[0], lost[1]}, fl = 0;
            for (uint32_t base = 0; base < n2p; base += kWave) {
                const uint32_t j = base + lane;
                const bool in = j < n2p;
                const double y = in ? e2y[j] : 0.0;
                const uint32_t sp = in ? e2s[j] : 0u;
#pragma unroll
                for (uint32_t s = 0; s < 2u; s++) {
                    const bool ok = in && sp == s && !sign_of(y), bad = in && sp == s && sign_of(y);
                    const uint64_t mo = __ballot(ok), mb = __ballot(bad);
                    if (ok) {
                        const uint32_t p = ak[s] + count_below(mo);
                        if (p < noise_cap) { double2 r; r.x = 0.0; r.y = y; st_rec(R[s] + p, r); }
                        else fl |= PCC_FLAG_RING_OVERFLOW;
                    }
                    ak[s] += (uint32_t)__popcll(mo);
                    ls[s] += (uint32_t)__popcll(mb);
                }
            }
            flags |= fl;
            if (lane == 0) { s_ak[0] = ak[0]; s_ak[1] = ak[1]; s_ls[0] = ls[0]; s_ls[1] = ls[1]; }
        }
        __syncthreads();
        acked[0] = s_ak[0]; acked[1] = s_ak[1]; lost[0] = s_ls[0]; lost[1] = s_ls[1];
        sent[0] += Ks[0]; sent[1] += Ks[1];

        // ---- the last sub-interval: the event that ends the interval -- the smallest of the senders' next SENDs, the first hop-1
        // and the first hop-2 arrival not yet due, by (time, sender, 'A' < 'S', hop)
        kind = 3u;
        double next0 = tsd[0][Ks[0]], next1 = tsd[1][Ks[1]];   // the senders' next SENDs after this sub-interval
        if (final) {
            if (tid == 0) {
                double bt = next0; uint32_t bsd = 0u, bk = 1u, bh = 0u, kd = 0u;   // kd: 0 = SEND of sender bsd, 1 = hop-1 arrival, 2 = hop-2 arrival
                auto better = [&](double t, uint32_t sdr, uint32_t knd, uint32_t hop) -> bool {
                    if (t != bt) return t < bt;
                    if (sdr != bsd) return sdr < bsd;
                    if (knd != bk) return knd < bk;
                    return hop < bh;
                };
                if (better(next1, 1u, 1u, 0u)) { bt = next1; bsd = 1u; bk = 1u; bh = 0u; kd = 0u; }
                if (n1p < n1 && better(e1x[n1p], e1s[n1p], 0u, 1u)) { bt = e1x[n1p]; bsd = e1s[n1p]; bk = 0u; bh = 1u; kd = 1u; }
                if (n2p < n2 && better(e2x[n2p], e2s[n2p], 0u, 2u)) { bt = e2x[n2p]; bsd = e2s[n2p]; bk = 0u; bh = 2u; kd = 2u; }
                LinkState L; L.q = s_q; L.tu = s_tu;
                double xx = 0.0, xy = 0.0, ns0 = next0, ns1 = next1;
                uint32_t fl = 0;
                if (kd == 0u) {
                    const uint32_t idx = draws + 2u * K + n1p;
                    const double u0 = draw_at(dc, idx, fl), u1 = draw_at(dc, idx + 1u, fl);
                    send_one(bt, u0, u1, dl, lr, maxq, ebw, span, L, xx, xy);   // a hop-1 event of sender bsd, not due
                    if (bsd == 0u) ns0 = bt + gap[0]; else ns1 = bt + gap[1];   // ns:161
                } else if (kd == 1u) {
                    const double a = e1x[n1p], y = e1y[n1p];
                    double ll = dl + max0(0.0 - (a - 0.0));
                    ll *= 1.0 + span * draw_at(dc, draws + 2u * K + n1p, fl);
                    const double lat = fabs(y) + ll;
                    xx = a + ll;            // a hop-2 event of sender bsd, not due
                    xy = sign_of(y) ? -lat : lat;
                } else {
                    const double y = e2y[n2p];
                    xx = sign_of(y) ? 1.0 : 0.0;   // (a loss report, or an acknowledgement: its RTT is the sender's last of the interval)
                    if (!sign_of(y)) {
                        const uint32_t p = bsd == 0u ? s_ak[0] : s_ak[1];
                        if (p < noise_cap) { double2 r; r.x = 0.0; r.y = y; st_rec(R[bsd] + p, r); }
                        else fl |= PCC_FLAG_RING_OVERFLOW;
                    }
                }
                s_xx = xx; s_xy = xy; s_kind = kd; s_ks = bsd; s_now = bt; s_nsend[0] = ns0; s_nsend[1] = ns1; s_q = L.q; s_tu = L.tu; s_flags = fl;
            }
            __syncthreads();
            kind = s_kind; ksd = s_ks;
            flags |= s_flags;
            if (kind == 2u) { if (s_xx != 0.0) lost[ksd]++; else acked[ksd]++; }
            if (kind == 0u) sent[ksd]++;
        }

        // ---- what was due and is still in flight, and the new events, go back behind what stayed, each to its sender's array
        const uint32_t f1 = n1p + (kind == 1u ? 1u : 0u), f2 = n2p + (kind == 2u ? 1u : 0u);
        __syncthreads();
        if (tid == 0) { s_cnt[0] = keep0; s_cnt[1] = keep1; }
        __syncthreads();
        for (uint32_t j = f1 + tid; j < n1; j += T) {
            const uint32_t sp = e1s[j];
            const uint32_t p = atomicAdd(&s_cnt[sp], 1u);
            if (p < noise_cap) { double2 r; r.x = e1x[j]; r.y = e1y[j]; st_rec(heap_node(H[sp], p), r); }
            else flags |= PCC_FLAG_RING_OVERFLOW;
        }
        for (uint32_t j = f2 + tid; j < n2; j += T) {
            const uint32_t sp = e2s[j];
            const uint32_t p = atomicAdd(&s_cnt[sp], 1u);
            if (p < noise_cap) { double2 r; r.x = -e2x[j]; r.y = e2y[j]; st_rec(heap_node(H[sp], p), r); }
            else flags |= PCC_FLAG_RING_OVERFLOW;
        }
        __syncthreads();
        if (tid == 0 && (kind == 0u || kind == 1u)) {
            const uint32_t p = s_cnt[ksd]++;
            if (p < noise_cap) { double2 r; r.x = kind == 0u ? s_xx : -s_xx; r.y = s_xy; st_rec(heap_node(H[ksd], p), r); }
            else flags |= PCC_FLAG_RING_OVERFLOW;
        }
        __threadfence_block();
        __syncthreads();   // (the arrays, and the senders' arrays in memory, are the next sub-interval's)
        n_heap[0] = s_cnt[0] < noise_cap ? s_cnt[0] : noise_cap; n_heap[1] = s_cnt[1] < noise_cap ? s_cnt[1] : noise_cap;
        draws += 2u * K + n1p + (kind == 0u ? 2u : kind == 1u ? 1u : 0u);
        cur[0] = next0; cur[1] = next1;
        first = false;
        if (final) break;
    }

    // every thread's flags
    for (uint32_t bit = 1u; bit <= PCC_FLAG_INTERNAL; bit <<= 1)
        if (__syncthreads_or((flags & bit) != 0u ? 1 : 0)) flags |= bit;
    if (tid == 0) {
        D.snd[sidx(D, 0, i)].heap_n = n_heap[0] | 0x80000000u;
        D.snd[sidx(D, 1, i)].heap_n = n_heap[1] | 0x80000000u;
        D.env[i].ep_draws = dc.ep0 + draws;
        D.env[i].mi_draws = dc.trace_row ? 0u : draws;
        if (give_up || (flags & PCC_FLAG_INTERNAL)) { s_nsend[0] = cur[0]; s_nsend[1] = cur[1]; }
        out0->now = s_now; out0->q = s_q; out0->tu = s_tu; out0->nsend = s_nsend[0]; out1->nsend = s_nsend[1];
        out0->sent = sent[0]; out0->acked = acked[0]; out0->lost = lost[0];
        out1->sent = sent[1]; out1->acked = acked[1]; out1->lost = lost[1];
        out0->flags = flags;
        __threadfence();
        out0->seq = D.noise_seq;
    }
}

}  // namespace

namespace pcc {

// One workgroup per env, two instances: up to 256 due events per hop and 128 SENDs in one go by one wavefront for most envs (a dozen
// workgroups per compute unit; an env that does not fit is left alone), then 1 024 / 512 with sub-intervals by four wavefronts for
// the rest (three workgroups per compute unit; it leaves at once where the first has been).
void launch_noise_sorted(const Dev &d, hipStream_t st, int warm, uint32_t warm_mi, int gate, const void *actions, int actions_f64,
                         int only_small) {
    const unsigned grid = (unsigned)d.n;
    if (d.ns == 2) {   // two senders: 256 due events per hop / 128 SENDs of both senders in one go by one wavefront, then 512 / 256 with sub-intervals by two
        // (measured at 16 384 envs x 2 senders, ms per step: this pair 11.8; 1 024 / 512 by four wavefronts as the second instance
        // 15.0 -- its 62 KB of LDS leave two workgroups per compute unit; one 256 / 128 instance with sub-intervals 15.1)
        hipLaunchKernelGGL((noise_sorted2_kernel<256, 128, 128, false, 64>), dim3(grid), dim3(64), 0, st, d, warm, warm_mi, gate, actions, actions_f64);
        if (only_small) return;
        hipLaunchKernelGGL((noise_sorted2_kernel<512, 256, 128, true, 128>), dim3(grid), dim3(128), 0, st, d, warm, warm_mi, gate, actions, actions_f64);
        return;
    }
    hipLaunchKernelGGL((noise_sorted_kernel<256, 128, 128, false, 64>), dim3(grid), dim3(64), 0, st, d, warm, warm_mi, gate, actions, actions_f64);
    if (only_small) return;
    hipLaunchKernelGGL((noise_sorted_kernel<1024, 512, 256, true, 256>), dim3(grid), dim3(256), 0, st, d, warm, warm_mi, gate, actions, actions_f64);
}

}  // namespace pcc
