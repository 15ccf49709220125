// pcc_sim.hip -- MI355X (gfx950) batched congestion-control simulator: kernels + C ABI.
//
// What this replaces (reference = PCCproject/PCC-RL; "ns" = src/gym/network_sim.py,
// "so" = src/common/sender_obs.py): the per-env heap-driven discrete-event loop
// Network.run_for_dur (ns:123-205) with its Link queue model (ns:56-96) and Sender
// accounting (ns:207-342), the monitor-interval metrics + history (so:20-206) and the env
// protocol around them (ns:344-496) -- for N independent envs advanced one monitor interval
// (MI) per step.
//
// Formulation (NOT the reference's heap; DESIGN.md section 3 has the arguments):
//   * For one sender the heap only ever holds the single pending SEND, packets on the forward
//     hop ("hop-1" events, time t1) and packets on the return hop ("hop-2" events, time
//     t2 = t1 + dl).  Link state and the loss RNG are touched only by SEND events and the rate
//     is constant inside an MI, so an MI splits into (1) the SEND stream -- a sequential
//     recurrence per env -- and (2) retiring the packets whose events fall before the MI end.
//   * In-flight packets live in two HBM rings per env per sender -- accepted packets and dropped
//     packets -- appended in send order as 16-byte records (fp64 t1, fp64 forward latency).
//     Records are never rewritten: t2 and the RTT are t1 + dl and latency + dl, recomputed.
//   * Accepted packets leave the queue >= 1/bw apart, so their send order IS event order and
//     every MI boundary on that ring is a monotone search; the RTT samples of an MI are a
//     contiguous slice of it.  Dropped packets between two accepted ones arrive at
//     mathematically equal times, so float rounding and the heap's tuple tie-break
//     (time, latency, dropped) decide their order: the dropped ring is in event order only up
//     to groups of near-equal times, and a small serial path orders the one group at each
//     boundary exactly.
//   * Every floating-point operation on the timeline is IEEE binary64 in the reference's order
//     (compile with -ffp-contract=off).  The per-MI RTT means replicate numpy's pairwise
//     summation bit for bit, because run_dur = 0.5 * mean feeds back into event boundaries.
//
// In-flight rings are tiered: small per-sender rings plus pools of 4x/16x/64x larger ones a sender
// is promoted into (by its whole wavefront, at the start of an MI that could overflow them).
//
// Kernels per step:
//   send_kernel    persistent wavefronts take work items off the class lists the previous retire filed,
//                  light items first (send_item): a light item is 64 envs of about the same predicted
//                  packet count sent lane-per-env in rounds (no loads in the loop); a heavy item is one
//                  env sent by all 64 lanes, 256 packets per pass, from closed forms (heavy_mi).
//   retire_kernel  retire_env, 16 lanes per env: searches of the rings for the hop-2 / hop-1
//                  boundaries (all four advanced together), the MI-ending event, RTT sums as numpy's
//                  pairwise tree with each 128-sample leaf summed in one round trip by an 8-lane
//                  subgroup, metrics, history, observation, reward, done; then files every env by
//                  its predicted packet count for the next send.
// No MFMA: there is no contraction anywhere on this path.
#include <hip/hip_runtime.h>
#include <cstddef>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <new>
#include <vector>

#include "pcc_sim.h"

namespace {

constexpr int kMaxFeatures = 16;
constexpr int kMaxSenders = 2;
constexpr int kWave = 64;
constexpr int kRetireBlock = 128;     // (16 lanes per env:) 8 envs per workgroup: 0.118 ms; 16 envs: 0.122 (a workgroup's slots are refilled
                                      // together); one wavefront per workgroup: 0.212 (the launch then waits for the
                                      // dispatcher, 16 384 workgroups at ~80 per us)
#ifndef PCC_RETIRE_OCC
#define PCC_RETIRE_OCC 4  // retire workgroups per SIMD the register budget is cut for: 5 spills (48 B/lane) and is slower
#endif
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

constexpr int kMaxTiers = 4;

// State of an env and of a sender, one 128-byte line each: an env's fields share a line instead of
// sharing it with the same field of 15 other envs -- both halves walk the envs in work-list order, so
// neighbours in a wavefront are not neighbours in memory.
// Fields are grouped in 16-byte pieces by who writes them, so that each half loads and writes an env
// with a few wide instructions: many narrow stores to ONE line queue up behind each other in the L2
// channel that owns it (measured: the same retire half ran 25 % slower with one 4/8-byte store per field).
struct alignas(128) EnvBlk {
    double bw, dl;    //   0  the episode's link (reset)
    double lr, maxq;  //  16
    double ebw;       //  32
    uint32_t episode;
    uint32_t pad_c;
    double q, tu;     //  48  link queue: send half, and the retire half's MI-ending event
    double now, run_dur;            //  64  retire half
    unsigned long long total_sent;  //  80  retire half
    uint32_t steps;
    uint8_t done, resetting;
    uint8_t pad0[2];
    uint32_t mi_draws;  //  96  send half: link-entry draws of the MI (a SEND the window blocks still draws)
    uint32_t ep_draws;  //      ... of the episode: the position in a replayed loss trace
    uint32_t flags;
    uint32_t pad_h;
};
struct alignas(128) SndBlk {
    double rate, rate0;          //  0  send half (rate)
    double next_send, min_lat;   // 16  both halves / retire half
    double ep_return, last_return;  // 32  retire half
    char *ring_base;             // 48  accepted ring of the sender (the dropped ring follows it)
    uint8_t ring_tier;
    uint8_t pad0[7];
    uint32_t ha, hd, ta, td;     // 64  accepted/dropped ring heads and tails
    uint32_t mi_sent;            // 80
    uint32_t ring_held[kMaxTiers];  // pool slot + 1 the sender holds in tier c (0 = none) until reset
    uint32_t cwnd;     // the reference's dormant USE_CWND option (ns:54): the sender's window in packets (ns:227: 25 at reset)
    uint32_t heap_n;   // event-loop build (event_engine): the sender's events in its heap = its packets in flight
    uint32_t pad1;
    // 112  retire half: where the last interval's four ring boundaries fell, as predictions of the next ones (speed only:
    // search_many verifies them) -- acknowledgements and loss reports per second of simulated time, and the packets that were
    // on the return hop at the interval's end (accepted / dropped ring)
    float ack_rate, loss_rate;
    uint32_t on_return_a, on_return_d;
};
static_assert(sizeof(EnvBlk) == 128 && sizeof(SndBlk) == 128, "one line per block");

// Everything a kernel needs, passed by value.
struct Dev {
    int64_t n;
    int ns, H, F, HF;
    int32_t fid[kMaxFeatures];
    // In-flight storage in tiers (see "In-flight packet storage" below): tier c rings hold
    // cap0 * 4^c accepted + twice as many dropped records.  Tier 0 is one slot per (env, sender);
    // the higher tiers are pools an env is promoted into when an MI could overflow its rings.
    int n_tiers;
    uint32_t cap0;
    char *tier_base[kMaxTiers];
    uint32_t *tier_free[kMaxTiers];  // [tier_slots[c]] free slot ids (a stack; c >= 1)
    int32_t *tier_top;               // [kMaxTiers] stack heights
    uint32_t tier_slots[kMaxTiers];  // slots of each pool
    uint32_t key0, key1, gid_base;
    double delta_scale;
    uint32_t max_steps;
    uint32_t *cls_count;  // [2][kClsStride] work lists of the send half (two buffers): envs per class, item cursor
    uint32_t *cls_list;   // [2][kClasses][N] env ids by class
    uint32_t *cursors;    // [3][kShards][kCursorStride] item cursors of the two list buffers (third block unused)
    uint32_t *any_done;   // [1] set by the retire half when an env finished its episode; gates the auto-reset launches
    uint64_t *timeline;  // profiling only (PCC_DEBUG_TIMELINE env): 8 words per send wavefront, see pcc_debug_timeline
    unsigned long long *pass_stats;  // profiling only (same switch): counters of the wave passes, see pcc_debug_pass_stats
    int pass_counters;               // ... per-pass counters on (PCC_DEBUG_TIMELINE=2: contended atomics, they slow the passes down)
    uint32_t send_wg_waves;  // tuning: wavefronts per send workgroup (each works on its own)
    uint32_t retire_sorted;  // debug: 0 = the retire launch walks the envs in index order even when there are lists
    int debug_skip;  // profile build only (PCC_DEBUG_SKIP env): bit0 skip RTT means, bit1 skip history/obs, bit2 skip the
                     // lane rounds' record stores, bit3 skip their Philox -- results wrong, timing only
    uint32_t round_packets, takeover_lanes, send_envs_per_wave, send_waves;
    double heavy_predict;  // predicted packets per MI above which an env goes to the heavy wave
    double team_predict;   // ... above which a whole workgroup sends it (team pass)
    float heavy_item_packets;  // a heavy work item is as many envs of its class as make up about this many packets (1..8 envs)
    float retire_wide_predict; // retire half: envs predicted above this many packets per interval get 16 lanes instead of 8
    double lo[5], hi[5];
    int rng_mode;
    const double *trace;
    int64_t trace_stride;
    const double *p_bw, *p_dl, *p_queue, *p_loss, *p_rate0;
    EnvBlk *env;  // [N] link + env state, one 128-byte block per env
    SndBlk *snd;  // [S][N] per sender, one 128-byte block each
    // the reference's dormant USE_CWND engine option (ns:54)
    int use_cwnd;
    // the reference's dormant USE_LATENCY_NOISE engine option (ns:51-52): packets overtake each other, so the in-flight
    // set is a real priority queue (see event_engine)
    int use_noise;
    // the event-loop build runs the interval (event_engine): with USE_LATENCY_NOISE, and with USE_CWND on two senders
    int engine;
    double noise_span;     // MAX_LATENCY_NOISE - 1.0: random.uniform(1.0, MAX) = 1.0 + span * random()
    uint32_t noise_cap;    // events / RTT samples per sender (a power of two)
    double2 *noise_heap;   // [S][N][noise_cap] (+-t, +-latency): sign of t = hop 2, sign of latency = dropped
    double2 *noise_rtt;    // [S][N][noise_cap] (-, rtt) of the packets acknowledged in the current MI, in ack order
    float *hist;    // [N][S][HF]
    double2 *ring;  // [N][S][2][cap]: accepted ring, dropped ring
};

// Profiling hooks (per-item timeline, pass counters, the "skip" switches that drop work to see what a phase costs --
// the latter make results WRONG) exist only in the -DPCC_PROFILE=1 build (libpcc_sim_prof.so, used by tools/): the
// product library carries none of it, and no environment variable can change what it computes.
#ifndef PCC_PROFILE
#define PCC_PROFILE 0
#endif
constexpr bool kProfile = PCC_PROFILE != 0;
__device__ __forceinline__ bool prof_on(const Dev &D) { return kProfile && D.timeline != nullptr; }
__device__ __forceinline__ bool prof_counters(const Dev &D) { return kProfile && D.pass_counters != 0; }
__device__ __forceinline__ bool prof_skip(const Dev &D, int bit) { return kProfile && (D.debug_skip & bit) != 0; }

// --------------------------------------------------------------------------------------
// small helpers
// --------------------------------------------------------------------------------------
__device__ __forceinline__ double max0(double x) { return x > 0.0 ? x : 0.0; }  // max(0.0, x)

__device__ __forceinline__ uint64_t mul_wide_u32(uint32_t a, uint32_t b) {
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b) : "vcc");
    return r;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one v_mad_u64_u32 per 32x32->64 product (integer multiplies are quarter rate: the
        // compiler's mul_hi + mul_lo pair costs twice as much)
        const uint64_t p0 = mul_wide_u32(c0, 0xD2511F53u), p1 = mul_wide_u32(c2, 0xCD9E8D57u);
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u32_to_unit(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }

// loss uniform of the j-th SEND on the env's link (any sender) in monitor interval mi (the draw of
// ns:73; one stream per env consumed in event order, like the reference's random.random())
__device__ __forceinline__ double philox_packet_uniform(const Dev &D, uint32_t gid, uint32_t episode, uint32_t mi,
                                                        uint32_t j) {
    uint32_t w[4];
    philox4x32_10(j >> 2, mi, episode, gid, D.key0, D.key1, w);
    const uint32_t i = j & 3u;
    return u32_to_unit(i == 0 ? w[0] : i == 1 ? w[1] : i == 2 ? w[2] : w[3]);
}

// Link.packet_enters_link + latency sampling for one SEND at time t: ns:66-84, 170-175.
// Returns the record (t + lat0, lat0) and whether the packet was dropped.  Branch-free: the three
// outcomes (random loss: queue untouched, ns:73-74; tail drop: queue drained but not grown,
// ns:75-81; accepted: ns:82) are selects over values computed in the reference's operation order.
__device__ __forceinline__ double2 link_send(double t, bool rnd /* random.random() < lr, ns:73 */, double dl,
                                             double maxq, double ebw, double &q, double &tu, bool &dropped) {
    const double qcur = max0(q - (t - tu));  // ns:66-67
    const double lat0 = dl + qcur;           // ns:170: latency before this packet queues
    const bool full = ebw + qcur > maxq;     // ns:79 (with queue_delay already = qcur)
    const double grown = qcur + ebw;         // ns:82
    q = rnd ? q : (full ? qcur : grown);
    tu = rnd ? tu : t;                       // ns:76
    dropped = rnd || full;                   // ns:175
    double2 rec;
    rec.x = t + lat0;                        // ns:174
    rec.y = lat0;                            // ns:173 (0.0 + lat0)
    return rec;
}

// ======================================================================================
// In-flight packet storage.  Per env and sender two rings of 16-byte records (t1, lat0):
//   accepted ring  packets that entered the queue, in send order.  Their arrival times grow
//                  by >= 1/bw per packet, so send order IS event order (exactly), every
//                  boundary is a monotone search, and the RTT samples of an MI are a
//                  contiguous slice.
//   dropped ring   packets lost at random or tail-dropped, in send order.  Consecutive drops
//                  with no accepted packet in between arrive at mathematically equal times
//                  (a dropped packet does not delay its successor), so rounding decides their
//                  order: send order is event order only up to "near groups" (neighbours
//                  within kNearTol relative time), which a serial path orders exactly.
// ======================================================================================
constexpr double kNearTol = 1e-12;  // >> the few-ulp spread of a tie group, << any 1/bw

__device__ __forceinline__ bool near_time(double a, double b) {
    return fabs(a - b) <= kNearTol * fmax(1.0, fabs(b));
}

// Ring accesses through explicit global-address-space pointers.  The ring addresses are loaded from
// memory (tiers), which makes them "generic" pointers to the compiler -- flat_load / flat_store,
// slower than global_load / global_store and counted against the LDS queue as well.
typedef double gvec2 __attribute__((ext_vector_type(2)));
#define PCC_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ double2 ld_rec(const double2 *p) {
    const gvec2 v = *(const PCC_GLOBAL gvec2 *)(const void *)p;
    double2 r;
    r.x = v.x; r.y = v.y;
    return r;
}
__device__ __forceinline__ void st_rec(double2 *p, const double2 &r) {
    gvec2 v;
    v.x = r.x; v.y = r.y;
    *(PCC_GLOBAL gvec2 *)(void *)p = v;
}
__device__ __forceinline__ double ld_f64(const void *p) { return *(const PCC_GLOBAL double *)p; }
__device__ __forceinline__ double ld_t1(const double2 *p) { return ld_f64(p); }  // .x of a record

// the rings of one sender: accepted ring of `cap` records at base, dropped ring of 2 * cap after it
struct RingRef {
    char *base;
    uint32_t cap;
    __device__ __forceinline__ double2 *accepted() const { return reinterpret_cast<double2 *>(base); }
    __device__ __forceinline__ double2 *dropped() const { return reinterpret_cast<double2 *>(base) + cap; }
    __device__ __forceinline__ uint32_t mask() const { return cap - 1u; }
    __device__ __forceinline__ uint32_t dmask() const { return 2u * cap - 1u; }
};

__device__ __forceinline__ uint32_t tier_cap(const Dev &D, uint32_t tier) { return D.cap0 << (2u * tier); }
__device__ __forceinline__ size_t tier_slot_bytes(const Dev &D, uint32_t tier) { return (size_t)3 * tier_cap(D, tier) * sizeof(double2); }

__device__ __forceinline__ RingRef ring_ref(const Dev &D, int64_t k /* s * n + i */) {
    RingRef r;
    r.base = D.snd[k].ring_base;
    r.cap = tier_cap(D, D.snd[k].ring_tier);
    return r;
}

// Smallest tier whose rings hold `need_a` accepted and `need_d` dropped records (n_tiers if none).
__device__ __forceinline__ uint32_t tier_for(const Dev &D, uint32_t need_a, uint32_t need_d) {
    uint32_t c = 0;
    while (c < (uint32_t)D.n_tiers && (tier_cap(D, c) < need_a || 2u * tier_cap(D, c) < need_d)) c++;
    return c;
}

// ======================================================================================
// send_kernel: apply_rate_delta (ns:235-241, 275-281) + every SEND event with time < end of the
// coming MI (ns:155-178).  One lane per env for the serial recurrence; envs with many packets in
// the MI ("heavy": deep queue, overloaded) are then processed one at a time by the whole wave,
// up to 256 packets per pass (heavy_mi below), the largest by the four wavefronts of a workgroup together.
// ======================================================================================

__device__ __forceinline__ uint32_t rl_u32(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ __forceinline__ double rl_f64(double v, uint32_t l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint64_t rl_u64(uint64_t v, uint32_t l) {
    const uint32_t lo = rl_u32((uint32_t)v, l), hi = rl_u32((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
// records [h, h + n) of one ring to another, by the whole wavefront, kCopyDepth KB in flight: a load-then-store loop
// waits one memory round trip per KB, and the promotion of a deep-queue env (10-25 k records) was the send launch's
// critical path in most steps (100-140 us of a 110-170 us launch)
constexpr int kCopyDepth = 8;
__device__ __forceinline__ void copy_records(double2 *dst, uint32_t dmask, const double2 *src, uint32_t smask, uint32_t h,
                                             uint32_t n, uint32_t lane) {
    for (uint32_t j0 = 0; j0 < n; j0 += kCopyDepth * kWave) {
        double2 r[kCopyDepth];
#pragma unroll
        for (int b = 0; b < kCopyDepth; b++) {
            const uint32_t j = j0 + (uint32_t)b * kWave + lane;
            r[b].x = 0.0; r[b].y = 0.0;
            if (j < n) r[b] = ld_rec(src + ((h + j) & smask));
        }
#pragma unroll
        for (int b = 0; b < kCopyDepth; b++) {
            const uint32_t j = j0 + (uint32_t)b * kWave + lane;
            if (j < n) st_rec(dst + ((h + j) & dmask), r[b]);
        }
    }
}

// Move the rings of sender k (lane `l` of the wavefront owns it) to a free slot of tier >= want:
// all 64 lanes copy the live records [ha, ta) / [hd, td); ring indices stay what they are, only
// the address of index j changes.  The slot the sender leaves stays reserved for it until its env
// is reset (pops happen only in send launches, pushes only in reset launches: no stack races).
// Returns false (and flags the env) when every pool from `want` up is empty.
__device__ __forceinline__ bool promote_rings(const Dev &D, uint32_t lane, uint32_t l, int64_t k, uint32_t want,
                                              uint32_t ha, uint32_t ta, uint32_t hd, uint32_t td) {
    uint32_t got = 0xFFFFFFFFu, slot = 0;
    if (lane == l) {
        for (uint32_t c = want; c < (uint32_t)D.n_tiers; c++) {
            const int32_t old = atomicSub(&D.tier_top[c], 1);
            if (old > 0) { slot = D.tier_free[c][old - 1]; got = c; break; }
            atomicAdd(&D.tier_top[c], 1);
        }
    }
    got = rl_u32(got, l);
    if (got == 0xFFFFFFFFu) return false;
    slot = rl_u32(slot, l);
    const int64_t kk = (int64_t)rl_u64((uint64_t)k, l);
    const RingRef from = ring_ref(D, kk);
    RingRef to;
    to.cap = tier_cap(D, got);
    to.base = D.tier_base[got < kMaxTiers ? got : 0] + (size_t)slot * tier_slot_bytes(D, got);
    const uint32_t h_a = rl_u32(ha, l), n_a = rl_u32(ta, l) - h_a, h_d = rl_u32(hd, l), n_d = rl_u32(td, l) - h_d;
    copy_records(to.accepted(), to.mask(), from.accepted(), from.mask(), h_a, n_a, lane);
    copy_records(to.dropped(), to.dmask(), from.dropped(), from.dmask(), h_d, n_d, lane);
    if (lane == l) {
        D.snd[k].ring_base = to.base;
        D.snd[k].ring_tier = (uint8_t)got;
        D.snd[k].ring_held[got] = slot + 1u;
    }
    return true;
}

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
template <bool TRACE, int W>
__device__ __forceinline__ void heavy_mi(const Dev &D, uint32_t lane, const uint32_t wv, TeamX *X, double dl, double lr,
                                         uint32_t thr, bool always, double maxq, double ebw, double gap, double end,
                                         uint32_t episode, uint32_t mi, uint32_t gid, const double *trace, char *base,
                                         uint32_t cap, SendState &st) {
    static_assert(W >= 1 && W <= kTeamMax, "team size");
    const uint32_t mask_b = (cap - 1u) << 4, dmask_b = (2u * cap - 1u) << 4, cap_b = cap << 4;
    const uint64_t lt = (1ull << lane) - 1ull;
    constexpr uint32_t kPass = 4u * kWave * W;
    const uint32_t glane = (W > 1 ? wv * kWave : 0u) + lane;  // lane of the team
    const bool first_lane = glane == 0u;
    const bool writer = W == 1 || wv == 0u;  // who stores what every wavefront of the team computes alike
    uint32_t serial_len = 8;
    uint32_t chain_left = 0;  // passes to send by the accept chain before the closed forms are tried again
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
                if (!((st.q > 0.0) && eM > 66u && eM < 1100u && (eq == eM || eq + 1u == eM) && (maxq - 64.0 * ebw < B) && (x0 > 0.0) &&
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
            const int kbase = 4 * (int)glane - (int)skip;  // packet index (within the pass) of position 0 of this lane
            uint32_t rnd4 = 0;
            if (TRACE) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int k = kbase + i;
                    const int64_t pos = (int64_t)((uint64_t)st.a + st.d) + k;
                    double uu = 1.0;
                    if (k >= 0 && pos < D.trace_stride) uu = trace[pos];
                    rnd4 |= (uu < lr ? 1u : 0u) << i;
                }
            } else {
                uint32_t w[4];
                philox4x32_10((st.sent >> 2) + glane, mi, episode, gid, D.key0, D.key1, w);
#pragma unroll
                for (int i = 0; i < 4; i++) rnd4 |= ((always || w[i] < thr) ? 1u : 0u) << i;
            }
            // ---- which positions hold a packet of this MI, and which of those reach the queue
            uint32_t ex4 = 0, m4 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
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
                for (int i = 0; i < 4; i++) {
                    const uint64_t bm = __ballot((acc4 >> i) & 1u);
                    j_base += (int)__popcll(bm & lt);
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
                    for (int i = 0; i < 4; i++) {
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
                    for (int i = 0; i < 4; i++) {
                        const uint64_t bm = __ballot((acc4 >> i) & 1u);
                        j_base += (int)__popcll(bm & lt);
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
                    for (int i = 0; i < 4; i++) {
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
                    for (int i = 0; i < 4; i++) {
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
                    for (int i = 0; i < 4; i++) {
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
                for (int i = 0; i < 4; i++) {
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
            uint32_t stop4 = (~ex4 | flag4) & 0xFu;
            if (first_lane) stop4 &= ~((1u << skip) - 1u);  // positions before `skip` are not part of the pass
            const uint64_t stop_lanes = __ballot(stop4 != 0u);
            uint32_t p_stop = 4u * kWave, j_stop;  // (in this wavefront's 256 positions)
            bool stopped_by_flag = false;
            if (stop_lanes) {
                const uint32_t ls = (uint32_t)__ffsll((unsigned long long)stop_lanes) - 1u;
                const uint32_t is = ((uint32_t)__ffs((int)stop4) - 1u) & 3u;
                p_stop = 4u * ls + rl_u32(is, ls);
                j_stop = rl_u32((uint32_t)j_base + (uint32_t)__popc(acc4 & ((1u << is) - 1u)), ls);
                stopped_by_flag = rl_u32((flag4 >> is) & 1u, ls) != 0u;
            } else {
                j_stop = rl_u32((uint32_t)j_base + (uint32_t)__popc(acc4), kWave - 1u);
            }
            if constexpr (W > 1) {  // exchange 2: the first wavefront with a stop ends the team's pass
                if (lane == 0) { X->pstop[wv] = p_stop; X->jstop[wv] = j_stop; X->sflag[wv] = stopped_by_flag ? 1u : 0u; }
                __syncthreads();
                uint32_t w2 = 0;
                while (w2 + 1u < (uint32_t)W && X->pstop[w2] == 4u * kWave) w2++;
                p_stop = w2 * 4u * kWave + X->pstop[w2];
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
            if (ncommit) {
                if (TRACE && (int64_t)((uint64_t)st.a + st.d + ncommit) > D.trace_stride) st.flags |= PCC_FLAG_TRACE_OVERRUN;
                // ---- records, in send order per ring (the lane replays its positions); the link state
                // behind the last packet that reached the queue
                double last_q = 0.0, last_t = 0.0;
                bool have_last = false;
                double x = x_base;
                int64_t xt = xi_base + c_before;  // regime C: the true queue in units of v
                uint32_t j = (uint32_t)j_base;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t p = 4u * glane + (uint32_t)i;
                    const bool a = (acc4 >> i) & 1u;
                    const int cpr = (int)((cp4 >> (2 * i)) & 3u) - 1;  // regime C: this packet's correction c' (0 elsewhere)
                    if (regime == 3) x = (double)xt * u;  // exact: even from B up
                    if (p >= skip && p < p_stop) {
                        const uint32_t kk = p - skip;             // packets of the pass before this one
                        const double tki = t0 + (double)kk * G;   // exact
                        const double qc = regime == 1 ? 0.0 : max0(x);  // ns:66-67
                        double2 rec;
                        rec.y = dl + qc;                          // ns:170
                        rec.x = tki + rec.y;                      // ns:174
                        const uint32_t off = a ? (((st.a + j) << 4) & mask_b) : cap_b + (((st.d + (kk - j)) << 4) & dmask_b);
                        st_rec(reinterpret_cast<double2 *>(base + off), rec);
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
                const uint32_t seg = (uint32_t)__popcll(amask & lt);
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
                const uint32_t off = my_drop ? cap_b + (((st.d + (uint32_t)__popcll(dm & lt)) << 4) & dmask_b)
                                             : (((st.a + (uint32_t)__popcll(am & lt)) << 4) & mask_b);
                st_rec(reinterpret_cast<double2 *>(base + off), rec);
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
};

template <bool TRACE>
__device__ __forceinline__ void heavy_mi2(const Dev &D, uint32_t lane, double dl, double lr, uint32_t thr, bool always,
                                          double maxq, double ebw, double gap0, double gap1, double end,
                                          uint32_t episode, uint32_t mi, uint32_t gid, const double *trace, char *base0,
                                          char *base1, uint32_t cap0, uint32_t cap1, SendState2 &st) {
    const uint32_t caps[2] = {cap0, cap1};
    const uint64_t lt = (1ull << lane) - 1ull;
    const double gap[2] = {gap0, gap1};
    uint32_t chain_left = 0;  // passes to send by the accept chain before the token pass is tried again
    uint32_t guard = 0;
    while ((st.t[0] < st.t[1] ? st.t[0] : st.t[1]) < end) {
        if (++guard > kMaxPasses) { st.flags |= PCC_FLAG_INTERNAL; break; }
        // ---- token pass, up to 256 packets: the queue stays backlogged in one binade (heavy_mi's regime B, here for the
        // merged stream; lane l owns the positions 4 l .. 4 l + 3 = one Philox block).  Every quantity is a multiple of
        // u = ulp(q): the queue in front of merged position p is x_p = Q0 + j_p R - D_p (j_p packets accepted before it,
        // D_p = t_p - tu); it is accepted iff it is not lost at random and x_p + R <= maxq, i.e. iff tokens are left:
        // b_p = N_p - j_p > 0 with N_p = floor((M - Q0 + D_p) / R).  b_{p+1} = max(b_p - m_p, 0) + (N_{p+1} - N_p) is a Lindley
        // map per position -- uneven token arrivals, because the two senders' send times interleave unevenly -- and the maps
        // compose by one prefix scan (lind_exclusive_scan).  A position whose queue runs empty or leaves the binade ends the
        // pass in front of it; the accept chain below (no such precondition, 64 packets) takes over from there.
        if (chain_left) {
            chain_left--;
        } else {
            constexpr uint32_t kPass = 4u * kWave;
            double G2[2];
            bool okb = true;
            double tend_max = 0.0;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const double t0 = st.t[s], t1s = t0 + gap[s];
                G2[s] = t1s - t0;
                const double t2s = t1s + gap[s], tend = t0 + (double)kPass * G2[s];
                // (t0 + c G is exact for c <= 256, and stays in t0's binade)
                okb = okb && (t2s - t1s == G2[s]) && (G2[s] > 0.0) && (t0 >= ((double)kPass + 4.0) * gap[s]) &&
                      (exponent_bits(t0) == exponent_bits(tend));
                tend_max = tend > tend_max ? tend : tend_max;
            }
            const uint32_t e = exponent_bits(st.q), eb = exponent_bits(ebw);
            const double T0 = st.t[0] <= st.t[1] ? st.t[0] : st.t[1];
            const double x0 = st.q - (T0 - st.tu);
            okb = okb && (st.tu + st.tu >= tend_max) && (st.q > 0.0) && e > 64u && e < 1100u && (x0 > 0.0) && eb <= e &&
                  exponent_bits(st.tu) >= e && exponent_bits(maxq) >= e;
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
            if (okb) {
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
                    const bool ex = there && tk[i] < end;
                    s4 |= (is1 ? 1u : 0u) << i;
                    ex4 |= (ex ? 1u : 0u) << i;
                    m4 |= ((ex && !((rnd4 >> i) & 1u)) ? 1u : 0u) << i;
                    if (there) { c0 += is1 ? 0u : 1u; c1 += is1 ? 1u : 0u; }   // (positions before `skip` all stand for the first packet)
                }
                // ---- accept decisions
                uint32_t acc4 = 0;
                int jb = 0;   // packets accepted before the lane's first position
                if (free_mode) {
                    acc4 = m4;
#pragma unroll
                    for (int i = 0; i < 4; i++) jb += (int)__popcll(__ballot((acc4 >> i) & 1u) & lt);
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
                uint32_t flag4 = 0;
                {
                    int j = jb;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int64_t xi = Q0i + (int64_t)j * Ri - Dp[i];
                        const double x = (double)xi * u;   // exact
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
                if (p_stop < kPass && ncommit < 32u) chain_left = 2u;   // q hovers around a binade edge or keeps running empty
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
                    double last_q = 0.0, last_t = 0.0;
                    bool have_last = false;
                    int j = jb;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool a = (acc4 >> i) & 1u;
                        if ((in4 >> i) & 1u) {
                            const bool sdr = (s4 >> i) & 1u;
                            const uint32_t kind = 2u * (sdr ? 1u : 0u) + (a ? 0u : 1u);
                            const uint32_t idx = (uint32_t)(before >> (16u * kind)) & 0xFFFFu;
                            before += 1ull << (16u * kind);
                            const double x = (double)(Q0i + (int64_t)j * Ri - Dp[i]) * u;   // exact (as above)
                            double2 rec;
                            rec.y = dl + max0(x);         // ns:66-67, 170
                            rec.x = tk[i] + rec.y;        // ns:174
                            const uint32_t cp = sdr ? cap1 : cap0;
                            const uint32_t off = a ? ((((sdr ? st.a[1] : st.a[0]) + idx) << 4) & ((cp - 1u) << 4))
                                                   : (cp << 4) + ((((sdr ? st.d[1] : st.d[0]) + idx) << 4) & ((2u * cp - 1u) << 4));
                            st_rec(reinterpret_cast<double2 *>((sdr ? base1 : base0) + off), rec);
                            if ((m4 >> i) & 1u) { have_last = true; last_t = tk[i]; last_q = a ? x + R : x; }   // ns:75-82
                        }
                        j += a ? 1 : 0;
                    }
                    const uint64_t lm = __ballot(have_last);
                    if (lm) {   // the link state behind the last committed packet that reached the queue
                        const uint32_t ll = 63u - (uint32_t)__clzll((long long)lm);
                        st.q = rl_f64(last_q, ll);
                        st.tu = rl_f64(last_t, ll);
                    }
                    const uint32_t a0n = (uint32_t)(total & 0xFFFFu), d0n = (uint32_t)((total >> 16) & 0xFFFFu);
                    const uint32_t a1n = (uint32_t)((total >> 32) & 0xFFFFu), d1n = (uint32_t)((total >> 48) & 0xFFFFu);
                    st.t[0] = st.t[0] + (double)(a0n + d0n) * G2[0];   // exact
                    st.t[1] = st.t[1] + (double)(a1n + d1n) * G2[1];
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
        bool ok = (st.tu >= maxq);
        double tend_max = 0.0;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const double t0 = st.t[s], t1s = t0 + gap[s];
            G[s] = t1s - t0;
            const double t2s = t1s + gap[s], tend = t0 + 64.0 * G[s];
            ok = ok && (t2s - t1s == G[s]) && (t0 >= 128.0 * gap[s]) && (exponent_bits(t0) == exponent_bits(tend)) &&
                 (G[s] > 0.0);
            tend_max = tend > tend_max ? tend : tend_max;
        }
        ok = ok && (st.tu + st.tu >= tend_max);
        double my_t = 0.0, my_lat = 0.0;
        bool my_drop = true;
        uint32_t my_s = 0, nv;
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
            const bool valid = tk < end;
            const uint64_t vmask = __ballot(valid);
            nv = (uint32_t)__popcll(vmask);  // merged times increase: valid lanes are a prefix
            const uint64_t rmask = rm;
            // phase 1: accepted packet to accepted packet
            double qm = st.q, tm = st.tu;
            uint64_t open = vmask & ~rmask, amask = 0;
            uint32_t na = 0;
            double seg_q = 0.0, seg_t = 0.0;
            while (open) {
                const double qc = max0(qm - (tk - tm));
                const bool full = ebw + qc > maxq;
                const uint64_t cm = open & ~__ballot(full);
                if (!cm) break;
                const uint32_t ks = (uint32_t)__ffsll((unsigned long long)cm) - 1u;
                qm = rl_f64(ebw + qc, ks);
                tm = rl_f64(tk, ks);
                if (lane == na) { seg_q = qm; seg_t = tm; }
                na++;
                amask |= 1ull << ks;
                open &= ~((2ull << ks) - 1ull);
            }
            // phase 2: every lane finishes its own packet
            const uint32_t seg = (uint32_t)__popcll(amask & lt);
            const int src = seg ? (int)seg - 1 : 0;
            double q_seg = __shfl(seg_q, src), t_seg = __shfl(seg_t, src);
            if (!seg) { q_seg = st.q; t_seg = st.tu; }
            const double qc = max0(q_seg - (tk - t_seg));
            my_lat = dl + qc;
            my_drop = !((amask >> lane) & 1ull);
            const double my_q_after = my_drop ? qc : ebw + qc;
            my_t = tk + my_lat;
            const uint64_t touch = vmask & ~rmask;
            if (touch) {
                const uint32_t kl = 63u - (uint32_t)__clzll((long long)touch);
                st.q = rl_f64(my_q_after, kl);
                st.tu = rl_f64(tk, kl);
            }
            const uint32_t n1 = (uint32_t)__popcll(__ballot(valid && my_s == 1u)), n0 = nv - n1;
            st.t[0] = st.t[0] + (double)n0 * G[0];   // exact
            st.t[1] = st.t[1] + (double)n1 * G[1];
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
                const uint32_t off = my_drop ? cap_b + (((st.d[s] + (uint32_t)__popcll(dm & lt)) << 4) & dmask_b)
                                             : (((st.a[s] + (uint32_t)__popcll(am & lt)) << 4) & mask_b);
                st_rec(reinterpret_cast<double2 *>((s ? base1 : base0) + off), rec);
            }
            st.a[s] += (uint32_t)__popcll(am);
            st.d[s] += (uint32_t)__popcll(dm);
            st.sent[s] += (uint32_t)__popcll(am) + (uint32_t)__popcll(dm);
        }
    }
}

// One work item of the send half: the SEND events of the coming monitor interval for the envs the
// lanes of this wavefront were given (lane l: env i, or none).  A light item is up to 64 envs of about
// the same predicted packet count, sent lane-per-env in rounds; a heavy item (`heavy_wave`) is ONE env
// sent by all 64 lanes (heavy_mi).  Which path sends an env is a performance choice only: every path
// is exact.  Lanes without an env stay in: the wave path needs all 64 lanes as workers.
// W > 1: a TEAM item -- one env (lane 0 of every wavefront names it) sent by the W wavefronts of the workgroup together
// (heavy_mi<.., W>); every wavefront loads the env's state and computes everything alike, wavefront 0 writes.
template <int NS, bool TRACE, int W = 1>
__device__ __forceinline__ void send_item(const Dev &D, const uint32_t lane, const int64_t i, const bool in_range,
                                          const bool heavy_wave, const uint32_t tl_slot, int warm, uint32_t warm_mi,
                                          const void *actions, int actions_f64, const uint32_t wv = 0, TeamX *X = nullptr) {
    static_assert(W == 1 || NS == 1, "team items are built for one sender");
    const bool writer = W == 1 || wv == 0u;
    const bool live = in_range && !(warm && !D.env[in_range ? i : 0].resetting);
    if (!__ballot(live)) return;
    const int64_t ii = live ? i : 0;
    const uint64_t tl0 = prof_on(D) ? wall_clock64() : 0;
    uint64_t tl1 = 0, tl_heavy = 0, tl_heavy_pk = 0, tl_closed = 0, tl_other = 0, tl_env = 0;

    const double dl = D.env[ii].dl, lr = D.env[ii].lr, maxq = D.env[ii].maxq, ebw = D.env[ii].ebw;
    double q = D.env[ii].q, tu = D.env[ii].tu;
    const double now = D.env[ii].now;
    const double end = now + D.env[ii].run_dur;  // ns:124
    const uint32_t episode = D.env[ii].episode - 1;
    const uint32_t mi = warm ? warm_mi : D.env[ii].steps + 2;
    const uint32_t gid = D.gid_base + (uint32_t)ii;
    uint32_t flags = 0;

    double gap[NS], nsend[NS];
    uint32_t ta[NS], td[NS], ha[NS], hd[NS], sent[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = (int64_t)s * D.n + ii;
        double rate = D.snd[k].rate;
        if (!warm && live) {
            const int64_t a = D.use_cwnd ? ii * 2 : ii * NS + s;  // USE_CWND: [rate action, cwnd action] per env
            double delta = actions_f64 ? ((const double *)actions)[a] : (double)((const float *)actions)[a];
            if (delta != delta) { delta = 0.0; flags |= PCC_FLAG_BAD_ACTION; }  // NaN: never silent, never in the clock
            delta *= D.delta_scale;
            rate = delta >= 0.0 ? rate * (1.0 + delta) : rate / (1.0 - delta);
            if (rate > kMaxRate) rate = kMaxRate;
            if (rate < kMinRate) rate = kMinRate;
            if constexpr (W == 1) D.snd[k].rate = rate;
        }
        if constexpr (W > 1) {  // the new rate is stored once every wavefront of the team has read the old one
            __syncthreads();
            if (writer && !warm && live) D.snd[k].rate = rate;
        }
        gap[s] = 1.0 / rate;  // ns:161
        nsend[s] = D.snd[k].next_send;
        ta[s] = D.snd[k].ta; td[s] = D.snd[k].td;
        ha[s] = D.snd[k].ha; hd[s] = D.snd[k].hd;
        sent[s] = 0;
    }
    const double *trace = TRACE ? D.trace + ii * D.trace_stride : nullptr;
    const bool run = live && now < end;

    // ---- ring tier: an upper bound of this MI's packets per sender is known up front (the send
    // times advance by gap up to rounding; one more SEND can follow in the retire half), so rings
    // that could overflow are moved to a bigger tier now, by the whole wavefront
    RingRef rings[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = (int64_t)s * D.n + ii;
        uint32_t want = 0;
        if (run) {
            const double ahead = nsend[s] < end ? (end - nsend[s]) / gap[s] + 4.0 : 1.0;
            const uint32_t n_max = ahead < 1e9 ? (uint32_t)ahead : 1000000000u;
            want = tier_for(D, ta[s] - ha[s] + n_max, td[s] - hd[s] + n_max);
        }
        uint64_t pm = __ballot(run && want > (uint32_t)D.snd[k].ring_tier && want < (uint32_t)D.n_tiers);
        while (pm && writer) {
            const uint32_t l = (uint32_t)__ffsll((unsigned long long)pm) - 1u;
            pm &= pm - 1ull;
            if (!promote_rings(D, lane, l, k, want, ha[s], ta[s], hd[s], td[s]) && lane == l) flags |= PCC_FLAG_POOL_EXHAUSTED;
        }
        if constexpr (W > 1) __syncthreads();  // the other wavefronts of a team read the address wavefront 0 just stored
        rings[s] = ring_ref(D, k);
    }
    const uint32_t mask_b = (rings[0].cap - 1u) << 4, dmask_b = (2u * rings[0].cap - 1u) << 4, cap_b = rings[0].cap << 4;

    if (NS == 1) {
        // u32_to_unit(x) < lr  <=>  x < ceil(lr * 2^32) for integer x (the scaling is exact)
        const double thr_d = ceil(lr * 4294967296.0);
        const bool always = thr_d >= 4294967296.0;
        const uint32_t thr = always ? 0xFFFFFFFFu : (thr_d > 0.0 ? (uint32_t)thr_d : 0u);
        char *base = rings[0].base;
        if (D.use_cwnd) {
            // ---- USE_CWND (ns:54, 251-255, 158-160): a SEND goes out only while fewer than cwnd
            // packets are unacknowledged.  That couples the SEND stream to the notifications, so this
            // path is lane-serial with two cursors over the lane's own rings: everything acknowledged
            // or reported lost at or before the SEND time (ACK events sort before a SEND of the same
            // time, ns:42-43) is no longer in flight.  A blocked SEND still passes through the link's
            // queue and takes its loss draw (ns:170-175 are outside the `if`): it updates (q, tu) and
            // the RNG position, but leaves no record and is not counted as sent.
            uint32_t cw = D.snd[ii].cwnd;
            if (!warm && live) {  // apply_cwnd_delta + set_cwnd: ns:243-249, 283-289
                const int64_t ai = ii * 2 + 1;
                double delta = actions_f64 ? ((const double *)actions)[ai] : (double)((const float *)actions)[ai];
                if (delta != delta) { delta = 0.0; flags |= PCC_FLAG_BAD_ACTION; }
                delta *= D.delta_scale;
                const double c = delta >= 0.0 ? (double)cw * (1.0 + delta) : (double)cw / (1.0 - delta);
                cw = c >= 5000.0 ? 5000u : (c < 4.0 ? 4u : (uint32_t)c);  // int(), then [MIN_CWND, MAX_CWND] (ns:33-34)
                D.snd[ii].cwnd = cw;
            }
            const double2 *acc = rings[0].accepted(), *drp = rings[0].dropped();
            const uint32_t amask_r = rings[0].mask(), dmask_r = rings[0].dmask();
            double t = nsend[0];
            uint32_t a = ta[0], d = td[0], pa = ha[0], pd = hd[0], draws = 0, nsent = 0;
            const uint32_t ep0 = D.env[ii].ep_draws;
            while (run && t < end) {
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
                    if ((int64_t)pos >= D.trace_stride) { flags |= PCC_FLAG_TRACE_OVERRUN; u = 1.0; }
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
                t += gap[0];  // ns:161: the next SEND is scheduled either way
            }
            if (live) {
                D.env[ii].mi_draws = draws;
                D.env[ii].ep_draws = ep0 + draws;
            }
            nsend[0] = t;
            sent[0] = nsent;
            ta[0] = a; td[0] = d;
        } else {
        const bool heavy = run && heavy_wave;
        double t = nsend[0];
        uint32_t a = ta[0], d = td[0];
        bool heavy_now = heavy;
        bool active = run && !heavy;
        uint32_t blk = 0;  // Philox block = packets sent in this MI / 4
        // Lane-serial rounds of round_packets packets per env.  After a round, if at most
        // takeover_lanes (default: one) lanes of the wave still have packets to send, they are the
        // tail everybody else would wait for: they go to the wave path, which sends ONE env's packets
        // faster than a lone lane does.  (The envs of a light item were filed together because they
        // are about the same length, so the lanes normally finish within a round of each other.)
        for (;;) {
            if (active) {
                if (!TRACE) {
                    // four packets per Philox block, no loads, no data-dependent branches.  The first
                    // `safe` packets are certainly before `end` (t advances by gap up to rounding; two
                    // packets of margin), so whole blocks run without the fp64 exit test.
                    const double ahead = (end - t) / gap[0] - 2.0;
                    uint32_t safe4 = ahead >= 4.0 ? (uint32_t)fmin(ahead, (double)D.round_packets) >> 2 : 0u;
                    uint32_t budget4 = D.round_packets / 4 - safe4;
                    for (; safe4; safe4--) {
                        uint32_t w[4];
                        if (prof_skip(D, 8)) { w[0] = blk * 2654435761u; w[1] = w[0] ^ gid; w[2] = w[1] * 40503u; w[3] = w[2] ^ mi; } else philox4x32_10(blk, mi, episode, gid, D.key0, D.key1, w);
                        blk++;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            bool dropped;
                            const double2 rec = link_send(t, always || w[k] < thr, dl, maxq, ebw, q, tu, dropped);
                            const uint32_t off = dropped ? cap_b + ((d << 4) & dmask_b) : ((a << 4) & mask_b);
                            if (!prof_skip(D, 4)) st_rec(reinterpret_cast<double2 *>(base + off), rec);
                            a += dropped ? 0u : 1u;
                            d += dropped ? 1u : 0u;
                            t += gap[0];  // ns:161
                        }
                    }
                    for (; budget4 && t < end; budget4--) {
                        uint32_t w[4];
                        if (prof_skip(D, 8)) { w[0] = blk * 2654435761u; w[1] = w[0] ^ gid; w[2] = w[1] * 40503u; w[3] = w[2] ^ mi; } else philox4x32_10(blk, mi, episode, gid, D.key0, D.key1, w);
                        blk++;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (k > 0 && !(t < end)) break;
                            bool dropped;
                            const double2 rec = link_send(t, always || w[k] < thr, dl, maxq, ebw, q, tu, dropped);
                            const uint32_t off = dropped ? cap_b + ((d << 4) & dmask_b) : ((a << 4) & mask_b);
                            if (!prof_skip(D, 4)) st_rec(reinterpret_cast<double2 *>(base + off), rec);
                            a += dropped ? 0u : 1u;
                            d += dropped ? 1u : 0u;
                            t += gap[0];  // ns:161
                        }
                    }
                } else {
                    for (uint32_t budget = D.round_packets; budget && t < end; budget--) {
                        const uint64_t pos = (uint64_t)a + d;
                        double u = 1.0;
                        if ((int64_t)pos >= D.trace_stride) flags |= PCC_FLAG_TRACE_OVERRUN;
                        else u = trace[pos];
                        bool dropped;
                        const double2 rec = link_send(t, u < lr, dl, maxq, ebw, q, tu, dropped);
                        const uint32_t off = dropped ? cap_b + ((d << 4) & dmask_b) : ((a << 4) & mask_b);
                        st_rec(reinterpret_cast<double2 *>(base + off), rec);
                        a += dropped ? 0u : 1u;
                        d += dropped ? 1u : 0u;
                        t += gap[0];
                    }
                }
                active = t < end;
            }
            const uint64_t am = __ballot(active);
            if (!am) break;
            if ((uint32_t)__popcll(am) <= D.takeover_lanes) {  // the wave path sends one env faster than a lone lane
                heavy_now = heavy_now || active;
                break;
            }
        }
        // the envs for the wave path (a heavy item's env, or the tail of a light item), one after the other
        uint64_t hm = __ballot(heavy_now);
        if (prof_on(D)) {
            tl1 = wall_clock64();
            tl_heavy = (uint64_t)__popcll(hm);
            tl_heavy_pk = (uint64_t)0 - ((a - ta[0]) + (d - td[0]));  // completed below with the final count
            if (!heavy_now) tl_heavy_pk = 0;
        }
        while (hm) {
            const uint32_t l = (uint32_t)__ffsll((unsigned long long)hm) - 1u;
            hm &= hm - 1ull;
            SendState st;
            if (prof_counters(D) && lane == 0 && writer) atomicAdd(&D.pass_stats[heavy_wave ? 11 : 12], 1ull);
            st.q = rl_f64(q, l); st.tu = rl_f64(tu, l); st.t = rl_f64(t, l);
            st.a = rl_u32(a, l); st.d = rl_u32(d, l); st.flags = 0;
            st.prof_closed = 0; st.prof_other = 0;
            st.sent = (st.a - rl_u32(ta[0], l)) + (st.d - rl_u32(td[0], l));  // packets of this MI already sent by the lane
            heavy_mi<TRACE, W>(D, lane, wv, X, rl_f64(dl, l), rl_f64(lr, l), rl_u32(thr, l), rl_u32(always ? 1u : 0u, l) != 0u,
                            rl_f64(maxq, l), rl_f64(ebw, l), rl_f64(gap[0], l), rl_f64(end, l), rl_u32(episode, l),
                            rl_u32(mi, l), rl_u32(gid, l),
                            reinterpret_cast<const double *>(rl_u64(reinterpret_cast<uint64_t>(trace), l)),
                            reinterpret_cast<char *>(rl_u64(reinterpret_cast<uint64_t>(base), l)), rl_u32(rings[0].cap, l), st);
            if (lane == l) { q = st.q; tu = st.tu; t = st.t; a = st.a; d = st.d; flags |= st.flags; }
            if (kProfile) { tl_closed += st.prof_closed; tl_other += st.prof_other; tl_env = (uint64_t)rl_u64((uint64_t)ii, l); }
        }
        nsend[0] = t;
        sent[0] = (a - ta[0]) + (d - td[0]);
        ta[0] = a; td[0] = d;
        if (prof_on(D) && heavy_now) tl_heavy_pk += sent[0];
        }  // !use_cwnd
    } else {
        // two senders merged in (time, sender id) order: lane-serial rounds, then the tail of the
        // wave goes to the two-sender wave path
        const double thr_d = ceil(lr * 4294967296.0);
        const bool always = thr_d >= 4294967296.0;
        const uint32_t thr = always ? 0xFFFFFFFFu : (thr_d > 0.0 ? (uint32_t)thr_d : 0u);
        char *bases[NS];
        uint32_t cap_bs[NS], mask_bs[NS], dmask_bs[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            bases[s] = rings[s].base;
            cap_bs[s] = rings[s].cap << 4; mask_bs[s] = (rings[s].cap - 1u) << 4; dmask_bs[s] = (2u * rings[s].cap - 1u) << 4;
        }
        bool active = run && !heavy_wave, heavy_now = run && heavy_wave;
        uint32_t blk = 0;  // Philox block = packets of this MI sent on the link / 4
        for (;;) {
            if (active) {
                if (!TRACE) {
                    // lockstep blocks of four packets of the merged stream; the sender of a packet
                    // is a select, not a branch, so lanes with different interleavings stay together
                    for (uint32_t budget4 = D.round_packets / 4;
                         budget4 && (nsend[NS - 1] < nsend[0] ? nsend[NS - 1] : nsend[0]) < end; budget4--) {
                        uint32_t w[4];
                        philox4x32_10(blk, mi, episode, gid, D.key0, D.key1, w);
                        blk++;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool s1 = nsend[NS - 1] < nsend[0];  // equal times: sender 0 first (heap order)
                            const double t = s1 ? nsend[NS - 1] : nsend[0];
                            if (k > 0 && !(t < end)) break;
                            bool dropped;
                            const double2 rec = link_send(t, always || w[k] < thr, dl, maxq, ebw, q, tu, dropped);
                            const uint32_t a_s = s1 ? ta[NS - 1] : ta[0], d_s = s1 ? td[NS - 1] : td[0];
                            const uint32_t off = dropped ? (s1 ? cap_bs[NS - 1] : cap_bs[0]) + ((d_s << 4) & (s1 ? dmask_bs[NS - 1] : dmask_bs[0]))
                                                         : ((a_s << 4) & (s1 ? mask_bs[NS - 1] : mask_bs[0]));
                            st_rec(reinterpret_cast<double2 *>((s1 ? bases[NS - 1] : bases[0]) + off), rec);
                            const uint32_t acc = dropped ? 0u : 1u, drp = dropped ? 1u : 0u;
                            if (s1) {
                                ta[NS - 1] += acc; td[NS - 1] += drp; sent[NS - 1]++;  // ns:260-262
                                nsend[NS - 1] = t + gap[NS - 1];                       // ns:161
                            } else {
                                ta[0] += acc; td[0] += drp; sent[0]++;
                                nsend[0] = t + gap[0];
                            }
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
                        if ((int64_t)pos >= D.trace_stride) flags |= PCC_FLAG_TRACE_OVERRUN;
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
            if (!am) break;
            if ((uint32_t)__popcll(am) <= D.takeover_lanes) {
                heavy_now = heavy_now || active;
                break;
            }
        }
        uint64_t hm = __ballot(heavy_now);
        while (hm) {
            const uint32_t l = (uint32_t)__ffsll((unsigned long long)hm) - 1u;
            hm &= hm - 1ull;
            SendState2 st;
            st.q = rl_f64(q, l); st.tu = rl_f64(tu, l); st.flags = 0;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                st.t[s] = rl_f64(nsend[s < NS ? s : 0], l);
                st.a[s] = rl_u32(ta[s < NS ? s : 0], l); st.d[s] = rl_u32(td[s < NS ? s : 0], l);
                st.sent[s] = rl_u32(sent[s < NS ? s : 0], l);
            }
            heavy_mi2<TRACE>(D, lane, rl_f64(dl, l), rl_f64(lr, l), rl_u32(thr, l), rl_u32(always ? 1u : 0u, l) != 0u,
                             rl_f64(maxq, l), rl_f64(ebw, l), rl_f64(gap[0], l), rl_f64(gap[NS - 1], l), rl_f64(end, l),
                             rl_u32(episode, l), rl_u32(mi, l), rl_u32(gid, l),
                             reinterpret_cast<const double *>(rl_u64(reinterpret_cast<uint64_t>(trace), l)),
                             reinterpret_cast<char *>(rl_u64(reinterpret_cast<uint64_t>(bases[0]), l)),
                             reinterpret_cast<char *>(rl_u64(reinterpret_cast<uint64_t>(bases[NS - 1]), l)),
                             rl_u32(rings[0].cap, l), rl_u32(rings[NS - 1].cap, l), st);
            if (lane == l) {
                q = st.q; tu = st.tu; flags |= st.flags;
#pragma unroll
                for (int s = 0; s < NS; s++) { nsend[s] = st.t[s]; ta[s] = st.a[s]; td[s] = st.d[s]; sent[s] = st.sent[s]; }
            }
        }
    }

    if (prof_on(D)) {
        // words: start, end of the lane rounds, end (100 MHz ticks), envs sent by the wave path,
        // packets of the wave, packets of its largest env, packets sent by the wave path, live lanes
        uint64_t sum = live ? sent[0] : 0, mx = sum, hp = tl_heavy_pk;
        for (int o = 32; o; o >>= 1) {
            sum += __shfl_xor(sum, o);
            hp += __shfl_xor(hp, o);
            const uint64_t other = __shfl_xor(mx, o);
            mx = other > mx ? other : mx;
        }
        if (lane == 0 && writer) {
            uint64_t *w = D.timeline + (int64_t)tl_slot * 8;
            w[0] = tl0; w[1] = tl1; w[2] = wall_clock64(); w[3] = tl_heavy; w[4] = sum; w[5] = mx; w[6] = hp;
            w[7] = (uint64_t)__popcll(__ballot(live)) | (tl_closed << 8) | (tl_other << 24);  // (closed-form | chain | serial passes)
            w[3] |= tl_env << 16;  // (the last env the wave path sent)
        }
    }
    if (!live || !writer) return;
    D.env[i].q = q; D.env[i].tu = tu;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        // never silent: more packets in flight than a ring holds means records were overwritten
        if (ta[s] - ha[s] > rings[s].cap || td[s] - hd[s] > 2u * rings[s].cap) flags |= PCC_FLAG_RING_OVERFLOW;
        const int64_t k = (int64_t)s * D.n + i;
        D.snd[k].next_send = nsend[s];
        D.snd[k].ta = ta[s]; D.snd[k].td = td[s];
        D.snd[k].mi_sent = sent[s];
    }
    if (flags) D.env[i].flags |= flags;
}

// ---- work lists ----------------------------------------------------------------------------
// The retire half knows every env's packet count of the NEXT monitor interval to within the effect
// of one action (run_dur x rate), so it files the env under one of kClasses geometric classes
// (class c >= 1: [8 * 1.25^(c-1), 8 * 1.25^c) packets; c = 0: fewer than 8).  The send half's work
// items come off those lists, heaviest class first: every env of a class at or above the heavy
// threshold is an item of its own (wave path), the envs of a lighter class go 64 at a time to
// lane-per-env rounds -- lanes of about the same length, so a wavefront's lanes finish together.
// Persistent wavefronts take the items off sharded cursors (send_kernel): nobody waits for a
// neighbour.  The lists are a permutation of the envs whatever the predictions say (a reset in
// between leaves stale predictions: harmless).
constexpr int kClasses = 32;
constexpr int kCntStride = 32;            // words between two counters: every class count has its own 128-byte line.  The
                                          // retire launch reads one buffer's counts while it files into the other with
                                          // atomics; a load from a line that atomics are queueing on waits behind them (a
                                          // shared line made the launch 45 % slower), and atomics on one line serialize
constexpr int kClsStride = (kClasses + 1) * kCntStride;  // words per buffer: kClasses counts + the count of the restart list
constexpr int kRestart = kClasses;        // row of the envs that finished their episode in the filing retire launch:
                                          // the next send launch runs their reset's two warm-up intervals first
constexpr int kListRows = kClasses + 1;

__device__ __forceinline__ int class_of(float pred) {
    if (!(pred >= 8.0f)) return 0;
    const int c = 1 + (int)(__log2f(pred * 0.125f) * 3.1062837f);  // 1 / log2(1.25)
    return c < kClasses - 1 ? c : kClasses - 1;
}

// read_buf < 0: no lists (after a reset, and for the warm-up intervals): the items are the envs in
// index order, 64 (send_envs_per_wave) at a time.  zero_buf: the buffer the coming retire launch files
// into; its counters are cleared here.
// Hand-out: item t belongs to shard t % kShards; wavefront w starts with item w (no atomic) and then
// claims the next item of its shard from the shard's cursor -- one returning device-scope atomic on
// one word saturates near 90 claims/us, 16 words in separate cache lines do not -- and helps the other
// shards when its own is empty (a plain look at their cursors first: no atomic on an empty shard).
constexpr uint32_t kShards = 16;
constexpr uint32_t kCursorStride = 32;  // words between shard cursors: one 128-byte line each

// (send_kernel itself follows retire_env below: a restart item runs the env's warm-up intervals through both halves)

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
        const uint32_t s_f = gbcast<G>(sidx, f);
        if (f) lo = gbcast<G>(sidx, f - 1) + 1;
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
    bool inside[K];
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
                double tn = __shfl(r[k][h].x, (int)((g.lane + 1u) & (G - 1)), G);
                if (h + 1 < R) {
                    const double tw = __shfl(r[k][h + 1 < R ? h + 1 : h].x, 0, G);  // the first record of the next row
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
            out[k].t = gbcast<G>(bx, lb & (G - 1));
            out[k].lat = gbcast<G>(by, lb & (G - 1));
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
    for (;;) {
        const bool active = !done_m && hi_m - lo_m > 12u;
        if (!gballot<G>(g, active)) break;
        const uint32_t stride = (hi_m - lo_m + 3u) / 4u;
        uint32_t sidx[PL];
        uint32_t passbits = 0;  // bit p: probe p of my search passes (probes j * PL + e of lane j)
#pragma unroll
        for (int e = 0; e < PL; e++) {
            uint32_t x = lo_m + (j * PL + (uint32_t)e + 1u) * stride;
            if (x > hi_m) x = hi_m;
            sidx[e] = x - 1u;
        }
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
        const uint32_t s_f = gbcast<G>(mine_f, LQ * q + f / PL);
        const uint32_t s_p = gbcast<G>(mine_p, LQ * q + fp / PL);
        if (active) {
            if (!mfail) {
                lo_m = hi_m;  // the last sample is record hi-1: everything passes
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
        if (need[k]) { lo[k] = gbcast<G>(lo_m, LQ * k); hi[k] = gbcast<G>(hi_m, LQ * k); }
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

// Records around a search transition b that may be out of event order: b-1 and b themselves plus
// everything chained to them by near-equal times.  [g0, g1) with h <= g0 <= b <= g1 <= tail.
__device__ __forceinline__ void near_window(const double2 *ring, uint32_t mask, uint32_t h, uint32_t tail, uint32_t b,
                                            uint32_t &g0, uint32_t &g1) {
    g0 = b;
    g1 = b;
    if (b > h) {
        g0 = b - 1;
        double t = ld_t1(ring + (g0 & mask));
        while (g0 > h) {
            const double tp = ld_t1(ring + ((g0 - 1) & mask));
            if (!near_time(tp, t)) break;
            t = tp;
            g0--;
        }
    }
    if (b < tail) {
        g1 = b + 1;
        double t = ld_t1(ring + (b & mask));
        while (g1 < tail) {
            const double tn = ld_t1(ring + (g1 & mask));
            if (!near_time(tn, t)) break;
            t = tn;
            g1++;
        }
    }
}

// Exact retire boundary of the dropped ring: on return records [h, p) are exactly those with
// t1 + dl < end (members of the boundary window that pass are moved in front, the rest keep
// their order).  Also reports the best hop-2 candidate (smallest (t2, lat2) key) among the
// window's unretired records that are already past the forward hop (t1 < end).
// (results by value: reference out-parameters of an out-of-line function live in scratch memory)
struct DropFix { uint32_t p, cand_idx; double cand_t, cand_lat; };
struct Cand { double t, lat; };

__device__ __noinline__ DropFix fix_drop_boundary(double2 *ring, uint32_t mask, uint32_t h, uint32_t tail, uint32_t b,
                                                  double dl, double end) {
    uint32_t cand_idx;
    double cand_t, cand_lat;
    uint32_t g0, g1;
    near_window(ring, mask, h, tail, b, g0, g1);
    uint32_t p = g0;
    for (uint32_t k = g0; k < g1; k++) {
        const double2 r = ld_rec(ring + (k & mask));
        if (r.x + dl < end) {
            if (k != p) rotate_to_front(ring, mask, p, k);
            p++;
        }
    }
    cand_idx = 0xFFFFFFFFu;
    cand_t = INFINITY;
    cand_lat = 0.0;
    for (uint32_t k = p; k < g1; k++) {
        const double2 r = ld_rec(ring + (k & mask));
        if (r.x < end) {
            const double t2 = r.x + dl, l2 = r.y + dl;
            if (cand_idx == 0xFFFFFFFFu || t2 < cand_t || (t2 == cand_t && l2 < cand_lat)) {
                cand_idx = k; cand_t = t2; cand_lat = l2;
            }
        }
    }
    DropFix out;
    out.p = p; out.cand_idx = cand_idx; out.cand_t = cand_t; out.cand_lat = cand_lat;
    return out;
}

// Best hop-1 candidate of the dropped ring: smallest (t1, lat) among the records still on the
// forward hop (t1 >= end); c = search transition for `t1 < end`.
__device__ __noinline__ Cand drop_hop1_candidate(const double2 *ring, uint32_t mask, uint32_t h, uint32_t tail, uint32_t c,
                                                 double end) {
    uint32_t g0, g1;
    near_window(ring, mask, h, tail, c, g0, g1);
    double cand_t = INFINITY, cand_lat = 0.0;
    for (uint32_t k = g0; k < g1; k++) {
        const double2 r = ld_rec(ring + (k & mask));
        if (!(r.x < end) && (r.x < cand_t || (r.x == cand_t && r.y < cand_lat))) { cand_t = r.x; cand_lat = r.y; }
    }
    Cand out;
    out.t = cand_t; out.lat = cand_lat;
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
struct NpSumWalk {
    // the stack lives in registers: every access is a select over the (at most 7) levels, a
    // dynamically indexed array would go to scratch memory
    static constexpr int kDepth = 7;  // 8192 -> 4096 -> ... -> 128
    uint32_t right_n[kDepth];
    double left_sum[kDepth];
    uint32_t have_left;
    int sp;
    uint32_t cur, pos, left_in_job;
    double tot;
    bool done;

    __device__ __forceinline__ void start(uint32_t beg, uint32_t n) {
        pos = beg; left_in_job = n; tot = 0.; sp = 0; have_left = 0; done = n == 0;
        cur = n < kNpBufsize ? n : kNpBufsize;
#pragma unroll
        for (int k = 0; k < kDepth; k++) { right_n[k] = 0; left_sum[k] = 0.; }
    }
    __device__ __forceinline__ bool single_leaf() const { return sp == 0 && cur == left_in_job && cur <= 128; }
    // the next leaf: [leaf_beg, leaf_beg + leaf_len)
    __device__ __forceinline__ void next(uint32_t &leaf_beg, uint32_t &leaf_len) {
        while (cur > 128) {
            uint32_t n2 = cur / 2;
            n2 -= n2 % 8;
#pragma unroll
            for (int k = 0; k < kDepth; k++)
                if (k == sp) right_n[k] = cur - n2;
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
#pragma unroll
                for (int k = 0; k < kDepth; k++)
                    if (k == top) { left_sum[k] = val; cur = right_n[k]; }
                have_left |= 1u << top;
                return;  // descend into the right part
            }
            double l = 0.;
#pragma unroll
            for (int k = 0; k < kDepth; k++)
                if (k == top) l = left_sum[k];
            val = l + val;
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
    NpSumWalk w;
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
        mean_all = gbcast<G>(res0, 0) / (double)n;
        lat_inc = halves ? gbcast<G>(res1, 8) / (double)(n - half) - gbcast<G>(res0, 8) / (double)half : 0.0;
    } else {
        mean_all = gbcast<G>(res0, 0) / (double)n;
        lat_inc = halves ? gbcast<G>(res2, 0) / (double)(n - half) - gbcast<G>(res1, 0) / (double)half : 0.0;
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
// env keeps the reference's own structure, a binary heap of its events, in global memory, and ONE
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
__device__ __forceinline__ void heap_push(double2 *H, uint32_t &n, const double2 v) {
    uint32_t pos = n++;
    while (pos > 0) {
        const uint32_t parent = (pos - 1u) >> 1;
        const double2 pv = ld_rec(H + parent);
        if (!heap_less(v, pv)) break;
        st_rec(H + pos, pv);
        pos = parent;
    }
    st_rec(H + pos, v);
}
__device__ __forceinline__ double2 heap_pop(double2 *H, uint32_t &n) {
    const double2 top = ld_rec(H);
    const double2 last = ld_rec(H + (--n));
    uint32_t pos = 0;
    for (;;) {
        uint32_t c = 2u * pos + 1u;
        if (c >= n) break;
        double2 cv = ld_rec(H + c);
        if (c + 1u < n) {
            const double2 rv = ld_rec(H + c + 1u);
            if (heap_less(rv, cv)) { cv = rv; c++; }
        }
        if (!heap_less(cv, last)) break;
        st_rec(H + pos, cv);
        pos = c;
    }
    if (n) st_rec(H + pos, last);
    return top;
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
        const int64_t k = (int64_t)s * D.n + i;
        hn[s] = D.snd[k].heap_n;
        H[s] = D.noise_heap + (size_t)k * D.noise_cap;
        R[s] = D.noise_rtt + (size_t)k * D.noise_cap;
        nsend[s] = nsend0[s];
        o.sent[s] = o.acked[s] = o.lost[s] = 0;
    }
    double now = start;
    auto draw = [&]() -> double {
        if (D.rng_mode == PCC_RNG_TRACE) {
            const uint32_t pos = ep_draws++;
            if ((int64_t)pos >= D.trace_stride) { o.flags |= PCC_FLAG_TRACE_OVERRUN; return 1.0; }
            return D.trace[i * D.trace_stride + pos];
        }
        ep_draws++;
        return philox_packet_uniform(D, D.gid_base + (uint32_t)i, episode, mi, mi_draws++);
    };
    auto noisy = [&](double ll) -> double {  // ns:150-151, 171-172
        if (D.use_noise) ll *= 1.0 + D.noise_span * draw();
        return ll;
    };
    while (now < end) {  // ns:128
        int bs = 0;
        bool from_heap = false;
        double bt = INFINITY;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (hn[s] > 0) {
                const double t = fabs(ld_rec(H[s]).x);
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
                        if (o.acked[s] < D.noise_cap) { double2 r; r.x = 0.0; r.y = lat; st_rec(R[s] + o.acked[s], r); }
                        else o.flags |= PCC_FLAG_RING_OVERFLOW;
                        o.acked[s]++;
                    }
                } else {  // hop 1: over the return link, which never queues (ns:147-153)
                    const double ll = noisy(dl + max0(0.0 - (now - 0.0)));
                    double2 nv;
                    nv.x = -(now + ll);
                    nv.y = sign_of(ev.y) ? -(lat + ll) : lat + ll;
                    if (hn[s] < D.noise_cap) heap_push(H[s], hn[s], nv);
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
                    if (hn[s] < D.noise_cap) heap_push(H[s], hn[s], nv);
                    else o.flags |= PCC_FLAG_RING_OVERFLOW;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        D.snd[(int64_t)s * D.n + i].heap_n = hn[s];
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
        const int64_t k = (int64_t)s * D.n + i;
        for (int c = 1; c < D.n_tiers; c++) {
            const uint32_t held = D.snd[k].ring_held[c];
            if (held) {
                if (push) D.tier_free[c][atomicAdd(&D.tier_top[c], 1)] = held - 1u;
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
    if (D.p_bw) {
        bw = D.p_bw[i]; lat = D.p_dl[i]; queue = D.p_queue[i]; loss = D.p_loss[i];
#pragma unroll
        for (int s = 0; s < NS; s++) rate0[s] = D.p_rate0[(int64_t)s * D.n + i];
    } else {  // ns:455-466
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
        const int64_t k = (int64_t)s * D.n + i;
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
        const int64_t k = (int64_t)s * D.n + i;
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
            const int64_t k = (int64_t)s * D.n + i;
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
            o = event_engine<NS>(D, i, start, end, rate, nsend, warm ? warm_mi : steps + 2, cw);
#pragma unroll
            for (int s = 0; s < NS; s++) D.snd[(int64_t)s * D.n + i].rate = rate[s];
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
            const int64_t k_s = (int64_t)s * D.n + i;
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
                if (lead) {
                    const DropFix fx = fix_drop_boundary(rd[s], dmasks[s], hd[s], td[s], bnd[2].b, dl, end);
                    pd = fx.p; dk = fx.cand_idx; d2_t = fx.cand_t; d2_l = fx.cand_lat;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                pd = gbcast<G>(pd, 0); dk = gbcast<G>(dk, 0); d2_t = gbcast<G>(d2_t, 0); d2_l = gbcast<G>(d2_l, 0);
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
                if (lead) {
                    const Cand c1 = drop_hop1_candidate(rd[s], dmasks[s], pd, td[s], cd, end);
                    d1_t = c1.t; d1_l = c1.lat;
                }
                d1_t = gbcast<G>(d1_t, 0); d1_l = gbcast<G>(d1_l, 0);
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
                    u = philox_packet_uniform(D, D.gid_base + (uint32_t)i, D.env[i].episode - 1,
                                              warm ? warm_mi : steps + 2, j);
                }
                if (D.use_cwnd && D.rng_mode == PCC_RNG_TRACE) {
                    const uint32_t pos = D.env[i].ep_draws;
                    if ((int64_t)pos >= D.trace_stride) { flags |= PCC_FLAG_TRACE_OVERRUN; u = 1.0; }
                    else u = D.trace[i * D.trace_stride + pos];
                }
                // USE_CWND (ns:251-255): everything due before this event is retired, so what is in
                // flight is exactly what the rings still hold
                const bool can_send = !D.use_cwnd || (ta[s] - ha[s]) + (td[s] - hd[s]) < D.snd[(int64_t)s * D.n + i].cwnd;
                if (D.use_cwnd && lead) D.env[i].ep_draws += 1u;
                const double rate = D.snd[(int64_t)s * D.n + i].rate;
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
            const int64_t k = (int64_t)s * D.n + i;
            D.snd[k].ha = ha[s]; D.snd[k].hd = hd[s]; D.snd[k].ta = ta[s]; D.snd[k].td = td[s];  // one 16-byte store
        }
    }
    if (warm) {  // reset(): the two warm-up MIs are not recorded (ns:478-479)
        if (lead) {
            D.env[i].now = now;
            D.env[i].total_sent = total_before + sent_total;
#pragma unroll
            for (int s = 0; s < NS; s++) D.snd[(int64_t)s * D.n + i].next_send = nsend[s];
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
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int64_t k = (int64_t)s * D.n + i;
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
                    hist[x] = v;
                    if (restarts) {
                        const int id = D.fid[x % D.F];
                        v = (float)(((id == PCC_M_SEND_RATIO || id == PCC_M_LATENCY_RATIO) ? 1.0 : 0.0) / c_metric_scale[id]);
                    }
                    if (obs) obs[x] = v;
                }
            }
        } else {
            for (int base = 0; base < D.HF && !prof_skip(D, 2); base += G) {
                const int x = base + (int)g.lane;
                float v = (x - keep) < G ? nf0 : nf1;
                if (x < keep) v = hist[x + D.F];
                if (x < D.HF) {
                    hist[x] = v;
                    if (restarts) {
                        const int id = D.fid[x % D.F];
                        v = (float)(((id == PCC_M_SEND_RATIO || id == PCC_M_LATENCY_RATIO) ? 1.0 : 0.0) / c_metric_scale[id]);
                    }
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
        if (done) *D.any_done = 1u;  // somebody needs the auto-reset launches of this step
        if (done_out) done_out[i] = done;
    }
    if (restart && steps + 1 >= D.max_steps) {
        // auto-reset of envs that are not in lockstep, without extra launches: the env is marked and filed in the restart
        // list, and the send launch of the next step gives it new links (ns:469-477) and runs the two warm-up
        // intervals (ns:478-479) right before its first interval -- nothing in between reads any of that (the first
        // observation of an episode is the empty history, written above)
        if (lead) {
            D.env[i].resetting = 2;
            release_ring_slots<NS>(D, i);  // here, not in the send launch: see release_ring_slots
        }
        return -2.0f;
    }
    PCC_TL_STAMP(11)  // outputs
    if (tl) atomicMax(reinterpret_cast<unsigned long long *>(&tlw[1]), (unsigned long long)tl_t);
#undef PCC_TL_STAMP
    // prediction for the next MI's send half: packets ~ MI length x current rate (the next action
    // moves the rate by at most a few percent)
    return (float)(new_run_dur * rate_sum);
}

// RESTART: the build that knows restart items (launched when the last retire launch may have filed some); the plain
// build carries none of that code (the reset and the warm-up retire inlined here cost ~5 % of the launch otherwise)
template <int NS, bool TRACE, bool RESTART>
// (The RESTART build -- the reset and the warm-up retire of restart items inlined -- is cut for 3 wavefronts per SIMD: at
// 128 registers it spilled 290-350 bytes per lane, ran the same work 21 % slower, and its code generation broke when the
// wave paths grew; at 168 it spills 130-160 bytes and --stagger runs 0.366 instead of 0.392 ms.)
__global__ __launch_bounds__(4 * kWave, RESTART ? 3 : 4) void send_kernel(Dev D, int read_buf, int zero_buf, int warm, uint32_t warm_mi,
                                                         int gate, const void *actions, int actions_f64) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;  // every wavefront works on its own
    const uint32_t n_waves = gridDim.x * (blockDim.x / kWave);                      // a multiple of kShards
    const uint32_t E = D.send_envs_per_wave;
    // auto-reset launches of a step in which no env finished have nothing to do (envs at different
    // points of their episodes: the host cannot know)
    if (gate && __hip_atomic_load(D.any_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    if (wave == 0 && lane == 0 && !warm) *D.any_done = 0u;  // consumed by the reset launches of the step before
    if (wave == 0 && zero_buf >= 0) {
        if (lane <= (uint32_t)kClasses) D.cls_count[zero_buf * kClsStride + lane * kCntStride] = 0u;
        if (lane < kShards) D.cursors[((uint32_t)zero_buf * kShards + lane) * kCursorStride] = 0u;
    }
    // ---- item table: lane l < kClasses looks after class kClasses-1-l and ranks it.  The light items go first,
    // longest first, then the envs of the wave path, longest first (~12 ns per packet; a light item lasts as long
    // as its lanes, ~0.4 us per packet of the class).  The lane rounds are bound by their scattered 16-byte stores
    // (without them the launch takes 0.085 instead of 0.151 ms) and slow every other wavefront of their CU down
    // while they run: started together at t = 0, one workgroup of them per CU (see the first item below), they are
    // out of the way soonest.
    const bool listed = read_buf >= 0;
    const int cls_mine = kClasses - 1 - (int)lane;
    const int cls_heavy = D.use_cwnd ? kClasses : (D.heavy_predict >= 1e9 ? kClasses : class_of((float)D.heavy_predict));
    // classes from cls_team up are TEAM items: envs of thousands of packets, sent by a whole workgroup (the four wavefronts
    // of the oldest quarter's workgroups, before they take other items) 1 024 packets per pass -- one wavefront at ~12 ns
    // per packet made the 10-13 k-packet envs of the later part of an episode the launch's critical path (150-180 us)
    constexpr bool kTeams = NS == 1 && !RESTART;
    constexpr uint32_t kFront = 32;  // the longest light items that trade places with items of the oldest quarter (see below)
    const uint32_t Qz0 = gridDim.x / 4u;
    const uint32_t team_wgs_max = Qz0 >= kFront ? Qz0 - kFront : Qz0;  // workgroups of the oldest quarter that can take team items
    const bool teams_on = kTeams && listed && !D.use_cwnd && D.team_predict < 1e9 && blockDim.x == kTeamMax * kWave &&
                          team_wgs_max > 0u;
    const int cls_team = teams_on ? (class_of((float)D.team_predict) > cls_heavy ? class_of((float)D.team_predict) : cls_heavy)
                                  : kClasses;
    uint32_t n_mine = 0, items_mine = 0, n_team_mine = 0, e_mine = 1;
    float est = -1.0f;  // lanes without a class sort last
    if (listed && lane < (uint32_t)kClasses) {
        n_mine = D.cls_count[read_buf * kClsStride + cls_mine * kCntStride];
        const bool hv = cls_mine >= cls_heavy;
        const float pk = 8.0f * __expf(0.22314355f * ((float)cls_mine - 0.5f));  // 8 * 1.25^(c - 1/2)
        // a heavy item is SEVERAL envs of a class when they are small (about heavy_item_packets packets together): the
        // claim, the list entry and the envs' state are three dependent round trips through a memory pipeline the lane
        // rounds keep full -- 10-17 us per 500-packet env, more than its passes take; lanes 0..e-1 load an env each and
        // the wavefront sends them one after the other
        e_mine = hv ? (uint32_t)fminf(fmaxf(D.heavy_item_packets / pk, 1.0f), 8.0f) : E;
        items_mine = (n_mine + e_mine - 1) / e_mine;
        if (cls_mine >= cls_team) { n_team_mine = n_mine; items_mine = 0; }
        est = hv ? (float)e_mine * pk * 0.012f : 1000.0f + pk * 0.4f;  // us; the light items, all of them, go first (see below)
    }
    // Lane kClasses holds a GAP of empty items between the light items and the wave-path envs: the light items fill up
    // the youngest quarter of the workgroups (see the first item below), and what the ranking puts right behind them
    // would share those CUs with four lane-round wavefronts each -- the largest wave-path envs, of all items, starved
    // of their CU's memory pipeline (1 500-2 200-packet envs that took 140-150 us instead of 35: the whole launch).
    // With the gap they start on the second-youngest quarter; the wavefronts left without a first item claim one.
    {
        uint32_t light_items = (lane < (uint32_t)kClasses && cls_mine < cls_heavy) ? items_mine : 0u;
        for (int o = 32; o; o >>= 1) light_items += (uint32_t)__shfl_xor((int)light_items, o);
        const uint32_t quarter = n_waves / 4u;
        if (lane == (uint32_t)kClasses && listed && !RESTART && light_items < quarter) {
            items_mine = quarter - light_items;
            est = 500.0f;  // behind every light item (>= 1000), in front of every wave-path env
        }
    }
    uint32_t rank = 0;  // classes that go before mine
    for (uint32_t l = 0; l <= (uint32_t)kClasses; l++) {
        const float other = __shfl(est, (int)l);
        rank += (other > est || (other == est && l < lane)) ? 1u : 0u;
    }
    // the table lives in LDS (one copy per wavefront: no barrier needed), indexed by rank
    constexpr int kRows = kClasses + 1;  // the classes and the gap
    __shared__ uint32_t s_tab[4][4][kRows];
    uint32_t (*tab)[kRows] = s_tab[threadIdx.x / kWave];
    if (lane < (uint32_t)kRows) { tab[1][rank] = items_mine; tab[2][rank] = n_mine; tab[3][rank] = (uint32_t)(lane < (uint32_t)kClasses ? cls_mine : 0) | (e_mine << 8); }
    uint32_t incl = lane < (uint32_t)kRows ? tab[1][lane] : 0u;  // inclusive prefix in rank order
    for (int o = 1; o < kRows; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += up;
    }
    if (lane < (uint32_t)kRows) tab[0][lane] = incl;
    // the restart list (envs the last retire launch reset: warm-up intervals first) goes in front, one env per item
    const uint32_t n_restart = (RESTART && listed) ? D.cls_count[read_buf * kClsStride + kRestart * kCntStride] : 0u;
    const uint32_t n_items = listed ? n_restart + rl_u32(incl, kRows - 1) : (uint32_t)((D.n + E - 1) / E);
    uint32_t *cursors = D.cursors + (uint32_t)(listed ? read_buf : 2) * kShards * kCursorStride;
    // team items: lane l looks after class kClasses-1-l, so the lane order is largest class first
    uint32_t incl_team = n_team_mine;
    for (int o = 1; o < kClasses; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl_team, o);
        if (lane >= (uint32_t)o) incl_team += up;
    }
    const uint32_t n_team = rl_u32(incl_team, kClasses - 1);
    if (kProfile && D.pass_stats && wave == 0 && lane == 0) D.pass_stats[15] = n_items + n_team;  // (team items: the last slots)
    const uint32_t s_mine = wave % kShards;
    // the first item, no claim.  The ranking is dealt over the workgroups from the YOUNGEST quarter (the last
    // to be dispatched) to the oldest: the light items land on the youngest workgroup of every CU, the largest
    // wave-path envs on the next, and so on.  A CU's memory pipeline serves its oldest wavefronts first, and four
    // lane-round wavefronts keep it busy all the time: as the oldest they starve everybody else on the CU (a
    // wave-path item next to them waited 100-160 us for its first loads), as the youngest they fill the gaps
    // (send 0.163 -> 0.148 ms; largest wave-path envs on the OLDEST workgroups instead: 0.156).  Speed only: which
    // wavefront sends an env never changes a result.
    // (The RESTART build deals oldest first: its ranking starts with the restart items, the launch's critical path.)
    uint32_t t = wave;
    uint32_t n_orph = 0, orph_base = 0, orph_qz = 1;  // orphan j is item orph_base + (j % 4) * orph_qz - j / 4 (see the team items)
    if (listed && !RESTART) {
        const uint32_t G = gridDim.x, Qz = G / 4u, inv = G - 1u - blockIdx.x;  // G is a multiple of kShards = 16
        const uint32_t blk_items = (blockDim.x / kWave) * Qz;                   // = n_waves / 4: a quarter's wavefronts
        t = (inv / Qz) * blk_items + (threadIdx.x / kWave) * Qz + inv % Qz;
        // ... except the 32 longest light items, the launch's critical path for most of an episode: they trade places
        // with items of the oldest quarter, one per CU (0.148 -> 0.143 ms; 16: 0.146, 64: 0.144, 256: 0.151)
        if (Qz >= kFront) {  // (a bijection of the first n_waves items only then)
            if (t >= 3u * blk_items && t < 3u * blk_items + kFront) t -= 3u * blk_items;
            else if (t < kFront) t += 3u * blk_items;
        }
        if (n_team) {
            // the oldest workgroups (first served by their CU's memory pipeline; not the kFront that hold the longest light
            // items) send the team items, largest class first -- workgroup b items b, b + n_tw, ... -- and then claim like
            // everybody; the first items the static hand-out gave their wavefronts ("orphans") go to the cursors instead
            if constexpr (kTeams) {
                const uint32_t n_tw = n_team < team_wgs_max ? n_team : team_wgs_max;  // workgroups that have a team item
                n_orph = n_tw * (blockDim.x / kWave);
                orph_base = 3u * blk_items + Qz - 1u;
                orph_qz = Qz;
                if (blockIdx.x < n_tw) {
                    __shared__ TeamX s_team;
                    const uint32_t wv = threadIdx.x / kWave;
                    for (uint32_t tt = blockIdx.x; tt < n_team; tt += n_tw) {
                        const uint64_t above = __ballot(lane < (uint32_t)kClasses && incl_team > tt);
                        const uint32_t L = (uint32_t)__ffsll((unsigned long long)above) - 1u;
                        const uint32_t off = tt - (rl_u32(incl_team, L) - rl_u32(n_team_mine, L));
                        const uint32_t *list = D.cls_list + ((size_t)read_buf * kListRows + (kClasses - 1u - L)) * (size_t)D.n;
                        const int64_t i = lane == 0 ? (int64_t)list[off] : 0;
                        send_item<NS, TRACE, kTeams ? kTeamMax : 1>(D, lane, i, lane == 0, true, n_items + tt, warm, warm_mi, actions,
                                                                    actions_f64, wv, &s_team);
                    }
                    t = 0xFFFFFFF0u;  // (no first item: claim)
                }
            }
        }
    }
    const uint32_t c0 = n_waves / kShards;  // items below n_waves are dealt statically, one per wavefront
    for (;;) {
        if (t >= n_items) {
            t = 0xFFFFFFFFu;
            if (lane == 0 && listed) {
                // the cursors hand out the orphans (if any) and then the items from n_waves on
                const uint64_t n_claim = (uint64_t)(n_items > n_waves ? n_items : n_waves) + n_orph;
                for (uint32_t k = 0; k < kShards && t == 0xFFFFFFFFu; k++) {
                    const uint32_t sh = (s_mine + k) % kShards;
                    uint32_t *cur = cursors + sh * kCursorStride;
                    const uint32_t seen = __hip_atomic_load(cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint64_t)(seen + c0) * kShards + sh >= n_claim) continue;  // looks empty: no atomic
                    const uint64_t cand = (uint64_t)(atomicAdd(cur, 1u) + c0) * kShards + sh;
                    if (cand >= n_claim) continue;
                    const uint32_t j = (uint32_t)cand - n_waves;
                    const uint32_t item = j < n_orph ? orph_base + (j % (blockDim.x / kWave)) * orph_qz - j / (blockDim.x / kWave)
                                                     : (uint32_t)cand - n_orph;
                    if (item < n_items) t = item;  // (an orphan past the last item: nothing)
                }
            }
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if (t == 0xFFFFFFFFu) break;
        }
        int64_t i;
        bool has, heavy = false;
        const bool restart_item = t < n_restart;
        if (restart_item) {
            heavy = true;
            has = lane == 0;
            i = has ? (int64_t)D.cls_list[((size_t)read_buf * kListRows + kRestart) * (size_t)D.n + t] : 0;
        } else if (listed) {
            const uint32_t tc = t - n_restart;
            const uint64_t above = __ballot(lane < (uint32_t)kRows && tab[0][lane < (uint32_t)kRows ? lane : 0u] > tc);
            const uint32_t L = (uint32_t)__ffsll((unsigned long long)above) - 1u;
            const int cls = (int)(tab[3][L] & 0xFFu);
            const uint32_t e_cls = tab[3][L] >> 8;  // envs per item of this class
            const uint32_t off = tc - (tab[0][L] - tab[1][L]);
            const uint32_t n_cls = tab[2][L];
            const uint32_t *list = D.cls_list + ((size_t)read_buf * kListRows + cls) * (size_t)D.n;
            heavy = cls >= cls_heavy;
            const uint32_t idx = off * e_cls + lane;
            has = lane < e_cls && idx < n_cls;
            i = has ? (int64_t)list[idx] : 0;
        } else {
            i = (int64_t)t * E + lane;
            has = lane < E && i < D.n;
        }
        // a restart item: warm-up interval 0, warm-up interval 1 (send + retire each, ns:478-479), then the env's
        // first interval like everybody's; any other item: that last pass only
        if constexpr (RESTART) {
            if (restart_item) {
                // new links and fresh state (ns:469-477) unless a flush already did all of it (pcc_get_state, a masked reset)
                if (has && D.env[i].resetting == 2) reset_env<NS>(D, i, nullptr);
                // what one lane wrote is read by the others of this wavefront: a workgroup-scope fence is enough, and an
                // agent-scope one (__threadfence) writes back and invalidates the XCD's whole L2 under everybody's feet --
                // with ~160 restart items per launch that made every other item 2.5 x slower
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            for (int pass = restart_item ? 0 : 2; pass < 3; pass++) {
                const bool wu = pass < 2;
                send_item<NS, TRACE>(D, lane, i, has, heavy, t, wu ? 1 : warm, wu ? (uint32_t)pass : warm_mi, actions, actions_f64);
                if (wu) {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // records and state just written are read by other lanes
                    const int64_t i0 = (int64_t)__builtin_amdgcn_readfirstlane((int)i) |
                                       ((int64_t)__builtin_amdgcn_readfirstlane((int)(i >> 32)) << 32);
                    if (lane < 8u) {
                        Group g;
                        g.lane = lane; g.shift = 0;
                        (void)retire_env<NS, false, 8>(D, i0, g, 1, (uint32_t)pass, pass == 1, 0, nullptr, nullptr, nullptr, nullptr,
                                                       nullptr, 0);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
            }
        } else {
            send_item<NS, TRACE>(D, lane, i, has, heavy, t, warm, warm_mi, actions, actions_f64);
        }
        t = listed ? n_items /* forces a claim */ : t + n_waves /* without lists the items are dealt statically */;
    }
}

// Order: with work lists (read_buf >= 0) the launch walks the classes the send half of this step
// read, longest first -- the acks an env retires now are about the packets predicted for it -- so
// that the envs of a wavefront carry similar work and the launch ends with its shortest envs.
// The classes from `cls_wide` up (long RTT lists: the sums are many leaves) go 16 lanes per env, 8 envs per
// workgroup; everybody else 8 lanes per env, 16 per workgroup (see "retire_kernel" above).  Without lists: index
// order, 8 lanes per env.
// Filing: every wavefront leaves its envs' classes in LDS and goes; the last one of the workgroup to
// arrive files all of them (one global atomic per class present) -- no barrier at the end, so a wavefront's
// registers are free for the next workgroup as soon as ITS envs are done.
constexpr int kRetireMaxPerBlock = kRetireBlock / 8;  // envs of a workgroup at 8 lanes per env

template <int NS, bool NOISE>
__global__ __launch_bounds__(kRetireBlock, PCC_RETIRE_OCC) void retire_kernel(Dev D, int read_buf, int fill_buf, int warm,
                                                              uint32_t warm_mi, int last_warm, int gate, int restart, float *obs_out,
                                                              float *reward_out, uint8_t *done_out, double *steps_out,
                                                              const void *actions, int actions_f64) {
    if (gate && __hip_atomic_load(D.any_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    __shared__ uint32_t s_env[kRetireMaxPerBlock], s_cls[kRetireMaxPerBlock], s_arrived;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    if (tid == 0) s_arrived = 0u;
    if (tid < (uint32_t)kRetireMaxPerBlock) s_env[tid] = 0xFFFFFFFFu;
    __syncthreads();  // the workgroup's wavefronts start together: this one is free
    int64_t i = D.n;   // (beyond the envs: nothing)
    bool wide = false;  // this workgroup: 16 lanes per env
    if (read_buf >= 0) {
        // lane l < kClasses looks after class kClasses-1-l, lane kClasses after the restart list (envs that were reset
        // by the retire launch before this one: last); inclusive prefix of the counts in that order
        const uint32_t row_mine = lane < (uint32_t)kClasses ? (uint32_t)(kClasses - 1) - lane : (uint32_t)kRestart;
        const uint32_t n_mine = lane <= (uint32_t)kClasses ? D.cls_count[read_buf * kClsStride + row_mine * kCntStride] : 0u;
        uint32_t incl = n_mine;
        for (int o = 1; o <= kClasses; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= (uint32_t)o) incl += up;
        }
        const uint32_t total = rl_u32(incl, kClasses);
        const int cls_wide = NOISE ? kClasses : (D.retire_wide_predict >= 1e9f ? kClasses : class_of(D.retire_wide_predict));
        const uint32_t n_top = cls_wide < kClasses ? rl_u32(incl, (uint32_t)(kClasses - 1 - cls_wide)) : 0u;  // envs of the wide classes
        const uint32_t wg_wide = (n_top + 7u) / 8u;  // workgroups that take them, 8 each
        wide = blockIdx.x < wg_wide;
        uint32_t p;  // this lane's position in the walk (the same for the lanes of a group)
        bool has;
        if (wide) {
            p = blockIdx.x * 8u + tid / 16u;
            has = p < n_top;
        } else {
            p = n_top + (blockIdx.x - wg_wide) * 16u + tid / 8u;
            has = p < total;
        }
        // the row whose inclusive prefix first exceeds p: binary search over lanes 0..kClasses (33 values)
        uint32_t lo_l = 0, hi_l = (uint32_t)kClasses;  // answer in [lo_l, hi_l]
        for (int it = 0; it < 6; it++) {
            const uint32_t mid = (lo_l + hi_l) >> 1;
            const uint32_t v = (uint32_t)__shfl((int)incl, (int)mid);
            if (lo_l < hi_l) { if (v > p) hi_l = mid; else lo_l = mid + 1u; }
        }
        const uint32_t L = lo_l;
        const uint32_t inc_L = (uint32_t)__shfl((int)incl, (int)L), n_L = (uint32_t)__shfl((int)n_mine, (int)L);
        if (has) {
            const uint32_t row = L < (uint32_t)kClasses ? (uint32_t)(kClasses - 1) - L : (uint32_t)kRestart;
            const uint32_t off = p - (inc_L - n_L);
            i = (int64_t)D.cls_list[((size_t)read_buf * kListRows + row) * (size_t)D.n + off];
        }
    } else {
        i = (int64_t)blockIdx.x * kRetireMaxPerBlock + tid / 8u;
    }
    float pred = -1.0f;
    Group g;
    uint32_t slot;  // the env's slot in the workgroup's filing table
    bool glead;
    if (wide) {  // (workgroup-uniform)
        g.lane = tid & 15u;
        g.shift = lane & ~15u;
        slot = tid / 16u;
        glead = g.lane == 0;
        if (i < D.n)
            pred = retire_env<NS, NOISE, 16>(D, i, g, warm, warm_mi, last_warm, restart, obs_out, reward_out, done_out, steps_out,
                                             actions, actions_f64);
    } else {
        g.lane = tid & 7u;
        g.shift = lane & ~7u;
        slot = tid / 8u;
        glead = g.lane == 0;
        if (i < D.n)
            pred = retire_env<NS, NOISE, 8>(D, i, g, warm, warm_mi, last_warm, restart, obs_out, reward_out, done_out, steps_out,
                                            actions, actions_f64);
    }
    if (fill_buf < 0) return;  // warm-up intervals do not file (kernel-uniform)
    // ---- file the workgroup's envs in the class lists of the next send (see "work lists")
    if (glead) {
        const bool restarted = pred == -2.0f;  // reset inside retire_env: its warm-up intervals come first in the next send
        // every env that was stepped is filed (-1 = warm-up / no env); a prediction that is not a number goes to class 0
        s_env[slot] = (pred != -1.0f) ? (uint32_t)i : 0xFFFFFFFFu;
        s_cls[slot] = restarted ? (uint32_t)kRestart : (uint32_t)class_of(pred);
    }
    __threadfence_block();
    uint32_t before = 0u;
    if (lane == 0) before = atomicAdd(&s_arrived, 1u);
    before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    if (before != kRetireBlock / kWave - 1) return;
    constexpr int kPerBlock = kRetireMaxPerBlock;
    const uint32_t e = lane < (uint32_t)kPerBlock ? s_env[lane & (kPerBlock - 1)] : 0xFFFFFFFFu;
    const uint32_t c = lane < (uint32_t)kPerBlock ? s_cls[lane & (kPerBlock - 1)] : 0xFFFFFFFFu;
    const bool files = e != 0xFFFFFFFFu;
    uint32_t rank = 0, same = 0, leader = lane;
#pragma unroll
    for (uint32_t l = 0; l < (uint32_t)kPerBlock; l++) {
        const uint32_t oc = (uint32_t)__shfl((int)c, (int)l), oe = (uint32_t)__shfl((int)e, (int)l);
        const bool match = oc == c && oe != 0xFFFFFFFFu;
        same += match ? 1u : 0u;
        rank += (match && l < lane) ? 1u : 0u;
        if (match && l < leader) leader = l;
    }
    uint32_t base = 0u;
    if (files && leader == lane) base = atomicAdd(&D.cls_count[fill_buf * kClsStride + c * kCntStride], same);
    base = (uint32_t)__shfl((int)base, (int)leader);
    if (files) D.cls_list[((size_t)fill_buf * kListRows + c) * (size_t)D.n + base + rank] = e;
}

// ======================================================================================
// step_small_kernel: both halves of a step in ONE launch, for batches too small for work lists (pcc_step of fewer than
// list_min_envs envs).  A workgroup owns 64 envs: its first wavefront sends them, a lane each (send_item, the tail by the
// wave path), then the four wavefronts retire them, 8 lanes per env.  No cross-workgroup dependency: an env's retire
// half needs only its own send half.  At 4 096 envs of two packets a step is launch overhead and dependent loads, and
// one launch instead of two is a third of it (config 2: 34 -> about 24 us per step).
// ======================================================================================
template <int NS, bool TRACE>
__global__ __launch_bounds__(4 * kWave, 4) void step_small_kernel(Dev D, const void *actions, int actions_f64, float *obs_out,
                                                               float *reward_out, uint8_t *done_out, double *steps_out) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int64_t base = (int64_t)blockIdx.x * kWave;
    if (blockIdx.x == 0 && threadIdx.x == 0) *D.any_done = 0u;  // (what the send launch does: consumed by the reset launches before)
    if (wv == 0) {
        const int64_t i = base + lane;
        const bool has = i < D.n;
        send_item<NS, TRACE>(D, lane, has ? i : 0, has, false, blockIdx.x, 0, 0, actions, actions_f64);
    }
    __syncthreads();  // the records and the state the first wavefront wrote are read by all four (same CU: workgroup scope)
#pragma unroll 1
    for (uint32_t r = 0; r < 2u; r++) {
        const int64_t i = base + (int64_t)((wv * 2u + r) * 8u + lane / 8u);
        Group g;
        g.lane = lane & 7u;
        g.shift = lane & ~7u;
        if (i < D.n)
            (void)retire_env<NS, false, 8>(D, i, g, 0, 0, 0, 0, obs_out, reward_out, done_out, steps_out, nullptr, 0);
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
    if (gate && __hip_atomic_load(D.any_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    const int64_t i = (int64_t)blockIdx.x * kWave + threadIdx.x;
    if (all_envs) {
        for (int c = 1; c < D.n_tiers; c++) {
            const int64_t slots = (int64_t)D.tier_slots[c];
            for (int64_t j = i; j < slots; j += (int64_t)gridDim.x * kWave) D.tier_free[c][j] = (uint32_t)(slots - 1 - j);
            if (i == 0) D.tier_top[c] = (int32_t)slots;
        }
    }
    if (i >= D.n) return;
    // use_done 1: the envs that finished their episode; 2: the envs a retire launch marked for a restart
    const bool sel = use_done == 2 ? D.env[i].resetting == 2 : (!mask || mask[i]) && (!use_done || D.env[i].done);
    D.env[i].resetting = sel ? 1 : 0;
    if (sel) {
        release_ring_slots<NS>(D, i, !all_envs);
        reset_env<NS>(D, i, obs_out);
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
    void *tier_blob[kMaxTiers];   // tier 0: one slot per (env, sender); tiers >= 1: pools
    void *tier_free_blob[kMaxTiers];
    uint32_t tier_slots[kMaxTiers];
    size_t tier_bytes[kMaxTiers];
    size_t ring_bytes;
    void *timeline_blob;
    size_t timeline_bytes;
    bool ever_reset;
    bool lockstep;      // every env was last reset by the same full reset (host knows when `done` fires)
    uint32_t host_steps;
    bool send_pending;  // pcc_step_send issued, pcc_step_retire not yet
    int read_buf;       // work-list buffer the next send launch reads (-1: none valid, items in index order)
    int fill_buf;       // ... the next retire launch files into
    int cu_count;       // compute units of the device
    void *list_blob;    // the class lists (Dev::cls_list)
    size_t list_bytes;
    void *noise_blob;   // heap + RTT samples of the latency-noise option (allocated when it is switched on)
    size_t noise_bytes;
    uint32_t ring_capacity;
    bool restarts_pending;  // a retire launch may have left envs in the restart list (their warm-up intervals are due)
    bool read_has_restarts; // the list buffer read_buf was filed by a retire launch that resets finished envs (restart list)
    uint32_t list_min_envs; // batches below this size are stepped without work lists (index order)
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
    d.env = c.take<EnvBlk>(n);
    d.snd = c.take<SndBlk>(sn);
    d.cls_count = c.take<uint32_t>(2 * kClsStride);
    d.cursors = c.take<uint32_t>(3 * 16 * 32);
    d.any_done = c.take<uint32_t>(1);
    d.tier_top = c.take<int32_t>(kMaxTiers);
    d.hist = c.take<float>(sn * d.HF);
    return (c.off + 255) & ~(size_t)255;
}

// pcc_set_ring_pools: every sender back in its own tier-0 rings, holding no pool slot (the pools are being replaced)
__global__ void forget_ring_slots_kernel(Dev D) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= D.n * D.ns) return;
    for (int c = 0; c < kMaxTiers; c++) D.snd[k].ring_held[c] = 0;
    D.snd[k].ring_tier = 0;
    const int64_t s = k / D.n, i = k % D.n;
    D.snd[k].ring_base = D.tier_base[0] + (size_t)(i * D.ns + s) * tier_slot_bytes(D, 0);
}

int check_hip(hipError_t err, const char *what) {
    if (err == hipSuccess) return PCC_OK;
    return fail(PCC_EHIP, "%s: %s", what, hipGetErrorString(err));
}

dim3 lane_grid(const Dev &d) { return dim3((unsigned)((d.n + kWave - 1) / kWave)); }

// SEND half of one monitor interval: all envs (warm = 0) or the envs being reset (warm = 1).
// Persistent single-wavefront workgroups, send_waves per compute unit, take work items off the class
// lists the last retire launch filed (sim->read_buf) -- or the envs in index order when there are none.
int launch_send(pcc_sim_t *sim, int warm, uint32_t warm_mi, int gate, const void *actions, int actions_f64, hipStream_t st) {
    const Dev &d = sim->d;
    const bool tr = d.rng_mode == PCC_RNG_TRACE;
    // small batches go without work lists (items = the envs in index order): at 4 096 envs of a few packets each the
    // launch IS its chain of dependent loads, and the lists put three more in front (counts -> list -> state)
    const bool lists = sim->d.n >= (int64_t)sim->list_min_envs;
    const int read_buf = (warm || !lists) ? -1 : sim->read_buf;
    // items <= chunks + heavy envs <= about n; more wavefronts than items would only take a failed claim each
    int64_t waves = (int64_t)sim->cu_count * d.send_waves;
    const int64_t chunks = (d.n + d.send_envs_per_wave - 1) / d.send_envs_per_wave;
    const int64_t most = read_buf < 0 ? chunks : d.n;
    if (waves > most) waves = most;
    const int64_t unit = (int64_t)kShards * d.send_wg_waves;  // whole workgroups, a multiple of kShards wavefronts
    waves = (waves + unit - 1) / unit * unit;
    const dim3 sgrid((unsigned)(waves / d.send_wg_waves)), sblock(kWave * d.send_wg_waves);
    // restart items can only be in lists that a retire launch with `restart` filed
    const bool rs = sim->read_has_restarts && read_buf >= 0;
#define PCC_LAUNCH_SEND(NS_, TR_, RS_)                                                                                      \
    hipLaunchKernelGGL((send_kernel<NS_, TR_, RS_>), sgrid, sblock, 0, st, d, read_buf, lists ? sim->fill_buf : -1, warm, warm_mi, gate, \
                       actions, actions_f64)
    if (d.ns == 1) {
        if (tr) { if (rs) PCC_LAUNCH_SEND(1, true, true); else PCC_LAUNCH_SEND(1, true, false); }
        else { if (rs) PCC_LAUNCH_SEND(1, false, true); else PCC_LAUNCH_SEND(1, false, false); }
    } else {
        if (tr) { if (rs) PCC_LAUNCH_SEND(2, true, true); else PCC_LAUNCH_SEND(2, true, false); }
        else { if (rs) PCC_LAUNCH_SEND(2, false, true); else PCC_LAUNCH_SEND(2, false, false); }
    }
#undef PCC_LAUNCH_SEND
    if (rs && !warm) sim->restarts_pending = false;  // this launch runs what the restart list's envs were owed
    return check_hip(hipGetLastError(), "send kernel launch");
}

// RETIRE half; a launch that is not a warm-up interval also files every env in the work lists of the
// next send (buffer sim->fill_buf, cleared by the send launch before it)
// restart: envs that finish their episode in this launch are reset inside it and filed in the restart list
int launch_retire(pcc_sim_t *sim, int warm, uint32_t warm_mi, int last_warm, int gate, int restart, float *obs_out,
                  float *reward_out, uint8_t *done_out, double *steps_out, hipStream_t st) {
    const Dev &d = sim->d;
    // workgroups: 8 envs each at 16 lanes per env, 16 at 8 lanes -- which envs go which way is decided on the device
    // (class counts), so the grid covers the worst case plus the one workgroup the split can leave partly filled
    const bool lists = d.n >= (int64_t)sim->list_min_envs;
    const int read = (warm || !d.retire_sorted || !lists) ? -1 : sim->read_buf;  // the lists this step's send launch read
    const int64_t per_block = read >= 0 ? 8 : kRetireMaxPerBlock;
    const dim3 grid((unsigned)((d.n + per_block - 1) / per_block + (read >= 0 ? 1 : 0)));
    const int fill = (warm || !lists) ? -1 : sim->fill_buf;
    if (d.ns == 1)
        hipLaunchKernelGGL((retire_kernel<1, false>), grid, dim3(kRetireBlock), 0, st, d, read, fill, warm, warm_mi, last_warm,
                           gate, restart, obs_out, reward_out, done_out, steps_out, nullptr, 0);
    else
        hipLaunchKernelGGL((retire_kernel<2, false>), grid, dim3(kRetireBlock), 0, st, d, read, fill, warm, warm_mi, last_warm,
                           gate, restart, obs_out, reward_out, done_out, steps_out, nullptr, 0);
    const int rc = check_hip(hipGetLastError(), "retire kernel launch");
    if (rc == PCC_OK && !warm && lists) {
        sim->read_buf = sim->fill_buf;
        sim->fill_buf ^= 1;
        sim->read_has_restarts = restart != 0;  // the buffer just filed may hold a restart list
        if (restart) sim->restarts_pending = true;
    }
    return rc;
}

int launch_mi(pcc_sim_t *sim, int warm, uint32_t warm_mi, int last_warm, int gate, int restart, const void *actions,
              int actions_f64, float *obs_out, float *reward_out, uint8_t *done_out, double *steps_out, hipStream_t st) {
    if (sim->d.engine) {
        // the event-loop build (latency noise; the congestion window with two senders): the whole interval is one launch of
        // the retire kernel's NOISE build (no send half, no work lists)
        const Dev &d = sim->d;
        const int64_t per_block = kRetireMaxPerBlock;
        const dim3 grid((unsigned)((d.n + per_block - 1) / per_block));
        if (d.ns == 1)
            hipLaunchKernelGGL((retire_kernel<1, true>), grid, dim3(kRetireBlock), 0, st, d, -1, -1, warm, warm_mi, last_warm, gate,
                               0, obs_out, reward_out, done_out, steps_out, actions, actions_f64);
        else
            hipLaunchKernelGGL((retire_kernel<2, true>), grid, dim3(kRetireBlock), 0, st, d, -1, -1, warm, warm_mi, last_warm, gate,
                               0, obs_out, reward_out, done_out, steps_out, actions, actions_f64);
        return check_hip(hipGetLastError(), "event-loop kernel launch");
    }
    const int rc = launch_send(sim, warm, warm_mi, gate, actions, actions_f64, st);
    if (rc != PCC_OK) return rc;
    return launch_retire(sim, warm, warm_mi, last_warm, gate, restart, obs_out, reward_out, done_out, steps_out, st);
}

// reset(): parameters + state, then the two unrecorded warm-up MIs (ns:469-484).  gate: the launches
// do nothing unless the retire half flagged a finished env in this step (auto-reset of envs that are
// not in lockstep)
int launch_reset(pcc_sim_t *sim, const uint8_t *mask, int use_done, int gate, float *obs_out, hipStream_t st) {
    const Dev &d = sim->d;
    // every env of the batch is reset: a full reset, or the episode boundary of a batch in lockstep (every env is done)
    const int all_envs = (!mask && !gate && (use_done == 0 || (use_done == 1 && sim->lockstep))) ? 1 : 0;
    if (d.ns == 1) hipLaunchKernelGGL(reset_init_kernel<1>, lane_grid(d), dim3(kWave), 0, st, d, mask, use_done, gate, all_envs, obs_out);
    else hipLaunchKernelGGL(reset_init_kernel<2>, lane_grid(d), dim3(kWave), 0, st, d, mask, use_done, gate, all_envs, obs_out);
    int rc = check_hip(hipGetLastError(), "reset kernel launch");
    for (uint32_t w = 0; w < 2 && rc == PCC_OK; w++)
        rc = launch_mi(sim, 1, w, w == 1, gate, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, st);
    return rc;
}

// Auto-reset of envs that are not in lockstep happens inside the step's own launches (the retire half resets a
// finished env and files it in the restart list, the next send half runs its warm-up intervals) whenever the
// send half can take such items: not with the congestion-window option (no wave path) or the latency-noise
// option (no send half) -- those keep the gated reset launches after the step.
bool restarts_in_step(const pcc_sim_t *sim, int auto_reset) {
    return auto_reset && !sim->lockstep && !sim->d.use_cwnd && !sim->d.engine && sim->d.n >= (int64_t)sim->list_min_envs;
}

// What is still owed to the envs of the restart list -- new links, fresh state, the two warm-up intervals -- is
// done now (before the state is read or the lists are dropped), by the ordinary reset launches over the marked envs.
int flush_restarts(pcc_sim_t *sim, hipStream_t st) {
    if (!sim->restarts_pending) return PCC_OK;
    sim->restarts_pending = false;
    return launch_reset(sim, nullptr, 2, 0, nullptr, st);
}

// (re)allocate tier c: tier 0 is one slot per sender, tiers >= 1 are pools of senders / divisor slots with a free stack
int alloc_tier(pcc_sim_t *sim, int c, unsigned divisor) {
    Dev &d = sim->d;
    const size_t senders = (size_t)d.n * d.ns;
    size_t slots = senders / (divisor ? divisor : 1);
    if (slots < 256) slots = senders < 256 ? senders : 256;
    if (c == 0) slots = senders;
    if (sim->tier_blob[c]) { (void)hipFree(sim->tier_blob[c]); sim->tier_blob[c] = nullptr; sim->ring_bytes -= sim->tier_bytes[c]; }
    if (sim->tier_free_blob[c]) { (void)hipFree(sim->tier_free_blob[c]); sim->tier_free_blob[c] = nullptr; }
    sim->tier_bytes[c] = 0;
    sim->tier_slots[c] = (uint32_t)slots;
    d.tier_slots[c] = (uint32_t)slots;
    const size_t bytes = slots * 3 * ((size_t)d.cap0 << (2 * c)) * sizeof(double2);
    if (hipMalloc(&sim->tier_blob[c], bytes) != hipSuccess) {
        sim->tier_blob[c] = nullptr;
        return fail(PCC_ENOMEM, "hipMalloc(%zu) for the tier-%d in-flight rings failed (%zu slots of 3*%u records)", bytes, c,
                    slots, d.cap0 << (2 * c));
    }
    // touch the rings once now: freshly allocated device memory is markedly slower on first use
    // (measured 1.8x on the first episode of a new handle), which would land in the caller's steps
    if (hipMemset(sim->tier_blob[c], 0, bytes) != hipSuccess) return fail(PCC_EHIP, "hipMemset of the tier-%d rings failed", c);
    size_t total = bytes;
    d.tier_base[c] = static_cast<char *>(sim->tier_blob[c]);
    if (c >= 1) {
        std::vector<uint32_t> ids(slots);
        for (size_t j = 0; j < slots; j++) ids[j] = (uint32_t)(slots - 1 - j);  // slot 0 is popped first
        if (hipMalloc(&sim->tier_free_blob[c], slots * sizeof(uint32_t)) != hipSuccess ||
            hipMemcpy(sim->tier_free_blob[c], ids.data(), slots * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
            return fail(PCC_ENOMEM, "allocating the tier-%d free list failed", c);
        total += slots * sizeof(uint32_t);
        d.tier_free[c] = static_cast<uint32_t *>(sim->tier_free_blob[c]);
    }
    sim->tier_bytes[c] = total;
    sim->ring_bytes += total;
    return PCC_OK;
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
    if (ring_capacity < 16 || ring_capacity > (1u << 27) || (ring_capacity & (ring_capacity - 1)))
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
    // tiers: cap0 * 4^c records, the top tier = ring_capacity; as many tiers (<= 4) as keep cap0 >= 256
    sim->ring_capacity = ring_capacity;
    d.n_tiers = 1; d.cap0 = ring_capacity;
    while (d.n_tiers < kMaxTiers && d.cap0 >= 4u * 256u) { d.cap0 >>= 2; d.n_tiers++; }
    d.key0 = (uint32_t)seed; d.key1 = (uint32_t)(seed >> 32); d.gid_base = env_gid_base;
    d.delta_scale = 0.025;  // src/common/config.py:17
    d.max_steps = 400;      // ns:41
    d.round_packets = 256;
    d.takeover_lanes = 1;  // the last lane of a light item goes to the wave path (more lanes handed over measured slower)
    d.send_waves = 16;  // persistent send wavefronts per compute unit (4 per SIMD at <= 128 VGPRs)
    d.send_envs_per_wave = 64;
    // (two senders: the lane rounds cost about the same per packet as with one, the wave passes more -- 64 positions per pass)
    d.heavy_predict = n_senders == 2 ? 1024.0 : 512.0;
    d.team_predict = 4096.0;
    d.heavy_item_packets = 2048.0f;
    d.retire_wide_predict = 1024.0f;
    d.send_wg_waves = getenv("PCC_SEND_WG_WAVES") ? (uint32_t)atoi(getenv("PCC_SEND_WG_WAVES")) : 4u;
    if (d.send_wg_waves < 1u || d.send_wg_waves > 4u) d.send_wg_waves = 4u;
    d.retire_sorted = getenv("PCC_RETIRE_SORTED") ? (uint32_t)atoi(getenv("PCC_RETIRE_SORTED")) : 1u;
    d.debug_skip = (kProfile && getenv("PCC_DEBUG_SKIP")) ? atoi(getenv("PCC_DEBUG_SKIP")) : 0;  // profile build only
    const double lo[5] = {100, 0.05, 0, 0.0, 0.3}, hi[5] = {500, 0.5, 8, 0.05, 1.5};  // ns:355-358,466
    memcpy(d.lo, lo, sizeof lo); memcpy(d.hi, hi, sizeof hi);
    d.rng_mode = PCC_RNG_PHILOX;
    sim->device = device;
    sim->state_bytes = carve_state(d, nullptr);
    if (hipMalloc(&sim->state_blob, sim->state_bytes) != hipSuccess) {
        const size_t want = sim->state_bytes;
        delete sim;
        return fail(PCC_ENOMEM, "hipMalloc(%zu) for env state failed", want);
    }
    carve_state(d, static_cast<char *>(sim->state_blob));
    // pool sizes: by default 1/2, 1/8, 1/32 of the senders can sit in tiers 1, 2, 3 at the same
    // time (measured need at the ICML'19 ranges with U(-1, 1) actions: ~25 %, ~2 %, ~0.02 %); pcc_set_ring_pools changes
    // the divisors (1 = every sender could, the worst case -- what a rate-maximising policy may need)
    const unsigned div[kMaxTiers] = {1, 2, 8, 32};
    sim->ring_bytes = 0;
    for (int c = 0; c < d.n_tiers; c++) {
        const int rc = alloc_tier(sim, c, div[c]);
        if (rc != PCC_OK) { pcc_destroy(sim); return rc; }
    }
    int32_t tops[kMaxTiers] = {0, 0, 0, 0};
    for (int c = 1; c < d.n_tiers; c++) tops[c] = (int32_t)sim->tier_slots[c];
    if (hipMemset(sim->state_blob, 0, sim->state_bytes) != hipSuccess ||
        hipMemcpy(d.tier_top, tops, sizeof tops, hipMemcpyHostToDevice) != hipSuccess) {
        pcc_destroy(sim);
        return fail(PCC_EHIP, "initialising the env state failed");
    }
    if (kProfile && getenv("PCC_DEBUG_TIMELINE") && atoi(getenv("PCC_DEBUG_TIMELINE"))) {  // profile build only
        sim->timeline_bytes = (size_t)n_envs * 4 * 8 * sizeof(uint64_t);
        if (hipMalloc(&sim->timeline_blob, sim->timeline_bytes) != hipSuccess ||
            hipMemset(sim->timeline_blob, 0, sim->timeline_bytes) != hipSuccess) {
            pcc_destroy(sim);
            return fail(PCC_ENOMEM, "hipMalloc for the debug timeline failed");
        }
        d.timeline = static_cast<uint64_t *>(sim->timeline_blob);
        d.pass_stats = reinterpret_cast<unsigned long long *>(d.timeline + (size_t)n_envs * 24);  // past everything the timeline uses
        d.pass_counters = atoi(getenv("PCC_DEBUG_TIMELINE")) >= 2;
    }
    // the send half's work lists: two buffers of kClasses lists, each able to hold every env
    sim->list_bytes = (size_t)2 * kListRows * (size_t)n_envs * sizeof(uint32_t);
    if (hipMalloc(&sim->list_blob, sim->list_bytes) != hipSuccess) {
        pcc_destroy(sim);
        return fail(PCC_ENOMEM, "hipMalloc(%zu) for the send work lists failed", sim->list_bytes);
    }
    d.cls_list = static_cast<uint32_t *>(sim->list_blob);
    sim->read_buf = -1;
    sim->fill_buf = 0;
    sim->list_min_envs = 8192;
    sim->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = sim;
    return PCC_OK;
}

int64_t pcc_debug_timeline(pcc_sim_t *sim, uint64_t *out, int64_t n_words) {
    if (!sim || !sim->timeline_blob) return 0;
    DeviceGuard guard(sim->device);
    unsigned long long items = 0;
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(&items, sim->d.pass_stats + 15, sizeof items, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(PCC_EHIP, "reading the debug timeline failed");
    const int64_t rblocks = (sim->d.n + 7) / 8 + 1;  // (the largest retire grid)
    const int64_t total = (int64_t)items * 8 + rblocks * 16;
    if (!out || n_words <= 0) return total;
    if (n_words < total) return fail(PCC_EINVAL, "pcc_debug_timeline needs room for %lld words", (long long)total);
    const char *src = static_cast<const char *>(sim->timeline_blob);
    if (hipMemcpy(out, src, (size_t)items * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(out + items * 8, src + (size_t)sim->d.n * 2 * 8 * sizeof(uint64_t), (size_t)rblocks * 16 * sizeof(uint64_t),
                  hipMemcpyDeviceToHost) != hipSuccess)
        return fail(PCC_EHIP, "reading the debug timeline failed");
    return total;
}

int pcc_debug_pass_stats(pcc_sim_t *sim, uint64_t *out16, int reset) {
    if (!sim || !out16) return fail(PCC_EINVAL, "NULL argument");
    if (!sim->d.pass_stats) return fail(PCC_ESTATE, "pass statistics are off (create the handle with PCC_DEBUG_TIMELINE=1)");
    DeviceGuard guard(sim->device);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(out16, sim->d.pass_stats, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess ||
        (reset && hipMemset(sim->d.pass_stats, 0, 16 * sizeof(uint64_t)) != hipSuccess))
        return fail(PCC_EHIP, "reading the pass statistics failed");
    return PCC_OK;
}

void pcc_destroy(pcc_sim_t *sim) {
    if (!sim) return;
    DeviceGuard guard(sim->device);
    if (sim->timeline_blob) (void)hipFree(sim->timeline_blob);
    if (sim->list_blob) (void)hipFree(sim->list_blob);
    if (sim->noise_blob) (void)hipFree(sim->noise_blob);

    if (sim->state_blob) (void)hipFree(sim->state_blob);
    for (int c = 0; c < kMaxTiers; c++) {
        if (sim->tier_blob[c]) (void)hipFree(sim->tier_blob[c]);
        if (sim->tier_free_blob[c]) (void)hipFree(sim->tier_free_blob[c]);
    }
    delete sim;
}

int64_t pcc_device_bytes(const pcc_sim_t *sim) { return sim ? (int64_t)(sim->state_bytes + sim->ring_bytes + sim->list_bytes + sim->noise_bytes) : 0; }

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
    for (int k = 0; k < 5; k++)
        if (!(lo[k] <= hi[k])) return fail(PCC_EINVAL, "range %d is empty", k);
    // what the exact formulation rests on: a physical link -- positive one-way
    // delay, 1/bw >= 1e-8 s -- a queue of at least one packet, a probability, a positive starting rate
    if (!(lo[0] > 0.0) || !(hi[0] <= 1e8)) return fail(PCC_EINVAL, "bandwidth range must lie in (0, 1e8] packets/s");
    if (!(lo[1] > 0.0) || !(hi[1] <= 1e6)) return fail(PCC_EINVAL, "latency range must lie in (0, 1e6] s");
    if (!(lo[2] >= 0.0) || !(hi[2] <= 20.0)) return fail(PCC_EINVAL, "queue exponent range must lie in [0, 20] (queue = 1 + floor(e^x))");
    if (!(lo[3] >= 0.0) || !(hi[3] <= 1.0)) return fail(PCC_EINVAL, "loss range must lie in [0, 1]");
    if (!(lo[4] > 0.0)) return fail(PCC_EINVAL, "the starting-rate factor must be positive");
    for (int k = 0; k < 5; k++) { sim->d.lo[k] = lo[k]; sim->d.hi[k] = hi[k]; }
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

int pcc_set_tuning(pcc_sim_t *sim, int key, double value) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    switch (key) {
        case PCC_TUNE_ROUND_PACKETS:
            if (value < 4 || value > 1048576) return fail(PCC_EINVAL, "round_packets out of range");
            sim->d.round_packets = ((uint32_t)value + 3u) & ~3u;
            return PCC_OK;
        case PCC_TUNE_SEND_ENVS_PER_WAVE:
            if (value < 1 || value > 64) return fail(PCC_EINVAL, "send_envs_per_wave out of range");
            sim->d.send_envs_per_wave = (uint32_t)value;
            return PCC_OK;
        case PCC_TUNE_HEAVY_PREDICT: sim->d.heavy_predict = value; return PCC_OK;
        case PCC_TUNE_TEAM_PREDICT: sim->d.team_predict = value; return PCC_OK;
        case PCC_TUNE_LIST_MIN_ENVS:
            if (!(value >= 0.0 && value <= 4e9)) return fail(PCC_EINVAL, "list_min_envs out of range");
            if (sim->send_pending) return fail(PCC_ESTATE, "pcc_set_tuning(LIST_MIN_ENVS) between pcc_step_send and pcc_step_retire");
            sim->list_min_envs = (uint32_t)value;
            sim->read_buf = -1;  // (whatever was filed is dropped: the next step walks the envs in index order)
            return PCC_OK;
        case PCC_TUNE_RETIRE_WIDE_PREDICT:
            if (!(value >= 0.0)) return fail(PCC_EINVAL, "retire_wide_predict out of range");
            sim->d.retire_wide_predict = value >= 1e9 ? 1e9f : (float)value;
            return PCC_OK;
        case PCC_TUNE_HEAVY_ITEM_PACKETS:
            if (!(value >= 0.0 && value <= 1e9)) return fail(PCC_EINVAL, "heavy_item_packets out of range");
            sim->d.heavy_item_packets = (float)value;
            return PCC_OK;
        case PCC_TUNE_SEND_WAVES:
            if (!(value >= 1.0 && value <= 32.0)) return fail(PCC_EINVAL, "send_waves out of range");
            sim->d.send_waves = (uint32_t)value;
            return PCC_OK;
        case PCC_TUNE_TAKEOVER_LANES:
            if (value < 0 || value > 64) return fail(PCC_EINVAL, "takeover_lanes out of range");
            sim->d.takeover_lanes = (uint32_t)value;
            return PCC_OK;
        default: return fail(PCC_EINVAL, "unknown tuning key %d", key);
    }
}

int pcc_set_ring_pools(pcc_sim_t *sim, uint32_t div1, uint32_t div2, uint32_t div3) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    if (sim->send_pending) return fail(PCC_ESTATE, "pcc_set_ring_pools between pcc_step_send and pcc_step_retire");
    const unsigned div[kMaxTiers] = {1, div1, div2, div3};
    for (int c = 1; c < kMaxTiers; c++)
        if (div[c] < 1) return fail(PCC_EINVAL, "pool divisors must be >= 1 (1 = a slot for every sender)");
    DeviceGuard guard(sim->device);
    if (hipDeviceSynchronize() != hipSuccess) return fail(PCC_EHIP, "hipDeviceSynchronize failed");
    Dev &d = sim->d;
    int32_t tops[kMaxTiers] = {0, 0, 0, 0};
    for (int c = 1; c < d.n_tiers; c++) {
        const int rc = alloc_tier(sim, c, div[c]);
        if (rc != PCC_OK) return rc;
        tops[c] = (int32_t)sim->tier_slots[c];
    }
    if (hipMemcpy(d.tier_top, tops, sizeof tops, hipMemcpyHostToDevice) != hipSuccess) return fail(PCC_EHIP, "resetting the pool stacks failed");
    const int64_t senders = d.n * d.ns;
    hipLaunchKernelGGL(forget_ring_slots_kernel, dim3((unsigned)((senders + 255) / 256)), dim3(256), 0, nullptr, d);
    if (hipDeviceSynchronize() != hipSuccess) return fail(PCC_EHIP, "forget_ring_slots_kernel failed");
    sim->ever_reset = false;  // whatever was in flight lived in the old pools: a reset must follow
    sim->restarts_pending = false;
    sim->read_buf = -1;
    return PCC_OK;
}

// The event-loop build (event_engine) runs the interval when packets can overtake each other (latency noise) and when
// windows couple two senders' SEND streams to their notifications; it needs a heap and an RTT list per sender.
int update_engine(pcc_sim_t *sim) {
    const int engine = (sim->d.use_noise || (sim->d.use_cwnd && sim->d.ns > 1)) ? 1 : 0;
    if (engine && !sim->noise_blob) {
        DeviceGuard guard(sim->device);
        const size_t per = (size_t)sim->d.n * sim->d.ns * sim->ring_capacity * sizeof(double2);
        void *p = nullptr;
        if (hipMalloc(&p, 2 * per) != hipSuccess)
            return fail(PCC_ENOMEM, "hipMalloc of %zu bytes for the event heaps failed", 2 * per);
        sim->noise_blob = p;
        sim->noise_bytes = 2 * per;
        sim->d.noise_heap = static_cast<double2 *>(p);
        sim->d.noise_rtt = sim->d.noise_heap + (size_t)sim->d.n * sim->d.ns * sim->ring_capacity;
        sim->d.noise_cap = sim->ring_capacity;
    }
    sim->d.engine = engine;
    sim->ever_reset = false;  // in-flight accounting differs / lives in another structure: a reset must follow
    sim->read_buf = -1;
    return PCC_OK;
}

int pcc_set_cwnd_mode(pcc_sim_t *sim, int enable) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    if (sim->send_pending) return fail(PCC_ESTATE, "pcc_set_cwnd_mode between pcc_step_send and pcc_step_retire");
    const int before = sim->d.use_cwnd;
    sim->d.use_cwnd = enable ? 1 : 0;
    const int rc = update_engine(sim);
    if (rc != PCC_OK) sim->d.use_cwnd = before;
    return rc;
}

int pcc_set_latency_noise(pcc_sim_t *sim, int enable, double max_noise) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    if (sim->send_pending) return fail(PCC_ESTATE, "pcc_set_latency_noise between pcc_step_send and pcc_step_retire");
    if (enable && (!(max_noise >= 1.0) || !(max_noise <= 16.0))) return fail(PCC_EINVAL, "max_noise must be in [1, 16] (the reference: 1.1)");
    const int before = sim->d.use_noise;
    sim->d.use_noise = enable ? 1 : 0;
    const int rc = update_engine(sim);
    if (rc != PCC_OK) { sim->d.use_noise = before; return rc; }
    if (enable) sim->d.noise_span = max_noise - 1.0;  // random.uniform(a, b) = a + (b - a) * random()
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

int pcc_reset(pcc_sim_t *sim, const uint8_t *mask, float *obs_out, void *stream) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    DeviceGuard guard(sim->device);
    if (!mask) sim->restarts_pending = false;  // everything starts over
    int rc = flush_restarts(sim, static_cast<hipStream_t>(stream));
    if (rc == PCC_OK) rc = launch_reset(sim, mask, 0, 0, obs_out, static_cast<hipStream_t>(stream));
    if (rc != PCC_OK) return rc;
    sim->send_pending = false;
    if (!mask) {
        sim->ever_reset = true;
        sim->lockstep = true;
        sim->host_steps = 0;
        sim->read_buf = -1;  // the filed predictions describe the links that were just replaced
    } else {
        sim->lockstep = false;  // some envs are now at a different step count
    }
    return PCC_OK;
}

namespace {
// host bookkeeping after the MI of a step: episode boundary, auto-reset (ns:444, the gym wrapper's reset)
int after_mi(pcc_sim_t *sim, float *obs_out, int auto_reset, hipStream_t st) {
    const Dev &d = sim->d;
    sim->host_steps++;
    if (auto_reset) {
        // when every env is in lockstep the host knows which step finishes the episode and
        // skips the (otherwise no-op) masked reset launches
        const bool may_be_done = !sim->lockstep || sim->host_steps >= d.max_steps;
        if (may_be_done && !restarts_in_step(sim, auto_reset)) {
            const int rc = launch_reset(sim, nullptr, 1, sim->lockstep ? 0 : 1, obs_out, st);
            if (rc != PCC_OK) return rc;
            if (sim->lockstep) sim->host_steps = 0;
        }
    } else if (sim->lockstep && sim->host_steps >= d.max_steps) {
        sim->lockstep = false;  // caller resets on its own schedule from here on
    }
    return PCC_OK;
}
}  // namespace

int pcc_step_send(pcc_sim_t *sim, const void *actions, int actions_f64, void *stream) {
    if (!sim || !actions) return fail(PCC_EINVAL, "NULL argument");
    if (!sim->ever_reset) return fail(PCC_ESTATE, "pcc_step before pcc_reset (the reference raises TypeError: run_dur is None)");
    if (sim->send_pending) return fail(PCC_ESTATE, "pcc_step_send called twice without pcc_step_retire");
    if (sim->d.engine) return fail(PCC_ESTATE, "the event-loop build (latency noise; congestion window with two senders) has no separate send half: use pcc_step");
    DeviceGuard guard(sim->device);
    const int rc = launch_send(sim, 0, 0, 0, actions, actions_f64, static_cast<hipStream_t>(stream));
    if (rc == PCC_OK) sim->send_pending = true;
    return rc;
}

int pcc_step_retire(pcc_sim_t *sim, float *obs_out, float *reward_out, uint8_t *done_out, double *steps_out,
                    int auto_reset, void *stream) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    if (!sim->send_pending) return fail(PCC_ESTATE, "pcc_step_retire without a preceding pcc_step_send");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rc = launch_retire(sim, 0, 0, 0, 0, restarts_in_step(sim, auto_reset) ? 1 : 0, obs_out, reward_out, done_out,
                                 steps_out, st);
    if (rc != PCC_OK) return rc;
    sim->send_pending = false;
    return after_mi(sim, obs_out, auto_reset, st);
}

int pcc_step(pcc_sim_t *sim, const void *actions, int actions_f64, float *obs_out, float *reward_out,
             uint8_t *done_out, double *steps_out, int auto_reset, void *stream) {
    if (!sim || !actions) return fail(PCC_EINVAL, "NULL argument");
    if (!sim->ever_reset) return fail(PCC_ESTATE, "pcc_step before pcc_reset (the reference raises TypeError: run_dur is None)");
    if (sim->send_pending) return fail(PCC_ESTATE, "pcc_step between pcc_step_send and pcc_step_retire");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Dev &d = sim->d;
    if (d.n < (int64_t)sim->list_min_envs && !d.engine) {
        // a small batch: both halves in one launch (step_small_kernel)
        const dim3 grid((unsigned)((d.n + kWave - 1) / kWave)), block(4 * kWave);
        const bool tr = d.rng_mode == PCC_RNG_TRACE;
#define PCC_LAUNCH_SMALL(NS_, TR_) \
    hipLaunchKernelGGL((step_small_kernel<NS_, TR_>), grid, block, 0, st, d, actions, actions_f64, obs_out, reward_out, done_out, steps_out)
        if (d.ns == 1) { if (tr) PCC_LAUNCH_SMALL(1, true); else PCC_LAUNCH_SMALL(1, false); }
        else { if (tr) PCC_LAUNCH_SMALL(2, true); else PCC_LAUNCH_SMALL(2, false); }
#undef PCC_LAUNCH_SMALL
        const int rc0 = check_hip(hipGetLastError(), "step kernel launch");
        if (rc0 != PCC_OK) return rc0;
        return after_mi(sim, obs_out, auto_reset, st);
    }
    const int rc = launch_mi(sim, 0, 0, 0, 0, restarts_in_step(sim, auto_reset) ? 1 : 0, actions, actions_f64, obs_out, reward_out,
                             done_out, steps_out, st);
    if (rc != PCC_OK) return rc;
    return after_mi(sim, obs_out, auto_reset, st);
}

int pcc_step_many(pcc_sim_t *sim, const void *actions, int actions_f64, int n_steps, float *obs_out, float *reward_out,
                  uint8_t *done_out, int auto_reset, void *stream) {
    if (!sim || !actions || n_steps < 1) return fail(PCC_EINVAL, "NULL argument or n_steps < 1");
    const Dev &d = sim->d;
    const size_t row = (size_t)d.n * d.ns;
    const size_t act_row = row * (d.use_cwnd ? 2u : 1u) * (actions_f64 ? sizeof(double) : sizeof(float));
    for (int t = 0; t < n_steps; t++) {
        const int rc = pcc_step(sim, static_cast<const char *>(actions) + (size_t)t * act_row, actions_f64,
                                obs_out ? obs_out + (size_t)t * row * d.HF : nullptr, reward_out ? reward_out + (size_t)t * row : nullptr,
                                done_out ? done_out + (size_t)t * d.n : nullptr, nullptr, auto_reset, stream);
        if (rc != PCC_OK) return rc;
    }
    return PCC_OK;
}

int pcc_get_state(pcc_sim_t *sim, int field, void *out, void *stream) {
    if (!sim || !out) return fail(PCC_EINVAL, "NULL argument");
    const Dev &d = sim->d;
    const size_t n = (size_t)d.n, sn = n * d.ns;
    // the fields live in one 128-byte block per env / per sender: a strided 2-D copy gathers one
    const char *src = nullptr;
    size_t width = 0, rows = n;
#define PCC_ENV_FIELD(f) src = reinterpret_cast<const char *>(d.env) + offsetof(EnvBlk, f); width = sizeof(EnvBlk::f); rows = n; break
#define PCC_SND_FIELD(f) src = reinterpret_cast<const char *>(d.snd) + offsetof(SndBlk, f); width = sizeof(SndBlk::f); rows = sn; break
    switch (field) {
        case PCC_F_BW: PCC_ENV_FIELD(bw);
        case PCC_F_DL: PCC_ENV_FIELD(dl);
        case PCC_F_LR: PCC_ENV_FIELD(lr);
        case PCC_F_MAXQ: PCC_ENV_FIELD(maxq);
        case PCC_F_QDELAY: PCC_ENV_FIELD(q);
        case PCC_F_QTIME: PCC_ENV_FIELD(tu);
        case PCC_F_NOW: PCC_ENV_FIELD(now);
        case PCC_F_RUN_DUR: PCC_ENV_FIELD(run_dur);
        case PCC_F_STEPS: PCC_ENV_FIELD(steps);
        case PCC_F_EPISODE: PCC_ENV_FIELD(episode);
        case PCC_F_FLAGS: PCC_ENV_FIELD(flags);
        case PCC_F_RATE: PCC_SND_FIELD(rate);
        case PCC_F_RATE0: PCC_SND_FIELD(rate0);
        case PCC_F_NEXT_SEND: PCC_SND_FIELD(next_send);
        case PCC_F_MIN_LAT: PCC_SND_FIELD(min_lat);
        case PCC_F_ACC_HEAD: PCC_SND_FIELD(ha);
        case PCC_F_ACC_TAIL: PCC_SND_FIELD(ta);
        case PCC_F_DROP_HEAD: PCC_SND_FIELD(hd);
        case PCC_F_DROP_TAIL: PCC_SND_FIELD(td);
        case PCC_F_EP_RETURN: PCC_SND_FIELD(ep_return);
        case PCC_F_LAST_RETURN: PCC_SND_FIELD(last_return);
        case PCC_F_TOTAL_SENT: PCC_ENV_FIELD(total_sent);
        case PCC_F_RING_TIER: PCC_SND_FIELD(ring_tier);
        case PCC_F_CWND: PCC_SND_FIELD(cwnd);
        default: return fail(PCC_EINVAL, "unknown field %d", field);
    }
#undef PCC_ENV_FIELD
#undef PCC_SND_FIELD
    DeviceGuard guard(sim->device);
    // envs that restarted in the last step show the state after their warm-up intervals, like after any reset
    if (!sim->send_pending) {
        const int rc = flush_restarts(sim, static_cast<hipStream_t>(stream));
        if (rc != PCC_OK) return rc;
    }
    return check_hip(hipMemcpy2DAsync(out, width, src, 128, width, rows, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)),
                     "pcc_get_state copy");
}

}  // extern "C"
