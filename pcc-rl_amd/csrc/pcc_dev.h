// pcc_dev.h -- MI355X (gfx950) batched congestion-control simulator: what every kernel file shares
// (state layout, the Dev argument block, RNG, the link model, ring addressing, work-list constants).
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
// Kernels per step, each in its own translation unit with its own register budget (csrc/*.hip; the C ABI and the
// launch logic are in pcc_sim.hip):
//   send_kernel          (pcc_send.hip) one launch, two kinds of workgroup: light items off the class lists the previous
//                        retire launch filed -- 64 envs of about the same predicted packet count, a lane each, in rounds
//                        (no loads in the loop) -- and persistent wavefronts over the wave-path items -- an env sent by all
//                        64 lanes, 256 packets per pass, from closed forms (heavy_mi) -- and the team items (the largest
//                        envs, four wavefronts of a workgroup per env);
//   send_restart_kernel  (pcc_send_restart.hip) envs that finished their episode out of lockstep: new links, the two
//                        warm-up intervals (send + retire each), then the first interval;
//   retire_kernel        (pcc_retire.hip) retire_env, 8 or 16 lanes per env: searches of the rings for the hop-2 / hop-1
//                        boundaries (all four advanced together), the MI-ending event, RTT sums as numpy's pairwise
//                        tree, metrics, history, observation, reward, done; then files every env by its predicted
//                        packet count for the next send;
//   step_small_kernel    (pcc_small.hip) both halves of a small batch's step in one launch; reset_init_kernel.
// No MFMA: there is no contraction anywhere on this path.
#pragma once
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

// (the types every translation unit passes around live in a named namespace: one definition, external linkage)
namespace pcc {

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
static __constant__ double c_metric_scale[PCC_N_METRICS] = {1e7, 1e7, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};

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
    uint32_t fill_seq;  // (a shadow block) sequence number of the step whose refill prepared it: usable three steps later
    double q, tu;     //  48  link queue: send half, and the retire half's MI-ending event
    double now, run_dur;            //  64  retire half
    unsigned long long total_sent;  //  80  retire half
    uint32_t steps;
    uint8_t done, resetting;
    uint8_t pad0[2];
    uint32_t mi_draws;  //  96  send half: link-entry draws of the MI (a SEND the window blocks still draws)
    uint32_t ep_draws;  //      ... of the episode: the position in a replayed loss trace
    uint32_t flags;
    uint32_t params_gen;  // (a shadow block) the generation of the caller's parameters -- link arrays, ranges, seed -- its episode was drawn from
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
    uint32_t heap_n;   // event-loop build (event_engine): the sender's events in its heap = its packets in flight; bit 31: the array is
                       // in no particular order (pcc_noise_sorted.hip wrote it; the event loop makes a heap of it first)
    uint32_t send_cost;   // what the wave path spent on this env's last interval: 100 MHz ticks (low 24 bits) | interval index << 24 -- written by the send
                          // half (sender 0's block), read by the retire half when it files the env for the next send: an env whose packets
                          // cost several times the usual (accept-chain passes: 90-150 ns per packet against ~25) is filed as if it had that many
                          // times the packets, so that the send launch starts it early instead of late (speed only; round 6)
    // 112  retire half: where the last interval's four ring boundaries fell, as predictions of the next ones (speed only:
    // search_many verifies them) -- acknowledgements and loss reports per second of simulated time, and the packets that were
    // on the return hop at the interval's end (accepted / dropped ring)
    float ack_rate, loss_rate;
    uint32_t on_return_a, on_return_d;
};
static_assert(sizeof(EnvBlk) == 128 && sizeof(SndBlk) == 128, "one line per block");

// Everything a kernel needs, passed by value.
// What pcc_noise_sorted.hip leaves for the retire launch of the same interval: event_engine's result.
struct NoiseOut {
    double now, q, tu, nsend;
    uint32_t sent, acked, lost, flags;
    uint32_t seq, pad[3];
};
static_assert(sizeof(NoiseOut) == 64, "one 64-byte record per env");

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
    // free slots of the pools (c >= 1): a stack per partition of the envs over its share of the pool's slots.  Addressed by
    // arithmetic and device memory, not by arrays of this struct: a kernel-argument array indexed by a run-time tier inside
    // the loop over the partitions ended up copied to scratch by the compiler
    uint32_t *pool_free;             // [kMaxTiers][pool_free_stride]: tier c's stacks, partition p's at p * pool_share[c]
    size_t pool_free_stride;
    uint32_t *pool_share;            // [kMaxTiers] slots of one partition's share (slots [p * share, (p + 1) * share) of the pool)
    int32_t *tier_top;               // [kMaxTiers][kParts] stack heights, kTopStride words apart
    uint32_t tier_slots[kMaxTiers];  // slots of each pool
    // XCD-affine partitions (see "partitions" below): the envs in `parts` contiguous id ranges of part_envs envs each
    uint32_t parts, parts_shift, part_envs;   // (parts = 1 << parts_shift)
    uint32_t key0, key1, gid_base;
    uint32_t params_gen;   // generation of what a reset draws from (p_bw.., lo / hi, key0 / key1): every setter moves it, and a shadow
                           // prepared under another generation is not swapped in (the env then restarts through the restart list,
                           // which samples at reset time like the reference, ns:455-477)
    double delta_scale;
    uint32_t max_steps;
    uint32_t *cls_count;  // [3][parts][kClsStride] work lists of the send half (three buffers in rotation -- read / filed / cleared for the step after -- a set per partition): envs per class
    uint32_t *cls_list;   // [3][parts][kListRows][part_envs] env ids by class
    uint32_t *cursors;    // [3][parts][kShards][kCursorStride] item cursors of the list buffers
    // the fused step (pcc_fused.hip: one launch, an env's retire half runs as soon as ITS send half is done): per list buffer
    // and partition the light-item cursor and the heads / tails of the two ready queues, a 128-byte line each; and the queues'
    // entries -- 8-byte granules (step sequence number << 32 | env id), written when an env's send half is complete
    uint32_t *fctl;                   // [3][kXcds][kFctlWords][kCursorStride]: word 0 of entry p = partition p's light-item cursor, words 1..4 of
                                      // entry x = heads and tails of XCD x's two queues, word 5 of entry 0 = envs published so far
    unsigned long long *q_entries;    // [2][kXcds][q_cap]: queue 0 = the envs of the light classes (retired 8 lanes per env), 1 = wave-path classes (16 lanes)
    uint32_t q_cap;                   // entries per queue (every env may end up in one)
    uint32_t fused_acquire;           // debug: 2 = an agent-scope acquire (L1 invalidate) before a ready env's state is read (see fused_retire_unit); 0 = none (default)
    uint32_t fused_spin_ticks;        // a wait of the fused step gives up (PCC_FLAG_INTERNAL) after this many 100 MHz ticks
    uint32_t fused_debug;             // debug: bit 0 = an agent-scope release (buffer_wbl2) in front of every publication, bit 2 = no retire work before every env is sent (the halves one after the other inside the launch)
    uint32_t fused_partial_naps;      // tuning: ... and takes a unit of fewer envs than its lanes hold once it has waited this many naps
    uint32_t fused_max_naps;          // tuning: an idle wavefront of the fused step looks at the ready queues every 1, 2, 4 .. this many naps of ~0.9 us
    // [1] the retire half writes the step's sequence number (step_seq, below) here when an env finishes its episode;
    // the gated auto-reset launches of that step run only if they find it.  Nobody ever clears the word: a clear by one
    // workgroup of a launch races with the sets of the others (the L2 of every XCD writes back on its own schedule)
    uint32_t *any_done;
    uint32_t step_seq;    // sequence number of the step this launch belongs to (host counter, never 0)
    uint64_t *timeline;  // profiling only (PCC_DEBUG_TIMELINE env): 8 words per send wavefront, see pcc_debug_timeline
    unsigned long long *pass_stats;  // profiling only (same switch): counters of the wave passes, see pcc_debug_pass_stats
    int pass_counters;               // ... per-pass counters on (PCC_DEBUG_TIMELINE=2: contended atomics, they slow the passes down)
    // tuning (speed only): light items dealt to the workgroups in snake order (a workgroup's four items add up to about the
    // same number of packets); the wave kernel's first items dealt oldest workgroup first (1) or youngest first (0);
    // s_setprio level for the first prio_light_items light items / the first prio_wave_items wave-path items / team items
    uint32_t light_snake, wave_oldest_first, prio_level, prio_light_items, prio_wave_items, prio_team;
    uint32_t retire_sorted;  // debug: 0 = the retire launch walks the envs in index order even when there are lists
    int debug_skip;  // profile build only (PCC_DEBUG_SKIP env): bit0 skip RTT means, bit1 skip history/obs, bit2 skip the
                     // lane rounds' record stores, bit3 skip their Philox -- results wrong, timing only
    uint32_t round_packets, takeover_lanes, send_envs_per_wave, send_waves;
    double heavy_predict;  // predicted packets per MI above which an env goes to the heavy wave
    double team_predict;   // ... above which a whole workgroup sends it (team pass)
    float heavy_item_packets;  // a heavy work item is as many envs of its class as make up about this many packets (1..8 envs)
    float light_half_predict;  // light items of the classes from this many predicted packets up hold 32 envs instead of 64
    float retire_wide_predict; // retire half: envs predicted above this many packets per interval get 16 lanes instead of 8
    double lo[5], hi[5];
    int rng_mode;
    const double *trace;
    int64_t trace_stride;
    const double *p_bw, *p_dl, *p_queue, *p_loss, *p_rate0;
    // State blocks.  Index i < N is env i; index N + i is its SHADOW: the next episode of env i, prepared ahead of time
    // (new links, the two warm-up intervals) by the refill kernel while the env is still running, and swapped in by the
    // retire half the moment the env finishes -- out of lockstep a restart is then no chain of dependent passes in the
    // step's critical path (pcc_send_restart.hip).  stride = 2 N.
    EnvBlk *env;  // [2N] link + env state, one 128-byte block per env
    SndBlk *snd;  // [S][2N] per sender, one 128-byte block each
    int64_t stride;       // 2 N: sender s of block i is snd[s * stride + i]
    char *shadow_rings;   // [N][S] private rings of the shadows, tier-1 size each (the warm-up intervals' packets in flight)
    uint32_t *refill_count;  // [4][kCntStride] envs whose shadow was swapped in (or invalidated) at step seq: row seq & 3
    uint32_t *refill_list;   // [4][N]
    int shadows;          // the retire half may swap shadows in (Philox uniforms, envs out of lockstep, lists on)
    unsigned long long *restart_stats;  // [2] episodes started by a shadow swap / through the restart list (running totals)
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
    NoiseOut *noise_out;  // [N]: an interval run ahead of the retire launch by pcc_noise_sorted.hip (nullptr: never)
    uint32_t noise_seq;    // ... counts the intervals launched: NoiseOut::seq == noise_seq says "this one has been run"
    float *hist;    // [N][S][HF]
    double2 *ring;  // [N][S][2][cap]: accepted ring, dropped ring
};

}  // namespace pcc
using namespace pcc;

namespace {

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

// instruction-issue priority of this wavefront among those of its SIMD (level wave-uniform; s_setprio takes an immediate)
__device__ __forceinline__ void set_prio(uint32_t level) {
    if (level == 1u) __builtin_amdgcn_s_setprio(1);
    else if (level == 2u) __builtin_amdgcn_s_setprio(2);
    else if (level >= 3u) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
}

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

// The same block cipher with the products written as 64-bit multiplies the compiler sees (it selects v_mad_u64_u32 for
// them on gfx950 when both halves of a product are used): unlike the inline assembly above, the instruction scheduler
// may interleave these rounds with independent work of the same basic block (the lane rounds, pcc_send_item.h).
__device__ __forceinline__ void philox4x32_10_sched(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                    uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)c0 * 0xD2511F53u, p1 = (uint64_t)c2 * 0xCD9E8D57u;
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
#if defined(PCC_NT_LD) && PCC_NT_LD   // (experiment: the records are read once -- nontemporal loads)
    const gvec2 v = __builtin_nontemporal_load((const PCC_GLOBAL gvec2 *)(const void *)p);
#else
    const gvec2 v = *(const PCC_GLOBAL gvec2 *)(const void *)p;
#endif
    double2 r;
    r.x = v.x; r.y = v.y;
    return r;
}
__device__ __forceinline__ void st_rec(double2 *p, const double2 &r) {
    gvec2 v;
    v.x = r.x; v.y = r.y;
#if defined(PCC_NT_ST) && PCC_NT_ST   // (experiment: every record store nontemporal -- the lane rounds' too: 3x slower)
    __builtin_nontemporal_store(v, (PCC_GLOBAL gvec2 *)(void *)p);
#else
    *(PCC_GLOBAL gvec2 *)(void *)p = v;
#endif
}
// ---- nontemporal record stores (round 6).  A record store that fills whole ring lines -- the wave path's staged runs: 1 KB of
// consecutive ring bytes per instruction -- is issued nontemporal (`global_store_dwordx4 ... nt`).  Measured, not derived: the
// send launch is unchanged and the RETIRE launch that reads the records one to three launches later is 8-10 % faster
// (0.0892 -> 0.0813 ms at 65 536 envs, tools/ab_libraries.py, profiles/r06_nontemporal.json): the streamed lines do not
// displace the env and sender blocks both launches re-read every step.  The lane rounds' scattered 16-byte stores must NOT be
// nontemporal: they become partial writes to memory (send launch 0.091 -> 0.287 ms), and so must not the position-by-position
// stores of a pass (two senders: 0.170 -> 0.312 ms) -- hence st_rec_run only where a wavefront writes runs of whole lines.
// Loads: nontemporal loads of the records in the retire half pin its launch at 0.087 ms whatever the box's mode (0.085 / 0.090)
// and gain nothing on top of the stores; -DPCC_NT_LD=1 / 2 builds them.  -DPCC_NT_RUN=0 builds plain stores.
#ifndef PCC_NT_RUN
#define PCC_NT_RUN 1
#endif
__device__ __forceinline__ void st_rec_nt(double2 *p, const double2 &r) {
    gvec2 v;
    v.x = r.x; v.y = r.y;
    __builtin_nontemporal_store(v, (PCC_GLOBAL gvec2 *)(void *)p);
}
// a record of a run of consecutive ring slots written by consecutive lanes (whole ring lines per instruction)
__device__ __forceinline__ void st_rec_run(double2 *p, const double2 &r) {
#if PCC_NT_RUN
    st_rec_nt(p, r);
#else
    st_rec(p, r);
#endif
}
// a record of a wave pass stored position by position: neighbouring lanes write neighbouring slots of a few dense runs and four
// instructions fill the lines between them -- plain stores (see above; -DPCC_NT_RUN=2 is the experiment)
__device__ __forceinline__ void st_rec_pass(double2 *p, const double2 &r) {
#if PCC_NT_RUN >= 2
    st_rec_nt(p, r);
#else
    st_rec(p, r);
#endif
}
#if defined(PCC_NT_LD) && PCC_NT_LD >= 2
__device__ __forceinline__ double ld_f64(const void *p) { return __builtin_nontemporal_load((const PCC_GLOBAL double *)p); }
#else
__device__ __forceinline__ double ld_f64(const void *p) { return *(const PCC_GLOBAL double *)p; }
#endif
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

// ring_tier of a sender whose rings are still its (former) shadow's private ones: the next send half moves the records
// in flight into the sender's own storage (load_env), after which the shadow can be refilled
constexpr uint32_t kTierBorrowed = 0xFEu;
constexpr uint32_t kShadowTier = 1u;  // a shadow's private rings have the size of tier 1

// A shadow's state word -- the aligned 32 bits that hold EnvBlk::done (always 0 in a shadow) and EnvBlk::resetting (0 = ready,
// 1 = being refilled, 2 = listed for a refill, 3 = unusable) -- is moved with atomics wherever two launches can meet:
//   shadow_list: "to be refilled", and whether the caller has to append it to a refill row (it was not listed already: a
//                swap and a masked reset of the same step, or two restarts, would otherwise list it twice -- two wavefronts
//                refilling one shadow at once, and a row of N ids overflowing);
//   shadow_claim (refill_kernel): listed -> being refilled, for exactly one wavefront.
static_assert(offsetof(EnvBlk, resetting) == offsetof(EnvBlk, done) + 1 && offsetof(EnvBlk, done) % 4 == 0, "the shadow's state word");
__device__ __forceinline__ bool shadow_list(EnvBlk *sh) {
    const uint32_t old = atomicExch(reinterpret_cast<uint32_t *>(&sh->done), 2u << 8);
    return ((old >> 8) & 0xFFu) != 2u;
}
__device__ __forceinline__ bool shadow_claim(EnvBlk *sh) {
    return atomicCAS(reinterpret_cast<uint32_t *>(&sh->done), 2u << 8, 1u << 8) == (2u << 8);
}

__device__ __forceinline__ int64_t sidx(const Dev &D, int s, int64_t i) { return (int64_t)s * D.stride + i; }
__device__ __forceinline__ int64_t env_of(const Dev &D, int64_t i) { return i >= D.n ? i - D.n : i; }  // block index -> env id

// Partitions of the batch (see "partitions" below): contiguous id ranges with work lists and pool stacks of their own
constexpr uint32_t kParts = 8;
constexpr uint32_t kTopStride = 32;   // words between two pool-stack heights: a line each (they are atomics)
__device__ __forceinline__ uint32_t part_of(const Dev &D, const int64_t env) { return (uint32_t)env / D.part_envs; }  // env < N

// A free slot of pool c for an env of partition p0: its own partition's stack first (the slots of a partition are one
// contiguous share of the pool), the others' when that one is empty.  -1: the pool is empty.  (Pops happen only in send
// launches, pushes only in reset and retire launches: no stack races.)
__device__ __forceinline__ int64_t pool_pop(const Dev &D, const uint32_t c, const uint32_t p0) {
    for (uint32_t k = 0; k < D.parts; k++) {
        uint32_t p = p0 + k;
        if (p >= D.parts) p -= D.parts;
        int32_t *top = D.tier_top + (size_t)(c * kParts + p) * kTopStride;
        const int32_t old = atomicSub(top, 1);
        if (old > 0) return (int64_t)D.pool_free[(size_t)c * D.pool_free_stride + (size_t)p * D.pool_share[c] + (uint32_t)(old - 1)];
        atomicAdd(top, 1);
    }
    return -1;
}
__device__ __forceinline__ void pool_push(const Dev &D, const uint32_t c, const uint32_t slot) {
    const uint32_t share = D.pool_share[c];
    uint32_t p = slot / share;
    if (p >= D.parts) p = D.parts - 1u;
    int32_t *top = D.tier_top + (size_t)(c * kParts + p) * kTopStride;
    D.pool_free[(size_t)c * D.pool_free_stride + (size_t)p * share + (uint32_t)atomicAdd(top, 1)] = slot;
}

__device__ __forceinline__ uint32_t tier_cap(const Dev &D, uint32_t tier) { return D.cap0 << (2u * (tier == kTierBorrowed ? kShadowTier : tier)); }
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

// how many bits of `mask` (a ballot) lie below this lane: v_mbcnt_lo / v_mbcnt_hi on the mask's scalar halves -- two
// instructions and no lane mask in vector registers (__popcll(mask & ((1ull << lane) - 1)) is five and two registers)
__device__ __forceinline__ uint32_t count_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t rl_u32(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
// v of the lane below (DPP wave_shr:1, no LDS); lane 0 gets `first`
__device__ __forceinline__ double wave_shr1_f64(double v, double first) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
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
// want == 0 (only for a sender on borrowed rings, see kTierBorrowed): into the sender's own tier-0 rings at `own0`.
__device__ __forceinline__ bool promote_rings(const Dev &D, uint32_t lane, uint32_t l, int64_t k, uint32_t want,
                                              uint32_t ha, uint32_t ta, uint32_t hd, uint32_t td, char *own0 = nullptr) {
    uint32_t got = 0xFFFFFFFFu, slot = 0;
    if (lane == l) {
        if (want == 0u) got = 0u;
        const uint32_t p0 = part_of(D, env_of(D, k % D.stride));
        for (uint32_t c = want; got == 0xFFFFFFFFu && c < (uint32_t)D.n_tiers; c++) {
            const int64_t sl = pool_pop(D, c, p0);
            if (sl >= 0) { slot = (uint32_t)sl; got = c; break; }
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
    if (got == 0u) to.base = reinterpret_cast<char *>(rl_u64(reinterpret_cast<uint64_t>(own0), l));  // (back into the sender's own tier-0 rings)
    const uint32_t h_a = rl_u32(ha, l), n_a = rl_u32(ta, l) - h_a, h_d = rl_u32(hd, l), n_d = rl_u32(td, l) - h_d;
    copy_records(to.accepted(), to.mask(), from.accepted(), from.mask(), h_a, n_a, lane);
    copy_records(to.dropped(), to.dmask(), from.dropped(), from.dmask(), h_d, n_d, lane);
    if (lane == l) {
        D.snd[k].ring_base = to.base;
        D.snd[k].ring_tier = (uint8_t)got;
        if (got) D.snd[k].ring_held[got] = slot + 1u;
    }
    return true;
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

// ---- partitions ----------------------------------------------------------------------------
// A scattered 16-byte access costs a compute unit 2.5x as much when the addresses its XCD touches span more than ~2 GB
// than when they span less (tools/microbench/store_bench4: 200 G records/s chip-wide up to 2 GB, 160 at 3 GB, 80 from 4 GB
// up -- the reach of an XCD's address translation, not a bandwidth), and the rings of 65 536 envs span 1.5 GB in tier 0 alone,
// plus the pools.  So the batch is cut into kParts contiguous id ranges, each with work lists, item cursors and pool stacks
// of its own: workgroup b of a launch works for partition b % kParts -- the hardware places block b on XCD b % 8 (observed,
// MI355X_MICROARCH.md; nothing depends on it but speed) -- so an XCD keeps touching the same eighth of tier 0 and of every
// pool.  Which workgroup handles an env never changes a result.  Batches too small for it (and the event-loop builds) run
// as one partition.

// the list set of partition `part` in buffer `buf`
__device__ __forceinline__ uint32_t list_view(const Dev &D, const int buf, const uint32_t part) { return (uint32_t)buf * D.parts + part; }
__device__ __forceinline__ uint32_t *cls_count_of(const Dev &D, const uint32_t view, const uint32_t row) {
    return D.cls_count + ((size_t)view * kClsStride + (size_t)row * kCntStride);
}
__device__ __forceinline__ uint32_t *cls_list_of(const Dev &D, const uint32_t view, const uint32_t row) {
    return D.cls_list + ((size_t)view * kListRows + row) * (size_t)D.part_envs;
}
__device__ __forceinline__ uint32_t *cursors_of(const Dev &D, const uint32_t view) { return D.cursors + (size_t)view * kShards * kCursorStride; }
constexpr int kListBufs = 3;   // list buffers in rotation: the one a step reads, the one it files into, the one it clears for the next step
// the fused step's words of a list buffer (a line each; Dev::fctl): per partition the light-item cursor, per XCD the ready queues'
// heads (claimed) and tails (reserved), and the count of published envs
constexpr uint32_t kFctlWords = 8;
constexpr uint32_t kFLight = 0, kFHead = 1 /* + queue */, kFTail = 3 /* + queue */;
// ---- the fused step's ready queues (pcc_fused.hip) ----------------------------------------
// Per-XCD L2s are write-back and NOT coherent with each other, and making a send half's stores visible across XCDs costs more
// than the fused step gains (measured: write-through record stores +0.05 ms, one buffer_wbl2 per light item +0.09 ms, a
// buffer_inv per retire unit +0.03 ms on a 0.18 ms step).  So the queues are per PHYSICAL XCD: a wavefront publishes an env in
// the queue of the XCD it runs on (XCC_ID, read from the hardware) and takes envs only from that queue -- producer and
// consumer then share one L2 BY CONSTRUCTION, wherever the dispatcher put their workgroups, and inside one L2 a plain store
// is visible once the storing wavefront's s_waitcnt vmcnt(0) has passed (the gfx90a rule: one L2 = the coherence point).  The
// consumer's L1 cannot hold a line of the env unless its own compute unit ran the env's send half (nobody else loads an env's
// lines in a launch), and a compute unit's L1 is coherent with that unit's own stores.
// An entry is one 8-byte granule: step sequence number << 32 | env id (the data is its own flag).
constexpr uint32_t kXcds = 8;              // XCC_ID & 7: the MI355X has 8
constexpr uint32_t kFPushed = 5;           // fctl word of XCD x: envs published on x so far (one counter for all XCDs is ~7 000
                                           // atomics on one word per step: 80 us of the L2 channel that owns it)
__device__ __forceinline__ void fused_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return xcc & (kXcds - 1u);
}
// the words of buffer `buf` that belong to XCD x (heads, tails), and the entries of its queue q
__device__ __forceinline__ uint32_t *fq_word(const Dev &D, const int buf, const uint32_t x, const uint32_t word) {
    return D.fctl + (((size_t)buf * kXcds + x) * kFctlWords + word) * kCursorStride;
}
__device__ __forceinline__ unsigned long long *fused_entries(const Dev &D, const uint32_t q, const uint32_t x) {
    return D.q_entries + ((size_t)q * kXcds + x) * (size_t)D.q_cap;
}
// The envs of the lanes in `mask` (lane l: env i) are sent -- every wavefront that stored for them has waited for vmcnt(0)
// -- : into ready queue q of this wavefront's XCD.  The XCD's count of published envs goes up AFTER the granules are out: who
// reads the counts of all XCDs and finds every env of the step knows that every queue's tail is final.
__device__ __forceinline__ void fused_push(const Dev &D, const int buf, const uint32_t xcc, const uint32_t q, const uint64_t mask,
                                           const uint32_t lane, const int64_t i) {
    if (!mask) return;
    if (D.fused_debug & 1u) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    const uint32_t n = (uint32_t)__popcll(mask);
    uint32_t base = 0;
    if (lane == 0) base = __hip_atomic_fetch_add(fq_word(D, buf, xcc, kFTail + q), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if ((mask >> lane) & 1ull)
        __hip_atomic_store(fused_entries(D, q, xcc) + base + count_below(mask), ((unsigned long long)D.step_seq << 32) | (uint32_t)i,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fused_drain();
    if (lane == 0) (void)__hip_atomic_fetch_add(fq_word(D, buf, xcc, kFPushed), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (send_kernel itself follows retire_env below: a restart item runs the env's warm-up intervals through both halves)

}  // namespace
