/*
 * pcc_sim.h -- C ABI of the MI355X-native batched congestion-control simulator.
 *
 * One shared library (pcc-rl_amd/lib/libpcc_sim.so, built by hipcc for gfx950) exports
 * exactly the entry points below.  They are what a binding for the reference's env path
 * would call: the reference (PCCproject/PCC-RL) is pure Python and has no native interface,
 * so every function cites the Python method it replaces ("ns" = src/gym/network_sim.py,
 * "so" = src/common/sender_obs.py in the reference tree).
 *
 * Conventions
 *   - all functions return 0 on success, a negative PCC_E* code on failure;
 *     pcc_last_error() returns a thread-local description of the last failure.
 *   - "device pointer" = memory of the GPU the handle was created on.  The caller (PyTorch,
 *     through tensor.data_ptr()) owns every in/out buffer; the handle owns the persistent
 *     structure-of-arrays env state and the per-env in-flight packet rings.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *     enqueue work on it; nothing here synchronizes the device, and no call copies between
 *     host and device except pcc_create/pcc_set_param_ranges (scalars).
 *   - a handle is not thread-safe; distinct handles are independent (the reference's module
 *     globals -- Sender._next_id ns:229, _conn_min_latencies so:158, the global RNG ns:73 --
 *     have no counterpart here).
 *   - batched shapes: N = n_envs, S = n_senders, H = history_len, F = n_features.
 */
#ifndef PCC_SIM_H
#define PCC_SIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcc_sim pcc_sim_t;

enum {
    PCC_OK = 0,
    PCC_EINVAL = -1,   /* bad argument */
    PCC_ENODEV = -2,   /* no usable gfx950 device / HIP runtime error at create */
    PCC_ENOMEM = -3,   /* device allocation failed */
    PCC_EHIP = -4,     /* HIP runtime error while enqueueing */
    PCC_ESTATE = -5    /* protocol error: step before reset (ns:368 leaves run_dur None) */
};

/* metric ids = positions in the reference registry so:193-206 */
enum {
    PCC_M_SEND_RATE = 0, PCC_M_RECV_RATE, PCC_M_RECV_DUR, PCC_M_SEND_DUR, PCC_M_AVG_LATENCY,
    PCC_M_LOSS_RATIO, PCC_M_ACK_LATENCY_INFLATION, PCC_M_SENT_LATENCY_INFLATION,
    PCC_M_CONN_MIN_LATENCY, PCC_M_LATENCY_INCREASE, PCC_M_LATENCY_RATIO, PCC_M_SEND_RATIO,
    PCC_N_METRICS
};

/* columns of the optional per-step record (doubles): the counters and clocks of ns:291-317,
 * the reward of ns:194,205 and the 12 metrics of so:110-191 */
enum {
    PCC_COL_SENT = 0, PCC_COL_ACKED, PCC_COL_LOST, PCC_COL_RATE, PCC_COL_CUR_TIME, PCC_COL_RUN_DUR,
    PCC_COL_REWARD, PCC_COL_METRIC0, PCC_STEP_COLS = PCC_COL_METRIC0 + PCC_N_METRICS
};

/* where the per-packet loss uniform (random.random() in ns:73) comes from */
enum {
    PCC_RNG_PHILOX = 0, /* Philox4x32-10 keyed by (seed, global env id, episode, MI, packet) */
    PCC_RNG_TRACE = 1   /* replay u[env][k], k = packets sent so far this episode (parity mode) */
};

/* fields readable with pcc_get_state (element type, shape) */
enum {
    PCC_F_BW = 0,        /* f64 [N]     link bandwidth, packets/s            (ns:59)  */
    PCC_F_DL,            /* f64 [N]     one-way propagation delay, s         (ns:60)  */
    PCC_F_LR,            /* f64 [N]     random loss probability              (ns:61)  */
    PCC_F_MAXQ,          /* f64 [N]     max queue delay = queue/bw           (ns:64)  */
    PCC_F_QDELAY,        /* f64 [N]     link-0 queue delay                   (ns:62)  */
    PCC_F_QTIME,         /* f64 [N]     link-0 queue delay update time       (ns:63)  */
    PCC_F_NOW,           /* f64 [N]     network clock                        (ns:102) */
    PCC_F_RUN_DUR,       /* f64 [N]     next MI duration                     (ns:437-438,467) */
    PCC_F_STEPS,         /* u32 [N]     steps taken this episode             (ns:418) */
    PCC_F_EPISODE,       /* u32 [N]     episodes started (resets)                     */
    PCC_F_FLAGS,         /* u32 [N]     sticky error bits, PCC_FLAG_*                 */
    PCC_F_RATE,          /* f64 [S][N]  sending rate, packets/s              (ns:211) */
    PCC_F_RATE0,         /* f64 [S][N]  starting rate                        (ns:210) */
    PCC_F_NEXT_SEND,     /* f64 [S][N]  time of the pending SEND event       (ns:111,161) */
    PCC_F_MIN_LAT,       /* f64 [S][N]  connection min of per-MI mean RTT, 0 = none (so:158-176) */
    PCC_F_ACC_HEAD,      /* u32 [S][N]  accepted packets acknowledged this episode    */
    PCC_F_ACC_TAIL,      /* u32 [S][N]  packets accepted by the queue this episode    */
    PCC_F_DROP_HEAD,     /* u32 [S][N]  dropped packets whose loss was reported       */
    PCC_F_DROP_TAIL,     /* u32 [S][N]  packets dropped (random loss or tail drop)    */
    PCC_F_EP_RETURN,     /* f64 [S][N]  reward summed over the running episode (ns:442) */
    PCC_F_LAST_RETURN,   /* f64 [S][N]  return of the last finished episode            */
    PCC_F_TOTAL_SENT,    /* u64 [N]     packets sent since create (all episodes, all senders) */
    PCC_F_RING_TIER,     /* u8  [S][N]  tier of the sender's in-flight rings (0 = its own small rings) */
    PCC_F_CWND,          /* u32 [S][N]  congestion window, packets (pcc_set_cwnd_mode)    (ns:227) */
    PCC_N_FIELDS
};

#define PCC_FLAG_RING_OVERFLOW 1u  /* more accepted or dropped packets in flight than ring_capacity: results invalid */
#define PCC_FLAG_TRACE_OVERRUN 2u  /* PCC_RNG_TRACE ran past trace_stride */
#define PCC_FLAG_POOL_EXHAUSTED 8u /* a sender needed a bigger ring tier and every pool from that tier up was empty (it may
                                      then also overflow: RING_OVERFLOW); raise the pools with pcc_set_ring_pools */
#define PCC_FLAG_INTERNAL 4u       /* an internal error: a wave-path loop of the send half did not finish an interval within
                                      4 M passes (every pass sends at least one packet) and gave up instead of spinning;
                                      results invalid */
#define PCC_FLAG_BAD_PARAMS 16u    /* pcc_set_link_params gave this env a link outside what the exact formulation covers
                                      (bw in (0, 1e8], latency > 0, queue >= 1, loss in [0, 1], rate0 > 0 and finite):
                                      results invalid; the env runs on a harmless stand-in link so that no kernel spins */
#define PCC_FLAG_BAD_ACTION 32u    /* an action was NaN (the reference would carry the NaN into its rate and clock): it was
                                      applied as 0.0 */
#define PCC_FLAG_TIME_RANGE 64u    /* the env's clock left the range in which event times 1e-12 apart (relative) are still far
                                      closer than one packet time 1/bw -- the assumption behind the ordering of dropped
                                      packets: shorten the episode (pcc_set_max_steps) or slow the link */

/* last error text of the calling thread ("" if none) */
const char *pcc_last_error(void);

/*
 * Create a batch of n_envs independent envs.  Replaces SimulatedNetworkEnv.__init__
 * (ns:346-394) for a whole batch: history_len and the feature list are its two constructor
 * arguments (ns:347-351); feature_ids are indices into the metric registry (so:193-206).
 *   n_senders       1 (the reference env, ns:466) or 2 (two senders on the shared bottleneck).
 *   seed            Philox key.  env_gid_base: global id of env 0 (rank * n_envs when the
 *                   batch is sharded over GPUs) so results do not depend on the sharding.
 *   ring_capacity   power of two: the most accepted packets one sender can have in flight (0 =
 *                   default 32768); twice as many dropped packets.  The worst case of the
 *                   default ranges is rate_max * (RTT_max + one MI) = 1000 * (30.8 + 15.4) =
 *                   46.2k packets in flight between two retire passes (nearly all of them
 *                   drops), of which at most bw_max * that time = 23k are accepted.
 *                   Storage is tiered: every sender owns small rings (ring_capacity / 4^k
 *                   records with k <= 3 chosen so that this is >= 256, i.e. 512 + 1024 by
 *                   default) and is moved -- at the start of a monitor interval whose packets
 *                   could overflow them -- into rings 4x, 16x, ... as large taken from shared
 *                   pools.  Pool rings are held until the env is reset, so a pool with a slot for
 *                   every sender can never run dry; by default the pools are allocated at the handle's
 *                   FIRST pcc_reset (not here: a caller that names their sizes with pcc_set_ring_pools allocates
 *                   them once, and handles created side by side do not all size themselves from the same free
 *                   figure) and get what a third of the
 *                   device memory that is free then pays for (65 536 senders on an idle
 *                   288 GB MI355X: a slot for every sender in tiers 1 and 2 and for every second one
 *                   in tier 3, 83 GB), never less than slots for 1/2, 1/8, 1/32 of the senders
 *                   (6.4 GB at 65 536: what U(-1, 1) policies need; pcc_set_ring_pools sets the
 *                   divisors explicitly, 1 = a slot for every sender).  An empty pool is flagged
 *                   (PCC_FLAG_POOL_EXHAUSTED), never silent.  pcc_device_bytes reports the total.
 *   device_id       HIP device ordinal (-1 = current device).
 * No env is usable before pcc_reset.
 */
int pcc_create(int64_t n_envs, int n_senders, int history_len, const int32_t *feature_ids,
               int n_features, uint64_t seed, uint32_t env_gid_base, uint32_t ring_capacity,
               int device_id, pcc_sim_t **out);

void pcc_destroy(pcc_sim_t *sim);

/*
 * Link/sender parameters used by every following reset: replaces the five global-RNG draws
 * of create_new_links_and_senders (ns:454-467).  Device pointers: bw, dl, queue (packets,
 * as double), loss are [N]; rate0 is [S][N].  All five NULL = draw them per reset from the
 * ranges (default), exactly as ns:455-466 does: bw~U, lat~U, queue=1+int(exp(U)), loss~U,
 * rate0=U*bw.  The arrays are read at reset time and must stay valid until then.
 */
int pcc_set_link_params(pcc_sim_t *sim, const double *bw, const double *dl, const double *queue,
                        const double *loss, const double *rate0);

/* ranges of ns:355-358 and the rate factor of ns:466: {bw, lat, queue exponent, loss, rate0/bw}
 * (host pointers to 5 doubles each) */
int pcc_set_param_ranges(pcc_sim_t *sim, const double *lo, const double *hi);

/* loss-uniform source.  trace: device pointer [N][trace_stride] doubles (PCC_RNG_TRACE only) */
int pcc_set_rng(pcc_sim_t *sim, int mode, const double *trace, int64_t trace_stride);

/* re-key the Philox generator; takes effect for draws made after the call.  (The reference's
 * seed(), ns:396-398, creates an RNG nothing reads; here it does what a caller expects.) */
int pcc_set_seed(pcc_sim_t *sim, uint64_t seed);

/* Performance knobs; results never depend on them (every send path is exact).
 * The retire half files every env by its predicted packet count for the next interval; the send
 * half takes work items off those lists.  An env predicted above HEAVY_PREDICT packets is sent by all 64
 * lanes of a wavefront 256 packets per pass (the wave kernel, persistent wavefronts, largest class
 * first); lighter envs go SEND_ENVS_PER_WAVE (64) of about the same length at a time to one wavefront
 * of the light kernel, a lane each, in rounds of ROUND_PACKETS (default 256); when at most
 * TAKEOVER_LANES (default 1) lanes still have packets left after a round, those envs are finished by
 * the whole wavefront.  Both kinds of workgroup are in one launch (send_kernel); envs that restart out of
 * lockstep are a kernel of their own on a side stream of the handle, joined before the call returns.
 * (Keys 0, 1, 6, 7 belonged to mechanisms of earlier builds and are rejected.) */
enum { PCC_TUNE_ROUND_PACKETS = 2, PCC_TUNE_TAKEOVER_LANES = 3,
       PCC_TUNE_SEND_ENVS_PER_WAVE = 4 /* envs per light work item, 1..64 */,
       PCC_TUNE_HEAVY_PREDICT = 5 /* predicted packets per interval above which an env is a work item of
                                    its own (wave path); default 480 (two senders: 640), 0 = every env, >= 1e9 = none */,
       PCC_TUNE_SEND_WAVES = 8 /* persistent wavefronts of the wave kernel per compute unit, 1..32 (default 13) */,
       PCC_TUNE_TEAM_PREDICT = 9 /* predicted packets per interval above which an env is sent by a whole workgroup (four
                                    wavefronts, 1 024 packets per pass); default 4096, >= 1e9 = never (one sender only) */,
       PCC_TUNE_HEAVY_ITEM_PACKETS = 10 /* a wave-path work item holds as many envs of one class (1..8) as make up about this
                                    many predicted packets; default 2048, 0 = one env per item */,
       PCC_TUNE_RETIRE_WIDE_PREDICT = 11 /* retire half: an env predicted above this many packets per interval is retired by
                                    16 lanes (whole-list and half sums side by side), the others by 8; default 256 --
                                    but never more envs than the launch's grid has room for (RETIRE_GRID_FRAC: the largest);
                                    0 = as many by 16 as fit, >= 1e9 = every env by 8 */,
       PCC_TUNE_LIST_MIN_ENVS = 12 /* batches of fewer envs are stepped without work lists, the envs in index order (a small
                                    batch's step is a chain of dependent loads, and the lists add three); default 8192,
                                    0 = always with lists */,
       PCC_TUNE_RETIRE_SORTED = 13 /* debug: 0 = the retire launch walks the envs in index order even when there are lists */,
       PCC_TUNE_LIGHT_SNAKE = 14 /* light kernel: a workgroup's four items in snake order over the ranking (their packets add up
                                    to about the same for every compute unit); default 1 */,
       PCC_TUNE_WAVE_OLDEST_FIRST = 15 /* wave kernel: the largest items go to the workgroups dispatched first (1, default) or last */,
       PCC_TUNE_PRIO_LEVEL = 16 /* s_setprio level (0..3, default 0 = off) of the items the next three keys select */,
       PCC_TUNE_PRIO_LIGHT_ITEMS = 17 /* ... the first (longest) this many light items */,
       PCC_TUNE_PRIO_WAVE_ITEMS = 18 /* ... the first (largest) this many wave-path items */,
       PCC_TUNE_PRIO_TEAM = 19 /* ... team items (0 / 1) */,
       PCC_TUNE_SPLIT_STREAMS = 20 /* gone (an experiment of round 4, measured slower: the two kinds of send workgroup as two kernels
                                    on two streams); only 0 is accepted */,
       PCC_TUNE_LIGHT_FRONT_WGS = 21 /* gone (light workgroups dispatched in front of the wave-path ones); only 0 is accepted */,
       PCC_TUNE_RETIRE_GRID_FRAC = 22 /* retire launch: the grid is n / 16 workgroups plus this share of as many again (for envs of
                                    the 16-lane classes, 8 per workgroup); workgroups loop when there are more.  Default 0.125;
                                    1 = the worst case (twice n / 16: the dispatch of ~8 200 workgroups alone takes 0.1 ms) */,
       PCC_TUNE_RESTART_FORK = 23 /* out of lockstep with shadows: the restart kernel (nearly always without work) beside the main
                                    send launch on a side stream (1) or behind it on the caller's stream (0, default: measured faster) */,
       PCC_TUNE_PARTS = 24 /* partitions of the batch (1 or 8): contiguous env-id ranges with work lists, item cursors and pool
                                    stacks of their own; workgroup b of a launch works for partition b % 8, i.e. an XCD keeps to
                                    one eighth of the rings (a scattered access costs 2.5x as much once the addresses an XCD
                                    touches span more than ~2 GB).  Default 8 from 8 192 envs up, else 1.  Changing it puts every
                                    sender back into its own tier-0 rings: pcc_reset must follow */,
       PCC_TUNE_LIGHT_HALF_PREDICT = 25 /* send launch: light items of the classes from this many predicted packets per interval up
                                    hold 32 envs instead of 64 (a lane-round iteration costs ~3.3 ns per lane that stores, and the
                                    longest light items are the launch's critical path); >= 1e9 = none */,
       PCC_TUNE_FUSED = 26 /* both halves of a full-size step in ONE launch, an env's retire half running as soon as its own send
                                    half is done (pcc_step only; whenever the step has work lists to read and no restart list to
                                    serve; pcc-rl_amd/csrc/pcc_fused.hip): 0 (default) = the send launch and the retire launch -- the
                                    one-launch step is exact (the parity suite runs through it) and measured SLOWER at full size
                                    (0.33 ms per step against 0.18: profiles/r05_fused_experiments.json); 1 = on; 2 = experiment: its
                                    send part as the send launch, then the retire launch.  EXPERIMENTAL: refused (PCC_EINVAL) unless the
                                    device reports 8 XCDs in one partition, the configuration its in-launch hand-off was validated on */,
       PCC_TUNE_FUSED_ACQUIRE = 27 /* fused step, debug: 2 = an agent-scope acquire (buffer_inv sc1) between the poll that finds an env
                                    ready and the first load of its state; 0 (default) = none: the ready queues are per physical XCD, so
                                    producer and consumer share an L2 (pcc-rl_amd/csrc/pcc_dev.h "ready queues") */,
       PCC_TUNE_FUSED_LIGHT_WGS = 28 /* fused step: workgroups per partition that start with the light items (dispatched last);
                                    default 32 */,
       PCC_TUNE_FUSED_MAX_NAPS = 29 /* fused step: a wavefront that finds no env ready looks again after 1, 2, 4, ... up to this many
                                    naps of ~0.9 us; default 4 */,
       PCC_TUNE_FUSED_PARTIAL_NAPS = 30 /* fused step: a wavefront that finds fewer envs ready than its lanes hold (8 at 8 lanes, 4 at
                                    16) takes them anyway once it has waited this many naps; default 2 */,
       PCC_TUNE_FUSED_LIGHT_FRONT = 32 /* fused step: so many of the light-first workgroups per partition are dispatched in FRONT of the
                                    wave-path workgroups (a compute unit's memory pipeline serves its oldest wavefronts first); default 0 */,
       PCC_TUNE_NOISE_SORTED = 33 /* USE_LATENCY_NOISE alone on one sender: 1 (default) = an interval is run by a wavefront per env as counts, two
                                    sorts and a scan (pcc-rl_amd/csrc/pcc_noise_sorted.hip) and the event loop takes only the envs whose events in
                                    flight do not fit its arrays; 2 = only its 256-event instance (the event loop takes the rest: what the
                                    tests use to cross the two); 0 = the event loop for every env.  Results do not depend on it. */,
       PCC_TUNE_LIGHT_WGS = 34 /* send launch: light workgroups (4 wavefronts, an item each per round) per partition; their wavefronts take
                                    further items when there are more items than wavefronts.  0 (default) = the worst case, an item per
                                    wavefront.  32 = what stays resident next to 12 wave-path wavefronts per compute unit: measured no
                                    faster (profiles/r06_knob_sweeps.json) */,
       PCC_TUNE_LIGHT_FRONT = 35 /* send launch: so many light workgroups per partition -- the ones with the longest lane-round items --
                                    are dispatched in FRONT of the wave-path workgroups (block order); the rest behind them as before.
                                    The launch ends with its longest lane-round items: in front they start ~5 us earlier (the
                                    dispatcher places the wave-path workgroups first otherwise).  Default 6 (send launch 0.0893 ->
                                    0.0864 ms at 65 536 envs); not applied to launches that carry restart items (out of lockstep: measured
                                    slower).  Speed only */,
       PCC_TUNE_FUSED_DEBUG = 31 /* fused step, experiments: bit 0 (1) = an agent-scope release (buffer_wbl2) in front of every publication,
                                    bit 2 (4) = no retire work before every env is sent (the halves one after the other inside the launch);
                                    default 0 */ };
int pcc_set_tuning(pcc_sim_t *sim, int key, double value);

/* How many steps of this handle ran as one launch so far (PCC_TUNE_FUSED; the others ran as a send and a retire launch):
 * what the tests and the bench check to know which path they measured.  No reference counterpart. */
int pcc_fused_steps(pcc_sim_t *sim, uint64_t *out);

/* Diagnostics: the device addresses of the handle's allocations -- state blob, ring tiers 0..3, work lists, shadow rings, history
 * (0 = not allocated).  Where the rings land in the address space moves the retire launch by ~10 % from one handle to the next
 * (profiles/r05_placement.json).  No reference counterpart. */
int pcc_debug_addresses(pcc_sim_t *sim, uint64_t *out8);

/* Sizes of the shared ring pools (see pcc_create): tiers 1, 2, 3 get a slot for one sender in div1, div2, div3 (defaults
 * 2, 8, 32, measured on U(-1, 1) policies; 1 = a slot for every sender -- what a policy that drives every env to its
 * rate limit can need: ~100 GB for 65 536 envs at the default ring_capacity).  Reallocates the pools and synchronizes the
 * device; whatever was in flight is dropped, so pcc_reset must follow.  No reference counterpart (its heap grows). */
int pcc_set_ring_pools(pcc_sim_t *sim, uint32_t div1, uint32_t div2, uint32_t div3);

/* The reference's dormant engine option USE_CWND (ns:54; off in the reference): window-limited
 * sending.  A SEND event launches a packet only while fewer than cwnd packets are unacknowledged
 * (ns:251-255; acknowledgements and loss reports arrive one RTT after the send, ns:264-273); a
 * blocked SEND still passes through the link's queue and loss draw, as in the reference
 * (ns:158-175).  Every new episode starts with cwnd = 25 (ns:209).  With the option on, the actions
 * of pcc_step / pcc_step_send are [N][S][2] = (rate action, cwnd action) per sender (ns:376-377, 412-414):
 * the second moves the sender's window like the first moves its rate (x (1 + a*delta_scale) or / (1 - a*
 * delta_scale)), truncated to an integer and clamped to [4, 5000] (ns:33-34, 283-289).  pcc_reset must
 * follow.  One sender: a lane-serial send path (no wave path).  Two senders: the windows couple both SEND
 * streams to the notifications, so the interval runs in the event-loop build (see pcc_set_latency_noise:
 * one lane per env, one launch per interval, no pcc_step_send / pcc_step_retire split; golden sets
 * two_sender_cwnd*).  Exact like the other paths.  It combines with pcc_set_latency_noise (see there). */
int pcc_set_cwnd_mode(pcc_sim_t *sim, int enable);

/* The reference's other dormant engine option, USE_LATENCY_NOISE / MAX_LATENCY_NOISE (ns:51-52; off /
 * 1.1 in the reference): every link latency -- forward hop at the SEND (ns:171-172), return hop at the
 * first ACK event (ns:150-151) -- is multiplied by random.uniform(1.0, max_noise), one more draw of the
 * env's stream per hop (at a SEND it precedes the loss draw).  Packets overtake each other, so with the
 * option on an env keeps the reference's own structure -- a heap of each sender's events, ring_capacity
 * events per sender, allocated by this call (2 x 16 bytes x ring_capacity per sender) -- and one lane runs the
 * reference's event loop over it ("event-loop build"); there is no pcc_step_send / pcc_step_retire split.  With ONE sender and
 * no window the interval itself runs ahead of that launch without the event loop (PCC_TUNE_NOISE_SORTED,
 * pcc-rl_amd/csrc/pcc_noise_sorted.hip: counts, two sorts and a scan by a workgroup per env; ~10x the event loop's speed).
 * Exact like the other paths either way (golden sets noise_*, two_sender_noise).  Uniforms: PCC_RNG_TRACE replays
 * the trace in draw order (three draws per packet); PCC_RNG_PHILOX numbers ALL draws of an interval
 * 0, 1, 2, ... in event order (word index of the interval's Philox stream).  One or two senders per env (events of
 * equal time: lower sender id first, ns:42-43).  Together with pcc_set_cwnd_mode (the reference's two flags are module
 * globals and apply together): a SEND goes out only while fewer than cwnd of the sender's events are in its heap, a
 * blocked one still takes its noise draw and its loss draw, actions are [N][S][2].  pcc_reset must follow.  More events in flight than ring_capacity, or
 * more acknowledgements in one interval, raise PCC_FLAG_RING_OVERFLOW. */
int pcc_set_latency_noise(pcc_sim_t *sim, int enable, double max_noise);

/* DELTA_SCALE (src/common/config.py:17, default 0.025) and MAX_STEPS (ns:41, default 400) */
int pcc_set_delta_scale(pcc_sim_t *sim, double delta_scale);
int pcc_set_max_steps(pcc_sim_t *sim, int max_steps);

/*
 * reset() for the envs whose mask byte is non-zero (mask NULL = all): ns:469-484 -- new
 * parameters, fresh sender + history, the first SEND at 1/rate, two unrecorded warm-up
 * monitor intervals of 3*lat.  obs_out (device, f32 [N][S][H*F], may be NULL) receives the
 * all-empty history observation for the envs that were reset; other rows are untouched.
 */
int pcc_reset(pcc_sim_t *sim, const uint8_t *mask, float *obs_out, void *stream);

/*
 * step() for every env: ns:406-444 -- apply the rate action (ns:235-241, clamp ns:275-281),
 * run one monitor interval (Network.run_for_dur, ns:123-205), evaluate the MI metrics and
 * roll the history (so:44-73), reward (ns:194,205), next run_dur (ns:437-438), done (ns:444).
 *   actions      device, [N][S], f32 (actions_f64 = 0) or f64 (actions_f64 = 1).
 *   obs_out      device f32 [N][S][H*F], oldest MI first (ns:400-404).          may be NULL
 *   reward_out   device f32 [N][S].                                              may be NULL
 *   done_out     device u8  [N].                                                 may be NULL
 *   steps_out    device f64 [N][S][PCC_STEP_COLS]: counters, clocks, reward, 12 metrics at
 *                full precision (the reference's per-step event record ns:422-436 is a
 *                subset).                                                         may be NULL
 *   auto_reset   non-zero: envs that finished are reset in the same call and their obs_out
 *                rows hold the first observation of the next episode.
 * Returns PCC_ESTATE if pcc_reset was never called on this handle.
 */
int pcc_step(pcc_sim_t *sim, const void *actions, int actions_f64, float *obs_out, float *reward_out,
             uint8_t *done_out, double *steps_out, int auto_reset, void *stream);

/* n_steps steps with pre-computed actions in one call: actions [n_steps][N][S] (with the congestion-window option
 * [n_steps][N][S][2]); obs_out [n_steps][N][S][H*F], reward_out [n_steps][N][S], done_out [n_steps][N], steps_out
 * [n_steps][N][S][PCC_STEP_COLS], any of them NULL.  Results are those of n_steps pcc_step calls.  A small batch (fewer
 * envs than LIST_MIN_ENVS) whose episode boundaries the host knows runs the steps up to the next boundary inside ONE
 * launch -- the loop over the steps is on the device, a workgroup per 64 envs, no launch boundary between steps (config 2:
 * a step is a chain of dependent loads, and a launch boundary is a third of it); other batches are stepped launch by launch
 * from C.  For open-loop drivers: a policy in the loop needs the observation of every step and calls pcc_step.  On an error
 * the message says after how many steps the call stopped.  The reference has no counterpart (its step() is one Python call
 * per interval, ns:406). */
int pcc_step_many(pcc_sim_t *sim, const void *actions, int actions_f64, int n_steps, float *obs_out, float *reward_out,
                  uint8_t *done_out, double *steps_out, int auto_reset, void *stream);

/*
 * The two halves of pcc_step as separate calls (pcc_step = pcc_step_send + pcc_step_retire with the
 * same arguments): `send` applies the actions and transmits every packet of the coming monitor
 * interval (the SEND events of ns:155-178); `retire` processes acknowledgements and losses, closes
 * the interval and writes the outputs.  Lets a caller overlap its own work with the first half or
 * time the halves separately (bench.py does).  Each send must be followed by exactly one retire.
 */
int pcc_step_send(pcc_sim_t *sim, const void *actions, int actions_f64, void *stream);
int pcc_step_retire(pcc_sim_t *sim, float *obs_out, float *reward_out, uint8_t *done_out,
                    double *steps_out, int auto_reset, void *stream);

/* How the episodes of envs that finished out of lockstep (auto_reset, after a masked reset) were started, since creation:
 * out2[0] by swapping in the env's shadow -- its next episode (new links, ns:469-477, and the two warm-up intervals,
 * ns:478-479) prepared ahead of time on a side stream -- out2[1] through the restart list (the warm-up intervals run in
 * front of the env's first interval inside the send half).  Host pointer; synchronizes `stream`.  No reference counterpart. */
int pcc_restart_stats(pcc_sim_t *sim, uint64_t *out2, void *stream);

/* copy one state field into a caller-owned device buffer (see the PCC_F_* table) */
int pcc_get_state(pcc_sim_t *sim, int field, void *out, void *stream);

/* bounds and scale of metric `id`: (min_val, max_val, scale) of so:193-206; host pointers.
 * get_min_obs_vector / get_max_obs_vector (so:95-108) are these tiled H times. */
int pcc_metric_info(int id, double *min_val, double *max_val, double *scale);

/* bytes of device memory the handle owns */
int64_t pcc_device_bytes(const pcc_sim_t *sim);

/* Profiling aid, no reference counterpart.  When the handle was created with the environment
 * variable PCC_DEBUG_TIMELINE=1, every work item of the LAST send launch leaves 8 words: start, end
 * of its lane rounds, end (100 MHz device ticks), envs it sent with the wave path, packets it
 * sent, packets of its largest env, packets sent by the wave path, live lanes -- item t at word
 * 8 t, items in the order they were handed out (heaviest first).  After the items come 16 words per
 * retire workgroup (word 0 / 1: start / end of the workgroup in the last launch; words 3..11: time by
 * phase, summed over its wavefronts and over the launches so far).  Synchronizes the device.  Returns
 * the number of words -- copied, or needed when out is NULL; 0 when the timeline is off, < 0 on
 * error. */
int64_t pcc_debug_timeline(pcc_sim_t *sim, uint64_t *out, int64_t n_words);

/* Profiling only (handles created with PCC_DEBUG_TIMELINE=1): counters of the send half's wave passes
 * since creation or the last reset of the counters -- out16[0..3] passes by regime (A "always empty",
 * B with the token scan, B without it, serial), [4..7] packets they carried, [8] B passes that could
 * not commit a packet, [9] passes refused for the send-time preconditions, [10] for the queue
 * preconditions, [11] envs sent as heavy items, [12] envs taken over from lane rounds, [13] / [14]
 * shader cycles spent in committed / serial passes, [15] work items of the last send launch.
 * Synchronizes the device.  No reference counterpart. */
int pcc_debug_pass_stats(pcc_sim_t *sim, uint64_t *out16, int reset);

#ifdef __cplusplus
}
#endif
#endif /* PCC_SIM_H */
