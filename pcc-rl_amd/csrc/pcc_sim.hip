// pcc_sim.hip -- MI355X (gfx950) batched congestion-control simulator: the C ABI (include/pcc_sim.h) and the launch
// logic.  The kernels live in their own translation units (pcc_send.hip, pcc_send_restart.hip,
// pcc_retire.hip, pcc_small.hip; pcc_kernels.h declares their launch functions, pcc_dev.h what they share).
//
// What this replaces (reference = PCCproject/PCC-RL; "ns" = src/gym/network_sim.py, "so" =
// src/common/sender_obs.py): SimulatedNetworkEnv.reset / step (ns:406-484) for N independent envs advanced one monitor
// interval per step -- see pcc_dev.h for the formulation.
#include "pcc_kernels.h"

using namespace pcc;

namespace {

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
    void *noise_out_blob;  // ... and the per-env results of the heap-free interval (pcc_noise_sorted.hip)
    uint32_t light_wgs; // PCC_TUNE_LIGHT_WGS: light workgroups per partition of the send launch (0 = what stays resident)
    uint32_t light_front; // PCC_TUNE_LIGHT_FRONT: ... of which so many per partition are dispatched in front of the wave-path workgroups
    int noise_sorted;   // PCC_TUNE_NOISE_SORTED: 1 = latency noise alone on one sender runs its intervals by sorting, 2 = only the small instance, 0 = the event loop
    size_t noise_bytes;
    uint32_t ring_capacity;
    bool pools_pending;     // the pools of tiers >= 1 are not allocated yet (they are sized at the first reset, or by pcc_set_ring_pools)
    bool restarts_pending;  // a retire launch may have left envs in the restart list (their warm-up intervals are due)
    bool read_has_restarts; // the list buffer read_buf was filed by a retire launch that resets finished envs (restart list)
    uint32_t list_min_envs; // batches below this size are stepped without work lists (index order)
    // side streams of the handle: the restart kernel (or, without shadows, the main send launch beside it) and the refill
    // kernel, forked from / joined to the caller's stream by events
    hipStream_t aux_wave, aux_restart;
    hipEvent_t ev_fork, ev_wave, ev_restart;
    int restart_fork;         // tuning: with shadows, the restart kernel beside the main send launch (side stream) or behind it
    double retire_grid_frac;  // tuning: share of the envs the retire grid expects in the wide classes (see launch_retire_half)
    uint32_t step_seq;      // sequence number of the last step (Dev::step_seq of its launches)
    void *shadow_blob;      // the shadows' private rings (allocated when envs first restart out of lockstep)
    size_t shadow_bytes;
    hipEvent_t ev_ret, ev_refill[4];
    bool refill_recorded[4];
    hipStream_t last_stream; // the stream of the last pcc_reset / pcc_step (what pcc_set_tuning's flush is queued on)
    // the fused step (pcc_fused.hip): on / off (tuning), the list buffer known to be clean (-1: none), workgroups of the
    // kernel a compute unit holds at once (per rng mode; 0 = not asked yet, -1 = unknown), light-first workgroups per partition
    int fused;
    int xcc_count;      // XCDs of the device (hipDeviceAttributeNumberOfXccs): the fused step's per-XCD queues are built for 8
    int clean_buf;
    int fused_blocks[2];
    uint32_t fused_light_wgs;
    uint32_t fused_light_front;   // ... of which so many per partition are dispatched in FRONT of the wave-path workgroups
    unsigned long long fused_steps;  // steps that ran as one launch (pcc_debug_fused_steps)
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
    d.env = c.take<EnvBlk>(2 * n);        // (env blocks and their shadows: pcc_dev.h)
    d.snd = c.take<SndBlk>(2 * sn);
    d.stride = (int64_t)(2 * n);
    d.refill_count = c.take<uint32_t>(4 * kCntStride);
    d.refill_list = c.take<uint32_t>(4 * n);
    d.restart_stats = c.take<unsigned long long>(2);
    d.cls_count = c.take<uint32_t>(kListBufs * kParts * kClsStride);
    d.cursors = c.take<uint32_t>(kListBufs * kParts * kShards * kCursorStride);
    d.fctl = c.take<uint32_t>(kListBufs * kXcds * kFctlWords * kCursorStride);
    // the fused step's ready queues: two per XCD, each able to hold every env (who retires an env is decided by where its send
    // half ran); batches beyond 4 M envs go without (and are stepped by two launches)
    d.q_cap = n <= ((size_t)1 << 22) ? (uint32_t)n : 0u;
    d.q_entries = c.take<unsigned long long>((size_t)2 * kXcds * d.q_cap);
    d.any_done = c.take<uint32_t>(1);
    d.tier_top = c.take<int32_t>(kMaxTiers * kParts * kTopStride);
    d.pool_share = c.take<uint32_t>(kMaxTiers);
    d.pool_free_stride = sn < 256 ? 256 : sn;   // (a pool has at most a slot per sender, at least 256: alloc_tier)
    d.pool_free = c.take<uint32_t>(kMaxTiers * d.pool_free_stride);
    d.hist = c.take<float>(sn * d.HF);
    return (c.off + 255) & ~(size_t)255;
}

int check_hip(hipError_t err, const char *what) {
    if (err == hipSuccess) return PCC_OK;
    return fail(PCC_EHIP, "%s: %s", what, hipGetErrorString(err));
}

// SEND half of one monitor interval: all envs (warm = 0) or the envs being reset (warm = 1).
// With work lists (the lists the last retire launch filed, sim->read_buf): ONE launch of send_kernel holds the light
// workgroups (the classes below the wave-path threshold, a lane per env) and the wave-path workgroups (the classes from the
// threshold up, a wavefront or a workgroup per env), in the dispatch order pcc_send.hip explains.  Envs the last retire
// launch reset out of lockstep (the restart list: warm-up intervals first) are a kernel of their own on a side stream of the
// handle, forked from and joined to the caller's stream by events; it never touches an env the main launch touches.
// Without lists (after a reset, warm-up intervals, small batches): light workgroups only, the envs in index order.
int launch_send(pcc_sim_t *sim, int warm, uint32_t warm_mi, int gate, const void *actions, int actions_f64, hipStream_t st) {
    const Dev &d = sim->d;
    const bool tr = d.rng_mode == PCC_RNG_TRACE;
    // small batches go without work lists (items = the envs in index order): at 4 096 envs of a few packets each the
    // launch IS its chain of dependent loads, and the lists put three more in front (counts -> list -> state)
    const bool lists = sim->d.n >= (int64_t)sim->list_min_envs;
    const int read_buf = (warm || !lists) ? -1 : sim->read_buf;
    const int zero_buf = lists ? sim->fill_buf : -1;
    // light workgroups: one item per wavefront; the grid covers the worst case (every env light: n / E items + a partial
    // one per class), wavefronts without an item leave at once.  With lists every partition has its own share of the
    // workgroups (pcc_dev.h "partitions"): both kinds' counts are multiples of d.parts
    const int64_t E = d.send_envs_per_wave;
    if (read_buf < 0) {
        const unsigned light_grid = (unsigned)(((d.n + E - 1) / E + 3) / 4);
        pcc::launch_send(d, tr, light_grid, 0u, 0u, st, read_buf, zero_buf, warm, warm_mi, gate, actions, actions_f64);
        return check_hip(hipGetLastError(), "send kernel launch");
    }
    const int64_t P = d.parts;
    // (the classes from light_half_predict up go E / 2 envs to an item: the worst case is all of them)
    const int64_t E_min = d.light_half_predict < 1e9f && E >= 2 ? E / 2 : E;
    const int64_t light_items_part = ((int64_t)d.part_envs + E_min - 1) / E_min + kClasses;
    const unsigned worst_light_grid = (unsigned)(P * ((light_items_part + 3) / 4));
    // restart items can only be in lists that a retire launch with `restart` filed
    const bool rs = sim->read_has_restarts;
    const bool wave = !d.use_cwnd && d.heavy_predict < 1e9;  // (USE_CWND sends every env lane-serially: no wave-path classes)
    // wave-path workgroups: persistent wavefronts, send_waves per compute unit, at most one wavefront per env
    int64_t waves = (int64_t)sim->cu_count * d.send_waves;
    if (waves > d.n) waves = d.n;
    const unsigned wave_grid = wave ? (unsigned)(P * (((waves + 3) / 4 + P - 1) / P)) : 0u;
    // Light workgroups: an item per wavefront in the worst case.  (Round 6 measured the alternative -- only as many as stay
    // resident next to the wave-path workgroups, their wavefronts taking a second item, PCC_TUNE_LIGHT_WGS: the ones that wait
    // for a slot start 40-60 us into the launch -- and it is no faster: 0.0974 against 0.0961 ms, settings alternating every
    // step on one handle, profiles/r06_knob_sweeps.json.)
    unsigned light_grid = worst_light_grid;
    if (wave && sim->light_wgs && (int64_t)sim->light_wgs * P < (int64_t)light_grid) light_grid = (unsigned)((int64_t)sim->light_wgs * P);
    // PCC_TUNE_LIGHT_FRONT: the light workgroups with the longest items in front of the wave-path ones (block order = dispatch
    // order): the launch ends with those items and the ~830 wave-path workgroups take the dispatcher ~5 us -- send launch
    // 0.0893 -> 0.0864 ms at 6 per partition (tools/ab_block.py, three episodes a setting).  Not out of lockstep: measured slower
    // there (0.132 -> 0.137-0.140 ms, bench.py --stagger)
    unsigned front = (wave && !rs) ? (unsigned)((int64_t)sim->light_front * P) : 0u;
    if (front > light_grid) front = light_grid / (unsigned)P * (unsigned)P;
    const unsigned restart_grid = (unsigned)(sim->cu_count < (d.n + 3) / 4 ? sim->cu_count : (d.n + 3) / 4);
    if (rs && d.shadows) {
        // with shadows nearly every restart is a swap inside the retire half: the restart list holds only the envs whose shadow
        // was not usable (a masked reset overtook it; links whose warm-up intervals overflow a shadow's rings) -- nearly always
        // nobody: the kernel follows the main launch on the caller's stream, no fork, no join
        if (sim->restart_fork) {   // ... or beside it on the side stream: it is done long before the main launch, so the join is free
            if (hipEventRecord(sim->ev_fork, st) != hipSuccess) return fail(PCC_EHIP, "hipEventRecord failed");
            (void)hipStreamWaitEvent(sim->aux_wave, sim->ev_fork, 0);
            launch_send_restart(d, tr, restart_grid, sim->aux_wave, read_buf, actions, actions_f64);
            (void)hipEventRecord(sim->ev_wave, sim->aux_wave);
            pcc::launch_send(d, tr, light_grid, wave_grid, front, st, read_buf, zero_buf, warm, warm_mi, gate, actions, actions_f64);
            (void)hipStreamWaitEvent(st, sim->ev_wave, 0);
        } else {
            pcc::launch_send(d, tr, light_grid, wave_grid, front, st, read_buf, zero_buf, warm, warm_mi, gate, actions, actions_f64);
            launch_send_restart(d, tr, restart_grid < 32u ? restart_grid : 32u, st, read_buf, actions, actions_f64);  // (a handful of items at most)
        }
    } else if (rs) {
        // The restart items are a chain of dependent passes (reset, two warm-up intervals, the first interval): longer than
        // the whole main launch.  So the MAIN launch goes to the side stream and the restart kernel stays on the caller's:
        // what the next launch of the caller's stream waits for across streams has then long finished (a cross-stream
        // dependency on a kernel that is just ending costs ~15 us, one that ended long ago next to nothing)
        if (hipEventRecord(sim->ev_fork, st) != hipSuccess) return fail(PCC_EHIP, "hipEventRecord failed");
        (void)hipStreamWaitEvent(sim->aux_wave, sim->ev_fork, 0);
        pcc::launch_send(d, tr, light_grid, wave_grid, front, sim->aux_wave, read_buf, zero_buf, warm, warm_mi, gate, actions, actions_f64);
        (void)hipEventRecord(sim->ev_wave, sim->aux_wave);
        launch_send_restart(d, tr, restart_grid, st, read_buf, actions, actions_f64);
        (void)hipStreamWaitEvent(st, sim->ev_wave, 0);
    } else {
        pcc::launch_send(d, tr, light_grid, wave_grid, front, st, read_buf, zero_buf, warm, warm_mi, gate, actions, actions_f64);
    }
    if (rs && !warm) sim->restarts_pending = false;  // this launch runs what the restart list's envs were owed
    return check_hip(hipGetLastError(), "send kernel launch");
}

// RETIRE half; a launch that is not a warm-up interval also files every env in the work lists of the
// next send (buffer sim->fill_buf, cleared by the send launch before it)
// restart: envs that finish their episode in this launch are reset inside it and filed in the restart list
int launch_retire_half(pcc_sim_t *sim, int warm, uint32_t warm_mi, int last_warm, int gate, int restart, float *obs_out,
                       float *reward_out, uint8_t *done_out, double *steps_out, hipStream_t st) {
    const Dev &d = sim->d;
    // workgroups: 8 envs each at 16 lanes per env, 16 at 8 lanes -- which envs go which way is decided on the device
    // (class counts), so the grid covers the worst case plus the one workgroup the split can leave partly filled
    const bool lists = d.n >= (int64_t)sim->list_min_envs;
    const int read = (warm || !d.retire_sorted || !lists) ? -1 : sim->read_buf;  // the lists this step's send launch read
    // With lists the walk is n / 16 workgroups plus one more for every 16 envs of the wide classes (8 per workgroup there): the
    // grid is sized for retire_grid_frac of the envs being wide (default 1/8: about 3 % are) and the workgroups loop if there
    // are more -- the worst case, twice n / 16, is ~8 200 workgroups for 65 536 envs, which the command processor needs
    // 0.1 ms to dispatch even when half of them find nothing to do
    // (with lists: that many for every partition's share, pcc_dev.h "partitions")
    const int64_t narrow = (d.n + kRetireEnvsPerBlockNarrow - 1) / kRetireEnvsPerBlockNarrow;
    const int64_t narrow_part = ((int64_t)d.part_envs + kRetireEnvsPerBlockNarrow - 1) / kRetireEnvsPerBlockNarrow;
    const unsigned grid = (unsigned)(read >= 0 ? (int64_t)d.parts * (narrow_part + (int64_t)(sim->retire_grid_frac * (double)narrow_part) + 1) : narrow);
    const int fill = (warm || !lists) ? -1 : sim->fill_buf;
    launch_retire(d, false, grid, st, read, fill, warm, warm_mi, last_warm, gate, restart, obs_out, reward_out, done_out, steps_out,
                  nullptr, 0);
    const int rc = check_hip(hipGetLastError(), "retire kernel launch");
    if (rc == PCC_OK && !warm && lists) {
        sim->read_buf = sim->fill_buf;
        sim->fill_buf = (sim->fill_buf + 1) % kListBufs;
        sim->clean_buf = -1;   // (the send launch cleared the buffer this launch filed; the next one is as its last reader left it)
        sim->read_has_restarts = restart != 0;  // the buffer just filed may hold a restart list
        if (restart) sim->restarts_pending = true;
    }
    return rc;
}

// Both halves of a step as ONE launch (pcc_fused.hip) -- whenever the step is an ordinary one: work lists to read, no warm-up
// interval, no restart list to serve or to fill (lockstep, or the caller resets), none of the engine options.  Returns
// PCC_OK + *done = true when the step was launched; *done = false: the caller launches the two halves.
int try_launch_fused(pcc_sim_t *sim, int restart, const void *actions, int actions_f64, float *obs_out, float *reward_out,
                     uint8_t *done_out, double *steps_out, hipStream_t st, bool *done) {
    const Dev &d = sim->d;
    *done = false;
    if (!sim->fused || d.engine || d.use_cwnd || restart || sim->read_has_restarts || sim->restarts_pending) return PCC_OK;
    if (d.n < (int64_t)sim->list_min_envs || sim->read_buf < 0 || !d.retire_sorted) return PCC_OK;
    if (d.q_cap == 0u) return PCC_OK;
    const bool tr = d.rng_mode == PCC_RNG_TRACE;
    int &blocks = sim->fused_blocks[tr ? 1 : 0];
    if (blocks == 0) { blocks = fused_resident_blocks(d.ns, tr); if (blocks <= 0) blocks = -1; }
    if (blocks < 0) return PCC_OK;
    // grid: what the device holds at once (every item is claimed dynamically, so a grid that is NOT all resident is only
    // slower: its late workgroups find nothing left), a multiple of the partitions: the light-first workgroups last
    const int64_t P = d.parts;
    int64_t cap = (int64_t)blocks * sim->cu_count / P * P;
    int64_t light = P * (int64_t)sim->fused_light_wgs;
    if (cap < 2 * P) cap = 2 * P;
    if (light > cap / 2) light = cap / 2 / P * P;
    if (light < P) light = P;
    const bool wave = d.heavy_predict < 1e9;
    int64_t waves = (int64_t)sim->cu_count * d.send_waves;
    if (waves > d.n) waves = d.n;
    int64_t wave_wgs = wave ? P * (((waves + 3) / 4 + P - 1) / P) : 0;
    if (wave_wgs > cap - light) wave_wgs = cap - light;
    // (a small batch: no more workgroups than a wavefront per 8 envs and per light item need)
    int64_t light_need = P * (((int64_t)d.part_envs / (int64_t)(d.send_envs_per_wave ? d.send_envs_per_wave : 1) + kClasses + 3) / 4);
    int64_t grid = cap;
    const int64_t by_envs = P * (((int64_t)d.part_envs / 8 + 3) / 4 + 1);
    if (grid > wave_wgs + (light_need > by_envs ? light_need : by_envs)) grid = wave_wgs + (light_need > by_envs ? light_need : by_envs);
    if (grid < wave_wgs + P) grid = wave_wgs + P;
    const int read = sim->read_buf, fill = sim->fill_buf, zero = (sim->fill_buf + 1) % kListBufs;
    // light-first workgroups in front of the wave-path ones (a multiple of the partitions; the others behind them)
    unsigned light_front = (unsigned)(P * (int64_t)sim->fused_light_front);
    if ((int64_t)light_front > grid - wave_wgs) light_front = (unsigned)((grid - wave_wgs) / P * P);
    if (sim->fused == 2) {   // experiment: the fused kernel's send part as the send launch, then the retire launch
        launch_step_fused(d, tr, (unsigned)grid, (unsigned)wave_wgs, light_front, st, read, fill, fill, 0, actions, actions_f64, obs_out, reward_out, done_out,
                          steps_out);
        const int rc2 = check_hip(hipGetLastError(), "fused step kernel launch");
        if (rc2 != PCC_OK) return rc2;
        *done = true;
        return launch_retire_half(sim, 0, 0, 0, 0, 0, obs_out, reward_out, done_out, steps_out, st);
    }
    if (sim->clean_buf != fill) launch_clear_list_buffer(d, st, fill);
    launch_step_fused(d, tr, (unsigned)grid, (unsigned)wave_wgs, light_front, st, read, fill, zero, 1, actions, actions_f64, obs_out, reward_out, done_out,
                      steps_out);
    const int rc = check_hip(hipGetLastError(), "fused step kernel launch");
    if (rc != PCC_OK) return rc;
    sim->read_buf = fill;
    sim->fill_buf = zero;
    sim->clean_buf = zero;
    sim->read_has_restarts = false;
    sim->fused_steps++;
    *done = true;
    return PCC_OK;
}

int launch_mi(pcc_sim_t *sim, int warm, uint32_t warm_mi, int last_warm, int gate, int restart, const void *actions,
              int actions_f64, float *obs_out, float *reward_out, uint8_t *done_out, double *steps_out, hipStream_t st) {
    if (sim->d.engine) {
        // the event-loop build (latency noise; the congestion window with two senders): the whole interval is one launch of
        // the retire kernel's NOISE build (no send half, no work lists)
        // Latency noise alone on one sender: the interval itself is run ahead of that launch, a wavefront per env, without
        // the event loop (pcc_noise_sorted.hip); the retire launch picks the results up env by env (NoiseOut::seq) and
        // runs the event loop for the envs that were left alone.
        sim->d.noise_seq++;
        Dev d = sim->d;
        const bool sorted = sim->noise_sorted && d.use_noise && !d.use_cwnd && d.noise_out != nullptr;
        if (!sorted) d.noise_out = nullptr;
        else launch_noise_sorted(d, st, warm, warm_mi, gate, actions, actions_f64, sim->noise_sorted == 2);
        const unsigned grid = (unsigned)((d.n + kRetireEnvsPerBlockNarrow - 1) / kRetireEnvsPerBlockNarrow);
        launch_retire(d, true, grid, st, -1, -1, warm, warm_mi, last_warm, gate, 0, obs_out, reward_out, done_out, steps_out,
                      actions, actions_f64);
        return check_hip(hipGetLastError(), "event-loop kernel launch");
    }
    if (!warm && !gate) {
        bool done = false;
        const int rf = try_launch_fused(sim, restart, actions, actions_f64, obs_out, reward_out, done_out, steps_out, st, &done);
        if (rf != PCC_OK || done) return rf;
    }
    const int rc = launch_send(sim, warm, warm_mi, gate, actions, actions_f64, st);
    if (rc != PCC_OK) return rc;
    return launch_retire_half(sim, warm, warm_mi, last_warm, gate, restart, obs_out, reward_out, done_out, steps_out, st);
}

// reset(): parameters + state, then the two unrecorded warm-up MIs (ns:469-484).  gate: the launches
// do nothing unless the retire half flagged a finished env in this step (auto-reset of envs that are
// not in lockstep)
int launch_reset(pcc_sim_t *sim, const uint8_t *mask, int use_done, int gate, float *obs_out, hipStream_t st) {
    const Dev &d = sim->d;
    // every env of the batch is reset: a full reset, or the episode boundary of a batch in lockstep (every env is done)
    const int all_envs = (!mask && !gate && (use_done == 0 || (use_done == 1 && sim->lockstep))) ? 1 : 0;
    launch_reset_init(d, st, mask, use_done, gate, all_envs, obs_out);
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

// Shadows (the next episode of an env prepared ahead of time, swapped in when it finishes: pcc_dev.h, pcc_send_restart.hip)
// need uniforms that do not depend on the order of events across episodes (Philox, not a replayed trace) and their private
// rings (tier-1 size per sender: allocated the first time envs restart out of lockstep -- 6.4 GB for 65 536 senders).
bool use_shadows(pcc_sim_t *sim) {
    Dev &d = sim->d;
    if (d.rng_mode != PCC_RNG_PHILOX) { d.shadows = 0; return false; }
    if (!sim->shadow_blob) {
        const size_t bytes = (size_t)d.n * d.ns * 3 * ((size_t)d.cap0 << (2 * kShadowTier)) * sizeof(double2);
        if (d.n_tiers < 2 || hipMalloc(&sim->shadow_blob, bytes) != hipSuccess) { sim->shadow_blob = nullptr; d.shadows = 0; return false; }
        (void)hipMemset(sim->shadow_blob, 0, bytes);   // (first use of fresh device memory is slow: not in somebody's step)
        sim->shadow_bytes = bytes;
        d.shadow_rings = static_cast<char *>(sim->shadow_blob);
    }
    d.shadows = 1;
    return true;
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
    sim->tier_bytes[c] = total;
    sim->ring_bytes += total;
    return PCC_OK;
}

// Every pool's free stacks, full: one stack per partition of the batch over its contiguous share of the pool's slots
// (pcc_dev.h "partitions"), the share's lowest slot on top.  (Slots beyond parts * share, fewer than `parts`, stay unused.)
int init_pool_stacks(pcc_sim_t *sim) {
    Dev &d = sim->d;
    std::vector<int32_t> tops((size_t)kMaxTiers * kParts * kTopStride, 0);
    uint32_t shares[kMaxTiers] = {0, 0, 0, 0};
    for (int c = 1; c < d.n_tiers; c++) {
        const uint32_t share = sim->tier_slots[c] / d.parts;
        shares[c] = share;
        std::vector<uint32_t> ids(sim->tier_slots[c], 0u);
        for (uint32_t p = 0; p < d.parts; p++) {
            for (uint32_t j = 0; j < share; j++) ids[(size_t)p * share + j] = p * share + (share - 1u - j);
            tops[(size_t)(c * kParts + p) * kTopStride] = (int32_t)share;
        }
        if (hipMemcpy(d.pool_free + (size_t)c * d.pool_free_stride, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
            return fail(PCC_EHIP, "filling the tier-%d free stacks failed", c);
    }
    if (hipMemcpy(d.tier_top, tops.data(), tops.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d.pool_share, shares, sizeof shares, hipMemcpyHostToDevice) != hipSuccess)
        return fail(PCC_EHIP, "resetting the pool stacks failed");
    return PCC_OK;
}

// The batch as `parts` partitions (1 or kParts) of part_envs consecutive env ids each (a multiple of 64: an index-order item
// or retire workgroup never straddles two).
void set_parts(pcc_sim_t *sim, uint32_t parts) {
    Dev &d = sim->d;
    d.parts = parts;
    d.parts_shift = 0;
    while ((1u << d.parts_shift) < parts) d.parts_shift++;
    const int64_t per = (d.n + parts - 1) / parts;
    d.part_envs = (uint32_t)((per + 63) / 64 * 64);
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
    d.send_waves = 13;  // persistent wavefronts of the wave path per compute unit (beside them: the light items).  12 until round 6; re-swept with
                        // the settings alternating every step (tools/ab_step.py, profiles/r06_knob_sweeps.json): 12 / 13 / 14 / 15 = 0.0915 / 0.0906 / 0.0924 / 0.115 ms
    d.send_envs_per_wave = 64;
    // (two senders: the lane rounds cost about the same per packet as with one, the wave passes more -- 64 positions per pass)
    // (measured on one handle each, profiles/r04_experiments.json -- heavy_predict / heavy_item_packets: config 5
    // 1024 / 2048 -> 0.252 ms per send launch, 640-768 / 768-1024 -> 0.219-0.223; config 3 512 / 2048 -> 0.0988,
    // 384 / 1024 -> 0.0942, 320 / 768 -> 0.105: the longest light items are the launch's critical path down to ~384)
    d.heavy_predict = n_senders == 2 ? 640.0 : 480.0;   // (one sender: 384 until round 6's lane rounds -- cheaper per packet, so more classes stay light: 384 / 448 / 512 / 576 = 0.0984 / 0.0915 / 0.0920 / 0.1029 ms)
    d.team_predict = 4096.0;
    d.heavy_item_packets = 1024.0f;
    // (the wide classes are cut by COUNT in the end: the grid has room for retire_grid_frac of the envs at 16 lanes, the
    // largest; measured on one handle, r04_experiments.json: threshold 1024 -> 0.097 ms, 200-512 with the cut -> 0.091)
    d.retire_wide_predict = 256.0f;
    d.light_half_predict = 1e9f;
    d.retire_sorted = 1u;
    d.light_snake = 1u;
    d.wave_oldest_first = 1u;
    d.prio_level = 0u; d.prio_light_items = 0u; d.prio_wave_items = 0u; d.prio_team = 0u;
    d.debug_skip = (kProfile && getenv("PCC_DEBUG_SKIP")) ? atoi(getenv("PCC_DEBUG_SKIP")) : 0;  // profile build only
    const double lo[5] = {100, 0.05, 0, 0.0, 0.3}, hi[5] = {500, 0.5, 8, 0.05, 1.5};  // ns:355-358,466
    memcpy(d.lo, lo, sizeof lo); memcpy(d.hi, hi, sizeof hi);
    d.rng_mode = PCC_RNG_PHILOX;
    d.params_gen = 1u;
    sim->device = device;
    sim->state_bytes = carve_state(d, nullptr);
    if (hipMalloc(&sim->state_blob, sim->state_bytes) != hipSuccess) {
        const size_t want = sim->state_bytes;
        delete sim;
        return fail(PCC_ENOMEM, "hipMalloc(%zu) for env state failed", want);
    }
    carve_state(d, static_cast<char *>(sim->state_blob));
    // Tier 0 (a slot per sender) now; the pools of tiers 1, 2, 3 when the handle is first reset (ensure_pools: sized from the
    // device memory that is free THEN) or when pcc_set_ring_pools names their sizes -- so a caller that sets them allocates
    // them once, and handles that are created side by side do not each claim a third of the same free memory before any of
    // them has allocated (round 4 allocated and zeroed the default pools, 85 GB for 65 536 envs, inside pcc_create).
    sim->ring_bytes = 0;
    {
        const int rc = alloc_tier(sim, 0, 1);
        if (rc != PCC_OK) { pcc_destroy(sim); return rc; }
    }
    sim->pools_pending = d.n_tiers > 1;
    // partitions: batches that run with work lists (every XCD then keeps to its own eighth of the rings); smaller ones and
    // whatever runs without lists gain nothing from them
    set_parts(sim, n_envs >= 8192 ? kParts : 1u);
    if (hipMemset(sim->state_blob, 0, sim->state_bytes) != hipSuccess) {
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
    // (room for either partitioning: kParts partitions of part_envs rounded up to 64, or one of n_envs)
    sim->list_bytes = (size_t)kListBufs * kListRows * ((size_t)n_envs + 64 * kParts) * sizeof(uint32_t);
    if (hipMalloc(&sim->list_blob, sim->list_bytes) != hipSuccess) {
        pcc_destroy(sim);
        return fail(PCC_ENOMEM, "hipMalloc(%zu) for the send work lists failed", sim->list_bytes);
    }
    d.cls_list = static_cast<uint32_t *>(sim->list_blob);
    sim->read_buf = -1;
    sim->fill_buf = 0;
    sim->clean_buf = -1;
    sim->noise_sorted = 1;
    sim->fused = 0;   // (measured slower than the two launches at full size: profiles/r05_fused_experiments.json)
    { int x = 0; sim->xcc_count = hipDeviceGetAttribute(&x, hipDeviceAttributeNumberOfXccs, device) == hipSuccess ? x : 0; }
    sim->light_front = 6;
    sim->fused_light_wgs = 32;   // light-first workgroups per partition (4 wavefronts each: a partition of 8 192 envs has ~105 light items)
    d.fused_acquire = 0u;
    d.fused_spin_ticks = 100000000u;   // 1 s of the 100 MHz clock
    d.fused_max_naps = 4u;
    d.fused_partial_naps = 2u;
    sim->list_min_envs = 8192;
    sim->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    sim->retire_grid_frac = 0.125;
    sim->restart_fork = 0;   // (measured, bench --stagger: behind the main launch 0.153 ms, beside it 0.166)
    if (hipStreamCreateWithFlags(&sim->aux_wave, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&sim->aux_restart, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&sim->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sim->ev_wave, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sim->ev_restart, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sim->ev_ret, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sim->ev_refill[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sim->ev_refill[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sim->ev_refill[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sim->ev_refill[3], hipEventDisableTiming) != hipSuccess) {
        pcc_destroy(sim);
        return fail(PCC_EHIP, "creating the side streams of the send half failed");
    }
    *out = sim;
    return PCC_OK;
}

int64_t pcc_debug_timeline(pcc_sim_t *sim, uint64_t *out, int64_t n_words) {
    if (!sim || !sim->timeline_blob) return 0;
    DeviceGuard guard(sim->device);
    // item slots: the light kernel's items from slot 0, the wave kernel's from slot n (team items after them), the
    // restart kernel's from slot 3n/2; unused slots are zero
    const unsigned long long items = 2ull * (unsigned long long)sim->d.n;
    if (hipDeviceSynchronize() != hipSuccess) return fail(PCC_EHIP, "reading the debug timeline failed");
    const int64_t rblocks = (sim->d.n + 7) / 8 + 1;  // (the largest retire grid)
    // ... and the retire units of the fused step (pcc_fused.hip: a running counter at word 19 n, then 4 words per unit --
    // sequence number << 32 | queue << 31 | envs, claimed, ready, done -- the last n / 2 units)
    const int64_t units = sim->d.n >= 1024 ? 32 + 2 * sim->d.n : 0;
    const int64_t total = (int64_t)items * 8 + rblocks * 16 + units;
    if (!out || n_words <= 0) return total;
    if (n_words < total) return fail(PCC_EINVAL, "pcc_debug_timeline needs room for %lld words", (long long)total);
    const char *src = static_cast<const char *>(sim->timeline_blob);
    if (hipMemcpy(out, src, (size_t)items * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(out + items * 8, src + (size_t)sim->d.n * 2 * 8 * sizeof(uint64_t), (size_t)rblocks * 16 * sizeof(uint64_t),
                  hipMemcpyDeviceToHost) != hipSuccess ||
        (units && hipMemcpy(out + items * 8 + rblocks * 16, src + (size_t)sim->d.n * 19 * sizeof(uint64_t), (size_t)units * sizeof(uint64_t),
                            hipMemcpyDeviceToHost) != hipSuccess))
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
    // (the side streams' work is always joined to the caller's stream before a call returns; drain them before they go)
    if (sim->aux_wave) { (void)hipStreamSynchronize(sim->aux_wave); (void)hipStreamDestroy(sim->aux_wave); }
    if (sim->aux_restart) { (void)hipStreamSynchronize(sim->aux_restart); (void)hipStreamDestroy(sim->aux_restart); }
    if (sim->ev_fork) (void)hipEventDestroy(sim->ev_fork);
    if (sim->ev_wave) (void)hipEventDestroy(sim->ev_wave);
    if (sim->ev_restart) (void)hipEventDestroy(sim->ev_restart);
    if (sim->ev_ret) (void)hipEventDestroy(sim->ev_ret);
    for (int k = 0; k < 4; k++)
        if (sim->ev_refill[k]) (void)hipEventDestroy(sim->ev_refill[k]);
    if (sim->shadow_blob) (void)hipFree(sim->shadow_blob);
    if (sim->timeline_blob) (void)hipFree(sim->timeline_blob);
    if (sim->list_blob) (void)hipFree(sim->list_blob);
    if (sim->noise_blob) (void)hipFree(sim->noise_blob);
    if (sim->noise_out_blob) (void)hipFree(sim->noise_out_blob);

    if (sim->state_blob) (void)hipFree(sim->state_blob);
    for (int c = 0; c < kMaxTiers; c++) {
        if (sim->tier_blob[c]) (void)hipFree(sim->tier_blob[c]);
    }
    delete sim;
}

int64_t pcc_device_bytes(const pcc_sim_t *sim) {
    return sim ? (int64_t)(sim->state_bytes + sim->ring_bytes + sim->list_bytes + sim->noise_bytes + sim->shadow_bytes) : 0;
}

namespace {
// What the next reset of an env draws from -- the caller's link arrays, the ranges, the seed, the uniforms -- has changed.
// Shadows (next episodes prepared ahead of time, pcc_send_restart.hip) were drawn from the OLD generation: none of them may be
// swapped in any more (retire_env compares generations; the env restarts through the restart list, which samples at reset
// time like the reference, ns:455-477, and its shadow is prepared again from there).  And no refill launch that is still queued
// on the side stream may read the caller's old arrays after this call returns: wait for them.
void params_changed(pcc_sim_t *sim) {
    if (++sim->d.params_gen == 0u) sim->d.params_gen = 1u;
    DeviceGuard guard(sim->device);
    for (int k = 0; k < 4; k++)
        if (sim->refill_recorded[k]) (void)hipEventSynchronize(sim->ev_refill[k]);
}
}  // namespace

int pcc_set_link_params(pcc_sim_t *sim, const double *bw, const double *dl, const double *queue, const double *loss,
                        const double *rate0) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    const int given = (bw != nullptr) + (dl != nullptr) + (queue != nullptr) + (loss != nullptr) + (rate0 != nullptr);
    if (given != 0 && given != 5) return fail(PCC_EINVAL, "pass all five parameter arrays or none");
    params_changed(sim);
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
    params_changed(sim);
    for (int k = 0; k < 5; k++) { sim->d.lo[k] = lo[k]; sim->d.hi[k] = hi[k]; }
    return PCC_OK;
}

int pcc_set_rng(pcc_sim_t *sim, int mode, const double *trace, int64_t trace_stride) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    params_changed(sim);   // (a replayed trace and Philox draws do not mix inside an episode prepared ahead of time)
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
    params_changed(sim);
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
        case PCC_TUNE_LIST_MIN_ENVS: {
            if (!(value >= 0.0 && value <= 4e9)) return fail(PCC_EINVAL, "list_min_envs out of range");
            if (sim->send_pending) return fail(PCC_ESTATE, "pcc_set_tuning(LIST_MIN_ENVS) between pcc_step_send and pcc_step_retire");
            // the lists are about to be dropped: what is still owed to the envs of the restart list (new links, warm-up
            // intervals) is done first, on the stream the caller last stepped on
            DeviceGuard guard(sim->device);
            const int rc = flush_restarts(sim, sim->last_stream);
            if (rc != PCC_OK) return rc;
            sim->list_min_envs = (uint32_t)value;
            sim->clean_buf = -1;
            sim->read_buf = -1;  // (whatever was filed is dropped: the next step walks the envs in index order)
            return PCC_OK;
        }
        case PCC_TUNE_RETIRE_SORTED: sim->d.retire_sorted = value != 0.0 ? 1u : 0u; return PCC_OK;
        case PCC_TUNE_LIGHT_SNAKE: sim->d.light_snake = value != 0.0 ? 1u : 0u; return PCC_OK;
        case PCC_TUNE_WAVE_OLDEST_FIRST: sim->d.wave_oldest_first = value != 0.0 ? 1u : 0u; return PCC_OK;
        case PCC_TUNE_PRIO_LEVEL:
            if (!(value >= 0.0 && value <= 3.0)) return fail(PCC_EINVAL, "prio_level must be 0..3");
            sim->d.prio_level = (uint32_t)value;
            return PCC_OK;
        case PCC_TUNE_PRIO_LIGHT_ITEMS: sim->d.prio_light_items = value >= 4e9 ? 0xFFFFFFFFu : (uint32_t)(value < 0.0 ? 0.0 : value); return PCC_OK;
        case PCC_TUNE_PRIO_WAVE_ITEMS: sim->d.prio_wave_items = value >= 4e9 ? 0xFFFFFFFFu : (uint32_t)(value < 0.0 ? 0.0 : value); return PCC_OK;
        case PCC_TUNE_PRIO_TEAM: sim->d.prio_team = value != 0.0 ? 1u : 0u; return PCC_OK;
        case PCC_TUNE_RESTART_FORK: sim->restart_fork = value != 0.0 ? 1 : 0; return PCC_OK;
        case PCC_TUNE_RETIRE_GRID_FRAC:
            if (!(value >= 0.0 && value <= 1.0)) return fail(PCC_EINVAL, "retire_grid_frac must be in [0, 1]");
            sim->retire_grid_frac = value;
            return PCC_OK;
        case PCC_TUNE_LIGHT_FRONT_WGS:   // (round 4 experiments; measured slower and removed: profiles/r04_experiments.json)
        case PCC_TUNE_SPLIT_STREAMS:
            if (value != 0.0) return fail(PCC_EINVAL, "tuning key %d was an experiment of round 4 and is gone (only 0 is accepted)", key);
            return PCC_OK;
        case PCC_TUNE_LIGHT_HALF_PREDICT:
            if (!(value >= 0.0)) return fail(PCC_EINVAL, "light_half_predict out of range");
            sim->d.light_half_predict = value >= 1e9 ? 1e9f : (float)value;
            return PCC_OK;
        case PCC_TUNE_PARTS: {
            if (value != 1.0 && value != (double)kParts) return fail(PCC_EINVAL, "parts must be 1 or %u", kParts);
            if (sim->send_pending) return fail(PCC_ESTATE, "pcc_set_tuning(PARTS) between pcc_step_send and pcc_step_retire");
            if ((uint32_t)value == sim->d.parts) return PCC_OK;
            // the pool stacks are per partition: everybody back into tier 0, the stacks rebuilt -- a reset must follow
            DeviceGuard guard(sim->device);
            if (hipDeviceSynchronize() != hipSuccess) return fail(PCC_EHIP, "hipDeviceSynchronize failed");
            set_parts(sim, (uint32_t)value);
            const int rc = sim->pools_pending ? PCC_OK : init_pool_stacks(sim);   // (pools that do not exist yet get their stacks with them)
            if (rc != PCC_OK) return rc;
            launch_forget_ring_slots(sim->d, nullptr);
            if (hipDeviceSynchronize() != hipSuccess) return fail(PCC_EHIP, "forget_ring_slots_kernel failed");
            sim->ever_reset = false;
            sim->restarts_pending = false;
            sim->read_buf = -1;
            sim->clean_buf = -1;   // (the views of the list buffers moved)
            return PCC_OK;
        }
        case PCC_TUNE_RETIRE_WIDE_PREDICT:
            if (!(value >= 0.0)) return fail(PCC_EINVAL, "retire_wide_predict out of range");
            sim->d.retire_wide_predict = value >= 1e9 ? 1e9f : (float)value;
            return PCC_OK;
        case PCC_TUNE_LIGHT_WGS:
            if (!(value >= 0.0 && value <= 65536.0)) return fail(PCC_EINVAL, "light_wgs out of range");
            sim->light_wgs = (uint32_t)value;
            return PCC_OK;
        case PCC_TUNE_LIGHT_FRONT:
            if (!(value >= 0.0 && value <= 65536.0)) return fail(PCC_EINVAL, "light_front out of range");
            sim->light_front = (uint32_t)value;
            return PCC_OK;
        case PCC_TUNE_NOISE_SORTED:
            if (value != 0.0 && value != 1.0 && value != 2.0) return fail(PCC_EINVAL, "noise_sorted must be 0, 1 or 2");
            sim->noise_sorted = (int)value;
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
        case PCC_TUNE_FUSED:
            // experimental: its hand-off inside a launch counts on what was validated on one configuration
            // only -- a gfx950 with 8 XCDs in one partition (XCC_ID & 7 names the L2 a wavefront shares with its consumers)
            // (0 = the runtime does not say: not refused)
            if (value != 0.0 && sim->xcc_count > 0 && sim->xcc_count != (int)kXcds)
                return fail(PCC_EINVAL, "the one-launch step is validated for a device of %u XCDs in one partition; this one reports %d", kXcds, sim->xcc_count);
            sim->fused = value == 2.0 ? 2 : (value != 0.0 ? 1 : 0);
            return PCC_OK;
        case PCC_TUNE_FUSED_ACQUIRE:
            if (value != 0.0 && value != 2.0) return fail(PCC_EINVAL, "fused_acquire must be 0 or 2");
            sim->d.fused_acquire = (uint32_t)value;
            return PCC_OK;
        case PCC_TUNE_FUSED_LIGHT_FRONT:
            if (!(value >= 0.0 && value <= 4096.0)) return fail(PCC_EINVAL, "fused_light_front out of range");
            sim->fused_light_front = (uint32_t)value;
            return PCC_OK;
        case PCC_TUNE_FUSED_DEBUG: sim->d.fused_debug = (uint32_t)value; return PCC_OK;
        case PCC_TUNE_FUSED_PARTIAL_NAPS:
            if (!(value >= 0.0 && value <= 1e6)) return fail(PCC_EINVAL, "fused_partial_naps out of range");
            sim->d.fused_partial_naps = (uint32_t)value;
            return PCC_OK;
        case PCC_TUNE_FUSED_MAX_NAPS:
            if (!(value >= 1.0 && value <= 1024.0)) return fail(PCC_EINVAL, "fused_max_naps out of range");
            sim->d.fused_max_naps = (uint32_t)value;
            return PCC_OK;
        case PCC_TUNE_FUSED_LIGHT_WGS:
            if (!(value >= 1.0 && value <= 4096.0)) return fail(PCC_EINVAL, "fused_light_wgs out of range");
            sim->fused_light_wgs = (uint32_t)value;
            return PCC_OK;
        default: return fail(PCC_EINVAL, "unknown tuning key %d", key);
    }
}

namespace {
// The pools of tiers 1, 2, 3 with a slot for one sender in div[c] (tier 0 is a slot per sender), their free stacks full.
int alloc_pools(pcc_sim_t *sim, const unsigned (&div)[kMaxTiers]) {
    Dev &d = sim->d;
    for (int c = 1; c < d.n_tiers; c++) {
        const int rc = alloc_tier(sim, c, div[c]);
        if (rc != PCC_OK) return rc;
    }
    sim->pools_pending = false;
    return init_pool_stacks(sim);
}

// Default pool sizes, at the handle's first reset.  A sender keeps the slots it was promoted into until its env is reset, so
// the pools of tiers 1, 2, 3 can never run dry when each has a slot for every sender (divisor 1) -- and a policy that climbs
// towards the rate limit on every link (what PPO learns on generous links) does need most of that.  An MI355X has 288 GB: the
// pools get what a third of the memory that is free right now pays for, the largest tier halved first (round 3 sized them
// for U(-1, 1) policies -- divisors 2, 8, 32, measured need ~25 %, ~2 %, ~0.02 % of the senders -- and a saturating policy
// ended a training run with PCC_FLAG_POOL_EXHAUSTED); those divisors are the floor.  pcc_set_ring_pools overrides.
int ensure_pools(pcc_sim_t *sim) {
    if (!sim->pools_pending) return PCC_OK;
    Dev &d = sim->d;
    unsigned div[kMaxTiers] = {1, 1, 1, 1};
    const unsigned floor_div[kMaxTiers] = {1, 2, 8, 32};
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    const double budget = 0.33 * (double)free_b;
    const double senders = (double)d.n * d.ns;
    auto tier_bytes = [&](int c) { return senders / div[c] * 3.0 * (double)((size_t)d.cap0 << (2 * c)) * sizeof(double2); };
    for (;;) {
        double total = 0.0;
        int big = -1;
        for (int c = 1; c < d.n_tiers; c++) {
            total += tier_bytes(c);
            if (div[c] < floor_div[c] && (big < 0 || tier_bytes(c) > tier_bytes(big))) big = c;
        }
        if (total <= budget || big < 0) break;
        div[big] *= 2;
    }
    return alloc_pools(sim, div);
}
}  // namespace

int pcc_set_ring_pools(pcc_sim_t *sim, uint32_t div1, uint32_t div2, uint32_t div3) {
    if (!sim) return fail(PCC_EINVAL, "sim is NULL");
    if (sim->send_pending) return fail(PCC_ESTATE, "pcc_set_ring_pools between pcc_step_send and pcc_step_retire");
    const unsigned div[kMaxTiers] = {1, div1, div2, div3};
    for (int c = 1; c < kMaxTiers; c++)
        if (div[c] < 1) return fail(PCC_EINVAL, "pool divisors must be >= 1 (1 = a slot for every sender)");
    DeviceGuard guard(sim->device);
    if (hipDeviceSynchronize() != hipSuccess) return fail(PCC_EHIP, "hipDeviceSynchronize failed");
    Dev &d = sim->d;
    { const int rc = alloc_pools(sim, div); if (rc != PCC_OK) return rc; }
    launch_forget_ring_slots(d, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return fail(PCC_EHIP, "forget_ring_slots_kernel failed");
    sim->ever_reset = false;  // whatever was in flight lived in the old pools: a reset must follow
    sim->restarts_pending = false;
    sim->read_buf = -1;
    return PCC_OK;
}

// The event-loop build (event_engine) runs the interval when packets can overtake each other (latency noise) and when
// windows couple two senders' SEND streams to their notifications; it needs a heap and an RTT list per sender.
constexpr size_t kHeapSlack = 8;   // pcc_retire_env.h: kHeapPad (the heaps' nodes sit 7 slots into their arrays: a node's children are one line)
int update_engine(pcc_sim_t *sim) {
    const int engine = (sim->d.use_noise || (sim->d.use_cwnd && sim->d.ns > 1)) ? 1 : 0;
    if (engine && !sim->noise_blob) {
        DeviceGuard guard(sim->device);
        const size_t senders = (size_t)sim->d.n * sim->d.ns;
        const size_t heaps = senders * (sim->ring_capacity + kHeapSlack) * sizeof(double2), lists = senders * sim->ring_capacity * sizeof(double2);
        // both blobs are allocated before either is published: a failure leaves the handle as it was (a retry starts over)
        void *p = nullptr, *po = nullptr;
        const size_t outs = (size_t)sim->d.n * sim->d.ns * sizeof(NoiseOut);   // [S][N]: entry s * N + i = sender s of env i (the env's own fields in sender 0's)
        if (hipMalloc(&p, heaps + lists) != hipSuccess)
            return fail(PCC_ENOMEM, "hipMalloc of %zu bytes for the event heaps failed", heaps + lists);
        if (hipMalloc(&po, outs) != hipSuccess) {
            (void)hipFree(p);
            return fail(PCC_ENOMEM, "hipMalloc of %zu bytes for the intervals' results failed", outs);
        }
        if (hipMemset(po, 0, outs) != hipSuccess) {
            (void)hipFree(po);
            (void)hipFree(p);
            return fail(PCC_EHIP, "hipMemset of the intervals' results failed");
        }
        sim->noise_blob = p;
        sim->noise_out_blob = po;
        sim->noise_bytes = heaps + lists + outs;
        sim->d.noise_heap = static_cast<double2 *>(p);
        sim->d.noise_rtt = sim->d.noise_heap + senders * (sim->ring_capacity + kHeapSlack);
        sim->d.noise_out = static_cast<NoiseOut *>(po);
        sim->d.noise_seq = 0;
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
    // (a masked reset pushes the ring slots its envs hold back onto the pools' stacks: before the first full reset -- or
    // after the pools were rebuilt, which asks for one -- the senders' slot ids mean nothing)
    if (mask && !sim->ever_reset) return fail(PCC_ESTATE, "a masked pcc_reset needs a full pcc_reset (mask = NULL) before it");
    DeviceGuard guard(sim->device);
    { const int rc = ensure_pools(sim); if (rc != PCC_OK) return rc; }
    sim->last_stream = static_cast<hipStream_t>(stream);
    if (!mask) {
        sim->restarts_pending = false;  // everything starts over
        // ... the shadows too: whatever refill is still in flight finishes first, the rows are emptied, nothing is trusted
        for (int k = 0; k < 4; k++) {
            if (sim->refill_recorded[k]) (void)hipStreamWaitEvent(static_cast<hipStream_t>(stream), sim->ev_refill[k], 0);
            sim->refill_recorded[k] = false;
        }
        (void)hipMemsetAsync(sim->d.refill_count, 0, 4 * kCntStride * sizeof(uint32_t), static_cast<hipStream_t>(stream));
        sim->d.shadows = 0;
    }
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
// every step has its sequence number (never 0): an env that finishes its episode leaves it in Dev::any_done, and the gated
// auto-reset launches of the same step look for it
void next_step_seq(pcc_sim_t *sim, hipStream_t st) {
    if (++sim->step_seq == 0u) sim->step_seq = 4u;   // (never 0; a multiple of 4 keeps the refill rows' rotation)
    sim->d.step_seq = sim->step_seq;
    sim->last_stream = st;
    // shadows prepared by the refill launch queued three steps ago may be swapped in from this step on (retire_env trusts a
    // shadow three steps after its fill_seq): the caller's stream waits for that launch -- it finished long ago
    const uint32_t row = (sim->step_seq - 3u) & 3u;
    if (sim->d.shadows && sim->refill_recorded[row]) (void)hipStreamWaitEvent(st, sim->ev_refill[row], 0);
}

// 0: finished envs are reset by the gated launches after the step (or the host knows the boundary: lockstep); 1: inside the
// step's own launches, through the restart list; 3: ... and by swapping in their shadows where those are ready
int restart_mode(pcc_sim_t *sim, int auto_reset) {
    if (!restarts_in_step(sim, auto_reset)) { sim->d.shadows = 0; return 0; }
    return use_shadows(sim) ? 3 : 1;
}

// Behind the retire launch of step seq: the refill of the shadows that were emptied (or invalidated) at step seq - 1 -- the
// envs they belong to have moved off the shadows' rings in this step's send half -- on the side stream, never joined into this
// step or the next.
void queue_refill(pcc_sim_t *sim, hipStream_t st) {
    const Dev &d = sim->d;
    const uint32_t seq = sim->step_seq - 1u, row = seq & 3u;
    if (hipEventRecord(sim->ev_ret, st) != hipSuccess) return;
    (void)hipStreamWaitEvent(sim->aux_restart, sim->ev_ret, 0);
    const unsigned grid = (unsigned)(sim->cu_count < (d.n + 3) / 4 ? sim->cu_count : (d.n + 3) / 4);
    launch_refill(d, grid, sim->aux_restart, row, seq);
    (void)hipMemsetAsync(d.refill_count + row * kCntStride, 0, sizeof(uint32_t), sim->aux_restart);
    (void)hipEventRecord(sim->ev_refill[row], sim->aux_restart);
    sim->refill_recorded[row] = true;
}

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
    next_step_seq(sim, static_cast<hipStream_t>(stream));
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
    const int restart = restart_mode(sim, auto_reset);
    const int rc = launch_retire_half(sim, 0, 0, 0, 0, restart, obs_out, reward_out, done_out, steps_out, st);
    if (rc != PCC_OK) return rc;
    if (restart & 2) queue_refill(sim, st);
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
    next_step_seq(sim, st);
    const Dev &d = sim->d;
    if (d.n < (int64_t)sim->list_min_envs && !d.engine) {
        // a small batch: both halves in one launch (step_small_kernel)
        launch_step_small(d, d.rng_mode == PCC_RNG_TRACE, st, actions, actions_f64, obs_out, reward_out, done_out, steps_out, 1, 0);
        const int rc0 = check_hip(hipGetLastError(), "step kernel launch");
        if (rc0 != PCC_OK) return rc0;
        return after_mi(sim, obs_out, auto_reset, st);
    }
    const int restart = restart_mode(sim, auto_reset);
    const int rc = launch_mi(sim, 0, 0, 0, 0, restart, actions, actions_f64, obs_out, reward_out, done_out, steps_out, st);
    if (rc != PCC_OK) return rc;
    if (restart & 2) queue_refill(sim, st);
    return after_mi(sim, obs_out, auto_reset, st);
}

int pcc_step_many(pcc_sim_t *sim, const void *actions, int actions_f64, int n_steps, float *obs_out, float *reward_out,
                  uint8_t *done_out, double *steps_out, int auto_reset, void *stream) {
    if (!sim || !actions || n_steps < 1) return fail(PCC_EINVAL, "NULL argument or n_steps < 1");
    // (what pcc_step would refuse is refused before the first step: after that only a failing launch can stop the loop)
    if (!sim->ever_reset) return fail(PCC_ESTATE, "pcc_step_many before pcc_reset (the reference raises TypeError: run_dur is None)");
    if (sim->send_pending) return fail(PCC_ESTATE, "pcc_step_many between pcc_step_send and pcc_step_retire");
    const Dev &d = sim->d;
    const size_t row = (size_t)d.n * d.ns;
    const size_t act_row = row * (d.use_cwnd ? 2u : 1u) * (actions_f64 ? sizeof(double) : sizeof(float));
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto obs_at = [&](int t) { return obs_out ? obs_out + (size_t)t * row * d.HF : nullptr; };
    int t = 0;
    if (d.n < (int64_t)sim->list_min_envs && !d.engine && (sim->lockstep || !auto_reset)) {
        // a small batch whose episode boundaries the host knows (or that is never reset here): the steps up to the next
        // boundary are ONE launch -- the loop over them runs inside step_small_kernel, a workgroup per 64 envs
        DeviceGuard guard(sim->device);
        while (t < n_steps) {
            int seg = n_steps - t;
            if (sim->lockstep && auto_reset) {
                const int left = (int)d.max_steps - (int)sim->host_steps;
                if (seg > (left > 1 ? left : 1)) seg = left > 1 ? left : 1;
            }
            next_step_seq(sim, st);
            launch_step_small(d, d.rng_mode == PCC_RNG_TRACE, st, static_cast<const char *>(actions) + (size_t)t * act_row, actions_f64,
                              obs_at(t), reward_out ? reward_out + (size_t)t * row : nullptr, done_out ? done_out + (size_t)t * d.n : nullptr,
                              steps_out ? steps_out + (size_t)t * row * PCC_STEP_COLS : nullptr, seg, (int64_t)act_row);
            int rc = check_hip(hipGetLastError(), "step kernel launch");
            sim->host_steps += (uint32_t)(seg - 1);
            if (rc == PCC_OK) rc = after_mi(sim, obs_at(t + seg - 1), auto_reset, st);  // (the boundary's reset writes that step's observation row)
            if (rc != PCC_OK) {
                char why[400];
                snprintf(why, sizeof why, "%s", g_err);
                return fail(rc, "pcc_step_many stopped after %d of %d steps: %s", t, n_steps, why);
            }
            t += seg;
        }
        return PCC_OK;
    }
    for (; t < n_steps; t++) {
        const int rc = pcc_step(sim, static_cast<const char *>(actions) + (size_t)t * act_row, actions_f64, obs_at(t),
                                reward_out ? reward_out + (size_t)t * row : nullptr, done_out ? done_out + (size_t)t * d.n : nullptr,
                                steps_out ? steps_out + (size_t)t * row * PCC_STEP_COLS : nullptr, auto_reset, stream);
        if (rc != PCC_OK) {  // a launch failed: say how far the batch got (its clocks have advanced that many steps)
            char why[400];
            snprintf(why, sizeof why, "%s", g_err);
            return fail(rc, "pcc_step_many stopped after %d of %d steps: %s", t, n_steps, why);
        }
    }
    return PCC_OK;
}

int pcc_debug_addresses(pcc_sim_t *sim, uint64_t *out8) {
    if (!sim || !out8) return fail(PCC_EINVAL, "NULL argument");
    out8[0] = reinterpret_cast<uint64_t>(sim->state_blob);
    for (int c = 0; c < kMaxTiers; c++) out8[1 + c] = reinterpret_cast<uint64_t>(sim->tier_blob[c]);
    out8[5] = reinterpret_cast<uint64_t>(sim->list_blob);
    out8[6] = reinterpret_cast<uint64_t>(sim->shadow_blob);
    out8[7] = reinterpret_cast<uint64_t>(sim->d.hist);
    return PCC_OK;
}

int pcc_fused_steps(pcc_sim_t *sim, uint64_t *out) {
    if (!sim || !out) return fail(PCC_EINVAL, "NULL argument");
    *out = sim->fused_steps;
    return PCC_OK;
}

int pcc_restart_stats(pcc_sim_t *sim, uint64_t *out2, void *stream) {
    if (!sim || !out2) return fail(PCC_EINVAL, "NULL argument");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(out2, sim->d.restart_stats, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(PCC_EHIP, "reading the restart statistics failed");
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
    // (sender blocks are [S][2N] -- every env's block is followed, N blocks on, by its shadow's: one strided copy per sender)
    const size_t parts = rows == sn ? (size_t)d.ns : 1;
    for (size_t sdr = 0; sdr < parts; sdr++) {
        const int rc = check_hip(hipMemcpy2DAsync(static_cast<char *>(out) + sdr * n * width, width, src + sdr * (size_t)d.stride * 128, 128, width, n,
                                                  hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)), "pcc_get_state copy");
        if (rc != PCC_OK) return rc;
    }
    return PCC_OK;
}

}  // extern "C"

