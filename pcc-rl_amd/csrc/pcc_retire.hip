// pcc_retire.hip -- retire_kernel: the retire half of a step for every env (pcc_retire_env.h: retire_env), and the filing
// of every env in the work lists of the next send half.  NOISE = the event-loop build of the reference's dormant engine
// options (one launch runs the whole interval; no send half).
#include "pcc_retire_env.h"
#include "pcc_kernels.h"

#ifndef PCC_RETIRE_OCC2
#define PCC_RETIRE_OCC2 3  // ... of the two-sender builds (156 registers, no scratch; at 4 they spill 128 B/lane: config 5 retire 0.147 -> 0.132 ms)
#endif

namespace {

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
__global__ __launch_bounds__(kRetireBlock, NS == 2 ? PCC_RETIRE_OCC2 : PCC_RETIRE_OCC) void retire_kernel(Dev D, int read_buf, int fill_buf, int warm,
                                                              uint32_t warm_mi, int last_warm, int gate, int restart, float *obs_out,
                                                              float *reward_out, uint8_t *done_out, double *steps_out,
                                                              const void *actions, int actions_f64) {
    if (gate && __hip_atomic_load(D.any_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != D.step_seq) return;
    __shared__ uint32_t s_env[kRetireMaxPerBlock], s_cls[kRetireMaxPerBlock], s_arrived;
    const uint32_t tid = threadIdx.x;
    if (prof_on(D) && !warm) {  // profile build: the send items' timeline slots are cleared for the next send launch
        for (int64_t slot = (int64_t)blockIdx.x * kRetireBlock + tid; slot < 2 * D.n; slot += (int64_t)gridDim.x * kRetireBlock)
            D.timeline[slot * 8] = 0;
    }
    if (prof_on(D) && !warm && blockIdx.x < 8u && tid == 0 && D.n >= 1024)   // profile build: which XCD the first blocks of this launch run on
        D.timeline[(int64_t)19 * D.n + 24 + blockIdx.x] = ((uint64_t)D.step_seq << 8) | xcc_id();
    const uint32_t lane = tid & (kWave - 1);
    if (tid == 0) s_arrived = 0u;
    if (tid < (uint32_t)kRetireMaxPerBlock) s_env[tid] = 0xFFFFFFFFu;
    __syncthreads();  // the workgroup's wavefronts start together: this one is free
    int64_t i = D.n;   // (beyond the envs: nothing)
    bool wide = false;  // this workgroup: 16 lanes per env
    // With lists, workgroup b walks the lists of partition b % parts (pcc_dev.h "partitions": the XCD that block b lands on
    // keeps reading the same eighth of the rings), as workgroup b / parts of that partition's share of the grid
    const uint32_t P = D.parts, part = blockIdx.x % P, b_loc = blockIdx.x / P, grid_loc = gridDim.x / P;
    if (read_buf >= 0) {
        const uint32_t view = list_view(D, read_buf, part);
        // lane l < kClasses looks after class kClasses-1-l, lane kClasses after the restart list (envs that were reset
        // by the retire launch before this one: last); inclusive prefix of the counts in that order
        const uint32_t row_mine = lane < (uint32_t)kClasses ? (uint32_t)(kClasses - 1) - lane : (uint32_t)kRestart;
        const uint32_t n_mine = lane <= (uint32_t)kClasses ? *cls_count_of(D, view, row_mine) : 0u;
        uint32_t incl = n_mine;
        for (int o = 1; o <= kClasses; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= (uint32_t)o) incl += up;
        }
        const uint32_t total = rl_u32(incl, kClasses);
        const int cls_wide = NOISE ? kClasses : (D.retire_wide_predict >= 1e9f ? kClasses : class_of(D.retire_wide_predict));
        uint32_t n_top = cls_wide < kClasses ? rl_u32(incl, (uint32_t)(kClasses - 1 - cls_wide)) : 0u;  // envs of the wide classes
        // The host sizes the grid for the usual share of wide envs, not for the worst case (twice n / 16 workgroups: the
        // command processor dispatches ~80 workgroups per microsecond, and 8 193 of them, half of them finding nothing to do,
        // ARE the 0.1 ms this launch took in round 3).  Should more envs sit in the wide classes than the grid has room for,
        // the smallest of them go 8 lanes like everybody else: lanes per env is a speed choice, every result is the same.
        {
            const uint32_t narrow_all = (total + 15u) / 16u;   // workgroups if nobody were wide
            const uint32_t spare = grid_loc > narrow_all + 1u ? grid_loc - narrow_all - 1u : 0u;   // each takes 16 wide envs' extra share
            if (n_top > 16u * spare) n_top = 16u * spare;
        }
        const uint32_t wg_wide = (n_top + 7u) / 8u;  // workgroups that take them, 8 each
        wide = b_loc < wg_wide;
        uint32_t p;  // this lane's position in the walk (the same for the lanes of a group)
        bool has;
        if (wide) {
            p = b_loc * 8u + tid / 16u;
            has = p < n_top;
        } else {
            p = n_top + (b_loc - wg_wide) * 16u + tid / 8u;
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
            i = (int64_t)cls_list_of(D, view, row)[off];
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
    // the workgroup's envs are all of one partition: the one whose lists it walked, or (index order) 16 consecutive ids
    const uint32_t fview = list_view(D, fill_buf, read_buf >= 0 ? part : min(part_of(D, (int64_t)blockIdx.x * kRetireMaxPerBlock), P - 1u));
    uint32_t base = 0u;
    if (files && leader == lane) base = atomicAdd(cls_count_of(D, fview, c), same);
    base = (uint32_t)__shfl((int)base, (int)leader);
    if (files) cls_list_of(D, fview, c)[base + rank] = e;
}

}  // namespace

namespace pcc {

void launch_retire(const Dev &d, bool noise, unsigned grid, hipStream_t st, int read_buf, int fill_buf, int warm, uint32_t warm_mi,
                   int last_warm, int gate, int restart, float *obs_out, float *reward_out, uint8_t *done_out, double *steps_out,
                   const void *actions, int actions_f64) {
#define PCC_RT(NS_, NZ_)                                                                                                          \
    hipLaunchKernelGGL((retire_kernel<NS_, NZ_>), dim3(grid), dim3(kRetireBlock), 0, st, d, read_buf, fill_buf, warm, warm_mi, last_warm, \
                       gate, restart, obs_out, reward_out, done_out, steps_out, actions, actions_f64)
    if (d.ns == 1) { if (noise) PCC_RT(1, true); else PCC_RT(1, false); }
    else { if (noise) PCC_RT(2, true); else PCC_RT(2, false); }
#undef PCC_RT
}

static_assert(kRetireEnvsPerBlockNarrow == kRetireMaxPerBlock, "pcc_kernels.h tells the host how many envs a retire workgroup takes");

}  // namespace pcc
