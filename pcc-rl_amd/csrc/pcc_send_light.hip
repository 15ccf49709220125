// pcc_send_light.hip -- send_light_kernel: the light items of the send half (pcc_send_item.h: send_light_item), i.e. the
// SEND events (ns:155-178) of the envs that send few enough packets per monitor interval that a lane each is the fastest
// way -- lane rounds: branch-free link model, four packets per Philox block, no loads in the loop.
//
// Items: with work lists (read_buf >= 0) the envs of every class below the wave-path threshold, 64 (send_envs_per_wave) of
// a class at a time, longest class first; without (after a reset, the warm-up intervals, small batches) the envs in index
// order.  One item per wavefront and round, dealt statically: the grid covers the worst case (every env light), a
// wavefront without an item leaves at once.  The four items of a workgroup share a compute unit, one per SIMD; dealt in
// snake order (workgroup b of Q: ranks b, 2Q-1-b, 2Q+b, 4Q-1-b) a workgroup's items add up to about the same number of
// packets -- the lane rounds' scattered 16-byte record stores go through the compute unit's one address path, ~0.5 G
// stores/s (profiles/r03_store_bench2.txt), and a compute unit that holds the longest item of every quartile of the
// ranking finishes last.
// The wave-path kernel (pcc_send_wave.hip) runs beside this one on another stream; the two never touch the same env.
#include "pcc_send_bodies.h"
#include "pcc_kernels.h"

#ifndef PCC_LIGHT_OCC2
#define PCC_LIGHT_OCC2 4  // ... of the two-sender builds
#endif
#ifndef PCC_LIGHT_OCC
#define PCC_LIGHT_OCC 4  // light workgroups (4 wavefronts) per compute unit the register budget is cut for
#endif

namespace {

template <int NS, bool TRACE>
__global__ __launch_bounds__(4 * kWave, NS == 2 ? PCC_LIGHT_OCC2 : PCC_LIGHT_OCC) void send_light_kernel(Dev D, int read_buf, int zero_buf, int warm,
                                                                             uint32_t warm_mi, int gate, const void *actions,
                                                                             int actions_f64) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    // auto-reset launches of a step in which no env finished have nothing to do (envs at different
    // points of their episodes: the host cannot know)
    if (gate && __hip_atomic_load(D.any_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != D.step_seq) return;
    if (blockIdx.x == 0 && wv == 0 && zero_buf >= 0) {  // the list buffer the coming retire launch files into
        if (lane <= (uint32_t)kClasses) D.cls_count[zero_buf * kClsStride + lane * kCntStride] = 0u;
        if (lane < kShards) D.cursors[((uint32_t)zero_buf * kShards + lane) * kCursorStride] = 0u;
    }
    __shared__ SendLds<NS> lds;
    light_body<NS, TRACE>(D, lds, lane, wv, blockIdx.x, gridDim.x, read_buf, warm, warm_mi, actions, actions_f64);
}

}  // namespace

namespace pcc {

void launch_send_light(const Dev &d, bool trace, unsigned grid, hipStream_t st, int read_buf, int zero_buf, int warm,
                       uint32_t warm_mi, int gate, const void *actions, int actions_f64) {
#define PCC_L(NS_, TR_) \
    hipLaunchKernelGGL((send_light_kernel<NS_, TR_>), dim3(grid), dim3(4 * kWave), 0, st, d, read_buf, zero_buf, warm, warm_mi, gate, actions, actions_f64)
    if (d.ns == 1) { if (trace) PCC_L(1, true); else PCC_L(1, false); }
    else { if (trace) PCC_L(2, true); else PCC_L(2, false); }
#undef PCC_L
}

}  // namespace pcc
