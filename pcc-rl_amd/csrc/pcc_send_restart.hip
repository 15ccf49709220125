// pcc_send_restart.hip -- send_restart_kernel: the restart items of the send half.  An env that finished its episode while
// the batch is out of lockstep (the host cannot know which step ends whose episode) was marked by the retire launch
// (resetting = 2) and filed in the restart list; here it gets its new links and fresh state (ns:469-477), the two
// unrecorded warm-up intervals (ns:478-479: send + retire each, by one wavefront) and then the send half of its first
// interval, like every other env of the step.  One env per item; runs beside the light and the wave kernel on a third
// stream.  Cut for two workgroups per compute unit (256 registers): the warm-up retire is inlined, and there are only as
// many items as envs finished in the last step.
#include "pcc_send_item.h"
#include "pcc_retire_env.h"
#include "pcc_kernels.h"

#ifndef PCC_RESTART_OCC
#define PCC_RESTART_OCC 2  // (a variant build cuts it for 4 -- 128 registers, heavy spilling -- to keep round 3's failing case under test)
#endif

namespace {

template <int NS, bool TRACE>
__global__ __launch_bounds__(4 * kWave, PCC_RESTART_OCC) void send_restart_kernel(Dev D, int read_buf, const void *actions, int actions_f64) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const uint32_t wave = blockIdx.x * 4u + wv, n_waves = gridDim.x * 4u;
    const uint32_t n_restart = D.cls_count[read_buf * kClsStride + kRestart * kCntStride];
    const uint32_t tl_base = (uint32_t)D.n + (uint32_t)(D.n / 2);  // profile build: timeline slots of the restart items
    __shared__ EnvSlot<NS> s_slots[4][kSlots];
    for (uint32_t t = wave; t < n_restart; t += n_waves) {
        const bool has = lane == 0;
        const int64_t i = has ? (int64_t)D.cls_list[((size_t)read_buf * kListRows + kRestart) * (size_t)D.n + t] : 0;
        // new links and fresh state (ns:469-477) unless a flush already did all of it (pcc_get_state, a masked reset)
        if (has && D.env[i].resetting == 2) reset_env<NS>(D, i, nullptr);
        // what one lane wrote is read by the others of this wavefront: a workgroup-scope fence is enough, and an
        // agent-scope one (__threadfence) writes back and invalidates the XCD's whole L2 under everybody's feet
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const int64_t i0 = (int64_t)__builtin_amdgcn_readfirstlane((int)i) | ((int64_t)__builtin_amdgcn_readfirstlane((int)(i >> 32)) << 32);
        for (int pass = 0; pass < 2; pass++) {  // the warm-up intervals: send, then retire by 8 lanes
            (void)send_wave_item<NS, TRACE, 1>(D, lane, i, has, true, 0xFFFFFFFFu, 1, (uint32_t)pass, actions, actions_f64, s_slots[wv]);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // records and state just written are read by other lanes
            if (lane < 8u) {
                Group g;
                g.lane = lane; g.shift = 0;
                (void)retire_env<NS, false, 8>(D, i0, g, 1, (uint32_t)pass, pass == 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        (void)send_wave_item<NS, TRACE, 1>(D, lane, i, has, true, tl_base + t, 0, 0, actions, actions_f64, s_slots[wv]);
    }
}

}  // namespace

namespace pcc {

void launch_send_restart(const Dev &d, bool trace, unsigned grid, hipStream_t st, int read_buf, const void *actions, int actions_f64) {
#define PCC_R(NS_, TR_) hipLaunchKernelGGL((send_restart_kernel<NS_, TR_>), dim3(grid), dim3(4 * kWave), 0, st, d, read_buf, actions, actions_f64)
    if (d.ns == 1) { if (trace) PCC_R(1, true); else PCC_R(1, false); }
    else { if (trace) PCC_R(2, true); else PCC_R(2, false); }
#undef PCC_R
}

}  // namespace pcc
