// pcc_send.hip -- send_kernel: the send half of a step as ONE launch that holds both kinds of workgroup
// (pcc_send_bodies.h): the SEND events (ns:155-178) of every env's coming monitor interval, apply_rate_delta
// (ns:235-241, 275-281) in front.
//
// Grid (workgroups of 4 wavefronts), in dispatch order -- the order matters: a compute unit's memory pipeline serves its
// oldest wavefronts first, and four lane-round wavefronts keep it busy all the time -- as the oldest they starve everybody
// else on the compute unit (a wave-path item next to them waits 100-160 us for its first loads), as the youngest they fill
// the gaps:
//   [0, front)                      light workgroups 0 .. front-1: the longest light items, the launch's critical path for
//                                   most of an episode, start on the first-served slots (one workgroup per compute unit);
//   [front, front + wave_wgs)       the wave-path workgroups (persistent; the first of them take the team items);
//   [front + wave_wgs, ...)         the other light workgroups.
// Without work lists (read_buf < 0) there are only light workgroups, the envs in index order.
// Both bodies fit 128 registers without spilling (tools/resources.sh; tests/test_abi_cpu.py checks the build's metadata):
// round 3's kernel of the same name kept a lane's env in registers across the wave passes and spilled 56-160 bytes per
// lane, and its code generation broke when the wave paths grew.
#include "pcc_send_bodies.h"
#include "pcc_kernels.h"

#ifndef PCC_SEND_OCC2
#define PCC_SEND_OCC2 4  // ... of the two-sender builds
#endif
#ifndef PCC_SEND_OCC
#define PCC_SEND_OCC 4  // workgroups (4 wavefronts) per compute unit the register budget is cut for
#endif

namespace {

template <int NS, bool TRACE>
__global__ __launch_bounds__(4 * kWave, NS == 2 ? PCC_SEND_OCC2 : PCC_SEND_OCC) void send_kernel(Dev D, int read_buf, int zero_buf, int warm,
                                                                                       uint32_t warm_mi, int gate, uint32_t front,
                                                                                       uint32_t wave_wgs, const void *actions,
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
    const uint32_t b = blockIdx.x, Q = gridDim.x - wave_wgs;  // Q light workgroups
    if (b >= front && b < front + wave_wgs)
        wave_body<NS, TRACE>(D, lds, lane, wv, b - front, wave_wgs, read_buf, actions, actions_f64);
    else
        light_body<NS, TRACE>(D, lds, lane, wv, b < front ? b : b - wave_wgs, Q, read_buf, warm, warm_mi, actions, actions_f64);
}

}  // namespace

namespace pcc {

void launch_send(const Dev &d, bool trace, unsigned light_wgs, unsigned wave_wgs, unsigned front, hipStream_t st, int read_buf,
                 int zero_buf, int warm, uint32_t warm_mi, int gate, const void *actions, int actions_f64) {
    if (front > light_wgs) front = light_wgs;
#define PCC_S(NS_, TR_)                                                                                                           \
    hipLaunchKernelGGL((send_kernel<NS_, TR_>), dim3(light_wgs + wave_wgs), dim3(4 * kWave), 0, st, d, read_buf, zero_buf, warm, warm_mi, \
                       gate, front, wave_wgs, actions, actions_f64)
    if (d.ns == 1) { if (trace) PCC_S(1, true); else PCC_S(1, false); }
    else { if (trace) PCC_S(2, true); else PCC_S(2, false); }
#undef PCC_S
}

}  // namespace pcc
