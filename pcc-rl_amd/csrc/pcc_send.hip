// pcc_send.hip -- send_kernel: the send half of a step as ONE launch that holds both kinds of workgroup
// (pcc_send_bodies.h): the SEND events (ns:155-178) of every env's coming monitor interval, apply_rate_delta
// (ns:235-241, 275-281) in front.
//
// Grid (workgroups of 4 wavefronts), in dispatch order -- the order matters: a compute unit's memory pipeline serves its
// oldest wavefronts first, and four lane-round wavefronts keep it busy all the time -- as the oldest they starve everybody
// else on the compute unit (a wave-path item next to them waits 100-160 us for its first loads), as the youngest they fill
// the gaps:
//   [0, wave_wgs)        the wave-path workgroups (persistent; the first of every partition take its team items);
//   [wave_wgs, ...)      the light workgroups.
// Both counts are multiples of the number of partitions (pcc_dev.h "partitions"): workgroup b of either kind works for
// partition b % parts -- the XCD that block b lands on -- as workgroup b / parts of that partition's share.
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
                                                                                       uint32_t warm_mi, int gate,
                                                                                       uint32_t wave_wgs, uint32_t light_front, const void *actions,
                                                                                       int actions_f64) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    // auto-reset launches of a step in which no env finished have nothing to do (envs at different
    // points of their episodes: the host cannot know)
    if (gate && __hip_atomic_load(D.any_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != D.step_seq) return;
    if (blockIdx.x == 0 && zero_buf >= 0) {  // the list buffer the coming retire launch files into: every partition's set
        for (uint32_t w = threadIdx.x; w < D.parts * (uint32_t)(kClasses + 1); w += blockDim.x)
            *cls_count_of(D, list_view(D, zero_buf, w / (uint32_t)(kClasses + 1)), w % (uint32_t)(kClasses + 1)) = 0u;
        for (uint32_t w = threadIdx.x; w < D.parts * kShards; w += blockDim.x)
            cursors_of(D, list_view(D, zero_buf, w / kShards))[(w % kShards) * kCursorStride] = 0u;
        for (uint32_t w = threadIdx.x; w < kXcds * kFctlWords; w += blockDim.x)   // (a fused step may read this buffer: pcc_fused.hip)
            *fq_word(D, zero_buf, w / kFctlWords, w % kFctlWords) = 0u;
    }
    if (prof_on(D) && blockIdx.x < 8u && threadIdx.x == 0 && D.n >= 1024)   // profile build: which XCD the first blocks of this launch run on
        D.timeline[(int64_t)19 * D.n + 16 + blockIdx.x] = ((uint64_t)D.step_seq << 8) | xcc_id();
    __shared__ SendLds<NS> lds;
    const uint32_t b = blockIdx.x;
    if (read_buf < 0) {   // (no lists: light workgroups only, the envs in index order)
        light_body<NS, TRACE>(D, lds, lane, wv, b, gridDim.x, -1, 0u, warm, warm_mi, actions, actions_f64);
        return;
    }
    // block order = dispatch order: light_front light workgroups (a multiple of P: the longest lane-round items of every partition),
    // the wave-path workgroups, the other light workgroups.  A block's partition is b % P in all three ranges.
    const uint32_t P = D.parts, part = b % P;
    if (b >= light_front && b < light_front + wave_wgs) {
        wave_body<NS, TRACE>(D, lds, lane, wv, wave_wgs, read_buf, actions, actions_f64, 0u, b - light_front);
    } else {
        // profile build: the timeline slots of a partition's light items (at most part_envs / 32 + a partial one per class)
        const uint32_t tl_base = part * (D.part_envs / 32u + (uint32_t)kClasses + 1u);
        light_body<NS, TRACE>(D, lds, lane, wv, (b < light_front ? b : b - wave_wgs) / P, (gridDim.x - wave_wgs) / P, (int)list_view(D, read_buf, part), tl_base,
                              warm, warm_mi, actions, actions_f64);
    }
}

}  // namespace

namespace pcc {

void launch_send(const Dev &d, bool trace, unsigned light_wgs, unsigned wave_wgs, unsigned light_front, hipStream_t st, int read_buf,
                 int zero_buf, int warm, uint32_t warm_mi, int gate, const void *actions, int actions_f64) {
#define PCC_S(NS_, TR_)                                                                                                           \
    hipLaunchKernelGGL((send_kernel<NS_, TR_>), dim3(light_wgs + wave_wgs), dim3(4 * kWave), 0, st, d, read_buf, zero_buf, warm, warm_mi, \
                       gate, wave_wgs, light_front, actions, actions_f64)
    if (d.ns == 1) { if (trace) PCC_S(1, true); else PCC_S(1, false); }
    else { if (trace) PCC_S(2, true); else PCC_S(2, false); }
#undef PCC_S
}

}  // namespace pcc
