// pcc_send_wave.hip -- send_wave_kernel: the wave-path items and the team items of the send half (pcc_send_item.h:
// send_wave_item; pcc_wave_pass.h: heavy_mi / heavy_mi2), i.e. the SEND events (ns:155-178) of the envs that send so many
// packets per monitor interval that all 64 lanes of a wavefront -- or the four wavefronts of a workgroup -- working on
// ONE env's closed-form passes beat a lane per env.
//
// Persistent wavefronts (send_waves per compute unit, workgroups of 4) take work items until none are left:
//   * a wave-path item: envs of a class from heavy_predict packets up.  An item holds as many envs as make up about
//     heavy_item_packets packets (1-8): lanes 0..e-1 load an env each, so that the claim, the list entry and the state --
//     three dependent round trips -- are paid once per item; the wavefront then sends them one after the other;
//   * a team item: one env predicted above team_predict packets, sent by the four wavefronts of a workgroup together
//     (heavy_mi<.., 4>, 1 024 positions per pass).  The oldest workgroups (first served by their compute unit's memory
//     pipeline) take them -- workgroup b items b, b + n_tw, ... -- and then claim like everybody.
// Items are ranked by class, largest first.  The first item of a wavefront is dealt statically (no atomic), the rest comes
// off 16 sharded cursors (one returning atomic on one word saturates near 90 claims/us).
// Speed only: which wavefront sends an env never changes a result.
#include "pcc_send_bodies.h"
#include "pcc_kernels.h"

#ifndef PCC_WAVE_OCC2
#define PCC_WAVE_OCC2 4  // ... of the two-sender builds
#endif
#ifndef PCC_WAVE_OCC
#define PCC_WAVE_OCC 4  // wave-path workgroups (4 wavefronts) per compute unit the register budget is cut for
#endif

namespace {

template <int NS, bool TRACE>
__global__ __launch_bounds__(4 * kWave, NS == 2 ? PCC_WAVE_OCC2 : PCC_WAVE_OCC) void send_wave_kernel(Dev D, int read_buf, const void *actions, int actions_f64) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    __shared__ SendLds<NS> lds;
    wave_body<NS, TRACE>(D, lds, lane, wv, blockIdx.x, gridDim.x, read_buf, actions, actions_f64);
}

}  // namespace

namespace pcc {

void launch_send_wave(const Dev &d, bool trace, unsigned grid, hipStream_t st, int read_buf, const void *actions, int actions_f64) {
#define PCC_W(NS_, TR_) hipLaunchKernelGGL((send_wave_kernel<NS_, TR_>), dim3(grid), dim3(4 * kWave), 0, st, d, read_buf, actions, actions_f64)
    if (d.ns == 1) { if (trace) PCC_W(1, true); else PCC_W(1, false); }
    else { if (trace) PCC_W(2, true); else PCC_W(2, false); }
#undef PCC_W
}

}  // namespace pcc
