// pcc_kernels.h -- the launch functions of the kernel files (each .hip defines its kernels and the host function that
// launches them; pcc_sim.hip, the C ABI, calls these).  One translation unit per kernel family, so that every kernel
// is compiled -- and register-allocated -- on its own.
#pragma once
#include "pcc_dev.h"

namespace pcc {

// pcc_send.hip: both kinds of workgroup in one launch -- wave_wgs wave-path workgroups, then light_wgs light workgroups (with
// lists both are multiples of Dev::parts: workgroup b works for partition b % parts).
void launch_send(const Dev &d, bool trace, unsigned light_wgs, unsigned wave_wgs, unsigned light_front, hipStream_t st, int read_buf,
                 int zero_buf, int warm, uint32_t warm_mi, int gate, const void *actions, int actions_f64);
// pcc_send_restart.hip.  grid: workgroups of 4 wavefronts, restart items dealt statically.
void launch_send_restart(const Dev &d, bool trace, unsigned grid, hipStream_t st, int read_buf, const void *actions, int actions_f64);
// ... refill_kernel: the shadows of the envs in refill row `row` (their next episodes: new links + warm-up intervals)
void launch_refill(const Dev &d, unsigned grid, hipStream_t st, uint32_t row, uint32_t fill_seq);
// pcc_retire.hip
void launch_retire(const Dev &d, bool noise, unsigned grid, hipStream_t st, int read_buf, int fill_buf, int warm, uint32_t warm_mi,
                   int last_warm, int gate, int restart, float *obs_out, float *reward_out, uint8_t *done_out, double *steps_out,
                   const void *actions, int actions_f64);
// pcc_fused.hip: both halves of a full-size step in one launch (an env's retire half follows its own send half); grid =
// wave_wgs workgroups that start with the wave-path work + the rest, both multiples of Dev::parts
void launch_noise_sorted(const Dev &d, hipStream_t st, int warm, uint32_t warm_mi, int gate, const void *actions, int actions_f64,
                         int only_small);
void launch_step_fused(const Dev &d, bool trace, unsigned grid, unsigned wave_wgs, unsigned light_front, hipStream_t st, int read_buf, int fill_buf, int zero_buf,
                       int retire_on, const void *actions, int actions_f64, float *obs_out, float *reward_out, uint8_t *done_out, double *steps_out);
void launch_clear_list_buffer(const Dev &d, hipStream_t st, int buf);
int fused_resident_blocks(int ns, bool trace);   // workgroups of step_fused_kernel a compute unit holds at once (0: unknown)
// pcc_small.hip
// n_steps steps inside one launch: step t takes actions + t * act_stride bytes and writes row t of every output
void launch_step_small(const Dev &d, bool trace, hipStream_t st, const void *actions, int actions_f64, float *obs_out,
                       float *reward_out, uint8_t *done_out, double *steps_out, int n_steps, int64_t act_stride);
void launch_reset_init(const Dev &d, hipStream_t st, const uint8_t *mask, int use_done, int gate, int all_envs, float *obs_out);
void launch_forget_ring_slots(const Dev &d, hipStream_t st);

constexpr int kRetireEnvsPerBlockNarrow = 16;  // envs of a retire workgroup at 8 lanes per env (16 lanes: half)

}  // namespace pcc
