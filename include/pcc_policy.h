/*
 * pcc_policy.h -- C ABI of the fused policy forward of the on-device PPO caller (libpcc_sim.so).
 *
 * Not part of the reference's env path: the reference's agent is stable-baselines PPO1 on
 * TensorFlow 1 (src/gym/stable_solve.py:39-58), a caller of the env.  This entry point evaluates that
 * script's policy architecture (ibid. :39-45: pi and vf MLPs with two tanh hidden layers, --arch, and a
 * state-independent log-std) for a whole env batch in one kernel launch, so that the batched env
 * (pcc_sim.h) is not starved by the agent.  fp32.
 */
#ifndef PCC_POLICY_H
#define PCC_POLICY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * obs       [n_envs][obs_dim] float32, device (the observation rows pcc_step / pcc_reset wrote)
 * params    device floats: pi {W1[h1][obs_dim], b1[h1], W2[h2][h1], b2[h2], W3[h2], b3, log_std},
 *           then vf {the same without log_std} -- row-major like torch.nn.Linear.weight
 * noise     [n_envs] standard-normal draws, device; NULL = deterministic (act = mean, logp of the mean)
 * mean_out, act_out, logp_out, value_out   [n_envs] each, device; any may be NULL
 * stream    HIP stream; nothing synchronizes
 * Returns 0; -1 bad arguments / too many parameters for LDS; -2 no kernel instantiated for this obs_dim
 * (30, 36, 3, 6, 12, 60 are: the caller falls back to its framework path); -3 launch failure.
 */
int pcc_policy_act(const float *obs, int64_t n_envs, int obs_dim, const float *params, int h1, int h2,
                   const float *noise, float *mean_out, float *act_out, float *logp_out, float *value_out,
                   void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCC_POLICY_H */
