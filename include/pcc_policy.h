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

/*
 * One optimiser step of PPO1's objective on one minibatch of a rollout -- what stable-baselines does per minibatch
 * inside PPO1.learn (src/gym/stable_solve.py:52: clip 0.2, entropy coefficient, Adam; its policy :39-45) -- as two
 * launches: the gradient of  -mean(min(r A, clip(r, 1 - clip, 1 + clip) A)) + 0.5 mean((v - ret)^2) - ent_coef * entropy
 * over the samples perm[start .. start + count) (perm NULL: start .. start + count), then Adam (torch.optim.Adam's
 * arithmetic) on `params` in place.  All pointers are device memory, fp32 (perm: int64).
 *
 * obs [n][obs_dim], act / logp_old / adv / ret [n]   the flattened rollout (adv already normalised by the caller)
 * params    the block pcc_policy_act reads; adam_m / adam_v [n_params] the optimiser state; adam_step = 1, 2, ...
 * lr == 0   gradient only: nothing is updated (adam_m / adam_v may be NULL)
 * scratch   pcc_ppo_scratch_floats(obs_dim, h1, h2) floats
 * grad_out  [n_params] or NULL: the gradient that was applied
 * stats_out [4] or NULL: {mean clipped surrogate (= -policy loss), mean squared value error (= 2 x value loss),
 *           fraction of samples with |r - 1| > clip, 0}
 * Returns 0; -1 bad arguments; -2 no kernel for this shape (h1, h2 = 32, 16 and obs_dim 30, 12, 6, 3 are built: the caller
 * falls back to its framework path); -3 launch failure.
 */
int pcc_ppo_scratch_floats(int obs_dim, int h1, int h2);
int pcc_ppo_minibatch_step(const float *obs, const float *act, const float *logp_old, const float *adv, const float *ret,
                           const int64_t *perm, int64_t start, int64_t count, int obs_dim, int h1, int h2, float *params,
                           float *adam_m, float *adam_v, int adam_step, float lr, float beta1, float beta2, float eps,
                           float clip, float ent_coef, float *scratch, float *grad_out, float *stats_out, void *stream);

/*
 * Generalised advantage estimation over the [T][n_envs] rows of a rollout, one launch: dones[t][i] != 0 marks that env i
 * was reset after step t.  adv_out / ret_out [T][n_envs].
 */
int pcc_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_value, int T, int64_t n_envs,
            float gamma, float lam, float *adv_out, float *ret_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCC_POLICY_H */
