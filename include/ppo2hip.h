/* ppo2hip — C ABI of the MI355X-native PPO2 rollout-to-update path.
 *
 * Drop-in boundary for pcchenxi/baselines `baselines/ppo2`: the Python host code
 * (baselines_amd/ppo2) binds exactly these entry points with ctypes.  Every
 * function returns 0 on success, a negative library code or a positive
 * hipError_t otherwise; ppo2hip_last_error() describes the last failure of the
 * calling thread.  Unless a parameter says "host", every pointer is a DEVICE
 * pointer owned by the caller (gfx950 HBM); `stream` is a hipStream_t passed as
 * void* and every call is asynchronous on it.  The library allocates no device
 * memory: scratch comes in through `workspace` arguments whose size the
 * *_workspace_bytes queries give.
 *
 * Citations are file:line in the reference tree (pcchenxi/baselines).
 */
#ifndef PPO2HIP_H
#define PPO2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPO2HIP_VERSION 1

/* library error codes (hipError_t values are positive) */
#define PPO2HIP_OK 0
#define PPO2HIP_EINVAL (-1)    /* bad argument (shape, null pointer, unsupported size) */
#define PPO2HIP_ENOSPACE (-2)  /* workspace too small */
#define PPO2HIP_EUNSUPPORTED (-3)
#define PPO2HIP_ECOMM (-4)     /* the collective library (RCCL) is missing or reported an error */

int ppo2hip_version(void);
const char *ppo2hip_last_error(void);

/* ------------------------------------------------------------------------
 * Rollout scans
 * ---------------------------------------------------------------------- */

/* GAE(lambda) advantages and returns over a time-major rollout.
 * Replaces the NumPy loop of Runner.run, baselines/ppo2/common_functions.py:308-328.
 *   rewards, values : float32 [T, E, R]     dones : uint8 [T, E] (done BEFORE step t, :258)
 *   last_values     : float32 [E, R] (:304) last_dones : uint8 [E] (:319)
 *   gamma           : HOST double[R] (self.gamma, :232), R <= 8      lam : scalar
 *   returns         : float32 [T, E, R] out (mb_returns :328)
 *   advs            : float32 [T, E, R] out or NULL (mb_advs :327 — Runner.run does not return it, :344)
 *   advs_ori        : float32 [T, E, R] out or NULL — `returns - values` as the driver
 *                     recomputes it in float32, baselines/ppo2/ppo2.py:92
 * float64 carry, float32 store, returns = advs + values in float32. */
int ppo2hip_gae_f32(const float *rewards, const float *values, const uint8_t *dones,
                    const float *last_values, const uint8_t *last_dones, const double *gamma,
                    double lam, int T, int E, int R, float *advs, float *returns, float *advs_ori,
                    void *stream);

/* n-step discounted returns with value bootstrap.
 * Replaces discount_with_dones, baselines/a2c/utils.py:154-160, together with the
 * bootstrap rule of the A2C runner, baselines/a2c/a2c.py:119-126, for all envs at once.
 *   rewards : float32 [T, E] time-major     dones : uint8 [T, E] (done AFTER step t, a2c.py:116)
 *   last_values : float32 [E] or NULL (NULL = plain discount_with_dones, no bootstrap)
 *   out : float32 [T, E] */
int ppo2hip_discount_with_dones_f32(const float *rewards, const uint8_t *dones,
                                    const float *last_values, double gamma, int T, int E,
                                    float *out, void *stream);

/* Discounted sums along axis 0: Y[t] = X[t] + gamma * Y[t+1] * (1 - starts[t+1]).
 * Replaces math_util.discount, baselines/common/math_util.py:5-23 (scipy's lfilter in float64:
 * starts NULL, out_f64 = 1), and discount_with_boundaries, baselines/common/math_util.py:63-73
 * (`Y = np.zeros_like(X)`: out has X's dtype and a float32 Y carries the rounded value).
 *   x : float32 or float64 (x_f64 != 0) [T, C]     starts : uint8 [T] episode-start flags or NULL
 *   out : float64 (out_f64 != 0) or float32 [T, C]; float64 input needs float64 output */
int ppo2hip_discount(const void *x, int x_f64, const uint8_t *starts, double gamma, int T, int64_t C,
                     void *out, int out_f64, void *stream);

/* sf01: [T, E, row] -> [E*T, row] (swap and flatten axes 0 and 1),
 * baselines/ppo2/common_functions.py:347-352.  row_bytes = bytes of one trailing row. */
int ppo2hip_sf01(const void *src, int T, int E, int64_t row_bytes, void *dst, void *stream);

/* ------------------------------------------------------------------------
 * Env-wrapper filters (the step before the rollout scan)
 * ---------------------------------------------------------------------- */

/* RunningMeanStd.update, baselines/common/running_mean_std.py:9-29: batch mean / variance over
 * axis 0 merged into the running statistics by the parallel-variance rule.
 *   x : float32 or float64 (x_f64 != 0) [n, d], n >= 1
 *   state : DEVICE double[3*d] = mean[d], var[d], count[d] (the reference's scalar count, kept
 *           once per column); initialise to 0, 1, epsilon as RunningMeanStd.__init__ does (:4-7)
 *   workspace : device, ppo2hip_rms_workspace_bytes(n, d) bytes */
size_t ppo2hip_rms_workspace_bytes(int64_t n, int64_t d);
int ppo2hip_rms_update(const void *x, int x_f64, int64_t n, int64_t d, double *state, void *workspace,
                       size_t workspace_bytes, void *stream);

/* The vector the MPI running statistics exchange, baselines/common/mpi_running_mean_std.py:37-42:
 * out = [ x.sum(0) | square(x).sum(0) | len(x) ] in float64 (x cast to float64 first).
 *   x [n, d] float32/float64, n >= 1;  zeros : device double[d] of zeros (the shift of the shared
 *   moments kernel);  out : device double[2d + 1];  workspace : ppo2hip_rms_workspace_bytes(n, d) */
int ppo2hip_column_moments(const void *x, int x_f64, int64_t n, int64_t d, const double *zeros, double *out,
                           void *workspace, size_t workspace_bytes, void *stream);
/* dst += src over n doubles: the assign_add of the all-reduced totals (:29-32). */
int ppo2hip_add_f64(double *dst, const double *src, int64_t n, void *stream);

/* clip((x - mean) / sqrt(var + epsilon), -clip, clip), the filter of VecNormalize._obfilt,
 * baselines/common/vec_env/vec_normalize.py:37; mean NULL = no centring (the reward filter, :31).
 *   x [n, d] float32/float64; mean, var : device double[d]; out [n, d] float32 or float64 */
int ppo2hip_normalize_clip(const void *x, int x_f64, int64_t n, int64_t d, const double *mean,
                           const double *var, double epsilon, double clip, void *out, int out_f64,
                           void *stream);

/* The reward half of VecNormalize.step_wait, baselines/common/vec_env/vec_normalize.py:27-31:
 * ret = ret*gamma + rews; ret_rms.update(ret); out = clip(rews / sqrt(ret_rms.var + epsilon)).
 *   rews [E]; ret : device double[E] in/out; state : device double[3] (mean, var, count) or NULL
 *   (ret=False: only ret is advanced, out = rews); workspace: ppo2hip_rms_workspace_bytes(E, 1) */
int ppo2hip_vecnorm_returns(const void *rews, int rews_f64, int64_t E, double gamma, double *ret,
                            double *state, double epsilon, double clip, void *out, int out_f64,
                            void *workspace, size_t workspace_bytes, void *stream);

/* VecFrameStack.step_wait, baselines/common/vec_env/vec_frame_stack.py:19-26, in place:
 * every pixel's nstack*channels values move ONE element towards the front (np.roll(shift=-1) on
 * the last axis: one frame when channels == 1, and the reference's behaviour as it stands for
 * channels > 1), envs flagged in `news` (uint8 [E], NULL = none) restart from zeros, `obs` fills
 * the last `channels` places.
 *   stacked [E, pixels, nstack*channels], obs [E, pixels, channels]; elem_bytes 1 (uint8) or 4 */
int ppo2hip_frame_stack(void *stacked, const void *obs, const uint8_t *news, int64_t E, int64_t pixels,
                        int channels, int nstack, int elem_bytes, void *stream);

/* ------------------------------------------------------------------------
 * Advantage post-processing
 * ---------------------------------------------------------------------- */

/* Fork-only scalarisation of the R-channel advantage,
 * baselines/ppo2/common_functions.py:359-401 (live terms only: the 0-weighted
 * advs_dir and the unused grad_v are not computed).
 *   values, advs_ori : float32 [N, R] (any row order)   max_return : HOST double[R]
 *   workspace : device, ppo2hip_compute_advs_workspace_bytes(N, R) bytes
 *   out : float64 [N] (the reference returns float64, :386) */
size_t ppo2hip_compute_advs_workspace_bytes(int64_t N, int R);
int ppo2hip_compute_advs(const float *values, const float *advs_ori, const double *max_return,
                         int64_t N, int R, void *workspace, size_t workspace_bytes, double *out,
                         void *stream);

/* Sums for the per-minibatch normalisation of Model.train,
 * baselines/ppo2/common_functions.py:67, for `nmb` consecutive minibatches of B rows each:
 * stats[3*j + 0..2] = {sum a, sum a^2, count} over rows idx[j*B .. (j+1)*B) in float64
 * (device).  `advs` is the rollout-wide array, float32 or float64 (adv_f64 != 0); rows are
 * picked through `idx` and (T, E) as described at ppo2hip_rollout (idx NULL = rows j*B + i).
 * workspace: ppo2hip_adv_stats_workspace_bytes(nmb) bytes. */
size_t ppo2hip_adv_stats_workspace_bytes(int nmb);
int ppo2hip_adv_stats(const void *advs, int adv_f64, const int32_t *idx, int64_t B, int nmb, int T,
                      int E, void *workspace, double *stats, void *stream);

/* ------------------------------------------------------------------------
 * Minibatch permutation (host side, bit-exact with NumPy's legacy RandomState)
 * ---------------------------------------------------------------------- */

/* MT19937 + legacy shuffle as np.random.seed / np.random.shuffle perform them at
 * baselines/ppo2/ppo2.py:113 and :130.  All pointers here are HOST pointers. */
typedef struct ppo2hip_rng ppo2hip_rng;
ppo2hip_rng *ppo2hip_rng_create(uint32_t seed);
void ppo2hip_rng_destroy(ppo2hip_rng *rng);
void ppo2hip_rng_seed(ppo2hip_rng *rng, uint32_t seed);
/* key: 624 words, pos in [0, 624] — the (key, pos) of np.random.get_state() */
int ppo2hip_rng_set_state(ppo2hip_rng *rng, const uint32_t *key, int pos);
int ppo2hip_rng_get_state(const ppo2hip_rng *rng, uint32_t *key, int *pos);
/* in-place Fisher-Yates over inds[0..n) */
int ppo2hip_shuffle_i64(ppo2hip_rng *rng, int64_t *inds, int64_t n);
int ppo2hip_shuffle_i32(ppo2hip_rng *rng, int32_t *inds, int64_t n);

/* The index lists of ALL epochs of one update, produced ahead of the GPU on worker
 * threads: out[e] (HOST int32[n], e.g. pinned) receives `inds` after the (e+1)-th in-place
 * shuffle of arange(n) — exactly the successive states of `inds` in the epoch loop of
 * baselines/ppo2/ppo2.py:110-113.  One job at a time per handle; the rng must not be used
 * by anything else between begin and wait_all (its state then equals NumPy's after the same
 * shuffles). */
typedef struct ppo2hip_perm ppo2hip_perm;
ppo2hip_perm *ppo2hip_perm_create(ppo2hip_rng *rng, int64_t n, int max_epochs);
void ppo2hip_perm_destroy(ppo2hip_perm *p);
int ppo2hip_perm_begin(ppo2hip_perm *p, int nepochs, int32_t *const *out);
/* Same job, but out[e] receives sigma_e = the (e+1)-th shuffle applied to arange(n) on its own;
 * the running order is inds_e = inds_{e-1}[sigma_e] (inds_0 = sigma_0), which
 * ppo2hip_compose_i32 evaluates on the device.  Takes the gather off the host's critical path. */
int ppo2hip_perm_begin_relative(ppo2hip_perm *p, int nepochs, int32_t *const *out);
int ppo2hip_perm_wait(ppo2hip_perm *p, int epoch);   /* blocks until out[epoch] is complete */
int ppo2hip_perm_wait_all(ppo2hip_perm *p);
/* Blocks until the running job has drawn all its swap targets: the generator has moved past the job (a second pipeline
 * on the same generator may begin the next update's job now), although the job's last swaps may still be running. */
int ppo2hip_perm_wait_drawn(ppo2hip_perm *p);
/* DEVICE pointers: out[k] = prev[sigma[k]] for k < n (out must not alias prev or sigma). */
int ppo2hip_compose_i32(const int32_t *prev, const int32_t *sigma, int32_t *out, int64_t n, void *stream);

/* Index lists are the caller's (the reference indexes NumPy arrays with them, baselines/ppo2/ppo2.py:117, and NumPy
 * raises IndexError on a bad one).  Every kernel that follows a list checks each entry where it uses it: an entry outside
 * [0, number of samples) is replaced by sample 0 and raises a bit of a per-GPU fault word instead of reading out of
 * bounds.  This call waits for `stream`, returns the word in *bits and clears it: 0 = every list used since the last
 * call was in range; PPO2HIP_FAULT_INDEX_RANGE = a minibatch index list (ppo_grad, ppo_epoch, adv_stats with T > 0);
 * PPO2HIP_FAULT_COMPOSE_RANGE = a sigma entry of ppo2hip_compose_i32.  Results computed from a flagged list are
 * meaningless; the Python layer raises Ppo2HipError(EINVAL) when it collects a non-zero word with the losses. */
#define PPO2HIP_FAULT_INDEX_RANGE 1u
#define PPO2HIP_FAULT_COMPOSE_RANGE 2u
#define PPO2HIP_FAULT_GRID_SYNC 4u      /* a launch whose workgroups hand values to each other (whole-sequence LSTM, the
                                         * optimiser step folded into ppo2hip_ppo_epoch) gave up waiting for its peers */
int ppo2hip_take_faults(uint32_t *bits, void *stream);
/* The same read-and-clear queued on `stream` WITHOUT waiting: a one-thread kernel stores the word into *host_bits
 * (page-locked HOST memory the device can address, e.g. from hipHostMalloc, that stays valid until the kernel has run;
 * the caller records an event behind this call and reads the word once the event has completed).  For callers that keep several updates in flight and cannot stall the stream per update; calls on
 * one stream at a time. */
int ppo2hip_post_faults(uint32_t *host_bits, void *stream);

/* ------------------------------------------------------------------------
 * Policy description and parameter arena
 * ---------------------------------------------------------------------- */

enum { PPO2HIP_POLICY_MLP = 0, PPO2HIP_POLICY_CNN = 1, PPO2HIP_POLICY_CNN_LSTM = 2, PPO2HIP_POLICY_CNN_LNLSTM = 3 };
enum { PPO2HIP_DIST_GAUSSIAN = 0, PPO2HIP_DIST_CATEGORICAL = 1 };
enum { PPO2HIP_ORDER_FORK = 0, PPO2HIP_ORDER_UPSTREAM = 1 };

/* Which network of the reference the arena holds.
 *   MLP:  two separate tanh trunks (actor, value): fork MlpPolicy
 *         baselines/ppo2/policies.py:132-181 (hidden 256,128, nvalue 5, order FORK) or the
 *         upstream-shaped one baselines/a2c/policies.py:116-146 (hidden 64,64, nvalue 1, order UPSTREAM).
 *   CNN*: nature_cnn + heads, baselines/ppo2/policies.py:6-130 (ob = H,W,C uint8 NHWC).
 * Parameters live in ONE flat float32 arena in the reference's variable creation
 * order (= checkpoint order of Model.save, baselines/ppo2/common_functions.py:115-119). */
typedef struct ppo2hip_policy_desc {
    int32_t kind;        /* PPO2HIP_POLICY_* */
    int32_t dist;        /* PPO2HIP_DIST_* */
    int32_t order;       /* PPO2HIP_ORDER_* (MLP only) */
    int32_t ob_dim;      /* MLP: flattened observation length */
    int32_t ob_h, ob_w, ob_c; /* CNN: NHWC image */
    int32_t hidden1, hidden2; /* MLP trunk widths */
    int32_t ac_dim;      /* Gaussian: action length; Categorical: number of actions */
    int32_t nvalue;      /* value-head outputs R */
    int32_t nlstm;       /* LSTM width (CNN_LSTM / CNN_LNLSTM) */
} ppo2hip_policy_desc;

/* Loss/optimiser variant of baselines/ppo2/common_functions.py:
 *   upstream = 0 : live fork code :26-54 — vf_loss = mean((v-R)^2) over [B,R]; actor and value
 *                  objectives minimised by separate, identically configured Adams; no clipping
 *                  (vf_coef and max_grad_norm ignored, as in the fork).
 *   upstream = 1 : the commented upstream code :30-33,:56-63 — vf_loss = .5*mean(max((v-R)^2,
 *                  (vclipped-R)^2)); loss = pg - ent_coef*H + vf_coef*vf; clip_by_global_norm. */
typedef struct ppo2hip_loss_desc {
    float ent_coef;
    float vf_coef;
    float max_grad_norm; /* <= 0: no clipping */
    int32_t upstream;
} ppo2hip_loss_desc;

/* number of float32 parameters, or < 0 on a bad description */
int64_t ppo2hip_param_count(const ppo2hip_policy_desc *desc);
/* Variable table in creation order: name (<= 31 chars), offset and shape in the arena.
 * Returns the number of variables (writes at most `max_vars` entries), < 0 on error.
 * HOST pointers. shapes: 4 int64 per variable, unused dims = 0. */
int ppo2hip_param_layout(const ppo2hip_policy_desc *desc, int max_vars, char (*names)[32],
                         int64_t *offsets, int64_t *shapes);

/* ------------------------------------------------------------------------
 * Forward, gradient and optimiser steps
 * ---------------------------------------------------------------------- */

/* Where a minibatch's rows come from.  Arrays are the time-major rollout buffers
 * ([T, E, ...], env fastest); a minibatch row i (0 <= i < B) is flat env-major sample
 * s = idx[i] (or s = i when idx is NULL), i.e. env = s / T, t = s % T — the indexing
 * that `arr[mbinds]` applies to the sf01-flattened arrays at baselines/ppo2/ppo2.py:117
 * and :134-136 — read in place at [t, env].  T = 0 means the arrays are already
 * flat [N, ...] and s indexes them directly (what Model.train receives, :65).
 * n = rows held by every array (T * E when T > 0): the range idx values must stay in. */
typedef struct ppo2hip_rollout {
    const void *obs;         /* float32 [.., ob_dim] or uint8 [.., H, W, C] */
    const void *actions;     /* float32 [.., ac_dim] (Gaussian) or int32 [..] (Categorical) */
    const float *returns;    /* [.., nvalue] */
    const float *values;     /* [.., nvalue]  (OLDVPRED) */
    const float *neglogpacs; /* [..]          (OLDNEGLOGPAC) */
    const void *advs;        /* [..] float32, or float64 when adv_f64 != 0 (un-normalised) */
    const uint8_t *masks;    /* [..] done-before-step flags; recurrent policies only */
    const float *states;     /* [E, 2*nlstm] initial LSTM states of the rollout; recurrent only */
    int32_t adv_f64;
    int32_t T, E;
    int64_t n;
} ppo2hip_rollout;

size_t ppo2hip_forward_workspace_bytes(const ppo2hip_policy_desc *desc, int64_t B);
/* Policy forward for Runner.run's model.step / model.value
 * (baselines/ppo2/policies.py:161-172): obs [B, ...] -> pi [B, ac_dim] (mean or logits),
 * vf [B, nvalue].  For recurrent policies B = nenv (one step): masks [B], states_in/out [B, 2*nlstm]. */
int ppo2hip_policy_forward(const ppo2hip_policy_desc *desc, const float *params, const void *obs,
                           int64_t B, const uint8_t *masks, const float *states_in,
                           float *pi, float *vf, float *states_out, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Per-row distribution terms for Runner.run (baselines/common/distributions.py:202-205,
 * :141-148): neglogp of `actions` under pi (and logstd from the arena). */
int ppo2hip_neglogp(const ppo2hip_policy_desc *desc, const float *params, const float *pi,
                    const void *actions, int64_t B, float *neglogp, void *stream);

size_t ppo2hip_grad_workspace_bytes(const ppo2hip_policy_desc *desc, int64_t B);
/* Exposed for tests: where ppo2hip_ppo_grad leaves the rectified activations of the nature_cnn trunk
 * (baselines/ppo2/policies.py:12-16) in its workspace after a call with the same (desc, B, nsteps_mb).
 *   name : "c1" [B,OH1,OW1,32], "c2" [B,OH2,OW2,64], "c3" [B, flat] (NHWC flatten), "fc1" [B,512]; float32
 *   offset_bytes, nelems : HOST out.  A parity test reads the on/off pattern of the rectifiers from them:
 *   a float64 restatement disagrees with any float32 evaluation on the few units within rounding of zero. */
int ppo2hip_grad_workspace_region(const ppo2hip_policy_desc *desc, int64_t B, int32_t nsteps_mb, const char *name,
                                  size_t *offset_bytes, int64_t *nelems);

/* Action sampling with its neglogp: `a0 = pd.sample(); neglogp0 = pd.neglogp(a0)` of the act graph,
 * baselines/ppo2/policies.py:168-170 (fork) and baselines/a2c/policies.py:137-139; Gaussian
 * `mean + std * N(0,1)` (baselines/common/distributions.py:211-212), Categorical Gumbel-max
 * `argmax(logits - log(-log(U)))` (:164-166).  The reference draws the noise inside TensorFlow;
 * here it is Philox4x32-10 keyed by `seed`, counter (row, element group, step), so a (seed, step)
 * pair reproduces a rollout step's noise on any launch geometry and no generator state is kept.
 *   pi : float32 [B, ac_dim] mean or logits (ppo2hip_policy_forward's output)
 *   deterministic != 0 : the mode (`step_max`, baselines/ppo2/policies.py:172-174)
 *   actions : float32 [B, ac_dim] (Gaussian) or int32 [B] out      neglogp : float32 [B] out
 *   noise_out : float32 [B, ac_dim] or NULL — the N(0,1) / U(0,1) draws that were used */
int ppo2hip_sample_actions(const ppo2hip_policy_desc *desc, const float *params, const float *pi, int64_t B,
                           uint64_t seed, uint64_t step, int deterministic, void *actions, float *neglogp,
                           float *noise_out, void *stream);

/* One act step of the rollout loop, `model.step(obs, states, dones)` of
 * baselines/ppo2/common_functions.py:249-251: policy forward, action sampling and neglogp in one
 * call, each output written where the caller points (a row of the time-major rollout arena).
 * Arguments as for ppo2hip_policy_forward and ppo2hip_sample_actions; the distribution
 * parameters stay in the workspace (ppo2hip_act_workspace_bytes(desc, B) bytes). */
size_t ppo2hip_act_workspace_bytes(const ppo2hip_policy_desc *desc, int64_t B);
int ppo2hip_act_step(const ppo2hip_policy_desc *desc, const float *params, const void *obs, int64_t B,
                     const uint8_t *masks, const float *states_in, uint64_t seed, uint64_t step,
                     int deterministic, void *actions, float *values, float *neglogp, float *states_out,
                     void *workspace, size_t workspace_bytes, void *stream);

/* The fork's bootstrap of finished envs off the value of the reset observation,
 * baselines/ppo2/common_functions.py:265-276: where dones[e], rewards[e] += gamma * v_reset[e] on
 * every channel if rewards[e][0] > 0, else on channels 2.. only.
 *   rewards : float32 [E, R] in/out    dones : uint8 [E]    v_reset : float32 [E, R]
 *   gamma : HOST double[R], R <= 8 */
int ppo2hip_done_bootstrap(float *rewards, const uint8_t *dones, const float *v_reset, const double *gamma,
                           int64_t E, int R, void *stream);

/* One Model.train graph evaluation up to (not including) the optimiser,
 * baselines/ppo2/common_functions.py:26-44 + the tf.gradients of :53-54/:57:
 * gathers the minibatch, normalises advantages with `stats` (from ppo2hip_adv_stats,
 * possibly all-reduced across ranks; NULL = advs already normalised), runs forward, loss
 * and backward.
 *   grads  : float32 [P] out — sum over the B rows of d(objective)/d(param) * inv_count terms,
 *            where the batch means of the loss use `inv_count` in place of 1/B (pass 1/B_global
 *            on every rank of a data-parallel job and SUM-all-reduce `grads` and `losses`).
 *   losses : float32 [8] out — {policy_loss, value_loss, policy_entropy, approxkl, clipfrac}
 *            (loss_names order, :107) scaled the same way; [5..7] reserved.
 * For recurrent policies B = nenvs_mb * nsteps_mb rows in env-major order and
 * `mb_envs` (int32 [B / nsteps_mb], device) names their envs for the state gather (:136). */
int ppo2hip_ppo_grad(const ppo2hip_policy_desc *desc, const ppo2hip_loss_desc *loss,
                     const float *params, const ppo2hip_rollout *rollout, const int32_t *idx,
                     int64_t B, const int32_t *mb_envs, int32_t nsteps_mb, const double *stats,
                     float cliprange, double inv_count, float *grads, float *losses,
                     void *workspace, size_t workspace_bytes, void *stream);

/* clip_by_global_norm (upstream, baselines/ppo2/common_functions.py:59-60) + Adam in the
 * arithmetic of baselines/common/mpi_adam.py:30-35 (epsilon 1e-5 as at
 * common_functions.py:51-52), one pass over the flat arenas.
 *   grad_scale : multiplies grads first (1/size of MpiAdam.update, mpi_adam.py:27-28)
 *   t          : step count AFTER this update (1 for the first)
 *   max_grad_norm <= 0 : no clipping
 *   scratch    : device, ppo2hip_adam_scratch_bytes(P); scratch[0] (float) = global norm after the call */
size_t ppo2hip_adam_scratch_bytes(int64_t P);
int ppo2hip_adam_step(float *params, const float *grads, float *m, float *v, int64_t P, double lr,
                      int64_t t, double beta1, double beta2, double epsilon, double grad_scale,
                      double max_grad_norm, void *scratch, void *stream);

/* ppo2hip_adam_step for callers that cannot allocate or pass device scratch (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  The norm header and partial sums live in a PPO2HIP_ADAM_STATIC_SCRATCH_BYTES-byte array
 * that is part of the library's device image (every GPU the library runs on holds one until the process
 * ends); the call allocates and frees nothing.  Arguments and arithmetic are those of ppo2hip_adam_step.
 *   P        : at most PPO2HIP_ADAM_STATIC_MAX_PARAMS, else PPO2HIP_EINVAL
 *   norm_out : device float[2] or NULL — {global gradient norm (written when clipping), applied scale}
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (different
 * streams or threads) race; calls on one stream, or otherwise ordered, are safe.  Every call rewrites
 * what it later reads, so the previous call's contents never matter. */
#define PPO2HIP_ADAM_STATIC_SCRATCH_BYTES 2304
#define PPO2HIP_ADAM_STATIC_MAX_PARAMS (1ll << 26)
int ppo2hip_adam_step_static(float *params, const float *grads, float *m, float *v, int64_t P, double lr,
                             int64_t t, double beta1, double beta2, double epsilon, double grad_scale,
                             double max_grad_norm, float *norm_out, void *stream);

/* One epoch's minibatch loop (baselines/ppo2/ppo2.py:114-118) of a feed-forward policy in one
 * call, single process: for j < N / nbatch_train, ppo2hip_ppo_grad on idx[j*nbatch_train ..),
 * normalised with stats[j] ({sum, sumsq, count} rows of ppo2hip_adv_stats, or NULL), losses to
 * losses[j*8 ..), then ppo2hip_adam_step number t_first + j with grad_scale 1.  Saves the host the
 * per-minibatch call overhead; the data-parallel path keeps the two calls apart because its
 * gradient all-reduce goes between them. */
int ppo2hip_ppo_epoch(const ppo2hip_policy_desc *desc, const ppo2hip_loss_desc *loss, float *params,
                      float *grads, float *adam_m, float *adam_v, const ppo2hip_rollout *rollout,
                      const int32_t *idx, int64_t N, int64_t nbatch_train, const double *stats,
                      float cliprange, double lr, int64_t t_first, double beta1, double beta2,
                      double epsilon, double max_grad_norm, float *losses, void *workspace,
                      size_t workspace_bytes, void *adam_scratch, void *stream);

/* ------------------------------------------------------------------------
 * Data-parallel exchange (one process per GPU, RCCL over xGMI)
 * ---------------------------------------------------------------------- */

/* The communicator that stands where MpiAdam holds `MPI.COMM_WORLD`, baselines/common/mpi_adam.py:19.
 * Rank 0 draws a unique id (HOST, PPO2HIP_UNIQUE_ID_BYTES bytes) and the host code hands it to every rank by
 * whatever channel it has (a torch.distributed store, a file, MPI); each rank then calls ppo2hip_comm_create
 * with ITS GPU current.  unique_id NULL is allowed for size 1: no collective library is touched.  Returns NULL
 * on failure (ppo2hip_last_error() says why).  One communicator per process; calls on it are not re-entrant. */
#define PPO2HIP_UNIQUE_ID_BYTES 128
typedef struct ppo2hip_comm ppo2hip_comm;
int ppo2hip_comm_unique_id(void *id_out);
ppo2hip_comm *ppo2hip_comm_create(const void *unique_id, int rank, int size);
void ppo2hip_comm_destroy(ppo2hip_comm *comm);
int ppo2hip_comm_rank(const ppo2hip_comm *comm);
int ppo2hip_comm_size(const ppo2hip_comm *comm);

/* `comm.Allreduce(localg, globalg, op=MPI.SUM)` of MpiAdam.update, baselines/common/mpi_adam.py:24-26, in place on
 * the flat float32 gradient arena (DEVICE), queued on `stream`; the division by the number of ranks (:27-28) is
 * the `grad_scale` of ppo2hip_adam_step.  comm NULL or of size 1: nothing to do. */
int ppo2hip_allreduce_grads(ppo2hip_comm *comm, float *buf, int64_t n, void *stream);
/* SUM all-reduce of float64 totals in place (DEVICE): the moment vectors [sum | sumsq | count] of
 * baselines/common/mpi_moments.py:12-17 and baselines/common/mpi_running_mean_std.py:37-43, and the
 * {sum a, sum a^2, n} rows of ppo2hip_adv_stats when every rank draws the same global minibatch. */
int ppo2hip_allreduce_f64(ppo2hip_comm *comm, double *buf, int64_t n, void *stream);
/* `comm.Bcast(theta, root=0)` of MpiAdam.sync, baselines/common/mpi_adam.py:37-40, in place (DEVICE). */
int ppo2hip_bcast_params(ppo2hip_comm *comm, float *buf, int64_t n, int root, void *stream);
/* MpiAdam.check_synced, baselines/common/mpi_adam.py:42-50 (every rank's parameters must equal rank 0's exactly),
 * without moving the parameters: each rank digests the bits of its arena into two 64-bit sums, the ranks MAX-reduce
 * {h, ~h}, and *synced (DEVICE int32) becomes 1 when max == min for both digests, else 0.
 *   workspace : DEVICE, 32 bytes, 8-byte aligned */
int ppo2hip_replica_checksum(ppo2hip_comm *comm, const float *params, int64_t n, void *workspace, int32_t *synced,
                             void *stream);

/* ppo2hip_ppo_epoch for a data-parallel job: between each minibatch's gradient and its optimiser step the gradient
 * arena is SUM-all-reduced over `comm` and scaled by `grad_scale` (1/size: MpiAdam.update, baselines/common/
 * mpi_adam.py:24-28), all queued on `stream` with no host synchronisation, so an epoch stays one library call.
 * Every rank must make the same sequence of calls (same N / nbatch_train). */
int ppo2hip_ppo_epoch_dp(const ppo2hip_policy_desc *desc, const ppo2hip_loss_desc *loss, float *params,
                         float *grads, float *adam_m, float *adam_v, const ppo2hip_rollout *rollout,
                         const int32_t *idx, int64_t N, int64_t nbatch_train, const double *stats,
                         float cliprange, double lr, int64_t t_first, double beta1, double beta2,
                         double epsilon, double max_grad_norm, float *losses, void *workspace,
                         size_t workspace_bytes, void *adam_scratch, ppo2hip_comm *comm, double grad_scale,
                         void *stream);

/* Env-sharded data parallelism in which every rank draws the SAME global minibatch order (the np.random stream of
 * baselines/ppo2/ppo2.py:110-116 over all ranks' samples) and works on the entries whose env it owns, so that N ranks
 * reproduce the single-process update up to summation order.  A rank owns a contiguous block of envs, i.e. the
 * env-major flat sample numbers [lo, hi); its own sample numbers are s - lo.  All pointers DEVICE unless marked.
 *
 * ppo2hip_filter_owned_i32: idx holds nmb global minibatches of nbatch_train entries back to back.  For minibatch j
 * the owned entries, re-based, in list order, go to out[j*out_stride ...] and their number to counts[j]; up to 64
 * slots behind them are set to sample 0.  out_stride >= the largest share expected (nbatch_train always suffices);
 * a larger share raises PPO2HIP_FAULT_INDEX_RANGE and is cut.
 *   workspace : ppo2hip_filter_owned_workspace_bytes(nbatch_train, nmb) */
size_t ppo2hip_filter_owned_workspace_bytes(int64_t nbatch_train, int nmb);
int ppo2hip_filter_owned_i32(const int32_t *idx, int64_t nbatch_train, int nmb, int32_t lo, int32_t hi, int32_t *out,
                             int64_t out_stride, int32_t *counts, void *workspace, size_t workspace_bytes, void *stream);
/* ppo2hip_adv_stats over this rank's share of each global minibatch: stats[j] = {sum a, sum a^2, entries owned}
 * (float64, fixed summation order) with advs the rank's own time-major [T, E] array; SUM-all-reduced over the ranks
 * (ppo2hip_allreduce_f64) the rows are the statistics of baselines/ppo2/common_functions.py:67 over the whole minibatch.
 *   workspace : ppo2hip_adv_stats_owned_workspace_bytes(nmb) */
size_t ppo2hip_adv_stats_owned_workspace_bytes(int nmb);
int ppo2hip_adv_stats_owned(const void *advs, int adv_f64, const int32_t *idx, int64_t nbatch_train, int nmb, int32_t lo,
                            int32_t hi, int T, int E, void *workspace, double *stats, void *stream);
/* An epoch over the shares: for minibatch j, ppo2hip_ppo_grad on idx[j*idx_stride ...] with counts[j] (HOST int64,
 * 0 allowed: the rank contributes zeros) entries, means taken with `inv_count` (1 / the GLOBAL minibatch size) and the
 * all-reduced `stats` rows, then the SUM all-reduce of the gradient arena over `comm` and the optimiser step
 * (gradient scale 1): gradient sums, not per-rank means, are what the ranks add.  losses [nmb, 8]: each rank's rows are
 * its shares of the global means.  Every rank makes the same call with the same nmb. */
int ppo2hip_ppo_epoch_shares(const ppo2hip_policy_desc *desc, const ppo2hip_loss_desc *loss, float *params, float *grads,
                             float *adam_m, float *adam_v, const ppo2hip_rollout *rollout, const int32_t *idx,
                             int64_t idx_stride, const int64_t *counts, int nmb, const double *stats, float cliprange,
                             double inv_count, double lr, int64_t t_first, double beta1, double beta2, double epsilon,
                             double max_grad_norm, float *losses, void *workspace, size_t workspace_bytes,
                             void *adam_scratch, ppo2hip_comm *comm, void *stream);

/* out[c] = scale * sum_r rows[r*stride + c] for c < ncols (<= 16): the mean of the per-minibatch
 * loss rows that learn reports, baselines/ppo2/ppo2.py:139 (scale = 1/nrows; data-parallel runs
 * fold 1/size in and SUM-all-reduce the result).  float64 sums, fixed order. */
int ppo2hip_row_mean(const float *rows, int nrows, int stride, int ncols, double scale, float *out, void *stream);

/* Plain fp32 GEMM on the MFMA path, exposed for tests: C[M,N] = A[M,K] * B[K,N] (row-major). */
int ppo2hip_gemm_f32(const float *A, const float *B, float *C, int M, int N, int K, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PPO2HIP_H */
