/*
 * gpmpc_hip.h - C-ABI of libgpmpc_hip.so: the MI355X (gfx950) implementation of the sampling-gpmpc
 * GP-posterior-sample rollout hot path.
 *
 * The reference (manish-pra/sampling-gpmpc) has no native/FFI layer: its boundary for this path is the Python
 * object `Agent` (+ `Agent.model_i`), whose arithmetic is delegated to gpytorch.  Each entry point below replaces
 * the stack of gpytorch/torch calls behind one reference interface; the Python facade
 * (sampling_gpmpc_amd/agent.py, same method names and shapes as reference src/agent.py) binds them via ctypes.
 *
 * Conventions
 *   - plain C, no torch types; every pointer marked [dev] is a device (HBM) pointer owned by the caller,
 *     [host] is ordinary host memory; all floating point is IEEE binary64 (the reference runs
 *     torch.set_default_dtype(float64), src/agent.py:15).
 *   - no hidden allocation: callers query *_workspace_bytes() and pass the workspace.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on it.
 *   - return value: 0 = launched OK, negative = GPMPC_E_* (see gpmpc_last_error_string()).
 *   - per-batch-element numerical status comes back in [dev] int32 `info` arrays (bit field GPMPC_INFO_*), so the
 *     facade can reproduce gpytorch's NumericalWarning / NotPSDError behaviour.
 *   - tensor layouts are the reference's (row-major / C-contiguous) unless a stride argument says otherwise.
 */
#ifndef GPMPC_HIP_H
#define GPMPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPMPC_ABI_VERSION 9

#define GPMPC_MAX_NY 4   /* GP outputs            (reference agent.g_dim.ny : 1 pendulum1D, 3 car)          */
#define GPMPC_MAX_D  4   /* GP input dimension    (g_nx + g_nu : 2 in all shipped configs)                 */
#define GPMPC_MAX_T  5   /* label slots per point (1 value-only, 1 + D value + gradient)                   */
#define GPMPC_MAX_NX 8   /* full state dimension  (2 pendulum1D, 4 car)                                    */
#define GPMPC_MAX_NU 4   /* input dimension       (1 pendulum1D, 2 car)                                    */

/* error codes */
#define GPMPC_OK            0
#define GPMPC_E_ARG        -1   /* inconsistent / unsupported argument                                     */
#define GPMPC_E_WORKSPACE  -2   /* workspace too small                                                     */
#define GPMPC_E_HIP        -3   /* a HIP runtime call failed (string has the hipError)                     */
#define GPMPC_E_UNSUPPORTED -4  /* size outside what the kernels were instantiated for                     */

/* info bits (per batch element) */
#define GPMPC_INFO_TRAIN_CHOL_FAIL   0x0001 /* non-positive pivot while factorising K_oo + Sigma (A.5)      */
#define GPMPC_INFO_ROOT_JITTER_MASK  0x000e /* (info >> 1) & 7 = highest retry level reached, 0 = none      */
#define GPMPC_INFO_ROOT_FAIL         0x0010 /* all 3 jitter retries failed for THIS chain; after an eigh redraw of the
                                             * batch under GPMPC_ROOT_AUTO: set for EVERY chain, with retry level 3 - the
                                             * batch's outcome (chains stop their own attempts once another one has failed
                                             * for good, so per-chain levels would be a matter of timing)               */
#define GPMPC_INFO_VAR_CLAMPED       0x0020 /* a posterior variance was raised to the 1e-10 floor (A.8)     */
#define GPMPC_INFO_NEG_1x1           0x0040 /* 1x1 covariance negative -> sqrt gives NaN (as gpytorch)      */
#define GPMPC_INFO_ROOT_EIGH         0x0080 /* y was drawn with the eigendecomposition root (A.7 step 4)    */
#define GPMPC_INFO_EIGH_NOCONV       0x0100 /* the Jacobi eigensolver hit its sweep limit (result still used) */
#define GPMPC_INFO_STATE_FULL        0x0200 /* gpmpc_rollout_seeded: the factor state had no room for a new point  */

/* root_mode of gpmpc_joint_sample */
#define GPMPC_ROOT_AUTO      0   /* gpytorch: Cholesky with the jitter chain; if ANY chain of the batch fails all  */
                                 /* three retries, the WHOLE batch is drawn with the eigendecomposition root       */
#define GPMPC_ROOT_EIGH      1   /* eigendecomposition root for every chain (a sharded batch whose failing chain   */
                                 /* lives on another rank; tests)                                                  */
#define GPMPC_ROOT_CHOLESKY  2   /* never fall back: failing chains return NaN samples + GPMPC_INFO_ROOT_FAIL      */

/* environment ids: the per-step maps of reference src/environments/{pendulum1D,car_model_residual}.py        */
#define GPMPC_ENV_PENDULUM1D   0
#define GPMPC_ENV_CAR_RESIDUAL 1

/* rollout modes */
#define GPMPC_MODE_INDEPENDENT   0   /* "I": every step conditions on the real data only                    */
#define GPMPC_MODE_RECONDITIONED 1   /* "R": every step conditions on real + the sample's own previous draws */

/*
 * The GP definition: replaces reference src/GP_model.py:94-143 (BatchMultitaskGPModelWithDerivatives_fromParams:
 * zero mean, ScaleKernel(RBFKernel[Grad]) with per-output ARD lengthscales / outputscale, and the
 * MultitaskGaussianLikelihood(rank=0) noise of src/agent.py:235-240).
 */
typedef struct gpmpc_gp_desc {
    int32_t g_ny;                 /* number of independent GP outputs                                      */
    int32_t D;                    /* GP input dimension                                                    */
    int32_t T;                    /* tasks: 1 (use_grad=False) or 1 + D (value + gradient)                  */
    int32_t N_r;                  /* number of real training points (shared by all samples)                */
    int32_t real_has_grad;        /* 0: real labels observe task 0 only (NaN elsewhere); 1: all T tasks    */
    int32_t grid_n0, grid_n1;     /* > 0: X_r is the tensor-product grid meshgrid(axis0[n0], axis1[n1], "ij") the  */
                                  /* reference builds (pendulum1D.py:36-47, car_model_residual.py:37-50), row    */
                                  /* i = a*n1 + c; enables the separable-kernel fast path.  0/0: unstructured.    */
    int32_t _pad;
    double  ell[GPMPC_MAX_NY][GPMPC_MAX_D];   /* Dyn_gp_lengthscale.both[o][d]                              */
    double  outputscale[GPMPC_MAX_NY];        /* Dyn_gp_outputscale.both[o]                                 */
    double  noise[GPMPC_MAX_T];               /* task_noises.val[t] * multiplier + Dyn_gp_noise             */
    double  jitter;                           /* Dyn_gp_jitter (gpytorch.settings.cholesky_jitter)          */
    double  var_floor;                        /* gpytorch.settings.min_variance (1e-10 for FP64)            */
} gpmpc_gp_desc_t;

/*
 * The per-step environment maps + feedback law of the forward-sampling loop: replaces, fused into the rollout
 * kernel, reference src/environments/pendulum1D.py:165-188 / car_model_residual.py:132-161,211-224 as composed by
 * src/agent.py:532-557 (dyn_fg_jacobians, value column) and benchmarking/simulate_forward_sampling_car.py:121-136.
 */
typedef struct gpmpc_env_desc {
    int32_t env_id;               /* GPMPC_ENV_*                                                           */
    int32_t nx, nu;
    int32_t use_feedback;         /* u = u_ff + K (x - x_goal)   (agent.feedback.use)                       */
    double  dt;
    double  p0, p1;               /* pendulum1D: l, g ; car: lf, lr (only used by the true plant, not here) */
    double  K[GPMPC_MAX_NU][GPMPC_MAX_NX];    /* optimizer.terminal_tightening.K                            */
    double  x_goal[GPMPC_MAX_NX];             /* env.goal_state                                             */
} gpmpc_env_desc_t;

/* ------------------------------------------------------------------------------------------------------------ */
int         gpmpc_abi_version(void);
const char* gpmpc_last_error_string(void);
/* fills name (<= cap bytes), CU count, LDS bytes per workgroup of device `dev`; used by bench.py for the roofline */
int         gpmpc_device_info(int dev, char* name /*[host]*/, int cap, int* cu_count, int* lds_bytes);
/* runs a one-wave kernel checking the cross-lane primitives (DPP reduction, readlane broadcast, small Cholesky)
 * against in-kernel references; synchronises `stream`.  0 = OK. */
int         gpmpc_selftest(void* stream);

/*
 * gpmpc_plan_build - factorise the shared real-data block once.
 * Replaces: the train-side half of gpytorch's ExactGP prediction strategy that the reference re-does from scratch
 * on every call (src/agent.py:241-250 builds the model, src/agent.py:640 triggers K_oo + Sigma -> Cholesky -> alpha;
 * SURVEY.md App. A.3-A.5), restricted to the real data, which are identical for every sample
 * (src/agent.py:204-214 tiles them Ns times).
 *   X_r   [dev] (N_r, D)            Dyn_gp_X_train
 *   Y_r   [dev] (g_ny, N_r, T)      Dyn_gp_Y_train (NaN in unobserved slots)
 *   plan  [dev] gpmpc_plan_bytes()  out: per output L_rr, L_rr^-1 (transposed), w_r = L^-1 y, alpha_r
 *   info  [dev] (g_ny) int32        out: GPMPC_INFO_TRAIN_CHOL_FAIL
 */
size_t gpmpc_plan_bytes(const gpmpc_gp_desc_t* gp);
int    gpmpc_plan_build(const gpmpc_gp_desc_t* gp, const double* X_r, const double* Y_r,
                        void* plan, int32_t* info, void* stream);

/*
 * gpmpc_rollout - H-step forward rollout of Ns sampled dynamics functions, whole horizon in one launch.
 * Replaces: the loop of reference benchmarking/simulate_forward_sampling_car.py:117-138 (and the equivalent loops
 * simulate_true_reachable_set.py:179-258, src/agent.py:362-415), i.e. per step: train_hallucinated_dynGP
 * (src/agent.py:216-272) -> get_batch_x_hat_u_diff (480-501) -> dyn_fg_jacobians value column (532-557) ->
 * sample_gp (629-708: model_i(x), .sample(base_samples), optional variance-is-zero replacement, beta clip) ->
 * update_hallucinated_Dyn_dataset (164-198, min-dist filter off) -> state hand-over.
 *   mode          GPMPC_MODE_*                  (I: use_model_without_derivatives=True as shipped; R otherwise)
 *   hall_tasks    label slots observed at appended points: T (sample_gp path) or 1 (value-only, src/agent.py:402)
 *   var_zero_thr  Dyn_gp_variance_is_zero (< 0 disables, src/agent.py:646)
 *   beta          Dyn_gp_beta (clip to mean +- beta sqrt(var), src/agent.py:701-708)
 *   x0      [dev] (Ns, nx) if x0_per_sample else (nx)
 *   u_ff    [dev] (H, nu)                       open-loop input sequence (input_traj[-1] of the reference's data.pkl)
 *   z       [dev] base samples; element (t, s, o, b) at z[t*z_step_stride + ((s*g_ny)+o)*T + b]
 *                 (the reference's epistimic_random_vector[t][1] slab: pass &erv[0][1] and
 *                  z_step_stride = n_itrs*Ns*g_ny*T)
 *   X_traj  [dev] (Ns, nx, H+1)                 out: the reachable tube (reference X_traj, :115,133,138)
 *   Y       [dev] (Ns, g_ny, H, T) or NULL      out: the clipped samples (what the reference appends as labels)
 *   Xi      [dev] (Ns, H, D) or NULL            out: GP inputs per step (the reference's Hallcinated_X_train rows)
 *   info    [dev] (Ns) int32                    out: OR of GPMPC_INFO_* over steps and outputs
 *   ws      [dev] gpmpc_rollout_workspace_bytes(...) scratch (per-sample factor storage when it exceeds LDS)
 */
size_t gpmpc_rollout_workspace_bytes(const gpmpc_gp_desc_t* gp, int32_t mode, int32_t hall_tasks,
                                     int64_t Ns, int32_t H);
/*
 * Kernel choice of gpmpc_rollout.  The dispatcher picks by shape AND launch size: GPMPC_KERNEL_ONE (one chain per wave, the
 * factor in AGPR-pinned MFMA panels: pendulum1D 4 x 9 grid, H <= 30, up to 2048 chains), GPMPC_KERNEL_TILES (four chains per
 * wave: larger launches), GPMPC_KERNEL_FAST (one chain per wave on the VALU), GPMPC_KERNEL_INDEP (mode I), GPMPC_KERNEL_GENERIC.
 * The kernels sum in different orders: a sample's trajectory is bit-identical between two launches ONLY if both ran the same
 * kernel (then it is independent of what else is in the launch).  A caller that needs bit-equality across launch sizes - a
 * sample-sharded run against the single-GPU run of the same samples - pins the kernel: gpmpc_rollout_pin_kernel(k) makes
 * every later gpmpc_rollout of this process take kernel k where the shape allows it (else the generic kernel),
 * GPMPC_KERNEL_AUTO restores the size heuristic.  Returns the previous pin.  gpmpc_rollout_last_kernel(): what the last
 * launch of this process ran.  (Environment: GPMPC_ROLLOUT_ONE=0/1, GPMPC_ROLLOUT_TILES=0/1, GPMPC_DISABLE_FAST_ROLLOUT=1.)
 */
#define GPMPC_KERNEL_AUTO    (-1)
#define GPMPC_KERNEL_GENERIC 0
#define GPMPC_KERNEL_FAST    1
#define GPMPC_KERNEL_INDEP   2
#define GPMPC_KERNEL_TILES   3
#define GPMPC_KERNEL_ONE     4
int    gpmpc_rollout_pin_kernel(int32_t kernel);
int    gpmpc_rollout_last_kernel(void);
/* the kernel gpmpc_rollout would take for this shape and launch size under the current pin (GPMPC_KERNEL_AUTO: bad descriptor) */
int    gpmpc_rollout_kernel_for(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int32_t mode, int32_t hall_tasks,
                                int64_t Ns, int32_t H);
int    gpmpc_rollout(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, const void* plan,
                     const double* X_r, int32_t mode, int32_t hall_tasks, double var_zero_thr, double beta,
                     int64_t Ns, int32_t H,
                     const double* x0, int32_t x0_per_sample, const double* u_ff,
                     const double* z, int64_t z_step_stride,
                     double* X_traj, double* Y, double* Xi, int32_t* info,
                     void* ws, size_t ws_bytes, void* stream);

/*
 * gpmpc_rollout_seeded - gpmpc_rollout whose chains START from given conditioning points and / or keep their factor.
 * Replaces, in addition to gpmpc_rollout: the reference loop's behaviour that train_hallucinated_dynGP(1) never resets
 * (a second rollout on the same Agent conditions on the points already there, benchmarking/
 * simulate_forward_sampling_car.py:118), and the forward sampling of src/agent.py:362-415 (prepare_dynamics_set), which
 * conditions on real + hallucinated + its own value-only draws.  SURVEY.md section 8b: the "final factor state" output.
 *   X_h0  [dev] (Ns, g_ny, n_h0, D), Y_h0 [dev] (Ns, g_ny, n_h0, T)   seed points, all T tasks observed (no NaN), or NULL / 0:
 *         every chain conditions on them (one append-row pass per point) before step 0
 *   X_v0, Y_v0 (same shapes with n_v0 points) further seed points observed with hall_tasks tasks only (the value-only
 *         forward-sampling points of src/agent.py:399-405); conditioned on after the full seeds
 *   state [dev] gpmpc_rollout_state_bytes(gp, Ns, state_slots, state_points) or NULL: per sample the counts, the points and
 *         per chain L_hr^T, L_hh (packed), w, 1/diag - written by the call (the last step's draw is appended too); with
 *         resume = 1 the chains continue from it instead of from seeds (same hall_tasks, same state_slots / state_points)
 *   limits: state_slots <= 256; without a state n_h0*T + hall_tasks*(n_v0 + H-1) <= 256 label slots per chain
 *   info bit GPMPC_INFO_STATE_FULL: a resumed state had no room left (label slots or points) for a new point: the point
 *         is not appended, the draw itself and X_traj / Y / Xi are still valid
 * Kernel: without a state, T = 3 and 3 (n_h0 + n_v0 + H - 1) <= 192 the call runs the tiled FP64-MFMA kernel
 * (csrc/rollout_tiles.hip; shapes and sizes as for gpmpc_rollout: from 256 chains on, or pinned) - the seed points are
 * conditioning-only passes of its step body, a value-only point (hall_tasks = 1) keeps three row slots of which the two
 * derivative rows are dead.  Otherwise, and with a kept / resumed state, the generic kernel (csrc/rollout.hip).
 */
size_t gpmpc_rollout_state_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int32_t state_slots, int32_t state_points);
/* workspace of a seeded call WITHOUT a state: the chains' factor covers n_h0*T + hall_tasks*(n_v0 + H-1) label slots and
 * leaves LDS sooner than gpmpc_rollout's (>= gpmpc_rollout_workspace_bytes for the same Ns, H) */
size_t gpmpc_rollout_seeded_workspace_bytes(const gpmpc_gp_desc_t* gp, int32_t mode, int32_t hall_tasks,
                                            int64_t Ns, int32_t H, int32_t n_h0, int32_t n_v0);
int    gpmpc_rollout_seeded(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, const void* plan,
                            const double* X_r, int32_t mode, int32_t hall_tasks, double var_zero_thr, double beta,
                            int64_t Ns, int32_t H,
                            const double* x0, int32_t x0_per_sample, const double* u_ff,
                            const double* z, int64_t z_step_stride,
                            double* X_traj, double* Y, double* Xi, int32_t* info,
                            void* ws, size_t ws_bytes, void* stream,
                            const double* X_h0, const double* Y_h0, int32_t n_h0,
                            const double* X_v0, const double* Y_v0, int32_t n_v0,
                            void* state, int32_t state_slots, int32_t state_points, int32_t resume);

/*
 * gpmpc_joint_sample - joint posterior draw at m test points per (sample, output), conditioning on the shared
 * real data plus per-sample hallucinated data.
 * Replaces: reference src/agent.py:629-708 (sample_gp): model_i(x_input) [gpytorch ExactGP eval call, SURVEY App.
 * A.4-A.6], .sample(base_samples) [A.7 root: Cholesky with jitter-on-failure, then the eigendecomposition root
 * R = U sqrt(max(lambda, 0)) for the whole batch - the branch params_car_residual.yaml:51 (Dyn_gp_jitter 1e-20)
 * takes on every draw; second kernel of the same call, no host round trip], .variance [A.8], the optional
 * variance-is-zero replacement and the beta clip.  The min-data-distance overwrite (src/agent.py:666-698) stays in
 * the facade (disabled in every shipped config).
 *   X_h     [dev] (Ns, g_ny, n_h, D)   Hallcinated_X_train (NULL if n_h == 0)
 *   Y_h     [dev] (Ns, g_ny, n_h, T)   Hallcinated_Y_train
 *   h_slots [dev] (n_ho) int32         observed hallucinated label slots, ascending, slot = point*T + task
 *                                      (gpytorch "mask" policy collapsed over the batch, A.4)
 *   X_s     [dev] (Ns, g_ny, m, D)     test inputs (g_xu_hat)
 *   z       [dev] (Ns, g_ny, m, T)     base samples
 *   mean, var, y [dev] (Ns, g_ny, m, T) out (var floored; y post-processed as above)
 *   covar   [dev] (Ns, g_ny, m*T, m*T) or NULL   out: posterior covariance (model_i_call.covariance_matrix)
 *   root    [dev] (Ns, g_ny, m*T, m*T) or NULL   out, eigh branch only: the root R actually used (columns ordered by
 *                                      ascending eigenvalue like torch.linalg.eigh; column signs are solver specific)
 *   root_mode  GPMPC_ROOT_*
 *   info    [dev] (Ns, g_ny) int32
 *   factor_cache [dev] gpmpc_joint_cache_bytes(gp, Ns, cache_rows) or NULL: per chain the hallucinated rows of the
 *           factor (L_hr, L_hh) and 1/diag.  The reference re-factorises K_oo from scratch on every call
 *           (src/agent.py:241-250, 640); in the SQP loop the hallucinated set only GROWS between two resets
 *           (src/agent.py:164-202, 261-272), so the rows of the slots that were already there are unchanged: with
 *           n_cached > 0 (<= n_ho) the first n_cached rows are taken from the cache - the CALLER vouches
 *           that h_slots[:n_cached] and their points X_h are the ones of the call that filled it (labels may differ:
 *           the factor does not depend on them) - and every call writes the rows it computed (n_ho <= cache_rows).
 *           cache_rows must be EVEN (>= 16) and the buffer 16-byte aligned (GPMPC_E_ARG otherwise; the matrix-pipe path
 *           moves rows in 16-byte units).
 *           Bit-identity: with the joint path PINNED (gpmpc_joint_pin_path(GPMPC_JOINT_VALU) or ..._MFMA) results are
 *           bit-identical with and without the cache.  Under GPMPC_JOINT_AUTO the dispatcher takes the matrix-pipe path
 *           from GPMPC_JOINT_MFMA_FROM (100; 48 for wide test blocks) observed hallucinated slots on, and chains that HAVE cache room take its
 *           factor extension while chains beyond the cache budget compute their rows on the VALU: a sample's low-order
 *           bits then depend on which side of the budget it sits (1e-13 on a Cholesky root, up to 1e-5 on the
 *           eigendecomposition root of params_car_residual.yaml).  Pin the path wherever
 *           bit-reproducibility across batch sizes / GPU counts is claimed (make_sharded_agent(..., pin_joint_path=...)
 *           of sampling_gpmpc_amd.distributed does it for a sharded run; the default leaves the dispatcher free).
 *           The cache may cover fewer chains than the batch: the host splits the batch into one call over the samples it
 *           has cache room for and one over the rest (factor_cache NULL); results are the same (a call's chains are
 *           independent), the caller applies the whole-batch eigh rule across the two calls (root_mode GPMPC_ROOT_EIGH).
 *   limits (the CONTRACT of this entry point): m*T <= 256 test slots and n_ho + 1 + m*T <= 2048 label rows per chain,
 *           GPMPC_E_UNSUPPORTED beyond.  In the SQP loop the hallucinated set grows by m*T = H*T slots per iteration and
 *           is reset at sqp_iter == 0, so a draw is possible for (2048 - 1 - H*T) / (H*T) iterations after a reset: 16 at
 *           H = 40 (configs[4] as benchmarked), 12 at the shipped H = 50 (reference params_car_residual.yaml:88 allows
 *           max_sqp_iter 150, which the reference itself cannot reach: its dense re-factorisation grows with the cube of
 *           the rows; shipped runs use <= 4).  The limit is the one-row-per-thread mapping (four rows per thread beyond
 *           512 rows); lifting it means tiling the row dimension over a second grid axis and is not done.
 */
size_t gpmpc_joint_cache_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int32_t cache_rows);
size_t gpmpc_joint_workspace_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int32_t n_ho, int32_t m);
int    gpmpc_joint_sample(const gpmpc_gp_desc_t* gp, const void* plan, const double* X_r,
                          int64_t Ns, int32_t n_h, const double* X_h, const double* Y_h,
                          const int32_t* h_slots, int32_t n_ho,
                          int32_t m, const double* X_s, const double* z,
                          double var_zero_thr, double beta, int32_t apply_clip,
                          double* mean, double* var, double* y, double* covar, double* root,
                          int32_t root_mode, int32_t* info,
                          void* ws, size_t ws_bytes, void* stream,
                          void* factor_cache, int32_t cache_rows, int32_t n_cached);

/*
 * gpmpc_joint_sample_pending (ABI 9) - gpmpc_joint_sample with PENDING ROWS of the factor cache.
 * In the SQP loop the points a draw is made at are appended to the hallucinated set (reference src/agent.py:629-641, then
 * :164-202): the next call's new slots ARE this call's test slots, and what this call computes for them - X = L^-1 K_o* and
 * S = K** - X^T X - is the new rows' block of the factor against the old columns and their Schur complement (up to the likelihood
 * noise of A.3 on its diagonal).  pending = bit mask:
 *   GPMPC_PENDING_WRITE  this call may also write X^T into the cache rows n_ho .. n_ho + m*T - 1 and S into their diagonal block
 *                        (matrix-pipe path, one test-mode launch, the caller's cache with room for the rows; T = 3, m*T <= 128);
 *                        gpmpc_joint_pending_written() says whether it did;
 *   GPMPC_PENDING_USE    the CALLER vouches that the rows n_cached .. n_ho - 1 of the cache were written that way by the previous
 *                        call (same cache, h_slots[n_cached:] = all tasks of exactly that call's test points, in order, and
 *                        h_slots[:n_cached] the set that call conditioned on): the factor extension is then the Cholesky of
 *                        (S + noise) in place and nothing else.  A permission: where the path or the shapes do not allow it the
 *                        rows are recomputed.
 * Results agree with gpmpc_joint_sample to rounding (the new rows are the same sums in another order).
 */
#define GPMPC_PENDING_USE   1
#define GPMPC_PENDING_WRITE 2
int    gpmpc_joint_sample_pending(const gpmpc_gp_desc_t* gp, const void* plan, const double* X_r,
                                  int64_t Ns, int32_t n_h, const double* X_h, const double* Y_h,
                                  const int32_t* h_slots, int32_t n_ho, int32_t m,
                                  const double* X_s, const double* z,
                                  double var_zero_thr, double beta, int32_t apply_clip,
                                  double* mean, double* var, double* y, double* covar, double* root,
                                  int32_t root_mode, int32_t* info, void* ws, size_t ws_bytes, void* stream,
                                  void* factor_cache, int32_t cache_rows, int32_t n_cached, int32_t pending);
int    gpmpc_joint_pending_written(void);    /* 1: the last gpmpc_joint_sample[_pending] call wrote pending rows */

/*
 * The two paths of gpmpc_joint_sample (ABI 7).  GPMPC_JOINT_VALU: one launch, one label row per thread, blocked left-looking
 * factorisation on the vector pipe (every size).  GPMPC_JOINT_MFMA: four launches on the same stream - joint_test_mfma_kernel
 * extends the factor by the rows of the new hallucinated slots (their entries against the old columns, the Schur complement),
 * joint_kernel factorises the Schur complement, joint_test_mfma_kernel forms V^T = L^-1 K_o*, the mean and S = K** - V^T V on the
 * FP64 matrix pipe with the whole test block in registers, the tail draws - instantiated for n_r <= 64 real slots,
 * n_r + n_ho <= 416 conditioning slots (<= 544 with the test rows in two launches; that form needs the caller's factor cache) and
 * m*T + 1 <= 128; taken from 100 hallucinated slots on, from 48 when m*T >= 84
 * (GPMPC_JOINT_MFMA_FROM, when set, is the whole rule).  Round 6: a draw WITHOUT hallucinated slots (T = 3, <= 64 real slots, m*T <= 128) also reports
 * GPMPC_JOINT_MFMA unless the VALU path is pinned - joint_real_mfma_kernel forms X = L_rr^-1 K_r*, the mean and S one wave per
 * chain against the plan's shared inverse factor (GPMPC_JOINT_REAL_KERNEL=0: joint_kernel's head as before); the same kernel
 * is the matrix-pipe path's factor extension while nothing is cached.  Results of the two paths agree to rounding, not bit for bit: a caller that compares launches bit
 * for bit (cache on / off, sample shards against the whole batch) pins the path.  gpmpc_joint_pin_path(GPMPC_JOINT_AUTO) releases
 * the pin; a pinned GPMPC_JOINT_MFMA falls back to the VALU path for sizes it is not instantiated for.
 */
#define GPMPC_JOINT_AUTO 0
#define GPMPC_JOINT_VALU 1
#define GPMPC_JOINT_MFMA 2
int    gpmpc_joint_pin_path(int32_t path);
int    gpmpc_joint_last_path(void);          /* GPMPC_JOINT_VALU / GPMPC_JOINT_MFMA: what the last gpmpc_joint_sample ran; 0 before */

/*
 * gpmpc_assemble_jacobians - full-state value and Jacobians from the GP sample.
 * Replaces: reference src/agent.py:532-557 (dyn_fg_jacobians: env.get_f_known_jacobian, env.transform_sensitivity,
 * scatter into pad_g columns, B_d matmul, split) for the two environments.
 *   xu      [dev] (Ns, nx, H, nx+nu)   batch_x_hat (row 0 of dim 1 is read; rows are replicas)
 *   y       [dev] (Ns, g_ny, H, T)     GP sample
 *   gp_val  [dev] (Ns, nx, H, 1), y_grad [dev] (Ns, nx, H, nx), u_grad [dev] (Ns, nx, H, nu)   out
 */
int    gpmpc_assemble_jacobians(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env,
                                int64_t Ns, int32_t H, const double* xu, const double* y,
                                double* gp_val, double* y_grad, double* u_grad, void* stream);

/*
 * gpmpc_pack_plin - stage parameter vectors for the acados OCP.
 * Replaces: the O(Ns^2) np.concatenate loop of reference src/solver.py:98-131.  Per stage:
 * for each sample [A_i row-major (nx*nx), B_i row-major (nx*nu), x_hat_i (nx), f_i (nx)], then
 * [u_hat (nu), xg (1), w (1), tilde_eps (nx+nu+1)].
 *   y_grad, u_grad, gp_val as above; x_h [dev] (H, Ns*nx); u_h [dev] (H, nu); xg [dev] (H); w [dev] (H);
 *   tilde_eps [dev] (H, nx+nu+1);  p_lin [dev] (H, Ns*(nx*nx+nx*nu+2*nx) + 2*nu... see gpmpc_plin_len)
 */
int64_t gpmpc_plin_len(int32_t nx, int32_t nu, int64_t Ns);
int     gpmpc_pack_plin(int32_t nx, int32_t nu, int64_t Ns, int32_t H,
                        const double* y_grad, const double* u_grad, const double* gp_val,
                        const double* x_h, const double* u_h, const double* xg, const double* w,
                        const double* tilde_eps, double* p_lin, void* stream);
/* the same with the feedback law folded in: A_i = y_grad + u_grad K (reference src/solver.py:90), K [dev] (nu, nx) or NULL */
int     gpmpc_pack_plin_fb(int32_t nx, int32_t nu, int64_t Ns, int32_t H,
                           const double* y_grad, const double* u_grad, const double* gp_val,
                           const double* x_h, const double* u_h, const double* xg, const double* w,
                           const double* tilde_eps, const double* K, double* p_lin, void* stream);

/*
 * gpmpc_build_x_hat (ABI 8) - batch_x_hat from the solver's iterate in one launch.
 * Replaces: reference src/agent.py:480-501 (get_batch_x_hat_u_diff) / 503-527 (get_batch_x_hat): reshape, input broadcast and
 * the nx-fold replication of the state row.
 *   x_h [dev] (H, Ns*nx); u_h [dev] (H, nu) shared by the samples (u_per_sample 0) or (H, Ns, nu) (u_per_sample 1)
 *   xu  [dev] (Ns, nx, H, nx+nu)   out
 */
int     gpmpc_build_x_hat(int32_t nx, int32_t nu, int64_t Ns, int32_t H, const double* x_h, const double* u_h,
                          int32_t u_per_sample, double* xu, void* stream);

/*
 * gpmpc_assemble_jacobians_plin (ABI 8) - gpmpc_assemble_jacobians and gpmpc_pack_plin_fb in ONE launch: the three arrays
 * (which may stay on the device) and the stage parameter vectors the reference's solver consumes (src/solver.py:98-131; x_hat_i
 * is read from xu).  Arguments as the two entry points; u_h [dev] (H, nu) is the nominal input of the stage tail.
 */
int     gpmpc_assemble_jacobians_plin(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int64_t Ns, int32_t H,
                                      const double* xu, const double* y, double* gp_val, double* y_grad, double* u_grad,
                                      const double* u_h, const double* xg, const double* w, const double* tilde_eps,
                                      const double* K, double* p_lin, void* stream);
/* bitwise OR of n int32 words ([dev], e.g. the per-chain info words of a launch) into out[0] ([dev], zeroed by the caller):
 * one launch and one word to read where the facade used a reduction per flag bit (nothing of the reference is replaced:
 * gpytorch raises / warns from host-side checks of its own, src/agent.py:629-641) */
int     gpmpc_or_reduce_words(const int32_t* v, int64_t n, int32_t* out, void* stream);

/*
 * gpmpc_base_samples - the Agent's base samples ("epistimic_random_vector") from a counter-based stream (ABI 7).
 * Replaces: reference src/agent.py:76-104 (random_vector_within_bounds: i.i.d. N(0,1) vectors of shape (g_ny, H, T), the whole
 * vector redrawn until every entry lies in [-beta, beta]) for the sample-sharded runs: vector (j, i, s) is a pure function of
 * (seed, MPC step j, SQP iteration i, GLOBAL sample id offset + s), so a rank generates exactly its own shard on its own GPU and
 * the assembled run does not depend on the GPU count.  The reference's own stream (one torch.normal call per candidate on a global
 * generator) stays available in the facade (base_sample_generator "reference").
 *   out      [dev] (n_mpc, n_itrs, Ns, V) float64, V = g_ny * H * T
 *   attempts [dev] (n_mpc, n_itrs, Ns) int32 or NULL: how many candidates were rejected before the kept one
 * One wave per vector, the rejection loop inside the wave.  GPMPC_E_ARG when beta is so small that an attempt is accepted with
 * probability < 1e-6.
 */
int     gpmpc_base_samples(uint64_t seed, int32_t n_mpc, int32_t n_itrs, int64_t offset, int64_t Ns, int32_t V, double beta,
                           double* out, int32_t* attempts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPMPC_HIP_H */
