/* bayesfilt.h -- C-ABI of the MI355X-native batched Bayesian-filtering engine.
 *
 * The reference (kostastsa/BayesianFiltering, package `gaussfiltax`) has no FFI/plugin
 * interface; its boundary for the filtering hot path is the Python call surface
 *
 *   gaussian_sum_filter(params, emissions, num_components, num_iter, inputs)
 *                                               gaussfiltax/inference.py:303-309
 *   bootstrap_particle_filter(params, emissions, num_particles, key, inputs, ess_threshold)
 *                                               gaussfiltax/inference.py:1302-1309
 *
 * Each entry point below replaces the `lax.scan` body of one of those drivers
 * (inference.py:333-371 and :1330-1377) for a whole batch of independent trajectories.
 * Conventions: extern "C", plain pointers and sizes, no exceptions; every function returns an
 * int status (0 = BF_OK, negative = BF_E*; text via bf_last_error()).  All array arguments
 * named `d_*` or carried in bf_stream/bf_cstream/bf_carry are DEVICE pointers owned by the
 * caller; the engine never allocates outputs.  Model parameter structs hold HOST pointers to
 * small row-major fp32 arrays which are copied into kernel arguments at launch.  `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  Launches are asynchronous: no entry point synchronises
 * the host with the stream (constant blocks are uploaded stream-ordered through a content-keyed cache and found
 * again on later calls, so a warmed-up call is also capturable into a hipGraph).
 *
 * Arithmetic is fp32 like the reference's JAX path (no jax_enable_x64 anywhere); the
 * reference's quirks are reproduced: update -> reweight -> predict order,
 * psd_solve = LU solve of (S + 1e-6 on every entry) (gaussfiltax/utils.py:256-259),
 * P+ = P - K S K^T with the un-jittered S, log-likelihood by Cholesky of the un-jittered S,
 * linear-domain weights.
 */
#ifndef BAYESFILT_H_
#define BAYESFILT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_VERSION 210 /* 0.2.0: major = ABI revision (struct layouts), see bf_abi_check */

#define BF_OK 0
#define BF_EINVAL (-1)       /* bad argument (NULL pointer, non-positive size, misaligned) */
#define BF_EUNSUPPORTED (-2) /* dimension / model / option not compiled into this build  */
#define BF_EHIP (-3)         /* HIP runtime error (text in bf_last_error)                */
#define BF_ENOGPU (-4)       /* no gfx950 device visible                                 */

/* Strided view of an fp32 output stream.  Element (b, k, t, e) lives at
 *   ptr[b*sB + k*sK + t*sT + e*sE]      (strides in ELEMENTS)
 * where e indexes the row-major flattened event (vector entry i, or matrix entry i*n+j).
 * ptr == NULL: the stream is not emitted.  Two layouts have tuned kernels:
 *   "reference" [B][K][T][E] contiguous (sE = 1, sT = E, sK = T*E, sB = K*T*E): what the
 *       reference returns per trajectory after swap_axes_on_values (inference.py:372), with
 *       a leading batch axis;
 *   "batch-inner" [K][T][E][B] (sB = 1): the scan-native order, perfectly coalesced.
 * Any other stride set runs on the generic strided path. */
typedef struct bf_stream {
  float* ptr;
  int64_t sB, sK, sT, sE;
} bf_stream;

typedef struct bf_cstream {
  const float* ptr;
  int64_t sB, sK, sT, sE; /* sK unused for observations / inputs */
} bf_cstream;

/* The five arrays of PosteriorGaussianSumFiltered (inference.py:29-39, emitted at
 * :357-363) plus the per-step component log-likelihoods (the `lls` of :345, which the
 * reference computes but does not return). */
typedef struct bf_out_desc {
  bf_stream weights;    /* E = 1    */
  bf_stream means;      /* E = n    */
  bf_stream covs;       /* E = n*n  */
  bf_stream pred_means; /* E = n    */
  bf_stream pred_covs;  /* E = n*n  */
  bf_stream loglik;     /* E = 1    */
  /* COLLAPSED mode (bf_gsf_ekf_f32 only; sK unused): the moment-matched single Gaussian of the
   * filtered mixture at every step -- utils.collapse (utils.py:10-18) and the point estimate
   * sum_k w_k m_k of BOT_Experiment_script.py:101 -- formed inside the scan, so a K-component run
   * can return 4(n + n*n) bytes per step instead of the K-fold streams above. */
  bf_stream coll_mean;  /* E = n    */
  bf_stream coll_cov;   /* E = n*n  */
} bf_out_desc;

/* Scan carry (weights, pred_means, pred_covs) of inference.py:334,356: the state the filter
 * starts from and (optionally) the state it ends with, so that T can be processed in chunks
 * when the full posterior history exceeds HBM.  Contiguous [B][K], [B][K][n], [B][K][n][n].
 * *_out may alias *_in; NULL *_out = not written.  w_in == NULL means 1/K (inference.py:369). */
typedef struct bf_carry {
  const float* w_in;
  const float* m_in; /* required: initial (predicted) means, inference.py:367 */
  const float* P_in; /* required: initial (predicted) covariances, inference.py:368 */
  float* w_out;
  float* m_out;
  float* P_out;
} bf_carry;

/* Linear-Gaussian SSM   f(x,q,u) = A x + G q,  h(x,r,u) = H x + D r   (the "Kalman" model:
 * gaussian_sum_filter with num_components = 1 and linear f, h; e.g.
 * docs/experiments/adaptive_experiment.py:59-64, BOT_Experiment_script.py:31-41).
 * HOST pointers, row-major fp32.  G == NULL: identity (dq = n); D == NULL: identity (dr = m);
 * q0 / r0 == NULL: zeros.  Q_steps / R_steps: 1 = time-invariant [d,d]; T = one matrix per
 * step [T,d,d] (the `_get_params(x, 2, t)` rule of inference.py:21,337-340). */
typedef struct bf_lgssm {
  int32_t n, dq, m, dr;
  const float *A, *G, *H, *D;
  const float *q0, *r0;
  const float *Q, *R;
  int32_t Q_steps, R_steps;
} bf_lgssm;

int bf_version(void);
/* ABI guard.  A binding that mirrors the structs of this header by hand (ctypes, cgo, JNI ...) calls this once after
 * loading the library, with the BF_VERSION it was written against and ITS sizes of the structs it fills; a mismatch
 * (e.g. a bf_out_desc with six streams against the library's eight) returns BF_EINVAL with the offending struct in
 * bf_last_error() instead of an out-of-bounds read later.  A size of 0 means "this binding does not mirror that struct". */
int bf_abi_check(int32_t header_version, size_t sizeof_out_desc, size_t sizeof_lgssm, size_t sizeof_model,
                 size_t sizeof_bpf_model, size_t sizeof_bpf_out);
const char* bf_last_error(void);
/* Number of visible gfx950 devices (0 if none / HIP unavailable). */
int bf_device_count(void);

/* Tuning / test hooks.
 *   "kf_emit_mode": -1 = choose the store path from the layout (default), 0 = strided dword
 *                   stores, 2 = LDS time-transpose (contiguous reference layout only).
 *   "kf_lanes":     lanes that cooperate on one trajectory (0 = default for the dimensions;
 *                   otherwise one of the compiled powers of two, e.g. 1, 2 or 4 at n = 4).
 *   "kf_mfma_variant": the n = 64, m = 32 Kalman kernel: 5 (default) = gain-free update (P+ = P - W^T W + c c^T with
 *                   W = L^-1 H P) with the five matrix products as three-term bf16 splits of the fp32 operands on the
 *                   bf16 matrix pipe (fp32-level rounding); 2 = the same update on fp32 MFMAs, Cholesky and forward
 *                   substitution fused in one wave's registers; 3 = 2 with the factorization itself as rank-2 MFMA
 *                   eliminations; 4 = 2 at three workgroups per CU; 1 = round 1's kernel (explicit inverse through LDS).
 *                   Same results to rounding (< 5e-6 against the test oracle over 2 000 steps); env BAYESFILT_MFMA_VARIANT
 *                   sets the default.
 *   "kf_small_mode": 1 (default) = Kalman models with 9 <= n <= 32, m <= 32 run on the one-wave-per-trajectory matrix-core
 *                   kernel (single 32 x 32 tiles, bf16 three-term products); 0 = off (n >= 24 then rides padded in the
 *                   (64, 32) kernel, smaller n on the run-time-dimension kernel); 2 = the same kernel with two chains per wave
 *                   (one factorization serves both half-waves; identical bits, measured 10-20 % slower: kept for experiments).
 *   "force_generic": 1 = bf_kalman_filter_f32 / bf_gsf_ekf_f32 run the run-time-dimension kernel (any n, m, K; state in
 *                   LDS) even where a compile-time-dimension instance exists (test hook; default 0).
 *   "gsf_structured": 1 (default) lets bf_gsf_ekf_f32 use the structure-aware kernel instances
 *                   (banded Lorenz-96 Jacobian, selection emission) when the model qualifies;
 *                   0 forces the dense generic instances.
 *   "bpf_variant":  workgroup geometry of the 4096-particle instance: 0 (default) = 1024 threads x 4 particles, 1 = 512 x 8.
 *   "bpf_hbm_mode": particle counts beyond the in-register capacities: 0 (default) = choose by batch
 *                   size, 1 = one workgroup per trajectory, 2 = one workgroup per 1024-particle chunk
 *                   (six launches per step; same results bit for bit).
 *   "bpf_spec":     1 (default) = models whose structure has a compile-time instance (Lorenz-96 dynamics with identity noise
 *                   input, diagonal chol(Q), selection emission, diagonal chol(R)) run on it; 0 = the run-time instance
 *                   (same results bit for bit).
 *   "bpf_arith":    arithmetic of the particle filter's weight path (the normal draws' erf_inv, the exp of the weights): 0
 *                   (default) = the engine's DEFINED fp32 arithmetic (IEEE operations in a fixed order; resampling indices
 *                   reproducible bit for bit); 1 = the hardware's v_log_f32 / v_exp_f32 (1 ulp, not
 *                   reproducible across architectures; results agree with mode 0 to rounding until a uniform draw lands within
 *                   that rounding of a CDF step).  Mode 1 compiles the kernel at run time (needs hiprtc, like bf_user_model_create)
 *                   and serves particle counts up to 4 096 with registry functions.
 * bf_set_option changes the PROCESS-WIDE default.  A library or a thread that must not disturb -- or be disturbed by -- other
 * callers uses bf_set_call_option instead: it arms the same option on the CALLING THREAD for the NEXT filter entry point
 * called on that thread (bf_kalman_filter_f32, bf_gsf_ekf_f32, bf_ugsf_ukf_f32, bf_agsf_*, bf_bpf_f32, bf_sample_ssm_f32,
 * bf_resample_f32, bf_optimal_resample_f32, bf_collapse_f32) and for that call only; every armed override is dropped when
 * that call returns, whatever its status. */
int bf_set_option(const char* name, int value);
int bf_set_call_option(const char* name, int value);

/* Batched Kalman filter: B independent trajectories, one component each (K = 1), T steps.
 * Replaces the lax.scan of gaussian_sum_filter (inference.py:333-371) with
 * _condition_on (:72-105) and _predict (:51-70) specialised to linear f, h.
 * y: observations, element (b, t, e) at y->ptr[b*sB + t*sT + e*sE], E = m. */
int bf_kalman_filter_f32(const bf_lgssm* model, const bf_cstream* y, int64_t B, int64_t T,
                         const bf_carry* carry, const bf_out_desc* out, void* stream);

/* A state-space model built from the function registry (the device-side replacement of the
 * Python callables f(x,q,u), h(x,r,u) of gaussfiltax/models.py:46-49; ids and theta layouts:
 * bayesianfiltering_amd/nonlinearities.py, formulas: csrc/models.hpp).  HOST pointers. */
typedef struct bf_model {
  int32_t dyn_id, emi_id;
  int32_t n, dq, m, dr;
  const float* dyn_theta;
  int32_t n_dyn_theta;
  const float* emi_theta;
  int32_t n_emi_theta;
  const float *q0, *r0; /* noise biases, NULL = zeros                */
  const float *Q, *R;   /* noise covariances [dq,dq], [dr,dr]; [steps,d,d] when Q_steps / R_steps > 1 */
  int32_t flags;        /* 0 = the JAX path's semantics; BF_MODEL_* bits for the legacy NumPy classes */
  int32_t Q_steps, R_steps; /* 0 or 1 = constant; T = one covariance per step, the (T,d,d) arrays that
                               _get_params(..., 2, t) selects from (inference.py:21, :337-340).  Honoured by
                               bf_gsf_ekf_f32 (emissions with a constant H_r); the sampling kernels need 0 / 1. */
  const struct bf_user_model* user; /* functions compiled from source (dyn_id / emi_id = BF_FN_USER), else NULL */
} bf_model;

/* ---- user-defined f / h ------------------------------------------------------------------------
 * The reference accepts arbitrary Python callables f(x, q, u), h(x, r, u) (gaussfiltax/models.py:46-49) and takes their
 * Jacobians with jacfwd (gaussfiltax/inference.py:328-329).  Here a function outside the registry is given as HIP C++
 * SOURCE TEXT, written once for any scalar type:
 *
 *     template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* theta, T* out);   // out[BF_N]
 *     template <class T> __device__ void emission(const T* x, const T* r, T u, const float* theta, T* out);   // out[BF_M]
 *
 * (BF_N, BF_DQ, BF_M, BF_DR are compile-time constants; sin cos tan exp log sqrt tanh atan atan2 pow abs are available
 * for T; theta = bf_model.dyn_theta / emi_theta).  bf_user_model_create compiles it with hiprtc into the
 * run-time-dimension scan kernel; values come from T = float, Jacobians w.r.t. state AND noise from forward-mode dual
 * numbers (what jacfwd computes), F_q Q F_q^T / H_r R H_r^T are formed on the device every step.  Either source may be
 * NULL (that side stays a registry function).  Compiled models are cached by source (memory + $BAYESFILT_CACHE_DIR,
 * default .jit_cache next to the library); a source that does not compile returns BF_EINVAL with the compiler's first error in
 * bf_last_error().  Use: set bf_model.user and dyn_id / emi_id = BF_FN_USER, then call bf_gsf_ekf_f32.  For n, dq, m, dr <= 8 (both
 * functions from source, or the other one the registry's linear function; no legacy flags, no collapsed streams) the handle also
 * builds, on first use, the Gaussian-sum scan with the state in REGISTERS (one lane per (trajectory, component), the same dual-number
 * Jacobians): two orders of magnitude faster than the run-time-dimension kernel and the default there ("force_generic" = 1 keeps
 * the LDS kernel). */
#define BF_FN_USER 100
typedef struct bf_user_model bf_user_model;
int bf_user_model_create(const char* dynamics_src, const char* emission_src, int32_t n, int32_t dq, int32_t m, int32_t dr,
                         bf_user_model** model);
/* ... and for the bootstrap particle filter (bf_bpf_f32), whose model is f, the emission log-density and nothing else
 * (gaussfiltax/inference.py:1344-1349: x' = f(x, q, u), lls = emission_distribution_log_prob(x', y, u), any callables):
 * the same handle also compiles the particle-filter kernel with the caller's functions, on first use and per particle
 * capacity (particles in registers up to 4096 for state_dim <= 16 / 1024 beyond; above that the particles-in-HBM kernel, up to
 * 2^20 per trajectory) -- and, likewise on first use, the unscented (bf_ugsf_ukf_f32) and augmented (bf_agsf_ekf_f32 /
 * bf_agsf_ukf_f32) scans and the data generator (bf_sample_ssm_f32) around the same functions (dimensions up to 8; sampler 32).  The density is either Gaussian around the (registry or
 * source) emission function, MVN(h(x, r_eval, u), lp_cov) as for registry models, or -- log_prob_src -- the caller's own
 *
 *     template <class T> __device__ T log_prob(const T* x, const float* y, T u, const float* theta);   // theta = bf_bpf_model.lp_theta
 *
 * In these kernels sin cos sincos atan atan2 exp log sqrt abs fma are the CANONICAL fp32 arithmetic of the weight path
 * (bf_canon_eval_f32), so a function written like a registry function gives that function's bits.  Any of the three
 * sources may be NULL (at least one is given). */
int bf_user_model_create_lp(const char* dynamics_src, const char* emission_src, const char* log_prob_src, int32_t n, int32_t dq,
                            int32_t m, int32_t dr, bf_user_model** model);
void bf_user_model_destroy(bf_user_model* model);

/* Hyper-parameters of the unscented transform -- ParamsUKF (inference.py:41-49): lambda =
 * alpha^2 (L + kappa) - L with L = state_dim + noise_dim of the augmented state. */
typedef struct bf_ukf_params {
  float alpha, beta, kappa;
} bf_ukf_params;

/* Legacy-class semantics (gaussfiltax/gaussfilt.py, gausssumfilt.py), honoured by bf_gsf_ekf_f32: */
#define BF_MODEL_PREDICT_FIRST 1  /* step order predict -> update (gaussfilt.py:113-121); carry = filtered state   */
#define BF_MODEL_NO_JITTER 2      /* gain from S itself, no 1e-6 (gaussfilt.py:118 `Cxy @ inv(Sy)`)                 */
#define BF_MODEL_LEGACY_GSF_COV 4 /* predicted covariance P + J P J^T, Q never added (gausssumfilt.py:59)           */

/* Batched Gaussian-sum filter (bank of K extended Kalman filters + weight update): replaces the
 * lax.scan of gaussian_sum_filter (inference.py:333-371) for K >= 1 and nonlinear f, h.
 * u: optional inputs, element (b, t) at u->ptr[b*sB + t*sT] (only u[0] is used by the registry
 * functions); NULL or u->ptr == NULL means zeros((T,1)) (inference.py:23).  The carry holds
 * the K initial component means / covariances per trajectory ([B][K][n], [B][K][n][n]) and,
 * optionally, weights [B][K] (NULL = 1/K, inference.py:369).  K rounded up to a power of two
 * times the lanes per chain must not exceed 256. */
int bf_gsf_ekf_f32(const bf_model* model, const bf_cstream* y, const bf_cstream* u, int64_t B, int64_t T, int32_t K,
                   const bf_carry* carry, const bf_out_desc* out, void* stream);

/* Batched unscented Gaussian-sum filter: K unscented Kalman filters with non-additive noise
 * (augmented sigma points) per trajectory plus the weight update.  Replaces the lax.scan of
 * unscented_gaussian_sum_filter (inference.py:379-456) with _ukf_condition_on_nonadditive
 * (:198-224), the reweight (:424-427) and _ukf_predict_nonadditive (:146-174); sigma points as
 * utils._get_sigma_points (utils.py:247-254) with the symmetric matrix square root computed on the
 * device.  Arguments as bf_gsf_ekf_f32; constant covariances, out->coll_* unsupported. */
int bf_ugsf_ukf_f32(const bf_model* model, const bf_ukf_params* uparams, const bf_cstream* y, const bf_cstream* u,
                    int64_t B, int64_t T, int32_t K, const bf_carry* carry, const bf_out_desc* out, void* stream);

/* Batched "speedy" augmented Gaussian-sum filter: replaces the lax.scan of
 * speedy_augmented_gaussian_sum_filter (inference.py:621-812).  num_components = (N0, N1, N2): every
 * step branches the N0 carried components into N1 z-samples each (predicted with covariance
 * opt_args[0] * P), every prediction into N2 s-samples (updated with covariance opt_args[1] * P-), and
 * draws N0 of the N0*N1*N2 leaves with jr.choice under PRNGKey(0).  key: the reference's rng_key (it is
 * never advanced: the same normals at every step).  The carry holds N0 components per trajectory;
 * out: weights / means / covs with K = N0 (the other streams must be unset).  leaf_idx: optional
 * DEVICE int32 [B][T][N0], the leaf each carried component was drawn from.  N0*N1*N2 <= 64 in general,
 * <= 1024 for state_dim <= 4 (one workgroup per trajectory; every variant).
 * variant: 0 = the speedy filter's two shared normal arrays (:672-688, :716-726); 1 = the branches of
 * augmented_gaussian_sum_filter (inference.py:458-620) through containers._branches_from_tree1/2
 * (containers.py:63-140): one key per node, jr.multivariate_normal per node, NaN samples replaced by the mean;
 * 2 = augmented_gaussian_sum_filter_optimal (:1157-1300): the branches of 1 and utils.optimal_resampling
 * (utils.py:216-244) instead of jr.choice, so the carried components keep unequal weights. */
int bf_agsf_ekf_f32(const bf_model* model, const bf_cstream* y, const bf_cstream* u, int64_t B, int64_t T,
                    const int32_t num_components[3], const uint32_t key[2], const float opt_args[2], const bf_carry* carry,
                    const bf_out_desc* out, int32_t* leaf_idx, int32_t variant, void* stream);

/* The augmented filter with unscented nodes: replaces the lax.scan of speedy_unscented_agsf
 * (inference.py:966-1156; variant 0) and unscented_agsf (:813-965; variant 1) -- the tree, sampling and
 * resampling of bf_agsf_ekf_f32 with _ukf_predict_nonadditive / _ukf_condition_on_nonadditive at the nodes. */
int bf_agsf_ukf_f32(const bf_model* model, const bf_ukf_params* uparams, const bf_cstream* y, const bf_cstream* u, int64_t B,
                    int64_t T, const int32_t num_components[3], const uint32_t key[2], const float opt_args[2],
                    const bf_carry* carry, const bf_out_desc* out, int32_t* leaf_idx, int32_t variant, void* stream);

/* utils.optimal_resampling(weights, N, key) (utils.py:216-244, Fearnhead & Clifford) for B weight vectors of
 * length M <= 1024 (one wave segment per vector up to 64, one workgroup beyond): d_weights [B][M] -> d_idx [B][N] (indices into the M particles), d_weights_out [B][N]. */
int bf_optimal_resample_f32(const float* d_weights, const uint32_t key[2], int64_t B, int32_t M, int32_t N, int32_t* d_idx,
                            float* d_weights_out, void* stream);

/* Moment-matching collapse of the mixture posterior per (trajectory, step): gaussfiltax/utils.py:10-18
 * and the point estimate sum_k w_k m_k (docs/experiments/BOT_Experiment_script.py:101).  weights /
 * means / covs are the strided streams a filter emitted (covs may be NULL when cov_out is NULL);
 * mean_out [B][T][n], cov_out [B][T][n][n] contiguous DEVICE buffers (either may be NULL). */
int bf_collapse_f32(const bf_stream* weights, const bf_stream* means, const bf_stream* covs, int64_t B, int64_t T,
                    int32_t K, int32_t n, float* mean_out, float* cov_out, void* stream);

/* ---- bootstrap particle filter ------------------------------------------------------- */
/* ParamsBPF (gaussfiltax/models.py:55-84): the state-space model plus the Gaussian emission
 * log-density  MVN(h(x, r_eval, u), lp_cov).log_prob(y)  (the form of every `*lp` function of the
 * reference's scripts, e.g. gaussfiltax/nonlinearities.py:51-52) and the initial law N(m0, P0).
 * HOST pointers. */
typedef struct bf_bpf_model {
  bf_model ssm;
  const float* m0;     /* [n]      */
  const float* P0;     /* [n,n]    */
  const float* lp_cov; /* [m,m] covariance R of the emission log-density (the stochastic-volatility
                        * emission uses M(x,u) R M(x,u)^T, adaptive_experiment.py:55-57)          */
  const float* r_eval; /* [dr] noise value h is evaluated at (NULL = zeros)     */
  const float* lp_theta; /* parameters of a log-density given as source (bf_user_model_create_lp), else NULL */
  int32_t n_lp_theta;
} bf_bpf_model;

/* Scan carry (weights, particles, key) of inference.py:1364 for chunked runs; DEVICE pointers,
 * contiguous [B][N][n], [B][N], [B][2].  x_in == NULL: draw the initial particles (:1369-1373). */
typedef struct bf_bpf_carry {
  const float* x_in;
  const float* w_in;
  const uint32_t* key_in;
  float* x_out;
  float* w_out;
  uint32_t* key_out;
} bf_bpf_carry;

/* Outputs (DEVICE pointers, NULL = not emitted).  weights / particles are what the reference
 * returns (inference.py:1359-1362): element (b,i,t) at weights[b*w_sB + i*w_sN + t*w_sT],
 * (b,i,t,d) at particles[b*x_sB + i*x_sN + t*x_sT + d]; ancestors share the weights' strides.
 * The per-step summaries are contiguous [B][T][n] / [B][T]. */
typedef struct bf_bpf_out {
  float* weights;
  int64_t w_sB, w_sN, w_sT;
  float* particles;
  int64_t x_sB, x_sN, x_sT;
  int32_t* ancestors;
  float* mean;      /* sum_i w_i x_i of the emitted weights / particles */
  float* ess;       /* 1 / sum w^2 before the resampling decision       */
  float* logz;      /* log sum_i w_{t-1,i} p(y_t | x_i)                 */
  float* resampled; /* 1.0 where the step resampled                     */
} bf_bpf_out;

/* Batched bootstrap particle filter: replaces the lax.scan of bootstrap_particle_filter
 * (inference.py:1330-1377) with _resample (utils.py:207-214).  One workgroup per
 * trajectory: N <= 4096 particles (16 384 for state / noise dimensions <= 4) stay in the workgroup's
 * registers; larger N up to 2^20 keep the particles in HBM scratch (stream-ordered allocation) and run
 * the same tree orders chunk by chunk (a workgroup per trajectory, or per chunk when B < 128).  key = {hi, lo} of jr.PRNGKey (used for every
 * trajectory unless carry->key_in is given).  resampler: 0 = multinomial inverse-CDF (the
 * reference's jr.choice), 1 = systematic. */
int bf_bpf_f32(const bf_bpf_model* model, const bf_cstream* y, const bf_cstream* u, int64_t B, int64_t T, int32_t N,
               const uint32_t key[2], float ess_threshold, int32_t resampler, const bf_bpf_carry* carry,
               const bf_bpf_out* out, void* stream);

/* Synthetic data: NonlinearSSM.sample (gaussfiltax/models.py:240-289) for B trajectories, one key
 * per trajectory (d_keys [B][2], DEVICE).  model->ssm carries f, h, q0, Q, r0, R; model->m0 / P0 the
 * initial law (lp_cov / r_eval unused).  d_states [B][T][n], d_emissions [B][T][m] contiguous
 * DEVICE buffers (either may be NULL). */
int bf_sample_ssm_f32(const bf_bpf_model* model, const uint32_t* d_keys, const bf_cstream* u, int64_t B, int64_t T,
                      float* d_states, float* d_emissions, void* stream);

/* Index draw of utils.py:210 alone: d_idx[b][i] = choice(d_keys[b], N, (N,), p = d_w[b]). */
int bf_resample_f32(const float* d_w, const uint32_t* d_keys, int64_t B, int32_t N, int32_t resampler,
                    int32_t* d_idx, void* stream);

/* The canonical fp32 arithmetic of the particle filter's weight path (csrc/bf_canon_math.hpp: IEEE add / mul / div /
 * sqrt / fma and integer operations in a fixed order, identical on host and device, restated by the oracle so that
 * in-filter resampling indices are bit-exact), evaluated elementwise: op 0 = log, 1 = exp, 2 = jax.random.normal's
 * bits -> N(0,1) map (the input words are the raw uint32 bits), 3 = sin, 4 = cos, 5 = atan2(in[i], in[n + i]) (the input
 * holds 2 n values).  on_device = 0: HOST buffers, evaluated by the host
 * build of the same functions; 1: DEVICE buffers, one lane per element.  A test / audit hook. */
int bf_canon_eval_f32(int32_t op, const float* in, int64_t n, float* out, int32_t on_device, void* stream);

/* jax.random.normal(key, (count,)) for the default threefry PRNG, written to a HOST buffer
 * (used for the reference's fixed `MVN(m0, P0).sample(K, PRNGKey(0))` draw of the initial
 * component means, inference.py:367: m0 + chol(P0) z).  key = {hi, lo} as jr.PRNGKey. */
int bf_random_normal_f32(const uint32_t key[2], int64_t count, float* host_out);
/* jax.random.split(key, num) -> num keys (2 words each) in a HOST buffer. */
int bf_random_split(const uint32_t key[2], int64_t num, uint32_t* host_out);

/* ---- multi-GPU: the path's one exchange step --------------------------------
 * Trajectories are independent: the batch axis is sharded contiguously over the ranks (one process per GPU), every
 * rank calls the bf_* entry points above on its own shard with no traffic during the scan, and the per-trajectory
 * posterior summaries are exchanged once: d_send (this rank's `bytes` bytes, DEVICE) -> d_recv (world_size * bytes,
 * rank-major, DEVICE) with one ncclAllGather on `stream` over the caller's communicator (`nccl_comm` = ncclComm_t as
 * void*; RCCL over xGMI).  RCCL is loaded on first use.  The Python layer does the same through torch.distributed
 * (bayesianfiltering_amd/distributed.py: backend "nccl" is RCCL). */
int bf_allgather_summaries(const void* d_send, void* d_recv, size_t bytes, void* nccl_comm, void* stream);

/* Bytes one (trajectory, timestep) moves for the streams enabled in `out`: the algorithmic
 * traffic figure of SURVEY.md 8(d)  (4m + 4K(1 + 2n + 2n^2) for all five streams). */
int64_t bf_bytes_per_step(int32_t n, int32_t m, int32_t K, const bf_out_desc* out);

#ifdef __cplusplus
}
#endif
#endif /* BAYESFILT_H_ */
