/*
 * blr_mi355x.h -- C ABI of the MI355X (gfx950) implementation of the posterior / logpdf / marginals /
 * rand hot path of BayesianLinearRegressors.jl.
 *
 * The reference has no FFI boundary: its "operator API" is Julia dispatch on
 * FiniteGP{<:BayesianLinearRegressor} (reference src/bayesian_linear_regression.jl:33-69).  The entry
 * points below are what a `ccall` from those methods binds; each one cites the reference lines it
 * replaces.  INTEGRATION.md shows the Julia side.
 *
 * Conventions
 *   - everything is column-major (Julia native); sizes/strides are int64_t (Julia Int), in ELEMENTS;
 *   - `_f64` / `_f32` suffix = element type of X, y, s, mw, Lw and of the array outputs;
 *     the log marginal likelihood is ALWAYS double (reference :84 promotes through log(2pi)::Float64);
 *   - the caller owns every buffer; no pointer is retained after a call returns;
 *   - calls are synchronous w.r.t. the handle's stream unless the handle was put in async mode
 *     (blr_set_async): then DEVICE-memspace calls only enqueue and the caller synchronises;
 *   - no C++ exception crosses this boundary.
 *
 * Return codes (LAPACK `info` semantics)
 *      0  success
 *     >0  (single-problem calls) Cholesky broke at leading minor k: the matrix is not positive
 *         definite -> the Julia shim throws PosDefException(k), as `cholesky` at reference :78/:86 would
 *     <0  argument -k is invalid (shape/stride/enum/NULL) -> DimensionMismatch / ErrorException
 *         (the reference's own checks are :74 and :26-31)
 *  <= -1000  HIP runtime failure: -(1000 + hipError_t); text via blr_last_error()
 *   Batched calls fill info[B] per regressor (0 / k>0) and return 0 when the launch itself succeeded:
 *   one non-SPD regressor does not poison the batch.
 */
#ifndef BLR_MI355X_H
#define BLR_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLR_ABI_VERSION 1

/* layout of X -- mirrors x_as_colvecs, reference :20-31 (index/shape work, bit-exact) */
#define BLR_LAYOUT_COLVECS 0 /* X is D x N column-major: element (d,n) at X[d + n*ldx], ldx >= D   (:22) */
#define BLR_LAYOUT_ROWVECS 1 /* X is N x D column-major: element (d,n) at X[n + d*ldx], ldx >= N   (:24) */

/* observation-noise covariance Sigma_y (FiniteGP field, reference :79) */
#define BLR_NOISE_ISOTROPIC 0 /* s points to ONE variance  (AbstractGPs f(x, sigma^2))  */
#define BLR_NOISE_DIAGONAL 1  /* s[N] variances            (Diagonal(v))                */
#define BLR_NOISE_DENSE 2     /* N x N symmetric matrix (upper triangle read, as LAPACK 'U'), column-major: only the
                                 *_dense_noise entry points and blr_mean_and_cov_* take it (moderate N, <= 16384) */

/* prior precision Lambda_w (struct field, reference :11-14; _cholesky at :78) */
#define BLR_PRIOR_DENSE 0        /* Lw: D x D symmetric, UPPER triangle read (as LAPACK potrf 'U'), ldl >= D */
#define BLR_PRIOR_UPPER_FACTOR 1 /* Lw: upper factor U with Lambda_w = U'U (PDMat / previous T, :93); strictly-lower part ignored */
#define BLR_PRIOR_DIAGONAL 2     /* Lw: d[D], Lambda_w = Diagonal(d); ldl ignored */

/* where the pointers of a call live */
#define BLR_MEM_HOST 0   /* host pointers: the library stages through its own device workspace */
#define BLR_MEM_DEVICE 1 /* device pointers (hipMalloc / torch / CUDA.jl-style allocator)       */

typedef struct blr_handle blr_handle;

/* ---- lifetime ------------------------------------------------------------------------------- */
int blr_abi_version(void);
int blr_device_count(void);                       /* number of visible HIP devices (0 if none)          */
int blr_create(int device, blr_handle** out);     /* one handle per Julia task / thread                  */
int blr_destroy(blr_handle* h);
const char* blr_last_error(blr_handle* h);        /* valid until the next call on h; never NULL         */
/* A handle's work is ordered by ONE stream at a time: switching streams drains the old one first.  Not supported: capturing a
 * handle's launches into a HIP graph and replaying them, and calls on one handle from two streams / threads at once (the large-D
 * factorisation keeps arrival counters and tagged exchange words per handle) -- use one handle per stream.  The stream being
 * left must still be alive when blr_set_stream / blr_reset_stream is called (it is drained); if it has been destroyed already the
 * whole device is drained instead and the switch still happens. */
int blr_set_stream(blr_handle* h, void* hip_stream); /* run on the caller's hipStream_t; NULL = the HIP null stream */
int blr_reset_stream(blr_handle* h);              /* back to the handle's own (non-blocking) stream      */
int blr_set_async(blr_handle* h, int async);      /* 1: DEVICE-memspace calls return after enqueue       */
int blr_synchronize(blr_handle* h);
/* Run-time switches of the handle (A/B measurements and tests; the defaults are the measured best).  `key` is one of NO_LDSDMA,
 * NO_WAVE_KERNEL, NO_GRAM_RING, NO_DIAG_SPLIT, NO_XCD_SWIZZLE, NO_MFMA_PROJECT, NO_I8_GRAM, NO_I8_DIAG, NO_I8_FACTOR, NO_I8_ROWVECS, NO_I8_DENSE, NO_I8_FALLBACK, NO_BF16X3, NO_PLANES, NO_FP16_PLANES, PLANES8, NO_SPEC_ROWMAX, NO_MULTI_PLANES, NO_MARG_GEMM, NO_GRAD_GEMM, PLAN_DEBUG (flags: any non-empty value = on),
 * WAVE_SPLIT = 1|2|4, CHAIN_BATCH = 1..128, CHAIN_WS_MB, I8_PROBE_MIN = 256..2^20, I8_GROUPS = 6|7, SWEEP = always|never|auto, GRAM_SPLITS = "o,d[,nlong]" (README.md);
 * a "BLR_MI355X_" prefix is accepted.  value NULL or "" restores the built-in default.  The environment variables
 * BLR_MI355X_<KEY> are read ONCE, by blr_create -- no entry point reads the environment.  -> 0, or -2 / -3 (unknown key /
 * malformed value). */
int blr_set_option(blr_handle* h, const char* key, const char* value);
/* Which kernel family the most recent blr_posterior_* / blr_logpdf_* dispatch of this handle launched -- what a profile of the call
 * shows: "fused_i8_kernel", "fused_small_kernel<double, 8, 4>", "fused_wave_kernel<double, 4, 1>", "gram_tile_kernel<float>" (the
 * large-D pipeline), ...; "none" before the first call.  The pointer stays valid until the next call on the handle.  (bench.py labels
 * its roofline with it instead of re-deriving the dispatcher's decision.)  The int8 route decides on the device what it keeps: when
 * it handed more than half of the call's regressors back to the fp64 kernel (heavy-tailed inputs: the probe slice sends the rest of
 * the batch there) the answer is "fused_small_kernel<double, 8, 4> (int8 route handed back K of B)" -- after an int8-route call this
 * function therefore drains the handle's stream. */
const char* blr_last_route(blr_handle* h);
/* Counters of the handle since blr_create / blr_reset_stats.  key: "i8_regressors" = regressors sent down the int8-sliced Gram route
 * (blr_posterior_batched_f64 above); "i8_handed_back" = those of them that route could not finish (rows outgrowing their scale
 * beyond what it corrects in place, non-finite inputs) and the fp64 kernel redid inside the same call -- each of these cost two
 * passes over its data (reading it synchronises the handle's stream); "planes_redone" = fp32 updates at D > 128 (ColVecs) whose
 * SAMPLED row scales did not hold -- the operand planes of the Gram product are scaled per row by a power of two taken from the first 32
 * columns of every column chunk (doubled: an entry up to 16 x the sample's largest still fits); an entry beyond that sends the call
 * through the exact row maxima and the planes pass a second time (one more read of X; option NO_SPEC_ROWMAX = 1 always takes the exact
 * maxima: one more read of X on every call); "workspace_bytes" = device scratch the handle holds now.
 * -> 0, -2 unknown key, -3 NULL value. */
int blr_get_stat(blr_handle* h, const char* key, int64_t* value);
int blr_reset_stats(blr_handle* h);
/* The handle's device scratch (factorisation workspaces of D > 128 calls -- up to 8 GiB for a large batched call, see CHAIN_WS_MB --
 * the feature matrix of blr_posterior_rff_*, the int8 / marginal side buffers) only ever GROWS between calls; this drains the
 * stream and frees all of it.  The next call allocates what it needs again. */
int blr_release_workspace(blr_handle* h);

/* ---- device memory helpers (so a host language needs no HIP binding of its own) -------------- */
int blr_device_alloc(blr_handle* h, size_t bytes, void** dptr);
int blr_device_free(blr_handle* h, void* dptr);
int blr_memcpy_h2d(blr_handle* h, void* dst_device, const void* src_host, size_t bytes);
int blr_memcpy_d2h(blr_handle* h, void* dst_host, const void* src_device, size_t bytes);

/* ---- timing helpers: HIP events on the handle's stream (bench.py roofline leg) ---------------- */
int blr_timer_start(blr_handle* h);
int blr_timer_stop(blr_handle* h, float* elapsed_ms); /* records, synchronises, returns the interval */

/* ---- fused inference: replaces __compute_inference_quantities + logpdf + posterior ------------
 * reference src/bayesian_linear_regression.jl:72-89 (shared quantities), :55-58 (logpdf),
 * :60-69 (posterior), :92-93 (__build_Lambda).  One pass produces everything both need:
 *   A = Lw + X S X'   T = chol(A).U   mw' = mw + A^-1 X S (y - X'mw)
 *   logpdf = -1/2 [N log 2pi + logdet Sy + d'Sd + logdet A - logdet Lw - |T^-T b|^2]
 * Outputs (each may be NULL to skip it):
 *   mw_post[D]; T_post D x D upper factor (strictly-lower part written as zero), ldt >= D;
 *   Lw_post D x D full symmetric A, ldlp >= D; logpdf (double).
 * In-place form: with prior_kind = BLR_PRIOR_UPPER_FACTOR, mw_post == mw and T_post == Lw are allowed (every read of the
 * prior state completes before the first write; blr_update_factor_* relies on it).  mw_post and T_post of a regressor whose
 * info != 0 are left untouched at every D (Lw_post may already hold A): a resident state survives a bad batch.
 * Batched form: regressor i reads X + i*strideX, y + i*stridey, s + i*strides, mw + i*stridemw,
 * Lw + i*strideLw and writes the outputs at their strides; a stride of 0 shares an input.
 * Numerics: results are bit-reproducible from call to call (fixed accumulation order, no floating-point atomics).  fp64
 * accuracy against the reference's op sequence in LAPACK: evidence 1e-10 relative, mw', T, Lw' 1e-9 (tests/test_gpu_parity.py).
 * One shape takes a different route to the same numbers: D = 128 in fp64 -- aligned ColVecs or RowVecs (16-byte aligned rows), isotropic or
 * diagonal noise, any prior kind, 512 <= N <= 16384 (+ a last partial block of up to 31 columns, added in fp64) -- forms X X' on the int8
 * matrix cores from an exact 48-bit splitting of the inputs against per-row power-of-two scales (csrc/blr_fused_i8.hpp); everything after
 * the Gram matrix is fp64 as elsewhere.  Its error model, stated BEFORE the tests that hold it to it (tests/test_gpu_parity.py test_i8_*,
 * test_c2_shape_fp64):
 *   - entries of Lw' within 1e-13 of sqrt(Lw'_ii Lw'_jj) (measured: 3e-14) instead of a few ulp of themselves -- an entry that nearly
 *     cancels is off by that much of its row's and column's scale, not of itself;
 *   - the evidence within the 1e-10 above; where delta'Sigma^-1 delta and |T^-T b|^2 cancel (data explained by the weights) the error is
 *     1e-14 of delta'Sigma^-1 delta: |d logpdf| <= 1e-11 |logpdf| + 1e-14 delta'Sigma^-1 delta is what the tests assert;
 *   - inputs whose low mantissa bits are zero (float32 values, integers, powers of two) are covered: the digits are BALANCED (bytes of
 *     the integer + 0x8080808080, minus 128), so a zero low digit is 0 and its truncated products vanish; what is truncated is zero-mean
 *     unless the low digits of a row are a constant non-zero pattern (every entry = integer + 1/3): 1e-13 there (tools/i8_digits_emul.py);
 *   - six digit groups under isotropic noise, seven under diagonal noise (the rows' bounds are then bounds of x times the LARGEST
 *     1 / sqrt(s_n), loose by the spread of the variances).  Option I8_GROUPS = 7 keeps the seventh group under isotropic noise too:
 *     entries within 1e-14 (measured) instead of 3e-14 of their scale at 0.8 x the rate; I8_GROUPS = 6 drops it under diagonal noise
 *     where the variances are of one magnitude: 3e-14 x (largest / typical 1 / sqrt(s_n))^2, 1.2 x the rate;
 *   - an entry that outgrows its row's scale (taken from the first 96 columns, 2 - 4 x their largest entry) is corrected in fp64 inside the
 *     kernel; a regressor with more than one such 32-column block in 16 (heavy-tailed features), Inf / NaN, or a prior mean that explains
 *     the data to three digits is REDONE on the fp64 matrix pipe inside the same call -- it then costs two passes; blr_get_stat(h,
 *     "i8_handed_back", ..) counts them, and a batch of more than 4096 regressors (option I8_PROBE_MIN) whose first 256 were handed back by more than a quarter
 *     sends the rest to the fp64 kernel directly.
 * blr_set_option(h, "NO_I8_GRAM", "1") keeps every regressor on the fp64 matrix pipe.
 * fp32, D > 128, aligned ColVecs: the Gram matrix is formed on the bf16 matrix cores from an EXACT three-way split of every fp32 operand
 * (x = h + m + l, each rounded to nearest), keeping the six products hh, hm, mh, mm, hl, lh under fp32 accumulation: the dropped ones are
 * 2^-24 of |a||b| each and zero-mean, the result is as accurate as an fp32 fma chain (tools/bf3_unit.hip; tests hold A to 4 x the error
 * of fp32 LAPACK on the same inputs, test_c3_full_size / test_c5_full_size).  blr_set_option(h, "NO_BF16X3", "1") uses the fp32 matrix
 * instruction instead.
 */
int blr_posterior_batched_f64(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N,
                              const double* X, int64_t ldx, int64_t strideX,
                              const double* y, int64_t stridey,
                              int noise_kind, const double* s, int64_t strides,
                              int prior_kind, const double* mw, int64_t stridemw,
                              const double* Lw, int64_t ldl, int64_t strideLw,
                              double* mw_post, int64_t stride_mwpost,
                              double* T_post, int64_t ldt, int64_t strideT,
                              double* Lw_post, int64_t ldlp, int64_t strideLp,
                              double* logpdf, int32_t* info);
int blr_posterior_batched_f32(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N,
                              const float* X, int64_t ldx, int64_t strideX,
                              const float* y, int64_t stridey,
                              int noise_kind, const float* s, int64_t strides,
                              int prior_kind, const float* mw, int64_t stridemw,
                              const float* Lw, int64_t ldl, int64_t strideLw,
                              float* mw_post, int64_t stride_mwpost,
                              float* T_post, int64_t ldt, int64_t strideT,
                              float* Lw_post, int64_t ldlp, int64_t strideLp,
                              double* logpdf, int32_t* info);

/* single regressor, host pointers; returns info (see "Return codes") */
int blr_posterior_f64(blr_handle* h, int layout, int64_t D, int64_t N, const double* X, int64_t ldx,
                      const double* y, int noise_kind, const double* s,
                      int prior_kind, const double* mw, const double* Lw, int64_t ldl,
                      double* mw_post, double* T_post, int64_t ldt, double* Lw_post, int64_t ldlp,
                      double* logpdf);
int blr_posterior_f32(blr_handle* h, int layout, int64_t D, int64_t N, const float* X, int64_t ldx,
                      const float* y, int noise_kind, const float* s,
                      int prior_kind, const float* mw, const float* Lw, int64_t ldl,
                      float* mw_post, float* T_post, int64_t ldt, float* Lw_post, int64_t ldlp,
                      double* logpdf);

/* ---- marginal stream: replaces mean (:33), var (:40-43), mean_and_var (:47) --------------------
 *   mean_n = x_n' mw          var_n = |Uw^-T x_n|^2 + Sy_nn
 * mean or var may be NULL (mean-only = evaluating a function sample, sampling_functions.jl:17-19).
 */
int blr_marginals_batched_f64(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N,
                              const double* X, int64_t ldx, int64_t strideX,
                              int noise_kind, const double* s, int64_t strides,
                              int prior_kind, const double* mw, int64_t stridemw,
                              const double* Lw, int64_t ldl, int64_t strideLw,
                              double* mean, int64_t stridemean, double* var, int64_t stridevar,
                              int32_t* info);
int blr_marginals_batched_f32(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N,
                              const float* X, int64_t ldx, int64_t strideX,
                              int noise_kind, const float* s, int64_t strides,
                              int prior_kind, const float* mw, int64_t stridemw,
                              const float* Lw, int64_t ldl, int64_t strideLw,
                              float* mean, int64_t stridemean, float* var, int64_t stridevar,
                              int32_t* info);

/* ---- draws: replaces rand (:49-53) and the weight draws of sampling_functions.jl:29,35,44 ------
 * The host keeps drawing the normals so its RNG stream is the reference's:
 *   Z1 = randn(rng, D, S) FIRST, then Z2 = randn(rng, N, S).
 *   W = mw .+ Uw \ Z1 ;  Y = X'W .+ sqrt.(s) .* Z2          (Y is N x S, ldy >= N)
 */
int blr_rand_f64(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S,
                 const double* X, int64_t ldx, int noise_kind, const double* s,
                 int prior_kind, const double* mw, const double* Lw, int64_t ldl,
                 const double* Z1, int64_t ldz1, const double* Z2, int64_t ldz2,
                 double* Y, int64_t ldy);
int blr_rand_f32(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S,
                 const float* X, int64_t ldx, int noise_kind, const float* s,
                 int prior_kind, const float* mw, const float* Lw, int64_t ldl,
                 const float* Z1, int64_t ldz1, const float* Z2, int64_t ldz2,
                 float* Y, int64_t ldy);
/*   Y = X'W for S GIVEN weight vectors W (D x S, ldw >= D): evaluation of S function samples at the inputs,
 *   replaces (s::BLRFunctionSample)(X) = ϕ(X)'s.w  (src/sampling_functions.jl:16-18) for a batch of samples; also the two
 *   large products of the reverse-mode rule of rand (README.md:56-60: W̄ = X Ȳ and X̄ = W Ȳ', both "apply" calls on
 *   re-interpreted layouts -- julia/BLRMI355X.jl rand_pullback).  Y is N x S, ldy >= N. */
int blr_apply_weights_f64(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S,
                          const double* X, int64_t ldx, const double* W, int64_t ldw, double* Y, int64_t ldy);
int blr_apply_weights_f32(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S,
                          const float* X, int64_t ldx, const float* W, int64_t ldw, float* Y, int64_t ldy);
/*   W = mw .+ Uw \ Z   (D x S) */
int blr_sample_weights_f64(blr_handle* h, int memspace, int64_t D, int64_t S,
                           int prior_kind, const double* mw, const double* Lw, int64_t ldl,
                           const double* Z, int64_t ldz, double* W, int64_t ldw);
int blr_sample_weights_f32(blr_handle* h, int memspace, int64_t D, int64_t S,
                           int prior_kind, const float* mw, const float* Lw, int64_t ldl,
                           const float* Z, int64_t ldz, float* W, int64_t ldw);

/* ---- random-Fourier basis (BASELINE config 5): phi(x) = scale * cos(Omega' x + phase) ---------------------
 * The reference's BasisFunctionRegressor takes any callable phi (src/basis_function_regression.jl:7-9,41) and ships
 * none; this is the feature map of config 5, applied on the device so Phi never crosses PCIe.
 *   Xin: Din x N column-major (ColVecs of the raw inputs), Omega: Din x D column-major, phase[D];
 *   Phi : D x N column-major (ldphi >= D).
 * blr_posterior_rff_* = features + fused inference in one call (Phi lives in the handle's workspace):
 * the remaining arguments are those of blr_posterior_batched_* with B = 1.
 */
int blr_rff_features_f64(blr_handle* h, int memspace, int64_t Din, int64_t D, int64_t N,
                         const double* Xin, int64_t ldxin, const double* Omega, int64_t ldo, const double* phase,
                         double scale, double* Phi, int64_t ldphi);
int blr_rff_features_f32(blr_handle* h, int memspace, int64_t Din, int64_t D, int64_t N,
                         const float* Xin, int64_t ldxin, const float* Omega, int64_t ldo, const float* phase,
                         float scale, float* Phi, int64_t ldphi);
int blr_posterior_rff_f64(blr_handle* h, int memspace, int64_t Din, int64_t D, int64_t N,
                          const double* Xin, int64_t ldxin, const double* Omega, int64_t ldo, const double* phase,
                          double scale, const double* y, int noise_kind, const double* s,
                          int prior_kind, const double* mw, const double* Lw, int64_t ldl,
                          double* mw_post, double* T_post, int64_t ldt, double* Lw_post, int64_t ldlp,
                          double* logpdf, int32_t* info);
int blr_posterior_rff_f32(blr_handle* h, int memspace, int64_t Din, int64_t D, int64_t N,
                          const float* Xin, int64_t ldxin, const float* Omega, int64_t ldo, const float* phase,
                          float scale, const float* y, int noise_kind, const float* s,
                          int prior_kind, const float* mw, const float* Lw, int64_t ldl,
                          float* mw_post, float* T_post, int64_t ldt, float* Lw_post, int64_t ldlp,
                          double* logpdf, int32_t* info);

/* ---- gradient of the log marginal likelihood -------------------------------
 * The reverse-mode rule of logpdf(fx, y) (reference src/bayesian_linear_regression.jl:55-58): what Zygote derives from
 * the reference's Julia code (README.md:56-71, examples/nn-blr.jl:35-37) and a ccall-backed logpdf has to supply
 * itself (a ChainRules rrule in the shim).  One call = fused posterior + two MFMA sweeps per tile of inputs.
 *   logpdf[B]            the value itself
 *   dX                   dL/dX, SAME layout as X (lddx, stridedX)
 *   dy[N], ds[N]         dL/dy_n; dL/ds_n per observation (isotropic noise: the scalar gradient is their sum)
 *   dmw[D]               dL/dmw
 *   mw_post[D], Ainv     posterior mean and A^-1 = (Lw + X S X')^-1 (D x D, ldai): the caller forms
 *                        dL/dLw = -(m m' + Ainv - Lw^-1)/2 with m = mw_post - mw (D x D host work)
 * Any output except logpdf/info may be NULL.  Arguments up to strideLw are those of blr_posterior_batched_*. */
int blr_logpdf_grad_batched_f64(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N,
                                const double* X, int64_t ldx, int64_t strideX, const double* y, int64_t stridey,
                                int noise_kind, const double* s, int64_t strides, int prior_kind, const double* mw,
                                int64_t stridemw, const double* Lw, int64_t ldl, int64_t strideLw, double* logpdf,
                                double* dX, int64_t lddx, int64_t stridedX, double* dy, int64_t stridedy, double* ds,
                                int64_t strideds, double* dmw, int64_t stridedmw, double* mw_post,
                                int64_t stride_mwpost, double* Ainv, int64_t ldai, int64_t strideAi, int32_t* info);
int blr_logpdf_grad_batched_f32(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N,
                                const float* X, int64_t ldx, int64_t strideX, const float* y, int64_t stridey,
                                int noise_kind, const float* s, int64_t strides, int prior_kind, const float* mw,
                                int64_t stridemw, const float* Lw, int64_t ldl, int64_t strideLw, double* logpdf,
                                float* dX, int64_t lddx, int64_t stridedX, float* dy, int64_t stridedy, float* ds,
                                int64_t strideds, float* dmw, int64_t stridedmw, float* mw_post,
                                int64_t stride_mwpost, float* Ainv, int64_t ldai, int64_t strideAi, int32_t* info);

/* ---- shared-X multi-output evidence -------------------------------------------
 * logpdf(fx, Y::AbstractMatrix) of the AbstractGPs secondary API (exercised through TestUtils at reference
 * test/bayesian_linear_regression.jl:7-9): the S columns of Y (N x S, ldY) share X, so the Gram matrix and its
 * Cholesky factor are formed ONCE; per column only X S (y_s - X'mw) (one D x N x S GEMM) and two triangular
 * solves remain.  logpdf[S]; mw_post (D x S, ldmp) optionally receives the posterior mean of every column
 * (NULL: skipped); info: one status for the shared factorisation (LAPACK semantics). */
int blr_logpdf_multi_f64(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const double* X,
                         int64_t ldx, const double* Y, int64_t ldY, int noise_kind, const double* s, int prior_kind,
                         const double* mw, const double* Lw, int64_t ldl, double* logpdf, double* mw_post, int64_t ldmp,
                         int32_t* info);
int blr_logpdf_multi_f32(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const float* X,
                         int64_t ldx, const float* Y, int64_t ldY, int noise_kind, const float* s, int prior_kind,
                         const float* mw, const float* Lw, int64_t ldl, double* logpdf, float* mw_post, int64_t ldmp,
                         int32_t* info);

/* ---- N-sharded single regressor ("one-exchange N-sharding") -------------------------
 * A regressor too large for one GPU's share of time (config 3) splits its N observations over ranks.  Everything the
 * update needs from the data is additive over column blocks:
 *   blr_gram_stats_*            per rank, on its columns: stats = (DP + 128) x DP column-major (lds >= DP + 128,
 *                               DP = 128 ceil(D/128)): lower triangle of X S X' in rows [0, DP), row DP = (X S (y - X'mw))';
 *                               scal[2] = { (y - X'mw)' S (y - X'mw), logdet Sigma_y } of the block.
 *   (host)                      ONE sum over ranks of `stats` and `scal` (RCCL all-reduce through torch.distributed /
 *                               MPI.jl) -- the only exchange.
 *   blr_posterior_from_stats_*  on every rank (redundant D x D work): adds the prior precision (dense or diagonal),
 *                               factorises, solves; outputs as blr_posterior_*.  `stats` is overwritten.
 * Device pointers only.  Works for any D (the large-D pipeline); N_total = observations over all ranks. */
int blr_gram_stats_f64(blr_handle* h, int layout, int64_t D, int64_t N, const double* X, int64_t ldx, const double* y,
                       int noise_kind, const double* s, const double* mw, double* stats, int64_t lds, double* scal);
int blr_gram_stats_f32(blr_handle* h, int layout, int64_t D, int64_t N, const float* X, int64_t ldx, const float* y,
                       int noise_kind, const float* s, const float* mw, float* stats, int64_t lds, double* scal);
int blr_posterior_from_stats_f64(blr_handle* h, int64_t D, int64_t N_total, double* stats, int64_t lds, const double* scal,
                                 int prior_kind, const double* mw, const double* Lw, int64_t ldl, double* mw_post,
                                 double* T_post, int64_t ldt, double* Lw_post, int64_t ldlp, double* logpdf, int32_t* info);
int blr_posterior_from_stats_f32(blr_handle* h, int64_t D, int64_t N_total, float* stats, int64_t lds, const double* scal,
                                 int prior_kind, const float* mw, const float* Lw, int64_t ldl, float* mw_post,
                                 float* T_post, int64_t ldt, float* Lw_post, int64_t ldlp, double* logpdf, int32_t* info);

/* ---- dense noise covariance and full predictive covariance -------------------------
 * reference src/bayesian_linear_regression.jl:79-82 (the general _cholesky(Sigma_y) branch of the shared quantities -- what
 * the reference's own toy problems use, test/test_utils.jl:7-8), :35-38 / :45 (cov, mean_and_cov), :52 (rand).
 * A dense Sigma_y (N x N, ldsy >= N, upper triangle read) is whitened away on the device: blocked Cholesky L L' = Sigma_y with
 * X and y carried through the panel solves (X L^-T, L^-1 y), then the ordinary update with unit noise; logpdf gets
 * -logdet(Sigma_y)/2.  N <= 16384.  Outputs as blr_posterior_*; info > 0: Sigma_y, Lw or Lw + X Sy^-1 X' not positive definite.
 *   blr_mean_and_cov_*   mean[N] (may be NULL) and C = X' Lw^-1 X + Sigma_y as the FULL symmetric N x N matrix (ldc >= N);
 *                        noise_kind ISOTROPIC / DIAGONAL / DENSE (s = the scalar, the N variances, or the N x N matrix with lds)
 *   blr_rand_dense_noise_*   Y = X'(mw + Uw \ Z1) + Us' Z2 with Us = chol(Sigma_y).U; RETURNS info (0, or k > 0) */
int blr_posterior_dense_noise_f64(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, const double* X, int64_t ldx,
                                  const double* y, const double* Sy, int64_t ldsy, int prior_kind, const double* mw,
                                  const double* Lw, int64_t ldl, double* mw_post, double* T_post, int64_t ldt, double* Lw_post,
                                  int64_t ldlp, double* logpdf, int32_t* info);
int blr_posterior_dense_noise_f32(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, const float* X, int64_t ldx,
                                  const float* y, const float* Sy, int64_t ldsy, int prior_kind, const float* mw,
                                  const float* Lw, int64_t ldl, float* mw_post, float* T_post, int64_t ldt, float* Lw_post,
                                  int64_t ldlp, double* logpdf, int32_t* info);
int blr_mean_and_cov_f64(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, const double* X, int64_t ldx,
                         int noise_kind, const double* s, int64_t lds, int prior_kind, const double* mw, const double* Lw,
                         int64_t ldl, double* mean, double* C, int64_t ldc, int32_t* info);
int blr_mean_and_cov_f32(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, const float* X, int64_t ldx,
                         int noise_kind, const float* s, int64_t lds, int prior_kind, const float* mw, const float* Lw,
                         int64_t ldl, float* mean, float* C, int64_t ldc, int32_t* info);
int blr_rand_dense_noise_f64(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const double* X,
                             int64_t ldx, const double* Sy, int64_t ldsy, int prior_kind, const double* mw, const double* Lw,
                             int64_t ldl, const double* Z1, int64_t ldz1, const double* Z2, int64_t ldz2, double* Y, int64_t ldy);
int blr_rand_dense_noise_f32(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const float* X,
                             int64_t ldx, const float* Sy, int64_t ldsy, int prior_kind, const float* mw, const float* Lw,
                             int64_t ldl, const float* Z1, int64_t ldz1, const float* Z2, int64_t ldz2, float* Y, int64_t ldy);

/* ---- rank-k update of a RESIDENT posterior state ----------------------------------------------
 * Replaces: reference test/bayesian_linear_regression.jl:49-70 ("repeated conditioning", posterior(f'1(X2, S2), y2)) and
 * src/bayesian_linear_regression.jl:93 (the posterior carries mw', Lw' forward; every further call re-derives :72-89 from
 * Lw' at O(D^3)).
 * State, updated IN PLACE: mw[B][D] and the upper factor T[B] (D x D, ldt, T'T = precision; the strictly-lower part is not
 * read and may be overwritten with zeros) -- exactly the (mw_post, T_post) pair blr_posterior_batched_* writes.
 * k new observations per regressor: X (D x k ColVecs / k x D RowVecs), y[k], isotropic or diagonal noise s.
 * logpdf[B] (may be NULL) = log p(y_k | state before the call), the evidence increment: summing it over successive calls
 * gives the evidence of all the data (chain rule).  info[B]: 0, or LAPACK-style i > 0 on BOTH routes, checked in the
 * reference's order (:78 prior, :79 noise, :86 posterior): T has a non-positive diagonal entry i (the state is not a Cholesky
 * factor), else s_i is not positive, else the leading minor of order i of the updated precision is not positive definite.
 * Routes (measured; BLR_MI355X_SWEEP=always|never at blr_create, or blr_set_option(h, "SWEEP", ...), overrides, "always" meaning D <= 128 and k <= 16):
 *   k <= 1, D <= 128 (and D > 64 or B < 256): one sweep of D Givens rotations over the factor held in LDS -- O(D^2),
 *     orthogonal transformations only; the state is untouched when info != 0;
 *   otherwise: the same state re-factored in place by blr_posterior_batched_* with the old factor entering as D
 *     pseudo-observations (cost independent of k; that entry point supports mw_post == mw and T_post == Lw for
 *     BLR_PRIOR_UPPER_FACTOR: every read of the old state completes before the first write).  The state is untouched
 *     when info != 0, at every D. */
int blr_update_factor_f64(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t k, const double* X,
                          int64_t ldx, int64_t strideX, const double* y, int64_t stridey, int noise_kind, const double* s,
                          int64_t strides, double* mw, int64_t stridemw, double* T, int64_t ldt, int64_t strideT,
                          double* logpdf, int32_t* info);
int blr_update_factor_f32(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t k, const float* X,
                          int64_t ldx, int64_t strideX, const float* y, int64_t stridey, int noise_kind, const float* s,
                          int64_t strides, float* mw, int64_t stridemw, float* T, int64_t ldt, int64_t strideT,
                          double* logpdf, int32_t* info);

/* ---- sharded log-evidence: fixed-order sum of logpdf[B] on the device ----------
 * Deterministic (no float atomics): the same bits for the same B regardless of launch geometry.
 * The cross-rank step is one RCCL all-gather of these per-rank partials done by the host framework
 * (torch.distributed / MPI.jl); the data path has no other collective. */
int blr_logpdf_sum(blr_handle* h, int memspace, int64_t B, const double* logpdf, double* total);

/* ---- the exchange itself, RCCL called directly (SURVEY.md 8b contract, 8e) ---------------------------------------
 * For hosts without a collective library of their own (a Julia session per GPU): one process per GPU, one handle per
 * process.  Rank 0 obtains 128 opaque bytes with blr_comm_unique_id and ships them to the other ranks by any means (a file,
 * Distributed.jl, MPI, a socket); every rank then calls blr_comm_init(h, nranks, rank, id) -- collective, like
 * ncclCommInitRank.  librccl is loaded on first use (no link-time dependency; a single-GPU host never needs it).
 *   blr_logpdf_allgather_sum   every rank passes its `count` per-regressor log evidences (device); logpdf_all (device,
 *                              nranks * count doubles, rank order == regressor order) receives the all-gather and *total
 *                              (device) the fixed-order sum of it -- the same bits on every rank, and for every rank count
 *                              that gathers the SAME vector (a batch divisible by the rank count; uneven blocks are
 *                              zero-padded to equal counts by the caller, which moves elements between the lanes of the
 *                              sum: the totals then agree to rounding only).
 *                              Without a communicator (nranks = 1) it degenerates to copy + blr_logpdf_sum.
 *   blr_allreduce_sum          in-place sum over ranks of a device buffer (the `stats` / `scal` exchange of the N-sharded
 *                              single regressor below: is_f64 = 0 for float, 1 for double)
 * Errors: -(2000 + ncclResult_t); -2001 when librccl cannot be loaded; text via blr_last_error. */
#define BLR_UNIQUE_ID_BYTES 128
int blr_comm_unique_id(void* id128);
int blr_comm_init(blr_handle* h, int nranks, int rank, const void* id128);
int blr_comm_destroy(blr_handle* h);
int blr_comm_size(blr_handle* h);
int blr_comm_rank(blr_handle* h);
int blr_logpdf_allgather_sum(blr_handle* h, int64_t count, const double* logpdf_local, double* logpdf_all, double* total);
int blr_allreduce_sum(blr_handle* h, int is_f64, void* buf, int64_t count);
/* The N-sharded single regressor in ONE call per rank: blr_gram_stats_* on this rank's N_local columns, the
 * in-place all-reduce of `stats` (lds * DP elements, DP = 128 ceil(D/128)) and `scal` over the handle's communicator, then
 * blr_posterior_from_stats_* -- every rank ends with the same posterior and evidence of all N_total observations.  `stats`
 * ((DP + 128) x DP, lds >= DP + 128) and `scal` (2 doubles) are caller-provided device scratch; all pointers device.
 * Without a communicator it is the single-GPU update through the statistics path.
 * Reproducibility: every rank gets the SAME bits, but -- unlike blr_logpdf_allgather_sum (all-gather + one fixed-order sum) --
 * the statistics go through ncclAllReduce, a floating-point sum whose order depends on the rank count and the ring: results
 * for different numbers of ranks agree to rounding, not bit for bit. */
int blr_posterior_nsharded_f64(blr_handle* h, int layout, int64_t D, int64_t N_local, int64_t N_total, const double* X,
                               int64_t ldx, const double* y, int noise_kind, const double* s, int prior_kind,
                               const double* mw, const double* Lw, int64_t ldl, double* stats, int64_t lds, double* scal,
                               double* mw_post, double* T_post, int64_t ldt, double* Lw_post, int64_t ldlp, double* logpdf,
                               int32_t* info);
int blr_posterior_nsharded_f32(blr_handle* h, int layout, int64_t D, int64_t N_local, int64_t N_total, const float* X,
                               int64_t ldx, const float* y, int noise_kind, const float* s, int prior_kind, const float* mw,
                               const float* Lw, int64_t ldl, float* stats, int64_t lds, double* scal, float* mw_post,
                               float* T_post, int64_t ldt, float* Lw_post, int64_t ldlp, double* logpdf, int32_t* info);

#ifdef __cplusplus
}
#endif
#endif /* BLR_MI355X_H */
