"""CPU oracle for the BayesianLinearRegressors.jl posterior / logpdf / marginals / rand path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker (or as the timed CPU baseline), never as the thing shipped.  The product path
(``bayesianlinearregressors.jl_amd``) never imports this module and has no CPU fallback.

What this is
------------
A restatement, in NumPy/SciPy on LAPACK (the same potrf/trtrs/syrk family Julia's LinearAlgebra
calls), of the reference's algorithm.  Two forms are provided and the tests assert they agree:

* ``*_literal`` functions follow the reference's operation sequence line by line
  (``/root/reference/src/bayesian_linear_regression.jl:33-93``).
* ``*_direct`` functions use the algebraically equivalent "Gram" form the HIP kernels compute
  (SURVEY.md section 0.1):  A = Lw + X S X',  T = chol(A).U,  mw' = mw + A^-1 X S (y - X'mw).

Pinning status
--------------
The reference is Julia and cannot be run in the build container (no ``julia``; dependencies not
vendored), and its tests draw every toy problem from ``MersenneTwister(123456)``, whose ``randn``
stream is not reproducible without Julia.  The oracle is therefore pinned against everything in the
reference's own tests that does not depend on that stream (tests/test_oracle_pins.py):

* the only literal golden vector in the repository, the doctest at
  ``/root/reference/src/basis_function_regression.jl:11-28``  (var = [2.0, 1.25, 1.0, 1.25, 2.0]);
* the known-answer identities of ``/root/reference/test/bayesian_linear_regression.jl``:
  naive N x N Gaussian logpdf (:22-38), low-noise interpolation (:40-48), repeated conditioning
  (:49-70), PDMat/Symmetric equivalence (:90-112), unknown-container error (:116-122);
* BFR == BLR o phi (``/root/reference/test/basis_function_regression.jl:13-28``);
* layout independence of function samples (``/root/reference/test/sampling_functions.jl:8-15``);
* a 50-digit mpmath evaluation of the N x N Gaussian formula on the committed fixtures.

Conventions (all follow the reference): X is D x N ("ColVecs": columns are inputs), the prior is
w ~ N(mw, inv(Lw)) with *precision* Lw, noise covariance Sy is N x N SPD (dense, 1-D vector of
variances = Diagonal, or a scalar = isotropic).  "U" always denotes an upper Cholesky factor.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla

LOG2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------------
# input containers: /root/reference/src/bayesian_linear_regression.jl:20-31
# --------------------------------------------------------------------------------------------
class ColVecs:
    """KernelFunctions.ColVecs: a D x N matrix whose columns are the inputs."""

    def __init__(self, X):
        self.X = np.asarray(X)

    def __len__(self):
        return self.X.shape[1]


class RowVecs:
    """KernelFunctions.RowVecs: an N x D matrix whose rows are the inputs."""

    def __init__(self, X):
        self.X = np.asarray(X)

    def __len__(self):
        return self.X.shape[0]


def x_as_colvecs(x):
    """bayesian_linear_regression.jl:20-31.  ColVecs -> itself, RowVecs -> lazy adjoint,
    a raw matrix is what AbstractGPs turns into ColVecs at f(X, ...) time, anything else errors."""
    if isinstance(x, ColVecs):
        return x.X
    if isinstance(x, RowVecs):
        return x.X.T
    if isinstance(x, np.ndarray) and x.ndim == 2:
        return x
    raise TypeError(
        f"{type(x).__name__} is not a subtype of AbstractVector that is known. "
        "Please provide either a ColVecs or RowVecs."
    )


# --------------------------------------------------------------------------------------------
# _cholesky of the three kinds of matrices the path sees
# --------------------------------------------------------------------------------------------
def chol_upper(A):
    """AbstractGPs._cholesky(A).U for dense / diagonal (1-D) / scalar-times-identity input.
    Raises numpy.linalg.LinAlgError when A is not positive definite (Julia: PosDefException)."""
    A = np.asarray(A)
    if A.ndim == 2:
        return sla.cholesky(A, lower=False, check_finite=False)
    if A.ndim == 1:
        if np.any(A <= 0):
            raise np.linalg.LinAlgError("diagonal matrix is not positive definite")
        return np.diag(np.sqrt(A))
    raise ValueError("chol_upper: expected a matrix or a vector of diagonal entries")


def dense_noise(Sy, N, dtype):
    """Materialise the noise covariance as a dense N x N matrix (test-size problems only)."""
    Sy = np.asarray(Sy, dtype=dtype)
    if Sy.ndim == 0:
        return np.eye(N, dtype=dtype) * Sy
    if Sy.ndim == 1:
        return np.diag(Sy)
    return Sy


def dense_precision(Lw, D, dtype):
    Lw = np.asarray(Lw, dtype=dtype)
    if Lw.ndim == 1:
        return np.diag(Lw)
    return Lw


def _solve_tri(U, B, trans):
    """U' \\ B (trans=True) or U \\ B (trans=False) with U upper triangular (LAPACK trtrs)."""
    return sla.solve_triangular(U, B, lower=False, trans=1 if trans else 0, check_finite=False)


# --------------------------------------------------------------------------------------------
# moments: bayesian_linear_regression.jl:33-47
# --------------------------------------------------------------------------------------------
def mean(mw, X):
    """:33  mean(fx) = X' mw."""
    return X.T @ mw


def cov(mw, Lw, X, Sy):
    """:35-38  alpha = Uw' \\ X ;  Symmetric(alpha' alpha + Sy)."""
    D, N = X.shape
    Uw = chol_upper(dense_precision(Lw, D, X.dtype))
    alpha = _solve_tri(Uw, X, trans=True)
    return alpha.T @ alpha + dense_noise(Sy, N, X.dtype)


def var(mw, Lw, X, Sy):
    """:40-43  vec(sum(abs2, Uw' \\ X; dims=1)) .+ diag(Sy)."""
    D, N = X.shape
    Uw = chol_upper(dense_precision(Lw, D, X.dtype))
    alpha = _solve_tri(Uw, X, trans=True)
    return np.sum(alpha * alpha, axis=0) + np.diag(dense_noise(Sy, N, X.dtype))


def rand(mw, Lw, X, Sy, Z1, Z2):
    """:49-53 with the two randn draws supplied by the caller in the reference's order:
    Z1 = randn(rng, D, S) first, then Z2 = randn(rng, N, S)."""
    D, N = X.shape
    Uw = chol_upper(dense_precision(Lw, D, X.dtype))
    w = mw[:, None] + _solve_tri(Uw, Z1, trans=False)
    Us = chol_upper(dense_noise(Sy, N, X.dtype))
    return X.T @ w + Us.T @ Z2


def rand_pullback(mw, Lw, X, Sy, Z1, Ybar):
    """Reverse-mode rule of `rand` (what Zygote derives through :49-53, README.md:56-60) for diagonal / isotropic noise, with
    the draws (Z1, Z2) held fixed:  Y = X'W + sqrt.(s) .* Z2,  W = mw .+ Uw \ Z1,  Uw = chol(Lw).U.
        Wbar = X Ybar,  Xbar = W Ybar',  mwbar = Wbar 1,  Ubar = -triu(Uw^-T Wbar V'),  V = Uw \ Z1,
        Lwbar: through the Cholesky (dense Lw) / the diagonal (Lw a vector);  sbar_n = sum_s Ybar[n, s] Z2[n, s] / (2 sqrt(s_n))
    is left to the caller (it needs Z2 only).  Returns dict(X, mw, Lw, U, W).  Pinned by finite differences of `rand` in
    tests/test_oracle_pins.py."""
    D, N = X.shape
    dt = X.dtype
    Lw = np.asarray(Lw, dtype=dt)
    Uw = chol_upper(dense_precision(Lw, D, dt))
    V = _solve_tri(Uw, Z1, trans=False)
    W = mw[:, None] + V
    Wbar = X @ Ybar
    Ubar = -np.triu(_solve_tri(Uw, Wbar, trans=True) @ V.T)
    if Lw.ndim == 1:   # Uw = diag(sqrt(d)): d_bar = Ubar_jj / (2 sqrt(d_j))
        Lbar = np.diag(Ubar) / (2.0 * np.sqrt(Lw))
    else:              # A = L L' with L = Uw': Abar = sym(L^-T Phi(L' Lbar) L^-1), Phi = lower triangle with halved diagonal
        L = Uw.T
        M = np.tril(L.T @ Ubar.T)
        M[np.diag_indices(D)] *= 0.5
        Ab = _solve_tri(Uw, _solve_tri(Uw, M.T, trans=False).T, trans=False)  # L^-T M L^-1 = Uw^-1 (Uw^-1 M')'
        Lbar = 0.5 * (Ab + Ab.T)
    return {"X": W @ Ybar.T, "mw": Wbar.sum(axis=1), "Lw": Lbar, "U": Ubar, "W": W}


def sample_weights(mw, Lw, Z):
    """sampling_functions.jl:29,35,44  w = mw .+ Uw \\ randn(...)."""
    D = mw.shape[0]
    Uw = chol_upper(dense_precision(Lw, D, Z.dtype))
    Z2 = Z.reshape(D, -1)
    return (mw[:, None] + _solve_tri(Uw, Z2, trans=False)).reshape(Z.shape)


# --------------------------------------------------------------------------------------------
# the hot loop, literally: bayesian_linear_regression.jl:72-89
# --------------------------------------------------------------------------------------------
class _NoiseChol:
    """_cholesky(fx.Sy) (:79).  Julia dispatches on the matrix type: a Diagonal (or Diagonal{Fill}) noise
    covariance factorises in O(N) and `Sy.U' \\ M` is a row scaling; a dense Sy goes through potrf/trtrs."""

    def __init__(self, Sy, N, dt):
        Sy = np.asarray(Sy, dtype=dt)
        if Sy.ndim == 2:
            self.U, self.d = chol_upper(Sy), None
        else:
            d = np.full(N, Sy, dtype=dt) if Sy.ndim == 0 else Sy
            if np.any(d <= 0):
                raise np.linalg.LinAlgError("noise covariance is not positive definite")
            self.U, self.d = None, np.sqrt(d)

    def solve_Ut(self, M):  # Sy.U' \ M
        if self.U is not None:
            return _solve_tri(self.U, M, trans=True)
        return M / (self.d[:, None] if M.ndim == 2 else self.d)

    def logdet(self):
        diag = np.diag(self.U) if self.U is not None else self.d
        return 2.0 * float(np.sum(np.log(diag.astype(np.float64))))


def compute_inference_quantities(mw, Lw, X, Sy, y):
    D, N = X.shape
    if y.shape[0] != N:  # :74
        raise ValueError("length(y) != size(fx.x.X, 2)")
    dt = X.dtype
    Uw = chol_upper(dense_precision(Lw, D, dt))  # :78
    Sc = _NoiseChol(Sy, N, dt)  # :79
    Bt = Sc.solve_Ut(np.ascontiguousarray(_solve_tri(Uw, X, trans=True).T))  # :81  N x D (the ' materialises)
    dy = Sc.solve_Ut(y - mean(mw, X))  # :82
    # :84 -- log(2pi) is a Float64 in Julia, so the scalar is promoted to double for f32 inputs
    logpdf_dy = -(N * LOG2PI + Sc.logdet() + float(np.sum(dy * dy))) / 2
    Lam = chol_upper(Bt.T @ Bt + np.eye(D, dtype=dt))  # :86  (this is .U of the Cholesky object)
    return Uw, Bt, dy, logpdf_dy, Lam


def logpdf_literal(mw, Lw, X, Sy, y):
    """:55-58."""
    _, Bt, dy, logpdf_dy, LamU = compute_inference_quantities(mw, Lw, X, Sy, y)
    logdet_Lam = 2.0 * float(np.sum(np.log(np.diag(LamU))))
    v = _solve_tri(LamU, Bt.T @ dy, trans=True)
    return -(logdet_Lam - float(np.sum(v * v))) / 2 + logpdf_dy


def posterior_literal(mw, Lw, X, Sy, y):
    """:60-69 and :92-93.  Returns (mw', T, Lw') with T upper, Lw' = T'T."""
    Uw, Bt, dy, _, LamU = compute_inference_quantities(mw, Lw, X, Sy, y)
    v = Bt.T @ dy
    m_eps = _solve_tri(LamU, _solve_tri(LamU, v, trans=True), trans=False)  # :64  Lam \ v
    T = LamU @ Uw  # :67
    mw_post = mw + _solve_tri(Uw, m_eps, trans=False)  # :68
    return mw_post, T, T.T @ T  # :92


# --------------------------------------------------------------------------------------------
# the direct ("Gram") form the GPU computes -- SURVEY.md 0.1
# --------------------------------------------------------------------------------------------
def _noise_diag(Sy, N, dt):
    Sy = np.asarray(Sy, dtype=dt)
    if Sy.ndim == 0:
        return np.full(N, Sy, dtype=dt)
    if Sy.ndim == 1:
        return Sy
    raise ValueError("direct form needs diagonal or isotropic noise")


def posterior_logpdf_direct(mw, Lw, X, Sy, y, prior_factor=None):
    """One fused update: returns (mw', T, Lw', logpdf).  ``prior_factor`` (upper U with
    Lw = U'U) may be given instead of Lw, mirroring a PDMat prior (:93)."""
    D, N = X.shape
    if y.shape[0] != N:
        raise ValueError("length(y) != size(fx.x.X, 2)")
    dt = X.dtype
    s = _noise_diag(Sy, N, dt)
    if prior_factor is not None:
        Uw = np.triu(np.asarray(prior_factor, dtype=dt))
        Lw_d = Uw.T @ Uw
    else:
        Lw_d = dense_precision(Lw, D, dt)
        Uw = chol_upper(Lw_d)
    delta = y - X.T @ mw
    r = delta / s
    A = Lw_d + (X / s) @ X.T
    b = X @ r
    T = chol_upper(A)
    u = _solve_tri(T, b, trans=True)
    m = _solve_tri(T, u, trans=False)
    logdet_A = 2.0 * float(np.sum(np.log(np.diag(T).astype(np.float64))))
    logdet_Lw = 2.0 * float(np.sum(np.log(np.diag(Uw).astype(np.float64))))
    logdet_Sy = float(np.sum(np.log(s.astype(np.float64))))
    quad = float(np.sum(delta.astype(np.float64) * r.astype(np.float64)))
    uu = float(np.sum(u.astype(np.float64) ** 2))
    lp = -0.5 * (N * LOG2PI + logdet_Sy + quad + logdet_A - logdet_Lw - uu)
    return mw + m, T, A, lp


def logpdf_grad(mw, Lw, X, Sy, y):
    """Closed-form gradient of the log marginal likelihood logpdf(fx, y) (reference :55-58) with respect to every input of
    the path: what reverse-mode AD of the reference (the Zygote use of README.md:56-71) yields.
    With S = diag(1/s), A = Lw + X S X', mw' the posterior mean, r = y - X'mw' (posterior residual):
        dL/dy  = -S r                dL/dmw = X S r
        dL/dX  = (mw' r' - A^-1 X) S                      (D x N)
        dL/ds_n = -(s_n - r_n^2 - x_n'A^-1 x_n) / (2 s_n^2)
        dL/dLw = -(m m' + A^-1 - Lw^-1) / 2,  m = mw' - mw   (symmetric D x D)
    Returns (logpdf, dict)."""
    D, N = X.shape
    s = _noise_diag(Sy, N, X.dtype)
    Lw_d = dense_precision(Lw, D, X.dtype)
    mw_p, T, A, lp = posterior_logpdf_direct(mw, Lw_d, X, s, y)
    r = y - X.T @ mw_p
    w = 1.0 / s
    Z = _solve_tri(T, X, trans=True)        # L^-1 X
    G = _solve_tri(T, Z, trans=False)       # A^-1 X
    v = np.sum(Z * Z, axis=0)
    Ainv = _solve_tri(T, _solve_tri(T, np.eye(D, dtype=X.dtype), trans=True), trans=False)
    m = mw_p - mw
    grads = dict(
        y=-w * r,
        mw=X @ (w * r),
        X=(np.outer(mw_p, r) - G) * w[None, :],
        s=-(s - r * r - v) / (2.0 * s * s),
        Lw=-0.5 * (np.outer(m, m) + Ainv - np.linalg.inv(Lw_d)),
        Ainv=Ainv,
        mw_post=mw_p,
    )
    return lp, grads


def marginals_direct(mw, U, X, Sy):
    """mean_n = x_n' mw, var_n = |U^-T x_n|^2 + Sy_nn, given the upper factor U of the precision."""
    D, N = X.shape
    s = _noise_diag(Sy, N, X.dtype)
    alpha = _solve_tri(np.triu(U), X, trans=True)
    return X.T @ mw, np.sum(alpha * alpha, axis=0) + s


def logpdf_naive(mw, Lw, X, Sy, y):
    """The independent formula of /root/reference/test/bayesian_linear_regression.jl:28-37:
    y ~ N(X'mw, X' inv(Lw) X + Sy), evaluated with a dense N x N Cholesky."""
    D, N = X.shape
    dt = X.dtype
    m = X.T @ mw
    Lw_d = dense_precision(Lw, D, dt)
    S = X.T @ np.linalg.solve(Lw_d, X) + dense_noise(Sy, N, dt)
    S = (S + S.T) / 2
    C = sla.cholesky(S, lower=False)
    d = y - m
    z = _solve_tri(C, d, trans=True)
    return -(N * LOG2PI + 2.0 * np.sum(np.log(np.diag(C))) + z @ z) / 2


def logpdf_naive_mp(mw, Lw, X, Sy, y, dps=50):
    """Same formula in mpmath at ``dps`` decimal digits (tiny problems only)."""
    import mpmath as mp

    mp.mp.dps = dps
    D, N = X.shape
    Xm = mp.matrix(X.tolist())
    Lm = mp.matrix(dense_precision(Lw, D, np.float64).tolist())
    Sm = mp.matrix(dense_noise(Sy, N, np.float64).tolist())
    m = Xm.T * mp.matrix(mw.tolist())
    Sig = Xm.T * mp.inverse(Lm) * Xm + Sm
    d = mp.matrix(y.tolist()) - m
    q = (d.T * mp.lu_solve(Sig, d))[0]
    ld = mp.log(mp.det(Sig))
    return float(-(N * mp.log(2 * mp.pi) + ld + q) / 2)


# --------------------------------------------------------------------------------------------
# toy problems: the construction of /root/reference/test/test_utils.jl:4-10 with our own seeds
# --------------------------------------------------------------------------------------------
def generate_toy_problem(rng, N, D, dense_noise_cov=True, dtype=np.float64):
    X = rng.standard_normal((D, N))
    B = rng.standard_normal((D, D))
    C = 0.1 * rng.standard_normal((N, N))
    mw = rng.standard_normal(D)
    Lw = B @ B.T + np.eye(D)
    Sy = C @ C.T + np.eye(N)
    if not dense_noise_cov:
        Sy = np.exp(rng.standard_normal(N))  # README.md:50 heteroscedastic Diagonal
    return (X.astype(dtype), mw.astype(dtype), Lw.astype(dtype), np.asarray(Sy, dtype=dtype))


def phi_test(x):
    """/root/reference/test/test_utils.jl:28-30: phi(x) = [1, prod(x)] per input."""
    if isinstance(x, RowVecs):
        return RowVecs(np.column_stack([np.ones(len(x)), np.prod(x.X, axis=1)]))
    if isinstance(x, ColVecs):
        return ColVecs(np.vstack([np.ones(len(x)), np.prod(x.X, axis=0)]))
    return phi_test(ColVecs(x)).X
