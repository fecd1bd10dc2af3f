"""Host-side mirror of BayesianLinearRegressors.jl's AbstractGPs surface, backed by the MI355X library.

Julia is not available in the build image, so this is the reference's operator interface restated
in Python over the same C ABI a Julia ``ccall`` shim binds: same names, same
argument meaning, same error behaviour.  Every number is computed by the HIP kernels; there is no
CPU fallback.

    f   = BayesianLinearRegressor(mw, Lw)            # reference src/bayesian_linear_regression.jl:11-14
    fx  = f(ColVecs(X), Sigma)                       # FiniteGP (AbstractGPs)
    logpdf(fx, y); posterior(fx, y)                  # :55-58, :60-69
    mean(fx); var(fx); mean_and_var(fx); marginals(fx)   # :33, :40-43, :47
    rand(rng, fx, S)                                 # :49-53
    rand(rng, f) / rand(rng, f, dims)                # src/sampling_functions.jl:27-38
    BasisFunctionRegressor(f, phi)                   # src/basis_function_regression.jl:34-65
"""
from __future__ import annotations

import math

import numpy as np

from . import _abi

__all__ = [
    "ColVecs", "RowVecs", "Diagonal", "Symmetric", "PDMat", "Normal", "RandomFourierFeatures",
    "BayesianLinearRegressor", "BasisFunctionRegressor", "BLRFunctionSample", "FiniteGP",
    "mean", "var", "cov", "std", "mean_and_var", "mean_and_cov", "marginals", "rand", "rand_b", "logpdf", "posterior",
]


# ---------------------------------------------------------------------------------------------------
# containers (KernelFunctions.ColVecs / RowVecs, LinearAlgebra.Diagonal / Symmetric, PDMats.PDMat)
# ---------------------------------------------------------------------------------------------------
class ColVecs:
    """D x N matrix whose columns are the inputs."""

    def __init__(self, X):
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("ColVecs expects a matrix")
        self.X = X

    def __len__(self):
        return self.X.shape[1]

    def __getitem__(self, idx):
        return ColVecs(self.X[:, idx])


class RowVecs:
    """N x D matrix whose rows are the inputs."""

    def __init__(self, X):
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("RowVecs expects a matrix")
        self.X = X

    def __len__(self):
        return self.X.shape[0]

    def __getitem__(self, idx):
        return RowVecs(self.X[idx, :])


class Diagonal:
    def __init__(self, diag):
        self.diag = np.asarray(diag)
        if self.diag.ndim != 1:
            raise ValueError("Diagonal expects a vector")

    @property
    def shape(self):
        n = self.diag.shape[0]
        return (n, n)

    def toarray(self):
        return np.diag(self.diag)


class Symmetric:
    def __init__(self, data):
        self.data = np.asarray(data)

    @property
    def shape(self):
        return self.data.shape

    def toarray(self):
        u = np.triu(self.data)
        return u + np.triu(self.data, 1).T


class PDMat:
    """Positive-definite matrix carried by its upper Cholesky factor (PDMats.PDMat(Cholesky(U)))."""

    def __init__(self, U):
        self.U = np.asarray(U)

    @property
    def shape(self):
        return self.U.shape

    def toarray(self):
        u = np.triu(self.U)
        return u.T @ u


class Normal:
    """Distributions.Normal(mu, sigma) -- what `marginals` returns per input."""

    __slots__ = ("mu", "sigma")

    def __init__(self, mu, sigma):
        self.mu, self.sigma = float(mu), float(sigma)

    def __repr__(self):
        return f"Normal(mu={self.mu}, sigma={self.sigma})"


def std(n):
    return n.sigma if isinstance(n, Normal) else np.array([k.sigma for k in n])


# ---------------------------------------------------------------------------------------------------
# x_as_colvecs: reference :20-31.  Index/shape work only -- resolved to (pointer, layout flag, ld)
# ---------------------------------------------------------------------------------------------------
def _x_layout(x, dtype):
    """-> (array kept alive, layout flag, ldx, D, N) with zero copies for C- or F-contiguous input."""
    if isinstance(x, ColVecs):
        M, colvecs = x.X, True
    elif isinstance(x, RowVecs):
        M, colvecs = x.X, False
    elif isinstance(x, np.ndarray) and x.ndim == 2:
        M, colvecs = x, True  # AbstractGPs turns a raw D x N matrix into ColVecs
    else:
        raise TypeError(
            f"{type(x).__name__} is not a subtype of AbstractVector that is known. "
            "Please provide either a ColVecs or RowVecs."
        )
    M = np.asarray(M, dtype=dtype)
    if not (M.flags.c_contiguous or M.flags.f_contiguous):
        M = np.ascontiguousarray(M)
    if colvecs:
        D, N = M.shape
        # (D, N) Fortran order  == D x N column-major == COLVECS;  C order == N x D column-major == ROWVECS
        if M.flags.f_contiguous:
            return M, _abi.LAYOUT_COLVECS, max(D, 1), D, N
        return M, _abi.LAYOUT_ROWVECS, max(N, 1), D, N
    N, D = M.shape
    if M.flags.f_contiguous:
        return M, _abi.LAYOUT_ROWVECS, max(N, 1), D, N
    return M, _abi.LAYOUT_COLVECS, max(D, 1), D, N


def _first_nonpositive(v):
    """1-based index of the first entry that is not > 0 (NaN included), 0 if all are: LAPACK `info` of a diagonal Cholesky."""
    bad = np.flatnonzero(~(np.asarray(v) > 0))
    return int(bad[0]) + 1 if bad.size else 0


def _noise(Sy, N, dtype, need_cholesky=False):
    """-> (array, noise_kind).  Scalar = f(x, sigma^2); vector / Diagonal = Diagonal(v).
    ``need_cholesky``: the caller's reference method runs _cholesky(Sigma_y) (rand, :52) and the kernel behind it does not
    report a non-positive variance itself -- raise PosDefException here, as the reference would (logpdf / posterior get it
    from the kernel's info)."""
    if isinstance(Sy, Diagonal):
        Sy = Sy.diag
    Sy = np.asarray(Sy, dtype=dtype)
    if Sy.ndim <= 1 and need_cholesky:
        k = _first_nonpositive(Sy.reshape(-1))
        if k:
            raise _abi.PosDefException(k)
    if Sy.ndim == 0:
        return Sy.reshape(1).copy(), _abi.NOISE_ISOTROPIC
    if Sy.ndim == 1:
        if Sy.shape[0] != N:
            raise ValueError("length of the noise diagonal != number of inputs")
        return np.ascontiguousarray(Sy), _abi.NOISE_DIAGONAL
    if Sy.ndim == 2:  # dense N x N covariance (the reference's own toy problems, test/test_utils.jl:7-8)
        if Sy.shape != (N, N):
            raise ValueError("size of the noise covariance != number of inputs")
        return np.asfortranarray(Sy), _abi.NOISE_DENSE
    raise ValueError("noise covariance must be a scalar, a vector / Diagonal, or an N x N matrix")


def _mean_vector(mw, D, dtype):
    """The prior mean as a contiguous vector of the working dtype; its length must be the input dimension (the library copies
    D elements from this buffer: a shorter one would be read past its end instead of raising the reference's DimensionMismatch)."""
    mw = np.ascontiguousarray(mw, dtype=dtype)
    if mw.ndim != 1 or mw.shape[0] != D:
        raise ValueError(f"length(mw) = {mw.shape[0] if mw.ndim == 1 else mw.shape} != dimension of the inputs ({D})")
    return mw


def _prior(Lw, D, dtype, need_cholesky=False):
    """-> (array, prior_kind, ldl).  ``need_cholesky``: see _noise (var / rand / weight draws with a Diagonal precision are
    elementwise kernels that do not report a non-positive entry; the reference's _cholesky(Lw) at :41 / :51 throws)."""
    if isinstance(Lw, Diagonal):
        d = np.ascontiguousarray(Lw.diag, dtype=dtype)
        if d.shape[0] != D:
            raise ValueError("size of the prior precision != length(mw)")
        if need_cholesky:
            k = _first_nonpositive(d)
            if k:
                raise _abi.PosDefException(k)
        return d, _abi.PRIOR_DIAGONAL, 1
    if isinstance(Lw, PDMat):
        U = np.asfortranarray(Lw.U, dtype=dtype)
        kind = _abi.PRIOR_UPPER_FACTOR
    else:
        A = Lw.data if isinstance(Lw, Symmetric) else Lw
        U = np.asfortranarray(A, dtype=dtype)  # upper triangle is read (LAPACK 'U')
        kind = _abi.PRIOR_DENSE
    if U.shape != (D, D):
        raise ValueError("size of the prior precision != length(mw)")
    return U, kind, max(D, 1)


def _dtype_of(*arrays):
    dts = [np.asarray(a).dtype for a in arrays if a is not None]
    if dts and all(dt == np.float32 for dt in dts):
        return np.float32
    return np.float64


# ---------------------------------------------------------------------------------------------------
# the regressors
# ---------------------------------------------------------------------------------------------------
class BayesianLinearRegressor:
    """w ~ Normal(mw, inv(Lw));  f(x) = dot(x, w).   reference :11-14"""

    def __init__(self, mw, Lw):
        self.mw = np.asarray(mw)
        if self.mw.ndim != 1:
            raise ValueError("mw must be a vector")
        self.Lw = Lw

    # field name used by the reference
    @property
    def Λw(self):  # noqa: PLC2401
        return self.Lw

    def __call__(self, x, Sy=1e-18):
        return FiniteGP(self, x, Sy)


class BasisFunctionRegressor:
    """bfr(X) = blr(phi(X)).   reference src/basis_function_regression.jl:34-37"""

    def __init__(self, blr, phi):
        if not isinstance(blr, BayesianLinearRegressor):
            raise TypeError("BasisFunctionRegressor wraps a BayesianLinearRegressor")
        self.blr = blr
        self.phi = phi

    @property
    def ϕ(self):  # noqa: PLC2401
        return self.phi

    def __call__(self, x, Sy=1e-18):
        return FiniteGP(self, x, Sy)


class RandomFourierFeatures:
    """phi(x) = scale * cos(Omega' x + phase): the random-Fourier basis of BASELINE config 5, usable as the `phi`
    of a BasisFunctionRegressor (the reference accepts any callable, src/basis_function_regression.jl:7-9).
    Omega is D_in x D.  Calling it materialises the features through the device kernel; `logpdf` / `posterior`
    on a BasisFunctionRegressor with this phi take the fused path (features never leave the GPU)."""

    def __init__(self, Omega, phase, scale=None):
        self.Omega = np.asarray(Omega)
        self.phase = np.asarray(phase)
        if self.Omega.ndim != 2 or self.phase.shape != (self.Omega.shape[1],):
            raise ValueError("Omega must be D_in x D and phase of length D")
        self.scale = float(np.sqrt(2.0 / self.Omega.shape[1])) if scale is None else float(scale)

    def _operands(self, x, dtype):
        X, layout, ldx, Din, N = _x_layout(x, dtype)
        if layout != _abi.LAYOUT_COLVECS:  # the feature kernel reads ColVecs inputs (D_in is tiny: one small copy)
            # ROWVECS layout = N x D_in column-major.  Which axis of the numpy array is N follows from the CONTAINER and the
            # memory order that _x_layout resolved -- never from comparing shapes (a square N == D_in input is ambiguous).
            M = X if isinstance(x, RowVecs) else X.T  # -> (N, D_in) view of the same memory
            X = np.asfortranarray(M.T, dtype=dtype)     # (D_in, N) column-major: ColVecs
            ldx = max(Din, 1)
        if Din != self.Omega.shape[0]:
            raise ValueError("input dimension != rows of Omega")
        Om = np.asfortranarray(self.Omega, dtype=dtype)
        ph = np.ascontiguousarray(self.phase, dtype=dtype)
        return X, ldx, Din, N, Om, ph

    def __call__(self, x):
        dtype = _dtype_of(self.Omega, x.X if hasattr(x, "X") else x)
        X, ldx, Din, N, Om, ph = self._operands(x, dtype)
        D = Om.shape[1]
        Phi = np.empty((D, N), dtype=dtype, order="F")
        _handle().rff_features(dtype, _abi.MEM_HOST, Din, D, N, X, ldx, Om, max(Din, 1), ph, self.scale, Phi, max(D, 1))
        if isinstance(x, RowVecs):
            return RowVecs(Phi.T)
        if isinstance(x, ColVecs):
            return ColVecs(Phi)
        return Phi


class FiniteGP:
    """AbstractGPs.FiniteGP: a regressor evaluated at inputs x with observation-noise covariance Sy."""

    def __init__(self, f, x, Sy):
        self.f, self.x, self.Sy = f, x, Sy

    @property
    def Σy(self):  # noqa: PLC2401
        return self.Sy


def _to_finite_blr(fx):
    """reference src/basis_function_regression.jl:41"""
    if isinstance(fx.f, BasisFunctionRegressor):
        return FiniteGP(fx.f.blr, fx.f.phi(fx.x), fx.Sy)
    if isinstance(fx.f, BayesianLinearRegressor):
        return fx
    raise TypeError("expected a FiniteGP over a BayesianLinearRegressor or BasisFunctionRegressor")


def _handle():
    return _abi.default_handle()


def _wrap_like(prior_Lw, T, A):
    """__build_Lambda, reference :92-93: PDMat prior -> PDMat posterior carrying T; else Symmetric(T'T)."""
    if isinstance(prior_Lw, PDMat):
        return PDMat(T)
    return Symmetric(A)


# ---------------------------------------------------------------------------------------------------
# AbstractGPs API
# ---------------------------------------------------------------------------------------------------
def _fused_rff(fx, y, want_posterior):
    """BasisFunctionRegressor with a RandomFourierFeatures phi: features + inference in one library call."""
    bfr, rff, blr = fx.f, fx.f.phi, fx.f.blr
    dtype = _dtype_of(blr.mw, y, rff.Omega)
    X, ldx, Din, N, Om, ph = rff._operands(fx.x, dtype)
    D = Om.shape[1]
    y = np.ascontiguousarray(y, dtype=dtype)
    if y.ndim != 1 or y.shape[0] != N:
        raise ValueError("length(y) != size(fx.x.X, 2)")  # reference :74
    mw = _mean_vector(blr.mw, D, dtype)
    s, noise_kind = _noise(fx.Sy, N, dtype)
    Lw, prior_kind, ldl = _prior(blr.Lw, D, dtype)
    lp = np.zeros(1, dtype=np.float64)
    info = np.zeros(1, dtype=np.int32)
    mw_post = T = A = None
    if want_posterior:
        mw_post = np.empty(D, dtype=dtype)
        T = np.empty((D, D), dtype=dtype, order="F")
        A = np.empty((D, D), dtype=dtype, order="F") if not isinstance(blr.Lw, PDMat) else None
    _handle().posterior_rff(dtype, _abi.MEM_HOST, Din, D, N, X, ldx, Om, max(Din, 1), ph, rff.scale, y, noise_kind, s,
                            prior_kind, mw, Lw, ldl, mw_post, T, max(D, 1), A, max(D, 1), lp, info)
    if info[0] > 0:
        raise _abi.PosDefException(int(info[0]))
    return float(lp[0]), mw_post, T, A


def _fused(fx, y, want_posterior):
    if isinstance(fx.f, BasisFunctionRegressor) and isinstance(fx.f.phi, RandomFourierFeatures):
        return _fused_rff(fx, y, want_posterior)
    fx = _to_finite_blr(fx)
    blr = fx.f
    dtype = _dtype_of(blr.mw, y)
    X, layout, ldx, D, N = _x_layout(fx.x, dtype)
    y = np.ascontiguousarray(y, dtype=dtype)
    if y.ndim != 1:
        raise ValueError("y must be a vector")
    if y.shape[0] != N:
        raise ValueError("length(y) != size(fx.x.X, 2)")  # reference :74
    mw = _mean_vector(blr.mw, D, dtype)
    s, noise_kind = _noise(fx.Sy, N, dtype)
    Lw, prior_kind, ldl = _prior(blr.Lw, D, dtype)
    lp = np.zeros(1, dtype=np.float64)
    if want_posterior:
        mw_post = np.empty(D, dtype=dtype)
        T = np.empty((D, D), dtype=dtype, order="F")
        A = np.empty((D, D), dtype=dtype, order="F") if not isinstance(blr.Lw, PDMat) else None
    else:
        mw_post = T = A = None
    if noise_kind == _abi.NOISE_DENSE:  # reference :79-82 general branch: whitened on the device (blr_posterior_dense_noise_*)
        info = np.zeros(1, dtype=np.int32)
        _handle().posterior_dense_noise(dtype, _abi.MEM_HOST, layout, D, N, X, ldx, y, s, max(N, 1), prior_kind, mw, Lw, ldl, mw_post, T,
                                        max(D, 1), A, max(D, 1), lp, info)
        if info[0] > 0:
            raise _abi.PosDefException(int(info[0]))
        return float(lp[0]), mw_post, T, A
    _handle().posterior(dtype, layout, D, N, X, ldx, y, noise_kind, s, prior_kind, mw, Lw, ldl, mw_post, T, max(D, 1),
                        A, max(D, 1), lp)
    return float(lp[0]), mw_post, T, A


def logpdf(fx, y):
    """reference :55-58.  A matrix Y (N x S) gives the column-wise log densities (AbstractGPs fallback)."""
    y = np.asarray(y)
    if y.ndim == 2:
        return logpdf_columns(fx, y)
    return _fused(fx, y, want_posterior=False)[0]


def logpdf_columns(fx, Y, return_means=False):
    """Shared-X multi-output evidence (AbstractGPs' logpdf(fx, Y::AbstractMatrix)): the Gram matrix and its factor are
    formed once, per column only one GEMM column and two triangular solves remain (blr_logpdf_multi_*).
    ``return_means=True`` also returns the D x S matrix of per-column posterior means."""
    fx = _to_finite_blr(fx)
    blr = fx.f
    dtype = _dtype_of(blr.mw, Y)
    X, layout, ldx, D, N = _x_layout(fx.x, dtype)
    Yf = np.asfortranarray(Y, dtype=dtype)
    if Yf.shape[0] != N:
        raise ValueError("length(y) != size(fx.x.X, 2)")
    S = Yf.shape[1]
    mw = _mean_vector(blr.mw, D, dtype)
    s, noise_kind = _noise(fx.Sy, N, dtype)
    if noise_kind == _abi.NOISE_DENSE:  # AbstractGPs' column-wise fallback (each column re-whitens; test sizes only)
        lps, means = [], []
        for j in range(S):
            lp_j, m_j, _, _ = _fused(fx, Yf[:, j], want_posterior=return_means)
            lps.append(lp_j)
            means.append(m_j)
        return (np.array(lps), np.stack(means, axis=1)) if return_means else np.array(lps)
    Lw, prior_kind, ldl = _prior(blr.Lw, D, dtype)
    lp = np.zeros(S, dtype=np.float64)
    if D <= 128 and S <= 256 and not return_means:
        # few columns of a small problem: S independent fused updates in one launch (X shared through strideX = 0) beat the
        # fixed cost of the shared-X pipeline (measured: 0.57 ms vs 1.13 ms at D=128, N=4096, S=64)
        infos = np.zeros(S, dtype=np.int32)
        _handle().posterior_batched(dtype, _abi.MEM_HOST, layout, S, D, N, X, ldx, 0, Yf, N, noise_kind, s, 0, prior_kind,
                                    mw, 0, Lw, ldl, 0, None, D, None, max(D, 1), D * D, None, max(D, 1), D * D, lp, infos)
        bad = np.flatnonzero(infos)
        if bad.size:
            raise _abi.PosDefException(int(infos[bad[0]]))
        return lp
    info = np.zeros(1, dtype=np.int32)
    M = np.empty((D, S), dtype=dtype, order="F") if return_means else None
    _handle().logpdf_multi(dtype, _abi.MEM_HOST, layout, D, N, S, X, ldx, Yf, max(N, 1), noise_kind, s, prior_kind, mw, Lw, ldl,
                           lp, M, max(D, 1), info)
    if info[0] != 0:
        raise _abi.PosDefException(int(info[0]))
    return (lp, M) if return_means else lp


def _upper_inverse_on_device(h, dtype, prior_kind, Lw, D):
    """Uw^-1 (D x D, float64) for Uw = chol(Lw).U or the given upper factor: blr_sample_weights_* on the identity (w = 0 + Uw^-1 z
    per column, reference sampling_functions.jl:29) -- the library factorises a dense prior itself; no host LAPACK."""
    eye = np.asfortranarray(np.eye(D, dtype=dtype))
    W = np.zeros((D, D), dtype=dtype, order="F")
    Lw_arr = np.asfortranarray(np.asarray(Lw, dtype=dtype))
    h.sample_weights(dtype, _abi.MEM_HOST, D, D, prior_kind, np.zeros(D, dtype=dtype), Lw_arr, max(D, 1), eye, D, W, D)
    return W.astype(np.float64)


def _prior_inverse_on_device(h, dtype, prior_kind, Lw, D):
    """Lw^-1 = Uw^-1 Uw^-T for a dense precision or an upper factor, by two triangular-solve calls on the device (the second one
    solves against the rows of the first result); used by the prior tangent of the evidence gradient."""
    Ui = _upper_inverse_on_device(h, dtype, prior_kind, Lw, D)
    Z = np.asfortranarray(Ui.T.astype(dtype))
    W = np.zeros((D, D), dtype=dtype, order="F")
    Lw_arr = np.asfortranarray(np.asarray(Lw, dtype=dtype))
    h.sample_weights(dtype, _abi.MEM_HOST, D, D, prior_kind, np.zeros(D, dtype=dtype), Lw_arr, max(D, 1), Z, D, W, D)
    Wd = W.astype(np.float64)
    return 0.5 * (Wd + Wd.T)


def logpdf_and_gradient(fx, y):
    """The value of logpdf(fx, y) (reference :55-58) and its gradient with respect to every input of the path -- the
    reverse-mode rule the reference gets from Zygote through its Julia code (README.md:56-71) and a ccall-backed logpdf
    must provide itself.

    Returns ``(lp, grads)`` with ``grads`` a dict: ``X`` (same container layout as the inputs: D x N for ColVecs / a
    matrix, N x D for RowVecs), ``y`` (N), ``noise`` (N for Diagonal noise, a scalar for isotropic noise), ``mw`` (D)
    and ``Lw`` -- the gradient with respect to the precision: symmetric D x D for a dense / Symmetric / PDMat prior
    (for PDMat(U) chain with dU = U (G + G')), the diagonal (D) for a Diagonal prior."""
    fx = _to_finite_blr(fx)
    blr = fx.f
    dtype = _dtype_of(blr.mw, y)
    X, layout, ldx, D, N = _x_layout(fx.x, dtype)
    y = np.ascontiguousarray(y, dtype=dtype)
    if y.ndim != 1 or y.shape[0] != N:
        raise ValueError("length(y) != size(fx.x.X, 2)")  # reference :74
    mw = _mean_vector(blr.mw, D, dtype)
    s, noise_kind = _noise(fx.Sy, N, dtype)
    if noise_kind == _abi.NOISE_DENSE:
        raise NotImplementedError("logpdf_and_gradient: closed-form rule implemented for isotropic / Diagonal noise only")
    Lw, prior_kind, ldl = _prior(blr.Lw, D, dtype)
    lp = np.zeros(1, dtype=np.float64)
    info = np.zeros(1, dtype=np.int32)
    dX = np.empty_like(X)  # same memory order and leading dimension as the staged inputs
    dy = np.empty(N, dtype=dtype)
    ds = np.empty(N, dtype=dtype)
    dmw = np.empty(D, dtype=dtype)
    mw_post = np.empty(D, dtype=dtype)
    Ainv = np.empty((D, D), dtype=dtype, order="F")
    _handle().logpdf_grad_batched(dtype, _abi.MEM_HOST, layout, 1, D, N, X, ldx, 0, y, 0, noise_kind, s, 0, prior_kind, mw, 0,
                                  Lw, ldl, 0, lp, dX, ldx, 0, dy, 0, ds, 0, dmw, 0, mw_post, 0, Ainv, D, 0, info)
    if info[0] != 0:
        raise _abi.PosDefException(int(info[0]))
    # dL/dLw = -(m m' + A^-1 - Lw^-1) / 2: D x D host work on the device's A^-1 and posterior mean
    m = (mw_post - mw).astype(np.float64)
    Ai = np.asarray(Ainv, dtype=np.float64)
    Ai = np.tril(Ai) + np.tril(Ai, -1).T  # symmetric by construction; use one triangle
    if prior_kind == _abi.PRIOR_DIAGONAL:
        gL = -0.5 * (m * m + np.diag(Ai) - 1.0 / np.asarray(Lw, dtype=np.float64))
    else:
        # (the prior's inverse in float64 whatever the element type of the call: D x D work, and the three terms cancel)
        gL = -0.5 * (np.outer(m, m) + Ai - _prior_inverse_on_device(_handle(), np.float64, prior_kind, Lw, D))
    # hand dX back in the caller's container orientation
    x = fx.x
    if isinstance(x, RowVecs):
        gX = dX if dX.shape == (N, D) else dX.T
    else:
        gX = dX if dX.shape == (D, N) else dX.T
    grads = {"X": np.asarray(gX), "y": dy, "noise": ds if noise_kind == _abi.NOISE_DIAGONAL else dtype(ds.sum(dtype=np.float64)),
             "mw": dmw, "Lw": gL.astype(dtype)}
    return float(lp[0]), grads


def posterior(fx, y):
    """reference :60-69 (and basis_function_regression.jl:62-65): same wrapper type as the prior."""
    _, mw_post, T, A = _fused(fx, y, want_posterior=True)
    base = fx.f.blr if isinstance(fx.f, BasisFunctionRegressor) else fx.f
    post = BayesianLinearRegressor(mw_post, _wrap_like(base.Lw, T, A))
    if isinstance(fx.f, BasisFunctionRegressor):
        return BasisFunctionRegressor(post, fx.f.phi)
    return post


# ---------------------------------------------------------------------------------------------------
# many independent regressors in ONE library call: what `map(posterior, fxs, ys)` / `logpdf.(fxs, ys)` are in the reference
# (config 4 of BASELINE.json is this with 8192 regressors; at D > 128 the regressors share every launch of the update)
# ---------------------------------------------------------------------------------------------------
def _fused_many(fxs, ys, want_posterior):
    """[(logpdf, mw', T, A)] for equally shaped problems through blr_posterior_batched_*; shapes, layouts, noise or prior
    kinds that differ (or a dense noise covariance, or a fused random-Fourier basis) fall back to one call per problem.
    The first problem (in order) whose prior, noise or posterior precision is not positive definite raises
    PosDefException, as the map over the reference's methods would; its position is the exception's ``index``."""
    fxs, ys = list(fxs), list(ys)
    if len(fxs) != len(ys):
        raise ValueError("as many observation vectors as finite regressors are needed")
    if not fxs:
        return []
    one_by_one = lambda: [_fused(fx, y, want_posterior) for fx, y in zip(fxs, ys)]  # noqa: E731
    if any(isinstance(fx.f, BasisFunctionRegressor) and isinstance(fx.f.phi, RandomFourierFeatures) for fx in fxs):
        return one_by_one()
    fbs = [_to_finite_blr(fx) for fx in fxs]
    dtype = np.float32 if all(_dtype_of(fb.f.mw, y) == np.float32 for fb, y in zip(fbs, ys)) else np.float64
    probs = []
    for fb, y in zip(fbs, ys):
        X, layout, ldx, D, N = _x_layout(fb.x, dtype)
        y = np.ascontiguousarray(y, dtype=dtype)
        if y.ndim != 1:
            raise ValueError("y must be a vector")
        if y.shape[0] != N:
            raise ValueError("length(y) != size(fx.x.X, 2)")  # reference :74
        s, noise_kind = _noise(fb.Sy, N, dtype)
        Lw, prior_kind, ldl = _prior(fb.f.Lw, D, dtype)
        probs.append((X, layout, ldx, D, N, y, s, noise_kind, _mean_vector(fb.f.mw, D, dtype), Lw, prior_kind, ldl,
                      isinstance(fb.f.Lw, PDMat)))
    sig = {(q[0].shape, q[0].flags.f_contiguous, q[1], q[3], q[4], q[7], q[10], q[12]) for q in probs}
    if len(sig) != 1 or probs[0][7] == _abi.NOISE_DENSE or probs[0][3] == 0 or probs[0][4] == 0:
        return one_by_one()
    X0, layout, ldx, D, N, _, s0, noise_kind, _, _, prior_kind, ldl, pdmat = probs[0]
    nb = len(probs)
    # one contiguous block per operand: problem b at b * stride (each X already is ldx x cols column-major in memory)
    Xb = np.stack([q[0].reshape(-1, order="A") for q in probs])
    yb = np.stack([q[5] for q in probs])
    sb = np.stack([q[6] for q in probs])
    mwb = np.stack([q[8] for q in probs])
    Lb = np.stack([q[9].reshape(-1, order="A") for q in probs])  # D, or D x D column-major
    lp = np.zeros(nb, dtype=np.float64)
    info = np.zeros(nb, dtype=np.int32)
    if want_posterior:
        mw_post = np.empty((nb, D), dtype=dtype)
        Tb = np.empty((nb, D * D), dtype=dtype)
        Ab = np.empty((nb, D * D), dtype=dtype) if not pdmat else None
    else:
        mw_post = Tb = Ab = None
    _handle().posterior_batched(dtype, _abi.MEM_HOST, layout, nb, D, N, Xb, ldx, Xb.shape[1], yb, yb.shape[1], noise_kind, sb,
                                sb.shape[1], prior_kind, mwb, D, Lb, ldl, Lb.shape[1], mw_post, D, Tb, D, D * D, Ab, D, D * D, lp, info)
    bad = np.flatnonzero(info > 0)
    if bad.size:
        e = _abi.PosDefException(int(info[bad[0]]))
        e.index = int(bad[0])
        raise e
    out = []
    for b in range(nb):
        if want_posterior:
            out.append((float(lp[b]), mw_post[b], Tb[b].reshape((D, D), order="F"), Ab[b].reshape((D, D), order="F") if Ab is not None else None))
        else:
            out.append((float(lp[b]), None, None, None))
    return out


def logpdf_map(fxs, ys):
    """[logpdf(fx, y) for fx, y in zip(fxs, ys)] (reference :55-58 under a map) in one library call."""
    return [r[0] for r in _fused_many(fxs, ys, want_posterior=False)]


def posterior_map(fxs, ys):
    """[posterior(fx, y) for fx, y in zip(fxs, ys)] (reference :60-69 under a map) in one library call."""
    fxs = list(fxs)
    posts = []
    for fx, (_, mw_post, T, A) in zip(fxs, _fused_many(fxs, ys, want_posterior=True)):
        base = fx.f.blr if isinstance(fx.f, BasisFunctionRegressor) else fx.f
        post = BayesianLinearRegressor(mw_post, _wrap_like(base.Lw, T, A))
        posts.append(BasisFunctionRegressor(post, fx.f.phi) if isinstance(fx.f, BasisFunctionRegressor) else post)
    return posts


class _DeviceBuffer:
    """device memory owned through the C ABI (blr_device_alloc): no GPU array library involved"""

    def __init__(self, handle, nbytes):
        self.handle, self.nbytes = handle, int(nbytes)
        self.ptr = handle.device_alloc(self.nbytes)

    @classmethod
    def of(cls, handle, host_array):
        buf = cls(handle, host_array.nbytes)
        handle.memcpy_h2d(buf.ptr, host_array)
        return buf

    def free(self):
        if self.ptr:
            try:
                self.handle.device_free(self.ptr)
            finally:
                self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ResidentPosterior:
    """A posterior kept ON THE DEVICE as its state (mw, T) -- T the upper factor of the precision -- and conditioned IN PLACE
    on further batches: the "repeated conditioning" pattern of reference test/bayesian_linear_regression.jl:49-70
    (``posterior(f'1(X2, S2), y2)``) without re-deriving reference :72-89 from a D x D precision every time
    (blr_update_factor_*: O(k D^2) Givens sweeps for single observations, in-place re-factorisation otherwise).

        st = ResidentPosterior(posterior(f(X1, S1), y1))      # or a prior: ResidentPosterior(f)
        lp2 = st.condition(X2, S2, y2)                         # log p(y2 | y1); state now = posterior given y1 and y2
        f12 = st.regressor()                                   # BayesianLinearRegressor(mw, PDMat(T))

    The state is created by the library (a Diagonal or dense prior precision is factorised on the device -- reference :78, the
    failing leading minor comes back as PosDefException(info) exactly as from `posterior`), lives in device buffers owned
    through blr_device_alloc and is only copied back by `regressor()`.  A BasisFunctionRegressor keeps its basis: `condition`
    maps the inputs through phi (RandomFourierFeatures: on the device, the features never visit the host) and `regressor()`
    returns BasisFunctionRegressor(posterior, phi) like reference basis_function_regression.jl:62-65."""

    def __init__(self, f):
        self.phi = f.phi if isinstance(f, BasisFunctionRegressor) else None
        base = f.blr if isinstance(f, BasisFunctionRegressor) else f
        if not isinstance(base, BayesianLinearRegressor):
            raise TypeError("ResidentPosterior wraps a BayesianLinearRegressor or a BasisFunctionRegressor")
        dtype = _dtype_of(base.mw)
        D = base.mw.shape[0]
        self.dtype, self.D = dtype, D
        h = self._h = _handle()
        item = np.dtype(dtype).itemsize
        mw = np.array(_mean_vector(base.mw, D, dtype))
        self._mw = _DeviceBuffer.of(h, mw)
        self._T = _DeviceBuffer(h, D * D * item)
        Lw = base.Lw
        if isinstance(Lw, PDMat):
            U = np.asfortranarray(np.triu(np.asarray(Lw.U, dtype=dtype)))
            if U.shape != (D, D):
                raise ValueError("size of the prior precision != length(mw)")
            k = _first_nonpositive(np.diag(U))  # a factor with a non-positive diagonal is not a Cholesky factor
            if k:
                raise _abi.PosDefException(k)
            h.memcpy_h2d(self._T.ptr, U)
            return
        # Diagonal / dense precision: T = chol(Lw).U by the library's own factorisation -- the posterior update on ZERO
        # observations (A = Lw), written straight into the resident buffers
        Lw_h, prior_kind, ldl = _prior(Lw, D, dtype)
        d_Lw = _DeviceBuffer.of(h, Lw_h)
        d_one = _DeviceBuffer.of(h, np.ones(1, dtype=dtype))
        d_info = _DeviceBuffer.of(h, np.zeros(1, dtype=np.int32))
        try:
            h.posterior_batched(dtype, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, 1, D, 0, None, max(D, 1), 0, None, 0,
                                _abi.NOISE_ISOTROPIC, d_one.ptr, 0, prior_kind, self._mw.ptr, 0, d_Lw.ptr, ldl, 0,
                                None, D, self._T.ptr, max(D, 1), D * D, None, max(D, 1), D * D, None, d_info.ptr)
            info = np.zeros(1, dtype=np.int32)
            h.memcpy_d2h(info, d_info.ptr)
        finally:
            for b in (d_Lw, d_one, d_info):
                b.free()
        if info[0] != 0:
            raise _abi.PosDefException(int(info[0]))

    def condition(self, x, Sy, y):
        """In-place update with the observations (x, Sy, y); returns log p(y | everything conditioned on so far)."""
        dtype, h, D = self.dtype, self._h, self.D
        temps = []

        def dev(a):
            temps.append(_DeviceBuffer.of(h, a))
            return temps[-1].ptr

        try:
            if isinstance(self.phi, RandomFourierFeatures):
                Xin, ldxin, Din, k, Om, ph = self.phi._operands(x, dtype)
                if Om.shape[1] != D:
                    raise ValueError(f"number of features ({Om.shape[1]}) != length(mw) = {D}")
                temps.append(_DeviceBuffer(h, D * max(k, 1) * np.dtype(dtype).itemsize))
                dX, layout, ldx = temps[-1].ptr, _abi.LAYOUT_COLVECS, max(D, 1)
                h.rff_features(dtype, _abi.MEM_DEVICE, Din, D, k, dev(Xin), ldxin, dev(Om), max(Din, 1), dev(ph), self.phi.scale,
                               dX, ldx)
            else:
                X, layout, ldx, Dx, k = _x_layout(self.phi(x) if self.phi is not None else x, dtype)
                if Dx != D:
                    raise ValueError(f"dimension of the inputs ({Dx}) != length(mw) = {D}")
                dX = dev(X)
            y = np.ascontiguousarray(y, dtype=dtype)
            if y.shape != (k,):
                raise ValueError("length(y) != number of inputs")  # reference :74
            s, noise_kind = _noise(Sy, k, dtype)
            if noise_kind == _abi.NOISE_DENSE:
                raise NotImplementedError("ResidentPosterior.condition takes scalar or diagonal noise (whiten a dense block first)")
            d_lp = dev(np.zeros(1, dtype=np.float64))
            d_info = dev(np.zeros(1, dtype=np.int32))
            h.update_factor(dtype, _abi.MEM_DEVICE, layout, 1, D, k, dX, ldx, 0, dev(y), 0, noise_kind, dev(s), 0, self._mw.ptr, 0,
                            self._T.ptr, max(D, 1), 0, d_lp, d_info)
            lp = np.zeros(1, dtype=np.float64)
            info = np.zeros(1, dtype=np.int32)
            h.memcpy_d2h(lp, d_lp)
            h.memcpy_d2h(info, d_info)
        finally:
            for b in temps:
                b.free()
        if info[0] != 0:
            raise _abi.PosDefException(int(info[0]))
        return float(lp[0])

    def state(self):
        """host copies (mw, T) of the resident state; T column-major upper"""
        mw = np.empty(self.D, dtype=self.dtype)
        T = np.empty((self.D, self.D), dtype=self.dtype, order="F")
        self._h.memcpy_d2h(mw, self._mw.ptr)
        self._h.memcpy_d2h(T, self._T.ptr)
        return mw, np.triu(T)

    def regressor(self):
        """The current state as a regressor of the type it was built from (precision carried by its factor, reference :93)."""
        mw, T = self.state()
        post = BayesianLinearRegressor(mw, PDMat(T))
        return BasisFunctionRegressor(post, self.phi) if self.phi is not None else post


def _marginals(fx, want_mean, want_var):
    fx = _to_finite_blr(fx)
    blr = fx.f
    dtype = _dtype_of(blr.mw)
    X, layout, ldx, D, N = _x_layout(fx.x, dtype)
    mw = _mean_vector(blr.mw, D, dtype)
    s, noise_kind = _noise(fx.Sy, N, dtype)  # var adds diag(Sigma_y) (:43): no factorisation of the noise, no positivity check
    if noise_kind == _abi.NOISE_DENSE:
        s, noise_kind = np.ascontiguousarray(np.diag(s)), _abi.NOISE_DIAGONAL
    Lw, prior_kind, ldl = _prior(blr.Lw, D, dtype, need_cholesky=want_var)  # :41 _cholesky(Lw)
    m = np.empty(N, dtype=dtype) if want_mean else None
    v = np.empty(N, dtype=dtype) if want_var else None
    info = np.zeros(1, dtype=np.int32)
    _handle().marginals_batched(dtype, _abi.MEM_HOST, layout, 1, D, N, X, ldx, 0, noise_kind, s, 0, prior_kind, mw, 0,
                                Lw, ldl, 0, m, N, v, N, info)
    if info[0] > 0:
        raise _abi.PosDefException(int(info[0]))
    return m, v


def mean(fx):
    """reference :33"""
    return _marginals(fx, True, False)[0]


def var(fx):
    """reference :40-43"""
    return _marginals(fx, False, True)[1]


def mean_and_var(fx):
    """reference :47"""
    return _marginals(fx, True, True)


def marginals(fx):
    """AbstractGPs.marginals: Normal.(mean, sqrt.(var))"""
    m, v = mean_and_var(fx)
    return [Normal(mi, math.sqrt(vi)) for mi, vi in zip(m, v)]


def _mean_and_cov(fx, want_mean):
    fx = _to_finite_blr(fx)
    blr = fx.f
    dtype = _dtype_of(blr.mw)
    X, layout, ldx, D, N = _x_layout(fx.x, dtype)
    mw = _mean_vector(blr.mw, D, dtype)
    s, noise_kind = _noise(fx.Sy, N, dtype)
    Lw, prior_kind, ldl = _prior(blr.Lw, D, dtype, need_cholesky=True)  # :36 _cholesky(Lw)
    m = np.empty(N, dtype=dtype) if want_mean else None
    Cv = np.empty((N, N), dtype=dtype, order="F")
    info = np.zeros(1, dtype=np.int32)
    _handle().mean_and_cov(dtype, _abi.MEM_HOST, layout, D, N, X, ldx, noise_kind, s, max(N, 1), prior_kind, mw, Lw, ldl, m, Cv,
                           max(N, 1), info)
    if info[0] > 0:
        raise _abi.PosDefException(int(info[0]))
    return m, Cv


def cov(fx):
    """reference :35-38: Symmetric(alpha' alpha + Sigma_y), alpha = Uw' \\ X -- the full N x N predictive covariance
    (blr_mean_and_cov_*; moderate N, <= 16384)."""
    return _mean_and_cov(fx, False)[1]


def mean_and_cov(fx):
    """reference :45"""
    return _mean_and_cov(fx, True)


def _randn(rng, rows, cols, dtype):
    """randn(rng, rows, cols) filled in column-major order like Julia (memory order == draw order)."""
    return np.asarray(rng.standard_normal((cols, rows)), dtype=dtype).T  # F-contiguous (rows, cols)


class BLRFunctionSample:
    """A function sampled from a regressor by fixing w ~ p(w).  reference src/sampling_functions.jl:12-19"""

    def __init__(self, w, phi):
        self.w = w
        self.phi = phi

    def __call__(self, X):
        x = self.phi(X) if self.phi is not None else X
        f = BayesianLinearRegressor(self.w, Diagonal(np.ones_like(self.w)))
        return _marginals(FiniteGP(f, x, 0.0), True, False)[0]  # phi(X)' w through the mean-only stream


def evaluate(samples, X):
    """Every function sample of `samples` (BLRFunctionSamples of ONE regressor, e.g. ``rand(rng, f, S)``) at the inputs X in one
    pass over phi(X): an N x S matrix whose column j is ``samples[j](X)`` -- reference sampling_functions.jl:16-18 for a batch
    (blr_apply_weights_*)."""
    flat = list(np.asarray(samples, dtype=object).reshape(-1, order="F"))
    if not flat:
        raise ValueError("no samples")
    phi = flat[0].phi
    x = phi(X) if phi is not None else X
    dtype = _dtype_of(*[f.w for f in flat])
    Xa, layout, ldx, D, N = _x_layout(x, dtype)
    W = np.asfortranarray(np.stack([np.asarray(f.w, dtype=dtype) for f in flat], axis=1))
    if W.shape[0] != D:
        raise ValueError("dimension of the inputs != length of the sampled weights")
    Y = np.empty((N, len(flat)), dtype=dtype, order="F")
    _handle().apply_weights(dtype, _abi.MEM_HOST, layout, D, N, len(flat), Xa, ldx, W, max(D, 1), Y, max(N, 1))
    return Y


def _blr_and_mapping(b):
    """reference src/sampling_functions.jl:51-52"""
    if isinstance(b, BasisFunctionRegressor):
        return b.blr, b.phi
    if isinstance(b, BayesianLinearRegressor):
        return b, None
    raise TypeError("expected a BayesianLinearRegressor or BasisFunctionRegressor")


def _sample_weights(rng, blr, S):
    dtype = _dtype_of(blr.mw)
    D = blr.mw.shape[0]
    mw = _mean_vector(blr.mw, D, dtype)
    Lw, prior_kind, ldl = _prior(blr.Lw, D, dtype, need_cholesky=True)  # sampling_functions.jl:29,35,44 _cholesky(Lw)
    Z = _randn(rng, D, S, dtype)
    W = np.empty((D, S), dtype=dtype, order="F")
    _handle().sample_weights(dtype, _abi.MEM_HOST, D, S, prior_kind, mw, Lw, ldl, Z, D, W, D)
    return W


def rand(rng, f, *dims):
    """rand(rng, fx[, S]) -- reference :49-53;  rand(rng, f[, dims...]) -- sampling_functions.jl:27-38."""
    if isinstance(f, FiniteGP):
        if len(dims) == 0:
            return rand(rng, f, 1)[:, 0]
        if len(dims) != 1:
            raise TypeError("rand(rng, fx, samples::Int)")
        return _rand_finite(rng, f, int(dims[0]))
    blr, phi = _blr_and_mapping(f)
    if len(dims) == 0:
        return BLRFunctionSample(_sample_weights(rng, blr, 1)[:, 0].copy(), phi)
    if len(dims) == 1 and isinstance(dims[0], (tuple, list)):
        dims = tuple(dims[0])
    S = int(np.prod(dims))
    W = _sample_weights(rng, blr, S)
    out = np.empty(S, dtype=object)
    for i in range(S):
        out[i] = BLRFunctionSample(W[:, i].copy(), phi)
    return out.reshape(dims, order="F")


def rand_b(rng, A, f):
    """rand!(rng, A, f): fill an existing array of samples.  sampling_functions.jl:40-49"""
    blr, phi = _blr_and_mapping(f)
    W = _sample_weights(rng, blr, A.size)
    flat = A.reshape(-1, order="F")
    for i in range(A.size):
        flat[i] = BLRFunctionSample(W[:, i].copy(), phi)
    A[...] = flat.reshape(A.shape, order="F")
    return A


def rand_and_pullback(rng, fx, S):
    """(Y, pullback): the draws of `rand(rng, fx, S)` and their reverse-mode rule -- what Zygote derives through reference
    :49-53 in README.md:56-60 (``Zygote.pullback((X, Σ, mw, Λw) -> rand(rng, BLR(mw, Λw)(X, Σ), S), ...)``); a ccall is opaque
    to it, so the rule is spelled out here (julia/BLRMI355X.jl: the rrule of rand).  ``pullback(Ybar)`` with Ybar N x S returns
    dict(X, noise, mw, Lw): X in the layout of the primal container, noise a vector (Diagonal) or a scalar (isotropic), Lw a
    vector (Diagonal), the upper-factor tangent (PDMat: what flows into ``chol.factors``) or a symmetric matrix (dense).

    The two O(D N S) products run on the device as `blr_apply_weights_*` calls on re-interpreted layouts
    (Wbar = X Ybar, Xbar = W Ybar'); what is left is O(D^2 S) bookkeeping of the rule."""
    fxb = _to_finite_blr(fx)
    blr = fxb.f
    dtype = _dtype_of(blr.mw)
    X, layout, ldx, D, N = _x_layout(fxb.x, dtype)
    mw = _mean_vector(blr.mw, D, dtype)
    Lw, prior_kind, ldl = _prior(blr.Lw, D, dtype, need_cholesky=True)
    s, noise_kind = _noise(fxb.Sy, N, dtype, need_cholesky=True)
    if noise_kind == _abi.NOISE_DENSE:
        raise NotImplementedError("rand_and_pullback takes scalar or diagonal noise")
    Z1 = _randn(rng, D, S, dtype)  # FIRST draw  (reference :51)
    Z2 = _randn(rng, N, S, dtype)  # SECOND draw (reference :52)
    h = _handle()
    W = np.empty((D, S), dtype=dtype, order="F")
    h.sample_weights(dtype, _abi.MEM_HOST, D, S, prior_kind, mw, Lw, ldl, Z1, D, W, D)
    Y = np.empty((N, S), dtype=dtype, order="F")
    h.rand(dtype, _abi.MEM_HOST, layout, D, N, S, X, ldx, noise_kind, s, prior_kind, mw, Lw, ldl, Z1, D, Z2, N, Y, N)
    flip = _abi.LAYOUT_ROWVECS if layout == _abi.LAYOUT_COLVECS else _abi.LAYOUT_COLVECS

    def pullback(Ybar):
        Ybar = np.asarray(Ybar, dtype=dtype)
        if Ybar.shape != (N, S):
            raise ValueError("the cotangent must have the shape of the draws (N x S)")
        # Wbar (D x S) = X Ybar: the same memory read as the design matrix of N "features" at D "inputs" (layout flipped)
        Yb_f = np.asfortranarray(Ybar)
        Wbar = np.empty((D, S), dtype=dtype, order="F")
        h.apply_weights(dtype, _abi.MEM_HOST, flip, N, D, S, X, ldx, Yb_f, max(N, 1), Wbar, max(D, 1))
        # Xbar = W Ybar' in the layout of X: contraction over the S draws
        if layout == _abi.LAYOUT_COLVECS:   # out[d + n D]: "inputs" d, "draws" n, design matrix W read as S x D RowVecs
            Xbar = np.empty((D, N), dtype=dtype, order="F")
            Yb_t = np.ascontiguousarray(Ybar)  # (N, S) C order == S x N column-major
            h.apply_weights(dtype, _abi.MEM_HOST, _abi.LAYOUT_ROWVECS, S, D, N, W, max(D, 1), Yb_t, max(S, 1), Xbar, max(D, 1))
        else:                               # out[n + d N]: "inputs" n, "draws" d, design matrix Ybar read as S x N RowVecs
            Xbar = np.empty((N, D), dtype=dtype, order="F")
            W_t = np.ascontiguousarray(W)      # (D, S) C order == S x D column-major
            h.apply_weights(dtype, _abi.MEM_HOST, _abi.LAYOUT_ROWVECS, S, N, D, Yb_f, max(N, 1), W_t, max(S, 1), Xbar, max(N, 1))
        gX = Xbar if Xbar.shape == np.shape(X) else Xbar.T
        dmw = Wbar.sum(axis=1, dtype=np.float64).astype(dtype)
        V = W - mw[:, None]  # Uw \ Z1
        if prior_kind == _abi.PRIOR_DIAGONAL:
            gL = (-0.5 * np.sum(Wbar.astype(np.float64) * V, axis=1) / Lw).astype(dtype)  # Uw = diag(sqrt(d))
        else:
            if prior_kind == _abi.PRIOR_UPPER_FACTOR:
                U = np.triu(Lw).astype(np.float64)
            else:  # Uw = chol(Lw).U from the library: the posterior update on zero observations (as ResidentPosterior does)
                Tf = np.zeros((D, D), dtype=dtype, order="F")
                lp0 = np.zeros(1)
                rc = h.posterior(dtype, _abi.LAYOUT_COLVECS, D, 0, None, max(D, 1), None, _abi.NOISE_ISOTROPIC, np.ones(1, dtype=dtype),
                                 prior_kind, mw, Lw, ldl, None, Tf, max(D, 1), None, max(D, 1), lp0)
                if rc > 0:
                    raise _abi.PosDefException(rc)
                U = np.triu(Tf).astype(np.float64)
            # D x D bookkeeping of the rule on the host (like the prior tangent of logpdf_and_gradient): Ubar = -triu(Uw^-T Wbar V')
            Ui = _upper_inverse_on_device(h, dtype, _abi.PRIOR_UPPER_FACTOR, U.astype(dtype), D)  # Uw^-1 from the device; the rest is D x D products
            Ubar = -np.triu(Ui.T @ Wbar.astype(np.float64) @ V.T)
            if prior_kind == _abi.PRIOR_UPPER_FACTOR:
                gL = Ubar.astype(dtype)
            else:  # through the Cholesky A = L L', L = Uw': Abar = sym(L^-T Phi(L' Lbar) L^-1), Phi = lower triangle, halved diagonal
                M = np.tril(U @ Ubar.T)
                M[np.diag_indices(D)] *= 0.5
                Ab = Ui @ (Ui @ M.T).T
                gL = (0.5 * (Ab + Ab.T)).astype(dtype)
        sd = np.sum(Ybar.astype(np.float64) * Z2, axis=1) / (2.0 * np.sqrt(np.broadcast_to(s, (N,)).astype(np.float64)))
        gs = sd.astype(dtype) if noise_kind == _abi.NOISE_DIAGONAL else dtype(sd.sum())
        return {"X": np.asarray(gX), "noise": gs, "mw": dmw, "Lw": gL}

    return Y, pullback


def _rand_finite(rng, fx, S):
    fx = _to_finite_blr(fx)
    blr = fx.f
    dtype = _dtype_of(blr.mw)
    X, layout, ldx, D, N = _x_layout(fx.x, dtype)
    mw = _mean_vector(blr.mw, D, dtype)
    Lw, prior_kind, ldl = _prior(blr.Lw, D, dtype, need_cholesky=True)  # :51 _cholesky(Lw) ...
    s, noise_kind = _noise(fx.Sy, N, dtype, need_cholesky=True)          # ... then :52 _cholesky(Sigma_y)
    Z1 = _randn(rng, D, S, dtype)  # FIRST draw  (reference :51)
    Z2 = _randn(rng, N, S, dtype)  # SECOND draw (reference :52)
    Y = np.empty((N, S), dtype=dtype, order="F")
    if noise_kind == _abi.NOISE_DENSE:
        _handle().rand_dense_noise(dtype, _abi.MEM_HOST, layout, D, N, S, X, ldx, s, max(N, 1), prior_kind, mw, Lw, ldl, Z1, D, Z2, N, Y, N)
        return Y
    _handle().rand(dtype, _abi.MEM_HOST, layout, D, N, S, X, ldx, noise_kind, s, prior_kind, mw, Lw, ldl, Z1, D, Z2, N,
                   Y, N)
    return Y
