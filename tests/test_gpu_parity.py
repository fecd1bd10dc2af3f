"""Parity of the HIP path (through the C ABI) against the CPU oracle.  All tests need an MI355X."""
import numpy as np
import pytest

from oracle import blr_oracle as O

pytestmark = pytest.mark.gpu

# fp64: the GPU forms A = Lw + X S X' directly; the oracle's literal sequence whitens first.  Both are
# backward stable; on the well-conditioned toy problems they agree to ~1e-12.  Tolerances are stated per test.
RTOL64 = 1e-10
RTOL32 = 2e-4


@pytest.fixture(scope="module")
def B():
    import blr_amd

    blr_amd._abi.default_handle()  # raises if the extension or the GPU is missing: no silent fallback
    return blr_amd


@pytest.fixture
def opt(B):
    """Run-time switches of the process-wide handle (blr_set_option), restored to their defaults after the test.  The library
    reads BLR_MI355X_* from the environment only once, in blr_create -- never on a launch path -- so tests set them here."""
    h = B._abi.default_handle()
    touched = []

    def set_(key, value):
        h.set_option(key, value)
        touched.append(key)

    yield set_
    for key in touched:
        h.set_option(key, None)


def _rng(i=0):
    return np.random.Generator(np.random.PCG64(987654 + i))


def _prior_variants(B, Lw):
    U = O.chol_upper(Lw)
    return {"dense": Lw, "symmetric": B.Symmetric(Lw), "pdmat": B.PDMat(U)}


def _x_variants(B, X):
    return {
        "matrix_F": np.asfortranarray(X),
        "matrix_C": np.ascontiguousarray(X),
        "colvecs": B.ColVecs(np.asfortranarray(X)),
        "rowvecs_C": B.RowVecs(np.ascontiguousarray(X.T)),
        "rowvecs_F": B.RowVecs(np.asfortranarray(X.T)),
    }


def test_readme_example_c1(B):
    # BASELINE config 1: D=2, N=10, Diagonal prior, heteroscedastic Diagonal noise (reference README.md:44-51)
    rng = _rng(1)
    N = 10
    X = np.vstack([np.linspace(-5.0, 5.0, N), np.ones(N)])
    s = np.exp(rng.standard_normal(N))
    f = B.BayesianLinearRegressor(np.zeros(2), B.Diagonal(np.ones(2)))
    fX = f(B.ColVecs(X), B.Diagonal(s))
    y = B.rand(rng, fX)
    assert y.shape == (N,)
    lp = B.logpdf(fX, y)
    assert lp == pytest.approx(O.logpdf_literal(np.zeros(2), np.ones(2), X, s, y), rel=1e-12)
    fp = B.posterior(fX, y)
    mw_o, T_o, L_o = O.posterior_literal(np.zeros(2), np.ones(2), X, s, y)
    np.testing.assert_allclose(fp.mw, mw_o, rtol=1e-12)
    np.testing.assert_allclose(fp.Lw.toarray(), L_o, rtol=1e-12)
    assert isinstance(fp.Lw, B.Symmetric)
    # posterior predictive marginals at new inputs (README.md:80-86)
    Xp = np.vstack([np.linspace(-6.0, 6.0, 1000), np.ones(1000)])
    eps = np.finfo(float).eps
    ms = B.marginals(fp(B.ColVecs(Xp), eps))
    m_o, v_o = O.mean(mw_o, Xp), O.var(mw_o, L_o, Xp, eps)
    np.testing.assert_allclose([n.mu for n in ms], m_o, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(B.std(ms), np.sqrt(v_o), rtol=1e-11)


def test_doctest_golden_vector_on_gpu(B):
    # reference src/basis_function_regression.jl:11-28
    x = B.RowVecs(np.linspace(-1.0, 1.0, 5)[:, None])
    blr = B.BayesianLinearRegressor(np.zeros(2), B.Diagonal(np.ones(2)))
    phi = lambda x: B.RowVecs(np.column_stack([np.ones(len(x)), np.prod(x.X, axis=1)]))
    bfr = B.BasisFunctionRegressor(blr, phi)
    np.testing.assert_allclose(B.var(bfr(x)), [2.0, 1.25, 1.0, 1.25, 2.0], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(B.var(bfr(x)), B.var(blr(phi(x))))  # README.md:120


@pytest.mark.parametrize("N,D", [(11, 3), (13, 7), (11, 2), (10, 2), (40, 16), (5, 9), (100, 17), (33, 100), (257, 128)])
@pytest.mark.parametrize("prior", ["dense", "symmetric", "pdmat", "diagonal"])
def test_posterior_logpdf_vs_oracle_f64(B, N, D, prior):
    rng = _rng(N * 1000 + D)
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    if prior == "diagonal":
        dvec = np.exp(rng.standard_normal(D))
        Lw, Lw_arg = np.diag(dvec), B.Diagonal(dvec)
    else:
        Lw_arg = _prior_variants(B, Lw)[prior]
    y = rng.standard_normal(N)
    mw_o, T_o, L_o = O.posterior_literal(mw, Lw, X, s, y)
    lp_o = O.logpdf_literal(mw, Lw, X, s, y)
    f = B.BayesianLinearRegressor(mw, Lw_arg)
    for name, x in _x_variants(B, X).items():
        for Sy in (s, B.Diagonal(s)):
            fx = f(x, Sy)
            assert B.logpdf(fx, y) == pytest.approx(lp_o, rel=RTOL64), name
        fp = B.posterior(f(x, s), y)
        np.testing.assert_allclose(fp.mw, mw_o, rtol=1e-9, atol=1e-11, err_msg=name)
        if prior == "pdmat":
            assert isinstance(fp.Lw, B.PDMat)  # reference :93 / test :109
            np.testing.assert_allclose(fp.Lw.U, T_o, rtol=1e-9, atol=1e-11)
            assert np.all(np.tril(fp.Lw.U, -1) == 0)
        else:
            assert isinstance(fp.Lw, B.Symmetric)  # reference :92 / test :110
            np.testing.assert_allclose(fp.Lw.toarray(), L_o, rtol=1e-9, atol=1e-11)
    # isotropic noise
    lp_iso = B.logpdf(f(X, 0.37), y)
    assert lp_iso == pytest.approx(O.logpdf_literal(mw, Lw, X, 0.37, y), rel=RTOL64)


@pytest.mark.parametrize("N,D", [(13, 7), (64, 32), (300, 128)])
def test_posterior_logpdf_vs_oracle_f32(B, N, D):
    rng = _rng(77 + D)
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False, dtype=np.float32)
    y = rng.standard_normal(N).astype(np.float32)
    # the oracle runs in fp64 on the fp32-rounded inputs: tolerance is fp32 accumulation error
    mw_o, T_o, L_o, lp_o = O.posterior_logpdf_direct(mw.astype(float), Lw.astype(float), X.astype(float),
                                                     s.astype(float), y.astype(float))
    f = B.BayesianLinearRegressor(mw, Lw)
    for x in (np.asfortranarray(X), B.RowVecs(np.ascontiguousarray(X.T)), np.ascontiguousarray(X)):
        fx = f(x, s)
        lp = B.logpdf(fx, y)
        assert isinstance(lp, float)
        assert lp == pytest.approx(lp_o, rel=RTOL32)
        fp = B.posterior(fx, y)
        assert fp.mw.dtype == np.float32
        # tolerance derived from fp32 LAPACK on the same inputs (see _fp32_lapack_yardstick), floor = a few fp32 ulps
        yard, _, _ = _fp32_lapack_yardstick(mw, Lw, X, s, y, mw_o, L_o, lp_o)
        e_gpu = _rel_errs(fp.mw, fp.Lw.toarray(), lp, mw_o, L_o, lp_o)
        assert e_gpu[0] <= 4 * yard[0] + 5e-7, (e_gpu, yard)
        assert e_gpu[1] <= 4 * yard[1] + 5e-7, (e_gpu, yard)


def test_c2_shape_fp64(B):
    # BASELINE config 2: D=128, N=4096 ColVecs, isotropic noise, fp64 -- Gram + Cholesky vs CPU
    rng = _rng(2)
    D, N = 128, 4096
    X = np.asfortranarray(rng.standard_normal((D, N)))
    w = rng.standard_normal(D)
    y = X.T @ w + np.sqrt(0.1) * rng.standard_normal(N)
    mw = rng.standard_normal(D)
    f = B.BayesianLinearRegressor(mw, B.Diagonal(np.ones(D)))
    fx = f(B.ColVecs(X), 0.1)
    lp = B.logpdf(fx, y)
    fp = B.posterior(fx, y)
    mw_o, T_o, L_o, lp_o = O.posterior_logpdf_direct(mw, np.ones(D), X, 0.1, y)
    # The evidence is delta'delta / s - |u|^2 + ...: on this instance (data explained by the weights) two terms of 1e7 leave 1.9e3.
    # The documented contract is 1e-10 of the evidence (include/blr_mi355x.h; asserted against the literal sequence below); the sharper
    # check here scales with what cancels: 1e-11 of the evidence + 1e-14 of delta'delta / s.  (Six digit groups on the int8 route put
    # A within 4e-14 of its diagonal scale, which this instance turns into 1.5e-11 of the evidence; the fp64 kernel sits at 1e-12.)
    dlt = y - X.T @ mw
    assert abs(lp - lp_o) <= 1e-11 * abs(lp_o) + 1e-14 * float(dlt @ dlt) / 0.1
    np.testing.assert_allclose(fp.mw, mw_o, rtol=1e-9, atol=1e-12)
    # A = Lw + X X' / s: every entry within 1e-13 of the scale of its row and column (the int8-sliced Gram's error model: 48 bits
    # per input relative to its ROW's bound -- an entry that nearly cancels is off by that much of sqrt(A_ii A_jj), not of itself),
    # and within the documented 1e-9 of itself
    dA = np.sqrt(np.diag(L_o))
    assert (np.abs(fp.Lw.toarray() - L_o) / np.outer(dA, dA)).max() <= 1e-13
    np.testing.assert_allclose(fp.Lw.toarray(), L_o, rtol=1e-9)
    assert lp == pytest.approx(O.logpdf_literal(mw, np.ones(D), X, 0.1, y), rel=1e-10)
    # determinism: fixed accumulation order, no float atomics -> bitwise reproducible
    assert B.logpdf(fx, y) == lp
    np.testing.assert_array_equal(B.posterior(fx, y).mw, fp.mw)


def test_logpdf_naive_identity(B):
    # reference test/bayesian_linear_regression.jl:22-38, on the GPU path
    rng = _rng(3)
    N, D = 13, 7
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    y = B.rand(rng, B.BayesianLinearRegressor(mw, Lw)(X, s))
    lp = B.logpdf(B.BayesianLinearRegressor(mw, Lw)(X, s), y)
    assert lp == pytest.approx(O.logpdf_naive(mw, Lw, X, s, y), rel=1.5e-8)
    assert lp == pytest.approx(O.logpdf_naive_mp(mw, Lw, X, s, y), rel=1e-11)


def test_posterior_low_noise(B):
    # reference test/bayesian_linear_regression.jl:40-48 (cov -> var: the N x N cov is out of GPU scope)
    rng = _rng(4)
    N, D = 13, 7
    X, mw, Lw, _ = O.generate_toy_problem(rng, N, D)
    eps = np.finfo(float).eps
    f = B.BayesianLinearRegressor(mw, Lw)
    y = B.rand(rng, f(X, eps))
    fp = B.posterior(f(X, eps), y)
    # The reference asserts mean ~ y at rtol sqrt(eps) on ITS seed; how closely the posterior mean
    # interpolates depends on the instance (the reference's own op sequence gives 1.9e-8 on this one),
    # so: interpolation at 1e-7, and agreement with the reference op sequence at 1e-9.
    np.testing.assert_allclose(B.mean(fp(X, eps)), y, rtol=1e-7)
    mw_o, _, L_o = O.posterior_literal(mw, Lw, X, eps, y)
    np.testing.assert_allclose(B.mean(fp(X, eps)), O.mean(mw_o, X), rtol=1e-9)
    np.testing.assert_allclose(fp.mw, mw_o, rtol=1e-7)
    assert np.all(B.var(fp(X, eps)) < 1000 * eps)
    assert np.all(O.cov(fp.mw, fp.Lw.toarray(), X, eps) < 1000 * eps)


def test_posterior_repeated_conditioning(B):
    # reference test/bayesian_linear_regression.jl:49-70 with diagonal noise
    rng = _rng(5)
    N, D = 13, 7
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    Xp = rng.standard_normal((D, N))
    for Lw_arg in (Lw, B.PDMat(O.chol_upper(Lw))):
        f = B.BayesianLinearRegressor(mw, Lw_arg)
        y = B.rand(rng, f(X, s))
        N1 = N - 3
        f1 = B.posterior(f(X[:, :N1], s[:N1]), y[:N1])
        f2 = B.posterior(f1(X[:, N1:], s[N1:]), y[N1:])
        fo = B.posterior(f(X, s), y)
        np.testing.assert_allclose(B.mean(fo(Xp, s)), B.mean(f2(Xp, s)), rtol=1.5e-8)
        np.testing.assert_allclose(B.var(fo(Xp, s)), B.var(f2(Xp, s)), rtol=1.5e-8)
        # evidence chain rule
        lp = B.logpdf(f(X[:, :N1], s[:N1]), y[:N1]) + B.logpdf(f1(X[:, N1:], s[N1:]), y[N1:])
        assert lp == pytest.approx(B.logpdf(f(X, s), y), rel=1e-11)


def test_pdmat_symmetric_closure(B):
    # reference test/bayesian_linear_regression.jl:90-112
    rng = _rng(6)
    N, D = 13, 7
    X, Xp = rng.standard_normal((D, N)), rng.standard_normal((D, N))
    U = np.triu(rng.standard_normal((D, D)))
    mw, s = rng.standard_normal(D), np.exp(rng.standard_normal(N))
    Lw = U.T @ U + np.eye(D)
    f_pd = B.BayesianLinearRegressor(mw, B.PDMat(O.chol_upper(Lw)))
    f_sym = B.BayesianLinearRegressor(mw, B.Symmetric(Lw))
    y = B.rand(rng, f_pd(X, s))
    p_pd, p_sym = B.posterior(f_pd(X, s), y), B.posterior(f_sym(X, s), y)
    assert isinstance(p_pd.Lw, B.PDMat) and isinstance(p_sym.Lw, B.Symmetric)
    np.testing.assert_allclose(B.mean(p_pd(Xp, s)), B.mean(p_sym(Xp, s)), rtol=1.5e-8)
    np.testing.assert_allclose(B.var(p_pd(Xp, s)), B.var(p_sym(Xp, s)), rtol=1.5e-8)


@pytest.mark.parametrize("D,N", [(3, 11), (7, 200), (64, 1000), (128, 130)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_marginals_vs_oracle(B, D, N, dtype):
    rng = _rng(8 + D)
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False, dtype=dtype)
    rt = 1e-10 if dtype == np.float64 else 2e-4
    m_o = O.mean(mw.astype(float), X.astype(float))
    v_o = O.var(mw.astype(float), Lw.astype(float), X.astype(float), s.astype(float))
    for name, Lw_arg in {**_prior_variants(B, Lw)}.items():
        f = B.BayesianLinearRegressor(mw, Lw_arg)
        for xname, x in _x_variants(B, X).items():
            m, v = B.mean_and_var(f(x, s))
            assert m.dtype == dtype and v.dtype == dtype
            np.testing.assert_allclose(m, m_o, rtol=rt, atol=rt, err_msg=f"{name}/{xname}")
            np.testing.assert_allclose(v, v_o, rtol=rt, err_msg=f"{name}/{xname}")
    dvec = np.exp(rng.standard_normal(D)).astype(dtype)
    f = B.BayesianLinearRegressor(mw, B.Diagonal(dvec))
    v = B.var(f(X, dtype(0.5)))
    np.testing.assert_allclose(v, O.var(mw.astype(float), np.diag(dvec).astype(float), X.astype(float), 0.5), rtol=rt)


@pytest.mark.parametrize("D,N,S", [(3, 11, 7), (7, 13, 100), (64, 200, 65), (128, 70, 129)])
def test_rand_given_normals_vs_oracle(B, D, N, S):
    # reference :49-53 with Z1 (D x S) drawn FIRST and Z2 (N x S) second, supplied through the ABI
    from blr_amd import _abi

    rng = _rng(9 + D)
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    Z1 = np.asfortranarray(rng.standard_normal((D, S)))
    Z2 = np.asfortranarray(rng.standard_normal((N, S)))
    Y_o = O.rand(mw, Lw, X, s, Z1, Z2)
    h = _abi.default_handle()
    for layout, Xa, ldx in ((_abi.LAYOUT_COLVECS, np.asfortranarray(X), D), (_abi.LAYOUT_ROWVECS, np.ascontiguousarray(X), N)):
        Y = np.empty((N, S), order="F")
        h.rand(np.float64, _abi.MEM_HOST, layout, D, N, S, Xa, ldx, _abi.NOISE_DIAGONAL, s, _abi.PRIOR_DENSE, mw,
               np.asfortranarray(Lw), D, Z1, D, Z2, N, Y, N)
        np.testing.assert_allclose(Y, Y_o, rtol=1e-10, atol=1e-11)
    W = np.empty((D, S), order="F")
    h.sample_weights(np.float64, _abi.MEM_HOST, D, S, _abi.PRIOR_UPPER_FACTOR, mw, np.asfortranarray(O.chol_upper(Lw)), D,
                     Z1, D, W, D)
    np.testing.assert_allclose(W, O.sample_weights(mw, Lw, Z1), rtol=1e-10, atol=1e-11)


def test_rand_moments_and_rng_order(B):
    # reference test/bayesian_linear_regression.jl:11-21 (2e5 samples); also checks the draw order Z1 then Z2
    rng = _rng(10)
    N, D, S = 11, 3, 200_000
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    f = B.BayesianLinearRegressor(mw, Lw)
    r1 = np.random.Generator(np.random.PCG64(42))
    Y = B.rand(r1, f(X, s), S)
    assert Y.shape == (N, S)
    r2 = np.random.Generator(np.random.PCG64(42))
    Z1 = r2.standard_normal((S, D)).T
    Z2 = r2.standard_normal((S, N)).T
    np.testing.assert_allclose(Y, O.rand(mw, Lw, X, s, Z1, Z2), rtol=1e-9, atol=1e-10)
    m_emp = Y.mean(axis=1)
    Yc = Y - m_emp[:, None]
    np.testing.assert_allclose(B.mean(f(X, s)), m_emp, atol=2e-2, rtol=2e-2)
    np.testing.assert_allclose(O.cov(mw, Lw, X, s), Yc @ Yc.T / S, atol=3e-2, rtol=3e-2)


def test_function_samples(B):
    # reference test/sampling_functions.jl:3-24
    rng = _rng(11)
    N, D = 11, 5
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    f = B.BayesianLinearRegressor(mw, Lw)
    g = B.rand(rng, f)
    assert isinstance(g, B.BLRFunctionSample)
    assert np.array_equal(g(X), g(X))
    np.testing.assert_allclose(g(X), g(B.ColVecs(X)), rtol=1e-15)
    np.testing.assert_allclose(g(X), g(B.RowVecs(np.ascontiguousarray(X.T))), rtol=1e-13)
    np.testing.assert_allclose(g(X), X.T @ g.w, rtol=1e-13)
    gs = B.rand(rng, f, 20, 30)
    assert gs.shape == (20, 30) and isinstance(gs[3, 4], B.BLRFunctionSample)
    S = 40_000
    gs = B.rand(rng, f, S)
    W = np.stack([h.w for h in gs], axis=1)
    np.testing.assert_allclose(W.mean(axis=1), mw, atol=2e-2)
    np.testing.assert_allclose(np.cov(W), np.linalg.inv(Lw), atol=2e-2)
    A = np.empty((4, 5), dtype=object)
    A = B.rand_b(rng, A, f)
    assert all(isinstance(a, B.BLRFunctionSample) for a in A.ravel())
    # basis-function version evaluates phi(X)'w
    phi = lambda x: O.phi_test(x) if not isinstance(x, (B.ColVecs, B.RowVecs)) else type(x)(O.phi_test(
        O.ColVecs(x.X) if isinstance(x, B.ColVecs) else O.RowVecs(x.X)).X)
    X2, mw2, Lw2, _ = O.generate_toy_problem(rng, N, 2, dense_noise_cov=False)
    fb = B.BasisFunctionRegressor(B.BayesianLinearRegressor(mw2, Lw2), phi)
    gb = B.rand(rng, fb)
    np.testing.assert_allclose(gb(X2), O.phi_test(X2).T @ gb.w, rtol=1e-13)


@pytest.mark.parametrize("container", ["matrix", "colvecs", "rowvecs"])
def test_bfr_consistency_with_blr(B, container):
    # reference test/basis_function_regression.jl:13-28
    rng = _rng(12)
    N, D = 11, 2
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    x = {"matrix": X, "colvecs": B.ColVecs(X), "rowvecs": B.RowVecs(np.ascontiguousarray(X.T))}[container]

    def phi(x):
        if isinstance(x, B.RowVecs):
            return B.RowVecs(np.column_stack([np.ones(len(x)), np.prod(x.X, axis=1)]))
        if isinstance(x, B.ColVecs):
            return B.ColVecs(np.vstack([np.ones(len(x)), np.prod(x.X, axis=0)]))
        return phi(B.ColVecs(x)).X

    f = B.BayesianLinearRegressor(mw, Lw)
    f_bf = B.BasisFunctionRegressor(f, phi)
    y = B.rand(rng, f_bf(x, s))
    assert B.logpdf(f(phi(x), s), y) == pytest.approx(B.logpdf(f_bf(x, s), y), rel=1e-14)
    p_bf, p = B.posterior(f_bf(x, s), y), B.posterior(f(phi(x), s), y)
    assert isinstance(p_bf, B.BasisFunctionRegressor)
    np.testing.assert_allclose(B.mean(p_bf(x)), B.mean(p(phi(x))), rtol=1e-14)


def test_logpdf_matrix_columns(B):
    # AbstractGPs secondary API exercised by TestUtils (reference test :7-9): logpdf(fx, Y::Matrix)
    rng = _rng(13)
    N, D, S = 13, 7, 5
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    f = B.BayesianLinearRegressor(mw, Lw)
    Y = B.rand(rng, f(X, s), S)
    lps = B.logpdf(f(X, s), Y)
    assert lps.shape == (S,)
    for j in range(S):
        assert lps[j] == pytest.approx(B.logpdf(f(X, s), Y[:, j]), rel=1e-14)


def test_error_paths(B):
    rng = _rng(14)
    N, D = 11, 5
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    f = B.BayesianLinearRegressor(mw, Lw)
    # reference test :116-122: a vector of vectors is neither ColVecs nor RowVecs
    with pytest.raises(TypeError):
        B.rand(rng, f([row for row in X.T], s))
    with pytest.raises(ValueError):  # reference :74
        B.logpdf(f(X, s), np.zeros(N - 1))
    # PosDefException from the prior (:78) and from the posterior precision (:86)
    with pytest.raises(B.PosDefException) as ei:
        B.logpdf(B.BayesianLinearRegressor(mw, -Lw)(X, s), np.zeros(N))
    assert ei.value.info == 1
    bad = Lw.copy()
    bad[3, 3] = -50.0
    with pytest.raises(np.linalg.LinAlgError) as ei:
        B.posterior(B.BayesianLinearRegressor(mw, bad)(X, s), np.zeros(N))
    assert ei.value.info == 4
    with pytest.raises(B.PosDefException):
        B.var(B.BayesianLinearRegressor(mw, -Lw)(X, s))
    # dense Sigma_y and cov(fx) run on the device since round 2 (blr_posterior_dense_noise_*, blr_mean_and_cov_*)
    assert B.logpdf(f(X, np.eye(N)), np.zeros(N)) == pytest.approx(O.logpdf_literal(mw, Lw, X, np.eye(N), np.zeros(N)), rel=1e-11)
    np.testing.assert_allclose(B.cov(f(X, s)), O.cov(mw, Lw, X, s), rtol=1e-10, atol=1e-12)
    with pytest.raises(ValueError):  # a noise covariance of the wrong size
        B.logpdf(f(X, np.eye(N + 1)), np.zeros(N))
    with pytest.raises(NotImplementedError):  # the closed-form gradient covers isotropic / Diagonal noise only
        B.logpdf_and_gradient(f(X, np.eye(N)), np.zeros(N))


def test_abi_argument_errors(B):
    from blr_amd import _abi

    h = _abi.default_handle()
    D, N = 4, 8
    X = np.zeros((D, N), order="F")
    v = np.zeros(N)
    lp = np.zeros(1)
    with pytest.raises(_abi.BLRError) as ei:  # ldx < D -> argument 6 of the single-problem form
        h.posterior(np.float64, _abi.LAYOUT_COLVECS, D, N, X, D - 1, v, _abi.NOISE_DIAGONAL, v, _abi.PRIOR_DIAGONAL,
                    np.zeros(D), np.ones(D), 1, None, None, D, None, D, lp)
    assert ei.value.code == -6
    with pytest.raises(_abi.BLRError) as ei:  # unknown layout (reference :26-31)
        h.posterior(np.float64, 7, D, N, X, D, v, _abi.NOISE_DIAGONAL, v, _abi.PRIOR_DIAGONAL, np.zeros(D), np.ones(D), 1,
                    None, None, D, None, D, lp)
    assert ei.value.code == -2
    with pytest.raises(_abi.BLRError) as ei:  # D beyond this build
        h.posterior(np.float64, _abi.LAYOUT_COLVECS, 100000, N, X, 100000, v, _abi.NOISE_DIAGONAL, v, _abi.PRIOR_DIAGONAL,
                    np.zeros(D), np.ones(D), 1, None, None, D, None, D, lp)
    assert ei.value.code == -3


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_device_path_c4_shape(B, dtype):
    # BASELINE config 4 shape (D=64, N=1024), a 96-regressor slice, device-resident, one launch
    import torch

    from blr_amd import _abi

    rng = _rng(15)
    Bn, D, N = 96, 64, 1024
    X = rng.standard_normal((Bn, N, D)).astype(dtype)  # each [N, D] C-order == D x N column-major
    w = rng.standard_normal((Bn, D))
    s = np.exp(0.3 * rng.standard_normal((Bn, N))).astype(dtype)
    y = (np.einsum("bnd,bd->bn", X, w) + np.sqrt(s) * rng.standard_normal((Bn, N))).astype(dtype)
    mw = rng.standard_normal((Bn, D)).astype(dtype)
    dprior = np.exp(0.2 * rng.standard_normal((Bn, D))).astype(dtype)
    bad = 17
    dprior[bad, 5] = -1.0  # one non-SPD regressor must not poison the batch
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    tX, ty, ts, tmw, td = (torch.from_numpy(a).to(dev) for a in (X, y, s, mw, dprior))
    t_mwp = torch.empty((Bn, D), dtype=tdt, device=dev)
    t_T = torch.empty((Bn, D, D), dtype=tdt, device=dev)
    t_A = torch.empty((Bn, D, D), dtype=tdt, device=dev)
    t_lp = torch.empty(Bn, dtype=torch.float64, device=dev)
    t_info = torch.full((Bn,), -7, dtype=torch.int32, device=dev)
    h = _abi.default_handle()
    h.posterior_batched(dtype, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, Bn, D, N, tX.data_ptr(), D, N * D, ty.data_ptr(), N,
                        _abi.NOISE_DIAGONAL, ts.data_ptr(), N, _abi.PRIOR_DIAGONAL, tmw.data_ptr(), D, td.data_ptr(), 1, D,
                        t_mwp.data_ptr(), D, t_T.data_ptr(), D, D * D, t_A.data_ptr(), D, D * D, t_lp.data_ptr(),
                        t_info.data_ptr())
    info = t_info.cpu().numpy()
    assert info[bad] == 6 and np.all(np.delete(info, bad) == 0)
    lp = t_lp.cpu().numpy()
    assert np.isnan(lp[bad])
    mwp, Tm, Am = t_mwp.cpu().numpy(), t_T.cpu().numpy(), t_A.cpu().numpy()
    rt_lp, rt = 1e-11, 1e-9
    for b in list(range(0, Bn, 7)) + [Bn - 1]:
        if b == bad:
            continue
        if dtype == np.float32:  # bound = 4x fp32 LAPACK's own error on the same inputs, not a hand-picked tolerance
            _assert_fp32_within_lapack(mw[b], dprior[b], np.asfortranarray(X[b].T), s[b], y[b], mwp[b], Am[b].T, lp[b],
                                       got_T=Tm[b].T, what=f"c4 slice, regressor {b}")
            continue
        Xb = X[b].T.astype(float)
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw[b].astype(float), dprior[b].astype(float), Xb,
                                                         s[b].astype(float), y[b].astype(float))
        assert lp[b] == pytest.approx(lp_o, rel=rt_lp)
        np.testing.assert_allclose(mwp[b], mw_o, rtol=rt, atol=rt * 1e-2)
        np.testing.assert_allclose(Tm[b].T, T_o, rtol=rt, atol=rt)  # [D, D] C-order holds the column-major T
        np.testing.assert_allclose(Am[b].T, A_o, rtol=rt, atol=rt)
    # fixed-order log-evidence sum on the device
    good = np.delete(np.arange(Bn), bad)
    t_good = t_lp[torch.from_numpy(good).to(dev)].contiguous()
    t_tot = torch.zeros(1, dtype=torch.float64, device=dev)
    h.logpdf_sum(_abi.MEM_DEVICE, len(good), t_good.data_ptr(), t_tot.data_ptr())
    tot = float(t_tot.cpu()[0])
    assert tot == pytest.approx(float(np.sum(lp[good])), rel=1e-13)
    h.logpdf_sum(_abi.MEM_DEVICE, len(good), t_good.data_ptr(), t_tot.data_ptr())
    assert float(t_tot.cpu()[0]) == tot  # bitwise reproducible


def test_full_size_properties_c2_batch(B):
    # BASELINE full size (D=128, N=4096, fp64), 32 regressors: size-independent properties --
    # (1) evidence chain rule over a split of the columns, carrying the factor T forward (reference :93);
    # (2) T'T == A;  (3) A mw' == Lw mw + X S y  (normal equations).
    import torch

    from blr_amd import _abi

    rng = _rng(16)
    Bn, D, N, N1 = 32, 128, 4096, 3000
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1234)
    X = torch.randn((Bn, N, D), generator=g, dtype=torch.float64)
    w = torch.randn((Bn, D), generator=g, dtype=torch.float64)
    y = torch.einsum("bnd,bd->bn", X, w) + 0.3 * torch.randn((Bn, N), generator=g, dtype=torch.float64)
    mw = torch.randn((Bn, D), generator=g, dtype=torch.float64)
    s2 = torch.tensor([0.09], dtype=torch.float64)
    ones = torch.ones((Bn, D), dtype=torch.float64)
    tX, ty, tmw, ts, td = (a.to(dev) for a in (X, y, mw, s2, ones))
    h = _abi.default_handle()

    def run(n0, n1, prior_kind, t_prior_mw, t_prior, ldl, strideL):
        t_mwp = torch.empty((Bn, D), dtype=torch.float64, device=dev)
        t_T = torch.empty((Bn, D, D), dtype=torch.float64, device=dev)
        t_A = torch.empty((Bn, D, D), dtype=torch.float64, device=dev)
        t_lp = torch.empty(Bn, dtype=torch.float64, device=dev)
        t_info = torch.empty(Bn, dtype=torch.int32, device=dev)
        h.posterior_batched(np.float64, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, Bn, D, n1 - n0,
                            tX.data_ptr() + n0 * D * 8, D, N * D, ty.data_ptr() + n0 * 8, N, _abi.NOISE_ISOTROPIC,
                            ts.data_ptr(), 0, prior_kind, t_prior_mw.data_ptr(), D, t_prior.data_ptr(), ldl, strideL,
                            t_mwp.data_ptr(), D, t_T.data_ptr(), D, D * D, t_A.data_ptr(), D, D * D, t_lp.data_ptr(),
                            t_info.data_ptr())
        assert int(t_info.abs().sum()) == 0
        return t_mwp, t_T, t_A, t_lp

    m_all, T_all, A_all, lp_all = run(0, N, _abi.PRIOR_DIAGONAL, tmw, td, 1, D)
    m1, T1, A1, lp1 = run(0, N1, _abi.PRIOR_DIAGONAL, tmw, td, 1, D)
    m2, T2, A2, lp2 = run(N1, N, _abi.PRIOR_UPPER_FACTOR, m1, T1, D, D * D)
    # (the evidence here is ~1.6e3, the two terms that cancel in it -- delta'S delta against |u|^2, prior mean ~ N(0, I) far from the
    # weights that explain y -- ~1.2e7 each: one part in 1e15 of THOSE is 1e-11 of the evidence, on the fp64 and the int8 path alike)
    torch.testing.assert_close(lp1 + lp2, lp_all, rtol=3e-11, atol=0)
    torch.testing.assert_close(m2, m_all, rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(A2, A_all, rtol=1e-11, atol=1e-9)
    Tm = T_all.transpose(1, 2)  # [b] C-order holds column-major T -> transpose gives T as a matrix
    torch.testing.assert_close(Tm.transpose(1, 2) @ Tm, A_all.transpose(1, 2), rtol=1e-11, atol=1e-9)
    Xm = tX.transpose(1, 2)  # D x N
    rhs = tmw + torch.einsum("bdn,bn->bd", Xm, ty) / 0.09
    lhs = torch.einsum("bij,bj->bi", A_all, m_all)  # A symmetric
    torch.testing.assert_close(lhs, rhs, rtol=1e-9, atol=1e-7)
    A_ref = torch.eye(D, dtype=torch.float64, device=dev)[None] + torch.einsum("bdn,ben->bde", Xm, Xm) / 0.09
    torch.testing.assert_close(A_all, A_ref, rtol=1e-11, atol=1e-9)


# ---- large-D path (D > 128): split-K MFMA Gram tiles + blocked global Cholesky (BASELINE configs 3 and 5) ----------
@pytest.mark.parametrize("N,D", [(50, 129), (300, 200), (1000, 256), (700, 300), (2500, 384)])
@pytest.mark.parametrize("prior", ["diagonal", "dense", "pdmat"])
def test_large_d_posterior_logpdf_f64(B, N, D, prior):
    rng = _rng(5000 + N + D)
    X = rng.standard_normal((D, N))
    mw = rng.standard_normal(D)
    s = np.exp(0.5 * rng.standard_normal(N))
    y = X.T @ rng.standard_normal(D) + np.sqrt(s) * rng.standard_normal(N)
    if prior == "diagonal":
        dvec = np.exp(rng.standard_normal(D))
        Lw, Lw_arg = np.diag(dvec), B.Diagonal(dvec)
    else:
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        Lw = Bm @ Bm.T + np.eye(D)
        Lw_arg = Lw if prior == "dense" else B.PDMat(O.chol_upper(Lw))
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw, Lw, X, s, y)
    f = B.BayesianLinearRegressor(mw, Lw_arg)
    for x in (np.asfortranarray(X), B.RowVecs(np.asfortranarray(X.T))):  # ColVecs (LDS-DMA) and RowVecs (generic) loaders
        fx = f(x, s)
        assert B.logpdf(fx, y) == pytest.approx(lp_o, rel=1e-10)
        fp = B.posterior(fx, y)
        np.testing.assert_allclose(fp.mw, mw_o, rtol=1e-8, atol=1e-10)
        if prior == "pdmat":
            np.testing.assert_allclose(fp.Lw.U, T_o, rtol=1e-8, atol=1e-9)
            assert np.all(np.tril(fp.Lw.U, -1) == 0)
        else:
            np.testing.assert_allclose(fp.Lw.toarray(), A_o, rtol=1e-10, atol=1e-10)
    assert B.logpdf(f(np.asfortranarray(X), 0.3), y) == pytest.approx(O.logpdf_literal(mw, Lw, X, 0.3, y), rel=1e-10)
    # sequential conditioning through the large path, carrying the factor (reference :93)
    if prior == "pdmat":
        N1 = N // 2
        f1 = B.posterior(f(np.asfortranarray(X[:, :N1]), s[:N1]), y[:N1])
        f2 = B.posterior(f1(np.asfortranarray(X[:, N1:]), s[N1:]), y[N1:])
        np.testing.assert_allclose(f2.mw, mw_o, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("noise", ["iso", "diag"])
def test_large_d_zero_prior_mean(B, dtype, noise):
    # A zero prior mean takes the branch of the column-statistics kernel that never reads X (delta = y exactly); the result
    # has to be the oracle's all the same, in both layouts, and has to agree with a prior mean that is merely tiny.
    rng = _rng(5151)
    D, N = 320, 900
    X = rng.standard_normal((D, N)).astype(dtype)
    s = np.exp(0.4 * rng.standard_normal(N)).astype(dtype) if noise == "diag" else dtype(0.37)
    y = (X.T.astype(float) @ rng.standard_normal(D) / np.sqrt(D) + rng.standard_normal(N)).astype(dtype)
    dvec = np.exp(0.3 * rng.standard_normal(D)).astype(dtype)
    s_full = s if noise == "diag" else np.full(N, float(s))
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(np.zeros(D), np.diag(dvec.astype(float)), X.astype(float), np.asarray(s_full, float),
                                                     y.astype(float))
    f = B.BayesianLinearRegressor(np.zeros(D, dtype), B.Diagonal(dvec))
    tol = 1e-9 if dtype == np.float64 else 2e-3
    for x in (np.asfortranarray(X), B.RowVecs(np.asfortranarray(X.T))):
        fx = f(x, s)
        assert B.logpdf(fx, y) == pytest.approx(lp_o, rel=tol)
        np.testing.assert_allclose(B.posterior(fx, y).mw, mw_o, rtol=tol * 10, atol=tol)
    f_tiny = B.BayesianLinearRegressor(np.full(D, 1e-30, dtype), B.Diagonal(dvec))  # the general branch, same numbers
    assert B.logpdf(f_tiny(np.asfortranarray(X), s), y) == pytest.approx(B.logpdf(f(np.asfortranarray(X), s), y), rel=tol)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("col", [0, 5, 16, 130, 255, 300, 329])
def test_large_d_factorisation_names_the_first_bad_pivot(B, dtype, col):
    # The panel kernel finds a non-positive pivot as a NaN on the factor's diagonal and has to name its column (1-based, as
    # LAPACK's info) wherever it falls: first tile of the first panel, a tile boundary, a later panel, the ragged last one.
    rng = _rng(5200 + col)
    D, N = 330, 400
    X = rng.standard_normal((D, N)).astype(dtype)
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = Bm @ Bm.T + np.eye(D)
    # make the leading minor of order col + 1 the first that is not positive definite
    Lc = np.linalg.cholesky(Lw)
    Lc[col, col] = 0.0
    Lw_bad = Lc @ Lc.T
    Lw_bad[col, col] -= 1.0
    fx_bad = B.BayesianLinearRegressor(np.zeros(D, dtype), Lw_bad.astype(dtype))(np.asfortranarray(X), dtype(0.5))
    with pytest.raises(B.PosDefException) as ei:
        B.posterior(fx_bad, np.zeros(N, dtype))
    assert ei.value.info == col + 1
    with pytest.raises(B.PosDefException) as ei:  # logpdf alone: no back substitution is launched, the status still arrives
        B.logpdf(fx_bad, np.zeros(N, dtype))
    assert ei.value.info == col + 1


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_large_d_beyond_4096_rows_below_a_panel(B, dtype):
    # More than 16 x 256 rows below the first blocks: those panels run with 32 rows of the block column per workgroup instead
    # of 16 (chol_large), the later ones with 16.  Evidence and posterior mean against the oracle.
    rng = _rng(5400)
    D, N = 4224, 160
    X = (rng.standard_normal((D, N)) / np.sqrt(D)).astype(dtype)
    y = rng.standard_normal(N).astype(dtype)
    dvec = np.exp(0.2 * rng.standard_normal(D)).astype(dtype)
    mw = (0.05 * rng.standard_normal(D)).astype(dtype)
    f = B.BayesianLinearRegressor(mw, B.Diagonal(dvec))
    fx = f(np.asfortranarray(X), dtype(0.25))
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw.astype(float), np.diag(dvec.astype(float)), X.astype(float), np.full(N, 0.25),
                                                     y.astype(float))
    tol = 1e-9 if dtype == np.float64 else 2e-3
    assert B.logpdf(fx, y) == pytest.approx(lp_o, rel=tol)
    np.testing.assert_allclose(B.posterior(fx, y).mw, mw_o, rtol=10 * tol, atol=tol)


def test_large_d_f32_c3_shape_reduced(B):
    # BASELINE config 3 shape family (D=1024, diagonal noise, fp32) at a reduced N the oracle finishes in seconds
    rng = _rng(6001)
    D, N = 1024, 8192
    X = rng.standard_normal((D, N)).astype(np.float32)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(np.float32)
    y = (X.T.astype(float) @ rng.standard_normal(D) / np.sqrt(D) + np.sqrt(s) * rng.standard_normal(N)).astype(np.float32)
    mw = np.zeros(D, dtype=np.float32)
    dvec = np.ones(D, dtype=np.float32)
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw.astype(float), dvec.astype(float), X.astype(float), s.astype(float),
                                                     y.astype(float))
    f = B.BayesianLinearRegressor(mw, B.Diagonal(dvec))
    fx = f(B.ColVecs(np.asfortranarray(X)), s)
    lp = B.logpdf(fx, y)
    assert lp == pytest.approx(lp_o, rel=2e-4)
    fp = B.posterior(fx, y)
    yard, _, _ = _fp32_lapack_yardstick(mw, dvec, X, s, y, mw_o, A_o, lp_o)
    e_gpu = _rel_errs(fp.mw, fp.Lw.toarray(), lp, mw_o, A_o, lp_o)
    assert e_gpu[0] <= 4 * yard[0] and e_gpu[1] <= 4 * yard[1], (e_gpu, yard)
    assert B.logpdf(fx, y) == lp  # deterministic split-K reduction


# ---- BASELINE configs 3 and 5 at their STATED size -------------------------------------------------------------------
# Tolerances are not hand-picked: the same fp32 inputs also go through LAPACK/BLAS in fp32 on the host (the reference's own
# algorithm in Float32 -- the literal sequence of :72-89 -- and the direct Gram form), both are measured against the fp64
# oracle on the fp32-rounded inputs, and the GPU may be at most 4x worse than the worse of the two.
def _rel_errs(mw, A, lp, mw_o, A_o, lp_o):
    """(posterior mean: relative 2-norm; posterior precision: max abs over max abs; log evidence: relative)"""
    return (float(np.linalg.norm(np.asarray(mw, float) - mw_o) / np.linalg.norm(mw_o)),
            float(np.max(np.abs(np.asarray(A, float) - A_o)) / np.max(np.abs(A_o))),
            abs(float(lp) - lp_o) / abs(lp_o))


def _fp32_lapack_yardstick(mw32, d32, X32, s32, y32, mw_o, A_o, lp_o):
    """errors of fp32 LAPACK on the same inputs: max over the literal sequence and the direct form"""
    m_d, T_d, A_d, lp_d = O.posterior_logpdf_direct(mw32, d32, X32, s32, y32)  # dtype follows X: sgemm / spotrf / strtrs
    e_direct = _rel_errs(m_d, A_d, lp_d, mw_o, A_o, lp_o)
    m_l, T_l, A_l = O.posterior_literal(mw32, d32, X32, s32, y32)
    lp_l = O.logpdf_literal(mw32, d32, X32, s32, y32)
    e_lit = _rel_errs(m_l, A_l, lp_l, mw_o, A_o, lp_o)
    assert m_d.dtype == np.float32 and A_l.dtype == np.float32
    return tuple(max(a, b) for a, b in zip(e_direct, e_lit)), e_direct, e_lit


def _assert_fp32_within_lapack(mw32, Lw32, X32, s32, y32, got_mw, got_A, got_lp, got_T=None, floor=(2e-6, 5e-7, 2e-7),
                               what=""):
    """fp32 results against the fp64 oracle run on the SAME fp32-rounded inputs; bound = 4x the error fp32 LAPACK makes on
    them with the reference's own op sequence (:72-89) -- nothing hand-picked but the floor of a few fp32 ulps that covers
    problems LAPACK happens to solve exactly (N = 0, N = 1).  Lw32: diagonal (1-D) or dense (2-D) prior precision; s32:
    scalar / vector / dense matrix, as the oracle takes them.  Returns (gpu errors, yardstick)."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    for a32 in (mw32, Lw32, X32, y32):
        assert np.asarray(a32).dtype == np.float32
    mw_o, T_o, A_o = O.posterior_literal(f64(mw32), f64(Lw32), f64(X32), f64(s32), f64(y32))
    lp_o = O.logpdf_literal(f64(mw32), f64(Lw32), f64(X32), f64(s32), f64(y32))
    m_l, _, A_l = O.posterior_literal(mw32, Lw32, X32, np.asarray(s32, dtype=np.float32), y32)
    lp_l = O.logpdf_literal(mw32, Lw32, X32, np.asarray(s32, dtype=np.float32), y32)
    assert m_l.dtype == np.float32 and A_l.dtype == np.float32
    lp_ref = lp_o if abs(lp_o) >= 1.0 else float(np.copysign(1.0, lp_o))  # an evidence near 0 (no data) is judged absolutely
    errs = lambda m, A, lp: _rel_errs(m, A, lp_ref + (float(lp) - lp_o), mw_o, A_o, lp_ref)
    yard = errs(m_l, A_l, lp_l)
    if np.ndim(s32) < 2:  # the one-pass Gram form in fp32 LAPACK as well (diagonal / isotropic noise only)
        m_d, _, A_d, lp_d = O.posterior_logpdf_direct(mw32, Lw32, X32, np.asarray(s32, dtype=np.float32), y32)
        yard = tuple(max(a, b) for a, b in zip(yard, errs(m_d, A_d, lp_d)))
    e = errs(got_mw, got_A, got_lp)
    # The evidence is a DIFFERENCE of large terms (reference :57-58: the quadratic form delta' Sy^-1 delta against |v|^2): fp32
    # cannot deliver it to better than a few ulps of the largest term, whatever the algorithm, and where fp32 LAPACK happens
    # to land inside one ulp of that term its error is luck, not a yardstick -- so the evidence bound never drops below
    # 4 eps32 x (quadratic form + N log 2pi + |logdet A| + |logdet Lw|), relative to the evidence.
    dy = f64(y32) - f64(X32).T @ f64(mw32)
    if np.ndim(s32) < 2:
        quad = float(np.sum(dy * dy / np.broadcast_to(f64(s32), dy.shape))) if dy.size else 0.0
    else:
        quad = float(dy @ np.linalg.solve(f64(s32), dy))
    Lw64 = f64(Lw32)
    ld_prior = float(np.sum(np.log(Lw64))) if Lw64.ndim == 1 else float(np.linalg.slogdet(Lw64)[1])
    ld_post = float(np.linalg.slogdet(A_o)[1])  # the other two large terms of the sum: logdet of the posterior / prior precision
    cancel = 4 * float(np.finfo(np.float32).eps) * (quad + dy.size * np.log(2 * np.pi) + abs(ld_prior) + abs(ld_post)) / abs(lp_ref)
    floor = (floor[0], floor[1], max(floor[2], cancel))
    for i, name in enumerate(("posterior mean", "posterior precision", "log evidence")):
        assert e[i] <= 4 * yard[i] + floor[i], (what, name, e, yard, floor)
    if got_T is not None:
        Tn = np.triu(np.asarray(got_T, dtype=np.float64))
        eT = float(np.max(np.abs(Tn.T @ Tn - A_o)) / np.max(np.abs(A_o)))
        assert eT <= 4 * yard[1] + floor[1], (what, "T'T", eT, yard)
        assert np.all(np.tril(np.asarray(got_T), -1) == 0)
    return e, yard


@pytest.mark.timeout(900)
def test_c3_full_size(B):
    # BASELINE config 3: D=1024, N=65536, ColVecs, diagonal noise, fp32 (reference :72-89; the fp32 accumulation over 65 k terms)
    rng = _rng(6100)
    D, N = 1024, 65536
    X = rng.standard_normal((D, N), dtype=np.float32)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(np.float32)
    w = (rng.standard_normal(D) / np.sqrt(D)).astype(np.float32)
    y = (X.T @ w + np.sqrt(s) * rng.standard_normal(N).astype(np.float32)).astype(np.float32)
    mw = (0.05 * rng.standard_normal(D)).astype(np.float32)
    dvec = np.exp(0.2 * rng.standard_normal(D)).astype(np.float32)
    X = np.asfortranarray(X)
    X64 = X.astype(np.float64)
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw.astype(float), dvec.astype(float), X64, s.astype(float), y.astype(float))
    del X64
    yard, e_direct, e_lit = _fp32_lapack_yardstick(mw, dvec, X, s, y, mw_o, A_o, lp_o)
    f = B.BayesianLinearRegressor(mw, B.Diagonal(dvec))
    fx = f(B.ColVecs(X), B.Diagonal(s))
    lp = B.logpdf(fx, y)
    fp = B.posterior(fx, y)
    assert fp.mw.dtype == np.float32
    e_gpu = _rel_errs(fp.mw, fp.Lw.toarray(), lp, mw_o, A_o, lp_o)
    print(f"c3 full size: rel err (mw', A, logpdf)  GPU {e_gpu}  fp32 LAPACK direct {e_direct}  literal {e_lit}")
    assert e_gpu[2] <= 2e-4                        # BASELINE tolerance on the evidence
    assert e_gpu[2] <= 4 * yard[2] + 1e-7, (e_gpu, yard)
    assert e_gpu[0] <= 4 * yard[0], (e_gpu, yard)  # posterior mean
    assert e_gpu[1] <= 4 * yard[1], (e_gpu, yard)  # posterior precision
    assert B.logpdf(fx, y) == lp                   # deterministic split-K reduction at full size


@pytest.mark.timeout(900)
def test_c5_full_size(B):
    # BASELINE config 5: random-Fourier basis D_in = 8 -> D = 2048 features, N = 16384, fp32, through blr_posterior_rff_f32
    rng = _rng(6200)
    Din, D, N = 8, 2048, 16384
    Xin = np.asfortranarray(rng.standard_normal((Din, N), dtype=np.float32))
    Om = np.asfortranarray(rng.standard_normal((Din, D), dtype=np.float32))
    beta = (2 * np.pi * rng.random(D)).astype(np.float32)
    scale = np.sqrt(2.0 / D)
    Phi_o = scale * np.cos(Om.astype(float).T @ Xin.astype(float) + beta.astype(float)[:, None])  # fp64 on the fp32-rounded inputs
    s = np.exp(0.3 * rng.standard_normal(N)).astype(np.float32)
    y = (Phi_o.T @ rng.standard_normal(D) + np.sqrt(s.astype(float)) * rng.standard_normal(N)).astype(np.float32)
    mw = (0.1 * rng.standard_normal(D)).astype(np.float32)
    dvec = np.ones(D, dtype=np.float32)
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw.astype(float), dvec.astype(float), Phi_o, s.astype(float), y.astype(float))
    # the same pipeline in fp32 on the host: features in fp32, then fp32 LAPACK
    Phi32 = np.asfortranarray((np.float32(scale) * np.cos(Om.T @ Xin + beta[:, None])).astype(np.float32))
    yard, e_direct, e_lit = _fp32_lapack_yardstick(mw, dvec, Phi32, s, y, mw_o, A_o, lp_o)
    del Phi_o
    rff = B.RandomFourierFeatures(Om, beta)
    bfr = B.BasisFunctionRegressor(B.BayesianLinearRegressor(mw, B.Diagonal(dvec)), rff)
    fx = bfr(B.ColVecs(Xin), B.Diagonal(s))
    lp = B.logpdf(fx, y)
    post = B.posterior(fx, y)
    assert isinstance(post, B.BasisFunctionRegressor) and post.phi is rff and post.blr.mw.dtype == np.float32
    e_gpu = _rel_errs(post.blr.mw, post.blr.Lw.toarray(), lp, mw_o, A_o, lp_o)
    print(f"c5 full size: rel err (mw', A, logpdf)  GPU {e_gpu}  fp32 LAPACK direct {e_direct}  literal {e_lit}")
    assert e_gpu[2] <= 2e-4
    assert e_gpu[2] <= 4 * yard[2] + 1e-7, (e_gpu, yard)
    assert e_gpu[0] <= 4 * yard[0], (e_gpu, yard)
    assert e_gpu[1] <= 4 * yard[1], (e_gpu, yard)


def test_large_d_not_spd(B):
    rng = _rng(6002)
    D, N = 200, 64
    X = rng.standard_normal((D, N))
    Lw = np.eye(D)
    Lw[150, 150] = -1e6
    f = B.BayesianLinearRegressor(np.zeros(D), Lw)
    with pytest.raises(B.PosDefException) as ei:
        B.logpdf(f(X, 0.5), np.zeros(N))
    assert ei.value.info == 151
    with pytest.raises(B.PosDefException) as ei:
        B.posterior(B.BayesianLinearRegressor(np.zeros(D), B.Diagonal(-np.ones(D)))(X, 0.5), np.zeros(N))
    assert ei.value.info == 1


@pytest.mark.parametrize("dtype,D,N", [(np.float64, 96, 300), (np.float64, 200, 500), (np.float32, 512, 2048)])
def test_rff_basis_config5_family(B, dtype, D, N):
    # BASELINE config 5 family: D_in = 8 -> D random-Fourier features, fused feature map + Gram.
    # Parity: materialise Phi on the CPU, then run the plain path.
    rng = _rng(7000 + D)
    Din = 8
    Xin = rng.standard_normal((Din, N)).astype(dtype)
    Om = rng.standard_normal((Din, D)).astype(dtype)
    beta = (2 * np.pi * rng.random(D)).astype(dtype)
    scale = np.sqrt(2.0 / D)
    Phi_ref = scale * np.cos(Om.astype(float).T @ Xin.astype(float) + beta.astype(float)[:, None])
    s = np.exp(0.3 * rng.standard_normal(N)).astype(dtype)
    y = (Phi_ref.T @ rng.standard_normal(D) + np.sqrt(s) * rng.standard_normal(N)).astype(dtype)
    mw = (0.1 * rng.standard_normal(D)).astype(dtype)
    dvec = np.ones(D, dtype=dtype)
    rff = B.RandomFourierFeatures(Om, beta)
    tol = 1e-12 if dtype == np.float64 else 2e-6
    Phi = rff(B.ColVecs(np.asfortranarray(Xin))).X
    assert Phi.dtype == dtype
    np.testing.assert_allclose(Phi, Phi_ref, rtol=0, atol=tol)
    np.testing.assert_allclose(rff(B.RowVecs(np.ascontiguousarray(Xin.T))).X, Phi_ref.T, rtol=0, atol=tol)
    blr = B.BayesianLinearRegressor(mw, B.Diagonal(dvec))
    bfr = B.BasisFunctionRegressor(blr, rff)
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw.astype(float), dvec.astype(float), Phi_ref, s.astype(float),
                                                     y.astype(float))
    lp = B.logpdf(bfr(B.ColVecs(np.asfortranarray(Xin)), s), y)
    post = B.posterior(bfr(B.ColVecs(np.asfortranarray(Xin)), s), y)
    assert isinstance(post, B.BasisFunctionRegressor) and post.phi is rff
    if dtype == np.float64:
        assert lp == pytest.approx(lp_o, rel=1e-10)
        # BFR == BLR o phi (reference test/basis_function_regression.jl:13-28)
        assert lp == pytest.approx(B.logpdf(blr(B.ColVecs(Phi), s), y), rel=1e-12)
        np.testing.assert_allclose(post.blr.mw, mw_o, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(post.blr.Lw.toarray(), A_o, rtol=1e-9, atol=1e-9)
    else:
        # the device feature map against fp32 LAPACK on the device's OWN features (so the yardstick sees the same inputs), and
        # the features themselves against the fp64 map above (atol 2e-6, asserted earlier)
        lp_phi = B.logpdf(blr(B.ColVecs(Phi), s), y)
        assert lp == lp_phi  # BFR == BLR o phi: the same kernels on the same features, bit for bit
        _assert_fp32_within_lapack(mw, dvec, np.asfortranarray(Phi), s, y, post.blr.mw, post.blr.Lw.toarray(), lp,
                                   what="rff family")


# ---- edge cases: empty and ragged inputs, padded leading dimensions, shared inputs ---------------------------------
def test_empty_and_tiny_inputs(B):
    from blr_amd import _abi

    rng = _rng(8000)
    h = _abi.default_handle()
    # N = 0: the posterior is the prior, the evidence of no data is 0 (log 1)
    for D in (1, 5, 130):
        mw = rng.standard_normal(D)
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        Lw = Bm @ Bm.T + np.eye(D)
        f = B.BayesianLinearRegressor(mw, Lw)
        X0 = np.zeros((D, 0), order="F")
        assert B.logpdf(f(X0, 0.3), np.zeros(0)) == pytest.approx(0.0, abs=1e-10)
        fp = B.posterior(f(X0, 0.3), np.zeros(0))
        np.testing.assert_allclose(fp.mw, mw, rtol=1e-13)
        np.testing.assert_allclose(fp.Lw.toarray(), Lw, rtol=1e-12)
        assert B.mean(f(X0, 0.3)).shape == (0,)
    # D = 1, N = 1
    f = B.BayesianLinearRegressor(np.array([0.5]), B.Diagonal(np.array([2.0])))
    X = np.array([[3.0]])
    lp = B.logpdf(f(X, 0.25), np.array([1.0]))
    var = 9.0 / 2.0 + 0.25
    assert lp == pytest.approx(-0.5 * (np.log(2 * np.pi) + np.log(var) + (1.0 - 1.5) ** 2 / var), rel=1e-13)
    # B = 0 is a no-op
    info = np.zeros(1, dtype=np.int32)
    assert h.posterior_batched(np.float64, _abi.MEM_HOST, _abi.LAYOUT_COLVECS, 0, 4, 8, None, 4, 32, None, 8, 0, None, 0, 2,
                               None, 4, None, 1, 4, None, 4, None, 4, 16, None, 4, 16, None, info) == 0


@pytest.mark.parametrize("N", [1, 3, 31, 32, 33, 63, 65, 127, 129, 1000])
def test_ragged_column_counts(B, N):
    # column counts around the stage size (32 / 64 columns) and the k-step size (4): tails are zero-filled
    rng = _rng(8100 + N)
    for D in (7, 64, 128):
        X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
        y = rng.standard_normal(N)
        f = B.BayesianLinearRegressor(mw, Lw)
        assert B.logpdf(f(np.asfortranarray(X), s), y) == pytest.approx(O.logpdf_literal(mw, Lw, X, s, y), rel=1e-10)


def test_padded_leading_dimensions_and_shared_inputs(B):
    from blr_amd import _abi

    rng = _rng(8200)
    h = _abi.default_handle()
    Bn, D, N = 5, 24, 77
    ldx, ldt, ldlp, ldl = D + 3, D + 2, D + 5, D + 1
    Xbuf = np.full((Bn, N, ldx), np.nan)  # rows beyond D are never read
    Xs = rng.standard_normal((Bn, D, N))
    for b in range(Bn):
        Xbuf[b, :, :D] = Xs[b].T
    y = rng.standard_normal((Bn, N))
    s = np.exp(rng.standard_normal(N))  # shared across the batch (stride 0)
    mw = rng.standard_normal(D)         # shared
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = Bm @ Bm.T + np.eye(D)
    Lwbuf = np.full((D, ldl), np.nan)
    Lwbuf[:, :D] = np.triu(Lw).T  # column-major with ldl: only the UPPER triangle is provided
    Lwbuf[:, :D][np.tril_indices(D, -1)[::-1]] = np.nan  # strictly-lower entries must never be read
    sentinel = -777.0
    T = np.full((Bn, D, ldt), sentinel)
    A = np.full((Bn, D, ldlp), sentinel)
    mwp = np.full((Bn, D), sentinel)
    lp = np.zeros(Bn)
    info = np.full(Bn, -1, dtype=np.int32)
    h.posterior_batched(np.float64, _abi.MEM_HOST, _abi.LAYOUT_COLVECS, Bn, D, N, Xbuf, ldx, N * ldx, y, N,
                        _abi.NOISE_DIAGONAL, s, 0, _abi.PRIOR_DENSE, mw, 0, Lwbuf, ldl, 0, mwp, D, T, ldt, D * ldt, A, ldlp,
                        D * ldlp, lp, info)
    assert np.all(info == 0)
    for b in range(Bn):
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw, Lw, Xs[b], s, y[b])
        assert lp[b] == pytest.approx(lp_o, rel=1e-11)
        np.testing.assert_allclose(mwp[b], mw_o, rtol=1e-9)
        np.testing.assert_allclose(T[b, :, :D].T, T_o, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(A[b, :, :D].T, A_o, rtol=1e-11)
        assert np.all(T[b, :, D:] == sentinel) and np.all(A[b, :, D:] == sentinel)  # gaps keep the caller's bytes
    # shared X, different targets: logpdf(fx, Y::Matrix) path (stride 0 on X)
    f = B.BayesianLinearRegressor(mw, Lw)
    Y = rng.standard_normal((N, 4))
    lps = B.logpdf(f(np.asfortranarray(Xs[0]), s), Y)
    for j in range(4):
        assert lps[j] == pytest.approx(O.logpdf_literal(mw, Lw, Xs[0], s, Y[:, j]), rel=1e-10)


@pytest.mark.parametrize("dtype,D,N", [(np.float64, 130, 77), (np.float64, 300, 1000), (np.float32, 1024, 3000), (np.float64, 256, 500),
                                       (np.float64, 400, 1000), (np.float64, 576, 95), (np.float32, 144, 100), (np.float32, 1040, 333),
                                       (np.float32, 1152, 31), (np.float32, 256, 8200)])
def test_large_d_marginals(B, dtype, D, N):
    # reference :33, :40-43 at D > 128: GEMV stream for the mean; var by block forward substitution on LDS-resident tiles of inputs
    # (16 | D, tile fits LDS: aligned ColVecs) or by the tall TRSM through the panel machinery of the factorisation (everything else)
    rng = _rng(9000 + D)
    X = rng.standard_normal((D, N)).astype(dtype)
    mw = rng.standard_normal(D).astype(dtype)
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = (Bm @ Bm.T + np.eye(D)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(dtype)
    m_o = O.mean(mw.astype(float), X.astype(float))
    v_o = O.var(mw.astype(float), Lw.astype(float), X.astype(float), s.astype(float))
    rt = 1e-10 if dtype == np.float64 else 3e-4
    U = O.chol_upper(Lw.astype(float)).astype(dtype)
    for Lw_arg in (Lw, B.PDMat(U)):
        f = B.BayesianLinearRegressor(mw, Lw_arg)
        for x in (np.asfortranarray(X), B.RowVecs(np.asfortranarray(X.T))):
            m, v = B.mean_and_var(f(x, s))
            np.testing.assert_allclose(m, m_o, rtol=rt, atol=rt * 10)
            np.testing.assert_allclose(v, v_o, rtol=rt)
            np.testing.assert_allclose(B.mean(f(x, s)), m_o, rtol=rt, atol=rt * 10)  # mean-only: the pure GEMV stream
    dvec = np.exp(rng.standard_normal(D)).astype(dtype)
    v = B.var(B.BayesianLinearRegressor(mw, B.Diagonal(dvec))(np.asfortranarray(X), dtype(0.5)))
    np.testing.assert_allclose(v, O.var(mw.astype(float), np.diag(dvec.astype(float)), X.astype(float), 0.5), rtol=rt)
    with pytest.raises(B.PosDefException):
        B.var(B.BayesianLinearRegressor(mw, -Lw)(np.asfortranarray(X), s))


def test_release_workspace_and_options_round_trip(B, opt):
    # blr_release_workspace frees the handle's device scratch (D > 128 workspace, marginal images, multi-output temporaries);
    # the next calls allocate it again and return the same bits.  blr_set_option rejects unknown keys / malformed values.
    a = B._abi
    h = a.default_handle()
    rng = _rng(9400)
    D, N = 256, 300
    X = np.asfortranarray(rng.standard_normal((D, N)))
    y = rng.standard_normal(N)
    mw = rng.standard_normal(D)
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = Bm @ Bm.T + np.eye(D)
    f = B.BayesianLinearRegressor(mw, Lw)

    def everything():
        fx = f(X, 0.3)
        fp = B.posterior(fx, y)
        m, v = B.mean_and_var(fp(X, 0.3))
        lps = B.logpdf(fx, np.asfortranarray(np.stack([y, 2 * y], axis=1)))
        return B.logpdf(fx, y), fp.mw.copy(), m, v, np.asarray(lps)

    first = everything()
    h.release_workspace()
    h.release_workspace()  # (nothing left to free: still fine)
    again = everything()
    for u, v in zip(first, again):
        np.testing.assert_array_equal(np.asarray(u), np.asarray(v))
    for key, value in (("NO_SUCH_SWITCH", "1"), ("WAVE_SPLIT", "3"), ("SWEEP", "sometimes"), ("GRAM_SPLITS", "7")):
        with pytest.raises(Exception):
            h.set_option(key, value)
    opt("WAVE_SPLIT", "2")  # a well-formed one round-trips (restored by the fixture)


@pytest.mark.parametrize("dtype,D,N,Bn,kind", [(np.float64, 256, 200, 5, "factor"), (np.float32, 400, 333, 3, "dense"), (np.float64, 272, 70, 4, "dense"),
                                                (np.float32, 1024, 100, 9, "factor")])
def test_large_d_marginals_batched_share_the_launches(B, dtype, D, N, Bn, kind):
    # a batch at D > 128 goes through ONE set of launches (blockIdx.y = regressor): every regressor against the oracle, a regressor
    # whose prior is not positive definite stops alone (LAPACK-style index in ITS info word), the others are untouched by it
    from blr_amd import _abi

    rng = _rng(9300 + D + Bn)
    X = rng.standard_normal((Bn, N, D)).astype(dtype)
    mw = rng.standard_normal((Bn, D)).astype(dtype)
    Lw = np.empty((Bn, D, D), dtype=dtype)
    for b in range(Bn):
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        Lw[b] = (Bm @ Bm.T + np.eye(D)).astype(dtype)
    bad = 1
    arg = np.empty((Bn, D, D), dtype=dtype)
    for b in range(Bn):
        if kind == "factor":
            U = O.chol_upper(Lw[b].astype(float)).astype(dtype)
            if b == bad:
                U[7, 7] = -U[7, 7]
            arg[b] = U.T  # (column-major storage of U)
        else:
            A = Lw[b].copy()
            if b == bad:
                A[7, 7] = -50.0
            arg[b] = A.T
    s = np.exp(0.3 * rng.standard_normal((Bn, N))).astype(dtype)
    mean = np.full((Bn, N), -7.0, dtype=dtype)
    var = np.full((Bn, N), -7.0, dtype=dtype)
    info = np.full(Bn, 99, dtype=np.int32)
    h = _abi.default_handle()
    pk = _abi.PRIOR_UPPER_FACTOR if kind == "factor" else _abi.PRIOR_DENSE
    h.marginals_batched(dtype, _abi.MEM_HOST, _abi.LAYOUT_COLVECS, Bn, D, N, X, D, N * D, _abi.NOISE_DIAGONAL, s, N, pk, mw, D, arg, D, D * D, mean, N, var, N,
                        info)
    rt = 1e-10 if dtype == np.float64 else 3e-4
    for b in range(Bn):
        if b == bad:
            assert info[b] == 8
            continue
        assert info[b] == 0
        m_o = O.mean(mw[b].astype(float), X[b].T.astype(float))
        v_o = O.var(mw[b].astype(float), Lw[b].astype(float), X[b].T.astype(float), s[b].astype(float))
        np.testing.assert_allclose(mean[b], m_o, rtol=rt, atol=rt * 10)
        np.testing.assert_allclose(var[b], v_o, rtol=rt)


@pytest.mark.parametrize("dtype,D,N,S", [(np.float64, 130, 40, 3), (np.float64, 384, 257, 70), (np.float32, 1024, 500, 5)])
def test_large_d_rand_and_weight_draws(B, dtype, D, N, S):
    # reference :46-53 at D > 128: blocked factorisation of the prior + wavefront back substitution, one grid column per draw
    from blr_amd import _abi

    rng = _rng(9100 + D)
    X = rng.standard_normal((D, N)).astype(dtype)
    mw = rng.standard_normal(D).astype(dtype)
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = (Bm @ Bm.T + np.eye(D)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(dtype)
    Z1 = np.asfortranarray(rng.standard_normal((D, S)).astype(dtype))
    Z2 = np.asfortranarray(rng.standard_normal((N, S)).astype(dtype))
    f64 = lambda a: np.asarray(a, dtype=float)
    rt = 1e-10 if dtype == np.float64 else 2e-4
    W_o = O.sample_weights(f64(mw), f64(Lw), f64(Z1))
    Y_o = O.rand(f64(mw), f64(Lw), f64(X), f64(s), f64(Z1), f64(Z2))
    h = _abi.default_handle()
    U = np.asfortranarray(O.chol_upper(f64(Lw)).astype(dtype))
    for kind, Larg in ((_abi.PRIOR_DENSE, np.asfortranarray(Lw)), (_abi.PRIOR_UPPER_FACTOR, U)):
        W = np.empty((D, S), dtype=dtype, order="F")
        h.sample_weights(dtype, _abi.MEM_HOST, D, S, kind, mw, Larg, D, Z1, D, W, D)
        np.testing.assert_allclose(W, W_o, rtol=rt, atol=rt * 10)
        for layout, Xa, ldx in ((_abi.LAYOUT_COLVECS, np.asfortranarray(X), D), (_abi.LAYOUT_ROWVECS, np.ascontiguousarray(X), N)):
            Y = np.empty((N, S), dtype=dtype, order="F")
            h.rand(dtype, _abi.MEM_HOST, layout, D, N, S, Xa, ldx, _abi.NOISE_DIAGONAL, s, kind, mw, Larg, D, Z1, D, Z2, N, Y, N)
            np.testing.assert_allclose(Y, Y_o, rtol=rt * 10, atol=rt * 100)
    dvec = np.exp(rng.standard_normal(D)).astype(dtype)
    W = np.empty((D, S), dtype=dtype, order="F")
    h.sample_weights(dtype, _abi.MEM_HOST, D, S, _abi.PRIOR_DIAGONAL, mw, dvec, D, Z1, D, W, D)
    np.testing.assert_allclose(W, f64(mw)[:, None] + f64(Z1) / np.sqrt(f64(dvec))[:, None], rtol=rt)
    # the host mirror draws through the same path; a non-SPD prior raises like cholesky() in the reference (:50)
    g = B.rand(np.random.default_rng(1), B.BayesianLinearRegressor(mw, Lw))
    assert g.w.shape == (D,)
    with pytest.raises(B.PosDefException):
        B.rand(np.random.default_rng(1), B.BayesianLinearRegressor(mw, -Lw))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dtype,D,S", [(np.float64, 300, 2500), (np.float32, 1100, 1500)])
def test_large_d_many_draws_wavefront_oversubscribed(B, dtype, D, S):
    # more wavefront workgroups than the chip can hold at once (blocks x draws >> 512 slots, several launches):
    # the start-order tickets must keep every waiting workgroup's dependencies running
    from blr_amd import _abi

    rng = _rng(9300 + D)
    mw = rng.standard_normal(D).astype(dtype)
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = (Bm @ Bm.T + np.eye(D)).astype(dtype)
    U = np.asfortranarray(O.chol_upper(np.asarray(Lw, dtype=float)).astype(dtype))
    Z = np.asfortranarray(rng.standard_normal((D, S)).astype(dtype))
    W = np.empty((D, S), dtype=dtype, order="F")
    h = _abi.default_handle()
    for _ in range(3):  # repeated launches reuse the exchange buffer with fresh epochs
        W[:] = 0
        h.sample_weights(dtype, _abi.MEM_HOST, D, S, _abi.PRIOR_UPPER_FACTOR, mw, U, D, Z, D, W, D)
        W_o = np.asarray(mw, dtype=float)[:, None] + np.linalg.solve(np.asarray(U, dtype=float), np.asarray(Z, dtype=float))
        rt = 1e-10 if dtype == np.float64 else 2e-4
        np.testing.assert_allclose(W, W_o, rtol=rt, atol=rt * 10)


# ---- gradient of the log marginal likelihood: closed form vs the oracle (itself pinned against
# ---- finite differences of the literal op sequence in tests/test_oracle_pins.py) ------------------------------------
@pytest.mark.parametrize("N,D", [(11, 3), (200, 17), (333, 64), (777, 128), (64, 128)])
@pytest.mark.parametrize("noise", ["diagonal", "isotropic"])
def test_logpdf_gradient_vs_oracle_f64(B, N, D, noise):
    rng = _rng(9500 + N + D)
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    y = rng.standard_normal(N)
    Sy = s if noise == "diagonal" else np.float64(0.37)
    lp_o, g_o = O.logpdf_grad(mw, Lw, X, Sy, y)
    U = O.chol_upper(Lw)
    dvec = np.exp(0.3 * rng.standard_normal(D))
    for prior, Lw_arg, Lw_ref in (("dense", Lw, Lw), ("pdmat", B.PDMat(U), Lw), ("diag", B.Diagonal(dvec), np.diag(dvec))):
        lp_r, g_r = (lp_o, g_o) if prior != "diag" else O.logpdf_grad(mw, Lw_ref, X, Sy, y)
        f = B.BayesianLinearRegressor(mw, Lw_arg)
        for x in (np.asfortranarray(X), B.ColVecs(np.ascontiguousarray(X)), B.RowVecs(np.ascontiguousarray(X.T)),
                  B.RowVecs(np.asfortranarray(X.T))):
            lp, g = B.logpdf_and_gradient(f(x, Sy if noise == "diagonal" else float(Sy)), y)
            assert lp == pytest.approx(lp_r, rel=1e-10)
            gX = g["X"].T if isinstance(x, B.RowVecs) else g["X"]
            scale = np.abs(g_r["X"]).max()
            np.testing.assert_allclose(gX, g_r["X"], rtol=1e-8, atol=1e-9 * scale)
            np.testing.assert_allclose(g["y"], g_r["y"], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(g["mw"], g_r["mw"], rtol=1e-8, atol=1e-9 * np.abs(g_r["mw"]).max())
            if noise == "diagonal":
                np.testing.assert_allclose(g["noise"], g_r["s"], rtol=1e-8, atol=1e-9 * np.abs(g_r["s"]).max())
            else:
                assert float(g["noise"]) == pytest.approx(float(np.sum(g_r["s"])), rel=1e-8)
            gL_ref = np.diag(g_r["Lw"]) if prior == "diag" else g_r["Lw"]
            np.testing.assert_allclose(g["Lw"], gL_ref, rtol=1e-7, atol=1e-9 * np.abs(gL_ref).max())


def test_logpdf_gradient_batched_device_f32_and_errors(B):
    import torch
    from blr_amd import _abi

    rng = _rng(9600)
    Bn, D, N = 5, 96, 700
    dev = torch.device("cuda:0")
    X = torch.tensor(rng.standard_normal((Bn, N, D)), dtype=torch.float32, device=dev)  # [N, D] row-major == D x N col-major
    y = torch.tensor(rng.standard_normal((Bn, N)), dtype=torch.float32, device=dev)
    s = torch.tensor(np.exp(0.3 * rng.standard_normal((Bn, N))), dtype=torch.float32, device=dev)
    mw = torch.tensor(rng.standard_normal((Bn, D)), dtype=torch.float32, device=dev)
    d = torch.tensor(np.exp(0.2 * rng.standard_normal((Bn, D))), dtype=torch.float32, device=dev)
    lp = torch.zeros(Bn, dtype=torch.float64, device=dev)
    info = torch.zeros(Bn, dtype=torch.int32, device=dev)
    dX = torch.empty_like(X); dy = torch.empty_like(y); ds = torch.empty_like(s); dmw = torch.empty_like(mw)
    mwp = torch.empty_like(mw); Ai = torch.empty((Bn, D, D), dtype=torch.float32, device=dev)
    h = _abi.default_handle()
    h.logpdf_grad_batched(np.float32, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, Bn, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N,
                          _abi.NOISE_DIAGONAL, s.data_ptr(), N, _abi.PRIOR_DIAGONAL, mw.data_ptr(), D, d.data_ptr(), 1, D,
                          lp.data_ptr(), dX.data_ptr(), D, N * D, dy.data_ptr(), N, ds.data_ptr(), N, dmw.data_ptr(), D,
                          mwp.data_ptr(), D, Ai.data_ptr(), D, D * D, info.data_ptr())
    h.synchronize()
    assert int(info.abs().sum()) == 0
    for b in range(Bn):
        Xb = X[b].double().cpu().numpy().T
        args = (mw[b].double().cpu().numpy(), np.diag(d[b].double().cpu().numpy()), Xb, s[b].double().cpu().numpy(),
                y[b].double().cpu().numpy())
        lp_o, g_o = O.logpdf_grad(*args)
        assert float(lp[b]) == pytest.approx(lp_o, rel=2e-4)
        np.testing.assert_allclose(dX[b].cpu().numpy().T, g_o["X"], rtol=2e-3, atol=2e-4 * np.abs(g_o["X"]).max())
        np.testing.assert_allclose(dy[b].cpu().numpy(), g_o["y"], rtol=2e-3, atol=1e-4)
        np.testing.assert_allclose(ds[b].cpu().numpy(), g_o["s"], rtol=2e-3, atol=2e-4 * np.abs(g_o["s"]).max())
        np.testing.assert_allclose(dmw[b].cpu().numpy(), g_o["mw"], rtol=2e-3, atol=2e-4 * np.abs(g_o["mw"]).max())
        np.testing.assert_allclose(Ai[b].cpu().numpy(), g_o["Ainv"], rtol=2e-3, atol=2e-4 * np.abs(g_o["Ainv"]).max())
        np.testing.assert_allclose(mwp[b].cpu().numpy(), g_o["mw_post"], rtol=2e-3, atol=1e-4)
    # a non-SPD prior raises like cholesky() in the reference; D > 8192 is rejected with the argument position
    Xs, mws, Lws, ss = O.generate_toy_problem(rng, 20, 4, dense_noise_cov=False)
    with pytest.raises(B.PosDefException):
        B.logpdf_and_gradient(B.BayesianLinearRegressor(mws, -Lws)(Xs, ss), rng.standard_normal(20))
    with pytest.raises(B.BLRError):
        B.logpdf_and_gradient(B.BayesianLinearRegressor(np.zeros(9000), B.Diagonal(np.ones(9000)))(np.zeros((9000, 8)), 0.1), np.zeros(8))


@pytest.mark.parametrize("dtype,D,N", [(np.float64, 130, 77), (np.float64, 300, 500), (np.float32, 1024, 700)])
def test_large_d_logpdf_gradient(B, dtype, D, N):
    # D > 128: forward + backward panel sweeps over the tall matrix [F; X'; I] (blr_large.hpp)
    rng = _rng(9700 + D)
    X = rng.standard_normal((D, N)).astype(dtype)
    mw = (0.3 * rng.standard_normal(D)).astype(dtype)
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = (Bm @ Bm.T + np.eye(D)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(dtype)
    y = rng.standard_normal(N).astype(dtype)
    f64 = lambda a: np.asarray(a, dtype=float)
    lp_o, g_o = O.logpdf_grad(f64(mw), f64(Lw), f64(X), f64(s), f64(y))
    rt = 1e-8 if dtype == np.float64 else 3e-3
    f = B.BayesianLinearRegressor(mw, Lw)
    for x in (np.asfortranarray(X), B.RowVecs(np.asfortranarray(X.T))):
        lp, g = B.logpdf_and_gradient(f(x, s), y)
        assert lp == pytest.approx(lp_o, rel=1e-10 if dtype == np.float64 else 3e-4)
        gX = g["X"].T if isinstance(x, B.RowVecs) else g["X"]
        for got, ref in ((gX, g_o["X"]), (g["y"], g_o["y"]), (g["noise"], g_o["s"]), (g["mw"], g_o["mw"]), (g["Lw"], g_o["Lw"])):
            np.testing.assert_allclose(got, ref, rtol=rt, atol=rt * np.abs(ref).max())
    lp, g = B.logpdf_and_gradient(f(np.asfortranarray(X), dtype(0.4)), y)  # isotropic noise: scalar gradient
    lp_i, g_i = O.logpdf_grad(f64(mw), f64(Lw), f64(X), np.float64(dtype(0.4)), f64(y))
    assert float(g["noise"]) == pytest.approx(float(np.sum(g_i["s"])), rel=rt * 10)


@pytest.mark.parametrize("dtype,N,noise", [(np.float64, 1000, "diag"), (np.float64, 64, "iso"), (np.float32, 1543, "diag"), (np.float32, 257, "iso")])
def test_logpdf_gradient_d128_product_form_vs_sweeps_and_oracle(B, opt, dtype, N, noise):
    # D = 128, aligned ColVecs: both triangular sweeps of the gradient as products with the explicit inverse factor
    # (grad_gemm_kernel) against the sweep kernel (NO_GRAD_GEMM) and the oracle: a batch with padded leading dimensions of X and dX,
    # N not a multiple of 16, a regressor whose prior is not positive definite, with and without A^-1 / dX.
    a = B._abi
    h = a.default_handle()
    rng = _rng(9850 + N)
    Bn, D, ldx = 5, 128, 128 + 4
    X = np.zeros((Bn, N, ldx), dtype=dtype); X[:, :, :D] = rng.standard_normal((Bn, N, D))
    y = rng.standard_normal((Bn, N)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal((Bn, N))).astype(dtype) if noise == "diag" else np.full((Bn, 1), 0.37, dtype=dtype)
    mw = (0.3 * rng.standard_normal((Bn, D))).astype(dtype)
    Lw = np.empty((Bn, D, D), dtype=dtype)
    for b in range(Bn):
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        Lw[b] = Bm @ Bm.T + np.eye(D)
    Lw[3] = -Lw[3]

    def run(want_dx, want_ai):
        lp = np.zeros(Bn); info = np.full(Bn, 9, dtype=np.int32)
        dX = np.full_like(X, -7.0) if want_dx else None
        dy = np.zeros_like(y); ds = np.zeros((Bn, N), dtype=dtype); dmw = np.zeros_like(mw); mwp = np.zeros_like(mw)
        Ai = np.zeros((Bn, D, D), dtype=dtype) if want_ai else None
        h.logpdf_grad_batched(dtype, a.MEM_HOST, a.LAYOUT_COLVECS, Bn, D, N, X, ldx, N * ldx, y, N,
                              a.NOISE_DIAGONAL if noise == "diag" else a.NOISE_ISOTROPIC, s, N if noise == "diag" else 1, a.PRIOR_DENSE,
                              mw, D, Lw, D, D * D, lp, dX, ldx, N * ldx, dy, N, ds, N, dmw, D, mwp, D, Ai, D, D * D, info)
        return lp, dX, dy, ds, dmw, mwp, Ai, info

    fast = run(True, True)
    lean = run(False, False)
    opt("NO_GRAD_GEMM", "1")
    slow = run(True, True)
    opt("NO_GRAD_GEMM", None)
    ok = [0, 1, 2, 4]
    assert fast[7].tolist() == slow[7].tolist() and fast[7][3] != 0 and all(fast[7][b] == 0 for b in ok)
    assert np.all(fast[1][:, :, D:] == -7.0) and np.all(fast[1][3] == -7.0)  # padding and the failed regressor untouched
    eq = 1e-9 if dtype == np.float64 else 2e-3
    for u, v in zip(fast[:7], slow[:7]):
        np.testing.assert_allclose(u[ok], v[ok], rtol=eq, atol=eq * np.abs(v[ok]).max())
    for i in (0, 2, 3, 4, 5):  # the outputs that do not depend on dX / A^-1 being asked for
        np.testing.assert_array_equal(fast[i][ok], lean[i][ok])
    rt = 1e-8 if dtype == np.float64 else 3e-3
    f64 = lambda v: np.asarray(v, dtype=float)
    for b in ok:
        sb = f64(s[b]) if noise == "diag" else float(s[b, 0])
        lp_o, g_o = O.logpdf_grad(f64(mw[b]), f64(Lw[b]), f64(X[b, :, :D]).T, sb, f64(y[b]))
        assert fast[0][b] == pytest.approx(lp_o, rel=1e-10 if dtype == np.float64 else 3e-4)
        for got, ref in ((fast[1][b][:, :D].T, g_o["X"]), (fast[2][b], g_o["y"]), (fast[3][b], g_o["s"]), (fast[4][b], g_o["mw"]),
                         (fast[5][b], g_o["mw_post"]), (fast[6][b], g_o["Ainv"])):
            np.testing.assert_allclose(got, ref, rtol=rt, atol=rt * np.abs(ref).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_d128_product_forms_rowvecs_give_the_bits_of_colvecs(B, dtype):
    # marginals_gemm_kernel / grad_gemm_kernel with RowVecs inputs (N x D column-major, padded leading dimension): the same
    # registers filled by scalar loads, the same MFMA sequence -- in fp64 the bits of the ColVecs call on the transposed copy (the
    # gradient: to rounding, its update runs on the kernel of the layout); N not a multiple of 16.
    a = B._abi
    h = a.default_handle()
    rng = _rng(9870)
    Bn, D, N = 3, 128, 1000 + 7
    ldr = N + 3
    Xc = rng.standard_normal((Bn, N, D)).astype(dtype)            # [N, D] row-major == D x N ColVecs
    Xr = np.zeros((Bn, D, ldr), dtype=dtype); Xr[:, :, :N] = Xc.transpose(0, 2, 1)  # N x D column-major, lda = N + 3
    y = rng.standard_normal((Bn, N)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal((Bn, N))).astype(dtype)
    mw = (0.3 * rng.standard_normal((Bn, D))).astype(dtype)
    U = np.empty((Bn, D, D), dtype=dtype)
    for b in range(Bn):
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        U[b] = O.chol_upper(Bm @ Bm.T + np.eye(D)).astype(dtype).T  # column-major upper factor

    def marg(layout, X, ldx, sx):
        mean = np.zeros((Bn, N), dtype=dtype); var = np.zeros((Bn, N), dtype=dtype); info = np.full(Bn, 9, dtype=np.int32)
        h.marginals_batched(dtype, a.MEM_HOST, layout, Bn, D, N, X, ldx, sx, a.NOISE_DIAGONAL, s, N, a.PRIOR_UPPER_FACTOR, mw, D, U, D, D * D,
                            mean, N, var, N, info)
        assert info.tolist() == [0] * Bn
        return mean, var

    mc, vc = marg(a.LAYOUT_COLVECS, Xc, D, N * D)
    mr, vr = marg(a.LAYOUT_ROWVECS, Xr, ldr, D * ldr)
    if dtype == np.float64:
        np.testing.assert_array_equal(vc, vr)
        np.testing.assert_array_equal(mc, mr)
    else:  # (fp32: hipcc packs pairs of multiplies / adds differently in the two instantiations: an ulp here and there)
        np.testing.assert_allclose(vr, vc, rtol=1e-6)
        np.testing.assert_allclose(mr, mc, rtol=1e-5, atol=1e-5)
    v_o = O.var(mw[0].astype(float), (U[0].T.astype(float).T @ U[0].T.astype(float)), Xc[0].T.astype(float), s[0].astype(float))
    np.testing.assert_allclose(vr[0], v_o, rtol=1e-9 if dtype == np.float64 else 5e-4)

    def grad(layout, X, ldx, sx):
        lp = np.zeros(Bn); info = np.full(Bn, 9, dtype=np.int32)
        dX = np.full_like(X, -7.0); dy = np.zeros_like(y); ds = np.zeros_like(s); dmw = np.zeros_like(mw); mwp = np.zeros_like(mw)
        Ai = np.zeros((Bn, D, D), dtype=dtype)
        h.logpdf_grad_batched(dtype, a.MEM_HOST, layout, Bn, D, N, X, ldx, sx, y, N, a.NOISE_DIAGONAL, s, N, a.PRIOR_UPPER_FACTOR,
                              mw, D, U, D, D * D, lp, dX, ldx, sx, dy, N, ds, N, dmw, D, mwp, D, Ai, D, D * D, info)
        assert info.tolist() == [0] * Bn
        return lp, dX, dy, ds, dmw, mwp, Ai

    gc = grad(a.LAYOUT_COLVECS, Xc, D, N * D)
    gr = grad(a.LAYOUT_ROWVECS, Xr, ldr, D * ldr)
    # (the update in front runs on the kernel of its layout: the posterior, hence everything after it, agrees to rounding only)
    eq = 1e-10 if dtype == np.float64 else 2e-3
    np.testing.assert_allclose(gr[1][:, :, :N].transpose(0, 2, 1), gc[1], rtol=eq, atol=eq * np.abs(gc[1]).max())
    assert np.all(gr[1][:, :, N:] == -7.0)
    for i in (0, 2, 3, 4, 5, 6):
        np.testing.assert_allclose(gr[i], gc[i], rtol=eq, atol=eq * np.abs(gc[i]).max())
    f64 = lambda v: np.asarray(v, dtype=float)
    Lw0 = f64(U[0].T).T @ f64(U[0].T)
    lp_o, g_o = O.logpdf_grad(f64(mw[0]), Lw0, f64(Xc[0]).T, f64(s[0]), f64(y[0]))
    rt = 1e-8 if dtype == np.float64 else 3e-3
    np.testing.assert_allclose(gr[1][0][:, :N], g_o["X"], rtol=rt, atol=rt * np.abs(g_o["X"]).max())
    np.testing.assert_allclose(gr[6][0], g_o["Ainv"], rtol=rt, atol=rt * np.abs(g_o["Ainv"]).max())


@pytest.mark.parametrize("Bn", [1, 3, 17])
@pytest.mark.parametrize("dtype,D,N", [(np.float64, 300, 500), (np.float32, 256, 1100)])
def test_large_d_logpdf_gradient_batched_share_the_launches(B, opt, Bn, dtype, D, N):
    # D > 128, a batch: the regressors go through the update and through the forward / backward sweeps of the tall matrix together
    # (blockIdx.y of every launch).  Each against the oracle; the grouped call agrees with one regressor at a time (CHAIN_BATCH=1) to
    # rounding; a regressor whose prior is not positive definite reports it and leaves the others alone.
    a = B._abi
    h = a.default_handle()
    rng = _rng(9750 + D + Bn)
    X = rng.standard_normal((Bn, N, D)).astype(dtype)  # [N, D] row-major == D x N column-major
    y = rng.standard_normal((Bn, N)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal((Bn, N))).astype(dtype)
    mw = (0.3 * rng.standard_normal((Bn, D))).astype(dtype)
    d = np.exp(0.2 * rng.standard_normal((Bn, D))).astype(dtype)
    bad = 1 if Bn > 1 else None
    if bad is not None:
        d[bad, 5] = -1.0

    def run(with_ainv):
        lp = np.zeros(Bn); info = np.full(Bn, 9, dtype=np.int32)
        dX = np.zeros_like(X); dy = np.zeros_like(y); ds = np.zeros_like(s); dmw = np.zeros_like(mw); mwp = np.zeros_like(mw)
        Ai = np.zeros((Bn, D, D), dtype=dtype) if with_ainv else None
        h.logpdf_grad_batched(dtype, a.MEM_HOST, a.LAYOUT_COLVECS, Bn, D, N, X, D, N * D, y, N, a.NOISE_DIAGONAL, s, N, a.PRIOR_DIAGONAL,
                              mw, D, d, 1, D, lp, dX, D, N * D, dy, N, ds, N, dmw, D, mwp, D, Ai, D, D * D, info)
        return lp, dX, dy, ds, dmw, mwp, Ai, info

    grouped = run(True)
    opt("CHAIN_BATCH", "1")
    single = run(True)
    opt("CHAIN_BATCH", None)
    no_ainv = run(False)
    ok = [b for b in range(Bn) if b != bad]
    # (not bit-equal: the split of the Gram launch over the observations is planned for the group that shares it)
    eq = 1e-11 if dtype == np.float64 else 1e-3  # (fp32: the sums over the observations are rounded differently, 2e-4 seen)
    for u, v in zip(grouped[:7], single[:7]):
        np.testing.assert_allclose(u[ok], v[ok], rtol=eq, atol=eq * np.abs(v[ok]).max())
    np.testing.assert_array_equal(grouped[7], single[7])
    for u, v in zip(grouped[:6], no_ainv[:6]):
        np.testing.assert_array_equal(u[ok], v[ok])  # same group, same plan: the identity rows do not touch the others
    lp, dX, dy, ds, dmw, mwp, Ai, info = grouped
    if bad is not None:
        assert info[bad] == 6 and np.isnan(lp[bad])
    rt = 1e-8 if dtype == np.float64 else 3e-3
    f64 = lambda v: np.asarray(v, dtype=float)
    for b in ok[:4]:
        assert info[b] == 0
        lp_o, g_o = O.logpdf_grad(f64(mw[b]), np.diag(f64(d[b])), f64(X[b]).T, f64(s[b]), f64(y[b]))
        assert lp[b] == pytest.approx(lp_o, rel=1e-10 if dtype == np.float64 else 3e-4)
        for got, ref in ((dX[b].T, g_o["X"]), (dy[b], g_o["y"]), (ds[b], g_o["s"]), (dmw[b], g_o["mw"]), (Ai[b], g_o["Ainv"]), (mwp[b], g_o["mw_post"])):
            np.testing.assert_allclose(got, ref, rtol=rt, atol=rt * np.abs(ref).max())


@pytest.mark.parametrize("dtype,D,N,S", [(np.float64, 5, 13, 4), (np.float64, 128, 900, 70), (np.float64, 300, 700, 130),
                                         (np.float32, 1024, 3000, 9)])
def test_shared_x_multi_output_logpdf(B, dtype, D, N, S):
    # AbstractGPs' logpdf(fx, Y::AbstractMatrix) (reference test/bayesian_linear_regression.jl:7-9 via TestUtils): one Gram +
    # factorisation for all columns against the per-column literal oracle
    rng = _rng(9800 + D)
    X = rng.standard_normal((D, N)).astype(dtype)
    mw = (0.5 * rng.standard_normal(D)).astype(dtype)
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = (Bm @ Bm.T + np.eye(D)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(dtype)
    Y = rng.standard_normal((N, S)).astype(dtype)
    f64 = lambda a: np.asarray(a, dtype=float)
    lp_o = np.array([O.logpdf_literal(f64(mw), f64(Lw), f64(X), f64(s), f64(Y[:, j])) for j in range(S)])
    m_o = np.stack([O.posterior_logpdf_direct(f64(mw), f64(Lw), f64(X), f64(s), f64(Y[:, j]))[0] for j in range(min(S, 6))], axis=1)
    rt = 1e-10 if dtype == np.float64 else 3e-4
    U = O.chol_upper(f64(Lw)).astype(dtype)
    for Lw_arg in (Lw, B.PDMat(U)):
        f = B.BayesianLinearRegressor(mw, Lw_arg)
        for x in (np.asfortranarray(X), B.RowVecs(np.asfortranarray(X.T))):
            lp, M = B.logpdf_columns(f(x, s), Y, return_means=True)
            np.testing.assert_allclose(lp, lp_o, rtol=rt)
            np.testing.assert_allclose(M[:, :m_o.shape[1]], m_o, rtol=rt * 100, atol=rt * 100)
            np.testing.assert_allclose(B.logpdf(f(x, s), Y), lp_o, rtol=rt)  # the matrix form of logpdf routes here
    lp_iso = B.logpdf_columns(B.BayesianLinearRegressor(mw, Lw)(np.asfortranarray(X), dtype(0.4)), Y)
    ref = [O.logpdf_literal(f64(mw), f64(Lw), f64(X), np.float64(dtype(0.4)), f64(Y[:, j])) for j in range(S)]
    np.testing.assert_allclose(lp_iso, ref, rtol=rt)
    with pytest.raises(B.PosDefException):
        B.logpdf_columns(B.BayesianLinearRegressor(mw, -Lw)(np.asfortranarray(X), s), Y)


@pytest.mark.parametrize("S,zero_mean,noise", [(64, True, "diag"), (100, False, "iso"), (3, False, "diag"), (1, True, "iso"), (128, True, "diag")])
def test_shared_x_multi_output_rides_the_update_of_column_0(B, opt, S, zero_mean, noise):
    # Round 6: at D > 128 in fp32 (ColVecs, up to 128 columns) the residuals of ALL columns of Y are one more row block of the operand
    # planes of column 0's update: b_s out of the same Gram launch, u_s = L^-1 b_s out of the rows the blocked factorisation carries
    # along (64, or all 128 when S > 64), q_s out of the planes pass -- no residual matrix, no second product over X, no panel sweep
    # (blr_planes.hpp).  Held to the route it replaces (option NO_MULTI_PLANES: residual product on the f32 matrix instruction + tall
    # panels) against the fp64 oracle, per column: evidence and posterior mean within 4 x its error + a floor; N not a multiple of 16.
    rng = _rng(4400 + S)
    D, N = 384, 5000 + 7
    X = np.asfortranarray(rng.standard_normal((D, N)).astype(np.float32))
    mw = np.zeros(D, dtype=np.float32) if zero_mean else (0.3 * rng.standard_normal(D)).astype(np.float32)
    dvec = np.exp(0.2 * rng.standard_normal(D)).astype(np.float32)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(np.float32) if noise == "diag" else np.float32(0.6)
    W = rng.standard_normal((D, S)) / np.sqrt(D)
    Y = (X.astype(float).T @ W + np.sqrt(np.asarray(s, float)).reshape(-1, 1) * rng.standard_normal((N, S))).astype(np.float32)
    f64 = lambda a: np.asarray(a, dtype=float)
    cols = sorted(set([0, min(1, S - 1), S // 2, S - 1]))
    ref = {j: O.posterior_logpdf_direct(f64(mw), f64(dvec), f64(X), f64(s), f64(Y[:, j])) for j in cols}
    lp_all = np.array([O.posterior_logpdf_direct(f64(mw), f64(dvec), f64(X), f64(s), f64(Y[:, j]))[3] for j in range(S)])
    fx = B.BayesianLinearRegressor(mw, B.Diagonal(dvec))(X, s if noise == "iso" else B.Diagonal(s))

    def errs():
        lp, M = B.logpdf_columns(fx, Y, return_means=True)
        e_lp = float(np.max(np.abs(lp - lp_all) / np.abs(lp_all)))
        e_m = max(float(np.linalg.norm(M[:, j] - ref[j][0]) / np.linalg.norm(ref[j][0])) for j in cols)
        lp2 = B.logpdf_columns(fx, Y)   # without the means: the factor is never transposed
        assert np.array_equal(lp, lp2) or float(np.max(np.abs(lp - lp2) / np.abs(lp))) < 1e-6
        return e_lp, e_m

    e_new = errs()
    opt("NO_MULTI_PLANES", "1")
    e_old = errs()
    print(f"multi-output evidence, S = {S}: max rel err (logpdf, posterior mean)  riding along {e_new}   residual product + panels {e_old}")
    assert e_new[0] <= 4 * e_old[0] + 2e-7 and e_new[1] <= 4 * e_old[1] + 2e-6, (e_new, e_old)
    assert e_new[0] <= 3e-5 and e_new[1] <= 3e-4


@pytest.mark.parametrize("dtype,D,N,noise,prior", [(np.float64, 5, 40, "diag", "dense"), (np.float64, 130, 500, "iso", "diag"),
                                                   (np.float64, 300, 900, "diag", "dense"), (np.float32, 1024, 4000, "diag", "diag")])
def test_n_sharded_single_regressor(B, dtype, D, N, noise, prior):
    # the observations of ONE regressor split into column blocks (as over ranks); the additive statistics of the
    # blocks are summed (here with torch, across ranks by ONE all-reduce) and every rank finishes redundantly
    import torch
    from blr_amd import _abi, sharding

    rng = _rng(9900 + D)
    X = rng.standard_normal((D, N)).astype(dtype)
    mw = (0.5 * rng.standard_normal(D)).astype(dtype)
    if prior == "dense":
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        Lw = (Bm @ Bm.T + np.eye(D)).astype(dtype)
    else:
        Lw = np.exp(0.3 * rng.standard_normal(D)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(dtype) if noise == "diag" else np.array([0.37], dtype=dtype)
    y = rng.standard_normal(N).astype(dtype)
    f64 = lambda a: np.asarray(a, dtype=float)
    Lw_ref = f64(Lw) if prior == "dense" else np.diag(f64(Lw))
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(f64(mw), Lw_ref, f64(X), f64(s) if noise == "diag" else np.float64(s[0]), f64(y))
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    h = _abi.default_handle()
    # raw ABI calls on torch tensors: the library must run on the stream torch fills them on (zero-fills included)
    h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    rows, cols = sharding.stats_shape(D)
    bounds = [0, N // 3, N // 3 + 1, N]  # three uneven "ranks", one of them a single column
    tot = torch.zeros((cols, rows), dtype=tdt, device=dev)
    sc = torch.zeros(2, dtype=torch.float64, device=dev)
    mw_t = torch.tensor(mw, device=dev)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        Xl = torch.tensor(np.ascontiguousarray(X[:, lo:hi].T), device=dev)
        yl = torch.tensor(y[lo:hi], device=dev)
        sl = torch.tensor(s[lo:hi] if noise == "diag" else s, device=dev)
        st = torch.zeros((cols, rows), dtype=tdt, device=dev)
        sca = torch.zeros(2, dtype=torch.float64, device=dev)
        h.gram_stats(dtype, _abi.LAYOUT_COLVECS, D, hi - lo, Xl.data_ptr(), D, yl.data_ptr(),
                     _abi.NOISE_DIAGONAL if noise == "diag" else _abi.NOISE_ISOTROPIC, sl.data_ptr(), mw_t.data_ptr(), st.data_ptr(), rows,
                     sca.data_ptr())
        h.synchronize()
        tot += st
        sc += sca
    Lw_t = torch.tensor(np.ascontiguousarray(Lw), device=dev)
    mw_p = torch.empty(D, dtype=tdt, device=dev); Tp = torch.zeros((D, D), dtype=tdt, device=dev)
    Ap = torch.zeros((D, D), dtype=tdt, device=dev)
    lp = torch.zeros(1, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    h.posterior_from_stats(dtype, D, N, tot.data_ptr(), rows, sc.data_ptr(), _abi.PRIOR_DENSE if prior == "dense" else _abi.PRIOR_DIAGONAL,
                           mw_t.data_ptr(), Lw_t.data_ptr(), D if prior == "dense" else 1, mw_p.data_ptr(), Tp.data_ptr(), D, Ap.data_ptr(), D,
                           lp.data_ptr(), info.data_ptr())
    h.synchronize()
    assert int(info.item()) == 0
    rt = 1e-9 if dtype == np.float64 else 3e-3
    assert float(lp.item()) == pytest.approx(lp_o, rel=1e-10 if dtype == np.float64 else 3e-4)
    np.testing.assert_allclose(mw_p.cpu().numpy(), mw_o, rtol=rt, atol=rt * 10)
    np.testing.assert_allclose(Tp.cpu().numpy().T, T_o, rtol=rt, atol=rt * 10)  # column-major upper factor
    np.testing.assert_allclose(Ap.cpu().numpy(), A_o, rtol=rt, atol=rt * 10)
    # the helper (no process group: a single "rank" holding everything) gives the same answer
    Xall = torch.tensor(np.ascontiguousarray(X.T), device=dev)
    m2, T2, lp2 = sharding.posterior_n_sharded(h, Xall, torch.tensor(y, device=dev), torch.tensor(s, device=dev), mw_t, Lw_t, N)
    assert lp2 == pytest.approx(lp_o, rel=1e-10 if dtype == np.float64 else 3e-4)
    np.testing.assert_allclose(m2.cpu().numpy(), mw_o, rtol=rt, atol=rt * 10)
    h.reset_stream()


# ---- round-2 robustness: observation-noise variances must be positive (reference :79 throws PosDefException) ------------
@pytest.mark.parametrize("D,N", [(5, 40), (128, 300), (128, 256), (200, 500)])
@pytest.mark.parametrize("bad", [-0.5, 0.0, np.nan])
def test_nonpositive_noise_is_reported(B, D, N, bad):
    rng = _rng(11000 + D + N)
    X = rng.standard_normal((D, N))
    y = rng.standard_normal(N)
    f = B.BayesianLinearRegressor(np.zeros(D), B.Diagonal(np.ones(D)))
    s = np.exp(0.2 * rng.standard_normal(N))
    k = 17
    s[k] = bad
    s[k + 9] = bad  # the FIRST offending index is reported, like LAPACK's info
    for fn in (B.logpdf, B.posterior):
        with pytest.raises(B.PosDefException) as ei:
            fn(f(np.asfortranarray(X), s), y)
        assert ei.value.info == k + 1
    with pytest.raises(B.PosDefException) as ei:  # isotropic: sigma^2 itself
        B.logpdf(f(np.asfortranarray(X), bad), y)
    assert ei.value.info == 1
    with pytest.raises(B.PosDefException) as ei:  # rand factorises Sigma_y (:52)
        B.rand(rng, f(np.asfortranarray(X), s), 2)
    assert ei.value.info == k + 1
    # var() only ADDS diag(Sigma_y) (:43): no exception, as in the reference
    v = B.var(f(np.asfortranarray(X), s))
    assert v.shape == (N,)
    # a non-positive Diagonal prior is caught wherever the reference factorises it (:41, :51, sampling_functions.jl:29)
    g = B.BayesianLinearRegressor(np.zeros(D), B.Diagonal(np.where(np.arange(D) == 3, bad, 1.0)))
    for call in (lambda: B.var(g(np.asfortranarray(X), 0.1)), lambda: B.rand(rng, g(np.asfortranarray(X), 0.1), 2), lambda: B.rand(rng, g)):
        with pytest.raises(B.PosDefException) as ei:
            call()
        assert ei.value.info == 4


def test_nonpositive_noise_does_not_poison_a_device_batch(B):
    import torch
    from blr_amd import _abi

    dev = torch.device("cuda:0")
    Bn, D, N = 6, 128, 512
    g = torch.Generator(device=dev).manual_seed(7)
    X = torch.randn((Bn, N, D), generator=g, dtype=torch.float64, device=dev)
    y = torch.randn((Bn, N), generator=g, dtype=torch.float64, device=dev)
    s = torch.exp(0.1 * torch.randn((Bn, N), generator=g, dtype=torch.float64, device=dev))
    s[2, 100] = -1.0
    s[4, 7] = 0.0
    mw = torch.zeros((Bn, D), dtype=torch.float64, device=dev)
    dpr = torch.ones(D, dtype=torch.float64, device=dev)
    lp = torch.zeros(Bn, dtype=torch.float64, device=dev)
    info = torch.full((Bn,), -77, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    h = _abi.default_handle()
    h.posterior_batched(np.float64, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, Bn, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N,
                        _abi.NOISE_DIAGONAL, s.data_ptr(), N, _abi.PRIOR_DIAGONAL, mw.data_ptr(), D, dpr.data_ptr(), 1, 0,
                        None, D, None, D, D * D, None, D, D * D, lp.data_ptr(), info.data_ptr())
    h.synchronize()
    assert info.cpu().tolist() == [0, 0, 101, 0, 8, 0]
    lpc = lp.cpu().numpy()
    assert np.isnan(lpc[[2, 4]]).all() and np.isfinite(lpc[[0, 1, 3, 5]]).all()
    Xh, yh, sh = X.cpu().numpy(), y.cpu().numpy(), s.cpu().numpy()
    for r in (0, 5):
        assert lpc[r] == pytest.approx(O.logpdf_literal(np.zeros(D), np.ones(D), Xh[r].T, sh[r], yh[r]), rel=1e-10)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nb,D,N,prior", [(2, 200, 300, "diagonal"), (3, 384, 500, "dense"), (5, 330, 260, "pdmat"), (6, 640, 900, "diagonal"),
                                          (9, 130, 64, "dense"), (21, 160, 90, "diagonal")])
def test_large_d_batched_regressors_share_the_factorisation_launches(B, dtype, nb, D, N, prior, opt):
    # B > 1 at D > 128: groups of regressors step through every launch of the update together
    # (posterior_large_group; blockIdx.y of the panel and trailing-update kernels).  Every regressor against the oracle, a
    # regressor whose system is not positive definite must fail alone (its neighbours in the group untouched), and the
    # result has to be what one-regressor-at-a-time launches give up to the summation order of the Gram partials (the
    # split-K factor is chosen for the whole group's launch).
    a = B._abi
    h = a.default_handle()
    rng = _rng(6100 + nb + D)
    X = rng.standard_normal((nb, N, D)).astype(dtype)
    mw = (0.1 * rng.standard_normal((nb, D))).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal((nb, N))).astype(dtype)
    y = rng.standard_normal((nb, N)).astype(dtype)
    bad = 1 if nb > 2 else None
    kb = D - 60  # where the bad regressor's prior breaks (0-based)
    if prior == "diagonal":
        Lw_arg = np.exp(0.3 * rng.standard_normal((nb, D))).astype(dtype)
        if bad is not None:
            Lw_arg[bad, kb] = -4.0
        Lw_dense = [np.diag(Lw_arg[b].astype(np.float64)) for b in range(nb)]
        pk, ldl, strideLw = a.PRIOR_DIAGONAL, 1, D
    else:
        Ms = []
        for b in range(nb):
            Bm = rng.standard_normal((D, D)) / np.sqrt(D)
            Ms.append(Bm @ Bm.T + np.eye(D))
        if prior == "dense":
            if bad is not None:
                Lc = np.linalg.cholesky(Ms[bad]); Lc[kb, kb] = 0.0
                Ms[bad] = Lc @ Lc.T; Ms[bad][kb, kb] -= 1.0
            Lw_arg = np.stack([M.astype(dtype) for M in Ms])  # symmetric: either order
            Lw_dense = [np.asarray(Lw_arg[b], dtype=np.float64) for b in range(nb)]
            pk = a.PRIOR_DENSE
        else:
            U = [O.chol_upper(M).astype(dtype) for M in Ms]
            if bad is not None:
                U[bad][7, 7] = 0.0  # a singular factor: reported as column 8 of the prior
            Lw_arg = np.stack([U[b].T.copy() for b in range(nb)])
            Lw_dense = [U[b].astype(np.float64).T @ U[b].astype(np.float64) for b in range(nb)]
            pk = a.PRIOR_UPPER_FACTOR
        ldl, strideLw = D, D * D

    def run():
        mw_post = np.zeros((nb, D), dtype=dtype); T_post = np.zeros((nb, D, D), dtype=dtype); Lw_post = np.zeros((nb, D, D), dtype=dtype)
        lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(dtype, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_DIAGONAL, s, N, pk,
                            mw, D, Lw_arg, ldl, strideLw, mw_post, D, T_post, D, D * D, Lw_post, D, D * D, lp, info)
        return mw_post, T_post, Lw_post, lp, info

    mw_post, T_post, Lw_post, lp, info = run()
    want = [0] * nb
    if bad is not None:
        want[bad] = {"diagonal": kb + 1, "dense": kb + 1, "pdmat": 8}[prior]
    assert info.tolist() == want
    tol = 1e-9 if dtype == np.float64 else 3e-3
    for b in range(nb):
        if b == bad:
            assert np.isnan(lp[b])
            continue
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw[b].astype(float), Lw_dense[b], X[b].T.astype(float), s[b].astype(float), y[b].astype(float))
        assert lp[b] == pytest.approx(lp_o, rel=tol)
        np.testing.assert_allclose(mw_post[b], mw_o, rtol=10 * tol, atol=tol * np.abs(mw_o).max())
        Tn = np.triu(T_post[b].T.astype(np.float64))
        np.testing.assert_allclose(Tn.T @ Tn, A_o, rtol=tol, atol=tol * np.abs(A_o).max())
    opt("CHAIN_BATCH", "1")
    mw1, T1, L1, lp1, info1 = run()
    assert info1.tolist() == want
    ok = [b for b in range(nb) if b != bad]
    eps = 1e-12 if dtype == np.float64 else 2e-5
    np.testing.assert_allclose(lp1[ok], lp[ok], rtol=eps * 10)
    np.testing.assert_allclose(mw1[ok], mw_post[ok], rtol=0, atol=eps * 100 * np.abs(mw1[ok]).max())
    np.testing.assert_allclose(T1[ok], T_post[ok], rtol=0, atol=eps * 10 * np.abs(T1[ok]).max())
    # a workspace bound that holds only a few regressors: the batch runs as several groups, the last one smaller
    opt("CHAIN_BATCH", None)
    opt("CHAIN_WS_MB", "12")
    mw2, T2, L2, lp2, info2 = run()
    assert info2.tolist() == want
    np.testing.assert_allclose(lp2[ok], lp[ok], rtol=eps * 10)
    np.testing.assert_allclose(mw2[ok], mw_post[ok], rtol=0, atol=eps * 100 * np.abs(mw1[ok]).max())


@pytest.mark.parametrize("D,N", [(3, 11), (64, 130), (128, 300), (200, 260)])
@pytest.mark.parametrize("prior", ["diagonal", "dense", "pdmat"])
def test_map_over_regressors_is_one_batched_call(B, D, N, prior):
    # logpdf_map / posterior_map: the reference's methods mapped over a collection of finite regressors (:55-69), served by
    # ONE blr_posterior_batched_* call when the problems have one shape.  Same results as the one-at-a-time calls (which are
    # pinned to the oracle above), same wrapper types, the first failing problem raises with its position, and collections
    # that cannot be batched (mixed layouts / priors) still work.
    rng = _rng(7000 + D)
    nb = 5
    fxs, ys = [], []
    for b in range(nb):
        X = rng.standard_normal((D, N))
        if prior == "diagonal":
            Lw = B.Diagonal(np.exp(0.3 * rng.standard_normal(D)))
        else:
            Bm = rng.standard_normal((D, D)) / np.sqrt(D)
            M = Bm @ Bm.T + np.eye(D)
            Lw = M if prior == "dense" else B.PDMat(O.chol_upper(M))
        f = B.BayesianLinearRegressor(0.2 * rng.standard_normal(D), Lw)
        fxs.append(f(B.ColVecs(np.asfortranarray(X)), np.exp(0.2 * rng.standard_normal(N))))
        ys.append(rng.standard_normal(N))
    lps = B.logpdf_map(fxs, ys)
    posts = B.posterior_map(fxs, ys)
    for fx, y, lp, fp in zip(fxs, ys, lps, posts):
        assert lp == pytest.approx(B.logpdf(fx, y), rel=1e-12)
        one = B.posterior(fx, y)
        np.testing.assert_allclose(fp.mw, one.mw, rtol=1e-10, atol=1e-12)
        assert type(fp.Lw) is type(one.Lw)
        if prior == "pdmat":
            np.testing.assert_allclose(fp.Lw.U, one.Lw.U, rtol=1e-10, atol=1e-12)
        else:
            np.testing.assert_allclose(fp.Lw.toarray(), one.Lw.toarray(), rtol=1e-10, atol=1e-10)
    # a basis-function regressor maps through its feature map and comes back wrapped
    phi = lambda x: B.ColVecs(np.asfortranarray(np.tanh(x.X)))  # noqa: E731
    bfs = [B.BasisFunctionRegressor(fx.f, phi)(fx.x, fx.Sy) for fx in fxs]
    pb = B.posterior_map(bfs, ys)
    assert all(isinstance(p_, B.BasisFunctionRegressor) for p_ in pb)
    np.testing.assert_allclose(pb[2].blr.mw, B.posterior(bfs[2], ys[2]).blr.mw, rtol=1e-10, atol=1e-12)
    # the first problem that is not positive definite raises, and says which one it was
    s_bad = np.ones(N); s_bad[min(4, N - 1)] = -1.0
    broken = list(fxs)
    broken[3] = fxs[3].f(fxs[3].x, s_bad)
    with pytest.raises(B.PosDefException) as ei:
        B.logpdf_map(broken, ys)
    assert ei.value.info == min(4, N - 1) + 1 and ei.value.index == 3
    # mixed collection (a RowVecs problem among ColVecs ones): no batch, same answers
    mixed = list(fxs)
    mixed[1] = fxs[1].f(B.RowVecs(np.asfortranarray(fxs[1].x.X.T)), fxs[1].Sy)
    np.testing.assert_allclose(B.logpdf_map(mixed, ys), lps, rtol=1e-12)
    assert B.logpdf_map([], []) == []


@pytest.mark.parametrize("splits", ["1,1", "3,3", "4,2", "3,3,1", "4,2,2", "5,5,3", "2,1,3", "6,3,5", "4,4,6"])
@pytest.mark.parametrize("prior", ["diagonal", "pdmat"])
def test_large_d_gram_split_plans(B, splits, prior, opt):
    # The Gram launch of the large-D path cuts every macro tile's columns into ranges: one factor for all tiles, the diagonal
    # tiles with their own (single-round launches), or three kinds of work items in planned dispatch order (multi-round launches:
    # "off-diagonal, diagonal, tiles with one range less").  Whatever the plan -- forced here through the measurement switch,
    # incl. ranges that come out empty -- the partials have to add up to the same posterior (fp32, the ring-loop path).
    rng = _rng(7300)
    D, N = 512, 700
    X = rng.standard_normal((D, N)).astype(np.float32)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(np.float32)
    y = rng.standard_normal(N).astype(np.float32)
    mw = (0.1 * rng.standard_normal(D)).astype(np.float32)
    if prior == "diagonal":
        dvec = np.exp(0.3 * rng.standard_normal(D)).astype(np.float32)
        Lw_d, Lw_arg = np.diag(dvec.astype(float)), B.Diagonal(dvec)
    else:
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        U = O.chol_upper(Bm @ Bm.T + np.eye(D)).astype(np.float32)
        Lw_d, Lw_arg = U.astype(float).T @ U.astype(float), B.PDMat(U)
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw.astype(float), Lw_d, X.astype(float), s.astype(float), y.astype(float))
    opt("GRAM_SPLITS", splits)
    fx = B.BayesianLinearRegressor(mw, Lw_arg)(np.asfortranarray(X), s)
    assert B.logpdf(fx, y) == pytest.approx(lp_o, rel=2e-4)
    fp = B.posterior(fx, y)
    np.testing.assert_allclose(fp.mw, mw_o, rtol=0, atol=2e-3 * np.abs(mw_o).max())
    A = fp.Lw.toarray() if prior == "diagonal" else fp.Lw.U.astype(float).T @ fp.Lw.U.astype(float)
    np.testing.assert_allclose(A, A_o, rtol=0, atol=2e-5 * np.abs(A_o).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_large_d_repeated_calls_on_resident_buffers(B, dtype):
    # The large-D chain keeps its synchronisation state on the device (arrival counter of the panel kernel, tickets and
    # exchange tags of the wavefront solve), re-armed by the kernels themselves: call after call on the same buffers with
    # new contents has to see the new contents, report a failure and recover from it, and survive a differently shaped
    # call in between.
    import torch
    from blr_amd import _abi

    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    D, N = 384, 700
    rng = _rng(5300)
    h = _abi.default_handle()
    X = torch.empty((N, D), dtype=tdt, device=dev)   # ColVecs: column n = X[n, :]
    y = torch.empty((N,), dtype=tdt, device=dev)
    s = torch.empty((N,), dtype=tdt, device=dev)
    mw = torch.zeros((D,), dtype=tdt, device=dev)
    dpr = torch.ones((D,), dtype=tdt, device=dev)
    mwp = torch.empty((D,), dtype=tdt, device=dev)
    Tp = torch.empty((D, D), dtype=tdt, device=dev)
    lp = torch.zeros((1,), dtype=torch.float64, device=dev)
    info = torch.full((1,), -77, dtype=torch.int32, device=dev)

    def call():
        h.posterior_batched(dtype, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, 1, D, N, X.data_ptr(), D, 0, y.data_ptr(), 0,
                            _abi.NOISE_DIAGONAL, s.data_ptr(), 0, _abi.PRIOR_DIAGONAL, mw.data_ptr(), 0, dpr.data_ptr(), 1, 0,
                            mwp.data_ptr(), D, Tp.data_ptr(), D, D * D, None, D, D * D, lp.data_ptr(), info.data_ptr())
        h.synchronize()

    tol = 1e-9 if dtype == np.float64 else 3e-3
    for it in range(7):
        Xh = rng.standard_normal((D, N))
        sh = np.exp(0.3 * rng.standard_normal(N))
        yh = Xh.T @ rng.standard_normal(D) / np.sqrt(D) + np.sqrt(sh) * rng.standard_normal(N)
        dh = np.exp(0.2 * rng.standard_normal(D))
        mh = 0.1 * rng.standard_normal(D) if it % 2 else np.zeros(D)
        bad = it == 4
        if bad:
            dh[200] = -1e9  # the prior is not positive: the replayed sequence has to say so and stay usable
        X.copy_(torch.from_numpy(np.ascontiguousarray(Xh.T)).to(tdt))
        y.copy_(torch.from_numpy(yh).to(tdt))
        s.copy_(torch.from_numpy(sh).to(tdt))
        dpr.copy_(torch.from_numpy(dh).to(tdt))
        mw.copy_(torch.from_numpy(mh).to(tdt))
        info.fill_(-77)
        call()
        if bad:
            assert int(info.cpu()[0]) == 201
            continue
        assert int(info.cpu()[0]) == 0
        Xr, yr, sr = X.cpu().numpy().astype(float).T, y.cpu().numpy().astype(float), s.cpu().numpy().astype(float)
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw.cpu().numpy().astype(float), np.diag(dpr.cpu().numpy().astype(float)), Xr, sr, yr)
        assert float(lp.cpu()[0]) == pytest.approx(lp_o, rel=tol)
        np.testing.assert_allclose(mwp.cpu().numpy(), mw_o, rtol=10 * tol, atol=tol)
        if it == 2:  # another shape through the same handle (may grow the workspace)
            f2 = B.BayesianLinearRegressor(np.zeros(520, dtype), B.Diagonal(np.ones(520, dtype)))
            X2 = rng.standard_normal((520, 300)).astype(dtype)
            y2 = rng.standard_normal(300).astype(dtype)
            lp2 = O.logpdf_literal(np.zeros(520), np.ones(520), X2.astype(float), 0.5, y2.astype(float))
            assert B.logpdf(f2(np.asfortranarray(X2), dtype(0.5)), y2) == pytest.approx(lp2, rel=tol)


def test_mean_length_mismatch_is_rejected(B):
    # a regressor of dimension 3 applied to 5-dimensional inputs: DimensionMismatch in the reference, never a read past mw
    rng = _rng(11500)
    f = B.BayesianLinearRegressor(np.zeros(3), B.Diagonal(np.ones(3)))
    X = rng.standard_normal((5, 20))
    y = rng.standard_normal(20)
    for call in (lambda: B.logpdf(f(X, 0.1), y), lambda: B.posterior(f(X, 0.1), y), lambda: B.mean(f(X, 0.1)), lambda: B.var(f(X, 0.1)),
                 lambda: B.rand(rng, f(X, 0.1), 2), lambda: B.logpdf(f(X, 0.1), np.stack([y, y], axis=1)),
                 lambda: B.logpdf_and_gradient(f(X, 0.1), y)):
        with pytest.raises(ValueError):
            call()


def test_rff_square_rowvecs_input(B):
    # N == D_in: the orientation of a RowVecs input must come from the container, not from a shape comparison
    rng = _rng(11600)
    Din = N = 8
    D = 32
    Xin = rng.standard_normal((Din, N))
    Om = rng.standard_normal((Din, D))
    beta = 2 * np.pi * rng.random(D)
    rff = B.RandomFourierFeatures(Om, beta)
    Phi_ref = np.sqrt(2.0 / D) * np.cos(Om.T @ Xin + beta[:, None])
    np.testing.assert_allclose(rff(B.ColVecs(np.asfortranarray(Xin))).X, Phi_ref, atol=1e-12)
    for arr in (np.asfortranarray(Xin.T), np.ascontiguousarray(Xin.T)):  # N x D_in in both memory orders
        np.testing.assert_allclose(rff(B.RowVecs(arr)).X, Phi_ref.T, atol=1e-12)
    np.testing.assert_allclose(rff(B.ColVecs(np.ascontiguousarray(Xin))).X, Phi_ref, atol=1e-12)


# ---- dense Sigma_y and cov(fx) on the device: the reference's own toy problems, unmodified --------
# /root/reference/test/test_utils.jl:4-10 builds every toy problem with a DENSE Sigma_y = C C' + I; the tests below are the
# reference's testsets (test/bayesian_linear_regression.jl) on that construction, for Matrix / ColVecs / RowVecs inputs.
def _toy_inputs(B, X):
    return {"matrix": np.asfortranarray(X), "colvecs": B.ColVecs(np.asfortranarray(X)), "rowvecs": B.RowVecs(np.ascontiguousarray(X.T))}


@pytest.mark.parametrize("Tx", ["matrix", "colvecs", "rowvecs"])
def test_reference_testsets_dense_noise(B, Tx):
    # logpdf: test/bayesian_linear_regression.jl:22-38 (N=13, D=7, independent N x N Gaussian formula)
    rng = _rng(12000)
    N, D = 13, 7
    X, mw, Lw, Sy = O.generate_toy_problem(rng, N, D)  # dense Sigma_y
    f = B.BayesianLinearRegressor(mw, Lw)
    x = _toy_inputs(B, X)[Tx]
    y = B.rand(rng, f(x, Sy))
    assert y.shape == (N,)
    m = X.T @ mw
    Sig = X.T @ np.linalg.solve(Lw, X) + Sy
    d = y - m
    naive = -(N * np.log(2 * np.pi) + np.linalg.slogdet(Sig)[1] + d @ np.linalg.solve(Sig, d)) / 2
    assert B.logpdf(f(x, Sy), y) == pytest.approx(naive, rel=1e-10)
    assert B.logpdf(f(x, Sy), y) == pytest.approx(O.logpdf_literal(mw, Lw, X, Sy, y), rel=1e-11)
    # posterior, low noise: :40-48 -- mean at X reproduces y, every entry of the predictive covariance < 1000 eps
    eps = np.finfo(float).eps
    y0 = B.rand(rng, f(x, eps))
    fp = B.posterior(f(x, eps), y0)
    m0 = B.mean(fp(x, eps))
    assert np.linalg.norm(m0 - y0) <= np.sqrt(eps) * max(np.linalg.norm(m0), np.linalg.norm(y0))  # Julia's `isapprox` (rtol = sqrt(eps), norm-wise)
    assert np.all(B.cov(fp(x, eps)) < 1000 * eps)
    # posterior vs the literal oracle on the dense problem (:60-69, :79-82)
    mw_o, T_o, A_o = O.posterior_literal(mw, Lw, X, Sy, y)
    fq = B.posterior(f(x, Sy), y)
    np.testing.assert_allclose(fq.mw, mw_o, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(fq.Lw.toarray(), A_o, rtol=1e-9, atol=1e-11)
    # repeated conditioning: :49-70 -- block-diagonal dense Sigma_y, two sequential updates == one shot
    N1 = N - 3
    S1, S2 = Sy[:N1, :N1], Sy[N1:, N1:]
    Sblk = np.zeros((N, N))
    Sblk[:N1, :N1], Sblk[N1:, N1:] = S1, S2
    X1, X2 = X[:, :N1], X[:, N1:]
    x1, x2 = _toy_inputs(B, X1)[Tx], _toy_inputs(B, X2)[Tx]
    f1 = B.posterior(f(x1, S1), y[:N1])
    f2 = B.posterior(f1(x2, S2), y[N1:])
    f12 = B.posterior(f(x, Sblk), y)
    Xp = rng.standard_normal((D, 9))
    xp = _toy_inputs(B, Xp)[Tx]
    np.testing.assert_allclose(B.mean(f2(xp, 0.1)), B.mean(f12(xp, 0.1)), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(B.cov(f2(xp, 0.1)), B.cov(f12(xp, 0.1)), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("N,D,prior", [(11, 3, "dense"), (40, 16, "pdmat"), (300, 64, "diagonal"), (200, 128, "dense"), (513, 130, "dense")])
def test_finitegp_interface_cov(B, N, D, prior):
    # what AbstractGPs.TestUtils.test_finitegp_primary_and_secondary_public_interface checks (test/...:3-10): cov is symmetric
    # PSD, var == diag(cov), mean_and_cov / mean_and_var agree with the separate calls, marginals carry mean and sqrt(var)
    rng = _rng(12100 + N)
    X, mw, Lw, Sy = O.generate_toy_problem(rng, N, D)
    if prior == "diagonal":
        dvec = np.exp(0.3 * rng.standard_normal(D))
        Lw, Lw_arg = np.diag(dvec), B.Diagonal(dvec)
    else:
        Lw_arg = Lw if prior == "dense" else B.PDMat(O.chol_upper(Lw))
    f = B.BayesianLinearRegressor(mw, Lw_arg)
    for noise in (Sy, np.exp(0.2 * rng.standard_normal(N)), 0.3):
        fx = f(np.asfortranarray(X), noise)
        Cv = B.cov(fx)
        C_o = O.cov(mw, Lw, X, noise)
        np.testing.assert_allclose(Cv, C_o, rtol=1e-9, atol=1e-10)
        assert np.array_equal(Cv, Cv.T) and np.min(np.linalg.eigvalsh(Cv)) > 0
        np.testing.assert_allclose(B.var(fx), np.diag(Cv), rtol=1e-9)
        m, C2 = B.mean_and_cov(fx)
        np.testing.assert_allclose(m, X.T @ mw, rtol=1e-10, atol=1e-12)
        np.testing.assert_array_equal(C2, Cv)
        m2, v2 = B.mean_and_var(fx)
        np.testing.assert_allclose(m2, m, rtol=1e-10, atol=1e-12)
        ms = B.marginals(fx)
        np.testing.assert_allclose([n.mu for n in ms], m, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(B.std(ms) ** 2, np.diag(Cv), rtol=1e-9)
    Cr = B.cov(f(B.RowVecs(np.ascontiguousarray(X.T)), Sy))
    np.testing.assert_allclose(Cr, O.cov(mw, Lw, X, Sy), rtol=1e-9, atol=1e-10)
    # logpdf(fx, Y::Matrix) == column-wise logpdfs (TestUtils), dense noise
    Y = rng.standard_normal((N, 3))
    lps = B.logpdf(f(np.asfortranarray(X), Sy), Y)
    for j in range(3):
        assert lps[j] == pytest.approx(O.logpdf_literal(mw, Lw, X, Sy, Y[:, j]), rel=1e-10)


def test_rand_dense_noise_given_normals_and_moments(B):
    from blr_amd import _abi

    rng = _rng(12200)
    N, D, S = 11, 3, 5
    X, mw, Lw, Sy = O.generate_toy_problem(rng, N, D)
    Z1, Z2 = rng.standard_normal((D, S)), rng.standard_normal((N, S))
    Y = np.empty((N, S), order="F")
    _abi.default_handle().rand_dense_noise(np.float64, _abi.MEM_HOST, _abi.LAYOUT_COLVECS, D, N, S, np.asfortranarray(X), D,
                                           np.asfortranarray(Sy), N, _abi.PRIOR_DENSE, mw, np.asfortranarray(Lw), D,
                                           np.asfortranarray(Z1), D, np.asfortranarray(Z2), N, Y, N)
    np.testing.assert_allclose(Y, O.rand(mw, Lw, X, Sy, Z1, Z2), rtol=1e-10, atol=1e-12)
    # test/bayesian_linear_regression.jl:11-21: empirical moments of many draws vs mean(fx) / cov(fx)
    f = B.BayesianLinearRegressor(mw, Lw)
    fx = f(np.asfortranarray(X), Sy)
    ys = B.rand(rng, fx, 400_000)
    np.testing.assert_allclose(ys.mean(axis=1), B.mean(fx), atol=2e-2, rtol=2e-2)
    np.testing.assert_allclose(np.cov(ys), B.cov(fx), atol=2e-2, rtol=2e-2)
    # a Sigma_y that is not positive definite: PosDefException from rand, logpdf and posterior (:52, :79)
    bad = Sy.copy()
    bad[4, 4] = -3.0
    y = rng.standard_normal(N)
    for call in (lambda: B.rand(rng, f(np.asfortranarray(X), bad)), lambda: B.logpdf(f(np.asfortranarray(X), bad), y),
                 lambda: B.posterior(f(np.asfortranarray(X), bad), y)):
        with pytest.raises(B.PosDefException) as ei:
            call()
        assert ei.value.info == 5


@pytest.mark.parametrize("dtype,N,D", [(np.float64, 1000, 64), (np.float64, 700, 300), (np.float32, 2048, 128)])
def test_dense_noise_moderate_sizes(B, dtype, N, D):
    rng = _rng(12300 + N)
    X = rng.standard_normal((D, N)).astype(dtype)
    Cm = 0.1 * rng.standard_normal((N, N)) / np.sqrt(N / 16)
    Sy = (Cm @ Cm.T + np.diag(np.exp(0.2 * rng.standard_normal(N)))).astype(dtype)
    mw = (0.2 * rng.standard_normal(D)).astype(dtype)
    dvec = np.exp(0.2 * rng.standard_normal(D)).astype(dtype)
    y = (X.T.astype(float) @ rng.standard_normal(D) / np.sqrt(D) + rng.standard_normal(N)).astype(dtype)
    f64 = lambda a: np.asarray(a, dtype=float)
    lp_o = O.logpdf_literal(f64(mw), np.diag(f64(dvec)), f64(X), f64(Sy), f64(y))
    mw_o, T_o, A_o = O.posterior_literal(f64(mw), np.diag(f64(dvec)), f64(X), f64(Sy), f64(y))
    f = B.BayesianLinearRegressor(mw, B.Diagonal(dvec))
    fx = f(np.asfortranarray(X), Sy)
    lp = B.logpdf(fx, y)
    fp = B.posterior(fx, y)
    if dtype == np.float64:
        assert lp == pytest.approx(lp_o, rel=1e-10)
        np.testing.assert_allclose(fp.mw, mw_o, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(fp.Lw.toarray(), A_o, rtol=1e-9, atol=1e-9)
    else:
        _assert_fp32_within_lapack(mw, np.diag(dvec), np.asfortranarray(X), Sy, y, fp.mw, fp.Lw.toarray(), lp,
                                   what="dense noise fp32")


# ---- the exchange through RCCL called directly from the C ABI (blr_comm_*, blr_logpdf_allgather_sum) --------------------
def test_rccl_direct_single_rank(B):
    import torch
    from blr_amd import _abi

    dev = torch.device("cuda:0")
    h = _abi.Handle(0)
    try:
        assert h.comm_size() == 1 and h.comm_rank() == 0  # no communicator yet
        lp = torch.randn(1000, dtype=torch.float64, device=dev)
        allv = torch.zeros(1000, dtype=torch.float64, device=dev)
        tot = torch.zeros(1, dtype=torch.float64, device=dev)
        ref = torch.zeros(1, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        h.logpdf_allgather_sum(1000, lp.data_ptr(), allv.data_ptr(), tot.data_ptr())  # degenerate form: copy + fixed-order sum
        h.logpdf_sum(_abi.MEM_DEVICE, 1000, lp.data_ptr(), ref.data_ptr())
        h.synchronize()
        assert torch.equal(allv, lp) and tot.item() == ref.item()
        # a real communicator of one rank: ncclCommInitRank / ncclAllGather / ncclAllReduce run on the handle's stream
        uid = _abi.Handle.comm_unique_id()
        assert len(uid) == 128
        h.comm_init(1, 0, uid)
        assert h.comm_size() == 1 and h.comm_rank() == 0
        allv.zero_(); tot.zero_()
        torch.cuda.synchronize()
        h.logpdf_allgather_sum(1000, lp.data_ptr(), allv.data_ptr(), tot.data_ptr())
        h.synchronize()
        assert torch.equal(allv, lp) and tot.item() == ref.item()  # bit-identical to the local fixed-order sum
        st = torch.arange(64, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        h.allreduce_sum(False, st.data_ptr(), 64)
        h.synchronize()
        assert torch.equal(st, torch.arange(64, dtype=torch.float32, device=dev))
        with pytest.raises(_abi.BLRError):
            h.comm_init(1, 0, uid)  # already has a communicator
        h.comm_destroy()
        assert h.comm_size() == 1
    finally:
        h.close()



# ---- rank-k update of a resident state (blr_update_factor_*) -----------------------------------
@pytest.mark.parametrize("route", ["always", "never"])
def test_update_factor_repeated_conditioning_10_plus_3(B, route, opt):
    opt("SWEEP", route)  # Givens sweep / in-place re-factorisation: same answers
    # reference test/bayesian_linear_regression.jl:49-70 (13 = 10 + 3 split) through the resident state, against the ONE-SHOT
    # oracle on all 13 observations; the second batch (k = 3) goes through the O(k D^2) Givens sweep
    rng = _rng(71)
    N, D = 13, 7
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    Xp = rng.standard_normal((D, N))
    f = B.BayesianLinearRegressor(mw, Lw)
    y = B.rand(rng, f(X, s))
    N1 = N - 3
    st = B.ResidentPosterior(B.posterior(f(X[:, :N1], s[:N1]), y[:N1]))
    lp2 = st.condition(X[:, N1:], s[N1:], y[N1:])
    f2 = st.regressor()
    mw_o, T_o, L_o = O.posterior_literal(mw, Lw, X, s, y)
    np.testing.assert_allclose(f2.mw, mw_o, rtol=1e-9)
    np.testing.assert_allclose(f2.Lw.toarray(), L_o, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(B.mean(f2(Xp, s)), O.mean(mw_o, Xp), rtol=1e-9)
    np.testing.assert_allclose(B.cov(f2(Xp, s)), O.cov(mw_o, L_o, Xp, s), rtol=1e-9, atol=1e-12)
    # evidence increment = log p(y2 | y1) (chain rule against the oracle's two evidences)
    lp_o = O.logpdf_literal(mw, Lw, X, s, y) - O.logpdf_literal(mw, Lw, X[:, :N1], s[:N1], y[:N1])
    assert lp2 == pytest.approx(lp_o, rel=1e-9)
    # starting from the PRIOR as the resident state and feeding everything in three batches gives the same posterior
    st0 = B.ResidentPosterior(f)
    tot = st0.condition(X[:, :5], s[:5], y[:5]) + st0.condition(X[:, 5:N1], s[5:N1], y[5:N1]) + st0.condition(X[:, N1:], s[N1:], y[N1:])
    np.testing.assert_allclose(st0.regressor().mw, mw_o, rtol=1e-9)
    np.testing.assert_allclose(st0.regressor().Lw.toarray(), L_o, rtol=1e-9, atol=1e-12)
    assert tot == pytest.approx(O.logpdf_literal(mw, Lw, X, s, y), rel=1e-9)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("D,k", [(1, 1), (5, 3), (16, 16), (64, 7), (100, 1), (128, 16), (128, 3), (128, 64), (37, 40), (200, 5)])
@pytest.mark.parametrize("noise", ["diagonal", "isotropic"])
@pytest.mark.parametrize("route", ["always", "auto"])
def test_update_factor_device_batched(B, dtype, D, k, noise, route, opt):
    if route == "always":
        opt("SWEEP", "always")
    else:
        opt("SWEEP", None)
    # device-resident batched state: every (D, k) class -- sweep (D <= 128, k <= 16), fused re-factorisation (k > 16) and the
    # large-D path (D > 128) -- against the oracle's one-shot posterior from the same prior state
    import torch

    a = B._abi
    h = a.default_handle()
    rng = _rng(72 + D + 3 * k)
    nb = 3 if D <= 128 else 1
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    X = rng.standard_normal((nb, k, D))  # [k, D] row-major == D x k ColVecs
    mw = rng.standard_normal((nb, D))
    U = np.triu(rng.standard_normal((nb, D, D))) * (0.3 / np.sqrt(D))  # well-conditioned: random triangular factors are not
    for b in range(nb):
        U[b][np.diag_indices(D)] = 1.0 + np.abs(U[b][np.diag_indices(D)])
    s = np.exp(0.3 * rng.standard_normal((nb, k))) if noise == "diagonal" else np.full((1,), 0.37)
    y = rng.standard_normal((nb, k))
    Xd, mwd, yd = (torch.tensor(v, dtype=tdt, device=dev) for v in (X, mw, y))
    sd = torch.tensor(s, dtype=tdt, device=dev)
    Td = torch.tensor(np.transpose(U, (0, 2, 1)).copy(), dtype=tdt, device=dev)  # column-major upper factor per regressor
    lp = torch.zeros(nb, dtype=torch.float64, device=dev)
    info = torch.full((nb,), 7, dtype=torch.int32, device=dev)
    kind = a.NOISE_DIAGONAL if noise == "diagonal" else a.NOISE_ISOTROPIC
    torch.cuda.synchronize()  # torch's fills / copies before the handle's stream touches the buffers
    h.update_factor(dtype, a.MEM_DEVICE, a.LAYOUT_COLVECS, nb, D, k, Xd.data_ptr(), D, k * D, yd.data_ptr(), k, kind, sd.data_ptr(),
                    k if noise == "diagonal" else 0, mwd.data_ptr(), D, Td.data_ptr(), D, D * D, lp.data_ptr(), info.data_ptr())
    torch.cuda.synchronize()
    assert info.cpu().tolist() == [0] * nb
    rtol = 1e-9
    for b in range(nb):
        Ub = U[b].astype(dtype).astype(np.float64)
        Xb, yb, mb = X[b].T.astype(dtype).astype(np.float64), y[b].astype(dtype).astype(np.float64), mw[b].astype(dtype).astype(np.float64)
        sb = (s[b] if noise == "diagonal" else np.full(k, s[0])).astype(dtype).astype(np.float64)
        if dtype == np.float32:
            # one-shot update from the same prior state in fp32 LAPACK is the yardstick (the prior precision U'U is formed in
            # fp64 from the fp32 factor and rounded once: the state the update starts from)
            Tg = Td[b].cpu().numpy().T
            A_g = np.triu(Tg).astype(np.float64).T @ np.triu(Tg).astype(np.float64)
            _assert_fp32_within_lapack(mb.astype(np.float32), (Ub.T @ Ub).astype(np.float32), np.asfortranarray(Xb.astype(np.float32)),
                                       sb.astype(np.float32), yb.astype(np.float32), mwd[b].cpu().numpy(), A_g, lp[b].item(),
                                       floor=(4e-6, 2e-6, 2e-6), what=f"update_factor D={D} k={k} {noise} {route}")
            continue
        mw_o, T_o, L_o = O.posterior_literal(mb, Ub.T @ Ub, Xb, sb, yb)
        Tn = np.triu(Td[b].cpu().numpy().T.astype(np.float64))
        scale = np.abs(L_o).max()
        np.testing.assert_allclose(Tn.T @ Tn, L_o, rtol=rtol, atol=rtol * scale)
        np.testing.assert_allclose(mwd[b].cpu().numpy(), mw_o, rtol=50 * rtol, atol=rtol * np.abs(mw_o).max())
        assert lp[b].item() == pytest.approx(O.logpdf_literal(mb, Ub.T @ Ub, Xb, sb, yb), rel=10 * rtol, abs=10 * rtol)


@pytest.mark.parametrize("route", ["always", "never"])
@pytest.mark.parametrize("D", [9, 200])
def test_update_factor_leaves_the_state_alone_on_failure(B, route, opt, D):
    # D = 200: the large-D pipeline writes the caller's factor from its transpose launch only when the status words are clean
    # (include/blr_mi355x.h: "the state is untouched when info != 0, at every D")
    import torch

    opt("SWEEP", route)

    a = B._abi
    h = a.default_handle()
    rng = _rng(73)
    k = 4
    dev = torch.device("cuda:0")
    U = np.triu(rng.standard_normal((D, D))) + 3 * np.eye(D)
    U[np.diag_indices(D)] = np.abs(U[np.diag_indices(D)])
    Td = torch.tensor(U.T.copy(), dtype=torch.float64, device=dev)
    mwd = torch.tensor(rng.standard_normal(D), dtype=torch.float64, device=dev)
    T0, m0 = Td.clone(), mwd.clone()
    Xd = torch.tensor(rng.standard_normal((k, D)), dtype=torch.float64, device=dev)
    yd = torch.tensor(rng.standard_normal(k), dtype=torch.float64, device=dev)
    sd = torch.tensor([0.5, 0.2, -0.1, 0.3], dtype=torch.float64, device=dev)
    lp = torch.zeros(1, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    h.update_factor(np.float64, a.MEM_DEVICE, a.LAYOUT_COLVECS, 1, D, k, Xd.data_ptr(), D, 0, yd.data_ptr(), 0, a.NOISE_DIAGONAL,
                    sd.data_ptr(), 0, mwd.data_ptr(), 0, Td.data_ptr(), D, 0, lp.data_ptr(), info.data_ptr())
    torch.cuda.synchronize()
    assert info.item() == 3 and np.isnan(lp.item())  # reference :79: PosDefException at the third variance
    assert torch.equal(Td, T0) and torch.equal(mwd, m0)
    # a state that is not a Cholesky factor (non-positive diagonal entry) is reported with the SAME code by both routes --
    # the 1-based index of the entry -- and wins over a bad variance (the reference factors the prior, :78, before the noise, :79)
    Tbad = T0.clone()
    Tbad[5, 5] = -2.0
    Tb0 = Tbad.clone()
    torch.cuda.synchronize()
    h.update_factor(np.float64, a.MEM_DEVICE, a.LAYOUT_COLVECS, 1, D, k, Xd.data_ptr(), D, 0, yd.data_ptr(), 0, a.NOISE_DIAGONAL,
                    sd.data_ptr(), 0, mwd.data_ptr(), 0, Tbad.data_ptr(), D, 0, lp.data_ptr(), info.data_ptr())
    torch.cuda.synchronize()
    assert info.item() == 6 and np.isnan(lp.item())
    assert torch.equal(Tbad, Tb0) and torch.equal(mwd, m0)
    st = B.ResidentPosterior(B.BayesianLinearRegressor(np.zeros(3), B.Diagonal(np.ones(3))))
    with pytest.raises(B.PosDefException):
        st.condition(np.ones((3, 2)), np.array([1.0, 0.0]), np.zeros(2))
    np.testing.assert_array_equal(st.state()[0], np.zeros(3))
    np.testing.assert_array_equal(st.state()[1], np.eye(3))


def test_resident_posterior_state_is_created_by_the_library(B):
    # dense prior: factorised on the device when the state is created (reference :78), same answers as posterior();
    # a prior that is not positive definite reports the failing leading minor, as _cholesky would
    rng = _rng(77)
    N, D = 9, 6
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    y = rng.standard_normal(N)
    st = B.ResidentPosterior(B.BayesianLinearRegressor(mw, Lw))
    m0, T0 = st.state()
    np.testing.assert_array_equal(m0, mw)
    np.testing.assert_allclose(T0.T @ T0, Lw, rtol=1e-12, atol=1e-13)
    assert np.all(np.diag(T0) > 0) and np.all(np.tril(T0, -1) == 0)
    lp = st.condition(X, s, y)
    mw_o, T_o, L_o = O.posterior_literal(mw, Lw, X, s, y)
    assert lp == pytest.approx(O.logpdf_literal(mw, Lw, X, s, y), rel=1e-10)
    np.testing.assert_allclose(st.regressor().mw, mw_o, rtol=1e-9)
    bad = Lw.copy()
    bad[3, 3] = -5.0
    with pytest.raises(B.PosDefException) as ei:
        B.ResidentPosterior(B.BayesianLinearRegressor(mw, bad))
    assert ei.value.info == 4
    with pytest.raises(B.PosDefException) as ei:
        B.ResidentPosterior(B.BayesianLinearRegressor(mw, B.Diagonal(np.array([1.0, 2.0, 0.0, 1.0, 1.0, 1.0]))))
    assert ei.value.info == 3
    U = np.triu(rng.standard_normal((D, D))) + 3 * np.eye(D)
    U[2, 2] = -1.0  # not a Cholesky factor: rejected when the state is created, not at the first condition()
    with pytest.raises(B.PosDefException) as ei:
        B.ResidentPosterior(B.BayesianLinearRegressor(mw, B.PDMat(U)))
    assert ei.value.info == 3


@pytest.mark.parametrize("basis", ["callable", "rff", "rff_square"])
def test_resident_posterior_keeps_the_basis(B, basis):
    # a BasisFunctionRegressor's resident state conditions on phi(x), not on x (with D_in == D the raw inputs used to
    # go in silently), and hands back BasisFunctionRegressor(posterior, phi) -- reference basis_function_regression.jl:62-65
    rng = _rng(78)
    Din, D, N = (5, 5, 40) if basis == "rff_square" else (3, 12, 40)
    Xin = rng.standard_normal((Din, N))
    s = np.exp(0.3 * rng.standard_normal(N))
    y = rng.standard_normal(N)
    mw = 0.3 * rng.standard_normal(D)
    d = np.exp(0.2 * rng.standard_normal(D))
    if basis == "callable":
        Wm = rng.standard_normal((D, Din))
        phi = lambda x: B.ColVecs(np.asfortranarray(np.tanh(Wm @ (x.X if hasattr(x, "X") else x))))
        Phi = np.tanh(Wm @ Xin)
    else:
        phi = B.RandomFourierFeatures(rng.standard_normal((Din, D)), 2 * np.pi * rng.random(D))
        Phi = phi.scale * np.cos(phi.Omega.T @ Xin + phi.phase[:, None])
    bfr = B.BasisFunctionRegressor(B.BayesianLinearRegressor(mw, B.Diagonal(d)), phi)
    st = B.ResidentPosterior(bfr)
    N1 = 25
    lp = st.condition(B.ColVecs(np.asfortranarray(Xin[:, :N1])), s[:N1], y[:N1])
    lp += st.condition(B.ColVecs(np.asfortranarray(Xin[:, N1:])), s[N1:], y[N1:])
    post = st.regressor()
    assert isinstance(post, B.BasisFunctionRegressor) and post.phi is phi
    mw_o, T_o, L_o = O.posterior_literal(mw, d, Phi, s, y)
    assert lp == pytest.approx(O.logpdf_literal(mw, d, Phi, s, y), rel=1e-9)
    np.testing.assert_allclose(post.blr.mw, mw_o, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(post.blr.Lw.toarray(), L_o, rtol=1e-9, atol=1e-11)
    # the state is of feature dimension D: raw inputs of another dimension are a dimension error, never a silent wrong update
    with pytest.raises(ValueError):
        st.condition(np.zeros((Din + 1, 2)), 1.0, np.zeros(2))


# ---- one wavefront per regressor (blr_fused_wave.hpp): D = 32 / 64, ColVecs, diagonal prior ------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("D", [32, 64])
@pytest.mark.parametrize("N", [0, 1, 7, 8, 16, 17, 100, 1000, 1024])
@pytest.mark.parametrize("noise", ["diagonal", "isotropic"])
@pytest.mark.parametrize("prior", ["diagonal", "dense", "factor"])
@pytest.mark.parametrize("split", ["auto", "1", "2"])
def test_wave_kernel_shapes_vs_oracle(B, dtype, D, N, noise, prior, split, opt):
    # the shapes blr_abi.hip routes to fused_wave_kernel (f64: D = 32 and 64; f32: D = 64; D = 32 in f32 stays on the four-wave
    # kernel and runs here as its cross-check): whole 4 KiB stages, ragged tails, no data at all, non-zero prior mean.
    # split: waves per regressor -- "auto" is the router's choice (4 for a batch of 5), 1 and 2 are forced
    if split == "auto":
        opt("WAVE_SPLIT", None)
    else:
        opt("WAVE_SPLIT", split)
    a = B._abi
    h = a.default_handle()
    rng = _rng(90 + D + N)
    nb = 5
    X = rng.standard_normal((nb, N, D)).astype(dtype)  # [N, D] row-major == D x N ColVecs
    mw = rng.standard_normal((nb, D)).astype(dtype)
    dpr = np.exp(0.5 * rng.standard_normal((nb, D))).astype(dtype)
    if prior == "diagonal":
        Lw_arg, pk, ldl, strideLw = dpr, a.PRIOR_DIAGONAL, 1, D
        Lw_dense = [np.diag(dpr[b].astype(np.float64)) for b in range(nb)]
    else:
        Bm = rng.standard_normal((nb, D, D)) / np.sqrt(D)
        Lw_dense = [Bm[b] @ Bm[b].T + np.eye(D) for b in range(nb)]
        if prior == "dense":
            Lw_arg = np.stack([np.asfortranarray(Lw_dense[b]).T for b in range(nb)]).astype(dtype)  # [b] = column-major D x D
            Lw_dense = [np.asarray(Lw_arg[b].T, dtype=np.float64) for b in range(nb)]
            Lw_dense = [np.triu(M) + np.triu(M, 1).T for M in Lw_dense]  # the upper triangle is what is read
            pk = a.PRIOR_DENSE
        else:
            U = [O.chol_upper(Lw_dense[b]).astype(dtype) for b in range(nb)]
            Lw_arg = np.stack([U[b].T.copy() for b in range(nb)])  # column-major upper factor
            Lw_dense = [U[b].astype(np.float64).T @ U[b].astype(np.float64) for b in range(nb)]
            pk = a.PRIOR_UPPER_FACTOR
        ldl, strideLw = D, D * D
    s = (np.exp(0.4 * rng.standard_normal((nb, max(N, 1)))) if noise == "diagonal" else np.full((nb, 1), 0.3)).astype(dtype)
    y = rng.standard_normal((nb, max(N, 1))).astype(dtype)
    mw_post = np.zeros((nb, D), dtype=dtype)
    T_post = np.zeros((nb, D, D), dtype=dtype)
    Lw_post = np.zeros((nb, D, D), dtype=dtype)
    lp = np.zeros(nb)
    info = np.full(nb, 9, dtype=np.int32)
    kind = a.NOISE_DIAGONAL if noise == "diagonal" else a.NOISE_ISOTROPIC
    h.posterior_batched(dtype, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, max(N, 1), kind, s, s.shape[1], pk,
                        mw, D, Lw_arg, ldl, strideLw, mw_post, D, T_post, D, D * D, Lw_post, D, D * D, lp, info)
    assert info.tolist() == [0] * nb
    rtol = 1e-10
    for b in range(nb):
        Xb = X[b].T.astype(np.float64)
        sb = (s[b, :N] if noise == "diagonal" else np.full(N, s[b, 0])).astype(np.float64)
        Lb = np.diag(Lw_dense[b]) if prior == "diagonal" else Lw_dense[b]
        if dtype == np.float32:
            Lb32 = dpr[b] if prior == "diagonal" else np.asarray(Lb, dtype=np.float32)
            _assert_fp32_within_lapack(mw[b], Lb32, np.asfortranarray(X[b].T), sb.astype(np.float32), y[b, :N], mw_post[b],
                                       Lw_post[b].T, lp[b], got_T=T_post[b].T, floor=(4e-6, 2e-6, 2e-6),
                                       what=f"wave kernel D={D} N={N} {noise} {prior}")
            continue
        mw_o, T_o, L_o = O.posterior_literal(mw[b].astype(np.float64), Lb, Xb, sb, y[b, :N].astype(np.float64))
        np.testing.assert_allclose(Lw_post[b].T, L_o, rtol=rtol, atol=rtol * np.abs(L_o).max())
        Tn = np.triu(T_post[b].T.astype(np.float64))
        np.testing.assert_allclose(Tn.T @ Tn, L_o, rtol=rtol, atol=rtol * np.abs(L_o).max())
        assert np.all(np.tril(T_post[b].T, -1) == 0)
        np.testing.assert_allclose(mw_post[b], mw_o, rtol=100 * rtol, atol=10 * rtol * np.abs(mw_o).max())
        lp_o = O.logpdf_literal(mw[b].astype(np.float64), Lb, Xb, sb, y[b, :N].astype(np.float64))
        assert lp[b] == pytest.approx(lp_o, rel=20 * rtol, abs=20 * rtol)


def test_wave_kernel_reports_bad_inputs_per_regressor(B):
    a = B._abi
    h = a.default_handle()
    rng = _rng(91)
    nb, D, N = 4, 64, 40
    X = rng.standard_normal((nb, N, D))
    mw = np.zeros((nb, D))
    dpr = np.ones((nb, D))
    dpr[1, 10] = -1.0                       # prior precision not positive definite: info = 11 (reference :78)
    s = np.exp(0.2 * rng.standard_normal((nb, N)))
    s[2, 17] = 0.0                          # variance not positive: info = 18 (reference :79)
    y = rng.standard_normal((nb, N))
    mw_post = np.zeros((nb, D)); T_post = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.zeros(nb, dtype=np.int32)
    h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_DIAGONAL, s, N, a.PRIOR_DIAGONAL,
                        mw, D, dpr, 1, D, mw_post, D, T_post, D, D * D, None, D, D * D, lp, info)
    assert info.tolist() == [0, 11, 18, 0]
    assert np.isnan(lp[1]) and np.isnan(lp[2]) and np.isfinite(lp[0]) and np.isfinite(lp[3])
    lp_o = O.logpdf_literal(mw[3], dpr[3], X[3].T, s[3], y[3])
    assert lp[3] == pytest.approx(lp_o, rel=1e-10)


def test_wave_kernel_is_bitwise_reproducible(B):
    # fixed accumulation order, no atomics: the same inputs give the same bits, whatever wave of whatever CU picks a regressor up
    import torch

    a = B._abi
    h = a.default_handle()
    dev = torch.device("cuda:0")
    nb, D, N = 3 * 2048 + 77, 64, 333  # four rounds of the kernel's grid-stride loop (the grid is capped at 2048 waves)
    g = torch.Generator(device=dev).manual_seed(5)
    X = torch.randn((nb, N, D), generator=g, dtype=torch.float64, device=dev)
    y = torch.randn((nb, N), generator=g, dtype=torch.float64, device=dev)
    s = torch.exp(0.3 * torch.randn((nb, N), generator=g, dtype=torch.float64, device=dev))
    mw = torch.randn((nb, D), generator=g, dtype=torch.float64, device=dev)
    dpr = torch.ones((D,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()  # the inputs were produced on torch's stream, the handle launches on its own
    outs = []
    for _ in range(2):
        mp = torch.empty((nb, D), dtype=torch.float64, device=dev)
        Tp = torch.empty((nb, D, D), dtype=torch.float64, device=dev)
        lp = torch.empty((nb,), dtype=torch.float64, device=dev)
        info = torch.empty((nb,), dtype=torch.int32, device=dev)
        h.posterior_batched(np.float64, a.MEM_DEVICE, a.LAYOUT_COLVECS, nb, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N, a.NOISE_DIAGONAL,
                            s.data_ptr(), N, a.PRIOR_DIAGONAL, mw.data_ptr(), D, dpr.data_ptr(), 1, 0, mp.data_ptr(), D, Tp.data_ptr(), D,
                            D * D, None, D, D * D, lp.data_ptr(), info.data_ptr())
        torch.cuda.synchronize()
        assert int(info.abs().sum().item()) == 0
        outs.append((mp, Tp, lp))
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    # ... and the right bits in EVERY round of the grid-stride loop: regressors picked up by a wave as its 1st, 2nd, 3rd and
    # 4th against the oracle's literal op sequence
    mp, Tp, lp = outs[0]
    for b in (0, 2047, 2048, 2049, 4095, 4096 + 17, 3 * 2048, 3 * 2048 + 5, nb - 1):
        Xb = X[b].cpu().numpy().T
        mw_o, T_o, L_o = O.posterior_literal(mw[b].cpu().numpy(), np.ones(D), Xb, s[b].cpu().numpy(), y[b].cpu().numpy())
        lp_o = O.logpdf_literal(mw[b].cpu().numpy(), np.ones(D), Xb, s[b].cpu().numpy(), y[b].cpu().numpy())
        assert lp[b].item() == pytest.approx(lp_o, rel=1e-10), b
        np.testing.assert_allclose(mp[b].cpu().numpy(), mw_o, rtol=1e-8, atol=1e-10)
        Tn = np.triu(Tp[b].cpu().numpy().T)
        np.testing.assert_allclose(Tn.T @ Tn, L_o, rtol=1e-10, atol=1e-10 * np.abs(L_o).max())


# ---- the int8-sliced Gram of the headline shape (blr_fused_i8.hpp): D = 128, fp64, aligned ColVecs, isotropic noise, diagonal prior ----
def _i8_case(rng, nb, N, kind):
    D = 128
    X = rng.standard_normal((nb, N, D))  # [N, D] row-major == D x N ColVecs
    rowscale = np.ones(D)
    if kind == "scales":      # rows from 2^-12 to 2^12: every row has its own power-of-two bound (y stays O(1): the weights scale back)
        rowscale = np.ldexp(1.0, (np.arange(D) % 7) * 4 - 12)
        X *= rowscale[None, None, :]
    elif kind == "zero_row":  # a feature that is identically zero, and one that only wakes up late
        X[:, :, 17] = 0.0
        X[1::2, : N // 2, 40] = 0.0
    elif kind == "tiny":      # denormal-range and zero entries mixed with ordinary ones
        X[:, ::3, 9] *= 1e-310
    w = rng.standard_normal((nb, D)) / rowscale[None, :]
    y = np.einsum("bnd,bd->bn", X, w) + np.sqrt(0.1) * rng.standard_normal((nb, N))
    if kind == "outlier":     # one entry far above its row's capacity, in the middle of the stream: wrapped digits, repaired in fp64 at the
        X[::2, N // 2, 5] = 1.0e3  # hand-over (y is left as it was: an outlier in y would only make the evidence ill-conditioned for everybody)
    return X, y


@pytest.mark.parametrize("kind", ["gauss", "scales", "outlier", "zero_row", "tiny"])
@pytest.mark.parametrize("N", [512, 543, 4096, 4127])  # (543, 4127: a last partial block of 31 columns, added in fp64 at the hand-over)
@pytest.mark.parametrize("prior_mean", [False, True])
def test_i8_gram_path_vs_oracle_and_fp64_kernel(B, opt, kind, N, prior_mean):
    # the SAME call with the fast path on (default) and off (NO_I8_GRAM: fused_small_kernel, the fp64 matrix pipe), both against
    # the oracle's direct form at the fp64 tolerances of test_c2_shape_fp64; the two device paths must agree far inside them:
    # the digit splitting keeps 48 bits of every entry relative to its row's bound and every digit-pair product down to
    # 2^-52 of the result (Gram entries within 3e-14 of sqrt(G_ii G_jj)).  Outliers / non-finite input send a regressor back to the fp64 kernel: then
    # the bits must be the fp64 kernel's.
    a = B._abi
    h = a.default_handle()
    rng = _rng(4200 + N + len(kind))
    nb, D = 6, 128
    X, y = _i8_case(rng, nb, N, kind)
    dpr = np.exp(0.3 * rng.standard_normal((nb, D)))
    mw = np.zeros((nb, D))
    if prior_mean:  # test/test_utils.jl:6: mw = randn(D) (in the units of the rows); on the fast path it enters through the finished A
        rowscale = np.ldexp(1.0, (np.arange(D) % 7) * 4 - 12) if kind == "scales" else np.ones(D)
        mw = rng.standard_normal((nb, D)) / rowscale[None, :]
    s = np.array([0.1])

    def run():
        mp = np.zeros((nb, D)); Tp = np.zeros((nb, D, D)); Ap = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_DIAGONAL,
                            mw, D, dpr, 1, D, mp, D, Tp, D, D * D, Ap, D, D * D, lp, info)
        return mp, Tp, Ap, lp, info

    h.reset_stats()
    fast = run()
    handed_back = h.get_stat("i8_handed_back")
    again = run()
    for u, v in zip(fast, again):
        np.testing.assert_array_equal(u, v)  # fixed accumulation order on the fast path too
    opt("NO_I8_GRAM", "1")
    slow = run()
    assert fast[4].tolist() == [0] * nb and slow[4].tolist() == [0] * nb
    for b in range(nb):
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw[b], dpr[b], X[b].T, 0.1, y[b])
        scale = np.abs(A_o).max()
        # evidence: the documented 1e-10 is the contract (header); the check here is sharper and scales with what cancels in
        # delta'delta / s - |u|^2 (well-explained data: terms ~1e3 times the evidence) -- 1e-11 of the evidence + 1e-14 of delta'delta / s
        dlt = y[b] - X[b] @ mw[b]
        lp_tol = 1e-11 * abs(lp_o) + 1e-14 * float(dlt @ dlt) / 0.1
        for mp, Tp, Ap, lp, _ in (fast, slow):
            assert abs(lp[b] - lp_o) <= lp_tol
            np.testing.assert_allclose(mp[b] * np.sqrt(np.diag(A_o)), mw_o * np.sqrt(np.diag(A_o)), rtol=1e-8, atol=1e-9 * np.abs(mw_o * np.sqrt(np.diag(A_o))).max())
            # entries of A against the scale of their row and column (rows of very different magnitude: "scales")
            dA = np.sqrt(np.diag(A_o))
            assert (np.abs(Ap[b] - A_o) / np.outer(dA, dA)).max() <= 1e-12
            Tn = np.triu(Tp[b].T)
            assert (np.abs(Tn.T @ Tn - A_o) / np.outer(dA, dA)).max() <= 1e-10
        assert abs(fast[3][b] - slow[3][b]) <= lp_tol
        dsc = np.sqrt(np.diag(O.posterior_logpdf_direct(mw[b], dpr[b], X[b].T, 0.1, y[b])[2]))
        assert np.abs((fast[0][b] - slow[0][b]) * dsc).max() <= 1e-10 * np.abs(slow[0][b] * dsc).max()
    # An entry beyond its row's capacity wraps around in the 48-bit integer and is put right in fp64 at the hand-over ("outlier": a
    # 1e3 among N(0,1) entries, 100 capacities out -- checked against the oracle above like everything else); only a row that wakes up
    # after the columns its scale came from ("zero_row", every other regressor) marks more blocks than are repaired and goes back.
    if kind in ("gauss", "scales", "outlier", "tiny"):
        assert handed_back == 0
    if kind == "zero_row":  # (N / 64 marked blocks against at most one in 16 repaired: handed back)
        assert handed_back == nb // 2
        for b in range(1, nb, 2):  # the fp64 kernel's bits
            assert fast[3][b] == slow[3][b]
            np.testing.assert_array_equal(fast[0][b], slow[0][b])
            np.testing.assert_array_equal(fast[1][b], slow[1][b])


def test_i8_gram_prior_mean_under_a_strong_prior(B, opt):
    # A prior mean enters the int8 route through the finished matrix: b = X y / s - (G / s) mw.  G / s must be the data term itself,
    # not A - Lw: with prior precisions up to 2^40 times the data term (rows scaled by 2^-20) the difference keeps no digit of G
    # (found by tools/fuzz_round4.py: evidence off by a factor 150).  Both device routes against the oracle and against each other.
    a = B._abi
    h = a.default_handle()
    rng = _rng(4400)
    nb, D, N = 4, 128, 1024
    scale = np.ldexp(1.0, rng.integers(-20, 21, size=D))
    X = rng.standard_normal((nb, N, D)) * scale[None, None, :]
    w = rng.standard_normal((nb, D)) / scale[None, :]
    y = np.einsum("bnd,bd->bn", X, w) + np.sqrt(0.1) * rng.standard_normal((nb, N))
    mw = rng.standard_normal((nb, D)) / scale[None, :]
    dpr = np.exp(0.3 * rng.standard_normal((nb, D))) / scale[None, :] ** 2
    s = np.array([0.1])

    def run():
        mp = np.zeros((nb, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_DIAGONAL,
                            mw, D, dpr, 1, D, mp, D, None, D, D * D, None, D, D * D, lp, info)
        return mp, lp, info

    fast = run()
    opt("NO_I8_GRAM", "1")
    slow = run()
    assert fast[2].tolist() == [0] * nb and slow[2].tolist() == [0] * nb
    for b in range(nb):
        mw_o, _, A_o, lp_o = O.posterior_logpdf_direct(mw[b], dpr[b], X[b].T, 0.1, y[b])
        dA = np.sqrt(np.diag(A_o))
        for mp, lp, _ in (fast, slow):
            assert lp[b] == pytest.approx(lp_o, rel=1e-10)
            np.testing.assert_allclose(mp[b] * dA, mw_o * dA, rtol=1e-7, atol=1e-8 * np.abs(mw_o * dA).max())
        assert fast[1][b] == pytest.approx(slow[1][b], rel=1e-10)


def test_i8_gram_path_hands_back_what_it_cannot_do(B, opt):
    # NaN / Inf in X, a bad noise variance, a non-positive prior entry: status, NaN evidence and untouched outputs exactly as the
    # fp64 kernel reports them (the fast path either reproduces the status or hands the regressor back); a prior mean stays on
    # the fast path
    a = B._abi
    h = a.default_handle()
    rng = _rng(4300)
    nb, D, N = 8, 128, 1024
    X, y = _i8_case(rng, nb, N, "gauss")
    dpr = np.ones((nb, D))
    mw = np.zeros((nb, D))
    mw[1] = rng.standard_normal(D)          # a prior mean
    X[2, 100, 3] = np.nan                   # NaN in the stream
    X[3, 900, 127] = np.inf                 # Inf late in the stream
    dpr[4, 77] = -1.0                       # prior not positive definite: info = 78
    X[5, 0, 0] = np.nan                     # NaN in the block the row bounds come from
    s = np.array([0.1])

    def run(svar):
        mp = np.full((nb, D), 7.0); Tp = np.full((nb, D, D), 7.0); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, svar, 0, a.PRIOR_DIAGONAL,
                            mw, D, dpr, 1, D, mp, D, Tp, D, D * D, None, D, D * D, lp, info)
        return mp, Tp, lp, info

    fast = run(s)
    opt("NO_I8_GRAM", "1")
    slow = run(s)
    opt("NO_I8_GRAM", None)
    assert fast[3].tolist() == slow[3].tolist()
    assert fast[3][4] == 78 and fast[3][0] == 0 and fast[3][1] == 0
    for b in range(nb):
        if b in (2, 3, 4, 5):  # handed back or failed before any arithmetic: identical bits / identical NaN pattern
            np.testing.assert_array_equal(fast[0][b], slow[0][b])
            np.testing.assert_array_equal(fast[1][b], slow[1][b])
            assert (fast[2][b] == slow[2][b]) or (np.isnan(fast[2][b]) and np.isnan(slow[2][b]))
        else:
            assert fast[2][b] == pytest.approx(slow[2][b], rel=1e-11)
            np.testing.assert_allclose(fast[0][b], slow[0][b], rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(fast[1][b], slow[1][b], rtol=1e-9, atol=1e-11)
    assert np.all(fast[0][4] == 7.0) and np.all(fast[1][4] == 7.0)  # failed regressor: outputs untouched
    bad = run(np.array([-0.5]))  # sigma^2 <= 0: PosDefException(1) at reference :79 for every regressor whose prior is fine
    assert bad[3].tolist() == [1, 1, 1, 1, 78, 1, 1, 1] and np.all(np.isnan(bad[2]))


@pytest.mark.parametrize("N", [512, 1055, 4096])
@pytest.mark.parametrize("prior_mean", [False, True])
def test_i8_gram_path_diagonal_noise(B, opt, N, prior_mean):
    # Diagonal noise on the int8 route: x / sqrt(s_n), y / sqrt(s_n) are what is sliced (i8_noise_prep_kernel supplies 1 / sqrt(s),
    # y / sqrt(s), sum log s).  Against the oracle, against the fp64 kernel (NO_I8_DIAG), bit-reproducible; a variance that is not
    # positive sends ITS regressor to the fp64 kernel, which reports the reference's error (:79).
    a = B._abi
    h = a.default_handle()
    rng = _rng(4600 + N + int(prior_mean))
    nb, D = 5, 128
    X, y = _i8_case(rng, nb, N, "gauss")
    s = np.exp(rng.standard_normal((nb, N)))  # variances over two orders of magnitude
    y = y + np.sqrt(s) * rng.standard_normal((nb, N))
    dpr = np.exp(0.3 * rng.standard_normal((nb, D)))
    mw = rng.standard_normal((nb, D)) if prior_mean else np.zeros((nb, D))

    def run(svar):
        mp = np.full((nb, D), 7.0); Tp = np.zeros((nb, D, D)); Ap = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_DIAGONAL, svar, N, a.PRIOR_DIAGONAL,
                            mw, D, dpr, 1, D, mp, D, Tp, D, D * D, Ap, D, D * D, lp, info)
        return mp, Tp, Ap, lp, info

    fast = run(s)
    again = run(s)
    for u, v in zip(fast, again):
        np.testing.assert_array_equal(u, v)
    opt("NO_I8_DIAG", "1")
    slow = run(s)
    opt("NO_I8_DIAG", None)
    assert fast[4].tolist() == [0] * nb and slow[4].tolist() == [0] * nb
    for b in range(nb):
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw[b], dpr[b], X[b].T, s[b], y[b])
        dA = np.sqrt(np.diag(A_o))
        for mp, Tp, Ap, lp, _ in (fast, slow):
            assert lp[b] == pytest.approx(lp_o, rel=1e-10)
            np.testing.assert_allclose(mp[b] * dA, mw_o * dA, rtol=1e-8, atol=1e-9 * np.abs(mw_o * dA).max())
            assert (np.abs(Ap[b] - A_o) / np.outer(dA, dA)).max() <= 1e-12
        assert fast[3][b] == pytest.approx(slow[3][b], rel=1e-10)
    sbad = s.copy()
    sbad[2, N // 2] = -1.0
    f2, s2 = run(sbad), None
    opt("NO_I8_DIAG", "1")
    s2 = run(sbad)
    assert f2[4].tolist() == s2[4].tolist() and f2[4][2] != 0 and f2[4][0] == 0
    assert np.all(f2[0][2] == 7.0)  # failed regressor: outputs untouched


@pytest.mark.parametrize("N", [512, 1055])
@pytest.mark.parametrize("prior_mean", [False, True])
def test_i8_gram_path_factor_prior(B, opt, N, prior_mean):
    # A prior given by its upper factor U (PDMat / a carried-forward posterior) on the int8 route: U'U joins the finished data matrix
    # in fp64 at the hand-over, after the prior-mean terms have been taken from the pure data matrix.  Against the oracle and the
    # fp64 kernel (NO_I8_FACTOR); in place (T_post == Lw, mw_post == mw) as blr_update_factor_* uses it; a factor with a
    # non-positive diagonal entry reports its index.
    a = B._abi
    h = a.default_handle()
    rng = _rng(4700 + N + int(prior_mean))
    nb, D = 4, 128
    X, y = _i8_case(rng, nb, N, "gauss")
    mw = rng.standard_normal((nb, D)) if prior_mean else np.zeros((nb, D))
    Lw = np.empty((nb, D, D)); Uc = np.empty((nb, D, D))
    for b in range(nb):
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        Lw[b] = Bm @ Bm.T + np.exp(rng.standard_normal()) * np.eye(D)
        Uc[b] = O.chol_upper(Lw[b]).T  # column-major storage of U
    s = np.array([0.1])

    def run(U_in, mw_in):
        mp = np.full((nb, D), 7.0); Tp = np.zeros((nb, D, D)); Ap = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_UPPER_FACTOR,
                            mw_in, D, U_in, D, D * D, mp, D, Tp, D, D * D, Ap, D, D * D, lp, info)
        return mp, Tp, Ap, lp, info

    fast = run(Uc, mw)
    opt("NO_I8_FACTOR", "1")
    slow = run(Uc, mw)
    opt("NO_I8_FACTOR", None)
    assert fast[4].tolist() == [0] * nb and slow[4].tolist() == [0] * nb
    for b in range(nb):
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw[b], Lw[b], X[b].T, 0.1, y[b])
        dA = np.sqrt(np.diag(A_o))
        for mp, Tp, Ap, lp, _ in (fast, slow):
            assert lp[b] == pytest.approx(lp_o, rel=1e-10)
            np.testing.assert_allclose(mp[b] * dA, mw_o * dA, rtol=1e-8, atol=1e-9 * np.abs(mw_o * dA).max())
            assert (np.abs(Ap[b] - A_o) / np.outer(dA, dA)).max() <= 1e-12
            Tn = np.triu(Tp[b].T)
            assert (np.abs(Tn.T @ Tn - A_o) / np.outer(dA, dA)).max() <= 1e-10
        assert fast[3][b] == pytest.approx(slow[3][b], rel=1e-10)
    Ubad = Uc.copy()
    Ubad[1, 40, 40] = -Ubad[1, 40, 40]
    fb = run(Ubad, mw)
    assert fb[4].tolist() == [0, 41, 0, 0] and np.all(fb[0][1] == 7.0)


@pytest.mark.parametrize("N,noise", [(512, "iso"), (1055, "diag"), (4096, "iso")])
@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("prior_mean", [False, True])
def test_i8_gram_path_dense_prior(B, opt, N, noise, shared, prior_mean):
    # A dense prior precision -- what the reference's own toy problems use (test/test_utils.jl:6-8: Lw = B B' + I; `_cholesky(blr.Lw)` at
    # src/bayesian_linear_regression.jl:78) -- on the int8 route: logdet Lw and the positive-definiteness check come from one blocked
    # Cholesky per prior before the launch (ONE for a prior the batch shares: strideLw = 0), Lw itself joins the finished matrix at the
    # hand-over after the prior-mean terms.  Only the UPPER triangle of the caller's matrix is read (the lower one holds NaN here).
    # Against the oracle and the fp64 kernel (NO_I8_DENSE); a prior that is not positive definite reports ITS leading minor.
    a = B._abi
    h = a.default_handle()
    rng = _rng(5000 + N + 2 * int(shared) + int(prior_mean))
    nb, D = 4, 128
    X, y = _i8_case(rng, nb, N, "gauss")
    mw = rng.standard_normal((nb, D)) if prior_mean else np.zeros((nb, D))
    npri = 1 if shared else nb
    Lw = np.empty((npri, D, D)); Lin = np.empty((npri, D, D))
    for b in range(npri):
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        Lw[b] = Bm @ Bm.T + np.exp(rng.standard_normal()) * np.eye(D)
        Lin[b] = np.triu(Lw[b]).T + np.tril(np.full((D, D), np.nan), -1).T  # column-major: upper triangle valid, lower one NaN
    svar = np.exp(0.5 * rng.standard_normal((nb, N))) * 0.1 if noise == "diag" else np.array([0.1])
    nk, sstr = (a.NOISE_DIAGONAL, N) if noise == "diag" else (a.NOISE_ISOTROPIC, 0)

    def run(L_in):
        mp = np.full((nb, D), 7.0); Tp = np.zeros((nb, D, D)); Ap = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, nk, svar, sstr, a.PRIOR_DENSE,
                            mw, D, L_in, D, 0 if shared else D * D, mp, D, Tp, D, D * D, Ap, D, D * D, lp, info)
        return mp, Tp, Ap, lp, info

    fast = run(Lin)
    assert h.last_route() == "fused_i8_kernel"
    again = run(Lin)
    for u, v in zip(fast, again):
        np.testing.assert_array_equal(u, v)
    opt("NO_I8_DENSE", "1")
    slow = run(Lin)
    assert h.last_route().startswith("fused_small_kernel<double, 8,")
    opt("NO_I8_DENSE", None)
    assert fast[4].tolist() == [0] * nb and slow[4].tolist() == [0] * nb
    for b in range(nb):
        sb = svar[b] if noise == "diag" else 0.1
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw[b], Lw[0 if shared else b], X[b].T, sb, y[b])
        dA = np.sqrt(np.diag(A_o))
        for mp, Tp, Ap, lp, _ in (fast, slow):
            assert lp[b] == pytest.approx(lp_o, rel=1e-10)
            np.testing.assert_allclose(mp[b] * dA, mw_o * dA, rtol=1e-8, atol=1e-9 * np.abs(mw_o * dA).max())
            assert (np.abs(Ap[b] - A_o) / np.outer(dA, dA)).max() <= 1e-12
            Tn = np.triu(Tp[b].T)
            assert (np.abs(Tn.T @ Tn - A_o) / np.outer(dA, dA)).max() <= 1e-10
        assert fast[3][b] == pytest.approx(slow[3][b], rel=1e-10)
    Lbad = Lin.copy()
    Lbad[npri - 1, 40, 40] = -50.0  # the leading 41 x 41 minor is the first that is not positive definite
    fb = run(Lbad)
    sbad = run(Lbad) if False else None
    expect = [41] * nb if shared else [0] * (nb - 1) + [41]
    assert fb[4].tolist() == expect
    assert np.all(fb[0][nb - 1] == 7.0) and np.isnan(fb[3][nb - 1])


@pytest.mark.parametrize("kind", ["gauss", "scales", "outlier"])
@pytest.mark.parametrize("N,noise,prior", [(512, "iso", "diag"), (543, "diag", "diag"), (4127, "iso", "factor"), (1055, "diag", "factor")])
def test_i8_gram_path_rowvecs(B, opt, kind, N, noise, prior):
    # RowVecs inputs (N x D column-major, lda >= N: src/bayesian_linear_regression.jl's X' handed over as stored) on the int8 route:
    # the stream gathers four feature rows per LDS-DMA piece instead of four columns, everything after the raw block is the ColVecs
    # code -- so the RowVecs call must give the BITS of the ColVecs call on the transposed copy (prior mean, ragged N, diagonal noise,
    # factor prior, a handed-back regressor included), and the oracle's numbers.  Rows that are not 16-byte aligned (odd lda) and
    # NO_I8_ROWVECS take the fp64 kernel: same numbers at its tolerances.
    a = B._abi
    h = a.default_handle()
    rng = _rng(4800 + N + len(kind))
    nb, D = 4, 128
    X, y = _i8_case(rng, nb, N, kind)
    mw = rng.standard_normal((nb, D)) / (np.ldexp(1.0, (np.arange(D) % 7) * 4 - 12) if kind == "scales" else 1.0)
    svar = np.exp(0.5 * rng.standard_normal((nb, N))) * 0.1 if noise == "diag" else np.array([0.1])
    if prior == "factor":
        Lw = np.empty((nb, D, D)); Lin = np.empty((nb, D, D))
        for b in range(nb):
            Bm = rng.standard_normal((D, D)) / np.sqrt(D)
            Lw[b] = Bm @ Bm.T + np.eye(D)
            Lin[b] = O.chol_upper(Lw[b]).T
        pk, ldl, sl = a.PRIOR_UPPER_FACTOR, D, D * D
    else:
        Lin = np.exp(0.3 * rng.standard_normal((nb, D)))
        Lw = Lin
        pk, ldl, sl = a.PRIOR_DIAGONAL, 1, D
    nk, ss = (a.NOISE_DIAGONAL, N) if noise == "diag" else (a.NOISE_ISOTROPIC, 0)

    def run(layout, Xin, ldx, strideX):
        mp = np.full((nb, D), 7.0); Tp = np.zeros((nb, D, D)); Ap = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, layout, nb, D, N, Xin, ldx, strideX, y, N, nk, svar, ss, pk,
                            mw, D, Lin, ldl, sl, mp, D, Tp, D, D * D, Ap, D, D * D, lp, info)
        return mp, Tp, Ap, lp, info

    def rowvecs(ld):
        Xr = np.full((nb, D, ld), 3.0e5)  # (the padding below row N must never be read: it would break every row bound)
        Xr[:, :, :N] = X.transpose(0, 2, 1)
        return Xr

    col = run(a.LAYOUT_COLVECS, X, D, N * D)
    ld = N + 2 + (N & 1)
    row = run(a.LAYOUT_ROWVECS, rowvecs(ld), ld, D * ld)
    assert col[4].tolist() == [0] * nb
    if kind != "outlier":  # (a handed-back regressor is redone by the fp64 kernel of ITS layout: other accumulation order)
        for u, v in zip(col, row):
            np.testing.assert_array_equal(u, v)
    else:
        for b in range(1, nb, 2):
            for u, v in zip(col, row):
                np.testing.assert_array_equal(u[b], v[b])
    odd = run(a.LAYOUT_ROWVECS, rowvecs(ld + 1), ld + 1, D * (ld + 1))
    opt("NO_I8_ROWVECS", "1")
    slow = run(a.LAYOUT_ROWVECS, rowvecs(ld), ld, D * ld)
    opt("NO_I8_ROWVECS", None)
    for u, v in zip(odd, slow):
        np.testing.assert_array_equal(u, v)  # both on the fp64 RowVecs kernel
    for b in range(nb):
        sv = svar[b] if noise == "diag" else 0.1
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw[b], Lw[b], X[b].T, sv, y[b])
        dA = np.sqrt(np.diag(A_o))
        for mp, Tp, Ap, lp, info in (row, slow):
            assert info[b] == 0
            assert lp[b] == pytest.approx(lp_o, rel=1e-10)
            np.testing.assert_allclose(mp[b] * dA, mw_o * dA, rtol=1e-8, atol=1e-9 * np.abs(mw_o * dA).max())
            assert (np.abs(Ap[b] - A_o) / np.outer(dA, dA)).max() <= 1e-12
            Tn = np.triu(Tp[b].T)
            assert (np.abs(Tn.T @ Tn - A_o) / np.outer(dA, dA)).max() <= 1e-10


def test_i8_gram_tail_columns_nonfinite_go_back_to_the_fp64_kernel(B, opt):
    # N = 1055: 32 whole k-steps through the int8 stream, 31 columns added in fp64 at the hand-over.  A NaN / Inf in THOSE columns
    # hands the regressor back like one in the stream does: status and bits of the fp64 kernel.
    a = B._abi
    h = a.default_handle()
    rng = _rng(4500)
    nb, D, N = 4, 128, 1055
    X, y = _i8_case(rng, nb, N, "gauss")
    X[1, N - 3, 9] = np.nan
    X[2, N - 31, 127] = np.inf
    dpr = np.ones((nb, D)); mw = np.zeros((nb, D)); s = np.array([0.1])

    def run():
        mp = np.full((nb, D), 7.0); Tp = np.full((nb, D, D), 7.0); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_DIAGONAL,
                            mw, D, dpr, 1, D, mp, D, Tp, D, D * D, None, D, D * D, lp, info)
        return mp, Tp, lp, info

    fast = run()
    opt("NO_I8_GRAM", "1")
    slow = run()
    assert fast[3].tolist() == slow[3].tolist()
    for b in (1, 2):
        np.testing.assert_array_equal(fast[0][b], slow[0][b])
        np.testing.assert_array_equal(fast[1][b], slow[1][b])
        assert (fast[2][b] == slow[2][b]) or (np.isnan(fast[2][b]) and np.isnan(slow[2][b]))
    for b in (0, 3):
        assert fast[3][b] == 0 and fast[2][b] == pytest.approx(slow[2][b], rel=1e-11)
        np.testing.assert_allclose(fast[0][b], slow[0][b], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("N", [8192, 16384, 16415])
@pytest.mark.parametrize("kind", ["gauss", "pow2"])
def test_i8_gram_path_at_its_largest_N(B, opt, N, kind):
    # VERDICT r4 weak #2b: kI8MaxN = 16384 whole columns (+ 31 in fp64) is where the int32 accumulators of the digit-pair products have
    # their smallest head-room.  "pow2": every entry is +-2^e of its row, so every unsigned lower digit is 0 and is stored as -128 --
    # each of a group's products then adds 16384 per column with the same sign: the largest sums the accumulators can see.
    # Against the oracle's direct form AND the fp64 kernel on the same inputs (NO_I8_GRAM), at the tolerances of
    # test_i8_gram_path_vs_oracle_and_fp64_kernel.
    a = B._abi
    h = a.default_handle()
    rng = _rng(4600 + N + len(kind))
    nb, D = 3, 128
    if kind == "gauss":
        X, y = _i8_case(rng, nb, N, "gauss")
    else:
        rowexp = rng.integers(-6, 7, size=D)
        X = np.ldexp(rng.choice([-1.0, 1.0], size=(nb, N, D)), rowexp[None, None, :])
        w = rng.standard_normal((nb, D)) / np.ldexp(1.0, rowexp)[None, :]
        y = np.einsum("bnd,bd->bn", X, w) + np.sqrt(0.1) * rng.standard_normal((nb, N))
    dpr = np.exp(0.3 * rng.standard_normal((nb, D)))
    mw = np.zeros((nb, D))
    s = np.array([0.1])

    def run():
        mp = np.zeros((nb, D)); Ap = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_DIAGONAL,
                            mw, D, dpr, 1, D, mp, D, None, D, D * D, Ap, D, D * D, lp, info)
        return mp, Ap, lp, info

    h.reset_stats()
    fast = run()
    assert h.last_route() == "fused_i8_kernel"
    assert h.get_stat("i8_regressors") == nb and h.get_stat("i8_handed_back") == 0
    opt("NO_I8_GRAM", "1")
    slow = run()
    assert h.last_route().startswith("fused_small_kernel<double, 8,")
    assert fast[3].tolist() == [0] * nb and slow[3].tolist() == [0] * nb
    for b in range(nb):
        mw_o, _, A_o, lp_o = O.posterior_logpdf_direct(mw[b], dpr[b], X[b].T, 0.1, y[b])
        dA = np.sqrt(np.diag(A_o))
        # (the evidence is y'y / s - |u|^2 + ..., two terms that cancel ~500-fold here and are each summed over N >= 8192 terms in a
        # different order by the oracle and by either kernel: agreement is 1e-14 of THOSE terms -- "pow2" has bit-identical Gram
        # matrices on both device routes and they still differ by that much -- so the tolerance carries a term in y'y / s)
        lp_tol = 1e-11 * abs(lp_o) + 1e-13 * float(y[b] @ y[b]) / 0.1
        for mp, Ap, lp, _ in (fast, slow):
            assert abs(lp[b] - lp_o) <= lp_tol
            assert (np.abs(Ap[b] - A_o) / np.outer(dA, dA)).max() <= 1e-12
            np.testing.assert_allclose(mp[b] * dA, mw_o * dA, rtol=1e-8, atol=1e-9 * np.abs(mw_o * dA).max())
        assert (np.abs(fast[1][b] - slow[1][b]) / np.outer(dA, dA)).max() <= 1e-13
        assert abs(fast[2][b] - slow[2][b]) <= lp_tol


@pytest.mark.parametrize("tails", ["gauss", "student_t3", "lognormal"])
def test_i8_gram_path_heavy_tails_are_counted_and_fall_back(B, opt, tails):
    # VERDICT r4 weak #2c: the row bounds of the int8 route come from the first 96 columns; heavy-tailed features outgrow them, the
    # regressor is handed back and pays BOTH kernels.  blr_get_stat("i8_handed_back") makes that visible, and a batch beyond
    # kI8ProbeMin = 4096 regressors (here: option I8_PROBE_MIN = 1024) starts with a probe slice of 256: when more than a quarter of it was handed back, the rest of the
    # call goes to the fp64 kernel directly (NO_I8_FALLBACK switches that off).  Results are checked on both routes either way.
    a = B._abi
    h = a.default_handle()
    rng = _rng(4700 + len(tails))
    nb, D, N = 1280, 128, 512
    if tails == "gauss":
        X = rng.standard_normal((nb, N, D))
    elif tails == "student_t3":
        X = rng.standard_t(3.0, size=(nb, N, D))
    else:
        X = np.exp(1.5 * rng.standard_normal((nb, N, D))) * rng.choice([-1.0, 1.0], size=(nb, N, D))
    w = rng.standard_normal((nb, D)) / np.sqrt(D)
    y = np.einsum("bnd,bd->bn", X, w) + np.sqrt(0.1) * rng.standard_normal((nb, N))
    dpr = np.ones(D); mw = np.zeros(D); s = np.array([0.1])

    def run():
        mp = np.zeros((nb, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_DIAGONAL,
                            mw, 0, dpr, 1, 0, mp, D, None, D, D * D, None, D, D * D, lp, info)
        return mp, lp, info

    opt("I8_PROBE_MIN", "1024")  # (the library probes batches beyond 4096 regressors: 2 GB of host inputs at this N; lowered for the test)
    h.reset_stats()
    fast = run()
    sent, back = h.get_stat("i8_regressors"), h.get_stat("i8_handed_back")
    opt("NO_I8_FALLBACK", "1")
    h.reset_stats()
    fast_nofb = run()
    back_nofb = h.get_stat("i8_handed_back")
    opt("NO_I8_GRAM", "1")
    slow = run()
    print(f"\n[{tails}] handed back {back_nofb} of {nb} regressors ({100.0 * back_nofb / nb:.1f} %); with the probe slice steering: {back}")
    assert sent == nb
    assert fast[2].tolist() == [0] * nb and fast_nofb[2].tolist() == [0] * nb and slow[2].tolist() == [0] * nb
    if tails == "gauss":
        assert back == 0 and back_nofb == 0
    else:
        assert back_nofb > 0
        if 4 * back_nofb > nb * 1.2:  # clearly above a quarter everywhere: the probe slice must have sent the rest to the fp64 kernel
            assert back >= nb - 256
    # every regressor, whichever route finished it, against the fp64 kernel on the same inputs; a sample against the oracle
    for r in (fast, fast_nofb):
        np.testing.assert_allclose(r[1], slow[1], rtol=1e-10)
        assert np.abs(r[0] - slow[0]).max() <= 1e-8 * np.abs(slow[0]).max()
    for b in (0, 255, 256, 700, nb - 1):
        mw_o, _, _, lp_o = O.posterior_logpdf_direct(mw, dpr, X[b].T, 0.1, y[b])
        assert fast[1][b] == pytest.approx(lp_o, rel=1e-10)
        np.testing.assert_allclose(fast[0][b], mw_o, rtol=1e-7, atol=1e-9)


def test_i8_gram_prior_mean_that_explains_the_data(B, opt):
    # the int8 route folds a prior mean in AFTER the stream -- delta'delta / s = y'y / s - 2 mw'Xy / s + mw'(G / s) mw --
    # and when mw already explains the data (a carried-forward posterior conditioned on more of the same stream, s small) that
    # is a difference of numbers 1e10 times its size.  The reference (:82) and the fp64 kernel form delta = y - X'mw first.  The
    # route now hands such a regressor back (three digits of cancellation are the most it keeps): evidence at the fp64 tolerance.
    a = B._abi
    h = a.default_handle()
    rng = _rng(4800)
    nb, D, N = 4, 128, 1024
    X = rng.standard_normal((nb, N, D))
    mw = rng.standard_normal((nb, D))
    sig2 = 1e-8
    y = np.einsum("bnd,bd->bn", X, mw) + np.sqrt(sig2) * rng.standard_normal((nb, N))
    y[3] = rng.standard_normal(N)  # (a regressor whose prior mean explains nothing stays on the fast path)
    dpr = np.ones((nb, D)); s = np.array([sig2])

    def run():
        mp = np.zeros((nb, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_DIAGONAL,
                            mw, D, dpr, 1, D, mp, D, None, D, D * D, None, D, D * D, lp, info)
        return mp, lp, info

    h.reset_stats()
    fast = run()
    assert h.get_stat("i8_handed_back") == 3
    opt("NO_I8_GRAM", "1")
    slow = run()
    assert fast[2].tolist() == [0] * nb
    for b in range(nb):
        mw_o, _, _, lp_o = O.posterior_logpdf_direct(mw[b], dpr[b], X[b].T, sig2, y[b])
        assert fast[1][b] == pytest.approx(lp_o, rel=1e-10)
        assert slow[1][b] == pytest.approx(lp_o, rel=1e-10)
        np.testing.assert_allclose(fast[0][b], mw_o, rtol=1e-7, atol=1e-9)
    for b in range(3):  # handed back: the fp64 kernel's bits
        assert fast[1][b] == slow[1][b]
        np.testing.assert_array_equal(fast[0][b], slow[0][b])


def test_switching_streams_between_large_d_calls(B):
    # the arrival counters of panel_chain_kernel alternate between two banks per handle and a launch clears its successor's
    # bank -- two streams on one handle could have a launch clear a bank its successor is already counting in.  blr_set_stream now drains
    # the old stream and re-arms both banks: D > 128 updates (an odd number of panel launches each: D = 384 has three) issued alternately on
    # the handle's own stream, a torch side stream and the null stream must return the bits of the same call on one stream.
    import torch

    a = B._abi
    h = a.default_handle()
    dev = torch.device("cuda:0")
    rng = _rng(5100)
    nb, D, N = 3, 384, 500
    X = rng.standard_normal((nb, N, D)); y = rng.standard_normal((nb, N)); mw = rng.standard_normal((nb, D))
    dpr = np.exp(0.3 * rng.standard_normal((nb, D))); s = np.array([0.3])

    def run():
        mp = np.zeros((nb, D)); Tp = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_DIAGONAL,
                            mw, D, dpr, 1, D, mp, D, Tp, D, D * D, None, D, D * D, lp, info)
        assert info.tolist() == [0] * nb
        return mp, Tp, lp

    ref = run()
    side = torch.cuda.Stream(dev)
    try:
        for k in range(7):
            if k % 3 == 0:
                h.set_stream(side.cuda_stream)
            elif k % 3 == 1:
                h.set_stream(0)
            else:
                h.reset_stream()
            got = run()
            for u, v in zip(ref, got):
                np.testing.assert_array_equal(u, v)
    finally:
        h.reset_stream()
    for b in range(nb):
        mw_o, _, _, lp_o = O.posterior_logpdf_direct(mw[b], dpr[b], X[b].T, 0.3, y[b])
        assert ref[2][b] == pytest.approx(lp_o, rel=1e-10)
        np.testing.assert_allclose(ref[0][b], mw_o, rtol=1e-8, atol=1e-10)


def test_last_route_and_option_codes(B, opt):
    # blr_last_route names the kernel family the dispatcher took (bench.py labels its roofline with it); blr_set_option tells an
    # unknown key (-2) from a malformed value (-3), and rejects numbers that are not numbers.
    a = B._abi
    h = a.default_handle()
    rng = _rng(4900)

    def post(D, N, nb, dtype=np.float64):
        X = rng.standard_normal((nb, N, D)).astype(dtype); y = rng.standard_normal((nb, N)).astype(dtype)
        lp = np.zeros(nb); info = np.zeros(nb, dtype=np.int32)
        h.posterior_batched(dtype, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, np.array([0.5], dtype=dtype), 0,
                            a.PRIOR_DIAGONAL, np.zeros(D, dtype=dtype), 0, np.ones(D, dtype=dtype), 1, 0, None, D, None, D, D * D, None, D, D * D, lp, info)
        assert info.tolist() == [0] * nb
        return h.last_route()

    assert post(128, 512, 2) == "fused_i8_kernel"
    assert post(128, 100, 2) == "fused_small_kernel<double, 8, 4>"
    assert post(64, 100, 2) == "fused_wave_kernel<double, 4, 4>"
    assert post(48, 100, 2, np.float32).startswith("fused_small_kernel<float, 3,")
    assert post(256, 300, 2, np.float32) == "gram_planes4_kernel"
    assert post(256, 300, 2) == "gram_tile_kernel<double>"
    opt("PLANES8", "1")
    assert post(256, 300, 2, np.float32) == "gram_planes_kernel<2>"
    opt("NO_FP16_PLANES", "1")
    assert post(256, 300, 2, np.float32) == "gram_planes_kernel<3>"
    opt("NO_PLANES", "1")
    assert post(256, 300, 2, np.float32) == "gram_tile_kernel<float, true>"
    opt("NO_I8_GRAM", "1")
    assert post(128, 512, 2) == "fused_small_kernel<double, 8, 4>"
    for key, value, code in (("NO_SUCH_SWITCH", "1", -2), ("WAVE_SPLIT", "3", -3), ("CHAIN_BATCH", "abc", -3), ("CHAIN_BATCH", "12x", -3),
                             ("CHAIN_WS_MB", "", 0), ("SWEEP", "sometimes", -3), ("GRAM_SPLITS", "7", -3)):
        if code == 0:
            h.set_option(key, value)
        else:
            with pytest.raises(a.BLRError) as ei:
                h.set_option(key, value)
            assert ei.value.code == code, (key, value, ei.value.code)
    with pytest.raises(a.BLRError):
        h.get_stat("no_such_counter")
    assert h.get_stat("workspace_bytes") >= 0


@pytest.mark.parametrize("noise", ["iso", "diag"])
def test_large_d_fp32_gram_on_bf16_matrix_cores_vs_f32_route(B, opt, noise):
    # D > 128 in fp32, aligned ColVecs, whole 128-row blocks and column ranges in multiples of 16: the Gram launch forms every macro tile
    # from an exact three-way bf16 split of the fp32 operands, six products, fp32 accumulation (blr_large.hpp, bf3_split_pack).  Against
    # the fp64 oracle it must be as good as the fp32 matrix instruction it replaces (option NO_BF16X3): within 4 x its error + 1e-7 on
    # the precision matrix, and both inside the fp32 tolerances of the other large-D tests.  Positive features on purpose: a split that
    # truncates instead of rounding is biased exactly there (same-sign residuals add up over the observations).
    a = B._abi
    h = a.default_handle()
    rng = _rng(5150)
    D, N = 384, 4096
    X = np.asfortranarray((0.5 + np.abs(rng.standard_normal((D, N)))).astype(np.float32))
    s = (np.exp(0.3 * rng.standard_normal(N)) if noise == "diag" else np.full(N, 0.7)).astype(np.float32)
    w0 = rng.standard_normal(D) / np.sqrt(D)
    y = (X.astype(float).T @ w0 + np.sqrt(s.astype(float)) * rng.standard_normal(N)).astype(np.float32)
    mw = (0.1 * rng.standard_normal(D)).astype(np.float32)
    dvec = np.exp(0.2 * rng.standard_normal(D)).astype(np.float32)
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw.astype(float), dvec.astype(float), X.astype(float), s.astype(float), y.astype(float))

    def run():
        mwp = np.zeros(D, dtype=np.float32); Tp = np.zeros((D, D), dtype=np.float32, order="F"); Ap = np.zeros((D, D), dtype=np.float32, order="F")
        lp = np.zeros(1); info = np.zeros(1, dtype=np.int32)
        if noise == "diag":
            nk, sv, ss = a.NOISE_DIAGONAL, s, N
        else:
            nk, sv, ss = a.NOISE_ISOTROPIC, s[:1].copy(), 0
        h.posterior_batched(np.float32, a.MEM_HOST, a.LAYOUT_COLVECS, 1, D, N, X, D, N * D, y, N, nk, sv, ss,
                            a.PRIOR_DIAGONAL, mw, 0, dvec, 1, 0, mwp, D, Tp, D, D * D, Ap, D, D * D, lp, info)
        assert info[0] == 0
        return _rel_errs(mwp, Ap, lp[0], mw_o, A_o, lp_o)

    # (round 6: the default route splits the operands ONCE, in a pass of its own -- blr_planes.hpp, "gram_planes_kernel"; option NO_PLANES
    # keeps round 5's kernel, which splits them inside the matrix loop; both are held to the f32 matrix instruction)
    e_pl = run()
    assert h.last_route() == "gram_planes4_kernel"   # two fp16 planes per operand under per-row power-of-two scales, three products, 64 x 64 per wave
    opt("PLANES8", "1")
    e_pl8 = run()
    assert h.last_route() == "gram_planes_kernel<2>"   # the same planes, eight waves of 32 x 64
    opt("NO_FP16_PLANES", "1")
    e_pl3 = run()
    assert h.last_route() == "gram_planes_kernel<3>"   # three bf16 planes, six products
    opt("NO_PLANES", "1")
    e_bf3 = run()
    assert h.last_route() == "gram_tile_kernel<float, true>"
    opt("NO_BF16X3", "1")
    e_f32 = run()
    assert h.last_route() == "gram_tile_kernel<float>"
    print(f"large-D fp32 Gram ({noise}): rel err (mw', A, logpdf)  fp16 x 2 planes {e_pl} (eight-wave kernel {e_pl8})  bf16 x 3 planes {e_pl3}  bf16 x 3 in the loop {e_bf3}  f32 matrix instruction {e_f32}")
    for e in (e_pl, e_pl8, e_pl3, e_bf3):
        assert e[1] <= 4 * e_f32[1] + 1e-7, (e, e_f32)
        assert e[0] <= 4 * e_f32[0] + 1e-6 and e[2] <= 4 * e_f32[2] + 1e-7, (e, e_f32)
        assert e[1] <= 2e-5 and e[2] <= 2e-4


def test_large_d_fp32_sampled_row_scales_and_their_exact_second_pass(B, opt):
    # Round 6: the row scales of the fp16 planes come from a SAMPLE of each row (the first 32 columns of every column chunk, doubled for
    # head-room) instead of a pass over X of its own; the planes pass checks that every entry still is a finite fp16 number and, when one
    # is not, the exact row maxima + the planes are made again (blr_planes.hpp; blr_get_stat "planes_redone").  (a) ordinary data: no
    # second pass, as good as the exact scales (option NO_SPEC_ROWMAX); (b) one entry 4000 x its row's other entries, in a column the
    # sample does not see: the second pass runs -- once, for that regressor -- and the result is as good as with exact scales; (c) an Inf
    # in the same place: info != 0 / NaN evidence either way, as before.
    a = B._abi
    h = a.default_handle()
    rng = _rng(6161)
    D, N = 256, 32768   # 2 row blocks, 256 column chunks of 128 columns: the sample is columns 0 .. 31 of each
    X0 = np.asfortranarray(rng.standard_normal((D, N)).astype(np.float32))
    s = np.full(1, 0.5, dtype=np.float32)
    w0 = rng.standard_normal(D) / np.sqrt(D)
    mw = np.zeros(D, dtype=np.float32)
    dvec = np.ones(D, dtype=np.float32)

    def run(X, y):
        mwp = np.zeros(D, dtype=np.float32); Tp = np.zeros((D, D), dtype=np.float32, order="F"); Ap = np.zeros((D, D), dtype=np.float32, order="F")
        lp = np.zeros(1); info = np.zeros(1, dtype=np.int32)
        h.posterior_batched(np.float32, a.MEM_HOST, a.LAYOUT_COLVECS, 1, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0,
                            a.PRIOR_DIAGONAL, mw, 0, dvec, 1, 0, mwp, D, Tp, D, D * D, Ap, D, D * D, lp, info)
        return mwp, Ap, lp[0], info[0]

    def case(X):
        y = (X.astype(float).T @ w0 + np.sqrt(0.5) * rng.standard_normal(N)).astype(np.float32)
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw.astype(float), dvec.astype(float), X.astype(float), 0.5, y.astype(float))
        h.reset_stats()
        m1, A1, l1, i1 = run(X, y)
        assert h.last_route() == "gram_planes4_kernel" and i1 == 0
        redone = h.get_stat("planes_redone")
        m1b, A1b, l1b, _ = run(X, y)
        assert np.array_equal(A1, A1b) and np.array_equal(m1, m1b) and l1 == l1b   # same inputs, same bits
        opt("NO_SPEC_ROWMAX", "1")
        m2, A2, l2, i2 = run(X, y)
        opt("NO_SPEC_ROWMAX", None)
        assert i2 == 0
        return redone, _rel_errs(m1, A1, l1, mw_o, A_o, lp_o), _rel_errs(m2, A2, l2, mw_o, A_o, lp_o)

    redone, e_s, e_x = case(X0)
    print(f"sampled row scales, gaussian rows: second pass ran {redone} x; rel err (mw', A, logpdf) sampled {e_s}  exact {e_x}")
    assert redone == 0
    assert e_s[1] <= 2 * e_x[1] + 5e-8 and e_s[0] <= 2 * e_x[0] + 1e-6 and e_s[2] <= 2 * e_x[2] + 1e-7, (e_s, e_x)
    Xm = X0.copy(order="F")
    Xm[200, 100] = 25.0     # 6 x the row's other entries, unseen by the sample: inside the head-room, a scale two binades off the exact one
    redone, e_s, e_x = case(Xm)
    print(f"sampled row scales, one entry 6 x its row: second pass ran {redone} x; rel err sampled {e_s}  exact {e_x}")
    assert redone == 0
    assert e_s[1] <= 2 * e_x[1] + 5e-8 and e_s[0] <= 2 * e_x[0] + 1e-6 and e_s[2] <= 2 * e_x[2] + 1e-7, (e_s, e_x)
    X1 = X0.copy(order="F")
    X1[200, 100] = 4000.0   # row 200 (second row block), column 100: chunk 0, beyond its first 32 columns
    redone, e_s, e_x = case(X1)
    print(f"sampled row scales, one entry 4000 x its row: second pass ran {redone} x; rel err sampled {e_s}  exact {e_x}")
    assert redone == 1
    assert e_s[1] <= 2 * e_x[1] + 5e-8 and e_s[0] <= 2 * e_x[0] + 1e-6 and e_s[2] <= 2 * e_x[2] + 1e-7, (e_s, e_x)
    X2 = X0.copy(order="F")
    X2[200, 100] = np.inf
    y = (X0.astype(float).T @ w0).astype(np.float32)
    for o in (None, "1"):
        opt("NO_SPEC_ROWMAX", o)
        _, _, lp, info = run(X2, y)
        assert info != 0 or not np.isfinite(lp)
    opt("NO_SPEC_ROWMAX", None)


def test_large_d_fp32_sampled_row_scales_in_a_group_of_regressors(B, opt):
    # Three regressors of one call share every launch of the large-D update (posterior_large_group); each has its OWN second-pass flag
    # in its slice of the workspace.  The middle one holds an entry 4000 x its row in a column the sample does not see: its planes are
    # made a second time, the others' are not (blr_get_stat "planes_redone" == 1), and all three are as good as with exact row maxima.
    a = B._abi
    h = a.default_handle()
    rng = _rng(6262)
    nb, D, N = 3, 256, 16384
    X = np.asfortranarray(rng.standard_normal((D, N * nb)).astype(np.float32))   # regressor b: columns b N .. (b + 1) N - 1
    X[70, N + 100] = 4000.0
    s = np.full(1, 0.5, dtype=np.float32)
    W = rng.standard_normal((D, nb)) / np.sqrt(D)
    y = np.concatenate([X[:, b * N:(b + 1) * N].astype(float).T @ W[:, b] + np.sqrt(0.5) * rng.standard_normal(N) for b in range(nb)]).astype(np.float32)
    mw = np.zeros(D, dtype=np.float32)
    dvec = np.ones(D, dtype=np.float32)

    def run():
        mwp = np.zeros((D, nb), dtype=np.float32, order="F"); lp = np.zeros(nb); info = np.zeros(nb, dtype=np.int32)
        h.posterior_batched(np.float32, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0,
                            a.PRIOR_DIAGONAL, mw, 0, dvec, 1, 0, mwp, D, None, D, D * D, None, D, D * D, lp, info)
        assert not info.any()
        return mwp, lp

    h.reset_stats()
    m1, l1 = run()
    assert h.get_stat("planes_redone") == 1
    opt("NO_SPEC_ROWMAX", "1")
    m2, l2 = run()
    opt("NO_SPEC_ROWMAX", None)
    for b in range(nb):
        Xb = X[:, b * N:(b + 1) * N].astype(float)
        mw_o, _, _, lp_o = O.posterior_logpdf_direct(mw.astype(float), dvec.astype(float), Xb, 0.5, y[b * N:(b + 1) * N].astype(float))
        e1 = (np.linalg.norm(m1[:, b] - mw_o) / np.linalg.norm(mw_o), abs(l1[b] - lp_o) / abs(lp_o))
        e2 = (np.linalg.norm(m2[:, b] - mw_o) / np.linalg.norm(mw_o), abs(l2[b] - lp_o) / abs(lp_o))
        assert e1[0] <= 2 * e2[0] + 1e-6 and e1[1] <= 2 * e2[1] + 1e-7, (b, e1, e2)


@pytest.mark.parametrize("nb", [8192, 1024])
def test_c4_at_its_stated_batch_vs_literal_oracle(B, nb):
    # BASELINE config 4 exactly as stated -- 8192 x (D = 64, N = 1024), fp64, isotropic noise, Lw = I -- and the 1024-regressor
    # block one of 8 GPUs gets (the router's 2-wave split).  Nine regressors, from every round of the grid-stride loop (2048
    # wave slots: rounds of 2048 regressors at one wave each, of 1024 at two), against the reference's LITERAL op sequence;
    # all of them finite with info = 0; the same bits when the call is repeated.
    import torch

    a = B._abi
    h = a.default_handle()
    dev = torch.device("cuda:0")
    D, N = 64, 1024
    g = torch.Generator(device=dev).manual_seed(2024 + nb)
    X = torch.randn((nb, N, D), generator=g, dtype=torch.float64, device=dev)
    wstar = torch.randn((nb, D), generator=g, dtype=torch.float64, device=dev)
    y = torch.einsum("bnd,bd->bn", X, wstar) + (0.1 ** 0.5) * torch.randn((nb, N), generator=g, dtype=torch.float64, device=dev)
    s = torch.full((1,), 0.1, dtype=torch.float64, device=dev)
    mw = torch.randn((nb, D), generator=g, dtype=torch.float64, device=dev)  # test/test_utils.jl:6: mw = randn(D)
    dpr = torch.ones((D,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    outs = []
    for _ in range(2):
        mp = torch.empty((nb, D), dtype=torch.float64, device=dev)
        Tp = torch.empty((nb, D, D), dtype=torch.float64, device=dev)
        lp = torch.empty((nb,), dtype=torch.float64, device=dev)
        info = torch.full((nb,), 7, dtype=torch.int32, device=dev)
        h.posterior_batched(np.float64, a.MEM_DEVICE, a.LAYOUT_COLVECS, nb, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N, a.NOISE_ISOTROPIC,
                            s.data_ptr(), 0, a.PRIOR_DIAGONAL, mw.data_ptr(), D, dpr.data_ptr(), 1, 0, mp.data_ptr(), D, Tp.data_ptr(), D,
                            D * D, None, D, D * D, lp.data_ptr(), info.data_ptr())
        torch.cuda.synchronize()
        assert int(info.abs().sum().item()) == 0
        assert bool(torch.isfinite(lp).all().item()) and bool(torch.isfinite(mp).all().item())
        outs.append((mp, Tp, lp))
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    mp, Tp, lp = outs[0]
    per_round = 2048 if nb >= 2048 else 1024  # wave slots per round at the router's split
    picks = sorted({0, 1, per_round - 1, per_round % nb, (per_round + 1) % nb, (2 * per_round + 3) % nb, (3 * per_round + 17) % nb,
                    nb // 2 + 5, nb - 1})
    for b in picks:
        Xb = X[b].cpu().numpy().T
        args = (mw[b].cpu().numpy(), np.ones(D), Xb, 0.1, y[b].cpu().numpy())
        mw_o, T_o, L_o = O.posterior_literal(*args)
        assert lp[b].item() == pytest.approx(O.logpdf_literal(*args), rel=1e-10), b
        np.testing.assert_allclose(mp[b].cpu().numpy(), mw_o, rtol=1e-8, atol=1e-10)
        Tn = np.triu(Tp[b].cpu().numpy().T)
        np.testing.assert_allclose(Tn.T @ Tn, L_o, rtol=1e-10, atol=1e-10 * np.abs(L_o).max())


@pytest.mark.parametrize("with_comm", [False, True])
@pytest.mark.parametrize("D,N,dtype", [(300, 700, np.float64), (64, 333, np.float64), (200, 1000, np.float32)])
def test_posterior_nsharded_one_call(B, with_comm, D, N, dtype):
    # blr_posterior_nsharded_*: statistics -> all-reduce over the handle's communicator -> finish, in one call per rank.  One
    # rank here (with and without a real 1-rank RCCL communicator); the multi-rank sum itself is covered by the gloo tests.
    import torch
    from blr_amd import _abi

    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    rng = _rng(95 + D)
    X = rng.standard_normal((D, N)).astype(dtype)
    mw = rng.standard_normal(D).astype(dtype)
    dpr = np.exp(0.3 * rng.standard_normal(D)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(dtype)
    y = rng.standard_normal(N).astype(dtype)
    DP = (D + 127) // 128 * 128
    lds = DP + 128
    h = _abi.Handle(0)
    try:
        if with_comm:
            h.comm_init(1, 0, _abi.Handle.comm_unique_id())
        Xd = torch.tensor(X.T.copy(), dtype=tdt, device=dev)  # [N, D] row-major == D x N ColVecs
        yd, sd, mwd, dd = (torch.tensor(v, dtype=tdt, device=dev) for v in (y, s, mw, dpr))
        stats = torch.zeros((DP, lds), dtype=tdt, device=dev)
        scal = torch.zeros(2, dtype=torch.float64, device=dev)
        mp = torch.zeros(D, dtype=tdt, device=dev)
        Tp = torch.zeros((D, D), dtype=tdt, device=dev)
        lp = torch.zeros(1, dtype=torch.float64, device=dev)
        info = torch.full((1,), 5, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        h.posterior_nsharded(dtype, _abi.LAYOUT_COLVECS, D, N, N, Xd.data_ptr(), D, yd.data_ptr(), _abi.NOISE_DIAGONAL, sd.data_ptr(),
                             _abi.PRIOR_DIAGONAL, mwd.data_ptr(), dd.data_ptr(), 1, stats.data_ptr(), lds, scal.data_ptr(), mp.data_ptr(),
                             Tp.data_ptr(), D, None, D, lp.data_ptr(), info.data_ptr())
        h.synchronize()
        assert info.item() == 0
        f64 = lambda a: np.asarray(a, dtype=np.float64)
        mw_o, T_o, L_o = O.posterior_literal(f64(mw), f64(dpr), f64(X), f64(s), f64(y))
        lp_o = O.logpdf_literal(f64(mw), f64(dpr), f64(X), f64(s), f64(y))
        rtol = 1e-9 if dtype == np.float64 else 3e-4
        Tn = np.triu(Tp.cpu().numpy().T.astype(np.float64))
        np.testing.assert_allclose(Tn.T @ Tn, L_o, rtol=rtol, atol=rtol * np.abs(L_o).max())
        np.testing.assert_allclose(mp.cpu().numpy(), mw_o, rtol=100 * rtol, atol=10 * rtol * np.abs(mw_o).max())
        assert lp.item() == pytest.approx(lp_o, rel=20 * rtol)
        if with_comm:
            h.comm_destroy()
    finally:
        h.close()


# ---- reverse-mode rule of rand (README.md:56-60) and blr_apply_weights_* --------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("layout", ["colvecs", "rowvecs"])
@pytest.mark.parametrize("D,N,S", [(2, 10, 5), (7, 13, 3), (128, 300, 64), (300, 77, 9)])
def test_apply_weights_vs_oracle(B, dtype, layout, D, N, S):
    # Y = X'W for S given weight vectors: evaluation of function samples, reference sampling_functions.jl:16-18
    a = B._abi
    rng = _rng(31000 + D + N)
    X = rng.standard_normal((D, N)).astype(dtype)
    W = np.asfortranarray(rng.standard_normal((D, S)).astype(dtype))
    Y = np.full((N, S), np.nan, dtype=dtype, order="F")
    if layout == "colvecs":
        Xa, lay, ldx = np.asfortranarray(X), a.LAYOUT_COLVECS, D
    else:
        Xa, lay, ldx = np.asfortranarray(X.T), a.LAYOUT_ROWVECS, N
    a.default_handle().apply_weights(dtype, a.MEM_HOST, lay, D, N, S, Xa, ldx, W, D, Y, N)
    ref = X.astype(np.float64).T @ W.astype(np.float64)
    tol = 1e-13 if dtype == np.float64 else 4 * np.finfo(np.float32).eps
    assert np.max(np.abs(Y - ref)) <= tol * np.sqrt(D) * np.max(np.abs(ref))


@pytest.mark.parametrize("prior", ["diagonal", "dense", "factor"])
@pytest.mark.parametrize("container", ["colvecs", "rowvecs"])
@pytest.mark.parametrize("noise", ["diagonal", "isotropic"])
def test_rand_pullback_vs_oracle(B, prior, container, noise):
    rng = _rng(32000)
    N, D, S = 40, 9, 6
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    if noise == "isotropic":
        s = np.float64(0.37)
    if prior == "diagonal":
        Lw_o = np.exp(0.3 * rng.standard_normal(D))
        Lw_b = B.Diagonal(Lw_o)
    elif prior == "dense":
        Lw_o, Lw_b = Lw, Lw
    else:
        U = O.chol_upper(Lw)
        Lw_o, Lw_b = U.T @ U, B.PDMat(U)
    x = B.ColVecs(np.asfortranarray(X)) if container == "colvecs" else B.RowVecs(np.ascontiguousarray(X.T))
    fx = B.BayesianLinearRegressor(mw, Lw_b)(x, B.Diagonal(s) if noise == "diagonal" else float(s))
    state = rng.bit_generator.state
    Y, pb = B.rand_and_pullback(rng, fx, S)
    rng.bit_generator.state = state
    Z1 = rng.standard_normal((S, D)).T  # the draws rand_and_pullback made, in the reference's order (:51 then :52), filled
    Z2 = rng.standard_normal((S, N)).T  # column by column like Julia's randn(rng, D, S)
    s_vec = np.broadcast_to(np.asarray(s, dtype=float), (N,))
    np.testing.assert_allclose(Y, O.rand(mw, Lw_o, X, s_vec, Z1, Z2), rtol=1e-10, atol=1e-11)
    Yb = rng.standard_normal((N, S))
    g = pb(Yb)
    go = O.rand_pullback(mw, Lw_o, X, s_vec, Z1, Yb)
    gX = g["X"] if container == "colvecs" else g["X"].T
    np.testing.assert_allclose(gX, go["X"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(g["mw"], go["mw"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(g["Lw"], go["U"] if prior == "factor" else go["Lw"], rtol=1e-8, atol=1e-10)
    sb = np.sum(Yb * Z2, axis=1) / (2 * np.sqrt(s_vec))
    np.testing.assert_allclose(g["noise"], sb if noise == "diagonal" else sb.sum(), rtol=1e-10)


def test_evaluate_function_samples_in_one_pass(B):
    # reference sampling_functions.jl:16-18 for a batch of samples: column j == samples[j](X), through a basis too
    rng = _rng(33000)
    D, N = 6, 50
    X = rng.standard_normal((D, N))
    f = B.BayesianLinearRegressor(rng.standard_normal(D), B.Diagonal(np.exp(rng.standard_normal(D))))
    smp = B.rand(rng, f, 4)
    Y = B.evaluate(smp, B.ColVecs(np.asfortranarray(X)))
    assert Y.shape == (N, 4)
    for j in range(4):
        np.testing.assert_allclose(Y[:, j], X.T @ smp[j].w, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(Y[:, j], smp[j](B.ColVecs(np.asfortranarray(X))), rtol=1e-12, atol=1e-13)
    phi = lambda x: B.ColVecs(np.asfortranarray(np.vstack([x.X, x.X ** 2])))
    bfr = B.BasisFunctionRegressor(B.BayesianLinearRegressor(np.zeros(2 * D), B.Diagonal(np.ones(2 * D))), phi)
    smp = B.rand(rng, bfr, (2, 3))
    Y = B.evaluate(smp, B.ColVecs(np.asfortranarray(X)))
    flat = smp.reshape(-1, order="F")
    for j in range(6):
        np.testing.assert_allclose(Y[:, j], np.vstack([X, X ** 2]).T @ flat[j].w, rtol=1e-12, atol=1e-12)



# ---- bench.py --gpus 2 on ONE GPU: the N > 1 code path of the measurement itself ----------------------------------
def test_bench_two_ranks_strong_scaling_line(bench_two_rank_runs):
    # `python -m torch.distributed.run --nproc-per-node 2 bench.py --config c4 --gpus 2` with both ranks on cuda:0 and the 64 KiB
    # all-gather staged through gloo (tests/conftest.py starts it, and a one-rank run of the same fixed batch, before this process
    # touches the GPU).  The line must describe BASELINE's strong-scaling case -- 8192 regressors in two contiguous blocks -- and
    # the total log evidence must have the SAME BITS as the one-rank run: same regressors (chunk-seeded generation), same
    # per-regressor kernel results wherever a regressor lives, one fixed-order sum over the gathered vector.
    one, two = bench_two_rank_runs["one"], bench_two_rank_runs["two"]
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["scaling"] == "strong" and one["scaling"] == "strong"
    assert two["config"]["global_batch"] == 8192 and one["config"]["global_batch"] == 8192
    assert two["config"]["batch_per_gpu"] == 4096
    assert two["steps"] == 3 and two["warmup"] == 1
    assert two["unit"] == "posterior-updates/s" and two["value"] > 0
    assert np.isfinite(two["total_log_evidence"])
    assert two["total_log_evidence"] == one["total_log_evidence"]  # bit for bit
    assert two["roofline"]["frac"] > 0 and two["roofline"]["kernel"].startswith("fused_wave_kernel")


@pytest.mark.gpu
def test_bench_two_ranks_bare_invocation_spawns_its_ranks(bench_two_rank_runs):
    # `python bench.py --config c4 --gpus 2` WITHOUT a launcher used to run one rank and print n_gpus = 1.
    # It now starts the two ranks itself (before touching the GPU) and relays rank 0's line: same line as the launcher-started run.
    bare, two = bench_two_rank_runs["bare2"], bench_two_rank_runs["two"]
    assert bare["n_gpus"] == 2 and bare["scaling"] == "strong"
    assert bare["config"]["batch_per_gpu"] == 4096 and bare["config"]["global_batch"] == 8192
    assert bare["total_log_evidence"] == two["total_log_evidence"]  # bit for bit



@pytest.mark.gpu
def test_bench_library_comm_and_its_fallback(bench_two_rank_runs):
    # The evidence exchange on the library's own RCCL binding (one-rank communicator, side stream, two alternating buffers) gives
    # the bits of the plain local sum; a binding that cannot be set up does not cost the job its line -- every rank is told (MIN of
    # a success flag) and the exchange goes through torch.distributed / the local sum, which the line's `config.sharding` says.
    one, comm, fail = bench_two_rank_runs["one"], bench_two_rank_runs["comm1"], bench_two_rank_runs["commfail"]
    assert "library RCCL" in comm["config"]["sharding"] and "failed" not in comm["config"]["sharding"]
    assert comm["total_log_evidence"] == one["total_log_evidence"]
    assert "library RCCL binding failed" in fail["config"]["sharding"]
    assert "library RCCL binding failed" in fail["_stderr"]
    assert fail["n_gpus"] == 1 and fail["total_log_evidence"] == one["total_log_evidence"]


# ---- round 6: a yardstick for the fp64-emulating route -----------------------------------------------------
def _extended_truth(mw, dpr, X, s, y):
    """(A, mw', logpdf) of reference :55-69 / :72-89 in the direct form, evaluated in np.longdouble (x86: 64-bit mantissa, 1e-19) --
    the truth both fp64 LAPACK and the device routes are measured against.  X: [N, D]; diagonal prior dpr; isotropic noise s."""
    L = np.longdouble
    assert np.finfo(L).nmant >= 63, "np.longdouble is not the x87 extended type on this host"
    Xl, yl, ml, dl, sl = X.astype(L), y.astype(L), mw.astype(L), dpr.astype(L), L(s)
    N, D = X.shape
    A = (Xl.T @ Xl) / sl
    A[np.diag_indices(D)] += dl
    dy = yl - Xl @ ml
    b = (Xl.T @ dy) / sl
    Lc = np.zeros((D, D), dtype=L)
    for j in range(D):  # Cholesky, column by column
        v = A[j:, j] - Lc[j:, :j] @ Lc[j, :j]
        Lc[j:, j] = v / np.sqrt(v[0])
    u = np.zeros(D, dtype=L)
    for j in range(D):
        u[j] = (b[j] - Lc[j, :j] @ u[:j]) / Lc[j, j]
    m = np.zeros(D, dtype=L)
    for j in range(D - 1, -1, -1):
        m[j] = (u[j] - Lc[j + 1:, j] @ m[j + 1:]) / Lc[j, j]
    two_pi = L(2) * np.arctan(L(1)) * L(4)
    lp = -(N * np.log(two_pi) + N * np.log(sl) + (dy @ dy) / sl + 2 * np.sum(np.log(np.diag(Lc))) - np.sum(np.log(dl)) - u @ u) / 2
    terms = float((dy @ dy) / sl) + N * float(np.log(two_pi)) + abs(N * float(np.log(sl))) + abs(2 * float(np.sum(np.log(np.diag(Lc))))) + abs(float(np.sum(np.log(dl))))
    return A, ml + m, lp, terms


def _errs_vs_truth(A, mwp, lp, At, mt, lpt):
    dA = np.sqrt(np.diag(At))
    eA = float(np.max(np.abs(A.astype(np.longdouble) - At) / np.outer(dA, dA)))
    em = float(np.max(np.abs((mwp.astype(np.longdouble) - mt) * dA)) / np.max(np.abs(mt * dA)))
    el = float(abs(np.longdouble(lp) - lpt) / abs(lpt))
    return eA, em, el


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gauss", "float32_origin", "row_scaled", "log_normal"])
def test_c2_size_forward_errors_against_extended_precision(B, opt, kind):
    # Until now the int8 route -- the HEADLINE kernel -- was held to hand-picked constants against an fp64 oracle whose own error is
    # of the same order.  Here, at BASELINE's size (D = 128, N = 4096), A, mw' and the evidence are computed in x87 extended precision
    # and three implementations are measured against that truth: fp64 LAPACK on the reference's literal op sequence (:72-89) and on
    # the direct form (the yardstick, as _assert_fp32_within_lapack does for fp32), the int8 route, the fp64 kernel.  Bound: 4 x the
    # yardstick + a floor.  Floors: for A, sqrt(N) eps of sqrt(A_ii A_jj) -- the error of a running sum over the N observations in a
    # FIXED order, which is what bit-reproducible kernels do (OpenBLAS's blocked, FMA-contracted dgemm lands at 7e-16 here: a fifth
    # of that; the fp64 kernel measures 4.5e-15) -- plus, on the int8 route, the 3e-14 of truncated digit products the header
    # documents for the six-group plan; for the evidence 4 eps x the terms that cancel in it.  All three implementations' numbers go
    # into the assertion message.
    a = B._abi
    h = a.default_handle()
    rng = _rng(6100 + len(kind))
    nb, D, N = 2, 128, 4096
    X = rng.standard_normal((nb, N, D))
    rowscale = np.ones(D)
    if kind == "float32_origin":
        X = X.astype(np.float32).astype(np.float64)
    elif kind == "row_scaled":
        rowscale = np.ldexp(1.0, (np.arange(D) % 7) * 4 - 12)
        X *= rowscale[None, None, :]
    elif kind == "log_normal":
        X = np.exp(0.3 * X) * rng.choice([-1.0, 1.0], size=X.shape)  # (sigma = 0.5 already sends every other regressor back to the fp64 kernel)
    w = rng.standard_normal((nb, D)) / rowscale[None, :]
    y = np.einsum("bnd,bd->bn", X, w) + np.sqrt(0.1) * rng.standard_normal((nb, N))
    dpr = np.exp(0.3 * rng.standard_normal((nb, D))) / rowscale[None, :] ** 2
    mw = np.zeros((nb, D))
    s = np.array([0.1])

    def run():
        mp = np.zeros((nb, D)); Ap = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_DIAGONAL,
                            mw, D, dpr, 1, D, mp, D, None, D, D * D, Ap, D, D * D, lp, info)
        assert info.tolist() == [0] * nb
        return mp, Ap, lp

    h.reset_stats()
    i8 = run()
    assert h.last_route() == "fused_i8_kernel" and h.get_stat("i8_handed_back") == 0
    opt("NO_I8_GRAM", "1")
    f64k = run()
    assert h.last_route().startswith("fused_small_kernel<double")
    eps = float(np.finfo(np.float64).eps)
    for b in range(nb):
        At, mt, lpt, terms = _extended_truth(mw[b], dpr[b], X[b], 0.1, y[b])
        m_l, _, A_l = O.posterior_literal(mw[b], dpr[b], X[b].T, 0.1, y[b])
        lp_l = O.logpdf_literal(mw[b], dpr[b], X[b].T, 0.1, y[b])
        m_d, _, A_d, lp_d = O.posterior_logpdf_direct(mw[b], dpr[b], X[b].T, 0.1, y[b])
        yard = tuple(max(u, v) for u, v in zip(_errs_vs_truth(A_l, m_l, lp_l, At, mt, lpt), _errs_vs_truth(A_d, m_d, lp_d, At, mt, lpt)))
        e_i8 = _errs_vs_truth(i8[1][b], i8[0][b], i8[2][b], At, mt, lpt)
        e_64 = _errs_vs_truth(f64k[1][b], f64k[0][b], f64k[2][b], At, mt, lpt)
        floor_lp = 4 * eps * terms / abs(float(lpt))
        msg = (f"[{kind}, regressor {b}] errors (A / sqrt(A_ii A_jj), mw' in the metric of A, evidence): fp64 LAPACK {yard}, int8 route {e_i8}, "
               f"fp64 kernel {e_64}, evidence floor {floor_lp:.2e}")
        print("\n" + msg)
        run_sum = np.sqrt(N) * eps
        # The evidence inherits the error of A through |u|^2 = m'A m (m = mw' - mw): to first order d logpdf = -m' dA m / 2 (+ tr(A^-1 dA) / 2,
        # D times smaller here).  With independent zero-mean entry errors of standard deviation floor_A / 4 x sqrt(A_ii A_jj) (the floor
        # bounds the LARGEST of 16 k entries, ~ 4 sigma) that is a sum with standard deviation (floor_A / 4) sum_i m_i^2 A_ii / 2; six
        # of those are allowed.  (On these problems -- y = X'w + noise, |w| ~ 11 -- delta'delta / s and |u|^2 are 3000 times the evidence.)
        mAm = float(np.sum(((mt - mw[b].astype(np.longdouble)) ** 2) * np.diag(At)))
        for e, floor_A in ((e_i8, run_sum + 3e-14), (e_64, run_sum)):
            assert e[0] <= 4 * yard[0] + floor_A, msg
            assert e[1] <= 4 * yard[1] + 64 * floor_A, msg  # (mw' = A^-1 b: the entries' error times the conditioning of these problems, ~ 50)
            assert e[2] <= 4 * yard[2] + floor_lp + 0.75 * floor_A * mAm / abs(float(lpt)), msg + f", m'A m = {mAm:.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("noise", ["isotropic", "diagonal"])
def test_i8_digit_groups_option(B, opt, noise):
    # six digit groups (the default under isotropic noise since round 5) cost a factor 3 in the error of A (3e-14 against
    # 1e-14 of the diagonal scale) and the evidence check went from rel 1e-11 to 1e-11 + a conditioning term.  Option I8_GROUPS = 7
    # keeps the seventh group (260 MFMAs per k-step) for callers who want the old numbers: held here to the OLD tolerances -- evidence
    # rel 1e-11 against the oracle, A within 1.5e-14 of max |A| of the fp64 kernel.  Under diagonal noise seven groups are the default
    # (the rows' bounds are loose by the spread of the variances) and I8_GROUPS = 6 is the faster plan for variances of one magnitude
    # (here: within a factor 1.5 of each other).
    a = B._abi
    h = a.default_handle()
    rng = _rng(6200 + len(noise))
    nb, D, N = 4, 128, 2048
    X, y = _i8_case(rng, nb, N, "gauss")
    dpr = np.exp(0.3 * rng.standard_normal((nb, D)))
    mw = np.zeros((nb, D))
    diag = noise == "diagonal"
    s = 0.1 * (1.0 + 0.5 * rng.random((nb, N))) if diag else np.array([0.1])

    def run():
        mp = np.zeros((nb, D)); Ap = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_DIAGONAL if diag else a.NOISE_ISOTROPIC,
                            s, N if diag else 0, a.PRIOR_DIAGONAL, mw, D, dpr, 1, D, mp, D, None, D, D * D, Ap, D, D * D, lp, info)
        assert info.tolist() == [0] * nb
        return mp, Ap, lp

    default = run()
    assert h.last_route() == "fused_i8_kernel"
    opt("I8_GROUPS", "7")
    g7 = run()
    assert h.last_route() == ("fused_i8_kernel" if diag else "fused_i8_kernel (7 digit groups)")
    opt("I8_GROUPS", "6")
    g6 = run()
    assert h.last_route() == ("fused_i8_kernel (6 digit groups)" if diag else "fused_i8_kernel")
    for u, v in zip(default, g7 if diag else g6):
        np.testing.assert_array_equal(u, v)  # the option naming the default plan changes nothing
    opt("NO_I8_GRAM", "1")
    f64k = run()
    e7 = e6 = 0.0
    for b in range(nb):
        sb = s[b] if diag else 0.1
        _, _, A_o, lp_o = O.posterior_logpdf_direct(mw[b], dpr[b], X[b].T, sb, y[b])
        assert abs(g7[2][b] - lp_o) <= 1e-11 * abs(lp_o)
        e7 = max(e7, float(np.abs(g7[1][b] - f64k[1][b]).max() / np.abs(f64k[1][b]).max()))
        e6 = max(e6, float(np.abs(g6[1][b] - f64k[1][b]).max() / np.abs(f64k[1][b]).max()))
    print(f"\n[{noise}] A against the fp64 kernel, of max |A|: seven groups {e7:.2e}, six groups {e6:.2e}")
    assert e7 <= 1.5e-14 and e6 <= 1e-13
    with pytest.raises(Exception):
        h.set_option("I8_GROUPS", "5")


@pytest.mark.gpu
def test_last_route_names_the_kernel_that_did_the_work(B, opt):
    # blr_last_route said "fused_i8_kernel" even when the probe slice had sent the whole batch to the fp64 kernel.
    # The int8 route decides on the device; blr_last_route now looks at the call's hand-back count.
    a = B._abi
    h = a.default_handle()
    rng = _rng(6300)
    nb, D, N = 1280, 128, 512
    dpr = np.ones(D); mw = np.zeros(D); s = np.array([0.1])

    def run(X):
        w = rng.standard_normal((nb, D)) / np.sqrt(D)
        y = np.einsum("bnd,bd->bn", X, w) + np.sqrt(0.1) * rng.standard_normal((nb, N))
        mp = np.zeros((nb, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
        h.posterior_batched(np.float64, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_ISOTROPIC, s, 0, a.PRIOR_DIAGONAL,
                            mw, 0, dpr, 1, 0, mp, D, None, D, D * D, None, D, D * D, lp, info)
        assert info.tolist() == [0] * nb

    opt("I8_PROBE_MIN", "1024")
    run(rng.standard_normal((nb, N, D)))
    assert h.last_route() == "fused_i8_kernel"
    h.reset_stats()
    run(np.exp(1.5 * rng.standard_normal((nb, N, D))) * rng.choice([-1.0, 1.0], size=(nb, N, D)))  # log-normal features: the probe slice falls back
    back = h.get_stat("i8_handed_back")
    route = h.last_route()
    assert 2 * back > nb, back
    assert route == f"fused_small_kernel<double, 8, 4> (int8 route handed back {back} of {nb})", route
    run(rng.standard_normal((nb, N, D)))  # the count is per call: an ordinary batch afterwards is the int8 kernel's again
    assert h.last_route() == "fused_i8_kernel"
