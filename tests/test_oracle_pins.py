"""Pins the CPU oracle against everything in the reference's own tests that does not depend on
Julia's MersenneTwister stream.  CPU only."""
import numpy as np
import pytest

from oracle import blr_oracle as O

RTOL = 1.5e-8  # Julia's default isapprox rtol = sqrt(eps(Float64))


def _rng(i=0):
    return np.random.Generator(np.random.PCG64(123456 + i))


def test_doctest_golden_vector():
    # /root/reference/src/basis_function_regression.jl:11-28 -- the only literal numbers in the repo
    x = O.RowVecs(np.linspace(-1.0, 1.0, 5)[:, None])
    phix = O.phi_test(x)
    X = O.x_as_colvecs(phix)
    v = O.var(np.zeros(2), np.ones(2), X, 1e-18)  # default noise 1e-18 [AbstractGPs]
    np.testing.assert_allclose(v, [2.0, 1.25, 1.0, 1.25, 2.0], rtol=0, atol=1e-15)
    m, v2 = O.marginals_direct(np.zeros(2), np.eye(2), X, 1e-18)
    np.testing.assert_allclose(v2, [2.0, 1.25, 1.0, 1.25, 2.0], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(m, np.zeros(5))


@pytest.mark.parametrize("N,D", [(13, 7), (11, 3), (11, 2)])
def test_logpdf_naive_identity_dense_noise(N, D):
    # /root/reference/test/bayesian_linear_regression.jl:22-38
    rng = _rng(1)
    X, mw, Lw, Sy = O.generate_toy_problem(rng, N, D)
    y = O.rand(mw, Lw, X, Sy, rng.standard_normal((D, 1)), rng.standard_normal((N, 1)))[:, 0]
    lp = O.logpdf_literal(mw, Lw, X, Sy, y)
    assert lp == pytest.approx(O.logpdf_naive(mw, Lw, X, Sy, y), rel=RTOL)
    assert lp == pytest.approx(O.logpdf_naive_mp(mw, Lw, X, Sy, y), rel=1e-12)


@pytest.mark.parametrize("N,D", [(13, 7), (10, 2), (40, 16), (5, 9)])
def test_direct_form_equals_literal_sequence(N, D):
    rng = _rng(2)
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    y = rng.standard_normal(N)
    mw_l, T_l, Lw_l = O.posterior_literal(mw, Lw, X, s, y)
    lp_l = O.logpdf_literal(mw, Lw, X, s, y)
    mw_d, T_d, Lw_d, lp_d = O.posterior_logpdf_direct(mw, Lw, X, s, y)
    np.testing.assert_allclose(mw_d, mw_l, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(T_d, T_l, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(Lw_d, Lw_l, rtol=1e-11, atol=1e-12)
    assert lp_d == pytest.approx(lp_l, rel=1e-12)
    assert lp_d == pytest.approx(O.logpdf_naive_mp(mw, Lw, X, s, y), rel=1e-12)
    # isotropic noise, prior given by its factor (PDMat closure, :93)
    Uw = O.chol_upper(Lw)
    mw_f, T_f, _, lp_f = O.posterior_logpdf_direct(mw, None, X, 0.1, y, prior_factor=Uw)
    mw_i, T_i, _ = O.posterior_literal(mw, Lw, X, 0.1, y)
    np.testing.assert_allclose(mw_f, mw_i, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(T_f, T_i, rtol=1e-11, atol=1e-12)
    assert lp_f == pytest.approx(O.logpdf_literal(mw, Lw, X, 0.1, y), rel=1e-12)


def test_posterior_low_noise():
    # /root/reference/test/bayesian_linear_regression.jl:40-48
    rng = _rng(3)
    N, D = 13, 7
    X, mw, Lw, _ = O.generate_toy_problem(rng, N, D)
    eps = np.finfo(np.float64).eps
    y = O.rand(mw, Lw, X, eps, rng.standard_normal((D, 1)), rng.standard_normal((N, 1)))[:, 0]
    for post in (O.posterior_literal(mw, Lw, X, eps, y)[:3], O.posterior_logpdf_direct(mw, Lw, X, eps, y)[:3]):
        mw_p, T, Lw_p = post
        np.testing.assert_allclose(O.mean(mw_p, X), y, rtol=RTOL)
        assert np.all(O.cov(mw_p, Lw_p, X, eps) < 1000 * eps)


def test_posterior_repeated_conditioning():
    # /root/reference/test/bayesian_linear_regression.jl:49-70
    rng = _rng(4)
    N, D = 13, 7
    X, mw, Lw, Sy = O.generate_toy_problem(rng, N, D)
    Xp = rng.standard_normal((D, N))
    y = O.rand(mw, Lw, X, Sy, rng.standard_normal((D, 1)), rng.standard_normal((N, 1)))[:, 0]
    N1 = N - 3
    S1, S2 = Sy[:N1, :N1], Sy[N1:, N1:]
    Syp = np.zeros((N, N))
    Syp[:N1, :N1], Syp[N1:, N1:] = S1, S2
    m1, _, L1 = O.posterior_literal(mw, Lw, X[:, :N1], S1, y[:N1])
    m2, _, L2 = O.posterior_literal(m1, L1, X[:, N1:], S2, y[N1:])
    m, _, L = O.posterior_literal(mw, Lw, X, Syp, y)
    np.testing.assert_allclose(O.mean(m, Xp), O.mean(m2, Xp), rtol=RTOL)
    np.testing.assert_allclose(O.cov(m, L, Xp, Sy), O.cov(m2, L2, Xp, Sy), rtol=RTOL)
    # same with the direct form and diagonal noise, carrying the factor forward (:93)
    s = np.exp(rng.standard_normal(N))
    a1 = O.posterior_logpdf_direct(mw, Lw, X[:, :N1], s[:N1], y[:N1])
    a2 = O.posterior_logpdf_direct(a1[0], None, X[:, N1:], s[N1:], y[N1:], prior_factor=a1[1])
    a = O.posterior_logpdf_direct(mw, Lw, X, s, y)
    np.testing.assert_allclose(a2[0], a[0], rtol=1e-10)
    np.testing.assert_allclose(a2[2], a[2], rtol=1e-10)
    # chain rule of the evidence: log p(y) = log p(y1) + log p(y2 | y1)
    assert a1[3] + a2[3] == pytest.approx(a[3], rel=1e-12)


def test_pdmat_symmetric_equivalence():
    # /root/reference/test/bayesian_linear_regression.jl:90-112
    rng = _rng(5)
    N, D = 13, 7
    X = rng.standard_normal((D, N))
    Xp = rng.standard_normal((D, N))
    U = np.triu(rng.standard_normal((D, D)))
    C = 0.1 * rng.standard_normal((N, N))
    mw, Sy = rng.standard_normal(D), C @ C.T + np.eye(N)
    Lw = U.T @ U + np.eye(D)
    y = O.rand(mw, Lw, X, Sy, rng.standard_normal((D, 1)), rng.standard_normal((N, 1)))[:, 0]
    m_sym, T_sym, L_sym = O.posterior_literal(mw, Lw, X, Sy, y)
    # the PDMat branch keeps T; the Symmetric branch forms T'T: predictions must agree
    np.testing.assert_allclose(O.mean(m_sym, Xp), Xp.T @ m_sym)
    np.testing.assert_allclose(O.cov(m_sym, T_sym.T @ T_sym, Xp, Sy), O.cov(m_sym, L_sym, Xp, Sy), rtol=RTOL)
    np.testing.assert_allclose(np.tril(T_sym, -1), 0)


def test_unknown_container_errors():
    # /root/reference/test/bayesian_linear_regression.jl:116-122
    rng = _rng(6)
    x = [row for row in rng.standard_normal((11, 5))]
    with pytest.raises(TypeError):
        O.x_as_colvecs(x)


@pytest.mark.parametrize("container", ["matrix", "colvecs", "rowvecs"])
def test_bfr_equals_blr_of_phi(container):
    # /root/reference/test/basis_function_regression.jl:13-28
    rng = _rng(7)
    N, D = 11, 2
    X, mw, Lw, Sy = O.generate_toy_problem(rng, N, D)
    x = {"matrix": X, "colvecs": O.ColVecs(X), "rowvecs": O.RowVecs(np.ascontiguousarray(X.T))}[container]
    Phi = O.x_as_colvecs(O.phi_test(x))
    np.testing.assert_allclose(Phi[0], 1.0)
    np.testing.assert_allclose(Phi[1], X[0] * X[1])
    y = rng.standard_normal(N)
    lp = O.logpdf_literal(mw, Lw, Phi, Sy, y)
    assert lp == pytest.approx(O.logpdf_naive(mw, Lw, Phi, Sy, y), rel=RTOL)


def test_layout_independence_of_samples():
    # /root/reference/test/sampling_functions.jl:8-15: g(X) == g(ColVecs(X)) == g(RowVecs(X'))
    rng = _rng(8)
    N, D = 11, 5
    X, mw, Lw, _ = O.generate_toy_problem(rng, N, D)
    w = O.sample_weights(mw, Lw, rng.standard_normal(D))
    g = lambda x: O.x_as_colvecs(x).T @ w
    assert np.array_equal(g(X), g(O.ColVecs(X)))
    np.testing.assert_allclose(g(X), g(O.RowVecs(np.ascontiguousarray(X.T))), rtol=1e-15)


def test_rand_moments():
    # /root/reference/test/bayesian_linear_regression.jl:11-21 (2e5 samples instead of 1e6)
    rng = _rng(9)
    N, D, S = 11, 3, 200_000
    X, mw, Lw, Sy = O.generate_toy_problem(rng, N, D)
    Y = O.rand(mw, Lw, X, Sy, rng.standard_normal((D, S)), rng.standard_normal((N, S)))
    m_emp = Y.mean(axis=1)
    Yc = Y - m_emp[:, None]
    np.testing.assert_allclose(O.mean(mw, X), m_emp, atol=2e-2, rtol=2e-2)
    np.testing.assert_allclose(O.cov(mw, Lw, X, Sy), Yc @ Yc.T / S, atol=3e-2, rtol=3e-2)


def test_var_is_diag_cov_and_length_check():
    rng = _rng(10)
    X, mw, Lw, Sy = O.generate_toy_problem(rng, 11, 3)
    np.testing.assert_allclose(O.var(mw, Lw, X, Sy), np.diag(O.cov(mw, Lw, X, Sy)), rtol=1e-13)
    with pytest.raises(ValueError):
        O.logpdf_literal(mw, Lw, X, Sy, np.zeros(10))  # :74
    with pytest.raises(np.linalg.LinAlgError):
        O.logpdf_literal(mw, -Lw, X, Sy, np.zeros(11))  # PosDefException


def test_closed_form_gradient_matches_finite_differences():
    """The gradient formulas the GPU path implements (oracle.logpdf_grad; SURVEY.md 8f rank 1: what AD of the reference's
    logpdf, src/bayesian_linear_regression.jl:55-58, produces) against central differences of the LITERAL op sequence."""
    rng = np.random.default_rng(77)
    N, D = 13, 5
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    y = rng.standard_normal(N)
    lp, g = O.logpdf_grad(mw, Lw, X, s, y)
    assert lp == pytest.approx(O.logpdf_literal(mw, Lw, X, s, y), rel=1e-12)

    def fd(fun, x0, eps=1e-6):
        out = np.zeros_like(x0)
        it = np.nditer(x0, flags=["multi_index"])
        for _ in it:
            i = it.multi_index
            xp, xm = x0.copy(), x0.copy()
            xp[i] += eps
            xm[i] -= eps
            out[i] = (fun(xp) - fun(xm)) / (2 * eps)
        return out

    tol = dict(rtol=1e-6, atol=1e-6)  # central differences with eps = 1e-6 carry ~1e-7 of noise on O(1) values
    np.testing.assert_allclose(g["y"], fd(lambda v: O.logpdf_literal(mw, Lw, X, s, v), y), **tol)
    np.testing.assert_allclose(g["mw"], fd(lambda v: O.logpdf_literal(v, Lw, X, s, y), mw), **tol)
    np.testing.assert_allclose(g["X"], fd(lambda v: O.logpdf_literal(mw, Lw, v, s, y), X), **tol)
    np.testing.assert_allclose(g["s"], fd(lambda v: O.logpdf_literal(mw, Lw, X, v, y), s), **tol)
    # symmetric perturbations of the precision: dL = sum_ij G_ij dLw_ij with G symmetric
    def f_sym(v):
        return O.logpdf_literal(mw, 0.5 * (v + v.T), X, s, y)
    np.testing.assert_allclose(g["Lw"], fd(f_sym, Lw), rtol=1e-5, atol=1e-6)
    # isotropic noise: the scalar gradient is the sum of the per-observation ones
    lp2, g2 = O.logpdf_grad(mw, Lw, X, np.float64(0.3), y)
    d_iso = (O.logpdf_literal(mw, Lw, X, np.float64(0.3 + 1e-6), y) - O.logpdf_literal(mw, Lw, X, np.float64(0.3 - 1e-6), y)) / 2e-6
    assert np.sum(g2["s"]) == pytest.approx(d_iso, rel=1e-6)


def test_rand_pullback_matches_finite_differences():
    """The reverse-mode rule of rand (oracle.rand_pullback; what Zygote derives through src/bayesian_linear_regression.jl:49-53 in
    README.md:56-60) against central differences of the LITERAL rand with the draws held fixed: dense, PDMat-factor and diagonal
    prior precision."""
    rng = np.random.default_rng(78)
    N, D, S = 9, 5, 3
    X, mw, Lw, s = O.generate_toy_problem(rng, N, D, dense_noise_cov=False)
    Z1, Z2, Yb = rng.standard_normal((D, S)), rng.standard_normal((N, S)), rng.standard_normal((N, S))
    f = lambda mw_, Lw_, X_: float(np.sum(Yb * O.rand(mw_, Lw_, X_, s, Z1, Z2)))

    def fd(fun, x0, eps=1e-6):
        out = np.zeros_like(x0)
        it = np.nditer(x0, flags=["multi_index"])
        for _ in it:
            i = it.multi_index
            xp, xm = x0.copy(), x0.copy()
            xp[i] += eps
            xm[i] -= eps
            out[i] = (fun(xp) - fun(xm)) / (2 * eps)
        return out

    tol = dict(rtol=1e-6, atol=1e-6)
    g = O.rand_pullback(mw, Lw, X, s, Z1, Yb)
    np.testing.assert_allclose(g["X"], fd(lambda v: f(mw, Lw, v), X), **tol)
    np.testing.assert_allclose(g["mw"], fd(lambda v: f(v, Lw, X), mw), **tol)
    np.testing.assert_allclose(g["Lw"], fd(lambda v: f(mw, 0.5 * (v + v.T), X), Lw), rtol=1e-5, atol=1e-6)  # symmetric perturbations
    # the factor tangent (PDMat prior: what flows into chol.factors): perturb U itself, Lw = U'U
    U = O.chol_upper(Lw)
    gU = fd(lambda v: f(mw, np.triu(v).T @ np.triu(v), X), U)
    np.testing.assert_allclose(g["U"], np.triu(gU), rtol=1e-5, atol=1e-6)
    # diagonal precision
    d = np.exp(0.3 * rng.standard_normal(D))
    gd = O.rand_pullback(mw, d, X, s, Z1, Yb)
    np.testing.assert_allclose(gd["Lw"], fd(lambda v: f(mw, v, X), d), **tol)
    # noise: sbar_n = sum_s Ybar[n, s] Z2[n, s] / (2 sqrt(s_n))
    fs = lambda v: float(np.sum(Yb * O.rand(mw, Lw, X, v, Z1, Z2)))
    np.testing.assert_allclose(np.sum(Yb * Z2, axis=1) / (2 * np.sqrt(s)), fd(fs, s), **tol)
