"""Generates tests/golden/blr_golden.json: small fixtures (inputs + expected outputs) for the posterior/logpdf path.

The reference is Julia and cannot run in the build image, and its own tests hold no numeric fixtures except one doctest
vector, so these fixtures are produced by the CPU oracle (oracle/blr_oracle.py) on the toy-problem CONSTRUCTION of
/root/reference/test/test_utils.jl:4-10 with a seeded NumPy generator, and every log density is cross-checked against a
50-digit mpmath evaluation of the naive N x N Gaussian formula of /root/reference/test/bayesian_linear_regression.jl:28-37
before it is written; the posterior mean, factor and precision that are stored come from a 50-digit mpmath solve of the
normal equations (posterior_mp) and both oracle forms are asserted against it.  Cases: (N, D) = (11, 3), (13, 7), (11, 2) as in the reference tests, the README example shape
(10, 2) with a Diagonal prior and heteroscedastic noise, one DENSE-noise toy problem exactly as the reference builds them
(test/test_utils.jl:7-8; pins the oracle's :79-82 dense branch), and the doctest of src/basis_function_regression.jl:11-28.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import blr_oracle as O  # noqa: E402


def posterior_mp(mw, Lw, X, Sy, y, dps=50):
    """mw' = (Lw + X S X')^-1 (Lw mw + X S y),  Lw' = Lw + X S X',  T = chol(Lw').U  with S = inv(Sy), evaluated in mpmath at
    `dps` digits -- an evaluation of the posterior that shares no code (and no algorithm: normal equations by LU) with the
    oracle's two forms.  Reference: the quantities returned at /root/reference/src/bayesian_linear_regression.jl:60-69, :92."""
    import mpmath as mp

    mp.mp.dps = dps
    D, N = X.shape
    Xm = mp.matrix(X.tolist())
    Lm = mp.matrix(O.dense_precision(Lw, D, np.float64).tolist())
    Sm = mp.inverse(mp.matrix(O.dense_noise(Sy, N, np.float64).tolist()))
    A = Lm + Xm * Sm * Xm.T
    rhs = Lm * mp.matrix(mw.tolist()) + Xm * Sm * mp.matrix(y.tolist())
    m = mp.lu_solve(A, rhs)
    L = mp.cholesky(A)  # lower, A = L L'
    to_np = lambda M: np.array([[float(M[i, j]) for j in range(M.cols)] for i in range(M.rows)])
    return np.array([float(v) for v in m]), to_np(L).T, to_np(A)


def case(name, rng, N, D, prior, noise):
    X = rng.standard_normal((D, N))
    mw = rng.standard_normal(D)
    if prior == "diagonal":
        dvec = np.exp(0.5 * rng.standard_normal(D))
        Lw = np.diag(dvec)
    else:
        Bm = rng.standard_normal((D, D))
        Lw = Bm @ Bm.T + np.eye(D)
    if noise == "dense":  # the reference's own toy problems: Sy = C C' + I with C = 0.1 randn (test/test_utils.jl:7-8)
        C = 0.1 * rng.standard_normal((N, N))
        s = C @ C.T + np.eye(N)
    else:
        s = np.exp(rng.standard_normal(N)) if noise == "diagonal" else np.float64(0.37)
    y = rng.standard_normal(N)
    mw_p, T, A = O.posterior_literal(mw, Lw, X, s, y)
    lp = O.logpdf_literal(mw, Lw, X, s, y)
    lp_mp = O.logpdf_naive_mp(mw, Lw, X, s, y)
    assert abs(lp - lp_mp) <= 1e-12 * abs(lp_mp), (name, lp, lp_mp)
    if noise != "dense":
        mw_d, T_d, A_d, lp_d = O.posterior_logpdf_direct(mw, Lw, X, s, y)
        assert np.allclose(mw_d, mw_p, rtol=1e-11) and np.allclose(T_d, T, rtol=1e-11) and abs(lp_d - lp) <= 1e-12 * abs(lp)
    # independent 50-digit pin of the POSTERIOR (not only of the evidence): stored values are the mpmath ones
    mw_mp, T_mp, A_mp = posterior_mp(mw, Lw, X, s, y)
    assert np.allclose(mw_p, mw_mp, rtol=1e-11, atol=1e-13) and np.allclose(T, T_mp, rtol=1e-11, atol=1e-13)
    assert np.allclose(A, A_mp, rtol=1e-12, atol=1e-13)
    mw_p, T, A = mw_mp, T_mp, A_mp
    Xs = rng.standard_normal((D, 6))
    mean, var = O.mean(mw_p, Xs), O.var(mw_p, A, Xs, 0.05)
    Z1, Z2 = rng.standard_normal((D, 3)), rng.standard_normal((N, 3))
    Y = O.rand(mw, Lw, X, s, Z1, Z2)
    out = dict(name=name, N=N, D=D, prior=prior, noise=noise, X=X, mw=mw, Lw=Lw, s=np.asarray(s), y=y,
               logpdf=lp_mp, mw_post=mw_p, T_post=T, Lw_post=A, X_star=Xs, noise_star=0.05, mean_star=mean, var_star=var,
               Z1=Z1, Z2=Z2, Y_rand=Y)
    return {k: (v.tolist() if isinstance(v, np.ndarray) else (float(v) if isinstance(v, (np.floating, float)) else v))
            for k, v in out.items()}


def seeded_inputs(seed, D, N):
    """Synthetic ColVecs problem of SURVEY.md 8d from a seed (shared with tests/test_golden.py)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((D, N))
    w = rng.standard_normal(D)
    s = 0.1
    y = X.T @ w + np.sqrt(s) * rng.standard_normal(N)
    return X, y, s, np.zeros(D), np.ones(D)  # prior: mw = 0, Lw = I (diagonal)


def checksums(mw_post, T, A, lp):
    return dict(logpdf=float(lp), sum_mw=float(mw_post.sum()), sumsq_mw=float((mw_post ** 2).sum()), trace_A=float(np.trace(A)),
                fro_A=float(np.sqrt((A ** 2).sum())), logdet_A=float(2.0 * np.log(np.diag(T)).sum()))


def seeded_case(name, seed, D, N):
    X, y, s, mw, dpr = seeded_inputs(seed, D, N)
    mw_p, T, A = O.posterior_literal(mw, dpr, X, s, y)
    lp = O.logpdf_literal(mw, dpr, X, s, y)
    return dict(name=name, seed=seed, D=D, N=N, noise_var=s, checksums=checksums(mw_p, T, A, lp))


def main():
    rng = np.random.Generator(np.random.PCG64(20261002))
    cases = [
        case("toy_11_3_dense_prior_diag_noise", rng, 11, 3, "dense", "diagonal"),
        case("toy_13_7_dense_prior_diag_noise", rng, 13, 7, "dense", "diagonal"),
        case("toy_11_2_dense_prior_iso_noise", rng, 11, 2, "dense", "isotropic"),
        case("readme_10_2_diag_prior_hetero_noise", rng, 10, 2, "diagonal", "diagonal"),
        case("toy_13_7_dense_prior_dense_noise", rng, 13, 7, "dense", "dense"),
    ]
    # full-size shapes of BASELINE.json as seed + checksums: the inputs are regenerated from the seed, the
    # expected numbers come from the fp64 oracle (literal op sequence; the direct form agrees to 1e-11, asserted by the test)
    seeded = [seeded_case("c2_128_4096_iso", 424242, 128, 4096), seeded_case("c4_64_1024_iso", 434343, 64, 1024)]
    doctest = dict(name="doctest_basis_function_regression_jl_11_28", x=np.linspace(-1.0, 1.0, 5).tolist(),
                   mw=[0.0, 0.0], Lw_diag=[1.0, 1.0], noise=1e-18, var=[2.0, 1.25, 1.0, 1.25, 2.0])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "blr_golden.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py", seed=20261002, cases=cases, doctest=doctest, seeded=seeded), f, indent=0)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
