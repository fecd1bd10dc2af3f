"""Randomised parity sweep through the C ABI against the oracle (shapes, leading dimensions, layouts, priors, noise kinds).
Not part of the test suite (minutes of GPU time): python tools/fuzz_parity.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd as B
from blr_amd import _abi
from oracle import blr_oracle as O


def one(rng, case):
    D = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 48, 63, 64, 65, 100, 127, 128, 129, 130, 200, 255, 256, 257, 300, 384, 385, 512, 640, 1000, 1024]))
    N = int(rng.choice([1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 257, 400, 777, 1024, 2048]))
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    layout = _abi.LAYOUT_COLVECS if rng.random() < 0.6 else _abi.LAYOUT_ROWVECS
    pad = int(rng.choice([0, 0, 1, 3, 8]))
    noise = "diag" if rng.random() < 0.6 else "iso"
    prior = rng.choice(["dense", "diag", "factor"])
    X = rng.standard_normal((D, N))
    mw = 0.5 * rng.standard_normal(D) if rng.random() < 0.7 else np.zeros(D)  # (a zero prior mean takes its own branch at D > 128)
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = Bm @ Bm.T + np.eye(D)
    if prior == "diag":
        dvec = np.exp(0.4 * rng.standard_normal(D)); Lw = np.diag(dvec)
    s = np.exp(0.4 * rng.standard_normal(N)) if noise == "diag" else np.float64(0.3 + rng.random())
    y = rng.standard_normal(N)
    Xd, mwd, Lwd, sd, yd = (np.asarray(a, dtype=dtype) for a in (X, mw, Lw, s, y))
    f64 = lambda a: np.asarray(a, dtype=float)
    mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(f64(mwd), f64(Lwd), f64(Xd), f64(sd), f64(yd))
    # ABI call with padded leading dimension
    if layout == _abi.LAYOUT_COLVECS:
        ldx = D + pad; Xa = np.zeros((ldx, N), dtype=dtype, order="F"); Xa[:D, :] = Xd
    else:
        ldx = N + pad; Xa = np.zeros((ldx, D), dtype=dtype, order="F"); Xa[:N, :] = Xd.T
    if prior == "dense":
        pk, Larg, ldl = _abi.PRIOR_DENSE, np.asfortranarray(Lwd), D
    elif prior == "diag":
        pk, Larg, ldl = _abi.PRIOR_DIAGONAL, np.ascontiguousarray(np.diag(Lwd)), 1
    else:
        pk, Larg, ldl = _abi.PRIOR_UPPER_FACTOR, np.asfortranarray(O.chol_upper(f64(Lwd)).astype(dtype)), D
    sv = np.atleast_1d(sd).astype(dtype)
    nk = _abi.NOISE_DIAGONAL if noise == "diag" else _abi.NOISE_ISOTROPIC
    h = _abi.default_handle()
    mw_p = np.empty(D, dtype=dtype); Tp = np.zeros((D, D), dtype=dtype, order="F"); Ap = np.zeros((D, D), dtype=dtype, order="F")
    lp = np.zeros(1); info = np.zeros(1, dtype=np.int32)
    h.posterior_batched(dtype, _abi.MEM_HOST, layout, 1, D, N, Xa, ldx, 0, yd, 0, nk, sv, 0, pk, mwd, 0, Larg, ldl, 0, mw_p, D, Tp, D,
                        D * D, Ap, D, D * D, lp, info)
    tol = 1e-9 if dtype == np.float64 else 5e-3
    tag = f"case {case}: D={D} N={N} {np.dtype(dtype).name} layout={layout} pad={pad} noise={noise} prior={prior}"
    assert info[0] == 0, tag
    assert abs(lp[0] - lp_o) <= (1e-10 if dtype == np.float64 else 5e-4) * max(1.0, abs(lp_o)), (tag, lp[0], lp_o)
    np.testing.assert_allclose(mw_p, mw_o, rtol=tol, atol=tol * 10, err_msg=tag)
    np.testing.assert_allclose(Tp, T_o, rtol=tol, atol=tol * 10, err_msg=tag)
    np.testing.assert_allclose(Ap, A_o, rtol=tol, atol=tol * 10, err_msg=tag)
    # marginals of the posterior at fresh inputs
    Ns = int(rng.choice([1, 7, 64, 130]))
    Xs = np.asfortranarray(rng.standard_normal((D, Ns)).astype(dtype))
    mean = np.empty(Ns, dtype=dtype); var = np.empty(Ns, dtype=dtype)
    h.marginals_batched(dtype, _abi.MEM_HOST, _abi.LAYOUT_COLVECS, 1, D, Ns, Xs, D, 0, _abi.NOISE_ISOTROPIC, np.array([0.1], dtype=dtype), 0,
                        _abi.PRIOR_UPPER_FACTOR, mw_p, 0, Tp, D, 0, mean, Ns, var, Ns, info)
    m_o, v_o = O.marginals_direct(f64(mw_p), f64(Tp), f64(Xs), np.float64(dtype(0.1)))
    np.testing.assert_allclose(mean, m_o, rtol=tol * 10, atol=tol * 100, err_msg=tag)
    np.testing.assert_allclose(var, v_o, rtol=tol * 10, atol=tol * 10, err_msg=tag)
    # gradient
    if rng.random() < 0.5:
        fx = B.BayesianLinearRegressor(mwd, Lwd if prior != "diag" else B.Diagonal(np.diag(Lwd)))(np.asfortranarray(Xd), sd if noise == "diag" else float(sd))
        lp_g, g = B.logpdf_and_gradient(fx, yd)
        _, g_o = O.logpdf_grad(f64(mwd), f64(Lwd), f64(Xd), f64(sd), f64(yd))
        gt = 1e-7 if dtype == np.float64 else 2e-2
        np.testing.assert_allclose(g["X"], g_o["X"], rtol=gt, atol=gt * np.abs(g_o["X"]).max(), err_msg=tag)
        np.testing.assert_allclose(g["mw"], g_o["mw"], rtol=gt, atol=gt * max(1e-30, np.abs(g_o["mw"]).max()), err_msg=tag)
    return tag


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    for c in range(cases):
        try:
            one(rng, c)
        except Exception as e:  # noqa: BLE001
            print("FAIL", type(e).__name__, str(e)[:600])
            raise SystemExit(1)
    print(f"{cases} random cases passed")
