"""Randomised parity sweep through the C ABI against the oracle (shapes, leading dimensions, layouts, priors, noise kinds).
Not part of the test suite (minutes of GPU time): python tools/fuzz_parity.py [cases] [seed] [group]
("group": batches of regressors at D > 128 instead of single updates)"""
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


def group(rng, case):
    """Batched regressors at D > 128 (posterior_large_group): strides with gaps, both layouts, one regressor in the batch may
    carry a prior or a noise vector that is not positive."""
    nb = int(rng.choice([2, 3, 5, 8, 16, 17, 23, 33]))
    D = int(rng.choice([129, 130, 200, 256, 257, 300, 384, 512]))
    N = int(rng.choice([1, 5, 64, 100, 257, 400, 777, 1500]))
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    layout = _abi.LAYOUT_COLVECS if rng.random() < 0.6 else _abi.LAYOUT_ROWVECS
    pad = int(rng.choice([0, 0, 1, 4]))
    gap = int(rng.choice([0, 0, 4, 8, 3]))  # extra elements between the regressors' X blocks (3: the group path steps aside)
    noise = "diag" if rng.random() < 0.6 else "iso"
    prior = str(rng.choice(["dense", "diag", "factor"]))
    rows, cols = (D, N) if layout == _abi.LAYOUT_COLVECS else (N, D)
    ldx = rows + pad
    strideX = ldx * cols + gap
    Xbuf = np.zeros(nb * strideX, dtype=dtype)
    Xs = []
    for b in range(nb):
        Xb = rng.standard_normal((D, N)).astype(dtype)
        Xs.append(Xb)
        blk = Xbuf[b * strideX:b * strideX + ldx * cols].reshape((cols, ldx))  # column-major (ldx x cols)
        blk[:, :rows] = Xb.T if layout == _abi.LAYOUT_COLVECS else Xb
    mw = (0.3 * rng.standard_normal((nb, D))).astype(dtype)
    y = rng.standard_normal((nb, N)).astype(dtype)
    ns = N if noise == "diag" else 1
    sv = (np.exp(0.4 * rng.standard_normal((nb, ns))) if noise == "diag" else 0.3 + rng.random((nb, 1))).astype(dtype)
    bad = int(rng.integers(nb)) if rng.random() < 0.4 else None
    bad_kind = str(rng.choice(["prior", "noise"])) if noise == "diag" else "prior"
    want = [0] * nb
    if prior == "diag":
        Larg = np.exp(0.3 * rng.standard_normal((nb, D))).astype(dtype)
        if bad is not None and bad_kind == "prior":
            k = int(rng.integers(D)); Larg[bad, k] = 0.0; want[bad] = k + 1
        Ld = [np.diag(Larg[b].astype(float)) for b in range(nb)]
        pk, ldl, strideL = _abi.PRIOR_DIAGONAL, 1, D
    else:
        Ms = []
        for b in range(nb):
            Bm = rng.standard_normal((D, D)) / np.sqrt(D)
            Ms.append(Bm @ Bm.T + np.eye(D))
        if prior == "dense":
            if bad is not None and bad_kind == "prior":
                k = int(rng.integers(D)); Lc = np.linalg.cholesky(Ms[bad]); Lc[k, k] = 0.0
                Ms[bad] = Lc @ Lc.T; Ms[bad][k, k] -= 1.0; want[bad] = k + 1
            Larg = np.stack([M.astype(dtype) for M in Ms])
            Ld = [Larg[b].astype(float) for b in range(nb)]
            pk = _abi.PRIOR_DENSE
        else:
            U = [O.chol_upper(M).astype(dtype) for M in Ms]
            if bad is not None and bad_kind == "prior":
                k = int(rng.integers(D)); U[bad][k, k] = -1.0; want[bad] = k + 1
            Larg = np.stack([u.T.copy() for u in U])
            Ld = [u.astype(float).T @ u.astype(float) for u in U]
            pk = _abi.PRIOR_UPPER_FACTOR
        ldl, strideL = D, D * D
    if bad is not None and bad_kind == "noise":
        k = int(rng.integers(N)); sv[bad, k] = -0.5; want[bad] = k + 1
    nk = _abi.NOISE_DIAGONAL if noise == "diag" else _abi.NOISE_ISOTROPIC
    h = _abi.default_handle()
    mw_p = np.zeros((nb, D), dtype=dtype); Tp = np.zeros((nb, D, D), dtype=dtype); Ap = np.zeros((nb, D, D), dtype=dtype)
    lp = np.zeros(nb); info = np.full(nb, 7, dtype=np.int32)
    # optional outputs: any subset may be absent (none of mw' / T: the evidence-only finish)
    w_m, w_T, w_A = (bool(v) for v in (rng.random(3) < 0.7))
    h.posterior_batched(dtype, _abi.MEM_HOST, layout, nb, D, N, Xbuf, ldx, strideX, y, N, nk, sv, ns, pk, mw, D, Larg, ldl, strideL,
                        mw_p if w_m else None, D, Tp if w_T else None, D, D * D, Ap if w_A else None, D, D * D, lp, info)
    tag = f"group case {case}: B={nb} D={D} N={N} {np.dtype(dtype).name} layout={layout} pad={pad} gap={gap} noise={noise} prior={prior} bad={bad}/{bad_kind} outputs={w_m, w_T, w_A}"
    assert info.tolist() == want, (tag, info.tolist(), want)
    tol = 1e-9 if dtype == np.float64 else 5e-3
    f64 = lambda a: np.asarray(a, dtype=float)
    for b in range(nb):
        if want[b]:
            assert np.isnan(lp[b]), tag
            continue
        sb = f64(sv[b]) if noise == "diag" else np.float64(sv[b, 0])
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(f64(mw[b]), Ld[b], f64(Xs[b]), sb, f64(y[b]))
        assert abs(lp[b] - lp_o) <= (1e-10 if dtype == np.float64 else 5e-4) * max(1.0, abs(lp_o)), (tag, b, lp[b], lp_o)
        if w_m:
            np.testing.assert_allclose(mw_p[b], mw_o, rtol=tol, atol=tol * 10, err_msg=tag)
        if w_T:
            np.testing.assert_allclose(Tp[b].T, T_o, rtol=tol, atol=tol * 10, err_msg=tag)
        if w_A:
            np.testing.assert_allclose(Ap[b].T, A_o, rtol=tol, atol=tol * 10, err_msg=tag)
    return tag


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    fn = group if (len(sys.argv) > 3 and sys.argv[3] == "group") else one
    for c in range(cases):
        try:
            fn(rng, c)
        except Exception as e:  # noqa: BLE001
            print("FAIL", type(e).__name__, str(e)[:600])
            raise SystemExit(1)
    print(f"{cases} random cases passed")
