"""Randomised parity sweep of the kernels added in round 4, through the C ABI against the oracle:
  marg   marginals at D > 128 (block substitution on LDS tiles, both tile heights, batches, factor / dense priors, padded ldx) and D = 128
  i8     the int8-sliced Gram route at D = 128 (N in 512 .. 16415, prior mean on / off, rows of different scale, both layouts)
  multi  logpdf(fx, Y::Matrix) (temporaries from the side buffer, parallel reductions)
  rand   rand(rng, fx, S) with given normals (rotated fragment images of the MFMA projection)
  grad   the evidence gradient at D = 128 (product form) and D > 128 with the regressors of a batch sharing every launch (both layouts, padded leading
         dimensions, all prior kinds, optional outputs left out, a failing regressor inside the group)
Not part of the test suite (minutes of GPU time): python tools/fuzz_round4.py [marg|i8|multi|rand|grad] [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd as B
from blr_amd import _abi
from oracle import blr_oracle as O


def marg(rng, case):
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    Ds = [128, 144, 160, 256, 272, 400, 512, 576] + ([1024, 1040, 1152] if dtype == np.float32 else [130, 300])
    D = int(rng.choice(Ds))
    N = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 1000, 3000, 9000]))
    Bn = int(rng.choice([1, 1, 2, 3, 5]))
    if D >= 1024 and N > 1000:
        Bn = 1
    kind = rng.choice(["factor", "dense"])
    diag = rng.random() < 0.5
    pad = int(rng.choice([0, 0, 4, 8, 16]))
    want_mean = rng.random() < 0.8
    ldx = D + pad
    X = np.zeros((Bn, N, ldx), dtype=dtype)
    X[:, :, :D] = rng.standard_normal((Bn, N, D))
    mw = rng.standard_normal((Bn, D)).astype(dtype)
    Lw = np.empty((Bn, D, D)); arg = np.empty((Bn, D, D), dtype=dtype)
    for b in range(Bn):
        Bm = rng.standard_normal((D, D)) / np.sqrt(D)
        Lw[b] = (Bm @ Bm.T + np.eye(D)).astype(dtype)
        arg[b] = (O.chol_upper(Lw[b]).astype(dtype) if kind == "factor" else Lw[b].astype(dtype)).T
    s = np.exp(0.3 * rng.standard_normal((Bn, N))).astype(dtype) if diag else np.full((Bn, 1), 0.4, dtype=dtype)
    mean = np.full((Bn, N), -7.0, dtype=dtype); var = np.full((Bn, N), -7.0, dtype=dtype); info = np.full(Bn, 9, dtype=np.int32)
    h = _abi.default_handle()
    h.marginals_batched(dtype, _abi.MEM_HOST, _abi.LAYOUT_COLVECS, Bn, D, N, X, ldx, N * ldx, _abi.NOISE_DIAGONAL if diag else _abi.NOISE_ISOTROPIC, s,
                        N if diag else 1, _abi.PRIOR_UPPER_FACTOR if kind == "factor" else _abi.PRIOR_DENSE, mw, D, arg, D, D * D,
                        mean if want_mean else None, N, var, N, info)
    rt = 1e-9 if dtype == np.float64 else 5e-4
    assert np.all(info == 0), (case, info)
    for b in range(Bn):
        Xb = X[b, :, :D].T.astype(float)
        sb = s[b].astype(float) if diag else float(s[b, 0])
        v_o = O.var(mw[b].astype(float), Lw[b].astype(dtype).astype(float), Xb, sb)
        np.testing.assert_allclose(var[b], v_o, rtol=rt, err_msg=f"case {case}: var D={D} N={N} B={Bn} {kind} {dtype.__name__}")
        if want_mean:
            m_o = O.mean(mw[b].astype(float), Xb)
            np.testing.assert_allclose(mean[b], m_o, rtol=rt, atol=rt * 30, err_msg=f"case {case}: mean D={D} N={N}")
    return f"D={D} N={N} B={Bn} {kind} {'diag' if diag else 'iso'} {dtype.__name__} ldx=D+{pad}"


def i8(rng, case):
    D = 128
    N = int(rng.integers(512, 16416)) if rng.random() < 0.4 else int(rng.choice([512, 543, 544, 1000, 1024, 4096, 4097, 16384, 16415]))
    nb = int(rng.choice([1, 3, 8]))
    X = rng.standard_normal((nb, N, D))
    u = rng.random()
    if u < 0.2:    # values that came from float32: the three low digits of the 48-bit integers are constant
        X = X.astype(np.float32).astype(np.float64)
    elif u < 0.3:  # fixed-point values / a constant feature with a repeating binary expansion
        X = np.round(X * 8.0) / 8.0
        X[:, :, 11] = 0.1
    scale = np.ldexp(1.0, rng.integers(-20, 21, size=D)) if rng.random() < 0.5 else np.ones(D)
    X *= scale[None, None, :]
    w = rng.standard_normal((nb, D)) / scale[None, :]
    y = np.einsum("bnd,bd->bn", X, w) + np.sqrt(0.1) * rng.standard_normal((nb, N))
    if rng.random() < 0.2:
        X[0, N // 3, 7] *= 1e4  # breaks its row's bound: that regressor goes back to the fp64 kernel
    mw = rng.standard_normal((nb, D)) / scale[None, :] if rng.random() < 0.6 else np.zeros((nb, D))
    dpr = np.exp(0.3 * rng.standard_normal((nb, D))) / scale[None, :] ** 2
    pr = rng.random()
    factor = pr < 0.35  # a prior given by its upper factor U (U'U joins at the hand-over)
    dense = 0.35 <= pr < 0.6  # a dense precision (upper triangle read; its Cholesky before the launch)
    Lw_o = [np.diag(dpr[b]) for b in range(nb)]
    prior_arg, pk, ldl, strideL = dpr, _abi.PRIOR_DIAGONAL, 1, D
    if factor:
        Uc = np.empty((nb, D, D))
        for b in range(nb):
            Bm = rng.standard_normal((D, D)) / np.sqrt(D) / scale[None, :]
            Lw_o[b] = Bm.T @ Bm + np.diag(dpr[b])
            Uc[b] = O.chol_upper(Lw_o[b]).T
        prior_arg, pk, ldl, strideL = Uc, _abi.PRIOR_UPPER_FACTOR, D, D * D
    if dense:
        Lc = np.empty((nb, D, D))
        for b in range(nb):
            Bm = rng.standard_normal((D, D)) / np.sqrt(D) / scale[None, :]
            Lw_o[b] = Bm.T @ Bm + np.diag(dpr[b])
            Lc[b] = np.triu(Lw_o[b]).T  # column-major upper triangle; the lower one stays zero (never read)
        prior_arg, pk, ldl, strideL = Lc, _abi.PRIOR_DENSE, D, D * D
    diag = rng.random() < 0.4
    s = np.exp(rng.choice([0.3, 1.0, 2.0]) * rng.standard_normal((nb, N))) if diag else np.array([0.1])
    if diag:
        y = y + np.sqrt(s) * rng.standard_normal((nb, N))
    mp = np.zeros((nb, D)); Tp = np.zeros((nb, D, D)); Ap = np.zeros((nb, D, D)); lp = np.zeros(nb); info = np.full(nb, 9, dtype=np.int32)
    h = _abi.default_handle()
    rowv = rng.random() < 0.4  # RowVecs storage (N x D column-major, padded leading dimension): four feature rows per DMA piece
    if rowv:
        ld = N + 2 * int(rng.integers(0, 4)) + (N & 1)
        Xin = np.full((nb, D, ld), 1.0e30)
        Xin[:, :, :N] = X.transpose(0, 2, 1)
        lay, ldx, sx = _abi.LAYOUT_ROWVECS, ld, D * ld
    else:
        Xin, lay, ldx, sx = X, _abi.LAYOUT_COLVECS, D, N * D
    h.posterior_batched(np.float64, _abi.MEM_HOST, lay, nb, D, N, Xin, ldx, sx, y, N, _abi.NOISE_DIAGONAL if diag else _abi.NOISE_ISOTROPIC,
                        s, N if diag else 0, pk, mw, D, prior_arg, ldl, strideL, mp, D, Tp, D, D * D, Ap, D, D * D, lp, info)
    assert np.all(info == 0), (case, info)
    for b in range(nb):
        mw_o, T_o, A_o, lp_o = O.posterior_logpdf_direct(mw[b], Lw_o[b] if (factor or dense) else dpr[b], X[b].T, s[b] if diag else 0.1, y[b])
        dA = np.sqrt(np.diag(A_o))
        # (the evidence is a difference of terms of size y'y / s: an injected outlier makes them 1e6 times the evidence itself)
        assert abs(lp[b] - lp_o) <= 2e-10 * abs(lp_o) + 1e-12 * float((y[b] * y[b] / (s[b] if diag else 0.1)).sum()) * max(1.0, float(np.abs(X[b]).max() / np.abs(X[b]).mean()) ** 2 / 1e4), (case, b, lp[b], lp_o)
        assert (np.abs(Ap[b] - A_o) / np.outer(dA, dA)).max() <= (1e-11 if diag else 1e-13), case
        np.testing.assert_allclose(mp[b] * dA, mw_o * dA, rtol=1e-7, atol=1e-8 * np.abs(mw_o * dA).max(), err_msg=f"case {case}")
    return f"N={N} B={nb} mw={'yes' if np.any(mw) else 'no'} scaled={'yes' if np.any(scale != 1) else 'no'} noise={'diag' if diag else 'iso'} prior={'factor' if factor else ('dense' if dense else 'diag')} layout={'RowVecs' if rowv else 'ColVecs'}"


def multi(rng, case):
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    D = int(rng.choice([3, 16, 64, 100, 128, 130, 256, 300, 512]))
    N = int(rng.choice([5, 64, 100, 257, 1000, 4096]))
    S = int(rng.choice([1, 2, 7, 64, 130]))
    X = np.asfortranarray(rng.standard_normal((D, N)).astype(dtype))
    Y = np.asfortranarray(rng.standard_normal((N, S)).astype(dtype))
    mw = rng.standard_normal(D).astype(dtype)
    diag = rng.random() < 0.5
    s = np.exp(0.3 * rng.standard_normal(N)).astype(dtype) if diag else dtype(0.5)
    prior = rng.choice(["dense", "diag"])
    if prior == "diag":
        Lw = B.Diagonal(np.exp(0.3 * rng.standard_normal(D)).astype(dtype)); Lw_o = np.diag(np.asarray(Lw.diag, dtype=float))
    else:
        Bm = rng.standard_normal((D, D)) / np.sqrt(D); Lw = (Bm @ Bm.T + np.eye(D)).astype(dtype); Lw_o = Lw.astype(float)
    f = B.BayesianLinearRegressor(mw, Lw)
    lps = np.asarray(B.logpdf(f(X, s), Y))
    rt = 1e-9 if dtype == np.float64 else 2e-3
    for j in range(S):
        lp_o = O.logpdf_literal(mw.astype(float), Lw_o, X.astype(float), np.asarray(s, dtype=float), Y[:, j].astype(float))
        assert abs(lps[j] - lp_o) <= rt * abs(lp_o), (case, j, lps[j], lp_o, D, N, S, dtype.__name__)
    return f"D={D} N={N} S={S} {prior} {'diag' if diag else 'iso'} {dtype.__name__}"


def rand(rng, case):
    dtype = np.float64 if rng.random() < 0.5 else np.float32
    D = int(rng.choice([4, 32, 64, 128, 130, 256, 384, 1024]))
    N = int(rng.choice([1, 17, 128, 129, 500, 1000, 5000]))
    S = int(rng.choice([1, 3, 64, 65, 200]))
    pad = int(rng.choice([0, 0, 4, 16]))
    ldx = D + pad
    Xa = np.zeros((ldx, N), dtype=dtype, order="F"); Xa[:D] = rng.standard_normal((D, N))
    mw = rng.standard_normal(D).astype(dtype)
    Bm = rng.standard_normal((D, D)) / np.sqrt(D)
    Lw = (Bm @ Bm.T + np.eye(D)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal(N)).astype(dtype)
    Z1 = np.asfortranarray(rng.standard_normal((D, S)).astype(dtype)); Z2 = np.asfortranarray(rng.standard_normal((N, S)).astype(dtype))
    f64 = lambda a: np.asarray(a, dtype=float)
    Y_o = O.rand(f64(mw), f64(Lw), f64(Xa[:D]), f64(s), f64(Z1), f64(Z2))
    h = _abi.default_handle()
    kind = rng.choice(["dense", "factor"])
    Larg = np.asfortranarray(Lw) if kind == "dense" else np.asfortranarray(O.chol_upper(f64(Lw)).astype(dtype))
    Y = np.empty((N, S), dtype=dtype, order="F")
    h.rand(dtype, _abi.MEM_HOST, _abi.LAYOUT_COLVECS, D, N, S, Xa, ldx, _abi.NOISE_DIAGONAL, s, _abi.PRIOR_DENSE if kind == "dense" else _abi.PRIOR_UPPER_FACTOR,
           mw, Larg, D, Z1, D, Z2, N, Y, N)
    rt = 1e-9 if dtype == np.float64 else 3e-3
    np.testing.assert_allclose(Y, Y_o, rtol=rt, atol=rt * 10 * np.abs(Y_o).max(), err_msg=f"case {case}: D={D} N={N} S={S} {kind} {dtype.__name__}")
    return f"D={D} N={N} S={S} {kind} {dtype.__name__} ldx=D+{pad}"


def grad(rng, case):
    dtype = np.float32 if rng.random() < 0.4 else np.float64
    D = int(rng.choice([128, 128, 129, 130, 160, 256, 300, 384, 520]))  # (128, aligned ColVecs, N >= 64: the product form of the sweeps)
    N = int(rng.choice([1, 7, 64, 100, 129, 500, 1100, 2500]))
    Bn = int(rng.choice([1, 2, 3, 5, 9, 17]))
    rowv = rng.random() < 0.4
    diag = rng.random() < 0.6
    kind = rng.choice(["diag", "dense", "factor"])
    pad = int(rng.choice([0, 0, 2, 4, 6]))
    Xc = rng.standard_normal((Bn, D, N))  # D x N
    if rowv:  # N x D column-major: [D][ldx] in memory
        ldx = N + pad
        X = np.zeros((Bn, D, ldx), dtype=dtype); X[:, :, :N] = Xc
        dX = np.full((Bn, D, ldx), -7.0, dtype=dtype)
        sx = D * ldx
    else:     # D x N column-major: [N][ldx] in memory
        ldx = D + pad
        X = np.zeros((Bn, N, ldx), dtype=dtype); X[:, :, :D] = Xc.transpose(0, 2, 1)
        dX = np.full((Bn, N, ldx), -7.0, dtype=dtype)
        sx = N * ldx
    Xc = (X[:, :, :N] if rowv else X[:, :, :D].transpose(0, 2, 1)).astype(float)
    y = rng.standard_normal((Bn, N)).astype(dtype)
    s = np.exp(0.3 * rng.standard_normal((Bn, N))).astype(dtype) if diag else np.full((Bn, 1), 0.4, dtype=dtype)
    mw = (0.3 * rng.standard_normal((Bn, D))).astype(dtype)
    Lw_o = np.empty((Bn, D, D))
    if kind == "diag":
        arg = np.exp(0.2 * rng.standard_normal((Bn, D))).astype(dtype)
        for b in range(Bn):
            Lw_o[b] = np.diag(arg[b].astype(float))
        pk, ldl, sl = _abi.PRIOR_DIAGONAL, 1, D
    else:
        arg = np.empty((Bn, D, D), dtype=dtype)
        for b in range(Bn):
            Bm = rng.standard_normal((D, D)) / np.sqrt(D)
            L = (Bm @ Bm.T + np.eye(D)).astype(dtype)
            if kind == "factor":
                U = O.chol_upper(L.astype(float)).astype(dtype)
                arg[b] = U.T
                Lw_o[b] = U.astype(float).T @ U.astype(float)
            else:
                arg[b] = L
                Lw_o[b] = L.astype(float)
        pk, ldl, sl = (_abi.PRIOR_UPPER_FACTOR if kind == "factor" else _abi.PRIOR_DENSE), D, D * D
    bad = int(rng.integers(0, Bn)) if (Bn > 1 and rng.random() < 0.3) else None
    if bad is not None:
        if kind == "diag":
            arg[bad, 3] = -1.0
        else:
            arg[bad, 3, 3] = -arg[bad, 3, 3] if kind == "factor" else -5.0
    want_ai = rng.random() < 0.6
    want_dx = rng.random() < 0.8
    want_mwp, want_dmw, want_dyds = rng.random() < 0.7, rng.random() < 0.8, rng.random() < 0.8
    lp = np.zeros(Bn); info = np.full(Bn, 9, dtype=np.int32)
    dy = np.zeros_like(y); ds = np.zeros((Bn, N), dtype=dtype); dmw = np.zeros_like(mw); mwp = np.zeros_like(mw)
    Ai = np.zeros((Bn, D, D), dtype=dtype) if want_ai else None
    h = _abi.default_handle()
    h.logpdf_grad_batched(dtype, _abi.MEM_HOST, _abi.LAYOUT_ROWVECS if rowv else _abi.LAYOUT_COLVECS, Bn, D, N, X, ldx, sx, y, N,
                          _abi.NOISE_DIAGONAL if diag else _abi.NOISE_ISOTROPIC, s, N if diag else 1, pk, mw, D, arg, ldl, sl, lp,
                          dX if want_dx else None, ldx, sx, dy if want_dyds else None, N, ds if want_dyds else None, N, dmw if want_dmw else None, D,
                          mwp if want_mwp else None, D, Ai, D, D * D, info)
    rt = 1e-8 if dtype == np.float64 else 5e-3
    tag = f"D={D} N={N} B={Bn} {'RowVecs' if rowv else 'ColVecs'} {kind} {'diag' if diag else 'iso'} {dtype.__name__} pad={pad} bad={bad} out={int(want_dx)}{int(want_ai)}{int(want_mwp)}{int(want_dmw)}{int(want_dyds)}"
    for b in range(Bn):
        if b == bad:
            assert info[b] != 0, (case, tag, info)
            continue
        assert info[b] == 0, (case, tag, info)
        if b > 3 and b != Bn - 1:
            continue
        sb = s[b].astype(float) if diag else float(s[b, 0])
        lp_o, g_o = O.logpdf_grad(mw[b].astype(float), Lw_o[b], Xc[b], sb, y[b].astype(float))
        assert abs(lp[b] - lp_o) <= (1e-10 if dtype == np.float64 else 3e-4) * abs(lp_o), (case, tag, lp[b], lp_o)
        got = []
        if want_dyds:
            got.append((dy[b], g_o["y"]))
            if diag:
                got.append((ds[b], g_o["s"]))
            else:  # isotropic noise: the scalar gradient is the SUM of the per-observation terms -- O(1) terms that cancel (0.07 out of
                # 500 terms in one fp32 case): measured against the sum of their magnitudes
                assert abs(float(ds[b].astype(float).sum()) - float(np.sum(g_o["s"]))) <= rt * float(np.abs(g_o["s"]).sum()), (case, tag)
        if want_dmw:
            got.append((dmw[b], g_o["mw"]))
        if want_mwp:
            got.append((mwp[b], g_o["mw_post"]))
        if want_dx:
            got.append(((dX[b][:, :N] if rowv else dX[b][:, :D].T), g_o["X"]))
        if want_ai:
            got.append((Ai[b], g_o["Ainv"]))
        for u, v in got:
            np.testing.assert_allclose(u, v, rtol=rt, atol=rt * np.abs(v).max(), err_msg=f"case {case}: {tag}")
    if want_dx and pad:  # the padding of dX is not written
        assert np.all((dX[:, :, N:] if rowv else dX[:, :, D:]) == -7.0), (case, tag)
    return tag


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "marg"
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 2027)
    fn = {"marg": marg, "i8": i8, "multi": multi, "rand": rand, "grad": grad}[which]
    for c in range(cases):
        d = fn(rng, c)
        if c % 10 == 0:
            print(f"case {c}: {d} ok", flush=True)
    print(f"{cases} {which} cases ok")
