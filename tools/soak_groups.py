"""Soak of the D > 128 path on ONE handle: batched calls of changing batch size, shape and precision interleaved with single
updates and large-D draws (which share the wavefront solve's exchange buffer and counters), each checked against the same
call run one regressor at a time.   python tools/soak_groups.py [rounds] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd as B
from blr_amd import _abi as a


def batched(h, dtype, nb, D, N, X, y, s, mw, d):
    mp = np.zeros((nb, D), dtype); T = np.zeros((nb, D, D), dtype); lp = np.zeros(nb); info = np.full(nb, 3, np.int32)
    h.posterior_batched(dtype, a.MEM_HOST, a.LAYOUT_COLVECS, nb, D, N, X, D, N * D, y, N, a.NOISE_DIAGONAL, s, N, a.PRIOR_DIAGONAL, mw, D, d, 1, D,
                        mp, D, T, D, D * D, None, D, D * D, lp, info)
    return mp, T, lp, info


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
    h = a.default_handle()
    for it in range(rounds):
        dtype = np.float64 if rng.random() < 0.5 else np.float32
        nb = int(rng.choice([1, 2, 3, 7, 16, 40, 130]))
        D = int(rng.choice([129, 200, 256, 384, 520]))
        N = int(rng.choice([3, 100, 513, 1200]))
        if nb * D * D > 3e7:
            nb = max(1, int(3e7 // (D * D)))
        X = rng.standard_normal((nb, N, D)).astype(dtype); y = rng.standard_normal((nb, N)).astype(dtype)
        s = np.exp(0.3 * rng.standard_normal((nb, N))).astype(dtype)
        mw = (0.2 * rng.standard_normal((nb, D))).astype(dtype); d = np.exp(0.3 * rng.standard_normal((nb, D))).astype(dtype)
        bad = int(rng.integers(nb)) if rng.random() < 0.3 else None
        if bad is not None:
            d[bad, int(rng.integers(D))] = -1.0
        h.set_option("CHAIN_BATCH", None)
        mp, T, lp, info = batched(h, dtype, nb, D, N, X, y, s, mw, d)
        h.set_option("CHAIN_BATCH", "1")
        mp1, T1, lp1, info1 = batched(h, dtype, nb, D, N, X, y, s, mw, d)
        h.set_option("CHAIN_BATCH", None)
        tag = f"round {it}: B={nb} D={D} N={N} {np.dtype(dtype).name} bad={bad}"
        assert info.tolist() == info1.tolist(), (tag, info.tolist(), info1.tolist())
        ok = info == 0
        eps = 1e-11 if dtype == np.float64 else 2e-4
        np.testing.assert_allclose(lp[ok], lp1[ok], rtol=eps, err_msg=tag)
        np.testing.assert_allclose(mp[ok], mp1[ok], rtol=0, atol=eps * 100 * max(1e-30, np.abs(mp1[ok]).max()) if ok.any() else 0, err_msg=tag)
        np.testing.assert_allclose(T[ok], T1[ok], rtol=0, atol=eps * 10 * max(1e-30, np.abs(T1[ok]).max()) if ok.any() else 0, err_msg=tag)
        if bad is not None:
            assert np.isnan(lp[bad]) and info[bad] > 0, tag
        if it % 3 == 0:  # draws through the same wavefront solve: S right-hand sides, one factor
            Dd = int(rng.choice([200, 384]))
            Bm = rng.standard_normal((Dd, Dd)) / np.sqrt(Dd)
            f = B.BayesianLinearRegressor(np.zeros(Dd), Bm @ Bm.T + np.eye(Dd))
            Xs = np.asfortranarray(rng.standard_normal((Dd, 50)))
            Y = B.rand(np.random.default_rng(5), f(Xs, 0.1), 9)
            assert Y.shape == (50, 9) and np.isfinite(Y).all(), tag
    print(f"{rounds} rounds passed")


if __name__ == "__main__":
    main()
