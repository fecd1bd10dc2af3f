"""Edge shapes of the multi-output evidence on the planes route (blr_logpdf_multi_f32 at D > 128: tiny N, 1 and 128 columns, D not a
multiple of 4 or 128, both noise kinds, zero and non-zero prior mean) against the oracle, column by column.  GPU box: python tools/edge_multi.py"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd as B
from oracle import blr_oracle as O
rng = np.random.default_rng(5)
bad = 0
for (D, N, S) in [(130, 1, 5), (130, 5, 1), (257, 17, 3), (129, 33, 128), (384, 100, 65), (200, 2049, 7)]:
    for noise in ("iso", "diag"):
        for zero in (True, False):
            X = np.asfortranarray(rng.standard_normal((D, N)).astype(np.float32))
            mw = np.zeros(D, np.float32) if zero else (0.3 * rng.standard_normal(D)).astype(np.float32)
            dvec = np.exp(0.2 * rng.standard_normal(D)).astype(np.float32)
            s = np.exp(0.3 * rng.standard_normal(N)).astype(np.float32) if noise == "diag" else np.float32(0.7)
            Y = rng.standard_normal((N, S)).astype(np.float32)
            f64 = lambda a: np.asarray(a, float)
            fx = B.BayesianLinearRegressor(mw, B.Diagonal(dvec))(X, B.Diagonal(s) if noise == "diag" else s)
            lp, M = B.logpdf_columns(fx, Y, return_means=True)
            for j in range(S):
                m_o, _, _, lp_o = O.posterior_logpdf_direct(f64(mw), f64(dvec), f64(X), f64(s), f64(Y[:, j]))
                e1 = abs(lp[j] - lp_o) / abs(lp_o); e2 = np.linalg.norm(M[:, j] - m_o) / max(np.linalg.norm(m_o), 1e-30)
                if e1 > 3e-5 or e2 > 3e-4:
                    bad += 1; print("BAD", D, N, S, noise, zero, j, e1, e2)
print("edge cases done, bad =", bad, "route", B._abi.default_handle().last_route())
