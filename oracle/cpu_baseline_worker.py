"""One CPU-baseline worker (test/bench infrastructure): times one posterior update + log evidence of the reference's
algorithm on the host for a bounded time.  Prints "<regressors done> <seconds used>".

    python cpu_baseline_worker.py D N seconds seed [form] [blas_threads]

form = literal : the reference's user sequence logpdf(fX, y); posterior(fX, y), literal op order of reference
                 src/bayesian_linear_regression.jl:55-89 (each call recomputes :72-89, as the reference does)
       direct  : the algebraically equivalent one-pass Gram form -- what a tuned CPU code would run
blas_threads   : OpenBLAS threads of this process (default 1: independent regressors are spread over processes)
"""
import os
import sys
import time

_threads = sys.argv[6] if len(sys.argv) > 6 else "1"
for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[k] = _threads
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from oracle import blr_oracle as O  # noqa: E402


def main():
    D, N, seconds, seed = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    form = sys.argv[5] if len(sys.argv) > 5 else "literal"
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.standard_normal((D, N)))
    w = rng.standard_normal(D)
    y = X.T @ w + np.sqrt(0.1) * rng.standard_normal(N)
    mw, Lw, s = np.zeros(D), np.eye(D), 0.1 * np.ones(N)

    def one():
        if form == "direct":
            O.posterior_logpdf_direct(mw, Lw, X, s, y)
        else:
            O.logpdf_literal(mw, Lw, X, s, y)
            O.posterior_literal(mw, Lw, X, s, y)

    one()  # warm-up
    done, used = 0, 0.0
    while used < seconds:
        t0 = time.perf_counter()
        one()
        used += time.perf_counter() - t0
        done += 1
    print(done, used)


if __name__ == "__main__":
    main()
