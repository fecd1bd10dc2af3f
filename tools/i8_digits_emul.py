#!/usr/bin/env python3
"""Exact-integer emulation of the digit arithmetic of fused_i8_kernel (blr_fused_i8.hpp) in NumPy: what the DROPPED digit pairs cost, for the
round-5 digits (bytes of Q minus 128, with and without their mean parts) and for balanced digits (bytes of Q + 0x8080808080 minus 128, round 6),
on several kinds of inputs; and the wrap + repair path.  Error = max |dropped products| / sqrt(G_ii G_jj).   python tools/i8_digits_emul.py"""
import numpy as np

rng = np.random.default_rng(1)
BAL = 0x8080808080


def wrap48(Q):
    return ((Q + (1 << 47)) & ((1 << 48) - 1)) - (1 << 47)


def digits(Q, balanced):
    Qp = wrap48(Q + BAL) if balanced else wrap48(Q)
    out = []
    for s in range(6):
        d = (Qp >> (8 * (5 - s))) & 0xFF
        d = np.where(d >= 128, d - 256, d) if s == 0 else d - 128
        out.append(d.astype(np.int64))
    return out


def quant(X):
    m = np.abs(X[:, :96]).max(axis=1)
    e = np.floor(np.log2(m)) + 2  # capacity 2^(E + 2)
    return np.rint(X * np.exp2(47 - e)[:, None]).astype(np.int64), e


def dropped(name, X, NG=6):
    D, N = X.shape
    Q, _ = quant(X)
    Q = np.clip(Q, -(1 << 47) + 1, int((1 << 47) * (1 - 2.0 ** -7)))
    G = Q.astype(np.float64) @ Q.astype(np.float64).T
    dg = np.sqrt(np.diag(G))
    res = []
    for bal in (False, True):
        d = digits(Q, bal)
        off = [0] * 6 if bal else [0] + [128] * 5
        assert (sum((d[s] + off[s]) * (1 << (8 * (5 - s))) for s in range(6)) == Q).all()
        err = np.zeros((D, D))
        errm = np.zeros((D, D))
        for s in range(6):
            for t in range(6):
                if s + t >= NG:
                    P = d[s].astype(np.float64) @ d[t].astype(np.float64).T  # exact (|.| < 2^53)
                    Pm = np.outer(d[s].sum(1), d[t].sum(1)) / N
                    if s == 3 and t == 3:  # the diagonal's share comes exactly from one more v_dot4
                        P = P - np.diag(np.diag(P))
                        Pm = Pm - np.diag(np.diag(Pm))
                    err += 2.0 ** (80 - 8 * (s + t)) * P
                    errm += 2.0 ** (80 - 8 * (s + t)) * (P - Pm)
        res.append((np.abs(err / np.outer(dg, dg)).max(), np.abs(errm / np.outer(dg, dg)).max()))
    print(f"{name:24s} bytes of Q - 128: dropped pairs {res[0][0]:.2e}, with their mean parts kept (round 5) {res[0][1]:.2e} | balanced (round 6): "
          f"{res[1][0]:.2e} (mean parts kept: {res[1][1]:.2e})")


def wrapped(name, X):
    """entries beyond the capacity: digits of the wrapped integer + the fp64 repair x x' - c c'"""
    D, N = X.shape
    Q, e = quant(X)
    cap = 1 << 47
    over = np.abs(Q) >= int(cap * (1 - 2.0 ** -7))
    d = digits(Q, True)
    c = sum(d[s] * (1 << (8 * (5 - s))) for s in range(6))  # what the digits stand for
    assert (c[~over] == Q[~over]).all()
    Gs = np.zeros((D, D))
    for s in range(6):
        for t in range(6):
            if s + t < 6:
                Gs += 2.0 ** (80 - 8 * (s + t)) * (d[s].astype(np.float64) @ d[t].astype(np.float64).T)
    Gs[np.diag_indices(D)] += 2.0 ** 32 * (d[3].astype(np.float64) ** 2).sum(1)
    sc = np.exp2(e - 47)
    Gs = Gs * np.outer(sc, sc)
    cols = np.where(over.any(axis=0))[0]
    Xq = Q * sc[:, None]  # the inputs on their rows' grids
    Xq[over] = X[over]
    cc = c * sc[:, None]
    for n in cols:
        Gs += np.outer(Xq[:, n], Xq[:, n]) - np.outer(cc[:, n], cc[:, n])
    Gx = np.array([[float(np.dot(X[i].astype(np.longdouble), X[j].astype(np.longdouble))) for j in range(D)] for i in range(D)])
    dg = np.sqrt(np.diag(Gx))
    print(f"{name:24s} {over.sum()} entries beyond their rows' capacity in {len(cols)} columns: G within {np.abs((Gs - Gx) / np.outer(dg, dg)).max():.2e} "
          f"of sqrt(G_ii G_jj), {np.abs(Gs - Gx).max() / np.abs(Gx).max():.2e} of max |G|")


if __name__ == "__main__":
    D, N = 128, 4096
    X = rng.standard_normal((D, N))
    dropped("gauss f64", X)
    dropped("gauss float32-origin", X.astype(np.float32).astype(np.float64))
    dropped("gauss float16-origin", X.astype(np.float16).astype(np.float64))
    dropped("integers 0..99", rng.integers(0, 100, (D, N)).astype(float))
    dropped("powers of two", np.exp2(rng.integers(-3, 3, (D, N))) * rng.choice([-1, 1], (D, N)))
    dropped("two decimals", np.round(rng.standard_normal((D, N)), 2))
    dropped("uniform [0, 1)", rng.random((D, N)))
    dropped("0.1 + 1e-3 N(0,1)", 0.1 + 1e-3 * rng.standard_normal((D, N)))
    dropped("integers / 3", rng.integers(-30, 30, (D, N)) / 3.0)
    dropped("integer + 1/3", rng.integers(-30, 30, (D, N)) + 1 / 3.0)
    X16 = rng.standard_normal((D, 16384))
    dropped("gauss f64, N = 16384", X16)
    wrapped("gauss f64", X)
    Xo = X.copy()
    Xo[rng.random((D, N)) < 2.0e-5] *= 6.0
    wrapped("gauss, 1e-5 of them x 6", Xo)
    Xs = X.copy()
    Xs[::8, 5] *= 6.0  # an entry 6 x larger among the first 96 columns of every eighth row: that row's grid is 4 - 8 x coarser
    dropped("coarse rows (x6 in col 5)", Xs)
