"""Host restatement of the three-way bf16 split the fp32 large-D Gram uses (csrc/blr_large.hpp, bf3_split_pack): what it promises,
checked without a GPU.  x = h + m + l with every level rounded to nearest-even to 8 significant bits; of the nine products of two split
numbers six are kept.  (The device test is test_large_d_fp32_gram_on_bf16_matrix_cores_vs_f32_route.)"""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (v_cvt_pk_bf16_f32)"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)      # exact: |r| <= half an ulp of h
    m = bf16_rne(r)
    r2 = (r - m).astype(np.float32)     # exact
    l = bf16_rne(r2)
    return h, m, l, r2


def test_three_way_split_is_exact_and_signed():
    rng = np.random.default_rng(11)
    x = (rng.standard_normal(200000) * np.exp(3.0 * rng.standard_normal(200000))).astype(np.float32)
    h, m, l, r2 = split3(x)
    # 24 significant bits = 8 + 8 + 8 when every remainder is rounded to nearest: the last level is exact for all but the rare
    # operand whose remainders keep a ninth bit, and then off by at most 2^-26 of x
    rest = (r2.astype(np.float64) - l.astype(np.float64))
    assert np.max(np.abs(rest) / np.abs(x.astype(np.float64))) <= 2.0 ** -25
    assert np.mean(rest == 0) > 0.95
    total = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.max(np.abs(total - x.astype(np.float64)) / np.abs(x.astype(np.float64))) <= 2.0 ** -25
    # sizes of the levels, and -- the point of rounding instead of masking -- remainders of both signs for operands of one sign
    ax = np.abs(x.astype(np.float64))
    assert np.max(np.abs(m) / ax) <= 2.0 ** -8 and np.max(np.abs(l) / ax) <= 2.0 ** -16
    pos = x > 0
    frac_neg = np.mean(m[pos] < 0)
    assert 0.4 < frac_neg < 0.6, frac_neg


def test_six_products_are_as_good_as_an_fp32_chain_and_unbiased():
    rng = np.random.default_rng(12)
    K = 4096
    a = (0.5 + np.abs(rng.standard_normal((64, K)))).astype(np.float32)   # positive operands: the coherent case
    b = (0.5 + np.abs(rng.standard_normal((64, K)))).astype(np.float32)
    ah, am, al, _ = split3(a)
    bh, bm, bl, _ = split3(b)
    f = lambda z: z.astype(np.float64)
    six = f(ah) * f(bh) + f(ah) * f(bm) + f(am) * f(bh) + f(am) * f(bm) + f(ah) * f(bl) + f(al) * f(bh)
    ref = f(a) * f(b)
    dropped = (six - ref) / ref                      # what the three dropped products leave, per term
    assert np.max(np.abs(dropped)) <= 3 * 2.0 ** -24
    # summed over the observations: zero-mean, so the relative error of a long sum shrinks like 1 / sqrt(K) ...
    rel_sum = np.abs(six.sum(axis=1) - ref.sum(axis=1)) / ref.sum(axis=1)
    assert np.max(rel_sum) <= 2.0 ** -24
    # ... where a split by truncation (masking the upper half of the word) keeps a bias of the order of 2^-17 of each product
    def trunc(x):
        return (np.asarray(x, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    th = trunc(a); tr = (a - th).astype(np.float32); tm = trunc(tr); tl = (tr - tm).astype(np.float32)
    uh = trunc(b); ur = (b - uh).astype(np.float32); um = trunc(ur); ul = (ur - um).astype(np.float32)
    six_t = f(th) * f(uh) + f(th) * f(um) + f(tm) * f(uh) + f(tm) * f(um) + f(th) * f(ul) + f(tl) * f(uh)
    rel_sum_t = np.abs(six_t.sum(axis=1) - ref.sum(axis=1)) / ref.sum(axis=1)
    assert np.min(rel_sum_t) > 20 * np.max(rel_sum), (np.min(rel_sum_t), np.max(rel_sum))
