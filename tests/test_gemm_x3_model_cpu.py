"""CPU model of the bf16 x 3 arithmetic of csrc/gemm_x3.hip (no GPU): the splits are exact, the six kept products reproduce an
fp32 product to its own rounding level, and the rounding mode of the splits decides whether the dropped terms are biased."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest-even bfloat16, returned as float32 (the value v_cvt_pk_bf16_f32 produces)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_trunc(x):
    return (np.asarray(x, np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x, rnd):
    x = np.asarray(x, np.float32)
    a1 = rnd(x)
    r1 = (x - a1).astype(np.float32)
    a2 = rnd(r1)
    r2 = (r1 - a2).astype(np.float32)
    a3 = rnd(r2)
    return a1, a2, a3, (r2 - a3).astype(np.float32)


def test_three_bf16_terms_are_exact_for_both_split_modes():
    """x3_split2 (gemm_x3.h): x = x1 + x2 + x3 exactly -- every remainder is representable (24 -> 16 -> 8 significant bits)."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.float32)
    for rnd in (bf16_rne, bf16_trunc):
        a1, a2, a3, rest = split3(x, rnd)
        assert np.all(rest == 0.0)
        assert np.array_equal((a1.astype(np.float64) + a2 + a3).astype(np.float32), x)
        for t in (a1, a2, a3):                                   # each term IS a bfloat16
            assert np.all((t.view(np.uint32) & 0xFFFF) == 0)


def test_six_products_reach_fp32_product_accuracy_and_rounding_removes_the_bias():
    """a*b ~ a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1): the three dropped terms are O(2^-24 |ab|) -- the size of the
    fp32 rounding of the product itself.  With TRUNCATED terms every dropped term has the sign of a*b (each product is a hair
    too small: a coherent bias, which the GPU kernel's matrix pipe amplified into a measurable error on an exactly-zero
    gradient, tests/diag/grad_accuracy.py); with ROUNDED terms the signs are independent and the mean error vanishes."""
    rng = np.random.default_rng(1)
    a = rng.standard_normal(400000).astype(np.float32)
    b = rng.standard_normal(400000).astype(np.float32)
    exact = a.astype(np.float64) * b.astype(np.float64)
    stats = {}
    for name, rnd in (("rne", bf16_rne), ("trunc", bf16_trunc)):
        a1, a2, a3, _ = split3(a, rnd)
        b1, b2, b3, _ = split3(b, rnd)
        six = (a1.astype(np.float64) * b1 + a1.astype(np.float64) * b2 + a2.astype(np.float64) * b1
               + a1.astype(np.float64) * b3 + a2.astype(np.float64) * b2 + a3.astype(np.float64) * b1)
        rel = (six - exact) / np.abs(exact)
        stats[name] = (np.abs(rel).max(), rel.mean() if False else ((six - exact) * np.sign(exact) / np.abs(exact)).mean())
    assert stats["rne"][0] < 2.0 ** -22 and stats["trunc"][0] < 2.0 ** -21          # product-rounding level either way
    assert stats["trunc"][1] < -1e-8                                                 # truncation: always toward zero
    assert abs(stats["rne"][1]) < 2e-10                                              # rounding: no preferred direction


def test_dot_product_with_separate_correction_accumulator_matches_fp64_better_than_an_fp32_chain():
    """The kernel keeps the leading products and the five correction products in two fp32 accumulators (the matrix pipe
    aligns all addends of an instruction to the largest and cuts the rest: small terms must not meet a large running sum).
    Model: K = 512 dot products, fp32 accumulation per 16-term block."""
    rng = np.random.default_rng(2)
    K, n = 512, 2000
    A = rng.standard_normal((n, K)).astype(np.float32)
    B = rng.standard_normal((n, K)).astype(np.float32)
    ref = (A.astype(np.float64) * B).sum(1)
    a1, a2, a3, _ = split3(A, bf16_rne)
    b1, b2, b3, _ = split3(B, bf16_rne)
    acc = np.zeros(n, np.float32)
    cor = np.zeros(n, np.float32)
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        lead = (a1[:, s].astype(np.float64) * b1[:, s]).sum(1)                      # exact products, one rounding per block
        corr = ((a1[:, s].astype(np.float64) * b2[:, s]) + (a2[:, s].astype(np.float64) * b1[:, s]) + (a1[:, s].astype(np.float64) * b3[:, s])
                + (a2[:, s].astype(np.float64) * b2[:, s]) + (a3[:, s].astype(np.float64) * b1[:, s])).sum(1)
        acc = (acc.astype(np.float64) + lead).astype(np.float32)
        cor = (cor.astype(np.float64) + corr).astype(np.float32)
    x3 = (acc.astype(np.float64) + cor).astype(np.float32)
    chain = np.zeros(n, np.float32)
    for k in range(K):                                                              # the k-ordered fp32 fma chain it replaces
        chain = (chain.astype(np.float64) + A[:, k].astype(np.float64) * B[:, k]).astype(np.float32)
    scale = (np.abs(A.astype(np.float64)) * np.abs(B)).sum(1)
    e_x3, e_chain = np.abs(x3 - ref) / scale, np.abs(chain - ref) / scale
    assert e_x3.max() < 4e-7 and np.sqrt((e_x3 ** 2).mean()) < np.sqrt((e_chain ** 2).mean())
