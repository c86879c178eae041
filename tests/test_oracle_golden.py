"""Pin the CPU oracle against every known-answer value the reference's committed logs hold.

216 `norm:` values (tests/golden/reference_norms.json): 5 hex logs x 14 sizes, 9 quad logs x 14
sizes, 20 bm01 sizes.  All printed with setprecision(10), so agreement is checked to 10
significant digits (rel 5e-10).  Every element of the reference's data is identical, so
norm(nelmt) = sqrt(nelmt) * ||out_elem||; we compute small sizes in full and large sizes from
one element's sum of squares (and say so).
"""
import math

import numpy as np
import pytest

FULL_LIMIT = 4096  # sizes computed with the full batch; larger ones by the per-element identity


def _sig10(x, ref_str):
    ref = float(ref_str)
    return abs(x - ref) <= 5.5e-10 * abs(ref)


@pytest.mark.parametrize("form", ["fused", "sweeps"])
def test_hex_norms(golden, oracle, form):
    for nq_s, entry in golden["hex"].items():
        nq = int(nq_s)
        nm = nq - 1
        b = oracle.fill_basis(nm, nq)
        one = oracle.bwdtrans_hex((nq,) * 3, 1, b, b, b, oracle.fill_sincos(1, nm ** 3), form=form)
        ss1 = oracle.sumsq(one)
        for row in entry["rows"]:
            n = row["n"]
            if n <= FULL_LIMIT and nq <= 6:
                out = oracle.bwdtrans_hex((nq,) * 3, n, b, b, b, oracle.fill_sincos(n, nm ** 3),
                                          form=form)
                norm = math.sqrt(oracle.sumsq(out))
            else:
                norm = math.sqrt(n * ss1)
            assert _sig10(norm, row["norm"]), (entry["file"], row, norm)


@pytest.mark.parametrize("form", ["fused", "sweeps"])
def test_quad_norms(golden, oracle, form):
    for nq_s, entry in golden["quad"].items():
        nq = int(nq_s)
        nm = nq - 1
        b = oracle.fill_basis(nm, nq)
        one = oracle.bwdtrans_quad((nq, nq), 1, b, b, oracle.fill_sincos(1, nm * nm), form=form)
        ss1 = oracle.sumsq(one)
        for row in entry["rows"]:
            n = row["n"]
            if n <= FULL_LIMIT:
                out = oracle.bwdtrans_quad((nq, nq), n, b, b, oracle.fill_sincos(n, nm * nm),
                                           form=form)
                norm = math.sqrt(oracle.sumsq(out))
            else:
                norm = math.sqrt(n * ss1)
            assert _sig10(norm, row["norm"]), (entry["file"], row, norm)


def test_l2norm_norms(golden, oracle):
    rows = golden["l2norm"]["rows"]
    assert len(rows) == 20
    for row in rows:
        n = row["n"]
        if n > (1 << 24):  # keep the CPU suite short; larger sizes are covered on the GPU box
            continue
        x = oracle.fill_l2norm(n)
        assert _sig10(math.sqrt(oracle.sumsq(x)), row["norm"]), row


def test_vecadd_norms(golden, oracle):
    """benchmark02: after the 40 timed in-place additions, sqrt(sum data1^2) as published."""
    rows = golden["vecadd"]["rows"]
    assert len(rows) == 20
    for row in rows:
        n = row["n"]
        if n > (1 << 22):
            continue
        x, y = oracle.fill_vecadd(n)
        oracle.vector_add(x, y, times=40)
        assert _sig10(math.sqrt(oracle.sumsq(x)), row["norm"]), row


def test_matvec_norms(golden, oracle):
    rows = golden["matvec"]["rows"]
    assert len(rows) == 8
    for row in rows:
        n = row["n"]
        if n > 4096:
            continue
        a, x = oracle.fill_matvec(n, n)
        y = oracle.matvec(n, n, a, x)
        assert _sig10(math.sqrt(oracle.sumsq(y)), row["norm"]), row
        if n <= 1024:
            assert oracle.rel_err(y, a.reshape(n, n) @ x) < 1e-12


def test_forms_agree_and_match_numpy(oracle):
    """fused nest == 3-sweep form == independent einsum, on per-element-distinct data."""
    for nq in [(2, 2, 2), (3, 3, 3), (4, 4, 4), (8, 8, 8), (10, 10, 10), (3, 5, 4), (8, 2, 6)]:
        nm = tuple(q - 1 for q in nq)
        nelmt = 7
        b = [oracle.fill_random(nm[d] * nq[d], 11 + d) for d in range(3)]
        x = oracle.fill_random(nelmt * nm[0] * nm[1] * nm[2], 5)
        a = oracle.bwdtrans_hex(nq, nelmt, *b, x, form="fused")
        s = oracle.bwdtrans_hex(nq, nelmt, *b, x, form="sweeps")
        e = oracle.bwdtrans_hex_numpy(nq, nelmt, *b, x)
        assert np.array_equal(a, s)  # same products, same summation order per dot product
        v = oracle.bwdtrans_hex(nq, nelmt, *b, x, form="vector")
        assert np.array_equal(a, v)  # CPU-friendly loop order, still the same sums
        assert oracle.rel_err(a, e) < 1e-13
    for nq in [(2, 2), (5, 5), (8, 8), (32, 32), (4, 9), (16, 3)]:
        nm = tuple(q - 1 for q in nq)
        nelmt = 9
        b = [oracle.fill_random(nm[d] * nq[d], 21 + d) for d in range(2)]
        x = oracle.fill_random(nelmt * nm[0] * nm[1], 6)
        a = oracle.bwdtrans_quad(nq, nelmt, *b, x, form="fused")
        s = oracle.bwdtrans_quad(nq, nelmt, *b, x, form="sweeps")
        e = oracle.bwdtrans_quad_numpy(nq, nelmt, *b, x)
        assert np.array_equal(a, s)
        assert oracle.rel_err(a, e) < 1e-13


def test_fast_build_matches_parity_build(oracle):
    """The -O3/FMA build timed as cpu_baseline computes the same thing (to rounding)."""
    nq, nelmt = (8, 8, 8), 33
    b = oracle.fill_basis(7, 8)
    x = oracle.fill_random(nelmt * 343, 3)
    a = oracle.bwdtrans_hex(nq, nelmt, b, b, b, x)
    f = oracle.bwdtrans_hex(nq, nelmt, b, b, b, x, fast=True)
    assert oracle.rel_err(f, a) < 1e-14
    v = oracle.bwdtrans_hex(nq, nelmt, b, b, b, x, form="vector", fast=True)
    assert oracle.rel_err(v, a) < 1e-14


def test_blocked_form_is_the_same_sums(oracle):
    """The register-blocked form timed as cpu_baseline: bit-identical to the 3-sweep form in the parity build
    (same products, same ascending summation), to rounding in the FMA build, every order it exists for."""
    for nq in range(2, 11):
        nm, nelmt = nq - 1, 37
        b = [oracle.fill_random(nm * nq, 31 + d) for d in range(3)]
        x = oracle.fill_random(nelmt * nm ** 3, 9)
        s = oracle.bwdtrans_hex((nq,) * 3, nelmt, *b, x, form="sweeps")
        k = oracle.bwdtrans_hex((nq,) * 3, nelmt, *b, x, form="blocked")
        assert oracle.has_blocked(nq) and np.array_equal(s, k), nq
        f = oracle.bwdtrans_hex((nq,) * 3, nelmt, *b, x, form="blocked", fast=True)
        assert oracle.rel_err(f, s) < 1e-14
        if oracle.host_has_avx512():    # the 8-wide build of the same source (bench.py times the faster of the two)
            w = oracle.bwdtrans_hex((nq,) * 3, nelmt, *b, x, form="blocked", fast="avx512")
            assert oracle.rel_err(w, s) < 1e-14
    assert not oracle.has_blocked(11)
    with pytest.raises(ValueError):
        oracle.bwdtrans_hex((3, 4, 3), 2, *[oracle.fill_basis(q - 1, q) for q in (3, 4, 3)],
                            oracle.fill_random(2 * 2 * 3 * 2, 1), form="blocked")


def test_random_generator_pinned(oracle):
    for seed, idx in [(0, 0), (0x5F3759DF, 12345), (7, (1 << 40) + 3)]:
        v = oracle.fill_random(1, seed, idx)[0]
        assert v == oracle.random_value_py(seed, idx)
        assert -1.0 <= v < 1.0
    a = oracle.fill_random(1000, 42)
    assert abs(a.mean()) < 0.1 and 0.5 < a.std() < 0.65
    assert np.array_equal(oracle.fill_random(10, 42, 5), a[5:15])


def test_edge_cases(oracle):
    b = oracle.fill_basis(1, 2)
    out = oracle.bwdtrans_hex((2, 2, 2), 0, b, b, b, np.empty(0))
    assert out.size == 0
    with pytest.raises(ValueError):
        oracle.bwdtrans_hex((1, 2, 2), 1, b, b, b, np.zeros(1))
