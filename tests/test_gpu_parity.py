"""GPU parity tests: the HIP path (through the C ABI of libsumfact.so) against the CPU oracle.

Bar: fp64, max|GPU - oracle| / max|oracle| <= 1e-12 (BASELINE.json north_star), on per-element-
distinct seeded data (catches element-offset bugs that the reference's identical data hides) and on
the reference's sin/cos data (golden `norm:` values of the committed logs, 10 digits).
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope="module")
def sf():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


def _np(t):
    return t.detach().cpu().numpy()


def _hex_case(sf, oracle, nq, nelmt, variant, seed=1):
    nm = [q - 1 for q in nq]
    bs = [sf.fill_random(nm[d] * nq[d], 100 + d + seed) for d in range(3)]
    x = sf.fill_random(nelmt * nm[0] * nm[1] * nm[2], seed)
    out = sf.bwdtrans_hex(nq, *bs, x, variant=variant)
    ref = oracle.bwdtrans_hex(tuple(nq), nelmt, *[_np(b) for b in bs], _np(x))
    return oracle.rel_err(_np(out), ref)


def _quad_case(sf, oracle, nq, nelmt, variant, seed=1):
    nm = [q - 1 for q in nq]
    bs = [sf.fill_random(nm[d] * nq[d], 200 + d + seed) for d in range(2)]
    x = sf.fill_random(nelmt * nm[0] * nm[1], seed)
    out = sf.bwdtrans_quad(nq, *bs, x, variant=variant)
    ref = oracle.bwdtrans_quad(tuple(nq), nelmt, *[_np(b) for b in bs], _np(x))
    return oracle.rel_err(_np(out), ref)


# ragged element counts: 1, below / at / above every chunk size, primes
RAGGED = [1, 2, 3, 5, 13, 14, 15, 63, 64, 65, 127, 257, 1000, 4099]


@pytest.mark.parametrize("nq", range(2, 12))
def test_hex_wave_parity_all_orders(sf, oracle, nq):
    for nelmt in RAGGED:
        err = _hex_case(sf, oracle, (nq,) * 3, nelmt, "wave", seed=nelmt)
        assert err <= TOL, (nq, nelmt, err)


@pytest.mark.parametrize("nq", list(range(2, 25)) + [32])
def test_quad_wave_parity_all_orders(sf, oracle, nq):
    for nelmt in RAGGED:
        err = _quad_case(sf, oracle, (nq, nq), nelmt, "wave", seed=nelmt)
        assert err <= TOL, (nq, nelmt, err)


@pytest.mark.parametrize("nq", range(11, 33))
def test_quad_mfma_parity_all_orders(sf, oracle, nq):
    """Matrix-core kernel (v_mfma_f64_16x16x4), every order it is built for, ragged counts."""
    for nelmt in (1, 2, 3, 7, 64, 129, 1000):
        err = _quad_case(sf, oracle, (nq, nq), nelmt, "mfma", seed=nelmt + nq)
        assert err <= TOL, (nq, nelmt, err)
        err = _quad_case(sf, oracle, (nq, nq), nelmt, "auto", seed=nelmt)
        assert err <= TOL, (nq, nelmt, err)


@pytest.mark.parametrize("nq", range(8, 33))
def test_quad_mfma4_parity_all_orders(sf, oracle, nq):
    """v_mfma_f64_4x4x4_4b kernel (tile granularity 4; AUTO runs it at nq 21..31): every order it is built for, ragged
    counts around its 1 / 2 / 4-element chunks and its one-chunk, several-chunk and persistent wave mappings."""
    for nelmt in (1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 129, 1000, 4099):
        err = _quad_case(sf, oracle, (nq, nq), nelmt, "mfma4", seed=nelmt + nq)
        assert err <= TOL, (nq, nelmt, err)
    # more chunks than a persistent grid has waves (256 CUs x 4..8 waves), ragged tail, sin/cos basis; from nq 25 the
    # grid draws batches of 4..8 chunks from a device-wide counter, two tickets up front per wave: 131 101 elements are
    # 4 100+ batches against ~2 048 initial tickets, so every wave redeems tickets issued inside its loop
    nelmt = 70001 if nq <= 24 else 131101
    b = sf.fill_basis(nq - 1, nq)
    x = sf.fill_random(nelmt * (nq - 1) ** 2, 77 + nq)
    got = sf.bwdtrans_quad((nq, nq), b, b, x, variant="mfma4")
    ref = oracle.bwdtrans_quad((nq, nq), nelmt, _np(b), _np(b), _np(x))
    assert oracle.rel_err(_np(got), ref) <= TOL


@pytest.mark.parametrize("nq", range(4, 17))
def test_hex_mfma_parity_all_orders(sf, oracle, nq):
    """3D matrix-core kernel: three chained v_mfma_f64_16x16x4 GEMMs, every order it is built for
    (AUTO routes nq 11..16 to it)."""
    for nelmt in (1, 2, 3, 7, 64, 129, 600):
        err = _hex_case(sf, oracle, (nq,) * 3, nelmt, "mfma", seed=nelmt + nq)
        assert err <= TOL, (nq, nelmt, err)
        if nq > 10:
            err = _hex_case(sf, oracle, (nq,) * 3, nelmt, "auto", seed=nelmt)
            assert err <= TOL, (nq, nelmt, err)


@pytest.mark.parametrize("nq", range(12, 17))
def test_hex_mfma4_parity_all_orders(sf, oracle, nq):
    """3D kernel on v_mfma_f64_4x4x4_4b (csrc/bwdtrans_hmfma4.h; AUTO runs it at nq 12 and 16): every order it is built
    for, both output paths, ragged element counts, plus more elements than the grid has waves with distinct bases per
    direction."""
    import os
    try:
        for cfg in ("1", "3"):  # output through an LDS image / accumulators stored directly
            os.environ["SF_HEX_MFMA4_CFG"] = cfg
            for nelmt in (1, 2, 3, 7, 64, 129, 600):
                err = _hex_case(sf, oracle, (nq,) * 3, nelmt, "mfma4", seed=nelmt + nq)
                assert err <= TOL, (cfg, nq, nelmt, err)
            nelmt = 5003
            bs = [sf.fill_random((nq - 1) * nq, 11 + d) for d in range(3)]
            x = sf.fill_random(nelmt * (nq - 1) ** 3, 5 + nq)
            got = sf.bwdtrans_hex((nq,) * 3, *bs, x, variant="mfma4")
            ref = oracle.bwdtrans_hex((nq,) * 3, nelmt, *[_np(b) for b in bs], _np(x))
            assert oracle.rel_err(_np(got), ref) <= TOL, cfg
    finally:
        os.environ.pop("SF_HEX_MFMA4_CFG", None)


RT_SHAPES = [(8, 8, 4), (4, 8, 6), (10, 6, 8), (2, 2, 2), (2, 16, 3), (16, 2, 2), (3, 2, 5), (7, 7, 7), (5, 9, 13),
             (16, 16, 16), (15, 16, 14), (12, 3, 9)]


@pytest.mark.parametrize("nq", RT_SHAPES)
def test_hex_runtime_extent_wave_kernel(sf, oracle, nq):
    """The wave-per-chunk kernel with RUN-TIME extents (csrc/bwdtrans_rt.h; what AUTO runs for anisotropic 3D extents
    up to 16, benchmark05/benchmark05.cc:291-297 takes them at run time): ragged element counts around its chunk sizes,
    explicit variant and AUTO, against the oracle."""
    for nelmt in (1, 2, 3, 5, 15, 16, 17, 63, 64, 65, 257, 1000):
        err = _hex_case(sf, oracle, nq, nelmt, "wave-rt", seed=nelmt)
        assert err <= TOL, (nq, nelmt, err)
    assert _hex_case(sf, oracle, nq, 4099, "auto", seed=3) <= TOL


TRIPLES = [(8, 8, 4), (8, 4, 8), (4, 8, 8), (4, 8, 6), (4, 6, 8), (8, 4, 6), (8, 6, 4), (6, 4, 8), (6, 8, 4), (10, 6, 8),
           (10, 8, 6), (6, 10, 8), (6, 8, 10), (8, 10, 6), (8, 6, 10), (8, 8, 6), (8, 6, 8), (6, 8, 8), (6, 6, 8), (6, 8, 6),
           (8, 6, 6), (8, 8, 10), (8, 10, 8), (10, 8, 8), (10, 10, 8), (10, 8, 10), (8, 10, 10), (6, 6, 4), (6, 4, 6),
           (4, 6, 6), (4, 4, 6), (4, 6, 4), (6, 4, 4)]


@pytest.mark.parametrize("nq", TRIPLES)
def test_hex_compile_time_anisotropic_triples(sf, oracle, nq):
    """csrc/bwdtrans_wave3.h: the flagship wave-per-chunk kernel instantiated for anisotropic extents (the table in
    csrc/bwdtrans_rt.hip; SF_VARIANT_WAVE with nq0 != nq1 != nq2, and what AUTO picks for these shapes): ragged element
    counts around the chunk sizes 1..8, a batch with a partial XCD window, against the oracle."""
    for nelmt in (1, 2, 3, 4, 5, 7, 8, 9, 15, 17, 63, 65, 257, 1000):
        err = _hex_case(sf, oracle, nq, nelmt, "wave", seed=nelmt)
        assert err <= TOL, (nq, nelmt, err)
    assert _hex_case(sf, oracle, nq, 12345, "auto", seed=5) <= TOL


def test_anisotropic_shape_outside_the_table_is_not_built_as_wave(sf):
    b = [sf.fill_basis(q - 1, q) for q in (5, 9, 13)]
    with pytest.raises(sf.capi.SumfactError) as ei:
        sf.bwdtrans_hex((5, 9, 13), *b, sf.fill_random(4 * 8 * 12 * 3, 1), variant="wave")
    assert ei.value.rc == sf.capi.SF_ENOTBUILT


def test_hex_runtime_extent_kernel_on_8_byte_aligned_views(sf, oracle, torch_mod):
    """`in` / `out` that are only 8-byte aligned (odd offsets into a larger buffer): chunks then start on either half
    of a 16-byte word; guard words either side of the output stay untouched."""
    for nq, nelmt in (((8, 8, 4), 333), ((4, 8, 6), 1001), ((3, 3, 3), 77), ((8, 8, 8), 129)):
        nm = [q - 1 for q in nq]
        nmt, nqt = nm[0] * nm[1] * nm[2], nq[0] * nq[1] * nq[2]
        bs = [sf.fill_random(nm[d] * nq[d], 300 + d) for d in range(3)]
        for off_in, off_out in ((1, 0), (0, 1), (1, 1), (3, 5)):
            xbuf = sf.fill_random(nelmt * nmt + 8, 17 + off_in)
            x = xbuf[off_in:off_in + nelmt * nmt]
            obuf = torch_mod.full((nelmt * nqt + 16,), 7.25, dtype=torch_mod.float64, device="cuda")
            o = obuf[off_out:off_out + nelmt * nqt]
            sf.bwdtrans_hex(nq, *bs, x, out=o, variant="auto")
            torch_mod.cuda.synchronize()
            ref = oracle.bwdtrans_hex(tuple(nq), nelmt, *[_np(b) for b in bs], _np(x))
            assert oracle.rel_err(_np(o), ref) <= TOL, (nq, off_in, off_out)
            assert bool((obuf[:off_out] == 7.25).all()) and bool((obuf[off_out + nelmt * nqt:] == 7.25).all())


def test_mfma_not_built_cases(sf):
    capi = sf.capi
    b = sf.fill_basis(7, 8)
    bb = sf.fill_basis(16, 17)
    x = sf.fill_random(16 ** 3 * 4, 1)
    with pytest.raises(capi.SumfactError) as ei:
        sf.bwdtrans_hex((17, 17, 17), bb, bb, bb, x, variant="mfma")
    assert ei.value.rc == capi.SF_ENOTBUILT
    x2 = sf.fill_random(49 * 4, 1)
    with pytest.raises(capi.SumfactError) as ei:
        sf.bwdtrans_quad((8, 8), b, b, x2, variant="mfma")
    assert ei.value.rc == capi.SF_ENOTBUILT
    b7 = sf.fill_basis(6, 7)
    with pytest.raises(capi.SumfactError) as ei:
        sf.bwdtrans_quad((7, 7), b7, b7, sf.fill_random(36 * 4, 1), variant="mfma4")
    assert ei.value.rc == capi.SF_ENOTBUILT
    with pytest.raises(capi.SumfactError) as ei:                      # 2D-only kernel
        sf.bwdtrans_hex((8, 8, 8), b, b, b, sf.fill_random(343 * 4, 1), variant="mfma4")
    assert ei.value.rc == capi.SF_ENOTBUILT


@pytest.mark.parametrize("variant", ["auto", "thread", "block-lds", "block-glb", "generic"])
def test_hex_variants(sf, oracle, variant):
    for nq in [(2, 2, 2), (4, 4, 4), (8, 8, 8), (3, 5, 4), (8, 2, 6), (10, 10, 10), (12, 12, 12)]:
        for nelmt in (1, 7, 130):
            err = _hex_case(sf, oracle, nq, nelmt, variant)
            assert err <= TOL, (variant, nq, nelmt, err)


@pytest.mark.parametrize("variant", ["auto", "thread", "block-lds", "block-glb", "generic"])
def test_quad_variants(sf, oracle, variant):
    for nq in [(2, 2), (8, 8), (4, 9), (16, 3), (20, 20), (32, 32), (40, 40)]:
        for nelmt in (1, 9, 300):
            err = _quad_case(sf, oracle, nq, nelmt, variant)
            assert err <= TOL, (variant, nq, nelmt, err)


def test_golden_norms_hex(sf, golden):
    """Reference sin/cos data -> sqrt(sum out^2) equals the `norm:` column of the committed logs."""
    for nq_s, entry in golden["hex"].items():
        nq = int(nq_s)
        nm = nq - 1
        b = sf.fill_basis(nm, nq)
        for row in entry["rows"]:
            n = row["n"]
            if n not in (128, 4096, 1048576):
                continue
            x = sf.fill_sincos(n, nm ** 3)
            out = sf.bwdtrans_hex((nq,) * 3, b, b, b, x)
            norm = math.sqrt(sf.sumsq(out))
            ref = float(row["norm"])
            assert abs(norm - ref) <= 5.5e-10 * ref, (entry["file"], row, norm)
            del x, out


def test_golden_norms_quad(sf, golden):
    for nq_s, entry in golden["quad"].items():
        nq = int(nq_s)
        nm = nq - 1
        b = sf.fill_basis(nm, nq)
        for row in entry["rows"]:
            n = row["n"]
            if n not in (128, 4096, 1048576):
                continue
            x = sf.fill_sincos(n, nm * nm)
            out = sf.bwdtrans_quad((nq, nq), b, b, x)
            norm = math.sqrt(sf.sumsq(out))
            ref = float(row["norm"])
            assert abs(norm - ref) <= 5.5e-10 * ref, (entry["file"], row, norm)


def test_golden_norms_l2norm(sf, golden):
    """bm01 data + sum of squares on the device, all 20 sizes (up to 536 870 912 doubles)."""
    for row in golden["l2norm"]["rows"]:
        x = sf.fill_l2norm(row["n"])
        norm = math.sqrt(sf.sumsq(x))
        ref = float(row["norm"])
        assert abs(norm - ref) <= 5.5e-10 * ref, (row, norm)
        del x


def test_fills_match_oracle(sf, oracle):
    a = _np(sf.fill_random(100003, 0x5F3759DF, 12345))
    assert np.array_equal(a, oracle.fill_random(100003, 0x5F3759DF, 12345))  # bit-exact
    s = _np(sf.fill_sincos(5, 343))
    assert np.max(np.abs(s - oracle.fill_sincos(5, 343))) <= 4e-16          # device libm: <= 2 ulp
    b = _np(sf.fill_basis(7, 8))
    assert np.max(np.abs(b - oracle.fill_basis(7, 8))) <= 4e-16
    x = _np(sf.fill_l2norm(300000))
    assert np.array_equal(x, oracle.fill_l2norm(300000))


def test_sumsq(sf, oracle, torch_mod):
    for n in (1, 2, 3, 255, 4096, 1000003):
        x = sf.fill_random(n, 9)
        got = sf.sumsq(x)
        ref = oracle.sumsq(_np(x))
        assert abs(got - ref) <= 1e-13 * ref, (n, got, ref)
        assert got == sf.sumsq(x)  # deterministic
    # unaligned view (8-byte aligned only)
    x = sf.fill_random(1001, 3)
    assert abs(sf.sumsq(x[1:]) - oracle.sumsq(_np(x)[1:])) <= 1e-13 * 400
    # from 2^24 values the reduction is a persistent strided grid (aux_kernels.hip, sumsq_stride_kernel): even / odd
    # counts either side of the switch, an 8-byte-aligned view of a large array (scalar kernel), determinism
    big = sf.fill_random((1 << 24) + 1025, 21)
    bh = _np(big)
    for lo, n in ((0, (1 << 24) - 2), (0, 1 << 24), (0, (1 << 24) + 1), (0, big.numel()), (1, (1 << 24) + 7)):
        got = sf.sumsq(big[lo:lo + n])
        ref = oracle.sumsq(bh[lo:lo + n])
        assert abs(got - ref) <= 1e-13 * ref, (lo, n, got, ref)
        assert got == sf.sumsq(big[lo:lo + n])


def test_vecadd_and_matvec(sf, oracle, golden, torch_mod):
    """Section 8(f) kernels: benchmark02 x += y (bit-exact vs the host statement, published norm after
    40 additions, all 20 sizes) and benchmark03 y = A x (<= 1e-12, published norms, all 8 sizes)."""
    for row in golden["vecadd"]["rows"]:
        n = row["n"]
        x, y = sf.fill_vecadd(n)
        if n <= (1 << 20):
            xr, yr = oracle.fill_vecadd(n)
            assert np.array_equal(_np(x), xr) and np.array_equal(_np(y), yr)
        for _ in range(40):
            sf.vector_add(x, y)
        if n <= (1 << 20):
            assert np.array_equal(_np(x), oracle.vector_add(xr, yr, times=40))
        ref = float(row["norm"])
        assert abs(math.sqrt(sf.sumsq(x)) - ref) <= 5.5e-10 * ref, row
        del x, y
    # odd length + unaligned views
    x, y = sf.fill_vecadd(1003)
    xr, yr = oracle.fill_vecadd(1003)
    sf.vector_add(x, y)
    assert np.array_equal(_np(x), oracle.vector_add(xr.copy(), yr, 1))
    sf.vector_add(x[1:], y[1:])
    for row in golden["matvec"]["rows"]:
        n = row["n"]
        a, x = sf.fill_matvec(n, n)
        yv = sf.matvec(n, n, a, x)
        ref = float(row["norm"])
        assert abs(math.sqrt(sf.sumsq(yv)) - ref) <= 5.5e-10 * ref, row
        if n <= 2048:
            yo = oracle.matvec(n, n, _np(a), _np(x))
            assert oracle.rel_err(_np(yv), yo) <= TOL
        del a, x, yv
    a, x = sf.fill_matvec(37, 101)       # odd extents take the 8-byte-lane path
    yo = oracle.matvec(37, 101, _np(a), _np(x))
    assert oracle.rel_err(_np(sf.matvec(37, 101, a, x)), yo) <= TOL


def test_interleaved_layout_variant(sf, oracle):
    """Wave-64 interleaved layout (corrected `_Coa`): layout round trip is exact, results match the
    oracle, including a ragged last group and anisotropic extents."""
    for nq, nelmt in (((8, 8, 8), 200), ((4, 4, 4), 64), ((3, 5, 4), 131), ((2, 2, 2), 1)):
        nm = [q - 1 for q in nq]
        nmt, nqt = nm[0] * nm[1] * nm[2], nq[0] * nq[1] * nq[2]
        bs = [sf.fill_random(nm[d] * nq[d], 70 + d) for d in range(3)]
        x = sf.fill_random(nelmt * nmt, nelmt)
        x_il = sf.interleave64(x, nelmt, nmt)
        assert np.array_equal(_np(sf.interleave64(x_il, nelmt, nmt, inverse=True)), _np(x))
        # spot-check the layout definition: element e, mode f sits at (e//64)*64*nmt + f*64 + e%64
        e, f = nelmt - 1, nmt - 1
        assert _np(x_il)[(e // 64) * 64 * nmt + f * 64 + e % 64] == _np(x)[e * nmt + f]
        out_il = sf.bwdtrans_hex_interleaved(nq, *bs, x_il, nelmt)
        out = sf.interleave64(out_il, nelmt, nqt, inverse=True)
        ref = oracle.bwdtrans_hex(nq, nelmt, *[_np(b) for b in bs], _np(x))
        assert oracle.rel_err(_np(out), ref) <= TOL, (nq, nelmt)


def test_randomised_shapes_and_alignment(sf, oracle, torch_mod):
    """Seeded fuzz over (dimension, order incl. anisotropic, element count, 8-byte-misaligned views):
    SF_VARIANT_AUTO must always agree with the oracle, whichever kernel it picks."""
    rng = np.random.default_rng(20251004)
    for case in range(60):
        dim = 3 if rng.random() < 0.5 else 2
        if rng.random() < 0.3:
            nq = tuple(int(v) for v in rng.integers(2, 8 if dim == 3 else 20, size=dim))
        else:
            nq = (int(rng.integers(2, 11 if dim == 3 else 33)),) * dim
        nelmt = int(rng.integers(1, 3000))
        mis_in, mis_out = bool(rng.random() < 0.2), bool(rng.random() < 0.2)
        nm = [q - 1 for q in nq]
        nmt, nqt = int(np.prod(nm)), int(np.prod(nq))
        bs = [sf.fill_random(nm[d] * nq[d], 500 + 7 * case + d) for d in range(dim)]
        xbuf = sf.fill_random(nelmt * nmt + 1, 900 + case)
        x = xbuf[1:] if mis_in else xbuf[:-1]
        obuf = torch_mod.zeros(nelmt * nqt + 1, dtype=torch_mod.float64, device="cuda")
        out = obuf[1:] if mis_out else obuf[:-1]
        if dim == 3:
            sf.bwdtrans_hex(nq, *bs, x, out=out)
            ref = oracle.bwdtrans_hex(nq, nelmt, *[_np(b) for b in bs], _np(x).copy())
        else:
            sf.bwdtrans_quad(nq, *bs, x, out=out)
            ref = oracle.bwdtrans_quad(nq, nelmt, *[_np(b) for b in bs], _np(x).copy())
        assert oracle.rel_err(_np(out), ref) <= TOL, (case, dim, nq, nelmt, mis_in, mis_out)
        # nothing written outside the view
        assert float(obuf[0 if mis_out else -1]) == 0.0


@pytest.mark.parametrize("dim", [2, 3])
def test_line_alignment_offsets(sf, oracle, torch_mod, dim):
    """The wave kernels shift their 16-byte lanes so wave-wide loads / stores cover whole 128-byte lines
    (wave_table.h, MF bits 2 and 3): every 16-byte-multiple offset of `in` and `out` inside a 128-byte line
    must give the same result, with nothing written before or after the output view."""
    GUARD = 32
    for nq in (range(2, 12) if dim == 3 else list(range(2, 25))):
        nmt, nqt = (nq - 1) ** dim, nq ** dim
        bs = [sf.fill_random((nq - 1) * nq, 40 + d) for d in range(dim)]
        for nelmt, off_in, off_out in ((257, 2, 0), (300, 6, 4), (1031, 10, 14), (64, 0, 8), (5, 12, 2)):
            xbuf = sf.fill_random(nelmt * nmt + 16, 7 * nq + nelmt)
            x = xbuf[off_in:off_in + nelmt * nmt]
            obuf = torch_mod.full((nelmt * nqt + 2 * GUARD,), -7.0, dtype=torch_mod.float64, device="cuda")
            out = obuf[GUARD + off_out - 16:GUARD + off_out - 16 + nelmt * nqt]
            if dim == 3:
                sf.bwdtrans_hex((nq,) * 3, *bs, x, out=out, variant="wave")
                ref = oracle.bwdtrans_hex((nq,) * 3, nelmt, *[_np(b) for b in bs], _np(x).copy())
            else:
                sf.bwdtrans_quad((nq,) * 2, *bs, x, out=out, variant="wave")
                ref = oracle.bwdtrans_quad((nq,) * 2, nelmt, *[_np(b) for b in bs], _np(x).copy())
            assert oracle.rel_err(_np(out), ref) <= TOL, (dim, nq, nelmt, off_in, off_out)
            head = obuf[:GUARD + off_out - 16]
            tail = obuf[GUARD + off_out - 16 + nelmt * nqt:]
            assert bool((head == -7.0).all()) and bool((tail == -7.0).all()), (dim, nq, nelmt)


@pytest.mark.parametrize("nq", range(12, 17))
def test_hex_matrix_core_orders_line_offsets_and_guard_bands(sf, oracle, torch_mod, nq):
    """3D nq 12..16 through AUTO (hex_mfma4_kernel at 12 / 14 / 16 -- nq 16 stores its accumulators directly --,
    hex_mfma_kernel at 13 / 15) and through both output paths of the 4x4x4 kernel: every 16-byte-multiple offset of `in`
    and `out` inside a 128-byte line gives the oracle's result, and nothing is written before or after the output view."""
    import os
    GUARD = 32
    nmt, nqt = (nq - 1) ** 3, nq ** 3
    bs = [sf.fill_random((nq - 1) * nq, 60 + d) for d in range(3)]
    try:
        for variant, cfg in (("auto", None), ("mfma4", "1"), ("mfma4", "3")):
            if cfg is None:
                os.environ.pop("SF_HEX_MFMA4_CFG", None)
            else:
                os.environ["SF_HEX_MFMA4_CFG"] = cfg
            for nelmt, off_in, off_out in ((131, 2, 0), (300, 6, 4), (77, 10, 14), (64, 0, 8), (5, 12, 2)):
                xbuf = sf.fill_random(nelmt * nmt + 16, 7 * nq + nelmt)
                x = xbuf[off_in:off_in + nelmt * nmt]
                obuf = torch_mod.full((nelmt * nqt + 2 * GUARD,), -7.0, dtype=torch_mod.float64, device="cuda")
                out = obuf[GUARD + off_out - 16:GUARD + off_out - 16 + nelmt * nqt]
                sf.bwdtrans_hex((nq,) * 3, *bs, x, out=out, variant=variant)
                ref = oracle.bwdtrans_hex((nq,) * 3, nelmt, *[_np(b) for b in bs], _np(x).copy())
                assert oracle.rel_err(_np(out), ref) <= TOL, (variant, cfg, nq, nelmt, off_in, off_out)
                head = obuf[:GUARD + off_out - 16]
                tail = obuf[GUARD + off_out - 16 + nelmt * nqt:]
                assert bool((head == -7.0).all()) and bool((tail == -7.0).all()), (variant, cfg, nq, nelmt)
    finally:
        os.environ.pop("SF_HEX_MFMA4_CFG", None)


@pytest.mark.parametrize("dim,nq", [(3, 8), (3, 10), (3, 5), (3, 13), (2, 16), (2, 20), (2, 28), (2, 9)])
def test_xcd_window_tails(sf, oracle, dim, nq):
    """Workgroups are renumbered in windows of 8 * 64 (XCD runs); an element count that leaves many full windows
    plus a partial tail window must still map every chunk exactly once."""
    nelmt = 70001 if dim == 3 and nq <= 10 else (20011 if dim == 3 else 300007)
    err = (_hex_case(sf, oracle, (nq,) * 3, nelmt, "auto", seed=nq) if dim == 3
           else _quad_case(sf, oracle, (nq, nq), nelmt, "auto", seed=nq))
    assert err <= TOL, (dim, nq, nelmt, err)


def test_non_default_streams(sf, oracle, torch_mod):
    """`stream` is honoured: two launches on two streams, each synchronised on its own stream."""
    nq, nelmt = 8, 5000
    b = sf.fill_basis(7, 8)
    xs = [sf.fill_random(nelmt * 343, 11 + k) for k in range(2)]
    streams = [torch_mod.cuda.Stream() for _ in range(2)]
    torch_mod.cuda.synchronize()
    outs = []
    for st, x in zip(streams, xs):
        with torch_mod.cuda.stream(st):
            outs.append(sf.bwdtrans_hex((nq,) * 3, b, b, b, x, stream=st))
    for st in streams:
        st.synchronize()
    for x, o in zip(xs, outs):
        ref = oracle.bwdtrans_hex((nq,) * 3, nelmt, _np(b), _np(b), _np(b), _np(x))
        assert oracle.rel_err(_np(o), ref) <= TOL


def test_persistent_quad_kernels_are_capture_safe_from_the_first_call(sf, oracle, torch_mod):
    """AUTO 2D nq 25..32 feed a persistent grid from a batch counter.  The counters live in a per-device buffer that is
    never evicted (csrc/aux_kernels.hip counter_acquire): (a) a capture that PRECEDES the buffer's allocation runs the
    fixed-share kernel instead of allocating inside the capture; (b) a launch captured later bakes a counter of its own,
    which neither eager launches on other streams nor scratch churn on 200 more streams invalidate.  All replays
    bit-identical to the eager result and within 1e-12 of the oracle."""
    nq, nelmt = 28, 30011
    lib = sf.capi.lib()
    b = sf.fill_basis(nq - 1, nq)
    x = sf.fill_random(nelmt * (nq - 1) ** 2, 5)
    ref = oracle.bwdtrans_quad((nq, nq), nelmt, _np(b), _np(b), _np(x))
    torch_mod.cuda.synchronize()
    assert lib.sf_shutdown() == 0                       # drops the counter buffer: the next use is a FIRST use

    def captured():
        o = torch_mod.zeros(nelmt * nq * nq, dtype=torch_mod.float64, device="cuda")
        side = torch_mod.cuda.Stream()
        side.wait_stream(torch_mod.cuda.current_stream())
        g = torch_mod.cuda.CUDAGraph()
        with torch_mod.cuda.stream(side):
            with torch_mod.cuda.graph(g, stream=side):
                sf.bwdtrans_quad((nq, nq), b, b, x, out=o, stream=side)
        torch_mod.cuda.current_stream().wait_stream(side)
        return g, o

    g1, o1 = captured()                                 # (a) first use of a counter inside a capture
    g1.replay()
    torch_mod.cuda.synchronize()
    assert oracle.rel_err(_np(o1), ref) <= TOL
    eager = sf.bwdtrans_quad((nq, nq), b, b, x)         # allocates the counter buffer
    torch_mod.cuda.synchronize()
    assert torch_mod.equal(eager, o1)
    g2, o2 = captured()                                 # (b) a counter of its own, baked into the graph
    streams = [torch_mod.cuda.Stream() for _ in range(200)]
    small = sf.fill_random(4096, 9)
    for st in streams:                                  # scratch churn: 200 streams > the 128 scratch slots
        sf.sumsq(small, stream=st)
        sf.bwdtrans_quad((nq, nq), b, b, x[:8 * (nq - 1) ** 2], stream=st)
    torch_mod.cuda.synchronize()
    for rep in range(3):
        o2.zero_()
        g2.replay()
        other = sf.bwdtrans_quad((nq, nq), b, b, x, stream=streams[7])   # eager launch racing the replay
        torch_mod.cuda.synchronize()
        assert torch_mod.equal(other, eager), rep
        assert torch_mod.equal(o2, eager), (rep, int((o2 != eager).sum()))


def test_launch_hint_belongs_to_the_calling_thread(sf, oracle):
    """sf_set_launch_hint is per host thread (include/sumfact.h): a hint set by another thread does not reshape this
    thread's launches -- observable through the block-glb variant, which refuses a workspace-less call only after
    its launch shape is fixed; here simply: results stay right on both threads while they disagree about the hint."""
    import threading
    lib = sf.capi.lib()
    errs = {}

    def other():
        lib.sf_set_launch_hint(64, 9)
        errs["other"] = max(_hex_case(sf, oracle, (6, 6, 6), 301, v) for v in ("thread", "block-lds", "block-glb"))

    t = threading.Thread(target=other)
    t.start()
    errs["main"] = max(_hex_case(sf, oracle, (6, 6, 6), 301, v) for v in ("thread", "block-lds", "block-glb"))
    t.join()
    assert errs["main"] <= TOL and errs["other"] <= TOL


TOL32 = 2e-5   # fp32: eps = 6e-8, sums of up to 3*31 products with cancellation


def test_fp32_parity(sf, oracle, golden, torch_mod):
    """T = float instantiations: the oracle evaluates the same float inputs in fp64."""
    f32 = torch_mod.float32
    for nq in list(range(2, 12)) + [(3, 5, 4)]:
        nqs = (nq,) * 3 if isinstance(nq, int) else nq
        nm = [q - 1 for q in nqs]
        for nelmt in (1, 3, 17, 130, 1001):
            bs = [sf.fill_random(nm[d] * nqs[d], 31 + d, dtype=f32) for d in range(3)]
            x = sf.fill_random(nelmt * nm[0] * nm[1] * nm[2], nelmt, dtype=f32)
            out = sf.bwdtrans_hex(nqs, *bs, x)
            assert out.dtype == f32
            ref = oracle.bwdtrans_hex(nqs, nelmt, *[_np(b).astype(np.float64) for b in bs],
                                      _np(x).astype(np.float64))
            assert oracle.rel_err(_np(out).astype(np.float64), ref) <= TOL32, (nqs, nelmt)
    for nq in list(range(2, 25)) + [32, (4, 9)]:
        nqs = (nq, nq) if isinstance(nq, int) else nq
        nm = [q - 1 for q in nqs]
        for nelmt in (1, 5, 64, 999):
            bs = [sf.fill_random(nm[d] * nqs[d], 41 + d, dtype=f32) for d in range(2)]
            x = sf.fill_random(nelmt * nm[0] * nm[1], nelmt, dtype=f32)
            out = sf.bwdtrans_quad(nqs, *bs, x)
            ref = oracle.bwdtrans_quad(nqs, nelmt, *[_np(b).astype(np.float64) for b in bs],
                                       _np(x).astype(np.float64))
            assert oracle.rel_err(_np(out).astype(np.float64), ref) <= TOL32, (nqs, nelmt)
    # nq = 2 runs dedicated one-vector-per-thread stream kernels (3D, 2D): many XCD windows and a ragged tail
    for dim, nelmt in ((3, 70001), (2, 140003)):
        bs = [sf.fill_random(2, 51 + d, dtype=f32) for d in range(dim)]
        x = sf.fill_random(nelmt, 77, dtype=f32)
        b64 = [_np(b).astype(np.float64) for b in bs]
        if dim == 3:
            out, ref = sf.bwdtrans_hex((2, 2, 2), *bs, x), oracle.bwdtrans_hex((2, 2, 2), nelmt, *b64, _np(x).astype(np.float64))
        else:
            out, ref = sf.bwdtrans_quad((2, 2), *bs, x), oracle.bwdtrans_quad((2, 2), nelmt, *b64, _np(x).astype(np.float64))
        assert oracle.rel_err(_np(out).astype(np.float64), ref) <= TOL32, (dim, nelmt)
    # fills: the fp64 generator rounded to float; sin/cos in float
    a = _np(sf.fill_random(10007, 5, 3, dtype=f32))
    assert np.array_equal(a, oracle.fill_random(10007, 5, 3).astype(np.float32))
    s = _np(sf.fill_sincos(3, 343, dtype=f32))
    assert np.max(np.abs(s - np.sin(np.arange(1, 344, dtype=np.float32)).repeat(1)[None, :].repeat(3, 0).ravel())) < 1e-6
    # golden norm on the reference's data, to fp32 accuracy
    b = sf.fill_basis(7, 8, dtype=f32)
    x = sf.fill_sincos(4096, 343, dtype=f32)
    norm = math.sqrt(sf.sumsq(sf.bwdtrans_hex((8, 8, 8), b, b, b, x)))
    want = [float(r["norm"]) for r in golden["hex"]["8"]["rows"] if r["n"] == 4096][0]
    assert abs(norm - want) <= 2e-5 * want


def test_runtime_arbitrary_orders(sf, oracle, torch_mod):
    """The reference accepts any atoi order (benchmark05/benchmark05.cc:1425-1429) and its global-workspace kernels
    run it (:203-289).  AUTO above every table: LDS-resident generic kernel while one element's images fit the
    160 KiB of LDS (3D nq <= 21), then the library's own bounded scratch (one w1/w2 pair per workgroup); explicit
    BLOCK_LDS reports SF_ENOTBUILT instead of falling back."""
    for nq, nelmt in (((17, 17, 17), 37), ((20, 20, 20), 1000), ((21, 21, 21), 9), ((22, 22, 22), 7),
                      ((26, 26, 26), 3), ((24, 20, 18), 5), ((33, 2, 30), 11), ((40, 3, 3), 13)):
        assert _hex_case(sf, oracle, nq, nelmt, "auto") <= TOL, nq
        assert _hex_case(sf, oracle, nq, nelmt, "generic") <= TOL, nq
    # more elements than scratch slots (grid-stride reuse of a workgroup's w1/w2) and a second stream
    assert _hex_case(sf, oracle, (22, 22, 22), 2100, "auto") <= TOL
    side = torch_mod.cuda.Stream()
    with torch_mod.cuda.stream(side):
        assert _hex_case(sf, oracle, (23, 23, 23), 5, "auto") <= TOL
    side.synchronize()
    with pytest.raises(sf.capi.SumfactError) as ei:
        _hex_case(sf, oracle, (22, 22, 22), 3, "block-lds")
    assert ei.value.rc == sf.capi.SF_ENOTBUILT
    assert _hex_case(sf, oracle, (22, 22, 22), 3, "block-glb") <= TOL      # caller-owned workspace, reference layout
    for nq, nelmt in (((33, 33), 129), ((40, 40), 77), ((72, 72), 9), ((100, 100), 5), ((100, 3), 40)):
        assert _quad_case(sf, oracle, nq, nelmt, "auto") <= TOL, nq


def test_fp32_beyond_the_fp64_wave_tables(sf, oracle, torch_mod):
    """T = float (a template parameter of every reference kernel, benchmark05/benchmark05.cc:291): hex nq 12 runs the
    fp32 wave kernel, 13..16 the fp32 matrix-core kernel (v_mfma_f32_16x16x4_f32, hex_mfma_kernel with T = float), quad
    nq 25..32 the fp32 matrix-core kernel (quad_mfma_kernel with T = float); above that the generic kernels.  The fp32
    matrix instruction sums the q of a 16-row tile in the order r, r+4, r+8, r+12 (its D register map): tolerance 2e-5."""
    f32 = torch_mod.float32
    for nq in list(range(12, 18)) + [20, 22]:
        nm = nq - 1
        for nelmt in (1, 2, 7, 65, 130, 1000) if nq <= 16 else (5,):
            bs = [sf.fill_random(nm * nq, 31 + d, dtype=f32) for d in range(3)]
            x = sf.fill_random(nelmt * nm ** 3, nelmt, dtype=f32)
            out = sf.bwdtrans_hex((nq,) * 3, *bs, x)
            ref = oracle.bwdtrans_hex((nq,) * 3, nelmt, *[_np(b).astype(np.float64) for b in bs],
                                      _np(x).astype(np.float64))
            assert oracle.rel_err(_np(out).astype(np.float64), ref) <= TOL32, (nq, nelmt)
    for nq in list(range(25, 33)) + [33, 48]:
        nm = nq - 1
        for nelmt in (1, 2, 3, 5, 64, 999, 4099):
            bs = [sf.fill_random(nm * nq, 41 + d, dtype=f32) for d in range(2)]
            x = sf.fill_random(nelmt * nm * nm, nelmt, dtype=f32)
            out = sf.bwdtrans_quad((nq, nq), *bs, x)
            ref = oracle.bwdtrans_quad((nq, nq), nelmt, *[_np(b).astype(np.float64) for b in bs],
                                       _np(x).astype(np.float64))
            assert oracle.rel_err(_np(out).astype(np.float64), ref) <= TOL32, (nq, nelmt)


def test_sumsq_concurrent_streams(sf, oracle, torch_mod):
    """Reductions in flight on several streams / host threads of one device use per-stream scratch: every result
    equals the single-stream result bit for bit."""
    import threading
    xs = [sf.fill_random(1_000_003 + 4099 * i, 50 + i) for i in range(6)]
    want = [sf.sumsq(x) for x in xs]
    got = [[None] * 8 for _ in xs]

    def work(i):
        st = torch_mod.cuda.Stream()
        for rep in range(8):
            got[i][rep] = sf.sumsq(xs[i], stream=st)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(xs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i, w in enumerate(want):
        assert all(g == w for g in got[i]), i
        assert abs(w - oracle.sumsq(_np(xs[i]))) <= 1e-12 * w


def test_launch_hints(sf, oracle):
    """threads / elblocks (the reference CLI's knobs) change launch shapes of the baseline variants only;
    results stay correct for every combination."""
    lib = sf.capi.lib()
    try:
        for threads, elblocks in ((64, 1), (128, 1), (256, 4), (1024, 7), (1, 1000)):
            assert lib.sf_set_launch_hint(threads, elblocks) == 0
            for variant in ("thread", "block-lds", "block-glb", "auto"):
                assert _hex_case(sf, oracle, (8, 8, 8), 77, variant) <= TOL
                assert _quad_case(sf, oracle, (8, 8), 77, variant) <= TOL
    finally:
        lib.sf_set_launch_hint(0, 0)


def test_empty_and_errors(sf, torch_mod):
    capi = sf.capi
    b = sf.fill_basis(7, 8)
    empty = torch_mod.empty(0, dtype=torch_mod.float64, device="cuda")
    assert sf.bwdtrans_hex((8, 8, 8), b, b, b, empty).numel() == 0
    assert sf.bwdtrans_quad((8, 8), b, b, empty).numel() == 0
    with pytest.raises(capi.SumfactError) as ei:
        sf.bwdtrans_hex((1, 8, 8), b, b, b, empty)
    assert ei.value.rc == capi.SF_EINVAL


def test_unaligned_input_falls_back(sf, oracle):
    """`in` only 8-byte aligned: AUTO must still be correct (generic path), WAVE refuses."""
    nq, nelmt = 8, 11
    b = sf.fill_basis(7, 8)
    buf = sf.fill_random(nelmt * 343 + 1, 5)
    x = buf[1:]
    out = sf.bwdtrans_hex((nq,) * 3, b, b, b, x)
    ref = oracle.bwdtrans_hex((nq,) * 3, nelmt, _np(b), _np(b), _np(b), _np(x))
    assert oracle.rel_err(_np(out), ref) <= TOL
    with pytest.raises(sf.capi.SumfactError) as ei:
        sf.bwdtrans_hex((nq,) * 3, b, b, b, x, variant="wave")
    assert ei.value.rc == sf.capi.SF_EALIGN


def test_hex_linearity_and_idempotence_full_size(sf, torch_mod):
    """BASELINE full size (1 048 576 elements, nq=8): properties that need no CPU reference:
    T(a*x + y) == a*T(x) + T(y) to rounding, and two runs are bit-identical."""
    nq, nelmt = 8, 1 << 20
    nm = nq - 1
    b = sf.fill_basis(nm, nq)
    x = sf.fill_random(nelmt * nm ** 3, 11)
    y = sf.fill_random(nelmt * nm ** 3, 12)
    tx = sf.bwdtrans_hex((nq,) * 3, b, b, b, x)
    ty = sf.bwdtrans_hex((nq,) * 3, b, b, b, y)
    z = 0.5 * x + y
    tz = sf.bwdtrans_hex((nq,) * 3, b, b, b, z)
    lin = 0.5 * tx + ty
    err = float((tz - lin).abs().max() / lin.abs().max())
    assert err <= TOL, err
    tx2 = sf.bwdtrans_hex((nq,) * 3, b, b, b, x)
    assert torch_mod.equal(tx, tx2)


def test_full_size_elementwise_parity(sf, oracle):
    """BASELINE configs 1 and 2 at their full size (1 048 576 elements, nq = 8), element by element
    against the oracle on seeded per-element-distinct data -- not only norms and properties."""
    nelmt, nq, nm = 1 << 20, 8, 7
    oracle.set_threads(oracle.usable_cpus())
    b = [sf.fill_random(nm * nq, 900 + d) for d in range(3)]
    bh = [_np(v) for v in b]
    x = sf.fill_random(nelmt * nm ** 3, 4242)
    out = sf.bwdtrans_hex((nq,) * 3, *b, x)
    ref = oracle.bwdtrans_hex((nq,) * 3, nelmt, *bh, _np(x), form="vector")
    assert oracle.rel_err(_np(out), ref) <= TOL
    del out, ref, x
    x = sf.fill_random(nelmt * nm ** 2, 4243)
    out = sf.bwdtrans_quad((nq, nq), b[0], b[1], x)
    ref = oracle.bwdtrans_quad((nq, nq), nelmt, bh[0], bh[1], _np(x))
    assert oracle.rel_err(_np(out), ref) <= TOL


@pytest.mark.parametrize("nq", [2, 3, 4, 5, 6, 7, 9, 10])
def test_full_size_sampled_parity_hex_sweep(sf, oracle, nq):
    """BASELINE config 3 (nq sweep at 1 048 576 elements): head, tail and two interior 4096-element
    windows of the full-size output against the oracle (windows straddle chunk / workgroup seams)."""
    nelmt, nm = 1 << 20, nq - 1
    b = [sf.fill_random(nm * nq, 300 + nq + d) for d in range(3)]
    bh = [_np(v) for v in b]
    x = sf.fill_random(nelmt * nm ** 3, 5000 + nq)
    out = sf.bwdtrans_hex((nq,) * 3, *b, x)
    for lo in (0, 333_333, 777_001, nelmt - 4096):
        xs = _np(x[lo * nm ** 3:(lo + 4096) * nm ** 3])
        ref = oracle.bwdtrans_hex((nq,) * 3, 4096, *bh, xs)
        got = _np(out[lo * nq ** 3:(lo + 4096) * nq ** 3])
        assert oracle.rel_err(got, ref) <= TOL, (nq, lo)


def test_hex_64bit_indexing(sf, oracle, torch_mod):
    """More than 2^32 output doubles (the reference's `unsigned` index overflows above 8 388 608
    elements at nq=8): check the tail elements against the oracle and a checksum identity."""
    nq, nelmt = 8, 8388608 + 1030
    nm = nq - 1
    free, _ = torch_mod.cuda.mem_get_info()
    need = 8 * nelmt * (nm ** 3 + nq ** 3) + (1 << 30)
    if free < need:
        pytest.skip("not enough device memory")
    b = sf.fill_basis(nm, nq)
    x = sf.fill_random(nelmt * nm ** 3, 77)
    out = sf.bwdtrans_hex((nq,) * 3, b, b, b, x)
    tail = 1500
    xt = _np(x[(nelmt - tail) * nm ** 3:])
    ref = oracle.bwdtrans_hex((nq,) * 3, tail, _np(b), _np(b), _np(b), xt)
    got = _np(out[(nelmt - tail) * nq ** 3:])
    assert oracle.rel_err(got, ref) <= TOL
    # head too
    ref0 = oracle.bwdtrans_hex((nq,) * 3, 64, _np(b), _np(b), _np(b), _np(x[:64 * nm ** 3]))
    assert oracle.rel_err(_np(out[:64 * nq ** 3]), ref0) <= TOL
    # sin/cos data: every element identical -> sum of squares = nelmt * one element's
    del x, out
    x = sf.fill_sincos(nelmt, nm ** 3)
    out = sf.bwdtrans_hex((nq,) * 3, b, b, b, x)
    ss = sf.sumsq(out)
    ss1 = sf.sumsq(out[:nq ** 3])
    assert abs(ss - nelmt * ss1) <= 1e-11 * ss


@pytest.mark.parametrize("nq", [7, 8])
def test_large_batches_enqueued_in_pieces(sf, oracle, torch_mod, nq):
    """From 2 x 524 288 elements up, nq = 7 / 8 batches are enqueued as back-to-back launches of 524 288
    elements (wave_table.h hex_piece(), bwdtrans_hex.hip): windows across every launch seam and over the short
    last piece against the oracle, and the whole output bit-identical to the same batch computed in two halves
    that stay below the threshold."""
    piece, nm = 1 << 19, nq - 1
    nelmt = 2 * piece + 4099                       # pieces: 524 288 + 524 288 + 4 099
    b = [sf.fill_random(nm * nq, 400 + nq + d) for d in range(3)]
    bh = [_np(v) for v in b]
    x = sf.fill_random(nelmt * nm ** 3, 6000 + nq)
    out = sf.bwdtrans_hex((nq,) * 3, *b, x)
    win = 300
    for lo in (0, piece - win // 2, 2 * piece - win // 2, nelmt - win):
        ref = oracle.bwdtrans_hex((nq,) * 3, win, *bh, _np(x[lo * nm ** 3:(lo + win) * nm ** 3]))
        assert oracle.rel_err(_np(out[lo * nq ** 3:(lo + win) * nq ** 3]), ref) <= TOL, (nq, lo)
    cut = 600_002                                   # two single launches (each <= 2 pieces), seam elsewhere
    halves = torch_mod.cat([sf.bwdtrans_hex((nq,) * 3, *b, x[:cut * nm ** 3]),
                            sf.bwdtrans_hex((nq,) * 3, *b, x[cut * nm ** 3:])])
    assert torch_mod.equal(out, halves)
