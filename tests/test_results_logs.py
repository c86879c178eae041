"""The logs committed under results/ (the counterpart of the logs the reference commits next to its
sources) must parse with the reference's grammar and carry the reference's published `norm:` values in
EVERY column -- host and device -- wherever the reference published one.  CPU only: reads committed text."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "results")
TOL = 5.5e-10  # the published values carry 10 significant digits


@pytest.fixture(scope="module")
def pkg():
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_norms.json")) as fh:
        return json.load(fh)


def _check(pkg, path, want, max_cols):
    log = pkg.logfmt.parse_log(open(path).read())
    assert 1 <= log.ncols <= max_cols, path
    checked = 0
    for size, norms in zip(log.sizes, log.norms):
        if int(size) not in want:
            continue
        for v in norms:
            assert abs(v - want[int(size)]) <= TOL * want[int(size)], (path, size, norms)
            checked += 1
    return checked


def test_hex_and_quad_logs(pkg, golden):
    n = 0
    for kind, sub, pat in (("hex", "benchmark05", "nq{0}x{0}x{0}.log"), ("quad", "benchmark04", "nq{0}x{0}.log")):
        for nq, rec in golden[kind].items():
            want = {int(r["n"]): float(r["norm"]) for r in rec["rows"]}
            n += _check(pkg, os.path.join(RES, sub, pat.format(nq)), want, 11)
    assert n >= 14 * 14 * 5          # 14 logs x 14 sizes x >= 5 columns


def test_every_committed_log_parses(pkg):
    logs = glob.glob(os.path.join(RES, "benchmark0[45]", "*.log"))
    assert len(logs) >= 21
    for path in logs:
        log = pkg.logfmt.parse_log(open(path).read())
        assert log.kind == "DOF/s" and len(log.sizes) == 14 and log.sizes[-1] == 1048576.0, path
        # all columns of one size agree with each other (same maths, different kernels) to the printed digits
        for norms in log.norms:
            assert max(norms) - min(norms) <= 2.0 * TOL * max(norms), (path, norms)


def test_stream_benchmark_logs(pkg, golden):
    for key, sub in (("l2norm", "benchmark01"), ("vecadd", "benchmark02"), ("matvec", "benchmark03")):
        want = {int(r["n"]): float(r["norm"]) for r in golden[key]["rows"]}
        assert _check(pkg, os.path.join(RES, sub, "outfile.log"), want, 5) >= 2 * len(want)
