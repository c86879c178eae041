"""Log-grammar contract: our drivers' logs must parse under the grammar the reference's postprocess.py
defines (line filters, split()[1], split()[3:], equal column counts, <= 11 / <= 5 columns), and the
reference's own logs must parse with the same code.  The excerpts under tests/golden are result data
of the reference (first lines of three committed logs)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "gpu-benchmarking_amd", "bin")


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(BIN, "benchmark01")):
        ge.build()
    return ge.load_package()


def test_reference_excerpts_parse(pkg, golden):
    ex = golden["log_excerpts"]
    hexlog = pkg.logfmt.parse_log(ex["benchmark05/nq8x8x8.log"])
    assert hexlog.kind == "DOF/s" and hexlog.ncols == 11
    assert hexlog.title == "BwdTrans (NQ = 8, 8, 8)"
    assert hexlog.sizes == [128.0, 256.0]
    assert abs(hexlog.norms[0][8] - 189.3141665) < 1e-9
    quad = pkg.logfmt.parse_log(ex["benchmark04/nq8x8.log"])
    assert quad.ncols == 11 and quad.title == "BwdTrans (NQ = 8, 8)"
    l2 = pkg.logfmt.parse_log(ex["benchmark01/outfile.log"])
    assert l2.kind == "GB/s" and l2.ncols == 5 and l2.sizes == [1024.0, 2048.0]
    assert abs(l2.norms[0][0] - 231.3925755) < 1e-9


def test_grammar_violations_rejected(pkg):
    with pytest.raises(ValueError):
        pkg.logfmt.parse_log("nothing here\n")
    bad = "nelmt 128 DOF/s: 1 2 3\nnelmt 256 DOF/s: 1 2\n"
    with pytest.raises(ValueError):
        pkg.logfmt.parse_log(bad)
    wide = "nelmt 128 DOF/s: " + " ".join(["1"] * 12) + "\n"
    with pytest.raises(ValueError):
        pkg.logfmt.parse_log(wide)


def test_benchmark01_host_path_log(pkg, golden, tmp_path):
    """BASELINE config 0: benchmark01 host path, no GPU needed: banner, grammar, published norms."""
    exe = os.path.join(BIN, "benchmark01")
    js = tmp_path / "bm01.json"
    # up to 2^26 values: beyond 2^25 the leaf sums outnumber 8192 and the reduction tree's upper levels matter
    # (an earlier version parallelised the in-place tree there and raced)
    env = dict(os.environ, OMP_NUM_THREADS="8")
    res = subprocess.run([exe, "--max-size", str(1 << 26), "--json", str(js)], capture_output=True,
                         text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr
    out = res.stdout
    lines = out.splitlines()
    assert lines[0] == "-" * 32 and lines[1] == "Benchmark01 : L2 norm reduction " and lines[2] == "-" * 32
    log = pkg.logfmt.parse_log(out)
    assert log.kind == "GB/s" and log.ncols == 2 and len(log.sizes) == 17
    assert log.sizes[0] == 1024.0 and log.sizes[-1] == float(1 << 26)
    # three lines per size
    assert len(lines) == 3 + 3 * len(log.sizes)
    want = {row["n"]: float(row["norm"]) for row in golden["l2norm"]["rows"]}
    for size, norms in zip(log.sizes, log.norms):
        assert abs(norms[0] - want[int(size)]) <= 5.5e-10 * want[int(size)]
    assert js.exists() and '"benchmark": "benchmark01"' in js.read_text()


def test_benchmark02_and_03_host_paths(pkg, golden):
    """Section 8(f) drivers: same banner/grammar contract, host columns reproduce the published norms."""
    env = dict(os.environ, OMP_NUM_THREADS="4")
    res = subprocess.run([os.path.join(BIN, "benchmark02"), "--max-size", str(1 << 18)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr
    assert res.stdout.splitlines()[1] == "Benchmark02 : Vector Addition   "
    log = pkg.logfmt.parse_log(res.stdout)
    want = {r["n"]: float(r["norm"]) for r in golden["vecadd"]["rows"]}
    assert log.kind == "GB/s" and log.ncols == 2 and len(log.sizes) == 9
    for size, norms in zip(log.sizes, log.norms):
        assert abs(norms[0] - want[int(size)]) <= 5.5e-10 * want[int(size)]
    res = subprocess.run([os.path.join(BIN, "benchmark03"), "--max-size", "1024"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr
    assert res.stdout.splitlines()[1] == "Benchmark03 : Matrix-Vector Mult"
    log = pkg.logfmt.parse_log(res.stdout)
    want = {r["n"]: float(r["norm"]) for r in golden["matvec"]["rows"]}
    assert log.sizes == [128.0, 256.0, 512.0, 1024.0]
    for size, norms in zip(log.sizes, log.norms):
        assert abs(norms[0] - want[int(size)]) <= 5.5e-10 * want[int(size)]
    ref2 = pkg.logfmt.parse_log(golden["log_excerpts"]["benchmark02/outfile.log"])
    ref3 = pkg.logfmt.parse_log(golden["log_excerpts"]["benchmark03/outfile.log"])
    assert ref2.ncols == 5 and ref3.ncols == 5 and ref3.sizes[0] == 128.0


def test_postprocess_plots(pkg, golden, tmp_path):
    pytest.importorskip("matplotlib")
    import importlib
    pp = importlib.import_module("gpu_benchmarking_amd.postprocess")
    path = tmp_path / "nq8x8x8.log"
    path.write_text(golden["log_excerpts"]["benchmark05/nq8x8x8.log"])
    png = pp.plot(str(path), roofline=True)
    assert png.endswith("nq8x8x8.png") and os.path.getsize(png) > 1000
    assert abs(pp.roofline_gdofs("BwdTrans (NQ = 8, 8, 8)") - 401.17) < 0.01
    assert abs(pp.roofline_gdofs("BwdTrans (NQ = 8, 8)") - 433.63) < 0.01


def test_ngpus_argument_is_validated_before_any_gpu_call():
    """`benchmark05 --ngpus 0` / a non-number: exit 1 with a message, on a box with or without a GPU."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-benchmarking_amd", "bin",
                       "benchmark05")
    if not os.path.exists(exe):
        import pytest
        pytest.skip("drivers not built")
    for bad in ("0", "x", "65"):
        res = subprocess.run([exe, "8", "8", "8", "--ngpus", bad], capture_output=True, text=True, timeout=60)
        assert res.returncode == 1 and "--ngpus" in res.stderr
