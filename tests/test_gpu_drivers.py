"""GPU: the C++ drivers (reference CLI + log grammar) run on the device, their logs parse under the
reference's postprocess grammar, and their `norm:` columns reproduce the published values."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "gpu-benchmarking_amd", "bin")


@pytest.fixture(scope="module")
def pkg():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    return ge.load_package()


def _run(args, timeout=900):
    res = subprocess.run(args, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-2000:]
    return res.stdout


def test_benchmark05_default_cli_sweep(pkg, golden, tmp_path):
    js = tmp_path / "b5.json"
    out = _run([os.path.join(BIN, "benchmark05"), "8", "8", "8", "--max-size", "16384",
                "--json", str(js)])
    lines = out.splitlines()
    assert lines[:4] == ["-" * 32, "Benchmark05 : BwdTrans (3D)     ", "-" * 32,
                         "BwdTrans (NQ = 8, 8, 8)"]
    log = pkg.logfmt.parse_log(out)
    assert log.kind == "DOF/s" and log.ncols == 6 and log.title == "BwdTrans (NQ = 8, 8, 8)"
    assert log.sizes == [float(128 << k) for k in range(8)]
    want = {r["n"]: float(r["norm"]) for r in golden["hex"]["8"]["rows"]}
    for size, norms in zip(log.sizes, log.norms):
        for v in norms:                       # every variant reproduces the published norm
            assert abs(v - want[int(size)]) <= 5.5e-10 * want[int(size)], (size, norms)
    rec = json.loads(js.read_text())
    assert rec["benchmark"] == "benchmark05" and len(rec["rows"]) == 8
    # side file: per column the wall-clock minimum the stdout rows are made of AND the HIP-event minimum of the same
    # launches (kernel level: no launch + synchronise floor), SURVEY s7 step 5
    for row, values in zip(rec["rows"], log.values):
        assert len(row["t_wall_min"]) == 6 and len(row["t_event_min"]) == 6 and len(row["gdof_s_event"]) == 6
        assert all(0 < e <= 1.5 * w for e, w in zip(row["t_event_min"], row["t_wall_min"]))
        assert abs(row["wave_gdof_s"] - values[3]) <= 1e-6 * values[3]
        assert row["wave_gdof_s_event"] >= 0.9 * row["wave_gdof_s"]


def test_benchmark05_headline_size(pkg, golden):
    out = _run([os.path.join(BIN, "benchmark05"), "8", "8", "8", "128", "1", "--nelmt", "1048576",
                "--no-baselines"])
    log = pkg.logfmt.parse_log(out)
    assert log.sizes == [1048576.0]
    assert abs(log.norms[0][3] - 17134.76235) <= 5.5e-10 * 17134.76235
    assert log.values[0][3] > 100.0           # GDOF/s of the flagship column (reference best: 26.4)


def test_benchmark04_default_cli_sweep(pkg, golden):
    out = _run([os.path.join(BIN, "benchmark04"), "8", "8", "--max-size", "16384"])
    lines = out.splitlines()
    assert lines[1] == "Benchmark04 : BwdTrans (2D)     " and lines[3] == "BwdTrans (NQ = 8, 8)"
    log = pkg.logfmt.parse_log(out)
    assert log.ncols == 5
    want = {r["n"]: float(r["norm"]) for r in golden["quad"]["8"]["rows"]}
    for size, norms in zip(log.sizes, log.norms):
        for v in norms:
            assert abs(v - want[int(size)]) <= 5.5e-10 * want[int(size)], (size, norms)


def test_benchmark01_device_column(pkg, golden):
    out = _run([os.path.join(BIN, "benchmark01"), "--max-size", str(1 << 24)])
    log = pkg.logfmt.parse_log(out)
    assert log.ncols == 2
    want = {r["n"]: float(r["norm"]) for r in golden["l2norm"]["rows"]}
    for size, norms in zip(log.sizes, log.norms):
        for v in norms:                       # host and device columns
            assert abs(v - want[int(size)]) <= 5.5e-10 * want[int(size)]


def test_benchmark02_and_03_device_columns(pkg, golden):
    out = _run([os.path.join(BIN, "benchmark02"), "--max-size", str(1 << 24)])
    log = pkg.logfmt.parse_log(out)
    want = {r["n"]: float(r["norm"]) for r in golden["vecadd"]["rows"]}
    for size, norms in zip(log.sizes, log.norms):
        for v in norms:
            assert abs(v - want[int(size)]) <= 5.5e-10 * want[int(size)], (size, norms)
    out = _run([os.path.join(BIN, "benchmark03"), "--max-size", "4096"])
    log = pkg.logfmt.parse_log(out)
    want = {r["n"]: float(r["norm"]) for r in golden["matvec"]["rows"]}
    for size, norms in zip(log.sizes, log.norms):
        for v in norms:
            assert abs(v - want[int(size)]) <= 5.5e-10 * want[int(size)], (size, norms)


def test_benchmark01_default_problem_size(pkg, golden):
    """BASELINE configs[0] at its DEFAULT size: `benchmark01` with no arguments runs size = 1024 .. 536 870 912 doubles
    (benchmark01/benchmark01.cc:343; 4.3 GB of host memory), host column on the granted cores and device column, and
    both reproduce all 20 published norms (benchmark01/outfile.log) -- so config 0 rests on a run, not on a
    committed log."""
    import oracle
    env = dict(os.environ, OMP_NUM_THREADS=str(oracle.usable_cpus()))
    res = subprocess.run([os.path.join(BIN, "benchmark01")], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    log = pkg.logfmt.parse_log(res.stdout)
    assert log.kind == "GB/s" and log.ncols == 2 and len(log.sizes) == 20
    assert log.sizes[0] == 1024.0 and log.sizes[-1] == 536870912.0
    want = {r["n"]: float(r["norm"]) for r in golden["l2norm"]["rows"]}
    assert len(want) == 20
    for size, norms, rates in zip(log.sizes, log.norms, log.values):
        for v in norms:                                   # host AND device column
            assert abs(v - want[int(size)]) <= 5.5e-10 * want[int(size)], (size, norms)
        assert all(r > 0 for r in rates)
    assert log.values[-1][1] > 2000.0                     # device column at 4.3 GB: measured 6.2-7.0 TB/s


def test_bench_contract_single_gpu(pkg):
    """`python bench.py` (N = 1): one JSON line with the contract keys, the roofline and cpu_baseline
    objects, the golden-norm self check, and a loose performance floor (reference's best: 26.4)."""
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3",
                          "--cpu-seconds", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # exactly ONE line on stdout
    rec = json.loads(lines[0])
    assert rec["metric"].startswith("GDOF/s for 3D hex") and rec["unit"] == "GDOF/s"
    assert rec["n_gpus"] == 1 and rec["steps"] == 20 and rec["warmup"] == 3
    assert rec["higher_is_better"] is True and rec["vs_baseline"] is None and rec["dtype"] == "f64"
    assert "workload" in rec["config"] and "model" not in rec["config"]
    rl = rec["roofline"]
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and rl["unit"] == "GB/s"
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    assert abs(rec["value"] * 19.94169 - rl["achieved"]) / rl["achieved"] < 0.02   # B/DOF consistency
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert rec["golden_norm_check"]["ok"] is True
    assert set(rec["extra"]["hex_sweep"]) == {str(n) for n in range(2, 11)}
    for entry in list(rec["extra"]["hex_sweep"].values()) + list(rec["extra"]["quad"].values()):
        assert entry["frac"] == entry["frac_mean"] <= entry["frac_min"]       # one protocol per number
    assert rec["ms_min_of_40"] <= rec["ms_mean_of_40_groups"] and rl["kernel_ms"] <= rec["ms_per_step"] * 1.001
    assert {"26", "30"} <= set(rec["extra"]["quad"])
    assert rec["value"] > 150.0 and rl["frac"] > 0.35   # measured 280-300 / 0.70-0.74


def test_bench_two_ranks_rehearsal(pkg, tmp_path):
    """bench.py's N>1 path on real hardware: 2 ranks share the one GPU of this box, scalar reductions
    over gloo (SF_BENCH_BACKEND); checks the contract keys and that the shards add up."""
    import sys
    env = dict(os.environ, SF_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "1", "--total-elements", "200001", "--cpu-seconds", "2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["dtype"] == "f64"
    assert rec["config"]["total_elements"] == 200001 and rec["value"] > 1.0
    # the N > 1 line carries the CPU baseline of the same run, per-GPU fractions and the speed-up over one GPU
    assert rec["cpu_baseline"]["value"] > 0 and len(rec["roofline"]["per_gpu_frac"]) == 2
    assert rec["speedup_vs_1gpu_same_batch"] > 0 and "traffic" in rec["roofline"]
    # the two shards together are the single-rank batch: same checksum as one rank over all elements
    x = pkg.fill_random(200001 * 343, 0x5F3759DF, 0)
    b = pkg.fill_basis(7, 8)
    import math
    full = math.sqrt(pkg.sumsq(pkg.bwdtrans_hex((8, 8, 8), b, b, b, x)))
    assert abs(rec["checksum_norm"] - full) <= 1e-12 * full


def test_bench_rccl_collectives_with_one_rank(pkg):
    """The RCCL half of bench.py's N > 1 path on the one GPU of this box: under the launcher with a single rank and
    SF_BENCH_FORCE_DIST=1 the run builds the nccl (= RCCL) communicator on its device and goes through the barrier,
    both all-reduces and the all-gather exactly as an N-rank job does (the 2-rank rehearsal above covers the sharding
    over gloo; RCCL refuses two ranks on one device)."""
    import sys
    env = dict(os.environ, SF_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SF_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "5", "--warmup", "1", "--elements-per-gpu", "65536", "--no-extra",
           "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["config"]["backend"] == "rccl"
    assert rec["config"]["collectives"] == "nccl communicator of 1 rank(s)"
    assert rec["config"]["total_elements"] == 65536 and rec["value"] > 1.0
    import math
    x = pkg.fill_random(65536 * 343, 0x5F3759DF, 0)
    b = pkg.fill_basis(7, 8)
    full = math.sqrt(pkg.sumsq(pkg.bwdtrans_hex((8, 8, 8), b, b, b, x)))
    assert abs(rec["checksum_norm"] - full) <= 1e-12 * full


def test_anisotropic_cli(pkg, oracle):
    """nq0 != nq1 != nq2 takes the generic path; norm checked against the oracle."""
    import math
    out = _run([os.path.join(BIN, "benchmark05"), "3", "5", "4", "--nelmt", "1000"])
    log = pkg.logfmt.parse_log(out)
    b = [oracle.fill_basis(q - 1, q) for q in (3, 5, 4)]
    ref = oracle.bwdtrans_hex((3, 5, 4), 1000, *b, oracle.fill_sincos(1000, 2 * 4 * 3))
    norm = math.sqrt(oracle.sumsq(ref))
    for v in log.norms[0]:
        assert abs(v - norm) <= 1e-9 * norm           # all six columns incl. rocBLAS, interleaved


def test_benchmark05_any_order_cli(pkg, oracle):
    """`./benchmark05 20 20 20` and `22 22 22` (the reference takes any atoi order, benchmark05.cc:1425-1429):
    every built column reproduces the oracle's norm; at nq = 22 the LDS-resident baseline no longer fits the 160 KiB
    of LDS and its column prints 0 instead of ending the run."""
    import math
    for nq, nelmt in ((20, 1000), (22, 300)):
        out = _run([os.path.join(BIN, "benchmark05"), str(nq), str(nq), str(nq), "--nelmt", str(nelmt)])
        log = pkg.logfmt.parse_log(out)
        b = oracle.fill_basis(nq - 1, nq)
        ref = oracle.bwdtrans_hex((nq,) * 3, nelmt, b, b, b, oracle.fill_sincos(nelmt, (nq - 1) ** 3))
        norm = math.sqrt(oracle.sumsq(ref))
        for col, (v, rate) in enumerate(zip(log.norms[0], log.values[0])):
            if nq == 22 and col == 2:
                assert v == 0.0 and rate == 0.0
            else:
                assert abs(v - norm) <= 1e-9 * norm and rate > 0.0, (nq, col, v, norm)


def test_benchmark05_fp32_above_the_fp64_tables(pkg, oracle):
    """`--precision f32` at an order the fp64 wave table does not hold (nq = 14: fp32 wave kernel)."""
    import math
    out = _run([os.path.join(BIN, "benchmark05"), "14", "14", "14", "--nelmt", "4096", "--precision", "f32"])
    log = pkg.logfmt.parse_log(out)
    b = oracle.fill_basis(13, 14)
    ref = oracle.bwdtrans_hex((14,) * 3, 4096, b, b, b, oracle.fill_sincos(4096, 13 ** 3))
    norm = math.sqrt(oracle.sumsq(ref))
    assert abs(log.norms[0][3] - norm) <= 5e-5 * norm and log.values[0][3] > 50.0


def test_c99_consumer_of_the_boundary(pkg):
    """examples/consumer.c: C99, built with gcc against include/sumfact.h + libsumfact.so + libamdhip64 (INTEGRATION.md
    s1 as written): hipMalloc -> sf_fill_* -> sf_bwdtrans_hex_f64 -> sf_sumsq_f64 reproduces the reference's published
    norm 17134.76235 (benchmark05/nq8x8x8.log:45) and exits 0."""
    exe = os.path.join(ROOT, "examples", "consumer")
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "-s", "consumer"], check=True)
    out = _run([exe])
    assert out.split() == ["nelmt", "1048576", "norm:", "17134.76235"]


def test_benchmark05_ngpus_row(pkg, golden, tmp_path):
    """`--ngpus N`: one process, N devices, RCCL MAX(time) / SUM(sum of squares) (host/multigpu.h).  This box has one
    GPU: N = the device count runs the whole multi-device path (communicator, streams, shard ranges, both
    all-reduces) and must reproduce the published norm; N above the device count refuses with exit code 5 instead of
    printing an aggregate row for fewer GPUs."""
    import torch
    ndev = torch.cuda.device_count()
    js = tmp_path / "multi.json"
    env = dict(os.environ, SF_FORCE_MULTIGPU_PATH="1")
    res = subprocess.run([os.path.join(BIN, "benchmark05"), "8", "8", "8", "--nelmt", "1048576", "--ngpus", str(ndev),
                          "--json", str(js)], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    log = pkg.logfmt.parse_log(res.stdout)
    assert log.ncols == 6 and log.sizes == [1048576.0]
    assert abs(log.norms[0][3] - 17134.76235) <= 5.5e-10 * 17134.76235       # benchmark05/nq8x8x8.log:45
    assert log.values[0][3] > 100.0 * ndev and all(v == 0.0 for i, v in enumerate(log.values[0]) if i != 3)
    rec = json.loads(js.read_text())
    assert rec["ngpus"] == ndev and len(rec["rows"][0]["per_device_ms"]) == ndev
    row = rec["rows"][0]
    # the printed row is the host-wall figure (same clock as the single-GPU rows); events and the speed-up over device
    # 0 alone on the same batch are in the side file
    assert abs(row["wave_gdof_s"] - log.values[0][3]) <= 1e-6 * log.values[0][3]
    assert row["wave_gdof_s_event"] >= 0.95 * row["wave_gdof_s"]
    assert row["single_gpu_same_batch_gdof_s"] > 100.0 and row["speedup_vs_1gpu_same_batch"] > 0.5 * ndev
    assert row["scaling_verified_on_hardware"] is (ndev > 1)
    # seeded data is generated from the global element index: the sharded norm equals the one-GPU norm
    res = subprocess.run([os.path.join(BIN, "benchmark05"), "8", "8", "8", "--nelmt", "300001", "--ngpus", str(ndev),
                          "--data", "random"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    import math
    x = pkg.fill_random(300001 * 343, 0x5F3759DF, 0)
    b = pkg.fill_basis(7, 8)
    want = math.sqrt(pkg.sumsq(pkg.bwdtrans_hex((8, 8, 8), b, b, b, x)))
    got = pkg.logfmt.parse_log(res.stdout).norms[0][3]
    assert abs(got - want) <= 5.5e-10 * want
    res = subprocess.run([os.path.join(BIN, "benchmark05"), "8", "8", "8", "--nelmt", "1000", "--ngpus", str(ndev + 1)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 5 and "refusing" in res.stderr
