"""BASELINE configs[4]: benchmark05 3D hex nq = 8, 10 000 000 elements sharded over 8 GPUs.

The reference is single-GPU (benchmark05/run.sh:7) and its 32-bit indexing cannot even address this batch
(nelmt*nq^3 > 2^32, benchmark05/benchmark05.cc:94); the sharding is this build's.  An 8-GPU node is the
driver's to launch, so on the one-GPU box the tests run (a) rank 7-of-8's shard exactly as bench.py would on
that rank, against the oracle, and (b) the whole 10 M batch on one GPU (68 GB) against all eight shards run
separately: the union of shard outputs must be bit-identical to the single-GPU output, and the reduced
checksum must equal the single-GPU checksum.  CPU tests cover bench.py's launch rules (no GPU call involved).
"""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL, WORLD, NQ, NM = 10_000_000, 8, 8, 7
SEED = 0x5F3759DF          # bench.py's seed
NMT, NQT = NM ** 3, NQ ** 3


@pytest.fixture(scope="module")
def sf():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    return ge.load_package()


def _bench():
    import importlib
    return importlib.import_module("bench")


# ---------------------------------------------------------------- CPU: launch rules ------------------------

def test_gpus_flag_and_launcher_must_agree():
    b = _bench()
    assert b.resolve_world(b.parse([]), {}) == (1, False)
    assert b.resolve_world(b.parse(["--gpus", "1"]), {}) == (1, False)
    assert b.resolve_world(b.parse(["--gpus", "8"]), {}) == (8, True)          # no launcher: spawn 8 ranks
    assert b.resolve_world(b.parse(["--gpus", "8"]), {"WORLD_SIZE": "8"}) == (8, False)
    assert b.resolve_world(b.parse([]), {"WORLD_SIZE": "4"}) == (4, False)
    with pytest.raises(SystemExit) as exc:                                     # never n_gpus: 1 for --gpus 8
        b.resolve_world(b.parse(["--gpus", "8"]), {"WORLD_SIZE": "1"})
    assert "WORLD_SIZE=1" in str(exc.value)
    with pytest.raises(SystemExit):
        b.resolve_world(b.parse(["--gpus", "2"]), {"WORLD_SIZE": "4"})
    with pytest.raises(SystemExit):
        b.resolve_world(b.parse(["--gpus", "0"]), {})


def test_default_workloads():
    b = _bench()
    assert b.pick_workload(b.parse([]), 1) == ("weak", 1 << 20, 1 << 20)
    for n in (2, 4, 8):                                                       # config 4 is the N > 1 default
        assert b.pick_workload(b.parse([]), n) == ("strong", TOTAL, None)
    assert b.pick_workload(b.parse(["--elements-per-gpu", "1000"]), 4) == ("weak", 4000, 1000)
    assert b.pick_workload(b.parse(["--total-elements", "77"]), 2) == ("strong", 77, None)
    with pytest.raises(SystemExit):
        b.pick_workload(b.parse(["--total-elements", "7", "--elements-per-gpu", "7"]), 2)


def test_mismatched_launch_exits_nonzero_without_touching_a_gpu():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr and not res.stdout.strip()


def _fake_topology(root, simd_counts):
    for i, simd in enumerate(simd_counts):
        d = root / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nunique_id {1000 + i}\n")
    return str(root)


def test_gpus_are_counted_from_sysfs_not_from_hip(tmp_path):
    """The spawning parent counts GPUs from the KFD topology (nodes with simd_count > 0; CPU nodes report 0) and
    applies the visibility variables itself -- no HIP call, no torch import."""
    b = _bench()
    nodes = _fake_topology(tmp_path, [0, 0, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024])
    assert b.visible_gpu_count({}, nodes) == 8
    assert b.visible_gpu_count({"HIP_VISIBLE_DEVICES": "0,1,2,3"}, nodes) == 4
    assert b.visible_gpu_count({"ROCR_VISIBLE_DEVICES": "2,5", "HIP_VISIBLE_DEVICES": "0,1,7"}, nodes) == 2
    assert b.visible_gpu_count({"CUDA_VISIBLE_DEVICES": "1"}, nodes) == 1
    assert b.visible_gpu_count({"CUDA_VISIBLE_DEVICES": "1", "HIP_VISIBLE_DEVICES": "0,1"}, nodes) == 2   # HIP_ wins
    assert b.visible_gpu_count({"HIP_VISIBLE_DEVICES": "0,9,1"}, nodes) == 1     # stops at the first bad index
    assert b.visible_gpu_count({"HIP_VISIBLE_DEVICES": ""}, nodes) == 0
    assert b.visible_gpu_count({}, str(tmp_path / "absent")) is None               # not a ROCm box: unknown


_PARENT_PROBE = """
import json, sys
sys.path.insert(0, {root!r})
import bench
seen = {{}}
def fake_call(cmd, env):
    seen.update(torch_imported='torch' in sys.modules, handles=bench.gpu_handles_open(), cmd=cmd,
                count=bench.visible_gpu_count())
    return 0
rc = bench.spawn_ranks(2, ['--steps', '2'], call=fake_call)
print(json.dumps(dict(seen, rc=rc)))
"""


def _probe_parent(env):
    res = subprocess.run([sys.executable, "-c", _PARENT_PROBE.format(root=ROOT)], env=env, capture_output=True,
                         text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_spawning_parent_stays_clear_of_torch_and_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SF_BENCH_BACKEND"] = "gloo"
    seen = _probe_parent(env)
    assert seen["rc"] == 0 and seen["torch_imported"] is False and seen["handles"] == []
    assert seen["cmd"][1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--gpus" in seen["cmd"]


def test_config4_shards_tile_the_batch():
    import __graft_entry__ as ge
    shard = ge.load_package().shard
    r = shard.all_ranges(TOTAL, WORLD)
    assert r[0][0] == 0 and r[-1][1] == TOTAL
    assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert {hi - lo for lo, hi in r} == {1_250_000}
    assert shard.element_range(TOTAL, WORLD, 7) == (8_750_000, 10_000_000)


# ---------------------------------------------------------------- GPU ---------------------------------------

def _oracle_window(oracle, first_elmt, n):
    b = oracle.fill_basis(NM, NQ)
    x = oracle.fill_random(n * NMT, SEED, first_elmt * NMT)
    return oracle.bwdtrans_hex((NQ,) * 3, n, b, b, b, x)


@pytest.mark.gpu
def test_rank7_of_8_shard_against_oracle(sf, oracle):
    """What rank 7 of the 8-GPU job computes: elements [8 750 000, 10 000 000), generated from the global
    counter; head / interior / tail windows element-wise against the oracle."""
    lo, hi = sf.shard.element_range(TOTAL, WORLD, 7)
    n = hi - lo
    b = sf.fill_basis(NM, NQ)
    x = sf.fill_random(n * NMT, SEED, lo * NMT)
    out = sf.bwdtrans_hex((NQ,) * 3, b, b, b, x)
    win = 257
    for off in (0, 1, 524_287, 1_048_575, n - win):
        ref = _oracle_window(oracle, lo + off, win)
        got = out[off * NQT:(off + win) * NQT].cpu().numpy()
        assert oracle.rel_err(got, ref) <= 1e-12, off
    # the shard's input really is the slice of the global array (first values of element `lo`)
    assert np.array_equal(x[:NMT].cpu().numpy(), oracle.fill_random(NMT, SEED, lo * NMT))


@pytest.mark.gpu
def test_full_10m_batch_on_one_gpu_equals_the_eight_shards(sf, oracle):
    """Strong-scaling identity at full size: one GPU over all 10 M elements (64-bit indexing: 5.12e9 output
    doubles) vs the eight shards run one after another; bit-identical union, equal checksum."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 85e9:
        pytest.skip("needs ~80 GB of free HBM")
    b = sf.fill_basis(NM, NQ)
    x = sf.fill_random(TOTAL * NMT, SEED, 0)
    full = sf.bwdtrans_hex((NQ,) * 3, b, b, b, x)
    full_ss = sf.sumsq(full)
    shard_ss = 0.0
    for r in range(WORLD):
        lo, hi = sf.shard.element_range(TOTAL, WORLD, r)
        xs = sf.fill_random((hi - lo) * NMT, SEED, lo * NMT)
        assert torch.equal(xs, x[lo * NMT:hi * NMT]), r
        outs = sf.bwdtrans_hex((NQ,) * 3, b, b, b, xs)
        assert torch.equal(outs, full[lo * NQT:hi * NQT]), r       # same kernel, same per-element arithmetic
        shard_ss += sf.sumsq(outs)
        del xs, outs
    assert abs(shard_ss - full_ss) <= 1e-12 * full_ss
    # and the single-GPU result is the oracle's at the shard seams and beyond 2^32 output doubles
    for first in (0, 1_249_999, 8_388_607, 8_750_000 - 3, TOTAL - 129):
        ref = _oracle_window(oracle, first, 129)
        got = full[first * NQT:(first + 129) * NQT].cpu().numpy()
        assert oracle.rel_err(got, ref) <= 1e-12, first


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus(sf):
    """`python bench.py --gpus 2` without a launcher on a 1-GPU box: fails loudly, prints no JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has 2+ GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SF_BENCH_BACKEND")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode != 0 and "1 GPU" in res.stderr and "{" not in res.stdout


@pytest.mark.gpu
def test_spawning_parent_holds_no_gpu_device_file(sf):
    """On the GPU box: at the moment bench.py starts the child launcher the parent has not imported torch, holds
    neither /dev/kfd nor a /dev/dri node, and its sysfs count equals what the HIP runtime reports to this (other,
    initialised) process."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SF_BENCH_BACKEND"] = "gloo"
    seen = _probe_parent(env)
    assert seen["torch_imported"] is False and seen["handles"] == [], seen
    assert seen["count"] == torch.cuda.device_count(), seen
    b = _bench()
    assert b.gpu_handles_open(), "this test process HAS initialised the GPU: the probe must see its device files"


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks_config4(sf):
    """`python bench.py --gpus 2` with no launcher starts 2 ranks itself (gloo rehearsal: the ranks share this
    box's GPU), defaults to the 10 M-element strong-scaling batch and reports the ranks actually reduced over."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SF_BENCH_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong"
    assert rec["config"]["total_elements"] == TOTAL and rec["config"]["elements_per_gpu"] == TOTAL // 2
    assert len(rec["roofline"]["per_gpu_frac"]) == 2 and "rehearsal" in rec
    assert rec["single_gpu_same_batch_gdof_s"] > 50 and rec["speedup_vs_1gpu_same_batch"] > 0
    # the N > 1 line is complete: CPU baseline in the same run, traffic of the shard shape, per-GPU fractions
    assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["cores"] >= 1
    assert "traffic" in rec["roofline"] and "frac_kernel_events" in rec["roofline"]
    assert len(rec["roofline"]["per_gpu_wall_ms_per_step"]) == 2
    # checksum of the sharded run == one rank over the whole batch
    b = sf.fill_basis(NM, NQ)
    x = sf.fill_random(TOTAL * NMT, SEED, 0)
    full = math.sqrt(sf.sumsq(sf.bwdtrans_hex((NQ,) * 3, b, b, b, x)))
    assert abs(rec["checksum_norm"] - full) <= 1e-12 * full


def test_dispatches_per_call_mirror_the_kernel_table():
    """shard.hex_dispatches_per_call() (bench.py's `dispatches_per_step`, the PMC summaries) restates
    csrc/wave_table.h hex_piece(): keep the two in step."""
    import re
    import __graft_entry__ as ge
    shard = ge.load_package().shard
    text = open(os.path.join(ROOT, "gpu-benchmarking_amd", "csrc", "wave_table.h")).read()
    m = re.search(r"hex_piece\(int nq\)\s*\{\s*return \(([^)]*)\) \? \(1ull << (\d+)\) : 0;", text)
    assert m, "hex_piece() changed shape: update shard.HEX_PIECE and this test"
    orders = sorted(int(x) for x in re.findall(r"nq == (\d+)", m.group(1)))
    assert shard.HEX_PIECE == {q: 1 << int(m.group(2)) for q in orders}
    assert shard.hex_dispatches_per_call(8, 1 << 20) == 1            # the headline batch stays one dispatch
    assert shard.hex_dispatches_per_call(8, 1_250_000) == 3          # config 4: one GPU's shard of the 8-GPU job
    assert shard.hex_dispatches_per_call(8, 10_000_000) == 20
    assert shard.hex_dispatches_per_call(10, 10_000_000) == 1
