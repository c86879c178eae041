#!/usr/bin/env python3
"""bench.py -- headline benchmark: 3D hex BwdTrans sum-factorisation, fp64, nq = 8, on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (one sf_bwdtrans_hex_f64 launch) over this rank's batch of
elements, inputs resident in HBM.  The element batch is embarrassingly parallel: ranks own disjoint
element ranges, there is NO data-path collective; torch.distributed (RCCL) carries only the barrier,
the MAX of the elapsed time and the SUM of the result checksum.

Metric (BASELINE.json): GDOF/s = 1e-9 * nelmt * nm^3 / t   (benchmark05/benchmark05.cc:1408; DOF =
input modes).  Roofline: HBM, algorithmic bytes 8*(nm^3 + nq^3) per element.

Rank 0 prints ONE JSON line.  Extra keys: "roofline", "cpu_baseline" (oracle port timed on this
box's host cores, N=1 only) and "extra" (nq sweep 2..10 and the 2D quad nq=8 config, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X datasheet HBM3E peak (/opt/skills/guides/MI355X_MICROARCH.md)
NQ = 8
GOLDEN_NORM_1M = 17134.76235  # benchmark05/nq8x8x8.log:45 (nelmt 1 048 576, sin/cos data)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--elements-per-gpu", type=int, default=1 << 20,
                    help="weak scaling: elements owned by each rank (default 1 048 576)")
    ap.add_argument("--total-elements", type=int, default=0,
                    help="strong scaling: shard this many elements over the ranks "
                         "(e.g. 10000000 for BASELINE config 4)")
    ap.add_argument("--nq", type=int, default=NQ)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def time_steps(fn, steps, warmup, torch, dist, world):
    """W untimed steps, then exactly K timed steps bracketed by barrier + synchronize.
    Returns (wall seconds of this rank, HIP-event seconds on the launch stream)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return t1 - t0, ev0.elapsed_time(ev1) * 1e-3


def cpu_baseline(nq, seconds):
    """Oracle (-O3/AVX2/FMA build, OpenMP over elements) on a bounded sample of the same workload."""
    import oracle
    nm = nq - 1
    sample = 262144
    cores = oracle.usable_cpus()      # not omp_get_max_threads(): the box grants a CPU share
    for fast in (False, True):
        oracle.set_threads(cores, fast=fast)
    b = oracle.fill_basis(nm, nq)
    x = oracle.fill_random(sample * nm ** 3, 0x5F3759DF)
    best, spent, reps = float("inf"), 0.0, 0
    while reps < 3 or (spent < seconds and reps < 200):
        t0 = time.perf_counter()
        oracle.bwdtrans_hex((nq,) * 3, sample, b, b, b, x, form="vector", fast=True)
        dt = time.perf_counter() - t0
        best = min(best, dt)
        spent += dt
        reps += 1
    return {"value": round(1e-9 * sample * nm ** 3 / best, 4), "unit": "GDOF/s",
            "cores": oracle.max_threads(fast=True), "kind": "port",
            "sample": f"hex nq={nq}, {sample} elements, seeded random data, min of {reps} passes "
                      f"({spent:.1f} s of CPU work), oracle/bwdtrans_ref.c 3-sweep form in CPU loop order (oracle_bwdtrans_hex_vector), "
                      f"-O3 -mavx2 -mfma, OpenMP"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    # one process per GPU; SF_BENCH_BACKEND=gloo (rehearsal: several ranks may then share a GPU,
    # the scalar reductions go through CPU tensors) -- default nccl = RCCL over xGMI
    backend = os.environ.get("SF_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rdev = dev if backend == "nccl" else torch.device("cpu")   # where the reduced scalars live
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    sf = ge.load_package()
    from gpu_benchmarking_amd import shard

    nq, nm = args.nq, args.nq - 1
    if args.total_elements > 0:
        lo, hi = shard.element_range(args.total_elements, world, rank)
        scaling, total = "strong", args.total_elements
    else:
        lo, hi = rank * args.elements_per_gpu, (rank + 1) * args.elements_per_gpu
        scaling, total = "weak", world * args.elements_per_gpu
    nelmt = hi - lo

    b = sf.fill_basis(nm, nq, dev)
    # seeded, per-value-distinct data; the global element index seeds the stream so every rank
    # generates exactly its shard of the one global array
    x = sf.fill_random(nelmt * nm ** 3, 0x5F3759DF, lo * nm ** 3, dev)
    out = torch.empty(nelmt * nq ** 3, dtype=torch.float64, device=dev)

    def step():
        sf.bwdtrans_hex((nq,) * 3, b, b, b, x, out=out)

    wall, evs = time_steps(step, args.steps, args.warmup, torch, dist, world)
    tmax = torch.tensor([wall, evs], dtype=torch.float64, device=rdev)
    checksum = torch.tensor([sf.sumsq(out)], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(checksum, op=dist.ReduceOp.SUM)
    wall_max, ev_max = float(tmax[0]), float(tmax[1])

    result = None
    if rank == 0:
        dof = total * nm ** 3
        bytes_per_elmt = 8 * (nm ** 3 + nq ** 3)
        kernel_s = ev_max / args.steps                 # average launch duration (HIP events)
        achieved = nelmt * bytes_per_elmt / kernel_s * 1e-9   # GB/s of ONE GPU's launch
        result = {
            "metric": "GDOF/s for 3D hex sum-factorisation, fp64, nq=2..10 sweep",
            "value": round(1e-9 * dof * args.steps / wall_max, 3),
            "unit": "GDOF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * wall_max / args.steps, 5),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"benchmark05 3D hex BwdTrans nq={nq}x{nq}x{nq}, "
                                   f"{nelmt} elements per GPU ({total} total), fp64, "
                                   f"seeded random modes, cos basis",
                       "nq": nq, "elements_per_gpu": nelmt, "total_elements": total,
                       "parallelism": f"element-range sharding x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": shard.recorded_traffic(ROOT, nq, nelmt),
                         "kernel": "sf::hex_wave_kernel", "bytes_per_element": bytes_per_elmt,
                         "kernel_ms": round(kernel_s * 1e3, 5)},
            "checksum_norm": math.sqrt(float(checksum[0])),
        }

    if rank == 0 and world == 1:
        # parity sanity on the reference's own data: golden norm of benchmark05/nq8x8x8.log:45
        if nq == 8:
            xs = sf.fill_sincos(1 << 20, nm ** 3, dev)
            os_ = sf.bwdtrans_hex((nq,) * 3, b, b, b, xs)
            norm = math.sqrt(sf.sumsq(os_))
            result["golden_norm_check"] = {"got": norm, "reference": GOLDEN_NORM_1M,
                                           "ok": abs(norm - GOLDEN_NORM_1M) <= 5.5e-10 * norm}
            del xs, os_
        if not args.no_extra:
            result["extra"] = extras(sf, torch, dev)
            # second denominator: the device's own measured stream rate (benchmark02's x += y,
            # 24 B/element), next to the 8 TB/s datasheet figure
            stream = measured_stream_gbs(sf, torch, dev)
            result["roofline"]["measured_stream_gb_s"] = round(stream, 1)
            result["roofline"]["frac_of_measured_stream"] = round(
                result["roofline"]["achieved"] / stream, 4)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(nq, args.cpu_seconds)
            result["vs_reference_published"] = {
                "value": round(result["value"] / 26.389, 2),
                "note": "reference's best variant, 26.389 GDOF/s on an unstated NVIDIA GPU "
                        "(benchmark05/nq8x8x8.log:46); not an MI355X number, so vs_baseline is null"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_stream_gbs(sf, torch, dev, n=1 << 28, reps=10):
    """benchmark02 on this device: x += y over 2 x 2 GiB, best of `reps` (HIP events)."""
    x, y = sf.fill_vecadd(n, dev)
    sf.vector_add(x, y)
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sf.vector_add(x, y)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return 24.0 * n / best * 1e-6


def extras(sf, torch, dev, nelmt=1 << 20, reps=10):
    """BASELINE configs 1 and 3 on one GPU: quad nq=8 and the hex nq = 2..10 sweep (min of reps,
    HIP events), each with its fraction of the 8 TB/s HBM roofline."""
    def best_ms(fn, inner=8):
        # `inner` back-to-back launches per event pair, replayed from a HIP graph when capture works: the
        # low orders run for ~10 us, where eager launches mostly measure the host's launch cadence
        fn()
        torch.cuda.synchronize()
        graph = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(inner):
                        fn()
            torch.cuda.current_stream().wait_stream(side)
            g.replay()
            torch.cuda.synchronize()
            graph = g
        except Exception:                      # capture unsupported here: eager launches
            torch.cuda.synchronize()
        best = float("inf")
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if graph is not None:
                graph.replay()
            else:
                for _ in range(inner):
                    fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / inner)
        used_graph[0] = used_graph[0] and graph is not None
        return best

    used_graph = [True]
    out = {"protocol": f"{nelmt} elements, min over {reps} groups of 8 back-to-back launches (HIP events; "
                       "the groups are HIP-graph replays when stream capture succeeds)",
           "hex_sweep": {}, "quad": {}}
    for nq in range(2, 11):
        nm = nq - 1
        b = sf.fill_basis(nm, nq, dev)
        x = sf.fill_random(nelmt * nm ** 3, 1, 0, dev)
        o = torch.empty(nelmt * nq ** 3, dtype=torch.float64, device=dev)
        ms = best_ms(lambda: sf.bwdtrans_hex((nq,) * 3, b, b, b, x, out=o))
        gbs = nelmt * 8 * (nm ** 3 + nq ** 3) / ms * 1e-6
        out["hex_sweep"][str(nq)] = {"gdof_s": round(nelmt * nm ** 3 / ms * 1e-6, 2),
                                     "gb_s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        del x, o
    for nq in (8, 16, 20, 24, 32):  # 20 / 24: vector-ALU kernel past the crossover; 16 / 32: matrix cores
        nm = nq - 1
        b = sf.fill_basis(nm, nq, dev)
        x = sf.fill_random(nelmt * nm ** 2, 1, 0, dev)
        o = torch.empty(nelmt * nq ** 2, dtype=torch.float64, device=dev)
        ms = best_ms(lambda: sf.bwdtrans_quad((nq, nq), b, b, x, out=o))
        gbs = nelmt * 8 * (nm ** 2 + nq ** 2) / ms * 1e-6
        out["quad"][str(nq)] = {"gdof_s": round(nelmt * nm ** 2 / ms * 1e-6, 2),
                                "gb_s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    out["hip_graph_replay"] = used_graph[0]
    return out


if __name__ == "__main__":
    main()
