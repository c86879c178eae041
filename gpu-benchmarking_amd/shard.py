"""Multi-GPU plumbing: the element batch is embarrassingly parallel, so it shards as contiguous
element ranges, one process per GPU.  No element data ever crosses GPUs; the only
collectives are a MAX over ranks of the elapsed time and a SUM of the sum-of-squares checksum
(8-byte messages: pure latency, RCCL/xGMI bandwidth is irrelevant here).  The reference is
single-GPU (CUDA_VISIBLE_DEVICES=1, benchmark05/run.sh:7); this dimension is new.
"""
import json
import os


def element_range(total, world, rank):
    """Contiguous range [lo, hi) of rank `rank`; sizes differ by at most one element."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    return (total * rank) // world, (total * (rank + 1)) // world


def all_ranges(total, world):
    return [element_range(total, world, r) for r in range(world)]


def reduce_time_and_checksum(dist, torch, elapsed_s, sumsq, device):
    """MAX(elapsed) and SUM(sumsq) over ranks; works with any torch.distributed backend."""
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    s = torch.tensor([sumsq], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t[0]), float(s[0])


def aggregate_gdofs(total_elements, nm_tot, steps, max_elapsed_s):
    """Whole-job metric: DOF of ALL ranks / max-over-ranks time (benchmark05.cc:1408 per GPU)."""
    return 1e-9 * total_elements * nm_tot * steps / max_elapsed_s


def recorded_traffic(root, nq, nelmt):
    """HBM bytes per launch from the PMC passes committed under profiles/ (None if not recorded for
    this shape).  bench.py cannot collect PMC counters itself; see profiles/README.md."""
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None
    for row in rec.get("rows", []):
        if row.get("nq") == nq and row.get("nelmt") == nelmt and row.get("dim", 3) == 3:
            return row.get("hbm_bytes_per_launch")
    return None
