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


KERNEL_SOURCES = ("bwdtrans_wave.h", "wave_table.h", "bwdtrans_hex.hip", "bwdtrans_quad.hip",
                  "bwdtrans_mfma.h", "bwdtrans_mfma4.h", "sf_common.h", "wave_launch.h",
                  "bwdtrans_wave3.h", "bwdtrans_rt.h", "bwdtrans_rt.hip", "bwdtrans_hmfma4.h")


def kernel_source_hash(root):
    """sha256 (16 hex digits) over the CODE of the kernel sources a PMC record describes (comments and white space
    do not count); a record whose hash differs from the tree's was taken on other code and is reported as stale."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(root, "gpu-benchmarking_amd", "csrc", name), "r") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


HEX_PIECE = {7: 1 << 19, 8: 1 << 19}   # csrc/wave_table.h hex_piece(): elements per dispatch of a large fp64 batch


def hex_dispatches_per_call(nq, nelmt):
    """Kernel dispatches one sf_bwdtrans_hex_f64 call enqueues (csrc/bwdtrans_hex.hip go<NQ>(): a batch above two
    pieces is enqueued piece by piece)."""
    piece = HEX_PIECE.get(nq, 0)
    return -(-nelmt // piece) if piece and nelmt > 2 * piece else 1


def recorded_traffic(root, dim, nq, nelmt):
    """HBM bytes per launch from the PMC passes committed under profiles/ (None if not recorded for
    this shape).  bench.py cannot collect PMC counters itself (gpurun keeps --pmc runs separate from
    every other trace); see profiles/README.md.  Returns {"bytes", "over_algorithmic", "source"}."""
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None
    for row in rec.get("rows", []):
        if row.get("nq") == nq and row.get("nelmt") == nelmt and row.get("dim", 3) == dim:
            stale = row.get("kernel_source_hash") != kernel_source_hash(root)
            return {"bytes": row.get("hbm_bytes_per_launch"),
                    "over_algorithmic": row.get("traffic_over_algorithmic"),
                    "source": "recorded: profiles/hbm_traffic.json, rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + "
                              f"WRITE_SIZE in separate passes, round {row.get('round', rec.get('round'))}"
                              + (" -- STALE: the kernel sources changed since that pass" if stale else
                                 ", same kernel sources as this tree")}
    return None
