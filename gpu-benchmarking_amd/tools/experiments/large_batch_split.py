"""Experiment: one launch over a large batch against the same batch as back-to-back sub-launches.

The headline shape (3D nq = 8) holds 0.79-0.80 of the HBM roofline at 1 Mi elements but 0.75 at the 10 M elements of
BASELINE configs[4].  Elements are independent, so a batch can be cut at any element boundary; this script times
the whole batch in one launch and in pieces of several sizes (same stream, no synchronisation between pieces).

    python gpu-benchmarking_amd/tools/large_batch_split.py [--nq 8] [--elements 10000000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, nargs="+", default=[8])
    ap.add_argument("--elements", type=int, nargs="+", default=[1 << 20, 2500000, 5000000, 10000000])
    ap.add_argument("--pieces", type=int, nargs="+", default=[0, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22])
    ap.add_argument("--reps", type=int, default=12)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    sf = ge.load_package()
    dev = torch.device("cuda:0")
    for nq in args.nq:
        sweep(args, torch, sf, dev, nq)


def sweep(args, torch, sf, dev, nq):
    nm = nq - 1
    b = sf.fill_basis(nm, nq, dev)
    for total in args.elements:
        x = sf.fill_random(total * nm ** 3, 0x5F3759DF, 0, dev)
        out = torch.empty(total * nq ** 3, dtype=torch.float64, device=dev)
        bytes_ = 8.0 * total * (nm ** 3 + nq ** 3)
        for piece in args.pieces:
            if piece >= total:
                continue

            def run():
                if piece == 0:
                    sf.bwdtrans_hex((nq,) * 3, b, b, b, x, out=out)
                    return
                for lo in range(0, total, piece):
                    hi = min(total, lo + piece)
                    sf.bwdtrans_hex((nq,) * 3, b, b, b, x[lo * nm ** 3:hi * nm ** 3],
                                    out=out[lo * nq ** 3:hi * nq ** 3])
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            best, tot = float("inf"), 0.0
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                dt = e0.elapsed_time(e1) * 1e-3
                best, tot = min(best, dt), tot + dt
            mean = tot / args.reps
            print(f"nq {nq} elements {total:>9} piece {piece if piece else 'whole':>8}: "
                  f"min {bytes_ / best * 1e-9:7.1f} GB/s ({bytes_ / best / 8e12:.3f})  "
                  f"mean {bytes_ / mean * 1e-9:7.1f} GB/s ({bytes_ / mean / 8e12:.3f})  "
                  f"{1e-9 * total * nm ** 3 / mean:6.1f} GDOF/s", flush=True)
        del x, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
