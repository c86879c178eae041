#!/usr/bin/env python3
"""Anisotropic 3D extents (nq0 != nq1 != nq2): SF_VARIANT_AUTO (the compile-time triple of bwdtrans_wave3.h where the shape
is in the table of bwdtrans_rt.hip, else the run-time-extent wave kernel of bwdtrans_rt.h) against that run-time kernel and
the barrier-per-sweep generic kernel, mean / min over reps of HIP-event-timed launches, fraction of the 8 TB/s HBM
roofline (algorithmic bytes 8 (nm0 nm1 nm2 + nq0 nq1 nq2) per element).  Usage: aniso_bench.py [nelmt] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge  # noqa: E402

SHAPES = [(8, 8, 4), (4, 8, 6), (10, 6, 8), (8, 8, 8), (6, 6, 12), (12, 10, 8), (16, 12, 14), (3, 5, 4), (16, 16, 16)]


def main():
    nelmt = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    sf = ge.load_package()
    for nq in SHAPES:
        nm = [q - 1 for q in nq]
        nmt, nqt = nm[0] * nm[1] * nm[2], nq[0] * nq[1] * nq[2]
        n = nelmt if nqt <= 1024 else nelmt // 8
        bs = [sf.fill_basis(nm[d], nq[d]) for d in range(3)]
        x = sf.fill_random(n * nmt, 1)
        out = torch.empty(n * nqt, dtype=torch.float64, device="cuda")
        row = f"nq {nq[0]:>2d}x{nq[1]:>2d}x{nq[2]:>2d} nelmt {n:>8d}"
        for variant in ("auto", "wave-rt", "generic"):
            sf.bwdtrans_hex(nq, *bs, x, out=out, variant=variant)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                sf.bwdtrans_hex(nq, *bs, x, out=out, variant=variant)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            tmean, tmin = sum(ts) / len(ts), min(ts)
            byt = n * 8 * (nmt + nqt)
            row += (f" | {variant:8s} {n * nmt / tmean * 1e-6:7.2f} GDOF/s {byt / tmean * 1e-6:7.1f} GB/s "
                    f"frac {byt / tmean * 1e-6 / 8000:.3f} (min-time {byt / tmin * 1e-6 / 8000:.3f})")
        print(row, flush=True)
        del x, out


if __name__ == "__main__":
    main()
