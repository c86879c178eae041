#!/usr/bin/env python3
"""Throughput of SF_VARIANT_AUTO over every isotropic order the library builds (development tool):
2D nq = 2..32 and 3D nq = 2..16, min / mean over reps of HIP-event-timed launches, with the fraction of the
8 TB/s HBM roofline.  Usage: sweep_auto.py [nelmt] [reps] [f64|f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge  # noqa: E402


def main():
    nelmt = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
    sf = ge.load_package()
    esz = 4 if dt == torch.float32 else 8
    for dim, orders in ((2, range(2, 33)), (3, range(2, 17))):
        for nq in orders:
            # fp32 wave rows: 2D every order 2..32, 3D 2..16 (fp64 uses the matrix cores where fp32 still fits the registers)
            nm = nq - 1
            n = nelmt if (dim == 2 or nq <= 10) else nelmt // 8
            bs = [sf.fill_basis(nm, nq, dtype=dt) for _ in range(dim)]
            x = sf.fill_random(n * nm ** dim, 1, dtype=dt)
            out = torch.empty(n * nq ** dim, dtype=dt, device="cuda")
            fn = sf.bwdtrans_hex if dim == 3 else sf.bwdtrans_quad
            fn((nq,) * dim, *bs, x, out=out)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn((nq,) * dim, *bs, x, out=out)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            tmin, tmean = min(ts), sum(ts) / len(ts)
            dof = n * nm ** dim
            byt = n * esz * (nm ** dim + nq ** dim)
            print(f"{dim}D nq{nq:<3d} nelmt {n:>8d}  {dof / tmin * 1e-6:8.2f} / {dof / tmean * 1e-6:8.2f} GDOF/s"
                  f"  {byt / tmean * 1e-6:8.1f} GB/s  frac {byt / tmean * 1e-6 / 8000:.3f}", flush=True)
            del x, out


if __name__ == "__main__":
    main()
