#!/usr/bin/env python3
"""Is a kernel held back by the chip's power management?  The same launch on all-zero inputs (operand buses do not
toggle, the chip holds a higher clock) against seeded random inputs: a memory-bound kernel does not move, a kernel at the
chip's sustained matrix / vector rate does (MI355X_MICROARCH.md, DVFS give-back).  Usage: zero_data_clock.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import __graft_entry__ as ge  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sum(ts) / len(ts)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    sf = ge.load_package()
    n = 1 << 20
    for dim, nq in ((2, 16), (2, 20), (2, 24), (2, 26), (2, 28), (2, 30), (2, 32), (3, 8), (3, 10)):
        nm = nq - 1
        b = sf.fill_basis(nm, nq)
        fn = sf.bwdtrans_hex if dim == 3 else sf.bwdtrans_quad
        bs = [b] * dim
        out = torch.empty(n * nq ** dim, dtype=torch.float64, device="cuda")
        byt = n * 8 * (nm ** dim + nq ** dim)
        row = f"{dim}D nq{nq:<3d}"
        for label, x in (("random", sf.fill_random(n * nm ** dim, 1)), ("zeros", torch.zeros(n * nm ** dim, dtype=torch.float64, device="cuda"))):
            ms = timed(lambda: fn((nq,) * dim, *bs, x, out=out), reps)
            row += f" | {label}: {ms:7.4f} ms frac {byt / ms * 1e-6 / 8000:.3f}"
        # zero basis too: the matrix pipe multiplies zeros by zeros
        zb = torch.zeros_like(b)
        x = torch.zeros(n * nm ** dim, dtype=torch.float64, device="cuda")
        ms = timed(lambda: fn((nq,) * dim, *([zb] * dim), x, out=out), reps)
        row += f" | zeros x zero basis: {ms:7.4f} ms frac {byt / ms * 1e-6 / 8000:.3f}"
        print(row, flush=True)


if __name__ == "__main__":
    main()
