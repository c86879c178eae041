#!/usr/bin/env python3
"""fp32 3D nq 12..16: the matrix-core kernel (hex_mfma_kernel with T = float) in several launch shapes against the
vector kernel, selected through the development knob SF_F32_HEX_CFG of bwdtrans_hex.hip (0: vector kernel, 1: one-wave
workgroups MINW 1, 2: MINW 2, 3: two-wave workgroups, 4: two-element chunks).  One process per configuration (the knob
is read once).  Usage: f32_hex_cfg.py [nelmt] [reps]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))

CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
import __graft_entry__ as ge
sf = ge.load_package()
nelmt, reps = int(sys.argv[1]), int(sys.argv[2])
for nq in range(12, 17):
    nm = nq - 1
    b = sf.fill_basis(nm, nq, dtype=torch.float32)
    x = sf.fill_random(nelmt * nm ** 3, 1, dtype=torch.float32)
    out = torch.empty(nelmt * nq ** 3, dtype=torch.float32, device="cuda")
    sf.bwdtrans_hex((nq,) * 3, b, b, b, x, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sf.bwdtrans_hex((nq,) * 3, b, b, b, x, out=out); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sum(ts) / len(ts)
    byt = nelmt * 4 * (nm ** 3 + nq ** 3)
    print(f"  3D nq{nq} nelmt {nelmt}  {nelmt * nm ** 3 / t * 1e-6:8.2f} GDOF/s  {byt / t * 1e-6:8.1f} GB/s  frac {byt / t * 1e-6 / 8000:.3f}", flush=True)
""" % ROOT


def main():
    nelmt = sys.argv[1] if len(sys.argv) > 1 else "131072"
    reps = sys.argv[2] if len(sys.argv) > 2 else "10"
    for cfg in range(5):
        print(f"== SF_F32_HEX_CFG={cfg}", flush=True)
        subprocess.run([sys.executable, "-c", CHILD, nelmt, reps], env=dict(os.environ, SF_F32_HEX_CFG=str(cfg)), check=False)


if __name__ == "__main__":
    main()
