#!/usr/bin/env python3
"""What the low 3D orders can reach at 1 Mi elements: their launches last 12-50 us and move 75-300 MB, so the
fixed cost per launch (dispatch ramp, the gap between dependent kernels in a replayed HIP graph) is a visible share.
Times arithmetic-free streams of the SAME byte counts (sf_stream_copy_f64: 16-byte lanes, one pass) under the
protocol of bench.py's order sweep (groups of 8 launches replayed from a HIP graph, min over 40 groups) next to
the BwdTrans launches themselves.  Usage: python3 tools/small_batch_ceiling.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

sf = ge.load_package()
dev = torch.device("cuda:0")


def best_ms(fn, inner=8, reps=40):
    fn()
    torch.cuda.synchronize()
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
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / inner)
    return best


nelmt = 1 << 20
print(f"{'case':44s} {'bytes/launch':>14s} {'us':>9s} {'GB/s':>9s} {'frac of 8 TB/s':>15s}")
for nq in (2, 3, 4, 5, 8):
    nm = nq - 1
    b = sf.fill_basis(nm, nq, dev)
    x = sf.fill_random(nelmt * nm ** 3, 1, 0, dev)
    o = torch.empty(nelmt * nq ** 3, dtype=torch.float64, device=dev)
    total = 8 * nelmt * (nm ** 3 + nq ** 3)
    ms = best_ms(lambda: sf.bwdtrans_hex((nq,) * 3, b, b, b, x, out=o))
    print(f"{'hex nq=%d BwdTrans' % nq:44s} {total:14d} {ms * 1e3:9.2f} {total / ms * 1e-6:9.1f} {total / ms * 1e-6 / 8000:15.4f}")
    # copy with the same total traffic (half read, half written)
    n = total // 16 // 2 * 2
    src = sf.fill_random(n, 2, 0, dev)
    dst = torch.empty_like(src)
    ms = best_ms(lambda: sf.stream_copy(src, dst))
    print(f"{'  copy, same bytes (50 % reads)':44s} {16 * n:14d} {ms * 1e3:9.2f} {16 * n / ms * 1e-6:9.1f} {16 * n / ms * 1e-6 / 8000:15.4f}")
    # x += y with the same total traffic (2/3 reads)
    n3 = total // 24 // 2 * 2
    xa, ya = sf.fill_vecadd(n3, dev)
    ms = best_ms(lambda: sf.vector_add(xa, ya))
    print(f"{'  x += y, same bytes (67 % reads)':44s} {24 * n3:14d} {ms * 1e3:9.2f} {24 * n3 / ms * 1e-6:9.1f} {24 * n3 / ms * 1e-6 / 8000:15.4f}")
    del x, o, src, dst, xa, ya
