#!/usr/bin/env python3
"""Plot benchmark logs: one PNG per *.log (semilog-x, value vs number of elements / size).

Counterpart of the reference's per-benchmark plotting scripts (benchmark05/postprocess.py:4-27,
benchmark01/postprocess.py:4-22): same inputs (every ./*.log in the working directory, or the paths
given), same line filters and column extraction (gpu_benchmarking_amd.logfmt), same output naming
(<log stem>.png).  Additions: legends come from a label table for THIS build's columns (falling back
to the reference's labels for the reference's own 11-/5-column logs), and an optional HBM-roofline
overlay for BwdTrans logs (--roofline).

    python postprocess.py [--roofline] [logs ...]
"""
import argparse
import glob
import os
import re
import sys

try:
    from . import logfmt
except ImportError:  # run as a script
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import logfmt

OUR_LABELS = {
    ("DOF/s", 5): ["HIP (thread/elmt)", "HIP (block/elmt glb)", "HIP (block/elmt LDS)",
                   "HIP (wave/chunk)", "rocBLAS"],
    ("DOF/s", 6): ["HIP (thread/elmt)", "HIP (block/elmt glb)", "HIP (block/elmt LDS)",
                   "HIP (wave/chunk)", "rocBLAS", "HIP (thread/elmt il64)"],
    ("GB/s", 2): ["Host (OpenMP)", "HIP (vl)"],
}
REFERENCE_LABELS = {
    ("DOF/s", 11): ["Kokkos (Uncoales)", "Kokkos (Coales)", "Kokkos (QP)", "Kokkos (QP/Shared)",
                    "cuBLAS", "Cuda (Uncoales)", "Cuda (Coales)", "Cuda (QP)", "Cuda (QP/Shared)",
                    "Cuda (QP-1D)", "Cuda (QP-1D/Shared)"],
    ("GB/s", 5): ["Kokkos", "Thrust", "CUDA", "CUDA (vl)", "CUDA (functor)"],
}
HBM_PEAK = 8.0e12


def labels_for(log):
    key = (log.kind, log.ncols)
    return OUR_LABELS.get(key) or REFERENCE_LABELS.get(key) or [f"col {i+1}" for i in range(log.ncols)]


def roofline_gdofs(title):
    """HBM-roofline DOF rate for the 'BwdTrans (NQ = a, b[, c])' title: 8*(nm^d + nq^d) B/element."""
    m = re.search(r"NQ = ([\d, ]+)", title)
    if not m:
        return None
    nq = [int(t) for t in m.group(1).replace(" ", "").split(",") if t]
    nmt = nqt = 1
    for q in nq:
        nmt *= q - 1
        nqt *= q
    return HBM_PEAK / (8.0 * (nmt + nqt)) * nmt * 1e-9


def plot(path, roofline=False):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    log = logfmt.parse_file(path)
    labels = labels_for(log)
    plt.figure()
    for c in range(log.ncols):
        plt.semilogx(log.sizes, log.column(c), label=labels[c])
    if roofline and log.kind == "DOF/s":
        roof = roofline_gdofs(log.title)
        if roof:
            plt.axhline(roof, color="k", linestyle=":", label="HBM roofline (8 TB/s)")
    plt.legend()
    plt.xlabel("Number of elmt." if log.kind == "DOF/s" else "Size")
    plt.ylabel("DOF (1e9/s)" if log.kind == "DOF/s" else "GB/s")
    if log.title:
        plt.title(log.title)
    out = path.split(".log")[0] + ".png"
    plt.savefig(out)
    plt.close()
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--roofline", action="store_true")
    ap.add_argument("logs", nargs="*")
    args = ap.parse_args(argv)
    for path in (args.logs or sorted(glob.glob("./*.log"))):
        print(plot(path, args.roofline))


if __name__ == "__main__":
    main()
