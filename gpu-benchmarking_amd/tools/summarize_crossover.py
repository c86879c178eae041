#!/usr/bin/env python3
"""Summarise tools/profile_crossover.sh output into a table (markdown + JSON).

Per configuration: kernel time (kernel-trace average of the BwdTrans kernel), HBM bytes
(FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, KB -> B), fp64 flops issued on the vector ALUs
(64 lanes x 2 x (FMA + 0.5 MUL) wave-instructions) and on the matrix cores (512 flop per
SQ_INSTS_VALU_MFMA_MOPS_F64), each as a fraction of the MI355X peaks (8 TB/s, 78.6 TFLOP/s fp64
vector = matrix), plus the busy counters."""
import csv
import glob
import json
import os
import sys

HBM_PEAK, FP64_PEAK = 8.0e12, 78.6e12
KERNELS = ("wave_kernel", "wave3_kernel", "wave_rt_kernel", "mfma_kernel", "mfma4_kernel", "stream_kernel")


def kernel_rows(path, key="wave_kernel"):
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f))
                 if any(k in r["Kernel_Name"] for k in KERNELS)]
    return rows


def counter_means(path):
    acc = {}
    for r in kernel_rows(path):
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def kernel_time_ns(path):
    for f in glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Name"] for k in KERNELS):
                return float(r["AverageNs"]), int(r["Calls"]), r["Name"].split("(")[0]
    return None, 0, ""


def main(root):
    table = []
    for tag in sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))):
        base = os.path.join(root, tag)
        ns, calls, name = kernel_time_ns(os.path.join(base, "kt"))
        if ns is None:
            continue
        sq = counter_means(os.path.join(base, "sq"))
        fetch = counter_means(os.path.join(base, "fetch")).get("FETCH_SIZE", 0.0)
        write = counter_means(os.path.join(base, "write")).get("WRITE_SIZE", 0.0)
        hbm = (2.0 * fetch + write) * 1024.0
        valu_flops = 64 * (2.0 * sq.get("SQ_INSTS_VALU_FMA_F64", 0) + sq.get("SQ_INSTS_VALU_MUL_F64", 0))
        mfma_flops = 512.0 * sq.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)
        t = ns * 1e-9
        gui = sq.get("GRBM_GUI_ACTIVE", 0) / 8.0        # summed over the 8 XCDs
        simd_cycles = gui * 256 * 4
        table.append({
            "config": tag, "kernel": name, "calls": calls, "kernel_ms": ns * 1e-6,
            "hbm_gb": hbm * 1e-9, "hbm_frac_of_8TBs": hbm / t / HBM_PEAK,
            "valu_tflops": valu_flops / t * 1e-12, "mfma_tflops": mfma_flops / t * 1e-12,
            "fp64_frac_of_78.6TF": (valu_flops + mfma_flops) / t / FP64_PEAK,
            "valu_busy": 4.0 * sq.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles if simd_cycles else None,
            "mfma_busy": sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles if simd_cycles else None,
            "lds_busy": 4.0 * sq.get("SQ_ACTIVE_INST_LDS", 0) / simd_cycles if simd_cycles else None,
            "clock_ghz": gui / t * 1e-9 if t else None,
        })
    json.dump(table, open(os.path.join(root, "crossover.json"), "w"), indent=1)
    hdr = ("| config | kernel ms | HBM GB | HBM frac of 8 TB/s | VALU TF/s | MFMA TF/s | fp64 frac of 78.6 TF | "
           "VALU busy | MFMA busy | LDS busy | clock GHz |")
    lines = [hdr, "|" + "---|" * 11]
    for r in table:
        f = lambda x, n=3: "-" if x is None else f"{x:.{n}f}"
        lines.append(f"| {r['config']} | {f(r['kernel_ms'], 4)} | {f(r['hbm_gb'])} | {f(r['hbm_frac_of_8TBs'])} | "
                     f"{f(r['valu_tflops'], 2)} | {f(r['mfma_tflops'], 2)} | {f(r['fp64_frac_of_78.6TF'])} | "
                     f"{f(r['valu_busy'])} | {f(r['mfma_busy'])} | {f(r['lds_busy'])} | {f(r['clock_ghz'], 2)} |")
    open(os.path.join(root, "crossover.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/crossover")
