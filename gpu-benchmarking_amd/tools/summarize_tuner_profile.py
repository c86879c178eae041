#!/usr/bin/env python3
"""tools/profile_tuner.sh output -> one row per kernel: time, clock, pipe-busy and wait fractions.
usage: summarize_tuner_profile.py OUTDIR [name-filter]"""
import csv
import glob
import os
import re
import sys


def counters(path):
    acc = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main(root, flt=""):
    times = {}
    for f in glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            times[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
    c1, c2 = counters(os.path.join(root, "p1")), counters(os.path.join(root, "p2"))
    print("| kernel | calls | ms | clock GHz | MFMA busy | MFMA TF/s | VALU busy | LDS busy | LDS conflict | wave-cycles: "
          "issuing / waiting any / issue-stalled / waiting LDS | VALU inst per wave |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, (ns, calls) in sorted(times.items(), key=lambda kv: kv[0]):
        if flt not in name or name not in c1:
            continue
        a, b = c1[name], c2.get(name, {})
        gui = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
        simd = gui * 256 * 4
        wc = a.get("SQ_WAVE_CYCLES", 0) or 1
        short = re.sub(r"^void sf::", "", name).split("(")[0]
        t = ns * 1e-9
        print(f"| {short} | {calls} | {ns * 1e-6:.3f} | {gui / t * 1e-9:.2f} | {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd:.2f} | "
              f"{512 * a.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0) / t * 1e-12:.1f} | {4 * a.get('SQ_ACTIVE_INST_VALU', 0) / simd:.2f} | "
              f"{4 * a.get('SQ_ACTIVE_INST_LDS', 0) / simd:.2f} | "
              f"{b.get('SQ_LDS_BANK_CONFLICT', 0) / (b.get('SQ_LDS_IDX_ACTIVE', 0) or 1):.2f} | "
              f"{b.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} / {b.get('SQ_WAIT_ANY', 0) / wc:.2f} / "
              f"{b.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} / {b.get('SQ_WAIT_INST_LDS', 0) / wc:.2f} | "
              f"{a.get('SQ_INSTS_VALU', 0) / (a.get('SQ_WAVE_CYCLES', 1) and 1):.0f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
