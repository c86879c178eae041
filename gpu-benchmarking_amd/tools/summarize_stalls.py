#!/usr/bin/env python3
"""tools/profile_stalls.sh output -> profiles/rNN/stalls/hex_nq8_issue_stall_counters.json + a few derived ratios.
usage: summarize_stalls.py STALLDIR ROUND   (repo root)"""
import csv
import glob
import json
import os
import sys


def means(path):
    acc = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "hex_wave_kernel" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(root, rnd):
    out = {}
    for prec in ("f64", "f32"):
        m = {}
        for n in (1, 2, 3):
            m.update(means(os.path.join(root, f"{prec}_p{n}")))
        out[prec] = dict(sorted(m.items()))
        if m:
            wc = m["SQ_WAVE_CYCLES"]
            elems = 1048576
            out[prec + "_derived"] = {
                "valu_insts_per_element": m["SQ_INSTS_VALU"] / elems,
                "lds_insts_per_element": m["SQ_INSTS_LDS"] / elems,
                "wave_cycles_issuing": m["SQ_ACTIVE_INST_ANY"] / wc,
                "wave_cycles_issuing_valu": m["SQ_ACTIVE_INST_VALU"] / wc,
                "wave_cycles_issuing_lds": m["SQ_ACTIVE_INST_LDS"] / wc,
                "wave_cycles_issuing_scalar": m["SQ_ACTIVE_INST_SCA"] / wc,
                "wave_cycles_waiting_any": m["SQ_WAIT_ANY"] / wc,
                "wave_cycles_issue_stalled": m["SQ_WAIT_INST_ANY"] / wc,
                "lds_bank_conflict_of_lds_active": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"],
                "waves": m["SQ_WAVES"],
            }
    dst = f"profiles/r{int(rnd):02d}/stalls"
    os.makedirs(dst, exist_ok=True)
    json.dump(out, open(os.path.join(dst, "hex_nq8_issue_stall_counters.json"), "w"), indent=1)
    for k in ("f64_derived", "f32_derived"):
        print(k, json.dumps(out.get(k), indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 1)
