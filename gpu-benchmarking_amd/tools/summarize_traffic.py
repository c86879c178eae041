#!/usr/bin/env python3
"""tools/collect_traffic.sh output -> rows of profiles/hbm_traffic.json (one per shape), each stamped with the hash of
the kernel sources it was measured on (bench.py reports a row taken on other sources as stale).
HBM bytes = FETCH_SIZE x 2 (gfx950 tallies a 128-byte request as 64, /opt/skills/guides/MI355X_MICROARCH.md, HBM
section) + WRITE_SIZE, KB per dispatch, averaged over the BwdTrans dispatches of the run.
usage: summarize_traffic.py TRAFFICDIR ROUND   (repo root)"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

KERNELS = ("wave_kernel", "wave3_kernel", "wave_rt_kernel", "mfma_kernel", "mfma4_kernel", "stream_kernel", "block_kernel")


def upsert(rec, row):
    rows = [r for r in rec.setdefault("rows", [])
            if not (r.get("dim", 3) == row["dim"] and r["nq"] == row["nq"] and r["nelmt"] == row["nelmt"])]
    rows.append(row)
    rec["rows"] = sorted(rows, key=lambda r: (r.get("dim", 3), r["nelmt"], isinstance(r["nq"], list), str(r["nq"]).zfill(12)))


def counter(path, name):
    vals, kern = [], ""
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and any(k in r["Kernel_Name"] for k in KERNELS):
                vals.append(float(r["Counter_Value"]))
                kern = re.sub(r"^void sf::", "", r["Kernel_Name"]).split("(")[0]
    return vals, kern


def main(root, rnd):
    shard = ge.load_package().shard
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    rec = json.load(open(path)) if os.path.exists(path) else {}
    rec["_how"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/collect_traffic.sh over the "
                   "C++ drivers, tools/collect_profiles.sh over bench.py); KB per dispatch of the BwdTrans kernel; "
                   "FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes on gfx950; "
                   "WRITE_SIZE as is.  kernel_source_hash = gpu_benchmarking_amd.shard.kernel_source_hash() at "
                   "collection time.")
    h = shard.kernel_source_hash(ROOT)
    for d in sorted(os.listdir(root)):
        m = re.match(r"(hex|quad)_(\d+(?:x\d+)*)_(\d+)$", d)
        if not m or not os.path.isdir(os.path.join(root, d)):
            continue
        dim, nelmt = (3 if m.group(1) == "hex" else 2), int(m.group(3))
        ext = [int(v) for v in m.group(2).split("x")]          # one order, or anisotropic extents a x b x c
        nq  = ext[0] if len(ext) == 1 else ext
        fetch, kern = counter(os.path.join(root, d, "fetch"), "FETCH_SIZE")
        write, _ = counter(os.path.join(root, d, "write"), "WRITE_SIZE")
        if not fetch or not write:
            print("no counters for", d)
            continue
        # a batch above 2 x 524 288 elements at 3D nq = 7 / 8 is one library call = several dispatches
        # (csrc/wave_table.h hex_piece()): bytes per CALL = mean per dispatch x dispatches per call
        per_call = shard.hex_dispatches_per_call(nq, nelmt) if (dim == 3 and len(ext) == 1) else 1
        full = ext * dim if len(ext) == 1 else ext
        nmt, nqt = 1, 1
        for q in full:
            nmt, nqt = nmt * (q - 1), nqt * q
        if len(fetch) % per_call or len(write) % per_call:
            print("dispatch count of", d, "is not a multiple of", per_call)
            continue
        rd, wr = 2048.0 * sum(fetch) / len(fetch) * per_call, 1024.0 * sum(write) / len(write) * per_call
        alg = 8 * nelmt * (nmt + nqt)
        row = {"dim": dim, "nq": nq, "nelmt": nelmt, "kernel": kern, "round": int(rnd), "dispatches": len(fetch),
               "dispatches_per_call": per_call,
               "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "hbm_bytes_per_launch": round(rd + wr),
               "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 4),
               "read_over_algorithmic": round(rd / (8 * nelmt * nmt), 4),
               "write_over_algorithmic": round(wr / (8 * nelmt * nqt), 4), "kernel_source_hash": h}
        upsert(rec, row)
        print(f"{d:22s} {kern[:48]:48s} traffic {row['traffic_over_algorithmic']:.4f} x algorithmic "
              f"(reads {row['read_over_algorithmic']:.3f} x, writes {row['write_over_algorithmic']:.3f} x)")
    rec["round"] = int(rnd)
    json.dump(rec, open(path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
