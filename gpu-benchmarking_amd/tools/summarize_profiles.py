#!/usr/bin/env python3
"""Turn tools/collect_profiles.sh output into the committed evidence:
   profiles/rNN/{kernel_stats.csv, bench_under_kernel_trace.json, bench_unprofiled.json,
                 pmc_fetch_hex_wave_kernel.csv, pmc_write_hex_wave_kernel.csv}  and  profiles/hbm_traffic.json
   usage: summarize_profiles.py PROFDIR ROUND   (run in the repo root)"""
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(prof, rnd):
    dst = f"profiles/r{int(rnd):02d}"
    os.makedirs(dst, exist_ok=True)
    shutil.copy(glob.glob(f"{prof}/kt/**/*kernel_stats.csv", recursive=True)[0], f"{dst}/kernel_stats.csv")
    shutil.copy(f"{prof}/kt_bench.json", f"{dst}/bench_under_kernel_trace.json")
    shutil.copy(f"{prof}/bench_unprofiled.json", f"{dst}/bench_unprofiled.json")
    summ = {}
    for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = glob.glob(f"{prof}/{name}/**/*counter_collection.csv", recursive=True)[0]
        rows = [r for r in csv.DictReader(open(f)) if "hex_wave_kernel" in r["Kernel_Name"]]
        vals = [float(r["Counter_Value"]) for r in rows]
        summ[ctr] = {"dispatches": len(vals), "mean_kb": sum(vals) / len(vals), "min_kb": min(vals),
                     "max_kb": max(vals)}
        with open(f"{dst}/{name}_hex_wave_kernel.csv", "w") as out:
            w = csv.writer(out)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name",
                        "Counter_Value"])
            for r in rows:
                w.writerow([r["Dispatch_Id"], r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"],
                            r["Counter_Name"], r["Counter_Value"]])
    bench = json.load(open(f"{prof}/kt_bench.json"))
    nq, nelmt = bench["config"]["nq"], bench["config"]["elements_per_gpu"]
    fetch_b = summ["FETCH_SIZE"]["mean_kb"] * 1024 * 2  # gfx950: 128-B requests tallied at 64 B -> x2
    write_b = summ["WRITE_SIZE"]["mean_kb"] * 1024
    alg = nelmt * 8 * ((nq - 1) ** 3 + nq ** 3)
    # one row of profiles/hbm_traffic.json (the other shapes: tools/collect_traffic.sh + summarize_traffic.py)
    sys.path.insert(0, os.getcwd())
    import __graft_entry__ as ge
    from summarize_traffic import upsert
    path = "profiles/hbm_traffic.json"
    rec = json.load(open(path)) if os.path.exists(path) else {}
    row = {"dim": 3, "nq": nq, "nelmt": nelmt, "kernel": "hex_wave_kernel", "round": int(rnd),
           "dispatches": summ["FETCH_SIZE"]["dispatches"], "hbm_read_bytes": round(fetch_b),
           "hbm_write_bytes": round(write_b), "hbm_bytes_per_launch": round(fetch_b + write_b),
           "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((fetch_b + write_b) / alg, 4),
           "kernel_source_hash": ge.load_package().shard.kernel_source_hash(os.getcwd()),
           "source": "python3 bench.py under rocprofv3 (tools/collect_profiles.sh)"}
    upsert(rec, row)
    rec["round"] = int(rnd)
    json.dump(rec, open(path, "w"), indent=1)
    rec = {"rows": [row]}
    ks = [r for r in csv.DictReader(open(f"{dst}/kernel_stats.csv")) if "hex_wave_kernel" in r["Name"]][0]
    print(f"kernel trace: {ks['Calls']} calls, average {float(ks['AverageNs']) * 1e-6:.4f} ms; "
          f"bench.py HIP events in that run {bench['roofline']['kernel_ms']:.4f} ms")
    un = json.load(open(f"{prof}/bench_unprofiled.json"))
    print(f"un-profiled: value {un['value']} GDOF/s, roofline {un['roofline']}")
    print(json.dumps(rec["rows"][0]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
