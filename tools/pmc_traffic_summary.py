#!/usr/bin/env python3
"""Turn the two PMC passes of tools/pmc_traffic.sh into profiles/pmc_strong_update.json.
usage: pmc_traffic_summary.py OUTDIR(with fetch/ and write/) > profiles/pmc_strong_update.json

Units and corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are KiB derived from the
L2's fabric-side request counters (Infinity-Cache hits included).  On gfx950 FETCH_SIZE tallies the
128-byte requests of 16 B/lane reads at 64 bytes: the fetch part is doubled; WRITE_SIZE is
uncalibrated and reported raw."""
import collections
import csv
import glob
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(launches[k]) for c, v in cs.items()} | {"dispatches": len(launches[k])} for k, cs in agg.items()}


def main():
    out = sys.argv[1]
    f = per_kernel(glob.glob(out + "/fetch/*_counter_collection.csv")[0])
    w = per_kernel(glob.glob(out + "/write/*_counter_collection.csv")[0])
    kern = [k for k in f if k.startswith("dvp_strong_update")]
    kern = max(kern, key=lambda k: f[k]["FETCH_SIZE"] * f[k]["dispatches"])
    fetch_kib, write_kib = f[kern]["FETCH_SIZE"], w[kern]["WRITE_SIZE"]
    hit, miss = w[kern]["TCC_HIT"], w[kern]["TCC_MISS"]
    allk = {}
    for k in sorted(set(f) | set(w)):
        allk[k] = {"dispatches": f.get(k, w.get(k))["dispatches"]}
        for src in (f, w):
            for c, v in src.get(k, {}).items():
                if c != "dispatches":
                    allk[k][c + "_per_launch"] = v
    bench = json.loads(open(out + "/fetch.json").read().strip().splitlines()[-1])
    res = {
        "kernel": kern,
        "workload": bench["config"]["workload"] + " (tools/pmc_traffic.sh)",
        "command": "rocprofv3 --pmc FETCH_SIZE ; rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS  (separate passes, counters only)",
        "launches_averaged": f[kern]["dispatches"],
        "fetch_size_kib_per_launch_raw": fetch_kib,
        "write_size_kib_per_launch_raw": write_kib,
        "fetch_bytes_per_launch_corrected_x2": 2.0 * fetch_kib * 1024.0,
        "write_bytes_per_launch": write_kib * 1024.0,
        "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
        "tcc_hit_per_launch": hit, "tcc_miss_per_launch": miss, "l2_hit_rate": hit / (hit + miss),
        "algorithmic_bytes_per_launch": bench["roofline"]["evals_per_launch"] * bench["roofline"]["bytes_per_eval"],
        "notes": "FETCH_SIZE doubled (gfx950 tallies 128-B requests of 16 B/lane reads at 64 B); WRITE_SIZE raw (uncalibrated); "
                 "Infinity-Cache hits are included in both. Most of the write traffic is the kernel's private per-view arrays.",
        "all_kernels": allk,
    }
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
