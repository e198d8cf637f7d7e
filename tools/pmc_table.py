#!/usr/bin/env python3
"""Fold the PMC passes of tools/pmc_collect.sh into profiles/pmc_r06.json.
usage: pmc_table.py OUTDIR [TABLE=profiles/pmc_r06.json]

Per kernel only the launches of the bench's LAST step are averaged (the untimed FIRST_INIT pass and
the counting warm-up step launch the same kernels earlier): the last `launches_per_step[kernel]`
dispatches, a number the bench line of the same run states.  Entries are keyed
"<kernel>|<W>x<H>|S<S>" so that bench.py only ever uses counters taken at its own problem size.

Units and corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are KiB derived from the
L2's fabric-side request counters (Infinity-Cache hits included).  On gfx950 FETCH_SIZE tallies the
128-byte requests of 16 B/lane reads at 64 bytes: the fetch part is doubled; WRITE_SIZE is
uncalibrated and reported raw.  SQ_INSTS_* count wave-level instructions; SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles.  lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) of one pass: the share of the
lanes that were switched on while the VALU worked (0.96 for dvp_strong_eval, 0.47 for dvp_strong_refine).  l1_hit_rate = 1 - TCP_TCC_READ_REQ /
TCP_TOTAL_CACHE_ACCESSES (the vector L1s' read requests to the L2 over their tag look-ups, summed over the CUs)."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel_last(path, launches_per_step):
    rows = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        rows[k][int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    out = {}
    for k, disp in rows.items():
        n = launches_per_step.get(k)
        if not n:
            continue
        ids = sorted(disp)[-n:]
        agg = collections.defaultdict(float)
        for i in ids:
            for c, v in disp[i].items():
                agg[c] += v
        out[k] = {c: v / len(ids) for c, v in agg.items()}
        out[k]["launches_averaged"] = len(ids)
    return out


def main():
    out = sys.argv[1]
    table_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_r06.json")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sha = bench.csrc_sha256()
    ids = set()
    for name in ("fetch", "write", "sq1", "sq2", "sq3", "tcp"):   # the library that produced the counters must be the tree's
        js = os.path.join(out, name + ".json")
        if os.path.exists(js):
            ids.add(json.loads(open(js).read().strip().splitlines()[-1]).get("library_build_id"))
    if ids and ids != {sha}:
        raise SystemExit("PMC passes ran on library build(s) %s, the tree's kernel sources hash to %s: rebuild first" % (sorted(map(str, ids)), sha))
    table = {"notes": __doc__.split("Units and corrections")[1].strip(), "kernels": {}, "csrc_sha256": sha}
    if os.path.exists(table_path):
        old = json.load(open(table_path))
        if old.get("csrc_sha256") == sha:      # counters of other kernel sources are never mixed in
            table = old
    merged = collections.defaultdict(dict)
    cfg = None
    for name in ("fetch", "write", "sq1", "sq2", "sq3", "tcp"):
        js = os.path.join(out, name + ".json")
        cs = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)
        if not (os.path.exists(js) and cs):
            print("pass %s missing" % name, file=sys.stderr)
            continue
        bench = json.loads(open(js).read().strip().splitlines()[-1])
        cfg = bench["config"]
        for k, v in per_kernel_last(cs[0], bench["launches_per_step"]).items():
            if name == "sq3":   # its own pair of counters -> one ratio (SQ_ACTIVE_INST_VALU of pass sq1 stays the table's figure)
                act = v.get("SQ_ACTIVE_INST_VALU", 0.0)
                v = {"SQ_THREAD_CYCLES_VALU": v.get("SQ_THREAD_CYCLES_VALU", 0.0),
                     "lane_utilisation": round(v.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * act), 4) if act else None}
            merged[k].update(v)
    if cfg is None:
        raise SystemExit("no PMC pass found under " + out)
    for k, v in merged.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["fetch_bytes_per_launch_corrected_x2"] = 2.0 * v["FETCH_SIZE"] * 1024.0
            v["write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024.0
            v["hbm_bytes_per_launch"] = v["fetch_bytes_per_launch_corrected_x2"] + v["write_bytes_per_launch"]
        if v.get("TCC_HIT", 0) + v.get("TCC_MISS", 0) > 0:
            v["l2_hit_rate"] = round(v["TCC_HIT"] / (v["TCC_HIT"] + v["TCC_MISS"]), 4)
        # L1 (TCP): line requests that went on to the L2 against all cache accesses of the CU's vector L1 (round 6)
        if v.get("TCP_TOTAL_CACHE_ACCESSES", 0) > 0 and "TCP_TCC_READ_REQ" in v:
            v["l1_hit_rate"] = round(1.0 - min(1.0, v["TCP_TCC_READ_REQ"] / v["TCP_TOTAL_CACHE_ACCESSES"]), 4)
        if v.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in v:
            v["wait_any_frac"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4)
        if v.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_VALU" in v:
            v["valu_active_over_wave_cycles"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 4)
        v["workload"] = cfg["workload"]
        v["command"] = "tools/pmc_collect.sh: rocprofv3 --pmc <group> -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --config %s (separate passes, counters only)" % cfg["baseline_config"]
        table["kernels"]["%s|%dx%d|S%d" % (k, cfg["width"], cfg["height"], cfg["src_views"])] = v
    json.dump(table, open(table_path, "w"), indent=1, sort_keys=True)
    print("wrote %s: %d kernels at %dx%d S=%d" % (table_path, len(merged), cfg["width"], cfg["height"], cfg["src_views"]))


if __name__ == "__main__":
    main()
