#!/usr/bin/env python3
"""Fold the DVP_HOST_TIMING log of `apd` (tools/e2e_timing.sh) into a per-pass table: mean wall time per view of every
host step and of the GPU's RunPatchMatch, host overhead = everything that is not GPU kernel time.
usage: e2e_summary.py e2e_apd.log W H"""
import collections
import re
import sys

log = open(sys.argv[1]).read().split("\n")
W, H = int(sys.argv[2]), int(sys.argv[3])
passes = collections.OrderedDict()     # iteration -> {step: [ms...]}
it = None
for ln in log:
    m = re.match(r"Iteration: (\d+)", ln)
    if m:
        it = int(m.group(1))
        passes.setdefault(it, collections.defaultdict(list))
    m = re.match(r"\s+\[host\] (.*): ([0-9.]+) ms", ln)
    if m and it is not None:
        passes[it][m.group(1)].append(float(m.group(2)))
    m = re.match(r"Pass (\d+): (\d+) views in ([0-9.]+) ms, (\d+) in flight", ln)
    if m:   # the pass' own clock: with several views in flight a view's wall time contains its neighbours' kernels
        passes.setdefault(int(m.group(1)), collections.defaultdict(list))["PASS"] = [float(m.group(3)), int(m.group(2)), int(m.group(4))]
    m = re.match(r"Cost time: (\d+) ms \(GPU RunPatchMatch ([0-9.]+) ms, ([0-9.]+) Mpx/s/iter\)", ln)
    if m and it is not None:
        passes[it]["TOTAL wall per view"].append(float(m.group(1)))
        passes[it]["GPU RunPatchMatch (kernel time)"].append(float(m.group(2)))
print("# apd end to end, %dx%d folder (JPEG images), DVP_HOST_TIMING=1; mean ms per view, per pass of the schedule" % (W, H))
for it, steps in passes.items():
    n = len(steps["TOTAL wall per view"])
    if not n:
        continue
    tot = sum(steps["TOTAL wall per view"]) / n
    flight = ""
    if steps.get("PASS"):
        tot = steps["PASS"][0] / max(1, steps["PASS"][1])
        if steps["PASS"][2] > 1:
            flight = " (%d views in flight: pass time / views)" % steps["PASS"][2]
    gpu = sum(steps["GPU RunPatchMatch (kernel time)"]) / n
    print("\npass %d (%d views): wall %.0f ms per view, GPU kernels %.0f ms, host overhead %.0f ms = %.1f %% of the kernel time%s" % (it, n, tot, gpu, tot - gpu, 100.0 * (tot - gpu) / gpu, flight))
    for k, v in steps.items():
        if k.startswith("TOTAL") or k.startswith("GPU") or k == "PASS":
            continue
        print("    %-62s %9.1f" % (k, sum(v) / len(v)))
for ln in log:
    if ln.startswith("real") or ln.startswith("user") or ln.startswith("sys"):
        print("# " + ln)
