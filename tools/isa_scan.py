#!/usr/bin/env python3
"""Loads that the compiler left inside the branch that guards their use, with a wait right behind them.

Compiles csrc/dvp_engine.hip to gfx950 assembly with the Makefile's flags and counts, per kernel, the vector memory loads whose NEXT
instruction is `s_waitcnt vmcnt(0)`: each is a dependent memory round trip of its own.  Round 6 found dvp_sweep_decide2 with 49 of
them (50 slots of a view fetched one after the other: 20.0 ms for 48 GB) — the source meant 50 loads in flight — and rewrote the
decision passes, the strong decision's random-normal draw and the anchor table's texel fetches accordingly.

usage: tools/isa_scan.py [extra hipcc flags]      (a minute of hipcc; no GPU needed)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dvp-mvs_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DDVP_BUILD_ID=\"scan\""]


def scan(path):
    lines = open(path).read().split("\n")
    stats, cur = {}, None
    for i, l in enumerate(lines):
        m = re.match(r"^(dvp_\w+):", l)
        if m:
            cur = m.group(1)
            stats[cur] = [0, 0, 0]
        if cur is None:
            continue
        if "global_load" in l or "buffer_load" in l or "scratch_load" in l:
            stats[cur][0] += 1
            j = i + 1
            while j < len(lines) and (not lines[j].strip() or lines[j].strip()[0] in ";."):
                j += 1
            if j < len(lines) and re.search(r"s_waitcnt vmcnt\(0\)", lines[j]):
                stats[cur][1] += 1
        if "scratch_" in l:
            stats[cur][2] += 1
        if "s_endpgm" in l:
            cur = None
    return stats


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "engine.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + ["-S", "--cuda-device-only", "-o", out, "dvp_engine.hip"],
                              cwd=CSRC, stderr=subprocess.DEVNULL)
        stats = scan(out)
    print("%-44s %6s %22s %10s" % ("kernel", "loads", "load + immediate wait", "scratch ops"))
    for k, (a, b, c) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        if b or c:
            print("%-44s %6d %22d %10d" % (k, a, b, c))


if __name__ == "__main__":
    main()
