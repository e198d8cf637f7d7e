#!/usr/bin/env python3
"""Predicted multi-GPU speed-up of the `apd` driver from MEASURED single-GPU per-view times — what can be said about the
"1/2/4/8 GPUs, >= 6x at 8" half of the metric without a multi-GPU node.  A discrete-event model of the
schedule, not a measurement: the RCCL calls across devices have never run.

Input: the per-view wall times of every pass of the coarse-to-fine schedule, from a DVP_HOST_TIMING log of `apd`
(tools/e2e_timing.sh writes gpurun_out/e2e_apd.log; `--dump` turns it into the small JSON kept under profiles/).

Work unit = (scene, pass, view).  Jacobi dependency (APD.cpp:1147-1166): pass p+1 of a view reads the pass-p depth maps of its
sources — with 9 sources out of 10 views that is every view of the scene, i.e. a barrier per pass; pair.txt graphs with fewer
sources per view than views loosen it (`--sources`).

Policies
  rr       view v -> rank v % N, barrier per pass, scenes one after the other              (round 4's driver)
  lpt      longest-predicted-first view -> rank table per scene, barrier per pass            (round 5's driver: AssignViews)
  ready    lpt + a view's next pass starts as soon as ITS sources' maps are there            (no global barrier)
  scenes   one rank per scene, scenes handed out longest-first from a queue                 (tools/run_scenes.py)
  hybrid   scenes queue, but a scene whose single-rank time exceeds the fair share runs on 2 / 4 ranks (intra-scene lpt)
  pool2/3  every scene on all ranks (lpt table per scene, per-view readiness), but 2 / 3 scenes IN FLIGHT: a rank that waits
           for the other ranks' maps of scene A runs its views of scene B meanwhile (tools/run_scenes.py --in-flight: two
           `apd` jobs over the same GPUs; two full-size contexts are 220 of the 288 GB)
  ideal    total work / N
`exchange_ms`: what publishing one view's depth map to the other ranks costs per pass (100 MB over xGMI at ~50 GB/s ~ 2 ms;
the host transport of the tests is far slower and is not what a node would use)."""
import argparse
import json
import re
import sys

import numpy as np

ETH3D_TRAIN_VIEWS = [38, 44, 45, 76, 31, 15, 26, 14, 38, 31, 31, 23, 42]   # courtyard ... terrains: 454 views, 13 scenes (cfg4)


def parse_log(path):
    """[{'iteration': i, 'views_ms': [...]}] from `apd`'s 'Iteration:' / 'Cost time:' lines"""
    passes = {}
    it = None
    for ln in open(path):
        m = re.match(r"Iteration: (\d+)", ln)
        if m:
            it = int(m.group(1))
        m = re.match(r"Cost time: (\d+) ms \(GPU RunPatchMatch ([0-9.]+) ms", ln)
        if m and it is not None:
            passes.setdefault(it, []).append(max(float(m.group(1)), float(m.group(2))))
        m = re.match(r"Pass (\d+): (\d+) views in ([0-9.]+) ms, (\d+) in flight", ln)
        if m and int(m.group(4)) > 1 and int(m.group(1)) in passes:
            # several views in flight: a view's own clock contains its neighbours' kernels; share the pass' clock out in proportion
            v = passes[int(m.group(1))]
            k = float(m.group(3)) / max(sum(v), 1e-9)
            passes[int(m.group(1))] = [x * k for x in v]
    return [{"iteration": k, "views_ms": v} for k, v in sorted(passes.items())]


def lpt_table(costs, n):
    owner, load = [0] * len(costs), [0.0] * n
    for v in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(n), key=lambda q: (load[q], q))
        owner[v] = r
        load[r] += costs[v]
    return owner


def scene_time(t, n, policy, exchange_ms, sources=None, rng=None):
    """wall ms of one scene on n ranks; t[p][v] = ms of view v in pass p"""
    P, V = len(t), len(t[0])
    if n == 1:
        return float(sum(sum(r) for r in t))
    if policy == "rr":
        owner = [v % n for v in range(V)]
    else:
        owner = lpt_table([sum(t[p][v] for p in range(P)) for v in range(V)], n)   # (the driver predicts with pixel counts; equal-size views: the same table as rr)
    ex = exchange_ms * V     # every rank takes part in every view's broadcast
    if policy in ("rr", "lpt"):
        total = 0.0
        for p in range(P):
            busy = [0.0] * n
            for v in range(V):
                busy[owner[v]] += t[p][v]
            total += max(busy) + ex
        return total
    # ready: event simulation; a rank runs its own views in index order, pass by pass, each as soon as its sources' previous maps exist
    srcs = sources if sources is not None else [[u for u in range(V) if u != v] for v in range(V)]
    done = [[0.0] * V for _ in range(P)]
    free = [0.0] * n
    for p in range(P):
        for v in sorted(range(V), key=lambda i: (owner[i], i)):
            ready = 0.0 if p == 0 else max([done[p - 1][u] for u in srcs[v]] + [done[p - 1][v]]) + exchange_ms
            start = max(free[owner[v]], ready)
            done[p][v] = start + t[p][v]
            free[owner[v]] = done[p][v]
    return max(done[P - 1])


def job_time(scenes, n, policy, exchange_ms, fusion_ms_per_view, sources_per_view=None, seed=0):
    """scenes: list of t[p][v].  A scene's fusion (host work: RunFusion) runs in the background on the host of a rank that
    finished the scene while the GPUs go on — it delays nothing but the end of the job (and the hosts cannot fuse faster
    than n scenes at a time).  The single-rank time is modelled the same way."""
    rng = np.random.default_rng(seed)

    def sources(V):
        if sources_per_view is None or sources_per_view >= V - 1:
            return None
        return [sorted(rng.choice([u for u in range(V) if u != v], sources_per_view, replace=False).tolist()) for v in range(V)]
    pm = [sum(sum(r) for r in t) for t in scenes]
    fus = [fusion_ms_per_view * len(t[0]) for t in scenes]
    floor = sum(fus) / n
    if policy == "ideal":
        return max(sum(pm) / n + min(fus), floor)
    if policy in ("rr", "lpt", "ready"):   # every scene on all n ranks, one after the other
        tt = [scene_time(t, n, policy, exchange_ms, sources(len(t[0]))) for t in scenes]
        return max(sum(tt) + fus[-1], floor)
    if policy.startswith("pool"):
        return max(pool_time(scenes, n, int(policy[4:]), exchange_ms, fus, sources), floor)
    # scene queue: longest first; `hybrid` gives the scenes above the fair share 2 or 4 ranks
    order = sorted(range(len(scenes)), key=lambda i: -pm[i])
    fair = sum(pm) / n
    free = [0.0] * n
    end = 0.0
    for i in order:
        k = 1
        if policy == "hybrid":
            while k < n and pm[i] / k > 0.75 * fair:
                k *= 2
            k = min(k, n)
        ranks = sorted(range(n), key=lambda r: free[r])[:k]
        start = max(free[r] for r in ranks)
        dur = scene_time(scenes[i], k, "lpt", exchange_ms, None)
        for r in ranks:
            free[r] = start + dur
        end = max(end, start + dur + fus[i])
    return max(end, floor)


def pool_time(scenes, n, in_flight, exchange_ms, fus, sources_of):
    """list scheduling with fixed owners: every rank runs, among ITS views whose sources' previous maps exist, the one of the
    earliest-started scene (then lowest pass); a new scene starts when fewer than `in_flight` are unfinished"""
    order = sorted(range(len(scenes)), key=lambda i: -sum(sum(r) for r in scenes[i]))
    S = len(scenes)
    owner = {i: lpt_table([sum(scenes[i][p][v] for p in range(len(scenes[i]))) for v in range(len(scenes[i][0]))], n) for i in range(S)}
    srcs = {}
    for i in range(S):
        V = len(scenes[i][0])
        q = sources_of(V)
        srcs[i] = q if q is not None else [[u for u in range(V) if u != v] for v in range(V)]
    done = {}                      # (scene, pass, view) -> finish time
    nxt = {i: [0] * len(scenes[i][0]) for i in range(S)}    # next pass of every view
    free = [0.0] * n
    started, finished_at = [], {}
    end = 0.0
    import heapq
    now = 0.0
    pending = list(order)
    while len(finished_at) < S:
        while pending and len([i for i in started if i not in finished_at]) < in_flight:
            started.append(pending.pop(0))
        # earliest possible (rank, task)
        best = None
        for r in range(n):
            for si, i in enumerate(started):
                if i in finished_at:
                    continue
                P = len(scenes[i])
                for v in range(len(scenes[i][0])):
                    if owner[i][v] != r or nxt[i][v] >= P:
                        continue
                    p = nxt[i][v]
                    ready = 0.0
                    if p > 0:
                        need = [done.get((i, p - 1, u)) for u in srcs[i][v]] + [done.get((i, p - 1, v))]
                        if any(x is None for x in need):
                            continue
                        ready = max(need) + exchange_ms
                    start = max(free[r], ready)
                    key = (start, si, p, v)
                    if best is None or key < best[0]:
                        best = (key, r, i, p, v)
        if best is None:
            raise RuntimeError("deadlock in the schedule model")
        (start, _, _, _), r, i, p, v = best
        fin = start + scenes[i][p][v]
        done[(i, p, v)] = fin
        free[r] = fin
        nxt[i][v] = p + 1
        if all(x >= len(scenes[i]) for x in nxt[i]):
            finished_at[i] = max(done[(i, len(scenes[i]) - 1, u)] for u in range(len(scenes[i][0])))
            end = max(end, finished_at[i] + fus[i])
    return end


def synth_scenes(passes, view_counts, seed=1):
    """scenes with the given view counts, every (pass, view) time drawn from the measured views of that pass"""
    rng = np.random.default_rng(seed)
    return [[rng.choice(p["views_ms"], n).tolist() for p in passes] for n in view_counts]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("views", help="per-view times: profiles/*_e2e_views.json, or an apd log (gpurun_out/e2e_apd.log)")
    ap.add_argument("--dump", help="write the parsed log as JSON and exit")
    ap.add_argument("--exchange-ms", type=float, default=2.0)
    ap.add_argument("--fusion-ms-per-view", type=float, default=1210.0, help="RunFusion per view of a scene (12.1 s per ten 25-Mpx views, profiles/r03_fusion_fullres.txt)")
    ap.add_argument("--sources", type=int, default=9, help="sources per view of the multi-scene mix (ETH3D pair.txt: 9 of 14-76 views)")
    a = ap.parse_args()
    passes = json.load(open(a.views))["passes"] if a.views.endswith(".json") else parse_log(a.views)
    if a.dump:
        json.dump({"what": "wall ms per view and pass of `apd` on one MI355X (tools/e2e_timing.sh), input of tools/scale_sim.py", "passes": passes}, open(a.dump, "w"))
        return
    one = [p["views_ms"] for p in passes]
    V = len(one[0])
    print("# tools/scale_sim.py %s   (exchange %.1f ms per view and pass, fusion %.0f ms per view)" % (a.views, a.exchange_ms, a.fusion_ms_per_view))
    print("# PREDICTION from single-GPU per-view times — the RCCL path across devices is unmeasured on hardware")
    jobs = [("one scene, %d views, %d sources each (the e2e folder), no fusion" % (V, V - 1), [one], 0.0, None, ("rr", "lpt", "ready", "ideal")),
            ("cfg4 mix: 13 scenes, %d views (ETH3D training view counts), %d sources each, fusion per scene" % (sum(ETH3D_TRAIN_VIEWS), a.sources),
             synth_scenes(passes, ETH3D_TRAIN_VIEWS), a.fusion_ms_per_view, a.sources, ("rr", "lpt", "ready", "scenes", "hybrid", "pool2", "pool3", "ideal"))]
    for title, scenes, fus, srcs, policies in jobs:
        t1 = job_time(scenes, 1, "rr", a.exchange_ms, fus, srcs)
        print("\n%s — single rank: %.1f s" % (title, t1 / 1e3))
        print("  %-8s %8s %8s %8s" % ("policy", "2 ranks", "4 ranks", "8 ranks"))
        for pol in policies:
            print("  %-8s %8.2f %8.2f %8.2f" % (pol, *[t1 / job_time(scenes, n, pol, a.exchange_ms, fus, srcs) for n in (2, 4, 8)]))


if __name__ == "__main__":
    main()
