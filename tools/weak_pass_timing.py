#!/usr/bin/env python3
"""Timing of a REFINE_ITER pass with WEAK pixels, labels, adaptive radius and geometric consistency
(BASELINE cfg3/cfg5-like) on one GPU: pass 1 (FIRST_INIT) then pass 2 with ~weak_frac of the pixels
forced WEAK.  usage: weak_pass_timing.py W H S iters weak_frac"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("dvp-mvs_amd")
synth, capi = pkg.synth, pkg.get_capi()
W, H, S, iters, frac = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
sc = synth.make_scene(W, H, S)
p = synth.default_params(S + 1, max_iterations=iters, state=synth.FIRST_INIT, use_APD=0)
p["depth_min"] = np.float32(2.5) * np.float32(0.6)
p["depth_max"] = np.float32(6.5) * np.float32(1.2)
g = capi.from_scene(sc, p)
g.upload_state(planes=np.zeros((H * W, 4), np.float32), edge=sc["edge"], label=sc["label"], radius=np.full(H * W, 5, np.int32))
g.run_patchmatch()
t1 = g.timings(reset=True)
planes, views, weak, radius = g.download_state()
bad = (planes[:, 3] < p["depth_min"]) | (planes[:, 3] > p["depth_max"])
planes[bad, 3] = 0
weak[bad] = synth.UNKNOWN
rng = np.random.default_rng(0)
# WEAK blocks: 32x32 tiles chosen at random until the target fraction is reached (+ the flat window)
wk = weak.reshape(H, W)
tiles = rng.random((H // 32 + 1, W // 32 + 1)) < frac
mask = np.kron(tiles, np.ones((32, 32), bool))[:H, :W] | sc["flat"]
mask[:8] = mask[-8:] = False
mask[:, :8] = mask[:, -8:] = False
wk[mask & (wk == synth.STRONG)] = synth.WEAK
radius[weak == synth.UNKNOWN] = 5
p2 = synth.default_params(S + 1, max_iterations=iters, state=synth.REFINE_ITER, use_APD=1, geom_consistency=1, weak_peak_radius=4)
p2["depth_min"], p2["depth_max"] = p["depth_min"], p["depth_max"]
g.set_params(p2)
g.set_depths(sc["depth_gt"])
g.upload_state(planes=planes, views=views, weak=weak, radius=radius)
g.set_profiling(True)
g.run_patchmatch()
t2 = g.timings()
out = dict(W=W, H=H, S=S, iters=iters, weak_count=g.weak_count(), weak_frac=g.weak_count() / (W * H),
           pass1_total_ms=t1["total_ms"], pass2_total_ms=t2["total_ms"],
           pass2_stage_ms={k: round(v, 2) for k, v in t2["stage_ms"].items() if v > 0},
           pass2_evals={k: int(v) for k, v in t2["ncc_evals"].items() if v > 0},
           weak_after=[int((g.get("weak_info") == k).sum()) for k in range(3)])
print(json.dumps(out))
