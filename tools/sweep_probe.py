#!/usr/bin/env python3
"""Sweep-shaped mapping probe (round 6): DepthToWeak's work per pixel — the 61 disparity slots around the
pixel's own plane + the current depth, against a set of source views — with (mode 2) lane = pixel, slots in a loop, the mapping
of dvp_sweep_eval, and (mode 3) wave = ONE pixel, lane = slot.  Evaluator only (no geometric term, no folds, no stores).
Needs a library built with -DDVP_PROBE (DVP_MVS_LIB; -DDVP_PROBE_LB=n = waves per SIMD of mode 3).
usage: sweep_probe.py [W H S]"""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
pkg = importlib.import_module("dvp-mvs_amd")
importlib.import_module("dvp-mvs_amd.workloads")
synth, wl, capi = pkg.synth, pkg.workloads, pkg.get_capi()
W, H, S = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (6208, 4128, 9)
dev = torch.device("cuda", 0)
sc = synth.make_scene_torch(W, H, S, dev)
ctx = capi.Context(W, H, S + 1, device=0)
ctx.set_images_device([sc["images"][i].data_ptr() for i in range(S + 1)], W)
ctx.set_cameras(sc["cameras"])
ctx.set_params(wl.first_init_params(S, 1))
ctx.set_seed(1)
L = W * H
ctx.upload_state(planes=np.zeros((L, 4), np.float32), views=np.zeros(L, np.uint32), weak=np.full(L, synth.STRONG, np.uint8),
                 edge=np.zeros(L, np.uint8), label=np.zeros(L, np.int32), radius=np.full(L, 5, np.int32))
for st in ("gen_edge_inform", "random_init"):
    ctx.run_stage(st)
for it in range(2):      # two strong iterations: partly converged planes
    ctx.run_stage("strong_update", it, 0)
    ctx.run_stage("strong_update", it, 1)
ctx.run_stage("get_depth_normal")   # (world normal, depth): what DepthToWeak reads
Lb = ctx.L
Lb.dvp_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
print("# build:", Lb.dvp_build_id().decode() if hasattr(Lb, "dvp_build_id") else "?")
cams = sc["cameras"]
c0 = np.asarray(cams[0]["c"], np.float64)


def run(mode, vmask, repeat=2):
    ms, ck = ctypes.c_float(0), ctypes.c_float(0)
    rc = Lb.dvp_probe(ctx.h, mode, vmask, 0, repeat, ctypes.byref(ms), ctypes.byref(ck))
    assert rc == 0
    return ms.value, ck.value


names = {2: "lane = pixel", 3: "wave = pixel, lane = slot"}
full = (1 << S) - 1
for mode in (2, 3):
    ms, ck = run(mode, full)
    print("all %d views    mode %d (%s): %.1f ms, %.2f G slot-evaluations/s (62 x S per pixel), checksum %.6f" % (S, mode, names[mode], ms, L * S * 62 / ms / 1e6, ck))
for v in range(S):
    cv = np.asarray(cams[v + 1]["c"], np.float64)
    # direction of the baseline in the reference camera's frame ~ direction of the epipolar lines
    R = np.asarray(cams[0]["R"], np.float64).reshape(3, 3)
    b = R @ (cv - c0)
    ang = np.degrees(np.arctan2(b[1], b[0]))
    line = "view %d (baseline at %6.1f deg)" % (v + 1, ang)
    for mode in (2, 3):
        ms, ck = run(mode, 1 << v, repeat=1)
        line += "  mode %d: %.1f ms, %.2f G/s" % (mode, ms, L * 62 / ms / 1e6)
    print(line)
