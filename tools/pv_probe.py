#!/usr/bin/env python3
"""Mapping probe (experiment): K plane evaluations x S views per pixel, lane = pixel (mode 0, the NCC kernels' mapping)
against lane = (pixel, view) (mode 1: 64 / S pixels per wave, patch table shared in LDS).  Needs a library built with
-DDVP_PROBE (DVP_MVS_LIB).  usage: pv_probe.py [W H S]"""
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
for it in range(2):      # two strong iterations: partly converged planes in the camera frame
    ctx.run_stage("strong_update", it, 0)
    ctx.run_stage("strong_update", it, 1)
Lb = ctx.L
Lb.dvp_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
for name, stride in (("neighbour planes", 3), ("unrelated planes", 977 * W + 131)):
    for mode in (0, 1):
        ms, ck = ctypes.c_float(0), ctypes.c_float(0)
        rc = Lb.dvp_probe(ctx.h, mode, 8, stride, 3, ctypes.byref(ms), ctypes.byref(ck))
        assert rc == 0
        ev = L * S * 8
        print("%-17s mode %d (%s): %.1f ms, %.2f G evals/s, checksum %.6f" % (name, mode, "lane = pixel" if mode == 0 else "lane = (pixel, view)", ms.value, ev / ms.value / 1e6, ck.value))
