#!/usr/bin/env python3
"""debug aid: cfg3-shaped two-pass run, engine vs oracle after every stage; prints which buffers differ"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import pkg, synth, count_diff, stage_sequence, CHECKED
from oracle import oracle as O
import test_baseline_configs as T
W, H, S, iters = int(sys.argv[1]), int(sys.argv[2]), 9, 2
sc = synth.make_scene(W, H, S)
capi = pkg("capi")
a = T._two_pass(lambda s, p: O.from_scene(s, p), sc, S, iters, 0.05)
b = T._two_pass(lambda s, p: capi.from_scene(s, p), sc, S, iters, 0.05)
for st, it, col in stage_sequence(iters):
    a.run_stage(st, it, col); b.run_stage(st, it, col)
    bad = {n: count_diff(a.get(n), b.get(n)) for n in CHECKED}
    bad = {k: v for k, v in bad.items() if v}
    if bad:
        print(st, it, col, bad)
        ca, cb = a.get("costs"), b.get("costs")
        idx = np.nonzero(ca.view(np.uint32) != cb.view(np.uint32))[0][:10]
        vwa, vwb = a.get("view_weight"), b.get("view_weight")
        for i in idx:
            print(" px", i % W, i // W, "cost", ca[i], cb[i], "vw", vwa[i][:S], vwb[i][:S], "weak", a.get("weak_info")[i])
        break
else:
    print("all equal")
