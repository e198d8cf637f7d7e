#!/usr/bin/env python3
"""How much of DepthToWeak / LocalRefine is lost to view-selection divergence?  Both loop over the
views a pixel SELECTED (selected_views bit set and view_weight != 0); a wave of 64 x-adjacent pixels
executes the union of its lanes' views.  Prints the mean number of views per pixel, the mean size of the
per-wave union (what a wave executes today) and the mean per-wave maximum (what a loop over each lane's OWN
i-th view would execute).  GPU box: python tools/view_divergence.py [W H S]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("dvp-mvs_amd")
synth, capi, wl = pkg.synth, pkg.get_capi(), pkg.workloads
W, H, S = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1552, 1032, 9)
sc = synth.make_scene(W, H, S)
L = W * H
g = capi.from_scene(sc, wl.first_init_params(S, 3))
g.upload_state(planes=np.zeros((L, 4), np.float32), views=np.zeros(L, np.uint32), weak=np.full(L, synth.STRONG, np.uint8),
               edge=sc["edge"], label=sc["label"], radius=np.full(L, 5, np.int32))
g.run_patchmatch()
planes, views, weak, radius = g.download_state()
st = wl.hand_over(planes, views, weak, radius, wl.first_init_params(S, 3), W, H, extra_weak=wl.weak_tiles(W, H, 0.05, sc["flat"]))
g.set_params(wl.refine_iter_params(S, 3))
g.set_depths([sc["depth_gt"][i] for i in range(S + 1)])
g.upload_state(planes=st[0], views=st[1], weak=st[2], radius=st[3])
g.run_patchmatch()
sel = g.get("selected_views").reshape(-1)[:L].astype(np.uint32)
vw = g.get("view_weight").reshape(L, 32)
mask = np.zeros(L, np.uint32)
for v in range(S):
    mask |= (((sel >> v) & 1).astype(bool) & (vw[:, v] != 0)).astype(np.uint32) << v
pop = np.zeros(L, np.int32)
for v in range(S):
    pop += ((mask >> v) & 1).astype(np.int32)
m2 = mask.reshape(H, W)[:, :W // 64 * 64].reshape(H, W // 64, 64)
p2 = pop.reshape(H, W)[:, :W // 64 * 64].reshape(H, W // 64, 64)
union = np.bitwise_or.reduce(m2, axis=2)
upop = np.zeros(union.shape, np.int32)
for v in range(S):
    upop += ((union >> v) & 1).astype(np.int32)
print("%dx%d S=%d: views per pixel mean %.2f | per-wave union mean %.2f | per-wave max mean %.2f | histogram of per-pixel counts %s"
      % (W, H, S, pop.mean(), upop.mean(), p2.max(axis=2).mean(), np.bincount(pop, minlength=S + 1).tolist()))
