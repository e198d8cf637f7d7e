#!/usr/bin/env python3
"""How many of the strong update's sampled planes are bitwise duplicates?  (Exact common-
subexpression elimination of NCC evaluations would need many.)  usage: dup_probe.py"""
import importlib, sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("dvp-mvs_amd")
synth, capi = pkg.synth, pkg.get_capi()
W, H, S, iters = 1552, 1032, 5, 6
sc = synth.make_scene(W, H, S)
p = synth.default_params(S + 1, max_iterations=iters, state=synth.FIRST_INIT, use_APD=0)
p["depth_min"] = np.float32(2.5) * np.float32(0.6)
p["depth_max"] = np.float32(6.5) * np.float32(1.2)
g = capi.from_scene(sc, p)
g.upload_state(planes=np.zeros((H * W, 4), np.float32), edge=sc["edge"], label=sc["label"], radius=np.full(H * W, 5, np.int32),
               weak=np.full(H * W, synth.STRONG, np.uint8), views=np.zeros(H * W, np.uint32))
def stat(tag):
    pl = g.get("planes").reshape(H, W, 4).view(np.uint32)
    out = []
    for dx, dy in [(0, -1), (0, -5), (0, -13), (0, -41), (5, 5), (21, 21), (-2, 0), (-22, 0)]:
        a = pl[max(0, -dy):H - max(0, dy), max(0, -dx):W - max(0, dx)]
        b = pl[max(0, dy):H - max(0, -dy), max(0, dx):W - max(0, -dx)]
        out.append(round(float((a == b).all(axis=2).mean()), 3))
    dirs = [(0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (1, 1), (-1, 1), (1, -1)]
    m = 48
    samples = [pl[m:H - m, m:W - m]]
    for dx, dy in dirs:
        for r in (6, 25):
            samples.append(pl[m + dy * r:H - m + dy * r, m + dx * r:W - m + dx * r])
    st = np.stack([s_.astype(np.uint64)[..., 0] * 1000003 ^ s_.astype(np.uint64)[..., 1] * 7919 ^ s_.astype(np.uint64)[..., 2] * 104729 ^ s_.astype(np.uint64)[..., 3] for s_ in samples], -1)
    st.sort(axis=-1)
    uniq = 1 + (np.diff(st, axis=-1) != 0).sum(-1)
    hh, ww = uniq.shape
    ww64 = ww // 64 * 64
    wave_max = uniq[:, :ww64].reshape(hh, ww64 // 64, 64).max(-1)
    print(tag, "pair-equal", out, "unique of 17: mean %.2f  wave-max mean %.2f" % (uniq.mean(), wave_max.mean()), flush=True)
for st in ("gen_edge_inform", "find_nearest_strong", "gen_neighbours", "neighbour_update", "random_init"):
    g.run_stage(st)
stat("init")
for it in range(iters):
    for col in (0, 1):
        g.run_stage("strong_update", it, col)
    stat("iter %d" % it)
