"""Oracle parity AT THE SIZES THE METRIC IS QUOTED ON (VERDICT r05 #2 / missing #3): cfg2 3104x2064, cfg3 6208x4128,
cfg5 1920x1080 — on sampled pixels, launch site by launch site, bit for bit.

The oracle needs minutes per launch at 25 Mpx, but every launch site is a per-pixel function of the pre-launch state
(oracle/ora_api.cpp: ora_run_stage_pixels).  So: the engine runs each launch at full size on the GPU; its PRE-launch buffers
are copied into an oracle context of the same size; the oracle runs the launch's per-pixel body on >= 20 000 sampled pixels
(every border class, edge and label pixels, >= 2 000 entries of the WEAK list) and what it leaves at those pixels must be the
engine's post-launch bits.  Every size-dependent path is on: 64 x 14 sweep tiles and XCD strip maps, the split strong update's
cost block, the phased weak update above its 8 192-pixel switch, the anchor table's grids.

Then the forms are tied together at full size on ALL pixels: dvp_run_patchmatch (fused sweeps, both weak colours in one launch
site) == the launch-by-launch run == dvp_run_patchmatch with every monolithic fall-back form (DVP_STRONG_SPLIT=0,
DVP_SWEEP_SPLIT=0, DVP_WEAK_PHASED=0)."""
import os
import time

import numpy as np
import pytest

from conftest import pkg, synth, count_diff, stage_sequence, CHECKED
from oracle import oracle as O

wl = pkg("workloads")
WEAK_IDX = ("neighbours", "complex", "label_boundary")     # indexed by neighbours_map, not by pixel


def _sample_pixels(W, H, weak, edge, label, n_random, rng):
    """[n, 2] (x, y): the whole-launch corner cases first, then random pixels"""
    px = []
    # the 6-pixel frame DepthToWeak marks UNKNOWN and its inner neighbours, along all four sides and in the corners
    for y in list(range(0, 8)) + list(range(H - 8, H)):
        xs = np.concatenate([np.arange(0, 10), np.arange(W - 10, W), rng.integers(0, W, 40)])
        px += [(int(x), y) for x in xs]
    for x in list(range(0, 8)) + list(range(W - 8, W)):
        px += [(x, int(y)) for y in rng.integers(0, H, 60)]
    # the last rows of the red/black half grid (APD.cu:4421-4424) and tile seams of the launch maps (64-pixel columns, 4- and 14-row tiles)
    for x in (63, 64, 65, 511, 512, 513, W // 2):
        px += [(x, int(y)) for y in rng.integers(0, H, 30) if x < W]
    for y in (3, 4, 13, 14, 15, 27, 28, H - 15, H - 14, H - 13):
        px += [(int(x), y) for x in rng.integers(0, W, 30) if 0 <= y < H]
    flat = lambda m: np.flatnonzero(m.reshape(-1))
    def take(idx, n):
        if len(idx) == 0:
            return
        for i in rng.choice(idx, size=min(n, len(idx)), replace=False):
            px.append((int(i % W), int(i // W)))
    take(flat(edge != 0), 1500)
    take(flat(label == -1), 500)
    widx = flat(weak == synth.WEAK)
    take(widx, 3000)
    # STRONG pixels next to WEAK ones (anchors, the propagation's neighbours)
    if len(widx):
        nb = widx[rng.integers(0, len(widx), 1500)] + rng.choice([-1, 1, -W, W, -2, 2, -3 * W, 3 * W], 1500)
        nb = nb[(nb >= 0) & (nb < W * H)]
        px += [(int(i % W), int(i // W)) for i in nb]
    xs, ys = rng.integers(0, W, n_random), rng.integers(0, H, n_random)
    px += list(zip(xs.tolist(), ys.tolist()))
    a = np.unique(np.asarray(px, np.int64), axis=0)
    return a.astype(np.int32)


def _rows(name, arr, L, wc):
    n = max(wc, 1) if name in WEAK_IDX else L
    return arr.reshape(n, -1)


def _engine_for(name, weak_frac, env=None):
    """the engine context of BASELINE config `name` at full size with its pass inputs uploaded and saved (not run)"""
    import torch
    c = wl.CONFIGS[name]
    W, H, S, iters = c["W"], c["H"], c["S"], c["iters"]
    L = W * H
    dev = torch.device("cuda", 0)
    sc = synth.make_scene_torch(W, H, S, dev)
    edge_t, label_t = synth.view_priors_torch(sc["sids"][0], sc["flats"][0])
    edge, label, flat = edge_t.cpu().numpy(), label_t.cpu().numpy(), sc["flats"][0].cpu().numpy()
    old = {}
    for k, v in (env or {}).items():     # the engine reads its form switches when a context is created; None = unset
        old[k] = os.environ.get(k)
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    try:
        g = pkg("capi").Context(W, H, S + 1)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    g.set_images_device([sc["images"][i].data_ptr() for i in range(S + 1)], W)
    g.set_cameras(sc["cameras"])
    p1 = wl.first_init_params(S, iters)
    g.set_params(p1)
    g.set_seed(77)
    first = dict(planes=np.zeros((L, 4), np.float32), views=np.zeros(L, np.uint32), weak=np.full(L, synth.STRONG, np.uint8),
                 edge=edge, label=label, radius=np.full(L, 5, np.int32))
    g.upload_state(**first)
    params = p1
    if c["refine"]:
        g.run_patchmatch()
        st = wl.hand_over(g.get("planes"), g.get("selected_views"), g.get("weak_info"), g.get("radius"), p1, W, H,
                          extra_weak=wl.weak_tiles(W, H, weak_frac, flat))
        params = wl.refine_iter_params(S, iters)
        g.set_params(params)
        g.set_depths_device([sc["depth_gt"][i].data_ptr() for i in range(S + 1)], W)
        g.upload_state(planes=st[0], views=st[1], weak=st[2], radius=st[3])
    g.save_state()
    return dict(g=g, sc=sc, W=W, H=H, S=S, iters=iters, params=params, edge=edge, label=label, refine=c["refine"])


def _oracle_like(r):
    """an oracle context of the same size holding the same images / depth maps / cameras / parameters"""
    sc, W, H, S = r["sc"], r["W"], r["H"], r["S"]
    o = O.Oracle(W, H, S + 1)
    o.set_images([sc["images"][i].cpu().numpy() for i in range(S + 1)])
    if r["refine"]:
        o.set_depths([sc["depth_gt"][i].cpu().numpy() for i in range(S + 1)])
    o.set_cameras(sc["cameras"])
    o.set_params(r["params"])
    o.set_seed(77)
    o.set_sampler(0)
    return o


def _sampled_stage_parity(name, weak_frac, n_random):
    r = _engine_for(name, weak_frac, env=dict(DVP_WEAK_PHASED_MIN=None, DVP_CAND_MASK="1"))   # the shipped weak-update dispatch (tests/conftest.py forces 0 for the small scenes); candidates at anchors only
    g, W, H, S, iters = r["g"], r["W"], r["H"], r["S"], r["iters"]
    L = W * H
    o = _oracle_like(r)
    rng = np.random.default_rng(2026)
    t0 = time.time()
    pre = {n: g.get(n) for n in O.BUFFERS}
    # sizes of the weak-indexed arrays follow the WEAK map (APD.cpp:1182-1193): upload_state first, then every buffer verbatim
    o.upload_state(planes=pre["planes"], views=pre["selected_views"], weak=pre["weak_info"], edge=pre["edge"], label=pre["label"], radius=pre["radius"])
    wc = g.weak_count()
    assert o.weak_count() == wc
    for n in O.BUFFERS:
        o.set(n, pre[n])
    px = _sample_pixels(W, H, pre["weak_info"], pre["edge"], pre["label"], n_random, rng)
    idx = px[:, 1].astype(np.int64) * W + px[:, 0]
    nmap = pre["neighbours_map"]
    was_weak = pre["weak_info"][idx] == synth.WEAK
    widx = nmap[idx[was_weak]].astype(np.int64)
    assert len(px) >= 20000
    if r["refine"]:
        assert was_weak.sum() >= 2000 and wc >= 8192      # the phased weak update's switch is passed at full size
    del pre
    compared = 0
    for st, it, col in stage_sequence(iters):
        g.run_stage(st, it, col)
        names = [n for n in CHECKED if n != "candidate" or st == "gen_edge_inform"]
        post = {n: g.get(n) for n in names}
        visited = o.run_stage_pixels(st, it, col, px)
        assert visited > 0
        for n in names:
            a, b = _rows(n, o.get(n), L, wc), _rows(n, post[n], L, wc)
            sel = widx if n in WEAK_IDX else idx
            if len(sel) == 0:
                continue
            nd = count_diff(a[sel], b[sel])
            assert nd == 0, "%s: %s differs in %d entries of the %d sampled pixels after %s(it=%d, colour=%d)" % (name, n, nd, len(sel), st, it, col)
            compared += 1
        for n in names:      # the engine's post-launch state is the next launch's pre-launch state, everywhere
            o.set(n, post[n])
    final = post
    final["candidate"] = g.get("candidate")
    assert count_diff(final["candidate"], o.get("candidate")) == 0   # nothing after GenEdgeInform writes it
    o.close()
    print("%s: %d sampled pixels (%d WEAK), %d buffer comparisons, %.0f s" % (name, len(px), int(was_weak.sum()), compared, time.time() - t0))
    # ---- the forms at full size, all pixels: one dvp_run_patchmatch from the same inputs ----
    # dvp_run_patchmatch forms the visibility-prior records at the ANCHOR pixels only (their one reader is the weak update, at
    # the anchors of WEAK pixels): `candidate` is compared there, every other buffer everywhere
    nbs = final["neighbours"].reshape(-1, 12, 2)[:, 1:, :].reshape(-1, 2).astype(np.int64)
    anchors = np.unique(nbs[nbs[:, 0] >= 0][:, 1] * W + nbs[nbs[:, 0] >= 0][:, 0]) if wc > 0 else np.zeros(0, np.int64)

    def same_as_final(eng, what):
        for n in CHECKED:
            a, b = final[n], eng.get(n)
            if n == "candidate":
                a, b = a.reshape(L, -1)[anchors], b.reshape(L, -1)[anchors]
            assert count_diff(a, b) == 0, "%s: %s in %s" % (name, what, n)

    g.set("candidate", np.zeros_like(final["candidate"]))      # (nothing left over from the launch-by-launch run)
    g.restore_state()
    g.run_patchmatch()
    same_as_final(g, "dvp_run_patchmatch and the launch-by-launch run differ")
    if wc > 0:
        assert len(anchors) > 1000 and len(anchors) < 0.6 * L
    g.close()
    del g, r
    # ... and with every monolithic fall-back form
    r2 = _engine_for(name, weak_frac, env=dict(DVP_STRONG_SPLIT="0", DVP_SWEEP_SPLIT="0", DVP_WEAK_PHASED="0", DVP_CAND_MASK="0"))
    r2["g"].run_patchmatch()
    same_as_final(r2["g"], "the fall-back forms differ from the default forms")
    r2["g"].close()


@pytest.mark.gpu
def test_cfg3_full_size_sampled_oracle_parity():
    """BASELINE configs[2] at 6208x4128, S = 9, REFINE_ITER, geom, >= 5 % WEAK, priors."""
    _sampled_stage_parity("cfg3", 0.05, 16000)


@pytest.mark.gpu
def test_cfg2_full_size_sampled_oracle_parity():
    """BASELINE configs[1] at 3104x2064, S = 5, 6 iterations, FIRST_INIT."""
    _sampled_stage_parity("cfg2", 0.0, 20000)


@pytest.mark.gpu
def test_cfg5_full_size_sampled_oracle_parity():
    """BASELINE configs[4]'s per-GPU workload at 1920x1080, S = 9, priors, >= 10 % WEAK."""
    _sampled_stage_parity("cfg5", 0.10, 16000)


def test_oracle_launch_on_listed_pixels_equals_the_whole_launch():
    """ora_run_stage_pixels is the check's premise: on the CPU, at a small size, every launch of a REFINE_ITER pass with WEAK
    pixels run (i) on ALL pixels through the pixel list and (ii) on a sample gives the whole launch's bits at those pixels and
    leaves every other pixel alone."""
    W, H, S, iters = 96, 72, 3, 2
    sc = synth.make_scene(W, H, S)
    L = W * H
    p1 = wl.first_init_params(S, iters)
    e = O.from_scene(sc, p1)
    e.upload_state(planes=np.zeros((L, 4), np.float32), views=np.zeros(L, np.uint32), weak=np.full(L, synth.STRONG, np.uint8),
                   edge=sc["edge"], label=sc["label"], radius=np.full(L, 5, np.int32))
    e.run_patchmatch()
    st = wl.hand_over(e.get("planes"), e.get("selected_views"), e.get("weak_info"), e.get("radius"), p1, W, H,
                      extra_weak=wl.weak_tiles(W, H, 0.15, sc["flat"], tile=8))
    e.close()
    p2 = wl.refine_iter_params(S, iters)
    def fresh():
        o = O.from_scene(sc, p2, depths=sc["depth_gt"])
        o.upload_state(planes=st[0], views=st[1], weak=st[2], radius=st[3], edge=sc["edge"], label=sc["label"])
        return o
    whole, listed, sampled = fresh(), fresh(), fresh()
    wc = whole.weak_count()
    assert wc > 50
    yy, xx = np.mgrid[0:H, 0:W]
    all_px = np.stack([xx.reshape(-1), yy.reshape(-1)], 1).astype(np.int32)
    rng = np.random.default_rng(5)
    some = all_px[rng.choice(L, 700, replace=False)]
    sidx = some[:, 1].astype(np.int64) * W + some[:, 0]
    for stg, it, col in stage_sequence(iters):
        pre = {n: whole.get(n) for n in O.BUFFERS}
        for n in O.BUFFERS:
            sampled.set(n, pre[n])
        was_weak = pre["weak_info"][sidx] == synth.WEAK
        widx = pre["neighbours_map"][sidx[was_weak]].astype(np.int64)
        whole.run_stage(stg, it, col)
        listed.run_stage_pixels(stg, it, col, all_px)
        sampled.run_stage_pixels(stg, it, col, some)
        for n in CHECKED:
            a = whole.get(n)
            assert count_diff(a, listed.get(n)) == 0, (n, stg, it, col)
            ra, rs, rp = _rows(n, a, L, wc), _rows(n, sampled.get(n), L, wc), _rows(n, pre[n], L, wc)
            sel = widx if n in WEAK_IDX else sidx
            assert count_diff(ra[sel], rs[sel]) == 0, (n, stg, it, col)
            rest = np.ones(len(ra), bool)
            rest[sel] = False
            assert count_diff(rp[rest], rs[rest]) == 0, (n, stg, it, col)
