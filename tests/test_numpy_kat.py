"""The oracle against an INDEPENDENT float64 numpy reading of the reference's source (tests/np_model.py)
on ROTATED cameras with per-view intrinsics (synth rig "rotated": every view incl. the reference has its
own R, K with fy != fx and an off-centre principal point, a centre off the origin; the geometric term also
with a skew K[1] != 0).  Rounds 1-2 tested with R = I and one shared K only, where a transposed R on both
sides of an oracle-vs-engine comparison would pass everything (missing #2).  Tolerances are
float32-vs-float64 rounding of the same formulas with sampler 1 (exact bilinear fractions)."""
import numpy as np
import pytest

from conftest import pkg, synth, make_params, first_pass_state, second_pass_inputs
from oracle import oracle as O
import np_model as M

wl = pkg("workloads")


pytestmark = pytest.mark.hostbox   # no GPU needed; joins the `-m gpu` run on a GPU box (conftest.py)

def _scene(W=128, H=96, S=4, **kw):
    sc = synth.make_scene(W, H, S, **kw)
    cams = [M.cam64(c) for c in sc["cameras"]]
    imgs = [im.astype(np.float64) for im in sc["images"]]
    deps = [d.astype(np.float64) for d in sc["depth_gt"]]
    return sc, cams, imgs, deps


def _sample_planes(sc, n, rng, depth_jitter=0.03, normal_jitter=0.1, margin=8):
    W, H = sc["width"], sc["height"]
    px = np.stack([rng.integers(margin, W - margin, n), rng.integers(margin, H - margin, n)], 1).astype(np.int32)
    gt = sc["depth_gt"][0]
    depth = gt[px[:, 1], px[:, 0]] * rng.uniform(1 - depth_jitter, 1 + depth_jitter, n)
    nw = np.tile(sc["normal_gt"], (n, 1)) + rng.normal(0, normal_jitter, (n, 3))
    nw /= np.linalg.norm(nw, axis=1, keepdims=True)
    return px, synth.planes_in_ref_cam(sc["cameras"][0], px, depth, nw)


def test_rig_is_really_rotated():
    sc = synth.make_scene(96, 64, 5)
    for cam in sc["cameras"]:
        R = cam["R"].reshape(3, 3).astype(np.float64)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)
        ang = np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))
        assert 5.0 < ang < 30.0                       # every view, the reference included
        assert abs(R[0, 1]) > 0.05 and abs(R[0, 2]) > 1e-3 and abs(R[1, 2]) > 1e-3   # all three axes
        K = cam["K"]
        assert K[0] != K[4] and abs(K[2] - 48) > 0.05 and abs(K[5] - 32) > 0.05
    Ks = np.stack([c["K"] for c in sc["cameras"]])
    assert len(np.unique(Ks[:, 0])) == len(Ks)       # per-view focal lengths
    assert np.linalg.norm(sc["cameras"][0]["c"]) > 0.01


def test_homography_rotated_cameras():
    sc, cams, _, _ = _scene(160, 120, 4)
    L = O.lib()
    rng = np.random.default_rng(3)
    px, planes = _sample_planes(sc, 40, rng, depth_jitter=0.0, normal_jitter=0.0)
    for v in range(1, 5):
        for (x, y), pl in zip(px, planes):
            H32 = np.zeros(9, np.float32)
            L.ora_homography(sc["cameras"][0:1].ctypes.data, sc["cameras"][v:v + 1].ctypes.data, pl.ctypes.data, H32.ctypes.data)
            H64 = np.array(M.homography(cams[0], cams[v], pl.astype(np.float64)))
            assert np.allclose(H32 / H32[8], H64 / H64[8], rtol=3e-4, atol=3e-3)
            # the homography of the TRUE plane maps the pixel onto the projection of its world point
            # (only where the pixel really lies on the slanted plane: surface 0)
            if sc["label"][y, x] != 1:
                continue
            z = float(sc["depth_gt"][0][y, x])
            Xw = M.point_on_world(float(x), float(y), z, cams[0])
            sx, sy, _ = M.project_on_camera(Xw, cams[v])
            qx, qy = M.corresponding_point(H64, float(x), float(y))
            assert abs(qx - sx) < 2e-3 and abs(qy - sy) < 2e-3, (v, x, y, qx - sx, qy - sy)


def test_ncc_old_rotated_cameras_vs_numpy():
    sc, cams, imgs, _ = _scene()
    p = make_params(5, use_radius=0)
    o = O.from_scene(sc, p, sampler=1)
    o.upload_state(**first_pass_state(sc))
    rng = np.random.default_rng(5)
    px, planes = _sample_planes(sc, 120, rng)
    worst, n_in = 0.0, 0
    diffs = []
    for (x, y), pl in zip(px, planes):
        for v in range(1, 5):
            a = o.ncc_old(int(x), int(y), v, pl)
            b = M.ncc_old(imgs, cams, int(x), int(y), v, pl.astype(np.float64))
            diffs.append(abs(a - b))
            n_in += b < 2.0
    diffs = np.array(diffs)
    print("ncc_old oracle vs numpy64: median %.2e p99 %.2e max %.2e (%d of %d inside)" % (np.median(diffs), np.quantile(diffs, 0.99), diffs.max(), n_in, len(diffs)))
    assert n_in > 0.7 * len(diffs)
    assert np.median(diffs) < 2e-5 and np.quantile(diffs, 0.99) < 5e-4 and diffs.max() < 5e-3
    # and the true plane scores low in every view that sees the pixel
    px2, good = _sample_planes(sc, 60, rng, depth_jitter=0.0, normal_jitter=0.0, margin=20)
    low = [o.ncc_old(int(x), int(y), v, pl) for (x, y), pl in zip(px2, good) for v in range(1, 5) if sc["label"][y, x] == 1 and not sc["flat"][y, x]]
    assert np.median(low) < 0.05


def test_geom_cost_rotated_cameras_with_skew_vs_numpy():
    sc, cams, imgs, deps = _scene(skew=0.6)
    assert all(abs(c["K"][1]) > 0.2 for c in cams)
    p = make_params(5, geom_consistency=1)
    o = O.from_scene(sc, p, sampler=1, depths=sc["depth_gt"])
    o.upload_state(**first_pass_state(sc))
    rng = np.random.default_rng(6)
    px, planes = _sample_planes(sc, 200, rng, depth_jitter=0.01, normal_jitter=0.02)
    d = []
    small = 0
    for (x, y), pl in zip(px, planes):
        for v in range(1, 5):
            b = M.geom_cost(deps, cams, int(x), int(y), v, pl.astype(np.float64))
            if b is None:
                continue
            a = o.geom_cost(int(x), int(y), v, pl)
            # the source depth is read at a TRUNCATED pixel: a float32/float64 difference that crosses a pixel
            # boundary changes the texel; those samples are skipped via the 2-px guard below
            sx, sy, _ = M.project_on_camera(M.point_on_world(float(x), float(y), M.depth_from_plane(cams[0], pl.astype(np.float64), x, y), cams[0]), cams[v])
            if abs(sx - round(sx)) < 1e-3 or abs(sy - round(sy)) < 1e-3:
                continue
            d.append(abs(a - b))
            small += b < 1.0
    d = np.array(d)
    print("geom oracle vs numpy64: median %.2e max %.2e, %d samples, %d below 1 px" % (np.median(d), d.max(), len(d), small))
    assert len(d) > 500 and small > 100
    assert np.median(d) < 1e-4 and d.max() < 5e-3


def _two_pass(sc, S, sampler):
    W, H = sc["width"], sc["height"]
    p1 = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
    o1 = O.from_scene(sc, p1, seed=777, sampler=sampler)
    o1.upload_state(**first_pass_state(sc))
    o1.run_patchmatch()
    st = second_pass_inputs(o1, sc)
    weak = st["weak"].reshape(H, W)
    weak[sc["flat"] & (weak == synth.STRONG)] = synth.WEAK
    st["weak"] = weak.reshape(-1)
    p2 = make_params(S + 1, max_iterations=1, state=synth.REFINE_ITER, use_APD=1, geom_consistency=1,
                     weak_peak_radius=4, rotate_time=2, ransac_threshold=0.01)
    o2 = O.from_scene(sc, p2, seed=778, sampler=sampler, depths=sc["depth_gt"])
    o2.upload_state(**st)
    return o2, p2


def test_ncc_new_rotated_cameras_vs_numpy():
    S = 4
    sc, cams, imgs, _ = _scene(128, 96, S)
    o, p = _two_pass(sc, S, 1)
    for st in ("gen_edge_inform", "find_nearest_strong", "gen_neighbours", "neighbour_update", "random_init"):
        o.run_stage(st)
    W, H = sc["width"], sc["height"]
    nmap, nbr = o.get("neighbours_map"), o.get("neighbours").reshape(-1, 12, 2)
    cand = o.get("candidate").reshape(H * W, S, 8, 2)
    views = o.get("selected_views")
    radius = o.get("radius")
    weak = np.flatnonzero(o.get("weak_info") == synth.WEAK)
    assert len(weak) > 50
    rng = np.random.default_rng(8)
    planes_now = o.get("planes")
    diffs, with_anchors = [], 0
    for c in rng.choice(weak, 60, replace=False):
        x, y = int(c % W), int(c // W)
        anchors = [tuple(int(t) for t in a) for a in nbr[nmap[c]]]
        assert anchors[0] == (x, y)
        av = [0 if a[0] == -1 else int(views[a[0] + a[1] * W]) for a in anchors]
        r = int(radius[c])
        inc = max(2, int(2.0 * r / 5.0))
        pl = planes_now[c]
        for v in range(1, S + 1):
            offs = [None if a[0] == -1 else cand[a[0] + a[1] * W, v - 1] for a in anchors]
            b = M.ncc_new(imgs, cams, x, y, v, pl.astype(np.float64), anchors, av, offs, radius=r, increment=inc)
            a_ = o.ncc_new(x, y, v, pl)
            diffs.append(abs(a_ - b))
        with_anchors += sum(1 for a in anchors[1:] if a[0] != -1) > 0
    diffs = np.array(diffs)
    print("ncc_new oracle vs numpy64: median %.2e p99 %.2e max %.2e; %d of 60 pixels have anchors" % (np.median(diffs), np.quantile(diffs, 0.99), diffs.max(), with_anchors))
    assert with_anchors > 30
    assert np.median(diffs) < 5e-5 and np.quantile(diffs, 0.99) < 1e-3 and diffs.max() < 1e-2


def test_depth_to_weak_rotated_cameras_vs_numpy():
    S = 3
    sc, cams, imgs, deps = _scene(112, 80, S)
    o, p = _two_pass(sc, S, 1)
    o.run_patchmatch()
    W, H = sc["width"], sc["height"]
    planes = o.get("planes").copy()            # (world normal, depth) after GetDepthandNormal
    views, vw, radius = o.get("selected_views"), o.get("view_weight").reshape(-1, 32), o.get("radius")
    before = o.get("weak_info").copy()
    o.run_stage("depth_to_weak")
    after = o.get("weak_info")
    rng = np.random.default_rng(9)
    agree = total = fragile = 0
    hist = np.zeros(3, int)
    for c in rng.choice(H * W, 90, replace=False):
        x, y = int(c % W), int(c // W)
        r = int(radius[c])
        st, line, mp = M.depth_to_weak(imgs, deps, cams, x, y, planes[c].astype(np.float64), int(views[c]), vw[c].astype(np.float64),
                                       float(p["depth_min"]), float(p["depth_max"]), True, weak_peak_radius=int(p["weak_peak_radius"]),
                                       radius=r, increment=max(2, int(2.0 * r / 5.0)))
        if st is None:
            fragile += 1
            continue
        total += 1
        agree += int(st == after[c])
        hist[st] += 1
    print("depth_to_weak oracle vs numpy64: %d / %d agree (%d fragile skipped), states %s" % (agree, total, fragile, hist))
    assert total >= 60 and hist[1] > 10
    assert agree >= total - 1     # a float32-vs-float64 flip inside the 61-entry line is possible, rarely


def test_local_refine_rotated_cameras_vs_numpy():
    """LocalRefine (APD.cu:4053-4139): the cost at the current depth, the sweep over disparity offsets -5..5 with the source's
    two ways of adding the geometric term, the 0.1 acceptance rule — numpy float64 reading against the oracle's launch."""
    S = 3
    sc, cams, imgs, deps = _scene(112, 80, S)
    o, p = _two_pass(sc, S, 1)
    o.run_patchmatch()
    W, H = sc["width"], sc["height"]
    o.run_stage("depth_to_weak")
    planes = o.get("planes").copy()
    # a converged pass leaves nothing to refine: push a third of the depths off by 8-30 % (one to four disparity steps at this scale) so that the sweep has something to find
    rng = np.random.default_rng(12)
    off = rng.random(H * W) < 0.33
    planes[off, 3] *= (1.0 + rng.choice([-1.0, 1.0], int(off.sum())) * rng.uniform(0.08, 0.3, int(off.sum()))).astype(np.float32)
    o.set("planes", planes)
    views, vw, radius = o.get("selected_views"), o.get("view_weight").reshape(-1, 32), o.get("radius")
    o.run_stage("local_refine")
    after = o.get("planes")
    rng = np.random.default_rng(10)
    agree = total = fragile = moved = 0
    for c in rng.choice(H * W, 120, replace=False):
        x, y = int(c % W), int(c // W)
        r = int(radius[c])
        z, frag = M.local_refine(imgs, deps, cams, x, y, planes[c].astype(np.float64), int(views[c]), vw[c].astype(np.float64),
                                 float(p["depth_min"]), float(p["depth_max"]), True, radius=r, increment=max(2, int(2.0 * r / 5.0)))
        if frag:
            fragile += 1
            continue
        total += 1
        want = float(planes[c][3]) if z is None else z
        moved += z is not None
        agree += abs(float(after[c][3]) - want) <= 2e-5 * abs(want) and (after[c][:3] == planes[c][:3]).all()
    print("local_refine oracle vs numpy64: %d / %d agree (%d fragile skipped), %d moved" % (agree, total, fragile, moved))
    assert total >= 80 and moved >= 3
    assert agree >= total - 1


def test_filter_get_depth_normal_nearest_strong_vs_numpy():
    """The three array-shaped launch sites against readings written from the source: CheckerboardFilterStrong (APD.cu:3184-3294,
    exact: a median of float32 depths), GetDepthandNormal (APD.cu:3167-3182), FindNearestStrongPoint (APD.cu:4159-4193, exact)."""
    S = 3
    sc, cams, imgs, deps = _scene(112, 80, S)
    o, p = _two_pass(sc, S, 1)
    W, H = sc["width"], sc["height"]
    rng = np.random.default_rng(11)
    # FindNearestStrongPoint
    weak = o.get("weak_info").copy()
    o.run_stage("find_nearest_strong")
    wns = o.get("weak_nearest_strong").reshape(-1, 2)
    wk = np.flatnonzero(weak == synth.WEAK)
    assert len(wk) > 50
    for c in rng.choice(wk, 60, replace=False):
        assert tuple(int(t) for t in wns[c]) == M.find_nearest_strong(weak, W, H, int(c % W), int(c // W), STRONG=synth.STRONG)
    assert (wns[weak != synth.WEAK] == -1).all()
    # the pass up to the conversion: planes in the camera frame with the plane offset in .w
    for st, colour in (("gen_edge_inform", 0), ("gen_neighbours", 0), ("neighbour_update", 0), ("random_init", 0), ("strong_update", 0), ("strong_update", 1)):
        o.run_stage(st, 0, colour)
    before = o.get("planes").copy()
    o.run_stage("get_depth_normal")
    after = o.get("planes").copy()
    d = []
    for c in rng.choice(H * W, 200, replace=False):
        x, y = int(c % W), int(c // W)
        want = np.array(M.get_depth_normal(cams[0], before[c].astype(np.float64), x, y))
        d.append(np.max(np.abs(after[c] - want) / np.maximum(1e-3, np.abs(want))))
    print("get_depth_normal oracle vs numpy64: max rel %.2e" % max(d))
    assert max(d) < 2e-5
    # CheckerboardFilterStrong, black then red
    costs = o.get("costs")
    state = o.get("weak_info")
    for colour in (0, 1):
        depth = o.get("planes")[:, 3].copy()
        o.run_stage("filter_strong", 0, colour)
        new = o.get("planes")[:, 3]
        checked = 0
        for c in rng.choice(H * W, 400, replace=False):
            x, y = int(c % W), int(c // W)
            if (x + y) % 2 != colour or state[c] == synth.WEAK:   # the launch's half of the checkerboard (APD.cu:3296-3328); WEAK pixels are skipped
                assert new[c].view(np.uint32) == depth[c].view(np.uint32) or np.isnan(new[c])
                continue
            want = M.filter_strong(depth, costs, state, W, H, x, y, STRONG=synth.STRONG)
            want = depth[c] if want is None else want
            assert np.float32(want).view(np.uint32) == new[c].view(np.uint32), (x, y, want, new[c], depth[c])
            checked += 1
        assert checked > 100


def test_strong_propagation_and_view_selection_vs_numpy():
    """The decision half of CheckerboardPropagationStrong (APD.cu:2010-2141, 2462-2567) read from the source into numpy — the
    edge-adaptive and the fixed sample scan with their quirks (`!edge_pt.x == -1`, `dir_index > 4`, `= { 2.0f }`), the
    replacement rule, the joint view selection from the 8 cost vectors and the neighbours' priors, the 15 draws against the
    CDF, FindMinCostIndex' tie rule, the adoption test — against what the oracle's launch leaves in view_weight and
    selected_views.  The random numbers are the contract's (ora_rand_u32 at the documented site); the costs are the float64
    model's, so pixels where a comparison sits within 2e-4 of flipping are skipped."""
    S = 3
    sc, cams, imgs, _ = _scene(112, 80, S)
    W, H = sc["width"], sc["height"]
    p = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
    seed = 4242
    o = O.from_scene(sc, p, seed=seed, sampler=1)
    o.upload_state(**first_pass_state(sc))
    for st in ("gen_edge_inform", "random_init"):
        o.run_stage(st)
    # one full iteration first, so that costs / planes / selected views are a real mid-pass state
    o.run_stage("strong_update", 0, 0)
    o.run_stage("strong_update", 0, 1)
    it = 1
    planes, costs = o.get("planes").copy(), o.get("costs").copy()
    views_before = o.get("selected_views").copy()
    edge, en = o.get("edge"), o.get("edge_neigh").reshape(H * W, 8, 2)
    radius = o.get("radius")
    o.run_stage("strong_update", it, 0)
    vw_after = o.get("view_weight").reshape(-1, 32)
    views_after = o.get("selected_views")
    planes_after, costs_after = o.get("planes"), o.get("costs")
    refined = improved = fragile2 = 0
    site = (2 << 16) | ((it & 0xFF) << 8) | 0          # rng_site(PH_STRONG, iter, SUB_VIEW), oracle/ora_common.h
    L = O.lib()
    rng = np.random.default_rng(13)
    black = [c for c in range(H * W) if (c % W + c // W) % 2 == 0 and 12 <= c % W < W - 12 and 12 <= c // W < H - 12]
    checked = adopted = fragile = on_edge = 0
    on_edges = [c for c in black if edge[c] != 0]
    sample = list(rng.choice(black, 70, replace=False)) + list(rng.choice(on_edges, min(20, len(on_edges)), replace=False))
    for c in sample:
        x, y = int(c % W), int(c // W)
        u = [((L.ora_rand_u32(seed, int(c), site, k) >> 8) + 1) / 16777216.0 for k in range(15)]
        r = int(radius[c])
        m = M.strong_propagation(imgs, cams, x, y, planes, costs, edge, en, views_before, W, H, it, u, float(p["depth_min"]), float(p["depth_max"]),
                                 radius=r, increment=max(2, int(2.0 * r / 5.0)) if int(p["use_radius"]) else int(p["strong_increment"]))
        if m["fragile"]:
            fragile += 1
            continue
        checked += 1
        on_edge += int(edge[c] != 0)
        assert list(vw_after[c][:S]) == m["view_weight"], (x, y, list(vw_after[c][:S]), m["view_weight"])
        if m["selected"] is not None:
            adopted += 1
            assert int(views_after[c]) == m["selected"], (x, y)
        else:
            assert int(views_after[c]) == int(views_before[c]), (x, y)
        # ... and the refinement that follows (APD.cu:1311-1383 with GenerateRandomNormal_YZL, :501-585): the plane and the
        # cost the launch leaves
        if m["plane"] is None:
            continue
        uni = lambda sub: (lambda k: ((L.ora_rand_u32(seed, int(c), (2 << 16) | ((it & 0xFF) << 8) | sub, k) >> 8) + 1) / 16777216.0)
        inc = max(2, int(2.0 * r / 5.0)) if int(p["use_radius"]) else int(p["strong_increment"])
        pl, cost, frag2 = M.strong_refinement(imgs, cams, x, y, m["plane"], m["cost"], m["view_weight"], m["wn"], int(views_after[c]),
                                              float(p["depth_min"]), float(p["depth_max"]), uni(1)(0), uni(3)(0), uni(2), radius=r, increment=inc)
        if frag2:
            fragile2 += 1
            continue
        refined += 1
        improved += tuple(pl) != tuple(m["plane"])
        got = planes_after[c].astype(np.float64)
        assert np.max(np.abs(got - np.array(pl)) / np.maximum(1e-2, np.abs(pl))) < 2e-4, (x, y, got, pl)
        assert abs(float(costs_after[c]) - cost) < 5e-4, (x, y, costs_after[c], cost)
    print("strong propagation oracle vs numpy: %d pixels checked (%d fragile skipped), %d adoptions, %d edge pixels; refinement: %d checked (%d fragile), %d changed the plane"
          % (checked, fragile, adopted, on_edge, refined, fragile2, improved))
    assert checked >= 40 and adopted >= 5 and on_edge >= 5
    assert refined >= 30 and improved >= 3


def test_gen_edge_inform_vs_numpy():
    """GenEdgeInform (APD.cu:3731-3890) read into numpy: visibility-prior offsets (sector winners ranked by colour weight: exact
    integers), nearest edge pixels (exact), the edge-density sigmoid of WEAK pixels, label boundaries (exact)."""
    S = 4
    sc, cams, imgs, _ = _scene(128, 96, S)
    o, p = _two_pass(sc, S, 1)
    W, H = sc["width"], sc["height"]
    views, edge, label, weak = o.get("selected_views").copy(), o.get("edge").copy(), o.get("label").copy(), o.get("weak_info").copy()
    o.run_stage("gen_edge_inform")
    cand = o.get("candidate").reshape(H * W, S, 8, 2)
    en = o.get("edge_neigh").reshape(H * W, 8, 2)
    nmap = o.get("neighbours_map")
    cx, lb = o.get("complex"), o.get("label_boundary").reshape(-1, 8, 2)
    rng = np.random.default_rng(14)
    wk = np.flatnonzero(weak == synth.WEAK)
    sample = list(rng.choice(H * W, 60, replace=False)) + list(rng.choice(wk, 40, replace=False))
    n_c = n_f = n_lab = 0
    dc = []
    for c in sample:
        x, y = int(c % W), int(c // W)
        m = M.gen_edge_inform(imgs[0], views, edge, label, weak, W, H, x, y, S, weak_radius=int(p["weak_radius"]), strong_radius=int(p["strong_radius"]),
                              sigma_color=float(p["sigma_color"]), WEAK=synth.WEAK)
        assert [tuple(int(t) for t in q) for q in en[c]] == m["edge_neigh"], (x, y)
        for v in range(S):
            if m["candidates"][v] is None:
                n_f += 1
                continue
            n_c += 1
            assert [tuple(int(t) for t in q) for q in cand[c, v]] == m["candidates"][v], (x, y, v)
        if weak[c] == synth.WEAK:
            dc.append(abs(float(cx[nmap[c]]) - m["complex"]))
            if m["label_boundary"] is not None:
                n_lab += 1
                assert [tuple(int(t) for t in q) for q in lb[nmap[c]]] == m["label_boundary"], (x, y)
    print("gen_edge_inform oracle vs numpy: %d candidate lists exact (%d fragile skipped), complex max diff %.1e on %d WEAK pixels, %d label boundaries exact"
          % (n_c, n_f, max(dc), len(dc), n_lab))
    assert n_c > 250 and len(dc) >= 40 and max(dc) < 1e-5 and n_lab >= 10


def test_ransac_fit_plane_vs_numpy():
    """RANSACToGetFitPlane (APD.cu:4195-4405) read into numpy — anchors -> 3-D points, 50 index triplets from the contract's
    stream, normal / triangle / edge-line rejections with the symmetric first-asker cache and BresenhamLine's walk from its
    SECOND argument, the plane through three anchors, the summed depth residual, orientation, the patch-radius rule (Heron,
    nearest anchor, edge and label limits, multiples of 2.5) — against the fit planes and radii the oracle's launch leaves."""
    S = 3
    sc, cams, imgs, _ = _scene(128, 96, S)
    o, p = _two_pass(sc, S, 1)
    W, H = sc["width"], sc["height"]
    for st in ("gen_edge_inform", "find_nearest_strong", "gen_neighbours", "neighbour_update", "random_init"):
        o.run_stage(st)
    o.run_stage("strong_update", 0, 0)
    o.run_stage("strong_update", 0, 1)
    it = 0
    planes, weak, edge = o.get("planes").copy(), o.get("weak_info").copy(), o.get("edge")
    nmap, nbr = o.get("neighbours_map"), o.get("neighbours").reshape(-1, 12, 2)
    en, lb, cx = o.get("edge_neigh").reshape(H * W, 8, 2), o.get("label_boundary").reshape(-1, 8, 2), o.get("complex")
    label = o.get("label")
    radius_before = o.get("radius").copy()
    o.run_stage("ransac_fit", it, 0)
    fit, radius_after = o.get("fit_planes"), o.get("radius")
    L = O.lib()
    seed = 778
    site = lambda sub: (3 << 16) | ((it & 0xFF) << 8) | sub       # rng_site(PH_RANSAC, iter, sub)
    rng = np.random.default_rng(15)
    wk = np.flatnonzero(weak == synth.WEAK)
    checked = fitted = none = frag = limited = 0
    for c in rng.choice(wk, min(150, len(wk)), replace=False):
        x, y = int(c % W), int(c // W)
        edge_limit = False
        if int(p["use_limit"]):
            edge_limit = True
            if int(p["use_edge"]):
                u = ((L.ora_rand_u32(seed, int(c), site(4), 0) >> 8) + 1) / 16777216.0 - 1.1920929e-07
                if abs(u - float(cx[nmap[c]])) < 1e-6:
                    continue
                if u < float(cx[nmap[c]]):
                    edge_limit = False
        anchors = [tuple(int(t) for t in a) for a in nbr[nmap[c]][1:]]
        lbc = [tuple(int(t) for t in q) for q in lb[nmap[c]]] if label[c] > 0 else None
        pl, rad, fragile = M.ransac_fit_plane(cams[0], planes, anchors, edge, [tuple(int(t) for t in q) for q in en[c]], lbc, W, H, x, y,
                                              lambda k: L.ora_rand_u32(seed, int(c), site(6), k), edge_limit, int(p["use_radius"]), int(p["use_edge"]),
                                              int(p["use_label"]), strong_radius=int(p["strong_radius"]))
        if fragile:
            frag += 1
            continue
        checked += 1
        limited += int(edge_limit)
        got = fit[c].astype(np.float64)
        if pl is None:
            none += 1
            assert (got == 0).all(), (x, y, got)
        else:
            fitted += 1
            assert np.max(np.abs(got - np.array(pl)) / np.maximum(1e-2, np.abs(pl))) < 5e-4, (x, y, got, pl)
        if rad is not None:
            assert int(radius_after[c]) == rad, (x, y, int(radius_after[c]), rad)
        else:
            assert int(radius_after[c]) == int(radius_before[c])
    print("ransac_fit_plane oracle vs numpy: %d WEAK pixels checked (%d fragile skipped): %d planes fitted, %d without a plane, %d with the edge limit on"
          % (checked, frag, fitted, none, limited))
    assert checked >= 80 and fitted >= 20 and limited >= 20
    assert (fit[weak != synth.WEAK] == planes[weak != synth.WEAK]).all()      # APD.cu:4208-4211


def test_weak_update_vs_numpy():
    """CheckerboardPropagationWeak + PlaneHypothesisRefinementWeak (APD.cu:2739-3089, 1897-2008) read into numpy on top of the
    NCCNew model: the anchors' planes as candidates, the priors from the anchors' selected views, view selection, the geometric
    term in the weighted costs (3.0 for an absent anchor), adoption, the fit-plane test (and its early return), the six
    refinement hypotheses with GenerateRandomNormal_YZL reading the source depth maps, and the launch's final plain-NCC cost —
    against the view weights, selected views, planes and costs the oracle's launch leaves."""
    S = 3
    sc, cams, imgs, deps = _scene(128, 96, S)
    o, p = _two_pass(sc, S, 1)
    W, H = sc["width"], sc["height"]
    for st in ("gen_edge_inform", "find_nearest_strong", "gen_neighbours", "neighbour_update", "random_init"):
        o.run_stage(st)
    it = 0
    o.run_stage("strong_update", it, 0)
    o.run_stage("strong_update", it, 1)
    o.run_stage("ransac_fit", it, 0)
    planes, weak, views_before = o.get("planes").copy(), o.get("weak_info").copy(), o.get("selected_views").copy()
    nmap, nbr = o.get("neighbours_map"), o.get("neighbours").reshape(-1, 12, 2)
    cand = o.get("candidate").reshape(H * W, S, 8, 2)
    fit, radius = o.get("fit_planes").copy(), o.get("radius").copy()
    o.run_stage("weak_update", it, 0)
    vw_after, views_after = o.get("view_weight").reshape(-1, 32), o.get("selected_views")
    planes_after, costs_after = o.get("planes"), o.get("costs")
    L = O.lib()
    seed = 778
    uni = lambda c, sub: (lambda k: ((L.ora_rand_u32(seed, int(c), (4 << 16) | ((it & 0xFF) << 8) | sub, k) >> 8) + 1) / 16777216.0)   # rng_site(PH_WEAK, ...)
    rng = np.random.default_rng(16)
    wk = [c for c in np.flatnonzero(weak == synth.WEAK) if (c % W + c // W) % 2 == 0]
    checked = frag = adopted = moved = 0
    for c in rng.choice(wk, min(60, len(wk)), replace=False):
        x, y = int(c % W), int(c // W)
        anchors = [tuple(int(t) for t in a) for a in nbr[nmap[c]]]
        r = int(radius[c])
        m = M.weak_update(imgs, deps, cams, x, y, planes, weak, views_before, anchors, cand, fit[c], W, H, it, [uni(c, 0)(k) for k in range(15)],
                          uni(c, 1)(0), uni(c, 3)(0), uni(c, 2), float(p["depth_min"]), float(p["depth_max"]), True, float(p["geom_factor"]),
                          r, max(2, int(2.0 * r / 5.0)), STRONG=synth.STRONG)
        if m["fragile"]:
            frag += 1
            continue
        checked += 1
        assert list(vw_after[c][:S]) == m["view_weight"], (x, y, list(vw_after[c][:S]), m["view_weight"])
        if m["selected"] is not None:
            adopted += 1
            assert int(views_after[c]) == m["selected"], (x, y)
        else:
            assert int(views_after[c]) == int(views_before[c]), (x, y)
        got = planes_after[c].astype(np.float64)
        assert np.max(np.abs(got - np.array(m["plane"])) / np.maximum(1e-2, np.abs(m["plane"]))) < 3e-4, (x, y, got, m["plane"])
        moved += int(np.max(np.abs(got - planes[c])) > 0)
        # the launch's last step (APD.cu:3072-3088): plain NCC of the final plane at the default radius
        sr = int(p["strong_radius"])
        inc = max(2, int(2.0 * sr / 5.0)) if int(p["use_radius"]) else int(p["strong_increment"])
        want = sum(m["view_weight"][j] * M.ncc_old(imgs, cams, x, y, j + 1, m["plane"], sr, inc) for j in range(S)) / m["wn"]
        assert abs(float(costs_after[c]) - want) < 1e-3, (x, y, costs_after[c], want)
    print("weak update oracle vs numpy: %d WEAK pixels checked (%d fragile skipped), %d adoptions, %d planes changed" % (checked, frag, adopted, moved))
    assert checked >= 25 and moved >= 5


def test_gen_neighbours_vs_numpy():
    """GenNeighbours (APD.cu:3330-3711) and NeigbourUpdate (:3713-3729) read into numpy: directional search with the unsigned
    shift arithmetic, nearest-STRONG mapping, duplicate / angle / edge-line tests, the label extension (MIN(1, MAX(..)) = 1
    step, the half-integer directions truncated to 0), the RANSAC with its one-normal test and strong-plane rule, the ranking —
    against the anchors and the reliability flags the oracle's launches leave."""
    S = 3
    sc, cams, imgs, _ = _scene(128, 96, S)
    o, p = _two_pass(sc, S, 1)
    W, H = sc["width"], sc["height"]
    for st in ("gen_edge_inform", "find_nearest_strong"):
        o.run_stage(st)
    planes, weak, edge, label = o.get("planes").copy(), o.get("weak_info").copy(), o.get("edge"), o.get("label")
    wns = o.get("weak_nearest_strong").reshape(-1, 2)
    nmap = o.get("neighbours_map")
    lb, cx = o.get("label_boundary").reshape(-1, 8, 2), o.get("complex")
    o.run_stage("gen_neighbours")
    nbr, reliable = o.get("neighbours").reshape(-1, 12, 2), o.get("weak_reliable")
    o.run_stage("neighbour_update")
    weak_after = o.get("weak_info")
    L = O.lib()
    seed = 778
    draw = lambda c, sub: (lambda k: L.ora_rand_u32(seed, int(c), (5 << 16) | sub, k))       # rng_site(PH_NEIGHBOURS, 0, sub)
    rng = np.random.default_rng(17)
    wk = np.flatnonzero(weak == synth.WEAK)
    checked = frag = rel = lab = 0
    for c in rng.choice(wk, min(200, len(wk)), replace=False):
        x, y = int(c % W), int(c // W)
        u = ((draw(c, 4)(0) >> 8) + 1) / 16777216.0
        got, r, fragile, tiny = M.gen_neighbours(cams[0], planes, weak, wns, edge, label, [tuple(int(t) for t in q) for q in lb[nmap[c]]], float(cx[nmap[c]]),
                                           W, H, x, y, u, draw(c, 5), draw(c, 6), float(p["depth_min"]), float(p["depth_max"]), int(p["rotate_time"]),
                                           float(p["ransac_threshold"]), int(p["use_limit"]), int(p["use_edge"]), int(p["use_label"]), STRONG=synth.STRONG)
        if fragile:
            frag += 1
            continue
        checked += 1
        lab += int(label[c] > 0)
        assert int(reliable[c]) == r, (x, y, int(reliable[c]), r)
        assert int(weak_after[c]) == (synth.WEAK if r == 1 else synth.UNKNOWN), (x, y)
        mine = [tuple(int(t) for t in q) for q in nbr[nmap[c]]]
        assert mine[0] == (x, y)
        if r == 1:
            rel += 1
            assert set(mine[1:1 + tiny]) == set(got[:tiny]) and mine[1 + tiny:] == got[tiny:], (x, y, tiny, mine[1:], got)
    print("gen_neighbours oracle vs numpy: %d WEAK pixels checked (%d fragile skipped), %d reliable with identical anchor lists, %d with a label"
          % (checked, frag, rel, lab))
    assert checked >= 60 and rel >= 30 and lab >= 20


def test_random_initialization_vs_numpy():
    """RandomInitialization (APD.cu:1273-1309), both branches, against the oracle's launch: the random hypothesis of a first
    pass (depth + GenerateRandomNormal_YZL from the contract's streams), the kept prior plane, the top-k cost / view rule;
    and the conversion + selected-view pruning (unSetBit clears bits 0..n) of a later pass."""
    S = 4
    sc, cams, imgs, _ = _scene(112, 80, S)
    W, H = sc["width"], sc["height"]
    L = O.lib()
    rng = np.random.default_rng(18)
    # --- FIRST_INIT: zero planes (out of range -> random) with a band of prior planes
    p1 = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0)
    seed = 909
    o = O.from_scene(sc, p1, seed=seed, sampler=1)
    st = first_pass_state(sc)
    prior = np.zeros((H * W, 4), np.float32)
    band = np.arange(H * W) % W < 30
    prior[band] = np.concatenate([np.tile(sc["normal_gt"], (int(band.sum()), 1)), sc["depth_gt"][0].reshape(-1, 1)[band]], 1)
    st["planes"] = prior
    o.upload_state(**st)
    o.run_stage("gen_edge_inform")
    o.run_stage("random_init")
    planes, costs, views, radius = o.get("planes"), o.get("costs"), o.get("selected_views"), o.get("radius")
    uni = lambda c, sub: (lambda k: ((L.ora_rand_u32(seed, int(c), (1 << 16) | sub, k) >> 8) + 1) / 16777216.0)   # rng_site(PH_RANDOM_INIT, 0, sub)
    n_rand = n_prior = frag = 0
    for c in rng.choice(H * W, 120, replace=False):
        x, y = int(c % W), int(c // W)
        r = int(radius[c])
        pl, cost, sel, fragile = M.random_initialization(imgs, cams, x, y, prior[c], 0, True, float(p1["depth_min"]), float(p1["depth_max"]), int(p1["top_k"]),
                                                          uni(c, 1)(0), uni(c, 2), radius=r, increment=max(2, int(2.0 * r / 5.0)) if int(p1["use_radius"]) else int(p1["strong_increment"]))
        if fragile:
            frag += 1
            continue
        if band[c]:
            n_prior += 1
            assert (planes[c] == prior[c]).all()
        else:
            n_rand += 1
            assert np.max(np.abs(planes[c] - np.array(pl)) / np.maximum(1e-2, np.abs(pl))) < 2e-4, (x, y, planes[c], pl)
        assert abs(float(costs[c]) - cost) < 5e-4 and int(views[c]) == sel, (x, y, costs[c], cost, int(views[c]), sel)
    # --- a later pass
    o2, p2 = _two_pass(sc, S, 1)
    before, vb = o2.get("planes").copy(), o2.get("selected_views").copy()
    o2.run_stage("gen_edge_inform")
    o2.run_stage("random_init")
    planes2, costs2, views2, radius2 = o2.get("planes"), o2.get("costs"), o2.get("selected_views"), o2.get("radius")
    pruned = 0
    for c in rng.choice(H * W, 100, replace=False):
        x, y = int(c % W), int(c // W)
        r = int(radius2[c])
        pl, cost, sel, _ = M.random_initialization(imgs, cams, x, y, before[c].astype(np.float64), int(vb[c]), False, float(p2["depth_min"]), float(p2["depth_max"]),
                                                   int(p2["top_k"]), 0.0, None, radius=r, increment=max(2, int(2.0 * r / 5.0)))
        assert np.max(np.abs(planes2[c] - np.array(pl)) / np.maximum(1e-2, np.abs(pl))) < 2e-5, (x, y)
        assert abs(float(costs2[c]) - cost) < 5e-4 and int(views2[c]) == sel, (x, y, costs2[c], cost, int(views2[c]), sel)
        pruned += int(sel != int(vb[c]))
    print("random_initialization oracle vs numpy: first pass %d random + %d prior pixels (%d fragile skipped); later pass 100 pixels, %d with pruned views" % (n_rand, n_prior, frag, pruned))
    assert n_rand >= 50 and n_prior >= 15
