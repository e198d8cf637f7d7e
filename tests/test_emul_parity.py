"""Kernel logic on CPU: the engine's device headers, compiled for the host (tests/emul), must
reproduce the oracle bit for bit, launch site by launch site (integer buffers exactly, float
buffers bitwise with NaN == NaN)."""
import numpy as np
import pytest

from conftest import (synth, make_params, count_diff, stage_sequence, CHECKED, first_pass_state,
                      second_pass_inputs)
from oracle import oracle as O
from tests.emul import emul as E


def _pair(scene, params, state, seed=1234, sampler=0, depths=None):
    a = O.from_scene(scene, params, seed=seed, sampler=sampler, depths=depths, cls=O.Oracle)
    b = O.from_scene(scene, params, seed=seed, sampler=sampler, depths=depths, cls=E.Emul)
    a.upload_state(**state)
    b.upload_state(**state)
    return a, b


def _run_and_compare(a, b, iters, names=CHECKED):
    for st, it, col in stage_sequence(iters):
        a.run_stage(st, it, col)
        b.run_stage(st, it, col)
        for n in names:
            nd = count_diff(a.get(n), b.get(n))
            assert nd == 0, "%s differs in %d entries after %s(it=%d, colour=%d)" % (n, nd, st, it, col)


@pytest.mark.parametrize("W,H,S,sampler", [(96, 72, 3, 0), (131, 67, 2, 1)])
def test_first_pass_strong_path(W, H, S, sampler):
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
    a, b = _pair(sc, p, first_pass_state(sc), sampler=sampler)
    _run_and_compare(a, b, 2)


@pytest.mark.parametrize("anchors", ["table", "one_wave", "alloc_fail", "per_item"])
@pytest.mark.parametrize("images", ["8bit", "float"])
def test_two_pass_weak_path_with_geom(images, anchors, monkeypatch):
    """`anchors`: the reference side of the anchor sub-patches from the pass' table (built once, before the first weak
    update of the pass) or formed per (view, anchor, plane) item as the source text does (DVP_WEAK_ANCHOR_TAB=0).  With the
    table the update runs as seven launches (dvp_weak_phased.hpp: "table"); "one_wave" (DVP_WEAK_PHASED=0) and "alloc_fail"
    (the hand-over buffers do not fit: DVP_TEST_WEAK_PHASE_ALLOC_FAIL) keep a pixel's whole update in one wave; all the same bits.
    pass 1 (FIRST_INIT) on the oracle, then a REFINE_ITER pass with WEAK pixels, labels, adaptive
    radius and geometric consistency on both.  `images`: integer grey levels (the weak update reads the
    byte planes) or non-integers (float planes)."""
    monkeypatch.setenv("DVP_WEAK_ANCHOR_TAB", "0" if anchors == "per_item" else "1")
    if anchors == "one_wave":
        monkeypatch.setenv("DVP_WEAK_PHASED", "0")
    if anchors == "alloc_fail":
        monkeypatch.setenv("DVP_TEST_WEAK_PHASE_ALLOC_FAIL", "1")
    W, H, S = 112, 80, 3
    sc = synth.make_scene(W, H, S)
    if images == "float":
        sc["images"] = (sc["images"] * np.float32(0.97) + np.float32(1.3)).astype(np.float32)
    p1 = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
    o = O.from_scene(sc, p1)
    o.upload_state(**first_pass_state(sc))
    o.run_patchmatch()
    st = second_pass_inputs(o, sc)
    # force a block of WEAK pixels in the low-texture window so the weak path has work
    weak = st["weak"].reshape(H, W)
    weak[sc["flat"] & (weak == synth.STRONG)] = synth.WEAK
    weak[:6, :] = synth.UNKNOWN
    st["weak"] = weak.reshape(-1)
    assert (st["weak"] == synth.WEAK).sum() > 50
    p2 = make_params(S + 1, max_iterations=2, state=synth.REFINE_ITER, use_APD=1, geom_consistency=1,
                     weak_peak_radius=4, rotate_time=2, ransac_threshold=0.01)
    depths = sc["depth_gt"]   # stand-in for the neighbours' depths.dmb
    a, b = _pair(sc, p2, st, depths=depths)
    assert a.weak_count() == b.weak_count() > 0
    _run_and_compare(a, b, 2)
    # the weak path really ran
    assert (a.get("weak_reliable") == 1).sum() > 0
    assert np.abs(a.get("fit_planes")).sum() > 0


@pytest.mark.parametrize("form,wpr", [("passes", None), ("fused", None), ("passes", 9), ("passes", 40)])
@pytest.mark.parametrize("geom", [0, 1])
def test_fused_sweeps_equal_the_two_launches(geom, form, wpr, monkeypatch):
    """`form`: the fused launch site as view-compacted passes (prepare / evaluate per view over the pixels that selected it /
    decide, dvp_strong.hpp: sweep_*) or as the one per-pixel kernel (DVP_SWEEP_SPLIT=0); `wpr`: weak_peak_radius — the
    central window that is evaluated first grows with it (9 -> 21 slots, 40 -> the whole line, no second stage).
    run_patchmatch does DepthToWeak + LocalRefine (APD.cu:4502-4505) per pixel in one launch; run_stage keeps them
    apart.  Same bits either way, and both equal the oracle: border pixels, sweep slots outside the depth range
    and the geometric term included."""
    W, H, S = 90, 61, 3
    monkeypatch.setenv("DVP_SWEEP_SPLIT", "2" if form == "passes" else "0")   # 2: the passes also where the geometric term is off (the default takes the fused kernel there)
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0, geom_consistency=geom)
    if wpr is not None:
        p["weak_peak_radius"] = wpr
    p["depth_min"] = np.float32(3.2)
    st = first_pass_state(sc)
    dm = sc["depth_gt"] if geom else None
    ora = O.from_scene(sc, p, depths=dm, cls=O.Oracle)
    whole = O.from_scene(sc, p, depths=dm, cls=E.Emul)
    steps = O.from_scene(sc, p, depths=dm, cls=E.Emul)
    for x in (ora, whole, steps):
        x.upload_state(**st)
    ora.run_patchmatch()
    whole.run_patchmatch()
    for stg, it, col in stage_sequence(1):
        steps.run_stage(stg, it, col)
    for n in ("planes", "weak_info", "radius", "costs", "selected_views"):
        assert count_diff(whole.get(n), steps.get(n)) == 0, n
        assert count_diff(whole.get(n), ora.get(n)) == 0, n
    assert (whole.get("weak_info") == synth.WEAK).sum() > 0


def find_nearest_strong_case(seed, pair):
    """FindNearestStrongPoint alone (APD.cu:4159-4193) on adversarial maps: the engine finds the first hit of a ring from
    row / column segments of bit tiles, the oracle walks the ring pixel by pixel.  WEAK blobs larger than a 32x32
    tile (segments spanning several words), sparse STRONG pixels (ties between the columns and rows of a ring,
    hits at distance > 64), image borders, and pixels with no STRONG pixel within the 100 rings."""
    rng = np.random.default_rng(seed)
    W, H, S = 300, 170, 1
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=1, state=synth.REFINE_ITER, use_APD=1)
    st = first_pass_state(sc)
    weak = np.full((H, W), synth.WEAK, np.uint8)
    n_strong = [40, 6, 400][seed]
    weak.reshape(-1)[rng.choice(W * H, n_strong, replace=False)] = synth.STRONG
    weak[rng.random((H, W)) < 0.05] = synth.UNKNOWN
    if seed == 1:
        weak[:, :120] = np.where(weak[:, :120] == synth.STRONG, synth.UNKNOWN, weak[:, :120])   # nothing within 100 rings on the left
    st["weak"] = weak.reshape(-1)
    a, b = pair(sc, p, st)
    for x in (a, b):
        x.run_stage("find_nearest_strong", 0, 0)
    na, nb = a.get("weak_nearest_strong"), b.get("weak_nearest_strong")
    assert count_diff(na, nb) == 0
    found = (na.reshape(-1, 2)[:, 0] >= 0)
    assert found.any() and (~found[weak.reshape(-1) == synth.WEAK]).any() == (seed == 1)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_find_nearest_strong_ring_order(seed):
    find_nearest_strong_case(seed, _pair)


def gen_neighbours_case(seed, pair, rotate_time=4):
    """GenNeighbours alone (APD.cu:3330-3711) where the directional search has work to do: large WEAK areas with few
    STRONG pixels (a direction runs through many tries — more than one 64-try round of the wave kernel — before it
    finds a new point or leaves the image), random edge segments (the edge-limited line walk rejects tries), label
    regions with boundaries (label extension: up to ~100 items, duplicates among them and against the directional
    points), rotate_time 4 (32 directions).  The oracle walks tries and items one at a time."""
    rng = np.random.default_rng(100 + seed)
    W, H, S = [420, 420, 640][seed], 260, 1
    side = [300, 300, 470][seed]   # a textured side; seed 2: rays from the far left need more than 64 tries to reach it
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=1, state=synth.REFINE_ITER, use_APD=1, rotate_time=rotate_time,
                    use_limit=[1, 1, 0][seed])   # (use_edge stays on: the engine rejects use_edge = 0, dvp_set_params)
    st = first_pass_state(sc)
    L = W * H
    n = np.tile(sc["normal_gt"], (L, 1)) + rng.normal(0, 0.03, (L, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    depth = (sc["depth_gt"][0].reshape(-1) * (1 + rng.normal(0, 0.01, L))).astype(np.float32)
    st["planes"] = np.concatenate([n, depth[:, None]], 1).astype(np.float32)
    st["views"] = np.ones(L, np.uint32)
    weak = np.full((H, W), synth.WEAK, np.uint8)
    weak.reshape(-1)[rng.choice(L, [60, 900, 9][seed], replace=False)] = synth.STRONG
    weak[rng.random((H, W)) < 0.03] = synth.UNKNOWN
    weak[:, side:] = np.where(rng.random((H, W - side)) < 0.4, synth.STRONG, weak[:, side:])
    st["weak"] = weak.reshape(-1)
    edge = np.zeros((H, W), np.uint8)
    for _ in range(25):      # random segments
        x0, y0 = rng.integers(0, W), rng.integers(0, H)
        ang, ln = rng.uniform(0, np.pi), rng.integers(10, 120)
        t = np.arange(ln)
        xs = np.clip((x0 + t * np.cos(ang)).astype(int), 0, W - 1)
        ys = np.clip((y0 + t * np.sin(ang)).astype(int), 0, H - 1)
        edge[ys, xs] = 1
    st["edge"] = edge.reshape(-1)
    yy, xx = np.mgrid[0:H, 0:W]
    label = (1 + xx // 70 + 10 * (yy // 65)).astype(np.int32)
    label[(xx % 70 == 0) | (yy % 65 == 0) | (edge == 1)] = -1
    label[100:140, 150:230] = 0
    st["label"] = label.reshape(-1)
    a, b = pair(sc, p, st)
    for x in (a, b):
        for stage in ("gen_edge_inform", "find_nearest_strong", "gen_neighbours"):
            x.run_stage(stage, 0, 0)
    for name in ("complex", "label_boundary", "weak_nearest_strong", "neighbours", "weak_reliable", "fit_planes"):
        assert count_diff(a.get(name), b.get(name)) == 0, name
    rel = b.get("weak_reliable").reshape(H, W)
    assert (rel[weak == synth.WEAK] == 1).sum() > 1000
    nb = b.get("neighbours").reshape(-1, 12, 2)
    assert (nb[:, 1:, 0] >= 0).any()


@pytest.mark.parametrize("form", ["wave", "per_lane"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gen_neighbours_search_forms_equal_the_oracle(form, seed, monkeypatch):
    """The per-lane search (dvp_gen_neighbours_list: angle test first, signature-filtered duplicate test) and the
    wave-per-pixel search (DVP_GN_WAVE=1, dvp_gen_neighbours_search: the tries of a direction over the lanes; measured
    slower on the GPU, kept as the record of that measurement): the same anchors, in the same order, from the same
    random numbers."""
    monkeypatch.setenv("DVP_GN_WAVE", "1" if form == "wave" else "0")
    gen_neighbours_case(seed, _pair)
    if seed == 0:
        gen_neighbours_case(seed, _pair, rotate_time=2)   # 16 directions in the slots 0,1, 4,5, ... (dir_index = 4 * octant + rotation)


@pytest.mark.parametrize("form", ["wave", "per_lane"])
def test_ransac_fit_plane_forms_equal_the_oracle(form, monkeypatch):
    """RANSACToGetFitPlane one wave per WEAK pixel with a lane per draw (dvp_ransac_fit_plane_wave: who asks a cached line test
    first decides its orientation, the strict minimum in draw order) and one lane per WEAK pixel (the definition): all launch
    sites of a REFINE_ITER pass with anchors, edges (use_limit) and adaptive radii against the oracle."""
    monkeypatch.setenv("DVP_RANSAC_WAVE", "1" if form == "wave" else "0")
    many_views_case(5, _pair, lambda sc, p: O.from_scene(sc, p, cls=O.Oracle))
    gen_neighbours_case(1, _pair)


def many_views_case(S, pair, make_engine):
    W, H = 88, 64
    sc = synth.make_scene(W, H, S)
    p1 = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0)
    g = make_engine(sc, p1)
    g.upload_state(**first_pass_state(sc))
    g.run_patchmatch()
    st = second_pass_inputs(g, sc)
    weak = st["weak"].reshape(H, W)
    weak[sc["flat"] & (weak == synth.STRONG)] = synth.WEAK
    weak[20:30, 40:60] = np.where(weak[20:30, 40:60] == synth.STRONG, synth.WEAK, weak[20:30, 40:60])
    st["weak"] = weak.reshape(-1)
    p2 = make_params(S + 1, max_iterations=1, state=synth.REFINE_ITER, use_APD=1, geom_consistency=1,
                     weak_peak_radius=4, rotate_time=2, ransac_threshold=0.01)
    a, b = pair(sc, p2, st, depths=sc["depth_gt"])
    assert a.weak_count() == b.weak_count() > 50
    _run_and_compare(a, b, 1)
    assert (b.get("weak_reliable") == 1).sum() > 0


@pytest.mark.parametrize("S", [12, 18])
def test_many_views_weak_path(S):
    """More than 9 and more than 16 source views (see tests/test_gpu_parity.py for the device kernels this selects)."""
    many_views_case(S, _pair, lambda sc, p: O.from_scene(sc, p, cls=O.Oracle))


def test_refine_init_and_generic_radius():
    """REFINE_INIT acceptance rule (cost must improve by 0.1) + radii that are not multiples of 5
    (generic tap loop) + use_radius off."""
    W, H, S = 80, 64, 2
    sc = synth.make_scene(W, H, S)
    p1 = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0)
    o = O.from_scene(sc, p1)
    o.upload_state(**first_pass_state(sc))
    o.run_patchmatch()
    st = second_pass_inputs(o, sc)
    rad = st["radius"].copy()
    rad[::7] = 7
    rad[3::11] = 10
    st["radius"] = rad
    p2 = make_params(S + 1, max_iterations=1, state=synth.REFINE_INIT, use_APD=1, use_detail=1, weak_peak_radius=6)
    a, b = _pair(sc, p2, st)
    _run_and_compare(a, b, 1)
    p3 = make_params(S + 1, max_iterations=1, state=synth.REFINE_INIT, use_APD=1, use_radius=0)
    a, b = _pair(sc, p3, st)
    _run_and_compare(a, b, 1)


def test_launch_geometry_covers_every_pixel_once():
    """block -> tile -> pixel map (csrc/dvp_stages.hpp): strips of 8 tile columns, ragged last strip,
    full and red/black launches, sizes around the tile and strip boundaries."""
    import ctypes
    L = ctypes.CDLL(E.lib()._name) if hasattr(E.lib(), "_name") else E.lib()
    L.emu_tile_map_check.restype = ctypes.c_longlong
    L.emu_tile_map_check.argtypes = [ctypes.c_int] * 4
    for (w, h) in ((1, 1), (16, 12), (63, 9), (64, 8), (65, 33), (511, 40), (512, 64), (513, 31), (1000, 70), (3104, 130)):
        assert L.emu_tile_map_check(w, h, 0, 0) == 0, (w, h, "full")
        for colour in (0, 1):
            assert L.emu_tile_map_check(w, h, 1, colour) == 0, (w, h, "half", colour)


@pytest.mark.parametrize("form", ["split", "split_lockstep_refine", "monolithic"])
@pytest.mark.parametrize("S", [3, 5, 9, 12])
def test_strong_update_forms_equal_the_oracle(form, S, monkeypatch):
    """split: evaluation of the bitwise-distinct planes of a pixel's 17 slots ((pixel, slot) items over the lanes on the
    engine, dvp_strong_eval_items), decisions through the slot -> source-slot table, refinement with every lane on its own
    (hypothesis, view) sequence; split_lockstep_refine: a pixel per lane in both (dvp_strong_eval, dvp_strong_refine)."""
    strong_update_forms_case(_pair, _run_and_compare, form, S, monkeypatch)


def strong_update_forms_case(_pair, _run_and_compare, form, S, monkeypatch):
    """The engine issues the strong update as three launches (dvp_strong_eval / _decide_vN / _refine: evaluator-only kernel,
    register-resident decisions, refinement with the exact early exits) or, with DVP_STRONG_SPLIT=0, as the one monolithic
    kernel; the emulation follows the same switch.  Both forms, every view-count bracket of the decision kernel, REFINE_INIT's
    write-back rule included: bit-identical to the oracle after every launch."""
    monkeypatch.setenv("DVP_STRONG_SPLIT", "0" if form == "monolithic" else "1")
    monkeypatch.setenv("DVP_REFINE_LANES", "0" if form == "split_lockstep_refine" else "1")   # dvp_strong_refine[_lanes]
    monkeypatch.setenv("DVP_EVAL_ITEMS", "0" if form == "split_lockstep_refine" else "1")     # engine: dvp_strong_eval[_items]
    W, H = 72, 56
    sc = synth.make_scene(W, H, S)
    for state in (synth.FIRST_INIT, synth.REFINE_INIT):
        p = make_params(S + 1, max_iterations=2, state=state, use_APD=0)
        st = first_pass_state(sc)
        if state == synth.REFINE_INIT:
            rng = np.random.default_rng(S)
            L = W * H
            n = np.tile(sc["normal_gt"], (L, 1)) + rng.normal(0, 0.05, (L, 3)).astype(np.float32)
            n /= np.linalg.norm(n, axis=1, keepdims=True)
            depth = (sc["depth_gt"][0].reshape(-1) * (1 + rng.normal(0, 0.02, L))).astype(np.float32)
            st["planes"] = np.concatenate([n, depth[:, None]], 1).astype(np.float32)
            st["views"] = rng.integers(1, 1 << S, L).astype(np.uint32)
        a, b = _pair(sc, p, st)
        _run_and_compare(a, b, 2, names=("planes", "costs", "selected_views", "view_weight"))


def test_line_test_closed_form_equals_the_stepping_walk():
    """bresenham_hits_edge (csrc/dvp_weak.hpp): the kernels' form jumps over blocks of eight steps whose bounding box holds no
    edge pixel, from the closed-form state of the reference's walk (BresenhamLine, APD.cu:267-311), instead of making every
    step.  Against the stepping definition on sparse, medium and dense edge maps: every pair of end points with
    |dx|, |dy| <= 40 around several centres (all the small slopes, both majors, the degenerate ones), long walks up to and
    beyond the step limit, walks that leave the image."""
    import ctypes
    from emul.emul import Emul, lib as emul_lib
    L = emul_lib()
    L.emu_line_test.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5
    L.emu_pack_edges.argtypes = [ctypes.c_void_p]
    rng = np.random.default_rng(5)
    for W, H, dens in ((300, 210, 0.002), (300, 210, 0.02), (640, 480, 0.0005), (200, 150, 0.3), (3100, 400, 0.001), (2400, 1800, 0.0002)):   # max_step = max(W, H) / 30 = 10, 10, 21, 6, 103, 80
        e = Emul(W, H, 2)
        edge = (rng.random((H, W)) < dens).astype(np.uint8)
        edge[H // 3, W // 4:W // 2] = 1          # a wall
        edge[H // 5:H // 2, 2 * W // 3] = 1
        e.upload_state(planes=np.zeros((H * W, 4), np.float32), views=np.zeros(H * W, np.uint32), weak=np.full(H * W, synth.STRONG, np.uint8),
                       edge=edge.reshape(-1), label=np.zeros(H * W, np.int32), radius=np.full(H * W, 5, np.int32))
        L.emu_pack_edges(e.h)
        n = hits = 0
        cases = []
        for cx, cy in ((W // 2, H // 2), (3, 4), (W - 2, H - 3), (W // 4 + 5, H // 3 + 2)):
            for dx in range(-40, 41, 1 if dens > 0.001 else 3):
                for dy in range(-40, 41, 1 if dens > 0.001 else 3):
                    cases.append((cx, cy, cx + dx, cy + dy))
        for _ in range(20000):
            ax, ay = int(rng.integers(0, W)), int(rng.integers(0, H))
            r = int(rng.integers(0, 4 * max(W, H) // 30))
            bx, by = ax + int(rng.integers(-r, r + 1)), ay + int(rng.integers(-r, r + 1))
            cases.append((ax, ay, bx, by))
        for ax, ay, bx, by in cases:
            if not (0 <= ax < W and 0 <= ay < H and 0 <= bx < W and 0 <= by < H):   # end points are image pixels (anchors, candidates)
                continue
            a = L.emu_line_test(e.h, ax, ay, bx, by, 0)
            b = L.emu_line_test(e.h, ax, ay, bx, by, 1)
            assert a == b, (W, H, dens, ax, ay, bx, by, a, b)
            n += 1
            hits += b
        assert hits > 50 and hits < n, (hits, n)
        e.close()
