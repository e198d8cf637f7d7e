"""GPU parity tests proper: the HIP engine, driven through the C ABI (include/dvp_mvs.h), against
the CPU oracle on the same seeded inputs.  Bar: bit-exact — integer buffers exactly, float buffers
bitwise (NaN == NaN).  The north_star's tolerance for depth/normal maps is 1e-3 relative; it is
also asserted explicitly on the final maps (trivially implied by bitwise equality)."""
import numpy as np
import pytest

from conftest import (pkg, synth, make_params, count_diff, stage_sequence, CHECKED, first_pass_state,
                      second_pass_inputs)
from oracle import oracle as O

pytestmark = pytest.mark.gpu
REL_TOL = 1e-3   # north_star: depth/normal maps within 1e-3 relative of the reference path


def capi():
    return pkg("capi")


def _pair(scene, params, state, seed=1234, sampler=0, depths=None):
    a = O.from_scene(scene, params, seed=seed, sampler=sampler, depths=depths)
    b = capi().from_scene(scene, params, seed=seed, sampler=sampler, depths=depths)
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


def test_library_is_the_hip_engine():
    """the product path is the HIP shared library (no CPU fallback exists)"""
    L = capi().lib()
    for name in capi().EXPORTS:
        assert hasattr(L, name)


def test_cost_vectors_kat():
    """ComputeMultiViewCostVectorOld on random (pixel, plane) pairs incl. planes that project
    outside the source images and border pixels (clamp addressing)."""
    W, H, S = 160, 120, 5
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1)
    a, b = _pair(sc, p, first_pass_state(sc))
    rng = np.random.default_rng(7)
    n = 4096
    px = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
    px[:64] = [[0, 0], [W - 1, H - 1], [0, H - 1], [W - 1, 0]] * 16
    depth = rng.uniform(1.5, 7.8, n).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm[:, 2] = -np.abs(nrm[:, 2]) - 0.3
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    K = sc["cameras"][0]["K"]
    X = depth * (px[:, 0] - K[2]) / K[0]
    Y = depth * (px[:, 1] - K[5]) / K[4]
    d = -(nrm[:, 0] * X + nrm[:, 1] * Y + nrm[:, 2] * depth)
    planes = np.concatenate([nrm, d[:, None]], 1).astype(np.float32)
    ca = a.eval_cost_vectors(px, planes)
    cb = b.eval_cost_vectors(px, planes)
    assert count_diff(ca, cb) == 0
    assert (ca < 2.0).mean() > 0.3   # the test really evaluates patches


@pytest.mark.parametrize("W,H,S,sampler", [(128, 96, 3, 0), (131, 67, 5, 1), (70, 33, 1, 0)])
def test_first_pass_stage_by_stage(W, H, S, sampler):
    """odd sizes (ragged tiles, odd H), S = 1 and S = 5, both samplers"""
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
    a, b = _pair(sc, p, first_pass_state(sc), sampler=sampler)
    _run_and_compare(a, b, 2)


def test_full_run_matches_and_converges():
    W, H, S = 192, 128, 3
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=3, state=synth.FIRST_INIT, use_APD=0)
    a, b = _pair(sc, p, first_pass_state(sc))
    a.run_patchmatch()
    b.run_patchmatch()
    pa, pb = a.get("planes"), b.get("planes")
    assert count_diff(pa, pb) == 0
    for n in ("selected_views", "weak_info", "radius"):
        assert count_diff(a.get(n), b.get(n)) == 0
    # tolerance form of the same statement
    da, db = pa[:, 3], pb[:, 3]
    ok = np.isfinite(da) & (da != 0)
    assert np.all(np.abs(da[ok] - db[ok]) <= REL_TOL * np.abs(da[ok]))
    assert np.all(np.abs(pa[ok, :3] - pb[ok, :3]) <= REL_TOL)
    # and it is a sensible reconstruction: median relative depth error vs ground truth
    gt = sc["depth_gt"][0].reshape(-1)
    m = np.zeros((H, W), bool)
    m[8:-8, 8:-8] = True
    rel = np.abs(db - gt) / gt
    assert np.median(rel[m.reshape(-1)]) < 2e-3


@pytest.mark.parametrize("anchors", ["table", "one_wave", "alloc_fail", "per_item"])
@pytest.mark.parametrize("images", ["8bit", "float"])
def test_two_pass_weak_path_with_geom(images, anchors, monkeypatch):
    """`anchors`: the reference side of the anchor sub-patches from the pass' table (built once, before the first weak
    update of the pass) or formed per (view, anchor, plane) item as the source text does (DVP_WEAK_ANCHOR_TAB=0).  With the
    table the update runs as seven launches (dvp_weak_phased.hpp: "table"); "one_wave" (DVP_WEAK_PHASED=0) and "alloc_fail"
    (the hand-over buffers do not fit: DVP_TEST_WEAK_PHASE_ALLOC_FAIL) keep a pixel's whole update in one wave; all the same bits.
    `images`: the synthetic grey levels are integers (8-bit files) -> the weak update reads the byte planes
    (Dev::images8); scaled to non-integers it reads the float planes.  Both against the oracle."""
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
    g = capi().from_scene(sc, p1)
    g.upload_state(**first_pass_state(sc))
    g.run_patchmatch()
    st = second_pass_inputs(g, sc)
    weak = st["weak"].reshape(H, W)
    weak[sc["flat"] & (weak == synth.STRONG)] = synth.WEAK
    weak[:6, :] = synth.UNKNOWN
    st["weak"] = weak.reshape(-1)
    p2 = make_params(S + 1, max_iterations=2, state=synth.REFINE_ITER, use_APD=1, geom_consistency=1,
                     weak_peak_radius=4, rotate_time=2, ransac_threshold=0.01)
    a, b = _pair(sc, p2, st, depths=sc["depth_gt"])
    assert b.image_format() == (1 if images == "8bit" else 0)
    assert a.weak_count() == b.weak_count() > 50
    _run_and_compare(a, b, 2)
    assert (b.get("weak_reliable") == 1).sum() > 0


@pytest.mark.parametrize("form,wpr", [("passes", None), ("fused", None), ("passes", 9), ("passes", 40)])
@pytest.mark.parametrize("geom", [0, 1])
def test_fused_sweeps_equal_the_two_launches(geom, form, wpr, monkeypatch):
    """`form`: the fused launch site as view-compacted passes (prepare / evaluate per view over the pixels that selected it /
    decide, dvp_strong.hpp: sweep_*) or as the one per-pixel kernel (DVP_SWEEP_SPLIT=0); `wpr`: weak_peak_radius — the
    central window that is evaluated first grows with it (9 -> 21 slots, 40 -> the whole line, no second stage).
    dvp_run_patchmatch issues DepthToWeak + LocalRefine (APD.cu:4502-4505) as ONE launch; dvp_run_stage keeps
    them separate.  Both must leave the same bits (and both equal the oracle's two functions, checked by the
    full-run tests): border pixels, out-of-range sweep slots and the geometric term included."""
    W, H, S = 150, 97, 4
    monkeypatch.setenv("DVP_SWEEP_SPLIT", "2" if form == "passes" else "0")   # 2: the passes also where the geometric term is off (the default takes the fused kernel there)
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0, geom_consistency=geom)
    if wpr is not None:
        p["weak_peak_radius"] = wpr
    p["depth_min"] = np.float32(3.2)   # part of the +-30 disparity sweep leaves the depth range
    st = first_pass_state(sc)
    dm = sc["depth_gt"] if geom else None
    whole = capi().from_scene(sc, p, depths=dm)
    steps = capi().from_scene(sc, p, depths=dm)
    ora = O.from_scene(sc, p, depths=dm)
    for x in (whole, steps, ora):
        x.upload_state(**st)
    whole.run_patchmatch()
    ora.run_patchmatch()
    for stg, it, col in stage_sequence(2):
        steps.run_stage(stg, it, col)
    for n in ("planes", "weak_info", "radius", "costs", "selected_views"):
        assert count_diff(whole.get(n), steps.get(n)) == 0, n
        assert count_diff(whole.get(n), ora.get(n)) == 0, n
    t = whole.timings()
    assert t["stage_launches"]["depth_to_weak"] == 1 and t["stage_launches"]["local_refine"] == 0
    assert steps.timings()["stage_launches"]["local_refine"] == 1


@pytest.mark.parametrize("band_gb", ["0.0004", "0.006"])
def test_sweep_passes_in_bands_of_rows_leave_the_same_bits(band_gb, monkeypatch):
    """DVP_SWEEP_BAND_GB: the sweep passes' cost records hold a band of rows and the passes run band after band (apd sets it: the
    records of a whole 25-Mpx view are 67 GB of fresh device memory).  Every pass is per pixel: same results as with the whole
    image in one go, and as the oracle.  150 x 97, S = 4: 17 MB of records -> 7 bands of one 14-row tile row / 3 bands of three at these limits."""
    W, H, S = 150, 97, 4
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0, geom_consistency=1)
    p["depth_min"] = np.float32(3.2)
    st = first_pass_state(sc)
    monkeypatch.setenv("DVP_SWEEP_SPLIT", "2")
    whole = capi().from_scene(sc, p, depths=sc["depth_gt"])
    monkeypatch.setenv("DVP_SWEEP_BAND_GB", band_gb)
    bands = capi().from_scene(sc, p, depths=sc["depth_gt"])
    ora = O.from_scene(sc, p, depths=sc["depth_gt"])
    for x in (whole, bands, ora):
        x.upload_state(**st)
        x.run_patchmatch()
    for n in ("planes", "weak_info", "radius", "costs", "selected_views"):
        assert count_diff(whole.get(n), bands.get(n)) == 0, n
        assert count_diff(bands.get(n), ora.get(n)) == 0, n


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_find_nearest_strong_ring_order(seed):
    from test_emul_parity import find_nearest_strong_case
    find_nearest_strong_case(seed, _pair)


@pytest.mark.parametrize("form", ["per_lane", "wave"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gen_neighbours_search_forms_equal_the_oracle(form, seed, monkeypatch):
    """GenNeighbours on the GPU — dvp_gen_neighbours_list (directional search, one lane per WEAK pixel, list in LDS) or
    dvp_gen_neighbours_search (DVP_GN_WAVE=1: one wave per pixel), then label extension + plane fit in
    dvp_gen_neighbours_fit — on maps where directions need tens of tries and the extension meets duplicates."""
    from test_emul_parity import gen_neighbours_case
    monkeypatch.setenv("DVP_GN_WAVE", "1" if form == "wave" else "0")
    gen_neighbours_case(seed, _pair)


@pytest.mark.parametrize("form", ["wave", "per_lane"])
def test_ransac_fit_plane_forms_equal_the_oracle(form, monkeypatch):
    """RANSACToGetFitPlane on the GPU: one wave per WEAK pixel with a lane per draw (default) and one lane per WEAK pixel"""
    from test_emul_parity import many_views_case, gen_neighbours_case
    monkeypatch.setenv("DVP_RANSAC_WAVE", "1" if form == "wave" else "0")
    many_views_case(5, _pair, lambda sc, p: capi().from_scene(sc, p))
    gen_neighbours_case(1, _pair)


@pytest.mark.parametrize("form", ["split", "split_lockstep_refine", "monolithic"])
@pytest.mark.parametrize("S", [3, 5, 9, 12])
def test_strong_update_forms_equal_the_oracle(form, S, monkeypatch):
    """the three launches with dvp_strong_refine_lanes (every lane on its own (hypothesis, view) sequence) or
    dvp_strong_refine (the wave in lock step), and the monolithic kernel: see test_emul_parity"""
    from test_emul_parity import strong_update_forms_case
    strong_update_forms_case(_pair, _run_and_compare, form, S, monkeypatch)


@pytest.mark.parametrize("S", [12, 18])
def test_many_views_weak_path(S):
    """More than 9 and more than 16 source views: the 16- and 32-view instantiations of the strong update, several
    batches of (view, plane) pairs per phase of the weak update, view selection over 12 / 18 candidates."""
    from test_emul_parity import many_views_case
    many_views_case(S, _pair, lambda sc, p: capi().from_scene(sc, p))


def test_refine_init_generic_radius():
    W, H, S = 80, 64, 2
    sc = synth.make_scene(W, H, S)
    p1 = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0)
    g = capi().from_scene(sc, p1)
    g.upload_state(**first_pass_state(sc))
    g.run_patchmatch()
    st = second_pass_inputs(g, sc)
    rad = st["radius"].copy()
    rad[::7] = 7
    rad[3::11] = 10
    st["radius"] = rad
    p2 = make_params(S + 1, max_iterations=1, state=synth.REFINE_INIT, use_APD=1, use_detail=1, weak_peak_radius=6)
    a, b = _pair(sc, p2, st)
    _run_and_compare(a, b, 1)


def test_nine_views_and_determinism():
    """S = 9 (the candidate buffer is [pixel][S][8]; the reference's 4-view stride would alias) and
    run-to-run determinism of the GPU path: same seed -> identical bits, other seed -> differs."""
    W, H, S = 96, 64, 9
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0)
    outs = []
    for seed in (5, 5, 6):
        g = capi().from_scene(sc, p, seed=seed)
        g.upload_state(**first_pass_state(sc))
        g.run_patchmatch()
        outs.append(g.get("planes"))
    assert count_diff(outs[0], outs[1]) == 0
    assert count_diff(outs[0], outs[2]) > 0
    a = O.from_scene(sc, p, seed=5)
    a.upload_state(**first_pass_state(sc))
    a.run_patchmatch()
    assert count_diff(a.get("planes"), outs[0]) == 0


def test_error_paths():
    c = capi()
    with pytest.raises(c.DvpError):
        c.Context(64, 64, 40)          # > 32 images (APD.cpp:1083-1086)
    g = c.Context(64, 48, 3)
    with pytest.raises(c.DvpError):    # kernels before set_params
        g.run_stage("random_init")
    p = make_params(3, geom_consistency=1)
    g.set_params(p)
    with pytest.raises(c.DvpError):    # geom on, no depth maps
        g.run_stage("depth_to_weak")
    # use_edge=false: the reference's legacy branch reads an uninitialised positions[] (APD.cu:2036 vs 2559-2563);
    # there is no result to reproduce, the engine says so instead of inventing one
    with pytest.raises(c.DvpError, match="positions"):
        g.set_params(make_params(3, use_edge=0))
    # the round-4 entry points: required maps missing
    z, n, v = np.zeros((24, 32), np.float32), np.zeros((24, 32, 3), np.float32), np.zeros((24, 32), np.uint32)
    with pytest.raises(c.DvpError, match="required"):
        g.upload_state_rescaled(32, 24, None, n, v)
    with pytest.raises(c.DvpError, match="required"):
        g.upload_state_rescaled(0, 24, z, n, v)
    L = g.L
    assert L.dvp_download_maps(g.h, None, None, None, None, None) != 0 and b"required" in L.dvp_last_error(g.h)
    g.upload_state_rescaled(32, 24, z, n, v)      # and the context is still usable


def test_size_independent_properties_at_bench_size():
    """BASELINE cfg2 size (3104x2064, S=5): too big for the oracle, so check properties:
    determinism of the cost kernel, costs in [0,2], planes face the camera, depth in range,
    and error vs analytic ground truth after 2 iterations."""
    W, H, S = 3104, 2064, 5
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
    g = capi().from_scene(sc, p)
    g.upload_state(**first_pass_state(sc))
    g.run_patchmatch()
    planes = g.get("planes")
    depth = planes[:, 3].reshape(H, W)
    inner = depth[16:-16, 16:-16]
    assert np.isfinite(inner).all()
    gt = sc["depth_gt"][0][16:-16, 16:-16]
    rel = np.abs(inner - gt) / gt
    assert np.median(rel) < 5e-3
    wi = g.get("weak_info")
    assert set(np.unique(wi)).issubset({0, 1, 2})
    views = g.get("selected_views")
    assert views.max() < (1 << S)
    t = g.timings()
    assert t["total_ms"] > 0 and t["stage_launches"]["strong_update"] == 4


def test_golden_fixtures_on_gpu():
    """the committed vectors (tests/golden/golden_small.npz) through the C ABI"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))
    W, H, S = int(g["W"]), int(g["H"]), int(g["S"])
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
    b = capi().from_scene(sc, p, seed=int(g["seed"]))
    b.upload_state(**first_pass_state(sc))
    assert count_diff(b.eval_cost_vectors(g["kat_px"], g["kat_planes"]), g["kat_costs"]) == 0
    b.run_patchmatch()
    planes, views, weak, radius = b.download_state()
    assert count_diff(planes, g["planes"]) == 0
    assert np.array_equal(views, g["selected_views"]) and np.array_equal(weak, g["weak_info"])


def test_reset_state_equals_fresh_context():
    W, H, S = 96, 64, 3
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0)
    g = capi().from_scene(sc, p, seed=9)
    g.upload_state(**first_pass_state(sc))
    g.run_patchmatch()
    first = g.get("planes")
    g.reset_state()
    g.run_patchmatch()
    assert count_diff(first, g.get("planes")) == 0


def _rescale_reference_rule(src, tw, th):
    """RescaleMatToTargetSize as the source text has it (APD.cpp:1773-1795): nearest neighbour, the ROW index divided by
    the WIDTH ratio and the column index by the height ratio (both binary32), truncated; outside the source: zero
    (the reference leaves those pixels uninitialised; defined as zero here).  Written from the source, array-wise."""
    sh, sw = src.shape[:2]
    scale_x = np.float32(tw) / np.float32(sw)
    scale_y = np.float32(th) / np.float32(sh)
    o_r = (np.arange(th, dtype=np.float32) / scale_x).astype(np.int64)
    o_c = (np.arange(tw, dtype=np.float32) / scale_y).astype(np.int64)
    ok = (o_r[:, None] < sh) & (o_c[None, :] < sw)
    out = np.zeros((th, tw) + src.shape[2:], src.dtype)
    out[ok] = src[np.minimum(o_r, sh - 1)[:, None], np.minimum(o_c, sw - 1)[None, :]][ok]
    return out


@pytest.mark.parametrize("W,H,sw,sh", [(96, 64, 48, 32), (101, 67, 50, 33), (90, 70, 45, 33), (97, 61, 51, 30)])
def test_upload_state_rescaled_is_the_reference_rescale(W, H, sw, sh):
    """dvp_upload_state_rescaled (the coarser pyramid level's maps up-sampled on the device) against the reference's rule
    read from the source: planes = (normal, depth), selected views, pixel states (+ WEAK count), radius with the
    UNKNOWN -> strong_radius rule (APD.cpp:1660-1666); sizes whose ratios differ per axis (the swapped scale factors
    matter, and some target pixels fall outside the source)."""
    rng = np.random.default_rng(W * 131 + sh)
    depth = rng.uniform(1.0, 9.0, (sh, sw)).astype(np.float32)
    normal = rng.normal(size=(sh, sw, 3)).astype(np.float32)
    views = rng.integers(0, 2 ** 9, (sh, sw)).astype(np.uint32)
    weak = rng.integers(0, 3, (sh, sw)).astype(np.uint8)
    radius = rng.integers(1, 12, (sh, sw)).astype(np.int32)
    g = capi().Context(W, H, 4)
    g.upload_state_rescaled(sw, sh, depth, normal, views, weak, radius, radius_fallback=7)
    planes, gv, gw, gr = g.download_state()
    e_w = _rescale_reference_rule(weak, W, H)
    e_r = _rescale_reference_rule(radius, W, H)
    e_r[e_w == synth.UNKNOWN] = 7
    e_p = np.concatenate([_rescale_reference_rule(normal, W, H), _rescale_reference_rule(depth, W, H)[..., None]], -1)
    assert count_diff(planes.reshape(H, W, 4), e_p) == 0
    assert (gv.reshape(H, W) == _rescale_reference_rule(views, W, H)).all()
    assert (gw.reshape(H, W) == e_w).all()
    assert (gr.reshape(H, W) == e_r).all()
    assert g.weak_count() == int((e_w == synth.WEAK).sum())
    nm = g.get("neighbours_map").reshape(H, W)
    assert (nm[e_w == synth.WEAK] == np.arange(int((e_w == synth.WEAK).sum()))).all()   # running index, raster order (APD.cpp:1182-1193)
    # without pixel states / radius map: every pixel STRONG, radius untouched
    before = g.get("radius").copy()
    g.upload_state_rescaled(sw, sh, depth, normal, views)
    _, _, gw2, gr2 = g.download_state()
    assert (gw2 == synth.STRONG).all() and (gr2 == before.reshape(-1)).all()


def test_download_maps_is_the_drivers_unpack_loop():
    """dvp_download_maps against the loop of the reference's driver (main.cpp:300-309): depth = plane.w inside
    [depth_min, depth_max], else 0 and the state UNKNOWN; normal = plane.xyz; a NaN depth passes both comparisons and is
    kept.  The device state itself is not modified."""
    W, H, S = 96, 64, 3
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0)
    g = capi().from_scene(sc, p, seed=5)
    rng = np.random.default_rng(3)
    planes = rng.normal(size=(H * W, 4)).astype(np.float32)
    planes[:, 3] = rng.uniform(0.5 * float(p["depth_min"]), 1.5 * float(p["depth_max"]), H * W)
    planes[::97, 3] = np.nan
    planes[5, 3] = p["depth_min"]
    planes[6, 3] = p["depth_max"]
    weak = rng.integers(0, 3, H * W).astype(np.uint8)
    g.upload_state(planes=planes, weak=weak)
    depth, normal, views, st, radius = g.download_maps()
    usable = ~((planes[:, 3] < p["depth_min"]) | (planes[:, 3] > p["depth_max"]))
    assert usable[5] and usable[6] and usable[0::97].all() and (~usable).sum() > 100
    assert count_diff(depth, np.where(usable, planes[:, 3], np.float32(0))) == 0
    assert count_diff(normal, planes[:, :3]) == 0
    assert (st == np.where(usable, weak, synth.UNKNOWN)).all()
    p2, v2, w2, r2 = g.download_state()
    assert (w2 == weak).all() and (v2 == views).all() and (r2 == radius).all() and count_diff(p2, planes) == 0


def test_download_maps_in_two_steps_survives_the_next_view():
    """dvp_download_maps_begin stages the maps on the device; the context is then given another view's state (and a finished
    pass of it) before dvp_download_maps_finish fetches the FIRST view's maps from another thread — what the driver's
    background job does.  Equal to the one-call download taken before; a finish without a begin is refused; the depth map
    also lands in a device buffer of the caller's."""
    import threading
    import torch
    W, H, S = 96, 64, 3
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0)
    g = capi().from_scene(sc, p, seed=5)
    state = first_pass_state(sc)
    g.upload_state(**state)
    g.run_patchmatch()
    want = g.download_maps()
    assert g.L.dvp_download_maps_finish(g.h, None, None, None, None, None) != 0 and b"without" in g.L.dvp_last_error(g.h)
    dev = torch.zeros(W * H, dtype=torch.float32, device="cuda:0")
    g.download_maps_begin(dev.data_ptr())
    # the next view on the same context: other planes, other states, another pass
    rng = np.random.default_rng(9)
    other = dict(state)
    other["planes"] = rng.normal(size=(H * W, 4)).astype(np.float32)
    other["weak"] = rng.integers(0, 3, H * W).astype(np.uint8)
    other["views"] = rng.integers(1, 1 << S, H * W).astype(np.uint32)
    g.upload_state(**other)
    g.set_seed(99)
    g.run_patchmatch()
    got = {}
    t = threading.Thread(target=lambda: got.setdefault("maps", g.download_maps_finish()))
    t.start()
    t.join()
    for a, b in zip(want, got["maps"]):
        assert count_diff(a, b) == 0
    assert count_diff(dev.cpu().numpy(), want[0]) == 0
    after = g.download_maps()
    assert count_diff(after[0], want[0]) > 0      # (the second view's own maps are something else)


def test_staged_maps_that_are_never_fetched_do_not_hang_the_context(monkeypatch):
    """a second dvp_download_maps_begin without a dvp_download_maps_finish fails after a bounded wait instead of
    blocking for ever, and a context whose staged maps were never fetched can still be destroyed."""
    import time
    monkeypatch.setenv("DVP_DOWNLOAD_WAIT_S", "1")
    W, H, S = 64, 48, 2
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=1, state=synth.FIRST_INIT, use_APD=0)
    g = capi().from_scene(sc, p, seed=5)
    g.upload_state(**first_pass_state(sc))
    g.run_patchmatch()
    g.download_maps_begin()
    t0 = time.time()
    assert g.L.dvp_download_maps_begin(g.h, None) != 0 and b"never fetched" in g.L.dvp_last_error(g.h)
    assert time.time() - t0 < 10
    maps = g.download_maps_finish()          # the staged maps are still there
    assert np.isfinite(maps[0]).any()
    g.download_maps_begin()
    t0 = time.time()
    g.close()                                # staged, never fetched: abandoned
    assert time.time() - t0 < 10


def test_reserved_buffers_change_nothing():
    """dvp_ctx_reserve (the driver's helper thread calls it on the context it prepares for the next level): a context whose
    optional buffers — split strong update, sweep passes, anchor table, weak hand-over — were allocated ahead, with room for
    fewer or more WEAK pixels than the pass has, gives the bits of one that allocates them inside its launches."""
    from test_baseline_configs import _two_pass
    W, H, S, iters = 160, 120, 5, 2
    sc = synth.make_scene(W, H, S)
    made = []

    def make(reserve):
        def f(scene, p):
            g = capi().from_scene(scene, p)
            if reserve is not None:
                g.reserve(weak_pixels=reserve, flags=3)
            made.append(g)
            return g
        return f
    outs = []
    for reserve in (None, 50, W * H):
        e = _two_pass(make(reserve), sc, S, iters, 0.08)
        assert e.weak_count() > 500
        e.run_patchmatch()
        outs.append({n: e.get(n) for n in ("planes", "costs", "selected_views", "weak_info", "radius", "fit_planes", "view_weight")})
        e.close()
    for other in outs[1:]:
        for n, a in outs[0].items():
            assert count_diff(a, other[n]) == 0, n


def test_golden_weak_pass_engine():
    """the committed REFINE_ITER / weak-path fixture through the C ABI"""
    from test_oracle_kat import _golden_weak_pass
    _golden_weak_pass(lambda sc, p, seed, dep: capi().from_scene(sc, p, seed=seed, depths=dep))


def test_split_strong_update_falls_back_when_its_buffer_does_not_fit(monkeypatch):
    """The split strong update keeps 17 x S floats per pixel of one colour (7.8 GB at 6208x4128, S = 9).  When that block
    cannot be allocated the launch site must carry on with the monolithic kernel — same bits — instead of failing the
    pass.  DVP_TEST_SPLIT_ALLOC_FAIL makes the allocation report failure."""
    monkeypatch.setenv("DVP_TEST_SPLIT_ALLOC_FAIL", "1")
    W, H, S = 80, 56, 5
    sc = synth.make_scene(W, H, S)
    p = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
    a, b = _pair(sc, p, first_pass_state(sc))
    a.run_patchmatch()
    b.run_patchmatch()
    for n in ("planes", "costs", "selected_views", "view_weight", "weak_info"):
        assert count_diff(a.get(n), b.get(n)) == 0, n
    launches = b.timings()["stage_launches"]
    assert launches["strong_update"] == 4
