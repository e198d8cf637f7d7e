"""Edge cases of the per-view path, oracle vs engine bit for bit: images smaller than one wave row,
odd sizes (incl. the odd-H launch quirk of APD.cu:4423), the maximum of 32 images, texture-less
images (every NCC hits the variance floor), passes whose pixels are all WEAK or all UNKNOWN, and the
error returns of the C ABI for inputs the reference rejects (APD.cpp:1083-1086).
CPU: host emulation of the kernels; GPU (-m gpu): the HIP library through the C ABI."""
import numpy as np
import pytest

from conftest import pkg, synth, make_params, count_diff, first_pass_state
from oracle import oracle as O
from tests.emul import emul as E

CHECK = ("planes", "costs", "selected_views", "weak_info", "radius", "view_weight", "neighbours", "weak_reliable")


def scene_with_views(W, H, S, seed=7):
    """synth.make_scene has 9 ring positions; more views reuse the ring at other baselines."""
    sc = synth.make_scene(W, H, min(S, 9), seed=seed)
    if S <= 9:
        return sc
    px_world = 4.0 / (0.9 * W)
    images, cams, depths = list(sc["images"]), list(sc["cameras"]), list(sc["depth_gt"])
    k = 0
    while len(images) < S + 1:
        a, b = synth._RING[k % len(synth._RING)]
        base = 0.15 + 0.05 * (k // len(synth._RING))
        cam = synth.make_camera(W, H, (base * a, base * b, 0.01 * k), R=synth._rot(0.02 * a, -0.03 * b, 0.1 * (k % 5 - 2)))
        img, dep, _, _ = synth.render_view(W, H, cam, px_world, with_step=True, with_flat=True)
        images.append(img)
        depths.append(dep)
        cams.append(cam)
        k += 1
    out = dict(sc)
    out["images"] = np.stack(images).astype(np.float32)
    out["depth_gt"] = np.stack(depths).astype(np.float32)
    c = np.zeros(S + 1, dtype=synth.CAMERA_DTYPE)
    for i, cam in enumerate(cams):
        c[i] = cam
    out["cameras"] = c
    return out


def cases():
    out = []
    # (name, W, H, S, params overrides, state mutator)
    out.append(("tiny_16x12_one_source", 16, 12, 1, dict(max_iterations=1), None))
    out.append(("narrower_than_a_wave_odd", 33, 21, 2, dict(max_iterations=2), None))
    out.append(("odd_height_quirk_H33", 40, 33, 2, dict(max_iterations=1), None))
    out.append(("max_images_32", 48, 40, 31, dict(max_iterations=1), None))

    def flat_images(sc, st, p):
        sc["images"][:] = 100.0
    out.append(("textureless", 40, 34, 3, dict(max_iterations=1), flat_images))

    def all_weak(sc, st, p):
        L = sc["width"] * sc["height"]
        gt = sc["depth_gt"][0].reshape(-1)
        st["planes"] = np.concatenate([np.tile(sc["normal_gt"], (L, 1)), gt[:, None]], 1).astype(np.float32)
        st["views"] = np.full(L, 3, np.uint32)
        st["weak"] = np.full(L, synth.WEAK, np.uint8)
    out.append(("all_pixels_weak", 48, 36, 2, dict(max_iterations=1, state=synth.REFINE_ITER, use_APD=1), all_weak))

    def all_unknown(sc, st, p):
        all_weak(sc, st, p)
        st["weak"][:] = synth.UNKNOWN
    out.append(("all_pixels_unknown", 48, 36, 2, dict(max_iterations=1, state=synth.REFINE_INIT, use_APD=1), all_unknown))

    def no_prior_flags(sc, st, p):
        all_weak(sc, st, p)
        st["weak"][:] = synth.STRONG
        st["weak"][sc["flat"].reshape(-1)] = synth.WEAK
    out.append(("priors_off", 64, 48, 3, dict(max_iterations=1, state=synth.REFINE_ITER, use_APD=1, use_limit=0, use_label=0,
                                              use_radius=0, geom_consistency=1), no_prior_flags))
    return out


def run_case(case, make_b):
    name, W, H, S, over, mutate = case
    sc = scene_with_views(W, H, S)
    sc["images"] = sc["images"].copy()
    p = make_params(S + 1, **dict(dict(state=synth.FIRST_INIT, use_APD=0), **over))
    st = first_pass_state(sc)
    if mutate:
        mutate(sc, st, p)
    depths = sc["depth_gt"] if p["geom_consistency"] else None
    a = O.from_scene(sc, p, seed=99, depths=depths)
    b = make_b(sc, p, 99, depths)
    a.upload_state(**st)
    b.upload_state(**st)
    a.run_patchmatch()
    b.run_patchmatch()
    for key in CHECK:
        assert count_diff(a.get(key), b.get(key)) == 0, (name, key)
    if name == "textureless":
        c = a.get("costs")   # every NCC is at the variance floor (2); where no view gets weight the
        assert np.all((c == 2.0) | np.isnan(c))   # weighted mean is 0/0 = NaN, as in the reference (APD.cu:2534-2554)
    return a


@pytest.mark.parametrize("case", cases(), ids=lambda c: c[0])
def test_edge_cases_emulated_kernels(case):
    run_case(case, lambda sc, p, seed, dep: O.from_scene(sc, p, seed=seed, depths=dep, cls=E.Emul))


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases(), ids=lambda c: c[0])
def test_edge_cases_gpu(case):
    capi = pkg("capi")
    run_case(case, lambda sc, p, seed, dep: capi.from_scene(sc, p, seed=seed, depths=dep))


def test_rejected_inputs_cpu_side():
    """Argument checks that need no device: > 32 images (APD.cpp:1083-1086), < 2 images, sizes that
    do not fit the short2 pixel coordinates, null out-pointer."""
    import ctypes
    lib = pkg("capi").lib()
    ctx = ctypes.c_void_p()
    for (w, h, ni) in ((64, 48, 33), (64, 48, 1), (0, 48, 3), (64, -1, 3), (40000, 48, 3), (30000, 30000, 3)):
        assert lib.dvp_ctx_create(0, w, h, ni, ctypes.byref(ctx)) != 0
        assert not ctx.value
        assert b"dvp_ctx_create" in lib.dvp_last_error(None)
    assert lib.dvp_ctx_create(0, 64, 48, 3, None) != 0
