"""Randomised configuration sweep: image sizes, number of views, flags, radii, pixel-state mixes,
samplers and seeds are drawn from a seeded generator; for every draw the engine must reproduce
the oracle bit for bit after a full RunPatchMatch.  CPU: host emulation of the kernels; GPU
(-m gpu): the HIP library through the C ABI."""
import os

import numpy as np
import pytest

from conftest import pkg, synth, make_params, count_diff, first_pass_state
from oracle import oracle as O
from tests.emul import emul as E


def draw_config(rng):
    W = int(rng.integers(40, 120))
    H = int(rng.integers(34, 90))
    S = int(rng.integers(1, 6))
    # every draw has its own rotated rig (per-view R, K, centre); half of them with a skew K[1] != 0, which only
    # ProjectonCamera_cu reads (geometric term, random normals) while ComputeHomography / Get3DPoint ignore it
    sc = synth.make_scene(W, H, S, seed=int(rng.integers(0, 1 << 30)), skew=float(rng.choice([0.0, 0.4, 1.5])))
    state = int(rng.choice([synth.FIRST_INIT, synth.REFINE_INIT, synth.REFINE_ITER]))
    geom = int(state == synth.REFINE_ITER and rng.random() < 0.7)
    p = make_params(S + 1, max_iterations=int(rng.integers(1, 3)), state=state, use_APD=int(state != synth.FIRST_INIT),
                    geom_consistency=geom, use_limit=int(rng.random() < 0.7), use_label=int(rng.random() < 0.6),
                    use_radius=int(rng.random() < 0.8), use_detail=int(rng.random() < 0.3),
                    weak_peak_radius=int(rng.choice([2, 4, 6])), rotate_time=int(rng.choice([1, 2, 4])),
                    top_k=int(rng.integers(1, 5)), ransac_threshold=float(rng.choice([0.005, 0.01, 0.00875])))
    L = W * H
    st = first_pass_state(sc)
    if state != synth.FIRST_INIT:
        # a plausible previous-pass state: noisy ground-truth planes, random view masks, WEAK blobs
        gt = sc["depth_gt"][0].reshape(-1)
        n = np.tile(sc["normal_gt"], (L, 1)) + rng.normal(0, 0.05, (L, 3)).astype(np.float32)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        depth = (gt * (1 + rng.normal(0, 0.01, L))).astype(np.float32)
        depth[rng.random(L) < 0.02] = 0.0
        st["planes"] = np.concatenate([n, depth[:, None]], 1).astype(np.float32)
        st["views"] = rng.integers(0, 1 << S, L).astype(np.uint32)
        weak = np.full((H, W), synth.STRONG, np.uint8)
        for _ in range(int(rng.integers(1, 5))):
            y0, x0 = int(rng.integers(6, H - 14)), int(rng.integers(6, W - 14))
            weak[y0:y0 + int(rng.integers(3, 12)), x0:x0 + int(rng.integers(3, 14))] = synth.WEAK
        weak[rng.random((H, W)) < 0.02] = synth.UNKNOWN
        weak[depth.reshape(H, W) == 0] = synth.UNKNOWN
        st["weak"] = weak.reshape(-1)
        rad = np.full(L, 5, np.int32)
        rad[rng.random(L) < 0.1] = 10
        rad[rng.random(L) < 0.02] = 7
        st["radius"] = rad
    depths = sc["depth_gt"] if geom else None
    return sc, p, st, depths, int(rng.integers(0, 2)), int(rng.integers(0, 1 << 40))


def run_pair(make_b, rng):
    sc, p, st, depths, sampler, seed = draw_config(rng)
    a = O.from_scene(sc, p, seed=seed, sampler=sampler, depths=depths)
    b = make_b(sc, p, seed, sampler, depths)
    a.upload_state(**st)
    b.upload_state(**st)
    a.run_patchmatch()
    b.run_patchmatch()
    for name in ("planes", "costs", "selected_views", "weak_info", "radius", "view_weight", "neighbours", "weak_reliable"):
        nd = count_diff(a.get(name), b.get(name))
        assert nd == 0, (name, nd, sc["width"], sc["height"], dict(zip(p.dtype.names, p.tolist())))


@pytest.mark.parametrize("case", range(8))
def test_random_configs_emulated_kernels(case):
    rng = np.random.default_rng(1000 + case)
    run_pair(lambda sc, p, seed, smp, dep: O.from_scene(sc, p, seed=seed, sampler=smp, depths=dep, cls=E.Emul), rng)


# DVP_RANDOM_CASES=N widens the sweep for soak runs (default: 50 cases on a GPU box — the driver's run —, 10 elsewhere)
@pytest.mark.gpu
@pytest.mark.parametrize("case", range(int(os.environ.get("DVP_RANDOM_CASES", "50" if os.path.exists("/dev/kfd") else "10"))))
def test_random_configs_gpu(case):
    rng = np.random.default_rng(5000 + case)
    capi = pkg("capi")
    run_pair(lambda sc, p, seed, smp, dep: capi.from_scene(sc, p, seed=seed, sampler=smp, depths=dep), rng)
