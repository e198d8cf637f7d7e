#!/usr/bin/env python3
"""Generate tests/golden/golden_small.npz from the CPU oracle (the reference itself cannot be built
or run in this image: it needs CUDA, OpenCV and Boost).  These are regression anchors for the
oracle + engine pair: inputs (seeded synthetic scene, KAT pixel/plane pairs) and expected outputs
(cost vectors, final planes / views / pixel states of a 2-iteration FIRST_INIT pass)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth, make_params, first_pass_state   # noqa: E402
from oracle import oracle as O   # noqa: E402

W, H, S, seed = 96, 64, 3, 4242
sc = synth.make_scene(W, H, S)
p = make_params(S + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
o = O.from_scene(sc, p, seed=seed)
o.upload_state(**first_pass_state(sc))
rng = np.random.default_rng(11)
n = 512
px = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
depth = rng.uniform(1.5, 7.8, n).astype(np.float32)
nrm = rng.normal(size=(n, 3)).astype(np.float32)
nrm[:, 2] = -np.abs(nrm[:, 2]) - 0.3
nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)


planes = synth.planes_in_ref_cam(sc["cameras"][0], px, depth, nrm)
costs = o.eval_cost_vectors(px, planes)
o.run_patchmatch()
np.savez_compressed(os.path.join(os.path.dirname(__file__), "golden_small.npz"), W=W, H=H, S=S, seed=seed,
                    images=sc["images"], kat_px=px, kat_planes=planes, kat_costs=costs, planes=o.get("planes"),
                    selected_views=o.get("selected_views"), weak_info=o.get("weak_info"))
print("written")


# ---- second fixture: a REFINE_ITER pass with WEAK pixels, priors and geometric consistency -----------
from conftest import second_pass_inputs   # noqa: E402

W2, H2, S2 = 96, 72, 3
sc2 = synth.make_scene(W2, H2, S2)
p1 = make_params(S2 + 1, max_iterations=2, state=synth.FIRST_INIT, use_APD=0)
o1 = O.from_scene(sc2, p1, seed=777)
o1.upload_state(**first_pass_state(sc2))
o1.run_patchmatch()
st = second_pass_inputs(o1, sc2)
weak = st["weak"].reshape(H2, W2)
weak[sc2["flat"] & (weak == synth.STRONG)] = synth.WEAK
weak[:6, :] = synth.UNKNOWN
st["weak"] = weak.reshape(-1)
p2 = make_params(S2 + 1, max_iterations=2, state=synth.REFINE_ITER, use_APD=1, geom_consistency=1,
                 weak_peak_radius=4, rotate_time=2, ransac_threshold=0.01)
o2 = O.from_scene(sc2, p2, seed=778, depths=sc2["depth_gt"])
o2.upload_state(**st)
o2.run_patchmatch()
np.savez_compressed(os.path.join(os.path.dirname(__file__), "golden_weak.npz"), W=W2, H=H2, S=S2,
                    in_planes=st["planes"], in_weak=st["weak"], in_views=st["views"], in_radius=st["radius"],
                    planes=o2.get("planes"), costs=o2.get("costs"), selected_views=o2.get("selected_views"),
                    weak_info=o2.get("weak_info"), radius=o2.get("radius"), neighbours=o2.get("neighbours"),
                    weak_count=o2.weak_count())
print("written weak fixture, weak_count", o2.weak_count())
