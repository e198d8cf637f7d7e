"""End-to-end distance between the numerics CONTRACT (what the oracle and the HIP engine implement bit
for bit) and a LITERAL reading of the reference's source (Oracle.set_numerics(1): true divisions in
ComputeHomography / ComputeCorrespondingPoint, the source's tap order with its per-x-offset partial sums, one division
per tap, tex2D(x + 0.5f) with the add/subtract pair rounded, libm expf, the `complex` sigmoid in
double — APD.cu:679-748, 1059-1089, 905-1000, 3844).

The reference cannot be built here, so this is the available evidence for the
north_star's "within 1e-3 relative" clause: the same seeded inputs and the same counter-based RNG
stream through both evaluations of a whole FIRST_INIT pass (3 iterations) followed by one REFINE_ITER
pass with geometric consistency, WEAK pixels and priors.  PatchMatch amplifies ulp-level cost
differences wherever two hypotheses tie, so agreement is a statement about the distribution:
SURVEY.md §8c measured, for the reference's own code: 0.4 % of pixels beyond 1e-3 run twice with the same
seed (its data races), 48 % with another seed, 21 % with the other bilinear weight rounding.
Measured here on the rotated camera rig (printed with pytest -s, recorded in DESIGN.md §2): contract vs literal
0.5-2.8 % of pixels beyond 1e-3, <= 0.10 % beyond 1e-2, mean relative depth difference 0.4e-4 .. 2.2e-4, p99 ~1e-3; the
same pipeline with another seed: 14-25 % beyond 1e-3, mean 1e-2 (two orders of magnitude more).  The gates below are those measurements with
a 1.5-2x margin: the test pins the DISTANCE, it does not claim per-pixel agreement."""
import numpy as np
import pytest

from conftest import pkg, synth
from oracle import oracle as O

wl = pkg("workloads")


pytestmark = pytest.mark.hostbox   # no GPU needed; joins the `-m gpu` run on a GPU box (conftest.py)

def _pipeline(sc, S, numerics, seed):
    W, H = sc["width"], sc["height"]
    L = W * H
    p1 = wl.first_init_params(S, 3)
    o = O.from_scene(sc, p1, seed=seed)
    o.set_numerics(numerics)
    o.upload_state(planes=np.zeros((L, 4), np.float32), views=np.zeros(L, np.uint32), weak=np.full(L, synth.STRONG, np.uint8),
                   edge=sc["edge"], label=sc["label"], radius=np.full(L, 5, np.int32))
    o.run_patchmatch()
    first = o.get("planes").copy()
    st = wl.hand_over(first, o.get("selected_views"), o.get("weak_info"), o.get("radius"), p1, W, H,
                      extra_weak=wl.weak_tiles(W, H, 0.05, sc["flat"]))
    o2 = O.from_scene(sc, wl.refine_iter_params(S, 3), seed=seed, depths=sc["depth_gt"])
    o2.set_numerics(numerics)
    o2.upload_state(planes=st[0], views=st[1], weak=st[2], radius=st[3], edge=sc["edge"], label=sc["label"])
    o2.run_patchmatch()
    return first, o2.get("planes").copy(), o2.get("weak_info").copy()


def _stats(a, b):
    da, db = a[:, 3], b[:, 3]
    ok = np.isfinite(da) & np.isfinite(db) & (da != 0)
    rel = np.abs(da[ok] - db[ok]) / np.abs(da[ok])
    nrm = np.abs(a[ok, :3] - b[ok, :3]).max(axis=1)
    return dict(mean=float(rel.mean()), median=float(np.median(rel)), p99=float(np.quantile(rel, 0.99)),
                over_1e3=float((rel > 1e-3).mean()), over_1e2=float((rel > 1e-2).mean()),
                normals_over_1e3=float((nrm > 1e-3).mean()), identical=float((rel == 0).mean()))


@pytest.mark.parametrize("W,H,S", [(192, 128, 3), (256, 192, 5)])
def test_contract_vs_literal_end_to_end(W, H, S):
    sc = synth.make_scene(W, H, S)
    try:
        c1, c2, cw = _pipeline(sc, S, 0, 1234)
        l1, l2, lw = _pipeline(sc, S, 1, 1234)
    finally:
        O.lib().ora_set_numerics.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int]
        tmp = O.Oracle(8, 8, 2)
        tmp.set_numerics(0)      # the literal switch is process-wide: back to the contract
    s1, s2 = _stats(c1, l1), _stats(c2, l2)
    states = float((cw != lw).mean())
    o1, o2, ow = _pipeline(sc, S, 0, 999)    # scale: the contract against itself with another seed
    print("\ncontract, seed 1234 vs seed 999 %dx%d S=%d: %s" % (W, H, S, _stats(c2, o2)))
    assert _stats(c2, o2)["over_1e3"] > 5 * s2["over_1e3"] and _stats(c2, o2)["mean"] > 8 * s2["mean"]
    print("\ncontract vs literal %dx%d S=%d  FIRST_INIT(3 it.): %s" % (W, H, S, s1))
    print("contract vs literal %dx%d S=%d  +REFINE_ITER(geom, WEAK): %s  pixel states differing: %.4f" % (W, H, S, s2, states))
    assert s1["median"] == 0.0                     # most pixels of the first pass agree to the last bit
    for s in (s1, s2):
        assert s["median"] <= 3e-4, s
        assert s["mean"] <= 5e-4, s
        assert s["p99"] <= 3e-3, s
        assert s["over_1e3"] <= 0.04, s
        assert s["over_1e2"] <= 0.002, s
    assert states <= 0.003
