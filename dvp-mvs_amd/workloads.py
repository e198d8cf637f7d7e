"""The BASELINE.json configurations as synthetic workloads: parameter sets as the
reference's driver leaves them (main.cpp:450-508) and the hand-over between two passes
(main.cpp:298-376 -> APD.cpp:1169-1195, 1428-1456).  Shared by bench.py, tests/ and tools/."""
import numpy as np

from . import synth

# name -> (W, H, S, iterations).  cfg1/cfg2: FIRST_INIT passes; cfg3/cfg5: a REFINE_ITER pass with
# geometric consistency, priors and WEAK pixels on top of an (untimed) FIRST_INIT pass.
CONFIGS = {
    "cfg1": dict(W=1552, H=1032, S=3, iters=2, refine=False, desc="ETH3D 'office' scale 4, 3 source views, 2 iterations, FIRST_INIT, geom off"),
    "cfg2": dict(W=3104, H=2064, S=5, iters=6, refine=False, desc="ETH3D 'office' half-res, 5 source views, 6 iterations, FIRST_INIT, geom off"),
    "cfg3": dict(W=6208, H=4128, S=9, iters=3, refine=True, desc="ETH3D 'delivery_area' full-res, 9 source views, REFINE_ITER pass, geometric consistency on, use_APD (WEAK pixels, edge/label/radius priors)"),
    "cfg5": dict(W=1920, H=1080, S=9, iters=3, refine=True, desc="T&T 'Family' full-res, 9 source views, REFINE_ITER pass, geom on, edge/visibility priors on"),
}


def depth_range(p):
    p["depth_min"] = np.float32(2.5) * np.float32(0.6)   # APD.cpp:1109-1110 on the scene's [2.5, 6.5]
    p["depth_max"] = np.float32(6.5) * np.float32(1.2)
    return p


def first_init_params(S, iters):
    """round 0, pass A (main.cpp:457-483)."""
    return depth_range(synth.default_params(S + 1, max_iterations=iters, state=synth.FIRST_INIT, use_APD=0,
                                            geom_consistency=0, weak_peak_radius=6))


def refine_iter_params(S, iters, round_index=3, j=0):
    """REFINE_ITER pass j of round `round_index` >= 1 (main.cpp:486-508); a 6208-wide image has
    round_num = 4, i.e. full resolution is round 3."""
    return depth_range(synth.default_params(
        S + 1, max_iterations=iters, state=synth.REFINE_ITER, use_APD=1, geom_consistency=1, use_detail=1,
        ransac_threshold=np.float32(0.01 - round_index * 0.00125), rotate_time=min(2 ** round_index, 4),
        weak_peak_radius=max(4 - 2 * j, 2)))


def hand_over(planes, views, weak, radius, params, W, H, extra_weak=None):
    """What ProcessProblem writes after a pass and the next pass reloads: depths out of range are
    zeroed and marked UNKNOWN (main.cpp:303-307), UNKNOWN pixels restart at the default radius
    (APD.cpp:1663-1666).  `extra_weak` (bool [H,W]): STRONG pixels to hand over as WEAK — the
    synthetic scenes are textured almost everywhere, so DepthToWeak alone leaves only a few percent
    WEAK; real ETH3D frames (white walls, floors) have 5-30 %."""
    planes = planes.copy()
    weak = weak.copy()
    radius = radius.copy()
    bad = (planes[:, 3] < params["depth_min"]) | (planes[:, 3] > params["depth_max"])
    planes[bad, 3] = 0
    weak[bad] = synth.UNKNOWN
    if extra_weak is not None:
        m = extra_weak.reshape(-1) & (weak == synth.STRONG)
        weak[m] = synth.WEAK
    radius[weak == synth.UNKNOWN] = 5
    return planes, views, weak, radius


def weak_regions(W, H, frac, flat=None, seed=0, border=8, cells=12):
    """bool [H,W]: a few LARGE connected regions covering ~frac of the image (+ the scene's low-albedo window) — the shape
    textureless walls and skies have in real scenes, where FindNearestStrongPoint walks many rings and GenNeighbours'
    directions run through tens of tries before they reach a STRONG pixel (32 x 32 tiles never ask for that).  A smooth
    random field (`cells` control points across the width, bilinear) thresholded at its (1 - frac) quantile."""
    rng = np.random.default_rng(seed)
    gy = max(2, int(round(cells * H / float(W))) + 1)
    g = rng.random((gy, cells + 1))
    xs, ys = np.linspace(0, cells, W), np.linspace(0, gy - 1, H)
    rows = np.stack([np.interp(xs, np.arange(cells + 1), g[j]) for j in range(gy)])            # [gy, W]
    y0 = np.minimum(ys.astype(np.int64), gy - 2)
    t = (ys - y0)[:, None]
    field = rows[y0] * (1.0 - t) + rows[y0 + 1] * t                                              # [H, W]
    mask = field > np.quantile(field[::8, ::8], 1.0 - frac)
    if flat is not None:
        mask = mask | flat
    mask[:border] = False
    mask[-border:] = False
    mask[:, :border] = False
    mask[:, -border:] = False
    return mask


def weak_mask(layout, W, H, frac, flat=None, seed=0):
    return weak_regions(W, H, frac, flat, seed) if layout == "regions" else weak_tiles(W, H, frac, flat, seed)


def weak_tiles(W, H, frac, flat=None, seed=0, tile=32, border=8):
    """bool [H,W]: `tile` x `tile` blocks drawn at random until ~frac of the image (+ the scene's
    low-albedo window), away from the border."""
    rng = np.random.default_rng(seed)
    tiles = rng.random((H // tile + 1, W // tile + 1)) < frac
    mask = np.kron(tiles, np.ones((tile, tile), bool))[:H, :W]
    if flat is not None:
        mask = mask | flat
    mask[:border] = False
    mask[-border:] = False
    mask[:, :border] = False
    mask[:, -border:] = False
    return mask
