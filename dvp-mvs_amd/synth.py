"""Seeded synthetic MVS scenes: no dataset is available offline, so every test and
bench input comes from here.

World: a slanted plane Z = 4 + 0.25 X + 0.1 Y for X < x_step, a second plane pushed back by
`step` for X >= x_step (depth discontinuity -> edge pixels), the vertical wall between them, and a
low-albedo window on the first plane (forces WEAK pixels).  Cameras: K = [0.9W 0 W/2; 0 0.9W H/2],
R = I, centres on a +-0.4 ring, depth range [2.5, 6.5] — the MVSNet-style `Camera` the reference
reads with ReadCamera (/root/reference/APD.cpp:651-692, main.h:58-67).  Images are float32 with
integer grey levels in [0, 255], like cv::imread(GRAYSCALE) -> CV_32F (APD.cpp:1057-1059).
"""
import numpy as np

CAMERA_DTYPE = np.dtype([
    ("K", np.float32, 9), ("R", np.float32, 9), ("t", np.float32, 3), ("c", np.float32, 3),
    ("height", np.int32), ("width", np.int32), ("depth_min", np.float32), ("depth_max", np.float32),
])
assert CAMERA_DTYPE.itemsize == 112  # main.h:58-67

# main.h:86-112: 76 bytes, bool = 1 byte, natural alignment
PARAMS_DTYPE = np.dtype({
    "names": ["max_iterations", "num_images", "sigma_spatial", "sigma_color", "top_k", "depth_min",
              "depth_max", "geom_consistency", "strong_radius", "strong_increment", "weak_radius",
              "weak_increment", "use_APD", "use_edge", "use_limit", "use_label", "use_detail",
              "use_radius", "weak_peak_radius", "rotate_time", "ransac_threshold", "geom_factor",
              "state"],
    "formats": [np.int32, np.int32, np.float32, np.float32, np.int32, np.float32, np.float32,
                np.uint8, np.int32, np.int32, np.int32, np.int32, np.uint8, np.uint8, np.uint8,
                np.uint8, np.uint8, np.uint8, np.int32, np.int32, np.float32, np.float32, np.int32],
    "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 49, 50, 51, 52, 53, 56, 60, 64,
                68, 72],
    "itemsize": 76,
})

FIRST_INIT, REFINE_INIT, REFINE_ITER = 0, 1, 2   # main.h:74-78
WEAK, STRONG, UNKNOWN = 0, 1, 2                  # main.h:80-84

# source-view offsets on the baseline ring (unit: 0.4 world units): the 9 of the BASELINE configurations, then more
# (tests of the > 9 and > 16 view code paths); a scene with S views uses the first S
_RING = [(1.0, 0.0), (-1.0, 0.0), (0.0, 1.0), (0.0, -1.0), (0.7, 0.7), (-0.7, 0.7), (0.7, -0.7),
         (-0.7, -0.7), (0.5, -0.2),
         (0.3, 0.9), (-0.3, 0.9), (0.9, 0.3), (-0.9, -0.3), (0.35, 0.35), (-0.35, 0.35), (0.35, -0.35),
         (-0.35, -0.35), (0.85, -0.5), (-0.85, 0.5), (0.15, 0.6)]


def default_params(num_images, **kw):
    """PatchMatchParams defaults of main.h:86-112 (+ the depth range APD.cpp:1109-1110 sets)."""
    p = np.zeros((), dtype=PARAMS_DTYPE)
    p["max_iterations"] = 3
    p["num_images"] = num_images
    p["sigma_spatial"] = 5.0
    p["sigma_color"] = 3.0
    p["top_k"] = 4
    p["depth_min"] = 0.0
    p["depth_max"] = 1.0
    p["geom_consistency"] = 0
    p["strong_radius"] = 5
    p["strong_increment"] = 2
    p["weak_radius"] = 5
    p["weak_increment"] = 5
    p["use_APD"] = 1
    p["use_edge"] = 1
    p["use_limit"] = 1
    p["use_label"] = 1
    p["use_detail"] = 0
    p["use_radius"] = 1
    p["weak_peak_radius"] = 2
    p["rotate_time"] = 4
    p["ransac_threshold"] = 0.005
    p["geom_factor"] = 0.2
    p["state"] = FIRST_INIT
    for k, v in kw.items():
        p[k] = v
    return p


def make_camera(W, H, centre, depth_min=2.5, depth_max=6.5):
    cam = np.zeros((), dtype=CAMERA_DTYPE)
    f = 0.9 * W
    cam["K"] = np.array([f, 0, W / 2.0, 0, f, H / 2.0, 0, 0, 1], np.float32)
    cam["R"] = np.eye(3, dtype=np.float32).ravel()
    c = np.asarray(centre, np.float64)
    cam["t"] = (-c).astype(np.float32)          # t = -R c
    R = cam["R"].astype(np.float64).reshape(3, 3)
    t = cam["t"].astype(np.float64)
    cam["c"] = (-(R.T @ t)).astype(np.float32)  # APD.cpp:673-677 (double, then float)
    cam["height"], cam["width"] = H, W
    cam["depth_min"], cam["depth_max"] = depth_min, depth_max
    return cam


def _texture(X, Y, px_world, flat_mask):
    """Procedural albedo: the survey's three base terms + three octaves whose wavelength is tied
    to the pixel footprint so that every resolution sees texture inside an 11x11 window."""
    base = 55.0 * np.sin(9.0 * X) * np.cos(7.0 * Y) + 35.0 * np.sin(23.0 * X + 17.0 * Y) \
        + 20.0 * np.sin(41.0 * Y - 13.0 * X)
    k1 = 2.0 * np.pi / (6.0 * px_world)
    k2 = 2.0 * np.pi / (11.0 * px_world)
    k3 = 2.0 * np.pi / (23.0 * px_world)
    fine = 22.0 * np.sin(k1 * (0.8 * X + 0.6 * Y)) + 18.0 * np.sin(k2 * (0.6 * X - 0.8 * Y) + 1.3) \
        + 14.0 * np.sin(k3 * (X + 0.3 * Y) + 0.4) * np.cos(k3 * (0.2 * X - Y))
    tex = base * 0.6 + fine
    tex = np.where(flat_mask, 0.02 * tex, tex)
    return 127.5 + tex


def render_view(W, H, cam, px_world, x_step=0.35, step=0.45, flat=(-0.55, -0.15, -0.35, 0.05),
                with_step=True, with_flat=True, rows=None):
    """Ray-cast one view.  Returns (image float32 HxW, depth float32 HxW, surface id uint8 HxW)."""
    K = cam["K"].astype(np.float64)
    C = cam["c"].astype(np.float64)
    fx, cx, fy, cy = K[0], K[2], K[4], K[5]
    ys = np.arange(H) if rows is None else np.asarray(rows)
    xg, yg = np.meshgrid(np.arange(W, dtype=np.float64), ys.astype(np.float64))
    dx, dy = (xg - cx) / fx, (yg - cy) / fy          # ray direction (dx, dy, 1); R = I
    # plane A: Z = 4 + 0.25 X + 0.1 Y ; plane B: Z = 4 + step + 0.25 X + 0.1 Y
    def hit(z0):
        # C + s d on the plane: Cz + s = z0 + 0.25 (Cx + s dx) + 0.1 (Cy + s dy)
        return (z0 + 0.25 * C[0] + 0.1 * C[1] - C[2]) / (1.0 - 0.25 * dx - 0.1 * dy)
    sA = hit(4.0)
    XA, YA = C[0] + sA * dx, C[1] + sA * dy
    if with_step:
        sB = hit(4.0 + step)
        XB, YB = C[0] + sB * dx, C[1] + sB * dy
        validA = XA < x_step
        validB = XB >= x_step
        # wall X = x_step between the two planes
        with np.errstate(divide="ignore", invalid="ignore"):
            sW = (x_step - C[0]) / dx
            YW, ZW = C[1] + sW * dy, C[2] + sW
        zA_w = 4.0 + 0.25 * x_step + 0.1 * YW
        validW = np.isfinite(sW) & (sW > 0) & (ZW >= zA_w) & (ZW <= zA_w + step)
        big = 1e30
        s = np.minimum(np.where(validA, sA, big), np.minimum(np.where(validB, sB, big), np.where(validW, sW, big)))
        sid = np.where(s == np.where(validA, sA, big), 0, np.where(s == np.where(validB, sB, big), 1, 2)).astype(np.uint8)
        s = np.where(s >= big, sA, s)
    else:
        s, sid = sA, np.zeros(sA.shape, np.uint8)
    X, Y, Z = C[0] + s * dx, C[1] + s * dy, C[2] + s
    # texture coordinates: planes use (X, Y); the wall uses (Z, Y)
    U = np.where(sid == 2, Z * 1.7, X)
    flat_mask = np.zeros(X.shape, bool)
    if with_flat:
        flat_mask = (sid == 0) & (X > flat[0]) & (X < flat[1]) & (Y > flat[2]) & (Y < flat[3])
    img = _texture(U, Y, px_world, flat_mask)
    img = np.clip(np.rint(img), 0, 255).astype(np.float32)
    return img, Z.astype(np.float32), sid


def make_scene(W, H, num_src, seed=1234, with_step=True, with_flat=True, baseline=0.4):
    """Reference view at the origin + `num_src` source views on the ring.

    Returns dict(images [NI,H,W] f32, cameras [NI] CAMERA_DTYPE, depth_gt [NI,H,W] f32,
    normal_gt (3,), edge [H,W] u8, label [H,W] i32, flat [H,W] bool)."""
    assert 1 <= num_src <= len(_RING)
    rng = np.random.default_rng(seed)
    centres = [(0.0, 0.0, 0.0)] + [(baseline * a, baseline * b, 0.02 * float(rng.standard_normal())) for a, b in _RING[:num_src]]
    cams = np.zeros(num_src + 1, dtype=CAMERA_DTYPE)
    px_world = 4.0 / (0.9 * W)
    images, depths = [], []
    sid0 = None
    for i, c in enumerate(centres):
        cams[i] = make_camera(W, H, c)
        img, dep, sid = render_view(W, H, cams[i], px_world, with_step=with_step, with_flat=with_flat)
        images.append(img)
        depths.append(dep)
        if i == 0:
            sid0 = sid
    # edge map of the reference view: surface-id changes (stand-in for the Canny map of
    # main.cpp:193-225) ; label map: surface id + 1 with -1 on edges (stand-in for labels_<s>.dmb)
    edge = np.zeros((H, W), np.uint8)
    edge[:, 1:] |= (sid0[:, 1:] != sid0[:, :-1]).astype(np.uint8)
    edge[1:, :] |= (sid0[1:, :] != sid0[:-1, :]).astype(np.uint8)
    label = sid0.astype(np.int32) + 1
    label[edge > 0] = -1
    n = np.array([0.25, 0.1, -1.0])
    n /= np.linalg.norm(n)
    # flat window mask in the reference view (for tests that want WEAK pixels)
    K = cams[0]["K"].astype(np.float64)
    xg, yg = np.meshgrid(np.arange(W), np.arange(H))
    Xr = depths[0] * (xg - K[2]) / K[0]
    Yr = depths[0] * (yg - K[5]) / K[4]
    flat = (sid0 == 0) & (Xr > -0.55) & (Xr < -0.15) & (Yr > -0.35) & (Yr < 0.05) if with_flat else np.zeros((H, W), bool)
    return dict(images=np.stack(images), cameras=cams, depth_gt=np.stack(depths), normal_gt=n.astype(np.float32),
                edge=edge, label=label, flat=flat, width=W, height=H)


# ---- the same scene rendered on the GPU (bench.py at full resolution: 10 views of 25.6 Mpx take
# minutes in numpy).  Same formulas in float64 torch ops; grey levels may differ from the numpy
# renderer by one level on a few pixels (libm vs device sin/cos), which is irrelevant for a timing
# workload — parity tests always use the numpy renderer on both sides.
def render_view_torch(W, H, cam, px_world, device, x_step=0.35, step=0.45, flat=(-0.55, -0.15, -0.35, 0.05)):
    import torch
    K = cam["K"].astype(np.float64)
    C = cam["c"].astype(np.float64)
    fx, cx, fy, cy = K[0], K[2], K[4], K[5]
    xs = torch.arange(W, dtype=torch.float64, device=device)
    ys = torch.arange(H, dtype=torch.float64, device=device)
    yg, xg = torch.meshgrid(ys, xs, indexing="ij")
    dx, dy = (xg - cx) / fx, (yg - cy) / fy
    den = 1.0 - 0.25 * dx - 0.1 * dy

    def hit(z0):
        return (z0 + 0.25 * C[0] + 0.1 * C[1] - C[2]) / den
    sA, sB = hit(4.0), hit(4.0 + step)
    validA = (C[0] + sA * dx) < x_step
    validB = (C[0] + sB * dx) >= x_step
    sW = (x_step - C[0]) / dx
    YW, ZW = C[1] + sW * dy, C[2] + sW
    zA_w = 4.0 + 0.25 * x_step + 0.1 * YW
    validW = torch.isfinite(sW) & (sW > 0) & (ZW >= zA_w) & (ZW <= zA_w + step)
    big = 1e30
    cA = torch.where(validA, sA, torch.full_like(sA, big))
    cB = torch.where(validB, sB, torch.full_like(sA, big))
    cW = torch.where(validW, sW, torch.full_like(sA, big))
    s = torch.minimum(cA, torch.minimum(cB, cW))
    sid = torch.where(s == cA, 0, torch.where(s == cB, 1, 2)).to(torch.uint8)
    s = torch.where(s >= big, sA, s)
    del cA, cB, cW, sB, sW, YW, ZW, zA_w, validA, validB, validW
    X, Y, Z = C[0] + s * dx, C[1] + s * dy, C[2] + s
    U = torch.where(sid == 2, Z * 1.7, X)
    flat_mask = (sid == 0) & (X > flat[0]) & (X < flat[1]) & (Y > flat[2]) & (Y < flat[3])
    base = 55.0 * torch.sin(9.0 * U) * torch.cos(7.0 * Y) + 35.0 * torch.sin(23.0 * U + 17.0 * Y) + 20.0 * torch.sin(41.0 * Y - 13.0 * U)
    k1, k2, k3 = (2.0 * np.pi / (q * px_world) for q in (6.0, 11.0, 23.0))
    fine = 22.0 * torch.sin(k1 * (0.8 * U + 0.6 * Y)) + 18.0 * torch.sin(k2 * (0.6 * U - 0.8 * Y) + 1.3) \
        + 14.0 * torch.sin(k3 * (U + 0.3 * Y) + 0.4) * torch.cos(k3 * (0.2 * U - Y))
    tex = base * 0.6 + fine
    tex = torch.where(flat_mask, 0.02 * tex, tex)
    img = torch.clamp(torch.round(127.5 + tex), 0, 255).to(torch.float32)
    return img, Z.to(torch.float32), sid, flat_mask


def make_scene_torch(W, H, num_src, device, seed=1234, baseline=0.4):
    """make_scene on `device`: dict(images [NI,H,W] f32, depth_gt [NI,H,W] f32, edge [H,W] u8, label
    [H,W] i32, flat [H,W] bool — torch tensors on `device`; cameras — numpy CAMERA_DTYPE)."""
    import torch
    assert 1 <= num_src <= len(_RING)
    rng = np.random.default_rng(seed)
    centres = [(0.0, 0.0, 0.0)] + [(baseline * a, baseline * b, 0.02 * float(rng.standard_normal())) for a, b in _RING[:num_src]]
    cams = np.zeros(num_src + 1, dtype=CAMERA_DTYPE)
    px_world = 4.0 / (0.9 * W)
    images = torch.empty((num_src + 1, H, W), dtype=torch.float32, device=device)
    depths = torch.empty((num_src + 1, H, W), dtype=torch.float32, device=device)
    sids, flats = [], []
    for i, c in enumerate(centres):
        cams[i] = make_camera(W, H, c)
        images[i], depths[i], sid, fm = render_view_torch(W, H, cams[i], px_world, device)
        sids.append(sid)
        flats.append(fm)
    return dict(images=images, cameras=cams, depth_gt=depths, sids=sids, flats=flats, width=W, height=H)


def view_priors_torch(sid, flat):
    """edge / label maps of one view from its surface-id map (same rule as make_scene)."""
    import torch
    edge = torch.zeros_like(sid)
    edge[:, 1:] |= (sid[:, 1:] != sid[:, :-1]).to(torch.uint8)
    edge[1:, :] |= (sid[1:, :] != sid[:-1, :]).to(torch.uint8)
    label = sid.to(torch.int32) + 1
    label[edge > 0] = -1
    return edge, label
