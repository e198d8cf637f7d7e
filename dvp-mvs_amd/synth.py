"""Seeded synthetic MVS scenes: no dataset is available offline, so every test and
bench input comes from here.

World: a slanted plane Z = 4 + 0.25 X + 0.1 Y for X < x_step, a second plane pushed back by
`step` for X >= x_step (depth discontinuity -> edge pixels), the vertical wall between them, and a
low-albedo window on the first plane (forces WEAK pixels).  Cameras (`rig="rotated"`, the default):
every view — the reference view included — has its own rotation (look-at the scene centre, tilted a
few degrees about x and y, rolled up to +-20 degrees about z), its own K (focal +-8 %, fy != fx,
principal point off-centre, optional skew K[1]) and a centre on a +-0.4 ring (the reference view is
off the origin too), depth range [2.5, 6.5] — the MVSNet-style `Camera` the reference reads with
ReadCamera (/root/reference/APD.cpp:651-692, main.h:58-67).  `rig="axis"` is the round-1/2 rig
(R = I, one shared K, reference view at the origin) kept for the analytic checks that need it.  Images are float32 with
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


def make_camera(W, H, centre, depth_min=2.5, depth_max=6.5, R=None, K=None):
    cam = np.zeros((), dtype=CAMERA_DTYPE)
    f = 0.9 * W
    cam["K"] = np.array([f, 0, W / 2.0, 0, f, H / 2.0, 0, 0, 1], np.float32) if K is None else np.asarray(K, np.float32).ravel()
    cam["R"] = (np.eye(3) if R is None else np.asarray(R, np.float64)).astype(np.float32).ravel()
    c = np.asarray(centre, np.float64)
    R32 = cam["R"].astype(np.float64).reshape(3, 3)
    cam["t"] = (-(R32 @ c)).astype(np.float32)          # t = -R c (what a cam.txt extrinsic holds)
    t = cam["t"].astype(np.float64)
    cam["c"] = (-(R32.T @ t)).astype(np.float32)  # APD.cpp:673-677 (double, then float)
    cam["height"], cam["width"] = H, W
    cam["depth_min"], cam["depth_max"] = depth_min, depth_max
    return cam


def _rot(ax, ay, az):
    """R = Rz(az) Rx(ax) Ry(ay), angles in radians (world -> camera)."""
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Rx @ Ry


def make_rig(W, H, num_src, seed=1234, baseline=0.4, rig="rotated", skew=0.0):
    """Cameras of a scene: [num_src + 1] CAMERA_DTYPE, index 0 = reference view."""
    assert 1 <= num_src <= len(_RING)
    rng = np.random.default_rng(seed)
    dz = [0.02 * float(rng.standard_normal()) for _ in range(num_src)]   # (same draws as rounds 1-2)
    cams = np.zeros(num_src + 1, dtype=CAMERA_DTYPE)
    if rig == "axis":
        centres = [(0.0, 0.0, 0.0)] + [(baseline * a, baseline * b, z) for (a, b), z in zip(_RING[:num_src], dz)]
        for i, c in enumerate(centres):
            cams[i] = make_camera(W, H, c)
        return cams
    assert rig == "rotated"
    centres = [(0.12 * baseline, -0.09 * baseline, 0.015)] + [(baseline * a, baseline * b, z) for (a, b), z in zip(_RING[:num_src], dz)]
    target = np.array([0.0, 0.0, 4.2])      # a point on the slanted plane, roughly mid-image
    deg = np.pi / 180.0
    for i, c in enumerate(centres):
        c = np.asarray(c, np.float64)
        v = target - c
        # look-at as two tilts (Rx(ax) Ry(ay) maps v onto +z), then perturbed and rolled
        ay = -np.arctan2(v[0], v[2])
        ax = np.arctan2(v[1], np.hypot(v[0], v[2]))
        ax += deg * float(rng.uniform(-4.0, 4.0))
        ay += deg * float(rng.uniform(-4.0, 4.0))
        az = deg * float(rng.uniform(5.0, 20.0)) * (1.0 if rng.random() < 0.5 else -1.0)
        f = 0.9 * W * (1.0 + 0.08 * float(rng.uniform(-1, 1)))
        fy = f * (1.0 + 0.02 * float(rng.uniform(-1, 1)))
        cx = W / 2.0 + 0.03 * W * float(rng.uniform(-1, 1))
        cy = H / 2.0 + 0.03 * H * float(rng.uniform(-1, 1))
        sk = skew * float(rng.uniform(0.5, 1.0)) * (1.0 if i % 2 == 0 else -1.0)
        cams[i] = make_camera(W, H, c, R=_rot(ax, ay, az), K=[f, sk, cx, 0, fy, cy, 0, 0, 1])
    return cams


def _rays(cam, xg, yg):
    """World-space ray of each pixel as (C, d) with X = C + s d and s = the depth in the camera
    (d = R^T K^-1 (x, y, 1), float64 from the float32 camera the engine sees)."""
    K = cam["K"].astype(np.float64)
    R = cam["R"].astype(np.float64).reshape(3, 3)
    yn = (yg - K[5]) / K[4]
    xn = (xg - K[2] - K[1] * yn) / K[0]
    d = [R[0, k] * xn + R[1, k] * yn + R[2, k] for k in range(3)]      # R^T (xn, yn, 1)
    return cam["c"].astype(np.float64), d


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
    """Ray-cast one view.  Returns (image float32 HxW, depth float32 HxW, surface id uint8 HxW,
    flat-window mask bool HxW)."""
    ys = np.arange(H) if rows is None else np.asarray(rows)
    xg, yg = np.meshgrid(np.arange(W, dtype=np.float64), ys.astype(np.float64))
    C, (dx, dy, dz) = _rays(cam, xg, yg)
    # plane A: Z = 4 + 0.25 X + 0.1 Y ; plane B: Z = 4 + step + 0.25 X + 0.1 Y
    den = dz - 0.25 * dx - 0.1 * dy
    def hit(z0):
        # C + s d on the plane: Cz + s dz = z0 + 0.25 (Cx + s dx) + 0.1 (Cy + s dy)
        return (z0 + 0.25 * C[0] + 0.1 * C[1] - C[2]) / den
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
            YW, ZW = C[1] + sW * dy, C[2] + sW * dz
        zA_w = 4.0 + 0.25 * x_step + 0.1 * YW
        validW = np.isfinite(sW) & (sW > 0) & (ZW >= zA_w) & (ZW <= zA_w + step)
        big = 1e30
        s = np.minimum(np.where(validA, sA, big), np.minimum(np.where(validB, sB, big), np.where(validW, sW, big)))
        sid = np.where(s == np.where(validA, sA, big), 0, np.where(s == np.where(validB, sB, big), 1, 2)).astype(np.uint8)
        s = np.where(s >= big, sA, s)
    else:
        s, sid = sA, np.zeros(sA.shape, np.uint8)
    X, Y, Z = C[0] + s * dx, C[1] + s * dy, C[2] + s * dz
    # texture coordinates: planes use (X, Y); the wall uses (Z, Y)
    U = np.where(sid == 2, Z * 1.7, X)
    flat_mask = np.zeros(X.shape, bool)
    if with_flat:
        flat_mask = (sid == 0) & (X > flat[0]) & (X < flat[1]) & (Y > flat[2]) & (Y < flat[3])
    img = _texture(U, Y, px_world, flat_mask)
    img = np.clip(np.rint(img), 0, 255).astype(np.float32)
    return img, s.astype(np.float32), sid, flat_mask


def make_scene(W, H, num_src, seed=1234, with_step=True, with_flat=True, baseline=0.4, rig="rotated", skew=0.0):
    """Reference view + `num_src` source views (cameras: make_rig).

    Returns dict(images [NI,H,W] f32, cameras [NI] CAMERA_DTYPE, depth_gt [NI,H,W] f32 (depth in each
    view's own camera), normal_gt (3,) world normal of the planes, edge [H,W] u8, label [H,W] i32,
    flat [H,W] bool)."""
    cams = make_rig(W, H, num_src, seed, baseline, rig, skew)
    px_world = 4.0 / (0.9 * W)
    images, depths = [], []
    sid0 = flat0 = None
    for i in range(num_src + 1):
        img, dep, sid, fm = render_view(W, H, cams[i], px_world, with_step=with_step, with_flat=with_flat)
        images.append(img)
        depths.append(dep)
        if i == 0:
            sid0, flat0 = sid, fm
    # edge map of the reference view: surface-id changes (stand-in for the Canny map of
    # main.cpp:193-225) ; label map: surface id + 1 with -1 on edges (stand-in for labels_<s>.dmb)
    edge = np.zeros((H, W), np.uint8)
    edge[:, 1:] |= (sid0[:, 1:] != sid0[:, :-1]).astype(np.uint8)
    edge[1:, :] |= (sid0[1:, :] != sid0[:-1, :]).astype(np.uint8)
    label = sid0.astype(np.int32) + 1
    label[edge > 0] = -1
    n = np.array([0.25, 0.1, -1.0])
    n /= np.linalg.norm(n)
    return dict(images=np.stack(images), cameras=cams, depth_gt=np.stack(depths), normal_gt=n.astype(np.float32),
                edge=edge, label=label, flat=flat0, width=W, height=H)


# ---- the same scene rendered on the GPU (bench.py at full resolution: 10 views of 25.6 Mpx take
# minutes in numpy).  Same formulas in float64 torch ops; grey levels may differ from the numpy
# renderer by one level on a few pixels (libm vs device sin/cos), which is irrelevant for a timing
# workload — parity tests always use the numpy renderer on both sides.
def render_view_torch(W, H, cam, px_world, device, x_step=0.35, step=0.45, flat=(-0.55, -0.15, -0.35, 0.05)):
    import torch
    xs = torch.arange(W, dtype=torch.float64, device=device)
    ys = torch.arange(H, dtype=torch.float64, device=device)
    yg, xg = torch.meshgrid(ys, xs, indexing="ij")
    C, (dx, dy, dz) = _rays(cam, xg, yg)
    den = dz - 0.25 * dx - 0.1 * dy

    def hit(z0):
        return (z0 + 0.25 * C[0] + 0.1 * C[1] - C[2]) / den
    sA, sB = hit(4.0), hit(4.0 + step)
    validA = (C[0] + sA * dx) < x_step
    validB = (C[0] + sB * dx) >= x_step
    sW = (x_step - C[0]) / dx
    YW, ZW = C[1] + sW * dy, C[2] + sW * dz
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
    X, Y, Z = C[0] + s * dx, C[1] + s * dy, C[2] + s * dz
    del dx, dy, dz, den
    U = torch.where(sid == 2, Z * 1.7, X)
    flat_mask = (sid == 0) & (X > flat[0]) & (X < flat[1]) & (Y > flat[2]) & (Y < flat[3])
    base = 55.0 * torch.sin(9.0 * U) * torch.cos(7.0 * Y) + 35.0 * torch.sin(23.0 * U + 17.0 * Y) + 20.0 * torch.sin(41.0 * Y - 13.0 * U)
    k1, k2, k3 = (2.0 * np.pi / (q * px_world) for q in (6.0, 11.0, 23.0))
    fine = 22.0 * torch.sin(k1 * (0.8 * U + 0.6 * Y)) + 18.0 * torch.sin(k2 * (0.6 * U - 0.8 * Y) + 1.3) \
        + 14.0 * torch.sin(k3 * (U + 0.3 * Y) + 0.4) * torch.cos(k3 * (0.2 * U - Y))
    tex = base * 0.6 + fine
    tex = torch.where(flat_mask, 0.02 * tex, tex)
    img = torch.clamp(torch.round(127.5 + tex), 0, 255).to(torch.float32)
    return img, s.to(torch.float32), sid, flat_mask


def make_scene_torch(W, H, num_src, device, seed=1234, baseline=0.4, rig="rotated", skew=0.0):
    """make_scene on `device`: dict(images [NI,H,W] f32, depth_gt [NI,H,W] f32, edge [H,W] u8, label
    [H,W] i32, flat [H,W] bool — torch tensors on `device`; cameras — numpy CAMERA_DTYPE)."""
    import torch
    cams = make_rig(W, H, num_src, seed, baseline, rig, skew)
    px_world = 4.0 / (0.9 * W)
    images = torch.empty((num_src + 1, H, W), dtype=torch.float32, device=device)
    depths = torch.empty((num_src + 1, H, W), dtype=torch.float32, device=device)
    sids, flats = [], []
    for i in range(num_src + 1):
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


def planes_in_ref_cam(cam, px, depth, n_world):
    """Plane hypotheses as the kernels hold them (normal in the reference CAMERA frame, offset d with
    n.X + d = 0; APD.cu:400-405) for pixels `px` [n,2] at camera depths `depth` [n] with WORLD normals
    `n_world` [n,3] or [3].  float64 in, float32 [n,4] out."""
    K = cam["K"].astype(np.float64)
    R = cam["R"].astype(np.float64).reshape(3, 3)
    px = np.asarray(px, np.float64).reshape(-1, 2)
    depth = np.asarray(depth, np.float64).reshape(-1)
    nw = np.broadcast_to(np.asarray(n_world, np.float64), (len(px), 3))
    nc = nw @ R.T
    # Get3DPoint (APD.cu:372-377) ignores the skew K[1], and so does this
    X = np.stack([depth * (px[:, 0] - K[2]) / K[0], depth * (px[:, 1] - K[5]) / K[4], depth], 1)
    d = -(nc * X).sum(1)
    return np.concatenate([nc, d[:, None]], 1).astype(np.float32)
