"""The host stages next to the hot path against something OTHER than themselves:
  * RunFusion (host/fusion.cpp) vs oracle/ora_host.cpp::ora_run_fusion, a sequential restatement of
    APD.cpp:1809-1960 — the point LIST (order, coordinates bit for bit, colours) must be identical, including the
    order-dependent claims, WEAK thresholds, zero depths, failing witnesses and blocks/ masks;
  * the Canny edge map (host/edges.cpp) vs an independent numpy/scipy Canny with the reference's thresholds;
  * the Depth-Anything prior's ratio map (host/prior.cpp) on a hand-built sparse point file vs the closed form.
CPU only (the host library needs no GPU for these)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synth
from oracle import oracle as O


pytestmark = pytest.mark.hostbox   # no GPU needed; joins the `-m gpu` run on a GPU box (conftest.py)

def _write_binmat(path, a, typ):
    with open(path, "wb") as f:
        f.write(np.array([1, a.shape[0], a.shape[1], typ], np.int32).tobytes())
        f.write(np.ascontiguousarray(a).tobytes())


def _host_tool(*args, env=None):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "dvp-mvs_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "host")])
    return subprocess.run([os.path.join(ROOT, "tests", "host", "test_host")] + [str(a) for a in args], capture_output=True, text=True, env=env)


def _read_ply(path):
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int(head.decode().split("element vertex ")[1].split("\n")[0])
    pts = np.frombuffer(body, np.dtype([("xyz", "<f4", 3), ("bgr", "u1", 3)]))
    assert len(pts) == n
    return pts


@pytest.mark.parametrize("kind,with_blocks", [("eth", False), ("eth", True), ("tat-intermediate", False), ("tat-intermediate", True), ("tat-advanced", False)])
def test_fusion_equals_sequential_restatement(tmp_path, kind, with_blocks):
    """RunFusion / RunFusion_TAT_Intermediate / RunFusion_TAT_advanced on the host's cores (host/fusion.cpp) against the
    sequential restatements of APD.cpp:1809-1960 / 1962-2130 / 2132-2279 (oracle/ora_host.cpp): the same points in the same
    order, coordinates bit for bit, colours as the PLY stores them."""
    _fusion_case(tmp_path, kind, with_blocks, "host")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,with_blocks,size", [("tat-intermediate", False, (80, 56)), ("tat-intermediate", True, (80, 56)), ("tat-advanced", False, (80, 56)),
                                                   ("tat-intermediate", False, (333, 250)), ("tat-advanced", True, (333, 250))])
def test_device_graded_fusion_equals_sequential_restatement(tmp_path, kind, with_blocks, size):
    """The Tanks & Temples variants on the GPU (dvp_fuse_view_graded: the stale-residual quirk as a maximum scan over the raster
    order) against their sequential restatements; the larger case spans several scan blocks per row of blocks."""
    _fusion_case(tmp_path, kind, with_blocks, "device", size=size)


@pytest.mark.gpu
@pytest.mark.parametrize("with_blocks,size", [(False, (80, 56)), (True, (80, 56)), (False, (333, 250))])
def test_device_fusion_equals_sequential_restatement(tmp_path, with_blocks, size):
    """RunFusion as the driver runs it — on the GPU through dvp_fuse_* (csrc/dvp_fuse.hip: candidates in parallel, the
    order-dependent claims resolved in rounds) — against the same sequential restatement: the same point list.  The larger
    case has tens of thousands of pixels that share witnesses (several resolve rounds)."""
    _fusion_case(tmp_path, "eth", with_blocks, "device", size=size)


def _fusion_case(tmp_path, kind, with_blocks, where, size=(80, 56)):
    W, H = size
    NV, NSRC = 5, (3 if kind == "eth" else 4)
    depth_noise = 0.0012 if kind == "eth" else 0.0004   # the graded variants accept k / 3500 ... k / 3000 of relative depth difference
    d = str(tmp_path / "scene")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), str(NSRC)], stdout=subprocess.DEVNULL)
    sc = synth.make_scene(W, H, NV - 1)
    rng = np.random.default_rng(21)
    n_true = sc["normal_gt"].astype(np.float64)
    depths, normals, weaks, colours, blocks = [], [], [], [], []
    for v in range(NV):
        dep = sc["depth_gt"][v].astype(np.float64)
        dep *= 1.0 + rng.normal(0, depth_noise, dep.shape)       # part of the pixels fail the depth test
        dep[rng.random(dep.shape) < 0.05] = 0.0                  # holes
        dep[rng.random(dep.shape) < 0.01] = -1.0                 # and negative depths
        nrm = np.tile(n_true, (H, W, 1)) + rng.normal(0, 0.025, (H, W, 3))   # some beyond 10 degrees
        nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
        weak = rng.integers(0, 3, (H, W)).astype(np.uint8)
        rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        r = os.path.join(d, "APD", "%08d" % v)
        os.makedirs(r, exist_ok=True)
        _write_binmat(os.path.join(r, "depths.dmb"), dep.astype(np.float32), 5)
        _write_binmat(os.path.join(r, "APD_normals.dmb"), nrm.astype(np.float32), 21)
        _write_binmat(os.path.join(r, "weak.bin"), weak, 0)
        os.remove(os.path.join(d, "images", "%08d.pgm" % v))
        with open(os.path.join(d, "images", "%08d.ppm" % v), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (W, H))
            f.write(rgb.tobytes())
        depths.append(np.ascontiguousarray(dep.astype(np.float32)))
        normals.append(np.ascontiguousarray(nrm.astype(np.float32)))
        weaks.append(weak)
        colours.append(np.ascontiguousarray(rgb[:, :, ::-1]))   # BGR, as cv::imread returns it
        if with_blocks:
            from PIL import Image
            os.makedirs(os.path.join(d, "blocks"), exist_ok=True)
            m = np.full((H, W), 255, np.uint8)
            m[:, : W // 3 + 4 * v] = 0
            fn = os.path.join(d, "blocks", "mask_%d.jpg" % v)
            Image.fromarray(m, "L").save(fn, quality=95)
            blocks.append(np.ascontiguousarray(np.array(Image.open(fn).convert("L"))))
    out = _host_tool("--fuse", d, env=dict(os.environ, DVP_FUSION_KIND=kind, DVP_FUSION_ON=where))
    if where == "device":
        assert "resolve rounds" in out.stdout, out.stdout[-400:]
        print(out.stdout.strip().split("\n")[-1])
    assert out.returncode == 0, out.stdout[-600:] + out.stderr[-600:]
    got = _read_ply(os.path.join(d, "APD", "APD.ply"))

    # the same inputs through the restatement
    lines = open(os.path.join(d, "pair.txt")).read().split("\n")
    src = np.full((NV, NSRC + 1), -1, np.int32)
    for v in range(NV):
        toks = lines[2 + 2 * v].split()
        ids = [int(toks[1 + 2 * i]) for i in range(int(toks[0])) if float(toks[2 + 2 * i]) > 0]
        src[v, :len(ids)] = ids
    cams = sc["cameras"].copy()
    # the driver parses cams/*.txt: R, t, K from "%.9g" text (exact for float32), c recomputed — same values
    L = O.lib()
    arr = lambda lst: (ctypes.c_void_p * NV)(*[a.ctypes.data for a in lst])
    cap = NV * W * H
    xyz = np.zeros((cap, 3), np.float32)
    bgr = np.zeros((cap, 3), np.float32)
    if kind == "eth":
        L.ora_run_fusion.restype = ctypes.c_int
        L.ora_run_fusion.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        n = L.ora_run_fusion(NV, H, W, cams.ctypes.data, arr(depths), arr(normals), arr(weaks), arr(colours),
                             arr(blocks) if with_blocks else None, src.ctypes.data, NSRC + 1, xyz.ctypes.data, bgr.ctypes.data, cap)
    else:
        L.ora_run_fusion_tat.restype = ctypes.c_int
        L.ora_run_fusion_tat.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        n = L.ora_run_fusion_tat(int(kind == "tat-advanced"), NV, H, W, cams.ctypes.data, arr(depths), arr(normals), arr(colours),
                                 arr(blocks) if with_blocks else None, src.ctypes.data, NSRC + 1, xyz.ctypes.data, bgr.ctypes.data, cap)
    print("fusion: %d points (oracle), %d (host)" % (n, len(got)))
    assert 0.2 * W * H < n < cap, n
    assert n == len(got), (n, len(got))
    assert np.array_equal(xyz[:n].view(np.uint32), got["xyz"].view(np.uint32))        # same points, same order, same bits
    assert np.array_equal(bgr[:n].astype(np.uint8), got["bgr"])                        # static_cast<uchar> of the mean colour
    # the data really exercise the rules: WEAK reference pixels, rejected witnesses, claimed pixels, masked columns
    assert n < 0.9 * sum((dp > 0).sum() for dp in depths)
    if with_blocks and kind == "eth":
        assert n < 0.75 * W * H * 1.6


def test_depth_image_point_cloud(tmp_path):
    """ExportDepthImagePointCloud (APD.cpp:2281-2314): the depth map of ONE view as a cloud — depths inside the range, no NaN,
    columns outer / rows inner, Get3DPointonWorld in binary32, colours of the image — against numpy."""
    W, H = 60, 44
    d = str(tmp_path / "scene")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), "2", "1"], stdout=subprocess.DEVNULL)
    sc = synth.make_scene(W, H, 1)
    rng = np.random.default_rng(3)
    dep = sc["depth_gt"][0].astype(np.float32).copy()
    dep[rng.random(dep.shape) < 0.1] = 0.0
    dep[rng.random(dep.shape) < 0.05] = np.nan
    dep[rng.random(dep.shape) < 0.05] = 100.0
    rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    _write_binmat(os.path.join(d, "dep.dmb"), dep, 5)
    img = os.path.join(d, "images", "colour.ppm")
    with open(img, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (W, H))
        f.write(rgb.tobytes())
    ply = os.path.join(d, "one.ply")
    out = _host_tool("--depth-cloud", os.path.join(d, "dep.dmb"), img, os.path.join(d, "cams", "00000000_cam.txt"), ply, 2.0, 8.0)
    assert out.returncode == 0, out.stderr[-400:]
    got = _read_ply(ply)
    cam = sc["cameras"][0]
    K, R, t = cam["K"].astype(np.float32), cam["R"].astype(np.float32), cam["t"].astype(np.float32)
    f32 = np.float32
    C = [-(R[0 + k] * t[0] + R[3 + k] * t[1] + R[6 + k] * t[2]) for k in range(3)]
    want_xyz, want_bgr = [], []
    for i in range(W):
        for j in range(H):
            z = dep[j, i]
            if np.isnan(z) or z < 2.0 or z > 8.0:
                continue
            px = z * (f32(i) - K[2]) / K[0]
            py = z * (f32(j) - K[5]) / K[4]
            want_xyz.append([(R[0] * px + R[3] * py + R[6] * z) + C[0], (R[1] * px + R[4] * py + R[7] * z) + C[1], (R[2] * px + R[5] * py + R[8] * z) + C[2]])
            want_bgr.append(rgb[j, i, ::-1])
    want_xyz = np.array(want_xyz, np.float32)
    assert len(got) == len(want_xyz) > 0.5 * W * H
    assert np.array_equal(got["xyz"].view(np.uint32), want_xyz.view(np.uint32))
    assert np.array_equal(got["bgr"], np.array(want_bgr, np.uint8))


def _np_canny(img, low, high):
    """Canny as OpenCV documents it for 8-bit input, aperture 3, L2gradient = true — written independently of
    host/edges.cpp with array operations: Sobel with replicated border, squared magnitude against squared thresholds,
    non-maximum suppression by gradient sector (boundaries tan 22.5 / tan 67.5 degrees; the comparison is `>` towards
    the left / upper neighbour and `>=` towards the right / lower one on the horizontal / vertical sectors, `>` both
    ways on the diagonals), hysteresis = 8-connected components of (candidate | strong) that contain a strong pixel."""
    from scipy import ndimage
    a = np.pad(img.astype(np.int64), 1, mode="edge")
    gx = (a[:-2, 2:] + 2 * a[1:-1, 2:] + a[2:, 2:]) - (a[:-2, :-2] + 2 * a[1:-1, :-2] + a[2:, :-2])
    gy = (a[2:, :-2] + 2 * a[2:, 1:-1] + a[2:, 2:]) - (a[:-2, :-2] + 2 * a[:-2, 1:-1] + a[:-2, 2:])
    mag = gx * gx + gy * gy
    m = np.pad(mag, 1)
    c = m[1:-1, 1:-1]
    ax, ay = np.abs(gx).astype(np.float64), np.abs(gy).astype(np.float64)
    t22, t67 = np.tan(np.pi / 8), np.tan(3 * np.pi / 8)
    horiz = ay < ax * t22
    vert = ay > ax * t67
    same_sign = (gx ^ gy) >= 0          # diagonal towards (+1, +1) when the signs agree
    keep_h = (c > m[1:-1, :-2]) & (c >= m[1:-1, 2:])
    keep_v = (c > m[:-2, 1:-1]) & (c >= m[2:, 1:-1])
    keep_d1 = (c > m[:-2, :-2]) & (c > m[2:, 2:])      # s = +1
    keep_d2 = (c > m[:-2, 2:]) & (c > m[2:, :-2])      # s = -1
    keep = np.where(horiz, keep_h, np.where(vert, keep_v, np.where(same_sign, keep_d1, keep_d2)))
    lo2, hi2 = int(np.floor(low * low)) if low > 0 else int(low), int(np.floor(high * high)) if high > 0 else int(high)
    cand = keep & (mag > lo2)
    strong = cand & (mag > hi2)
    lab, n = ndimage.label(cand, structure=np.ones((3, 3)))
    good = np.zeros(n + 1, bool)
    good[np.unique(lab[strong])] = True
    good[0] = False
    return good[lab]


def test_canny_edge_map_vs_independent_canny(tmp_path):
    """host/edges.cpp (EdgeSegment mode 0: median-adaptive Canny low = (1 - 0.67) median, high = median, aperture 3, L2
    gradient, APD.cpp:404-433, + the border fix-ups :452-463) on the synthetic views against _np_canny."""
    W, H = 160, 120
    sc = synth.make_scene(W, H, 2)
    rng = np.random.default_rng(3)
    for k, img in enumerate([sc["images"][0], sc["images"][1], np.clip(sc["images"][2] + rng.normal(0, 6, (H, W)), 0, 255)]):
        u8 = np.rint(img).astype(np.uint8)
        fn = str(tmp_path / ("v%d.pgm" % k))
        with open(fn, "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (W, H))
            f.write(u8.tobytes())
        out = _host_tool("--edges", fn, 0, str(tmp_path / "e.dmb"))
        assert out.returncode == 0, out.stderr
        raw = open(str(tmp_path / "e.dmb"), "rb").read()
        hdr = np.frombuffer(raw[:16], np.int32)
        assert tuple(hdr) == (1, H, W, 0)
        got = np.frombuffer(raw[16:], np.uint8).reshape(H, W) > 0
        # thresholds exactly as APD.cpp:420-431 (histogram median over grey levels 0..254)
        hist = np.bincount(u8.ravel(), minlength=256)
        cum, med = 0, -1
        for i in range(255):
            cum += hist[i]
            if cum > (H * W) // 2:
                med = i
                break
        t1, t2 = int(np.float32(1 - np.float32(0.67)) * med), med
        want = _np_canny(u8, t1, t2)
        # border fix-ups of APD.cpp:452-463 (sequential, columns first then rows — the corners see both)
        w2 = want.copy()
        for y in range(H):
            if not w2[y, 1]:
                w2[y, 0] = False
            if not w2[y, W - 2]:
                w2[y, W - 1] = False
        for x in range(W):
            if not w2[1, x]:
                w2[0, x] = False
            if not w2[H - 2, x]:
                w2[H - 1, x] = False
        assert 0.01 < w2.mean() < 0.5
        diff = int((w2 != got).sum())
        print("canny view %d: %d edge pixels, %d differ" % (k, int(w2.sum()), diff))
        assert diff == 0, (k, diff, int(w2.sum()))


def _read_binmat(fn):
    raw = open(fn, "rb").read()
    ver, rows, cols, typ = np.frombuffer(raw[:16], np.int32)
    dt = {0: np.uint8, 4: np.int32, 5: np.float32}[int(typ)]
    return np.frombuffer(raw[16:], dt).reshape(rows, cols)


def test_label_stages_vs_independent_reading(tmp_path):
    """The label half of the priors (host/labels.cpp = EdgeSegment mode 1, APD.cpp:348-401, 437-499), stage by stage against
    an array-based numpy / scipy reading written from the source text.  Deterministic integer work, compared exactly:
      * Roberts cross (APD.cpp:120-136: (uchar)sqrt(t1^2 + t2^2) on the interior, 50 / 50 on the frame) + threshold 4;
      * the frame clean-up (:452-463, sequential: columns first, then rows);
      * 4-connected regions of the zero pixels (Connect + Label_Update, :192-346): the PARTITION must be scipy.ndimage.label's,
        label 0 = textured, a region of at most weak_tex_num = rows * cols / (1024 << 2 scale) pixels becomes -1, the others
        keep distinct positive labels numbered in raster order of their first pixel's provisional label.
    Third-party arithmetic in between — cv::resize (bilinear on 8-bit), cv::HoughLinesP, cv::line — stays property-tested
    (test_boundary.py::test_label_segmentation): the Hough stage may only ADD white pixels, the resized map is fed to the
    independent reading as it comes."""
    from scipy import ndimage
    rng = np.random.default_rng(11)
    H, W = 360, 480
    img = np.full((H, W), 90, np.uint8)
    img[:, 250:] = 170
    img[:, 238:262] = rng.integers(0, 255, (H, 24))
    img[120:170, 60:120] = rng.integers(0, 255, (50, 60))
    img[200:206, 300:420] = rng.integers(0, 255, (6, 120))          # a thin textured bar inside the right wall
    img[128:140, 80:92] = 128                                         # a flat island inside the textured block: a region below weak_tex_num
    img[148:160, 100:112] = 40
    img += rng.integers(0, 2, (H, W)).astype(np.uint8)                # grain below the Roberts threshold
    img[300:340, 20:200] = (np.arange(180)[None, :] * 1.4).astype(np.uint8) + 20   # a ramp: gradient 1.4 per pixel at full size
    pgm = str(tmp_path / "i.pgm")
    with open(pgm, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (W, H))
        f.write(img.tobytes())
    for scale in (0, 1):
        pre = str(tmp_path / ("s%d" % scale))
        r = _host_tool("--label-stages", pgm, scale, pre)
        assert r.returncode == 0, r.stderr
        quarter, texture, lines, resized, cleaned, labels = [_read_binmat("%s_%s.dmb" % (pre, n)) for n in ("quarter", "texture", "lines", "resized", "cleaned", "labels")]
        weak_tex_num = int(open(pre + "_n.txt").read())
        assert weak_tex_num == int(1.0 * H * W / (1024 << scale << scale))
        assert quarter.shape == (H // 2 // 2, W // 2 // 2)
        # --- Roberts cross + threshold on the quarter-size image
        q = quarter.astype(np.int64)
        t1 = np.full(q.shape, 50, np.int64)
        t2 = np.full(q.shape, 50, np.int64)
        t1[1:-1, 1:-1] = q[1:-1, 1:-1] - q[2:, 2:]
        t2[1:-1, 1:-1] = q[2:, 1:-1] - q[1:-1, 2:]
        rob = np.floor(np.sqrt((t1 * t1 + t2 * t2).astype(np.float64))).astype(np.int64)
        rob = np.where(rob > 255, rob & 255, rob)                     # (uchar) of a value above 255 wraps (|t| <= 255: sqrt <= 360)
        want_tex = np.where(rob > 4, 255, 0).astype(np.uint8)
        assert np.array_equal(texture, want_tex), int((texture != want_tex).sum())
        assert 0.02 < (want_tex > 0).mean() < 0.6
        # --- the Hough stage only draws white lines
        assert np.all(lines[texture > 0] == 255) and set(np.unique(lines)) <= {0, 255}
        # --- frame clean-up of the resized, re-thresholded map
        rows, cols = resized.shape
        assert (rows, cols) == (int(round(H / (1 << scale))), int(round(W / (1 << scale)))) and set(np.unique(resized)) <= {0, 255}
        c = resized.copy()
        for y in range(rows):
            if c[y, 1] == 0:
                c[y, 0] = 0
            if c[y, cols - 2] == 0:
                c[y, cols - 1] = 0
        for x in range(cols):
            if c[1, x] == 0:
                c[0, x] = 0
            if c[rows - 2, x] == 0:
                c[rows - 1, x] = 0
        assert np.array_equal(c, cleaned)
        # --- regions of zero pixels: partition, sizes, the small-region rule
        lab, n = ndimage.label(cleaned == 0, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
        sizes = np.bincount(lab.ravel(), minlength=n + 1)
        small = (lab > 0) & (sizes[lab] <= weak_tex_num)
        assert np.array_equal(labels == 0, cleaned == 255)
        assert np.array_equal(labels == -1, small)
        big = (lab > 0) & ~small
        assert big.any() and small.any()
        pairs = np.unique(np.stack([lab[big], labels[big]], 1), axis=0)   # (scipy region, engine label): must be a bijection
        assert len(pairs) == len(np.unique(pairs[:, 0])) == len(np.unique(pairs[:, 1])) and np.all(pairs[:, 1] > 0)
        print("labels scale %d: %d regions, %d kept, %d pixels marked -1, weak_tex_num %d" % (scale, n, len(pairs), int(small.sum()), weak_tex_num))


def test_prior_ratio_map_on_a_hand_built_point_file(tmp_path):
    """The Depth-Anything prior (APD.cpp:1210-1424, host/prior.cpp) on a hand-built sfm/ file of 5 points: the metric depth
    inside the triangulated hull must be (255 - raw) / (barycentric interpolation of the per-point ratio
    raw'/projected depth) — triangulation by scipy.spatial.Delaunay, interpolation in float64, nothing shared with the host
    code —, outside the hull (255 - raw) / rates[n / 2]; the planes' normals must be the finite-difference normals of
    that depth map turned towards the camera and rotated to the world (R^T n)."""
    from scipy.spatial import Delaunay
    W, H = 96, 72
    d = str(tmp_path / "scene")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), "3", "2"], stdout=subprocess.DEVNULL)
    sc = synth.make_scene(W, H, 2)
    cam = sc["cameras"][0]
    K, R, t = cam["K"].astype(np.float64).reshape(3, 3), cam["R"].astype(np.float64).reshape(3, 3), cam["t"].astype(np.float64)
    gt = sc["depth_gt"][0].astype(np.float64)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    scale = 20.0 * (1.0 + 0.3 * xx / W - 0.2 * yy / H)            # the unknown, slowly varying scale of the relative map
    raw = (255.0 - scale * gt).astype(np.float32)
    os.makedirs(os.path.join(d, "dep"), exist_ok=True)
    os.makedirs(os.path.join(d, "sfm"), exist_ok=True)
    _write_binmat(os.path.join(d, "dep", "%08d.dmb" % 0), raw, 5)
    pts = [(12, 10), (80, 14), (70, 60), (15, 55), (44, 33)]       # integer pixels, no four cocircular
    lines, rates = [], []
    for (x, y) in pts:
        z = gt[y, x]
        Xc = np.array([z * (x - K[0, 2]) / K[0, 0], z * (y - K[1, 2]) / K[1, 1], z])
        Xw = R.T @ (Xc - t)
        lines.append("%d %d %.9f %.9f %.9f 10 20 30" % (x, y, Xw[0], Xw[1], Xw[2]))
        rates.append((255.0 - float(raw[y, x])) / z)              # dep'(pixel) / projected depth (the point projects onto its own pixel)
    open(os.path.join(d, "sfm", "%08d.txt" % 0), "w").write("\n".join(lines) + "\n")
    out = _host_tool("--prior", d, 0, W, H, str(tmp_path / "planes.bin"))
    assert out.returncode == 0, out.stdout + out.stderr
    planes = np.fromfile(str(tmp_path / "planes.bin"), np.float32).reshape(H, W, 4)
    dep = planes[:, :, 3].astype(np.float64)

    P = np.array(pts, np.float64)
    tri = Delaunay(P)
    q = np.stack([xx.ravel(), yy.ravel()], 1)
    simplex = tri.find_simplex(q)
    Tm = tri.transform[np.maximum(simplex, 0)]
    b2 = np.einsum("nij,nj->ni", Tm[:, :2, :], q - Tm[:, 2, :])
    bary = np.concatenate([b2, 1 - b2.sum(1, keepdims=True)], 1)
    rate_tri = (bary * np.array(rates)[tri.simplices[np.maximum(simplex, 0)]]).sum(1).reshape(H, W)
    inside = (simplex >= 0).reshape(H, W) & (bary.min(1).reshape(H, W) > 0.04)       # away from the triangle edges
    want_in = (255.0 - raw.astype(np.float64)) / rate_tri
    rel = np.abs(dep - want_in) / want_in
    assert inside.sum() > 1500
    ok = rel[inside] < 2e-5
    print("prior: %d interior pixels, %.4f within 2e-5 of the closed form, worst of those %.2e" % (inside.sum(), ok.mean(), rel[inside][ok].max()))
    assert ok.mean() > 0.995        # the reference's barycentric sweep with truncating casts leaves a few pixels unvisited
    # far outside the hull: the median-position ratio (rates[n / 2], APD.cpp:1279)
    outside = np.zeros((H, W), bool)
    outside[:6, :] = True
    outside[-6:, :] = True
    want_out = (255.0 - raw.astype(np.float64)) / np.float32(rates[len(rates) // 2])
    assert np.abs(dep - want_out)[outside].max() / want_out[outside].max() < 2e-6
    # and the interpolated metric depth is close to the truth where the scale is interpolated linearly
    assert np.median(np.abs(dep - gt)[inside] / gt[inside]) < 0.02
    # normals: finite differences of `dep` in the camera frame, flipped towards the camera, rotated by R^T (APD.cpp:1365-1409)
    Kf = cam["K"].astype(np.float64)

    def X3(x, y):
        z = dep[y, x]
        return np.array([z * (x - Kf[2]) / Kf[0], z * (y - Kf[5]) / Kf[4], z])
    worst = 0.0
    for (x, y) in [(30, 20), (50, 40), (60, 25), (25, 45), (44, 30)]:
        X = X3(x, y)
        n = np.cross(X3(x + 1, y) - X, X3(x, y + 1) - X)
        n /= np.linalg.norm(n)
        if n @ X > 0:
            n = -n
        worst = max(worst, float(np.abs(R.T @ n - planes[y, x, :3]).max()))
    assert worst < 1e-3, worst
    assert np.all(planes[0, :, :3] == 0) and np.all(planes[:, 0, :3] == 0)     # border pixels keep a zero normal
