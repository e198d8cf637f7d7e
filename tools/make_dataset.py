#!/usr/bin/env python3
"""Write a synthetic scene as an MVSNet-layout dense folder the `apd` driver can consume
(layout of /root/reference/colmap2mvsnet.py:424-469): images/%08d.pgm (+ .ppm), cams/%08d_cam.txt,
pair.txt.  With --prior also the inputs of the FIRST_INIT plane prior (APD.cpp:1210-1424):
dep/%08d.dmb = 255 - s(x,y) * true depth (a stand-in for a Depth-Anything map: right up to a slowly
varying unknown scale) and sfm/%08d.txt = sparse points "x2d y2d X Y Z r g b".
usage: make_dataset.py OUT W H NUM_VIEWS [SRC_PER_VIEW] [--prior] [--jpg] [--torch]
--torch renders on cuda:0 (full-resolution folders: the numpy renderer needs ~30 s per 25 Mpx view)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_cam(path, cam):
    R = cam["R"].reshape(3, 3)
    t = cam["t"]
    K = cam["K"].reshape(3, 3)
    with open(path, "w") as f:
        f.write("extrinsic\n")
        for i in range(3):
            f.write("%.9g %.9g %.9g %.9g\n" % (R[i, 0], R[i, 1], R[i, 2], t[i]))
        f.write("0.0 0.0 0.0 1.0\n\nintrinsic\n")
        for i in range(3):
            f.write("%.9g %.9g %.9g\n" % tuple(K[i]))
        dmin, dmax = float(cam["depth_min"]), float(cam["depth_max"])
        f.write("\n%.9g %.9g %d %.9g\n" % (dmin, (dmax - dmin) / 192.0, 192, dmax))


def write_prior(out, i, cam, depth, rng):
    H, W = depth.shape
    os.makedirs(os.path.join(out, "dep"), exist_ok=True)
    os.makedirs(os.path.join(out, "sfm"), exist_ok=True)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    s = 25.0 * (1.0 + 0.08 * xx / W - 0.05 * yy / H)
    raw = (255.0 - s * depth).astype(np.float32)
    with open(os.path.join(out, "dep", "%08d.dmb" % i), "wb") as f:   # BinMat: version, rows, cols, CV_32FC1
        f.write(np.array([1, H, W, 5], np.int32).tobytes())
        f.write(raw.tobytes())
    K = cam["K"].reshape(3, 3).astype(np.float64)
    R = cam["R"].reshape(3, 3).astype(np.float64)
    t = cam["t"].astype(np.float64)
    npts = max(30, W * H // 250)     # sparse, like SfM keypoints: ~16 px apart (dense integer-grid points make sliver triangles)
    px = rng.integers(3, W - 3, npts)
    py = rng.integers(3, H - 3, npts)
    with open(os.path.join(out, "sfm", "%08d.txt" % i), "w") as f:
        for x, y in zip(px, py):
            Z = float(depth[y, x])
            Xc = np.array([Z * (x - K[0, 2]) / K[0, 0], Z * (y - K[1, 2]) / K[1, 1], Z])
            Xw = R.T @ (Xc - t)
            f.write("%.3f %.3f %.7f %.7f %.7f 128 128 128\n" % (x + 0.2, y + 0.3, Xw[0], Xw[1], Xw[2]))


def main():
    prior = "--prior" in sys.argv
    if prior:
        sys.argv.remove("--prior")
    use_torch = "--torch" in sys.argv
    if use_torch:
        sys.argv.remove("--torch")
    jpg = "--jpg" in sys.argv     # images/%08d.jpg (colour, like colmap2mvsnet.py:424-430 writes) instead of .pgm
    if jpg:
        sys.argv.remove("--jpg")
    out, W, H, NV = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    nsrc = int(sys.argv[5]) if len(sys.argv) > 5 else min(NV - 1, 4)
    synth = importlib.import_module("dvp-mvs_amd.synth")
    if use_torch:
        import torch
        st = synth.make_scene_torch(W, H, NV - 1, torch.device("cuda", 0))
        sc = dict(images=st["images"].cpu().numpy(), depth_gt=st["depth_gt"].cpu().numpy(), cameras=st["cameras"])
        del st
    else:
        sc = synth.make_scene(W, H, NV - 1)
    os.makedirs(os.path.join(out, "images"), exist_ok=True)
    os.makedirs(os.path.join(out, "cams"), exist_ok=True)
    for i in range(NV):
        img = sc["images"][i].astype(np.uint8)
        if jpg:
            from PIL import Image
            Image.fromarray(np.stack([img, img, img], 2), "RGB").save(os.path.join(out, "images", "%08d.jpg" % i), quality=98, subsampling=2)
        else:
            with open(os.path.join(out, "images", "%08d.pgm" % i), "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (W, H))
                f.write(img.tobytes())
        write_cam(os.path.join(out, "cams", "%08d_cam.txt" % i), sc["cameras"][i])
        if prior:
            write_prior(out, i, sc["cameras"][i], sc["depth_gt"][i], np.random.default_rng(100 + i))
    with open(os.path.join(out, "pair.txt"), "w") as f:
        f.write("%d\n" % NV)
        for i in range(NV):
            c = sc["cameras"]["c"]
            d = np.linalg.norm(c - c[i], axis=1)
            order = [j for j in np.argsort(d) if j != i][:nsrc]
            f.write("%d\n%d " % (i, len(order)) + " ".join("%d %.3f" % (j, 100.0 / (1.0 + d[j])) for j in order) + "\n")
    np.save(os.path.join(out, "depth_gt.npy"), sc["depth_gt"])
    print("wrote", out)


if __name__ == "__main__":
    main()
