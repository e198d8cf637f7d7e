"""Wall time of RunFusion (host/fusion.cpp) on a synthetic folder: W x H, NV views, NSRC sources per view, noisy ground-truth
depth / normal maps written as the driver would have left them.   python tools/fusion_timing.py W H NV NSRC [kind]
Prints the time of `tests/host/test_host --fuse` (reading the maps + fusing + writing the .ply) and the point count."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
synth = importlib.import_module("dvp-mvs_amd.synth")


def write_binmat(path, a, typ):
    with open(path, "wb") as f:
        f.write(np.array([1, a.shape[0], a.shape[1], typ], np.int32).tobytes())
        f.write(np.ascontiguousarray(a).tobytes())


def main():
    W, H, NV, NSRC = [int(a) for a in sys.argv[1:5]]
    kind = sys.argv[5] if len(sys.argv) > 5 else None
    keep = os.environ.get("FUSE_KEEP")   # keep the folder there for more runs of `tests/host/test_host --fuse <folder>`
    d = keep or tempfile.mkdtemp(prefix="fuse_")
    if keep: subprocess.call(["rm", "-rf", d])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), str(NSRC)], stdout=subprocess.DEVNULL)
    sc = synth.make_scene(W, H, NV - 1)
    rng = np.random.default_rng(21)
    for v in range(NV):
        dep = sc["depth_gt"][v].astype(np.float64) * (1.0 + rng.normal(0, 0.0012, (H, W)))
        dep[rng.random(dep.shape) < 0.05] = 0.0
        nrm = np.tile(sc["normal_gt"].astype(np.float64), (H, W, 1)) + rng.normal(0, 0.025, (H, W, 3))
        nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
        r = os.path.join(d, "APD", "%08d" % v)
        os.makedirs(r, exist_ok=True)
        write_binmat(os.path.join(r, "depths.dmb"), dep.astype(np.float32), 5)
        write_binmat(os.path.join(r, "APD_normals.dmb"), nrm.astype(np.float32), 21)
        write_binmat(os.path.join(r, "weak.bin"), rng.integers(0, 3, (H, W)).astype(np.uint8), 0)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "dvp-mvs_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "host")])
    t0 = time.time()
    out = subprocess.run([os.path.join(ROOT, "tests", "host", "test_host"), "--fuse", d], capture_output=True, text=True,
                         env=dict(os.environ, DVP_FUSION_KIND=kind or "eth"))
    dt = time.time() - t0
    tail = [l for l in out.stdout.split("\n") if "Fusion" in l or "[fusion]" in l]
    print("RunFusion %dx%d, %d views x %d sources: %.2f s   %s" % (W, H, NV, NSRC, dt, " | ".join(tail)))
    if not keep: subprocess.call(["rm", "-rf", d])


if __name__ == "__main__":
    main()
