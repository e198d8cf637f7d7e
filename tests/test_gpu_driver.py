"""The C++ host mirror end to end on the GPU: `apd` (class APD + schedule of main.cpp) on a
synthetic MVSNet-layout folder; files of the per-pass contract appear, the first pass equals a
C-ABI run with the same seed, the final depth maps are close to ground truth and the fused PLY
has points."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg, synth, make_params, count_diff

pytestmark = pytest.mark.gpu


def read_binmat(path):
    with open(path, "rb") as f:
        ver, rows, cols, typ = np.frombuffer(f.read(16), np.int32)
        assert ver == 1
        dt, ch = {0: (np.uint8, 1), 4: (np.int32, 1), 5: (np.float32, 1), 21: (np.float32, 3)}[int(typ)]
        a = np.frombuffer(f.read(), dt)
    return a.reshape(rows, cols, ch) if ch > 1 else a.reshape(rows, cols)


def test_apd_driver(tmp_path):
    W, H, NV = 192, 144, 4
    d = str(tmp_path / "scene")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "3"])
    apd = os.path.join(ROOT, "dvp-mvs_amd", "apd")
    # W,H <= 800 -> one round at scale 1: FIRST_INIT pass + 1 geom pass
    out = subprocess.run([apd, d, "0", "--iters", "3", "--passes", "1", "--min-scale", "1", "--seed", "77"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    gt = np.load(os.path.join(d, "depth_gt.npy"))
    for v in range(NV):
        r = os.path.join(d, "APD", "%08d" % v)
        for fn in ("depths.dmb", "APD_normals.dmb", "weak.bin", "selected_views.bin", "radius.bin"):
            assert os.path.exists(os.path.join(r, fn)), fn
        dep = read_binmat(os.path.join(r, "depths.dmb"))
        assert dep.shape == (H, W)
        m = dep[10:-10, 10:-10] > 0
        rel = np.abs(dep - gt[v])[10:-10, 10:-10][m] / gt[v][10:-10, 10:-10][m]
        assert m.mean() > 0.7 and np.median(rel) < 1.5e-2, (v, m.mean(), np.median(rel))
        nrm = read_binmat(os.path.join(r, "APD_normals.dmb"))
        assert nrm.shape == (H, W, 3)
        # the last pass also leaves the ACMM-format maps (reference main.cpp:378-385: depths_geom.dmb + normals.dmb)
        for fn, ref, ch in (("depths_geom.dmb", dep, 1), ("normals.dmb", nrm, 3)):
            raw = open(os.path.join(r, fn), "rb").read()
            assert tuple(np.frombuffer(raw[:16], np.int32)) == (1, H, W, ch), fn
            assert np.array_equal(np.frombuffer(raw[16:], np.float32).reshape(ref.shape).view(np.uint32), ref.view(np.uint32)), fn
    ply = os.path.join(d, "APD", "APD.ply")
    assert os.path.exists(ply)
    head = open(ply, "rb").read(200).decode("latin1")
    npts = int(head.split("element vertex ")[1].split("\n")[0])
    assert npts > 1000


def test_apd_with_depth_prior(tmp_path):
    """FIRST_INIT with dep/ + sfm/ inputs (APD.cpp:1210-1424): the prior planes are kept by
    RandomInitialization when their depth is in range, so after a single iteration the depth map is
    already far closer to the truth than from random planes."""
    W, H, NV = 128, 96, 3
    med = {}
    for tag, extra in (("prior", ["--prior"]), ("random", [])):
        d = str(tmp_path / tag)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "2"] + extra)
        out = subprocess.run([os.path.join(ROOT, "dvp-mvs_amd", "apd"), d, "0", "--iters", "1", "--passes", "0", "--min-scale", "1",
                              "--seed", "9", "--no-fusion"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-1500:]
        assert ("Plane prior from dep/ and sfm/" in out.stdout) == (tag == "prior"), out.stdout[-800:]
        gt = np.load(os.path.join(d, "depth_gt.npy"))
        dep = read_binmat(os.path.join(d, "APD", "00000000", "depths.dmb"))
        m = dep[8:-8, 8:-8] > 0
        rel = np.abs(dep - gt[0])[8:-8, 8:-8][m] / gt[0][8:-8, 8:-8][m]
        med[tag] = (float(np.mean(rel < 1e-2)), float(np.mean(rel < 1e-3)), float(m.mean()))
    print(med)
    # With rotated reference cameras the prior helps less than it did with R = I: FIRST_INIT keeps the prior plane as
    # (WORLD normal, depth) and the kernels then read it as (camera normal, offset) — the reference's own behaviour
    # (APD.cpp:1393-1420 + APD.cu:1289-1298, quirk ledger in DESIGN.md) — so its normals are off by the camera rotation.
    assert med["prior"][0] > 0.7 and med["prior"][0] >= med["random"][0] and med["prior"][1] > med["random"][1] + 0.05, med


def test_apd_first_pass_equals_capi(tmp_path):
    """class APD is a thin layer: its FIRST_INIT pass on view 0 must be bit-identical to driving the
    C ABI directly with the same images / cameras / seed."""
    W, H, NV = 96, 64, 3
    d = str(tmp_path / "scene")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "2"])
    apd = os.path.join(ROOT, "dvp-mvs_amd", "apd")
    out = subprocess.run([apd, d, "0", "--iters", "1", "--passes", "0", "--min-scale", "1", "--seed", "5", "--no-fusion"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    dep = read_binmat(os.path.join(d, "APD", "00000000", "depths.dmb"))
    # same pass through the C ABI: view 0 with its pair.txt neighbours, seed = 5 + 0*1000003 + 0
    lines = open(os.path.join(d, "pair.txt")).read().split("\n")
    toks = lines[2].split()
    src = [int(toks[1 + 2 * i]) for i in range(int(toks[0]))]
    sc = synth.make_scene(W, H, NV - 1)
    order = [0] + src
    sub = dict(sc, images=sc["images"][order], cameras=sc["cameras"][order].copy())
    for c in sub["cameras"]:
        pass
    # cameras as the driver sees them: parsed from text (%.9g round-trips float32) with c recomputed
    p = make_params(len(order), max_iterations=1, state=synth.FIRST_INIT, use_APD=0, weak_peak_radius=6)
    g = pkg("capi").from_scene(sub, p, seed=5)
    g.upload_state(planes=np.zeros((H * W, 4), np.float32), radius=np.full(H * W, 5, np.int32))
    g.run_patchmatch()
    planes = g.get("planes")
    d2 = planes[:, 3].reshape(H, W).copy()
    d2[(d2 < p["depth_min"]) | (d2 > p["depth_max"])] = 0
    assert count_diff(dep, d2) == 0


def test_pipeline_on_gpu_equals_oracle_pipeline():
    """ScenePipeline (FIRST_INIT + geom pass with depth exchange) driven by the HIP engine vs the
    same pipeline driven by the oracle: bit-identical per-view planes / views / depths."""
    from oracle import oracle as O
    NV, W, H = 3, 72, 56
    sc = synth.make_scene(W, H, NV - 1)
    c = sc["cameras"]["c"]
    pairs = [[int(j) for j in np.argsort(np.linalg.norm(c - c[i], axis=1)) if j != i][:2] for i in range(NV)]
    capi = pkg("capi")
    pg = pkg("pipeline").ScenePipeline(lambda w, h, ni: capi.Context(w, h, ni), sc["images"], sc["cameras"], pairs, seed=3)
    po = pkg("pipeline").ScenePipeline(lambda w, h, ni: O.Oracle(w, h, ni), sc["images"], sc["cameras"], pairs, seed=3)
    pg.run_round(iters=1, geom_passes=1)
    po.run_round(iters=1, geom_passes=1)
    assert count_diff(pg.depths, po.depths) == 0
    for v in range(NV):
        assert count_diff(pg.state[v]["planes"], po.state[v]["planes"]) == 0
        assert np.array_equal(pg.state[v]["views"], po.state[v]["views"])


def test_apd_exchange_mode(tmp_path):
    """--jacobi: the multi-GPU data flow with one rank — every view's depth map of the previous pass is
    kept resident on the device (DepthExchange / dvp_upload_depths_device) instead of being re-read from
    APD/<id>/depths.dmb, and a pass only sees the previous pass' maps.  Deterministic run to run; the
    FIRST_INIT pass is identical to the in-place mode (no geometric term yet), the geometric passes are not
    (in-place: a view sees the maps its predecessors wrote in the same pass)."""
    W, H, NV = 160, 120, 4
    outs = {}
    for tag, extra in (("inplace", []), ("jacobi_a", ["--jacobi"]), ("jacobi_b", ["--jacobi"])):
        d = str(tmp_path / tag)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "3"])
        out = subprocess.run([os.path.join(ROOT, "dvp-mvs_amd", "apd"), d, "0", "--iters", "2", "--passes", "2", "--min-scale", "1", "--seed", "5",
                              "--no-fusion"] + extra, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        outs[tag] = [read_binmat(os.path.join(d, "APD", "%08d" % v, "depths.dmb")) for v in range(NV)]
        gt = np.load(os.path.join(d, "depth_gt.npy"))
    for v in range(NV):
        assert np.array_equal(outs["jacobi_a"][v], outs["jacobi_b"][v])
        dep = outs["jacobi_a"][v]
        m = dep[10:-10, 10:-10] > 0
        rel = np.abs(dep - gt[v])[10:-10, 10:-10][m] / gt[v][10:-10, 10:-10][m]
        assert m.mean() > 0.7 and np.median(rel) < 1.5e-2
    # view 0 is processed first in both modes and sees only previous-pass maps either way
    assert not all(np.array_equal(outs["inplace"][v], outs["jacobi_a"][v]) for v in range(1, NV))


def test_apd_jpeg_folder_with_label_files(tmp_path):
    """A folder in the converter's own format (images/%08d.jpg, 4:2:0 colour JPEG) runs unmodified through
    the built-in decoder; --labels makes SupportInitialization load the labels_<s>.dmb that GetProblemEdges
    writes (Roberts + components + Hough, host/labels.cpp)."""
    W, H, NV = 192, 144, 4
    d = str(tmp_path / "scene")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "3", "--jpg"])
    assert not [f for f in os.listdir(os.path.join(d, "images")) if not f.endswith(".jpg")]
    out = subprocess.run([os.path.join(ROOT, "dvp-mvs_amd", "apd"), d, "0", "--iters", "3", "--passes", "1", "--min-scale", "1", "--seed", "3", "--labels"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    gt = np.load(os.path.join(d, "depth_gt.npy"))
    for v in range(NV):
        r = os.path.join(d, "APD", "%08d" % v)
        lab = read_binmat(os.path.join(r, "labels_0.dmb"))
        assert lab.shape == (H, W) and lab.dtype == np.int32 and lab.min() >= -1
        dep = read_binmat(os.path.join(r, "depths.dmb"))
        m = dep[10:-10, 10:-10] > 0
        rel = np.abs(dep - gt[v])[10:-10, 10:-10][m] / gt[v][10:-10, 10:-10][m]
        assert m.mean() > 0.7 and np.median(rel) < 2e-2, (v, m.mean(), np.median(rel))
    assert os.path.exists(os.path.join(d, "APD", "APD.ply"))


def _apd(d, *args, **kw):
    return subprocess.Popen([os.path.join(ROOT, "dvp-mvs_amd", "apd"), d, "0"] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)


def test_apd_two_ranks_equal_single_rank_jacobi(tmp_path):
    """world = 2 through the WHOLE multi-rank code path of the driver — rendezvous file, view v -> rank v % 2, every rank
    decoding its own share of the images and broadcasting it (ShareLevelImages), the per-pass DepthExchange with per-view
    dimensions, AllOk, atomic result files — against the single-rank --jacobi run of the same folder: every view's depth
    map bit-identical.  Two processes share the one GPU of the box, so the collectives run over the host transport
    (--transport host: TCP, star through rank 0); RCCL refuses two ranks on one device, and a multi-GPU box has never
    been available.  Only the ncclBroadcast / ncclAllReduce calls themselves stay unexercised."""
    W, H, NV = 128, 96, 5
    outs = {}
    for tag in ("single", "world2"):
        d = str(tmp_path / tag)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "3"])
        common = ["--iters", "2", "--passes", "2", "--min-scale", "1", "--seed", "5", "--no-fusion"]
        if tag == "single":
            procs = [_apd(d, "--jacobi", *common)]
        else:
            # a stale rendezvous record of an "earlier run" with the same job id must not be picked up
            os.makedirs(os.path.join(d, "APD"), exist_ok=True)
            with open(os.path.join(d, "APD", ".rccl_id"), "wb") as f:
                f.write(b"dvp-host-id job42\n" + b"127.0.0.1:1\0" + bytes(52))
            os.utime(os.path.join(d, "APD", ".rccl_id"), (1, 1))
            procs = [_apd(d, "--rank", str(r), "--world", "2", "--job", "job42", "--transport", "host", "--collective-timeout", "120", *common) for r in (1, 0)]
        for p in procs:
            so, se = p.communicate(timeout=900)
            assert p.returncode == 0, so[-1500:] + se[-1500:]
        outs[tag] = [read_binmat(os.path.join(d, "APD", "%08d" % v, "depths.dmb")) for v in range(NV)]
        assert not os.path.exists(os.path.join(d, "APD", ".rccl_id"))
    for v in range(NV):
        assert np.array_equal(outs["single"][v], outs["world2"][v]), v


def test_apd_eight_ranks_host_transport(tmp_path):
    """world = 8 — the rank count of an MI355X node — without the node: eight `apd` processes on the one GPU of the box over
    the host transport, 17 views of unequal size (three image files are cropped: per-view dimensions in the depth exchange,
    and views whose sources have another size rescale the exchanged previous-pass map — reading the owner's depths.dmb instead
    gave the previous or the current pass' map depending on timing, which this test caught — ADVICE r03).  Every depth map bit-identical to the single-rank --jacobi run;
    every rank reports its host-thread budget = cores / 8 (SetHostThreadShare), not 32 each.  The RCCL calls stay
    "unmeasured on hardware"."""
    import re
    import time
    W, H, NV = 96, 72, 17
    outs, wall = {}, {}
    for tag in ("single", "world8"):
        d = str(tmp_path / tag)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "4"], stdout=subprocess.DEVNULL)
        sc = synth.make_scene(W, H, NV - 1)
        for v, (w, h) in ((2, (80, 72)), (9, (96, 56)), (14, (88, 64))):
            with open(os.path.join(d, "images", "%08d.pgm" % v), "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (w, h))
                f.write(sc["images"][v][:h, :w].astype(np.uint8).tobytes())
        common = ["--iters", "1", "--passes", "2", "--min-scale", "1", "--seed", "9", "--no-fusion"]
        t0 = time.time()
        if tag == "single":
            procs = [_apd(d, "--jacobi", *common)]
        else:
            procs = [_apd(d, "--rank", str(r), "--world", "8", "--job", "job8", "--transport", "host", "--collective-timeout", "300", *common) for r in range(7, -1, -1)]
        texts = []
        for p in procs:
            so, se = p.communicate(timeout=1500)
            assert p.returncode == 0, so[-1500:] + se[-1500:]
            texts.append(so)
        wall[tag] = time.time() - t0
        if tag == "world8":
            cores = os.cpu_count()
            seen = sorted((int(m.group(1)), int(m.group(2))) for t in texts for m in re.finditer(r"rank (\d+) of 8: (\d+) host threads", t))
            assert [r for r, _ in seen] == list(range(8)), seen
            if "DVP_HOST_THREADS" not in os.environ:
                assert all(n == max(1, min(32, cores // 8)) for _, n in seen), (seen, cores)
            # view -> rank: longest-predicted-first on the pixel counts of the image headers (host/main.cpp: AssignViews) — the 14
            # full-size views first, then the cropped ones (14: 88x64, 9: 96x56, 2: 80x72) onto the least-loaded ranks; every rank
            # prints the same table's row for itself
            owned = {int(m.group(1)): [int(x) for x in m.group(2).split()] for t in texts for m in re.finditer(r"rank (\d+) owns views((?: \d+)+)", t)}
            px = {v: W * H for v in range(NV)}
            px.update({2: 80 * 72, 9: 96 * 56, 14: 88 * 64})
            load, want = [0] * 8, {r: [] for r in range(8)}
            for v in sorted(range(NV), key=lambda i: (-px[i], i)):
                r = min(range(8), key=lambda q: (load[q], q))
                want[r].append(v)
                load[r] += px[v]
            assert {r: sorted(v) for r, v in owned.items()} == {r: sorted(v) for r, v in want.items()}, (owned, want)
            assert owned != {r: [v for v in range(NV) if v % 8 == r] for r in range(8)}   # (not round-robin on this folder)
        outs[tag] = [read_binmat(os.path.join(d, "APD", "%08d" % v, "depths.dmb")) for v in range(NV)]
    print("apd 17 views, whole schedule (--passes 2): single rank %.1f s, 8 ranks on one GPU (host transport) %.1f s" % (wall["single"], wall["world8"]))
    shapes = {o.shape for o in outs["single"]}
    assert len(shapes) == 4, shapes
    for v in range(NV):
        assert outs["single"][v].shape == outs["world8"][v].shape and np.array_equal(outs["single"][v], outs["world8"][v]), v


def test_run_scenes_two_in_flight_equals_one_after_the_other(tmp_path):
    """tools/run_scenes.py (BASELINE cfg4: several scenes over the GPUs of a node): two scenes IN FLIGHT on the same GPU(s) — the
    way a node's ranks fill the waits at the per-pass exchange of one scene with the views of another — leave the files of the
    scenes run one after the other: a scene's result does not depend on what else runs.  On the one GPU of the box that is
    two `apd` processes sharing the device; the mode with one GPU per scene is exercised with its queue of one worker."""
    scenes = []
    for i, (W, H, NV) in enumerate(((112, 80, 4), (96, 64, 5), (128, 72, 3))):
        for tag in ("seq", "par", "queue"):
            d = str(tmp_path / ("%s_scene%d" % (tag, i)))
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "2"], stdout=subprocess.DEVNULL)
        scenes.append((W, H, NV))
    common = ["--iters", "1", "--passes", "1", "--min-scale", "1", "--seed", "21", "--no-fusion"]
    for i in range(len(scenes)):
        so, se = _apd(str(tmp_path / ("seq_scene%d" % i)), *common).communicate(timeout=600)
    tool = os.path.join(ROOT, "tools", "run_scenes.py")
    for tag, extra in (("par", ["--mode", "node", "--in-flight", "2"]), ("queue", ["--mode", "scenes"])):
        out = subprocess.run([sys.executable, tool, "--gpus", "1", "--log-dir", str(tmp_path / ("logs_" + tag))] + extra +
                             [str(tmp_path / ("%s_scene%d" % (tag, i))) for i in range(len(scenes))] + ["--"] + common, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        assert "3 scene(s)" in out.stdout
        for i, (W, H, NV) in enumerate(scenes):
            for v in range(NV):
                a = read_binmat(os.path.join(str(tmp_path / ("seq_scene%d" % i)), "APD", "%08d" % v, "depths.dmb"))
                b = read_binmat(os.path.join(str(tmp_path / ("%s_scene%d" % (tag, i))), "APD", "%08d" % v, "depths.dmb"))
                assert a.shape == (H, W) and np.array_equal(a, b), (tag, i, v)


def test_run_scenes_two_ranks_per_scene_two_scenes_in_flight(tmp_path):
    """The cfg4 way of running a node — every scene on all ranks, two scenes in flight — with TWO ranks per scene (host transport:
    both on the box's one GPU, four `apd` processes at a time): the files of each scene are those of its single-rank `--jacobi`
    run (the same data flow with one rank)."""
    scenes = []
    for i, (W, H, NV) in enumerate(((112, 80, 5), (96, 64, 4), (104, 72, 3))):
        for tag in ("one", "two"):
            d = str(tmp_path / ("%s_scene%d" % (tag, i)))
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "2"], stdout=subprocess.DEVNULL)
        scenes.append((W, H, NV))
    common = ["--iters", "1", "--passes", "2", "--min-scale", "1", "--seed", "33", "--no-fusion"]
    for i in range(len(scenes)):
        so, se = _apd(str(tmp_path / ("one_scene%d" % i)), "--jacobi", *common).communicate(timeout=600)
    tool = os.path.join(ROOT, "tools", "run_scenes.py")
    out = subprocess.run([sys.executable, tool, "--gpus", "2", "--gpu-map", "0,0", "--transport", "host", "--mode", "node", "--in-flight", "2",
                          "--log-dir", str(tmp_path / "logs")] + [str(tmp_path / ("two_scene%d" % i)) for i in range(len(scenes))] + ["--"] + common,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "3 scene(s)" in out.stdout and "2 in flight" in out.stdout
    for i, (W, H, NV) in enumerate(scenes):
        for v in range(NV):
            for name in ("depths.dmb", "APD_normals.dmb"):
                a = read_binmat(os.path.join(str(tmp_path / ("one_scene%d" % i)), "APD", "%08d" % v, name))
                b = read_binmat(os.path.join(str(tmp_path / ("two_scene%d" % i)), "APD", "%08d" % v, name))
                assert a.shape[:2] == (H, W) and np.array_equal(a, b), (i, v, name)


def test_apd_rank_failure_takes_the_job_down(tmp_path):
    """A rank that hits a fatal error (here: the image of one of ITS views is unreadable) must not leave its peer hanging
    in a collective: DvpFatal -> RankComm::Abort drops the .abort marker, the peer's wait sees it and exits non-zero too."""
    import time
    W, H, NV = 96, 64, 4
    d = str(tmp_path / "scene")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "2"])
    with open(os.path.join(d, "images", "%08d.pgm" % 1), "wb") as f:      # view 1 belongs to rank 1
        f.write(b"not an image")
    t0 = time.time()
    procs = [_apd(d, "--rank", str(r), "--world", "2", "--job", "jobX", "--transport", "host", "--collective-timeout", "300",
                  "--iters", "1", "--passes", "1", "--min-scale", "1", "--no-fusion") for r in (0, 1)]
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode != 0, so[-800:]
    assert time.time() - t0 < 100           # far below the collective timeout: the marker, not the watchdog
    assert "rank 1" in open(os.path.join(d, "APD", ".rccl_id.abort")).read()
    # and without a job id a multi-rank run refuses to start
    p = _apd(d, "--rank", "0", "--world", "2", "--transport", "host", env={k: v for k, v in os.environ.items() if k not in ("DVP_JOB_ID", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID")})
    so, se = p.communicate(timeout=60)
    assert p.returncode != 0 and "--job" in se


def test_apd_source_image_smaller_than_the_reference(tmp_path):
    """APD.cpp:1066-1082: a source image of another size is copied into a zero image of the reference's size (cropped
    if larger, zero-padded if smaller).  View 2's file is 24 columns / 16 rows short; view 0 uses it as a source: the
    driver's first pass on view 0 must equal a C-ABI run on the explicitly zero-padded image, bit for bit."""
    W, H, NV = 96, 64, 3
    d = str(tmp_path / "scene")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "2"])
    sc = synth.make_scene(W, H, NV - 1)
    small = sc["images"][2][:H - 16, :W - 24].astype(np.uint8)
    with open(os.path.join(d, "images", "%08d.pgm" % 2), "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (W - 24, H - 16))
        f.write(small.tobytes())
    out = subprocess.run([os.path.join(ROOT, "dvp-mvs_amd", "apd"), d, "0", "--iters", "1", "--passes", "0", "--min-scale", "1", "--seed", "5", "--no-fusion"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    dep = read_binmat(os.path.join(d, "APD", "00000000", "depths.dmb"))
    toks = open(os.path.join(d, "pair.txt")).read().split("\n")[2].split()
    src = [int(toks[1 + 2 * i]) for i in range(int(toks[0]))]
    assert 2 in src
    padded = sc["images"].copy()
    padded[2] = 0
    padded[2][:H - 16, :W - 24] = small
    order = [0] + src
    sub = dict(sc, images=padded[order], cameras=sc["cameras"][order].copy())
    p = make_params(len(order), max_iterations=1, state=synth.FIRST_INIT, use_APD=0, weak_peak_radius=6)
    g = pkg("capi").from_scene(sub, p, seed=5)
    g.upload_state(planes=np.zeros((H * W, 4), np.float32), radius=np.full(H * W, 5, np.int32))
    g.run_patchmatch()
    d2 = g.get("planes")[:, 3].reshape(H, W).copy()
    d2[(d2 < p["depth_min"]) | (d2 > p["depth_max"])] = 0
    assert count_diff(dep, d2) == 0


@pytest.mark.parametrize("W,H,NV,levels", [(160, 120, 4, 1), (838, 126, 3, 2)])
def test_apd_result_cache_and_background_worker_leave_the_same_files(tmp_path, W, H, NV, levels):
    """The driver's write-back result cache, background finisher (visibility clean-up + writes behind the next view),
    resident depth maps, decode prefetch (host/store.cpp, main.cpp) and — second case, two pyramid levels — the device-side
    up-sampling of the coarser level's maps (dvp_upload_state_rescaled; 419x63 -> 838x126) against the synchronous file
    flow of the reference (--sync-io: every pass re-reads its inputs from the files the previous one wrote and rescales
    them on the host): every result file byte-identical.  (Until round 4 this test started both runs with the same
    command line — the flag was never passed.)"""
    import filecmp
    outs = {}
    for tag, extra in (("async", []), ("sync", ["--sync-io"])):
        d = str(tmp_path / tag)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "3", "--jpg"])
        out = subprocess.run([os.path.join(ROOT, "dvp-mvs_amd", "apd"), d, "0", "--iters", "2", "--passes", "2", "--min-scale", "1", "--seed", "11", "--labels"] + extra,
                             capture_output=True, text=True, timeout=600)
        if tag == "async" and levels > 1:
            assert out.stdout.count("Weak count") > 0
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        outs[tag] = d
    n = 0
    for v in range(NV):
        ra, rs = (os.path.join(outs[t], "APD", "%08d" % v) for t in ("async", "sync"))
        names = sorted(os.listdir(rs))
        assert names == sorted(os.listdir(ra)), (names, sorted(os.listdir(ra)))
        for fn in names:
            assert not fn.endswith(".part")
            assert filecmp.cmp(os.path.join(ra, fn), os.path.join(rs, fn), shallow=False), (v, fn)
            n += 1
    assert n >= NV * 6
    assert filecmp.cmp(os.path.join(outs["async"], "APD", "APD.ply"), os.path.join(outs["sync"], "APD", "APD.ply"), shallow=False)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["inplace", "jacobi"])
def test_apd_views_in_flight_leave_the_same_files(tmp_path, mode):
    """Several views of a pass at once, each on its own engine context and driver thread (main.cpp: photometric passes in the
    reference's in-place order, every pass with the depth exchange), against one view after the other: every result file
    byte-identical, the views' log blocks whole (one 'Processing image ... done!' per view and pass, never interleaved)."""
    import filecmp
    W, H, NV = 838, 126, 5
    outs = {}
    for tag, extra in (("one", ["--views-in-flight", "1"]), ("three", ["--views-in-flight", "3"]), ("default", [])):
        d = str(tmp_path / tag)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_dataset.py"), d, str(W), str(H), str(NV), "3", "--jpg"])
        out = subprocess.run([os.path.join(ROOT, "dvp-mvs_amd", "apd"), d, "0", "--iters", "2", "--passes", "1", "--min-scale", "1", "--seed", "7", "--no-fusion"]
                             + (["--jacobi"] if mode == "jacobi" else []) + extra,
                             capture_output=True, text=True, timeout=600, env=dict(os.environ, DVP_HOST_TIMING="1"))
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        # (the pass lines by pattern: a background job's timing line may land inside another line of the log)
        import re
        passes = re.findall(r"Pass \d+: \d+ views in [0-9.e+-]+ ms, \d+ in flight", out.stdout)
        assert len(passes) == 4, passes      # two levels x (photometric + one geometric pass)
        flights = [int(re.search(r"(\d+) in flight", ln).group(1)) for ln in passes]
        if tag == "one":
            assert flights == [1, 1, 1, 1]
        elif mode == "jacobi":
            assert flights == ([3] * 4 if tag == "three" else [2] * 4), passes
        else:
            assert flights == ([3, 1, 3, 1] if tag == "three" else [2, 1, 2, 1]), passes
        # a view's block starts with its 'Processing image: N...' line and ends with 'Cost time'; blocks do not interleave
        depth = 0
        for ln in out.stdout.split("\n"):
            if ln.startswith("Processing image:") and ln.endswith("..."):
                assert depth == 0, ln
                depth = 1
            elif ln.startswith("Cost time:"):
                assert depth == 1, ln
                depth = 0
        outs[tag] = d
    for tag in ("three", "default"):
        for v in range(NV):
            ra, rb = (os.path.join(outs[t], "APD", "%08d" % v) for t in ("one", tag))
            names = sorted(os.listdir(ra))
            assert names == sorted(os.listdir(rb))
            for fn in names:
                assert filecmp.cmp(os.path.join(ra, fn), os.path.join(rb, fn), shallow=False), (tag, v, fn)
