"""The fusion entry points of include/dvp_mvs.h driven straight through ctypes (the C++ host goes through the same calls):
argument checks, the life cycle of a job, and a case whose answer is known in closed form."""
import ctypes

import numpy as np
import pytest

from conftest import pkg, synth

pytestmark = pytest.mark.gpu


def _lib():
    L = pkg("capi").lib()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.dvp_fuse_create.argtypes = [ci, ci, ctypes.POINTER(vp)]
    L.dvp_fuse_destroy.argtypes = [vp]
    L.dvp_fuse_last_error.restype = ctypes.c_char_p
    L.dvp_fuse_last_error.argtypes = [vp]
    L.dvp_fuse_set_view.argtypes = [vp, ci, vp, ci, ci, vp, vp, vp, vp, vp]
    L.dvp_fuse_view.argtypes = [vp, ci, vp, ci]
    L.dvp_fuse_view_graded.argtypes = [vp, ci, vp, ci, ci]
    L.dvp_fuse_count.restype = ctypes.c_longlong
    L.dvp_fuse_count.argtypes = [vp]
    L.dvp_fuse_download.argtypes = [vp, vp]
    L.dvp_fuse_last_rounds.argtypes = [vp, vp, vp]
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_fusion_job_life_cycle_and_argument_checks():
    L = _lib()
    job = ctypes.c_void_p()
    assert L.dvp_fuse_create(0, 0, ctypes.byref(job)) != 0 and b"bad arguments" in L.dvp_fuse_last_error(None)
    assert L.dvp_fuse_create(0, 3, ctypes.byref(job)) == 0
    W, H = 48, 32
    sc = synth.make_scene(W, H, 2)
    cams = np.ascontiguousarray(sc["cameras"])
    dep = [np.ascontiguousarray(sc["depth_gt"][v], np.float32) for v in range(3)]
    nrm = np.ascontiguousarray(np.tile(sc["normal_gt"].astype(np.float32), (H, W, 1)))
    bgr = np.full((H, W, 3), 77, np.uint8)

    def set_view(v):
        return L.dvp_fuse_set_view(job, v, ctypes.c_void_p(cams.ctypes.data + 112 * v), W, H, _p(dep[v]), _p(nrm), None, _p(bgr), None)
    assert set_view(0) == 0 and set_view(1) == 0
    assert set_view(1) != 0 and b"set before" in L.dvp_fuse_last_error(job)
    assert L.dvp_fuse_set_view(job, 5, _p(cams), W, H, _p(dep[0]), _p(nrm), None, _p(bgr), None) != 0
    src = np.array([1, 2], np.int32)
    assert L.dvp_fuse_view(job, 0, _p(src), 2) != 0 and b"bad source" in L.dvp_fuse_last_error(job)   # view 2 has no maps yet
    assert L.dvp_fuse_view(job, 0, _p(np.array([0], np.int32)), 1) != 0                                # a view is not its own source
    assert L.dvp_fuse_view(job, 0, _p(np.zeros(65, np.int32)), 65) != 0 and b"64" in L.dvp_fuse_last_error(job)
    assert set_view(2) == 0
    assert L.dvp_fuse_count(job) == 0
    assert L.dvp_fuse_view(job, 0, _p(src), 2) == 0
    n0 = L.dvp_fuse_count(job)
    rounds, rest = ctypes.c_int(-1), ctypes.c_int(-1)
    assert L.dvp_fuse_last_rounds(job, ctypes.byref(rounds), ctypes.byref(rest)) == 0 and rounds.value >= 1 and rest.value >= 0
    # true depth maps, one true normal: nearly every pixel of view 0 that both other views see is kept, exactly once
    assert 0.5 * W * H < n0 <= W * H
    assert L.dvp_fuse_view(job, 1, _p(np.array([0, 2], np.int32)), 2) == 0
    assert L.dvp_fuse_view(job, 2, _p(np.array([0, 1], np.int32)), 2) == 0
    n = L.dvp_fuse_count(job)
    # the witnesses of view 0's points were claimed: views 1 and 2 add only what view 0 did not see or could not confirm
    assert n0 <= n < n0 + 0.8 * W * H
    pts = np.zeros((n, 6), np.float32)
    assert L.dvp_fuse_download(job, _p(pts)) == 0
    assert np.isfinite(pts).all() and (pts[:, 3:] == 77).all()          # mean of equal colours
    # the points of view 0 lie on the scene's surface: lift(x, y, depth_gt) — the first n0 records, in raster order
    X = pts[:n0, :3].astype(np.float64)
    cam = sc["cameras"][0]
    R, t, K = cam["R"].astype(np.float64).reshape(3, 3), cam["t"].astype(np.float64), cam["K"].astype(np.float64)
    xc = (R @ X.T).T + t
    u, v = K[0] * xc[:, 0] / xc[:, 2] + K[2], K[4] * xc[:, 1] / xc[:, 2] + K[5]
    assert np.abs(u - np.round(u)).max() < 1e-2 and np.abs(v - np.round(v)).max() < 1e-2          # pixel centres of view 0
    order = np.round(v).astype(np.int64) * W + np.round(u).astype(np.int64)
    assert (np.diff(order) > 0).all()                                                                # scan order
    assert np.abs(xc[:, 2] - dep[0].reshape(-1)[order]).max() < 1e-3
    assert L.dvp_fuse_destroy(job) == 0


def test_graded_fusion_counts_every_listed_source():
    """dvp_fuse_view_graded: a source without maps (-1) takes part in the number of sources the acceptance loop runs to but
    never agrees; with one real source no pixel reaches k = 2 agreeing sources."""
    L = _lib()
    job = ctypes.c_void_p()
    assert L.dvp_fuse_create(0, 2, ctypes.byref(job)) == 0
    W, H = 40, 30
    sc = synth.make_scene(W, H, 1)
    cams = np.ascontiguousarray(sc["cameras"])
    nrm = np.ascontiguousarray(np.tile(sc["normal_gt"].astype(np.float32), (H, W, 1)))
    bgr = np.zeros((H, W, 3), np.uint8)
    for v in range(2):
        dep = np.ascontiguousarray(sc["depth_gt"][v], np.float32)
        assert L.dvp_fuse_set_view(job, v, ctypes.c_void_p(cams.ctypes.data + 112 * v), W, H, _p(dep), _p(nrm), None, _p(bgr), None) == 0
    assert L.dvp_fuse_view_graded(job, 0, _p(np.array([1, -1, -1], np.int32)), 3, 0) == 0
    assert L.dvp_fuse_count(job) == 0
    assert L.dvp_fuse_view_graded(job, 0, _p(np.array([1, 7], np.int32)), 2, 1) != 0      # slot 7 does not exist
    assert L.dvp_fuse_destroy(job) == 0
