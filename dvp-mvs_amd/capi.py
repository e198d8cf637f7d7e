"""ctypes binding of the engine's C ABI (include/dvp_mvs.h -> libdvp_mvs_hip.so).

This is harness glue for tests and bench.py; the product is the shared library.  Loading fails
loudly when the HIP library is missing or cannot be built — there is no CPU fallback.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("DVP_MVS_LIB") or os.path.join(_HERE, "libdvp_mvs_hip.so")   # override: A/B builds
_LIB = None

STAGES = dict(gen_edge_inform=0, find_nearest_strong=1, gen_neighbours=2, neighbour_update=3,
              random_init=4, strong_update=5, ransac_fit=6, weak_update=7, get_depth_normal=8,
              filter_strong=9, depth_to_weak=10, local_refine=11)
STAGE_NAMES = {v: k for k, v in STAGES.items()}
STAGE_NAMES[12] = "strong_prep"   # timing bucket only (snapshot copies + sample search of every strong_update), not launchable
N_TIMING = 13
BUFFERS = dict(planes=(0, np.float32, 4), costs=(1, np.float32, 1), selected_views=(2, np.uint32, 1),
               view_weight=(3, np.uint8, 32), weak_info=(4, np.uint8, 1), weak_reliable=(5, np.uint8, 1),
               weak_nearest_strong=(6, np.int16, 2), neighbours_map=(7, np.int32, 1),
               neighbours=(8, np.int16, 2), fit_planes=(9, np.float32, 4), candidate=(10, np.int16, 2),
               edge=(11, np.uint8, 1), edge_neigh=(12, np.int16, 2), label=(13, np.int32, 1),
               label_boundary=(14, np.int16, 2), complex=(15, np.float32, 1), radius=(16, np.int32, 1))

# every symbol include/dvp_mvs.h declares
EXPORTS = ["dvp_ctx_create", "dvp_ctx_destroy", "dvp_ctx_reserve", "dvp_last_error", "dvp_upload_images", "dvp_upload_depths",
           "dvp_upload_images_device", "dvp_upload_depths_device", "dvp_upload_cameras", "dvp_upload_state", "dvp_upload_state_rescaled",
           "dvp_reset_state", "dvp_save_state", "dvp_restore_state", "dvp_set_params", "dvp_set_seed", "dvp_set_sampler", "dvp_set_profiling", "dvp_image_format", "dvp_run_patchmatch",
           "dvp_run_stage", "dvp_synchronize", "dvp_download_state", "dvp_download_maps", "dvp_download_maps_begin", "dvp_download_maps_finish", "dvp_buffer_bytes", "dvp_download_buffer",
           "dvp_upload_buffer", "dvp_weak_count", "dvp_get_timings", "dvp_reset_timings", "dvp_eval_cost_vectors",
           "dvp_bench_cost_kernel", "dvp_build_id",
           "dvp_fuse_create", "dvp_fuse_destroy", "dvp_fuse_last_error", "dvp_fuse_set_view", "dvp_fuse_view", "dvp_fuse_view_graded", "dvp_fuse_count", "dvp_fuse_download", "dvp_fuse_last_rounds"]


class DvpTimings(ctypes.Structure):
    _fields_ = [("stage_ms", ctypes.c_double * 13), ("stage_launches", ctypes.c_int32 * 13),
                ("iter_loop_ms", ctypes.c_double), ("total_ms", ctypes.c_double),
                ("ncc_evals", ctypes.c_uint64 * 13)]


def build(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        args.insert(1, "-B")
    subprocess.check_call(args)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)   # raises OSError if the HIP runtime / library is missing
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.dvp_ctx_create.argtypes = [ci, ci, ci, ci, ctypes.POINTER(vp)]
        L.dvp_ctx_destroy.argtypes = [vp]
        L.dvp_ctx_reserve.argtypes = [vp, ci, ci]
        L.dvp_last_error.restype = ctypes.c_char_p
        L.dvp_last_error.argtypes = [vp]
        for n in ("dvp_upload_images", "dvp_upload_depths", "dvp_upload_images_device", "dvp_upload_depths_device"):
            getattr(L, n).argtypes = [vp, vp, ci]
        L.dvp_upload_cameras.argtypes = [vp, vp, ci]
        L.dvp_upload_state.argtypes = [vp] * 7
        L.dvp_upload_state_rescaled.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp]
        L.dvp_reset_state.argtypes = [vp]
        L.dvp_save_state.argtypes = [vp]
        L.dvp_restore_state.argtypes = [vp]
        L.dvp_set_params.argtypes = [vp, vp]
        L.dvp_set_seed.argtypes = [vp, ctypes.c_uint64]
        L.dvp_set_sampler.argtypes = [vp, ci]
        L.dvp_set_profiling.argtypes = [vp, ci]
        L.dvp_image_format.argtypes = [vp]
        L.dvp_run_patchmatch.argtypes = [vp]
        L.dvp_run_stage.argtypes = [vp, ci, ci, ci]
        L.dvp_synchronize.argtypes = [vp]
        L.dvp_download_state.argtypes = [vp] * 5
        L.dvp_download_maps.argtypes = [vp] * 6
        L.dvp_download_maps_begin.argtypes = [vp, vp]
        L.dvp_download_maps_finish.argtypes = [vp] * 6
        L.dvp_buffer_bytes.restype = ctypes.c_longlong
        L.dvp_buffer_bytes.argtypes = [vp, ci]
        L.dvp_download_buffer.argtypes = [vp, ci, vp]
        L.dvp_upload_buffer.argtypes = [vp, ci, vp]
        L.dvp_weak_count.argtypes = [vp]
        L.dvp_get_timings.argtypes = [vp, ctypes.POINTER(DvpTimings)]
        L.dvp_reset_timings.argtypes = [vp]
        L.dvp_eval_cost_vectors.argtypes = [vp, vp, vp, ci, vp, ctypes.POINTER(ctypes.c_float)]
        L.dvp_build_id.restype = ctypes.c_char_p
        L.dvp_build_id.argtypes = []
        L.dvp_bench_cost_kernel.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint64)]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class DvpError(RuntimeError):
    pass


class Context:
    """One reference view on one GPU (dvp_ctx).  Method names follow oracle.oracle.Oracle so the
    parity tests drive both through the same code."""

    def __init__(self, width, height, num_images, device=0):
        self.L = lib()
        self.W, self.H, self.NI = width, height, num_images
        h = ctypes.c_void_p()
        if self.L.dvp_ctx_create(device, width, height, num_images, ctypes.byref(h)) != 0:
            raise DvpError(self.L.dvp_last_error(None).decode())
        self.h = h

    def _ck(self, rc):
        if rc != 0:
            raise DvpError(self.L.dvp_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.dvp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _planes_ptrs(self, imgs):
        arrs = [np.ascontiguousarray(imgs[i], np.float32) for i in range(self.NI)]
        for a in arrs:
            assert a.shape == (self.H, self.W)
        ptrs = (ctypes.c_void_p * self.NI)(*[a.ctypes.data for a in arrs])
        return arrs, ptrs

    def set_images(self, images):
        arrs, ptrs = self._planes_ptrs(images)
        self._ck(self.L.dvp_upload_images(self.h, ptrs, self.W))

    def set_depths(self, depths):
        arrs, ptrs = self._planes_ptrs(depths)
        self._ck(self.L.dvp_upload_depths(self.h, ptrs, self.W))

    def set_images_device(self, dev_ptrs, pitch_floats):
        ptrs = (ctypes.c_void_p * self.NI)(*dev_ptrs)
        self._ck(self.L.dvp_upload_images_device(self.h, ptrs, pitch_floats))

    def set_depths_device(self, dev_ptrs, pitch_floats):
        ptrs = (ctypes.c_void_p * self.NI)(*dev_ptrs)
        self._ck(self.L.dvp_upload_depths_device(self.h, ptrs, pitch_floats))

    def set_cameras(self, cams):
        a = np.ascontiguousarray(cams)
        assert a.dtype.itemsize == 112
        self._ck(self.L.dvp_upload_cameras(self.h, _p(a), len(a)))

    def set_params(self, params):
        a = np.ascontiguousarray(params).reshape(1)
        assert a.dtype.itemsize == 76
        self._ck(self.L.dvp_set_params(self.h, _p(a)))
        self.params = a

    def reserve(self, weak_pixels=0, flags=3):
        """optional buffers ahead of their first use (include/dvp_mvs.h: dvp_ctx_reserve)"""
        self._ck(self.L.dvp_ctx_reserve(self.h, int(weak_pixels), int(flags)))

    def set_seed(self, seed):
        self._ck(self.L.dvp_set_seed(self.h, seed))

    def set_sampler(self, s):
        self._ck(self.L.dvp_set_sampler(self.h, s))

    def set_profiling(self, on):
        self._ck(self.L.dvp_set_profiling(self.h, int(on)))

    def upload_state(self, planes=None, views=None, weak=None, edge=None, label=None, radius=None):
        c = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        args = [c(planes, np.float32), c(views, np.uint32), c(weak, np.uint8), c(edge, np.uint8), c(label, np.int32), c(radius, np.int32)]
        self._ck(self.L.dvp_upload_state(self.h, *[_p(a) for a in args]))

    def upload_state_rescaled(self, src_w, src_h, depth, normal, views, weak=None, radius=None, radius_fallback=5, edge=None, label=None):
        """the coarser level's maps at their own size, up-sampled on the device (include/dvp_mvs.h)"""
        c = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        args = [c(depth, np.float32), c(normal, np.float32), c(views, np.uint32), c(weak, np.uint8), c(radius, np.int32)]
        tail = [c(edge, np.uint8), c(label, np.int32)]
        self._ck(self.L.dvp_upload_state_rescaled(self.h, int(src_w), int(src_h), *[_p(a) for a in args], int(radius_fallback), *[_p(a) for a in tail]))

    def reset_state(self):
        self._ck(self.L.dvp_reset_state(self.h))

    def image_format(self):
        """0: float planes, 1: byte planes (8-bit exact image set) — include/dvp_mvs.h dvp_image_format"""
        return int(self.L.dvp_image_format(self.h))

    def save_state(self):
        self._ck(self.L.dvp_save_state(self.h))

    def restore_state(self):
        self._ck(self.L.dvp_restore_state(self.h))

    def download_state(self):
        L = self.W * self.H
        planes = np.empty((L, 4), np.float32)
        views = np.empty(L, np.uint32)
        weak = np.empty(L, np.uint8)
        radius = np.empty(L, np.int32)
        self._ck(self.L.dvp_download_state(self.h, _p(planes), _p(views), _p(weak), _p(radius)))
        return planes, views, weak, radius

    def download_maps(self):
        """depth / normal / selected_views / states / radius as the driver stores them (include/dvp_mvs.h)"""
        L = self.W * self.H
        depth = np.empty(L, np.float32)
        normal = np.empty((L, 3), np.float32)
        views = np.empty(L, np.uint32)
        weak = np.empty(L, np.uint8)
        radius = np.empty(L, np.int32)
        self._ck(self.L.dvp_download_maps(self.h, _p(depth), _p(normal), _p(views), _p(weak), _p(radius)))
        return depth, normal, views, weak, radius

    def download_maps_begin(self, depth_device_copy=None):
        """first step of download_maps: the maps are staged on the device (and the depth map copied to the device address
        `depth_device_copy`, if given); the context may be reset / uploaded to / run again before download_maps_finish"""
        self._ck(self.L.dvp_download_maps_begin(self.h, ctypes.c_void_p(depth_device_copy) if depth_device_copy else None))

    def download_maps_finish(self):
        """second step (any thread): the staged maps -> host"""
        L = self.W * self.H
        depth = np.empty(L, np.float32)
        normal = np.empty((L, 3), np.float32)
        views = np.empty(L, np.uint32)
        weak = np.empty(L, np.uint8)
        radius = np.empty(L, np.int32)
        self._ck(self.L.dvp_download_maps_finish(self.h, _p(depth), _p(normal), _p(views), _p(weak), _p(radius)))
        return depth, normal, views, weak, radius

    def get(self, name):
        bid, dt, k = BUFFERS[name]
        nbytes = self.L.dvp_buffer_bytes(self.h, bid)
        out = np.empty(nbytes // np.dtype(dt).itemsize, dt)
        self._ck(self.L.dvp_download_buffer(self.h, bid, _p(out)))
        return out.reshape(-1, k) if k > 1 else out

    def set(self, name, arr):
        bid, dt, k = BUFFERS[name]
        a = np.ascontiguousarray(arr, dt)
        assert a.nbytes == self.L.dvp_buffer_bytes(self.h, bid), (name, a.nbytes)
        self._ck(self.L.dvp_upload_buffer(self.h, bid, _p(a)))

    def weak_count(self):
        return self.L.dvp_weak_count(self.h)

    def run_stage(self, name, it=0, colour=0):
        self._ck(self.L.dvp_run_stage(self.h, STAGES[name], it, colour))

    def run_patchmatch(self):
        self._ck(self.L.dvp_run_patchmatch(self.h))

    def synchronize(self):
        self._ck(self.L.dvp_synchronize(self.h))

    def timings(self, reset=False):
        t = DvpTimings()
        self._ck(self.L.dvp_get_timings(self.h, ctypes.byref(t)))
        out = dict(stage_ms={STAGE_NAMES[i]: t.stage_ms[i] for i in range(N_TIMING)},
                   stage_launches={STAGE_NAMES[i]: t.stage_launches[i] for i in range(N_TIMING)},
                   ncc_evals={STAGE_NAMES[i]: t.ncc_evals[i] for i in range(N_TIMING)},
                   iter_loop_ms=t.iter_loop_ms, total_ms=t.total_ms)
        if reset:
            self._ck(self.L.dvp_reset_timings(self.h))
        return out

    def eval_cost_vectors(self, px, planes):
        px = np.ascontiguousarray(px, np.int32)
        planes = np.ascontiguousarray(planes, np.float32)
        n = len(px)
        out = np.empty((n, self.NI - 1), np.float32)
        ms = ctypes.c_float(0)
        self._ck(self.L.dvp_eval_cost_vectors(self.h, _p(px), _p(planes), n, _p(out), ctypes.byref(ms)))
        self.last_kernel_ms = ms.value
        return out

    def bench_cost_kernel(self, repeat=5):
        ms = ctypes.c_float(0)
        ev = ctypes.c_uint64(0)
        self._ck(self.L.dvp_bench_cost_kernel(self.h, repeat, ctypes.byref(ms), ctypes.byref(ev)))
        return ms.value, ev.value


def from_scene(scene, params, seed=1234, sampler=0, depths=None, device=0):
    c = Context(scene["width"], scene["height"], len(scene["cameras"]), device=device)
    c.set_images(scene["images"])
    c.set_cameras(scene["cameras"])
    c.set_params(params)
    c.set_seed(seed)
    c.set_sampler(sampler)
    if depths is not None:
        c.set_depths(depths)
    return c
