"""ORACLE — TEST INFRASTRUCTURE ONLY (ctypes wrapper around libdvp_oracle.so).  PARITY UNPINNED.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STAGES = dict(gen_edge_inform=0, find_nearest_strong=1, gen_neighbours=2, neighbour_update=3,
              random_init=4, strong_update=5, ransac_fit=6, weak_update=7, get_depth_normal=8,
              filter_strong=9, depth_to_weak=10, local_refine=11)
BUFFERS = dict(planes=(0, np.float32, 4), costs=(1, np.float32, 1), selected_views=(2, np.uint32, 1),
               view_weight=(3, np.uint8, 32), weak_info=(4, np.uint8, 1), weak_reliable=(5, np.uint8, 1),
               weak_nearest_strong=(6, np.int16, 2), neighbours_map=(7, np.int32, 1),
               neighbours=(8, np.int16, 2), fit_planes=(9, np.float32, 4), candidate=(10, np.int16, 2),
               edge=(11, np.uint8, 1), edge_neigh=(12, np.int16, 2), label=(13, np.int32, 1),
               label_boundary=(14, np.int16, 2), complex=(15, np.float32, 1), radius=(16, np.int32, 1))


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def bind_cpu_engine(L, pre):
    """Declare the C signatures of a CPU engine library whose entry points start with `pre`
    (`ora_` here; the host-emulation build of the HIP kernels in tests/emul uses `emu_`)."""
    g = lambda n: getattr(L, pre + n)
    g("create").restype = ctypes.c_void_p
    g("create").argtypes = [ctypes.c_int] * 3
    g("destroy").argtypes = [ctypes.c_void_p]
    for name in ("set_image", "set_depth"):
        g(name).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    g("set_cameras").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    g("set_params").argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    g("set_seed").argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    g("set_sampler").argtypes = [ctypes.c_void_p, ctypes.c_int]
    g("count_evals").argtypes = [ctypes.c_void_p, ctypes.c_int]
    g("get_evals").restype = ctypes.c_longlong
    g("get_evals").argtypes = [ctypes.c_void_p]
    g("upload_state").argtypes = [ctypes.c_void_p] * 7
    g("buffer_bytes").restype = ctypes.c_longlong
    g("buffer_bytes").argtypes = [ctypes.c_void_p, ctypes.c_int]
    g("get_buffer").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    g("set_buffer").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    g("weak_count").argtypes = [ctypes.c_void_p]
    g("run_stage").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    g("expf").restype = ctypes.c_float
    g("expf").argtypes = [ctypes.c_float]
    g("eval_cost_vectors").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return L


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libdvp_oracle.so")
        if not os.path.exists(so):
            build()
        L = bind_cpu_engine(ctypes.CDLL(so), "ora_")
        L.ora_run_patchmatch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ora_rand_u32.restype = ctypes.c_uint32
        L.ora_rand_u32.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.ora_tex_linear.restype = ctypes.c_float
        L.ora_tex_linear.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int]
        L.ora_homography.argtypes = [ctypes.c_void_p] * 4
        for name in ("ora_ncc_old", "ora_ncc_new", "ora_geom_cost"):
            f = getattr(L, name)
            f.restype = ctypes.c_float
            f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class _Pre:
    """`obj.L.ora_x(...)` -> `<prefix>x(...)` of the bound library."""

    def __init__(self, L, pre):
        self._L, self._pre = L, pre

    def __getattr__(self, name):
        assert name.startswith("ora_")
        return getattr(self._L, self._pre + name[4:])


class Oracle:
    """One reference view: mirrors the C-ABI context of include/dvp_mvs.h on the CPU."""

    def __init__(self, width, height, num_images, _lib=None, _prefix="ora_"):
        self.L = _Pre(_lib if _lib is not None else lib(), _prefix)
        self.W, self.H, self.NI = width, height, num_images
        self.h = ctypes.c_void_p(self.L.ora_create(width, height, num_images))

    def close(self):
        if self.h:
            self.L.ora_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_images(self, images):
        for i in range(self.NI):
            a = np.ascontiguousarray(images[i], np.float32)
            assert a.shape == (self.H, self.W)
            self.L.ora_set_image(self.h, i, _p(a))

    def set_depths(self, depths):
        for i in range(self.NI):
            a = np.ascontiguousarray(depths[i], np.float32)
            self.L.ora_set_depth(self.h, i, _p(a))

    def set_cameras(self, cams):
        a = np.ascontiguousarray(cams)
        assert a.dtype.itemsize == 112
        self.L.ora_set_cameras(self.h, _p(a), len(a))

    def set_params(self, params):
        a = np.ascontiguousarray(params).reshape(1)
        assert a.dtype.itemsize == 76
        self.L.ora_set_params(self.h, _p(a))
        self.params = a

    def set_seed(self, seed):
        self.L.ora_set_seed(self.h, seed)

    def set_sampler(self, s):
        self.L.ora_set_sampler(self.h, s)

    def set_numerics(self, n):
        """0 = the numerics contract (default); 1 = literal per-operator evaluation of the reference's
        NCC expressions (the source's per-x-offset partial sums, one division per tap, tex2D(x+0.5) with the
        add/subtract pair rounded).  Oracle only: a cross-check of how far the contract is from it."""
        self.L.ora_set_numerics.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.L.ora_set_numerics(self.h, n)

    def upload_state(self, planes=None, views=None, weak=None, edge=None, label=None, radius=None):
        c = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        args = [c(planes, np.float32), c(views, np.uint32), c(weak, np.uint8), c(edge, np.uint8), c(label, np.int32), c(radius, np.int32)]
        self.L.ora_upload_state(self.h, *[_p(a) for a in args])

    def get(self, name):
        bid, dt, k = BUFFERS[name]
        nbytes = self.L.ora_buffer_bytes(self.h, bid)
        out = np.empty(nbytes // np.dtype(dt).itemsize, dt)
        self.L.ora_get_buffer(self.h, bid, _p(out))
        return out.reshape(-1, k) if k > 1 else out

    def set(self, name, arr):
        bid, dt, k = BUFFERS[name]
        a = np.ascontiguousarray(arr, dt)
        assert a.nbytes == self.L.ora_buffer_bytes(self.h, bid), (name, a.nbytes)
        self.L.ora_set_buffer(self.h, bid, _p(a))

    def weak_count(self):
        return self.L.ora_weak_count(self.h)

    def run_stage(self, name, it=0, colour=0):
        r = self.L.ora_run_stage(self.h, STAGES[name], it, colour)
        assert r == 0

    def run_stage_pixels(self, name, it, colour, px):
        """the launch's per-pixel body on the listed pixels only (px: [n, 2] of (x, y)); returns how many of them the launch visits"""
        a = np.ascontiguousarray(px, np.int32).reshape(-1, 2)
        self.L.ora_run_stage_pixels.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        r = self.L.ora_run_stage_pixels(self.h, STAGES[name], it, colour, _p(a), len(a))
        assert r >= 0
        return r

    def run_patchmatch(self):
        if self.L._pre != "ora_":
            self.L._L.emu_run_patchmatch.argtypes = [ctypes.c_void_p]
            self.L._L.emu_run_patchmatch(self.h)
            return 0.0
        t = ctypes.c_double(0)
        self.L.ora_run_patchmatch(self.h, ctypes.byref(t))
        return t.value

    def count_evals(self, on=True):
        self.L.ora_count_evals(self.h, int(on))

    def evals(self):
        return self.L.ora_get_evals(self.h)

    def ncc_old(self, x, y, src_idx, plane):
        pl = np.ascontiguousarray(plane, np.float32)
        return self.L.ora_ncc_old(self.h, x, y, src_idx, _p(pl))

    def ncc_new(self, x, y, src_idx, plane):
        pl = np.ascontiguousarray(plane, np.float32)
        return self.L.ora_ncc_new(self.h, x, y, src_idx, _p(pl))

    def geom_cost(self, x, y, src_idx, plane):
        pl = np.ascontiguousarray(plane, np.float32)
        return self.L.ora_geom_cost(self.h, x, y, src_idx, _p(pl))

    def eval_cost_vectors(self, px, planes):
        px = np.ascontiguousarray(px, np.int32)
        planes = np.ascontiguousarray(planes, np.float32)
        n = len(px)
        out = np.empty((n, self.NI - 1), np.float32)
        self.L.ora_eval_cost_vectors(self.h, _p(px), _p(planes), n, _p(out))
        return out


def from_scene(scene, params, seed=1234, sampler=0, depths=None, cls=None):
    o = (cls or Oracle)(scene["width"], scene["height"], len(scene["cameras"]))
    o.set_images(scene["images"])
    o.set_cameras(scene["cameras"])
    o.set_params(params)
    o.set_seed(seed)
    o.set_sampler(sampler)
    if depths is not None:
        o.set_depths(depths)
    return o
