"""dvp-mvs_amd — MI355X-native PatchMatch MVS engine (drop-in for the per-view depth/normal path of
ZhenlongYuan/DVP-MVS).  The product is `libdvp_mvs_hip.so` (HIP kernels for gfx950 behind the C ABI
of include/dvp_mvs.h) plus the C++ host mirror of `class APD` under host/; this Python package is
harness glue: ctypes binding (capi), synthetic scenes (synth), view sharding (sharding).

The directory name contains a hyphen, so import it with
    importlib.import_module("dvp-mvs_amd")
"""
from . import synth, sharding, pipeline, workloads  # noqa: F401


def get_capi():
    """ctypes binding of libdvp_mvs_hip.so (loads — and if needed builds — the HIP library)."""
    import importlib
    return importlib.import_module(__name__ + ".capi")
