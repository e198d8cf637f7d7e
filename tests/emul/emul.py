"""TEST INFRASTRUCTURE — ctypes wrapper of the host-emulation build of the HIP kernels
(tests/emul/emul.cpp).  Same interface as oracle.oracle.Oracle."""
import ctypes
import os
import subprocess

from oracle import oracle as _o

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = _o.bind_cpu_engine(ctypes.CDLL(os.path.join(_HERE, "libdvp_emul.so")), "emu_")
    return _LIB


class Emul(_o.Oracle):
    def __init__(self, width, height, num_images):
        super().__init__(width, height, num_images, _lib=lib(), _prefix="emu_")
