#!/usr/bin/env python3
"""sha256 over the kernel sources (dvp-mvs_amd/csrc/*.hip, *.hpp, *.inc, Makefile; name + NUL + content, sorted by name).
The library embeds it at build time (dvp_build_id()), the PMC table records it, bench.py compares the three."""
import glob
import hashlib
import os
import sys


def csrc_sha256(csrc_dir=None):
    d = csrc_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dvp-mvs_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(d, "*.inc")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256(sys.argv[1] if len(sys.argv) > 1 else None))
