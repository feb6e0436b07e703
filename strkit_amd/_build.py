"""Builds libstrkit_amd.so (HIP, gfx950 only) in-tree with hipcc.

The shared library is the product's compute path; there is no CPU fallback.  hipcc cross-compiles
gfx950 code objects without a GPU present, so this runs in the build container and on the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libstrkit_amd.so")
SOURCES = ["strk_api.hip"]
# every header and include fragment under csrc/ (picked up by name, so that a new one can never be left out of the staleness hash) + the C ABI
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + [os.path.join("..", "..", "include", "strkit_amd.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wall", "-Wextra", "-Wno-unused-parameter", "-lz", "-lpthread"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm); strkit_amd has no CPU fallback")


STAMP_PATH = LIB_PATH + ".srchash"


def source_hash() -> str:
    """Content hash of everything the library is built from (file times do not survive a copy of the tree)."""
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for rel in SOURCES + HEADERS:
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read())
    return h.hexdigest()


def stale() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the library if it is missing or older than its sources; returns its path."""
    if not force and not stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = [_hipcc(), *HIPCC_FLAGS, "-o", tmp, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB_PATH)
    with open(STAMP_PATH + ".tmp.%d" % os.getpid(), "w") as f:
        f.write(source_hash() + "\n")
    os.replace(STAMP_PATH + ".tmp.%d" % os.getpid(), STAMP_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
