"""Build libadn.so (hand-written HIP for gfx950) in-tree: ``python -m audiodenoiser_amd.build``.

One hipcc invocation over ``csrc/*.hip``; output ``audiodenoiser_amd/_lib/libadn.so`` (git-ignored, shipped to
the GPU box with the working tree).  A content hash of the sources is stored next to the library so that a
copied tree with fresh mtimes does not trigger a rebuild.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libadn.so")
STAMP = os.path.join(LIBDIR, "libadn.sha256")
ARCH = "gfx950"


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest() -> str:
    h = hashlib.sha256()
    for p in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(INCLUDE, "adn.h")]:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(ARCH.encode())
    return h.hexdigest()


def hipcc_path():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def up_to_date() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == _digest()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libadn.so (ROCm toolchain required)")
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", f"-I{INCLUDE}", "-o", LIB] + _sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(_digest() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
