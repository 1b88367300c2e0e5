"""Build libadn.so (hand-written HIP for gfx950) in-tree: ``python -m audiodenoiser_amd.build``.

Every ``csrc/*.hip`` is compiled to an object (in parallel, per-file flags in ``FILE_FLAGS``) and the objects are
linked into ``audiodenoiser_amd/_lib/libadn.so`` (git-ignored, shipped to the GPU box with the working tree).  A
content hash of the sources and flags is stored next to the library so that a copied tree with fresh mtimes does
not trigger a rebuild (stamp = two lines: the code digest that identifies the build in bench.py / profiles, and the
raw-bytes digest that decides about rebuilding).
"""
from __future__ import annotations

import concurrent.futures
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libadn.so")
STAMP = os.path.join(LIBDIR, "libadn.sha256")
LOCK = os.path.join(LIBDIR, ".build.lock")
ARCH = "gfx950"
COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-fvisibility-inlines-hidden"]
# only the C ABI of include/adn.h leaves the library (ADN_API there + this version script: `adn_*` global, the rest local)
VERSION_SCRIPT = os.path.join(CSRC, "libadn.map")
LINK_FLAGS = ["-fPIC", "-shared", f"-Wl,--version-script={VERSION_SCRIPT}"]
# wino4_kernels.hip: the SLP vectoriser pairs unrelated scalars of the 6x6 transform into v_pk_* operations and pays
# for it with ~140 v_mov per K-chunk (the transform is scalar by design: one channel per lane and pass)
FILE_FLAGS = {"wino4_kernels.hip": ["-fno-slp-vectorize"]}


def _extra_flags():
    """Extra ``-D`` switches of a local A/B build (``ADN_BUILD_DEFINES``); empty for the production library, whose sources carry
    no timing-experiment code (closed experiments are kept as patches under ``profiles/experiments/``)."""
    return os.environ.get("ADN_BUILD_DEFINES", "").split()


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _strip_comments(text: str) -> str:
    """C / C++ source with comments removed and whitespace normalised, so that the digest below identifies the CODE of a
    build: editing a comment or re-wrapping a line does not orphan profiles/pmc_traffic.json.  String and character literals
    are kept byte for byte (spacing inside them is code), and a preprocessor directive keeps the newline that ends it
    (moving a token across the end of a ``#define`` changes the program)."""
    out, lits = [], []
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":                                   # literal: set aside verbatim up to the closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            lits.append(text[i:j + 1])
            out.append(f"\x00{len(lits) - 1}\x00")
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            while j > 0 and text[j - 1] == "\\":             # a line comment continued by a backslash
                j = text.find("\n", j + 1)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" ")
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    parts, pending = [], ""
    for line in "".join(out).split("\n"):
        line = pending + line
        if line.rstrip().endswith("\\"):                   # continued line (multi-line #define)
            pending = line.rstrip()[:-1] + " "
            continue
        pending = ""
        norm = " ".join(line.split())
        if norm:
            parts.append(norm + ("\n" if norm.startswith("#") else " "))
    code = "".join(parts)
    for k, lit in enumerate(lits):
        code = code.replace(f"\x00{k}\x00", lit)
    return code


def _digest_files():
    return _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(INCLUDE, "adn.h"), VERSION_SCRIPT]


def _flags_tag() -> bytes:
    return (ARCH + "|" + " ".join(_extra_flags()) + "|" + repr(sorted(FILE_FLAGS.items())) + "|" + " ".join(COMMON_FLAGS)).encode()


def _digest() -> str:
    """Digest of the CODE (comments and layout stripped) + flags: the identity bench.py and profiles/pmc_traffic.json use."""
    h = hashlib.sha256()
    for p in _digest_files():
        h.update(os.path.basename(p).encode())
        with open(p, "r", encoding="utf-8") as f:
            h.update(_strip_comments(f.read()).encode())
    h.update(_flags_tag())
    return h.hexdigest()


def _raw_digest() -> str:
    """Digest of the source BYTES + flags: what decides whether the in-tree library is rebuilt (any edit rebuilds)."""
    h = hashlib.sha256()
    for p in _digest_files():
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(_flags_tag())
    return h.hexdigest()


def code_digest_of_built_library() -> str:
    """First line of the stamp written next to libadn.so: the code digest of the build that produced it ("" if none)."""
    try:
        with open(STAMP) as f:
            return f.readline().strip()
    except OSError:
        return ""


def hipcc_path():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def up_to_date() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        lines = f.read().split()
    return len(lines) == 2 and lines[1] == _raw_digest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Build (or reuse) the library.  Safe to call from several processes at once (one rank per GPU all import the
    package): an exclusive file lock serialises the builders, the first one compiles into a temporary file and
    renames it into place, the others find the library up to date when they get the lock."""
    if not force and up_to_date():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    with open(LOCK, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and up_to_date():
                return LIB
            hipcc = hipcc_path()
            if hipcc is None:
                raise RuntimeError("hipcc not found: cannot build libadn.so (ROCm toolchain required)")
            tmp = f"{LIB}.{os.getpid()}.tmp"
            with tempfile.TemporaryDirectory(prefix="adn_build_") as objdir:
                def compile_one(src):
                    obj = os.path.join(objdir, os.path.basename(src) + ".o")
                    cmd = ([hipcc, f"--offload-arch={ARCH}"] + COMMON_FLAGS + [f"-I{INCLUDE}"] + _extra_flags()
                           + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])
                    if verbose:
                        print(" ".join(cmd), file=sys.stderr)
                    subprocess.run(cmd, check=True)
                    return obj
                workers = max(1, min(len(_sources()), (os.cpu_count() or 2) // 2))
                with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
                    objs = list(pool.map(compile_one, _sources()))
                cmd = [hipcc, f"--offload-arch={ARCH}"] + LINK_FLAGS + ["-o", tmp] + objs
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                try:
                    subprocess.run(cmd, check=True)
                    os.replace(tmp, LIB)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
            with open(STAMP + ".tmp", "w") as f:
                f.write(_digest() + "\n" + _raw_digest() + "\n")
            os.replace(STAMP + ".tmp", STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


def build_variant(name: str, defines) -> str:
    """Cross-compile a build-time VARIANT of the library (extra -D switches of a work-in-progress A/B) to
    ``_lib/variants/libadn_<name>.so`` without touching the production library.  Variants are compiled in the build container
    and shipped to the GPU box with the tree; a process picks one with ``ADN_LIBADN_PATH`` (see ``_lib.load``).  Tools only:
    nothing in the product path sets that variable."""
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found")
    vdir = os.path.join(LIBDIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    dst = os.path.join(vdir, f"libadn_{name}.so")
    flags = list(defines)
    with tempfile.TemporaryDirectory(prefix="adn_variant_") as objdir:
        def compile_one(src):
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            subprocess.run([hipcc, f"--offload-arch={ARCH}"] + COMMON_FLAGS + [f"-I{INCLUDE}"] + flags
                           + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj], check=True)
            return obj
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, (os.cpu_count() or 2) // 2)) as pool:
            objs = list(pool.map(compile_one, _sources()))
        subprocess.run([hipcc, f"--offload-arch={ARCH}"] + LINK_FLAGS + ["-o", dst] + objs, check=True)
    return dst


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":          # python -m audiodenoiser_amd.build --variant NAME [-D...]
        print(build_variant(sys.argv[2], [a for a in sys.argv[3:] if a != "--production"]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
