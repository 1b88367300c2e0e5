#!/usr/bin/env python3
"""Per-kernel digest of the gfx950 ISA hipcc generates for csrc/*.hip (no GPU needed): a refactoring that must not change the
generated code (deleting a closed experiment switch, renaming) is checked by comparing two runs.

    python tools/kernel_isa_digest.py > /tmp/before.json ; ...edit... ; python tools/kernel_isa_digest.py --compare /tmp/before.json
"""
import concurrent.futures
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiodenoiser_amd import build as B  # noqa: E402


def asm_of(src):
    cmd = ([B.hipcc_path(), f"--offload-arch={B.ARCH}"] + B.COMMON_FLAGS + [f"-I{B.INCLUDE}"] + B._extra_flags()
           + B.FILE_FLAGS.get(os.path.basename(src), []) + ["--cuda-device-only", "-S", src, "-o", "-"])
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def digests(text):
    out, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            name, body = m.group(1), []
            continue
        if name and line.startswith(".Lfunc_end"):
            code = "\n".join(body)
            out[name] = [hashlib.sha256(code.encode()).hexdigest()[:16], len(body)]
            name = None
            continue
        if name:
            ins = line.split(";")[0].rstrip()
            if ins.strip() and not ins.strip().startswith("."):          # instructions only (labels kept: they start with .L -> dropped, branch targets stay in the text)
                body.append(re.sub(r"\.LBB\d+_", ".LBB_", ins.strip()))      # (a label carries its function's index in the file: not code)
    return out


def main():
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        texts = list(pool.map(asm_of, B._sources()))
    cur = {}
    for src, t in zip(B._sources(), texts):
        for k, v in digests(t).items():
            cur[os.path.basename(src) + ":" + k] = v
    if "--compare" in sys.argv:
        old = json.load(open(sys.argv[sys.argv.index("--compare") + 1]))
        names = subprocess.run(["c++filt"] + [k.split(":", 1)[1] for k in sorted(set(old) | set(cur))], capture_output=True, text=True).stdout.splitlines()
        diff = 0
        for k, dn in zip(sorted(set(old) | set(cur)), names):
            a, b = old.get(k), cur.get(k)
            if a != b:
                diff += 1
                print(f"DIFF {k.split(':')[0]} {dn[:110]}: {a} -> {b}")
        print(f"{len(cur)} kernels, {diff} differ")
        sys.exit(1 if diff else 0)
    json.dump(cur, sys.stdout, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
