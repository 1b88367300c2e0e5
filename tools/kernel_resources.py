#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one csrc/*.hip translation unit, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (no GPU needed: cross-compiles for gfx950).

    python tools/kernel_resources.py audiodenoiser_amd/csrc/conv_kernels.hip [-D...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiodenoiser_amd.build import FILE_FLAGS  # noqa: E402


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", f"-I{ROOT}/include",
               "-Rpass-analysis=kernel-resource-usage", src, *FILE_FLAGS.get(os.path.basename(src), []), "-o", os.path.join(d, "x.o")] + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        sys.exit(r.returncode)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]+\])?: +(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = [r_["name"] for r_ in rows]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>5s} {'scratch':>7s} {'occ':>4s} {'SGPR':>5s}")
    for r_, n in zip(rows, dem):
        n = n.replace("void adn::(anonymous namespace)::", "").split("(")[0]
        print(f"{n[:78]:78s} {r_.get('VGPRs', '?'):>5s} {r_.get('AGPRs', '?'):>5s} {r_.get('VGPRs Spill', '?'):>5s} "
              f"{r_.get('ScratchSize', '?'):>7s} {r_.get('Occupancy', '?'):>4s} {r_.get('SGPRs', '?'):>5s}")


if __name__ == "__main__":
    main()
