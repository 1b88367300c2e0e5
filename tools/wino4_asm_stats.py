#!/usr/bin/env python3
"""Instruction mix of wino4_conv_f32's K loop per epilogue variant, from hipcc's gfx950 assembly (no GPU needed).

    python tools/wino4_asm_stats.py [-DW4_PERSIST=1 ...]

For every kernel of csrc/wino4_kernels.hip: registers / spills from the .amdhsa metadata, and for the basic-block range that
holds the 72 MFMAs of a chunk (the loop body = from the label the loop's backward branch targets to that branch) the
number of MFMA, VALU (non-MFMA), v_mov, LDS, LDS-DMA, scratch, SALU and waitcnt instructions."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiodenoiser_amd.build import FILE_FLAGS  # noqa: E402

SRC = os.path.join(ROOT, "audiodenoiser_amd", "csrc", "wino4_kernels.hip")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_load") or op.startswith("global_load_lds"):
        return "dma"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    extra = sys.argv[1:]
    keep = os.environ.get("W4_ASM_OUT")
    with tempfile.TemporaryDirectory() as d:
        out = keep or os.path.join(d, "w4.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", "--offload-device-only",
               "-S", SRC, *FILE_FLAGS.get("wino4_kernels.hip", []), "-o", out] + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            sys.exit(r.returncode)
        text = open(out).read()
    # split per function
    funcs = re.split(r"\n\s*\.globl\s+", text)
    for f in funcs[1:]:
        name = f.split("\n", 1)[0].strip()
        if "wino4_conv_f32" not in name:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("void adn::(anonymous namespace)::", "").split("(")[0]
        lines = f.split("\n")
        meta = {}
        for k in ("next_free_vgpr", "next_free_sgpr", "accum_offset", "private_segment_fixed_size"):
            m = re.search(r"\.amdhsa_" + k + r"\s+(\d+)", f)
            meta[k] = int(m.group(1)) if m else -1
        m = re.search(r"; ScratchSize: (\d+)", f)
        # instruction list with labels
        ins = []   # (kind, text) ; labels as ("label", name)
        for ln in lines:
            s = ln.strip()
            if not s or s.startswith(";") or s.startswith("."):
                if re.match(r"^\.LBB\d+_\d+:", s):
                    ins.append(("label", s.split(":")[0]))
                continue
            if re.match(r"^\.?[A-Za-z_0-9$]+:", s):
                ins.append(("label", s.split(":")[0]))
                continue
            op = s.split()[0]
            ins.append((classify(op), s))
        # find loops: backward branches
        label_pos = {t: i for i, (k, t) in enumerate(ins) if k == "label"}
        loops = []
        for i, (k, t) in enumerate(ins):
            if k == "salu" and t.startswith("s_cbranch") or (k == "salu" and t.startswith("s_branch")):
                tgt = t.split()[-1]
                if tgt in label_pos and label_pos[tgt] < i:
                    body = ins[label_pos[tgt]:i + 1]
                    nm = sum(1 for kk, _ in body if kk == "mfma")
                    loops.append((label_pos[tgt], i, nm))
        print(f"{dem}: VGPR {meta['next_free_vgpr']} SGPR {meta['next_free_sgpr']} scratch {meta['private_segment_fixed_size']} total instr {sum(1 for k, _ in ins if k != 'label')}")
        for a, b, nm in loops:
            if nm < 72:
                continue
            body = ins[a:b + 1]
            cnt = {}
            for kk, _ in body:
                if kk != "label":
                    cnt[kk] = cnt.get(kk, 0) + 1
            inner = "  ".join(f"{k} {v}" for k, v in sorted(cnt.items()))
            print(f"    loop [{a}:{b}] mfma {nm}: {inner}")


if __name__ == "__main__":
    main()
