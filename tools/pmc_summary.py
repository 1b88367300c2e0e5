#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into profiles/: per-kernel stats, and HBM traffic from separate PMC passes.

    python tools/pmc_summary.py --stats gpurun_out/prof/stats --fetch gpurun_out/prof/fetch \
        --write gpurun_out/prof/write --tag r02_bench --cmd "python3 bench.py --steps 5 ..." [--traffic-json]

HBM bytes per dispatch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled
(/opt/skills/guides/MI355X_MICROARCH.md, HBM).  FETCH and WRITE come from two separate rocprofv3 --pmc runs of the
same command (they do not fit one pass).

With --traffic-json the per-family means are written to profiles/pmc_traffic.json together with the source digest of
the libadn.so that was profiled (audiodenoiser_amd/_lib/libadn.sha256); bench.py reports `roofline.traffic` only
when that digest equals the library it is running.
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel families bench.py asks for (key -> predicate on the short kernel name)
FAMILIES = {
    "wino4_conv_f32": lambda n: n.startswith("wino4_conv_f32"),
    "wino_conv_dma_f32": lambda n: n.startswith("wino_conv_dma_f32"),
    "conv_mfma_f32": lambda n: n.startswith("conv_mfma<float") and ", 9, " in n,
    # fp16 3x3 layers: conv_dma<_Float16, 32, 64, ...> (32x32x16 MFMA) and conv16_f16<...> (16x16x32 MFMA, round 4)
    "conv_mfma_f16": lambda n: (n.startswith(("conv_mfma<_Float16", "conv_dma<_Float16")) and ", 9, " in n) or n.startswith("conv16_f16"),
    "conv16_f16": lambda n: n.startswith("conv16_f16"),
    # transposed convolutions: <T, TH = 8, BN = 128, WM = 2, WN = 2, TAPS = 1, ...> (split-bf16 form: conv_dma<float, 8, 128, 2, 2, 1, 2, 2, 3, 1>)
    "convt_f32": lambda n: n.startswith(("conv_mfma<float", "conv_dma<float")) and ", 8, 128, 2, 2, 1, " in n,
    "convt_f16": lambda n: (n.startswith(("conv_mfma<_Float16", "conv_dma<_Float16")) and ", 8, 128, 2, 2, 1, " in n) or n.startswith("convt16_f16"),
    "stft_wave_kernel": lambda n: n.startswith("stft_wave_kernel") and not n.rstrip().endswith("true>"),
    "stft_wave_kernel_fit": lambda n: n.startswith("stft_wave_kernel") and n.rstrip().endswith("true>"),
    "stft_fit_kernel": lambda n: n.startswith("stft_fit_kernel"),
    "conv_first_kernel": lambda n: n.startswith(("conv_first_kernel<float", "conv_first_c8_kernel")),
    "conv_out_kernel": lambda n: n.startswith("conv_out_kernel<float"),
    "conv_first_kernel_f16": lambda n: n.startswith("conv_first_kernel<_Float16"),
    "conv_out_kernel_f16": lambda n: n.startswith("conv_out_kernel<_Float16"),
}


def demangle_f16(name: str) -> str:
    """rocprofv3 leaves names with _Float16 template arguments mangled (DF16_); rebuild `kernel<_Float16, 16, ...>`."""
    m = re.match(r"_ZN3adn12_GLOBAL__N_1(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    rest = name[m.end():]
    kern, rest = rest[:n], rest[n:]
    if not rest.startswith("I"):
        return kern
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("DF16_", i):
            args.append("_Float16")
            i += 5
        elif rest[i] == "f":
            args.append("float")
            i += 1
        elif rest[i] == "L":                       # Li<digits>E
            j = rest.index("E", i)
            args.append(rest[i + 2:j])
            i = j + 1
        else:
            break
    return f"{kern}<{', '.join(args)}>"


def short(name: str) -> str:
    name = demangle_f16(name)
    name = name.replace("void ", "").replace("adn::(anonymous namespace)::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name[:90]


def ours(n: str) -> bool:
    return n.startswith(("conv_", "conv16", "convt16", "stft", "per_clip", "nhwc", "quantize", "wino", "loss_", "gl_", "istft", "dot_finish"))


def read_counter(d, counter):
    per = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"))
    ap.add_argument("--cmd", default="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline")
    ap.add_argument("--traffic-json", action="store_true", help="(re)write profiles/pmc_traffic.json")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    lines = []

    if args.stats:
        for path in glob.glob(os.path.join(args.stats, "**", "*_kernel_stats.csv"), recursive=True):
            with open(path) as fh:
                rows = list(csv.DictReader(fh))
            lines.append(f"# rocprofv3 --kernel-trace --stats -- {args.cmd}")
            lines.append(f"{'kernel':92s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>9s} {'pct':>6s}")
            fam = defaultdict(lambda: [0, 0.0])
            for r in rows:
                n = short(r["Name"])
                if not ours(n):
                    continue
                lines.append(f"{n:92s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:10.3f} "
                             f"{float(r['AverageNs']) / 1e6:9.4f} {float(r['Percentage']):6.2f}")
                for key, pred in FAMILIES.items():
                    if pred(n):
                        fam[key][0] += int(r["Calls"])
                        fam[key][1] += float(r["TotalDurationNs"]) / 1e6
            for key, (calls, tot) in fam.items():
                lines.append(f"family {key}: {calls} launches, {tot:.3f} ms total, {tot / calls:.4f} ms average per launch")
    traffic = {}
    if args.fetch and args.write:
        fetch = read_counter(args.fetch, "FETCH_SIZE")
        write = read_counter(args.write, "WRITE_SIZE")
        lines.append("")
        lines.append(f"# HBM traffic per dispatch from two separate --pmc passes of `{args.cmd}`: "
                     "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes")
        lines.append(f"{'kernel':92s} {'disp':>5s} {'fetch_MiB(x2)':>14s} {'write_MiB':>10s} {'total_MiB':>10s}")
        agg = defaultdict(lambda: [0, 0.0])
        for n in sorted(fetch):
            if not ours(n):
                continue
            f = fetch[n]
            w = write.get(n, [0.0])
            fb = 2 * sum(f) / len(f) * 1024
            wb = sum(w) / len(w) * 1024
            lines.append(f"{n:92s} {len(f):5d} {fb / 2**20:14.2f} {wb / 2**20:10.2f} {(fb + wb) / 2**20:10.2f}")
            for key, pred in FAMILIES.items():
                if pred(n):
                    agg[key][0] += len(f)
                    agg[key][1] += (fb + wb) * len(f)
        for key, (cnt, tot) in agg.items():
            traffic[key] = {"bytes_per_launch": round(tot / cnt), "dispatches": cnt, "cmd": args.cmd}
            lines.append(f"family {key}: mean HBM bytes per launch {round(tot / cnt)} over {cnt} dispatches")
    with open(os.path.join(args.out, f"{args.tag}_rocprof_summary.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    if traffic and args.traffic_json:
        digest = ""
        try:
            with open(os.path.join(ROOT, "audiodenoiser_amd", "_lib", "libadn.sha256")) as fh:
                digest = fh.readline().strip()        # first line = code digest (audiodenoiser_amd/build.py)
        except OSError:
            pass
        rec = {"lib_digest": digest,
               "note": "mean over each kernel family's dispatches: (2*FETCH_SIZE + WRITE_SIZE)*1024 from two separate "
                       "rocprofv3 --pmc passes; the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md; "
                       "lib_digest = source digest of the libadn.so that was profiled",
               "kernels": traffic}
        with open(os.path.join(args.out, "pmc_traffic.json"), "w") as fh:
            json.dump(rec, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
