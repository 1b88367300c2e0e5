#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into profiles/: per-kernel stats, and HBM traffic from separate PMC passes.

    python tools/pmc_summary.py --stats gpurun_out/prof/stats --fetch gpurun_out/prof/fetch \
        --write gpurun_out/prof/write --tag r01

HBM bytes per dispatch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled
(/opt/skills/guides/MI355X_MICROARCH.md §HBM, cdna_hip_programming.md §7).  FETCH and WRITE come from two
separate rocprofv3 --pmc runs of the same command (they do not fit one pass).
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "").replace("adn::(anonymous namespace)::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name[:90]


def ours(n: str) -> bool:
    return n.startswith(("conv_", "stft", "per_clip", "nhwc", "quantize", "wino", "loss_", "gl_", "istft"))


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
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--cmd", default="python bench.py --steps 5 --warmup 2 --no-cpu-baseline")
    ap.add_argument("--dominant", default="wino_conv_dma_f32", help="kernel-name prefix of the dominant kernel")
    ap.add_argument("--dominant-filter", default="", help="substring selecting its instantiations")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    lines = []

    def dominant(n):
        return n.startswith(args.dominant) and args.dominant_filter in n

    if args.stats:
        for path in glob.glob(os.path.join(args.stats, "**", "*_kernel_stats.csv"), recursive=True):
            with open(path) as fh:
                rows = list(csv.DictReader(fh))
            lines.append(f"# rocprofv3 --kernel-trace --stats -- {args.cmd}")
            lines.append(f"{'kernel':92s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>9s} {'pct':>6s}")
            tot = [0, 0.0]
            for r in rows:
                n = short(r["Name"])
                if not ours(n):
                    continue
                lines.append(f"{n:92s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:10.3f} "
                             f"{float(r['AverageNs']) / 1e6:9.4f} {float(r['Percentage']):6.2f}")
                if dominant(n):
                    tot[0] += int(r["Calls"])
                    tot[1] += float(r["TotalDurationNs"]) / 1e6
            if tot[0]:
                lines.append(f"dominant kernel {args.dominant} (all selected instantiations): {tot[0]} launches, "
                             f"{tot[1]:.3f} ms total, {tot[1] / tot[0]:.4f} ms average per launch")
    traffic = {}
    if args.fetch and args.write:
        fetch = read_counter(args.fetch, "FETCH_SIZE")
        write = read_counter(args.write, "WRITE_SIZE")
        lines.append("")
        lines.append("# HBM traffic per dispatch from two separate --pmc passes: (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes")
        lines.append(f"{'kernel':92s} {'disp':>5s} {'fetch_MiB(x2)':>14s} {'write_MiB':>10s} {'total_MiB':>10s}")
        agg = [0, 0.0]
        for n in sorted(fetch):
            if not ours(n):
                continue
            f = fetch[n]
            w = write.get(n, [0.0])
            fb = 2 * sum(f) / len(f) * 1024
            wb = sum(w) / len(w) * 1024
            lines.append(f"{n:92s} {len(f):5d} {fb / 2**20:14.2f} {wb / 2**20:10.2f} {(fb + wb) / 2**20:10.2f}")
            if dominant(n):
                agg[0] += len(f)
                agg[1] += (fb + wb) * len(f)
        if agg[0]:
            traffic["dominant_bytes_per_launch"] = round(agg[1] / agg[0])
            traffic["dominant_kernel"] = args.dominant
            traffic["note"] = ("mean over the dominant kernel's dispatches of bench.py (batch 64): (2*FETCH_SIZE + "
                               "WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes; the x2 is the gfx950 FETCH_SIZE "
                               "correction of MI355X_MICROARCH.md")
            lines.append(f"{args.dominant} mean HBM bytes per launch: {traffic['dominant_bytes_per_launch']}")
    with open(os.path.join(args.out, f"{args.tag}_rocprof_summary.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    if traffic:
        with open(os.path.join(args.out, "pmc_traffic.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
