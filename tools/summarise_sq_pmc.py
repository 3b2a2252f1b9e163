#!/usr/bin/env python3
"""Per-kernel SQ counter summary of a `tools/r04_sq_pmc.sh` pass.

usage: summarise_sq_pmc.py <counter_collection.csv> <kernel_stats.txt> <label>
Durations come from the (un-profiled) kernel trace summary of the same bench command; the SIMD-time
normalisation assumes 1024 SIMDs at 2.0 GHz (quad-cycle counters)."""
import collections
import csv
import re
import sys

CLK, SIMDS = 2.0e9, 1024
COUNTERS = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
            "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS")


def main(csv_path, stats_path, label):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(csv_path)):
        k = r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")
        if k.startswith("dpr::"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    for line in open(stats_path):
        m = re.search(r"void (dpr::\w+).*calls=\s*(\d+)\s+avg_us=\s*([\d.]+)", line)
        if m:
            d = dur[m.group(1)]
            d[0] += int(m.group(2)) * float(m.group(3))
            d[1] += int(m.group(2))
    out = [f"{label}",
           f"{'kernel':24s} {'us':>6s} {'waves/SIMD':>10s} {'VALU busy':>10s} {'LDS busy':>9s} "
           f"{'wave time waiting':>18s} {'... on LDS':>10s}"]
    for k, v in sorted(agg.items()):
        a = {c: sum(x) / len(x) for c, x in v.items()}
        if dur[k][1] == 0:
            continue
        us = dur[k][0] / dur[k][1]
        st = us * 1e-6 * CLK / 4 * SIMDS
        wc = a["SQ_WAVE_CYCLES"]
        out.append(f"{k:24s} {us:6.1f} {wc / st:10.1f} {100 * a['SQ_ACTIVE_INST_VALU'] / st:9.0f}% "
                   f"{100 * a['SQ_ACTIVE_INST_LDS'] / st:8.0f}% "
                   f"{100 * (a['SQ_WAIT_ANY'] + a['SQ_WAIT_INST_ANY']) / wc:17.0f}% "
                   f"{100 * a['SQ_WAIT_INST_LDS'] / wc:9.0f}%")
    out += ["", "Raw averages per launch:"]
    for k, v in sorted(agg.items()):
        a = {c: sum(x) / len(x) for c, x in v.items()}
        wc = a.get("SQ_WAVE_CYCLES", 1)
        out.append(k)
        for c in COUNTERS:
            if c in a:
                out.append(f"    {c:22s} {a[c]:12.4g}   {100 * a[c] / wc:5.1f} % of wave cycles")
    print("\n".join(out))


if __name__ == "__main__":
    main(*sys.argv[1:4])
