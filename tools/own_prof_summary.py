#!/usr/bin/env python3
"""Summary of a tools/own_prof.sh collection: per dpr kernel the average duration and SQ counters."""
import collections, csv, glob, os, sys

def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None

def short(k):
    return k.split("<")[0].split("(")[0].replace("void ", "")

O = sys.argv[1]
dur = {}
f = find(os.path.join(O, "kt"), "*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k.startswith("dpr::"):
            d = dur.setdefault(k, [0.0, 0])
            d[0] += float(r["TotalDurationNs"]); d[1] += int(r["Calls"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq1", "sq2"):
    f = find(os.path.join(O, sub), "*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith("dpr::"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
CLK, SIMDS = 2.4e9, 1024
for k in sorted(dur, key=lambda k: -dur[k][0]):
    us = dur[k][0] / dur[k][1] / 1e3
    a = {c: sum(v) / len(v) for c, v in agg.get(k, {}).items()}
    line = f"{k:28s} calls {dur[k][1]:4d} avg {us:8.1f} us"
    if "SQ_WAVE_CYCLES" in a:
        st = us * 1e-6 * CLK / 4 * SIMDS  # SIMD quad-cycles available
        wc = a["SQ_WAVE_CYCLES"]
        line += (f" | waves/SIMD {wc / st:4.1f} VALU busy {100 * a.get('SQ_ACTIVE_INST_VALU', 0) / st:3.0f}%"
                 f" LDS busy {100 * a.get('SQ_ACTIVE_INST_LDS', 0) / st:3.0f}%"
                 f" waiting {100 * (a.get('SQ_WAIT_ANY', 0) + a.get('SQ_WAIT_INST_ANY', 0)) / wc:3.0f}% (LDS {100 * a.get('SQ_WAIT_INST_LDS', 0) / wc:3.0f}%)")
    if "SQ_INSTS_VALU" in a:
        line += (f" | insts VALU {a['SQ_INSTS_VALU']:.3g} SALU {a.get('SQ_INSTS_SALU', 0):.3g} LDS {a.get('SQ_INSTS_LDS', 0):.3g}"
                 f" VMEM_RD {a.get('SQ_INSTS_VMEM_RD', 0):.3g} bank-conflict cyc {a.get('SQ_LDS_BANK_CONFLICT', 0):.3g}"
                 f" lds idx active {a.get('SQ_LDS_IDX_ACTIVE', 0):.3g}")
    print(line)
