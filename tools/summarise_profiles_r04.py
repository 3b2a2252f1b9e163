"""Turns gpurun_out/final4/* (tools/collect_profiles_r04.sh) into the committed profiles/r04_* files."""
import collections, csv, glob, json, os, shutil
O = "gpurun_out/final4"
for a, b in [("bench_default.json", "r04_bench_default.json"), ("bench_under_rocprof.json", "r04_bench_under_rocprof.json"),
             ("bench_c2.json", "r04_bench_c2.json"), ("bench_c4_64poses.json", "r04_bench_c4_64poses.json"),
             ("bench_c4_64poses_coherent.json", "r04_bench_c4_64poses_coherent.json"),
             ("bench_c5_8poses.json", "r04_bench_c5_8poses.json"), ("other_configs.txt", "r04_other_configs.txt")
             ]:
    if os.path.exists(f"{O}/{a}"):
        shutil.copy(f"{O}/{a}", f"profiles/{b}")

def kernel_stats(sub, title, dst):
    ks_path = max(glob.glob(f"{O}/{sub}/*/*_kernel_stats.csv"), key=os.path.getmtime)  # the latest run
    shutil.copy(ks_path, f"profiles/{dst}.csv")
    ks = list(csv.DictReader(open(ks_path)))
    lines = [title, ""]
    for r in ks[:18]:
        lines.append(f'{r["Name"][:92]:92s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:>9.1f} pct={float(r["Percentage"]):6.2f}')
    open(f"profiles/{dst}.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

kernel_stats("stats", "rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-scaling-reference\n"
             "(C3: 10M points 0.4*N(0,I) random order -> 256^3 fp32, tiled algorithm, MI355X)", "r04_c3_kernel_stats")
kernel_stats("stats_coherent", "rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --order hilbert --coherent\n"
             "(C3, Hilbert-sorted cloud + DPR_FLAG_COHERENT_POINTS: local binning)", "r04_c3_coherent_kernel_stats")
kernel_stats("stats_c4", "rocprofv3 --kernel-trace --stats -- python bench.py --config C4 --poses 64 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-scaling-reference\n"
             "(C4 at the share of one GPU of 8: 10M points -> 512^2 fp32, 64 poses, chunk-owner algorithm)", "r04_c4_kernel_stats")
kernel_stats("stats_c5", "rocprofv3 --kernel-trace --stats -- python bench.py --config C5 --poses 8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-scaling-reference\n"
             "(C5 at the share of one GPU of 8: 50M points -> 512^3 fp64, 8 poses, tiled algorithm: cell sort + local binning of all poses, the pullback reuses the binning of every pose)", "r04_c5_kernel_stats")

def pmc(pattern, name):
    rows = list(csv.DictReader(open(max(glob.glob(pattern), key=os.path.getmtime))))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}

# What each kernel MUST read per launch at C3 (10 M points, 256^3 fp32, one pose; bytes), from the
# data structures it walks -- the yardstick that decides, per kernel, whether FETCH_SIZE is one of
# the half-reported ones (gfx950 reports half the bytes of wide coalesced reads,
# MI355X_MICROARCH.md "HBM / rocprofv3"): a raw value below 0.75 x this floor cannot be right and is
# doubled; everything else is left as reported.
P, G = 10_000_000, 256 ** 3
MUST_READ = {
    "dpr::k_count": 12 * P,                       # the point array
    "dpr::k_scatter_wc": 12 * P,                  # the point array (+ 2 MB of prefixes)
    "dpr::k_scatter": 12 * P,
    "dpr::k_bin_local": 12 * P,
    "dpr::k_tile_splat": 16 * P,                  # 16-byte records (every 32-byte sector of them)
    "dpr::k_tile_splat_runs": 16 * P,
    "dpr::k_tile_gather": 16 * P + 4 * G,         # records + the ds_dout tiles
    "dpr::k_tile_gather_runs": 16 * P + 4 * G,
    "dpr::k_unpermute": 4 * P + 16 * P,           # slot map + one gradient record per point
    "dpr::k_halo_gather": 0,                      # (latency-bound; no floor used)
}
out = {"note": "Per-launch HBM traffic of the C3 forward / pullback kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary [--order hilbert --coherent]`). Counter unit is KiB. Correction per MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads. Decided PER KERNEL from the bytes the kernel must read (`must_read_bytes`, from the data structures it walks): a raw FETCH_SIZE below 0.75 x that floor is doubled (`fetch_doubled: true`), everything else is left as reported. WRITE_SIZE is exact.",
       "kernels": {}, "forward": {}, "pullback": {}}
fwd_k = ["dpr::k_count", "dpr::k_colscan", "dpr::k_tilescan", "dpr::k_scatter", "dpr::k_scatter_wc", "dpr::k_tile_splat",
         "dpr::k_halo_gather", "dpr::k_bin_local", "dpr::k_runscan", "dpr::k_place_desc", "dpr::k_tile_splat_runs",
         "__amd_rocclr_fillBufferAligned"]
bwd_k = ["dpr::k_tile_gather", "dpr::k_tile_gather_runs", "dpr::k_unpermute", "dpr::k_pose_reduce"]
for mode, key in (("random", "C3/tiled/random"), ("coherent", "C3/tiled/hilbert+coherent")):
    f = pmc(f"{O}/fetch_{mode}/*/*_counter_collection.csv", "FETCH_SIZE")
    w = pmc(f"{O}/write_{mode}/*/*_counter_collection.csv", "WRITE_SIZE")
    tot = {"forward": [0, 0, 0], "pullback": [0, 0, 0]}
    for k in fwd_k + bwd_k:
        if k not in f and k not in w:
            continue
        fr = f.get(k, 0) * 1024; wr = w.get(k, 0) * 1024
        floor = MUST_READ.get(k, 0)
        doubled = floor > 0 and fr < 0.75 * floor
        fc = fr * 2 if doubled else fr
        out["kernels"][f"{k}/{mode}"] = {"FETCH_SIZE_bytes_raw": round(fr), "must_read_bytes": floor, "fetch_doubled": doubled,
                                         "fetch_bytes_corrected": round(fc), "WRITE_SIZE_bytes": round(wr)}
        grp = "forward" if k in fwd_k else "pullback"
        tot[grp][0] += fr; tot[grp][1] += fc; tot[grp][2] += wr
    for grp in tot:
        out[grp][key] = {"fetch_bytes_raw": round(tot[grp][0]), "fetch_bytes_corrected": round(tot[grp][1]),
                         "write_bytes": round(tot[grp][2]), "hbm_bytes_corrected": round(tot[grp][1] + tot[grp][2])}
json.dump(out, open("profiles/r04_c3_hbm_traffic.json", "w"), indent=1)
for grp in ("forward", "pullback"):
    for k, v in out[grp].items():
        print(grp, k, {a: round(b / 1e6, 1) for a, b in v.items()})
for k, v in out["kernels"].items():
    print(f"{k:44s} raw {v['FETCH_SIZE_bytes_raw']/1e6:8.1f} MB  floor {v['must_read_bytes']/1e6:7.1f}  doubled {str(v['fetch_doubled']):5s} write {v['WRITE_SIZE_bytes']/1e6:8.1f}")
d = json.load(open("profiles/r04_bench_default.json"))
print("bench:", d["value"], d["ms_per_step"], d.get("ms_per_step_cold"), d["roofline"]["frac"], d["roofline"]["traffic"], d["coherent_input"]["value"], d["cpu_baseline"]["value"], d["no_share"])
print(json.dumps(d["roofline"]["stages"]))


# SQ counter passes (tools/r04_sq_pmc.sh, tools/r04_sq_insts.sh): busy shares per kernel + executed instructions
import subprocess
def sq(sub_busy, sub_insts, stats_txt, label, dst, mode="w"):
    busy = max(glob.glob(f"{O}/{sub_busy}/sq/*/*_counter_collection.csv"), key=os.path.getmtime)
    txt = subprocess.run(["python3", "tools/summarise_sq_pmc.py", busy, stats_txt, label], capture_output=True, text=True).stdout
    rows = list(csv.DictReader(open(max(glob.glob(f"{O}/{sub_insts}/sq/*/*_counter_collection.csv"), key=os.path.getmtime))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")
        if k.startswith("dpr::"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["", "Executed wave-instructions per launch (tools/r04_sq_insts.sh, same command):",
             f"{'kernel':24s} {'VALU':>10s} {'SALU':>10s} {'LDS':>10s} {'SMEM':>10s} {'VMEM rd':>10s}"]
    for k, v in sorted(agg.items()):
        a = {c: sum(x) / len(x) for c, x in v.items()}
        lines.append(f"{k:24s} {a.get('SQ_INSTS_VALU', 0):10.4g} {a.get('SQ_INSTS_SALU', 0):10.4g} {a.get('SQ_INSTS_LDS', 0):10.4g} "
                     f"{a.get('SQ_INSTS_SMEM', 0):10.4g} {a.get('SQ_INSTS_VMEM_RD', 0):10.4g}")
    open(f"profiles/{dst}", mode).write(txt + "\n".join(lines) + "\n\n")

HEAD = ("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-scaling-reference [...]\n"
        "One pass of 8 SQ slots per command (tools/r04_sq_pmc.sh); averages per launch.  SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* are quad-cycles summed over all\n"
        "waves (MI355X_MICROARCH.md), SQ_BUSY_CYCLES cycles summed over the 32 shader engines.  Derived columns: simd_time = kernel duration (the kernel trace of the\n"
        "same command, launch-weighted over template instances) x 1024 SIMDs x 2.0 GHz / 4 (the clock is an assumption, +-10 %); waves/SIMD = WAVE_CYCLES / simd_time;\n"
        "VALU busy = ACTIVE_INST_VALU / simd_time (a SIMD issues one vector instruction per quad-cycle); LDS busy likewise.\n\n")
if glob.glob(f"{O}/sq_c3/sq/*/*_counter_collection.csv"):
    open("profiles/r04_c3_sq_counters.txt", "w").write(HEAD)
    sq("sq_c3", "sqi_c3", "profiles/r04_c3_kernel_stats.txt", "C3 step, random order", "r04_c3_sq_counters.txt", "a")
    sq("sq_c3coh", "sqi_c3coh", "profiles/r04_c3_coherent_kernel_stats.txt", "C3 step, --order hilbert --coherent (k_hilbert_keys / k_gather_points: the bench's untimed pre-sort)", "r04_c3_sq_counters.txt", "a")
    open("profiles/r04_c4_sq_counters.txt", "w").write(HEAD)
    sq("sq_c4", "sqi_c4", "profiles/r04_c4_kernel_stats.txt", "C4 share: --config C4 --poses 64", "r04_c4_sq_counters.txt", "a")
