"""Turns gpurun_out/final6/* (tools/collect_profiles_r06.sh) into the committed profiles/r06_* files."""
import collections, csv, glob, json, os, shutil, subprocess
O = "gpurun_out/final6"
for a, b in [("bench_default.json", "r06_bench_default.json"), ("bench_under_rocprof.json", "r06_bench_under_rocprof.json"),
             ("bench_c2.json", "r06_bench_c2.json"), ("bench_c4_64poses.json", "r06_bench_c4_64poses.json"),
             ("bench_c5_8poses.json", "r06_bench_c5_8poses.json"),
             ("bench_c3_coherent_auto.json", "r06_bench_c3_coherent_auto.json"),
             ("bench_c3_coherent_chunked.json", "r06_bench_c3_coherent_chunked.json"),
             ("bench_c5_8poses_coherent.json", "r06_bench_c5_8poses_coherent.json"), ("bench_c5_full.json", "r06_bench_c5_full.json"), ("ablations.jsonl", "r06_ablations.jsonl"), ("microbench_valu.txt", "r06_microbench_valu.txt"), ("microbench_write.txt", "r06_microbench_write_bandwidth.txt"), ("readme_timings.txt", "r06_reference_readme_timings.txt"), ("other_configs.txt", "r06_other_configs.txt")]:
    if os.path.exists(f"{O}/{a}") and os.path.getsize(f"{O}/{a}") > 0:
        shutil.copy(f"{O}/{a}", f"profiles/{b}")

def kernel_stats(sub, title, dst):
    ks_path = max(glob.glob(f"{O}/{sub}/*/*_kernel_stats.csv"), key=os.path.getmtime)
    shutil.copy(ks_path, f"profiles/{dst}.csv")
    ks = list(csv.DictReader(open(ks_path)))
    lines = [title, ""]
    for r in ks[:18]:
        lines.append(f'{r["Name"][:92]:92s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:>9.1f} pct={float(r["Percentage"]):6.2f}')
    open(f"profiles/{dst}.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

CMD = "rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-scaling-reference"
kernel_stats("stats", CMD + "\n(C3: 10M points 0.4*N(0,I) random order -> 256^3 fp32, tiled algorithm, KEEP / REUSE pair, MI355X)", "r06_c3_kernel_stats")
kernel_stats("stats_coh_auto", CMD + " --order hilbert --coherent\n(C3, Hilbert-sorted cloud + DPR_FLAG_COHERENT_POINTS, DPR_ALGO_AUTO: tiled forward with local binning, direct 3-D pullback)", "r06_c3_coherent_auto_kernel_stats")
kernel_stats("stats_coh_chunked", CMD + " --order hilbert --coherent --algo chunked\n(C3, Hilbert-sorted cloud, DPR_ALGO_CHUNKED: owner-computes forward over the box hierarchy, direct pullback)", "r06_c3_coherent_chunked_kernel_stats")
kernel_stats("stats_c4", "rocprofv3 --kernel-trace --stats -- python bench.py --config C4 --poses 64 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-scaling-reference\n(C4 at the share of one GPU of 8: 10M points -> 512^2 fp32, 64 poses, chunk-owner algorithm)", "r06_c4_kernel_stats")
kernel_stats("stats_c5", "rocprofv3 --kernel-trace --stats -- python bench.py --config C5 --poses 8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-scaling-reference\n(C5 at the share of one GPU of 8: 50M points -> 512^3 fp64, 8 poses, tiled algorithm)", "r06_c5_kernel_stats")

def pmc(pattern, name):
    rows = list(csv.DictReader(open(max(glob.glob(pattern), key=os.path.getmtime))))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}

# What each kernel MUST read per launch at C3 (bytes), from the data structures it walks: decides, per
# kernel, whether FETCH_SIZE is one of the half-reported ones (gfx950 reports half the bytes of wide
# coalesced reads, MI355X_MICROARCH.md "HBM / rocprofv3"): a raw value below 0.75 x this floor is doubled.
P, G = 10_000_000, 256 ** 3
MUST_READ = {
    "dpr::k_count": 12 * P, "dpr::k_scatter_wc": 12 * P, "dpr::k_scatter": 12 * P, "dpr::k_bin_local": 12 * P,
    "dpr::k_tile_splat": 16 * P, "dpr::k_tile_splat_runs": 16 * P,
    "dpr::k_tile_gather": 16 * P + 4 * G, "dpr::k_tile_gather_runs": 16 * P + 4 * G,
    "dpr::k_unpermute": 4 * P + 16 * P, "dpr::k_halo_gather": 0,
    "dpr::k_own_boxes": 12 * P,            # the point array
    "dpr::k_own_splat": 12 * P,            # every point at least once (visited 1.6 times; repeats hit the caches)
    "dpr::k_own_pullback": 12 * P + 4 * G,  # points + every ds_dout cell (grid sum)
}
out = {"note": "Per-launch HBM traffic of the C3 forward / pullback kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-scaling-reference [--order hilbert --coherent [--algo chunked]]`). Counter unit is KiB. Correction per MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads. Decided PER KERNEL from the bytes the kernel must read (`must_read_bytes`): a raw FETCH_SIZE below 0.75 x that floor is doubled (`fetch_doubled: true`), everything else is left as reported. WRITE_SIZE is exact. `hbm_bytes_upper_bound` doubles every kernel's FETCH_SIZE (the value if all reads were of the half-reported kind): kernels that mix 16-byte and 4-byte loads (k_own_splat, k_tile_splat_runs: raw value just below the floor) lie between the two.",
       "kernels": {}, "forward": {}, "pullback": {}}
fwd_k = ["dpr::k_count", "dpr::k_colscan", "dpr::k_tilescan", "dpr::k_scatter", "dpr::k_scatter_wc", "dpr::k_tile_splat",
         "dpr::k_halo_gather", "dpr::k_bin_local", "dpr::k_runscan", "dpr::k_place_desc", "dpr::k_tile_splat_runs",
         "__amd_rocclr_fillBufferAligned", "dpr::k_own_boxes", "dpr::k_own_boxes2", "dpr::k_own_plan", "dpr::k_own_splat",
         "dpr::k_own_combine"]
bwd_k = ["dpr::k_tile_gather", "dpr::k_tile_gather_runs", "dpr::k_unpermute", "dpr::k_pose_reduce", "dpr::k_own_pullback",
         "dpr::k_own_reduce"]
for mode, key_f, key_b in (("random", "C3/tiled/random", "C3/tiled/random"),
                           ("coh_auto", "C3/tiled/hilbert+coherent", "C3/chunked/hilbert+coherent"),
                           ("coh_chunked", "C3/chunked/hilbert+coherent", None)):
    if not glob.glob(f"{O}/fetch_{mode}/*/*_counter_collection.csv"):
        continue
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
    for grp, key in (("forward", key_f), ("pullback", key_b)):
        if key is None:
            continue
        out[grp][key] = {"fetch_bytes_raw": round(tot[grp][0]), "fetch_bytes_corrected": round(tot[grp][1]),
                         "write_bytes": round(tot[grp][2]), "hbm_bytes_corrected": round(tot[grp][1] + tot[grp][2]),
                         "hbm_bytes_upper_bound": round(2 * tot[grp][0] + tot[grp][2]),
                         "collected_with": mode}
# C5 share (50 M points -> 512^3 fp64, 8 poses): HBM bytes per CALL (all launches of one forward / one pullback
# over the 8 poses).  Calls in the profiled run = launches of k_bin_local (one per forward call: all 8 poses in
# one launch; the pullback of the KEEP / REUSE pair does not bin).
if glob.glob(f"{O}/fetch_c5/*/*_counter_collection.csv"):
    def whole_run(pattern, name):
        rows = list(csv.DictReader(open(max(glob.glob(pattern), key=os.path.getmtime))))
        tot = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in rows:
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")
                tot[k] += float(r["Counter_Value"]) * 1024; n[k] += 1
        return tot, n
    fa, fn = whole_run(f"{O}/fetch_c5/*/*_counter_collection.csv", "FETCH_SIZE")
    wa, wn = whole_run(f"{O}/write_c5/*/*_counter_collection.csv", "WRITE_SIZE")
    ncall = max(fn.get("dpr::k_bin_local", 0), 1)
    P5, G5, B5 = 50_000_000, 512 ** 3, 8
    # what a kernel must read per CALL (bytes): decides the doubling of its FETCH_SIZE as above
    floors = {"dpr::k_cell_count": 24 * P5, "dpr::k_cell_scatter": 24 * P5, "dpr::k_bin_local": 24 * P5,
              "dpr::k_tile_splat_runs": 32 * P5 * B5, "dpr::k_tile_gather_runs": (32 * P5 + 8 * G5) * B5,
              "dpr::k_unpermute_batch": 36 * P5 * B5, "dpr::k_unpermute": 36 * P5 * B5}
    bwd5 = ("dpr::k_tile_gather", "dpr::k_tile_gather_runs", "dpr::k_unpermute", "dpr::k_unpermute_batch", "dpr::k_pose_reduce",
            "dpr::k_unsort", "dpr::k_own_pullback", "dpr::k_own_pullback_batch", "dpr::k_own_reduce", "dpr::k_own_reduce_batch")
    tot = {"forward": [0.0, 0.0, 0.0], "pullback": [0.0, 0.0, 0.0]}
    for k in sorted(set(fa) | set(wa)):
        if not (k.startswith("dpr::") or k.startswith("__amd_rocclr_fill")):
            continue
        fr, wr = fa.get(k, 0.0) / ncall, wa.get(k, 0.0) / ncall
        floor = floors.get(k, 0)
        doubled = floor > 0 and fr < 0.75 * floor
        fc = 2 * fr if doubled else fr
        out["kernels"][f"{k}/c5_share_per_call"] = {"FETCH_SIZE_bytes_raw": round(fr), "must_read_bytes": floor, "fetch_doubled": doubled,
                                                    "fetch_bytes_corrected": round(fc), "WRITE_SIZE_bytes": round(wr),
                                                    "launches_per_call": round(fn.get(k, wn.get(k, 0)) / ncall, 2)}
        grp = "pullback" if k in bwd5 else "forward"
        tot[grp][0] += fr; tot[grp][1] += fc; tot[grp][2] += wr
    for grp in ("forward", "pullback"):
        out[grp]["C5/tiled/random/B8"] = {"fetch_bytes_raw": round(tot[grp][0]), "fetch_bytes_corrected": round(tot[grp][1]),
                                          "write_bytes": round(tot[grp][2]), "hbm_bytes_corrected": round(tot[grp][1] + tot[grp][2]),
                                          "hbm_bytes_upper_bound": round(2 * tot[grp][0] + tot[grp][2]),
                                          "collected_with": f"bench.py --config C5 --poses 8 ({ncall} calls in the profiled run)"}
    # the whole 64-pose job runs its forward in eight batches of 8 poses (one cell sort + binning per batch) and its
    # pullback per 64-pose launch: the forward's traffic is 8 x the share's; recorded as such (not a PMC pass of its own)
    e = dict(out["forward"]["C5/tiled/random/B8"])
    for k in ("fetch_bytes_raw", "fetch_bytes_corrected", "write_bytes", "hbm_bytes_corrected", "hbm_bytes_upper_bound"):
        e[k] = 8 * e[k]
    e["collected_with"] = "8 x the C5/tiled/random/B8 entry (the forward walks the 64 poses in batches of 8); no PMC pass of its own"
    out["forward"]["C5/tiled/random/B64"] = e
json.dump(out, open("profiles/r06_hbm_traffic.json", "w"), indent=1)
for grp in ("forward", "pullback"):
    for k, v in out[grp].items():
        print(grp, k, {a: (round(b / 1e6, 1) if isinstance(b, (int, float)) else b) for a, b in v.items()})
for k, v in out["kernels"].items():
    print(f"{k:44s} raw {v['FETCH_SIZE_bytes_raw']/1e6:8.1f} MB  floor {v['must_read_bytes']/1e6:7.1f}  doubled {str(v['fetch_doubled']):5s} write {v['WRITE_SIZE_bytes']/1e6:8.1f}")

def sq(sub_busy, sub_insts, stats_txt, label, dst, mode="w"):
    busy = max(glob.glob(f"{O}/{sub_busy}/*/*_counter_collection.csv"), key=os.path.getmtime)
    txt = subprocess.run(["python3", "tools/summarise_sq_pmc.py", busy, stats_txt, label], capture_output=True, text=True).stdout
    rows = list(csv.DictReader(open(max(glob.glob(f"{O}/{sub_insts}/*/*_counter_collection.csv"), key=os.path.getmtime))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")
        if k.startswith("dpr::"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["", "Executed wave-instructions per launch (same command, second pass):",
             f"{'kernel':24s} {'VALU':>10s} {'SALU':>10s} {'LDS':>10s} {'SMEM':>10s} {'VMEM rd':>10s}"]
    for k, v in sorted(agg.items()):
        a = {c: sum(x) / len(x) for c, x in v.items()}
        lines.append(f"{k:24s} {a.get('SQ_INSTS_VALU', 0):10.4g} {a.get('SQ_INSTS_SALU', 0):10.4g} {a.get('SQ_INSTS_LDS', 0):10.4g} "
                     f"{a.get('SQ_INSTS_SMEM', 0):10.4g} {a.get('SQ_INSTS_VMEM_RD', 0):10.4g}")
    open(f"profiles/{dst}", mode).write(txt + "\n".join(lines) + "\n\n")

HEAD = ("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-scaling-reference [...]\n"
        "One pass of 8 SQ slots per command; averages per launch.  Derived columns: simd_time = kernel duration (kernel trace of the same command) x 1024 SIMDs x 2.0 GHz / 4\n"
        "(the clock is an assumption, +-10 %); waves/SIMD = WAVE_CYCLES / simd_time; VALU busy = ACTIVE_INST_VALU / simd_time; LDS busy likewise.\n\n")
if glob.glob(f"{O}/sq_random/*/*_counter_collection.csv"):
    open("profiles/r06_c3_sq_counters.txt", "w").write(HEAD)
    sq("sq_random", "sqi_random", "profiles/r06_c3_kernel_stats.txt", "C3 step, random order", "r06_c3_sq_counters.txt", "a")
    sq("sq_coh_chunked", "sqi_coh_chunked", "profiles/r06_c3_coherent_chunked_kernel_stats.txt",
       "C3 step, --order hilbert --coherent --algo chunked (k_hilbert_keys / k_gather_points: the bench's untimed pre-sort)", "r06_c3_sq_counters.txt", "a")
d = json.load(open("profiles/r06_bench_default.json"))
print("bench:", d["value"], d["ms_per_step"], d.get("ms_per_step_cold"), d["roofline"]["frac"], d["roofline"]["traffic"], d["coherent_input"]["value"], d["cpu_baseline"]["value"], d["no_share"])
