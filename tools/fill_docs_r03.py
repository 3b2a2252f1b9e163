"""Substitutes the @@PLACEHOLDERS of DESIGN.md / README.md / profiles/r03_summary.md with the numbers
of profiles/r03_bench_*.json (the collection run of tools/collect_profiles_r03.sh)."""
import csv, json, re, sys
d = json.load(open("profiles/r03_bench_default.json"))
c4 = json.load(open("profiles/r03_bench_c4_64poses.json"))
c5 = json.load(open("profiles/r03_bench_c5_8poses.json"))
tr = json.load(open("profiles/r03_c3_hbm_traffic.json"))["forward"]["C3/tiled/random"]["hbm_bytes_corrected"]
gather = None
for r in csv.DictReader(open("profiles/r03_c4_kernel_stats.csv")):
    if "k_co_gather" in r["Name"]:
        gather = float(r["AverageNs"]) / 1e6
r = d["roofline"]
val = {
    "C3STEP": f'{d["ms_per_step"]:.3f}', "C3VAL": f'{d["value"] / 1e3:.1f}',
    "C3NOSHARE": f'{d["no_share"]["ms_per_step"]:.3f}', "C3FWD": f'{r["ms"]:.3f}',
    "C3FRAC": f'{100 * r["frac"]:.1f}', "C3TRAFFIC": f'{tr / 1e6:.0f}',
    "C3AMP": f'{tr / r["algorithmic_bytes"]:.1f}', "C3BWD": f'{r["pullback"]["ms"]:.3f}',
    "COHSTEP": f'{d["coherent_input"]["ms_per_step"]:.3f}', "COHFWD": f'{d["coherent_input"]["raster_ms"]:.3f}',
    "CPU": f'{d["cpu_baseline"]["value"]:.1f}', "CORES": str(d["cpu_baseline"]["cores"]),
    "C4STEP": f'{c4["ms_per_step"]:.2f}', "C4VAL": f'{c4["value"] / 1e3:.1f}',
    "C4REF": f'{d["scaling_reference"]["ms_per_step"]:.1f}', "C4GATHER": f"{gather:.2f}",
    "C5STEP": f'{c5["ms_per_step"]:.1f}', "C5NOSHARE": f'{c5["no_share"]["ms_per_step"]:.1f}',
}
for path in sys.argv[1:]:
    s = open(path).read()
    for k, v in val.items():
        s = s.replace("@@" + k, v)
    left = re.findall(r"@@[A-Z0-9]+", s)
    if left:
        print(path, "unfilled:", sorted(set(left)))
    open(path, "w").write(s)
print(val)
