"""AUTO's choice against the best measured algorithm on the coherent 3-D shapes of round 5's two sweeps
(profiles/r05_owner_batch_sweep*.txt: forward; profiles/r05_batch_pullback_probe.txt: pullback).  Host arithmetic
only (dpr_resolve_algo_ex): runs without a GPU.  Usage: regret_coherent3d.py > profiles/r05_auto_regret_coherent3d.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
rows, worst = [], {}
for path, dt in (("profiles/r05_owner_batch_sweep.txt", "fp32"), ("profiles/r05_owner_batch_sweep_f64.txt", "fp64")):
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        cloud, P, n, B, _, t_own, t_tiled, _ = line.split()
        P, n, B, t = int(P), int(n), int(B), {"chunked": float(t_own), "tiled": float(t_tiled)}
        auto = dpr_amd.resolve_algo("raster", (n, n, n), P, B, 3, coherent_points=True)
        if auto not in t:
            continue  # (atomic: not part of this sweep)
        best = min(t, key=t.get)
        rows.append(("forward", dt, cloud, P, n, B, auto, best, t[auto] / t[best]))
cur = None
for line in open("profiles/r05_batch_pullback_probe.txt"):
    if line.startswith("#"):
        cur = "fp64" if "float64" in line else "fp32"
        continue
    if not line.strip():
        continue
    P, n, B, tc, tt, ta, _ = line.split()
    P, n, B, t = int(P), int(n), int(B), {"chunked": float(tc), "tiled": float(tt), "atomic": float(ta)}
    auto = dpr_amd.resolve_algo("pullback", (n, n, n), P, B, 3, coherent_points=True)
    best = min(t, key=t.get)
    rows.append(("pullback", cur, "gauss", P, n, B, auto, best, t[auto] / t[best]))
print("# regret = t(AUTO's choice) / t(best measured); coherent flag set; forward: owner tiles / chunk lists (`chunked`) vs tiled; pullback: chunked vs tiled vs atomic")
print(f"# {'op':8s} {'type':5s} {'cloud':8s} {'P':>9s} {'grid':>5s} {'B':>3s} {'AUTO':8s} {'best':8s} {'regret':>6s}")
for r in rows:
    print(f"  {r[0]:8s} {r[1]:5s} {r[2]:8s} {r[3]:9d} {r[4]:5d} {r[5]:3d} {r[6]:8s} {r[7]:8s} {r[8]:6.2f}")
    key = (r[0], r[1])
    worst[key] = max(worst.get(key, 1.0), r[8])
print("# max regret:", ", ".join(f"{k[0]} {k[1]} {v:.2f}" for k, v in sorted(worst.items())),
      "; rows above 1.25:", sum(1 for r in rows if r[8] > 1.25), "of", len(rows))
