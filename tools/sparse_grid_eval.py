#!/usr/bin/env python3
"""Re-evaluates DPR_ALGO_AUTO against the measured times of tools/sparse_grid_probe.py (no GPU:
dpr_resolve_algo_ex is host arithmetic).  Usage: python tools/sparse_grid_eval.py probe.txt [--rewrite out.txt]"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd as dpr  # noqa: E402

src = sys.argv[1]
out = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--rewrite" else None
worst = {"fwd": (1.0, ""), "bwd": (1.0, "")}
hi = {"fwd": 0, "bwd": 0}
lines = ["# few points on large grids (fp32, Gaussian 0.4 sigma cloud, as-generated order): ms per call of every algorithm,",
         "# what DPR_ALGO_AUTO resolves to and its regret = t(AUTO) / t(best); tools/sparse_grid_probe.py, re-evaluated",
         "# with the library as built (tools/sparse_grid_eval.py)"]
n = 0
for l in open(src):
    m = re.match(r"P=\s*(\d+) grid=\s*(\S+) B=(\d+)\s+fwd a/t/c\s+(\S+)\s+(\S+)\s+(\S+) auto=(\w+)\s+regret\s+\S+ \| bwd a/t/c\s+(\S+)\s+(\S+)\s+(\S+) auto=(\w+)", l)
    if not m:
        continue
    P, grid, B = int(m.group(1)), tuple(int(x) for x in m.group(2).split("x")), int(m.group(3))
    t = {"fwd": dict(atomic=float(m.group(4)), tiled=float(m.group(5)), chunked=float(m.group(6))),
         "bwd": dict(atomic=float(m.group(8)), tiled=float(m.group(9)), chunked=float(m.group(10)))}
    n += 1
    cells = []
    for op, name in (("raster", "fwd"), ("pullback", "bwd")):
        a = dpr.resolve_algo(op, grid, P, B, 3)
        best = min(v for v in t[name].values() if v == v)
        r = t[name][a] / best
        if r > 1.25:
            hi[name] += 1
        if r > worst[name][0]:
            worst[name] = (r, f"P={P} {'x'.join(map(str, grid))} B={B}")
        cells.append(f"{name} a/t/c {t[name]['atomic']:7.3f} {t[name]['tiled']:7.3f} {t[name]['chunked']:7.3f} auto={a:7s} regret {r:4.2f}")
    lines.append(f"P={P:>8d} grid={'x'.join(map(str, grid)):>12s} B={B}  " + " | ".join(cells))
lines.append(f"# {n} rows; max regret raster {worst['fwd'][0]:.2f} at {worst['fwd'][1]} ({hi['fwd']} above 1.25), "
             f"pullback {worst['bwd'][0]:.2f} at {worst['bwd'][1]} ({hi['bwd']} above 1.25)")
print("\n".join(lines[-8:]))
if out:
    open(out, "w").write("\n".join(lines) + "\n")
