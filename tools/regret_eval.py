"""Re-evaluate DPR_ALGO_AUTO against a MEASURED regret table (tools/auto_regret.py output) without
a GPU: the times stay, AUTO's choice is recomputed with the library as built now
(dpr_resolve_algo_ex is host arithmetic).  Prints the rows above --show and the max regret.
Usage: python tools/regret_eval.py profiles/r03_auto_regret_measured.txt [--show 1.25] [--rewrite out.txt]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("table")
ap.add_argument("--show", type=float, default=1.25)
ap.add_argument("--rewrite", default=None)
a = ap.parse_args()
section = "random"
worst = {"raster": (1.0, ""), "pullback": (1.0, "")}
out_lines = []
counts = {"raster": [0, 0], "pullback": [0, 0]}
for line in open(a.table):
    line = line.rstrip("\n")
    if line.startswith("## point order:"):
        section = "coherent" if "coherent" in line else "random"
        out_lines.append(line)
        continue
    if line.startswith("#") or "|" not in line:
        if not line.startswith("# max regret"):
            out_lines.append(line)
        continue
    head, f, b = line.split("|")
    kind, P, B, grid = head.split()
    P, B = int(P), int(B)
    grid = tuple(int(x) for x in grid.split("x"))
    kw = dict(coherent_points=True) if section == "coherent" else {}
    cells_out = []
    for op, cells in (("raster", f), ("pullback", b)):
        toks = cells.split()[:3]
        times = {}
        for name, tok in zip(("atomic", "tiled", "chunked"), toks):
            if tok != "-":
                times[name] = float(tok.rstrip("*"))
        auto = dpr_amd.resolve_algo(op, grid, P, B, 3, **kw)
        if auto not in times:  # (3-D chunk lists are not timed on random-order input)
            raise SystemExit(f"AUTO picks {auto} where the table has no time: {line}")
        best = min(times.values())
        regret = times[auto] / best
        counts[op][0] += 1
        counts[op][1] += regret > a.show
        tag = f"{kind} P={P} B={B} {'x'.join(map(str, grid))} {section}"
        if regret > worst[op][0]:
            worst[op] = (regret, tag)
        cells_out.append(" ".join((f"{times[k]:9.3f}" + ("*" if k == auto else " ")) if k in times else "        - "
                                  for k in ("atomic", "tiled", "chunked")) + f" {regret:5.2f}")
    new = f"{kind:10s} {P:9d} {B:3d} {'x'.join(map(str, grid)):12s} | {cells_out[0]} | {cells_out[1]}"
    out_lines.append(new)
    if max(float(cells_out[0].split()[-1]), float(cells_out[1].split()[-1])) > a.show:
        print(section[:3], new)
for op in ("raster", "pullback"):
    s = f"# max regret {op}: {worst[op][0]:.2f} at {worst[op][1]}  ({counts[op][1]} of {counts[op][0]} rows above {a.show})"
    print(s)
    out_lines.append(s)
if a.rewrite:
    with open(a.rewrite, "w") as fh:
        fh.write("\n".join(out_lines) + "\n")
