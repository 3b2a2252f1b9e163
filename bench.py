#!/usr/bin/env python3
"""Headline benchmark: M points/s for forward + pullback, 10 M 3-D points -> 256^3 fp32 grid
(BASELINE.json metric; config C3 = `configs[2]`), one pose per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A "step" is one raster! + one raster_pullback! of this rank's pose over the full point
cloud, inputs already resident in HBM.  At N > 1 the global problem is a batch of N poses
sharded one per rank (weak scaling) and each step ends with the all-reduce(sum) of the
fused [ds_dpoints | ds_dpoint_weight] buffer -- the only exchange the batched pullback has
(/root/reference/src/raster_pullback.jl:141,146).  By default that all-reduce runs on RCCL's
stream while the next step's kernels run (double-buffered; all of them complete inside the
timed region); `--no-overlap-exchange` serialises it.

Besides the contract's JSON line fields this prints `roofline` (dominant kernel, HIP-event
timed on the stream the kernels run on) and, on rank 0 at N = 1, `cpu_baseline` (the CPU
oracle's threaded port of the reference algorithm on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (P, n_in, grid, dtype)
    "C2": (1_000_000, 3, (128, 128, 128), "f32"),
    "C3": (10_000_000, 3, (256, 256, 256), "f32"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # measured float4 copy ceiling (same guide)


def synth_inputs(cfg, rank, device, order, dist_kind="gauss"):
    """points 0.4*N(0,I) seed 0 (test/data.jl:22,27); rotation uniform on
    SO(3), translation 0.1*N(0,I), seed 1 (+rank); ds_dout N(0,1) seed 2 (+rank)."""
    import torch

    from tests import data as D

    P, n_in, grid, dt = CONFIGS[cfg]
    npdt = np.float32 if dt == "f32" else np.float64
    rng = np.random.default_rng(0)
    pts = (0.4 * rng.standard_normal(size=(P, n_in), dtype=np.float32)).astype(npdt)
    if dist_kind == "uniform":  # diagnostic only (balanced tiles); not the headline workload
        pts = (1.1 * rng.random(size=(P, n_in), dtype=np.float32) - 0.55).astype(npdt)
    if dist_kind == "tight":  # diagnostic only: a compact cluster, few heavily loaded tiles
        pts = (0.1 * rng.standard_normal(size=(P, n_in), dtype=np.float32)).astype(npdt)
    if order == "morton":
        pts = pts[morton_order(pts)]
    prng = np.random.default_rng(1 + rank)
    R = D.random_rotations(prng, 1, n_in)[:, : len(grid), :].astype(npdt)
    t = (0.1 * prng.normal(size=(1, len(grid)))).astype(npdt)
    tdt = torch.float32 if dt == "f32" else torch.float64
    gen = torch.Generator(device=device)
    gen.manual_seed(2 + rank)
    g = torch.randn((1,) + tuple(reversed(grid)), device=device, dtype=tdt, generator=gen)
    g = g.permute(*reversed(range(g.ndim)))  # [i1, i2, i3, b] view, reference memory order
    to = lambda a: torch.as_tensor(a, device=device)
    return dict(points=to(pts), R=to(R), t=to(t), ds_dout=g, np_points=pts, np_R=R, np_t=t)


def morton_order(pts, bits=10):
    """Pose-independent spatial pre-sort of the model-frame points (Morton / Z-order)."""
    q = np.clip(((pts * 0.5 + 0.5) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(pts), dtype=np.int64)
    for b in range(bits):
        for d in range(pts.shape[1]):
            code |= ((q[:, d] >> b) & 1) << (pts.shape[1] * b + d)
    return np.argsort(code, kind="stable")


def algorithmic_bytes(cfg, with_point_weight=False):
    """A_fwd = s[P(N_in+w) + B G]; A_bwd = s[P(N_in+w) + B G + P N_in + P]."""
    P, n_in, grid, dt = CONFIGS[cfg]
    s = 4 if dt == "f32" else 8
    G = int(np.prod(grid))
    w = 1 if with_point_weight else 0
    a_fwd = s * (P * (n_in + w) + G)
    a_bwd = s * (P * (n_in + w) + G + P * n_in + P)
    return a_fwd, a_bwd


def load_traffic_profile(args, algo_f):
    """HBM bytes per forward call from the PMC counters (FETCH_SIZE / WRITE_SIZE, collected in
    separate rocprofv3 passes of this same command and corrected as MI355X_MICROARCH.md
    prescribes); measured offline, committed under profiles/ -- bench.py cannot profile itself."""
    path = os.path.join(ROOT, "profiles", "r01_c3_hbm_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        prof = json.load(f)
    key = f"{args.config}/{algo_f}/{args.order}"
    ent = prof.get("forward", {}).get(key)
    if not ent:
        return None
    return {"bytes": ent["hbm_bytes_corrected"], "source": f"profiles/r01_c3_hbm_traffic.json[{key}]"}


def cpu_baseline(cfg, inp, budget_s=20.0):
    """Threaded CPU port of the reference algorithm (oracle/, kind "port") on a bounded
    sample: the first `n` points of the same cloud into the same grid, n chosen so the
    fwd+bwd pair takes roughly `budget_s` seconds."""
    from oracle import oracle

    P, n_in, grid, dt = CONFIGS[cfg]
    npdt = np.float32 if dt == "f32" else np.float64
    threads = oracle.max_threads()
    g = np.asfortranarray(inp["ds_dout"].cpu().numpy())

    def run(n):
        t0 = time.perf_counter()
        oracle.raster(grid, inp["np_points"][:n], inp["np_R"], inp["np_t"], dtype=npdt, threaded=True)
        t1 = time.perf_counter()
        oracle.raster_pullback(g, inp["np_points"][:n], inp["np_R"], inp["np_t"], dtype=npdt,
                               threaded=True)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    n = min(P, 200_000)
    f, b = run(n)  # calibration (includes the fixed grid fill / grid sum cost)
    rate = n / max(f + b, 1e-9)
    n2 = int(min(P, max(n, rate * budget_s)))
    f, b = run(n2)
    return {
        "value": round(n2 / (f + b) / 1e6, 4), "unit": "M points/s", "cores": threads,
        "kind": "port",
        "sample": f"first {n2} of {P} points, same grid/pose; fwd {f:.2f}s on {threads} threads "
                  f"(atomic scatter), bwd {b:.2f}s on 1 thread (the reference's batched pullback "
                  f"is serial within a pose, src/raster_pullback.jl:39,115-139)",
        "fwd_M_points_s": round(n2 / f / 1e6, 4), "bwd_M_points_s": round(n2 / b / 1e6, 4),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS))
    ap.add_argument("--algo", default="auto", choices=["auto", "atomic", "tiled", "chunked"])
    ap.add_argument("--order", default="random", choices=["random", "morton"],
                    help="point order in memory: as generated, or pre-sorted (pose-independent)")
    ap.add_argument("--dist", default="gauss", choices=["gauss", "uniform", "tight"])
    ap.add_argument("--no-overlap-exchange", action="store_true",
                    help="N > 1: finish each step's all-reduce before the next step starts")
    ap.add_argument("--no-share-binning", action="store_true",
                    help="make the pullback redo the binning instead of reusing the forward's")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurement on Morton-sorted points")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    import torch

    import dpr_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # DPR_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs
    # than ranks (ranks share devices, the all-reduce goes through host memory); never the
    # measured configuration.
    backend = os.environ.get("DPR_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    P, n_in, grid, dt = CONFIGS[args.config]
    s_elem = 4 if dt == "f32" else 8
    inp = synth_inputs(args.config, rank, device, args.order, args.dist)
    tdt = inp["points"].dtype
    out = dpr_amd.empty_grid(grid, 1, tdt, device)
    fused = torch.empty(P * (n_in + 1), dtype=tdt, device=device)
    d_pts = fused[: P * n_in].view(P, n_in)
    d_pw = fused[P * n_in:]
    ws_bytes = max(dpr_amd.workspace_bytes("raster", grid, P, 1, n_in, tdt, args.algo),
                   dpr_amd.workspace_bytes("pullback", grid, P, 1, n_in, tdt, args.algo))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=device)

    algo_f = args.algo if args.algo != "auto" else dpr_amd.resolve_algo("raster", grid, P, 1, n_in)
    algo_b = args.algo if args.algo != "auto" else dpr_amd.resolve_algo("pullback", grid, P, 1, n_in)
    # The pullback reuses the tile binning its forward call built in the same step (what an
    # rrule caches between `raster` and its pullback closure); nothing is carried across steps.
    share = (not args.no_share_binning) and algo_f == algo_b and algo_f in ("tiled", "chunked")

    def fwd(keep=share):
        dpr_amd.raster_(out, inp["points"], inp["R"], inp["t"], algo=algo_f, workspace=ws,
                        keep_binning=keep)

    def bwd(reuse=share):
        dpr_amd.raster_pullback_(inp["ds_dout"], inp["points"], inp["R"], inp["t"],
                                 ds_dpoints=d_pts, ds_dpoint_weight=d_pw, algo=algo_b,
                                 workspace=ws, reuse_binning=reuse)

    # Exchange of the point gradients (N > 1).  Default: the all-reduce of step k runs on RCCL's
    # own stream while step k+1 computes into the other half of a double buffer (a rank that
    # works through its poses one after the other overlaps exactly like this); every
    # all-reduce is finished before the timed region's closing barrier.  --no-overlap-exchange
    # waits for it inside the step.
    overlap = world > 1 and backend == "nccl" and not args.no_overlap_exchange
    fused_alt = torch.empty_like(fused) if overlap else None
    pending = [None, None]
    step_no = [0]

    def exchange(buf):
        if world > 1:
            if backend == "nccl":
                return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=overlap)
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)
        return None

    def step():
        k = step_no[0] & 1 if overlap else 0
        step_no[0] += 1
        buf = fused_alt if k else fused
        if pending[k] is not None:
            pending[k].wait()  # the launch stream waits until this buffer's all-reduce is done
            pending[k] = None
        fwd()
        dpr_amd.raster_pullback_(inp["ds_dout"], inp["points"], inp["R"], inp["t"],
                                 ds_dpoints=buf[: P * n_in].view(P, n_in),
                                 ds_dpoint_weight=buf[P * n_in:], algo=algo_b, workspace=ws,
                                 reuse_binning=share)
        pending[k] = exchange(buf)

    def drain():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def barrier():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu",
                          dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    ms_per_step = elapsed / args.steps * 1e3
    value = world * P / (elapsed / args.steps) / 1e6  # M points/s, whole job

    # ---- per-pass device time with HIP events on the launch stream (torch's current stream);
    # forward and pullback are timed inside fwd+bwd pairs (the pullback consumes -- and, when it
    # reuses the binning, destroys -- what its forward left in the workspace)
    reps = max(5, min(args.steps, 20))
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(reps)]
    for e0, e1, e2 in evs:
        e0.record()
        fwd()
        e1.record()
        bwd()
        e2.record()
    torch.cuda.synchronize()
    ms_fwd = float(np.mean([e0.elapsed_time(e1) for e0, e1, _ in evs]))
    ms_bwd = float(np.mean([e1.elapsed_time(e2) for _, e1, e2 in evs]))
    a_fwd, a_bwd = algorithmic_bytes(args.config)
    st_f = dpr_amd.stage_times(fwd, "raster", algo_f, reps)
    st_b = dpr_amd.stage_times(bwd, "pullback", algo_b, reps, prepare=fwd)
    stages = {"raster": {"algo": algo_f, **{k: round(v, 4) for k, v in st_f.items()}},
              "pullback": {"algo": algo_b, **{k: round(v, 4) for k, v in st_b.items()}}}
    # dominant kernel of the forward call, with the bytes that kernel itself has to move
    kernel_bytes = {  # tiled pipeline, fp32/fp64 record = 4 values
        "count": s_elem * P * n_in, "scatter": s_elem * P * n_in + 4 * s_elem * P,
        "tile_splat": 4 * s_elem * P + s_elem * int(np.prod(grid)), "splat": a_fwd,
        "chunk_splat": a_fwd}
    dom = max((k for k in st_f if k != "total"), key=lambda k: st_f[k])
    dominant = {"stage": dom, "ms": round(st_f[dom], 4)}
    if dom in kernel_bytes:
        dominant["bytes_moved_by_this_kernel"] = kernel_bytes[dom]
        dominant["GBps"] = round(kernel_bytes[dom] / (st_f[dom] * 1e-3) / 1e9, 1)
        dominant["frac_of_peak"] = round(kernel_bytes[dom] / (st_f[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    traffic = load_traffic_profile(args, algo_f)

    # the same forward as a stand-alone call (no binning kept for a pullback: compact records,
    # no slot map) -- what a forward-only user of raster! gets
    fevs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            for _ in range(reps)]
    fwd(keep=False)
    for e0, e1 in fevs:
        e0.record()
        fwd(keep=False)
        e1.record()
    torch.cuda.synchronize()
    ms_fwd_alone = float(np.mean([e0.elapsed_time(e1) for e0, e1 in fevs]))
    roof = {
        "bound": "hbm", "kernel": "raster! (all launches of one forward call, as run in the step)",
        "achieved": round(a_fwd / (ms_fwd * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(a_fwd / (ms_fwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "traffic": traffic["bytes"] if traffic else None,
        "traffic_source": traffic["source"] if traffic else None,
        "dominant_kernel": dominant,
        "algorithmic_bytes": a_fwd, "ms": round(ms_fwd, 4),
        "frac_of_measured_copy_peak": round(a_fwd / (ms_fwd * 1e-3) / 1e9 / HBM_COPY_GBS, 4),
        "forward_stand_alone": {"ms": round(ms_fwd_alone, 4),
                                "achieved": round(a_fwd / (ms_fwd_alone * 1e-3) / 1e9, 2),
                                "frac": round(a_fwd / (ms_fwd_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "pullback": {"algorithmic_bytes": a_bwd, "ms": round(ms_bwd, 4),
                     "achieved": round(a_bwd / (ms_bwd * 1e-3) / 1e9, 2),
                     "frac": round(a_bwd / (ms_bwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    if stages:
        roof["stages"] = stages

    line = {
        "metric": "M points/s fwd+bwd, 10M pts→256³ grid; HBM GB/s vs roofline",
        "value": round(value, 3), "unit": "M points/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dt, "data": "synthetic",
        "config": {"workload": f"{args.config}: {P} 3-D points ({ {'gauss': '0.4*N(0,I)', 'uniform': 'uniform(-.55,.55)', 'tight': '0.1*N(0,I)'}[args.dist] }, {args.order} order) -> "
                               f"{'x'.join(map(str, grid))} {dt} grid, one pose per GPU, "
                               f"raster! + raster_pullback!",
                   "algo": {"raster": algo_f, "pullback": algo_b}, "pullback_reuses_forward_binning": share,
                   "poses_global": world, "point_order": args.order,
                   "exchange": (f"all-reduce(sum) of [ds_dpoints|ds_dpoint_weight] ({backend})"
                                + (", overlapped with the next step's kernels (double buffer)"
                                   if overlap else ", inside the step")
                                if world > 1 else "none")},
        "roofline": roof,
    }
    if world == 1 and args.order == "random" and not args.no_secondary:
        # secondary line: the same cloud pre-sorted once in the model frame (Morton order; the
        # sort is pose-independent, so a user amortises it over poses and iterations)
        inp["points"], _perm = dpr_amd.sort_points(inp["points"])  # dpr_sort_points_f32
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps
        st_fm = dpr_amd.stage_times(fwd, "raster", algo_f, reps)
        st_bm = dpr_amd.stage_times(bwd, "pullback", algo_b, reps, prepare=fwd)
        line["coherent_input"] = {
            "point_order": "morton (dpr_sort_points once, not timed)", "value": round(P / el / 1e6, 3),
            "unit": "M points/s", "ms_per_step": round(el * 1e3, 4),
            "raster_ms": round(st_fm["total"], 4), "pullback_ms": round(st_bm["total"], 4),
            "raster_frac_of_hbm_peak": round(a_fwd / (st_fm["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.config, inp, args.cpu_budget)
    if rank == 0:
        print(json.dumps(line, ensure_ascii=False))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
