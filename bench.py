#!/usr/bin/env python3
"""Benchmark of the raster! / raster_pullback! hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config C1..C5] [--shard poses|points]

BASELINE.json metric: "M points/s fwd+bwd, 10M pts -> 256^3 grid; HBM GB/s vs roofline".
A "step" is one raster! + one raster_pullback! over this rank's share of the workload with all
inputs already resident in HBM.  `value` counts (point, pose) pairs processed per second by
the whole job.

  N = 1 (default)   config C3 = BASELINE.json configs[2]: 10 M 3-D points -> 256^3 fp32, one pose.
  N > 1 (default)   config C4 = configs[3]: 10 M points -> 512^2 projections, 512 poses GLOBAL,
                    sharded over the ranks with `shard_range` (STRONG scaling: the job is fixed),
                    one all-reduce(sum) of the fused [ds_dpoints | ds_dpoint_weight] buffer per
                    step -- the only exchange the batched pullback has
                    (/root/reference/src/raster_pullback.jl:112-147).  `--config C4 --gpus 1` is
                    the 1-GPU point of the same curve; `--config C5` is the 50 M -> 512^3 fp64,
                    64-pose job.
  single-pose configs at N > 1:  `--shard poses` (weak: one pose per rank, all-reduce of the
                    point gradients) or `--shard points` (strong: each rank owns a block of the
                    points, all-reduce of the grid forward, of 13 scalars backward).

Launching: with N > 1 and no RANK/WORLD_SIZE in the environment this script starts its own N
rank processes (children are started BEFORE anything touches the GPU; the parent never does)
with MASTER_ADDR=127.0.0.1.  Under `python -m torch.distributed.run ... bench.py --gpus N` the
ranks are already there and each process runs as one rank.  One rank per GPU, RCCL ("nccl").
DPR_BENCH_BACKEND=gloo is a rehearsal of the control flow on a box with fewer GPUs than ranks
(ranks share devices, the exchange goes through host memory); never a measured configuration.

Besides the contract's JSON fields the line carries `roofline` (forward call, HIP events on the
launch stream, algorithmic bytes over device time), `no_share` (the same step when the pullback
re-bins instead of reusing the forward's binning: what the plain C entry points give) and, on
rank 0 at N = 1, `cpu_baseline` (threaded CPU port of the reference algorithm from oracle/, on
a bounded sample of the same workload).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "M points/s fwd+bwd, 10M pts→256³ grid; HBM GB/s vs roofline"
CONFIGS = {
    # BASELINE.json configs[0..4]
    "C1": dict(P=1_000, n_in=2, grid=(5, 5), dt="f64", B=1, identity=True, ops="fwd+bwd",
               what="1k random 2-D points -> 5x5, identity pose (plumbing)"),
    "C2": dict(P=1_000_000, n_in=3, grid=(128, 128, 128), dt="f32", B=1, ops="fwd",
               what="1M 3-D points -> 128^3 fp32, single pose, forward raster!"),
    "C3": dict(P=10_000_000, n_in=3, grid=(256, 256, 256), dt="f32", B=1, ops="fwd+bwd",
               what="10M 3-D points -> 256^3 fp32, single pose, raster! + raster_pullback!"),
    "C4": dict(P=10_000_000, n_in=3, grid=(512, 512), dt="f32", B=512, ops="fwd+bwd",
               what="10M 3-D points -> 512^2 orthographic projections, batch of 512 poses"),
    "C5": dict(P=50_000_000, n_in=3, grid=(512, 512, 512), dt="f64", B=64, ops="fwd+bwd",
               what="50M 3-D points -> 512^3 fp64, batch of 64 poses"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # measured float4 copy ceiling (same guide)


def morton_order(pts, bits=10):
    """Pose-independent spatial pre-sort of the model-frame points (Morton / Z-order)."""
    q = np.clip(((pts * 0.5 + 0.5) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(pts), dtype=np.int64)
    for b in range(bits):
        for d in range(pts.shape[1]):
            code |= ((q[:, d] >> b) & 1) << (pts.shape[1] * b + d)
    return np.argsort(code, kind="stable")


def synth_points(cfg, order="random", dist_kind="gauss"):
    """points 0.4*N(0,I), seed 0 (test/data.jl:22,27)."""
    c = CONFIGS[cfg]
    P, n_in = c["P"], c["n_in"]
    npdt = np.float32 if c["dt"] == "f32" else np.float64
    rng = np.random.default_rng(0)
    pts = (0.4 * rng.standard_normal(size=(P, n_in), dtype=np.float32)).astype(npdt)
    if dist_kind == "uniform":  # diagnostic only (balanced tiles); not the headline workload
        pts = (1.1 * rng.random(size=(P, n_in), dtype=np.float32) - 0.55).astype(npdt)
    if dist_kind == "tight":  # diagnostic only: a compact cluster, few heavily loaded tiles
        pts = (0.1 * rng.standard_normal(size=(P, n_in), dtype=np.float32)).astype(npdt)
    if order == "morton":
        pts = pts[morton_order(pts)]
    return pts


def synth_poses(cfg, B, seed):
    """rotation uniform on SO(3) (projection = its first rows), translation 0.1*N(0,I)
    (test/data.jl:29-30, 54-56); C1: identity pose."""
    from tests import data as D

    c = CONFIGS[cfg]
    n_in, n_out = c["n_in"], len(c["grid"])
    npdt = np.float32 if c["dt"] == "f32" else np.float64
    prng = np.random.default_rng(seed)
    if c.get("identity"):
        R = np.broadcast_to(np.eye(n_out, n_in), (B, n_out, n_in)).copy()
        t = np.zeros((B, n_out))
    else:
        R = D.random_rotations(prng, B, n_in)[:, :n_out, :]
        t = 0.1 * prng.normal(size=(B, n_out))
    return R.astype(npdt), t.astype(npdt)


def algorithmic_bytes(cfg, B, P=None, with_point_weight=False):
    """A_fwd = s[P(N_in+w) + B G]; A_bwd = s[P(N_in+w) + B G + P N_in + P]."""
    c = CONFIGS[cfg]
    P = c["P"] if P is None else P
    s = 4 if c["dt"] == "f32" else 8
    G = int(np.prod(c["grid"]))
    w = 1 if with_point_weight else 0
    a_fwd = s * (P * (c["n_in"] + w) + B * G)
    a_bwd = s * (P * (c["n_in"] + w) + B * G + P * c["n_in"] + P)
    return a_fwd, a_bwd


def load_traffic_profile(cfg, algo_f, order, poses=None):
    """HBM bytes per forward call from the PMC counters (FETCH_SIZE / WRITE_SIZE, collected in
    separate rocprofv3 passes of this same command and corrected as MI355X_MICROARCH.md
    prescribes); measured offline, committed under profiles/ -- bench.py cannot profile itself."""
    for name in ("r06_hbm_traffic.json", "r05_c3_hbm_traffic.json", "r04_c3_hbm_traffic.json", "r03_c3_hbm_traffic.json",
                 "r02_c3_hbm_traffic.json", "r01_c3_hbm_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            prof = json.load(f)
        key = f"{cfg}/{algo_f}/{order}" + (f"/B{poses}" if poses and poses > 1 else "")
        ent = prof.get("forward", {}).get(key)
        if ent:
            return {"bytes": ent["hbm_bytes_corrected"], "source": f"profiles/{name}[{key}]"}
    return None


def cpu_baseline(cfg, np_points, np_R, np_t, np_g, budget_s=20.0):
    """Threaded CPU port of the reference algorithm (oracle/, kind "port") on a bounded sample:
    the first `n` points of the same cloud (and at most 2 poses) into the same grid, n chosen
    so that the pass takes roughly `budget_s` seconds."""
    from oracle import oracle

    c = CONFIGS[cfg]
    P, grid = c["P"], c["grid"]
    npdt = np.float32 if c["dt"] == "f32" else np.float64
    threads = oracle.max_threads()
    nb = min(2, np_R.shape[0])
    R, t = np_R[:nb], np_t[:nb]  # the oracle takes batched arguments
    bwd = "bwd" in c["ops"]
    g = None
    if bwd:
        g = np_g if np_g.ndim == len(grid) + 1 else np_g[..., None]
        g = np.asfortranarray(g[..., :nb])

    def run(n):
        t0 = time.perf_counter()
        oracle.raster(grid, np_points[:n], R, t, dtype=npdt, threaded=True)
        t1 = time.perf_counter()
        if bwd:
            oracle.raster_pullback(g, np_points[:n], R, t, dtype=npdt, threaded=True)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    n = min(P, 200_000)
    f, b = run(n)  # calibration = the warm-up pass (includes the fixed grid fill / grid sum cost)
    rate = n / max(f + b, 1e-9)
    # best of 3 after one warm-up -- the budget covers the three passes
    n2 = int(min(P, max(n, rate * budget_s / 3.0)))
    passes = [run(n2) for _ in range(3)]
    f, b = min(passes, key=lambda fb: fb[0] + fb[1])
    poses = nb
    res = {
        "value": round(n2 * poses / (f + b) / 1e6, 4), "unit": "M points/s", "cores": threads,
        "kind": "port",
        "timing": "best of 3 passes after one warm-up pass (BASELINE.md section 4)",
        "passes_s": [round(x + y, 3) for x, y in passes],
        "sample": f"first {n2} of {P} points x {poses} pose(s), same grid/poses; fwd {f:.2f}s on "
                  f"{threads} threads (atomic scatter)"
                  + (f", bwd {b:.2f}s (threads over pose chunks; a single pose is serial, "
                     f"src/raster_pullback.jl:39,115-139)" if bwd else ""),
        "fwd_M_points_s": round(n2 * poses / f / 1e6, 4),
    }
    if bwd:
        res["bwd_M_points_s"] = round(n2 * poses / b / 1e6, 4)
    return res


# ------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """Start `n` rank processes of this script (one per GPU) and relay rank 0's output.  The
    parent imports neither torch nor the HIP library: nothing here touches a GPU."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:  # one rank failed: the others would wait for it forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


# ------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import torch

    import dpr_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} "
                         f"(launch with --nproc-per-node {args.gpus}, or without a launcher)")
    backend = os.environ.get("DPR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()  # does not initialise the GPU
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if backend == "nccl" and world > ndev:
        raise SystemExit(
            f"bench.py --gpus {world} needs {world} visible HIP devices, found {ndev} (one rank "
            f"per GPU over RCCL).  DPR_BENCH_BACKEND=gloo rehearses the control flow with ranks "
            f"sharing devices; it is not a measured configuration.")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # --force-exchange: a one-rank RCCL group, no launcher needed
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    cfg = args.config or ("C3" if world == 1 else "C4")
    if args.force_exchange and not CONFIGS[cfg]["B"] > 1:
        raise SystemExit("--force-exchange drives the pose-sharded exchange: use a batched config (C4, C5)")
    line = run_job(args, cfg, rank, world, device, dist, backend)
    if world == 1 and args.config is None and not args.no_scaling_reference and not args.force_exchange:
        # The N > 1 runs of this script time another job than the metric's C3 (C4: 512 poses,
        # strong scaling).  Its one-GPU point, measured here in the same process, is what
        # value(N) of an N-GPU line has to be divided by for same-job scaling.
        ref_args = argparse.Namespace(**vars(args))
        ref_args.config, ref_args.poses, ref_args.order, ref_args.coherent = "C4", None, "random", False
        ref_args.dist, ref_args.algo, ref_args.shard = "gauss", "auto", None
        ref_args.steps, ref_args.warmup = max(2, min(args.steps, 5)), 1
        torch.cuda.empty_cache()
        ref = run_job(ref_args, "C4", rank, world, device, dist, backend, lean=True)
        # ... and the share one of 8 GPUs owns (64 poses), for the ceiling of that curve: the job
        # on one GPU over 8 x the share's time is the speed-up 8 GPUs reach BEFORE any exchange.
        share_args = argparse.Namespace(**vars(ref_args))
        share_args.poses = 64
        torch.cuda.empty_cache()
        sh = run_job(share_args, "C4", rank, world, device, dist, backend, lean=True)
        ceiling = ref["ms_per_step"] / sh["ms_per_step"]
        # ... and the same two for a caller that sorts the cloud ONCE outside the step (dpr_sort_points_* +
        # DPR_FLAG_COHERENT_POINTS: pose refinement over a fixed cloud -- what the reference's example does):
        # the in-call sort is a per-call cost that does not shrink with the pose share
        so_args = argparse.Namespace(**vars(ref_args))
        so_args.order, so_args.coherent = "hilbert", True
        torch.cuda.empty_cache()
        so_ref = run_job(so_args, "C4", rank, world, device, dist, backend, lean=True)
        so_share_args = argparse.Namespace(**vars(so_args))
        so_share_args.poses = 64
        torch.cuda.empty_cache()
        so_sh = run_job(so_share_args, "C4", rank, world, device, dist, backend, lean=True)
        line["scaling_reference"] = {
            "what": "the job of the --gpus N > 1 runs (C4: 10M points -> 512^2, 512 poses, fwd+bwd, "
                    "AUTO) on ONE GPU, same process: value(N) / this value = same-job strong scaling",
            "command": "python bench.py --config C4 --gpus 1",
            "value": ref["value"], "unit": ref["unit"], "ms_per_step": ref["ms_per_step"],
            "steps": ref["steps"], "warmup": ref["warmup"], "poses_global": ref["config"]["poses_global"],
            "algo": ref["config"]["algo"],
            "pullback_reuses_forward_binning": ref["config"]["pullback_reuses_forward_binning"],
            "share_of_one_gpu_of_8": {"poses": 64, "ms_per_step": sh["ms_per_step"],
                                      "command": "python bench.py --config C4 --poses 64"},
            "predicted_speedup_at_8_gpus_before_exchange": round(ceiling, 2),
            "exposed_exchange_budget_ms_for_6x": round(ref["ms_per_step"] / 6.0 - sh["ms_per_step"], 3),
            "cloud_sorted_once_outside_the_step": {
                "what": "the same job and share on a cloud Hilbert-sorted once by the caller (dpr_sort_points, "
                        "not timed) + DPR_FLAG_COHERENT_POINTS",
                "ms_per_step": so_ref["ms_per_step"], "value": so_ref["value"],
                "share_of_one_gpu_of_8_ms_per_step": so_sh["ms_per_step"],
                "predicted_speedup_at_8_gpus_before_exchange": round(so_ref["ms_per_step"] / so_sh["ms_per_step"], 2),
                "exposed_exchange_budget_ms_for_6x": round(so_ref["ms_per_step"] / 6.0 - so_sh["ms_per_step"], 3),
            },
            "prediction_is": "a ceiling computed from one-GPU measurements (job time / share time); "
                             "the 160 MB all-reduce per step overlaps the next step's kernels, what "
                             "of it stays exposed has to fit the budget above for >= 6x; NOT measured "
                             "on more than one GPU",
        }
    if rank == 0:
        print(json.dumps(line, ensure_ascii=False), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_job(args, cfg, rank, world, device, dist, backend, lean=False):
    """One measured job on this rank; returns the JSON line (rank 0 prints it).  `lean`: only
    the timed steps (no stage split, no no-share / coherent / CPU secondary measurements)."""
    import torch

    import dpr_amd

    c = CONFIGS[cfg]
    P, n_in, grid, dt = c["P"], c["n_in"], c["grid"], c["dt"]
    n_out = len(grid)
    s_elem = 4 if dt == "f32" else 8
    tdt = torch.float32 if dt == "f32" else torch.float64
    batched = c["B"] > 1
    do_bwd = "bwd" in c["ops"]
    shard = args.shard or ("poses" if batched or world == 1 else "poses")
    if batched and shard != "poses":
        raise SystemExit("batched configs shard over poses")

    # ---- the rank's share of the job
    np_pts = synth_points(cfg, args.order, args.dist)
    if batched:
        B_global = args.poses or c["B"]
        lo, hi = dpr_amd.shard_range(B_global, rank, world)
        np_R, np_t = synth_poses(cfg, B_global, seed=1)
        np_R, np_t = np_R[lo:hi], np_t[lo:hi]
        scaling = "strong"
        P_local, p_lo = P, 0
    elif shard == "points" and world > 1:
        B_global = 1
        lo, hi = 0, 1
        np_R, np_t = synth_poses(cfg, 1, seed=1)
        p_lo, p_hi = dpr_amd.shard_range(P, rank, world)
        np_pts = np_pts[p_lo:p_hi]
        P_local = p_hi - p_lo
        scaling = "strong"
    else:  # one pose per rank
        B_global = world
        lo, hi = rank, rank + 1
        np_R, np_t = synth_poses(cfg, 1, seed=1 + rank)
        scaling = "weak"
        P_local, p_lo = P, 0
    B_local = hi - lo
    to = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)
    points = to(np_pts)
    if args.order == "hilbert":
        points = dpr_amd.sort_points(points)[0]
    co = dict(coherent_points=True) if args.coherent else {}
    single_call = not batched  # single-pose signature (rotation is a matrix)
    # (the pose does not change between the steps: it is handed over once in the C ABI's memory order,
    # what a Julia host's Vector{SMatrix} is anyway -- no transpose kernels inside the timed step)
    R = dpr_amd.column_major_rotation(to(np_R[0] if single_call else np_R))
    t = to(np_t[0] if single_call else np_t)
    gen = torch.Generator(device=device)
    gen.manual_seed(2 + (0 if shard == "points" else rank))
    gshape = tuple(reversed(grid)) if single_call else (B_local,) + tuple(reversed(grid))
    g = None
    if do_bwd and B_local > 0:
        g = torch.randn(gshape, device=device, dtype=tdt, generator=gen)
        g = g.permute(*reversed(range(g.ndim)))  # [i1, .., iN(, b)] view, reference memory order
    out = dpr_amd.empty_grid(grid, None if single_call else B_local, tdt, device)
    fused = torch.empty(P_local * (n_in + 1), dtype=tdt, device=device)
    Bq = max(B_local, 1)
    # The pullback reuses the tile binning its forward call built in the same step (what an
    # rrule caches between `raster` and its pullback closure); nothing is carried across steps.
    # Batched calls share on the chunk-owner path (2-D grids): what is kept there is the sorted copy
    # of the cloud, for any number of poses.  With keep / reuse flags DPR_ALGO_AUTO decides for
    # the pair of calls (include/dpr.h); the names are resolved here so that the no-share run and
    # the stage list use exactly the algorithms of the timed step.
    def shareable(a):
        if a == "chunked" and n_out == 3:
            return False  # (its pullback gathers directly and reads nothing a forward could leave)
        if (single_call and a in ("tiled", "chunked")) or (a == "chunked" and n_out == 2):
            return True
        # a batch on the tiled path: every pose keeps its own binning (B-fold records) where
        # DPR_ALGO_AUTO judges that affordable (dpr_resolve_flags_ex)
        return a == "tiled" and dpr_amd.sharing_effective(grid, P_local, Bq, n_in, **co)

    def algos(sharing):
        if args.algo != "auto":
            return args.algo, args.algo
        return tuple(dpr_amd.resolve_algo(op, grid, P_local, Bq, n_in, sharing=sharing, **co)
                     for op in ("raster", "pullback"))

    algo_f, algo_b = algos(sharing=do_bwd and not args.no_share_binning)
    can_share = do_bwd and algo_f == algo_b and shareable(algo_f)
    share = can_share and not args.no_share_binning
    if not share:
        algo_f, algo_b = algos(sharing=False)
    ws_bytes = max(dpr_amd.workspace_bytes(op, grid, P_local, Bq, n_in, tdt, a, sharing=sh, **co)
                   for op in ("raster", "pullback")
                   for a in {algo_f, algo_b} | set(algos(False))
                   for sh in ({False, True} if share else {False}))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=device)

    def fwd(keep=None):
        if B_local > 0:
            dpr_amd.raster_(out, points, R, t, algo=algo_f, workspace=ws,
                            keep_binning=share if keep is None else keep, **co)

    def bwd(buf=fused, reuse=None):
        if B_local == 0:
            buf.zero_()  # a rank may own no pose when B < world size
            return None
        return dpr_amd.raster_pullback_(g, points, R, t, ds_dpoints=buf[: P_local * n_in].view(P_local, n_in),
                                 ds_dpoint_weight=buf[P_local * n_in:], algo=algo_b, workspace=ws,
                                 reuse_binning=share if reuse is None else reuse, **co)

    # ---- exchange.  Pose sharding: all-reduce(sum) of the fused point-gradient buffer; by
    # default step k's all-reduce runs on RCCL's stream under step k+1's kernels (double buffer;
    # every all-reduce completes inside the timed region).  Point sharding: the forward
    # all-reduces the grid, the pullback the per-pose scalars (13 values), inside the step.
    # --force-exchange: the N-rank exchange step (double buffer, async all-reduce, wait before
    # the buffer is written again) over a ONE-rank RCCL group -- the code path of every N > 1 run,
    # executable on a one-GPU box
    pose_exchange = (world > 1 or args.force_exchange) and shard == "poses" and do_bwd
    overlap = pose_exchange and backend == "nccl" and not args.no_overlap_exchange
    fused_alt = torch.empty_like(fused) if overlap else None
    pending = [None, None]
    step_no = [0]

    def all_reduce(buf, async_op=False):
        if backend == "nccl":
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=async_op)
        host = buf.cpu()  # rehearsal backend: through host memory
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        buf.copy_(host)
        return None

    def step(share_step=None):
        k = step_no[0] & 1 if overlap else 0
        step_no[0] += 1
        buf = fused_alt if k else fused
        if pending[k] is not None:
            pending[k].wait()  # the launch stream waits until this buffer's all-reduce is done
            pending[k] = None
        fwd(share_step)
        if world > 1 and shard == "points":
            all_reduce(out.permute(*reversed(range(out.ndim))))  # the contiguous grid buffer
        if do_bwd:
            res = bwd(buf, share_step)
            if pose_exchange:
                pending[k] = all_reduce(buf, async_op=overlap)
            elif world > 1 and shard == "points":
                # the per-pose sums of the one pose: ds_drotation | ds_dtranslation | ds_dout_weight
                all_reduce(torch.cat([res.rotation.reshape(-1), res.translation.reshape(-1),
                                      res.out_weight.reshape(-1)]))

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

    def timed(n_steps, share_step=None):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(share_step)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt[0])
        return el

    # Clock spin-up (untimed, before the W warm-up steps): the timed region starts at idle clocks
    # otherwise -- W = 3 steps of C3 are 1.3 ms of GPU work, and 20 timed steps 8 ms: measured
    # 0.420 ms per step without, 0.396 with ~100 ms of load in front.  The same step is used;
    # the count comes from one timed step (max over ranks), so every rank runs the same number.
    step()  # first call: code objects are loaded lazily
    t_one = timed(1)
    # NO-SPIN-UP figure first: the same K steps as the timed region below without the ~100 ms of
    # load in front (only the first call and one probe step precede it, so the clocks are partly
    # up) -- what rounds 1-2 reported as ms_per_step; kept next to the spun-up figure so that
    # rounds stay comparable.
    ms_cold = timed(args.steps) / args.steps * 1e3 if not lean else None
    n_spin = max(0, min(2000, int(args.spin_up_ms * 1e-3 / max(t_one, 1e-6)))) if args.spin_up else 0
    if n_spin > 0:
        timed(n_spin)
    for _ in range(args.warmup):
        step()
    # The timed region: K steps between barrier + synchronize on both sides (max over ranks).
    # It is ALWAYS timed `--loops` times (default 3) and the MEDIAN loop is the line's value --
    # unconditionally, so that a stray stall of tens of ms (host or device, ~3 % of the runs on
    # these boxes) inside a region of a few ms neither becomes the number nor triggers a
    # conditional re-run that would bias it downwards; every loop is listed in the line.
    loops = [timed(args.steps) for _ in range(1 if lean else max(1, args.loops))]
    elapsed = float(np.median(loops))
    ms_per_step = elapsed / args.steps * 1e3
    units = (B_global * P) if shard != "points" else P  # (point, pose) pairs per step, whole job
    value = units / (elapsed / args.steps) / 1e6
    untimed_steps = 2 + (0 if lean else args.steps) + n_spin + args.warmup

    # ---- per-pass device time with HIP events on the launch stream (torch's current stream);
    # forward and pullback are timed inside fwd+bwd pairs (the pullback consumes -- and, when it
    # reuses the binning, destroys -- what its forward left in the workspace)
    reps = max(3, min(args.steps, 20))
    ev = lambda: torch.cuda.Event(enable_timing=True)
    evs = [(ev(), ev(), ev()) for _ in range(reps)]
    for e0, e1, e2 in evs:
        e0.record()
        fwd()
        e1.record()
        if do_bwd:
            bwd()
        e2.record()
    torch.cuda.synchronize()
    def avg_ms(samples):
        """Median of the event-timed calls (robust against a stray stall -- one 47 ms hiccup in 20
        calls of 0.08 ms once made the mean 2.4 ms -- without dropping samples)."""
        return float(np.median(np.asarray(samples, dtype=np.float64)))

    ms_fwd = avg_ms([e0.elapsed_time(e1) for e0, e1, _ in evs])
    ms_bwd = avg_ms([e1.elapsed_time(e2) for _, e1, e2 in evs])
    a_fwd, a_bwd = algorithmic_bytes(cfg, max(B_local, 1), P_local)
    gbs = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9
    roof = {
        "bound": "hbm", "kernel": "raster! (all launches of one forward call, as run in the step)",
        "achieved": round(gbs(a_fwd, ms_fwd), 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(gbs(a_fwd, ms_fwd) / HBM_PEAK_GBS, 4), "traffic": None,
        "algorithmic_bytes": a_fwd, "ms": round(ms_fwd, 4),
        "ms_is": f"median of {reps} event-timed calls",
        "frac_of_measured_copy_peak": round(gbs(a_fwd, ms_fwd) / HBM_COPY_GBS, 4),
    }
    traffic = load_traffic_profile(cfg, algo_f, args.order, B_local if batched else None)
    if traffic:
        roof["traffic"], roof["traffic_source"] = traffic["bytes"], traffic["source"]
    if do_bwd:
        roof["pullback"] = {"algorithmic_bytes": a_bwd, "ms": round(ms_bwd, 4),
                            "achieved": round(gbs(a_bwd, ms_bwd), 2),
                            "frac": round(gbs(a_bwd, ms_bwd) / HBM_PEAK_GBS, 4)}
    if do_bwd and B_local > 0 and not lean:
        # what the reference's rrule needs when point_weight was defaulted (the benchmark's case): no
        # ds_dpoint_weight (DPR_FLAG_NO_POINT_WEIGHT_GRAD) -- a P-element store less
        def bwd_nopw():
            return dpr_amd.raster_pullback_(g, points, R, t, ds_dpoints=fused[: P_local * n_in].view(P_local, n_in),
                                            algo=algo_b, workspace=ws, reuse_binning=share, point_weight_grad=False, **co)
        nevs = [(ev(), ev()) for _ in range(reps)]
        for e0, e1 in nevs:
            fwd()
            e0.record()
            bwd_nopw()
            e1.record()
        torch.cuda.synchronize()
        ms_nopw = avg_ms([e0.elapsed_time(e1) for e0, e1 in nevs])
        a_nopw = a_bwd - s_elem * P_local
        roof["pullback"]["without_point_weight_grad"] = {
            "flag": "DPR_FLAG_NO_POINT_WEIGHT_GRAD", "ms": round(ms_nopw, 4),
            "algorithmic_bytes": a_nopw, "algorithmic_bytes_is": "A_bwd without the P-element ds_dpoint_weight store",
            "achieved": round(gbs(a_nopw, ms_nopw), 2), "frac": round(gbs(a_nopw, ms_nopw) / HBM_PEAK_GBS, 4)}
    if batched:
        roof["note"] = ("batched poses re-read the points per pose (group): the per-call "
                        "algorithmic bytes count them once, BASELINE.md section 3")
    if single_call and B_local == 1 and not lean:
        local = args.coherent and algo_f == "tiled" and n_out == 3 or (args.coherent and algo_f == "tiled")
        sname = lambda a: ("tiled_local" if (local and a == "tiled") else
                           ("chunked2d" if (a == "chunked" and n_out == 2) else a))
        st_f = dpr_amd.stage_times(fwd, "raster", sname(algo_f), reps)
        stages = {"raster": {"algo": algo_f, **{k: round(v, 4) for k, v in st_f.items()}}}
        if do_bwd:
            st_b = dpr_amd.stage_times(bwd, "pullback", sname(algo_b), reps, prepare=fwd)
            stages["pullback"] = {"algo": algo_b, **{k: round(v, 4) for k, v in st_b.items()}}
        roof["stages"] = stages
        dom = max((k for k in st_f if k != "total"), key=lambda k: st_f[k])
        roof["dominant_stage"] = {"stage": dom, "ms": round(st_f[dom], 4)}
        if can_share:
            # the same forward as a stand-alone call (no binning kept for a pullback)
            fevs = [(ev(), ev()) for _ in range(reps)]
            fwd(keep=False)
            for e0, e1 in fevs:
                e0.record()
                fwd(keep=False)
                e1.record()
            torch.cuda.synchronize()
            ms_alone = avg_ms([e0.elapsed_time(e1) for e0, e1 in fevs])
            roof["forward_stand_alone"] = {"ms": round(ms_alone, 4),
                                           "achieved": round(gbs(a_fwd, ms_alone), 2),
                                           "frac": round(gbs(a_fwd, ms_alone) / HBM_PEAK_GBS, 4)}

    exchange = "none"
    if pose_exchange:
        exchange = (f"all-reduce(sum) of [ds_dpoints|ds_dpoint_weight] ({backend})"
                    + (", overlapped with the next step's kernels (double buffer)" if overlap
                       else ", inside the step"))
    elif world > 1 and shard == "points":
        exchange = f"all-reduce(sum) of the grid (forward) and of the per-pose scalars ({backend})"
    line = {
        "metric": METRIC, "value": round(value, 3), "unit": "M points/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "timing": f"median of {len(loops)} loop(s) of {args.steps} steps, each between barrier + synchronize",
        "ms_per_step_loops": [round(x / args.steps * 1e3, 4) for x in loops],
        **({"ms_per_step_cold": round(ms_cold, 4),
            "ms_per_step_cold_is": "the same K steps timed before the clock spin-up (after the first call "
                                   "and one probe step only): 'no spin-up', not 'from idle'"} if ms_cold is not None else {}),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": dt,
        "data": "synthetic",
        "config": {
            "workload": f"{cfg}: {c['what']}; points "
                        f"{ {'gauss': '0.4*N(0,I)', 'uniform': 'uniform(-.55,.55)', 'tight': '0.1*N(0,I)'}[args.dist] }"
                        f", {args.order} order; {c['ops']}",
            "algo": {"raster": algo_f, "pullback": algo_b if do_bwd else None},
            "pullback_reuses_forward_binning": bool(share), "poses_global": B_global,
            "poses_per_rank": B_local if world == 1 else f"{B_global // world}..{-(-B_global // world)}",
            "sharding": shard if world > 1 else "none", "point_order": args.order,
            "coherent_points_flag": bool(args.coherent),
            "exchange": exchange,
            "value_counts": "points x poses of the whole job per second (a point counts once per pose)",
            "untimed_steps": untimed_steps,
            "untimed_before_the_timed_steps": f"first call + 1 probe step"
                + ("" if lean else f" + {args.steps} steps timed from idle clocks (ms_per_step_cold)")
                + f" + {n_spin} spin-up steps (~{args.spin_up_ms:.0f} ms of load so that the clocks are up) + {args.warmup} warm-up steps",
            **({"ranks_seen": int(dist.get_world_size()),
                "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version())
                                 if backend == "nccl" else f"({backend}: rehearsal, no RCCL)")}
               if dist is not None else {}),
            **({"same_job_on_one_gpu": f"python bench.py --config {cfg} --gpus 1"
                                        + (f" --poses {B_global}" if args.poses else "")
                                        + "  (the default --gpus 1 run is the metric's config C3, a different job)"}
               if world > 1 else {})},
        "roofline": roof,
    }
    if can_share and not args.no_share_binning and world == 1 and not lean:
        # the drop-in number: plain entry points (no KEEP/REUSE flags), the pullback re-bins
        for _ in range(max(1, args.warmup)):
            step(False)
        # (a secondary figure: median of 3 loops, like the headline)
        el = float(np.median([timed(args.steps, False) for _ in range(3 if world == 1 else 1)]))
        line["config"]["drop_in_ms_per_step"] = round(el / args.steps * 1e3, 4)
        line["config"]["rotation_layout"] = ("column-major N_out x N_in per pose (the reference's SMatrix memory, what the C "
                                             "ABI takes), prepared ONCE outside the step with column_major_rotation(); "
                                             "rounds 1-4 passed a row-major torch tensor and the host mirror transposed it "
                                             "inside every call (two copy kernels, ~9 us per step at C3): their "
                                             "ms_per_step includes that, round 5+ does not")
        line["config"]["drop_in_is"] = ("the same step through the plain dpr_raster_* / dpr_raster_pullback_* "
                                        "entry points (no KEEP / REUSE flags): the `no_share` entry")
        line["no_share"] = {"timing": "median of 3 loops" if world == 1 else "one loop",
                            "ms_per_step": round(el / args.steps * 1e3, 4),
                            "value": round(units / (el / args.steps) / 1e6, 3),
                            "unit": "M points/s",
                            "what": "pullback re-bins (plain dpr_raster_* / dpr_raster_pullback_* "
                                    "entry points, no DPR_FLAG_KEEP/REUSE_BINNING)"}
    if (world == 1 and args.order == "random" and not args.coherent and not args.no_secondary
            and cfg in ("C2", "C3") and not lean):
        # secondary line: the same cloud pre-sorted once in the model frame (Morton order; the
        # sort is pose-independent, so a user amortises it over poses and iterations)
        sorted_pts, _perm = dpr_amd.sort_points(points)  # dpr_sort_points_f32
        points_random = points
        points = sorted_pts
        # the caller of dpr_sort_points may say so: DPR_FLAG_COHERENT_POINTS.  DPR_ALGO_AUTO then
        # decides anew for the pair (on a 3-D grid the pullback of one pose gathers directly in cloud
        # order and the pair shares nothing; include/dpr.h)
        coherent_kw = dict(coherent_points=True)
        if args.algo != "auto":
            algo_fc = algo_bc = args.algo
        else:
            algo_fc, algo_bc = (dpr_amd.resolve_algo(op, grid, P, 1, n_in, sharing=do_bwd, coherent_points=True)
                                for op in ("raster", "pullback"))
        share_c = (do_bwd and algo_fc == algo_bc and not args.no_share_binning
                   and dpr_amd.sharing_effective(grid, P, 1, n_in, coherent_points=True))
        ws_c = torch.empty(max(16, *(dpr_amd.workspace_bytes(op, grid, P, 1, n_in, tdt, a, coherent_points=True,
                                                             sharing=share_c)
                                     for op, a in (("raster", algo_fc), ("pullback", algo_bc)))),
                           dtype=torch.uint8, device=device)

        def fwd_c(keep=None):
            dpr_amd.raster_(out, points, R, t, algo=algo_fc, workspace=ws_c,
                            keep_binning=share_c if keep is None else keep, **coherent_kw)

        def bwd_c():
            dpr_amd.raster_pullback_(g, points, R, t, ds_dpoints=fused[: P * n_in].view(P, n_in),
                                     ds_dpoint_weight=fused[P * n_in:], algo=algo_bc,
                                     workspace=ws_c, reuse_binning=share_c, **coherent_kw)

        def step_c():
            fwd_c()
            if do_bwd:
                bwd_c()

        for _ in range(args.warmup):
            step_c()
        # (a secondary entry: median of three loops, like the headline)
        els = []
        for _rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_c()
            torch.cuda.synchronize()
            els.append((time.perf_counter() - t0) / args.steps)
        el = float(np.median(els))
        sname_c = lambda a: "tiled_local" if a == "tiled" else a
        st_fm = dpr_amd.stage_times(fwd_c, "raster", sname_c(algo_fc), reps)
        coh = {"point_order": "Hilbert-sorted (dpr_sort_points once, not timed) + DPR_FLAG_COHERENT_POINTS",
               "algo": {"raster": algo_fc, "pullback": algo_bc if do_bwd else None},
               "pullback_reuses_forward_binning": bool(share_c),
               "timing": f"median of 3 loops of {args.steps} steps",
               "value": round(P / el / 1e6, 3), "unit": "M points/s",
               "ms_per_step": round(el * 1e3, 4), "raster_ms": round(st_fm["total"], 4),
               "raster_frac_of_hbm_peak": round(gbs(a_fwd, st_fm["total"]) / HBM_PEAK_GBS, 4),
               "raster_stages": {k: round(v, 4) for k, v in st_fm.items()}}
        roof_c = {"bound": "hbm", "kernel": "raster! (all launches of one forward call)",
                  "achieved": round(gbs(a_fwd, st_fm["total"]), 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(gbs(a_fwd, st_fm["total"]) / HBM_PEAK_GBS, 4), "traffic": None,
                  "algorithmic_bytes": a_fwd, "ms": round(st_fm["total"], 4)}
        traffic_c = load_traffic_profile(cfg, algo_fc, "hilbert+coherent")
        if traffic_c:
            roof_c["traffic"], roof_c["traffic_source"] = traffic_c["bytes"], traffic_c["source"]
        if do_bwd:
            st_bm = dpr_amd.stage_times(bwd_c, "pullback", sname_c(algo_bc), reps, prepare=fwd_c)
            coh["pullback_ms"] = round(st_bm["total"], 4)
            coh["pullback_stages"] = {k: round(v, 4) for k, v in st_bm.items()}
            roof_c["pullback"] = {"algorithmic_bytes": a_bwd, "ms": round(st_bm["total"], 4),
                                  "achieved": round(gbs(a_bwd, st_bm["total"]), 2),
                                  "frac": round(gbs(a_bwd, st_bm["total"]) / HBM_PEAK_GBS, 4)}
        coh["roofline"] = roof_c
        if cfg == "C3" and do_bwd and args.algo == "auto":
            # the same sorted cloud over a BATCH of 8 poses: where DPR_ALGO_AUTO takes the record-free 3-D paths
            # (owner-computes forward, direct pullback with the pose loop inside) -- next to the tiled pair it
            # took before round 5
            Bb = 8
            gen = torch.Generator(device=device)
            gen.manual_seed(7)
            Rb = torch.linalg.qr(torch.randn(Bb, 3, 3, device=device, dtype=tdt, generator=gen))[0]
            tb = 0.05 * torch.randn(Bb, 3, device=device, dtype=tdt, generator=gen)
            outb = dpr_amd.empty_grid(grid, Bb, tdt, device)
            gb = dpr_amd.empty_grid(grid, Bb, tdt, device)
            gb.normal_(generator=gen)

            def batch_ms(algo_f, algo_b):
                wsb = torch.empty(max(16, *(dpr_amd.workspace_bytes(op, grid, P, Bb, n_in, tdt, a, coherent_points=True)
                                            for op, a in (("raster", algo_f), ("pullback", algo_b)))),
                                  dtype=torch.uint8, device=device)

                def stepb():
                    dpr_amd.raster_(outb, points, Rb, tb, algo=algo_f, workspace=wsb, **coherent_kw)
                    dpr_amd.raster_pullback_(gb, points, Rb, tb, ds_dpoints=fused[: P * n_in].view(P, n_in),
                                             ds_dpoint_weight=fused[P * n_in:], algo=algo_b, workspace=wsb,
                                             **coherent_kw)
                for _ in range(2):
                    stepb()
                nb_steps = max(5, args.steps // 3)
                elb = []
                for _rep in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(nb_steps):
                        stepb()
                    torch.cuda.synchronize()
                    elb.append((time.perf_counter() - t0) / nb_steps)
                del wsb
                return float(np.median(elb))

            algo_fb, algo_bb = (dpr_amd.resolve_algo(op, grid, P, Bb, n_in, coherent_points=True)
                                for op in ("raster", "pullback"))
            el_auto = batch_ms(algo_fb, algo_bb)
            el_tiled = batch_ms("tiled", "tiled")
            coh["batch_of_8_poses"] = {
                "what": "the same sorted cloud, 8 poses per call (raster! + raster_pullback! through the plain "
                        "entry points, DPR_ALGO_AUTO + DPR_FLAG_COHERENT_POINTS); median of 3 loops",
                "algo": {"raster": algo_fb, "pullback": algo_bb},
                "ms_per_step": round(el_auto * 1e3, 4),
                "value": round(P * Bb / el_auto / 1e6, 3), "unit": "M point-poses/s",
                "tiled_pair_ms_per_step": round(el_tiled * 1e3, 4),
                "tiled_pair_is": "the same step with algo = tiled for both calls (AUTO's choice before round 5)"}
            # ... and 16 poses of the cloud in its AS-GENERATED order (no flag): from 16 poses on AUTO sorts a
            # 3-D cloud inside the pullback call and runs the same direct kernels on the copy
            del outb, gb
            Bb = 16
            Rb = torch.linalg.qr(torch.randn(Bb, 3, 3, device=device, dtype=tdt, generator=gen))[0]
            tb = 0.05 * torch.randn(Bb, 3, device=device, dtype=tdt, generator=gen)
            outb = dpr_amd.empty_grid(grid, Bb, tdt, device)
            gb = dpr_amd.empty_grid(grid, Bb, tdt, device)
            gb.normal_(generator=gen)
            points, coherent_kw = points_random, {}
            algo_fr, algo_br = (dpr_amd.resolve_algo(op, grid, P, Bb, n_in) for op in ("raster", "pullback"))

            def batch_ms_random(algo_f, algo_b):
                wsb = torch.empty(max(16, *(dpr_amd.workspace_bytes(op, grid, P, Bb, n_in, tdt, a)
                                            for op, a in (("raster", algo_f), ("pullback", algo_b)))),
                                  dtype=torch.uint8, device=device)

                def stepb():
                    dpr_amd.raster_(outb, points, Rb, tb, algo=algo_f, workspace=wsb)
                    dpr_amd.raster_pullback_(gb, points, Rb, tb, ds_dpoints=fused[: P * n_in].view(P, n_in),
                                             ds_dpoint_weight=fused[P * n_in:], algo=algo_b, workspace=wsb)
                for _ in range(2):
                    stepb()
                nb_steps = max(5, args.steps // 6)
                elb = []
                for _rep in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(nb_steps):
                        stepb()
                    torch.cuda.synchronize()
                    elb.append((time.perf_counter() - t0) / nb_steps)
                del wsb
                return float(np.median(elb))

            el_auto_r = batch_ms_random(algo_fr, algo_br)
            el_tiled_r = batch_ms_random("tiled", "tiled")
            line["batch_of_16_poses_any_order"] = {
                "what": "the C3 cloud in as-generated order (no flags), 16 poses per call, raster! + raster_pullback! "
                        "through the plain entry points with DPR_ALGO_AUTO; median of 3 loops",
                "algo": {"raster": algo_fr, "pullback": algo_br},
                "ms_per_step": round(el_auto_r * 1e3, 4),
                "value": round(P * Bb / el_auto_r / 1e6, 3), "unit": "M point-poses/s",
                "tiled_pair_ms_per_step": round(el_tiled_r * 1e3, 4),
                "tiled_pair_is": "the same step with algo = tiled for both calls (AUTO's choice before round 5)"}
            del outb, gb
        line["coherent_input"] = coh
        points = points_random
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not lean:
        np_g = None
        if do_bwd:
            np_g = (g[..., :2] if batched else g).cpu().numpy()
        line["cpu_baseline"] = cpu_baseline(cfg, np_pts, np_R, np_t, np_g, args.cpu_budget)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loops", type=int, default=3,
                    help="the K timed steps are run this many times; the median loop is the value")
    ap.add_argument("--no-spin-up", dest="spin_up", action="store_false",
                    help="skip the untimed steps that bring the clocks up before the warm-up")
    ap.add_argument("--spin-up-ms", type=float, default=100.0,
                    help="length of the untimed clock spin-up before the warm-up steps")
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: C3 at --gpus 1, C4 (512 poses, strong scaling) at --gpus N > 1")
    ap.add_argument("--shard", default=None, choices=["poses", "points"],
                    help="single-pose configs at N > 1: one pose per rank (weak) or a block of "
                         "the points per rank (strong)")
    ap.add_argument("--poses", type=int, default=None,
                    help="batched configs: global number of poses (default: the config's)")
    ap.add_argument("--algo", default="auto", choices=["auto", "atomic", "tiled", "chunked"])
    ap.add_argument("--order", default="random", choices=["random", "morton", "hilbert"],
                    help="point order in memory: as generated, Morton-sorted on the host, or "
                         "Hilbert-sorted with dpr_sort_points (both pose-independent, not timed)")
    ap.add_argument("--coherent", action="store_true",
                    help="pass DPR_FLAG_COHERENT_POINTS (the caller vouches for a sorted cloud)")
    ap.add_argument("--dist", default="gauss", choices=["gauss", "uniform", "tight"])
    ap.add_argument("--force-exchange", action="store_true",
                    help="--gpus 1 with a batched config: run the pose-sharded exchange step of the "
                         "N > 1 runs (double-buffered async all-reduce) over a one-rank RCCL group")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="--gpus 1 default run: skip the one-GPU point of the N > 1 job (C4, 512 poses)")
    ap.add_argument("--no-overlap-exchange", action="store_true",
                    help="N > 1: finish each step's all-reduce before the next step starts")
    ap.add_argument("--no-share-binning", action="store_true",
                    help="make the pullback redo the binning instead of reusing the forward's")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurement on Morton-sorted points")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # children first; this process never touches a GPU
    run_rank(args)


if __name__ == "__main__":
    main()
