#!/usr/bin/env python3
"""bench.py -- scan-to-map registrations/sec on MI355X (BASELINE.json metric).

Workload (config.workload): 64x1800-point synthetic scans registered against a
200-keyframe local map resident in HBM.  One "step" = one batch of B scans,
each with its own initial guess, run through the whole Gauss-Newton loop
(<= 30 iterations, every scan stops at its own convergence, MO:1848-1859).
Scans, map and hash grid are resident before the timed region; each step
uploads only the B initial poses and reads back the B results.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: the MAP is sharded across ranks in slabs of grid cells with a one-cell
halo (north_star / SURVEY 8e); every rank processes the scan points whose
transformed position falls into a cell it owns and the per-scan 6x6 JtJ / 6x1
Jtr / N_c are all-reduced (RCCL over xGMI) once per Gauss-Newton iteration.
The batch grows with N (B = batch x N scans) => weak scaling.
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_POINT_ITER = 72  # 12 B scan xyz + 5 x 12 B winning neighbours


def rot_angle(pa, pb, synth):
    Ra = synth.rpy_matrix(*[float(v) for v in pa[:3]])
    Rb = synth.rpy_matrix(*[float(v) for v in pb[:3]])
    c = (np.trace(Ra @ Rb.T) - 1.0) / 2.0
    return math.acos(max(-1.0, min(1.0, c)))


def rmse_pair(poses_a, poses_b, synth):
    dt = np.linalg.norm(np.asarray(poses_a)[:, 3:6] - np.asarray(poses_b)[:, 3:6], axis=1)
    dr = np.array([rot_angle(a, b, synth) for a, b in zip(poses_a, poses_b)])
    return float(np.sqrt((dt ** 2).mean())), float(np.sqrt((dr ** 2).mean()))


def host_cores():
    """CPU threads this process may really use: the cgroup quota when there is one
    (a 1-GPU box gets a 16-core share of a 256-thread host), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(scans, map_xyz, poses0, budget_s, log):
    """CPU restatement of the reference loop (oracle, own kd-tree incl. the per-scan
    tree build MO:1846, OpenMP over scan points MO:1622) on a bounded sample."""
    from oracle.oracle import Oracle, build
    import tempfile
    cores = host_cores()
    try:
        so = build(fast=True, out_dir=tempfile.mkdtemp(prefix="lio_oracle_"))
        kind_flags = "-O3 -march=native"
    except Exception as e:  # no gcc on the box: use the prebuilt portable checker
        log(f"cpu_baseline: native build failed ({e}); using prebuilt -O2 library")
        so = os.path.join(ROOT, "oracle", "liblio_oracle.so")
        kind_flags = "-O2"
    orc = Oracle(so)
    cfg = orc.default_config(knn_mode=1, n_threads=cores)
    done, t_used, poses, iters = 0, 0.0, [], []
    # the batch's scans in order, wrapping around until the time budget is used (poses are kept for the first pass)
    while done < 8 * len(scans) and (done < 2 or t_used < budget_s):
        i = done % len(scans)
        t0 = time.perf_counter()
        p, res, _, _ = orc.scan2map(cfg, scans[i], map_xyz, poses0[i])
        t_used += time.perf_counter() - t0
        if done < len(scans):
            poses.append(p)
            iters.append(res.iters)
        done += 1
    return {
        "value": done / t_used, "unit": "registrations/s", "cores": cores, "kind": "port",
        "sample": f"{done} registrations over the batch's {len(scans)} scans, {t_used:.1f} s; oracle/lio_oracle.c ({kind_flags}, "
                  f"-ffp-contract=off), own kd-tree rebuilt per scan, OpenMP {cores} threads",
        "ms_per_registration": 1e3 * t_used / done,
    }, np.array(poses), iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (one step = the whole GN loop for the batch, ~2 ms: the default keeps the timed window at ~0.2 s so that one host hiccup cannot dominate it)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="scans per GPU per step")
    ap.add_argument("--sensor", default="hdl64")
    ap.add_argument("--keyframes", type=int, default=200)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--variant", type=int, default=1, help="scan points per thread (1, 2, 4)")
    ap.add_argument("--latency", action="store_true", help="also time single-scan registrations")
    ap.add_argument("--case-cache", default="", help="npz path: load the synthetic case if present, else generate and save")
    ap.add_argument("--lds", type=int, default=0)
    ap.add_argument("--sort", type=int, default=1)
    ap.add_argument("--celldiv", type=int, default=2)
    ap.add_argument("--xcd", type=int, default=1)
    ap.add_argument("--sortbatch", type=int, default=1)
    ap.add_argument("--tile", type=float, default=0.0)
    ap.add_argument("--lookahead", type=int, default=0)
    ap.add_argument("--graph", type=int, default=12, help="replay a hipGraph of this many GN iterations per launch unit (0 = eager launches)")
    ap.add_argument("--shard", choices=["map", "scan"], default="scan",
                    help="N>1: map = slabs of the map + halo, owner-computes (north_star); scan = map replicated, workgroups of every scan dealt round-robin")
    ap.add_argument("--nncache", type=int, default=1, help="1 = bound each point's search by its previous neighbours (exact)")
    ap.add_argument("--maxsq", type=float, default=1.0, help="DIAGNOSTIC: squared 5-NN gate (reference: 1.0); smaller values shrink the "
                    "searched neighbourhood and change the results -- only for timing what-if runs")
    ap.add_argument("--pipeline", type=int, default=0, help="0 = auto, 1 = fused k_s2m_iterate, 2 = split cert/scan/fit")
    ap.add_argument("--lawnmower", action="store_true", help="keyframes on a lawn-mower path inside a 50 m radius (configs[3])")
    args = ap.parse_args()

    # Exactly one line goes to stdout: native libraries print there too (RCCL writes a version banner at
    # communicator creation), so file descriptor 1 is pointed at stderr for the whole run and the JSON line
    # is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    # BENCH_FORCE_SHARDED=1 drives the N>1 code path (process group, broadcasts, ShardedRunner, all-reduce
    # per GN iteration) with a single rank: a rehearsal of the RCCL calls on a one-GPU box, not a bench mode
    sharded = world > 1 or os.environ.get("BENCH_FORCE_SHARDED") == "1"
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL over xGMI; BENCH_BACKEND=gloo only exists to rehearse the N>1 path with ranks sharing one GPU
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    pkg = importlib.import_module("lio-slam_amd")
    synth = importlib.import_module("lio-slam_amd.synth")
    multi = importlib.import_module("lio-slam_amd.multigpu") if sharded else None

    # ---------------------------------------------------------------- data
    B = args.batch * world
    t0 = time.time()
    keyframes = []          # (cloud in lidar frame, pose) of every map keyframe, when generated here

    def generate():
        if args.case_cache and os.path.exists(args.case_cache):
            z = np.load(args.case_cache)
            offs = np.concatenate([[0], np.cumsum(z["lens"])])
            sc = [np.ascontiguousarray(z["scans"][offs[i]:offs[i + 1]]) for i in range(len(z["lens"]))]
            assert len(sc) == B, "case cache was generated for another batch size"
            return z["map"], sc, z["poses0"], z["poses_true"]
        case = synth.make_case(args.sensor, n_keyframes=args.keyframes, seed=synth.BASE_SEED, n_queries=B,
                               device=f"cuda:{local_rank}", lawnmower=args.lawnmower,
                               progress=lambda k, n: log(f"map keyframe {k}/{n}"))
        keyframes.extend(case["keyframes"])
        sc = [q["scan"] for q in case["queries"]]
        p0 = np.stack([q["pose_init"] for q in case["queries"]]).astype(np.float32)
        pt = np.stack([q["pose_true"] for q in case["queries"]]).astype(np.float32)
        if args.case_cache:
            np.savez(args.case_cache, map=case["map"], poses0=p0, poses_true=pt,
                     scans=np.concatenate(sc), lens=np.array([len(x) for x in sc]))
        return case["map"], sc, p0, pt

    if not sharded:
        map_xyz, scans, poses0, poses_true = generate()
    else:
        # Every rank ray-casts its own share of the scans on its own GPU (a query is the same whichever rank
        # makes it); rank 0 also builds the map.  Then everyone receives the SAME bytes (the ranks must agree
        # on the map shards and on every convergence decision, or their collectives would not match).
        def bcast(arr, shape, dtype, src=0):
            t = torch.from_numpy(np.ascontiguousarray(arr)).cuda() if rank == src else torch.zeros(shape, dtype=dtype, device="cuda")
            dist.broadcast(t, src)
            return t.cpu().numpy()

        per = args.batch
        if args.case_cache and os.path.exists(args.case_cache):
            if rank == 0:
                map_xyz, scans_all, poses0, poses_true = generate()
            share = None
        else:
            case = synth.make_case(args.sensor, n_keyframes=args.keyframes, seed=synth.BASE_SEED, n_queries=B,
                                   device=f"cuda:{local_rank}", lawnmower=args.lawnmower, q_range=(rank * per, (rank + 1) * per),
                                   with_map=(rank == 0), progress=lambda k, n: log(f"map keyframe {k}/{n}"))
            if rank == 0:
                map_xyz = case["map"]
                keyframes.extend(case["keyframes"])
            share = [q for q in case["queries"] if q is not None]
        hdr = torch.tensor([len(map_xyz) if rank == 0 else 0], dtype=torch.int64, device="cuda")
        dist.broadcast(hdr, 0)
        map_xyz = bcast(map_xyz if rank == 0 else None, (int(hdr[0]), 3), torch.float32)
        if share is None:                                    # case cache: rank 0 holds everything
            lens = bcast(np.array([len(x) for x in scans_all], np.int64) if rank == 0 else None, (B,), torch.int64)
            cat = bcast(np.concatenate(scans_all) if rank == 0 else None, (int(lens.sum()), 3), torch.float32)
            poses0 = bcast(poses0 if rank == 0 else None, (B, 6), torch.float32)
            poses_true = bcast(poses_true if rank == 0 else None, (B, 6), torch.float32)
        else:
            lens_l, cats, p0s, pts = [], [], [], []
            for r in range(world):                           # one broadcast round per rank's share
                mine = rank == r
                ln = bcast(np.array([len(q["scan"]) for q in share], np.int64) if mine else None, (per,), torch.int64, src=r)
                cats.append(bcast(np.concatenate([q["scan"] for q in share]) if mine else None, (int(ln.sum()), 3), torch.float32, src=r))
                p0s.append(bcast(np.stack([q["pose_init"] for q in share]).astype(np.float32) if mine else None, (per, 6), torch.float32, src=r))
                pts.append(bcast(np.stack([q["pose_true"] for q in share]).astype(np.float32) if mine else None, (per, 6), torch.float32, src=r))
                lens_l.append(ln)
            lens, cat = np.concatenate(lens_l), np.concatenate(cats)
            poses0, poses_true = np.concatenate(p0s), np.concatenate(pts)
        offs = np.concatenate([[0], np.cumsum(lens)])
        scans = [np.ascontiguousarray(cat[offs[i]:offs[i + 1]]) for i in range(B)]
    n_s = np.array([len(s) for s in scans])
    log(f"data: N_m={len(map_xyz)} N_s mean={n_s.mean():.0f} min={n_s.min()} max={n_s.max()} "
        f"B={B} gen {time.time() - t0:.1f}s")

    # -------------------------------------------------------------- engine
    kcfg = dict(device_id=local_rank, profile=1, use_graph=1 if args.graph else 0, graph_iters=max(args.graph, 1), lookahead=args.lookahead, kernel_variant=args.variant,
                use_lds=args.lds, sort_scan=args.sort, cell_div=args.celldiv, xcd_remap=args.xcd, tile_size=args.tile, sort_batch=args.sortbatch, nn_cache=args.nncache, max_sq_dist=args.maxsq, pipeline=args.pipeline)
    if sharded:
        # the host runs `lookahead` GN iterations ahead of the convergence check (never 0 here: polling the
        # iteration just enqueued would drain the GPU once per iteration and sub-batch)
        kcfg_sh = dict(kcfg)
        kcfg_sh.pop("lookahead")
        runner = multi.ShardedRunner(pkg, map_xyz, rank, world, dist, torch, mode=args.shard, groups=2,
                                     lookahead=max(args.lookahead, int(os.environ.get("BENCH_SHARD_LAG", "2"))), **kcfg_sh)
        runner.upload(scans)
        s2m = runner.handles[0]
    else:
        runner = None
        s2m = pkg.ScanToMap(**kcfg)
        s2m.set_map(map_xyz)
        s2m.batch_upload(scans)
    prof0 = s2m.profile()

    def step():
        if runner:
            runner.set_poses(poses0)
            runner.run()
            return runner.results(with_results=False)[0]
        s2m.batch_set_poses(poses0)
        s2m.batch_run()
        return s2m.batch_results(with_results=False)[0]

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        poses = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1 only: the same registrations with NO collective -- every rank registers its own share of the batch
    # against the replicated map through the single-GPU path (hipGraph loop).  Registrations are independent
    # objects, so this is the natural partition of the batch (BASELINE.json configs[4]); it is reported next to
    # `value`, which stays the north_star's form (one registration's work spread over the ranks, all-reduce per
    # GN iteration).
    replica = None
    if runner and world > 1:
        per = args.batch
        mine = slice(rank * per, (rank + 1) * per)
        rep = pkg.ScanToMap(**kcfg)
        rep.set_map(map_xyz)
        rep.batch_upload(scans[mine])

        def rep_step():
            rep.batch_set_poses(poses0[mine])
            rep.batch_run()
            return rep.batch_results(with_results=False)[0]

        for _ in range(args.warmup):
            rep_step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rep_poses = rep_step()
        sync_all()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        replica = {"value": B * args.steps / float(t.item()), "unit": "registrations/s",
                   "note": "whole scans dealt to the ranks, map replicated, no collective (independent registrations)"}
        rep.close()

    # per-launch accounting from the last timed step
    if runner:
        poses, results = runner.results(with_results=True)
        iters = np.array([r.iters for r in results])
        lms, pts_per_launch = [], []
        for (a, b_), hh in zip(runner.split, runner.handles):      # every sub-batch has its own launches
            pr = hh.profile()
            for i in range(pr.n_launches):
                lms.append(pr.launch_ms[i])
                # this rank handles 1/world of the points of the scans still iterating
                pts_per_launch.append(float(n_s[a:b_][iters[a:b_] > i].sum()) / world)
        lms, pts_per_launch = np.array(lms, dtype=np.float64), np.array(pts_per_launch, dtype=np.float64)
        n_launch = len(lms)
    else:
        poses, results = s2m.batch_results(with_results=True)
        prof = s2m.profile()
        iters = np.array([r.iters for r in results])
        # a unit = one launch (eager) or one replay of a captured chunk of unit_iters launches; the
        # average launch duration and the average algorithmic bytes are both taken over ALL launches
        # of k_s2m_iterate, like the rocprofv3 kernel-trace average
        ui = max(prof.unit_iters, 1)
        n_launch = prof.n_units * ui
        unit_ms = np.array(prof.launch_ms[:prof.n_units], dtype=np.float64)
        lms = np.repeat(unit_ms / ui, ui)
        pts_per_launch = np.array([int(n_s[iters > i].sum()) for i in range(n_launch)], dtype=np.float64)
    live = lms > 0
    bytes_per_launch = BYTES_PER_POINT_ITER * pts_per_launch[live].mean() if live.any() else 0.0
    ms_per_launch = float(lms[live].mean()) if live.any() else float("nan")
    achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9 if live.any() else 0.0

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tpath) and not sharded:
        tj = json.load(open(tpath))
        if (tj.get("scans_per_step"), tj.get("N_m")) == (B, int(len(map_xyz))):
            traffic = tj["hbm_bytes_per_launch"]      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, see DESIGN.md

    value = B * args.steps / elapsed
    out = {
        "metric": "scan-to-map registrations/sec, 64x1800 scan vs 200-keyframe map; pose RMSE",
        "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{args.sensor} {synth.SENSORS[args.sensor][0]}x{synth.SENSORS[args.sensor][1]} synthetic street-canyon "
                        f"scans vs {args.keyframes}-keyframe map (BASELINE.json headline = hdl64 64x1800 vs 200; "
                        f"configs[4] batched form)",
            "scans_per_step": B, "N_s_mean": float(n_s.mean()), "N_m": int(len(map_xyz)),
            "gn_iters_mean": float(iters.mean()), "gn_iters_max": int(iters.max()),
            "parallelism": "single GPU" if not sharded else
            (f"map sharded x{world} (slabs + halo, owner-computes)" if args.shard == "map" else
             f"map replicated, scan workgroups dealt over {world} ranks") + " + RCCL all-reduce of JtJ/Jtr per GN iteration",
            "kernel": {"points_per_thread": int(args.variant), "lds_staging": int(args.lds),
                       "nn_cache": args.nncache, "tile_sorted_scans": int(args.sort), "cell_div": int(args.celldiv)},
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": "k_s2m_iterate", "ms_per_launch": ms_per_launch,
            "limiter": "not HBM: the per-lane candidate stream is bound by divergent 16-byte load instructions through the "
                       "texture addresser (~77 % busy) with VALU issue at ~60 %; see DESIGN.md section 6",
            "launches_per_step": int(live.sum()), "launches_per_graph_replay": int(args.graph), "algorithmic_bytes_per_launch": bytes_per_launch,
            "launch_ms": [round(float(v), 4) for v in lms], "launch_points": [int(v) for v in pts_per_launch],
        },
        "gn_pipeline": None if runner else {"kind": "split (k_s2m_cert + k_s2m_scan + k_s2m_fit per iteration)" if prof.pipeline == 2 else "fused (k_s2m_iterate)",
                                            "points_per_iteration": [int(v) for v in prof.cert_points[:n_launch]],
                                            "points_scanned_per_iteration": [int(v) for v in prof.scan_points[:n_launch]]},
        "map_build_ms": prof0.map_build_ms, "map_upload_ms": prof0.map_upload_ms,
        "grid_cells": int(prof0.n_cells),
    }

    if rank == 0:
        if replica:
            out["batch_sharded_no_collective"] = replica
        rt, rr = rmse_pair(poses, poses_true, synth)
        out["pose_rmse_vs_truth"] = {"trans_m": rt, "rot_rad": rr}
        if not args.no_cpu and world == 1:      # the CPU baseline is timed at N = 1 only (the other ranks would idle)
            cb, cpu_poses, cpu_iters = cpu_baseline(scans, map_xyz, poses0, args.cpu_seconds, log)
            out["cpu_baseline"] = cb
            k = len(cpu_poses)
            rt, rr = rmse_pair(poses[:k], cpu_poses, synth)
            out["pose_rmse_vs_cpu"] = {"trans_m": rt, "rot_rad": rr, "scans": k,
                                       "bit_identical": int(sum(np.array_equal(a, b) for a, b in zip(poses[:k], cpu_poses))),
                                       "iters_equal": bool(list(iters[:k]) == list(cpu_iters))}
        if keyframes and not sharded:
            # feeder: extractCloud MO:1556-1588 on the GPU, host clouds in, map resident out
            kc = [np.concatenate([c, np.zeros((len(c), 1), np.float32)], 1) for c, _ in keyframes]
            kp = np.stack([p for _, p in keyframes])
            asm = pkg.ScanToMap(device_id=local_rank)
            store = pkg.KeyframeStore(device_id=local_rank)
            t0 = time.perf_counter()
            ids = [store.add(c) for c in kc]                       # once per keyframe, MO:2138-2142
            t_add = time.perf_counter() - t0
            store.assemble(ids, kp, 0.5, s2m=asm, want_output=False)
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):                                  # once per scan, MO:1556-1588 + MO:1846
                _, n_asm, _ = store.assemble(ids, kp, 0.5, s2m=asm, want_output=False)
            out["map_assembly"] = {"ms_per_scan_resident_keyframes_incl_grid_build": 1e3 * (time.perf_counter() - t0) / reps,
                                   "ms_keyframe_upload_total": 1e3 * t_add,
                                   "keyframes": len(kc), "points_in": int(sum(len(c) for c in kc)), "points_out": int(n_asm)}
            store.close()
            asm.close()
        if keyframes and not sharded:
            # the step after the deskew: FeatureExtraction::laserCloudInfoHandler FE:67-77 for one
            # organised 64x1800 sweep, host arrays in, corner/surface clouds out
            boxes = synth.make_scene(synth.BASE_SEED, length=max(60.0, float(args.keyframes) + 20.0))
            org = synth.organize_scan(synth.cast_scan(boxes, poses_true[0], args.sensor, seed=77, device=f"cuda:{local_rank}"))
            fe_args = (org["cloud"], org["start_ring"], org["end_ring"], org["col"], org["range"])
            fe = pkg.extract_features(*fe_args, device_id=local_rank)
            t0 = time.perf_counter()
            for _ in range(5):
                fe = pkg.extract_features(*fe_args, device_id=local_rank)
            out["feature_extraction"] = {"ms_per_sweep_incl_h2d_d2h": 1e3 * (time.perf_counter() - t0) / 5,
                                         "points": int(len(org["cloud"])), "corner": int(len(fe["corner"])),
                                         "surface": int(len(fe["surface"]))}
        if args.latency and not sharded:
            lat = pkg.ScanToMap(device_id=local_rank)
            lat.set_map(map_xyz)
            for i in range(3):
                lat.scan2MapOptimization(scans[i % B], poses0[i % B])
            t0 = time.perf_counter()
            n_lat = min(B, 32)
            for i in range(n_lat):
                lat.scan2MapOptimization(scans[i], poses0[i])
            out["single_scan_ms_incl_h2d"] = 1e3 * (time.perf_counter() - t0) / n_lat
            lat.close()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if runner:
        runner.close()
    else:
        s2m.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
