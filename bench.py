#!/usr/bin/env python3
"""bench.py -- scan-to-map registrations/sec on MI355X (BASELINE.json metric).

Workload (config.workload): 64x1800-point synthetic scans registered against a 200-keyframe local map
resident in HBM.  One "step" = one batch of B scans taken through the WHOLE device-side path of the
boundary: the raw pcl::PointXYZI records of the batch (resident in HBM when the timed region starts) are
staged, tile-sorted into the SoA layout, every scan gets its own initial guess and runs the Gauss-Newton
loop (<= 30 iterations, each scan stops at its own convergence, MO:1848-1859), and the B results are read
back.  Every step registers a DIFFERENT batch (`--batches` distinct batches are cycled); `--handles` (3)
handles sharing the resident map form a software pipeline: batches k+1 and k+2 are staged, sorted and STARTED
while batch k still iterates, and only then are the results of batch k collected (two handles leave the GPU
alone with the tail of one batch while the host prepares the next: 357 k against 416 k registrations/s).  The roofline of the dominant kernel is measured
in the same run by a separate pass in which the kernel has the GPU to itself (`--roofline-pass-only` makes
that pass the only timed region, for rocprofv3).  Also reported, never as `value`: the same stream with the records coming from pinned HOST memory
over PCIe (`streamed_h2d`), the GN loop alone on a pre-sorted resident batch (`gn_loop_only`, round 1's
definition), and the single-scan sequence a patched node issues per callback (`single_scan_node_path_ms`).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python bench.py --gpus N ...            (starts its own N ranks: a child `python -m torch.distributed.run`, before any GPU call)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (the same, started by the caller)
  python bench.py --gpus N --inlib ...    (ONE process, cfg.n_devices = N: the in-library multi-GPU mode a patched node uses)

N > 1 (north_star / SURVEY 8e): the MAP is sharded across ranks in slabs of grid cells with a one-cell halo;
every rank processes the scan points whose transformed position falls into a cell it owns and the per-scan
6x6 JtJ / 6x1 Jtr / N_c are all-reduced (RCCL over xGMI) once per Gauss-Newton iteration.  The batch grows
with N (B = batch x N scans per step) => weak scaling.
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
PARITY_BASIS = ("CPU restatement of the reference source (oracle/lio_oracle.c); Eigen / OpenCV / FLANN / PCL arithmetic "
                "restated from their published algorithms; unpinned by any reference-held fixture (the reference has none "
                "and cannot be built here)")


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scans, map_xyz, poses0, budget_s, log):
    """CPU restatement of the reference loop (oracle: own kd-tree rebuilt per scan as MO:1846 does, OpenMP over
    scan points as MO:1622) on a bounded sample of the batch, at the reference's own thread counts
    (numberOfCores: 4 in lio_sam_default.yaml:63, 12 in 6t.yaml:119) and at all cores of the box."""
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
    by_threads, poses, iters = {}, [], []
    counts = sorted({min(4, cores), min(12, cores), cores})
    for nt in counts:
        cfg = orc.default_config(knn_mode=1, n_threads=nt)
        done, t_used, t_build = 0, 0.0, 0.0
        budget = budget_s * (0.5 if nt == cores else 0.5 / max(len(counts) - 1, 1))
        # the batch's scans in order, wrapping around until the time budget is used
        while done < 8 * len(scans) and (done < 2 or t_used < budget):
            i = done % len(scans)
            t0 = time.perf_counter()
            p, res, _, _ = orc.scan2map(cfg, scans[i], map_xyz, poses0[i])
            t_used += time.perf_counter() - t0
            t_build += orc.last_kdtree_build_seconds()
            if nt == cores and done < len(scans):
                poses.append(p)
                iters.append(res.iters)
            done += 1
        by_threads[str(nt)] = {"registrations_per_s": done / t_used, "ms_per_registration": 1e3 * t_used / done,
                               "ms_kdtree_build": 1e3 * t_build / done, "ms_gn_loop": 1e3 * (t_used - t_build) / done,
                               "registrations": done, "seconds": t_used}
        log(f"cpu_baseline {nt} threads: {done / t_used:.1f} reg/s ({done} registrations, kd-tree build {1e3 * t_build / done:.2f} ms each)")
    top = by_threads[str(cores)]
    return {
        "value": top["registrations_per_s"], "unit": "registrations/s", "cores": cores, "kind": "port",
        "sample": f"{top['registrations']} registrations over the first scans of the workload, {top['seconds']:.1f} s; "
                  f"oracle/lio_oracle.c ({kind_flags}, -ffp-contract=off), own kd-tree rebuilt per scan (MO:1846, "
                  f"{top['ms_kdtree_build']:.2f} ms of the {top['ms_per_registration']:.2f} ms), OpenMP {cores} threads on a {cpu_model()}",
        "ms_per_registration": top["ms_per_registration"], "by_threads": by_threads, "cpu_model": cpu_model(),
        "label": "CPU restatement of the reference algorithm (own kd-tree), not the reference binary",
    }, np.array(poses), iters


def to_records(scans, stride):
    """Back-to-back records of `stride` bytes, x,y,z at byte 0,4,8 (stride 32 = pcl::PointXYZI, UT:65)."""
    n = sum(len(s) for s in scans)
    rec = np.zeros((n, stride // 4), np.float32)
    rec[:, :3] = np.concatenate(scans)
    if stride >= 32:
        rec[:, 3] = 1.0
    return rec


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher (the form the driver records): start the N ranks as a child
    `python -m torch.distributed.run` BEFORE this process touches the GPU (nothing is re-exec'ed: the parent never
    initialises HIP), relay rank 0's single JSON line and the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for raw in child.stdout:
        txt = raw.decode(errors="replace").strip()
        if txt.startswith("{") and '"metric"' in txt:
            line = txt
        elif txt:
            print(txt, file=sys.stderr, flush=True)
    rc = child.wait()
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    raise SystemExit(rc if rc else (0 if line is not None else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64, help="timed steps (one step = one batch through staging, tile sort, the whole GN loop and the result read-back)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=512, help="scans per GPU per step")
    ap.add_argument("--batches", type=int, default=0, help="distinct batches cycled through (0 = 8 on one GPU, 2 on several)")
    ap.add_argument("--stride", type=int, default=32, help="bytes per input record (32 = pcl::PointXYZI, 16 = xyzi floats, 12 = packed xyz)")
    ap.add_argument("--sensor", default="hdl64")
    ap.add_argument("--keyframes", type=int, default=200)
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (streamed H2D, GN loop only, node path, feeders)")
    ap.add_argument("--variant", type=int, default=1, help="scan points per thread (1, 2, 4)")
    ap.add_argument("--case-cache", default="", help="npz path: load the synthetic case if present, else generate and save")
    ap.add_argument("--lds", type=int, default=0)
    ap.add_argument("--sort", type=int, default=1)
    ap.add_argument("--celldiv", type=int, default=2)
    ap.add_argument("--xcd", type=int, default=1)
    ap.add_argument("--sortbatch", type=int, default=1)
    ap.add_argument("--tile", type=float, default=0.0)
    ap.add_argument("--lookahead", type=int, default=0)
    ap.add_argument("--graph", type=int, default=12, help="replay a hipGraph of this many GN iterations per launch unit (0 = eager launches)")
    ap.add_argument("--shard", choices=["map", "scan"], default="map",
                    help="N>1: map = slabs of the map + halo, owner-computes (north_star); scan = map replicated, workgroups of every scan dealt round-robin")
    ap.add_argument("--nncache", type=int, default=1, help="1 = bound each point's search by its previous neighbours (exact)")
    ap.add_argument("--pipeline", type=int, default=0, help="cfg.pipeline: 0 = auto, 1 = one launch per GN iteration, 4 = one-launch loop where it fits")
    ap.add_argument("--roofline-pass-only", action="store_true", help="profiling: make the roofline pass (GN loop alone on one "
                    "pre-sorted resident batch, launches of k_s2m_iterate never overlapping anything) the ONLY timed region")
    ap.add_argument("--single-buffer", action="store_true", help="A/B: one handle, no overlap of staging and GN loop")
    ap.add_argument("--handles", type=int, default=3, help="handles (batches in flight) of the streamed pipeline; measured on one MI355X: "
                    "2 -> 357 k, 3 -> 416 k, 4 -> 375 k, 5 -> 382 k, 6 -> 357 k registrations/s")
    ap.add_argument("--maxsq", type=float, default=1.0, help="DIAGNOSTIC: squared 5-NN gate (reference: 1.0); smaller values shrink the "
                    "searched neighbourhood and change the results -- only for timing what-if runs")
    ap.add_argument("--lawnmower", action="store_true", help="keyframes on a lawn-mower path inside a 50 m radius (configs[3])")
    ap.add_argument("--inlib", action="store_true", help="N > 1 in ONE process: cfg.n_devices = N (in-library multi-GPU mode, device-side "
                    "exchange of the per-scan sums); with fewer than N GPUs visible the same device is listed N times (emulation)")
    ap.add_argument("--leaf-scan", type=float, default=0.4, help="mappingSurfLeafSize (0.4 lio_sam_default.yaml:56; 0.2 jeep.yaml:99; 0.15 "
                    "lio_sam_livox.yaml:56; <= 0 = PCL pass-through as with 0.01 in 6t.yaml:112: the scan is NOT downsampled)")
    ap.add_argument("--leaf-map", type=float, default=0.5, help="surroundingKeyframeMapLeafSize (0.5 lio_sam_default.yaml:71; 0.3 lio_sam_livox.yaml:71)")
    ap.add_argument("--density", type=float, default=0.02, help="box obstacles per m^2 of the synthetic street (SURVEY 8d: 0.02)")
    ap.add_argument("--label", default="", help="free text copied into config.label (parameter-set runs)")
    args = ap.parse_args()
    if args.gpus > 1 and not args.inlib and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                         # (does not return)

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
    inlib = bool(args.inlib and args.gpus > 1)
    if world != (1 if inlib else args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = args.gpus if inlib else 1                # devices driven by THIS process
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
    B = args.batch * world * n_dev               # scans per step, job-wide
    NB = args.batches if args.batches > 0 else (8 if not sharded else 2)
    NQ = B * NB
    t0 = time.time()
    keyframes = []          # (cloud in lidar frame, pose) of every map keyframe, when generated here
    leafs = dict(scan_leaf=args.leaf_scan, map_leaf=args.leaf_map, density=args.density)

    def generate():
        if args.case_cache and os.path.exists(args.case_cache):
            z = np.load(args.case_cache)
            lens = z["lens"]
            if len(lens) == NQ:
                offs = np.concatenate([[0], np.cumsum(lens)])
                cat = z["scans"]                     # (an NpzFile re-reads the member on every access: take it ONCE)
                sc = [np.ascontiguousarray(cat[offs[i]:offs[i + 1]]) for i in range(NQ)]
                del cat
                return z["map"], sc, z["poses0"], z["poses_true"]
            log("case cache was generated for another workload size: regenerating")
        case = synth.make_case(args.sensor, n_keyframes=args.keyframes, seed=synth.BASE_SEED, n_queries=NQ,
                               device=f"cuda:{local_rank}", lawnmower=args.lawnmower, workers=min(8, host_cores()),
                               progress=lambda k, n: log(f"map keyframe {k}/{n}"), **leafs)
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

        per = NQ // world
        case = synth.make_case(args.sensor, n_keyframes=args.keyframes, seed=synth.BASE_SEED, n_queries=NQ,
                               device=f"cuda:{local_rank}", lawnmower=args.lawnmower, q_range=(rank * per, (rank + 1) * per),
                               with_map=(rank == 0), workers=min(8, max(1, host_cores() // max(world, 1))),
                               progress=lambda k, n: log(f"map keyframe {k}/{n}"), **leafs)
        if rank == 0:
            map_xyz = case["map"]
            keyframes.extend(case["keyframes"])
        share = [q for q in case["queries"] if q is not None]
        hdr = torch.tensor([len(map_xyz) if rank == 0 else 0], dtype=torch.int64, device="cuda")
        dist.broadcast(hdr, 0)
        map_xyz = bcast(map_xyz if rank == 0 else None, (int(hdr[0]), 3), torch.float32)
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
        scans = [np.ascontiguousarray(cat[offs[i]:offs[i + 1]]) for i in range(NQ)]
    n_s = np.array([len(s) for s in scans])
    log(f"data: N_m={len(map_xyz)} N_s mean={n_s.mean():.0f} min={n_s.min()} max={n_s.max()} "
        f"{NB} batches x {B} scans, gen {time.time() - t0:.1f}s")

    # raw records of every batch, resident in HBM (what the boundary is handed: pcl::PointXYZI AoS)
    stride = args.stride
    batch_scans = [scans[b * B:(b + 1) * B] for b in range(NB)]
    batch_npts = [[len(s) for s in bs] for bs in batch_scans]
    batch_rec = [to_records(bs, stride) for bs in batch_scans]
    dev_rec = [torch.from_numpy(r).cuda() for r in batch_rec]
    torch.cuda.synchronize()

    # -------------------------------------------------------------- engine
    kcfg = dict(device_id=local_rank, profile=1, max_batch=args.batch, use_graph=1 if args.graph else 0, graph_iters=max(args.graph, 1), lookahead=args.lookahead, kernel_variant=args.variant,
                use_lds=args.lds, sort_scan=args.sort, cell_div=args.celldiv, xcd_remap=args.xcd, tile_size=args.tile, sort_batch=args.sortbatch, nn_cache=args.nncache,
                max_sq_dist=args.maxsq, pipeline=args.pipeline)

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    if sharded:
        # the host runs `lookahead` GN iterations ahead of the convergence check (never 0 here: polling the
        # iteration just enqueued would drain the GPU once per iteration and sub-batch)
        kcfg_sh = dict(kcfg)
        kcfg_sh.pop("lookahead")
        # slabs balanced by where the REGISTERED points will fall (a sample of batch 0 at its initial poses), not by where
        # the map points are: scan returns are dense around the sensor, the map is not
        load = np.concatenate([multi.transform_f32(synth.pose_matrix(poses0[i])[:3].astype(np.float32).reshape(12), scans[i][::16])
                               for i in range(0, B, 4)])
        runner = multi.ShardedRunner(pkg, map_xyz, rank, world, dist, torch, mode=args.shard, groups=2, load_xyz=load,
                                     lookahead=max(args.lookahead, int(os.environ.get("BENCH_SHARD_LAG", "2"))), **kcfg_sh)
        handles = [runner.handles[0]]
        prof0 = handles[0].profile()

        def run_stream(n_steps, sources, first_batch=0, keep=None):
            """One step = stage + tile-sort batch k (every rank holds every scan: ownership follows the pose), GN loop
            with one all-reduce per iteration, results."""
            out = None
            for k in range(n_steps):
                b = (first_batch + k) % NB
                runner.upload_raw(sources[b].data_ptr(), batch_npts[b], stride)
                runner.set_poses(poses0[b * B:(b + 1) * B])
                runner.run()
                out = runner.results(with_results=False)[0]
                if keep is not None and b == 0:
                    keep["poses"] = out
            return out
    else:
        runner = None
        if inlib:
            # ONE process, one handle, cfg.n_devices = N: the map is cut into slabs inside the library, the per-scan sums are
            # joined on the devices once per GN iteration (lio_multi.hip).  Fewer than N GPUs visible: the same ordinal is
            # listed N times -- every code path runs, only the peer traffic is local (an emulation, labelled as such).
            dev_ids = list(range(n_dev)) if torch.cuda.device_count() >= n_dev else [local_rank] * n_dev
            kc = dict(kcfg)
            kc.update(n_devices=n_dev, device_ids=dev_ids, use_graph=0, lookahead=-1)
            hA = pkg.ScanToMap(**kc)
        else:
            hA = pkg.ScanToMap(**kcfg)
        hA.set_map(map_xyz)
        prof0 = hA.profile()
        handles = [hA]
        if not args.single_buffer and not inlib:
            for _ in range(max(1, int(os.environ.get("BENCH_HANDLES", args.handles)) - 1)):     # (BENCH_HANDLES: A/B runs)
                hB = pkg.ScanToMap(**kcfg)
                hB.share_map(hA)
                handles.append(hB)

        def run_stream(n_steps, sources, first_batch=0, keep=None):
            """Software pipeline over the handles: while batch k iterates (asynchronous launch loop on its handle's stream),
            batch k+1 is staged, tile-sorted and STARTED on the other handle's stream -- its first, large launches overlap
            the last, nearly empty ones of batch k -- and only then are the results of batch k collected."""
            nh = len(handles)
            ptr = lambda b: sources[b].data_ptr() if hasattr(sources[b], "data_ptr") else sources[b].ptr

            def start(k):
                b = (first_batch + k) % NB
                h = handles[k % nh]
                h.batch_upload_raw(ptr(b), batch_npts[b], stride)
                h.batch_set_poses(poses0[b * B:(b + 1) * B])
                h.batch_run()

            out = None
            for j in range(min(nh - 1, n_steps)):         # nh - 1 batches in flight ahead of the one being collected
                start(j)
            for k in range(n_steps):
                b = (first_batch + k) % NB
                h = handles[k % nh]
                if nh == 1:
                    start(k)
                elif k + nh - 1 < n_steps:
                    start(k + nh - 1)
                out = h.batch_results(with_results=False)[0]
                if keep is not None:                      # launch accounting of EVERY timed step (HIP events on h's stream)
                    pr = h.profile()
                    keep.setdefault("unit_ms", []).extend(pr.launch_ms[:pr.n_units])
                    keep["launches"] = keep.get("launches", 0) + pr.n_units * max(pr.unit_iters, 1)
                    keep["point_iters"] = keep.get("point_iters", 0) + int(pr.point_iters)
                if keep is not None and b == 0:
                    keep["poses"] = out
            return out

    keep = {}
    if not (args.roofline_pass_only and not sharded):
        run_stream(args.warmup, dev_rec, keep=keep)
        sync_all()
        for k_ in ("unit_ms", "launches", "point_iters"):      # (accounting restarts with the timed region)
            keep.pop(k_, None)
        t0 = time.perf_counter()
        run_stream(args.steps, dev_rec, first_batch=args.warmup, keep=keep)
        sync_all()
        elapsed = time.perf_counter() - t0
        if dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
    streamed_acct = {k_: keep.get(k_) for k_ in ("unit_ms", "launches", "point_iters")}

    value = B * args.steps / elapsed if not (args.roofline_pass_only and not sharded) else 0.0
    last_b = 0 if (args.roofline_pass_only and not sharded) else (args.warmup + args.steps - 1) % NB   # the batch of the last timed step

    # per-launch accounting of the dominant kernel from the LAST timed step (HIP events on the handle's stream)
    ns_last = n_s[last_b * B:(last_b + 1) * B]
    if runner:
        poses_last, results = runner.results(with_results=True)
        iters = np.array([r.iters for r in results])
        lms, pts_per_launch = [], []
        for (a, b_), hh in zip(runner.split, runner.handles):      # every sub-batch has its own launches
            pr = hh.profile()
            for i in range(pr.n_launches):
                lms.append(pr.launch_ms[i])
                # this rank handles 1/world of the points of the scans still iterating
                pts_per_launch.append(float(ns_last[a:b_][iters[a:b_] > i].sum()) / world)
        lms, pts_per_launch = np.array(lms, dtype=np.float64), np.array(pts_per_launch, dtype=np.float64)
        prof = None
    else:
        h_last = handles[(args.steps - 1) % len(handles)]
        if args.roofline_pass_only:                   # (profiling run: the streamed region was skipped; one plain batch for the trace)
            h_last = handles[0]
            h_last.batch_upload(batch_scans[0]); h_last.batch_set_poses(poses0[:B]); h_last.batch_run()
        poses_last, results = h_last.batch_results(with_results=True)
        prof = h_last.profile()
        iters = np.array([r.iters for r in results])
        # a unit = one launch (eager) or one replay of a captured chunk of unit_iters launches; the
        # average launch duration and the average algorithmic bytes are both taken over ALL launches
        # of k_s2m_iterate, like the rocprofv3 kernel-trace average
        ui = max(prof.unit_iters, 1)
        n_launch = prof.n_units * ui
        unit_ms = np.array(prof.launch_ms[:prof.n_units], dtype=np.float64)
        lms = np.repeat(unit_ms / ui, ui)
        # (in-library multi-GPU: the launches timed are device 0's, which handles 1/n_dev of the live points)
        pts_per_launch = np.array([int(ns_last[iters > i].sum()) for i in range(n_launch)], dtype=np.float64) / n_dev
    # Roofline pass (single GPU): the dominant kernel ALONE on the GPU.  In the streamed region the launch loops of two
    # batches overlap on purpose (that is what hides the tails), so the duration of a launch there measures how the
    # GPU is shared, not the kernel.  Here one pre-sorted resident batch is re-registered from its initial poses with
    # nothing else in flight, and every launch of every step is bracketed by HIP events on the handle's stream.
    roof = None
    if not sharded and not inlib:
        g = handles[0]
        g.batch_upload(batch_scans[0])
        n_roof = args.steps if args.roofline_pass_only else 32

        def gn_step(acct=None):
            g.batch_set_poses(poses0[:B])
            g.batch_run()
            p_ = g.batch_results(with_results=False)[0]
            if acct is not None:
                pr = g.profile()
                acct["unit_ms"].extend(pr.launch_ms[:pr.n_units])
                acct["launches"] += pr.n_units * max(pr.unit_iters, 1)
                acct["point_iters"] += int(pr.point_iters)
            return p_

        for _ in range(max(3, args.warmup if args.roofline_pass_only else 3)):
            gn_step()
        sync_all()
        roof = {"unit_ms": [], "launches": 0, "point_iters": 0}
        t0 = time.perf_counter()
        for _ in range(n_roof):
            p_roof = gn_step(roof)
        sync_all()
        roof["elapsed"] = time.perf_counter() - t0
        roof["steps"] = n_roof
        if args.roofline_pass_only:
            elapsed = roof["elapsed"]
            value = B * args.steps / elapsed
            keep["poses"] = p_roof
    live = lms > 0
    in_stream = None
    if roof is not None and roof["launches"]:
        # single GPU: averages over ALL launches of ALL steps of the roofline pass (what rocprofv3's kernel statistics
        # average too when the same pass is traced: `--roofline-pass-only`)
        tot_ms = float(np.sum(roof["unit_ms"]))
        ms_per_launch = tot_ms / roof["launches"]
        bytes_per_launch = BYTES_PER_POINT_ITER * roof["point_iters"] / roof["launches"]
        achieved = BYTES_PER_POINT_ITER * roof["point_iters"] / (tot_ms * 1e-3) / 1e9
        if streamed_acct.get("launches"):
            in_stream = {"ms_per_launch_while_two_batches_overlap": float(np.sum(streamed_acct["unit_ms"])) / streamed_acct["launches"],
                         "launches": int(streamed_acct["launches"]),
                         "algorithmic_GBs_of_the_whole_pipeline": BYTES_PER_POINT_ITER * streamed_acct["point_iters"] / elapsed / 1e9,
                         "note": "72 B x every live point-iteration of the timed region / its wall time (staging and tile sort included)"}
    else:
        bytes_per_launch = BYTES_PER_POINT_ITER * pts_per_launch[live].mean() if live.any() else 0.0
        ms_per_launch = float(lms[live].mean()) if live.any() else float("nan")
        achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9 if live.any() else 0.0

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
    if os.path.exists(tpath) and not sharded and not inlib:
        tj = json.load(open(tpath))
        if (tj.get("scans_per_step"), tj.get("N_m")) == (B, int(len(map_xyz))):
            traffic = tj["hbm_bytes_per_launch"]      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, see DESIGN.md

    # The roofline of the resource that actually binds the kernel: vector-ALU issue, with the texture addresser next to it.
    # Counter-derived (SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE, GRBM_TA_BUSY need rocprofv3 --pmc passes), so it is a STORED figure
    # of the builder's profiling session for this exact workload, like `roofline.traffic`; see tools/issue_roofline.py.
    issue = None
    ipath = os.path.join(ROOT, "profiles", "r03_issue_roofline.json")
    if os.path.exists(ipath) and traffic is not None:
        issue = json.load(open(ipath))
        issue["source"] = ("stored figure: separate rocprofv3 --pmc passes of `bench.py --roofline-pass-only` by the builder (profiles/r03_pmc_SQ.txt, "
                           "r03_pmc_SQ2.txt, tools/issue_roofline.py), same workload as this run (scans_per_step, N_m checked); not a counter of this run")
        issue["note"] = (f"k_s2m_iterate issues a vector instruction in {100 * issue['frac']:.0f} % of all cycles of every SIMD while its divergent "
                         f"16-byte candidate gathers keep the texture addresser busy {100 * issue['texture_addresser_busy_frac']:.0f} % of the time: "
                         "it sits on both limiters (and on the dependent loads five waves per SIMD only partly hide), HBM is at 12 %")

    out = {
        "metric": "scan-to-map registrations/sec, 64x1800 scan vs 200-keyframe map; pose RMSE",
        "value": value, "unit": "registrations/s", "n_gpus": world * n_dev, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "parity_basis": PARITY_BASIS,
        "config": {
            "workload": f"{args.sensor} {synth.SENSORS[args.sensor][0]}x{synth.SENSORS[args.sensor][1]} synthetic street-canyon "
                        f"scans vs {args.keyframes}-keyframe map (BASELINE.json headline = hdl64 64x1800 vs 200; "
                        f"configs[4] batched form), {NB} distinct batches streamed",
            "mappingSurfLeafSize": args.leaf_scan, "surroundingKeyframeMapLeafSize": args.leaf_map, "obstacle_density": args.density,
            "label": args.label,
            "timed_region": "ROOFLINE PASS ONLY (profiling run): GN loop of one pre-sorted resident batch" if (args.roofline_pass_only and not sharded) else
                            "per step, for a batch not seen in the previous step: staging of the raw records (resident in HBM) + AoS->SoA + "
                            "tile sort, B initial poses in, whole GN loop, B results out" + ("" if sharded else
                            f"; software-pipelined over {len(handles)} handles sharing the map ({len(handles) - 1} batches in flight ahead of the one being collected)" if len(handles) > 1 else "; single handle"),
            "inputs_resident_in_hbm": True, "h2d_in_timed_region": False, "input_record_bytes": stride,
            "scans_per_step": B, "distinct_batches": NB, "N_s_mean": float(n_s.mean()), "N_m": int(len(map_xyz)),
            "gn_iters_mean": float(iters.mean()), "gn_iters_max": int(iters.max()),
            "parallelism": (f"ONE process, cfg.n_devices = {n_dev} (device_ids {dev_ids}): map sharded inside the library (slabs + 16-cell halo, "
                            "owner-computes), per-scan JtJ/Jtr joined on the devices once per GN iteration (peer stores + event waits)"
                            + ("" if len(set(dev_ids)) == n_dev else "; EMULATED: fewer GPUs visible than listed, the same device repeats")) if inlib else
            "single GPU" if not sharded else
            (f"map sharded x{world} (slabs + halo, owner-computes)" if args.shard == "map" else
             f"map replicated, scan workgroups dealt over {world} ranks") + " + RCCL all-reduce of JtJ/Jtr per GN iteration",
            "kernel": {"points_per_thread": int(args.variant), "lds_staging": int(args.lds), "pipeline": "fused",
                       "nn_cache": args.nncache, "tile_sorted_scans": int(args.sort), "cell_div": int(args.celldiv)},
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": None if traffic is None else "stored figure: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command run by the "
                              "builder (profiles/r03_traffic.json, profiles/README.md), not a counter of this run",
            "kernel": "k_s2m_iterate",
            "ms_per_launch": ms_per_launch,
            "limiter": "not HBM: VALU issue (about 2100 vector instructions per 64 live points and iteration; the bit-exact plane fit is "
                       "~40 % of them) with the divergent candidate gathers keeping the texture addresser ~78 % busy, over a chain of "
                       "dependent loads per workgroup; see `roofline_issue` and DESIGN.md section 6",
            "launches_per_step": int(live.sum()), "launches_per_graph_replay": int(args.graph), "algorithmic_bytes_per_launch": bytes_per_launch,
            "launches_measured": int(roof["launches"]) if roof else int(live.sum()),
            "in_streamed_region": in_stream,
            "last_step_launch_ms": [round(float(v), 4) for v in lms], "last_step_launch_points": [int(v) for v in pts_per_launch],
            "measured": "roofline pass of this run: one pre-sorted resident batch re-registered with nothing else in flight, HIP events "
                        "around every graph replay of every step on the handle's stream; algorithmic bytes = 72 B x live point-iterations "
                        "of the same steps.  (In the streamed region two batches' launch loops overlap on purpose, see in_streamed_region.)"
                        if not (sharded or inlib) else "HIP events around the launches of the last timed step" + (" on device 0" if inlib else ""),
        },
        "roofline_issue": issue,
        "map_build_ms": prof0.map_build_ms, "map_upload_ms": prof0.map_upload_ms,
        "map_rows": {"x_sub": prof0.map_x_sub, "tight_tables": prof0.map_tight_tables, "first_try": prof0.map_first_try, "points_per_occupied_cell": round(prof0.map_pts_per_cell, 2)},
        "grid_cells": int(prof0.n_cells),
    }

    if inlib:
        pm = handles[0].profile()
        out["inlib_exchange"] = {"form": {0: "peer stores from a kernel", 1: "hipMemcpyPeerAsync", 2: "host memory (round 2)"}.get(int(pm.multi_exchange)),
                                 "gn_iterations_enqueued_last_step": int(pm.multi_iterations), "stream_syncs_in_the_loop_last_step": int(pm.multi_stream_syncs),
                                 "convergence_event_waits_last_step": int(pm.multi_event_waits)}
    extras = not args.no_extras and not inlib
    # -------------------------------------------------- secondary measurements (never `value`)
    if extras and not sharded:
        # (1) the same stream with the records in pinned HOST memory: per-scan H2D over PCIe inside the timed region
        pins = []
        for r in batch_rec:
            p = pkg.PinnedBuffer(r.nbytes)
            p.array[:] = r.view(np.uint8).reshape(-1)
            pins.append(p)
        n_st = max(8, args.steps // 2)
        run_stream(4, pins)
        sync_all()
        t0 = time.perf_counter()
        run_stream(n_st, pins, first_batch=4)
        sync_all()
        el = time.perf_counter() - t0
        out["streamed_h2d"] = {"value": B * n_st / el, "unit": "registrations/s", "ms_per_step": 1e3 * el / n_st, "steps": n_st,
                               "h2d_in_timed_region": True, "h2d_bytes_per_step": int(batch_rec[0].nbytes),
                               "pcie_gbs_sustained": batch_rec[0].nbytes * n_st / el / 1e9,
                               "note": "the metric as SURVEY 8(d) words it: scans arrive from (pinned) host memory as "
                                       f"{stride}-byte records; upload of batch k+1 overlaps the GN loop of batch k"}
        for p in pins:
            p.close()
    if roof is not None:
        out["gn_loop_only"] = {"value": B * roof["steps"] / roof["elapsed"], "unit": "registrations/s", "ms_per_step": 1e3 * roof["elapsed"] / roof["steps"],
                               "note": "the roofline pass: inputs uploaded and tile-sorted once outside the timer (round 1's `value`)"}

    # N > 1 only: the same registrations with NO collective -- every rank registers its own share of the batch
    # against the replicated map through the single-GPU path.  Registrations are independent objects, so this is
    # the natural partition of the batch (BASELINE.json configs[4]); it is reported next to `value`, which stays
    # the north_star's form (one registration's work spread over the ranks, all-reduce per GN iteration).
    if runner and extras and args.shard == "map":
        # the other partition SURVEY 8(e) allows: map replicated, every rank takes one N-th of every scan's workgroups
        # (no halo, no ownership tests, balanced); same timed region, same collective
        r2 = multi.ShardedRunner(pkg, map_xyz, rank, world, dist, torch, mode="scan", groups=2,
                                 lookahead=max(args.lookahead, int(os.environ.get("BENCH_SHARD_LAG", "2"))), **kcfg_sh)

        def scan_stream(n_steps):
            for k in range(n_steps):
                b = k % NB
                r2.upload_raw(dev_rec[b].data_ptr(), batch_npts[b], stride)
                r2.set_poses(poses0[b * B:(b + 1) * B])
                r2.run()
                r2.results(with_results=False)

        scan_stream(max(2, args.warmup // 2))
        sync_all()
        t0 = time.perf_counter()
        n_st = max(4, args.steps // 2)
        scan_stream(n_st)
        sync_all()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["scan_partition"] = {"value": B * n_st / float(t.item()), "unit": "registrations/s", "steps": n_st,
                                 "note": "map replicated, scan workgroups dealt over the ranks, all-reduce of JtJ/Jtr per GN iteration"}
        r2.close()
    if runner and world > 1 and extras:
        per = args.batch
        rep = pkg.ScanToMap(**kcfg)
        rep.set_map(map_xyz)
        rep2 = pkg.ScanToMap(**kcfg)
        rep2.share_map(rep)
        hs = [rep, rep2]
        my = [[scans[b * B + rank * per + i] for i in range(per)] for b in range(NB)]
        my_rec = [torch.from_numpy(to_records(m, stride)).cuda() for m in my]
        my_n = [[len(s) for s in m] for m in my]
        my_p0 = [poses0[b * B + rank * per: b * B + (rank + 1) * per] for b in range(NB)]

        def rep_stream(n_steps):
            hs[0].batch_upload_raw(my_rec[0].data_ptr(), my_n[0], stride)
            for k in range(n_steps):
                b = k % NB
                h = hs[k % 2]
                h.batch_set_poses(my_p0[b]); h.batch_run()
                if k + 1 < n_steps:
                    hs[(k + 1) % 2].batch_upload_raw(my_rec[(b + 1) % NB].data_ptr(), my_n[(b + 1) % NB], stride)
                h.batch_results(with_results=False)

        rep_stream(args.warmup)
        sync_all()
        t0 = time.perf_counter()
        rep_stream(args.steps)
        sync_all()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["batch_sharded_no_collective"] = {"value": B * args.steps / float(t.item()), "unit": "registrations/s",
                                              "note": "whole scans dealt to the ranks, map replicated, no collective (independent registrations), same timed region"}
        rep2.close(); rep.close()

    # N > 1, process per GPU: the SAME job-wide batch once more through the in-library mode (ONE process -- rank 0 -- driving all N
    # devices with cfg.n_devices = N, the form a patched node uses), while the other ranks wait on the host (a store key, not a
    # collective: no kernel of theirs spins on a GPU).  Reported beside `value`; a failure here is recorded, never fatal.
    # OPT-IN (BENCH_INLIB_EXTRA=1): this leg has never run on more than one physical GPU, and a hang inside an extra would
    # take the whole N > 1 line down with it -- the driver's scaling runs must not depend on it.  `python bench.py --gpus N
    # --inlib` measures the same thing on its own.
    if runner and world > 1 and extras and os.environ.get("BENCH_BACKEND", "nccl") == "nccl" and os.environ.get("BENCH_INLIB_EXTRA") == "1":
        from torch.distributed.distributed_c10d import _get_default_store
        kv = _get_default_store()
        sync_all()
        if rank == 0:
            try:
                kc = dict(kcfg)
                kc.update(device_id=0, n_devices=world, device_ids=list(range(world)), use_graph=0, lookahead=-1)
                hm = pkg.ScanToMap(**kc)
                hm.set_map(map_xyz)

                def inlib_step(b):
                    hm.batch_upload_raw(dev_rec[b].data_ptr(), batch_npts[b], stride)
                    hm.batch_set_poses(poses0[b * B:(b + 1) * B])
                    hm.batch_run()
                    return hm.batch_results(with_results=False)[0]

                for k in range(2):
                    p_in = inlib_step(k % NB)
                n_st = max(4, args.steps // 4)
                t0 = time.perf_counter()
                for k in range(n_st):
                    p_in = inlib_step(k % NB)
                el = time.perf_counter() - t0
                pm = hm.profile()
                p_b0 = inlib_step(0)                                # batch 0 again, for the comparison with the process-per-GPU poses
                out["inlib_multi_device"] = {
                    "value": B * n_st / el, "unit": "registrations/s", "ms_per_step": 1e3 * el / n_st, "steps": n_st, "n_devices": world,
                    "exchange": {0: "peer stores from a kernel", 1: "hipMemcpyPeerAsync", 2: "host memory"}.get(int(pm.multi_exchange)),
                    "stream_syncs_in_the_loop_last_step": int(pm.multi_stream_syncs), "gn_iterations_enqueued_last_step": int(pm.multi_iterations),
                    "max_abs_pose_diff_vs_process_per_gpu": float(np.abs(p_b0 - keep["poses"]).max()) if keep.get("poses") is not None else None,
                    "note": "one process, cfg.n_devices = N, same job-wide batch; the raw records live on device 0 and reach the other devices "
                            "over xGMI inside the timer"}
                hm.close()
            except Exception as e:      # noqa: BLE001 -- an extra must not take the headline down
                out["inlib_multi_device"] = {"error": f"{type(e).__name__}: {e}"}
            kv.set("inlib_done", "1")
        else:
            kv.wait(["inlib_done"])

    if rank == 0:
        gpu0 = keep.get("poses")
        if gpu0 is not None:
            rt, rr = rmse_pair(gpu0, poses_true[:B], synth)
            out["pose_rmse_vs_truth"] = {"trans_m": rt, "rot_rad": rr, "scans": B}
        if not args.no_cpu and world == 1 and not inlib:      # the CPU baseline is timed at N = 1 only (the other ranks would idle)
            cb, cpu_poses, cpu_iters = cpu_baseline(scans[:B], map_xyz, poses0[:B], args.cpu_seconds, log)
            out["cpu_baseline"] = cb
            k = len(cpu_poses)
            if gpu0 is not None and k:
                rt, rr = rmse_pair(gpu0[:k], cpu_poses, synth)
                out["pose_rmse_vs_cpu"] = {"trans_m": rt, "rot_rad": rr, "scans": k,
                                           "bit_identical": int(sum(np.array_equal(a, b) for a, b in zip(gpu0[:k], cpu_poses))),
                                           "basis": "parity with the CPU restatement, see parity_basis"}
        if keyframes and not sharded and extras:
            # the sequence INTEGRATION.md wires into the node, per callback (MO:1846-1861): local map from the resident
            # keyframes (extractCloud MO:1556-1588) or set_map from the host, ONE registration incl. H2D of the scan and
            # D2H of the result, transformUpdate
            kc = [np.concatenate([c, np.zeros((len(c), 1), np.float32)], 1) for c, _ in keyframes]
            kp = np.stack([p for _, p in keyframes])
            node = pkg.ScanToMap(device_id=local_rank)
            store = pkg.KeyframeStore(device_id=local_rank)
            t0 = time.perf_counter()
            ids = [store.add(c) for c in kc]                       # once per keyframe, MO:2138-2142
            t_add = time.perf_counter() - t0
            n_lat = 32
            # the RAW deskewed sweeps of the first n_lat queries (cloud_info.cloud_deskewed, ~115 k points for 64x1800): what the
            # reference's callback starts from -- downsampleCurrentScan MO:1605-1611 runs on them in every callback
            boxes_n = synth.make_scene(synth.BASE_SEED, length=(max(60.0, float(args.keyframes) + 20.0) if not args.lawnmower else 80.0),
                                       density=args.density)
            raws = []
            for i in range(n_lat):
                synth.make_query(boxes_n, poses_true[i].astype(np.float64), args.sensor, seed=synth.BASE_SEED + 5000 + i,
                                 device=f"cuda:{local_rank}", scan_leaf=args.leaf_scan, raw_out=raws)
            raw_rec = []
            for r in raws:
                rec = np.zeros((len(r), 8), np.float32)
                rec[:, :3], rec[:, 3], rec[:, 4] = r[:, :3], 1.0, r[:, 3]
                raw_rec.append(rec)
            lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=-1, off_time=-1, pin_host=0)
            leaf_s = args.leaf_scan if args.leaf_scan > 0 else 0.01
            pins = []
            for rec in raw_rec:                                    # the same blobs in pinned memory (a node that keeps its message pool pinned)
                pb = pkg.PinnedBuffer(rec.nbytes)
                pb.array[:] = rec.view(np.uint8).reshape(-1)
                pins.append(pb)

            def callback_loop(h, pinned):
                """assemble (extractCloud MO:1556-1588) -> downsample + register from the raw cloud (MO:1605-1611, MO:1839-1865) ->
                transformUpdate (MO:1867-1907), per callback; returns the three times and the mean N_s."""
                ta = tr = tu = 0.0
                nds = 0
                for i in range(n_lat):
                    t0_ = time.perf_counter()
                    store.assemble(ids, kp, args.leaf_map, s2m=h, want_output=False)
                    t1_ = time.perf_counter()
                    if pinned:
                        p_, _, _, nd = h.downsampleAndScan2MapOptimization(None, len(raw_rec[i]), lay, leaf_s, poses0[i], device_ptr=pins[i].ptr)
                    else:
                        p_, _, _, nd = h.downsampleAndScan2MapOptimization(raw_rec[i], len(raw_rec[i]), lay, leaf_s, poses0[i])
                    t2_ = time.perf_counter()
                    pkg.transform_update(p_)
                    t3_ = time.perf_counter()
                    ta += t1_ - t0_; tr += t2_ - t1_; tu += t3_ - t2_
                    nds += nd
                return ta / n_lat, tr / n_lat, tu / n_lat, nds / n_lat

            callback_loop(node, False)                              # warm-up (workspaces, pools)
            t_asm, t_raw, t_upd, n_ds_mean = callback_loop(node, False)
            _, t_raw_pin, _, _ = callback_loop(node, True)
            node_pipeline = int(node.profile().pipeline)
            # the same callbacks with the launch loop (one launch per GN iteration) instead of the one-launch loop, under the
            # SAME conditions: own handle, map installed by the same asynchronous assemble right before every registration
            loop_node = pkg.ScanToMap(device_id=local_rank, pipeline=1)
            callback_loop(loop_node, False)
            _, t_raw_loop, _, _ = callback_loop(loop_node, False)
            loop_node.close()
            # the registration ALONE: pre-downsampled scan (what round 2 timed), map build finished before the timer starts
            pcl_scans = [to_records([scans[i]], 32) for i in range(n_lat)]
            node.profile()                                          # (resolves the pending build: waits for the "map ready" event)
            for i in range(3):
                node.scan2MapOptimization(pcl_scans[i], poses0[i])
            t0 = time.perf_counter()
            for i in range(n_lat):
                node.scan2MapOptimization(pcl_scans[i], poses0[i])
            t_reg_only = (time.perf_counter() - t0) / n_lat
            # ... and the voxel filter as the stand-alone host-in / host-out call (what a chain through the host would add)
            t0 = time.perf_counter()
            for i in range(8):
                pkg.voxel_grid(raws[i], leaf_s, device_id=local_rank)
            t_vox_host = (time.perf_counter() - t0) / 8
            map_rec = to_records([map_xyz], 32)
            t_set = 0.0
            for i in range(8):
                t0 = time.perf_counter()
                node.set_map(map_rec)
                t_set += time.perf_counter() - t0
            n_asm = int(node.profile().n_map)
            out["single_scan_node_path_ms"] = {
                "assemble_map_resident_keyframes": 1e3 * t_asm,
                "downsample_and_register_from_raw_cloud": 1e3 * t_raw, "the_same_from_pinned_memory": 1e3 * t_raw_pin,
                "transform_update": 1e3 * t_upd, "total_resident_keyframes": 1e3 * (t_asm + t_raw + t_upd),
                "registrations_per_s_single_stream": 1.0 / (t_asm + t_raw + t_upd),
                "raw_points_mean": float(np.mean([len(r) for r in raws])), "N_s_after_filter_mean": float(n_ds_mean),
                "register_pipeline": node_pipeline, "the_same_with_the_launch_loop_instead": 1e3 * t_raw_loop,
                "register_only_ms": 1e3 * t_reg_only, "voxel_filter_host_in_host_out_ms": 1e3 * t_vox_host,
                "set_map_from_host_instead": 1e3 * t_set / 8, "persist_fallbacks": int(node.profile().persist_fallbacks),
                "scans": n_lat,
                "note": "one scan per callback as the reference issues it (MO:432-476): extractCloud from 200 resident keyframes, then "
                        "downsampleCurrentScan + scan2MapOptimization from the RAW deskewed cloud as one device chain (lio_s2m_register_raw: one "
                        "H2D of the 32-byte records from pageable memory, voxel filter on the handle's stream, registration on its output, "
                        "D2H of the result), then transformUpdate; Python ctypes caller.  register_only_ms = lio_s2m_register on a "
                        "pre-downsampled scan with no map build pending (round 2's figure); register_pipeline 4 = the whole GN loop in one launch "
                        "(k_s2m_persist); the launch-loop comparison runs the identical callbacks on its own handle"}
            out["map_assembly"] = {"ms_keyframe_upload_total": 1e3 * t_add, "keyframes": len(kc),
                                   "points_in": int(sum(len(c) for c in kc)), "points_out": n_asm}
            for pb in pins:
                pb.close()
            store.close()
            node.close()
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
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if runner:
        runner.close()
    else:
        for h in reversed(handles):
            h.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
