"""Synthetic street-canyon scenes, spinning-lidar scans and keyframe maps.

Data generator for tests and bench.py: analytic scene (ground
plane + axis-aligned boxes), closed-form ray casting, range noise, per-point
time/ring, keyframe map assembly with a PCL-VoxelGrid-style centroid filter
(the feeders of the hot path, MO:1605-1611 / MO:1556-1588 -- generator only,
not a parity-checked kernel).  torch is used as an array library so the ray
caster runs on the GPU when one is present; nothing here is on the product
path.
"""
import math

import numpy as np
import torch

SENSOR_HEIGHT = 1.8
BASE_SEED = 20241022

SENSORS = {
    # name: (n_rings, n_cols, elev_min_deg, elev_max_deg)
    "vlp16": (16, 1800, -15.0, 15.0),
    "hdl64": (64, 1800, -24.8, 2.0),       # headline 64x1800
    "os1_64": (64, 1024, -22.5, 22.5),
    "os1_128": (128, 2048, -22.5, 22.5),
}


# --------------------------------------------------------------------- scene
def make_scene(seed=BASE_SEED, length=260.0, kind="street", density=0.02):
    """Returns boxes float32 [n,6] = (xmin,ymin,zmin,xmax,ymax,zmax); ground is z=0.
    density: box obstacles per m^2 of street (0.02)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    boxes = []
    x0 = -80.0
    x1 = length + 80.0
    if kind == "corridor":
        # ground + two long parallel walls: translation along x is unobservable
        boxes.append((x0 - 200.0, 8.0, 0.0, x1 + 200.0, 9.0, 12.0))
        boxes.append((x0 - 200.0, -9.0, 0.0, x1 + 200.0, -8.0, 12.0))
        return np.array(boxes, np.float32)
    # canyon walls in 20 m blocks, half-width 8..15 m, height 6..15 m
    xb = x0
    while xb < x1:
        for side in (+1.0, -1.0):
            w = rng.uniform(8.0, 15.0)
            h = rng.uniform(6.0, 15.0)
            lo, hi = (w, w + 1.5) if side > 0 else (-w - 1.5, -w)
            boxes.append((xb, lo, 0.0, xb + 20.0, hi, h))
            # return wall closing the step to the deepest facade (gives x-normals)
            boxes.append((xb + 19.0, min(lo, side * 8.0), 0.0, xb + 20.0, max(hi, side * 16.5), h))
        xb += 20.0
    # end caps
    boxes.append((x0 - 1.5, -17.0, 0.0, x0, 17.0, 12.0))
    boxes.append((x1, -17.0, 0.0, x1 + 1.5, 17.0, 12.0))
    # box obstacles, density 0.02 / m^2 over the street, keeping |y| > 2.5 m free
    area = (x1 - x0) * 16.0
    n_obs = int(density * area)
    for _ in range(n_obs):
        sx, sy, sz = rng.uniform(1.0, 4.0, 3)
        cx = rng.uniform(x0, x1)
        cy = rng.uniform(2.5 + sy / 2, 8.0) * (1.0 if rng.uniform() < 0.5 else -1.0)
        boxes.append((cx - sx / 2, cy - sy / 2, 0.0, cx + sx / 2, cy + sy / 2, sz))
    return np.array(boxes, np.float32)


# ------------------------------------------------------------------ rotations
def rpy_matrix(roll, pitch, yaw):
    """Rz(yaw) Ry(pitch) Rx(roll) in float64 (pcl::getTransformation convention, MO:889)."""
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]], np.float64)


def pose_matrix(pose):
    """pose = [roll,pitch,yaw,x,y,z] -> 4x4 float64."""
    T = np.eye(4)
    T[:3, :3] = rpy_matrix(pose[0], pose[1], pose[2])
    T[:3, 3] = pose[3:6]
    return T


def matrix_to_pose(T):
    pitch = -math.asin(max(-1.0, min(1.0, T[2, 0])))
    roll = math.atan2(T[2, 1], T[2, 2])
    yaw = math.atan2(T[1, 0], T[0, 0])
    return np.array([roll, pitch, yaw, T[0, 3], T[1, 3], T[2, 3]], np.float64)


# ----------------------------------------------------------------- ray caster
def _sensor_dirs(sensor, device):
    n_rings, n_cols, e0, e1 = SENSORS[sensor] if isinstance(sensor, str) else sensor
    elev = torch.linspace(math.radians(e0), math.radians(e1), n_rings, dtype=torch.float64, device=device)
    az = torch.arange(n_cols, dtype=torch.float64, device=device) * (2.0 * math.pi / n_cols)
    # column-major in time: for each column all rings (like a real driver)
    ce, se = torch.cos(elev)[None, :], torch.sin(elev)[None, :]
    ca, sa = torch.cos(az)[:, None], torch.sin(az)[:, None]
    d = torch.stack([ce * ca, ce * sa, se.expand(n_cols, n_rings)], dim=-1)  # [cols, rings, 3]
    ring = torch.arange(n_rings, device=device)[None, :].expand(n_cols, n_rings)
    col = torch.arange(n_cols, device=device)[:, None].expand(n_cols, n_rings)
    return d.reshape(-1, 3), ring.reshape(-1), col.reshape(-1), n_rings, n_cols


def cast_scan(boxes, pose, sensor="vlp16", seed=0, noise=0.02, max_range=100.0,
              omega=(0.0, 0.0, 0.0), scan_period=0.1, device=None, chunk=1 << 15):
    """Ray-cast one sweep.

    pose: sensor pose at the START of the sweep [roll,pitch,yaw,x,y,z] (lidar -> world).
    omega: body-frame angular rate (rad/s); the sensor orientation at relative
    time t is R0 * Rz(wz t) Ry(wy t) Rx(wx t) (what an ideal gyro integrates to,
    IP:405-407), points are reported in the instantaneous sensor frame.
    Returns dict of numpy arrays in driver order: xyz[n,3] f32, intensity, ring (u16),
    time (f32, seconds from sweep start), col (i32), range (f32).
    """
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    d_s, ring, col, n_rings, n_cols = _sensor_dirs(sensor, device)
    n = d_s.shape[0]
    t_rel = col.to(torch.float64) * (scan_period / n_cols)
    T0 = torch.tensor(pose_matrix(pose), dtype=torch.float64, device=device)
    R0, o = T0[:3, :3], T0[:3, 3]
    w = torch.tensor(omega, dtype=torch.float64, device=device)
    if float(w.abs().sum()) > 0:
        ax, ay, az_ = (w[0] * t_rel), (w[1] * t_rel), (w[2] * t_rel)
        cr, sr, cp, sp, cy, sy = torch.cos(ax), torch.sin(ax), torch.cos(ay), torch.sin(ay), torch.cos(az_), torch.sin(az_)
        Rt = torch.stack([
            torch.stack([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], -1),
            torch.stack([sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], -1),
            torch.stack([-sp, cp * sr, cp * cr], -1)], -2)          # [n,3,3]
        d_w = torch.einsum("ij,njk,nk->ni", R0, Rt, d_s)
    else:
        d_w = d_s @ R0.T
    bx = torch.tensor(np.asarray(boxes, np.float64), dtype=torch.float64, device=device)
    # cull boxes farther than max_range from the sensor
    if bx.shape[0]:
        c = torch.clamp(o[None, :], bx[:, :3], bx[:, 3:])
        bx = bx[((c - o[None, :]) ** 2).sum(-1) <= max_range ** 2]
    t_hit = torch.full((n,), float("inf"), dtype=torch.float64, device=device)
    for s in range(0, n, chunk):
        d = d_w[s:s + chunk]
        # ground z = 0
        tz = torch.where(d[:, 2] < -1e-12, -o[2] / d[:, 2], torch.full_like(d[:, 2], float("inf")))
        best = tz
        if bx.shape[0]:
            inv = 1.0 / torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)   # [m,3]
            t1 = (bx[None, :, :3] - o[None, None, :]) * inv[:, None, :]
            t2 = (bx[None, :, 3:] - o[None, None, :]) * inv[:, None, :]
            tn = torch.minimum(t1, t2).amax(-1)
            tf = torch.maximum(t1, t2).amin(-1)
            hit = (tf >= tn) & (tf > 0)
            tb = torch.where(hit, torch.where(tn > 0, tn, tf), torch.full_like(tn, float("inf")))
            best = torch.minimum(best, tb.amin(-1))
        t_hit[s:s + chunk] = best
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    nz = torch.randn(n, generator=g, dtype=torch.float64).to(device) * noise
    inten = (torch.rand(n, generator=g, dtype=torch.float64) * 100.0).to(device)
    rng_m = t_hit + nz
    keep = torch.isfinite(t_hit) & (rng_m < max_range) & (rng_m > 0.5)
    p = d_s * rng_m[:, None]
    keep_idx = torch.nonzero(keep).squeeze(-1)
    out = {
        "xyz": p[keep_idx].to(torch.float32).cpu().numpy(),
        "intensity": inten[keep_idx].to(torch.float32).cpu().numpy(),
        "ring": ring[keep_idx].to(torch.int32).cpu().numpy().astype(np.uint16),
        "time": t_rel[keep_idx].to(torch.float32).cpu().numpy(),
        "col": col[keep_idx].to(torch.int32).cpu().numpy(),
        "range": rng_m[keep_idx].to(torch.float32).cpu().numpy(),
        "n_rings": n_rings, "n_cols": n_cols,
    }
    return out


# ----------------------------------------------------------------- voxel grid
def voxel_downsample(xyz, leaf):
    """Centroid voxel filter with pcl::VoxelGrid's voxel indexing and output
    order (ascending x-fastest linear voxel index) -- generator only (A12)."""
    xyz = np.asarray(xyz, np.float32)
    if len(xyz) == 0:
        return xyz.reshape(0, 3)
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(xyz * inv).astype(np.int64)
    mn = ijk.min(0)
    div = ijk.max(0) - mn + 1
    key = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * div[0] + (ijk[:, 2] - mn[2]) * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    first = np.concatenate([[True], ks[1:] != ks[:-1]])
    starts = np.nonzero(first)[0]
    sums = np.add.reduceat(xyz[order].astype(np.float64), starts, axis=0)
    cnt = np.diff(np.concatenate([starts, [len(ks)]]))
    return (sums / cnt[:, None]).astype(np.float32)


def transform_points(xyz, pose):
    """fp32 rigid transform with the reference's row-by-row form (MO:849-868)."""
    T = pose_matrix(pose).astype(np.float32)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    out = np.empty_like(xyz, dtype=np.float32)
    for r in range(3):
        out[:, r] = T[r, 0] * x + T[r, 1] * y + T[r, 2] * z + T[r, 3]
    return out


# ------------------------------------------------------- trajectories & maps
def keyframe_poses(n_keyframes, spacing=1.0, seed=BASE_SEED, lawnmower=False):
    """True keyframe poses [n,6].  Straight +x run every `spacing` m (UT:312), or
    a lawn-mower sweep inside a 50 m radius for very large keyframe counts."""
    rng = np.random.Generator(np.random.MT19937(seed + 7))
    poses = np.zeros((n_keyframes, 6), np.float64)
    if not lawnmower:
        poses[:, 3] = np.arange(n_keyframes) * spacing
        poses[:, 4] = 0.3 * np.sin(np.arange(n_keyframes) * 0.05)
        poses[:, 2] = 0.02 * np.sin(np.arange(n_keyframes) * 0.08)
    else:
        # rows along x of length 70 m, y stepping by 0.1 m within +-2 m of the centre line
        per_row = 70
        for k in range(n_keyframes):
            row, j = divmod(k, per_row)
            xpos = j * spacing if row % 2 == 0 else (per_row - 1 - j) * spacing
            poses[k, 3] = xpos
            poses[k, 4] = -2.0 + 0.28 * row
            poses[k, 2] = 0.0 if row % 2 == 0 else math.pi
    poses[:, 5] = SENSOR_HEIGHT
    poses[:, 0] += rng.normal(0, 0.003, n_keyframes)
    poses[:, 1] += rng.normal(0, 0.003, n_keyframes)
    return poses


def build_map(boxes, kf_poses, sensor="vlp16", seed=BASE_SEED, scan_leaf=0.4, map_leaf=0.5,
              pose_noise=(0.01, math.radians(0.05)), device=None, progress=None, keep=None):
    """Keyframe scans at true poses -> voxel 0.4 -> to world with noisy poses ->
    union -> voxel 0.5 (extractCloud MO:1556-1588).  Returns float32 [N_m,3]."""
    rng = np.random.Generator(np.random.MT19937(seed + 13))
    clouds = []
    for k, pose in enumerate(kf_poses):
        sc = cast_scan(boxes, pose, sensor, seed=seed + 1000 + k, device=device)
        ds = voxel_downsample(sc["xyz"], scan_leaf)
        noisy = np.array(pose, np.float64)
        noisy[3:6] += rng.normal(0, pose_noise[0], 3)
        noisy[0:3] += rng.normal(0, pose_noise[1], 3)
        clouds.append(transform_points(ds, noisy))
        if keep is not None:        # keyframe clouds in the lidar frame + the poses they are placed with
            keep.append((ds, noisy.astype(np.float32)))
        if progress and (k % 20 == 0):
            progress(k, len(kf_poses))
    return voxel_downsample(np.concatenate(clouds, 0), map_leaf)


def make_query(boxes, true_pose, sensor="vlp16", seed=0, scan_leaf=0.4,
               guess_sigma=(0.10, math.radians(1.0)), device=None, raw_out=None):
    """Query scan at true_pose, voxel-downsampled (downsampleCurrentScan MO:1605-1611),
    and a perturbed initial guess.  Returns (scan_xyz f32 [N_s,3], pose_init f32[6]).
    scan_leaf <= 0: PCL's pass-through (the leaf overflows the voxel index, config/6t.yaml:112): the raw cloud.
    raw_out (a list): also receives the un-downsampled sweep as float32 [n,4] x,y,z,intensity (cloud_deskewed)."""
    rng = np.random.Generator(np.random.MT19937(seed + 17))
    sc = cast_scan(boxes, true_pose, sensor, seed=seed, device=device)
    ds = voxel_downsample(sc["xyz"], scan_leaf) if scan_leaf > 0 else sc["xyz"]
    if raw_out is not None:
        raw_out.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], 1).astype(np.float32))
    init = np.array(true_pose, np.float64)
    init[3:6] += rng.normal(0, guess_sigma[0], 3)
    init[0:3] += rng.normal(0, guess_sigma[1], 3)
    return ds, init.astype(np.float32)


def make_case(sensor="vlp16", n_keyframes=10, seed=BASE_SEED, kind="street", n_queries=1,
              lawnmower=False, device=None, progress=None, sensor_override=None, q_range=None, with_map=True, workers=1,
              scan_leaf=0.4, map_leaf=0.5, density=0.02, n_raw=0):
    """One registration workload: map + n_queries (scan, true pose, initial guess).

    q_range=(a, b): only queries a..b-1 are ray-cast (the others are None) -- the ranks of a multi-GPU bench
    each generate their share; every query is the same whichever rank makes it.  with_map=False skips the map.
    scan_leaf / map_leaf: mappingSurfLeafSize / surroundingKeyframeMapLeafSize (0.4 / 0.5 in lio_sam_default.yaml:56,71;
    the keyframe clouds are filtered with scan_leaf as MO:2136-2142 stores laserCloudSurfLastDS); density: obstacles per m^2;
    n_raw: the first n_raw queries also keep their raw sweep ("raw": [n,4] x,y,z,intensity)."""
    sensor = sensor_override or sensor
    length = max(60.0, float(n_keyframes) + 20.0) if not lawnmower else 80.0
    boxes = make_scene(seed, length=length, kind=kind, density=density)
    kfs = keyframe_poses(n_keyframes, seed=seed, lawnmower=lawnmower)
    kept = []
    kf_leaf = scan_leaf if scan_leaf > 0 else 0.05          # (a pass-through keyframe cloud would be ~115 k points each)
    map_xyz = build_map(boxes, kfs, sensor, seed=seed, device=device, progress=progress, keep=kept,
                        scan_leaf=kf_leaf, map_leaf=map_leaf) if with_map else None
    rng = np.random.Generator(np.random.MT19937(seed + 29))
    # along the path, 0.5 m past a keyframe (the last one for q == 0); drawn for all queries up front
    ks = [n_keyframes - 1 if q == 0 else int(rng.integers(0, n_keyframes)) for q in range(n_queries)]
    a, b = q_range if q_range is not None else (0, n_queries)

    def one(q):
        tp = np.array(kfs[ks[q]], np.float64)
        tp[3] += 0.5 * math.cos(tp[2])
        tp[4] += 0.5 * math.sin(tp[2])
        raw = [] if q < n_raw else None
        scan, init = make_query(boxes, tp, sensor, seed=seed + 5000 + q, device=device, scan_leaf=scan_leaf, raw_out=raw)
        out = {"scan": scan, "pose_true": tp.astype(np.float32), "pose_init": init}
        if raw:
            out["raw"] = raw[0]
        return out

    todo = [q for q in range(n_queries) if a <= q < b]
    if workers > 1 and len(todo) > 8:
        # every query has its own seed, so the result does not depend on the schedule; the host half of a query
        # (numpy sort of ~100 k points for the voxel filter) releases the GIL
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            done = dict(zip(todo, ex.map(one, todo)))
    else:
        done = {q: one(q) for q in todo}
    queries = [done.get(q) for q in range(n_queries)]
    return {"map": map_xyz, "queries": queries, "boxes": boxes, "kf_poses": kfs, "keyframes": kept}


# ------------------------------------------------------------- edge features
# Extension data: this reference has no corner features; for the point-to-line
# residuals of upstream LIO-SAM the generator samples the vertical edges of the scene boxes the way
# a spinning lidar sees them -- one return per ring where the ring's cone meets the edge.
def cast_edges(boxes, pose, sensor="vlp16", seed=0, noise=0.01, max_range=40.0):
    """Edge returns of one sweep in the LIDAR frame, float32 [n,3] (occlusion is ignored)."""
    n_rings, _, e0, e1 = SENSORS[sensor] if isinstance(sensor, str) else sensor
    rng = np.random.Generator(np.random.MT19937(seed + 41))
    pose = np.asarray(pose, np.float64)
    b = np.asarray(boxes, np.float64)
    ex = np.concatenate([b[:, 0], b[:, 0], b[:, 3], b[:, 3]])
    ey = np.concatenate([b[:, 1], b[:, 4], b[:, 1], b[:, 4]])
    ez0 = np.tile(b[:, 2], 4)
    ez1 = np.tile(b[:, 5], 4)
    d = np.hypot(ex - pose[3], ey - pose[4])
    keep = (d > 1.0) & (d < max_range)
    ex, ey, ez0, ez1, d = ex[keep], ey[keep], ez0[keep], ez1[keep], d[keep]
    elev = np.radians(np.linspace(e0, e1, n_rings))
    z = pose[5] + d[:, None] * np.tan(elev)[None, :]
    ok = (z > ez0[:, None] + 0.05) & (z < ez1[:, None] - 0.05)
    wx = np.broadcast_to(ex[:, None], z.shape)[ok]
    wy = np.broadcast_to(ey[:, None], z.shape)[ok]
    world = np.stack([wx, wy, z[ok]], 1)
    T = pose_matrix(pose)
    local = (world - T[:3, 3]) @ T[:3, :3]                    # R^T (p - t)
    local += rng.normal(0, noise, local.shape)
    return local.astype(np.float32)


def add_corners(case, sensor="vlp16", seed=BASE_SEED, map_leaf=0.2,
                pose_noise=(0.01, math.radians(0.05))):
    """Adds "corner_map" and per-query "corners" to a make_case() dict (in place; returns it)."""
    rng = np.random.Generator(np.random.MT19937(seed + 43))
    clouds = []
    for k, pose in enumerate(case["kf_poses"]):
        loc = cast_edges(case["boxes"], pose, sensor, seed=seed + 2000 + k)
        noisy = np.array(pose, np.float64)
        noisy[3:6] += rng.normal(0, pose_noise[0], 3)
        noisy[0:3] += rng.normal(0, pose_noise[1], 3)
        clouds.append(transform_points(loc, noisy))
    allc = np.concatenate(clouds, 0) if clouds else np.zeros((0, 3), np.float32)
    case["corner_map"] = voxel_downsample(allc, map_leaf) if len(allc) else allc
    for q, qu in enumerate(case["queries"]):
        qu["corners"] = cast_edges(case["boxes"], qu["pose_true"], sensor, seed=seed + 7000 + q)
    return case


# ------------------------------------------------------- cloud_info arrays (A4)
# The reference's featureExtraction consumes startRingIndex / endRingIndex / pointColInd / pointRange
# (MSG:4-8) but nothing in it produces them.  This builds them the way upstream
# LIO-SAM's cloudExtraction does from an organised sweep: ring-major, ascending column, one return per
# (ring, column) cell; startRingIndex = first + 5, endRingIndex = last - 5.
def organize_scan(sc):
    """sc = cast_scan() dict -> dict(cloud [n,4] xyzi, start_ring, end_ring, col, range) in ring-major order."""
    n_rings = int(sc["n_rings"])
    order = np.lexsort((sc["col"], sc["ring"]))
    ring = sc["ring"][order].astype(np.int64)
    cloud = np.concatenate([sc["xyz"][order], sc["intensity"][order][:, None]], 1).astype(np.float32)
    start = np.zeros(n_rings, np.int32)
    end = np.zeros(n_rings, np.int32)
    count = 0
    for i in range(n_rings):
        start[i] = count - 1 + 5
        count += int((ring == i).sum())
        end[i] = count - 1 - 5
    return {"cloud": cloud, "start_ring": start, "end_ring": end,
            "col": sc["col"][order].astype(np.int32), "range": sc["range"][order].astype(np.float32)}
