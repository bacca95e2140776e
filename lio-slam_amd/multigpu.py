"""Map sharding across the GPUs of one node (north_star).

One process per GPU.  The local keyframe map (laserCloudSurfFromMapDS, MO:149)
is cut into slabs of hash-grid cells along its longest axis, balanced by point
count; each rank keeps its slab plus a ONE-CELL halo on both sides.  Because a
cell edge is >= 1.001 m and the reference only accepts a plane whose 5th
neighbour is closer than 1 m (MO:1641), every accepted scan point finds its
exact 5 neighbours inside the shard of the rank that owns the point's cell, so
the correspondence sets are identical to the unsharded run.  A scan point is
processed by exactly one rank (owner-computes on the cell of its transformed
position), and the only exchange is the per-iteration sum of the 6x6 JtJ,
6x1 Jtr and the correspondence count -- one all-reduce of n_scans x 32 doubles.

`plan_shards`/`shard_points`/`owner_mask` are plain numpy host logic (tested on
CPU with gloo); `ShardedRunner` drives the HIP path through the C ABI hooks.
"""
import numpy as np

SUMS = 32   # doubles per scan: 21 upper JtJ, 6 Jtr, N_c, pad (csrc/lio_types.h)


def default_cell(max_sq_dist=1.0):
    return np.float32(np.sqrt(np.float32(max_sq_dist)) * np.float32(1.001))


def cell_coord(v, origin, inv_cell, n):
    """fp32 cell index exactly as lio_cell_coord() in csrc/lio_kernels.hip."""
    c = np.floor((np.asarray(v, np.float32) - np.float32(origin)) * np.float32(inv_cell))
    c = np.minimum(np.maximum(c, np.float32(-4.0)), np.float32(n + 3))
    return c.astype(np.int64)


def plan_shards(map_xyz, world, cell=None, load_xyz=None):
    """Global grid + slab boundaries.  Deterministic, identical on every rank.

    Returns dict(origin f32[3], dims int[3], cell, inv_cell, axis, bounds int[world+1]);
    rank r owns global cells [bounds[r], bounds[r+1]) along `axis`.
    load_xyz: optional world-frame sample of the points that will be REGISTERED (scan points at their initial poses).
    A rank's work is the scan points that fall into its slab, not the map points it stores; with a sample the slabs
    are balanced by it (scan returns are dense around the sensor, the map is not).  Any bounds give the same results.
    """
    map_xyz = np.asarray(map_xyz, np.float32)
    cell = np.float32(cell if cell is not None else default_cell())
    inv_cell = np.float32(1.0) / cell
    if len(map_xyz) == 0:
        return {"origin": np.zeros(3, np.float32), "dims": np.ones(3, np.int64), "cell": cell,
                "inv_cell": inv_cell, "axis": 0, "bounds": np.zeros(world + 1, np.int64)}
    mn = map_xyz.min(0)
    mx = map_xyz.max(0)
    origin = (mn - np.float32(0.5) * cell).astype(np.float32)
    dims = (np.floor((mx.astype(np.float64) - origin) * inv_cell) + 2).astype(np.int64)
    axis = int(np.argmax(dims))
    src = map_xyz if load_xyz is None or len(load_xyz) == 0 else np.asarray(load_xyz, np.float32)
    c = np.clip(cell_coord(src[:, axis], origin[axis], inv_cell, dims[axis]), 0, dims[axis] - 1)
    hist = np.bincount(c, minlength=int(dims[axis]))
    cum = np.cumsum(hist)
    total = cum[-1]
    bounds = np.zeros(world + 1, np.int64)
    bounds[world] = dims[axis]
    for r in range(1, world):
        bounds[r] = int(np.searchsorted(cum, total * r / world, side="left")) + 1
    bounds = np.maximum.accumulate(np.minimum(bounds, dims[axis]))
    return {"origin": origin, "dims": dims, "cell": cell, "inv_cell": inv_cell, "axis": axis, "bounds": bounds}


def shard_points(map_xyz, plan, rank, halo=1):
    """Indices (ascending) of the map points rank `rank` must hold: its slab + `halo` cells on each side."""
    map_xyz = np.asarray(map_xyz, np.float32)
    a = plan["axis"]
    lo, hi = int(plan["bounds"][rank]), int(plan["bounds"][rank + 1])
    if len(map_xyz) == 0:
        return np.zeros(0, np.int64)
    c = np.clip(cell_coord(map_xyz[:, a], plan["origin"][a], plan["inv_cell"], plan["dims"][a]), 0, plan["dims"][a] - 1)
    return np.nonzero((c >= lo - halo) & (c < hi + halo))[0]


def owner_mask(q_world, plan, rank):
    """True for transformed scan points (fp32 [n,3]) that rank `rank` processes."""
    a = plan["axis"]
    lo, hi = int(plan["bounds"][rank]), int(plan["bounds"][rank + 1])
    c = np.clip(cell_coord(q_world[:, a], plan["origin"][a], plan["inv_cell"], plan["dims"][a]), 0, plan["dims"][a] - 1)
    return (c >= lo) & (c < hi)


def transform_f32(T, xyz):
    """pointAssociateToMap MO:841-847 in fp32 with the reference's operator order."""
    T = np.asarray(T, np.float32).reshape(3, 4)
    x, y, z = (np.asarray(xyz[:, k], np.float32) for k in range(3))
    out = np.empty((len(xyz), 3), np.float32)
    for r in range(3):
        out[:, r] = ((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3]
    return out


class ShardedRunner:
    """Gauss-Newton loop of one rank when a job is spread over `world` GPUs.

    mode "map" : the map is cut into slabs + halo, owner-computes (north_star).
    mode "scan": the map is replicated and every rank takes one world-th of the workgroups of
                 every scan (SURVEY 8e alternative): no halo, no ownership tests, balanced.
    Either way the per-scan sums (n_scans x 32 doubles) are all-reduced once per iteration and every
    rank runs the same solve, so poses, iteration counts and degeneracy flags are identical on all
    ranks.  The batch is processed as `groups` sub-batches in a software pipeline: while the
    all-reduce of one sub-batch is in flight (RCCL runs on its own stream), the association kernel of
    the next one runs, so the collective latency is hidden behind compute.
    """

    def __init__(self, pkg, map_xyz, rank, world, dist, torch, mode="map", groups=2, deterministic=False,
                 lookahead=2, load_xyz=None, halo_cells=16, **cfg):
        self.rank, self.world, self.dist, self.torch = rank, world, dist, torch
        self.mode, self.deterministic, self.lookahead = mode, deterministic, lookahead
        self.handles = []
        map_xyz = np.ascontiguousarray(map_xyz, np.float32)
        if mode == "map":
            # the slab plan must use the cell the device-side owner test uses (lio_s2m_set_shard: cfg.cell_size, or
            # sqrt(max_sq_dist) * 1.001): a one-cell halo only guarantees exact 5-NN sets for cells of at least the gate radius
            cell = np.float32(cfg.get("cell_size") or 0.0)
            if not cell > 0:
                cell = default_cell(cfg.get("max_sq_dist", 1.0))
            self.plan = plan_shards(map_xyz, world, cell=cell, load_xyz=load_xyz)
            # a halo wider than the one cell exactness needs lets whole workgroups be owned by one rank (lio_s2m_set_shard_plan)
            self.halo = max(1, int(halo_cells))
            self.idx = shard_points(map_xyz, self.plan, rank, self.halo)
        else:
            self.idx = np.arange(len(map_xyz))
        # one stream per sub-batch: the latency-bound pieces of one sub-batch (all-reduce, solve) run under the
        # association kernel of the other one
        self.streams = [torch.cuda.Stream() for _ in range(max(1, groups))]
        for k in range(max(1, groups)):
            stream = self.streams[k].cuda_stream
            s2m = pkg.ScanToMap(**cfg)
            if mode == "map":
                s2m.set_map(np.ascontiguousarray(map_xyz[self.idx]))
                s2m.set_global_grid([float(v) for v in self.plan["origin"]], [int(v) for v in self.plan["dims"]])
                s2m.set_shard_plan(self.plan["axis"], rank, [int(v) for v in self.plan["bounds"]], self.halo)
            else:
                s2m.set_map(map_xyz)
                s2m.set_scan_shard(rank, world)
            s2m.set_stream(stream)           # kernels and collectives of a sub-batch are ordered through its stream
            self.handles.append(s2m)
        self.split = []                      # (first, last) scan of every group
        self.sums, self.gathered = [], []

    def upload(self, scans):
        n, g = len(scans), len(self.handles)
        if n < g:
            g = 1
        bounds = [n * k // g for k in range(g + 1)]
        self.split = [(bounds[k], bounds[k + 1]) for k in range(g)]
        self.sums, self.gathered = [], []
        for k, ((a, b), h) in enumerate(zip(self.split, self.handles)):
            h.batch_upload(scans[a:b])
            with self.torch.cuda.stream(self.streams[k]):
                self.sums.append(self.torch.zeros((b - a, SUMS), dtype=self.torch.float64, device="cuda"))
                self.gathered.append(self.torch.zeros((self.world * (b - a), SUMS), dtype=self.torch.float64, device="cuda")
                                     if self.deterministic else None)
        self.torch.cuda.synchronize()

    def upload_raw(self, base_ptr, n_pts, stride):
        """Batch laid out back to back at `base_ptr` (pinned host or device memory), see ScanToMap.batch_upload_raw."""
        n, g = len(n_pts), len(self.handles)
        if n < g:
            g = 1
        bounds = [n * k // g for k in range(g + 1)]
        split = [(bounds[k], bounds[k + 1]) for k in range(g)]
        offs = np.concatenate([[0], np.cumsum(np.asarray(n_pts, np.int64) * stride)])
        for k, ((a, b), h) in enumerate(zip(split, self.handles)):
            h.batch_upload_raw(int(base_ptr) + int(offs[a]), list(n_pts[a:b]), stride, asynchronous=False)
        if split != self.split or not self.sums:
            self.split = split
            self.sums, self.gathered = [], []
            for k, (a, b) in enumerate(split):
                with self.torch.cuda.stream(self.streams[k]):
                    self.sums.append(self.torch.zeros((b - a, SUMS), dtype=self.torch.float64, device="cuda"))
                    self.gathered.append(self.torch.zeros((self.world * (b - a), SUMS), dtype=self.torch.float64, device="cuda")
                                         if self.deterministic else None)
            self.torch.cuda.synchronize()

    def set_poses(self, poses):
        poses = np.ascontiguousarray(poses, np.float32)
        for (a, b), h in zip(self.split, self.handles):
            h.batch_set_poses(poses[a:b])

    def _reduce(self, k):
        with self.torch.cuda.stream(self.streams[k]):
            return self._reduce_on_current_stream(k)

    def _reduce_on_current_stream(self, k):
        dist = self.dist
        if self.deterministic:
            # bitwise reproducible across runs: gather, then sum in rank order
            dist.all_gather_into_tensor(self.gathered[k], self.sums[k])
            g = self.gathered[k].view(self.world, -1, SUMS)
            self.sums[k].copy_(g[0])
            for r in range(1, self.world):
                self.sums[k].add_(g[r])
            return None
        return dist.all_reduce(self.sums[k], op=dist.ReduceOp.SUM, async_op=True)

    def run(self):
        hs = self.handles[:len(self.split)]
        live = [True] * len(hs)
        work = [None] * len(hs)
        for k, h in enumerate(hs):                                   # prologue: iteration 0 of every group
            h.batch_begin()
            h.batch_iter_partial(self.sums[k].data_ptr())
            work[k] = self._reduce(k)
        iters = 0
        max_iters = hs[0].cfg.max_iters
        for it in range(max_iters):                                  # MO:1848
            for k, h in enumerate(hs):
                if not live[k]:
                    continue
                if work[k] is not None:
                    with self.torch.cuda.stream(self.streams[k]):
                        work[k].wait()                               # the sub-batch's stream waits, the host does not
                h.batch_iter_apply(self.sums[k].data_ptr())          # MO:1784-1835 from the global sums
                # MO:1857-1858 for every scan of the group.  Every rank solves the same sums, so every
                # rank sees the same counts and stops issuing collectives at the same iteration.
                chk = it - self.lookahead
                if (chk >= 0 and h.batch_poll_active(chk) == 0) or it + 1 >= max_iters:
                    live[k] = False
                    continue
                h.batch_iter_partial(self.sums[k].data_ptr())        # next iteration of this group ...
                work[k] = self._reduce(k)                            # ... its all-reduce overlaps the other group
            iters += 1
            if not any(live):
                break
        return iters

    def results(self, with_results=True):
        poses, res = [], []
        for (a, b), h in zip(self.split, self.handles):
            p, r = h.batch_results(with_results)
            poses.append(p)
            if with_results:
                res.extend(list(r))
        return np.concatenate(poses), (res if with_results else None)

    def close(self):
        for h in self.handles:
            h.close()
