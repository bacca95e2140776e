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


def plan_shards(map_xyz, world, cell=None):
    """Global grid + slab boundaries.  Deterministic, identical on every rank.

    Returns dict(origin f32[3], dims int[3], cell, inv_cell, axis, bounds int[world+1]);
    rank r owns global cells [bounds[r], bounds[r+1]) along `axis`.
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
    c = np.clip(cell_coord(map_xyz[:, axis], origin[axis], inv_cell, dims[axis]), 0, dims[axis] - 1)
    hist = np.bincount(c, minlength=int(dims[axis]))
    cum = np.cumsum(hist)
    total = cum[-1]
    bounds = np.zeros(world + 1, np.int64)
    bounds[world] = dims[axis]
    for r in range(1, world):
        bounds[r] = int(np.searchsorted(cum, total * r / world, side="left")) + 1
    bounds = np.maximum.accumulate(np.minimum(bounds, dims[axis]))
    return {"origin": origin, "dims": dims, "cell": cell, "inv_cell": inv_cell, "axis": axis, "bounds": bounds}


def shard_points(map_xyz, plan, rank):
    """Indices (ascending) of the map points rank `rank` must hold: its slab + 1-cell halo."""
    map_xyz = np.asarray(map_xyz, np.float32)
    a = plan["axis"]
    lo, hi = int(plan["bounds"][rank]), int(plan["bounds"][rank + 1])
    if len(map_xyz) == 0:
        return np.zeros(0, np.int64)
    c = np.clip(cell_coord(map_xyz[:, a], plan["origin"][a], plan["inv_cell"], plan["dims"][a]), 0, plan["dims"][a] - 1)
    return np.nonzero((c >= lo - 1) & (c < hi + 1))[0]


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
    """GN loop over a sharded map: partial sums on this GPU -> all-reduce -> solve.

    The shard's neighbour indices refer to the shard-local order; poses, iteration
    counts and degeneracy flags are identical on every rank because every rank
    solves from the same all-reduced sums.
    """

    def __init__(self, s2m, map_xyz, rank, world, dist, torch, deterministic=False, lookahead=1):
        self.s2m, self.rank, self.world, self.dist, self.torch = s2m, rank, world, dist, torch
        self.plan = plan_shards(map_xyz, world)
        self.idx = shard_points(map_xyz, self.plan, rank)
        s2m.set_map(np.ascontiguousarray(np.asarray(map_xyz, np.float32)[self.idx]))
        s2m.set_global_grid([float(v) for v in self.plan["origin"]], [int(v) for v in self.plan["dims"]])
        s2m.set_shard(self.plan["axis"], int(self.plan["bounds"][rank]), int(self.plan["bounds"][rank + 1]))
        # kernels and the collective share torch's current stream
        s2m.set_stream(torch.cuda.current_stream().cuda_stream)
        self.deterministic = deterministic
        self.lookahead = lookahead      # iterations enqueued ahead of the all-scans-done check
        self.sums = None

    def run(self):
        s2m, torch, dist = self.s2m, self.torch, self.dist
        n = s2m._n_scans
        if self.sums is None or self.sums.shape[0] != n:
            self.sums = torch.zeros((n, SUMS), dtype=torch.float64, device="cuda")
            self.gathered = torch.zeros((self.world * n, SUMS), dtype=torch.float64, device="cuda")
        s2m.batch_begin()
        iters = 0
        for it in range(s2m.cfg.max_iters):                      # MO:1848
            s2m.batch_iter_partial(self.sums.data_ptr())
            if self.deterministic:
                # bitwise reproducible across runs: gather, then sum in rank order
                dist.all_gather_into_tensor(self.gathered, self.sums)
                g = self.gathered.view(self.world, n, SUMS)
                self.sums.copy_(g[0])
                for r in range(1, self.world):
                    self.sums.add_(g[r])
            else:
                dist.all_reduce(self.sums, op=dist.ReduceOp.SUM)
            s2m.batch_iter_apply(self.sums.data_ptr())
            iters += 1
            # MO:1857-1858 for every scan.  Every rank solves the same sums, so every rank sees the
            # same count and leaves the loop at the same iteration (the collectives stay matched).
            chk = it - self.lookahead
            if chk >= 0 and s2m.batch_poll_active(chk) == 0:
                break
        return iters
