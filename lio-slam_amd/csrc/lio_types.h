// lio_types.h -- device-resident layouts shared by the kernels and the C-ABI
// implementation.  Names follow the reference's domain (scan, map, cell,
// correspondence), MO = src/liorf/src/mapOptmization.cpp.
#pragma once
#include <stdint.h>

#define LIO_SUMS 32          // per-scan sums: 21 upper JtJ + 6 Jtr + N_c + pad to 32 doubles
#define LIO_SUM_NC 27
#define LIO_BLOCK 256        // threads per association workgroup (4 waves)
#define LIO_ROW_ALIGN 8      // records: every neighbourhood row list starts at a multiple of this and is padded to one

// Hash grid over the voxel-downsampled local map (laserCloudSurfFromMapDS,
// MO:149).  cell(v) = floor((v - origin) * inv_cell), linear id x-fastest so
// that three x-adjacent cells are one contiguous run of sorted points.
#define LIO_TB_MAX 3         // tight row tables per map (LioGrid::tb_*)
struct LioGrid {
    float ox, oy, oz;
    float inv_cell;
    int32_t nx, ny, nz;
    int32_t n_cells;
    int32_t k;          // neighbourhood radius in cells (cell edge = search radius / k)
    // The replicated neighbourhood rows are bucketed along x `xs` times finer than the cells (rows stay one cell high and
    // deep): a query's run [x-cell(qx - R), x-cell(qx + R)] is cut to a quarter of a cell instead of a whole one -- fewer
    // candidates for the same exact search (round 3; xs = 1 is the round-2 layout).
    int32_t xs;         // x subdivision of the row buckets (1, 2, 4, 8)
    int32_t nxf;        // nx * xs
    float inv_cell_x;   // inv_cell * xs (exact: xs is a power of two)
    // "Tight rows": up to LIO_TB_MAX further sets of replicated rows over the same points, each with k = 1 and a cell of
    // tb_reach[l] (+0.1 %), bucketed along x like the first and stored behind it in the same nbr_start / nbr_pts arrays,
    // tb_reach descending (0.6, 0.3, 0.15 of the gate radius).  A query whose search bound (the previous iteration's fifth
    // neighbour + its own movement) is at most tb_reach[l] walks its 3x3-cell row of THAT table, the tightest one that
    // holds its bound: at 0.6 of a 1 m gate a (1.8 m)^2 cross-section instead of (2.5 m)^2, at 0.3 (0.9 m)^2 -- the same
    // exact search over a fraction of the candidates.  tb_row0[l] = index of the table's first bucket in nbr_start;
    // tb_reach[l] < 0: no such table.
    int32_t tb_row0[LIO_TB_MAX], tb_ny[LIO_TB_MAX], tb_nz[LIO_TB_MAX];
    float tb_oy[LIO_TB_MAX], tb_oz[LIO_TB_MAX], tb_inv_cell[LIO_TB_MAX], tb_reach[LIO_TB_MAX];
    // Progressive search of a query WITHOUT a bound (first iteration; a point that failed the gate last time): on a map dense
    // enough that its fifth neighbour is almost surely within tb_reach[tb_try], that table is tried first with the
    // acceptance bound tb_try_b2[l] = (tb_reach[l] / 1.0002)^2 -- five points inside it prove that nothing outside the
    // table's row can be closer --, then the coarser tables, then the full search at the gate.  tb_try < 0: the full search
    // at once (sparse maps: nearly every wave would pay both).
    int32_t tb_try;
    float tb_try_b2[LIO_TB_MAX];
};

// Owner-computes predicate for a map sharded across GPUs: a scan
// point is processed by the rank owning the GLOBAL cell of its transformed
// position along `axis`.
struct LioShard {
    float gorigin;      // global grid origin along the axis
    float inv_cell;
    int32_t axis;       // 0,1,2 ; -1 = not sharded (own everything)
    int32_t gdim;       // global cell count along the axis
    int32_t lo, hi;     // owned cell range [lo, hi)
};

struct LioConsts {
    double plane_tol, weight, min_s, conv_deg, conv_cm;
    float  max_sq_dist, eig_thresh;
    int32_t min_corr, max_iters, jac_exact, force_all, record_iter, min_scan_pts;
};

// Per-scan Gauss-Newton state (transformTobeMapped MO:171, isDegenerate/matP
// MO:176-177, plus the transform of updatePointAssociateToMap MO:1613-1616 and
// the trig of LMOptimization MO:1714-1719, recomputed once per iteration by the
// lane that solves the step).
struct LioScanState {
    float pose[6];
    float T[12];
    float Tp[12];            // T of the previous GN iteration
    float trig[6];
    int32_t n_pts;
    int32_t offset;          // first point of this scan in the batch SoA
    int32_t c_n_pts;         // EXTENSION (corner residuals): the scan's edge points, in the corner batch SoA
    int32_t c_offset;
    int32_t iter;            // iterations executed
    int32_t done;
    int32_t converged;
    int32_t is_degenerate;
    int32_t n_corr_last;
    int32_t status;
    float matP[36];
    float AtA[36];
    float AtB[6];
    int32_t n_corr_iter[32];
    float pose_iter[32][6];
};

// One association workgroup = one contiguous chunk of one scan.
struct LioBlockDesc {
    int32_t scan;
    int32_t first;           // first point (index within the scan)
    int32_t blk;             // chunk index within the scan
    int32_t n_blk;           // chunks of this scan
};

// Scan-local tile grid used to re-order a scan at upload (4 m tiles by default).
struct LioScanTiles {
    float ox, oy, oz;
    float inv_tile;
    int32_t ntx, nty, ntz;
    int32_t key_offset;      // first tile key of this scan in the batch-wide key space
    int32_t mx, my, mz;      // key = key_offset + tx*mx + ty*my + tz*mz: x-fastest by default; with a sharded map the shard
                             // axis is the SLOWEST one, so that a workgroup's points form a thin slice across that axis
};
