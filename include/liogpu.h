/*
 * liogpu.h -- C ABI of the MI355X-native scan-to-map registration path.
 *
 * Drop-in boundary for the hot path of the reference's mapping node
 * (src/liorf/src/mapOptmization.cpp = MO, imageProjection.cpp = IP,
 * featureExtraction.cpp = FE, include/utility.h = UT).  The reference has no
 * FFI/plugin interface of its own; the seam is cut where the member-state
 * crossing is narrowest.  Plain pointers and sizes only; all
 * device memory is owned by the opaque handle.  Nothing here throws or aborts:
 * every entry point returns an int status (0 ok, >0 soft condition that the
 * reference also treats as "skip", <0 hard HIP/argument error).
 *
 * Point clouds cross the edge in the caller's layout: `stride_bytes` between
 * points, float x,y,z at byte offsets 0,4,8 (pcl::PointXYZI: stride 32, UT:65;
 * a packed float[3] array: stride 12).
 */
#ifndef LIOGPU_H
#define LIOGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIO_VERSION 102
#define LIO_MAX_ITERS 32

/* status codes */
enum {
    LIO_OK               = 0,
    LIO_TOO_FEW_POINTS   = 1,   /* N_s <= 30, MO:1844 / MO:1862-1864: pose unchanged     */
    LIO_TOO_FEW_CORR     = 2,   /* last iteration had < 50 correspondences, MO:1721-1724 */
    LIO_ERR_ARG          = -1,
    LIO_ERR_HIP          = -2,
    LIO_ERR_CAPACITY     = -3,
    LIO_ERR_NO_MAP       = -4,
    LIO_ERR_NO_DEVICE    = -5
};

/* Constants of the scan-to-map loop; the caller fills them from the untouched
 * ParamServer (UT:72-367) or keeps the defaults, which are the literals
 * hard-coded in MO. */
typedef struct lio_s2m_config {
    int32_t k;               /* 5     neighbours, MO:1631 (only 5 is supported)            */
    float   max_sq_dist;     /* 1.0   gate on the 5th neighbour, MO:1641                   */
    double  plane_tol;       /* 0.2   MO:1662 (double literal in the reference)            */
    double  weight;          /* 0.9   MO:1671 (double literal)                             */
    double  min_s;           /* 0.1   MO:1679 (double literal)                             */
    int32_t min_corr;        /* 50    MO:1722                                              */
    int32_t max_iters;       /* 30    MO:1848 (<= LIO_MAX_ITERS)                           */
    float   eig_thresh;      /* 100   MO:1796                                              */
    double  conv_deg;        /* 0.05  MO:1833                                              */
    double  conv_cm;         /* 0.05  MO:1833                                              */
    int32_t min_scan_pts;    /* 30    MO:1844 (N_s must exceed it)                         */
    int32_t jacobian_mode;   /* 0 = reference (MO:1764 as written), 1 = exact derivative   */
    int32_t force_all_iters; /* 1 = ignore the convergence break MO:1857-1858              */
    int32_t device_id;       /* HIP device ordinal                                         */
    float   cell_size;       /* search radius covered by the grid neighbourhood, metres;
                                0 = sqrt(max_sq_dist)*1.001.  Cell edge = cell_size/cell_div */
    int32_t max_batch;       /* hint: scans per batch (buffers grow on demand; may be 0); also decides x_sub = auto */
    int32_t max_scan_pts;    /* hint: points per scan (buffers grow on demand; may be 0)    */
    int32_t record_corr_iter;/* iteration whose correspondences are kept for
                                lio_s2m_get_correspondences (-1 = none)                    */
    int32_t kernel_variant;  /* scan points per thread: 0 = auto (1), or 1 / 2 / 4 (A/B testing)   */
    int32_t profile;         /* 1 = bracket every GN-iteration launch with HIP events      */
    int32_t lookahead;       /* GN launches enqueued ahead of the convergence check;
                                0 = never enqueue an empty launch, -1 = auto               */
    int32_t use_lds;         /* 1 = stage the workgroup's map region through LDS (measured 2.5x slower);
                                0 (default) = stream the replicated neighbourhood rows     */
    int32_t sort_scan;       /* 1 = re-order scans by tiles at upload when the batch has >= 65536
                                points (default), 2 = always, 0 = never, 3 = always and by the
                                multi-kernel counting sort even when every scan fits the one-launch
                                LDS sort (A/B and tests); results are reported in the caller's
                                order either way                                            */
    int32_t cell_div;        /* k: cells per search radius (1..3, default 2); the candidate
                                scan visits (2k+1)^3 cells, map rows are replicated (2k+1)^2 x */
    int32_t xcd_remap;       /* 1 (default) = XCD-aware workgroup order (L2 locality only)  */
    float   tile_size;       /* edge of the upload-time sort tiles in metres (0 = 4 m)      */
    int32_t use_graph;       /* 1 = replay a hipGraph of `graph_iters` captured GN iterations
                                per unit of the launch loop (BASELINE config 5); ignored when
                                record_corr_iter or the diagnostic profile=2 is set        */
    int32_t graph_iters;     /* iterations per captured chunk (default 4)                   */
    int32_t sort_batch;      /* 1 (default) = order the workgroups of a batch by the scans'
                                positions (L2 locality only; results are unaffected)       */
    int32_t nn_cache;        /* 1 (default) = from the second GN iteration on, bound each point's
                                search by the distances to its previous 5 neighbours (exact:
                                the candidate run shrinks, the result does not change)      */
    int32_t pipeline;        /* how the Gauss-Newton loop MO:1848-1859 is issued; results are identical for every value.
                                0 = auto: one launch per iteration (k_s2m_iterate), or -- for a batch of at most half
                                a workgroup per compute unit, e.g. a lone registration, unless use_graph or profile = 2
                                is set -- the whole loop as ONE launch (k_s2m_persist: per-scan barrier between
                                iterations; 0.17 against 0.25 ms per registration on MI355X); 1 = always one launch per
                                iteration; 4 = the one-launch loop for every batch of at most one workgroup per compute
                                unit.  Other values are refused (2, 3 and 5 were per-point-cache experiments of round 2,
                                measured slower or equal and removed: DESIGN.md appendix)                                */
    int32_t n_devices;       /* 1 (default) = the single device `device_id`.  > 1: in-library multi-GPU -- the local
                                map is cut into slabs (+ a halo of 16 cells, i.e. ~16 m of extra map per side and
                                device) over device_ids[0..n_devices), every
                                registration's points are processed by the device owning their map cell and the
                                6x6 JtJ / 6x1 Jtr / N_c are summed across the devices once per GN iteration
                                (the join of the OpenMP loop MO:1622-1686); set_map / register / batch_* work
                                unchanged on such a handle, so a patched node needs no other change              */
    int32_t device_ids[8];   /* HIP device ordinals of the multi-GPU mode (an ordinal may repeat)               */
    int32_t x_sub;           /* the replicated neighbourhood rows are bucketed along x this many times finer than the
                                cells (1, 2, 4, 8): a query's candidate run is cut to [x(q - R), x(q + R)] at that
                                resolution -- 20 % fewer candidates at 4, -6 % per launch, +5 % registrations/s on the
                                512-scan batches (same results: the search stays exact) -- at the price of a 4x longer
                                bucket table, +0.08 ms per map build.  0 = auto: 4 for a handle set up for batches
                                (max_batch >= 8: the map is built once and searched by many scans), 1 for a node's handle
                                (the map is rebuilt for every scan; a lone registration is latency-bound and gains nothing) */
    int32_t tight_rows;      /* further sets of neighbourhood rows over the same points with 3x3 cells of 0.6 / 0.3 / 0.15 x
                                the gate radius (+36 % map memory and build time each): a query whose search bound -- the
                                previous iteration's fifth neighbour plus its own movement -- is at most a table's cell walks
                                that table's row: a (1.8 m)^2 / (0.9 m)^2 / (0.45 m)^2 cross-section instead of (2.5 m)^2 at
                                the 1 m gate; same results.  -3 % per launch on the 512-scan batches of the default 0.5 m map
                                (one table), 2.1x the registrations/s on a 0.1 m map (three).  1..3 = that many tables, -1 =
                                none, 0 = auto: none for a node's handle (max_batch < 8), else one table plus the finer ones
                                the map's point density has queries for (lio_s2m_profile.map_tight_tables says how many)  */
} lio_s2m_config;

/* What scan2MapOptimization leaves behind (MO:1817-1822 pose is returned in
 * place; MO:176-177 isDegenerate/matP; the rest is diagnostics). */
typedef struct lio_s2m_result {
    int32_t status;
    int32_t iters;             /* loop bodies executed (<= max_iters)                    */
    int32_t converged;         /* LMOptimization returned true, MO:1833-1835             */
    int32_t is_degenerate;     /* MO:176 -> odometry_incremental covariance[0], MO:2309  */
    int32_t n_corr_last;       /* N_c of the last executed iteration                     */
    int32_t n_corr_iter[LIO_MAX_ITERS];
    float   matP[36];          /* row-major, MO:177/1807                                 */
    float   AtA[36];           /* last normal matrix, row-major, MO:1782                 */
    float   AtB[6];            /* MO:1783                                                */
    float   pose_iter[LIO_MAX_ITERS][6]; /* pose after each iteration                    */
} lio_s2m_result;

typedef struct lio_s2m_profile {
    float   map_build_ms;      /* last set_map: H2D excluded, grid build kernels only     */
    float   map_upload_ms;     /* last set_map: wall time of staging + H2D                */
    int32_t n_launches;        /* GN-iteration launches of the last run                   */
    int32_t n_units;           /* launch units: single launches, or graph replays          */
    int32_t unit_iters;        /* launches per unit (1, or cfg.graph_iters)                */
    float   launch_ms[LIO_MAX_ITERS]; /* device time of each unit (profile=1)            */
    int32_t launch_active[LIO_MAX_ITERS]; /* scans in the first launch of each unit       */
    int64_t point_iters;       /* scan points processed by active scans over the run      */
    int64_t n_map;             /* resident map points                                     */
    int64_t n_cells;           /* grid cells                                              */
    int32_t pipeline;          /* what the last run used: 1 = one launch per iteration, 4 = one-launch loop */
    int32_t multi_iterations;  /* in-library multi-GPU mode (cfg.n_devices > 1), last run: GN iterations enqueued,        */
    int32_t multi_stream_syncs;/* ... stream / device synchronisations issued inside the loop (0 with the device-side
                                  exchange) and                                                                        */
    int32_t multi_event_waits; /* ... waits for a convergence count (an event wait, `lookahead` iterations behind)      */
    int32_t multi_exchange;    /* ... 0 = peer stores from a kernel, 1 = hipMemcpyPeerAsync, 2 = through host memory    */
    int32_t persist_fallbacks; /* one-launch loops of this handle that timed out at a barrier and were re-run through the
                                  launch loop inside the same call (cumulative; see lio_s2m_batch_results)              */
    int32_t map_x_sub;         /* last set_map: x subdivision of the row buckets in effect (cfg.x_sub)                  */
    int32_t map_tight_tables;  /* ... tight row tables built (cfg.tight_rows)                                            */
    int32_t map_first_try;     /* ... the tight table a query without a search bound tries before the full search (its
                                  index, 0 = the widest; -1: none -- the full search at once)                            */
    float   map_pts_per_cell;  /* ... map points per occupied grid cell, the density estimate behind the automatic choice
                                  of map_tight_tables (0: not measured)                                                  */
} lio_s2m_profile;

typedef struct lio_s2m_handle lio_s2m_handle;

int  lio_version(void);
void lio_s2m_default_config(lio_s2m_config *cfg);
const char *lio_last_error(void);

int  lio_s2m_create(const lio_s2m_config *cfg, lio_s2m_handle **out);
void lio_s2m_destroy(lio_s2m_handle *h);

/* Replaces kdtreeSurfFromMap->setInputCloud(laserCloudSurfFromMapDS), MO:1846:
 * copies the local map to the device as SoA and builds the hash grid. */
int  lio_s2m_set_map(lio_s2m_handle *h, const void *pts, size_t n, size_t stride_bytes);

/* Replaces the loop MO:1848-1859 for one scan (laserCloudSurfLastDS, MO:138).
 * pose = transformTobeMapped [roll,pitch,yaw,x,y,z] (MO:171), in/out.
 * matP / isDegenerate persist on the handle between calls like the members
 * MO:176-177.  Synchronous. */
int  lio_s2m_register(lio_s2m_handle *h, const void *scan, size_t n, size_t stride_bytes,
                      float pose[6], lio_s2m_result *res);

/* Batched form (BASELINE config 5): many scans against the same resident map,
 * every scan its own pose / matP / convergence.  upload -> set_poses -> run
 * (asynchronous on the handle's stream) -> results (synchronises). */
int  lio_s2m_batch_upload(lio_s2m_handle *h, int32_t n_scans, const void *const *scans,
                          const size_t *n_pts, size_t stride_bytes);
int  lio_s2m_batch_set_poses(lio_s2m_handle *h, const float *poses /* n_scans x 6 */);
int  lio_s2m_batch_run(lio_s2m_handle *h);
int  lio_s2m_batch_sync(lio_s2m_handle *h);
int  lio_s2m_batch_results(lio_s2m_handle *h, float *poses /* n_scans x 6 */,
                           lio_s2m_result *results /* n_scans, may be NULL */);

/* ---- streaming: upload batch k+1 while batch k iterates (SURVEY 8d puts the per-scan H2D inside the metric;
 * the reference analogue is one cloud_info message arriving per callback, MO:432-476).
 * lio_s2m_share_map: handle `h` searches `map_owner`'s resident map (no copy); it keeps its own HIP stream,
 * scan buffers and per-scan state, so several handles form a software pipeline -- two a double buffer:
 *     run(A) ... upload_async(B, next batch) ... results(A) ... run(B) ... upload_async(A, ...) ...
 * THREE (two batches in flight ahead of the one whose results are awaited) measured 16 % faster than two: the host's
 * preparation of the next upload then never leaves the GPU alone with the last, nearly empty launches of a batch
 * (more than three streams oversubscribe the runtime's four hardware queues).
 * upload_async returns once the scans' H2D copies are DONE on h's stream (the caller's buffers are free
 * again) while the tile sort is still in flight; set_poses/run/results queue behind it.  Copies are true DMA
 * only from pinned memory: lio_host_alloc / lio_host_register (hipHostMalloc / hipHostRegister).  A batch laid
 * out contiguously (scans[s+1] == scans[s] + n_pts[s]*stride) goes in one copy.  scans[] may also be DEVICE
 * pointers (inputs already resident in HBM, e.g. the output of lio_deskew kept on the device).
 * The owner must outlive the sharer; call lio_s2m_set_map on the owner only while both streams are idle. */
int   lio_s2m_share_map(lio_s2m_handle *h, lio_s2m_handle *map_owner);
int   lio_s2m_batch_upload_async(lio_s2m_handle *h, int32_t n_scans, const void *const *scans,
                                 const size_t *n_pts, size_t stride_bytes);
void *lio_host_alloc(size_t bytes);
void  lio_host_free(void *p);
int   lio_host_register(void *p, size_t bytes);
int   lio_host_unregister(void *p);
/* Device memory for callers that keep clouds in HBM between calls (memory of the handle's own device is read in place by
 * lio_s2m_batch_upload*, lio_s2m_register_raw, lio_kf_store_add_device).  lio_device_upload is a synchronous H2D copy. */
void *lio_device_alloc(int32_t device_id, size_t bytes);
void  lio_device_free(int32_t device_id, void *p);
int   lio_device_upload(int32_t device_id, void *dst, const void *src, size_t bytes);

/* Persistent members MO:176-177 for batch slot `scan` (slot 0 = lio_s2m_register). */
int  lio_s2m_set_degeneracy(lio_s2m_handle *h, int32_t scan, const float matP[36], int32_t is_degenerate);

/* Association of iteration cfg.record_corr_iter for batch slot `scan`
 * (laserCloudOriSurfFlag / coeffSelSurfVec, MO:143-145, plus the 5 neighbour
 * indices into the caller's map order).  Arrays sized by that scan's N_s. */
int  lio_s2m_get_correspondences(lio_s2m_handle *h, int32_t scan, uint8_t *flag,
                                 float *coeff4, int32_t *nn_idx5);

/* ---- EXTENSION beyond this reference: point-to-line ("corner") residuals ------------------
 * BASELINE.json's north_star names cornerOptimization, which this fork removed (its
 * scan2MapOptimization MO:1839-1865 is surf-only).  These entry points
 * add the cornerOptimization of upstream LIO-SAM (kdtreeCornerFromMap 5-NN, 3x3 covariance,
 * cv::eigen, lambda0 > 3*lambda1, point-to-line distance) to the same Gauss-Newton loop:
 * corner rows are appended to the surf rows exactly as upstream's combineOptimizationCoeffs
 * does, and LMOptimization MO:1702-1837 is shared.  There is no reference oracle for them
 * in /root/reference (parity unpinned: checked against oracle/lio_oracle.c lo_scan2map_cs
 * only).  Without a corner batch every other entry point behaves exactly as before.
 *
 * set_corner_map          : kdtreeCornerFromMap->setInputCloud(laserCloudCornerFromMapDS)
 * batch_upload_corners    : the scans' edge points (laserCloudCornerLastDS); call after
 *                           lio_s2m_batch_upload with the same n_scans and before
 *                           lio_s2m_batch_set_poses; n_pts[s] may be 0
 * register_cs             : lio_s2m_register with both clouds
 * get_corner_correspondences : as lio_s2m_get_correspondences, for the edge points */
int  lio_s2m_set_corner_map(lio_s2m_handle *h, const void *pts, size_t n, size_t stride_bytes);
int  lio_s2m_batch_upload_corners(lio_s2m_handle *h, int32_t n_scans, const void *const *scans,
                                  const size_t *n_pts, size_t stride_bytes);
int  lio_s2m_register_cs(lio_s2m_handle *h, const void *corner_scan, size_t n_corner,
                         const void *surf_scan, size_t n_surf, size_t stride_bytes,
                         float pose[6], lio_s2m_result *res);
int  lio_s2m_get_corner_correspondences(lio_s2m_handle *h, int32_t scan, uint8_t *flag,
                                        float *coeff4, int32_t *nn_idx5);

int  lio_s2m_get_profile(lio_s2m_handle *h, lio_s2m_profile *out);
/* Test hook for the one-launch loop (cfg.pipeline 0 / 4): spin_max = polls before a workgroup waiting at its scan's barrier
 * gives up (0 = default: 4096, a few milliseconds; LIO_PERSIST_SPIN_MAX in the environment), withhold_wg = index of an
 * association workgroup that never arrives (-1 = none).  A launch that times out is re-run through the launch loop inside
 * lio_s2m_batch_results / lio_s2m_register and counted in lio_s2m_profile.persist_fallbacks; results are the same. */
int  lio_s2m_debug_persist_spin(lio_s2m_handle *h, int32_t spin_max, int32_t withhold_wg);
/* Diagnostic (cfg.profile == 2): per-wave phase clock of the last GN launch,
 * n_blocks x 4 x 8 cycle counters; returns n_blocks.  Not for production. */
int  lio_s2m_debug_stamps(lio_s2m_handle *h, long long *out, size_t cap_entries);

/* Multi-GPU hooks (one process per GPU; the caller owns the collective).
 * The map given to set_map is this rank's shard INCLUDING a halo of one cell;
 * lio_s2m_set_shard restricts which transformed scan points this rank owns
 * (owner-computes): cells whose index along `axis` lies in [lo, hi) of the
 * GLOBAL grid described by origin/dims.  iter_partial leaves per-scan sums
 * (n_scans x 32 doubles: 21 upper JtJ, 6 Jtr, N_c, pad) in a device buffer;
 * the caller all-reduces it and iter_apply solves + updates every scan. */
int  lio_s2m_set_stream(lio_s2m_handle *h, void *hip_stream);
int  lio_s2m_set_global_grid(lio_s2m_handle *h, const float origin[3], const int32_t dims[3]);
int  lio_s2m_set_shard(lio_s2m_handle *h, int32_t axis, int32_t lo, int32_t hi);
/* The whole slab plan instead of this rank's range: rank r of n_ranks owns cells [bounds[r], bounds[r+1]) and holds the
 * map of [bounds[r] - halo_cells, bounds[r+1] + halo_cells).  With halo_cells > 1 a workgroup of scan points (256
 * consecutive points of the tile-sorted scan) whose extent fits the held region of the rank owning its middle is
 * processed WHOLLY by that rank and skipped by all others; longer workgroups keep per-point ownership.  Same results;
 * no wave runs for a handful of owned lanes.  Every rank must be given identical bounds and halo. */
int  lio_s2m_set_shard_plan(lio_s2m_handle *h, int32_t axis, int32_t n_ranks, int32_t rank, const int32_t *bounds,
                            int32_t halo_cells);
/* Alternative partition: the map is replicated, every rank processes each world-th
 * workgroup of every scan (call before lio_s2m_batch_upload); same iter_partial / all-reduce /
 * iter_apply protocol.  No halo, no ownership tests, perfectly balanced. */
int  lio_s2m_set_scan_shard(lio_s2m_handle *h, int32_t rank, int32_t world);
int  lio_s2m_batch_begin(lio_s2m_handle *h);
int  lio_s2m_batch_iter_partial(lio_s2m_handle *h, double *d_sums /* device, n_scans x 32 */);
int  lio_s2m_batch_iter_apply(lio_s2m_handle *h, const double *d_sums);
int  lio_s2m_batch_n_active(lio_s2m_handle *h, int32_t *n_active);
/* Scans still iterating after applied iteration `iteration` (0-based); waits only for that
 * iteration, so the caller can keep later iterations enqueued. */
int  lio_s2m_batch_poll_active(lio_s2m_handle *h, int32_t iteration, int32_t *n_active);

/* transformUpdate + constraintTransformation, MO:1867-1907 (host, fp64 slerp). */
void lio_transform_update(float pose[6], int32_t imu_available, int32_t imu_type,
                          float imu_roll_init, float imu_pitch_init, float imu_rpy_weight,
                          float rotation_tollerance, float z_tollerance);

/* ------------------------------------------------------------------ deskew */
/* Filters of projectPointCloud IP:577-615, keys UT:275-285. */
typedef struct lio_deskew_config {
    int32_t N_SCAN;
    int32_t downsampleRate;
    int32_t point_filter_num;
    float   lidarMinFront, lidarMinBack, lidarMinLeft, lidarMinRight;
    float   lidarMaxRange;
    float   lidarMaxIntensity;
    int32_t deskew_flag;      /* -1 = cloud has no per-point time field, IP:547 */
    int32_t device_id;
} lio_deskew_config;

void lio_deskew_default_config(lio_deskew_config *cfg);

/* imuDeskewInfo, IP:359-418 (host, fp64): integrates the gyro over the sweep.
 * Tables have room for 2000 entries (queueLength, IP:62).  Returns
 * imuPointerCur (> 0 <=> cloudInfo.imuAvailable). */
int  lio_imu_deskew_info(const double *stamp, const double *gyro_x, const double *gyro_y,
                         const double *gyro_z, int32_t n_imu,
                         double time_scan_cur, double time_scan_end,
                         double *imuTime, double *imuRotX, double *imuRotY, double *imuRotZ);

/* projectPointCloud + deskewPoint, IP:545-615, on the device.
 * pts: PointXYZIRT records (IP:4-15): float x,y,z @0,4,8; float intensity @16;
 * uint16 ring @20; float time @24; stride 32.  out: pcl::PointXYZI-compatible
 * records (x,y,z @0,4,8, intensity @16) with `out_stride_bytes` (>= 20),
 * input order preserved.  Returns status; *n_out = survivors. */
int  lio_deskew(const lio_deskew_config *cfg, const void *pts, size_t n, size_t stride_bytes,
                double time_scan_cur,
                const double *imuTime, const double *imuRotX, const double *imuRotY,
                const double *imuRotZ, int32_t imuPointerCur,
                void *out, size_t out_stride_bytes, size_t *n_out);

/* ---- cloud_info wire path: sensor_msgs/PointCloud2 blobs read in place ----------------------
 * The reference moves every cloud through pcl::fromROSMsg / pcl::moveFromROSMsg (MO:440, IP:226-232) and
 * pcl::toROSMsg (publishCloud UT:369-379); cloud_info.cloud_deskewed is such a blob (cloud_info.msg:27).  These
 * entry points take the message's `data` pointer, `width*height` and the byte offsets of its `fields` directly.
 * Constraints: little-endian (is_bigendian == 0); x, y, z are three consecutive FLOAT32 fields; intensity FLOAT32.
 * ring / time variants cover the four sensor layouts cachePointCloud converts on the host (IP:226-285). */
enum { LIO_PC2_UINT8 = 2, LIO_PC2_UINT16 = 4, LIO_PC2_INT32 = 5 };            /* sensor_msgs/PointField datatype codes */
enum { LIO_PC2_TIME_F32_SECONDS = 0,   /* Velodyne / Livox "time": float seconds from the sweep start          */
       LIO_PC2_TIME_U32_NS      = 1,   /* Ouster "t": dst.time = src.t * 1e-9f, IP:243                          */
       LIO_PC2_TIME_U32_RAW     = 2,   /* Mulran "t": dst.time = static_cast<float>(src.t), IP:262              */
       LIO_PC2_TIME_F64_STAMP   = 3 }; /* Robosense "timestamp": src.timestamp - points[0].timestamp, IP:269-281 */
typedef struct lio_pc2_layout {
    uint32_t point_step;      /* bytes per point (PointCloud2.point_step)                               */
    uint32_t off_x;           /* byte offset of x; y and z follow at +4, +8                              */
    int32_t  off_intensity;   /* FLOAT32, -1 = no such field (intensity reads as 0)                       */
    int32_t  off_ring;        /* deskew only; the reference refuses clouds without it, IP:313-329         */
    int32_t  ring_type;       /* LIO_PC2_UINT8 / LIO_PC2_UINT16 / LIO_PC2_INT32                           */
    int32_t  off_time;        /* deskew only; -1 = no per-point time (deskewFlag = -1, IP:341-356)        */
    int32_t  time_type;       /* LIO_PC2_TIME_*                                                           */
    int32_t  pin_host;        /* 1 = hipHostRegister the blob for the duration of the call (the caller
                                 allows its pages to be pinned): the H2D copy is then a true DMA         */
} lio_pc2_layout;

/* lio_s2m_register on the blob of cloud_info.cloud_deskewed (or any PointCloud2 with xyz): replaces
 * pcl::fromROSMsg MO:440 + the loop MO:1848-1859.  Only off_x / point_step / pin_host of the layout are used. */
int  lio_s2m_register_pc2(lio_s2m_handle *h, const void *data, size_t n_points, const lio_pc2_layout *layout,
                          float pose[6], lio_s2m_result *res);
/* One mapping callback on the device: downsampleCurrentScan MO:1605-1611 (`downSizeFilterSurf.filter`, leaf =
 * mappingSurfLeafSize) + scan2MapOptimization MO:1839-1865 on the blob of cloud_info.cloud_deskewed (MO:440), without a host
 * round trip in between: one H2D copy of the blob (true DMA with layout->pin_host or lio_host_alloc memory; none when
 * `data` already is memory of the handle's device), the voxel filter on the handle's stream with its workspace kept on
 * the handle, the Gauss-Newton loop on the filter's output where it lies.  Bit-identical to lio_voxel_grid followed by
 * lio_s2m_register on its output (including PCL's pass-through when the leaf overflows the voxel index: the function
 * then still registers the unfiltered cloud, as the reference does).  off_x / off_intensity / point_step / pin_host of the
 * layout are used.  ds_out (may be NULL): receives laserCloudSurfLastDS as PointXYZI-compatible records, room for
 * n_points; *n_ds (may be NULL) = its size.  The filtered cloud stays staged on the handle: lio_kf_store_add_from_handle
 * (h, 0) makes it a keyframe (MO:2136-2142) without a copy through the host. */
int  lio_s2m_register_raw(lio_s2m_handle *h, const void *data, size_t n_points, const lio_pc2_layout *layout, float leaf,
                          float pose[6], lio_s2m_result *res, void *ds_out, size_t ds_out_stride, size_t *n_ds);
/* lio_deskew on the raw driver message (IP:214-232 + IP:577-615).  out: PointXYZI-compatible records as for lio_deskew. */
int  lio_deskew_pc2(const lio_deskew_config *cfg, const void *data, size_t n_points, const lio_pc2_layout *layout,
                    double time_scan_cur,
                    const double *imuTime, const double *imuRotX, const double *imuRotY,
                    const double *imuRotZ, int32_t imuPointerCur,
                    void *out, size_t out_stride_bytes, size_t *n_out);

/* ---- EXTENSION beyond this reference: the range-image build -------------------------------
 * BASELINE.json's north_star names "imageProjection's deskew/range-image build"; this fork's
 * projectPointCloud IP:577-615 no longer builds one and nothing in the fork fills
 * startRingIndex / endRingIndex / pointColInd / pointRange (MSG:4-8), which its own
 * featureExtraction.cpp reads.  lio_range_image is projectPointCloud +
 * cloudExtraction of upstream LIO-SAM (Velodyne/Ouster column rule, first point per range-image
 * cell wins) with this fork's deskewPoint IP:545-575, i.e. the producer lio_extract_features
 * needs.  No reference oracle exists for it (parity unpinned; checked against
 * oracle/lio_oracle.c lo_range_image only).
 * pts: PointXYZIRT records as for lio_deskew.  out: PointXYZI-compatible records, room for
 * N_SCAN * Horizon_SCAN; pointColInd / pointRange the same; startRingIndex / endRingIndex:
 * N_SCAN entries.  *n_out = points kept (ring-major, ascending column). */
typedef struct lio_range_image_config {
    int32_t N_SCAN, Horizon_SCAN, downsampleRate;
    float   lidarMinRange, lidarMaxRange;
    int32_t deskew_flag;     /* -1 = no per-point time field */
    int32_t device_id;
} lio_range_image_config;
void lio_range_image_default_config(lio_range_image_config *cfg);
int  lio_range_image(const lio_range_image_config *cfg, const void *pts, size_t n, size_t stride_bytes,
                     double time_scan_cur,
                     const double *imuTime, const double *imuRotX, const double *imuRotY,
                     const double *imuRotZ, int32_t imuPointerCur,
                     void *out, size_t out_stride_bytes, size_t *n_out,
                     int32_t *startRingIndex, int32_t *endRingIndex,
                     int32_t *pointColInd, float *pointRange);

/* calculateSmoothness, FE:81-101: curvature[i] for i in [5, n-5), also zeroes
 * neighbor_picked / label there (either may be NULL).  Host pointers. */
int  lio_curvature(int32_t device_id, const float *range, size_t n, float *curvature,
                   int32_t *neighbor_picked, int32_t *label);

/* The rest of FeatureExtraction::laserCloudInfoHandler FE:67-79: calculateSmoothness FE:81-101,
 * markOccludedPoints FE:103-139 and extractFeatures FE:141-238 on the device, consuming the
 * cloud_info arrays (MSG:4-8) as the reference does.
 * cloud: extractedCloud records (x,y,z @0,4,8, intensity @16), n points, ring-major.
 * startRingIndex/endRingIndex: cfg->N_SCAN entries.  Rings must own disjoint, ascending index
 * windows [start-5, end+4] (what upstream's cloudExtraction produces); at most 4086 points per
 * ring and 1024 per sector; columns must fit int16.
 * corner_out: room for 120 * N_SCAN records (FE:171: <= 20 per sector); surface_out: room for n.
 * curvature / neighbor_picked / label (each n entries, may be NULL) receive cloudCurvature,
 * cloudNeighborPicked and cloudLabel as the reference leaves them after the handler.
 * Defined here where the reference is not (see oracle/lio_oracle.c lo_extract_features): the
 * flag arrays start from zero for every scan, equal curvatures keep ascending point index
 * (std::sort FE:162 leaves their order open), and a suppression walk FE:178-193 stops at the
 * array boundary instead of indexing pointColInd[-1]. */
typedef struct lio_feature_config {
    int32_t N_SCAN;          /* UT:164 */
    float   edgeThreshold;   /* UT:186, 1.0 in the yaml configs */
    float   surfThreshold;   /* UT:187, 0.1 */
    float   surfLeafSize;    /* mappingSurfLeafSize FE:56 */
    int32_t device_id;
} lio_feature_config;
void lio_feature_default_config(lio_feature_config *cfg);
int  lio_extract_features(const lio_feature_config *cfg, const void *cloud, size_t n, size_t stride_bytes,
                          const int32_t *startRingIndex, const int32_t *endRingIndex,
                          const int32_t *pointColInd, const float *pointRange,
                          void *corner_out, size_t *n_corner, void *surface_out, size_t *n_surface,
                          size_t out_stride_bytes, float *curvature, int32_t *neighbor_picked, int32_t *label);

/* ------------------------------------------------ local-map assembly (feeders) */
/* pcl::VoxelGrid<PointXYZI>::filter as used by downsampleCurrentScan MO:1605-1611
 * (downSizeFilterSurf, leaf = mappingSurfLeafSize) and MO:1581-1583: centroid per
 * voxel (x, y, z and intensity), output in ascending voxel index.  Records: x,y,z
 * @0,4,8 and intensity @16.  Returns LIO_OK, or 1 when PCL would pass the cloud
 * through unfiltered ("Leaf size is too small", voxel index overflow). */
int  lio_voxel_grid(int32_t device_id, const void *pts, size_t n, size_t stride_bytes, float leaf,
                    void *out, size_t out_stride_bytes, size_t *n_out);

/* extractCloud MO:1556-1588: laserCloudSurfFromMap = sum over the nearby keyframes of
 * transformPointCloud(surfCloudKeyFrames[i], cloudKeyPoses6D[i]) (MO:849-868), then
 * VoxelGrid(surroundingKeyframeMapLeafSize) -> laserCloudSurfFromMapDS.  poses are
 * [roll,pitch,yaw,x,y,z] per keyframe.  If h != NULL the result becomes h's resident map
 * (replaces lio_s2m_set_map, no round trip through the host); `out` (may be NULL) receives
 * the downsampled map as PointXYZI-compatible records. */
int  lio_assemble_map(lio_s2m_handle *h, int32_t device_id, int32_t n_keyframes, const void *const *clouds,
                      const size_t *n_pts, size_t stride_bytes, const float *poses, float leaf,
                      void *out, size_t out_stride_bytes, size_t *n_out);

/* Device-resident keyframe store = surfCloudKeyFrames (MO:128): every keyframe cloud is
 * uploaded once when it is saved (MO:2138-2142); lio_assemble_map_resident then builds the local
 * map of a scan from keyframe ids + their current poses (cloudKeyPoses6D) without touching the
 * host clouds again.  ids index the store in insertion order. */
typedef struct lio_kf_store lio_kf_store;
int  lio_kf_store_create(int32_t device_id, lio_kf_store **out);
void lio_kf_store_destroy(lio_kf_store *s);
int  lio_kf_store_add(lio_kf_store *s, const void *cloud, size_t n, size_t stride_bytes, int32_t *id_out);
int  lio_kf_store_count(const lio_kf_store *s);
/* Points of keyframe `id` (0 for an id the store does not hold): what a caller adds up to size `out` of
 * lio_assemble_map_resident. */
size_t lio_kf_store_points(const lio_kf_store *s, int32_t id);
/* The same from DEVICE memory (x,y,z @0,4,8, intensity @16 when stride >= 20, else 0): no host round trip. */
int  lio_kf_store_add_device(lio_kf_store *s, const void *d_cloud, size_t n, size_t stride_bytes, int32_t *id_out);
/* saveKeyFramesAndFactor MO:2136-2142 (`pcl::copyPointCloud(*laserCloudSurfLastDS, *thisSurfKeyFrame);
 * surfCloudKeyFrames.push_back(thisSurfKeyFrame)`): appends the scan that batch slot `scan` of `h` has just
 * registered (its records are still staged on the device) as a keyframe -- no D2H, no H2D. */
int  lio_kf_store_add_from_handle(lio_kf_store *s, lio_s2m_handle *h, int32_t scan, int32_t *id_out);
/* `out` (may be NULL) needs room for the SUM of the selected keyframes' points: the filter's output is at most that many,
 * and exactly that many when the leaf overflows PCL's voxel index (or the clouds hold non-finite coordinates) and the
 * input is passed through, as pcl::VoxelGrid does.  The same holds for `out` of lio_assemble_map (sum of n_pts). */
int  lio_assemble_map_resident(lio_s2m_handle *h, lio_kf_store *s, int32_t n_selected, const int32_t *ids,
                               const float *poses, float leaf, void *out, size_t out_stride_bytes, size_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* LIOGPU_H */
