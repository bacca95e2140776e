// lio_s2m_device.h -- device code shared by the Gauss-Newton kernels of the scan-to-map path:
// k_s2m_iterate (lio_kernels.hip, one fused launch per iteration) and k_s2m_persist (lio_persist.hip, the whole loop in one launch).  MO = /root/reference/src/liorf/src/mapOptmization.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include "lio_types.h"
#include "lio_device_math.h"
#include "lio_kernels.h"

// Map coordinates beyond this magnitude (or non-finite) are left out of the grid:
// fp32 spacing there is > 1e7 m, they cannot be a neighbour within 1 m of anything,
// and keeping them out keeps every squared distance finite.
#define LIO_MAX_COORD 1.0e15f
#ifndef LIO_PREFETCH
#define LIO_PREFETCH 1        // candidate groups in flight ahead of the one being evaluated (1 or 2).  Measured: 2 is 10 % SLOWER at the
                              // same occupancy (0.144 vs 0.131 ms/launch) -- the loop is bound by the 64 B/clk L1 delivery of
                              // 1 KiB per wave-wide 16-byte load, not by latency; more loads in flight only queue up.
#endif
#ifndef LIO_ARRIVE_ACQREL
#define LIO_ARRIVE_ACQREL 0
#endif
#define LIO_IDX_MASK 0x1fffffff      // index carried by the dummy records that pad the neighbourhood rows (never a winner)

// ------------------------------------------------------------------ helpers
LIO_DEV int lio_cell_coord(float v, float origin, float inv_cell, int n)
{
    // monotone in v; clamped to [-4, n+3] so that NaN / far-away queries land
    // outside every neighbourhood instead of overflowing the int conversion
    float c = floorf((v - origin) * inv_cell);
    c = fminf(fmaxf(c, -4.0f), (float)(n + 3));
    return (int)c;
}

// -------------------------------------------------------------- GN iterate
// top-5 keys in the (d2, index) order of the exact k-NN (pcl::KdTreeFLANN::nearestKSearch MO:1631: ascending
// squared distance; ties by the smaller map index), kept sorted with v_min_f64 / v_max_f64 only.
LIO_DEV double lio_dmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
LIO_DEV double lio_dmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Key = the bit pattern (d2 as fp32 in the high word, map index in the low word) read as a double.
// d2 >= 0 and never NaN (map records and queries are finite, see the `act` test), so the high word is a
// non-negative fp32 pattern: its top 11 bits are never 0x7ff, i.e. the double is finite, and finite
// non-negative doubles order exactly like their bit patterns -- numeric order of the keys =
// lexicographic (d2, index).  No conversion instruction, no index-width limit; d2 = 0 gives a
// denormal double, which v_min_f64 / v_max_f64 keep (fp64 denormals are on in the kernel's FP mode).
LIO_DEV double lio_make_key(float d2, int idx) { return __hiloint2double(__float_as_int(d2), idx); }
LIO_DEV float lio_key_d2(double key) { return __int_as_float(__double2hiint(key)); }   // the squared distance, exactly
LIO_DEV int lio_key_idx(double key) { return __double2loint(key); }

struct LioTop5 { double k0, k1, k2, k3, k4; };

LIO_DEV void lio_top5_insert(LioTop5& t, double x)
{
    double c;
    c = lio_dmax(t.k0, x); t.k0 = lio_dmin(t.k0, x); x = c;
    c = lio_dmax(t.k1, x); t.k1 = lio_dmin(t.k1, x); x = c;
    c = lio_dmax(t.k2, x); t.k2 = lio_dmin(t.k2, x); x = c;
    c = lio_dmax(t.k3, x); t.k3 = lio_dmin(t.k3, x); x = c;
    t.k4 = lio_dmin(t.k4, x);
}

// Four candidates at once: sort the four new keys (5 compare-exchanges), take the element-wise minimum of
// the sorted top-5 with the reversed new list (the five smallest of the nine, as an ascending-then-
// descending sequence), and sort that with a 5-element bitonic merge (5 compare-exchanges): 24 min/max
// operations instead of 4 x 9.  Same result as four single insertions (keys are distinct).
#define LIO_CE_ASC(a, b) do { const double lo_ = lio_dmin(a, b); b = lio_dmax(a, b); a = lo_; } while (0)
#define LIO_CE_DESC(a, b) do { const double hi_ = lio_dmax(a, b); b = lio_dmin(a, b); a = hi_; } while (0)
LIO_DEV void lio_top5_insert4(LioTop5& t, double b0, double b1, double b2, double b3)
{
    LIO_CE_ASC(b0, b1); LIO_CE_ASC(b2, b3); LIO_CE_ASC(b0, b2); LIO_CE_ASC(b1, b3); LIO_CE_ASC(b1, b2);
    double c0 = t.k0, c1 = lio_dmin(t.k1, b3), c2 = lio_dmin(t.k2, b2), c3 = lio_dmin(t.k3, b1), c4 = lio_dmin(t.k4, b0);
    LIO_CE_DESC(c0, c4); LIO_CE_DESC(c0, c2); LIO_CE_DESC(c1, c3); LIO_CE_DESC(c0, c1); LIO_CE_DESC(c2, c3);
    t.k0 = c4; t.k1 = c3; t.k2 = c2; t.k3 = c1; t.k4 = c0;
}

// FLANN L2_Simple: ((dx*dx) + dy*dy) + dz*dz, accumulated from 0
LIO_DEV float lio_sqdist(float ax, float ay, float az, float bx, float by, float bz)
{
    float r = 0.0f, d;
    d = ax - bx; r += d * d;
    d = ay - by; r += d * d;
    d = az - bz; r += d * d;
    return r;
}

// The serial Gauss-Newton step of one scan: LMOptimization MO:1702-1837 from
// the reduced sums onward, plus the loop control of scan2MapOptimization
// MO:1848-1859.  `ws` is LDS working storage.
// Called by ALL lanes of one wave.  Lane 0 carries the serial algorithm, with the 6x6 system of cv::solve in its
// registers (lio_solve6_qr_reg); the eigen-decomposition and the 6x6 product of the first iteration are spread over the
// wave (lio_eigen6_sym_wave, lio_gemm6_wave), and so are the six fp64 sines / cosines of the next transform
// (lio_pose_to_transform_wave).  This step ends every Gauss-Newton launch and is the critical path of a lone
// registration ("one-launch loop").
// The last three arguments serve the one-launch loop's speculation (lio_persist.hip) and default to the plain step:
//   deg_override >= 0  use this isDegenerate instead of the scan's stored one (iterations > 0);
//   skip_eigen         iteration 0 without the eigen-decomposition: isDegenerate / matP are left alone, the update is
//                      the non-degenerate one (a helper workgroup computes them meanwhile);
//   hold_if_done       if this step would end the registration (converged, last iteration, too few correspondences),
//                      change nothing that depends on that decision and return 2 -- the caller settles the speculation first.
// Returns 0, or 2 when held.
__device__ static int lio_gn_step(LioScanState* st, const double* sums, const LioConsts& c, LioSolveWs* ws,
                                  int* n_active, int lane, int deg_override = -1, bool skip_eigen = false, bool hold_if_done = false)
{
    const int it = st->iter;
    const int nc = (int)sums[LIO_SUM_NC];
    const bool solve = nc >= c.min_corr;                       // MO:1721-1724 (wave-uniform)
    float pose[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    float X[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    bool conv = false;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pose[k] = st->pose[k];
        st->n_corr_last = nc;
        if (it < 32) st->n_corr_iter[it] = nc;
    }

    if (solve) {
        int deg = 0;
        if (lane == 0) {
            deg = deg_override >= 0 ? deg_override : st->is_degenerate;
            float A[36];
            int p = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) {
                    const float v = (float)sums[p++];
                    A[a * 6 + b] = v; A[b * 6 + a] = v;
                }
#pragma unroll
            for (int a = 0; a < 6; ++a) X[a] = (float)sums[21 + a];
#pragma unroll
            for (int k = 0; k < 36; ++k) { st->AtA[k] = A[k]; if (it == 0) ws->AtA[k] = A[k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) st->AtB[k] = X[k];
            lio_solve6_qr_reg(A, X);                           // MO:1784
        }
        const float* matP = st->matP;
        if (it == 0 && !skip_eigen) {                          // MO:1786-1808
            LIO_LDS_FENCE();
            float Wd[6], vrow;
            lio_eigen6_sym_lanes(ws->AtA, Wd, vrow, lane);      // cv::eigen, MO:1792
            // MO:1794-1805: from the smallest eigenvalue up, rows of matV2 are cleared while the eigenvalue is below the threshold
            bool zr[6];
            bool z = true;
#pragma unroll
            for (int i = 5; i >= 0; --i) { z = z && (Wd[i] < c.eig_thresh); zr[i] = z; }
            deg = zr[5] ? 1 : 0;
            if (lane < 36) {
                const int r = lane / 6;
                bool zero = zr[0];
#pragma unroll
                for (int i = 1; i < 6; ++i) if (r == i) zero = zr[i];
                ws->V2[lane] = zero ? 0.0f : vrow;
            }
            if (lane == 0) st->is_degenerate = deg;
            lio_inv6_lu_wave(vrow, ws->B, lane);               // matV.inv(), MO:1807
            LIO_LDS_FENCE();
            lio_gemm6_wave(ws->B, ws->V2, ws->A, lane);        // matP = matV.inv() * matV2, MO:1807
            LIO_LDS_FENCE();
            if (lane < 36) st->matP[lane] = ws->A[lane];
            matP = ws->A;
        }
        if (lane == 0) {
            if (deg) {                                         // MO:1810-1815
                for (int k = 0; k < 6; ++k) ws->X2[k] = X[k];
                lio_gemm32f(matP, ws->X2, ws->X, 6, 6, 1);
#pragma unroll
                for (int k = 0; k < 6; ++k) X[k] = ws->X[k];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) pose[k] += X[k];       // MO:1817-1822

            // MO:1824-1831: rad2deg in float, squares/sqrt in double, stored as float
            const double r0 = (double)(X[0] * 57.29578f), r1 = (double)(X[1] * 57.29578f), r2 = (double)(X[2] * 57.29578f);
            const float deltaR = (float)sqrt(r0 * r0 + r1 * r1 + r2 * r2);
            const double t0 = (double)(X[3] * 100), t1 = (double)(X[4] * 100), t2 = (double)(X[5] * 100);
            const float deltaT = (float)sqrt(t0 * t0 + t1 * t1 + t2 * t2);
            conv = ((double)deltaR < c.conv_deg) && ((double)deltaT < c.conv_cm);   // MO:1833
        }
    }

    int done = 0;
    if (hold_if_done) {
        int d = 0;
        if (lane == 0) d = ((conv && !c.force_all) || it + 1 >= c.max_iters || nc < c.min_corr) ? 1 : 0;
        if (__shfl(d, 0)) return 2;
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) st->pose[k] = pose[k];
        if (it < 32) for (int k = 0; k < 6; ++k) st->pose_iter[it][k] = pose[k];
        int iters = it + 1;
        if (conv) { st->converged = 1; if (!c.force_all) done = 1; }   // MO:1857-1858
        if (iters >= c.max_iters) done = 1;                            // MO:1848
        if (nc < c.min_corr && !done) {
            // LMOptimization returned false WITHOUT touching the pose (MO:1721-1724):
            // every remaining iteration would redo identical work.  Fast-forward.
            for (int k = iters; k < c.max_iters && k < 32; ++k) {
                st->n_corr_iter[k] = nc;
                for (int j = 0; j < 6; ++j) st->pose_iter[k][j] = pose[j];
            }
            iters = c.max_iters;
            done = 1;
        }
        st->iter = iters;
        st->done = done;
        st->status = (nc < c.min_corr) ? 2 : 0;
        if (done && n_active) atomicSub(n_active, 1);
    }
    done = __shfl(done, 0);
    if (!done) {                                                   // wave-uniform
        float Tn[12], trn[6];
        lio_pose_to_transform_wave(pose, Tn, trn, lane);           // MO:1613-1616 for the next pass
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) st->Tp[k] = st->T[k];     // (search bound of the next pass, see k_s2m_iterate)
#pragma unroll
            for (int k = 0; k < 12; ++k) st->T[k] = Tn[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) st->trig[k] = trn[k];
        }
    }
    return 0;
}

// ---- candidate scan, global-memory form -----------------------------------
// One contiguous run of the replicated neighbourhood row (see k_map_nbr_*), walked in ALIGNED
// groups of four records (64 B) with the next group already in flight.  Aligning the group
// boundaries may pull in up to three records before and after the run: they are other map points
// of the SAME row list (row lists are 4-aligned and padded with far-away dummies, see
// k_map_nbr_pad_rows), i.e. a duplicate-free superset of the neighbourhood, which keeps the
// search exact and needs no predication.  Two candidates are
// evaluated per VALU op (v_pk_add_f32 / v_pk_mul_f32 on the pair-transposed records).
typedef float lio_f2 __attribute__((ext_vector_type(2)));

LIO_DEV void lio_knn_pair(const float4& a, const float4& b, lio_f2 qx, lio_f2 qy, lio_f2 qz, double& k0, double& k1)
{
    const lio_f2 X = { a.x, a.y }, Y = { a.z, a.w }, Z = { b.x, b.y };
    const lio_f2 dx = X - qx, dy = Y - qy, dz = Z - qz;
    // FLANN L2_Simple per candidate: ((dx*dx) + dy*dy) + dz*dz
    const lio_f2 d2 = (dx * dx + dy * dy) + dz * dz;
    k0 = lio_make_key(d2.x, __float_as_int(b.z));
    k1 = lio_make_key(d2.y, __float_as_int(b.w));
}

LIO_DEV void lio_knn_group(const float4& c0, const float4& c1, const float4& c2, const float4& c3,
                           lio_f2 qx, lio_f2 qy, lio_f2 qz, LioTop5& top)
{
    double k0, k1, k2, k3;
    lio_knn_pair(c0, c1, qx, qy, qz, k0, k1);
    lio_knn_pair(c2, c3, qx, qy, qz, k2, k3);
    lio_top5_insert4(top, k0, k1, k2, k3);
}

// The candidate run of one query: [beg, end) records of nbr_pts (beg aligned down to a whole group), empty: beg >= end.
LIO_DEV void lio_knn_range(const LioIterParams& P, const LioGrid& g, float qx, float qy, float qz,
                           int cy, int cz, float Rx, unsigned& beg, unsigned& end)
{
    // Rx: the current search bound -- the gate radius, or the tighter bound the previous iteration's neighbours gave (see
    // "neighbour cache" in the kernels); [xlo, xhi]: the FINE x cells (LioGrid::xs per cell) that can hold a point within it
    const int xlo = lio_cell_coord(qx - Rx, g.ox, g.inv_cell_x, g.nxf), xhi = lio_cell_coord(qx + Rx, g.ox, g.inv_cell_x, g.nxf);
    const int x0 = max(xlo, 0), x1 = min(xhi, g.nxf - 1);
    beg = end = 0;
    if (x0 > x1) return;
    int row = (min(max(cz, 0), g.nz - 1) * g.ny + min(max(cy, 0), g.ny - 1)) * g.nxf;
    // tight rows (LioGrid::tb_*): every map point within Rx of the query lies in the 3x3 cells around the query's cell of a
    // table whose cell is at least Rx, i.e. in that cell's row; the tightest such table wins (reach descending); a query
    // outside the grid is clamped as above
#pragma unroll
    for (int l = 0; l < LIO_TB_MAX; ++l) {
        if (Rx <= g.tb_reach[l]) {
            const int by = min(max(lio_cell_coord(qy, g.tb_oy[l], g.tb_inv_cell[l], g.tb_ny[l]), 0), g.tb_ny[l] - 1);
            const int bz = min(max(lio_cell_coord(qz, g.tb_oz[l], g.tb_inv_cell[l], g.tb_nz[l]), 0), g.tb_nz[l] - 1);
            row = g.tb_row0[l] + (bz * g.tb_ny[l] + by) * g.nxf;
        }
    }
#if LIO_PREFETCH == 3
    beg = (unsigned)P.nbr_start[row + x0] & ~7u;
#else
    beg = (unsigned)P.nbr_start[row + x0] & ~3u;
#endif
    end = (unsigned)P.nbr_start[row + x1 + 1];
}

// The exact 5-NN of (qx, qy, qz) among the records [beg, end) of nbr_pts, merged into `top`.
LIO_DEV void lio_knn_run(const LioIterParams& P, unsigned beg, unsigned end, float qx, float qy, float qz, LioTop5& top)
{
    if (beg >= end) return;
    const lio_f2 QX = { qx, qx }, QY = { qy, qy }, QZ = { qz, qz };
    const float4* p = P.nbr_pts + beg;                     // float4 index == record index (2 float4 per pair)
#if LIO_PREFETCH == 2
    // two groups in flight (the table is padded, reading up to two groups past a run is harmless)
    float4 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
    float4 n0 = p[4], n1 = p[5], n2 = p[6], n3 = p[7];
    for (unsigned j = beg;;) {
        float4 f0 = n0, f1 = n1, f2 = n2, f3 = n3;
        if (j + 8 < end) { f0 = p[8]; f1 = p[9]; f2 = p[10]; f3 = p[11]; }
        lio_knn_group(c0, c1, c2, c3, QX, QY, QZ, top);
        j += 4;
        if (j >= end) break;
        p += 4;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        n0 = f0; n1 = f1; n2 = f2; n3 = f3;
    }
#elif LIO_PREFETCH == 0
    // no register double-buffering: the other waves of the SIMD cover the load (A/B experiment)
    for (unsigned j = beg; j < end; j += 4, p += 4) {
        const float4 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
        lio_knn_group(c0, c1, c2, c3, QX, QY, QZ, top);
    }
#elif LIO_PREFETCH == 3
    // Two groups per trip with a single exit at the bottom (`beg` is aligned down to eight records for this; row lists are
    // padded to eight, so the extra group still belongs to the same row list): group b is loaded while a is evaluated
    // and the next trip's a while b is, each into its own registers -- no rotation copies (8 of the 57 vector
    // instructions of the rotating form were v_mov_b64).  Costs up to one extra group per point.
    float4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
    unsigned j = beg;
    do {
        const float4 b0 = p[4], b1 = p[5], b2 = p[6], b3 = p[7];
        lio_knn_group(a0, a1, a2, a3, QX, QY, QZ, top);
        p += 8;
        a0 = p[0]; a1 = p[1]; a2 = p[2]; a3 = p[3];            // (reads past the run at the last trip: the table is padded)
        lio_knn_group(b0, b1, b2, b3, QX, QY, QZ, top);
        j += 8;
    } while (j < end);
#else
    float4 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
    for (unsigned j = beg + 4; j < end; j += 4) {
        p += 4;
        const float4 n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3];
        lio_knn_group(c0, c1, c2, c3, QX, QY, QZ, top);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    lio_knn_group(c0, c1, c2, c3, QX, QY, QZ, top);
#endif
}

// The exact 5-NN of one query over the replicated rows.  (Rx, bound2): the search radius and the sentinel's squared distance --
// the gate, or (`bounded`) the tighter pair the previous iteration's neighbours gave.  An unbounded query on a dense map goes
// through the tight tables first (LioGrid::tb_try): one copy of the candidate loop serves every round.
LIO_DEV void lio_knn_global(const LioIterParams& P, const LioGrid& g, float qx, float qy, float qz,
                            int cy, int cz, float Rx, float bound2, bool bounded, LioTop5& top)
{
    int lvl = bounded ? -1 : g.tb_try;
    for (;;) {
        float rx = Rx, b2 = bound2;
        if (lvl >= 0) {
            rx = lvl == 2 ? g.tb_reach[2] : (lvl == 1 ? g.tb_reach[1] : g.tb_reach[0]);
            b2 = lvl == 2 ? g.tb_try_b2[2] : (lvl == 1 ? g.tb_try_b2[1] : g.tb_try_b2[0]);
        }
        // (d2 == b2 with any real index sorts below the sentinel, so ties at the bound are kept)
        const double sentinel = lio_make_key(b2, -1);                     // index 0xffffffff: above every real index
        top.k0 = top.k1 = top.k2 = top.k3 = top.k4 = sentinel;
        unsigned beg, end;
        lio_knn_range(P, g, qx, qy, qz, cy, cz, rx, beg, end);
        lio_knn_run(P, beg, end, qx, qy, qz, top);
        // five points within the tried radius: every point outside the table's row is farther than that -- done
        if (lvl < 0 || lio_key_idx(top.k4) != -1) break;
        --lvl;
    }
}

// Association of one scan point from its five nearest map points (indices into the caller's map order):
// surfOptimization MO:1642-1683 -- plane through the neighbours, plane test, weight, coefficients -- or, for
// the CORNER extension, the point-to-line form of upstream LIO-SAM.  Returns "accepted" (MO:1679).
template <bool CORNER>
LIO_DEV bool lio_assoc_point(const LioIterParams& P, const int nn[5], float qx, float qy, float qz,
                             float px, float py, float pz, float& cxx, float& cyy, float& czz, float& cww)
{
    bool accept = false;
    // MO:1642-1646: neighbours in the caller's map order (original xyz)
    float a[5][3], m[5][3];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float4 mp = P.map_xyz4[nn[j]];
        m[j][0] = a[j][0] = mp.x;
        m[j][1] = a[j][1] = mp.y;
        m[j][2] = a[j][2] = mp.z;
    }
    if (CORNER) {
        accept = lio_corner_assoc(m, qx, qy, qz, P.c.weight, P.c.min_s, cxx, cyy, czz, cww);
    } else {
        float X0[3];
        lio_plane_qr5x3(a, X0);                                  // MO:1648
        float pa = X0[0], pb = X0[1], pc = X0[2], pd = 1;         // MO:1650-1653
        const float ps = sqrtf(pa * pa + pb * pb + pc * pc);      // MO:1655
        pa /= ps; pb /= ps; pc /= ps; pd /= ps;                   // MO:1656
        bool planeValid = true;                                   // MO:1658-1666
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float v = fabsf(pa * m[j][0] + pb * m[j][1] + pc * m[j][2] + pd);
            if ((double)v > P.c.plane_tol) planeValid = false;
        }
        if (planeValid) {
            const float pd2 = pa * qx + pb * qy + pc * qz + pd;   // MO:1669
            const float r2 = px * px + py * py + pz * pz;
            // MO:1671-1672 (product, quotient and difference in double)
            const float s = (float)(1 - P.c.weight * (double)fabsf(pd2) / (double)sqrtf(sqrtf(r2)));
            cxx = s * pa; cyy = s * pb; czz = s * pc; cww = s * pd2;   // MO:1674-1677
            accept = (double)s > P.c.min_s;                       // MO:1679
        }
    }
    return accept;
}

// (a, b) column pair of each of the 28 sums: 21 upper-triangle JtJ, 6 Jtr, N_c
static __constant__ int c_pair_a[32] = { 0,0,0,0,0,0, 1,1,1,1,1, 2,2,2,2, 3,3,3, 4,4, 5,  0,1,2,3,4,5, 7, 0,0,0,0 };
static __constant__ int c_pair_b[32] = { 0,1,2,3,4,5, 1,2,3,4,5, 2,3,4,5, 3,4,5, 4,5, 5,  6,6,6,6,6,6, 7, 0,0,0,0 };

// Tail of an association workgroup, executed by its wave 0 after the partial sums were stored
// write-through: drain, arrive on the scan's counter; the workgroup whose arrival is last adds up
// the scan's partials in workgroup order and runs the Gauss-Newton step (or publishes the sums).
LIO_DEV void lio_arrive_and_finish(const LioIterParams& P, const LioBlockDesc& bd, LioScanState* st, int lane,
                                   double* s_sum, LioSolveWs* s_ws, long long* stamp)
{
    // ORDERING (release half of the arrive protocol): the caller's 28 write-through (sc1) stores of this workgroup's partial
    // sums have left the wave's vector-memory queue -- they are complete at the agent-coherent level -- before the arrival
    // below can be observed by another workgroup.  An agent-scope release fence would order them too, but it writes back the
    // whole L2 of the XCD; the stores are sc1 precisely so that this counter wait is all that is needed.
#if !LIO_ARRIVE_ACQREL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    int last = 0;
    if (lane == 0) {
#if LIO_ARRIVE_ACQREL
        // A/B variant (make EXTRA=-DLIO_ARRIVE_ACQREL=1): the same protocol in the language's terms -- an acquire-release RMW at
        // agent scope instead of the hand-placed wait.  Measured in round 3; the default stays the wait.
        const unsigned old = __hip_atomic_fetch_add(&P.arrive[bd.scan], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
        const unsigned old = atomicAdd(&P.arrive[bd.scan], 1u);
#endif
        last = (old == (unsigned)bd.n_blk - 1u);
    }
    last = __shfl(last, 0);
    if (stamp && lane == 0) stamp[6] = (long long)__builtin_readcyclecounter();
    if (!last) return;

    // last workgroup of this scan: fixed-order sum over the scan's chunks
    if (lane < 28) {
        const double* base_p = P.partials + (size_t)bd.scan * P.max_blk * LIO_SUMS + lane;
        double v = 0.0;
        for (int b = 0; b < bd.n_blk; b += 8) {          // 8 write-through loads in flight, summed in chunk order
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = __hip_atomic_load(base_p + (size_t)min(b + u, bd.n_blk - 1) * LIO_SUMS,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) v += (b + u < bd.n_blk) ? t[u] : 0.0;
        }
        s_sum[lane] = v;
        if (P.sums_out) P.sums_out[(size_t)bd.scan * LIO_SUMS + lane] = v;
    }
    // ORDERING: the LDS writes of s_sum by lanes 0..27 are complete before lane 0 of the SAME wave reads them in lio_gn_step
    // (acquire half: the partials themselves were read with sc1 loads issued AFTER the atomic returned -- `last` depends on
    // its result --, so they cannot be older than the last arrival)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) P.arrive[bd.scan] = 0;                  // re-arm for the next launch
    if (!P.sums_out) lio_gn_step(st, s_sum, P.c, s_ws, P.n_active, lane);
}

