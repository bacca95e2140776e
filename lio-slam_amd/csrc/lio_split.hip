// lio_split.hip -- the split Gauss-Newton pipeline of the scan-to-map path (cfg.pipeline): neighbour
// certificate, balanced candidate scan and fit as three gfx950 kernels per iteration.
// MO = /root/reference/src/liorf/src/mapOptmization.cpp.  Compile with -ffp-contract=off.
#include "lio_s2m_device.h"

// ===================================================================== split pipeline
// One Gauss-Newton iteration as three launches (cfg.pipeline), results bit-identical to k_s2m_iterate:
//
//   k_s2m_cert : per point, the NEIGHBOUR CERTIFICATE.  The last candidate scan of a point left its
//                LIO_CACHE_K nearest map points C (indices), the position q_ref it searched from and a lower
//                bound lb on the distance from q_ref to every map point outside C.  At the new position q,
//                with m = |q - q_ref|: every point outside C is at least lb - m away; the exact fp32
//                distances of q to the members of C are recomputed (8 gathers) and sorted by (d2, index).
//                If sqrt(d2_5th) + m < lb (rounded against us), no outsider can be among -- or tie with --
//                the five nearest: the exact 5-NN of pcl's nearestKSearch (MO:1631) are the first five of
//                the sorted cache and NO candidate scan is needed.  Otherwise the point is queued for
//                k_s2m_scan, bounded by the 8th cached distance (an upper bound on the true 8th).
//                Queued points of a group (<= 1024 consecutive points) are ordered by candidate-run length.
//   k_s2m_scan : the candidate scan (as in k_s2m_iterate, top-8 instead of top-5) over the queued points
//                only, one wave per 64 queue entries: the lanes of a wave walk runs of similar length and
//                no workgroup barrier couples the waves.  Rewrites the point's cache.
//   k_s2m_fit  : per point with flag 1: plane fit, weight, Jacobian row from cache[0..4]; the fp64
//                normal-equation sums, the arrival counter and the in-launch solve exactly as in
//                k_s2m_iterate (same chunks, same summation order => the same bits).
LIO_DEV bool lio_point_active(const LioIterParams& P, const LioGrid& g, float qx, float qy, float qz,
                              int& cx, int& cy, int& cz)
{
    bool a = true;
    if (P.shard.axis >= 0) {                                     // owner-computes (multi-GPU)
        const float qa = P.shard.axis == 0 ? qx : (P.shard.axis == 1 ? qy : qz);
        int gc = lio_cell_coord(qa, P.shard.gorigin, P.shard.inv_cell, P.shard.gdim);
        gc = min(max(gc, 0), P.shard.gdim - 1);
        a = gc >= P.shard.lo && gc < P.shard.hi;
    }
    cx = lio_cell_coord(qx, g.ox, g.inv_cell, g.nx);
    cy = lio_cell_coord(qy, g.oy, g.inv_cell, g.ny);
    cz = lio_cell_coord(qz, g.oz, g.inv_cell, g.nz);
    a = a && (fabsf(qx) <= 3.0e38f) && (fabsf(qy) <= 3.0e38f) && (fabsf(qz) <= 3.0e38f);
    a = a && cx >= -g.k && cx < g.nx + g.k && cy >= -g.k && cy < g.ny + g.k && cz >= -g.k && cz < g.nz + g.k;
    return a;
}

// x-cell range of a bounded search and the length of the candidate run (in groups of four records)
LIO_DEV int lio_run_groups(const LioIterParams& P, const LioGrid& g, float qx, float bound2, int cx, int cy, int cz,
                           int& xlo, int& xhi)
{
    xlo = -0x7fffffff; xhi = 0x7fffffff;
    if (bound2 < P.c.max_sq_dist) {
        const float R = sqrtf(bound2) * 1.0001f + 1e-6f;
        xlo = lio_cell_coord(qx - R, g.ox, g.inv_cell, g.nx);
        xhi = lio_cell_coord(qx + R, g.ox, g.inv_cell, g.nx);
    }
    const int x0 = max(max(cx - g.k, 0), xlo), x1 = min(min(cx + g.k, g.nx - 1), xhi);
    if (x0 > x1) return 0;
    const int row = (min(max(cz, 0), g.nz - 1) * g.ny + min(max(cy, 0), g.ny - 1)) * g.nx;
    const unsigned beg = (unsigned)P.nbr_start[row + x0] & ~3u;
    const unsigned end = (unsigned)P.nbr_start[row + x1 + 1];
    return beg < end ? (int)((end - beg + 3u) >> 2) : 0;
}

#define LIO_XCD_REMAP(wg)                                                        \
    do {                                                                         \
        if (P.xcd_remap) {                                                       \
            const int n8_ = gridDim.x >> 3;                                      \
            if ((wg) < n8_ * 8) (wg) = ((wg) & 7) * n8_ + ((wg) >> 3);           \
        }                                                                        \
    } while (0)

__global__ __launch_bounds__(LIO_BLOCK, 6) void k_s2m_cert(LioSplitParams S)
{
    const LioIterParams& P = S.it;
    __shared__ int s_hist[64], s_base[64];
    int wg = blockIdx.x;
    LIO_XCD_REMAP(wg);
    const LioGroupDesc gd = S.groups[wg];
    const LioScanState* st = &P.state[gd.scan];
    if (st->done) return;                                  // workgroup-uniform
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = st->T[k];
    const bool use_cache = st->iter > 0;                   // every registration starts cold
    const int base = st->offset;
    const LioGrid g = P.grid;
    if (threadIdx.x < 64) s_hist[threadIdx.x] = 0;
    __syncthreads();

    int slot[LIO_GROUP_BLOCKS], rank[LIO_GROUP_BLOCKS], bucket[LIO_GROUP_BLOCKS];
#pragma unroll
    for (int pp = 0; pp < LIO_GROUP_BLOCKS; ++pp) {
        const int li = pp * LIO_BLOCK + (int)threadIdx.x;
        bucket[pp] = -1; rank[pp] = 0;
        slot[pp] = base + gd.first + li;
        if (li >= gd.n) continue;
        const int ci = slot[pp];
        const float px = P.sx[ci], py = P.sy[ci], pz = P.sz[ci];
        const float qx = T[0] * px + T[1] * py + T[2]  * pz + T[3];     // pointAssociateToMap, MO:841-847
        const float qy = T[4] * px + T[5] * py + T[6]  * pz + T[7];
        const float qz = T[8] * px + T[9] * py + T[10] * pz + T[11];
        int cx, cy, cz;
        const bool act = lio_point_active(P, g, qx, qy, qz, cx, cy, cz);
        int flag = 0;
        if (act) {
            bool need = true;
            float bound2 = P.c.max_sq_dist;
            if (use_cache) {
                const float4 cq = S.cache_q[ci];
                if (cq.w >= 0.0f) {
                    const int4 ia = reinterpret_cast<const int4*>(S.cache_idx)[(size_t)ci * 2];
                    const int4 ib = reinterpret_cast<const int4*>(S.cache_idx)[(size_t)ci * 2 + 1];
                    const int id[8] = { ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w };
                    double k[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 mp = P.map_xyz4[max(id[j], 0)];
                        const float d2 = lio_sqdist(mp.x, mp.y, mp.z, qx, qy, qz);
                        k[j] = id[j] >= 0 ? lio_make_key(d2, id[j]) : lio_make_key(3.0e38f, -1);
                    }
                    lio_sort8(k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7]);
                    const float d5 = lio_key_d2(k[4]);
                    const float mv = sqrtf(lio_sqdist(qx, qy, qz, cq.x, cq.y, cq.z));
                    if ((sqrtf(d5) + mv) * 1.0001f + 1e-6f < cq.w) {
                        // certified: the five nearest are cache members; keep the cache in its new order
                        need = false;
                        flag = d5 < P.c.max_sq_dist ? 1 : 0;             // gate MO:1641
                        reinterpret_cast<int4*>(S.cache_idx)[(size_t)ci * 2] =
                            make_int4(lio_key_idx(k[0]), lio_key_idx(k[1]), lio_key_idx(k[2]), lio_key_idx(k[3]));
                        reinterpret_cast<int4*>(S.cache_idx)[(size_t)ci * 2 + 1] =
                            make_int4(lio_key_idx(k[4]), lio_key_idx(k[5]), lio_key_idx(k[6]), lio_key_idx(k[7]));
                    } else {
                        // the 8th cached distance bounds the true 8th from above (ties at the bound are kept)
                        bound2 = fminf(bound2, lio_key_d2(k[7]));
                    }
                }
            }
            if (need) {
                flag = 2;
                S.scan_bound2[ci] = bound2;
                int xlo, xhi;
                const int len = lio_run_groups(P, g, qx, bound2, cx, cy, cz, xlo, xhi);
                // order of the group's work list: by candidate-run length (balanced waves, S.sort_mode 1), by chunk of 256
                // points and then by length (2: keeps a wave's points close together), or as they come (0)
                bucket[pp] = S.sort_mode == 1 ? min(len, 63) : (S.sort_mode == 2 ? (3 - pp) * 16 + min(len >> 2, 15) : 0);
                rank[pp] = atomicAdd(&s_hist[bucket[pp]], 1);
            }
        }
        S.pt_flag[ci] = flag;
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                // longest runs first: exclusive suffix sums
        const int lane = threadIdx.x;
        const int v = s_hist[63 - lane];
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        s_base[63 - lane] = incl - v;
        if (lane == 63) {
            S.scan_cnt[wg] = incl;
            if (S.stats) S.stats[(size_t)min(st->iter, 31) * gridDim.x + wg] = incl;   // diagnostics: scans queued per iteration
        }
    }
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < LIO_GROUP_BLOCKS; ++pp)
        if (bucket[pp] >= 0) S.scan_list[gd.list_base + s_base[bucket[pp]] + rank[pp]] = slot[pp];
}

__global__ __launch_bounds__(LIO_BLOCK, 5) void k_s2m_scan(LioSplitParams S)
{
    const LioIterParams& P = S.it;
    int b = blockIdx.x;
    LIO_XCD_REMAP(b);
    const int grp = b / (LIO_GROUP_BLOCKS * LIO_BLOCK / 256);
    const LioGroupDesc gd = S.groups[grp];
    const LioScanState* st = &P.state[gd.scan];
    if (st->done) return;
    const int cnt = S.scan_cnt[grp];
    const int lane = threadIdx.x & 63;
    const int w = (b % (LIO_GROUP_BLOCKS * LIO_BLOCK / 256)) * (LIO_BLOCK / 64) + (int)(threadIdx.x >> 6);   // wave of the group
    if (w * 64 >= cnt) return;                              // wave-uniform
    const int e = w * 64 + lane;
    const bool valid = e < cnt;
    const int ci = S.scan_list[gd.list_base + min(e, cnt - 1)];
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = st->T[k];
    const LioGrid g = P.grid;
    const float px = P.sx[ci], py = P.sy[ci], pz = P.sz[ci];
    const float qx = T[0] * px + T[1] * py + T[2]  * pz + T[3];
    const float qy = T[4] * px + T[5] * py + T[6]  * pz + T[7];
    const float qz = T[8] * px + T[9] * py + T[10] * pz + T[11];
    const int cx = lio_cell_coord(qx, g.ox, g.inv_cell, g.nx);
    const int cy = lio_cell_coord(qy, g.oy, g.inv_cell, g.ny);
    const int cz = lio_cell_coord(qz, g.oz, g.inv_cell, g.nz);
    const float bound2 = S.scan_bound2[ci];
    int xlo = -0x7fffffff, xhi = 0x7fffffff;
    if (bound2 < P.c.max_sq_dist) {
        const float R = sqrtf(bound2) * 1.0001f + 1e-6f;
        xlo = lio_cell_coord(qx - R, g.ox, g.inv_cell, g.nx);
        xhi = lio_cell_coord(qx + R, g.ox, g.inv_cell, g.nx);
    }
    // (d2 == bound2 with any real index sorts below the sentinel, so ties at the bound are kept)
    const double sentinel = lio_make_key(bound2, -1);
    LioTop8 top = { sentinel, sentinel, sentinel, sentinel, sentinel, sentinel, sentinel, sentinel };
    if (valid) lio_knn_global8(P, g, qx, qy, qz, cx, cy, cz, xlo, xhi, top);
    if (!valid) return;
    const double kk[8] = { top.k0, top.k1, top.k2, top.k3, top.k4, top.k5, top.k6, top.k7 };
    int id[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) id[j] = kk[j] < sentinel ? lio_key_idx(kk[j]) : -1;
    reinterpret_cast<int4*>(S.cache_idx)[(size_t)ci * 2] = make_int4(id[0], id[1], id[2], id[3]);
    reinterpret_cast<int4*>(S.cache_idx)[(size_t)ci * 2 + 1] = make_int4(id[4], id[5], id[6], id[7]);
    // every map point outside the cache is farther than the 8th member -- or, with fewer than eight
    // candidates inside the bound, farther than the bound (rounded down)
    const float far2 = id[7] >= 0 ? lio_key_d2(top.k7) : bound2;
    S.cache_q[ci] = make_float4(qx, qy, qz, sqrtf(far2) * 0.9999f - 1e-6f);
    S.pt_flag[ci] = (id[4] >= 0 && lio_key_d2(top.k4) < P.c.max_sq_dist) ? 1 : 0;   // gate MO:1641
}

__global__ __launch_bounds__(LIO_BLOCK, 5) void k_s2m_fit(LioSplitParams S)
{
    const LioIterParams& P = S.it;
    __shared__ __attribute__((aligned(16))) double s_rows[LIO_BLOCK][8];
    __shared__ double s_part[8][28];
    __shared__ double s_sum[28];
    __shared__ LioSolveWs s_ws;
    int wg = blockIdx.x;
    LIO_XCD_REMAP(wg);
    const LioBlockDesc bd = P.blocks[wg];
    LioScanState* st = &P.state[bd.scan];
    if (st->done) return;
    float T[12], tr[6];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = st->T[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) tr[k] = st->trig[k];
    const int n_pts = st->n_pts, base = st->offset;
    const bool record = (P.rec_flag != nullptr) && (st->iter == P.c.record_iter);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = bd.first + (int)threadIdx.x;
    const bool inr = li < n_pts;
    const int ci = base + (inr ? li : 0);
    const float px = P.sx[ci], py = P.sy[ci], pz = P.sz[ci];
    const float qx = T[0] * px + T[1] * py + T[2]  * pz + T[3];
    const float qy = T[4] * px + T[5] * py + T[6]  * pz + T[7];
    const float qz = T[8] * px + T[9] * py + T[10] * pz + T[11];
    const bool ok = inr && S.pt_flag[ci] == 1;
    int nn[5] = { -1, -1, -1, -1, -1 };
    if (ok) {
        const int4 ia = reinterpret_cast<const int4*>(S.cache_idx)[(size_t)ci * 2];
        nn[0] = ia.x; nn[1] = ia.y; nn[2] = ia.z; nn[3] = ia.w;
        nn[4] = S.cache_idx[(size_t)ci * 8 + 4];
    }
    float cxx = 0.0f, cyy = 0.0f, czz = 0.0f, cww = 0.0f;
    bool accept = false;
    if (ok) accept = lio_assoc_point<false>(P, nn, qx, qy, qz, px, py, pz, cxx, cyy, czz, cww);
    if (record && inr) {
        const int oi = P.perm ? P.perm[base + li] : base + li;           // the record is kept in the CALLER's point order
        P.rec_flag[oi] = accept ? 1 : 0;
        reinterpret_cast<float4*>(P.rec_coeff)[oi] = make_float4(cxx, cyy, czz, cww);
#pragma unroll
        for (int j = 0; j < 5; ++j) P.rec_nn[(size_t)oi * 5 + j] = ok ? nn[j] : -1;
    }
    float row[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, rhs = 0.0f;
    if (accept) lio_jacobian_row(tr, px, py, pz, cxx, cyy, czz, cww, P.c.jac_exact, row, rhs);
    {
        double2* dst = reinterpret_cast<double2*>(s_rows[threadIdx.x]);
        dst[0] = make_double2((double)row[0], (double)row[1]);
        dst[1] = make_double2((double)row[2], (double)row[3]);
        dst[2] = make_double2((double)row[4], (double)row[5]);
        dst[3] = make_double2((double)rhs, accept ? 1.0 : 0.0);
    }
    const int red_g = threadIdx.x >> 5, red_s = threadIdx.x & 31;
    const int red_a = c_pair_a[red_s], red_b = c_pair_b[red_s];
    double red_acc = 0.0;
    __syncthreads();
    if (red_s < 28) {
#pragma unroll 8
        for (int p = red_g; p < LIO_BLOCK; p += 8)
            red_acc = __builtin_fma(s_rows[p][red_a], s_rows[p][red_b], red_acc);   // same order as k_s2m_iterate
    }
    if (red_s < 28) s_part[red_g][red_s] = red_acc;
    __syncthreads();
    if (wave != 0) return;
    double* part = P.partials + ((size_t)bd.scan * P.max_blk + bd.blk) * LIO_SUMS;
    if (lane < 28) {
        double v = s_part[0][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) v += s_part[w][lane];
        __hip_atomic_store(part + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lio_arrive_and_finish(P, bd, st, lane, s_sum, &s_ws, nullptr);
}

void lio_launch_split_iteration(const LioSplitParams& S, int n_blocks, hipStream_t s)
{
    if (n_blocks <= 0 || S.n_groups <= 0) return;
    hipLaunchKernelGGL(k_s2m_cert, dim3(S.n_groups), dim3(LIO_BLOCK), 0, s, S);
    hipLaunchKernelGGL(k_s2m_scan, dim3(S.n_groups * (LIO_GROUP_BLOCKS * LIO_BLOCK / 256)), dim3(LIO_BLOCK), 0, s, S);
    hipLaunchKernelGGL(k_s2m_fit, dim3(n_blocks), dim3(LIO_BLOCK), 0, s, S);
}

