// lio_cert.hip -- k_s2m_iterate_cert: the fused Gauss-Newton launch with the neighbour certificate inside
// (cfg.pipeline = 3).  Same results as k_s2m_iterate, bit for bit; see lio_split.hip for the certificate's proof.
// MO = /root/reference/src/liorf/src/mapOptmization.cpp.  Compile with -ffp-contract=off.
//
// Per workgroup (256 consecutive points of one scan):
//   A  transform, ownership, cells (as k_s2m_iterate)
//   B  certificate: from the second iteration on, the exact distances to the point's 8 cached neighbours are
//      recomputed and sorted; if sqrt(d2_5th) + |q - q_ref| < lb (lb = a lower bound on the distance from q_ref to every
//      map point outside the cache), the exact 5-NN of MO:1631 are the first five and the point needs no candidate scan
//   C  the points that do need one are compacted through LDS and scanned by the FIRST threads of the workgroup (top-8,
//      bounded by the 8th cached distance): when few points fail, one wave scans while the others wait at the barrier
//      without issuing; the scanning thread rewrites the point's cache and hands the five neighbours back through LDS
//      -- and fits the plane (column-pivoted QR, MO:1643-1666).  The plane depends on the ORDERED neighbour tuple only, not
//      on the pose: a certified point whose tuple is the one of the previous iteration re-uses its plane and skips the queue
//   D  weight, coefficients, Jacobian row, fp64 sums, arrival, in-launch solve (k_s2m_iterate's arithmetic and order =>
//      the same bits)
#include "lio_s2m_device.h"

#define LIO_XCD_REMAP(wg)                                                        \
    do {                                                                         \
        if (P.xcd_remap) {                                                       \
            const int n8_ = gridDim.x >> 3;                                      \
            if ((wg) < n8_ * 8) (wg) = ((wg) & 7) * n8_ + ((wg) >> 3);           \
        }                                                                        \
    } while (0)

__global__ __launch_bounds__(LIO_BLOCK, 5) void k_s2m_iterate_cert(LioSplitParams S)
{
    const LioIterParams& P = S.it;
    // 16 KiB shared by the queue structures of phase C and, afterwards, the Jacobian rows of phase D
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[LIO_BLOCK * 64];
    double (*s_rows)[8] = reinterpret_cast<double (*)[8]>(s_raw);
    float4* s_q = reinterpret_cast<float4*>(s_raw);                      // [256] queue: (q, squared search bound; < 0: plane only)
    int* s_owner = reinterpret_cast<int*>(s_raw + 4096);                 // [256] queue position -> owner thread
    int (*s_res)[6] = reinterpret_cast<int (*)[6]>(s_raw + 5120);        // [256] per owner: five neighbours, flags
    float4* s_plane = reinterpret_cast<float4*>(s_raw + 11264);          // [256] per owner: the plane
    __shared__ double s_part[8][28];
    __shared__ double s_sum[28];
    __shared__ LioSolveWs s_ws;
    __shared__ int s_cnt;

    int wg = blockIdx.x;
    LIO_XCD_REMAP(wg);
    const int wg_mode = P.blk_skip != nullptr ? (int)P.blk_skip[wg] : 0;   // map sharding, see k_shard_cull
    if (wg_mode == 1) return;
    const LioBlockDesc bd = P.blocks[wg];
    LioScanState* st = &P.state[bd.scan];
    if (st->done) return;                                  // workgroup-uniform
    float T[12], tr[6];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = st->T[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) tr[k] = st->trig[k];
    const int n_pts = st->n_pts, base = st->offset;
    const bool record = (P.rec_flag != nullptr) && (st->iter == P.c.record_iter);
    const bool use_cache = st->iter > 0;                   // every registration starts cold
    const LioGrid g = P.grid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_cnt = 0;

    // ---- A: transform (pointAssociateToMap, MO:841-847), ownership
    const int li = bd.first + (int)threadIdx.x;
    const bool inr = li < n_pts;
    const int ci = base + (inr ? li : 0);
    const float px = P.sx[ci], py = P.sy[ci], pz = P.sz[ci];
    const float qx = T[0] * px + T[1] * py + T[2]  * pz + T[3];
    const float qy = T[4] * px + T[5] * py + T[6]  * pz + T[7];
    const float qz = T[8] * px + T[9] * py + T[10] * pz + T[11];
    bool act = inr;
    if (P.shard.axis >= 0 && wg_mode != 2) {               // owner-computes (multi-GPU)
        const float qa = P.shard.axis == 0 ? qx : (P.shard.axis == 1 ? qy : qz);
        int gc = lio_cell_coord(qa, P.shard.gorigin, P.shard.inv_cell, P.shard.gdim);
        gc = min(max(gc, 0), P.shard.gdim - 1);
        act = act && gc >= P.shard.lo && gc < P.shard.hi;
    }
    {
        const int cx = lio_cell_coord(qx, g.ox, g.inv_cell, g.nx), cy = lio_cell_coord(qy, g.oy, g.inv_cell, g.ny),
                  cz = lio_cell_coord(qz, g.oz, g.inv_cell, g.nz);
        act = act && (fabsf(qx) <= 3.0e38f) && (fabsf(qy) <= 3.0e38f) && (fabsf(qz) <= 3.0e38f);
        act = act && cx >= -g.k && cx < g.nx + g.k && cy >= -g.k && cy < g.ny + g.k && cz >= -g.k && cz < g.nz + g.k;
    }

    // ---- B: certificate, and whether the plane of the previous iteration still stands
    bool need_scan = act, need_plane = false, ok = false, planeValid = false;
    float bound2 = P.c.max_sq_dist;
    int nn[5] = { -1, -1, -1, -1, -1 };
    float pa = 0.0f, pb = 0.0f, pc = 0.0f, pd = 0.0f;
    if (act && use_cache) {
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
                need_scan = false;                                     // certified: no outsider can be among (or tie with) the five
                ok = d5 < P.c.max_sq_dist;                             // gate MO:1641
#pragma unroll
                for (int j = 0; j < 5; ++j) nn[j] = lio_key_idx(k[j]);
                // the plane depends on the ORDERED neighbour tuple only: the same tuple as last time => the same plane
                const bool same = nn[0] == id[0] && nn[1] == id[1] && nn[2] == id[2] && nn[3] == id[3] && nn[4] == id[4];
                const int pstate = same ? S.plane_state[ci] : 0;
                if (ok && pstate != 0) {
                    const float4 pl = S.plane[ci];
                    pa = pl.x; pb = pl.y; pc = pl.z; pd = pl.w;
                    planeValid = pstate == 1;
                } else {
                    need_plane = ok;
                    if (!same) {                                       // keep "cache_idx[0..4] is the tuple the kept plane belongs to"
                        reinterpret_cast<int4*>(S.cache_idx)[(size_t)ci * 2] =
                            make_int4(nn[0], nn[1], nn[2], nn[3]);
                        reinterpret_cast<int4*>(S.cache_idx)[(size_t)ci * 2 + 1] =
                            make_int4(nn[4], lio_key_idx(k[5]), lio_key_idx(k[6]), lio_key_idx(k[7]));
                        if (!ok) S.plane_state[ci] = 0;
                    }
                }
            } else {
                bound2 = fminf(bound2, lio_key_d2(k[7]));              // the 8th cached distance bounds the true 8th from above
            }
        }
    }

    // ---- C: one queue for the points that need a candidate scan (and then a plane) or a plane only, worked off by the
    //         first threads of the workgroup
    __syncthreads();                                                   // s_cnt = 0 is visible
    const bool in_q = need_scan || need_plane;
    {
        const unsigned long long m = __ballot(in_q);
        int wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&s_cnt, (int)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (in_q) {
            const int pos = wbase + (int)__popcll(m & ((1ull << lane) - 1ull));
            s_q[pos] = make_float4(qx, qy, qz, need_scan ? bound2 : -1.0f);
            s_owner[pos] = (int)threadIdx.x;
            if (!need_scan) {
#pragma unroll
                for (int j = 0; j < 5; ++j) s_res[threadIdx.x][j] = nn[j];
            }
        }
    }
    __syncthreads();
    const int cnt = s_cnt;                                             // workgroup-uniform
    if ((int)threadIdx.x < cnt) {
        const float4 e = s_q[threadIdx.x];
        const int owner = s_owner[threadIdx.x];
        const int co = base + bd.first + owner;
        int n5[5];
        bool okp = true;
        if (e.w >= 0.0f) {
            const int cx = lio_cell_coord(e.x, g.ox, g.inv_cell, g.nx), cy = lio_cell_coord(e.y, g.oy, g.inv_cell, g.ny),
                      cz = lio_cell_coord(e.z, g.oz, g.inv_cell, g.nz);
            int xlo = -0x7fffffff, xhi = 0x7fffffff;
            if (e.w < P.c.max_sq_dist) {
                const float R = sqrtf(e.w) * 1.0001f + 1e-6f;
                xlo = lio_cell_coord(e.x - R, g.ox, g.inv_cell, g.nx);
                xhi = lio_cell_coord(e.x + R, g.ox, g.inv_cell, g.nx);
            }
            // (d2 == bound with any real index sorts below the sentinel, so ties at the bound are kept)
            const double sentinel = lio_make_key(e.w, -1);
            LioTop8 top = { sentinel, sentinel, sentinel, sentinel, sentinel, sentinel, sentinel, sentinel };
            lio_knn_global8(P, g, e.x, e.y, e.z, cx, cy, cz, xlo, xhi, top);
            const double kk[8] = { top.k0, top.k1, top.k2, top.k3, top.k4, top.k5, top.k6, top.k7 };
            int id[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) id[j] = kk[j] < sentinel ? lio_key_idx(kk[j]) : -1;
            reinterpret_cast<int4*>(S.cache_idx)[(size_t)co * 2] = make_int4(id[0], id[1], id[2], id[3]);
            reinterpret_cast<int4*>(S.cache_idx)[(size_t)co * 2 + 1] = make_int4(id[4], id[5], id[6], id[7]);
            const float far2 = id[7] >= 0 ? lio_key_d2(top.k7) : e.w;
            S.cache_q[co] = make_float4(e.x, e.y, e.z, sqrtf(far2) * 0.9999f - 1e-6f);
#pragma unroll
            for (int j = 0; j < 5; ++j) n5[j] = id[j];
            okp = id[4] >= 0 && lio_key_d2(top.k4) < P.c.max_sq_dist;   // gate MO:1641
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) n5[j] = s_res[owner][j];
        }
        float qa = 0.0f, qb = 0.0f, qc = 0.0f, qd = 0.0f;
        int pstate = 0;
        if (okp) {
            bool valid;
            lio_plane_from_nn(P, n5, qa, qb, qc, qd, valid);
            pstate = valid ? 1 : 2;
            S.plane[co] = make_float4(qa, qb, qc, qd);
        }
        S.plane_state[co] = pstate;
        s_plane[owner] = make_float4(qa, qb, qc, qd);
#pragma unroll
        for (int j = 0; j < 5; ++j) s_res[owner][j] = n5[j];
        s_res[owner][5] = pstate;                                      // 0: gate failed, 1 / 2: plane valid / not valid
    }
    if (cnt > 0) __syncthreads();
    if (in_q) {
#pragma unroll
        for (int j = 0; j < 5; ++j) nn[j] = s_res[threadIdx.x][j];
        const int pstate = s_res[threadIdx.x][5];
        const float4 pl = s_plane[threadIdx.x];
        ok = pstate != 0;
        planeValid = pstate == 1;
        pa = pl.x; pb = pl.y; pc = pl.z; pd = pl.w;
    }

    // ---- D: coefficients, row, sums, arrival (k_s2m_iterate's arithmetic and order)
    float cxx = 0.0f, cyy = 0.0f, czz = 0.0f, cww = 0.0f;
    bool accept = false;
    if (ok && planeValid) accept = lio_coeff_from_plane(P, pa, pb, pc, pd, qx, qy, qz, px, py, pz, cxx, cyy, czz, cww);
    if (record && inr) {
        const int oi = P.perm ? P.perm[base + li] : base + li;           // the record is kept in the CALLER's point order
        P.rec_flag[oi] = accept ? 1 : 0;
        reinterpret_cast<float4*>(P.rec_coeff)[oi] = make_float4(cxx, cyy, czz, cww);
#pragma unroll
        for (int j = 0; j < 5; ++j) P.rec_nn[(size_t)oi * 5 + j] = ok ? nn[j] : -1;
    }
    float row[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, rhs = 0.0f;
    if (accept) lio_jacobian_row(tr, px, py, pz, cxx, cyy, czz, cww, P.c.jac_exact, row, rhs);
    __syncthreads();                                                   // every thread has taken its results: the queue storage becomes the rows
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

void lio_launch_iterate_cert(const LioSplitParams& S, int n_blocks, hipStream_t s)
{
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(k_s2m_iterate_cert, dim3(n_blocks), dim3(LIO_BLOCK), 0, s, S);
}
