// lio_reuse.hip -- k_s2m_iterate_reuse (cfg.pipeline = 5): k_s2m_iterate with the plane fit done only where the
// neighbour tuple changed.  Same results as k_s2m_iterate, bit for bit.
// MO = /root/reference/src/liorf/src/mapOptmization.cpp.  Compile with -ffp-contract=off.
//
// The plane of surfOptimization (MO:1642-1666: column-pivoted QR on the five neighbours, normalisation, the 0.2 m test)
// depends on the ORDERED tuple of the five nearest map points only, not on the pose.  From the second or third
// Gauss-Newton iteration on most points keep their tuple, and the fit is ~40 % of an iteration's vector instructions.
// Per workgroup (256 consecutive points of one scan):
//   A  transform, ownership, cells; B  the exact 5-NN by the bounded candidate scan -- every thread, as in k_s2m_iterate;
//   C  a point whose ordered tuple equals the cached one takes the cached plane; the others are compacted through LDS
//      and fitted by the FIRST threads of the workgroup (one or two dense waves instead of four sparse ones; the idle
//      waves leave their issue slots to the other workgroups of the CU), which also rewrite the cache;
//   D  weight, coefficients, Jacobian row, fp64 sums, arrival, in-launch solve: k_s2m_iterate's arithmetic and order.
// The certificate kernels (lio_cert.hip) tried to skip the SCAN as well and paid for re-measuring the cached neighbours;
// this one keeps the scan and only drops repeated fits.
#include "lio_s2m_device.h"

__global__ __launch_bounds__(LIO_BLOCK, 5) void k_s2m_iterate_reuse(LioSplitParams S)
{
    const LioIterParams& P = S.it;
    // 16 KiB shared by the queue of phase C and, afterwards, the Jacobian rows of phase D
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[LIO_BLOCK * 64];
    double (*s_rows)[8] = reinterpret_cast<double (*)[8]>(s_raw);
    int (*s_qnn)[6] = reinterpret_cast<int (*)[6]>(s_raw);               // [256] queue: five neighbours + owner thread
    float4* s_plane = reinterpret_cast<float4*>(s_raw + 6144);           // [256] per owner: the plane
    int* s_pstate = reinterpret_cast<int*>(s_raw + 10240);               // [256] per owner: 1 plane valid, 2 not valid
    __shared__ double s_part[8][28];
    __shared__ double s_sum[28];
    __shared__ LioSolveWs s_ws;
    __shared__ int s_cnt;

    int wg = blockIdx.x;
    if (P.xcd_remap) {
        const int n8 = gridDim.x >> 3;
        if (wg < n8 * 8) wg = (wg & 7) * n8 + (wg >> 3);
    }
    const int wg_mode = P.blk_skip != nullptr ? (int)P.blk_skip[wg] : 0;   // map sharding, see k_shard_cull
    if (wg_mode == 1) return;
    const LioBlockDesc bd = P.blocks[wg];
    LioScanState* st = &P.state[bd.scan];
    if (st->done) return;                                  // workgroup-uniform
    float T[12], tr[6], Tp[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = st->T[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) tr[k] = st->trig[k];
    const int n_pts = st->n_pts, base = st->offset;
    const bool record = (P.rec_flag != nullptr) && (st->iter == P.c.record_iter);
    const bool warm = st->iter > 0;                        // every registration starts cold (no bound, no kept plane)
    const bool use_cache = (P.d5_cache != nullptr) && warm;
#pragma unroll
    for (int k = 0; k < 12; ++k) Tp[k] = use_cache ? st->Tp[k] : 0.0f;
    const LioGrid g = P.grid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_cnt = 0;

    // ---- A: transform (pointAssociateToMap, MO:841-847), ownership, cells
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
    const int cx = lio_cell_coord(qx, g.ox, g.inv_cell, g.nx), cy = lio_cell_coord(qy, g.oy, g.inv_cell, g.ny),
              cz = lio_cell_coord(qz, g.oz, g.inv_cell, g.nz);
    act = act && (fabsf(qx) <= 3.0e38f) && (fabsf(qy) <= 3.0e38f) && (fabsf(qz) <= 3.0e38f);
    act = act && cx >= -g.k && cx < g.nx + g.k && cy >= -g.k && cy < g.ny + g.k && cz >= -g.k && cz < g.nz + g.k;

    // ---- B: exact 5-NN (MO:1631) inside the search bound of the previous iteration, as in k_s2m_iterate
    float bound2 = P.c.max_sq_dist;
    int xlo = -0x7fffffff, xhi = 0x7fffffff;
    const int cs = base + bd.first + (int)threadIdx.x;                   // slot in the batch SoA
    if (use_cache && act) {
        const float d5 = P.d5_cache[cs];
        if (d5 >= 0.0f) {
            const float ox = Tp[0] * px + Tp[1] * py + Tp[2]  * pz + Tp[3];
            const float oy = Tp[4] * px + Tp[5] * py + Tp[6]  * pz + Tp[7];
            const float oz = Tp[8] * px + Tp[9] * py + Tp[10] * pz + Tp[11];
            const float mv = sqrtf(lio_sqdist(qx, qy, qz, ox, oy, oz));
            const float R = (sqrtf(d5) + mv) * 1.0001f + 1e-6f;
            const float r2 = R * R * 1.0001f;
            if (r2 < bound2) {
                bound2 = r2;
                xlo = lio_cell_coord(qx - R, g.ox, g.inv_cell, g.nx);
                xhi = lio_cell_coord(qx + R, g.ox, g.inv_cell, g.nx);
            }
        }
    }
    const double sentinel = lio_make_key(bound2, -1);
    LioTop5 top = { sentinel, sentinel, sentinel, sentinel, sentinel };
    if (act) lio_knn_global(P, g, qx, qy, qz, cx, cy, cz, xlo, xhi, top);
    bool ok = act && (lio_key_d2(top.k4) < P.c.max_sq_dist);             // gate MO:1641
    const int nn[5] = { lio_key_idx(top.k0), lio_key_idx(top.k1), lio_key_idx(top.k2), lio_key_idx(top.k3), lio_key_idx(top.k4) };
    if (P.d5_cache && inr) P.d5_cache[cs] = ok ? lio_key_d2(top.k4) : -1.0f;

    // ---- C: the plane -- kept if the ordered tuple is the cached one, fitted (compacted) otherwise
    float pa = 0.0f, pb = 0.0f, pc = 0.0f, pd = 0.0f;
    bool planeValid = false, need_plane = ok;
    if (ok && warm) {
        const int pstate = S.plane_state[cs];
        if (pstate != 0) {
            const int4 ia = reinterpret_cast<const int4*>(S.cache_idx)[(size_t)cs * 2];
            const int i4 = S.cache_idx[(size_t)cs * 8 + 4];
            if (ia.x == nn[0] && ia.y == nn[1] && ia.z == nn[2] && ia.w == nn[3] && i4 == nn[4]) {
                const float4 pl = S.plane[cs];
                pa = pl.x; pb = pl.y; pc = pl.z; pd = pl.w;
                planeValid = pstate == 1;
                need_plane = false;
            }
        }
    }
    __syncthreads();                                                   // s_cnt = 0 is visible
    {
        const unsigned long long m = __ballot(need_plane);
        int wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&s_cnt, (int)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (need_plane) {
            const int pos = wbase + (int)__popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
            for (int j = 0; j < 5; ++j) s_qnn[pos][j] = nn[j];
            s_qnn[pos][5] = (int)threadIdx.x;
        }
    }
    __syncthreads();
    const int cnt = s_cnt;                                             // workgroup-uniform
    if ((int)threadIdx.x < cnt) {
        int n5[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) n5[j] = s_qnn[threadIdx.x][j];
        const int owner = s_qnn[threadIdx.x][5];
        const int co = base + bd.first + owner;
        float qa, qb, qc, qd;
        bool valid;
        lio_plane_from_nn(P, n5, qa, qb, qc, qd, valid);               // MO:1642-1666
        const float4 pl = make_float4(qa, qb, qc, qd);
        S.plane[co] = pl;
        S.plane_state[co] = valid ? 1 : 2;
        reinterpret_cast<int4*>(S.cache_idx)[(size_t)co * 2] = make_int4(n5[0], n5[1], n5[2], n5[3]);
        S.cache_idx[(size_t)co * 8 + 4] = n5[4];
        s_plane[owner] = pl;
        s_pstate[owner] = valid ? 1 : 2;
    }
    if (cnt > 0) __syncthreads();
    if (need_plane) {
        const float4 pl = s_plane[threadIdx.x];
        pa = pl.x; pb = pl.y; pc = pl.z; pd = pl.w;
        planeValid = s_pstate[threadIdx.x] == 1;
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
    __syncthreads();                                                   // every thread has taken its plane: the queue storage becomes the rows
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

void lio_launch_iterate_reuse(const LioSplitParams& S, int n_blocks, hipStream_t s)
{
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(k_s2m_iterate_reuse, dim3(n_blocks), dim3(LIO_BLOCK), 0, s, S);
}
