// lio_persist.hip -- k_s2m_persist: the whole Gauss-Newton loop of scan2MapOptimization (MO:1848-1859) in ONE launch
// (cfg.pipeline = 4), for batches whose association workgroups are all resident at once (a single registration is
// ~26 workgroups on 256 CUs).  Same results as the launch loop of k_s2m_iterate, bit for bit: the per-iteration body is
// the same arithmetic in the same order (phase A transform + cells, bounded exact 5-NN, lio_assoc_point, Jacobian row,
// LDS-transposed fp64 sums, write-through partials, arrival counter, solve by the last workgroup of the scan).
//
// What replaces the kernel boundary between two iterations is a PER-SCAN barrier, not a grid-wide one: the workgroups
// of scan s only wait for scan s's solve.  The solving wave re-arms the arrival counter, releases (agent scope: its
// XCD's L2 is written back) and publishes gen[s] = iterations done; wave 0 of every other workgroup of the scan polls
// gen[s] (epoch + iterations done, so that nothing needs clearing between runs) with write-through loads, then the whole workgroup acquires (L1 / non-local L2 lines invalidated) and re-reads
// the scan's state.  Every poll loop is bounded: after `spin_max` polls (a few milliseconds) the workgroup leaves with the
// scan's `done` flag still clear, so every wave reaches an exit whatever happens -- the workgroups of a launch are only
// guaranteed to make progress together when they are all resident, which another client of the GPU can prevent.  The host
// finds a scan that is not done after the launch, clears the arrival counters and runs the SAME registration through the
// launch loop from the saved initial guess inside the same call (lio_s2m_batch_results; counted in
// lio_s2m_profile.persist_fallbacks): a contended GPU costs a few milliseconds, never a wrong or missing result.
//
// Speculation on isDegenerate.  The first solve of a registration carries cv::eigen for isDegenerate / matP (MO:1786-1808),
// ~36 us of Jacobi rotations that the next pose formally depends on -- but only if a direction IS degenerate.  So the
// first solve publishes the non-degenerate update right after the QR solve and hands the eigen-decomposition to a helper
// workgroup (one per scan, behind the association workgroups in the grid), which writes isDegenerate / matP and answers
// through spec[scan].  Later solves look at the answer: not there yet -> carry on with isDegenerate = 0, unless the step
// would END the registration, which waits for the answer; "not degenerate" -> nothing to repair; "degenerate" -> roll the
// scan back to its first solve (saved sums, initial guess) and redo that step the plain way, the iterations in between are
// discarded.  Results are the launch loop's in every case (tests/test_gpu_persist.py: corridor scenes take the roll-back).
#include <hip/hip_runtime.h>
#ifndef LIO_PREFETCH
#define LIO_PREFETCH 2       // a lone registration runs ONE wave per SIMD: nothing else hides the candidate loads, so two groups are
                             // kept in flight (the batched kernel, five waves per SIMD, is faster with one; lio_s2m_device.h)
#endif
#include "lio_kernels.h"
#include "lio_device_math.h"
#include "lio_s2m_device.h"

namespace {

__global__ __launch_bounds__(LIO_BLOCK, 2) void k_s2m_persist(LioIterParams P, unsigned* __restrict__ gen, unsigned epoch, const unsigned char* __restrict__ stage, unsigned stride,
                   int n_main, unsigned* __restrict__ spec, double* __restrict__ spec_sums, const float* __restrict__ poses0,
                   unsigned spin_max, int withhold_wg)
{
    __shared__ __attribute__((aligned(16))) double s_rows[LIO_BLOCK][8];
    __shared__ double s_part[8][28];
    __shared__ double s_sum[28];
    __shared__ LioSolveWs s_ws;
    __shared__ int s_ctl;                                  // 0 = next iteration, 1 = the scan is done, 2 = poll timed out

    const int wg = blockIdx.x;                             // (no XCD remap: a registration's few workgroups are spread over the XCDs as dealt)
    if (wg >= n_main) {
        // ---- helper of scan `wg - n_main`: the eigen-decomposition / inverse / product of the first solve, off the critical path
        const int scan = wg - n_main;
        if (threadIdx.x >= 64) return;
        const int hl = threadIdx.x;
        LioScanState* hs = &P.state[scan];
        if (hs->done) return;                              // (refused at init: nothing will ever be asked)
        unsigned spins = 0;
        for (;;) {
            const int v = (int)(__hip_atomic_load(&spec[scan], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch);
            if (v >= 1) break;                             // asked
            if (__hip_atomic_load(&hs->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the registration ended without asking
            if (++spins > spin_max * 2u) return;
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (hl == 0) {                                     // matAtA as LMOptimization builds it from the sums of the first iteration
            const double* sm = spec_sums + (size_t)scan * LIO_SUMS;
            int p = 0;
            for (int a = 0; a < 6; ++a)
                for (int b = a; b < 6; ++b) {
                    const float v = (float)__hip_atomic_load(sm + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ++p;
                    s_ws.AtA[a * 6 + b] = v; s_ws.AtA[b * 6 + a] = v;
                }
        }
        LIO_LDS_FENCE();
        float Wd[6], vrow;
        lio_eigen6_sym_lanes(s_ws.AtA, Wd, vrow, hl);       // cv::eigen, MO:1792
        bool zr[6];
        bool z = true;
#pragma unroll
        for (int i = 5; i >= 0; --i) { z = z && (Wd[i] < P.c.eig_thresh); zr[i] = z; }     // MO:1794-1805
        const int deg = zr[5] ? 1 : 0;
        if (hl < 36) {
            const int r = hl / 6;
            bool zero = zr[0];
#pragma unroll
            for (int i = 1; i < 6; ++i) if (r == i) zero = zr[i];
            s_ws.V2[hl] = zero ? 0.0f : vrow;
        }
        lio_inv6_lu_wave(vrow, s_ws.B, hl);                 // matV.inv(), MO:1807
        LIO_LDS_FENCE();
        lio_gemm6_wave(s_ws.B, s_ws.V2, s_ws.A, hl);        // matP = matV.inv() * matV2
        LIO_LDS_FENCE();
        if (hl < 36) hs->matP[hl] = s_ws.A[hl];
        if (hl == 0) hs->is_degenerate = deg;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (hl == 0) __hip_atomic_store(&spec[scan], epoch + 2u + (unsigned)deg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (wg == withhold_wg) return;                         // test hook (lio_s2m_debug_persist_spin): this workgroup never arrives
    const LioBlockDesc bd = P.blocks[wg];
    LioScanState* st = &P.state[bd.scan];
    const LioGrid g = P.grid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_pts = st->n_pts, base = st->offset;
    const int li = bd.first + (int)threadIdx.x;
    const bool inr = li < n_pts;
    const int gi = base + (inr ? li : 0);
    // the scan point never changes: loaded once -- straight from the staged records when the batch was not tile-sorted
    // (the AoS -> SoA launch is skipped for it, see lio_s2m_batch_upload)
    float px, py, pz;
    if (stage) {
        const float* rec = reinterpret_cast<const float*>(stage + (size_t)gi * stride);
        px = rec[0]; py = rec[1]; pz = rec[2];
    } else {
        px = P.sx[gi]; py = P.sy[gi]; pz = P.sz[gi];
    }
    const int ci = base + bd.first + (int)threadIdx.x;                   // slot in the batch SoA
    const int red_g = threadIdx.x >> 5, red_s = threadIdx.x & 31;
    const int red_a = c_pair_a[red_s], red_b = c_pair_b[red_s];
    if (st->done) return;                                                // (too few points: k_s2m_init_state) workgroup-uniform
    // diagnostic phase clock (cfg.profile = 2): time of wave 0 of every workgroup, summed over the iterations, in 10 ns ticks:
    // [0] state + transform, [1] candidate scan, [2] plane + row, [3] sums, [4] partial + arrival, [5] solve (solving
    // workgroup only), [6] waiting at the scan's barrier, [7] acquire
    long long* stamp = (P.stamps && wave == 0 && lane == 0) ? P.stamps + (size_t)wg * (LIO_BLOCK / 64) * 8 : nullptr;
    long long t_prev = stamp ? (long long)wall_clock64() : 0;
#define LIO_TICK(k) do { if (stamp) { const long long t_ = (long long)wall_clock64(); stamp[k] += t_ - t_prev; t_prev = t_; } } while (0)

#pragma unroll 1
    for (int it = 0;; ++it) {
        // per-scan values of this iteration (fresh: kernel start, or acquired at the bottom of the previous trip)
        float T[12], tr[6], Tp[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = st->T[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) tr[k] = st->trig[k];
        const int st_iter = st->iter;
        const bool record = (P.rec_flag != nullptr) && (st_iter == P.c.record_iter);
        const bool use_cache = (P.d5_cache != nullptr) && st_iter > 0;
#pragma unroll
        for (int k = 0; k < 12; ++k) Tp[k] = use_cache ? st->Tp[k] : 0.0f;

        // ---- phase A: pointAssociateToMap MO:841-847, cells
        const float qx = T[0] * px + T[1] * py + T[2]  * pz + T[3];
        const float qy = T[4] * px + T[5] * py + T[6]  * pz + T[7];
        const float qz = T[8] * px + T[9] * py + T[10] * pz + T[11];
        const int cx = lio_cell_coord(qx, g.ox, g.inv_cell, g.nx);
        const int cy = lio_cell_coord(qy, g.oy, g.inv_cell, g.ny);
        const int cz = lio_cell_coord(qz, g.oz, g.inv_cell, g.nz);
        bool act = inr;
        act = act && (fabsf(qx) <= 3.0e38f) && (fabsf(qy) <= 3.0e38f) && (fabsf(qz) <= 3.0e38f);
        act = act && cx >= -g.k && cx < g.nx + g.k && cy >= -g.k && cy < g.ny + g.k && cz >= -g.k && cz < g.nz + g.k;

        LIO_TICK(0);
        // ---- exact 5-NN (MO:1631) inside the search bound of the previous iteration (see k_s2m_iterate)
        float bound2 = P.c.max_sq_dist;
        float Rx = sqrtf(P.c.max_sq_dist) * 1.0001f + 1e-6f;              // reach along x: the gate, unless the bound below is tighter
        bool bounded = false;
        if (use_cache && act) {
            const float d5 = P.d5_cache[ci];
            if (d5 >= 0.0f) {
                const float ox = Tp[0] * px + Tp[1] * py + Tp[2]  * pz + Tp[3];
                const float oy = Tp[4] * px + Tp[5] * py + Tp[6]  * pz + Tp[7];
                const float oz = Tp[8] * px + Tp[9] * py + Tp[10] * pz + Tp[11];
                const float mv = sqrtf(lio_sqdist(qx, qy, qz, ox, oy, oz));
                const float R = (sqrtf(d5) + mv) * 1.0001f + 1e-6f;
                const float r2 = R * R * 1.0001f;
                if (r2 < bound2) { bound2 = r2; Rx = R; bounded = true; }
            }
        }
        const double sentinel = lio_make_key(bound2, -1);
        LioTop5 top = { sentinel, sentinel, sentinel, sentinel, sentinel };
        if (act) lio_knn_global(P, g, qx, qy, qz, cy, cz, Rx, bound2, bounded, top);
        const bool ok = act && (lio_key_d2(top.k4) < P.c.max_sq_dist);       // gate MO:1641
        const int nn[5] = { lio_key_idx(top.k0), lio_key_idx(top.k1), lio_key_idx(top.k2), lio_key_idx(top.k3), lio_key_idx(top.k4) };
        if (P.d5_cache && inr) P.d5_cache[ci] = ok ? lio_key_d2(top.k4) : -1.0f;
        LIO_TICK(1);

        // ---- plane, weight, coefficients MO:1642-1683
        float cxx = 0.0f, cyy = 0.0f, czz = 0.0f, cww = 0.0f;
        bool accept = false;
        if (ok) accept = lio_assoc_point<false>(P, nn, qx, qy, qz, px, py, pz, cxx, cyy, czz, cww);
        if (record && inr) {
            const int oi = P.perm ? P.perm[base + li] : base + li;          // the record is kept in the CALLER's point order
            P.rec_flag[oi] = accept ? 1 : 0;
            reinterpret_cast<float4*>(P.rec_coeff)[oi] = make_float4(cxx, cyy, czz, cww);
#pragma unroll
            for (int j = 0; j < 5; ++j) P.rec_nn[(size_t)oi * 5 + j] = ok ? nn[j] : -1;
        }
        // ---- row of matA / matB MO:1735-1778, sums MO:1781-1783 in k_s2m_iterate's order
        float row[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, rhs = 0.0f;
        if (accept) lio_jacobian_row(tr, px, py, pz, cxx, cyy, czz, cww, P.c.jac_exact, row, rhs);
        {
            double2* dst = reinterpret_cast<double2*>(s_rows[threadIdx.x]);
            dst[0] = make_double2((double)row[0], (double)row[1]);
            dst[1] = make_double2((double)row[2], (double)row[3]);
            dst[2] = make_double2((double)row[4], (double)row[5]);
            dst[3] = make_double2((double)rhs, accept ? 1.0 : 0.0);
        }
        __syncthreads();
        LIO_TICK(2);
        double red_acc = 0.0;
        if (red_s < 28) {
#pragma unroll 8
            for (int p = red_g; p < LIO_BLOCK; p += 8)
                red_acc = __builtin_fma(s_rows[p][red_a], s_rows[p][red_b], red_acc);
            s_part[red_g][red_s] = red_acc;
        }
        __syncthreads();
        LIO_TICK(3);

        if (wave == 0) {
            double* part = P.partials + ((size_t)bd.scan * P.max_blk + bd.blk) * LIO_SUMS;
            if (lane < 28) {
                double v = s_part[0][lane];
#pragma unroll
                for (int w = 1; w < 8; ++w) v += s_part[w][lane];
                __hip_atomic_store(part + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
            }
            // ORDERING: the 28 write-through (sc1) stores above have left this wave's vector-memory queue, i.e. they are
            // complete at the agent-coherent level, before the arrival below can be observed -- the release half of the
            // arrive protocol, for these stores only (cheaper than an agent-scope release fence, which writes back the whole L2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int last = 0;
            if (lane == 0) {
                const unsigned old = atomicAdd(&P.arrive[bd.scan], 1u);
                last = (old == (unsigned)bd.n_blk - 1u);
            }
            last = __shfl(last, 0);
            LIO_TICK(4);
            if (last) {
                // last workgroup of the scan in this iteration: fixed-order sum over the chunks, then LMOptimization
                if (lane < 28) {
                    const double* base_p = P.partials + (size_t)bd.scan * P.max_blk * LIO_SUMS + lane;
                    double v = 0.0;
                    for (int b = 0; b < bd.n_blk; b += 32) {        // 32 write-through loads in flight (a lone scan has ~26 chunks), summed in chunk order
                        double t[32];
#pragma unroll
                        for (int u = 0; u < 32; ++u)
                            t[u] = __hip_atomic_load(base_p + (size_t)min(b + u, bd.n_blk - 1) * LIO_SUMS,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int u = 0; u < 32; ++u) v += (b + u < bd.n_blk) ? t[u] : 0.0;
                    }
                    s_sum[lane] = v;
                }
                // ORDERING: the LDS writes of s_sum by lanes 0..27 are complete before lane 0 (same wave) reads them in lio_gn_step
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(&P.arrive[bd.scan], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
                // (solver detail of the phase clock, slots 8-14: gather / step / release of iteration 0 and of the later ones, solves counted)
                long long t_s = stamp ? (long long)wall_clock64() : 0;
                const int so = it == 0 ? 8 : 11;
                if (stamp) { stamp[so] += t_s - t_prev; if (it) stamp[14] += 1; }
                bool ask = false, lost = false;
                // a discarded pass must not leave a recorded association behind: speculate only when none can be recorded after pass 0
                const bool may_spec = spec != nullptr && (P.rec_flag == nullptr || P.c.record_iter <= 0);
                const int st_it = st->iter;                                  // (0 also after a roll-back, which settles the speculation first)
                int sp = spec ? (int)(__hip_atomic_load(&spec[bd.scan], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) : 0;
                if (sp < 0 || sp > 4) sp = 0;                                // 0 none, 1 asked, 2 answer: not degenerate, 3 answer: degenerate, 4 settled
                const bool first_spec = st_it == 0 && sp == 0 && may_spec;
                bool plain = true;
                if (first_spec || sp == 1) {
                    // first solve: the non-degenerate update now, the eigen-decomposition by the helper; later solves without an
                    // answer yet: carry on as "not degenerate".  Either way a step that would END the registration is held back.
                    if (lio_gn_step(st, s_sum, P.c, &s_ws, P.n_active, lane, 0, first_spec, true) == 2) {
                        if (!first_spec) {                                   // wait for the answer, then decide below
                            unsigned spins = 0;
                            while (sp == 1) {
                                sp = (int)(__hip_atomic_load(&spec[bd.scan], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch);
                                if (++spins > spin_max) { lost = true; break; }
                                __builtin_amdgcn_s_sleep(4);
                            }
                        }
                    } else {
                        plain = false;                                       // committed (speculatively)
                        if (first_spec) {
                            if (lane < 28) __hip_atomic_store(spec_sums + (size_t)bd.scan * LIO_SUMS + lane, s_sum[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ask = true;
                        }
                    }
                }
                if (plain && !lost) {
                    if (sp == 2 || sp == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the helper's isDegenerate / matP
                    if (sp == 3) {
                        // a degenerate direction: back to the first solve.  Initial guess, iteration counter and the records of the
                        // discarded passes are restored; st->T stays (it becomes Tp, which is what the search bounds of the last pass refer to)
                        if (lane == 0) {
                            for (int k = 0; k < 6; ++k) st->pose[k] = poses0[bd.scan * 6 + k];
                            for (int j = 1; j <= st_it && j < 32; ++j) {
                                st->n_corr_iter[j] = 0;
                                for (int k = 0; k < 6; ++k) st->pose_iter[j][k] = 0.0f;
                            }
                            st->iter = 0;
                            st->converged = 0;
                        }
                        if (lane < 28) s_sum[lane] = __hip_atomic_load(spec_sums + (size_t)bd.scan * LIO_SUMS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    }
                    lio_gn_step(st, s_sum, P.c, &s_ws, P.n_active, lane);        // the plain step: settled, answered, never asked, or ending
                    if (sp == 3 && lane == 0) __hip_atomic_store(&spec[bd.scan], epoch + 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (stamp) { const long long t_ = (long long)wall_clock64(); stamp[so + 1] += t_ - t_s; t_s = t_; }
                // publish: the state written above becomes visible to the other XCDs before the generation number does
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (stamp) { const long long t_ = (long long)wall_clock64(); stamp[so + 2] += t_ - t_s; }
                if (lane == 0 && ask) __hip_atomic_store(&spec[bd.scan], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0 && !lost) __hip_atomic_store(&gen[bd.scan], epoch + (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (last) LIO_TICK(5);
            // per-scan barrier: wait for this iteration's solve (bounded)
            int ctl = 0;
            if (lane == 0) {
                unsigned spins = 0;
                while ((int)(__hip_atomic_load(&gen[bd.scan], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (epoch + (unsigned)(it + 1))) < 0) {
                    if (++spins > spin_max) { ctl = 2; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");         // pairs with the solver's release
                s_ctl = ctl;
            }
        }
        __syncthreads();
        LIO_TICK(6);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                 // the solver's writes to *st, not stale cache lines
        if (s_ctl == 2) return;
        if (__hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // MO:1857-1858 / MO:1848, workgroup-uniform
        if (it + 1 >= 2 * 32 + 2) return;                                  // (LIO_MAX_ITERS passes, at most doubled by one roll-back; `done` ends the loop before this)
        __syncthreads();                                                   // s_ctl / s_part are rewritten by the next trip
        LIO_TICK(7);
    }
#undef LIO_TICK
}

}  // namespace

// `epoch`: a number that grows by at least 128 from one launch on these buffers to the next (generation numbers and
// speculation states of a run are epoch + small numbers, compared modulo 2^32), so nothing has to be cleared between runs.
// spec != nullptr: n_scans helper workgroups follow the n_blocks association workgroups (see "Speculation" above).
// spin_max: polls before a waiting workgroup gives up (a poll is a short sleep + one L2-bypassing load, ~1-2 us);
// withhold_wg: test hook, -1 in production.
void lio_launch_persist(const LioIterParams& P, int n_blocks, unsigned* gen, unsigned epoch, const unsigned char* stage, size_t stride,
                        int n_scans, unsigned* spec, double* spec_sums, const float* poses0, unsigned spin_max, int withhold_wg, hipStream_t s)
{
    if (n_blocks <= 0) return;
    const int n_help = spec ? n_scans : 0;
    hipLaunchKernelGGL(k_s2m_persist, dim3(n_blocks + n_help), dim3(LIO_BLOCK), 0, s, P, gen, epoch, stage, (unsigned)stride,
                       n_blocks, spec, spec_sums, poses0, spin_max, withhold_wg);
}
