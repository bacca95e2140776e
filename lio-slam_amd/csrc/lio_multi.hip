// lio_multi.hip -- in-library multi-GPU mode of the scan-to-map path (cfg.n_devices > 1).
// MO = /root/reference/src/liorf/src/mapOptmization.cpp.
//
// The handle returned by lio_s2m_create is a front for one child handle per entry of cfg.device_ids.  lio_s2m_set_map cuts
// the map into slabs of 1.001 m cells along its longest axis, balanced by point count; every child HOLDS its slab plus a
// halo of LIO_MULTI_HALO cells on either side (~16 m of extra map per side) and OWNS the scan points whose transformed
// position falls into a cell of its slab (the exact per-point test, or whole workgroups through k_shard_cull).  The only
// coupling between the devices is the per-scan sum of the 6x6 JtJ / 6x1 Jtr / N_c -- the join the reference gets from its
// OpenMP barrier at MO:1622-1686 -- exchanged once per Gauss-Newton iteration:
//
//   device c, own stream:   association (k_shard_cull + k_s2m_iterate) -> partial sums d_part[c]
//                           k_multi_publish: ONE kernel stores d_part[c] into slot c of EVERY device's gather buffer
//                           (peer-to-peer stores over xGMI; hipDeviceEnablePeerAccess at create time)  -> event pub[c][it]
//   device p, own stream:   hipStreamWaitEvent on pub[c][it] of every other device c (four devices and more: on ONE event that a
//                           joining stream records after waiting for all of them -- 2N + 1 host calls instead of N(N-1))
//                           k_s2m_apply: adds the slots IN DEVICE ORDER (bitwise reproducible, identical on every
//                           device) and runs LMOptimization MO:1702-1837 for every scan -> identical poses everywhere
//
// No device or stream is synchronised inside the loop and nothing passes through host memory: the host thread only
// enqueues, `lookahead` iterations ahead of the last convergence count it has seen (a 4-byte pinned word per iteration,
// waited for with an EVENT, as in the single-device launch loop).  The gather buffers are double-buffered by iteration
// parity: device c stores iteration it+2 only after its own apply of it+1, which waited for every device's publish of
// it+1, which is ordered behind that device's apply of `it` on its stream -- so nobody overwrites a slot that is still
// being read.  The same device may be listed more than once (that is how the mode is tested on a one-GPU box: the "peer"
// pointers are then local and the children are separate streams of one device).  Where peer access is refused
// (hipDeviceCanAccessPeer == 0) the publish falls back to one hipMemcpyPeerAsync per destination; LIO_MULTI_EXCHANGE=copy
// forces that form (tests), LIO_MULTI_EXCHANGE=host the round-2 exchange through pinned host memory with a stream
// synchronisation per device and iteration (kept for A/B timing only).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "lio_handle.h"
#include "lio_multi.h"

#define LIO_MULTI_HALO 16      // cells of map a device holds beyond its slab: workgroups up to ~30 m long stay whole (k_shard_cull)

struct LioMulti {
    std::vector<lio_s2m_handle*> dev;
    std::vector<double*> d_part;                      // per child: its partial sums [n_scans][LIO_SUMS]
    std::vector<double*> d_gather;                    // per child: [2 parities][n_dev slots][cap_scans][LIO_SUMS]
    std::vector<double*> h_part;                      // (host exchange only) pinned partial sums per child
    double* h_tot = nullptr;                          // (host exchange only)
    std::vector<double*> d_tot;                       // (host exchange only)
    std::vector<hipEvent_t> ev_pub;                   // [n_dev][LIO_MAX_ITERS]
    hipStream_t join_stream = nullptr;                // four devices and more: ONE stream waits for all publish events of an iteration
    std::vector<hipEvent_t> ev_all;                   // [LIO_MAX_ITERS] ... and records "everybody has published" for every device to wait on
    size_t cap_scans = 0;
    std::vector<std::vector<int>> shard_idx;          // per child: caller's map index of every point of its shard
    std::vector<unsigned char> gather;                // host staging of one shard's records
    int n_scans = 0;
    int exchange = 0;                                 // 0 = peer stores from a kernel, 1 = hipMemcpyPeerAsync, 2 = through the host (round 2)
    bool peer_ok = true;
    // diagnostics of the last run (lio_s2m_profile.multi_*)
    int stream_syncs = 0, event_waits = 0, iterations = 0;
};

namespace {

struct LioPeerPtrs { double* p[8]; };

// Device c's partial sums into slot c of every device's gather buffer (dst.p[d] already points at that slot).
__global__ void k_multi_publish(const double* __restrict__ part, int n_val, LioPeerPtrs dst, int n_dev)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_val) return;
    const double v = part[i];
#pragma unroll 1
    for (int d = 0; d < n_dev; ++d) dst.p[d][i] = v;
}

}  // namespace

static int lio_multi_reserve(lio_s2m_handle* h, size_t n_scans)
{
    LioMulti* m = h->multi;
    if (n_scans <= m->cap_scans) return LIO_OK;
    const size_t nd = m->dev.size();
    const size_t cap = n_scans + n_scans / 4 + 16, bytes = cap * LIO_SUMS * sizeof(double);
    for (size_t c = 0; c < nd; ++c) {
        HIPCHK(hipSetDevice(m->dev[c]->cfg.device_id));
        HIPCHK(hipStreamSynchronize(m->dev[c]->stream));
        if (m->d_part[c]) HIPCHK(hipFree(m->d_part[c]));
        if (m->d_gather[c]) HIPCHK(hipFree(m->d_gather[c]));
        if (m->d_tot[c]) HIPCHK(hipFree(m->d_tot[c]));
        if (m->h_part[c]) HIPCHK(hipHostFree(m->h_part[c]));
        m->d_part[c] = m->d_gather[c] = m->d_tot[c] = m->h_part[c] = nullptr;
        HIPCHK(hipMalloc((void**)&m->d_part[c], bytes));
        HIPCHK(hipMalloc((void**)&m->d_gather[c], 2 * nd * bytes));
        HIPCHK(hipMalloc((void**)&m->d_tot[c], bytes));
        if (m->exchange == 2) HIPCHK(hipHostMalloc((void**)&m->h_part[c], bytes, hipHostMallocPortable));
    }
    if (m->h_tot) HIPCHK(hipHostFree(m->h_tot));
    m->h_tot = nullptr;
    if (m->exchange == 2) HIPCHK(hipHostMalloc((void**)&m->h_tot, bytes, hipHostMallocPortable));
    m->cap_scans = cap;
    return LIO_OK;
}

int lio_multi_create(const lio_s2m_config* cfg, lio_s2m_handle** out)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lio_fail(LIO_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (cfg->n_devices > 8) return lio_fail(LIO_ERR_ARG, "n_devices must be <= 8");
    for (int i = 0; i < cfg->n_devices; ++i)
        if (cfg->device_ids[i] < 0 || cfg->device_ids[i] >= ndev) return lio_fail(LIO_ERR_ARG, "device_ids entry out of range");
    if (cfg->use_lds || cfg->kernel_variant > 1)
        return lio_fail(LIO_ERR_ARG, "the multi-device mode runs the default kernel only");
    lio_s2m_handle* f = new lio_s2m_handle();
    f->cfg = *cfg;
    f->shard.axis = -1;
    LioMulti* m = f->multi = new LioMulti();
    const char* ex = getenv("LIO_MULTI_EXCHANGE");
    m->exchange = (ex && !strcmp(ex, "copy")) ? 1 : ((ex && !strcmp(ex, "host")) ? 2 : 0);
    for (int i = 0; i < cfg->n_devices; ++i) {
        lio_s2m_config cc = *cfg;
        cc.n_devices = 1;
        cc.device_id = cfg->device_ids[i];
        cc.use_graph = 0;                        // the loop is driven iteration by iteration (one exchange each)
        cc.pipeline = 1;
        lio_s2m_handle* ch = nullptr;
        const int rc = lio_s2m_create(&cc, &ch);
        if (rc != LIO_OK) { lio_multi_destroy(f); return rc; }
        m->dev.push_back(ch);
        m->d_part.push_back(nullptr); m->d_gather.push_back(nullptr); m->d_tot.push_back(nullptr); m->h_part.push_back(nullptr);
    }
    m->shard_idx.resize((size_t)cfg->n_devices);
    // peer access between every pair of DISTINCT devices (a repeated ordinal needs none: its pointers are local)
    for (int a = 0; a < cfg->n_devices && m->exchange == 0; ++a)
        for (int b = 0; b < cfg->n_devices; ++b) {
            const int da = cfg->device_ids[a], db = cfg->device_ids[b];
            if (da == db) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) { m->peer_ok = false; continue; }
            if (hipSetDevice(da) != hipSuccess) { m->peer_ok = false; continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) m->peer_ok = false;
            (void)hipGetLastError();
        }
    if (!m->peer_ok && m->exchange == 0) m->exchange = 1;
    m->ev_pub.assign((size_t)cfg->n_devices * LIO_MAX_ITERS, nullptr);
    for (int c = 0; c < cfg->n_devices; ++c) {
        if (hipSetDevice(cfg->device_ids[c]) != hipSuccess) { lio_multi_destroy(f); return lio_fail(LIO_ERR_HIP, "hipSetDevice"); }
        for (int i = 0; i < LIO_MAX_ITERS; ++i)
            if (hipEventCreateWithFlags(&m->ev_pub[(size_t)c * LIO_MAX_ITERS + i], hipEventDisableTiming) != hipSuccess) {
                lio_multi_destroy(f);
                return lio_fail(LIO_ERR_HIP, "hipEventCreate");
            }
    }
    if (cfg->n_devices >= 4) {
        // N devices waiting for N-1 events each is N(N-1) host calls per iteration (56 at N = 8); through one joining stream
        // it is N + 1 + N (17): the host thread that enqueues everything is the scarce resource of this mode
        bool ok = hipSetDevice(cfg->device_ids[0]) == hipSuccess && hipStreamCreateWithFlags(&m->join_stream, hipStreamNonBlocking) == hipSuccess;
        m->ev_all.assign(LIO_MAX_ITERS, nullptr);
        for (int i = 0; i < LIO_MAX_ITERS && ok; ++i) ok = hipEventCreateWithFlags(&m->ev_all[(size_t)i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { lio_multi_destroy(f); return lio_fail(LIO_ERR_HIP, "hipStreamCreate / hipEventCreate (join stream)"); }
    }
    *out = f;
    return LIO_OK;
}

void lio_multi_destroy(lio_s2m_handle* h)
{
    LioMulti* m = h->multi;
    for (size_t c = 0; c < m->dev.size(); ++c) {
        (void)hipSetDevice(m->dev[c]->cfg.device_id);
        (void)hipStreamSynchronize(m->dev[c]->stream);
    }
    for (size_t c = 0; c < m->dev.size(); ++c) {
        (void)hipSetDevice(m->dev[c]->cfg.device_id);
        if (m->d_part[c]) (void)hipFree(m->d_part[c]);
        if (m->d_gather[c]) (void)hipFree(m->d_gather[c]);
        if (m->d_tot[c]) (void)hipFree(m->d_tot[c]);
        if (m->h_part[c]) (void)hipHostFree(m->h_part[c]);
        for (int i = 0; i < LIO_MAX_ITERS; ++i) {
            const size_t k = c * LIO_MAX_ITERS + (size_t)i;
            if (k < m->ev_pub.size() && m->ev_pub[k]) (void)hipEventDestroy(m->ev_pub[k]);
        }
        lio_s2m_destroy(m->dev[c]);
    }
    if (m->h_tot) (void)hipHostFree(m->h_tot);
    if (!m->dev.empty()) (void)hipSetDevice(h->cfg.device_ids[0]);
    for (hipEvent_t e : m->ev_all) if (e) (void)hipEventDestroy(e);
    if (m->join_stream) (void)hipStreamDestroy(m->join_stream);
    delete m;
    delete h;
}

// Multi-device set_map: slab plan on the host (the same arithmetic as lio-slam_amd/multigpu.py plan_shards and as the
// device-side owner test), then every child receives its slab + halo in the caller's record layout.
int lio_multi_set_map(lio_s2m_handle* h, const void* pts, size_t n, size_t stride)
{
    LioMulti* m = h->multi;
    const int world = (int)m->dev.size();
    const unsigned char* src = (const unsigned char*)pts;
    const float cell = h->cfg.cell_size > 0.0f ? h->cfg.cell_size : sqrtf(h->cfg.max_sq_dist) * 1.001f;
    const float inv_cell = 1.0f / cell;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < n; ++i) {
        const float* p = (const float*)(src + i * stride);
        for (int a = 0; a < 3; ++a)
            if (fabsf(p[a]) <= 1.0e15f) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
    }
    float origin[3];
    int32_t dims[3];
    int axis = 0;
    for (int a = 0; a < 3; ++a) {
        if (!(mn[a] <= mx[a])) { mn[a] = 0.0f; mx[a] = 0.0f; }
        origin[a] = mn[a] - 0.5f * cell;
        dims[a] = (int32_t)floor(((double)mx[a] - origin[a]) * inv_cell) + 2;
        if (dims[a] > dims[axis]) axis = a;
    }
    auto cell_of = [&](float v) {                          // lio_cell_coord(), clamped to the grid
        float c = floorf((v - origin[axis]) * inv_cell);
        c = fminf(fmaxf(c, -4.0f), (float)(dims[axis] + 3));
        int ci = (int)c;
        return ci < 0 ? 0 : (ci > dims[axis] - 1 ? dims[axis] - 1 : ci);
    };
    std::vector<long long> cum((size_t)dims[axis], 0);
    std::vector<int> pc(n);
    for (size_t i = 0; i < n; ++i) {
        const float* p = (const float*)(src + i * stride);
        const bool ok = fabsf(p[0]) <= 1.0e15f && fabsf(p[1]) <= 1.0e15f && fabsf(p[2]) <= 1.0e15f;
        pc[i] = ok ? cell_of(p[axis]) : -1000;             // non-finite points belong to no slab (they are no neighbour of anything)
        if (ok) cum[(size_t)pc[i]]++;
    }
    for (size_t c = 1; c < cum.size(); ++c) cum[c] += cum[c - 1];
    const long long total = cum.empty() ? 0 : cum.back();
    std::vector<int> bounds((size_t)world + 1, 0);
    bounds[(size_t)world] = dims[axis];
    for (int r = 1; r < world; ++r) {
        const double want = (double)total * r / world;
        size_t lo = 0;
        while (lo < cum.size() && (double)cum[lo] < want) ++lo;   // first cell whose cumulative count reaches the share
        bounds[(size_t)r] = (int)lo + 1;
        if (bounds[(size_t)r] > dims[axis]) bounds[(size_t)r] = dims[axis];
        if (bounds[(size_t)r] < bounds[(size_t)r - 1]) bounds[(size_t)r] = bounds[(size_t)r - 1];
    }
    for (int r = 0; r < world; ++r) {
        const int lo = bounds[(size_t)r], hi = bounds[(size_t)r + 1];
        std::vector<int>& idx = m->shard_idx[(size_t)r];
        idx.clear();
        for (size_t i = 0; i < n; ++i)
            if (pc[i] >= lo - LIO_MULTI_HALO && pc[i] < hi + LIO_MULTI_HALO) idx.push_back((int)i);
        m->gather.resize((idx.size() ? idx.size() : 1) * stride);
        for (size_t k = 0; k < idx.size(); ++k) memcpy(m->gather.data() + k * stride, src + (size_t)idx[k] * stride, stride);
        lio_s2m_handle* ch = m->dev[(size_t)r];
        int rc = lio_s2m_set_map(ch, m->gather.data(), idx.size(), stride);
        if (rc == LIO_OK) rc = lio_s2m_set_global_grid(ch, origin, dims);
        if (rc == LIO_OK) rc = lio_s2m_set_shard_plan(ch, axis, world, r, bounds.data(), LIO_MULTI_HALO);
        if (rc != LIO_OK) return rc;
    }
    h->has_map = true;
    h->n_map = n;
    h->prof = m->dev[0]->prof;
    h->prof.n_map = (int64_t)n;
    return LIO_OK;
}

int lio_multi_upload(lio_s2m_handle* h, int32_t n_scans, const void* const* scans, const size_t* n_pts, size_t stride)
{
    // Every device holds every scan: which device owns a point follows the pose, iteration by iteration.  All H2D
    // copies are in flight together (each device has its own PCIe link), one wait per device below.
    LioMulti* m = h->multi;
    for (lio_s2m_handle* ch : m->dev) {
        ch->defer_sync = true;
        ch->xyz_off = h->xyz_off;
        const int rc = lio_s2m_batch_upload(ch, n_scans, scans, n_pts, stride);
        ch->defer_sync = false;
        if (rc != LIO_OK) return rc;
    }
    h->xyz_off = 0;
    for (lio_s2m_handle* ch : m->dev) { const int rc = lio_s2m_batch_sync(ch); if (rc != LIO_OK) return rc; }
    m->n_scans = n_scans;
    h->n_scans = n_scans;
    return lio_multi_reserve(h, (size_t)n_scans);
}

int lio_multi_set_poses(lio_s2m_handle* h, const float* poses)
{
    for (lio_s2m_handle* ch : h->multi->dev) { const int rc = lio_s2m_batch_set_poses(ch, poses); if (rc != LIO_OK) return rc; }
    h->poses_set = true;
    return LIO_OK;
}

int lio_multi_set_degeneracy(lio_s2m_handle* h, int32_t scan, const float matP[36], int32_t is_degenerate)
{
    for (lio_s2m_handle* ch : h->multi->dev) { const int rc = lio_s2m_set_degeneracy(ch, scan, matP, is_degenerate); if (rc != LIO_OK) return rc; }
    return LIO_OK;
}

// The round-2 exchange (LIO_MULTI_EXCHANGE=host): D2H of every device's sums, a stream synchronisation per device, a host
// loop adding them in device order, H2D.  Kept for A/B timing of the device-side form only.
static int lio_multi_exchange_host(LioMulti* m, size_t n_val)
{
    const size_t bytes = n_val * sizeof(double);
    for (size_t c = 0; c < m->dev.size(); ++c)
        HIPCHK(hipMemcpyAsync(m->h_part[c], m->d_part[c], bytes, hipMemcpyDeviceToHost, m->dev[c]->stream));
    for (lio_s2m_handle* ch : m->dev) {
        HIPCHK(hipSetDevice(ch->cfg.device_id));
        HIPCHK(hipStreamSynchronize(ch->stream));
        m->stream_syncs++;
    }
    for (size_t k = 0; k < n_val; ++k) {
        double v = m->h_part[0][k];
        for (size_t c = 1; c < m->dev.size(); ++c) v += m->h_part[c][k];
        m->h_tot[k] = v;
    }
    for (size_t c = 0; c < m->dev.size(); ++c) {
        HIPCHK(hipSetDevice(m->dev[c]->cfg.device_id));
        HIPCHK(hipMemcpyAsync(m->d_tot[c], m->h_tot, bytes, hipMemcpyHostToDevice, m->dev[c]->stream));
    }
    return LIO_OK;
}

// Multi-device Gauss-Newton loop MO:1848-1859 with the join of MO:1622-1686 (see the head of this file).
int lio_multi_run(lio_s2m_handle* h)
{
    LioMulti* m = h->multi;
    if (!h->has_map) return lio_fail(LIO_ERR_NO_MAP, "set_map has not been called");
    if (m->n_scans < 1 || !h->poses_set) return lio_fail(LIO_ERR_ARG, "batch_upload and batch_set_poses first");
    const size_t nd = m->dev.size();
    const int n_val = m->n_scans * LIO_SUMS;
    const size_t slot = m->cap_scans * LIO_SUMS;                            // doubles per slot
    m->stream_syncs = m->event_waits = m->iterations = 0;
    for (lio_s2m_handle* ch : m->dev) { const int rc = lio_s2m_batch_begin(ch); if (rc != LIO_OK) return rc; }
    int look = h->cfg.lookahead;
    if (look < 0) look = 2;
    if (m->exchange == 2) look = 0;
    for (int it = 0; it < h->cfg.max_iters && it < LIO_MAX_ITERS; ++it) {   // MO:1848
        // convergence (MO:1857-1858 for every scan): the count of still-iterating scans after iteration it-1-look;
        // every device solves the same sums, device 0's count speaks for all
        const int chk = it - 1 - look;
        if (chk >= 0) {
            int32_t active = 0;
            const int rc = lio_s2m_batch_poll_active(m->dev[0], chk, &active);   // (an EVENT wait, not a stream synchronisation)
            m->event_waits++;
            if (rc != LIO_OK) return rc;
            if (active == 0) break;
        }
        const size_t par = (size_t)(it & 1) * nd * slot;
        // association + publish on every device
        for (size_t c = 0; c < nd; ++c) {
            lio_s2m_handle* ch = m->dev[c];
            const int rc = lio_s2m_batch_iter_partial(ch, m->d_part[c]);    // (sets the device)
            if (rc != LIO_OK) return rc;
            if (m->exchange == 0) {
                LioPeerPtrs dst;
                for (size_t d = 0; d < 8; ++d) dst.p[d] = d < nd ? m->d_gather[d] + par + c * slot : nullptr;
                hipLaunchKernelGGL(k_multi_publish, dim3((n_val + 255) / 256), dim3(256), 0, ch->stream, m->d_part[c], n_val, dst, (int)nd);
            } else if (m->exchange == 1) {
                for (size_t d = 0; d < nd; ++d)
                    HIPCHK(hipMemcpyPeerAsync(m->d_gather[d] + par + c * slot, m->dev[d]->cfg.device_id, m->d_part[c], ch->cfg.device_id,
                                              (size_t)n_val * sizeof(double), ch->stream));
            }
            if (m->exchange != 2) HIPCHK(hipEventRecord(m->ev_pub[c * LIO_MAX_ITERS + (size_t)it], ch->stream));
        }
        if (m->exchange == 2) { const int rc = lio_multi_exchange_host(m, (size_t)n_val); if (rc != LIO_OK) return rc; }
        // join + solve on every device
        const bool via_join = m->exchange != 2 && m->join_stream != nullptr;
        if (via_join) {
            HIPCHK(hipSetDevice(h->cfg.device_ids[0]));
            for (size_t c = 0; c < nd; ++c) HIPCHK(hipStreamWaitEvent(m->join_stream, m->ev_pub[c * LIO_MAX_ITERS + (size_t)it], 0));
            HIPCHK(hipEventRecord(m->ev_all[(size_t)it], m->join_stream));
        }
        for (size_t p = 0; p < nd; ++p) {
            lio_s2m_handle* ch = m->dev[p];
            HIPCHK(hipSetDevice(ch->cfg.device_id));
            if (via_join) {
                HIPCHK(hipStreamWaitEvent(ch->stream, m->ev_all[(size_t)it], 0));
            } else if (m->exchange != 2) {
                for (size_t c = 0; c < nd; ++c)
                    if (c != p) HIPCHK(hipStreamWaitEvent(ch->stream, m->ev_pub[c * LIO_MAX_ITERS + (size_t)it], 0));
            }
            // the slots are added in device order inside the solving kernel (bitwise reproducible, identical on every device)
            const int rc = m->exchange != 2 ? lio_s2m_iter_apply_slots(ch, m->d_gather[p] + par, slot, (int)nd)
                                            : lio_s2m_batch_iter_apply(ch, m->d_tot[p]);
            if (rc != LIO_OK) return rc;
        }
        m->iterations = it + 1;
    }
    h->ran = true;
    h->prof.multi_iterations = m->iterations;
    h->prof.multi_stream_syncs = m->stream_syncs;
    h->prof.multi_event_waits = m->event_waits;
    h->prof.multi_exchange = m->exchange;
    return LIO_OK;
}

int lio_multi_sync(lio_s2m_handle* h)
{
    for (lio_s2m_handle* ch : h->multi->dev) { const int rc = lio_s2m_batch_sync(ch); if (rc != LIO_OK) return rc; }
    return LIO_OK;
}

int lio_multi_results(lio_s2m_handle* h, float* poses, lio_s2m_result* results)
{
    if (!h->ran) return lio_fail(LIO_ERR_ARG, "nothing has been run");
    // every device holds the same per-scan state; the others are read as well so that their host mirrors
    // (persistent matP / isDegenerate, MO:176-177) stay current
    LioMulti* m = h->multi;
    int rc = LIO_OK;
    for (size_t c = m->dev.size(); c-- > 0 && rc == LIO_OK;)
        rc = lio_s2m_batch_results(m->dev[c], c == 0 ? poses : nullptr, c == 0 ? results : nullptr);
    const int mi = h->prof.multi_iterations, ms = h->prof.multi_stream_syncs, me = h->prof.multi_event_waits, mx = h->prof.multi_exchange;
    h->prof = m->dev[0]->prof;
    h->prof.multi_iterations = mi; h->prof.multi_stream_syncs = ms; h->prof.multi_event_waits = me; h->prof.multi_exchange = mx;
    return rc;
}

int lio_multi_get_correspondences(lio_s2m_handle* h, int32_t scan, uint8_t* flag, float* coeff4, int32_t* nn_idx5)
{
    // a point's record lives on the device that owned it in that iteration; neighbour indices come back in the
    // caller's map order (a shard numbers its points locally)
    LioMulti* m = h->multi;
    const size_t n = (size_t)m->dev[0]->h_state[scan].n_pts;
    std::vector<uint8_t> f(n);
    std::vector<float> cf(n * 4);
    std::vector<int32_t> nn(n * 5);
    if (flag) memset(flag, 0, n);
    if (coeff4) memset(coeff4, 0, n * 4 * sizeof(float));
    if (nn_idx5) for (size_t i = 0; i < n * 5; ++i) nn_idx5[i] = -1;
    for (size_t c = 0; c < m->dev.size(); ++c) {
        const int rc = lio_s2m_get_correspondences(m->dev[c], scan, f.data(), cf.data(), nn.data());
        if (rc != LIO_OK) return rc;
        const std::vector<int>& idx = m->shard_idx[c];
        for (size_t i = 0; i < n; ++i) {
            if (nn[i * 5] < 0) continue;                                // not processed (not owned / gate failed) on this device
            if (flag) flag[i] = f[i];
            if (coeff4) memcpy(coeff4 + i * 4, cf.data() + i * 4, 4 * sizeof(float));
            if (nn_idx5) for (int j = 0; j < 5; ++j) nn_idx5[i * 5 + j] = idx[(size_t)nn[i * 5 + j]];
        }
    }
    return LIO_OK;
}
