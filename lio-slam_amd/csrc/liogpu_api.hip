// liogpu_api.hip -- implementation of the C ABI in include/liogpu.h.
// Host-side orchestration only: device buffers, H2D/D2H, kernel launches on
// the handle's HIP stream.  MO = /root/reference/src/liorf/src/mapOptmization.cpp.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <string>
#include <vector>

#include "lio_handle.h"
#include "lio_multi.h"
#include "lio_pool.h"
#include <mutex>
#include <algorithm>


static thread_local std::string g_last_error;

int lio_fail(int code, const char* what, hipError_t e)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof(buf), "%s", what);
    g_last_error = buf;
    return code;
}

extern "C" int lio_version(void) { return LIO_VERSION; }
extern "C" const char* lio_last_error(void) { return g_last_error.c_str(); }

extern "C" void lio_s2m_default_config(lio_s2m_config* c)
{
    memset(c, 0, sizeof(*c));
    c->k = 5;                 // MO:1631
    c->max_sq_dist = 1.0f;    // MO:1641
    c->plane_tol = 0.2;       // MO:1662
    c->weight = 0.9;          // MO:1671
    c->min_s = 0.1;           // MO:1679
    c->min_corr = 50;         // MO:1722
    c->max_iters = 30;        // MO:1848
    c->eig_thresh = 100.0f;   // MO:1796
    c->conv_deg = 0.05;       // MO:1833
    c->conv_cm = 0.05;        // MO:1833
    c->min_scan_pts = 30;     // MO:1844
    c->jacobian_mode = 0;
    c->force_all_iters = 0;
    c->device_id = 0;
    c->cell_size = 0.0f;
    c->max_batch = 1;
    c->max_scan_pts = 0;
    c->record_corr_iter = -1;
    c->kernel_variant = 0;
    c->profile = 0;
    c->lookahead = -1;
    c->use_lds = 0;
    c->cell_div = 2;
    c->xcd_remap = 1;
    c->tile_size = 0.0f;
    c->use_graph = 0;
    c->sort_batch = 1;
    c->graph_iters = 4;
    c->sort_scan = 1;
    c->nn_cache = 1;
    c->pipeline = 0;
    c->n_devices = 1;
    for (int i = 0; i < 8; ++i) c->device_ids[i] = i;
    c->x_sub = 0;
    c->tight_rows = 0;
}

static void lio_fill_consts(lio_s2m_handle* h)
{
    const lio_s2m_config& g = h->cfg;
    h->c.plane_tol = g.plane_tol; h->c.weight = g.weight; h->c.min_s = g.min_s;
    h->c.conv_deg = g.conv_deg; h->c.conv_cm = g.conv_cm;
    h->c.max_sq_dist = g.max_sq_dist; h->c.eig_thresh = g.eig_thresh;
    h->c.min_corr = g.min_corr; h->c.max_iters = g.max_iters;
    h->c.jac_exact = g.jacobian_mode ? 1 : 0; h->c.force_all = g.force_all_iters ? 1 : 0;
    h->c.record_iter = g.record_corr_iter; h->c.min_scan_pts = g.min_scan_pts;
}

static int lio_s2m_init_resources(lio_s2m_handle* h)
{
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (int i = 0; i < LIO_MAX_ITERS; ++i) {
        HIPCHK(hipEventCreate(&h->ev_beg[i]));
        HIPCHK(hipEventCreate(&h->ev_end[i]));
        HIPCHK(hipEventCreateWithFlags(&h->ev_chk[i], hipEventDisableTiming));
    }
    HIPCHK(hipHostMalloc((void**)&h->h_active, sizeof(int) * LIO_MAX_ITERS, hipHostMallocDefault));
    HIPCHK(hipEventCreate(&h->ev_map[0]));
    HIPCHK(hipEventCreate(&h->ev_map[1]));
    HIPCHK(hipEventCreate(&h->ev_mapl[0]));
    HIPCHK(hipEventCreate(&h->ev_mapl[1]));
    h->ev_ok = true;
    HIPCHK(hipMalloc((void**)&h->d_bbox, 6 * sizeof(unsigned)));
    HIPCHK(hipMalloc((void**)&h->d_active, sizeof(int)));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->cfg.device_id) == hipSuccess) h->n_cu = prop.multiProcessorCount;
    (void)hipGetLastError();
    return LIO_OK;
}

extern "C" int lio_s2m_create(const lio_s2m_config* cfg, lio_s2m_handle** out)
{
    if (!cfg || !out) return lio_fail(LIO_ERR_ARG, "null argument");
    if (cfg->k != 5) return lio_fail(LIO_ERR_ARG, "only k = 5 is supported (MO:1631)");
    if (cfg->max_iters < 1 || cfg->max_iters > LIO_MAX_ITERS) return lio_fail(LIO_ERR_ARG, "max_iters out of range");
    if (!(cfg->max_sq_dist > 0.0f)) return lio_fail(LIO_ERR_ARG, "max_sq_dist must be positive");
    if (cfg->x_sub != 0 && cfg->x_sub != 1 && cfg->x_sub != 2 && cfg->x_sub != 4 && cfg->x_sub != 8)
        return lio_fail(LIO_ERR_ARG, "cfg.x_sub must be 0 (auto), 1, 2, 4 or 8");
    if (cfg->tight_rows < -1 || cfg->tight_rows > LIO_TB_MAX) return lio_fail(LIO_ERR_ARG, "cfg.tight_rows must be -1 (none), 0 (auto) or 1..3 tables");
    if (cfg->pipeline != 0 && cfg->pipeline != 1 && cfg->pipeline != 4)
        return lio_fail(LIO_ERR_ARG, "cfg.pipeline must be 0 (auto), 1 (one launch per iteration) or 4 (one-launch loop)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lio_fail(LIO_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return lio_fail(LIO_ERR_ARG, "device_id out of range");
    if (cfg->n_devices > 1) {                           // in-library multi-GPU: a front handle + one child per listed device (lio_multi.hip)
        const int rc = lio_multi_create(cfg, out);
        if (rc == LIO_OK) lio_fill_consts(*out);
        return rc;
    }
    HIPCHK(hipSetDevice(cfg->device_id));
    lio_s2m_handle* h = new lio_s2m_handle();
    h->cfg = *cfg;
    if (h->cfg.cell_size > 0.0f && h->cfg.cell_size < sqrtf(h->cfg.max_sq_dist) * 1.001f) {   // (before division by cell_div)
        delete h;
        return lio_fail(LIO_ERR_ARG, "cell_size must be >= sqrt(max_sq_dist)*1.001 for an exact 27-cell search");
    }
    lio_fill_consts(h);
    h->shard.axis = -1;
    const int rc = lio_s2m_init_resources(h);
    if (rc != LIO_OK) { lio_s2m_destroy(h); return rc; }    // (releases whatever had been created)
    *out = h;
    return LIO_OK;
}

extern "C" void lio_s2m_destroy(lio_s2m_handle* h)
{
    if (!h) return;
    if (h->multi) { lio_multi_destroy(h); return; }
    (void)hipSetDevice(h->cfg.device_id);
    (void)hipStreamSynchronize(h->stream);
    if (h->corner) { lio_s2m_destroy(h->corner); h->corner = nullptr; }
    if (h->raw_ws) { lio_raw_ws_free(h->raw_ws); h->raw_ws = nullptr; }
    void* ptrs[] = { h->d_mx, h->d_my, h->d_mz, h->d_map4, h->d_sorted, h->d_cell_of, h->d_cell_count,
                     h->d_cell_start, h->d_tile_sums, h->d_bbox, h->d_stage, h->d_map_stage, h->d_sx, h->d_sy, h->d_sz,
                     h->d_state, h->d_poses, h->d_blocks, h->d_partials, h->d_arrive, h->d_rec_flag,
                     h->d_rec_coeff, h->d_rec_nn, h->d_active, h->d_tiles, h->d_prep_blocks, h->d_key_of,
                     h->d_key_count, h->d_key_start, h->d_key_tiles, h->d_tmp_idx, h->d_perm, h->d_stamps, h->d_nbr_start, h->d_nbr_pts,
                     h->d_nn_cache, h->d_summary, h->d_big_list, h->d_scan_bbox, h->d_block_box, h->d_blk_skip, h->d_gen, h->d_spec_sums, h->d_nbr_slot };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < LIO_MAX_ITERS; ++i) {               // (a handle whose creation failed half-way holds nulls)
        if (h->ev_beg[i]) (void)hipEventDestroy(h->ev_beg[i]);
        if (h->ev_end[i]) (void)hipEventDestroy(h->ev_end[i]);
        if (h->ev_chk[i]) (void)hipEventDestroy(h->ev_chk[i]);
    }
    if (h->h_active) (void)hipHostFree(h->h_active);
    if (h->h_summary) (void)hipHostFree(h->h_summary);
    if (h->h_scan_bbox) (void)hipHostFree(h->h_scan_bbox);
    if (h->ev_map[0]) (void)hipEventDestroy(h->ev_map[0]);
    if (h->ev_map[1]) (void)hipEventDestroy(h->ev_map[1]);
    if (h->ev_mapl[0]) (void)hipEventDestroy(h->ev_mapl[0]);
    if (h->ev_mapl[1]) (void)hipEventDestroy(h->ev_mapl[1]);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int lio_s2m_set_stream(lio_s2m_handle* h, void* hip_stream)
{
    if (h && h->multi) return lio_fail(LIO_ERR_ARG, "not available on a multi-device handle (cfg.n_devices > 1 shards inside the library)");
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) HIPCHK(hipStreamDestroy(h->stream));
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    h->graph_dirty = true;
    if (h->corner) return lio_s2m_set_stream(h->corner, hip_stream);
    return LIO_OK;
}

static float lio_ord2f(unsigned u)
{
    unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

static const lio_s2m_handle* lio_map_of(const lio_s2m_handle* h) { return h->map_src ? h->map_src : h; }

// ------------------------------------------------------------------ set_map
static int lio_map_reserve(lio_s2m_handle* h, size_t n)
{
    const size_t nn = n ? n : 1;
    HIPCHK(lio_grow(&h->d_mx, &h->cap_mxyz[0], nn));
    HIPCHK(lio_grow(&h->d_my, &h->cap_mxyz[1], nn));
    HIPCHK(lio_grow(&h->d_mz, &h->cap_mxyz[2], nn));
    HIPCHK(lio_grow(&h->d_map4, &h->cap_map4, nn));
    HIPCHK(lio_grow(&h->d_sorted, &h->cap_sorted, nn));
    HIPCHK(lio_grow(&h->d_cell_of, &h->cap_cell_of, nn));
    return LIO_OK;
}

// common tail: d_mx/d_my/d_mz/d_map4 hold the n map points -> bounding box, grid, neighbourhood rows
// `box` (optional): min[3], max[3] of a box known to contain every map point -- the grid is laid over it without a bounding-box
// pass and its read-back (any enclosing box gives an exact search; the grid is only a few cells larger) -- and the call
// returns with the build still in flight on the handle's stream: the registrations that follow are ordered behind it,
// sharers wait for ev_mapl[1], the build time is resolved when the profile is read.
static int lio_map_finish(lio_s2m_handle* h, size_t n, std::chrono::steady_clock::time_point t0, const float* box = nullptr)
{
    const size_t nn = n ? n : 1;
    float mn[3], mx[3];
    // (a box that is not finite -- a keyframe cloud of nothing but non-finite points -- is ignored: the box is then computed here)
    for (int a = 0; a < 3 && box; ++a)
        if (!(box[a] <= box[3 + a]) || !(fabsf(box[a]) <= 1.0e15f) || !(fabsf(box[3 + a]) <= 1.0e15f)) box = nullptr;   // (LIO_MAX_COORD)
    if (box) {
        for (int a = 0; a < 3; ++a) { mn[a] = n ? box[a] : 0.0f; mx[a] = n ? box[3 + a] : 0.0f; }
    } else {
        unsigned init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
        HIPCHK(hipMemcpyAsync(h->d_bbox, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
        if (n) lio_launch_map_bbox(h->d_mx, h->d_my, h->d_mz, (int)n, h->d_bbox, h->stream);
        unsigned hb[6];
        HIPCHK(hipMemcpyAsync(hb, h->d_bbox, sizeof(hb), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        bool empty = (n == 0) || hb[0] == 0xffffffffu;
        // (a map without a single finite point leaves the reduction at its identities, max < min: an empty grid, every point outside it)
        for (int a = 0; a < 3 && !empty; ++a) empty = !(lio_ord2f(hb[a]) <= lio_ord2f(hb[3 + a]));
        for (int a = 0; a < 3; ++a) { mn[a] = empty ? 0.0f : lio_ord2f(hb[a]); mx[a] = empty ? 0.0f : lio_ord2f(hb[3 + a]); }
    }

    auto t1 = std::chrono::steady_clock::now();

    // cell edge = gate radius (+0.1 %) / k; the candidate scan visits (2k+1)^3 cells
    int kdiv = h->cfg.cell_div;
    if (kdiv != 1 && kdiv != 2 && kdiv != 3) kdiv = 2;
    float cell = (h->cfg.cell_size > 0.0f ? h->cfg.cell_size : sqrtf(h->cfg.max_sq_dist) * 1.001f) / (float)kdiv;
    // x subdivision of the row buckets (LioGrid::xs, cfg.x_sub): auto = 4 for a batch handle, 1 for a node's handle (measured,
    // DESIGN.md section 6); LIO_X_SUB = 1 | 2 | 4 | 8 in the environment overrides it for A/B runs
    int xsub = h->cfg.x_sub > 0 ? h->cfg.x_sub : (h->cfg.max_batch >= 8 ? 4 : 1);
    { const char* e = getenv("LIO_X_SUB"); const int v = e ? atoi(e) : 0; if (v == 1 || v == 2 || v == 4 || v == 8) xsub = v; }
    LioGrid g;
    for (;;) {
        g.k = kdiv;
        g.inv_cell = 1.0f / cell;
        g.ox = mn[0] - 0.5f * cell; g.oy = mn[1] - 0.5f * cell; g.oz = mn[2] - 0.5f * cell;
        const double ex = ((double)mx[0] - g.ox) * g.inv_cell, ey = ((double)mx[1] - g.oy) * g.inv_cell,
                     ez = ((double)mx[2] - g.oz) * g.inv_cell;
        const double cells = (floor(ex) + 2.0) * (floor(ey) + 2.0) * (floor(ez) + 2.0);
        if (cells * xsub <= 256.0 * 1024.0 * 1024.0) {
            g.nx = (int)floor(ex) + 2; g.ny = (int)floor(ey) + 2; g.nz = (int)floor(ez) + 2;
            g.n_cells = g.nx * g.ny * g.nz;
            break;
        }
        cell *= 1.5f;   // larger cells keep the search exact, only less selective
    }
    g.xs = xsub; g.nxf = g.nx * xsub; g.inv_cell_x = g.inv_cell * (float)xsub;
    // tight rows (LioGrid::tb_*, cfg.tight_rows): further row tables with k = 1 and cells of 0.6 / 0.3 / 0.15 x the gate radius
    // for the queries whose search bound is that small -- most of them from the second iteration on.  LIO_TIGHT = a comma
    // separated, descending list of fractions (0 = none) in the environment for A/B runs.
    float frac[LIO_TB_MAX] = { 0.6f, 0.3f, 0.15f };
    int n_tb = h->cfg.tight_rows > 0 ? std::min(h->cfg.tight_rows, LIO_TB_MAX) : 0;
    float pts_per_cell = 0.0f;
    int try_lvl = -1;
    if (h->cfg.tight_rows == 0 && h->cfg.max_batch >= 8) {
        // auto, a handle set up for batches: one table always; the finer ones where the map is dense enough to have queries for
        // them.  Points per occupied cell -> point spacing on the surfaces, s = cell / sqrt(points per cell); the fifth neighbour
        // of a query on such a surface is about 1.26 s away; a table is built when its reach is at least that.  (Measured on
        // the parameter sets of tools/param_sets.sh: the sparse 0.5 m maps are fastest with one table -- a second one only adds
        // build time --, the 0.1 m map with three: +39 % registrations/s over one.)  Costs one small kernel and a 4-byte
        // read-back here; a map whose bounding box is still on the device (`box`) stays asynchronous and gets one table.
        n_tb = 1;
        if (!box && n > 0) {
            HIPCHK(lio_grow(&h->d_cell_count, &h->cap_cell_count, (size_t)g.n_cells + 1));
            lio_launch_map_occupancy(g, h->d_mx, h->d_my, h->d_mz, (int)n, h->d_cell_count, h->stream);
            int occ = 0;
            HIPCHK(hipMemcpyAsync(&occ, h->d_cell_count + g.n_cells, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (occ > 0) {
                pts_per_cell = (float)n / (float)occ;
                const float spacing = cell / sqrtf(pts_per_cell);
                while (n_tb < LIO_TB_MAX && sqrtf(h->cfg.max_sq_dist) * frac[n_tb] >= 1.26f * spacing) ++n_tb;
                // the table an unbounded query tries first (LioGrid::tb_try): the tightest one 1.75x as wide as the expected
                // distance of the fifth neighbour.  (Measured: +27 % registrations/s on the 0.1 m map, +5 % on the 0.3 m map,
                // -4 % on the 0.5 m maps if forced there: nearly every wave then pays both searches -- the rule leaves those out.)
                for (int l = 0; l < n_tb; ++l)
                    if (sqrtf(h->cfg.max_sq_dist) * frac[l] >= 2.2f * spacing) try_lvl = l;
            }
        }
    }
    if (const char* e = getenv("LIO_TIGHT")) {
        n_tb = 0;
        for (const char* p = e; *p && n_tb < LIO_TB_MAX;) {
            char* q = nullptr;
            const float v = strtof(p, &q);
            if (q == p) break;
            if (v >= 0.05f && v <= 1.0f && (n_tb == 0 || v < frac[n_tb - 1])) frac[n_tb++] = v;
            p = (*q == ',') ? q + 1 : q;
            if (*q != ',') break;
        }
    }
    if ((double)g.n_cells * xsub > 64.0 * 1024.0 * 1024.0) n_tb = 0;        // (a grid that large is mostly empty buckets already)
    size_t len_b = 0, rows_b = 0;
    for (int l = 0; l < LIO_TB_MAX; ++l) {
        g.tb_row0[l] = 0; g.tb_ny[l] = g.tb_nz[l] = 0; g.tb_oy[l] = g.tb_oz[l] = 0.0f; g.tb_inv_cell[l] = 0.0f; g.tb_reach[l] = -1.0f;
        if (l >= n_tb) continue;
        const float reach = sqrtf(h->cfg.max_sq_dist) * frac[l], cb = reach * 1.001f;
        const float inv = 1.0f / cb, oy = mn[1] - 0.5f * cb, oz = mn[2] - 0.5f * cb;
        const double ny = floor(((double)mx[1] - oy) * inv) + 2.0, nz = floor(((double)mx[2] - oz) * inv) + 2.0;
        // (bucket indices stay far inside 31 bits, and so do record offsets: every table adds 9 records per point + row padding)
        const double recs = (double)nn * ((2 * g.k + 1) * (2 * g.k + 1) + 9.0 * (l + 1)) + LIO_ROW_ALIGN * ((double)g.ny * g.nz + (double)rows_b + ny * nz);
        if ((double)g.n_cells * xsub + (double)len_b + ny * nz * g.nxf > 192.0 * 1024.0 * 1024.0 || recs > 2040.0 * 1024.0 * 1024.0) { n_tb = l; continue; }
        g.tb_inv_cell[l] = inv; g.tb_oy[l] = oy; g.tb_oz[l] = oz; g.tb_ny[l] = (int)ny; g.tb_nz[l] = (int)nz;
        g.tb_row0[l] = (int)((size_t)g.n_cells * xsub + len_b);
        g.tb_reach[l] = reach;
        rows_b += (size_t)g.tb_ny[l] * g.tb_nz[l];
        len_b += (size_t)g.tb_ny[l] * g.tb_nz[l] * g.nxf;
    }
    if (const char* e = getenv("LIO_TRY")) try_lvl = atoi(e);            // (A/B runs)
    g.tb_try = (try_lvl >= 0 && try_lvl < n_tb) ? try_lvl : (try_lvl >= n_tb ? n_tb - 1 : -1);
    for (int l = 0; l < LIO_TB_MAX; ++l) {
        // (reach / 1.0002)^2, rounded towards zero by one more part in 10^6
        const float r = g.tb_reach[l] > 0.0f ? g.tb_reach[l] / 1.0002f : 0.0f;
        g.tb_try_b2[l] = r * r * 0.999999f;
    }
    h->grid = g;
    const size_t len_a = (size_t)g.n_cells * xsub, reps = (size_t)((2 * g.k + 1) * (2 * g.k + 1)) + 9 * (size_t)n_tb;
    HIPCHK(lio_grow(&h->d_cell_count, &h->cap_cell_count, (size_t)g.n_cells + len_a + len_b));   // (point counts + row bucket lengths of both tables)
    HIPCHK(lio_grow(&h->d_cell_start, &h->cap_cell_start, (size_t)g.n_cells + 1));
    HIPCHK(lio_grow(&h->d_nbr_start, &h->cap_nbr_start, len_a + len_b + 1));
    HIPCHK(lio_grow(&h->d_nbr_pts, &h->cap_nbr_pts, nn * reps + LIO_ROW_ALIGN * ((size_t)g.ny * g.nz + rows_b) + 4 * LIO_ROW_ALIGN, 1.05));
    HIPCHK(lio_grow(&h->d_tile_sums, &h->cap_tile_sums, 2 * ((size_t)lio_scan_tiles((int)(len_a + len_b)) + 1)));   // (64-bit pair sums)
    HIPCHK(lio_grow(&h->d_nbr_slot, &h->cap_nbr_slot, nn * reps, 1.05));

    HIPCHK(hipEventRecord(box ? h->ev_mapl[0] : h->ev_map[0], h->stream));
    if (n) {
        lio_launch_map_build(g, h->d_mx, h->d_my, h->d_mz, (int)n, h->d_cell_of, h->d_cell_count,
                             h->d_cell_start, h->d_tile_sums, h->d_sorted, h->d_nbr_start, h->d_nbr_pts, h->d_nbr_slot, h->cfg.use_lds != 0, h->stream);
    } else {
        HIPCHK(hipMemsetAsync(h->d_cell_start, 0, sizeof(int) * ((size_t)g.n_cells + 1), h->stream));
        HIPCHK(hipMemsetAsync(h->d_nbr_start, 0, sizeof(int) * (len_a + len_b + 1), h->stream));
    }
    HIPCHK(hipEventRecord(box ? h->ev_mapl[1] : h->ev_map[1], h->stream));
    HIPCHK(hipGetLastError());
    if (box) {
        h->map_timing_pending = true;
    } else {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipGetLastError());
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev_map[0], h->ev_map[1]));
        h->prof.map_build_ms = ms;
        h->map_timing_pending = false;
        HIPCHK(hipEventRecord(h->ev_mapl[1], h->stream));            // "map ready" for sharers (already true)
    }
    h->prof.map_upload_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    h->prof.n_map = (int64_t)n;
    h->prof.n_cells = g.n_cells;
    h->prof.map_x_sub = g.xs;
    h->prof.map_tight_tables = n_tb;
    h->prof.map_first_try = g.tb_try;
    h->prof.map_pts_per_cell = pts_per_cell;
    h->n_map = n;
    h->has_map = true;
    h->graph_dirty = true;
    h->map_epoch++;
    return LIO_OK;
}

extern "C" int lio_s2m_set_map(lio_s2m_handle* h, const void* pts, size_t n, size_t stride)
{
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    if (n > 0 && !pts) return lio_fail(LIO_ERR_ARG, "null map pointer");
    if (stride < 12 || (stride & 3)) return lio_fail(LIO_ERR_ARG, "stride_bytes must be >= 12 and a multiple of 4");
    if (h->multi) return lio_multi_set_map(h, pts, n, stride);
    if (n >= (1ull << 25)) return lio_fail(LIO_ERR_CAPACITY, "map too large ((2k+1)^2 x n records must fit a 31-bit offset)");
    if (h->map_src) return lio_fail(LIO_ERR_ARG, "this handle searches another handle's map (lio_s2m_share_map)");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();   // drop stale codes left by other HIP users of this thread (e.g. hipErrorNotReady)
    auto t0 = std::chrono::steady_clock::now();
    h->has_map = false;
    int rc = lio_map_reserve(h, n);
    if (rc != LIO_OK) return rc;
    HIPCHK(lio_grow(&h->d_map_stage, &h->cap_map_stage, (n ? n : 1) * stride));
    if (n) {
        HIPCHK(hipMemcpyAsync(h->d_map_stage, pts, n * stride, hipMemcpyHostToDevice, h->stream));
        lio_launch_aos_to_soa(h->d_map_stage, stride, (int)n, h->d_mx, h->d_my, h->d_mz, h->d_map4, h->stream);
    }
    return lio_map_finish(h, n, t0);
}

// Device-resident form used by lio_assemble_map: d_xyzi = float4 (x,y,z,intensity)[n] on h's device.
int lio_s2m_set_map_device_xyzi(lio_s2m_handle* h, const float4* d_xyzi, size_t n)
{
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    if (n >= (1ull << 25)) return lio_fail(LIO_ERR_CAPACITY, "map too large ((2k+1)^2 x n records must fit a 31-bit offset)");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();
    auto t0 = std::chrono::steady_clock::now();
    h->has_map = false;
    int rc = lio_map_reserve(h, n);
    if (rc != LIO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());        // the producer ran on another stream
    lio_launch_xyzi4_to_soa(d_xyzi, (int)n, h->d_mx, h->d_my, h->d_mz, h->d_map4, h->stream);
    return lio_map_finish(h, n, t0);
}

// The same for a producer that ran on THIS handle's stream (lio_assemble_map_resident) and knows a box around the
// points: no device-wide wait, no bounding-box pass, no wait at the end (see lio_map_finish).
int lio_s2m_set_map_device_bbox(lio_s2m_handle* h, const float4* d_xyzi, size_t n, const float box[6])
{
    if (!h || !box) return lio_fail(LIO_ERR_ARG, "null argument");
    if (h->multi || h->map_src) return lio_fail(LIO_ERR_ARG, "this handle cannot take a device-resident map");
    if (n >= (1ull << 25)) return lio_fail(LIO_ERR_CAPACITY, "map too large ((2k+1)^2 x n records must fit a 31-bit offset)");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();
    auto t0 = std::chrono::steady_clock::now();
    h->has_map = false;
    h->run_pending = false;
    int rc = lio_map_reserve(h, n);
    if (rc != LIO_OK) return rc;
    lio_launch_xyzi4_to_soa(d_xyzi, (int)n, h->d_mx, h->d_my, h->d_mz, h->d_map4, h->stream);
    return lio_map_finish(h, n, t0, box);
}

hipStream_t lio_s2m_stream_of(lio_s2m_handle* h) { return h ? h->stream : nullptr; }
bool lio_s2m_takes_device_map(const lio_s2m_handle* h) { return h && !h->multi && !h->map_src; }

extern "C" int lio_s2m_set_global_grid(lio_s2m_handle* h, const float origin[3], const int32_t dims[3])
{
    if (!h || !origin || !dims) return lio_fail(LIO_ERR_ARG, "null argument");
    for (int a = 0; a < 3; ++a) { h->gorigin[a] = origin[a]; h->gdims[a] = dims[a]; }
    h->has_global = true;
    return LIO_OK;
}

extern "C" int lio_s2m_set_scan_shard(lio_s2m_handle* h, int32_t rank, int32_t world)
{
    if (h && h->multi) return lio_fail(LIO_ERR_ARG, "not available on a multi-device handle (cfg.n_devices > 1 shards inside the library)");
    if (!h || world < 1 || rank < 0 || rank >= world) return lio_fail(LIO_ERR_ARG, "need 0 <= rank < world");
    h->block_rank = rank;
    h->block_world = world;
    return LIO_OK;
}

// Streaming (the metric includes the per-scan H2D): a second handle searches THIS handle's resident
// map, with its own stream and its own scan buffers, so that batch k+1 can be uploaded and tile-sorted while
// batch k iterates.  The map owner must outlive the sharer; replacing the map needs both streams idle.
extern "C" int lio_s2m_share_map(lio_s2m_handle* h, lio_s2m_handle* map_owner)
{
    if (!h || h == map_owner) return lio_fail(LIO_ERR_ARG, "need two different handles");
    if (h->multi || (map_owner && map_owner->multi)) return lio_fail(LIO_ERR_ARG, "not available on a multi-device handle");
    if (map_owner && (map_owner->map_src || map_owner->cfg.device_id != h->cfg.device_id))
        return lio_fail(LIO_ERR_ARG, "the map owner must hold its own map on the same device");
    if (map_owner && h->cfg.use_lds && !map_owner->cfg.use_lds)
        return lio_fail(LIO_ERR_ARG, "cfg.use_lds needs the cell-sorted copy of the map, which only an owner created with use_lds builds");
    h->map_src = map_owner;
    h->map_epoch = map_owner ? map_owner->map_epoch : 0;
    h->graph_dirty = true;
    return LIO_OK;
}

extern "C" void* lio_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void lio_host_free(void* p) { if (p) (void)hipHostFree(p); }

// Device memory for callers that keep their clouds in HBM between calls (every entry point that takes `scans` / `data`
// reads memory of the handle's own device in place).
extern "C" void* lio_device_alloc(int32_t device_id, size_t bytes)
{
    void* p = nullptr;
    if (hipSetDevice(device_id) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void lio_device_free(int32_t device_id, void* p)
{
    if (p && hipSetDevice(device_id) == hipSuccess) (void)hipFree(p);
}
extern "C" int lio_device_upload(int32_t device_id, void* dst, const void* src, size_t bytes)
{
    if (!dst || (!src && bytes)) return lio_fail(LIO_ERR_ARG, "null pointer");
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return LIO_OK;
}
extern "C" int lio_host_register(void* p, size_t bytes)
{
    if (!p || !bytes) return lio_fail(LIO_ERR_ARG, "null range");
    HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return LIO_OK;
}
extern "C" int lio_host_unregister(void* p)
{
    if (!p) return lio_fail(LIO_ERR_ARG, "null pointer");
    HIPCHK(hipHostUnregister(p));
    return LIO_OK;
}

extern "C" int lio_s2m_batch_upload_async(lio_s2m_handle* h, int32_t n_scans, const void* const* scans,
                                          const size_t* n_pts, size_t stride)
{
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    const bool keep = h->defer_sync;
    h->defer_sync = true;
    h->async_upload = true;
    const int rc = lio_s2m_batch_upload(h, n_scans, scans, n_pts, stride);
    h->defer_sync = keep;
    h->async_upload = false;
    return rc;
}

extern "C" int lio_s2m_set_shard(lio_s2m_handle* h, int32_t axis, int32_t lo, int32_t hi)
{
    if (h && h->multi) return lio_fail(LIO_ERR_ARG, "not available on a multi-device handle (cfg.n_devices > 1 shards inside the library)");
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    if (axis < 0) { h->shard.axis = -1; return LIO_OK; }
    if (axis > 2 || !h->has_global) return lio_fail(LIO_ERR_ARG, "set_global_grid first; axis in 0..2");
    const float cell = h->cfg.cell_size > 0.0f ? h->cfg.cell_size : sqrtf(h->cfg.max_sq_dist) * 1.001f;
    h->shard.axis = axis;
    h->shard.gorigin = h->gorigin[axis];
    h->shard.inv_cell = 1.0f / cell;
    h->shard.gdim = h->gdims[axis];
    h->shard.lo = lo;
    h->shard.hi = hi;
    h->plan_ranks = 0;                 // (per-point ownership only; lio_s2m_set_shard_plan adds whole-workgroup ownership)
    h->graph_dirty = true;
    return LIO_OK;
}

// The full slab plan: rank `rank` of `n_ranks` owns cells [bounds[rank], bounds[rank+1]) along `axis` and HOLDS the map
// points of the cells [bounds[rank] - halo_cells, bounds[rank+1] + halo_cells) (what the caller passed to lio_s2m_set_map).
// With the whole plan known, a workgroup of scan points that fits one rank's held region is processed wholly by that rank
// (k_shard_cull); halo_cells = 1 reduces to lio_s2m_set_shard.  Every rank must be given the same bounds and halo.
extern "C" int lio_s2m_set_shard_plan(lio_s2m_handle* h, int32_t axis, int32_t n_ranks, int32_t rank, const int32_t* bounds,
                                      int32_t halo_cells)
{
    if (!h || !bounds) return lio_fail(LIO_ERR_ARG, "null argument");
    if (h->multi) return lio_fail(LIO_ERR_ARG, "not available on a multi-device handle (cfg.n_devices > 1 shards inside the library)");
    if (n_ranks < 1 || n_ranks > 8 || rank < 0 || rank >= n_ranks || halo_cells < 1)
        return lio_fail(LIO_ERR_ARG, "need 1 <= n_ranks <= 8, 0 <= rank < n_ranks, halo_cells >= 1");
    for (int r = 0; r < n_ranks; ++r)
        if (bounds[r] > bounds[r + 1]) return lio_fail(LIO_ERR_ARG, "bounds must be non-decreasing");
    const int rc = lio_s2m_set_shard(h, axis, bounds[rank], bounds[rank + 1]);
    if (rc != LIO_OK) return rc;
    h->plan_ranks = halo_cells > 1 ? n_ranks : 0;
    h->plan_rank = rank;
    h->plan_halo = halo_cells;
    for (int r = 0; r <= n_ranks; ++r) h->plan_bounds[r] = bounds[r];
    return LIO_OK;
}

// ------------------------------------------------------------------- batch
// The whole Gauss-Newton loop as one launch (k_s2m_persist, lio_persist.hip): possible when every workgroup of the batch is
// resident at once and the batch uses nothing but the default surf association; measured faster than the launch loop
// for every batch that fits (0.17 against 0.25 ms for a lone registration).  cfg.pipeline = 4 takes
// it for up to one workgroup per compute unit; auto (0) for up to half of that -- a lone registration is ~26 workgroups
// at the default 0.4 m scan leaf, 76 / 111 at the reference's 0.2 / 0.15 m leaves (6-8 % faster there than the launch loop,
// profiles/r03_callback_leaf_sizes.txt) --, so that two handles doing so at the same time still fit the device together (a
// workgroup that waits at its scan's barrier keeps its slot; the polls are bounded, and a launch that cannot make progress
// is re-run through the launch loop inside lio_s2m_batch_results: a few milliseconds, never an error).
// Auto steps aside for an explicit cfg.use_graph and for the diagnostic phase clock of the launch loop (cfg.profile = 2).
static bool lio_persist_eligible(const lio_s2m_handle* h)
{
    int limit = 0;
    if (h->cfg.pipeline == 4) limit = h->n_cu;
    else if (h->cfg.pipeline == 0 && !h->cfg.use_graph && h->cfg.profile != 2) limit = h->n_cu / 2;
    return limit > 0 && !h->no_persist && !h->cfg.use_lds && h->ppt == 1 && h->shard.axis < 0 &&
           h->block_world == 1 && h->n_blocks > 0 && h->n_blocks + h->n_scans <= limit;      // (+ one helper workgroup per scan)
}

// Polls before a workgroup of the one-launch loop stops waiting for its scan's solve.  A poll is a short sleep plus one
// L2-bypassing load (~1-2 us): the default of 4096 bounds a stalled launch to a few milliseconds, after which the host re-runs
// the registration through the launch loop (lio_s2m_batch_results).  A healthy solve is answered within ~50 us.
static unsigned lio_persist_spin_max(const lio_s2m_handle* h)
{
    if (h->persist_spin_max) return h->persist_spin_max;
    static const unsigned env = [] { const char* e = getenv("LIO_PERSIST_SPIN_MAX"); const long v = e ? atol(e) : 0; return v > 0 ? (unsigned)v : 4096u; }();
    return env;
}

// Test hook: bound of the one-launch loop's barrier polls for this handle (0 = default) and the index of an association
// workgroup that never arrives (-1 = none), which forces the time-out and with it the fall-back to the launch loop.
extern "C" int lio_s2m_debug_persist_spin(lio_s2m_handle* h, int32_t spin_max, int32_t withhold_wg)
{
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    h->persist_spin_max = spin_max > 0 ? (unsigned)spin_max : 0u;
    h->persist_withhold = withhold_wg;
    return LIO_OK;
}

// The SoA copy of the resident batch, for the paths that read it, if the upload skipped it.
static int lio_ensure_soa(lio_s2m_handle* h)
{
    if (h->soa_valid || !h->total_pts) return LIO_OK;
    lio_launch_aos_to_soa(h->last_stage + h->last_xyz_off, h->last_stride, (int)h->total_pts, h->d_sx, h->d_sy, h->d_sz, nullptr, h->stream);
    h->soa_valid = true;
    HIPCHK(hipGetLastError());
    return LIO_OK;
}

extern "C" int lio_s2m_batch_upload(lio_s2m_handle* h, int32_t n_scans, const void* const* scans,
                                    const size_t* n_pts, size_t stride)
{
    if (!h || !scans || !n_pts) return lio_fail(LIO_ERR_ARG, "null argument");
    if (n_scans < 1) return lio_fail(LIO_ERR_ARG, "n_scans must be >= 1");
    if (stride < 12 || (stride & 3)) return lio_fail(LIO_ERR_ARG, "stride_bytes must be >= 12 and a multiple of 4");
    if (h->multi) return lio_multi_upload(h, n_scans, scans, n_pts, stride);
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();   // drop stale codes left by other HIP users of this thread (e.g. hipErrorNotReady)
    size_t total = 0, max_n = 0;
    for (int s = 0; s < n_scans; ++s) {
        if (n_pts[s] && !scans[s]) return lio_fail(LIO_ERR_ARG, "null scan pointer");
        total += n_pts[s];
        if (n_pts[s] > max_n) max_n = n_pts[s];
    }
    if (total > 0x7fffffffull - 1024) return lio_fail(LIO_ERR_CAPACITY, "batch too large");
    h->run_pending = false;            // (a run whose results were never fetched is abandoned)
    if (h->host_state_stale && h->n_scans > 0 && h->d_state) {
        // the persistent members (matP / isDegenerate, MO:176-177) live in the device copy: bring the host copy
        // up to date before it is edited and uploaded again
        HIPCHK(hipMemcpyAsync(h->h_state.data(), h->d_state, (size_t)h->n_scans * sizeof(LioScanState),
                              hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    h->host_state_stale = false;
    const size_t tt = total ? total : 1;
    HIPCHK(lio_grow(&h->d_sx, &h->cap_sxyz[0], tt));
    HIPCHK(lio_grow(&h->d_sy, &h->cap_sxyz[1], tt));
    HIPCHK(lio_grow(&h->d_sz, &h->cap_sxyz[2], tt));
    HIPCHK(lio_grow(&h->d_stage, &h->cap_stage, tt * stride));
    HIPCHK(lio_grow(&h->d_state, &h->cap_state, (size_t)n_scans));
    HIPCHK(lio_grow(&h->d_poses, &h->cap_poses, (size_t)n_scans * 6));
    {
        // every launch leaves the counters at zero (the last workgroup of a scan re-arms it): clear at allocation only
        const unsigned* before = h->d_arrive;
        const size_t cap_before = h->cap_arrive;
        HIPCHK(lio_grow(&h->d_arrive, &h->cap_arrive, (size_t)n_scans));
        if (h->d_arrive != before || h->cap_arrive != cap_before)
            HIPCHK(hipMemsetAsync(h->d_arrive, 0, h->cap_arrive * sizeof(unsigned), h->stream));
    }
    if (h->cfg.nn_cache && !h->cfg.use_lds) HIPCHK(lio_grow(&h->d_nn_cache, &h->cap_nn_cache, tt));

    // launch geometry: one workgroup = LIO_BLOCK * ppt consecutive points of one scan
    int ppt = h->cfg.kernel_variant;
    if (ppt != 1 && ppt != 2 && ppt != 4) ppt = 1;       // auto: one point per thread measured fastest at every batch size tried
    h->ppt = ppt;
    const size_t per_blk = (size_t)LIO_BLOCK * ppt;
    std::vector<LioBlockDesc>& blocks = h->v_blocks;
    blocks.clear();
    int max_blk = 1;
    const size_t old_scans = h->h_state.size();
    h->h_state.resize((size_t)n_scans);
    size_t off = 0;
    for (int s = 0; s < n_scans; ++s) {
        LioScanState& st = h->h_state[s];
        if ((size_t)s >= old_scans) memset(&st, 0, sizeof(st));   // keep matP / is_degenerate of live slots
        st.n_pts = (int)n_pts[s];
        st.offset = (int)off;
        if (h->reg_pose && n_scans == 1) memcpy(st.pose, h->reg_pose, sizeof(float) * 6);
        st.c_n_pts = 0;                // a corner batch has to be uploaded again after every surf batch
        st.c_offset = 0;
        st.done = 1;
        const int nb = (int)((n_pts[s] + per_blk - 1) / per_blk);
        // scan-range sharding ("replicate the map, shard the scan"): this rank takes one
        // block_world-th of the workgroups of every scan; the per-scan sums are all-reduced by the caller
        // a contiguous range of the (tile-sorted) workgroups, rotated from scan to scan so that no rank
        // always gets the same part of the sweep
        const int bw = h->block_world > 1 ? h->block_world : 1, br = h->block_world > 1 ? h->block_rank : 0;
        const int part = (br + s) % bw;
        const int b0 = (int)((long long)nb * part / bw), b1 = (int)((long long)nb * (part + 1) / bw);
        const int nb_local = b1 - b0;
        if (nb_local > max_blk) max_blk = nb_local;
        for (int b = b0; b < b1; ++b) blocks.push_back({ s, (int)(b * per_blk), b - b0, nb_local });
        off += n_pts[s];
    }
    // The caller's records as they are.  A batch that is one contiguous block in the caller's memory -- what a
    // streaming front end keeps in a pinned ring -- goes to the device in ONE copy; pageable or scattered scans are
    // copied one by one; a contiguous batch that already lives in THIS device's memory is not copied at all: the
    // sort kernels read it in place (it must stay valid until the next lio_s2m_batch_sync / _results).
    const unsigned char* stage = h->d_stage;
    bool host_copy = false;            // the caller's (host) buffers are in flight: they are borrowed until the copy is done
    {
        bool contiguous = true;
        for (int s = 0; s + 1 < n_scans && contiguous; ++s)
            contiguous = (const unsigned char*)scans[s + 1] == (const unsigned char*)scans[s] + n_pts[s] * stride;
        bool in_place = false;
        if (contiguous && total) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, scans[0]) == hipSuccess)
                in_place = at.type == hipMemoryTypeDevice && at.device == h->cfg.device_id;
            (void)hipGetLastError();                     // (pageable host memory is reported as an error)
        }
        if (in_place) {
            stage = (const unsigned char*)scans[0];
        } else if (contiguous && total) {
            HIPCHK(hipMemcpyAsync(h->d_stage, scans[0], total * stride, hipMemcpyDefault, h->stream));
            host_copy = true;
        } else {
            host_copy = true;
            size_t o = 0;
            for (int s = 0; s < n_scans; ++s) {
                if (n_pts[s])
                    HIPCHK(hipMemcpyAsync(h->d_stage + o * stride, scans[s], n_pts[s] * stride, hipMemcpyDefault, h->stream));
                o += n_pts[s];
            }
        }
    }
    if (host_copy && h->async_upload) {
        HIPCHK(hipEventRecord(h->ev_map[0], h->stream));   // (async upload: wait for the H2D only, the sort stays in flight)
        HIPCHK(hipEventSynchronize(h->ev_map[0]));
    }
    h->last_stage = stage; h->last_stride = stride; h->last_xyz_off = h->xyz_off;
    h->last_int_off = h->int_off != -2 ? h->int_off : ((h->xyz_off == 0 && stride >= 20) ? 16 : -1);
    h->int_off = -2;
    stage += h->xyz_off;                                 // (x, y, z are read at +0, +4, +8 from here; xyz_off + 12 <= stride)
    h->xyz_off = 0;
    h->v_first_orig.assign((size_t)n_scans + 1, 0);
    for (const LioBlockDesc& b : blocks) h->v_first_orig[b.scan + 1]++;
    for (int s = 0; s < n_scans; ++s) h->v_first_orig[s + 1] += h->v_first_orig[s];
    h->n_scans = n_scans;
    h->total_pts = total;
    h->n_blocks = (int)blocks.size();
    h->max_blk = max_blk;
    HIPCHK(lio_grow(&h->d_blocks, &h->cap_blocks, blocks.size() ? blocks.size() : 1));
    HIPCHK(lio_grow(&h->d_partials, &h->cap_partials, (size_t)n_scans * max_blk * LIO_SUMS));
    if (h->cfg.profile == 2) {
        HIPCHK(lio_grow(&h->d_stamps, &h->cap_stamps, blocks.size() * (LIO_BLOCK / 64) * 8 + 8));
        HIPCHK(hipMemsetAsync(h->d_stamps, 0, (blocks.size() * (LIO_BLOCK / 64) * 8 + 8) * sizeof(long long), h->stream));
    }
    if (h->cfg.record_corr_iter >= 0) {
        HIPCHK(lio_grow(&h->d_rec_flag, &h->cap_rec_flag, tt));
        HIPCHK(lio_grow(&h->d_rec_coeff, &h->cap_rec_coeff, tt * 4));
        HIPCHK(lio_grow(&h->d_rec_nn, &h->cap_rec_nn, tt * 5));
        HIPCHK(hipMemsetAsync(h->d_rec_flag, 0, tt, h->stream));
        HIPCHK(hipMemsetAsync(h->d_rec_coeff, 0, tt * 4 * sizeof(float), h->stream));
        HIPCHK(hipMemsetAsync(h->d_rec_nn, 0xff, tt * 5 * sizeof(int), h->stream));
    }
    if (!blocks.empty())
        HIPCHK(hipMemcpyAsync(h->d_blocks, blocks.data(), blocks.size() * sizeof(LioBlockDesc),
                              hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_state, h->h_state.data(), (size_t)n_scans * sizeof(LioScanState),
                          hipMemcpyHostToDevice, h->stream));
    std::vector<LioScanTiles>& tiles = h->v_tiles;
    std::vector<LioBlockDesc>& prep = h->v_prep;
    tiles.clear(); prep.clear();
    h->sorted = false;
    // the tile sort pays for itself on batches; a lone small scan skips its eight launches
    const bool want_sort = total && (h->cfg.sort_scan >= 2 || (h->cfg.sort_scan == 1 && total >= 65536));
    // (up to 16383 points: the key (tile id << 14) | index of a 16384th point in tile 262143 would be the 0xffffffff that marks
    // an empty slot of k_scan_sort_radix)
    if (want_sort && max_n < 16384 && h->cfg.sort_scan != 3) {
        // every scan fits one workgroup's LDS: bounding box, tile keys, sort and gather in ONE launch, nothing read back
        HIPCHK(lio_grow(&h->d_perm, &h->cap_perm, tt));
        lio_launch_scan_sort_lds(stage, stride, h->d_state, n_scans, (int)max_n, h->cfg.tile_size > 0.0f ? h->cfg.tile_size : 4.0f,
                                 h->shard.axis, h->d_perm, h->d_sx, h->d_sy, h->d_sz, h->stream);
        h->sorted = true;
    } else if (want_sort) {
        // scan-local tile grids from the scans' bounding boxes, reduced on the device (the caller's memory is
        // never read by the host: it may be pinned, pageable or device memory)
        for (int s = 0; s < n_scans; ++s) {
            const int nb1 = (int)((n_pts[s] + LIO_BLOCK - 1) / LIO_BLOCK);
            for (int b = 0; b < nb1; ++b) prep.push_back({ s, b * LIO_BLOCK, b, nb1 });
        }
        HIPCHK(lio_grow(&h->d_prep_blocks, &h->cap_prep_blocks, prep.size()));
        HIPCHK(lio_grow(&h->d_scan_bbox, &h->cap_scan_bbox, (size_t)n_scans * 6));
        if (h->cap_h_scan_bbox < (size_t)n_scans * 6) {
            if (h->h_scan_bbox) HIPCHK(hipHostFree(h->h_scan_bbox));
            h->h_scan_bbox = nullptr; h->cap_h_scan_bbox = 0;
            HIPCHK(hipHostMalloc((void**)&h->h_scan_bbox, ((size_t)n_scans * 6 + 64) * sizeof(unsigned), hipHostMallocDefault));
            h->cap_h_scan_bbox = (size_t)n_scans * 6 + 64;
        }
        for (int s = 0; s < n_scans; ++s)
            for (int a = 0; a < 6; ++a) h->h_scan_bbox[s * 6 + a] = a < 3 ? 0xffffffffu : 0u;
        HIPCHK(hipMemcpyAsync(h->d_scan_bbox, h->h_scan_bbox, (size_t)n_scans * 6 * sizeof(unsigned), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_prep_blocks, prep.data(), prep.size() * sizeof(LioBlockDesc), hipMemcpyHostToDevice, h->stream));
        lio_launch_scan_bbox(stage, stride, h->d_prep_blocks, (int)prep.size(), h->d_state, h->d_scan_bbox, h->stream);
        HIPCHK(hipMemcpyAsync(h->h_scan_bbox, h->d_scan_bbox, (size_t)n_scans * 6 * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));          // (this handle's stream only: a sibling handle keeps computing)
        tiles.resize((size_t)n_scans);
        long long n_keys = 0;
        for (int s = 0; s < n_scans; ++s) {
            float mn[3], mx[3];
            for (int a = 0; a < 3; ++a) {
                const unsigned lo = h->h_scan_bbox[s * 6 + a], hi = h->h_scan_bbox[s * 6 + 3 + a];
                const bool none = lo == 0xffffffffu;
                mn[a] = none ? 0.0f : lio_ord2f(lo);
                mx[a] = none ? 0.0f : lio_ord2f(hi);
            }
            float tile = h->cfg.tile_size > 0.0f ? h->cfg.tile_size : 4.0f;
            LioScanTiles t;
            for (;;) {
                t.inv_tile = 1.0f / tile;
                const double ex = floor(((double)mx[0] - mn[0]) * t.inv_tile) + 1.0, ey = floor(((double)mx[1] - mn[1]) * t.inv_tile) + 1.0,
                             ez = floor(((double)mx[2] - mn[2]) * t.inv_tile) + 1.0;
                if (ex * ey * ez <= 262144.0) { t.ntx = (int)ex; t.nty = (int)ey; t.ntz = (int)ez; break; }
                tile *= 2.0f;
            }
            t.ox = mn[0]; t.oy = mn[1]; t.oz = mn[2];
            t.key_offset = (int)n_keys;
            // Order of the tiles = order of the association workgroups' points.  Unsharded: x fastest.  With a sharded map
            // the shard axis varies slowest, so a workgroup's 256 points form a slice about one tile thick ACROSS that
            // axis and k_shard_cull can drop it on every rank but one or two.  (Lidar frame; effective while the
            // vehicle's heading stays within ~45 degrees of a map axis, merely less effective -- never wrong -- otherwise.)
            t.mx = 1; t.my = t.ntx; t.mz = t.ntx * t.nty;
            if (h->shard.axis == 0) { t.mz = 1; t.my = t.ntz; t.mx = t.ntz * t.nty; }
            else if (h->shard.axis == 1) { t.mx = 1; t.mz = t.ntx; t.my = t.ntx * t.ntz; }
            tiles[s] = t;
            n_keys += (long long)t.ntx * t.nty * t.ntz;
        }
        if (n_keys < 0x7fffffffLL - 1024) {
            HIPCHK(lio_grow(&h->d_tiles, &h->cap_tiles, (size_t)n_scans));
            HIPCHK(lio_grow(&h->d_key_of, &h->cap_key_of, tt));
            HIPCHK(lio_grow(&h->d_big_list, &h->cap_big_list, tt / 1024 + 2));
            HIPCHK(lio_grow(&h->d_tmp_idx, &h->cap_tmp_idx, tt));
            HIPCHK(lio_grow(&h->d_perm, &h->cap_perm, tt));
            HIPCHK(lio_grow(&h->d_key_count, &h->cap_key_count, (size_t)n_keys));
            HIPCHK(lio_grow(&h->d_key_start, &h->cap_key_start, (size_t)n_keys + 1));
            HIPCHK(lio_grow(&h->d_key_tiles, &h->cap_key_tiles, (size_t)lio_scan_tiles((int)n_keys) + 1));
            HIPCHK(hipMemcpyAsync(h->d_tiles, tiles.data(), tiles.size() * sizeof(LioScanTiles), hipMemcpyHostToDevice, h->stream));
            lio_launch_scan_tile_sort(stage, stride, (int)total, h->d_prep_blocks, (int)prep.size(), h->d_state,
                                      h->d_tiles, (int)n_keys, h->d_key_of, h->d_key_count, h->d_key_start,
                                      h->d_key_tiles, h->d_tmp_idx, h->d_perm, h->d_big_list, h->d_big_list + (h->cap_big_list - 1),
                                      h->d_sx, h->d_sy, h->d_sz, h->stream);
            h->sorted = true;
        }
    }
    h->soa_valid = true;
    if (total && !h->sorted) {
        // a batch the one-launch loop will take (lio_persist_eligible) reads its points once, from the records: no AoS -> SoA launch
        if (lio_persist_eligible(h)) h->soa_valid = false;
        else lio_launch_aos_to_soa(stage, stride, (int)total, h->d_sx, h->d_sy, h->d_sz, nullptr, h->stream);
    }
    h->has_block_box = false;
    if (h->shard.axis >= 0 && ppt == 1 && !blocks.empty()) {
        // map sharding: the box of every workgroup's points, for the cull at the head of k_s2m_iterate
        HIPCHK(lio_grow(&h->d_block_box, &h->cap_block_box, (tt / LIO_BLOCK + (size_t)n_scans + 2) * 6));
        HIPCHK(lio_grow(&h->d_blk_skip, &h->cap_blk_skip, blocks.size()));
        lio_launch_block_boxes(h->d_blocks, (int)blocks.size(), h->d_state, h->d_sx, h->d_sy, h->d_sz, h->d_block_box, h->stream);
        h->has_block_box = true;
    }
    if (!h->defer_sync) HIPCHK(hipStreamSynchronize(h->stream));   // the caller's scans are borrowed only for this call
    HIPCHK(hipGetLastError());
    h->poses_set = false;
    h->pose_in_state = h->reg_pose != nullptr && n_scans == 1;
    h->ran = false;
    h->graph_dirty = true;
    h->corner_active = false;
    return LIO_OK;
}

// ------------------------------------------------- corner residuals (extension)
extern "C" int lio_s2m_set_corner_map(lio_s2m_handle* h, const void* pts, size_t n, size_t stride)
{
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    if (h->multi) return lio_fail(LIO_ERR_ARG, "the corner extension is not available on a multi-device handle");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();
    if (!h->corner) {
        lio_s2m_config cc = h->cfg;
        cc.kernel_variant = 1;         // edge sets are small: one point per thread, candidates from global memory
        cc.use_lds = 0;
        cc.profile = 0;
        cc.use_graph = 0;
        cc.pipeline = 1;               // the edge launch is the fused kernel's CORNER instantiation
        int rc = lio_s2m_create(&cc, &h->corner);
        if (rc != LIO_OK) return rc;
        if ((rc = lio_s2m_set_stream(h->corner, h->stream)) != LIO_OK) return rc;
    }
    h->corner_active = false;
    h->graph_dirty = true;
    return lio_s2m_set_map(h->corner, pts, n, stride);
}

extern "C" int lio_s2m_batch_upload_corners(lio_s2m_handle* h, int32_t n_scans, const void* const* scans,
                                            const size_t* n_pts, size_t stride)
{
    if (!h || !scans || !n_pts) return lio_fail(LIO_ERR_ARG, "null argument");
    if (!h->corner || !h->corner->has_map) return lio_fail(LIO_ERR_NO_MAP, "set_corner_map has not been called");
    if (h->n_scans < 1 || n_scans != h->n_scans)
        return lio_fail(LIO_ERR_ARG, "upload the surf batch first; the corner batch must have the same number of scans");
    lio_s2m_handle* ch = h->corner;
    ch->block_rank = h->block_rank;
    ch->block_world = h->block_world;
    ch->defer_sync = h->defer_sync;
    int rc = lio_s2m_batch_upload(ch, n_scans, scans, n_pts, stride);
    ch->defer_sync = false;
    if (rc != LIO_OK) return rc;
    // One arrival counter and one partial-sum table per scan serve both launches: the scan's corner
    // chunks are numbered after its surf chunks (combineOptimizationCoeffs appends one list to the other).
    int max_blk = 1;
    for (LioBlockDesc& b : h->v_blocks) {
        const int ns = h->v_first_orig[b.scan + 1] - h->v_first_orig[b.scan];
        const int nc = ch->v_first_orig[b.scan + 1] - ch->v_first_orig[b.scan];
        b.n_blk = ns + nc;
        if (b.n_blk > max_blk) max_blk = b.n_blk;
    }
    for (LioBlockDesc& b : ch->v_blocks) {
        const int ns = h->v_first_orig[b.scan + 1] - h->v_first_orig[b.scan];
        const int nc = ch->v_first_orig[b.scan + 1] - ch->v_first_orig[b.scan];
        b.blk += ns;
        b.n_blk = ns + nc;
        if (b.n_blk > max_blk) max_blk = b.n_blk;
    }
    h->max_blk = max_blk;
    HIPCHK(lio_grow(&h->d_partials, &h->cap_partials, (size_t)n_scans * max_blk * LIO_SUMS));
    if (!h->v_blocks.empty())
        HIPCHK(hipMemcpyAsync(h->d_blocks, h->v_blocks.data(), h->v_blocks.size() * sizeof(LioBlockDesc),
                              hipMemcpyHostToDevice, h->stream));
    if (!ch->v_blocks.empty())
        HIPCHK(hipMemcpyAsync(ch->d_blocks, ch->v_blocks.data(), ch->v_blocks.size() * sizeof(LioBlockDesc),
                              hipMemcpyHostToDevice, h->stream));
    for (int s = 0; s < n_scans; ++s) {
        h->h_state[s].c_n_pts = ch->h_state[s].n_pts;
        h->h_state[s].c_offset = ch->h_state[s].offset;
    }
    HIPCHK(hipMemcpyAsync(h->d_state, h->h_state.data(), (size_t)n_scans * sizeof(LioScanState),
                          hipMemcpyHostToDevice, h->stream));
    if (!h->defer_sync) HIPCHK(hipStreamSynchronize(h->stream));
    h->corner_active = true;
    h->poses_set = false;             // (the workgroup list was rewritten: batch_set_poses re-orders it)
    h->graph_dirty = true;
    return LIO_OK;
}

extern "C" int lio_s2m_get_corner_correspondences(lio_s2m_handle* h, int32_t scan, uint8_t* flag,
                                                  float* coeff4, int32_t* nn_idx5)
{
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    if (!h->corner || !h->corner_active) return lio_fail(LIO_ERR_ARG, "no corner batch is resident");
    return lio_s2m_get_correspondences(h->corner, scan, flag, coeff4, nn_idx5);
}

extern "C" int lio_s2m_batch_set_poses(lio_s2m_handle* h, const float* poses)
{
    if (!h || !poses) return lio_fail(LIO_ERR_ARG, "null argument");
    if (h->n_scans < 1) return lio_fail(LIO_ERR_ARG, "no batch uploaded");
    if (h->multi) return lio_multi_set_poses(h, poses);
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();   // drop stale codes left by other HIP users of this thread (e.g. hipErrorNotReady)
    h->run_pending = false;            // (a run whose results were never fetched is abandoned)
    h->pose_in_state = false;
    HIPCHK(hipMemcpyAsync(h->d_poses, poses, (size_t)h->n_scans * 6 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (h->cfg.sort_batch && h->n_scans > 8 && !h->v_blocks.empty()) {
        // Locality only: order the workgroup list by where the scans ARE (position along the map's
        // longest axis).  Together with the XCD-aware workgroup order every XCD then streams the
        // map rows of one stretch of the trajectory, which fit its private 4 MB L2.
        const LioGrid& mg = lio_map_of(h)->grid;
        const float ext[3] = { (float)mg.nx, (float)mg.ny, (float)mg.nz };
        const int ax = (ext[0] >= ext[1] && ext[0] >= ext[2]) ? 0 : (ext[1] >= ext[2] ? 1 : 2);
        std::vector<int>& order = h->v_order;
        order.resize((size_t)h->n_scans);
        for (int s = 0; s < h->n_scans; ++s) order[s] = s;
        std::stable_sort(order.begin(), order.end(),
                         [&](int a, int b) { return poses[a * 6 + 3 + ax] < poses[b * 6 + 3 + ax]; });
        // v_blocks is grouped by scan in upload order: gather the groups in the new order
        std::vector<int>& first = h->v_first;
        first.assign((size_t)h->n_scans + 1, 0);
        for (const LioBlockDesc& b : h->v_blocks) first[b.scan + 1]++;
        for (int s = 0; s < h->n_scans; ++s) first[s + 1] += first[s];
        std::vector<LioBlockDesc>& sorted = h->v_blocks_sorted;
        sorted.clear();
        sorted.reserve(h->v_blocks.size());
        // (v_blocks itself stays in scan order so that the grouping survives repeated calls)
        for (int k = 0; k < h->n_scans; ++k) {
            const int s = order[k];
            // blocks of scan s are the ones whose .scan == s; they are contiguous in the ORIGINAL list
            for (int i = h->v_first_orig[s]; i < h->v_first_orig[s + 1]; ++i) sorted.push_back(h->v_blocks[i]);
        }
        HIPCHK(hipMemcpyAsync(h->d_blocks, sorted.data(), sorted.size() * sizeof(LioBlockDesc), hipMemcpyHostToDevice, h->stream));
    }
    if (!h->defer_sync) HIPCHK(hipStreamSynchronize(h->stream));
    h->poses_set = true;
    return LIO_OK;
}

extern "C" int lio_s2m_set_degeneracy(lio_s2m_handle* h, int32_t scan, const float matP[36], int32_t is_degenerate)
{
    if (!h || !matP) return lio_fail(LIO_ERR_ARG, "null argument");
    if (scan < 0 || scan >= h->n_scans) return lio_fail(LIO_ERR_ARG, "scan slot out of range");
    if (h->multi) return lio_multi_set_degeneracy(h, scan, matP, is_degenerate);
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();   // drop stale codes left by other HIP users of this thread (e.g. hipErrorNotReady)
    LioScanState& st = h->h_state[scan];
    memcpy(st.matP, matP, sizeof(float) * 36);
    st.is_degenerate = is_degenerate;
    char* base = (char*)(h->d_state + scan);
    HIPCHK(hipMemcpyAsync(base + offsetof(LioScanState, matP), st.matP, sizeof(float) * 36, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(base + offsetof(LioScanState, is_degenerate), &st.is_degenerate, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LIO_OK;
}

static void lio_fill_params(lio_s2m_handle* h, LioIterParams& P, double* sums_out)
{
    const lio_s2m_handle* m = lio_map_of(h);       // (a sibling's resident map after lio_s2m_share_map)
    P.grid = m->grid;
    P.shard = h->shard;
    P.c = h->c;
    P.map_sorted = m->d_sorted;
    P.map_xyz4 = m->d_map4;
    P.cell_start = m->d_cell_start;
    P.nbr_pts = m->d_nbr_pts;
    P.nbr_start = m->d_nbr_start;
    P.sx = h->d_sx; P.sy = h->d_sy; P.sz = h->d_sz;
    P.perm = h->sorted ? h->d_perm : nullptr;
    P.state = h->d_state;
    P.blocks = h->d_blocks;
    P.partials = h->d_partials;
    P.arrive = h->d_arrive;
    P.max_blk = h->max_blk;
    P.xcd_remap = h->cfg.xcd_remap;
    P.n_active = h->d_active;
    P.sums_out = sums_out;
    const bool rec = h->cfg.record_corr_iter >= 0;
    P.rec_flag = rec ? h->d_rec_flag : nullptr;
    P.rec_coeff = rec ? h->d_rec_coeff : nullptr;
    P.rec_nn = rec ? h->d_rec_nn : nullptr;
    P.stamps = (h->cfg.profile == 2) ? h->d_stamps : nullptr;
    P.d5_cache = (h->cfg.nn_cache && !h->cfg.use_lds) ? h->d_nn_cache : nullptr;
    P.blk_skip = (h->shard.axis >= 0 && h->has_block_box && sums_out != nullptr && !h->cfg.use_lds && h->ppt == 1) ? h->d_blk_skip : nullptr;
    // With a sharded map the workgroups a rank keeps belong to the scans near its slab, and the batch's workgroups are
    // ordered by scan position: a contiguous stretch of the list, i.e. ONE OR TWO of the eight XCDs under the XCD-aware
    // order (measured: 0.82 ms against 0.47 ms for the same launch).  Deal them round-robin instead.
    if (P.blk_skip) P.xcd_remap = 0;
}

// Arguments of the corner launch: the child's map, grid, edge points and workgroup list; everything
// that is per scan (state, partial sums, arrival counters, active count) is the parent's.
static void lio_fill_params_corner(lio_s2m_handle* h, LioIterParams& Pc, double* sums_out)
{
    lio_fill_params(h->corner, Pc, sums_out);
    Pc.shard = h->shard;
    Pc.state = h->d_state;
    Pc.partials = h->d_partials;
    Pc.arrive = h->d_arrive;
    Pc.max_blk = h->max_blk;
    Pc.n_active = h->d_active;
    Pc.stamps = nullptr;
}

// One Gauss-Newton iteration of the batch: cornerOptimization (if a corner batch is resident), then
// surfOptimization MO:1618-1687; the workgroup that arrives last on a scan's counter runs
// LMOptimization MO:1702-1837 for it.
static void lio_launch_gn(lio_s2m_handle* h, const LioIterParams& P, const LioIterParams* Pc)
{
    if (Pc) lio_launch_iterate(*Pc, h->corner->n_blocks, 1, false, h->stream, true);
    lio_launch_iterate(P, h->n_blocks, h->ppt, h->cfg.use_lds != 0, h->stream);
}

extern "C" int lio_s2m_batch_begin(lio_s2m_handle* h)
{
    if (h && h->multi) return lio_fail(LIO_ERR_ARG, "not available on a multi-device handle (cfg.n_devices > 1 shards inside the library)");
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    if (!lio_map_of(h)->has_map) return lio_fail(LIO_ERR_NO_MAP, "set_map has not been called");
    if (h->n_scans < 1 || !h->poses_set) return lio_fail(LIO_ERR_ARG, "batch_upload and batch_set_poses first");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();   // drop stale codes left by other HIP users of this thread (e.g. hipErrorNotReady)
    if (h->map_src && h->map_epoch != h->map_src->map_epoch) {      // the shared map was replaced since the last run
        h->map_epoch = h->map_src->map_epoch;
        h->graph_dirty = true;
        HIPCHK(hipStreamWaitEvent(h->stream, h->map_src->ev_mapl[1], 0));   // (its build may still be in flight on the owner's stream)
    }
    lio_launch_init_state(h->d_state, h->n_scans, h->d_poses, h->pose_in_state, h->c, h->d_active, h->stream);
    h->pose_in_state = false;          // (d_poses holds the guess from now on)
    // search-bound cache: iteration 0 never reads it and rewrites the entry of every point it processes; entries
    // of points it does not process (owned by another rank) could date from an earlier run -> drop them
    // (0xff bytes = NaN, which fails the `>= 0` validity test like -1 does)
    if (h->cfg.nn_cache && !h->cfg.use_lds && h->d_nn_cache && h->shard.axis >= 0)
        HIPCHK(hipMemsetAsync(h->d_nn_cache, 0xff, (h->total_pts ? h->total_pts : 1) * sizeof(float), h->stream));
    if (h->corner_active && h->corner->d_nn_cache && h->shard.axis >= 0)
        HIPCHK(hipMemsetAsync(h->corner->d_nn_cache, 0xff, (h->corner->total_pts ? h->corner->total_pts : 1) * sizeof(float), h->stream));
    h->launches_this_run = 0;
    h->units_this_run = 0;
    h->unit_iters = 1;
    h->ran = true;
    return LIO_OK;
}

// (Re)capture `chunk` consecutive GN-iteration launches into a hipGraph.  Every launch has the
// same arguments -- all per-iteration state lives in device memory -- so one executable graph
// serves the whole loop; scans that are done turn their workgroups into immediate exits.
static int lio_graph_prepare(lio_s2m_handle* h, const LioIterParams& P, const LioIterParams* Pc, int chunk)
{
    // the cached graph stays valid as long as the kernel arguments and the launch geometry are the same
    const int nbc = Pc ? h->corner->n_blocks : -1;
    if (h->graph_exec && h->graph_chunk == chunk && h->graph_blocks == h->n_blocks && h->graph_ppt == h->ppt &&
        memcmp(&h->graph_params, &P, sizeof(P)) == 0 && h->graph_blocks_c == nbc &&
        (!Pc || memcmp(&h->graph_params_c, Pc, sizeof(P)) == 0))
        return LIO_OK;
    if (h->graph_exec) { HIPCHK(hipGraphExecDestroy(h->graph_exec)); h->graph_exec = nullptr; }
    if (h->graph) { HIPCHK(hipGraphDestroy(h->graph)); h->graph = nullptr; }
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < chunk; ++i) lio_launch_gn(h, P, Pc);
    HIPCHK(hipStreamEndCapture(h->stream, &h->graph));
    HIPCHK(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
    h->graph_chunk = chunk; h->graph_blocks = h->n_blocks; h->graph_ppt = h->ppt;
    memcpy(&h->graph_params, &P, sizeof(P));
    h->graph_blocks_c = nbc;
    if (Pc) memcpy(&h->graph_params_c, Pc, sizeof(P));
    h->graph_dirty = false;
    return LIO_OK;
}

// The GN loop (MO:1848-1859) as a resumable launch loop.  After every launch unit (one launch, or one replay of a
// captured chunk of `graph_iters` launches) the count of still-iterating scans is copied to pinned memory; unit u
// is only enqueued once the count after unit u-1-lookahead is known to be non-zero (lookahead 0 never enqueues an
// empty launch; larger values keep the queue fed for small batches).  `blocking == false` (lio_s2m_batch_run)
// enqueues what can be enqueued WITHOUT waiting for the device and returns -- the caller is free to upload the
// next batch on a sibling handle --; lio_s2m_batch_sync / _results resume the loop with `blocking == true`.
static int lio_run_continue(lio_s2m_handle* h, bool blocking)
{
    if (!h->run_pending) return LIO_OK;
    const bool prof = h->cfg.profile != 0;
    const LioIterParams* Pc = h->run_has_c ? &h->run_Pc : nullptr;
    while (h->run_next < h->run_units && h->run_next < LIO_MAX_ITERS) {          // MO:1848
        const int u = h->run_next;
        const int chk = u - 1 - h->run_look;
        if (chk >= 0) {
            if (blocking) {
                HIPCHK(hipEventSynchronize(h->ev_chk[chk]));
            } else {
                const hipError_t q = hipEventQuery(h->ev_chk[chk]);
                if (q == hipErrorNotReady) { (void)hipGetLastError(); return LIO_OK; }   // resumed by sync / results
                HIPCHK(q);
            }
            if (h->h_active[chk] == 0) break;
        }
        if (prof) HIPCHK(hipEventRecord(h->ev_beg[u], h->stream));
        if (h->run_persist) {
            lio_launch_persist(h->run_P, h->n_blocks, h->d_gen, h->gen_epoch, h->soa_valid ? nullptr : h->last_stage + h->last_xyz_off,
                               h->last_stride, h->n_scans, getenv("LIO_NO_SPEC") ? nullptr : h->d_gen + h->n_scans, h->d_spec_sums, h->d_poses,
                               lio_persist_spin_max(h), h->persist_withhold, h->stream);
        } else if (h->run_graph) HIPCHK(hipGraphLaunch(h->graph_exec, h->stream));
        else lio_launch_gn(h, h->run_P, Pc);
        if (prof) HIPCHK(hipEventRecord(h->ev_end[u], h->stream));
        if (!h->run_persist) HIPCHK(hipMemcpyAsync(&h->h_active[u], h->d_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipEventRecord(h->ev_chk[u], h->stream));
        h->run_next = u + 1;
        h->units_this_run = u + 1;
        h->launches_this_run = (u + 1) * h->unit_iters;
    }
    h->run_pending = false;
    HIPCHK(hipGetLastError());
    return LIO_OK;
}

extern "C" int lio_s2m_batch_run(lio_s2m_handle* h)
{
    if (h && h->multi) return lio_multi_run(h);
    int rc = lio_s2m_batch_begin(h);
    if (rc != LIO_OK) return rc;
    LioIterParams& P = h->run_P;
    memset(&P, 0, sizeof(P));          // (padding bytes take part in the graph-cache comparison)
    lio_fill_params(h, P, nullptr);
    memset(&h->run_Pc, 0, sizeof(h->run_Pc));
    h->run_has_c = h->corner_active && h->corner->n_blocks > 0;
    if (h->run_has_c) lio_fill_params_corner(h, h->run_Pc, nullptr);
    int look = h->cfg.lookahead;
    if (look < 0) look = (h->total_pts >= 200000) ? 0 : 2;
    // use_graph: a unit is one replay of a captured chunk of `graph_iters` iterations
    h->run_persist = lio_persist_eligible(h) && !h->run_has_c;         // (see lio_persist_eligible)
    if (!h->run_persist && (rc = lio_ensure_soa(h)) != LIO_OK) return rc;
    const bool graph = !h->run_persist && h->cfg.use_graph != 0 && h->cfg.profile != 2 && h->cfg.record_corr_iter < 0 && h->n_blocks > 0;
    int chunk = 1;
    if (graph) {
        chunk = h->cfg.graph_iters > 0 ? h->cfg.graph_iters : 4;
        if (chunk > h->cfg.max_iters) chunk = h->cfg.max_iters;
        if ((rc = lio_graph_prepare(h, P, h->run_has_c ? &h->run_Pc : nullptr, chunk)) != LIO_OK) return rc;
        if (h->cfg.lookahead < 0) look = 0;              // a chunk already is a run-ahead of `chunk` launches
    }
    if (h->run_persist) {
        if ((size_t)h->n_scans * 2 > h->cap_gen || !h->d_gen) {
            HIPCHK(lio_grow(&h->d_gen, &h->cap_gen, (size_t)h->n_scans * 2));
            HIPCHK(hipMemsetAsync(h->d_gen, 0, h->cap_gen * sizeof(unsigned), h->stream));   // (fresh generation numbers / speculation states)
            h->gen_epoch = 0;
        }
        HIPCHK(lio_grow(&h->d_spec_sums, &h->cap_spec_sums, (size_t)h->n_scans * LIO_SUMS));
        h->gen_epoch += 128;
        chunk = h->cfg.max_iters;                        // one unit = the whole loop
    }
    h->run_graph = graph;
    h->run_look = look;
    h->run_units = (h->cfg.max_iters + chunk - 1) / chunk;
    h->run_next = 0;
    h->unit_iters = h->run_persist ? 1 : chunk;
    h->units_this_run = 0;
    h->launches_this_run = 0;
    h->run_pending = true;
    // a lone registration (lio_s2m_register) has nothing to overlap with: drive the loop to the end right away
    return lio_run_continue(h, h->defer_sync);
}

extern "C" int lio_s2m_batch_iter_partial(lio_s2m_handle* h, double* d_sums)
{
    if (!h || !d_sums) return lio_fail(LIO_ERR_ARG, "null argument");
    if (!h->ran) return lio_fail(LIO_ERR_ARG, "batch_begin first");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();
    { const int rcs = lio_ensure_soa(h); if (rcs != LIO_OK) return rcs; }
    LioIterParams P;
    lio_fill_params(h, P, d_sums);
    HIPCHK(hipMemsetAsync(d_sums, 0, (size_t)h->n_scans * LIO_SUMS * sizeof(double), h->stream));
    const int it = h->launches_this_run;
    const bool prof = h->cfg.profile != 0 && it < LIO_MAX_ITERS;
    if (prof) HIPCHK(hipEventRecord(h->ev_beg[it], h->stream));
    LioIterParams Pcs;
    const bool with_corners = h->corner_active && h->corner->n_blocks > 0;
    if (with_corners) lio_fill_params_corner(h, Pcs, d_sums);
    if (P.blk_skip && !getenv("LIO_NO_CULL")) {
        lio_launch_shard_cull(P, h->plan_ranks, h->plan_rank, h->plan_halo, h->plan_bounds, h->d_block_box, h->n_blocks, h->d_blk_skip,
                              h->stream);
        if (getenv("LIO_CULL_STATS")) {                  // diagnostics only (synchronises)
            std::vector<unsigned char> sk((size_t)h->n_blocks);
            HIPCHK(hipMemcpyAsync(sk.data(), h->d_blk_skip, sk.size(), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            size_t c = 0;
            for (unsigned char v : sk) c += v;
            size_t c2 = 0;
            for (unsigned char v : sk) c2 += (v == 2);
            c = 0;
            for (unsigned char v : sk) c += (v == 1);
            fprintf(stderr, "[liogpu] shard cull: %zu of %d workgroups skipped, %zu wholly owned (iteration %d)\n", c, h->n_blocks, c2, it);
        }
    } else {
        P.blk_skip = nullptr;
    }
    lio_launch_gn(h, P, with_corners ? &Pcs : nullptr);
    if (prof) HIPCHK(hipEventRecord(h->ev_end[it], h->stream));
    h->launches_this_run++;
    h->units_this_run = h->launches_this_run;
    HIPCHK(hipGetLastError());
    return LIO_OK;
}

extern "C" int lio_s2m_batch_iter_apply(lio_s2m_handle* h, const double* d_sums)
{
    return lio_s2m_iter_apply_slots(h, d_sums, 0, 1);
}

int lio_s2m_iter_apply_slots(lio_s2m_handle* h, const double* d_sums, size_t slot_stride, int n_slots)
{
    if (!h || !d_sums || n_slots < 1) return lio_fail(LIO_ERR_ARG, "null argument");
    if (!h->ran || h->launches_this_run < 1) return lio_fail(LIO_ERR_ARG, "batch_iter_partial first");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();
    lio_launch_apply(h->d_state, h->n_scans, d_sums, slot_stride, n_slots, h->c, h->d_active, h->stream);
    // publish the number of still-iterating scans after this iteration (see lio_s2m_batch_poll_active)
    const int it = h->launches_this_run - 1;
    if (it < LIO_MAX_ITERS) {
        HIPCHK(hipMemcpyAsync(&h->h_active[it], h->d_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipEventRecord(h->ev_chk[it], h->stream));
    }
    HIPCHK(hipGetLastError());
    return LIO_OK;
}

extern "C" int lio_s2m_batch_poll_active(lio_s2m_handle* h, int32_t iteration, int32_t* n_active)
{
    if (!h || !n_active) return lio_fail(LIO_ERR_ARG, "null argument");
    if (iteration < 0 || iteration >= h->launches_this_run || iteration >= LIO_MAX_ITERS)
        return lio_fail(LIO_ERR_ARG, "iteration has not been applied");
    HIPCHK(hipEventSynchronize(h->ev_chk[iteration]));
    *n_active = h->h_active[iteration];
    return LIO_OK;
}

extern "C" int lio_s2m_batch_n_active(lio_s2m_handle* h, int32_t* n_active)
{
    if (!h || !n_active) return lio_fail(LIO_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();   // drop stale codes left by other HIP users of this thread (e.g. hipErrorNotReady)
    HIPCHK(hipMemcpyAsync(&h->h_active[0], h->d_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *n_active = h->h_active[0];
    return LIO_OK;
}

extern "C" int lio_s2m_batch_sync(lio_s2m_handle* h)
{
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    if (h->multi) return lio_multi_sync(h);
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();   // drop stale codes left by other HIP users of this thread (e.g. hipErrorNotReady)
    { const int rcc = lio_run_continue(h, true); if (rcc != LIO_OK) return rcc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    return LIO_OK;
}

extern "C" int lio_s2m_batch_results(lio_s2m_handle* h, float* poses, lio_s2m_result* results)
{
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    if (h->multi) return lio_multi_results(h, poses, results);
    if (!h->ran) return lio_fail(LIO_ERR_ARG, "nothing has been run");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();   // drop stale codes left by other HIP users of this thread (e.g. hipErrorNotReady)
    { const int rcc = lio_run_continue(h, true); if (rcc != LIO_OK) return rcc; }
    if (results) {
        HIPCHK(hipMemcpyAsync(h->h_state.data(), h->d_state, (size_t)h->n_scans * sizeof(LioScanState),
                              hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->host_state_stale = false;
    } else {
        // poses only: 40 bytes per scan through a pinned buffer instead of the whole state (1.5 KB per scan)
        const size_t need = (size_t)h->n_scans * 10;
        HIPCHK(lio_grow(&h->d_summary, &h->cap_summary, need));
        if (h->cap_h_summary < need) {
            if (h->h_summary) HIPCHK(hipHostFree(h->h_summary));
            h->h_summary = nullptr; h->cap_h_summary = 0;
            HIPCHK(hipHostMalloc((void**)&h->h_summary, (need + 64) * sizeof(float), hipHostMallocDefault));
            h->cap_h_summary = need + 64;
        }
        h->host_state_stale = true;
        lio_launch_pack_summary(h->d_state, h->n_scans, h->d_summary, h->stream);
        HIPCHK(hipMemcpyAsync(h->h_summary, h->d_summary, need * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int s = 0; s < h->n_scans; ++s) {
            LioScanState& st = h->h_state[s];
            const float* o = h->h_summary + (size_t)s * 10;
            memcpy(st.pose, o, sizeof(float) * 6);
            memcpy(&st.iter, o + 6, 4); memcpy(&st.status, o + 7, 4);
            st.done = (st.status & 0x100) ? 0 : 1;
            st.status &= 0xff;
            memcpy(&st.converged, o + 8, 4); memcpy(&st.is_degenerate, o + 9, 4);
        }
    }
    HIPCHK(hipGetLastError());
    if (h->run_persist) {
        // Every scan leaves a one-launch loop done; one that did not gave up waiting at its barrier (bounded polls): its
        // workgroups were not all resident (another client holds compute units) or never arrived.  Recover inside this
        // call: clear the arrival counters the aborted launch left armed (every later run on this handle would otherwise
        // find its "last workgroup" one arrival early -- round-2 advisor finding), and run the same registrations through
        // the launch loop from the saved initial guesses (d_poses).  What the aborted launch may already have written into
        // the persistent members (isDegenerate / matP of a first solve) is what the launch loop's first solve computes from
        // the same sums, bit for bit, so nothing else needs restoring.
        bool incomplete = false;
        for (int s = 0; s < h->n_scans; ++s) incomplete = incomplete || !h->h_state[s].done;
        if (incomplete) {
            h->persist_fallbacks++;
            h->prof.persist_fallbacks = h->persist_fallbacks;
            HIPCHK(hipMemsetAsync(h->d_arrive, 0, h->cap_arrive * sizeof(unsigned), h->stream));
            const bool keep_defer = h->defer_sync;
            h->no_persist = true; h->defer_sync = true;
            h->poses_set = true; h->pose_in_state = false;           // (k_s2m_init_state left the guesses in d_poses)
            int rc = lio_s2m_batch_run(h);
            h->defer_sync = keep_defer;
            if (rc == LIO_OK) rc = lio_s2m_batch_results(h, poses, results);
            h->no_persist = false;
            return rc;
        }
    }
    int64_t pit = 0;
    for (int s = 0; s < h->n_scans; ++s) {
        const LioScanState& st = h->h_state[s];
        if (poses) memcpy(poses + (size_t)s * 6, st.pose, sizeof(float) * 6);
        // a "< 50 correspondences" scan was fast-forwarded: it did its work once
        pit += (int64_t)(st.n_pts + st.c_n_pts) * (st.status == 2 ? 1 : st.iter) * (st.status == 1 ? 0 : 1);
        if (results) {
            lio_s2m_result& r = results[s];
            memset(&r, 0, sizeof(r));
            r.status = st.status;
            r.iters = st.status == 1 ? 0 : st.iter;
            r.converged = st.converged;
            r.is_degenerate = st.is_degenerate;
            r.n_corr_last = st.n_corr_last;
            memcpy(r.n_corr_iter, st.n_corr_iter, sizeof(r.n_corr_iter));
            memcpy(r.matP, st.matP, sizeof(r.matP));
            memcpy(r.AtA, st.AtA, sizeof(r.AtA));
            memcpy(r.AtB, st.AtB, sizeof(r.AtB));
            memcpy(r.pose_iter, st.pose_iter, sizeof(r.pose_iter));
        }
    }
    h->prof.point_iters = pit;
    h->prof.n_launches = h->launches_this_run;
    for (int i = 0; i < LIO_MAX_ITERS; ++i) { h->prof.launch_ms[i] = 0.0f; h->prof.launch_active[i] = 0; }
    // launch_ms[u] = device time of launch unit u (one launch, or one graph replay of unit_iters
    // launches); launch_active[u] = scans that took part in the unit's first launch
    const int n_units = h->units_this_run > 0 ? h->units_this_run : h->launches_this_run;
    for (int u = 0; u < n_units && u < LIO_MAX_ITERS; ++u) {
        int act = 0;
        for (int s = 0; s < h->n_scans; ++s) {
            const LioScanState& st = h->h_state[s];
            const int ran = st.status == 1 ? 0 : (st.status == 2 ? 1 : st.iter);
            if (u * h->unit_iters < ran) ++act;
        }
        h->prof.launch_active[u] = act;
        if (h->cfg.profile && h->units_this_run > 0) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, h->ev_beg[u], h->ev_end[u]) != hipSuccess) ms = -1.0f;
            h->prof.launch_ms[u] = ms;
        }
    }
    h->prof.n_units = n_units;
    h->prof.unit_iters = h->unit_iters;
    h->prof.pipeline = h->run_persist ? 4 : 1;
    h->prof.persist_fallbacks = h->persist_fallbacks;
    return LIO_OK;
}

extern "C" int lio_s2m_register(lio_s2m_handle* h, const void* scan, size_t n, size_t stride,
                                float pose[6], lio_s2m_result* res)
{
    if (!h || !pose) return lio_fail(LIO_ERR_ARG, "null argument");
    if (!lio_map_of(h)->has_map) return lio_fail(LIO_ERR_NO_MAP, "set_map has not been called");
    const void* scans[1] = { scan };
    size_t np[1] = { n };
    // upload, poses and the GN loop are chained on the stream; the only host wait is for the results
    // (the caller's buffers stay valid for the whole call)
    h->defer_sync = true;
    if (!h->multi) h->reg_pose = pose;                   // one host-to-device copy less: the guess rides with the state
    int rc = lio_s2m_batch_upload(h, 1, scans, np, stride);
    h->reg_pose = nullptr;
    if (rc == LIO_OK) {
        if (h->pose_in_state) h->poses_set = true;
        else rc = lio_s2m_batch_set_poses(h, pose);
    }
    if (rc == LIO_OK) rc = lio_s2m_batch_run(h);
    h->defer_sync = false;
    if (rc != LIO_OK) { (void)hipStreamSynchronize(h->stream); return rc; }
    lio_s2m_result local;
    if ((rc = lio_s2m_batch_results(h, pose, res ? res : &local)) != LIO_OK) return rc;
    return (res ? res : &local)->status;
}

// Field offsets of a PointCloud2 come off the wire: every comparison is written so that it cannot wrap (off + 12 > step
// passes for off = 0xfffffff4 in 32-bit arithmetic -- round-2 advisor finding).
int lio_pc2_check_xyz(const lio_pc2_layout* L)
{
    if (!L) return lio_fail(LIO_ERR_ARG, "null layout");
    const uint32_t st = L->point_step;
    if (st < 12 || (st & 3) || (L->off_x & 3) || L->off_x > st - 12)
        return lio_fail(LIO_ERR_ARG, "x, y, z must be three consecutive FLOAT32 fields inside the record");
    if (L->off_intensity >= 0 && ((L->off_intensity & 3) || (uint32_t)L->off_intensity > st - 4))
        return lio_fail(LIO_ERR_ARG, "intensity must be a FLOAT32 field inside the record");
    return LIO_OK;
}

// lio_s2m_register on a sensor_msgs/PointCloud2 `data` blob (cloud_info.cloud_deskewed, cloud_info.msg:27): what
// pcl::fromROSMsg(msgIn->cloud_deskewed, *laserCloudSurfLast) MO:440 + scan2MapOptimization do, without the copy
// into a pcl::PointCloud.  x, y, z are read in place at layout->off_x.
extern "C" int lio_s2m_register_pc2(lio_s2m_handle* h, const void* data, size_t n_points, const lio_pc2_layout* layout,
                                    float pose[6], lio_s2m_result* res)
{
    if (!h || !layout || !pose) return lio_fail(LIO_ERR_ARG, "null argument");
    if (lio_pc2_check_xyz(layout) != LIO_OK) return LIO_ERR_ARG;
    bool pinned = false;
    if (layout->pin_host && data && n_points) {
        HIPCHK(hipSetDevice(h->cfg.device_id));
        pinned = hipHostRegister(const_cast<void*>(data), n_points * layout->point_step, hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();
    }
    h->xyz_off = layout->off_x;
    h->int_off = layout->off_intensity >= 0 ? layout->off_intensity : -1;
    const int rc = lio_s2m_register(h, data, n_points, layout->point_step, pose, res);
    h->xyz_off = 0;
    h->int_off = -2;
    if (pinned) (void)hipHostUnregister(const_cast<void*>(data));
    return rc;
}

// Internal (lio_mapbuild.hip): the staged records of batch slot `scan` as they were uploaded.
int lio_s2m_staged_scan(lio_s2m_handle* h, int scan, const unsigned char** d_rec, size_t* n, size_t* stride, size_t* xyz_off, int* int_off,
                        int* device_id, hipStream_t* stream)
{
    if (!h || scan < 0 || scan >= h->n_scans || !h->last_stage) return lio_fail(LIO_ERR_ARG, "no such resident scan");
    const LioScanState& st = h->h_state[scan];
    *d_rec = h->last_stage + (size_t)st.offset * h->last_stride;
    *n = (size_t)st.n_pts; *stride = h->last_stride; *xyz_off = h->last_xyz_off; *int_off = h->last_int_off;
    *device_id = h->cfg.device_id; *stream = h->stream;
    return LIO_OK;
}

extern "C" int lio_s2m_register_cs(lio_s2m_handle* h, const void* corner_scan, size_t n_corner,
                                   const void* surf_scan, size_t n_surf, size_t stride,
                                   float pose[6], lio_s2m_result* res)
{
    if (!h || !pose) return lio_fail(LIO_ERR_ARG, "null argument");
    if (!h->has_map) return lio_fail(LIO_ERR_NO_MAP, "set_map has not been called");
    if (n_corner && (!h->corner || !h->corner->has_map)) return lio_fail(LIO_ERR_NO_MAP, "set_corner_map has not been called");
    const void* scans[1] = { surf_scan };
    const void* cscans[1] = { corner_scan };
    size_t np[1] = { n_surf }, ncp[1] = { n_corner };
    h->defer_sync = true;
    int rc = lio_s2m_batch_upload(h, 1, scans, np, stride);
    if (rc == LIO_OK && n_corner) rc = lio_s2m_batch_upload_corners(h, 1, cscans, ncp, stride);
    if (rc == LIO_OK) rc = lio_s2m_batch_set_poses(h, pose);
    if (rc == LIO_OK) rc = lio_s2m_batch_run(h);
    h->defer_sync = false;
    if (rc != LIO_OK) { (void)hipStreamSynchronize(h->stream); return rc; }
    lio_s2m_result local;
    if ((rc = lio_s2m_batch_results(h, pose, res ? res : &local)) != LIO_OK) return rc;
    return (res ? res : &local)->status;
}

extern "C" int lio_s2m_get_correspondences(lio_s2m_handle* h, int32_t scan, uint8_t* flag,
                                           float* coeff4, int32_t* nn_idx5)
{
    if (!h) return lio_fail(LIO_ERR_ARG, "null handle");
    if (h->cfg.record_corr_iter < 0) return lio_fail(LIO_ERR_ARG, "record_corr_iter was not set at create time");
    if (scan < 0 || scan >= h->n_scans) return lio_fail(LIO_ERR_ARG, "scan slot out of range");
    if (h->multi) return lio_multi_get_correspondences(h, scan, flag, coeff4, nn_idx5);
    HIPCHK(hipSetDevice(h->cfg.device_id));
    (void)hipGetLastError();   // drop stale codes left by other HIP users of this thread (e.g. hipErrorNotReady)
    { const int rcc = lio_run_continue(h, true); if (rcc != LIO_OK) return rcc; }   // (a run started by batch_run may still be pending)
    const size_t off = (size_t)h->h_state[scan].offset, n = (size_t)h->h_state[scan].n_pts;
    if (n == 0) return LIO_OK;
    if (flag) HIPCHK(hipMemcpyAsync(flag, h->d_rec_flag + off, n, hipMemcpyDeviceToHost, h->stream));
    if (coeff4) HIPCHK(hipMemcpyAsync(coeff4, h->d_rec_coeff + off * 4, n * 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (nn_idx5) HIPCHK(hipMemcpyAsync(nn_idx5, h->d_rec_nn + off * 5, n * 5 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LIO_OK;
}

extern "C" int lio_s2m_get_profile(lio_s2m_handle* h, lio_s2m_profile* out)
{
    if (!h || !out) return lio_fail(LIO_ERR_ARG, "null argument");
    if (h->multi) h->prof.n_map = (int64_t)h->n_map;
    if (h->map_timing_pending) {
        HIPCHK(hipSetDevice(h->cfg.device_id));
        HIPCHK(hipEventSynchronize(h->ev_mapl[1]));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev_mapl[0], h->ev_mapl[1]));
        h->prof.map_build_ms = ms;
        h->map_timing_pending = false;
    }
    *out = h->prof;
    return LIO_OK;
}

int lio_fail_ext(int code, const char* what, hipError_t e) { return lio_fail(code, what, e); }

// ------------------------------------------------------ host-side scalar code
// transformUpdate, MO:1867-1897: roll/pitch are slerp-blended towards the IMU
// attitude with tf (Bullet) quaternions in fp64, then clamped (MO:1892-1894,
// constraintTransformation MO:1899-1907).  tf::Quaternion::setRPY / slerp /
// tf::Matrix3x3::getRPY restated from their published formulas.
namespace {
struct Quat { double x, y, z, w; };

Quat quat_from_rpy(double roll, double pitch, double yaw)
{
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
    return { sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
             cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy };
}

double quat_dot(const Quat& a, const Quat& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

Quat quat_slerp(const Quat& a, const Quat& q, double t)
{
    const double s = sqrt(quat_dot(a, a) * quat_dot(q, q));
    const double d = quat_dot(a, q);
    const double theta = ((d < 0) ? acos(-d / s) * 2.0 : acos(d / s) * 2.0) / 2.0;   // angleShortestPath / 2
    if (theta != 0.0) {
        const double inv = 1.0 / sin(theta);
        const double s0 = sin((1.0 - t) * theta), s1 = sin(t * theta);
        if (d < 0)
            return { (a.x * s0 + -q.x * s1) * inv, (a.y * s0 + -q.y * s1) * inv,
                     (a.z * s0 + -q.z * s1) * inv, (a.w * s0 + -q.w * s1) * inv };
        return { (a.x * s0 + q.x * s1) * inv, (a.y * s0 + q.y * s1) * inv,
                 (a.z * s0 + q.z * s1) * inv, (a.w * s0 + q.w * s1) * inv };
    }
    return a;
}

void quat_to_rpy(const Quat& q, double& roll, double& pitch, double& yaw)
{
    const double s = 2.0 / quat_dot(q, q);
    const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    const double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    const double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (fabs(m20) >= 1) {
        yaw = 0;
        const double delta = atan2(m21, m22);
        pitch = (m20 < 0) ? M_PI / 2.0 : -M_PI / 2.0;
        roll = delta;
    } else {
        pitch = -asin(m20);
        const double cp = cos(pitch);
        roll = atan2(m21 / cp, m22 / cp);
        yaw = atan2(m10 / cp, m00 / cp);
    }
}

float clamp_abs(float v, float limit)
{
    if (v < -limit) v = -limit;
    if (v > limit) v = limit;
    return v;
}
}  // namespace

extern "C" void lio_transform_update(float pose[6], int32_t imu_available, int32_t imu_type,
                                     float imu_roll_init, float imu_pitch_init, float imu_rpy_weight,
                                     float rotation_tollerance, float z_tollerance)
{
    if (imu_available && imu_type) {                     // MO:1869
        if (fabsf(imu_pitch_init) < 1.4) {               // MO:1871
            const double w = imu_rpy_weight;
            double r, p, y;
            quat_to_rpy(quat_slerp(quat_from_rpy(pose[0], 0, 0), quat_from_rpy(imu_roll_init, 0, 0), w), r, p, y);
            pose[0] = (float)r;                          // MO:1879-1882
            quat_to_rpy(quat_slerp(quat_from_rpy(0, pose[1], 0), quat_from_rpy(0, imu_pitch_init, 0), w), r, p, y);
            pose[1] = (float)p;                          // MO:1885-1888
        }
    }
    pose[0] = clamp_abs(pose[0], rotation_tollerance);   // MO:1892
    pose[1] = clamp_abs(pose[1], rotation_tollerance);   // MO:1893
    pose[5] = clamp_abs(pose[5], z_tollerance);          // MO:1894
}

// imuDeskewInfo, IP:359-418: gyro integration over [timeScanCur-0.01, timeScanEnd+0.01]
extern "C" int lio_imu_deskew_info(const double* stamp, const double* gx, const double* gy, const double* gz,
                                   int32_t n_imu, double t_cur, double t_end,
                                   double* imuTime, double* imuRotX, double* imuRotY, double* imuRotZ)
{
    if (!stamp || !gx || !gy || !gz || !imuTime || !imuRotX || !imuRotY || !imuRotZ) return 0;
    int i = 0;
    while (i < n_imu && stamp[i] < t_cur - 0.01) ++i;    // IP:363-369
    if (i >= n_imu) return 0;                            // IP:371-372
    int cur = 0;
    for (; i < n_imu; ++i) {
        const double t = stamp[i];
        if (t > t_end + 0.01) break;                     // IP:387-388
        if (cur == 0) {                                  // IP:390-397
            imuRotX[0] = 0; imuRotY[0] = 0; imuRotZ[0] = 0; imuTime[0] = t;
            cur = 1;
            continue;
        }
        if (cur >= 2000) break;                          // table length, IP:62
        const double dt = t - imuTime[cur - 1];          // IP:404
        imuRotX[cur] = imuRotX[cur - 1] + gx[i] * dt;    // IP:405-407
        imuRotY[cur] = imuRotY[cur - 1] + gy[i] * dt;
        imuRotZ[cur] = imuRotZ[cur - 1] + gz[i] * dt;
        imuTime[cur] = t;
        ++cur;
    }
    return cur - 1;                                      // IP:412
}

// Diagnostic: phase clock of the LAST iterate launch (cfg.profile == 2):
// n_blocks x 4 waves x 8 cycle counters.  Returns the number of blocks.
extern "C" int lio_s2m_debug_stamps(lio_s2m_handle* h, long long* out, size_t cap_entries)
{
    if (!h || !h->d_stamps) return 0;
    const size_t n = (size_t)h->n_blocks * (LIO_BLOCK / 64) * 8;
    if (out && cap_entries >= n) {
        if (hipMemcpy(out, h->d_stamps, n * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    return h->n_blocks;
}

// ------------------------------------------------------------- temporary pool
namespace {
struct PoolBlock { void* p; size_t cap; int device; bool busy; };
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool;
size_t g_pool_bytes = 0;
const size_t kPoolLimit = (size_t)8 << 30;      // idle + busy bytes kept before idle blocks are dropped
}

hipError_t lio_pool_acquire(void** p, size_t bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (size_t i = 0; i < g_pool.size(); ++i) {
        const PoolBlock& b = g_pool[i];
        if (!b.busy && b.device == dev && b.cap >= bytes && b.cap <= 2 * bytes + 4096 &&
            (best < 0 || b.cap < g_pool[best].cap))
            best = (int)i;
    }
    if (best >= 0) { g_pool[best].busy = true; *p = g_pool[best].p; return hipSuccess; }
    if (g_pool_bytes + bytes > kPoolLimit) {            // drop idle blocks before growing further
        for (size_t i = 0; i < g_pool.size();) {
            if (!g_pool[i].busy) {
                (void)hipSetDevice(g_pool[i].device);
                (void)hipFree(g_pool[i].p);
                g_pool_bytes -= g_pool[i].cap;
                g_pool[i] = g_pool.back();
                g_pool.pop_back();
            } else {
                ++i;
            }
        }
        (void)hipSetDevice(dev);
    }
    const size_t cap = bytes + bytes / 4 + 256;
    e = hipMalloc(p, cap);
    if (e != hipSuccess) return e;
    g_pool.push_back({ *p, cap, dev, true });
    g_pool_bytes += cap;
    return hipSuccess;
}

void lio_pool_release(void* p)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (PoolBlock& b : g_pool)
        if (b.p == p) { b.busy = false; return; }
}

void lio_pool_trim(void)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (size_t i = 0; i < g_pool.size();) {
        if (!g_pool[i].busy) {
            (void)hipSetDevice(g_pool[i].device);
            (void)hipFree(g_pool[i].p);
            g_pool_bytes -= g_pool[i].cap;
            g_pool[i] = g_pool.back();
            g_pool.pop_back();
        } else {
            ++i;
        }
    }
    (void)hipSetDevice(dev);
}
