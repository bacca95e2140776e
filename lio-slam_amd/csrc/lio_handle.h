// lio_handle.h -- the opaque handle behind include/liogpu.h and the small helpers every translation unit of the
// C-ABI implementation uses (liogpu_api.hip, lio_multi.hip).  Host-side only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/liogpu.h"
#include "lio_kernels.h"
#include "lio_types.h"

int lio_fail(int code, const char* what, hipError_t e = hipSuccess);
int lio_pc2_check_xyz(const lio_pc2_layout* L);            // liogpu_api.hip
void lio_raw_ws_free(struct LioRawWs* ws);                 // lio_mapbuild.hip

#define HIPCHK(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) return lio_fail(LIO_ERR_HIP, #expr, _e);            \
    } while (0)

template <typename T>
static hipError_t lio_grow(T** p, size_t* cap, size_t need, double slack = 1.25)
{
    if (need <= *cap && *p) return hipSuccess;
    if (*p) { hipError_t e = hipFree(*p); if (e != hipSuccess) return e; *p = nullptr; }
    size_t n = (size_t)((double)need * slack) + 64;
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e == hipSuccess) *cap = n;
    else *cap = 0;
    return e;
}

struct lio_s2m_handle {
    lio_s2m_config cfg;
    LioConsts c;
    hipStream_t stream = nullptr;
    bool own_stream = true;

    // ---- resident local map (laserCloudSurfFromMapDS, MO:149) ----
    bool has_map = false;
    size_t n_map = 0;
    float *d_mx = nullptr, *d_my = nullptr, *d_mz = nullptr; size_t cap_mxyz[3] = {0, 0, 0};
    float4* d_map4 = nullptr;   size_t cap_map4 = 0;
    float4* d_sorted = nullptr; size_t cap_sorted = 0;
    int* d_cell_of = nullptr;   size_t cap_cell_of = 0;
    int* d_cell_count = nullptr; size_t cap_cell_count = 0;
    int* d_cell_start = nullptr; size_t cap_cell_start = 0;
    int* d_tile_sums = nullptr;  size_t cap_tile_sums = 0;
    int* d_nbr_start = nullptr;  size_t cap_nbr_start = 0;
    float4* d_nbr_pts = nullptr; size_t cap_nbr_pts = 0;
    int* d_nbr_slot = nullptr;   size_t cap_nbr_slot = 0;      // [n_map][(2k+1)^2] place of every replica inside its row-cell list
    unsigned* d_bbox = nullptr;
    unsigned char* d_stage = nullptr; size_t cap_stage = 0;          // the resident batch's records as uploaded
    unsigned char* d_map_stage = nullptr; size_t cap_map_stage = 0;  // lio_s2m_set_map's upload (its own buffer: the batch's staged
                                                                     // records stay valid -- the one-launch loop and
                                                                     // lio_kf_store_add_from_handle read them after a new map)
    LioGrid grid{};

    // ---- resident scan batch (laserCloudSurfLastDS, MO:138) ----
    int n_scans = 0;
    size_t total_pts = 0;
    float *d_sx = nullptr, *d_sy = nullptr, *d_sz = nullptr; size_t cap_sxyz[3] = {0, 0, 0};
    LioScanState* d_state = nullptr; size_t cap_state = 0;
    std::vector<LioScanState> h_state;
    std::vector<LioBlockDesc> v_blocks, v_prep;      // launch descriptors (kept alive for async H2D)
    std::vector<LioScanTiles> v_tiles;
    std::vector<LioBlockDesc> v_blocks_sorted;       // v_blocks re-ordered by scan position (sort_batch)
    std::vector<int> v_order, v_first, v_first_orig;
    bool defer_sync = false;                         // lio_s2m_register: one sync at the end of the call
    bool async_upload = false;                       // lio_s2m_batch_upload_async: wait for the H2D copy only
    float* d_poses = nullptr; size_t cap_poses = 0;
    const float* reg_pose = nullptr;  // lio_s2m_register: the initial guess travels inside the state upload
    bool pose_in_state = false;       // ... and is taken from there by the next k_s2m_init_state (which copies it to d_poses)
    float* d_summary = nullptr; size_t cap_summary = 0;   // [n_scans][10] compact results (lio_s2m_batch_results without `results`)
    float* h_summary = nullptr; size_t cap_h_summary = 0; // pinned
    bool host_state_stale = false;    // h_state misses device-side updates (matP ...) since a summary-only read
    LioBlockDesc* d_blocks = nullptr; size_t cap_blocks = 0;
    int n_blocks = 0, ppt = 1, max_blk = 1;
    double* d_partials = nullptr; size_t cap_partials = 0;
    unsigned* d_arrive = nullptr; size_t cap_arrive = 0;
    bool poses_set = false, ran = false;
    // upload-time tile sort of the scans
    LioScanTiles* d_tiles = nullptr; size_t cap_tiles = 0;
    LioBlockDesc* d_prep_blocks = nullptr; size_t cap_prep_blocks = 0;
    int* d_key_of = nullptr; size_t cap_key_of = 0;
    int* d_key_count = nullptr; size_t cap_key_count = 0;
    int* d_key_start = nullptr; size_t cap_key_start = 0;
    int* d_key_tiles = nullptr; size_t cap_key_tiles = 0;
    int* d_tmp_idx = nullptr; size_t cap_tmp_idx = 0;
    int* d_perm = nullptr; size_t cap_perm = 0;
    float* d_block_box = nullptr; size_t cap_block_box = 0;   // map sharding: per-workgroup bounding boxes (cull)
    bool has_block_box = false;
    unsigned char* d_blk_skip = nullptr; size_t cap_blk_skip = 0;
    int* d_big_list = nullptr; size_t cap_big_list = 0;   // tiles with more than LIO_TILE_CAP points (+ their count in the last slot)
    unsigned* d_scan_bbox = nullptr; size_t cap_scan_bbox = 0;   // [n_scans][6] ordered-uint bounding boxes
    unsigned* h_scan_bbox = nullptr; size_t cap_h_scan_bbox = 0; // pinned mirror
    bool sorted = false;
    const unsigned char* last_stage = nullptr;   // the records of the resident batch as uploaded (d_stage, or the caller's device buffer)
    size_t last_stride = 0, last_xyz_off = 0;
    size_t xyz_off = 0;                  // byte offset of x inside a record for the NEXT upload (lio_s2m_register_pc2)
    int int_off = -2;                    // byte offset of the FLOAT32 intensity for the NEXT upload: -1 = the records carry none,
                                         // -2 = PCL convention (byte 16 of a record of >= 20 bytes whose x sits at byte 0, else none)
    int last_int_off = -1;               // ... of the resident batch (lio_kf_store_add_from_handle)
    struct LioRawWs* raw_ws = nullptr;   // lio_s2m_register_raw: staged cloud + voxel-filter workspace (lio_mapbuild.hip)
    int persist_fallbacks = 0;           // one-launch loops that timed out and were re-run through the launch loop
    bool no_persist = false;             // the re-run itself: the launch loop, whatever cfg.pipeline says
    unsigned persist_spin_max = 0;       // polls before a waiting workgroup gives up (0 = default, LIO_PERSIST_SPIN_MAX in the environment)
    int persist_withhold = -1;           // test hook (lio_s2m_debug_persist_spin)
    lio_s2m_handle* map_src = nullptr;   // lio_s2m_share_map: the handle whose resident map this one searches
    unsigned long long map_epoch = 0;    // bumped by every set_map
    float* d_nn_cache = nullptr; size_t cap_nn_cache = 0;   // [total_pts] squared 5th-neighbour distance of the previous GN iteration
    long long* d_stamps = nullptr; size_t cap_stamps = 0;
    // one-launch loop (cfg.pipeline = 4, k_s2m_persist): per-scan generation numbers
    unsigned* d_gen = nullptr; size_t cap_gen = 0;     // [2][cap_gen / 2]: generation numbers, then the speculation states
    double* d_spec_sums = nullptr; size_t cap_spec_sums = 0;   // sums of the first solve of every scan (roll-back of the speculation)
    unsigned gen_epoch = 0;           // grows by 128 per run: generation numbers are never cleared
    int n_cu = 0;                     // compute units of the device: every workgroup of a one-launch loop must be resident
    bool run_persist = false;
    bool soa_valid = true;            // d_sx/d_sy/d_sz hold the resident batch (false: a one-launch batch still only staged as records)
    // hipGraph-captured chunk of GN iterations (cfg.use_graph)
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool graph_dirty = true;
    int graph_chunk = 0, graph_blocks = 0, graph_ppt = 0;
    LioIterParams graph_params;       // arguments the cached graph was captured with
    int units_this_run = 0, unit_iters = 1;

    // correspondence record (debug / parity)
    unsigned char* d_rec_flag = nullptr; size_t cap_rec_flag = 0;
    float* d_rec_coeff = nullptr; size_t cap_rec_coeff = 0;
    int* d_rec_nn = nullptr; size_t cap_rec_nn = 0;

    // in-library multi-GPU mode (cfg.n_devices > 1): this handle is only a front; see struct LioMulti
    struct LioMulti* multi = nullptr;

    // resumable launch loop (lio_s2m_batch_run / lio_run_continue)
    bool run_pending = false, run_graph = false, run_has_c = false;
    int run_next = 0, run_units = 0, run_look = 0;
    LioIterParams run_P, run_Pc;

    // sharding
    LioShard shard{};
    float gorigin[3] = {0, 0, 0};
    int gdims[3] = {0, 0, 0};
    bool has_global = false;
    int block_rank = 0, block_world = 1;   // scan-range sharding
    int plan_ranks = 0, plan_rank = 0, plan_halo = 1, plan_bounds[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // lio_s2m_set_shard_plan

    // EXTENSION: point-to-line residuals.  `corner` is a child handle that owns the corner
    // map (its grid and neighbourhood rows) and the batch of edge points; its association launch writes
    // into THIS handle's per-scan partial sums and state.
    lio_s2m_handle* corner = nullptr;
    bool corner_active = false;       // a corner batch matching the current surf batch is resident
    LioIterParams graph_params_c;
    int graph_blocks_c = 0;

    // profiling
    hipEvent_t ev_beg[LIO_MAX_ITERS] = {}, ev_end[LIO_MAX_ITERS] = {}, ev_chk[LIO_MAX_ITERS] = {};
    hipEvent_t ev_map[2] = {};
    hipEvent_t ev_mapl[2] = {};        // asynchronous map installation (lio_s2m_set_map_device_bbox): build time resolved on demand; [1] = "map ready"
    bool map_timing_pending = false;
    int* h_active = nullptr;          // pinned: active-scan count after each launch
    bool ev_ok = false;
    lio_s2m_profile prof{};
    int launches_this_run = 0;
    int* d_active = nullptr;
};
