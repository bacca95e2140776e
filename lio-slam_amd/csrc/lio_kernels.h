// lio_kernels.h -- launch interface between the C-ABI implementation and the
// gfx950 kernels (lio_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "lio_types.h"

struct LioIterParams {
    LioGrid grid;
    LioShard shard;
    LioConsts c;
    const float4* map_sorted;      // cell-sorted (x,y,z,bits(orig index))
    const float4* map_xyz4;        // caller order (x,y,z,0) -- the 5 winners are re-read here
    const int* cell_start;         // n_cells + 1
    const float4* nbr_pts;         // 9x replicated neighbourhood rows (default candidate scan)
    const int* nbr_start;          // n_cells + 1 run offsets into nbr_pts
    const float* sx;               // batch scan SoA (all scans concatenated)
    const float* sy;
    const float* sz;
    const int* perm;               // sorted slot -> caller's point index (batch-global); may be null
    LioScanState* state;
    const LioBlockDesc* blocks;
    double* partials;              // [scan][max_blk][LIO_SUMS]
    unsigned* arrive;              // [scan] arrival counters
    int max_blk;
    int xcd_remap;                 // 1 = XCD-aware workgroup order
    int* n_active;                 // scans still iterating (device counter)
    double* sums_out;              // sharded mode: [scan][LIO_SUMS]; nullptr = solve in place
    unsigned char* rec_flag;       // optional correspondence record (iteration c.record_iter)
    float* rec_coeff;
    int* rec_nn;
    float* d5_cache;               // [total_pts] squared 5th-neighbour distance of the previous iteration (-1: none), or null
    long long* stamps;             // diagnostic phase clock: [block][wave][8] cycle counters, or null
    const unsigned char* blk_skip; // map sharding: [n_blocks] per-workgroup decision of k_shard_cull (0 per-point ownership, 1 answered
                                   // for by the cull, 2 wholly ours), or null
};

void lio_launch_aos_to_soa(const void* src, size_t stride, int n, float* x, float* y, float* z,
                           float4* xyz4, hipStream_t s);
void lio_launch_map_bbox(const float* x, const float* y, const float* z, int n, unsigned* bbox, hipStream_t s);
void lio_launch_map_build(const LioGrid& g, const float* x, const float* y, const float* z, int n,
                          int* cell_of, int* cell_count, int* cell_start, int* tile_sums,
                          float4* sorted, int* nbr_start, float4* nbr_pts, int* nbr_slot, bool with_cell_sorted, hipStream_t s);
void lio_launch_map_occupancy(const LioGrid& g, const float* x, const float* y, const float* z, int n, int* flags, hipStream_t s);
int  lio_scan_tiles(int n_cells);
void lio_launch_init_state(LioScanState* st, int n_scans, float* poses, bool from_state, const LioConsts& c,
                           int* n_active, hipStream_t s);
void lio_launch_iterate(const LioIterParams& P, int n_blocks, int ppt, bool stage, hipStream_t s, bool corner = false);
void lio_launch_persist(const LioIterParams& P, int n_blocks, unsigned* gen, unsigned epoch, const unsigned char* stage, size_t stride,
                        int n_scans, unsigned* spec, double* spec_sums, const float* poses0, unsigned spin_max, int withhold_wg, hipStream_t s);
void lio_launch_pack_summary(const LioScanState* st, int n_scans, float* out, hipStream_t s);
void lio_launch_apply(LioScanState* st, int n_scans, const double* sums, size_t slot_stride, int n_slots, const LioConsts& c,
                      int* n_active, hipStream_t s);
void lio_launch_shard_cull(const LioIterParams& P, int n_ranks, int rank, int halo, const int* bounds, const float* block_box,
                           int n_blocks, unsigned char* skip, hipStream_t s);
void lio_launch_block_boxes(const LioBlockDesc* blocks, int n_blocks, const LioScanState* st, const float* sx, const float* sy,
                            const float* sz, float* box, hipStream_t s);
void lio_launch_scan_sort_lds(const void* stage, size_t stride, const LioScanState* st, int n_scans, int max_pts, float tile0,
                              int shard_axis, int* perm, float* x, float* y, float* z, hipStream_t s);
void lio_launch_scan_bbox(const void* stage, size_t stride, const LioBlockDesc* prep_blocks, int n_prep_blocks,
                          const LioScanState* st, unsigned* bbox, hipStream_t s);
void lio_launch_scan_tile_sort(const void* stage, size_t stride, int total_pts,
                               const LioBlockDesc* prep_blocks, int n_prep_blocks,
                               const LioScanState* st, const LioScanTiles* tiles, int n_keys,
                               int* key_of, int* key_count, int* key_start, int* tile_sums,
                               int* tmp_idx, int* perm, int* big_list, int* big_cnt,
                               float* x, float* y, float* z, hipStream_t s);
void lio_launch_exclusive_scan(const int* in, int n, int* tile_sums, int* out, hipStream_t s);
void lio_launch_xyzi4_to_soa(const float4* src, int n, float* x, float* y, float* z, float4* xyz4, hipStream_t s);
