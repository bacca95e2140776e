// lio_multi.h -- in-library multi-GPU mode (lio_multi.hip), called by the C-ABI entry points of liogpu_api.hip for
// handles created with cfg.n_devices > 1.
#pragma once
#include "lio_handle.h"

int  lio_multi_create(const lio_s2m_config* cfg, lio_s2m_handle** out);
void lio_multi_destroy(lio_s2m_handle* h);
int  lio_multi_set_map(lio_s2m_handle* h, const void* pts, size_t n, size_t stride);
int  lio_multi_upload(lio_s2m_handle* h, int32_t n_scans, const void* const* scans, const size_t* n_pts, size_t stride);
int  lio_multi_set_poses(lio_s2m_handle* h, const float* poses);
int  lio_multi_set_degeneracy(lio_s2m_handle* h, int32_t scan, const float matP[36], int32_t is_degenerate);
int  lio_multi_run(lio_s2m_handle* h);
int  lio_multi_sync(lio_s2m_handle* h);
int  lio_multi_results(lio_s2m_handle* h, float* poses, lio_s2m_result* results);
int  lio_multi_get_correspondences(lio_s2m_handle* h, int32_t scan, uint8_t* flag, float* coeff4, int32_t* nn_idx5);

// liogpu_api.hip: lio_s2m_batch_iter_apply for sums that arrive as `n_slots` partial tables of `slot_stride` doubles each
// (one per device, added in slot order inside the solving kernel).
int  lio_s2m_iter_apply_slots(lio_s2m_handle* h, const double* d_slots, size_t slot_stride, int n_slots);
