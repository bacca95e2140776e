// lio_pool.h -- recycling device-memory pool for the stateless entry points (lio_deskew,
// lio_curvature, lio_voxel_grid, lio_assemble_map*).  hipMalloc/hipFree cost milliseconds each;
// the temporaries of those calls are taken from and returned to this pool instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

hipError_t lio_pool_acquire(void** p, size_t bytes);   // on the current device
void lio_pool_release(void* p);
void lio_pool_trim(void);                               // frees every idle block

struct LioTemp {            // RAII temporary
    void* p = nullptr;
    ~LioTemp() { if (p) lio_pool_release(p); }
    hipError_t alloc(size_t bytes) { return lio_pool_acquire(&p, bytes ? bytes : 16); }
    template <typename T> T* as() { return (T*)p; }
};
