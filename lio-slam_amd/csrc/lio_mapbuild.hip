// lio_mapbuild.hip -- the feeders of the registration path:
//   K6 transformPointCloud           MO:849-868
//   K7 pcl::VoxelGrid centroid filter MO:1605-1611 (scan), MO:1581-1583 (local map)
//   extractCloud = sum of K6 over the nearby keyframes, then K7   MO:1556-1588
// MO = /root/reference/src/liorf/src/mapOptmization.cpp.  -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <string.h>
#include <vector>

#include "lio_handle.h"
#include "lio_pool.h"
#include "lio_device_math.h"
#include "lio_scan2.h"
#include "lio_voxsort.h"

int lio_fail_ext(int code, const char* what, hipError_t e);                    // liogpu_api.hip
int lio_s2m_set_map_device_xyzi(lio_s2m_handle* h, const float4* d_xyzi, size_t n);   // liogpu_api.hip
int lio_s2m_set_map_device_bbox(lio_s2m_handle* h, const float4* d_xyzi, size_t n, const float box[6]);
hipStream_t lio_s2m_stream_of(lio_s2m_handle* h);
bool lio_s2m_takes_device_map(const lio_s2m_handle* h);

struct LioKfDesc {       // one selected keyframe
    int src;             // first point of the keyframe in the resident store
    int first;           // first point in the concatenated world-frame cloud
    int n;
    int pad;
    float T[12];         // pclPointToAffine3f of its pose (MO:856), filled on the device
};

// pose [roll,pitch,yaw,x,y,z] -> 3x4 transform, same trig definition as the GN loop
__global__ void k_kf_transforms(LioKfDesc* __restrict__ kf, const float* __restrict__ poses, int n_kf)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_kf) return;
    float pose[6], T[12], trig[6];
    for (int j = 0; j < 6; ++j) pose[j] = poses[k * 6 + j];
    lio_pose_to_transform(pose, T, trig);
    for (int j = 0; j < 12; ++j) kf[k].T[j] = T[j];
}

// K6: resident keyframe clouds (float4 x,y,z,intensity, lidar frame) -> world-frame float4
__global__ __launch_bounds__(256) void k_transform_clouds(const float4* __restrict__ store,
                                                          const LioKfDesc* __restrict__ kf,
                                                          const int2* __restrict__ chunks /* (kf, first) */,
                                                          float4* __restrict__ dst)
{
    const int2 c = chunks[blockIdx.x];
    const LioKfDesc d = kf[c.x];
    const int li = c.y + (int)threadIdx.x;
    if (li >= d.n) return;
    const float4 p = store[d.src + li];
    dst[d.first + li] = make_float4(d.T[0] * p.x + d.T[1] * p.y + d.T[2]  * p.z + d.T[3],
                                    d.T[4] * p.x + d.T[5] * p.y + d.T[6]  * p.z + d.T[7],
                                    d.T[8] * p.x + d.T[9] * p.y + d.T[10] * p.z + d.T[11], p.w);   // MO:861-864
}

__global__ void k_aos_to_xyzi4(const unsigned char* __restrict__ src, size_t stride, int n, float4* __restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(src + (size_t)i * stride);
    dst[i] = make_float4(p[0], p[1], p[2], p[4]);
}

__device__ __forceinline__ unsigned lio_f2ord2(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// getMinMax3D (PCL): bbox[0..2] = min, bbox[3..5] = max as order-preserving uints
__global__ void k_vox_bbox(const float4* __restrict__ p, int n, unsigned* __restrict__ bbox)
{
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 v = p[i];
        mn[0] = fminf(mn[0], v.x); mx[0] = fmaxf(mx[0], v.x);
        mn[1] = fminf(mn[1], v.y); mx[1] = fmaxf(mx[1], v.y);
        mn[2] = fminf(mn[2], v.z); mx[2] = fmaxf(mx[2], v.z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    __shared__ float s_mn[4][3], s_mx[4][3];          // one set of atomics per workgroup
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = s_mn[0][a], hi = s_mx[0][a];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fminf(lo, s_mn[w][a]); hi = fmaxf(hi, s_mx[w][a]); }
        atomicMin(&bbox[a], lio_f2ord2(lo));
        atomicMax(&bbox[3 + a], lio_f2ord2(hi));
    }
}

// K6 with getMinMax3D folded in: the bounding box of the transformed cloud (what the voxel filter starts with) is
// accumulated while the points are written, one set of atomics per workgroup -- no separate pass over the 1.3 M points.
__global__ __launch_bounds__(256) void k_transform_clouds_bbox(const float4* __restrict__ store, const LioKfDesc* __restrict__ kf,
                                                               const int2* __restrict__ chunks /* (kf, first) */,
                                                               float4* __restrict__ dst, float* __restrict__ blk_box /* [grid][6] */)
{
    const int2 c = chunks[blockIdx.x];
    const LioKfDesc d = kf[c.x];
    const int li = c.y + (int)threadIdx.x;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (li < d.n) {
        const float4 p = store[d.src + li];
        const float4 q = make_float4(d.T[0] * p.x + d.T[1] * p.y + d.T[2]  * p.z + d.T[3],
                                     d.T[4] * p.x + d.T[5] * p.y + d.T[6]  * p.z + d.T[7],
                                     d.T[8] * p.x + d.T[9] * p.y + d.T[10] * p.z + d.T[11], p.w);   // MO:861-864
        dst[d.first + li] = q;
        mn[0] = mx[0] = q.x; mn[1] = mx[1] = q.y; mn[2] = mx[2] = q.z;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = s_mn[0][a], hi = s_mx[0][a];
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, s_mn[w][a]); hi = fmaxf(hi, s_mx[w][a]); }
        blk_box[(size_t)blockIdx.x * 6 + a] = lo;            // (+inf / -inf for an empty chunk)
        blk_box[(size_t)blockIdx.x * 6 + 3 + a] = hi;
    }
}

// the per-workgroup boxes of k_transform_clouds_bbox -> one box, in k_vox_bbox's order-preserving encoding (one workgroup;
// ~5 000 workgroups hammering six words with atomics cost 124 us, this costs 3)
__global__ __launch_bounds__(256) void k_bbox_reduce(const float* __restrict__ blk_box, int n_blk, unsigned* __restrict__ bbox)
{
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = threadIdx.x; i < n_blk; i += 256)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], blk_box[(size_t)i * 6 + a]); mx[a] = fmaxf(mx[a], blk_box[(size_t)i * 6 + 3 + a]); }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = s_mn[0][a], hi = s_mx[0][a];
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, s_mn[w][a]); hi = fmaxf(hi, s_mx[w][a]); }
        bbox[a] = lo <= hi ? lio_f2ord2(lo) : 0xffffffffu;
        bbox[3 + a] = lo <= hi ? lio_f2ord2(hi) : 0u;
    }
}

struct LioVoxGrid { float inv; int min_b0, min_b1, min_b2, mul1, mul2, n_keys; };

__global__ void k_xyzi4_to_aos(const float4* __restrict__ src, int n, unsigned char* __restrict__ dst, size_t stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = src[i];
    float* o = reinterpret_cast<float*>(dst + (size_t)i * stride);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = 1.0f; o[4] = v.w;
}

// Grow-only device buffer owned by a keyframe store: the workspace of lio_assemble_map_resident, kept from one call to
// the next so that nothing has to be waited for before the call returns (a pool temporary is recycled on return).
struct LioKeep {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t alloc(size_t bytes)
    {
        if (!bytes) bytes = 16;
        if (p && bytes <= cap) return hipSuccess;
        if (p) {                                          // growing: whoever still reads the old block must be done
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) return e;
            e = hipFree(p);
            p = nullptr; cap = 0;
            if (e != hipSuccess) return e;
        }
        const size_t want = bytes + bytes / 4 + 256;
        const hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() { return (T*)p; }
};

namespace {
typedef LioTemp Buf;         // temporaries come from the recycling pool (lio_pool.h)

template <class B> struct LioVoxWs { B bbox, large, pairs_a, pairs_b, hist, blk_heads, seg_start, d_no, row_total; };

float ord2f(unsigned u)
{
    const unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

// K7 proper (lio_voxsort.h): keys -> stable LSD radix sort of (key, index) pairs -> segment heads -> in-order sums.  g describes
// the voxel grid, n_keys its size (< 2^31).  (Rounds 1-2 used a counting sort with count / start / rank arrays over the whole
// key space, atomics for the slots and a per-voxel sort for the order; round 3 first replaced the per-voxel sorts, then measured
// the sorting form faster on every input -- map assembly 0.37 against 0.43 ms, the raw-sweep chain 0.50 against 0.57 ms at leaf
// 0.4 m and 0.88 against 1.09 ms at 0.15 m, profiles/r03_k7_forms.txt -- and removed the counting form.)  `out` is allocated for the worst case (n voxels) so that the centroid kernels are
// enqueued without waiting for the count; the one host wait (*n_out) comes last and overlaps them.
template <class B>
static int voxel_grid_sorted(const float4* d_in, int n, const LioVoxGrid& g, long long n_keys, B& out, int* n_out, hipStream_t s, LioVoxWs<B>& ws)
{
    int bits = 1;
    while (bits < 31 && (1LL << bits) < n_keys) ++bits;
    const int passes = (bits + 7) / 8, dbits = (bits + passes - 1) / passes;      // e.g. 25 bits -> 4 passes of 7
    const unsigned mask = (1u << dbits) - 1u;
    const int items = n > (1 << 18) ? 8 : 4;
    const int tile = LIO_VS_THREADS * items, n_blocks = (n + tile - 1) / tile;
    const int n_hblk = (n + 1023) / 1024;
    HIPCHK(ws.pairs_a.alloc(sizeof(uint2) * (size_t)n));
    HIPCHK(ws.pairs_b.alloc(sizeof(uint2) * (size_t)n));
    HIPCHK(ws.hist.alloc(sizeof(int) * (size_t)LIO_VS_BINS * n_blocks));
    HIPCHK(ws.blk_heads.alloc(sizeof(int) * (size_t)(n_hblk + 1)));
    HIPCHK(ws.seg_start.alloc(sizeof(int) * ((size_t)n + 1)));
    HIPCHK(ws.d_no.alloc(sizeof(int) * 2));
    HIPCHK(ws.row_total.alloc(sizeof(int) * LIO_VS_BINS));
    HIPCHK(ws.large.alloc(sizeof(int) * ((size_t)n + 1)));
    HIPCHK(out.alloc(sizeof(float4) * (size_t)n));
    LioVsGrid vg = { g.inv, g.min_b0, g.min_b1, g.min_b2, g.mul1, g.mul2 };
    uint2 *a = ws.pairs_a.template as<uint2>(), *b = ws.pairs_b.template as<uint2>();
    hipLaunchKernelGGL(k_vsort_keys, dim3((n + 255) / 256), dim3(256), 0, s, vg, d_in, n, a);
    for (int p = 0; p < passes; ++p) {
        const int shift = p * dbits;
        int* hist = ws.hist.template as<int>();
        if (items == 8) hipLaunchKernelGGL(k_vsort_hist<8>, dim3(n_blocks), dim3(LIO_VS_THREADS), 0, s, a, n, shift, mask, hist, n_blocks);
        else hipLaunchKernelGGL(k_vsort_hist<4>, dim3(n_blocks), dim3(LIO_VS_THREADS), 0, s, a, n, shift, mask, hist, n_blocks);
        int* row_total = ws.row_total.template as<int>();
        hipLaunchKernelGGL(k_vsort_scan_rows, dim3(LIO_VS_BINS), dim3(256), 0, s, hist, n_blocks, row_total);
        if (items == 8) hipLaunchKernelGGL(k_vsort_scatter<8>, dim3(n_blocks), dim3(LIO_VS_THREADS), 0, s, a, n, shift, mask, hist, row_total, n_blocks, b);
        else hipLaunchKernelGGL(k_vsort_scatter<4>, dim3(n_blocks), dim3(LIO_VS_THREADS), 0, s, a, n, shift, mask, hist, row_total, n_blocks, b);
        uint2* t = a; a = b; b = t;
    }
    int* d_no = ws.d_no.template as<int>();
    HIPCHK(hipMemsetAsync(d_no, 0, 2 * sizeof(int), s));                          // [0] voxels, [1] crowded voxels queued
    hipLaunchKernelGGL(k_vsort_head_count, dim3(n_hblk), dim3(256), 0, s, a, n, ws.blk_heads.template as<int>());
    hipLaunchKernelGGL(k_vsort_scan_small, dim3(1), dim3(256), 0, s, ws.blk_heads.template as<int>(), n_hblk, d_no);
    hipLaunchKernelGGL(k_vsort_head_emit, dim3(n_hblk), dim3(256), 0, s, a, n, ws.blk_heads.template as<int>(), d_no, ws.seg_start.template as<int>());
    int no = 0;
    HIPCHK(hipMemcpyAsync(&no, d_no, sizeof(int), hipMemcpyDeviceToHost, s));
    hipLaunchKernelGGL(k_vsort_centroid, dim3((n_hblk + 3) / 4), dim3(256), 0, s, d_in, a, n, ws.seg_start.template as<int>(),
                       ws.blk_heads.template as<int>(), n_hblk, d_no, out.template as<float4>(), ws.large.template as<int>(), d_no + 1);
    hipLaunchKernelGGL(k_vsort_centroid_large, dim3(n < 1024 * 64 ? (n + 63) / 64 : 1024), dim3(256), 0, s, d_in, a, ws.seg_start.template as<int>(),
                       out.template as<float4>(), ws.large.template as<int>(), d_no + 1);
    HIPCHK(hipStreamSynchronize(s));                                              // (`no`; the centroid kernels ran under this wait)
    HIPCHK(hipGetLastError());
    *n_out = no;
    return LIO_OK;
}

// K7 on a device-resident float4 cloud.  *d_out receives a freshly allocated device array.
// Returns LIO_OK, or 1 when PCL would pass the cloud through (voxel index overflow).
// `ws`: the temporaries; `wait`: block until the result is complete (required when ws is made of pool temporaries, which
// are recycled when the caller returns); `box` (optional) receives min[3], max[3] of the INPUT cloud, a box around the output.
// have_box: ws.bbox already holds the bounding box of d_in (k_transform_clouds_bbox), no pass for it.
template <class B>
int voxel_grid_device(const float4* d_in, int n, float leaf, B& out, int* n_out, hipStream_t s, LioVoxWs<B>& ws, bool wait, float* box,
                      bool have_box = false)
{
    *n_out = 0;
    if (box) for (int a = 0; a < 6; ++a) box[a] = 0.0f;
    if (n == 0) return LIO_OK;
    B& bbox = ws.bbox;
    if (!have_box) {
        HIPCHK(bbox.alloc(6 * sizeof(unsigned)));
        const unsigned init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
        HIPCHK(hipMemcpyAsync(bbox.p, init, sizeof(init), hipMemcpyHostToDevice, s));
        int nbb = (n + 1023) / 1024; if (nbb > 512) nbb = 512; if (nbb < 1) nbb = 1;
        hipLaunchKernelGGL(k_vox_bbox, dim3(nbb), dim3(256), 0, s, d_in, n, bbox.template as<unsigned>());
    }
    unsigned hb[6];
    HIPCHK(hipMemcpyAsync(hb, bbox.p, sizeof(hb), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a] = ord2f(hb[a]); mx[a] = ord2f(hb[3 + a]); }
    if (box) for (int a = 0; a < 3; ++a) { box[a] = mn[a]; box[3 + a] = mx[a]; }
    const float inv = 1.0f / leaf;
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1,
                    dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    // (a box that is not finite -- inf coordinates, or no finite point at all -- is outside what PCL defines: the conversions
    // above are then meaningless; such a cloud takes the same way out as an overflowing index, deterministically)
    bool finite_box = true;
    for (int a = 0; a < 3; ++a) finite_box = finite_box && (mn[a] <= mx[a]) && fabsf(mn[a]) <= 3.0e38f && fabsf(mx[a]) <= 3.0e38f;
    if (!finite_box || dx <= 0 || dy <= 0 || dz <= 0 || (double)dx * (double)dy * (double)dz > 2147483647.0) {   // "Leaf size is too small": PCL copies the input
        HIPCHK(out.alloc(sizeof(float4) * (size_t)n));
        HIPCHK(hipMemcpyAsync(out.p, d_in, sizeof(float4) * (size_t)n, hipMemcpyDeviceToDevice, s));
        *n_out = n;
        return 1;
    }
    LioVoxGrid g;
    g.inv = inv;
    g.min_b0 = (int)floorf(mn[0] * inv); g.min_b1 = (int)floorf(mn[1] * inv); g.min_b2 = (int)floorf(mn[2] * inv);
    const int d0 = (int)floorf(mx[0] * inv) - g.min_b0 + 1, d1 = (int)floorf(mx[1] * inv) - g.min_b1 + 1,
              d2 = (int)floorf(mx[2] * inv) - g.min_b2 + 1;
    g.mul1 = d0; g.mul2 = d0 * d1;
    const long long n_keys_ll = (long long)d0 * d1 * d2;
    if (n_keys_ll > 2147483647LL) return lio_fail_ext(LIO_ERR_CAPACITY, "voxel grid has more than 2^31 - 1 voxels", hipSuccess);
    g.n_keys = 0;
    (void)wait;                                  // (the sorting form always ends with the wait for the voxel count)
    return voxel_grid_sorted<B>(d_in, n, g, n_keys_ll, out, n_out, s, ws);
}

int voxel_grid_device(const float4* d_in, int n, float leaf, Buf& out, int* n_out, hipStream_t s)
{
    LioVoxWs<Buf> ws;
    return voxel_grid_device<Buf>(d_in, n, leaf, out, n_out, s, ws, true, nullptr);
}

int copy_out(const float4* d_pts, int n, void* out, size_t out_stride, hipStream_t s)
{
    if (!out || n == 0) return LIO_OK;
    Buf aos;
    HIPCHK(aos.alloc((size_t)n * out_stride));
    HIPCHK(hipMemsetAsync(aos.p, 0, (size_t)n * out_stride, s));
    hipLaunchKernelGGL(k_xyzi4_to_aos, dim3((n + 255) / 256), dim3(256), 0, s, d_pts, n, aos.as<unsigned char>(), out_stride);
    HIPCHK(hipMemcpyAsync(out, aos.p, (size_t)n * out_stride, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return LIO_OK;
}

int check_device(int device_id)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lio_fail_ext(LIO_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)", hipSuccess);
    HIPCHK(hipSetDevice(device_id));
    (void)hipGetLastError();
    return LIO_OK;
}
}  // namespace

extern "C" int lio_voxel_grid(int32_t device_id, const void* pts, size_t n, size_t stride, float leaf,
                              void* out, size_t out_stride, size_t* n_out)
{
    if (!n_out || (n && (!pts || !out))) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if (stride < 20 || (stride & 3) || out_stride < 20 || (out_stride & 3) || !(leaf > 0.0f))
        return lio_fail_ext(LIO_ERR_ARG, "strides must be >= 20 and multiples of 4, leaf > 0", hipSuccess);
    if (n > 0x7fffffffull - 1024) return lio_fail_ext(LIO_ERR_CAPACITY, "cloud too large", hipSuccess);
    *n_out = 0;
    if (n == 0) return LIO_OK;
    int rc = check_device(device_id);
    if (rc != LIO_OK) return rc;
    hipStream_t s = nullptr;
    Buf raw, xyzi, ds;
    HIPCHK(raw.alloc(n * stride));
    HIPCHK(xyzi.alloc(n * sizeof(float4)));
    HIPCHK(hipMemcpyAsync(raw.p, pts, n * stride, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_aos_to_xyzi4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw.as<unsigned char>(), stride, (int)n, xyzi.as<float4>());
    int no = 0;
    rc = voxel_grid_device(xyzi.as<float4>(), (int)n, leaf, ds, &no, s);
    if (rc < 0) return rc;
    const int rc2 = copy_out(ds.as<float4>(), no, out, out_stride, s);
    if (rc2 < 0) return rc2;
    *n_out = (size_t)no;
    return rc;
}

// ------------------------------------------------------- resident keyframe store
// surfCloudKeyFrames (MO:128): every keyframe cloud is uploaded ONCE (MO:2138-2142) and stays in
// HBM; assembling the local map for a scan only needs the selected ids and their current poses.
struct lio_kf_store {
    int device_id = 0;
    float4* d_pts = nullptr;
    size_t cap = 0, used = 0;
    std::vector<size_t> off, cnt;
    // workspace of lio_assemble_map_resident when the map is installed in a handle (kept between calls)
    LioVoxWs<LioKeep> vws;
    LioKeep world, ds, d_kf, d_poses, d_chunks, blk_box;
    std::vector<LioKfDesc> v_kf;
    std::vector<int2> v_chunks;
};

extern "C" int lio_kf_store_create(int32_t device_id, lio_kf_store** out)
{
    if (!out) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    int rc = check_device(device_id);
    if (rc != LIO_OK) return rc;
    lio_kf_store* s = new lio_kf_store();
    s->device_id = device_id;
    *out = s;
    return LIO_OK;
}

extern "C" void lio_kf_store_destroy(lio_kf_store* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device_id);
    (void)hipDeviceSynchronize();
    if (s->d_pts) (void)hipFree(s->d_pts);
    LioKeep* keep[] = { &s->vws.bbox, &s->vws.large, &s->vws.pairs_a, &s->vws.pairs_b, &s->vws.hist, &s->vws.blk_heads, &s->vws.seg_start, &s->vws.d_no, &s->vws.row_total,
                        &s->world, &s->ds, &s->d_kf, &s->d_poses, &s->d_chunks, &s->blk_box };
    for (LioKeep* k : keep) k->release();
    delete s;
}

extern "C" int lio_kf_store_count(const lio_kf_store* s) { return s ? (int)s->off.size() : 0; }
extern "C" size_t lio_kf_store_points(const lio_kf_store* s, int32_t id) { return (s && id >= 0 && (size_t)id < s->cnt.size()) ? s->cnt[(size_t)id] : 0; }

static int kf_store_reserve(lio_kf_store* s, size_t n)
{
    if (s->used + n > 0x7fffffffull - 1024) return lio_fail_ext(LIO_ERR_CAPACITY, "keyframe store is full", hipSuccess);
    if (s->used + n > s->cap) {                         // grow geometrically, keep the resident clouds
        size_t ncap = (s->cap ? s->cap * 2 : (size_t)1 << 20);
        while (ncap < s->used + n) ncap *= 2;
        float4* np_ = nullptr;
        HIPCHK(hipMalloc((void**)&np_, ncap * sizeof(float4)));
        if (s->used) HIPCHK(hipMemcpy(np_, s->d_pts, s->used * sizeof(float4), hipMemcpyDeviceToDevice));
        if (s->d_pts) HIPCHK(hipFree(s->d_pts));
        s->d_pts = np_;
        s->cap = ncap;
    }
    return LIO_OK;
}

static void kf_store_commit(lio_kf_store* s, size_t n, int32_t* id_out)
{
    if (id_out) *id_out = (int32_t)s->off.size();
    s->off.push_back(s->used);
    s->cnt.push_back(n);
    s->used += n;
}

// records (x,y,z at xyz_off, FLOAT32 intensity at int_off, < 0 = the record carries none) -> float4 (x,y,z,intensity)
__global__ void k_rec_to_xyzi4(const unsigned char* __restrict__ src, size_t stride, size_t xyz_off, int int_off, int n,
                               float4* __restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char* rec = src + (size_t)i * stride;
    const float* p = reinterpret_cast<const float*>(rec + xyz_off);
    dst[i] = make_float4(p[0], p[1], p[2], int_off >= 0 ? *reinterpret_cast<const float*>(rec + int_off) : 0.0f);
}

extern "C" int lio_kf_store_add(lio_kf_store* s, const void* cloud, size_t n, size_t stride, int32_t* id_out)
{
    if (!s || (n && !cloud)) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if (stride < 20 || (stride & 3)) return lio_fail_ext(LIO_ERR_ARG, "stride must be >= 20 and a multiple of 4", hipSuccess);
    int rc = check_device(s->device_id);
    if (rc != LIO_OK) return rc;
    if ((rc = kf_store_reserve(s, n)) != LIO_OK) return rc;
    if (n) {
        Buf raw;
        HIPCHK(raw.alloc(n * stride));
        HIPCHK(hipMemcpyAsync(raw.p, cloud, n * stride, hipMemcpyDefault, nullptr));
        hipLaunchKernelGGL(k_aos_to_xyzi4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr,
                           raw.as<unsigned char>(), stride, (int)n, s->d_pts + s->used);
        HIPCHK(hipStreamSynchronize(nullptr));
        HIPCHK(hipGetLastError());
    }
    kf_store_commit(s, n, id_out);
    return LIO_OK;
}

extern "C" int lio_kf_store_add_device(lio_kf_store* s, const void* d_cloud, size_t n, size_t stride, int32_t* id_out)
{
    if (!s || (n && !d_cloud)) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if (stride < 12 || (stride & 3)) return lio_fail_ext(LIO_ERR_ARG, "stride must be >= 12 and a multiple of 4", hipSuccess);
    int rc = check_device(s->device_id);
    if (rc != LIO_OK) return rc;
    if ((rc = kf_store_reserve(s, n)) != LIO_OK) return rc;
    if (n) {
        HIPCHK(hipDeviceSynchronize());                  // the producer of d_cloud may have used any stream
        hipLaunchKernelGGL(k_rec_to_xyzi4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr,
                           (const unsigned char*)d_cloud, stride, (size_t)0, stride >= 20 ? 16 : -1, (int)n, s->d_pts + s->used);
        HIPCHK(hipStreamSynchronize(nullptr));
        HIPCHK(hipGetLastError());
    }
    kf_store_commit(s, n, id_out);
    return LIO_OK;
}

int lio_s2m_staged_scan(lio_s2m_handle* h, int scan, const unsigned char** d_rec, size_t* n, size_t* stride, size_t* xyz_off, int* int_off,
                        int* device_id, hipStream_t* stream);   // liogpu_api.hip

extern "C" int lio_kf_store_add_from_handle(lio_kf_store* s, lio_s2m_handle* h, int32_t scan, int32_t* id_out)
{
    if (!s || !h) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    const unsigned char* rec = nullptr;
    size_t n = 0, stride = 0, xyz_off = 0;
    int dev = 0, int_off = -1;
    hipStream_t st = nullptr;
    int rc = lio_s2m_staged_scan(h, scan, &rec, &n, &stride, &xyz_off, &int_off, &dev, &st);
    if (rc != LIO_OK) return rc;
    if (dev != s->device_id) return lio_fail_ext(LIO_ERR_ARG, "the handle and the keyframe store live on different devices", hipSuccess);
    if ((rc = check_device(s->device_id)) != LIO_OK) return rc;
    if ((rc = kf_store_reserve(s, n)) != LIO_OK) return rc;
    if (n) {
        // the intensity sits where the upload said it does (lio_pc2_layout.off_intensity; byte 16 for PCL records; byte 12 for
        // the float4 records lio_s2m_register_raw stages), not at a guessed offset
        hipLaunchKernelGGL(k_rec_to_xyzi4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rec, stride, xyz_off, int_off, (int)n,
                           s->d_pts + s->used);
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
    }
    kf_store_commit(s, n, id_out);
    return LIO_OK;
}

extern "C" int lio_assemble_map_resident(lio_s2m_handle* h, lio_kf_store* st, int32_t n_sel, const int32_t* ids,
                                         const float* poses, float leaf, void* out, size_t out_stride, size_t* n_out)
{
    if (!st || n_sel < 0 || (n_sel && (!ids || !poses))) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if ((out && (out_stride < 20 || (out_stride & 3))) || !(leaf > 0.0f))
        return lio_fail_ext(LIO_ERR_ARG, "output stride must be >= 20 and a multiple of 4, leaf > 0", hipSuccess);
    int rc = check_device(st->device_id);
    if (rc != LIO_OK) return rc;
    size_t total = 0;
    std::vector<LioKfDesc> kf((size_t)n_sel);
    std::vector<int2> chunks;
    for (int k = 0; k < n_sel; ++k) {
        if (ids[k] < 0 || (size_t)ids[k] >= st->off.size()) return lio_fail_ext(LIO_ERR_ARG, "unknown keyframe id", hipSuccess);
        kf[k].src = (int)st->off[ids[k]]; kf[k].first = (int)total; kf[k].n = (int)st->cnt[ids[k]]; kf[k].pad = 0;
        for (size_t b = 0; b < st->cnt[ids[k]]; b += 256) chunks.push_back(make_int2(k, (int)b));
        total += st->cnt[ids[k]];
    }
    if (total > 0x7fffffffull - 1024) return lio_fail_ext(LIO_ERR_CAPACITY, "too many points", hipSuccess);
    if (n_out) *n_out = 0;
    if (lio_s2m_takes_device_map(h)) {
        // The node path (MO:1556-1588 straight into the resident map of `h`): everything on the handle's stream, the
        // temporaries kept in the store, the map's grid laid over the bounding box the voxel filter measured anyway, and no
        // wait at the end -- the registration that follows is ordered behind the build.  The two waits left are the voxel
        // filter's (bounding box, then the number of occupied voxels: both size what comes next).
        hipStream_t s = lio_s2m_stream_of(h);
        st->v_kf.swap(kf);                 // (host arrays of the asynchronous copies live in the store)
        st->v_chunks.swap(chunks);
        HIPCHK(st->world.alloc(total * sizeof(float4)));
        HIPCHK(st->d_kf.alloc(sizeof(LioKfDesc) * (size_t)(n_sel ? n_sel : 1)));
        HIPCHK(st->d_poses.alloc(sizeof(float) * 6 * (size_t)(n_sel ? n_sel : 1)));
        HIPCHK(st->d_chunks.alloc(sizeof(int2) * (st->v_chunks.size() ? st->v_chunks.size() : 1)));
        if (n_sel) {
            HIPCHK(hipMemcpyAsync(st->d_kf.p, st->v_kf.data(), sizeof(LioKfDesc) * (size_t)n_sel, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(st->d_poses.p, poses, sizeof(float) * 6 * (size_t)n_sel, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_kf_transforms, dim3((n_sel + 63) / 64), dim3(64), 0, s, st->d_kf.as<LioKfDesc>(), st->d_poses.as<float>(), n_sel);
        }
        HIPCHK(st->vws.bbox.alloc(6 * sizeof(unsigned)));
        HIPCHK(st->blk_box.alloc(sizeof(float) * 6 * (st->v_chunks.size() ? st->v_chunks.size() : 1)));
        if (!st->v_chunks.empty()) {
            HIPCHK(hipMemcpyAsync(st->d_chunks.p, st->v_chunks.data(), sizeof(int2) * st->v_chunks.size(), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_transform_clouds_bbox, dim3((unsigned)st->v_chunks.size()), dim3(256), 0, s, st->d_pts,
                               st->d_kf.as<LioKfDesc>(), st->d_chunks.as<int2>(), st->world.as<float4>(), st->blk_box.as<float>());
        }
        hipLaunchKernelGGL(k_bbox_reduce, dim3(1), dim3(256), 0, s, st->blk_box.as<float>(), (int)st->v_chunks.size(), st->vws.bbox.as<unsigned>());
        int no = 0;
        float box[6];
        rc = voxel_grid_device<LioKeep>(st->world.as<float4>(), (int)total, leaf, st->ds, &no, s, st->vws, false, box, true);
        if (rc < 0) return rc;
        // (`poses` was consumed by the voxel filter's first wait, which follows its copy on the same stream)
        const int rc3 = lio_s2m_set_map_device_bbox(h, st->ds.as<float4>(), (size_t)no, box);
        if (rc3 != LIO_OK) return rc3;
        const int rc2 = copy_out(st->ds.as<float4>(), no, out, out_stride, s);
        if (rc2 < 0) return rc2;
        if (n_out) *n_out = (size_t)no;
        return rc;
    }
    hipStream_t s = nullptr;
    Buf d_kf, d_poses, d_chunks, world, ds;
    HIPCHK(world.alloc(total * sizeof(float4)));
    HIPCHK(d_kf.alloc(sizeof(LioKfDesc) * (size_t)(n_sel ? n_sel : 1)));
    HIPCHK(d_poses.alloc(sizeof(float) * 6 * (size_t)(n_sel ? n_sel : 1)));
    HIPCHK(d_chunks.alloc(sizeof(int2) * (chunks.size() ? chunks.size() : 1)));
    if (n_sel) {
        HIPCHK(hipMemcpyAsync(d_kf.p, kf.data(), sizeof(LioKfDesc) * (size_t)n_sel, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_poses.p, poses, sizeof(float) * 6 * (size_t)n_sel, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_kf_transforms, dim3((n_sel + 63) / 64), dim3(64), 0, s, d_kf.as<LioKfDesc>(), d_poses.as<float>(), n_sel);
    }
    if (!chunks.empty()) {
        HIPCHK(hipMemcpyAsync(d_chunks.p, chunks.data(), sizeof(int2) * chunks.size(), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_transform_clouds, dim3((unsigned)chunks.size()), dim3(256), 0, s, st->d_pts,
                           d_kf.as<LioKfDesc>(), d_chunks.as<int2>(), world.as<float4>());
    }
    HIPCHK(hipStreamSynchronize(s));       // kf / chunks are stack vectors
    int no = 0;
    rc = voxel_grid_device(world.as<float4>(), (int)total, leaf, ds, &no, s);
    if (rc < 0) return rc;
    if (h) {                               // (a multi-device or map-sharing handle: through the generic path)
        const int rc3 = lio_s2m_set_map_device_xyzi(h, ds.as<float4>(), (size_t)no);
        if (rc3 != LIO_OK) return rc3;
    }
    const int rc2 = copy_out(ds.as<float4>(), no, out, out_stride, s);
    if (rc2 < 0) return rc2;
    if (n_out) *n_out = (size_t)no;
    return rc;
}

// One-shot form: host keyframe clouds in, map out (uploads into a temporary store).
extern "C" int lio_assemble_map(lio_s2m_handle* h, int32_t device_id, int32_t n_kf, const void* const* clouds,
                                const size_t* n_pts, size_t stride, const float* poses, float leaf,
                                void* out, size_t out_stride, size_t* n_out)
{
    if (!clouds || !n_pts || !poses || n_kf < 0) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    lio_kf_store* st = nullptr;
    int rc = lio_kf_store_create(device_id, &st);
    if (rc != LIO_OK) return rc;
    std::vector<int32_t> ids((size_t)n_kf);
    for (int k = 0; k < n_kf && rc == LIO_OK; ++k) rc = lio_kf_store_add(st, clouds[k], n_pts[k], stride, &ids[k]);
    if (rc == LIO_OK) rc = lio_assemble_map_resident(h, st, n_kf, ids.data(), poses, leaf, out, out_stride, n_out);
    lio_kf_store_destroy(st);
    return rc;
}


// ------------------------------------------------ one callback on the device: downsample + register (SURVEY 8f / verdict r2)
// Staged cloud and voxel-filter workspace of lio_s2m_register_raw, kept on the handle from one callback to the next.
struct LioRawWs {
    LioKeep raw, xyzi, ds;
    LioVoxWs<LioKeep> vws;
    // the upload and the voxel filter of the sweep run on a stream of their own: they do not depend on the local map, whose
    // assembly (lio_assemble_map_resident: K6 + K7 + grid build, ~0.25 ms of small kernels) is usually still in flight on the
    // handle's stream when the node calls lio_s2m_register_raw -- the two chains overlap on the GPU, and the filter's host
    // waits (bounding box, voxel count) no longer wait for the map as well
    hipStream_t aux = nullptr;
    hipEvent_t ev_in = nullptr, ev_done = nullptr;
};

void lio_raw_ws_free(LioRawWs* w)
{
    if (!w) return;
    LioKeep* keep[] = { &w->raw, &w->xyzi, &w->ds, &w->vws.bbox, &w->vws.large, &w->vws.pairs_a, &w->vws.pairs_b, &w->vws.hist,
                        &w->vws.blk_heads, &w->vws.seg_start, &w->vws.d_no, &w->vws.row_total };
    for (LioKeep* k : keep) k->release();
    if (w->ev_in) (void)hipEventDestroy(w->ev_in);
    if (w->ev_done) (void)hipEventDestroy(w->ev_done);
    if (w->aux) (void)hipStreamDestroy(w->aux);
    delete w;
}

// downsampleCurrentScan MO:1605-1611 + scan2MapOptimization MO:1839-1865 without a host round trip in between:
// the deskewed cloud goes up once (or is read in place when it already lives on the device), is voxel-filtered on the
// handle's stream with the workspace kept on the handle, and the Gauss-Newton loop runs on the filter's output where it
// lies (float4 x,y,z,intensity records).  The result is bit-identical to lio_voxel_grid followed by lio_s2m_register
// on its output: same filter code, same registration code.  The filtered cloud stays staged on the handle, so
// lio_kf_store_add_from_handle turns it into a keyframe (saveKeyFramesAndFactor MO:2136-2142) with its intensities.
extern "C" int lio_s2m_register_raw(lio_s2m_handle* h, const void* data, size_t n_points, const lio_pc2_layout* layout, float leaf,
                                    float pose[6], lio_s2m_result* res, void* ds_out, size_t ds_out_stride, size_t* n_ds)
{
    if (!h || !layout || !pose) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if (h->multi || h->corner_active) return lio_fail_ext(LIO_ERR_ARG, "lio_s2m_register_raw needs a plain single-device handle", hipSuccess);
    if (lio_pc2_check_xyz(layout) != LIO_OK) return LIO_ERR_ARG;
    if (!(leaf > 0.0f)) return lio_fail_ext(LIO_ERR_ARG, "leaf must be positive", hipSuccess);
    if (ds_out && (ds_out_stride < 20 || (ds_out_stride & 3))) return lio_fail_ext(LIO_ERR_ARG, "output stride must be >= 20 and a multiple of 4", hipSuccess);
    if (n_points && !data) return lio_fail_ext(LIO_ERR_ARG, "null cloud", hipSuccess);
    if (n_points > 0x7fffffffull - 1024) return lio_fail_ext(LIO_ERR_CAPACITY, "cloud too large", hipSuccess);
    if (n_ds) *n_ds = 0;
    int rc = check_device(h->cfg.device_id);
    if (rc != LIO_OK) return rc;
    if (!h->raw_ws) h->raw_ws = new LioRawWs();
    LioRawWs* w = h->raw_ws;
    if (!w->aux) {
        HIPCHK(hipStreamCreateWithFlags(&w->aux, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&w->ev_in, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&w->ev_done, hipEventDisableTiming));
    }
    hipStream_t s = w->aux;                              // upload + filter here; the registration on h->stream, behind the map
    const size_t step = layout->point_step, n = n_points;
    // the blob: read in place when it is device memory of this device, else one H2D copy (a true DMA when pinned)
    const unsigned char* d_rec = nullptr;
    struct PinGuard {                                    // hipHostRegister for the duration of the call, released on every path
        void* p = nullptr;
        ~PinGuard() { if (p) (void)hipHostUnregister(p); }
    } pin;
    if (n) {
        hipPointerAttribute_t at;
        bool in_place = false;
        if (hipPointerGetAttributes(&at, data) == hipSuccess) in_place = at.type == hipMemoryTypeDevice && at.device == h->cfg.device_id;
        (void)hipGetLastError();                         // (pageable host memory is reported as an error)
        if (in_place) {
            d_rec = (const unsigned char*)data;
            // (device memory may have been produced by work queued on the handle's stream: keep that order)
            HIPCHK(hipEventRecord(w->ev_in, h->stream));
            HIPCHK(hipStreamWaitEvent(s, w->ev_in, 0));
        } else {
            if (layout->pin_host) {
                if (hipHostRegister(const_cast<void*>(data), n * step, hipHostRegisterDefault) == hipSuccess) pin.p = const_cast<void*>(data);
                (void)hipGetLastError();
            }
            HIPCHK(w->raw.alloc(n * step));
            HIPCHK(hipMemcpyAsync(w->raw.p, data, n * step, hipMemcpyHostToDevice, s));
            d_rec = w->raw.as<unsigned char>();
        }
        HIPCHK(w->xyzi.alloc(n * sizeof(float4)));
        hipLaunchKernelGGL(k_rec_to_xyzi4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_rec, step, (size_t)layout->off_x,
                           layout->off_intensity >= 0 ? layout->off_intensity : -1, (int)n, w->xyzi.as<float4>());
    }
    int no = 0;
    // (the filter's first host wait -- the bounding box -- also covers the H2D copy: the caller's blob is free again)
    rc = voxel_grid_device<LioKeep>(w->xyzi.as<float4>(), (int)n, leaf, w->ds, &no, s, w->vws, false, nullptr);
    if (rc < 0) return rc;                               // (rc == 1: PCL would pass the cloud through -- and so did we)
    if (n == 0) HIPCHK(w->ds.alloc(sizeof(float4)));
    // the filter's last kernels (centroids) may still be in flight: the registration on the handle's stream waits for them
    HIPCHK(hipEventRecord(w->ev_done, s));
    HIPCHK(hipStreamWaitEvent(h->stream, w->ev_done, 0));
    h->int_off = 12;                                     // the staged records are float4 (x, y, z, intensity)
    const int rr = lio_s2m_register(h, w->ds.p, (size_t)no, sizeof(float4), pose, res);
    h->int_off = -2;
    if (rr < 0) return rr;
    if (ds_out) { const int rc2 = copy_out(w->ds.as<float4>(), no, ds_out, ds_out_stride, h->stream); if (rc2 < 0) return rc2; }
    if (n_ds) *n_ds = (size_t)no;
    return rr;
}
