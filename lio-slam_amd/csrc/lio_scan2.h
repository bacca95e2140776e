// lio_scan2.h -- two exclusive scans over one index range in a single pass.  The pair of running sums travels in one
// 64-bit word (each sum < 2^31).  NZ = false: the second sequence is b[]; NZ = true: it is (a[i] > 0) -- "occupied
// entries before i".  Tiles of 4096 entries; global loads and stores are coalesced (entry k * 256 + thread), the
// per-thread runs of 16 consecutive entries go through LDS (padded by one word per 16: conflict-free both ways).
// Used by the voxel filter (first point / output slot of every voxel) and by the map build (cell starts /
// neighbourhood-row starts).  out_a and out_b have n + 1 entries.
#pragma once
#include <hip/hip_runtime.h>

#define LIO_S2_ITEMS 16
#define LIO_S2_TILE (256 * LIO_S2_ITEMS)

__device__ __forceinline__ unsigned long long lio_s2_block_exscan(unsigned long long v, unsigned long long* total, unsigned long long* s_wave)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const unsigned long long q = s_wave[w]; if (w < wave) wave_off += q; tot += q; }
    __syncthreads();
    *total = tot;
    return wave_off + incl - v;
}

template <bool NZ>
__device__ __forceinline__ unsigned long long lio_s2_pack(const int* __restrict__ a, const int* __restrict__ b, int i)
{
    const int va = a[i];
    return (unsigned long long)(unsigned)va + (NZ ? (va > 0 ? (1ull << 32) : 0ull) : ((unsigned long long)(unsigned)b[i] << 32));
}

template <bool NZ>
static __global__ __launch_bounds__(256) void k_s2_tile_sums(const int* __restrict__ a, const int* __restrict__ b, int n,
                                                             unsigned long long* __restrict__ tile_sums)
{
    __shared__ unsigned long long s_wave[4];
    const int base = blockIdx.x * LIO_S2_TILE;
    unsigned long long acc = 0;
#pragma unroll
    for (int k = 0; k < LIO_S2_ITEMS; ++k) {
        const int i = base + k * 256 + (int)threadIdx.x;
        if (i < n) acc += lio_s2_pack<NZ>(a, b, i);
    }
    unsigned long long tot;
    lio_s2_block_exscan(acc, &tot, s_wave);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

static __global__ __launch_bounds__(256) void k_s2_tile_offsets(unsigned long long* __restrict__ tile_sums, int n_tiles)
{
    __shared__ unsigned long long s_wave[4];
    unsigned long long carry = 0;
    for (int t = 0; t < n_tiles; t += 256) {
        const int i = t + (int)threadIdx.x;
        const unsigned long long v = i < n_tiles ? tile_sums[i] : 0ull;
        unsigned long long tot;
        const unsigned long long ex = lio_s2_block_exscan(v, &tot, s_wave);
        if (i < n_tiles) tile_sums[i] = carry + ex;
        carry += tot;
    }
}

template <bool NZ>
static __global__ __launch_bounds__(256) void k_s2_apply(const int* __restrict__ a, const int* __restrict__ b, int n,
                                                         const unsigned long long* __restrict__ tile_offsets,
                                                         int* __restrict__ out_a, int* __restrict__ out_b)
{
    __shared__ unsigned long long s_wave[4];
    __shared__ unsigned long long s_v[LIO_S2_TILE + 256];
    const int base = blockIdx.x * LIO_S2_TILE, tid = (int)threadIdx.x;
#pragma unroll
    for (int k = 0; k < LIO_S2_ITEMS; ++k) {
        const int i = k * 256 + tid;
        s_v[i + (i >> 4)] = (base + i < n) ? lio_s2_pack<NZ>(a, b, base + i) : 0ull;
    }
    __syncthreads();
    unsigned long long v[LIO_S2_ITEMS], acc = 0;
#pragma unroll
    for (int k = 0; k < LIO_S2_ITEMS; ++k) { v[k] = s_v[tid * 17 + k]; acc += v[k]; }
    unsigned long long tot;
    unsigned long long run = tile_offsets[blockIdx.x] + lio_s2_block_exscan(acc, &tot, s_wave);
#pragma unroll
    for (int k = 0; k < LIO_S2_ITEMS; ++k) {
        s_v[tid * 17 + k] = run;
        run += v[k];
        if (base + tid * 16 + k == n - 1) { out_a[n] = (int)(unsigned)run; out_b[n] = (int)(run >> 32); }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LIO_S2_ITEMS; ++k) {
        const int i = k * 256 + tid;
        if (base + i < n) {
            const unsigned long long r = s_v[i + (i >> 4)];
            out_a[base + i] = (int)(unsigned)r;
            out_b[base + i] = (int)(r >> 32);
        }
    }
}

// tiles: (n + LIO_S2_TILE - 1) / LIO_S2_TILE + 1 unsigned long longs
template <bool NZ>
static inline void lio_launch_scan2(const int* a, const int* b, int n, unsigned long long* tiles, int* out_a, int* out_b, hipStream_t s)
{
    const int n_tiles = (n + LIO_S2_TILE - 1) / LIO_S2_TILE;
    hipLaunchKernelGGL(k_s2_tile_sums<NZ>, dim3(n_tiles), dim3(256), 0, s, a, b, n, tiles);
    hipLaunchKernelGGL(k_s2_tile_offsets, dim3(1), dim3(256), 0, s, tiles, n_tiles);
    hipLaunchKernelGGL(k_s2_apply<NZ>, dim3(n_tiles), dim3(256), 0, s, a, b, n, tiles, out_a, out_b);
}
