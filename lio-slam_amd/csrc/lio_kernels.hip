// lio_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the scan-to-map
// registration path.  MO = /root/reference/src/liorf/src/mapOptmization.cpp.
//
//   map build : k_map_bbox, k_map_cell_count, k_scan_*, k_map_scatter
//               (replaces the kd-tree build at MO:1846)
//   GN iterate: k_s2m_iterate  = surfOptimization MO:1618-1687
//                              + combineOptimizationCoeffs MO:1689-1700 (as a sum)
//                              + LMOptimization MO:1702-1837 (last workgroup of a scan)
//
// Compile with -ffp-contract=off (bit parity with the reference's arithmetic).
#include <hip/hip_runtime.h>
#include "lio_types.h"
#include "lio_device_math.h"
#include "lio_kernels.h"

#include "lio_s2m_device.h"
#include "lio_scan2.h"

LIO_DEV unsigned lio_f2ord(float f)   // order-preserving float -> uint
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ------------------------------------------------------------- AoS -> SoA
// pcl::PointXYZI records (x,y,z at byte 0,4,8; stride given) -> x[],y[],z[]
__global__ void k_aos_to_soa(const unsigned char* __restrict__ src, size_t stride, int n,
                             float* __restrict__ x, float* __restrict__ y, float* __restrict__ z,
                             float4* __restrict__ xyz4 /* optional */)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(src + (size_t)i * stride);
    const float a = p[0], b = p[1], c = p[2];
    x[i] = a; y[i] = b; z[i] = c;
    if (xyz4) xyz4[i] = make_float4(a, b, c, 0.0f);
}

__global__ void k_xyzi4_to_soa(const float4* __restrict__ src, int n, float* __restrict__ x, float* __restrict__ y,
                               float* __restrict__ z, float4* __restrict__ xyz4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = src[i];
    x[i] = v.x; y[i] = v.y; z[i] = v.z;
    xyz4[i] = make_float4(v.x, v.y, v.z, 0.0f);
}

// ---------------------------------------------------------------- map build
// bbox[0..2] = min (ordered uint), bbox[3..5] = max
__global__ void k_map_bbox(const float* __restrict__ x, const float* __restrict__ y,
                           const float* __restrict__ z, int n, unsigned* __restrict__ bbox)
{
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float v[3] = { x[i], y[i], z[i] };
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // NaN / inf coordinates are ignored for the box (they can never be a
            // neighbour within 1 m of anything)
            if (fabsf(v[a]) <= LIO_MAX_COORD) { mn[a] = fminf(mn[a], v[a]); mx[a] = fmaxf(mx[a], v[a]); }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    }
    // one set of atomics per workgroup (six hot words: per-wave atomics serialise)
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = s_mn[0][a], hi = s_mx[0][a];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fminf(lo, s_mn[w][a]); hi = fmaxf(hi, s_mx[w][a]); }
        atomicMin(&bbox[a], lio_f2ord(lo));
        atomicMax(&bbox[3 + a], lio_f2ord(hi));
    }
}

LIO_DEV int lio_map_cell(const LioGrid& g, float px, float py, float pz)
{
    if (!(fabsf(px) <= LIO_MAX_COORD && fabsf(py) <= LIO_MAX_COORD && fabsf(pz) <= LIO_MAX_COORD)) return -1;
    const int cx = lio_cell_coord(px, g.ox, g.inv_cell, g.nx);
    const int cy = lio_cell_coord(py, g.oy, g.inv_cell, g.ny);
    const int cz = lio_cell_coord(pz, g.oz, g.inv_cell, g.nz);
    if (cx < 0 || cx >= g.nx || cy < 0 || cy >= g.ny || cz < 0 || cz >= g.nz) return -1;
    return (cz * g.ny + cy) * g.nx + cx;
}

__global__ void k_map_cell_count(LioGrid g, const float* __restrict__ x, const float* __restrict__ y,
                                 const float* __restrict__ z, int n,
                                 int* __restrict__ cell_of, int* __restrict__ cell_count)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = lio_map_cell(g, x[i], y[i], z[i]);
    cell_of[i] = c;
    if (c >= 0 && cell_count) atomicAdd(&cell_count[c], 1);       // (the counts serve the cell-sorted copy of the LDS-staged variant only)
}

// Occupied cells of the grid (flags[0..n_cells) zeroed by the caller, the count lands in flags[n_cells]): points per occupied
// cell is the density estimate that decides how many tight row tables a map is worth (lio_map_finish).
__global__ void k_map_occupancy(LioGrid g, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n,
                                int* __restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool first = false;
    if (i < n) {
        const int c = lio_map_cell(g, x[i], y[i], z[i]);
        first = c >= 0 && atomicExch(&flags[c], 1) == 0;
    }
    const unsigned long long m = __ballot(first);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&flags[g.n_cells], __popcll(m));
}

// exclusive scan of cell_count[0..n) -> cell_start[0..n], three phases
#define LIO_SCAN_ITEMS 16
#define LIO_SCAN_TILE (256 * LIO_SCAN_ITEMS)

LIO_DEV int lio_block_exclusive_scan(int v, int* total, int* s_wave /* >= 4 ints */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int s = s_wave[w]; if (w < wave) wave_off += s; tot += s; }
    __syncthreads();
    *total = tot;
    return wave_off + incl - v;
}

__global__ __launch_bounds__(256) void k_scan_tile_sums(const int* __restrict__ in, int n, int* __restrict__ tile_sums)
{
    __shared__ int s_wave[4];
    const int base = blockIdx.x * LIO_SCAN_TILE + threadIdx.x * LIO_SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < LIO_SCAN_ITEMS; ++k) if (base + k < n) s += in[base + k];
    int tot;
    lio_block_exclusive_scan(s, &tot, s_wave);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_tile_offsets(int* __restrict__ tile_sums, int n_tiles)
{
    __shared__ int s_wave[4];
    int carry = 0;
    for (int b = 0; b < n_tiles; b += 256) {
        const int i = b + threadIdx.x;
        const int v = i < n_tiles ? tile_sums[i] : 0;
        int tot;
        const int ex = lio_block_exclusive_scan(v, &tot, s_wave);
        if (i < n_tiles) tile_sums[i] = carry + ex;
        carry += tot;
    }
}

__global__ __launch_bounds__(256) void k_scan_apply(const int* __restrict__ in, int n,
                                                    const int* __restrict__ tile_offsets,
                                                    int* __restrict__ out /* n+1 */)
{
    __shared__ int s_wave[4];
    const int base = blockIdx.x * LIO_SCAN_TILE + threadIdx.x * LIO_SCAN_ITEMS;
    int v[LIO_SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < LIO_SCAN_ITEMS; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    int tot;
    int run = tile_offsets[blockIdx.x] + lio_block_exclusive_scan(s, &tot, s_wave);
#pragma unroll
    for (int k = 0; k < LIO_SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
        if (base + k == n - 1) out[n] = run;
    }
}

// sorted[cell_start[c] + slot] = (x, y, z, bits(original index))
__global__ void k_map_scatter(const float* __restrict__ x, const float* __restrict__ y,
                              const float* __restrict__ z, int n,
                              const int* __restrict__ cell_of, const int* __restrict__ cell_start,
                              int* __restrict__ cell_fill, float4* __restrict__ sorted)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of[i];
    if (c < 0) return;
    const int slot = atomicAdd(&cell_fill[c], 1);
    sorted[cell_start[c] + slot] = make_float4(x[i], y[i], z[i], __int_as_float(i));
}

// ---- 9x replicated neighbourhood rows -------------------------------------
// For the candidate scan every (y,z) row R of the grid gets its own list: all
// map points of the (2k+1)x(2k+1) rows around R, bucketed by x cell (k = cells
// per metre of search radius, LioGrid::k).  The (2k+1)^3-cell neighbourhood of
// a query in cell (cx,cy,cz) is then ONE contiguous run
// [nbr_start[R*nx + cx-k], nbr_start[R*nx + cx+k+1]) of 16-byte records -- one
// range lookup and a unit-stride stream per lane instead of many short,
// divergent runs.  Costs (2k+1)^2 x 16 B of HBM per map point (k=2: 400 MB per
// million points; sized for 288 GB).  Finer cells (k=2) cut the searched
// volume from 27 to 15.6 m^3 around a 1 m gate, i.e. ~1.7x fewer candidates.
// One thread per (map point, neighbouring (y,z) row): the point joins the row lists of the (2k+1)^2 rows around its own,
// at its own x cell.  slot[t] = its arrival number in that cell's list (the value the counting atomic returns; -1: no such
// row), so that k_map_nbr_scatter needs no second round of atomics.  The order inside a list is arbitrary: the candidate
// scan keeps the five smallest (d2, index) keys whatever order they come in.
// The bucket of a replica along x is the point's FINE x cell (LioGrid::xs per cell): (4 t) floors into [4 cx, 4 cx + 3] for
// t = (v - ox) * inv_cell in cell cx (the scaling by a power of two is exact), so a point inside the grid is inside the fine grid.
// (g.oy/oz/inv_cell/ny/nz/k describe the table being built: the grid's own rows, or its tight rows -- lio_tight_grid.)
__global__ void k_map_nbr_count(LioGrid g, const int* __restrict__ cell_of, const float* __restrict__ x_, const float* __restrict__ y_,
                                const float* __restrict__ z_, int n, int* __restrict__ nbr_count, int* __restrict__ slot)
{
    const int side = 2 * g.k + 1, reps = side * side;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * reps) return;
    const int i = (int)(t / reps), r = (int)(t - (long long)i * reps);
    int sl = -1;
    if (cell_of[i] >= 0) {                                              // (a finite point inside the grid)
        const int y = min(max(lio_cell_coord(y_[i], g.oy, g.inv_cell, g.ny), 0), g.ny - 1);
        const int z = min(max(lio_cell_coord(z_[i], g.oz, g.inv_cell, g.nz), 0), g.nz - 1);
        const int xf = min(max(lio_cell_coord(x_[i], g.ox, g.inv_cell_x, g.nxf), 0), g.nxf - 1);
        const int yy = y + r % side - g.k, zz = z + r / side - g.k;
        if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) sl = atomicAdd(&nbr_count[(zz * g.ny + yy) * g.nxf + xf], 1);
    }
    slot[t] = sl;
}

// Record layout of nbr_pts: records are stored in PAIRS, transposed, so that one 16-byte load
// yields two candidates' x (and y), the next one their z (and index):
//   float4[2p]   = (x_2p, x_2p+1, y_2p, y_2p+1)     float4[2p+1] = (z_2p, z_2p+1, w_2p, w_2p+1)
// which is exactly the operand shape of v_pk_add_f32 / v_pk_mul_f32 (two candidates per VALU op).
LIO_DEV void lio_nbr_store(float* __restrict__ base, int pos, float x, float y, float z, float w)
{
    float* p = base + (size_t)(pos >> 1) * 8 + (pos & 1);
    p[0] = x; p[2] = y; p[4] = z; p[6] = w;
}

// Every (y,z) row list is padded to a multiple of LIO_ROW_ALIGN (eight) records (the pad goes to its last cell and
// stays filled with dummies), so row lists start 4-aligned and an aligned group of four never
// straddles two row lists -- neighbouring rows hold copies of the same map points, and a candidate
// seen twice would corrupt the top-5.
__global__ __launch_bounds__(256) void k_map_nbr_pad_rows(LioGrid g, int* __restrict__ nbr_count)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // one wave per row
    if (row >= g.ny * g.nz) return;
    int s = 0;
    for (int x = lane; x < g.nxf; x += 64) s += nbr_count[(size_t)row * g.nxf + x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) nbr_count[(size_t)row * g.nxf + g.nxf - 1] += (LIO_ROW_ALIGN - (s & (LIO_ROW_ALIGN - 1))) & (LIO_ROW_ALIGN - 1);
}

__global__ void k_map_nbr_fill(float4* __restrict__ nbr_pts, int n_rec4)
{
    // unused slots (array tail, never-written padding) hold far-away dummies with a huge index:
    // finite squared distance, never among the five nearest
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec4) return;
    nbr_pts[i] = (i & 1) ? make_float4(LIO_MAX_COORD, LIO_MAX_COORD, __int_as_float(LIO_IDX_MASK), __int_as_float(LIO_IDX_MASK))
                         : make_float4(LIO_MAX_COORD, LIO_MAX_COORD, LIO_MAX_COORD, LIO_MAX_COORD);
}

__global__ void k_map_nbr_scatter(LioGrid g, const float* __restrict__ x_, const float* __restrict__ y_,
                                  const float* __restrict__ z_, int n,
                                  const int* __restrict__ nbr_start, const int* __restrict__ slot,
                                  float4* __restrict__ nbr_pts)
{
    const int side = 2 * g.k + 1, reps = side * side;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * reps) return;
    const int sl = slot[t];
    if (sl < 0) return;
    const int i = (int)(t / reps), r = (int)(t - (long long)i * reps);
    const int y = min(max(lio_cell_coord(y_[i], g.oy, g.inv_cell, g.ny), 0), g.ny - 1);
    const int z = min(max(lio_cell_coord(z_[i], g.oz, g.inv_cell, g.nz), 0), g.nz - 1);
    const int xf = min(max(lio_cell_coord(x_[i], g.ox, g.inv_cell_x, g.nxf), 0), g.nxf - 1);
    const int yy = y + r % side - g.k, zz = z + r / side - g.k;
    lio_nbr_store(reinterpret_cast<float*>(nbr_pts), nbr_start[(zz * g.ny + yy) * g.nxf + xf] + sl, x_[i], y_[i], z_[i], __int_as_float(i));
}

// ------------------------------------------------- scan tile sort (upload)
// The scan (laserCloudSurfLastDS, MO:138) is re-ordered once per upload so that
// the 256*PPT consecutive points of an association workgroup lie in a compact
// blob: points are bucketed by 4 m tiles of a scan-local grid (x-fastest tile
// id, scans in batch order), ties inside a tile keep the caller's order, so the
// order -- and with it every reduction tree -- is deterministic.
LIO_DEV int lio_tile_coord(float v, float origin, float inv_tile, int n)
{
    float c = floorf((v - origin) * inv_tile);
    c = fminf(fmaxf(c, 0.0f), (float)(n - 1));      // NaN -> 0
    return (int)c;
}

// ---- the whole upload-time reorder of ONE scan by ONE workgroup, in LDS ------------------------------------------
// For scans of at most 16383 points (every downsampled scan; raw sweeps take the multi-kernel path below): bounding box
// (finite coordinates), tile grid (the same arithmetic as the host code of the multi-kernel path: tile edge doubled
// until the grid has <= 262144 tiles), key = (linear tile id << 14) | caller index, a stable sort of the 32-bit keys by tile
// id, then perm[] and the SoA are written in sorted order.  Same permutation as the counting sort + rank sort (tile id
// ascending, caller index ascending inside a tile), but no global atomics, no histogram over a million mostly empty
// tiles, no host round trip for the bounding boxes, and deterministic by construction.
#define LIO_SORT_THREADS 512
// The sort is a stable LSD radix sort on the tile id (round 3; rounds 1-2 ran a bitonic network over the keys in LDS: 91
// compare-exchange stages and barriers for 8192 keys, 173 us per 512-scan batch against 60 us now): the keys live in
// REGISTERS (ROWS x 64 consecutive keys per wave, wave-striped so that (wave, row, lane) order is the caller's order), every
// pass ranks them inside the wave by digit with eight ballots per row (no atomics: a wave runs in lockstep), adds a prefix
// over (digit, wave) and scatters into the one LDS buffer, from which the next pass -- or the output phase -- reads them back
// in sorted order.  8-bit digits over the bits the scan's tile grid really uses: two passes for the ~10^4 tiles of a street
// scene.  The permutation is the one the text above defines by construction: tile id ascending, caller index ascending inside
// a tile (the sort is stable and starts from the caller's order).
template <int ROWS>
__global__ __launch_bounds__(LIO_SORT_THREADS) void k_scan_sort_radix(const unsigned char* __restrict__ stage, size_t stride,
                                                                      const LioScanState* __restrict__ st, float tile0, int shard_axis,
                                                                      int* __restrict__ perm, float* __restrict__ x,
                                                                      float* __restrict__ y, float* __restrict__ z)
{
    constexpr int WAVES = LIO_SORT_THREADS / 64;
    __shared__ unsigned s_key[ROWS * LIO_SORT_THREADS];
    __shared__ volatile int s_cnt[WAVES][256];             // per wave and digit: keys seen so far in this pass
    __shared__ int s_base[WAVES][256];                     // position of the first key of (digit, wave)
    __shared__ int s_dig[256];
    __shared__ float s_mn[WAVES][3], s_mx[WAVES][3];
    __shared__ float s_o[4];
    __shared__ int s_nt[7];
    const int scan = blockIdx.x;
    const int n = st[scan].n_pts, base = st[scan].offset;
    if (n <= 0) return;
    const unsigned char* src = stage + (size_t)base * stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 1. bounding box (finite coordinates)
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += LIO_SORT_THREADS) {          // (unrolled: four strided records in flight per lane)
        const float* p = reinterpret_cast<const float*>(src + (size_t)i * stride);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = p[a];
            if (fabsf(v) <= 3.0e38f) { mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = s_mn[0][a]; hi[a] = s_mx[0][a];
            for (int w = 1; w < WAVES; ++w) { lo[a] = fminf(lo[a], s_mn[w][a]); hi[a] = fmaxf(hi[a], s_mx[w][a]); }
            if (!(lo[a] <= hi[a])) { lo[a] = 0.0f; hi[a] = 0.0f; }
        }
        float tile = tile0, inv_tile;
        int ntx, nty, ntz;
        for (;;) {                                          // (the host loop of lio_s2m_batch_upload, verbatim)
            inv_tile = 1.0f / tile;
            const double ex = floor(((double)hi[0] - lo[0]) * inv_tile) + 1.0, ey = floor(((double)hi[1] - lo[1]) * inv_tile) + 1.0,
                         ez = floor(((double)hi[2] - lo[2]) * inv_tile) + 1.0;
            if (ex * ey * ez <= 262144.0) { ntx = (int)ex; nty = (int)ey; ntz = (int)ez; break; }
            tile *= 2.0f;
        }
        s_o[0] = lo[0]; s_o[1] = lo[1]; s_o[2] = lo[2]; s_o[3] = inv_tile;
        int kx = 1, ky = ntx, kz = ntx * nty;               // x fastest; with a sharded map the shard axis slowest
        if (shard_axis == 0) { kz = 1; ky = ntz; kx = ntz * nty; }
        else if (shard_axis == 1) { kx = 1; kz = ntx; ky = ntx * ntz; }
        s_nt[0] = ntx; s_nt[1] = nty; s_nt[2] = ntz; s_nt[3] = kx; s_nt[4] = ky; s_nt[5] = kz;
        int bits = 1;                                       // bits of the largest tile id
        while (bits < 18 && (1 << bits) < ntx * nty * ntz) ++bits;
        s_nt[6] = (bits + 7) / 8;                           // radix passes
    }
    __syncthreads();
    // 2. keys, in registers: row k of wave w holds the caller's points w * rows * 64 + k * 64 + [0, 64)
    const int rows = (n + LIO_SORT_THREADS - 1) / LIO_SORT_THREADS;       // <= ROWS (host: max_pts <= ROWS * LIO_SORT_THREADS)
    const int first = wave * rows * 64;
    unsigned item[ROWS];
    unsigned short rank[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = first + k * 64 + lane;
        unsigned key = 0xffffffffu;
        if (k < rows && i < n) {
            const float* p = reinterpret_cast<const float*>(src + (size_t)i * stride);
            const int tx = lio_tile_coord(p[0], s_o[0], s_o[3], s_nt[0]);
            const int ty = lio_tile_coord(p[1], s_o[1], s_o[3], s_nt[1]);
            const int tz = lio_tile_coord(p[2], s_o[2], s_o[3], s_nt[2]);
            key = ((unsigned)(tx * s_nt[3] + ty * s_nt[4] + tz * s_nt[5]) << 14) | (unsigned)i;
        }
        item[k] = key;
        rank[k] = 0;
    }
    const int n_pass = s_nt[6];
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;      // lanes below this one
    for (int pass = 0; pass < n_pass; ++pass) {
        const int shift = 14 + 8 * pass;
        for (int d = threadIdx.x; d < WAVES * 256; d += LIO_SORT_THREADS) (&s_cnt[0][0])[d] = 0;
        __syncthreads();                                     // (also: every key of the previous pass has been read back)
        // 3a. rank inside the wave, row by row
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const bool valid = item[k] != 0xffffffffu;
            unsigned long long peers = __ballot(valid);
            if (peers == 0ull) continue;                     // wave-uniform
            const unsigned d = (item[k] >> shift) & 255u;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            int r = 0;
            if (valid) {
                const int seen = s_cnt[wave][d];             // every lane reads before the leader below writes (one wave, in-order LDS)
                r = seen + __popcll(peers & lt);
                __builtin_amdgcn_wave_barrier();
                if ((peers & lt) == 0ull) s_cnt[wave][d] = seen + __popcll(peers);   // leader = lowest lane of the group
            }
            __builtin_amdgcn_wave_barrier();
            rank[k] = (unsigned short)r;
        }
        __syncthreads();
        // 3b. exclusive prefix over the digits (totals over the waves; a shuffle scan inside each of the four waves that hold the
        //     256 digits, their totals through LDS), then over the waves inside a digit
        int tot = 0, incl = 0;
        if (threadIdx.x < 256) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) tot += s_cnt[w][threadIdx.x];
            incl = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
            if (lane == 63) s_dig[wave] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            int run = incl - tot;
            for (int w = 0; w < wave; ++w) run += s_dig[w];
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { s_base[w][threadIdx.x] = run; run += s_cnt[w][threadIdx.x]; }
        }
        __syncthreads();
        // 3c. scatter into the LDS buffer, read back in sorted order
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
            if (item[k] != 0xffffffffu) s_key[s_base[wave][(item[k] >> shift) & 255u] + (int)rank[k]] = item[k];
        __syncthreads();
        if (pass + 1 < n_pass) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const int i = first + k * 64 + lane;
                item[k] = (k < rows && i < n) ? s_key[i] : 0xffffffffu;
            }
        }
    }
    // 4. permutation and SoA in sorted order
#pragma unroll 4
    for (int j = threadIdx.x; j < n; j += LIO_SORT_THREADS) {
        const int i = (int)(s_key[j] & 0x3fffu);
        const float* p = reinterpret_cast<const float*>(src + (size_t)i * stride);
        perm[base + j] = base + i;
        x[base + j] = p[0]; y[base + j] = p[1]; z[base + j] = p[2];
    }
}

// Bounding box of every scan of the batch (finite coordinates only), one workgroup per 256 points:
// bbox[scan][0..2] = min, [3..5] = max as order-preserving uints (initialised to ~0 / 0 by the host).
__global__ __launch_bounds__(256) void k_scan_bbox(const unsigned char* __restrict__ stage, size_t stride,
                                                   const LioBlockDesc* __restrict__ prep_blocks,
                                                   const LioScanState* __restrict__ st, unsigned* __restrict__ bbox)
{
    const LioBlockDesc bd = prep_blocks[blockIdx.x];
    const int li = bd.first + (int)threadIdx.x;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (li < st[bd.scan].n_pts) {
        const float* p = reinterpret_cast<const float*>(stage + (size_t)(st[bd.scan].offset + li) * stride);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = p[a];
            if (fabsf(v) <= 3.0e38f) { mn[a] = v; mx[a] = v; }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    }
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = s_mn[0][a], hi = s_mx[0][a];
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, s_mn[w][a]); hi = fmaxf(hi, s_mx[w][a]); }
        if (lo <= hi) {
            atomicMin(&bbox[bd.scan * 6 + a], lio_f2ord(lo));
            atomicMax(&bbox[bd.scan * 6 + 3 + a], lio_f2ord(hi));
        }
    }
}

// Wave-aggregated counter update: the points of a wave mostly fall into a handful of tiles (scans arrive in
// voxel / ring order), so the lanes holding equal keys elect one leader that adds their count in ONE atomic;
// scattered single-dword atomics run ~17x below the coalesced rate on gfx950.  After eight rounds whatever is
// left (incoherent input) falls back to one atomic per lane.  Returns base + rank of the lane within its key
// (lane order = caller order inside the wave) when WANT_SLOT, else nothing.
template <bool WANT_SLOT>
LIO_DEV int lio_wave_key_add(int* __restrict__ counters, int key, bool valid)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    int slot = 0;
    bool mine = valid;
#pragma unroll 1
    for (int round = 0; round < 8 && todo; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k0 = __shfl(key, leader);
        const unsigned long long m = __ballot(mine && key == k0);
        if (mine && key == k0) {
            int b = 0;
            if (lane == leader) {
                if (WANT_SLOT) b = atomicAdd(&counters[k0], (int)__popcll(m));
                else atomicAdd(&counters[k0], (int)__popcll(m));
            }
            if (WANT_SLOT) slot = __shfl(b, leader) + (int)__popcll(m & ((1ull << lane) - 1ull));
            mine = false;
        }
        todo &= ~m;
    }
    if (mine) {
        if (WANT_SLOT) slot = atomicAdd(&counters[key], 1);
        else atomicAdd(&counters[key], 1);
    }
    return slot;
}

__global__ __launch_bounds__(256) void k_scan_tile_keys(const unsigned char* __restrict__ stage, size_t stride,
                                                        const LioBlockDesc* __restrict__ prep_blocks,
                                                        const LioScanState* __restrict__ st,
                                                        const LioScanTiles* __restrict__ tiles,
                                                        int* __restrict__ key_of, int* __restrict__ key_count)
{
    const LioBlockDesc bd = prep_blocks[blockIdx.x];
    const int li = bd.first + (int)threadIdx.x;
    const bool valid = li < st[bd.scan].n_pts;
    int key = 0;
    if (valid) {
        const int gi = st[bd.scan].offset + li;
        const LioScanTiles t = tiles[bd.scan];
        const float* p = reinterpret_cast<const float*>(stage + (size_t)gi * stride);
        const int tx = lio_tile_coord(p[0], t.ox, t.inv_tile, t.ntx);
        const int ty = lio_tile_coord(p[1], t.oy, t.inv_tile, t.nty);
        const int tz = lio_tile_coord(p[2], t.oz, t.inv_tile, t.ntz);
        key = t.key_offset + tx * t.mx + ty * t.my + tz * t.mz;
        key_of[gi] = key;
    }
    lio_wave_key_add<false>(key_count, key, valid);
}

__global__ __launch_bounds__(256) void k_scan_tile_scatter(const int* __restrict__ key_of, int n, const int* __restrict__ key_start,
                                                           int* __restrict__ key_fill, int* __restrict__ tmp_idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const int k = valid ? key_of[i] : 0;
    const int slot = lio_wave_key_add<true>(key_fill, k, valid);
    if (valid) tmp_idx[key_start[k] + slot] = i;
}

// Order every tile's points by their caller index: one thread per SLOT of the counting sort's output counts the
// smaller indices in its tile's slot range (tiles hold ~15 points; the lanes of a wave sit in the same one or two
// tiles, so those reads are broadcasts) and writes its index at that rank.  Launched over the points, not over the
// ~10^6 mostly empty tiles.  Tiles with more than LIO_TILE_CAP points (raw sweeps, very small leaf sizes) are copied
// as they are and queued once for k_scan_tile_bigsort, so that the order -- and with it every fp64 summation
// order -- never depends on the arrival order of the atomics in k_scan_tile_scatter.
#define LIO_TILE_CAP 1024
__global__ __launch_bounds__(256) void k_scan_tile_ranksort(const int* __restrict__ key_start, const int* __restrict__ key_of, int n_pts,
                                                            const int* __restrict__ tmp_idx, int* __restrict__ perm,
                                                            int* __restrict__ big_list, int* __restrict__ big_cnt)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pts) return;
    const int v = tmp_idx[j];
    const int k = key_of[v];
    const int b = key_start[k], n = key_start[k + 1] - b;
    if (n > LIO_TILE_CAP) {                          // oversized tile: sorted by a whole workgroup afterwards
        perm[j] = v;
        if (j == b) big_list[atomicAdd(big_cnt, 1)] = k;
        return;
    }
    int rank = 0;
    for (int i = 0; i < n; ++i) rank += (tmp_idx[b + i] < v) ? 1 : 0;
    perm[b + rank] = v;
}

// Oversized tiles: ascending sort of the tile's point indices in place (global memory, one workgroup per
// tile, bitonic network in its all-ascending "flip" form so that the missing elements of the last
// power-of-two block behave as +infinity).  The set of indices is fixed by the counting sort; only their
// order came from atomics, and this removes it.
__global__ __launch_bounds__(256) void k_scan_tile_bigsort(const int* __restrict__ key_start, const int* __restrict__ big_list,
                                                           const int* __restrict__ big_cnt, int* __restrict__ perm)
{
    const int cnt = *big_cnt;
    for (int t = blockIdx.x; t < cnt; t += gridDim.x) {
        const int k = big_list[t];
        const int b = key_start[k], n = key_start[k + 1] - b;
        int* a = perm + b;
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        for (int size = 2; size <= np2; size <<= 1) {
            const int half = size >> 1;
            for (int i = threadIdx.x; i < (np2 >> 1); i += 256) {
                const int blk = i / half, off = i - blk * half;
                const int lo = blk * size + off, hi = blk * size + size - 1 - off;
                if (hi < n) { const int x = a[lo], y = a[hi]; if (x > y) { a[lo] = y; a[hi] = x; } }
            }
            __syncthreads();
            for (int st = half >> 1; st >= 1; st >>= 1) {
                for (int i = threadIdx.x; i < (np2 >> 1); i += 256) {
                    const int lo = (i / st) * 2 * st + (i % st), hi = lo + st;
                    if (hi < n) { const int x = a[lo], y = a[hi]; if (x > y) { a[lo] = y; a[hi] = x; } }
                }
                __syncthreads();
            }
        }
    }
}

__global__ void k_scan_gather_sorted(const unsigned char* __restrict__ stage, size_t stride, int n,
                                     const int* __restrict__ perm,
                                     float* __restrict__ x, float* __restrict__ y, float* __restrict__ z)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float* p = reinterpret_cast<const float*>(stage + (size_t)perm[j] * stride);
    x[j] = p[0]; y[j] = p[1]; z[j] = p[2];
}

// Bounding box (lidar frame, finite coordinates only) of the <= LIO_BLOCK points of every association workgroup, from the
// tile-sorted SoA: what the map-sharded launch culls workgroups with.  box[(offset + first) / LIO_BLOCK + scan][6].
__global__ __launch_bounds__(LIO_BLOCK) void k_block_boxes(const LioBlockDesc* __restrict__ blocks, const LioScanState* __restrict__ st,
                                                           const float* __restrict__ sx, const float* __restrict__ sy,
                                                           const float* __restrict__ sz, float* __restrict__ box)
{
    const LioBlockDesc bd = blocks[blockIdx.x];
    const int li = bd.first + (int)threadIdx.x;
    const int base = st[bd.scan].offset;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (li < st[bd.scan].n_pts) {
        const float v[3] = { sx[base + li], sy[base + li], sz[base + li] };
#pragma unroll
        for (int a = 0; a < 3; ++a) if (fabsf(v[a]) <= 3.0e38f) { mn[a] = v[a]; mx[a] = v[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    }
    __shared__ float s_mn[LIO_BLOCK / 64][3], s_mx[LIO_BLOCK / 64][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = s_mn[0][a], hi = s_mx[0][a];
        for (int w = 1; w < LIO_BLOCK / 64; ++w) { lo = fminf(lo, s_mn[w][a]); hi = fmaxf(hi, s_mx[w][a]); }
        float* o = box + (size_t)((base + bd.first) / LIO_BLOCK + bd.scan) * 6;
        o[a] = lo; o[3 + a] = hi;
    }
}

// Map sharding: decide the fate of whole workgroups BEFORE the association launch.  One wave per workgroup of the
// list: the workgroup's points sit in a box (lidar frame, k_block_boxes); its eight corners, transformed, bound every
// transformed point along the shard axis (an affine image of a box lies inside the hull of the mapped corners), giving
// a cell interval [c_lo, c_hi] (widened against fp32 rounding).  Every rank evaluates the same rule on the same data:
//   * WHOLE-WORKGROUP OWNERSHIP (plan.n_ranks > 0): let R be the rank owning the interval's middle cell.  If the
//     interval, plus the one cell a neighbour search can reach, lies inside R's slab extended by the plan's halo -- the
//     part of the map R holds --, R processes ALL points of the workgroup (mode 2, no per-point test) and every other
//     rank skips it (mode 1).  A point is then processed where its whole workgroup is, with every map point within
//     1 m present, so its 5-NN set is the unsharded one; no wave runs for a handful of owned lanes any more.
//   * otherwise (a workgroup longer than the halo allows, or no plan): per-point ownership by the cell of the point's
//     own position (mode 0) on the ranks whose slab the interval meets, skipped (mode 1) elsewhere.
// For a skipped workgroup this kernel reports the all-zero partial sum and the arrival in its place and drops the
// points' search bounds; k_s2m_iterate returns on the flag before loading a point.  The rule never changes a result.
struct LioShardPlan {
    int n_ranks;            // 0 = no whole-workgroup ownership (lio_s2m_set_shard)
    int rank;
    int halo;               // cells of map held beyond the slab on each side (>= 1)
    int bounds[9];          // rank r owns cells [bounds[r], bounds[r+1])
};

__global__ __launch_bounds__(256) void k_shard_cull(LioIterParams P, LioShardPlan plan, const float* __restrict__ block_box, int n_blocks,
                                                    unsigned char* __restrict__ skip)
{
    // eight lanes per workgroup of the list (one per corner of its box), eight workgroups per wave
    const int lane = threadIdx.x & 63, c = lane & 7;
    const int i = (blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * 8 + (lane >> 3);
    if (i >= n_blocks) return;
    const LioBlockDesc bd = P.blocks[i];
    const LioScanState* st = &P.state[bd.scan];
    if (st->done) { if (c == 0) skip[i] = 0; return; }
    const int base = st->offset, ax = P.shard.axis;
    const float* bb = block_box + (size_t)((base + bd.first) / LIO_BLOCK + bd.scan) * 6;
    const float bx = (c & 1) ? bb[3] : bb[0], by = (c & 2) ? bb[4] : bb[1], bz = (c & 4) ? bb[5] : bb[2];
    const float qa = st->T[ax * 4 + 0] * bx + st->T[ax * 4 + 1] * by + st->T[ax * 4 + 2] * bz + st->T[ax * 4 + 3];
    float lo = qa, hi = qa;
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); }
    const float eps = 1.0e-4f + 1.0e-5f * fmaxf(fabsf(lo), fabsf(hi));
    int c_lo = lio_cell_coord(lo - eps, P.shard.gorigin, P.shard.inv_cell, P.shard.gdim);
    int c_hi = lio_cell_coord(hi + eps, P.shard.gorigin, P.shard.inv_cell, P.shard.gdim);
    c_lo = min(max(c_lo, 0), P.shard.gdim - 1);
    c_hi = min(max(c_hi, 0), P.shard.gdim - 1);
    int mode = 0;
    if (lo <= hi) {                                         // (an empty box -- no finite point -- stays on the exact path)
        bool whole = false;
        if (plan.n_ranks > 0) {
            const int mid = (c_lo + c_hi) >> 1;
            int r = 0;
            while (r + 1 < plan.n_ranks && mid >= plan.bounds[r + 1]) ++r;
            whole = (c_lo - 1 >= plan.bounds[r] - plan.halo) && (c_hi + 1 < plan.bounds[r + 1] + plan.halo);
            if (whole) mode = (r == plan.rank) ? 2 : 1;
        }
        if (!whole && (c_hi < P.shard.lo || c_lo >= P.shard.hi)) mode = 1;
    }
    // The search bound of a point is only valid from one pass to the very next.  A workgroup that sat the LAST pass out
    // and takes part in this one drops its points' bounds here (its previous decision is still in skip[i]; pass 0 never
    // reads a bound and the host clears them before it).  Dropping them on every skipped workgroup instead cost 24 M
    // scattered stores per launch on an 8-way partition.
    const int prev_mode = skip[i];
    if (P.d5_cache && mode != 1 && prev_mode == 1 && st->iter > 0) {
        for (int j = c; j < LIO_BLOCK; j += 8)
            if (bd.first + j < st->n_pts) P.d5_cache[base + bd.first + j] = -1.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // (all eight lanes have read skip[i] before lane 0 rewrites it)
    if (c == 0) skip[i] = (unsigned char)mode;
    if (mode != 1) return;
    double* part0 = P.partials + ((size_t)bd.scan * P.max_blk + bd.blk) * LIO_SUMS;
    for (int j = c; j < 28; j += 8) __hip_atomic_store(part0 + j, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (every lane of the wave has drained its stores before any lane arrives)
    if (c == 0) {
        const unsigned old = atomicAdd(&P.arrive[bd.scan], 1u);
        // every workgroup of the scan skipped on this rank: its (all-zero) sums are already in sums_out; re-arm the counter
        if (old == (unsigned)bd.n_blk - 1u) P.arrive[bd.scan] = 0;
    }
}

// Start of a registration: transformTobeMapped <- caller's guess, transform and
// trig for the first pass, counters cleared.  matP / is_degenerate persist
// (members MO:176-177).  Scans with N_s <= min_scan_pts are skipped (MO:1844).
// from_state: the guess already sits in st[].pose (lio_s2m_register sends it with the state) and is copied to poses[],
// where a second run on the same batch finds it.
// SINGLE: the whole batch is one workgroup, which counts the scans itself -- no memset of *n_active before the launch.
template <bool SINGLE>
__global__ void k_s2m_init_state(LioScanState* __restrict__ st, int n_scans,
                                 float* __restrict__ poses, int from_state, LioConsts c, int* __restrict__ n_active)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool enough = false;
    if (s < n_scans) {
        LioScanState* p = &st[s];
        float T[12], trig[6], pose[6];
        if (from_state) { for (int k = 0; k < 6; ++k) { pose[k] = p->pose[k]; poses[s * 6 + k] = pose[k]; } }
        else { for (int k = 0; k < 6; ++k) { pose[k] = poses[s * 6 + k]; p->pose[k] = pose[k]; } }
        lio_pose_to_transform(pose, T, trig);
        for (int k = 0; k < 12; ++k) p->T[k] = T[k];
        for (int k = 0; k < 6; ++k) p->trig[k] = trig[k];
        p->iter = 0; p->converged = 0; p->n_corr_last = 0;
        for (int k = 0; k < 32; ++k) { p->n_corr_iter[k] = 0; for (int j = 0; j < 6; ++j) p->pose_iter[k][j] = 0.0f; }
        for (int k = 0; k < 36; ++k) p->AtA[k] = 0.0f;
        for (int k = 0; k < 6; ++k) p->AtB[k] = 0.0f;
        enough = p->n_pts > c.min_scan_pts;
        p->done = enough ? 0 : 1;
        p->status = enough ? 0 : 1;
    }
    if (SINGLE) {
        const int n = __syncthreads_count(enough ? 1 : 0);
        if (threadIdx.x == 0) *n_active = n;
    } else if (enough) {
        atomicAdd(n_active, 1);
    }
}

// ---- candidate scan, LDS form ---------------------------------------------
// s_pts holds the map points of the workgroup's cell region (rows of the
// region are contiguous runs), s_cell the run offsets of every cell of the
// region: rx+1 entries per (y,z) row.
LIO_DEV void lio_knn_lds(const float4* s_pts, const int* s_cell, int rx1, int ry,
                         int rx0, int ry0, int rz0, int ry1, int rz1, int nx, int k,
                         float qx, float qy, float qz, int cx, int cy, int cz, LioTop5& top)
{
    const int xa = max(cx - k, 0), xb = min(cx + k, nx - 1);
    if (xa > xb) return;
#pragma unroll 1
    for (int dz = -k; dz <= k; ++dz) {
        const int z = cz + dz;
        if (z < rz0 || z > rz1) continue;
#pragma unroll 1
        for (int dy = -k; dy <= k; ++dy) {
            const int y = cy + dy;
            if (y < ry0 || y > ry1) continue;
            const int r = (z - rz0) * ry + (y - ry0);
            const int b = s_cell[r * rx1 + (xa - rx0)];
            const int e = s_cell[r * rx1 + (xb + 1 - rx0)];
            for (int j = b; j < e; ++j) {
                const float4 m = s_pts[j];
                const float d2 = lio_sqdist(m.x, m.y, m.z, qx, qy, qz);
                lio_top5_insert(top, lio_make_key(d2, __float_as_int(m.w)));
            }
        }
    }
}

#ifndef LIO_MIN_WAVES
#define LIO_MIN_WAVES 5      // waves per SIMD asked of the register allocator for the default instantiation (<= 96 VGPRs, no scratch;
                             // measured: 4 -> 5 is -6 % time, 6 spills and is 20 % slower); the other instantiations keep 4
#endif
#define LIO_LDS_PTS   2048     // staged map points per workgroup (32 KiB)
#define LIO_LDS_CELLS 4096     // staged run offsets (16 KiB)
#define LIO_LDS_ROWS  256      // (y,z) rows of a region (one thread each)

// One thread = one scan point (x PPT points, strided by the workgroup size).
// STAGE: the workgroup's points are spatially sorted at upload, so their
// 27-cell neighbourhoods overlap heavily: the union region of the map is staged
// through LDS once (coalesced 16-byte loads) and every lane scans its
// candidates from LDS.  Regions that do not fit fall back to the global form;
// both forms visit the same candidate set, so results are identical.
// CORNER (extension): the same workgroup structure over the scan's EDGE points against
// the corner map with the point-to-line association of upstream LIO-SAM; its rows join the same
// per-scan sums (combineOptimizationCoeffs) through the partials of chunks n_surf_chunks.. .
template <int PPT, bool STAGE, bool CORNER>
__global__ __launch_bounds__(LIO_BLOCK, (PPT == 1 && !STAGE && !CORNER) ? LIO_MIN_WAVES : 4) void k_s2m_iterate(LioIterParams P)
{
    __shared__ __attribute__((aligned(16))) double s_rows[LIO_BLOCK][8];  // [arz ary arx cx cy cz | -cw | accepted], widened once
    __shared__ double s_part[8][28];
    __shared__ double s_sum[28];
    __shared__ LioSolveWs s_ws;
    __shared__ __attribute__((aligned(16))) float4 s_pts[STAGE ? LIO_LDS_PTS : 1];
    __shared__ int s_cell[STAGE ? LIO_LDS_CELLS : 1];
    __shared__ int s_row_beg[STAGE ? LIO_LDS_ROWS : 1];
    __shared__ int s_row_off[STAGE ? LIO_LDS_ROWS + 1 : 1];
    __shared__ int s_box[8];
    __shared__ int s_scan4[4];

    // XCD-aware order (speed only): consecutive workgroup ids are dealt round-robin over the 8
    // XCDs, each with a private L2.  Give every XCD one contiguous eighth of the scan-major,
    // tile-sorted work list so that the map rows it streams stay in ITS L2.
    int wg = blockIdx.x;
    if (P.xcd_remap) {
        const int n8 = gridDim.x >> 3;                     // full groups of 8
        if (wg < n8 * 8) wg = (wg & 7) * n8 + (wg >> 3);
    }
    // map sharding, decided per workgroup by k_shard_cull: 1 = it has already reported for this workgroup (not ours),
    // 2 = the WHOLE workgroup is ours (no per-point ownership test), 0 = per-point ownership
    const int wg_mode = P.blk_skip != nullptr ? (int)P.blk_skip[wg] : 0;
    if (wg_mode == 1) return;
    const LioBlockDesc bd = P.blocks[wg];
    LioScanState* st = &P.state[bd.scan];
    if (st->done) return;                                  // workgroup-uniform
    // diagnostic phase clock (P.stamps is null outside profiling experiments)
    long long* stamp = P.stamps ? P.stamps + ((size_t)wg * (LIO_BLOCK / 64) + (threadIdx.x >> 6)) * 8 : nullptr;
#define LIO_STAMP(k) do { if (stamp && (threadIdx.x & 63) == 0) stamp[k] = (long long)__builtin_readcyclecounter(); } while (0)
    LIO_STAMP(0);

    // wave-uniform per-scan values
    float T[12], tr[6];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = st->T[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) tr[k] = st->trig[k];
    const int n_pts = CORNER ? st->c_n_pts : st->n_pts;
    const int base = CORNER ? st->c_offset : st->offset;
    const bool record = (P.rec_flag != nullptr) && (st->iter == P.c.record_iter);
    const bool use_cache = (P.d5_cache != nullptr) && !STAGE && st->iter > 0;   // iteration 0 has nothing to re-use
    float Tp[12];                                          // the transform of the previous iteration
#pragma unroll
    for (int k = 0; k < 12; ++k) Tp[k] = use_cache ? st->Tp[k] : 0.0f;
    const LioGrid g = P.grid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // ---- phase A: transform (pointAssociateToMap, MO:841-847), cells, ownership
    float px[PPT], py[PPT], pz[PPT], qx[PPT], qy[PPT], qz[PPT];
    int cx[PPT], cy[PPT], cz[PPT];
    bool act[PPT], inr[PPT];
    int bmn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, bmx[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
#pragma unroll
    for (int pp = 0; pp < PPT; ++pp) {
        const int li = bd.first + pp * LIO_BLOCK + (int)threadIdx.x;
        inr[pp] = li < n_pts;
        const int gi = base + (inr[pp] ? li : 0);
        px[pp] = P.sx[gi]; py[pp] = P.sy[gi]; pz[pp] = P.sz[gi];     // coalesced SoA
        qx[pp] = T[0] * px[pp] + T[1] * py[pp] + T[2]  * pz[pp] + T[3];
        qy[pp] = T[4] * px[pp] + T[5] * py[pp] + T[6]  * pz[pp] + T[7];
        qz[pp] = T[8] * px[pp] + T[9] * py[pp] + T[10] * pz[pp] + T[11];
        bool a = inr[pp];
        if (P.shard.axis >= 0 && wg_mode != 2) {                     // owner-computes (multi-GPU)
            const float qa = P.shard.axis == 0 ? qx[pp] : (P.shard.axis == 1 ? qy[pp] : qz[pp]);
            int gc = lio_cell_coord(qa, P.shard.gorigin, P.shard.inv_cell, P.shard.gdim);
            gc = min(max(gc, 0), P.shard.gdim - 1);
            a = a && gc >= P.shard.lo && gc < P.shard.hi;
        }
        cx[pp] = lio_cell_coord(qx[pp], g.ox, g.inv_cell, g.nx);
        cy[pp] = lio_cell_coord(qy[pp], g.oy, g.inv_cell, g.ny);
        cz[pp] = lio_cell_coord(qz[pp], g.oz, g.inv_cell, g.nz);
        // a point whose 27 cells all lie outside the grid has no candidates at all; a non-finite point has no
        // neighbours either (and would put NaN into the distance keys)
        a = a && (fabsf(qx[pp]) <= 3.0e38f) && (fabsf(qy[pp]) <= 3.0e38f) && (fabsf(qz[pp]) <= 3.0e38f);
        a = a && cx[pp] >= -g.k && cx[pp] < g.nx + g.k && cy[pp] >= -g.k && cy[pp] < g.ny + g.k &&
            cz[pp] >= -g.k && cz[pp] < g.nz + g.k;
        act[pp] = a;
        if (STAGE && a) {
            bmn[0] = min(bmn[0], cx[pp]); bmx[0] = max(bmx[0], cx[pp]);
            bmn[1] = min(bmn[1], cy[pp]); bmx[1] = max(bmx[1], cy[pp]);
            bmn[2] = min(bmn[2], cz[pp]); bmx[2] = max(bmx[2], cz[pp]);
        }
    }

    // A workgroup none of whose points is active (typically: owned by other ranks) only reports
    // an all-zero partial sum and leaves.
    if (P.shard.axis >= 0) {
        bool any = false;
#pragma unroll
        for (int pp = 0; pp < PPT; ++pp) any = any || act[pp];
        if (!__syncthreads_or(any ? 1 : 0)) {
            // the search bound of a point is only valid from one pass to the very next: drop it for points that sit this pass out
            if (P.d5_cache) {
#pragma unroll
                for (int pp = 0; pp < PPT; ++pp)
                    if (inr[pp]) P.d5_cache[base + bd.first + pp * LIO_BLOCK + (int)threadIdx.x] = -1.0f;
            }
            if (wave != 0) return;
            double* part0 = P.partials + ((size_t)bd.scan * P.max_blk + bd.blk) * LIO_SUMS;
            if (lane < 28) __hip_atomic_store(part0 + lane, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lio_arrive_and_finish(P, bd, st, lane, s_sum, &s_ws, stamp);   // (recorded flags stay "rejected")
            return;
        }
    }

    // ---- phase B: stage the union cell region of the workgroup through LDS
    bool staged = false;
    int rx0 = 0, ry0 = 0, rz0 = 0, rx1c = 0, ry1 = -1, rz1 = -1, rxn1 = 1, ryn = 1;
    if (STAGE) {
        if (threadIdx.x < 3) { s_box[threadIdx.x] = 0x7fffffff; s_box[3 + threadIdx.x] = -0x7fffffff; }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                bmn[a] = min(bmn[a], __shfl_xor(bmn[a], off));
                bmx[a] = max(bmx[a], __shfl_xor(bmx[a], off));
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { atomicMin(&s_box[a], bmn[a]); atomicMax(&s_box[3 + a], bmx[a]); }
        }
        __syncthreads();
        rx0 = max(s_box[0] - g.k, 0); rx1c = min(s_box[3] + g.k, g.nx - 1);
        ry0 = max(s_box[1] - g.k, 0); ry1 = min(s_box[4] + g.k, g.ny - 1);
        rz0 = max(s_box[2] - g.k, 0); rz1 = min(s_box[5] + g.k, g.nz - 1);
        const bool any = s_box[0] != 0x7fffffff && rx0 <= rx1c && ry0 <= ry1 && rz0 <= rz1;
        rxn1 = rx1c - rx0 + 2;                      // run offsets per row (cells + 1)
        ryn = ry1 - ry0 + 1;
        const int rzn = rz1 - rz0 + 1;
        const int n_rows = ryn * rzn;
        bool fits = any && n_rows <= LIO_LDS_ROWS && n_rows * rxn1 <= LIO_LDS_CELLS;
        if (fits) {                                  // workgroup-uniform
            // run of every (y,z) row of the region, exclusive scan of the run lengths
            int cnt = 0;
            if ((int)threadIdx.x < n_rows) {
                const int z = rz0 + (int)threadIdx.x / ryn, y = ry0 + (int)threadIdx.x % ryn;
                const int row = (z * g.ny + y) * g.nx;
                const int b = P.cell_start[row + rx0];
                cnt = P.cell_start[row + rx1c + 1] - b;
                s_row_beg[threadIdx.x] = b;
            }
            int total;
            const int off = lio_block_exclusive_scan(cnt, &total, s_scan4);
            if ((int)threadIdx.x < n_rows) s_row_off[threadIdx.x] = off;
            if ((int)threadIdx.x == n_rows) s_row_off[n_rows] = total;
            fits = total <= LIO_LDS_PTS;
            __syncthreads();
            if (fits) {
                for (int i = threadIdx.x; i < n_rows * rxn1; i += LIO_BLOCK) {
                    const int r = i / rxn1, xi = i - r * rxn1;
                    const int z = rz0 + r / ryn, y = ry0 + r % ryn;
                    const int row = (z * g.ny + y) * g.nx;
                    s_cell[i] = s_row_off[r] + (P.cell_start[row + rx0 + xi] - s_row_beg[r]);
                }
                for (int i = threadIdx.x; i < total; i += LIO_BLOCK) {
                    // row of staged slot i: last r with s_row_off[r] <= i
                    int lo = 0, hi = n_rows - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (s_row_off[mid] <= i) lo = mid; else hi = mid - 1;
                    }
                    s_pts[i] = P.map_sorted[s_row_beg[lo] + (i - s_row_off[lo])];
                }
                staged = true;
            }
            __syncthreads();
        }
    }

    // Normal equations (matAtA = matAt*matA, matAtB = matAt*matB, MO:1781-1783) as
    // a transposed reduction: every lane parks its Jacobian row in LDS, then lane
    // (g, s) = (tid / 32, tid % 32) accumulates sum s over the points p = g (mod 8)
    // in fp64.  One accumulator per lane instead of 28, no cross-lane shuffles,
    // fixed summation order.
    const int red_g = threadIdx.x >> 5, red_s = threadIdx.x & 31;
    const int red_a = c_pair_a[red_s], red_b = c_pair_b[red_s];
    double red_acc = 0.0;
    LIO_STAMP(1);

#pragma unroll 1
    for (int pp = 0; pp < PPT; ++pp) {
        // ---- exact 5-NN over the 27-cell neighbourhood (MO:1631) ----
        // Search bound from the previous iteration (speed only).  The 5 neighbours found last time lie within
        // sqrt(d5_prev) of the point's previous position q_prev, hence within R = sqrt(d5_prev) + |q - q_prev| of
        // its new position q: the new 5th distance cannot exceed R, every member of the new 5-NN set lies
        // inside the x-cells [cell(qx - R), cell(qx + R)], and candidates beyond R^2 can be turned away by
        // the sentinel.  R is rounded up by 1e-4 (fp32 rounding of the distances is 1e-7).  Typically
        // R ~ 0.5 m against the 1 m gate: half the candidate run.  One float per point is kept.
        float bound2 = P.c.max_sq_dist;
        float Rx = sqrtf(P.c.max_sq_dist) * 1.0001f + 1e-6f;          // reach along x: the gate, unless the bound below is tighter
        bool bounded = false;
        const int ci = base + bd.first + pp * LIO_BLOCK + (int)threadIdx.x;   // slot in the batch SoA
        if (use_cache && act[pp]) {
            const float d5 = P.d5_cache[ci];
            if (d5 >= 0.0f) {
                const float ox = Tp[0] * px[pp] + Tp[1] * py[pp] + Tp[2]  * pz[pp] + Tp[3];
                const float oy = Tp[4] * px[pp] + Tp[5] * py[pp] + Tp[6]  * pz[pp] + Tp[7];
                const float oz = Tp[8] * px[pp] + Tp[9] * py[pp] + Tp[10] * pz[pp] + Tp[11];
                const float mv = sqrtf(lio_sqdist(qx[pp], qy[pp], qz[pp], ox, oy, oz));
                const float R = (sqrtf(d5) + mv) * 1.0001f + 1e-6f;
                const float r2 = R * R * 1.0001f;
                if (r2 < bound2) { bound2 = r2; Rx = R; bounded = true; }
            }
        }
        // (d2 == bound2 with any real index sorts below the sentinel, so ties at the bound are kept)
        const double sentinel = lio_make_key(bound2, -1);                 // index 0xffffffff: above every real index
        LioTop5 top = { sentinel, sentinel, sentinel, sentinel, sentinel };
        if (act[pp]) {
            if (STAGE && staged)
                lio_knn_lds(s_pts, s_cell, rxn1, ryn, rx0, ry0, rz0, ry1, rz1, g.nx, g.k,
                            qx[pp], qy[pp], qz[pp], cx[pp], cy[pp], cz[pp], top);
            else
                lio_knn_global(P, g, qx[pp], qy[pp], qz[pp], cy[pp], cz[pp], Rx, bound2, bounded, top);
        }
        int nn[5] = { lio_key_idx(top.k0), lio_key_idx(top.k1), lio_key_idx(top.k2), lio_key_idx(top.k3), lio_key_idx(top.k4) };
        const float d2_5 = lio_key_d2(top.k4);
        // gate MO:1641: pointSearchSqDis[4] < 1.0
        const bool ok = act[pp] && (d2_5 < P.c.max_sq_dist);
        if (pp == 0) LIO_STAMP(2);

        if (P.d5_cache && inr[pp])                                    // for the next iteration (-1: nothing to re-use)
            P.d5_cache[ci] = ok ? d2_5 : -1.0f;
        float cxx = 0.0f, cyy = 0.0f, czz = 0.0f, cww = 0.0f;
        bool accept = false;
        // plane through the five neighbours, plane test, weight, coefficients MO:1642-1683 (the CORNER extension: point-to-line);
        // the same function the one-launch loop calls (lio_s2m_device.h): one copy of the arithmetic
        if (ok) accept = lio_assoc_point<CORNER>(P, nn, qx[pp], qy[pp], qz[pp], px[pp], py[pp], pz[pp], cxx, cyy, czz, cww);
        if (record && inr[pp]) {
            // the record is kept in the CALLER's point order
            const int li = bd.first + pp * LIO_BLOCK + (int)threadIdx.x;
            const int oi = P.perm ? P.perm[base + li] : base + li;
            P.rec_flag[oi] = accept ? 1 : 0;
            reinterpret_cast<float4*>(P.rec_coeff)[oi] = make_float4(cxx, cyy, czz, cww);
#pragma unroll
            for (int j = 0; j < 5; ++j) P.rec_nn[(size_t)oi * 5 + j] = ok ? nn[j] : -1;
        }
        // MO:1735-1778: row of matA / matB (zero row when the point is rejected)
        float row[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, rhs = 0.0f;
        if (accept) lio_jacobian_row(tr, px[pp], py[pp], pz[pp], cxx, cyy, czz, cww, P.c.jac_exact, row, rhs);
        if (PPT > 1) __syncthreads();                                  // previous pass finished reading
        {   // each value is converted to fp64 once here instead of once per product below
            double2* dst = reinterpret_cast<double2*>(s_rows[threadIdx.x]);
            dst[0] = make_double2((double)row[0], (double)row[1]);
            dst[1] = make_double2((double)row[2], (double)row[3]);
            dst[2] = make_double2((double)row[4], (double)row[5]);
            dst[3] = make_double2((double)rhs, accept ? 1.0 : 0.0);
        }
        if (pp == 0) LIO_STAMP(3);
        __syncthreads();
        if (red_s < 28) {
#pragma unroll 8
            for (int p = red_g; p < LIO_BLOCK; p += 8)
                red_acc = __builtin_fma(s_rows[p][red_a], s_rows[p][red_b], red_acc);   // the product of two widened fp32 is exact: fma == mul, add
        }
    }
    LIO_STAMP(4);
    if (red_s < 28) s_part[red_g][red_s] = red_acc;
    __syncthreads();
    LIO_STAMP(5);
    if (wave != 0) return;

    double* part = P.partials + ((size_t)bd.scan * P.max_blk + bd.blk) * LIO_SUMS;
    if (lane < 28) {
        double v = s_part[0][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) v += s_part[w][lane];
        // write-through (sc1) store: visible to the other XCDs without a release fence
        __hip_atomic_store(part + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lio_arrive_and_finish(P, bd, st, lane, s_sum, &s_ws, stamp);
    LIO_STAMP(7);
#undef LIO_STAMP
}

// Compact per-scan summary for callers that only want the poses: [pose x6 | iter | status | converged | is_degenerate]
__global__ void k_s2m_pack_summary(const LioScanState* __restrict__ st, int n_scans, float* __restrict__ out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_scans) return;
    float* o = out + (size_t)s * 10;
    for (int k = 0; k < 6; ++k) o[k] = st[s].pose[k];
    o[6] = __int_as_float(st[s].iter); o[7] = __int_as_float(st[s].status | (st[s].done ? 0 : 0x100));   // (bit 8: still iterating)
    o[8] = __int_as_float(st[s].converged); o[9] = __int_as_float(st[s].is_degenerate);
}

// Sharded mode: solve every scan from all-reduced sums (one wave per scan).  n_slots > 1: the sums arrive as one partial
// table per device (slot d at sums + d * slot_stride, in-library multi-GPU mode) and are added here in slot order --
// the same order on every device, so every device computes bit-identical poses.
__global__ void k_s2m_apply(LioScanState* __restrict__ st, int n_scans, const double* __restrict__ sums, size_t slot_stride, int n_slots,
                            LioConsts c, int* __restrict__ n_active)
{
    const int s = blockIdx.x;                                // one wave per scan
    if (s >= n_scans) return;
    if (st[s].done) return;
    __shared__ LioSolveWs s_ws;
    __shared__ double s_tot[LIO_SUMS];
    const int lane = (int)threadIdx.x;
    if (lane < LIO_SUMS) {
        const double* p = sums + (size_t)s * LIO_SUMS + lane;
        double v = p[0];
        for (int d = 1; d < n_slots; ++d) v += p[(size_t)d * slot_stride];
        s_tot[lane] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // one wave: the sums are in LDS before any lane reads them
    lio_gn_step(&st[s], s_tot, c, &s_ws, n_active, lane);
}

// ------------------------------------------------------------ launch glue
void lio_launch_aos_to_soa(const void* src, size_t stride, int n, float* x, float* y, float* z,
                           float4* xyz4, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_aos_to_soa, dim3((n + 255) / 256), dim3(256), 0, s,
                       (const unsigned char*)src, stride, n, x, y, z, xyz4);
}

void lio_launch_map_bbox(const float* x, const float* y, const float* z, int n, unsigned* bbox, hipStream_t s)
{
    int blocks = (n + 1023) / 1024;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_map_bbox, dim3(blocks), dim3(256), 0, s, x, y, z, n, bbox);
}

void lio_launch_exclusive_scan(const int* in, int n, int* tile_sums, int* out, hipStream_t s)
{
    const int n_tiles = (n + LIO_SCAN_TILE - 1) / LIO_SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(256), 0, s, in, n, tile_sums);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(256), 0, s, tile_sums, n_tiles);
    hipLaunchKernelGGL(k_scan_apply, dim3(n_tiles), dim3(256), 0, s, in, n, tile_sums, out);
}

// cell_count: 2 x n_cells ints (point counts, then neighbourhood-row lengths; both reused as fill cursors);
// tile_sums: 2 x (lio_scan_tiles(n_cells) + 1) ints, 8-byte aligned (one 64-bit pair sum per tile)
// nbr_slot: n x (2k+1)^2 ints
void lio_launch_map_occupancy(const LioGrid& g, const float* x, const float* y, const float* z, int n, int* flags, hipStream_t s)
{
    (void)hipMemsetAsync(flags, 0, sizeof(int) * ((size_t)g.n_cells + 1), s);
    hipLaunchKernelGGL(k_map_occupancy, dim3((n + 255) / 256), dim3(256), 0, s, g, x, y, z, n, flags);
}

static LioGrid lio_tight_grid(const LioGrid& g, int l)
{
    LioGrid b = g;                      // same x buckets; its own (y, z) cells, k = 1
    b.oy = g.tb_oy[l]; b.oz = g.tb_oz[l]; b.inv_cell = g.tb_inv_cell[l]; b.ny = g.tb_ny[l]; b.nz = g.tb_nz[l]; b.k = 1;
    return b;
}

void lio_launch_map_build(const LioGrid& g, const float* x, const float* y, const float* z, int n,
                          int* cell_of, int* cell_count, int* cell_start, int* tile_sums,
                          float4* sorted, int* nbr_start, float4* nbr_pts, int* nbr_slot, bool with_cell_sorted, hipStream_t s)
{
    const int nb = (n + 255) / 256;
    const int repsA = (2 * g.k + 1) * (2 * g.k + 1), repsB = 9;
    int n_tb = 0;
    while (n_tb < LIO_TB_MAX && g.tb_reach[n_tb] > 0.0f) ++n_tb;
    const int lenA = g.n_cells * g.xs;                                     // buckets of the grid's own rows
    int len_all = lenA, rows_all = g.ny * g.nz;
    for (int l = 0; l < n_tb; ++l) { len_all += g.tb_ny[l] * g.tb_nz[l] * g.nxf; rows_all += g.tb_ny[l] * g.tb_nz[l]; }
    const unsigned nbrA = (unsigned)(((long long)n * repsA + 255) / 256), nbrB = (unsigned)(((long long)n * repsB + 255) / 256);
    int* nbr_count = cell_count + g.n_cells;
    unsigned long long* tiles64 = reinterpret_cast<unsigned long long*>(tile_sums);
    // with_cell_sorted: also the cell-sorted 1x copy + cell_start the LDS-staged variant (cfg.use_lds) walks; the default
    // candidate scan only needs the replicated rows, and a node rebuilds this for every scan: nothing it does not read
    if (with_cell_sorted) (void)hipMemsetAsync(cell_count, 0, sizeof(int) * ((size_t)g.n_cells + len_all), s);
    else (void)hipMemsetAsync(nbr_count, 0, sizeof(int) * (size_t)len_all, s);
    hipLaunchKernelGGL(k_map_cell_count, dim3(nb), dim3(256), 0, s, g, x, y, z, n, cell_of, with_cell_sorted ? cell_count : (int*)nullptr);
    hipLaunchKernelGGL(k_map_nbr_count, dim3(nbrA), dim3(256), 0, s, g, cell_of, x, y, z, n, nbr_count, nbr_slot);
    hipLaunchKernelGGL(k_map_nbr_pad_rows, dim3((g.ny * g.nz + 3) / 4), dim3(256), 0, s, g, nbr_count);
    for (int l = 0; l < n_tb; ++l) {
        const LioGrid gb = lio_tight_grid(g, l);
        hipLaunchKernelGGL(k_map_nbr_count, dim3(nbrB), dim3(256), 0, s, gb, cell_of, x, y, z, n, nbr_count + g.tb_row0[l],
                           nbr_slot + (size_t)n * (repsA + repsB * l));
        hipLaunchKernelGGL(k_map_nbr_pad_rows, dim3((gb.ny * gb.nz + 3) / 4), dim3(256), 0, s, gb, nbr_count + g.tb_row0[l]);
    }
    // one exclusive scan over the buckets of all tables: the tight rows' records follow the grid's own in nbr_pts
    if (with_cell_sorted && g.xs == 1 && n_tb == 0) {
        // (one pass carrying both sums while the two tables have the same length)
        lio_launch_scan2<false>(cell_count, nbr_count, g.n_cells, tiles64, cell_start, nbr_start, s);
    } else {
        if (with_cell_sorted) lio_launch_exclusive_scan(cell_count, g.n_cells, tile_sums, cell_start, s);
        lio_launch_exclusive_scan(nbr_count, len_all, tile_sums, nbr_start, s);
    }
    if (with_cell_sorted) {
        (void)hipMemsetAsync(cell_count, 0, sizeof(int) * (size_t)g.n_cells, s);       // reused as the fill cursor of the cell-sorted copy
        hipLaunchKernelGGL(k_map_scatter, dim3(nb), dim3(256), 0, s, x, y, z, n, cell_of, cell_start, cell_count, sorted);
    }
    {
        const int n_rec4 = n * (repsA + repsB * n_tb) + LIO_ROW_ALIGN * rows_all + 2 * LIO_ROW_ALIGN;   // + row and tail padding
        hipLaunchKernelGGL(k_map_nbr_fill, dim3((n_rec4 + 255) / 256), dim3(256), 0, s, nbr_pts, n_rec4);
    }
    hipLaunchKernelGGL(k_map_nbr_scatter, dim3(nbrA), dim3(256), 0, s, g, x, y, z, n, nbr_start, nbr_slot, nbr_pts);
    for (int l = 0; l < n_tb; ++l)
        hipLaunchKernelGGL(k_map_nbr_scatter, dim3(nbrB), dim3(256), 0, s, lio_tight_grid(g, l), x, y, z, n, nbr_start + g.tb_row0[l],
                           nbr_slot + (size_t)n * (repsA + repsB * l), nbr_pts);
}

int lio_scan_tiles(int n_cells) { return (n_cells + LIO_SCAN_TILE - 1) / LIO_SCAN_TILE; }

void lio_launch_init_state(LioScanState* st, int n_scans, float* poses, bool from_state, const LioConsts& c,
                           int* n_active, hipStream_t s)
{
    const int fs = from_state ? 1 : 0;
    if (n_scans <= 256) {              // one workgroup counts the active scans itself: one stream operation instead of two
        hipLaunchKernelGGL(k_s2m_init_state<true>, dim3(1), dim3(n_scans <= 64 ? 64 : 256), 0, s, st, n_scans, poses, fs, c, n_active);
        return;
    }
    (void)hipMemsetAsync(n_active, 0, sizeof(int), s);
    hipLaunchKernelGGL(k_s2m_init_state<false>, dim3((n_scans + 63) / 64), dim3(64), 0, s, st, n_scans, poses, fs, c, n_active);
}

void lio_launch_iterate(const LioIterParams& P, int n_blocks, int ppt, bool stage, hipStream_t s, bool corner)
{
    if (n_blocks <= 0) return;
    const dim3 gr(n_blocks), bl(LIO_BLOCK);
    if (corner) {                      // extension: edge points, always one point per thread from global memory
        hipLaunchKernelGGL((k_s2m_iterate<1, false, true>), gr, bl, 0, s, P);
        return;
    }
    if (stage) {
        switch (ppt) {
        case 1: hipLaunchKernelGGL((k_s2m_iterate<1, true, false>), gr, bl, 0, s, P); break;
        case 2: hipLaunchKernelGGL((k_s2m_iterate<2, true, false>), gr, bl, 0, s, P); break;
        default: hipLaunchKernelGGL((k_s2m_iterate<4, true, false>), gr, bl, 0, s, P); break;
        }
    } else {
        switch (ppt) {
        case 1: hipLaunchKernelGGL((k_s2m_iterate<1, false, false>), gr, bl, 0, s, P); break;
        case 2: hipLaunchKernelGGL((k_s2m_iterate<2, false, false>), gr, bl, 0, s, P); break;
        default: hipLaunchKernelGGL((k_s2m_iterate<4, false, false>), gr, bl, 0, s, P); break;
        }
    }
}

void lio_launch_pack_summary(const LioScanState* st, int n_scans, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_s2m_pack_summary, dim3((n_scans + 255) / 256), dim3(256), 0, s, st, n_scans, out);
}

void lio_launch_apply(LioScanState* st, int n_scans, const double* sums, size_t slot_stride, int n_slots, const LioConsts& c,
                      int* n_active, hipStream_t s)
{
    hipLaunchKernelGGL(k_s2m_apply, dim3(n_scans), dim3(64), 0, s, st, n_scans, sums, slot_stride, n_slots, c, n_active);
}

void lio_launch_shard_cull(const LioIterParams& P, int n_ranks, int rank, int halo, const int* bounds, const float* block_box,
                           int n_blocks, unsigned char* skip, hipStream_t s)
{
    if (n_blocks <= 0) return;
    LioShardPlan plan;
    plan.n_ranks = n_ranks; plan.rank = rank; plan.halo = halo;
    for (int r = 0; r < 9; ++r) plan.bounds[r] = (n_ranks > 0 && r <= n_ranks) ? bounds[r] : 0;
    hipLaunchKernelGGL(k_shard_cull, dim3((n_blocks + 31) / 32), dim3(256), 0, s, P, plan, block_box, n_blocks, skip);
}

void lio_launch_block_boxes(const LioBlockDesc* blocks, int n_blocks, const LioScanState* st, const float* sx, const float* sy,
                            const float* sz, float* box, hipStream_t s)
{
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(k_block_boxes, dim3(n_blocks), dim3(LIO_BLOCK), 0, s, blocks, st, sx, sy, sz, box);
}

void lio_launch_scan_sort_lds(const void* stage, size_t stride, const LioScanState* st, int n_scans, int max_pts, float tile0,
                              int shard_axis, int* perm, float* x, float* y, float* z, hipStream_t s)
{
    if (n_scans <= 0) return;
    if (max_pts <= 16 * LIO_SORT_THREADS)
        hipLaunchKernelGGL(k_scan_sort_radix<16>, dim3(n_scans), dim3(LIO_SORT_THREADS), 0, s,
                           (const unsigned char*)stage, stride, st, tile0, shard_axis, perm, x, y, z);
    else                                                    // (the caller sends scans of at most 16383 points this way)
        hipLaunchKernelGGL(k_scan_sort_radix<32>, dim3(n_scans), dim3(LIO_SORT_THREADS), 0, s,
                           (const unsigned char*)stage, stride, st, tile0, shard_axis, perm, x, y, z);
}

void lio_launch_scan_bbox(const void* stage, size_t stride, const LioBlockDesc* prep_blocks, int n_prep_blocks,
                          const LioScanState* st, unsigned* bbox, hipStream_t s)
{
    if (n_prep_blocks <= 0) return;
    hipLaunchKernelGGL(k_scan_bbox, dim3(n_prep_blocks), dim3(256), 0, s, (const unsigned char*)stage, stride, prep_blocks, st, bbox);
}

void lio_launch_scan_tile_sort(const void* stage, size_t stride, int total_pts,
                               const LioBlockDesc* prep_blocks, int n_prep_blocks,
                               const LioScanState* st, const LioScanTiles* tiles, int n_keys,
                               int* key_of, int* key_count, int* key_start, int* tile_sums,
                               int* tmp_idx, int* perm, int* big_list, int* big_cnt,
                               float* x, float* y, float* z, hipStream_t s)
{
    if (total_pts <= 0 || n_prep_blocks <= 0) return;
    (void)hipMemsetAsync(key_count, 0, sizeof(int) * (size_t)n_keys, s);
    (void)hipMemsetAsync(big_cnt, 0, sizeof(int), s);
    hipLaunchKernelGGL(k_scan_tile_keys, dim3(n_prep_blocks), dim3(256), 0, s,
                       (const unsigned char*)stage, stride, prep_blocks, st, tiles, key_of, key_count);
    const int n_tiles = (n_keys + LIO_SCAN_TILE - 1) / LIO_SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(256), 0, s, key_count, n_keys, tile_sums);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(256), 0, s, tile_sums, n_tiles);
    hipLaunchKernelGGL(k_scan_apply, dim3(n_tiles), dim3(256), 0, s, key_count, n_keys, tile_sums, key_start);
    (void)hipMemsetAsync(key_count, 0, sizeof(int) * (size_t)n_keys, s);
    const int nb = (total_pts + 255) / 256;
    hipLaunchKernelGGL(k_scan_tile_scatter, dim3(nb), dim3(256), 0, s, key_of, total_pts, key_start, key_count, tmp_idx);
    hipLaunchKernelGGL(k_scan_tile_ranksort, dim3(nb), dim3(256), 0, s, key_start, key_of, total_pts, tmp_idx, perm, big_list, big_cnt);
    hipLaunchKernelGGL(k_scan_tile_bigsort, dim3(64), dim3(256), 0, s, key_start, big_list, big_cnt, perm);
    hipLaunchKernelGGL(k_scan_gather_sorted, dim3(nb), dim3(256), 0, s, (const unsigned char*)stage, stride,
                       total_pts, perm, x, y, z);
}

void lio_launch_xyzi4_to_soa(const float4* src, int n, float* x, float* y, float* z, float4* xyz4, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_xyzi4_to_soa, dim3((n + 255) / 256), dim3(256), 0, s, src, n, x, y, z, xyz4);
}
