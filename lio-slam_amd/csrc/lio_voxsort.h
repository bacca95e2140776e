// lio_voxsort.h -- K7 (pcl::VoxelGrid centroid filter, MO:1605-1611 / MO:1581-1583) by SORTING: a stable LSD radix sort of
// (voxel key, input index) pairs through LDS, then one segmented in-order fp32 sum per voxel.
//
// Why a sort (rounds 1-2 used a counting sort): a counting sort keeps three arrays over the whole KEY space (count / start /
// rank, one int per voxel of the cloud's bounding box, occupied or not).  That is ~1 M keys for a
// 200-keyframe local map at leaf 0.5 m -- fine -- but a raw 100 m sweep at the reference's smaller scan leaves
// (mappingSurfLeafSize 0.2 in jeep.yaml:99 / m1.yaml:88, 0.15 in lio_sam_livox.yaml:56) spans 10^8..10^9 voxels of which
// ~20 k are occupied: gigabytes of memset and scan per callback, and nothing at all above 2^29 voxels although PCL filters
// up to 2^31.  The sort never looks at the key space: its cost follows the number of POINTS (8 B per point and pass), it
// handles every key PCL handles, and because an LSD radix sort is stable and the pairs start in input order, every
// voxel's points arrive in ascending input index -- the order pcl::VoxelGrid's sorted index vector visits them in, which
// the fp32 running sum depends on -- with no per-voxel sort at all.  Output: ascending voxel key.  Measured faster than the counting
// sort on every input (profiles/r03_k7_forms.txt), which was then removed.
//
// One pass = three launches: per-workgroup digit histograms (LDS atomics), an exclusive scan of every digit's row of the
// [digit][workgroup] table, and the scatter (which adds the prefix over the digits itself).  A workgroup's tile is laid out wave-striped (item k of lane l of wave w = tile + w*64*ITEMS +
// k*64 + l: coalesced 8-byte loads AND ascending input order along (w, k, l)); the rank of an item among the tile's items
// of the same digit is found with eight ballots (the lanes holding the same digit), a per-wave LDS counter per digit and
// a prefix over the four waves -- no atomics, no sort inside the tile, deterministic.
#pragma once
#include <hip/hip_runtime.h>

#define LIO_VS_THREADS 256
#define LIO_VS_BINS 256

struct LioVsGrid { float inv; int min_b0, min_b1, min_b2, mul1, mul2; };

// pairs[i] = (voxel key of point i, i): the voxel index of pcl::VoxelGrid, x-fastest over the cloud's own bounding box
__global__ __launch_bounds__(256) void k_vsort_keys(LioVsGrid g, const float4* __restrict__ p, int n, uint2* __restrict__ pairs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = p[i];
    const int i0 = (int)(floorf(v.x * g.inv) - (float)g.min_b0);
    const int i1 = (int)(floorf(v.y * g.inv) - (float)g.min_b1);
    const int i2 = (int)(floorf(v.z * g.inv) - (float)g.min_b2);
    pairs[i] = make_uint2((unsigned)(i0 + i1 * g.mul1 + i2 * g.mul2), (unsigned)i);
}

// hist[d * n_blocks + block] = items of this workgroup's tile whose digit is d
template <int ITEMS>
__global__ __launch_bounds__(LIO_VS_THREADS) void k_vsort_hist(const uint2* __restrict__ in, int n, int shift, unsigned mask,
                                                               int* __restrict__ hist, int n_blocks)
{
    __shared__ int s_hist[LIO_VS_BINS];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * (LIO_VS_THREADS * ITEMS) + (long long)wave * (64 * ITEMS);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const long long i = tile + k * 64 + lane;
        if (i < n) atomicAdd(&s_hist[(in[i].x >> shift) & mask], 1);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = s_hist[threadIdx.x];
}

// Exclusive scan of every digit's row of the [digit][workgroup] table in place (one workgroup per digit, coalesced chunks of
// 256 with a carry) and the row totals; the scatter adds the prefix over the digits itself.  (A first version scanned the
// flattened table with one workgroup: 160 k entries, strided reads -- ~50 us per pass for the 1.3 M-point map.)
__global__ __launch_bounds__(256) void k_vsort_scan_rows(int* __restrict__ hist, int n_blocks, int* __restrict__ row_total)
{
    __shared__ int s_wave[4];
    int* row = hist + (size_t)blockIdx.x * n_blocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int b = 0; b < n_blocks; b += 256) {
        const int i = b + threadIdx.x;
        const int v = i < n_blocks ? row[i] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int sw = s_wave[w]; if (w < wave) woff += sw; tot += sw; }
        __syncthreads();
        if (i < n_blocks) row[i] = carry + woff + incl - v;
        carry += tot;
    }
    if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

// exclusive scan of `n` ints in place by ONE workgroup (the per-workgroup head counts: n / 1024 entries); *total receives the sum
__global__ __launch_bounds__(256) void k_vsort_scan_small(int* __restrict__ a, int n, int* __restrict__ total)
{
    __shared__ int s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int b = 0; b < n; b += 256) {
        const int i = b + threadIdx.x;
        const int v = i < n ? a[i] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int sw = s_wave[w]; if (w < wave) woff += sw; tot += sw; }
        __syncthreads();
        if (i < n) a[i] = carry + woff + incl - v;
        carry += tot;
    }
    if (total && threadIdx.x == 0) *total = carry;
}

// Stable scatter of one pass.  rank of an item = (items of the same digit in earlier workgroups: the scanned table) +
// (in earlier waves of this workgroup) + (earlier in this wave's tile: rows k' < k, then lanes l' < l of row k).
template <int ITEMS>
__global__ __launch_bounds__(LIO_VS_THREADS) void k_vsort_scatter(const uint2* __restrict__ in, int n, int shift, unsigned mask,
                                                                  const int* __restrict__ hist_scanned, const int* __restrict__ row_total, int n_blocks,
                                                                  uint2* __restrict__ out)
{
    __shared__ int s_dig[LIO_VS_BINS];                      // items of all digits below d, over the whole input
    __shared__ volatile int s_cnt[4][LIO_VS_BINS];          // per wave and digit: items seen so far (no atomics: one wave runs in lockstep)
    __shared__ int s_base[4][LIO_VS_BINS];                  // global position of the first item of (wave, digit)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < 4; ++w) s_cnt[w][threadIdx.x] = 0;
    {   // exclusive prefix of the 256 row totals (Hillis-Steele in LDS)
        s_dig[threadIdx.x] = row_total[threadIdx.x];
        __syncthreads();
        for (int off = 1; off < LIO_VS_BINS; off <<= 1) {
            const int v = (int)threadIdx.x >= off ? s_dig[threadIdx.x - off] : 0;
            __syncthreads();
            s_dig[threadIdx.x] += v;
            __syncthreads();
        }
    }
    const long long tile = (long long)blockIdx.x * (LIO_VS_THREADS * ITEMS) + (long long)wave * (64 * ITEMS);
    uint2 item[ITEMS];
    int rank[ITEMS];
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;      // lanes below this one
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const long long i = tile + k * 64 + lane;
        const bool valid = i < n;
        item[k] = valid ? in[i] : make_uint2(0xffffffffu, 0u);
        const unsigned d = (item[k].x >> shift) & mask;
        unsigned long long peers = __ballot(valid);          // the valid lanes of this row holding the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        int r = 0;
        if (valid) {
            const int seen = s_cnt[wave][d];                 // every lane reads before the leader below writes (one wave, in-order LDS)
            r = seen + __popcll(peers & lt);
            __builtin_amdgcn_wave_barrier();
            if ((peers & lt) == 0ull) s_cnt[wave][d] = seen + __popcll(peers);   // leader = lowest lane of the group
        }
        __builtin_amdgcn_wave_barrier();
        rank[k] = r;
    }
    __syncthreads();
    {   // prefix over the waves for digit `threadIdx.x`, on top of the scanned [digit][workgroup] table
        const int d = threadIdx.x;
        int run = (d ? s_dig[d - 1] : 0) + hist_scanned[(size_t)d * n_blocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; ++w) { s_base[w][d] = run; run += s_cnt[w][d]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const long long i = tile + k * 64 + lane;
        if (i < n) out[s_base[wave][(item[k].x >> shift) & mask] + rank[k]] = item[k];
    }
}

// Segment heads of the sorted pairs (a voxel = a run of equal keys): per workgroup of 1024 consecutive pairs, their number
__global__ __launch_bounds__(256) void k_vsort_head_count(const uint2* __restrict__ sorted, int n, int* __restrict__ blk_heads)
{
    __shared__ int s_w[4];
    const long long base = (long long)blockIdx.x * 1024;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < n) c += (i == 0 || sorted[i].x != sorted[i - 1].x) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_heads[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// seg_start[o] = position of the head of the o-th voxel (ascending key); seg_start[*n_seg] = n.  blk_heads holds the
// exclusive scan of k_vsort_head_count's output.
__global__ __launch_bounds__(256) void k_vsort_head_emit(const uint2* __restrict__ sorted, int n, const int* __restrict__ blk_heads,
                                                         const int* __restrict__ n_seg, int* __restrict__ seg_start)
{
    __shared__ int s_w[4];
    const long long base = (long long)blockIdx.x * 1024;
    int run = blk_heads[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < 4; ++k) {                            // rows of 256 consecutive pairs, in order
        const long long i = base + k * 256 + threadIdx.x;
        const bool head = i < n && (i == 0 || sorted[i].x != sorted[i - 1].x);
        const unsigned long long m = __ballot(head);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        const int row_total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (head) seg_start[run + before + __popcll(m & (lane ? (~0ull >> (64 - lane)) : 0ull))] = (int)i;
        run += row_total;
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) seg_start[*n_seg] = n;
}

#define LIO_VS_SERIAL_MAX 128    // voxels with at most this many points: one thread adds them up; larger ones are queued

// The fp32 running sums of every voxel's points in ascending input index (the order they sit in after the stable sort),
// divided by the count (PCL's CentroidPoint accumulators).  Work is cut by POINTS, not by voxels: wave-chunk c owns the voxels
// whose head lies in the 1024 sorted positions [1024 c, 1024 (c + 1)) -- blk_heads[c] .. blk_heads[c + 1], the same blocking
// the head count used -- gathers positions [1024 c, 1024 (c + 1) + LIO_VS_SERIAL_MAX) into LDS with all lanes (index loads,
// then point gathers, all independent), and then every lane adds its voxels up from LDS.  The additions are a serial chain
// by contract; the loads need not be.  (Two earlier forms: one thread per voxel chasing sorted[j] -> p[...] itself, 42 us for
// 67 k voxels at one wave per SIMD; 32 voxels per wave staged when they fit -- the local map's busiest voxels hold ~100
// points each, so the chunks that matter never fitted and one lane chasing 100 points set the kernel's time, 45 us.)
// Voxels above LIO_VS_SERIAL_MAX points go to k_vsort_centroid_large.  Nothing here needs the host to know the voxel count.
#define LIO_VS_CHUNK 1024
__global__ __launch_bounds__(256) void k_vsort_centroid(const float4* __restrict__ p, const uint2* __restrict__ sorted, int n,
                                                        const int* __restrict__ seg_start, const int* __restrict__ blk_heads, int n_hblk,
                                                        const int* __restrict__ n_seg, float4* __restrict__ out,
                                                        int* __restrict__ large_list, int* __restrict__ n_large)
{
    __shared__ __attribute__((aligned(16))) float4 s_pt[4][LIO_VS_CHUNK + LIO_VS_SERIAL_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;                     // this wave's chunk
    if (c >= n_hblk) return;                                 // wave-uniform
    const int o_lo = blk_heads[c], o_hi = c + 1 < n_hblk ? blk_heads[c + 1] : *n_seg;
    if (o_lo >= o_hi) return;                                // no voxel starts here (the inside of a crowded voxel)
    const int base = c * LIO_VS_CHUNK, m = min(n - base, LIO_VS_CHUNK + LIO_VS_SERIAL_MAX);
    {   // stage: 18 index loads, then 18 point gathers, then the LDS writes
        constexpr int R = (LIO_VS_CHUNK + LIO_VS_SERIAL_MAX) / 64;
        unsigned idx[R];
#pragma unroll
        for (int k = 0; k < R; ++k) idx[k] = (k * 64 + lane < m) ? sorted[base + k * 64 + lane].y : 0u;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const float4 v = p[idx[k]];
            if (k * 64 + lane < m) s_pt[wave][k * 64 + lane] = v;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // one wave: its LDS writes are complete before its lanes read them
        __builtin_amdgcn_wave_barrier();
    }
    for (int o = o_lo + lane; o < o_hi; o += 64) {
        const int b = seg_start[o], cnt_o = seg_start[o + 1] - b;
        if (cnt_o > LIO_VS_SERIAL_MAX) { large_list[atomicAdd(n_large, 1)] = o; continue; }
        const float4* q = &s_pt[wave][b - base];
        float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
        int j = 0;
        for (; j + 4 <= cnt_o; j += 4) {                     // four LDS reads in flight, the additions stay in order
            const float4 v0 = q[j], v1 = q[j + 1], v2 = q[j + 2], v3 = q[j + 3];
            sx += v0.x; sy += v0.y; sz += v0.z; si += v0.w;
            sx += v1.x; sy += v1.y; sz += v1.z; si += v1.w;
            sx += v2.x; sy += v2.y; sz += v2.z; si += v2.w;
            sx += v3.x; sy += v3.y; sz += v3.z; si += v3.w;
        }
        for (; j < cnt_o; ++j) { const float4 v = q[j]; sx += v.x; sy += v.y; sz += v.z; si += v.w; }
        const float cnt = (float)cnt_o;
        out[o] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
    }
}

// Crowded voxels (the rings next to the sensor put thousands of returns into one voxel): one workgroup per queued voxel
// loads its points, already in order, 1024 at a time through LDS; four lanes carry the four running sums (x, y, z,
// intensity are independent chains; the order inside each is the contract).
__global__ __launch_bounds__(256) void k_vsort_centroid_large(const float4* __restrict__ p, const uint2* __restrict__ sorted,
                                                              const int* __restrict__ seg_start, float4* __restrict__ out,
                                                              const int* __restrict__ large_list, const int* __restrict__ n_large)
{
    __shared__ __attribute__((aligned(16))) float4 s_pt[1024];
    const int total = *n_large;
    for (int q = blockIdx.x; q < total; q += gridDim.x) {
        const int o = large_list[q];
        const int b = seg_start[o], n = seg_start[o + 1] - b;
        float acc = 0.0f;
        for (int c0 = 0; c0 < n; c0 += 1024) {
            const int m = min(1024, n - c0);
            for (int j = threadIdx.x; j < m; j += 256) s_pt[j] = p[sorted[b + c0 + j].y];
            __syncthreads();
            if (threadIdx.x < 4) {
                const float* comp = reinterpret_cast<const float*>(s_pt) + threadIdx.x;
                for (int j = 0; j < m; ++j) acc += comp[j * 4];
            }
            __syncthreads();
        }
        if (threadIdx.x < 4) reinterpret_cast<float*>(out + o)[threadIdx.x] = acc / (float)n;
    }
}
