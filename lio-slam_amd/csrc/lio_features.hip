// lio_features.hip -- the rest of FeatureExtraction::laserCloudInfoHandler (FE:67-79), SURVEY 8f rank 2:
//   calculateSmoothness FE:81-101 (as K2), markOccludedPoints FE:103-139, extractFeatures FE:141-238.
// FE = /root/reference/src/liorf/src/featureExtraction.cpp, MSG = msg/cloud_info.msg.
// Compile with -ffp-contract=off.
//
// Parallel structure.  The smoothness and occlusion passes are per point.  extractFeatures is
// sequential-greedy inside a ring (a pick in one sector suppresses neighbours that the next sector
// reads), but rings only ever touch their own index window [startRingIndex-5, endRingIndex+4]: one
// workgroup per ring keeps that window in LDS, sorts every sector cooperatively (bitonic, keys =
// (curvature bits, point index): the reference's by_value order with ties in ascending index), lets
// one lane run the greedy picks FE:165-222, and then voxel-filters the ring's surface candidates
// (FE:232-236, pcl::VoxelGrid per ring) without leaving LDS.
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../include/liogpu.h"
#include "lio_pool.h"

int lio_fail_ext(int code, const char* what, hipError_t e);   // liogpu_api.hip

#define HIPCHK(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) return lio_fail_ext(LIO_ERR_HIP, #expr, _e);        \
    } while (0)

typedef LioTemp DevBuf;
typedef unsigned long long u64;

#define FEAT_BLOCK     256
#define FEAT_MAX_RING  4096     // points of one ring window held in LDS
#define FEAT_MAX_SECT  1024     // points of one sector (sorted at once)
#define FEAT_MAX_PICK  120      // 6 sectors x 20 corners, FE:171

// ---- per point: curvature FE:86-93 (0 outside [5, n-5)), flags cleared for the whole scan
__global__ __launch_bounds__(256) void k_feat_smoothness(const float* __restrict__ r, int n, float* __restrict__ curv,
                                                         int* __restrict__ picked, int* __restrict__ label)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float c = 0.0f;
    if (i >= 5 && i < n - 5) {
        const float d = r[i - 5] + r[i - 4] + r[i - 3] + r[i - 2] + r[i - 1] - r[i] * 10
                      + r[i + 1] + r[i + 2] + r[i + 3] + r[i + 4] + r[i + 5];
        c = d * d;
    }
    curv[i] = c;
    picked[i] = 0;
    label[i] = 0;
}

// ---- per point: markOccludedPoints FE:103-139.  Every write is "= 1", so the order of the
// reference's loop does not matter.
__global__ __launch_bounds__(256) void k_feat_occlusion(const float* __restrict__ r, const int* __restrict__ col, int n,
                                                        int* __restrict__ picked)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 5 || i >= n - 6) return;
    const float depth1 = r[i], depth2 = r[i + 1];
    const int columnDiff = abs(col[i + 1] - col[i]);
    if (columnDiff < 10) {
        if ((double)(depth1 - depth2) > 0.3) {
            for (int l = -5; l <= 0; ++l) picked[i + l] = 1;
        } else if ((double)(depth2 - depth1) > 0.3) {
            for (int l = 1; l <= 6; ++l) picked[i + l] = 1;
        }
    }
    const float diff1 = fabsf(r[i - 1] - r[i]), diff2 = fabsf(r[i + 1] - r[i]);
    if ((double)diff1 > 0.02 * (double)r[i] && (double)diff2 > 0.02 * (double)r[i]) picked[i] = 1;
}

// ---- LDS helpers
__device__ static void feat_bitonic_sort(u64* a, int n_pow2)
{
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n_pow2 >> 1); t += FEAT_BLOCK) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const bool up = (i & k) == 0;
                const u64 x = a[i], y = a[p];
                if ((x > y) == up) { a[i] = y; a[p] = x; }
            }
            __syncthreads();
        }
    }
}

__device__ static int feat_pow2_at_least(int n)
{
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}

// exclusive scan of one int per thread over the workgroup; *total = sum
__device__ static int feat_block_scan(int v, int* total, int* s_tmp /* [8] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < FEAT_BLOCK / 64; ++w) { if (w < wave) base += s_tmp[w]; tot += s_tmp[w]; }
    *total = tot;
    return base + inc - v;
}

struct FeatParams {
    const unsigned char* cloud; size_t stride; int n;       // extractedCloud, x,y,z @0,4,8, intensity @16
    const int* start_ring; const int* end_ring;             // MSG:4-5
    const int* col; const float* curv;                      // MSG:7, cloudCurvature
    int* picked; int* label;
    float edge_thr, surf_thr, leaf;
    int* corner_idx;        // [ring][FEAT_MAX_PICK] picked corner point indices, in pick order
    int* corner_cnt;        // [ring]
    float4* surf_stage;     // [n] ring r writes its filtered surface points from its window start
    int* surf_cnt;          // [ring]
    int* status;            // != 0: a capacity limit was hit
};

// The suppression walk FE:178-193 / FE:210-221 on the ring window held in LDS, by one wave: lanes 0..4 test
// the forward steps l = 1..5, lanes 8..12 the backward steps l = -1..-5; a walk marks every step before its
// first "stop" (column gap > 10, or the end of the arrays).
__device__ static void feat_suppress_wave(volatile unsigned char* s_picked, const short* s_col, int ind, int w0, int n, int lane)
{
    const bool fwd = lane < 5, bwd = lane >= 8 && lane < 13;
    const int l = fwd ? lane + 1 : -(lane - 7);
    bool stop = false;
    if (fwd) stop = (ind + l >= n) || abs((int)s_col[min(ind + l, n - 1) - w0] - (int)s_col[ind + l - 1 - w0]) > 10;
    if (bwd) stop = (ind + l < 0) || abs((int)s_col[max(ind + l, 0) - w0] - (int)s_col[ind + l + 1 - w0]) > 10;
    const unsigned long long sm = __ballot(stop);
    const unsigned fm = (unsigned)(sm & 0x1f), bm = (unsigned)((sm >> 8) & 0x1f);
    const int ff = fm ? __builtin_ctz(fm) : 5, bf = bm ? __builtin_ctz(bm) : 5;   // first stopping step (0-based)
    if (lane == 0) s_picked[ind - w0] = 1;
    if (fwd && lane < ff) s_picked[ind + l - w0] = 1;
    if (bwd && lane - 8 < bf) s_picked[ind + l - w0] = 1;
}

// One greedy pass of a sector, executed by one wave in chunks of 64 visit positions.  Visit order:
// corner pass = element ep, then the sorted range descending (FE:165 `k = ep .. sp`); surface pass = the
// sorted range ascending, then element ep (FE:197).  Within a chunk every lane holds one element; the wave
// repeatedly takes the first lane whose element is still unpicked and passes the threshold -- exactly the
// element the sequential loop would reach next -- applies the pick, and re-reads the flags.
template <bool CORNER>
__device__ static int feat_greedy_pass(const u64* s_sort, unsigned char* s_picked_, signed char* s_label, const float* s_curv,
                                       const short* s_col, int sp, int ep, int w0, int n, float thr,
                                       int* corner_idx, int n_corner)
{
    volatile unsigned char* s_picked = s_picked_;
    const int lane = threadIdx.x & 63;
    const int m = ep - sp;                                   // sorted elements; visits = m + 1
    int largestPickedNum = 0;
    for (int q0 = 0; q0 <= m; q0 += 64) {
        const int q = q0 + lane;
        const bool valid = q <= m;
        int ind = ep;
        if (valid) {
            if (CORNER) { if (q > 0) ind = w0 + (int)(unsigned)(s_sort[m - q] & 0x1fffu); }     // (keys carry the window-relative position)
            else        { if (q < m) ind = w0 + (int)(unsigned)(s_sort[q] & 0x1fffu); }
        }
        const float cv = s_curv[ind - w0];
        const bool cand = valid && (CORNER ? cv > thr : cv < thr);
        int cursor = 0;
        for (;;) {
            const bool live = cand && lane >= cursor && s_picked[ind - w0] == 0;
            const unsigned long long mask = __ballot(live);
            if (!mask) break;
            const int j = __builtin_ctzll(mask);
            const int pick = __shfl(ind, j);
            if (CORNER) {
                ++largestPickedNum;                                          // FE:169-176
                if (largestPickedNum > 20) return n_corner;
                if (lane == 0) { s_label[pick - w0] = 1; corner_idx[n_corner] = pick; }
                ++n_corner;
            } else {
                if (lane == 0) s_label[pick - w0] = -1;                      // FE:202
            }
            feat_suppress_wave(s_picked, s_col, pick, w0, n, lane);
            cursor = j + 1;
        }
    }
    return n_corner;
}

__global__ __launch_bounds__(FEAT_BLOCK) void k_feat_ring(FeatParams P)
{
    // 56 KiB pool: greedy phase = curvature (16 K) + sort keys of all six sectors (32 K) + columns (8 K);
    //              voxel phase  = (voxel index, list position) sort keys (32 K)
    __shared__ __attribute__((aligned(16))) u64 s_pool[FEAT_MAX_RING / 2 + FEAT_MAX_RING + FEAT_MAX_RING / 4];
    __shared__ unsigned char s_picked[FEAT_MAX_RING];
    __shared__ signed char s_label[FEAT_MAX_RING];
    __shared__ unsigned short s_list[FEAT_MAX_RING];        // surface candidates (window-relative), FE:224-229
    __shared__ int s_tmp[8];
    __shared__ int s_n_corner, s_n_list;
    __shared__ float s_box[6];
    float* s_curv = reinterpret_cast<float*>(s_pool);                       // [4096]
    u64* s_sort = s_pool + FEAT_MAX_RING / 2;                               // [4096]
    short* s_col = reinterpret_cast<short*>(s_pool + FEAT_MAX_RING / 2 + FEAT_MAX_RING);   // [4096]

    const int ring = blockIdx.x;
    const int start = P.start_ring[ring], end = P.end_ring[ring];
    if (threadIdx.x == 0) { P.corner_cnt[ring] = 0; P.surf_cnt[ring] = 0; }
    if (end < start) return;                                 // every sector has sp >= ep (FE:159)
    const int w0 = max(start - 5, 0), w1 = min(end + 5, P.n - 1);           // window [w0, w1]
    const int wl = w1 - w0 + 1;
    if (wl > FEAT_MAX_RING) { if (threadIdx.x == 0) atomicExch(P.status, 1); return; }

    for (int t = threadIdx.x; t < wl; t += FEAT_BLOCK) {
        s_curv[t] = P.curv[w0 + t];
        s_col[t] = (short)P.col[w0 + t];
        s_picked[t] = (unsigned char)(P.picked[w0 + t] != 0);
        s_label[t] = 0;
    }
    if (threadIdx.x == 0) { s_n_corner = 0; s_n_list = 0; }
    __syncthreads();

    // The six std::sort calls FE:162 as ONE sort: key = (sector, curvature, position) -- the sectors are disjoint ranges, so the
    // sorted array is the six sorted sectors back to back (66 compare-exchange stages for 2048 keys instead of 6 x 45 for
    // 6 x 512); ties inside a sector by position, as before.
    int sec_sp[6], sec_ep[6], sec_off[7];
    int n_keys = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        sec_sp[j] = (start * (6 - j) + end * j) / 6;                        // FE:156
        sec_ep[j] = (start * (5 - j) + end * (j + 1)) / 6 - 1;              // FE:157
        sec_off[j] = n_keys;
        if (sec_sp[j] < sec_ep[j]) n_keys += sec_ep[j] - sec_sp[j];         // std::sort range [sp, ep), FE:162
    }
    sec_off[6] = n_keys;
    if (n_keys > 0) {
        const int mp = feat_pow2_at_least(n_keys);
        for (int t = threadIdx.x; t < mp; t += FEAT_BLOCK) {
            u64 key = ~0ull;
            if (t < n_keys) {
                int j = 0;
#pragma unroll
                for (int q = 1; q < 6; ++q) if (t >= sec_off[q]) j = q;     // (an empty sector shares its offset with the next one: the last match wins)
                const int pos = sec_sp[j] + (t - sec_off[j]);
                key = ((u64)j << 45) | ((u64)(__float_as_uint(s_curv[pos - w0]) & 0x7fffffffu) << 13) | (u64)(pos - w0);
            }
            s_sort[t] = key;
        }
        __syncthreads();
        feat_bitonic_sort(s_sort, mp);
    }

    int first_valid = -1, last_valid = -1;                   // union of the sectors that were processed
    for (int j = 0; j < 6; ++j) {
        const int sp = sec_sp[j], ep = sec_ep[j];
        if (sp >= ep) continue;                                             // FE:159 (workgroup-uniform)
        const u64* s_sec = s_sort + sec_off[j];                             // this sector's keys, ascending curvature
        if (threadIdx.x < 64) {                                             // wave 0; LDS ops of one wave retire in order
            int n_corner = s_n_corner;
            n_corner = feat_greedy_pass<true>(s_sec, s_picked, s_label, s_curv, s_col, sp, ep, w0, P.n, P.edge_thr,
                                              P.corner_idx + ring * FEAT_MAX_PICK, n_corner);     // FE:165-195
            if (threadIdx.x == 0) s_n_corner = n_corner;
            feat_greedy_pass<false>(s_sec, s_picked, s_label, s_curv, s_col, sp, ep, w0, P.n, P.surf_thr,
                                    nullptr, 0);                                                   // FE:197-222
        }
        __syncthreads();
        if (first_valid < 0) first_valid = sp;
        last_valid = ep;
        // FE:224-229: positions sp..ep whose label <= 0, in position order (appended sector by sector)
        for (int b = sp; b <= ep; b += FEAT_BLOCK) {
            const int k = b + (int)threadIdx.x;
            const int f = (k <= ep && s_label[k - w0] <= 0) ? 1 : 0;
            int tot;
            const int off = feat_block_scan(f, &tot, s_tmp);
            const int base = s_n_list;
            if (f) s_list[base + off] = (unsigned short)(k - w0);
            __syncthreads();
            if (threadIdx.x == 0) s_n_list = base + tot;
            __syncthreads();
        }
    }
    (void)first_valid; (void)last_valid;

    // write the window back (cloudNeighborPicked / cloudLabel as the reference leaves them)
    for (int t = threadIdx.x; t < wl; t += FEAT_BLOCK) {
        P.picked[w0 + t] = s_picked[t];
        P.label[w0 + t] = s_label[t];
    }
    if (threadIdx.x == 0) P.corner_cnt[ring] = s_n_corner;
    const int nl = s_n_list;
    if (nl == 0) return;
    __syncthreads();                                         // the pool changes hands

    // ---- pcl::VoxelGrid on the ring's candidates (FE:232-236; same arithmetic as K7 / lo_voxel_grid)
    const float inv = 1.0f / P.leaf;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int t = threadIdx.x; t < nl; t += FEAT_BLOCK) {
        const float* p = reinterpret_cast<const float*>(P.cloud + (size_t)(w0 + s_list[t]) * P.stride);
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], p[a]); mx[a] = fmaxf(mx[a], p[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    }
    float* s_red = reinterpret_cast<float*>(s_pool);        // [4 waves][6]
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; ++a) { s_red[(threadIdx.x >> 6) * 6 + a] = mn[a]; s_red[(threadIdx.x >> 6) * 6 + 3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_red[threadIdx.x];
        for (int w = 1; w < FEAT_BLOCK / 64; ++w)
            v = threadIdx.x < 3 ? fminf(v, s_red[w * 6 + threadIdx.x]) : fmaxf(v, s_red[w * 6 + threadIdx.x]);
        s_box[threadIdx.x] = v;
    }
    __syncthreads();
    for (int a = 0; a < 3; ++a) { mn[a] = s_box[a]; mx[a] = s_box[3 + a]; }
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1,
                    dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    float4* out = P.surf_stage + w0;
    if (dx * dy * dz > (long long)INT32_MAX) {               // PCL passes the cloud through unfiltered
        for (int t = threadIdx.x; t < nl; t += FEAT_BLOCK) {
            const float* p = reinterpret_cast<const float*>(P.cloud + (size_t)(w0 + s_list[t]) * P.stride);
            out[t] = make_float4(p[0], p[1], p[2], p[4]);
        }
        if (threadIdx.x == 0) P.surf_cnt[ring] = nl;
        return;
    }
    const int min_b0 = (int)floorf(mn[0] * inv), min_b1 = (int)floorf(mn[1] * inv), min_b2 = (int)floorf(mn[2] * inv);
    const int mul1 = (int)floorf(mx[0] * inv) - min_b0 + 1;
    const int mul2 = mul1 * ((int)floorf(mx[1] * inv) - min_b1 + 1);
    __syncthreads();                                         // s_red was read
    const int np = feat_pow2_at_least(nl);
    for (int t = threadIdx.x; t < np; t += FEAT_BLOCK) {
        u64 key = ~0ull;
        if (t < nl) {
            const float* p = reinterpret_cast<const float*>(P.cloud + (size_t)(w0 + s_list[t]) * P.stride);
            const int i0 = (int)(floorf(p[0] * inv) - (float)min_b0);
            const int i1 = (int)(floorf(p[1] * inv) - (float)min_b1);
            const int i2 = (int)(floorf(p[2] * inv) - (float)min_b2);
            key = ((u64)(unsigned)(i0 + i1 * mul1 + i2 * mul2) << 32) | (unsigned)t;      // ties: input order
        }
        s_pool[t] = key;
    }
    __syncthreads();
    feat_bitonic_sort(s_pool, np);
    // one thread per voxel run: centroid of x,y,z,intensity summed in input order, output in voxel order
    int n_out = 0;
    for (int b = 0; b < nl; b += FEAT_BLOCK) {
        const int t = b + (int)threadIdx.x;
        const bool head = t < nl && (t == 0 || (unsigned)(s_pool[t] >> 32) != (unsigned)(s_pool[t - 1] >> 32));
        int tot;
        const int off = feat_block_scan(head ? 1 : 0, &tot, s_tmp);
        if (head) {
            const unsigned vox = (unsigned)(s_pool[t] >> 32);
            float sx = 0, sy = 0, sz = 0, si = 0;
            int e = t;
            while (e < nl && (unsigned)(s_pool[e] >> 32) == vox) {
                const int li = (int)(unsigned)(s_pool[e] & 0xffffffffu);
                const float* p = reinterpret_cast<const float*>(P.cloud + (size_t)(w0 + s_list[li]) * P.stride);
                sx += p[0]; sy += p[1]; sz += p[2]; si += p[4];
                ++e;
            }
            const float cnt = (float)(e - t);
            out[n_out + off] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
        }
        n_out += tot;
    }
    if (threadIdx.x == 0) P.surf_cnt[ring] = n_out;
}

// ---- ring-major concatenation of the per-ring results (cornerCloud push_back order FE:172, surfaceCloud += FE:236)
__global__ __launch_bounds__(256) void k_feat_offsets(const int* __restrict__ corner_cnt, const int* __restrict__ surf_cnt,
                                                      int n_scan, int* __restrict__ corner_off, int* __restrict__ surf_off)
{
    if (threadIdx.x == 0) {
        int c = 0, s = 0;
        for (int i = 0; i < n_scan; ++i) { corner_off[i] = c; surf_off[i] = s; c += corner_cnt[i]; s += surf_cnt[i]; }
        corner_off[n_scan] = c; surf_off[n_scan] = s;
    }
}

__global__ __launch_bounds__(256) void k_feat_gather(FeatParams P, const int* __restrict__ corner_off,
                                                     const int* __restrict__ surf_off,
                                                     unsigned char* __restrict__ corner_out,
                                                     unsigned char* __restrict__ surf_out, size_t out_stride)
{
    const int ring = blockIdx.x;
    const int nc = P.corner_cnt[ring], ns = P.surf_cnt[ring];
    for (int t = threadIdx.x; t < nc; t += 256) {
        const int ind = P.corner_idx[ring * FEAT_MAX_PICK + t];
        const float* p = reinterpret_cast<const float*>(P.cloud + (size_t)ind * P.stride);
        float* o = reinterpret_cast<float*>(corner_out + (size_t)(corner_off[ring] + t) * out_stride);
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[4] = p[4];
    }
    const int start = P.start_ring[ring];
    const int w0 = max(start - 5, 0);
    for (int t = threadIdx.x; t < ns; t += 256) {
        const float4 v = P.surf_stage[w0 + t];
        float* o = reinterpret_cast<float*>(surf_out + (size_t)(surf_off[ring] + t) * out_stride);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[4] = v.w;
    }
}

extern "C" void lio_feature_default_config(lio_feature_config* c)
{
    memset(c, 0, sizeof(*c));
    c->N_SCAN = 16;
    c->edgeThreshold = 1.0f;       // config/*.yaml edgeThreshold, UT:186
    c->surfThreshold = 0.1f;       // surfThreshold, UT:187
    c->surfLeafSize = 0.2f;        // mappingSurfLeafSize, FE:56
    c->device_id = 0;
}

extern "C" int lio_extract_features(const lio_feature_config* cfg, const void* cloud, size_t n, size_t stride,
                                    const int32_t* startRingIndex, const int32_t* endRingIndex,
                                    const int32_t* pointColInd, const float* pointRange,
                                    void* corner_out, size_t* n_corner, void* surface_out, size_t* n_surface,
                                    size_t out_stride, float* curvature, int32_t* neighbor_picked, int32_t* label)
{
    if (!cfg || !n_corner || !n_surface || !startRingIndex || !endRingIndex)
        return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if (n && (!cloud || !pointColInd || !pointRange || !corner_out || !surface_out))
        return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if (stride < 20 || (stride & 3) || out_stride < 20 || (out_stride & 3))
        return lio_fail_ext(LIO_ERR_ARG, "point strides must be >= 20 and multiples of 4", hipSuccess);
    if (cfg->N_SCAN < 1 || cfg->N_SCAN > 1024) return lio_fail_ext(LIO_ERR_ARG, "N_SCAN out of range", hipSuccess);
    if (!(cfg->surfLeafSize > 0.0f)) return lio_fail_ext(LIO_ERR_ARG, "surfLeafSize must be positive", hipSuccess);
    if (n > 0x7fffffffull - 1024) return lio_fail_ext(LIO_ERR_CAPACITY, "cloud too large", hipSuccess);
    *n_corner = 0; *n_surface = 0;
    // Rings must own disjoint index windows [start-5, end+4] in ascending order -- what cloudExtraction
    // produces -- or the reference's ring-after-ring order would matter.
    long prev_hi = -1000;
    for (int i = 0; i < cfg->N_SCAN; ++i) {
        const long s = startRingIndex[i], e = endRingIndex[i];
        if (e < s) continue;
        if (s < 0 || e > (long)n - 1) return lio_fail_ext(LIO_ERR_ARG, "ring index range outside the cloud", hipSuccess);
        if (s - 5 <= prev_hi) return lio_fail_ext(LIO_ERR_ARG, "ring windows overlap or are not ascending", hipSuccess);
        prev_hi = e + 4;
        if (e - s + 11 > FEAT_MAX_RING) return lio_fail_ext(LIO_ERR_CAPACITY, "more than 4086 points in one ring", hipSuccess);
    }
    if (n == 0) return LIO_OK;
    for (size_t i = 0; i < n; ++i)
        if (pointColInd[i] < -32768 || pointColInd[i] > 32767)
            return lio_fail_ext(LIO_ERR_CAPACITY, "pointColInd does not fit int16", hipSuccess);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lio_fail_ext(LIO_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)", hipSuccess);
    HIPCHK(hipSetDevice(cfg->device_id));
    (void)hipGetLastError();

    const int ns = cfg->N_SCAN;
    DevBuf d_cloud, d_rings, d_col, d_range, d_curv, d_picked, d_label, d_cidx, d_cnt, d_stage, d_cout, d_sout;
    HIPCHK(d_cloud.alloc(n * stride));
    HIPCHK(d_rings.alloc(sizeof(int) * 2 * (size_t)ns));
    HIPCHK(d_col.alloc(n * 4)); HIPCHK(d_range.alloc(n * 4)); HIPCHK(d_curv.alloc(n * 4));
    HIPCHK(d_picked.alloc(n * 4)); HIPCHK(d_label.alloc(n * 4));
    HIPCHK(d_cidx.alloc(sizeof(int) * (size_t)ns * FEAT_MAX_PICK));
    HIPCHK(d_cnt.alloc(sizeof(int) * (4 * (size_t)ns + 3)));            // corner_cnt, surf_cnt, corner_off[+1], surf_off[+1], status
    HIPCHK(d_stage.alloc(n * sizeof(float4)));
    HIPCHK(d_cout.alloc((size_t)ns * FEAT_MAX_PICK * out_stride));
    HIPCHK(d_sout.alloc(n * out_stride));
    hipStream_t s = nullptr;
    HIPCHK(hipMemcpyAsync(d_cloud.p, cloud, n * stride, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_rings.p, startRingIndex, sizeof(int) * ns, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_rings.as<int>() + ns, endRingIndex, sizeof(int) * ns, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_col.p, pointColInd, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_range.p, pointRange, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d_cnt.p, 0, sizeof(int) * (4 * (size_t)ns + 3), s));
    HIPCHK(hipMemsetAsync(d_cout.p, 0, (size_t)ns * FEAT_MAX_PICK * out_stride, s));
    HIPCHK(hipMemsetAsync(d_sout.p, 0, n * out_stride, s));

    FeatParams P;
    P.cloud = d_cloud.as<unsigned char>(); P.stride = stride; P.n = (int)n;
    P.start_ring = d_rings.as<int>(); P.end_ring = d_rings.as<int>() + ns;
    P.col = d_col.as<int>(); P.curv = d_curv.as<float>();
    P.picked = d_picked.as<int>(); P.label = d_label.as<int>();
    P.edge_thr = cfg->edgeThreshold; P.surf_thr = cfg->surfThreshold; P.leaf = cfg->surfLeafSize;
    P.corner_idx = d_cidx.as<int>();
    int* cnt = d_cnt.as<int>();
    P.corner_cnt = cnt; P.surf_cnt = cnt + ns;
    int* corner_off = cnt + 2 * ns; int* surf_off = corner_off + ns + 1;
    P.status = surf_off + ns + 1;
    P.surf_stage = d_stage.as<float4>();

    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_feat_smoothness, dim3(nb), dim3(256), 0, s, d_range.as<float>(), (int)n, d_curv.as<float>(),
                       P.picked, P.label);
    hipLaunchKernelGGL(k_feat_occlusion, dim3(nb), dim3(256), 0, s, d_range.as<float>(), d_col.as<int>(), (int)n, P.picked);
    hipLaunchKernelGGL(k_feat_ring, dim3(ns), dim3(FEAT_BLOCK), 0, s, P);
    hipLaunchKernelGGL(k_feat_offsets, dim3(1), dim3(256), 0, s, P.corner_cnt, P.surf_cnt, ns, corner_off, surf_off);
    hipLaunchKernelGGL(k_feat_gather, dim3(ns), dim3(256), 0, s, P, corner_off, surf_off,
                       d_cout.as<unsigned char>(), d_sout.as<unsigned char>(), out_stride);
    std::vector<int> h_cnt(4 * (size_t)ns + 3);
    HIPCHK(hipMemcpyAsync(h_cnt.data(), cnt, sizeof(int) * h_cnt.size(), hipMemcpyDeviceToHost, s));
    if (curvature) HIPCHK(hipMemcpyAsync(curvature, d_curv.p, n * 4, hipMemcpyDeviceToHost, s));
    if (neighbor_picked) HIPCHK(hipMemcpyAsync(neighbor_picked, d_picked.p, n * 4, hipMemcpyDeviceToHost, s));
    if (label) HIPCHK(hipMemcpyAsync(label, d_label.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (h_cnt[4 * (size_t)ns + 2] != 0)
        return lio_fail_ext(LIO_ERR_CAPACITY, "a ring window (> 4096 points) or a sector (> 1024 points) exceeds the LDS budget", hipSuccess);
    const size_t nc = (size_t)h_cnt[2 * (size_t)ns + ns], nsf = (size_t)h_cnt[3 * (size_t)ns + 1 + ns];
    if (nc) HIPCHK(hipMemcpy(corner_out, d_cout.p, nc * out_stride, hipMemcpyDeviceToHost));
    if (nsf) HIPCHK(hipMemcpy(surface_out, d_sout.p, nsf * out_stride, hipMemcpyDeviceToHost));
    *n_corner = nc;
    *n_surface = nsf;
    return LIO_OK;
}
