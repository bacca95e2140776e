// lio_prepare.hip -- the two per-point passes upstream of registration:
//   K1 filter + IMU-rotation deskew  (projectPointCloud / deskewPoint, IP:545-615)
//   K2 range curvature               (calculateSmoothness, FE:81-101)
// IP = /root/reference/src/liorf/src/imageProjection.cpp,
// FE = /root/reference/src/liorf/src/featureExtraction.cpp.
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <string>

#include "../../include/liogpu.h"
#include "lio_pool.h"

#define LIO_DEV __device__ __forceinline__

int lio_fail_ext(int code, const char* what, hipError_t e);   // liogpu_api.hip

#define HIPCHK(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) return lio_fail_ext(LIO_ERR_HIP, #expr, _e);        \
    } while (0)

struct LioDeskewParams {
    const unsigned char* pts;   // PointXYZIRT records, IP:4-15
    size_t stride;
    int n;
    int N_SCAN, downsampleRate, point_filter_num;
    float minFront, minBack, minLeft, minRight, maxRange, maxIntensity;
    int do_deskew;              // !(deskewFlag == -1 || !imuAvailable), IP:547
    double time_scan_cur;
    const double* imuTime; const double* imuRotX; const double* imuRotY; const double* imuRotZ;
    int imuPointerCur;
    // field layout of a record (sensor_msgs/PointCloud2 `fields`); the PointXYZIRT defaults are 0 / 16 / 20 / 24
    int off_xyz, off_intensity, off_ring, off_time;   // byte offsets; off_intensity / off_time < 0: absent
    int ring_type;              // 0 = uint16 (Velodyne, Robosense), 1 = uint8 (Ouster), 2 = int32 (Mulran)
    int time_type;              // 0 = float seconds; 1 = uint32 ns -> t * 1e-9f (IP:243); 2 = uint32 -> (float)t (IP:262);
                                // 3 = double stamp minus the first point's (IP:281)
};

// The fields of record i as cachePointCloud IP:226-285 hands them to projectPointCloud (Velodyne layout).
LIO_DEV void lio_rec_xyzi(const LioDeskewParams& P, const unsigned char* rec, float& x, float& y, float& z, float& inten)
{
    const float* f = reinterpret_cast<const float*>(rec + P.off_xyz);
    x = f[0]; y = f[1]; z = f[2];
    inten = P.off_intensity >= 0 ? *reinterpret_cast<const float*>(rec + P.off_intensity) : 0.0f;
}
LIO_DEV int lio_rec_ring(const LioDeskewParams& P, const unsigned char* rec)
{
    const unsigned char* r = rec + P.off_ring;
    if (P.ring_type == 1) return (int)*r;
    if (P.ring_type == 2) return (int)(unsigned short)*reinterpret_cast<const int*>(r);   // `dst.ring = src.ring` narrows to uint16_t, IP:261
    return (int)*reinterpret_cast<const unsigned short*>(r);
}
LIO_DEV float lio_rec_time(const LioDeskewParams& P, const unsigned char* rec)
{
    if (P.off_time < 0) return 0.0f;
    const unsigned char* t = rec + P.off_time;
    if (P.time_type == 1) return (float)*reinterpret_cast<const unsigned*>(t) * 1e-9f;          // IP:243
    if (P.time_type == 2) return (float)*reinterpret_cast<const unsigned*>(t);                   // IP:262
    if (P.time_type == 3) {                                                                      // IP:269, 281
        double t0, ti;                                   // (8-byte fields of a PointCloud2 need not be 8-byte aligned)
        memcpy(&t0, P.pts + P.off_time, 8);
        memcpy(&ti, t, 8);
        return (float)(ti - t0);
    }
    return *reinterpret_cast<const float*>(t);
}

LIO_DEV float lio_sinf64(float x) { return (float)sin((double)x); }
LIO_DEV float lio_cosf64(float x) { return (float)cos((double)x); }

// IP:596-609: vehicle box / range / intensity, ring validity, ring and point stride
LIO_DEV bool lio_keep_point(const LioDeskewParams& P, int i, float x, float y, float z, float inten, int ring)
{
    const float range = sqrtf(x * x + y * y + z * z);          // common_lib.cpp:27-31
    if ((y < P.minFront && -P.minBack < y && x < P.minLeft && -P.minRight < x) ||
        range > P.maxRange || inten > P.maxIntensity) return false;
    if (ring < 0 || ring >= P.N_SCAN) return false;
    if (ring % P.downsampleRate != 0) return false;
    if (i % P.point_filter_num != 0) return false;
    return true;
}

// findRotation, IP:502-527, tables in LDS (fp64 interpolation, cast to float)
LIO_DEV void lio_find_rotation(double t, const double* sT, const double* sX, const double* sY,
                               const double* sZ, int cur, float& rx, float& ry, float& rz)
{
    int front = 0;
    while (front < cur) {
        if (t < sT[front]) break;
        ++front;
    }
    if (t > sT[front] || front == 0) {
        rx = (float)sX[front]; ry = (float)sY[front]; rz = (float)sZ[front];
    } else {
        const int back = front - 1;
        const double rf = (t - sT[back]) / (sT[front] - sT[back]);
        const double rb = (sT[front] - t) / (sT[front] - sT[back]);
        rx = (float)(sX[front] * rf + sX[back] * rb);
        ry = (float)(sY[front] * rf + sY[back] * rb);
        rz = (float)(sZ[front] * rf + sZ[back] * rb);
    }
}

// linear part of pcl::getTransformation(0,0,0,roll,pitch,yaw), IP:560/565
LIO_DEV void lio_rot_rpy(float roll, float pitch, float yaw, float L[9])
{
    const float A = lio_cosf64(yaw), B = lio_sinf64(yaw);
    const float C = lio_cosf64(pitch), D = lio_sinf64(pitch);
    const float E = lio_cosf64(roll), F = lio_sinf64(roll);
    const float DE = D * E, DF = D * F;
    L[0] = A * C;  L[1] = A * DF - B * E;  L[2] = B * F + A * DE;
    L[3] = B * C;  L[4] = A * E + B * DF;  L[5] = B * DE - A * F;
    L[6] = -D;     L[7] = C * F;           L[8] = C * E;
}

LIO_DEV float lio_cof3(const float m[9], int i, int j)
{
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}

// Eigen general 3x3 inverse (Affine3f::inverse of the first survivor, IP:560)
LIO_DEV void lio_inv3(const float m[9], float r[9])
{
    const float c0 = lio_cof3(m, 0, 0), c1 = lio_cof3(m, 1, 0), c2 = lio_cof3(m, 2, 0);
    const float det = c0 * m[0] + c1 * m[3] + c2 * m[6];
    const float invdet = 1.0f / det;
    r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
    r[3] = lio_cof3(m, 0, 1) * invdet; r[4] = lio_cof3(m, 1, 1) * invdet; r[5] = lio_cof3(m, 2, 1) * invdet;
    r[6] = lio_cof3(m, 0, 2) * invdet; r[7] = lio_cof3(m, 1, 2) * invdet; r[8] = lio_cof3(m, 2, 2) * invdet;
}

// pass 1: survivor flags, per-workgroup survivor counts, index of the first survivor
__global__ __launch_bounds__(256) void k_deskew_flags(LioDeskewParams P, unsigned char* __restrict__ keep,
                                                      int* __restrict__ blk_count, int* __restrict__ first_idx)
{
    __shared__ int s_cnt[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool k = false;
    if (i < P.n) {
        const unsigned char* rec = P.pts + (size_t)i * P.stride;
        float fx, fy, fz, fi;
        lio_rec_xyzi(P, rec, fx, fy, fz, fi);
        k = lio_keep_point(P, i, fx, fy, fz, fi, lio_rec_ring(P, rec));
        keep[i] = k ? 1 : 0;
    }
    const unsigned long long m = __ballot(k);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_cnt[wave] = __popcll(m);
        // (one address for the whole sweep: 1 800 same-address atomics cost ~18 us of this kernel's 23; a wave whose candidate
        //  cannot lower the value it sees -- stale or not, the minimum only falls -- skips the atomic: a few dozen remain)
        if (m) {
            const int cand = blockIdx.x * 256 + wave * 64 + (__ffsll((long long)m) - 1);
            if (cand < __hip_atomic_load(first_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(first_idx, cand);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) blk_count[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive scan of the workgroup counts by one workgroup (<= a few thousand entries)
__global__ __launch_bounds__(256) void k_deskew_scan(int* __restrict__ blk_count, int n_blk, int* __restrict__ total)
{
    __shared__ int s_wave[4];
    int carry = 0;
    for (int b = 0; b < n_blk; b += 256) {
        const int i = b + threadIdx.x;
        const int v = i < n_blk ? blk_count[i] : 0;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
        for (int w = 0; w < 4; ++w) { const int s = s_wave[w]; if (w < wave) woff += s; tot += s; }
        __syncthreads();
        if (i < n_blk) blk_count[i] = carry + woff + incl - v;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

// pass 2: deskew survivors and write them compacted, input order preserved (IP:611-613)
__global__ __launch_bounds__(256) void k_deskew_emit(LioDeskewParams P, const unsigned char* __restrict__ keep,
                                                     const int* __restrict__ blk_off, const int* __restrict__ first_idx,
                                                     unsigned char* __restrict__ out, size_t out_stride)
{
    // all LDS in the dynamic region (keeps the fp64 tables 16-byte aligned)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_inv = reinterpret_cast<float*>(smem);            // 9 floats (+pad to 48 B)
    int* s_cnt = reinterpret_cast<int*>(smem + 48);           // 4 ints
    double* sT = reinterpret_cast<double*>(smem + 64);
    const int nt = P.imuPointerCur + 1;
    double* sX = sT + nt; double* sY = sX + nt; double* sZ = sY + nt;

    if (P.do_deskew) {
        for (int k = threadIdx.x; k < nt; k += 256) {
            sT[k] = P.imuTime[k]; sX[k] = P.imuRotX[k]; sY[k] = P.imuRotY[k]; sZ[k] = P.imuRotZ[k];
        }
    }
    __syncthreads();
    if (P.do_deskew && threadIdx.x == 0) {
        // transStartInverse from the first survivor (firstPointFlag, IP:558-562)
        const int fi = *first_idx;
        float inv[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        if (fi < P.n) {
            const float tf = lio_rec_time(P, P.pts + (size_t)fi * P.stride);
            float rx, ry, rz, L[9];
            lio_find_rotation(P.time_scan_cur + (double)tf, sT, sX, sY, sZ, P.imuPointerCur, rx, ry, rz);
            lio_rot_rpy(rx, ry, rz, L);
            lio_inv3(L, inv);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) s_inv[k] = inv[k];
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool k = (i < P.n) && keep[i];
    const unsigned long long m = __ballot(k);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    if (!k) return;
    int pos = blk_off[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_cnt[w];

    const unsigned char* rec = P.pts + (size_t)i * P.stride;
    float px, py, pz, pi;
    lio_rec_xyzi(P, rec, px, py, pz, pi);
    float ox = px, oy = py, oz = pz;
    if (P.do_deskew) {
        float rx, ry, rz, L[9], Bt[9];
        lio_find_rotation(P.time_scan_cur + (double)lio_rec_time(P, rec), sT, sX, sY, sZ, P.imuPointerCur, rx, ry, rz);  // IP:550-553
        lio_rot_rpy(rx, ry, rz, L);                                                                      // IP:565
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)                                                                  // IP:566
                Bt[r * 3 + c] = s_inv[r * 3 + 0] * L[0 * 3 + c] + s_inv[r * 3 + 1] * L[1 * 3 + c]
                              + s_inv[r * 3 + 2] * L[2 * 3 + c];
        ox = Bt[0] * px + Bt[1] * py + Bt[2] * pz + 0.0f;                                                // IP:569-571
        oy = Bt[3] * px + Bt[4] * py + Bt[5] * pz + 0.0f;
        oz = Bt[6] * px + Bt[7] * py + Bt[8] * pz + 0.0f;
    }
    float* o = reinterpret_cast<float*>(out + (size_t)pos * out_stride);
    o[0] = ox; o[1] = oy; o[2] = oz; o[3] = 1.0f;   // PCL_ADD_POINT4D padding is 1.0
    o[4] = pi;                                       // IP:572
}

// ------------------------------------------------------------------ range image (extension, row A4)
// projectPointCloud + cloudExtraction of upstream LIO-SAM on top of this fork's deskewPoint: see
// oracle/lio_oracle.c lo_range_image for the semantics and why this is an extension.
struct LioRangeImageParams {
    LioDeskewParams d;          // pts, stride, n, N_SCAN, downsampleRate, maxRange, IMU tables, do_deskew
    int H;                      // Horizon_SCAN
    float minRange;
};

// row / column of a raw point, or -1 when upstream's projectPointCloud drops it
LIO_DEV int lio_ri_cell(const LioRangeImageParams& P, float x, float y, float z, int ring, float& range)
{
    range = sqrtf(x * x + y * y + z * z);
    if (range < P.minRange || range > P.d.maxRange) return -1;
    if (ring < 0 || ring >= P.d.N_SCAN) return -1;
    if (ring % P.d.downsampleRate != 0) return -1;
    const float at = (float)atan2((double)x, (double)y);
    const float at180 = at * 180;
    const float horizonAngle = (float)((double)at180 / 3.14159265358979323846);
    const float ang_res_x = (float)(360.0 / (double)(float)P.H);
    int col = (int)(-round(((double)horizonAngle - 90.0) / (double)ang_res_x) + (double)(P.H / 2));
    if (col >= P.H) col -= P.H;
    if (col < 0 || col >= P.H) return -1;
    return ring * P.H + col;
}

// pass 1: the first input point of every cell (atomicMin on the input index) and the first accepted point overall
__global__ __launch_bounds__(256) void k_ri_first(LioRangeImageParams P, int* __restrict__ cell_first, int* __restrict__ first_idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.d.n) return;
    const unsigned char* rec = P.d.pts + (size_t)i * P.d.stride;
    const float* f = reinterpret_cast<const float*>(rec);
    float range;
    const int cell = lio_ri_cell(P, f[0], f[1], f[2], lio_rec_ring(P.d, rec), range);
    if (cell < 0) return;
    atomicMin(&cell_first[cell], i);
    // (same-address atomic of every point of the sweep: skipped when it cannot lower the value this lane sees)
    if (i < __hip_atomic_load(first_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(first_idx, i);
}

// pass 2: one thread per cell: deskew the winner, write it at its rank among the occupied cells (ring-major,
// ascending column = ascending cell index), with its column and raw range
__global__ __launch_bounds__(256) void k_ri_extract(LioRangeImageParams P, const int* __restrict__ cell_first,
                                                    const int* __restrict__ cell_rank, const int* __restrict__ first_idx,
                                                    unsigned char* __restrict__ out, size_t out_stride,
                                                    int* __restrict__ col_out, float* __restrict__ range_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_inv = reinterpret_cast<float*>(smem);
    double* sT = reinterpret_cast<double*>(smem + 64);
    const int nt = P.d.imuPointerCur + 1;
    double* sX = sT + nt; double* sY = sX + nt; double* sZ = sY + nt;
    if (P.d.do_deskew) {
        for (int k = threadIdx.x; k < nt; k += 256) {
            sT[k] = P.d.imuTime[k]; sX[k] = P.d.imuRotX[k]; sY[k] = P.d.imuRotY[k]; sZ[k] = P.d.imuRotZ[k];
        }
    }
    __syncthreads();
    if (P.d.do_deskew && threadIdx.x == 0) {
        const int fi = *first_idx;                           // firstPointFlag: the first point that reaches deskewPoint
        float inv[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        if (fi < P.d.n) {
            const float tf = lio_rec_time(P.d, P.d.pts + (size_t)fi * P.d.stride);
            float rx, ry, rz, L[9];
            lio_find_rotation(P.d.time_scan_cur + (double)tf, sT, sX, sY, sZ, P.d.imuPointerCur, rx, ry, rz);
            lio_rot_rpy(rx, ry, rz, L);
            lio_inv3(L, inv);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) s_inv[k] = inv[k];
    }
    __syncthreads();
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= P.d.N_SCAN * P.H) return;
    const int i = cell_first[cell];
    if (i == 0x7fffffff) return;
    const int pos = cell_rank[cell];
    const unsigned char* rec = P.d.pts + (size_t)i * P.d.stride;
    const float* f = reinterpret_cast<const float*>(rec);
    const float px = f[0], py = f[1], pz = f[2], pi = f[4];
    float ox = px, oy = py, oz = pz;
    if (P.d.do_deskew) {
        float rx, ry, rz, L[9], Bt[9];
        lio_find_rotation(P.d.time_scan_cur + (double)f[6], sT, sX, sY, sZ, P.d.imuPointerCur, rx, ry, rz);
        lio_rot_rpy(rx, ry, rz, L);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Bt[r * 3 + c] = s_inv[r * 3 + 0] * L[0 * 3 + c] + s_inv[r * 3 + 1] * L[1 * 3 + c]
                              + s_inv[r * 3 + 2] * L[2 * 3 + c];
        ox = Bt[0] * px + Bt[1] * py + Bt[2] * pz + 0.0f;
        oy = Bt[3] * px + Bt[4] * py + Bt[5] * pz + 0.0f;
        oz = Bt[6] * px + Bt[7] * py + Bt[8] * pz + 0.0f;
    }
    float* o = reinterpret_cast<float*>(out + (size_t)pos * out_stride);
    o[0] = ox; o[1] = oy; o[2] = oz; o[3] = 1.0f; o[4] = pi;
    col_out[pos] = cell % P.H;
    range_out[pos] = sqrtf(px * px + py * py + pz * pz);
}

__global__ void k_ri_fill(int* __restrict__ p, int n, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_ri_flags(const int* __restrict__ cell_first, int n_cells, int* __restrict__ flag)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cells) flag[c] = cell_first[c] != 0x7fffffff ? 1 : 0;
}

// startRingIndex = (points before the ring) - 1 + 5, endRingIndex = (points up to and including the ring) - 1 - 5
__global__ void k_ri_rings(const int* __restrict__ cell_rank, int n_scan, int H, int* __restrict__ start, int* __restrict__ end)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_scan) return;
    start[i] = cell_rank[i * H] - 1 + 5;
    end[i] = cell_rank[(i + 1) * H] - 1 - 5;
}

// K2: 11-tap range curvature, summation order exactly as FE:86-91
__global__ __launch_bounds__(256) void k_curvature(const float* __restrict__ r, int n, float* __restrict__ curv,
                                                   int* __restrict__ picked, int* __restrict__ label)
{
    __shared__ float s[256 + 10];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int base = blockIdx.x * 256 - 5;
    for (int k = threadIdx.x; k < 266; k += 256) {
        const int g = base + k;
        s[k] = (g >= 0 && g < n) ? r[g] : 0.0f;
    }
    __syncthreads();
    if (i < 5 || i >= n - 5) return;                                     // FE:84
    const float* c = &s[threadIdx.x + 5];
    const float d = c[-5] + c[-4] + c[-3] + c[-2] + c[-1] - c[0] * 10
                  + c[1] + c[2] + c[3] + c[4] + c[5];                    // FE:86-91
    curv[i] = d * d;                                                     // FE:93
    if (picked) picked[i] = 0;                                           // FE:95
    if (label) label[i] = 0;                                             // FE:96
}

// ------------------------------------------------------------------ C ABI
extern "C" void lio_deskew_default_config(lio_deskew_config* c)
{
    memset(c, 0, sizeof(*c));
    c->N_SCAN = 16;              // UT:275
    c->downsampleRate = 1;       // UT:277
    c->point_filter_num = 3;     // UT:278
    c->lidarMinFront = 1.0f; c->lidarMinBack = 5.0f; c->lidarMinLeft = 2.0f; c->lidarMinRight = 2.0f;  // UT:280-283
    c->lidarMaxRange = 1000.0f;  // UT:284
    c->lidarMaxIntensity = 100.0f; // UT:285
    c->deskew_flag = 1;
    c->device_id = 0;
}

typedef LioTemp DevBuf;      // temporaries come from the recycling pool (lio_pool.h)

static const lio_pc2_layout kXyzirtLayout = { 32, 0, 16, 20, LIO_PC2_UINT16, 24, LIO_PC2_TIME_F32_SECONDS, 0 };

static int lio_check_layout(const lio_pc2_layout* L, bool need_ring)
{
    if (!L) return lio_fail_ext(LIO_ERR_ARG, "null layout", hipSuccess);
    // offsets come off the wire: every bound is written without an addition that could wrap in 32 bits
    // (`off_x + 12 > st` passes for off_x = 0xfffffff4 -- round-2 advisor finding)
    const uint32_t st = L->point_step;
    if (st < 12 || (L->off_x & 3) || L->off_x > st - 12)
        return lio_fail_ext(LIO_ERR_ARG, "x, y, z must be three consecutive FLOAT32 fields inside the record", hipSuccess);
    if (L->off_intensity >= 0 && ((L->off_intensity & 3) || (uint32_t)L->off_intensity > st - 4))
        return lio_fail_ext(LIO_ERR_ARG, "intensity must be an aligned FLOAT32 field inside the record", hipSuccess);
    if (need_ring) {
        const uint32_t rs = L->ring_type == LIO_PC2_UINT8 ? 1 : (L->ring_type == LIO_PC2_UINT16 ? 2 : (L->ring_type == LIO_PC2_INT32 ? 4 : 0));
        if (!rs || L->off_ring < 0 || ((uint32_t)L->off_ring % rs) || (uint32_t)L->off_ring > st - rs)
            return lio_fail_ext(LIO_ERR_ARG, "ring must be an aligned UINT8 / UINT16 / INT32 field inside the record (IP:313-329)", hipSuccess);
        if (L->off_time >= 0) {
            const uint32_t ts = L->time_type == LIO_PC2_TIME_F64_STAMP ? 8 : 4;
            if (L->time_type < 0 || L->time_type > 3 || (L->off_time & 3) || st < ts || (uint32_t)L->off_time > st - ts)
                return lio_fail_ext(LIO_ERR_ARG, "time field outside the record or of an unknown type", hipSuccess);
        }
    }
    return LIO_OK;
}

static void lio_fill_layout(LioDeskewParams& P, const lio_pc2_layout& L)
{
    P.off_xyz = (int)L.off_x; P.off_intensity = L.off_intensity; P.off_ring = L.off_ring; P.off_time = L.off_time;
    P.ring_type = L.ring_type == LIO_PC2_UINT8 ? 1 : (L.ring_type == LIO_PC2_INT32 ? 2 : 0);
    P.time_type = L.time_type;
}

static int lio_deskew_impl(const lio_deskew_config* cfg, const void* pts, size_t n, const lio_pc2_layout& L,
                           double time_scan_cur,
                           const double* imuTime, const double* imuRotX, const double* imuRotY,
                           const double* imuRotZ, int32_t imuPointerCur,
                           void* out, size_t out_stride, size_t* n_out);

extern "C" int lio_deskew(const lio_deskew_config* cfg, const void* pts, size_t n, size_t stride,
                          double time_scan_cur,
                          const double* imuTime, const double* imuRotX, const double* imuRotY,
                          const double* imuRotZ, int32_t imuPointerCur,
                          void* out, size_t out_stride, size_t* n_out)
{
    if (stride < 28 || (stride & 3))
        return lio_fail_ext(LIO_ERR_ARG, "PointXYZIRT stride must be >= 28, output stride >= 20, multiples of 4", hipSuccess);
    lio_pc2_layout L = kXyzirtLayout;
    L.point_step = (uint32_t)stride;
    return lio_deskew_impl(cfg, pts, n, L, time_scan_cur, imuTime, imuRotX, imuRotY, imuRotZ, imuPointerCur, out, out_stride, n_out);
}

// projectPointCloud IP:577-615 straight from the `data` blob of the incoming sensor_msgs/PointCloud2: replaces
// pcl::moveFromROSMsg and the per-sensor conversion loops of cachePointCloud IP:226-285 as well.
extern "C" int lio_deskew_pc2(const lio_deskew_config* cfg, const void* data, size_t n_points, const lio_pc2_layout* layout,
                              double time_scan_cur,
                              const double* imuTime, const double* imuRotX, const double* imuRotY,
                              const double* imuRotZ, int32_t imuPointerCur,
                              void* out, size_t out_stride, size_t* n_out)
{
    const int rc = lio_check_layout(layout, true);
    if (rc != LIO_OK) return rc;
    lio_deskew_config c2;
    if (cfg) { c2 = *cfg; if (layout->off_time < 0) c2.deskew_flag = -1; }   // no per-point time field: IP:341-356 sets deskewFlag = -1
    bool pinned = false;
    if (layout->pin_host && data && n_points)
        pinned = hipHostRegister(const_cast<void*>(data), n_points * layout->point_step, hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
    const int r = lio_deskew_impl(cfg ? &c2 : nullptr, data, n_points, *layout, time_scan_cur, imuTime, imuRotX, imuRotY, imuRotZ,
                                  imuPointerCur, out, out_stride, n_out);
    if (pinned) (void)hipHostUnregister(const_cast<void*>(data));
    return r;
}

static int lio_deskew_impl(const lio_deskew_config* cfg, const void* pts, size_t n, const lio_pc2_layout& L,
                           double time_scan_cur,
                           const double* imuTime, const double* imuRotX, const double* imuRotY,
                           const double* imuRotZ, int32_t imuPointerCur,
                           void* out, size_t out_stride, size_t* n_out)
{
    const size_t stride = L.point_step;
    if (!cfg || !n_out || (n && (!pts || !out))) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if (out_stride < 20 || (out_stride & 3))
        return lio_fail_ext(LIO_ERR_ARG, "PointXYZIRT stride must be >= 28, output stride >= 20, multiples of 4", hipSuccess);
    if (cfg->downsampleRate < 1 || cfg->point_filter_num < 1)
        return lio_fail_ext(LIO_ERR_ARG, "downsampleRate and point_filter_num must be >= 1", hipSuccess);
    if (n > 0x7fffffffull - 1024) return lio_fail_ext(LIO_ERR_CAPACITY, "cloud too large", hipSuccess);
    const bool imu_available = imuPointerCur > 0;                        // IP:414-417
    const bool do_deskew = !(cfg->deskew_flag == -1 || !imu_available);  // IP:547
    if (do_deskew && (!imuTime || !imuRotX || !imuRotY || !imuRotZ || imuPointerCur >= 2000))
        return lio_fail_ext(LIO_ERR_ARG, "IMU tables missing or imuPointerCur >= 2000 (IP:62)", hipSuccess);
    *n_out = 0;
    if (n == 0) return LIO_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lio_fail_ext(LIO_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)", hipSuccess);
    HIPCHK(hipSetDevice(cfg->device_id));
    (void)hipGetLastError();

    const int nb = (int)((n + 255) / 256);
    const int nt = do_deskew ? imuPointerCur + 1 : 0;
    DevBuf d_pts, d_keep, d_cnt, d_misc, d_imu, d_out;
    HIPCHK(d_pts.alloc(n * stride));
    HIPCHK(d_keep.alloc(n));
    HIPCHK(d_cnt.alloc(sizeof(int) * (size_t)nb));
    HIPCHK(d_misc.alloc(sizeof(int) * 2));
    HIPCHK(d_imu.alloc(sizeof(double) * 4 * (size_t)(nt ? nt : 1)));
    HIPCHK(d_out.alloc(n * out_stride));
    hipStream_t s = nullptr;
    HIPCHK(hipMemcpyAsync(d_pts.p, pts, n * stride, hipMemcpyDefault, s));
    double* di = d_imu.as<double>();
    if (nt) {
        HIPCHK(hipMemcpyAsync(di, imuTime, sizeof(double) * nt, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(di + nt, imuRotX, sizeof(double) * nt, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(di + 2 * nt, imuRotY, sizeof(double) * nt, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(di + 3 * nt, imuRotZ, sizeof(double) * nt, hipMemcpyHostToDevice, s));
    }
    int init[2] = { 0x7fffffff, 0 };
    HIPCHK(hipMemcpyAsync(d_misc.p, init, sizeof(init), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d_out.p, 0, n * out_stride, s));

    LioDeskewParams P;
    P.pts = d_pts.as<unsigned char>(); P.stride = stride; P.n = (int)n;
    lio_fill_layout(P, L);
    P.N_SCAN = cfg->N_SCAN; P.downsampleRate = cfg->downsampleRate; P.point_filter_num = cfg->point_filter_num;
    P.minFront = cfg->lidarMinFront; P.minBack = cfg->lidarMinBack; P.minLeft = cfg->lidarMinLeft;
    P.minRight = cfg->lidarMinRight; P.maxRange = cfg->lidarMaxRange; P.maxIntensity = cfg->lidarMaxIntensity;
    P.do_deskew = do_deskew ? 1 : 0; P.time_scan_cur = time_scan_cur;
    P.imuTime = di; P.imuRotX = di + nt; P.imuRotY = di + 2 * nt; P.imuRotZ = di + 3 * nt;
    P.imuPointerCur = do_deskew ? imuPointerCur : 0;

    int* misc = d_misc.as<int>();
    hipLaunchKernelGGL(k_deskew_flags, dim3(nb), dim3(256), 0, s, P, d_keep.as<unsigned char>(), d_cnt.as<int>(), misc);
    hipLaunchKernelGGL(k_deskew_scan, dim3(1), dim3(256), 0, s, d_cnt.as<int>(), nb, misc + 1);
    const size_t lds = 64 + sizeof(double) * 4 * (size_t)(nt ? nt : 1);
    hipLaunchKernelGGL(k_deskew_emit, dim3(nb), dim3(256), lds, s, P, d_keep.as<unsigned char>(), d_cnt.as<int>(),
                       misc, d_out.as<unsigned char>(), out_stride);
    int res[2];
    HIPCHK(hipMemcpyAsync(res, misc, sizeof(res), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    *n_out = (size_t)res[1];
    if (res[1] > 0) HIPCHK(hipMemcpy(out, d_out.p, (size_t)res[1] * out_stride, hipMemcpyDeviceToHost));
    return LIO_OK;
}

extern "C" int lio_curvature(int32_t device_id, const float* range, size_t n, float* curvature,
                             int32_t* neighbor_picked, int32_t* label)
{
    if (n && (!range || !curvature)) return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if (n > 0x7fffffffull - 1024) return lio_fail_ext(LIO_ERR_CAPACITY, "array too large", hipSuccess);
    if (n < 11) return LIO_OK;                                           // FE:84: empty loop
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lio_fail_ext(LIO_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)", hipSuccess);
    HIPCHK(hipSetDevice(device_id));
    (void)hipGetLastError();
    DevBuf d_r, d_c, d_p, d_l;
    HIPCHK(d_r.alloc(n * 4)); HIPCHK(d_c.alloc(n * 4));
    if (neighbor_picked) HIPCHK(d_p.alloc(n * 4));
    if (label) HIPCHK(d_l.alloc(n * 4));
    hipStream_t s = nullptr;
    HIPCHK(hipMemcpyAsync(d_r.p, range, n * 4, hipMemcpyHostToDevice, s));
    // untouched entries (i < 5, i >= n-5) keep the caller's values
    HIPCHK(hipMemcpyAsync(d_c.p, curvature, n * 4, hipMemcpyHostToDevice, s));
    if (neighbor_picked) HIPCHK(hipMemcpyAsync(d_p.p, neighbor_picked, n * 4, hipMemcpyHostToDevice, s));
    if (label) HIPCHK(hipMemcpyAsync(d_l.p, label, n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_curvature, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       d_r.as<float>(), (int)n, d_c.as<float>(), d_p.as<int>(), d_l.as<int>());
    HIPCHK(hipMemcpyAsync(curvature, d_c.p, n * 4, hipMemcpyDeviceToHost, s));
    if (neighbor_picked) HIPCHK(hipMemcpyAsync(neighbor_picked, d_p.p, n * 4, hipMemcpyDeviceToHost, s));
    if (label) HIPCHK(hipMemcpyAsync(label, d_l.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return LIO_OK;
}

void lio_launch_exclusive_scan(const int* in, int n, int* tile_sums, int* out, hipStream_t s);   // lio_kernels.hip
int  lio_scan_tiles(int n_cells);

extern "C" void lio_range_image_default_config(lio_range_image_config* c)
{
    memset(c, 0, sizeof(*c));
    c->N_SCAN = 16;
    c->Horizon_SCAN = 1800;
    c->downsampleRate = 1;
    c->lidarMinRange = 1.0f;       // upstream params.yaml
    c->lidarMaxRange = 1000.0f;
    c->deskew_flag = 1;
    c->device_id = 0;
}

extern "C" int lio_range_image(const lio_range_image_config* cfg, const void* pts, size_t n, size_t stride,
                               double time_scan_cur,
                               const double* imuTime, const double* imuRotX, const double* imuRotY,
                               const double* imuRotZ, int32_t imuPointerCur,
                               void* out, size_t out_stride, size_t* n_out,
                               int32_t* startRingIndex, int32_t* endRingIndex,
                               int32_t* pointColInd, float* pointRange)
{
    if (!cfg || !n_out || !startRingIndex || !endRingIndex || (n && !pts))
        return lio_fail_ext(LIO_ERR_ARG, "null argument", hipSuccess);
    if (stride < 28 || (stride & 3) || out_stride < 20 || (out_stride & 3))
        return lio_fail_ext(LIO_ERR_ARG, "PointXYZIRT stride must be >= 28, output stride >= 20, multiples of 4", hipSuccess);
    if (cfg->N_SCAN < 1 || cfg->N_SCAN > 1024 || cfg->Horizon_SCAN < 1 || cfg->Horizon_SCAN > 32767 || cfg->downsampleRate < 1)
        return lio_fail_ext(LIO_ERR_ARG, "N_SCAN in 1..1024, Horizon_SCAN in 1..32767, downsampleRate >= 1", hipSuccess);
    if (n > 0x7fffffffull - 1024) return lio_fail_ext(LIO_ERR_CAPACITY, "cloud too large", hipSuccess);
    const bool imu_available = imuPointerCur > 0;
    const bool do_deskew = !(cfg->deskew_flag == -1 || !imu_available);
    if (do_deskew && (!imuTime || !imuRotX || !imuRotY || !imuRotZ || imuPointerCur >= 2000))
        return lio_fail_ext(LIO_ERR_ARG, "IMU tables missing or imuPointerCur >= 2000 (IP:62)", hipSuccess);
    const int ns = cfg->N_SCAN, H = cfg->Horizon_SCAN, cells = ns * H;
    *n_out = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lio_fail_ext(LIO_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)", hipSuccess);
    HIPCHK(hipSetDevice(cfg->device_id));
    (void)hipGetLastError();

    const int nt = do_deskew ? imuPointerCur + 1 : 0;
    DevBuf d_pts, d_first, d_flag, d_rank, d_tiles, d_misc, d_imu, d_out, d_col, d_range, d_rings;
    HIPCHK(d_pts.alloc((n ? n : 1) * stride));
    HIPCHK(d_first.alloc(sizeof(int) * (size_t)cells));
    HIPCHK(d_flag.alloc(sizeof(int) * (size_t)cells));
    HIPCHK(d_rank.alloc(sizeof(int) * ((size_t)cells + 1)));
    HIPCHK(d_tiles.alloc(sizeof(int) * ((size_t)lio_scan_tiles(cells) + 1)));
    HIPCHK(d_misc.alloc(sizeof(int)));
    HIPCHK(d_imu.alloc(sizeof(double) * 4 * (size_t)(nt ? nt : 1)));
    HIPCHK(d_out.alloc((size_t)cells * out_stride));
    HIPCHK(d_col.alloc(sizeof(int) * (size_t)cells));
    HIPCHK(d_range.alloc(sizeof(float) * (size_t)cells));
    HIPCHK(d_rings.alloc(sizeof(int) * 2 * (size_t)ns));
    hipStream_t s = nullptr;
    if (n) HIPCHK(hipMemcpyAsync(d_pts.p, pts, n * stride, hipMemcpyHostToDevice, s));
    double* di = d_imu.as<double>();
    if (nt) {
        HIPCHK(hipMemcpyAsync(di, imuTime, sizeof(double) * nt, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(di + nt, imuRotX, sizeof(double) * nt, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(di + 2 * nt, imuRotY, sizeof(double) * nt, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(di + 3 * nt, imuRotZ, sizeof(double) * nt, hipMemcpyHostToDevice, s));
    }
    const int big = 0x7fffffff;
    HIPCHK(hipMemcpyAsync(d_misc.p, &big, sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d_out.p, 0, (size_t)cells * out_stride, s));

    LioRangeImageParams P;
    P.d.pts = d_pts.as<unsigned char>(); P.d.stride = stride; P.d.n = (int)n;
    P.d.off_xyz = 0; P.d.off_intensity = 16; P.d.off_ring = 20; P.d.off_time = 24; P.d.ring_type = 0; P.d.time_type = 0;
    P.d.N_SCAN = ns; P.d.downsampleRate = cfg->downsampleRate; P.d.point_filter_num = 1;
    P.d.minFront = P.d.minBack = P.d.minLeft = P.d.minRight = 0.0f;
    P.d.maxRange = cfg->lidarMaxRange; P.d.maxIntensity = 0.0f;
    P.d.do_deskew = do_deskew ? 1 : 0; P.d.time_scan_cur = time_scan_cur;
    P.d.imuTime = di; P.d.imuRotX = di + nt; P.d.imuRotY = di + 2 * nt; P.d.imuRotZ = di + 3 * nt;
    P.d.imuPointerCur = do_deskew ? imuPointerCur : 0;
    P.H = H; P.minRange = cfg->lidarMinRange;

    const int nb = (int)((n + 255) / 256), nc = (cells + 255) / 256;
    hipLaunchKernelGGL(k_ri_fill, dim3(nc), dim3(256), 0, s, d_first.as<int>(), cells, 0x7fffffff);   // rangeMat = FLT_MAX
    if (n) hipLaunchKernelGGL(k_ri_first, dim3(nb), dim3(256), 0, s, P, d_first.as<int>(), d_misc.as<int>());
    hipLaunchKernelGGL(k_ri_flags, dim3(nc), dim3(256), 0, s, d_first.as<int>(), cells, d_flag.as<int>());
    lio_launch_exclusive_scan(d_flag.as<int>(), cells, d_tiles.as<int>(), d_rank.as<int>(), s);
    const size_t lds = 64 + sizeof(double) * 4 * (size_t)(nt ? nt : 1);
    hipLaunchKernelGGL(k_ri_extract, dim3(nc), dim3(256), lds, s, P, d_first.as<int>(), d_rank.as<int>(), d_misc.as<int>(),
                       d_out.as<unsigned char>(), out_stride, d_col.as<int>(), d_range.as<float>());
    hipLaunchKernelGGL(k_ri_rings, dim3((ns + 255) / 256), dim3(256), 0, s, d_rank.as<int>(), ns, H,
                       d_rings.as<int>(), d_rings.as<int>() + ns);
    int total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_rank.as<int>() + cells, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(startRingIndex, d_rings.p, sizeof(int) * (size_t)ns, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(endRingIndex, d_rings.as<int>() + ns, sizeof(int) * (size_t)ns, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    *n_out = (size_t)total;
    if (total > 0) {
        if (out) HIPCHK(hipMemcpy(out, d_out.p, (size_t)total * out_stride, hipMemcpyDeviceToHost));
        if (pointColInd) HIPCHK(hipMemcpy(pointColInd, d_col.p, sizeof(int) * (size_t)total, hipMemcpyDeviceToHost));
        if (pointRange) HIPCHK(hipMemcpy(pointRange, d_range.p, sizeof(float) * (size_t)total, hipMemcpyDeviceToHost));
    }
    return LIO_OK;
}
