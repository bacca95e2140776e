// lio_device_math.h -- per-thread fp32 arithmetic of the scan-to-map path for
// gfx950.  Compiled with -ffp-contract=off: every expression below keeps the
// reference's operator order so that results are bit-identical to a CPU build
// of the reference without FMA contraction (src/liorf/CMakeLists.txt:7).
//
//   MO = /root/reference/src/liorf/src/mapOptmization.cpp
//
// Everything here lives in registers: loops have compile-time trip counts and
// are fully unrolled, pivot swaps are selects, so no scratch memory is touched.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#define LIO_DEV __device__ __forceinline__

// ---------------------------------------------------------------- plane fit
// matA0.colPivHouseholderQr().solve(matB0) with matB0 = -1, MO:1633-1648
// (Eigen 3.3 ColPivHouseholderQR: pivot on the largest updated column norm,
// LAPACK-style norm downdating, makeHouseholderInPlace, column-oriented
// back-substitution over nonzeroPivots()).  a[r][c]: 5 neighbours x (x,y,z).
LIO_DEV void lio_plane_qr5x3(float a[5][3], float x[3])
{
    float hc[3], direct[3], upd[3];
    int trans[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; ++i) s += a[i][k] * a[i][k];
        direct[k] = sqrtf(s);
        upd[k] = direct[k];
    }
    float maxn = upd[0];
    if (upd[1] > maxn) maxn = upd[1];
    if (upd[2] > maxn) maxn = upd[2];
    const float th = maxn * FLT_EPSILON;
    const float threshold_helper = (th * th) / 5.0f;
    const float downdate_thr = sqrtf(FLT_EPSILON);
    int nz = 3;

#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int big = k;
        float bigv = upd[k];
#pragma unroll
        for (int j = k + 1; j < 3; ++j)
            if (upd[j] > bigv) { bigv = upd[j]; big = j; }
        const float big_sq = bigv * bigv;
        if (nz == 3 && big_sq < threshold_helper * (float)(5 - k)) nz = k;
        trans[k] = big;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            const bool sw = (big == j);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float u = a[i][k], v = a[i][j];
                a[i][k] = sw ? v : u;
                a[i][j] = sw ? u : v;
            }
            const float u0 = upd[k], u1 = upd[j], d0 = direct[k], d1 = direct[j];
            upd[k] = sw ? u1 : u0;       upd[j] = sw ? u0 : u1;
            direct[k] = sw ? d1 : d0;    direct[j] = sw ? d0 : d1;
        }
        float tail_sq = 0.0f;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) tail_sq += a[i][k] * a[i][k];
        const float c0 = a[k][k];
        float beta, tau;
        if (tail_sq <= FLT_MIN) {
            tau = 0.0f; beta = c0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][k] = 0.0f;
        } else {
            beta = sqrtf(c0 * c0 + tail_sq);
            if (c0 >= 0.0f) beta = -beta;
            const float den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][k] = a[i][k] / den;
            tau = (beta - c0) / beta;
        }
        hc[k] = tau;
        a[k][k] = beta;
        if (tau != 0.0f) {
#pragma unroll
            for (int j = k + 1; j < 3; ++j) {
                float tmp = 0.0f;
#pragma unroll
                for (int i = k + 1; i < 5; ++i) tmp += a[i][k] * a[i][j];
                tmp += a[k][j];
                a[k][j] -= tau * tmp;
#pragma unroll
                for (int i = k + 1; i < 5; ++i) a[i][j] -= (tau * a[i][k]) * tmp;
            }
        }
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (upd[j] != 0.0f) {
                float temp = fabsf(a[k][j]) / upd[j];
                temp = (1.0f + temp) * (1.0f - temp);
                temp = temp < 0.0f ? 0.0f : temp;
                const float ratio = upd[j] / direct[j];
                const float temp2 = temp * (ratio * ratio);
                if (temp2 <= downdate_thr) {
                    float s = 0.0f;
#pragma unroll
                    for (int i = k + 1; i < 5; ++i) s += a[i][j] * a[i][j];
                    direct[j] = sqrtf(s);
                    upd[j] = direct[j];
                } else {
                    upd[j] *= sqrtf(temp);
                }
            }
        }
    }
    // permutation from the transpositions: perm = identity; swap(perm[k], perm[trans[k]])
    int p0 = 0, p1 = 1, p2 = 2;
    {   // k = 0
        const int t = trans[0];
        const int q0 = p0, q1 = p1, q2 = p2;
        p0 = (t == 1) ? q1 : (t == 2) ? q2 : q0;
        p1 = (t == 1) ? q0 : q1;
        p2 = (t == 2) ? q0 : q2;
    }
    {   // k = 1
        const int t = trans[1];
        const int q1 = p1, q2 = p2;
        p1 = (t == 2) ? q2 : q1;
        p2 = (t == 2) ? q1 : q2;
    }
    // k = 2: trans[2] == 2 always

    // c = Q^T b with b = -1 (matB0.fill(-1), MO:1638), first nz reflectors
    float c[5] = { -1.0f, -1.0f, -1.0f, -1.0f, -1.0f };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < nz) {
            const float tau = hc[k];
            if (tau != 0.0f) {
                float tmp = 0.0f;
#pragma unroll
                for (int i = k + 1; i < 5; ++i) tmp += a[i][k] * c[i];
                tmp += c[k];
                c[k] -= tau * tmp;
#pragma unroll
                for (int i = k + 1; i < 5; ++i) c[i] -= (tau * a[i][k]) * tmp;
            }
        }
    }
    // upper-triangular solve on the leading nz x nz block, column oriented
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        if (i < nz) {
            if (c[i] != 0.0f) {
                c[i] /= a[i][i];
#pragma unroll
                for (int r = 0; r < i; ++r) c[r] -= c[i] * a[r][i];
            }
        }
    }
    const float v0 = (0 < nz) ? c[0] : 0.0f;
    const float v1 = (1 < nz) ? c[1] : 0.0f;
    const float v2 = (2 < nz) ? c[2] : 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        x[j] = (p0 == j) ? v0 : (p1 == j) ? v1 : (p2 == j) ? v2 : 0.0f;
}

// --------------------------------------------------------------- Jacobian
// One row of matA and matB, MO:1760-1778.  tr = {srx,crx,sry,cry,srz,crz} with
// the reference's names: rx <- yaw, ry <- pitch, rz <- roll (MO:1714-1719).
// jac_exact swaps the MO:1764 term for the analytic derivative.
LIO_DEV void lio_jacobian_row(const float tr[6], float px, float py, float pz,
                              float cx, float cy, float cz, float cw, int jac_exact,
                              float row[6], float &rhs)
{
    const float srx = tr[0], crx = tr[1], sry = tr[2], cry = tr[3], srz = tr[4], crz = tr[5];
    const float arx = (-srx * cry * px - (srx * sry * srz + crx * crz) * py + (crx * srz - srx * sry * crz) * pz) * cx
                    + (crx * cry * px - (srx * crz - crx * sry * srz) * py + (crx * sry * crz + srx * srz) * pz) * cy;
    const float mid = jac_exact ? (srx * cry * srz * py) : (srx * sry * srz * py);
    const float ary = (-crx * sry * px + crx * cry * srz * py + crx * cry * crz * pz) * cx
                    + (-srx * sry * px + mid + srx * cry * crz * pz) * cy
                    + (-cry * px - sry * srz * py - sry * crz * pz) * cz;
    const float arz = ((crx * sry * crz + srx * srz) * py + (srx * crz - crx * sry * srz) * pz) * cx
                    + ((-crx * srz + srx * sry * crz) * py + (-srx * sry * srz - crx * crz) * pz) * cy
                    + (cry * crz * py - cry * srz * pz) * cz;
    row[0] = arz; row[1] = ary; row[2] = arx;
    row[3] = cx;  row[4] = cy;  row[5] = cz;
    rhs = -cw;
}

__device__ static float lio_cv_hypot(float a, float b)
{
    a = fabsf(a); b = fabsf(b);
    if (a > b) { b /= a; return a * sqrtf(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrtf(1 + a * a); }
    return 0.0f;
}

// ------------------------------------------------- point-to-line residual (EXTENSION)
// BASELINE.json's north_star names cornerOptimization; this reference has none, so
// this follows upstream LIO-SAM's cornerOptimization and is checked against oracle/lio_oracle.c
// lo_corner_point only ("parity unpinned": no reference fixture exists for it).
//
// cv::eigen on the symmetric 3x3 covariance: OpenCV's JacobiImpl_ specialised to n = 3 and held
// entirely in registers.  Only the upper triangle (a01, a02, a12) is ever touched; of the pivot
// trackers only indR[0] (r0) and indC[2] (c2) have a choice, indR[1] = 2 and indC[1] = 0 always.
// They are refreshed exactly when the general algorithm refreshes them (rows/columns k and l of
// the rotation just applied), including its stale-tracker behaviour.
// Returns the two largest eigenvalues and the eigenvector (row 0 after the descending sort).
LIO_DEV void lio_eigen3_sym(float a00, float a01, float a02, float a11, float a12, float a22,
                            float& e0, float& e1, float axis[3])
{
    float w0 = a00, w1 = a11, w2 = a22;
    float v00 = 1.0f, v01 = 0.0f, v02 = 0.0f, v10 = 0.0f, v11 = 1.0f, v12 = 0.0f, v20 = 0.0f, v21 = 0.0f, v22 = 1.0f;
    int r0 = (fabsf(a01) < fabsf(a02)) ? 2 : 1;
    int c2 = (fabsf(a02) < fabsf(a12)) ? 1 : 0;
#pragma unroll 1
    for (int iters = 0; iters < 3 * 3 * 30; ++iters) {
        // pivot: rows first (k = 0 via indR[0], k = 1 via indR[1] = 2), then columns 1 and 2
        int pair = (r0 == 1) ? 0 : 1;                       // (0,1) -> 0, (0,2) -> 1, (1,2) -> 2
        float mv = fabsf(r0 == 1 ? a01 : a02);
        if (mv < fabsf(a12)) { mv = fabsf(a12); pair = 2; }
        if (mv < fabsf(a01)) { mv = fabsf(a01); pair = 0; }
        const float vc = (c2 == 0) ? a02 : a12;
        if (mv < fabsf(vc)) { mv = fabsf(vc); pair = (c2 == 0) ? 1 : 2; }
        const float p = pair == 0 ? a01 : (pair == 1 ? a02 : a12);
        if (fabsf(p) <= FLT_EPSILON) break;
        const float wk = pair == 2 ? w1 : w0, wl = pair == 0 ? w1 : w2;
        const float y = (wl - wk) * 0.5f;                    // == (float)((W[l] - W[k]) * 0.5)
        float t = fabsf(y) + lio_cv_hypot(p, y);
        float s = lio_cv_hypot(p, t);
        const float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) { s = -s; t = -t; }
        float a0, b0;
#define LIO_ROT(u0, u1) do { a0 = (u0); b0 = (u1); (u0) = a0 * c - b0 * s; (u1) = a0 * s + b0 * c; } while (0)
        if (pair == 0) {            // (k,l) = (0,1), third index 2 > l: rotate (A[k][2], A[l][2])
            a01 = 0.0f; w0 -= t; w1 += t;
            LIO_ROT(a02, a12);
            LIO_ROT(v00, v10); LIO_ROT(v01, v11); LIO_ROT(v02, v12);
            r0 = (fabsf(a01) < fabsf(a02)) ? 2 : 1;
        } else if (pair == 1) {     // (0,2), third index 1 in between: rotate (A[k][1], A[1][l])
            a02 = 0.0f; w0 -= t; w2 += t;
            LIO_ROT(a01, a12);
            LIO_ROT(v00, v20); LIO_ROT(v01, v21); LIO_ROT(v02, v22);
            r0 = (fabsf(a01) < fabsf(a02)) ? 2 : 1;
            c2 = (fabsf(a02) < fabsf(a12)) ? 1 : 0;
        } else {                    // (1,2), third index 0 < k: rotate (A[0][k], A[0][l])
            a12 = 0.0f; w1 -= t; w2 += t;
            LIO_ROT(a01, a02);
            LIO_ROT(v10, v20); LIO_ROT(v11, v21); LIO_ROT(v12, v22);
            c2 = (fabsf(a02) < fabsf(a12)) ? 1 : 0;
        }
#undef LIO_ROT
    }
    // descending selection sort (first maximum wins), eigenvectors are rows
    int m = 0;
    if (w0 < w1) m = 1;
    if ((m == 0 ? w0 : w1) < w2) m = 2;
    float t0;
#define LIO_SWAP(x, y) do { t0 = (x); (x) = (y); (y) = t0; } while (0)
    if (m == 1) { LIO_SWAP(w0, w1); LIO_SWAP(v00, v10); LIO_SWAP(v01, v11); LIO_SWAP(v02, v12); }
    if (m == 2) { LIO_SWAP(w0, w2); LIO_SWAP(v00, v20); LIO_SWAP(v01, v21); LIO_SWAP(v02, v22); }
    if (w1 < w2) LIO_SWAP(w1, w2);
#undef LIO_SWAP
    e0 = w0; e1 = w1;
    axis[0] = v00; axis[1] = v01; axis[2] = v02;
}

// Upstream LIO-SAM cornerOptimization, after the 5-NN gate: m = the five neighbours (caller's map
// order), q = the transformed scan point.  Writes coeff = s * (la, lb, lc, ld2); returns s > min_s.
LIO_DEV bool lio_corner_assoc(const float m[5][3], float x0, float y0, float z0, double weight, double min_s,
                              float& ox, float& oy, float& oz, float& ow)
{
    float cx = 0, cy = 0, cz = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) { cx += m[j][0]; cy += m[j][1]; cz += m[j][2]; }
    cx /= 5; cy /= 5; cz /= 5;
    float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float ax = m[j][0] - cx, ay = m[j][1] - cy, az = m[j][2] - cz;
        a11 += ax * ax; a12 += ax * ay; a13 += ax * az;
        a22 += ay * ay; a23 += ay * az;
        a33 += az * az;
    }
    a11 /= 5; a12 /= 5; a13 /= 5; a22 /= 5; a23 /= 5; a33 /= 5;
    float e0, e1, v[3];
    lio_eigen3_sym(a11, a12, a13, a22, a23, a33, e0, e1, v);
    ox = oy = oz = ow = 0.0f;
    if (!(e0 > 3 * e1)) return false;
    const float x1 = (float)((double)cx + 0.1 * (double)v[0]), y1 = (float)((double)cy + 0.1 * (double)v[1]),
                z1 = (float)((double)cz + 0.1 * (double)v[2]);
    const float x2 = (float)((double)cx - 0.1 * (double)v[0]), y2 = (float)((double)cy - 0.1 * (double)v[1]),
                z2 = (float)((double)cz - 0.1 * (double)v[2]);
    const float m1 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1);
    const float m2 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1);
    const float m3 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
    const float a012 = sqrtf(m1 * m1 + m2 * m2 + m3 * m3);
    const float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
    const float la = ((y1 - y2) * m1 + (z1 - z2) * m2) / a012 / l12;
    const float lb = -((x1 - x2) * m1 - (z1 - z2) * m3) / a012 / l12;
    const float lc = -((x1 - x2) * m2 + (y1 - y2) * m3) / a012 / l12;
    const float ld2 = a012 / l12;
    const float s = (float)(1 - weight * (double)fabsf(ld2));
    ox = s * la; oy = s * lb; oz = s * lc; ow = s * ld2;
    return (double)s > min_s;
}

// ------------------------------------------------- 6x6 normal-equation step
// Executed by ONE lane per scan per iteration (the serial tail of the GN
// step).  All working storage is a caller-provided LDS workspace so that the
// association kernel needs no scratch memory.
struct LioSolveWs {      // LDS working storage of lio_gn_step
    float A[36];      // product matV.inv() * matV2 (matP)
    float B[36];      // matV.inv()
    float V2[36];     // eigenvectors with the degenerate rows cleared
    float AtA[36];    // matAtA of the first iteration (input of the eigen-decomposition)
    float X[6], X2[6];
};

// cv::solve(AtA, AtB, X, DECOMP_QR), MO:1784 (OpenCV hal::QR32f: Householder
// with unit-norm reflectors, then back substitution; singular -> X = 0).
// A (6x6 row-major) is destroyed, b is replaced by the solution.  PRIVATE arrays, every loop unrolled, so that the 6x6
// system lives in registers (the solve sits at the end of every Gauss-Newton launch, on the critical path of a lone
// registration; an earlier form that walked it through LDS cost ~10 us per iteration).
__device__ static __forceinline__ int lio_solve6_qr_reg(float (&A)[36], float (&b)[6])
{
    const float eps = FLT_EPSILON * 10;
    float vl[6], hf[6];
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        const int vs = 6 - l;
        float vnorm = 0.0f;
#pragma unroll
        for (int i = 0; i < vs; ++i) { vl[i] = A[(l + i) * 6 + l]; vnorm += vl[i] * vl[i]; }
        const float tmpv = vl[0];
        const float sg = vl[0] >= 0.0f ? 1.0f : -1.0f;
        vl[0] = vl[0] + sg * sqrtf(vnorm);
        vnorm = sqrtf(vnorm + vl[0] * vl[0] - tmpv * tmpv);
#pragma unroll
        for (int i = 0; i < vs; ++i) vl[i] /= vnorm;
#pragma unroll
        for (int j = l; j < 6; ++j) {
            float va = 0.0f;
#pragma unroll
            for (int i = l; i < 6; ++i) va += vl[i - l] * A[i * 6 + j];
#pragma unroll
            for (int i = l; i < 6; ++i) A[i * 6 + j] -= 2 * vl[i - l] * va;
        }
        hf[l] = vl[0] * vl[0];
#pragma unroll
        for (int i = 1; i < vs; ++i) A[(l + i) * 6 + l] = vl[i] / vl[0];
    }
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        vl[0] = 1.0f;
#pragma unroll
        for (int j = 1; j < 6 - l; ++j) vl[j] = A[(j + l) * 6 + l];
        float vb = 0.0f;
#pragma unroll
        for (int i = l; i < 6; ++i) vb += vl[i - l] * b[i];
#pragma unroll
        for (int i = l; i < 6; ++i) b[i] -= 2 * vl[i - l] * vb * hf[l];
    }
    bool singular = false;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        if (!singular) {
#pragma unroll
            for (int j = 5; j > i; --j) b[i] -= b[j] * A[i * 6 + j];
            if (fabsf(A[i * 6 + i]) < eps) singular = true;
            else b[i] /= A[i * 6 + i];
        }
    }
    if (singular) {
#pragma unroll
        for (int p = 0; p < 6; ++p) b[p] = 0.0f;
        return 0;
    }
    return 1;
}

#define LIO_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// cv::eigen(matAtA, matE, matV), MO:1792 (OpenCV JacobiImpl_: largest off-diagonal pivot tracked per row / column through
// indR / indC; eigenvalues sorted descending, eigenvectors as rows), executed by one wave with the matrices in
// REGISTERS, one element per lane (lane e < 36 holds A[e/6][e%6] and V[e/6][e%6], lanes 0..5 also W[lane], indR[lane],
// indC[lane]): no LDS round trips and no fences inside the loop (~34 rotations per registration; a form that kept the
// matrices in LDS cost 1.7 us per rotation, this one 1.06).  Element reads with a wave-uniform
// index are v_readlane, per-lane indices go through ds_bpermute; every floating-point operation and comparison is
// the serial code's:
//   pivot     lanes 0-4 fetch |A[i][indR[i]]|, lanes 1-5 |A[indC[i]][i]|; the serial scan (rows 0..4, then columns 1..5,
//             strict `<`) picks the first row holding the row maximum unless the column maximum is strictly larger, then
//             the first column holding it;
//   rotation  the pairs the serial loops rotate are disjoint, so every lane that holds a member fetches its partner
//             and applies its half of the rotation; the eigenvector rows k and l likewise;
//   trackers  every lane 0..5 recomputes indR / indC of its own row / column from the rotated matrix, lanes k and l
//             keep the result (the serial code recomputes exactly those).
// All lanes of the wave must call it; Ain (36) is read from LDS; the results stay in registers: Wd[0..5] = the eigenvalues
// in descending order (wave-uniform), vrow = element (lane/6, lane%6) of the eigenvector matrix (eigenvectors as rows) in lanes 0..35.
__device__ static void lio_eigen6_sym_lanes(const float* Ain, float (&Wd)[6], float& vrow, int lane)
{
#define LIO_BP(x, idx) __int_as_float(__builtin_amdgcn_ds_bpermute((idx) << 2, __float_as_int(x)))
#define LIO_RL(x, idx) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (idx)))
#define LIO_SHR(x, n) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x110 + (n), 0xf, 0xf, false))
    // indR of row t (first maximum of |A[t][m]|, m > t) and indC of column t (first maximum of |A[i][t]|, i < t)
#define LIO_TRACK(nr, nc) do {                                                                   \
        float mv_ = -1.0f, mc_ = -1.0f; nr = 0; nc = 0;                                          \
        _Pragma("unroll") for (int m_ = 1; m_ < 6; ++m_) {                                       \
            const float val_ = fabsf(LIO_BP(a, t6 + m_));                                        \
            if (m_ == t + 1 || (m_ > t && mv_ < val_)) { mv_ = val_; nr = m_; }                  \
        }                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) {                                       \
            const float val_ = fabsf(LIO_BP(a, i_ * 6 + t));                                     \
            if (i_ < t && (i_ == 0 || mc_ < val_)) { mc_ = val_; nc = i_; }                      \
        }                                                                                        \
    } while (0)
    const float eps = FLT_EPSILON;
    const int r = lane / 6, c = lane - r * 6;
    const bool in = lane < 36;
    const int t = lane < 5 ? lane : 5, t6 = t * 6;
    float a = in ? Ain[lane] : 0.0f;
    float v = (in && r == c) ? 1.0f : 0.0f;
    float w = LIO_BP(a, t * 7);                              // lanes 0..5: W[lane] = A[lane][lane]
    int ir, ic;
    LIO_TRACK(ir, ic);
    for (int iters = 0; iters < 6 * 6 * 30; ++iters) {
        // pivot
        const float rvl = fabsf(LIO_BP(a, t6 + ir)), cvl = fabsf(LIO_BP(a, ic * 6 + t));
        const bool is_row = lane < 5, is_col = lane >= 1 && lane <= 5;
        const float rv = is_row ? rvl : -1.0f, cv = is_col ? cvl : -1.0f;
        float x = rv, y_ = cv;
        x = fmaxf(x, LIO_SHR(x, 1)); y_ = fmaxf(y_, LIO_SHR(y_, 1));
        x = fmaxf(x, LIO_SHR(x, 2)); y_ = fmaxf(y_, LIO_SHR(y_, 2));
        x = fmaxf(x, LIO_SHR(x, 4)); y_ = fmaxf(y_, LIO_SHR(y_, 4));
        const float rmax = LIO_RL(x, 7), cmax = LIO_RL(y_, 7);       // maxima over lanes 0..7 (6, 7 hold -1)
        const int kr = __builtin_ctzll(__ballot(is_row && rv == rmax) | (1ull << 63));
        const int lc = __builtin_ctzll(__ballot(is_col && cv == cmax) | (1ull << 63));
        const int l_row = __builtin_amdgcn_readlane(ir, kr & 7), k_col = __builtin_amdgcn_readlane(ic, lc & 7);
        const bool colwins = cmax > rmax;
        const int k = __builtin_amdgcn_readfirstlane(colwins ? k_col : kr);
        const int l = __builtin_amdgcn_readfirstlane(colwins ? lc : l_row);
        const float p = LIO_RL(a, k * 6 + l);
        if (fabsf(p) <= eps) break;                          // wave-uniform
        const float wk = LIO_RL(w, k), wl = LIO_RL(w, l);
        const float y = (float)((wl - wk) * 0.5);
        float tt = fabsf(y) + lio_cv_hypot(p, y);
        float s = lio_cv_hypot(p, tt);
        const float cc = tt / s;
        s = p / s; tt = (p / tt) * p;
        if (y < 0) { s = -s; tt = -tt; }
        // rotation of the upper triangle: role 1 = first member of a pair (a0), role 2 = second member (b0)
        int role = 0, pe = lane;
        if (c == k && r < k) { role = 1; pe = r * 6 + l; }
        else if (c == l && r < k) { role = 2; pe = r * 6 + k; }
        else if (r == k && c > k && c < l) { role = 1; pe = c * 6 + l; }
        else if (c == l && r > k && r < l) { role = 2; pe = k * 6 + r; }
        else if (r == k && c > l) { role = 1; pe = l * 6 + c; }
        else if (r == l && c > l) { role = 2; pe = k * 6 + c; }
        const float pa = LIO_BP(a, pe);
        const float n0 = a * cc - pa * s, n1 = pa * s + a * cc;
        a = role == 1 ? n0 : (role == 2 ? n1 : a);
        if (lane == k * 6 + l) a = 0;
        // eigenvector rows k and l
        const bool vk = in && r == k, vl = in && r == l;
        const float pv = LIO_BP(v, vk ? l * 6 + c : (vl ? k * 6 + c : lane));
        const float m0 = v * cc - pv * s, m1 = pv * s + v * cc;
        v = vk ? m0 : (vl ? m1 : v);
        if (lane == k) w = wk - tt;
        if (lane == l) w = wl + tt;
        int nr, nc;
        LIO_TRACK(nr, nc);
        if (lane == k || lane == l) { ir = nr; ic = nc; }
    }
    // descending selection sort of the eigenvalues, eigenvectors are rows (the permutation is wave-uniform)
    float W[6];
    int perm[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { W[i] = LIO_RL(w, i); perm[i] = i; }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int m = k;
        float wm = W[k];
        int pm = perm[k];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) if (wm < W[i]) { wm = W[i]; m = i; pm = perm[i]; }
#pragma unroll
        for (int i = k + 1; i < 6; ++i) if (m == i) { W[i] = W[k]; perm[i] = perm[k]; }
        W[k] = wm; perm[k] = pm;
    }
    int pr = perm[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) if (r == i) pr = perm[i];
    vrow = LIO_BP(v, in ? pr * 6 + c : lane);
#pragma unroll
    for (int i = 0; i < 6; ++i) Wd[i] = W[i];
#undef LIO_TRACK
#undef LIO_SHR
#undef LIO_RL
#undef LIO_BP
}

// C(6x6) = A(6x6) * B(6x6), one lane per output element; the same double accumulation over k as lio_gemm32f.
__device__ static void lio_gemm6_wave(const float* A, const float* B, float* C, int lane)
{
    if (lane < 36) {
        const int i = lane / 6, j = lane % 6;
        double s = 0.0;
        for (int p = 0; p < 6; ++p) s += (double)A[i * 6 + p] * (double)B[p * 6 + j];
        C[lane] = (float)s;
    }
}

// matV.inv(), MO:1807 (OpenCV hal::LU32f on [A | I], partial pivoting; singular -> zero matrix) by a wave, ALL lanes
// calling: lane j (0..5) carries column j of [A | I] through the elimination with A in registers (every lane eliminates
// its own copy of A: the pivot decisions are wave-uniform); the operations on each element are the serial code's
// (oracle/lio_oracle.c), in its order.  The matrix arrives one element per lane (vrow of lane e < 36 =
// element (e/6, e%6)); lanes 0..5 write B (LDS).
__device__ static void lio_inv6_lu_wave(float vrow, float* B, int lane)
{
    const float eps = FLT_EPSILON * 10;
    float A[36], Bc[6];
#pragma unroll
    for (int k = 0; k < 36; ++k) A[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vrow), k));
    const int col = lane % 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) Bc[i] = (i == col) ? 1.0f : 0.0f;
    bool singular = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        if (!singular) {                                     // wave-uniform
            int k = i;
            float best = fabsf(A[i * 6 + i]);
#pragma unroll
            for (int j = i + 1; j < 6; ++j) { const float v = fabsf(A[j * 6 + i]); if (v > best) { best = v; k = j; } }
            if (best < eps) {
                singular = true;
            } else {
#pragma unroll
                for (int j = i + 1; j < 6; ++j)
                    if (k == j) {                            // wave-uniform: rows i and j change places
#pragma unroll
                        for (int c = i; c < 6; ++c) { const float t = A[i * 6 + c]; A[i * 6 + c] = A[j * 6 + c]; A[j * 6 + c] = t; }
                        const float t = Bc[i]; Bc[i] = Bc[j]; Bc[j] = t;
                    }
                const float d = -1 / A[i * 6 + i];
#pragma unroll
                for (int j = i + 1; j < 6; ++j) {
                    const float alpha = A[j * 6 + i] * d;
#pragma unroll
                    for (int c = i + 1; c < 6; ++c) A[j * 6 + c] += alpha * A[i * 6 + c];
                    Bc[j] += alpha * Bc[i];
                }
            }
        }
    }
    if (!singular) {
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            float s = Bc[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) s -= A[i * 6 + k] * Bc[k];
            Bc[i] = s / A[i * 6 + i];
        }
    }
    if (lane < 6) {
#pragma unroll
        for (int i = 0; i < 6; ++i) B[i * 6 + lane] = singular ? 0.0f : Bc[i];
    }
}

// CV_32F matrix product (double accumulation over k, rounded once).
__device__ static void lio_gemm32f(const float* A, const float* B, float* C, int m, int k, int n)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int p = 0; p < k; ++p) s += (double)A[i * k + p] * (double)B[p * n + j];
            C[i * n + j] = (float)s;
        }
}

// fp32 sine/cosine defined as the fp64 function rounded once (see DESIGN.md,
// "trig"): the reference's sin(float)/cos(float) (MO:1714-1719, PCL
// getTransformation) up to the last-bit freedom of the platform libm.
__device__ static float lio_sinf(float x) { return (float)sin((double)x); }
__device__ static float lio_cosf(float x) { return (float)cos((double)x); }

// pcl::getTransformation(x,y,z,roll,pitch,yaw) as used by trans2Affine3f, MO:887-890.
__device__ static void lio_pose_to_transform(const float pose[6], float T[12], float trig[6])
{
    const float A = lio_cosf(pose[2]), B = lio_sinf(pose[2]);   // yaw
    const float C = lio_cosf(pose[1]), D = lio_sinf(pose[1]);   // pitch
    const float E = lio_cosf(pose[0]), F = lio_sinf(pose[0]);   // roll
    const float DE = D * E, DF = D * F;
    T[0] = A * C;  T[1] = A * DF - B * E;  T[2]  = B * F + A * DE;  T[3]  = pose[3];
    T[4] = B * C;  T[5] = A * E + B * DF;  T[6]  = B * DE - A * F;  T[7]  = pose[4];
    T[8] = -D;     T[9] = C * F;           T[10] = C * E;           T[11] = pose[5];
    // LMOptimization's names (MO:1714-1719): srx=sin(yaw) ... crz=cos(roll)
    trig[0] = B; trig[1] = A; trig[2] = D; trig[3] = C; trig[4] = F; trig[5] = E;
}

// The same for a pose held by lane 0 of a wave, ALL lanes calling: the six fp64 sines / cosines (a few hundred
// instructions each) are evaluated by six lanes side by side -- the same function of the same argument, so the same bits --
// and lane 0 assembles T and trig (only lane 0's T / trig are written).
__device__ static void lio_pose_to_transform_wave(const float pose[6], float* T, float* trig, int lane)
{
    const float roll = __shfl(pose[0], 0), pitch = __shfl(pose[1], 0), yaw = __shfl(pose[2], 0);
    const int f = lane % 6;
    const float ang = f < 2 ? yaw : (f < 4 ? pitch : roll);
    const float v = (f & 1) ? lio_sinf(ang) : lio_cosf(ang);          // lane f: 0 cos yaw, 1 sin yaw, 2 cos pitch, 3 sin pitch, 4 cos roll, 5 sin roll
    const float A = __shfl(v, 0), B = __shfl(v, 1), C = __shfl(v, 2), D = __shfl(v, 3), E = __shfl(v, 4), F = __shfl(v, 5);
    if (lane != 0) return;
    const float DE = D * E, DF = D * F;
    T[0] = A * C;  T[1] = A * DF - B * E;  T[2]  = B * F + A * DE;  T[3]  = pose[3];
    T[4] = B * C;  T[5] = A * E + B * DF;  T[6]  = B * DE - A * F;  T[7]  = pose[4];
    T[8] = -D;     T[9] = C * F;           T[10] = C * E;           T[11] = pose[5];
    trig[0] = B; trig[1] = A; trig[2] = D; trig[3] = C; trig[4] = F; trig[5] = E;
}
