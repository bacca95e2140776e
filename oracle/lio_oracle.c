/*
 * lio_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See lio_oracle.h.
 *
 * CPU restatement (plain C, fp32 with the reference's mixed fp64 spots) of
 *   MO:841-847, 887-890, 1613-1907   scan-to-map registration
 *   IP:359-418, 502-615              IMU-rotation deskew
 *   FE:81-101                        range curvature
 * of /root/reference/src/liorf/src/{mapOptmization,imageProjection,featureExtraction}.cpp.
 *
 * PARITY STATUS: parity unpinned (no reference fixtures exist; see header).
 *
 * Build with -ffp-contract=off: the reference is compiled -O3 without -march
 * (CMakeLists.txt:7), i.e. SSE2 and no FMA contraction.
 *
 * Third-party arithmetic restated here (sources are NOT under /root/reference):
 *   PCL 1.10   pcl::getTransformation, KdTreeFLANN::nearestKSearch (exact kNN,
 *              FLANN L2_Simple distance), pcl::rad2deg
 *   Eigen 3.3  ColPivHouseholderQR<Matrix<float,5,3>>::solve, Affine3f inverse/product
 *   OpenCV 4.2 cv::solve(DECOMP_QR), cv::eigen (Jacobi), Mat::inv (LU), gemm
 *   tf         Quaternion::setRPY/slerp, Matrix3x3::getRPY
 * Versions are not pinned by the reference (find_package(... REQUIRED) only);
 * the distro implied by its README (ROS Noetic) ships the versions above.
 * Reductions that those libraries may vectorise (Eigen squaredNorm) are
 * restated in plain left-to-right order.
 */
#include "lio_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ config */

void lo_s2m_default_config(lo_s2m_config *c)
{
    c->k = 5;
    c->max_sq_dist = 1.0f;
    c->plane_tol = 0.2;
    c->weight = 0.9;
    c->min_s = 0.1;
    c->min_corr = 50;
    c->max_iters = 30;
    c->eig_thresh = 100.0f;
    c->conv_deg = 0.05;
    c->conv_cm = 0.05;
    c->min_scan_pts = 30;
    c->jacobian_mode = 0;
    c->force_all_iters = 0;
    c->n_threads = 1;
    c->knn_mode = 1;
    c->trig_mode = 0;
}

/* ------------------------------------------------------------------- trig */
/*
 * The reference writes sin(x)/cos(x) on float arguments under `using namespace
 * std` (MO:1714-1719; PCL getTransformation), i.e. the float overloads.  libm
 * sinf is not correctly rounded on every platform, so its last bit is platform
 * dependent.  trig_mode 0 defines the value as the fp64 function rounded to
 * fp32 (correctly rounded fp32 sine in all but ~2^-28 of cases); trig_mode 1
 * calls libm's float functions.
 */
static float lo_sinf(float x, int mode) { return mode ? sinf(x) : (float)sin((double)x); }
static float lo_cosf(float x, int mode) { return mode ? cosf(x) : (float)cos((double)x); }

/* -------------------------------------------------------- rigid transforms */

/* pcl::getTransformation(x,y,z,roll,pitch,yaw) -- PCL common/impl/eigen.hpp:
 *   A=cos(yaw) B=sin(yaw) C=cos(pitch) D=sin(pitch) E=cos(roll) F=sin(roll)
 *   DE=D*E DF=D*F
 *   [ A*C   A*DF-B*E   B*F+A*DE   x ]
 *   [ B*C   A*E+B*DF   B*DE-A*F   y ]
 *   [ -D    C*F        C*E        z ]
 * used at MO:889 (pose packing [roll,pitch,yaw,x,y,z]) and IP:560,565. */
void lo_get_transformation(float x, float y, float z, float roll, float pitch, float yaw,
                           float T[12], int trig_mode)
{
    float A = lo_cosf(yaw, trig_mode), B = lo_sinf(yaw, trig_mode);
    float C = lo_cosf(pitch, trig_mode), D = lo_sinf(pitch, trig_mode);
    float E = lo_cosf(roll, trig_mode), F = lo_sinf(roll, trig_mode);
    float DE = D * E, DF = D * F;
    T[0] = A * C;  T[1] = A * DF - B * E;  T[2]  = B * F + A * DE;  T[3]  = x;
    T[4] = B * C;  T[5] = A * E + B * DF;  T[6]  = B * DE - A * F;  T[7]  = y;
    T[8] = -D;     T[9] = C * F;           T[10] = C * E;           T[11] = z;
}

/* pointAssociateToMap, MO:841-847 */
void lo_point_associate(const float T[12], const float pi[3], float po[3])
{
    po[0] = T[0] * pi[0] + T[1] * pi[1] + T[2]  * pi[2] + T[3];
    po[1] = T[4] * pi[0] + T[5] * pi[1] + T[6]  * pi[2] + T[7];
    po[2] = T[8] * pi[0] + T[9] * pi[1] + T[10] * pi[2] + T[11];
}

/* ------------------------------------------------------------------- k-NN */
/*
 * pcl::KdTreeFLANN::nearestKSearch(p, 5, ...) (MO:1631) is an exact search;
 * results ascending by squared distance.  FLANN L2_Simple accumulates
 * diff*diff over x,y,z in order starting from 0.  Ties between equal
 * distances are resolved by FLANN's traversal order, which is unspecified;
 * this restatement resolves them by the smaller map index.
 */
static inline float lo_sqdist(const float *a, const float *b)
{
    float r = 0.0f, d;
    d = a[0] - b[0]; r += d * d;
    d = a[1] - b[1]; r += d * d;
    d = a[2] - b[2]; r += d * d;
    return r;
}

typedef struct { int32_t idx[5]; float d2[5]; int n; } lo_top5;

static inline void lo_top5_init(lo_top5 *t) { t->n = 0; }

static inline int lo_better(float d, int32_t i, float dk, int32_t ik)
{
    return d < dk || (d == dk && i < ik);
}

static inline void lo_top5_push(lo_top5 *t, float d, int32_t i)
{
    int pos;
    if (t->n == 5) {
        if (!lo_better(d, i, t->d2[4], t->idx[4])) return;
        pos = 4;
    } else {
        pos = t->n++;
    }
    while (pos > 0 && lo_better(d, i, t->d2[pos - 1], t->idx[pos - 1])) {
        t->d2[pos] = t->d2[pos - 1];
        t->idx[pos] = t->idx[pos - 1];
        --pos;
    }
    t->d2[pos] = d;
    t->idx[pos] = i;
}

int lo_knn5_brute(const float *xyz, size_t n, const float q[3], int32_t idx[5], float d2[5])
{
    lo_top5 t;
    lo_top5_init(&t);
    for (size_t i = 0; i < n; ++i)
        lo_top5_push(&t, lo_sqdist(xyz + 3 * i, q), (int32_t)i);
    for (int k = 0; k < 5; ++k) {
        idx[k] = k < t.n ? t.idx[k] : -1;
        d2[k] = k < t.n ? t.d2[k] : INFINITY;
    }
    return t.n;
}

/* Own kd-tree (the reference uses FLANN's KDTreeSingleIndex through PCL,
 * MO:1846 build + MO:1631 query).  Median split on the widest axis, leaves of
 * <= LO_LEAF points, exact search.  A far subtree is skipped only when the
 * squared distance to its splitting plane is strictly larger than the current
 * 5th-best distance, so ties are always visited (float subtraction, squaring
 * and the x,y,z accumulation are monotone, hence the plane distance is a true
 * lower bound also in fp32). */
#define LO_LEAF 12

typedef struct {
    int32_t left, right;   /* children, or -1 for a leaf */
    int32_t begin, end;    /* leaf: range in perm */
    int32_t axis;
    float   split;
} lo_kdnode;

struct lo_kdtree {
    const float *xyz;
    size_t n;
    int32_t *perm;
    lo_kdnode *nodes;
    int32_t n_nodes, cap_nodes;
};

static int32_t lo_kd_new_node(lo_kdtree *t)
{
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 1024;
        t->nodes = (lo_kdnode *)realloc(t->nodes, sizeof(lo_kdnode) * (size_t)t->cap_nodes);
    }
    return t->n_nodes++;
}

/* nth_element on perm[b..e) by coordinate axis (ties by index for determinism) */
static inline int lo_kd_less(const float *xyz, int axis, int32_t a, int32_t b)
{
    float va = xyz[3 * (size_t)a + axis], vb = xyz[3 * (size_t)b + axis];
    return va < vb || (va == vb && a < b);
}

static void lo_kd_select(const float *xyz, int32_t *perm, int32_t b, int32_t e, int32_t nth, int axis)
{
    while (e - b > 1) {
        int32_t mid = b + (e - b) / 2;
        /* median-of-three pivot */
        int32_t lo = perm[b], mi = perm[mid], hi = perm[e - 1], piv;
        if (lo_kd_less(xyz, axis, lo, mi)) {
            if (lo_kd_less(xyz, axis, mi, hi)) piv = mi;
            else piv = lo_kd_less(xyz, axis, lo, hi) ? hi : lo;
        } else {
            if (lo_kd_less(xyz, axis, lo, hi)) piv = lo;
            else piv = lo_kd_less(xyz, axis, mi, hi) ? hi : mi;
        }
        int32_t i = b, j = e - 1;
        while (i <= j) {
            while (lo_kd_less(xyz, axis, perm[i], piv)) ++i;
            while (lo_kd_less(xyz, axis, piv, perm[j])) --j;
            if (i <= j) { int32_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp; ++i; --j; }
        }
        if (nth <= j) e = j + 1;
        else if (nth >= i) b = i;
        else return;
    }
}

static int32_t lo_kd_build_rec(lo_kdtree *t, int32_t b, int32_t e)
{
    int32_t id = lo_kd_new_node(t);
    if (e - b <= LO_LEAF) {
        lo_kdnode *nd = &t->nodes[id];
        nd->left = nd->right = -1; nd->begin = b; nd->end = e; nd->axis = 0; nd->split = 0.0f;
        return id;
    }
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int32_t i = b; i < e; ++i)
        for (int a = 0; a < 3; ++a) {
            float v = t->xyz[3 * (size_t)t->perm[i] + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    int axis = 0;
    if (mx[1] - mn[1] > mx[axis] - mn[axis]) axis = 1;
    if (mx[2] - mn[2] > mx[axis] - mn[axis]) axis = 2;
    int32_t mid = b + (e - b) / 2;
    lo_kd_select(t->xyz, t->perm, b, e, mid, axis);
    float split = t->xyz[3 * (size_t)t->perm[mid] + axis];
    int32_t l = lo_kd_build_rec(t, b, mid);
    int32_t r = lo_kd_build_rec(t, mid, e);
    lo_kdnode *nd = &t->nodes[id];
    nd->left = l; nd->right = r; nd->begin = b; nd->end = e; nd->axis = axis; nd->split = split;
    return id;
}

lo_kdtree *lo_kdtree_build(const float *xyz, size_t n)
{
    lo_kdtree *t = (lo_kdtree *)calloc(1, sizeof(lo_kdtree));
    t->xyz = xyz; t->n = n;
    t->perm = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) t->perm[i] = (int32_t)i;
    if (n) lo_kd_build_rec(t, 0, (int32_t)n);
    return t;
}

void lo_kdtree_free(lo_kdtree *t)
{
    if (!t) return;
    free(t->perm); free(t->nodes); free(t);
}

static void lo_kd_search(const lo_kdtree *t, int32_t id, const float q[3], lo_top5 *top)
{
    const lo_kdnode *nd = &t->nodes[id];
    if (nd->left < 0) {
        for (int32_t i = nd->begin; i < nd->end; ++i) {
            int32_t p = t->perm[i];
            lo_top5_push(top, lo_sqdist(t->xyz + 3 * (size_t)p, q), p);
        }
        return;
    }
    /* points with coord < split (or equal, smaller index) are left; the right
     * child holds coord >= split.  Left child points satisfy coord <= split. */
    float diff = q[nd->axis] - nd->split;
    int32_t nearc = diff < 0.0f ? nd->left : nd->right;
    int32_t farc  = diff < 0.0f ? nd->right : nd->left;
    lo_kd_search(t, nearc, q, top);
    if (top->n < 5 || !(diff * diff > top->d2[4]))
        lo_kd_search(t, farc, q, top);
}

int lo_kdtree_knn5(const lo_kdtree *t, const float q[3], int32_t idx[5], float d2[5])
{
    lo_top5 top;
    lo_top5_init(&top);
    if (t->n) lo_kd_search(t, 0, q, &top);
    for (int k = 0; k < 5; ++k) {
        idx[k] = k < top.n ? top.idx[k] : -1;
        d2[k] = k < top.n ? top.d2[k] : INFINITY;
    }
    return top.n;
}

/* ------------------------------------- Eigen ColPivHouseholderQR 5x3 solve */
/*
 * matX0 = matA0.colPivHouseholderQr().solve(matB0)  (MO:1648), Eigen 3.3:
 *   ColPivHouseholderQR::computeInPlace  -- pivot on largest updated column
 *     norm, LAPACK-style norm downdating (lawn176), makeHouseholderInPlace,
 *     applyHouseholderOnTheLeft;
 *   _solve_impl -- c = Q^T b over nonzeroPivots() reflectors, back-substitute
 *     (column-oriented triangular_solve_vector), un-permute, zero the rest.
 */
static float lo_norm_tail(const float a[5][3], int col, int from)
{
    float s = 0.0f;
    for (int i = from; i < 5; ++i) s += a[i][col] * a[i][col];
    return sqrtf(s);
}

static void lo_colpiv_qr_impl_5x3(const float A[15], const float b[5], float x[3], int32_t *perm_out, float *rdiag_out, int32_t *nzp_out)
{
    enum { R = 5, C = 3 };
    float qr[5][3], hc[3], direct[3], updated[3], c[5];
    int trans[3], perm[3];
    for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) qr[i][j] = A[i * 3 + j];

    for (int k = 0; k < C; ++k) { direct[k] = lo_norm_tail(qr, k, 0); updated[k] = direct[k]; }
    float maxn = updated[0];
    if (updated[1] > maxn) maxn = updated[1];
    if (updated[2] > maxn) maxn = updated[2];
    float th = maxn * FLT_EPSILON;
    const float threshold_helper = (th * th) / (float)R;
    const float norm_downdate_threshold = sqrtf(FLT_EPSILON);
    int nonzero_pivots = C;

    for (int k = 0; k < C; ++k) {
        /* biggest remaining column (first maximum wins) */
        int big = k;
        float bigv = updated[k];
        for (int j = k + 1; j < C; ++j) if (updated[j] > bigv) { bigv = updated[j]; big = j; }
        float big_sq = bigv * bigv;
        if (nonzero_pivots == C && big_sq < threshold_helper * (float)(R - k))
            nonzero_pivots = k;
        trans[k] = big;
        if (k != big) {
            for (int i = 0; i < R; ++i) { float t = qr[i][k]; qr[i][k] = qr[i][big]; qr[i][big] = t; }
            float t;
            t = updated[k]; updated[k] = updated[big]; updated[big] = t;
            t = direct[k];  direct[k]  = direct[big];  direct[big]  = t;
        }
        /* makeHouseholderInPlace on qr[k..R-1][k] */
        float tail_sq = 0.0f;
        for (int i = k + 1; i < R; ++i) tail_sq += qr[i][k] * qr[i][k];
        float c0 = qr[k][k], beta, tau;
        if (tail_sq <= FLT_MIN) {
            tau = 0.0f; beta = c0;
            for (int i = k + 1; i < R; ++i) qr[i][k] = 0.0f;
        } else {
            beta = sqrtf(c0 * c0 + tail_sq);
            if (c0 >= 0.0f) beta = -beta;
            float den = c0 - beta;
            for (int i = k + 1; i < R; ++i) qr[i][k] = qr[i][k] / den;
            tau = (beta - c0) / beta;
        }
        hc[k] = tau;
        qr[k][k] = beta;
        /* apply H_k to the trailing columns */
        if (tau != 0.0f) {
            for (int j = k + 1; j < C; ++j) {
                float tmp = 0.0f;
                for (int i = k + 1; i < R; ++i) tmp += qr[i][k] * qr[i][j];
                tmp += qr[k][j];
                qr[k][j] -= tau * tmp;
                for (int i = k + 1; i < R; ++i) qr[i][j] -= (tau * qr[i][k]) * tmp;
            }
        }
        /* norm downdate */
        for (int j = k + 1; j < C; ++j) {
            if (updated[j] != 0.0f) {
                float temp = fabsf(qr[k][j]) / updated[j];
                temp = (1.0f + temp) * (1.0f - temp);
                temp = temp < 0.0f ? 0.0f : temp;
                float ratio = updated[j] / direct[j];
                float temp2 = temp * (ratio * ratio);
                if (temp2 <= norm_downdate_threshold) {
                    direct[j] = lo_norm_tail(qr, j, k + 1);
                    updated[j] = direct[j];
                } else {
                    updated[j] *= sqrtf(temp);
                }
            }
        }
    }
    /* column permutation from the transpositions */
    for (int k = 0; k < C; ++k) perm[k] = k;
    for (int k = 0; k < C; ++k) { int t = perm[k]; perm[k] = perm[trans[k]]; perm[trans[k]] = t; }

    if (perm_out) for (int k = 0; k < C; ++k) { perm_out[k] = perm[k]; rdiag_out[k] = qr[k][k]; }
    if (nzp_out) *nzp_out = nonzero_pivots;

    /* solve */
    x[0] = x[1] = x[2] = 0.0f;
    if (nonzero_pivots == 0) return;
    for (int i = 0; i < R; ++i) c[i] = b[i];
    for (int k = 0; k < nonzero_pivots; ++k) {
        float tau = hc[k];
        if (k == R - 1) { c[k] *= 1.0f - tau; continue; }
        if (tau != 0.0f) {
            float tmp = 0.0f;
            for (int i = k + 1; i < R; ++i) tmp += qr[i][k] * c[i];
            tmp += c[k];
            c[k] -= tau * tmp;
            for (int i = k + 1; i < R; ++i) c[i] -= (tau * qr[i][k]) * tmp;
        }
    }
    for (int i = nonzero_pivots - 1; i >= 0; --i) {
        if (c[i] != 0.0f) {
            c[i] /= qr[i][i];
            for (int r = 0; r < i; ++r) c[r] -= c[i] * qr[r][i];
        }
    }
    for (int i = 0; i < nonzero_pivots; ++i) x[perm[i]] = c[i];
}

void lo_colpiv_qr_solve_5x3(const float A[15], const float b[5], float x[3])
{
    lo_colpiv_qr_impl_5x3(A, b, x, NULL, NULL, NULL);
}

/* Test hook (tests/test_oracle_math.py): the column order the factorisation chose (perm[k] = original column that ended
 * up in position k), the diagonal of R and Eigen's nonzeroPivots() -- compared against LAPACK sgeqp3 through scipy, an
 * implementation that shares no code and no author with this file. */
void lo_colpiv_qr_debug_5x3(const float A[15], int32_t perm[3], float rdiag[3], int32_t *nonzero_pivots)
{
    const float b[5] = { -1.0f, -1.0f, -1.0f, -1.0f, -1.0f };
    float x[3];
    lo_colpiv_qr_impl_5x3(A, b, x, perm, rdiag, nonzero_pivots);
}

/* ------------------------------------------------------- OpenCV restatements */

/* CV_32F gemm: OpenCV's generic float kernel accumulates each output element
 * in double over k ascending, then rounds to float. */
void lo_gemm32f(const float *A, const float *B, float *Cm, int m, int k, int n)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int p = 0; p < k; ++p) s += (double)A[i * k + p] * (double)B[p * n + j];
            Cm[i * n + j] = (float)s;
        }
}

/* cv::solve(DECOMP_QR) on a square CV_32F system -> hal::QR32f (Householder,
 * unit-norm reflector v, factor 2 v v^T), then back substitution; eps =
 * FLT_EPSILON*10?  OpenCV passes FLT_EPSILON * 10 to QRImpl for 32F. */
int lo_solve6_qr(const float A_in[36], const float b_in[6], float x[6])
{
    enum { N = 6 };
    float A[N][N], b[N], vl[N], hf[N];
    const float eps = FLT_EPSILON * 10;
    for (int i = 0; i < N; ++i) { b[i] = b_in[i]; for (int j = 0; j < N; ++j) A[i][j] = A_in[i * N + j]; }

    for (int l = 0; l < N; ++l) {
        int vs = N - l;
        float vnorm = 0.0f;
        for (int i = 0; i < vs; ++i) { vl[i] = A[l + i][l]; vnorm += vl[i] * vl[i]; }
        float tmpv = vl[0];
        float sg = vl[0] >= 0.0f ? 1.0f : -1.0f;
        vl[0] = vl[0] + sg * sqrtf(vnorm);
        vnorm = sqrtf(vnorm + vl[0] * vl[0] - tmpv * tmpv);
        for (int i = 0; i < vs; ++i) vl[i] /= vnorm;
        for (int j = l; j < N; ++j) {
            float va = 0.0f;
            for (int i = l; i < N; ++i) va += vl[i - l] * A[i][j];
            for (int i = l; i < N; ++i) A[i][j] -= 2 * vl[i - l] * va;
        }
        hf[l] = vl[0] * vl[0];
        for (int i = 1; i < vs; ++i) A[l + i][l] = vl[i] / vl[0];
    }
    for (int l = 0; l < N; ++l) {
        vl[0] = 1.0f;
        for (int j = 1; j < N - l; ++j) vl[j] = A[j + l][l];
        float vb = 0.0f;
        for (int i = l; i < N; ++i) vb += vl[i - l] * b[i];
        for (int i = l; i < N; ++i) b[i] -= 2 * vl[i - l] * vb * hf[l];
    }
    for (int i = N - 1; i >= 0; --i) {
        for (int j = N - 1; j > i; --j) b[i] -= b[j] * A[i][j];
        if (fabsf(A[i][i]) < eps) { for (int p = 0; p < N; ++p) x[p] = 0.0f; return 0; }
        b[i] /= A[i][i];
    }
    for (int i = 0; i < N; ++i) x[i] = b[i];
    return 1;
}

static float lo_cv_hypot(float a, float b)
{
    a = fabsf(a); b = fabsf(b);
    if (a > b) { b /= a; return a * sqrtf(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrtf(1 + a * a); }
    return 0.0f;
}

/* cv::eigen on a symmetric CV_32F matrix -> JacobiImpl_: pivot on the largest
 * off-diagonal element (tracked per row/column), rotate, until |p| <= eps;
 * then selection-sort eigenvalues descending with eigenvectors as rows. */
void lo_eigen6_sym(const float A_in[36], float W[6], float V[36])
{
    enum { N = 6 };
    float A[N][N];
    int indR[N], indC[N];
    const float eps = FLT_EPSILON;
    int i, j, k, m;
    float mv;
    for (i = 0; i < N; ++i) for (j = 0; j < N; ++j) { A[i][j] = A_in[i * N + j]; V[i * N + j] = (i == j) ? 1.0f : 0.0f; }

    for (k = 0; k < N; ++k) {
        W[k] = A[k][k];
        if (k < N - 1) {
            for (m = k + 1, mv = fabsf(A[k][m]), i = k + 2; i < N; ++i) {
                float val = fabsf(A[k][i]);
                if (mv < val) { mv = val; m = i; }
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabsf(A[0][k]), i = 1; i < k; ++i) {
                float val = fabsf(A[i][k]);
                if (mv < val) { mv = val; m = i; }
            }
            indC[k] = m;
        }
    }

    int maxIters = N * N * 30;
    for (int iters = 0; iters < maxIters; ++iters) {
        for (k = 0, mv = fabsf(A[0][indR[0]]), i = 1; i < N - 1; ++i) {
            float val = fabsf(A[i][indR[i]]);
            if (mv < val) { mv = val; k = i; }
        }
        int l = indR[k];
        for (i = 1; i < N; ++i) {
            float val = fabsf(A[indC[i]][i]);
            if (mv < val) { mv = val; k = indC[i]; l = i; }
        }
        float p = A[k][l];
        if (fabsf(p) <= eps) break;
        float y = (float)((W[l] - W[k]) * 0.5);
        float t = fabsf(y) + lo_cv_hypot(p, y);
        float s = lo_cv_hypot(p, t);
        float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) { s = -s; t = -t; }
        A[k][l] = 0;
        W[k] -= t;
        W[l] += t;
        float a0, b0;
#define LO_ROT(v0, v1) do { a0 = (v0); b0 = (v1); (v0) = a0 * c - b0 * s; (v1) = a0 * s + b0 * c; } while (0)
        for (i = 0; i < k; ++i)     LO_ROT(A[i][k], A[i][l]);
        for (i = k + 1; i < l; ++i) LO_ROT(A[k][i], A[i][l]);
        for (i = l + 1; i < N; ++i) LO_ROT(A[k][i], A[l][i]);
        for (i = 0; i < N; ++i)     LO_ROT(V[k * N + i], V[l * N + i]);
#undef LO_ROT
        for (j = 0; j < 2; ++j) {
            int idx = j == 0 ? k : l;
            if (idx < N - 1) {
                for (m = idx + 1, mv = fabsf(A[idx][m]), i = idx + 2; i < N; ++i) {
                    float val = fabsf(A[idx][i]);
                    if (mv < val) { mv = val; m = i; }
                }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = fabsf(A[0][idx]), i = 1; i < idx; ++i) {
                    float val = fabsf(A[i][idx]);
                    if (mv < val) { mv = val; m = i; }
                }
                indC[idx] = m;
            }
        }
    }
    for (k = 0; k < N - 1; ++k) {
        m = k;
        for (i = k + 1; i < N; ++i) if (W[m] < W[i]) m = i;
        if (k != m) {
            float t = W[m]; W[m] = W[k]; W[k] = t;
            for (i = 0; i < N; ++i) { t = V[m * N + i]; V[m * N + i] = V[k * N + i]; V[k * N + i] = t; }
        }
    }
}

/* cv::Mat::inv() default DECOMP_LU on 6x6 CV_32F -> hal::LU32f on [A | I]
 * with partial pivoting, eps = FLT_EPSILON*10; singular -> zero matrix. */
int lo_inv6_lu(const float A_in[36], float Ainv[36])
{
    enum { N = 6 };
    float A[N][N], B[N][N];
    const float eps = FLT_EPSILON * 10;
    int i, j, k;
    for (i = 0; i < N; ++i) for (j = 0; j < N; ++j) { A[i][j] = A_in[i * N + j]; B[i][j] = (i == j) ? 1.0f : 0.0f; }
    for (i = 0; i < N; ++i) {
        k = i;
        for (j = i + 1; j < N; ++j) if (fabsf(A[j][i]) > fabsf(A[k][i])) k = j;
        if (fabsf(A[k][i]) < eps) { memset(Ainv, 0, sizeof(float) * 36); return 0; }
        if (k != i) {
            for (j = i; j < N; ++j) { float t = A[i][j]; A[i][j] = A[k][j]; A[k][j] = t; }
            for (j = 0; j < N; ++j) { float t = B[i][j]; B[i][j] = B[k][j]; B[k][j] = t; }
        }
        float d = -1 / A[i][i];
        for (j = i + 1; j < N; ++j) {
            float alpha = A[j][i] * d;
            for (k = i + 1; k < N; ++k) A[j][k] += alpha * A[i][k];
            for (k = 0; k < N; ++k) B[j][k] += alpha * B[i][k];
        }
    }
    for (i = N - 1; i >= 0; --i)
        for (j = 0; j < N; ++j) {
            float s = B[i][j];
            for (k = i + 1; k < N; ++k) s -= A[i][k] * B[k][j];
            B[i][j] = s / A[i][i];
        }
    for (i = 0; i < N; ++i) for (j = 0; j < N; ++j) Ainv[i * N + j] = B[i][j];
    return 1;
}

/* ------------------------------------------------------- surfOptimization */

static void lo_surf_point(const lo_s2m_config *cfg, const float T[12],
                          const float *pointOri, const float *map_xyz, size_t n_map,
                          const lo_kdtree *tree, uint8_t *flag, float *coeff, int32_t *nn)
{
    float pointSel[3];
    int32_t ind[5];
    float sq[5];
    *flag = 0;
    coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0.0f;
    for (int j = 0; j < 5; ++j) nn[j] = -1;

    lo_point_associate(T, pointOri, pointSel);                         /* MO:1630 */
    if (cfg->knn_mode == 0 || tree == NULL) lo_knn5_brute(map_xyz, n_map, pointSel, ind, sq);
    else                                    lo_kdtree_knn5(tree, pointSel, ind, sq); /* MO:1631 */

    if (!((double)sq[4] < (double)cfg->max_sq_dist)) return;            /* MO:1641 */
    for (int j = 0; j < 5; ++j) nn[j] = ind[j];

    float A0[15], B0[5], X0[3];
    for (int j = 0; j < 5; ++j) {                                       /* MO:1642-1646 */
        A0[j * 3 + 0] = map_xyz[3 * (size_t)ind[j] + 0];
        A0[j * 3 + 1] = map_xyz[3 * (size_t)ind[j] + 1];
        A0[j * 3 + 2] = map_xyz[3 * (size_t)ind[j] + 2];
        B0[j] = -1.0f;                                                  /* MO:1638 */
    }
    lo_colpiv_qr_solve_5x3(A0, B0, X0);                                 /* MO:1648 */

    float pa = X0[0], pb = X0[1], pc = X0[2], pd = 1;                   /* MO:1650-1653 */
    float ps = sqrtf(pa * pa + pb * pb + pc * pc);                      /* MO:1655 */
    pa /= ps; pb /= ps; pc /= ps; pd /= ps;                             /* MO:1656 */

    int planeValid = 1;
    for (int j = 0; j < 5; ++j) {                                       /* MO:1658-1666 */
        float v = fabsf(pa * A0[j * 3 + 0] + pb * A0[j * 3 + 1] + pc * A0[j * 3 + 2] + pd);
        if ((double)v > cfg->plane_tol) { planeValid = 0; break; }
    }
    if (!planeValid) return;

    float pd2 = pa * pointSel[0] + pb * pointSel[1] + pc * pointSel[2] + pd; /* MO:1669 */
    /* MO:1671-1672: `1 - 0.9 * fabs(pd2) / sqrt(sqrt(x*x+y*y+z*z))` -- 0.9 is a
     * double literal, fabs/sqrt resolve to the float overloads, so the product,
     * quotient and difference are evaluated in double and rounded once. */
    float r2 = pointOri[0] * pointOri[0] + pointOri[1] * pointOri[1] + pointOri[2] * pointOri[2];
    float s = (float)(1 - cfg->weight * (double)fabsf(pd2) / (double)sqrtf(sqrtf(r2)));

    coeff[0] = s * pa; coeff[1] = s * pb; coeff[2] = s * pc; coeff[3] = s * pd2; /* MO:1674-1677 */
    if ((double)s > cfg->min_s) *flag = 1;                          /* MO:1679-1683 */
}

void lo_surf_optimization(const lo_s2m_config *cfg, const float pose[6],
                          const float *scan_xyz, size_t n_scan,
                          const float *map_xyz, size_t n_map, const lo_kdtree *tree,
                          uint8_t *flag, float *coeff, int32_t *nn_idx)
{
    float T[12];
    /* updatePointAssociateToMap MO:1613-1616 -> trans2Affine3f MO:887-890 */
    lo_get_transformation(pose[3], pose[4], pose[5], pose[0], pose[1], pose[2], T, cfg->trig_mode);
    long n = (long)n_scan;
    int nt = cfg->n_threads > 0 ? cfg->n_threads : 1;
    (void)nt;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)   /* MO:1622 */
#endif
    for (long i = 0; i < n; ++i)
        lo_surf_point(cfg, T, scan_xyz + 3 * i, map_xyz, n_map, tree,
                      flag + i, coeff + 4 * i, nn_idx + 5 * i);
}

/* ---------------------------------------------------------- LMOptimization */

/* MO:1760-1778.  trig = {srx,crx,sry,cry,srz,crz} with the reference's LOAM
 * leftover names: rx <- pose[2] (yaw), ry <- pose[1] (pitch), rz <- pose[0]
 * (roll) (MO:1714-1719).  Operator order is the reference's. */
void lo_jacobian_row(const float trig[6], const float p[3], const float c[4],
                     int jacobian_mode, float row[6], float *rhs)
{
    const float srx = trig[0], crx = trig[1], sry = trig[2], cry = trig[3], srz = trig[4], crz = trig[5];
    const float px = p[0], py = p[1], pz = p[2];
    const float cx = c[0], cy = c[1], cz = c[2];

    float arx = (-srx * cry * px - (srx * sry * srz + crx * crz) * py + (crx * srz - srx * sry * crz) * pz) * cx
              + (crx * cry * px - (srx * crz - crx * sry * srz) * py + (crx * sry * crz + srx * srz) * pz) * cy;

    /* MO:1764 as written has `srx * sry * srz * py`; the analytic derivative is
     * `srx * cry * srz * py`.  jacobian_mode 0 keeps the reference. */
    float mid = jacobian_mode ? (srx * cry * srz * py) : (srx * sry * srz * py);
    float ary = (-crx * sry * px + crx * cry * srz * py + crx * cry * crz * pz) * cx
              + (-srx * sry * px + mid + srx * cry * crz * pz) * cy
              + (-cry * px - sry * srz * py - sry * crz * pz) * cz;

    float arz = ((crx * sry * crz + srx * srz) * py + (srx * crz - crx * sry * srz) * pz) * cx
              + ((-crx * srz + srx * sry * crz) * py + (-srx * sry * srz - crx * crz) * pz) * cy
              + (cry * crz * py - cry * srz * pz) * cz;

    row[0] = arz; row[1] = ary; row[2] = arx;                          /* MO:1772-1774 */
    row[3] = cx;  row[4] = cy;  row[5] = cz;                           /* MO:1775-1777 */
    *rhs = -c[3];                                                      /* MO:1778 */
}

int lo_lm_optimization(const lo_s2m_config *cfg, int iter_count,
                       const float *ori_xyz, const float *coeff4, int n_corr,
                       float pose[6], float matP[36], int32_t *is_degenerate,
                       float AtA_out[36], float AtB_out[6])
{
    float trig[6];
    trig[0] = lo_sinf(pose[2], cfg->trig_mode); trig[1] = lo_cosf(pose[2], cfg->trig_mode); /* MO:1714-1715 */
    trig[2] = lo_sinf(pose[1], cfg->trig_mode); trig[3] = lo_cosf(pose[1], cfg->trig_mode); /* MO:1716-1717 */
    trig[4] = lo_sinf(pose[0], cfg->trig_mode); trig[5] = lo_cosf(pose[0], cfg->trig_mode); /* MO:1718-1719 */

    if (n_corr < cfg->min_corr) return 0;                              /* MO:1721-1724 */

    /* matA (n x 6), matB (n x 1); matAtA = matAt*matA, matAtB = matAt*matB
     * (MO:1781-1783) with OpenCV's double accumulation over rows in order. */
    double acc[6][6], accb[6];
    memset(acc, 0, sizeof(acc)); memset(accb, 0, sizeof(accb));
    for (int i = 0; i < n_corr; ++i) {
        float row[6], rhs;
        lo_jacobian_row(trig, ori_xyz + 3 * (size_t)i, coeff4 + 4 * (size_t)i, cfg->jacobian_mode, row, &rhs);
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) acc[a][b] += (double)row[a] * (double)row[b];
            accb[a] += (double)row[a] * (double)rhs;
        }
    }
    float AtA[36], AtB[6], X[6];
    for (int a = 0; a < 6; ++a) { for (int b = 0; b < 6; ++b) AtA[a * 6 + b] = (float)acc[a][b]; AtB[a] = (float)accb[a]; }
    if (AtA_out) memcpy(AtA_out, AtA, sizeof(AtA));
    if (AtB_out) memcpy(AtB_out, AtB, sizeof(AtB));

    lo_solve6_qr(AtA, AtB, X);                                          /* MO:1784 */

    if (iter_count == 0) {                                              /* MO:1786-1808 */
        float E[6], V[36], V2[36], Vinv[36];
        lo_eigen6_sym(AtA, E, V);
        memcpy(V2, V, sizeof(V));
        *is_degenerate = 0;
        for (int i = 5; i >= 0; --i) {
            if (E[i] < cfg->eig_thresh) {
                for (int j = 0; j < 6; ++j) V2[i * 6 + j] = 0;
                *is_degenerate = 1;
            } else {
                break;
            }
        }
        lo_inv6_lu(V, Vinv);
        lo_gemm32f(Vinv, V2, matP, 6, 6, 6);                            /* MO:1807 */
    }

    if (*is_degenerate) {                                               /* MO:1810-1815 */
        float X2[6];
        memcpy(X2, X, sizeof(X));
        lo_gemm32f(matP, X2, X, 6, 6, 1);
    }

    for (int k = 0; k < 6; ++k) pose[k] += X[k];                        /* MO:1817-1822 */

    /* MO:1824-1831: pcl::rad2deg(float) = a * 57.29578f; pow(float,int) and the
     * sqrt are evaluated in double, the result is stored in a float. */
    double r0 = (double)(X[0] * 57.29578f), r1 = (double)(X[1] * 57.29578f), r2 = (double)(X[2] * 57.29578f);
    float deltaR = (float)sqrt(r0 * r0 + r1 * r1 + r2 * r2);
    double t0 = (double)(X[3] * 100), t1 = (double)(X[4] * 100), t2 = (double)(X[5] * 100);
    float deltaT = (float)sqrt(t0 * t0 + t1 * t1 + t2 * t2);

    if ((double)deltaR < cfg->conv_deg && (double)deltaT < cfg->conv_cm) return 1; /* MO:1833-1835 */
    return 0;
}

/* ----------------------------------------------------- scan2MapOptimization */

/* wall time of the last kd-tree build in lo_scan2map (the reference rebuilds its tree for every scan, MO:1846):
 * lets the CPU baseline report the build / query split */
static double lo_last_build_s = 0.0;
double lo_last_kdtree_build_seconds(void) { return lo_last_build_s; }

int lo_scan2map(const lo_s2m_config *cfg,
                const float *scan_xyz, size_t n_scan,
                const float *map_xyz, size_t n_map,
                float pose[6], float matP_io[36], int32_t *is_degenerate_io,
                lo_s2m_result *res,
                int corr_iter, uint8_t *corr_flag, float *corr_coeff, int32_t *corr_nn)
{
    memset(res, 0, sizeof(*res));
    res->is_degenerate = *is_degenerate_io;
    memcpy(res->matP, matP_io, sizeof(float) * 36);
    if (cfg->max_iters < 1 || cfg->max_iters > 32) {                    /* the per-iteration trace holds 32 entries: refuse, never clamp */
        res->status = -1;
        return res->status;
    }
    if (!((long)n_scan > (long)cfg->min_scan_pts)) {                    /* MO:1844 */
        res->status = LO_TOO_FEW_POINTS;
        return res->status;
    }
    lo_kdtree *tree = NULL;
    const double t_build0 = omp_get_wtime();
    if (cfg->knn_mode == 1) tree = lo_kdtree_build(map_xyz, n_map);     /* MO:1846 */
    lo_last_build_s = omp_get_wtime() - t_build0;

    uint8_t *flag = (uint8_t *)malloc(n_scan);
    float *coeff = (float *)malloc(sizeof(float) * 4 * n_scan);
    int32_t *nn = (int32_t *)malloc(sizeof(int32_t) * 5 * n_scan);
    float *ori_sel = (float *)malloc(sizeof(float) * 3 * n_scan);
    float *coeff_sel = (float *)malloc(sizeof(float) * 4 * n_scan);

    const int max_iters = cfg->max_iters;
    for (int it = 0; it < max_iters; ++it) {                            /* MO:1848 */
        lo_surf_optimization(cfg, pose, scan_xyz, n_scan, map_xyz, n_map, tree, flag, coeff, nn);
        if (it == corr_iter) {
            if (corr_flag)  memcpy(corr_flag, flag, n_scan);
            if (corr_coeff) memcpy(corr_coeff, coeff, sizeof(float) * 4 * n_scan);
            if (corr_nn)    memcpy(corr_nn, nn, sizeof(int32_t) * 5 * n_scan);
        }
        /* combineOptimizationCoeffs MO:1689-1700: ordered compaction */
        int nc = 0;
        for (size_t i = 0; i < n_scan; ++i)
            if (flag[i]) {
                memcpy(ori_sel + 3 * (size_t)nc, scan_xyz + 3 * i, sizeof(float) * 3);
                memcpy(coeff_sel + 4 * (size_t)nc, coeff + 4 * i, sizeof(float) * 4);
                ++nc;
            }
        res->n_corr_iter[it] = nc;
        res->n_corr_last = nc;
        int conv = lo_lm_optimization(cfg, it, ori_sel, coeff_sel, nc, pose, matP_io, is_degenerate_io,
                                      nc >= cfg->min_corr ? res->AtA : NULL,
                                      nc >= cfg->min_corr ? res->AtB : NULL);
        memcpy(res->pose_iter[it], pose, sizeof(float) * 6);
        res->iters = it + 1;
        if (conv) {
            res->converged = 1;
            if (!cfg->force_all_iters) break;                           /* MO:1857-1858 */
        }
    }
    res->is_degenerate = *is_degenerate_io;
    memcpy(res->matP, matP_io, sizeof(float) * 36);
    res->status = (res->n_corr_last < cfg->min_corr) ? LO_TOO_FEW_CORR : LO_OK;

    free(flag); free(coeff); free(nn); free(ori_sel); free(coeff_sel);
    lo_kdtree_free(tree);
    return res->status;
}

/* ---------------------------------------------------------- transformUpdate */
/* tf (bullet LinearMath) restated in fp64: Quaternion::setRPY, slerp via
 * angleShortestPath, Matrix3x3::setRotation + getRPY (solution 1). */
typedef struct { double x, y, z, w; } lo_quat;

static lo_quat lo_q_set_rpy(double roll, double pitch, double yaw)
{
    double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
    lo_quat q;
    q.x = sr * cp * cy - cr * sp * sy;
    q.y = cr * sp * cy + sr * cp * sy;
    q.z = cr * cp * sy - sr * sp * cy;
    q.w = cr * cp * cy + sr * sp * sy;
    return q;
}
static double lo_q_dot(lo_quat a, lo_quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

static lo_quat lo_q_slerp(lo_quat a, lo_quat q, double t)
{
    double s = sqrt(lo_q_dot(a, a) * lo_q_dot(q, q));
    double dt = lo_q_dot(a, q);
    double ang = (dt < 0) ? acos(-dt / s) * 2.0 : acos(dt / s) * 2.0;
    double theta = ang / 2.0;
    if (theta != 0.0) {
        double d = 1.0 / sin(theta);
        double s0 = sin((1.0 - t) * theta);
        double s1 = sin(t * theta);
        lo_quat r;
        if (dt < 0) {
            r.x = (a.x * s0 + -q.x * s1) * d; r.y = (a.y * s0 + -q.y * s1) * d;
            r.z = (a.z * s0 + -q.z * s1) * d; r.w = (a.w * s0 + -q.w * s1) * d;
        } else {
            r.x = (a.x * s0 + q.x * s1) * d; r.y = (a.y * s0 + q.y * s1) * d;
            r.z = (a.z * s0 + q.z * s1) * d; r.w = (a.w * s0 + q.w * s1) * d;
        }
        return r;
    }
    return a;
}

static void lo_q_get_rpy(lo_quat q, double *roll, double *pitch, double *yaw)
{
    double d = lo_q_dot(q, q), s = 2.0 / d;
    double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    double m00 = 1.0 - (yy + zz), m10 = xy + wz;
    double m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (fabs(m20) >= 1) {
        *yaw = 0;
        double delta = atan2(m21, m22);
        if (m20 < 0) { *pitch = M_PI / 2.0; *roll = delta; }
        else         { *pitch = -M_PI / 2.0; *roll = delta; }
    } else {
        *pitch = -asin(m20);
        double cp = cos(*pitch);
        *roll = atan2(m21 / cp, m22 / cp);
        *yaw = atan2(m10 / cp, m00 / cp);
    }
}

static float lo_constraint(float value, float limit)                    /* MO:1899-1907 */
{
    if (value < -limit) value = -limit;
    if (value > limit) value = limit;
    return value;
}

void lo_transform_update(float pose[6], int imu_available, int imu_type,
                         float imu_roll_init, float imu_pitch_init, float imu_rpy_weight,
                         float rotation_tollerance, float z_tollerance)
{
    if (imu_available && imu_type) {                                    /* MO:1869 */
        if (fabsf(imu_pitch_init) < 1.4) {                              /* MO:1871 */
            double w = imu_rpy_weight, r, p, y;
            lo_quat tq = lo_q_set_rpy(pose[0], 0, 0);                   /* MO:1879-1882 */
            lo_quat iq = lo_q_set_rpy(imu_roll_init, 0, 0);
            lo_q_get_rpy(lo_q_slerp(tq, iq, w), &r, &p, &y);
            pose[0] = (float)r;
            tq = lo_q_set_rpy(0, pose[1], 0);                           /* MO:1885-1888 */
            iq = lo_q_set_rpy(0, imu_pitch_init, 0);
            lo_q_get_rpy(lo_q_slerp(tq, iq, w), &r, &p, &y);
            pose[1] = (float)p;
        }
    }
    pose[0] = lo_constraint(pose[0], rotation_tollerance);              /* MO:1892-1894 */
    pose[1] = lo_constraint(pose[1], rotation_tollerance);
    pose[5] = lo_constraint(pose[5], z_tollerance);
}

/* ------------------------------------------------------------------ deskew */

int lo_imu_deskew_info(const double *stamp, const double *gx, const double *gy, const double *gz,
                       int n_imu, double time_scan_cur, double time_scan_end,
                       double *imuTime, double *imuRotX, double *imuRotY, double *imuRotZ)
{
    /* IP:363-369: drop samples older than timeScanCur - 0.01 */
    int first = 0;
    while (first < n_imu && stamp[first] < time_scan_cur - 0.01) ++first;
    if (first >= n_imu) return 0;                                       /* IP:371-372 */
    int cur = 0;
    for (int i = first; i < n_imu; ++i) {                               /* IP:376 */
        double t = stamp[i];
        if (t > time_scan_end + 0.01) break;                            /* IP:387-388 */
        if (cur == 0) {                                                 /* IP:390-397 */
            imuRotX[0] = 0; imuRotY[0] = 0; imuRotZ[0] = 0; imuTime[0] = t;
            ++cur;
            continue;
        }
        if (cur >= 2000) break;                     /* queueLength IP:62 (arrays are 2000 long) */
        double dt = t - imuTime[cur - 1];                               /* IP:404 */
        imuRotX[cur] = imuRotX[cur - 1] + gx[i] * dt;                   /* IP:405-407 */
        imuRotY[cur] = imuRotY[cur - 1] + gy[i] * dt;
        imuRotZ[cur] = imuRotZ[cur - 1] + gz[i] * dt;
        imuTime[cur] = t;
        ++cur;
    }
    --cur;                                                              /* IP:412 */
    return cur;                                                         /* >0 => imuAvailable */
}

void lo_find_rotation(double point_time, const double *imuTime, const double *imuRotX,
                      const double *imuRotY, const double *imuRotZ, int imuPointerCur,
                      float *rx, float *ry, float *rz)
{
    int front = 0;
    while (front < imuPointerCur) {                                     /* IP:507-512 */
        if (point_time < imuTime[front]) break;
        ++front;
    }
    if (point_time > imuTime[front] || front == 0) {                    /* IP:514-518 */
        *rx = (float)imuRotX[front]; *ry = (float)imuRotY[front]; *rz = (float)imuRotZ[front];
    } else {                                                            /* IP:519-526 */
        int back = front - 1;
        double ratioFront = (point_time - imuTime[back]) / (imuTime[front] - imuTime[back]);
        double ratioBack = (imuTime[front] - point_time) / (imuTime[front] - imuTime[back]);
        *rx = (float)(imuRotX[front] * ratioFront + imuRotX[back] * ratioBack);
        *ry = (float)(imuRotY[front] * ratioFront + imuRotY[back] * ratioBack);
        *rz = (float)(imuRotZ[front] * ratioFront + imuRotZ[back] * ratioBack);
    }
}

/* Eigen 3.3 general 3x3 inverse (Transform<float,3,Affine>::inverse on the
 * linear part): cofactors, det from column 0, multiply by 1/det. */
static float lo_cof(const float m[9], int i, int j)
{
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void lo_inv3(const float m[9], float r[9])
{
    float c0 = lo_cof(m, 0, 0), c1 = lo_cof(m, 1, 0), c2 = lo_cof(m, 2, 0);
    float det = c0 * m[0] + c1 * m[3] + c2 * m[6];
    float invdet = 1.0f / det;
    r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
    r[3] = lo_cof(m, 0, 1) * invdet; r[4] = lo_cof(m, 1, 1) * invdet; r[5] = lo_cof(m, 2, 1) * invdet;
    r[6] = lo_cof(m, 0, 2) * invdet; r[7] = lo_cof(m, 1, 2) * invdet; r[8] = lo_cof(m, 2, 2) * invdet;
}

size_t lo_project_point_cloud(const lo_deskew_config *cfg,
                              const float *x, const float *y, const float *z,
                              const float *intensity, const uint16_t *ring, const float *time,
                              size_t n, double time_scan_cur,
                              const double *imuTime, const double *imuRotX,
                              const double *imuRotY, const double *imuRotZ, int imuPointerCur,
                              float *out_xyzi, int32_t *keep_idx)
{
    size_t n_out = 0;
    int first_point = 1;                                                /* firstPointFlag */
    float startInv[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (size_t i = 0; i < n; ++i) {                                    /* IP:581 */
        float px = x[i], py = y[i], pz = z[i], pi = intensity[i];
        float range = sqrtf(px * px + py * py + pz * pz);               /* common_lib.cpp:27-31, IP:589 */
        if ((py < cfg->lidarMinFront && -cfg->lidarMinBack < py &&
             px < cfg->lidarMinLeft && -cfg->lidarMinRight < px) ||
            range > cfg->lidarMaxRange || pi > cfg->lidarMaxIntensity)  /* IP:596-599 */
            continue;
        int rowIdn = ring[i];                                           /* IP:601 */
        if (rowIdn < 0 || rowIdn >= cfg->N_SCAN) continue;              /* IP:602-603 */
        if (rowIdn % cfg->downsampleRate != 0) continue;                /* IP:605-606 */
        if (i % (size_t)cfg->point_filter_num != 0) continue;           /* IP:608-609 */

        float ox = px, oy = py, oz = pz;
        if (!(cfg->deskew_flag == -1 || !cfg->imu_available)) {         /* IP:547-548 */
            double pointTime = time_scan_cur + (double)time[i];         /* IP:550 */
            float rx, ry, rz;
            lo_find_rotation(pointTime, imuTime, imuRotX, imuRotY, imuRotZ, imuPointerCur, &rx, &ry, &rz);
            float T[12];
            lo_get_transformation(0, 0, 0, rx, ry, rz, T, cfg->trig_mode); /* IP:565; findPosition == 0 */
            float L[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] };
            if (first_point) {                                          /* IP:558-562 */
                lo_inv3(L, startInv);
                first_point = 0;
            }
            /* transBt = transStartInverse * transFinal (IP:566): linear part
             * sum_k a(i,k)*b(k,j) in k order; translation is +0. */
            float Bt[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    Bt[r * 3 + c] = startInv[r * 3 + 0] * L[0 * 3 + c] + startInv[r * 3 + 1] * L[1 * 3 + c]
                                  + startInv[r * 3 + 2] * L[2 * 3 + c];
            ox = Bt[0] * px + Bt[1] * py + Bt[2] * pz + 0.0f;           /* IP:569-571 */
            oy = Bt[3] * px + Bt[4] * py + Bt[5] * pz + 0.0f;
            oz = Bt[6] * px + Bt[7] * py + Bt[8] * pz + 0.0f;
        }
        out_xyzi[4 * n_out + 0] = ox; out_xyzi[4 * n_out + 1] = oy;
        out_xyzi[4 * n_out + 2] = oz; out_xyzi[4 * n_out + 3] = pi;     /* IP:572, 613 */
        if (keep_idx) keep_idx[n_out] = (int32_t)i;
        ++n_out;
    }
    return n_out;
}

/* ---------------------------------------------------------------- curvature */

void lo_calculate_smoothness(const float *r, size_t n, float *curvature,
                             int32_t *neighbor_picked, int32_t *label)
{
    if (n < 11) return;
    for (size_t i = 5; i < n - 5; ++i) {                                /* FE:84 */
        float d = r[i - 5] + r[i - 4] + r[i - 3] + r[i - 2] + r[i - 1] - r[i] * 10
                + r[i + 1] + r[i + 2] + r[i + 3] + r[i + 4] + r[i + 5]; /* FE:86-91 */
        curvature[i] = d * d;                                           /* FE:93 */
        if (neighbor_picked) neighbor_picked[i] = 0;                    /* FE:95 */
        if (label) label[i] = 0;                                        /* FE:96 */
    }
}

/* ------------------------------------------------------- local-map assembly */
/* transformPointCloud, MO:849-868: pose = [roll,pitch,yaw,x,y,z] (PointTypePose
 * fields), xyzi packed [n][4]. */
void lo_transform_point_cloud(const float *in_xyzi, size_t n, const float pose[6], float *out_xyzi, int trig_mode)
{
    float T[12];
    lo_get_transformation(pose[3], pose[4], pose[5], pose[0], pose[1], pose[2], T, trig_mode);  /* MO:856 */
    for (size_t i = 0; i < n; ++i) {                                    /* MO:858-866 */
        const float *p = in_xyzi + 4 * i;
        float *o = out_xyzi + 4 * i;
        o[0] = T[0] * p[0] + T[1] * p[1] + T[2]  * p[2] + T[3];
        o[1] = T[4] * p[0] + T[5] * p[1] + T[6]  * p[2] + T[7];
        o[2] = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
        o[3] = p[3];
    }
}

/* pcl::VoxelGrid<PointXYZI>::applyFilter (PCL 1.10, all fields, no filter
 * field, min_points_per_voxel 0) as used by downSizeFilterSurf (MO:1605-1611)
 * and downSizeFilterSurroundingKeyFrames (MO:1581-1583).  Voxel index =
 * x-fastest linear index over the cloud's own bounding box; output in
 * ascending voxel index; centroid = fp32 running sums / count.  PCL sorts the
 * (voxel, point) pairs with std::sort, whose order among equal voxels is
 * unspecified; this restatement sums a voxel's points in ascending input order.
 * Returns 0, or 1 when PCL would pass the cloud through unfiltered (voxel
 * index overflow, "Leaf size is too small"). */
typedef struct { uint32_t idx; uint32_t pt; } lo_vox_pair;
static int lo_vox_cmp(const void *a, const void *b)
{
    const lo_vox_pair *x = (const lo_vox_pair *)a, *y = (const lo_vox_pair *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0);
}

int lo_voxel_grid(const float *in_xyzi, size_t n, float leaf, float *out_xyzi, size_t *n_out)
{
    *n_out = 0;
    if (n == 0) return 0;
    const float inv = 1.0f / leaf;                                      /* inverse_leaf_size_ */
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (size_t i = 0; i < n; ++i)                                      /* getMinMax3D */
        for (int a = 0; a < 3; ++a) {
            float v = in_xyzi[4 * i + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1,
            dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) {                            /* passes the input through */
        memcpy(out_xyzi, in_xyzi, sizeof(float) * 4 * n);
        *n_out = n;
        return 1;
    }
    int min_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = (int)floorf(mn[a] * inv);
        div_b[a] = (int)floorf(mx[a] * inv) - min_b[a] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    lo_vox_pair *pairs = (lo_vox_pair *)malloc(sizeof(lo_vox_pair) * n);
    for (size_t i = 0; i < n; ++i) {
        const float *p = in_xyzi + 4 * i;
        int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        pairs[i].idx = (uint32_t)(i0 + i1 * mul1 + i2 * mul2);
        pairs[i].pt = (uint32_t)i;
    }
    qsort(pairs, n, sizeof(lo_vox_pair), lo_vox_cmp);
    size_t o = 0;
    for (size_t b = 0; b < n;) {
        size_t e = b;
        float s[4] = { 0, 0, 0, 0 };
        while (e < n && pairs[e].idx == pairs[b].idx) {
            const float *p = in_xyzi + 4 * (size_t)pairs[e].pt;
            s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];     /* AccumulatorXYZ / Intensity */
            ++e;
        }
        const float cnt = (float)(e - b);
        out_xyzi[4 * o + 0] = s[0] / cnt; out_xyzi[4 * o + 1] = s[1] / cnt;
        out_xyzi[4 * o + 2] = s[2] / cnt; out_xyzi[4 * o + 3] = s[3] / cnt;
        ++o;
        b = e;
    }
    free(pairs);
    *n_out = o;
    return 0;
}

/* =========================================================================
 * EXTENSION BEYOND THE REFERENCE -- the range-image build.
 *
 * BASELINE.json's north_star names "imageProjection's deskew/range-image build".  This fork's
 * projectPointCloud IP:577-615 no longer builds a range image (it filters and deskews in input
 * order), and nothing in it fills startRingIndex / endRingIndex / pointColInd / pointRange
 * (MSG:4-8), which its featureExtraction.cpp still reads.  What follows restates projectPointCloud
 * + cloudExtraction of upstream LIO-SAM (TixiaoShan/LIO-SAM imageProjection.cpp, Velodyne/Ouster
 * column rule; not present under /root/reference) on top of THIS fork's deskewPoint IP:545-575.
 * There is no reference oracle for it in this tree: parity unpinned, self-consistency tests only.
 *
 *   per input point, in order: range = |p|, drop if outside [lidarMinRange, lidarMaxRange]; row = ring,
 *   drop if outside [0, N_SCAN) or row % downsampleRate != 0; column from the azimuth,
 *   horizonAngle = atan2(x, y) * 180 / pi, col = -round((horizonAngle - 90) / (360 / H)) + H / 2,
 *   wrapped once, dropped if outside [0, H); a cell keeps the FIRST point that lands in it; that
 *   point is deskewed (the reference transform comes from the first point that gets this far).
 *   cloudExtraction: ring-major, ascending column; startRingIndex = first - 1 + 5, endRingIndex =
 *   last - 5; pointColInd / pointRange per kept point (range of the RAW point).
 * atan2 is defined as the fp64 function rounded once to fp32, like the trig of the pose.
 * ========================================================================= */
size_t lo_range_image(const lo_deskew_config *cfg, int horizon_scan, float lidar_min_range,
                      const float *x, const float *y, const float *z,
                      const float *intensity, const uint16_t *ring, const float *time,
                      size_t n, double time_scan_cur,
                      const double *imuTime, const double *imuRotX,
                      const double *imuRotY, const double *imuRotZ, int imuPointerCur,
                      float *out_xyzi, int32_t *startRingIndex, int32_t *endRingIndex,
                      int32_t *pointColInd, float *pointRange)
{
    const int H = horizon_scan, NS = cfg->N_SCAN;
    const size_t cells = (size_t)H * (size_t)NS;
    float *rangeMat = (float *)malloc(sizeof(float) * (cells ? cells : 1));
    float *full = (float *)malloc(sizeof(float) * 4 * (cells ? cells : 1));
    for (size_t c = 0; c < cells; ++c) rangeMat[c] = FLT_MAX;
    int first_point = 1;
    float startInv[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    const float ang_res_x = (float)(360.0 / (double)(float)H);
    for (size_t i = 0; i < n; ++i) {
        float px = x[i], py = y[i], pz = z[i], pi = intensity[i];
        float range = sqrtf(px * px + py * py + pz * pz);
        if (range < lidar_min_range || range > cfg->lidarMaxRange) continue;
        int rowIdn = ring[i];
        if (rowIdn < 0 || rowIdn >= NS) continue;
        if (rowIdn % cfg->downsampleRate != 0) continue;
        float at = (float)atan2((double)px, (double)py);               /* atan2(float, float) */
        float at180 = at * 180;                                         /* float * int */
        float horizonAngle = (float)((double)at180 / M_PI);             /* ... / M_PI in double, stored as float */
        int columnIdn = (int)(-round(((double)horizonAngle - 90.0) / (double)ang_res_x) + (double)(H / 2));
        if (columnIdn >= H) columnIdn -= H;
        if (columnIdn < 0 || columnIdn >= H) continue;
        const size_t index = (size_t)columnIdn + (size_t)rowIdn * (size_t)H;
        if (rangeMat[index] != FLT_MAX) continue;

        float ox = px, oy = py, oz = pz;
        if (!(cfg->deskew_flag == -1 || !cfg->imu_available)) {         /* deskewPoint, IP:545-575 */
            double pointTime = time_scan_cur + (double)time[i];
            float rx, ry, rz;
            lo_find_rotation(pointTime, imuTime, imuRotX, imuRotY, imuRotZ, imuPointerCur, &rx, &ry, &rz);
            float T[12];
            lo_get_transformation(0, 0, 0, rx, ry, rz, T, cfg->trig_mode);
            float L[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] };
            if (first_point) { lo_inv3(L, startInv); first_point = 0; }
            float Bt[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    Bt[r * 3 + c] = startInv[r * 3 + 0] * L[0 * 3 + c] + startInv[r * 3 + 1] * L[1 * 3 + c]
                                  + startInv[r * 3 + 2] * L[2 * 3 + c];
            ox = Bt[0] * px + Bt[1] * py + Bt[2] * pz + 0.0f;
            oy = Bt[3] * px + Bt[4] * py + Bt[5] * pz + 0.0f;
            oz = Bt[6] * px + Bt[7] * py + Bt[8] * pz + 0.0f;
        }
        rangeMat[index] = range;
        full[4 * index + 0] = ox; full[4 * index + 1] = oy; full[4 * index + 2] = oz; full[4 * index + 3] = pi;
    }
    size_t count = 0;                                                    /* cloudExtraction */
    for (int i = 0; i < NS; ++i) {
        startRingIndex[i] = (int32_t)count - 1 + 5;
        for (int j = 0; j < H; ++j) {
            const size_t index = (size_t)j + (size_t)i * (size_t)H;
            if (rangeMat[index] != FLT_MAX) {
                pointColInd[count] = j;
                pointRange[count] = rangeMat[index];
                memcpy(out_xyzi + 4 * count, full + 4 * index, sizeof(float) * 4);
                ++count;
            }
        }
        endRingIndex[i] = (int32_t)count - 1 - 5;
    }
    free(rangeMat); free(full);
    return count;
}

/* =========================================================================
 * featureExtraction.cpp, the rest of laserCloudInfoHandler FE:67-79:
 * markOccludedPoints FE:103-139 and extractFeatures FE:141-238, consuming the cloud_info arrays
 * (startRingIndex, endRingIndex, pointColInd, pointRange; MSG:4-8) exactly as the reference does.
 * The reference has no PRODUCER of those arrays, so the tests feed synthetic ones.
 *
 * Three places where the reference's behaviour is not a function of its inputs, and what is fixed here:
 *  - with the producer upstream LIO-SAM has (startRingIndex[0] = 0 - 1 + 5 = 4), ring 0's first sector
 *    starts at index 4: FE:183/215 then read pointColInd[-1] and FE:186/221 may write
 *    cloudNeighborPicked[-1] (and the same past the end for the last ring).  Here a neighbour walk
 *    stops at the array boundary, and curvature outside [5, n-5) is 0.
 *  - cloudNeighborPicked / cloudLabel / cloudSmoothness are `new[]`-allocated and only indices
 *    [5, n-5) are reset per scan (FE:95-99): indices outside keep whatever earlier scans (or the
 *    allocator) left.  Here every scan starts from zeros.
 *  - std::sort FE:162 does not define the order of equal curvatures.  Here ties keep ascending point
 *    index (a stable sort), which is one of the orders std::sort may produce.
 * ========================================================================= */
typedef struct { float value; int32_t ind; } lo_smooth;

static void lo_sort_smooth(lo_smooth *a, int n)      /* stable insertion/merge: n <= a few hundred */
{
    if (n < 2) return;
    lo_smooth *tmp = (lo_smooth *)malloc(sizeof(lo_smooth) * (size_t)n);
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (a[j].value < a[i].value) ? a[j++] : a[i++];
            while (i < mid) tmp[k++] = a[i++];
            while (j < hi) tmp[k++] = a[j++];
        }
        memcpy(a, tmp, sizeof(lo_smooth) * (size_t)n);
    }
    free(tmp);
}

void lo_mark_occluded(const float *pointRange, const int32_t *pointColInd, size_t n, int32_t *picked)
{
    const int cloudSize = (int)n;
    for (int i = 5; i < cloudSize - 6; ++i) {                                   /* FE:107 */
        float depth1 = pointRange[i], depth2 = pointRange[i + 1];
        int columnDiff = abs((int)(pointColInd[i + 1] - pointColInd[i]));
        if (columnDiff < 10) {                                                  /* FE:114 */
            if ((double)(depth1 - depth2) > 0.3) {
                for (int l = -5; l <= 0; ++l) picked[i + l] = 1;
            } else if ((double)(depth2 - depth1) > 0.3) {
                for (int l = 1; l <= 6; ++l) picked[i + l] = 1;
            }
        }
        float diff1 = fabsf(pointRange[i - 1] - pointRange[i]);                 /* FE:133-134 */
        float diff2 = fabsf(pointRange[i + 1] - pointRange[i]);
        if ((double)diff1 > 0.02 * (double)pointRange[i] && (double)diff2 > 0.02 * (double)pointRange[i])
            picked[i] = 1;
    }
}

/* Whole handler FE:67-77.  cloud = extractedCloud as (x,y,z,intensity)[n].  Outputs sized n each;
 * curvature/picked/label may be NULL.  Returns 0, or -1 when a ring's [start, end] lies outside [0, n-1]. */
int lo_extract_features(const float *cloud_xyzi, size_t n, int n_scan,
                        const int32_t *startRingIndex, const int32_t *endRingIndex,
                        const int32_t *pointColInd, const float *pointRange,
                        float edgeThreshold, float surfThreshold, float surfLeaf,
                        float *corner_xyzi, size_t *n_corner, float *surf_xyzi, size_t *n_surf,
                        float *curvature_out, int32_t *picked_out, int32_t *label_out)
{
    *n_corner = 0; *n_surf = 0;
    for (int i = 0; i < n_scan; ++i)
        if (endRingIndex[i] >= startRingIndex[i] && (startRingIndex[i] < 0 || (long)endRingIndex[i] > (long)n - 1)) return -1;
    size_t nn = n ? n : 1;
    float *curv = (float *)calloc(nn, sizeof(float));
    int32_t *picked = (int32_t *)calloc(nn, sizeof(int32_t)), *label = (int32_t *)calloc(nn, sizeof(int32_t));
    lo_smooth *sm = (lo_smooth *)calloc(nn, sizeof(lo_smooth));
    float *ring_in = (float *)malloc(sizeof(float) * 4 * nn), *ring_out = (float *)malloc(sizeof(float) * 4 * nn);
    if (n) lo_calculate_smoothness(pointRange, n, curv, NULL, NULL);            /* FE:81-101 */
    for (size_t i = 0; i < n; ++i) { sm[i].value = curv[i]; sm[i].ind = (int32_t)i; }
    lo_mark_occluded(pointRange, pointColInd, n, picked);

    for (int i = 0; i < n_scan; ++i) {                                          /* FE:149 */
        size_t n_ring = 0;
        for (int j = 0; j < 6; ++j) {
            int sp = (startRingIndex[i] * (6 - j) + endRingIndex[i] * j) / 6;                 /* FE:156 */
            int ep = (startRingIndex[i] * (5 - j) + endRingIndex[i] * (j + 1)) / 6 - 1;       /* FE:157 */
            if (sp >= ep) continue;
            lo_sort_smooth(sm + sp, ep - sp);                                   /* FE:162: [sp, ep) -- element ep stays */
            int largestPickedNum = 0;
            for (int k = ep; k >= sp; --k) {                                    /* FE:165 */
                int ind = sm[k].ind;
                if (picked[ind] == 0 && curv[ind] > edgeThreshold) {
                    ++largestPickedNum;
                    if (largestPickedNum <= 20) {
                        label[ind] = 1;
                        memcpy(corner_xyzi + 4 * (*n_corner), cloud_xyzi + 4 * (size_t)ind, sizeof(float) * 4);
                        ++*n_corner;
                    } else {
                        break;
                    }
                    picked[ind] = 1;
                    for (int l = 1; l <= 5; ++l) {
                        if (ind + l >= (int)n) break;
                        if (abs((int)(pointColInd[ind + l] - pointColInd[ind + l - 1])) > 10) break;
                        picked[ind + l] = 1;
                    }
                    for (int l = -1; l >= -5; --l) {
                        if (ind + l < 0) break;
                        if (abs((int)(pointColInd[ind + l] - pointColInd[ind + l + 1])) > 10) break;
                        picked[ind + l] = 1;
                    }
                }
            }
            for (int k = sp; k <= ep; ++k) {                                    /* FE:197 */
                int ind = sm[k].ind;
                if (picked[ind] == 0 && curv[ind] < surfThreshold) {
                    label[ind] = -1;
                    picked[ind] = 1;
                    for (int l = 1; l <= 5; ++l) {
                        if (ind + l >= (int)n) break;
                        if (abs((int)(pointColInd[ind + l] - pointColInd[ind + l - 1])) > 10) break;
                        picked[ind + l] = 1;
                    }
                    for (int l = -1; l >= -5; --l) {
                        if (ind + l < 0) break;
                        if (abs((int)(pointColInd[ind + l] - pointColInd[ind + l + 1])) > 10) break;
                        picked[ind + l] = 1;
                    }
                }
            }
            for (int k = sp; k <= ep; ++k)                                      /* FE:224: label by POSITION k */
                if (label[k] <= 0) { memcpy(ring_in + 4 * n_ring, cloud_xyzi + 4 * (size_t)k, sizeof(float) * 4); ++n_ring; }
        }
        size_t n_ds = 0;                                                        /* FE:232-236 */
        lo_voxel_grid(ring_in, n_ring, surfLeaf, ring_out, &n_ds);
        memcpy(surf_xyzi + 4 * (*n_surf), ring_out, sizeof(float) * 4 * n_ds);
        *n_surf += n_ds;
    }
    if (curvature_out) memcpy(curvature_out, curv, sizeof(float) * n);
    if (picked_out) memcpy(picked_out, picked, sizeof(int32_t) * n);
    if (label_out) memcpy(label_out, label, sizeof(int32_t) * n);
    free(curv); free(picked); free(label); free(sm); free(ring_in); free(ring_out);
    return 0;
}

/* =========================================================================
 * EXTENSION BEYOND THE REFERENCE -- point-to-line ("corner") residuals.
 *
 * BASELINE.json's north_star names `cornerOptimization`, but this reference (a liorf fork) has
 * none: its scan-to-map loop is surf-only.  What follows restates the
 * cornerOptimization of upstream LIO-SAM (TixiaoShan/LIO-SAM mapOptmization.cpp, the code liorf
 * was forked from; not present under /root/reference), so that the kernels can be exercised with
 * both residual types.  There is NO reference oracle for it in this tree: parity unpinned,
 * self-consistency tests only.  With no corner input everything above is unchanged.
 *
 * upstream: 5-NN in the corner map, gate sqDis[4] < 1.0, centroid + covariance of the five
 * neighbours (each /5), cv::eigen 3x3, line iff lambda0 > 3*lambda1, two points +-0.1 along the
 * principal axis, point-to-line distance ld2 = a012 / l12 and its gradient (la, lb, lc),
 * s = 1 - 0.9*fabs(ld2), coeff = s*(la, lb, lc, ld2), accept iff s > 0.1.
 * ========================================================================= */

/* cv::eigen for a symmetric 3x3 CV_32F matrix: the same JacobiImpl_ as lo_eigen6_sym with n = 3 */
void lo_eigen3_sym(const float A_in[9], float W[3], float V[9])
{
    enum { N = 3 };
    float A[N][N];
    int indR[N], indC[N];
    const float eps = FLT_EPSILON;
    int i, j, k, m;
    float mv;
    for (i = 0; i < N; ++i) for (j = 0; j < N; ++j) { A[i][j] = A_in[i * N + j]; V[i * N + j] = (i == j) ? 1.0f : 0.0f; }
    for (k = 0; k < N; ++k) {
        W[k] = A[k][k];
        if (k < N - 1) {
            for (m = k + 1, mv = fabsf(A[k][m]), i = k + 2; i < N; ++i) {
                float val = fabsf(A[k][i]);
                if (mv < val) { mv = val; m = i; }
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabsf(A[0][k]), i = 1; i < k; ++i) {
                float val = fabsf(A[i][k]);
                if (mv < val) { mv = val; m = i; }
            }
            indC[k] = m;
        }
    }
    for (int iters = 0; iters < N * N * 30; ++iters) {
        for (k = 0, mv = fabsf(A[0][indR[0]]), i = 1; i < N - 1; ++i) {
            float val = fabsf(A[i][indR[i]]);
            if (mv < val) { mv = val; k = i; }
        }
        int l = indR[k];
        for (i = 1; i < N; ++i) {
            float val = fabsf(A[indC[i]][i]);
            if (mv < val) { mv = val; k = indC[i]; l = i; }
        }
        float p = A[k][l];
        if (fabsf(p) <= eps) break;
        float y = (float)((W[l] - W[k]) * 0.5);
        float t = fabsf(y) + lo_cv_hypot(p, y);
        float s = lo_cv_hypot(p, t);
        float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) { s = -s; t = -t; }
        A[k][l] = 0;
        W[k] -= t;
        W[l] += t;
        float a0, b0;
#define LO_ROT(v0, v1) do { a0 = (v0); b0 = (v1); (v0) = a0 * c - b0 * s; (v1) = a0 * s + b0 * c; } while (0)
        for (i = 0; i < k; ++i)     LO_ROT(A[i][k], A[i][l]);
        for (i = k + 1; i < l; ++i) LO_ROT(A[k][i], A[i][l]);
        for (i = l + 1; i < N; ++i) LO_ROT(A[k][i], A[l][i]);
        for (i = 0; i < N; ++i)     LO_ROT(V[k * N + i], V[l * N + i]);
#undef LO_ROT
        for (j = 0; j < 2; ++j) {
            int idx = j == 0 ? k : l;
            if (idx < N - 1) {
                for (m = idx + 1, mv = fabsf(A[idx][m]), i = idx + 2; i < N; ++i) {
                    float val = fabsf(A[idx][i]);
                    if (mv < val) { mv = val; m = i; }
                }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = fabsf(A[0][idx]), i = 1; i < idx; ++i) {
                    float val = fabsf(A[i][idx]);
                    if (mv < val) { mv = val; m = i; }
                }
                indC[idx] = m;
            }
        }
    }
    for (k = 0; k < N - 1; ++k) {
        m = k;
        for (i = k + 1; i < N; ++i) if (W[m] < W[i]) m = i;
        if (k != m) {
            float t = W[m]; W[m] = W[k]; W[k] = t;
            for (i = 0; i < N; ++i) { t = V[m * N + i]; V[m * N + i] = V[k * N + i]; V[k * N + i] = t; }
        }
    }
}

static void lo_corner_point(const lo_s2m_config *cfg, const float T[12], const float *pointOri,
                            const float *map_xyz, size_t n_map, const lo_kdtree *tree,
                            uint8_t *flag, float *coeff, int32_t *nn)
{
    float pointSel[3];
    int32_t ind[5];
    float sq[5];
    *flag = 0;
    coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0.0f;
    for (int j = 0; j < 5; ++j) nn[j] = -1;
    lo_point_associate(T, pointOri, pointSel);
    if (cfg->knn_mode == 0 || tree == NULL) lo_knn5_brute(map_xyz, n_map, pointSel, ind, sq);
    else                                    lo_kdtree_knn5(tree, pointSel, ind, sq);
    if (!((double)sq[4] < (double)cfg->max_sq_dist)) return;
    for (int j = 0; j < 5; ++j) nn[j] = ind[j];

    float cx = 0, cy = 0, cz = 0;
    for (int j = 0; j < 5; ++j) {
        cx += map_xyz[3 * (size_t)ind[j] + 0]; cy += map_xyz[3 * (size_t)ind[j] + 1]; cz += map_xyz[3 * (size_t)ind[j] + 2];
    }
    cx /= 5; cy /= 5; cz /= 5;
    float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
    for (int j = 0; j < 5; ++j) {
        float ax = map_xyz[3 * (size_t)ind[j] + 0] - cx;
        float ay = map_xyz[3 * (size_t)ind[j] + 1] - cy;
        float az = map_xyz[3 * (size_t)ind[j] + 2] - cz;
        a11 += ax * ax; a12 += ax * ay; a13 += ax * az;
        a22 += ay * ay; a23 += ay * az;
        a33 += az * az;
    }
    a11 /= 5; a12 /= 5; a13 /= 5; a22 /= 5; a23 /= 5; a33 /= 5;
    float A1[9] = { a11, a12, a13, a12, a22, a23, a13, a23, a33 }, D1[3], V1[9];
    lo_eigen3_sym(A1, D1, V1);
    if (!(D1[0] > 3 * D1[1])) return;

    float x0 = pointSel[0], y0 = pointSel[1], z0 = pointSel[2];
    /* `cx + 0.1 * v`: 0.1 is a double literal, the sum is rounded to float once */
    float x1 = (float)(cx + 0.1 * V1[0]), y1 = (float)(cy + 0.1 * V1[1]), z1 = (float)(cz + 0.1 * V1[2]);
    float x2 = (float)(cx - 0.1 * V1[0]), y2 = (float)(cy - 0.1 * V1[1]), z2 = (float)(cz - 0.1 * V1[2]);

    float m1 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1);
    float m2 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1);
    float m3 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
    float a012 = sqrtf(m1 * m1 + m2 * m2 + m3 * m3);
    float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
    float la = ((y1 - y2) * m1 + (z1 - z2) * m2) / a012 / l12;
    float lb = -((x1 - x2) * m1 - (z1 - z2) * m3) / a012 / l12;
    float lc = -((x1 - x2) * m2 + (y1 - y2) * m3) / a012 / l12;
    float ld2 = a012 / l12;
    float s = (float)(1 - cfg->weight * (double)fabsf(ld2));
    coeff[0] = s * la; coeff[1] = s * lb; coeff[2] = s * lc; coeff[3] = s * ld2;
    if ((double)s > cfg->min_s) *flag = 1;
}

void lo_corner_optimization(const lo_s2m_config *cfg, const float pose[6],
                            const float *scan_xyz, size_t n_scan,
                            const float *map_xyz, size_t n_map, const lo_kdtree *tree,
                            uint8_t *flag, float *coeff, int32_t *nn_idx)
{
    float T[12];
    lo_get_transformation(pose[3], pose[4], pose[5], pose[0], pose[1], pose[2], T, cfg->trig_mode);
    long n = (long)n_scan;
    int nt = cfg->n_threads > 0 ? cfg->n_threads : 1;
    (void)nt;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
    for (long i = 0; i < n; ++i)
        lo_corner_point(cfg, T, scan_xyz + 3 * i, map_xyz, n_map, tree, flag + i, coeff + 4 * i, nn_idx + 5 * i);
}

/* scan2MapOptimization with both residual types (upstream order: corner correspondences first,
 * then surf, in combineOptimizationCoeffs).  n_corner == 0 reduces to lo_scan2map. */
int lo_scan2map_cs(const lo_s2m_config *cfg,
                   const float *corner_xyz, size_t n_corner, const float *cmap_xyz, size_t n_cmap,
                   const float *surf_xyz, size_t n_surf, const float *smap_xyz, size_t n_smap,
                   float pose[6], float matP_io[36], int32_t *is_degenerate_io, lo_s2m_result *res,
                   int corr_iter, uint8_t *cflag_out, float *ccoeff_out, int32_t *cnn_out)
{
    memset(res, 0, sizeof(*res));
    res->is_degenerate = *is_degenerate_io;
    memcpy(res->matP, matP_io, sizeof(float) * 36);
    if (cfg->max_iters < 1 || cfg->max_iters > 32) { res->status = -1; return res->status; }
    if (!((long)n_surf > (long)cfg->min_scan_pts)) { res->status = LO_TOO_FEW_POINTS; return res->status; }
    lo_kdtree *stree = cfg->knn_mode == 1 ? lo_kdtree_build(smap_xyz, n_smap) : NULL;
    lo_kdtree *ctree = (cfg->knn_mode == 1 && n_corner) ? lo_kdtree_build(cmap_xyz, n_cmap) : NULL;
    size_t n_all = n_corner + n_surf;
    uint8_t *flag = (uint8_t *)malloc(n_all ? n_all : 1);
    float *coeff = (float *)malloc(sizeof(float) * 4 * (n_all ? n_all : 1));
    int32_t *nn = (int32_t *)malloc(sizeof(int32_t) * 5 * (n_all ? n_all : 1));
    float *ori_sel = (float *)malloc(sizeof(float) * 3 * (n_all ? n_all : 1));
    float *coeff_sel = (float *)malloc(sizeof(float) * 4 * (n_all ? n_all : 1));
    const int max_iters = cfg->max_iters;
    for (int it = 0; it < max_iters; ++it) {
        if (n_corner) lo_corner_optimization(cfg, pose, corner_xyz, n_corner, cmap_xyz, n_cmap, ctree, flag, coeff, nn);
        lo_surf_optimization(cfg, pose, surf_xyz, n_surf, smap_xyz, n_smap, stree,
                             flag + n_corner, coeff + 4 * n_corner, nn + 5 * n_corner);
        if (it == corr_iter && n_corner) {
            if (cflag_out)  memcpy(cflag_out, flag, n_corner);
            if (ccoeff_out) memcpy(ccoeff_out, coeff, sizeof(float) * 4 * n_corner);
            if (cnn_out)    memcpy(cnn_out, nn, sizeof(int32_t) * 5 * n_corner);
        }
        int nc = 0;
        for (size_t i = 0; i < n_all; ++i)
            if (flag[i]) {
                const float *src = i < n_corner ? corner_xyz + 3 * i : surf_xyz + 3 * (i - n_corner);
                memcpy(ori_sel + 3 * (size_t)nc, src, sizeof(float) * 3);
                memcpy(coeff_sel + 4 * (size_t)nc, coeff + 4 * i, sizeof(float) * 4);
                ++nc;
            }
        res->n_corr_iter[it] = nc;
        res->n_corr_last = nc;
        int conv = lo_lm_optimization(cfg, it, ori_sel, coeff_sel, nc, pose, matP_io, is_degenerate_io,
                                      nc >= cfg->min_corr ? res->AtA : NULL, nc >= cfg->min_corr ? res->AtB : NULL);
        memcpy(res->pose_iter[it], pose, sizeof(float) * 6);
        res->iters = it + 1;
        if (conv) { res->converged = 1; if (!cfg->force_all_iters) break; }
    }
    res->is_degenerate = *is_degenerate_io;
    memcpy(res->matP, matP_io, sizeof(float) * 36);
    res->status = (res->n_corr_last < cfg->min_corr) ? LO_TOO_FEW_CORR : LO_OK;
    free(flag); free(coeff); free(nn); free(ori_sel); free(coeff_sel);
    lo_kdtree_free(stree); lo_kdtree_free(ctree);
    return res->status;
}
