/*
 * lio_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-thread-semantics CPU restatement of the scan-to-map
 * registration path of the reference (JiLiBIT/LIO-SLAM, a liorf fork):
 *
 *   MO = src/liorf/src/mapOptmization.cpp
 *   IP = src/liorf/src/imageProjection.cpp
 *   FE = src/liorf/src/featureExtraction.cpp
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (lio-slam_amd/) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned".  The reference holds no tests, golden
 * vectors or fixtures for this path and cannot be
 * built here (needs ROS1, PCL, Eigen, OpenCV, GTSAM).  The arithmetic that
 * lives in those absent third-party libraries is restated from their
 * published algorithms (see the comment at each function).  The oracle is
 * pinned only by closed-form checks (tests/test_oracle_*.py) and by
 * known-answer registration scenes.
 */
#ifndef LIO_ORACLE_H
#define LIO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- configuration of the scan-to-map loop (constants of MO) ---- */
typedef struct lo_s2m_config {
    int   k;              /* 5      MO:1631 (fixed to 5 in this restatement)   */
    float max_sq_dist;    /* 1.0    MO:1641                                    */
    double plane_tol;     /* 0.2    MO:1662  (double literal in the reference) */
    double weight;        /* 0.9    MO:1671  (double literal in the reference) */
    double min_s;         /* 0.1    MO:1679  (double literal in the reference) */
    int   min_corr;       /* 50     MO:1722                                    */
    int   max_iters;      /* 30     MO:1848                                    */
    float eig_thresh;     /* 100    MO:1796                                    */
    double conv_deg;      /* 0.05   MO:1833  (double literal)                  */
    double conv_cm;       /* 0.05   MO:1833  (double literal)                  */
    int   min_scan_pts;   /* 30     MO:1844  (N_s must be > this)              */
    int   jacobian_mode;  /* 0 = reference (MO:1764 as written), 1 = exact     */
    int   force_all_iters;/* ignore the convergence break MO:1857-1858         */
    int   n_threads;      /* OpenMP threads for the per-point loop MO:1622     */
    int   knn_mode;       /* 0 = brute force, 1 = kd-tree (both exact)         */
    int   trig_mode;      /* 0 = (float)sin((double)x) ["portable", default],  */
                          /* 1 = libm sinf/cosf (what `sin(float)` resolves to */
                          /*     in the reference's C++)                       */
} lo_s2m_config;

void lo_s2m_default_config(lo_s2m_config *cfg);

/* status codes (mirrors include/liogpu.h) */
enum {
    LO_OK = 0,
    LO_TOO_FEW_POINTS = 1,  /* N_s <= 30: MO:1844/1862-1864, pose unchanged */
    LO_TOO_FEW_CORR   = 2   /* last executed iteration had < 50 corr.       */
};

typedef struct lo_s2m_result {
    int32_t status;
    int32_t iters;          /* number of loop bodies executed (<=30)         */
    int32_t converged;      /* LMOptimization returned true                   */
    int32_t is_degenerate;  /* MO:176                                         */
    int32_t n_corr_last;    /* N_c of the last executed iteration             */
    int32_t n_corr_iter[32];/* N_c per iteration (debug)                      */
    float   matP[36];       /* MO:177, row-major, persists across calls       */
    float   AtA[36];        /* last iteration's normal matrix (row-major)     */
    float   AtB[6];
    float   pose_iter[32][6]; /* pose after each iteration (debug)            */
} lo_s2m_result;

/* opaque kd-tree */
typedef struct lo_kdtree lo_kdtree;
lo_kdtree *lo_kdtree_build(const float *xyz, size_t n);   /* packed xyz[n][3] */
void       lo_kdtree_free(lo_kdtree *t);
/* exact 5-NN, ascending (d2, index); returns number found (min(n,5)) */
int lo_kdtree_knn5(const lo_kdtree *t, const float q[3], int32_t idx[5], float d2[5]);
int lo_knn5_brute(const float *xyz, size_t n, const float q[3], int32_t idx[5], float d2[5]);

/* pcl::getTransformation (PCL common/eigen.hpp): T = 3x4 row-major */
void lo_get_transformation(float x, float y, float z, float roll, float pitch, float yaw,
                           float T[12], int trig_mode);
/* MO:841-847 */
void lo_point_associate(const float T[12], const float pi[3], float po[3]);

/* Eigen::Matrix<float,5,3>::colPivHouseholderQr().solve(b)  (Eigen 3.3 semantics) */
void lo_colpiv_qr_solve_5x3(const float A[15] /*row-major 5x3*/, const float b[5], float x[3]);
void lo_colpiv_qr_debug_5x3(const float A[15], int32_t perm[3], float rdiag[3], int32_t *nonzero_pivots);   /* test hook */

/* cv::solve(A,b,x,DECOMP_QR) for 6x6 CV_32F; returns 0 when singular (x=0) */
int  lo_solve6_qr(const float A[36], const float b[6], float x[6]);
/* cv::eigen for symmetric 6x6 CV_32F: eigenvalues descending, eigenvectors as rows */
void lo_eigen6_sym(const float A[36], float evals[6], float evecs[36]);
/* cv::Mat::inv() (DECOMP_LU) 6x6 CV_32F; returns 0 when singular (inv=0) */
int  lo_inv6_lu(const float A[36], float Ainv[36]);
/* CV_32F matrix product with double accumulation: C[m x n] = A[m x k] * B[k x n] */
void lo_gemm32f(const float *A, const float *B, float *C, int m, int k, int n);

/*
 * surfOptimization, MO:1618-1687, for all scan points.
 * scan/map: packed xyz (intensity is not used by the arithmetic).
 * Outputs per scan point i (caller-allocated, n_scan entries):
 *   flag[i]      laserCloudOriSurfFlag
 *   coeff[i*4..] coeffSelSurfVec[i] = (s*pa, s*pb, s*pc, s*pd2)
 *   nn_idx[i*5..] the 5 neighbour indices used (valid when sqDis[4] < 1.0), -1 else
 */
void lo_surf_optimization(const lo_s2m_config *cfg, const float pose[6],
                          const float *scan_xyz, size_t n_scan,
                          const float *map_xyz, size_t n_map, const lo_kdtree *tree,
                          uint8_t *flag, float *coeff, int32_t *nn_idx);

/*
 * LMOptimization, MO:1702-1837.  ori/coeff are the compacted (ascending i)
 * correspondences of combineOptimizationCoeffs MO:1689-1700.
 * pose is updated in place; matP/is_degenerate are the persistent members.
 * Returns 1 when converged.
 */
int lo_lm_optimization(const lo_s2m_config *cfg, int iter_count,
                       const float *ori_xyz, const float *coeff4, int n_corr,
                       float pose[6], float matP[36], int32_t *is_degenerate,
                       float AtA_out[36], float AtB_out[6]);

/* one row of matA / matB, MO:1735-1778: row = [arz, ary, arx, cx, cy, cz], rhs = -c.w */
void lo_jacobian_row(const float trig[6] /* srx,crx,sry,cry,srz,crz */, const float p[3],
                     const float c[4], int jacobian_mode, float row[6], float *rhs);

/*
 * scan2MapOptimization, MO:1839-1865 (without the keyframe-empty guard, which
 * stays in the caller, and without transformUpdate, exported separately).
 * matP_io / is_degenerate_io carry the persistent members MO:176-177.
 * If corr_* pointers are non-NULL they receive the association of iteration
 * `corr_iter` (flag[n_scan], coeff[n_scan*4], nn_idx[n_scan*5]).
 */
int lo_scan2map(const lo_s2m_config *cfg,
                const float *scan_xyz, size_t n_scan,
                const float *map_xyz, size_t n_map,
                float pose[6], float matP_io[36], int32_t *is_degenerate_io,
                lo_s2m_result *res,
                int corr_iter, uint8_t *corr_flag, float *corr_coeff, int32_t *corr_nn);

/* wall time (s) of the kd-tree build of the last lo_scan2map call in this process (MO:1846) */
double lo_last_kdtree_build_seconds(void);

/* transformUpdate + constraintTransformation, MO:1867-1907 */
void lo_transform_update(float pose[6], int imu_available, int imu_type,
                         float imu_roll_init, float imu_pitch_init, float imu_rpy_weight,
                         float rotation_tollerance, float z_tollerance);

/* ---- deskew path (IP) ---- */
typedef struct lo_deskew_config {
    int   N_SCAN;            /* UT:275 */
    int   downsampleRate;    /* UT:277 */
    int   point_filter_num;  /* UT:278 */
    float lidarMinFront, lidarMinBack, lidarMinLeft, lidarMinRight; /* UT:280-283 */
    float lidarMaxRange;     /* UT:284 */
    float lidarMaxIntensity; /* UT:285 */
    int   deskew_flag;       /* IP: deskewFlag (-1 = no per-point time) */
    int   imu_available;     /* cloudInfo.imuAvailable */
    int   trig_mode;
} lo_deskew_config;

/*
 * imuDeskewInfo, IP:359-418, over an already time-windowed IMU queue
 * (stamp[i], gyro xyz[i]); writes imuTime/imuRotX/Y/Z (capacity 2000) and
 * returns imuPointerCur (>0 means imuAvailable).
 */
int lo_imu_deskew_info(const double *stamp, const double *gx, const double *gy, const double *gz,
                       int n_imu, double time_scan_cur, double time_scan_end,
                       double *imuTime, double *imuRotX, double *imuRotY, double *imuRotZ);

/* findRotation, IP:502-527 */
void lo_find_rotation(double point_time, const double *imuTime, const double *imuRotX,
                      const double *imuRotY, const double *imuRotZ, int imuPointerCur,
                      float *rx, float *ry, float *rz);

/*
 * projectPointCloud + deskewPoint, IP:545-615.  Inputs are SoA views of
 * PointXYZIRT; output packed xyzi[n_out][4] in input order; returns n_out.
 * keep_idx (optional) receives the input index of every survivor.
 */
size_t lo_project_point_cloud(const lo_deskew_config *cfg,
                              const float *x, const float *y, const float *z,
                              const float *intensity, const uint16_t *ring, const float *time,
                              size_t n, double time_scan_cur,
                              const double *imuTime, const double *imuRotX,
                              const double *imuRotY, const double *imuRotZ, int imuPointerCur,
                              float *out_xyzi, int32_t *keep_idx);

/* ---- curvature (FE:81-101) ---- */
void lo_calculate_smoothness(const float *range, size_t n, float *curvature,
                             int32_t *neighbor_picked, int32_t *label);

/* ---- local-map assembly ---- */
/* transformPointCloud MO:849-868 on packed xyzi[n][4]; pose = [roll,pitch,yaw,x,y,z] */
void lo_transform_point_cloud(const float *in_xyzi, size_t n, const float pose[6], float *out_xyzi, int trig_mode);
/* pcl::VoxelGrid centroid filter (MO:1605-1611, MO:1581-1583); out has room for n points */
int  lo_voxel_grid(const float *in_xyzi, size_t n, float leaf, float *out_xyzi, size_t *n_out);

/* ---- EXTENSION beyond the reference: range-image build + cloudExtraction of upstream LIO-SAM (row A4;
 * parity unpinned).  out_xyzi / pointColInd / pointRange need room for N_SCAN * horizon_scan entries. */
size_t lo_range_image(const lo_deskew_config *cfg, int horizon_scan, float lidar_min_range,
                      const float *x, const float *y, const float *z,
                      const float *intensity, const uint16_t *ring, const float *time,
                      size_t n, double time_scan_cur,
                      const double *imuTime, const double *imuRotX,
                      const double *imuRotY, const double *imuRotZ, int imuPointerCur,
                      float *out_xyzi, int32_t *startRingIndex, int32_t *endRingIndex,
                      int32_t *pointColInd, float *pointRange);

/* featureExtraction.cpp FE:103-238 */
void lo_mark_occluded(const float *pointRange, const int32_t *pointColInd, size_t n, int32_t *picked);
int lo_extract_features(const float *cloud_xyzi, size_t n, int n_scan,
                        const int32_t *startRingIndex, const int32_t *endRingIndex,
                        const int32_t *pointColInd, const float *pointRange,
                        float edgeThreshold, float surfThreshold, float surfLeaf,
                        float *corner_xyzi, size_t *n_corner, float *surf_xyzi, size_t *n_surf,
                        float *curvature_out, int32_t *picked_out, int32_t *label_out);

/* ---- EXTENSION beyond the reference: point-to-line residuals (upstream LIO-SAM
 * cornerOptimization; absent from this fork; parity unpinned) ---- */
void lo_eigen3_sym(const float A[9], float evals[3], float evecs[9]);
void lo_corner_optimization(const lo_s2m_config *cfg, const float pose[6],
                            const float *scan_xyz, size_t n_scan,
                            const float *map_xyz, size_t n_map, const lo_kdtree *tree,
                            uint8_t *flag, float *coeff, int32_t *nn_idx);
int lo_scan2map_cs(const lo_s2m_config *cfg,
                   const float *corner_xyz, size_t n_corner, const float *cmap_xyz, size_t n_cmap,
                   const float *surf_xyz, size_t n_surf, const float *smap_xyz, size_t n_smap,
                   float pose[6], float matP_io[36], int32_t *is_degenerate_io, lo_s2m_result *res,
                   int corr_iter, uint8_t *cflag_out, float *ccoeff_out, int32_t *cnn_out);

#ifdef __cplusplus
}
#endif
#endif
