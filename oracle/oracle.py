"""ctypes loader for the CPU oracle (oracle/lio_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.  PARITY STATUS:
parity unpinned (see oracle/lio_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class S2MConfig(C.Structure):
    _fields_ = [
        ("k", C.c_int), ("max_sq_dist", C.c_float),
        ("plane_tol", C.c_double), ("weight", C.c_double), ("min_s", C.c_double),
        ("min_corr", C.c_int), ("max_iters", C.c_int), ("eig_thresh", C.c_float),
        ("conv_deg", C.c_double), ("conv_cm", C.c_double),
        ("min_scan_pts", C.c_int), ("jacobian_mode", C.c_int), ("force_all_iters", C.c_int),
        ("n_threads", C.c_int), ("knn_mode", C.c_int), ("trig_mode", C.c_int),
    ]


class S2MResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("iters", C.c_int32), ("converged", C.c_int32),
        ("is_degenerate", C.c_int32), ("n_corr_last", C.c_int32),
        ("n_corr_iter", C.c_int32 * 32),
        ("matP", C.c_float * 36), ("AtA", C.c_float * 36), ("AtB", C.c_float * 6),
        ("pose_iter", (C.c_float * 6) * 32),
    ]


class DeskewConfig(C.Structure):
    _fields_ = [
        ("N_SCAN", C.c_int), ("downsampleRate", C.c_int), ("point_filter_num", C.c_int),
        ("lidarMinFront", C.c_float), ("lidarMinBack", C.c_float),
        ("lidarMinLeft", C.c_float), ("lidarMinRight", C.c_float),
        ("lidarMaxRange", C.c_float), ("lidarMaxIntensity", C.c_float),
        ("deskew_flag", C.c_int), ("imu_available", C.c_int), ("trig_mode", C.c_int),
    ]


def build(fast=False, out_dir=None):
    """Compile the oracle with gcc (seconds).  fast=True -> -O3 -march=native."""
    out_dir = out_dir or _HERE
    name = "liblio_oracle_fast.so" if fast else "liblio_oracle.so"
    out = os.path.join(out_dir, name)
    src = os.path.join(_HERE, "lio_oracle.c")
    flags = ["-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
             "-D_GNU_SOURCE"]
    flags += ["-O3", "-march=native"] if fast else ["-O2"]
    subprocess.check_call(["gcc"] + flags + ["-o", out, src, "-lm"])
    return out


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "liblio_oracle.so")
        if not os.path.exists(path):
            path = build(fast=False)
        L = self.lib = C.CDLL(path)
        L.lo_s2m_default_config.argtypes = [C.POINTER(S2MConfig)]
        L.lo_kdtree_build.restype = C.c_void_p
        L.lo_kdtree_build.argtypes = [_f32p, C.c_size_t]
        L.lo_kdtree_free.argtypes = [C.c_void_p]
        L.lo_kdtree_knn5.argtypes = [C.c_void_p, _f32p, _i32p, _f32p]
        L.lo_knn5_brute.argtypes = [_f32p, C.c_size_t, _f32p, _i32p, _f32p]
        L.lo_get_transformation.argtypes = [C.c_float] * 6 + [_f32p, C.c_int]
        L.lo_point_associate.argtypes = [_f32p, _f32p, _f32p]
        L.lo_colpiv_qr_solve_5x3.argtypes = [_f32p, _f32p, _f32p]
        L.lo_colpiv_qr_debug_5x3.argtypes = [_f32p, _i32p, _f32p, C.POINTER(C.c_int32)]
        L.lo_solve6_qr.argtypes = [_f32p, _f32p, _f32p]
        L.lo_eigen6_sym.argtypes = [_f32p, _f32p, _f32p]
        L.lo_inv6_lu.argtypes = [_f32p, _f32p]
        L.lo_gemm32f.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int]
        L.lo_surf_optimization.argtypes = [C.POINTER(S2MConfig), _f32p, _f32p, C.c_size_t, _f32p,
                                           C.c_size_t, C.c_void_p, _u8p, _f32p, _i32p]
        L.lo_lm_optimization.argtypes = [C.POINTER(S2MConfig), C.c_int, _f32p, _f32p, C.c_int,
                                         _f32p, _f32p, _i32p, _f32p, _f32p]
        L.lo_jacobian_row.argtypes = [_f32p, _f32p, _f32p, C.c_int, _f32p, _f32p]
        L.lo_scan2map.argtypes = [C.POINTER(S2MConfig), _f32p, C.c_size_t, _f32p, C.c_size_t,
                                  _f32p, _f32p, _i32p, C.POINTER(S2MResult),
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_last_kdtree_build_seconds.restype = C.c_double
        L.lo_last_kdtree_build_seconds.argtypes = []
        L.lo_transform_update.argtypes = [_f32p, C.c_int, C.c_int] + [C.c_float] * 5
        L.lo_imu_deskew_info.restype = C.c_int
        L.lo_imu_deskew_info.argtypes = [_f64p, _f64p, _f64p, _f64p, C.c_int, C.c_double, C.c_double,
                                         _f64p, _f64p, _f64p, _f64p]
        L.lo_find_rotation.argtypes = [C.c_double, _f64p, _f64p, _f64p, _f64p, C.c_int,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.lo_project_point_cloud.restype = C.c_size_t
        L.lo_project_point_cloud.argtypes = [C.POINTER(DeskewConfig), _f32p, _f32p, _f32p, _f32p, _u16p,
                                             _f32p, C.c_size_t, C.c_double, _f64p, _f64p, _f64p, _f64p,
                                             C.c_int, _f32p, C.c_void_p]
        L.lo_calculate_smoothness.argtypes = [_f32p, C.c_size_t, _f32p, C.c_void_p, C.c_void_p]
        L.lo_transform_point_cloud.argtypes = [_f32p, C.c_size_t, _f32p, _f32p, C.c_int]
        L.lo_voxel_grid.argtypes = [_f32p, C.c_size_t, C.c_float, _f32p, C.POINTER(C.c_size_t)]
        L.lo_eigen3_sym.argtypes = [_f32p, _f32p, _f32p]
        L.lo_range_image.argtypes = [C.POINTER(DeskewConfig), C.c_int, C.c_float, _f32p, _f32p, _f32p, _f32p,
                                     np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS"), _f32p, C.c_size_t, C.c_double,
                                     _f64p, _f64p, _f64p, _f64p, C.c_int, _f32p, _i32p, _i32p, _i32p, _f32p]
        L.lo_range_image.restype = C.c_size_t
        L.lo_mark_occluded.argtypes = [_f32p, _i32p, C.c_size_t, _i32p]
        L.lo_extract_features.argtypes = [_f32p, C.c_size_t, C.c_int, _i32p, _i32p, _i32p, _f32p, C.c_float, C.c_float,
                                          C.c_float, _f32p, C.POINTER(C.c_size_t), _f32p, C.POINTER(C.c_size_t),
                                          _f32p, _i32p, _i32p]
        L.lo_extract_features.restype = C.c_int
        L.lo_corner_optimization.argtypes = [C.POINTER(S2MConfig), _f32p, _f32p, C.c_size_t, _f32p,
                                             C.c_size_t, C.c_void_p, _u8p, _f32p, _i32p]
        L.lo_scan2map_cs.argtypes = [C.POINTER(S2MConfig), _f32p, C.c_size_t, _f32p, C.c_size_t, _f32p, C.c_size_t,
                                     _f32p, C.c_size_t, _f32p, _f32p, _i32p, C.POINTER(S2MResult),
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

    def last_kdtree_build_seconds(self):
        """Wall time of the kd-tree build inside the last scan2map() call (MO:1846)."""
        return float(self.lib.lo_last_kdtree_build_seconds())

    # ---- helpers -----------------------------------------------------------
    def default_config(self, **kw):
        cfg = S2MConfig()
        self.lib.lo_s2m_default_config(C.byref(cfg))
        for k, v in kw.items():
            setattr(cfg, k, v)
        return cfg

    def get_transformation(self, x, y, z, roll, pitch, yaw, trig_mode=0):
        T = np.zeros(12, np.float32)
        self.lib.lo_get_transformation(x, y, z, roll, pitch, yaw, T, trig_mode)
        return T.reshape(3, 4)

    def knn5(self, map_xyz, queries, mode="kdtree"):
        map_xyz = np.ascontiguousarray(map_xyz, np.float32)
        queries = np.ascontiguousarray(queries, np.float32)
        idx = np.zeros((len(queries), 5), np.int32)
        d2 = np.zeros((len(queries), 5), np.float32)
        tree = self.lib.lo_kdtree_build(map_xyz, len(map_xyz)) if mode == "kdtree" else None
        try:
            for i, q in enumerate(queries):
                if tree:
                    self.lib.lo_kdtree_knn5(tree, q, idx[i], d2[i])
                else:
                    self.lib.lo_knn5_brute(map_xyz, len(map_xyz), q, idx[i], d2[i])
        finally:
            if tree:
                self.lib.lo_kdtree_free(tree)
        return idx, d2

    def plane_fit(self, pts5):
        A = np.ascontiguousarray(pts5, np.float32).reshape(15)
        b = np.full(5, -1.0, np.float32)
        x = np.zeros(3, np.float32)
        self.lib.lo_colpiv_qr_solve_5x3(A, b, x)
        return x

    def qr_solve_5x3(self, A, b):
        x = np.zeros(3, np.float32)
        self.lib.lo_colpiv_qr_solve_5x3(np.ascontiguousarray(A, np.float32).reshape(15),
                                        np.ascontiguousarray(b, np.float32), x)
        return x

    def qr_pivots_5x3(self, A):
        """Column order, diag(R) and nonzeroPivots() of the ColPivHouseholderQR restatement (test hook)."""
        perm = np.zeros(3, np.int32)
        rd = np.zeros(3, np.float32)
        nz = C.c_int32(0)
        self.lib.lo_colpiv_qr_debug_5x3(np.ascontiguousarray(A, np.float32).reshape(15), perm, rd, C.byref(nz))
        return perm, rd, int(nz.value)

    def solve6(self, A, b):
        x = np.zeros(6, np.float32)
        ok = self.lib.lo_solve6_qr(np.ascontiguousarray(A, np.float32).reshape(36),
                                   np.ascontiguousarray(b, np.float32), x)
        return x, ok

    def eigen6(self, A):
        w = np.zeros(6, np.float32)
        v = np.zeros(36, np.float32)
        self.lib.lo_eigen6_sym(np.ascontiguousarray(A, np.float32).reshape(36), w, v)
        return w, v.reshape(6, 6)

    def inv6(self, A):
        out = np.zeros(36, np.float32)
        ok = self.lib.lo_inv6_lu(np.ascontiguousarray(A, np.float32).reshape(36), out)
        return out.reshape(6, 6), ok

    def jacobian_row(self, pose, p, c, jacobian_mode=0):
        pose = np.asarray(pose, np.float64)
        f = lambda v: np.float32(v)
        trig = np.array([f(np.sin(pose[2])), f(np.cos(pose[2])), f(np.sin(pose[1])), f(np.cos(pose[1])),
                         f(np.sin(pose[0])), f(np.cos(pose[0]))], np.float32)
        row = np.zeros(6, np.float32)
        rhs = np.zeros(1, np.float32)
        self.lib.lo_jacobian_row(trig, np.ascontiguousarray(p, np.float32),
                                 np.ascontiguousarray(c, np.float32), jacobian_mode, row, rhs)
        return row, float(rhs[0])

    def surf_optimization(self, cfg, pose, scan_xyz, map_xyz):
        scan_xyz = np.ascontiguousarray(scan_xyz, np.float32)
        map_xyz = np.ascontiguousarray(map_xyz, np.float32)
        n = len(scan_xyz)
        flag = np.zeros(n, np.uint8)
        coeff = np.zeros((n, 4), np.float32)
        nn = np.zeros((n, 5), np.int32)
        tree = self.lib.lo_kdtree_build(map_xyz, len(map_xyz)) if cfg.knn_mode == 1 else None
        try:
            self.lib.lo_surf_optimization(C.byref(cfg), np.ascontiguousarray(pose, np.float32), scan_xyz, n,
                                          map_xyz, len(map_xyz), tree, flag, coeff.reshape(-1), nn.reshape(-1))
        finally:
            if tree:
                self.lib.lo_kdtree_free(tree)
        return flag, coeff, nn

    def scan2map(self, cfg, scan_xyz, map_xyz, pose, matP=None, is_degenerate=0, corr_iter=-1):
        """Returns (pose_out, result, matP, corr) -- corr is None unless corr_iter >= 0."""
        scan_xyz = np.ascontiguousarray(scan_xyz, np.float32)
        map_xyz = np.ascontiguousarray(map_xyz, np.float32)
        pose = np.array(pose, np.float32).copy()
        matP = np.zeros(36, np.float32) if matP is None else np.array(matP, np.float32).reshape(36).copy()
        deg = np.array([is_degenerate], np.int32)
        res = S2MResult()
        n = len(scan_xyz)
        corr = None
        a = b = c = None
        if corr_iter >= 0:
            flag = np.zeros(n, np.uint8)
            coeff = np.zeros((n, 4), np.float32)
            nn = np.full((n, 5), -1, np.int32)
            corr = (flag, coeff, nn)
            a, b, c = flag.ctypes.data, coeff.ctypes.data, nn.ctypes.data
        self.lib.lo_scan2map(C.byref(cfg), scan_xyz, n, map_xyz, len(map_xyz), pose, matP, deg,
                             C.byref(res), corr_iter, a, b, c)
        return pose, res, matP.reshape(6, 6), corr

    def transform_update(self, pose, imu_available=0, imu_type=0, imu_roll_init=0.0, imu_pitch_init=0.0,
                         imu_rpy_weight=0.01, rotation_tollerance=1000.0, z_tollerance=1000.0):
        pose = np.array(pose, np.float32).copy()
        self.lib.lo_transform_update(pose, imu_available, imu_type, imu_roll_init, imu_pitch_init,
                                     imu_rpy_weight, rotation_tollerance, z_tollerance)
        return pose

    def imu_deskew_info(self, stamp, gyro, t_cur, t_end):
        stamp = np.ascontiguousarray(stamp, np.float64)
        g = np.ascontiguousarray(gyro, np.float64)
        gx, gy, gz = (np.ascontiguousarray(g[:, k]) for k in range(3))
        T = np.zeros(2000); RX = np.zeros(2000); RY = np.zeros(2000); RZ = np.zeros(2000)
        cur = self.lib.lo_imu_deskew_info(stamp, gx, gy, gz, len(stamp), t_cur, t_end, T, RX, RY, RZ)
        return cur, T, RX, RY, RZ

    def project_point_cloud(self, dcfg, x, y, z, intensity, ring, time, t_cur, imu):
        cur, T, RX, RY, RZ = imu
        n = len(x)
        out = np.zeros((n, 4), np.float32)
        keep = np.zeros(n, np.int32)
        n_out = self.lib.lo_project_point_cloud(
            C.byref(dcfg), np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32),
            np.ascontiguousarray(z, np.float32), np.ascontiguousarray(intensity, np.float32),
            np.ascontiguousarray(ring, np.uint16), np.ascontiguousarray(time, np.float32), n, t_cur,
            T, RX, RY, RZ, cur, out.reshape(-1), keep.ctypes.data)
        return out[:n_out].copy(), keep[:n_out].copy()

    def calculate_smoothness(self, rng):
        rng = np.ascontiguousarray(rng, np.float32)
        curv = np.zeros(len(rng), np.float32)
        picked = np.full(len(rng), -1, np.int32)
        label = np.full(len(rng), -1, np.int32)
        self.lib.lo_calculate_smoothness(rng, len(rng), curv, picked.ctypes.data, label.ctypes.data)
        return curv, picked, label

    def transform_point_cloud(self, xyzi, pose, trig_mode=0):
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        out = np.zeros_like(xyzi)
        self.lib.lo_transform_point_cloud(xyzi.reshape(-1), len(xyzi), np.ascontiguousarray(pose, np.float32),
                                          out.reshape(-1), trig_mode)
        return out

    def voxel_grid(self, xyzi, leaf):
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        out = np.zeros_like(xyzi)
        n_out = C.c_size_t()
        rc = self.lib.lo_voxel_grid(xyzi.reshape(-1), len(xyzi), leaf, out.reshape(-1), C.byref(n_out))
        return out[:n_out.value].copy(), rc

    # ---- extension beyond the reference: point-to-line residuals (upstream LIO-SAM) ----
    def eigen3(self, A):
        w = np.zeros(3, np.float32)
        v = np.zeros(9, np.float32)
        self.lib.lo_eigen3_sym(np.ascontiguousarray(A, np.float32).reshape(9), w, v)
        return w, v.reshape(3, 3)

    def corner_optimization(self, cfg, pose, scan_xyz, map_xyz):
        scan_xyz = np.ascontiguousarray(scan_xyz, np.float32)
        map_xyz = np.ascontiguousarray(map_xyz, np.float32)
        n = len(scan_xyz)
        flag = np.zeros(n, np.uint8); coeff = np.zeros((n, 4), np.float32); nn = np.zeros((n, 5), np.int32)
        tree = self.lib.lo_kdtree_build(map_xyz, len(map_xyz)) if cfg.knn_mode == 1 else None
        try:
            self.lib.lo_corner_optimization(C.byref(cfg), np.ascontiguousarray(pose, np.float32), scan_xyz, n,
                                            map_xyz, len(map_xyz), tree, flag, coeff.reshape(-1), nn.reshape(-1))
        finally:
            if tree:
                self.lib.lo_kdtree_free(tree)
        return flag, coeff, nn

    def scan2map_cs(self, cfg, corner_scan, corner_map, surf_scan, surf_map, pose, corr_iter=-1):
        cs = np.ascontiguousarray(corner_scan, np.float32).reshape(-1, 3)
        cm = np.ascontiguousarray(corner_map, np.float32).reshape(-1, 3)
        ss = np.ascontiguousarray(surf_scan, np.float32)
        sm = np.ascontiguousarray(surf_map, np.float32)
        pose = np.array(pose, np.float32).copy()
        matP = np.zeros(36, np.float32); deg = np.zeros(1, np.int32); res = S2MResult()
        corr, a, b, c = None, None, None, None
        if corr_iter >= 0 and len(cs):
            flag = np.zeros(len(cs), np.uint8); coeff = np.zeros((len(cs), 4), np.float32)
            nn = np.full((len(cs), 5), -1, np.int32)
            corr = (flag, coeff, nn); a, b, c = flag.ctypes.data, coeff.ctypes.data, nn.ctypes.data
        dummy = np.zeros(3, np.float32)
        self.lib.lo_scan2map_cs(C.byref(cfg), cs.reshape(-1) if len(cs) else dummy, len(cs),
                                cm.reshape(-1) if len(cm) else dummy, len(cm), ss.reshape(-1), len(ss),
                                sm.reshape(-1), len(sm), pose, matP, deg, C.byref(res), corr_iter, a, b, c)
        return pose, res, matP.reshape(6, 6), corr

    # ---- featureExtraction.cpp FE:103-238 ----
    def mark_occluded(self, point_range, point_col):
        r = np.ascontiguousarray(point_range, np.float32)
        col = np.ascontiguousarray(point_col, np.int32)
        picked = np.zeros(len(r), np.int32)
        self.lib.lo_mark_occluded(r, col, len(r), picked)
        return picked

    def extract_features(self, cloud_xyzi, start_ring, end_ring, point_col, point_range,
                         edge_threshold=1.0, surf_threshold=0.1, surf_leaf=0.2):
        cloud = np.ascontiguousarray(cloud_xyzi, np.float32).reshape(-1, 4)
        n = len(cloud)
        sr = np.ascontiguousarray(start_ring, np.int32); er = np.ascontiguousarray(end_ring, np.int32)
        col = np.ascontiguousarray(point_col, np.int32); rng = np.ascontiguousarray(point_range, np.float32)
        corner = np.zeros((max(n, 1), 4), np.float32); surf = np.zeros((max(n, 1), 4), np.float32)
        nc, ns = C.c_size_t(0), C.c_size_t(0)
        curv = np.zeros(max(n, 1), np.float32); picked = np.zeros(max(n, 1), np.int32); label = np.zeros(max(n, 1), np.int32)
        rc = self.lib.lo_extract_features(cloud.reshape(-1) if n else np.zeros(4, np.float32), n, len(sr), sr, er, col, rng,
                                          edge_threshold, surf_threshold, surf_leaf, corner.reshape(-1), C.byref(nc),
                                          surf.reshape(-1), C.byref(ns), curv, picked, label)
        if rc != 0:
            raise ValueError("ring index ranges outside [0, n-1]")
        return {"corner": corner[:nc.value].copy(), "surface": surf[:ns.value].copy(),
                "curvature": curv[:n], "picked": picked[:n], "label": label[:n]}

    # ---- extension: range-image build + cloudExtraction of upstream LIO-SAM (row A4) ----
    def range_image(self, dcfg, horizon_scan, lidar_min_range, xyz, intensity, ring, time, t_cur, imu):
        cur, T, RX, RY, RZ = imu
        xyz = np.asarray(xyz, np.float32)
        cells = int(dcfg.N_SCAN) * int(horizon_scan)
        out = np.zeros((max(cells, 1), 4), np.float32)
        start = np.zeros(dcfg.N_SCAN, np.int32); end = np.zeros(dcfg.N_SCAN, np.int32)
        col = np.zeros(max(cells, 1), np.int32); rng = np.zeros(max(cells, 1), np.float32)
        n = self.lib.lo_range_image(C.byref(dcfg), horizon_scan, lidar_min_range,
                                    np.ascontiguousarray(xyz[:, 0]), np.ascontiguousarray(xyz[:, 1]), np.ascontiguousarray(xyz[:, 2]),
                                    np.ascontiguousarray(intensity, np.float32), np.ascontiguousarray(ring, np.uint16),
                                    np.ascontiguousarray(time, np.float32), len(xyz), t_cur, T, RX, RY, RZ, cur,
                                    out.reshape(-1), start, end, col, rng)
        return {"cloud": out[:n].copy(), "start_ring": start, "end_ring": end, "col": col[:n].copy(), "range": rng[:n].copy()}
