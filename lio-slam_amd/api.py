"""ctypes binding of include/liogpu.h -- one Python method per C entry point.

Method names follow the reference's member functions where one exists
(mapOptimization::scan2MapOptimization MO:1839, transformUpdate MO:1867,
ImageProjection::projectPointCloud IP:577, FeatureExtraction::calculateSmoothness
FE:81) so the parity tests read like tests of the reference's own classes.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIO_MAX_ITERS = 32
STATUS_NAMES = {0: "OK", 1: "TOO_FEW_POINTS", 2: "TOO_FEW_CORR", -1: "ERR_ARG", -2: "ERR_HIP",
                -3: "ERR_CAPACITY", -4: "ERR_NO_MAP", -5: "ERR_NO_DEVICE"}


class LioError(RuntimeError):
    pass


class S2MConfig(C.Structure):
    _fields_ = [
        ("k", C.c_int32), ("max_sq_dist", C.c_float),
        ("plane_tol", C.c_double), ("weight", C.c_double), ("min_s", C.c_double),
        ("min_corr", C.c_int32), ("max_iters", C.c_int32), ("eig_thresh", C.c_float),
        ("conv_deg", C.c_double), ("conv_cm", C.c_double),
        ("min_scan_pts", C.c_int32), ("jacobian_mode", C.c_int32), ("force_all_iters", C.c_int32),
        ("device_id", C.c_int32), ("cell_size", C.c_float), ("max_batch", C.c_int32),
        ("max_scan_pts", C.c_int32), ("record_corr_iter", C.c_int32), ("kernel_variant", C.c_int32),
        ("profile", C.c_int32), ("lookahead", C.c_int32), ("use_lds", C.c_int32), ("sort_scan", C.c_int32),
        ("cell_div", C.c_int32), ("xcd_remap", C.c_int32), ("tile_size", C.c_float),
        ("use_graph", C.c_int32), ("graph_iters", C.c_int32), ("sort_batch", C.c_int32), ("nn_cache", C.c_int32),
        ("pipeline", C.c_int32), ("n_devices", C.c_int32), ("device_ids", C.c_int32 * 8), ("x_sub", C.c_int32), ("tight_rows", C.c_int32),
    ]


class S2MResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("iters", C.c_int32), ("converged", C.c_int32),
        ("is_degenerate", C.c_int32), ("n_corr_last", C.c_int32),
        ("n_corr_iter", C.c_int32 * LIO_MAX_ITERS),
        ("matP", C.c_float * 36), ("AtA", C.c_float * 36), ("AtB", C.c_float * 6),
        ("pose_iter", (C.c_float * 6) * LIO_MAX_ITERS),
    ]


class S2MProfile(C.Structure):
    _fields_ = [
        ("map_build_ms", C.c_float), ("map_upload_ms", C.c_float), ("n_launches", C.c_int32),
        ("n_units", C.c_int32), ("unit_iters", C.c_int32),
        ("launch_ms", C.c_float * LIO_MAX_ITERS), ("launch_active", C.c_int32 * LIO_MAX_ITERS),
        ("point_iters", C.c_int64),
        ("n_map", C.c_int64), ("n_cells", C.c_int64),
        ("pipeline", C.c_int32), ("multi_iterations", C.c_int32), ("multi_stream_syncs", C.c_int32),
        ("multi_event_waits", C.c_int32), ("multi_exchange", C.c_int32), ("persist_fallbacks", C.c_int32),
        ("map_x_sub", C.c_int32), ("map_tight_tables", C.c_int32), ("map_first_try", C.c_int32), ("map_pts_per_cell", C.c_float),
    ]


class DeskewConfig(C.Structure):
    _fields_ = [
        ("N_SCAN", C.c_int32), ("downsampleRate", C.c_int32), ("point_filter_num", C.c_int32),
        ("lidarMinFront", C.c_float), ("lidarMinBack", C.c_float),
        ("lidarMinLeft", C.c_float), ("lidarMinRight", C.c_float),
        ("lidarMaxRange", C.c_float), ("lidarMaxIntensity", C.c_float),
        ("deskew_flag", C.c_int32), ("device_id", C.c_int32),
    ]


class RangeImageConfig(C.Structure):
    _fields_ = [("N_SCAN", C.c_int32), ("Horizon_SCAN", C.c_int32), ("downsampleRate", C.c_int32),
                ("lidarMinRange", C.c_float), ("lidarMaxRange", C.c_float), ("deskew_flag", C.c_int32),
                ("device_id", C.c_int32)]


class PC2Layout(C.Structure):
    """Field layout of a sensor_msgs/PointCloud2 (include/liogpu.h lio_pc2_layout)."""
    _fields_ = [("point_step", C.c_uint32), ("off_x", C.c_uint32), ("off_intensity", C.c_int32), ("off_ring", C.c_int32),
                ("ring_type", C.c_int32), ("off_time", C.c_int32), ("time_type", C.c_int32), ("pin_host", C.c_int32)]


PC2_UINT8, PC2_UINT16, PC2_INT32 = 2, 4, 5
PC2_TIME_F32_SECONDS, PC2_TIME_U32_NS, PC2_TIME_U32_RAW, PC2_TIME_F64_STAMP = 0, 1, 2, 3


class FeatureConfig(C.Structure):
    _fields_ = [("N_SCAN", C.c_int32), ("edgeThreshold", C.c_float), ("surfThreshold", C.c_float),
                ("surfLeafSize", C.c_float), ("device_id", C.c_int32)]


def lib_path():
    # LIOGPU_LIB: A/B-test another build of the same library (kernel experiments); default = the in-tree build
    return os.environ.get("LIOGPU_LIB") or os.path.join(_HERE, "libliogpu.so")


def build_library(verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU (seconds)."""
    out = subprocess.run(["make", "-j8", "-C", os.path.join(_HERE, "csrc")], capture_output=True, text=True)
    if out.returncode != 0:
        raise LioError("building libliogpu.so failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)
    return lib_path()


_LIB = None

# every symbol include/liogpu.h declares
EXPORTS = [
    "lio_version", "lio_s2m_default_config", "lio_last_error", "lio_s2m_create", "lio_s2m_destroy",
    "lio_s2m_set_map", "lio_s2m_register", "lio_s2m_batch_upload", "lio_s2m_batch_set_poses",
    "lio_s2m_batch_run", "lio_s2m_batch_sync", "lio_s2m_batch_results", "lio_s2m_set_degeneracy",
    "lio_s2m_get_correspondences", "lio_s2m_get_profile", "lio_s2m_set_stream",
    "lio_s2m_set_global_grid", "lio_s2m_set_shard", "lio_s2m_batch_begin",
    "lio_s2m_batch_iter_partial", "lio_s2m_batch_iter_apply", "lio_s2m_batch_n_active",
    "lio_transform_update", "lio_deskew_default_config", "lio_imu_deskew_info", "lio_deskew",
    "lio_curvature", "lio_s2m_debug_stamps", "lio_s2m_batch_poll_active", "lio_voxel_grid", "lio_assemble_map", "lio_kf_store_create", "lio_kf_store_destroy", "lio_kf_store_add",
    "lio_kf_store_count", "lio_kf_store_points", "lio_assemble_map_resident", "lio_s2m_set_scan_shard",
    "lio_s2m_set_corner_map", "lio_s2m_batch_upload_corners", "lio_s2m_register_cs",
    "lio_s2m_get_corner_correspondences", "lio_feature_default_config", "lio_extract_features",
    "lio_range_image_default_config", "lio_range_image",
    "lio_s2m_share_map", "lio_s2m_batch_upload_async", "lio_host_alloc", "lio_host_free", "lio_host_register",
    "lio_host_unregister", "lio_s2m_set_shard_plan", "lio_s2m_register_pc2", "lio_deskew_pc2", "lio_kf_store_add_device", "lio_kf_store_add_from_handle",
    "lio_s2m_register_raw", "lio_s2m_debug_persist_spin", "lio_device_alloc", "lio_device_free", "lio_device_upload",
]


def load_library():
    """Loads lio-slam_amd/libliogpu.so; raises LioError when it is missing (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise LioError(f"{p} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                       "There is no CPU fallback.")
    L = C.CDLL(p)
    vp, i32, f32, f64, sz = C.c_void_p, C.c_int32, C.c_float, C.c_double, C.c_size_t
    L.lio_version.restype = C.c_int
    L.lio_last_error.restype = C.c_char_p
    L.lio_s2m_default_config.argtypes = [C.POINTER(S2MConfig)]
    L.lio_s2m_default_config.restype = None
    L.lio_s2m_create.argtypes = [C.POINTER(S2MConfig), C.POINTER(vp)]
    L.lio_s2m_destroy.argtypes = [vp]
    L.lio_s2m_destroy.restype = None
    L.lio_s2m_set_map.argtypes = [vp, vp, sz, sz]
    L.lio_s2m_register.argtypes = [vp, vp, sz, sz, C.POINTER(f32), C.POINTER(S2MResult)]
    L.lio_s2m_register_pc2.argtypes = [vp, vp, sz, C.POINTER(PC2Layout), C.POINTER(f32), C.POINTER(S2MResult)]
    L.lio_s2m_register_raw.argtypes = [vp, vp, sz, C.POINTER(PC2Layout), f32, C.POINTER(f32), C.POINTER(S2MResult), vp, sz, C.POINTER(sz)]
    L.lio_s2m_debug_persist_spin.argtypes = [vp, i32, i32]
    L.lio_s2m_batch_upload.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz), sz]
    L.lio_s2m_batch_upload_async.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz), sz]
    L.lio_s2m_share_map.argtypes = [vp, vp]
    L.lio_host_alloc.argtypes = [sz]
    L.lio_host_alloc.restype = vp
    L.lio_host_free.argtypes = [vp]
    L.lio_host_free.restype = None
    L.lio_device_alloc.argtypes = [i32, sz]
    L.lio_device_alloc.restype = vp
    L.lio_device_free.argtypes = [i32, vp]
    L.lio_device_free.restype = None
    L.lio_device_upload.argtypes = [i32, vp, vp, sz]
    L.lio_host_register.argtypes = [vp, sz]
    L.lio_host_unregister.argtypes = [vp]
    L.lio_s2m_batch_set_poses.argtypes = [vp, C.POINTER(f32)]
    L.lio_s2m_batch_run.argtypes = [vp]
    L.lio_s2m_batch_sync.argtypes = [vp]
    L.lio_s2m_batch_results.argtypes = [vp, C.POINTER(f32), C.POINTER(S2MResult)]
    L.lio_s2m_set_degeneracy.argtypes = [vp, i32, C.POINTER(f32), i32]
    L.lio_s2m_get_correspondences.argtypes = [vp, i32, vp, vp, vp]
    L.lio_s2m_get_profile.argtypes = [vp, C.POINTER(S2MProfile)]
    L.lio_s2m_set_corner_map.argtypes = [vp, vp, sz, sz]
    L.lio_s2m_batch_upload_corners.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz), sz]
    L.lio_s2m_register_cs.argtypes = [vp, vp, sz, vp, sz, sz, C.POINTER(f32), C.POINTER(S2MResult)]
    L.lio_s2m_get_corner_correspondences.argtypes = [vp, i32, vp, vp, vp]
    L.lio_s2m_set_stream.argtypes = [vp, vp]
    L.lio_s2m_set_global_grid.argtypes = [vp, C.POINTER(f32), C.POINTER(i32)]
    L.lio_s2m_set_shard.argtypes = [vp, i32, i32, i32]
    L.lio_s2m_set_scan_shard.argtypes = [vp, i32, i32]
    L.lio_s2m_set_shard_plan.argtypes = [vp, i32, i32, i32, C.POINTER(i32), i32]
    L.lio_s2m_batch_begin.argtypes = [vp]
    L.lio_s2m_batch_iter_partial.argtypes = [vp, vp]
    L.lio_s2m_batch_iter_apply.argtypes = [vp, vp]
    L.lio_s2m_batch_n_active.argtypes = [vp, C.POINTER(i32)]
    L.lio_s2m_batch_poll_active.argtypes = [vp, i32, C.POINTER(i32)]
    L.lio_transform_update.argtypes = [C.POINTER(f32), i32, i32, f32, f32, f32, f32, f32]
    L.lio_transform_update.restype = None
    L.lio_deskew_default_config.argtypes = [C.POINTER(DeskewConfig)]
    L.lio_deskew_default_config.restype = None
    dp = C.POINTER(f64)
    L.lio_imu_deskew_info.argtypes = [dp, dp, dp, dp, i32, f64, f64, dp, dp, dp, dp]
    L.lio_deskew.argtypes = [C.POINTER(DeskewConfig), vp, sz, sz, f64, dp, dp, dp, dp, i32, vp, sz,
                             C.POINTER(sz)]
    L.lio_deskew_pc2.argtypes = [C.POINTER(DeskewConfig), vp, sz, C.POINTER(PC2Layout), f64, dp, dp, dp, dp, i32, vp, sz,
                                 C.POINTER(sz)]
    L.lio_curvature.argtypes = [i32, vp, sz, vp, vp, vp]
    L.lio_range_image_default_config.argtypes = [C.POINTER(RangeImageConfig)]
    L.lio_range_image_default_config.restype = None
    L.lio_range_image.argtypes = [C.POINTER(RangeImageConfig), vp, sz, sz, f64, dp, dp, dp, dp, i32, vp, sz, C.POINTER(sz),
                                  vp, vp, vp, vp]
    L.lio_feature_default_config.argtypes = [C.POINTER(FeatureConfig)]
    L.lio_feature_default_config.restype = None
    L.lio_extract_features.argtypes = [C.POINTER(FeatureConfig), vp, sz, sz, vp, vp, vp, vp, vp, C.POINTER(sz), vp,
                                       C.POINTER(sz), sz, vp, vp, vp]
    L.lio_s2m_debug_stamps.argtypes = [vp, vp, sz]
    L.lio_voxel_grid.argtypes = [i32, vp, sz, sz, f32, vp, sz, C.POINTER(sz)]
    L.lio_kf_store_create.argtypes = [i32, C.POINTER(vp)]
    L.lio_kf_store_destroy.argtypes = [vp]
    L.lio_kf_store_destroy.restype = None
    L.lio_kf_store_add.argtypes = [vp, vp, sz, sz, C.POINTER(i32)]
    L.lio_kf_store_count.argtypes = [vp]
    L.lio_kf_store_points.argtypes = [vp, C.c_int32]
    L.lio_kf_store_points.restype = C.c_size_t
    L.lio_kf_store_add_device.argtypes = [vp, vp, sz, sz, C.POINTER(i32)]
    L.lio_kf_store_add_from_handle.argtypes = [vp, vp, i32, C.POINTER(i32)]
    L.lio_assemble_map_resident.argtypes = [vp, vp, i32, C.POINTER(i32), C.POINTER(f32), f32, vp, sz, C.POINTER(sz)]
    L.lio_assemble_map.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(sz), sz, C.POINTER(f32), f32, vp, sz,
                                   C.POINTER(sz)]
    _LIB = L
    return L


def _check(rc, what):
    if rc < 0:
        msg = load_library().lio_last_error()
        raise LioError(f"{what} failed: {STATUS_NAMES.get(rc, rc)}: {msg.decode() if msg else ''}")
    return rc


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _as_points(a):
    """Accepts [n,3] float32 (stride 12) or any C-contiguous [n,k>=3] float32 (stride 4k)."""
    a = np.ascontiguousarray(a, np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("points must be [n, >=3] float32")
    return a, a.shape[1] * 4


class ScanToMap:
    """Device-resident replacement of mapOptimization's scan-to-map block (MO:1839-1865)."""

    def __init__(self, **cfg_overrides):
        self.lib = load_library()
        self.cfg = S2MConfig()
        self.lib.lio_s2m_default_config(C.byref(self.cfg))
        for k, v in cfg_overrides.items():
            if not hasattr(self.cfg, k):
                raise AttributeError(k)
            if k == "device_ids":
                for i, d in enumerate(v):
                    self.cfg.device_ids[i] = int(d)
                continue
            setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        _check(self.lib.lio_s2m_create(C.byref(self.cfg), C.byref(self.h)), "lio_s2m_create")
        self._n_scans = 0
        self._npts = []

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.lio_s2m_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # kdtreeSurfFromMap->setInputCloud(laserCloudSurfFromMapDS), MO:1846
    def set_map(self, map_pts):
        a, stride = _as_points(map_pts)
        _check(self.lib.lio_s2m_set_map(self.h, a.ctypes.data, len(a), stride), "lio_s2m_set_map")

    # scan2MapOptimization, MO:1839-1865 (loop MO:1848-1859)
    def scan2MapOptimization(self, scan_pts, pose):
        a, stride = _as_points(scan_pts)
        p = np.array(pose, np.float32).copy()
        res = S2MResult()
        rc = _check(self.lib.lio_s2m_register(self.h, a.ctypes.data, len(a), stride, _f32p(p), C.byref(res)),
                    "lio_s2m_register")
        self._n_scans, self._npts = 1, [len(a)]
        return p, res, rc

    # pcl::fromROSMsg(msgIn->cloud_deskewed, ...) MO:440 + scan2MapOptimization on the PointCloud2 blob itself
    def scan2MapOptimizationPC2(self, blob, n_points, layout, pose):
        b = np.ascontiguousarray(blob).view(np.uint8).reshape(-1)
        p = np.array(pose, np.float32).copy()
        res = S2MResult()
        rc = _check(self.lib.lio_s2m_register_pc2(self.h, b.ctypes.data, n_points, C.byref(layout), _f32p(p), C.byref(res)),
                    "lio_s2m_register_pc2")
        self._n_scans, self._npts = 1, [n_points]
        return p, res, rc

    # downsampleCurrentScan MO:1605-1611 + scan2MapOptimization MO:1839-1865 on the blob of cloud_info.cloud_deskewed,
    # one H2D copy, no host round trip between the voxel filter and the registration
    def downsampleAndScan2MapOptimization(self, blob, n_points, layout, leaf, pose, want_ds=False, device_ptr=None):
        p = np.array(pose, np.float32).copy()
        res = S2MResult()
        if device_ptr is None:
            b = np.ascontiguousarray(blob).view(np.uint8).reshape(-1)
            ptr = b.ctypes.data
        else:
            ptr = int(device_ptr)
        out = np.zeros((n_points, 8), np.float32) if want_ds else None
        n_ds = C.c_size_t(0)
        rc = _check(self.lib.lio_s2m_register_raw(self.h, ptr, n_points, C.byref(layout), float(leaf), _f32p(p), C.byref(res),
                                                  out.ctypes.data if want_ds else None, 32, C.byref(n_ds)), "lio_s2m_register_raw")
        self._n_scans, self._npts = 1, [int(n_ds.value)]
        if want_ds:
            o = out[:n_ds.value]
            return p, res, rc, np.concatenate([o[:, :3], o[:, 4:5]], 1)
        return p, res, rc, int(n_ds.value)

    # batched form
    def batch_upload(self, scans):
        arrs = [_as_points(s) for s in scans]
        strides = {s for _, s in arrs}
        if len(strides) != 1:
            raise ValueError("all scans of a batch must share one stride")
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a, _ in arrs])
        npts = (C.c_size_t * n)(*[len(a) for a, _ in arrs])
        _check(self.lib.lio_s2m_batch_upload(self.h, n, ptrs, npts, strides.pop()), "lio_s2m_batch_upload")
        self._n_scans, self._npts = n, [len(a) for a, _ in arrs]

    def share_map(self, owner):
        """This handle searches `owner`'s resident map (double-buffered streaming, see include/liogpu.h)."""
        _check(self.lib.lio_s2m_share_map(self.h, owner.h if owner is not None else None), "lio_s2m_share_map")
        self._map_owner = owner          # keep it alive

    def batch_upload_raw(self, base_ptr, n_pts, stride, asynchronous=True):
        """Scans laid out back to back at `base_ptr` (pinned host or device memory): n_pts[s] records of `stride` bytes."""
        n = len(n_pts)
        offs = np.concatenate([[0], np.cumsum(np.asarray(n_pts, np.int64) * stride)])
        ptrs = (C.c_void_p * n)(*[int(base_ptr) + int(o) for o in offs[:-1]])
        npts = (C.c_size_t * n)(*[int(v) for v in n_pts])
        fn = self.lib.lio_s2m_batch_upload_async if asynchronous else self.lib.lio_s2m_batch_upload
        _check(fn(self.h, n, ptrs, npts, stride), "lio_s2m_batch_upload(_async)")
        self._n_scans, self._npts = n, [int(v) for v in n_pts]

    def batch_set_poses(self, poses):
        p = np.ascontiguousarray(poses, np.float32).reshape(self._n_scans, 6)
        _check(self.lib.lio_s2m_batch_set_poses(self.h, _f32p(p)), "lio_s2m_batch_set_poses")

    def batch_run(self):
        _check(self.lib.lio_s2m_batch_run(self.h), "lio_s2m_batch_run")

    def batch_sync(self):
        _check(self.lib.lio_s2m_batch_sync(self.h), "lio_s2m_batch_sync")

    def batch_results(self, with_results=True):
        poses = np.zeros((self._n_scans, 6), np.float32)
        res = (S2MResult * self._n_scans)() if with_results else None
        _check(self.lib.lio_s2m_batch_results(self.h, _f32p(poses), res), "lio_s2m_batch_results")
        return poses, res

    def set_degeneracy(self, scan, matP, is_degenerate):
        m = np.ascontiguousarray(matP, np.float32).reshape(36)
        _check(self.lib.lio_s2m_set_degeneracy(self.h, scan, _f32p(m), int(is_degenerate)), "lio_s2m_set_degeneracy")

    def get_correspondences(self, scan=0):
        n = self._npts[scan]
        flag = np.zeros(n, np.uint8)
        coeff = np.zeros((n, 4), np.float32)
        nn = np.full((n, 5), -1, np.int32)
        _check(self.lib.lio_s2m_get_correspondences(self.h, scan, flag.ctypes.data, coeff.ctypes.data,
                                                    nn.ctypes.data), "lio_s2m_get_correspondences")
        return flag, coeff, nn

    # ---- extension beyond this reference: point-to-line residuals (upstream LIO-SAM cornerOptimization) ----
    def set_corner_map(self, map_pts):
        a, stride = _as_points(np.asarray(map_pts, np.float32).reshape(-1, 3) if len(map_pts) == 0 else map_pts)
        _check(self.lib.lio_s2m_set_corner_map(self.h, a.ctypes.data, len(a), stride), "lio_s2m_set_corner_map")

    def batch_upload_corners(self, scans):
        arrs = [_as_points(np.zeros((0, 3), np.float32) if len(s) == 0 else s) for s in scans]
        n = len(arrs)
        stride = {s for a, s in arrs if len(a)} or {12}
        if len(stride) != 1:
            raise ValueError("all scans of a batch must share one stride")
        ptrs = (C.c_void_p * n)(*[a.ctypes.data if len(a) else None for a, _ in arrs])
        npts = (C.c_size_t * n)(*[len(a) for a, _ in arrs])
        _check(self.lib.lio_s2m_batch_upload_corners(self.h, n, ptrs, npts, stride.pop()), "lio_s2m_batch_upload_corners")
        self._ncorner = [len(a) for a, _ in arrs]

    def scan2MapOptimizationCS(self, corner_pts, surf_pts, pose):
        cpts, cstride = _as_points(corner_pts)
        a, stride = _as_points(surf_pts)
        if cstride != stride:
            raise ValueError("corner and surf clouds must share one stride")
        p = np.array(pose, np.float32).copy()
        res = S2MResult()
        rc = _check(self.lib.lio_s2m_register_cs(self.h, cpts.ctypes.data, len(cpts), a.ctypes.data, len(a), stride,
                                                 _f32p(p), C.byref(res)), "lio_s2m_register_cs")
        self._n_scans, self._npts, self._ncorner = 1, [len(a)], [len(cpts)]
        return p, res, rc

    def get_corner_correspondences(self, scan=0):
        n = self._ncorner[scan]
        flag = np.zeros(n, np.uint8)
        coeff = np.zeros((n, 4), np.float32)
        nn = np.full((n, 5), -1, np.int32)
        _check(self.lib.lio_s2m_get_corner_correspondences(self.h, scan, flag.ctypes.data, coeff.ctypes.data,
                                                           nn.ctypes.data), "lio_s2m_get_corner_correspondences")
        return flag, coeff, nn

    def profile(self):
        p = S2MProfile()
        _check(self.lib.lio_s2m_get_profile(self.h, C.byref(p)), "lio_s2m_get_profile")
        return p

    def debug_persist_spin(self, spin_max=0, withhold_wg=-1):
        """Test hook of the one-launch loop: poll bound and a workgroup that never arrives (include/liogpu.h)."""
        _check(self.lib.lio_s2m_debug_persist_spin(self.h, int(spin_max), int(withhold_wg)), "lio_s2m_debug_persist_spin")

    def debug_stamps(self):
        nb = self.lib.lio_s2m_debug_stamps(self.h, None, 0)
        out = np.zeros((max(nb, 0), 4, 8), np.int64)
        if nb > 0:
            self.lib.lio_s2m_debug_stamps(self.h, out.ctypes.data, out.size)
        return out

    # multi-GPU hooks
    def set_stream(self, hip_stream):
        _check(self.lib.lio_s2m_set_stream(self.h, C.c_void_p(hip_stream)), "lio_s2m_set_stream")

    def set_global_grid(self, origin, dims):
        o = (C.c_float * 3)(*origin)
        d = (C.c_int32 * 3)(*dims)
        _check(self.lib.lio_s2m_set_global_grid(self.h, o, d), "lio_s2m_set_global_grid")

    def set_shard(self, axis, lo, hi):
        _check(self.lib.lio_s2m_set_shard(self.h, axis, lo, hi), "lio_s2m_set_shard")

    def set_shard_plan(self, axis, rank, bounds, halo_cells):
        b = (C.c_int32 * len(bounds))(*[int(v) for v in bounds])
        _check(self.lib.lio_s2m_set_shard_plan(self.h, axis, len(bounds) - 1, rank, b, halo_cells), "lio_s2m_set_shard_plan")

    def set_scan_shard(self, rank, world):
        _check(self.lib.lio_s2m_set_scan_shard(self.h, rank, world), "lio_s2m_set_scan_shard")

    def batch_begin(self):
        _check(self.lib.lio_s2m_batch_begin(self.h), "lio_s2m_batch_begin")

    def batch_iter_partial(self, d_sums_ptr):
        _check(self.lib.lio_s2m_batch_iter_partial(self.h, C.c_void_p(d_sums_ptr)), "lio_s2m_batch_iter_partial")

    def batch_iter_apply(self, d_sums_ptr):
        _check(self.lib.lio_s2m_batch_iter_apply(self.h, C.c_void_p(d_sums_ptr)), "lio_s2m_batch_iter_apply")

    def batch_poll_active(self, iteration):
        v = C.c_int32()
        _check(self.lib.lio_s2m_batch_poll_active(self.h, iteration, C.byref(v)), "lio_s2m_batch_poll_active")
        return v.value

    def batch_n_active(self):
        v = C.c_int32()
        _check(self.lib.lio_s2m_batch_n_active(self.h, C.byref(v)), "lio_s2m_batch_n_active")
        return v.value


class DeviceBuffer:
    """hipMalloc'ed bytes on `device_id` holding a copy of a numpy array (a cloud that already lives in HBM)."""

    def __init__(self, array, device_id=0):
        self.lib = load_library()
        self.device_id = int(device_id)
        a = np.ascontiguousarray(array)
        self.nbytes = a.nbytes
        self.ptr = self.lib.lio_device_alloc(self.device_id, self.nbytes)
        if not self.ptr:
            raise LioError("lio_device_alloc failed")
        _check(self.lib.lio_device_upload(self.device_id, self.ptr, a.ctypes.data, self.nbytes), "lio_device_upload")

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.lio_device_free(self.device_id, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedBuffer:
    """hipHostMalloc'ed bytes as a numpy array (true-DMA source for batch_upload_raw)."""

    def __init__(self, nbytes):
        self.lib = load_library()
        self.nbytes = int(nbytes)
        self.ptr = self.lib.lio_host_alloc(self.nbytes)
        if not self.ptr:
            raise LioError("lio_host_alloc failed")
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr))

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.lib.lio_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# transformUpdate, MO:1867-1907
def transform_update(pose, imu_available=0, imu_type=0, imu_roll_init=0.0, imu_pitch_init=0.0,
                     imu_rpy_weight=0.01, rotation_tollerance=1000.0, z_tollerance=1000.0):
    p = np.array(pose, np.float32).copy()
    load_library().lio_transform_update(_f32p(p), imu_available, imu_type, imu_roll_init, imu_pitch_init,
                                        imu_rpy_weight, rotation_tollerance, z_tollerance)
    return p


# imuDeskewInfo, IP:359-418
def imu_deskew_info(stamp, gyro, t_cur, t_end):
    stamp = np.ascontiguousarray(stamp, np.float64)
    g = np.ascontiguousarray(gyro, np.float64)
    gx, gy, gz = (np.ascontiguousarray(g[:, k]) for k in range(3))
    T, RX, RY, RZ = (np.zeros(2000) for _ in range(4))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    cur = load_library().lio_imu_deskew_info(dp(stamp), dp(gx), dp(gy), dp(gz), len(stamp), t_cur, t_end,
                                             dp(T), dp(RX), dp(RY), dp(RZ))
    return cur, T, RX, RY, RZ


_XYZIRT = np.dtype({"names": ["x", "y", "z", "intensity", "ring", "time"],
                    "formats": ["<f4", "<f4", "<f4", "<f4", "<u2", "<f4"],
                    "offsets": [0, 4, 8, 16, 20, 24], "itemsize": 32})


def pack_xyzirt(xyz, intensity, ring, time):
    """VelodynePointXYZIRT records (IP:4-15): 32-byte stride."""
    rec = np.zeros(len(xyz), _XYZIRT)
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["intensity"], rec["ring"], rec["time"] = intensity, ring, time
    return rec


# projectPointCloud + deskewPoint, IP:545-615
_SCRATCH = {}


def _scratch(name, shape, dtype=np.float32):
    """Re-used host output buffer (like a node's member clouds).  A fresh np.zeros per call hands the HIP
    runtime never-touched pages to pin for the D2H copy, which costs milliseconds and says nothing about
    the library; results are copied out of the scratch before they are returned."""
    need = int(np.prod(shape))
    buf = _SCRATCH.get((name, np.dtype(dtype).str))
    if buf is None or buf.size < need:
        buf = np.zeros(max(need, 1), dtype)
        _SCRATCH[(name, np.dtype(dtype).str)] = buf
    return buf[:need].reshape(shape)


def deskew(dcfg, records, t_cur, imu):
    cur, T, RX, RY, RZ = imu
    L = load_library()
    n = len(records)
    out = _scratch("deskew", (n, 8))       # pcl::PointXYZI, 32-byte stride
    n_out = C.c_size_t()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rec = np.ascontiguousarray(records)
    _check(L.lio_deskew(C.byref(dcfg), rec.ctypes.data, n, rec.dtype.itemsize, t_cur,
                        dp(T), dp(RX), dp(RY), dp(RZ), cur, out.ctypes.data, 32, C.byref(n_out)), "lio_deskew")
    o = out[:n_out.value]
    return np.concatenate([o[:, :3], o[:, 4:5]], axis=1).copy()


def deskew_pc2(dcfg, blob, n_points, layout, t_cur, imu):
    """projectPointCloud on the raw PointCloud2 `data` blob (any of the four sensor layouts of IP:226-285)."""
    cur, T, RX, RY, RZ = imu
    L = load_library()
    b = np.ascontiguousarray(blob).view(np.uint8).reshape(-1)
    out = _scratch("deskew", (max(n_points, 1), 8))
    n_out = C.c_size_t()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    _check(L.lio_deskew_pc2(C.byref(dcfg), b.ctypes.data, n_points, C.byref(layout), t_cur,
                            dp(T), dp(RX), dp(RY), dp(RZ), cur, out.ctypes.data, 32, C.byref(n_out)), "lio_deskew_pc2")
    o = out[:n_out.value]
    return np.concatenate([o[:, :3], o[:, 4:5]], axis=1).copy()


def deskew_default_config(**kw):
    c = DeskewConfig()
    load_library().lio_deskew_default_config(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


# calculateSmoothness, FE:81-101
def curvature(rng, device_id=0):
    r = np.ascontiguousarray(rng, np.float32)
    curv = np.zeros(len(r), np.float32)
    picked = np.full(len(r), -1, np.int32)
    label = np.full(len(r), -1, np.int32)
    _check(load_library().lio_curvature(device_id, r.ctypes.data, len(r), curv.ctypes.data,
                                        picked.ctypes.data, label.ctypes.data), "lio_curvature")
    return curv, picked, label


# extension (row A4): projectPointCloud + cloudExtraction of upstream LIO-SAM -> the cloud_info arrays FE consumes
def range_image(records, t_cur, imu, device_id=0, **cfg_overrides):
    """PointXYZIRT records -> dict(cloud [n,4], start_ring, end_ring, col, range) like synth.organize_scan."""
    L = load_library()
    cfg = RangeImageConfig()
    L.lio_range_image_default_config(C.byref(cfg))
    cfg.device_id = device_id
    for k, v in cfg_overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    cur, T, RX, RY, RZ = imu
    rec = np.ascontiguousarray(records)
    cells = int(cfg.N_SCAN) * int(cfg.Horizon_SCAN)
    out = _scratch("ri_cloud", (max(cells, 1), 8))
    col = _scratch("ri_col", (max(cells, 1),), np.int32)
    rng = _scratch("ri_range", (max(cells, 1),))
    start = np.zeros(cfg.N_SCAN, np.int32)
    end = np.zeros(cfg.N_SCAN, np.int32)
    n_out = C.c_size_t()
    dpp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    _check(L.lio_range_image(C.byref(cfg), rec.ctypes.data, len(rec), rec.dtype.itemsize, t_cur, dpp(T), dpp(RX), dpp(RY), dpp(RZ),
                             cur, out.ctypes.data, 32, C.byref(n_out), start.ctypes.data, end.ctypes.data, col.ctypes.data,
                             rng.ctypes.data), "lio_range_image")
    n = n_out.value
    return {"cloud": _from_records(out, n), "start_ring": start, "end_ring": end, "col": col[:n].copy(), "range": rng[:n].copy()}


# markOccludedPoints + extractFeatures, FE:103-238 (with calculateSmoothness: the whole handler FE:67-77)
def extract_features(cloud_xyzi, start_ring, end_ring, point_col, point_range, device_id=0, **cfg_overrides):
    """cloud_xyzi [n,4] -> dict(corner [n_c,4], surface [n_s,4], curvature, picked, label)."""
    L = load_library()
    cfg = FeatureConfig()
    L.lio_feature_default_config(C.byref(cfg))
    cfg.N_SCAN = len(start_ring)
    cfg.device_id = device_id
    for k, v in cfg_overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    rec = _as_xyzi_records(np.asarray(cloud_xyzi, np.float32).reshape(-1, 4))
    n = len(rec)
    sr = np.ascontiguousarray(start_ring, np.int32)
    er = np.ascontiguousarray(end_ring, np.int32)
    col = np.ascontiguousarray(point_col, np.int32)
    rng = np.ascontiguousarray(point_range, np.float32)
    corner = _scratch("fe_corner", (120 * max(cfg.N_SCAN, 1), 8))
    surf = _scratch("fe_surf", (max(n, 1), 8))
    curv = np.zeros(max(n, 1), np.float32)
    picked = np.zeros(max(n, 1), np.int32)
    label = np.zeros(max(n, 1), np.int32)
    nc, ns = C.c_size_t(0), C.c_size_t(0)
    _check(L.lio_extract_features(C.byref(cfg), rec.ctypes.data, n, 32, sr.ctypes.data, er.ctypes.data,
                                  col.ctypes.data, rng.ctypes.data, corner.ctypes.data, C.byref(nc),
                                  surf.ctypes.data, C.byref(ns), 32, curv.ctypes.data, picked.ctypes.data,
                                  label.ctypes.data), "lio_extract_features")
    return {"corner": _from_records(corner, nc.value), "surface": _from_records(surf, ns.value),
            "curvature": curv[:n], "picked": picked[:n], "label": label[:n]}


def _as_xyzi_records(a):
    """[n,4] (x,y,z,intensity) float32 -> pcl::PointXYZI records [n,8] (32-byte stride)."""
    a = np.ascontiguousarray(a, np.float32)
    rec = np.zeros((len(a), 8), np.float32)
    rec[:, :3] = a[:, :3]
    rec[:, 3] = 1.0
    rec[:, 4] = a[:, 3]
    return rec


def _from_records(rec, n):
    return np.concatenate([rec[:n, :3], rec[:n, 4:5]], axis=1).copy()


# downsampleCurrentScan, MO:1605-1611 (pcl::VoxelGrid)
def voxel_grid(xyzi, leaf, device_id=0):
    rec = _as_xyzi_records(xyzi)
    out = _scratch("voxel", rec.shape)
    n_out = C.c_size_t()
    rc = _check(load_library().lio_voxel_grid(device_id, rec.ctypes.data, len(rec), 32, leaf, out.ctypes.data, 32,
                                              C.byref(n_out)), "lio_voxel_grid")
    return _from_records(out, n_out.value), rc


# extractCloud, MO:1556-1588
def assemble_map(clouds_xyzi, poses, leaf, s2m=None, device_id=0, want_output=True):
    recs = [_as_xyzi_records(c) for c in clouds_xyzi]
    n = len(recs)
    ptrs = (C.c_void_p * max(n, 1))(*[r.ctypes.data for r in recs])
    npts = (C.c_size_t * max(n, 1))(*[len(r) for r in recs])
    p = np.ascontiguousarray(poses, np.float32).reshape(n, 6)
    total = sum(len(r) for r in recs)
    out = _scratch("assemble", (max(total, 1), 8)) if want_output else None
    n_out = C.c_size_t()
    rc = _check(load_library().lio_assemble_map(s2m.h if s2m is not None else None, device_id, n, ptrs, npts, 32,
                                                _f32p(p), leaf, out.ctypes.data if want_output else None, 32,
                                                C.byref(n_out)), "lio_assemble_map")
    return (_from_records(out, n_out.value) if want_output else None), n_out.value, rc


class KeyframeStore:
    """surfCloudKeyFrames (MO:128) kept in HBM."""

    def __init__(self, device_id=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        _check(self.lib.lio_kf_store_create(device_id, C.byref(self.h)), "lio_kf_store_create")

    def add(self, cloud_xyzi):                      # surfCloudKeyFrames.push_back, MO:2141
        rec = _as_xyzi_records(cloud_xyzi)
        kid = C.c_int32()
        _check(self.lib.lio_kf_store_add(self.h, rec.ctypes.data, len(rec), 32, C.byref(kid)), "lio_kf_store_add")
        return kid.value

    def add_from_handle(self, s2m, scan=0):         # MO:2136-2142 without leaving the device
        kid = C.c_int32()
        _check(self.lib.lio_kf_store_add_from_handle(self.h, s2m.h, scan, C.byref(kid)), "lio_kf_store_add_from_handle")
        return kid.value

    def add_device(self, dev_ptr, n, stride):
        kid = C.c_int32()
        _check(self.lib.lio_kf_store_add_device(self.h, C.c_void_p(dev_ptr), n, stride, C.byref(kid)), "lio_kf_store_add_device")
        return kid.value

    def __len__(self):
        return self.lib.lio_kf_store_count(self.h)

    def assemble(self, ids, poses, leaf, s2m=None, want_output=True, max_out=None):   # extractCloud, MO:1556-1588
        ids_a = np.ascontiguousarray(ids, np.int32)
        p = np.ascontiguousarray(poses, np.float32).reshape(len(ids_a), 6)
        if want_output and max_out is None:          # room for every selected point: the filter may pass its input through
            max_out = sum(int(self.lib.lio_kf_store_points(self.h, int(i))) for i in ids_a)
        out = np.zeros((max(max_out or 1, 1), 8), np.float32) if want_output else None
        n_out = C.c_size_t()
        rc = _check(self.lib.lio_assemble_map_resident(
            s2m.h if s2m is not None else None, self.h, len(ids_a), ids_a.ctypes.data_as(C.POINTER(C.c_int32)), _f32p(p),
            leaf, out.ctypes.data if want_output else None, 32, C.byref(n_out)), "lio_assemble_map_resident")
        return (_from_records(out, n_out.value) if want_output else None), n_out.value, rc

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.lio_kf_store_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
