"""MI355X-native scan-to-map registration path (C ABI: include/liogpu.h).

This package is the Python test/bench harness around lio-slam_amd/libliogpu.so
(hand-written HIP for gfx950).  The directory name is not a Python identifier;
import it with `importlib.import_module("lio-slam_amd")`.  There is no CPU
fallback: a missing library or GPU raises.
"""
from .api import (LioError, ScanToMap, S2MConfig, S2MResult, S2MProfile, DeskewConfig,
                  lib_path, load_library, build_library, deskew, curvature, imu_deskew_info,
                  transform_update, pack_xyzirt, deskew_default_config, voxel_grid, assemble_map, KeyframeStore, STATUS_NAMES,
                  extract_features, FeatureConfig, range_image, RangeImageConfig, PinnedBuffer, DeviceBuffer, PC2Layout, deskew_pc2)

__all__ = ["LioError", "ScanToMap", "S2MConfig", "S2MResult", "S2MProfile", "DeskewConfig",
           "lib_path", "load_library", "build_library", "deskew", "curvature", "imu_deskew_info",
           "transform_update", "pack_xyzirt", "deskew_default_config", "voxel_grid", "assemble_map", "KeyframeStore", "STATUS_NAMES",
           "extract_features", "FeatureConfig", "range_image", "RangeImageConfig", "PinnedBuffer", "DeviceBuffer", "PC2Layout", "deskew_pc2"]
