// liogpu.hpp -- header-only C++ mirror of the reference's member functions over
// the C ABI in liogpu.h.  Names follow class mapOptimization (MO =
// src/liorf/src/mapOptmization.cpp) so that the patched node reads like the
// original: scan2MapOptimization() MO:1839, transformUpdate() MO:1867.
// No PCL/ROS dependency: clouds are passed as (pointer, count, stride).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "liogpu.h"

namespace liogpu {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

class ScanToMap {
public:
    // transformTobeMapped [roll,pitch,yaw,x,y,z] (MO:171) and isDegenerate (MO:176) live here
    // exactly like the members they replace.
    float transformTobeMapped[6] = {0, 0, 0, 0, 0, 0};
    bool isDegenerate = false;
    lio_s2m_result last{};

    explicit ScanToMap(const lio_s2m_config* cfg = nullptr)
    {
        lio_s2m_config c;
        if (cfg) c = *cfg; else lio_s2m_default_config(&c);
        check(lio_s2m_create(&c, &h_), "lio_s2m_create");
    }
    ~ScanToMap() { lio_s2m_destroy(h_); }
    ScanToMap(const ScanToMap&) = delete;
    ScanToMap& operator=(const ScanToMap&) = delete;

    // kdtreeSurfFromMap->setInputCloud(laserCloudSurfFromMapDS), MO:1846
    void setInputCloud(const void* pts, size_t n, size_t stride_bytes)
    {
        check(lio_s2m_set_map(h_, pts, n, stride_bytes), "lio_s2m_set_map");
    }

    // The loop MO:1848-1859.  Returns the status: LIO_OK, LIO_TOO_FEW_POINTS (MO:1862-1864: the
    // caller prints the ROS_WARN) or LIO_TOO_FEW_CORR.  The caller keeps the guards MO:1841-1844
    // and calls transformUpdate() afterwards (MO:1861).
    int scan2MapOptimization(const void* laserCloudSurfLastDS, size_t n, size_t stride_bytes)
    {
        const int rc = lio_s2m_register(h_, laserCloudSurfLastDS, n, stride_bytes, transformTobeMapped, &last);
        if (rc < 0) check(rc, "lio_s2m_register");
        isDegenerate = last.is_degenerate != 0;
        return rc;
    }

    // The same on the `data` blob of msgIn->cloud_deskewed (sensor_msgs/PointCloud2, cloud_info.msg:27): what
    // pcl::fromROSMsg(msgIn->cloud_deskewed, *laserCloudSurfLast) MO:440 + downsample + the loop do, minus the copy into
    // a pcl::PointCloud.  point_step / off_x come from msg.point_step / the "x" entry of msg.fields.
    int scan2MapOptimizationPC2(const void* data, size_t width_times_height, uint32_t point_step, uint32_t off_x, bool pin_host = false)
    {
        lio_pc2_layout lay{};
        lay.point_step = point_step; lay.off_x = off_x; lay.off_intensity = -1; lay.off_ring = -1; lay.off_time = -1;
        lay.pin_host = pin_host ? 1 : 0;
        const int rc = lio_s2m_register_pc2(h_, data, width_times_height, &lay, transformTobeMapped, &last);
        if (rc < 0) check(rc, "lio_s2m_register_pc2");
        isDegenerate = last.is_degenerate != 0;
        return rc;
    }

    // downsampleCurrentScan() MO:1605-1611 + the loop MO:1848-1859 in one device chain, from the blob of
    // msgIn->cloud_deskewed (MO:440): `downSizeFilterSurf.filter(*laserCloudSurfLastDS)` runs on the GPU and the registration
    // starts from its output without a host round trip.  laserCloudSurfLastDSNum = n_downsampled afterwards (MO:1610); pass
    // laserCloudSurfLastDS->points.data() as ds_out (room for width*height PointType records) if the host copy is needed.
    int downsampleAndScan2MapOptimization(const void* data, size_t width_times_height, uint32_t point_step, uint32_t off_x,
                                          int32_t off_intensity, float mappingSurfLeafSize, void* ds_out = nullptr,
                                          size_t ds_stride = 32, bool pin_host = false)
    {
        lio_pc2_layout lay{};
        lay.point_step = point_step; lay.off_x = off_x; lay.off_intensity = off_intensity; lay.off_ring = -1; lay.off_time = -1;
        lay.pin_host = pin_host ? 1 : 0;
        const int rc = lio_s2m_register_raw(h_, data, width_times_height, &lay, mappingSurfLeafSize, transformTobeMapped, &last,
                                            ds_out, ds_stride, &n_downsampled);
        if (rc < 0) check(rc, "lio_s2m_register_raw");
        isDegenerate = last.is_degenerate != 0;
        return rc;
    }
    size_t n_downsampled = 0;

    // ---- extension beyond this reference (upstream LIO-SAM) ----
    // kdtreeCornerFromMap->setInputCloud(laserCloudCornerFromMapDS)
    void setInputCloudCorner(const void* pts, size_t n, size_t stride_bytes)
    {
        check(lio_s2m_set_corner_map(h_, pts, n, stride_bytes), "lio_s2m_set_corner_map");
    }
    // upstream loop body: cornerOptimization(); surfOptimization(); combineOptimizationCoeffs(); LMOptimization(iterCount)
    int scan2MapOptimization(const void* laserCloudCornerLastDS, size_t n_corner,
                             const void* laserCloudSurfLastDS, size_t n_surf, size_t stride_bytes)
    {
        const int rc = lio_s2m_register_cs(h_, laserCloudCornerLastDS, n_corner, laserCloudSurfLastDS, n_surf,
                                           stride_bytes, transformTobeMapped, &last);
        if (rc < 0) check(rc, "lio_s2m_register_cs");
        isDegenerate = last.is_degenerate != 0;
        return rc;
    }

    // transformUpdate + constraintTransformation, MO:1867-1907
    void transformUpdate(bool imuAvailable, int imuType, float imuRollInit, float imuPitchInit,
                         float imuRPYWeight, float rotation_tollerance, float z_tollerance)
    {
        lio_transform_update(transformTobeMapped, imuAvailable ? 1 : 0, imuType, imuRollInit, imuPitchInit,
                             imuRPYWeight, rotation_tollerance, z_tollerance);
    }

    lio_s2m_handle* handle() { return h_; }

private:
    static void check(int rc, const char* what)
    {
        if (rc < 0) throw Error(rc, std::string(what) + ": " + lio_last_error());
    }
    lio_s2m_handle* h_ = nullptr;
};

}  // namespace liogpu
