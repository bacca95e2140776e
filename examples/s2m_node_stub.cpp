// s2m_node_stub.cpp -- what the patched mapOptimization::scan2MapOptimization
// (MO:1839-1865) looks like on top of liogpu.hpp, with pcl::PointXYZI stood in
// by a 32-byte POD of the same layout (UT:65).  Builds with plain g++:
//   g++ -std=c++17 -Iinclude examples/s2m_node_stub.cpp -Llio-slam_amd -lliogpu \
//       -Wl,-rpath,$PWD/lio-slam_amd -o /tmp/s2m_node_stub
// Reads two raw float32 xyz files (scan, map) and a 6-float initial pose.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "liogpu.hpp"

struct alignas(16) PointXYZI { float x, y, z, pad; float intensity; float pad2[3]; };   // sizeof == 32
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI layout");

static std::vector<PointXYZI> load_xyz(const char* path)
{
    std::vector<PointXYZI> out;
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    float v[3];
    while (std::fread(v, sizeof(float), 3, f) == 3) out.push_back({ v[0], v[1], v[2], 1.0f, 0.0f, {0, 0, 0} });
    std::fclose(f);
    return out;
}

int main(int argc, char** argv)
{
    if (argc < 9) { std::fprintf(stderr, "usage: %s scan.f32 map.f32 roll pitch yaw x y z\n", argv[0]); return 2; }
    std::vector<PointXYZI> laserCloudSurfLastDS = load_xyz(argv[1]);
    std::vector<PointXYZI> laserCloudSurfFromMapDS = load_xyz(argv[2]);
    try {
        liogpu::ScanToMap s2m;                                   // constants default to the literals in MO
        for (int k = 0; k < 6; ++k) s2m.transformTobeMapped[k] = std::strtof(argv[3 + k], nullptr);

        // ---- body of scan2MapOptimization, MO:1839-1865 ----
        const int laserCloudSurfLastDSNum = (int)laserCloudSurfLastDS.size();
        if (laserCloudSurfLastDSNum > 30) {                      // MO:1844 (kept in the caller)
            s2m.setInputCloud(laserCloudSurfFromMapDS.data(), laserCloudSurfFromMapDS.size(), sizeof(PointXYZI));  // MO:1846
            s2m.scan2MapOptimization(laserCloudSurfLastDS.data(), laserCloudSurfLastDS.size(), sizeof(PointXYZI)); // MO:1848-1859
            s2m.transformUpdate(false, 0, 0.0f, 0.0f, 0.01f, 1000.0f, 1000.0f);                                    // MO:1861
        } else {
            std::fprintf(stderr, "Not enough features! Only %d planar features available.\n", laserCloudSurfLastDSNum);
        }
        std::printf("iters %d converged %d degenerate %d corr %d pose %.9g %.9g %.9g %.9g %.9g %.9g\n", s2m.last.iters,
                    s2m.last.converged, (int)s2m.isDegenerate, s2m.last.n_corr_last, s2m.transformTobeMapped[0],
                    s2m.transformTobeMapped[1], s2m.transformTobeMapped[2], s2m.transformTobeMapped[3],
                    s2m.transformTobeMapped[4], s2m.transformTobeMapped[5]);

        // ---- variant (a) of INTEGRATION.md: the same registration straight from the PointCloud2 blob of
        //      msgIn->cloud_deskewed (MO:440, cloud_info.msg:27): data pointer, width*height, point_step, offset of "x"
        liogpu::ScanToMap pc2;
        for (int k = 0; k < 6; ++k) pc2.transformTobeMapped[k] = std::strtof(argv[3 + k], nullptr);
        pc2.setInputCloud(laserCloudSurfFromMapDS.data(), laserCloudSurfFromMapDS.size(), sizeof(PointXYZI));
        pc2.scan2MapOptimizationPC2(laserCloudSurfLastDS.data(), laserCloudSurfLastDS.size(), (uint32_t)sizeof(PointXYZI), 0, true);
        pc2.transformUpdate(false, 0, 0.0f, 0.0f, 0.01f, 1000.0f, 1000.0f);
        std::printf("pc2 iters %d pose %.9g %.9g %.9g %.9g %.9g %.9g\n", pc2.last.iters, pc2.transformTobeMapped[0],
                    pc2.transformTobeMapped[1], pc2.transformTobeMapped[2], pc2.transformTobeMapped[3],
                    pc2.transformTobeMapped[4], pc2.transformTobeMapped[5]);

        // ---- variant (a'): downsampleCurrentScan MO:1605-1611 + the loop as one device chain, from the blob.  The scan given to
        //      this stub is already downsampled, so a leaf of half its spacing leaves it (nearly) as it is;
        //      what is shown is the call, the returned N_s and that the registration runs from the filter's output
        liogpu::ScanToMap raw;
        for (int k = 0; k < 6; ++k) raw.transformTobeMapped[k] = std::strtof(argv[3 + k], nullptr);
        raw.setInputCloud(laserCloudSurfFromMapDS.data(), laserCloudSurfFromMapDS.size(), sizeof(PointXYZI));
        std::vector<PointXYZI> ds(laserCloudSurfLastDS.size());
        raw.downsampleAndScan2MapOptimization(laserCloudSurfLastDS.data(), laserCloudSurfLastDS.size(), (uint32_t)sizeof(PointXYZI), 0, 16,
                                              0.2f, ds.data(), sizeof(PointXYZI));
        std::printf("raw iters %d n_ds %zu of %zu converged %d\n", raw.last.iters, raw.n_downsampled, laserCloudSurfLastDS.size(), raw.last.converged);

        // ---- variant (b): several GPUs behind the same calls (here the one device listed twice)
        lio_s2m_config cfg;
        lio_s2m_default_config(&cfg);
        cfg.n_devices = 2; cfg.device_ids[0] = 0; cfg.device_ids[1] = 0;
        liogpu::ScanToMap multi(&cfg);
        for (int k = 0; k < 6; ++k) multi.transformTobeMapped[k] = std::strtof(argv[3 + k], nullptr);
        multi.setInputCloud(laserCloudSurfFromMapDS.data(), laserCloudSurfFromMapDS.size(), sizeof(PointXYZI));
        multi.scan2MapOptimization(laserCloudSurfLastDS.data(), laserCloudSurfLastDS.size(), sizeof(PointXYZI));
        multi.transformUpdate(false, 0, 0.0f, 0.0f, 0.01f, 1000.0f, 1000.0f);
        std::printf("multi iters %d pose %.9g %.9g %.9g %.9g %.9g %.9g\n", multi.last.iters, multi.transformTobeMapped[0],
                    multi.transformTobeMapped[1], multi.transformTobeMapped[2], multi.transformTobeMapped[3],
                    multi.transformTobeMapped[4], multi.transformTobeMapped[5]);
    } catch (const liogpu::Error& e) {
        // hard errors (<0): the node falls back to its own CPU loop MO:1846-1859
        std::fprintf(stderr, "liogpu: %s (code %d)\n", e.what(), e.code);
        return 1;
    }
    return 0;
}
