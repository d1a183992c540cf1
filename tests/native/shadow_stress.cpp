// CPU check of the drop-in LocalMap's shadow grid (include/eskf_lio_shim/LocalMap.hpp) without a device: the C ABI is
// replaced by stubs that accept every call, so what runs is the HOST side of the class — the worker thread that keeps
// the raw points of every voxel while the grid proper would live on the device.  The same clouds go through a map with
// the defaults (deviceResident + keepRawPoints: insertion and eviction on the worker thread) and through a
// host-authoritative map (the reference's loops on the caller's thread): save() must write the same points.  Built with
// and without ThreadSanitizer by tests/test_capi_cpu.py.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "eskf_lio_shim/LocalMap.hpp"

// ---- the C ABI, stubbed: every call succeeds and does nothing ----
struct vgicp_ctx { int unused; };
static vgicp_ctx g_ctx;
extern "C" {
int vgicp_create(int, vgicp_ctx** out) { *out = &g_ctx; return VGICP_OK; }
int vgicp_create_multi(const int*, int, vgicp_ctx** out) { *out = &g_ctx; return VGICP_OK; }
int vgicp_destroy(vgicp_ctx*) { return VGICP_OK; }
const char* vgicp_last_error(const vgicp_ctx*) { return ""; }
int vgicp_map_reset(vgicp_ctx*, double, size_t) { return VGICP_OK; }
int vgicp_map_upsert(vgicp_ctx*, size_t, const int32_t*, const double*, const double*) { return VGICP_OK; }
int vgicp_map_erase(vgicp_ctx*, size_t, const int32_t*) { return VGICP_OK; }
int vgicp_map_size(const vgicp_ctx*, size_t* voxels, size_t* slots) { if (voxels) *voxels = 0; if (slots) *slots = 0; return VGICP_OK; }
int vgicp_map_export(vgicp_ctx*, size_t, int32_t*, double*, double*, uint64_t*, size_t* written) { if (written) *written = 0; return VGICP_OK; }
int vgicp_map_insert_scan(vgicp_ctx*, size_t, const double*, const double*, const double*, size_t, size_t* new_voxels) { if (new_voxels) *new_voxels = 0; return VGICP_OK; }
int vgicp_map_insert_resident_async(vgicp_ctx*, const double*, size_t) { return VGICP_OK; }
int vgicp_map_evict(vgicp_ctx*, const double*, double, size_t* removed) { if (removed) *removed = 0; return VGICP_OK; }
int vgicp_match(vgicp_ctx*, size_t, const double*, const double*, double*, double*, double*, double*, uint64_t*, size_t* matched) { if (matched) *matched = 0; return VGICP_OK; }
int vgicp_get_counter(const vgicp_ctx*, int, uint64_t* value) { if (value) *value = 0; return VGICP_OK; }
int vgicp_scan_download(vgicp_ctx*, size_t, double*, double*, size_t* n) { if (n) *n = 0; return VGICP_OK; }
}

static std::vector<std::string> sorted_points(const std::string& path) {
  std::ifstream f(path);
  std::vector<std::string> lines;
  std::string line;
  bool data = false;
  while (std::getline(f, line)) {
    if (data) lines.push_back(line);
    if (line.rfind("DATA", 0) == 0) data = true;
  }
  std::sort(lines.begin(), lines.end());
  return lines;
}

int main(int argc, char** argv) {
  using namespace ESKF_LIO;
  const int frames = argc > 1 ? std::atoi(argv[1]) : 120;
  const std::string dir = argc > 2 ? argv[2] : "/tmp";
  LocalMapConfig fast;             // the defaults: grid on the device, shadow grid for save()
  fast.removePeriod = 0.0;         // evict at every update that inserts (deterministic)
  fast.distanceThreshold = 12.0;
  fast.maxNumPointsPerVoxel = 6;
  LocalMapConfig plain = fast;
  plain.deviceResident = false;    // the host grid is authoritative: the reference's loops on this thread
  LocalMap a(fast, false, &g_ctx), b(plain, false, &g_ctx);
  for (int f = 0; f < frames; ++f) {
    Isometry3d T = Isometry3d::Identity();
    T.matrix()(0, 3) = 0.4 * f;                              // the platform drives away: old voxels are evicted
    T.matrix()(1, 3) = 0.1 * f;
    for (int which = 0; which < 2; ++which) {
      auto cloud = std::make_shared<PointCloud>();
      // round 6: the clouds' storage comes out of the pool that clouds dying inside the classes fill — here the shadow
      // grid's worker on another thread (what CloudPreprocessor::process does before it sizes a cloud)
      shim::adoptStorage(cloud->covariances_, shim::storagePool().covariances, 1500);
      shim::adoptStorage(cloud->points_, shim::storagePool().points, 1500);
      if (!cloud->covariances_.empty() || !cloud->points_.empty()) { std::printf("adopted storage is not empty\n"); return 1; }
      std::mt19937_64 gen(1000 + f);
      std::uniform_real_distribution<double> v(-8.0, 8.0);
      for (int i = 0; i < 1500; ++i) {
        const double x = v(gen), y = v(gen), z = 0.2 * v(gen);
        cloud->points_.push_back(Vector3d{{x, y, z}});
        Matrix3d C;
        std::memset(C.m, 0, sizeof C.m);
        C(0, 0) = 1.0 + 0.01 * i; C(1, 1) = 1.0; C(2, 2) = 1.0;
        cloud->covariances_.push_back(C);
      }
      (which == 0 ? a : b).updateLocalMap(std::move(cloud), T, f == 0);
    }
    if (f % 17 == 5) (void)a.grid().size();   // a reader in between: waits for the worker, sees a consistent grid
  }
  a.save(dir + "/shadow_a.pcd", dir + "/shadow_a.txt");
  b.save(dir + "/shadow_b.pcd", dir + "/shadow_b.txt");
  const auto la = sorted_points(dir + "/shadow_a.pcd"), lb = sorted_points(dir + "/shadow_b.pcd");
  if (la.empty() || la != lb) {
    std::printf("save() differs: %zu points with the shadow grid, %zu from the host-authoritative map\n", la.size(), lb.size());
    return 1;
  }
  if (!a.savesRawPoints()) { std::printf("the shadow grid lost a frame\n"); return 1; }
  std::printf("ok %d frames, %zu points saved\n", frames, la.size());
  return 0;
}
