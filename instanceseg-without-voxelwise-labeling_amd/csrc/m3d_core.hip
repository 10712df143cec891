// Version / error reporting for libm3d.so.
#include "m3d_common.h"

namespace m3d {
thread_local char g_last_hip_error[256] = "";
}

// Tuning options.  The RELEASE library (libm3d.so) has no process-wide state: opt() returns compile-time defaults and m3d_set_option
// refuses.  Only the tuning build (libm3d_tune.so = the same objects + this file with -DM3D_TUNING; A/B tools and the kernel-family
// tests load it beside the release library) keeps a mutable table.
namespace m3d {
[[maybe_unused]] static const int kOptDefaults[OPT_COUNT] = {1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0};
static const char* const kOptNames[OPT_COUNT] = {"xcd_map", "tune_k3", "tune_wino", "tune_wino2", "tune_wino2_xt", "tune_fc_slices", "tune_fc_slices_tail", "tune_fc_x3_rows", "tune_stem", "tune_fc_x_alias", "tune_roi_xcd"};
}  // namespace m3d
#ifdef M3D_TUNING
#include <atomic>
namespace m3d {
static std::atomic<int> g_opt[OPT_COUNT] = {{1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {0}};
int opt(Opt o) { return g_opt[o].load(std::memory_order_relaxed); }
}  // namespace m3d
#else
namespace m3d {
int opt(Opt o) { return kOptDefaults[o]; }
}  // namespace m3d
#endif

M3D_API int m3d_version(void) { return 200; }

M3D_API int m3d_set_option(const char* name, int value) {
  if (!name) return M3D_EINVAL;
  for (int i = 0; i < m3d::OPT_COUNT; ++i)
    if (!strcmp(name, m3d::kOptNames[i])) {
#ifdef M3D_TUNING
      m3d::g_opt[i].store(value, std::memory_order_relaxed);
      return M3D_OK;
#else
      (void)value;
      return M3D_EUNSUPPORTED;          // the release library keeps no mutable state (include/m3d.h)
#endif
    }
  return M3D_EINVAL;
}

M3D_API int m3d_get_option(const char* name, int* value) {
  if (!name || !value) return M3D_EINVAL;
  for (int i = 0; i < m3d::OPT_COUNT; ++i)
    if (!strcmp(name, m3d::kOptNames[i])) { *value = m3d::opt((m3d::Opt)i); return M3D_OK; }
  return M3D_EINVAL;
}

/* 1 in libm3d_tune.so, 0 in the release library */
M3D_API int m3d_tuning_build(void) {
#ifdef M3D_TUNING
  return 1;
#else
  return 0;
#endif
}

M3D_API const char* m3d_error_string(int code) {
  switch (code) {
    case M3D_OK: return "ok";
    case M3D_EINVAL: return "invalid argument";
    case M3D_ELAUNCH: return "HIP launch/runtime error";
    case M3D_EWORKSPACE: return "workspace too small";
    case M3D_EUNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}

M3D_API const char* m3d_last_hip_error(void) { return m3d::g_last_hip_error; }
