// Version / error reporting for libm3d.so.
#include "m3d_common.h"

namespace m3d {
thread_local char g_last_hip_error[256] = "";
}

M3D_API int m3d_version(void) { return 100; }

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
