// Shared helpers for libm3d.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/m3d.h"

#define M3D_API extern "C" __attribute__((visibility("default")))

namespace m3d {
extern thread_local char g_last_hip_error[256];
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s: %s", what, hipGetErrorString(e));
    return M3D_ELAUNCH;
  }
  return M3D_OK;
}
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr int kWave = 64;
// Explicit tuning options (m3d_set_option): the library never reads the environment.  -1 = "not set" for the tile overrides.
enum Opt { OPT_XCD_MAP = 0, OPT_TUNE_K3, OPT_TUNE_WINO, OPT_TUNE_WINO2, OPT_TUNE_WINO2_XT, OPT_TUNE_FC_SLICES, OPT_TUNE_FC_SLICES_TAIL, OPT_TUNE_FC_X3_ROWS, OPT_TUNE_STEM, OPT_COUNT };
int opt(Opt o);
// sums[p] = sum of win[p, 0..w3) in a FIXED order (per-thread strided partial sums, then a fixed tree): the per-peak normaliser of the
// response maps must not depend on the order in which workgroups finish (float atomics gave maps that differed by one uint8 level from
// run to run)
int window_sums(const float* d_win, long long w3, int num_peaks, float* d_sums, hipStream_t st);
}  // namespace m3d
