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
enum Opt { OPT_XCD_MAP = 0, OPT_TUNE_K3, OPT_TUNE_WINO, OPT_TUNE_WINO2, OPT_TUNE_WINO2_XT, OPT_TUNE_FC_SLICES, OPT_TUNE_FC_SLICES_TAIL, OPT_TUNE_FC_X3_ROWS, OPT_TUNE_STEM, OPT_TUNE_FC_X_ALIAS, OPT_TUNE_ROI_XCD, OPT_COUNT };
int opt(Opt o);
// sums[p] = sum of win[p, 0..w3) in a FIXED order (per-thread strided partial sums, then a fixed tree): the per-peak normaliser of the
// response maps must not depend on the order in which workgroups finish (float atomics gave maps that differed by one uint8 level from
// run to run)
// Strip layouts of the PRM window batches [C, n, n, L]: the P windows of a layer side by side along x, window p at columns
// lead + p * pitch .. + n - 1, every other column zero.
//   mode 1: pitch n + 1, lead 0, L = P (n + 1) - the exactly-local F(2x2,3x3) kernel only needs one zero column between windows;
//   mode 2: quad-aligned for the F(2x4,3x3) kernel, whose F(4,3) along x is local only in exact arithmetic: an output quad
//           [4t, 4t+4) reads columns 4t-1 .. 4t+4, so no quad that holds a window's outputs may touch another window's data:
//           pitch = the smallest multiple of 4 > max(n + r, 4 floor((r + n - 1) / 4) + 4 - r) over the start residue r = lead,
//           L = P pitch + roundup(lead, 4)  (n = 38: pitch 40, lead 1; n = 40: 44, 0; n = 16: 20, 0; n = 18: 20, 1).
__host__ __device__ inline void strip_geom(int n, int mode, int num_peaks, int* pitch, int* lead, long long* L) {
  if (mode != 2) { *pitch = n + 1; *lead = 0; *L = (long long)num_peaks * (n + 1); return; }
  int bp = 1 << 30, br = 0;
  for (int r = 0; r < 4; ++r) {
    const int a = n + r, b = 4 * ((r + n - 1) / 4) + 4 - r;
    const int pp = ((a > b ? a : b) + 1 + 3) / 4 * 4;
    if (pp < bp) { bp = pp; br = r; }
  }
  *pitch = bp; *lead = br; *L = (long long)num_peaks * bp + (br ? 4 : 0);
}
int window_sums(const float* d_win, long long w3, int num_peaks, float* d_sums, hipStream_t st);
// f16x2 split (fc_gemm.hip, prm_small_f16.hip): the power of two s with bound * s in [2^14, 2^15) - an operand whose largest magnitude is
// `bound` then fits fp16 with headroom - and its inverse, from the bound's exponent field
__device__ __forceinline__ void f16_scale_of(float bound, float& s, float& inv) {
  int f = 268 - (int)((__float_as_uint(bound) >> 23) & 255u);
  f = f < 2 ? 2 : (f > 252 ? 252 : f);
  s = __uint_as_float((unsigned)f << 23);
  inv = __uint_as_float((unsigned)(254 - f) << 23);
}
}  // namespace m3d
