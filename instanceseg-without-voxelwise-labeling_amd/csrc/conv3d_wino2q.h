// Internal interface of the "quad" Winograd F(2x2,3x3) kernel (conv3d_wino2q.hip) used by conv3d_wino2.hip's entry points.
#pragma once
#include "m3d_common.h"

namespace m3d_w2q {

// element index inside one (channel pair, cout block) segment of the packed weights: [dz*4 + eta][lane64][xi] (conv3d_wino2.hip)
__device__ __host__ __forceinline__ constexpr int w2_slot(int dz, int eta, int xi, int lane) { return ((dz * 4 + eta) * 64 + lane) * 4 + xi; }

struct Epi {
  const float* scale;
  const float* shift;
  int relu;
  int xcd_map;
  // split-K over workgroups: blockIdx.z = slice; slice s handles 4-channel chunks [s*cps, (s+1)*cps) and writes its un-scaled
  // partial result to out + s*slice_stride (ksplit <= 1: no split)
  int ksplit, cps;
  size_t slice_stride;
  unsigned char* argmax;   // fused pool + arg-max: index 0..7 = (dz, dy, dx) of each pooled value's first maximum
#ifdef M3D_W2_STAMPS
  unsigned long long* stamps;
#endif
};

// xt = x pairs per wave block: 32 (tile 64 x 2 x 2 outputs), 16 (32 x 4 x 2), 8 (16 x 8 x 2; no fused pool)
int launch(int xt, bool pool, bool argmax, const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W,
           Epi ep, hipStream_t st);
void tile_dims(int xt, int* tx, int* ty, int* tz);

}  // namespace m3d_w2q
