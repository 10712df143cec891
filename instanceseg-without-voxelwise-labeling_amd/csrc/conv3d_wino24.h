// Internal interface of the Winograd F(2x4, 3x3) kernel family (conv3d_wino24.hip) used by conv3d_wino2.hip's entry points.
#pragma once
#include "conv3d_wino2q.h"

namespace m3d_w24 {

// Epilogue of the PRM strip back-propagation fused with the NEXT layer's `prepare` step (lib/prm/peak_backprop_3d.py:8-34; what
// prm_prepare_kernel does as a pass of its own): the conv's input is the prepared gradient strip of layer L+1 ("A": windows U wide), its
// bare backward-data result never reaches memory - every element goes through PreHook (x (X - min X) of layer L+1's input), ReLU mask,
// eval-BatchNorm scale and PostHook (/ (|N| + 1e-10), 0 where N < 1e-10) of layer L and lands in layer L's prepared strip ("B": windows
// U + 2 wide, one border voxel on every side, its own pitch / lead; pre-zeroed by the host).
struct PrepEpi {
  const float* xnext;     // [cout, MD, MH, MW]  X_{L+1}: input of the conv being back-propagated = post-activation output of layer L
  const float* norm;      // [cout, MD, MH, MW]  norm conv of layer L
  const float* scale;     // [cout] eval-BatchNorm scale of layer L, or null
  const float* xoff;      // scalar: min X_{L+1}
  const int* origin;      // [P,3] origin (z,y,x) of the A windows in map coordinates
  int* origin_out;        // [P,3] = origin - 1
  int P, U, MD, MH, MW;
  int pitchA, leadA, slabA;          // slab: the strip's planes are the MAP's planes (plane index = qz), else the window's
  int pitchB, leadB, slabB;
  long long LB, ocs;                 // B row length; B channel stride (floats)
  int ozs;                           // B plane stride (floats) = (U + 2) * LB
  float inv_pitchA;
};

size_t packed_floats(int cin, int cout);                       // its own weight pack: 72 slots per (cout, cin)
int pack(const float* d_weight, int cin, int cout, float* d_packed, hipStream_t st);
// xt = tile id of the shared tile choice: 32 -> 64 x 4 x 2 outputs per workgroup, 16 -> 32 x 8 x 2, 8 -> 16 x 16 x 2 (no fused pool)
int launch(int xt, bool pool, bool argmax, const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W,
           m3d_w2q::Epi ep, hipStream_t st);
// the non-pooled launch with the fused prepare epilogue: `out` is the B strip (already zero), D / H / W are the A strip's dimensions
int launch_prep(int xt, const float* in, const float* wp, float* out, int cin, int cout, int D, int H, int W, m3d_w2q::Epi ep, const PrepEpi& pe,
                hipStream_t st);

}  // namespace m3d_w24
