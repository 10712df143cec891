// Internal interface of the "wide" Winograd F(2x4, 3x3) kernel family (conv3d_wino24w.hip; family 5) used by conv3d_wino2.hip's entry
// points.  It reads conv3d_wino24.hip's weight pack.
#pragma once
#include "conv3d_wino2q.h"

namespace m3d_w24w {

// xt = tile id of the shared tile choice: 32 -> 64 x 2 x 2 outputs x 64 channels per workgroup, 16 -> 32 x 4 x 2, 8 -> 16 x 8 x 2 (no fused pool)
int launch(int xt, bool pool, bool argmax, const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W,
           m3d_w2q::Epi ep, hipStream_t st);

}  // namespace m3d_w24w
