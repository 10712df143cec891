// Internal interface of the Winograd F(2x4, 3x3) kernel family (conv3d_wino24.hip) used by conv3d_wino2.hip's entry points.
#pragma once
#include "conv3d_wino2q.h"

namespace m3d_w24 {

size_t packed_floats(int cin, int cout);                       // its own weight pack: 72 slots per (cout, cin)
int pack(const float* d_weight, int cin, int cout, float* d_packed, hipStream_t st);
// xt = tile id of the shared tile choice: 32 -> 64 x 4 x 2 outputs per workgroup, 16 -> 32 x 8 x 2, 8 -> 16 x 16 x 2 (no fused pool)
int launch(int xt, bool pool, bool argmax, const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W,
           m3d_w2q::Epi ep, hipStream_t st);

}  // namespace m3d_w24
