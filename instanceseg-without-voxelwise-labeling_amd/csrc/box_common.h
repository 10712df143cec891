// Device helpers shared by box_ops.hip (multi-launch, any size) and box_fused.hip (one workgroup per item, batched).
// Every fp32 expression here is part of the bit-exactness contract with the reference (compile with -ffp-contract=off).
#pragma once
#include "m3d_common.h"

namespace m3dbox {

__device__ inline float fmax32(float a, float b) { return a >= b ? a : b; }   // cython_nms_3d.pyx:30-31
__device__ inline float fmin32(float a, float b) { return a <= b ? a : b; }   // cython_nms_3d.pyx:33-34

__device__ inline bool key_before(float ka, int ia, float kb, int ib) {
  // true if (ka, ia) is visited before (kb, ib): descending key, ties in descending index
  // ("argsort()[::-1]" under the build's stable-sort rule, SURVEY 8c caveat i)
  if (ka > kb) return true;
  if (ka < kb) return false;
  return ia > ib;
}

struct SBox { float x1, y1, z1, x2, y2, z2, vol, pad; };


// ---- decode + clip (boxes_3d.py:167-225, 144-163; NumPy-2 promotion: dw/dh/ds path in fp64) --------------------
struct XformParams { double w[6]; double clip; double cs, ch, cw; };

__device__ inline void decode_one(const float* b, const float* d, const XformParams& p, float* o) {
  float w = b[3] - b[0]; w = w + 1.0f;                                    // :177-179
  float h = b[4] - b[1]; h = h + 1.0f;
  float s = b[5] - b[2]; s = s + 1.0f;
  const float hw = 0.5f * w, hh = 0.5f * h, hs = 0.5f * s;
  const float cx = b[0] + hw, cy = b[1] + hh, cz = b[2] + hs;             // :180-182
  const float dx = d[0] / (float)p.w[0], dy = d[1] / (float)p.w[1], dz = d[2] / (float)p.w[2];   // :185-190
  const float dwf = d[3] / (float)p.w[3], dhf = d[4] / (float)p.w[4], dsf = d[5] / (float)p.w[5];
  const double dw = fmin((double)dwf, p.clip), dh = fmin((double)dhf, p.clip), ds = fmin((double)dsf, p.clip);   // :193-195
  float px = dx * w; px = px + cx;                                        // :197-199
  float py = dy * h; py = py + cy;
  float pz = dz * s; pz = pz + cz;
  const double pw = exp(dw) * (double)w, ph = exp(dh) * (double)h, ps = exp(ds) * (double)s;   // :200-202
  o[0] = (float)((double)px - 0.5 * pw);                                  // :213-223
  o[1] = (float)((double)py - 0.5 * ph);
  o[2] = (float)((double)pz - 0.5 * ps);
  o[3] = (float)(((double)px + 0.5 * pw) - 1.0);
  o[4] = (float)(((double)py + 0.5 * ph) - 1.0);
  o[5] = (float)(((double)pz + 0.5 * ps) - 1.0);
  if (p.cs > 0) {                                                         // clip_tiled_boxes_3d :152-162
    const double hi[6] = {p.cw - 1, p.ch - 1, p.cs - 1, p.cw - 1, p.ch - 1, p.cs - 1};
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      if (o[c] == o[c]) {   // NaN propagates in NumPy
        double v = (double)o[c];
        v = v < hi[c] ? v : hi[c];
        v = v > 0.0 ? v : 0.0;
        o[c] = (float)v;
      }
    }
  }
}


// 64-bit selection key: high word = fp32 score bits (scores are probabilities >= 0, so the unsigned
// bit pattern orders like the value; negatives/NaN are mapped to keep a total order), low word =
// ~flat_index so that among equal scores the SMALLER flat (S,H,W,A) index is the larger key.
__device__ inline unsigned int score_bits(float s) {
  unsigned int u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float bits_score(unsigned int b) {
  return __uint_as_float((b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b);
}
__device__ inline unsigned long long make_key(float s, unsigned int flat) {
  return ((unsigned long long)score_bits(s) << 32) | (unsigned long long)(0xFFFFFFFFu - flat);
}


struct PropParams {
  double anchors[6 * 64];   // A <= 64
  double stride, im_s, im_h, im_w, im_scale, min_size;
  XformParams xf;
  int A, S, H, W, batch_index;
};


// IoU test of cython_nms_3d.pyx:82-93 between sorted boxes bi (the earlier one) and bj
__device__ inline bool nms_suppresses(const SBox& bi, const SBox& bj, float thresh) {
  const float xx1 = fmax32(bi.x1, bj.x1), yy1 = fmax32(bi.y1, bj.y1), zz1 = fmax32(bi.z1, bj.z1);   // pyx:82-87
  const float xx2 = fmin32(bi.x2, bj.x2), yy2 = fmin32(bi.y2, bj.y2), zz2 = fmin32(bi.z2, bj.z2);
  float w = xx2 - xx1; w = w + 1.0f; w = fmax32(0.0f, w);                                             // pyx:88-90
  float h = yy2 - yy1; h = h + 1.0f; h = fmax32(0.0f, h);
  float s = zz2 - zz1; s = s + 1.0f; s = fmax32(0.0f, s);
  float inter = w * h; inter = inter * s;                                                            // pyx:91
  float uni = bi.vol + bj.vol; uni = uni - inter;
  const float ovr = inter / uni;                                                                     // pyx:92
  return ovr >= thresh;                                                                              // pyx:93
}

// volume of one detection row, NumPy fp32 left-to-right (pyx:48)
__device__ inline float det_volume(const float* d) {
  float a = d[3] - d[0]; a = a + 1.0f;
  float b = d[4] - d[1]; b = b + 1.0f;
  float c = d[5] - d[2]; c = c + 1.0f;
  float ab = a * b;
  return ab * c;
}

}  // namespace m3dbox
