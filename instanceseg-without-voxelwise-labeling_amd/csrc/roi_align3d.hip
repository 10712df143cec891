// RoIAlign 3D forward/backward for gfx950.
//
// Semantics follow lib/modeling/roi_xfrom/roi_align_3d/src/roi_align_kernel_3d.cu:16-151 (forward) and
// :180-338 (backward) of the reference, including the (n,c,ph,pw,ps) output memory order (:87-91) and the
// backward's different top_diff permutation (:274-275) and `z < -0.1` test (:187).
//
// Design (not the reference's one-thread-per-element grid): one workgroup owns one RoI and a chunk of
// channels.  The per-axis sample tables (low/high index, low/high weight, validity) depend only on the RoI,
// so they are computed once per workgroup into LDS (3 * A * grid entries) instead of once per output
// element per channel; each lane then produces output elements in flat (ph,pw,ps) order so that the
// [R,C,343] output — the HBM-write-bound part, 351 MB at R=1000 — is written fully coalesced.
// This translation unit is compiled with -ffp-contract=off: the fp32 operation order is the contract
// with oracle/m3d_oracle.c.
#include <stddef.h>

#include "m3d_common.h"

namespace {

struct AxisSample {  // one (bin, sub-sample) along one axis
  int lo, hi;        // clamped corner indices
  float l, h;        // weights of hi / lo corners (l = frac, h = 1 - frac)
  int valid;         // 0 => whole sample contributes 0 (coordinate outside [-1, dim])
};

constexpr int kMaxTable = 64;   // A * grid per axis (7 bins * adaptive grid <= 9); larger adaptive grids take roi_untabled_range

__device__ inline AxisSample make_sample(float start, float bin, int p, int i, int grid, int dim, double lo_limit) {
  // coordinate: roi_start + p*bin + (i + .5f)*bin/grid   (roi_align_kernel_3d.cu:130-138)
  float c = start + p * bin;
  c = c + (i + .5f) * bin / grid;
  AxisSample s;
  s.valid = !((double)c < lo_limit || c > dim);  // :19 (forward: -1.0 on every axis; backward: -0.1 on z, :187)
  if (c <= 0) c = 0;                     // :23-31
  int lo = (int)c;
  int hi;
  if (lo >= dim - 1) { hi = lo = dim - 1; c = (float)lo; } else { hi = lo + 1; }   // :40-58
  float l = c - lo;
  float h = (float)(1. - l);             // :63 (double literal)
  s.lo = lo; s.hi = hi; s.l = l; s.h = h;
  return s;
}

struct RoiGeom {
  float start_w, start_h, start_s, bin_w, bin_h, bin_s;
  int grid_w, grid_h, grid_s, batch;
};

__device__ inline RoiGeom roi_geom(const float* r, float scale, int AS, int AH, int AW, int ratio, int B) {
  RoiGeom g;
  g.batch = min(max((int)r[0], 0), B - 1);                              // :94 (clamped: a garbage row must not fault the GPU)
  g.start_w = r[1] * scale; g.start_h = r[2] * scale; g.start_s = r[3] * scale;   // :97-102
  float end_w = r[4] * scale, end_h = r[5] * scale, end_s = r[6] * scale;
  float roi_s = fmaxf(end_s - g.start_s, 1.f);                          // :105-107
  float roi_w = fmaxf(end_w - g.start_w, 1.f);
  float roi_h = fmaxf(end_h - g.start_h, 1.f);
  g.bin_s = roi_s / AS; g.bin_h = roi_h / AH; g.bin_w = roi_w / AW;     // :108-110
  g.grid_s = ratio > 0 ? ratio : (int)ceilf(roi_s / AS);                // :116-123
  g.grid_h = ratio > 0 ? ratio : (int)ceilf(roi_h / AH);
  g.grid_w = ratio > 0 ? ratio : (int)ceilf(roi_w / AW);
  return g;
}

// exact reference operation order for output elements [c0,c1) x bins of RoI n (tables already in LDS)
__device__ inline void roi_exact_forward_range(const float* __restrict__ feat, float* __restrict__ out, const RoiGeom& g,
                                               const AxisSample* tz, const AxisSample* ty, const AxisSample* tx, int n, int c0,
                                               int c1, int C, int S, int H, int W, int AS, int AH, int AW) {
  const int bins = AS * AH * AW;
  const float count = (float)(g.grid_s * g.grid_h * g.grid_w);         // :126
  const int HW = H * W;
  const int total = (c1 - c0) * bins;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int c = c0 + e / bins;
    const int b = e % bins;
    const int ps = b % AS;                                             // thread order of the reference (:87-89)
    const int pw = (b / AS) % AW;
    const int ph = b / AS / AW;
    const float* data = feat + ((size_t)g.batch * C + c) * S * HW;     // :112-113
    float acc = 0.f;
    for (int iz = 0; iz < g.grid_s; ++iz) {
      const AxisSample z = tz[ps * g.grid_s + iz];
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const AxisSample y = ty[ph * g.grid_h + iy];
        const float hzhy = z.h * y.h, hzly = z.h * y.l, lzhy = z.l * y.h, lzly = z.l * y.l;
        const float* p00 = data + z.lo * HW + y.lo * W;
        const float* p01 = data + z.lo * HW + y.hi * W;
        const float* p10 = data + z.hi * HW + y.lo * W;
        const float* p11 = data + z.hi * HW + y.hi * W;
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const AxisSample x = tx[pw * g.grid_w + ix];
          float val = 0.f;
          if (z.valid & y.valid & x.valid) {
            const float w1 = hzhy * x.h, w2 = hzhy * x.l, w3 = hzly * x.h, w4 = hzly * x.l;   // :73-74
            const float w5 = lzhy * x.h, w6 = lzhy * x.l, w7 = lzly * x.h, w8 = lzly * x.l;
            val = w1 * p00[x.lo];                                       // :76 left-to-right
            val = val + w2 * p00[x.hi];
            val = val + w3 * p01[x.lo];
            val = val + w4 * p01[x.hi];
            val = val + w5 * p10[x.lo];
            val = val + w6 * p10[x.hi];
            val = val + w7 * p11[x.lo];
            val = val + w8 * p11[x.hi];
          }
          acc += val;                                                   // :142
        }
      }
    }
    acc /= count;                                                       // :147
    out[((size_t)n * C + c) * bins + b] = acc;                          // :149 (index == (n,c,ph,pw,ps))
  }
}

// Adaptive sampling grids (sampling_ratio <= 0, roi_align_kernel_3d.cu:116-123) grow with the RoI: grid = ceil(roi / A) per axis,
// so a 100-voxel RoI has 15 samples per bin and no longer fits the LDS tables.  Such RoIs take this untabled path: the same
// make_sample arithmetic evaluated per sample (same values, same summation order as the tabled path and the oracle), just slower.
template <bool kBackward>
__device__ inline void roi_untabled_range(const float* __restrict__ feat_or_top, float* __restrict__ out_or_grad, const RoiGeom& g,
                                          int n, int c0, int c1, int C, int S, int H, int W, int AS, int AH, int AW) {
  const int bins = AS * AH * AW;
  const float count = (float)(g.grid_s * g.grid_h * g.grid_w);
  const int HW = H * W;
  const int total = (c1 - c0) * bins;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int c = c0 + e / bins;
    const int b = e % bins;
    const int ps = b % AS, pw = (b / AS) % AW, ph = b / AS / AW;
    const size_t plane = ((size_t)g.batch * C + c) * S * HW;
    float acc = 0.f;
    const float t = kBackward ? feat_or_top[((size_t)n * C + c) * bins + ps * AH * AW + ph * AW + pw] : 0.f;   // :272-275
    for (int iz = 0; iz < g.grid_s; ++iz) {
      const AxisSample z = make_sample(g.start_s, g.bin_s, ps, iz, g.grid_s, S, kBackward ? -0.1 : -1.0);
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const AxisSample y = make_sample(g.start_h, g.bin_h, ph, iy, g.grid_h, H, -1.0);
        const float hzhy = z.h * y.h, hzly = z.h * y.l, lzhy = z.l * y.h, lzly = z.l * y.l;
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const AxisSample x = make_sample(g.start_w, g.bin_w, pw, ix, g.grid_w, W, -1.0);
          const bool ok = z.valid & y.valid & x.valid;
          const float w[8] = {hzhy * x.h, hzhy * x.l, hzly * x.h, hzly * x.l, lzhy * x.h, lzhy * x.l, lzly * x.h, lzly * x.l};
          const int idx[8] = {z.lo * HW + y.lo * W + x.lo, z.lo * HW + y.lo * W + x.hi, z.lo * HW + y.hi * W + x.lo,
                              z.lo * HW + y.hi * W + x.hi, z.hi * HW + y.lo * W + x.lo, z.hi * HW + y.lo * W + x.hi,
                              z.hi * HW + y.hi * W + x.lo, z.hi * HW + y.hi * W + x.hi};
          if (kBackward) {
            if (!ok) continue;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              float gq = t * w[q];
              gq = gq / count;
              atomicAdd(out_or_grad + plane + idx[q], gq);
            }
          } else {
            float val = 0.f;
            if (ok) {
              const float* data = feat_or_top + plane;
              val = w[0] * data[idx[0]];
#pragma unroll
              for (int q = 1; q < 8; ++q) val = val + w[q] * data[idx[q]];
            }
            acc += val;
          }
        }
      }
    }
    if (!kBackward) {
      acc /= count;
      out_or_grad[((size_t)n * C + c) * bins + b] = acc;
    }
  }
}

// grid = (num_rois, channel_chunks); block = 256
template <bool kBackward>
__global__ __launch_bounds__(256) void roi_align3d_kernel(const float* __restrict__ feat_or_top, const float* __restrict__ rois,
                                                          float* __restrict__ out_or_grad, int B, int C, int S, int H, int W,
                                                          int AS, int AH, int AW, float scale, int ratio, int ch_per_block,
                                                          int* __restrict__ status) {
  __shared__ AxisSample tz[kMaxTable], ty[kMaxTable], tx[kMaxTable];
  __shared__ RoiGeom sg;
  const int n = blockIdx.x;
  if (threadIdx.x == 0) sg = roi_geom(rois + 7 * n, scale, AS, AH, AW, ratio, B);
  __syncthreads();
  const RoiGeom g = sg;
  if (AS * g.grid_s > kMaxTable || AH * g.grid_h > kMaxTable || AW * g.grid_w > kMaxTable) {   // adaptive grid beyond the tables
    const int c0u = blockIdx.y * ch_per_block;
    roi_untabled_range<kBackward>(feat_or_top, out_or_grad, g, n, c0u, min(C, c0u + ch_per_block), C, S, H, W, AS, AH, AW);
    return;
  }
  for (int t = threadIdx.x; t < AS * g.grid_s; t += blockDim.x)
    tz[t] = make_sample(g.start_s, g.bin_s, t / g.grid_s, t % g.grid_s, g.grid_s, S, kBackward ? -0.1 : -1.0);
  for (int t = threadIdx.x; t < AH * g.grid_h; t += blockDim.x)
    ty[t] = make_sample(g.start_h, g.bin_h, t / g.grid_h, t % g.grid_h, g.grid_h, H, -1.0);
  for (int t = threadIdx.x; t < AW * g.grid_w; t += blockDim.x)
    tx[t] = make_sample(g.start_w, g.bin_w, t / g.grid_w, t % g.grid_w, g.grid_w, W, -1.0);
  __syncthreads();

  const int bins = AS * AH * AW;
  const int c0 = blockIdx.y * ch_per_block;
  const int c1 = min(C, c0 + ch_per_block);
  if (!kBackward) {
    roi_exact_forward_range(feat_or_top, out_or_grad, g, tz, ty, tx, n, c0, c1, C, S, H, W, AS, AH, AW);
    return;
  }
  const float count = (float)(g.grid_s * g.grid_h * g.grid_w);         // :288
  const int HW = H * W;
  const int total = (c1 - c0) * bins;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int c = c0 + e / bins;
    const int b = e % bins;
    const int ps = b % AS;
    const int pw = (b / AS) % AW;
    const int ph = b / AS / AW;
    {
      float* gd = out_or_grad + ((size_t)g.batch * C + c) * S * HW;
      const float t = feat_or_top[((size_t)n * C + c) * bins + ps * AH * AW + ph * AW + pw];   // :272-275
      for (int iz = 0; iz < g.grid_s; ++iz) {
        const AxisSample z = tz[ps * g.grid_s + iz];
        for (int iy = 0; iy < g.grid_h; ++iy) {
          const AxisSample y = ty[ph * g.grid_h + iy];
          const float hzhy = z.h * y.h, hzly = z.h * y.l, lzhy = z.l * y.h, lzly = z.l * y.l;
          for (int ix = 0; ix < g.grid_w; ++ix) {
            const AxisSample x = tx[pw * g.grid_w + ix];
            if (!(z.valid & y.valid & x.valid)) continue;             // :187-192, :320
            const float w[8] = {hzhy * x.h, hzhy * x.l, hzly * x.h, hzly * x.l, lzhy * x.h, lzhy * x.l, lzly * x.h, lzly * x.l};
            const int idx[8] = {z.lo * HW + y.lo * W + x.lo, z.lo * HW + y.lo * W + x.hi, z.lo * HW + y.hi * W + x.lo,
                                z.lo * HW + y.hi * W + x.hi, z.hi * HW + y.lo * W + x.lo, z.hi * HW + y.lo * W + x.hi,
                                z.hi * HW + y.hi * W + x.lo, z.hi * HW + y.hi * W + x.hi};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              float gq = t * w[q];
              gq = gq / count;                                        // :311-318
              atomicAdd(gd + idx[q], gq);                             // :325-332
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Fast forward: separable form.  Trilinear weights are products of per-axis weights and the bin average is
// a sum over a product grid, so  out[ps,ph,pw] = 1/count * sum_z Az[ps][z] * sum_y Ay[ph][y] * sum_x Ax[pw][x] * f[z,y,x]
// with <= 2*grid taps per bin and axis.  Three 1-D passes through LDS need ~4x fewer operand reads than the
// 64-loads-per-output reference order (the kernel above), which is what separates it from the HBM-write
// roofline.  Same sample positions, validity rule and weights; only the fp32 summation order differs
// (|diff| <~ 1e-6 * max|f|, tests use 1e-5).  One workgroup = one RoI x a range of channels, processed CH
// channels at a time so that sub-volume + both intermediates fit in LDS.
// ------------------------------------------------------------------------------------------------------
constexpr int kSepLdsFloats = 8 * 1024;    // 32 KB of dynamic LDS per workgroup, 2048 floats per wave
constexpr int kSepStageMax = 1536;         // largest sub-volume a wave stages through registers (24 per lane)

// ------------------------------------------------------------------------------------------------------
// v3 fast path for the shipped geometry (7x7x7 bins, sampling grid 2).  Per axis and bin the two samples' four taps lie on at most
// four CONSECUTIVE voxels (bins up to 4 voxels wide), so they fold into one base index + four weights; the passes run
//   X: lanes (row mod 9, pw)   t1[z][y][pw]   = sum_k xw[pw][k]  * f [z][y][xb+k]       taps fixed per lane
//   Z: lanes (pw, ps)          t2[y][pw][ps]  = sum_k zw[ps][k]  * t1[zb+k][y][pw]      taps fixed per lane
//   Y: lanes (pw, ps), ph loop out[ph][pw][ps] = sum_k yw[ph][k] * t2[yb+k][pw][ps]     taps wave-uniform (LDS broadcast)
// so a lane carries 10 tap registers instead of 72, the last pass writes 49 consecutive floats per iteration straight to HBM
// (7 stores of 196 B per channel, no scattered dwords), and the kernel fits 5 waves per SIMD instead of 2.  Each wave works
// alone in its LDS slice (no workgroup barriers after the set-up) on TWO channels at a time - two independent dependency chains
// per pass - and the next channels' sub-volumes arrive by LDS-DMA (global_load_lds_dword) while the current ones are reduced.
// RoIs that do not fit (wider bins, sub-volume + intermediates beyond the slice) are left to roi_align3d_fwd_sep_kernel.
// ------------------------------------------------------------------------------------------------------
struct Fold { int k[4]; float w[4]; };     // positions (relative to the sub-volume origin, clamped to the extent) and weights

__device__ inline bool fold_bin(const AxisSample& s0, const AxisSample& s1, int origin, int extent, float scale, Fold* f) {
  int base = 1 << 30;
  if (s0.valid) base = min(base, s0.lo);
  if (s1.valid) base = min(base, s1.lo);
  if (base == (1 << 30)) base = origin;
  bool ok = true;
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
  auto add = [&](int pos, float w) {
    const int k = pos - base;
    const float v = w * scale;
    if (k < 0 || k > 3) ok = false;
    w0 += k == 0 ? v : 0.f; w1 += k == 1 ? v : 0.f; w2 += k == 2 ? v : 0.f; w3 += k == 3 ? v : 0.f;
  };
  if (s0.valid) { add(s0.lo, s0.h); add(s0.hi, s0.l); }
  if (s1.valid) { add(s1.lo, s1.h); add(s1.hi, s1.l); }
  if (f) {
    f->w[0] = w0; f->w[1] = w1; f->w[2] = w2; f->w[3] = w3;
#pragma unroll
    for (int k = 0; k < 4; ++k) f->k[k] = min(base - origin + k, extent - 1);
  }
  return ok;
}

struct V3Dims { int ez, ey, ex, sub, subp, n1, n2, per_ch; };   // subp: sub-volume padded to whole 64-lane DMA pieces
__device__ inline V3Dims v3_dims(const int* rng) {
  V3Dims d;
  d.ez = rng[1] - rng[0] + 1; d.ey = rng[3] - rng[2] + 1; d.ex = rng[5] - rng[4] + 1;
  // The four taps of a fold are consecutive voxels, read UNCLAMPED as base + 0..3 so that the compiler pairs them into ds_read2_b32
  // (the kernel is bound by the LDS instruction rate); taps past the extent have weight 0 and must only hit finite memory: the
  // sub-volume is padded by >= 3 floats (the DMA's tail lanes re-copy the last voxel there) and t2 by three zeroed rows.
  d.sub = d.ez * d.ey * d.ex; d.subp = (d.sub + 3 + 63) / 64 * 64; d.n1 = d.ez * d.ey * 7; d.n2 = (d.ey + 3) * 49;
  d.per_ch = d.subp + d.n1 + d.n2;
  return d;
}

// one thread: does the v3 kernel take this RoI?  (tables for 7 bins x 2 samples per axis; rng = valid index ranges)
__device__ inline bool v3_qualifies(const AxisSample* tz, const AxisSample* ty, const AxisSample* tx, const int* rng) {
  if (rng[1] < 0 || rng[3] < 0 || rng[5] < 0) return false;
  const V3Dims d = v3_dims(rng);
  if (d.per_ch > kSepLdsFloats) return false;              // even one wave with the whole workgroup's LDS cannot hold a channel
  for (int p = 0; p < 7; ++p) {
    if (!fold_bin(tz[2 * p], tz[2 * p + 1], rng[0], d.ez, 1.f, nullptr)) return false;
    if (!fold_bin(ty[2 * p], ty[2 * p + 1], rng[2], d.ey, 1.f, nullptr)) return false;
    if (!fold_bin(tx[2 * p], tx[2 * p + 1], rng[4], d.ex, 1.f, nullptr)) return false;
  }
  return true;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned int kDeclinedBits = 0x7FC0DEADu;      // quiet NaN with a payload: arithmetic only ever yields 0x7FC00000 / 0xFFC00000

template <bool DUAL>
__device__ inline void v3_run(const float* __restrict__ fbase /* feature map of the RoI's batch item, channel 0 */, size_t chan_stride,
                              float* __restrict__ obase /* out + n*C*343 */, float* wl /* this wave's LDS slice */, const Fold* fz,
                              const Fold* fy, const Fold* fx, const V3Dims d, int HW, int W, int gofs /* z0*HW + y0*W + x0 */,
                              int c_first, int c_end, int c_step /* channels between two of this wave's channels */) {
  const int lane = threadIdx.x & 63;
  constexpr int NQ = DUAL ? 2 : 1;
  // X-pass lane role
  const int rx = lane / 7, pw_x = lane % 7;
  const Fold FX = fx[pw_x];
  // Z/Y-pass lane role: l = pw*7 + ps
  const int pw_z = (lane % 49) / 7, ps_z = lane % 7;
  Fold FZ = fz[ps_z];
#pragma unroll
  for (int k = 0; k < 4; ++k) FZ.k[k] *= d.ey * 7;
  // the Y folds are wave-uniform and the same for every channel: scalar registers, not 56 LDS broadcast reads per iteration (the
  // kernel is bound by the CU's LDS instruction rate: ~170 wave-level LDS operations per channel pair with 16-20 waves per CU)
  float yw[7][4]; int yk[7];
#pragma unroll
  for (int ph = 0; ph < 7; ++ph) {
    yk[ph] = __builtin_amdgcn_readfirstlane(fy[ph].k[0]) * 49;
#pragma unroll
    for (int k = 0; k < 4; ++k) yw[ph][k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fy[ph].w[k])));
  }
  const int nrows = d.ez * d.ey;
  const float inv_ex = 1.0f / (float)d.ex, inv_ey = 1.0f / (float)d.ey;
  const int nst = d.subp / 64;
  auto dma = [&](int c, float* dst) __attribute__((always_inline)) {     // sub-volume of channel c -> LDS, element e = lane + 64 i
    const float* fc = fbase + (size_t)c * chan_stride + gofs;
    for (int i = 0; i < nst; ++i) {
      const int e = min(lane + 64 * i, d.sub - 1);                       // tail lanes re-copy the last element (into the padding)
      const int r = (int)(((float)e + 0.5f) * inv_ex), x = e - r * d.ex;
      const int z = (int)(((float)r + 0.5f) * inv_ey), y = r - z * d.ey;
#ifndef RA_NO_DMA
      __builtin_amdgcn_global_load_lds(fc + (size_t)z * HW + y * W + x, (lds_ptr_t)(dst + 64 * i), 4, 0, 0);
#else
      if (z == 12345) __builtin_amdgcn_global_load_lds(fc + (size_t)z * HW + y * W + x, (lds_ptr_t)(dst + 64 * i), 4, 0, 0);
#endif
    }
  };
  // LDS slice: per channel q: fsub[subp], t1[n1], t2[n2]
  float* fs[NQ]; float* t1[NQ]; float* t2[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { fs[q] = wl + q * d.per_ch; t1[q] = fs[q] + d.subp; t2[q] = t1[q] + d.n1; }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    for (int e = lane; e < 3 * 49; e += 64) t2[q][d.ey * 49 + e] = 0.f;      // the rows the unclamped Y taps may touch
    if (c_first + c_step * q < c_end) dma(c_first + c_step * q, fs[q]);
  }
  bool first = true;
  for (int c = c_first; c < c_end; c += c_step * NQ) {
    // The sub-volumes of this iteration were requested BEFORE the previous iteration's 7*NQ output stores; vector-memory
    // operations retire in order, so leaving those stores in flight still guarantees the copies have landed.
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DUAL) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    first = false;
    if (rx < 9) {                                                        // pass X
      for (int zy = rx; zy < nrows; zy += 9) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float* row = fs[q] + zy * d.ex + FX.k[0];
          t1[q][zy * 7 + pw_x] = (FX.w[0] * row[0] + FX.w[1] * row[1]) + (FX.w[2] * row[2] + FX.w[3] * row[3]);
        }
      }
    }
    const bool more = c + c_step * (2 * NQ - 1) < c_end;                 // a FULL next iteration follows (its waits count 7*NQ stores)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // every read of fsub has returned: the buffers are free
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (c + c_step * (NQ + q) < c_end) dma(c + c_step * (NQ + q), fs[q]);
    if (lane < 49) {
      for (int y = 0; y < d.ey; ++y) {                                   // pass Z
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float* col = t1[q] + y * 7 + pw_z;
          t2[q][y * 49 + lane] = (FZ.w[0] * col[FZ.k[0]] + FZ.w[1] * col[FZ.k[1]]) + (FZ.w[2] * col[FZ.k[2]] + FZ.w[3] * col[FZ.k[3]]);
        }
      }
    }
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) {                                     // pass Y (1/count folded into fy) -> HBM, 49 floats in a row
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float* col = t2[q] + (lane < 49 ? lane : 0);
        const float* cy = col + yk[ph];
        const float v = (yw[ph][0] * cy[0] + yw[ph][1] * cy[49]) + (yw[ph][2] * cy[98] + yw[ph][3] * cy[147]);
        // the store is issued by the wave even when this channel does not exist (exec = 0): vmcnt counts per wave instruction
        const bool on = c + c_step * q < c_end;
        float* dst = obase + (size_t)(on ? c + c_step * q : c) * 343 + ph * 49 + lane;
#ifndef RA_NO_STORE
        if (lane < 49 && on) *dst = v;
#else
        if (lane < 49 && on && v == 123.456f) *dst = v;
#endif
      }
    }
    if (!more) first = true;                                             // a ragged last iteration: wait for everything
  }
}

// The per-RoI set-up of the v3 path (geometry, 7 x 2 sample tables per axis, valid ranges, the 21 folds); 256 threads, barriers
// inside.  Returns through shared memory: sg, rng, s_ok (1: the v3 kernel takes the RoI), fz / fy / fx.
struct V3Shared {
  AxisSample tz[14], ty[14], tx[14];
  Fold fz[7], fy[7], fx[7];
  RoiGeom sg;
  int rng[6];
  int s_ok;
};
__device__ inline V3Dims v3_setup(const float* __restrict__ rois, int n, float scale, int B, int S, int H, int W, V3Shared& sh) {
  const int tid = threadIdx.x;
  if (tid == 0) sh.sg = roi_geom(rois + 7 * n, scale, 7, 7, 7, 2, B);
  __syncthreads();
  const RoiGeom g = sh.sg;
  if (tid < 14) sh.tz[tid] = make_sample(g.start_s, g.bin_s, tid / 2, tid % 2, 2, S, -1.0);
  else if (tid >= 64 && tid < 78) sh.ty[tid - 64] = make_sample(g.start_h, g.bin_h, (tid - 64) / 2, (tid - 64) % 2, 2, H, -1.0);
  else if (tid >= 128 && tid < 142) sh.tx[tid - 128] = make_sample(g.start_w, g.bin_w, (tid - 128) / 2, (tid - 128) % 2, 2, W, -1.0);
  __syncthreads();
  if (tid < 3) {
    const AxisSample* t = tid == 0 ? sh.tz : (tid == 1 ? sh.ty : sh.tx);
    int lo = 1 << 30, hi = -1;
    for (int i = 0; i < 14; ++i)
      if (t[i].valid) { lo = min(lo, t[i].lo); hi = max(hi, t[i].hi); }
    sh.rng[2 * tid] = lo; sh.rng[2 * tid + 1] = hi;
  }
  __syncthreads();
  if (tid == 0) sh.s_ok = (sh.rng[1] >= 0 && sh.rng[3] >= 0 && sh.rng[5] >= 0) ? 1 : 0;
  __syncthreads();
  const V3Dims d = v3_dims(sh.rng);
  if (tid < 21 && sh.s_ok) {                                // the 21 (axis, bin) folds in parallel; any failure clears s_ok
    const int ax = tid / 7, p = tid % 7;
    bool ok;
    if (ax == 0) ok = fold_bin(sh.tz[2 * p], sh.tz[2 * p + 1], sh.rng[0], d.ez, 1.f, &sh.fz[p]);
    else if (ax == 1) ok = fold_bin(sh.ty[2 * p], sh.ty[2 * p + 1], sh.rng[2], d.ey, 0.125f, &sh.fy[p]);      // 1 / (2*2*2 samples)
    else ok = fold_bin(sh.tx[2 * p], sh.tx[2 * p + 1], sh.rng[4], d.ex, 1.f, &sh.fx[p]);
    if (!ok || d.per_ch > kSepLdsFloats) atomicAnd(&sh.s_ok, 0);
  }
  __syncthreads();
  return d;
}

// The part of V3Shared the passes read (folds, geometry, ranges, s_ok: the members from `fz` to the end) as a per-RoI record in global
// memory: roi_class_kernel runs the set-up ONCE per RoI and stores it, the working groups of the v3 launch load it (one coalesced read,
// one barrier) instead of repeating the set-up - geometry on one thread, 42 samples, three serial range scans, 21 folds, six barriers -
// in every one of them.  Ablation (round 5, timing-only builds): with passes, LDS-DMA and stores compiled out the launch set still took
// 0.087 of 0.232 ms at R = 1281 and 0.353 of 0.974 ms at R = 3582 (883 large RoIs x 32 channel octets = 28 k set-ups).
constexpr int kV3TabWords = (int)((sizeof(V3Shared) - offsetof(V3Shared, fz)) / sizeof(int));
__device__ inline void v3_store_tab(const V3Shared& sh, int* __restrict__ tab) {
  const int* src = reinterpret_cast<const int*>(&sh.fz);
  for (int t = threadIdx.x; t < kV3TabWords; t += 256) tab[t] = src[t];
}
__device__ inline V3Dims v3_load_tab(const int* __restrict__ tab, V3Shared& sh) {
  int* dst = reinterpret_cast<int*>(&sh.fz);
  for (int t = threadIdx.x; t < kV3TabWords; t += 256) dst[t] = tab[t];
  __syncthreads();
  return v3_dims(sh.rng);
}

// waves that work on a RoI: all 4 with a quarter of the LDS each, or - sub-volume + intermediates too large for that - 2 or 1 with
// a half / all of it
__device__ inline int v3_waves(const V3Dims& d) {
  int nw = 4;
  while (nw > 1 && d.per_ch > kSepLdsFloats / nw) nw >>= 1;
  return nw;
}

// Work split of the forward pass, decided once per RoI and handed to the workgroups THROUGH THE OUTPUT: a NaN payload no arithmetic
// produces, at the first element of every 8th channel (the places a channel chunk can start; real results overwrite them).
//   small   (all four waves fit; 94 % of the detection RoIs): ONE workgroup does all channels - the set-up (geometry, tables,
//           folds: several barriers) is paid once per RoI instead of once per 32 channels;
//   medium  (1-2 waves fit): one workgroup per 8 channels, so the few long RoIs are cut fine and do not form the tail;
//   declined (bins wider than 4 voxels / sub-volume beyond the LDS): the complement pass (roi_align3d_fwd_sep_kernel) does it.
// Every other workgroup of the (RoI, C/8) grid leaves after one load.
constexpr unsigned int kSmallBits = 0x7FC05A11u, kMedBits = 0x7FC03ED0u;
// round 6: RoIs whose sub-volume has <= 64 / <= 128 voxels go to the matrix-core form (roi_align3d_fwd_gemm_kernel<4> / <8>) when the caller
// supplied the feature maps' largest magnitude (the f16x2 split needs an operand scale)
constexpr unsigned int kGemm4Bits = 0x7FC06E34u, kGemm8Bits = 0x7FC06E38u;

// LDS floats one channel of RoI m needs in the v3 kernel, estimated from its corners alone (extent of the sample positions per axis):
// the launch order's sort key - heavy RoIs first.  Any deterministic function of the RoI gives a valid order; this one tracks v3_dims.
__device__ inline int roi_cost(const float* __restrict__ r, float scale, int S, int H, int W) {
  int e[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int dim = a == 0 ? W : (a == 1 ? H : S);
    const float lo = r[1 + a] * scale, hi = r[4 + a] * scale;
    const float roi = fmaxf(hi - lo, 1.f);
    const float c0 = lo + 0.25f * roi / 7.f, c1 = lo + roi - 0.25f * roi / 7.f;
    const int l0 = min(max((int)floorf(fmaxf(c0, 0.f)), 0), dim - 1), l1 = min(max((int)floorf(fmaxf(c1, 0.f)) + 1, 0), dim - 1);
    e[a] = max(l1 - l0 + 1, 1);
  }
  return e[0] * e[1] * e[2] + e[2] * e[1] * 7 + (e[1] + 3) * 49;
}

// order (optional, int [R]): order[k] = the RoI the k-th workgroup column of the v3 launch takes - RoIs by descending cost (ties by
// index), so that the few long RoIs start first instead of forming the launch's tail (measured on the bench's RoIs: 0.269 -> 0.214 ms).
// Every workgroup ranks its own RoI against all R (R / 256 cost evaluations per thread): no second launch, no atomics.
__global__ __launch_bounds__(256) void roi_class_kernel(const float* __restrict__ rois, float* __restrict__ out, int B, int C, int S, int H,
                                                        int W, float scale, int R, int* __restrict__ order, int* __restrict__ tabs, int gemm) {
  __shared__ V3Shared sh;
  __shared__ int s_rank[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const V3Dims d = v3_setup(rois, n, scale, B, S, H, W, sh);
  unsigned int cls = !sh.s_ok ? kDeclinedBits : (v3_waves(d) == 4 ? kSmallBits : kMedBits);
  if (gemm && sh.s_ok && d.sub <= 128 && d.ez <= 16 && d.ey <= 16 && d.ex <= 16) cls = d.sub <= 64 ? kGemm4Bits : kGemm8Bits;
  for (int k = tid; 8 * k < C; k += 256) out[((size_t)n * C + 8 * k) * 343] = __uint_as_float(cls);
  if (tabs) v3_store_tab(sh, tabs + (size_t)n * kV3TabWords);
  if (order) {
    const int mine = roi_cost(rois + 7 * (size_t)n, scale, S, H, W);
    int before = 0;
    for (int m = tid; m < R; m += 256) {
      const int c = roi_cost(rois + 7 * (size_t)m, scale, S, H, W);
      before += (c > mine || (c == mine && m < n)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o, 64);
    if ((tid & 63) == 0) s_rank[tid >> 6] = before;
    __syncthreads();
    if (tid == 0) order[s_rank[0] + s_rank[1] + s_rank[2] + s_rank[3]] = n;
  }
}

// marks = 1: grid (R, C / 8), work split read from the markers.  marks = 0 (C not a multiple of 32): grid (R, ceil(C / ch_per_block)),
// every workgroup does the set-up and takes its ch_per_block channels if the RoI qualifies.
__global__ __launch_bounds__(256) void roi_align3d_fwd_v3_kernel(const float* __restrict__ feat, const float* __restrict__ rois,
                                                                 float* __restrict__ out, int B, int C, int S, int H, int W, float scale,
                                                                 int ch_per_block, int marks, const int* __restrict__ order,
                                                                 const int* __restrict__ tabs) {
  __shared__ V3Shared sh;
  extern __shared__ float dyn[];
  const int tid = threadIdx.x;
  int n, c0, c1;
  if (marks == 2) {
    // XCD-aware split (round 6; C a multiple of 64, 1-D grid of R * C / 8 workgroups): workgroups are dealt to the 8 XCDs round-robin by
    // their linear id, so id & 7 IS the XCD, and XCD g owns the channels [g C / 8, (g + 1) C / 8) of EVERY RoI: the part of the feature
    // maps an XCD's L2 (4 MB) has to hold is 1/8 of them (4 x 128^3 volumes: 2.1 MB instead of 16.8 MB).  The first 8 R workgroups are
    // (RoI rank, XCD): a small RoI's C / 8 channels, or the first octet of a medium one; the remaining (C / 64 - 1) 8 R are the other
    // octets of the medium RoIs.  (Working workgroups FIRST in dispatch order: with them spread through the grid, one in 32, the
    // launch is bound by the dispatch of the 41 000 that exit at once - 0.53 instead of 0.22 ms at R = 1281.)
    const int per = C >> 6;                                   // channel octets per XCD
    const int R8 = 8 * (int)(gridDim.x / (unsigned)(8 * per));
    int xcd, r, sub;
    if ((int)blockIdx.x < R8) { xcd = blockIdx.x & 7; r = blockIdx.x >> 3; sub = 0; }
    else {
      if (per == 1) return;
      const int id2 = (int)blockIdx.x - R8, k = id2 >> 3;
      xcd = id2 & 7; r = k / (per - 1); sub = 1 + k - r * (per - 1);
    }
    const int oct = xcd * per + sub;
    n = order ? order[r] : r;                                 // launch order: heavy RoIs first (roi_class_kernel)
    const unsigned int m = __float_as_uint(out[((size_t)n * C + 8 * oct) * 343]);
    if (m == kMedBits) { c0 = 8 * oct; c1 = c0 + 8; }
    else if (m == kSmallBits && sub == 0) { c0 = 8 * xcd * per; c1 = c0 + 8 * per; }
    else return;
  } else {
    n = order ? order[blockIdx.x] : (int)blockIdx.x;
    c0 = blockIdx.y * ch_per_block; c1 = min(C, c0 + ch_per_block);
    if (marks) {
      const unsigned int m = __float_as_uint(out[((size_t)n * C + 8 * blockIdx.y) * 343]);
      if (m == kMedBits) { c0 = 8 * blockIdx.y; c1 = min(C, c0 + 8); }
      else if (m == kSmallBits && blockIdx.y == 0) { c0 = 0; c1 = C; }
      else return;
    }
  }
  const V3Dims d = tabs ? v3_load_tab(tabs + (size_t)n * kV3TabWords, sh) : v3_setup(rois, n, scale, B, S, H, W, sh);
  if (!sh.s_ok) return;                                     // roi_align3d_fwd_sep_kernel (skip_v3 mode) does this RoI
  const RoiGeom g = sh.sg;
  const int wave = tid >> 6;
  const int HW = H * W;
  const float* fbase = feat + (size_t)g.batch * C * S * HW;
  float* obase = out + (size_t)n * C * 343;
  const int gofs = sh.rng[0] * HW + sh.rng[2] * W + sh.rng[4];
  const size_t cs = (size_t)S * HW;
  // Two channels at a time whenever the wave's LDS slice holds them.
  const int nw = v3_waves(d);
  if (wave >= nw) return;
  const int slice = kSepLdsFloats / nw;
  float* wl = dyn + (size_t)wave * slice;
  if (2 * d.per_ch <= slice) v3_run<true>(fbase, cs, obase, wl, sh.fz, sh.fy, sh.fx, d, HW, W, gofs, c0 + wave, c1, nw);
  else v3_run<false>(fbase, cs, obase, wl, sh.fz, sh.fy, sh.fx, d, HW, W, gofs, c0 + wave, c1, nw);
}

// ------------------------------------------------------------------------------------------------------
// Round 6: the matrix-core form for SMALL sub-volumes.  With the folds, the whole RoIAlign of one RoI is ONE linear map from its
// sub-volume (K = ez ey ex voxels, the same for every channel) to its 343 bins:
//     out[c][bin] = sum_k f[c][k] * M[k][bin],      M[k = (z, y, x)][bin = (ph, pw, ps)] = Wz[ps][z] * Wy[ph][y] * Wx[pw][x]
// (W?[p][t] = the fold's four weights spread over the axis; Wy carries the 1 / 8 of the sample mean) - a [C x K] x [K x 343] GEMM per RoI.
// On the fp32 matrix pipe that was slower than the three LDS passes (round 5: 0.71 ms); with the f16x2 split of fc_gemm.hip (both
// operands scaled and cut into two fp16 numbers, three v_mfma_f32_32x32x16_f16 products per fp32 product: 16 / 3 of the fp32 MFMA
// rate) K <= 128 costs 264 - 528 MFMAs of 32 cycles per wave and RoI, and no LDS pass at all.  Error: <= 7e-7 of max |f| per product
// (the fast mode's contract is 1e-5 max |f|; the exact-order kernel stays the bit-exact one).
//   M dimension = channels (A = features: lane (channel, k half) gathers its 8 voxels of a k16 step straight from the L2-resident map and
//                 cuts them once per RoI; 16 KS registers per wave), N dimension = bins in blocks of 32 (B = the operator, built per bin
//                 block into LDS by the four waves together, double-buffered, one barrier per block), K = voxels in steps of 16.
//   A wave owns C / 128 channel blocks of 32 and walks the 11 bin blocks; an accumulator register holds 32 consecutive bins of one
//   channel across lanes 0 - 31 (and of channel + 4 across lanes 32 - 63): every store instruction writes two 128-byte runs.
// KS = k16 steps (4: sub-volumes <= 64 voxels, 8: <= 128).  Grid: one workgroup per RoI rank; the RoI's class marker decides.
template <int KS>
__global__ __launch_bounds__(256) void roi_align3d_fwd_gemm_kernel(const float* __restrict__ feat, float* __restrict__ out, int B, int C, int S,
                                                                   int H, int W, const int* __restrict__ order, const int* __restrict__ tabs,
                                                                   const float* __restrict__ feat_absmax) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  constexpr int K = 16 * KS;
  constexpr int NCB = 2;                                     // channel blocks of 32 per wave and pass (C = 256: one pass)
  __shared__ V3Shared sh;
  __shared__ float wtab[3][7][16];                           // [axis z, y, x][bin][position]: the folds as dense rows
  __shared__ int koff[K];                                    // voxel k -> offset inside one channel map (relative to the sub-volume origin)
  __shared__ unsigned kzyx[K];                               // voxel k -> z | y << 8 | x << 16 (k >= sub: a position whose weights are zero)
  __shared__ u32x4 bimg[2][KS * 2 * 64];                     // operator image of one bin block: [k16 step][hi / lo][lane]
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, h = l >> 5, nl = l & 31;
  const int n = order ? order[blockIdx.x] : (int)blockIdx.x;
  const unsigned int cls = __float_as_uint(out[(size_t)n * C * 343]);
  if (cls != (KS == 4 ? kGemm4Bits : kGemm8Bits)) return;
  const V3Dims d = v3_load_tab(tabs + (size_t)n * kV3TabWords, sh);
  const RoiGeom g = sh.sg;
  const int HW = H * W;
  // ---- per-RoI tables
  for (int e = tid; e < 3 * 7 * 16; e += 256) (&wtab[0][0][0])[e] = 0.f;
  __syncthreads();
  if (tid < 21) {                                            // one thread per (axis, bin): its four taps (clamped duplicates carry weight 0)
    const int ax = tid / 7, pb = tid % 7;
    const Fold f = ax == 0 ? sh.fz[pb] : (ax == 1 ? sh.fy[pb] : sh.fx[pb]);
#pragma unroll
    for (int i = 0; i < 4; ++i) wtab[ax][pb][f.k[i]] += f.w[i];
  }
  for (int k = tid; k < K; k += 256) {
    const bool in = k < d.sub;
    const int x = in ? k % d.ex : 0, r = in ? k / d.ex : 0;
    const int y = r % d.ey, z = r / d.ey;
    koff[k] = z * HW + y * W + x;
    kzyx[k] = (unsigned)(z | (y << 8) | (x << 16));                            // (k >= sub: never used, the operator is zero there)
  }
  __syncthreads();
  float fs, inv_f;
  m3d::f16_scale_of(*feat_absmax, fs, inv_f);
  constexpr float kMs = 16384.f, kInvMs = 1.f / 16384.f;     // operator scale 2^14: an axis weight is <= 2 (two samples on one voxel), the
                                                             // y row carries 1 / 8: M <= 2 * 2 * 0.25 = 1 -> <= 2^14 in fp16
  const float* fbase = feat + (size_t)g.batch * C * S * HW + (size_t)sh.rng[0] * HW + sh.rng[2] * W + sh.rng[4];
  const size_t cs = (size_t)S * HW;
  float* obase = out + (size_t)n * C * 343;

  // B image of bin block `blk` into bimg[buf]: wave w builds the k16 steps s = w, w + 4 (KS = 8); lane (bin n, k half h)
  auto build_b = [&](int blk, int buf) __attribute__((always_inline)) {
    const int bin = 32 * blk + nl;
    const bool bok = bin < 343;
    const int ph = bok ? bin / 49 : 0, pw = bok ? (bin / 7) % 7 : 0, ps = bok ? bin % 7 : 0;
    const float* wz = wtab[0][ps]; const float* wy = wtab[1][ph]; const float* wx = wtab[2][pw];
#pragma unroll
    for (int s = 0; s < KS / 4; ++s) {
      const int st = wave + 4 * s;
      u32x4 ph4, pl4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f16x2 hh, ll;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = 16 * st + 8 * h + 2 * j + u;
          const unsigned c = kzyx[k];
          float m = (wz[c & 255] * wy[(c >> 8) & 255]) * wx[(c >> 16) & 255];
          m *= (k < d.sub && bok) ? kMs : 0.f;                                 // (a select, not a branch around the table reads)
          const _Float16 hv = (_Float16)m;
          hh[u] = hv; ll[u] = (_Float16)(m - (float)hv);
        }
        ph4[j] = __builtin_bit_cast(unsigned, hh); pl4[j] = __builtin_bit_cast(unsigned, ll);
      }
      bimg[buf][(st * 2 + 0) * 64 + l] = ph4;
      bimg[buf][(st * 2 + 1) * 64 + l] = pl4;
    }
  };

  // every wave runs every pass (the operator image is built by all four and barriers are workgroup-wide); a wave's channel blocks in
  // a pass: `per` = 1 (C <= 128: all four waves have one) or 2; blocks beyond C / 32 only skip their MFMAs and stores
  const int ncb_total = C / 32;
  const int per = ncb_total <= 4 ? 1 : NCB;
  for (int pass0 = 0; pass0 < ncb_total; pass0 += 4 * per) {
    const int cb0 = pass0 + wave * per;
    // ---- A fragments of this wave's channel blocks: gathered and cut once per RoI
    f16x8 ah[NCB][KS], al[NCB][KS];
#pragma unroll
    for (int q = 0; q < NCB; ++q) {
      const int ch = 32 * (cb0 + (q < per ? q : 0)) + nl;
      const float* fc = fbase + (size_t)min(ch, C - 1) * cs;
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fc[koff[16 * st + 8 * h + j]];      // unconditional (k >= sub reads voxel 0: the operator is zero
                                                                                // there): a guarded load waits for its data at the join
        f16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = v[j] * fs;
          const _Float16 hv = (_Float16)t;
          vh[j] = hv; vl[j] = (_Float16)(t - (float)hv);
        }
        ah[q][st] = vh; al[q][st] = vl;
      }
    }
    build_b(0, 0);
    __syncthreads();
    for (int blk = 0; blk < 11; ++blk) {
      if (blk + 1 < 11) build_b(blk + 1, (blk + 1) & 1);              // the next block's operator, under this block's MFMAs
      f32x16 acc[NCB];
#pragma unroll
      for (int q = 0; q < NCB; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
      const u32x4* bi = bimg[blk & 1];
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        const f16x8 bh = __builtin_bit_cast(f16x8, bi[(st * 2 + 0) * 64 + l]);
        const f16x8 bl = __builtin_bit_cast(f16x8, bi[(st * 2 + 1) * 64 + l]);
#pragma unroll
        for (int q = 0; q < NCB; ++q) {
          if (q < per && cb0 + q < ncb_total) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q][st], bh, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][st], bl, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][st], bh, acc[q], 0, 0, 0);
          }
        }
      }
      // accumulator register e: channel row 8 (e / 4) + 4 h + e % 4 of the block, bin 32 blk + nl: 32 consecutive floats per half wave
      const int bin = 32 * blk + nl;
      if (bin < 343) {
#pragma unroll
        for (int q = 0; q < NCB; ++q) {
          if (q < per && cb0 + q < ncb_total) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int ch = 32 * (cb0 + q) + 8 * (e >> 2) + 4 * h + (e & 3);
              obase[(size_t)ch * 343 + bin] = (acc[q][e] * inv_f) * kInvMs;
            }
          }
        }
      }
      __syncthreads();                                                // block blk's image is free, block blk + 1's is complete
    }
  }
}

struct AxisTaps { int lo, hi, n; };        // sub-volume range [lo, hi] and number of taps per bin

template <int kDummy>
__global__ __launch_bounds__(256) void roi_align3d_fwd_sep_kernel(const float* __restrict__ feat, const float* __restrict__ rois,
                                                                  float* __restrict__ out, int B, int C, int S, int H, int W, int AS,
                                                                  int AH, int AW, float scale, int ratio, int ch_per_block,
                                                                  int skip_v3 /* RoIs the v3 kernel handles are skipped */) {
  __shared__ AxisSample tz[kMaxTable], ty[kMaxTable], tx[kMaxTable];
  __shared__ RoiGeom sg;
  __shared__ int rng[6];
  __shared__ int s_v3;
  extern __shared__ float dyn[];
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  if (skip_v3 == 2 &&                                       // v3 marked the RoIs it declined (see there); everyone else is done
      __float_as_uint(out[((size_t)n * C + (size_t)blockIdx.y * ch_per_block) * (AS * AH * AW)]) != kDeclinedBits) return;
  if (tid == 0) sg = roi_geom(rois + 7 * n, scale, AS, AH, AW, ratio, B);
  __syncthreads();
  const RoiGeom g = sg;
  const int nz = AS * g.grid_s, ny = AH * g.grid_h, nx = AW * g.grid_w;
  if (nz > kMaxTable || ny > kMaxTable || nx > kMaxTable) {          // adaptive grid beyond the tables: untabled reference order
    const int c0u = blockIdx.y * ch_per_block;
    roi_untabled_range<false>(feat, out, g, n, c0u, min(C, c0u + ch_per_block), C, S, H, W, AS, AH, AW);
    return;
  }
  for (int t = tid; t < nz; t += 256) tz[t] = make_sample(g.start_s, g.bin_s, t / g.grid_s, t % g.grid_s, g.grid_s, S, -1.0);
  for (int t = tid; t < ny; t += 256) ty[t] = make_sample(g.start_h, g.bin_h, t / g.grid_h, t % g.grid_h, g.grid_h, H, -1.0);
  for (int t = tid; t < nx; t += 256) tx[t] = make_sample(g.start_w, g.bin_w, t / g.grid_w, t % g.grid_w, g.grid_w, W, -1.0);
  __syncthreads();
  if (tid < 3) {
    const AxisSample* t = tid == 0 ? tz : (tid == 1 ? ty : tx);
    const int cnt = tid == 0 ? nz : (tid == 1 ? ny : nx);
    int lo = 1 << 30, hi = -1;
    for (int i = 0; i < cnt; ++i)
      if (t[i].valid) { lo = min(lo, t[i].lo); hi = max(hi, t[i].hi); }
    rng[2 * tid] = lo; rng[2 * tid + 1] = hi;
  }
  __syncthreads();
  const int bins = AS * AH * AW;
  const int c0 = blockIdx.y * ch_per_block, c1 = min(C, c0 + ch_per_block);
  if (rng[1] < 0 || rng[3] < 0 || rng[5] < 0) {          // no valid sample on some axis: everything is 0
    for (int e = tid; e < (c1 - c0) * bins; e += 256) out[((size_t)n * C + c0) * bins + e] = 0.f;
    return;
  }
  if (skip_v3) {                                          // same predicate as roi_align3d_fwd_v3_kernel: exactly one of the two runs
    if (tid == 0) s_v3 = 1;
    __syncthreads();
    if (tid < 21) {                                       // v3_qualifies, its 21 (axis, bin) folds in parallel
      const V3Dims d = v3_dims(rng);
      const int ax = tid / 7, p = tid % 7;
      const bool ok = ax == 0 ? fold_bin(tz[2 * p], tz[2 * p + 1], rng[0], d.ez, 1.f, nullptr)
                    : ax == 1 ? fold_bin(ty[2 * p], ty[2 * p + 1], rng[2], d.ey, 1.f, nullptr)
                              : fold_bin(tx[2 * p], tx[2 * p + 1], rng[4], d.ex, 1.f, nullptr);
      if (!ok || d.per_ch > kSepLdsFloats) atomicAnd(&s_v3, 0);
    }
    __syncthreads();
    if (s_v3) return;
  }
  const int z0 = rng[0], y0 = rng[2], x0 = rng[4];
  const int ez = rng[1] - z0 + 1, ey = rng[3] - y0 + 1, ex = rng[5] - x0 + 1;
  const int sub = ez * ey * ex, n1 = ez * ey * AW, n2 = ez * AH * AW;
  const int per_ch = sub + n1 + n2;
  const float inv_count = 1.0f / (float)(g.grid_s * g.grid_h * g.grid_w);
  const int HW = H * W;
  // one pass set over elements [first, first+stride, ...) of one channel held in (fsub, t1, t2)
  auto passes = [&](const float* fsub, float* t1, float* t2, float* oc, int first, int stride, bool block_sync)
                    __attribute__((always_inline)) {
    for (int e = first; e < n1; e += stride) {           // pass X: t1[z][y][pw] = sum_samples h*f[lo] + l*f[hi]
      const int pw = e % AW, zy = e / AW;
      const float* row = fsub + zy * ex - x0;
      float acc = 0.f;
      for (int i = 0; i < g.grid_w; ++i) {
        const AxisSample sm = tx[pw * g.grid_w + i];
        if (sm.valid) acc += sm.h * row[sm.lo] + sm.l * row[sm.hi];
      }
      t1[e] = acc;
    }
    if (block_sync) __syncthreads();
    for (int e = first; e < n2; e += stride) {           // pass Y: t2[z][ph][pw]
      const int pw = e % AW, ph = (e / AW) % AH, z = e / (AW * AH);
      const float* col = t1 + (z * ey - y0) * AW + pw;
      float acc = 0.f;
      for (int i = 0; i < g.grid_h; ++i) {
        const AxisSample sm = ty[ph * g.grid_h + i];
        if (sm.valid) acc += sm.h * col[sm.lo * AW] + sm.l * col[sm.hi * AW];
      }
      t2[e] = acc;
    }
    if (block_sync) __syncthreads();
    for (int e = first; e < bins; e += stride) {         // pass Z + average, output memory order (ph, pw, ps)
      const int ps = e % AS, pw = (e / AS) % AW, ph = e / AS / AW;
      const float* col = t2 - z0 * AH * AW + ph * AW + pw;
      float acc = 0.f;
      for (int i = 0; i < g.grid_s; ++i) {
        const AxisSample sm = tz[ps * g.grid_s + i];
        if (sm.valid) acc += sm.h * col[sm.lo * AH * AW] + sm.l * col[sm.hi * AH * AW];
      }
      oc[e] = acc * inv_count;
    }
  };
  const int wave = tid >> 6, lane = tid & 63;
  constexpr int kSlice = kSepLdsFloats / 4;                  // LDS floats of one wave
  // register-tap paths (7x7x7 bins, sampling grid 2 - the shipped configs): every lane keeps a FIXED (ph, pw) for the whole RoI, so
  // all interpolation taps, weights and LDS offsets are precomputed once per RoI; per channel the three passes are bare
  // {read x4, fma x4, write} bodies.  Every WAVE works alone on one channel at a time in its own LDS slice (LDS operations of one
  // wave execute in order, so a pass may read what other lanes of the same wave wrote: no workgroup barriers), and the 343 results
  // of a channel leave through LDS as six fully coalesced 256-byte stores instead of 7 x 49 scattered dwords.
  //   staged:   the channel's sub-volume is copied into LDS (next channel's copy is fetched into registers meanwhile)
  //   global-x: sub-volumes too large for that: the x pass reads its taps straight from the (L2-resident) feature map
  const bool reg_taps = (g.grid_s == 2) & (g.grid_h == 2) & (g.grid_w == 2) & (AS == 7) & (AH == 7) & (AW == 7);
  const bool staged = sub <= kSepStageMax && per_ch + 343 <= kSlice;
  const bool globalx = !staged && reg_taps && n1 + n2 + 343 <= kSlice;
  if (!(staged || globalx)) {
    if (per_ch > kSepLdsFloats) {                          // sub-volume and intermediates do not fit LDS at all: reference order
      roi_exact_forward_range(feat, out, g, tz, ty, tx, n, c0, c1, C, S, H, W, AS, AH, AW);
      return;
    }
    // the whole workgroup cooperates on one channel at a time (block barriers between the passes)
    float* fsub = dyn; float* t1 = fsub + sub; float* t2 = t1 + n1;
    for (int c = c0; c < c1; ++c) {
      const float* fc = feat + ((size_t)g.batch * C + c) * S * HW + (size_t)z0 * HW + y0 * W + x0;
      __syncthreads();
      for (int e = tid; e < sub; e += 256) fsub[e] = fc[(size_t)(e / (ey * ex)) * HW + ((e / ex) % ey) * W + e % ex];
      __syncthreads();
      passes(fsub, t1, t2, out + ((size_t)n * C + c) * bins, tid, 256, true);
    }
    return;
  }
  float* wl = dyn + (size_t)wave * kSlice;
  float* fsub = wl;                                        // [ez][ey][ex]   (staged only)
  float* t1 = staged ? fsub + sub : wl;                    // [ez][ey][AW]
  float* t2 = t1 + n1;                                     // [ez][AH][AW]
  float* obuf = t2 + n2;                                   // [343] results of one channel in output order
  constexpr int kMaxStage = kSepStageMax / 64;             // registers per lane that can hold one staged sub-volume
  float stage[kMaxStage];
  int soff[kMaxStage];
#pragma unroll
  for (int i = 0; i < kMaxStage; ++i) {
    const int e = lane + i * 64;
    soff[i] = -1;
    if (staged && i * 64 < sub && e < sub) soff[i] = (e / (ey * ex)) * HW + ((e / ex) % ey) * W + e % ex;   // wave-uniform skip
  }
  if (!reg_taps) {                                         // generic grids: table-driven passes, staged sub-volume
    auto fetch = [&](int c) __attribute__((always_inline)) {
      const float* fc = feat + ((size_t)g.batch * C + c) * S * HW + (size_t)z0 * HW + y0 * W + x0;
#pragma unroll
      for (int i = 0; i < kMaxStage; ++i) stage[i] = fc[soff[i] < 0 ? 0 : soff[i]];
    };
    if (c0 + wave < c1) fetch(c0 + wave);
    for (int c = c0 + wave; c < c1; c += 4) {
#pragma unroll
      for (int i = 0; i < kMaxStage; ++i)
        if (soff[i] >= 0) fsub[lane + i * 64] = stage[i];
      if (c + 4 < c1) fetch(c + 4);
      passes(fsub, t1, t2, out + ((size_t)n * C + c) * bins, lane, 64, false);
    }
    return;
  }
  const int my_pw = lane % 7, my_rx = lane / 7, my_ph = (lane / 7) % 7;
  float xw[4], yw[4], zw[7][4];
  int xo[4], yo[4], zo[7][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const AxisSample sx = tx[my_pw * 2 + i], sy = ty[my_ph * 2 + i];
    xw[2 * i] = sx.valid ? sx.h : 0.f; xw[2 * i + 1] = sx.valid ? sx.l : 0.f;
    xo[2 * i] = sx.valid ? sx.lo - x0 : 0; xo[2 * i + 1] = sx.valid ? sx.hi - x0 : 0;
    yw[2 * i] = sy.valid ? sy.h : 0.f; yw[2 * i + 1] = sy.valid ? sy.l : 0.f;
    yo[2 * i] = sy.valid ? (sy.lo - y0) * 7 : 0; yo[2 * i + 1] = sy.valid ? (sy.hi - y0) * 7 : 0;
  }
#pragma unroll
  for (int ps = 0; ps < 7; ++ps)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const AxisSample sz = tz[ps * 2 + i];
      zw[ps][2 * i] = sz.valid ? sz.h * inv_count : 0.f; zw[ps][2 * i + 1] = sz.valid ? sz.l * inv_count : 0.f;
      zo[ps][2 * i] = sz.valid ? (sz.lo - z0) * 49 : 0; zo[ps][2 * i + 1] = sz.valid ? (sz.hi - z0) * 49 : 0;
    }
  const int nrows = ez * ey;
  const int nst = staged ? (sub + 63) / 64 : 0;            // staging registers actually needed (wave-uniform)
  auto fetch = [&](int c) __attribute__((always_inline)) {
    const float* fc = feat + ((size_t)g.batch * C + c) * S * HW + (size_t)z0 * HW + y0 * W + x0;
#pragma unroll
    for (int i = 0; i < kMaxStage; ++i) if (i < nst) stage[i] = fc[soff[i] < 0 ? 0 : soff[i]];
  };
  if (staged && c0 + wave < c1) fetch(c0 + wave);
  // global-x: this lane's first row (z, y) of the sub-volume and its step of 9 rows
  const int gz0 = my_rx / ey, gy0 = my_rx % ey;
  for (int c = c0 + wave; c < c1; c += 4) {
    if (staged) {
#pragma unroll
      for (int i = 0; i < kMaxStage; ++i)
        if (i < nst && soff[i] >= 0) fsub[lane + i * 64] = stage[i];
      if (c + 4 < c1) fetch(c + 4);
      if (my_rx < 9) {                                     // pass X: lanes 0..62 <-> (row mod 9, pw)
        const float* row = fsub + my_rx * ex;
        float* dst = t1 + my_rx * 7 + my_pw;
        for (int zy = my_rx; zy < nrows; zy += 9) {
          *dst = (xw[0] * row[xo[0]] + xw[1] * row[xo[1]]) + (xw[2] * row[xo[2]] + xw[3] * row[xo[3]]);
          row += 9 * ex; dst += 63;
        }
      }
    } else if (my_rx < 9) {                                // pass X from the feature map itself
      const float* fc = feat + ((size_t)g.batch * C + c) * S * HW + (size_t)z0 * HW + y0 * W + x0;
      float* dst = t1 + my_rx * 7 + my_pw;
      int z = gz0, y = gy0;
      for (int zy = my_rx; zy < nrows; zy += 9) {
        const float* row = fc + (size_t)z * HW + y * W;
        *dst = (xw[0] * row[xo[0]] + xw[1] * row[xo[1]]) + (xw[2] * row[xo[2]] + xw[3] * row[xo[3]]);
        dst += 63;
        y += 9;
        while (y >= ey) { y -= ey; ++z; }
      }
    }
    if (lane < 49) {
      const float* col = t1 + my_pw;                       // pass Y: lane <-> (ph, pw), loop over z
      float* dst = t2 + lane;
      for (int z = 0; z < ez; ++z) {
        *dst = (yw[0] * col[yo[0]] + yw[1] * col[yo[1]]) + (yw[2] * col[yo[2]] + yw[3] * col[yo[3]]);
        col += ey * 7; dst += 49;
      }
      const float* cz = t2 + lane;                         // pass Z (+ 1/count folded into the weights) -> obuf in output order
      float* ob = obuf + lane * 7;
#pragma unroll
      for (int ps = 0; ps < 7; ++ps)
        ob[ps] = (zw[ps][0] * cz[zo[ps][0]] + zw[ps][1] * cz[zo[ps][1]]) + (zw[ps][2] * cz[zo[ps][2]] + zw[ps][3] * cz[zo[ps][3]]);
    }
    float* oc = out + ((size_t)n * C + c) * 343;           // 343 contiguous floats: 6 coalesced stores
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int e = lane + 64 * k;
      if (e < 343) oc[e] = obuf[e];
    }
  }
}

// ---- f-1: per-RoI tap tables for a GEMM whose A-operand loader performs the RoIAlign gather (fc_gemm.hip).
// For the shipped geometry (7^3 bins, sampling grid 2) a RoI has 14 samples per axis; sample s contributes to bin s / 2 through its
// two corners (lo, hi) with weights (h, l) - zero when the coordinate is out of range (roi_align_kernel_3d.cu:19-22).  Entry
// tab[r][axis][s] = (lo, hi, bits(h * valid), bits(l * valid)), axis 0 = z (bin ps), 1 = y (ph), 2 = x (pw).
__global__ __launch_bounds__(64) void roi_tap_table_kernel(const float* __restrict__ rois, int R, float scale, int B, int S, int H, int W,
                                                           int4* __restrict__ tab, int* __restrict__ batch) {
  const int r = blockIdx.x, t = threadIdx.x;
  if (r >= R || t >= 42) return;
  const RoiGeom g = roi_geom(rois + 7 * r, scale, 7, 7, 7, 2, B);
  const int ax = t / 14, s = t % 14;
  const float start = ax == 0 ? g.start_s : (ax == 1 ? g.start_h : g.start_w);
  const float bin = ax == 0 ? g.bin_s : (ax == 1 ? g.bin_h : g.bin_w);
  const int dim = ax == 0 ? S : (ax == 1 ? H : W);
  const AxisSample a = make_sample(start, bin, s / 2, s % 2, 2, dim, -1.0);
  tab[(size_t)r * 42 + t] = make_int4(a.lo, a.hi, __float_as_int(a.valid ? a.h : 0.f), __float_as_int(a.valid ? a.l : 0.f));
  if (t == 0) batch[r] = g.batch;
}

int launch(int mode /*0 fast fwd, 1 exact fwd, 2 backward*/, int AS, int AH, int AW, float scale, int ratio, const float* a, const float* rois, float* o, int B,
           int C, int S, int H, int W, int R, int roi_cols, void* stream, void* ws = nullptr, size_t ws_bytes = 0,
           const float* feat_absmax = nullptr) {
  if (roi_cols != 7) return M3D_EINVAL;   // roi_align_cuda_3d.c:19-22
  if (R < 0 || B <= 0 || C <= 0 || S <= 0 || H <= 0 || W <= 0 || AS <= 0 || AH <= 0 || AW <= 0) return M3D_EINVAL;
  if (R == 0) return M3D_OK;
  if (!a || !rois || !o) return M3D_EINVAL;
  if (ratio > 0 && (AS * ratio > kMaxTable || AH * ratio > kMaxTable || AW * ratio > kMaxTable)) return M3D_EUNSUPPORTED;
  // enough workgroups to fill 256 CUs several times over; at least 8 channels per workgroup so the LDS
  // tables are amortised.
  // one workgroup per (RoI, channel chunk): the per-RoI setup (sample tables, tap registers) is amortised over the
  // chunk, so chunks are as large as possible while the grid still holds >= ~2048 workgroups (8 per CU)
  int chunks = 1;
  while ((long)R * chunks < 2048 && chunks * 16 < C) chunks *= 2;
  int cpb = (C + chunks - 1) / chunks;
  chunks = (C + cpb - 1) / cpb;
  dim3 grid(R, chunks), block(256);
  const bool backward = (mode == 2);
  if (mode == 0) {
    const size_t lds = sizeof(float) * kSepLdsFloats;
    const int v3 = (ratio == 2 && AS == 7 && AH == 7 && AW == 7) ? 1 : 0;      // the shipped geometry: two launches, each RoI in one
    // complement pass (v3: only the RoIs v3 declined - wide bins, huge sub-volumes).  Those RoIs are few and heavy (45 us each on
    // one workgroup: soma tile, 25 of 274 RoIs = 1.1 ms), so their channels are cut into enough chunks for ~4096 workgroups in all.
    int cchunks = 1;
    while ((long)R * cchunks < 4096 && cchunks * 8 < C) cchunks *= 2;
    const int ccpb = (C + cchunks - 1) / cchunks;
    // the per-RoI work split travels through markers in the output (every 8th channel); usable when all chunk starts fall there
    const int marks = (C % 32 == 0 && ccpb % 8 == 0) ? 1 : 0;
    if (v3 && marks) {              // work split per RoI through markers in the output (roi_class_kernel)
      // workspace: [R] launch order (heaviest RoI first) + [R][kV3TabWords] set-up records (see v3_store_tab)
      int* order = (ws && ws_bytes >= sizeof(int) * (size_t)R * (1 + kV3TabWords)) ? reinterpret_cast<int*>(ws) : nullptr;
      int* tabs = order ? order + R : nullptr;
      // with the feature maps' largest magnitude (and the set-up records): small sub-volumes take the matrix-core form
      const int gemm = (feat_absmax && tabs) ? 1 : 0;
      hipLaunchKernelGGL(roi_class_kernel, dim3(R), block, 0, m3d::as_stream(stream), rois, o, B, C, S, H, W, scale, R, order, tabs, gemm);
      if (gemm) {
        hipLaunchKernelGGL(roi_align3d_fwd_gemm_kernel<4>, dim3(R), block, 0, m3d::as_stream(stream), a, o, B, C, S, H, W, (const int*)order,
                           (const int*)tabs, feat_absmax);
        hipLaunchKernelGGL(roi_align3d_fwd_gemm_kernel<8>, dim3(R), block, 0, m3d::as_stream(stream), a, o, B, C, S, H, W, (const int*)order,
                           (const int*)tabs, feat_absmax);
      }
      // option tune_roi_xcd = 1 (tuning build; A/B and the PMC passes of profiles/r06_roi_xcd_ab.txt): the XCD-aware channel split
      // described in the kernel.  It removes the sub-volume reads' L2 misses and leaves the time where it was (0.237 vs 0.225 ms at
      // R = 1281: 8 set-ups per small RoI instead of one) - the launch is bound by LDS array cycles, not by those fetches - so the
      // release library keeps one workgroup per small RoI.
      if (m3d::opt(m3d::OPT_TUNE_ROI_XCD) == 1 && C % 64 == 0 && (long long)R * (C / 8) < 0x7FFFFFFFll)
        hipLaunchKernelGGL(roi_align3d_fwd_v3_kernel, dim3((unsigned)(R * (C / 8))), block, lds, m3d::as_stream(stream), a, rois, o, B, C, S,
                           H, W, scale, 8, 2, (const int*)order, (const int*)tabs);
      else
        hipLaunchKernelGGL(roi_align3d_fwd_v3_kernel, dim3(R, C / 8), block, lds, m3d::as_stream(stream), a, rois, o, B, C, S, H, W, scale,
                           8, 1, (const int*)order, (const int*)tabs);
    } else if (v3) {                // 32 channels per workgroup
      const int cpb3 = 32;
      hipLaunchKernelGGL(roi_align3d_fwd_v3_kernel, dim3(R, (C + cpb3 - 1) / cpb3), block, lds, m3d::as_stream(stream), a, rois, o, B, C, S,
                         H, W, scale, cpb3, 0, (const int*)nullptr, (const int*)nullptr);
    }
    hipLaunchKernelGGL(roi_align3d_fwd_sep_kernel<0>, v3 ? dim3(R, (C + ccpb - 1) / ccpb) : grid, block, lds, m3d::as_stream(stream), a,
                       rois, o, B, C, S, H, W, AS, AH, AW, scale, ratio, v3 ? ccpb : cpb, v3 ? (marks ? 2 : 1) : 0);
  } else if (!backward)
    hipLaunchKernelGGL(roi_align3d_kernel<false>, grid, block, 0, m3d::as_stream(stream), a, rois, o, B, C, S, H, W, AS, AH, AW,
                       scale, ratio, cpb, (int*)nullptr);
  else
    hipLaunchKernelGGL(roi_align3d_kernel<true>, grid, block, 0, m3d::as_stream(stream), a, rois, o, B, C, S, H, W, AS, AH, AW,
                       scale, ratio, cpb, (int*)nullptr);
  return m3d::check_launch("roi_align3d");
}

}  // namespace

M3D_API int m3d_roi_align3d_forward(int AS, int AH, int AW, float spatial_scale, int sampling_ratio, const float* d_features,
                                    int batch, int channels, int slices, int height, int width, const float* d_rois,
                                    int num_rois, int roi_cols, float* d_output, void* stream) {
  return launch(0, AS, AH, AW, spatial_scale, sampling_ratio, d_features, d_rois, d_output, batch, channels, slices, height,
                width, num_rois, roi_cols, stream);
}

/* The same forward with a caller workspace of m3d_roi_align3d_workspace_bytes(num_rois): the launch then takes the RoIs in descending
 * order of their work (sub-volume size) instead of index order - identical results (each output row is written by the same arithmetic),
 * a shorter tail.  A null / too small workspace gives the index order. */
M3D_API size_t m3d_roi_align3d_workspace_bytes(int num_rois) { return num_rois > 0 ? sizeof(int) * (size_t)num_rois * (1 + kV3TabWords) : 0; }

M3D_API int m3d_roi_align3d_forward_ws(int AS, int AH, int AW, float spatial_scale, int sampling_ratio, const float* d_features,
                                       int batch, int channels, int slices, int height, int width, const float* d_rois,
                                       int num_rois, int roi_cols, float* d_output, void* d_ws, size_t ws_bytes, void* stream) {
  return launch(0, AS, AH, AW, spatial_scale, sampling_ratio, d_features, d_rois, d_output, batch, channels, slices, height,
                width, num_rois, roi_cols, stream, d_ws, ws_bytes);
}

/* Round 6: ... and with d_feat_absmax, a device pointer to ONE float >= max |d_features| (m3d_absmax): RoIs whose sub-volume has <= 128
 * voxels then run as one [channels x K] x [K x 343] GEMM on the f16 matrix cores (two scaled fp16 pieces per operand, three products;
 * <= 7e-7 max |f| from the separable kernel's result, inside the fast mode's 1e-5 max |f|); the others as before.  NULL: as _ws. */
M3D_API int m3d_roi_align3d_forward_ws2(int AS, int AH, int AW, float spatial_scale, int sampling_ratio, const float* d_features,
                                        int batch, int channels, int slices, int height, int width, const float* d_rois,
                                        int num_rois, int roi_cols, float* d_output, void* d_ws, size_t ws_bytes,
                                        const float* d_feat_absmax, void* stream) {
  return launch(0, AS, AH, AW, spatial_scale, sampling_ratio, d_features, d_rois, d_output, batch, channels, slices, height,
                width, num_rois, roi_cols, stream, d_ws, ws_bytes, d_feat_absmax);
}

M3D_API int m3d_roi_align3d_backward(int AS, int AH, int AW, float spatial_scale, int sampling_ratio, const float* d_top_grad,
                                     const float* d_rois, int num_rois, int roi_cols, float* d_bottom_grad, int batch,
                                     int channels, int slices, int height, int width, void* stream) {
  return launch(2, AS, AH, AW, spatial_scale, sampling_ratio, d_top_grad, d_rois, d_bottom_grad, batch, channels, slices,
                height, width, num_rois, roi_cols, stream);
}

M3D_API int m3d_roi_align3d_forward_exact(int AS, int AH, int AW, float spatial_scale, int sampling_ratio,
                                          const float* d_features, int batch, int channels, int slices, int height, int width,
                                          const float* d_rois, int num_rois, int roi_cols, float* d_output, void* stream) {
  return launch(1, AS, AH, AW, spatial_scale, sampling_ratio, d_features, d_rois, d_output, batch, channels, slices, height,
                width, num_rois, roi_cols, stream);
}

/* f-1 A/B: tap tables of the RoIs for m3d_linear_bf16x3_roi_forward, the fc1 GEMM whose operand loader performs the
 * RoIAlign gather (7^3 bins, sampling grid 2 only).  d_tab: int4 [num_rois, 3, 14]; d_batch: int32 [num_rois]. */
M3D_API int m3d_roi_align3d_tap_tables(const float* d_rois, int num_rois, float spatial_scale, int batch, int slices, int height, int width,
                                       void* d_tab, int32_t* d_batch, void* stream) {
  if (num_rois < 0 || batch <= 0 || slices <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_rois || !d_tab || !d_batch) return M3D_EINVAL;
  hipLaunchKernelGGL(roi_tap_table_kernel, dim3(num_rois), dim3(64), 0, m3d::as_stream(stream), d_rois, num_rois, spatial_scale, batch, slices,
                     height, width, (int4*)d_tab, d_batch);
  return m3d::check_launch("roi_align3d_tap_tables");
}
