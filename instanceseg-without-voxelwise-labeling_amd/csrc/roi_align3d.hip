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
#include "m3d_common.h"

namespace {

struct AxisSample {  // one (bin, sub-sample) along one axis
  int lo, hi;        // clamped corner indices
  float l, h;        // weights of hi / lo corners (l = frac, h = 1 - frac)
  int valid;         // 0 => whole sample contributes 0 (coordinate outside [-1, dim])
};

constexpr int kMaxTable = 256;  // A * grid per axis (7 bins * adaptive grid <= 36 fits comfortably)

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

__device__ inline RoiGeom roi_geom(const float* r, float scale, int AS, int AH, int AW, int ratio) {
  RoiGeom g;
  g.batch = (int)r[0];                                                  // :94
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

// grid = (num_rois, channel_chunks); block = 256
template <bool kBackward>
__global__ __launch_bounds__(256) void roi_align3d_kernel(const float* __restrict__ feat_or_top, const float* __restrict__ rois,
                                                          float* __restrict__ out_or_grad, int C, int S, int H, int W,
                                                          int AS, int AH, int AW, float scale, int ratio, int ch_per_block,
                                                          int* __restrict__ status) {
  __shared__ AxisSample tz[kMaxTable], ty[kMaxTable], tx[kMaxTable];
  __shared__ RoiGeom sg;
  const int n = blockIdx.x;
  if (threadIdx.x == 0) sg = roi_geom(rois + 7 * n, scale, AS, AH, AW, ratio);
  __syncthreads();
  const RoiGeom g = sg;
  if (AS * g.grid_s > kMaxTable || AH * g.grid_h > kMaxTable || AW * g.grid_w > kMaxTable) {
    if (threadIdx.x == 0 && status) atomicExch(status, 1);   // reported by the host wrapper on request
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
  const float count = (float)(g.grid_s * g.grid_h * g.grid_w);         // :126
  const int HW = H * W;
  const int total = (c1 - c0) * bins;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int c = c0 + e / bins;
    const int b = e % bins;
    // flat bin index in the reference's thread order: ps fastest, then pw, then ph (:87-89)
    const int ps = b % AS;
    const int pw = (b / AS) % AW;
    const int ph = b / AS / AW;
    if (!kBackward) {
      const float* data = feat_or_top + ((size_t)g.batch * C + c) * S * HW;   // :112-113
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
              val = w1 * p00[x.lo];                                   // :76 left-to-right
              val = val + w2 * p00[x.hi];
              val = val + w3 * p01[x.lo];
              val = val + w4 * p01[x.hi];
              val = val + w5 * p10[x.lo];
              val = val + w6 * p10[x.hi];
              val = val + w7 * p11[x.lo];
              val = val + w8 * p11[x.hi];
            }
            acc += val;                                               // :142
          }
        }
      }
      acc /= count;                                                   // :147
      out_or_grad[((size_t)n * C + c) * bins + b] = acc;              // :149 (index == (n,c,ph,pw,ps))
    } else {
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

int launch(bool backward, int AS, int AH, int AW, float scale, int ratio, const float* a, const float* rois, float* o, int B,
           int C, int S, int H, int W, int R, int roi_cols, void* stream) {
  if (roi_cols != 7) return M3D_EINVAL;   // roi_align_cuda_3d.c:19-22
  if (R < 0 || B <= 0 || C <= 0 || S <= 0 || H <= 0 || W <= 0 || AS <= 0 || AH <= 0 || AW <= 0) return M3D_EINVAL;
  if (R == 0) return M3D_OK;
  if (!a || !rois || !o) return M3D_EINVAL;
  if (ratio > 0 && (AS * ratio > kMaxTable || AH * ratio > kMaxTable || AW * ratio > kMaxTable)) return M3D_EUNSUPPORTED;
  // enough workgroups to fill 256 CUs several times over; at least 8 channels per workgroup so the LDS
  // tables are amortised.
  int chunks = 1;
  while ((long)R * chunks < 4096 && chunks * 8 < C) chunks *= 2;
  int cpb = (C + chunks - 1) / chunks;
  chunks = (C + cpb - 1) / cpb;
  dim3 grid(R, chunks), block(256);
  if (!backward)
    hipLaunchKernelGGL(roi_align3d_kernel<false>, grid, block, 0, m3d::as_stream(stream), a, rois, o, C, S, H, W, AS, AH, AW,
                       scale, ratio, cpb, (int*)nullptr);
  else
    hipLaunchKernelGGL(roi_align3d_kernel<true>, grid, block, 0, m3d::as_stream(stream), a, rois, o, C, S, H, W, AS, AH, AW,
                       scale, ratio, cpb, (int*)nullptr);
  return m3d::check_launch("roi_align3d");
}

}  // namespace

M3D_API int m3d_roi_align3d_forward(int AS, int AH, int AW, float spatial_scale, int sampling_ratio, const float* d_features,
                                    int batch, int channels, int slices, int height, int width, const float* d_rois,
                                    int num_rois, int roi_cols, float* d_output, void* stream) {
  return launch(false, AS, AH, AW, spatial_scale, sampling_ratio, d_features, d_rois, d_output, batch, channels, slices, height,
                width, num_rois, roi_cols, stream);
}

M3D_API int m3d_roi_align3d_backward(int AS, int AH, int AW, float spatial_scale, int sampling_ratio, const float* d_top_grad,
                                     const float* d_rois, int num_rois, int roi_cols, float* d_bottom_grad, int batch,
                                     int channels, int slices, int height, int width, void* stream) {
  return launch(true, AS, AH, AW, spatial_scale, sampling_ratio, d_top_grad, d_rois, d_bottom_grad, batch, channels, slices,
                height, width, num_rois, roi_cols, stream);
}
