// PRM post-processing on device: uint8 quantisation of peak response maps and the per-RoI intensity / PRM
// normalisation that feeds the 2D-Otsu kernel.
// Reference: tools/infer_simple.py:233-238 (per-map (fm - min) / max * 255 -> uint8, float32 arithmetic);
// tools/binarization_soma.py:78-91 and tools/binarization_nuclei.py:98-121 (box crop + normalisation, float64
// arithmetic, np.round = round-half-to-even, astype(uint16) = truncation).  Compiled with -ffp-contract=off:
// the fp32 / fp64 operation order is the contract.
#include "m3d_common.h"

namespace {

__device__ inline float block_reduce_min(float v, float* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = fminf(fminf(sm[0], sm[1]), fminf(sm[2], sm[3]));
  __syncthreads();
  return r;
}
__device__ inline float block_reduce_max(float v, float* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  __syncthreads();
  return r;
}

// one workgroup per map: fm -= min(fm); fm /= max(fm); fm *= 255; uint8 (infer_simple.py:234-238)
__global__ __launch_bounds__(256) void prm_quantize_kernel(const float* __restrict__ prm, long long n, uint8_t* __restrict__ out) {
  __shared__ float sm[4];
  const float* p = prm + (size_t)blockIdx.x * n;
  uint8_t* o = out + (size_t)blockIdx.x * n;
  float mn = INFINITY;
  for (long long i = threadIdx.x; i < n; i += 256) mn = fminf(mn, p[i]);
  mn = block_reduce_min(mn, sm);
  float mx = -INFINITY;
  for (long long i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, p[i] - mn);
  mx = block_reduce_max(mx, sm);
  for (long long i = threadIdx.x; i < n; i += 256) {
    float v = p[i] - mn;
    v = v / mx;
    v = v * 255.f;
    o[i] = (uint8_t)v;                       // astype(np.uint8): truncation (values in [0, 255])
  }
}

struct RoiBox { int x1, y1, z1, x2, y2, z2; };   // inclusive crop [z1..z2] x [y1..y2] x [x1..x2] inside the tile

// one workgroup per RoI.  image: uint16 tile [D,H,W]; prm: uint8 maps [P,D,H,W] (map index = roi index).
// mode 0 = soma (binarization_soma.py:85-91), 1 = nuclei (binarization_nuclei.py:110-121).
__global__ __launch_bounds__(256) void roi_normalize_kernel(const uint16_t* __restrict__ image, const uint8_t* __restrict__ prm,
                                                            const int* __restrict__ boxes, const int64_t* __restrict__ offsets,
                                                            int D, int H, int W, int mode, uint16_t* __restrict__ out_img,
                                                            uint16_t* __restrict__ out_prm) {
  __shared__ float sm[4];
  const int r = blockIdx.x;
  const int x1 = boxes[6 * r], y1 = boxes[6 * r + 1], z1 = boxes[6 * r + 2], x2 = boxes[6 * r + 3], y2 = boxes[6 * r + 4],
            z2 = boxes[6 * r + 5];
  const int ex = x2 - x1 + 1, ey = y2 - y1 + 1, ez = z2 - z1 + 1;
  const long long V = (long long)ex * ey * ez;
  if (V <= 0 || offsets[r + 1] - offsets[r] != V) return;
  const uint8_t* pm = prm + (size_t)r * D * H * W;
  uint16_t* oi = out_img + offsets[r];
  uint16_t* op = out_prm + offsets[r];
  float gmax = 0.f, gmin = 65535.f, pmax = 0.f, pmin = 255.f;
  for (long long e = threadIdx.x; e < V; e += 256) {
    const int x = x1 + (int)(e % ex), y = y1 + (int)((e / ex) % ey), z = z1 + (int)(e / ((long long)ex * ey));
    const size_t idx = ((size_t)z * H + y) * W + x;
    const float g = (float)image[idx], p = (float)pm[idx];
    gmax = fmaxf(gmax, g); gmin = fminf(gmin, g); pmax = fmaxf(pmax, p); pmin = fminf(pmin, p);
  }
  gmax = block_reduce_max(gmax, sm); gmin = block_reduce_min(gmin, sm);
  pmax = block_reduce_max(pmax, sm); pmin = block_reduce_min(pmin, sm);
  const double gmaxd = (double)gmax, pmaxd = (double)pmax, pmind = (double)pmin;
  if (mode == 0) {
    for (long long e = threadIdx.x; e < V; e += 256) {
      const int x = x1 + (int)(e % ex), y = y1 + (int)((e / ex) % ey), z = z1 + (int)(e / ((long long)ex * ey));
      const size_t idx = ((size_t)z * H + y) * W + x;
      double f = (double)image[idx] / gmaxd * 300.0;                  // :86-87
      f = f < 0.0 ? 0.0 : (f > 300.0 ? 300.0 : f);                    // np.clip
      oi[e] = (uint16_t)(f + 30.0);                                   // astype(np.uint16)
      const double q = (double)pm[idx] / pmaxd * 300.0 + 30.0;        // :90-91
      op[e] = (uint16_t)rint(q);                                      // np.round (half to even)
    }
    return;
  }
  // nuclei: stretch to 400 levels if the grey range is < 400 (:114-117), PRM mapped into [gray_min, gray_max] (:118-121)
  const bool stretch = ((int)gmax - (int)gmin + 1) < 400;
  float g2max = 0.f, g2min = 65535.f;
  for (long long e = threadIdx.x; e < V; e += 256) {
    const int x = x1 + (int)(e % ex), y = y1 + (int)((e / ex) % ey), z = z1 + (int)(e / ((long long)ex * ey));
    const size_t idx = ((size_t)z * H + y) * W + x;
    uint16_t v = image[idx];
    if (stretch) v = (uint16_t)((uint16_t)((double)v / gmaxd * 400.0) + (uint16_t)gmin);   // uint16 + uint16
    oi[e] = v;
    g2max = fmaxf(g2max, (float)v); g2min = fminf(g2min, (float)v);
  }
  g2max = block_reduce_max(g2max, sm); g2min = block_reduce_min(g2min, sm);
  const double span = (double)(uint16_t)((uint16_t)g2max - (uint16_t)g2min), base = (double)g2min;
  for (long long e = threadIdx.x; e < V; e += 256) {
    const int x = x1 + (int)(e % ex), y = y1 + (int)((e / ex) % ey), z = z1 + (int)(e / ((long long)ex * ey));
    const size_t idx = ((size_t)z * H + y) * W + x;
    const double q = ((double)pm[idx] - pmind) / (pmaxd - pmind) * span + base;
    op[e] = (uint16_t)rint(q);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 2: the same two steps, element-parallel.  (One workgroup per map read 839 MB of dense float maps at 105 GB/s: 8 ms of a
// soma tile; one workgroup per RoI cropped the nuclei tile's boxes in 5.3 ms.)

// Quantisation straight from the cone-cropped windows of the back-propagation: the dense map is win / sum inside the window and 0
// elsewhere, so min(map) = 0 unless the window covers the whole tile, max(map) = max(window), and only the window has to be
// written into the zero-filled uint8 map.  Same float operations, in the same order, as prm_scatter + prm_quantize_kernel.
struct WinQ { unsigned int fmin_bits, fmax_bits; int inside; int pad; };          // per peak; values >= 0: float order == uint order

__global__ __launch_bounds__(256) void winq_stats_kernel(const float* __restrict__ win, const float* __restrict__ sums,
                                                         const int* __restrict__ origins, int Wn, int D, int H, int W,
                                                         WinQ* __restrict__ st) {
  const int p = blockIdx.y, w3 = Wn * Wn * Wn;
  const int oz = origins[3 * p], oy = origins[3 * p + 1], ox = origins[3 * p + 2];
  const float sum = sums[p];
  const float* wp = win + (size_t)p * w3;
  const float inv_w = 1.0f / (float)Wn;
  float mn = INFINITY, mx = 0.f;
  int cnt = 0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < w3; e += gridDim.x * 256) {
    const int r = (int)(((float)e + 0.5f) * inv_w), x = e - r * Wn;          // exact for Wn <= 100 (host-checked)
    const int z = (int)(((float)r + 0.5f) * inv_w), y = r - z * Wn;
    const int qz = oz + z, qy = oy + y, qx = ox + x;
    if ((qz >= 0) & (qz < D) & (qy >= 0) & (qy < H) & (qx >= 0) & (qx < W)) {
      const float f = wp[e] / sum;                         // prm / prm.sum(), peak_response_mapping_3d.py:171
      mn = fminf(mn, f); mx = fmaxf(mx, f); ++cnt;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_down(mn, o, 64)); mx = fmaxf(mx, __shfl_down(mx, o, 64)); cnt += __shfl_down(cnt, o, 64);
  }
  // one atomic triple per WORKGROUP (round 4): per wave they were 100 k atomics on 128 addresses - the kernel spent its time queueing
  __shared__ float s_mn[4], s_mx[4];
  __shared__ int s_cnt[4];
  if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; s_cnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int c4 = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (c4) {
      atomicMin(&st[p].fmin_bits, __float_as_uint(fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]))));
      atomicMax(&st[p].fmax_bits, __float_as_uint(fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]))));
      atomicAdd(&st[p].inside, c4);
    }
  }
}

template <bool COMPACT>
__global__ __launch_bounds__(256) void winq_apply_kernel(const float* __restrict__ win, const float* __restrict__ sums,
                                                         const int* __restrict__ origins, int Wn, int D, int H, int W,
                                                         WinQ* __restrict__ st, uint8_t* __restrict__ out) {
  const int p = blockIdx.y, w3 = Wn * Wn * Wn;
  const int oz = origins[3 * p], oy = origins[3 * p + 1], ox = origins[3 * p + 2];
  const float sum = sums[p];
  const WinQ s = st[p];
  const bool covers = (long long)s.inside == (long long)D * H * W;
  const float mn = covers ? __uint_as_float(s.fmin_bits) : 0.f;      // some voxel of the tile is outside the window: the map holds a 0
  const float mx = __uint_as_float(s.fmax_bits) - mn;                // max(fm - min)
  const float* wp = win + (size_t)p * w3;
  // COMPACT: the uint8 window itself ([P, Wn^3], 0 at window voxels outside the tile) instead of the dense map - all a writer needs
  // to rebuild the map (zero outside the window unless the window covers the tile, and then the window holds every voxel)
  uint8_t* o = out + (COMPACT ? (size_t)p * w3 : (size_t)p * D * H * W);
  const float inv_w = 1.0f / (float)Wn;
  int any = 0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < w3; e += gridDim.x * 256) {
    const int r = (int)(((float)e + 0.5f) * inv_w), x = e - r * Wn;          // exact for Wn <= 100 (host-checked)
    const int z = (int)(((float)r + 0.5f) * inv_w), y = r - z * Wn;
    const int qz = oz + z, qy = oy + y, qx = ox + x;
    if ((qz >= 0) & (qz < D) & (qy >= 0) & (qy < H) & (qx >= 0) & (qx < W)) {
      float v = wp[e] / sum;
      v = v - mn;
      v = v / mx;
      v = v * 255.f;
      const uint8_t u = (uint8_t)v;
      if (COMPACT) o[e] = u; else o[((size_t)qz * H + qy) * W + qx] = u;
      any |= u;
    } else if (COMPACT) {
      o[e] = 0;
    }
  }
  if (__ballot(any != 0) && (threadIdx.x & 63) == 0) atomicOr(&st[p].pad, 1);      // the map is not all zero (binarization_soma.py:74-76)
}

struct RoiStat { int gmax, gmin, pmax, pmin, g2max, g2min; };

__device__ inline bool roi_box(const int* boxes, const int64_t* offsets, int r, int& x1, int& y1, int& z1, int& ex, int& ey, long long& V) {
  x1 = boxes[6 * r]; y1 = boxes[6 * r + 1]; z1 = boxes[6 * r + 2];
  ex = boxes[6 * r + 3] - x1 + 1; ey = boxes[6 * r + 4] - y1 + 1;
  const int ez = boxes[6 * r + 5] - z1 + 1;
  V = (long long)ex * ey * ez;
  return V > 0 && offsets[r + 1] - offsets[r] == V;
}

// Where a RoI's uint8 peak response map comes from: a dense stack [*, D, H, W] (win == 0) or the compact windows [*, win^3] of
// m3d_prm_quantize_windows_compact_u8 with their origins (a map is zero outside its window); `map`: RoI r reads entry map[r].
struct PrmSrc { const uint8_t* prm; const int* map; const int* org; int win; };
struct PrmView { const uint8_t* p; int win, oz, oy, ox; };
__device__ inline PrmView prm_view(const PrmSrc& s, int r, size_t DHW) {
  const int e = s.map ? s.map[r] : r;
  PrmView v;
  v.win = s.win;
  if (s.win == 0) { v.p = s.prm + (size_t)e * DHW; v.oz = v.oy = v.ox = 0; }
  else { v.p = s.prm + (size_t)e * s.win * s.win * s.win; v.oz = s.org[3 * e]; v.oy = s.org[3 * e + 1]; v.ox = s.org[3 * e + 2]; }
  return v;
}
__device__ inline int prm_at(const PrmView& v, size_t idx, int z, int y, int x) {
  if (v.win == 0) return v.p[idx];
  const unsigned wz = (unsigned)(z - v.oz), wy = (unsigned)(y - v.oy), wx = (unsigned)(x - v.ox), w = (unsigned)v.win;
  return (wz < w && wy < w && wx < w) ? v.p[((size_t)wz * w + wy) * w + wx] : 0;
}

__global__ __launch_bounds__(256) void roi_stats_kernel(const uint16_t* __restrict__ image, PrmSrc src,
                                                        const int* __restrict__ boxes, const int64_t* __restrict__ offsets, int D,
                                                        int H, int W, RoiStat* __restrict__ st) {
  const int r = blockIdx.y;
  int x1, y1, z1, ex, ey; long long V;
  if (!roi_box(boxes, offsets, r, x1, y1, z1, ex, ey, V)) return;
  const PrmView pv = prm_view(src, r, (size_t)D * H * W);
  int gmax = 0, gmin = 65535, pmax = 0, pmin = 255;
  const int Vi = (int)V, exy = ex * ey;                              // 32-bit index arithmetic: a crop is a sub-box of one tile
  for (int e = blockIdx.x * 256 + threadIdx.x; e < Vi; e += gridDim.x * 256) {
    const int zq = e / exy, rq = e - zq * exy, yq = rq / ex;
    const int x = x1 + (rq - yq * ex), y = y1 + yq, z = z1 + zq;
    const size_t idx = ((size_t)z * H + y) * W + x;
    const int g = image[idx], p = prm_at(pv, idx, z, y, x);
    gmax = max(gmax, g); gmin = min(gmin, g); pmax = max(pmax, p); pmin = min(pmin, p);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gmax = max(gmax, __shfl_down(gmax, o, 64)); gmin = min(gmin, __shfl_down(gmin, o, 64));
    pmax = max(pmax, __shfl_down(pmax, o, 64)); pmin = min(pmin, __shfl_down(pmin, o, 64));
  }
  __shared__ int s_r[4][4];                                        // one atomic set per workgroup, not per wave (see winq_stats_kernel)
  if ((threadIdx.x & 63) == 0) { int* q = s_r[threadIdx.x >> 6]; q[0] = gmax; q[1] = gmin; q[2] = pmax; q[3] = pmin; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(&st[r].gmax, max(max(s_r[0][0], s_r[1][0]), max(s_r[2][0], s_r[3][0])));
    atomicMin(&st[r].gmin, min(min(s_r[0][1], s_r[1][1]), min(s_r[2][1], s_r[3][1])));
    atomicMax(&st[r].pmax, max(max(s_r[0][2], s_r[1][2]), max(s_r[2][2], s_r[3][2])));
    atomicMin(&st[r].pmin, min(min(s_r[0][3], s_r[1][3]), min(s_r[2][3], s_r[3][3])));
  }
}

// mode 0 (soma): writes both crops.  mode 1 (nuclei): writes the (stretched) image crop and the range of the values written.
__global__ __launch_bounds__(256) void roi_apply1_kernel(const uint16_t* __restrict__ image, PrmSrc src,
                                                         const int* __restrict__ boxes, const int64_t* __restrict__ offsets, int D,
                                                         int H, int W, int mode, RoiStat* __restrict__ st,
                                                         uint16_t* __restrict__ out_img, uint16_t* __restrict__ out_prm) {
  const int r = blockIdx.y;
  int x1, y1, z1, ex, ey; long long V;
  if (!roi_box(boxes, offsets, r, x1, y1, z1, ex, ey, V)) return;
  const PrmView pv = prm_view(src, r, (size_t)D * H * W);
  uint16_t* oi = out_img + offsets[r];
  uint16_t* op = out_prm + offsets[r];
  const RoiStat s = st[r];
  const double gmaxd = (double)s.gmax, pmaxd = (double)s.pmax;
  const bool stretch = (s.gmax - s.gmin + 1) < 400;
  int g2max = 0, g2min = 65535;
  const int Vi = (int)V, exy = ex * ey;                              // 32-bit index arithmetic: a crop is a sub-box of one tile
  for (int e = blockIdx.x * 256 + threadIdx.x; e < Vi; e += gridDim.x * 256) {
    const int zq = e / exy, rq = e - zq * exy, yq = rq / ex;
    const int x = x1 + (rq - yq * ex), y = y1 + yq, z = z1 + zq;
    const size_t idx = ((size_t)z * H + y) * W + x;
    if (mode == 0) {
      double f = (double)image[idx] / gmaxd * 300.0;                  // binarization_soma.py:86-87
      f = f < 0.0 ? 0.0 : (f > 300.0 ? 300.0 : f);                    // np.clip
      oi[e] = (uint16_t)(f + 30.0);                                   // astype(np.uint16)
      const double q = (double)prm_at(pv, idx, z, y, x) / pmaxd * 300.0 + 30.0;        // :90-91
      op[e] = (uint16_t)rint(q);                                      // np.round (half to even)
    } else {
      uint16_t v = image[idx];
      if (stretch) v = (uint16_t)((uint16_t)((double)v / gmaxd * 400.0) + (uint16_t)s.gmin);   // binarization_nuclei.py:114-117
      oi[e] = v;
      g2max = max(g2max, (int)v); g2min = min(g2min, (int)v);
    }
  }
  if (mode == 1) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { g2max = max(g2max, __shfl_down(g2max, o, 64)); g2min = min(g2min, __shfl_down(g2min, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMax(&st[r].g2max, g2max); atomicMin(&st[r].g2min, g2min); }
  }
}

// nuclei: PRM mapped into [gray_min, gray_max] of the (stretched) image crop (binarization_nuclei.py:118-121)
__global__ __launch_bounds__(256) void roi_apply2_kernel(PrmSrc src, const int* __restrict__ boxes,
                                                         const int64_t* __restrict__ offsets, int D, int H, int W,
                                                         const RoiStat* __restrict__ st, uint16_t* __restrict__ out_prm) {
  const int r = blockIdx.y;
  int x1, y1, z1, ex, ey; long long V;
  if (!roi_box(boxes, offsets, r, x1, y1, z1, ex, ey, V)) return;
  const PrmView pv = prm_view(src, r, (size_t)D * H * W);
  uint16_t* op = out_prm + offsets[r];
  const RoiStat s = st[r];
  const double pmaxd = (double)s.pmax, pmind = (double)s.pmin;
  const double span = (double)(uint16_t)((uint16_t)s.g2max - (uint16_t)s.g2min), base = (double)s.g2min;
  const int Vi = (int)V, exy = ex * ey;                              // 32-bit index arithmetic: a crop is a sub-box of one tile
  for (int e = blockIdx.x * 256 + threadIdx.x; e < Vi; e += gridDim.x * 256) {
    const int zq = e / exy, rq = e - zq * exy, yq = rq / ex;
    const int x = x1 + (rq - yq * ex), y = y1 + yq, z = z1 + zq;
    const size_t idx = ((size_t)z * H + y) * W + x;
    const double q = ((double)prm_at(pv, idx, z, y, x) - pmind) / (pmaxd - pmind) * span + base;
    op[e] = (uint16_t)rint(q);
  }
}

__global__ void roi_stat_init_kernel(RoiStat* st, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st[i] = RoiStat{0, 65535, 0, 255, 0, 65535};
}
__global__ void winq_init_kernel(WinQ* st, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st[i] = WinQ{0x7F800000u, 0u, 0, 0};
}

}  // namespace

M3D_API int m3d_prm_quantize_u8(const float* d_prm, int num_maps, int64_t voxels_per_map, uint8_t* d_out, void* stream) {
  if (num_maps < 0 || voxels_per_map <= 0) return M3D_EINVAL;
  if (num_maps == 0) return M3D_OK;
  if (!d_prm || !d_out) return M3D_EINVAL;
  hipLaunchKernelGGL(prm_quantize_kernel, dim3(num_maps), dim3(256), 0, m3d::as_stream(stream), d_prm, (long long)voxels_per_map,
                     d_out);
  return m3d::check_launch("prm_quantize_u8");
}

M3D_API int m3d_roi_normalize(const uint16_t* d_image, const uint8_t* d_prm_u8, const int32_t* d_boxes, const int64_t* d_offsets,
                              int num_rois, int depth, int height, int width, int mode, uint16_t* d_out_image,
                              uint16_t* d_out_prm, void* stream) {
  if (num_rois < 0 || depth <= 0 || height <= 0 || width <= 0 || (mode != 0 && mode != 1)) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_image || !d_prm_u8 || !d_boxes || !d_offsets || !d_out_image || !d_out_prm) return M3D_EINVAL;
  hipLaunchKernelGGL(roi_normalize_kernel, dim3(num_rois), dim3(256), 0, m3d::as_stream(stream), d_image, d_prm_u8, d_boxes,
                     d_offsets, depth, height, width, mode, d_out_image, d_out_prm);
  return m3d::check_launch("roi_normalize");
}

/* Round 2 (element-parallel forms; results identical to the functions above).
 * m3d_prm_quantize_windows_u8: the uint8 maps straight from the cone-cropped windows of the back-propagation (d_windows [P,win^3],
 *   d_sums [P], d_origins int32 [P,3]) = m3d_prm_quantize_u8(m3d_prm_scatter(...)) without the dense float maps.  d_ws: 16 * P bytes.
 * m3d_roi_normalize_ws: m3d_roi_normalize over grid (chunks, RoI).  d_ws: 24 * num_rois bytes. */
M3D_API int m3d_prm_quantize_windows_u8(const float* d_windows, const float* d_sums, const int32_t* d_origins, int num_peaks, int win,
                                        int depth, int height, int width, uint8_t* d_out, void* d_ws, size_t ws_bytes, void* stream) {
  if (num_peaks < 0 || win <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_windows || !d_sums || !d_origins || !d_out || !d_ws) return M3D_EINVAL;
  if (ws_bytes < sizeof(WinQ) * (size_t)num_peaks) return M3D_EWORKSPACE;
  if (num_peaks > 65535 || win > 100) return M3D_EUNSUPPORTED;      // float-reciprocal index split in the kernels
  hipStream_t st = m3d::as_stream(stream);
  WinQ* q = (WinQ*)d_ws;
  (void)hipMemsetAsync(d_out, 0, (size_t)num_peaks * depth * height * width, st);
  hipLaunchKernelGGL(winq_init_kernel, dim3((num_peaks + 255) / 256), dim3(256), 0, st, q, num_peaks);
  const int w3 = win * win * win;
  int chunks = (w3 + 255) / 256;
  chunks = chunks > 64 ? 64 : chunks;
  int schunks = (4096 + num_peaks - 1) / num_peaks;            // the statistics pass ends in atomics on one record per peak: ~4 k workgroups in all
  schunks = schunks < 1 ? 1 : (schunks > chunks ? chunks : schunks);
  hipLaunchKernelGGL(winq_stats_kernel, dim3(schunks, num_peaks), dim3(256), 0, st, d_windows, d_sums, d_origins, win, depth, height,
                     width, q);
  hipLaunchKernelGGL(winq_apply_kernel<false>, dim3(chunks, num_peaks), dim3(256), 0, st, d_windows, d_sums, d_origins, win, depth, height,
                     width, q, d_out);
  return m3d::check_launch("prm_quantize_windows_u8");
}

/* Round 4: the same quantisation, written as uint8 WINDOWS [P, win^3] (0 where a window voxel lies outside the tile): the whole-volume
 * driver moves these to the host (0.59 MB per peak of a nuclei tile instead of 2.56 MB of mostly-zero map) and the file writer rebuilds
 * each page around them (m3d_tiff_encode_window_stack_u8).  Same d_ws contract as m3d_prm_quantize_windows_u8. */
M3D_API int m3d_prm_quantize_windows_compact_u8(const float* d_windows, const float* d_sums, const int32_t* d_origins, int num_peaks, int win,
                                                int depth, int height, int width, uint8_t* d_out_windows, void* d_ws, size_t ws_bytes,
                                                void* stream) {
  if (num_peaks < 0 || win <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_windows || !d_sums || !d_origins || !d_out_windows || !d_ws) return M3D_EINVAL;
  if (ws_bytes < sizeof(WinQ) * (size_t)num_peaks) return M3D_EWORKSPACE;
  if (num_peaks > 65535 || win > 100) return M3D_EUNSUPPORTED;
  hipStream_t st = m3d::as_stream(stream);
  WinQ* q = (WinQ*)d_ws;
  hipLaunchKernelGGL(winq_init_kernel, dim3((num_peaks + 255) / 256), dim3(256), 0, st, q, num_peaks);
  const int w3 = win * win * win;
  int chunks = (w3 + 255) / 256;
  chunks = chunks > 64 ? 64 : chunks;
  int schunks = (4096 + num_peaks - 1) / num_peaks;
  schunks = schunks < 1 ? 1 : (schunks > chunks ? chunks : schunks);
  hipLaunchKernelGGL(winq_stats_kernel, dim3(schunks, num_peaks), dim3(256), 0, st, d_windows, d_sums, d_origins, win, depth, height,
                     width, q);
  hipLaunchKernelGGL(winq_apply_kernel<true>, dim3(chunks, num_peaks), dim3(256), 0, st, d_windows, d_sums, d_origins, win, depth, height,
                     width, q, d_out_windows);
  return m3d::check_launch("prm_quantize_windows_compact_u8");
}

/* m3d_roi_normalize_ws with an indirection and an optional compact source: RoI r reads map d_map_index[r] (NULL: r) of d_prm_u8, which
 * is the dense stack [*, depth, height, width] (win = 0) or the uint8 windows [*, win^3] of m3d_prm_quantize_windows_compact_u8 at
 * d_win_origins (int32 [*, 3]; a map is zero outside its window).  The detections with a valid crop box are a subset of a tile's peaks:
 * gathering their maps was a 200 MB copy per soma tile, the dense maps themselves 210 MB written to be read at the crop boxes only. */
M3D_API int m3d_roi_normalize_idx(const uint16_t* d_image, const uint8_t* d_prm_u8, const int32_t* d_map_index, int win,
                                  const int32_t* d_win_origins, const int32_t* d_boxes, const int64_t* d_offsets, int num_rois,
                                  int64_t total_voxels, int depth, int height, int width, int mode, uint16_t* d_out_image,
                                  uint16_t* d_out_prm, void* d_ws, size_t ws_bytes, void* stream) {
  if (win < 0 || (win > 0 && !d_win_origins)) return M3D_EINVAL;
  const PrmSrc src{d_prm_u8, d_map_index, d_win_origins, win};
  if (num_rois < 0 || depth <= 0 || height <= 0 || width <= 0 || (mode != 0 && mode != 1)) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_image || !d_prm_u8 || !d_boxes || !d_offsets || !d_out_image || !d_out_prm || !d_ws) return M3D_EINVAL;
  if (ws_bytes < sizeof(RoiStat) * (size_t)num_rois) return M3D_EWORKSPACE;
  if (num_rois > 65535 || (long long)depth * height * width >= 0x7FFFFFFFll) return M3D_EUNSUPPORTED;
  hipStream_t st = m3d::as_stream(stream);
  RoiStat* rs = (RoiStat*)d_ws;
  long long per = (total_voxels / num_rois + 255) / 256, want = (8192 + num_rois - 1) / num_rois;
  long long c = want < per ? want : per;
  const int chunks = (int)(c < 1 ? 1 : (c > 1024 ? 1024 : c));
  const dim3 grid(chunks, num_rois), block(256);
  hipLaunchKernelGGL(roi_stat_init_kernel, dim3((num_rois + 255) / 256), dim3(256), 0, st, rs, num_rois);
  hipLaunchKernelGGL(roi_stats_kernel, grid, block, 0, st, d_image, src, d_boxes, d_offsets, depth, height, width, rs);
  hipLaunchKernelGGL(roi_apply1_kernel, grid, block, 0, st, d_image, src, d_boxes, d_offsets, depth, height, width, mode, rs,
                     d_out_image, d_out_prm);
  if (mode == 1)
    hipLaunchKernelGGL(roi_apply2_kernel, grid, block, 0, st, src, d_boxes, d_offsets, depth, height, width, (const RoiStat*)rs,
                       d_out_prm);
  return m3d::check_launch("roi_normalize_idx");
}

M3D_API int m3d_roi_normalize_ws(const uint16_t* d_image, const uint8_t* d_prm_u8, const int32_t* d_boxes, const int64_t* d_offsets,
                                 int num_rois, int64_t total_voxels, int depth, int height, int width, int mode, uint16_t* d_out_image,
                                 uint16_t* d_out_prm, void* d_ws, size_t ws_bytes, void* stream) {
  return m3d_roi_normalize_idx(d_image, d_prm_u8, nullptr, 0, nullptr, d_boxes, d_offsets, num_rois, total_voxels, depth, height, width, mode, d_out_image,
                               d_out_prm, d_ws, ws_bytes, stream);
}
