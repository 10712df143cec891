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
