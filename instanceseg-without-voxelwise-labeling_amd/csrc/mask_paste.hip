// Paste of the mask branch's soft M^3 masks into full-volume uint8 masks: lib/core/test.py:886-945 (segm_results).
// Per detection: zero-pad the mask by one voxel (:899-908), resize the (M+2)^3 block to the integer box's (s, h, w) with
// skimage.transform.resize(order 1, mode 'reflect', anti_aliasing=True) (:919), threshold (:920), copy the part of the box that
// lies inside the volume (:923-931).  The resize is restated from scikit-image's n-D algorithm on scipy.ndimage, operation for
// operation (tests compare with scipy itself):
//   * Gaussian pre-filter per axis with sigma = max(0, (in/out - 1)/2), truncated at 4 sigma, 'mirror' boundary; scipy's
//     correlate1d for symmetric kernels: t = x[l] w0, then for d = r..1: t += (x[l-d] + x[l+d]) w[d] in double, every axis pass
//     rounded to float32 (the array dtype).  The weights come from the caller (NumPy's exp / pairwise sum on the host: see ops.py);
//   * order-1 map_coordinates at c = (i + 0.5) in/out - 0.5, the coordinate mirrored into [0, n-1], corner index i0 + 1 mirrored,
//     value = sum over the 8 corners in (z, y, x) offset order of ((v wz) wy) wx in double, rounded to float32;
//   * the final clip to the filtered block's range is the identity for a convex combination and is not executed.
// Not a hot path (MODEL.MASK_ON is False in both shipped configs): plain element-parallel kernels over a global workspace.
#include "m3d_common.h"

namespace {

__device__ __forceinline__ int mirror_index(int i, int n) {       // ndimage 'mirror': d c b | a b c d | c b a
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i = (i < 0 ? -i : i) % p;
  return i > n - 1 ? p - i : i;
}

__device__ __forceinline__ double mirror_coord(double c, int n) {  // scipy ni_interpolation.c map_coordinate, NI_EXTEND_MIRROR
  if (n <= 1) return 0.0;
  const double s2 = 2.0 * n - 2.0;
  if (c < 0) {
    c = s2 * (double)(long long)(-c / s2) + c;
    c = c <= 1 - n ? c + s2 : -c;
  } else if (c > n - 1) {
    c -= s2 * (double)(long long)(c / s2);
    if (c >= n) c = s2 - c;
  }
  return c;
}

// pad[r][P][P][P] <- masks[r][channel[r]][M][M][M] with a one-voxel zero border
__global__ __launch_bounds__(256) void paste_pad_kernel(const float* __restrict__ masks, const int* __restrict__ channel, int C, int M,
                                                        float* __restrict__ pad, long long total) {
  const int P = M + 2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int x = (int)(e % P), y = (int)((e / P) % P), z = (int)((e / ((long long)P * P)) % P), r = (int)(e / ((long long)P * P * P));
    float v = 0.f;
    if (x >= 1 && x <= M && y >= 1 && y <= M && z >= 1 && z <= M) {
      const int ch = min(max(channel[r], 0), C - 1);
      v = masks[(((size_t)r * C + ch) * M + (z - 1)) * M * M + (size_t)(y - 1) * M + (x - 1)];
    }
    pad[e] = v;
  }
}

// one Gaussian pass along `axis` (0 = z, 1 = y, 2 = x); radius 0 = the axis is not filtered (copy)
__global__ __launch_bounds__(256) void paste_gauss_kernel(const float* __restrict__ src, float* __restrict__ dst, int P, int axis,
                                                          const int* __restrict__ radius, const double* __restrict__ weights, int wstride,
                                                          long long total) {
  const long long P3 = (long long)P * P * P;
  const int stride = axis == 0 ? P * P : (axis == 1 ? P : 1);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int r = (int)(e / P3);
    const int rad = radius[3 * r + axis];
    if (rad <= 0) { dst[e] = src[e]; continue; }
    const long long q = e - (long long)r * P3;
    const int l = axis == 0 ? (int)(q / ((long long)P * P)) : (axis == 1 ? (int)((q / P) % P) : (int)(q % P));
    const float* line = src + e - (long long)l * stride;
    const double* w = weights + ((size_t)r * 3 + axis) * wstride;   // w[d] = weight at distance d
    double t = (double)line[(long long)l * stride] * w[0];
    for (int d = rad; d >= 1; --d)
      t += ((double)line[(long long)mirror_index(l - d, P) * stride] + (double)line[(long long)mirror_index(l + d, P) * stride]) * w[d];
    dst[e] = (float)t;
  }
}

// out[r][z][y][x] = 1 where the resized block exceeds thresh, over the part of the box inside the volume.  grid = (chunks, R)
__global__ __launch_bounds__(256) void paste_interp_kernel(const float* __restrict__ blk, int P, const int* __restrict__ boxes, float thresh,
                                                           int S, int H, int W, unsigned char* __restrict__ out) {
  const int r = blockIdx.y;
  const int* rb = boxes + 6 * r;                                   // x0, y0, z0, x1, y1, z1 (inclusive, may leave the volume)
  const int bw = max(rb[3] - rb[0] + 1, 1), bh = max(rb[4] - rb[1] + 1, 1), bs = max(rb[5] - rb[2] + 1, 1);
  const int x0 = max(rb[0], 0), x1 = min(rb[3] + 1, W), y0 = max(rb[1], 0), y1 = min(rb[4] + 1, H), z0 = max(rb[2], 0), z1 = min(rb[5] + 1, S);
  // (a box whose upper corner lies below its lower one has size 1 but an empty slice, as in the reference: nothing is written)
  if (x1 <= x0 || y1 <= y0 || z1 <= z0) return;
  const int nx = x1 - x0, ny = y1 - y0, nz = z1 - z0;
  const double fx = (double)P / (double)bw, fy = (double)P / (double)bh, fz = (double)P / (double)bs;
  const float* b = blk + (size_t)r * P * P * P;
  const long long n = (long long)nx * ny * nz;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int x = x0 + (int)(e % nx), y = y0 + (int)((e / nx) % ny), z = z0 + (int)(e / ((long long)nx * ny));
    const int ix = x - rb[0], iy = y - rb[1], iz = z - rb[2];
    // the reference slices mask[(z0 - rb2):(z1 - rb2), ...] of an array of shape (bs, bh, bw): indices beyond it do not exist
    if (ix >= bw || iy >= bh || iz >= bs) continue;
    const double cx = mirror_coord(fx * (ix + 0.5) - 0.5, P), cy = mirror_coord(fy * (iy + 0.5) - 0.5, P), cz = mirror_coord(fz * (iz + 0.5) - 0.5, P);
    const int ax = (int)floor(cx), ay = (int)floor(cy), az = (int)floor(cz);
    const double tx = cx - ax, ty = cy - ay, tz = cz - az;
    const int jx[2] = {ax, mirror_index(ax + 1, P)}, jy[2] = {ay, mirror_index(ay + 1, P)}, jz[2] = {az, mirror_index(az + 1, P)};
    const double wx[2] = {1.0 - tx, tx}, wy[2] = {1.0 - ty, ty}, wz[2] = {1.0 - tz, tz};
    double t = 0.0;
#pragma unroll
    for (int oz = 0; oz < 2; ++oz)
#pragma unroll
      for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) {
          double c = (double)b[((size_t)jz[oz] * P + jy[oy]) * P + jx[ox]];
          c *= wz[oz]; c *= wy[oy]; c *= wx[ox];
          t += c;
        }
    if ((float)t > thresh) out[(((size_t)r * S + z) * H + y) * W + x] = 1;
  }
}

}  // namespace

M3D_API size_t m3d_mask_paste3d_workspace_bytes(int num_dets, int resolution) {
  if (num_dets <= 0 || resolution <= 0) return 0;
  const size_t P = (size_t)resolution + 2;
  return 2 * sizeof(float) * (size_t)num_dets * P * P * P;
}

M3D_API int m3d_mask_paste3d(const float* d_masks, int num_dets, int channels, int resolution, const int32_t* d_channel,
                             const int32_t* d_boxes, const int32_t* d_radius, const double* d_weights, int weight_stride, float thresh,
                             int depth, int height, int width, unsigned char* d_out, void* d_ws, size_t ws_bytes, void* stream) {
  if (num_dets < 0 || channels <= 0 || resolution <= 0 || depth <= 0 || height <= 0 || width <= 0 || weight_stride <= 0) return M3D_EINVAL;
  if (num_dets == 0) return M3D_OK;
  if (!d_masks || !d_channel || !d_boxes || !d_radius || !d_weights || !d_out) return M3D_EINVAL;
  if (!d_ws || ws_bytes < m3d_mask_paste3d_workspace_bytes(num_dets, resolution)) return M3D_EWORKSPACE;
  hipStream_t st = m3d::as_stream(stream);
  const int P = resolution + 2;
  const long long total = (long long)num_dets * P * P * P;
  float* a = (float*)d_ws;
  float* b = a + total;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (hipMemsetAsync(d_out, 0, (size_t)num_dets * depth * height * width, st) != hipSuccess) return M3D_ELAUNCH;
  hipLaunchKernelGGL(paste_pad_kernel, dim3(blocks), dim3(256), 0, st, d_masks, d_channel, channels, resolution, a, total);
  hipLaunchKernelGGL(paste_gauss_kernel, dim3(blocks), dim3(256), 0, st, (const float*)a, b, P, 0, d_radius, d_weights, weight_stride, total);
  hipLaunchKernelGGL(paste_gauss_kernel, dim3(blocks), dim3(256), 0, st, (const float*)b, a, P, 1, d_radius, d_weights, weight_stride, total);
  hipLaunchKernelGGL(paste_gauss_kernel, dim3(blocks), dim3(256), 0, st, (const float*)a, b, P, 2, d_radius, d_weights, weight_stride, total);
  hipLaunchKernelGGL(paste_interp_kernel, dim3(64, num_dets), dim3(256), 0, st, (const float*)b, P, d_boxes, thresh, depth, height, width, d_out);
  return m3d::check_launch("mask_paste3d");
}
