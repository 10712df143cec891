// Volume pre-filters of tools/binarization_nuclei.py:44-45 on device, bit-exact with SciPy:
//   img = ndimage.gaussian_filter(img, sigma=1)      (uint16 in -> uint16 out after EVERY axis pass)
//   img = ndimage.median_filter(img, size=3)
// Restated from the published SciPy algorithm (scipy/ndimage/src/ni_filters.c, NI_Correlate1D symmetric branch;
// _filters.py gaussian_filter / median_filter; SciPy 1.15.3 is the version in this image): separable correlate1d along
// axes 0, 1, 2 in that order, 'reflect' boundary (d c b a | a b c d | d c b a), fp64 accumulation in the order
// tmp = in[0]*w[0]; for j = -r..-1: tmp += (in[j] + in[-j]) * w[j], C cast of the double to uint16 (truncation);
// the weights are computed by the HOST exactly as SciPy does (NumPy exp / sum) and passed in.  Median: rank 13 of the
// 27 reflect-padded neighbours.  Compiled with -ffp-contract=off (no FMA: SciPy's C loop has separate multiply and add).
#include "m3d_common.h"

namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {   // half-sample symmetric, any distance
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - 1 - i : i;
}

// one pass along `axis` (0 = z, 1 = y, 2 = x); radius <= 8; w[0] = centre weight, w[j] = weight at distance j
__global__ __launch_bounds__(256) void gauss1d_u16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int D, int H,
                                                          int W, int axis, int radius, const double* __restrict__ w) {
  const long long total = (long long)D * H * W;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % W), y = (int)((e / W) % H), z = (int)(e / ((long long)W * H));
  const int n = axis == 0 ? D : axis == 1 ? H : W;
  const int c = axis == 0 ? z : axis == 1 ? y : x;
  const long long stride = axis == 0 ? (long long)H * W : axis == 1 ? W : 1;
  const uint16_t* line = in + (e - (long long)c * stride);
  double tmp = (double)line[(long long)c * stride] * w[0];
  for (int j = radius; j >= 1; --j) {                     // SciPy: jj = -size1 .. -1
    const double a = (double)line[(long long)reflect_idx(c - j, n) * stride];
    const double b = (double)line[(long long)reflect_idx(c + j, n) * stride];
    tmp += (a + b) * w[j];
  }
  out[e] = (uint16_t)tmp;
}

__device__ __forceinline__ void cswap(uint16_t& a, uint16_t& b) {
  const uint16_t lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo; b = hi;
}

// median of the 27 reflect-padded neighbours: Batcher odd-even merge sort of 32 (padded with 0xFFFF), element 13
__global__ __launch_bounds__(256) void median3_u16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int D, int H,
                                                          int W) {
  const long long total = (long long)D * H * W;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % W), y = (int)((e / W) % H), z = (int)(e / ((long long)W * H));
  uint16_t v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = 0xFFFF;
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        v[(dz + 1) * 9 + (dy + 1) * 3 + dx + 1] =
            in[((long long)reflect_idx(z + dz, D) * H + reflect_idx(y + dy, H)) * W + reflect_idx(x + dx, W)];
#pragma unroll
  for (int p = 1; p < 32; p <<= 1)
#pragma unroll
    for (int k = p; k >= 1; k >>= 1)
#pragma unroll
      for (int j = k % p; j + k < 32; j += 2 * k)
#pragma unroll
        for (int i = 0; i < k; ++i)
          if (i + j + k < 32 && (i + j) / (2 * p) == (i + j + k) / (2 * p)) cswap(v[i + j], v[i + j + k]);
  out[e] = v[13];
}

}  // namespace

M3D_API int m3d_gaussian_filter_u16(const uint16_t* d_in, uint16_t* d_out, uint16_t* d_tmp, int depth, int height, int width,
                                    const double* d_weights, int radius, void* stream) {
  if (!d_in || !d_out || !d_tmp || !d_weights || depth <= 0 || height <= 0 || width <= 0 || radius < 0 || radius > 64) return M3D_EINVAL;
  const long long total = (long long)depth * height * width;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipStream_t st = m3d::as_stream(stream);
  // axis 0: in -> out, axis 1: out -> tmp, axis 2: tmp -> out   (SciPy: input -> output, then output -> output per axis)
  hipLaunchKernelGGL(gauss1d_u16_kernel, dim3(blocks), dim3(256), 0, st, d_in, d_out, depth, height, width, 0, radius, d_weights);
  hipLaunchKernelGGL(gauss1d_u16_kernel, dim3(blocks), dim3(256), 0, st, (const uint16_t*)d_out, d_tmp, depth, height, width, 1, radius,
                     d_weights);
  hipLaunchKernelGGL(gauss1d_u16_kernel, dim3(blocks), dim3(256), 0, st, (const uint16_t*)d_tmp, d_out, depth, height, width, 2, radius,
                     d_weights);
  return m3d::check_launch("gaussian_filter_u16");
}

M3D_API int m3d_median_filter3_u16(const uint16_t* d_in, uint16_t* d_out, int depth, int height, int width, void* stream) {
  if (!d_in || !d_out || d_in == d_out || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  const long long total = (long long)depth * height * width;
  hipLaunchKernelGGL(median3_u16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, m3d::as_stream(stream), d_in, d_out, depth,
                     height, width);
  return m3d::check_launch("median_filter3_u16");
}
