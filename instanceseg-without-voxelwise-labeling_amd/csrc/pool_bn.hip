// MaxPool3d(2,2) forward/backward and small reductions for gfx950 (HBM-bound elementwise work).
// Reference call sites: lib/modeling/DSN.py:21,26,32 (nn.MaxPool3d(2, 2, padding=0), floor mode) and
// lib/prm/peak_backprop_3d.py:38 (input.min()).
#include "m3d_common.h"

namespace {

// one thread per output voxel; window scanned in (z,y,x) order, first maximum wins (PyTorch semantics;
// NaN propagates: a NaN beats everything once seen).
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           uint8_t* __restrict__ argmax, long long total, int D, int H, int W,
                                                           int OD, int OH, int OW) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    long long t = e / OW;
    const int oy = (int)(t % OH); t /= OH;
    const int oz = (int)(t % OD);
    const long long bc = t / OD;
    const float* p = in + ((bc * D + 2 * oz) * H + 2 * oy) * (long long)W + 2 * ox;
    float best = p[0];
    int bi = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      const float v = p[((q >> 2) * H + ((q >> 1) & 1)) * (long long)W + (q & 1)];
      if (v > best || (v != v && best == best)) { best = v; bi = q; }
    }
    out[e] = best;
    if (argmax) argmax[e] = (uint8_t)bi;
  }
}

// gradient routed to the argmax voxel; every input voxel belongs to at most one window => plain stores.
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ gout, const uint8_t* __restrict__ argmax,
                                                           float* __restrict__ gin, long long total_in, int D, int H, int W,
                                                           int OD, int OH, int OW) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total_in; e += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(e % W);
    long long t = e / W;
    const int y = (int)(t % H); t /= H;
    const int z = (int)(t % D);
    const long long bc = t / D;
    const int ox = x >> 1, oy = y >> 1, oz = z >> 1;
    float g = 0.f;
    if (ox < OW && oy < OH && oz < OD) {
      const long long o = ((bc * OD + oz) * OH + oy) * (long long)OW + ox;
      const int q = ((z & 1) << 2) | ((y & 1) << 1) | (x & 1);
      if (argmax[o] == q) g = gout[o];
    }
    gin[e] = g;
  }
}

__device__ inline float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_down(v, off, 64));
  return v;
}

// two-stage min reduction; partial[blocks] then final (x.min(), peak_backprop_3d.py:38)
__global__ __launch_bounds__(256) void min_partial_kernel(const float* __restrict__ in, long long n, float* __restrict__ partial) {
  __shared__ float sm[4];
  float v = INFINITY;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    v = fminf(v, in[e]);
  v = wave_min(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fminf(fminf(sm[0], sm[1]), fminf(sm[2], sm[3]));
}
__global__ __launch_bounds__(256) void min_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float sm[4];
  float v = INFINITY;
  for (int e = threadIdx.x; e < n; e += blockDim.x) v = fminf(v, partial[e]);
  v = wave_min(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fminf(fminf(sm[0], sm[1]), fminf(sm[2], sm[3]));
}

// the same for up to 12 arrays at once (blockIdx.y = array): the PRM forward needs the minimum of every layer's input (one norm conv
// each, peak_backprop_3d.py:38) - two launches for all of them instead of two per layer
constexpr int kMinMulti = 12, kMinBlocks = 128;
struct MinMultiArgs { const float* in[kMinMulti]; long long n[kMinMulti]; };
__global__ __launch_bounds__(256) void min_partial_multi_kernel(MinMultiArgs a, float* __restrict__ partial) {
  __shared__ float sm[4];
  const float* in = a.in[blockIdx.y];
  const long long n = a.n[blockIdx.y];
  float v = INFINITY;
  if ((n & 3) == 0 && ((uintptr_t)in & 15) == 0) {
    const float4* in4 = reinterpret_cast<const float4*>(in);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (n >> 2); e += (long long)gridDim.x * 256) {
      const float4 q = in4[e];
      v = fminf(fminf(v, fminf(q.x, q.y)), fminf(q.z, q.w));
    }
  } else {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) v = fminf(v, in[e]);
  }
  v = wave_min(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * kMinBlocks + blockIdx.x] = fminf(fminf(sm[0], sm[1]), fminf(sm[2], sm[3]));
}
__global__ __launch_bounds__(64) void min_final_multi_kernel(const float* __restrict__ partial, float* __restrict__ out) {
  float v = fminf(partial[blockIdx.x * kMinBlocks + threadIdx.x], partial[blockIdx.x * kMinBlocks + 64 + threadIdx.x]);
  v = wave_min(v);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

// minima AND maxima of up to 12 arrays in the same two launches (round 6: the f16x2 norm convolutions scale their operand X - min X by its
// largest value, max X - min X)
__global__ __launch_bounds__(256) void minmax_partial_multi_kernel(MinMultiArgs a, float* __restrict__ partial) {
  __shared__ float smn[4], smx[4];
  const float* in = a.in[blockIdx.y];
  const long long n = a.n[blockIdx.y];
  float v = INFINITY, u = -INFINITY;
  if ((n & 3) == 0 && ((uintptr_t)in & 15) == 0) {
    const float4* in4 = reinterpret_cast<const float4*>(in);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (n >> 2); e += (long long)gridDim.x * 256) {
      const float4 q = in4[e];
      v = fminf(fminf(v, fminf(q.x, q.y)), fminf(q.z, q.w));
      u = fmaxf(fmaxf(u, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
    }
  } else {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) { v = fminf(v, in[e]); u = fmaxf(u, in[e]); }
  }
  v = wave_min(v);
  u = -wave_min(-u);
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = v; smx[threadIdx.x >> 6] = u; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[(2 * blockIdx.y) * kMinBlocks + blockIdx.x] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    partial[(2 * blockIdx.y + 1) * kMinBlocks + blockIdx.x] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}
__global__ __launch_bounds__(64) void minmax_final_multi_kernel(const float* __restrict__ partial, float* __restrict__ mins, float* __restrict__ maxs) {
  const float* pm = partial + (2 * blockIdx.x) * kMinBlocks;
  float v = fminf(pm[threadIdx.x], pm[64 + threadIdx.x]);
  float u = fmaxf(pm[kMinBlocks + threadIdx.x], pm[kMinBlocks + 64 + threadIdx.x]);
  v = wave_min(v);
  u = -wave_min(-u);
  if (threadIdx.x == 0) { mins[blockIdx.x] = v; maxs[blockIdx.x] = u; }
}

inline unsigned grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

M3D_API int m3d_maxpool3d_2x_forward(const float* d_in, float* d_out, uint8_t* d_argmax, int batch_channels, int depth,
                                     int height, int width, void* stream) {
  if (!d_in || !d_out || batch_channels <= 0 || depth < 2 || height < 2 || width < 2) return M3D_EINVAL;
  const int OD = depth / 2, OH = height / 2, OW = width / 2;
  const long long total = (long long)batch_channels * OD * OH * OW;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, m3d::as_stream(stream), d_in, d_out, d_argmax,
                     total, depth, height, width, OD, OH, OW);
  return m3d::check_launch("maxpool3d_2x_forward");
}

M3D_API int m3d_maxpool3d_2x_backward(const float* d_grad_out, const uint8_t* d_argmax, float* d_grad_in, int batch_channels,
                                      int depth, int height, int width, void* stream) {
  if (!d_grad_out || !d_argmax || !d_grad_in || batch_channels <= 0 || depth < 2 || height < 2 || width < 2) return M3D_EINVAL;
  const int OD = depth / 2, OH = height / 2, OW = width / 2;
  const long long total = (long long)batch_channels * depth * height * width;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, m3d::as_stream(stream), d_grad_out, d_argmax,
                     d_grad_in, total, depth, height, width, OD, OH, OW);
  return m3d::check_launch("maxpool3d_2x_backward");
}

M3D_API size_t m3d_reduce_min_workspace_bytes(void) { return 1024 * sizeof(float); }

M3D_API int m3d_reduce_min(const float* d_in, int64_t n, float* d_out, void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_in || !d_out || !d_ws || n <= 0) return M3D_EINVAL;
  if (ws_bytes < 1024 * sizeof(float)) return M3D_EWORKSPACE;
  long long blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(min_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, m3d::as_stream(stream), d_in, (long long)n,
                     (float*)d_ws);
  hipLaunchKernelGGL(min_final_kernel, dim3(1), dim3(256), 0, m3d::as_stream(stream), (const float*)d_ws, (int)blocks, d_out);
  return m3d::check_launch("reduce_min");
}

/* Minima of `count` (<= 12) device fp32 arrays in two launches: d_out[i] = min(d_ins[i][0 .. counts[i])).  d_ins / counts are HOST arrays
 * (the pointers travel in the kernel arguments); d_ws: m3d_reduce_min_multi_workspace_bytes() bytes. */
M3D_API size_t m3d_reduce_min_multi_workspace_bytes(void) { return sizeof(float) * kMinMulti * kMinBlocks; }

M3D_API int m3d_reduce_min_multi(const float* const* d_ins, const int64_t* counts, int count, float* d_out, void* d_ws, size_t ws_bytes,
                                 void* stream) {
  if (count < 0 || count > kMinMulti) return M3D_EINVAL;
  if (count == 0) return M3D_OK;
  if (!d_ins || !counts || !d_out || !d_ws) return M3D_EINVAL;
  if (ws_bytes < sizeof(float) * kMinMulti * kMinBlocks) return M3D_EWORKSPACE;
  MinMultiArgs a;
  for (int i = 0; i < kMinMulti; ++i) {
    const int k = i < count ? i : 0;
    if (!d_ins[k] || counts[k] <= 0) return M3D_EINVAL;
    a.in[i] = d_ins[k]; a.n[i] = counts[k];
  }
  hipLaunchKernelGGL(min_partial_multi_kernel, dim3(kMinBlocks, count), dim3(256), 0, m3d::as_stream(stream), a, (float*)d_ws);
  hipLaunchKernelGGL(min_final_multi_kernel, dim3(count), dim3(64), 0, m3d::as_stream(stream), (const float*)d_ws, d_out);
  return m3d::check_launch("reduce_min_multi");
}

/* Minima and maxima of `count` (<= 12) device fp32 arrays in two launches (m3d_reduce_min_multi's contract; d_mins / d_maxs [count]). */
M3D_API size_t m3d_reduce_minmax_multi_workspace_bytes(void) { return sizeof(float) * 2 * kMinMulti * kMinBlocks; }

M3D_API int m3d_reduce_minmax_multi(const float* const* d_ins, const int64_t* counts, int count, float* d_mins, float* d_maxs, void* d_ws,
                                    size_t ws_bytes, void* stream) {
  if (count < 0 || count > kMinMulti) return M3D_EINVAL;
  if (count == 0) return M3D_OK;
  if (!d_ins || !counts || !d_mins || !d_maxs || !d_ws) return M3D_EINVAL;
  if (ws_bytes < sizeof(float) * 2 * kMinMulti * kMinBlocks) return M3D_EWORKSPACE;
  MinMultiArgs a;
  for (int i = 0; i < kMinMulti; ++i) {
    const int k = i < count ? i : 0;
    if (!d_ins[k] || counts[k] <= 0) return M3D_EINVAL;
    a.in[i] = d_ins[k]; a.n[i] = counts[k];
  }
  hipLaunchKernelGGL(minmax_partial_multi_kernel, dim3(kMinBlocks, count), dim3(256), 0, m3d::as_stream(stream), a, (float*)d_ws);
  hipLaunchKernelGGL(minmax_final_multi_kernel, dim3(count), dim3(64), 0, m3d::as_stream(stream), (const float*)d_ws, d_mins, d_maxs);
  return m3d::check_launch("reduce_minmax_multi");
}
