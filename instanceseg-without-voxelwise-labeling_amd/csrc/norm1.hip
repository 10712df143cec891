// norm1 volume pre-processing on the device (HBM-bound: one 2-byte read x3 + one 4-byte write per voxel).
//   mask = im > 0;  im = (im - mean(im[mask])) / std(im[mask])
// Reference call sites: lib/utils/blob.py:179-184 (prep_im_for_blob, float32: the detection branch via
// lib/core/test.py:1020-1027) and tools/infer_simple.py:180-183 (PRM branch, float64, crops cast to float32 at :217).
// The reference does this with NumPy on the host and ships the fp32 tile over PCIe; here the RAW uint16 volume is what
// crosses PCIe (half the bytes) and the statistics are exact two-pass fp64 sums in a fixed order (deterministic: per-block
// partials reduced in index order by every block - no atomics).
#include "m3d_common.h"

namespace {

constexpr int kBlocks = 1024;   // partials per pass

template <typename T>
__device__ inline double load_as_double(const T* p, long long i) { return (double)p[i]; }

__device__ inline double block_sum(double v, double* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return r;
}

// every block reduces the kBlocks partials itself, in the same order -> the same value everywhere
__device__ inline double reduce_partials(const double* __restrict__ part, double* sm) {
  double v = 0.0;
  for (int e = threadIdx.x; e < kBlocks; e += 256) v += part[e];
  return block_sum(v, sm);
}

template <typename T>
__global__ __launch_bounds__(256) void norm1_sum_kernel(const T* __restrict__ in, long long n, double* __restrict__ ws) {
  __shared__ double sm[4];
  in += (size_t)blockIdx.y * n; ws += (size_t)blockIdx.y * 3 * kBlocks;          // blockIdx.y: volume of the batch
  double s = 0.0, c = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)kBlocks * 256) {
    const double v = load_as_double(in, e);
    if (v > 0.0) { s += v; c += 1.0; }
  }
  s = block_sum(s, sm);
  c = block_sum(c, sm);
  if (threadIdx.x == 0) { ws[blockIdx.x] = s; ws[kBlocks + blockIdx.x] = c; }
}

template <typename T>
__global__ __launch_bounds__(256) void norm1_var_kernel(const T* __restrict__ in, long long n, double* __restrict__ ws) {
  __shared__ double sm[4];
  in += (size_t)blockIdx.y * n; ws += (size_t)blockIdx.y * 3 * kBlocks;
  const double mean = reduce_partials(ws, sm) / reduce_partials(ws + kBlocks, sm);
  double q = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)kBlocks * 256) {
    const double v = load_as_double(in, e);
    if (v > 0.0) { const double d = v - mean; q += d * d; }
  }
  q = block_sum(q, sm);
  if (threadIdx.x == 0) ws[2 * kBlocks + blockIdx.x] = q;
}

// f32_arith = 1: (float(x) - float(mean)) / float(std) in fp32 (blob.py:179-184 works on a float32 array);
// f32_arith = 0: fp64 arithmetic then one rounding to fp32 (infer_simple.py:180-183 + the astype(np.float32) of :217).
template <typename T>
__global__ __launch_bounds__(256) void norm1_apply_kernel(const T* __restrict__ in, long long n, const double* __restrict__ ws,
                                                          int f32_arith, float* __restrict__ out, double* __restrict__ stats) {
  __shared__ double sm[4];
  in += (size_t)blockIdx.y * n; ws += (size_t)blockIdx.y * 3 * kBlocks; out += (size_t)blockIdx.y * n;
  if (stats) stats += 3 * blockIdx.y;
  const double cnt = reduce_partials(ws + kBlocks, sm);
  const double mean = reduce_partials(ws, sm) / cnt;
  const double sd = sqrt(reduce_partials(ws + 2 * kBlocks, sm) / cnt);      // np.std: population (ddof = 0)
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = mean; stats[1] = sd; stats[2] = cnt; }
  const float mf = (float)mean, sf = (float)sd;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    if (f32_arith) out[e] = ((float)in[e] - mf) / sf;
    else out[e] = (float)((load_as_double(in, e) - mean) / sd);
  }
}

template <typename T>
int run(const T* in, int batch, long long n, int f32_arith, float* out, double* ws, double* stats, hipStream_t st) {
  hipLaunchKernelGGL(norm1_sum_kernel<T>, dim3(kBlocks, batch), dim3(256), 0, st, in, n, ws);
  hipLaunchKernelGGL(norm1_var_kernel<T>, dim3(kBlocks, batch), dim3(256), 0, st, in, n, ws);
  long long blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(norm1_apply_kernel<T>, dim3((unsigned)blocks, batch), dim3(256), 0, st, in, n, (const double*)ws, f32_arith, out, stats);
  return m3d::check_launch("norm1");
}

}  // namespace

M3D_API size_t m3d_norm1_workspace_bytes(void) { return 3 * kBlocks * sizeof(double); }

M3D_API int m3d_norm1(const void* d_in, int in_dtype, int64_t n, int f32_arith, float* d_out, double* d_stats, void* d_ws,
                      size_t ws_bytes, void* stream) {
  if (!d_in || !d_out || !d_ws || n <= 0) return M3D_EINVAL;
  if (ws_bytes < m3d_norm1_workspace_bytes()) return M3D_EWORKSPACE;
  hipStream_t st = m3d::as_stream(stream);
  if (in_dtype == 0) return run((const uint16_t*)d_in, 1, (long long)n, f32_arith, d_out, (double*)d_ws, d_stats, st);
  if (in_dtype == 1) return run((const float*)d_in, 1, (long long)n, f32_arith, d_out, (double*)d_ws, d_stats, st);
  return M3D_EINVAL;
}

/* `batch` volumes of n voxels each, contiguous in d_in and d_out; every volume is normalised with its OWN mean / std
 * (blob.py:179-184 is called per image).  d_stats: [batch, 3] or null.  d_ws: batch * m3d_norm1_workspace_bytes(). */
M3D_API int m3d_norm1_batched(const void* d_in, int in_dtype, int batch, int64_t n, int f32_arith, float* d_out, double* d_stats,
                              void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_in || !d_out || !d_ws || n <= 0 || batch <= 0 || batch > 65535) return M3D_EINVAL;
  if (ws_bytes < (size_t)batch * m3d_norm1_workspace_bytes()) return M3D_EWORKSPACE;
  hipStream_t st = m3d::as_stream(stream);
  if (in_dtype == 0) return run((const uint16_t*)d_in, batch, (long long)n, f32_arith, d_out, (double*)d_ws, d_stats, st);
  if (in_dtype == 1) return run((const float*)d_in, batch, (long long)n, f32_arith, d_out, (double*)d_ws, d_stats, st);
  return M3D_EINVAL;
}
