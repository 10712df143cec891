// Backward-data of a 3x3x3 conv on batches of SMALL windows (3^3, 5^3, 7^3 voxels) for the peak back-propagation: the stride-8 / 4
// stages of lib/prm/peak_response_mapping_3d.py:157-172 with the rule of lib/prm/peak_backprop_3d.py:8-44
//     out[p, co, v] = (X[co][origin_p + v] - off) * sum_{ci, t} relu(W)[ci][co][26 - t] * G_N[p, ci][v + t - 1]      (zero padding).
// The direct kernel tiles one window at a time (8 x 8 x 4 voxels: 10 % / 49 % / 67 % of a tile is useful for 3^3 / 5^3 / 7^3 and
// most CUs idle).  Here the peaks are batched into the GEMM N dimension: N = (peak, voxel) flattened DENSELY over the whole batch,
// M = 32 output channels, K = (input channel, tap).  A workgroup owns 128 consecutive columns (they may span several peaks) and
// 32 output channels; each of its 4 waves owns one 32-column block, i.e. one accumulator block: the batch becomes thousands of
// equal waves (several per SIMD, which also hides the staging), instead of a few hundred fat ones.
// Per chunk of 4 input channels the windows of the peaks a workgroup touches sit in LDS with a one-voxel zero border (no bounds
// tests in the K loop: a tap is an immediate offset), next to the chunk's weights in A-operand order.
#include "m3d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCC = 4;                        // input channels per chunk
constexpr int kWSeg = (kCC / 2) * 27 * 64;    // floats of one (cout block, chunk) weight segment: [cp][tap][lane]

// wp[cb][chunk][cp][t][lane] = relu(W[ci][co][26 - t]),  co = 32 cb + (lane & 31) (an INPUT channel of the forward conv),
// ci = 4 chunk + 2 cp + (lane >> 5) (an OUTPUT channel of the forward conv); W: [cout_fwd][cin_fwd][27]
__global__ __launch_bounds__(256) void small_pack_kernel(const float* __restrict__ w, int cout_fwd, int cin_fwd, float* __restrict__ wp,
                                                         int nchunk, long long total) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int lane = (int)(e & 63);
    long long r = e >> 6;
    const int t = (int)(r % 27); r /= 27;
    const int cp = (int)(r % (kCC / 2)); r /= (kCC / 2);
    const int chunk = (int)(r % nchunk);
    const int cb = (int)(r / nchunk);
    const int co = 32 * cb + (lane & 31), ci = kCC * chunk + 2 * cp + (lane >> 5);
    float v = 0.f;
    if (co < cin_fwd && ci < cout_fwd) {
      v = w[((size_t)ci * cin_fwd + co) * 27 + (26 - t)];
      v = v > 0.f ? v : 0.f;
    }
    wp[e] = v;
  }
}

struct SmallArgs {
  const float* gn;        // [P, Cin, V]
  const float* wp;        // packed weights
  const float* full;      // [Cout, D, H, W]  X of this layer (PreHook multiply)
  const float* full_off;  // scalar
  const int* origins;     // [P, 3] window origin in X coordinates
  float* out;             // [P, Cout, V]
  int P, cin, cout, nchunk, D, H, W;
};

template <int WN>
__global__ __launch_bounds__(256, 2) void prm_small_dgrad_kernel(SmallArgs q) {
  constexpr int V = WN * WN * WN, PW = WN + 2, CSB = PW * PW * PW;
  constexpr int PK = 127 / V + 2;                          // peaks a 128-column block can touch
  extern __shared__ float sm[];
  float* const lin = sm;                                   // [PK][kCC][CSB]
  float* const lw = sm + PK * kCC * CSB;                   // [kWSeg]
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, h = l >> 5, nl = l & 31;
  const int cb = blockIdx.y;
  const long long n0 = (long long)blockIdx.x * 128, ntot = (long long)q.P * V;
  const int p_first = (int)(n0 / V);
  // this lane's column
  const long long n = n0 + 32 * w + nl;
  const bool col_ok = n < ntot;
  const int p = col_ok ? (int)(n / V) : p_first;
  const int v = col_ok ? (int)(n - (long long)p * V) : 0;
  const int vz = v / (WN * WN), vy = (v / WN) % WN, vx = v % WN;
  const int bbase = (p - p_first) * kCC * CSB + h * CSB + (vz * PW + vy) * PW + vx;     // the (-1,-1,-1) corner of the 3^3 stencil

  for (int e = tid; e < PK * kCC * CSB; e += 256) lin[e] = 0.f;                           // the borders stay zero for the whole kernel

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  const f32x4* wsrc = reinterpret_cast<const f32x4*>(q.wp + (size_t)cb * q.nchunk * kWSeg);
  constexpr int NW4 = (kWSeg / 4 + 255) / 256;                                            // float4 per thread and chunk
  constexpr int NIN = (PK * kCC * V + 255) / 256;                                         // interior voxels per thread and chunk
#pragma unroll 1
  for (int ch = 0; ch < q.nchunk; ++ch) {
    __syncthreads();                                                                      // previous chunk's reads are done
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int e = tid + 256 * i;
      if (e < kWSeg / 4) reinterpret_cast<f32x4*>(lw)[e] = wsrc[(size_t)ch * (kWSeg / 4) + e];
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int e = tid + 256 * i;
      if (e < PK * kCC * V) {
        const int vv = e % V, r = e / V;
        const int cc = r % kCC, s = r / kCC;
        const int pp = p_first + s, c = kCC * ch + cc;
        float g = 0.f;
        if (pp < q.P && c < q.cin) g = q.gn[((size_t)pp * q.cin + c) * V + vv];
        const int z = vv / (WN * WN), y = (vv / WN) % WN, x = vv % WN;
        lin[(s * kCC + cc) * CSB + ((z + 1) * PW + y + 1) * PW + x + 1] = g;
      }
    }
    __syncthreads();
#pragma unroll
    for (int cp = 0; cp < kCC / 2; ++cp)
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const float a = lw[(cp * 27 + t) * 64 + l];
        const float b = lin[bbase + cp * 2 * CSB + ((t / 9) * PW + (t / 3) % 3) * PW + t % 3];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
  }
  // PreHook multiply (peak_backprop_3d.py:16-18) and store; accumulator register e holds output channel 8*(e/4) + 4*h + e%4
  if (!col_ok) return;
  const int qz = q.origins[3 * p] + vz, qy = q.origins[3 * p + 1] + vy, qx = q.origins[3 * p + 2] + vx;
  const bool in = (qz >= 0) & (qz < q.D) & (qy >= 0) & (qy < q.H) & (qx >= 0) & (qx < q.W);
  const size_t pos = in ? ((size_t)qz * q.H + qy) * q.W + qx : 0;
  const size_t DHW = (size_t)q.D * q.H * q.W;
  const float off = *q.full_off;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = 32 * cb + 8 * (e >> 2) + 4 * h + (e & 3);
    if (co < q.cout) {
      const float m = in ? q.full[(size_t)co * DHW + pos] - off : 0.f;
      q.out[((size_t)p * q.cout + co) * V + v] = in ? m * acc[e] : 0.f;
    }
  }
}

template <int WN>
int launch_small(const SmallArgs& q, hipStream_t st) {
  constexpr int V = WN * WN * WN, PW = WN + 2, CSB = PW * PW * PW, PK = 127 / V + 2;
  const size_t lds = sizeof(float) * ((size_t)PK * kCC * CSB + kWSeg);
  const long long ntot = (long long)q.P * V;
  auto kern = prm_small_dgrad_kernel<WN>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)((ntot + 127) / 128), (q.cout + 31) / 32), dim3(256), lds, st, q);
  return m3d::check_launch("prm_small_dgrad");
}

}  // namespace

M3D_API size_t m3d_prm_small_dgrad_packed_bytes(int cout_fwd, int cin_fwd) {
  if (cout_fwd <= 0 || cin_fwd <= 0) return 0;
  return sizeof(float) * (size_t)((cin_fwd + 31) / 32) * ((cout_fwd + kCC - 1) / kCC) * kWSeg;
}

M3D_API int m3d_prm_small_dgrad_pack(const float* d_weight, int cout_fwd, int cin_fwd, float* d_packed, void* stream) {
  if (!d_weight || !d_packed || cout_fwd <= 0 || cin_fwd <= 0) return M3D_EINVAL;
  const int nchunk = (cout_fwd + kCC - 1) / kCC;
  const long long total = (long long)((cin_fwd + 31) / 32) * nchunk * kWSeg;
  hipLaunchKernelGGL(small_pack_kernel, dim3(1024), dim3(256), 0, m3d::as_stream(stream), d_weight, cout_fwd, cin_fwd, d_packed, nchunk,
                     total);
  return m3d::check_launch("prm_small_dgrad_pack");
}

/* d_gn [P, cout_fwd, win^3] -> d_out [P, cin_fwd, win^3]; win in {3, 5, 7} */
M3D_API int m3d_prm_small_dgrad(const float* d_gn, const float* d_packed, int num_peaks, int cout_fwd, int cin_fwd, int win,
                                const float* d_full, const float* d_full_offset, const int32_t* d_origins, int depth, int height,
                                int width, float* d_out, void* stream) {
  if (num_peaks < 0 || cout_fwd <= 0 || cin_fwd <= 0) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_gn || !d_packed || !d_full || !d_full_offset || !d_origins || !d_out || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (win != 3 && win != 5 && win != 7) return M3D_EUNSUPPORTED;
  SmallArgs q;
  q.gn = d_gn; q.wp = d_packed; q.full = d_full; q.full_off = d_full_offset; q.origins = d_origins; q.out = d_out; q.P = num_peaks;
  q.cin = cout_fwd; q.cout = cin_fwd; q.nchunk = (cout_fwd + kCC - 1) / kCC; q.D = depth; q.H = height; q.W = width;
  hipStream_t st = m3d::as_stream(stream);
  if (win == 3) return launch_small<3>(q, st);
  if (win == 5) return launch_small<5>(q, st);
  return launch_small<7>(q, st);
}
