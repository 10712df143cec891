// Backward-data of a 3x3x3 conv on batches of SMALL windows (3^3, 5^3, 7^3 voxels) on the f16 matrix cores at fp32 accuracy: the GEMM of
// prm_small.hip (N = (peak, voxel) flattened densely over the batch, M = 32 output channels, K = (input channel, tap);
//     out[p, co, v] = (X[co][origin_p + v] - off) * sum_{ci, t} relu(W)[ci][co][26 - t] * G_N[p, ci][v + t - 1],
// lib/prm/peak_response_mapping_3d.py:157-172 with lib/prm/peak_backprop_3d.py:8-44) with the "f16x2 split" of fc_gemm.hip: every fp32
// operand is scaled by a power of two and cut into two fp16 numbers (22 significand bits), three v_mfma_f32_32x32x16_f16 products
// (lo.hi, hi.lo, hi.hi) accumulate in fp32 what one fp32 MFMA step does, at 16 / 3 of the fp32 matrix rate.
//   * the gradient windows of different peaks differ by orders of magnitude (each starts from its own (1 - y) y), so the scale is PER
//     PEAK: peak_absmax_kernel sweeps every peak's [Cin, V] block once per layer; a column's accumulator is un-scaled with its own
//     peak's factor in the epilogue.  A column never sees another peak's data or scale: a sub-batch computes the batch's rows bit for bit.
//   * relu(W): one scale per layer from its largest entry, at pack time (kept behind the packed planes).
// Layout: K = 16 input channels per MFMA at one tap (lane half h = 8 channels).  LDS per 16-channel chunk:
//   B image  [plane hi / lo][h][peak slot][padded voxel] units of 8 fp16 (16 B): a tap is an immediate offset, consecutive columns read
//            consecutive units (ds_read_b128, no bank conflicts), the one-voxel border stays zero for the whole kernel;
//   W image  [tap][plane][lane] units: the A fragments of the workgroup's 32 output channels.
// One workgroup = 256 columns x 32 output channels, 8 waves, one accumulator block each; the next chunk's global loads are issued
// before the chunk's 81 MFMAs per wave and cut / written to LDS after them.
#include "m3d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kC16 = 16;                        // input channels per chunk = K of one MFMA
constexpr int kWUnits = 27 * 2 * 64;            // 16-byte units of one (cout block, chunk) weight image
// WAVES = 8 (5^3, 7^3): 256 columns per workgroup, two waves per SIMD (one's MFMAs beside the other's staging), the 55 KB weight image shared
// by twice the columns (5^3 0.219 -> 0.182 ms, 7^3 0.280 -> 0.226 at 67 peaks); 3^3 batches are too small for that (64 workgroups): WAVES = 4.
// LDS <= 148 KB in every case.

// max |x| of every peak's block: amax[p] (as float bits ordered like unsigned) over gn[p * n .. (p + 1) * n)
__global__ __launch_bounds__(256) void peak_absmax_kernel(const float* __restrict__ gn, long long n, float* __restrict__ amax) {
  const float* g = gn + (size_t)blockIdx.x * n;
  unsigned m = 0;
  for (long long e = threadIdx.x; e < n; e += 256) { const unsigned b = __float_as_uint(g[e]) & 0x7FFFFFFFu; m = b > m ? b : m; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
  __shared__ unsigned wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) amax[blockIdx.x] = __uint_as_float(max(max(wm[0], wm[1]), max(wm[2], wm[3])));
}

// wamax = max relu(W); one workgroup (the weights of a layer: <= 1.8 M floats, once per model)
__global__ __launch_bounds__(1024) void relu_absmax_kernel(const float* __restrict__ w, long long n, float* __restrict__ out) {
  unsigned m = 0;
  for (long long e = threadIdx.x; e < n; e += 1024) { const float v = w[e]; const unsigned b = v > 0.f ? __float_as_uint(v) : 0u; m = b > m ? b : m; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
  __shared__ unsigned wm[16];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) { for (int k = 1; k < 16; ++k) m = max(m, wm[k]); *out = __uint_as_float(m); }
}

// packed[cb][chunk][tap][plane][lane] (8 fp16 each) = relu(W[ci][co][26 - t]) * s,  co = 32 cb + (lane & 31) (an INPUT channel of the forward
// conv), ci = 16 chunk + 8 (lane >> 5) + j (an OUTPUT channel of the forward conv); W: [cout_fwd][cin_fwd][27]
__global__ __launch_bounds__(256) void small_pack_f16_kernel(const float* __restrict__ w, int cout_fwd, int cin_fwd, u32x4* __restrict__ packed,
                                                             int nchunk, long long total, const float* __restrict__ wamax) {
  float sw, inv;
  m3d::f16_scale_of(*wamax, sw, inv);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int lane = (int)(e & 63);
    long long r = e >> 6;
    const int t = (int)(r % 27); r /= 27;
    const int chunk = (int)(r % nchunk);
    const int cb = (int)(r / nchunk);
    const int co = 32 * cb + (lane & 31);
    u32x4 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16x2 hh, ll;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ci = kC16 * chunk + 8 * (lane >> 5) + 2 * j + u;
        float v = 0.f;
        if (co < cin_fwd && ci < cout_fwd) {
          v = w[((size_t)ci * cin_fwd + co) * 27 + (26 - t)];
          v = v > 0.f ? v * sw : 0.f;
        }
        const _Float16 h = (_Float16)v;
        hh[u] = h; ll[u] = (_Float16)(v - (float)h);
      }
      ph[j] = __builtin_bit_cast(unsigned, hh); pl[j] = __builtin_bit_cast(unsigned, ll);
    }
    u32x4* dst = packed + ((size_t)(cb * nchunk + chunk) * 27 + t) * 128 + lane;
    dst[0] = ph; dst[64] = pl;
  }
}

struct SmallF16Args {
  const float* gn;        // [P, Cin, V]
  const u32x4* wp;        // packed weights
  const float* wamax;     // largest relu(W) (the weight scale's source)
  const float* pamax;     // [P] largest |gn[p]|
  const float* full;      // [Cout, D, H, W]  X of this layer (PreHook multiply)
  const float* full_off;  // scalar
  const int* origins;     // [P, 3] window origin in X coordinates
  float* out;             // [P, Cout, V]
  int P, cin, cout, nchunk, D, H, W;
};

template <int WN, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void prm_small_dgrad_f16_kernel(SmallF16Args q) {
  constexpr int kNT = 64 * WAVES, kCols = 32 * WAVES;
  constexpr int V = WN * WN * WN, PW = WN + 2, CSB = PW * PW * PW;
  constexpr int PK = (kCols - 1) / V + 2;                  // peaks a workgroup's columns can touch
  constexpr int BPL = 2 * PK * CSB;                        // units of one B plane: [h][slot][padded voxel]
  extern __shared__ float sm_f[];
  u32x4* const lB = reinterpret_cast<u32x4*>(sm_f);        // [plane][h][slot][CSB]
  u32x4* const lW = lB + 2 * BPL;                          // [tap][plane][lane]
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, h = l >> 5, nl = l & 31;
  const int cb = blockIdx.y;
  const long long n0 = (long long)blockIdx.x * kCols, ntot = (long long)q.P * V;
  const int p_first = (int)(n0 / V);
  const long long n = n0 + 32 * w + nl;
  const bool col_ok = n < ntot;
  const int p = col_ok ? (int)(n / V) : p_first;
  const int v = col_ok ? (int)(n - (long long)p * V) : 0;
  const int vz = v / (WN * WN), vy = (v / WN) % WN, vx = v % WN;
  const int bunit = (h * PK + (p - p_first)) * CSB + (vz * PW + vy) * PW + vx;    // the (-1,-1,-1) corner of the 3^3 stencil, plane hi

  for (int e = tid; e < 2 * BPL; e += kNT) lB[e] = u32x4{0u, 0u, 0u, 0u};         // the borders (and absent peaks) stay zero

  // v_mfma_f32_32x32x16_f16 truncates when it adds its products to the accumulator (conv3d_x3.hip): with operands of one sign - post-ReLU
  // gradients, relu(W) - a sum kept in the accumulator over all of K drifts low by ~3e-9 of itself per MFMA, 4e-6 at K = 256 x 27 x 3.
  // The accumulator therefore restarts every second chunk (162 MFMAs) and its runs are added with round-to-nearest adds.
  f32x16 acc, tot;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc[e] = 0.f; tot[e] = 0.f; }

  // ---- staging plan (constant over the chunks): weight units and (slot, voxel, 8-channel group) tasks
  constexpr int NW = (kWUnits + kNT - 1) / kNT;
  constexpr int NIN = (2 * PK * V + kNT - 1) / kNT;
  int g_src[NIN], g_dst[NIN];
  float g_s[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int e = tid + kNT * i;
    g_src[i] = -1; g_dst[i] = -1; g_s[i] = 1.f;
    if (e < 2 * PK * V) {
      const int g = e / (PK * V), r = e - g * (PK * V);
      const int s = r / V, vv = r - s * V;
      const int pp = p_first + s;
      const int z = vv / (WN * WN), y = (vv / WN) % WN, x = vv % WN;
      g_dst[i] = (g * PK + s) * CSB + ((z + 1) * PW + y + 1) * PW + x + 1;
      if (pp < q.P) {
        g_src[i] = (int)(((long long)s * q.cin + 8 * g) * V + vv);
        float inv;
        m3d::f16_scale_of(q.pamax[pp], g_s[i], inv);
      }
    }
  }
  const float* const gbase = q.gn + (size_t)p_first * q.cin * V;
  const u32x4* const wbase = q.wp + (size_t)cb * q.nchunk * kWUnits;
  u32x4 sw[NW];
  float sg[NIN][8];
  auto fetch = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + kNT * i;
      sw[i] = wbase[(size_t)ch * kWUnits + (e < kWUnits ? e : 0)];
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const float* ptr = gbase + (g_src[i] >= 0 ? (size_t)g_src[i] + (size_t)kC16 * ch * V : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) sg[i][j] = ptr[(size_t)j * V];
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + kNT * i;
      if (e < kWUnits) lW[e] = sw[i];
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      if (g_dst[i] < 0 || g_src[i] < 0) continue;
      u32x4 ph, pl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = sg[i][2 * j] * g_s[i], v1 = sg[i][2 * j + 1] * g_s[i];
        const f16x2 hh = {(_Float16)v0, (_Float16)v1};
        const f16x2 ll = {(_Float16)(v0 - (float)hh[0]), (_Float16)(v1 - (float)hh[1])};
        ph[j] = __builtin_bit_cast(unsigned, hh); pl[j] = __builtin_bit_cast(unsigned, ll);
      }
      lB[g_dst[i]] = ph;
      lB[BPL + g_dst[i]] = pl;
    }
  };
  fetch(0);
  __syncthreads();
  commit();
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < q.nchunk; ++ch) {
    if (ch + 1 < q.nchunk) fetch(ch + 1);                                                 // in flight under this chunk's MFMAs
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int toff = ((t / 9) * PW + (t / 3) % 3) * PW + t % 3;
      const f16x8 ah = __builtin_bit_cast(f16x8, lW[(t * 2 + 0) * 64 + l]);
      const f16x8 al = __builtin_bit_cast(f16x8, lW[(t * 2 + 1) * 64 + l]);
      const f16x8 bh = __builtin_bit_cast(f16x8, lB[bunit + toff]);
      const f16x8 bl = __builtin_bit_cast(f16x8, lB[BPL + bunit + toff]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);                // small products first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
    __syncthreads();                                                                      // this chunk's LDS reads are done
    if (ch + 1 < q.nchunk) commit();
    if ((ch & 1) || ch + 1 == q.nchunk) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { tot[e] += acc[e]; acc[e] = 0.f; }
    }
    __syncthreads();
  }
  // un-scale (powers of two: exact), PreHook multiply (peak_backprop_3d.py:16-18) and store; accumulator register e holds output
  // channel 8*(e/4) + 4*h + e%4
  if (!col_ok) return;
  float sp, inv_p, swt, inv_w;
  m3d::f16_scale_of(q.pamax[p], sp, inv_p);
  m3d::f16_scale_of(*q.wamax, swt, inv_w);
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
      q.out[((size_t)p * q.cout + co) * V + v] = in ? m * ((tot[e] * inv_p) * inv_w) : 0.f;
    }
  }
}

template <int WN, int WAVES>
int launch_small_f16(const SmallF16Args& q, hipStream_t st) {
  constexpr int kNT = 64 * WAVES, kCols = 32 * WAVES;
  constexpr int V = WN * WN * WN, PW = WN + 2, CSB = PW * PW * PW, PK = (kCols - 1) / V + 2;
  const size_t lds = 16 * ((size_t)4 * PK * CSB + kWUnits);
  const long long ntot = (long long)q.P * V;
  auto kern = prm_small_dgrad_f16_kernel<WN, WAVES>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)((ntot + kCols - 1) / kCols), (q.cout + 31) / 32), dim3(kNT), lds, st, q);
  return m3d::check_launch("prm_small_dgrad_f16");
}

inline size_t f16_plane_bytes(int cout_fwd, int cin_fwd) { return (size_t)((cin_fwd + 31) / 32) * (cout_fwd / kC16) * kWUnits * 16; }

}  // namespace

M3D_API int m3d_prm_small_dgrad_f16_supported(int cout_fwd, int cin_fwd) { return (cout_fwd > 0 && cin_fwd > 0 && cout_fwd % kC16 == 0) ? 1 : 0; }

M3D_API size_t m3d_prm_small_dgrad_f16_packed_bytes(int cout_fwd, int cin_fwd) {
  if (!m3d_prm_small_dgrad_f16_supported(cout_fwd, cin_fwd)) return 0;
  return f16_plane_bytes(cout_fwd, cin_fwd) + 256;               // + the largest relu(W) behind the planes
}

M3D_API int m3d_prm_small_dgrad_f16_pack(const float* d_weight, int cout_fwd, int cin_fwd, void* d_packed, void* stream) {
  if (!d_weight || !d_packed || cout_fwd <= 0 || cin_fwd <= 0) return M3D_EINVAL;
  if (!m3d_prm_small_dgrad_f16_supported(cout_fwd, cin_fwd) || ((uintptr_t)d_packed & 15)) return M3D_EUNSUPPORTED;
  hipStream_t st = m3d::as_stream(stream);
  float* wamax = reinterpret_cast<float*>(static_cast<char*>(d_packed) + f16_plane_bytes(cout_fwd, cin_fwd));
  hipLaunchKernelGGL(relu_absmax_kernel, dim3(1), dim3(1024), 0, st, d_weight, (long long)cout_fwd * cin_fwd * 27, wamax);
  const int nchunk = cout_fwd / kC16;
  const long long total = (long long)((cin_fwd + 31) / 32) * nchunk * 27 * 64;
  hipLaunchKernelGGL(small_pack_f16_kernel, dim3(1024), dim3(256), 0, st, d_weight, cout_fwd, cin_fwd, reinterpret_cast<u32x4*>(d_packed), nchunk,
                     total, (const float*)wamax);
  return m3d::check_launch("prm_small_dgrad_f16_pack");
}

M3D_API size_t m3d_prm_small_dgrad_f16_workspace_bytes(int num_peaks) { return num_peaks > 0 ? (size_t)num_peaks * sizeof(float) + 256 : 256; }

/* d_gn [P, cout_fwd, win^3] -> d_out [P, cin_fwd, win^3]; win in {3, 5, 7}; cout_fwd a multiple of 16 */
M3D_API int m3d_prm_small_dgrad_f16(const float* d_gn, const void* d_packed, int num_peaks, int cout_fwd, int cin_fwd, int win,
                                    const float* d_full, const float* d_full_offset, const int32_t* d_origins, int depth, int height,
                                    int width, float* d_out, void* d_ws, size_t ws_bytes, void* stream) {
  if (num_peaks < 0 || cout_fwd <= 0 || cin_fwd <= 0) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_gn || !d_packed || !d_full || !d_full_offset || !d_origins || !d_out || !d_ws || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if ((win != 3 && win != 5 && win != 7) || !m3d_prm_small_dgrad_f16_supported(cout_fwd, cin_fwd) || num_peaks > 65535) return M3D_EUNSUPPORTED;
  if (ws_bytes < (size_t)num_peaks * sizeof(float)) return M3D_EWORKSPACE;
  hipStream_t st = m3d::as_stream(stream);
  float* pamax = reinterpret_cast<float*>(d_ws);
  hipLaunchKernelGGL(peak_absmax_kernel, dim3(num_peaks), dim3(256), 0, st, d_gn, (long long)cout_fwd * win * win * win, pamax);
  SmallF16Args q;
  q.gn = d_gn; q.wp = reinterpret_cast<const u32x4*>(d_packed);
  q.wamax = reinterpret_cast<const float*>(static_cast<const char*>(d_packed) + f16_plane_bytes(cout_fwd, cin_fwd));
  q.pamax = pamax; q.full = d_full; q.full_off = d_full_offset; q.origins = d_origins; q.out = d_out; q.P = num_peaks;
  q.cin = cout_fwd; q.cout = cin_fwd; q.nchunk = cout_fwd / kC16; q.D = depth; q.H = height; q.W = width;
  if (win == 3) return launch_small_f16<3, 4>(q, st);
  if (win == 5) return launch_small_f16<5, 8>(q, st);
  return launch_small_f16<7, 8>(q, st);
}
