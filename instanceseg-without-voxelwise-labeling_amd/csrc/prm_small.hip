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

#ifndef M3D_SMALL_OCC
#define M3D_SMALL_OCC 2
#endif
constexpr int kCC = 4;                        // input channels per chunk
constexpr int kTQ = 7;                        // tap quads: 27 taps padded to 28 (the 28th weight is zero and its MFMA is never issued)
constexpr int kWSeg = (kCC / 2) * kTQ * 64 * 4;   // floats of one (cout block, chunk) weight segment: [cp][tap quad][lane][4]

// wp[cb][chunk][cp][t / 4][lane][t % 4] = relu(W[ci][co][26 - t]),  co = 32 cb + (lane & 31) (an INPUT channel of the forward conv),
// ci = 4 chunk + 2 cp + (lane >> 5) (an OUTPUT channel of the forward conv); W: [cout_fwd][cin_fwd][27].  Four consecutive taps of a
// lane are one 16-byte LDS read (round 4; one ds_read_b32 per tap before).
__global__ __launch_bounds__(256) void small_pack_kernel(const float* __restrict__ w, int cout_fwd, int cin_fwd, float* __restrict__ wp,
                                                         int nchunk, long long total) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int tj = (int)(e & 3), lane = (int)((e >> 2) & 63);
    long long r = e >> 8;
    const int t = 4 * (int)(r % kTQ) + tj; r /= kTQ;
    const int cp = (int)(r % (kCC / 2)); r /= (kCC / 2);
    const int chunk = (int)(r % nchunk);
    const int cb = (int)(r / nchunk);
    const int co = 32 * cb + (lane & 31), ci = kCC * chunk + 2 * cp + (lane >> 5);
    float v = 0.f;
    if (co < cin_fwd && ci < cout_fwd && t < 27) {
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

// NCB = output-channel blocks of 32 per workgroup: a wave's B value feeds NCB MFMAs, and with the weights read four taps at a time the
// K loop issues (NCB + 4) / (4 NCB) LDS reads per MFMA - 1.25 at NCB = 1 - where round 2's kernel issued 2.  The next chunk's global loads (weights and window voxels) are issued into registers BEFORE the chunk's MFMAs and
// written to LDS after them: round 2 loaded global -> LDS between the two barriers of every chunk, with the latency exposed.
template <int WN, int NCB>
__global__ __launch_bounds__(256, M3D_SMALL_OCC) void prm_small_dgrad_kernel(SmallArgs q) {
  constexpr int V = WN * WN * WN, PW = WN + 2, CSB = PW * PW * PW;
  constexpr int PK = 127 / V + 2;                          // peaks a 128-column block can touch
  extern __shared__ float sm[];
  float* const lin = sm;                                   // [PK][kCC][CSB]
  float* const lw = sm + PK * kCC * CSB;                   // [NCB][kWSeg]
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, h = l >> 5, nl = l & 31;
  const int cb0 = blockIdx.y * NCB;
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

  f32x16 acc[NCB];
#pragma unroll
  for (int k = 0; k < NCB; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

  const int ncb_total = (q.cout + 31) / 32;
  constexpr int W4 = kWSeg / 4;                                                           // float4 of one segment
  constexpr int NW4 = (NCB * W4 + 255) / 256;                                             // float4 per thread and chunk
  constexpr int NIN = (PK * kCC * V + 255) / 256;                                         // interior voxels per thread and chunk
  // per-thread staging plan (constant over the chunks): weight quads and window voxels
  int w_src[NW4];                                                                         // float4 index inside the chunk, -1: none
#pragma unroll
  for (int i = 0; i < NW4; ++i) {
    const int e = tid + 256 * i;
    const int k = e / W4;
    w_src[i] = (e < NCB * W4 && cb0 + k < ncb_total) ? e : -1;
  }
  int g_src[NIN], g_dst[NIN];                                                             // gn offset of chunk 0 (-1: zero), LDS offset (-1: none)
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int e = tid + 256 * i;
    g_src[i] = -1; g_dst[i] = -1;
    if (e < PK * kCC * V) {
      const int vv = e % V, r = e / V;
      const int cc = r % kCC, s = r / kCC;
      const int pp = p_first + s;
      const int z = vv / (WN * WN), y = (vv / WN) % WN, x = vv % WN;
      g_dst[i] = (s * kCC + cc) * CSB + ((z + 1) * PW + y + 1) * PW + x + 1;
      if (pp < q.P) g_src[i] = (int)(((long long)(pp - p_first) * q.cin + cc) * V + vv);
    }
  }
  const float* const gbase = q.gn + (size_t)p_first * q.cin * V;
  const f32x4* const wbase = reinterpret_cast<const f32x4*>(q.wp);
  f32x4 sw[NW4];
  float sg[NIN];
  auto fetch = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int e = w_src[i] >= 0 ? w_src[i] : 0;
      const int k = e / W4, o = e - k * W4;
      sw[i] = wbase[((size_t)(cb0 + (w_src[i] >= 0 ? k : 0)) * q.nchunk + ch) * W4 + o];
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const bool ok = g_src[i] >= 0 && kCC * ch + (g_dst[i] / CSB) % kCC < q.cin;
      sg[i] = gbase[ok ? (size_t)g_src[i] + (size_t)kCC * ch * V : 0];
      if (!ok) sg[i] = 0.f;
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NW4; ++i)
      if (w_src[i] >= 0) reinterpret_cast<f32x4*>(lw)[w_src[i]] = sw[i];
#pragma unroll
    for (int i = 0; i < NIN; ++i)
      if (g_dst[i] >= 0) lin[g_dst[i]] = sg[i];
  };
  // blocks beyond the layer's channels (NCB = 2 with an odd block count): their weights stay zero
  for (int e = tid; e < NCB * kWSeg; e += 256) lw[e] = 0.f;
  fetch(0);
  __syncthreads();
  commit();
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < q.nchunk; ++ch) {
    if (ch + 1 < q.nchunk) fetch(ch + 1);                                                 // in flight under this chunk's MFMAs
#pragma unroll
    for (int cp = 0; cp < kCC / 2; ++cp)
#pragma unroll
      for (int tq = 0; tq < kTQ; ++tq) {
        f32x4 a4[NCB];
#pragma unroll
        for (int k = 0; k < NCB; ++k) a4[k] = *reinterpret_cast<const f32x4*>(lw + k * kWSeg + ((cp * kTQ + tq) * 64 + l) * 4);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const int t = 4 * tq + jt;
          if (t < 27) {
            const float b = lin[bbase + cp * 2 * CSB + ((t / 9) * PW + (t / 3) % 3) * PW + t % 3];
#pragma unroll
            for (int k = 0; k < NCB; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[k][jt], b, acc[k], 0, 0, 0);
          }
        }
      }
    __syncthreads();                                                                      // this chunk's LDS reads are done
    if (ch + 1 < q.nchunk) commit();
    __syncthreads();
  }
  // PreHook multiply (peak_backprop_3d.py:16-18) and store; accumulator register e holds output channel 8*(e/4) + 4*h + e%4
  if (!col_ok) return;
  const int qz = q.origins[3 * p] + vz, qy = q.origins[3 * p + 1] + vy, qx = q.origins[3 * p + 2] + vx;
  const bool in = (qz >= 0) & (qz < q.D) & (qy >= 0) & (qy < q.H) & (qx >= 0) & (qx < q.W);
  const size_t pos = in ? ((size_t)qz * q.H + qy) * q.W + qx : 0;
  const size_t DHW = (size_t)q.D * q.H * q.W;
  const float off = *q.full_off;
#pragma unroll
  for (int k = 0; k < NCB; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = 32 * (cb0 + k) + 8 * (e >> 2) + 4 * h + (e & 3);
      if (co < q.cout) {
        const float m = in ? q.full[(size_t)co * DHW + pos] - off : 0.f;
        q.out[((size_t)p * q.cout + co) * V + v] = in ? m * acc[k][e] : 0.f;
      }
    }
}

template <int WN, int NCB>
int launch_small_n(const SmallArgs& q, hipStream_t st) {
  constexpr int V = WN * WN * WN, PW = WN + 2, CSB = PW * PW * PW, PK = 127 / V + 2;
  const size_t lds = sizeof(float) * ((size_t)PK * kCC * CSB + (size_t)NCB * kWSeg);
  const long long ntot = (long long)q.P * V;
  auto kern = prm_small_dgrad_kernel<WN, NCB>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ncb = (q.cout + 31) / 32;
  hipLaunchKernelGGL(kern, dim3((unsigned)((ntot + 127) / 128), (ncb + NCB - 1) / NCB), dim3(256), lds, st, q);
  return m3d::check_launch("prm_small_dgrad");
}

template <int WN>
int launch_small(const SmallArgs& q, hipStream_t st) {
  // NCB = 2 (two output-channel blocks per workgroup: 0.75 instead of 1.25 LDS reads per MFMA) was measured and is NOT used: on the
  // nuclei tile's 67 peaks the 5^3 / 7^3 launches took 0.450 / 0.472 ms against 0.355 / 0.384 ms with one block per workgroup - these
  // launches want the occupancy (twice as many, half as heavy waves) more than they want fewer LDS reads
  return launch_small_n<WN, 1>(q, st);
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
