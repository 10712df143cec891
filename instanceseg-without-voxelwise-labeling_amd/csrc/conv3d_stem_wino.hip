// conv1a (lib/modeling/DSN.py:19: Conv3d(1, 32, k=5, pad=2) + BN + ReLU, followed by MaxPool3d(2,2) at DSN.py:58) with
// Winograd F(2,5) along x, fp32 MFMA for gfx950: 6 instead of 10 multiplies per output pair and (dz,dy) row ->
// 0.62x the matrix-core work of conv3d.hip's direct stem kernel (K = 25 rows x 6 xi vs 125 taps).
//
//   d0..d5 = in[2t-2 .. 2t+3] of row (z+dz-2, y+dy-2),  g0..g4 = the row's five x taps
//   v = B^T d:  v0 = 4d0-5d2+d4   v1 = -4d1-4d2+d3+d4   v2 = 4d1-4d2-d3+d4   v3 = -2d1-d2+2d3+d4   v4 = 2d1-d2-2d3+d4   v5 = 4d1-5d3+d5
//   u = G g  (interpolation points 0, +-1, +-2, inf; computed offline in fp64, stored fp32)
//   m = u * v summed over the 25 rows;   y[2t] = m0+m1+m2+m3+m4,   y[2t+1] = m1-m2+2m3-2m4+m5
// fp32 error of this transform on stem-like data: max 4.8e-6 abs on outputs of magnitude ~1.4 (direct fp32 sum: 1.9e-6;
// simulated in NumPy before the kernel was written), i.e. ~1e-6 relative to the tensor maximum.
//
// GEMM view per xi: i = 32 output channels (A = u, packed [13 row pairs][6 xi][lane]), j = 32 x-pairs of one (z, y) output
// row, k = 2 (dz,dy) rows (lanes 0-31 take row 2p, lanes 32-63 row 2p+1; row 25 is a zero pad).  Cin = 1, so the whole K
// dimension (13 steps) and all weights (20 KB) sit in LDS at once: one staging phase per workgroup, no chunk loop.
// A workgroup = 4 waves = 2 (z) x 2 (y) output rows x 64 x = one pooling row; 96 accumulator registers per wave, four
// workgroups per CU cover each other's staging and epilogue.
#include <stdlib.h>

#include "m3d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NP = 13;                 // row pairs (25 rows + 1 zero row)
constexpr int SW_ELEMS = NP * 6 * 64;  // packed weights of one 32-channel block (floats)

// Wp[cb][pair][xi][lane] = (G g_row)[xi], co = cb*32 + (lane&31), row = 2*pair + (lane>>5) = dz*5 + dy
__global__ __launch_bounds__(256) void stem_wino_pack_kernel(const float* __restrict__ w, int cout, float* __restrict__ wp, int ncb) {
  const int total = ncb * SW_ELEMS;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int lane = e & 63;
    const int xi = (e >> 6) % 6;
    const int pair = (e >> 6) / 6 % NP;
    const int cb = e / SW_ELEMS;
    const int co = cb * 32 + (lane & 31), row = 2 * pair + (lane >> 5);
    float v = 0.f;
    if (co < cout && row < 25) {
      const float* g = w + (size_t)co * 125 + row * 5;
      const double g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4];
      double u;
      switch (xi) {
        case 0: u = g0 / 4.0; break;
        case 1: u = -(g0 + g1 + g2 + g3 + g4) / 6.0; break;
        case 2: u = -(g0 - g1 + g2 - g3 + g4) / 6.0; break;
        case 3: u = (g0 + 2.0 * g1 + 4.0 * g2 + 8.0 * g3 + 16.0 * g4) / 24.0; break;
        case 4: u = (g0 - 2.0 * g1 + 4.0 * g2 - 8.0 * g3 + 16.0 * g4) / 24.0; break;
        default: u = g4; break;
      }
      v = (float)u;
    }
    wp[e] = v;
  }
}

struct SEpi {
  const float* scale;
  const float* shift;
  int relu;
};

constexpr int TXo = 64, EPs = TXo / 2 + 2;       // 34 entries per plane: E[u] = in[x0+2u-2], O[u] = in[x0+2u-1]
constexpr int HXs = 2 * EPs;                      // 68 floats per halo row
constexpr int QRs = HXs / 4;                      // 17 quads per row
constexpr int HYs = 2 + 4, HZs = 2 + 4;           // 2x2 (z,y) output rows + 4 halo
constexpr int IN_S = HXs * HYs * HZs;             // 2448 floats
constexpr int NQs = HYs * HZs * QRs;              // 612 quads
constexpr int NIs = (NQs + 255) / 256;            // 3 per thread
constexpr int NWs = (SW_ELEMS / 4 + 255) / 256;   // 5 float4 per thread (4992 floats)
constexpr int DUMPs = IN_S + SW_ELEMS;            // dump slot for out-of-tile quads
constexpr int SMEM_S = IN_S + SW_ELEMS + 40;

template <bool POOL>
__global__ __launch_bounds__(256, POOL ? 4 : 3) void conv3d_stem_wino_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                                  float* __restrict__ out, int cout, int D, int H, int W,
                                                                  int tiles_x, int tiles_y, int tiles_z, SEpi ep) {
  __shared__ __attribute__((aligned(16))) float lds[SMEM_S];
  float* lds_w = lds + IN_S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wz = wave >> 1, wy = wave & 1;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int tz = bid % tiles_z;
  const int cot = bid / tiles_z;
  const int b = blockIdx.y;
  const int x0 = tx * TXo, y0 = ty * 2, z0 = tz * 2;
  const size_t DHW = (size_t)D * H * W;
  const float* in_b = in + (size_t)b * DHW;

  // ---- stage the halo tile (16-byte quads through buffer loads, x de-interleaved) and the 20 KB of weights
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0,
                                                                         (unsigned)(DHW * sizeof(float)), 0x00020000);
  f32x4 rin[NIs];
  int mq[NIs], lq[NIs];
#pragma unroll
  for (int i = 0; i < NIs; ++i) {
    const int e = tid + i * 256;
    int m = 0, l = DUMPs, voff = 0;
    if (e < NQs) {
      const int q = e % QRs, row = e / QRs;
      const int hy = row % HYs, hz = row / HYs;
      const int z = z0 + hz - 2, y = y0 + hy - 2, xf = x0 - 2 + 4 * q;
      const bool rok = (z >= 0) & (z < D) & (y >= 0) & (y < H);
#pragma unroll
      for (int j = 0; j < 4; ++j) m |= (rok && xf + j >= 0 && xf + j < W) ? (1 << j) : 0;
      long long lin = ((long long)z * H + y) * W + xf;
      if (rok && lin < 0) { m |= (int)(-lin) << 4; lin = 0; }      // head of the tensor: load x = 0..3, shift right by 1 or 2
      voff = rok ? (int)(lin * 4) : 0;
      l = row * HXs + 2 * q;
    }
    mq[i] = m; lq[i] = l;
    rin[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
  }
  f32x4 rw[NWs];
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp + (size_t)cot * SW_ELEMS);
#pragma unroll
  for (int i = 0; i < NWs; ++i) {
    int e = tid + i * 256;
    if (e >= SW_ELEMS / 4) e = SW_ELEMS / 4 - 1;
    rw[i] = wp4[e];
  }
#pragma unroll
  for (int i = 0; i < NIs; ++i) {
    const int m = mq[i];
    f32x4 v = rin[i];
    const int sh = m >> 4;
    if (sh == 1) v = f32x4{0.f, v[0], v[1], v[2]};
    else if (sh == 2) v = f32x4{0.f, 0.f, v[0], v[1]};
    const f32x2 ev = {(m & 1) ? v[0] : 0.f, (m & 4) ? v[2] : 0.f};     // x0-2+4q, +2  -> E[2q], E[2q+1]
    const f32x2 ov = {(m & 2) ? v[1] : 0.f, (m & 8) ? v[3] : 0.f};     // +1, +3       -> O[2q], O[2q+1]
    *reinterpret_cast<f32x2*>(lds + lq[i]) = ev;
    *reinterpret_cast<f32x2*>(lds + lq[i] + EPs) = ov;
  }
#pragma unroll
  for (int i = 0; i < NWs; ++i) {
    const int e = tid + i * 256;
    if (e < SW_ELEMS / 4) reinterpret_cast<f32x4*>(lds_w)[e] = rw[i];
  }
  __syncthreads();

  f32x16 acc[6];
#pragma unroll
  for (int x = 0; x < 6; ++x)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[x][g] = 0.f;

  const int jt = lane & 31, hi = lane >> 5;
  // this wave's output row is (z0+wz, y0+wy): halo row of (dz,dy) = (wz+dz)*HYs + (wy+dy)
  const float* base = lds + (wz * HYs + wy) * HXs + jt;
  auto row_off = [&](int r) __attribute__((always_inline)) -> int {      // r = dz*5 + dy (r = 25: zero weights, any valid row)
    const int rr = r < 25 ? r : 24;
    return ((rr / 5) * HYs + rr % 5) * HXs;
  };
  auto read_raw = [&](int p, float (&d)[6]) __attribute__((always_inline)) {
    const float* q = base + (hi ? row_off(2 * p + 1) : row_off(2 * p));
    d[0] = q[0]; d[1] = q[EPs]; d[2] = q[1]; d[3] = q[EPs + 1]; d[4] = q[2]; d[5] = q[EPs + 2];   // E[t],O[t],E[t+1],O[t+1],E[t+2],O[t+2]
  };
  auto transform = [&](const float (&d)[6], float (&v)[6]) __attribute__((always_inline)) {
    const float a = d[4] - 4.f * d[2], bq = d[3] - 4.f * d[1];
    const float c = d[4] - d[2], e2 = 2.f * (d[3] - d[1]);
    v[0] = 4.f * d[0] + (d[4] - 5.f * d[2]);
    v[1] = a + bq; v[2] = a - bq;
    v[3] = c + e2; v[4] = c - e2;
    v[5] = 4.f * d[1] + (d[5] - 5.f * d[3]);
  };
  float raw[2][6], bf[2][6], af[2][6];
  read_raw(0, raw[0]);
  read_raw(1, raw[1]);
#pragma unroll
  for (int x = 0; x < 6; ++x) af[0][x] = lds_w[(0 * 6 + x) * 64 + lane];
  transform(raw[0], bf[0]);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (p + 1 < NP) transform(raw[(p + 1) & 1], bf[(p + 1) & 1]);
    if (p + 2 < NP) read_raw(p + 2, raw[p & 1]);
    if (p + 1 < NP) {
#pragma unroll
      for (int x = 0; x < 6; ++x) af[(p + 1) & 1][x] = lds_w[((p + 1) * 6 + x) * 64 + lane];
    }
#pragma unroll
    for (int x = 0; x < 6; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[p & 1][x], bf[p & 1][x], acc[x], 0, 0, 0);
  }

  // ---- inverse transform, scale/shift, ReLU; x pair in the lane
  const int co0 = cot * 32 + 4 * hi;
  f32x16 y0v = acc[0] + acc[1] + acc[2] + acc[3] + acc[4];
  f32x16 y1v = (acc[1] - acc[2]) + 2.f * (acc[3] - acc[4]) + acc[5];
  const int z = z0 + wz, y = y0 + wy, x = x0 + 2 * jt;
  if constexpr (POOL) {
    __syncthreads();                                   // everyone is done with the tile: reuse it for the 2x2 (z,y) exchange
    float* red = lds + (size_t)wave * 16 * 64 + lane;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = min(co0 + (g & 3) + 8 * (g >> 2), cout - 1);
      const float sc = ep.scale ? ep.scale[co] : 1.f, sh = ep.shift ? ep.shift[co] : 0.f;
      float v0 = y0v[g] * sc + sh, v1 = y1v[g] * sc + sh;
      if (ep.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      red[g * 64] = fmaxf(v0, v1);
    }
    __syncthreads();
    const int PD = D / 2, PH = H / 2, PW = W / 2;
    const int zp = z0 >> 1, yp = y0 >> 1;
    // 32 channels x 32 pooled x = 1024 outputs, 4 per thread: thread -> (channel i = tid / 8 .. , x run)
    for (int o = tid; o < 32 * 32; o += 256) {
      const int t = o & 31, i = o >> 5;
      const int h2 = (i >> 2) & 1, g = (i & 3) | ((i >> 3) << 2);
      const int co = cot * 32 + i, xp = (x0 >> 1) + t;
      if (co >= cout || zp >= PD || yp >= PH || xp >= PW) continue;
      float m = -INFINITY;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) m = fmaxf(m, lds[((size_t)w4 * 16 + g) * 64 + h2 * 32 + t]);
      out[((size_t)b * cout + co) * ((size_t)PD * PH * PW) + ((size_t)zp * PH + yp) * PW + xp] = m;
    }
    return;
  } else {
    if (!(z < D && y < H && x < W)) return;
    const bool pair_ok = ((W & 1) == 0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = co0 + (g & 3) + 8 * (g >> 2);
      if (co >= cout) continue;
      const float sc = ep.scale ? ep.scale[co] : 1.f, sh = ep.shift ? ep.shift[co] : 0.f;
      float v0 = y0v[g] * sc + sh, v1 = y1v[g] * sc + sh;
      if (ep.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      float* o = out + ((size_t)b * cout + co) * DHW + ((size_t)z * H + y) * W + x;
      if (pair_ok) {
        *reinterpret_cast<f32x2*>(o) = f32x2{v0, v1};
      } else {
        o[0] = v0;
        if (x + 1 < W) o[1] = v1;
      }
    }
  }
}

template <bool POOL>
int launch_stem_wino(const float* in, const float* wp, float* out, int B, int cout, int D, int H, int W, SEpi ep, hipStream_t st) {
  const int tiles_x = (W + TXo - 1) / TXo, tiles_y = (H + 1) / 2, tiles_z = (D + 1) / 2;
  const int co_tiles = (cout + 31) / 32;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * co_tiles;
  if (blocks > 0x7FFFFFFFll || B > 65535) return M3D_EUNSUPPORTED;
  hipLaunchKernelGGL(conv3d_stem_wino_kernel<POOL>, dim3((unsigned)blocks, B), dim3(256), 0, st, in, wp, out, cout, D, H, W, tiles_x,
                     tiles_y, tiles_z, ep);
  return m3d::check_launch("conv3d_stem_wino");
}

}  // namespace

M3D_API size_t m3d_conv3d_stem_wino_packed_weight_bytes(int cout) {
  return cout <= 0 ? 0 : sizeof(float) * (size_t)((cout + 31) / 32) * SW_ELEMS;
}

M3D_API int m3d_conv3d_stem_wino_pack_weights(const float* d_weight /*[cout,1,5,5,5]*/, int cout, float* d_packed, void* stream) {
  if (!d_weight || !d_packed || cout <= 0) return M3D_EINVAL;
  hipLaunchKernelGGL(stem_wino_pack_kernel, dim3(64), dim3(256), 0, m3d::as_stream(stream), d_weight, cout, d_packed, (cout + 31) / 32);
  return m3d::check_launch("stem_wino_pack");
}

M3D_API int m3d_conv3d_stem_wino_forward(const float* d_in, const float* d_packed, float* d_out, int batch, int cout, int depth,
                                         int height, int width, const float* d_scale, const float* d_shift, int relu, int pool,
                                         void* stream) {
  if (!d_in || !d_packed || !d_out || batch <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if ((size_t)depth * height * width * sizeof(float) >= 0x7FFFFFFFull) return M3D_EUNSUPPORTED;
  if (width < 32) return M3D_EUNSUPPORTED;           // 64-wide tiles: narrower maps use the direct stem kernel
  if (pool && (depth < 2 || height < 2 || width < 2)) return M3D_EINVAL;
  SEpi ep{d_scale, d_shift, relu};
  hipStream_t st = m3d::as_stream(stream);
  return pool ? launch_stem_wino<true>(d_in, d_packed, d_out, batch, cout, depth, height, width, ep, st)
              : launch_stem_wino<false>(d_in, d_packed, d_out, batch, cout, depth, height, width, ep, st);
}
