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
  unsigned* out_max;     // or null: 32 slots that receive the largest |output| (atomic maxima; the operand bound of the f16x2 conv that follows)
#ifdef M3D_W2_STAMPS
  unsigned long long* stamps;
#endif
};
// diagnostic build only (make w2_stamps; tools/stem_stamps.py): s_memtime stamps of wave 0 of every workgroup of the rows kernel
#ifdef M3D_W2_STAMPS
static unsigned long long* g_stem_stamps = nullptr;
M3D_API void m3d_debug_set_stamp_buffer_stem(void* p) { g_stem_stamps = (unsigned long long*)p; }
#define STEM_STAMP(k) do { if (ep.stamps && threadIdx.x == 0) ep.stamps[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 16 + (k)] = \
    (k) >= 12 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STEM_STAMP(k) do { } while (0)
#endif

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

// ------------------------------------------------------------------------------------------------ round 3: the "rows" kernel
// The kernel above gives a workgroup ONE pooling row (4 waves x 78 MFMAs) and stages the block's 20 KB of transformed weights for it:
// 8192 workgroups per 128^3 volume fetch 164 MB of weights for 33 MB of input, every wave reads its six weight fragments from LDS in
// every K step, and four workgroups per CU leave 128 registers per wave.  Stamps of a first rows kernel (tools/stem_stamps.py) showed
// where the time of such a workgroup goes: K loops 39 %, epilogues (with the z-pair exchange and its barrier) 31 %, prologue 18 %.  So:
//   * the wave keeps its 78 weight fragments in REGISTERS (loaded once from global memory, 20 x dwordx4; no weight LDS at all);
//   * a wave owns one y PAIR of the tile and walks TZ planes with it (tile 64 x 8 y x TZ z, four waves = four y pairs; 256 registers, two
//     workgroups per CU): per K step 2 LDS reads (the six raw values of a row in natural x order), 12 transform VALU, 6 MFMAs;
//   * the 2x2x2 pooling window is complete in the wave - x pair in the lane, y pair and z pair in consecutive rows (running maximum
//     parked in the wave's own LDS slots, 16 more live registers would spill) - so there is NO barrier after the prologue and the
//     shift / ReLU run once per pooled value (max commutes with them; the scale is applied before the maximum, it may be negative);
//   * the halo tile is loaded once per 2 TZ rows of a wave: (TZ + 4) / TZ planes per output plane instead of 3.
constexpr int TY4 = 8, HY4 = TY4 + 4;
constexpr int HX4 = 68;                             // s = x - (x0 - 2), 0..67
constexpr int SW4_ELEMS = 20 * 64 * 4;              // register pack of one 32-channel block: [20 groups][lane][4]; slot q = pair*6 + xi
template <int TZ>
struct Rows {
  static constexpr int HZ = TZ + 4;
  static constexpr int IN = HX4 * HY4 * HZ;         // TZ = 4: 6528 floats, TZ = 8: 9792
  static_assert(HY4 * (HX4 / 4) <= 256, "one staging thread per (halo row, quad)");
};

// Wr[cb][g][lane][j] = (G g_row)[xi] for slot q = 4g + j = pair*6 + xi (q >= 78: 0), co = cb*32 + (lane&31), row = 2*pair + (lane>>5)
__global__ __launch_bounds__(256) void stem_wino_pack4_kernel(const float* __restrict__ w, int cout, float* __restrict__ wp, int ncb) {
  const int total = ncb * SW4_ELEMS;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int j = e & 3, lane = (e >> 2) & 63, g = (e >> 8) % 20, cb = e / SW4_ELEMS;
    const int q = 4 * g + j, pair = q / 6, xi = q % 6;
    const int co = cb * 32 + (lane & 31), row = 2 * pair + (lane >> 5);
    float v = 0.f;
    if (q < 78 && co < cout && row < 25) {
      const float* gw = w + (size_t)co * 125 + row * 5;
      const double g0 = gw[0], g1 = gw[1], g2 = gw[2], g3 = gw[3], g4 = gw[4];
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

template <bool POOL, int TZ>
__global__ __launch_bounds__(256, 2) void conv3d_stem_wino4_kernel(const float* __restrict__ in, const float* __restrict__ wr,
                                                                  float* __restrict__ out, int cout, int D, int H, int W,
                                                                  int tiles_x, int tiles_y, int tiles_z, SEpi ep) {
  using C = Rows<TZ>;
  __shared__ __attribute__((aligned(16))) float lds[C::IN + 8 + 64 + (POOL ? 4 * 16 * 64 : 0)];
  float* const aff = lds + C::IN + 8;                  // scale[32], shift[32] of the block (1 / 0 where absent or beyond cout)
  float* const park = aff + 64 + (threadIdx.x >> 6) * (16 * 64) + (threadIdx.x & 63);      // POOL: running maximum, per wave
  STEM_STAMP(0); STEM_STAMP(12);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int tz = bid % tiles_z;
  const int cot = bid / tiles_z;
  const int b = blockIdx.y;
  const int x0 = tx * TXo, y0 = ty * TY4, z0 = tz * TZ;
  const size_t DHW = (size_t)D * H * W;
  const float* in_b = in + (size_t)b * DHW;

  float vmaxt = 0.f;                                   // largest |stored value| of this thread (ep.out_max)
  // ---- the wave's 78 weight fragments -> registers (the block's pack is L2-resident: every workgroup reads the same 20 KB)
  f32x4 aw[20];
  {
    const f32x4* wp4 = reinterpret_cast<const f32x4*>(wr + (size_t)cot * SW4_ELEMS) + lane;
#pragma unroll
    for (int g = 0; g < 20; ++g) aw[g] = wp4[g * 64];
  }
  // ---- the halo tile -> LDS in natural x order.  A thread owns one 16-byte quad of one halo row (hy, q) and walks the TZ + 4 planes
  // with it: validity of the row and of its four x is per-thread constant, validity of the plane uniform, the addresses advance by
  // one plane - about 6 VALU per quad.  (The first version decoded a flat quad index per load, ~70 VALU each: stamps showed a
  // 29 000-cycle prologue, because a wave whose SIMD neighbour streams fp32 MFMAs gets a VALU slot only every MFMA or so.)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0,
                                                                         (unsigned)(DHW * sizeof(float)), 0x00020000);
  {
    const int hy = tid / (HX4 / 4), q = tid - hy * (HX4 / 4);
    const bool active = hy < HY4;
    const int y = y0 + hy - 2, xf = x0 - 2 + 4 * q;
    const bool yok = active && y >= 0 && y < H;
    const bool m0 = yok && xf >= 0 && xf < W, m1 = yok && xf + 1 >= 0 && xf + 1 < W, m2 = yok && xf + 2 < W, m3 = yok && xf + 3 < W;
    const bool head = yok && y == 0 && xf < 0;                        // z = 0 too: the quad starts 2 floats before the tensor
    const int vrow = (y * W + xf) * 4;                                // byte offset inside a plane (may be negative; clamped per load)
    const int plane = H * W * 4;
    float* const ldst = lds + hy * HX4 + 4 * q;
    f32x4 rin[C::HZ];
#pragma unroll
    for (int i = 0; i < C::HZ; ++i)       // every plane is loaded, valid or not: below the tensor the offset clamps to 0, beyond it
                                          // the buffer's range check returns 0; the masks below zero what is not input
      rin[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, max(vrow + (z0 + i - 2) * plane, 0), 0, 0));
#pragma unroll
    for (int i = 0; i < C::HZ; ++i) {
      const int z = z0 + i - 2;
      const bool zok = z >= 0 && z < D;
      f32x4 v = rin[i];
      if (z == 0 && head) v = f32x4{0.f, 0.f, v[0], v[1]};
      f32x4 o = {(m0 && zok) ? v[0] : 0.f, (m1 && zok) ? v[1] : 0.f, (m2 && zok) ? v[2] : 0.f, (m3 && zok) ? v[3] : 0.f};
      asm volatile("" : "+v"(o));                          // one register tuple -> ds_write_b128 (a split store conflicts 4-way)
      if (active) *reinterpret_cast<f32x4*>(__builtin_assume_aligned(ldst + i * (HY4 * HX4), 16)) = o;
    }
  }
  if (tid < 64) {
    const int co = cot * 32 + (tid & 31);
    const float* src = tid < 32 ? ep.scale : ep.shift;
    aff[tid] = (src && co < cout) ? src[co] : (tid < 32 ? 1.f : 0.f);
  }
  __syncthreads();
  STEM_STAMP(1);

  const int jt = lane & 31, hi = lane >> 5;
  // Row addressing.  K pair p = rows (2p, 2p+1) of r = dz*5 + dy; lanes 0-31 take row 2p, lanes 32-63 row 2p+1.  The halo row of
  // (dz, dy) for output row (zz, yl) is (zz + dz)*HY4 + yl + dy.  For ten of the thirteen pairs the second row is the NEXT halo row,
  // for p = 2, 7 (dy = 4 -> next dz, dy = 0) it lies HY4 - 4 rows on, for p = 12 the second row is the zero pad (read the first row
  // again): three per-lane base registers, every step an immediate offset.
  const float* const tile = lds + 2 * jt;
  const int co0 = cot * 32 + 4 * hi;
  const int PD = D / 2, PH = H / 2, PW = W / 2;
#pragma unroll 1
  for (int rr = 0; rr < 2 * TZ; ++rr) {
    const int zz = rr >> 1, yl = 2 * wave + (rr & 1);
    const float* b0 = tile + (size_t)(zz * HY4 + yl) * HX4;      // both halves the same row (pad pair)
    const float* b1 = b0 + (hi ? HX4 : 0);                       // second row = next halo row
    const float* b2 = b0 + (hi ? (HY4 - 4) * HX4 : 0);           // second row = first row of the next z plane
    f32x16 acc[6];
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[x][g] = 0.f;
    auto read_raw = [&](int p, float (&d)[6]) __attribute__((always_inline)) {
      const int r0 = 2 * p;
      const int off = ((r0 / 5) * HY4 + r0 % 5) * HX4;
      const float* q = (p == 12 ? b0 : (p == 2 || p == 7) ? b2 : b1) + off;
      const f32x2 u = *reinterpret_cast<const f32x2*>(q), v = *reinterpret_cast<const f32x2*>(q + 2), w2 = *reinterpret_cast<const f32x2*>(q + 4);
      d[0] = u[0]; d[1] = u[1]; d[2] = v[0]; d[3] = v[1]; d[4] = w2[0]; d[5] = w2[1];      // in[2t-2 .. 2t+3]
    };
    // B^T d of F(2,5), every product fused explicitly: left to the compiler the contraction differed between the pooled and the
    // unpooled instantiation (1-ulp different activations) and the uncontracted one carried 40 more instructions per row
    auto transform = [&](const float (&d)[6], float (&v)[6]) __attribute__((always_inline)) {
      const float a = __builtin_fmaf(-4.f, d[2], d[4]), bq = __builtin_fmaf(-4.f, d[1], d[3]);
      const float c = d[4] - d[2], e2 = d[3] - d[1];
      v[0] = __builtin_fmaf(4.f, d[0], __builtin_fmaf(-5.f, d[2], d[4]));
      v[1] = a + bq; v[2] = a - bq;
      v[3] = __builtin_fmaf(2.f, e2, c); v[4] = __builtin_fmaf(-2.f, e2, c);
      v[5] = __builtin_fmaf(4.f, d[1], __builtin_fmaf(-5.f, d[3], d[5]));
    };
    float raw[2][6], bf[2][6];
    read_raw(0, raw[0]);
    read_raw(1, raw[1]);
    transform(raw[0], bf[0]);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p + 2 < NP) read_raw(p + 2, raw[p & 1]);
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const int q = p * 6 + x;
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[q >> 2][q & 3], bf[p & 1][x], acc[x], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (p + 1 < NP) {
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(raw[(p + 1) & 1][i]));      // keep the transform out of the read's shadow
        transform(raw[(p + 1) & 1], bf[(p + 1) & 1]);
      }
#pragma unroll
      for (int x = 3; x < 6; ++x) {
        const int q = p * 6 + x;
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[q >> 2][q & 3], bf[p & 1][x], acc[x], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    STEM_STAMP(rr < 4 ? 2 + 2 * rr : 10);
    // ---- inverse transform; x pair in the lane
    const f32x16 y0v = acc[0] + acc[1] + acc[2] + acc[3] + acc[4];
    const f32x16 y1v = (acc[1] - acc[2]) + 2.f * (acc[3] - acc[4]) + acc[5];
    const int z = z0 + zz, y = y0 + yl, x = x0 + 2 * jt;
    // scale / shift of the lane's 16 channels (co0 + (g&3) + 8*(g>>2)): four 16-byte LDS reads each.  Read per row - held across the K
    // loop they would cost 32 registers and spill; fetched per channel from global memory (round-3 first version) the sixteen
    // dependent load-branch-wait sequences took 12 000 cycles per storing row.
    int ao = 4 * hi;
    asm volatile("" : "+v"(ao));                       // (the offset, not the pointer: an opaque pointer becomes a flat load)
    const float* affl = aff + ao;
    const bool all8 = (cout & 7) == 0;                 // then a group of four channels is valid or not as a whole (uniform test)
    if constexpr (POOL) {
      float cur[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(affl + 8 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[4 * q + j] = fmaxf(y0v[4 * q + j] * sc[j], y1v[4 * q + j] * sc[j]);
      }
      if ((rr & 3) != 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) cur[g] = fmaxf(cur[g], park[g * 64]);
      }
      if ((rr & 3) != 3) {
#pragma unroll
        for (int g = 0; g < 16; ++g) park[g * 64] = cur[g];
      } else {
        const int py = y >> 1, pz = z >> 1, px = (x0 >> 1) + jt;
        if (pz < PD && py < PH && px < PW) {
          const size_t PDHW = (size_t)PD * PH * PW;
          float* ob = out + ((size_t)b * cout + co0) * PDHW + ((size_t)pz * PH + py) * PW + px;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 sh = *reinterpret_cast<const f32x4*>(affl + 32 + 8 * q);
            if (cot * 32 + 8 * q >= cout) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = cur[4 * q + j] + sh[j];
              if (ep.relu) v = fmaxf(v, 0.f);
              if (all8 || co0 + 8 * q + j < cout) { ob[(size_t)(8 * q + j) * PDHW] = v; vmaxt = fmaxf(vmaxt, fabsf(v)); }
            }
          }
        }
      }
    } else {
      if (z < D && y < H && x < W) {
        const bool pair_ok = ((W & 1) == 0);
        float* ob = out + ((size_t)b * cout + co0) * DHW + ((size_t)z * H + y) * W + x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(affl + 8 * q), sh = *reinterpret_cast<const f32x4*>(affl + 32 + 8 * q);
          if (cot * 32 + 8 * q >= cout) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // product and sum rounded separately, as the pooled path must (maximum between them): pooled == max_pool3d(unpooled) bit for bit
            float p0 = y0v[4 * q + j] * sc[j], p1 = y1v[4 * q + j] * sc[j];
            asm volatile("" : "+v"(p0), "+v"(p1));       // (no contraction into an FMA; __fmul_rn is a plain product to this compiler)
            float v0 = p0 + sh[j], v1 = p1 + sh[j];
            if (ep.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            if (!(all8 || co0 + 8 * q + j < cout)) continue;
            float* o = ob + (size_t)(8 * q + j) * DHW;
            vmaxt = fmaxf(vmaxt, fabsf(v0));
            if (pair_ok) {
              *reinterpret_cast<f32x2*>(o) = f32x2{v0, v1};
              vmaxt = fmaxf(vmaxt, fabsf(v1));
            } else {
              o[0] = v0;
              if (x + 1 < W) { o[1] = v1; vmaxt = fmaxf(vmaxt, fabsf(v1)); }
            }
          }
        }
      }
    }
    STEM_STAMP(rr < 4 ? 3 + 2 * rr : 11);
  }
  if (ep.out_max) {                                      // one atomic per wave, spread over the 32 slots
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmaxt = fmaxf(vmaxt, __shfl_xor(vmaxt, o));
    if (lane == 0 && vmaxt > 0.f) atomicMax(ep.out_max + ((blockIdx.x * 4 + wave) & 31), __float_as_uint(vmaxt));
  }
  STEM_STAMP(13);
}

template <bool POOL, int TZ>
int launch_stem_rows(const float* in, const float* wr, float* out, int B, int cout, int D, int H, int W, SEpi ep, hipStream_t st) {
  const int tiles_x = (W + TXo - 1) / TXo, tiles_y = (H + TY4 - 1) / TY4, tiles_z = (D + TZ - 1) / TZ;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * ((cout + 31) / 32);
  if (blocks > 0x7FFFFFFFll) return M3D_EUNSUPPORTED;
  hipLaunchKernelGGL((conv3d_stem_wino4_kernel<POOL, TZ>), dim3((unsigned)blocks, B), dim3(256), 0, st, in, wr, out, cout, D, H, W, tiles_x,
                     tiles_y, tiles_z, ep);
  return m3d::check_launch("conv3d_stem_wino4");
}

template <bool POOL>
int launch_stem_wino(const float* in, const float* wp, float* out, int B, int cout, int D, int H, int W, SEpi ep, hipStream_t st) {
  const int co_tiles = (cout + 31) / 32;
  if (B > 65535) return M3D_EUNSUPPORTED;
  // option "tune_stem": 1 = the one-row kernel of round 2 (A/B measurements), 4 / 8 = rows kernel with that many planes per workgroup
  const int tune = m3d::opt(m3d::OPT_TUNE_STEM);
  if (tune != 1) {
    const float* wr = wp + (size_t)co_tiles * SW_ELEMS;
    // eight planes per workgroup while that still leaves every CU its two workgroups twice over
    const long long wg8 = (long long)((W + TXo - 1) / TXo) * ((H + TY4 - 1) / TY4) * ((D + 7) / 8) * co_tiles * B;
    const bool tz8 = tune == 8 || (tune != 4 && wg8 >= 2 * 512);
    return tz8 ? launch_stem_rows<POOL, 8>(in, wr, out, B, cout, D, H, W, ep, st)
               : launch_stem_rows<POOL, 4>(in, wr, out, B, cout, D, H, W, ep, st);
  }
  const int tiles_x = (W + TXo - 1) / TXo, tiles_y = (H + 1) / 2, tiles_z = (D + 1) / 2;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * co_tiles;
  if (blocks > 0x7FFFFFFFll) return M3D_EUNSUPPORTED;
  hipLaunchKernelGGL(conv3d_stem_wino_kernel<POOL>, dim3((unsigned)blocks, B), dim3(256), 0, st, in, wp, out, cout, D, H, W, tiles_x,
                     tiles_y, tiles_z, ep);
  return m3d::check_launch("conv3d_stem_wino");
}

}  // namespace

M3D_API size_t m3d_conv3d_stem_wino_packed_weight_bytes(int cout) {
  return cout <= 0 ? 0 : sizeof(float) * (size_t)((cout + 31) / 32) * (SW_ELEMS + SW4_ELEMS);      // both kernels' packs
}

M3D_API int m3d_conv3d_stem_wino_pack_weights(const float* d_weight /*[cout,1,5,5,5]*/, int cout, float* d_packed, void* stream) {
  if (!d_weight || !d_packed || cout <= 0) return M3D_EINVAL;
  const int ncb = (cout + 31) / 32;
  hipLaunchKernelGGL(stem_wino_pack_kernel, dim3(64), dim3(256), 0, m3d::as_stream(stream), d_weight, cout, d_packed, ncb);
  hipLaunchKernelGGL(stem_wino_pack4_kernel, dim3(64), dim3(256), 0, m3d::as_stream(stream), d_weight, cout,
                     d_packed + (size_t)ncb * SW_ELEMS, ncb);
  return m3d::check_launch("stem_wino_pack");
}

M3D_API int m3d_conv3d_stem_wino_forward_bound(const float* d_in, const float* d_packed, float* d_out, int batch, int cout, int depth,
                                               int height, int width, const float* d_scale, const float* d_shift, int relu, int pool,
                                               float* d_out_max, void* stream);
M3D_API int m3d_conv3d_stem_wino_forward(const float* d_in, const float* d_packed, float* d_out, int batch, int cout, int depth,
                                         int height, int width, const float* d_scale, const float* d_shift, int relu, int pool,
                                         void* stream) {
  return m3d_conv3d_stem_wino_forward_bound(d_in, d_packed, d_out, batch, cout, depth, height, width, d_scale, d_shift, relu, pool, nullptr, stream);
}

M3D_API int m3d_conv3d_stem_wino_forward_bound(const float* d_in, const float* d_packed, float* d_out, int batch, int cout, int depth,
                                               int height, int width, const float* d_scale, const float* d_shift, int relu, int pool,
                                               float* d_out_max, void* stream) {
  if (!d_in || !d_packed || !d_out || batch <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if ((size_t)depth * height * width * sizeof(float) >= 0x7FFFFFFFull) return M3D_EUNSUPPORTED;
  if (width < 32) return M3D_EUNSUPPORTED;           // 64-wide tiles: narrower maps use the direct stem kernel
  if (pool && (depth < 2 || height < 2 || width < 2)) return M3D_EINVAL;
  if (d_out_max && m3d::opt(m3d::OPT_TUNE_STEM) == 1) return M3D_EUNSUPPORTED;      // the one-row kernel of round 2 (A/B builds) has no bound output
  SEpi ep{d_scale, d_shift, relu, reinterpret_cast<unsigned*>(d_out_max)};
#ifdef M3D_W2_STAMPS
  ep.stamps = g_stem_stamps;
#endif
  hipStream_t st = m3d::as_stream(stream);
  return pool ? launch_stem_wino<true>(d_in, d_packed, d_out, batch, cout, depth, height, width, ep, st)
              : launch_stem_wino<false>(d_in, d_packed, d_out, batch, cout, depth, height, width, ep, st);
}
