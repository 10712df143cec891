// 3-D convolution backward-weights (wgrad) as an fp32 MFMA implicit GEMM for gfx950, plus the bias gradient.
//   dW[co,ci,dz,dy,dx] = sum_{b,z,y,x} gy[b,co,z,y,x] * x[b,ci,z+dz-p,y+dy-p,x+dx-p]      (stride 1, pad k/2)
// Reference: the gradient autograd computes for F.conv3d in lib/prm/peak_backprop_3d.py:40-42 / every nn.Conv3d of
// lib/modeling/DSN.py:19-36 during training (SURVEY 8b "conv boundary": fwd / dgrad / wgrad).
//
// GEMM view: M = cout (32 per block), N = cin (32 per block) for one tap, K = voxels (v_mfma_f32_32x32x2f32, k = two
// x-adjacent voxels).  The reduction dimension is the voxel index, which is the CONTIGUOUS dimension of both operands
// in HBM, so both tiles are staged through LDS: coalesced reads along x, written as [channel][voxel] with an odd channel
// stride, then read back with lane = channel (conflict-free ds_read_b32).  One workgroup = (cout block, cin block,
// split-K slot); its 4 waves share the staged tiles and split the k^3 taps (k = 3) or the voxel pairs (k = 1).
// Split-K partials are written to a workspace and summed by a second kernel in a fixed order: deterministic.
#include <stdlib.h>

#include "m3d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TZ = 2, TY = 4, TX = 16, TV = TZ * TY * TX;   // voxel tile (128 voxels = 64 MFMA k-steps)
constexpr int GS = TV + 1;                                   // odd channel stride of the gy tile in LDS

template <int K>
struct WG {
  static constexpr int P = K / 2, HZ = TZ + 2 * P, HY = TY + 2 * P, HX = TX + 2 * P, HV = HZ * HY * HX;
  static constexpr int XS = HV | 1;                          // odd channel stride of the x halo tile
  static constexpr int TAPS = K * K * K;
  static constexpr int TW = TAPS >= 4 ? 4 : 1;               // waves splitting the taps
  static constexpr int VG = 4 / TW;                          // waves splitting the voxel pairs
  static constexpr int NTW = (TAPS + TW - 1) / TW;           // taps per wave (upper bound)
  static constexpr int LDS_FLOATS = 32 * GS + 32 * XS;
};

// partial[slot][co][ci][tap], slot = split * VG + voxel group
template <int K>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           float* __restrict__ partial, int B, int cin, int cout, int D, int H,
                                                           int W, int tiles_x, int tiles_y, int tiles_z, int S) {
  using C = WG<K>;
  extern __shared__ float lds[];
  float* lds_g = lds;
  float* lds_x = lds + 32 * GS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = wave % C::TW, vg = wave / C::TW;
  const int ci_blocks = (cin + 31) / 32;
  int bid = blockIdx.x;
  const int ib = bid % ci_blocks; bid /= ci_blocks;
  const int s = bid % S;
  const int cb = bid / S;
  const int NT = B * tiles_z * tiles_y * tiles_x;
  const size_t DHW = (size_t)D * H * W;

  f32x16 acc[C::NTW];
#pragma unroll
  for (int n = 0; n < C::NTW; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  int tapoff[C::NTW];
#pragma unroll
  for (int n = 0; n < C::NTW; ++n) {
    int t = tw + C::TW * n;
    if (t >= C::TAPS) t = C::TAPS - 1;                  // surplus slot of the last wave: computed, never stored
    const int dz = t / (K * K), dy = (t / K) % K, dx = t % K;
    tapoff[n] = (dz * C::HY + dy) * C::HX + dx;
  }
  const int cl = lane & 31, kh = lane >> 5;

  for (int t = s; t < NT; t += S) {
    int q = t;
    const int tx = q % tiles_x; q /= tiles_x;
    const int ty = q % tiles_y; q /= tiles_y;
    const int tz = q % tiles_z;
    const int b = q / tiles_z;
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    __syncthreads();                                   // the previous tile's MFMAs have read the LDS
    // ---- stage gy[cb*32 + c][tile] and x[ib*32 + c][halo tile] (zero padded) as 16-byte quads of four consecutive x:
    // one buffer_load_dwordx4 each (hardware range check returns 0 for channels past cout / cin; rows and columns outside
    // the volume are masked), all loads of a batch issued before the first LDS write.
    {
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(gy + (size_t)b * cout * DHW), 0, (unsigned)((size_t)cout * DHW * sizeof(float)), 0x00020000);
      constexpr int NQG = 32 * (TZ * TY) * (TX / 4);                   // 1024 quads
      constexpr int NG = NQG / 256;
      f32x4 v[NG];
      int mk[NG], ld[NG];
#pragma unroll
      for (int u = 0; u < NG; ++u) {
        const int e = tid + u * 256;
        const int q = e & 3, row = (e >> 2) & 7, c = e >> 5;
        const int zz = z0 + (row >> 2), yy = y0 + (row & 3), xf = x0 + 4 * q;
        const bool rok = (zz < D) & (yy < H);
        int m = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) m |= (rok && xf + j < W) ? (1 << j) : 0;
        const long long lin = (long long)(cb * 32 + c) * (long long)DHW + ((long long)zz * H + yy) * W + xf;
        mk[u] = m; ld[u] = c * GS + row * TX + 4 * q;
        v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, rok ? (int)(lin * 4) : 0, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < NG; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_g[ld[u] + j] = ((mk[u] >> j) & 1) ? v[u][j] : 0.f;
    }
    {
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(x + (size_t)b * cin * DHW), 0, (unsigned)((size_t)cin * DHW * sizeof(float)), 0x00020000);
      constexpr int QX = (C::HX + 3) / 4;                              // quads per halo row (x0-P .. x0-P+4*QX-1)
      constexpr int NQX = 32 * C::HZ * C::HY * QX;
      constexpr int UB = 5;                                            // quads per batch
#pragma unroll 1
      for (int e0 = 0; e0 < NQX; e0 += 256 * UB) {
        f32x4 v[UB];
        int mk[UB], ld[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int e = e0 + u * 256 + tid;
          const int q = e % QX, row = (e / QX) % (C::HZ * C::HY), c = e / (QX * C::HZ * C::HY);
          const int hz = row / C::HY, hy = row % C::HY;
          const int zz = z0 + hz - C::P, yy = y0 + hy - C::P, xf = x0 - C::P + 4 * q;
          const bool rok = (e < NQX) & ((unsigned)zz < (unsigned)D) & ((unsigned)yy < (unsigned)H);
          int m = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) m |= (rok && xf + j >= 0 && xf + j < W && 4 * q + j < C::HX) ? (1 << j) : 0;
          long long lin = (long long)(ib * 32 + c) * (long long)DHW + ((long long)zz * H + yy) * W + xf;
          if (rok && lin < 0) { m |= (int)(-lin) << 4; lin = 0; }      // a negative offset drops the whole quad: shift instead
          mk[u] = m; ld[u] = c * C::XS + row * C::HX + 4 * q;
          v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, rok ? (int)(lin * 4) : 0, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int sh = mk[u] >> 4;
          f32x4 w4 = v[u];
          if (sh == 1) w4 = f32x4{0.f, w4[0], w4[1], w4[2]};
          else if (sh == 2) w4 = f32x4{0.f, 0.f, w4[0], w4[1]};
          else if (sh == 3) w4 = f32x4{0.f, 0.f, 0.f, w4[0]};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((e0 + u * 256 + tid) < NQX && 4 * ((e0 + u * 256 + tid) % QX) + j < C::HX) lds_x[ld[u] + j] = ((mk[u] >> j) & 1) ? w4[j] : 0.f;
        }
      }
    }
    __syncthreads();
    // ---- MFMA: k-step = voxel pair (v, v+1); A = gy[co = lane%32][v + lane/32], B = x[ci = lane%32][v + lane/32 + tap]
    const float* ga = lds_g + cl * GS + kh;
    const float* xa = lds_x + cl * C::XS + kh;
#pragma unroll 4
    for (int i = 0; i < TV / 2 / C::VG; ++i) {
      const int v = (vg + i * C::VG) * 2;
      const float a = ga[v];
      const int hb = (((v >> 6)) * C::HY + ((v >> 4) & 3)) * C::HX + (v & 15);
      float bv[C::NTW];
#pragma unroll
      for (int n = 0; n < C::NTW; ++n) bv[n] = xa[hb + tapoff[n]];
#pragma unroll
      for (int n = 0; n < C::NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[n], acc[n], 0, 0, 0);
    }
  }
  // ---- write this workgroup's partial: acc[i = co][j = ci]; col j = lane&31, row i = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int slot = s * C::VG + vg;
  float* pp = partial + (size_t)slot * cout * cin * C::TAPS;
  const int ci = ib * 32 + cl;
#pragma unroll
  for (int n = 0; n < C::NTW; ++n) {
    const int tp = tw + C::TW * n;
    if (tp >= C::TAPS) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (co < cout && ci < cin) pp[((size_t)co * cin + ci) * C::TAPS + tp] = acc[n][r];
    }
  }
}

// Stem: cin = 1, k = 5.  N dimension = taps (125 -> 4 blocks of 32, one per wave): B[k][j] = x[v_k + shift(tap_j)].
constexpr int SP = 2, SHZ = TZ + 4, SHY = TY + 4, SHX = TX + 4, SHV = SHZ * SHY * SHX;

__global__ __launch_bounds__(256) void conv3d_wgrad_stem5_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                 float* __restrict__ partial, int B, int cout, int D, int H, int W,
                                                                 int tiles_x, int tiles_y, int tiles_z, int S) {
  __shared__ float lds_g[32 * GS];
  __shared__ float lds_x[SHV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int s = bid % S;
  const int cb = bid / S;
  const int NT = B * tiles_z * tiles_y * tiles_x;
  const size_t DHW = (size_t)D * H * W;
  const int cl = lane & 31, kh = lane >> 5;
  const int tap = wave * 32 + cl;                       // this lane's tap (column j of the MFMA)
  const bool tap_ok = tap < 125;
  const int tq = tap_ok ? tap : 0;
  const int toff = ((tq / 25) * SHY + (tq / 5) % 5) * SHX + tq % 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int t = s; t < NT; t += S) {
    int q = t;
    const int tx = q % tiles_x; q /= tiles_x;
    const int ty = q % tiles_y; q /= tiles_y;
    const int tz = q % tiles_z;
    const int b = q / tiles_z;
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    __syncthreads();
    for (int e = tid; e < 32 * TV; e += 256) {
      const int c = e / TV, v = e % TV;
      const int xx = x0 + (v & 15), yy = y0 + ((v >> 4) & 3), zz = z0 + (v >> 6);
      const int co = cb * 32 + c;
      float val = 0.f;
      if (co < cout && xx < W && yy < H && zz < D) val = gy[((size_t)b * cout + co) * DHW + ((size_t)zz * H + yy) * W + xx];
      lds_g[c * GS + v] = val;
    }
    for (int r = tid; r < SHV; r += 256) {
      const int hx = r % SHX, hy = (r / SHX) % SHY, hz = r / (SHX * SHY);
      const int xx = x0 + hx - SP, yy = y0 + hy - SP, zz = z0 + hz - SP;
      float val = 0.f;
      if ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && (unsigned)zz < (unsigned)D)
        val = x[(size_t)b * DHW + ((size_t)zz * H + yy) * W + xx];
      lds_x[r] = val;
    }
    __syncthreads();
    const float* ga = lds_g + cl * GS + kh;
#pragma unroll 4
    for (int vp = 0; vp < TV / 2; ++vp) {
      const int v = vp * 2 + kh;
      const float a = ga[vp * 2];
      const int hb = ((v >> 6) * SHY + ((v >> 4) & 3)) * SHX + (v & 15);
      const float bv = tap_ok ? lds_x[hb + toff] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
  }
  float* pp = partial + (size_t)s * cout * 125;
  if (tap_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (co < cout) pp[(size_t)co * 125 + tap] = acc[r];
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int slots, long long n,
                                                           float* __restrict__ dw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float sum = 0.f;
  for (int s = 0; s < slots; ++s) sum += partial[(size_t)s * n + i];      // fixed order: deterministic
  dw[i] = sum;
}

// db[co] = sum_{b,v} gy[b,co,v]: one workgroup per channel, fixed-order tree
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ gy, int B, int cout, long long DHW,
                                                        float* __restrict__ db) {
  __shared__ float sm[256];
  const int co = blockIdx.x;
  float sum = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* p = gy + ((size_t)b * cout + co) * DHW;
    for (long long i = threadIdx.x; i < DHW; i += 256) sum += p[i];
  }
  sm[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[co] = sm[0];
}

struct Plan { int tiles_x, tiles_y, tiles_z, NT, blocks_c, S, slots; long long n; };

Plan make_plan(int batch, int cin, int cout, int D, int H, int W, int k) {
  Plan p;
  p.tiles_x = (W + TX - 1) / TX; p.tiles_y = (H + TY - 1) / TY; p.tiles_z = (D + TZ - 1) / TZ;
  p.NT = batch * p.tiles_x * p.tiles_y * p.tiles_z;
  const int cbs = (cout + 31) / 32, ibs = (k == 5) ? 1 : (cin + 31) / 32;
  p.blocks_c = cbs * ibs;
  int S = (1024 + p.blocks_c - 1) / p.blocks_c;          // aim at >= 1024 workgroups (4 per CU)
  if (S > p.NT) S = p.NT;
  if (S < 1) S = 1;
  p.S = S;
  p.slots = (k == 1) ? S * 4 : S;
  p.n = (long long)cout * cin * k * k * k;
  return p;
}

}  // namespace

M3D_API size_t m3d_conv3d_wgrad_workspace_bytes(int batch, int cin, int cout, int depth, int height, int width, int k) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return 0;
  const Plan p = make_plan(batch, cin, cout, depth, height, width, k);
  return (size_t)p.slots * p.n * sizeof(float) + 256;
}

M3D_API int m3d_conv3d_wgrad(const float* d_in, const float* d_grad_out, float* d_grad_weight, int batch, int cin, int cout,
                             int depth, int height, int width, int k, void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_in || !d_grad_out || !d_grad_weight || !d_ws || batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 ||
      width <= 0)
    return M3D_EINVAL;
  if (!(k == 1 || k == 3 || (k == 5 && cin == 1))) return M3D_EUNSUPPORTED;
  {   // the k = 1 / 3 kernels address one batch item of x and gy with 32-bit buffer offsets
    const size_t dhw = (size_t)depth * height * width;
    if (k != 5 && ((size_t)cin * dhw * sizeof(float) >= 0x7FFFFFFFull || (size_t)cout * dhw * sizeof(float) >= 0x7FFFFFFFull))
      return M3D_EUNSUPPORTED;
  }
  if (ws_bytes < m3d_conv3d_wgrad_workspace_bytes(batch, cin, cout, depth, height, width, k)) return M3D_EWORKSPACE;
  const Plan p = make_plan(batch, cin, cout, depth, height, width, k);
  float* partial = (float*)m3d::align_up((size_t)d_ws, 256);
  hipStream_t st = m3d::as_stream(stream);
  const long long blocks = (long long)p.blocks_c * p.S;
  if (blocks > 0x7FFFFFFFll) return M3D_EUNSUPPORTED;
  if (k == 5) {
    hipLaunchKernelGGL(conv3d_wgrad_stem5_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_in, d_grad_out, partial, batch, cout,
                       depth, height, width, p.tiles_x, p.tiles_y, p.tiles_z, p.S);
  } else if (k == 3) {
    const size_t lds = sizeof(float) * WG<3>::LDS_FLOATS;
    auto kern = conv3d_wgrad_kernel<3>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, d_in, d_grad_out, partial, batch, cin, cout, depth, height,
                       width, p.tiles_x, p.tiles_y, p.tiles_z, p.S);
  } else {
    const size_t lds = sizeof(float) * WG<1>::LDS_FLOATS;
    hipLaunchKernelGGL(conv3d_wgrad_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, st, d_in, d_grad_out, partial, batch, cin,
                       cout, depth, height, width, p.tiles_x, p.tiles_y, p.tiles_z, p.S);
  }
  int rc = m3d::check_launch("conv3d_wgrad");
  if (rc != M3D_OK) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, partial, p.slots, p.n, d_grad_weight);
  return m3d::check_launch("conv3d_wgrad_reduce");
}

M3D_API int m3d_conv3d_bias_grad(const float* d_grad_out, float* d_grad_bias, int batch, int cout, int depth, int height, int width,
                                 void* stream) {
  if (!d_grad_out || !d_grad_bias || batch <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  hipLaunchKernelGGL(bias_grad_kernel, dim3(cout), dim3(256), 0, m3d::as_stream(stream), d_grad_out, batch, cout,
                     (long long)depth * height * width, d_grad_bias);
  return m3d::check_launch("conv3d_bias_grad");
}
