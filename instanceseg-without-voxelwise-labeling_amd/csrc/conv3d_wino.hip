// 3x3x3 forward convolution with the Winograd F(2,3) minimal-filtering transform applied along x, as an fp32 MFMA
// implicit GEMM for gfx950.  Same operation as conv3d.hip's direct kernel (stride 1, pad 1, fused per-channel
// scale/shift + ReLU: the conv + eval-BN + ReLU triples of lib/modeling/DSN.py:58-67 and RPN_conv of
// lib/modeling/rpn_heads.py:94), with 2/3 of the matrix-core work:
//
//   for a pair of outputs (x = 2t, 2t+1) and the four inputs d0..d3 = in[2t-1 .. 2t+2] of one (z+dz, y+dy) row
//     m0 = (d0 - d2) * g0          m1 = (d1 + d2) * (g0 + g1 + g2)/2
//     m2 = (d2 - d1) * (g0 - g1 + g2)/2          m3 = (d1 - d3) * g2
//     y[2t] = m0 + m1 + m2         y[2t+1] = m1 - m2 - m3                 (g = the row's three x taps)
//   4 multiplications per 2 outputs instead of 6; the other two axes stay a direct 3x3 sum.
//
// GEMM view per transform position xi in 0..3:  M_xi[co][(z,y,t)] = sum_{ci,dz,dy} U_xi[co][ci][dz][dy] * V_xi[ci][z+dz][y+dy][t]
//   i = 32 output channels (A = transformed weights, packed offline), j = 32 (row, x-pair) positions, k = 2 input channels.
// * V is never materialised: the input halo tile sits in LDS with its x positions DE-INTERLEAVED per row
//   (E[u] = in[x0+2u], O[u] = in[x0+2u-1]), so d1,d3 / d0,d2 are unit-stride reads across lanes (no bank conflicts) and
//   the four B fragments are 4 ds_read_b32 + 4 VALU add/sub: the same number of LDS reads as reading a stored V.
// * the inverse transform is 3 adds on accumulators of the SAME lane, and leaves each lane with two adjacent x
//   outputs: 8-byte stores, 256-byte runs per row.
// * fp32 throughout; coefficients are +-1 and 1/2, so the error stays at the few-ulp level (tests: < 1e-5 relative
//   against fp64, north_star tolerance 1e-4).  Not used for the PRM norm / backward convolutions, whose masks test
//   exact zeros.
#include <stdlib.h>

#include "m3d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WT = 36;   // weight slots per (cout, cin): 9 (dz,dy) x 4 xi

// Wp[cin_pair][cout_block32][(dz*3+dy)*4 + xi][lane64] = U_xi of W[co = cb*32 + (lane&31)][ci = 2*pair + (lane>>5)][dz][dy][0..2]
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, int cin, int cout, float* __restrict__ wp, int ncb,
                                                        int npair) {
  const long long total = (long long)npair * ncb * WT * 64;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63);
    long long t = e >> 6;
    const int slot = (int)(t % WT); t /= WT;
    const int cb = (int)(t % ncb); t /= ncb;
    const int cpair = (int)t;
    const int co = cb * 32 + (lane & 31), ci = 2 * cpair + (lane >> 5);
    float v = 0.f;
    if (co < cout && ci < cin) {
      const int tap9 = slot >> 2, xi = slot & 3;
      const float* g = w + ((size_t)co * cin + ci) * 27 + tap9 * 3;
      const float g0 = g[0], g1 = g[1], g2 = g[2];
      v = xi == 0 ? g0 : xi == 1 ? 0.5f * (g0 + g1 + g2) : xi == 2 ? 0.5f * (g0 - g1 + g2) : g2;
    }
    wp[e] = v;
  }
}

struct WEpi {
  const float* scale;
  const float* shift;
  int relu;
  int xcd_map;
};

__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
  const int per = n >> 3, rem = n & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// CC input channels per chunk; XT x-pairs per row block (32 / 16); NCB cout blocks; NW = WZ*WY waves per K group,
// each wave owns ROWS row blocks (YB = 32/XT rows each); KS K-split groups.
template <int CC, int XT, int ROWS, int NCB, int WZ, int WY, int KS, bool POOL = false>
struct WCfg {
  static constexpr int NW = WZ * WY;
  static constexpr int NT = 64 * NW * KS;
  static constexpr int PP = CC / 2 / KS;
  static constexpr int YB = 32 / XT;
  static constexpr int TX = 2 * XT, TY = WY * ROWS * YB, TZ = WZ;
  static constexpr int EP = XT + 2;                           // entries per (even | odd) plane of one row: E[0..XT+1], O[0..XT+1]
  static constexpr int HXP = 2 * EP, HY = TY + 2, HZ = TZ + 2;
  static constexpr int QR = EP / 2;                           // 16-byte quads per row: x0-1+4q .. x0+2+4q -> O[2q],E[2q],O[2q+1],E[2q+1]
  static constexpr int CS = HXP * HY * HZ;
  static constexpr int IN_ELEMS = CC * CS;
  static constexpr int NQUAD = CC * HZ * HY * QR;
  static constexpr int W_SEG = NCB * WT * 64;
  static constexpr int W_ELEMS = (CC / 2) * W_SEG;
  static constexpr int NI = (NQUAD + NT - 1) / NT;           // input staging quads per thread
  static constexpr int NW4 = (W_ELEMS / 4 + NT - 1) / NT;
  static constexpr int LDS_FLOATS = IN_ELEMS + W_ELEMS;
  static_assert(CC % (2 * KS) == 0, "whole channel pairs per K group");
  static_assert(KS == 1 || NW * 2 * NCB * ROWS * 16 * 64 <= 2 * LDS_FLOATS, "split-K reduction buffer fits the staging area");
  static_assert(!POOL || (KS == 1 && ROWS == 1 && XT == 32 && WZ % 2 == 0 && WY % 2 == 0), "fused pool: one row per wave, 2x2 wave groups");
  static_assert(!POOL || NW * NCB * 16 * 64 <= 2 * LDS_FLOATS, "pool exchange buffer fits the staging area");
};

template <int CC, int XT, int ROWS, int NCB, int WZ, int WY, int KS, bool POOL>
__global__ __launch_bounds__(64 * WZ * WY * KS, 2) void conv3d_wino_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                                           float* __restrict__ out, int cin, int cout, int D, int H,
                                                                           int W, int tiles_x, int tiles_y, int tiles_z,
                                                                           int ncb_total, WEpi ep) {
  using C = WCfg<CC, XT, ROWS, NCB, WZ, WY, KS, POOL>;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) % C::NW, ks = (tid >> 6) / C::NW;
  const int wz = wave / WY, wy = wave % WY;

  int bid = blockIdx.x;
  const int co_tiles = ((cout + 31) / 32 + NCB - 1) / NCB;
  int cot;
  if (ep.xcd_map) {
    bid = xcd_contiguous(bid, gridDim.x);
    const int sp = tiles_x * tiles_y * tiles_z;
    cot = bid / sp; bid -= cot * sp;
  } else {
    cot = bid % co_tiles; bid /= co_tiles;
  }
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int tz = bid;
  const int b = blockIdx.y;
  const int x0 = tx * C::TX, y0 = ty * C::TY, z0 = tz * C::TZ;
  const size_t DHW = (size_t)D * H * W;
  const float* in_b = in + (size_t)b * cin * DHW;

  // Input staging works on 16-byte quads of four consecutive x (one buffer_load_dwordx4 + two ds_write_b64 instead of
  // four dword loads / writes): quad e = (ci, hz, hy, q) covers x = x0-1+4q .. x0+2+4q = O[2q], E[2q], O[2q+1], E[2q+1]
  // of its row.  Buffer loads return 0 for every dword at or beyond num_records (channels past cin in the last chunk, the
  // tail of the last row); zero padding inside the tensor (x / y / z outside the volume) is a 4-bit mask applied at commit time.
  int gq[C::NI], mq[C::NI], lq[C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int e = tid + i * C::NT;
    gq[i] = 0; mq[i] = -1; lq[i] = 0;
    if (e < C::NQUAD) {
      const int q = e % C::QR;
      const int row = e / C::QR;                      // (ci * HZ + hz) * HY + hy
      const int hy = row % C::HY, hz = (row / C::HY) % C::HZ, ci = row / (C::HY * C::HZ);
      const int z = z0 + hz - 1, y = y0 + hy - 1, xf = x0 - 1 + 4 * q;
      const bool rok = (z >= 0) & (z < D) & (y >= 0) & (y < H);
      int m = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) m |= (rok && xf + j >= 0 && xf + j < W) ? (1 << j) : 0;
      long long lin = (long long)ci * (long long)DHW + ((long long)z * H + y) * W + xf;
      if (rok && lin < 0) { lin = 0; m |= 16; }      // a negative buffer offset drops the whole quad (measured): load x = 0..3
      mq[i] = m;                                     // and shift by one element at commit time (bit 4)
      gq[i] = rok ? (int)(lin * 4) : 0;              // byte offset inside the chunk (host guarantees it fits 31 bits)
      lq[i] = row * C::HXP + 2 * q;
    }
  }

  f32x4 rin[C::NI];
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(in_b), 0, (unsigned)((size_t)cin * DHW * sizeof(float)), 0x00020000);
  f32x4 rw[C::NW4];
  const int nchunk = (cin + CC - 1) / CC;
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  const size_t w_pair_stride4 = (size_t)ncb_total * WT * 64 / 4;
  const size_t w_tile_off4 = (size_t)cot * NCB * WT * 64 / 4;
  constexpr int NL = C::NI + C::NW4;
  auto issue = [&](int idx, int chunk) __attribute__((always_inline)) {
    if (idx < C::NI) {
      const int voff = gq[idx] + chunk * (int)(CC * DHW * sizeof(float));
      rin[idx] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, voff, 0, 0));
    } else {
      const int i = idx - C::NI;
      int e = tid + i * C::NT;
      if (e >= C::W_ELEMS / 4) e = C::W_ELEMS / 4 - 1;
      const int pr = e / (C::W_SEG / 4), o = e % (C::W_SEG / 4);
      rw[i] = (wp4 + (size_t)chunk * (CC / 2) * w_pair_stride4 + w_tile_off4)[(size_t)pr * w_pair_stride4 + o];
    }
  };
  auto commit1 = [&](int idx, int chunk, float* dst_in, float* dst_w) __attribute__((always_inline)) {
    if (idx < C::NI) {
      const int m = mq[idx];
      if (m >= 0) {
        f32x4 v = rin[idx];
        if (m & 16) v = f32x4{0.f, v[0], v[1], v[2]};
        const f32x2 ev = {(m & 2) ? v[1] : 0.f, (m & 8) ? v[3] : 0.f};
        const f32x2 ov = {(m & 1) ? v[0] : 0.f, (m & 4) ? v[2] : 0.f};
        *reinterpret_cast<f32x2*>(dst_in + lq[idx]) = ev;
        *reinterpret_cast<f32x2*>(dst_in + lq[idx] + C::EP) = ov;
      }
    } else {
      const int i = idx - C::NI;
      const int e = tid + i * C::NT;
      if (e < C::W_ELEMS / 4) reinterpret_cast<f32x4*>(dst_w)[e] = rw[i];
    }
  };

  f32x16 acc[4][NCB][ROWS];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[x][c][r][g] = 0.f;

  const int jt = (lane & 31) % XT, jy = (lane & 31) / XT;
  // B-fragment base: channel half (lane>>5), the wave's z plane, its first row, E[jt]
  const int b_base = (lane >> 5) * C::CS + wz * (C::HY * C::HXP) + (wy * ROWS * C::YB + jy) * C::HXP + jt;

#pragma unroll
  for (int i = 0; i < NL; ++i) issue(i, 0);
#pragma unroll
  for (int i = 0; i < NL; ++i) commit1(i, 0, lds, lds + C::IN_ELEMS);
  __syncthreads();
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const float* cur_in = lds + (chunk & 1) * C::LDS_FLOATS;
    const float* cur_w = cur_in + C::IN_ELEMS;
    float* nxt_in = lds + ((chunk + 1) & 1) * C::LDS_FLOATS;
    float* nxt_w = nxt_in + C::IN_ELEMS;
    const int nchk = min(chunk + 1, nchunk - 1);
    const float* in_k = cur_in + b_base + ks * (C::PP * 2 * C::CS);
    const float* w_k = cur_w + ks * (C::PP * C::W_SEG) + lane;
    constexpr int NS = 9 * C::PP;
    // B fragments: raw LDS reads run TWO steps ahead, the +-transform one step ahead, so the VALU ops never wait on
    // the LDS latency in front of a step's MFMAs (a transform fed by a read issued in the same step stalls the wave
    // once per step).  A fragments (plain LDS reads) run one step ahead.
    auto read_raw = [&](int s, float (&rw4)[ROWS][4]) __attribute__((always_inline)) {
      const int tap9 = s / C::PP, pp = s % C::PP;
      const int dz = tap9 / 3, dy = tap9 % 3;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const float* p = in_k + pp * 2 * C::CS + dz * (C::HY * C::HXP) + (dy + r * C::YB) * C::HXP;
        rw4[r][0] = p[0]; rw4[r][1] = p[1]; rw4[r][2] = p[C::EP]; rw4[r][3] = p[C::EP + 1];   // E[t], E[t+1], O[t], O[t+1]
      }
    };
    auto transform = [&](const float (&rw4)[ROWS][4], float (&bf)[ROWS][4]) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const float e0 = rw4[r][0], e1 = rw4[r][1], o0 = rw4[r][2], o1 = rw4[r][3];
        bf[r][0] = o0 - o1; bf[r][1] = e0 + o1; bf[r][2] = o1 - e0; bf[r][3] = e0 - e1;
      }
    };
    auto load_a = [&](int s, float (&af)[NCB][4]) __attribute__((always_inline)) {
      const int tap9 = s / C::PP, pp = s % C::PP;
#pragma unroll
      for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int x = 0; x < 4; ++x) af[c][x] = w_k[pp * C::W_SEG + (c * WT + tap9 * 4 + x) * 64];
    };
    constexpr int THIRD = NS / 3 > 0 ? NS / 3 : 1;
    constexpr int LPS = (NL + THIRD - 1) / THIRD;
    constexpr int CSTART = NS - (NL + LPS - 1) / LPS;
    float rawq[2][ROWS][4], bfq[2][ROWS][4], afq[2][NCB][4];
    read_raw(0, rawq[0]);
    if (NS > 1) read_raw(1, rawq[1]);
    load_a(0, afq[0]);
    transform(rawq[0], bfq[0]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) transform(rawq[(s + 1) & 1], bfq[(s + 1) & 1]);     // raw(s+1) was requested a full step ago
      if (s + 2 < NS) read_raw(s + 2, rawq[s & 1]);
      if (s + 1 < NS) load_a(s + 1, afq[(s + 1) & 1]);
#pragma unroll
      for (int q = 0; q < LPS; ++q)
        if (s * LPS + q < NL) issue(s * LPS + q, nchk);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
          for (int r = 0; r < ROWS; ++r)
            acc[x][c][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[s & 1][c][x], bfq[s & 1][r][x], acc[x][c][r], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < LPS; ++q)
        if (s >= CSTART && (s - CSTART) * LPS + q < NL) commit1((s - CSTART) * LPS + q, nchk, nxt_in, nxt_w);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- inverse transform (linear, so it commutes with the K-split reduction and halves what that has to move)
  f32x16 y0v[NCB][ROWS], y1v[NCB][ROWS];
#pragma unroll
  for (int c = 0; c < NCB; ++c)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      y0v[c][r] = acc[0][c][r] + acc[1][c][r] + acc[2][c][r];
      y1v[c][r] = acc[1][c][r] - acc[2][c][r] - acc[3][c][r];
    }
  if constexpr (KS == 2) {
    float* red = lds + ((size_t)wave * 2 * NCB * ROWS * 16) * 64 + lane;
    if (ks == 1) {
#pragma unroll
      for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            red[(((0 * NCB + c) * ROWS + r) * 16 + g) * 64] = y0v[c][r][g];
            red[(((1 * NCB + c) * ROWS + r) * 16 + g) * 64] = y1v[c][r][g];
          }
    }
    __syncthreads();
    if (ks == 1) return;
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          y0v[c][r][g] += red[(((0 * NCB + c) * ROWS + r) * 16 + g) * 64];
          y1v[c][r][g] += red[(((1 * NCB + c) * ROWS + r) * 16 + g) * 64];
        }
  }

  if constexpr (POOL) {
    // conv + scale/shift + ReLU + MaxPool3d(2,2) (DSN.py:60-61): x pairs are in-lane; the 2x2 (z,y) footprint lives in
    // four waves, exchanged through the (now free) staging area.  Writes only [B,cout,D/2,H/2,W/2].
    float* red = lds + ((size_t)wave * NCB * 16) * 64 + lane;
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const int co0 = (cot * NCB + c) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int co = min(co0 + (g & 3) + 8 * (g >> 2), cout - 1);
        const float sc = ep.scale ? ep.scale[co] : 1.f, sh = ep.shift ? ep.shift[co] : 0.f;
        float v0 = y0v[c][0][g] * sc + sh, v1 = y1v[c][0][g] * sc + sh;
        if (ep.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        red[(c * 16 + g) * 64] = fmaxf(v0, v1);
      }
    }
    __syncthreads();
    const int PD = D / 2, PH = H / 2, PW = W / 2;
    constexpr int NOUT = (C::NW / 4) * NCB * 32 * 32;
    for (int o = tid; o < NOUT; o += C::NT) {
      const int t = o & 31;
      const int col = (o >> 5) % (NCB * 32);
      const int grp = (o >> 5) / (NCB * 32);
      const int gz = grp / (WY / 2), gy = grp % (WY / 2);
      const int c = col >> 5, i = col & 31;
      const int h = (i >> 2) & 1, g = (i & 3) | ((i >> 3) << 2);
      const int co = (cot * NCB + c) * 32 + i;
      const int zp = (z0 >> 1) + gz, yp = (y0 >> 1) + gy, xp = (x0 >> 1) + t;
      if (co >= cout || zp >= PD || yp >= PH || xp >= PW) continue;
      float m = -INFINITY;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int w = (2 * gz + (q >> 1)) * WY + 2 * gy + (q & 1);
        m = fmaxf(m, lds[((size_t)(w * NCB + c) * 16 + g) * 64 + h * 32 + t]);
      }
      out[((size_t)b * cout + co) * ((size_t)PD * PH * PW) + ((size_t)zp * PH + yp) * PW + xp] = m;
    }
    return;
  }

  // ---- scale/shift + ReLU + paired stores: every lane owns two adjacent x of 16 channels
  const int z = z0 + wz;
  const int x = x0 + 2 * jt;
  const bool pair_ok = ((W & 1) == 0);
#pragma unroll
  for (int c = 0; c < NCB; ++c) {
    const int co0 = (cot * NCB + c) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int y = y0 + (wy * ROWS + r) * C::YB + jy;
      if (!(z < D && y < H && x < W)) continue;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int co = co0 + (g & 3) + 8 * (g >> 2);
        if (co >= cout) continue;
        float v0 = y0v[c][r][g], v1 = y1v[c][r][g];
        const float sc = ep.scale ? ep.scale[co] : 1.f, sh = ep.shift ? ep.shift[co] : 0.f;
        v0 = v0 * sc + sh; v1 = v1 * sc + sh;
        if (ep.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        float* o = out + ((size_t)b * cout + co) * DHW + ((size_t)z * H + y) * W + x;
        if (pair_ok) {
          *reinterpret_cast<f32x2*>(o) = f32x2{v0, v1};       // x even, W even -> 8-byte aligned; x+1 < W
        } else {
          o[0] = v0;
          if (x + 1 < W) o[1] = v1;
        }
      }
    }
  }
}

inline int xcd_map_enabled() {
  return m3d::opt(m3d::OPT_XCD_MAP) != 0;
}

template <int CC, int XT, int ROWS, int NCB, int WZ, int WY, int KS, bool POOL = false>
int launch_wino(const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W, WEpi ep, hipStream_t st) {
  using C = WCfg<CC, XT, ROWS, NCB, WZ, WY, KS, POOL>;
  const int tiles_x = (W + C::TX - 1) / C::TX, tiles_y = (H + C::TY - 1) / C::TY, tiles_z = (D + C::TZ - 1) / C::TZ;
  const int ncb_total = ((cout + 31) / 32 + 1) / 2 * 2;
  const int co_tiles = ((cout + 31) / 32 + NCB - 1) / NCB;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * co_tiles;
  if (blocks > 0x7FFFFFFFll || B > 65535) return M3D_EUNSUPPORTED;
  ep.xcd_map = xcd_map_enabled();
  const size_t lds = sizeof(float) * 2 * C::LDS_FLOATS;
  if (lds > 160 * 1024) return M3D_EUNSUPPORTED;
  auto kern = conv3d_wino_kernel<CC, XT, ROWS, NCB, WZ, WY, KS, POOL>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, B), dim3(C::NT), lds, st, in, wp, out, cin, cout, D, H, W, tiles_x, tiles_y, tiles_z,
                     ncb_total, ep);
  return m3d::check_launch("conv3d_wino");
}

}  // namespace

M3D_API size_t m3d_conv3d_wino_packed_weight_bytes(int cin, int cout) {
  if (cin <= 0 || cout <= 0) return 0;
  const size_t npair = ((cin + 1) / 2 + 15) / 16 * 16, ncb = ((cout + 31) / 32 + 1) / 2 * 2;
  return sizeof(float) * npair * ncb * (size_t)WT * 64;
}

M3D_API int m3d_conv3d_wino_pack_weights(const float* d_weight, int cin, int cout, float* d_packed, void* stream) {
  if (!d_weight || !d_packed || cin <= 0 || cout <= 0) return M3D_EINVAL;
  const int npair = ((cin + 1) / 2 + 15) / 16 * 16, ncb = ((cout + 31) / 32 + 1) / 2 * 2;
  hipLaunchKernelGGL(wino_pack_kernel, dim3(1024), dim3(256), 0, m3d::as_stream(stream), d_weight, cin, cout, d_packed, ncb, npair);
  return m3d::check_launch("wino_pack");
}

M3D_API int m3d_conv3d_wino_forward(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout, int depth,
                                    int height, int width, const float* d_scale, const float* d_shift, int relu, void* stream) {
  if (!d_in || !d_packed || !d_out || batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0)
    return M3D_EINVAL;
  const size_t DHW = (size_t)depth * height * width;
  if ((size_t)cin * DHW * sizeof(float) >= 0x7FFFFFFFull) return M3D_EUNSUPPORTED;   // 32-bit buffer offsets per batch item
  WEpi ep{d_scale, d_shift, relu, 0};
  hipStream_t st = m3d::as_stream(stream);
  const int variant = m3d::opt(m3d::OPT_TUNE_WINO);
#define M3D_W(i, ...) if (variant == i) return launch_wino<__VA_ARGS__>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
  M3D_W(0, 4, 32, 1, 2, 2, 4, 1)      // 64 x (4 y) x (2 z) voxels x 64 channels, 8 waves
  M3D_W(1, 4, 32, 1, 2, 4, 2, 1)
  M3D_W(4, 4, 16, 1, 2, 2, 2, 2)      // 32 x 4 x 2 voxels, split K over two groups of 4 waves
  M3D_W(5, 8, 16, 1, 1, 4, 2, 1)      // 32 x 4 x 4 voxels x 32 channels, 8 waves
  M3D_W(6, 4, 16, 1, 2, 2, 4, 1)
#undef M3D_W
  if (variant >= 0) return M3D_EUNSUPPORTED;
  if (width >= 48) return launch_wino<4, 32, 1, 2, 4, 2, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
  // 24..47: 32 x 4 x 4 voxels x 32 channels per workgroup, 8 waves, 8 channels per barrier (4 % faster than the split-K tile)
  if (width >= 24 && cout >= 64) return launch_wino<8, 16, 1, 1, 4, 2, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
  if (width >= 24) return launch_wino<4, 16, 1, 2, 2, 2, 2>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
  return M3D_EUNSUPPORTED;
}

/* conv + scale/shift + ReLU + MaxPool3d(2,2) in one launch (the Winograd counterpart of m3d_conv3d_forward_pool2, without
 * the argmax output): writes [batch,cout,D/2,H/2,W/2].  Needs width >= 48. */
M3D_API int m3d_conv3d_wino_forward_pool2(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                          int depth, int height, int width, const float* d_scale, const float* d_shift, int relu,
                                          void* stream) {
  if (!d_in || !d_packed || !d_out || batch <= 0 || cin <= 0 || cout <= 0 || depth < 2 || height < 2 || width < 2) return M3D_EINVAL;
  const size_t DHW = (size_t)depth * height * width;
  if ((size_t)cin * DHW * sizeof(float) >= 0x7FFFFFFFull || width < 48) return M3D_EUNSUPPORTED;
  WEpi ep{d_scale, d_shift, relu, 0};
  return launch_wino<4, 32, 1, 2, 2, 4, 1, true>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, m3d::as_stream(stream));
}
