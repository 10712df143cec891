// 3x3x3 forward convolution with Winograd F(2x2, 3x3) on the (y, x) plane, fp32 MFMA implicit GEMM for gfx950.
// Same operation as conv3d.hip's direct kernel and conv3d_wino.hip's F(2,3)-along-x kernel (stride 1, pad 1, fused
// per-channel scale/shift + ReLU, optional fused MaxPool3d(2,2): the conv + eval-BN + ReLU [+ pool] groups of
// lib/modeling/DSN.py:58-67), with 16/36 = 4/9 of the matrix-core work:
//
//   per 2x2 output patch and per (ci, dz):  M[eta][xi] = (By^T d Bx)[eta][xi] * (G g G^T)[eta][xi],   Y = A^T M A
//   with the 1-D F(2,3) matrices  B^T d = (d0-d2, d1+d2, d2-d1, d1-d3),  G g = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2),
//   A^T m = (m0+m1+m2, m1-m2-m3)  applied along y (eta) and x (xi); the z taps stay a direct 3-term sum.
//
// GEMM view per (eta, xi): i = 32 output channels (A = transformed weights, packed offline: 48 slots per cout x cin),
// j = 32 patches (XT x-pairs x YT y-pairs), k = 2 input channels.  16 accumulator blocks per wave (256 AGPRs), so the
// kernel runs ONE wave per SIMD (4-wave workgroups, one per CU) and relies on the long MFMA runs between fragment loads
// (16 MFMAs = 1024 cycles per K step) instead of a second wave to hide LDS latency.
// * the input halo tile sits in LDS with x de-interleaved per row (E[u] = in[x0+2u], O[u] = in[x0+2u-1]) exactly as in
//   conv3d_wino.hip; a lane reads its 4 rows x 4 x-values (16 ds_read_b32, conflict-free: the row pitch is padded so
//   that two y-pairs land 16 banks apart), combines rows (y transform) and columns (x transform) with 32 VALU ops;
// * the inverse transform is 12 adds per channel on accumulators of the same lane and leaves each lane with a 2x2
//   output patch: 8-byte stores, and the fused pool needs no cross-lane step in (y, x), only one LDS exchange in z;
// * coefficients are +-1, 1/2, 1/4: errors stay at the fp32 few-ulp level (tests: < 1e-5 relative against fp64).
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "m3d_common.h"
#include "conv3d_wino2q.h"
#include "conv3d_wino24.h"
#include "conv3d_wino24w.h"

// timing-only ablation builds (tools/ablate_wino2.sh): 1 = no staging, 2 = no chunk barrier, 4 = no B transform VALU (one-wave kernel),
// 8 = 4 instead of 16 transform VALU per K step in the eta-split kernel, 16 = no raw-row LDS reads, 32 = no weight-fragment LDS reads, 64 = no input loads, 128 = input loads without LDS commit,
// 256 = no weight DMA, 512 = every workgroup stages the same (cache-resident) input tile (eta-split kernel)
#ifndef M3D_EXP
#define M3D_EXP 0
#endif

// diagnostic build only (make w2_stamps; tools/w2_stamps.py): s_memtime stamps of one wave per workgroup around the prologue, the K loop,
// the eta exchange and the epilogue, written to a buffer of their own (m3d_debug_set_stamp_buffer); no output value depends on them
#ifdef M3D_W2_STAMPS
static unsigned long long* g_w2_stamps = nullptr;
M3D_API void m3d_debug_set_stamp_buffer(void* p) { g_w2_stamps = (unsigned long long*)p; }
#define W2_STAMP(k) do { if (ep.stamps && tid == 0) ep.stamps[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 8 + (k)] = \
    (k) == 5 || (k) == 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W2_STAMP(k) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_cfloat;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;   // LDS-space element type (32-bit addresses, ds_* instructions)
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

// Pins values at this program point: the empty asm is a use + redefinition, so arithmetic on them cannot be placed earlier (the
// pre-RA scheduler otherwise hoists the y transform to right behind the LDS reads of the previous step and the wave
// waits out the full LDS latency four times per K step: tools/asm_trace.py shows "r r r [lgkmcnt(1)] v v").
template <int N>
__device__ __forceinline__ void pin_regs(float (&r)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i]));
}
template <int A, int B>
__device__ __forceinline__ void pin_regs(float (&r)[A][B]) {
#pragma unroll
  for (int a = 0; a < A; ++a) pin_regs(r[a]);
}

template <int A, int B>
__device__ __forceinline__ void pin_regs(f32x2 (&r)[A][B]) {      // 64-bit words stay register pairs (pinning their halves costs copies)
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b < B; ++b) asm volatile("" : "+v"(r[a][b]));
}

constexpr int WT2 = 48;   // weight slots per (cout, cin): 3 dz x 4 eta x 4 xi

// Wp[cin_pair][cout_block32][dz*4 + eta][lane64][xi] = (G g_dz G^T)[eta][xi], co = cb*32 + (lane&31), ci = 2*pair + (lane>>5):
// a lane's four xi fragments of one (dz, eta) are 16 contiguous bytes (one ds_read_b128 in the kernels; round 2 kept [slot][lane],
// four-byte reads), a (pair, cout block) is 12 KB contiguous (LDS-DMA pieces of 1 KB).
using m3d_w2q::w2_slot;
__global__ __launch_bounds__(256) void wino2_pack_kernel(const float* __restrict__ w, int cin, int cout, float* __restrict__ wp,
                                                         int ncb, int npair) {
  const long long total = (long long)npair * ncb * WT2 * 64;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int xi = (int)(e & 3), lane = (int)((e >> 2) & 63);
    long long t = e >> 8;
    const int grp = (int)(t % 12); t /= 12;
    const int cb = (int)(t % ncb); t /= ncb;
    const int cpair = (int)t;
    const int co = cb * 32 + (lane & 31), ci = 2 * cpair + (lane >> 5);
    float v = 0.f;
    if (co < cout && ci < cin) {
      const int dz = grp >> 2, eta = grp & 3;
      const float* g = w + ((size_t)co * cin + ci) * 27 + dz * 9;      // g[dy*3 + dx]
      float col[3];                                                      // (G g)[eta][dx]
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float a = g[dx], b = g[3 + dx], c = g[6 + dx];
        col[dx] = eta == 0 ? a : eta == 1 ? 0.5f * (a + b + c) : eta == 2 ? 0.5f * (a - b + c) : c;
      }
      v = xi == 0 ? col[0] : xi == 1 ? 0.5f * (col[0] + col[1] + col[2]) : xi == 2 ? 0.5f * (col[0] - col[1] + col[2]) : col[2];
    }
    wp[e] = v;
  }
}

struct W2Epi {
  const float* scale;
  const float* shift;
  int relu;
  int xcd_map;
  // split-K over workgroups (small maps: too few output blocks to fill the chip): blockIdx.z = slice; slice s handles
  // channel chunks [s*cps, (s+1)*cps) and writes its un-scaled partial result to out + s*slice_stride
  int ksplit, cps;
  size_t slice_stride;
  unsigned char* argmax;   // fused-pool kernels instantiated with AM: index 0..7 = (dz, dy, dx) of each pooled value's first maximum
#ifdef M3D_W2_STAMPS
  unsigned long long* stamps;
#endif
};

__device__ __forceinline__ int xcd_contiguous2(int bid, int n) {
  const int per = n >> 3, rem = n & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// CC input channels per chunk; XT x-pairs and YT = 32/XT y-pairs per wave block; WZ x WY waves (4 per workgroup).
template <int CC, int XT, int WZ, int WY, bool POOL>
struct W2Cfg {
  static constexpr int NT = 256;
  static constexpr int PP = CC / 2;
  static constexpr int YT = 32 / XT;
  static constexpr int TX = 2 * XT, TY = 2 * YT * WY, TZ = WZ;
  static constexpr int EP = XT + 2;
  static constexpr int QR = EP / 2;                            // 16-byte quads per row
  // row pitch: the YT y-pairs of a block sit 2*HXP floats apart; 2*HXP = XT (mod 32) puts them on disjoint banks
  static constexpr int HXP = (YT == 1) ? 2 * EP : ((2 * EP - XT / 2 + 15) / 16 * 16 + XT / 2);
  static constexpr int HY = TY + 2, HZ = TZ + 2;
  static constexpr int CS = HXP * HY * HZ;
  static constexpr int IN_ELEMS = CC * CS;
  static constexpr int NQUAD = CC * HZ * HY * QR;
  static constexpr int W_SEG = WT2 * 64;                       // one cout block
  static constexpr int W_ELEMS = PP * W_SEG;
  static constexpr int NI = (NQUAD + NT - 1) / NT;
  static constexpr int NW4 = (W_ELEMS / 4 + NT - 1) / NT;
  static constexpr int DUMP = IN_ELEMS + W_ELEMS;              // 2 x 8-byte dump slots behind each buffer (branch-free staging)
  static constexpr int LDS_FLOATS = IN_ELEMS + W_ELEMS + ((EP + 2 + 3) / 4) * 4;
  static constexpr int RED_FLOATS = POOL ? 4 * 16 * 64 : 0;
  static constexpr int SMEM_FLOATS = 2 * LDS_FLOATS > RED_FLOATS ? 2 * LDS_FLOATS : RED_FLOATS;
  static_assert(WZ * WY == 4, "4 waves per workgroup, one per SIMD");
  static_assert(!POOL || WZ == 2, "fused pool: the z pair lives in waves wz = 0, 1");
  static_assert(HXP % 2 == 0, "8-byte LDS stores");
  static_assert((W_ELEMS / 4) % NT == 0, "weight staging is branch-free: whole float4 rounds");
};

template <int CC, int XT, int WZ, int WY, bool POOL>
__global__ __launch_bounds__(256, 1) void conv3d_wino2_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                              float* __restrict__ out, int cin, int cout, int D, int H, int W,
                                                              int tiles_x, int tiles_y, int tiles_z, int ncb_total, W2Epi ep) {
  using C = W2Cfg<CC, XT, WZ, WY, POOL>;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wz = wave / WY, wy = wave % WY;

  int bid = blockIdx.x;
  const int co_tiles = (cout + 31) / 32;
  int cot;
  // XCD-contiguous order with the cout tile FASTEST: the co_tiles workgroups that share one input tile run next to
  // each other on the same XCD, so the slab is fetched into one L2 once (these layers are input-dominated: measured
  // 225 MB -> see profiles/r01_pmc_traffic.json per conv2b launch when the cout tiles sat on different XCDs)
  if (ep.xcd_map) bid = xcd_contiguous2(bid, gridDim.x);
  cot = bid % co_tiles; bid /= co_tiles;
  const int tx = bid % tiles_x; bid /= tiles_x;
  // z tiles in groups of 4 inside the y sweep: the tiles an XCD works on at the same time form a compact (y, z) block whose
  // halo planes stay in its 4 MB L2 (a full-y, single-z slab order re-fetches every z halo plane for the next slab)
  constexpr int ZG = 4;
  int ty, tz;
  {
    const int n_full = tiles_z / ZG, full = n_full * ZG * tiles_y;
    if (bid < full) {
      const int zl = bid % ZG; bid /= ZG;
      ty = bid % tiles_y; tz = (bid / tiles_y) * ZG + zl;
    } else {
      const int zr = tiles_z - n_full * ZG, rem = bid - full;
      ty = rem / zr; tz = n_full * ZG + rem % zr;
    }
  }
  const int b = blockIdx.y;
  const int x0 = tx * C::TX, y0 = ty * C::TY, z0 = tz * C::TZ;
  const size_t DHW = (size_t)D * H * W;
  const float* in_b = in + (size_t)b * cin * DHW;

  // ---- input staging descriptors: 16-byte quads, see conv3d_wino.hip
  int gq[C::NI], mq[C::NI], lq[C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int e = tid + i * C::NT;
    gq[i] = 0; mq[i] = 0; lq[i] = C::DUMP;          // quads beyond the tile: masked to zero, written to a dump slot
    if (e < C::NQUAD) {
      const int q = e % C::QR;
      const int row = e / C::QR;
      const int hy = row % C::HY, hz = (row / C::HY) % C::HZ, ci = row / (C::HY * C::HZ);
      const int z = z0 + hz - 1, y = y0 + hy - 1, xf = x0 - 1 + 4 * q;
      const bool rok = (z >= 0) & (z < D) & (y >= 0) & (y < H);
      int m = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) m |= (rok && xf + j >= 0 && xf + j < W) ? (1 << j) : 0;
      long long lin = (long long)ci * (long long)DHW + ((long long)z * H + y) * W + xf;
      if (rok && lin < 0) { lin = 0; m |= 16; }
      mq[i] = m;
      gq[i] = rok ? (int)(lin * 4) : 0;
      lq[i] = row * C::HXP + 2 * q;
    }
  }
  // weights and input quads are staged one after the other through the SAME registers (weights: loads in step 0,
  // LDS writes in step 2; input: loads in step 2, writes in step NS-2)
  constexpr int NSTG = C::NI > C::NW4 ? C::NI : C::NW4;
  f32x4 stg[NSTG];
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(in_b), 0, (unsigned)((size_t)cin * DHW * sizeof(float)), 0x00020000);
  const int nchunk_all = (cin + CC - 1) / CC;
  const int c_begin = ep.ksplit > 1 ? (int)blockIdx.z * ep.cps : 0;
  const int nchunk = ep.ksplit > 1 ? min(nchunk_all, c_begin + ep.cps) : nchunk_all;     // one past this slice's last chunk
  if (ep.ksplit > 1) out += (size_t)blockIdx.z * ep.slice_stride;
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  const size_t w_pair_stride4 = (size_t)ncb_total * WT2 * 64 / 4;
  const size_t w_tile_off4 = (size_t)cot * WT2 * 64 / 4;
  auto issue = [&](int idx, int chunk) __attribute__((always_inline)) {
    if (idx < C::NI) {
      const int voff = gq[idx] + chunk * (int)(CC * DHW * sizeof(float));
      stg[idx] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, voff, 0, 0));
    } else {
      const int i = idx - C::NI;
      const int e = tid + i * C::NT;
      const int pr = e / (C::W_SEG / 4), o = e % (C::W_SEG / 4);
      stg[i] = (wp4 + (size_t)chunk * (CC / 2) * w_pair_stride4 + w_tile_off4)[(size_t)pr * w_pair_stride4 + o];
    }
  };
  auto commit1 = [&](int idx, float* dst_in, float* dst_w) __attribute__((always_inline)) {
    if (idx < C::NI) {                                 // branch-free: the K loop must stay one scheduling region
      const int m = mq[idx];
      const f32x4 v = stg[idx];
      const bool sh = (m & 16) != 0;
      const float v0 = sh ? 0.f : v[0], v1 = sh ? v[0] : v[1], v2 = sh ? v[1] : v[2], v3 = sh ? v[2] : v[3];
      const f32x2 ev = {(m & 2) ? v1 : 0.f, (m & 8) ? v3 : 0.f};
      const f32x2 ov = {(m & 1) ? v0 : 0.f, (m & 4) ? v2 : 0.f};
      *reinterpret_cast<f32x2*>(dst_in + lq[idx]) = ev;
      *reinterpret_cast<f32x2*>(dst_in + lq[idx] + C::EP) = ov;
    } else {
      const int i = idx - C::NI;
      reinterpret_cast<f32x4*>(dst_w)[tid + i * C::NT] = stg[i];
    }
  };

  f32x16 acc[4][4];   // [eta][xi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[a][x][g] = 0.f;

  const int jt = (lane & 31) % XT, ju = (lane & 31) / XT;
  // B base: channel half, the wave's z plane, halo row 2*(wy*YT + ju) (= output row pair's y-1), E[jt]
  const int b_base = (lane >> 5) * C::CS + wz * (C::HY * C::HXP) + 2 * (wy * C::YT + ju) * C::HXP + jt;

  constexpr int NS = 3 * C::PP;                        // K steps per chunk: dz x channel pair
  static_assert(NS % 2 == 0 && NS >= 4, "the fragment rings are indexed statically across the chunk loop");
  auto read_raw = [&](const float* in_k, int s, float (&r)[4][4]) __attribute__((always_inline)) {
    const int dz = s / C::PP, pp = s % C::PP;
    const float* p = in_k + pp * 2 * C::CS + dz * (C::HY * C::HXP);
#pragma unroll
    for (int a = 0; a < 4; ++a) {                      // 4 halo rows x (E[t], E[t+1], O[t], O[t+1])
      r[a][0] = p[a * C::HXP]; r[a][1] = p[a * C::HXP + 1]; r[a][2] = p[a * C::HXP + C::EP]; r[a][3] = p[a * C::HXP + C::EP + 1];
    }
  };
  auto transform = [&](const float (&r)[4][4], float (&bf)[4][4]) __attribute__((always_inline)) {
#if (M3D_EXP & 4)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int v = 0; v < 4; ++v) bf[a][v] = r[a][v];
    return;
#endif
    float c[4][4];                                     // rows combined (y transform), still raw in x
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      c[0][v] = r[0][v] - r[2][v]; c[1][v] = r[1][v] + r[2][v]; c[2][v] = r[2][v] - r[1][v]; c[3][v] = r[1][v] - r[3][v];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {                      // x transform: e0 = c[a][0], e1 = c[a][1], o0 = c[a][2], o1 = c[a][3]
      bf[a][0] = c[a][2] - c[a][3]; bf[a][1] = c[a][0] + c[a][3]; bf[a][2] = c[a][3] - c[a][0]; bf[a][3] = c[a][0] - c[a][1];
    }
  };
  auto load_a = [&](const float* w_k, int s, float (&af)[16]) __attribute__((always_inline)) {
    const int dz = s / C::PP, pp = s % C::PP;
#pragma unroll
    for (int q = 0; q < 16; ++q) af[q] = w_k[pp * C::W_SEG + w2_slot(dz, q >> 2, q & 3, 0)];     // w_k = segment + 4 * lane
  };

  // ---- prologue: chunk 0 -> buffer 0, first fragments
#pragma unroll
  for (int i = 0; i < C::NW4; ++i) issue(C::NI + i, c_begin);
#pragma unroll
  for (int i = 0; i < C::NW4; ++i) commit1(C::NI + i, lds, lds + C::IN_ELEMS);
#pragma unroll
  for (int i = 0; i < C::NI; ++i) issue(i, c_begin);
#pragma unroll
  for (int i = 0; i < C::NI; ++i) commit1(i, lds, lds + C::IN_ELEMS);
  __syncthreads();
  float raw[2][4][4], bfq[2][4][4], afq[2][16];
  read_raw(lds + b_base, 0, raw[0]);
  read_raw(lds + b_base, 1, raw[1]);
  load_a(lds + C::IN_ELEMS + 4 * lane, 0, afq[0]);
  transform(raw[0], bfq[0]);

  // ---- K loop, software-pipelined ACROSS chunks.  With one wave per SIMD nothing hides a refill of the fragment
  // pipeline after the chunk barrier, so the barrier sits at the end of step NS-2 (all staging writes of the next chunk
  // are done by then and every LDS read of the current chunk has been issued and waited for) and the last step's 16
  // MFMAs cover the first fragment reads of the next chunk.
  constexpr int SW = 0, SX = (NS - 2) / 2;             // weights: loads in step SW, writes in step SX; input: loads SX, writes NS-2
  for (int chunk = c_begin; chunk < nchunk; ++chunk) {
    const float* cur_in = lds + ((chunk - c_begin) & 1) * C::LDS_FLOATS;
    const float* cur_w = cur_in + C::IN_ELEMS;
    float* nxt_in = lds + ((chunk - c_begin + 1) & 1) * C::LDS_FLOATS;
    float* nxt_w = nxt_in + C::IN_ELEMS;
    const int nchk = min(chunk + 1, nchunk - 1);
    const float* in_k = cur_in + b_base;
    const float* w_k = cur_w + 4 * lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) transform(raw[(s + 1) & 1], bfq[(s + 1) & 1]);
      if (s + 2 < NS) read_raw(in_k, s + 2, raw[s & 1]);
      if (s + 1 < NS) load_a(w_k, s + 1, afq[(s + 1) & 1]);
      if (s == NS - 1) {                               // next chunk's first fragments (its buffer is complete: barrier below)
        read_raw(nxt_in + b_base, 0, raw[0]);
        read_raw(nxt_in + b_base, 1, raw[1]);
        load_a(nxt_w + 4 * lane, 0, afq[0]);
        transform(raw[0], bfq[0]);
      }
#if !(M3D_EXP & 1)
      if (s == SW) {
#pragma unroll
        for (int i = 0; i < C::NW4; ++i) issue(C::NI + i, nchk);
      }
      if (s == SX) {
#pragma unroll
        for (int i = 0; i < C::NW4; ++i) commit1(C::NI + i, nxt_in, nxt_w);
#pragma unroll
        for (int i = 0; i < C::NI; ++i) issue(i, nchk);
      }
#endif
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int x = 0; x < 4; ++x)
          acc[a][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[s & 1][a * 4 + x], bfq[s & 1][a][x], acc[a][x], 0, 0, 0);
#if !(M3D_EXP & 1)
      if (s == NS - 2) {
#pragma unroll
        for (int i = 0; i < C::NI; ++i) commit1(i, nxt_in, nxt_w);
      }
#endif
      // One wave per SIMD: nothing else fills the matrix pipe while this wave issues LDS / VALU / VMEM work, so that work
      // (all of it for LATER steps, independent of this step's MFMAs) is spread between the 16 MFMAs instead of sitting
      // in front of them, in a fixed pattern per MFMA slot (measured: spreading the LDS writes and VMEM loads as well is
      // worth 4 % over letting the scheduler place them).
#ifndef M3D_SG
#define M3D_SG 2, 8, 1, 1      /* per MFMA slot: DS reads, VALU, DS writes, VMEM reads (best of the patterns tried) */
#endif
      {
        constexpr int sg[4] = {M3D_SG};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, sg[0], 0);     // DS read
          __builtin_amdgcn_sched_group_barrier(0x002, sg[1], 0);     // VALU
          __builtin_amdgcn_sched_group_barrier(0x200, sg[2], 0);     // DS write
          __builtin_amdgcn_sched_group_barrier(0x020, sg[3], 0);     // VMEM read
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#if !(M3D_EXP & 2)
      if (s == NS - 2) __syncthreads();
#endif
    }
  }
  __syncthreads();                                     // the pool exchange below reuses the staging area

  // ---- inverse transform: over xi, then over eta -> y[row][col] for the lane's 2x2 patch
  f32x16 yv[2][2];
  {
    f32x16 p0[4], p1[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      p0[a] = acc[a][0] + acc[a][1] + acc[a][2];
      p1[a] = acc[a][1] - acc[a][2] - acc[a][3];
    }
    yv[0][0] = p0[0] + p0[1] + p0[2]; yv[1][0] = p0[1] - p0[2] - p0[3];
    yv[0][1] = p1[0] + p1[1] + p1[2]; yv[1][1] = p1[1] - p1[2] - p1[3];
  }
  const int co0 = cot * 32 + 4 * (lane >> 5);
  const int z = z0 + wz;
  const int x = x0 + 2 * jt;
  const int y = y0 + 2 * (wy * C::YT + ju);

  if constexpr (POOL) {
    // conv + scale/shift + ReLU + MaxPool3d(2,2): the (y, x) 2x2 footprint is in the lane; the z pair is wave wz = 0 / 1
    float pooled[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = min(co0 + (g & 3) + 8 * (g >> 2), cout - 1);
      const float sc = ep.scale ? ep.scale[co] : 1.f, sh = ep.shift ? ep.shift[co] : 0.f;
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float v = yv[r][c][g] * sc + sh;
          if (ep.relu) v = fmaxf(v, 0.f);
          m = fmaxf(m, v);
        }
      pooled[g] = m;
    }
    float* red = lds + (size_t)wy * 16 * 64 + lane;     // final barrier of the chunk loop has passed: staging area is free
    if (wz == 1) {
#pragma unroll
      for (int g = 0; g < 16; ++g) red[g * 64] = pooled[g];
    }
    __syncthreads();
    if (wz == 1) return;
    const int PD = D / 2, PH = H / 2, PW = W / 2;
    const int zp = z0 >> 1, yp = y >> 1, xp = x >> 1;
    if (zp >= PD || yp >= PH || xp >= PW) return;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = co0 + (g & 3) + 8 * (g >> 2);
      if (co < cout)
        out[((size_t)b * cout + co) * ((size_t)PD * PH * PW) + ((size_t)zp * PH + yp) * PW + xp] = fmaxf(pooled[g], red[g * 64]);
    }
    return;
  }

  if (!(z < D && y < H && x < W)) return;
  const bool pair_ok = ((W & 1) == 0);
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int co = co0 + (g & 3) + 8 * (g >> 2);
    if (co >= cout) continue;
    const float sc = ep.scale ? ep.scale[co] : 1.f, sh = ep.shift ? ep.shift[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (y + r >= H) continue;
      float v0 = yv[r][0][g] * sc + sh, v1 = yv[r][1][g] * sc + sh;
      if (ep.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      float* o = out + ((size_t)b * cout + co) * DHW + ((size_t)z * H + y + r) * W + x;
      if (pair_ok) {
        *reinterpret_cast<f32x2*>(o) = f32x2{v0, v1};
      } else {
        o[0] = v0;
        if (x + 1 < W) o[1] = v1;
      }
    }
  }
}

// eta-split variant (conv3d_wino2e_kernel): same tile, 8 waves - waves w and w + 4 share a SIMD and split the 16 (eta, xi)
// accumulator blocks by eta half.
template <int CC, int XT, int WZ, int WY, bool POOL>
struct W2CfgE {
  static constexpr int NT = 512;
  static constexpr int PP = CC / 2;
  static constexpr int YT = 32 / XT;
  static constexpr int TX = 2 * XT, TY = 2 * YT * WY, TZ = WZ;
  // halo rows in natural x order: element s <-> x = x0 - 1 + s, s = 0 .. 2*XT + 1; the lane of x pair t reads s = 2t .. 2t + 3 as two
  // 8-byte words.  Row pitch: the YT y-pairs of a block sit 2*HXP floats apart and must fall on disjoint banks for those reads
  // (XT * 8 bytes per y-pair out of 256): XT = 32 any pitch, XT = 16 2*HXP = 32 (mod 64), XT = 8 2*HXP = 16 (mod 32)
  static constexpr int ROW = 2 * XT + 2;
  static constexpr int HXP = XT == 32 ? 68 : (XT == 16 ? 48 : 24);     // multiples of 4: rows are whole quads
  static constexpr int HY = TY + 2, HZ = TZ + 2;
  static constexpr int CS = HXP * HY * HZ;
  static constexpr int QR = HXP / 4;                           // 16-byte quads per row
  static constexpr int NQUAD = CC * HZ * HY * QR;
  static constexpr int NI = (NQUAD + NT - 1) / NT;
  static constexpr int IN_ELEMS = CC * CS + 4;                 // + a 16-byte dump slot (branch-free staging of the ragged last round)
  static constexpr int DUMP = CC * CS;
  static constexpr int W_SEG = WT2 * 64;                       // one cout block of one channel pair
  static constexpr int W_ELEMS = PP * W_SEG;
  static constexpr int NWD = W_ELEMS / 256 / 8;                // 16-byte LDS-DMA pieces (1 KB) per wave and chunk
  static constexpr int LDS_FLOATS = IN_ELEMS + W_ELEMS;
  static constexpr int XCH_FLOATS = 4 * 64 * 64;               // eta-half exchange: 4 wave pairs x 64 floats x 64 lanes
  static constexpr int RED_FLOATS = XCH_FLOATS + (POOL ? 4 * 16 * 64 : 0);
  static constexpr int SMEM_FLOATS = 2 * LDS_FLOATS > RED_FLOATS ? 2 * LDS_FLOATS : RED_FLOATS;
  static_assert(WZ * WY == 4, "4 wave pairs per workgroup, one pair per SIMD");
  static_assert(!POOL || WZ == 2, "fused pool: the z pair lives in waves wz = 0, 1");
  static_assert(ROW <= HXP && HXP % 4 == 0, "rows are whole 16-byte quads");
  static_assert(W_ELEMS % (256 * 8) == 0, "weight staging: whole 1 KB pieces per wave");
  static_assert((3 * HXP + 4) * 4 < 65536, "row reads address one register with 16-bit offsets");
};

template <int CC, int XT, int WZ, int WY, bool POOL, bool AM = false>
__global__ __launch_bounds__(512, 2) void conv3d_wino2e_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                              float* __restrict__ out, int cin, int cout, int D, int H, int W,
                                                              int tiles_x, int tiles_y, int tiles_z, int ncb_total, W2Epi ep) {
  using C = W2CfgE<CC, XT, WZ, WY, POOL>;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  W2_STAMP(0); W2_STAMP(5);
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int eh = wave8 >> 2, wave = wave8 & 3;          // eta half (waves w and w + 4 sit on the same SIMD), wave position in the tile
  const int wz = wave / WY, wy = wave % WY;

  int bid = blockIdx.x;
  const int co_tiles = (cout + 31) / 32;
  int cot;
  // XCD-contiguous order with the cout tile FASTEST: the co_tiles workgroups that share one input tile run next to
  // each other on the same XCD, so the slab is fetched into one L2 once (these layers are input-dominated: measured
  // 225 MB -> see profiles/r01_pmc_traffic.json per conv2b launch when the cout tiles sat on different XCDs)
  if (ep.xcd_map) bid = xcd_contiguous2(bid, gridDim.x);
  cot = bid % co_tiles; bid /= co_tiles;
  const int tx = bid % tiles_x; bid /= tiles_x;
  // z tiles in groups of 4 inside the y sweep: the tiles an XCD works on at the same time form a compact (y, z) block whose
  // halo planes stay in its 4 MB L2 (a full-y, single-z slab order re-fetches every z halo plane for the next slab)
  constexpr int ZG = 4;
  int ty, tz;
  {
    const int n_full = tiles_z / ZG, full = n_full * ZG * tiles_y;
    if (bid < full) {
      const int zl = bid % ZG; bid /= ZG;
      ty = bid % tiles_y; tz = (bid / tiles_y) * ZG + zl;
    } else {
      const int zr = tiles_z - n_full * ZG, rem = bid - full;
      ty = rem / zr; tz = n_full * ZG + rem % zr;
    }
  }
  const int b = blockIdx.y;
  const int x0 = tx * C::TX, y0 = (M3D_EXP & 512) ? 0 : ty * C::TY, z0 = (M3D_EXP & 512) ? 0 : tz * C::TZ;
  const size_t DHW = (size_t)D * H * W;
  const float* in_b = in + (size_t)((M3D_EXP & 512) ? 0 : b) * cin * DHW;

  // ---- staging.  Weights: LDS-DMA, 1 KB pieces (`buffer_load_dwordx4 ... lds`: no registers, no LDS stores).  Input: 16-byte quads
  // through registers (global load in step 0, masked `ds_write_b128` in step NS-2): the x borders of the volume (x = -1, x >= W) fall
  // inside quads, so they need the per-element masks.  (Per-dword LDS-DMA with one lane per LDS element needs no masks at all - the
  // hardware range check supplies every zero - but costs 13 + 3 vector-memory instructions per wave and chunk instead of 4 + 3, and a
  // vector-memory instruction is the most expensive thing a wave can issue beside fp32 MFMAs: conv3b 0.461 -> 0.504 ms, measured.)
  int gq[C::NI], mq[C::NI], lq[C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int e = tid + i * C::NT;
    gq[i] = 0; mq[i] = 0; lq[i] = C::DUMP;          // quads beyond the tile: masked to zero, written to a dump slot
    if (e < C::NQUAD) {
      const int q = e % C::QR;
      const int row = e / C::QR;
      const int hy = row % C::HY, hz = (row / C::HY) % C::HZ, ci = row / (C::HY * C::HZ);
      const int z = z0 + hz - 1, y = y0 + hy - 1, xf = x0 - 1 + 4 * q;
      const bool rok = (z >= 0) & (z < D) & (y >= 0) & (y < H);
      int m = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) m |= (rok && xf + j >= 0 && xf + j < W) ? (1 << j) : 0;
      const long long lin = (long long)ci * (long long)DHW + ((long long)z * H + y) * W + xf;
      // The very first quad of the tensor starts one float BEFORE it: bit 16.  Only the first chunk of a slice that starts at channel 0
      // meets it, i.e. only the PROLOGUE's load (clamped to offset 0, shifted by one element at its commit); every later chunk adds
      // chunk_bytes to the same negative offset and is an ordinary load - the K loop's commit carries no shift code.
      if (rok && lin < 0) m |= 16;
      mq[i] = m;
      gq[i] = rok ? (int)(lin * 4) : 0;
      lq[i] = row * C::HXP + 4 * q;
    }
  }
  f32x4 stg[C::NI];
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(in_b), 0, (unsigned)((size_t)cin * DHW * sizeof(float)), 0x00020000);
  const int nchunk_all = (cin + CC - 1) / CC;
  const int c_begin = ep.ksplit > 1 ? (int)blockIdx.z * ep.cps : 0;
  const int nchunk = ep.ksplit > 1 ? min(nchunk_all, c_begin + ep.cps) : nchunk_all;     // one past this slice's last chunk
  if (ep.ksplit > 1) out += (size_t)blockIdx.z * ep.slice_stride;
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned w_pair_bytes = (unsigned)ncb_total * C::W_SEG * 4, w_tile_bytes = (unsigned)cot * C::W_SEG * 4;
  const int lane16 = lane * 16;
  const int chunk_bytes = (int)(CC * DHW * sizeof(float));
  auto issue_in = [&](int chunk, auto first) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      int off = gq[i] + chunk * chunk_bytes;
      if constexpr (decltype(first)::value) off = max(off, 0);
      stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));
    }
  };
  // LDS byte address of every staged quad in buffer 0 / 1: the K loop's two chunk bodies name their buffer at compile time
  unsigned lqa[2][C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    lqa[0][i] = (unsigned)(uintptr_t)(lds + lq[i]);
    lqa[1][i] = (unsigned)(uintptr_t)(lds + C::LDS_FLOATS + lq[i]);
  }
  auto commit_in = [&](auto kbuf, auto first) __attribute__((always_inline)) {
    constexpr int KB = decltype(kbuf)::value;
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {                    // branch-free: the K loop must stay one scheduling region
      const int m = mq[i];
      const f32x4 v = stg[i];
      float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
      if constexpr (decltype(first)::value) {
        const bool sh = (m & 16) != 0 && gq[i] + c_begin * chunk_bytes < 0;
        v3 = sh ? v2 : v3; v2 = sh ? v1 : v2; v1 = sh ? v0 : v1; v0 = sh ? 0.f : v0;
      }
      f32x4 o = {(m & 1) ? v0 : 0.f, (m & 2) ? v1 : 0.f, (m & 4) ? v2 : 0.f, (m & 8) ? v3 : 0.f};
      // pinned as ONE 16-byte register tuple: otherwise the store is split into two ds_write2_b32, whose 16-byte lane stride is a
      // 4-way bank conflict (PMC: half of the kernel's LDS cycles were conflicts, profiles/r03_mfma_busy.txt)
      asm volatile("" : "+v"(o));
      *reinterpret_cast<lds_f32x4*>((uintptr_t)lqa[KB][i]) = o;
    }
  };
  auto stage_w = [&](int chunk, auto kbuf) __attribute__((always_inline)) {
    constexpr int buf = decltype(kbuf)::value * C::LDS_FLOATS;
#pragma unroll
    for (int i = 0; i < C::NWD; ++i) {
      const int pc = wave8 * C::NWD + i;                 // 1 KB piece of the chunk's PP x 12 KB
      const int pr = pc / (C::W_SEG / 256), o = pc % (C::W_SEG / 256);
      lds_void* dst = reinterpret_cast<lds_void*>((uintptr_t)(lds + buf + C::IN_ELEMS + pc * 256));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, lane16,
                                               (int)((unsigned)(chunk * C::PP + pr) * w_pair_bytes + w_tile_bytes + (unsigned)o * 1024u), 0, 0);
    }
  };

  f32x16 acc[2][4];   // [eta - 2 * eh][xi]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[a][x][g] = 0.f;

  const int jt = (lane & 31) % XT, ju = (lane & 31) / XT;
  // B base: channel half, the wave's z plane, halo row 2*(wy*YT + ju) (= output row pair's y-1), s = 2*jt
  const int b_base = (lane >> 5) * C::CS + wz * (C::HY * C::HXP) + 2 * (wy * C::YT + ju) * C::HXP + 2 * jt;

  constexpr int NS = 3 * C::PP;                        // K steps per chunk: dz x channel pair
  static_assert(NS % 2 == 0 && NS >= 4, "the fragment rings are indexed statically across the chunk loop");
  // y transform of this eta half from three of the four halo rows:  cA = U - V,  cB = V +- P
  //   eh = 0 (eta 0, 1): U = row 0, V = row 2, P = row 1, cB = V + P   (d0 - d2, d1 + d2)
  //   eh = 1 (eta 2, 3): U = row 2, V = row 1, P = row 3, cB = V - P   (d2 - d1, d1 - d3)
  // The K loop is instantiated once per eta half (wave-uniform branch below): row offsets and the sign are immediates.
  f32x2 raw[2][3][2];
  float bfq[2][2][4], afq[2][8];
  auto kloop = [&](auto ehc) __attribute__((always_inline)) {
    constexpr int EH = decltype(ehc)::value;
    constexpr int rowU = (EH ? 2 : 0) * C::HXP, rowV = (EH ? 1 : 2) * C::HXP, rowP = (EH ? 3 : 1) * C::HXP;
    // Per buffer: two B bases (the low and the high 8-byte word of a row are read from DIFFERENT registers, or the load/store optimiser
    // fuses them into ds_read2_b64 - half the LDS rate of two ds_read_b64) and the A base; pinned, every read = base + immediate.
    unsigned bB0[2], bB1[2], bA[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      bB0[k] = (unsigned)(uintptr_t)(lds + k * C::LDS_FLOATS + b_base);
      bB1[k] = bB0[k] + 8;
      bA[k] = (unsigned)(uintptr_t)(lds + k * C::LDS_FLOATS + C::IN_ELEMS + 4 * lane);
      asm volatile("" : "+v"(bB0[k]), "+v"(bB1[k]), "+v"(bA[k]));
    }
    auto read_raw = [&](auto kbuf, int s, f32x2 (&r)[3][2]) __attribute__((always_inline)) {
      constexpr int KB = decltype(kbuf)::value;
      const int dz = s / C::PP, pp = s % C::PP;
#if (M3D_EXP & 16)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "=v"(r[i][j]));      // timing ablation: no raw LDS reads (opaque values)
      return;
#endif
      const unsigned off = (unsigned)(pp * 2 * C::CS + dz * (C::HY * C::HXP)) * 4u;
      const lds_f32x2* p0 = reinterpret_cast<const lds_f32x2*>((uintptr_t)(bB0[KB] + off));
      const lds_f32x2* p1 = reinterpret_cast<const lds_f32x2*>((uintptr_t)(bB1[KB] + off));
      r[0][0] = p0[rowU / 2]; r[0][1] = p1[rowU / 2];      // r[row][0] = (x 2t-1, x 2t) = (O[t], E[t]),  r[row][1] = (x 2t+1, x 2t+2) = (O[t+1], E[t+1])
      r[1][0] = p0[rowV / 2]; r[1][1] = p1[rowV / 2];
      r[2][0] = p0[rowP / 2]; r[2][1] = p1[rowP / 2];
    };
    auto transform = [&](const f32x2 (&rw)[3][2], float (&bf)[2][4]) __attribute__((always_inline)) {
      float r[3][4];                                     // (E[t], E[t+1], O[t], O[t+1]) per row
#pragma unroll
      for (int i = 0; i < 3; ++i) { r[i][0] = rw[i][0][1]; r[i][1] = rw[i][1][1]; r[i][2] = rw[i][0][0]; r[i][3] = rw[i][1][0]; }
#if (M3D_EXP & 8)
      for (int v = 0; v < 4; ++v) { bf[0][v] = r[0][v]; bf[1][v] = r[2][v] + r[1][v]; }   // timing ablation: 4 instead of 16 VALU
      return;
#endif
      float c[2][4];                                     // rows combined (y transform), still raw in x
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        c[0][v] = r[0][v] - r[1][v];
        c[1][v] = EH ? r[1][v] - r[2][v] : r[1][v] + r[2][v];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {                      // x transform: e0 = c[a][0], e1 = c[a][1], o0 = c[a][2], o1 = c[a][3]
        bf[a][0] = c[a][2] - c[a][3]; bf[a][1] = c[a][0] + c[a][3]; bf[a][2] = c[a][3] - c[a][0]; bf[a][3] = c[a][0] - c[a][1];
      }
    };
    auto load_a = [&](auto kbuf, int s, float (&af)[8]) __attribute__((always_inline)) {     // this half's 2 x 4 fragments of the step
      constexpr int KB = decltype(kbuf)::value;
      const int dz = s / C::PP, pp = s % C::PP;
#if (M3D_EXP & 32)
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("" : "=v"(af[q]));            // timing ablation: no weight-fragment LDS reads
      return;
#endif
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const f32x4 v = *reinterpret_cast<const lds_f32x4*>((uintptr_t)(bA[KB] + (unsigned)(pp * C::W_SEG + w2_slot(dz, EH * 2 + a, 0, 0)) * 4u));
        af[a * 4 + 0] = v[0]; af[a * 4 + 1] = v[1]; af[a * 4 + 2] = v[2]; af[a * 4 + 3] = v[3];
      }
    };

    constexpr std::integral_constant<int, 0> B0{};
    constexpr std::integral_constant<int, 1> B1{};
    read_raw(B0, 0, raw[0]);
    read_raw(B0, 1, raw[1]);
    load_a(B0, 0, afq[0]);
    transform(raw[0], bfq[0]);

    // ---- K loop, software-pipelined ACROSS chunks: raw rows are read two steps ahead, transformed one step ahead; weight fragments
    // are read one step ahead.  Every step is two scheduling regions:
    //   A: the step's LDS reads (raw rows of step s + 2, weight fragments of step s + 1) between MFMAs 0..3; step 0 also issues the
    //      next chunk's LDS-DMA (its buffer was last read before the previous chunk barrier)
    //   B: pin + transform of the rows read in the PREVIOUS step between MFMAs 4..7
    // The pin matters: without it the pre-RA scheduler places the y transform right behind its LDS reads (fewer live registers) and
    // every step waits out the LDS latency four times (tools/asm_trace.py: "r r r [lgkmcnt(1)] v v").  The chunk barrier sits at the
    // end of step NS-2, behind the input commit and a wait for the weight DMA issued in step 0; every LDS read of the current chunk has been
    // issued by then; the last step reads the next chunk's first fragments in its region A and transforms them in B.
    // (two chunk bodies, one per LDS buffer: the buffer is a compile-time constant in every address - see conv3d_wino24.hip)
    auto chunk_body = [&](auto curc, int chunk) __attribute__((always_inline)) {
      constexpr std::integral_constant<int, decltype(curc)::value> cur{};
      constexpr std::integral_constant<int, 1 - decltype(curc)::value> nxt{};
      const int nchk = min(chunk + 1, nchunk - 1);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        // ---------------- region A
        if (s + 2 < NS) read_raw(cur, s + 2, raw[s & 1]);
        if (s + 1 < NS) load_a(cur, s + 1, afq[(s + 1) & 1]);
        if (s == NS - 1) {                               // next chunk's first fragments (its buffer is complete: barrier below)
          read_raw(nxt, 0, raw[0]);
          read_raw(nxt, 1, raw[1]);
          load_a(nxt, 0, afq[0]);
        }
#if !(M3D_EXP & 1)
        if (s == 0) {
#if !(M3D_EXP & 256)
          stage_w(nchk, nxt);
#endif
#if !(M3D_EXP & 64)
          issue_in((M3D_EXP & 512) ? 0 : nchk, std::false_type{});
#endif
        }
#endif
#pragma unroll
        for (int x = 0; x < 4; ++x)
          acc[0][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[s & 1][x], bfq[s & 1][0][x], acc[0][x], 0, 0, 0);
#ifndef M3D_SGE_OFF
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         // VALU (addresses)
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);         // DS read
          if (s == 0) {                                              // the next chunk's loads; DMA pieces behind their M0 set-up
#pragma unroll
            for (int k = 0; k < (C::NI + C::NWD + 3) / 4; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);     // SALU
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
            }
          }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- region B
#ifndef M3D_NOPIN
        if (s + 1 < NS) pin_regs(raw[(s + 1) & 1]);
        if (s == NS - 1) pin_regs(raw[0]);
#endif
        if (s + 1 < NS) transform(raw[(s + 1) & 1], bfq[(s + 1) & 1]);
        if (s == NS - 1) transform(raw[0], bfq[0]);
#if !(M3D_EXP & 1)
#if (M3D_EXP & 128)
        if (s == NS - 2) {                               // timing ablation: wait for the loads, no LDS commit
#pragma unroll
          for (int i = 0; i < C::NI; ++i) asm volatile("" :: "v"(stg[i]));
        }
#else
        if (s == NS - 2) commit_in(nxt, std::false_type{});
#endif
#endif
#pragma unroll
        for (int x = 0; x < 4; ++x)
          acc[1][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[s & 1][4 + x], bfq[s & 1][1][x], acc[1][x], 0, 0, 0);
#ifndef M3D_SGE_OFF
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
          if (s == NS - 2) __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);   // VALU (transform; masks of the input commit)
          else __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);         // DS write
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
#if !(M3D_EXP & 2)
        if (s == NS - 2) __syncthreads();                // (waits for this wave's DMA and LDS reads first)
#endif
      }
    };
    for (int chunk = c_begin;;) {
      chunk_body(B0, chunk);
      if (++chunk >= nchunk) break;
      chunk_body(B1, chunk);
      if (++chunk >= nchunk) break;
    }
  };
  // ---- prologue: chunk 0 -> buffer 0
  stage_w(c_begin, std::integral_constant<int, 0>{}); issue_in(c_begin, std::true_type{});
  commit_in(std::integral_constant<int, 0>{}, std::true_type{});
  __syncthreads();
  W2_STAMP(1);
  if (eh) kloop(std::integral_constant<int, 1>{}); else kloop(std::integral_constant<int, 0>{});
  W2_STAMP(2);
  __syncthreads();                                     // the pool exchange below reuses the staging area

  // ---- inverse transform: over xi in the lane, over eta across the two halves:  y0 = m0 + m1 + m2,  y1 = m1 - m2 - m3
  // half 1 (eta 2, 3) hands (m2, -m2 - m3) for both columns to half 0 through LDS; half 0 finishes and stores.
  f32x16 yv[2][2];
  {
    f32x16 p0[2], p1[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      p0[a] = acc[a][0] + acc[a][1] + acc[a][2];
      p1[a] = acc[a][1] - acc[a][2] - acc[a][3];
    }
    float* xch = lds + (size_t)wave * 64 * 64 + lane;      // final barrier of the chunk loop has passed: staging area is free
    if (eh == 1) {
      const f32x16 a0 = p0[0], a1 = -p0[0] - p0[1], b0 = p1[0], b1 = -p1[0] - p1[1];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        xch[g * 64] = a0[g]; xch[(16 + g) * 64] = a1[g]; xch[(32 + g) * 64] = b0[g]; xch[(48 + g) * 64] = b1[g];
      }
    }
    __syncthreads();
    W2_STAMP(3);
    if (eh == 1) return;
    yv[0][0] = p0[0] + p0[1]; yv[1][0] = p0[1];
    yv[0][1] = p1[0] + p1[1]; yv[1][1] = p1[1];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      yv[0][0][g] += xch[g * 64]; yv[1][0][g] += xch[(16 + g) * 64]; yv[0][1][g] += xch[(32 + g) * 64]; yv[1][1][g] += xch[(48 + g) * 64];
    }
  }
  const int co0 = cot * 32 + 4 * (lane >> 5);
  const int z = z0 + wz;
  const int x = x0 + 2 * jt;
  const int y = y0 + 2 * (wy * C::YT + ju);

  if constexpr (POOL) {
    // conv + scale/shift + ReLU + MaxPool3d(2,2): the (y, x) 2x2 footprint is in the lane; the z pair is wave wz = 0 / 1
    float pooled[16];
    int pidx[16];                                      // AM: (dy, dx) of the first maximum inside the lane's 2x2 patch
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = min(co0 + (g & 3) + 8 * (g >> 2), cout - 1);
      const float sc = ep.scale ? ep.scale[co] : 1.f, sh = ep.shift ? ep.shift[co] : 0.f;
      float m = -INFINITY;
      int mi = 0;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float v = yv[r][c][g] * sc + sh;
          if (ep.relu) v = fmaxf(v, 0.f);
          if constexpr (AM) {
            if (v > m) { m = v; mi = 2 * r + c; }      // strict >: the first maximum in (dz, dy, dx) order, as maxpool2_fwd_kernel
          } else {
            m = fmaxf(m, v);
          }
        }
      pooled[g] = m;
      pidx[g] = mi;
    }
    float* red = lds + C::XCH_FLOATS + (size_t)wy * 16 * 64 + lane;   // behind the eta exchange area
    float* redi = red + 2 * 16 * 64;                   // the second half of the pool exchange area: the indices
    if (wz == 1) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        red[g * 64] = pooled[g];
        if constexpr (AM) redi[g * 64] = __int_as_float(pidx[g]);
      }
    }
    __syncthreads();
    if (wz == 1) return;
    const int PD = D / 2, PH = H / 2, PW = W / 2;
    const int zp = z0 >> 1, yp = y >> 1, xp = x >> 1;
    if (zp >= PD || yp >= PH || xp >= PW) return;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = co0 + (g & 3) + 8 * (g >> 2);
      if (co < cout) {
        const size_t o = ((size_t)b * cout + co) * ((size_t)PD * PH * PW) + ((size_t)zp * PH + yp) * PW + xp;
        const float up = red[g * 64];
        if constexpr (AM) {
          const bool upper = up > pooled[g];            // the z + 1 plane only wins when strictly larger
          out[o] = upper ? up : pooled[g];
          ep.argmax[o] = (unsigned char)(upper ? 4 + __float_as_int(redi[g * 64]) : pidx[g]);
        } else {
          out[o] = fmaxf(pooled[g], up);
        }
      }
    }
    W2_STAMP(4); W2_STAMP(6);
    return;
  }

  if (!(z < D && y < H && x < W)) return;
  const bool pair_ok = ((W & 1) == 0);
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int co = co0 + (g & 3) + 8 * (g >> 2);
    if (co >= cout) continue;
    const float sc = ep.scale ? ep.scale[co] : 1.f, sh = ep.shift ? ep.shift[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (y + r >= H) continue;
      float v0 = yv[r][0][g] * sc + sh, v1 = yv[r][1][g] * sc + sh;
      if (ep.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      float* o = out + ((size_t)b * cout + co) * DHW + ((size_t)z * H + y + r) * W + x;
      if (pair_ok) {
        *reinterpret_cast<f32x2*>(o) = f32x2{v0, v1};
      } else {
        o[0] = v0;
        if (x + 1 < W) o[1] = v1;
      }
    }
  }
  W2_STAMP(4); W2_STAMP(6);
}

// split-K epilogue: out = act(scale * sum_s partial[s] + shift), fixed summation order
__global__ __launch_bounds__(256) void wino2_reduce_kernel(const float* __restrict__ ws, int ksplit, size_t slice_stride,
                                                           float* __restrict__ out, int cout, size_t DHW, size_t total,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, int relu) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    float v = 0.f;
    for (int s = 0; s < ksplit; ++s) v += ws[s * slice_stride + i];
    const int co = (int)((i / DHW) % cout);
    v = v * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
    out[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

inline int xcd_map_enabled2() {
  return m3d::opt(m3d::OPT_XCD_MAP) != 0;
}

template <int CC, int XT, int WZ, int WY, bool POOL = false>
int launch_wino2_one(const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W, W2Epi ep, hipStream_t st,
                 int ksplit = 1) {
  using C = W2Cfg<CC, XT, WZ, WY, POOL>;
  const int tiles_x = (W + C::TX - 1) / C::TX, tiles_y = (H + C::TY - 1) / C::TY, tiles_z = (D + C::TZ - 1) / C::TZ;
  const int ncb_total = ((cout + 31) / 32 + 1) / 2 * 2;
  const int co_tiles = (cout + 31) / 32;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * co_tiles;
  if (blocks > 0x7FFFFFFFll || B > 65535) return M3D_EUNSUPPORTED;
  ep.xcd_map = xcd_map_enabled2();
  const size_t lds = sizeof(float) * C::SMEM_FLOATS;
  if (lds > 160 * 1024) return M3D_EUNSUPPORTED;
  auto kern = conv3d_wino2_kernel<CC, XT, WZ, WY, POOL>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, B, ksplit), dim3(C::NT), lds, st, in, wp, out, cin, cout, D, H, W, tiles_x, tiles_y,
                     tiles_z, ncb_total, ep);
  return m3d::check_launch("conv3d_wino2");
}


// eta-split variant: 8 waves, two per SIMD
template <int CC, int XT, int WZ, int WY, bool POOL = false, bool AM = false>
int launch_wino2e(const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W, W2Epi ep, hipStream_t st,
                  int ksplit = 1) {
  using C = W2CfgE<CC, XT, WZ, WY, POOL>;
  const int tiles_x = (W + C::TX - 1) / C::TX, tiles_y = (H + C::TY - 1) / C::TY, tiles_z = (D + C::TZ - 1) / C::TZ;
  const int ncb_total = ((cout + 31) / 32 + 1) / 2 * 2;
  const int co_tiles = (cout + 31) / 32;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * co_tiles;
  if (blocks > 0x7FFFFFFFll || B > 65535) return M3D_EUNSUPPORTED;
  ep.xcd_map = xcd_map_enabled2();
  const size_t lds = sizeof(float) * C::SMEM_FLOATS;
  if (lds > 160 * 1024) return M3D_EUNSUPPORTED;
#ifdef M3D_W2_STAMPS
  ep.stamps = g_w2_stamps;
#endif
  auto kern = conv3d_wino2e_kernel<CC, XT, WZ, WY, POOL, AM>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, B, ksplit), dim3(C::NT), lds, st, in, wp, out, cin, cout, D, H, W, tiles_x, tiles_y,
                     tiles_z, ncb_total, ep);
  return m3d::check_launch("conv3d_wino2e");
}

// tune_wino2 / 100 == 1 selects the one-wave-per-SIMD kernels (A/B measurements), else the eta-split kernels
template <int CC, int XT, int WZ, int WY, bool POOL = false>
int launch_wino2(const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W, W2Epi ep, hipStream_t st,
                 int ksplit = 1) {
  if (m3d::opt(m3d::OPT_TUNE_WINO2) / 100 == 1)
    return launch_wino2_one<CC, XT, WZ, WY, POOL>(in, wp, out, B, cin, cout, D, H, W, ep, st, ksplit);
  return launch_wino2e<CC, XT, WZ, WY, POOL>(in, wp, out, B, cin, cout, D, H, W, ep, st, ksplit);
}

}  // namespace

// The packed buffer holds BOTH weight packs: the F(2x2,3x3) pack (48 slots per (cout, cin); families 1-3) followed by the F(2x4,3x3)
// pack (72 slots; family 4), so the family can be switched between pack and forward (A/B runs) and one buffer serves every kernel.
namespace {
inline size_t pack22_floats(int cin, int cout) {
  const size_t npair = ((cin + 1) / 2 + 15) / 16 * 16, ncb = ((cout + 31) / 32 + 1) / 2 * 2;
  return npair * ncb * (size_t)WT2 * 64;
}
}  // namespace

M3D_API size_t m3d_conv3d_wino2_packed_weight_bytes(int cin, int cout) {
  if (cin <= 0 || cout <= 0) return 0;
  return sizeof(float) * (pack22_floats(cin, cout) + m3d_w24::packed_floats(cin, cout));
}

M3D_API int m3d_conv3d_wino2_pack_weights(const float* d_weight, int cin, int cout, float* d_packed, void* stream) {
  if (!d_weight || !d_packed || cin <= 0 || cout <= 0) return M3D_EINVAL;
  const int npair = ((cin + 1) / 2 + 15) / 16 * 16, ncb = ((cout + 31) / 32 + 1) / 2 * 2;
  hipLaunchKernelGGL(wino2_pack_kernel, dim3(1024), dim3(256), 0, m3d::as_stream(stream), d_weight, cin, cout, d_packed, ncb, npair);
  const int rc = m3d::check_launch("wino2_pack");
  if (rc != M3D_OK) return rc;
  return m3d_w24::pack(d_weight, cin, cout, d_packed + pack22_floats(cin, cout), m3d::as_stream(stream));
}

// ---- kernel family and tile choice.
// Families (option "tune_wino2" / 100): 0 = library default (family 4), 1 = one wave per SIMD (round 1), 2 = eta-split 8-wave
// workgroups (round 2 / 3), 3 = quad kernel (4-wave workgroups, two per CU; conv3d_wino2q.hip), 4 = F(2x4,3x3): F(4,3) along x, 3/4 of
// F(2x2)'s matrix-core work (conv3d_wino24.hip), 5 = the same arithmetic with 4-wave workgroups that feed each B fragment to two
// output-channel blocks (conv3d_wino24w.hip).  "tune_wino2" % 100: 99 = library tile
// choice, 0..5 = one of the fixed tiles of families 1 / 2 (A/B runs).
// Tiles (x, y, z outputs per workgroup): eta-split / one-wave 64 x 2 x 4, 32 x 8 x 2, 16 x 16 x 2; quad 64 x 2 x 2, 32 x 4 x 2,
// 16 x 8 x 2; the narrowest with split-K over workgroups when the map has too few tiles to fill the chip.  Score = useful fraction of
// the computed tile volume x how much of the chip the grid fills; ties go to the wider tile (fewer halo columns per output).
namespace {
constexpr int kDefaultFamily = 4;
inline int family() {
  const int f = m3d::opt(m3d::OPT_TUNE_WINO2) / 100;
  return f <= 0 ? kDefaultFamily : f;
}
struct Tile { int tx, ty, tz; };
inline Tile tile_of(int fam, int xt) {
  if (fam == 3) return xt == 32 ? Tile{64, 2, 2} : xt == 16 ? Tile{32, 4, 2} : Tile{16, 8, 2};
  if (fam == 4) return xt == 32 ? Tile{64, 4, 2} : xt == 16 ? Tile{32, 8, 2} : Tile{16, 16, 2};
  if (fam == 5) return xt == 32 ? Tile{64, 2, 2} : xt == 16 ? Tile{32, 4, 2} : Tile{16, 8, 2};      // x 64 output channels
  return xt == 32 ? Tile{64, 2, 4} : xt == 16 ? Tile{32, 8, 2} : Tile{16, 16, 2};
}
inline int chip_slots(int fam) { return fam == 3 ? 512 : 256; }   // resident workgroups: two per CU for the quad kernel
inline int cout_tile(int fam) { return fam == 5 ? 64 : 32; }      // output channels per workgroup
// family 5 (conv3d_wino24w.hip) feeds a B fragment to two 32-channel blocks: layers whose cout is not a multiple of 64 run family 4
inline int family_for(int fam, int cout) { return (fam == 5 && (cout & 63)) ? 4 : fam; }

struct SplitPlan { int ksplit, cps; size_t slice; };
SplitPlan plan_splitk(int fam, int batch, int cin, int cout, int depth, int height, int width) {
  const Tile t = tile_of(fam, 8);
  const long long tiles = (long long)((width + t.tx - 1) / t.tx) * ((height + t.ty - 1) / t.ty) * ((depth + t.tz - 1) / t.tz) *
                          ((cout + cout_tile(fam) - 1) / cout_tile(fam)) * batch;
  const int nchunk = (cin + 3) / 4;
  int ks = (int)(chip_slots(fam) / (tiles > 0 ? tiles : 1));   // one resident round, never a ragged second one
  if (ks > 8) ks = 8;
  if (ks > nchunk) ks = nchunk;
  if (ks < 1) ks = 1;
  const int cps = (nchunk + ks - 1) / ks;
  ks = (nchunk + cps - 1) / cps;                             // no empty slice
  return SplitPlan{ks, cps, (size_t)batch * cout * depth * height * width};
}

int choose_xt(int fam, int batch, int cin, int cout, int D, int H, int W, double* best_score = nullptr) {
  if (best_score) *best_score = 0.0;
  if (const int tv = m3d::opt(m3d::OPT_TUNE_WINO2_XT); tv >= 0) { if (best_score) *best_score = 1.0; return tv; }
  if (W < 12) return 0;
  auto up = [](int v, int t) { return (double)((v + t - 1) / t) * t; };
  const double vol = (double)D * H * W, cot = (cout + cout_tile(fam) - 1) / cout_tile(fam), slots = chip_slots(fam);
  double best = -1.0; int xt = 0;
  const int id[3] = {32, 16, 8};
  for (int i = 0; i < 3; ++i) {
    if (id[i] == 32 && W < 48) continue;
    if (id[i] == 16 && W < 24) continue;
    const Tile t = tile_of(fam, id[i]);
    const double eff = vol / (up(W, t.tx) * up(H, t.ty) * up(D, t.tz));
    double wgs = up(W, t.tx) / t.tx * up(H, t.ty) / t.ty * up(D, t.tz) / t.tz * cot * batch;
    if (id[i] == 8) wgs *= plan_splitk(fam, batch, cin, cout, D, H, W).ksplit;
    // the grid runs in ceil(wgs / slots) rounds and a ragged last round costs a full one
    const double rounds = (double)((long long)((wgs + slots - 1.0) / slots));
    const double score = eff * wgs / (slots * rounds);
    if (score > best * 1.02) { best = score; xt = id[i]; }
  }
  if (best_score) *best_score = best;
  return xt;
}

m3d_w2q::Epi quad_epi(const W2Epi& e) {
  m3d_w2q::Epi q{};
  q.scale = e.scale; q.shift = e.shift; q.relu = e.relu; q.xcd_map = m3d::opt(m3d::OPT_XCD_MAP);   // 0 off, 1 on (tile order chosen per shape), 2 / 3: A/B
  q.ksplit = e.ksplit; q.cps = e.cps; q.slice_stride = e.slice_stride; q.argmax = e.argmax;
  return q;
}

// one launch of the chosen family on tile xt (8: the split-K capable tile)
int launch_family(int fam, int xt, const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W, W2Epi ep,
                  hipStream_t st, int ksplit) {
  if (fam == 3) return m3d_w2q::launch(xt, false, false, in, wp, out, B, cin, cout, D, H, W, quad_epi(ep), st);
  if (fam == 4) return m3d_w24::launch(xt, false, false, in, wp + pack22_floats(cin, cout), out, B, cin, cout, D, H, W, quad_epi(ep), st);
  if (fam == 5) return m3d_w24w::launch(xt, false, false, in, wp + pack22_floats(cin, cout), out, B, cin, cout, D, H, W, quad_epi(ep), st);
  if (xt == 32) return launch_wino2<4, 32, 4, 1>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  if (xt == 16) return launch_wino2<4, 16, 2, 2>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  if (xt == 8) return launch_wino2<4, 8, 2, 2>(in, wp, out, B, cin, cout, D, H, W, ep, st, ksplit);
  return M3D_EUNSUPPORTED;
}
}  // namespace

/* the 2-D Winograd family the library currently runs (1, 2, 3: F(2x2,3x3), 16/36 of the direct multiplies; 4: F(2x4,3x3), 24/72) */
M3D_API int m3d_conv3d_wino2_family(void) { return family(); }

/* useful-work x chip-fill score (0..1) of the best tile for this shape; callers use the direct kernel below ~0.3
 * (measured with the F(2x4,3x3) family: 128 -> 128 channels on 16 x 40 x 40: score 0.39, 0.194 ms vs 0.253 ms direct) */
M3D_API double m3d_conv3d_wino2_score(int batch, int cin, int cout, int depth, int height, int width) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return 0.0;
  double sc = 0.0;
  (void)choose_xt(family_for(family(), cout), batch, cin, cout, depth, height, width, &sc);
  return sc;
}

/* the same score for the exactly-local F(2x2,3x3) family (m3d_conv3d_wino2_local_forward_ws): its tiles differ from the default
 * family's (64 x 2 x 4 against 64 x 4 x 2), so a go / no-go decision for the local path must ask about ITS tiles */
M3D_API double m3d_conv3d_wino2_local_score(int batch, int cin, int cout, int depth, int height, int width) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return 0.0;
  double sc = 0.0;
  (void)choose_xt(family_for(2, cout), batch, cin, cout, depth, height, width, &sc);
  return sc;
}

namespace {
size_t workspace_bytes_for(int fam, int batch, int cin, int cout, int depth, int height, int width) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return 0;
  fam = family_for(fam, cout);
  if (choose_xt(fam, batch, cin, cout, depth, height, width) != 8) return 0;
  const SplitPlan p = plan_splitk(fam, batch, cin, cout, depth, height, width);
  return p.ksplit > 1 ? p.ksplit * p.slice * sizeof(float) : 0;
}
}  // namespace

M3D_API size_t m3d_conv3d_wino2_workspace_bytes(int batch, int cin, int cout, int depth, int height, int width) {
  return workspace_bytes_for(family(), batch, cin, cout, depth, height, width);
}

/* The "_local" pair: the F(2x2,3x3) family, whose outputs depend on NOTHING outside their own 3x3 (y, x) support - not even by
 * rounding (F(2,3)'s two outputs are sums of products that contain only their own three inputs).  F(4,3) along x is local only in
 * exact arithmetic: the other inputs of its 6-wide footprint cancel to ~1e-7 of THEIR magnitude.  The PRM back-propagation lays
 * windows of different peaks side by side with one zero column between them (m3d_prm_prepare_ex) and needs the exact form. */
M3D_API size_t m3d_conv3d_wino2_local_workspace_bytes(int batch, int cin, int cout, int depth, int height, int width) {
  return workspace_bytes_for(2, batch, cin, cout, depth, height, width);
}

namespace {
int forward_ws_family(int fam, const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                      int depth, int height, int width, const float* d_scale, const float* d_shift, int relu,
                      void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_in || !d_packed || !d_out || batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0)
    return M3D_EINVAL;
  const size_t DHW = (size_t)depth * height * width;
  if ((size_t)cin * DHW * sizeof(float) >= 0x7FFFFFFFull || batch > 65535) return M3D_EUNSUPPORTED;
  hipStream_t st = m3d::as_stream(stream);
  fam = family_for(fam, cout);
  W2Epi ep{d_scale, d_shift, relu, 0, 1, 0, 0};
  const int variant = m3d::opt(m3d::OPT_TUNE_WINO2) % 100;
  if (fam != 3 && fam != 4 && fam != 5) {
#define M3D_W2(i, ...) if (variant == i) return launch_wino2<__VA_ARGS__>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
    M3D_W2(0, 4, 32, 2, 2)      // 64 x 4 y x 2 z outputs x 32 channels
    M3D_W2(1, 4, 32, 4, 1)
    M3D_W2(2, 4, 32, 1, 4)
    M3D_W2(3, 4, 16, 2, 2)      // 32 x 8 y x 2 z
    M3D_W2(4, 4, 16, 4, 1)
    M3D_W2(5, 4, 16, 1, 4)
#undef M3D_W2
  }
  if (variant >= 0 && variant != 99 && m3d::opt(m3d::OPT_TUNE_WINO2) >= 0) return M3D_EUNSUPPORTED;    // 99: library tile choice
  const int xt = choose_xt(fam, batch, cin, cout, depth, height, width);
  if (xt == 32 || xt == 16) return launch_family(fam, xt, d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st, 1);
  if (xt != 8) return M3D_EUNSUPPORTED;
  const SplitPlan p = plan_splitk(fam, batch, cin, cout, depth, height, width);
  if (p.ksplit <= 1) return launch_family(fam, 8, d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st, 1);
  if (!d_ws || ws_bytes < p.ksplit * p.slice * sizeof(float)) return M3D_EWORKSPACE;
  W2Epi eps{nullptr, nullptr, 0, 0, p.ksplit, p.cps, p.slice};
  const int rc = launch_family(fam, 8, d_in, d_packed, (float*)d_ws, batch, cin, cout, depth, height, width, eps, st, p.ksplit);
  if (rc != M3D_OK) return rc;
  const size_t total = p.slice;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino2_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)d_ws, p.ksplit, p.slice, d_out, cout,
                     DHW, total, d_scale, d_shift, relu);
  return m3d::check_launch("wino2_reduce");
}
}  // namespace

/* Peak back-propagation through a 3^3 conv on the quad-aligned strip layout FUSED with the next layer's prepare step (csrc/conv3d_wino24.h
 * PrepEpi): d_gn = the prepared gradient strip of the conv being back-propagated [cin, in_planes, window, L(window)] (in_planes = window,
 * or with in_slab the map's depth), d_packed = m3d_conv3d_wino2_pack_weights of that conv's backward-data weights (flipped, transposed
 * relu(W): cin -> cout).  Writes the prepared strip of the layer below, d_out [cout, out_planes, window + 2, L(window + 2)] (out_planes =
 * window + 2, or with out_slab the map's depth), zero-filled here, and d_origin_out = d_origin - 1.  d_xnext / d_norm / d_scale /
 * d_up_offset: as m3d_prm_prepare_ex2 of the layer below with pool = 0, border = 1.  M3D_EUNSUPPORTED when the library would run this
 * shape through another kernel family or with split-K (the caller then takes the two-launch path: conv, then m3d_prm_prepare_ex2). */
M3D_API int m3d_prm_strip_dgrad_prepare(const float* d_gn, const float* d_packed, int cin, int cout, int num_peaks, int window, int in_slab,
                                        const int32_t* d_origin, const float* d_xnext, const float* d_norm, const float* d_scale,
                                        const float* d_up_offset, int depth, int height, int width, int out_slab, float* d_out,
                                        int32_t* d_origin_out, void* stream) {
  if (num_peaks < 0 || cin <= 0 || cout <= 0 || window <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_gn || !d_packed || !d_origin || !d_xnext || !d_norm || !d_up_offset || !d_out || !d_origin_out) return M3D_EINVAL;
  const int fam = family_for(family(), cout);
  if (fam != 4 || m3d::opt(m3d::OPT_TUNE_WINO2) >= 0) return M3D_EUNSUPPORTED;
  int pitchA, leadA, pitchB, leadB; long long LA, LB;
  m3d::strip_geom(window, 2, num_peaks, &pitchA, &leadA, &LA);
  m3d::strip_geom(window + 2, 2, num_peaks, &pitchB, &leadB, &LB);
  const int ZA = in_slab ? depth : window, ZB = out_slab ? depth : window + 2;
  if ((size_t)cin * ZA * window * LA * sizeof(float) >= 0x7FFFFFFFull || (long long)(window + 2) * LB >= 0x7FFFFFFFll || LA >= (1 << 22))
    return M3D_EUNSUPPORTED;
  if ((long long)cout * depth * height * width >= 0x7FFFFFFFll || width < 4) return M3D_EUNSUPPORTED;   // 32-bit map offsets, 16-byte row reads
  const int xt = choose_xt(fam, 1, cin, cout, ZA, window, (int)LA);
  if (xt != 32 && xt != 16 && !(xt == 8 && plan_splitk(fam, 1, cin, cout, ZA, window, (int)LA).ksplit <= 1)) return M3D_EUNSUPPORTED;
  hipStream_t st = m3d::as_stream(stream);
  const size_t out_bytes = (size_t)cout * ZB * (window + 2) * LB * sizeof(float);
  if (hipMemsetAsync(d_out, 0, out_bytes, st) != hipSuccess) return M3D_ELAUNCH;
  m3d_w24::PrepEpi pe{};
  pe.xnext = d_xnext; pe.norm = d_norm; pe.scale = d_scale; pe.xoff = d_up_offset; pe.origin = d_origin; pe.origin_out = d_origin_out;
  pe.P = num_peaks; pe.U = window; pe.MD = depth; pe.MH = height; pe.MW = width;
  pe.pitchA = pitchA; pe.leadA = leadA; pe.slabA = in_slab ? 1 : 0;
  pe.pitchB = pitchB; pe.leadB = leadB; pe.slabB = out_slab ? 1 : 0;
  pe.LB = LB; pe.ocs = (long long)ZB * (window + 2) * LB; pe.ozs = (int)((window + 2) * LB);
  pe.inv_pitchA = 1.0f / (float)pitchA;
  W2Epi ep{nullptr, nullptr, 0, 0, 1, 0, 0};
  return m3d_w24::launch_prep(xt, d_gn, d_packed + pack22_floats(cin, cout), d_out, cin, cout, ZA, window, (int)LA, quad_epi(ep), pe, st);
}

/* What m3d_conv3d_wino2_forward_ws (local = 0) / m3d_conv3d_wino2_local_forward_ws (local = 1) would launch for this shape: the kernel
 * family (2: F(2x2,3x3), 4: F(2x4,3x3), ...), the tile id (32 / 16 / 8, 0: no tile) and the K split (1: every output is ONE accumulation
 * chain over the input channels; s > 1: s partial sums over channel slices added in a fixed order - a different summation order).
 * The plan is a pure function of the shape, so it tells apart two calls whose results may differ in the last bits (tests/test_gpu_prm.py:
 * the strips of a batch of P peaks and of a sub-batch are different shapes).  No launch, no device access. */
M3D_API int m3d_conv3d_wino2_plan(int local, int batch, int cin, int cout, int depth, int height, int width, int* family_out, int* tile_out,
                                  int* ksplit_out) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  const int fam = family_for(local ? 2 : family(), cout);
  const int xt = choose_xt(fam, batch, cin, cout, depth, height, width);
  if (family_out) *family_out = fam;
  if (tile_out) *tile_out = xt;
  if (ksplit_out) *ksplit_out = xt == 8 ? plan_splitk(fam, batch, cin, cout, depth, height, width).ksplit : 1;
  return M3D_OK;
}

M3D_API int m3d_conv3d_wino2_forward_ws(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                        int depth, int height, int width, const float* d_scale, const float* d_shift, int relu,
                                        void* d_ws, size_t ws_bytes, void* stream) {
  return forward_ws_family(family(), d_in, d_packed, d_out, batch, cin, cout, depth, height, width, d_scale, d_shift, relu, d_ws, ws_bytes,
                           stream);
}

M3D_API int m3d_conv3d_wino2_local_forward_ws(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                              int depth, int height, int width, const float* d_scale, const float* d_shift, int relu,
                                              void* d_ws, size_t ws_bytes, void* stream) {
  return forward_ws_family(2, d_in, d_packed, d_out, batch, cin, cout, depth, height, width, d_scale, d_shift, relu, d_ws, ws_bytes, stream);
}

/* without a workspace: fails with M3D_EWORKSPACE where the split-K tile would be chosen */
M3D_API int m3d_conv3d_wino2_forward(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout, int depth,
                                     int height, int width, const float* d_scale, const float* d_shift, int relu, void* stream) {
  return m3d_conv3d_wino2_forward_ws(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, d_scale, d_shift, relu, nullptr, 0,
                                     stream);
}

M3D_API int m3d_conv3d_wino2_forward_pool2(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                           int depth, int height, int width, const float* d_scale, const float* d_shift, int relu,
                                           void* stream) {
  if (!d_in || !d_packed || !d_out || batch <= 0 || cin <= 0 || cout <= 0 || depth < 2 || height < 2 || width < 2) return M3D_EINVAL;
  const size_t DHW = (size_t)depth * height * width;
  if ((size_t)cin * DHW * sizeof(float) >= 0x7FFFFFFFull || width < 24) return M3D_EUNSUPPORTED;
  W2Epi ep{d_scale, d_shift, relu, 0, 1, 0, 0};
  hipStream_t st = m3d::as_stream(stream);
  // 64-wide tiles from 48 voxels on, else 32-wide (conv3b on 32^3 maps: the pool of the 16^3-class layers is fused as well)
  if (family() == 3)
    return m3d_w2q::launch(width < 48 ? 16 : 32, true, false, d_in, d_packed, d_out, batch, cin, cout, depth, height, width, quad_epi(ep), st);
  if (family_for(family(), cout) == 5)
    return m3d_w24w::launch(width < 48 ? 16 : 32, true, false, d_in, d_packed + pack22_floats(cin, cout), d_out, batch, cin, cout, depth, height,
                            width, quad_epi(ep), st);
  if (family() == 4 || family() == 5)
    return m3d_w24::launch(width < 48 ? 16 : 32, true, false, d_in, d_packed + pack22_floats(cin, cout), d_out, batch, cin, cout, depth, height,
                           width, quad_epi(ep), st);
  if (width < 48) return launch_wino2<4, 16, 2, 2, true>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
  return launch_wino2<4, 32, 2, 2, true>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
}

/* m3d_conv3d_wino2_forward_pool2 that also returns the pool's argmax (uint8 [batch, cout, D/2, H/2, W/2], index = dz*4 + dy*2 + dx
 * of the first maximum, the rule of m3d_maxpool3d_2x_forward): the response conv of the pooled layers in PRM mode. */
M3D_API int m3d_conv3d_wino2_forward_pool2_argmax(const float* d_in, const float* d_packed, float* d_out, unsigned char* d_argmax,
                                                  int batch, int cin, int cout, int depth, int height, int width, const float* d_scale,
                                                  const float* d_shift, int relu, void* stream) {
  if (!d_in || !d_packed || !d_out || !d_argmax || batch <= 0 || cin <= 0 || cout <= 0 || depth < 2 || height < 2 || width < 2)
    return M3D_EINVAL;
  const size_t DHW = (size_t)depth * height * width;
  if ((size_t)cin * DHW * sizeof(float) >= 0x7FFFFFFFull || width < 24) return M3D_EUNSUPPORTED;
  W2Epi ep{d_scale, d_shift, relu, 0, 1, 0, 0, d_argmax};
  hipStream_t st = m3d::as_stream(stream);
  if (family() == 3)
    return m3d_w2q::launch(width < 48 ? 16 : 32, true, true, d_in, d_packed, d_out, batch, cin, cout, depth, height, width, quad_epi(ep), st);
  if (family_for(family(), cout) == 5)
    return m3d_w24w::launch(width < 48 ? 16 : 32, true, true, d_in, d_packed + pack22_floats(cin, cout), d_out, batch, cin, cout, depth, height,
                            width, quad_epi(ep), st);
  if (family() == 4 || family() == 5)
    return m3d_w24::launch(width < 48 ? 16 : 32, true, true, d_in, d_packed + pack22_floats(cin, cout), d_out, batch, cin, cout, depth, height,
                           width, quad_epi(ep), st);
  if (width < 48) return launch_wino2e<4, 16, 2, 2, true, true>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
  return launch_wino2e<4, 32, 2, 2, true, true>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
}
