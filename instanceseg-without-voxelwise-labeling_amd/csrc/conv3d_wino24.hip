// 3x3x3 forward convolution with Winograd F(2x4, 3x3) on the (y, x) plane: F(2,3) along y, F(4,3) along x - 24 multiplies per
// 2 x 4 output patch and z tap instead of 72 (direct) or 32 (the F(2x2,3x3) kernels of conv3d_wino2.hip): ONE THIRD of the
// direct convolution's matrix-core work, 3/4 of F(2x2)'s.  Same operation and epilogues as the other families (conv + eval-BN
// + ReLU [+ MaxPool3d(2,2) [+ arg-max]] of lib/modeling/DSN.py:58-67; split-K partial sums), same workgroup tiles
// (64 x 4 x 2, 32 x 8 x 2, 16 x 16 x 2 outputs x 32 output channels), so the host-side tile choice is shared.
//
//   per 2 x 4 output patch and (ci, dz):  M[eta][xi] = (By^T d Bx)[eta][xi] * (Gy g Gx^T)[eta][xi],  Y = Ay^T M Ax,
//   eta = 0..3 (F(2,3): the matrices of conv3d_wino2.hip), xi = 0..5 (F(4,3), interpolation points 0, +-1, +-2, inf):
//     Bx^T d = (4d0 - 5d2 + d4, -4d1 - 4d2 + d3 + d4, 4d1 - 4d2 - d3 + d4, -2d1 - d2 + 2d3 + d4, 2d1 - d2 - 2d3 + d4, 4d1 - 5d3 + d5)
//     Gx g   = (g0/4, -(g0+g1+g2)/6, -(g0-g1+g2)/6, (g0+2g1+4g2)/24, (g0-2g1+4g2)/24, g2)
//     Ax^T m = (m0+m1+m2+m3+m4, m1-m2+2m3-2m4, m1+m2+4m3+4m4, m1-m2+8m3-8m4+m5)
//   fp32 error against fp64: about twice F(2x2)'s (simulated and measured: <= 8e-6 of max|out| at 256 input channels), far inside
//   the 1e-4 contract.
//
// Decomposition (what round 3's ablations of the F(2x2) kernel asked for - fewer non-MFMA instructions per MFMA matter, overlap
// does not): 8 waves = 4 eta x 2 tile positions; a wave owns ONE eta row = 6 accumulator blocks (96 registers) for 32 output
// channels x 32 patches.  Per K step (one channel pair, one dz) it reads two halo rows (16 + 8 bytes each), forms its 6 B
// fragments with 18 VALU operations and issues 6 MFMAs; the eta rows meet in an LDS exchange at the end, after which every wave
// finishes one quarter of the channels (both output rows).  Staging as in the F(2x2) kernel: input quads through registers, weights by LDS-DMA.
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "conv3d_wino24.h"

// diagnostic build only (make w2_stamps; tools/w2_stamps.py): s_memtime stamps of one wave per workgroup, to a buffer of their own
#ifdef M3D_W2_STAMPS
static unsigned long long* g_w24_stamps = nullptr;
M3D_API void m3d_debug_set_stamp_buffer_24(void* p) { g_w24_stamps = (unsigned long long*)p; }
#define W24_STAMP(k) do { if (ep.stamps && tid == 0) ep.stamps[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 8 + (k)] = \
    (k) == 5 || (k) == 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W24_STAMP(k) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;

constexpr int WT24 = 72;                         // weight slots per (cout, cin): 3 dz x 4 eta x 6 xi
constexpr int SEG24 = WT24 * 64;                 // floats of one (channel pair, cout block): [12][64][4] (xi 0..3) then [12][64][2] (xi 4, 5)
constexpr int SEG24_HI = 12 * 64 * 4;            // offset of the xi 4, 5 part

// Wp24[cin_pair][cout_block32] = { [dz*4 + eta][lane64][xi 0..3], [dz*4 + eta][lane64][xi 4..5] } with
// (Gy g_dz Gx^T)[eta][xi], co = cb*32 + (lane&31), ci = 2*pair + (lane>>5); transformed in double
__global__ __launch_bounds__(256) void wino24_pack_kernel(const float* __restrict__ w, int cin, int cout, float* __restrict__ wp,
                                                          int ncb, int npair) {
  const long long total = (long long)npair * ncb * SEG24;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(e % SEG24);
    long long t = e / SEG24;
    const int cb = (int)(t % ncb);
    const int cpair = (int)(t / ncb);
    int grp, lane, xi;
    if (r < SEG24_HI) { xi = r & 3; lane = (r >> 2) & 63; grp = r >> 8; }
    else { const int q = r - SEG24_HI; xi = 4 + (q & 1); lane = (q >> 1) & 63; grp = q >> 7; }
    const int co = cb * 32 + (lane & 31), ci = 2 * cpair + (lane >> 5);
    float v = 0.f;
    if (co < cout && ci < cin) {
      const int dz = grp >> 2, eta = grp & 3;
      const float* g = w + ((size_t)co * cin + ci) * 27 + dz * 9;      // g[dy*3 + dx]
      double col[3];                                                     // (Gy g)[eta][dx]
      for (int dx = 0; dx < 3; ++dx) {
        const double a = g[dx], b = g[3 + dx], c = g[6 + dx];
        col[dx] = eta == 0 ? a : eta == 1 ? 0.5 * (a + b + c) : eta == 2 ? 0.5 * (a - b + c) : c;
      }
      const double g0 = col[0], g1 = col[1], g2 = col[2];
      const double r6 = 1.0 / 6.0, r24 = 1.0 / 24.0;
      const double u = xi == 0 ? 0.25 * g0 : xi == 1 ? -r6 * (g0 + g1 + g2) : xi == 2 ? -r6 * (g0 - g1 + g2)
                     : xi == 3 ? r24 * (g0 + 2.0 * g1 + 4.0 * g2) : xi == 4 ? r24 * (g0 - 2.0 * g1 + 4.0 * g2) : g2;
      v = (float)u;
    }
    wp[e] = v;
  }
}

__device__ __forceinline__ int xcd_contiguous24(int bid, int n) {
  const int per = n >> 3, rem = n & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

template <int A, int B>
__device__ __forceinline__ void pin24(float (&r)[A][B]) {
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b < B; ++b) asm volatile("" : "+v"(r[a][b]));
}

// CC input channels per chunk; XQ x-quads and YT = 32/XQ y-pairs per wave block (a patch = 2 rows x 4 columns); WZ x WY = 2 positions.
template <int CC, int XQ, int WZ, int WY, bool POOL>
struct Cfg24 {
  static constexpr int NT = 512;
  static constexpr int PP = CC / 2;
  static constexpr int YT = 32 / XQ;
  static constexpr int TX = 4 * XQ, TY = 2 * YT * WY, TZ = WZ;
  // halo rows in natural x order: element s <-> x = x0 - 1 + s; the lane of x quad t reads s = 4t .. 4t + 5 as one 16-byte and one
  // 8-byte word.  Row pitch HXP: the YT y-pairs of a block sit 2*HXP floats apart and must fall on disjoint banks for the 16-byte
  // reads (XQ * 16 bytes per y-pair out of 256, served in the lane groups of ds_read_b128): XQ = 16 2*HXP*4 = 0 (mod 256),
  // XQ = 8 = 128 (mod 256), XQ = 4 = 192 (mod 256) - checked group by group
  static constexpr int ROW = 4 * XQ + 2;
  static constexpr int QR = (ROW + 3) / 4;                     // 16-byte quads per row that carry data
  static constexpr int HXP = XQ == 16 ? 96 : (XQ == 8 ? 48 : 24);
  static constexpr int HY = TY + 2, HZ = TZ + 2;
  static constexpr int CS = HXP * HY * HZ;
  static constexpr int NQUAD = CC * HZ * HY * QR;
  static constexpr int NI = (NQUAD + NT - 1) / NT;
  static constexpr int IN_ELEMS = CC * CS + 4;                 // + a 16-byte dump slot
  static constexpr int DUMP = CC * CS;
  static constexpr int W_ELEMS = PP * SEG24;
  static constexpr int NPIECE = W_ELEMS / 256;                 // 1 KB LDS-DMA pieces per chunk
  static constexpr int NWD = (NPIECE + 7) / 8;                 // per wave (the last round may be partial)
  static constexpr int W_PAD = 256;                            // dump piece for the partial round
  static constexpr int LDS_FLOATS = IN_ELEMS + W_ELEMS + W_PAD;
  static constexpr int XCH_FLOATS = 2 * 4 * 64 * 64;           // eta exchange: 2 positions x 4 eta rows x 64 floats x 64 lanes
  static constexpr int RED_FLOATS = XCH_FLOATS + (POOL ? 8 * 16 * 64 : 0);   // pool exchange: values + arg-max indices of 2 positions x 2 x-halves
  static constexpr int SMEM_FLOATS = 2 * LDS_FLOATS > RED_FLOATS ? 2 * LDS_FLOATS : RED_FLOATS;
  static_assert(WZ * WY == 2, "2 tile positions per workgroup");
  static_assert(!POOL || (WZ == 2 && WY == 1), "fused pool: the z pair lives in the two positions");
  static_assert(ROW <= HXP && HXP % 4 == 0, "rows are whole 16-byte quads");
  static_assert(SMEM_FLOATS * 4 <= 160 * 1024, "LDS");
};

template <int CC, int XQ, int WZ, int WY, bool POOL, bool AM, bool PREP = false>
__global__ __launch_bounds__(512, 2) void conv3d_wino24_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                              float* __restrict__ out, int cin, int cout, int D, int H, int W,
                                                              int tiles_x, int tiles_y, int tiles_z, int ncb_total, m3d_w2q::Epi ep,
                                                              m3d_w24::PrepEpi pe) {
  using C = Cfg24<CC, XQ, WZ, WY, POOL>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  W24_STAMP(0); W24_STAMP(5);
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int eta = wave8 & 3, pos = wave8 >> 2;            // waves w and w + 4 (one SIMD): the same eta row of the two tile positions
  const int wz = pos / WY, wy = pos % WY;

  int bid = blockIdx.x;
  const int co_tiles = (cout + 31) / 32;
  // XCD-contiguous order with the cout tile FASTEST, z tiles in groups of 4 inside the y sweep (as conv3d_wino2.hip)
  if (ep.xcd_map) bid = xcd_contiguous24(bid, gridDim.x);
  const int cot = bid % co_tiles; bid /= co_tiles;
  // Wide volumes (the PRM strips: every peak's window side by side along x, 36 tiles of 64 columns) take the x tile SLOWEST: the ~64
  // workgroups an XCD holds at a time are then a compact (z, y) brick of one x range, whose halo rows and planes (6 rows for 4, 4 planes
  // for 2: 3.1 x the tile's own voxels) are in that XCD's L2 while the neighbours fetch them, instead of 32 x tiles of one (y, z) row that
  // share nothing (round 4's PMC: 2.97 GB fetched for 0.71 GB of strip).  xcd_map 2 / 3 force the x-fast / x-slow order (A/B).
  // Only strips are that wide: the forward / pooled convs of the 128- and 200-wide maps have 2 - 13 x tiles whatever the tile id and keep
  // the x-fast order their A/B runs and PMC data were taken with (round 5 switched every launch with more than 4 x tiles).  The other
  // families' kernels read xcd_map as a boolean: 2 / 3 mean "on" there.
  const bool x_slow = ep.xcd_map == 3 || (ep.xcd_map == 1 && tiles_x > 16);
  int tx;
  if (x_slow) { const int per_x = tiles_y * tiles_z; tx = bid / per_x; bid -= tx * per_x; }
  else { tx = bid % tiles_x; bid /= tiles_x; }
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

  // ---- staging: input 16-byte quads through registers (x borders need per-element masks), weights by LDS-DMA (1 KB pieces)
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
      // The very first quad of the tensor starts one float BEFORE it (x = -1 of row 0, plane 0, channel 0): bit 16.  Only the first
      // chunk of a slice that starts at channel 0 can meet it, i.e. only the PROLOGUE's load (clamped to offset 0, shifted by one
      // element at its commit); every later chunk adds chunk_bytes to the same negative offset and is an ordinary load, so the K
      // loop's commit carries no shift code (it was 4 v_cndmask per quad for one quad of the whole tensor).
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
  const unsigned w_pair_bytes = (unsigned)ncb_total * SEG24 * 4, w_tile_bytes = (unsigned)cot * SEG24 * 4;
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
  // LDS byte address of every staged quad in buffer 0 / 1 (no address arithmetic in the K loop: its two chunk bodies name their buffer
  // at compile time, and 75 KB of buffer offset do not fit the 16-bit offset field of a DS instruction)
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
      const int pc = wave8 + 8 * i;                      // 1 KB piece of the chunk's PP x 18 KB; pieces beyond it go to the dump piece
      const bool ok = pc < C::NPIECE;
      const int pcs = ok ? pc : 0;
      const int pr = pcs / (SEG24 / 256), o = pcs % (SEG24 / 256);
      lds_void* dst = reinterpret_cast<lds_void*>((uintptr_t)(lds + buf + C::IN_ELEMS + (ok ? pc * 256 : C::W_ELEMS)));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, lane16,
                                               (int)((unsigned)(chunk * C::PP + pr) * w_pair_bytes + w_tile_bytes + (unsigned)o * 1024u), 0, 0);
    }
  };

  f32x16 acc[6];      // [xi] of this wave's eta row
#pragma unroll
  for (int x = 0; x < 6; ++x)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[x][g] = 0.f;

  const int jt = (lane & 31) % XQ, ju = (lane & 31) / XQ;
  // B base: channel half, the wave's z plane, halo row 2*(wy*YT + ju) (= output row pair's y-1), s = 4*jt
  const int b_base = (lane >> 5) * C::CS + wz * (C::HY * C::HXP) + 2 * (wy * C::YT + ju) * C::HXP + 4 * jt;

  constexpr int NS = 3 * C::PP;                        // K steps per chunk: dz x channel pair
  static_assert(NS % 2 == 0 && NS >= 4, "the fragment rings are indexed statically across the chunk loop");
  float raw[2][2][6], bfq[2][6], afq[2][6];
  auto kloop = [&](auto ehc) __attribute__((always_inline)) {
    constexpr int EH = decltype(ehc)::value;
    // y transform of eta row EH from two of the four halo rows:  c = U -+ V
    //   0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
    constexpr int rowU = (EH == 0 ? 0 : EH == 2 ? 2 : 1) * C::HXP, rowV = (EH == 0 ? 2 : EH == 1 ? 2 : EH == 2 ? 1 : 3) * C::HXP;
    // Per buffer: the lane's B base, a second B base for the 8-byte tail of row V (from ONE register the load/store optimiser fuses the
    // two tails into ds_read2_b64, which runs at half the LDS rate of two ds_read_b64), the lane's A bases (16-byte and 8-byte part).
    // Pinned, so that each stays one register and every read of the K loop is base + immediate offset.
    unsigned bB[2], bV[2], bA[2], bAh[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      bB[k] = (unsigned)(uintptr_t)(lds + k * C::LDS_FLOATS + b_base);
      bV[k] = bB[k] + 16 + rowV * 4;
      bA[k] = (unsigned)(uintptr_t)(lds + k * C::LDS_FLOATS + C::IN_ELEMS + EH * 256 + lane * 4);
      bAh[k] = (unsigned)(uintptr_t)(lds + k * C::LDS_FLOATS + C::IN_ELEMS + SEG24_HI + EH * 128 + lane * 2);
      asm volatile("" : "+v"(bB[k]), "+v"(bV[k]), "+v"(bA[k]), "+v"(bAh[k]));
    }
    auto read_raw = [&](auto kbuf, int s, float (&r)[2][6]) __attribute__((always_inline)) {
      constexpr int KB = decltype(kbuf)::value;
      const int dz = s / C::PP, pp = s % C::PP;
      const unsigned off = (unsigned)(pp * 2 * C::CS + dz * (C::HY * C::HXP)) * 4u;
      const lds_f32x4* p4 = reinterpret_cast<const lds_f32x4*>((uintptr_t)(bB[KB] + off));
      const lds_f32x2* p2 = reinterpret_cast<const lds_f32x2*>((uintptr_t)(bB[KB] + off + 16));
      const lds_f32x2* p2v = reinterpret_cast<const lds_f32x2*>((uintptr_t)(bV[KB] + off));
      const f32x4 u4 = p4[rowU / 4], v4 = p4[rowV / 4];
      const f32x2 u2 = p2[rowU / 2], v2 = p2v[0];
      r[0][0] = u4[0]; r[0][1] = u4[1]; r[0][2] = u4[2]; r[0][3] = u4[3]; r[0][4] = u2[0]; r[0][5] = u2[1];
      r[1][0] = v4[0]; r[1][1] = v4[1]; r[1][2] = v4[2]; r[1][3] = v4[3]; r[1][4] = v2[0]; r[1][5] = v2[1];
    };
    auto transform = [&](const float (&r)[2][6], float (&bf)[6]) __attribute__((always_inline)) {
      float c[6];                                        // rows combined (y transform), still raw in x: x = 4t-1 .. 4t+4
#pragma unroll
      for (int v = 0; v < 6; ++v) c[v] = EH == 1 ? r[0][v] + r[1][v] : r[0][v] - r[1][v];
      const float t0 = fmaf(-4.f, c[2], c[4]), t1 = fmaf(-4.f, c[1], c[3]);
      const float t2 = c[4] - c[2], t3 = c[3] - c[1];
      bf[0] = fmaf(4.f, c[0], fmaf(-5.f, c[2], c[4]));
      bf[1] = t0 + t1;
      bf[2] = t0 - t1;
      bf[3] = fmaf(2.f, t3, t2);
      bf[4] = fmaf(-2.f, t3, t2);
      bf[5] = fmaf(4.f, c[1], fmaf(-5.f, c[3], c[5]));
    };
    auto load_a = [&](auto kbuf, int s, float (&af)[6]) __attribute__((always_inline)) {     // this eta row's 6 fragments of the step
      constexpr int KB = decltype(kbuf)::value;
      const int dz = s / C::PP, pp = s % C::PP;
      const f32x4 lo = *reinterpret_cast<const lds_f32x4*>((uintptr_t)(bA[KB] + (unsigned)(pp * SEG24 + dz * 4 * 256) * 4u));
      const f32x2 hi = *reinterpret_cast<const lds_f32x2*>((uintptr_t)(bAh[KB] + (unsigned)(pp * SEG24 + dz * 4 * 128) * 4u));
      af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3]; af[4] = hi[0]; af[5] = hi[1];
    };

    constexpr std::integral_constant<int, 0> B0{};
    constexpr std::integral_constant<int, 1> B1{};
    read_raw(B0, 0, raw[0]);
    read_raw(B0, 1, raw[1]);
    load_a(B0, 0, afq[0]);
    transform(raw[0], bfq[0]);

    // ---- K loop, software-pipelined across chunks as in conv3d_wino2e_kernel: raw rows two steps ahead, transform and weight
    // fragments one step ahead; region A = LDS reads (+ the next chunk's loads in step 0) between MFMAs 0..2, region B = pinned
    // transform (+ the input commit in step NS-2) between MFMAs 3..5; chunk barrier at the end of step NS-2.
    // Two chunk bodies, one per LDS buffer, so that the buffer is a compile-time constant in every address (27 v_add_u32 per chunk
    // and wave otherwise; beside fp32 MFMAs every VALU instruction costs matrix-pipe time).
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
        if (s == 0) { stage_w(nchk, nxt); issue_in(nchk, std::false_type{}); }
#pragma unroll
        for (int x = 0; x < 3; ++x)
          acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[s & 1][x], bfq[s & 1][x], acc[x], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         // VALU (addresses)
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);         // DS read
          if (s == 0) {
#pragma unroll
            for (int k = 0; k < (C::NI + C::NWD + 2) / 3; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);     // SALU (M0 set-up of a DMA piece)
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- region B
        if (s + 1 < NS) { pin24(raw[(s + 1) & 1]); transform(raw[(s + 1) & 1], bfq[(s + 1) & 1]); }
        if (s == NS - 1) { pin24(raw[0]); transform(raw[0], bfq[0]); }
        if (s == NS - 2) commit_in(nxt, std::false_type{});
#pragma unroll
        for (int x = 3; x < 6; ++x)
          acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[s & 1][x], bfq[s & 1][x], acc[x], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
          if (s == NS - 2) __builtin_amdgcn_sched_group_barrier(0x002, 18, 0);   // VALU (transform; masks of the input commit)
          else __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);         // DS write
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s == NS - 2) __syncthreads();                // (waits for this wave's DMA and LDS operations first)
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
  // scale / shift of the four channels this wave finishes in the epilogue (co0e + j): fetched with the first chunk - read where they
  // are used, the workgroup (alone on its CU) sits out a global-load latency after the eta exchange
  const int co0e = cot * 32 + 4 * (lane >> 5) + 8 * eta;
  float scv[4], shv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = min(co0e + j, cout - 1);
    scv[j] = ep.scale ? ep.scale[co] : 1.f;
    shv[j] = ep.shift ? ep.shift[co] : 0.f;
  }
  commit_in(std::integral_constant<int, 0>{}, std::true_type{});
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(scv[j]), "+v"(shv[j]));       // (loaded here, not sunk to the epilogue)
  __syncthreads();
  W24_STAMP(1);
  if (eta == 0) kloop(std::integral_constant<int, 0>{});
  else if (eta == 1) kloop(std::integral_constant<int, 1>{});
  else if (eta == 2) kloop(std::integral_constant<int, 2>{});
  else kloop(std::integral_constant<int, 3>{});
  W24_STAMP(2);
  __syncthreads();                                     // the exchanges below reuse the staging area

  // ---- inverse transform.  Over xi in the lane (4 output columns from 6 xi), over eta across the four waves of a position:
  //   row 0 = q0 + q1 + q2 (finished by wave eta = 1),  row 1 = q1 - q2 - q3 (finished by wave eta = 2)
  f32x16 q[4];
  {
    const f32x16 d12 = acc[1] - acc[2], s12 = acc[1] + acc[2], d34 = acc[3] - acc[4], s34 = acc[3] + acc[4];
    q[0] = acc[0] + s12 + s34;
    q[1] = d12 + 2.f * d34;
    q[2] = s12 + 4.f * s34;
    q[3] = d12 + 8.f * d34 + acc[5];
  }
  // every wave hands its q to the exchange area [pos][eta][col][g/4][lane][4]
  {
    float* xw = lds + ((size_t)(pos * 4 + eta) * 64) * 64 + 4 * lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f32x4*>(xw + (c * 4 + k) * 256) = f32x4{q[c][4 * k], q[c][4 * k + 1], q[c][4 * k + 2], q[c][4 * k + 3]};
  }
  __syncthreads();
  W24_STAMP(3);
  // every wave finishes ONE channel quarter (g = 4*eta .. 4*eta+3 of the lane's 16 channels) of BOTH output rows: the finishing work
  // (scale / shift / ReLU / pool / stores) runs on all eight waves instead of two, the fused pool has its (y, x) window in the lane
  f32x4 r0[4], r1[4];                                  // [column][channel of the quarter]
  {
    const float* xq = lds + ((size_t)(pos * 4) * 64) * 64 + eta * 256 + 4 * lane;       // slot (pos, e, c, k = eta): + e*4096 + c*1024
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(xq + 0 * 4096 + c * 1024);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(xq + 1 * 4096 + c * 1024);
      const f32x4 q2 = *reinterpret_cast<const f32x4*>(xq + 2 * 4096 + c * 1024);
      const f32x4 q3 = *reinterpret_cast<const f32x4*>(xq + 3 * 4096 + c * 1024);
      r0[c] = (q0 + q1) + q2;
      r1[c] = (q1 - q2) - q3;
    }
  }
  const int co0 = cot * 32 + 4 * (lane >> 5) + 8 * eta;   // channel of element j of the quarter: co0 + j
  const int z = z0 + wz;
  const int x = x0 + 4 * jt;
  const int y = y0 + 2 * (wy * C::YT + ju);

  if constexpr (POOL) {
    // conv + scale/shift + ReLU + MaxPool3d(2,2): the (y, x) 2x2 windows (x pairs (0,1), (2,3) of both rows) are in the lane; the z
    // pair is position 0 / 1.  First maximum in (dz, dy, dx) order (strict >), as maxpool2_fwd_kernel.
    float pooled[2][4];
    int pidx[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sc = scv[j], sh = shv[j];
#pragma unroll
      for (int hx = 0; hx < 2; ++hx) {
        float m = -INFINITY;
        int mi = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            float v = (r == 0 ? r0[2 * hx + c][j] : r1[2 * hx + c][j]) * sc + sh;
            if (ep.relu) v = fmaxf(v, 0.f);
            if constexpr (AM) {
              if (v > m) { m = v; mi = 2 * r + c; }
            } else {
              m = fmaxf(m, v);
            }
          }
        pooled[hx][j] = m;
        pidx[hx][j] = mi;
      }
    }
    float* red = lds + C::XCH_FLOATS + (size_t)eta * 8 * 64 + lane;     // behind the eta exchange area: [eta][hx*4 + j][lane]
    float* redi = red + 4 * 8 * 64;
    if (wz == 1) {
#pragma unroll
      for (int hx = 0; hx < 2; ++hx)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          red[(hx * 4 + j) * 64] = pooled[hx][j];
          if constexpr (AM) redi[(hx * 4 + j) * 64] = __int_as_float(pidx[hx][j]);
        }
    }
    __syncthreads();
    if (wz == 1) return;
    const int PD = D / 2, PH = H / 2, PW = W / 2;
    const int zp = z0 >> 1, yp = y >> 1, xp = x >> 1;
    if (zp >= PD || yp >= PH) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + j;
      if (co >= cout) continue;
      const size_t o = ((size_t)b * cout + co) * ((size_t)PD * PH * PW) + ((size_t)zp * PH + yp) * PW + xp;
      float v[2]; int id[2];
#pragma unroll
      for (int hx = 0; hx < 2; ++hx) {
        const float up = red[(hx * 4 + j) * 64];
        if constexpr (AM) {
          const bool upper = up > pooled[hx][j];        // the z + 1 plane only wins when strictly larger
          v[hx] = upper ? up : pooled[hx][j];
          id[hx] = upper ? 4 + __float_as_int(redi[(hx * 4 + j) * 64]) : pidx[hx][j];
        } else {
          v[hx] = fmaxf(pooled[hx][j], up); id[hx] = 0;
        }
      }
      if (xp + 1 < PW && (PW & 1) == 0) {
        *reinterpret_cast<f32x2*>(out + o) = f32x2{v[0], v[1]};
        if constexpr (AM) { ep.argmax[o] = (unsigned char)id[0]; ep.argmax[o + 1] = (unsigned char)id[1]; }
      } else {
        if (xp < PW) { out[o] = v[0]; if constexpr (AM) ep.argmax[o] = (unsigned char)id[0]; }
        if (xp + 1 < PW) { out[o + 1] = v[1]; if constexpr (AM) ep.argmax[o + 1] = (unsigned char)id[1]; }
      }
    }
    W24_STAMP(4); W24_STAMP(6);
    return;
  }

  if (!(z < D && y < H && x < W)) return;
  if constexpr (PREP) {
    // ---- fused `prepare` of the layer below (see PrepEpi): strip column x -> (peak, window column), then element by element
    constexpr float kEpsP = 1e-10f;                                    // peak_backprop_3d.py:29
    const int p = (int)(((float)x + 0.5f) * pe.inv_pitchA);            // exact: x < 2^22
    if (p >= pe.P) return;                                             // the A strip's tail columns
    const int cx = x - p * pe.pitchA, ix0 = cx - pe.leadA;
    const int oz = pe.origin[3 * p], oy = pe.origin[3 * p + 1], ox = pe.origin[3 * p + 2];
    if (z == 0 && y == 0 && cx == 0 && co0 == 0) { pe.origin_out[3 * p] = oz - 1; pe.origin_out[3 * p + 1] = oy - 1; pe.origin_out[3 * p + 2] = ox - 1; }
    const int iz = pe.slabA ? z - oz : z;
    if (iz < 0 || iz >= pe.U) return;
    const int qz = oz + iz;
    if (pe.slabB && (qz < 0 || qz >= pe.MD)) return;                   // B stores the map's planes only
    const int zB = pe.slabB ? qz : iz + 1;
    const float xoff = *pe.xoff;
    const long long colB = (long long)p * pe.pitchB + pe.leadB + ix0 + 1;      // B column of the quad's first element
    const bool quadB = ((colB & 3) == 0) && colB >= 0 && colB + 3 < pe.LB;
    const int MHW = pe.MH * pe.MW;
    // Branch-free: ALL 64 map reads of the lane (2 rows x 4 channels x 4 columns x 2 maps; clamped addresses where an element is not
    // valid) are issued before the first use.  With a `continue` per row / channel the compiler kept each (row, channel)'s eight loads
    // behind the previous one's stores - eight dependent memory round trips in a workgroup that is alone on its CU (measured: the
    // fused launch 1 ms slower than conv + prepare).
    bool okr[2], okc[4], okx[4];
    int rbase[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int qy = oy + y + r;
      okr[r] = (y + r < H) & (qz >= 0) & (qz < pe.MD) & (qy >= 0) & (qy < pe.MH);
      rbase[r] = okr[r] ? qz * MHW + qy * pe.MW : 0;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int ix = ix0 + c, qx = ox + ix; okx[c] = (ix >= 0) & (ix < pe.U) & (qx >= 0) & (qx < pe.MW); }
    const int MV = pe.MD * MHW;
    float xn[2][4][4], nn[2][4][4], scv2[4];
    // A quad whose four columns are all valid (or all invalid) is read as ONE 16-byte load per map (dword-aligned: the maps' rows are not
    // quad-aligned to the strip); only a wave that holds a quad CROSSING the map's x border reads element by element - with four
    // 4-byte loads per quad the epilogue was bound by the L1's instruction rate (lanes 16 bytes apart: a quarter of every line per load)
    const bool allx = okx[0] & okx[1] & okx[2] & okx[3], anyx = okx[0] | okx[1] | okx[2] | okx[3];
    const bool wave_fast = __builtin_amdgcn_ballot_w64(anyx & !allx) == 0ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      okc[j] = co0 + j < cout;
      scv2[j] = (pe.scale && okc[j]) ? pe.scale[co0 + j] : 1.f;
    }
    if (wave_fast) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cb = okc[j] ? (co0 + j) * MV : 0;                    // (host: cout * MV < 2^31)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int pos = cb + rbase[r] + ((okr[r] & allx) ? ox + ix0 : 0);
          typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
          const f32x4u a = *reinterpret_cast<const f32x4u*>(pe.xnext + pos), b = *reinterpret_cast<const f32x4u*>(pe.norm + pos);
#pragma unroll
          for (int c = 0; c < 4; ++c) { xn[r][j][c] = a[c]; nn[r][j][c] = b[c]; }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cb = okc[j] ? (co0 + j) * MV : 0;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int pos = cb + rbase[r] + ((okr[r] & okx[c]) ? ox + ix0 + c : 0);
            xn[r][j][c] = pe.xnext[pos];
            nn[r][j][c] = pe.norm[pos];
          }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float g[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = r == 0 ? r0[c][j] : r1[c][j];
          v = (xn[r][j][c] - xoff) * v;                                // PreHook of the layer just back-propagated (:16-18)
          if (!(xn[r][j][c] > 0.f)) v = 0.f;                           // ReLU backward
          if (pe.scale) v = v * scv2[j];                               // eval-BatchNorm backward
          v = (nn[r][j][c] < kEpsP) ? 0.f : v / (fabsf(nn[r][j][c]) + kEpsP);   // PostHook (:30-33)
          g[c] = (okr[r] & okx[c]) ? v : 0.f;
        }
        if (!(okc[j] & (y + r < H))) continue;
        float* dst = out + (size_t)(co0 + j) * pe.ocs + (size_t)zB * pe.ozs + (size_t)(y + r + 1) * pe.LB + colB;
        if (quadB) {
          *reinterpret_cast<f32x4*>(dst) = f32x4{g[0], g[1], g[2], g[3]};
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {                                // B window columns 0 .. U + 1 only (others belong to other cells)
            const int ix = ix0 + c;
            if (ix >= -1 && ix <= pe.U && colB + c >= 0 && colB + c < pe.LB) dst[c] = g[c];
          }
        }
      }
    }
    return;
  }
  const bool quad_ok = ((W & 3) == 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = co0 + j;
    if (co >= cout) continue;
    const float sc = scv[j], sh = shv[j];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (y + r >= H) continue;
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v[c] = (r == 0 ? r0[c][j] : r1[c][j]) * sc + sh;
        if (ep.relu) v[c] = fmaxf(v[c], 0.f);
      }
      float* o = out + ((size_t)b * cout + co) * DHW + ((size_t)z * H + y + r) * W + x;
      if (quad_ok) {
        *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (x + c < W) o[c] = v[c];
      }
    }
  }
  W24_STAMP(4); W24_STAMP(6);
}

template <int XQ, int WZ, int WY, bool POOL, bool AM, bool PREP = false>
int launch24(const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W, m3d_w2q::Epi ep, hipStream_t st,
             const m3d_w24::PrepEpi& pe = m3d_w24::PrepEpi{}) {
  using C = Cfg24<4, XQ, WZ, WY, POOL>;
  const int tiles_x = (W + C::TX - 1) / C::TX, tiles_y = (H + C::TY - 1) / C::TY, tiles_z = (D + C::TZ - 1) / C::TZ;
  const int ncb_total = ((cout + 31) / 32 + 1) / 2 * 2;
  const int co_tiles = (cout + 31) / 32;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * co_tiles;
  if (blocks > 0x7FFFFFFFll || B > 65535) return M3D_EUNSUPPORTED;
  const size_t lds = sizeof(float) * C::SMEM_FLOATS;
#ifdef M3D_W2_STAMPS
  ep.stamps = g_w24_stamps;
#endif
  auto kern = conv3d_wino24_kernel<4, XQ, WZ, WY, POOL, AM, PREP>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, B, ep.ksplit > 1 ? ep.ksplit : 1), dim3(C::NT), lds, st, in, wp, out, cin, cout, D, H, W,
                     tiles_x, tiles_y, tiles_z, ncb_total, ep, pe);
  return m3d::check_launch("conv3d_wino24");
}

}  // namespace

namespace m3d_w24 {

size_t packed_floats(int cin, int cout) {
  const size_t npair = ((cin + 1) / 2 + 15) / 16 * 16, ncb = ((cout + 31) / 32 + 1) / 2 * 2;
  return npair * ncb * (size_t)SEG24;
}

int pack(const float* d_weight, int cin, int cout, float* d_packed, hipStream_t st) {
  const int npair = ((cin + 1) / 2 + 15) / 16 * 16, ncb = ((cout + 31) / 32 + 1) / 2 * 2;
  hipLaunchKernelGGL(wino24_pack_kernel, dim3(1024), dim3(256), 0, st, d_weight, cin, cout, d_packed, ncb, npair);
  return m3d::check_launch("wino24_pack");
}

// xt = the tile id of conv3d_wino2.hip's tile choice: 32 -> 64 x 4 x 2 outputs, 16 -> 32 x 8 x 2, 8 -> 16 x 16 x 2
int launch(int xt, bool pool, bool argmax, const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W,
           m3d_w2q::Epi ep, hipStream_t st) {
  if (argmax && !pool) return M3D_EINVAL;
  if (xt == 32) {
    if (!pool) return launch24<16, 2, 1, false, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    if (!argmax) return launch24<16, 2, 1, true, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    return launch24<16, 2, 1, true, true>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  }
  if (xt == 16) {
    if (!pool) return launch24<8, 2, 1, false, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    if (!argmax) return launch24<8, 2, 1, true, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    return launch24<8, 2, 1, true, true>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  }
  if (xt == 8 && !pool) return launch24<4, 2, 1, false, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  return M3D_EUNSUPPORTED;
}

int launch_prep(int xt, const float* in, const float* wp, float* out, int cin, int cout, int D, int H, int W, m3d_w2q::Epi ep, const PrepEpi& pe,
                hipStream_t st) {
  if (ep.ksplit > 1 || ep.scale || ep.shift || ep.relu) return M3D_EINVAL;       // the fused epilogue is the whole epilogue
  if (xt == 32) return launch24<16, 2, 1, false, false, true>(in, wp, out, 1, cin, cout, D, H, W, ep, st, pe);
  if (xt == 16) return launch24<8, 2, 1, false, false, true>(in, wp, out, 1, cin, cout, D, H, W, ep, st, pe);
  if (xt == 8) return launch24<4, 2, 1, false, false, true>(in, wp, out, 1, cin, cout, D, H, W, ep, st, pe);
  return M3D_EUNSUPPORTED;
}

}  // namespace m3d_w24
