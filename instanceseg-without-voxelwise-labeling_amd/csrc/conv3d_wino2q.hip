// 3x3x3 forward convolution with Winograd F(2x2, 3x3) on the (y, x) plane - the "quad" kernel: FOUR waves per workgroup, TWO
// workgroups per CU.  Same arithmetic, weight pack and epilogues as conv3d_wino2.hip's eta-split kernel (conv + eval-BN + ReLU
// [+ MaxPool3d(2,2) [+ arg-max]] of lib/modeling/DSN.py:58-67), different decomposition:
//
//   * a workgroup = 2 tile positions x 2 eta halves (one wave per SIMD), 32 output channels x 256 outputs; each wave holds the 8
//     accumulator blocks of its eta half (128 registers), so a SECOND, independent workgroup fits on the same CU.  Round 2's 8-wave
//     workgroup owned its CU alone: its prologue / eta exchange / epilogue (8.8 % of a workgroup's cycles), its chunk barrier and
//     the lock-step of its two waves per SIMD (both in their VALU-heavy or staging phase at the same time) left the matrix pipe
//     idle 25 % of the K loop (profiles/r03_w2_ablation.txt).  Two workgroups that drift freely cover each other.
//   * LDS per workgroup must stay under 80 KB: the K loop walks HALF-chunks (one input-channel pair: 2 x halo tile + its 48
//     transformed-weight slots = 21 KB) through a ring of three slots - one being read, one ready, one being filled - with one
//     barrier per half-chunk (3 K steps).
//   * every K step is two scheduling regions: A = LDS reads (raw rows two steps ahead, weight fragments one step ahead) + global
//     loads between MFMAs 0..3; B = pin + transform of the rows read a step earlier + staging commits between MFMAs 4..7.
//     The K loop is instantiated per eta half (row offsets and signs are immediates; one address register per step).
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "conv3d_wino2q.h"

#ifndef M3D_QEXP
#define M3D_QEXP 0     // timing-only ablations: 1 = no staging, 2 = no barriers
#endif

#ifdef M3D_W2_STAMPS
static unsigned long long* g_w2q_stamps = nullptr;
M3D_API void m3d_debug_set_stamp_buffer_q(void* p) { g_w2q_stamps = (unsigned long long*)p; }
#define W2Q_STAMP(k) do { if (ep.stamps && tid == 0) ep.stamps[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 8 + (k)] = \
    (k) == 5 || (k) == 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W2Q_STAMP(k) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_cfloat;

constexpr int WT2 = 48;   // weight slots per (cout, cin): 3 dz x 4 eta x 4 xi (conv3d_wino2.hip's pack)

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

__device__ __forceinline__ int xcd_contiguous_q(int bid, int n) {
  const int per = n >> 3, rem = n & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// XT x-pairs and YT = 32/XT y-pairs per wave block; WZ x WY = 2 tile positions per workgroup.
template <int XT, int WZ, int WY, bool POOL>
struct QCfg {
  static constexpr int NT = 256;
  static constexpr int YT = 32 / XT;
  static constexpr int TX = 2 * XT, TY = 2 * YT * WY, TZ = WZ;
  static constexpr int EP = XT + 2;
  static constexpr int QR = EP / 2;                            // 16-byte quads per row
  // row pitch: the YT y-pairs of a block sit 2*HXP floats apart; 2*HXP = XT (mod 32) puts them on disjoint banks
  static constexpr int HXP = (YT == 1) ? 2 * EP : ((2 * EP - XT / 2 + 15) / 16 * 16 + XT / 2);
  static constexpr int HY = TY + 2, HZ = TZ + 2;
  static constexpr int CS = HXP * HY * HZ;                     // one channel's halo tile
  static constexpr int IN_H = 2 * CS;                          // a half-chunk: one channel pair
  static constexpr int NQUAD = 2 * HZ * HY * QR;
  static constexpr int W_H = WT2 * 64;                         // the pair's 48 slots of one cout block
  static constexpr int NI = (NQUAD + NT - 1) / NT;
  static constexpr int NW4 = W_H / 4 / NT;
  static constexpr int DUMP = IN_H + W_H;                      // 2 x 8-byte dump slots behind each ring slot (branch-free staging)
  static constexpr int SLOT = IN_H + W_H + ((EP + 2 + 3) / 4) * 4;
  static constexpr int XCH_FLOATS = 2 * 64 * 64;               // eta-half exchange: 2 wave pairs x 64 floats x 64 lanes
  static constexpr int RED_FLOATS = XCH_FLOATS + (POOL ? 2 * 16 * 64 : 0);
  static constexpr int SMEM_FLOATS = 3 * SLOT > RED_FLOATS ? 3 * SLOT : RED_FLOATS;
  static_assert(WZ * WY == 2, "2 tile positions per workgroup");
  static_assert(!POOL || (WZ == 2 && WY == 1), "fused pool: the z pair lives in the two positions");
  static_assert(HXP % 2 == 0, "8-byte LDS stores");
  static_assert(W_H / 4 % NT == 0, "weight staging is branch-free: whole float4 rounds");
  static_assert(3 * HXP + EP + 1 < 256, "the six row reads of a step must reach from one address register (8-bit dword offsets)");
  static_assert(SMEM_FLOATS * 4 <= 80 * 1024, "two workgroups per CU");
};

template <int XT, int WZ, int WY, bool POOL, bool AM>
__global__ __launch_bounds__(256, 2) void conv3d_wino2q_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                              float* __restrict__ out, int cin, int cout, int D, int H, int W,
                                                              int tiles_x, int tiles_y, int tiles_z, int ncb_total, m3d_w2q::Epi ep) {
  using C = QCfg<XT, WZ, WY, POOL>;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave4 = tid >> 6;
  W2Q_STAMP(0); W2Q_STAMP(5);
  const int eh = __builtin_amdgcn_readfirstlane(wave4 >> 1), pos = wave4 & 1;     // eta half, tile position
  const int wz = pos / WY, wy = pos % WY;

  int bid = blockIdx.x;
  const int co_tiles = (cout + 31) / 32;
  // XCD-contiguous order with the cout tile FASTEST (the workgroups that share an input tile sit next to each other on one XCD),
  // z tiles in groups of 4 inside the y sweep (a compact (y, z) block of tiles per XCD at any time: halo planes stay in its L2)
  if (ep.xcd_map) bid = xcd_contiguous_q(bid, gridDim.x);
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int tx = bid % tiles_x; bid /= tiles_x;
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

  // ---- input staging descriptors: 16-byte quads of the de-interleaved halo rows (E[u] = in[x0+2u], O[u] = in[x0+2u-1])
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
  f32x4 stg[C::NI], stgw[C::NW4];
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(in_b), 0, (unsigned)((size_t)cin * DHW * sizeof(float)), 0x00020000);
  // half-chunks of this workgroup: [h_begin, h_end), an even count (a pair beyond cin reads zeros: buffer range check / zero-padded pack)
  const int nh_all = ((cin + 3) / 4) * 2;
  const int h_begin = ep.ksplit > 1 ? (int)blockIdx.z * ep.cps * 2 : 0;
  const int h_end = ep.ksplit > 1 ? min(nh_all, h_begin + ep.cps * 2) : nh_all;
  if (ep.ksplit > 1) out += (size_t)blockIdx.z * ep.slice_stride;
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  const size_t w_pair_stride4 = (size_t)ncb_total * WT2 * 64 / 4;
  const size_t w_tile_off4 = (size_t)cot * WT2 * 64 / 4;
  const int pair_bytes = (int)(2 * DHW * sizeof(float));
  auto issue_in = [&](int h) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::NI; ++i)
      stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, gq[i], h * pair_bytes, 0));
  };
  auto issue_w = [&](int h) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::NW4; ++i) stgw[i] = (wp4 + (size_t)h * w_pair_stride4 + w_tile_off4)[tid + i * C::NT];
  };
  auto commit_in = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {                    // branch-free: the K loop must stay one scheduling region
      const int m = mq[i];
      const f32x4 v = stg[i];
      const bool sh = (m & 16) != 0;
      const float v0 = sh ? 0.f : v[0], v1 = sh ? v[0] : v[1], v2 = sh ? v[1] : v[2], v3 = sh ? v[2] : v[3];
      const f32x2 ev = {(m & 2) ? v1 : 0.f, (m & 8) ? v3 : 0.f};
      const f32x2 ov = {(m & 1) ? v0 : 0.f, (m & 4) ? v2 : 0.f};
      *reinterpret_cast<f32x2*>(dst + lq[i]) = ev;
      *reinterpret_cast<f32x2*>(dst + lq[i] + C::EP) = ov;
    }
  };
  auto commit_w = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::NW4; ++i) reinterpret_cast<f32x4*>(dst + C::IN_H)[tid + i * C::NT] = stgw[i];
  };

  f32x16 acc[2][4];   // [eta - 2 * eh][xi]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[a][x][g] = 0.f;

  const int jt = (lane & 31) % XT, ju = (lane & 31) / XT;
  // B base: channel of the pair, the wave's z plane, halo row 2*(wy*YT + ju) (= output row pair's y-1), E[jt]
  const int b_base = (lane >> 5) * C::CS + wz * (C::HY * C::HXP) + 2 * (wy * C::YT + ju) * C::HXP + jt;

  // ---- prologue: half-chunks h_begin, h_begin + 1 -> ring slots 0, 1
  issue_w(h_begin); issue_in(h_begin);
  commit_w(lds); commit_in(lds);
  issue_w(h_begin + 1); issue_in(h_begin + 1);
  commit_w(lds + C::SLOT); commit_in(lds + C::SLOT);
  __syncthreads();
  W2Q_STAMP(1);

  float raw[2][3][4], bfq[2][2][4], afq[2][8];
  auto kloop = [&](auto ehc) __attribute__((always_inline)) {
    constexpr int EH = decltype(ehc)::value;
    // y transform of this eta half from three of the four halo rows:  cA = U - V,  cB = V +- P
    //   EH = 0 (eta 0, 1): U = row 0, V = row 2, P = row 1, cB = V + P   (d0 - d2, d1 + d2)
    //   EH = 1 (eta 2, 3): U = row 2, V = row 1, P = row 3, cB = V - P   (d2 - d1, d1 - d3)
    constexpr int rowU = (EH ? 2 : 0) * C::HXP, rowV = (EH ? 1 : 2) * C::HXP, rowP = (EH ? 3 : 1) * C::HXP;
    auto read_raw = [&](const float* slot, int dz, float (&r)[3][4]) __attribute__((always_inline)) {
      unsigned a = (unsigned)(uintptr_t)(slot + b_base + dz * (C::HY * C::HXP));
      asm volatile("" : "+v"(a));                        // ONE address register per step: the reads below use their 8-bit offsets
      const lds_cfloat* p = reinterpret_cast<const lds_cfloat*>((uintptr_t)a);
      const lds_cfloat* pu = p + rowU; const lds_cfloat* pv = p + rowV; const lds_cfloat* pq = p + rowP;
      r[0][0] = pu[0]; r[0][1] = pu[1]; r[0][2] = pu[C::EP]; r[0][3] = pu[C::EP + 1];      // (E[t], E[t+1], O[t], O[t+1])
      r[1][0] = pv[0]; r[1][1] = pv[1]; r[1][2] = pv[C::EP]; r[1][3] = pv[C::EP + 1];
      r[2][0] = pq[0]; r[2][1] = pq[1]; r[2][2] = pq[C::EP]; r[2][3] = pq[C::EP + 1];
    };
    auto transform = [&](const float (&r)[3][4], float (&bf)[2][4]) __attribute__((always_inline)) {
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
    auto load_a = [&](const float* slot, int dz, float (&af)[8]) __attribute__((always_inline)) {     // this half's 8 slots of the step
#pragma unroll
      for (int q = 0; q < 8; ++q) af[q] = slot[C::IN_H + m3d_w2q::w2_slot(dz, EH * 2 + (q >> 2), q & 3, 0) + 4 * lane];
    };

    read_raw(lds, 0, raw[0]);
    read_raw(lds, 1, raw[1]);
    load_a(lds, 0, afq[0]);
    transform(raw[0], bfq[0]);

    // ---- K loop: one iteration = two half-chunks (A, B) = six K steps; C = the half-chunk after B, D the one after C (D goes into
    // A's slot).  Step j works on (j < 3 ? A : B, dz = j % 3); its region A reads the raw rows of step j + 2 and the weight fragments of
    // step j + 1.  Barrier after step 0 publishes B (committed in steps 5', 0 of ... see below) and frees the slot before A's for C;
    // barrier after step 3 publishes C and frees A's slot for D:
    //   C: global loads in step 0, LDS commit in step 2 (after the step-0 barrier: the slot's last reads were step 4 of the iteration
    //      before), published by the step-3 barrier, first read in step 4
    //   D: global loads in step 3, LDS commit in step 5 (after the step-3 barrier: A's last read is step 1), published by the next
    //      iteration's step-0 barrier, first read there in step 1 (as its B)
    int oA = 0, oB = C::SLOT, oC = 2 * C::SLOT;
    for (int h = h_begin; h < h_end; h += 2) {
      const float* sA = lds + oA; const float* sB = lds + oB; float* sC = lds + oC;
      const int hC = min(h + 2, h_end - 1), hD = min(h + 3, h_end - 1);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        // ---------------- region A
        {
          const int jr = j + 2;                            // raw rows of step j + 2
          const float* s = jr < 3 ? sA : (jr < 6 ? sB : sC);
          read_raw(s, jr % 3, raw[j & 1]);
          const int ja = j + 1;                            // weight fragments of step j + 1
          const float* t = ja < 3 ? sA : (ja < 6 ? sB : sC);
          load_a(t, ja % 3, afq[(j + 1) & 1]);
        }
#if !(M3D_QEXP & 1)
        if (j == 0) { issue_w(hC); issue_in(hC); }
        if (j == 3) { issue_w(hD); issue_in(hD); }
#endif
#pragma unroll
        for (int x = 0; x < 4; ++x)
          acc[0][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[j & 1][x], bfq[j & 1][0][x], acc[0][x], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);         // DS read
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         // VALU (addresses)
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);         // VMEM read
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- region B
        pin_regs(raw[(j + 1) & 1]);
        transform(raw[(j + 1) & 1], bfq[(j + 1) & 1]);
#if !(M3D_QEXP & 1)
        if (j == 2) { commit_w(sC); commit_in(sC); }
        if (j == 5) { commit_w(const_cast<float*>(sA)); commit_in(const_cast<float*>(sA)); }
#endif
#pragma unroll
        for (int x = 0; x < 4; ++x)
          acc[1][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[j & 1][4 + x], bfq[j & 1][1][x], acc[1][x], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
          if (j == 2 || j == 5) __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);   // VALU (transform; masks of the input commit)
          else __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);         // DS write
        }
        __builtin_amdgcn_sched_barrier(0);
#if !(M3D_QEXP & 2)
        if (j == 0 || j == 3) __syncthreads();
#endif
      }
      const int t = oA; oA = oC; oC = oB; oB = t;          // (A, B, C) <- (C, A's slot now holding D, B's slot)
    }
  };
  if (eh) kloop(std::integral_constant<int, 1>{}); else kloop(std::integral_constant<int, 0>{});
  W2Q_STAMP(2);
  __syncthreads();                                     // the exchanges below reuse the ring

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
    float* xch = lds + (size_t)pos * 64 * 64 + lane;
    if (eh == 1) {
      const f32x16 a0 = p0[0], a1 = -p0[0] - p0[1], b0 = p1[0], b1 = -p1[0] - p1[1];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        xch[g * 64] = a0[g]; xch[(16 + g) * 64] = a1[g]; xch[(32 + g) * 64] = b0[g]; xch[(48 + g) * 64] = b1[g];
      }
    }
    __syncthreads();
    W2Q_STAMP(3);
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
    // conv + scale/shift + ReLU + MaxPool3d(2,2): the (y, x) 2x2 footprint is in the lane; the z pair is position 0 / 1
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
    float* red = lds + C::XCH_FLOATS + lane;           // behind the eta exchange area
    float* redi = red + 16 * 64;
    if (wz == 1) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        red[g * 64] = pooled[g];
        if constexpr (AM) redi[g * 64] = __int_as_float(pidx[g]);
      }
    }
    __syncthreads();                                   // only the two eta-half-0 waves are left (ended waves no longer count)
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
    W2Q_STAMP(4); W2Q_STAMP(6);
    return;
  }

  if (z < D && y < H && x < W) {
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
  W2Q_STAMP(4); W2Q_STAMP(6);
}

template <int XT, int WZ, int WY, bool POOL, bool AM>
int launch_q(const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W, m3d_w2q::Epi ep,
             hipStream_t st) {
  using C = QCfg<XT, WZ, WY, POOL>;
  const int tiles_x = (W + C::TX - 1) / C::TX, tiles_y = (H + C::TY - 1) / C::TY, tiles_z = (D + C::TZ - 1) / C::TZ;
  const int ncb_total = ((cout + 31) / 32 + 1) / 2 * 2;
  const int co_tiles = (cout + 31) / 32;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * co_tiles;
  if (blocks > 0x7FFFFFFFll || B > 65535) return M3D_EUNSUPPORTED;
  const size_t lds = sizeof(float) * C::SMEM_FLOATS;
  auto kern = conv3d_wino2q_kernel<XT, WZ, WY, POOL, AM>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#ifdef M3D_W2_STAMPS
  ep.stamps = g_w2q_stamps;
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, B, ep.ksplit > 1 ? ep.ksplit : 1), dim3(C::NT), lds, st, in, wp, out, cin, cout, D, H, W,
                     tiles_x, tiles_y, tiles_z, ncb_total, ep);
  return m3d::check_launch("conv3d_wino2q");
}

}  // namespace

namespace m3d_w2q {

int launch(int xt, bool pool, bool argmax, const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W,
           Epi ep, hipStream_t st) {
  if (argmax && !pool) return M3D_EINVAL;
  if (xt == 32) {
    if (!pool) return launch_q<32, 2, 1, false, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    if (!argmax) return launch_q<32, 2, 1, true, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    return launch_q<32, 2, 1, true, true>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  }
  if (xt == 16) {
    if (!pool) return launch_q<16, 2, 1, false, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    if (!argmax) return launch_q<16, 2, 1, true, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    return launch_q<16, 2, 1, true, true>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  }
  if (xt == 8 && !pool) return launch_q<8, 2, 1, false, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  return M3D_EUNSUPPORTED;
}

void tile_dims(int xt, int* tx, int* ty, int* tz) {
  *tx = 2 * xt; *ty = 2 * (32 / xt); *tz = 2;
}

}  // namespace m3d_w2q
