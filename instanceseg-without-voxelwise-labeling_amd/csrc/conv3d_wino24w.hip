// Winograd F(2x4, 3x3) forward convolution, "wide" decomposition (family 5): the arithmetic, weight pack and epilogues of
// conv3d_wino24.hip, cut differently.  Round 3 measured that beside fp32 MFMAs every other instruction costs matrix-pipe time
// ("Round 3"), and that kernel's K loop carries 7.4 of them per MFMA, half of them the input transform.  Here a wave
// feeds each transformed B fragment to TWO output-channel blocks:
//   workgroup = 4 waves (one per SIMD, up to 512 registers each: the 12 accumulator blocks = 192 registers live in AGPRs),
//   wave = one eta row x 64 output channels x 32 patches (2 x 4 outputs each) of ONE tile = 4*XQ x 2*YT x 2 outputs, 32 = XQ x YT x 2 z;
//   per K step (one channel pair, one dz): two halo rows read (16 + 8 bytes each), 18 transform VALU, 2 x (16 + 8) bytes of weight
//   fragments, 12 MFMAs - 3.3 other instructions per MFMA in the whole loop instead of 7.4.
// Chunks are ONE channel pair (the two blocks' weights are 36 KB per pair; two pairs double-buffered would not fit beside the input).
// The z pair of the fused pool sits in lanes l and l ^ (XQ*YT): it meets through a lane permute, no LDS and no barrier.
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "conv3d_wino24w.h"

#ifdef M3D_W2_STAMPS
static unsigned long long* g_w24w_stamps = nullptr;
M3D_API void m3d_debug_set_stamp_buffer_24w(void* p) { g_w24w_stamps = (unsigned long long*)p; }
#define W24W_STAMP(k) do { if (ep.stamps && tid == 0) ep.stamps[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 8 + (k)] = \
    (k) == 5 || (k) == 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W24W_STAMP(k) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;

constexpr int SEG24 = 72 * 64;                   // floats of one (channel pair, cout block) of conv3d_wino24.hip's pack
constexpr int SEG24_HI = 12 * 64 * 4;            // offset of the xi 4, 5 part

__device__ __forceinline__ int xcd_contiguous_w(int bid, int n) {
  const int per = n >> 3, rem = n & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

template <int A, int B>
__device__ __forceinline__ void pin_w(float (&r)[A][B]) {
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b < B; ++b) asm volatile("" : "+v"(r[a][b]));
}

template <int XQ, bool POOL>
struct CfgW {
  static constexpr int NT = 256;
  static constexpr int YT = 16 / XQ;                           // y pairs per plane in a wave's 32 patches
  static constexpr int TX = 4 * XQ, TY = 2 * YT, TZ = 2;
  static constexpr int ROW = 4 * XQ + 2;
  static constexpr int QR = (ROW + 3) / 4;
  static constexpr int HXP = XQ == 16 ? 96 : (XQ == 8 ? 48 : 24);   // row pitch: bank rules as conv3d_wino24.hip (16-lane groups)
  static constexpr int HY = TY + 2, HZ = TZ + 2;
  static constexpr int CS = HXP * HY * HZ;                     // floats per channel
  static constexpr int NQUAD = 2 * HZ * HY * QR;               // one channel pair per chunk
  static constexpr int NI = (NQUAD + NT - 1) / NT;
  static constexpr int IN_ELEMS = 2 * CS + 4;                  // + a 16-byte dump slot
  static constexpr int DUMP = 2 * CS;
  static constexpr int W_ELEMS = 2 * SEG24;                    // two cout blocks of the pair: 36 KB, one contiguous run of the pack
  static constexpr int NWD = W_ELEMS / 256 / 4;                // 1 KB LDS-DMA pieces per wave: 9
  // LDS: two input buffers, THREE weight buffers.  With one wave per SIMD nothing covers a wait, so every global access needs a whole
  // chunk (36 MFMAs = 2300 cycles) of lead: the weights of chunk c + 2 are DMA'd during chunk c (into the buffer chunk c - 1 read), the
  // input quads of chunk c + 2 are loaded into registers at the end of chunk c and committed in the middle of chunk c + 1.
  static constexpr int W_OFF = 2 * IN_ELEMS;
  static constexpr int STAGE_FLOATS = 2 * IN_ELEMS + 3 * W_ELEMS;
  static constexpr int XCH_FLOATS = 2 * 4 * 64 * 64;           // eta exchange: 2 cout blocks x 4 eta rows x 64 floats x 64 lanes
  static constexpr int SMEM_FLOATS = STAGE_FLOATS > XCH_FLOATS ? STAGE_FLOATS : XCH_FLOATS;
  static_assert(W_ELEMS % (256 * 4) == 0, "whole DMA pieces per wave");
  static_assert(ROW <= HXP && HXP % 4 == 0, "rows are whole 16-byte quads");
  static_assert(SMEM_FLOATS * 4 <= 160 * 1024, "LDS");
};

template <int XQ, bool POOL, bool AM>
__global__ __launch_bounds__(256, 1) void conv3d_wino24w_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                               float* __restrict__ out, int cin, int cout, int D, int H, int W,
                                                               int tiles_x, int tiles_y, int tiles_z, int ncb_total, m3d_w2q::Epi ep) {
  using C = CfgW<XQ, POOL>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  W24W_STAMP(0); W24W_STAMP(5);
  const int eta = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bid = blockIdx.x;
  const int co_tiles = (cout + 63) / 64;
  if (ep.xcd_map) bid = xcd_contiguous_w(bid, gridDim.x);
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

  // ---- staging: input 16-byte quads through registers (x borders need per-element masks), weights by LDS-DMA (1 KB pieces)
  int gq[C::NI], mq[C::NI];
  unsigned lqa[2][C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int e = tid + i * C::NT;
    gq[i] = 0; mq[i] = 0;
    int l = C::DUMP;                                  // quads beyond the tile: masked to zero, written to a dump slot
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
      if (rok && lin < 0) m |= 16;                    // the first quad of the tensor: met by the prologue's load only (conv3d_wino24.hip)
      mq[i] = m;
      gq[i] = rok ? (int)(lin * 4) : 0;
      l = row * C::HXP + 4 * q;
    }
    lqa[0][i] = (unsigned)(uintptr_t)(lds + l);
    lqa[1][i] = (unsigned)(uintptr_t)(lds + C::IN_ELEMS + l);
  }
  f32x4 stg[C::NI];
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(in_b), 0, (unsigned)((size_t)cin * DHW * sizeof(float)), 0x00020000);
  // chunks are channel PAIRS here; the split-K plan (ep.cps) counts the 4-channel chunks of the other families
  const int nchunk_all = (cin + 1) / 2;
  const int c_begin = ep.ksplit > 1 ? (int)blockIdx.z * ep.cps * 2 : 0;
  const int nchunk = ep.ksplit > 1 ? min(nchunk_all, c_begin + ep.cps * 2) : nchunk_all;
  if (ep.ksplit > 1) out += (size_t)blockIdx.z * ep.slice_stride;
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned w_pair_bytes = (unsigned)ncb_total * SEG24 * 4, w_tile_bytes = (unsigned)(2 * cot) * SEG24 * 4;
  const int lane16 = lane * 16;
  const int chunk_bytes = (int)(2 * DHW * sizeof(float));
  auto issue_in = [&](int chunk, auto first) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      int off = gq[i] + chunk * chunk_bytes;
      if constexpr (decltype(first)::value) off = max(off, 0);
      stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));
    }
  };
  auto commit_in = [&](auto kbuf, auto first) __attribute__((always_inline)) {
    constexpr int KB = decltype(kbuf)::value;
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      const int m = mq[i];
      const f32x4 v = stg[i];
      float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
      if constexpr (decltype(first)::value) {
        const bool sh = (m & 16) != 0 && gq[i] + c_begin * chunk_bytes < 0;
        v3 = sh ? v2 : v3; v2 = sh ? v1 : v2; v1 = sh ? v0 : v1; v0 = sh ? 0.f : v0;
      }
      f32x4 o = {(m & 1) ? v0 : 0.f, (m & 2) ? v1 : 0.f, (m & 4) ? v2 : 0.f, (m & 8) ? v3 : 0.f};
      asm volatile("" : "+v"(o));                        // one register tuple -> ds_write_b128 (a split store conflicts 4-way)
      *reinterpret_cast<lds_f32x4*>((uintptr_t)lqa[KB][i]) = o;
    }
  };
  auto stage_w = [&](int chunk, int wbuf) __attribute__((always_inline)) {       // wbuf: 0..2 (wave-uniform)
#pragma unroll
    for (int i = 0; i < C::NWD; ++i) {
      const int pc = eta + 4 * i;                        // 1 KB piece of the pair's 36 KB (both cout blocks, contiguous in the pack)
      lds_void* dst = reinterpret_cast<lds_void*>((uintptr_t)(lds + C::W_OFF + wbuf * C::W_ELEMS + pc * 256));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, lane16,
                                               (int)((unsigned)chunk * w_pair_bytes + w_tile_bytes + (unsigned)pc * 1024u), 0, 0);
    }
  };

  f32x16 acc[2][6];   // [cout block][xi] of this wave's eta row
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[c][x][g] = 0.f;

  const int pl = lane & 31;
  const int jt = pl % XQ, ju = (pl / XQ) % C::YT, jz = pl / (XQ * C::YT);
  // B base: channel of the pair, the patch's z plane, halo row 2*ju (= the output row pair's y - 1), s = 4*jt
  const int b_base = (lane >> 5) * C::CS + jz * (C::HY * C::HXP) + 2 * ju * C::HXP + 4 * jt;

  constexpr int NS = 3;                                // K steps per chunk: dz
  float raw[2][2][6], bfq[2][6], afq[2][12];
  auto kloop = [&](auto ehc) __attribute__((always_inline)) {
    constexpr int EH = decltype(ehc)::value;
    // y transform of eta row EH from two of the four halo rows:  c = U -+ V   (0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3)
    constexpr int rowU = (EH == 0 ? 0 : EH == 2 ? 2 : 1) * C::HXP, rowV = (EH == 0 ? 2 : EH == 1 ? 2 : EH == 2 ? 1 : 3) * C::HXP;
    // pinned per-buffer bases: every LDS read of the loop = base + immediate.  The 8-byte reads that sit a constant apart (row tails of
    // U and V; the xi 4, 5 fragments of the two cout blocks) get a base register EACH: from one register the load/store optimiser
    // fuses them into ds_read2(st64)_b64, which runs at half the LDS rate of two ds_read_b64
    unsigned bB[2], bV[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      bB[k] = (unsigned)(uintptr_t)(lds + k * C::IN_ELEMS + b_base);
      bV[k] = bB[k] + 16 + rowV * 4;
      asm volatile("" : "+v"(bB[k]), "+v"(bV[k]));
    }
    // A bases of weight buffer 0; the current / next chunk's buffers (index mod 3, a run-time value) add wbuf * W_ELEMS: three v_add per chunk
    const unsigned a0 = (unsigned)(uintptr_t)(lds + C::W_OFF + EH * 256 + lane * 4);
    const unsigned ah0 = (unsigned)(uintptr_t)(lds + C::W_OFF + SEG24_HI + EH * 128 + lane * 2);
    unsigned bA[2], bAh[2][2];                         // [0] = current chunk's weight buffer, [1] = next chunk's
    auto set_a = [&](int which, int wbuf) __attribute__((always_inline)) {
      bA[which] = a0 + (unsigned)(wbuf * C::W_ELEMS * 4);
      bAh[which][0] = ah0 + (unsigned)(wbuf * C::W_ELEMS * 4);
      bAh[which][1] = bAh[which][0] + SEG24 * 4;
      asm volatile("" : "+v"(bA[which]), "+v"(bAh[which][0]), "+v"(bAh[which][1]));
    };
    auto read_raw = [&](auto kbuf, int dz, float (&r)[2][6]) __attribute__((always_inline)) {
      constexpr int KB = decltype(kbuf)::value;
      const unsigned off = (unsigned)(dz * (C::HY * C::HXP)) * 4u;
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
    auto load_a = [&](auto which, int dz, float (&af)[12]) __attribute__((always_inline)) {    // both cout blocks' 6 fragments of the step
      constexpr int KB = decltype(which)::value;        // 0: the current chunk's weight buffer, 1: the next chunk's
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const f32x4 lo = *reinterpret_cast<const lds_f32x4*>((uintptr_t)(bA[KB] + (unsigned)(cb * SEG24 + dz * 4 * 256) * 4u));
        const f32x2 hi = *reinterpret_cast<const lds_f32x2*>((uintptr_t)(bAh[KB][cb] + (unsigned)(dz * 4 * 128) * 4u));
        af[cb * 6 + 0] = lo[0]; af[cb * 6 + 1] = lo[1]; af[cb * 6 + 2] = lo[2]; af[cb * 6 + 3] = lo[3];
        af[cb * 6 + 4] = hi[0]; af[cb * 6 + 5] = hi[1];
      }
    };
    constexpr std::integral_constant<int, 0> B0{};
    constexpr std::integral_constant<int, 1> B1{};
    int wb = 0;                                        // weight buffer of the current chunk (chunk - c_begin) mod 3
    set_a(0, 0); set_a(1, 1);
    read_raw(B0, 0, raw[0]);
    read_raw(B0, 1, raw[1]);
    load_a(B0, 0, afq[0]);
    transform(raw[0], bfq[0]);

    // ---- K loop.  Global step g = 3 * chunk + s; rings (raw, B, A fragments) are indexed by g & 1 = (KB + s) & 1 in the body of buffer
    // KB (the bodies alternate).  Raw rows are read two steps ahead, transformed one step ahead, weight fragments read one step ahead;
    // step 0 issues the next chunk's loads, step 1 commits its input and ends with the chunk barrier, step 2 reads the next chunk's
    // first fragments.  Region A (cout block 0's MFMAs) carries the LDS reads, region B (block 1's) the pinned transform.
    auto chunk_body = [&](auto curc, int chunk) __attribute__((always_inline)) {
      constexpr int KB = decltype(curc)::value;
      constexpr std::integral_constant<int, KB> cur{};
      constexpr std::integral_constant<int, 1 - KB> nxt{};
      const int far = min(chunk + 2, nchunk - 1);          // the chunk whose weights / input start their way now
      const int wb2 = wb == 0 ? 2 : wb - 1;                // (wb + 2) mod 3: the buffer chunk - 1 read, free since its barrier
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int r = (KB + s) & 1, rn = r ^ 1;
        // ---------------- region A
        if (s == 0) read_raw(cur, 2, raw[r]);              // raw of step g + 2 (same parity as g)
        if (s + 1 < NS) load_a(B0, s + 1, afq[rn]);
        if (s == NS - 1) {                                 // next chunk's first fragments (its buffers are complete: barrier of step 1)
          read_raw(nxt, 0, raw[rn]);
          read_raw(nxt, 1, raw[r]);
          load_a(B1, 0, afq[rn]);
        }
        if (s == 0) stage_w(far, wb2);                     // 9 LDS-DMA pieces: two chunks of lead
        if (s == NS - 1) issue_in(far, std::false_type{}); // the registers were committed in step 1: one chunk of lead
#pragma unroll
        for (int x = 0; x < 6; ++x)
          acc[0][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[r][x], bfq[r][x], acc[0][x], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);         // DS read
          if (s == 0 || s == NS - 1) {
#pragma unroll
            for (int k = 0; k < (C::NWD + 5) / 6; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);     // SALU (M0 set-up of a DMA piece)
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- region B
        if (s + 1 < NS) { pin_w(raw[rn]); transform(raw[rn], bfq[rn]); }
        if (s == NS - 1) { pin_w(raw[rn]); transform(raw[rn], bfq[rn]); }
        if (s == NS - 2) commit_in(nxt, std::false_type{});
#pragma unroll
        for (int x = 0; x < 6; ++x)
          acc[1][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[r][6 + x], bfq[r][x], acc[1][x], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
          if (s == NS - 2) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);    // VALU (transform; masks of the input commit)
          else __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);         // DS write
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s == NS - 2) {
          // chunk barrier: the next chunk's input is committed (LDS writes: lgkmcnt) and its weights have landed - they were DMA'd a
          // chunk ago, the only vector-memory operations issued since are this chunk's NWD pieces, and those return in order behind them
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(C::NWD) : "memory");
        }
      }
      wb = wb == 2 ? 0 : wb + 1;
      bA[0] = bA[1]; bAh[0][0] = bAh[1][0]; bAh[0][1] = bAh[1][1];
      set_a(1, wb == 2 ? 0 : wb + 1);
    };
    for (int chunk = c_begin;;) {
      chunk_body(B0, chunk);
      if (++chunk >= nchunk) break;
      chunk_body(B1, chunk);
      if (++chunk >= nchunk) break;
    }
  };
  // ---- prologue: weights of the first two chunks -> weight buffers 0, 1; input of the first chunk -> LDS, of the second -> registers
  stage_w(c_begin, 0); stage_w(min(c_begin + 1, nchunk - 1), 1); issue_in(c_begin, std::true_type{});
  // scale / shift of the 2 x 4 channels this wave finishes in the epilogue (co0e + 32*cb + j)
  const int co0e = cot * 64 + 4 * (lane >> 5) + 8 * eta;
  float scv[2][4], shv[2][4];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = min(co0e + 32 * cb + j, cout - 1);
      scv[cb][j] = ep.scale ? ep.scale[co] : 1.f;
      shv[cb][j] = ep.shift ? ep.shift[co] : 0.f;
    }
  commit_in(std::integral_constant<int, 0>{}, std::true_type{});
  issue_in(min(c_begin + 1, nchunk - 1), std::false_type{});
  pin_w(scv); pin_w(shv);
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(C::NI) : "memory");     // everything but the second chunk's input quads
  W24W_STAMP(1);
  if (eta == 0) kloop(std::integral_constant<int, 0>{});
  else if (eta == 1) kloop(std::integral_constant<int, 1>{});
  else if (eta == 2) kloop(std::integral_constant<int, 2>{});
  else kloop(std::integral_constant<int, 3>{});
  W24W_STAMP(2);
  __syncthreads();                                     // the exchange below reuses the staging area

  // ---- inverse transform.  Over xi in the lane (4 output columns from 6 xi); over eta across the four waves through LDS
  // [cb][eta][col][g/4][lane][4]; then every wave finishes channel quarter eta (g = 4*eta .. 4*eta+3) of both cout blocks.
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const f32x16 d12 = acc[cb][1] - acc[cb][2], s12 = acc[cb][1] + acc[cb][2], d34 = acc[cb][3] - acc[cb][4], s34 = acc[cb][3] + acc[cb][4];
    f32x16 q[4];
    q[0] = acc[cb][0] + s12 + s34;
    q[1] = d12 + 2.f * d34;
    q[2] = s12 + 4.f * s34;
    q[3] = d12 + 8.f * d34 + acc[cb][5];
    float* xw = lds + ((size_t)(cb * 4 + eta) * 64) * 64 + 4 * lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f32x4*>(xw + (c * 4 + k) * 256) = f32x4{q[c][4 * k], q[c][4 * k + 1], q[c][4 * k + 2], q[c][4 * k + 3]};
  }
  __syncthreads();
  W24W_STAMP(3);
  const int z = z0 + jz;
  const int x = x0 + 4 * jt;
  const int y = y0 + 2 * ju;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    f32x4 r0[4], r1[4];                                // [column][channel of the quarter]: output rows y, y + 1
    {
      const float* xq = lds + ((size_t)(cb * 4) * 64) * 64 + eta * 256 + 4 * lane;       // slot (cb, e, c, k = eta): + e*4096 + c*1024
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
    const int co0 = co0e + 32 * cb;
    if constexpr (POOL) {
      // conv + scale/shift + ReLU + MaxPool3d(2,2): the (y, x) 2x2 windows are in the lane, the z pair in lanes l and l ^ (XQ*YT) of the
      // same half.  First maximum in (dz, dy, dx) order (strict >), as maxpool2_fwd_kernel.
      const int PD = D / 2, PH = H / 2, PW = W / 2;
      const int zp = z0 >> 1, yp = y >> 1, xp = x >> 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sc = scv[cb][j], sh = shv[cb][j];
        float v[2]; int id[2];
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          float m = -INFINITY;
          int mi = 0;
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              float t = (r == 0 ? r0[2 * hx + c][j] : r1[2 * hx + c][j]) * sc + sh;
              if (ep.relu) t = fmaxf(t, 0.f);
              if constexpr (AM) {
                if (t > m) { m = t; mi = 2 * r + c; }
              } else {
                m = fmaxf(m, t);
              }
            }
          const float up = __shfl_xor(m, XQ * C::YT, 64);            // the other z plane's window maximum
          if constexpr (AM) {
            const int upi = __shfl_xor(mi, XQ * C::YT, 64);
            // lane of plane 0 decides: the z + 1 plane only wins when strictly larger
            const bool upper = up > m;
            v[hx] = upper ? up : m;
            id[hx] = upper ? 4 + upi : mi;
          } else {
            v[hx] = fmaxf(m, up); id[hx] = 0;
          }
        }
        const int co = co0 + j;
        if (jz != 0 || co >= cout || zp >= PD || yp >= PH) continue;
        const size_t o = ((size_t)b * cout + co) * ((size_t)PD * PH * PW) + ((size_t)zp * PH + yp) * PW + xp;
        if (xp + 1 < PW && (PW & 1) == 0) {
          *reinterpret_cast<f32x2*>(out + o) = f32x2{v[0], v[1]};
          if constexpr (AM) { ep.argmax[o] = (unsigned char)id[0]; ep.argmax[o + 1] = (unsigned char)id[1]; }
        } else {
          if (xp < PW) { out[o] = v[0]; if constexpr (AM) ep.argmax[o] = (unsigned char)id[0]; }
          if (xp + 1 < PW) { out[o + 1] = v[1]; if constexpr (AM) ep.argmax[o + 1] = (unsigned char)id[1]; }
        }
      }
    } else {
      if (!(z < D && y < H && x < W)) continue;
      const bool quad_ok = ((W & 3) == 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = co0 + j;
        if (co >= cout) continue;
        const float sc = scv[cb][j], sh = shv[cb][j];
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
    }
  }
  W24W_STAMP(4); W24W_STAMP(6);
}

template <int XQ, bool POOL, bool AM>
int launch_w(const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W, m3d_w2q::Epi ep, hipStream_t st) {
  using C = CfgW<XQ, POOL>;
  const int tiles_x = (W + C::TX - 1) / C::TX, tiles_y = (H + C::TY - 1) / C::TY, tiles_z = (D + C::TZ - 1) / C::TZ;
  const int ncb_total = ((cout + 31) / 32 + 1) / 2 * 2;
  const int co_tiles = (cout + 63) / 64;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * co_tiles;
  if (blocks > 0x7FFFFFFFll || B > 65535) return M3D_EUNSUPPORTED;
  const size_t lds = sizeof(float) * C::SMEM_FLOATS;
#ifdef M3D_W2_STAMPS
  ep.stamps = g_w24w_stamps;
#endif
  auto kern = conv3d_wino24w_kernel<XQ, POOL, AM>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, B, ep.ksplit > 1 ? ep.ksplit : 1), dim3(C::NT), lds, st, in, wp, out, cin, cout, D, H, W,
                     tiles_x, tiles_y, tiles_z, ncb_total, ep);
  return m3d::check_launch("conv3d_wino24w");
}

}  // namespace

namespace m3d_w24w {

// xt = the tile id of the shared tile choice: 32 -> 64 x 2 x 2 outputs x 64 channels per workgroup, 16 -> 32 x 4 x 2, 8 -> 16 x 8 x 2
int launch(int xt, bool pool, bool argmax, const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W,
           m3d_w2q::Epi ep, hipStream_t st) {
  if (argmax && !pool) return M3D_EINVAL;
  if (xt == 32) {
    if (!pool) return launch_w<16, false, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    if (!argmax) return launch_w<16, true, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    return launch_w<16, true, true>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  }
  if (xt == 16) {
    if (!pool) return launch_w<8, false, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    if (!argmax) return launch_w<8, true, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
    return launch_w<8, true, true>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  }
  if (xt == 8 && !pool) return launch_w<4, false, false>(in, wp, out, B, cin, cout, D, H, W, ep, st);
  return M3D_EUNSUPPORTED;
}

}  // namespace m3d_w24w
