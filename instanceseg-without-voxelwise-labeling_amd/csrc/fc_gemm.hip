// Fully-connected layers of the box head as an fp32 MFMA split-K GEMM for gfx950:  out[M,N] = act(x[M,K] . W[N,K]^T + b).
//
// What it replaces: the cuBLAS SGEMMs behind nn.Linear at lib/modeling/fast_rcnn_heads.py:84-85,114-115 (fc1: K = C*7^3 =
// 87 808, N = 1024 - 180 GFLOP at 1000 RoIs, the second-largest FLOP block of the pipeline and 360 MB of weights; fc2) and
// :15-19,42-45 (cls_score / bbox_pred).  Both operands are K-contiguous exactly as PyTorch stores them (x = the RoIAlign
// output viewed [R, C*343]; W = nn.Linear.weight [out, in]): no transposed copy of the 360 MB matrix is ever made.
//
// Roofline: MFMA-bound (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s): 2*M*N*K FLOP vs (M+N)*K*4 bytes read once.
// Design:
//  * workgroup tile 128 (rows of x) x 128 (rows of W) x BK = 32, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA blocks
//    (64 accumulator registers), two workgroups per CU so one's staging / barrier hides under the other's MFMAs;
//  * K is split into `slices`; one workgroup = (slice, tile).  Workgroups that run at the same time on one XCD are given the
//    SAME slice and neighbouring tiles (blockIdx -> unit remap below), so the x / W panels of a slice are fetched from HBM once
//    per XCD round and re-used out of that XCD's L2 by the other tiles: HBM traffic stays near (M+N)*K*4 although every
//    workgroup streams 256 rows x K/slices;
//  * staging global -> registers -> LDS in 16-byte quads (rows are 128-byte runs: 8 lanes x 16 B), LDS rows padded to 36
//    floats so the ds_read_b128 fragment reads (lane = row, half-wave = k quad) are conflict-free; double-buffered, one
//    barrier per chunk, next chunk's global loads issued before the chunk's 64 MFMAs per wave;
//  * the k index inside an 8-deep group is permuted (lane half h holds k = 4h..4h+3) - identical on both operands, so each
//    b128 fragment feeds four MFMAs with no shuffles;
//  * split-K partials go to a caller workspace and are summed in slice order by a second kernel (deterministic; fused with
//    bias + ReLU).  slices == 1 stores directly.
#include <type_traits>

#include "m3d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, LS = BK + 4;           // LS: padded LDS row stride (floats)
constexpr int kStageFloats = (BM + BN) * LS;                       // one buffer
constexpr int KGmap(int g) { return (g & 3) + 8 * (g >> 2); }      // accumulator register g -> row inside a 32x32 block (+4*(lane>>5))

struct FcArgs {
  const float* x; const float* w; const float* bias; float* out; float* part;
  int M, N, K, mt, nt, slices, chunks, relu;
  int slices_tail;        // K slices of the ragged last row tile's units (its own count: they are shorter and fill the left-over slots)
  int mt_full;            // row tiles run by the full-tile path; mt - mt_full (0 or 1) ragged last tile runs the cheap tail path
  int full_per_xcd, tail_per_xcd;   // units of each kind given to one XCD (grid = 8 * (full_per_xcd + tail_per_xcd))
};

__global__ __launch_bounds__(256, 2) void fc_gemm_kernel(FcArgs a) {
  extern __shared__ float lds[];                                   // 2 x [(BM + BN) x LS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- unit = (slice, tile).  Workgroups dealt to one XCD (blockIdx % 8) get a contiguous run of full-tile units (same slice,
  // neighbouring tiles: their x / W panels are shared through that XCD's L2) FOLLOWED by its share of the cheap ragged-tail units:
  // dispatched last, the short workgroups fill the slots that are left instead of pushing full tiles into a second round.
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tiles_full = a.mt_full * a.nt, nfull = tiles_full * a.slices, ntail = (a.mt - a.mt_full) * a.nt * a.slices_tail;
  int slice, tile;
  if (idx < a.full_per_xcd) {
    const int u = xcd * a.full_per_xcd + idx;
    if (u >= nfull) return;
    slice = u / tiles_full; tile = u - slice * tiles_full;
  } else {
    const int v = xcd * a.tail_per_xcd + (idx - a.full_per_xcd);
    if (idx - a.full_per_xcd >= a.tail_per_xcd || v >= ntail) return;
    slice = v / a.nt; tile = tiles_full + (v - slice * a.nt);
  }
  const int tm = tile / a.nt, tn = tile - tm * a.nt;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nsl = tile >= tiles_full ? a.slices_tail : a.slices;
  const int c0 = (int)((long long)slice * a.chunks / nsl), c1 = (int)((long long)(slice + 1) * a.chunks / nsl);

  // ---- staging: thread -> (row = tid/8 + 32 i, quad = tid%8), i = 0..3 for x and for W
  const int quad = tid & 7, row0 = tid >> 3;
  const float* px[4]; const float* pw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rm = min(m0 + row0 + 32 * i, a.M - 1), rn = min(n0 + row0 + 32 * i, a.N - 1);   // clamped: masked at the store
    px[i] = a.x + (size_t)rm * a.K + 4 * quad;
    pw[i] = a.w + (size_t)rn * a.K + 4 * quad;
  }
  // The loads of chunk c+1 are issued BEFORE chunk c's MFMA loop and nothing touches the loaded registers until commit()
  // after it - any arithmetic on them (even a K-tail select) makes the compiler wait for the loads in front of the loop.
  // The main loop therefore covers whole chunks only; a ragged last chunk (K % 32 != 0) is staged once, with the select, at the end.
  f32x4 sx[4], sw[4];
  auto fetch = [&](int c) __attribute__((always_inline)) {
    const int k = c * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sx[i] = *reinterpret_cast<const f32x4*>(px[i] + k);
      sw[i] = *reinterpret_cast<const f32x4*>(pw[i] + k);
    }
  };
  auto commit = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(buf + (row0 + 32 * i) * LS + 4 * quad) = sx[i];
      *reinterpret_cast<f32x4*>(buf + (BM + row0 + 32 * i) * LS + 4 * quad) = sw[i];
    }
  };

  const int full = a.K / BK;                                      // whole chunks in K
  const int c1f = min(c1, full);                                  // this slice's whole chunks are [c0, c1f)
  const bool direct = nsl == 1;
  float* dst = direct ? a.out : a.part + (size_t)slice * a.M * a.N;

  // ---- ragged last row tile (M % 128 in 1..64): only nb = 1 or 2 of its four 32-row blocks hold rows, so the four waves split
  // the COLUMNS instead - wave w owns row block w >> (nb == 2 ? 1 : 2)... one (nb = 1) or two (nb = 2) MFMA blocks per wave - and the
  // tile costs a quarter / a half of a full one (M = 1281 RoIs: 10.25 tile times instead of 11).  Separate, simple loop: the
  // full-tile path below keeps its registers and schedule.
  const int nb = min(4, (a.M - m0 + 31) / 32);
  if (tm >= a.mt_full) {                                           // nb is 1 or 2 here (make_plan)
    const int fr = lane & 31, fh = lane >> 5;
    const int rbk = nb == 2 ? (wave >> 1) : 0;                    // row block of this wave
    const int cb0 = nb == 2 ? 2 * (wave & 1) : wave;              // first column block; nb == 2: also cb0 + 1
    const int ncb = nb == 2 ? 2 : 1;
    const int oA = (rbk * 32 + fr) * LS + 4 * fh, oB = (BM + cb0 * 32 + fr) * LS + 4 * fh;
    f32x16 t[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) t[j][g] = 0.f;
    auto tail_compute = [&](const float* cur) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        const f32x4 fa = *reinterpret_cast<const f32x4*>(cur + oA + g * 8);
        const f32x4 fb0 = *reinterpret_cast<const f32x4*>(cur + oB + g * 8);
        const f32x4 fb1 = *reinterpret_cast<const f32x4*>(cur + oB + 32 * LS + g * 8);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          t[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb0[kk], t[0], 0, 0, 0);
          if (ncb == 2) t[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb1[kk], t[1], 0, 0, 0);
        }
      }
    };
    if (c0 < c1f) { fetch(c0); commit(lds); }
    __syncthreads();
    for (int c = c0; c < c1f; ++c) {
      const float* cur = lds + ((c - c0) & 1) * kStageFloats;
      fetch(c + 1 < c1f ? c + 1 : c);
      __builtin_amdgcn_sched_barrier(0);
      tail_compute(cur);
      commit(lds + ((c + 1 - c0) & 1) * kStageFloats);
      __syncthreads();
    }
    if (c1 > c1f) {                                               // ragged tail chunk of K
      const bool ok = full * BK + 4 * quad < a.K;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const int k = ok ? full * BK : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 vx = *reinterpret_cast<const f32x4*>(px[i] + k), vw = *reinterpret_cast<const f32x4*>(pw[i] + k);
        *reinterpret_cast<f32x4*>(lds + (row0 + 32 * i) * LS + 4 * quad) = ok ? vx : z;
        *reinterpret_cast<f32x4*>(lds + (BM + row0 + 32 * i) * LS + 4 * quad) = ok ? vw : z;
      }
      __syncthreads();
      tail_compute(lds);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j >= ncb) continue;
      const int n = n0 + (cb0 + j) * 32 + fr;
      const float b = (direct && a.bias && n < a.N) ? a.bias[n] : 0.f;
      const int mb = m0 + rbk * 32 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int m = mb + KGmap(g);
        if (m < a.M && n < a.N) {
          float v = t[j][g];
          if (direct) { v += b; if (a.relu) v = fmaxf(v, 0.f); }
          dst[(size_t)m * a.N + n] = v;
        }
      }
    }
    return;
  }

  // ---- fragments: wave (wm, wn) owns rows [wm*64, +64) x cols [wn*64, +64); lane = (row r, k-quad half h)
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
  const int offA = (wm * 64 + fr) * LS + 4 * fh;
  const int offB = (BM + wn * 64 + fr) * LS + 4 * fh;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  f32x4 fa[2][2], fb[2][2];                                       // fragment double buffer: group g+1 is read under group g's MFMAs
  auto read_frags = [&](const float* cur, int g, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[slot][i] = *reinterpret_cast<const f32x4*>(cur + offA + i * 32 * LS + g * 8);
      fb[slot][i] = *reinterpret_cast<const f32x4*>(cur + offB + i * 32 * LS + g * 8);
    }
  };
  auto compute = [&](const float* cur) __attribute__((always_inline)) {
    read_frags(cur, 0, 0);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      if (g + 1 < BK / 8) read_frags(cur, g + 1, (g + 1) & 1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][kk], fb[g & 1][j][kk], acc[i][j], 0, 0, 0);
    }
  };

  if (c0 < c1f) {
    fetch(c0);
    commit(lds);
  }
  __syncthreads();
  for (int c = c0; c < c1f; ++c) {
    const float* cur = lds + ((c - c0) & 1) * kStageFloats;
    fetch(c + 1 < c1f ? c + 1 : c);                               // last chunk: harmless re-fetch instead of a branch around the loads
    __builtin_amdgcn_sched_barrier(0);                            // keep the loads in front of the loop (the scheduler would sink them)
    compute(cur);
    commit(lds + ((c + 1 - c0) & 1) * kStageFloats);              // unconditional (after the last chunk nobody reads that buffer)
    __syncthreads();
  }
  if (c1 > c1f) {                                                 // ragged tail chunk of K (only the last slice, only if K % 32 != 0)
    const bool ok = full * BK + 4 * quad < a.K;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const int k = ok ? full * BK : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 vx = *reinterpret_cast<const f32x4*>(px[i] + k), vw = *reinterpret_cast<const f32x4*>(pw[i] + k);
      *reinterpret_cast<f32x4*>(lds + (row0 + 32 * i) * LS + 4 * quad) = ok ? vx : z;
      *reinterpret_cast<f32x4*>(lds + (BM + row0 + 32 * i) * LS + 4 * quad) = ok ? vw : z;
    }
    __syncthreads();
    compute(lds);
  }

  // ---- epilogue: register g of block (i, j) is out[m0 + wm*64 + i*32 + KG(g) + 4*fh][n0 + wn*64 + j*32 + fr]: for a fixed g the
  // half-wave writes one 128-byte row segment
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + fr;
    const float b = (direct && a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mb = m0 + wm * 64 + i * 32 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int m = mb + KGmap(g);
        if (m < a.M && n < a.N) {
          float v = acc[i][j][g];
          if (direct) { v += b; if (a.relu) v = fmaxf(v, 0.f); }
          dst[(size_t)m * a.N + n] = v;
        }
      }
    }
  }
}

// out[e] = act(bias[n] + sum_s part[s][e]), slices summed in index order (deterministic).  Rows below m_full were cut into
// `slices` K ranges, the ragged last row tile into `slices_tail`; a part with one slice was stored directly (bias + act applied).
__global__ __launch_bounds__(256) void fc_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                        float* __restrict__ out, long long MN, int N, int slices, int slices_tail,
                                                        long long full_elems /* m_full * N */, int relu) {
  // four consecutive outputs per thread (N % 4 == 0: a quad never straddles a row or the full / tail boundary), every slice's quad one
  // 16-byte load, all issued before the first add; the sum order per element is slice 0, 1, 2, ... as before (bit-identical results)
  if ((N & 3) == 0 && ((((size_t)part) | ((size_t)out) | ((size_t)bias)) & 15) == 0) {
    typedef float f32x4r __attribute__((ext_vector_type(4)));
    const long long Q = MN >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < Q; q += (long long)gridDim.x * 256) {
      const long long e = q << 2;
      const int ns = e < full_elems ? slices : slices_tail;
      if (ns == 1) continue;
      f32x4r v = *reinterpret_cast<const f32x4r*>(part + e);
      int s = 1;
      for (; s + 3 < ns; s += 4) {
        const f32x4r a = *reinterpret_cast<const f32x4r*>(part + (size_t)s * MN + e);
        const f32x4r b = *reinterpret_cast<const f32x4r*>(part + (size_t)(s + 1) * MN + e);
        const f32x4r c = *reinterpret_cast<const f32x4r*>(part + (size_t)(s + 2) * MN + e);
        const f32x4r d = *reinterpret_cast<const f32x4r*>(part + (size_t)(s + 3) * MN + e);
        v += a; v += b; v += c; v += d;
      }
      for (; s < ns; ++s) v += *reinterpret_cast<const f32x4r*>(part + (size_t)s * MN + e);
      if (bias) v += *reinterpret_cast<const f32x4r*>(bias + (int)(e % N));
      if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      *reinterpret_cast<f32x4r*>(out + e) = v;
    }
    return;
  }
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < MN; e += (long long)gridDim.x * 256) {
    const int ns = e < full_elems ? slices : slices_tail;
    if (ns == 1) continue;
    float v = part[e];
    for (int s = 1; s < ns; ++s) v += part[(size_t)s * MN + e];
    if (bias) v += bias[(int)(e % N)];
    out[e] = relu ? fmaxf(v, 0.f) : v;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same GEMM on the bf16 matrix cores, WITHOUT giving up fp32 accuracy ("bf16x3 split", 6 products per fp32 product).
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the FLOP rate of the fp32-input MFMA.  An fp32 number has a 24-bit significand;
// cut by TRUNCATION into three bf16 numbers (8-bit significands, the same exponent range as fp32)
//     x = xh + xm + xl          xh = top 8 bits,  xm = top 8 bits of (x - xh),  xl = x - xh - xm   (8 bits are left: exact)
// the cut is exact, each bf16 x bf16 product is exact in fp32 (16 bits), and
//     x.w = xh.wh + (xh.wm + xm.wh) + (xm.wm + xh.wl + xl.wh)  +  [xm.wl + xl.wm + xl.wl  <= 2^-23 |x.w|: dropped]
// so six bf16 MFMAs accumulate in fp32 what one fp32 MFMA step does, to the same accuracy (the dropped terms are below the
// rounding of the fp32 accumulation itself; tests/test_gpu_ops.py compares both kernels with fp64), at 16 / 6 = 2.7x the rate.
// The weights are cut once, off line (m3d_linear_bf16x3_pack: three bf16 planes in tile order = 6 bytes per weight instead of 4);
// x (the RoIAlign output) is cut on the way from global memory to LDS (4 VALU + 1.5 pack ops per element, under the other
// resident workgroup's MFMAs).  Same tiling as the fp32 kernel: 128 x 128 workgroup tile, 4 waves as 2 x 2, each 64 x 64 = 2 x 2
// MFMA blocks; per 32-deep K chunk a wave issues 2 k16 steps x 4 blocks x 6 products = 48 MFMAs of 32 cycles (the fp32 kernel:
// 64 of 64 cycles).  One LDS buffer of [2 operands][3 planes][128 rows][64 B] = 48 KB; 16-byte units XOR-swizzled by (row >> 2) & 3:
// ds_read_b128 serves lanes {0-3,12-15,20-27} / {4-11,16-19,28-31} (and the same + 32) per LDS cycle over 64 banks, and in each
// of those groups the four rows that share row % 4 differ in (row >> 2) & 3, so a fragment read touches every bank once
// (a swizzle by (row >> 1) & 3, right for 8 consecutive lanes over 32 banks, measured 36 % of the LDS cycles as conflicts), three workgroups per CU (<= 168 VGPRs): whenever one wave of a
// SIMD sits in its load-wait / cut / barrier phase two others can issue MFMAs; the next chunk waits in registers during the MFMAs.
// Roofline: MFMA (bf16, 2.5 PFLOP/s dense): 6 x 2MNK issued FLOP.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int X3_RSW = 16;                        // LDS row stride in dwords: 64 B, no pad; the 16-byte unit u of row r sits at u ^ ((r >> 2) & 3)
constexpr int X3_PLANE = 128 * X3_RSW;            // dwords of one plane
constexpr int X3_OPER = 3 * X3_PLANE;             // dwords of one operand (three planes)
constexpr int X3_CHUNK_U4 = 3 * 128 * 4;          // 16-byte units of one packed (tile, chunk) weight block

// packed[tn][c][s][row][32 k] (bf16) <- W[N][K] fp32; rows beyond N are zero.  One thread = 8 consecutive k of one row.
__global__ __launch_bounds__(256) void fc_x3_pack_kernel(const float* __restrict__ w, int N, int K, u32x4* __restrict__ packed,
                                                         int nt, int chunks) {
  const long long total = (long long)nt * chunks * 128 * 4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int q = (int)(e & 3), row = (int)((e >> 2) & 127);
    const long long tc = e >> 9;
    const int c = (int)(tc % chunks), tn = (int)(tc / chunks);
    const int n = tn * 128 + row;
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = n < N ? w[(size_t)n * K + (size_t)c * 32 + 8 * q + j] : 0.f;
      const unsigned uh = __float_as_uint(v) & 0xFFFF0000u;
      const float r1 = v - __uint_as_float(uh);
      const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
      const float r2 = r1 - __uint_as_float(um);
      h[j] = uh >> 16; m[j] = um >> 16; l[j] = __float_as_uint(r2) >> 16;
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ph[j] = h[2 * j] | (h[2 * j + 1] << 16); pm[j] = m[2 * j] | (m[2 * j + 1] << 16); pl[j] = l[2 * j] | (l[2 * j + 1] << 16);
    }
    u32x4* dst = packed + (size_t)tc * X3_CHUNK_U4 + row * 4 + q;
    dst[0] = ph; dst[512] = pm; dst[1024] = pl;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// "f16x2 split" (round 6): the same GEMM with THREE products per fp32 product instead of six.  fp16 carries an 11-bit significand, so
//     x s = xh + xl + e,   xh = fp16(x s) (round to nearest),  xl = fp16(x s - xh),  |e| <= 2^-22 |x s|
// (s = a power of two that puts the operand's largest magnitude just under 2^15: fp16 has 30 binades, values 2^18 below the largest keep
// all 22 bits, smaller ones an absolute error of 2^-40 of the largest), and
//     x.w = xh.wh + (xh.wl + xl.wh)  +  [xl.wl <= 2^-22 |x.w|, e-terms <= 2^-22 |x.w|: dropped]
// The dropped terms are random-sign and four times the rounding of ONE fp32 product; the fp32 accumulation of K = 87 808 products that
// both this kernel and an SGEMM perform rounds by 2^-24 of the running SUM at every step, which is what the error against fp64
// consists of in both (tests/test_gpu_ops.py measures the two against fp64 on the shipped shape).  v_mfma_f32_32x32x16_f16 runs at
// the bf16 form's rate, so the MFMA time halves; the packed weights are 4 bytes per element (two planes) instead of 6.
// Scales: the weight's from its largest magnitude at pack time (kept beside the planes); x's from a caller-supplied bound of max|x|
// (RoIAlign averages trilinear interpolations, convex combinations of feature-map values: max|feature map| bounds it and is 16 MB to
// sweep instead of 440) or, without one, from a sweep of x itself.  Products are exact in fp32 (22 bits), sums stay below 2^47.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

using m3d::f16_scale_of;      // scale 2^k with bound * 2^k in [2^14, 2^15) and its inverse (m3d_common.h; no host read of the bound)

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
  unsigned m = 0;
  const long long n4 = (((uintptr_t)x & 15) == 0) ? n / 4 : 0;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
    const f32x4 v = x4[e];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const unsigned b = __float_as_uint(v[j]) & 0x7FFFFFFFu; m = b > m ? b : m; }
  }
  for (long long e = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const unsigned b = __float_as_uint(x[e]) & 0x7FFFFFFFu; m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
  __shared__ unsigned wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {                                         // ONE atomic per workgroup (per-wave atomics on one word serialise: 8192 of
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));                // them made a 16 MB sweep take 50-150 us)
    if (m) atomicMax(out, m);                                     // |x| as bits: non-negative floats order like their bit patterns
  }
}

// packed[tn][c][plane h, l][row][32 k] (fp16, scaled) <- W[N][K] fp32; rows beyond N are zero.  One thread = 8 consecutive k of one row.
__global__ __launch_bounds__(256) void fc_f16_pack_kernel(const float* __restrict__ w, int N, int K, u32x4* __restrict__ packed,
                                                          int nt, int chunks, const float* __restrict__ wamax) {
  float sw, inv;
  f16_scale_of(*wamax, sw, inv);
  const long long total = (long long)nt * chunks * 128 * 4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int q = (int)(e & 3), row = (int)((e >> 2) & 127);
    const long long tc = e >> 9;
    const int c = (int)(tc % chunks), tn = (int)(tc / chunks);
    const int n = tn * 128 + row;
    u32x4 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16x2 hh, ll;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float v = (n < N ? w[(size_t)n * K + (size_t)c * 32 + 8 * q + 2 * j + t] : 0.f) * sw;
        const _Float16 h = (_Float16)v;
        hh[t] = h; ll[t] = (_Float16)(v - (float)h);
      }
      ph[j] = __builtin_bit_cast(unsigned, hh); pl[j] = __builtin_bit_cast(unsigned, ll);
    }
    u32x4* dst = packed + (size_t)tc * (2 * 128 * 4) + row * 4 + q;
    dst[0] = ph; dst[512] = pl;
  }
}

struct FcX3Args {
  const float* x; const u32x4* wp; const float* bias; float* out; float* part;
  int M, N, K, mt, nt, slices, chunks, relu, per_xcd;
  const float* xbound; const float* wamax;     // F16 = 1: device scalars, a bound of max|x| and max|W| (the scales' sources)
  int x_alias;             // > 0 (tuning build only, option tune_fc_x_alias): row m of x is read from row m % x_alias - a cache-resident A
                           // operand: what the GEMM costs when its operand is free (WRONG results; the f-1 lower bound, tools/f1_ab.py)
  // XM = 1 (f-1 A/B): x[m][k] is not read but computed - the RoIAlign gather in the operand loader (taps: roi_tap_table_kernel)
  const float* feat; const int4* taps; const int* roi_batch; int fC, fS, fH, fW;
};

// one element of the RoIAlign output of RoI row `tp` (its 42 tap entries): k = channel * 343 + (ph * 7 + pw) * 7 + ps, the memory order
// of the reference's output tensor (roi_align_kernel_3d.cu:87-91); 2 x 2 x 2 samples x 8 corners, mean over the samples
__device__ inline float roi_gather_element(const float* __restrict__ fb /* feature map of the RoI's batch item */, const int4* __restrict__ tp,
                                           int k, int S, int H, int W) {
  const int ch = k / 343, bin = k - ch * 343;
  const int ph = bin / 49, rem = bin - ph * 49, pw = rem / 7, ps = rem - pw * 7;
  const float* f = fb + (size_t)ch * S * H * W;
  int zi[4], yi[4], xi[4];
  float zw[4], yw[4], xw[4];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int4 ez = tp[2 * ps + a], ey = tp[14 + 2 * ph + a], ex = tp[28 + 2 * pw + a];
    zi[2 * a] = ez.x * H * W; zi[2 * a + 1] = ez.y * H * W; zw[2 * a] = __int_as_float(ez.z); zw[2 * a + 1] = __int_as_float(ez.w);
    yi[2 * a] = ey.x * W; yi[2 * a + 1] = ey.y * W; yw[2 * a] = __int_as_float(ey.z); yw[2 * a + 1] = __int_as_float(ey.w);
    xi[2 * a] = ex.x; xi[2 * a + 1] = ex.y; xw[2 * a] = __int_as_float(ex.z); xw[2 * a + 1] = __int_as_float(ex.w);
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float az = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* row = f + zi[i] + yi[j];
      const float ax = (xw[0] * row[xi[0]] + xw[1] * row[xi[1]]) + (xw[2] * row[xi[2]] + xw[3] * row[xi[3]]);
      az += yw[j] * ax;
    }
    acc += zw[i] * az;
  }
  return 0.125f * acc;
}

// WR = wave rows of the workgroup: 2 -> 128 x 128 tile, 256 threads, 2 workgroups per CU;  4 -> 256 x 128 tile, 512 threads, one
// workgroup per CU (x rows are re-used by twice the MFMAs: 28 instead of 40 KB of global loads per 128 x 128 x 32 of work - the
// loads, not the MFMAs, were what the 128-row kernel waited on: 0.97 ms without them, 1.15 with every load an L2 hit, 1.29 real).
// Wave schedule of one K chunk (F0 / F1 = the fragments of its two k16 steps, 12 ds_read_b128 each):
//     read F1 | 24 MFMA(F0) | barrier A (the chunk's LDS image is free) | cut + LDS writes of chunk c+1 under 16 MFMA(F1) |
//     barrier B | global loads of chunk c+2 -> registers | read F0 of chunk c+1 | 8 MFMA(F1)
// so every LDS / barrier latency has MFMAs of the same wave to hide under, not only those of the other wave of the SIMD.
// F16 = 1: the f16x2 split (two planes per operand, three products; see above) - same tiles, staging and schedule.
template <int WR, int XM = 0, int F16 = 0>
__global__ __launch_bounds__(128 * WR, WR == 2 ? 2 : 1) void fc_x3_gemm_kernel(FcX3Args a) {
  constexpr int TBM = 64 * WR, NT = 128 * WR;                      // tile rows, threads
  constexpr int NPL = F16 ? 2 : 3;                                 // planes per operand
  constexpr int CHUNK_U4 = NPL * 128 * 4;                          // 16-byte units of one packed (tile, chunk) weight block
  constexpr int XPL = TBM * X3_RSW, XOP = NPL * XPL;               // dwords of an x plane / of the x operand; W planes: X3_PLANE
  constexpr int NXI = TBM * 4 / NT, NWI = NPL * 128 * 4 / NT;      // rows of x (two float4 each) / 16-byte units of W per thread and chunk
  using frag_t = std::conditional_t<F16 != 0, f16x8, bf16x8>;
  float xs = 1.f, out_s0 = 1.f, out_s1 = 1.f;                      // F16: x scale; the two factors that undo the x and W scales
  if constexpr (F16) {
    float ws_, iw;
    f16_scale_of(*a.xbound, xs, out_s0);
    f16_scale_of(*a.wamax, ws_, iw);
    out_s1 = iw;
  }
  extern __shared__ float lds_f[];
  unsigned* const lds = reinterpret_cast<unsigned*>(lds_f);        // [x planes h, m, l: TBM rows][W planes h, m, l: 128 rows] x 16 dwords
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // unit = (slice, tile): the workgroups one XCD runs together share a slice and neighbouring tiles (as in the fp32 kernel)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tiles = a.mt * a.nt, u = xcd * a.per_xcd + idx;
  if (u >= tiles * a.slices) return;
  const int slice = u / tiles, tile = u - slice * tiles;
  const int tm = tile / a.nt, tn = tile - tm * a.nt;
  const int m0 = tm * TBM, n0 = tn * BN;
  const int c0 = (int)((long long)slice * a.chunks / a.slices), c1 = (int)((long long)(slice + 1) * a.chunks / a.slices);

  // ---- staging: x: thread -> (row = tid/4 + (NT/4) i, 8 k at 8 (tid % 4)), two float4 per row;  W: 16-byte unit tid + NT i of the
  // packed (tile, chunk) block = (plane, row, unit tid % 4).  LDS unit u of row r sits at u ^ ((r >> 2) & 3).
  const int oct = tid & 3, row0 = tid >> 2;
  const float* px[NXI];
  const int4* ptap[NXI];                                           // XM = 1: the row's tap table
  const float* pfeat[NXI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int mrow = min(m0 + row0 + (NT / 4) * i, a.M - 1);       // clamped: masked at the store
    if constexpr (XM == 1) {
      ptap[i] = a.taps + (size_t)mrow * 42;
      pfeat[i] = a.feat + (size_t)a.roi_batch[mrow] * a.fC * a.fS * a.fH * a.fW;
      px[i] = nullptr;
    } else {
      px[i] = a.x + (size_t)(a.x_alias > 0 ? mrow % a.x_alias : mrow) * a.K + 8 * oct;
    }
  }
  const u32x4* pw = a.wp + (size_t)tn * a.chunks * CHUNK_U4 + tid;
  const int swq = 4 * (oct ^ ((row0 >> 2) & 3));                   // NT/4 is a multiple of 16: every row of this thread swizzles alike
  f32x4 sx[NXI][2]; u32x4 sw[NWI];
  auto fetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      if constexpr (XM == 1) {
        const int k0 = c * 32 + 8 * oct;
#pragma unroll
        for (int e = 0; e < 8; ++e) sx[i][e >> 2][e & 3] = roi_gather_element(pfeat[i], ptap[i], k0 + e, a.fS, a.fH, a.fW);
      } else {
        sx[i][0] = *reinterpret_cast<const f32x4*>(px[i] + (size_t)c * 32);
        sx[i][1] = *reinterpret_cast<const f32x4*>(px[i] + (size_t)c * 32 + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < NWI; ++i) sw[i] = pw[(size_t)c * CHUNK_U4 + NT * i];
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      u32x4 ph, pm, pl;
      if constexpr (F16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0 = sx[i][j >> 1][2 * (j & 1)] * xs, v1 = sx[i][j >> 1][2 * (j & 1) + 1] * xs;
          const f16x2 hh = {(_Float16)v0, (_Float16)v1};
          const f16x2 ll = {(_Float16)(v0 - (float)hh[0]), (_Float16)(v1 - (float)hh[1])};
          ph[j] = __builtin_bit_cast(unsigned, hh); pl[j] = __builtin_bit_cast(unsigned, ll);
        }
        unsigned* d = lds + (row0 + (NT / 4) * i) * X3_RSW + swq;
        *reinterpret_cast<u32x4*>(d) = ph;
        *reinterpret_cast<u32x4*>(d + XPL) = pl;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {                              // elements 2j, 2j+1 of this row's 8
          const float v0 = sx[i][j >> 1][2 * (j & 1)], v1 = sx[i][j >> 1][2 * (j & 1) + 1];
          const unsigned h0 = __float_as_uint(v0) & 0xFFFF0000u, h1 = __float_as_uint(v1) & 0xFFFF0000u;
          const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
          const unsigned q0 = __float_as_uint(r0) & 0xFFFF0000u, q1 = __float_as_uint(r1) & 0xFFFF0000u;
          const float t0 = r0 - __uint_as_float(q0), t1 = r1 - __uint_as_float(q1);
          ph[j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
          pm[j] = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
          pl[j] = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
        }
        unsigned* d = lds + (row0 + (NT / 4) * i) * X3_RSW + swq;
        *reinterpret_cast<u32x4*>(d) = ph;
        *reinterpret_cast<u32x4*>(d + XPL) = pm;
        *reinterpret_cast<u32x4*>(d + 2 * XPL) = pl;
      }
    }
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int un = NT * i;                                       // unit index of thread 0: plane un / 512, row (un % 512) / 4 + row0
      *reinterpret_cast<u32x4*>(lds + XOP + (un >> 9) * X3_PLANE + (((un & 511) >> 2) + row0) * X3_RSW + swq) = sw[i];
    }
  };

  // ---- fragments: wave (wm, wn) owns rows [wm*64, +64) x cols [wn*64, +64); lane = (row r, k half h): 8 bf16 at k = 16 t + 8 h.
  // unit of k16 step t = (2t + h) ^ swz = (h ^ swz) ^ 2t: the two steps' addresses differ by XOR 8 dwords
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
  const int swz = (fr >> 2) & 3;
  const int offA0 = (wm * 64 + fr) * X3_RSW + 4 * (fh ^ swz), offA1 = offA0 ^ 8;
  const int offB0 = XOP + (wn * 64 + fr) * X3_RSW + 4 * (fh ^ swz), offB1 = offB0 ^ 8;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  struct Frags { frag_t a[2][NPL], b[2][NPL]; };
  Frags f0, f1;
  auto read_frags = [&](Frags& f, int oa, int ob) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < NPL; ++s) {
        f.a[i][s] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(lds + oa + s * XPL + i * 32 * X3_RSW));
        f.b[i][s] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(lds + ob + s * X3_PLANE + i * 32 * X3_RSW));
      }
  };
  // products p0..p1 of one k16 step, small terms first: bf16x3 (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); f16x2 (l,h) (h,l) (h,h)
  constexpr int NP = F16 ? 3 : 6, NP_MID = F16 ? 2 : 4;
  auto mfmas = [&](const Frags& f, int p0, int p1) __attribute__((always_inline)) {
    constexpr int PA[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0}, PB[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p < p0 || p >= p1) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][PA[p]], f.b[j][PB[p]], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA[p]], f.b[j][PB[p]], acc[i][j], 0, 0, 0);
        }
    }
  };

  if (c0 < c1) {
    fetch(c0); commit();
    __syncthreads();
    fetch(min(c0 + 1, c1 - 1));
    read_frags(f0, offA0, offB0);
  }
  for (int c = c0; c < c1; ++c) {
    read_frags(f1, offA1, offB1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f0, 0, NP);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                              // A: everybody holds the chunk's fragments; its LDS image is free
    commit();                                                     // chunk c + 1 (after the last chunk: a re-write nobody reads) ...
    mfmas(f1, 0, NP_MID);                                         // ... the compiler interleaves these 16 (8) MFMAs with the cut
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                              // B: the image of chunk c + 1 is complete
    fetch(min(c + 2, c1 - 1));
    read_frags(f0, offA0, offB0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f1, NP_MID, NP);
  }

  const bool direct = a.slices == 1;
  float* dst = direct ? a.out : a.part + (size_t)slice * a.M * a.N;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + fr;
    const float b = (direct && a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mb = m0 + wm * 64 + i * 32 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int m = mb + KGmap(g);
        if (m < a.M && n < a.N) {
          float v = acc[i][j][g];
          if constexpr (F16) v = (v * out_s0) * out_s1;            // undo the operand scales (powers of two: exact)
          if (direct) { v += b; if (a.relu) v = fmaxf(v, 0.f); }
          dst[(size_t)m * a.N + n] = v;
        }
      }
    }
  }
}

// 256 x 256 tiles, BOTH operands fp32 in HBM and cut on the way to LDS (no packed weights): 64 KB of global loads per chunk for
// four 128 x 128 x 32 units of work = 16 KB per unit, against 28 (256 x 128 tile, packed W) and 40 (128 x 128): the bf16x3 GEMM is
// bound by the bytes it loads per MFMA, so this is the variant for many rows.  8 waves as 4 x 2, each 64 rows x
// 128 columns = 2 x 4 MFMA blocks (128 accumulator registers), one workgroup per CU; LDS [x: 3 planes x 256 rows][W: 3 x 256]
// x 64 B = 96 KB, swizzled as above.  The fragments of a k16 step are read in two column halves so that 48 fragment registers
// suffice beside the accumulators.
// F16 = 1 (round 6): the f16x2 split on these tiles - x (fp32) cut in the kernel, W from the packed fp16 planes (two 128-row blocks per
// 256-column tile): 32 + 32 KB of loads per chunk for 48 MFMAs per wave, LDS 64 KB.
template <int F16 = 0>
__global__ __launch_bounds__(512, 1) void fc_x3b_gemm_kernel(FcX3Args a) {
  constexpr int TB = 256;                                          // 512 threads
  constexpr int NPL = F16 ? 2 : 3;
  constexpr int PL = TB * X3_RSW, OP = NPL * PL;                   // dwords of a plane / of an operand
  using frag_t = std::conditional_t<F16 != 0, f16x8, bf16x8>;
  float xs = 1.f, out_s0 = 1.f, out_s1 = 1.f;
  if constexpr (F16) {
    float ws_, iw;
    f16_scale_of(*a.xbound, xs, out_s0);
    f16_scale_of(*a.wamax, ws_, iw);
    out_s1 = iw;
  }
  extern __shared__ float lds_f[];
  unsigned* const lds = reinterpret_cast<unsigned*>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tiles = a.mt * a.nt, u = xcd * a.per_xcd + idx;
  if (u >= tiles * a.slices) return;
  const int slice = u / tiles, tile = u - slice * tiles;
  const int tm = tile / a.nt, tn = tile - tm * a.nt;
  const int m0 = tm * TB, n0 = tn * TB;
  const int c0 = (int)((long long)slice * a.chunks / a.slices), c1 = (int)((long long)(slice + 1) * a.chunks / a.slices);
  const float* wsrc = reinterpret_cast<const float*>(a.wp);       // here: the nn.Linear weight itself, [N, K] fp32

  // ---- staging: thread -> (row = tid/4 + 128 i, 8 k at 8 (tid % 4)), i = 0, 1, for x and for W
  const int oct = tid & 3, row0 = tid >> 2;
  const float* px[2]; const float* pq[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    px[i] = a.x + (size_t)min(m0 + row0 + 128 * i, a.M - 1) * a.K + 8 * oct;      // clamped: masked at the store
    pq[i] = wsrc + (size_t)min(n0 + row0 + 128 * i, a.N - 1) * a.K + 8 * oct;
  }
  const int swq = 4 * (oct ^ ((row0 >> 2) & 3));
  f32x4 sx[2][2], sq[2][2];
  // F16: unit tid of (128-row block 2 tn + b, plane s) = row tid / 4, 16-byte unit tid % 4 of the packed (block, chunk) image
  const u32x4* pwq[2] = {a.wp + (size_t)(2 * tn) * a.chunks * (2 * 128 * 4) + tid,
                         a.wp + (size_t)min(2 * tn + 1, (a.N + 127) / 128 - 1) * a.chunks * (2 * 128 * 4) + tid};
  auto fetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      sx[i][0] = *reinterpret_cast<const f32x4*>(px[i] + (size_t)c * 32);
      sx[i][1] = *reinterpret_cast<const f32x4*>(px[i] + (size_t)c * 32 + 4);
      if constexpr (F16) {
        sq[i][0] = __builtin_bit_cast(f32x4, pwq[i][(size_t)c * (2 * 128 * 4)]);            // plane h of block i
        sq[i][1] = __builtin_bit_cast(f32x4, pwq[i][(size_t)c * (2 * 128 * 4) + 512]);      // plane l
      } else {
        sq[i][0] = *reinterpret_cast<const f32x4*>(pq[i] + (size_t)c * 32);
        sq[i][1] = *reinterpret_cast<const f32x4*>(pq[i] + (size_t)c * 32 + 4);
      }
    }
  };
  auto cut_store = [&](const f32x4 (&v)[2], unsigned* d) __attribute__((always_inline)) {
    u32x4 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v0 = v[j >> 1][2 * (j & 1)], v1 = v[j >> 1][2 * (j & 1) + 1];
      const unsigned h0 = __float_as_uint(v0) & 0xFFFF0000u, h1 = __float_as_uint(v1) & 0xFFFF0000u;
      const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
      const unsigned q0 = __float_as_uint(r0) & 0xFFFF0000u, q1 = __float_as_uint(r1) & 0xFFFF0000u;
      const float t0 = r0 - __uint_as_float(q0), t1 = r1 - __uint_as_float(q1);
      ph[j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
      pm[j] = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
      pl[j] = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
    }
    *reinterpret_cast<u32x4*>(d) = ph;
    *reinterpret_cast<u32x4*>(d + PL) = pm;
    *reinterpret_cast<u32x4*>(d + 2 * PL) = pl;
  };
  auto commit = [&](int base = 0) __attribute__((always_inline)) {      // base: dword offset of the LDS image (F16: two images)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (F16) {
        u32x4 ph, pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0 = sx[i][j >> 1][2 * (j & 1)] * xs, v1 = sx[i][j >> 1][2 * (j & 1) + 1] * xs;
          const f16x2 hh = {(_Float16)v0, (_Float16)v1};
          const f16x2 ll = {(_Float16)(v0 - (float)hh[0]), (_Float16)(v1 - (float)hh[1])};
          ph[j] = __builtin_bit_cast(unsigned, hh); pl[j] = __builtin_bit_cast(unsigned, ll);
        }
        unsigned* d = lds + base + (row0 + 128 * i) * X3_RSW + swq;
        *reinterpret_cast<u32x4*>(d) = ph;
        *reinterpret_cast<u32x4*>(d + PL) = pl;
        unsigned* dw = lds + base + OP + (row0 + 128 * i) * X3_RSW + swq;
        *reinterpret_cast<u32x4*>(dw) = __builtin_bit_cast(u32x4, sq[i][0]);
        *reinterpret_cast<u32x4*>(dw + PL) = __builtin_bit_cast(u32x4, sq[i][1]);
      } else {
        cut_store(sx[i], lds + (row0 + 128 * i) * X3_RSW + swq);
        cut_store(sq[i], lds + OP + (row0 + 128 * i) * X3_RSW + swq);
      }
    }
  };

  // ---- fragments: wave (wm, wn): rows [wm*64, +64), columns [wn*128, +128)
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
  const int swz = (fr >> 2) & 3;
  const int offA0 = (wm * 64 + fr) * X3_RSW + 4 * (fh ^ swz), offA1 = offA0 ^ 8;
  const int offB0 = OP + (wn * 128 + fr) * X3_RSW + 4 * (fh ^ swz), offB1 = offB0 ^ 8;
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  frag_t fa[2][NPL], fb[2][NPL];
  auto read_a = [&](int oa) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s3 = 0; s3 < NPL; ++s3)
        fa[i][s3] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(lds + oa + s3 * PL + i * 32 * X3_RSW));
  };
  auto read_b = [&](int ob, int jh) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s3 = 0; s3 < NPL; ++s3)
        fb[j][s3] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(lds + ob + s3 * PL + (2 * jh + j) * 32 * X3_RSW));
  };
  auto mfmas = [&](int jh) __attribute__((always_inline)) {
    constexpr int PA[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0}, PB[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
#pragma unroll
    for (int p = 0; p < (F16 ? 3 : 6); ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (F16)
            acc[i][2 * jh + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][PA[p]], fb[j][PB[p]], acc[i][2 * jh + j], 0, 0, 0);
          else
            acc[i][2 * jh + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[p]], fb[j][PB[p]], acc[i][2 * jh + j], 0, 0, 0);
        }
  };

  if constexpr (F16) {
    // f16x2: half the MFMAs per fragment, so a fragment read in front of the MFMAs that need it is no longer hidden by the SIMD's other
    // wave (both run the same program between the same barriers).  Two sets of A and B fragments: every read is issued one MFMA group
    // (12 MFMAs) ahead of its use; the first fragments of the next chunk are read behind barrier B under the chunk's last 4 MFMAs.
    frag_t ga[2][NPL], gb[2][NPL];
    auto rd_a = [&](frag_t (&f)[2][NPL], int oa) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s3 = 0; s3 < NPL; ++s3)
          f[i][s3] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(lds + oa + s3 * PL + i * 32 * X3_RSW));
    };
    auto rd_b = [&](frag_t (&f)[2][NPL], int ob, int jh) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s3 = 0; s3 < NPL; ++s3)
          f[j][s3] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(lds + ob + s3 * PL + (2 * jh + j) * 32 * X3_RSW));
    };
    auto mm = [&](const frag_t (&A)[2][NPL], const frag_t (&Bf)[2][NPL], int jh, int p0, int p1) __attribute__((always_inline)) {
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};                  // (l,h) (h,l) (h,h)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (p < p0 || p >= p1) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][2 * jh + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i][PA[p]], Bf[j][PB[p]], acc[i][2 * jh + j], 0, 0, 0);
      }
    };
    // Two LDS images (2 x 64 KB): chunk c + 1 is cut and written into the OTHER image at any time during chunk c, so a chunk has ONE
    // barrier (the image of c + 1 is complete, nobody reads the image of c any more) instead of two with the cut squeezed between them.
    constexpr int IMG = 2 * OP;
    if (c0 < c1) {
      fetch(c0); commit(0);
      __syncthreads();
      fetch(min(c0 + 1, c1 - 1));
      rd_a(fa, offA0); rd_b(fb, offB0, 0);
    }
    for (int c = c0; c < c1; ++c) {
      const int bo = ((c - c0) & 1) * IMG, bn = IMG - bo;
      rd_b(gb, bo + offB0, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa, fb, 0, 0, 3);
      __builtin_amdgcn_sched_barrier(0);
      rd_a(ga, bo + offA1); rd_b(fb, bo + offB1, 0);
      commit(bn);                                                 // chunk c + 1 (after the last chunk: a write nobody reads) ...
      mm(fa, gb, 1, 0, 3);                                        // ... the compiler interleaves the cut with these 12 MFMAs
      __builtin_amdgcn_sched_barrier(0);
      rd_b(gb, bo + offB1, 1);
      fetch(min(c + 2, c1 - 1));
      __builtin_amdgcn_sched_barrier(0);
      mm(ga, fb, 0, 0, 3);
      __builtin_amdgcn_sched_barrier(0);
      mm(ga, gb, 1, 0, 2);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                                            // the image of chunk c + 1 is complete; the image of chunk c is free
      rd_a(fa, bn + offA0); rd_b(fb, bn + offB0, 0);              // first fragments of chunk c + 1, under the chunk's last 4 MFMAs
      __builtin_amdgcn_sched_barrier(0);
      mm(ga, gb, 1, 2, 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  if (c0 < c1) {
    fetch(c0); commit();
    __syncthreads();
    fetch(min(c0 + 1, c1 - 1));
  }
  for (int c = c0; c < c1; ++c) {
    read_a(offA0); read_b(offB0, 0);
    mfmas(0);
    read_b(offB0, 1);
    mfmas(1);
    read_a(offA1); read_b(offB1, 0);
    mfmas(0);
    read_b(offB1, 1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                              // A: every wave holds its last fragments; the image is free
    commit();                                                     // chunk c + 1, under the last 24 MFMAs
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                              // B
    fetch(min(c + 2, c1 - 1));
    __builtin_amdgcn_sched_barrier(0);                            // or the scheduler sinks the loads to just before their use
  }
  }

  const bool direct = a.slices == 1;
  float* dst = direct ? a.out : a.part + (size_t)slice * a.M * a.N;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 128 + j * 32 + fr;
    const float b = (direct && a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mb = m0 + wm * 64 + i * 32 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int m = mb + KGmap(g);
        if (m < a.M && n < a.N) {
          float v = acc[i][j][g];
          if constexpr (F16) v = (v * out_s0) * out_s1;
          if (direct) { v += b; if (a.relu) v = fmaxf(v, 0.f); }
          dst[(size_t)m * a.N + n] = v;
        }
      }
    }
  }
}

int x3b_slices(int M, int N, int K) {
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256), chunks = K / 32;
  if (const int ts = m3d::opt(m3d::OPT_TUNE_FC_SLICES); ts > 0) return ts < chunks ? ts : chunks;
  int s = 256 / (tiles > 0 ? tiles : 1);
  const int smax = chunks / 8 + 1 < 64 ? chunks / 8 + 1 : 64;
  return s < 1 ? 1 : (s > smax ? smax : s);
}

// Tile height and split-K factor of the bf16x3 kernel.  256-row tiles (one workgroup per CU: 256 slots) unless the rows they pad
// cost more than their re-use saves (few RoIs); 128-row tiles: two workgroups per CU (512 slots).
struct X3Plan { int wr, mt, nt, slices, per_xcd; };
X3Plan x3_plan(int M, int N, int K) {
  X3Plan best{2, 0, 0, 1, 0};
  double best_t = 1e30;
  const int chunks = K / 32, nt = (N + BN - 1) / BN;
  const int force = m3d::opt(m3d::OPT_TUNE_FC_SLICES);
  for (int wr = 2; wr <= 4; wr += 2) {
    const int tbm = 64 * wr, mt = (M + tbm - 1) / tbm, tiles = mt * nt;
    const double slots = wr == 2 ? 64.0 : 32.0;                   // per XCD
    const double rate = (wr == 2 ? 1.65e14 : 2.1e14) / (8.0 * slots) * (wr == 2 ? 1.0 : 1.0);   // fp32-equivalent FLOP/s of one resident workgroup
    const int smax = chunks / 8 + 1 < 64 ? chunks / 8 + 1 : 64;
    for (int s = 1; s <= smax; ++s) {
      if (force > 0 && s != (force < chunks ? force : chunks)) continue;
      const double rounds = ceil(ceil((double)tiles * s / 8.0) / slots);
      double t = rounds * (2.0 * tbm * BN * (double)K / s) / rate + rounds * 4e-6;
      if (s > 1) t += (double)s * M * N * 8.0 / 4e12 + 4e-6;
      if (t < best_t) { best_t = t; best = X3Plan{wr, mt, nt, s, (tiles * s + 7) / 8}; }
    }
  }
  if (const int w = m3d::opt(m3d::OPT_TUNE_FC_X3_ROWS); w == 128 || w == 256) {      // A/B tooling: force the tile height
    const int wr = w / 64, tbm = w, mt = (M + tbm - 1) / tbm, tiles = mt * nt;
    int s = best.slices;
    if (wr != best.wr) { const double slots = wr == 2 ? 512.0 : 256.0; s = (int)(slots / tiles); s = s < 1 ? 1 : (s > chunks / 8 + 1 ? chunks / 8 + 1 : s); }
    if (force > 0) s = force < chunks ? force : chunks;
    best = X3Plan{wr, mt, nt, s, (tiles * s + 7) / 8};
  }
  return best;
}

struct Plan { int mt, nt, chunks, slices, slices_tail, mt_full, full_per_xcd, tail_per_xcd; };

Plan make_plan(int M, int N, int K) {
  Plan p;
  p.mt = (M + BM - 1) / BM; p.nt = (N + BN - 1) / BN; p.chunks = (K + BK - 1) / BK;
  const int rem_blocks = ((M - (p.mt - 1) * BM) + 31) / 32;       // 32-row blocks of the last row tile
  p.mt_full = (rem_blocks <= 2) ? p.mt - 1 : p.mt;                // a last tile with 1-2 blocks runs the tail path (1/4, 1/2 tile time)
  const double tail_cost = rem_blocks == 1 ? 0.45 : 0.6;          // measured: the staging of 128 W rows per chunk does not shrink
  const double slots = 64.0;                                      // per XCD: 32 CUs x 2 resident workgroups
  const double flop = 2.0 * BM * BN * (double)K;                  // one full tile over all of K
  const double rate = 1.25e14 / 512.0;                            // a workgroup's share of ~80 % of the MFMA rate
  const int ntail_tiles = (p.mt - p.mt_full) * p.nt;
  const int smax = p.chunks / 4 + 1 < 64 ? p.chunks / 4 + 1 : 64;
  double best = 1e30; int bs = 1, bt = 1;
  // The full tiles choose the slice count as if they were alone.  The ragged last row tile gets its OWN count: its units are
  // staging-bound (0.45 / 0.6 of a full unit per chunk), so they are cut fine enough to be no longer than a full unit and, when
  // the slots the full units leave free allow it, so that everything is resident at once - M = 1281: 80 x 6 + 8 x 4 = the 512
  // slots, 1.82 ms; one common count (5: 440 units) left 14 % of the slots idle: 1.88 ms, and 6 + 6 = 528 units 2.2 ms
  // (tools/sweep_fc_tail.py).
  const int nlong = p.mt_full ? p.mt_full * p.nt : ntail_tiles;    // no full tile at all: the tail tiles are the whole problem
  for (int s = 1; s <= smax; ++s) {
    const double longs = ceil((double)nlong * s / 8.0), rounds = ceil(longs / slots);
    double t = rounds * (p.mt_full ? 1.0 : tail_cost) * flop / s / rate + rounds * 4e-6;
    if (s > 1) t += (double)s * M * N * 8.0 / 4e12 + 4e-6;          // partial write + read, reduce launch
    if (t < best) { best = t; bs = s; }
  }
  bt = bs;
  if (p.mt_full && ntail_tiles) {
    const double longs = ceil((double)nlong * bs / 8.0);
    const int spare = (int)(ceil(longs / slots) * slots - longs);
    const int need = (int)ceil(tail_cost * bs), fit = spare * 8 / ntail_tiles;
    bt = fit >= need ? (fit < 2 * bs ? fit : 2 * bs) : 2 * bs;
    bt = bt < 1 ? 1 : (bt > smax ? smax : bt);
  }
  if (const int ts = m3d::opt(m3d::OPT_TUNE_FC_SLICES); ts > 0) bs = bt = ts < p.chunks ? ts : p.chunks;   // A/B tooling only
  if (const int ts = m3d::opt(m3d::OPT_TUNE_FC_SLICES_TAIL); ts > 0) bt = ts < p.chunks ? ts : p.chunks;
  p.slices = bs; p.slices_tail = ntail_tiles ? bt : 1;
  p.full_per_xcd = (p.mt_full * p.nt * bs + 7) / 8;
  p.tail_per_xcd = (ntail_tiles * p.slices_tail + 7) / 8;
  return p;
}

}  // namespace

M3D_API size_t m3d_linear_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const Plan p = make_plan(M, N, K);
  const int sm = p.slices > p.slices_tail ? p.slices : p.slices_tail;
  return sm > 1 ? (size_t)sm * M * N * sizeof(float) : 16;
}

M3D_API int m3d_linear_forward(const float* d_x, const float* d_weight, const float* d_bias, float* d_out, int M, int N, int K,
                               int relu, void* d_ws, size_t ws_bytes, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return M3D_EINVAL;
  if (M == 0) return M3D_OK;
  if (!d_x || !d_weight || !d_out) return M3D_EINVAL;
  if (K % 4 != 0 || ((uintptr_t)d_x & 15) || ((uintptr_t)d_weight & 15)) return M3D_EUNSUPPORTED;   // 16-byte row quads
  const Plan p = make_plan(M, N, K);
  const int sm = p.slices > p.slices_tail ? p.slices : p.slices_tail;
  if (sm > 1 && (!d_ws || ws_bytes < (size_t)sm * M * N * sizeof(float))) return M3D_EWORKSPACE;
  FcArgs a{d_x, d_weight, d_bias, d_out, (float*)d_ws, M, N, K, p.mt, p.nt, p.slices, p.chunks, relu, p.slices_tail, p.mt_full,
           p.full_per_xcd, p.tail_per_xcd};
  const size_t lds = sizeof(float) * 2 * kStageFloats;
  hipStream_t st = m3d::as_stream(stream);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fc_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(fc_gemm_kernel, dim3(8 * (p.full_per_xcd + p.tail_per_xcd)), dim3(256), lds, st, a);
  if (sm > 1) {
    const long long MN = (long long)M * N;
    long long blocks = (MN + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fc_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)d_ws, d_bias, d_out, MN, N, p.slices,
                       p.slices_tail, (long long)(p.mt_full < p.mt ? p.mt_full * BM : M) * N, relu);
  }
  return m3d::check_launch("linear_forward");
}

/* ---- bf16x3 split variant (see the kernel's header comment): weights are cut once into three bf16 planes ---- */
M3D_API size_t m3d_linear_bf16x3_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 32 != 0) return 0;
  return (size_t)((N + BN - 1) / BN) * (K / 32) * X3_CHUNK_U4 * 16;
}

M3D_API int m3d_linear_bf16x3_pack(const float* d_weight, int N, int K, void* d_packed, void* stream) {
  if (!d_weight || !d_packed || N <= 0 || K <= 0) return M3D_EINVAL;
  if (K % 32 != 0) return M3D_EUNSUPPORTED;
  const int nt = (N + BN - 1) / BN, chunks = K / 32;
  hipLaunchKernelGGL(fc_x3_pack_kernel, dim3(4096), dim3(256), 0, m3d::as_stream(stream), d_weight, N, K, (u32x4*)d_packed, nt, chunks);
  return m3d::check_launch("linear_bf16x3_pack");
}

M3D_API size_t m3d_linear_bf16x3_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0) return 0;
  const int s = x3_plan(M, N, K).slices;
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 16;
}

M3D_API int m3d_linear_bf16x3_forward(const float* d_x, const void* d_packed, const float* d_bias, float* d_out, int M, int N, int K,
                                      int relu, void* d_ws, size_t ws_bytes, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return M3D_EINVAL;
  if (M == 0) return M3D_OK;
  if (!d_x || !d_packed || !d_out) return M3D_EINVAL;
  if (K % 32 != 0 || ((uintptr_t)d_x & 15) || ((uintptr_t)d_packed & 15)) return M3D_EUNSUPPORTED;
  const X3Plan p = x3_plan(M, N, K);
  const int s = p.slices;
  if (s > 1 && (!d_ws || ws_bytes < (size_t)s * M * N * sizeof(float))) return M3D_EWORKSPACE;
  FcX3Args a{d_x, (const u32x4*)d_packed, d_bias, d_out, (float*)d_ws, M, N, K, p.mt, p.nt, s, K / 32, relu, p.per_xcd};
  a.x_alias = m3d::opt(m3d::OPT_TUNE_FC_X_ALIAS) > 0 ? m3d::opt(m3d::OPT_TUNE_FC_X_ALIAS) : 0;     // tuning build only (timing, wrong results)
  a.feat = nullptr; a.taps = nullptr; a.roi_batch = nullptr; a.fC = a.fS = a.fH = a.fW = 0;
  hipStream_t st = m3d::as_stream(stream);
  const size_t lds = sizeof(unsigned) * (3 * (64 * p.wr) * X3_RSW + X3_OPER);
  if (p.wr == 2) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fc_x3_gemm_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fc_x3_gemm_kernel<2>, dim3(8 * a.per_xcd), dim3(256), lds, st, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fc_x3_gemm_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fc_x3_gemm_kernel<4>, dim3(8 * a.per_xcd), dim3(512), lds, st, a);
  }
  if (s > 1) {
    const long long MN = (long long)M * N;
    long long blocks = (MN + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fc_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)d_ws, d_bias, d_out, MN, N, s, s, MN, relu);
  }
  return m3d::check_launch("linear_bf16x3_forward");
}

/* ---- f16x2 split variant (round 6; see the kernel's header comment): two fp16 planes per operand, three products per fp32 product ---- */
M3D_API int m3d_absmax(const float* d_x, long long n, float* d_out, void* stream) {
  if (n < 0 || !d_out || (n > 0 && !d_x)) return M3D_EINVAL;
  hipStream_t st = m3d::as_stream(stream);
  if (hipMemsetAsync(d_out, 0, sizeof(float), st) != hipSuccess) return M3D_ELAUNCH;
  if (n > 0) {
    long long blocks = (n / 4 + 255) / 256 + 1;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_x, n, reinterpret_cast<unsigned*>(d_out));
  }
  return m3d::check_launch("absmax");
}

namespace {
inline size_t f16x2_plane_bytes(int N, int K) { return (size_t)((N + BN - 1) / BN) * (K / 32) * (2 * 128 * 4) * 16; }
}

M3D_API size_t m3d_linear_f16x2_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 32 != 0) return 0;
  return f16x2_plane_bytes(N, K) + 256;                            // + the weight's largest magnitude (one float) behind the planes
}

M3D_API int m3d_linear_f16x2_pack(const float* d_weight, int N, int K, void* d_packed, void* stream) {
  if (!d_weight || !d_packed || N <= 0 || K <= 0) return M3D_EINVAL;
  if (K % 32 != 0 || ((uintptr_t)d_packed & 15)) return M3D_EUNSUPPORTED;
  const int nt = (N + BN - 1) / BN, chunks = K / 32;
  float* wamax = reinterpret_cast<float*>(static_cast<char*>(d_packed) + f16x2_plane_bytes(N, K));
  if (const int rc = m3d_absmax(d_weight, (long long)N * K, wamax, stream)) return rc;
  hipLaunchKernelGGL(fc_f16_pack_kernel, dim3(4096), dim3(256), 0, m3d::as_stream(stream), d_weight, N, K, (u32x4*)d_packed, nt, chunks,
                     (const float*)wamax);
  return m3d::check_launch("linear_f16x2_pack");
}

namespace {
// 256 x 256 tiles (fc_x3b_gemm_kernel<1>) from 384 rows and 256 columns on: 64 KB of loads per chunk and workgroup for 48 MFMAs per wave,
// against 48 KB for 24 on 256 x 128 - the f16x2 GEMM has half the matrix work of bf16x3 per byte, so the bytes per MFMA decide earlier.
// Option tune_fc_x3_rows (tuning build): 128 / 256 force the 128-column kernels, 512 the 256 x 256 tiles.
inline bool f16x2_big_tiles(int M, int N) {
  const int w = m3d::opt(m3d::OPT_TUNE_FC_X3_ROWS);
  if (w == 128 || w == 256) return false;
  if (w == 512) return true;
  return M >= 384 && N >= 256;
}
inline int f16x2_slices(int M, int N, int K) { return f16x2_big_tiles(M, N) ? x3b_slices(M, N, K) : x3_plan(M, N, K).slices; }
}

M3D_API size_t m3d_linear_f16x2_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0) return 0;
  const int s = f16x2_slices(M, N, K);
  return (s > 1 ? (size_t)s * M * N * sizeof(float) : 0) + 256;    // split-K partials + the bound of |x| when the caller gives none
}

M3D_API int m3d_linear_f16x2_forward(const float* d_x, const void* d_packed, const float* d_bias, float* d_out, int M, int N, int K,
                                     int relu, const float* d_x_bound, void* d_ws, size_t ws_bytes, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return M3D_EINVAL;
  if (M == 0) return M3D_OK;
  if (!d_x || !d_packed || !d_out || !d_ws) return M3D_EINVAL;
  if (K % 32 != 0 || ((uintptr_t)d_x & 15) || ((uintptr_t)d_packed & 15) || ((uintptr_t)d_ws & 15)) return M3D_EUNSUPPORTED;
  const bool big = f16x2_big_tiles(M, N);
  X3Plan p = x3_plan(M, N, K);
  if (big) { p.wr = 8; p.mt = (M + 255) / 256; p.nt = (N + 255) / 256; p.slices = x3b_slices(M, N, K); p.per_xcd = (p.mt * p.nt * p.slices + 7) / 8; }
  const int s = p.slices;
  const size_t part_bytes = s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
  if (ws_bytes < part_bytes + 256) return M3D_EWORKSPACE;
  hipStream_t st = m3d::as_stream(stream);
  if (!d_x_bound) {                                               // no bound from the producer: sweep x itself
    float* b = reinterpret_cast<float*>(static_cast<char*>(d_ws) + part_bytes);
    if (const int rc = m3d_absmax(d_x, (long long)M * K, b, stream)) return rc;
    d_x_bound = b;
  }
  FcX3Args a{d_x, (const u32x4*)d_packed, d_bias, d_out, (float*)d_ws, M, N, K, p.mt, p.nt, s, K / 32, relu, p.per_xcd};
  a.xbound = d_x_bound;
  a.wamax = reinterpret_cast<const float*>(static_cast<const char*>(d_packed) + f16x2_plane_bytes(N, K));
  a.x_alias = 0;
  a.feat = nullptr; a.taps = nullptr; a.roi_batch = nullptr; a.fC = a.fS = a.fH = a.fW = 0;
  const size_t lds = big ? sizeof(unsigned) * 2 * 4 * 256 * X3_RSW : sizeof(unsigned) * (2 * (64 * p.wr) * X3_RSW + 2 * X3_PLANE);   // big: two images
  if (big) {
    auto kern = fc_x3b_gemm_kernel<1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(8 * a.per_xcd), dim3(512), lds, st, a);
  } else if (p.wr == 2) {
    auto kern = fc_x3_gemm_kernel<2, 0, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(8 * a.per_xcd), dim3(256), lds, st, a);
  } else {
    auto kern = fc_x3_gemm_kernel<4, 0, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(8 * a.per_xcd), dim3(512), lds, st, a);
  }
  if (s > 1) {
    const long long MN = (long long)M * N;
    long long blocks = (MN + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fc_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)d_ws, d_bias, d_out, MN, N, s, s, MN, relu);
  }
  return m3d::check_launch("linear_f16x2_forward");
}

/* f-1 A/B ("hand-written MFMA GEMM whose A-operand loader performs the RoIAlign gather"): out[M, N] = act(x W^T + b) where
 * x[m, :] = RoIAlign3D(features, roi m) (7^3 bins, sampling grid 2, the reference's memory order) is COMPUTED by the loader from the
 * feature maps and the tap tables of m3d_roi_align3d_tap_tables - the [M, C*343] intermediate never exists.  Same tiles, cut, MFMA loop
 * and split-K as m3d_linear_bf16x3_forward (256 x 128 tiles).  Measured against the two-launch path in profiles/r04_f1_ab.json. */
namespace {
// the plan m3d_linear_bf16x3_roi_forward launches: x3_plan's when it is the 256-row tile (the fused loader's only instance), else a
// 256-row plan of its own - ONE helper for the workspace query and the launch, so the reported bytes cover the `s` that runs
X3Plan x3_roi_plan(int M, int N, int K) {
  X3Plan p = x3_plan(M, N, K);
  if (p.wr != 4) {
    const int mt = (M + 255) / 256, nt = (N + BN - 1) / BN, tiles = mt * nt;
    int s = 256 / tiles; s = s < 1 ? 1 : (s > K / 256 + 1 ? K / 256 + 1 : s);
    p = X3Plan{4, mt, nt, s, (tiles * s + 7) / 8};
  }
  return p;
}
}  // namespace

M3D_API size_t m3d_linear_bf16x3_roi_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0) return 0;
  const X3Plan p = x3_roi_plan(M, N, K);
  return p.slices > 1 ? (size_t)p.slices * M * N * sizeof(float) : 16;
}

M3D_API int m3d_linear_bf16x3_roi_forward(const float* d_features, int batch, int channels, int slices, int height, int width, const void* d_tab,
                                          const int32_t* d_roi_batch, const void* d_packed, const float* d_bias, float* d_out, int M, int N,
                                          int relu, void* d_ws, size_t ws_bytes, void* stream) {
  const int K = channels * 343;
  if (M < 0 || N <= 0 || channels <= 0 || batch <= 0) return M3D_EINVAL;
  if (M == 0) return M3D_OK;
  if (!d_features || !d_tab || !d_roi_batch || !d_packed || !d_out) return M3D_EINVAL;
  if (K % 32 != 0 || ((uintptr_t)d_packed & 15)) return M3D_EUNSUPPORTED;
  const X3Plan p = x3_roi_plan(M, N, K);                           // the fused loader is instantiated for the 256-row tile
  const int s = p.slices;
  if (s > 1 && (!d_ws || ws_bytes < (size_t)s * M * N * sizeof(float))) return M3D_EWORKSPACE;
  FcX3Args a{nullptr, (const u32x4*)d_packed, d_bias, d_out, (float*)d_ws, M, N, K, p.mt, p.nt, s, K / 32, relu, p.per_xcd};
  a.x_alias = 0; a.feat = d_features; a.taps = (const int4*)d_tab; a.roi_batch = d_roi_batch;
  a.fC = channels; a.fS = slices; a.fH = height; a.fW = width;
  hipStream_t st = m3d::as_stream(stream);
  const size_t lds = sizeof(unsigned) * (3 * 256 * X3_RSW + X3_OPER);
  auto kern = fc_x3_gemm_kernel<4, 1>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(8 * a.per_xcd), dim3(512), lds, st, a);
  if (s > 1) {
    const long long MN = (long long)M * N;
    long long blocks = (MN + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fc_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)d_ws, d_bias, d_out, MN, N, s, s, MN, relu);
  }
  return m3d::check_launch("linear_bf16x3_roi_forward");
}

/* 256 x 256 tiles on the nn.Linear weight itself (fp32, cut in the kernel): the variant for many rows, no packed copy */
M3D_API size_t m3d_linear_bf16x3_w32_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0) return 0;
  const int s = x3b_slices(M, N, K);
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 16;
}

M3D_API int m3d_linear_bf16x3_w32_forward(const float* d_x, const float* d_weight, const float* d_bias, float* d_out, int M, int N, int K,
                                          int relu, void* d_ws, size_t ws_bytes, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return M3D_EINVAL;
  if (M == 0) return M3D_OK;
  if (!d_x || !d_weight || !d_out) return M3D_EINVAL;
  if (K % 32 != 0 || ((uintptr_t)d_x & 15) || ((uintptr_t)d_weight & 15)) return M3D_EUNSUPPORTED;
  const int s = x3b_slices(M, N, K);
  if (s > 1 && (!d_ws || ws_bytes < (size_t)s * M * N * sizeof(float))) return M3D_EWORKSPACE;
  FcX3Args a{d_x, (const u32x4*)d_weight, d_bias, d_out, (float*)d_ws, M, N, K, (M + 255) / 256, (N + 255) / 256, s, K / 32, relu, 0};
  a.per_xcd = (a.mt * a.nt * s + 7) / 8;
  a.x_alias = 0; a.feat = nullptr; a.taps = nullptr; a.roi_batch = nullptr; a.fC = a.fS = a.fH = a.fW = 0;
  hipStream_t st = m3d::as_stream(stream);
  const size_t lds = sizeof(unsigned) * 6 * 256 * X3_RSW;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fc_x3b_gemm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(fc_x3b_gemm_kernel<0>, dim3(8 * a.per_xcd), dim3(512), lds, st, a);
  if (s > 1) {
    const long long MN = (long long)M * N;
    long long blocks = (MN + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fc_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)d_ws, d_bias, d_out, MN, N, s, s, MN, relu);
  }
  return m3d::check_launch("linear_bf16x3_w32_forward");
}
