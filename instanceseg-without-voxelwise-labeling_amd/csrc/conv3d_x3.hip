// 3x3x3 "same" convolution at fp32 accuracy on the bf16 matrix cores: the direct (implicit-GEMM) kernel for the convolutions that must
// stay EXACT sums of products - the norm convolutions of the peak back-propagation, N = conv3d(X - min X, relu(W))
// (lib/prm/peak_backprop_3d.py:37-44: N is a divisor and its exact zeros gate the `N < 1e-10` test of :30-33, so Winograd is out).
//
// Arithmetic: the bf16x3 split of fc_gemm.hip.  Both fp32 operands are cut by truncation into three bf16 pieces (x = xh + xm + xl,
// exact); six bf16 products xh.wh + xh.wm + xm.wh + xm.wm + xh.wl + xl.wh accumulate in fp32 what one fp32 MFMA step does (the three
// dropped products are below 2^-23 of the product).  For the norm convolutions every operand is >= 0, so every piece and every
// product is >= 0: a sum is zero exactly when every product in it is, as with the fp32 kernel.  v_mfma_f32_32x32x16_bf16 runs at 16 x
// the fp32 MFMA's rate: 16 / 6 = 2.7 x the direct fp32 kernel's ceiling.
//
// GEMM view: M = output channels (A = weights, cut and packed once: m3d_conv3d_x3_pack), N = voxels (B = the input tile), K = 16 input
// channels per MFMA, per tap.  A workgroup (4 waves) owns 64 output channels x a 16 x 4 x 4 (x, y, z) voxel tile; wave w owns z plane w:
// 64 channels x 64 voxels = 2 x 2 blocks of 32 x 32 (64 accumulator registers).  Per 16-channel chunk the halo tile (18 x 6 x 6
// voxels) is loaded once, cut on the way (x - offset inside the volume, 0 outside: the zero padding applies to the shifted input) and
// serves all 27 taps; the tap's weights (64 x 16 x 3 pieces = 6 KB) stream through three LDS slots, one barrier per tap; a step's
// fragments are read during the previous step's MFMAs.
// LDS (16-byte units = 8 bf16 along K): input [piece][k half][halo voxel], weights [buffer][piece][k half][row]: a fragment read is
// one ds_read_b128 per lane at consecutive units - the lane -> voxel map below follows the lane groups ds_read_b128 is served in
// ({0-3,12-15,20-27} / {4-11,16-19,28-31}, fc_gemm.hip), 16 consecutive x per group: no bank conflicts.  80.6 KB: two workgroups per CU.
// Roofline: MFMA (bf16, 2.5 PFLOP/s dense): 6 x 2 x 27 x Cin x Cout issued FLOP per voxel.
#include <type_traits>

#include "m3d_common.h"

// timing-only ablation builds (make x3_ablate, tools/bench_x3.py; WRONG results): 1 = no MFMAs, 2 = no fragment reads, 4 = no weight
// staging (loads, LDS writes), 8 = no per-step barrier, 16 = no flush
#ifndef X3_EXP
#define X3_EXP 0
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TX = 16, TY = 4, TZ = 4, HX = TX + 2, HY = TY + 2, HZ = TZ + 2, NHV = HX * HY * HZ;   // 648 halo voxels
constexpr int TM = 64;                                  // output channels per workgroup
constexpr int NT = 256;
// NPL = pieces per operand: 3 (bf16x3: exact three-way bf16 cut, six products) or 2 (round 6, "f16x2": both operands scaled by a power of
// two and cut into two fp16 numbers, 22 bits, three products - fc_gemm.hip; conv3d_x3_kernel<.., 1>)
constexpr int in_units(int npl) { return 2 * npl * NHV; }                 // [piece][k half 2][halo voxel]
constexpr int w_units(int npl) { return 2 * npl * TM; }                  // one tap: [piece][k half 2][row 64]
constexpr int IN_UNITS = in_units(3);
constexpr int W_UNITS = w_units(3);
constexpr int W_SLOTS = 3;
constexpr int lds_bytes(int npl) { return (in_units(npl) + W_SLOTS * w_units(npl)) * 16; }
constexpr int LDS_BYTES = lds_bytes(3);                                   // 80 640 B: two workgroups per CU (161 280 of 163 840); f16x2: 53 760 B
constexpr int NTASK = (2 * NHV + NT - 1) / NT;          // staging tasks per thread and chunk: (halo voxel, 8-channel group)
constexpr int acc_row(int g) { return (g & 3) + 8 * (g >> 2); }      // accumulator register g -> row of its 32 x 32 block (+ 4 * (lane >> 5))

// packed[cout tile][step = chunk * 27 + tap][piece][k half][row 64] units of 8 bf16 <- weight [cout][cin][27] fp32 (relu: relu(W) first)
__global__ __launch_bounds__(256) void conv3d_x3_pack_kernel(const float* __restrict__ w, int cin, int cout, int relu, u32x4* __restrict__ packed) {
  const int chunks = cin / 16, ntile = (cout + TM - 1) / TM;
  const long long total = (long long)ntile * chunks * 27 * 2 * TM;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int row = (int)(e % TM);
    long long r = e / TM;
    const int kh = (int)(r & 1); r >>= 1;
    const int tap = (int)(r % 27); r /= 27;
    const int ch = (int)(r % chunks), ct = (int)(r / chunks);
    const int co = ct * TM + row;
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = co < cout ? w[((size_t)co * cin + ch * 16 + 8 * kh + j) * 27 + tap] : 0.f;
      if (relu && !(v > 0.f)) v = 0.f;                                    // peak_backprop_3d.py:41
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
    u32x4* dst = packed + ((size_t)(ct * chunks + ch) * 27 + tap) * W_UNITS + kh * TM + row;
    dst[0] = ph; dst[2 * TM] = pm; dst[4 * TM] = pl;
  }
}

// f16x2: packed[cout tile][step][piece 2][k half][row 64] units of 8 fp16 <- weight * s (s from the largest |W| / relu(W): *wamax)
__global__ __launch_bounds__(256) void conv3d_x3f_pack_kernel(const float* __restrict__ w, int cin, int cout, int relu, u32x4* __restrict__ packed,
                                                              const float* __restrict__ wamax) {
  float sw, inv;
  m3d::f16_scale_of(*wamax, sw, inv);
  const int chunks = cin / 16, ntile = (cout + TM - 1) / TM;
  const long long total = (long long)ntile * chunks * 27 * 2 * TM;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int row = (int)(e % TM);
    long long r = e / TM;
    const int kh = (int)(r & 1); r >>= 1;
    const int tap = (int)(r % 27); r /= 27;
    const int ch = (int)(r % chunks), ct = (int)(r / chunks);
    const int co = ct * TM + row;
    u32x4 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16x2 hh, ll;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float v = co < cout ? w[((size_t)co * cin + ch * 16 + 8 * kh + 2 * j + u) * 27 + tap] : 0.f;
        if (relu && !(v > 0.f)) v = 0.f;
        v *= sw;
        const _Float16 h = (_Float16)v;
        hh[u] = h; ll[u] = (_Float16)(v - (float)h);
      }
      ph[j] = __builtin_bit_cast(unsigned, hh); pl[j] = __builtin_bit_cast(unsigned, ll);
    }
    u32x4* dst = packed + ((size_t)(ct * chunks + ch) * 27 + tap) * w_units(2) + kh * TM + row;
    dst[0] = ph; dst[2 * TM] = pl;
  }
}

// max |w| (relu: max of the positive entries) of a weight tensor; one workgroup, once per model
__global__ __launch_bounds__(1024) void x3f_wamax_kernel(const float* __restrict__ w, long long n, int relu, float* __restrict__ out) {
  unsigned m = 0;
  for (long long e = threadIdx.x; e < n; e += 1024) {
    const float v = w[e];
    const unsigned b = relu ? (v > 0.f ? __float_as_uint(v) : 0u) : (__float_as_uint(v) & 0x7FFFFFFFu);
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
  __shared__ unsigned wm[16];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) { for (int k = 1; k < 16; ++k) m = wm[k] > m ? wm[k] : m; *out = __uint_as_float(m); }
}

struct X3Args {
  const float* x; const u32x4* wp; float* out; const float* in_off;
  const float* in_max; const float* wamax;   // F16: max x (with in_off: the bound of x - off is *in_max - *in_off; without: max |x|), max |W|
  int B, cin, cout, D, H, W;
  int tx, ty, tz, ntile, per_xcd, units;
  int ksplit;              // > 1: the input channels are cut into `ksplit` ranges of whole chunk pairs, one workgroup each; range k writes its
  float* part;             // partial sums to part + k * B * cout * D * H * W and conv3d_x3_reduce_kernel adds them in range order
};

// column (lane % 32) of a 32-voxel block -> (x, y) inside its 16 x 2 patch
__device__ __forceinline__ void col_xy(int c, int* x, int* y) {
  if (c < 4) { *x = c; *y = 0; }
  else if (c < 12) { *x = c - 4; *y = 1; }
  else if (c < 16) { *x = c - 8; *y = 0; }
  else if (c < 20) { *x = c - 8; *y = 1; }
  else if (c < 28) { *x = c - 12; *y = 0; }
  else { *x = c - 16; *y = 1; }
}

// SPLIT: the launch cuts K into a.ksplit ranges (small maps); the unsplit instantiation keeps its compile-time zero range start
template <bool SPLIT, int F16 = 0>
__global__ __launch_bounds__(NT, 2) void conv3d_x3_kernel(X3Args a) {
  constexpr int NPL = F16 ? 2 : 3;
  constexpr int IN_UNITS = in_units(NPL), W_UNITS = w_units(NPL);         // (shadow the bf16x3 constants of the file scope)
  using frag_t = std::conditional_t<F16 != 0, f16x8, bf16x8>;
  extern __shared__ float lds_f[];
  u32x4* const lds = reinterpret_cast<u32x4*>(lds_f);              // [input: 6 * NHV][weights: 2 * 6 * TM] units
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // unit = (batch, spatial tile, cout tile), cout tile fastest: the workgroups one XCD runs together share their input tiles in its L2
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int u = xcd * a.per_xcd + idx;
  if (idx >= a.per_xcd || u >= a.units) return;
  const int ks = SPLIT ? u % a.ksplit : 0;                         // K range fastest, then the cout tile: neighbours share the input tile in L2
  const int ct = SPLIT ? (u / a.ksplit) % a.ntile : u % a.ntile;
  int sp = SPLIT ? u / (a.ksplit * a.ntile) : u / a.ntile;
  const int bx = sp % a.tx; sp /= a.tx;
  const int by = sp % a.ty; sp /= a.ty;
  const int bz = sp % a.tz;
  const int b = sp / a.tz;
  const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
  const size_t HW = (size_t)a.H * a.W, DHW = HW * a.D;
  const int pairs = (a.cin / 16 + 1) / 2;                          // ranges start on even chunks: the parity-unrolled chunk bodies below
  const int cb0 = SPLIT ? 2 * (int)((long long)ks * pairs / a.ksplit) : 0;
  const int chunks = SPLIT ? min(a.cin / 16, 2 * (int)((long long)(ks + 1) * pairs / a.ksplit)) : a.cin / 16;
  const int steps = chunks * 27;                                   // (names as in the unsplit kernel: `chunks` / `steps` = END of this range)
  const float off = a.in_off ? *a.in_off : 0.f;
  float xs = 1.f, inv_x = 1.f, inv_w = 1.f;                        // F16: operand scales (powers of two) and what undoes them at the flush
  if constexpr (F16) {
    float sw_;
    m3d::f16_scale_of(*a.in_max - off, xs, inv_x);
    m3d::f16_scale_of(*a.wamax, sw_, inv_w);
  }

  // ---- staging tasks of this thread: task = tid + NT * i -> (8-channel group g = task / NHV, halo voxel hv = task % NHV)
  int t_src[NTASK];                                               // offset of the voxel inside one channel map, or -1 (outside the volume)
  int t_dst[NTASK];                                               // LDS unit of piece 0 (-1: no task)
#pragma unroll
  for (int i = 0; i < NTASK; ++i) {
    const int task = tid + NT * i;
    if (task < 2 * NHV) {
      const int g = task >= NHV ? 1 : 0, hv = task - g * NHV;
      const int hx = hv % HX, r = hv / HX, hy = r % HY, hz = r / HY;
      const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool in = (z >= 0) & (z < a.D) & (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
      t_src[i] = in ? (int)(((size_t)z * a.H + y) * a.W + x) + g * 8 * (int)DHW : -1;
      t_dst[i] = g * NHV + hv;
    } else {
      t_src[i] = -1; t_dst[i] = -1;
    }
  }
  const float* const xb = a.x + (size_t)b * a.cin * DHW;
  float sx[NTASK][8];
  auto fetch_in = [&](int c) __attribute__((always_inline)) {
    const float* base = xb + (size_t)c * 16 * DHW;
#pragma unroll
    for (int i = 0; i < NTASK; ++i) {
      const float* p = base + (t_src[i] >= 0 ? t_src[i] : 0);       // loads only: arithmetic on a loaded value would wait for it here
#pragma unroll
      for (int j = 0; j < 8; ++j) sx[i][j] = p[(size_t)j * DHW];
    }
  };
  auto commit_in = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NTASK; ++i) {
      if (t_dst[i] < 0) continue;
      const bool in = t_src[i] >= 0;                                // outside the volume: the zero padding of the SHIFTED input
      u32x4 ph, pm, pl;
      if constexpr (F16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0 = in ? (sx[i][2 * j] - off) * xs : 0.f, v1 = in ? (sx[i][2 * j + 1] - off) * xs : 0.f;
          const f16x2 hh = {(_Float16)v0, (_Float16)v1};
          const f16x2 ll = {(_Float16)(v0 - (float)hh[0]), (_Float16)(v1 - (float)hh[1])};
          ph[j] = __builtin_bit_cast(unsigned, hh); pl[j] = __builtin_bit_cast(unsigned, ll);
        }
        lds[t_dst[i]] = ph;
        lds[t_dst[i] + 2 * NHV] = pl;
        continue;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = in ? sx[i][2 * j] - off : 0.f, v1 = in ? sx[i][2 * j + 1] - off : 0.f;
        const unsigned h0 = __float_as_uint(v0) & 0xFFFF0000u, h1 = __float_as_uint(v1) & 0xFFFF0000u;
        const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
        const unsigned q0 = __float_as_uint(r0) & 0xFFFF0000u, q1 = __float_as_uint(r1) & 0xFFFF0000u;
        const float s0 = r0 - __uint_as_float(q0), s1 = r1 - __uint_as_float(q1);
        ph[j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
        pm[j] = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
        pl[j] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
      }
      lds[t_dst[i]] = ph;
      lds[t_dst[i] + 2 * NHV] = pm;
      lds[t_dst[i] + 4 * NHV] = pl;
    }
  };
  // ---- weights: one tap = 384 units; thread -> unit tid (+ unit 256 + tid for tid < 128)
  const u32x4* const wsrc = a.wp + (size_t)ct * (a.cin / 16) * 27 * W_UNITS;      // the tile's steps of ALL chunks (`steps` ends this range)
  // two register sets, one per step parity, and three LDS slots (slot of step s = tap % 3; 27 taps: the same in every chunk): the weights
  // of step s + 4 are requested when step s ends, written to LDS when step s + 2 ends and first read (prefetched) during step s + 3
  u32x4 sw[2][2];
  auto fetch_w = [&](int s, int set) __attribute__((always_inline)) {
    const u32x4* p = wsrc + (size_t)s * W_UNITS;
    sw[set][0] = p[tid];
    if constexpr (W_UNITS > NT) { if (tid < W_UNITS - NT) sw[set][1] = p[NT + tid]; }
  };
  auto commit_w = [&](int set, int slot) __attribute__((always_inline)) {
    u32x4* d = lds + IN_UNITS + slot * W_UNITS;
    d[tid] = sw[set][0];
    if constexpr (W_UNITS > NT) { if (tid < W_UNITS - NT) d[NT + tid] = sw[set][1]; }
  };

  // ---- fragments
  const int fr = lane & 31, fh = lane >> 5;
  int cx, cy;
  col_xy(fr, &cx, &cy);
  const int hvB = (wave * HY + cy) * HX + cx + fh * NHV;           // + (2 * cb + dy) * HX + dz * HY * HX + dx + piece * 2 * NHV
  const int rowA = IN_UNITS + fh * TM + fr;                        // + slot * W_UNITS + piece * 2 * TM + 32 * rb
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  // ---- flush: the accumulators of ONE 16-channel chunk (27 taps x 6 products = 162 accumulation steps) are added to the output tile with
  // ordinary round-to-nearest adds.  v_mfma_f32_32x32x16_bf16 TRUNCATES when it adds its products to the accumulator: with non-negative
  // operands (the norm conv) a sum kept in the accumulator over all of K drifts low by ~3e-9 of itself per step - 1.6e-5 at K = 256 x 27,
  // eight times the fp32 kernel's error.  Restarting the run every second chunk (324 steps) bounds the drift by the run's length (3e-7); the read-modify-write
  // of the 64 KB output tile per chunk stays in L2.
  const int z = z0 + wave;
  float* const ob = (SPLIT ? a.part + (size_t)ks * a.B * a.cout * DHW : a.out) + ((size_t)b * a.cout + ct * TM) * DHW;
  auto flush = [&](bool first) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int y = y0 + 2 * cb + cy, x = x0 + cx;
      const bool vok = (z < a.D) & (y < a.H) & (x < a.W);
      size_t vo = ((size_t)(vok ? z : 0) * a.H + (vok ? y : 0)) * a.W + (vok ? x : 0) + (size_t)(4 * fh) * DHW;
      asm volatile("" : "+v"(vo));                                   // not loop-invariant: 32 hoisted row addresses would cost 64 registers in the tap loop
      float* const o = ob + vo;
      float v[2][16];
      if (!first) {                                                  // all 32 loads of the block column in flight together (the staging registers are free)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const bool ok = vok & (ct * TM + 32 * rb + 4 * fh + acc_row(g) < a.cout);
            v[rb][g] = ok ? o[(size_t)(32 * rb + acc_row(g)) * DHW] : 0.f;
          }
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const bool ok = vok & (ct * TM + 32 * rb + 4 * fh + acc_row(g) < a.cout);
          if (ok) {
            float av = acc[rb][cb][g];
            if constexpr (F16) av = (av * inv_x) * inv_w;            // powers of two: exact
            o[(size_t)(32 * rb + acc_row(g)) * DHW] = first ? av : v[rb][g] + av;
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
  };

  struct Frags { frag_t a[2][NPL], b[2][NPL]; };
  Frags F0, F1;                                                    // fragments of the even / odd steps
  constexpr int NFR = 4 * NPL;                                     // fragment reads per step
  // fragment read i of NFR: i = 4 * piece + {a[0], a[1], b[0], b[1]}
  auto read_frag = [&](Frags& f, int i, int t, int slot) __attribute__((always_inline)) {
    const int p = i >> 2, k = i & 3;
    const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
    if (k < 2) f.a[k][p] = __builtin_bit_cast(frag_t, lds[(X3_EXP & 2) ? rowA : rowA + slot * W_UNITS + p * 2 * TM + 32 * k]);
    else f.b[k - 2][p] = __builtin_bit_cast(frag_t, lds[(X3_EXP & 2) ? hvB : hvB + (dz * HY + 2 * (k - 2) + dy) * HX + dx + p * 2 * NHV]);
  };

  const int sb0 = cb0 * 27;                                        // first step of the range (even: register set 0, LDS slot 0)
  fetch_in(cb0);
  fetch_w(sb0, 0);
  if (sb0 + 1 < steps) fetch_w(sb0 + 1, 1);
  commit_in();
  commit_w(0, 0);
  if (sb0 + 1 < steps) commit_w(1, 1);
  __syncthreads();
  if (sb0 + 2 < steps) fetch_w(sb0 + 2, 0);
  if (sb0 + 3 < steps) fetch_w(sb0 + 3, 1);
  if (cb0 + 1 < chunks) fetch_in(cb0 + 1);
#pragma unroll
  for (int i = 0; i < NFR; ++i) read_frag(F0, i, 0, 0);

  // one chunk = 27 steps; `par` = parity of its first step (27 is odd: the parity of the chunk index), a literal at both call sites so that
  // register-set choices fold to constants in the unrolled taps.  During the 24 MFMAs of a step the 12 fragments of the NEXT step are read,
  // two ds_read_b128 per four MFMAs (the LDS array keeps up with that rate beside the matrix pipe; reads bunched in front of the MFMAs
  // that need them cost 40 % of the kernel: tools/x3_ablate.sh), except over a chunk boundary, where the next input tile is not in LDS yet.
  auto run_chunk = [&](int c, const int par) __attribute__((always_inline)) {
    const int s0 = c * 27;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int ps = (par + t) & 1, s = s0 + t;
      Frags& cur = ps ? F1 : F0;
      Frags& nxt = ps ? F0 : F1;
      // small products first: bf16x3 (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); f16x2 (l,h) (h,l) (h,h)
      constexpr int NPR = F16 ? 3 : 6;
      constexpr int PA[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0}, PB[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
      constexpr int RD0[7] = {0, F16 ? 3 : 2, F16 ? 6 : 4, F16 ? 8 : 6, 8, 10, 12};      // reads [RD0[q], RD0[q + 1]) of the next step ride on product q
#pragma unroll
      for (int q = 0; q < NPR; ++q) {
        if (t < 26) {
#pragma unroll
          for (int i = RD0[q]; i < RD0[q + 1]; ++i) read_frag(nxt, i, t + 1, (t + 1) % 3);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            if constexpr (F16) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[rb][PA[q]], cur.b[cb][PB[q]], acc[rb][cb], 0, 0, 0);
            else if (!(X3_EXP & 1)) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[rb][PA[q]], cur.b[cb][PB[q]], acc[rb][cb], 0, 0, 0);
            else acc[rb][cb][q] += (float)cur.a[rb][PA[q]][0] * (float)cur.b[cb][PB[q]][0];
          }
      }
      if constexpr (F16) {                                          // pin the interleaving: 3 / 3 / 2 LDS reads, 4 MFMAs each
        if (t < 26) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (t < 26) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (t < 26) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      } else {
#pragma unroll
        for (int q = 0; q < NPR; ++q) {                             // pin the interleaving: 2 LDS reads, 4 MFMAs
          if (t < 26) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
      }
      if (!(X3_EXP & 4) && s + 2 < steps) commit_w(ps, (t + 2) % 3); // the slot step s - 1 used: its fragments were read during step s - 2
      if (!(X3_EXP & 8)) __syncthreads();
      if (!(X3_EXP & 4) && s + 4 < steps) fetch_w(s + 4, ps);
    }
    if (c + 1 < chunks) {
      commit_in();                                                  // the chunk's last fragments were read during tap 25: two barriers ago
      __syncthreads();
    }
    // the accumulators restart every second chunk (see `flush`): runs of at most 324 accumulation steps
    if ((c & 1) || c + 1 == chunks) { if (!(X3_EXP & 16) || c + 1 == chunks) flush(c < cb0 + 2); }
    if (c + 2 < chunks) fetch_in(c + 2);
    if (c + 1 < chunks) {
      Frags& first = par ? F0 : F1;                                 // parity of the next chunk's first step
#pragma unroll
      for (int i = 0; i < NFR; ++i) read_frag(first, i, 0, 0);
    }
  };
#pragma unroll 1
  for (int c = cb0; c < chunks; c += 2) {
    run_chunk(c, 0);
    if (c + 1 < chunks) run_chunk(c + 1, 1);
  }
}

// out[e] = part[0][e] + part[1][e] + ... in range order (deterministic); n4 = elements / 4
__global__ __launch_bounds__(256) void conv3d_x3_reduce_kernel(const float4* __restrict__ part, float4* __restrict__ out, long long n4, int ksplit) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
    float4 s = part[e];
    for (int k = 1; k < ksplit; ++k) {
      const float4 v = part[(size_t)k * n4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[e] = s;
  }
}

}  // namespace

M3D_API size_t m3d_conv3d_x3_packed_bytes(int cin, int cout) {
  if (cin <= 0 || cout <= 0 || cin % 16) return 0;
  return (size_t)((cout + TM - 1) / TM) * (cin / 16) * 27 * W_UNITS * 16;
}

M3D_API int m3d_conv3d_x3_supported(int cin, int cout) { return (cin > 0 && cout > 0 && cin % 16 == 0) ? 1 : 0; }

M3D_API int m3d_conv3d_x3_pack(const float* d_weight, int cin, int cout, int relu_weights, void* d_packed, void* stream) {
  if (!d_weight || !d_packed || cin <= 0 || cout <= 0) return M3D_EINVAL;
  if (cin % 16) return M3D_EUNSUPPORTED;
  const long long total = (long long)((cout + TM - 1) / TM) * (cin / 16) * 27 * 2 * TM;
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(conv3d_x3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, m3d::as_stream(stream), d_weight, cin, cout,
                     relu_weights ? 1 : 0, reinterpret_cast<u32x4*>(d_packed));
  return m3d::check_launch("conv3d_x3_pack");
}

namespace {
// K ranges for a launch: none while the (spatial tile, cout tile) units give most CUs a workgroup; below 192 units up to 4 ranges of whole
// chunk pairs (the nuclei net's 8 x 25 x 25 maps: 112 units -> 4 ranges, 0.313 -> 0.164 ms; a 16 x 40 x 40 map's 240 units gained 2-6 % in
// isolation and nothing in the tile - the partial sums and the reduce launch cost what the second range saves - and stays whole)
int x3_ksplit(long long units, int cin, size_t out_elems) {
  const int pairs = (cin / 16 + 1) / 2;
  int s = 1;
  while (units < 192 && s < 4 && units * s < 384 && 2 * s <= pairs) s *= 2;
  return (out_elems % 4 == 0) ? s : 1;                             // the reduce kernel adds float4s
}
}  // namespace

M3D_API size_t m3d_conv3d_x3_workspace_bytes(int batch, int cin, int cout, int depth, int height, int width) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0 || cin % 16) return 0;
  const long long units = (long long)batch * ((width + TX - 1) / TX) * ((height + TY - 1) / TY) * ((depth + TZ - 1) / TZ) * ((cout + TM - 1) / TM);
  const size_t elems = (size_t)batch * cout * depth * height * width;
  const int s = x3_ksplit(units, cin, elems);
  return s > 1 ? (size_t)s * elems * sizeof(float) : 0;
}

M3D_API long long m3d_conv3d_x3_launch_units(int batch, int cin, int cout, int depth, int height, int width) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0 || cin % 16) return 0;
  const long long units = (long long)batch * ((width + TX - 1) / TX) * ((height + TY - 1) / TY) * ((depth + TZ - 1) / TZ) * ((cout + TM - 1) / TM);
  return units * x3_ksplit(units, cin, (size_t)batch * cout * depth * height * width);
}

/* ---- f16x2 variant (round 6): two scaled fp16 pieces per operand, three products; see the kernel's template parameter ---- */
namespace {
inline size_t x3f_plane_bytes(int cin, int cout) { return (size_t)((cout + TM - 1) / TM) * (cin / 16) * 27 * w_units(2) * 16; }
}

M3D_API size_t m3d_conv3d_x3f_packed_bytes(int cin, int cout) {
  if (cin <= 0 || cout <= 0 || cin % 16) return 0;
  return x3f_plane_bytes(cin, cout) + 256;                        // + the largest |W| behind the planes
}

M3D_API int m3d_conv3d_x3f_pack(const float* d_weight, int cin, int cout, int relu_weights, void* d_packed, void* stream) {
  if (!d_weight || !d_packed || cin <= 0 || cout <= 0) return M3D_EINVAL;
  if (cin % 16 || ((uintptr_t)d_packed & 15)) return M3D_EUNSUPPORTED;
  hipStream_t st = m3d::as_stream(stream);
  float* wamax = reinterpret_cast<float*>(static_cast<char*>(d_packed) + x3f_plane_bytes(cin, cout));
  hipLaunchKernelGGL(x3f_wamax_kernel, dim3(1), dim3(1024), 0, st, d_weight, (long long)cout * cin * 27, relu_weights ? 1 : 0, wamax);
  const long long total = (long long)((cout + TM - 1) / TM) * (cin / 16) * 27 * 2 * TM;
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(conv3d_x3f_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_weight, cin, cout, relu_weights ? 1 : 0,
                     reinterpret_cast<u32x4*>(d_packed), (const float*)wamax);
  return m3d::check_launch("conv3d_x3f_pack");
}

/* d_in_max: device pointer to max(x) when d_in_offset is given (the operand x - offset is then bounded by max - offset), to max |x| when it
 * is not.  A bound that is too small overflows fp16 (inf / NaN); one up to 2^8 too large costs no accuracy.  Workspace, K split and launch
 * geometry as m3d_conv3d_x3_forward_ws (m3d_conv3d_x3_workspace_bytes / _launch_units). */
M3D_API int m3d_conv3d_x3f_forward_ws(const float* d_x, const void* d_packed, float* d_out, int batch, int cin, int cout, int depth, int height,
                                      int width, const float* d_in_offset, const float* d_in_max, void* d_workspace, size_t workspace_bytes,
                                      void* stream) {
  if (batch < 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (batch == 0) return M3D_OK;
  if (!d_x || !d_packed || !d_out || !d_in_max) return M3D_EINVAL;
  if (cin % 16) return M3D_EUNSUPPORTED;
  if ((size_t)cin * depth * height * width >= 0x7FFFFFFFull / 4) return M3D_EUNSUPPORTED;       // 32-bit offsets inside one item
  X3Args a;
  a.x = d_x; a.wp = reinterpret_cast<const u32x4*>(d_packed); a.out = d_out; a.in_off = d_in_offset; a.in_max = d_in_max;
  a.wamax = reinterpret_cast<const float*>(static_cast<const char*>(d_packed) + x3f_plane_bytes(cin, cout));
  a.B = batch; a.cin = cin; a.cout = cout; a.D = depth; a.H = height; a.W = width;
  a.tx = (width + TX - 1) / TX; a.ty = (height + TY - 1) / TY; a.tz = (depth + TZ - 1) / TZ; a.ntile = (cout + TM - 1) / TM;
  long long units = (long long)batch * a.tx * a.ty * a.tz * a.ntile;
  const size_t elems = (size_t)batch * cout * depth * height * width;
  a.ksplit = x3_ksplit(units, cin, elems);
  if (a.ksplit > 1 && (!d_workspace || workspace_bytes < (size_t)a.ksplit * elems * sizeof(float))) a.ksplit = 1;
  a.part = a.ksplit > 1 ? reinterpret_cast<float*>(d_workspace) : nullptr;
  units *= a.ksplit;
  if (units > 0x3FFFFFFFll) return M3D_EUNSUPPORTED;
  a.units = (int)units; a.per_xcd = (int)((units + 7) / 8);
  auto launch = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(2));
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * a.per_xcd)), dim3(NT), lds_bytes(2), m3d::as_stream(stream), a);
  };
  if (a.ksplit > 1) launch(conv3d_x3_kernel<true, 1>);
  else launch(conv3d_x3_kernel<false, 1>);
  if (a.ksplit > 1) {
    const long long n4 = (long long)(elems / 4);
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3d_x3_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, m3d::as_stream(stream),
                       reinterpret_cast<const float4*>(a.part), reinterpret_cast<float4*>(d_out), n4, a.ksplit);
  }
  return m3d::check_launch("conv3d_x3f_forward");
}

M3D_API int m3d_conv3d_x3_forward(const float* d_x, const void* d_packed, float* d_out, int batch, int cin, int cout, int depth, int height,
                                  int width, const float* d_in_offset, void* stream) {
  return m3d_conv3d_x3_forward_ws(d_x, d_packed, d_out, batch, cin, cout, depth, height, width, d_in_offset, nullptr, 0, stream);
}

/* With a workspace of m3d_conv3d_x3_workspace_bytes the launch may cut K into ranges (small maps: more workgroups, partial sums added in
 * range order by a second kernel - deterministic); without one (null / too small) every workgroup runs all of K. */
M3D_API int m3d_conv3d_x3_forward_ws(const float* d_x, const void* d_packed, float* d_out, int batch, int cin, int cout, int depth, int height,
                                     int width, const float* d_in_offset, void* d_workspace, size_t workspace_bytes, void* stream) {
  if (batch < 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (batch == 0) return M3D_OK;
  if (!d_x || !d_packed || !d_out) return M3D_EINVAL;
  if (cin % 16) return M3D_EUNSUPPORTED;
  if ((size_t)cin * depth * height * width >= 0x7FFFFFFFull / 4) return M3D_EUNSUPPORTED;       // 32-bit offsets inside one item
  X3Args a;
  a.x = d_x; a.wp = reinterpret_cast<const u32x4*>(d_packed); a.out = d_out; a.in_off = d_in_offset; a.in_max = nullptr; a.wamax = nullptr;
  a.B = batch; a.cin = cin; a.cout = cout; a.D = depth; a.H = height; a.W = width;
  a.tx = (width + TX - 1) / TX; a.ty = (height + TY - 1) / TY; a.tz = (depth + TZ - 1) / TZ; a.ntile = (cout + TM - 1) / TM;
  long long units = (long long)batch * a.tx * a.ty * a.tz * a.ntile;
  const size_t elems = (size_t)batch * cout * depth * height * width;
  a.ksplit = x3_ksplit(units, cin, elems);
  if (a.ksplit > 1 && (!d_workspace || workspace_bytes < (size_t)a.ksplit * elems * sizeof(float))) a.ksplit = 1;
  a.part = a.ksplit > 1 ? reinterpret_cast<float*>(d_workspace) : nullptr;
  units *= a.ksplit;
  if (units > 0x3FFFFFFFll) return M3D_EUNSUPPORTED;
  a.units = (int)units; a.per_xcd = (int)((units + 7) / 8);
  auto launch = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * a.per_xcd)), dim3(NT), LDS_BYTES, m3d::as_stream(stream), a);
  };
  if (a.ksplit > 1) launch(conv3d_x3_kernel<true>);
  else launch(conv3d_x3_kernel<false>);
  if (a.ksplit > 1) {
    const long long n4 = (long long)(elems / 4);
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3d_x3_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, m3d::as_stream(stream),
                       reinterpret_cast<const float4*>(a.part), reinterpret_cast<float4*>(d_out), n4, a.ksplit);
  }
  return m3d::check_launch("conv3d_x3_forward");
}
