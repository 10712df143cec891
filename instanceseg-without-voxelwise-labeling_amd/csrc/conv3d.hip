// 3D convolution for gfx950 as an fp32 MFMA implicit GEMM (no im2col buffer).
//
// What it replaces: the cuDNN-backed torch.nn.Conv3d calls of lib/modeling/DSN.py:19-36,57-68 and
// lib/modeling/rpn_heads.py:54-61,94-98 (forward), and — with M3D_W_RELU / M3D_W_DGRAD_RELU weight packs —
// the norm conv and the backward-data conv of lib/prm/peak_backprop_3d.py:16-18,37-44.
//
// GEMM view (per batch item):  Out[co][v] = sum_{ci,tap} Wt[co][ci][tap] * In[ci][v + tap]
//   MFMA v_mfma_f32_32x32x2_f32:  D[i][j] += A[i][k] * B[k][j],  i = 32 output channels (A = weights),
//   j = 32 voxels (B = input, XB consecutive x times 32/XB rows of y), k = 2 input channels at one tap.
//   Accumulator layout puts j on the lane, so every accumulator register is a 128-byte (XB=32) run of
//   consecutive x for one output channel: NCDHW stores are coalesced without any transpose.
//
// Data movement: the fp32 MFMA issues one 32x32x2 per 64 cycles per SIMD (the f32 vector rate), so the
// kernel is MFMA-issue bound by construction; per K-step a wave needs only (ROWS + NCB) ds_read_b32 for
// ROWS*NCB MFMAs.  A workgroup (4 waves) stages, per chunk of CC input channels, the halo tile
// [CC][TZ+2p][TY+2p][TX+2p] and the matching pre-packed weight slab into LDS.  The next chunk's global
// loads are issued into registers BEFORE the current chunk's MFMA loop and written to LDS after it
// (issue-early / write-late), so HBM/L2 latency hides under the matrix pipe.
//
// All B-fragment addresses are "one per-lane base VGPR + compile-time immediate": the tap loop is fully
// unrolled and the (ci pair, dz, dy, dx) offsets fold into the ds_read offset field.
#include "m3d_common.h"
#include <stdlib.h>

#ifndef M3D_STEM_ROWS
#define M3D_STEM_ROWS 4
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------
// Weight packing.  Packed layout: Wp[cpair][cb][tap][lane]  (lane 0..63; cpair = input-channel pair)
//   value = Wsrc[co = cb*32 + (lane & 31)][ci = 2*cpair + (lane >> 5)][tap]   (0 outside)
// A workgroup that owns NCB consecutive cb reads, per channel pair, one contiguous NCB*K3*64-float run, so
// the chunk size CC (channels staged per barrier) is a pure kernel choice, not a packing property.
// For the k=5 / Cin=1 stem the K index is the tap itself: Wp[cb][pair][lane] = W[co][0][tap = 2*pair + (lane>>5)].
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int cin, int cout, int k, int mode,
                                                           float* __restrict__ wp, int ncb, int npair) {
  const int k3 = k * k * k;
  const long long total = (long long)npair * ncb * k3 * 64;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63);
    long long t = e >> 6;
    const int tap = (int)(t % k3); t /= k3;
    const int cb = (int)(t % ncb); t /= ncb;
    const int cpair = (int)t;
    const int co = cb * 32 + (lane & 31);
    const int ci = 2 * cpair + (lane >> 5);
    // logical conv: out channels `cout_l`, in channels `cin_l`
    const bool dgrad = (mode == M3D_W_DGRAD || mode == M3D_W_DGRAD_RELU);
    const int cout_l = dgrad ? cin : cout, cin_l = dgrad ? cout : cin;
    float v = 0.f;
    if (co < cout_l && ci < cin_l) {
      // dgrad of a stride-1 "same" conv = conv with W'[co'][ci'][tap'] = W[ci'][co'][k3-1-tap']
      v = dgrad ? w[((size_t)ci * cin + co) * k3 + (k3 - 1 - tap)] : w[((size_t)co * cin + ci) * k3 + tap];
      if ((mode == M3D_W_RELU || mode == M3D_W_DGRAD_RELU) && v < 0.f) v = 0.f;   // peak_backprop_3d.py:41
    }
    wp[e] = v;
  }
}

__global__ __launch_bounds__(256) void pack_stem_kernel(const float* __restrict__ w, int cout, int k3, int mode,
                                                        float* __restrict__ wp, int npair) {
  const int ncb = (cout + 31) / 32;
  const int total = ncb * npair * 64;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int lane = e & 63;
    const int pair = (e >> 6) % npair;
    const int cb = (e >> 6) / npair;
    const int co = cb * 32 + (lane & 31);
    const int tap = 2 * pair + (lane >> 5);
    float v = 0.f;
    if (co < cout && tap < k3) {
      v = w[(size_t)co * k3 + tap];
      if (mode == M3D_W_RELU && v < 0.f) v = 0.f;
    }
    wp[e] = v;
  }
}

struct Epilogue {
  const float* scale;   // per output channel or null
  const float* shift;   // per output channel or null
  const float* mul;     // same shape as out, or null
  const float* in_off;  // 1 float (device) or null
  int relu;
  // windowed multiply (PRM PreHook on cropped peak windows): out[b,co,z,y,x] *= full[co, o_b + (z,y,x)] - *full_off,
  // and 0 where the window position lies outside the full tensor.
  const float* full;    // [cout, FD, FH, FW] or null
  const float* full_off;
  const int* origins;   // [batch, 3] (z, y, x)
  int FD, FH, FW;
  uint8_t* argmax;      // fused 2x2x2 max-pool variants only: [batch, cout, D/2, H/2, W/2] or null
  int xcd_map;          // 1: XCD-contiguous tile order, cout tile slowest (set by the launcher, see xcd_contiguous)
  // split output with a sigmoid (the two 1x1x1 RPN heads as ONE conv, rpn_heads.py:96-98,116): channels [0, split_at) go through
  // 1 / (1 + exp(-v)) into `out` viewed as [batch, split_at, ...], channels [split_at, cout) raw into out2 [batch, cout - split_at, ...]
  int split_at;         // 0: off
  float* out2;
};

__device__ inline float apply_epilogue(const Epilogue& ep, float v, int b, int co, int z, int y, int x, size_t o) {
  if (ep.scale) v = v * ep.scale[co];
  if (ep.shift) v = v + ep.shift[co];
  if (ep.relu) v = v > 0.f ? v : 0.f;
  if (ep.mul) v = v * ep.mul[o];
  if (ep.full) {
    const int qz = ep.origins[3 * b] + z, qy = ep.origins[3 * b + 1] + y, qx = ep.origins[3 * b + 2] + x;
    if ((qz >= 0) & (qz < ep.FD) & (qy >= 0) & (qy < ep.FH) & (qx >= 0) & (qx < ep.FW))
      v = v * (ep.full[(((size_t)co * ep.FD + qz) * ep.FH + qy) * ep.FW + qx] - (ep.full_off ? *ep.full_off : 0.f));
    else
      v = 0.f;
  }
  return v;
}


// ---- lean epilogues --------------------------------------------------------------------------------------
// An accumulator register g of a 32x32 block belongs to output channel co0 + KG(g), co0 = block base + 4*(lane>>5).
__device__ __host__ constexpr int KG(int g) { return (g & 3) + 8 * (g >> 2); }

struct ChanAffine { float sc[16], sh[16]; };

// per-channel scale/shift of this lane's 16 channels, loaded ONCE per 32-channel block (not per row / element)
__device__ inline void load_affine(const Epilogue& ep, int co0, int cout, ChanAffine& A) {
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int co = min(co0 + KG(g), cout - 1);
    A.sc[g] = ep.scale ? ep.scale[co] : 1.f;
    A.sh[g] = ep.shift ? ep.shift[co] : 0.f;
  }
}

// one accumulator block row -> NCDHW stores: pointer arithmetic is one add per store (KG(g) * DHW)
__device__ inline void store_block(const Epilogue& ep, float* __restrict__ out, const f32x16& a, const ChanAffine& A, int b,
                                   int cout, int co0, size_t DHW, int H, int W, int z, int y, int x) {
  const size_t sp = ((size_t)z * H + y) * W + x;
  if (ep.split_at) {                  // RPN heads: sigmoid scores and raw deltas leave as two contiguous tensors
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = co0 + KG(g);
      if (co < cout) {
        float v = a[g] * A.sc[g] + A.sh[g];
        if (ep.relu) v = fmaxf(v, 0.f);
        if (co < ep.split_at) out[((size_t)b * ep.split_at + co) * DHW + sp] = 1.f / (1.f + expf(-v));      // torch.sigmoid's own formula
        else ep.out2[((size_t)b * (cout - ep.split_at) + (co - ep.split_at)) * DHW + sp] = v;
      }
    }
    return;
  }
  if (ep.mul || ep.full) {            // PRM paths: element-wise multiply tensors (rare, keep the general form)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = co0 + KG(g);
      if (co < cout) {
        const size_t o = ((size_t)b * cout + co) * DHW + sp;
        out[o] = apply_epilogue(ep, a[g], b, co, z, y, x, o);
      }
    }
    return;
  }
  float* p = out + ((size_t)b * cout + co0) * DHW + sp;
  const bool all = co0 + KG(15) < cout;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    if (all || co0 + KG(g) < cout) {
      float v = a[g] * A.sc[g] + A.sh[g];
      if (ep.relu) v = fmaxf(v, 0.f);
      p[(size_t)KG(g) * DHW] = v;
    }
  }
}

// Fused 2x2x2 max-pool epilogue (XB = 32, four rows per wave = the 2x2 (z,y) footprint of one pooling window row).
// Row r = zz*2 + yy; window index q = zz*4 + yy*2 + (x&1), (z,y,x) scan order, first maximum wins like
// nn.MaxPool3d (DSN.py:21,26).
__device__ inline void pool_block(const Epilogue& ep, float* __restrict__ out, const f32x16 (&a)[4], const ChanAffine& A, int b,
                                  int cout, int co0, int lane_x, int oz, int oy, int ox, int OD, int OH, int OW) {
  const size_t ODHW = (size_t)OD * OH * OW;
  const size_t o0 = ((size_t)b * cout + co0) * ODHW + ((size_t)oz * OH + oy) * OW + ox;
  const bool inb = ((lane_x & 1) == 0) & (oz < OD) & (oy < OH) & (ox < OW);
  const int xq = lane_x & 1;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    float best = a[0][g] * A.sc[g] + A.sh[g];
    if (ep.relu) best = fmaxf(best, 0.f);
    int q = xq;
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      float v = a[r][g] * A.sc[g] + A.sh[g];
      if (ep.relu) v = fmaxf(v, 0.f);
      if (v > best) { best = v; q = r * 2 + xq; }
    }
    const float ov = __shfl_xor(best, 1, 64);
    const int oq = __shfl_xor(q, 1, 64);
    if (ov > best || (ov == best && oq < q)) { best = ov; q = oq; }
    if (inb && co0 + KG(g) < cout) {
      const size_t o = o0 + (size_t)KG(g) * ODHW;
      out[o] = best;
      if (ep.argmax) ep.argmax[o] = (uint8_t)q;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Generic k in {1,3} kernel.
//   XB   : x extent of a 32-voxel block (32, 16 or 8);  YB = 32 / XB rows of y per block
//   ROWS : voxel blocks per wave (stacked along y);  NCB : 32-channel output blocks per workgroup
//   WZ, WY: wave grid inside the workgroup (WZ * WY == 4); tile = XB x (WY*ROWS*YB) x WZ voxels
// ------------------------------------------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin (id % 8) and every XCD has its own L2.  Map the ids one XCD receives
// onto a CONTIGUOUS range of tiles, so that the halo planes neighbouring tiles share are re-read from that XCD's L2
// instead of being fetched over the fabric once per XCD (measured with FETCH_SIZE: profiles/r01_pmc_traffic.json).
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
  const int per = n >> 3, rem = n & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

template <int K, int CC, int XB, int ROWS, int NCB, int WZ, int WY, bool POOL = false, int KS = 1, int DIL = 1>
struct Cfg {
  static constexpr int NT = 256 * KS;                    // threads per workgroup (KS = in-workgroup split of K)
  static constexpr int PP = CC / 2 / KS;                  // channel pairs per chunk handled by one K-split group
  static constexpr int P = (K / 2) * DIL;                 // halo = padding = dilation * (K / 2) ("same" convolution)
  static constexpr int K3 = K * K * K;
  static constexpr int YB = 32 / XB;
  // rows of a wave: r -> (zz = r / RY, yy = r % RY).  POOL: 2 z-levels x 2 y-rows = one pooling footprint.
  static constexpr int RY = POOL ? ROWS / 2 : ROWS, RZ = ROWS / RY;
  static constexpr int TX = XB, TY = WY * RY * YB, TZ = WZ * RZ;
  static constexpr int HX = TX + 2 * P, HY = TY + 2 * P, HZ = TZ + 2 * P;
  static constexpr int CS = HX * HY * HZ;                 // per-channel LDS stride (floats)
  static constexpr int IN_ELEMS = CC * CS;
  static constexpr int W_SEG = NCB * K3 * 64;              // floats per channel pair for this workgroup
  static constexpr int W_ELEMS = (CC / 2) * W_SEG;
  static constexpr int NI = (IN_ELEMS + NT - 1) / NT;     // input staging registers per thread
  static constexpr int NW4 = (W_ELEMS / 4 + NT - 1) / NT; // weight staging float4 per thread
  static constexpr int LDS_FLOATS = IN_ELEMS + W_ELEMS;
  static_assert(WZ * WY == 4, "4 waves per workgroup");
  static_assert(W_SEG % 4 == 0, "weights staged as float4");
  static_assert(!POOL || (ROWS == 4 && XB == 32), "fused pool: 32-wide blocks, 2x2 rows per wave");
  static_assert(KS == 1 || (KS == 2 && !POOL && CC % 4 == 0), "split-K: two groups of 4 waves");
  static_assert(KS == 1 || 4 * NCB * ROWS * 16 * 64 <= 2 * LDS_FLOATS, "split-K reduction buffer fits the staging area");
};

template <int K, int CC, int XB, int ROWS, int NCB, int WZ, int WY, bool POOL, int KS, int DIL = 1>
__global__ __launch_bounds__(256 * KS, 2) void conv3d_mfma_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                          float* __restrict__ out, int cin, int cout, int D, int H, int W,
                                                          int tiles_x, int tiles_y, int tiles_z, int ncb_total, Epilogue ep) {
  using C = Cfg<K, CC, XB, ROWS, NCB, WZ, WY, POOL, KS, DIL>;
  extern __shared__ float lds[];
  float* lds_in = lds;
  float* lds_w = lds + C::IN_ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = (tid >> 6) & 3;      // position inside a K-split group
  const int ks = tid >> 8;               // K-split group (0 when KS == 1)
  const int wz = wave / WY, wy = wave % WY;

  // block -> (cout tile, x tile, y tile, z tile), batch = blockIdx.y
  int bid = blockIdx.x;
  const int co_tiles = ((cout + 31) / 32 + NCB - 1) / NCB;
  int cot;
  if (ep.xcd_map) {   // each XCD: one contiguous run of spatial tiles of (mostly) one cout tile -> its L2 holds that slab + those weights
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
  const float in_off = ep.in_off ? *ep.in_off : 0.f;

  // ---- per-thread staging descriptors (constant across chunks except for the channel base) ----
  // goff: offset inside one channel chunk of the input; -1 = zero padding, -2 = beyond the tile (no LDS write).
  int goff[C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int e = tid + i * C::NT;
    int g = -2;
    if (e < C::IN_ELEMS) {
      const int ci = e / C::CS;
      const int r = e % C::CS;
      const int hz = r / (C::HY * C::HX), hy = (r / C::HX) % C::HY, hx = r % C::HX;
      const int z = z0 + hz - C::P, y = y0 + hy - C::P, x = x0 + hx - C::P;
      const bool ok = (z >= 0) & (z < D) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
      g = ok ? (int)(ci * DHW + ((size_t)z * H + y) * W + x) : -1;   // channel chunk fits in int (checked on host)
    }
    goff[i] = g;
  }

  float rin[C::NI];
  f32x4 rw[C::NW4];
  const int nchunk = (cin + CC - 1) / CC;
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  // packed weights are zero-padded to a multiple of 16 channel pairs and 2 cout blocks (conv3d_packed_dims),
  // so the slab loads below never need a predicate.
  const size_t w_pair_stride4 = (size_t)ncb_total * C::K3 * 64 / 4;      // one channel pair, all (padded) cout blocks
  const size_t w_tile_off4 = (size_t)cot * NCB * C::K3 * 64 / 4;

  // Staging of chunk c+1 is spread THROUGH the MFMA loop of chunk c (loads in its first third, LDS writes to the
  // other buffer in its last third): a block of ~100 address/VMEM/LDS instructions in front of the loop would
  // leave the matrix pipe idle once per chunk (measured: 8-19 % of kernel time).  Loads are unconditional
  // (index clamped to a valid element) and carry no arithmetic; zero padding and the PRM input offset are
  // applied at commit time.
  constexpr int NL = C::NI + C::NW4;                         // staging operations per thread and chunk
  auto issue = [&](int idx, int chunk) __attribute__((always_inline)) {
    if (idx < C::NI) {
      const int g = goff[idx];
      const int cvalid = min(CC, cin - chunk * CC);
      const bool ok = (g >= 0) & (((tid + idx * C::NT) / C::CS) < cvalid);
      rin[idx] = (in_b + (size_t)chunk * CC * DHW)[ok ? g : 0];
    } else {
      const int i = idx - C::NI;
      int e = tid + i * C::NT;
      if (e >= C::W_ELEMS / 4) e = C::W_ELEMS / 4 - 1;       // only the last i can overshoot
      const int pr = e / (C::W_SEG / 4), o = e % (C::W_SEG / 4);
      rw[i] = (wp4 + (size_t)chunk * (CC / 2) * w_pair_stride4 + w_tile_off4)[(size_t)pr * w_pair_stride4 + o];
    }
  };
  auto commit1 = [&](int idx, int chunk, float* dst_in, float* dst_w) __attribute__((always_inline)) {
    if (idx < C::NI) {
      const int g = goff[idx];
      const int cvalid = min(CC, cin - chunk * CC);
      const bool ok = (g >= 0) & (((tid + idx * C::NT) / C::CS) < cvalid);
      if (g != -2) dst_in[tid + idx * C::NT] = ok ? rin[idx] - in_off : 0.f;
    } else {
      const int i = idx - C::NI;
      const int e = tid + i * C::NT;
      if (e < C::W_ELEMS / 4) reinterpret_cast<f32x4*>(dst_w)[e] = rw[i];
    }
  };

  f32x16 acc[NCB][ROWS];
#pragma unroll
  for (int c = 0; c < NCB; ++c)
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[c][r][g] = 0.f;

  // per-lane LDS base (floats) of the B fragment for row block r = 0, tap (0,0,0), pair 0
  const int jx = (lane & 31) % XB, jy = (lane & 31) / XB;
  const int b_base = (lane >> 5) * C::CS + wz * C::RZ * (C::HY * C::HX) + (wy * C::RY * C::YB + jy) * C::HX + jx;

  // double-buffered LDS: chunk c is computed from buffer c&1 while chunk c+1 is staged into the other one;
  // one barrier per chunk.
#pragma unroll
  for (int i = 0; i < NL; ++i) issue(i, 0);
#pragma unroll
  for (int i = 0; i < NL; ++i) commit1(i, 0, lds_in, lds_w);
  __syncthreads();
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const float* cur_in = lds + (chunk & 1) * C::LDS_FLOATS;
    const float* cur_w = cur_in + C::IN_ELEMS;
    float* nxt_in = lds + ((chunk + 1) & 1) * C::LDS_FLOATS;
    float* nxt_w = nxt_in + C::IN_ELEMS;
    const int nchk = min(chunk + 1, nchunk - 1);   // the last chunk re-stages itself (never read): keeps the loop branch-free
    // K loop of this chunk, software-pipelined: the fragments of step s+DIST are requested from LDS before the
    // MFMAs of step s issue, so a wave with few accumulators does not sit on the LDS latency.
    {
      const float* in_k = cur_in + b_base + ks * (C::PP * 2 * C::CS);       // K-split group offset (0 when KS == 1)
      const float* w_k = cur_w + ks * (C::PP * C::W_SEG) + lane;
      constexpr int NS = C::K3 * C::PP;
      auto load_frag = [&](int s, float (&b)[ROWS], float (&a)[NCB]) __attribute__((always_inline)) {
        const int tap = s / C::PP, pp = s % C::PP;
        const int dz = tap / (K * K), dy = (tap / K) % K, dx = tap % K;
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          b[r] = in_k[(r / C::RY) * (C::HY * C::HX) + (r % C::RY) * C::YB * C::HX + pp * 2 * C::CS + DIL * (dz * (C::HY * C::HX) +
                      dy * C::HX + dx)];
#pragma unroll
        for (int c = 0; c < NCB; ++c) a[c] = w_k[pp * C::W_SEG + (c * C::K3 + tap) * 64];
      };
      constexpr int DIST = (ROWS * NCB >= 4) ? 1 : 3;
      constexpr int THIRD = NS / 3 > 0 ? NS / 3 : 1;
      constexpr int LPS = (NL + THIRD - 1) / THIRD;                 // staging ops per K step
      constexpr int CSTART = NS - (NL + LPS - 1) / LPS;             // first step of the commit phase
      float bfq[DIST + 1][ROWS], afq[DIST + 1][NCB];
#pragma unroll
      for (int s = 0; s < DIST && s < NS; ++s) load_frag(s, bfq[s % (DIST + 1)], afq[s % (DIST + 1)]);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + DIST < NS) load_frag(s + DIST, bfq[(s + DIST) % (DIST + 1)], afq[(s + DIST) % (DIST + 1)]);
#pragma unroll
        for (int q = 0; q < LPS; ++q)
          if (s * LPS + q < NL) issue(s * LPS + q, nchk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
          for (int r = 0; r < ROWS; ++r)
            acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[s % (DIST + 1)][c], bfq[s % (DIST + 1)][r], acc[c][r], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < LPS; ++q)
          if (s >= CSTART && (s - CSTART) * LPS + q < NL) commit1((s - CSTART) * LPS + q, nchk, nxt_in, nxt_w);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  if constexpr (KS == 2) {
    // combine the two K halves through LDS (the staging buffers are free after the final barrier)
    float* red = lds + ((size_t)wave * NCB * ROWS * 16) * 64 + lane;
    if (ks == 1) {
#pragma unroll
      for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
          for (int g = 0; g < 16; ++g) red[((c * ROWS + r) * 16 + g) * 64] = acc[c][r][g];
    }
    __syncthreads();
    if (ks == 1) return;
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[c][r][g] += red[((c * ROWS + r) * 16 + g) * 64];
  }

  // ---- epilogue: y = acc*scale + shift ; relu ; * mul ; coalesced NCDHW stores (or fused 2x2x2 max-pool) ----
#pragma unroll
  for (int c = 0; c < NCB; ++c) {
    const int co0 = (cot * NCB + c) * 32 + 4 * (lane >> 5);
    ChanAffine A;
    load_affine(ep, co0, cout, A);
    if constexpr (POOL) {
      pool_block(ep, out, acc[c], A, b, cout, co0, jx, (z0 + wz * 2) >> 1, (y0 + wy * 2) >> 1, (x0 + jx) >> 1, D / 2, H / 2, W / 2);
    } else {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const int z = z0 + wz * C::RZ + r / C::RY;
        const int x = x0 + jx;
        const int y = y0 + (wy * C::RY + r % C::RY) * C::YB + jy;
        if (x < W && y < H && z < D) store_block(ep, out, acc[c][r], A, b, cout, co0, DHW, H, W, z, y, x);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Stem kernel: k = 5, Cin = 1 (conv1a, DSN.py:19).  K index = tap (125 -> 63 pairs).  The two halves of
// a wave read taps 2p and 2p+1; their LDS distance is +1 inside an x run, HX-4 at a dx wrap, and
// HY*HX-4*HX-4 at a dy wrap: three per-lane base registers, every other offset is an immediate.
// Tile: 32 x (ROWS*WY) x WZ voxels, all (<=32*NCB) output channels.
// ------------------------------------------------------------------------------------------------------
template <int ROWS, int NCB, int WZ, int WY, bool POOL>
__global__ __launch_bounds__(256, 2) void conv3d_stem5_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                           float* __restrict__ out, int cout, int D, int H, int W, int tiles_x,
                                                           int tiles_y, Epilogue ep) {
  constexpr int K = 5, P = 2, K3 = 125, NPAIR = 63;
  constexpr int RY = POOL ? ROWS / 2 : ROWS, RZ = ROWS / RY;   // rows of a wave: r -> (zz = r / RY, yy = r % RY)
  constexpr int TX = 32, TY = RY * WY, TZ = WZ * RZ;
  constexpr int HX = TX + 4, HY = TY + 4, HZ = TZ + 4;
  static_assert(!POOL || ROWS == 4, "fused pool: 2x2 rows per wave");
  constexpr int IN_ELEMS = HX * HY * HZ;
  constexpr int IN_PAD = (IN_ELEMS + 8 + 3) / 4 * 4;
  constexpr int W_ELEMS = NCB * NPAIR * 64;
  extern __shared__ float lds[];
  float* lds_in = lds;
  float* lds_w = lds + IN_PAD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wz = wave / WY, wy = wave % WY;
  int bid = ep.xcd_map ? xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int tz = bid;
  const int b = blockIdx.y;
  const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
  const size_t DHW = (size_t)D * H * W;
  const float* in_b = in + (size_t)b * DHW;
  const float in_off = ep.in_off ? *ep.in_off : 0.f;
  // all loads of the halo tile are issued back to back (unconditional, clamped index), then written to LDS
  constexpr int NI = (IN_PAD + 255) / 256;
  float rin[NI];
  bool okv[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + i * 256;
    const int hz = e / (HY * HX), hy = (e / HX) % HY, hx = e % HX;
    const int z = z0 + hz - P, y = y0 + hy - P, x = x0 + hx - P;
    okv[i] = (e < IN_ELEMS) & (z >= 0) & (z < D) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
    rin[i] = in_b[okv[i] ? ((size_t)z * H + y) * W + x : 0];
  }
  constexpr int NWv = (W_ELEMS / 4 + 255) / 256;
  f32x4 rwv[NWv];
#pragma unroll
  for (int i = 0; i < NWv; ++i) {
    int e = tid + i * 256;
    if (e >= W_ELEMS / 4) e = W_ELEMS / 4 - 1;
    rwv[i] = reinterpret_cast<const f32x4*>(wp)[e];
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + i * 256;
    if (e < IN_PAD) lds_in[e] = okv[i] ? rin[i] - in_off : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NWv; ++i) {
    const int e = tid + i * 256;
    if (e < W_ELEMS / 4) reinterpret_cast<f32x4*>(lds_w)[e] = rwv[i];
  }
  __syncthreads();

  f32x16 acc[NCB][ROWS];
#pragma unroll
  for (int c = 0; c < NCB; ++c)
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[c][r][g] = 0.f;

  const int jx = lane & 31;
  const int base0 = wz * RZ * (HY * HX) + (wy * RY) * HX + jx;
  const int hi = lane >> 5;
  const int baseA = base0 + hi * 1;
  const int baseB = base0 + hi * (HX - 4);
  const int baseC = base0 + hi * (HY * HX - 4 * HX - 4);
#pragma unroll
  for (int pair = 0; pair < NPAIR; ++pair) {
    const int tap = 2 * pair;
    const int dz = tap / 25, dy = (tap / 5) % 5, dx = tap % 5;
    const int imm = dz * (HY * HX) + dy * HX + dx;
    // which delta takes tap -> tap+1
    const int base = (dx < 4) ? baseA : ((dy < 4) ? baseB : baseC);
    float bf[ROWS], af[NCB];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) bf[r] = lds_in[base + (r / RY) * (HY * HX) + (r % RY) * HX + imm];
#pragma unroll
    for (int c = 0; c < NCB; ++c) af[c] = lds_w[(c * NPAIR + pair) * 64 + lane];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c], bf[r], acc[c][r], 0, 0, 0);
  }
  (void)K; (void)K3;
#pragma unroll
  for (int c = 0; c < NCB; ++c) {
    const int co0 = c * 32 + 4 * (lane >> 5);
    ChanAffine A;
    load_affine(ep, co0, cout, A);
    if constexpr (POOL) {
      pool_block(ep, out, acc[c], A, b, cout, co0, jx, (z0 + wz * 2) >> 1, (y0 + wy * 2) >> 1, (x0 + jx) >> 1, D / 2, H / 2, W / 2);
    } else {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const int z = z0 + wz * RZ + r / RY;
        const int x = x0 + jx, y = y0 + wy * RY + r % RY;
        if (x < W && y < H && z < D) store_block(ep, out, acc[c][r], A, b, cout, co0, DHW, H, W, z, y, x);
      }
    }
  }
}

inline int xcd_map_enabled() {   // M3D_XCD_MAP=0 restores the plain round-robin order (A/B measurements)
  return m3d::opt(m3d::OPT_XCD_MAP) != 0;
}

template <int K, int CC, int XB, int ROWS, int NCB, int WZ, int WY, bool POOL = false, int KS = 1, int DIL = 1>
int launch_cfg(const float* in, const float* wp, float* out, int B, int cin, int cout, int D, int H, int W, Epilogue ep,
               hipStream_t st) {
  using C = Cfg<K, CC, XB, ROWS, NCB, WZ, WY, POOL, KS, DIL>;
  const int tiles_x = (W + C::TX - 1) / C::TX, tiles_y = (H + C::TY - 1) / C::TY, tiles_z = (D + C::TZ - 1) / C::TZ;
  const int ncb_total = ((cout + 31) / 32 + 1) / 2 * 2;   // as packed (padded to 2 blocks)
  const int co_tiles = ((cout + 31) / 32 + NCB - 1) / NCB;
  const long long blocks = (long long)tiles_x * tiles_y * tiles_z * co_tiles;
  if (blocks > 0x7FFFFFFFll || B > 65535) return M3D_EUNSUPPORTED;
  ep.xcd_map = xcd_map_enabled();
  const size_t lds = sizeof(float) * 2 * C::LDS_FLOATS;   // double-buffered
  auto kern = conv3d_mfma_kernel<K, CC, XB, ROWS, NCB, WZ, WY, POOL, KS, DIL>;
  if (lds > 160 * 1024) return M3D_EUNSUPPORTED;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, B), dim3(C::NT), lds, st, in, wp, out, cin, cout, D, H, W, tiles_x, tiles_y,
                     tiles_z, ncb_total, ep);
  return m3d::check_launch("conv3d_mfma");
}

}  // namespace

M3D_API size_t m3d_conv3d_packed_weight_bytes(int cin, int cout, int k, int mode) {
  const bool dgrad = (mode == M3D_W_DGRAD || mode == M3D_W_DGRAD_RELU);
  const int cin_l = dgrad ? cout : cin, cout_l = dgrad ? cin : cout;
  if (k == 5 && cin_l == 1) return sizeof(float) * (size_t)((cout_l + 31) / 32) * 63 * 64;
  const size_t npair = ((cin_l + 1) / 2 + 15) / 16 * 16, ncb = ((cout_l + 31) / 32 + 1) / 2 * 2;   // zero-padded
  return sizeof(float) * npair * ncb * (size_t)(k * k * k) * 64;
}

M3D_API int m3d_conv3d_pack_weights(const float* d_weight, int cin, int cout, int k, int mode, float* d_packed, void* stream) {
  if (!d_weight || !d_packed || cin <= 0 || cout <= 0 || mode < 0 || mode > 3) return M3D_EINVAL;
  const bool dgrad = (mode == M3D_W_DGRAD || mode == M3D_W_DGRAD_RELU);
  const int cin_l = dgrad ? cout : cin, cout_l = dgrad ? cin : cout;
  hipStream_t st = m3d::as_stream(stream);
  if (k == 5 && cin_l == 1) {
    if (dgrad) return M3D_EUNSUPPORTED;
    hipLaunchKernelGGL(pack_stem_kernel, dim3(64), dim3(256), 0, st, d_weight, cout, 125, mode, d_packed, 63);
    return m3d::check_launch("pack_stem");
  }
  if (k != 1 && k != 3) return M3D_EUNSUPPORTED;
  const int npair = ((cin_l + 1) / 2 + 15) / 16 * 16, ncb = ((cout_l + 31) / 32 + 1) / 2 * 2;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(1024), dim3(256), 0, st, d_weight, cin, cout, k, mode, d_packed, ncb, npair);
  return m3d::check_launch("pack_weights");
}

static int conv_dispatch(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout, int depth,
                         int height, int width, int k, Epilogue ep, hipStream_t st, bool pool = false) {
  if (!d_in || !d_packed || !d_out || batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0)
    return M3D_EINVAL;
  const size_t DHW = (size_t)depth * height * width;
  if (DHW * 32 >= 0x7FFFFFFFull) return M3D_EUNSUPPORTED;   // int offsets inside a channel chunk
  if (k == 5) {
    ep.xcd_map = xcd_map_enabled();
    if (cin != 1 || cout > 64) return M3D_EUNSUPPORTED;
    const int tiles_x = (width + 31) / 32;
    if (batch > 65535) return M3D_EUNSUPPORTED;
    auto lds_bytes = [](int ty, int tz, int ncb) { return sizeof(float) * ((36 * (ty + 4) * (tz + 4) + 8 + 3) / 4 * 4 + ncb * 63 * 64); };
    if (pool) {      // tile 32 x 4 x 4, waves 2(z) x 2(y), each wave = one 2x2 (z,y) pooling footprint
      if (cout > 32) return M3D_EUNSUPPORTED;
      const int ty = (height + 3) / 4, tz = (depth + 3) / 4;
      hipLaunchKernelGGL((conv3d_stem5_kernel<4, 1, 2, 2, true>), dim3((unsigned)(tiles_x * ty * tz), batch), dim3(256),
                         lds_bytes(4, 4, 1), st, d_in, d_packed, d_out, cout, depth, height, width, tiles_x, ty, ep);
    } else if (cout <= 32) {
      constexpr int ROWS = M3D_STEM_ROWS;
      const int ty = (height + ROWS - 1) / ROWS, tz = (depth + 3) / 4;
      hipLaunchKernelGGL((conv3d_stem5_kernel<ROWS, 1, 4, 1, false>), dim3((unsigned)(tiles_x * ty * tz), batch), dim3(256),
                         lds_bytes(ROWS, 4, 1), st, d_in, d_packed, d_out, cout, depth, height, width, tiles_x, ty, ep);
    } else {
      const int ty = (height + 3) / 4, tz = (depth + 3) / 4;
      hipLaunchKernelGGL((conv3d_stem5_kernel<4, 2, 4, 1, false>), dim3((unsigned)(tiles_x * ty * tz), batch), dim3(256),
                         lds_bytes(4, 4, 2), st, d_in, d_packed, d_out, cout, depth, height, width, tiles_x, ty, ep);
    }
    return m3d::check_launch("conv3d_stem5");
  }
  if (k == 3) {
    // Tile choice: keep >= ~2 workgroups per CU in flight; prefer big tiles (more MFMAs per staged byte);
    // pick the x-block (32/16/8) that wastes the fewest lanes on this width.
    const long long vox = (long long)batch * DHW;
    const int ncb_total = (cout + 31) / 32;
    auto waste = [&](int xb) { return (double)((width + xb - 1) / xb * xb) / width; };
    int xb = 32;
    if (waste(16) < waste(xb) - 0.05) xb = 16;
    if (waste(8) < waste(xb) - 0.05) xb = 8;
    if (pool) {
      if (width < 24) return M3D_EUNSUPPORTED;
      return launch_cfg<3, 2, 32, 4, 2, 2, 2, true>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
    }
    // tuning override (tools/bench_layers.py only): m3d_set_option("tune_k3", <variant index>)
    if (const int v = m3d::opt(m3d::OPT_TUNE_K3); v >= 0) {
#define M3D_V(i, ...) if (v == i) return launch_cfg<__VA_ARGS__>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      M3D_V(0, 3, 2, 32, 4, 2, 4, 1)
      M3D_V(1, 3, 2, 32, 2, 2, 4, 1)
      M3D_V(2, 3, 4, 32, 1, 2, 4, 1)
      M3D_V(3, 3, 2, 32, 4, 1, 4, 1)
      M3D_V(4, 3, 4, 32, 4, 1, 4, 1)
      M3D_V(5, 3, 4, 32, 2, 2, 4, 1)
      M3D_V(6, 3, 2, 32, 2, 1, 4, 1)
      M3D_V(7, 3, 4, 32, 2, 1, 4, 1)
      M3D_V(8, 3, 2, 32, 4, 2, 2, 2)
      M3D_V(10, 3, 4, 16, 1, 1, 4, 1)
      M3D_V(11, 3, 8, 16, 1, 1, 4, 1)
      M3D_V(12, 3, 4, 16, 2, 1, 4, 1)
      M3D_V(13, 3, 8, 16, 2, 1, 4, 1)
      M3D_V(14, 3, 4, 16, 1, 2, 4, 1)
      M3D_V(15, 3, 8, 16, 1, 2, 4, 1)
      M3D_V(16, 3, 4, 16, 2, 2, 4, 1)
      M3D_V(20, 3, 8, 16, 1, 1, 4, 1, false, 2)
      M3D_V(21, 3, 4, 16, 1, 1, 4, 1, false, 2)
      M3D_V(22, 3, 16, 16, 1, 1, 4, 1, false, 2)
      M3D_V(23, 3, 8, 16, 1, 2, 4, 1, false, 2)
      M3D_V(30, 3, 4, 32, 2, 2, 4, 1, false, 2)
      M3D_V(31, 3, 8, 32, 2, 2, 4, 1, false, 2)
      M3D_V(33, 3, 4, 32, 1, 2, 4, 1, false, 2)
      M3D_V(34, 3, 8, 32, 1, 2, 4, 1, false, 2)
      M3D_V(35, 3, 8, 32, 2, 1, 4, 1, false, 2)
      M3D_V(36, 3, 8, 32, 1, 1, 4, 1, false, 2)
      M3D_V(24, 3, 8, 16, 2, 1, 4, 1, false, 2)
#undef M3D_V
    }
    if (ncb_total == 1) {   // <= 32 output channels (e.g. the dgrad of conv2a): never pad to a second, empty cout block
      if (xb == 32) return launch_cfg<3, 4, 32, 4, 1, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      if (xb == 16) return launch_cfg<3, 4, 16, 2, 1, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      return launch_cfg<3, 4, 8, 2, 1, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
    }
    if (xb == 32) {
      const long long wg_big = (vox / 512) * ((ncb_total + 1) / 2);
      if (wg_big >= 512) return launch_cfg<3, 2, 32, 4, 2, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      const long long wg_mid = (vox / 256) * ((ncb_total + 1) / 2);
      // 32^3-class maps: one workgroup per CU -> split K over two groups of 4 waves (2 waves/SIMD)
      if (wg_mid >= 256 && wg_mid < 1024 && cin >= 16)
        return launch_cfg<3, 8, 32, 2, 2, 4, 1, false, 2>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      // fewer than one such workgroup per CU (e.g. 128 -> 64 channels on 32^3, the dgrad of conv3a): halve the voxel tile
      if (wg_mid >= 64 && wg_mid < 256 && cin >= 16)
        return launch_cfg<3, 8, 32, 1, 2, 4, 1, false, 2>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      if (wg_mid >= 256) return launch_cfg<3, 2, 32, 2, 2, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      return launch_cfg<3, 4, 32, 1, 2, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
    }
    if (xb == 16) {
      const long long wg = (vox / 256) * ((ncb_total + 1) / 2);
      if (wg >= 1024) return launch_cfg<3, 4, 16, 2, 2, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      // one to two rounds of the big tile (the soma tile's 32 x 80 x 80 maps: 800 workgroups on 512 slots) leave a ragged second round:
      // the quarter-size tile (tools/tune_k3_prm.py, round 4: conv2a 0.256 -> 0.202 ms, conv2b 0.438 -> 0.382 ms; equal from 1250 up)
      if (wg >= 512) return launch_cfg<3, 4, 16, 1, 1, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      // 16^3-class maps: exactly one 32x32 output block per SIMD -> split K in the workgroup for 2 waves/SIMD
      if (cin >= 16) return launch_cfg<3, 8, 16, 1, 1, 4, 1, false, 2>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      return launch_cfg<3, 4, 16, 1, 1, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
    }
    {
      const long long wg = (vox / 256) * ((ncb_total + 1) / 2);
      if (wg >= 512) return launch_cfg<3, 4, 8, 2, 2, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
      return launch_cfg<3, 4, 8, 1, 1, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
    }
  }
  if (pool) return M3D_EUNSUPPORTED;
  if (k == 1) {
    if (width >= 24) return launch_cfg<1, 32, 32, 1, 2, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
    if (width >= 12) return launch_cfg<1, 32, 16, 1, 2, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
    return launch_cfg<1, 32, 8, 1, 2, 4, 1>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep, st);
  }
  return M3D_EUNSUPPORTED;
}

M3D_API int m3d_conv3d_forward(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout, int depth,
                               int height, int width, int k, const float* d_in_offset, const float* d_scale,
                               const float* d_shift, int relu, const float* d_mul, void* stream) {
  Epilogue ep{d_scale, d_shift, d_mul, d_in_offset, relu, nullptr, nullptr, nullptr, 0, 0, 0, nullptr};
  return conv_dispatch(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, k, ep, m3d::as_stream(stream));
}

/* 3x3x3 "same" convolution with dilation 2 (padding 2): the convs of the mask head, lib/modeling/mask_rcnn_heads.py:148-151 with
 * MRCNN.DILATION = 2 (lib/core/config.py:767).  Same packed weights as m3d_conv3d_forward; small maps (the 7^3 / 14^3 RoI grids). */
/* The two 1x1x1 RPN heads as ONE convolution with both epilogues (lib/modeling/rpn_heads.py:96-98 + the sigmoid of :116): output
 * channels [0, split) -> sigmoid -> d_out_sigmoid [batch, split, D, H, W]; channels [split, cout) -> d_out_rest [batch, cout - split, ...].
 * Replaces a conv + torch.sigmoid + two slice copies (three library launches of the detection step). */
M3D_API int m3d_conv3d_forward_split_sigmoid(const float* d_in, const float* d_packed, float* d_out_sigmoid, float* d_out_rest, int batch, int cin,
                                             int cout, int split, int depth, int height, int width, int k, const float* d_shift, void* stream) {
  if (!d_out_sigmoid || !d_out_rest || split <= 0 || split >= cout) return M3D_EINVAL;
  Epilogue ep{nullptr, d_shift, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr};
  ep.split_at = split; ep.out2 = d_out_rest;
  return conv_dispatch(d_in, d_packed, d_out_sigmoid, batch, cin, cout, depth, height, width, k, ep, m3d::as_stream(stream));
}

M3D_API int m3d_conv3d_forward_dilated(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout, int depth,
                                       int height, int width, int k, int dilation, const float* d_scale, const float* d_shift, int relu,
                                       void* stream) {
  if (dilation == 1)
    return m3d_conv3d_forward(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, k, nullptr, d_scale, d_shift, relu, nullptr,
                              stream);
  if (!d_in || !d_packed || !d_out || batch <= 0 || cin <= 0 || cout <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (k != 3 || dilation != 2) return M3D_EUNSUPPORTED;
  if ((size_t)min(cin, 4) * depth * height * width >= 0x7FFFFFFFull) return M3D_EUNSUPPORTED;
  Epilogue ep{d_scale, d_shift, nullptr, nullptr, relu, nullptr, nullptr, nullptr, 0, 0, 0, nullptr};
  // 8 x 8 x 4 voxel tile, 64 output channels per workgroup: the RoI grids are 7 or 14 voxels wide
  return launch_cfg<3, 4, 8, 2, 2, 4, 1, false, 1, 2>(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, ep,
                                                        m3d::as_stream(stream));
}

M3D_API int m3d_conv3d_forward_windowed(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                        int depth, int height, int width, int k, const float* d_full, const float* d_full_offset,
                                        const int32_t* d_origins, int full_depth, int full_height, int full_width, void* stream) {
  if (!d_full || !d_origins || full_depth <= 0 || full_height <= 0 || full_width <= 0) return M3D_EINVAL;
  Epilogue ep{nullptr, nullptr, nullptr, nullptr, 0, d_full, d_full_offset, d_origins, full_depth, full_height, full_width, nullptr};
  return conv_dispatch(d_in, d_packed, d_out, batch, cin, cout, depth, height, width, k, ep, m3d::as_stream(stream));
}

M3D_API int m3d_conv3d_forward_pool2(const float* d_in, const float* d_packed, float* d_out_pooled, uint8_t* d_argmax, int batch,
                                     int cin, int cout, int depth, int height, int width, int k, const float* d_in_offset,
                                     const float* d_scale, const float* d_shift, int relu, void* stream) {
  if (depth < 2 || height < 2 || width < 2) return M3D_EINVAL;
  Epilogue ep{d_scale, d_shift, nullptr, d_in_offset, relu, nullptr, nullptr, nullptr, 0, 0, 0, d_argmax};
  return conv_dispatch(d_in, d_packed, d_out_pooled, batch, cin, cout, depth, height, width, k, ep, m3d::as_stream(stream), true);
}
