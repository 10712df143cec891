// 3x3x3 forward convolution (stride 1, pad 1) + eval-BN + ReLU [+ MaxPool3d(2,2)] of lib/modeling/DSN.py:57-68 on the f16 matrix cores at
// fp32 accuracy: the "f16x2" cut of fc_gemm.hip (both operands scaled by a power of two and cut into two fp16 numbers, 22 bits; three
// v_mfma_f32_32x32x16_f16 per fp32 product, fp32 accumulation) combined with Winograd F(2,3) ALONG Z:
//
//   per output z pair (z0, z0 + 1) and input planes d0..d3 = in[z0 - 1 .. z0 + 2]:
//     V0 = d0 - d2,  V1 = d1 + d2,  V2 = d2 - d1,  V3 = d1 - d3                  (input side: two-plane sums, formed while staging)
//     U0 = g0,  U1 = (g0 + g1 + g2) / 2,  U2 = (g0 - g1 + g2) / 2,  U3 = g2      (weights g_dz, transformed in double at pack time)
//     M_k = sum over (ci, dy, dx) of U_k * V_k   (four direct 3 x 3 convolutions on the (y, x) plane: the matrix-core work)
//     out[z0] = M0 + M1 + M2,  out[z0 + 1] = M1 - M2 - M3
//   36 products per output pair and (ci, co) instead of 54: 2/3 of the direct convolution's matrix work, x 3 for the f16x2 cut = 2 f16
//   products per algorithmic multiply-add, at 16 x the fp32 MFMA's rate (the F(2x4,3x3) fp32 kernel of conv3d_wino24.hip issues 1/3 of
//   the direct products at the fp32 rate: 2.7 x the matrix-pipe time of this form).
//
// Decomposition: a UNIT is 64 output channels x a tile of XB x 4*RY x 2 outputs (XB = 32, RY = 1 or XB = 16, RY = 2); workgroups are
// persistent (one per CU, 8 waves) and walk units; wave = (z point k, 32-channel block): its four accumulator blocks are the unit's four
// 32-voxel column blocks (XB x RY voxels each) of M_k.  GEMM view per (dy, dx) tap and 16-channel chunk: A = U (32 co x 16 ci, packed
// fragment-ready in global memory, read straight into registers two taps ahead - every wave has its own (k, block), nothing to share
// through LDS), B = V_k (16 ci x 32 voxels: one ds_read_b128 per fragment from the channel-contiguous LDS image
// [hi / lo][k half][point][row][x]).  The halo tile of the next chunk (in a unit's last chunk: of the next unit) is requested plane by
// plane at taps 0..3, each behind that tap's weight request (loads complete in order), cut point by point (z transform, scale, hi / lo)
// and written to the other LDS buffer at taps 4..7: ONE barrier per chunk.  The four M_k meet in an LDS exchange at the end; un-pooled, the
// exchange is transposed so that a lane stores 16 bytes (four x of one channel and row); with the fused pool a wave finishes one row pair
// of half the channels.
//
// Strip mode (PRM back-propagation: the windows of all peaks side by side along x): one operand scale PER WINDOW (col_bound: one float per
// peak at a stride of 32), looked up by the column of a staged halo item and of an output quad, optionally with the prepare step of the
// layer below fused into the epilogue (PREP: m3d_w24::PrepEpi, the fp32 strip kernel's contract).
//
// Operand scales: the input's largest magnitude comes from the PRODUCER (a 32-slot device array of non-negative floats, the largest is
// the bound; this kernel's epilogue fills the array for the next layer: `d_out_max`), so no sweep of the activations is ever needed.
// The f16 MFMA truncates when it adds into the accumulator (conv3d_x3.hip): with signed weights the bias is ~3e-9 of the running sum
// per MFMA, 5e-6 at 256 input channels - below the fp32 Winograd kernels' error; the accumulators run over all of K.
#include <type_traits>

#include "m3d_common.h"
#include "conv3d_wino24.h"      // m3d_w24::PrepEpi: the fused `prepare` epilogue of the PRM strips (same contract as the fp32 kernel's)

// diagnostic build only (make zw_stamps; tools/zw_stamps.py): s_memtime of wave 0 of every workgroup at every tap of its SECOND unit
#ifdef M3D_ZW_STAMPS
static unsigned long long* g_zw_stamps = nullptr;
M3D_API void m3d_debug_set_stamp_buffer_zw(void* p) { g_zw_stamps = (unsigned long long*)p; }
#define ZW_STAMP(k) do { if (a.stamps && it == 1 && tid == 0 && (k) < 256) a.stamps[(size_t)blockIdx.x * 256 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ZW_STAMP(k) do { } while (0)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ZW_NT = 512;
constexpr int ZW_SLOTS = 32;                       // slots of an absmax array
constexpr int ZW_STEP_UNITS = 4 * 2 * 128;         // packed weights of one (chunk, tap): [point 4][block 2][hi / lo][k half][row 32] 16-byte units
constexpr int acc_row(int g) { return (g & 3) + 8 * (g >> 2); }

template <int XB>
struct ZwCfg {
  static constexpr int RY = 32 / XB;
  static constexpr int TX = XB, TY = 4 * RY, TZ = 2;
  static constexpr int HXN = XB + 2, HYN = TY + 2;
  static constexpr int HXP = HXN;                  // row pitch in units
  static constexpr int PLANE = HYN * HXP;
  static constexpr int BUF_UNITS = 16 * PLANE;     // [hi / lo][k half][point]
  static constexpr int ITEMS = 2 * HYN * HXN;      // staging items of a chunk: (8-channel half, halo row, halo column)
  static constexpr int XCH_BYTES = 2 * 4 * 4 * 4 * 64 * 16;      // exchange: [block][point][column block][g / 4][lane] x 16 bytes
  static constexpr int AFF_OFF = 2 * BUF_UNITS * 16 > XCH_BYTES ? 2 * BUF_UNITS * 16 : XCH_BYTES;   // [unit parity][scale 64 | shift 64] floats
  static constexpr int LDS_BYTES = AFF_OFF + 2 * 128 * 4;
  static_assert(ITEMS <= ZW_NT, "one staging item per thread");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// column (lane % 32) of a 16 x 2 column block -> (x, y): the lane groups ds_read_b128 is served in ({0-3,12-15,20-27} / {4-11,16-19,28-31},
// conv3d_x3.hip) each read 16 consecutive units of ONE row
__device__ __forceinline__ void col_xy16(int c, int* x, int* y) {
  if (c < 4) { *x = c; *y = 0; }
  else if (c < 12) { *x = c - 4; *y = 1; }
  else if (c < 16) { *x = c - 8; *y = 0; }
  else if (c < 20) { *x = c - 8; *y = 1; }
  else if (c < 28) { *x = c - 12; *y = 0; }
  else { *x = c - 16; *y = 1; }
}

// packed[cout group 64][step = chunk * 9 + dy * 3 + dx][point][block][hi / lo][k half][row 32] <- weight [cout][cin][3][3][3] fp32
__global__ __launch_bounds__(256) void zw_pack_kernel(const float* __restrict__ w, int cin, int cout, u32x4* __restrict__ packed,
                                                      const float* __restrict__ wamax) {
  float sw, inv;
  m3d::f16_scale_of(1.5f * *wamax, sw, inv);        // |U1|, |U2| <= 1.5 max |g|
  const int chunks = cin / 16, ncg = (cout + 63) / 64;
  const long long total = (long long)ncg * chunks * 9 * 4 * 2 * 64;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int row = (int)(e & 31), kh = (int)((e >> 5) & 1), cb = (int)((e >> 6) & 1), point = (int)((e >> 7) & 3);
    long long r = e >> 9;
    const int tap = (int)(r % 9); r /= 9;
    const int ch = (int)(r % chunks), cg = (int)(r / chunks);
    const int co = cg * 64 + cb * 32 + row;
    u32x4 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16x2 hh, ll;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float v = 0.f;
        if (co < cout) {
          const float* g = w + ((size_t)co * cin + ch * 16 + 8 * kh + 2 * j + u) * 27 + tap;
          const double g0 = g[0], g1 = g[9], g2 = g[18];
          const double t = point == 0 ? g0 : point == 1 ? 0.5 * (g0 + g1 + g2) : point == 2 ? 0.5 * (g0 - g1 + g2) : g2;
          v = (float)t * sw;
        }
        const _Float16 h = (_Float16)v;
        hh[u] = h; ll[u] = (_Float16)(v - (float)h);
      }
      ph[j] = __builtin_bit_cast(unsigned, hh); pl[j] = __builtin_bit_cast(unsigned, ll);
    }
    u32x4* dst = packed + ((size_t)(cg * chunks + ch) * 9 + tap) * ZW_STEP_UNITS + (point * 2 + cb) * 128 + kh * 32 + row;
    dst[0] = ph; dst[64] = pl;
  }
}

struct ZwArgs {
  const float* x; const u32x4* wp; float* out; const float* scale; const float* shift;
  const float* in_max; unsigned* out_max; const float* wamax;
  int B, cin, cout, D, H, W, relu;
  int tiles_x, tiles_y, tiles_z, ncg;
  // strip mode (PRM windows side by side along x, one scale per window): col_bound[p] = largest |input| of window p, cell p = columns
  // [pitch p, pitch (p + 1)) (pitch a multiple of 4: a 16-byte output quad never straddles two cells); col_bound == null: one scale (in_max)
  const float* col_bound; int pitch, npeaks;
  m3d_w24::PrepEpi pe;     // PREP instantiation only
#ifdef M3D_ZW_STAMPS
  unsigned long long* stamps;
#endif
};

__device__ __forceinline__ int zw_xcd_contiguous(int bid, int n) {
  const int per = n >> 3, rem = n & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

template <int XB, bool POOL, bool PREP = false>
__global__ __launch_bounds__(ZW_NT, 2) void conv3d_zw_kernel(ZwArgs a) {
  static_assert(!PREP || (XB == 32 && !POOL), "fused prepare: the un-pooled 32-wide form");
  using C = ZwCfg<XB>;
  static_assert(!POOL || XB == 32, "fused pool: 32-wide column blocks (a row pair = two blocks of the finishing wave)");
  extern __shared__ float lds_f[];
  u32x4* const lds = reinterpret_cast<u32x4*>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int point = wave & 3, cb = wave >> 2;
  const size_t HW = (size_t)a.H * a.W, DHW = HW * a.D;
  const int chunks = a.cin / 16, steps = chunks * 9;
  const int ch_bytes = (int)(DHW * sizeof(float));

  // ---- operand scales (powers of two): input bound = the largest slot x 2 (a V is a sum of two planes), weights as packed
  float xs, inv_x, inv_w;
  {
    float im = a.col_bound ? 0.f : a.in_max[lane & (ZW_SLOTS - 1)];      // (strip mode: one scale per window, see Stage::xs)
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) im = fmaxf(im, __shfl_xor(im, o));
    float sw_;
    m3d::f16_scale_of(2.f * im, xs, inv_x);
    m3d::f16_scale_of(1.5f * *a.wamax, sw_, inv_w);
  }
  const float un = inv_x * inv_w;

  // ---- units: (cout group fastest, x, y, z, batch item); a workgroup walks units l0, l0 + grid, ... (persistent: the next unit's first
  // halo tile and weight fragments are requested during the current unit's last chunk, so that only the first unit of a workgroup waits
  // for a global load).  Neighbouring workgroups of an XCD hold neighbouring units: shared halos and weights in that XCD's L2.
  const int units = a.ncg * a.tiles_x * a.tiles_y * a.tiles_z * a.B;
  struct Unit { int cg, x0, y0, z0, b; };
  auto decode = [&](int u) __attribute__((always_inline)) {
    Unit r;
    r.cg = u % a.ncg; u /= a.ncg;
    r.x0 = (u % a.tiles_x) * C::TX; u /= a.tiles_x;
    r.y0 = (u % a.tiles_y) * C::TY; u /= a.tiles_y;
    r.z0 = (u % a.tiles_z) * C::TZ;
    r.b = u / a.tiles_z;
    return r;
  };

  // ---- staging item of this thread: (8-channel half g, halo row hy, halo column hx); the four planes z0 - 1 .. z0 + 2.
  // 32-bit buffer offsets inside one batch item (host: cin * D * H * W * 4 < 2^31): four per-plane byte offsets, channel and chunk scalar.
  const bool has = tid < C::ITEMS;
  const int sg = has ? tid / (C::HYN * C::HXN) : 0;
  const int sr = has ? tid % (C::HYN * C::HXN) : 0;
  const int shy = sr / C::HXN, shx = sr % C::HXN;
  struct Stage { int voff[4]; int okm; const float* base; float xs; };
  auto stage_of = [&](const Unit& q) __attribute__((always_inline)) {
    Stage st;
    const int sy = q.y0 - 1 + shy, sx = q.x0 - 1 + shx;
    const bool okyx = has & (sy >= 0) & (sy < a.H) & (sx >= 0) & (sx < a.W);
    st.okm = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int z = q.z0 - 1 + k;
      const bool ok = okyx & (z >= 0) & (z < a.D);
      st.okm |= ok ? (1 << k) : 0;
      // outside the volume: an offset beyond the buffer's num_records (< 2^31; offset + any scalar part stays below 2^32) - the load returns 0,
      // which is the zero padding: no per-value select when the tile is cut
      st.voff[k] = ok ? (int)(((size_t)z * HW + (size_t)sy * a.W + sx + (size_t)8 * sg * DHW) * 4) : 0x7FFFFF00;
    }
    st.base = a.x + (size_t)q.b * a.cin * DHW;
    st.xs = xs;
    if (a.col_bound) {                                  // this halo column's window: its own power-of-two scale
      const int p = min(max(sx, 0) / a.pitch, a.npeaks - 1);
      float inv_;
      m3d::f16_scale_of(2.f * a.col_bound[32 * p], st.xs, inv_);
    }
    return st;
  };
  float raw[4][8];
  // plane k of the staged item's four (8 dword loads).  The K loop requests one plane per tap, BEHIND that tap's weight-fragment request:
  // loads complete in order, so a wait for weight fragments is a wait for every plane requested before them
  auto fetch_plane = [&](const Stage& st, int c, const int k) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.base), 0,
                                                                        (unsigned)((size_t)a.cin * DHW * sizeof(float)), 0x00020000);
    const int cbase = c * 16 * ch_bytes;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      raw[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, st.voff[k], cbase + j * ch_bytes, 0));
  };
  auto fetch_in = [&](const Stage& st, int c) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) fetch_plane(st, c, k);
  };
  const int st_unit = (sg * 4) * C::PLANE + shy * C::HXP + shx;               // + point * PLANE + hilo * 8 * PLANE + buf * BUF_UNITS
  // one z point k of the staged item: V_k of its 8 channels, scaled and cut -> two 16-byte units.  Called point by point from different
  // taps (40 VALU each ride in the MFMAs' shadow; all four at once stalled both waves of a SIMD at the same tap)
  auto commit_point = [&](int buf, float xs_, const int k) __attribute__((always_inline)) {
    // branch-free (the tap that carries it must stay one scheduling region): threads without an item write to a dump unit behind the two
    // staging buffers (inside the exchange area, which nobody reads before the K loop's closing barrier)
    constexpr int PA_[4] = {0, 1, 2, 1}, PB_[4] = {2, 2, 1, 3};               // V_k = d[PA] -+ d[PB]: d0 - d2, d1 + d2, d2 - d1, d1 - d3
    u32x4 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16x2 hh, ll;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ch = 2 * j + u;
        const float da = raw[PA_[k]][ch], db = raw[PB_[k]][ch];
        const float v = (k == 1 ? da + db : da - db) * xs_;
        const _Float16 h = (_Float16)v;
        hh[u] = h; ll[u] = (_Float16)(v - (float)h);
      }
      ph[j] = __builtin_bit_cast(unsigned, hh); pl[j] = __builtin_bit_cast(unsigned, ll);
    }
    u32x4* d = lds + (has ? buf * C::BUF_UNITS + st_unit + k * C::PLANE : 2 * C::BUF_UNITS);
    d[0] = ph; d[8 * C::PLANE] = pl;
  };
  static_assert((2 * C::BUF_UNITS + 8 * C::PLANE + 1) * 16 <= C::XCH_BYTES, "the dump units lie inside the exchange area");
  auto commit_in = [&](int buf, float xs_) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) commit_point(buf, xs_, k);
  };

  // ---- fragments
  const int fr = lane & 31, fh = lane >> 5;
  int cx, cy;
  if constexpr (XB == 32) { cx = fr; cy = 0; } else { col_xy16(fr, &cx, &cy); }
  const int bB = (fh * 4 + point) * C::PLANE + cy * C::HXP + cx;              // + (RY * j + dy) * HXP + dx + piece * 8 * PLANE + buf * BUF_UNITS
  // weight fragments: buffer loads with the (cout group, step) part of the address in the scalar offset - no 64-bit vector address arithmetic
  // in the K loop.  wsrc = byte offset of a cout group's first step.
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(a.wp), 0, (unsigned)((size_t)a.ncg * chunks * 9 * ZW_STEP_UNITS * 16), 0x00020000);
  const int w_lane = ((point * 2 + cb) * 128 + lane) * 16;
  auto wsrc_of = [&](int cg) __attribute__((always_inline)) { return cg * chunks * 9 * (ZW_STEP_UNITS * 16); };

  struct BF { f16x8 b[4][2]; };
  BF F0, F1;
  f16x8 A[3][2];                                                             // A of step s: set s % 3 (9 taps per chunk: the tap's index % 3)
  auto read_b = [&](BF& f, int i, int t, int buf) __attribute__((always_inline)) {     // read i of 8: column block i >> 1, piece i & 1
    const int j = i >> 1, p = i & 1, dy = t / 3, dx = t % 3;
    f.b[j][p] = __builtin_bit_cast(f16x8, lds[buf * C::BUF_UNITS + bB + (C::RY * j + dy) * C::HXP + dx + p * 8 * C::PLANE]);
  };
  auto fetch_a = [&](int step_bytes, int set) __attribute__((always_inline)) {
    A[set][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, step_bytes, 0));
    A[set][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane + 1024, step_bytes, 0));
  };

  const int l0 = zw_xcd_contiguous(blockIdx.x, gridDim.x);
  int u = l0;
  Unit cur = decode(u);
  Stage S = stage_of(cur);
  int wsrc = wsrc_of(cur.cg);
  fetch_in(S, 0);
  fetch_a(wsrc, 0);
  fetch_a(wsrc + ZW_STEP_UNITS * 16, 1);
  float vmax = 0.f;
  f32x4* const xch = reinterpret_cast<f32x4*>(lds_f);

  for (int it = 0;; ++it) {
    const int un_ = u + (int)gridDim.x;
    const bool hasn = un_ < units;
    const Unit nxtu = decode(hasn ? un_ : u);
    const Stage SN = stage_of(nxtu);
    const int wsrcN = wsrc_of(nxtu.cg);

    float inv_e = 1.f;                                   // strip mode: the inverse input scale of the window this lane's output quad lies in
    if (a.col_bound) {
      const int p = min((cur.x0 + 4 * ((lane & 31) % (XB / 4))) / a.pitch, a.npeaks - 1);
      float s_;
      m3d::f16_scale_of(2.f * a.col_bound[32 * p], s_, inv_e);
    }
    ZW_STAMP(0);
    commit_in(0, S.xs);
    float* const aff = reinterpret_cast<float*>(reinterpret_cast<char*>(lds_f) + C::AFF_OFF) + (it & 1) * 128;
    if (tid < 64) {                                      // the unit's 64 channels: scale x the operand scales' inverse, shift
      const int co = min(cur.cg * 64 + tid, a.cout - 1);
      aff[tid] = (a.scale ? a.scale[co] : 1.f) * (a.col_bound ? inv_w : un);
      aff[64 + tid] = a.shift ? a.shift[co] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) read_b(F0, i, 0, 0);

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;

    // one chunk = 9 taps on LDS buffer `par` (9 is odd: the chunk's parity = the parity of its first step), a literal at both call sites.
    // During the 12 MFMAs of a tap the 8 B fragments of the next tap are read (3 / 3 / 2 per four MFMAs) and the A fragments of the tap
    // after it are requested (beyond the unit's last step: the NEXT unit's first two).  The next chunk's tile: loads at tap 0 (last
    // chunk: the next unit's chunk 0, which stays in registers through the epilogue), cut + LDS writes at taps 4..7, barrier before tap 8.
    auto run_chunk = [&](int c, const int par) __attribute__((always_inline)) {
      const int buf = par;
      const bool last = c + 1 >= chunks;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int s = c * 9 + t;
        BF& curf = ((par + t) & 1) ? F1 : F0;
        BF& nxtf = ((par + t) & 1) ? F0 : F1;
        ZW_STAMP(1 + s);
        if (t == 8) __syncthreads();
        {                                                // weight fragments two taps ahead (set (t + 2) % 3 was consumed by tap t - 1)
          const int idx = s + 2;
          const bool over = idx >= steps;
          fetch_a((over ? wsrcN : wsrc) + (over ? idx - steps : idx) * (ZW_STEP_UNITS * 16), (t + 2) % 3);
        }
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};                   // small products first: (lo, hi) (hi, lo) (hi, hi)
        constexpr int RD0[4] = {0, 3, 6, 8};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int i = RD0[q]; i < RD0[q + 1]; ++i) read_b(nxtf, i, t < 8 ? t + 1 : 0, t < 8 ? buf : buf ^ 1);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[t % 3][PA[q]], curf.b[j][PB[q]], acc[j], 0, 0, 0);
        }
        // The next chunk's halo planes, each behind a tap's weight request (loads complete in order): planes 2, 0, 1, 3 at tap 8 of the
        // PREVIOUS chunk (plane 2's registers are free from tap 7 on) and taps 0, 1, 2 - their points are cut at taps 4..7, and tap 4 used
        // to wait ~700 cycles for DRAM (tools/zw_stamps.py).  A unit's first chunk has no previous tap 8: it requests two planes at tap 0.
        auto plane = [&](const int k, int cc, bool nextu) __attribute__((always_inline)) {
          Stage q;
          q.voff[k] = nextu ? SN.voff[k] : S.voff[k];
          q.base = nextu ? SN.base : S.base; q.okm = 0; q.xs = 0.f;
          fetch_plane(q, cc, k);
        };
        if (t == 0 && c == 0) plane(2, last ? 0 : 1, last);
        if (t == 0) plane(0, last ? 0 : c + 1, last);
        if (t == 1) plane(1, last ? 0 : c + 1, last);
        if (t == 2) plane(3, last ? 0 : c + 1, last);
        if (t == 8 && !last) { const bool nu = c + 2 >= chunks; plane(2, nu ? 0 : c + 2, nu); }
        if (t >= 4 && t <= 7 && !last) {                 // points 0 (planes 0, 2), 2 (2, 1), 1 (1, 2), 3 (1, 3)
          constexpr int PT[4] = {0, 2, 1, 3};
          commit_point(buf ^ 1, S.xs, PT[t - 4]);
        }
        if (t >= 4 && t <= 7) {                          // the cut's ~40 VALU between the MFMAs, three per MFMA (an MFMA holds the issue 8 of its 32 cycles)
#define ZW_MV4 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); \
               __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); \
               __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); \
               __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); ZW_MV4
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); ZW_MV4
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); ZW_MV4
#undef ZW_MV4
          __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
      }
    };
#pragma unroll 1
    for (int c = 0; c < chunks; c += 2) {
      run_chunk(c, 0);
      if (c + 1 < chunks) run_chunk(c + 1, 1);
    }

    ZW_STAMP(200);
    // ---- the four M_k meet in LDS
    __syncthreads();
    const int x0 = cur.x0, y0 = cur.y0, z0 = cur.z0;
    float* const ob = a.out + (size_t)cur.b * a.cout * (POOL ? DHW / 8 : DHW);
    if constexpr (POOL) {                                // [block][point][column block][g / 4][lane] x 16 bytes: readers keep the writers' lanes
      f32x4* xw = xch + ((size_t)((cb * 4 + point) * 4) * 4) * 64 + lane;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          xw[(j * 4 + gq) * 64] = f32x4{acc[j][4 * gq], acc[j][4 * gq + 1], acc[j][4 * gq + 2], acc[j][4 * gq + 3]};
    } else {
      // TRANSPOSED: [point][channel 64][row 4 RY][x XB] floats (128 per channel), so that a reader lane owns four consecutive x of one
      // (channel, row) and stores 16 bytes per plane - the un-pooled epilogue was bound by the issue rate of its 256 dword wave-stores
      // per unit.  The lanes of k half 1 (channel + 4) write 32 floats further (XOR 32 inside the channel's 128): no bank conflicts.
      float* xt = lds_f + ((size_t)point * 64 + cb * 32 + 4 * fh) * 128 + (XB == 32 ? cx : 16 * cy + cx);     // + channel row, + 32 j (^ 32 for k half 1)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 16; ++g)
          xt[acc_row(g) * 128 + ((32 * j) ^ (fh << 5))] = acc[j][g];
    }
    __syncthreads();
    if constexpr (!POOL) {
      // thread -> items i = 0..3: channel 16 i + 2 wave + (lane >> 5), (row, x quad) from lane & 31; both planes
      constexpr int QX = XB / 4;                        // x quads per row
      const int ridx = lane & 31, rrow = ridx / QX, rxq = ridx % QX;
      const int x = x0 + 4 * rxq, y = y0 + rrow;
      f32x4 o0[4], o1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cl = 16 * i + 2 * wave + (lane >> 5);
        const float* src = lds_f + (size_t)cl * 128 + ((rrow * XB + 4 * rxq) ^ (((cl >> 2) & 1) << 5));
        f32x4 m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = *reinterpret_cast<const f32x4*>(src + (size_t)k * 64 * 128);
        o0[i] = (m[0] + m[1]) + m[2]; o1[i] = (m[1] - m[2]) - m[3];
      }
      if (hasn) __syncthreads();                       // every wave has its sums: the staging buffers are free for the next unit
      if constexpr (PREP) {
        // ---- fused `prepare` of the layer below (m3d_w24::PrepEpi; the element rule of prm_prepare_kernel, evaluated in the same order on
        // the value the un-fused launch would have stored: the two-launch path's strip bit for bit).  Strip column x -> (peak, window column).
        const m3d_w24::PrepEpi& pe = a.pe;
        constexpr float kEpsP = 1e-10f;                                    // peak_backprop_3d.py:29
        const int p = (int)(((float)x + 0.5f) * pe.inv_pitchA);            // exact: x < 2^22
        if (p < pe.P && y < a.H && x < a.W) {
          const int cxw = x - p * pe.pitchA, ix0 = cxw - pe.leadA;
          const int oz = pe.origin[3 * p], oy = pe.origin[3 * p + 1], ox = pe.origin[3 * p + 2];
          const float xoff = *pe.xoff;
          const long long colB = (long long)p * pe.pitchB + pe.leadB + ix0 + 1;       // B column of the quad's first element
          const bool quadB = ((colB & 3) == 0) && colB >= 0 && colB + 3 < pe.LB;
          const int MHW = pe.MH * pe.MW, MV = pe.MD * MHW;
          const int qy = oy + y;
          bool okx[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) { const int ix = ix0 + c, qx = ox + ix; okx[c] = (ix >= 0) & (ix < pe.U) & (qx >= 0) & (qx < pe.MW); }
          const bool allx = okx[0] & okx[1] & okx[2] & okx[3], anyx = okx[0] | okx[1] | okx[2] | okx[3];
          const bool wave_fast = __builtin_amdgcn_ballot_w64(anyx & !allx) == 0ull;
          // per plane: validity and map row
          bool okp[2]; int zB[2], rbase[2];
#pragma unroll
          for (int zz = 0; zz < 2; ++zz) {
            const int z = z0 + zz;
            if (z == 0 && y == 0 && cxw == 0 && cur.cg == 0 && wave == 0 && (lane >> 5) == 0 && zz == 0) {
              pe.origin_out[3 * p] = oz - 1; pe.origin_out[3 * p + 1] = oy - 1; pe.origin_out[3 * p + 2] = ox - 1;
            }
            const int iz = pe.slabA ? z - oz : z;
            const int qz = oz + iz;
            bool ok = (z < a.D) & (iz >= 0) & (iz < pe.U);
            if (pe.slabB) ok = ok & (qz >= 0) & (qz < pe.MD);               // B stores the map's planes only
            okp[zz] = ok;
            zB[zz] = pe.slabB ? qz : iz + 1;
            const bool okr = ok & (qz >= 0) & (qz < pe.MD) & (qy >= 0) & (qy < pe.MH);
            rbase[zz] = okr ? qz * MHW + qy * pe.MW : -1;
          }
          typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
          f32x4 xn[4][2], nn[4][2];
          float scv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int co = cur.cg * 64 + 16 * i + 2 * wave + (lane >> 5);
            const bool okc = co < a.cout;
            scv[i] = (pe.scale && okc) ? pe.scale[co] : 1.f;
            const int cbm = okc ? co * MV : 0;                               // (host: cout * MV < 2^31)
#pragma unroll
            for (int zz = 0; zz < 2; ++zz) {
              if (wave_fast) {
                const int pos = cbm + ((rbase[zz] >= 0 && allx) ? rbase[zz] + ox + ix0 : 0);
                xn[i][zz] = *reinterpret_cast<const f32x4u*>(pe.xnext + pos);
                nn[i][zz] = *reinterpret_cast<const f32x4u*>(pe.norm + pos);
              } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  const int pos = cbm + ((rbase[zz] >= 0 && okx[c]) ? rbase[zz] + ox + ix0 + c : 0);
                  xn[i][zz][c] = pe.xnext[pos];
                  nn[i][zz][c] = pe.norm[pos];
                }
              }
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cl = 16 * i + 2 * wave + (lane >> 5), co = cur.cg * 64 + cl;
            if (co >= a.cout) continue;
            const float sc = aff[cl] * inv_e;
#pragma unroll
            for (int zz = 0; zz < 2; ++zz) {
              if (!okp[zz]) continue;
              float g[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                float v = (zz ? o1[i][c] : o0[i][c]) * sc;                  // what the un-fused launch stores (powers of two: exact)
                v = (xn[i][zz][c] - xoff) * v;                              // PreHook of the layer just back-propagated (:16-18)
                if (!(xn[i][zz][c] > 0.f)) v = 0.f;                         // ReLU backward
                if (pe.scale) v = v * scv[i];                               // eval-BatchNorm backward
                v = (nn[i][zz][c] < kEpsP) ? 0.f : v / (fabsf(nn[i][zz][c]) + kEpsP);   // PostHook (:30-33)
                g[c] = (rbase[zz] >= 0 && okx[c]) ? v : 0.f;
              }
              float* dst = a.out + (size_t)co * pe.ocs + (size_t)zB[zz] * pe.ozs + (size_t)(y + 1) * pe.LB + colB;
              if (quadB) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{g[0], g[1], g[2], g[3]};
              } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {                               // B window columns 0 .. U + 1 only (others belong to other cells)
                  const int ix = ix0 + c;
                  if (ix >= -1 && ix <= pe.U && colB + c >= 0 && colB + c < pe.LB) dst[c] = g[c];
                }
              }
            }
          }
        }
      } else {
      const bool quad_ok = ((a.W & 3) == 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cl = 16 * i + 2 * wave + (lane >> 5), co = cur.cg * 64 + cl;
        if (!(y < a.H && x < a.W && co < a.cout)) continue;
        const float sc = aff[cl] * inv_e, sh = aff[64 + cl];        // (inv_e: a power of two, 1 outside strip mode)
#pragma unroll
        for (int zz = 0; zz < 2; ++zz) {
          if (z0 + zz >= a.D) continue;
          f32x4 v = (zz ? o1[i] : o0[i]) * sc + sh;
#pragma unroll
          for (int e = 0; e < 4; ++e) if (a.relu) v[e] = fmaxf(v[e], 0.f);
          float* o = ob + (size_t)co * DHW + (size_t)(z0 + zz) * HW + (size_t)y * a.W + x;
          if (quad_ok) {                                 // (x is a multiple of 4 and W is: the quad is inside the row)
            *reinterpret_cast<f32x4*>(o) = v;
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (x + e < a.W) { o[e] = v[e]; vmax = fmaxf(vmax, fabsf(v[e])); }
          }
        }
      }
      }
    } else {
      // fused MaxPool3d(2,2): wave (point p, block) finishes row pair p & 1 (column blocks 2 rp, 2 rp + 1) for channel quads 2 (p >> 1) + {0, 1};
      // z pair and y pair in the lane, x pair in lanes x, x ^ 1
      const int rp = point & 1, gh = point >> 1;
      const int PD = a.D / 2, PH = a.H / 2, PW = a.W / 2;
      const int x = x0 + cx, yp = (y0 >> 1) + rp, zp = z0 >> 1, xp = x >> 1;
      f32x4 o0[2][2], o1[2][2];
#pragma unroll
      for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int gq = 2 * gh + gi, j = 2 * rp + jj;
          f32x4 m[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) m[k] = xch[((size_t)((cb * 4 + k) * 4 + j) * 4 + gq) * 64 + lane];
          o0[gi][jj] = (m[0] + m[1]) + m[2]; o1[gi][jj] = (m[1] - m[2]) - m[3];
        }
      if (hasn) __syncthreads();
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        const int gq = 2 * gh + gi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cl = cb * 32 + 4 * fh + acc_row(4 * gq + e), co = cur.cg * 64 + cl;
          const float sc = aff[cl], sh = aff[64 + cl];
          float best = -INFINITY;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            float v0 = o0[gi][jj][e] * sc + sh, v1 = o1[gi][jj][e] * sc + sh;
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            best = fmaxf(best, fmaxf(v0, v1));
          }
          const float v = fmaxf(best, __shfl_xor(best, 1));
          if ((lane & 1) == 0 && co < a.cout && xp < PW && yp < PH && zp < PD) {
            vmax = fmaxf(vmax, fabsf(v));
            ob[(size_t)co * ((size_t)PD * PH * PW) + ((size_t)zp * PH + yp) * PW + xp] = v;
          }
        }
      }
    }
    ZW_STAMP(202);
    if (!hasn) break;
    u = un_; cur = nxtu; S = SN; wsrc = wsrcN;
  }
  // ---- the next layer's operand bound: one atomic per workgroup into slot (block % 32)
  if (a.out_max) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    __shared__ float wm[8];
    if (lane == 0) wm[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
      float m = wm[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) m = fmaxf(m, wm[k]);
      if (m > 0.f) atomicMax(a.out_max + (blockIdx.x & (ZW_SLOTS - 1)), __float_as_uint(m));
    }
  }
}

template <int XB, bool POOL, bool PREP = false>
int launch_zw(ZwArgs a, hipStream_t st) {
  using C = ZwCfg<XB>;
  a.tiles_x = (a.W + C::TX - 1) / C::TX; a.tiles_y = (a.H + C::TY - 1) / C::TY; a.tiles_z = (a.D + C::TZ - 1) / C::TZ;
  a.ncg = (a.cout + 63) / 64;
  const long long units = (long long)a.ncg * a.tiles_x * a.tiles_y * a.tiles_z * a.B;
  if (units > 0x7FFFFFFFll) return M3D_EUNSUPPORTED;
  // persistent workgroups, one per CU (131 KB of LDS each): every workgroup the same number of units where the count allows it
  static int cus = 0;
  if (!cus) {
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const long long rounds = (units + cus - 1) / cus;
  const long long blocks = (units + rounds - 1) / rounds;
#ifdef M3D_ZW_STAMPS
  a.stamps = g_zw_stamps;
#endif
  auto kern = conv3d_zw_kernel<XB, POOL, PREP>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(ZW_NT), C::LDS_BYTES, st, a);
  return m3d::check_launch("conv3d_zw");
}

inline size_t zw_plane_bytes(int cin, int cout) { return (size_t)((cout + 63) / 64) * (cin / 16) * 9 * ZW_STEP_UNITS * 16; }

}  // namespace

M3D_API int m3d_conv3d_zw_supported(int cin, int cout, int depth, int height, int width, int pool) {
  if (cin <= 0 || cout <= 0 || cin % 16 != 0 || depth < 2 || height < 4 || width < 12) return 0;
  if ((size_t)depth * height * width >= 0x7FFFFFFFull) return 0;
  if (pool && ((depth | height | width) & 1)) return 0;
  if (pool && width < 24) return 0;                  // the fused pool exists for the 32-wide column blocks
  return 1;
}

M3D_API size_t m3d_conv3d_zw_packed_bytes(int cin, int cout) {
  if (cin <= 0 || cout <= 0 || cin % 16 != 0) return 0;
  return zw_plane_bytes(cin, cout) + 256;            // + the weight's largest magnitude (one float) behind the fragments
}

M3D_API int m3d_conv3d_zw_pack(const float* d_weight, int cin, int cout, void* d_packed, void* stream) {
  if (!d_weight || !d_packed || cin <= 0 || cout <= 0 || cin % 16 != 0) return M3D_EINVAL;
  float* wamax = reinterpret_cast<float*>(static_cast<char*>(d_packed) + zw_plane_bytes(cin, cout));
  if (const int rc = m3d_absmax(d_weight, (long long)cout * cin * 27, wamax, stream)) return rc;
  hipLaunchKernelGGL(zw_pack_kernel, dim3(1024), dim3(256), 0, m3d::as_stream(stream), d_weight, cin, cout, (u32x4*)d_packed, (const float*)wamax);
  return m3d::check_launch("conv3d_zw_pack");
}

M3D_API int m3d_conv3d_zw_slots(void) { return ZW_SLOTS; }

/* d_slots[0] = max |x|, the other 31 slots 0: an operand bound for m3d_conv3d_zw_forward from a sweep of x (the first layer of a chain) */
M3D_API int m3d_conv3d_zw_bound_of(const float* d_x, long long n, float* d_slots, void* stream) {
  if (!d_slots) return M3D_EINVAL;
  hipStream_t st = m3d::as_stream(stream);
  if (hipMemsetAsync(d_slots, 0, ZW_SLOTS * sizeof(float), st) != hipSuccess) return M3D_ELAUNCH;
  return m3d_absmax(d_x, n, d_slots, stream);
}

namespace {
// largest |value| of every window of a strip [rows][L] (rows = channels x planes x window rows), cell p = columns [pitch p, pitch (p + 1)):
// lanes walk a row (coalesced), maxima meet in an LDS array per workgroup, one global atomic per (workgroup, window)
__global__ __launch_bounds__(256) void zw_strip_absmax_kernel(const float* __restrict__ x, long long rows, int L, int pitch, int P,
                                                              unsigned* __restrict__ out) {
  extern __shared__ unsigned pm[];
  for (int i = threadIdx.x; i < P; i += 256) pm[i] = 0;
  __syncthreads();
  const int L4 = L / 4;                                  // (L is a multiple of 4 and pitch is: a quad lies in one cell)
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const f32x4* row = reinterpret_cast<const f32x4*>(x + r * L);
    for (int q = threadIdx.x; q < L4; q += 256) {
      const f32x4 v = row[q];
      const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
      if (m > 0.f) atomicMax(&pm[min(4 * q / pitch, P - 1)], __float_as_uint(m));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P; i += 256)
    if (pm[i]) atomicMax(out + 32 * i, pm[i]);
}
}  // namespace

/* d_out[32 p] (one cache line per window: the producers' atomic maxima do not queue on one line) = largest |x| inside cell p (columns [pitch p, pitch (p + 1))) of the strip d_strip [rows][L]; L and pitch multiples of 4 */
M3D_API int m3d_prm_strip_absmax(const float* d_strip, long long rows, int L, int pitch, int num_peaks, float* d_out, void* stream) {
  if (!d_strip || !d_out || rows <= 0 || L <= 0 || pitch <= 0 || num_peaks <= 0 || (L & 3) || (pitch & 3) || num_peaks > 12288) return M3D_EINVAL;
  hipStream_t st = m3d::as_stream(stream);
  if (hipMemsetAsync(d_out, 0, (size_t)num_peaks * 32 * sizeof(float), st) != hipSuccess) return M3D_ELAUNCH;
  const long long blocks = rows < 2048 ? rows : 2048;
  hipLaunchKernelGGL(zw_strip_absmax_kernel, dim3((unsigned)blocks), dim3(256), (size_t)num_peaks * 4, st, d_strip, rows, L, pitch, num_peaks,
                     reinterpret_cast<unsigned*>(d_out));
  return m3d::check_launch("prm_strip_absmax");
}

/* The convolution of m3d_conv3d_zw_forward on a PRM window strip [cin, depth, height, width] (windows side by side along x, cell p =
 * columns [pitch p, pitch (p + 1)), pitch a multiple of 4) with ONE OPERAND SCALE PER WINDOW: d_col_bound [num_peaks x 32], element 32 p = the largest
 * |input| of window p (m3d_prm_strip_absmax, or m3d_prm_prepare_ex3's d_peak_max).  Windows of very different magnitude keep their own 22 bits, and a window's outputs do not depend on
 * which other windows share the strip (a sub-batch gives the batch's values bit for bit).  No scale / shift / ReLU / pool. */
M3D_API int m3d_conv3d_zw_forward_strip(const float* d_in, const void* d_packed, float* d_out, int cin, int cout, int depth, int height,
                                        int width, const float* d_col_bound, int num_peaks, int pitch, void* stream) {
  if (!d_in || !d_packed || !d_out || !d_col_bound || num_peaks <= 0 || pitch <= 0 || (pitch & 3)) return M3D_EINVAL;
  if (!m3d_conv3d_zw_supported(cin, cout, depth, height, width, 0) || width < 24) return M3D_EUNSUPPORTED;
  if ((size_t)cin * depth * height * width * sizeof(float) >= 0x7FFFFF00ull) return M3D_EUNSUPPORTED;
  ZwArgs a{};
  a.x = d_in; a.wp = static_cast<const u32x4*>(d_packed); a.out = d_out;
  a.in_max = d_col_bound;                                // (read, not used: the per-window scales replace it)
  a.wamax = reinterpret_cast<const float*>(static_cast<const char*>(d_packed) + zw_plane_bytes(cin, cout));
  a.B = 1; a.cin = cin; a.cout = cout; a.D = depth; a.H = height; a.W = width;
  a.col_bound = d_col_bound; a.pitch = pitch; a.npeaks = num_peaks;
  return launch_zw<32, false>(a, m3d::as_stream(stream));
}

/* m3d_conv3d_zw_forward_strip FUSED with the prepare step of the layer below - the contract of m3d_prm_strip_dgrad_prepare (fp32 F(2x4,3x3)
 * kernel), arguments as there plus d_col_bound: d_gn = the prepared gradient strip [cin, in_planes, window, L(window)], d_out = the layer
 * below's prepared strip [cout, out_planes, window + 2, L(window + 2)] (zero-filled here), d_origin_out = d_origin - 1.  The values are
 * those of the two launches (strip conv, then m3d_prm_prepare_ex2 with pool = 0, border = 1) bit for bit. */
M3D_API int m3d_prm_strip_dgrad_prepare_zw(const float* d_gn, const void* d_packed, int cin, int cout, int num_peaks, int window, int in_slab,
                                           const int32_t* d_origin, const float* d_xnext, const float* d_norm, const float* d_scale,
                                           const float* d_up_offset, int depth, int height, int width, int out_slab,
                                           const float* d_col_bound, float* d_out, int32_t* d_origin_out, void* stream) {
  if (num_peaks < 0 || cin <= 0 || cout <= 0 || window <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_gn || !d_packed || !d_origin || !d_xnext || !d_norm || !d_up_offset || !d_out || !d_origin_out || !d_col_bound) return M3D_EINVAL;
  int pitchA, leadA, pitchB, leadB; long long LA, LB;
  m3d::strip_geom(window, 2, num_peaks, &pitchA, &leadA, &LA);
  m3d::strip_geom(window + 2, 2, num_peaks, &pitchB, &leadB, &LB);
  const int ZA = in_slab ? depth : window, ZB = out_slab ? depth : window + 2;
  if (!m3d_conv3d_zw_supported(cin, cout, ZA, window, (int)LA, 0) || LA < 24 || (pitchA & 3)) return M3D_EUNSUPPORTED;
  if ((size_t)cin * ZA * window * LA * sizeof(float) >= 0x7FFFFF00ull || (long long)(window + 2) * LB >= 0x7FFFFFFFll || LA >= (1 << 22))
    return M3D_EUNSUPPORTED;
  if ((long long)cout * depth * height * width >= 0x7FFFFFFFll || width < 4) return M3D_EUNSUPPORTED;   // 32-bit map offsets, 16-byte row reads
  hipStream_t st = m3d::as_stream(stream);
  const size_t out_bytes = (size_t)cout * ZB * (window + 2) * LB * sizeof(float);
  if (hipMemsetAsync(d_out, 0, out_bytes, st) != hipSuccess) return M3D_ELAUNCH;
  ZwArgs a{};
  a.x = d_gn; a.wp = static_cast<const u32x4*>(d_packed); a.out = d_out;
  a.in_max = d_col_bound;
  a.wamax = reinterpret_cast<const float*>(static_cast<const char*>(d_packed) + zw_plane_bytes(cin, cout));
  a.B = 1; a.cin = cin; a.cout = cout; a.D = ZA; a.H = window; a.W = (int)LA;
  a.col_bound = d_col_bound; a.pitch = pitchA; a.npeaks = num_peaks;
  m3d_w24::PrepEpi& pe = a.pe;
  pe.xnext = d_xnext; pe.norm = d_norm; pe.scale = d_scale; pe.xoff = d_up_offset; pe.origin = d_origin; pe.origin_out = d_origin_out;
  pe.P = num_peaks; pe.U = window; pe.MD = depth; pe.MH = height; pe.MW = width;
  pe.pitchA = pitchA; pe.leadA = leadA; pe.slabA = in_slab ? 1 : 0;
  pe.pitchB = pitchB; pe.leadB = leadB; pe.slabB = out_slab ? 1 : 0;
  pe.LB = LB; pe.ocs = (long long)ZB * (window + 2) * LB; pe.ozs = (int)((window + 2) * LB);
  pe.inv_pitchA = 1.0f / (float)pitchA;
  return launch_zw<32, false, true>(a, st);
}

M3D_API int m3d_conv3d_zw_forward(const float* d_in, const void* d_packed, float* d_out, int batch, int cin, int cout, int depth, int height,
                                  int width, const float* d_scale, const float* d_shift, int relu, int pool, const float* d_in_max,
                                  float* d_out_max, void* stream) {
  if (!d_in || !d_packed || !d_out || !d_in_max || batch <= 0) return M3D_EINVAL;
  if (!m3d_conv3d_zw_supported(cin, cout, depth, height, width, pool)) return M3D_EUNSUPPORTED;
  if ((size_t)cin * depth * height * width * sizeof(float) >= 0x7FFFFF00ull) return M3D_EUNSUPPORTED;      // 32-bit buffer offsets inside one batch item
  ZwArgs a{};
  a.x = d_in; a.wp = static_cast<const u32x4*>(d_packed); a.out = d_out; a.scale = d_scale; a.shift = d_shift;
  a.in_max = d_in_max; a.out_max = reinterpret_cast<unsigned*>(d_out_max);
  a.wamax = reinterpret_cast<const float*>(static_cast<const char*>(d_packed) + zw_plane_bytes(cin, cout));
  a.B = batch; a.cin = cin; a.cout = cout; a.D = depth; a.H = height; a.W = width; a.relu = relu;
  hipStream_t st = m3d::as_stream(stream);
  if (pool) return launch_zw<32, true>(a, st);
  if (width >= 24) return launch_zw<32, false>(a, st);
  return launch_zw<16, false>(a, st);
}
