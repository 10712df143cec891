// Peak-response back-propagation through the stem (conv1a: 5x5x5, 1 -> 32 channels, followed by MaxPool3d(2,2)) on the
// matrix cores, with the max-unpool / ReLU / BatchNorm / PostHook "prepare" step fused into the operand staging.
//
// Reference: one autograd backward per kept detection (lib/prm/peak_response_mapping_3d.py:157-172) through
// lib/prm/peak_backprop_3d.py:8-44; for conv1a that is  out[v] = (data[v] - off) * sum_{c,t} relu(W)[c][124 - t] * G_N[c][v + t - 2]
// with G_N = unpool(G_up masked by ReLU) * bn_scale / (|N| + 1e-10)  (zero where N < 1e-10).
//
// The backward-data of a 32 -> 1 channel conv has ONE output channel, so a plain implicit GEMM would fill 1/32 of an MFMA
// tile.  Here the (dy, dx) taps take the M dimension instead:
//     T[(dy,dx)][z, y', x'] = sum_{c, dz} Wf[c][dz][dy][dx] * G_N[c][z + dz - 2][y'][x']          (M = 25 of 32, K = 160)
//     out[z, y, x]          = sum_{dy,dx} T[(dy,dx)][z][y + dy - 2][x + dx - 2]
// A workgroup owns one peak and a slab of TZ output planes and marches along y': per fine input row it computes T for
// the whole row (N = (z-plane, x') flattened, 5 MFMA blocks of 32 columns per wave), writes the 25 partial rows to
// wave-private LDS shifted by dx, and folds them into a rolling buffer of 5 output rows; a row is finished (PreHook
// multiply, clamp, per-peak sum: peak_response_mapping_3d.py:170-171) as soon as its last contributing input row has
// passed.  No halo is recomputed in y or x, the z halo only costs staging.
// Staging reads the COARSE upstream gradient (8x fewer voxels than the un-pooled window the VALU kernel consumed),
// the pool's argmax and a per-tile denominator map (prm_den_pool_kernel: |N| + eps at the argmax child where the pooled
// activation is positive, else 0), and writes each cell's 2x2x2 children (one value, seven zeros) into the LDS tile:
// the un-pooled window never exists in memory.
#include "m3d_common.h"

// timing-only ablation builds (tools/bench_stem.py): 1 = no fold, 2 = no commit, 4 = no fetch, 8 = no MFMA, 16 = no per-step barrier
#ifndef STEM_EXP
#define STEM_EXP 0
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float kEpsM = 1e-10f;   // peak_backprop_3d.py:29

// den[c][b] = |N[c][child(b)]| + eps  if the pooled activation xnext[c][b] > 0 and N >= eps, else 0   (peak-independent)
__global__ __launch_bounds__(256) void prm_den_pool_kernel(const uint8_t* __restrict__ argmax, const float* __restrict__ xnext,
                                                           const float* __restrict__ norm, int C, int UD, int UH, int UW, int D, int H,
                                                           int W, float* __restrict__ den) {
  const long long total = (long long)C * UD * UH * UW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int ax = (int)(e % UW);
    long long t = e / UW;
    const int ay = (int)(t % UH); t /= UH;
    const int az = (int)(t % UD);
    const int c = (int)(t / UD);
    float d = 0.f;
    if (xnext[e] > 0.f) {
      const int child = argmax[e];
      const int qz = 2 * az + (child >> 2), qy = 2 * ay + ((child >> 1) & 1), qx = 2 * ax + (child & 1);
      if ((qz < D) & (qy < H) & (qx < W)) {
        const float n = norm[(((size_t)c * D + qz) * H + qy) * W + qx];
        if (!(n < kEpsM)) d = fabsf(n) + kEpsM;
      }
    }
    den[e] = d;
  }
}

// A operand, one float per lane and K step: wA[s][lane] = Wf[c][dz][dy][dx], s = chunk*10 + dz*2 + cp, c = 4*chunk + 2*cp + (lane>>5),
// (dy,dx) = divmod(lane & 31, 5) (rows 25..31 zero), Wf = tap-flipped relu(W) as in prm_stem_prep_kernel.
__global__ void prm_stem_mfma_pack_kernel(const float* __restrict__ w /*[32,125]*/, float* __restrict__ wA) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 80 * 64) return;
  const int lane = e & 63, s = e >> 6;
  const int chunk = s / 10, dz = (s % 10) >> 1, cp = s & 1;
  const int c = 4 * chunk + 2 * cp + (lane >> 5), i = lane & 31;
  float v = 0.f;
  if (i < 25) {
    const int t = (dz * 5 + i / 5) * 5 + i % 5;
    v = w[c * 125 + (124 - t)];
    v = v > 0.f ? v : 0.f;
  }
  wA[e] = v;
}

struct StemMArgs {
  const float* gup;        // [P, 32, U, U, U] gradient w.r.t. the pooled stem output, window of U^3 coarse cells
  const int* origin_up;    // [P, 3] window origin in pooled coordinates
  const float* den;        // [32, UD, UH, UW] prm_den_pool_kernel
  const uint8_t* argmax;   // [32, UD, UH, UW]
  const float* scale;      // [32] eval-BatchNorm scale or null
  const float* wA;         // [80][64]
  const float* data;       // [D, H, W]
  const float* data_off;   // scalar: min(data)
  float* out;              // [P, Wn, Wn, Wn], Wn = 2U + 4
  float* sums;             // [P]
  int* origins_out;        // [P, 3]
  int P, U, ZW;
  int UD, UH, UW, D, H, W;
  // gup element (p, c, z, y, x) at p*gps + c*gcs + z*gzs + y*gys + x (batch-major or strip layout, see prm.hip PrepParams)
  long long gps;
  int gcs, gzs, gys;
  const float* xnext;      // with up_off: [32, UD, UH, UW] pooled stem activation - gup is a bare backward-data result and the PreHook
  const float* up_off;     // multiply by (xnext - *up_off) of the layer above happens in the staging
  int gzabs;               // strip gup stores the pooled LAYER's planes (plane = uz0 + window plane; prm.hip PrepParams "slab" strips)
};

constexpr int kNCell = 2;    // coarse cells staged per thread and chunk
constexpr int kNT = 512;     // 8 waves: wave (w, r) owns the planes of w and fine row r of the row pair

// Two waves per SIMD.  Everything that is not an MFMA (operand staging, the fold of finished rows) is latency-bound, so the
// two row-halves of the workgroup run the same work in a different ORDER inside a step - r = 0: MFMAs, fold, commit;
// r = 1: fold, commit, MFMAs - and one half's MFMAs cover the other half's staging.
// U (window size in pooled cells) and ZW (output planes per wave) are compile-time: every LDS offset of the MFMA run is then an
// immediate of its ds_read.  The reference's two nets give U = 40 (stride 8) and U = 18 (stride 4).
template <int U, int ZW, int NB>
__global__ __launch_bounds__(kNT, 1) void prm_stem_dgrad_mfma_kernel(StemMArgs q) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, wv = tid >> 6, w = wv & 3, r = wv >> 2, l = tid & 63, h = l >> 5, nl = l & 31;
  const int p = blockIdx.y, slab = blockIdx.x;
  constexpr int NX = 2 * U, Wn = NX + 4, W4 = Wn >> 2, TZ = 4 * ZW, TZH = TZ + 4, NBZ = TZH >> 1;
  constexpr int CS = TZH * 2 * NX;         // floats per channel of a tile buffer: [TZH planes][2 fine rows][NX]
  constexpr int BUF = 4 * CS;
  constexpr int ZWn = ZW * Wn;
  static_assert(4 * NBZ * U <= kNT * kNCell && 32 * NB >= ZW * NX && Wn % 4 == 0, "tile shape");
  float* const tq = sm + 2 * BUF + w * (25 * ZWn);                    // per plane group [25][ZW][Wn], shifted by dx; the two row
  float* const roll = sm + 2 * BUF + 4 * (25 * ZWn) + w * (5 * ZWn);  // waves of a group use it in turn.  roll: [5][ZW][Wn]
  float* const dump = sm + 2 * BUF + 4 * (30 * ZWn) + 4 * wv;         // wave-private slot for T rows 25..31 / columns past the row
  const int z0 = slab * TZ;
  const int uz0 = q.origin_up[3 * p], uy0 = q.origin_up[3 * p + 1], ux0 = q.origin_up[3 * p + 2];
  const int oz = 2 * uz0 - 2, oy = 2 * uy0 - 2, ox = 2 * ux0 - 2;     // origin of the output window in data coordinates
  if (slab == 0 && tid == 0) { q.origins_out[3 * p] = oz; q.origins_out[3 * p + 1] = oy; q.origins_out[3 * p + 2] = ox; }

  // A slab whose output planes all lie outside the volume is zero (the PreHook factor is zero there): nothing to compute.  The
  // nuclei tile is 64 planes deep against 84-plane windows, so about a quarter of the slabs go this way.
  if (oz + z0 >= q.D || oz + z0 + TZ <= 0) {
    const int zend = z0 + TZ < Wn ? z0 + TZ : Wn;
    const size_t base = ((size_t)p * Wn + z0) * Wn * Wn, cnt = (size_t)(zend - z0) * Wn * Wn;
    for (size_t e = (size_t)tid * 4; e < cnt; e += (size_t)kNT * 4)
      *reinterpret_cast<float4*>(q.out + base + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  if (r == 0) {
    for (int e = l; e < 25 * ZWn; e += 64) tq[e] = 0.f;               // the never-written borders of the shifted rows stay zero
    for (int e = l; e < 5 * ZWn; e += 64) roll[e] = 0.f;
  }

  // ---- per-lane constants of the MFMA side
  int bbase[NB], tbase[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    int n = 32 * j + nl;
    const bool ok = n < ZW * NX;
    n = ok ? n : 0;
    const int zl = n / NX, xq = n - zl * NX;
    bbase[j] = h * CS + (ZW * w + zl) * 2 * NX + r * NX + xq;
    tbase[j] = ok ? zl * Wn + xq + 4 : -(1 << 28);
  }
  int toff[13];                                                       // accumulator register e holds T row i = 8*(e/4) + 4*h + e%4
#pragma unroll
  for (int e = 0; e < 13; ++e) {
    const int i = 8 * (e >> 2) + 4 * h + (e & 3);
    toff[e] = i < 25 ? i * ZWn - (i % 5) : -(1 << 28);
  }

  // ---- staging of one chunk (4 channels) of one coarse row.  A thread's cells keep their (channel-in-chunk, coarse plane,
  // coarse x) for the whole kernel; only the coarse row and the chunk move.
  constexpr int ncell = 4 * NBZ * U;
  const int M3 = q.UD * q.UH * q.UW;
  const float* const gp = q.gup + (size_t)p * q.gps;
  int cg[kNCell], cm[kNCell], cl[kNCell], ccv[kNCell];                // gup / map / LDS offsets of the cell (-1: always zero)
#pragma unroll
  for (int i = 0; i < kNCell; ++i) {
    const int e = tid + kNT * i;
    const int bx = e % U, t = e / U;
    const int bzl = t % NBZ, cc = t / NBZ;
    const int bz = (z0 >> 1) - 2 + bzl;
    const int az = uz0 + bz, ax = ux0 + bx;
    const bool ok = (e < ncell) & (bz >= 0) & (bz < U) & (az >= 0) & (az < q.UD) & (ax >= 0) & (ax < q.UW);
    cg[i] = ok ? cc * q.gcs + (q.gzabs ? az : bz) * q.gzs + bx : -1;
    cm[i] = ok ? (cc * q.UD + az) * q.UH * q.UW + ax : 0;
    cl[i] = e < ncell ? cc * CS + (2 * bzl) * 2 * NX + 2 * bx : -1;
    ccv[i] = e < ncell ? cc : 0;
  }
  // fetch issues loads only: any arithmetic on a loaded value would make the compiler wait for it before the MFMA run
  float sg[kNCell], sd[kNCell], ss[kNCell], sx[kNCell];
  int sa[kNCell];
  float a_next[10], a_cur[10];
  bool fetched_rowok = false;
  const float* const scp = q.scale ? q.scale : q.wA;                   // any readable address when there is no scale
  const float* const xnp = q.up_off ? q.xnext : q.den;
  const float up_off = q.up_off ? *q.up_off : 0.f;
  auto fetch = [&](int g) __attribute__((always_inline)) {
    const int by = g >> 3, ch = g & 7;
    const int ay = uy0 + by;
    const bool rowok = (ay >= 0) & (ay < q.UH);
    fetched_rowok = rowok;
    const int go = ch * 4 * q.gcs + by * q.gys, mo = ch * 4 * M3 + (rowok ? ay : 0) * q.UW;
#pragma unroll
    for (int i = 0; i < kNCell; ++i) {
      const bool ok = rowok & (cg[i] >= 0);
      const int gi = ok ? cg[i] + go : 0, mi = ok ? cm[i] + mo : 0;
      sg[i] = gp[gi];
      sd[i] = q.den[mi];
      sa[i] = q.argmax[mi];
      ss[i] = scp[4 * ch + ccv[i]];
      sx[i] = xnp[mi];
    }
  };
  auto fetch_a = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 10; ++s) a_next[s] = q.wA[((g & 7) * 10 + s) * 64 + l];
  };
  auto commit = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kNCell; ++i) {
      if (cl[i] >= 0) {
        const bool ok = fetched_rowok & (cg[i] >= 0);
        float g = q.up_off ? (sx[i] - up_off) * sg[i] : sg[i];                   // PreHook of conv2a, peak_backprop_3d.py:16-18
        g = q.scale ? g * ss[i] : g;                                             // eval-BatchNorm backward
        const float v = (ok & (sd[i] > 0.f)) ? g / sd[i] : 0.f;                  // PostHook division, peak_backprop_3d.py:30-33
        const int child = sa[i];
        float* cell = dst + cl[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                            // (zz, row) of the 2x2x2 children, x pair per store
          const int sel = child - 2 * k;                                        // 0 / 1: the routed child is the left / right x
          float2 pr;
          pr.x = sel == 0 ? v : 0.f;
          pr.y = sel == 1 ? v : 0.f;
          *reinterpret_cast<float2*>(cell + (k >> 1) * 2 * NX + (k & 1) * NX) = pr;
        }
      }
    }
  };

  f32x16 acc[NB];
  float local = 0.f;
  const float off = *q.data_off;

  // ---- fold tasks of a lane: task t = l + 64*it handles (dy = 4 - t / (ZW*W4), plane, x quad); the dy = 4 tasks (the row that
  // is finished) are t < ZW*W4 <= 64, i.e. all in round 0, and their `data` values are fetched a whole step earlier
  constexpr int kRounds = 4;
  constexpr int ntask = 5 * ZW * W4;
  int ft_src[kRounds], ft_dst[kRounds], ft_dy[kRounds];
#pragma unroll
  for (int it = 0; it < kRounds; ++it) {
    const int t = l + 64 * it;
    const int x4 = t % W4, u = t / W4;
    const int zz = u % ZW, dy = 4 - u / ZW;
    ft_src[it] = (dy * 5 * ZW + zz) * Wn + 4 * x4;
    ft_dst[it] = zz * Wn + 4 * x4;
    ft_dy[it] = t < ntask ? dy : -1;
  }
  // finishing lanes (round 0, dy == 4): constant part of the data / output addresses
  const bool fin_lane = ft_dy[0] == 4;
  int fin_doff = 0, fin_xmask = 0;
  size_t fin_out = 0;
  {
    const int x4 = l % W4, zz = (l / W4) % ZW;
    const int z = z0 + ZW * w + zz, qz = oz + z, qx = ox + 4 * x4;
    const bool zok = fin_lane & (z < Wn);
    const bool dok = zok & (qz >= 0) & (qz < q.D);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (dok & (qx + i >= 0) & (qx + i < q.W)) fin_xmask |= 1 << i;
    if (zok) fin_xmask |= 16;                                           // bit 4: the lane stores
    fin_doff = dok ? qz * q.H * q.W + qx : 0;
    fin_out = ((size_t)p * Wn + (zok ? z : 0)) * Wn * Wn + 4 * x4;
  }
  float dpre[4];
  bool dpre_ok = false;
  auto prefetch_data = [&](int y) __attribute__((always_inline)) {
    const int qy = oy + y;
    const bool rowok = (qy >= 0) & (qy < q.H);
    dpre_ok = rowok;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = rowok & ((fin_xmask >> i) & 1);
      dpre[i] = q.data[ok ? fin_doff + qy * q.W + i : 0];
    }
  };
  // PreHook multiply by (data - off), clamp(min = 0), store, accumulate the peak's sum (peak_response_mapping_3d.py:170-171)
  auto finish_row = [&](int y, float4 s) __attribute__((always_inline)) {
    float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o = 0.f;
      if (dpre_ok & ((fin_xmask >> i) & 1)) {
        o = (dpre[i] - off) * v[i];
        o = o > 0.f ? o : 0.f;
      }
      v[i] = o;
      local += o;
    }
    if (fin_xmask & 16) *reinterpret_cast<float4*>(q.out + fin_out + (size_t)y * Wn) = make_float4(v[0], v[1], v[2], v[3]);
  };

  int ym = r;                                                          // (fine row this wave folds next) mod 5
  // fold the T rows of fine input row yi (the accumulators) into the rolling output rows yi .. yi+4
  auto fold = [&](int yi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 13; ++e) {
        const int o = tbase[j] + toff[e];
        float* dst = o >= 0 ? tq + o : dump;
        *dst = acc[j][e];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < kRounds; ++it) {
      const int dy = ft_dy[it];
      if (dy >= 0) {
        const float* tp = tq + ft_src[it];
        float4 s = *reinterpret_cast<const float4*>(tp);
#pragma unroll
        for (int dx = 1; dx < 5; ++dx) {
          const float4 v = *reinterpret_cast<const float4*>(tp + dx * ZWn);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        int slot = ym + 4 - dy;                                        // output row y = yi + 4 - dy lives in slot y mod 5
        slot = slot >= 5 ? slot - 5 : slot;
        float4* rp = reinterpret_cast<float4*>(roll + slot * ZWn + ft_dst[it]);
        if (!(dy == 0 || yi == 0)) {                                   // the first contribution to a row overwrites its slot
          const float4 v = *rp;
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (it == 0 && dy == 4) finish_row(yi, s);                     // input row yi is the last one output row yi receives
        else *rp = s;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ym = ym + 2;
    ym = ym >= 5 ? ym - 5 : ym;
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  };
  // 10 groups (dz, cp) of 5 MFMAs (column blocks); the B values of group k+1 are read while group k runs
  auto mfma_step = [&](const float* tile) __attribute__((always_inline)) {
    float bv[2][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[0][j] = tile[bbase[j]];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int dz = k >> 1, cp = k & 1;
      if (k + 1 < 10) {
        const int o1 = ((k + 1) & 1) * 2 * CS + ((k + 1) >> 1) * 2 * NX;
#pragma unroll
        for (int j = 0; j < NB; ++j) bv[(k + 1) & 1][j] = tile[bbase[j] + o1];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (!(STEM_EXP & 8)) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[dz * 2 + cp], bv[k & 1][j], acc[j], 0, 0, 0);
        else acc[j][0] += bv[k & 1][j];
      }
      __builtin_amdgcn_sched_barrier(0);                               // keep group k+1's reads inside group k's MFMA run
    }
  };

  constexpr int steps = U * 8;
  fetch(0);
  fetch_a(0);
  commit(sm);
#pragma unroll
  for (int s = 0; s < 10; ++s) a_cur[s] = a_next[s];
  if (r == 1 && steps > 1) fetch(1);                                   // the late half stages one step further ahead
  zero_acc();
  __syncthreads();

#pragma unroll 1
  for (int g = 0; g < steps; ++g) {
    const int ch = g & 7, by = g >> 3;
    const float* tile = sm + (g & 1) * BUF;
    float* next = sm + ((g + 1) & 1) * BUF;
    const bool row_in = (uy0 + by >= 0) & (uy0 + by < q.UH);            // a coarse row outside the map stages zeros: skip its MFMAs
    if (g + 1 < steps) fetch_a(g + 1);
    if (r == 0) {
      if (ch == 0) zero_acc();
      if (!(STEM_EXP & 4) && g + 1 < steps) fetch(g + 1);
      if (ch == 7) prefetch_data(2 * by);
      if (row_in) mfma_step(tile);
      if (!(STEM_EXP & 1) && ch == 7) fold(2 * by);
      if (!(STEM_EXP & 2) && g + 1 < steps) commit(next);
    } else {
      if (ch == 0) {
        if (!(STEM_EXP & 1) && g > 0) fold(2 * by - 1);                // row 1 of the previous pair, after the early half's row 0
        zero_acc();
      }
      if (!(STEM_EXP & 2) && g + 1 < steps) commit(next);
      if (!(STEM_EXP & 4) && g + 2 < steps) fetch(g + 2);
      if (ch == 7) prefetch_data(2 * by + 1);
      if (row_in) mfma_step(tile);
    }
#pragma unroll
    for (int s = 0; s < 10; ++s) a_cur[s] = a_next[s];
    if (!(STEM_EXP & 16)) __syncthreads();
  }
  if (r == 1) {
    if (!(STEM_EXP & 1)) fold(NX - 1);
    // rows NX .. NX+3 received their last contribution from input row NX-1
    ym = ym - 1;                                                       // NX mod 5
    ym = ym < 0 ? ym + 5 : ym;
    constexpr int nt = 4 * ZW * W4;
    for (int t = l; t < nt; t += 64) {
      const int x4 = t % W4, u = t / W4;
      const int zz = u % ZW, k = u / ZW;
      int slot = ym + k;
      slot = slot >= 5 ? slot - 5 : slot;
      const float4 s = *reinterpret_cast<const float4*>(roll + (slot * ZW + zz) * Wn + 4 * x4);
      const int y = NX + k, z = z0 + ZW * w + zz;
      if (z < Wn) {
        const int qz = oz + z, qy = oy + y, qx = ox + 4 * x4;
        const bool rowok = (qz >= 0) & (qz < q.D) & (qy >= 0) & (qy < q.H);
        const float* drow = q.data + ((size_t)(rowok ? qz : 0) * q.H + (rowok ? qy : 0)) * q.W;
        float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int x = qx + i;
          float o = 0.f;
          if (rowok & (x >= 0) & (x < q.W)) {
            o = (drow[x] - off) * v[i];
            o = o > 0.f ? o : 0.f;
          }
          v[i] = o;
          local += o;
        }
        *reinterpret_cast<float4*>(q.out + (((size_t)p * Wn + z) * Wn + y) * Wn + 4 * x4) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  (void)local;                       // the per-peak sums come from m3d::window_sums (fixed order), not from atomics
}

}  // namespace

M3D_API int m3d_prm_den_pool(const uint8_t* d_argmax, const float* d_xnext, const float* d_norm, int channels, int up_depth,
                             int up_height, int up_width, int depth, int height, int width, float* d_den, void* stream) {
  if (!d_argmax || !d_xnext || !d_norm || !d_den || channels <= 0 || up_depth <= 0 || up_height <= 0 || up_width <= 0) return M3D_EINVAL;
  const long long total = (long long)channels * up_depth * up_height * up_width;
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(prm_den_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, m3d::as_stream(stream), d_argmax, d_xnext, d_norm,
                     channels, up_depth, up_height, up_width, depth, height, width, d_den);
  return m3d::check_launch("prm_den_pool");
}

M3D_API int m3d_prm_stem_mfma_prepare_weights(const float* d_weight, int channels, float* d_wa, void* stream) {
  if (!d_weight || !d_wa) return M3D_EINVAL;
  if (channels != 32) return M3D_EUNSUPPORTED;
  hipLaunchKernelGGL(prm_stem_mfma_pack_kernel, dim3(20), dim3(256), 0, m3d::as_stream(stream), d_weight, d_wa);
  return m3d::check_launch("prm_stem_mfma_prepare_weights");
}

namespace {
struct StemPlan { int zw, nslab; size_t lds; };
bool stem_plan(int U, StemPlan* pl) {
  if (U != 40 && U != 18) return false;                               // the two instantiations below
  const int NX = 2 * U, Wn = NX + 4;
  const int zw = U == 40 ? 2 : 3;                                     // planes per wave (see the instantiations)
  const int TZ = 4 * zw, TZH = TZ + 4;
  const size_t floats = (size_t)2 * 4 * TZH * 2 * NX + (size_t)4 * 30 * zw * Wn + 32;
  pl->zw = zw; pl->nslab = (Wn + TZ - 1) / TZ; pl->lds = floats * sizeof(float);
  return pl->lds <= 160 * 1024;
}
}  // namespace

/* 1 when m3d_prm_stem_dgrad_fused has a configuration for windows of up_size^3 pooled cells and `channels` stem channels */
M3D_API int m3d_prm_stem_dgrad_fused_supported(int channels, int up_size) {
  StemPlan pl;
  return channels == 32 && stem_plan(up_size, &pl) ? 1 : 0;
}

M3D_API int m3d_prm_stem_dgrad_fused_ex(const float* d_gup, int gup_strip, const float* d_xnext, const float* d_up_offset,
                                        const int32_t* d_origin_up, int num_peaks, int channels, int up_size, const float* d_den, const uint8_t* d_argmax, const float* d_scale, int up_depth, int up_height,
                                     int up_width, const float* d_wa, const float* d_data, const float* d_data_offset, int depth,
                                     int height, int width, float* d_out, float* d_sums, int32_t* d_origins_out, void* stream) {
  return m3d_prm_stem_dgrad_fused_ex2(d_gup, gup_strip, 0, d_xnext, d_up_offset, d_origin_up, num_peaks, channels, up_size, d_den, d_argmax, d_scale,
                                      up_depth, up_height, up_width, d_wa, d_data, d_data_offset, depth, height, width, d_out, d_sums, d_origins_out,
                                      stream);
}

/* gup_slab != 0 (strip layouts only): d_gup stores the up_depth planes of the pooled layer instead of each window's up_size planes */
M3D_API int m3d_prm_stem_dgrad_fused_ex2(const float* d_gup, int gup_strip, int gup_slab, const float* d_xnext, const float* d_up_offset,
                                         const int32_t* d_origin_up, int num_peaks, int channels, int up_size, const float* d_den,
                                         const uint8_t* d_argmax, const float* d_scale, int up_depth, int up_height, int up_width,
                                         const float* d_wa, const float* d_data, const float* d_data_offset, int depth, int height, int width,
                                         float* d_out, float* d_sums, int32_t* d_origins_out, void* stream) {
  if (num_peaks < 0 || channels <= 0 || up_size <= 0 || (gup_slab && !gup_strip)) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_gup || !d_origin_up || !d_den || !d_argmax || !d_wa || !d_data || !d_data_offset || !d_out || !d_sums || !d_origins_out)
    return M3D_EINVAL;
  StemPlan pl;
  if (channels != 32 || !stem_plan(up_size, &pl) || num_peaks > 65535) return M3D_EUNSUPPORTED;
  if (2 * up_depth > depth || 2 * up_height > height || 2 * up_width > width) return M3D_EINVAL;     // MaxPool3d(2,2) floors
  hipStream_t st = m3d::as_stream(stream);
  StemMArgs q;
  q.gup = d_gup; q.origin_up = d_origin_up; q.den = d_den; q.argmax = d_argmax; q.scale = d_scale; q.wA = d_wa; q.data = d_data;
  q.data_off = d_data_offset; q.out = d_out; q.sums = d_sums; q.origins_out = d_origins_out; q.P = num_peaks; q.U = up_size;
  q.ZW = pl.zw; q.UD = up_depth; q.UH = up_height; q.UW = up_width; q.D = depth; q.H = height; q.W = width;
  if ((d_up_offset != nullptr) != (d_xnext != nullptr)) return M3D_EINVAL;
  q.xnext = d_xnext; q.up_off = d_up_offset; q.gzabs = gup_slab ? 1 : 0;
  {
    const long long n = up_size;
    if (gup_strip) {
      if (gup_strip < 0 || gup_strip > 2) return M3D_EINVAL;
      int pitch, lead; long long L;
      m3d::strip_geom(up_size, gup_strip, num_peaks, &pitch, &lead, &L);
      const long long zn = gup_slab ? up_depth : n;
      if (zn * n * L >= 0x7FFFFFFFll / 32) return M3D_EUNSUPPORTED;    // 32-bit offsets inside the gradient tensor
      q.gps = pitch; q.gcs = (int)(zn * n * L); q.gzs = (int)(n * L); q.gys = (int)L;
      q.gup = d_gup + lead;
    } else {
      q.gps = 32 * n * n * n; q.gcs = (int)(n * n * n); q.gzs = (int)(n * n); q.gys = (int)n;
    }
  }
  if ((long long)32 * up_depth * up_height * up_width >= 0x7FFFFFFFll) return M3D_EUNSUPPORTED;
  auto launch = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
    hipLaunchKernelGGL(kern, dim3(pl.nslab, num_peaks), dim3(kNT), pl.lds, st, q);
  };
  // U = 40: 2 planes x 80 columns = 5 full blocks per wave, 11 slabs of 8 planes.  U = 18: 3 planes x 36 columns in 4 blocks (84 %),
  // 4 slabs of 12 planes - 128 peaks give 512 workgroups = two full rounds of the 256 CUs (4 planes in 5 blocks: 384 workgroups,
  // a ragged second round)
  if (up_size == 40) launch(prm_stem_dgrad_mfma_kernel<40, 2, 5>);
  else launch(prm_stem_dgrad_mfma_kernel<18, 3, 4>);
  if (int rc = m3d::check_launch("prm_stem_dgrad_fused")) return rc;
  const long long wn = 2ll * up_size + 4;
  return m3d::window_sums(d_out, wn * wn * wn, num_peaks, d_sums, st);
}

M3D_API int m3d_prm_stem_dgrad_fused(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size,
                                     const float* d_den, const uint8_t* d_argmax, const float* d_scale, int up_depth, int up_height,
                                     int up_width, const float* d_wa, const float* d_data, const float* d_data_offset, int depth,
                                     int height, int width, float* d_out, float* d_sums, int32_t* d_origins_out, void* stream) {
  return m3d_prm_stem_dgrad_fused_ex(d_gup, 0, nullptr, nullptr, d_origin_up, num_peaks, channels, up_size, d_den, d_argmax, d_scale,
                                     up_depth, up_height, up_width, d_wa, d_data, d_data_offset, depth, height, width, d_out, d_sums,
                                     d_origins_out, stream);
}
