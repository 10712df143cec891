// Peak-response back-propagation support kernels for gfx950.
//
// Reference: lib/prm/peak_response_mapping_3d.py:157-172 runs one full-volume autograd backward per kept
// detection (<= 300 per tile) through the hooks of lib/prm/peak_backprop_3d.py:8-34.  A one-hot seed has a
// bounded receptive-field cone (3^3 -> 5^3 -> 7^3 at stride 8, 16^3/18^3 at stride 4, 38^3/40^3 at stride 2,
// 84^3 at stride 1), outside of which every gradient is exactly zero.  The build therefore back-propagates
// all peaks of a tile as ONE batch of cropped windows: per layer a gather/"prepare" kernel (this file) turns
// the upper window into the conv's input window — max-unpool routing, ReLU mask, BatchNorm scale and the
// PostHook division by the norm conv (peak_backprop_3d.py:30-33) fused, reading the saved full-size forward
// tensors at each peak's own origin — and the MFMA conv (conv3d.hip, dgrad-packed relu(W)) finishes with the
// PreHook multiply (peak_backprop_3d.py:16-18) fused in its epilogue.  Windows live in virtual coordinates
// (they may stick out of the tile; such positions are zero), so all peaks share one window size per layer.
#include "m3d_common.h"

namespace {

constexpr float kEps = 1e-10f;   // peak_backprop_3d.py:29

// ---- seed: sigmoid' and the 1x1x1 RPN_cls_score conv, whose upstream gradient has a single non-zero channel ----
// out[p, c] = (h[c,pos] - off_h) * relu(Wcls[a,c]) * gn,  gn = s(1-s) / (|Ncls[a,pos]| + eps)  (0 if Ncls < eps)
__global__ __launch_bounds__(256) void prm_seed_kernel(const int* __restrict__ peaks /*[P,4] a,s,h,w*/, int P,
                                                       const float* __restrict__ prob, const float* __restrict__ ncls,
                                                       const float* __restrict__ wcls /*[A,C]*/, const float* __restrict__ h,
                                                       const float* __restrict__ h_off, int A, int C, int S, int H, int W,
                                                       float* __restrict__ out /*[P,C]*/, int* __restrict__ origin_out /*[P,3] or null*/) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P * C) return;
  const int p = e / C, c = e % C;
  const int a = peaks[4 * p], s = peaks[4 * p + 1], hh = peaks[4 * p + 2], w = peaks[4 * p + 3];
  if (origin_out && c == 0) { origin_out[3 * p] = s; origin_out[3 * p + 1] = hh; origin_out[3 * p + 2] = w; }   // the 1^3 window's origin
  const size_t pos = ((size_t)s * H + hh) * W + w, SHW = (size_t)S * H * W;
  const float y = prob[a * SHW + pos];
  const float g = (1.f - y) * y;                       // torch sigmoid backward: grad * (1 - y) * y
  const float n = ncls[a * SHW + pos];
  const float gn = (n < kEps) ? 0.f : g / (fabsf(n) + kEps);
  float wv = wcls[(size_t)a * C + c];
  wv = wv > 0.f ? wv : 0.f;
  out[e] = (h[c * SHW + pos] - *h_off) * (wv * gn);
}

// ---- peak selection (peak_response_mapping_3d.py:124-139,161-163) on the device ----
// The reference takes class 1's kept detections (cls_keep_idx[1], :125), turns each one's flat score index into
// (b, a, s, h, w) = unravel_index(idx, (B,S,H,W,A)) reordered (:136-139) and keeps those whose score exceeds peak_threshold
// (:161-162), in detection order.  One workgroup compacts them in that order (ballot + wave prefix) and writes the count, the peaks
// and the detections to the device (for the back-propagation kernels) and, when given, to a host-mapped mirror (pinned memory: the
// caller waits for ONE event and has everything it needs - count, peaks, detections - without a further read).
__global__ __launch_bounds__(256) void prm_select_peaks_kernel(const float* __restrict__ dets /*[rows,7]*/, const long long* __restrict__ keep /*[rows]*/,
                                                               const int* __restrict__ count, int rows, float thr, int A, int S, int H, int W,
                                                               int cap, int* __restrict__ num, int* __restrict__ peaks /*[cap,4]*/,
                                                               float* __restrict__ out /*[cap,7]*/, int* __restrict__ h_num,
                                                               int* __restrict__ h_peaks, float* __restrict__ h_out,
                                                               const float* __restrict__ prob /*[A,S,H,W] or null*/, int* __restrict__ dead /*[cap] or null*/,
                                                               int* __restrict__ h_dead) {
  __shared__ int wave_tot[4];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(max(*count, 0), rows);
  if (tid == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    float d[7];
    bool ok = false;
    if (i < n) {
#pragma unroll
      for (int k = 0; k < 7; ++k) d[k] = dets[(size_t)i * 7 + k];
      ok = d[6] > thr;                                                // :161-162
    }
    const unsigned long long m = __ballot(ok);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int pos = base + before;
    for (int w = 0; w < wave; ++w) pos += wave_tot[w];
    if (ok && pos < cap) {
      const long long idx = keep[i];                                  // flat index into (S,H,W,A), generate_proposals_3d.py:160
      const int a = (int)(idx % A);
      const long long q = idx / A;
      const int w_ = (int)(q % W), h_ = (int)((q / W) % H), s_ = (int)(q / ((long long)W * H));
      peaks[4 * pos] = a; peaks[4 * pos + 1] = s_; peaks[4 * pos + 2] = h_; peaks[4 * pos + 3] = w_;
#pragma unroll
      for (int k = 0; k < 7; ++k) out[(size_t)pos * 7 + k] = d[k];
      if (h_peaks) { h_peaks[4 * pos] = a; h_peaks[4 * pos + 1] = s_; h_peaks[4 * pos + 2] = h_; h_peaks[4 * pos + 3] = w_; }
      if (h_out) {
#pragma unroll
        for (int k = 0; k < 7; ++k) h_out[(size_t)pos * 7 + k] = d[k];
      }
      if (prob) {
        // a peak whose sigmoid derivative (1 - y) y is exactly 0 (prm_seed_kernel's expression: y == 1.0f, a saturated RPN score)
        // seeds an all-zero gradient: every layer of its back-propagation is zero and its map is 0 / 0 (peak_response_mapping_3d.py:170-171)
        const float y = prob[(size_t)a * S * H * W + ((size_t)s_ * H + h_) * W + w_];
        const int dd = ((1.f - y) * y == 0.f) ? 1 : 0;
        if (dead) dead[pos] = dd;
        if (h_dead) h_dead[pos] = dd;
      }
    }
    __syncthreads();
    if (tid == 0) base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  if (tid == 0) {
    const int t = min(base, cap);
    *num = t;
    if (h_num) *h_num = t;
  }
}

// ---- prepare: upper gradient window -> G_N window of this layer ----
struct PrepParams {
  const float* gup;        // [P, C, U, U, U]
  const int* origin_up;    // [P,3] origin of gup in X_{L+1} coordinates
  const uint8_t* argmax;   // pool: [C, UD, UH, UW] argmax of the 2x2x2 windows, else null
  const float* xnext;      // [C, UD, UH, UW]  X_{L+1} (pooled values if pool) — ReLU mask source
  const float* scale;      // [C] BatchNorm gamma/sqrt(var+eps) or null
  const float* norm;       // [C, D, H, W] norm conv output of this layer
  float* out;              // [P, C, Wn, Wn, Wn]
  int* origin_out;         // [P,3]
  int P, C, U, Wn, border, pool;
  int D, H, W;             // this layer's resolution
  int UD, UH, UW;          // X_{L+1} resolution
  // element (p, c, z, y, x) of gup / out sits at p*ps + c*cs + z*zs + y*ys + x.  Batch-major [P,C,n,n,n]: ps = C n^3, cs = n^3,
  // zs = n^2, ys = n.  Strip [C,n,n,P*(n+1)] (the windows side by side along x, one separator column after each - the layout the
  // Winograd kernel convolves as ONE wide volume): ps = n + 1, cs = n^2 L, zs = n L, ys = L with L = P (n + 1).
  long long ips, ics, ops, ocs;
  int izs, iys, ozs, oys;
  int out_sep;             // strip output: zero columns after each window (pitch - Wn), zero-filled here
  int out_lead, out_tail;  // strip output: zero columns before window 0 / after the last window's separator
  const float* xoff;       // non-null: gup is a bare backward-data result; the PreHook multiply by (X_{L+1} - *xoff) happens here
  unsigned* peak_max;      // or null: [P] largest |value| written for each peak (atomic maxima of float bits, zeroed by the host entry): the
                           // per-window operand scale of the f16x2 strip convolution that reads `out` (m3d_conv3d_zw_forward_strip)
  // Depth-clipped strips ("slab" strips, izn / ozn > 0): a strip [C, zn, n, L] stores the planes of the LAYER (plane k of every window =
  // plane k of the layer's map, zn = the layer's depth) instead of the n planes of each window.  Where the tile is thinner than the cone
  // (nuclei: 32 planes against 38 / 40-plane windows) the planes of a window that lie outside the volume - zero gradient in, results
  // never read - are then not stored, convolved or streamed at all; in y and x the layout is unchanged.
  int izn, ozn;
};

// One thread per output voxel (no pooling between this layer and the upper one).  grid = (chunks, C, P): a workgroup walks
// its share of the (p, c) window in strides of 256 voxels - a million 256-voxel workgroups were bound by the dispatch rate, not
// by memory - and channel and peak come from the block index (no 64-bit div/mod chains).
// largest |value| a wave stored for peak p -> q.peak_max[p] (one atomic per wave; every lane of the wave must call)
__device__ __forceinline__ void prep_peak_max(const PrepParams& q, int p, float vm) {
  if (!q.peak_max) return;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) vm = fmaxf(vm, __shfl_xor(vm, o));
  // one 128-byte line per peak (stride 32 floats): thousands of waves add to a peak's maximum, and with the peaks' words side by side in
  // two cache lines every atomic of the launch queued on the same L2 line (the pool kernel's time doubled)
  if ((threadIdx.x & 63) == 0 && vm > 0.f) atomicMax(q.peak_max + 32 * p, __float_as_uint(vm));
}

__global__ __launch_bounds__(256) void prm_prepare_kernel(PrepParams q) {
  const int w3 = (q.ozn ? q.ozn : q.Wn) * q.Wn * q.Wn;
  const int c = blockIdx.y, p = blockIdx.z;
  const int oz = q.origin_up[3 * p] - q.border, oy = q.origin_up[3 * p + 1] - q.border, ox = q.origin_up[3 * p + 2] - q.border;
  if (blockIdx.x == 0 && threadIdx.x == 0 && c == 0) { q.origin_out[3 * p] = oz; q.origin_out[3 * p + 1] = oy; q.origin_out[3 * p + 2] = ox; }
  const int zshift = q.ozn ? oz : 0;                                          // stored plane k holds window plane k - zshift
  const int izshift = q.izn ? q.origin_up[3 * p] : 0;                        // window plane iz of gup is stored at iz + izshift
  const float sc = q.scale ? q.scale[c] : 1.f;
  const float xoff = q.xoff ? *q.xoff : 0.f;
  const float* gup = q.gup + (size_t)p * q.ips + (size_t)c * q.ics;
  const float* xnext = q.xnext + (size_t)c * q.D * q.H * q.W;
  const float* norm = q.norm + (size_t)c * q.D * q.H * q.W;
  float* out = q.out + (size_t)p * q.ops + (size_t)c * q.ocs;
  const float inv_w = 1.0f / (float)q.Wn;
  float vm = 0.f;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < w3; e += gridDim.x * 256) {
    const int r = (int)(((float)e + 0.5f) * inv_w), x = e - r * q.Wn;      // exact for e < 2^22 (Wn <= 100)
    const int zk = (int)(((float)r + 0.5f) * inv_w), y = r - zk * q.Wn;
    const int z = zk - zshift;
    const int iz = z - q.border, iy = y - q.border, ix = x - q.border;       // inner (un-padded) window coords == upper coords
    const int qz = oz + z, qy = oy + y, qx = ox + x;                         // position in this layer's tensor (== X_{L+1})
    float g = 0.f;
    if ((iz >= 0) & (iz < q.U) & (iy >= 0) & (iy < q.U) & (ix >= 0) & (ix < q.U) & (qz >= 0) & (qz < q.D) & (qy >= 0) & (qy < q.H) &
        (qx >= 0) & (qx < q.W)) {
      const size_t pos = ((size_t)qz * q.H + qy) * q.W + qx;
      g = gup[(size_t)(iz + izshift) * q.izs + (size_t)iy * q.iys + ix];     // slab input: plane qz < D = izn
      const float xn = xnext[pos];
      if (q.xoff) g = (xn - xoff) * g;                                       // PreHook of the layer above, peak_backprop_3d.py:16-18
      if (!(xn > 0.f)) g = 0.f;                                              // ReLU backward (output > 0)
      if (q.scale) g = g * sc;                                               // eval-mode BatchNorm backward
      const float n = norm[pos];
      g = (n < kEps) ? 0.f : g / (fabsf(n) + kEps);                          // PostHook, peak_backprop_3d.py:30-33
    }
    float* o = out + (size_t)zk * q.ozs + (size_t)y * q.oys + x;
    o[0] = g;
    vm = fmaxf(vm, fabsf(g));
    if (q.out_sep && x == q.Wn - 1) {
      const int nz = q.out_sep + (p == q.P - 1 ? q.out_tail : 0);
      for (int j = 1; j <= nz; ++j) o[j] = 0.f;
    }
    if (q.out_lead && p == 0 && x == 0)
      for (int j = 1; j <= q.out_lead; ++j) o[-j] = 0.f;
  }
  prep_peak_max(q, p, vm);
}

// The same element rule for a QUAD-ALIGNED output strip (mode 2: pitch % 4 == 0, window p inside the cell [p pitch, (p + 1) pitch) at
// offset `lead`), four output columns per thread: one index computation, three 4-wide reads (gradient window, X_{L+1}, norm map - consecutive
// voxels of one row) and ONE 16-byte store per quad instead of four of everything (round 4's kernel moved 1.05 GB in 1.0-1.6 ms on the nuclei
// tile's 40^3 layer: ~1 TB/s, a fifth of what the chip streams).  Every column of a cell is written (zeros outside the window: separators,
// the lead, the strip's tail), every value is the expression of prm_prepare_kernel evaluated in the same order: bit-identical strips.
// Threads: TX lanes across the quads of a row (>= pitch / 4 + 1), 256 / TX rows per pass; grid = (row chunks, C, P).
template <int TX>
__global__ __launch_bounds__(256) void prm_prepare_quad_kernel(PrepParams q, int pitch, int lead, int tail) {
  const int c = blockIdx.y, p = blockIdx.z;
  const int oz = q.origin_up[3 * p] - q.border, oy = q.origin_up[3 * p + 1] - q.border, ox = q.origin_up[3 * p + 2] - q.border;
  if (blockIdx.x == 0 && threadIdx.x == 0 && c == 0) { q.origin_out[3 * p] = oz; q.origin_out[3 * p + 1] = oy; q.origin_out[3 * p + 2] = ox; }
  const int zshift = q.ozn ? oz : 0;
  const int izshift = q.izn ? q.origin_up[3 * p] : 0;
  const float sc = q.scale ? q.scale[c] : 1.f;
  const float xoff = q.xoff ? *q.xoff : 0.f;
  const float* gup = q.gup + (size_t)p * q.ips + (size_t)c * q.ics;
  const float* xnext = q.xnext + (size_t)c * q.D * q.H * q.W;
  const float* norm = q.norm + (size_t)c * q.D * q.H * q.W;
  // q.out points at window 0's first column (strip base + lead): the cell of peak p starts `lead` columns before its window
  float* out = q.out - lead + (size_t)p * pitch + (size_t)c * q.ocs;
  const int j = threadIdx.x % TX, ry = threadIdx.x / TX;
  const int nq = pitch / 4 + ((p == q.P - 1 && tail) ? tail / 4 : 0);       // quads of this cell (+ the strip's tail behind the last one)
  const int rows = (j < nq) ? (q.ozn ? q.ozn : q.Wn) * q.Wn : 0;            // (lanes beyond the cell's quads idle; they join the wave's maximum below)
  float vm = 0.f;
  const float inv_w = 1.0f / (float)q.Wn;
  const int x0 = 4 * j - lead;                                               // window column of the quad's first element
  const int ix0 = x0 - q.border, qx0 = ox + x0;
  for (int r = blockIdx.x * (256 / TX) + ry; r < rows; r += gridDim.x * (256 / TX)) {
    const int zk = (int)(((float)r + 0.5f) * inv_w), y = r - zk * q.Wn;      // exact for r < 2^22
    const int z = zk - zshift;
    const int iz = z - q.border, iy = y - q.border;
    const int qz = oz + z, qy = oy + y;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if ((iz >= 0) & (iz < q.U) & (iy >= 0) & (iy < q.U) & (qz >= 0) & (qz < q.D) & (qy >= 0) & (qy < q.H)) {
      const float* gr = gup + (size_t)(iz + izshift) * q.izs + (size_t)iy * q.iys;
      const size_t rowpos = ((size_t)qz * q.H + qy) * q.W;
      float gv[4], xn[4], nn[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {                                          // loads first (clamped addresses), arithmetic after
        ok[k] = (ix0 + k >= 0) & (ix0 + k < q.U) & (qx0 + k >= 0) & (qx0 + k < q.W);
        gv[k] = gr[ok[k] ? ix0 + k : 0];
        const size_t pos = rowpos + (ok[k] ? qx0 + k : 0);
        xn[k] = xnext[pos];
        nn[k] = norm[pos];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = gv[k];
        if (q.xoff) v = (xn[k] - xoff) * v;                                  // PreHook of the layer above, peak_backprop_3d.py:16-18
        if (!(xn[k] > 0.f)) v = 0.f;                                         // ReLU backward (output > 0)
        if (q.scale) v = v * sc;                                             // eval-mode BatchNorm backward
        v = (nn[k] < kEps) ? 0.f : v / (fabsf(nn[k]) + kEps);                // PostHook, peak_backprop_3d.py:30-33
        g[k] = ok[k] ? v : 0.f;
      }
    }
    *reinterpret_cast<float4*>(out + (size_t)zk * q.ozs + (size_t)y * q.oys + 4 * j) = make_float4(g[0], g[1], g[2], g[3]);
    vm = fmaxf(vm, fmaxf(fmaxf(fabsf(g[0]), fabsf(g[1])), fmaxf(fabsf(g[2]), fabsf(g[3]))));
  }
  prep_peak_max(q, p, vm);
}

// MaxPool3d(2,2) between this layer and the upper one: one thread per 2x2x2 output block, i.e. per UPPER voxel (plus a
// one-block shell for the zero border).  Exactly one child of an inner block receives the routed gradient, so the
// upper-resolution tensors (gradient, argmax, pooled value) are read once per block, the norm conv once per block,
// and the eight outputs are written as four 8-byte stores.  grid = (ceil((U+2)^3/256), C, P).
__global__ __launch_bounds__(256) void prm_prepare_pool_kernel(PrepParams q) {
  const int UB = q.U + 2;                                                  // blocks per axis incl. the shell
  const int c = blockIdx.y, p = blockIdx.z;
  const int uz0 = q.origin_up[3 * p], uy0 = q.origin_up[3 * p + 1], ux0 = q.origin_up[3 * p + 2];
  const int oz = 2 * uz0 - q.border, oy = 2 * uy0 - q.border, ox = 2 * ux0 - q.border;
  if (blockIdx.x == 0 && threadIdx.x == 0 && c == 0) { q.origin_out[3 * p] = oz; q.origin_out[3 * p + 1] = oy; q.origin_out[3 * p + 2] = ox; }
  const float sc = q.scale ? q.scale[c] : 1.f;
  const float xoff = q.xoff ? *q.xoff : 0.f;
  const float* gup = q.gup + (size_t)p * q.ips + (size_t)c * q.ics;
  const size_t umap = (size_t)c * q.UD * q.UH * q.UW;
  const float* norm = q.norm + (size_t)c * q.D * q.H * q.W;
  float* o = q.out + (size_t)p * q.ops + (size_t)c * q.ocs;
  const float inv_u = 1.0f / (float)UB;
  // slab output: the z blocks are the pooling blocks of the LAYER (az = 0 .. ceil(ozn / 2)), so that every stored plane is written
  const int ub3 = (q.ozn ? (q.ozn + 1) / 2 : UB) * UB * UB;
  float vm = 0.f;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < ub3; e += gridDim.x * 256) {
    // block b covers inner coordinates 2*(b-1) .. 2*(b-1)+1 shifted so that every window voxel belongs to one block:
    // window coordinate w = inner + border; blocks are aligned to the INNER grid (pooling windows).
    const int r = (int)(((float)e + 0.5f) * inv_u), bx = e - r * UB - 1;           // upper-window voxel index, -1 .. U
    const int rz = (int)(((float)r + 0.5f) * inv_u), by = r - rz * UB - 1, bz = q.ozn ? rz - uz0 : rz - 1;
    float vals[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) vals[k] = 0.f;
    const int az = uz0 + bz, ay = uy0 + by, ax = ux0 + bx;                          // upper tensor position
    if ((bz >= 0) & (bz < q.U) & (by >= 0) & (by < q.U) & (bx >= 0) & (bx < q.U) & (az >= 0) & (az < q.UD) & (ay >= 0) & (ay < q.UH) &
        (ax >= 0) & (ax < q.UW)) {
      const size_t upos = umap + ((size_t)az * q.UH + ay) * q.UW + ax;
      float g = gup[(size_t)(q.izn ? az : bz) * q.izs + (size_t)by * q.iys + bx];
      const float xn = q.xnext[upos];
      if (q.xoff) g = (xn - xoff) * g;                                               // PreHook of the layer above
      if (!(xn > 0.f)) g = 0.f;                                                      // ReLU backward on the pooled value
      if (q.scale) g = g * sc;
      const int child = q.argmax[upos];                                              // max-unpool routing
      const int qz = 2 * az + (child >> 2), qy = 2 * ay + ((child >> 1) & 1), qx = 2 * ax + (child & 1);
      if ((qz < q.D) & (qy < q.H) & (qx < q.W)) {
        const float n = norm[((size_t)qz * q.H + qy) * q.W + qx];
        vals[child] = (n < kEps) ? 0.f : g / (fabsf(n) + kEps);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int wy = 2 * by + ((k >> 1) & 1) + q.border, wx = 2 * bx + (k & 1) + q.border;
      const int wz = q.ozn ? 2 * (uz0 + bz) + (k >> 2) : 2 * bz + (k >> 2) + q.border;          // stored plane (slab: the layer's plane)
      if ((wz >= 0) & (wz < (q.ozn ? q.ozn : q.Wn)) & (wy >= 0) & (wy < q.Wn) & (wx >= 0) & (wx < q.Wn)) {
        float* d = o + (size_t)wz * q.ozs + (size_t)wy * q.oys + wx;
        d[0] = vals[k];
        vm = fmaxf(vm, fabsf(vals[k]));
        if (q.out_sep && wx == q.Wn - 1) {
          const int nz = q.out_sep + (p == q.P - 1 ? q.out_tail : 0);
          for (int j = 1; j <= nz; ++j) d[j] = 0.f;
        }
        if (q.out_lead && p == 0 && wx == 0)
          for (int j = 1; j <= q.out_lead; ++j) d[-j] = 0.f;
      }
    }
  }
  prep_peak_max(q, p, vm);
}

// ---- stem dgrad: conv1a is 1 -> 32 channels, 5^3; its backward-data has ONE output channel, so an MFMA tile
// would be 1/32 full.  VALU direct form:  out[p,v] = (data[v] - off) * sum_{c,t} Wf[c][t] * G[p,c][v + t - 2],
// Wf = flipped relu(W) (prepared once by prm_stem_prep_kernel so that the tap weights are wave-uniform scalar
// loads, not LDS traffic).  Each thread owns 8 x  by 2 y outputs: one dz-slab of 6 rows x 12 inputs (18 ds_read_b128)
// feeds 16 * 25 = 400 FMAs.  Tile 32 x 16 x 8 voxels, the next channel's halo tile is fetched into registers
// while the current one is consumed (double-buffered LDS).
__global__ void prm_stem_prep_kernel(const float* __restrict__ w /*[C,125]*/, int C, float* __restrict__ wf, int relu) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= C * 125) return;
  const int c = e / 125, t = e % 125;
  const float v = w[c * 125 + (124 - t)];          // dgrad of a "same" conv: W'[t'] = W[124 - t']
  wf[e] = (relu && !(v > 0.f)) ? 0.f : v;           // relu(W), peak_backprop_3d.py:41 (plain flip for the autograd dgrad)
}

// Thread layout NXT (x) x NYT (y) x 8 (z), 8 x 2 outputs per thread: tile 8*NXT x 2*NYT x 8.  <4,8>: 32 x 16 x 8 (the
// 84^3 windows of the stride-8 net); <5,6>: 40 x 12 x 8 with 240 of 256 threads computing (the 40^3 windows of the stride-4
// net fit x exactly: 78 % useful work instead of 52 %).
template <int NXT, int NYT>
struct SDG {
  static constexpr int TX = 8 * NXT, TY = 2 * NYT, TZ = 8, HX = TX + 4, HY = TY + 4, HZ = TZ + 4;
  static constexpr int TILE = HZ * HY * HX;
  static constexpr int NI = (TILE + 255) / 256;
  static_assert(NXT * NYT * 8 <= 256 && HX % 4 == 0, "thread layout / float4 rows");
};

// PLAIN = true: the bare backward-data of the stem conv for autograd (m3d_conv3d_stem5_dgrad): windows are whole
// [Wz,Wy,Wx] maps, no PreHook multiply / clamp / sum.
template <int NXT, int NYT, bool PLAIN>
__global__ __launch_bounds__(256) void prm_stem_dgrad_kernel(const float* __restrict__ gn /*[P,C,Wz,Wy,Wx]*/,
                                                             const float* __restrict__ wf /*[C,125] flipped (relu) W*/,
                                                             const float* __restrict__ data /*[D,H,W]*/,
                                                             const float* __restrict__ data_off, const int* __restrict__ origins,
                                                             int Wz, int Wy, int Wx, int D, int H, int W, int C,
                                                             float* __restrict__ out /*[P,Wz,Wy,Wx]*/, float* __restrict__ sums /*[P]*/) {
  using G = SDG<NXT, NYT>;
  constexpr int SD_TX = G::TX, SD_TY = G::TY, SD_TZ = G::TZ, SD_HX = G::HX, SD_HY = G::HY, SD_TILE = G::TILE, SD_NI = G::NI;
  extern __shared__ float sd_lds[];                       // 2 x SD_TILE
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int tiles = (Wx + SD_TX - 1) / SD_TX, tilesy = (Wy + SD_TY - 1) / SD_TY;
  int bid = blockIdx.x;
  const int tx = bid % tiles; bid /= tiles;
  const int ty = bid % tilesy; bid /= tilesy;
  const int tz = bid;
  const int p = blockIdx.y;
  const int x0 = tx * SD_TX, y0 = ty * SD_TY, z0 = tz * SD_TZ;
  const bool active = tid < NXT * NYT * 8;                // the other threads only help staging
  const int ct = active ? tid : 0;
  const int lx = (ct % NXT) * 8, ly = ((ct / NXT) % NYT) * 2, lz = ct / (NXT * NYT);
  const size_t w3 = (size_t)Wz * Wy * Wx;

  // staging: element e of the halo tile <-> offset inside one channel window (-1 = zero); recomputed per channel
  // (cheap VALU) rather than kept in 34 registers
  auto elem_off = [&](int e) __attribute__((always_inline)) -> int {
    const int hz = e / (SD_HY * SD_HX), hy = (e / SD_HX) % SD_HY, hx = e % SD_HX;
    const int z = z0 + hz - 2, y = y0 + hy - 2, x = x0 + hx - 2;
    const bool ok = (e < SD_TILE) & (z >= 0) & (z < Wz) & (y >= 0) & (y < Wy) & (x >= 0) & (x < Wx);
    return ok ? (int)(((size_t)z * Wy + y) * Wx + x) : -1;
  };
  float rin[SD_NI];
  auto prefetch = [&](int c) __attribute__((always_inline)) {
    const float* g = gn + ((size_t)p * C + c) * w3;
#pragma unroll
    for (int i = 0; i < SD_NI; ++i) {
      const int o = elem_off(tid + i * 256);
      rin[i] = g[o < 0 ? 0 : o];
    }
  };
  auto commit = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < SD_NI; ++i) {
      const int e = tid + i * 256;
      if (e < SD_TILE) dst[e] = elem_off(e) < 0 ? 0.f : rin[i];
    }
  };

  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;

  prefetch(0);
  commit(sd_lds);
  __syncthreads();
  for (int c = 0; c < C; ++c) {
    const float* tile = sd_lds + (c & 1) * SD_TILE;
    if (c + 1 < C) prefetch(c + 1);
    const float* wc = wf + c * 125;
#pragma unroll 1
    for (int dz = 0; dz < 5; ++dz) {
      float r[6][12];
#pragma unroll
      for (int yy = 0; yy < 6; ++yy) {
        const float4* row = reinterpret_cast<const float4*>(tile + ((lz + dz) * SD_HY + (ly + yy)) * SD_HX + lx);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const float4 v = row[q];
          r[yy][4 * q] = v.x; r[yy][4 * q + 1] = v.y; r[yy][4 * q + 2] = v.z; r[yy][4 * q + 3] = v.w;
        }
      }
#pragma unroll
      for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const float wv = wc[(dz * 5 + dy) * 5 + dx];      // wave-uniform -> scalar load
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(wv, r[j + dy][i + dx], acc[j][i]);
        }
    }
    if (c + 1 < C) commit(sd_lds + ((c + 1) & 1) * SD_TILE);
    __syncthreads();
  }
  // PreHook multiply, clamp(min=0) (peak_response_mapping_3d.py:170), per-peak sum for the normalisation (:171)
  const int z = z0 + lz;
  if constexpr (PLAIN) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int y = y0 + ly + j;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int x = x0 + lx + i;
        if (active && z < Wz && y < Wy && x < Wx) out[(size_t)p * w3 + ((size_t)z * Wy + y) * Wx + x] = acc[j][i];
      }
    }
    return;
  }
  const int oz = origins[3 * p], oy = origins[3 * p + 1], ox = origins[3 * p + 2];
  float local = 0.f;
  const float off = *data_off;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int y = y0 + ly + j;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int x = x0 + lx + i;
      if (active && z < Wz && y < Wy && x < Wx) {
        const int qz = oz + z, qy = oy + y, qx = ox + x;
        float v = 0.f;
        if ((qz >= 0) & (qz < D) & (qy >= 0) & (qy < H) & (qx >= 0) & (qx < W)) {
          v = (data[((size_t)qz * H + qy) * W + qx] - off) * acc[j][i];
          v = v > 0.f ? v : 0.f;
        }
        out[(size_t)p * w3 + ((size_t)z * Wy + y) * Wx + x] = v;
        local += v;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = local;
  __syncthreads();
  (void)red; (void)sums;            // the per-peak sums come from m3d::window_sums (fixed order), not from atomics
}

__global__ __launch_bounds__(1024) void window_sums_kernel(const float* __restrict__ win, long long w3, float* __restrict__ sums) {
  __shared__ float red[1024];
  const int p = blockIdx.x, tid = threadIdx.x;
  const float* w = win + (size_t)p * w3;
  float acc = 0.f;
  if ((w3 & 3) == 0 && ((uintptr_t)w & 15) == 0) {
    const float4* w4 = reinterpret_cast<const float4*>(w);
    for (long long i = tid; i < (w3 >> 2); i += 1024) {
      const float4 v = w4[i];
      acc += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long long i = tid; i < w3; i += 1024) acc += w[i];
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) sums[p] = red[0];
}

// ---- dense [P, D, H, W] maps: every voxel outside a peak's window is 0 / sum - 0 for a regular peak, NaN for the degenerate one whose
// back-propagated map is all zero (a saturated RPN sigmoid: y (1 - y) == 0), where the reference's `prm / prm.sum()` is 0 / 0 at EVERY voxel
// (peak_response_mapping_3d.py:170-171; tests/golden/prm_saturated.npz)
__global__ __launch_bounds__(256) void prm_fill_kernel(const float* __restrict__ sums, long long n4, float4* __restrict__ dense) {
  const float v = 0.f / sums[blockIdx.y];
  float4* d = dense + (size_t)blockIdx.y * n4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) d[e] = make_float4(v, v, v, v);
}
// maps whose voxel count is not a multiple of 4 (or an unaligned buffer): per-peak bases are not 16-byte aligned, so dword stores - but over
// the same (chunks, peak) grid as the float4 form, not one workgroup per map
__global__ __launch_bounds__(256) void prm_fill_scalar_kernel(const float* __restrict__ sums, long long n, float* __restrict__ dense) {
  const float v = 0.f / sums[blockIdx.y];
  float* d = dense + (size_t)blockIdx.y * n;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) d[e] = v;
}

// ---- normalise + scatter windows into the dense maps ----
__global__ __launch_bounds__(256) void prm_scatter_kernel(const float* __restrict__ win, const float* __restrict__ sums,
                                                          const int* __restrict__ origins, int P, int Wn, int D, int H, int W,
                                                          float* __restrict__ dense) {
  const long long w3 = (long long)Wn * Wn * Wn;
  const long long total = (long long)P * w3;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(e % Wn);
    long long t = e / Wn;
    const int y = (int)(t % Wn); t /= Wn;
    const int z = (int)(t % Wn);
    const int p = (int)(t / Wn);
    const int qz = origins[3 * p] + z, qy = origins[3 * p + 1] + y, qx = origins[3 * p + 2] + x;
    if ((qz >= 0) & (qz < D) & (qy >= 0) & (qy < H) & (qx >= 0) & (qx < W))
      dense[(((size_t)p * D + qz) * H + qy) * W + qx] = win[e] / sums[p];   // prm / prm.sum()  (:171)
  }
}

}  // namespace

namespace m3d {
int window_sums(const float* d_win, long long w3, int num_peaks, float* d_sums, hipStream_t st) {
  if (num_peaks <= 0) return M3D_OK;
  hipLaunchKernelGGL(window_sums_kernel, dim3(num_peaks), dim3(1024), 0, st, d_win, w3, d_sums);
  return check_launch("prm_window_sums");
}
}  // namespace m3d

M3D_API int m3d_prm_seed(const int32_t* d_peaks, int num_peaks, const float* d_prob, const float* d_norm_cls,
                         const float* d_w_cls, const float* d_h, const float* d_h_offset, int A, int C, int S, int H, int W,
                         float* d_out, void* stream) {
  return m3d_prm_seed_ex(d_peaks, num_peaks, d_prob, d_norm_cls, d_w_cls, d_h, d_h_offset, A, C, S, H, W, d_out, nullptr, stream);
}

/* + d_origin_out int32 [P,3] (may be null): the (s, h, w) of every peak - the origin of its 1^3 window, written by the same launch */
M3D_API int m3d_prm_seed_ex(const int32_t* d_peaks, int num_peaks, const float* d_prob, const float* d_norm_cls,
                            const float* d_w_cls, const float* d_h, const float* d_h_offset, int A, int C, int S, int H, int W,
                            float* d_out, int32_t* d_origin_out, void* stream) {
  if (num_peaks < 0 || A <= 0 || C <= 0 || S <= 0 || H <= 0 || W <= 0) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_peaks || !d_prob || !d_norm_cls || !d_w_cls || !d_h || !d_h_offset || !d_out) return M3D_EINVAL;
  hipLaunchKernelGGL(prm_seed_kernel, dim3((num_peaks * C + 255) / 256), dim3(256), 0, m3d::as_stream(stream), d_peaks, num_peaks,
                     d_prob, d_norm_cls, d_w_cls, d_h, d_h_offset, A, C, S, H, W, d_out, d_origin_out);
  return m3d::check_launch("prm_seed");
}

// in_strip / out_strip: 0 = batch-major [P,C,n,n,n], 1 = strip [C,n,n,P*(n+1)] (see PrepParams).  d_up_offset non-null: d_gup is the
// bare backward-data of the layer above (no PreHook multiply in its epilogue) and the multiply by (d_xnext - *d_up_offset) is done here.
M3D_API int m3d_prm_select_peaks(const float* d_dets, const int64_t* d_keep_idx, const int32_t* d_count, int rows, float peak_threshold,
                                 int A, int S, int H, int W, int cap, int32_t* d_num, int32_t* d_peaks, float* d_out_dets, int32_t* h_num,
                                 int32_t* h_peaks, float* h_out_dets, void* stream) {
  return m3d_prm_select_peaks_ex(d_dets, d_keep_idx, d_count, rows, peak_threshold, A, S, H, W, cap, d_num, d_peaks, d_out_dets, h_num, h_peaks,
                                 h_out_dets, nullptr, nullptr, nullptr, stream);
}

/* + d_prob [A,S,H,W] (the class response map) -> d_dead / h_dead int32 [cap] (each may be null): 1 where the peak's sigmoid derivative
 * (1 - y) y is exactly 0, i.e. where the whole back-propagation of that peak is zero (callers may skip it: its map is 0 / 0) */
M3D_API int m3d_prm_select_peaks_ex(const float* d_dets, const int64_t* d_keep_idx, const int32_t* d_count, int rows, float peak_threshold,
                                    int A, int S, int H, int W, int cap, int32_t* d_num, int32_t* d_peaks, float* d_out_dets, int32_t* h_num,
                                    int32_t* h_peaks, float* h_out_dets, const float* d_prob, int32_t* d_dead, int32_t* h_dead, void* stream) {
  if (!d_dets || !d_keep_idx || !d_count || !d_num || !d_peaks || !d_out_dets) return M3D_EINVAL;
  if (rows <= 0 || cap <= 0 || A <= 0 || S <= 0 || H <= 0 || W <= 0) return M3D_EINVAL;
  hipLaunchKernelGGL(prm_select_peaks_kernel, dim3(1), dim3(256), 0, m3d::as_stream(stream), d_dets, (const long long*)d_keep_idx, d_count, rows,
                     peak_threshold, A, S, H, W, cap, d_num, d_peaks, d_out_dets, h_num, h_peaks, h_out_dets, d_prob, d_dead, h_dead);
  return m3d::check_launch("prm_select_peaks");
}

M3D_API int m3d_prm_prepare_ex(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size, int pool,
                               int border, const uint8_t* d_argmax, const float* d_xnext, int up_depth, int up_height, int up_width,
                               const float* d_scale, const float* d_norm, int depth, int height, int width, int in_strip,
                               int out_strip, const float* d_up_offset, float* d_out, int32_t* d_origin_out, void* stream) {
  return m3d_prm_prepare_ex2(d_gup, d_origin_up, num_peaks, channels, up_size, pool, border, d_argmax, d_xnext, up_depth, up_height, up_width,
                             d_scale, d_norm, depth, height, width, in_strip, 0, out_strip, 0, d_up_offset, d_out, d_origin_out, stream);
}

/* in_slab / out_slab != 0: that strip stores the LAYER's planes (up_depth planes for d_gup, `depth` for d_out) instead of each window's
 * (PrepParams: depth-clipped strips); only with the strip layouts. */
M3D_API int m3d_prm_prepare_ex2(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size, int pool,
                                int border, const uint8_t* d_argmax, const float* d_xnext, int up_depth, int up_height, int up_width,
                                const float* d_scale, const float* d_norm, int depth, int height, int width, int in_strip, int in_slab,
                                int out_strip, int out_slab, const float* d_up_offset, float* d_out, int32_t* d_origin_out, void* stream) {
  return m3d_prm_prepare_ex3(d_gup, d_origin_up, num_peaks, channels, up_size, pool, border, d_argmax, d_xnext, up_depth, up_height, up_width,
                             d_scale, d_norm, depth, height, width, in_strip, in_slab, out_strip, out_slab, d_up_offset, d_out, d_origin_out,
                             nullptr, stream);
}

/* ... and d_peak_max [num_peaks x 32] (or NULL; peak p's value at element 32 p: one cache line per peak): the largest |value| written for
 * each peak, the per-window operand bound of
 * m3d_conv3d_zw_forward_strip (zeroed here, filled by the launch: no sweep of the strip). */
M3D_API int m3d_prm_prepare_ex3(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size, int pool,
                                int border, const uint8_t* d_argmax, const float* d_xnext, int up_depth, int up_height, int up_width,
                                const float* d_scale, const float* d_norm, int depth, int height, int width, int in_strip, int in_slab,
                                int out_strip, int out_slab, const float* d_up_offset, float* d_out, int32_t* d_origin_out,
                                float* d_peak_max, void* stream) {
  if (num_peaks < 0 || channels <= 0 || up_size <= 0 || border < 0) return M3D_EINVAL;
  if ((in_slab && !in_strip) || (out_slab && !out_strip) || depth <= 0 || up_depth <= 0) return M3D_EINVAL;
  if (!pool && (up_depth != depth || up_height != height || up_width != width)) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_gup || !d_origin_up || !d_xnext || !d_norm || !d_out || !d_origin_out || (pool && !d_argmax)) return M3D_EINVAL;
  PrepParams q;
  q.gup = d_gup; q.origin_up = d_origin_up; q.argmax = pool ? d_argmax : nullptr; q.xnext = d_xnext; q.scale = d_scale;
  q.norm = d_norm; q.out = d_out; q.origin_out = d_origin_out; q.P = num_peaks; q.C = channels; q.U = up_size;
  q.Wn = (pool ? 2 : 1) * up_size + 2 * border; q.border = border; q.pool = pool ? 1 : 0;
  q.D = depth; q.H = height; q.W = width; q.UD = up_depth; q.UH = up_height; q.UW = up_width;
  q.xoff = d_up_offset; q.out_sep = 0; q.out_lead = 0; q.out_tail = 0;
  q.izn = in_slab ? up_depth : 0; q.ozn = out_slab ? depth : 0;
  q.peak_max = reinterpret_cast<unsigned*>(d_peak_max);
  if (d_peak_max && hipMemsetAsync(d_peak_max, 0, (size_t)num_peaks * 32 * sizeof(float), m3d::as_stream(stream)) != hipSuccess) return M3D_ELAUNCH;
  long long ibase = 0, obase = 0;
  auto strides = [&](int n, int zn, int strip, long long* ps, long long* cs, int* zs, int* ys, long long* base, bool is_out) {
    if (strip) {
      int pitch, lead; long long L;
      m3d::strip_geom(n, strip, num_peaks, &pitch, &lead, &L);
      if ((long long)n * L >= 0x7FFFFFFFll) return false;
      *ps = pitch; *cs = (long long)(zn ? zn : n) * n * L; *zs = (int)(n * L); *ys = (int)L; *base = lead;
      if (is_out) { q.out_sep = pitch - n; q.out_lead = lead; q.out_tail = (int)(L - ((long long)num_peaks * pitch + lead)); }
    } else {
      *ps = (long long)channels * n * n * n; *cs = (long long)n * n * n; *zs = n * n; *ys = n; *base = 0;
    }
    return true;
  };
  if (in_strip < 0 || in_strip > 2 || out_strip < 0 || out_strip > 2) return M3D_EINVAL;
  if (!strides(up_size, q.izn, in_strip, &q.ips, &q.ics, &q.izs, &q.iys, &ibase, false) ||
      !strides(q.Wn, q.ozn, out_strip, &q.ops, &q.ocs, &q.ozs, &q.oys, &obase, true))
    return M3D_EUNSUPPORTED;
  q.gup = d_gup + ibase; q.out = d_out + obase;
  if (channels > 65535 || num_peaks > 65535) return M3D_EUNSUPPORTED;
  if (pool) {
    if (border > 2) return M3D_EUNSUPPORTED;      // the one-block shell covers borders of 1 or 2 voxels
    const int ub = up_size + 2;
    if (ub > 100 || q.ozn > 200) return M3D_EUNSUPPORTED;        // float reciprocal index split in the kernel
    const int blocks = ((q.ozn ? (q.ozn + 1) / 2 : ub) * ub * ub + 255) / 256;
    int chunks = (int)((16384 + (long long)channels * num_peaks - 1) / ((long long)channels * num_peaks));
    chunks = chunks < 1 ? 1 : (chunks > blocks ? blocks : chunks);
    hipLaunchKernelGGL(prm_prepare_pool_kernel, dim3(chunks, channels, num_peaks), dim3(256), 0, m3d::as_stream(stream), q);
  } else {
    if (q.Wn > 100 || q.ozn > 100) return M3D_EUNSUPPORTED;        // float reciprocal index split in the kernel
    if (out_strip == 2) {                                           // quad-aligned strip: four columns per thread, 16-byte stores
      int pitch, lead; long long L;
      m3d::strip_geom(q.Wn, 2, num_peaks, &pitch, &lead, &L);
      const int tail = (int)(L - (long long)num_peaks * pitch);
      const int nq = pitch / 4 + tail / 4;
      if (pitch % 4 == 0 && tail % 4 == 0 && nq <= 32 && (((uintptr_t)d_out) & 15) == 0 && (q.ozs % 4 == 0) && (q.oys % 4 == 0) && (q.ocs % 4 == 0)) {
        const int tx = nq <= 8 ? 8 : (nq <= 16 ? 16 : 32);
        const int rows = (q.ozn ? q.ozn : q.Wn) * q.Wn, rpb = 256 / tx;
        int chunks = (int)((16384 + (long long)channels * num_peaks - 1) / ((long long)channels * num_peaks));
        const int maxc = (rows + rpb - 1) / rpb;
        chunks = chunks < 1 ? 1 : (chunks > maxc ? maxc : chunks);
        const dim3 grid(chunks, channels, num_peaks);
        if (tx == 8) hipLaunchKernelGGL(prm_prepare_quad_kernel<8>, grid, dim3(256), 0, m3d::as_stream(stream), q, pitch, lead, tail);
        else if (tx == 16) hipLaunchKernelGGL(prm_prepare_quad_kernel<16>, grid, dim3(256), 0, m3d::as_stream(stream), q, pitch, lead, tail);
        else hipLaunchKernelGGL(prm_prepare_quad_kernel<32>, grid, dim3(256), 0, m3d::as_stream(stream), q, pitch, lead, tail);
        return m3d::check_launch("prm_prepare");
      }
    }
    const int blocks = ((q.ozn ? q.ozn : q.Wn) * q.Wn * q.Wn + 255) / 256;
    int chunks = (int)((16384 + (long long)channels * num_peaks - 1) / ((long long)channels * num_peaks));   // >= 16 k workgroups
    chunks = chunks < 1 ? 1 : (chunks > blocks ? blocks : chunks);
    hipLaunchKernelGGL(prm_prepare_kernel, dim3(chunks, channels, num_peaks), dim3(256), 0, m3d::as_stream(stream), q);
  }
  return m3d::check_launch("prm_prepare");
}

/* The strip layouts' geometry (m3d_common.h strip_geom) for callers that allocate the buffers: mode 1 = pitch n + 1 (F(2x2,3x3), exactly
 * local), mode 2 = quad-aligned windows for the F(2x4,3x3) kernel.  Returns the row length L; window p starts at column lead + p * pitch. */
M3D_API int64_t m3d_prm_strip_geometry(int window, int mode, int num_peaks, int32_t* pitch, int32_t* lead) {
  int pp, ll; long long L;
  m3d::strip_geom(window, mode, num_peaks, &pp, &ll, &L);
  if (pitch) *pitch = pp;
  if (lead) *lead = ll;
  return (int64_t)L;
}

M3D_API int m3d_prm_prepare(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size, int pool,
                            int border, const uint8_t* d_argmax, const float* d_xnext, int up_depth, int up_height, int up_width,
                            const float* d_scale, const float* d_norm, int depth, int height, int width, float* d_out,
                            int32_t* d_origin_out, void* stream) {
  return m3d_prm_prepare_ex(d_gup, d_origin_up, num_peaks, channels, up_size, pool, border, d_argmax, d_xnext, up_depth, up_height,
                            up_width, d_scale, d_norm, depth, height, width, 0, 0, nullptr, d_out, d_origin_out, stream);
}

M3D_API int m3d_prm_stem_prepare_weights(const float* d_weight, int channels, float* d_wf, void* stream) {
  if (!d_weight || !d_wf || channels <= 0) return M3D_EINVAL;
  hipLaunchKernelGGL(prm_stem_prep_kernel, dim3((channels * 125 + 255) / 256), dim3(256), 0, m3d::as_stream(stream), d_weight,
                     channels, d_wf, 1);
  return m3d::check_launch("prm_stem_prepare_weights");
}

// Backward-data of the 5^3 / Cin = 1 stem conv for autograd (the F.conv3d interception when the input requires grad - exactly
// the reference's PRM mode, peak_response_mapping_3d.py:88 + peak_backprop_3d.py:37-44): gx[b,0,v] = sum_{c,t} W[c][124-t] gy[b,c][v+t-2].
M3D_API int m3d_conv3d_stem5_prepare_dgrad_weights(const float* d_weight, int channels, float* d_wf, void* stream) {
  if (!d_weight || !d_wf || channels <= 0) return M3D_EINVAL;
  hipLaunchKernelGGL(prm_stem_prep_kernel, dim3((channels * 125 + 255) / 256), dim3(256), 0, m3d::as_stream(stream), d_weight,
                     channels, d_wf, 0);
  return m3d::check_launch("conv3d_stem5_prepare_dgrad_weights");
}

M3D_API int m3d_conv3d_stem5_dgrad(const float* d_grad_out, const float* d_wf, float* d_grad_in, int batch, int channels, int depth,
                                   int height, int width, void* stream) {
  if (batch < 0 || channels <= 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (batch == 0) return M3D_OK;
  if (!d_grad_out || !d_wf || !d_grad_in || batch > 65535) return M3D_EINVAL;
  if ((size_t)depth * height * width >= 0x7FFFFFFFull) return M3D_EUNSUPPORTED;      // 32-bit offsets inside one channel map
  using K = SDG<4, 8>;
  auto kern = prm_stem_dgrad_kernel<4, 8, true>;
  const int tx = (width + K::TX - 1) / K::TX, ty = (height + K::TY - 1) / K::TY, tz = (depth + 7) / 8;
  const size_t lds = sizeof(float) * 2 * K::TILE;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(tx * ty * tz, batch), dim3(256), lds, m3d::as_stream(stream), d_grad_out, d_wf,
                     (const float*)nullptr, (const float*)nullptr, (const int*)nullptr, depth, height, width, depth, height, width,
                     channels, d_grad_in, (float*)nullptr);
  return m3d::check_launch("conv3d_stem5_dgrad");
}

M3D_API int m3d_prm_stem_dgrad(const float* d_gn, const float* d_wf, const float* d_data, const float* d_data_offset,
                               const int32_t* d_origins, int num_peaks, int channels, int win, int depth, int height, int width,
                               float* d_out, float* d_sums, void* stream) {
  if (num_peaks < 0 || channels <= 0 || win <= 0) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_gn || !d_wf || !d_data || !d_data_offset || !d_origins || !d_out || !d_sums || num_peaks > 65535) return M3D_EINVAL;
  auto launch = [&](auto kern, int TX, int TY, int TILE) {
    const int tx = (win + TX - 1) / TX, ty = (win + TY - 1) / TY, tz = (win + 7) / 8;
    const size_t lds = sizeof(float) * 2 * TILE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(tx * ty * tz, num_peaks), dim3(256), lds, m3d::as_stream(stream), d_gn, d_wf, d_data,
                       d_data_offset, d_origins, win, win, win, depth, height, width, channels, d_out, d_sums);
  };
  // useful fraction of the computed tile volume decides the layout (84^3: 32x16x8 -> 73 %, 40x12x8 -> 70 %; 40^3: 52 % vs 78 %)
  auto eff = [&](int TX, int TY, double threads) {
    const double cx = (double)((win + TX - 1) / TX) * TX, cy = (double)((win + TY - 1) / TY) * TY, cz = (double)((win + 7) / 8) * 8;
    return (double)win * win * win / (cx * cy * cz) * threads;
  };
  if (eff(40, 12, 240.0 / 256.0) > eff(32, 16, 1.0))
    launch(prm_stem_dgrad_kernel<5, 6, false>, 40, 12, SDG<5, 6>::TILE);
  else
    launch(prm_stem_dgrad_kernel<4, 8, false>, 32, 16, SDG<4, 8>::TILE);
  if (int rc = m3d::check_launch("prm_stem_dgrad")) return rc;
  return m3d::window_sums(d_out, (long long)win * win * win, num_peaks, d_sums, m3d::as_stream(stream));
}

M3D_API int m3d_prm_scatter(const float* d_windows, const float* d_sums, const int32_t* d_origins, int num_peaks, int win,
                            int depth, int height, int width, float* d_dense, void* stream) {
  if (num_peaks < 0 || win <= 0) return M3D_EINVAL;
  if (num_peaks == 0) return M3D_OK;
  if (!d_windows || !d_sums || !d_origins || !d_dense || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (num_peaks > 65535) return M3D_EUNSUPPORTED;
  const long long n = (long long)depth * height * width;
  if (n % 4 == 0 && ((uintptr_t)d_dense & 15) == 0) {
    long long fb = (n / 4 + 255) / 256;
    if (fb > 1024) fb = 1024;
    hipLaunchKernelGGL(prm_fill_kernel, dim3((unsigned)fb, num_peaks), dim3(256), 0, m3d::as_stream(stream), d_sums, n / 4,
                       reinterpret_cast<float4*>(d_dense));
  } else {
    long long fb = (n + 1023) / 1024;                     // 4 dwords per thread and sweep
    if (fb > 1024) fb = 1024;
    hipLaunchKernelGGL(prm_fill_scalar_kernel, dim3((unsigned)fb, num_peaks), dim3(256), 0, m3d::as_stream(stream), d_sums, n, d_dense);
  }
  const long long total = (long long)num_peaks * win * win * win;
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(prm_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, m3d::as_stream(stream), d_windows, d_sums,
                     d_origins, num_peaks, win, depth, height, width, d_dense);
  return m3d::check_launch("prm_scatter");
}
