// Per-RoI 2D-Otsu binarisation (intensity x PRM histogram, oblique threshold line of slope -1) for gfx950.
// Reference: tools/otsu.py:199-284 (otsu_py_2d_fast), called per detection by
// tools/binarization_soma.py:94 and tools/binarization_nuclei.py:124 on uint16 crops.
//
// Restructuring (integer work, HBM/L2 bound): with slope k = -1 a histogram cell (row r = PRM bin, column
// c = intensity bin) joins the background at line parameter b = r + c + 2*g_min + 1 (and never if r or c is
// the last bin, otsu.py:233-235).  The reference's incremental 2D sweep (:251-266) is therefore a running
// sum over the anti-diagonal index d = r + c.  One workgroup per RoI accumulates three INTEGER diagonal
// histograms — N[d] (voxels), M1[d] (sum of c), M2[d] (sum of r) — with LDS/global atomics (exact and
// order-independent), prefix-sums them, evaluates the between-class variance for every b in fp64 and takes the
// first strict maximum (:271-274).  The G x G fp64 tables of the reference (G up to thousands) are never
// materialised: traffic is 2 reads + 1 write per voxel plus O(G).
// fp64 moments are formed from the integer sums with the affine bin-centre formula (centre = lo + step*(i+.5));
// the reference sums per-cell fp64 products, so var_b agrees to ~1e-15 relative, the mask is identical unless
// two different b tie to that precision.  Compiled with -ffp-contract=off (NumPy linspace edges: i*step + lo).
#include "m3d_common.h"

namespace {

struct Edges { double lo, hi, step, inv_step; int G; };

__device__ inline double edge_at(const Edges& e, int i) { return i == e.G ? e.hi : (double)i * e.step + e.lo; }

// np.searchsorted(edges, v, 'right') - 1 with v == last edge -> G-1 (numpy histogramdd)
__device__ inline int bin_of(const Edges& e, double v) {
  int i = (int)((v - e.lo) * e.inv_step);      // initial guess only; the two loops below make it exact
  i = i < 0 ? 0 : (i > e.G ? e.G : i);
  while (i < e.G && edge_at(e, i + 1) <= v) ++i;
  while (i > 0 && edge_at(e, i) > v) --i;
  return i == e.G ? e.G - 1 : i;
}

__device__ inline Edges make_edges(int vmin, int vmax, int G) {
  Edges e;
  e.lo = (double)vmin; e.hi = (double)vmax; e.G = G;
  if (vmin == vmax) { e.lo -= 0.5; e.hi += 0.5; }       // histogramdd: flat axis widened
  e.step = (e.hi - e.lo) / (double)G;                    // linspace: delta / div
  e.inv_step = 1.0 / e.step;
  return e;
}

__device__ inline unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Round 2: four launches over a (chunks, RoI) grid instead of one 1024-thread workgroup per RoI walking its crop three times
// (the nuclei tile's crops reach 10^6 voxels: 2.3 ms): min/max -> diagonal histograms (per-workgroup LDS partials merged with
// atomics: integer sums, order-independent) -> one workgroup per RoI for the O(G) prefix + variance sweep -> mask.
// The arithmetic of every step is unchanged.
constexpr int kOtsuThreads = 1024;                       // evaluation kernel
constexpr int kOtsuWaves = kOtsuThreads / 64;
constexpr int kPT = 256;                                 // element-parallel kernels

struct OtsuState { int gmin, gmax, pmin, pmax; unsigned long long t1, t2; int found, best_b; int pad[6]; };   // 64 bytes per RoI

__global__ void otsu_init_kernel(OtsuState* st, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { OtsuState s{}; s.gmin = 65535; s.gmax = 0; s.pmin = 65535; s.pmax = 0; st[i] = s; }
}

// ---- pass 1: min / max of both channels (otsu.py:201, histogram2d range=None) ----
__global__ __launch_bounds__(kPT) void otsu_minmax_kernel(const uint16_t* __restrict__ image, const uint16_t* __restrict__ prm,
                                                          const int64_t* __restrict__ offsets, OtsuState* __restrict__ st) {
  const int roi = blockIdx.y;
  const int64_t beg = offsets[roi], V = offsets[roi + 1] - beg;
  if (V <= 0) return;
  const uint16_t* img = image + beg;
  const uint16_t* pr = prm + beg;
  int gmin = 65535, gmax = 0, pmin = 65535, pmax = 0;
  for (int64_t i = (int64_t)blockIdx.x * kPT + threadIdx.x; i < V; i += (int64_t)gridDim.x * kPT) {
    const int a = img[i], p = pr[i];
    gmin = min(gmin, a); gmax = max(gmax, a); pmin = min(pmin, p); pmax = max(pmax, p);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    gmin = min(gmin, __shfl_down(gmin, off, 64)); gmax = max(gmax, __shfl_down(gmax, off, 64));
    pmin = min(pmin, __shfl_down(pmin, off, 64)); pmax = max(pmax, __shfl_down(pmax, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&st[roi].gmin, gmin); atomicMax(&st[roi].gmax, gmax); atomicMin(&st[roi].pmin, pmin); atomicMax(&st[roi].pmax, pmax);
  }
}

// ---- pass 2: integer diagonal histograms + totals ----
__global__ __launch_bounds__(kPT) void otsu_hist_kernel(const uint16_t* __restrict__ image, const uint16_t* __restrict__ prm,
                                                        const int64_t* __restrict__ offsets, int max_g, int lds_g,
                                                        unsigned long long* __restrict__ ws, OtsuState* __restrict__ st) {
  const int roi = blockIdx.y, tid = threadIdx.x;
  const int64_t beg = offsets[roi], V = offsets[roi + 1] - beg;
  if (V <= 0) return;
  const OtsuState s = st[roi];
  const int gmin = s.gmin, gmax = s.gmax, pmin = s.pmin, pmax = s.pmax;
  const int G = gmax - gmin + 1;                                          // :202
  if (G > max_g) return;
  const Edges e1 = make_edges(gmin, gmax, G), e2 = make_edges(pmin, pmax, G);   // :203
  const int ND = 2 * G;
  extern __shared__ unsigned long long lds_hist[];
  const bool in_lds = G <= lds_g;
  unsigned long long* gN = ws + (size_t)roi * 3 * (2 * (size_t)max_g);
  unsigned long long* gM1 = gN + 2 * (size_t)max_g;
  unsigned long long* gM2 = gM1 + 2 * (size_t)max_g;
  unsigned long long* dN = in_lds ? lds_hist : gN;
  unsigned long long* dM1 = in_lds ? lds_hist + 2 * (size_t)lds_g : gM1;
  unsigned long long* dM2 = in_lds ? lds_hist + 4 * (size_t)lds_g : gM2;
  if (in_lds) {
    for (int d = tid; d < ND; d += kPT) { dN[d] = 0ull; dM1[d] = 0ull; dM2[d] = 0ull; }
    __syncthreads();
  }
  const uint16_t* img = image + beg;
  const uint16_t* pr = prm + beg;
  unsigned long long t1 = 0, t2 = 0;
  for (int64_t i = (int64_t)blockIdx.x * kPT + tid; i < V; i += (int64_t)gridDim.x * kPT) {
    const int c = bin_of(e1, (double)img[i]);
    const int r = bin_of(e2, (double)pr[i]);
    t1 += (unsigned long long)c; t2 += (unsigned long long)r;
    if (c <= G - 2 && r <= G - 2) {                                       // last row / column never enter (:233-235)
      atomicAdd(&dN[r + c], 1ull);
      atomicAdd(&dM1[r + c], (unsigned long long)c);
      atomicAdd(&dM2[r + c], (unsigned long long)r);
    }
  }
  t1 = wave_sum_u64(t1); t2 = wave_sum_u64(t2);
  if ((tid & 63) == 0) { atomicAdd(&st[roi].t1, t1); atomicAdd(&st[roi].t2, t2); }
  if (in_lds) {
    __syncthreads();
    for (int d = tid; d < ND; d += kPT) {
      const unsigned long long n = dN[d];
      if (n) { atomicAdd(&gN[d], n); atomicAdd(&gM1[d], dM1[d]); atomicAdd(&gM2[d], dM2[d]); }
    }
  }
}

// ---- passes 3 + 4: prefix over the diagonals, var_b sweep; one workgroup per RoI (O(G) work) ----
__global__ __launch_bounds__(kOtsuThreads) void otsu_eval_kernel(const int64_t* __restrict__ offsets, int max_g,
                                                                 unsigned long long* __restrict__ ws, OtsuState* __restrict__ st,
                                                                 int32_t* __restrict__ kb, int32_t* __restrict__ status) {
  const int roi = blockIdx.x;
  const int64_t V = offsets[roi + 1] - offsets[roi];
  const int tid = threadIdx.x;
  __shared__ double s_best[kOtsuWaves];
  __shared__ int s_bestb[kOtsuWaves];
  __shared__ unsigned long long s_part[3][kOtsuThreads];
  if (V <= 0) {
    if (tid == 0) { kb[2 * roi] = 0; kb[2 * roi + 1] = 0; status[roi] = 2; st[roi].found = 0; }
    return;
  }
  const OtsuState s = st[roi];
  const int gmin = s.gmin, gmax = s.gmax, pmin = s.pmin, pmax = s.pmax;
  const int G = gmax - gmin + 1;
  if (G > max_g) {
    if (tid == 0) { kb[2 * roi] = 0; kb[2 * roi + 1] = 0; status[roi] = 3; st[roi].found = 0; }
    return;
  }
  const Edges e1 = make_edges(gmin, gmax, G), e2 = make_edges(pmin, pmax, G);
  const int ND = 2 * G;                                                   // diagonals 0 .. 2G-2
  unsigned long long* dN = ws + (size_t)roi * 3 * (2 * (size_t)max_g);
  unsigned long long* dM1 = dN + 2 * (size_t)max_g;
  unsigned long long* dM2 = dM1 + 2 * (size_t)max_g;
  // inclusive prefix over diagonals: each thread owns L consecutive entries (local sums, block scan of the per-thread totals,
  // then the local running sums are written back)
  {
    const int L = (ND + kOtsuThreads - 1) / kOtsuThreads;
    const int d0 = tid * L, d1 = min(ND, d0 + L);
    unsigned long long a = 0, b = 0, c = 0;
    for (int d = d0; d < d1; ++d) { a += dN[d]; b += dM1[d]; c += dM2[d]; }
    s_part[0][tid] = a; s_part[1][tid] = b; s_part[2][tid] = c;
    __syncthreads();
    for (int off = 1; off < kOtsuThreads; off <<= 1) {       // Hillis-Steele inclusive scan of the totals
      unsigned long long va = 0, vb = 0, vc = 0;
      if (tid >= off) { va = s_part[0][tid - off]; vb = s_part[1][tid - off]; vc = s_part[2][tid - off]; }
      __syncthreads();
      s_part[0][tid] += va; s_part[1][tid] += vb; s_part[2][tid] += vc;
      __syncthreads();
    }
    a = s_part[0][tid] - a; b = s_part[1][tid] - b; c = s_part[2][tid] - c;   // exclusive prefix of this thread's range
    for (int d = d0; d < d1; ++d) { a += dN[d]; b += dM1[d]; c += dM2[d]; dN[d] = a; dM1[d] = b; dM2[d] = c; }
  }
  __threadfence_block();
  __syncthreads();

  // var_b for every b in [2*gmin+1, 2*gmax-1)  (:226-227,251)
  const double Vd = (double)V;
  const double ut0 = (e1.lo * Vd + e1.step * ((double)s.t1 + 0.5 * Vd)) / Vd;   // :216
  const double ut1 = (e2.lo * Vd + e2.step * ((double)s.t2 + 0.5 * Vd)) / Vd;   // :217
  const int b_dw = 2 * gmin + 1, b_up = 2 * gmax - 1;
  double best = 0.0;                                                      // var_b_max = 0 (:219)
  int best_b = 0x7FFFFFFF;
  const int nb = b_up > b_dw ? (b_up - b_dw) : 1;                          // b_dw itself is always evaluated (:243-250)
  for (int t = tid; t < nb; t += kOtsuThreads) {
    const int b = b_dw + t;
    const int d = b - 2 * gmin - 1;                                        // cells with r + c <= d are background
    double p0 = 0, u00 = 0, u01 = 0;
    if (d >= 0 && d < ND) {
      const double n = (double)dN[d];
      p0 = n / Vd;
      u00 = (e1.lo * n + e1.step * ((double)dM1[d] + 0.5 * n)) / Vd;
      u01 = (e2.lo * n + e2.step * ((double)dM2[d] + 0.5 * n)) / Vd;
    }
    const double p1 = 1. - p0;                                             // :267
    const double u10 = (ut0 - p0 * u00) / p1, u11 = (ut1 - p0 * u01) / p1; // :268-269 (u0 un-normalised, as is)
    const double d00 = u00 - ut0, d01 = u01 - ut1, d10 = u10 - ut0, d11 = u11 - ut1;
    const double var_b = (p0 * d00 * d00 + p1 * d10 * d10) + (p0 * d01 * d01 + p1 * d11 * d11);   // :270
    if (var_b > best) { best = var_b; best_b = b; }                        // strict >, ascending b per thread (:271)
  }
  // reduce: largest var, ties -> smallest b  (== first strict maximum of the sequential sweep)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_down(best, off, 64);
    const int ob = __shfl_down(best_b, off, 64);
    if (ov > best || (ov == best && ob < best_b)) { best = ov; best_b = ob; }
  }
  if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_bestb[tid >> 6] = best_b; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kOtsuWaves; ++w)
      if (s_best[w] > s_best[0] || (s_best[w] == s_best[0] && s_bestb[w] < s_bestb[0])) { s_best[0] = s_best[w]; s_bestb[0] = s_bestb[w]; }
    best = s_best[0]; best_b = s_bestb[0];
    const bool found = best > 0.0 && best_b != 0x7FFFFFFF;
    kb[2 * roi] = found ? -1 : 0;
    kb[2 * roi + 1] = found ? best_b : 0;
    status[roi] = found ? 0 : 1;   // reference raises UnboundLocalError on k_max (:277)
    st[roi].found = found ? 1 : 0;
    st[roi].best_b = best_b;
  }
}

// ---- pass 5: mask (:276-282) ----
__global__ __launch_bounds__(kPT) void otsu_mask_kernel(const uint16_t* __restrict__ image, const uint16_t* __restrict__ prm,
                                                        const int64_t* __restrict__ offsets, int max_g,
                                                        const OtsuState* __restrict__ st, uint8_t* __restrict__ mask) {
  const int roi = blockIdx.y;
  const int64_t beg = offsets[roi], V = offsets[roi + 1] - beg;
  if (V <= 0) return;
  const OtsuState s = st[roi];
  if (s.gmax - s.gmin + 1 > max_g) return;                                 // status 3: the mask is left untouched, as before
  const uint16_t* img = image + beg;
  const uint16_t* pr = prm + beg;
  uint8_t* out = mask + beg;
  const bool found = s.found != 0;
  const int gmin = s.gmin, gmax = s.gmax, best_b = s.best_b;
  const int x_hi = found ? min(best_b - gmin, gmax) : gmin;                // x_g_min = (g_min - b_max)/k_max ; :277-278
  for (int64_t i = (int64_t)blockIdx.x * kPT + threadIdx.x; i < V; i += (int64_t)gridDim.x * kPT) {
    uint8_t m = 255;
    const int ix = img[i];
    if (found && ix >= gmin && ix < x_hi) {
      const int y0_up = min(-ix + best_b, gmax + 1);                       // :280
      if ((int)pr[i] < y0_up) m = 0;                                       // :281
    }
    out[i] = m;
  }
}

}  // namespace

M3D_API size_t m3d_otsu2d_workspace_bytes(int num_rois, int max_gray_range) {
  if (num_rois <= 0 || max_gray_range <= 0) return 256;
  return (size_t)num_rois * 3 * 2 * (size_t)max_gray_range * sizeof(unsigned long long) + (size_t)num_rois * sizeof(OtsuState) + 512;
}

M3D_API int m3d_otsu2d_batch(const uint16_t* d_image, const uint16_t* d_prm, const int64_t* d_offsets, int num_rois,
                             int max_gray_range, uint8_t* d_mask, int32_t* d_kb, int32_t* d_status, void* d_ws, size_t ws_bytes,
                             void* stream) {
  if (num_rois < 0 || max_gray_range <= 0 || max_gray_range > 65536) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_image || !d_prm || !d_offsets || !d_mask || !d_kb || !d_status || !d_ws) return M3D_EINVAL;
  if (ws_bytes < m3d_otsu2d_workspace_bytes(num_rois, max_gray_range)) return M3D_EWORKSPACE;
  if (num_rois > 65535) return M3D_EUNSUPPORTED;
  hipStream_t st = m3d::as_stream(stream);
  unsigned long long* ws = (unsigned long long*)m3d::align_up((size_t)d_ws, 256);
  const size_t hist_bytes = (size_t)num_rois * 3 * 2 * (size_t)max_gray_range * sizeof(unsigned long long);
  OtsuState* state = (OtsuState*)((char*)ws + hist_bytes);
  (void)hipMemsetAsync(ws, 0, hist_bytes, st);
  hipLaunchKernelGGL(otsu_init_kernel, dim3((num_rois + 255) / 256), dim3(256), 0, st, state, num_rois);
  // the crop sizes live on the device: a fixed number of chunks per RoI, each workgroup walks its share with a grid stride
  // (a chunk of a small crop is a few hundred voxels; the histogram merge of an LDS partial costs O(G) per workgroup)
  const int chunks = num_rois >= 2048 ? 1 : (num_rois >= 512 ? 4 : (num_rois >= 64 ? 16 : 64));
  const dim3 grid(chunks, num_rois);
  const int lds_g = max_gray_range < 1024 ? max_gray_range : 1024;          // 3 * 2 * 1024 * 8 B = 48 KB
  const size_t lds = sizeof(unsigned long long) * 3 * 2 * (size_t)lds_g;
  hipLaunchKernelGGL(otsu_minmax_kernel, grid, dim3(kPT), 0, st, d_image, d_prm, d_offsets, state);
  hipLaunchKernelGGL(otsu_hist_kernel, grid, dim3(kPT), lds, st, d_image, d_prm, d_offsets, max_gray_range, lds_g, ws, state);
  hipLaunchKernelGGL(otsu_eval_kernel, dim3(num_rois), dim3(kOtsuThreads), 0, st, d_offsets, max_gray_range, ws, state, d_kb, d_status);
  hipLaunchKernelGGL(otsu_mask_kernel, grid, dim3(kPT), 0, st, d_image, d_prm, d_offsets, max_gray_range, (const OtsuState*)state,
                     d_mask);
  return m3d::check_launch("otsu2d_batch");
}
