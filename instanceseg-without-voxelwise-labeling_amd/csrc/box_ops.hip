// 3D box operations for gfx950: greedy NMS, IoU matrix, box decode/clip, RPN proposal generation.
//
// Reference semantics: lib/utils/cython_nms_3d.pyx:39-159, lib/utils/cython_bbox_3d.pyx:32-80,
// lib/utils/boxes_3d.py:144-225, lib/modeling/generate_proposals_3d.py:19-192.
// Everything here is integer/index work on top of a handful of fp32 expressions whose operation order is
// the contract with the reference: this file is compiled with -ffp-contract=off and uses IEEE divides.
//
// NMS design: (1) volumes + a rank sort (O(N^2) compares, embarrassingly parallel, gives a deterministic
// total order incl. the tie rule), (2) upper-triangular 64x64 IoU bitmask tiles, (3) one workgroup walks
// the rows in 64-row chunks staged in LDS: a scalar 64-step loop resolves the diagonal tile, then all
// lanes OR the surviving rows into the "removed" bitmap, (4) flags are mapped back to input order and
// compacted with a block scan.  That form keeps the whole N x N/64 bitmask and a 64-row chunk of it in LDS: N <= 16384.
//
// Beyond 16384 boxes (cross-tile NMS of a whole volume's detections: tools/binarization_nuclei.py:81, tools/binarization_soma.py:57,
// lib/core/test.py:159 call nms_3d on every tile's rows at once, unbounded) the same greedy rule runs BLOCKED, with no N x N mask:
// the sorted rows are cut into chunks of 1024; (2') one launch computes every chunk's own 1024 x 1024 diagonal bitmask; then per
// chunk (3a) one workgroup resolves the chunk against the global "removed" bitmap (visiting only rows that are still alive) and
// writes the chunk's KEPT boxes, (3b) a wide launch tests every later, still-alive box against those kept boxes (IoU on the fly,
// first hit wins) and ORs the hits into the bitmap.  A box is removed iff an earlier KEPT box overlaps it by >= thresh - the
// reference's loop (pyx:67-95), bit for bit, in O(kept x alive) IoUs.  N <= 2^20.
#include "box_common.h"

namespace {
using namespace m3dbox;

constexpr int kMaxNms = 16384;          // the one-workgroup resolve (LDS: 64 rows x N / 64 words)
constexpr int kMaxNmsBlocked = 1 << 20; // the blocked form (O(N^2) rank sort: ~0.1 s at the cap)
constexpr int kNmsChunk = 1024;         // rows per chunk of the blocked form = 16 bitmask words
constexpr int kNmsChunkWords = kNmsChunk / 64;

// ---- (1) volumes + keys ------------------------------------------------------------------------------
__global__ void nms_prepare_kernel(const float* __restrict__ dets, int n_max, const int* __restrict__ d_n, int by_volume,
                                   float* __restrict__ vol, float* __restrict__ key) {
  const int n = d_n ? *d_n : n_max;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* d = dets + 7 * (size_t)i;
  float a = d[3] - d[0]; a = a + 1.0f;      // pyx:48, NumPy fp32 left-to-right
  float b = d[4] - d[1]; b = b + 1.0f;
  float c = d[5] - d[2]; c = c + 1.0f;
  float ab = a * b;
  float v = ab * c;
  vol[i] = v;
  key[i] = by_volume ? v : d[6];
}

// ---- rank sort: order[rank] = i ; sorted boxes -----------------------------------------------------
__global__ __launch_bounds__(256) void nms_rank_kernel(const float* __restrict__ dets, const float* __restrict__ vol,
                                                       const float* __restrict__ key, int n_max, const int* __restrict__ d_n,
                                                       int* __restrict__ order, SBox* __restrict__ sboxes) {
  __shared__ float skey[256];
  const int n = d_n ? *d_n : n_max;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= n) return;
  const float ki = i < n ? key[i] : 0.f;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + threadIdx.x;
    skey[threadIdx.x] = j < n ? key[j] : 0.f;
    __syncthreads();
    const int m = min(256, n - j0);
    if (i < n)
      for (int t = 0; t < m; ++t) rank += key_before(skey[t], j0 + t, ki, i) ? 1 : 0;
    __syncthreads();
  }
  if (i < n) {
    order[rank] = i;
    const float* d = dets + 7 * (size_t)i;
    SBox s{d[0], d[1], d[2], d[3], d[4], d[5], vol[i], 0.f};
    sboxes[rank] = s;
  }
}

// ---- (2) IoU bitmask over sorted boxes: mask[i*nblk + cb] bit t <=> box (cb*64+t) is suppressed by i ----
__global__ __launch_bounds__(64) void nms_mask_kernel(const SBox* __restrict__ sboxes, int n_max, const int* __restrict__ d_n,
                                                      float thresh, int nblk, unsigned long long* __restrict__ mask) {
  const int n = d_n ? *d_n : n_max;
  const int cb = blockIdx.x, rb = blockIdx.y;
  if (rb * 64 >= n || cb * 64 >= n) return;
  const int i = rb * 64 + threadIdx.x;
  if (cb < rb) {
    if (i < n) mask[(size_t)i * nblk + cb] = 0ull;
    return;
  }
  __shared__ SBox cols[64];
  const int jc = cb * 64 + threadIdx.x;
  if (jc < n) cols[threadIdx.x] = sboxes[jc];
  __syncthreads();
  if (i >= n) return;
  const SBox bi = sboxes[i];
  unsigned long long bits = 0ull;
  const int m = min(64, n - cb * 64);
  for (int t = 0; t < m; ++t) {
    const int j = cb * 64 + t;
    if (j <= i) continue;
    const SBox bj = cols[t];
    const float xx1 = fmax32(bi.x1, bj.x1), yy1 = fmax32(bi.y1, bj.y1), zz1 = fmax32(bi.z1, bj.z1);   // pyx:82-87
    const float xx2 = fmin32(bi.x2, bj.x2), yy2 = fmin32(bi.y2, bj.y2), zz2 = fmin32(bi.z2, bj.z2);
    float w = xx2 - xx1; w = w + 1.0f; w = fmax32(0.0f, w);                                             // pyx:88-90
    float h = yy2 - yy1; h = h + 1.0f; h = fmax32(0.0f, h);
    float s = zz2 - zz1; s = s + 1.0f; s = fmax32(0.0f, s);
    float inter = w * h; inter = inter * s;                                                            // pyx:91
    float uni = bi.vol + bj.vol; uni = uni - inter;
    const float ovr = inter / uni;                                                                     // pyx:92
    if (ovr >= thresh) bits |= 1ull << t;                                                              // pyx:93
  }
  mask[(size_t)i * nblk + cb] = bits;
}

// ---- (3)+(4) sequential resolve + compaction; ONE workgroup of 1024 threads ---------------------------
__global__ __launch_bounds__(1024) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ order,
                                                        int n_max, const int* __restrict__ d_n, int nblk_alloc,
                                                        int64_t* __restrict__ keep, int32_t* __restrict__ num_keep,
                                                        unsigned char* __restrict__ flag /* [n] scratch, input order */,
                                                        int keep_limit) {
  extern __shared__ unsigned long long lds[];   // [64 rows][nblk] chunk + removed[nblk] + misc
  const int n = d_n ? *d_n : n_max;
  const int nblk = (n + 63) / 64;
  unsigned long long* chunk = lds;
  unsigned long long* removed = lds + (size_t)64 * nblk_alloc;
  __shared__ unsigned long long kept_word;
  __shared__ int scan_tmp[1024];
  __shared__ int scan_base;
  for (int w = threadIdx.x; w < nblk; w += blockDim.x) removed[w] = 0ull;
  __syncthreads();
  for (int c = 0; c < nblk; ++c) {
    const int rows = min(64, n - c * 64);
    // stage rows [c*64, c*64+rows) x words [c, nblk)
    const int wcount = nblk - c;
    for (int e = threadIdx.x; e < rows * wcount; e += blockDim.x) {
      const int r = e / wcount, w = c + e % wcount;
      chunk[(size_t)r * nblk_alloc + w] = mask[(size_t)(c * 64 + r) * nblk_alloc + w];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long rem = removed[c], kept = 0ull;
      for (int b = 0; b < rows; ++b) {
        if (!((rem >> b) & 1ull)) {
          kept |= 1ull << b;
          rem |= chunk[(size_t)b * nblk_alloc + c];
        }
      }
      removed[c] = rem;
      kept_word = kept;
    }
    __syncthreads();
    const unsigned long long kept = kept_word;
    for (int w = c + 1 + threadIdx.x; w < nblk; w += blockDim.x) {
      unsigned long long acc = removed[w];
      unsigned long long kk = kept;
      while (kk) {
        const int b = __ffsll((long long)kk) - 1;
        kk &= kk - 1;
        acc |= chunk[(size_t)b * nblk_alloc + w];
      }
      removed[w] = acc;
    }
    // flags in input order for this chunk
    if (threadIdx.x < rows) flag[order[c * 64 + threadIdx.x]] = (unsigned char)((kept >> threadIdx.x) & 1ull);
    __syncthreads();
  }
  // compaction in ascending input index (np.where(suppressed == 0)[0], pyx:96)
  if (threadIdx.x == 0) scan_base = 0;
  __threadfence_block();
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int f = (i < n) ? flag[i] : 0;
    scan_tmp[threadIdx.x] = f;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
      int v = threadIdx.x >= off ? scan_tmp[threadIdx.x - off] : 0;
      __syncthreads();
      scan_tmp[threadIdx.x] += v;
      __syncthreads();
    }
    const int pos = scan_base + scan_tmp[threadIdx.x] - f;
    if (f && (keep_limit <= 0 || pos < keep_limit)) keep[pos] = i;
    __syncthreads();
    if (threadIdx.x == 1023) scan_base += scan_tmp[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *num_keep = (keep_limit > 0 && scan_base > keep_limit) ? keep_limit : scan_base;
}

// ================================ blocked greedy NMS (N > 16384) ==========================================================
// (2') diagonal bitmasks: dmask[i * 16 + cw] bit t <=> sorted box (chunk(i) * 1024 + cw * 64 + t) is suppressed by sorted box i (t later than i)
__global__ __launch_bounds__(64) void nms_diag_mask_kernel(const SBox* __restrict__ sboxes, int n_max, const int* __restrict__ d_n,
                                                           float thresh, unsigned long long* __restrict__ dmask) {
  const int n = d_n ? *d_n : n_max;
  const int cw = blockIdx.x, rb = blockIdx.y;                 // column word inside the chunk, 64-row group (global)
  const int cb = (rb / kNmsChunkWords) * kNmsChunkWords + cw; // global 64-column block
  const int i = rb * 64 + threadIdx.x;
  if (rb * 64 >= n) return;
  if (cb < rb || cb * 64 >= n) {
    if (i < n) dmask[(size_t)i * kNmsChunkWords + cw] = 0ull;
    return;
  }
  __shared__ SBox cols[64];
  const int jc = cb * 64 + threadIdx.x;
  if (jc < n) cols[threadIdx.x] = sboxes[jc];
  __syncthreads();
  if (i >= n) return;
  const SBox bi = sboxes[i];
  unsigned long long bits = 0ull;
  const int m = min(64, n - cb * 64);
  for (int t = 0; t < m; ++t) {
    if (cb * 64 + t <= i) continue;
    if (nms_suppresses(bi, cols[t], thresh)) bits |= 1ull << t;
  }
  dmask[(size_t)i * kNmsChunkWords + cw] = bits;
}

__global__ void nms_blocked_init_kernel(unsigned long long* __restrict__ removed, int nwords) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < nwords) removed[w] = 0ull;
}

// (3a) chunk c: resolve rows [c * 1024, ...) against removed[]; kept boxes -> kept_boxes[0 .. *kept_count), flags in input order
__global__ __launch_bounds__(1024) void nms_chunk_resolve_kernel(const unsigned long long* __restrict__ dmask, const SBox* __restrict__ sboxes,
                                                                 const int* __restrict__ order, int n_max, const int* __restrict__ d_n,
                                                                 int chunk, unsigned long long* __restrict__ removed,
                                                                 SBox* __restrict__ kept_boxes, int* __restrict__ kept_count,
                                                                 unsigned char* __restrict__ flag) {
  extern __shared__ unsigned long long lds[];                 // [1024 rows][16 words]
  __shared__ unsigned long long rem[kNmsChunkWords], keptw[kNmsChunkWords];
  __shared__ int kept_prefix[kNmsChunkWords + 1];
  const int n = d_n ? *d_n : n_max;
  const int r0 = chunk * kNmsChunk;
  if (r0 >= n) { if (threadIdx.x == 0) *kept_count = 0; return; }
  const int rows = min(kNmsChunk, n - r0);
  const int words = (rows + 63) / 64;
  for (int e = threadIdx.x; e < rows * kNmsChunkWords; e += blockDim.x) lds[e] = dmask[(size_t)r0 * kNmsChunkWords + e];
  if (threadIdx.x < kNmsChunkWords) {
    rem[threadIdx.x] = threadIdx.x < words ? removed[chunk * kNmsChunkWords + threadIdx.x] : ~0ull;
    keptw[threadIdx.x] = 0ull;
  }
  __syncthreads();
  for (int g = 0; g < words; ++g) {
    if (threadIdx.x == 0) {
      const int grows = min(64, rows - g * 64);
      const unsigned long long valid = grows == 64 ? ~0ull : ((1ull << grows) - 1ull);
      unsigned long long rm = rem[g], kept = 0ull;
      unsigned long long avail = ~rm & valid;
      while (avail) {                                         // only rows that are still alive are visited
        const int b = __ffsll((long long)avail) - 1;
        kept |= 1ull << b;
        rm |= lds[(size_t)(g * 64 + b) * kNmsChunkWords + g];
        avail = ~rm & valid & ~((2ull << b) - 1ull);          // alive rows after b ((2 << 63) wraps to 0: mask ~(0 - 1) = 0)
        if (b == 63) avail = 0ull;
      }
      rem[g] = rm;
      keptw[g] = kept;
    }
    __syncthreads();
    const unsigned long long kept = keptw[g];
    if (threadIdx.x > g && threadIdx.x < words) {             // later words of the chunk: OR the kept rows' masks
      unsigned long long acc = rem[threadIdx.x], kk = kept;
      while (kk) {
        const int b = __ffsll((long long)kk) - 1;
        kk &= kk - 1;
        acc |= lds[(size_t)(g * 64 + b) * kNmsChunkWords + threadIdx.x];
      }
      rem[threadIdx.x] = acc;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int g = 0; g < kNmsChunkWords; ++g) { kept_prefix[g] = acc; acc += g < words ? __popcll(keptw[g]) : 0; }
    kept_prefix[kNmsChunkWords] = acc;
    *kept_count = acc;
  }
  if (threadIdx.x < words) removed[chunk * kNmsChunkWords + threadIdx.x] = rem[threadIdx.x];
  __syncthreads();
  if ((int)threadIdx.x < rows) {
    const int g = threadIdx.x >> 6, b = threadIdx.x & 63;
    const unsigned long long kw = keptw[g];
    const bool k = (kw >> b) & 1ull;
    flag[order[r0 + threadIdx.x]] = k ? 1 : 0;
    if (k) kept_boxes[kept_prefix[g] + __popcll(kw & ((1ull << b) - 1ull))] = sboxes[r0 + threadIdx.x];
  }
}

// (3b) every alive box after chunk c against the chunk's kept boxes; one wave = one word of the bitmap
__global__ __launch_bounds__(256) void nms_chunk_apply_kernel(const SBox* __restrict__ sboxes, int n_max, const int* __restrict__ d_n,
                                                              int chunk, float thresh, const SBox* __restrict__ kept_boxes,
                                                              const int* __restrict__ kept_count, unsigned long long* __restrict__ removed) {
  __shared__ SBox kb[kNmsChunk];
  const int n = d_n ? *d_n : n_max;
  const int nk = *kept_count;
  const int j0 = (chunk + 1) * kNmsChunk + blockIdx.x * 256;
  if (nk == 0 || j0 >= n) return;
  for (int e = threadIdx.x; e < nk; e += 256) kb[e] = kept_boxes[e];
  __syncthreads();
  const int j = j0 + threadIdx.x;
  const int w = j >> 6;
  const unsigned long long rm = (w * 64 < n) ? removed[w] : ~0ull;
  bool hit = false;
  if (j < n && !((rm >> (j & 63)) & 1ull)) {
    const SBox bj = sboxes[j];
    for (int e = 0; e < nk; ++e)
      if (nms_suppresses(kb[e], bj, thresh)) { hit = true; break; }
  }
  const unsigned long long ball = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && ball && w * 64 < n) removed[w] = rm | ball;    // this wave is the only writer of word w in this launch
}

// (4) kept input indices in ascending order from the flags; ONE workgroup
__global__ __launch_bounds__(1024) void nms_compact_kernel(const unsigned char* __restrict__ flag, int n_max, const int* __restrict__ d_n,
                                                           int64_t* __restrict__ keep, int32_t* __restrict__ num_keep, int keep_limit) {
  __shared__ int wave_sum[16];
  __shared__ int scan_base;
  const int n = d_n ? *d_n : n_max;
  if (threadIdx.x == 0) scan_base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const bool f = i < n && flag[i];
    const unsigned long long ball = __ballot(f);
    if (lane == 0) wave_sum[wv] = __popcll(ball);
    __syncthreads();
    int before = scan_base;
    for (int k = 0; k < wv; ++k) before += wave_sum[k];
    const int pos = before + __popcll(ball & ((1ull << lane) - 1ull));
    if (f && (keep_limit <= 0 || pos < keep_limit)) keep[pos] = i;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wave_sum[k]; scan_base += t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) *num_keep = (keep_limit > 0 && scan_base > keep_limit) ? keep_limit : scan_base;
}

struct NmsWs {
  float* vol; float* key; int* order; SBox* sboxes; unsigned long long* mask; unsigned char* flag;
  // blocked form: mask = the diagonal bitmasks [n][16]
  unsigned long long* removed; SBox* kept_boxes; int* kept_count;
};

inline size_t nms_mask_words_per_row(int n) { return n > kMaxNms ? (size_t)kNmsChunkWords : (size_t)(n + 63) / 64; }

size_t nms_ws_bytes(int n) {
  const size_t nblk = (n + 63) / 64;
  size_t b = 0;
  b += m3d::align_up(sizeof(float) * n, 256) * 2;
  b += m3d::align_up(sizeof(int) * n, 256);
  b += m3d::align_up(sizeof(SBox) * n, 256);
  b += m3d::align_up(sizeof(unsigned long long) * n * nms_mask_words_per_row(n), 256);
  b += m3d::align_up((size_t)n, 256);
  if (n > kMaxNms) {
    b += m3d::align_up(sizeof(unsigned long long) * (nblk + kNmsChunkWords), 256);
    b += m3d::align_up(sizeof(SBox) * kNmsChunk, 256) + 256;
  }
  return b + 256;
}

NmsWs nms_carve(void* ws, int n) {
  const size_t nblk = (n + 63) / 64;
  char* p = (char*)m3d::align_up((size_t)ws, 256);
  NmsWs w{};
  w.vol = (float*)p; p += m3d::align_up(sizeof(float) * n, 256);
  w.key = (float*)p; p += m3d::align_up(sizeof(float) * n, 256);
  w.order = (int*)p; p += m3d::align_up(sizeof(int) * n, 256);
  w.sboxes = (SBox*)p; p += m3d::align_up(sizeof(SBox) * n, 256);
  w.mask = (unsigned long long*)p; p += m3d::align_up(sizeof(unsigned long long) * n * nms_mask_words_per_row(n), 256);
  w.flag = (unsigned char*)p; p += m3d::align_up((size_t)n, 256);
  if (n > kMaxNms) {
    w.removed = (unsigned long long*)p; p += m3d::align_up(sizeof(unsigned long long) * (nblk + kNmsChunkWords), 256);
    w.kept_boxes = (SBox*)p; p += m3d::align_up(sizeof(SBox) * kNmsChunk, 256);
    w.kept_count = (int*)p;
  }
  return w;
}

// the blocked form: 2 + 1 + 1 launches and two per chunk of 1024 sorted rows
int nms_launch_blocked(const float* d_dets, int n_max, const int* d_n, float thresh, int by_volume, int64_t* d_keep, int32_t* d_num_keep,
                       const NmsWs& w, int keep_limit, hipStream_t st) {
  const int nblk = (n_max + 63) / 64, nchunk = (n_max + kNmsChunk - 1) / kNmsChunk;
  hipLaunchKernelGGL(nms_prepare_kernel, dim3((n_max + 255) / 256), dim3(256), 0, st, d_dets, n_max, d_n, by_volume, w.vol, w.key);
  hipLaunchKernelGGL(nms_rank_kernel, dim3((n_max + 255) / 256), dim3(256), 0, st, d_dets, w.vol, w.key, n_max, d_n, w.order, w.sboxes);
  hipLaunchKernelGGL(nms_diag_mask_kernel, dim3(kNmsChunkWords, nblk), dim3(64), 0, st, w.sboxes, n_max, d_n, thresh, w.mask);
  hipLaunchKernelGGL(nms_blocked_init_kernel, dim3((nblk + kNmsChunkWords + 255) / 256), dim3(256), 0, st, w.removed, nblk + kNmsChunkWords);
  const size_t lds = sizeof(unsigned long long) * kNmsChunk * kNmsChunkWords;      // 128 KB
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_chunk_resolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int c = 0; c < nchunk; ++c) {
    hipLaunchKernelGGL(nms_chunk_resolve_kernel, dim3(1), dim3(1024), lds, st, w.mask, w.sboxes, w.order, n_max, d_n, c, w.removed,
                       w.kept_boxes, w.kept_count, w.flag);
    const int after = n_max - (c + 1) * kNmsChunk;
    if (after > 0)
      hipLaunchKernelGGL(nms_chunk_apply_kernel, dim3((after + 255) / 256), dim3(256), 0, st, w.sboxes, n_max, d_n, c, thresh, w.kept_boxes,
                         w.kept_count, w.removed);
  }
  hipLaunchKernelGGL(nms_compact_kernel, dim3(1), dim3(1024), 0, st, w.flag, n_max, d_n, d_keep, d_num_keep, keep_limit);
  return m3d::check_launch("nms3d(blocked)");
}

// n_max: capacity (host); d_n: optional device count (<= n_max).
int nms_launch(const float* d_dets, int n_max, const int* d_n, float thresh, int by_volume, int64_t* d_keep, int32_t* d_num_keep,
               void* d_ws, size_t ws_bytes, int keep_limit, hipStream_t st) {
  if (n_max < 0 || n_max > kMaxNmsBlocked) return n_max < 0 ? M3D_EINVAL : M3D_EUNSUPPORTED;
  if (!d_num_keep) return M3D_EINVAL;
  if (n_max == 0) {
    (void)hipMemsetAsync(d_num_keep, 0, sizeof(int32_t), st);   // boxes_3d.py:366-367
    return m3d::check_launch("nms3d(empty)");
  }
  if (!d_dets || !d_keep || !d_ws) return M3D_EINVAL;
  if (ws_bytes < nms_ws_bytes(n_max)) return M3D_EWORKSPACE;
  const NmsWs w = nms_carve(d_ws, n_max);
  if (n_max > kMaxNms) return nms_launch_blocked(d_dets, n_max, d_n, thresh, by_volume, d_keep, d_num_keep, w, keep_limit, st);
  const int nblk = (n_max + 63) / 64;
  hipLaunchKernelGGL(nms_prepare_kernel, dim3((n_max + 255) / 256), dim3(256), 0, st, d_dets, n_max, d_n, by_volume, w.vol, w.key);
  hipLaunchKernelGGL(nms_rank_kernel, dim3((n_max + 255) / 256), dim3(256), 0, st, d_dets, w.vol, w.key, n_max, d_n, w.order,
                     w.sboxes);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, nblk), dim3(64), 0, st, w.sboxes, n_max, d_n, thresh, nblk, w.mask);
  const size_t lds = sizeof(unsigned long long) * ((size_t)64 * nblk + nblk);
  if (lds > 150 * 1024) return M3D_EUNSUPPORTED;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(1024), lds, st, w.mask, w.order, n_max, d_n, nblk, d_keep, d_num_keep,
                     w.flag, keep_limit);
  return m3d::check_launch("nms3d");
}

// ---- IoU matrix (cython_bbox_3d.pyx:32-80; C typing per lib/utils/cython_bbox_3d.c:2042,2105,2288) -------------
__global__ __launch_bounds__(256) void overlaps_kernel(const float* __restrict__ boxes, int N, const float* __restrict__ query,
                                                       int K, float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)N * K) return;
  const int n = (int)(e / K), k = (int)(e % K);
  const float* b = boxes + 6 * (size_t)n;
  const float* q = query + 6 * (size_t)k;
  float r = 0.f;                                                          // pyx:46
  const float box_volume = (float)((((double)(q[3] - q[0]) + 1.0) * ((double)(q[4] - q[1]) + 1.0)) *
                                   ((double)(q[5] - q[2]) + 1.0));         // pyx:52-56
  const float iw = (float)((double)(fmin32(b[3], q[3]) - fmax32(b[0], q[0])) + 1.0);   // pyx:58-61
  if (iw > 0) {
    const float ih = (float)((double)(fmin32(b[4], q[4]) - fmax32(b[1], q[1])) + 1.0);
    if (ih > 0) {
      const float is = (float)((double)(fmin32(b[5], q[5]) - fmax32(b[2], q[2])) + 1.0);
      if (is > 0) {
        float inter = iw * ih; inter = inter * is;
        const double uv = ((((double)(b[3] - b[0]) + 1.0) * ((double)(b[4] - b[1]) + 1.0)) * ((double)(b[5] - b[2]) + 1.0) +
                           (double)box_volume) - (double)inter;            // pyx:73-78
        r = (float)((double)inter / uv);                                   // pyx:79
      }
    }
  }
  out[e] = r;
}

__global__ __launch_bounds__(256) void transform_kernel(const float* __restrict__ boxes, const float* __restrict__ deltas, int n,
                                                        int classes, XformParams p, float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * classes) return;
  const int i = e / classes;
  float o[6];
  decode_one(boxes + 6 * (size_t)i, deltas + 6 * (size_t)e, p, o);
#pragma unroll
  for (int c = 0; c < 6; ++c) out[6 * (size_t)e + c] = o[c];
}

// The box head's outputs in one launch (fast_rcnn_heads.py:42-45 + core/test.py:225,250-251): `outs` [M, nc + 6 nc] is the ONE GEMM of
// cls_score and bbox_pred -> cls = softmax over the nc scores (torch's formula: exp(x - max) / sum, the sum in ascending class order),
// bbox = the raw deltas, pred = decode + clip of every class's deltas around the RoI (rois [M,7] = (batch, x1..z2): columns 1..6).
// One thread per (row, class).  Replaces a slice copy, torch.softmax, two more copies and the decode launch.
__global__ __launch_bounds__(256) void head_outputs_kernel(const float* __restrict__ outs, const float* __restrict__ rois, int M, int nc,
                                                           XformParams p, float* __restrict__ cls, float* __restrict__ bbox,
                                                           float* __restrict__ pred) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * nc) return;
  const int i = e / nc, j = e - i * nc;
  const float* o = outs + (size_t)i * (7 * nc);
  float mx = o[0];
  for (int k = 1; k < nc; ++k) mx = fmaxf(mx, o[k]);
  float sum = 0.f;
  for (int k = 0; k < nc; ++k) sum += expf(o[k] - mx);
  cls[e] = expf(o[j] - mx) / sum;
  const float* d = o + nc + 6 * j;
  float q[6];
  decode_one(rois + 7 * (size_t)i + 1, d, p, q);
#pragma unroll
  for (int c = 0; c < 6; ++c) { bbox[6 * (size_t)e + c] = d[c]; pred[6 * (size_t)e + c] = q[c]; }
}

// ======================================================================================================
// RPN proposals (generate_proposals_3d.py:19-192)
// ======================================================================================================
struct SelState {           // device-resident radix-select state
  unsigned long long prefix;   // selected high bits so far
  unsigned int remaining;      // how many still to take inside the current prefix bucket
  unsigned int done;           // 1 once every element with the prefix is selected
  unsigned int hist[256];
  unsigned int count;          // compaction cursor
  unsigned int nvalid;         // after the filter
};

// memory index m = a*SHW + pos  <->  flat index f = pos*A + a  (generate_proposals_3d.py:121,129)
__global__ __launch_bounds__(256) void sel_hist_kernel(const float* __restrict__ scores, int A, int SHW, int pass, SelState* st) {
  __shared__ unsigned int h[256];
  if (st->done) return;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int shift = 56 - 8 * pass;
  const unsigned long long prefix = st->prefix;
  const long long total = (long long)A * SHW;
  for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (long long)gridDim.x * blockDim.x) {
    const unsigned int a = (unsigned int)(m / SHW), pos = (unsigned int)(m % SHW);
    const unsigned long long key = make_key(scores[m], pos * (unsigned int)A + a);
    if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&h[(unsigned int)(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void sel_pick_kernel(int pass, SelState* st) {
  if (threadIdx.x != 0 || st->done) return;
  const int shift = 56 - 8 * pass;
  unsigned int rem = st->remaining;
  unsigned int total = 0;
  for (int d = 0; d < 256; ++d) total += st->hist[d];
  if (total <= rem) {           // everything under the prefix is selected (also covers total elements <= K)
    st->done = 1;
    // prefix stays: lower bits zero => threshold = prefix
  } else {
    for (int d = 255; d >= 0; --d) {
      const unsigned int c = st->hist[d];
      if (c >= rem) { st->prefix |= (unsigned long long)d << shift; break; }
      rem -= c;
    }
    st->remaining = rem;
    if (pass == 7) st->done = 1;
  }
  for (int d = 0; d < 256; ++d) st->hist[d] = 0;
}

__global__ void sel_init_kernel(SelState* st, unsigned int k) {
  if (threadIdx.x == 0) { st->prefix = 0; st->remaining = k; st->done = 0; st->count = 0; st->nvalid = 0; }
  if (threadIdx.x < 256) st->hist[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void sel_compact_kernel(const float* __restrict__ scores, int A, int SHW, SelState* st,
                                                          unsigned long long* __restrict__ keys, unsigned int cap) {
  const unsigned long long thr = st->prefix;
  const long long total = (long long)A * SHW;
  for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (long long)gridDim.x * blockDim.x) {
    const unsigned int a = (unsigned int)(m / SHW), pos = (unsigned int)(m % SHW);
    const unsigned long long key = make_key(scores[m], pos * (unsigned int)A + a);
    if (key >= thr) {
      const unsigned int slot = atomicAdd(&st->count, 1u);
      if (slot < cap) keys[slot] = key;
    }
  }
}

// rank sort of the (distinct) selected keys, descending
__global__ __launch_bounds__(256) void sel_rank_kernel(const unsigned long long* __restrict__ keys, const SelState* st,
                                                       unsigned int cap, unsigned long long* __restrict__ sorted) {
  __shared__ unsigned long long sk[256];
  const unsigned int n = min(st->count, cap);
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= n) return;
  const unsigned long long ki = i < n ? keys[i] : 0ull;
  unsigned int rank = 0;
  for (unsigned int j0 = 0; j0 < n; j0 += 256) {
    sk[threadIdx.x] = (j0 + threadIdx.x < n) ? keys[j0 + threadIdx.x] : 0ull;
    __syncthreads();
    const unsigned int m = min(256u, n - j0);
    for (unsigned int t = 0; t < m; ++t) rank += sk[t] > ki ? 1u : 0u;
    __syncthreads();
  }
  if (i < n) sorted[rank] = ki;
}

// decode + clip + filter for the sorted candidates; one thread each
__global__ __launch_bounds__(256) void prop_decode_kernel(const unsigned long long* __restrict__ sorted, const SelState* st,
                                                          unsigned int cap, const float* __restrict__ deltas, PropParams p,
                                                          float* __restrict__ boxes /*[cap,6]*/, unsigned char* __restrict__ valid) {
  const unsigned int n = min(st->count, cap);
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = sorted[i];
  const unsigned int flat = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull);
  const int a = flat % p.A;
  const int pos = flat / p.A;
  const int w = pos % p.W, h = (pos / p.W) % p.H, s = pos / (p.W * p.H);
  const double sx = (double)w * p.stride, sy = (double)h * p.stride, sz = (double)s * p.stride;   // :68-77
  const double* an = p.anchors + 6 * a;
  float b[6] = {(float)(an[0] + sx), (float)(an[1] + sy), (float)(an[2] + sz),                      // :88, boxes_3d.py:175
                (float)(an[3] + sx), (float)(an[4] + sy), (float)(an[5] + sz)};
  const size_t SHW = (size_t)p.S * p.H * p.W;
  float d[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) d[c] = deltas[(size_t)(a * 6 + c) * SHW + pos];                      // :121
  float o[6];
  decode_one(b, d, p.xf, o);                                                                      // :149,154
#pragma unroll
  for (int c = 0; c < 6; ++c) boxes[6 * (size_t)i + c] = o[c];
  // _filter_boxes_3d :180-192 (y/z centres use the x-side `ss`: reference behaviour, kept)
  const double ms = p.min_size * p.im_scale;
  float ss = o[3] - o[0]; ss = ss + 1.0f;
  const float half = ss / 2.0f;
  const float xc = o[0] + half, yc = o[1] + half, zc = o[2] + half;
  valid[i] = ((double)ss >= ms && (double)xc < p.im_w && (double)yc < p.im_h && (double)zc < p.im_s) ? 1 : 0;
}

// ordered compaction of valid candidates into dets [n,7] + flat index; ONE workgroup
__global__ __launch_bounds__(1024) void prop_compact_kernel(const unsigned long long* __restrict__ sorted, SelState* st,
                                                            unsigned int cap, const float* __restrict__ boxes,
                                                            const unsigned char* __restrict__ valid, float* __restrict__ dets,
                                                            int64_t* __restrict__ flat_idx) {
  __shared__ int tmp[1024];
  __shared__ int base_s;
  const unsigned int n = min(st->count, cap);
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (unsigned int b0 = 0; b0 < n; b0 += 1024) {
    const unsigned int i = b0 + threadIdx.x;
    const int f = (i < n) ? valid[i] : 0;
    tmp[threadIdx.x] = f;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      int v = (int)threadIdx.x >= off ? tmp[threadIdx.x - off] : 0;
      __syncthreads();
      tmp[threadIdx.x] += v;
      __syncthreads();
    }
    if (f) {
      const int pos = base_s + tmp[threadIdx.x] - 1;
      const unsigned long long key = sorted[i];
#pragma unroll
      for (int c = 0; c < 6; ++c) dets[7 * (size_t)pos + c] = boxes[6 * (size_t)i + c];
      dets[7 * (size_t)pos + 6] = bits_score((unsigned int)(key >> 32));
      flat_idx[pos] = (int64_t)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
    }
    __syncthreads();
    if (threadIdx.x == 1023) base_s += tmp[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) st->nvalid = (unsigned int)base_s;
}

__global__ __launch_bounds__(256) void prop_gather_kernel(const float* __restrict__ dets, const int64_t* __restrict__ flat_idx,
                                                          const int64_t* __restrict__ keep, const int32_t* __restrict__ num_keep,
                                                          int batch_index, float* __restrict__ rois, float* __restrict__ probs,
                                                          int64_t* __restrict__ keep_idx, int32_t* __restrict__ d_num) {
  const int n = *num_keep;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *d_num = n;
  if (i >= n) return;
  const int64_t k = keep[i];
  rois[7 * (size_t)i + 0] = (float)batch_index;                           // :98-100
#pragma unroll
  for (int c = 0; c < 6; ++c) rois[7 * (size_t)i + 1 + c] = dets[7 * (size_t)k + c];
  probs[i] = dets[7 * (size_t)k + 6];
  keep_idx[i] = flat_idx[k];                                              // :160,174-175
}

// all candidates kept (nms_thresh <= 0): identity keep list
__global__ void prop_iota_kernel(const SelState* st, int limit, int64_t* keep, int32_t* num_keep) {
  int n = (int)st->nvalid;
  if (limit > 0 && n > limit) n = limit;
  for (int i = threadIdx.x; i < n; i += blockDim.x) keep[i] = i;
  if (threadIdx.x == 0) *num_keep = n;
}

struct PropWs {
  SelState* st; unsigned long long* keys; unsigned long long* sorted; float* boxes; unsigned char* valid; float* dets;
  int64_t* flat_idx; int64_t* keep; int32_t* num_keep; void* nms_ws; size_t nms_bytes;
};

size_t prop_ws_bytes(int K) {
  size_t b = 256;
  b += m3d::align_up(sizeof(SelState), 256);
  b += m3d::align_up(sizeof(unsigned long long) * K, 256) * 2;
  b += m3d::align_up(sizeof(float) * 6 * K, 256);
  b += m3d::align_up((size_t)K, 256);
  b += m3d::align_up(sizeof(float) * 7 * K, 256);
  b += m3d::align_up(sizeof(int64_t) * K, 256) * 2;
  b += 256;
  b += nms_ws_bytes(K);
  return b;
}

PropWs prop_carve(void* ws, int K) {
  char* p = (char*)m3d::align_up((size_t)ws, 256);
  PropWs w;
  w.st = (SelState*)p; p += m3d::align_up(sizeof(SelState), 256);
  w.keys = (unsigned long long*)p; p += m3d::align_up(sizeof(unsigned long long) * K, 256);
  w.sorted = (unsigned long long*)p; p += m3d::align_up(sizeof(unsigned long long) * K, 256);
  w.boxes = (float*)p; p += m3d::align_up(sizeof(float) * 6 * K, 256);
  w.valid = (unsigned char*)p; p += m3d::align_up((size_t)K, 256);
  w.dets = (float*)p; p += m3d::align_up(sizeof(float) * 7 * K, 256);
  w.flat_idx = (int64_t*)p; p += m3d::align_up(sizeof(int64_t) * K, 256);
  w.keep = (int64_t*)p; p += m3d::align_up(sizeof(int64_t) * K, 256);
  w.num_keep = (int32_t*)p; p += 256;
  w.nms_ws = p; w.nms_bytes = nms_ws_bytes(K);
  return w;
}

}  // namespace

M3D_API size_t m3d_nms3d_workspace_bytes(int n) { return n <= 0 ? 256 : nms_ws_bytes(n); }

M3D_API int m3d_nms3d(const float* d_dets, int n, float thresh, int by_volume, int64_t* d_keep, int32_t* d_num_keep, void* d_ws,
                      size_t ws_bytes, void* stream) {
  return nms_launch(d_dets, n, nullptr, thresh, by_volume, d_keep, d_num_keep, d_ws, ws_bytes, 0, m3d::as_stream(stream));
}

M3D_API int m3d_bbox_overlaps3d(const float* d_boxes, int n, const float* d_query, int k, float* d_out, void* stream) {
  if (n < 0 || k < 0) return M3D_EINVAL;
  if ((long long)n * k == 0) return M3D_OK;
  if (!d_boxes || !d_query || !d_out) return M3D_EINVAL;
  const long long total = (long long)n * k;
  hipLaunchKernelGGL(overlaps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, m3d::as_stream(stream), d_boxes, n,
                     d_query, k, d_out);
  return m3d::check_launch("bbox_overlaps3d");
}

M3D_API int m3d_bbox_transform3d(const float* d_boxes, const float* d_deltas, int n, int classes, const double* weights,
                                 double xform_clip, double clip_slices, double clip_height, double clip_width, float* d_out,
                                 void* stream) {
  if (n < 0 || classes <= 0 || !weights) return M3D_EINVAL;
  if (n == 0) return M3D_OK;                                              // boxes_3d.py:172-173
  if (!d_boxes || !d_deltas || !d_out) return M3D_EINVAL;
  XformParams p;
  for (int i = 0; i < 6; ++i) p.w[i] = weights[i];
  p.clip = xform_clip; p.cs = clip_slices; p.ch = clip_height; p.cw = clip_width;
  hipLaunchKernelGGL(transform_kernel, dim3((n * classes + 255) / 256), dim3(256), 0, m3d::as_stream(stream), d_boxes, d_deltas,
                     n, classes, p, d_out);
  return m3d::check_launch("bbox_transform3d");
}

M3D_API int m3d_box_head_outputs(const float* d_outs, const float* d_rois, int num_rois, int num_classes, const double* weights,
                                 double xform_clip, double clip_slices, double clip_height, double clip_width, float* d_cls, float* d_bbox,
                                 float* d_pred, void* stream) {
  if (num_rois < 0 || num_classes < 1 || !weights) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_outs || !d_rois || !d_cls || !d_bbox || !d_pred) return M3D_EINVAL;
  XformParams p;
  for (int i = 0; i < 6; ++i) p.w[i] = weights[i];
  p.clip = xform_clip; p.cs = clip_slices; p.ch = clip_height; p.cw = clip_width;
  hipLaunchKernelGGL(head_outputs_kernel, dim3((num_rois * num_classes + 255) / 256), dim3(256), 0, m3d::as_stream(stream), d_outs, d_rois,
                     num_rois, num_classes, p, d_cls, d_bbox, d_pred);
  return m3d::check_launch("box_head_outputs");
}

M3D_API size_t m3d_generate_proposals3d_workspace_bytes(int A, int S, int H, int W, int pre_nms_topN) {
  long long total = (long long)A * S * H * W;
  long long K = (pre_nms_topN <= 0 || pre_nms_topN >= total) ? total : pre_nms_topN;
  if (K <= 0) K = 1;
  return prop_ws_bytes((int)K);
}

M3D_API int m3d_generate_proposals3d(const float* d_scores, const float* d_deltas, int A, int S, int H, int W,
                                     const double* anchors, double feat_stride, const double* im_info, int pre_nms_topN,
                                     int post_nms_topN, float nms_thresh, double min_size, double xform_clip, int batch_index,
                                     float* d_rois, float* d_probs, int64_t* d_keep_idx, int32_t* d_num, void* d_ws,
                                     size_t ws_bytes, void* stream) {
  if (A <= 0 || A > 64 || S <= 0 || H <= 0 || W <= 0 || !anchors || !im_info) return M3D_EINVAL;
  if (!d_scores || !d_deltas || !d_rois || !d_probs || !d_keep_idx || !d_num || !d_ws) return M3D_EINVAL;
  const long long total = (long long)A * S * H * W;
  if (total >= 0xFFFFFFFFll) return M3D_EUNSUPPORTED;
  const long long Kll = (pre_nms_topN <= 0 || pre_nms_topN >= total) ? total : pre_nms_topN;   // :135
  if (Kll > kMaxNms) return M3D_EUNSUPPORTED;
  const int K = (int)Kll;
  if (ws_bytes < prop_ws_bytes(K)) return M3D_EWORKSPACE;
  hipStream_t st = m3d::as_stream(stream);
  const PropWs w = prop_carve(d_ws, K);
  const int SHW = S * H * W;
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(256), 0, st, w.st, (unsigned)K);
  for (int pass = 0; pass < 8; ++pass) {
    hipLaunchKernelGGL(sel_hist_kernel, dim3(grid), dim3(256), 0, st, d_scores, A, SHW, pass, w.st);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(64), 0, st, pass, w.st);
  }
  hipLaunchKernelGGL(sel_compact_kernel, dim3(grid), dim3(256), 0, st, d_scores, A, SHW, w.st, w.keys, (unsigned)K);
  hipLaunchKernelGGL(sel_rank_kernel, dim3((K + 255) / 256), dim3(256), 0, st, w.keys, w.st, (unsigned)K, w.sorted);
  PropParams p;
  for (int i = 0; i < 6 * A; ++i) p.anchors[i] = anchors[i];
  p.stride = feat_stride; p.im_s = im_info[0]; p.im_h = im_info[1]; p.im_w = im_info[2]; p.im_scale = im_info[3];
  p.min_size = min_size; p.A = A; p.S = S; p.H = H; p.W = W; p.batch_index = batch_index;
  for (int i = 0; i < 6; ++i) p.xf.w[i] = 1.0;                            // :149-150
  p.xf.clip = xform_clip; p.xf.cs = im_info[0]; p.xf.ch = im_info[1]; p.xf.cw = im_info[2];   // :154
  hipLaunchKernelGGL(prop_decode_kernel, dim3((K + 255) / 256), dim3(256), 0, st, w.sorted, w.st, (unsigned)K, d_deltas, p,
                     w.boxes, w.valid);
  hipLaunchKernelGGL(prop_compact_kernel, dim3(1), dim3(1024), 0, st, w.sorted, w.st, (unsigned)K, w.boxes, w.valid, w.dets,
                     w.flat_idx);
  int rc = m3d::check_launch("generate_proposals3d(select/decode)");
  if (rc != M3D_OK) return rc;
  if (nms_thresh > 0) {                                                   // :167-171
    rc = nms_launch(w.dets, K, (const int*)&w.st->nvalid, nms_thresh, 0, w.keep, w.num_keep, w.nms_ws, w.nms_bytes,
                    post_nms_topN, st);
    if (rc != M3D_OK) return rc;
  } else {
    hipLaunchKernelGGL(prop_iota_kernel, dim3(1), dim3(256), 0, st, w.st, 0, w.keep, w.num_keep);
  }
  hipLaunchKernelGGL(prop_gather_kernel, dim3((K + 255) / 256), dim3(256), 0, st, w.dets, w.flat_idx, w.keep, w.num_keep,
                     batch_index, d_rois, d_probs, d_keep_idx, d_num);
  return m3d::check_launch("generate_proposals3d");
}
