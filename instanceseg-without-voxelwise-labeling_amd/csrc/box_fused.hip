// Batched, fused box post-processing for gfx950: ONE launch per stage for a whole batch of tiles, one 1024-thread workgroup
// per tile running every phase of the stage back to back (no host round trips, no per-phase launch boundaries).
//
//   proposals_fused_kernel    GenerateProposalsOp_3d.forward per tile (lib/modeling/generate_proposals_3d.py:19-192):
//                             top-N radix select -> sort -> decode/clip/filter -> nms_3d -> rois / probs / kept flat indices
//   box_results_fused_kernel  box_results_with_nms_and_limit per tile (lib/core/test.py:806-883): score threshold, per-class
//                             nms_3d, DETECTIONS_PER_IM cap, kept anchor indices carried through
//   nms_pack_kernel           nms_3d / nms_3d_volume per item (the cross-tile NMS of lib/core/test.py:159 when every volume is one
//                             tile) writing the padded [cap+1,7] block m3d.shard all-gathers
//
// These stages are latency-bound integer / index work on <= 2048 boxes per tile ("report microseconds, not a roofline
// fraction"): the multi-launch forms in box_ops.hip cost 26 + 8 + 4 launches and three host read-backs PER TILE; here a batch of
// tiles costs three launches in total and the tiles' workgroups run side by side on different CUs.
// Same fp32 operation order, tie rules and index semantics as box_ops.hip (shared helpers in box_common.h; -ffp-contract=off).
#include "box_common.h"

namespace {
using namespace m3dbox;

constexpr int kWG = 1024;
constexpr int kFusedMax = 2048;                  // boxes per item the one-workgroup NMS handles
constexpr int kNblkMax = kFusedMax / 64;

struct NmsScratch {                               // global scratch of ONE item
  float* vol; int* order; SBox* sboxes; unsigned long long* mask; unsigned char* flag; int64_t* keep;
};

__host__ __device__ inline size_t nms_scratch_bytes() {
  return m3d::align_up(sizeof(float) * kFusedMax, 256) + m3d::align_up(sizeof(int) * kFusedMax, 256) +
         m3d::align_up(sizeof(SBox) * kFusedMax, 256) + m3d::align_up(sizeof(unsigned long long) * kFusedMax * kNblkMax, 256) +
         m3d::align_up((size_t)kFusedMax, 256) + m3d::align_up(sizeof(int64_t) * kFusedMax, 256);
}

__device__ inline NmsScratch nms_scratch_carve(char* p) {
  NmsScratch s;
  s.vol = (float*)p; p += m3d::align_up(sizeof(float) * kFusedMax, 256);
  s.order = (int*)p; p += m3d::align_up(sizeof(int) * kFusedMax, 256);
  s.sboxes = (SBox*)p; p += m3d::align_up(sizeof(SBox) * kFusedMax, 256);
  s.mask = (unsigned long long*)p; p += m3d::align_up(sizeof(unsigned long long) * kFusedMax * kNblkMax, 256);
  s.flag = (unsigned char*)p; p += m3d::align_up((size_t)kFusedMax, 256);
  s.keep = (int64_t*)p;
  return s;
}

struct WgLds {                                    // static LDS of the workgroup; phases reuse it
  union {
    float keys[kFusedMax];                        // NMS rank phase
    unsigned int hist[256];                       // radix select
  };
  union {
    unsigned long long chunk[64 * kNblkMax];      // NMS resolve: 64 mask rows
    unsigned long long skeys[kFusedMax];          // proposal sort keys
  };
  unsigned long long removed[kNblkMax];
  unsigned long long kept_word;
  unsigned long long prefix;
  unsigned int remaining, done, count;
  int scan_tmp[kWG];
  int scan_base;
};

// inclusive block scan of one int per thread (Hillis-Steele over LDS); returns this thread's inclusive value
__device__ inline int block_scan_inclusive(int v, int* tmp) {
  tmp[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < kWG; off <<= 1) {
    const int a = (int)threadIdx.x >= off ? tmp[threadIdx.x - off] : 0;
    __syncthreads();
    tmp[threadIdx.x] += a;
    __syncthreads();
  }
  return tmp[threadIdx.x];
}

// Greedy NMS of dets[0..n) (rows of 7 floats) by the whole workgroup: cython_nms_3d.pyx:39-96 (by_volume: :102-159).
// Writes the kept input indices in ascending order to sc.keep (at most keep_limit if > 0) and returns their number.
__device__ int wg_nms(const float* __restrict__ dets, int n, float thresh, int by_volume, int keep_limit, const NmsScratch& sc,
                      WgLds& L) {
  const int tid = threadIdx.x;
  if (n <= 0) return 0;
  // (1) volumes + keys
  for (int i = tid; i < n; i += kWG) {
    const float v = det_volume(dets + 7 * (size_t)i);
    sc.vol[i] = v;
    L.keys[i] = by_volume ? v : dets[7 * (size_t)i + 6];
  }
  __syncthreads();
  // (2) rank sort: descending key, ties in descending index (box_ops.hip key_before)
  for (int i = tid; i < n; i += kWG) {
    const float ki = L.keys[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += key_before(L.keys[j], j, ki, i) ? 1 : 0;
    sc.order[rank] = i;
    const float* d = dets + 7 * (size_t)i;
    sc.sboxes[rank] = SBox{d[0], d[1], d[2], d[3], d[4], d[5], sc.vol[i], 0.f};
  }
  __syncthreads();
  // (3) upper-triangular 64x64 suppression bitmask tiles, one wave per tile
  const int nblk = (n + 63) / 64;
  const int wave = tid >> 6, lane = tid & 63, nwaves = kWG / 64;
  int t = 0;
  for (int rb = 0; rb < nblk; ++rb)
    for (int cb = rb; cb < nblk; ++cb, ++t) {
      if (t % nwaves != wave) continue;
      const int i = rb * 64 + lane;
      if (i >= n) continue;
      const SBox bi = sc.sboxes[i];
      unsigned long long bits = 0ull;
      const int m = min(64, n - cb * 64);
      for (int q = 0; q < m; ++q) {
        const int j = cb * 64 + q;
        if (j <= i) continue;
        if (nms_suppresses(bi, sc.sboxes[j], thresh)) bits |= 1ull << q;
      }
      sc.mask[(size_t)i * kNblkMax + cb] = bits;
    }
  for (int w = tid; w < nblk; w += kWG) L.removed[w] = 0ull;
  __syncthreads();
  // (4) sequential resolve in 64-row chunks staged in LDS
  for (int c = 0; c < nblk; ++c) {
    const int rows = min(64, n - c * 64), wcount = nblk - c;
    for (int e = tid; e < rows * wcount; e += kWG) {
      const int r = e / wcount, w = c + e % wcount;
      L.chunk[r * kNblkMax + w] = sc.mask[(size_t)(c * 64 + r) * kNblkMax + w];
    }
    __syncthreads();
    if (tid == 0) {
      unsigned long long rem = L.removed[c], kept = 0ull;
      for (int b = 0; b < rows; ++b)
        if (!((rem >> b) & 1ull)) { kept |= 1ull << b; rem |= L.chunk[b * kNblkMax + c]; }
      L.removed[c] = rem;
      L.kept_word = kept;
    }
    __syncthreads();
    const unsigned long long kept = L.kept_word;
    for (int w = c + 1 + tid; w < nblk; w += kWG) {
      unsigned long long acc = L.removed[w], kk = kept;
      while (kk) { const int b = __ffsll((long long)kk) - 1; kk &= kk - 1; acc |= L.chunk[b * kNblkMax + w]; }
      L.removed[w] = acc;
    }
    if (tid < rows) sc.flag[sc.order[c * 64 + tid]] = (unsigned char)((kept >> tid) & 1ull);
    __syncthreads();
  }
  // (5) compaction in ascending input index (np.where(suppressed == 0)[0], pyx:96)
  int base = 0;
  for (int b0 = 0; b0 < n; b0 += kWG) {
    const int i = b0 + tid;
    const int f = i < n ? sc.flag[i] : 0;
    const int incl = block_scan_inclusive(f, L.scan_tmp);
    const int pos = base + incl - f;
    if (f && (keep_limit <= 0 || pos < keep_limit)) sc.keep[pos] = i;
    base += L.scan_tmp[kWG - 1];
    __syncthreads();
  }
  return (keep_limit > 0 && base > keep_limit) ? keep_limit : base;
}

// ------------------------------------------------------------------------------------------------ proposals
struct PropFusedArgs {
  const float* scores; const float* deltas;          // [B,A,S,H,W], [B,6A,S,H,W]
  float* rois; float* probs; int64_t* keep_idx; int32_t* num;   // [B,post,7], [B,post], [B,post], [B]
  char* ws; size_t ws_item;                          // per-item scratch
  int K, post, cap_out;                              // pre_nms_topN (clamped), post_nms_topN, rows per item in the outputs
  float nms_thresh;
  int first_batch_index;
  PropParams p;
};

__host__ __device__ inline size_t prop_item_bytes(int K) {
  return m3d::align_up(sizeof(unsigned long long) * K, 256) * 2 + m3d::align_up(sizeof(float) * 6 * K, 256) +
         m3d::align_up((size_t)K, 256) + m3d::align_up(sizeof(float) * 7 * K, 256) + m3d::align_up(sizeof(int64_t) * K, 256) +
         nms_scratch_bytes() + 256;
}

__global__ __launch_bounds__(kWG) void proposals_fused_kernel(PropFusedArgs a) {
  __shared__ WgLds L;
  const int tid = threadIdx.x, b = blockIdx.x;
  const PropParams& p = a.p;
  const int A = p.A, SHW = p.S * p.H * p.W, K = a.K;
  const long long total = (long long)A * SHW;
  const float* scores = a.scores + (size_t)b * total;
  const float* deltas = a.deltas + (size_t)b * total * 6;
  char* w = a.ws + (size_t)b * a.ws_item;
  unsigned long long* keys = (unsigned long long*)w; w += m3d::align_up(sizeof(unsigned long long) * K, 256);
  unsigned long long* sorted = (unsigned long long*)w; w += m3d::align_up(sizeof(unsigned long long) * K, 256);
  float* boxes = (float*)w; w += m3d::align_up(sizeof(float) * 6 * K, 256);
  unsigned char* valid = (unsigned char*)w; w += m3d::align_up((size_t)K, 256);
  float* dets = (float*)w; w += m3d::align_up(sizeof(float) * 7 * K, 256);
  int64_t* flat_idx = (int64_t*)w; w += m3d::align_up(sizeof(int64_t) * K, 256);
  const NmsScratch sc = nms_scratch_carve(w);

  // ---- top-K by 8 radix-256 passes over the 64-bit (score, ~flat index) keys (generate_proposals_3d.py:135-146)
  if (tid == 0) { L.prefix = 0ull; L.remaining = (unsigned)K; L.done = 0u; L.count = 0u; }
  __syncthreads();
  for (int pass = 0; pass < 8; ++pass) {
    if (L.done) break;                                             // uniform: read after a barrier
    if (tid < 256) L.hist[tid] = 0u;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const unsigned long long prefix = L.prefix;
    for (long long m0 = tid; m0 < total; m0 += 4 * kWG) {          // 4 independent loads in flight per lane
      float sv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long long m = m0 + (long long)u * kWG; sv[u] = scores[m < total ? m : 0]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long m = m0 + (long long)u * kWG;
        if (m >= total) break;
        const unsigned int an = (unsigned int)(m / SHW), pos = (unsigned int)(m % SHW);
        const unsigned long long key = make_key(sv[u], pos * (unsigned int)A + an);
        if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&L.hist[(unsigned int)(key >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int rem = L.remaining, tot = 0;
      for (int d = 0; d < 256; ++d) tot += L.hist[d];
      if (tot <= rem) {
        L.done = 1u;                                               // everything under the prefix is selected
      } else {
        for (int d = 255; d >= 0; --d) {
          const unsigned int c = L.hist[d];
          if (c >= rem) { L.prefix |= (unsigned long long)d << shift; break; }
          rem -= c;
        }
        L.remaining = rem;
        if (pass == 7) L.done = 1u;
      }
    }
    __syncthreads();
  }
  // ---- compaction of the selected keys, then rank sort (keys are distinct), descending
  const unsigned long long thr = L.prefix;
  for (long long m0 = tid; m0 < total; m0 += 4 * kWG) {
    float sv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const long long m = m0 + (long long)u * kWG; sv[u] = scores[m < total ? m : 0]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long m = m0 + (long long)u * kWG;
      if (m >= total) break;
      const unsigned int an = (unsigned int)(m / SHW), pos = (unsigned int)(m % SHW);
      const unsigned long long key = make_key(sv[u], pos * (unsigned int)A + an);
      if (key >= thr) {
        const unsigned int slot = atomicAdd(&L.count, 1u);
        if (slot < (unsigned)K) keys[slot] = key;
      }
    }
  }
  __syncthreads();
  const int n = (int)min(L.count, (unsigned)K);
  for (int i = tid; i < n; i += kWG) L.skeys[i] = keys[i];
  __syncthreads();
  for (int i = tid; i < n; i += kWG) {
    const unsigned long long ki = L.skeys[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += L.skeys[j] > ki ? 1 : 0;
    sorted[rank] = ki;
  }
  __syncthreads();
  // ---- decode + clip + _filter_boxes_3d for every candidate (:149-160,180-192)
  for (int i = tid; i < n; i += kWG) {
    const unsigned long long key = sorted[i];
    const unsigned int flat = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull);
    const int an = flat % A, pos = flat / A;
    const int x = pos % p.W, y = (pos / p.W) % p.H, z = pos / (p.W * p.H);
    const double sx = (double)x * p.stride, sy = (double)y * p.stride, sz = (double)z * p.stride;   // :68-77
    const double* anc = p.anchors + 6 * an;
    float bx[6] = {(float)(anc[0] + sx), (float)(anc[1] + sy), (float)(anc[2] + sz),                   // :88, boxes_3d.py:175
                   (float)(anc[3] + sx), (float)(anc[4] + sy), (float)(anc[5] + sz)};
    float d[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) d[c] = deltas[(size_t)(an * 6 + c) * SHW + pos];                    // :121
    float o[6];
    decode_one(bx, d, p.xf, o);                                                                     // :149,154
#pragma unroll
    for (int c = 0; c < 6; ++c) boxes[6 * (size_t)i + c] = o[c];
    const double ms = p.min_size * p.im_scale;
    float ss = o[3] - o[0]; ss = ss + 1.0f;
    const float half = ss / 2.0f;
    const float xc = o[0] + half, yc = o[1] + half, zc = o[2] + half;
    valid[i] = ((double)ss >= ms && (double)xc < p.im_w && (double)yc < p.im_h && (double)zc < p.im_s) ? 1 : 0;
  }
  __syncthreads();
  // ---- ordered compaction of the valid candidates -> dets [nvalid,7] + flat index
  int nvalid = 0;
  for (int b0 = 0; b0 < n; b0 += kWG) {
    const int i = b0 + tid;
    const int f = i < n ? valid[i] : 0;
    const int incl = block_scan_inclusive(f, L.scan_tmp);
    if (f) {
      const int pos = nvalid + incl - 1;
      const unsigned long long key = sorted[i];
#pragma unroll
      for (int c = 0; c < 6; ++c) dets[7 * (size_t)pos + c] = boxes[6 * (size_t)i + c];
      dets[7 * (size_t)pos + 6] = bits_score((unsigned int)(key >> 32));
      flat_idx[pos] = (int64_t)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
    }
    nvalid += L.scan_tmp[kWG - 1];
    __syncthreads();
  }
  // ---- nms_3d + keep[:post_nms_topN] (:167-171), then rois / probs / kept flat indices (:98-100,160,174-175)
  int nk;
  if (a.nms_thresh > 0) {
    nk = wg_nms(dets, nvalid, a.nms_thresh, 0, a.post, sc, L);
  } else {
    nk = (a.post > 0 && nvalid > a.post) ? a.post : nvalid;
    for (int i = tid; i < nk; i += kWG) sc.keep[i] = i;
  }
  __syncthreads();
  if (nk > a.cap_out) nk = a.cap_out;
  float* rois = a.rois + (size_t)b * a.cap_out * 7;
  float* probs = a.probs + (size_t)b * a.cap_out;
  int64_t* kidx = a.keep_idx + (size_t)b * a.cap_out;
  for (int i = tid; i < nk; i += kWG) {
    const int64_t k = sc.keep[i];
    rois[7 * (size_t)i] = (float)(a.first_batch_index + b);
#pragma unroll
    for (int c = 0; c < 6; ++c) rois[7 * (size_t)i + 1 + c] = dets[7 * (size_t)k + c];
    probs[i] = dets[7 * (size_t)k + 6];
    kidx[i] = flat_idx[k];
  }
  if (tid == 0) a.num[b] = nk;
}

// ------------------------------------------------------------------------------------------------ box results
struct BoxResArgs {
  const float* scores; const float* boxes; const int64_t* keep_idx;   // [R,nc], [R,6nc], [R] or NULL (rows of all items)
  const int32_t* offsets;                                             // [B+1] row ranges
  float* cls_boxes; int64_t* cls_keep; int32_t* counts;               // [B,nc,cap,7], [B,nc,cap] or NULL, [B,nc]
  char* ws; size_t ws_item;
  int nc, cap, det_per_im;
  float score_thresh, nms_thresh;
};

__host__ __device__ inline size_t boxres_item_bytes() {
  return m3d::align_up(sizeof(float) * 7 * kFusedMax, 256) + m3d::align_up(sizeof(int) * kFusedMax, 256) + nms_scratch_bytes() + 256;
}

__global__ __launch_bounds__(kWG) void box_results_fused_kernel(BoxResArgs a) {
  __shared__ WgLds L;
  __shared__ float s_thresh;
  __shared__ int s_found;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int r0 = a.offsets[b], R = a.offsets[b + 1] - r0;
  char* w = a.ws + (size_t)b * a.ws_item;
  float* dets = (float*)w; w += m3d::align_up(sizeof(float) * 7 * kFusedMax, 256);
  int* src = (int*)w; w += m3d::align_up(sizeof(int) * kFusedMax, 256);
  const NmsScratch sc = nms_scratch_carve(w);
  float* out0 = a.cls_boxes + (size_t)b * a.nc * a.cap * 7;
  int64_t* keep0 = a.cls_keep ? a.cls_keep + (size_t)b * a.nc * a.cap : nullptr;
  if (tid == 0) a.counts[b * a.nc] = 0;                                // class 0 = background: always empty (:833)
  int total = 0;
  for (int j = 1; j < a.nc; ++j) {
    // inds = scores[:, j] > SCORE_THRESH, in row order (:836-841)
    int n = 0;
    for (int b0 = 0; b0 < R; b0 += kWG) {
      const int i = b0 + tid;
      const float s = i < R ? a.scores[(size_t)(r0 + i) * a.nc + j] : 0.f;
      const int f = (i < R && s > a.score_thresh) ? 1 : 0;
      const int incl = block_scan_inclusive(f, L.scan_tmp);
      if (f) {
        const int pos = n + incl - 1;
        const float* bx = a.boxes + (size_t)(r0 + i) * 6 * a.nc + 6 * j;
#pragma unroll
        for (int c = 0; c < 6; ++c) dets[7 * (size_t)pos + c] = bx[c];
        dets[7 * (size_t)pos + 6] = s;
        src[pos] = i;
      }
      n += L.scan_tmp[kWG - 1];
      __syncthreads();
    }
    const int nk = wg_nms(dets, n, a.nms_thresh, 0, 0, sc, L);         // :851
    __syncthreads();
    float* out = out0 + (size_t)j * a.cap * 7;
    for (int i = tid; i < nk; i += kWG) {
      const int64_t k = sc.keep[i];
#pragma unroll
      for (int c = 0; c < 7; ++c) out[7 * (size_t)i + c] = dets[7 * (size_t)k + c];
      if (keep0) keep0[(size_t)j * a.cap + i] = a.keep_idx ? a.keep_idx[r0 + src[k]] : (int64_t)src[k];   // :840,854
    }
    if (tid == 0) a.counts[b * a.nc + j] = nk;
    total += nk;
    __syncthreads();
  }
  // DETECTIONS_PER_IM cap (:869-878): image_thresh = np.sort(all scores)[-cap]; keep scores >= image_thresh per class, in order
  if (a.det_per_im > 0 && total > a.det_per_im) {
    if (tid == 0) s_found = 0;
    __syncthreads();
    for (int j = 1; j < a.nc && !s_found; ++j) {
      const int nj = a.counts[b * a.nc + j];
      for (int i = tid; i < nj; i += kWG) {
        const float s = out0[((size_t)j * a.cap + i) * 7 + 6];
        int gt = 0, ge = 0;
        for (int jj = 1; jj < a.nc; ++jj) {
          const int njj = a.counts[b * a.nc + jj];
          for (int q = 0; q < njj; ++q) {
            const float o = out0[((size_t)jj * a.cap + q) * 7 + 6];
            gt += o > s ? 1 : 0; ge += o >= s ? 1 : 0;
          }
        }
        if (gt < a.det_per_im && a.det_per_im <= ge) { s_thresh = s; s_found = 1; }     // every such s has the same value
      }
      __syncthreads();
    }
    const float th = s_thresh;
    for (int j = 1; j < a.nc; ++j) {
      const int nj = a.counts[b * a.nc + j];
      float* out = out0 + (size_t)j * a.cap * 7;
      int kept = 0;
      for (int b0 = 0; b0 < nj; b0 += kWG) {
        const int i = b0 + tid;
        float row[7]; int64_t kk = 0;
        const int f = (i < nj && out[7 * (size_t)i + 6] >= th) ? 1 : 0;
        if (f) {
#pragma unroll
          for (int c = 0; c < 7; ++c) row[c] = out[7 * (size_t)i + c];
          if (keep0) kk = keep0[(size_t)j * a.cap + i];
        }
        const int incl = block_scan_inclusive(f, L.scan_tmp);       // barriers inside: every read above precedes every write below
        if (f) {
          const int pos = kept + incl - 1;
#pragma unroll
          for (int c = 0; c < 7; ++c) out[7 * (size_t)pos + c] = row[c];
          if (keep0) keep0[(size_t)j * a.cap + pos] = kk;
        }
        kept += L.scan_tmp[kWG - 1];
        __syncthreads();
      }
      if (tid == 0) a.counts[b * a.nc + j] = kept;
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------ batched NMS + pack
struct NmsPackArgs {
  const float* dets; const int32_t* counts;          // [B, in_cap, 7] (item stride in_stride floats), counts[b * count_stride]
  float* out; int64_t* keep; int32_t* num;           // packed [B, out_cap + 1, 7] (or NULL), keep [B, in_cap] (or NULL), num [B] (or NULL)
  char* ws; size_t ws_item;
  size_t in_stride; int count_stride, in_cap, out_cap, by_volume;
  float thresh;
};

__global__ __launch_bounds__(kWG) void nms_pack_kernel(NmsPackArgs a) {
  __shared__ WgLds L;
  const int tid = threadIdx.x, b = blockIdx.x;
  const float* dets = a.dets + (size_t)b * a.in_stride;
  int n = a.counts ? a.counts[(size_t)b * a.count_stride] : a.in_cap;
  if (n > a.in_cap) n = a.in_cap;
  const NmsScratch sc = nms_scratch_carve(a.ws + (size_t)b * a.ws_item);
  const int nk = wg_nms(dets, n, a.thresh, a.by_volume, 0, sc, L);
  __syncthreads();
  if (a.keep)
    for (int i = tid; i < nk; i += kWG) a.keep[(size_t)b * a.in_cap + i] = sc.keep[i];
  if (a.num && tid == 0) a.num[b] = nk;
  if (a.out) {
    float* out = a.out + (size_t)b * (a.out_cap + 1) * 7;
    const int m = min(nk, a.out_cap);
    for (int e = tid; e < (a.out_cap + 1) * 7; e += kWG) {
      const int i = e / 7, c = e % 7;
      float v = 0.f;
      if (i < m) v = dets[7 * (size_t)sc.keep[i] + c];
      else if (i == a.out_cap && c == 0) v = (float)m;            // trailer row: the count (m3d.shard format)
      out[e] = v;
    }
  }
}

}  // namespace

M3D_API int m3d_fused_max_boxes(void) { return kFusedMax; }

M3D_API size_t m3d_generate_proposals3d_batched_workspace_bytes(int batch, int A, int S, int H, int W, int pre_nms_topN) {
  long long total = (long long)A * S * H * W;
  long long K = (pre_nms_topN <= 0 || pre_nms_topN >= total) ? total : pre_nms_topN;
  if (K <= 0) K = 1;
  if (K > kFusedMax) K = kFusedMax;
  return (size_t)(batch > 0 ? batch : 1) * prop_item_bytes((int)K) + 256;
}

M3D_API int m3d_generate_proposals3d_batched(const float* d_scores, const float* d_deltas, int batch, int A, int S, int H, int W,
                                             const double* anchors, double feat_stride, const double* im_info, int pre_nms_topN,
                                             int post_nms_topN, float nms_thresh, double min_size, double xform_clip,
                                             int first_batch_index, int out_rows, float* d_rois, float* d_probs,
                                             int64_t* d_keep_idx, int32_t* d_num, void* d_ws, size_t ws_bytes, void* stream) {
  if (batch <= 0 || batch > 65535 || A <= 0 || A > 64 || S <= 0 || H <= 0 || W <= 0 || !anchors || !im_info || out_rows <= 0)
    return M3D_EINVAL;
  if (!d_scores || !d_deltas || !d_rois || !d_probs || !d_keep_idx || !d_num || !d_ws) return M3D_EINVAL;
  const long long total = (long long)A * S * H * W;
  if (total >= 0xFFFFFFFFll) return M3D_EUNSUPPORTED;
  const long long Kll = (pre_nms_topN <= 0 || pre_nms_topN >= total) ? total : pre_nms_topN;   // :135
  if (Kll > kFusedMax) return M3D_EUNSUPPORTED;        // larger pre-NMS sets: the multi-launch m3d_generate_proposals3d
  const int K = (int)Kll;
  if (ws_bytes < (size_t)batch * prop_item_bytes(K) + 256) return M3D_EWORKSPACE;
  PropFusedArgs a;
  a.scores = d_scores; a.deltas = d_deltas; a.rois = d_rois; a.probs = d_probs; a.keep_idx = d_keep_idx; a.num = d_num;
  a.ws = (char*)m3d::align_up((size_t)d_ws, 256); a.ws_item = prop_item_bytes(K);
  a.K = K; a.post = post_nms_topN; a.cap_out = out_rows; a.nms_thresh = nms_thresh; a.first_batch_index = first_batch_index;
  PropParams& p = a.p;
  for (int i = 0; i < 6 * A; ++i) p.anchors[i] = anchors[i];
  p.stride = feat_stride; p.im_s = im_info[0]; p.im_h = im_info[1]; p.im_w = im_info[2]; p.im_scale = im_info[3];
  p.min_size = min_size; p.A = A; p.S = S; p.H = H; p.W = W; p.batch_index = first_batch_index;
  for (int i = 0; i < 6; ++i) p.xf.w[i] = 1.0;                            // :149-150
  p.xf.clip = xform_clip; p.xf.cs = im_info[0]; p.xf.ch = im_info[1]; p.xf.cw = im_info[2];   // :154
  hipLaunchKernelGGL(proposals_fused_kernel, dim3(batch), dim3(kWG), 0, m3d::as_stream(stream), a);
  return m3d::check_launch("generate_proposals3d_batched");
}

M3D_API size_t m3d_box_results3d_batched_workspace_bytes(int batch) {
  return (size_t)(batch > 0 ? batch : 1) * boxres_item_bytes() + 256;
}

M3D_API int m3d_box_results3d_batched(const float* d_scores, const float* d_boxes, const int64_t* d_keep_idx,
                                      const int32_t* d_offsets, int batch, int num_classes, float score_thresh, float nms_thresh,
                                      int detections_per_im, int max_rows_per_item, float* d_cls_boxes, int64_t* d_cls_keep,
                                      int32_t* d_counts, void* d_ws, size_t ws_bytes, void* stream) {
  if (batch <= 0 || batch > 65535 || num_classes < 1 || max_rows_per_item <= 0) return M3D_EINVAL;
  if (!d_scores || !d_boxes || !d_offsets || !d_cls_boxes || !d_counts || !d_ws) return M3D_EINVAL;
  if (max_rows_per_item > kFusedMax) return M3D_EUNSUPPORTED;
  if (ws_bytes < (size_t)batch * boxres_item_bytes() + 256) return M3D_EWORKSPACE;
  BoxResArgs a;
  a.scores = d_scores; a.boxes = d_boxes; a.keep_idx = d_keep_idx; a.offsets = d_offsets; a.cls_boxes = d_cls_boxes;
  a.cls_keep = d_cls_keep; a.counts = d_counts; a.ws = (char*)m3d::align_up((size_t)d_ws, 256); a.ws_item = boxres_item_bytes();
  a.nc = num_classes; a.cap = max_rows_per_item; a.det_per_im = detections_per_im; a.score_thresh = score_thresh;
  a.nms_thresh = nms_thresh;
  hipLaunchKernelGGL(box_results_fused_kernel, dim3(batch), dim3(kWG), 0, m3d::as_stream(stream), a);
  return m3d::check_launch("box_results3d_batched");
}

M3D_API size_t m3d_nms3d_batched_workspace_bytes(int batch) {
  return (size_t)(batch > 0 ? batch : 1) * (nms_scratch_bytes() + 256) + 256;
}

M3D_API int m3d_nms3d_batched(const float* d_dets, size_t item_stride_floats, const int32_t* d_counts, int count_stride, int batch,
                              int max_boxes, float thresh, int by_volume, int out_cap, float* d_packed, int64_t* d_keep,
                              int32_t* d_num_keep, void* d_ws, size_t ws_bytes, void* stream) {
  if (batch <= 0 || batch > 65535 || max_boxes < 0 || out_cap < 0) return M3D_EINVAL;
  if (!d_dets || !d_ws || (!d_packed && !d_keep && !d_num_keep)) return M3D_EINVAL;
  if (max_boxes > kFusedMax) return M3D_EUNSUPPORTED;
  if (ws_bytes < m3d_nms3d_batched_workspace_bytes(batch)) return M3D_EWORKSPACE;
  NmsPackArgs a;
  a.dets = d_dets; a.counts = d_counts; a.out = d_packed; a.keep = d_keep; a.num = d_num_keep;
  a.ws = (char*)m3d::align_up((size_t)d_ws, 256); a.ws_item = nms_scratch_bytes() + 256;
  a.in_stride = item_stride_floats; a.count_stride = count_stride; a.in_cap = max_boxes; a.out_cap = out_cap;
  a.by_volume = by_volume; a.thresh = thresh;
  hipLaunchKernelGGL(nms_pack_kernel, dim3(batch), dim3(kWG), 0, m3d::as_stream(stream), a);
  return m3d::check_launch("nms3d_batched");
}
