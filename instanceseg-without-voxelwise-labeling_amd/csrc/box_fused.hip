// Batched, fused box post-processing for gfx950: a whole batch of tiles per launch, no host round trips.
//
//   proposals     GenerateProposalsOp_3d.forward per tile (lib/modeling/generate_proposals_3d.py:19-192):
//                 top-N radix select -> sort -> decode/clip/filter -> nms_3d -> rois / probs / kept flat indices
//   box results   box_results_with_nms_and_limit per tile (lib/core/test.py:806-883): score threshold, per-class nms_3d,
//                 DETECTIONS_PER_IM cap, kept anchor indices carried through
//   nms + pack    nms_3d / nms_3d_volume per item (the cross-tile NMS of lib/core/test.py:159 when every volume is one tile)
//                 writing the padded [cap+1,7] block m3d.shard all-gathers
//
// These stages are latency-bound integer / index work on <= 2048 boxes per tile ("report microseconds, not a roofline
// fraction").  The per-tile multi-launch forms in box_ops.hip cost 26 + 8 + 4 launches and three host read-backs PER TILE.
// Here every stage is three launches for the WHOLE batch:
//   stage 1  one 1024-thread workgroup per tile: everything up to the score-sorted box list (all phases back to back in LDS);
//   stage 2  the suppression bitmask, the one embarrassingly parallel O(n^2) part: 64x64 tiles of all items spread over the chip
//            (one CU issues a wave instruction every ~8 cycles per wave at 16 waves - 500 k IoUs on one CU cost 150 us);
//   stage 3  one workgroup per tile: greedy resolve over 64-row chunks, compaction, gather into the stage's outputs.
// Same fp32 operation order, tie rules and index semantics as box_ops.hip (shared helpers in box_common.h; -ffp-contract=off).
#include "box_common.h"

// Diagnostic builds only (-DM3D_BOX_STAMPS, `make stamps`): s_memtime per phase of item 0 into a buffer of its own
// (tools/bench_box.py); never compiled into libm3d.so.
#ifdef M3D_BOX_STAMPS
__device__ unsigned long long g_box_stamps[64];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_box_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int m3d_debug_read_stamps(unsigned long long* host64) {
  return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_box_stamps), sizeof(unsigned long long) * 64);
}
#else
#define STAMP(i) do { } while (0)
#endif

namespace {
using namespace m3dbox;
typedef unsigned long long u64;

constexpr int kWG = 1024;
constexpr int kFusedMax = 2048;                  // boxes per item
constexpr int kNblkMax = kFusedMax / 64;
constexpr int kCandMax = 4096;                   // threshold-bucket keys kept in LDS by the radix select

struct NmsHdr { int n; int nk; int pad[62]; };   // n: boxes handed from stage 1 to stages 2 / 3

struct NmsScratch {                               // global scratch of ONE item
  NmsHdr* hdr; int* order; SBox* sboxes; u64* mask; unsigned char* flag; int64_t* keep;
};

__host__ __device__ inline size_t nms_scratch_bytes() {
  return m3d::align_up(sizeof(NmsHdr), 256) + m3d::align_up(sizeof(int) * kFusedMax, 256) +
         m3d::align_up(sizeof(SBox) * kFusedMax, 256) + m3d::align_up(sizeof(u64) * kFusedMax * kNblkMax, 256) +
         m3d::align_up((size_t)kFusedMax, 256) + m3d::align_up(sizeof(int64_t) * kFusedMax, 256);
}

__host__ __device__ inline NmsScratch nms_scratch_carve(char* p) {
  NmsScratch s;
  s.hdr = (NmsHdr*)p; p += m3d::align_up(sizeof(NmsHdr), 256);
  s.order = (int*)p; p += m3d::align_up(sizeof(int) * kFusedMax, 256);
  s.sboxes = (SBox*)p; p += m3d::align_up(sizeof(SBox) * kFusedMax, 256);
  s.mask = (u64*)p; p += m3d::align_up(sizeof(u64) * kFusedMax * kNblkMax, 256);
  s.flag = (unsigned char*)p; p += m3d::align_up((size_t)kFusedMax, 256);
  s.keep = (int64_t*)p;
  return s;
}

struct WgLds {                                    // static LDS of a stage-1 / stage-3 workgroup; phases reuse it
  unsigned int hist[256];                         // radix select
  union {
    u64 chunk[64 * kNblkMax];                     // NMS resolve: 64 mask rows
    u64 skeys[kFusedMax];                         // sort keys (proposal order, NMS order)
    u64 cand[kCandMax];                           // radix select: keys inside the threshold bucket
  };
  u64 removed[kNblkMax];
  u64 kept_word;
  u64 prefix;
  unsigned int remaining, done, count, bucket, ncand;
  int scan_tmp[kWG];
};

__device__ inline float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ inline u64 readlane_u64(u64 v, int l) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v & 0xFFFFFFFFull), l);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}

// hist[bin] += 1 from every calling lane.  The keys of one radix pass crowd into a few bins (pass 0 of sigmoid scores: the
// exponent byte) and 64 lanes adding to one LDS address serialise: the lanes that share the first lane's bin are counted with
// one ballot and added once, everyone else adds for itself.  (Measured on 143 k scores, one workgroup: 34 us per sweep this way;
// three rounds of aggregation cost more in instruction issue than they save in LDS time: 60 us.)  May be called under divergence.
__device__ inline void hist_add(unsigned int* hist, unsigned int bin) {
  const unsigned int lb = (unsigned int)__builtin_amdgcn_readfirstlane((int)bin);
  const u64 same = __ballot(bin == lb);
  if (bin == lb) {
    if (__ffsll((long long)same) - 1 == (int)(threadIdx.x & 63)) atomicAdd(&hist[lb], (unsigned int)__popcll(same));
  } else {
    atomicAdd(&hist[bin], 1u);
  }
}

// inclusive block scan of one 0/1 flag per thread: ballot + popcount inside the wave, 16 wave totals through LDS (two
// barriers).  Returns this thread's inclusive count; tmp[kWG - 1] holds the block total until the caller's next barrier.
__device__ inline int block_scan_inclusive(int f, int* tmp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 m = __ballot(f != 0);
  const int incl = __popcll(m & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull)));
  if (lane == 0) tmp[wave] = __popcll(m);
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kWG / 64; ++w) { const int t = tmp[w]; total += t; before += w < wave ? t : 0; }
  __syncthreads();
  if (threadIdx.x == kWG - 1) tmp[kWG - 1] = total;
  __syncthreads();
  return before + incl;
}

// descending bitonic sort of s[0..npad) (npad a power of two <= 2048) by the whole workgroup
// Thread t owns elements t and t + 1024, so for j < 64 both partners of a compare-exchange belong to the same wave: LDS
// operations of one wave execute in order, and those stages need no workgroup barrier (10 instead of 55 barriers at 1024 keys).
__device__ inline void wg_sort_desc(u64* s, int npad) {
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += kWG) {
        const int p = i ^ j;
        if (p > i) {
          const u64 a = s[i], b = s[p];
          const bool desc = (i & k) == 0;
          if ((a < b) == desc) { s[i] = b; s[p] = a; }
        }
      }
      if (j >= 64 || j == 1) __syncthreads();                      // j == 1 ends a merge round: the next round may start >= 64
      else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // ordering inside the wave only
    }
}

__device__ inline int pow2_at_least(int n) { int p = 2; while (p < n) p <<= 1; return p; }

// NMS stage 1 tail: visiting order of dets[0..n) - descending key, ties in descending index (box_ops.hip key_before;
// cython_nms_3d.pyx:49 / :112 under the build's tie rule) - and the boxes in that order.
__device__ void wg_nms_prepare(const float* __restrict__ dets, int n, int by_volume, const NmsScratch& sc, WgLds& L) {
  const int tid = threadIdx.x;
  if (tid == 0) { sc.hdr->n = n; sc.hdr->nk = 0; }
  if (n <= 0) return;
  const int npad = pow2_at_least(n);
  for (int i = tid; i < npad; i += kWG) {
    u64 k = 0ull;
    if (i < n) {
      float key = by_volume ? det_volume(dets + 7 * (size_t)i) : dets[7 * (size_t)i + 6];
      key = key + 0.0f;                                            // -0 -> +0: equal floats must have equal bit patterns
      k = ((u64)score_bits(key) << 32) | (unsigned int)i;
    }
    L.skeys[i] = k;
  }
  __syncthreads();
  wg_sort_desc(L.skeys, npad);
  for (int r = tid; r < n; r += kWG) {
    const int i = (int)(unsigned int)(L.skeys[r] & 0xFFFFFFFFull);
    const float* d = dets + 7 * (size_t)i;
    sc.order[r] = i;
    sc.sboxes[r] = SBox{d[0], d[1], d[2], d[3], d[4], d[5], det_volume(d), 0.f};
  }
  __syncthreads();
}

// The same for dets already sorted by descending score (the proposal list: score desc, flat index asc): the visiting order
// differs only inside runs of equal scores, which the tie rule walks from the highest index down - reverse each run.
__device__ void wg_nms_prepare_sorted(const float* __restrict__ dets, int n, const NmsScratch& sc) {
  const int tid = threadIdx.x;
  if (tid == 0) { sc.hdr->n = n; sc.hdr->nk = 0; }
  for (int i = tid; i < n; i += kWG) {
    const float k = dets[7 * (size_t)i + 6];
    int lo = 0, hi = i;                                  // s0 = first index whose score is not above k (scores descend)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (dets[7 * (size_t)mid + 6] > k) lo = mid + 1; else hi = mid; }
    const int s0 = lo;
    lo = i; hi = n - 1;                                  // e0 = last index whose score is not below k
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (dets[7 * (size_t)mid + 6] < k) hi = mid - 1; else lo = mid; }
    const int e0 = lo;
    const int r = s0 + (e0 - i);
    const float* d = dets + 7 * (size_t)i;
    sc.order[r] = i;
    sc.sboxes[r] = SBox{d[0], d[1], d[2], d[3], d[4], d[5], det_volume(d), 0.f};
  }
  __syncthreads();
}

// NMS stage 3: greedy resolve over the bitmask in 64-row chunks, then the kept input indices in ascending order
// (np.where(suppressed == 0)[0], pyx:96) into sc.keep (at most keep_limit if > 0).  Returns their number.
__device__ int wg_nms_resolve(int n, int keep_limit, const NmsScratch& sc, WgLds& L) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (n <= 0) return 0;
  const int nblk = (n + 63) / 64;
  for (int w = tid; w < nblk; w += kWG) L.removed[w] = 0ull;
  // rows of chunk c, words [c, nblk): <= 2 per thread; the next chunk's words are fetched while this one is resolved
  u64 pre[2];
  auto fetch = [&](int c) __attribute__((always_inline)) {
    const int rows = min(64, n - c * 64), wcount = nblk - c;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * kWG;
      pre[u] = (c < nblk && e < rows * wcount) ? sc.mask[(size_t)(c * 64 + e / wcount) * kNblkMax + c + e % wcount] : 0ull;
    }
  };
  fetch(0);
  for (int c = 0; c < nblk; ++c) {
    const int rows = min(64, n - c * 64), wcount = nblk - c;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * kWG;
      if (e < rows * wcount) L.chunk[(e / wcount) * kNblkMax + c + e % wcount] = pre[u];
    }
    __syncthreads();
    fetch(c + 1);
    if (wave == 0) {                                   // the diagonal tile: 64 dependent steps, operands in lanes not in LDS
      const u64 diag = lane < rows ? L.chunk[lane * kNblkMax + c] : 0ull;
      const u64 rowmask = rows == 64 ? ~0ull : ((1ull << rows) - 1ull);
      u64 rem = L.removed[c], kept = 0ull;
      u64 todo = ~rem & rowmask;                       // rows not suppressed so far: visit only those, lowest first
      while (todo) {
        const int b = __ffsll((long long)todo) - 1;
        kept |= 1ull << b;
        rem |= readlane_u64(diag, b);
        todo &= ~rem & ~((2ull << b) - 1ull);
      }
      if (lane == 0) { L.removed[c] = rem; L.kept_word = kept; }
    }
    __syncthreads();
    const u64 kept = L.kept_word;
    if ((kept >> lane) & 1ull)                          // lane = kept row, wave = word: every later word ORs its kept rows in
      for (int w = c + 1 + wave; w < nblk; w += kWG / 64) atomicOr(&L.removed[w], L.chunk[lane * kNblkMax + w]);
    if (tid < rows) sc.flag[sc.order[c * 64 + tid]] = (unsigned char)((kept >> tid) & 1ull);
    __syncthreads();
  }
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

// ------------------------------------------------------------------------------------------------ stage 2 (shared)
// grid (column block, row block, item), one wave per 64x64 tile of the upper triangle: bit q of mask[i][cb] <=> sorted box
// cb*64+q is suppressed by sorted box i (cython_nms_3d.pyx:82-93).  Lane q holds column box q; rows read it by v_readlane.
struct MaskArgs { char* ws; size_t ws_item; size_t nms_offset; float thresh; };

__global__ __launch_bounds__(64) void nms_mask_tiles_kernel(MaskArgs a) {
  const NmsScratch sc = nms_scratch_carve(a.ws + (size_t)blockIdx.z * a.ws_item + a.nms_offset);
  const int n = sc.hdr->n;
  const int cb = blockIdx.x, rb = blockIdx.y, lane = threadIdx.x;
  if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
  const int i = rb * 64 + lane;
  const SBox bi = sc.sboxes[min(i, n - 1)];
  const SBox bc = sc.sboxes[min(cb * 64 + lane, n - 1)];
  u64 bits = 0ull;
  const int m = min(64, n - cb * 64);
  for (int q = 0; q < m; ++q) {
    SBox bj;
    bj.x1 = readlane_f(bc.x1, q); bj.y1 = readlane_f(bc.y1, q); bj.z1 = readlane_f(bc.z1, q);
    bj.x2 = readlane_f(bc.x2, q); bj.y2 = readlane_f(bc.y2, q); bj.z2 = readlane_f(bc.z2, q);
    bj.vol = readlane_f(bc.vol, q);
    if (cb * 64 + q > i && nms_suppresses(bi, bj, a.thresh)) bits |= 1ull << q;
  }
  if (i < n) sc.mask[(size_t)i * kNblkMax + cb] = bits;
}

// ------------------------------------------------------------------------------------------------ proposals
struct PropFusedArgs {
  const float* scores; const float* deltas;          // [B,A,S,H,W], [B,6A,S,H,W]
  float* rois; float* probs; int64_t* keep_idx; int32_t* num;   // [B,rows,7], [B,rows], [B,rows], [B]
  char* ws; size_t ws_item;                          // per-item scratch
  int K, post, cap_out;                              // pre_nms_topN (clamped), post_nms_topN, rows per item in the outputs
  float nms_thresh;
  int first_batch_index;
  PropParams p;
};

struct PropScratch { u64* keys; float* boxes; unsigned char* valid; float* dets; int64_t* flat_idx; char* nms; };

// Multi-workgroup radix select (round 2): the four score-digit histograms, the two list counters and the list of the keys inside
// the threshold bucket live behind the per-item scratch.
struct SelCtrl { unsigned int count, ncand, pad[62]; };
struct SelScratch { unsigned int* hist /*[4][256]*/; SelCtrl* ctrl; u64* cand /*[total]*/; };

__host__ __device__ inline size_t prop_nms_offset(int K) {
  return m3d::align_up(sizeof(u64) * K, 256) + m3d::align_up(sizeof(float) * 6 * K, 256) + m3d::align_up((size_t)K, 256) +
         m3d::align_up(sizeof(float) * 7 * K, 256) + m3d::align_up(sizeof(int64_t) * K, 256);
}
__host__ __device__ inline size_t prop_sel_offset(int K) { return prop_nms_offset(K) + nms_scratch_bytes() + 256; }
__host__ __device__ inline size_t prop_item_bytes(int K, long long total) {
  return prop_sel_offset(K) + 4096 + sizeof(SelCtrl) + m3d::align_up(sizeof(u64) * (size_t)total, 256) + 256;
}
__device__ inline SelScratch sel_carve(char* item, int K) {
  SelScratch s;
  char* w = item + prop_sel_offset(K);
  s.hist = (unsigned int*)w; w += 4096;
  s.ctrl = (SelCtrl*)w; w += sizeof(SelCtrl);
  s.cand = (u64*)w;
  return s;
}

__device__ inline PropScratch prop_carve(char* w, int K) {
  PropScratch s;
  s.keys = (u64*)w; w += m3d::align_up(sizeof(u64) * K, 256);
  s.boxes = (float*)w; w += m3d::align_up(sizeof(float) * 6 * K, 256);
  s.valid = (unsigned char*)w; w += m3d::align_up((size_t)K, 256);
  s.dets = (float*)w; w += m3d::align_up(sizeof(float) * 7 * K, 256);
  s.flat_idx = (int64_t*)w; w += m3d::align_up(sizeof(int64_t) * K, 256);
  s.nms = w;
  return s;
}

// ---- state of the select after the score digits whose histograms exist.  Same arithmetic as the single-workgroup `pick`:
// at a level, tot = keys under the prefix; tot <= remaining: all of them are selected (all_sel); else the digit d with
// count(bins > d) < remaining <= count(bins >= d) joins the prefix, remaining -= count(bins > d), bucket = hist[d].
// The sweeps stop after level p when all_sel, or p >= 1 and the bucket fits the LDS candidate list, or p == 3 (score resolved).
struct SelState { unsigned int prefix32; unsigned int remaining, bucket; int nd /* digits in the prefix */; int all_sel, stop; };

// wave 0 of the calling workgroup; nlev histograms are complete.  Result through `out` (LDS); caller barriers.
__device__ inline void sel_replay(const unsigned int* __restrict__ hist, int nlev, int K, SelState* out) {
  const int lane = threadIdx.x & 63;
  unsigned int prefix = 0u, rem = (unsigned int)K, bucket = 0u;
  int nd = 0, all_sel = 0, stop = 0;
  for (int p = 0; p < nlev && !stop; ++p) {
    unsigned int h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = hist[p * 256 + 4 * lane + q];
    const unsigned int mine = (h[0] + h[1]) + (h[2] + h[3]);
    unsigned int suf = mine;                                         // sum over lanes >= lane (bins >= 4 * lane)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned int v = (unsigned int)__shfl_down((int)suf, off, 64);
      if (lane + off < 64) suf += v;
    }
    const unsigned int tot = (unsigned int)__shfl((int)suf, 0, 64);
    if (tot <= rem) { all_sel = 1; stop = 1; break; }
    const unsigned int above = suf - mine;
    const bool here = above < rem && rem <= above + mine;            // exactly one lane
    unsigned int d = 0u, nb = 0u, nr = 0u;
    if (here) {
      unsigned int r = rem - above;
#pragma unroll
      for (int q = 3; q >= 0; --q) {
        if (r != 0u) {
          if (h[q] >= r) { d = (unsigned int)(4 * lane + q); nb = h[q]; nr = r; r = 0u; }
          else r -= h[q];
        }
      }
    }
    const int src = __ffsll((long long)__ballot(here)) - 1;
    d = (unsigned int)__shfl((int)d, src, 64); nb = (unsigned int)__shfl((int)nb, src, 64); nr = (unsigned int)__shfl((int)nr, src, 64);
    prefix |= d << (24 - 8 * p); rem = nr; bucket = nb; nd = p + 1;
    if ((p >= 1 && bucket <= (unsigned int)kCandMax) || p == 3) stop = 1;
  }
  if (lane == 0) { out->prefix32 = prefix; out->remaining = rem; out->bucket = bucket; out->nd = nd; out->all_sel = all_sel; out->stop = stop; }
}

constexpr int kSelT = 256;

__global__ void prop_sel_init_kernel(PropFusedArgs a) {
  const SelScratch ss = sel_carve(a.ws + (size_t)blockIdx.x * a.ws_item, a.K);
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) ss.hist[i] = 0u;
  if (threadIdx.x == 0) { ss.ctrl->count = 0u; ss.ctrl->ncand = 0u; }
}

// histogram of score digit `level` over the keys under the prefix of the levels before it; grid (chunks, items)
__global__ __launch_bounds__(kSelT) void prop_sel_hist_kernel(PropFusedArgs a, int level) {
  __shared__ unsigned int lh[256];
  __shared__ SelState st;
  const int tid = threadIdx.x, b = blockIdx.y;
  const long long total = (long long)a.p.A * a.p.S * a.p.H * a.p.W;
  const float* scores = a.scores + (size_t)b * total;
  const SelScratch ss = sel_carve(a.ws + (size_t)b * a.ws_item, a.K);
  lh[tid] = 0u;
  if (tid < 64) sel_replay(ss.hist, level, a.K, &st);
  __syncthreads();
  if (st.stop) return;
  const unsigned int pfx = st.prefix32;
  const int sh_digit = 24 - 8 * level, sh_top = 32 - 8 * level;       // digit of this level; bits above it = the prefix digits
  for (long long e = (long long)blockIdx.x * kSelT + tid; e < total; e += (long long)gridDim.x * kSelT) {
    const unsigned int sb = score_bits(scores[e]);
    if (level == 0 || (sb >> sh_top) == (pfx >> sh_top)) hist_add(lh, (sb >> sh_digit) & 255u);
  }
  __syncthreads();
  const unsigned int v = lh[tid];
  if (v) atomicAdd(&ss.hist[level * 256 + tid], v);
}

// keys above the threshold bucket -> the selected list, keys inside it -> the candidate list; grid (chunks, items)
__global__ __launch_bounds__(kSelT) void prop_sel_compact_kernel(PropFusedArgs a) {
  __shared__ SelState st;
  const int tid = threadIdx.x, b = blockIdx.y;
  const int A = a.p.A, SHW = a.p.S * a.p.H * a.p.W, K = a.K;
  const long long total = (long long)A * SHW;
  const float* scores = a.scores + (size_t)b * total;
  char* item = a.ws + (size_t)b * a.ws_item;
  const PropScratch ps = prop_carve(item, K);
  const SelScratch ss = sel_carve(item, K);
  if (tid < 64) sel_replay(ss.hist, 4, K, &st);
  __syncthreads();
  const int sh = 32 - 8 * st.nd;
  const unsigned int pfx = st.nd ? (st.prefix32 >> sh) : 0u;
  const bool all_sel = st.all_sel != 0;
  for (long long e = (long long)blockIdx.x * kSelT + tid; e < total; e += (long long)gridDim.x * kSelT) {
    const unsigned int sb = score_bits(scores[e]);
    const unsigned int top = st.nd ? (sb >> sh) : 0u;
    if (top < pfx) continue;
    const int an = (int)(e / SHW), pos = (int)(e - (long long)an * SHW);
    const unsigned int flat = (unsigned int)pos * (unsigned int)A + (unsigned int)an;     // generate_proposals_3d.py:121,129
    const u64 key = ((u64)sb << 32) | (u64)(0xFFFFFFFFu - flat);
    if (top > pfx || all_sel) {
      const unsigned int slot = atomicAdd(&ss.ctrl->count, 1u);
      if (slot < (unsigned int)K) ps.keys[slot] = key;
    } else {
      const unsigned int slot = atomicAdd(&ss.ctrl->ncand, 1u);
      ss.cand[slot] = key;
    }
  }
}

__global__ __launch_bounds__(kWG) void proposals_stage1_kernel(PropFusedArgs a) {
  __shared__ WgLds L;
  const int tid = threadIdx.x, b = blockIdx.x;
  const PropParams& p = a.p;
  const int A = p.A, SHW = p.S * p.H * p.W, K = a.K;
  const long long total = (long long)A * SHW;
  const float* deltas = a.deltas + (size_t)b * total * 6;
  const PropScratch ps = prop_carve(a.ws + (size_t)b * a.ws_item, K);
  const NmsScratch sc = nms_scratch_carve(ps.nms);

  // ---- top-K of the 64-bit keys (score bits, ~flat index) by radix-256 select (generate_proposals_3d.py:135-146).  The sweeps
  // over the A*S*H*W scores ran before this kernel, on many workgroups (prop_sel_hist_kernel x 4, prop_sel_compact_kernel): the
  // keys above the threshold bucket are already in ps.keys, the bucket's own keys in ss.cand.  What is left is to take the
  // `remaining` largest keys of the bucket - in LDS when it fits (the usual case), else by more radix passes over the list.
  const SelScratch ss = sel_carve(a.ws + (size_t)b * a.ws_item, K);
  __shared__ SelState st;
  if (tid < 64) sel_replay(ss.hist, 4, K, &st);
  __syncthreads();
  const int ncand_g = st.all_sel ? 0 : (int)ss.ctrl->ncand;
  if (tid == 0) {
    L.prefix = (u64)st.prefix32 << 32; L.remaining = st.remaining; L.done = st.all_sel ? 1u : 0u;
    L.count = min(ss.ctrl->count, (unsigned int)K); L.bucket = st.bucket; L.ncand = 0u;
  }
  __syncthreads();
  STAMP(0);
  auto pick = [&](int shift) __attribute__((always_inline)) {       // wave 0: digit of the K-th key at this level
    const int lane = tid & 63;
    unsigned int h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = L.hist[4 * lane + q];
    const unsigned int mine = (h[0] + h[1]) + (h[2] + h[3]);
    unsigned int suf = mine;                                         // sum over lanes >= lane (bins >= 4 * lane)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned int v = (unsigned int)__shfl_down((int)suf, off, 64);
      if (lane + off < 64) suf += v;
    }
    const unsigned int tot = (unsigned int)__shfl((int)suf, 0, 64);
    const unsigned int rem = L.remaining;
    if (tot <= rem) {
      if (lane == 0) L.done = 1u;                                    // everything under the prefix is selected
    } else {
      const unsigned int above = suf - mine;                         // keys in higher bins than this lane's four
      if (above < rem && rem <= above + mine) {                      // exactly one lane
        unsigned int r = rem - above;
#pragma unroll
        for (int q = 3; q >= 0; --q) {
          if (r != 0u) {
            if (h[q] >= r) { L.prefix |= (u64)(4 * lane + q) << shift; L.bucket = h[q]; L.remaining = r; r = 0u; }
            else r -= h[q];
          }
        }
        if (shift == 0) L.done = 1u;
      }
    }
  };
  // visits every key of the threshold bucket (global list)
  auto sweep = [&](auto&& f) __attribute__((always_inline)) {
    for (int e = tid; e < ncand_g; e += kWG) f(ss.cand[e]);
  };
  bool in_lds = false;
  if (!st.all_sel && ncand_g <= kCandMax) {                           // the bucket fits: its keys move into LDS right away
    for (int e = tid; e < ncand_g; e += kWG) L.cand[e] = ss.cand[e];
    if (tid == 0) L.ncand = (unsigned int)ncand_g;
    in_lds = true;
    __syncthreads();
  }
  for (int pass = st.nd; pass < 8; ++pass) {                         // the digits below the st.nd resolved ones
    if (L.done) break;                                               // uniform: read after a barrier
    if (tid < 256) L.hist[tid] = 0u;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const u64 prefix = L.prefix;
    if (in_lds) {
      const int nc = (int)L.ncand;
      for (int e = tid; e < nc; e += kWG) {
        const u64 key = L.cand[e];
        if ((key >> (shift + 8)) == (prefix >> (shift + 8))) hist_add(L.hist, (unsigned int)(key >> shift) & 255u);
      }
    } else {                                                         // > kCandMax exact score ties at the threshold: list sweeps
      sweep([&](u64 key) {
        if ((key >> (shift + 8)) == (prefix >> (shift + 8))) hist_add(L.hist, (unsigned int)(key >> shift) & 255u);
      });
    }
    __syncthreads();
    if (tid < 64) pick(shift);
    __syncthreads();
    if (!in_lds && !L.done && L.bucket <= (unsigned)kCandMax) {
      const u64 pfx = L.prefix >> shift;
      sweep([&](u64 key) {
        const u64 top = key >> shift;
        if (top > pfx) {
          const unsigned int slot = atomicAdd(&L.count, 1u);
          if (slot < (unsigned)K) ps.keys[slot] = key;
        } else if (top == pfx) {
          const unsigned int slot = atomicAdd(&L.ncand, 1u);
          if (slot < (unsigned)kCandMax) L.cand[slot] = key;
        }
      });
      in_lds = true;
      __syncthreads();
    }
  }
  STAMP(1);
  // ---- the selected keys: key >= threshold (the prefix's lower bits are zero when the select stopped early)
  const u64 thr = L.prefix;
  if (in_lds) {
    const int nc = (int)L.ncand;
    for (int e = tid; e < nc; e += kWG) {
      const u64 key = L.cand[e];
      if (key >= thr) {
        const unsigned int slot = atomicAdd(&L.count, 1u);
        if (slot < (unsigned)K) ps.keys[slot] = key;
      }
    }
  } else if (!st.all_sel) {
    sweep([&](u64 key) {
      if (key >= thr) {
        const unsigned int slot = atomicAdd(&L.count, 1u);
        if (slot < (unsigned)K) ps.keys[slot] = key;
      }
    });
  }
  __syncthreads();
  STAMP(2);
  // ---- sort the (distinct) keys, descending: L.skeys[0..n)
  const int n = (int)min(L.count, (unsigned)K);
  const int npad = pow2_at_least(n);
  for (int i = tid; i < npad; i += kWG) L.skeys[i] = i < n ? ps.keys[i] : 0ull;
  __syncthreads();
  wg_sort_desc(L.skeys, npad);
  STAMP(3);
  // ---- decode + clip + _filter_boxes_3d for every candidate (:149-160,180-192)
  for (int i = tid; i < n; i += kWG) {
    const u64 key = L.skeys[i];
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
    for (int c = 0; c < 6; ++c) ps.boxes[6 * (size_t)i + c] = o[c];
    const double ms = p.min_size * p.im_scale;
    float ss = o[3] - o[0]; ss = ss + 1.0f;
    const float half = ss / 2.0f;
    const float xc = o[0] + half, yc = o[1] + half, zc = o[2] + half;
    ps.valid[i] = ((double)ss >= ms && (double)xc < p.im_w && (double)yc < p.im_h && (double)zc < p.im_s) ? 1 : 0;
  }
  __syncthreads();
  STAMP(4);
  // ---- ordered compaction of the valid candidates -> dets [nvalid,7] + flat index
  int nvalid = 0;
  for (int b0 = 0; b0 < n; b0 += kWG) {
    const int i = b0 + tid;
    const int f = i < n ? ps.valid[i] : 0;
    const int incl = block_scan_inclusive(f, L.scan_tmp);
    if (f) {
      const int pos = nvalid + incl - 1;
      const u64 key = L.skeys[i];
#pragma unroll
      for (int c = 0; c < 6; ++c) ps.dets[7 * (size_t)pos + c] = ps.boxes[6 * (size_t)i + c];
      ps.dets[7 * (size_t)pos + 6] = bits_score((unsigned int)(key >> 32));
      ps.flat_idx[pos] = (int64_t)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
    }
    nvalid += L.scan_tmp[kWG - 1];
    __syncthreads();
  }
  STAMP(5);
  wg_nms_prepare_sorted(ps.dets, nvalid, sc);                         // visiting order for nms_3d (:167-168)
  STAMP(6);
}

__global__ __launch_bounds__(kWG) void proposals_stage3_kernel(PropFusedArgs a) {
  __shared__ WgLds L;
  const int tid = threadIdx.x, b = blockIdx.x;
  const PropScratch ps = prop_carve(a.ws + (size_t)b * a.ws_item, a.K);
  const NmsScratch sc = nms_scratch_carve(ps.nms);
  const int nvalid = sc.hdr->n;
  STAMP(10);
  int nk;
  if (a.nms_thresh > 0) {                                             // keep = nms_3d(...)[:post_nms_topN] (:167-171)
    nk = wg_nms_resolve(nvalid, a.post, sc, L);
  } else {                                                            // post_nms_topN is applied only inside `if nms_thresh > 0`
    nk = nvalid;                                                      // (:167-171): without NMS every valid box is kept
    for (int i = tid; i < nk; i += kWG) sc.keep[i] = i;
  }
  __syncthreads();
  STAMP(11);
  if (nk > a.cap_out) nk = a.cap_out;
  float* rois = a.rois + (size_t)b * a.cap_out * 7;
  float* probs = a.probs + (size_t)b * a.cap_out;
  int64_t* kidx = a.keep_idx + (size_t)b * a.cap_out;
  for (int i = tid; i < nk; i += kWG) {                               // rois / probs / kept flat indices (:98-100,160,174-175)
    const int64_t k = sc.keep[i];
    rois[7 * (size_t)i] = (float)(a.first_batch_index + b);
#pragma unroll
    for (int c = 0; c < 6; ++c) rois[7 * (size_t)i + 1 + c] = ps.dets[7 * (size_t)k + c];
    probs[i] = ps.dets[7 * (size_t)k + 6];
    kidx[i] = ps.flat_idx[k];
  }
  if (tid == 0) a.num[b] = nk;
  STAMP(12);
}

// ------------------------------------------------------------------------------------------------ box results
struct BoxResArgs {
  const float* scores; const float* boxes; const int64_t* keep_idx;   // [R,nc], [R,6nc], [R] or NULL (rows of all items)
  const int32_t* offsets;                                             // [B+1] row ranges
  float* cls_boxes; int64_t* cls_keep; int32_t* counts;               // [B,nc,cap,7], [B,nc,cap] or NULL, [B,nc]
  char* ws; size_t ws_item;
  int nc, cap, det_per_im, cls;                                       // cls: the class this launch handles
  float score_thresh, nms_thresh;
};

__host__ __device__ inline size_t boxres_nms_offset() {
  return m3d::align_up(sizeof(float) * 7 * kFusedMax, 256) + m3d::align_up(sizeof(int) * kFusedMax, 256);
}
__host__ __device__ inline size_t boxres_item_bytes() { return boxres_nms_offset() + nms_scratch_bytes() + 256; }

__global__ __launch_bounds__(kWG) void box_results_stage1_kernel(BoxResArgs a) {
  __shared__ WgLds L;
  const int tid = threadIdx.x, b = blockIdx.x, j = a.cls;
  // an item with more rows than max_rows_per_item breaks the offsets contract (include/m3d.h): its surplus rows are ignored
  // rather than written past the scratch / the outputs sized for `cap` rows
  const int r0 = a.offsets[b], R = min(max(a.offsets[b + 1] - r0, 0), a.cap);
  char* w = a.ws + (size_t)b * a.ws_item;
  float* dets = (float*)w;
  int* src = (int*)(w + m3d::align_up(sizeof(float) * 7 * kFusedMax, 256));
  const NmsScratch sc = nms_scratch_carve(w + boxres_nms_offset());
  if (tid == 0 && j == 1) a.counts[b * a.nc] = 0;                      // class 0 = background: always empty (:833)
  int n = 0;                                                           // inds = scores[:, j] > SCORE_THRESH, in row order (:836-841)
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
  wg_nms_prepare(dets, n, 0, sc, L);                                   // :851
}

__global__ __launch_bounds__(kWG) void box_results_stage3_kernel(BoxResArgs a) {
  __shared__ WgLds L;
  __shared__ float s_thresh;
  __shared__ int s_found;
  const int tid = threadIdx.x, b = blockIdx.x, j = a.cls;
  const int r0 = a.offsets[b];
  char* w = a.ws + (size_t)b * a.ws_item;
  const float* dets = (const float*)w;
  const int* src = (const int*)(w + m3d::align_up(sizeof(float) * 7 * kFusedMax, 256));
  const NmsScratch sc = nms_scratch_carve(w + boxres_nms_offset());
  float* out0 = a.cls_boxes + (size_t)b * a.nc * a.cap * 7;
  int64_t* keep0 = a.cls_keep ? a.cls_keep + (size_t)b * a.nc * a.cap : nullptr;
  const int nk = wg_nms_resolve(sc.hdr->n, 0, sc, L);
  __syncthreads();
  float* out = out0 + (size_t)j * a.cap * 7;
  for (int i = tid; i < nk; i += kWG) {
    const int64_t k = sc.keep[i];
#pragma unroll
    for (int c = 0; c < 7; ++c) out[7 * (size_t)i + c] = dets[7 * (size_t)k + c];
    if (keep0) keep0[(size_t)j * a.cap + i] = a.keep_idx ? a.keep_idx[r0 + src[k]] : (int64_t)src[k];   // :840,854
  }
  if (tid == 0) a.counts[b * a.nc + j] = nk;
  __syncthreads();
  if (j != a.nc - 1 || a.det_per_im <= 0) return;
  // DETECTIONS_PER_IM cap once the last class is done (:869-878): image_thresh = np.sort(all scores)[-cap]; keep scores >=
  // image_thresh per class, in order
  int total = 0;
  for (int jj = 1; jj < a.nc; ++jj) total += a.counts[b * a.nc + jj];
  if (total <= a.det_per_im) return;
  if (tid == 0) s_found = 0;
  __syncthreads();
  for (int jj = 1; jj < a.nc && !s_found; ++jj) {
    const int nj = a.counts[b * a.nc + jj];
    for (int i = tid; i < nj; i += kWG) {
      const float s = out0[((size_t)jj * a.cap + i) * 7 + 6];
      int gt = 0, ge = 0;
      for (int j2 = 1; j2 < a.nc; ++j2) {
        const int n2 = a.counts[b * a.nc + j2];
        for (int q = 0; q < n2; ++q) {
          const float o = out0[((size_t)j2 * a.cap + q) * 7 + 6];
          gt += o > s ? 1 : 0; ge += o >= s ? 1 : 0;
        }
      }
      if (gt < a.det_per_im && a.det_per_im <= ge) { s_thresh = s; s_found = 1; }     // every such s has the same value
    }
    __syncthreads();
  }
  const float th = s_thresh;
  for (int jj = 1; jj < a.nc; ++jj) {
    const int nj = a.counts[b * a.nc + jj];
    float* oj = out0 + (size_t)jj * a.cap * 7;
    int kept = 0;
    for (int b0 = 0; b0 < nj; b0 += kWG) {
      const int i = b0 + tid;
      float row[7]; int64_t kk = 0;
      const int f = (i < nj && oj[7 * (size_t)i + 6] >= th) ? 1 : 0;
      if (f) {
#pragma unroll
        for (int c = 0; c < 7; ++c) row[c] = oj[7 * (size_t)i + c];
        if (keep0) kk = keep0[(size_t)jj * a.cap + i];
      }
      const int incl = block_scan_inclusive(f, L.scan_tmp);       // barriers inside: every read above precedes every write below
      if (f) {
        const int pos = kept + incl - 1;
#pragma unroll
        for (int c = 0; c < 7; ++c) oj[7 * (size_t)pos + c] = row[c];
        if (keep0) keep0[(size_t)jj * a.cap + pos] = kk;
      }
      kept += L.scan_tmp[kWG - 1];
      __syncthreads();
    }
    if (tid == 0) a.counts[b * a.nc + jj] = kept;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ batched NMS + pack
struct NmsPackArgs {
  const float* dets; const int32_t* counts;          // item b at dets + b*in_stride; counts[b * count_stride] (or NULL: in_cap)
  float* out; int64_t* keep; int32_t* num;           // packed [B, out_cap + 1, 7] (or NULL), keep [B, in_cap] (or NULL), num [B] (or NULL)
  char* ws; size_t ws_item;
  size_t in_stride; int count_stride, in_cap, out_cap, by_volume;
  float thresh;
};

__global__ __launch_bounds__(kWG) void nms_pack_stage1_kernel(NmsPackArgs a) {
  __shared__ WgLds L;
  const int b = blockIdx.x;
  int n = a.counts ? a.counts[(size_t)b * a.count_stride] : a.in_cap;
  if (n > a.in_cap) n = a.in_cap;
  if (n < 0) n = 0;
  wg_nms_prepare(a.dets + (size_t)b * a.in_stride, n, a.by_volume, nms_scratch_carve(a.ws + (size_t)b * a.ws_item), L);
}

__global__ __launch_bounds__(kWG) void nms_pack_stage3_kernel(NmsPackArgs a) {
  __shared__ WgLds L;
  const int tid = threadIdx.x, b = blockIdx.x;
  const float* dets = a.dets + (size_t)b * a.in_stride;
  const NmsScratch sc = nms_scratch_carve(a.ws + (size_t)b * a.ws_item);
  const int nk = wg_nms_resolve(sc.hdr->n, 0, sc, L);
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

void launch_mask(char* ws, size_t ws_item, size_t nms_offset, int batch, int max_boxes, float thresh, hipStream_t st) {
  int nblk = (max_boxes + 63) / 64;
  if (nblk < 1) nblk = 1;
  MaskArgs m{ws, ws_item, nms_offset, thresh};
  hipLaunchKernelGGL(nms_mask_tiles_kernel, dim3(nblk, nblk, batch), dim3(64), 0, st, m);
}

}  // namespace

M3D_API int m3d_fused_max_boxes(void) { return kFusedMax; }

M3D_API size_t m3d_generate_proposals3d_batched_workspace_bytes(int batch, int A, int S, int H, int W, int pre_nms_topN) {
  long long total = (long long)A * S * H * W;
  long long K = (pre_nms_topN <= 0 || pre_nms_topN >= total) ? total : pre_nms_topN;
  if (K <= 0) K = 1;
  if (K > kFusedMax) K = kFusedMax;
  return (size_t)(batch > 0 ? batch : 1) * prop_item_bytes((int)K, total) + 256;
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
  if (total >= 0x7FFFFFFFll) return M3D_EUNSUPPORTED;
  const long long Kll = (pre_nms_topN <= 0 || pre_nms_topN >= total) ? total : pre_nms_topN;   // :135
  if (Kll > kFusedMax) return M3D_EUNSUPPORTED;        // larger pre-NMS sets: the multi-launch m3d_generate_proposals3d
  const int K = (int)Kll;
  // without NMS every valid box of the pre-NMS set is a proposal (generate_proposals_3d.py:167-171: post_nms_topN only cuts behind the
  // NMS): a caller that sized its outputs by post_nms_topN would get a silently truncated list - refuse instead
  if (nms_thresh <= 0 && out_rows < K) return M3D_EINVAL;
  if (ws_bytes < (size_t)batch * prop_item_bytes(K, (long long)A * S * H * W) + 256) return M3D_EWORKSPACE;
  PropFusedArgs a;
  a.scores = d_scores; a.deltas = d_deltas; a.rois = d_rois; a.probs = d_probs; a.keep_idx = d_keep_idx; a.num = d_num;
  a.ws = (char*)m3d::align_up((size_t)d_ws, 256); a.ws_item = prop_item_bytes(K, (long long)A * S * H * W);
  a.K = K; a.post = post_nms_topN; a.cap_out = out_rows; a.nms_thresh = nms_thresh; a.first_batch_index = first_batch_index;
  PropParams& p = a.p;
  for (int i = 0; i < 6 * A; ++i) p.anchors[i] = anchors[i];
  p.stride = feat_stride; p.im_s = im_info[0]; p.im_h = im_info[1]; p.im_w = im_info[2]; p.im_scale = im_info[3];
  p.min_size = min_size; p.A = A; p.S = S; p.H = H; p.W = W; p.batch_index = first_batch_index;
  for (int i = 0; i < 6; ++i) p.xf.w[i] = 1.0;                            // :149-150
  p.xf.clip = xform_clip; p.xf.cs = im_info[0]; p.xf.ch = im_info[1]; p.xf.cw = im_info[2];   // :154
  hipStream_t st = m3d::as_stream(stream);
  {   // the score sweeps of the radix select on many workgroups: ~4 k scores per workgroup and pass
    const long long total = (long long)A * S * H * W;
    long long ch = (total + 4095) / 4096;
    const int chunks = (int)(ch < 1 ? 1 : (ch > 128 ? 128 : ch));
    hipLaunchKernelGGL(prop_sel_init_kernel, dim3(batch), dim3(256), 0, st, a);
    for (int level = 0; level < 4; ++level)
      hipLaunchKernelGGL(prop_sel_hist_kernel, dim3(chunks, batch), dim3(kSelT), 0, st, a, level);
    hipLaunchKernelGGL(prop_sel_compact_kernel, dim3(chunks, batch), dim3(kSelT), 0, st, a);
  }
  hipLaunchKernelGGL(proposals_stage1_kernel, dim3(batch), dim3(kWG), 0, st, a);
  if (nms_thresh > 0) launch_mask(a.ws, a.ws_item, prop_nms_offset(K), batch, K, nms_thresh, st);
  hipLaunchKernelGGL(proposals_stage3_kernel, dim3(batch), dim3(kWG), 0, st, a);
  return m3d::check_launch("generate_proposals3d_batched");
}

// ---- rows [0, counts[b]) of every item, packed in item order (the RoIs of a batch as the box head wants them: lib/core/test.py runs the
// head per tile on rois[:num]; the batched head takes all tiles' rows at once).  One workgroup per item; the row offset of an item is the
// sum of the counts before it (batch <= 65535, summed by every workgroup: no second launch, no host round trip).
namespace {
struct CompactSrc { const unsigned* src; long long item_stride_words; int row_words; unsigned* dst; };

// grid (batch, sources): source blockIdx.y of item blockIdx.x.  The first source's workgroups also publish the offsets and - when the
// caller gave a host-mapped mirror - the counts themselves (the host then learns them from the stream event behind this launch: no
// separate device-to-host copy).
__global__ __launch_bounds__(256) void compact_rows_kernel(CompactSrc a, CompactSrc b2, const int32_t* __restrict__ counts, int max_rows,
                                                           int32_t* __restrict__ offsets, int32_t* __restrict__ h_counts) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const CompactSrc c = blockIdx.y == 0 ? a : b2;
  __shared__ long long s_part[256];
  long long part = 0;
  for (int i = tid; i < b; i += 256) part += min(max(counts[i], 0), max_rows);
  s_part[tid] = part;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) s_part[tid] += s_part[tid + st];
    __syncthreads();
  }
  const long long off = s_part[0];
  const int n = min(max(counts[b], 0), max_rows);
  if (blockIdx.y == 0 && tid == 0) {
    if (offsets) {
      offsets[b] = (int32_t)off;
      if (b == (int)gridDim.x - 1) offsets[b + 1] = (int32_t)(off + n);
    }
    if (h_counts) h_counts[b] = n;
  }
  const unsigned* s = c.src + (long long)b * c.item_stride_words;
  unsigned* d = c.dst + off * c.row_words;
  const long long words = (long long)n * c.row_words;
  for (long long e = tid; e < words; e += 256) d[e] = s[e];
}
}  // namespace

M3D_API int m3d_compact_rows(const void* d_src, size_t item_stride_bytes, size_t row_bytes, const int32_t* d_counts, int batch,
                             int max_rows, void* d_dst, int32_t* d_offsets, void* stream) {
  if (!d_src || !d_counts || !d_dst || batch <= 0 || batch > 65535 || max_rows <= 0 || row_bytes == 0) return M3D_EINVAL;
  if ((row_bytes & 3) || (item_stride_bytes & 3) || ((size_t)d_src & 3) || ((size_t)d_dst & 3)) return M3D_EINVAL;
  const CompactSrc a{(const unsigned*)d_src, (long long)(item_stride_bytes / 4), (int)(row_bytes / 4), (unsigned*)d_dst};
  hipLaunchKernelGGL(compact_rows_kernel, dim3(batch, 1), dim3(256), 0, m3d::as_stream(stream), a, a, d_counts, max_rows, d_offsets,
                     (int32_t*)nullptr);
  return m3d::check_launch("compact_rows");
}

/* Two row sets with the same counts (the RoIs and their score indices) in ONE launch, + an optional host-mapped mirror of the (clamped)
 * counts (pinned memory the kernel writes: the caller waits for one stream event instead of issuing a device-to-host copy). */
M3D_API int m3d_compact_rows2(const void* d_src_a, size_t item_stride_bytes_a, size_t row_bytes_a, const void* d_src_b,
                              size_t item_stride_bytes_b, size_t row_bytes_b, const int32_t* d_counts, int batch, int max_rows, void* d_dst_a,
                              void* d_dst_b, int32_t* d_offsets, int32_t* h_counts, void* stream) {
  if (!d_src_a || !d_src_b || !d_counts || !d_dst_a || !d_dst_b || batch <= 0 || batch > 65535 || max_rows <= 0) return M3D_EINVAL;
  if (row_bytes_a == 0 || row_bytes_b == 0 || ((row_bytes_a | row_bytes_b | item_stride_bytes_a | item_stride_bytes_b) & 3)) return M3D_EINVAL;
  if (((size_t)d_src_a | (size_t)d_src_b | (size_t)d_dst_a | (size_t)d_dst_b) & 3) return M3D_EINVAL;
  const CompactSrc a{(const unsigned*)d_src_a, (long long)(item_stride_bytes_a / 4), (int)(row_bytes_a / 4), (unsigned*)d_dst_a};
  const CompactSrc b{(const unsigned*)d_src_b, (long long)(item_stride_bytes_b / 4), (int)(row_bytes_b / 4), (unsigned*)d_dst_b};
  hipLaunchKernelGGL(compact_rows_kernel, dim3(batch, 2), dim3(256), 0, m3d::as_stream(stream), a, b, d_counts, max_rows, d_offsets, h_counts);
  return m3d::check_launch("compact_rows2");
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
  hipStream_t st = m3d::as_stream(stream);
  BoxResArgs a;
  a.scores = d_scores; a.boxes = d_boxes; a.keep_idx = d_keep_idx; a.offsets = d_offsets; a.cls_boxes = d_cls_boxes;
  a.cls_keep = d_cls_keep; a.counts = d_counts; a.ws = (char*)m3d::align_up((size_t)d_ws, 256); a.ws_item = boxres_item_bytes();
  a.nc = num_classes; a.cap = max_rows_per_item; a.det_per_im = detections_per_im; a.score_thresh = score_thresh;
  a.nms_thresh = nms_thresh; a.cls = 0;
  if (num_classes == 1) {
    (void)hipMemsetAsync(d_counts, 0, sizeof(int32_t) * batch, st);
    return m3d::check_launch("box_results3d_batched");
  }
  for (int j = 1; j < num_classes; ++j) {                                 // classes one after the other (:835), each for all items
    a.cls = j;
    hipLaunchKernelGGL(box_results_stage1_kernel, dim3(batch), dim3(kWG), 0, st, a);
    launch_mask(a.ws, a.ws_item, boxres_nms_offset(), batch, max_rows_per_item, nms_thresh, st);
    hipLaunchKernelGGL(box_results_stage3_kernel, dim3(batch), dim3(kWG), 0, st, a);
  }
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
  hipStream_t st = m3d::as_stream(stream);
  hipLaunchKernelGGL(nms_pack_stage1_kernel, dim3(batch), dim3(kWG), 0, st, a);
  launch_mask(a.ws, a.ws_item, 0, batch, max_boxes, thresh, st);
  hipLaunchKernelGGL(nms_pack_stage3_kernel, dim3(batch), dim3(kWG), 0, st, a);
  return m3d::check_launch("nms3d_batched");
}
