// Per-RoI connected-component post-processing for gfx950 (integer / byte work, L2-bound):
// largest 26-connected component, hole filling, 6-connected binary closing, and instance painting.
// Reference (the step right after Otsu):
//   soma   tools/binarization_soma.py:96-102   labels = skimage.measure.label(box_bi)  (full connectivity = 26),
//          largestCC = labels == argsort(bincount(labels.flat)[1:])[-1] + 1, painted only where the label volume is 0
//   nuclei tools/binarization_nuclei.py:125-145 cc3d.connected_components (26), largest by np.argmax(vol); the same on
//          the complement to fill holes; scipy.ndimage.binary_closing (6-neighbourhood, 1 iteration, border_value 0)
// One workgroup per RoI.  Labels are "index of the component's first voxel in raster order + 1", i.e. the same
// ORDER as skimage / cc3d / scipy label ids, so the reference's tie rules can be stated on them:
//   tie_last = 1  -> among equally large components the one with the HIGHEST id (argsort(...)[-1], soma)
//   tie_last = 0  -> the LOWEST id (np.argmax, nuclei)
// Algorithm (round 2): union-find over the 13 raster-preceding neighbours (atomicMin links the larger root under the smaller, so a
// component's root is its FIRST voxel in raster order - the label order above), then flatten + count, select, write: five
// element-parallel launches over grid (chunks, RoI) instead of one workgroup per RoI sweeping its crop until nothing changes
// (soma tile, 127 crops: 5.1 ms -> see DESIGN; the nuclei crops are up to 10^6 voxels).
#include "m3d_common.h"

namespace {

constexpr int kT = 256;

struct CcArgs {
  const uint8_t* mask; const int64_t* offsets; const int* dims;
  int* parent; int* counts; unsigned long long* key; uint8_t* out; int32_t* status;
  int invert, tie_last;
};

struct RoiView { int64_t beg; int V, ez, ey, ex; bool ok; };
__device__ inline RoiView roi_view(const int64_t* offsets, const int* dims, int r) {
  RoiView q;
  q.beg = offsets[r];
  q.V = (int)(offsets[r + 1] - q.beg);
  q.ez = dims[3 * r]; q.ey = dims[3 * r + 1]; q.ex = dims[3 * r + 2];
  q.ok = q.V > 0 && (long long)q.ez * q.ey * q.ex == q.V;
  return q;
}

__device__ inline int uf_find(const int* parent, int x) {
  int p;
  while ((p = parent[x]) != x) x = p;
  return x;
}
// find with path halving: parents only ever move towards the root, so the unsynchronised shortcut stores are benign
__device__ inline int uf_find_halve(int* parent, int x) {
  while (true) {
    const int p = parent[x];
    if (p == x) return x;
    const int gp = parent[p];
    if (gp != p) parent[x] = gp;
    x = p;
  }
}
__device__ inline void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find_halve(parent, a); b = uf_find_halve(parent, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }      // a > b: hang root a under b
    const int old = atomicMin(&parent[a], b);
    if (old == a) return;
    a = old;                                           // somebody linked a meanwhile: retry from its new parent
  }
}

__global__ __launch_bounds__(kT) void cc_init_kernel(CcArgs a) {
  const int r = blockIdx.y;
  const RoiView q = roi_view(a.offsets, a.dims, r);
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.key[r] = 0ull; if (!q.ok && a.status) a.status[r] = 2; }
  if (!q.ok) return;
  const uint8_t* m = a.mask + q.beg;
  const bool inv = a.invert != 0;
  const int lane = threadIdx.x & 63;
  const int nloop = (q.V + gridDim.x * kT - 1) / (gridDim.x * kT);           // same trip count for the whole wave (ballots below)
  for (int it = 0; it < nloop; ++it) {
    const int v = (it * gridDim.x + blockIdx.x) * kT + threadIdx.x;         // a wave holds 64 consecutive voxels
    const bool in = v < q.V;
    const bool fg = in && ((m[v] != 0) != inv);
    const int x = in ? v % q.ex : 0;
    // x runs start out linked WITHOUT atomics and almost flat: every voxel points at the first voxel of its run inside the wave;
    // a run that continues from the previous wave points one voxel to the left of the wave (whose parent is that run's start there)
    const unsigned long long fgm = __ballot(fg);
    const bool left = lane ? ((fgm >> (lane - 1)) & 1ull) != 0 : (fg && x > 0 && ((m[v - 1] != 0) != inv));
    const bool start = fg && (x == 0 || !left);
    const unsigned long long below = __ballot(start) & ((2ull << lane) - 1ull);
    if (in) {
      int p = -1;
      if (fg) p = below ? (v - lane) + (63 - __clzll((long long)below)) : (v - lane) - 1;
      a.parent[q.beg + v] = p;
      a.counts[q.beg + v] = 0;
    }
  }
}

__global__ __launch_bounds__(kT) void cc_union_kernel(CcArgs a) {
  const int r = blockIdx.y;
  const RoiView q = roi_view(a.offsets, a.dims, r);
  if (!q.ok) return;
  int* parent = a.parent + q.beg;
  const uint8_t* m = a.mask + q.beg;
  const bool inv = a.invert != 0;
  const int sy = q.ex, sz = q.ex * q.ey;
  auto fg = [&](int n) __attribute__((always_inline)) { return (m[n] != 0) != inv; };
  for (int v = blockIdx.x * kT + threadIdx.x; v < q.V; v += gridDim.x * kT) {
    if (!fg(v)) continue;
    const int x = v % q.ex, y = (v / q.ex) % q.ey, z = v / sz;
    const bool xl = x > 0, xr = x + 1 < q.ex;
    // Of the 13 neighbours that precede v in raster order (26-connectivity) only one per group that is connected WITHOUT v needs a
    // union: a row's centre voxel is x-adjacent to both its sides (runs are linked by cc_init_kernel), the centre of the previous
    // plane's 3x3 block is adjacent to all eight others.  Foreground tests read the mask bytes, not the parent words.
    // same plane, previous row
    if (y > 0) {
      const int c = v - sy;
      // inside two overlapping runs only the FIRST voxel of the overlap links them (v-1 and c-1 set: v-1 did, or its own left
      // neighbour): in a solid region that is one union per row pair instead of one per voxel
      if (fg(c)) { if (!(xl && fg(v - 1) && fg(c - 1))) uf_union(parent, v, c); }
      else {
        if (xl && fg(c - 1) && !fg(v - 1)) uf_union(parent, v, c - 1);          // with v-1 set, v-1's own centre is c-1
        if (xr && fg(c + 1)) uf_union(parent, v, c + 1);
      }
    }
    if (z > 0) {
      const int cc = v - sz;
      if (fg(cc)) { if (!(xl && fg(v - 1) && fg(cc - 1))) uf_union(parent, v, cc); }
      else {
        for (int dy = -1; dy <= 1; ++dy) {
          const int yy = y + dy;
          if (yy < 0 || yy >= q.ey) continue;
          const int c = cc + dy * sy;
          if (dy != 0 && fg(c)) uf_union(parent, v, c);
          else {
            if (xl && fg(c - 1)) uf_union(parent, v, c - 1);
            if (xr && fg(c + 1)) uf_union(parent, v, c + 1);
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(kT) void cc_count_kernel(CcArgs a) {
  const int r = blockIdx.y;
  const RoiView q = roi_view(a.offsets, a.dims, r);
  if (!q.ok) return;
  int* parent = a.parent + q.beg;
  int* cnt = a.counts + q.beg;
  const int nloop = (q.V + gridDim.x * kT - 1) / (gridDim.x * kT);           // same trip count for the whole wave (ballots below)
  for (int it = 0; it < nloop; ++it) {
    const int v = (it * gridDim.x + blockIdx.x) * kT + threadIdx.x;
    int root = -1;
    if (v < q.V && parent[v] >= 0) {
      root = uf_find_halve(parent, v);
      parent[v] = root;                                                      // flatten: cc_write_kernel's find is one step
    }
    // one atomic per distinct root of the wave instead of one per voxel: the inverted (hole-filling) pass has ONE huge
    // background component, i.e. millions of increments of a single counter (11 ms on the nuclei tile before this)
    unsigned long long todo = __ballot(root >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int lr = __shfl(root, leader, 64);
      const unsigned long long same = __ballot(root == lr) & todo;
      if ((int)(threadIdx.x & 63) == leader) atomicAdd(&cnt[lr], __popcll(same));
      todo &= ~same;
    }
  }
}

// largest component; ties: tie_last -> highest label (root + 1), else lowest
__global__ __launch_bounds__(kT) void cc_select_kernel(CcArgs a) {
  const int r = blockIdx.y;
  const RoiView q = roi_view(a.offsets, a.dims, r);
  if (!q.ok) return;
  const int* cnt = a.counts + q.beg;
  unsigned long long best = 0ull;
  for (int v = blockIdx.x * kT + threadIdx.x; v < q.V; v += gridDim.x * kT) {
    const int c = cnt[v];
    if (c > 0) {
      const unsigned int lab = (unsigned int)(v + 1);
      const unsigned long long k = ((unsigned long long)(unsigned int)c << 32) | (a.tie_last ? lab : 0xFFFFFFFFu - lab);
      best = k > best ? k : best;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_down(best, off, 64);
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63) == 0 && best) atomicMax(&a.key[r], best);
}

__global__ __launch_bounds__(kT) void cc_write_kernel(CcArgs a) {
  const int r = blockIdx.y;
  const RoiView q = roi_view(a.offsets, a.dims, r);
  if (!q.ok) return;
  const unsigned long long k = a.key[r];
  const bool none = k == 0ull;
  const unsigned int low = (unsigned int)(k & 0xFFFFFFFFu);
  const int best_root = (int)(a.tie_last ? low : 0xFFFFFFFFu - low) - 1;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.status) a.status[r] = none ? 1 : 0;   // 1: no component (the reference raises on an empty list)
  const int* parent = a.parent + q.beg;
  uint8_t* o = a.out + q.beg;
  for (int v = blockIdx.x * kT + threadIdx.x; v < q.V; v += gridDim.x * kT) {
    const bool sel = !none && parent[v] >= 0 && uf_find(parent, v) == best_root;
    o[v] = a.invert ? (sel ? 0 : 255) : (sel ? 255 : 0);
  }
}

// scipy.ndimage.binary_closing(x): dilation then erosion, 6-neighbourhood, one iteration, border_value = 0
__global__ __launch_bounds__(kT) void dilate6_kernel(const uint8_t* __restrict__ mask, const int64_t* __restrict__ offsets,
                                                     const int* __restrict__ dims, uint8_t* __restrict__ tmp) {
  const RoiView q = roi_view(offsets, dims, blockIdx.y);
  if (!q.ok) return;
  const uint8_t* m = mask + q.beg;
  uint8_t* t = tmp + q.beg;
  const int ex = q.ex, ey = q.ey, ez = q.ez, sy = ex, sz = ex * ey;
  for (int v = blockIdx.x * kT + threadIdx.x; v < q.V; v += gridDim.x * kT) {
    const int x = v % ex, y = (v / ex) % ey, z = v / sz;
    bool d = m[v] != 0;
    d |= (x > 0 && m[v - 1]) | (x + 1 < ex && m[v + 1]) | (y > 0 && m[v - sy]) | (y + 1 < ey && m[v + sy]) | (z > 0 && m[v - sz]) |
         (z + 1 < ez && m[v + sz]);
    t[v] = d ? 255 : 0;
  }
}
__global__ __launch_bounds__(kT) void erode6_kernel(const uint8_t* __restrict__ tmp, const int64_t* __restrict__ offsets,
                                                    const int* __restrict__ dims, uint8_t* __restrict__ out) {
  const RoiView q = roi_view(offsets, dims, blockIdx.y);
  if (!q.ok) return;
  const uint8_t* t = tmp + q.beg;
  uint8_t* o = out + q.beg;
  const int ex = q.ex, ey = q.ey, ez = q.ez, sy = ex, sz = ex * ey;
  for (int v = blockIdx.x * kT + threadIdx.x; v < q.V; v += gridDim.x * kT) {
    const int x = v % ex, y = (v / ex) % ey, z = v / sz;
    // erosion with border_value 0: a voxel on the crop border has an outside (= 0) neighbour and is removed
    const bool e = t[v] && x > 0 && x + 1 < ex && y > 0 && y + 1 < ey && z > 0 && z + 1 < ez && t[v - 1] && t[v + 1] && t[v - sy] &&
                   t[v + sy] && t[v - sz] && t[v + sz];
    o[v] = e ? 255 : 0;
  }
}

// Paint instance ids into the label volume: detection i (ids ascending = processing order of the reference loop)
// writes id only where the volume is still 0 (binarization_soma.py:99-102) <=> every voxel keeps the SMALLEST id that
// covers it.  vol must be pre-filled with 0xFFFF; paint_finish turns the sentinel back into 0.
__global__ __launch_bounds__(256) void paint_kernel(const uint8_t* __restrict__ mask, const int64_t* __restrict__ offsets,
                                                    const int* __restrict__ boxes, const int* __restrict__ ids, int D, int H,
                                                    int W, unsigned int* __restrict__ vol32) {
  const int r = blockIdx.y;
  const int x1 = boxes[6 * r], y1 = boxes[6 * r + 1], z1 = boxes[6 * r + 2];
  const int ex = boxes[6 * r + 3] - x1 + 1, ey = boxes[6 * r + 4] - y1 + 1;
  const int64_t beg = offsets[r];
  const int V = (int)(offsets[r + 1] - beg);
  for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
    if (!mask[beg + v]) continue;
    const int x = x1 + v % ex, y = y1 + (v / ex) % ey, z = z1 + v / (ex * ey);
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H && (unsigned)z < (unsigned)D)
      atomicMin(&vol32[((size_t)z * H + y) * W + x], (unsigned int)ids[r]);
  }
}

// After all tiles are painted: the 0xFFFFFFFF sentinel back to 0 (the reference's label volume starts at 0) and, for every id in
// [1, max_id], whether it occurs at all (`mask_id in np.unique(seg)`, binarization_soma.py:103) - one pass, no host read
// (torch.bincount sizes its result from a device-side max and makes the host wait for the whole tile).
__global__ __launch_bounds__(256) void paint_finish_kernel(uint32_t* __restrict__ vol, long long n, int max_id, uint8_t* __restrict__ present) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    uint32_t v = vol[i];
    if (v == 0xFFFFFFFFu) vol[i] = 0u;
    else if (v >= 1u && v <= (uint32_t)max_id && present) present[v] = 1;
  }
}

// Start of a tile's painting in ONE launch: the label volume to the 0xFFFFFFFF sentinel, the `present` flags to 0, and the id of every
// processed detection - idx[r] + first_id when its Otsu and component stages succeeded and its response map is not all zero
// (binarization_soma.py:74-76, :94-98), else 0xFFFFFFFF (never paints).  Replaces a chain of seven element-wise launches.
__global__ __launch_bounds__(256) void paint_begin_kernel(uint32_t* __restrict__ vol, long long n, uint8_t* __restrict__ present, int npresent,
                                                          const int32_t* __restrict__ st_otsu, const int32_t* __restrict__ st_cc,
                                                          const int32_t* __restrict__ qstat /*[maps][4], word 3 = map has a non-zero voxel*/,
                                                          const int64_t* __restrict__ idx, int num_rois, int first_id, int32_t* __restrict__ ids) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
  if ((n & 3) == 0 && ((uintptr_t)vol & 15) == 0) {
    uint4* v4 = reinterpret_cast<uint4*>(vol);
    for (long long i = t; i < (n >> 2); i += step) v4[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  } else {
    for (long long i = t; i < n; i += step) vol[i] = 0xFFFFFFFFu;
  }
  for (long long i = t; i < npresent; i += step) present[i] = 0;
  for (long long r = t; r < num_rois; r += step) {
    const long long m = idx[r];
    const bool ok = st_otsu[r] == 0 && st_cc[r] == 0 && (qstat == nullptr || qstat[4 * m + 3] != 0);
    ids[r] = ok ? (int32_t)(m + first_id) : -1;
  }
}

}  // namespace

constexpr size_t kCcKeyBytes = 65536 * sizeof(unsigned long long);   // one selection key per RoI (grid.y <= 65535)

// chunks per RoI: enough workgroups to fill the chip a few times over, no more than the average crop has 256-voxel pieces
static int cc_chunks(int num_rois, int64_t total_voxels) {
  long long per = (total_voxels / (num_rois > 0 ? num_rois : 1) + kT - 1) / kT;
  long long want = (8192 + num_rois - 1) / num_rois;
  long long c = want < per ? want : per;
  return (int)(c < 1 ? 1 : (c > 1024 ? 1024 : c));
}

M3D_API size_t m3d_cc_workspace_bytes(int64_t total_voxels) {
  return (total_voxels <= 0 ? 256 : (size_t)total_voxels * (2 * sizeof(int) + 1) + 1024) + kCcKeyBytes;
}

M3D_API int m3d_cc_largest_batch(const uint8_t* d_mask, const int64_t* d_offsets, const int32_t* d_dims, int num_rois,
                                 int64_t total_voxels, int invert, int tie_last, uint8_t* d_out, int32_t* d_status, void* d_ws,
                                 size_t ws_bytes, void* stream) {
  if (num_rois < 0 || total_voxels < 0) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_mask || !d_offsets || !d_dims || !d_out || !d_ws) return M3D_EINVAL;
  if (ws_bytes < m3d_cc_workspace_bytes(total_voxels)) return M3D_EWORKSPACE;
  if (num_rois > 65535) return M3D_EUNSUPPORTED;
  CcArgs a;
  a.mask = d_mask; a.offsets = d_offsets; a.dims = d_dims; a.out = d_out; a.status = d_status; a.invert = invert; a.tie_last = tie_last;
  a.key = (unsigned long long*)m3d::align_up((size_t)d_ws, 256);
  a.parent = (int*)((char*)a.key + kCcKeyBytes);
  a.counts = a.parent + total_voxels;
  const dim3 grid(cc_chunks(num_rois, total_voxels), num_rois), block(kT);
  hipStream_t st = m3d::as_stream(stream);
  hipLaunchKernelGGL(cc_init_kernel, grid, block, 0, st, a);
  hipLaunchKernelGGL(cc_union_kernel, grid, block, 0, st, a);
  hipLaunchKernelGGL(cc_count_kernel, grid, block, 0, st, a);
  hipLaunchKernelGGL(cc_select_kernel, grid, block, 0, st, a);
  hipLaunchKernelGGL(cc_write_kernel, grid, block, 0, st, a);
  return m3d::check_launch("cc_largest_batch");
}

M3D_API int m3d_binary_closing6_batch(const uint8_t* d_mask, const int64_t* d_offsets, const int32_t* d_dims, int num_rois,
                                      int64_t total_voxels, uint8_t* d_out, void* d_ws, size_t ws_bytes, void* stream) {
  if (num_rois < 0 || total_voxels < 0) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_mask || !d_offsets || !d_dims || !d_out || !d_ws) return M3D_EINVAL;
  if (ws_bytes < (size_t)total_voxels + 256) return M3D_EWORKSPACE;
  if (num_rois > 65535) return M3D_EUNSUPPORTED;
  uint8_t* tmp = (uint8_t*)m3d::align_up((size_t)d_ws, 256);
  const dim3 grid(cc_chunks(num_rois, total_voxels), num_rois), block(kT);
  hipLaunchKernelGGL(dilate6_kernel, grid, block, 0, m3d::as_stream(stream), d_mask, d_offsets, d_dims, tmp);
  hipLaunchKernelGGL(erode6_kernel, grid, block, 0, m3d::as_stream(stream), (const uint8_t*)tmp, d_offsets, d_dims, d_out);
  return m3d::check_launch("binary_closing6_batch");
}

M3D_API int m3d_paint_instances(const uint8_t* d_mask, const int64_t* d_offsets, const int32_t* d_boxes, const int32_t* d_ids,
                                int num_rois, int depth, int height, int width, uint32_t* d_volume, void* stream) {
  if (num_rois < 0 || depth <= 0 || height <= 0 || width <= 0) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_mask || !d_offsets || !d_boxes || !d_ids || !d_volume) return M3D_EINVAL;
  hipLaunchKernelGGL(paint_kernel, dim3(64, num_rois), dim3(256), 0, m3d::as_stream(stream), d_mask, d_offsets, d_boxes, d_ids,
                     depth, height, width, d_volume);
  return m3d::check_launch("paint_instances");
}

M3D_API int m3d_paint_finish(uint32_t* d_volume, int64_t num_voxels, int max_id, uint8_t* d_present, void* stream) {
  if (!d_volume || num_voxels < 0 || max_id < 0) return M3D_EINVAL;
  if (num_voxels == 0) return M3D_OK;
  const int blocks = (int)min((long long)4096, (long long)((num_voxels + 255) / 256));
  hipLaunchKernelGGL(paint_finish_kernel, dim3(blocks), dim3(256), 0, m3d::as_stream(stream), d_volume, (long long)num_voxels, max_id, d_present);
  return m3d::check_launch("paint_finish");
}

/* Sentinel fill of the label volume + zero fill of d_present [num_present] + the paint ids of num_rois processed detections (d_ids[r] =
 * d_idx[r] + first_id, or -1 where d_status_otsu[r] / d_status_cc[r] != 0 or map d_idx[r] is all zero: d_map_stats = the workspace of
 * m3d_prm_quantize_windows*_u8, null = every map counts as non-empty), in one launch.  Then m3d_paint_instances, m3d_paint_finish. */
M3D_API int m3d_paint_begin(uint32_t* d_volume, int64_t num_voxels, uint8_t* d_present, int num_present, const int32_t* d_status_otsu,
                            const int32_t* d_status_cc, const int32_t* d_map_stats, const int64_t* d_idx, int num_rois, int first_id,
                            int32_t* d_ids, void* stream) {
  if (!d_volume || num_voxels <= 0 || num_present < 0 || num_rois < 0) return M3D_EINVAL;
  if ((num_present && !d_present) || (num_rois && (!d_status_otsu || !d_status_cc || !d_idx || !d_ids))) return M3D_EINVAL;
  const int blocks = (int)min((long long)2048, (long long)((num_voxels / 4 + 255) / 256 + 1));
  hipLaunchKernelGGL(paint_begin_kernel, dim3(blocks), dim3(256), 0, m3d::as_stream(stream), d_volume, (long long)num_voxels, d_present,
                     num_present, d_status_otsu, d_status_cc, d_map_stats, d_idx, num_rois, first_id, d_ids);
  return m3d::check_launch("paint_begin");
}
