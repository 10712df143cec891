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
// Algorithm: iterative minimum-label propagation over the 26-neighbourhood with pointer jumping; labels only ever
// decrease, so unsynchronised updates are benign and the loop ends after one full pass without change.
#include "m3d_common.h"

namespace {

constexpr int kT = 1024;

struct RoiDims { int ez, ey, ex; };

__device__ inline int block_or(int v, int* sm) {
  __syncthreads();
  if (threadIdx.x == 0) *sm = 0;
  __syncthreads();
  if (v) atomicOr(sm, 1);
  __syncthreads();
  return *sm;
}

// in: mask bytes (non-zero = foreground; invert != 0 flips it).  out: 255 where the voxel belongs to the selected
// component (complement mode: 255 everywhere EXCEPT the largest component of the inverted mask).
__global__ __launch_bounds__(kT) void cc_largest_kernel(const uint8_t* __restrict__ mask, const int64_t* __restrict__ offsets,
                                                        const int* __restrict__ dims, int invert, int tie_last,
                                                        int* __restrict__ labels, int* __restrict__ counts,
                                                        uint8_t* __restrict__ out, int32_t* __restrict__ status) {
  __shared__ int s_flag;
  __shared__ int s_cnt[kT / 64], s_lab[kT / 64];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int64_t beg = offsets[r];
  const int V = (int)(offsets[r + 1] - beg);
  const int ez = dims[3 * r], ey = dims[3 * r + 1], ex = dims[3 * r + 2];
  if (V <= 0 || (long long)ez * ey * ex != V) { if (tid == 0 && status) status[r] = 2; return; }
  const uint8_t* m = mask + beg;
  int* lab = labels + beg;
  int* cnt = counts + beg;
  uint8_t* o = out + beg;
  for (int v = tid; v < V; v += kT) {
    const bool fg = (m[v] != 0) != (invert != 0);
    lab[v] = fg ? v + 1 : 0;
    cnt[v] = 0;
  }
  __syncthreads();
  bool converged = false;
  for (int iter = 0; iter < 8192; ++iter) {            // bounded: every iteration strictly lowers some label
    int changed = 0;
    for (int v = tid; v < V; v += kT) {
      int l = lab[v];
      if (!l) continue;
      const int x = v % ex, y = (v / ex) % ey, z = v / (ex * ey);
      int best = l;
      for (int dz = -1; dz <= 1; ++dz) {
        const int zz = z + dz; if (zz < 0 || zz >= ez) continue;
        for (int dy = -1; dy <= 1; ++dy) {
          const int yy = y + dy; if (yy < 0 || yy >= ey) continue;
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx; if (xx < 0 || xx >= ex) continue;
            const int nl = lab[(zz * ey + yy) * ex + xx];
            if (nl && nl < best) best = nl;
          }
        }
      }
      // pointer jumping: follow the chain of roots a few steps
      for (int k = 0; k < 4; ++k) { const int up = lab[best - 1]; if (up && up < best) best = up; else break; }
      if (best < l) { lab[v] = best; changed = 1; }
    }
    if (!block_or(changed, &s_flag)) { converged = true; break; }
  }
  if (!converged) {                                    // never seen (a 50^3 serpentine needs < 100 sweeps); fail loudly
    if (tid == 0 && status) status[r] = 3;
    for (int v = tid; v < V; v += kT) o[v] = 0;
    return;
  }
  // component sizes (root = label; a converged component has the label of its first voxel)
  for (int v = tid; v < V; v += kT) {
    const int l = lab[v];
    if (l) atomicAdd(&cnt[l - 1], 1);
  }
  __syncthreads();
  int bc = 0, bl = 0;
  for (int v = tid; v < V; v += kT) {                   // ascending v per thread: first/last max handled by compare form
    const int c = cnt[v];
    if (c > bc || (tie_last && c == bc && c > 0)) { bc = c; bl = v + 1; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int oc = __shfl_down(bc, off, 64), ol = __shfl_down(bl, off, 64);
    if (oc > bc || (oc == bc && oc > 0 && (tie_last ? ol > bl : ol < bl))) { bc = oc; bl = ol; }
  }
  if ((tid & 63) == 0) { s_cnt[tid >> 6] = bc; s_lab[tid >> 6] = bl; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kT / 64; ++w)
      if (s_cnt[w] > s_cnt[0] || (s_cnt[w] == s_cnt[0] && s_cnt[w] > 0 && (tie_last ? s_lab[w] > s_lab[0] : s_lab[w] < s_lab[0]))) {
        s_cnt[0] = s_cnt[w]; s_lab[0] = s_lab[w];
      }
    if (status) status[r] = s_cnt[0] > 0 ? 0 : 1;       // 1: no component at all (the reference raises on an empty list)
  }
  __syncthreads();
  const int best = s_lab[0];
  const bool none = s_cnt[0] == 0;
  for (int v = tid; v < V; v += kT) {
    const bool sel = !none && lab[v] == best;
    o[v] = invert ? (sel ? 0 : 255) : (sel ? 255 : 0);
  }
}

// scipy.ndimage.binary_closing(x): dilation then erosion, 6-neighbourhood, one iteration, border_value = 0
__global__ __launch_bounds__(kT) void closing6_kernel(const uint8_t* __restrict__ mask, const int64_t* __restrict__ offsets,
                                                      const int* __restrict__ dims, uint8_t* __restrict__ tmp,
                                                      uint8_t* __restrict__ out) {
  const int r = blockIdx.x, tid = threadIdx.x;
  const int64_t beg = offsets[r];
  const int V = (int)(offsets[r + 1] - beg);
  const int ez = dims[3 * r], ey = dims[3 * r + 1], ex = dims[3 * r + 2];
  if (V <= 0 || (long long)ez * ey * ex != V) return;
  const uint8_t* m = mask + beg;
  uint8_t* t = tmp + beg;
  uint8_t* o = out + beg;
  const int sy = ex, sz = ex * ey;
  for (int v = tid; v < V; v += kT) {
    const int x = v % ex, y = (v / ex) % ey, z = v / sz;
    bool d = m[v] != 0;
    d |= (x > 0 && m[v - 1]) | (x + 1 < ex && m[v + 1]) | (y > 0 && m[v - sy]) | (y + 1 < ey && m[v + sy]) | (z > 0 && m[v - sz]) |
         (z + 1 < ez && m[v + sz]);
    t[v] = d ? 255 : 0;
  }
  __syncthreads();
  __threadfence_block();
  for (int v = tid; v < V; v += kT) {
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

}  // namespace

M3D_API size_t m3d_cc_workspace_bytes(int64_t total_voxels) {
  return total_voxels <= 0 ? 256 : (size_t)total_voxels * (2 * sizeof(int) + 1) + 1024;
}

M3D_API int m3d_cc_largest_batch(const uint8_t* d_mask, const int64_t* d_offsets, const int32_t* d_dims, int num_rois,
                                 int64_t total_voxels, int invert, int tie_last, uint8_t* d_out, int32_t* d_status, void* d_ws,
                                 size_t ws_bytes, void* stream) {
  if (num_rois < 0 || total_voxels < 0) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_mask || !d_offsets || !d_dims || !d_out || !d_ws) return M3D_EINVAL;
  if (ws_bytes < m3d_cc_workspace_bytes(total_voxels)) return M3D_EWORKSPACE;
  int* labels = (int*)m3d::align_up((size_t)d_ws, 256);
  int* counts = labels + total_voxels;
  hipLaunchKernelGGL(cc_largest_kernel, dim3(num_rois), dim3(kT), 0, m3d::as_stream(stream), d_mask, d_offsets, d_dims, invert,
                     tie_last, labels, counts, d_out, d_status);
  return m3d::check_launch("cc_largest_batch");
}

M3D_API int m3d_binary_closing6_batch(const uint8_t* d_mask, const int64_t* d_offsets, const int32_t* d_dims, int num_rois,
                                      int64_t total_voxels, uint8_t* d_out, void* d_ws, size_t ws_bytes, void* stream) {
  if (num_rois < 0 || total_voxels < 0) return M3D_EINVAL;
  if (num_rois == 0) return M3D_OK;
  if (!d_mask || !d_offsets || !d_dims || !d_out || !d_ws) return M3D_EINVAL;
  if (ws_bytes < (size_t)total_voxels + 256) return M3D_EWORKSPACE;
  uint8_t* tmp = (uint8_t*)m3d::align_up((size_t)d_ws, 256);
  hipLaunchKernelGGL(closing6_kernel, dim3(num_rois), dim3(kT), 0, m3d::as_stream(stream), d_mask, d_offsets, d_dims, tmp, d_out);
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
