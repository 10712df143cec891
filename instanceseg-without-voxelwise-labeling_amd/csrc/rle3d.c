/* 3D run-length masks of the reference (lib/utils/cython_mask_3d.pyx:19-84, lib/utils/mask_3d.py:15-73): the container
 * lib/core/test.py:164-173 stores instance masks in when MODEL.MASK_ON is set ({'counts': [...], 'size': [S,H,W]}).
 * Runs are counted over the mask in FORTRAN order (first index fastest: f = s + S*(h + H*w)), starting with a run of
 * zeros (0 long when the mask starts set); any non-zero byte is "set" (np.where); an all-zero mask is the single count S*H*W.
 * Host-side format code, no GPU. */
#include <stddef.h>
#include <stdint.h>

#define M3D_IO_API __attribute__((visibility("default")))

/* mask: C-contiguous uint8 [S,H,W].  Writes at most cap counts; returns the number of counts the encoding has
 * (> cap: counts was too small, call again with that capacity). */
M3D_IO_API size_t m3d_rle3d_encode(const uint8_t* mask, int S, int H, int W, int64_t* counts, size_t cap) {
  const size_t n = (size_t)S * H * W;
  size_t nc = 0;
  int cur = 0;                       /* value of the run being counted; the first run is zeros */
  int64_t run = 0;
  for (int w = 0; w < W; ++w)
    for (int h = 0; h < H; ++h) {
      const uint8_t* p = mask + (size_t)h * W + w;             /* element (s, h, w) at p[s * H * W] */
      for (int s = 0; s < S; ++s) {
        const int v = p[(size_t)s * H * W] != 0;
        if (v == cur) { ++run; continue; }
        if (nc < cap) counts[nc] = run;
        ++nc;
        cur = v; run = 1;
      }
    }
  if (n == 0) return 0;
  if (nc < cap) counts[nc] = run;    /* last run (for an all-zero mask: the only one, = S*H*W, mask_3d.py:34-36) */
  ++nc;
  return nc;
}

/* counts -> C-contiguous uint8 [S,H,W] of 0/1.  Returns 0 on success, -1 if the counts do not sum to S*H*W
 * (the reference asserts, cython_mask_3d.pyx:63). */
M3D_IO_API int m3d_rle3d_decode(const int64_t* counts, size_t ncounts, int S, int H, int W, uint8_t* mask) {
  const size_t n = (size_t)S * H * W;
  int64_t total = 0;
  for (size_t i = 0; i < ncounts; ++i) { if (counts[i] < 0) return -1; total += counts[i]; }
  if ((size_t)total != n) return -1;
  size_t f = 0;                       /* Fortran-order position */
  int val = 0;
  for (size_t i = 0; i < ncounts; ++i) {
    for (int64_t c = 0; c < counts[i]; ++c, ++f) {
      const size_t s = f % (size_t)S, h = (f / (size_t)S) % (size_t)H, w = f / ((size_t)S * H);
      mask[(s * H + h) * W + w] = (uint8_t)val;
    }
    val = !val;
  }
  return 0;
}
