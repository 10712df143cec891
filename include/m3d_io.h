/* libm3dio.so: host-side codec of the reference's on-disk container - no GPU, no HIP runtime.
 * The reference writes uint8 PRM stacks and uint16 label stacks as multi-page TIFF with LZW compression through
 * libtiff (tools/infer_simple.py:241-245, tools/binarization_soma.py:106-109, tools/binarization_nuclei.py:151-154)
 * and reads them back with skimage.io.imread (binarization_soma.py:68, binarization_nuclei.py:95).  These three entry
 * points are the strip codec; the IFD framing lives in m3d/io.py. */
#ifndef M3D_IO_H_
#define M3D_IO_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
size_t m3d_tiff_lzw_bound(size_t n);                                                  /* dst capacity that always suffices */
size_t m3d_tiff_lzw_encode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);   /* bytes written; 0 = dst too small */
size_t m3d_tiff_lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);   /* bytes produced (<= cap) */

/* 3D run-length masks ({'counts', 'size'}) of lib/utils/cython_mask_3d.pyx:19-84 / lib/utils/mask_3d.py:15-73:
 * runs over the mask in Fortran order, zeros first.  mask: C-contiguous uint8 [S,H,W].
 * m3d_rle3d_encode returns the number of counts (call again with a larger `cap` if it exceeds it);
 * m3d_rle3d_decode returns 0, or -1 when the counts do not sum to S*H*W (the reference asserts). */
size_t m3d_rle3d_encode(const uint8_t* mask, int S, int H, int W, int64_t* counts, size_t cap);
int m3d_rle3d_decode(const int64_t* counts, size_t ncounts, int S, int H, int W, uint8_t* mask);
#ifdef __cplusplus
}
#endif
#endif
