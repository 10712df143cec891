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
/* The byte-per-probe textbook encoder that m3d_tiff_lzw_encode (which jumps through runs of zero bytes) must equal byte for byte. */
size_t m3d_tiff_lzw_encode_plain(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);

/* Whole multi-page files built in memory - little-endian classic TIFF, one LZW strip per page, BlackIsZero, no predictor: what
 * libtiff's write_image(page, compression='lzw') gives for 2-D pages (tools/infer_simple.py:241-245, binarization_*.py tails).
 *   m3d_tiff_stack_bound           dst capacity that always suffices
 *   m3d_tiff_encode_stack          vol [pages,height,width] uint8 (bits 8) / little-endian uint16 (bits 16) -> file bytes (0 = failure)
 *   m3d_tiff_encode_window_stack_u8  the stack of ONE uint8 peak response map from its non-zero window alone: page q is slice
 *                                  z_first + q of a [*,height,width] tile that is zero except win[wn,wn,wn] at origin (oz,oy,ox);
 *                                  equals m3d_tiff_encode_stack of the dense map (infer_simple.py:233-245 per map) */
size_t m3d_tiff_stack_bound(int pages, int height, int width, int bits);
size_t m3d_tiff_encode_stack(const void* vol, int pages, int height, int width, int bits, uint8_t* dst, size_t cap);
size_t m3d_tiff_encode_window_stack_u8(const uint8_t* win, int wn, int oz, int oy, int ox, int z_first, int pages, int height,
                                       int width, uint8_t* dst, size_t cap);
/* A tile's whole instance tree in one call: `{dir}/{ch}.tif` = m3d_tiff_encode_window_stack_u8 of wins[ch] ([num_peaks, wn^3] uint8) at
 * origins[ch] (int32 [num_peaks, 3] = (oz, oy, ox)), encoded and written by `threads` worker threads (tools/infer_simple.py:239-245 for
 * all peaks of a tile).  Returns the number of files that could not be written. */
int m3d_tiff_write_window_stacks_u8(const char* dir, const uint8_t* wins, const int32_t* origins, int num_peaks, int wn, int z_first,
                                    int pages, int height, int width, int threads);

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
