/* TIFF-flavoured LZW (TIFF 6.0 section 13) for the on-disk formats of the reference's drivers: host-side byte work,
 * no GPU involved.  The reference writes its PRM / label stacks through libtiff with compression='lzw'
 * (tools/infer_simple.py:241-245, tools/binarization_soma.py:106-109, tools/binarization_nuclei.py:151-154) and reads
 * them back through skimage.io.imread; this is the codec of that container, written from the specification:
 * MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, first free code 258, "early change" (the code width grows one
 * code before the table is full), table reset with ClearCode when it reaches 4094 entries. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define M3D_EXPORT __attribute__((visibility("default")))

enum { CLEAR = 256, EOI = 257, FIRST = 258, MAXCODE = 4094, HSIZE = 16384 };

typedef struct { uint8_t* p; size_t cap, n; uint32_t acc; int nbits; int overflow; } BitW;

static void put(BitW* w, int code, int width) {
  w->acc = (w->acc << width) | (uint32_t)code;
  w->nbits += width;
  while (w->nbits >= 8) {
    if (w->n < w->cap) w->p[w->n] = (uint8_t)(w->acc >> (w->nbits - 8)); else w->overflow = 1;
    w->n++;
    w->nbits -= 8;
  }
}

/* worst case output size for n input bytes (12-bit codes + clears + EOI) */
M3D_EXPORT size_t m3d_tiff_lzw_bound(size_t n) { return n + n / 2 + 64; }

/* returns the number of bytes written, or 0 when dst is too small */
M3D_EXPORT size_t m3d_tiff_lzw_encode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  static const int32_t EMPTY = -1;
  int32_t* hkey = (int32_t*)malloc(sizeof(int32_t) * HSIZE);   /* (prefix << 8) | byte */
  uint16_t* hval = (uint16_t*)malloc(sizeof(uint16_t) * HSIZE);
  if (!hkey || !hval) { free(hkey); free(hval); return 0; }
  BitW w = {dst, cap, 0, 0, 0, 0};
  int width = 9, next = FIRST;
  for (int i = 0; i < HSIZE; ++i) hkey[i] = EMPTY;
  put(&w, CLEAR, width);
  if (n > 0) {
    int prefix = src[0];
    for (size_t i = 1; i < n; ++i) {
      const int c = src[i];
      const int32_t key = (prefix << 8) | c;
      uint32_t h = ((uint32_t)key * 2654435761u) >> 18;        /* 14 bits */
      int found = -1;
      while (hkey[h] != EMPTY) {
        if (hkey[h] == key) { found = hval[h]; break; }
        h = (h + 1) & (HSIZE - 1);
      }
      if (found >= 0) { prefix = found; continue; }
      put(&w, prefix, width);
      hkey[h] = key; hval[h] = (uint16_t)next; next++;
      if (next == 512 || next == 1024 || next == 2048) width++;            /* the decoder's table lags one entry behind, */
      if (next == MAXCODE) {                                               /* so it switches at 511 / 1023 / 2047         */
        put(&w, CLEAR, width);
        for (int j = 0; j < HSIZE; ++j) hkey[j] = EMPTY;
        width = 9; next = FIRST;
      }
      prefix = c;
    }
    put(&w, prefix, width);
    next++;                                                     /* the decoder adds an entry for this code too */
    if (next == MAXCODE) { put(&w, CLEAR, width); width = 9; }
    else if (next == 512 || next == 1024 || next == 2048) width++;
  }
  put(&w, EOI, width);
  if (w.nbits > 0) put(&w, 0, 8 - w.nbits);
  free(hkey); free(hval);
  return w.overflow ? 0 : w.n;
}

/* returns the number of bytes produced (<= cap); stops at EOI, end of input or a full output buffer */
M3D_EXPORT size_t m3d_tiff_lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  uint16_t* prefix = (uint16_t*)malloc(sizeof(uint16_t) * 4096);
  uint8_t* suffix = (uint8_t*)malloc(4096);
  uint16_t* length = (uint16_t*)malloc(sizeof(uint16_t) * 4096);
  if (!prefix || !suffix || !length) { free(prefix); free(suffix); free(length); return 0; }
  for (int i = 0; i < 256; ++i) { prefix[i] = 0xFFFF; suffix[i] = (uint8_t)i; length[i] = 1; }
  size_t ip = 0, op = 0;
  uint32_t acc = 0; int nbits = 0, width = 9, next = FIRST, old = -1;
  for (;;) {
    while (nbits < width && ip < n) { acc = (acc << 8) | src[ip++]; nbits += 8; }
    if (nbits < width) break;
    const int code = (int)((acc >> (nbits - width)) & ((1u << width) - 1));
    nbits -= width;
    if (code == EOI) break;
    if (code == CLEAR) { width = 9; next = FIRST; old = -1; continue; }
    if (old < 0) {                                              /* first code after a clear: a literal */
      if (code >= 256 || op >= cap) break;
      dst[op++] = (uint8_t)code; old = code; continue;
    }
    int cur = code;
    uint8_t first;
    if (code < next) {                                          /* known string */
      int len = length[cur];
      if (op + len > cap) break;
      int c = cur;
      for (int k = len - 1; k >= 0; --k) { dst[op + k] = suffix[c]; c = prefix[c]; }
      first = dst[op];
      op += len;
    } else if (code == next) {                                  /* KwKwK: old string + its first byte */
      int len = length[old] + 1;
      if (op + len > cap) break;
      int c = old;
      for (int k = len - 2; k >= 0; --k) { dst[op + k] = suffix[c]; c = prefix[c]; }
      first = dst[op];
      dst[op + len - 1] = first;
      op += len;
    } else break;                                               /* corrupt stream */
    if (next < 4096) {
      prefix[next] = (uint16_t)old; suffix[next] = first; length[next] = (uint16_t)(length[old] + 1);
      next++;
      if (next == 511 || next == 1023 || next == 2047) width++;
    }
    old = cur;
  }
  free(prefix); free(suffix); free(length);
  return op;
}
