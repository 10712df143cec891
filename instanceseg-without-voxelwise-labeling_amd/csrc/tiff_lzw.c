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
/* The textbook greedy encoder, one hash probe per input byte: kept as the definition the run-accelerated encoder below must
 * reproduce byte for byte (tests/test_io_formats.py). */
M3D_EXPORT size_t m3d_tiff_lzw_encode_plain(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
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

/* ---- the production encoder: the same greedy algorithm, the same bytes, but runs of zero bytes cost one step per CODE instead of
 * one hash probe per BYTE.  The stacks this codec exists for are peak response maps and label volumes: a cone-limited response
 * window (84^3 of a 64 x 200 x 200 tile) or a few instances in a sea of zeros.  In a zero run the greedy match walks the chain of
 * dictionary strings 0, 00, 000, ... - zc[k] is the code of 0^k - so the encoder jumps min(run, chain) bytes at once, and extends the
 * chain by one entry when the run outlasts it (exactly the entry the byte-wise encoder would add).  A prefix can only become a chain
 * code through the chain itself (after an emit the new prefix is the literal 0 = zc[1]), so chain entries never need the hash. */
typedef struct { int32_t* hkey; uint16_t* hval; uint16_t* zc; } LzwTables;

static int tables_alloc(LzwTables* t) {
  t->hkey = (int32_t*)malloc(sizeof(int32_t) * HSIZE);
  t->hval = (uint16_t*)malloc(sizeof(uint16_t) * HSIZE);
  t->zc = (uint16_t*)malloc(sizeof(uint16_t) * 4096);
  return t->hkey && t->hval && t->zc;
}
static void tables_free(LzwTables* t) { free(t->hkey); free(t->hval); free(t->zc); }

static size_t zero_run(const uint8_t* p, size_t limit) {      /* number of leading zero bytes of p[0..limit) */
  size_t k = 0;
  while (k + 8 <= limit) {
    uint64_t w;
    memcpy(&w, p + k, 8);
    if (w) break;
    k += 8;
  }
  while (k < limit && p[k] == 0) ++k;
  return k;
}

static size_t lzw_strip(LzwTables* t, const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  int32_t* hkey = t->hkey;
  uint16_t* hval = t->hval;
  uint16_t* zc = t->zc;
  BitW w = {dst, cap, 0, 0, 0, 0};
  int width = 9, next = FIRST;
  memset(hkey, 0xFF, sizeof(int32_t) * HSIZE);                 /* EMPTY = -1 */
  int zmax = 1;                                                /* zc[1..zmax] defined */
  zc[1] = 0;
  put(&w, CLEAR, width);
  if (n > 0) {
    int prefix = src[0];
    int zk = prefix == 0 ? 1 : 0;                              /* prefix == zc[zk] when zk > 0 */
    size_t i = 1;
    while (i < n) {
      const int c = src[i];
      if (zk > 0 && c == 0) {
        const size_t room = (size_t)(zmax - zk);               /* chain entries beyond the current prefix */
        const size_t r = zero_run(src + i, (n - i) < room + 1 ? (n - i) : room + 1);
        const size_t step = r < room ? r : room;
        zk += (int)step; i += step; prefix = zc[zk];
        if (r <= room) continue;                               /* the run (or the input) ended inside the chain */
        /* (zc[zmax], 0) is not in the dictionary: emit, add it, restart from the literal 0 */
        put(&w, prefix, width);
        zc[zmax + 1] = (uint16_t)next; zmax++; next++;
        if (next == 512 || next == 1024 || next == 2048) width++;
        if (next == MAXCODE) {
          put(&w, CLEAR, width);
          memset(hkey, 0xFF, sizeof(int32_t) * HSIZE);
          width = 9; next = FIRST; zmax = 1;
        }
        prefix = 0; zk = 1; ++i;
        continue;
      }
      const int32_t key = (prefix << 8) | c;
      uint32_t h = ((uint32_t)key * 2654435761u) >> 18;        /* 14 bits */
      int found = -1;
      while (hkey[h] != -1) {
        if (hkey[h] == key) { found = hval[h]; break; }
        h = (h + 1) & (HSIZE - 1);
      }
      ++i;
      if (found >= 0) { prefix = found; zk = 0; continue; }
      put(&w, prefix, width);
      hkey[h] = key; hval[h] = (uint16_t)next; next++;
      if (next == 512 || next == 1024 || next == 2048) width++;
      if (next == MAXCODE) {
        put(&w, CLEAR, width);
        memset(hkey, 0xFF, sizeof(int32_t) * HSIZE);
        width = 9; next = FIRST; zmax = 1;
      }
      prefix = c; zk = c == 0 ? 1 : 0;
    }
    put(&w, prefix, width);
    next++;                                                     /* the decoder adds an entry for this code too */
    if (next == MAXCODE) { put(&w, CLEAR, width); width = 9; }
    else if (next == 512 || next == 1024 || next == 2048) width++;
  }
  put(&w, EOI, width);
  if (w.nbits > 0) put(&w, 0, 8 - w.nbits);
  return w.overflow ? 0 : w.n;
}

/* returns the number of bytes written, or 0 when dst is too small */
M3D_EXPORT size_t m3d_tiff_lzw_encode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  LzwTables t;
  if (!tables_alloc(&t)) { tables_free(&t); return 0; }
  const size_t r = lzw_strip(&t, src, n, dst, cap);
  tables_free(&t);
  return r;
}

/* ---- whole multi-page files, built in memory: little-endian classic TIFF, one LZW strip per page, BlackIsZero, no predictor - the
 * layout of m3d/io.py:write_tiff_stack (libtiff's write_image(page, compression='lzw') for 2-D pages, tools/infer_simple.py:241-245),
 * byte for byte. */
typedef struct { uint8_t* p; size_t cap, n; int overflow; } Buf;
static void bput(Buf* b, const void* src, size_t k) {
  if (b->n + k <= b->cap) memcpy(b->p + b->n, src, k); else b->overflow = 1;
  b->n += k;
}
static void bput16(Buf* b, unsigned v) { uint8_t t[2] = {(uint8_t)v, (uint8_t)(v >> 8)}; bput(b, t, 2); }
static void bput32(Buf* b, uint32_t v) { uint8_t t[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; bput(b, t, 4); }
static void bpatch32(Buf* b, size_t at, uint32_t v) {
  if (at + 4 <= b->cap) { b->p[at] = (uint8_t)v; b->p[at + 1] = (uint8_t)(v >> 8); b->p[at + 2] = (uint8_t)(v >> 16); b->p[at + 3] = (uint8_t)(v >> 24); }
}
static void tag(Buf* b, unsigned id, unsigned type, uint32_t val) {
  bput16(b, id); bput16(b, type); bput32(b, 1);
  if (type == 3) { bput16(b, val); bput16(b, 0); } else bput32(b, val);
}

M3D_EXPORT size_t m3d_tiff_stack_bound(int pages, int height, int width, int bits) {
  const size_t page = (size_t)height * width * (bits / 8);
  return 8 + (size_t)pages * (m3d_tiff_lzw_bound(page) + 2 + 11 * 12 + 4 + 4);
}

static void page_trailer(Buf* b, size_t* prev_next, size_t data_off, size_t data_len, int H, int W, int bits) {
  if (b->n & 1) { const uint8_t z = 0; bput(b, &z, 1); }
  const size_t ifd = b->n;
  bpatch32(b, *prev_next, (uint32_t)ifd);
  bput16(b, 11);
  tag(b, 256, 4, (uint32_t)W); tag(b, 257, 4, (uint32_t)H); tag(b, 258, 3, (uint32_t)bits); tag(b, 259, 3, 5); tag(b, 262, 3, 1);
  tag(b, 273, 4, (uint32_t)data_off); tag(b, 277, 3, 1); tag(b, 278, 4, (uint32_t)H); tag(b, 279, 4, (uint32_t)data_len);
  tag(b, 284, 3, 1); tag(b, 339, 3, 1);
  *prev_next = b->n;
  bput32(b, 0);
}

/* vol: [pages, height, width] uint8 (bits = 8) or little-endian uint16 (bits = 16), C-contiguous.  Returns the file size, 0 on
 * failure (dst too small: m3d_tiff_stack_bound always suffices). */
M3D_EXPORT size_t m3d_tiff_encode_stack(const void* vol, int pages, int height, int width, int bits, uint8_t* dst, size_t cap) {
  if (!vol || !dst || pages < 0 || height <= 0 || width <= 0 || (bits != 8 && bits != 16) || cap < 8) return 0;
  LzwTables t;
  if (!tables_alloc(&t)) { tables_free(&t); return 0; }
  Buf b = {dst, cap, 0, 0};
  const uint8_t hdr[8] = {'I', 'I', 42, 0, 0, 0, 0, 0};
  bput(&b, hdr, 8);
  size_t prev_next = 4;
  const size_t page = (size_t)height * width * (bits / 8);
  for (int p = 0; p < pages && !b.overflow; ++p) {
    if (b.n & 1) { const uint8_t z = 0; bput(&b, &z, 1); }
    const size_t off = b.n;
    const size_t k = off < cap ? lzw_strip(&t, (const uint8_t*)vol + (size_t)p * page, page, dst + off, cap - off) : 0;
    if (k == 0) { b.overflow = 1; break; }
    b.n += k;
    page_trailer(&b, &prev_next, off, k, height, width, bits);
  }
  tables_free(&t);
  return b.overflow ? 0 : b.n;
}

/* The uint8 stack of ONE peak response map given only its non-zero window: page q (q = 0 .. pages-1) is slice z = z_first + q of a
 * [*, height, width] tile that is zero everywhere except win[wn,wn,wn] placed at origin (oz, oy, ox) (window voxels outside the tile
 * are ignored).  The file equals m3d_tiff_encode_stack of the dense map; the dense map itself (2.56 MB per peak of a nuclei tile,
 * against 0.59 MB of window) never crosses PCIe and never exists on the host: one page buffer is composed at a time. */
M3D_EXPORT size_t m3d_tiff_encode_window_stack_u8(const uint8_t* win, int wn, int oz, int oy, int ox, int z_first, int pages, int height,
                                                  int width, uint8_t* dst, size_t cap) {
  if (!win || !dst || wn <= 0 || pages < 0 || height <= 0 || width <= 0 || cap < 8) return 0;
  LzwTables t;
  const size_t page = (size_t)height * width;
  uint8_t* pg = (uint8_t*)malloc(page);
  if (!tables_alloc(&t) || !pg) { tables_free(&t); free(pg); return 0; }
  Buf b = {dst, cap, 0, 0};
  const uint8_t hdr[8] = {'I', 'I', 42, 0, 0, 0, 0, 0};
  bput(&b, hdr, 8);
  size_t prev_next = 4;
  const int y0 = oy < 0 ? 0 : oy, y1 = oy + wn > height ? height : oy + wn;
  const int x0 = ox < 0 ? 0 : ox, x1 = ox + wn > width ? width : ox + wn;
  int dirty = 1;                                               /* the page buffer holds something other than zeros */
  for (int q = 0; q < pages && !b.overflow; ++q) {
    const int z = z_first + q, wz = z - oz;
    if (dirty) { memset(pg, 0, page); dirty = 0; }
    if (wz >= 0 && wz < wn && y1 > y0 && x1 > x0) {
      for (int y = y0; y < y1; ++y)
        memcpy(pg + (size_t)y * width + x0, win + ((size_t)wz * wn + (y - oy)) * wn + (x0 - ox), (size_t)(x1 - x0));
      dirty = 1;
    }
    if (b.n & 1) { const uint8_t zz = 0; bput(&b, &zz, 1); }
    const size_t off = b.n;
    const size_t k = off < cap ? lzw_strip(&t, pg, page, dst + off, cap - off) : 0;
    if (k == 0) { b.overflow = 1; break; }
    b.n += k;
    page_trailer(&b, &prev_next, off, k, height, width, 8);
  }
  tables_free(&t); free(pg);
  return b.overflow ? 0 : b.n;
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

/* ---- a whole tile's instance tree in ONE call: `{dir}/{ch}.tif` for ch = 0 .. num_peaks-1, each the uint8 stack of window ch
 * (m3d_tiff_encode_window_stack_u8), encoded and written by `threads` worker threads of this call.  The Python driver hands a tile
 * over with one foreign call (the interpreter lock is free for its whole duration: per-peak Python calls - a future, a buffer, a
 * file object each - cost the launching thread more than the encoding cost the workers, tools/volume_gaps.py).
 * Returns the number of files that could not be written (0 = success). */
#include <pthread.h>
#include <stdio.h>

typedef struct {
  const char* dir; const uint8_t* wins; const int32_t* origins;
  int num_peaks, wn, z_first, pages, height, width, threads, tid;
  int failed;
} TileJob;

static void* tile_worker(void* arg) {
  TileJob* j = (TileJob*)arg;
  const size_t cap = m3d_tiff_stack_bound(j->pages, j->height, j->width, 8);
  uint8_t* buf = (uint8_t*)malloc(cap);
  char path[4096];
  if (!buf) { j->failed = j->num_peaks; return NULL; }
  const size_t w3 = (size_t)j->wn * j->wn * j->wn;
  for (int ch = j->tid; ch < j->num_peaks; ch += j->threads) {
    const size_t n = m3d_tiff_encode_window_stack_u8(j->wins + (size_t)ch * w3, j->wn, j->origins[3 * ch], j->origins[3 * ch + 1],
                                                     j->origins[3 * ch + 2], j->z_first, j->pages, j->height, j->width, buf, cap);
    int ok = n > 0 && snprintf(path, sizeof(path), "%s/%d.tif", j->dir, ch) < (int)sizeof(path);
    if (ok) {
      FILE* f = fopen(path, "wb");
      ok = f != NULL;
      if (f) { ok = fwrite(buf, 1, n, f) == n; ok = (fclose(f) == 0) && ok; }
    }
    if (!ok) j->failed++;
  }
  free(buf);
  return NULL;
}

M3D_EXPORT int m3d_tiff_write_window_stacks_u8(const char* dir, const uint8_t* wins, const int32_t* origins, int num_peaks, int wn, int z_first,
                                               int pages, int height, int width, int threads) {
  if (!dir || !wins || !origins || num_peaks < 0 || wn <= 0 || pages < 0 || height <= 0 || width <= 0) return num_peaks > 0 ? num_peaks : 1;
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  if (threads > num_peaks) threads = num_peaks > 0 ? num_peaks : 1;
  TileJob jobs[64];
  pthread_t th[64];
  int started[64];
  for (int t = 0; t < threads; ++t) {
    jobs[t] = (TileJob){dir, wins, origins, num_peaks, wn, z_first, pages, height, width, threads, t, 0};
    started[t] = (t > 0 && pthread_create(&th[t], NULL, tile_worker, &jobs[t]) == 0);
  }
  tile_worker(&jobs[0]);                                       /* the calling thread is worker 0 */
  int failed = jobs[0].failed;
  for (int t = 1; t < threads; ++t) {
    if (started[t]) pthread_join(th[t], NULL); else tile_worker(&jobs[t]);   /* a thread that could not start: do its share here */
    failed += jobs[t].failed;
  }
  return failed;
}
