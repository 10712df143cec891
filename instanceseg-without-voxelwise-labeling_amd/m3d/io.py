"""On-disk formats of the reference's drivers, host side only.

* multi-page grayscale TIFF, LZW strips - what libtiff's `write_image(page, compression='lzw')` produces for 2-D pages
  (tools/infer_simple.py:241-245; binarization_soma.py:106-109; binarization_nuclei.py:151-154) and what
  skimage.io.imread reads back (binarization_soma.py:68, binarization_nuclei.py:95);
* `dets.npy` beside the per-peak PRM stacks (infer_simple.py:247), the `{name}.npy` score / box tables of the
  binarisation scripts (binarization_soma.py:105, binarization_nuclei.py:150);
* the detection pickle `dict(all_boxes=cls_boxes)` (infer_simple.py:258-265, utils/my_io.py:40-44).
The LZW strip codec is C (csrc/tiff_lzw.c -> libm3dio.so, include/m3d_io.h); the IFD framing is here."""
import ctypes as C
import os
import pickle
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_io = None


def _lib():
    global _io
    if _io is None:
        path = os.environ.get("M3D_IO_LIB_PATH", os.path.join(os.path.dirname(_HERE), "csrc", "libm3dio.so"))
        if not os.path.exists(path):
            raise RuntimeError("libm3dio.so not built (make -C csrc); the TIFF codec has no Python fallback: %s" % path)
        lib = C.CDLL(path)
        lib.m3d_tiff_lzw_bound.restype = C.c_size_t
        lib.m3d_tiff_lzw_bound.argtypes = [C.c_size_t]
        for f in (lib.m3d_tiff_lzw_encode, lib.m3d_tiff_lzw_decode):
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.m3d_tiff_lzw_encode_plain.restype = C.c_size_t
        lib.m3d_tiff_lzw_encode_plain.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.m3d_tiff_stack_bound.restype = C.c_size_t
        lib.m3d_tiff_stack_bound.argtypes = [C.c_int] * 4
        lib.m3d_tiff_encode_stack.restype = C.c_size_t
        lib.m3d_tiff_encode_stack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.m3d_tiff_encode_window_stack_u8.restype = C.c_size_t
        lib.m3d_tiff_encode_window_stack_u8.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p, C.c_size_t]
        lib.m3d_tiff_write_window_stacks_u8.restype = C.c_int
        lib.m3d_tiff_write_window_stacks_u8.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7
        lib.m3d_rle3d_encode.restype = C.c_size_t
        lib.m3d_rle3d_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.m3d_rle3d_decode.restype = C.c_int
        lib.m3d_rle3d_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _io = lib
    return _io


# ---------------------------------------------------------------------------- 3D run-length masks
def binary_mask_to_rle(binary_mask):
    """lib/utils/cython_mask_3d.pyx:19-52 (= lib/utils/mask_3d.py:15-47): uint8 [S,H,W] -> {'counts': [...], 'size': [S,H,W]};
    the container of instance masks in lib/core/test.py:164-173.  Like the Cython original, only uint8 3-D arrays are accepted."""
    m = np.asarray(binary_mask)
    if m.ndim != 3:
        raise ValueError("Buffer has wrong number of dimensions (expected 3, got %d)" % m.ndim)
    if m.dtype != np.uint8:
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' but got '%s'" % m.dtype)
    m = np.ascontiguousarray(m)
    S, H, W = m.shape
    cap = 1024
    while True:
        counts = np.empty((cap,), np.int64)
        n = _lib().m3d_rle3d_encode(m.ctypes.data, S, H, W, counts.ctypes.data, cap)
        if n <= cap:
            return {"counts": [int(v) for v in counts[:n]], "size": [S, H, W]}
        cap = int(n)


def rle_to_binary_mask(rle):
    """lib/utils/cython_mask_3d.pyx:54-84: {'counts', 'size'} -> uint8 [S,H,W] of 0/1."""
    counts = np.ascontiguousarray(rle["counts"], dtype=np.int64)
    S, H, W = (int(v) for v in rle["size"])
    out = np.zeros((S, H, W), np.uint8)
    if _lib().m3d_rle3d_decode(counts.ctypes.data, counts.size, S, H, W, out.ctypes.data) != 0:
        raise AssertionError("sum(counts) != prod(size)")           # cython_mask_3d.pyx:63
    return out


def lzw_encode(raw):
    raw = bytes(raw)
    cap = _lib().m3d_tiff_lzw_bound(len(raw))
    dst = C.create_string_buffer(cap)
    n = _lib().m3d_tiff_lzw_encode(raw, len(raw), dst, cap)
    if n == 0:
        raise RuntimeError("tiff_lzw_encode failed")
    return dst.raw[:n]


def lzw_decode(comp, nbytes):
    comp = bytes(comp)
    dst = C.create_string_buffer(max(nbytes, 1))
    n = _lib().m3d_tiff_lzw_decode(comp, len(comp), dst, nbytes)
    if n != nbytes:
        raise ValueError("tiff_lzw_decode: stream ended after %d of %d bytes" % (n, nbytes))
    return dst.raw[:nbytes]


# TIFF tags used
_W, _H, _BPS, _COMP, _PHOTO, _STRIPOFF, _SPP, _RPS, _STRIPCNT, _PLANAR, _PREDICTOR, _SFMT = \
    256, 257, 258, 259, 262, 273, 277, 278, 279, 284, 317, 339


def encode_tiff_stack(vol):
    """uint8 / uint16 [pages, H, W] -> the bytes of the LZW multi-page TIFF (csrc/tiff_lzw.c builds the whole file; the call
    releases the GIL, so stacks are encoded in parallel on a thread pool)."""
    vol = np.asarray(vol)
    if vol.ndim == 2:
        vol = vol[None]
    if vol.ndim != 3 or vol.dtype not in (np.uint8, np.uint16):
        raise ValueError("write_tiff_stack: need a uint8/uint16 [pages,H,W] array, got %s %s" % (vol.dtype, vol.shape))
    vol = np.ascontiguousarray(vol.astype(vol.dtype.newbyteorder("<"), copy=False))
    P, H, W = vol.shape
    bits = vol.dtype.itemsize * 8
    cap = _lib().m3d_tiff_stack_bound(P, H, W, bits)
    dst = np.empty((cap,), np.uint8)
    n = _lib().m3d_tiff_encode_stack(vol.ctypes.data, P, H, W, bits, dst.ctypes.data, cap)
    if n == 0:
        raise RuntimeError("tiff_encode_stack failed")
    return dst[:n]


def encode_window_stack_u8(win, origin, z_first, pages, height, width, dst=None):
    """The LZW TIFF of one uint8 peak response map from its non-zero window alone: win uint8 [n,n,n] at origin (oz,oy,ox) of a
    [*, height, width] tile; pages = slices z_first .. z_first + pages - 1.  Byte-identical to encode_tiff_stack of the dense map."""
    win = np.ascontiguousarray(win, dtype=np.uint8)
    n = int(win.shape[0])
    assert win.shape == (n, n, n)
    cap = _lib().m3d_tiff_stack_bound(int(pages), int(height), int(width), 8)
    if dst is None or dst.size < cap:
        dst = np.empty((cap,), np.uint8)
    k = _lib().m3d_tiff_encode_window_stack_u8(win.ctypes.data, n, int(origin[0]), int(origin[1]), int(origin[2]), int(z_first), int(pages),
                                               int(height), int(width), dst.ctypes.data, dst.size)
    if k == 0:
        raise RuntimeError("tiff_encode_window_stack_u8 failed")
    return dst[:k]


def write_window_stacks_u8(save_path, wins, origins, z_first, pages, height, width, threads=8):
    """`{save_path}/{ch}.tif` for every window of a tile in ONE foreign call (csrc/tiff_lzw.c: `threads` worker threads encode and write;
    the GIL is free meanwhile).  wins uint8 [P,n,n,n] (C-contiguous), origins int32 [P,3]."""
    wins = np.ascontiguousarray(wins, dtype=np.uint8)
    origins = np.ascontiguousarray(origins, dtype=np.int32)
    P, n = int(wins.shape[0]), int(wins.shape[1])
    assert wins.shape == (P, n, n, n) and origins.shape == (P, 3)
    bad = _lib().m3d_tiff_write_window_stacks_u8(os.fsencode(save_path), wins.ctypes.data, origins.ctypes.data, P, n, int(z_first), int(pages),
                                                 int(height), int(width), int(threads))
    if bad:
        raise IOError("write_window_stacks_u8: %d of %d files could not be written under %s" % (bad, P, save_path))


def window_to_dense_u8(win, origin, z_first, pages, height, width):
    """The dense uint8 map [pages, height, width] a window stands for (zero outside it)."""
    win = np.asarray(win)
    n = win.shape[0]
    oz, oy, ox = (int(v) for v in origin)
    out = np.zeros((pages, height, width), np.uint8)
    z0, z1 = max(z_first, oz), min(z_first + pages, oz + n)
    y0, y1, x0, x1 = max(0, oy), min(height, oy + n), max(0, ox), min(width, ox + n)
    if z1 > z0 and y1 > y0 and x1 > x0:
        out[z0 - z_first:z1 - z_first, y0:y1, x0:x1] = win[z0 - oz:z1 - oz, y0 - oy:y1 - oy, x0 - ox:x1 - ox]
    return out


def write_tiff_stack(path, vol, compression="lzw"):
    """vol: uint8 / uint16 array [pages, H, W] (a 2-D array is one page).  Little-endian classic TIFF, one strip per
    page, BlackIsZero, no predictor (libtiff's default for write_image)."""
    if compression != "lzw":
        return write_tiff_stack_py(path, vol, compression)
    data = encode_tiff_stack(vol)
    with open(path, "wb") as f:
        f.write(memoryview(data))


def write_tiff_stack_py(path, vol, compression="lzw"):
    """The framing of write_tiff_stack in Python, page by page (uncompressed stacks; the independent statement the C file builder
    is tested against)."""
    vol = np.asarray(vol)
    if vol.ndim == 2:
        vol = vol[None]
    if vol.ndim != 3 or vol.dtype not in (np.uint8, np.uint16):
        raise ValueError("write_tiff_stack: need a uint8/uint16 [pages,H,W] array, got %s %s" % (vol.dtype, vol.shape))
    bps = vol.dtype.itemsize * 8
    P, H, W = vol.shape
    comp = {"lzw": 5, None: 1, "none": 1}[compression]
    out = bytearray(b"II*\x00\x00\x00\x00\x00")          # header; first-IFD offset patched below
    prev_next_field = 4
    for p in range(P):
        raw = np.ascontiguousarray(vol[p]).astype("<u%d" % (bps // 8), copy=False).tobytes()
        data = lzw_encode(raw) if comp == 5 else raw
        if len(out) & 1:
            out += b"\x00"
        data_off = len(out)
        out += data
        if len(out) & 1:
            out += b"\x00"
        ifd_off = len(out)
        struct.pack_into("<I", out, prev_next_field, ifd_off)
        tags = [(_W, 4, W), (_H, 4, H), (_BPS, 3, bps), (_COMP, 3, comp), (_PHOTO, 3, 1), (_STRIPOFF, 4, data_off), (_SPP, 3, 1),
                (_RPS, 4, H), (_STRIPCNT, 4, len(data)), (_PLANAR, 3, 1), (_SFMT, 3, 1)]
        out += struct.pack("<H", len(tags))
        for tag, typ, val in tags:
            out += struct.pack("<HHI", tag, typ, 1) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
        prev_next_field = len(out)
        out += b"\x00\x00\x00\x00"
    with open(path, "wb") as f:
        f.write(out)


def read_tiff_stack(path):
    """Reads what the reference's tools produce or consume: classic TIFF, 8/16-bit single-sample pages, strips,
    uncompressed or LZW, optional horizontal predictor, either byte order.  Returns [pages, H, W]."""
    with open(path, "rb") as f:
        buf = f.read()
    bo = {b"II": "<", b"MM": ">"}.get(buf[:2])
    if bo is None or struct.unpack(bo + "H", buf[2:4])[0] != 42:
        raise ValueError("%s: not a classic TIFF" % path)
    tsize = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 16: 8}
    tfmt = {1: "B", 3: "H", 4: "I", 16: "Q"}
    pages = []
    off = struct.unpack(bo + "I", buf[4:8])[0]
    while off:
        n = struct.unpack(bo + "H", buf[off:off + 2])[0]
        tags = {}
        for i in range(n):
            e = off + 2 + 12 * i
            tag, typ, cnt = struct.unpack(bo + "HHI", buf[e:e + 8])
            if typ not in tfmt:
                continue
            size = tsize[typ] * cnt
            src = e + 8 if size <= 4 else struct.unpack(bo + "I", buf[e + 8:e + 12])[0]
            tags[tag] = struct.unpack(bo + "%d%s" % (cnt, tfmt[typ]), buf[src:src + size])
        off = struct.unpack(bo + "I", buf[off + 2 + 12 * n:off + 6 + 12 * n])[0]
        W, H = tags[_W][0], tags[_H][0]
        bps = tags.get(_BPS, (1,))[0]
        comp = tags.get(_COMP, (1,))[0]
        if tags.get(_SPP, (1,))[0] != 1 or bps not in (8, 16) or comp not in (1, 5):
            raise ValueError("%s: unsupported page (samples %s, bits %s, compression %s)" % (path, tags.get(_SPP), bps, comp))
        rps = min(tags.get(_RPS, (H,))[0], H)
        offs, cnts = tags[_STRIPOFF], tags[_STRIPCNT]
        bpp = bps // 8
        raw = bytearray()
        for s, (so, sc) in enumerate(zip(offs, cnts)):
            rows = min(rps, H - s * rps)
            chunk = buf[so:so + sc]
            raw += lzw_decode(chunk, rows * W * bpp) if comp == 5 else chunk[:rows * W * bpp]
        page = np.frombuffer(bytes(raw), dtype=np.dtype("u%d" % bpp).newbyteorder(bo)).reshape(H, W)
        page = page.astype("u%d" % bpp)
        if tags.get(_PREDICTOR, (1,))[0] == 2:
            page = np.cumsum(page, axis=1, dtype=page.dtype)
        pages.append(page)
    return np.stack(pages) if pages else np.zeros((0, 0, 0), np.uint8)


def _parallel(fn, items):
    """One file per peak response map: the LZW codec is C behind ctypes (the GIL is released during the call), so the maps of a
    tile are encoded / decoded on a small thread pool; results in input order."""
    if len(items) < 4:
        return [fn(it) for it in items]
    from concurrent.futures import ThreadPoolExecutor
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    with ThreadPoolExecutor(max_workers=max(1, min(16, n))) as ex:
        return list(ex.map(fn, items))


def save_prm_instances(save_path, prms_u8, dets):
    """tools/infer_simple.py:233-247: one `{ch}.tif` per peak response map (already quantised to uint8 and un-padded)
    and `dets.npy` (float64 [P,7])."""
    os.makedirs(save_path, exist_ok=True)
    _parallel(lambda a: write_tiff_stack(os.path.join(save_path, "%d.tif" % a[0]), np.asarray(a[1], dtype=np.uint8)),
              list(enumerate(prms_u8)))
    np.save(os.path.join(save_path, "dets.npy"), np.asarray(dets))


def load_prm_instances(instance_path):
    """What tools/binarization_*.py read per tile: dets.npy and the n `{i}.tif` stacks (binarization_nuclei.py:60-69,95)."""
    dets = np.load(os.path.join(instance_path, "dets.npy"))
    prms = _parallel(lambda i: read_tiff_stack(os.path.join(instance_path, "%d.tif" % i)), list(range(dets.shape[0])))
    return dets, prms


def save_segmentation(save_path, name, seg_u16, table):
    """binarization_soma.py:104-109 / binarization_nuclei.py:149-154: `{name}.npy` (score / id-box table) and the uint16
    label stack `{name}.tif`."""
    os.makedirs(save_path, exist_ok=True)
    np.save(os.path.join(save_path, "%s.npy" % name), np.asarray(table))
    write_tiff_stack(os.path.join(save_path, "%s.tif" % name), np.asarray(seg_u16, dtype=np.uint16))


def save_detections(det_file, cls_boxes):
    """infer_simple.py:258-265 + utils/my_io.py:40-44: pickle of dict(all_boxes=cls_boxes), highest protocol."""
    with open(os.path.abspath(det_file), "wb") as f:
        pickle.dump(dict(all_boxes=cls_boxes), f, pickle.HIGHEST_PROTOCOL)
