"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/m3d.h declares
(no compute calls without a GPU); the product wrappers refuse CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "m3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(m3d_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from m3d._lib import LIB_PATH
    return LIB_PATH


def test_exports_every_declared_symbol(built):
    L = ctypes.CDLL(built)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), s
    from m3d._lib import SYMBOLS
    assert sorted(SYMBOLS) == syms


def test_io_library_exports_its_header(built):
    """include/m3d_io.h (host-side TIFF-LZW and 3-D RLE codecs) <-> csrc/libm3dio.so"""
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "m3d_io.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(m3d_[a-z0-9_]+)\s*\(", src)))
    assert syms == ["m3d_rle3d_decode", "m3d_rle3d_encode", "m3d_tiff_encode_stack", "m3d_tiff_encode_window_stack_u8", "m3d_tiff_lzw_bound",
                    "m3d_tiff_lzw_decode", "m3d_tiff_lzw_encode", "m3d_tiff_lzw_encode_plain", "m3d_tiff_stack_bound", "m3d_tiff_write_window_stacks_u8"]
    L = ctypes.CDLL(os.path.join(os.path.dirname(built), "libm3dio.so"))
    for s in syms:
        assert hasattr(L, s), s


def test_error_strings(built):
    L = ctypes.CDLL(built)
    L.m3d_error_string.restype = ctypes.c_char_p
    assert L.m3d_version() >= 100
    assert L.m3d_error_string(0) == b"ok" and b"invalid" in L.m3d_error_string(-1)


def test_argument_validation_without_gpu(built):
    """Entry points validate shapes before touching the device (reference: roi_align_cuda_3d.c:19-22)."""
    L = ctypes.CDLL(built)
    assert L.m3d_roi_align3d_forward(7, 7, 7, ctypes.c_float(0.125), 2, None, 1, 1, 4, 4, 4, None, 3, 5, None, None) == -1
    assert L.m3d_bbox_overlaps3d(None, 0, None, 0, None, None) == 0
    assert L.m3d_conv3d_forward(None, None, None, 1, 1, 1, 4, 4, 4, 3, None, None, None, 0, None, None) == -1


def test_no_cpu_fallback(built):
    import torch
    import m3d
    with pytest.raises(m3d.M3DError):
        m3d.nms3d(torch.zeros((4, 7)), 0.3)
    with pytest.raises(m3d.M3DError):
        m3d.roi_align3d_forward(torch.zeros((1, 1, 4, 4, 4)), torch.zeros((1, 7)), 7, 7, 7, 0.125, 2)
