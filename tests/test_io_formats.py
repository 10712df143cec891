"""On-disk formats: the LZW TIFF writer/reader against an independent implementation (Pillow's libtiff),
plus the npy / pickle side files.  Host-only; runs without a GPU."""
import os
import pickle
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
from m3d import io as mio  # noqa: E402

PIL = pytest.importorskip("PIL.Image")


def volumes():
    rs = np.random.RandomState(0)
    zz, yy, xx = np.mgrid[0:7, 0:33, 0:41]
    blob = (255 * np.exp(-((zz - 3) ** 2 / 4.0 + (yy - 16) ** 2 / 40.0 + (xx - 20) ** 2 / 60.0))).astype(np.uint8)
    lab = np.zeros((5, 64, 48), np.uint16); lab[1:4, 10:30, 5:25] = 7; lab[2:5, 28:60, 20:44] = 300
    return [blob, rs.randint(0, 256, (3, 17, 19)).astype(np.uint8), np.zeros((2, 8, 8), np.uint8), lab,
            rs.randint(0, 65536, (2, 31, 29)).astype(np.uint16), np.full((1, 200, 300), 255, np.uint8),
            np.tile(np.arange(256, dtype=np.uint8), (4, 64, 16)).reshape(4, 64, 4096)[:, :, :700].copy()]


def test_lzw_round_trip_including_table_resets():
    rs = np.random.RandomState(1)
    for raw in (b"", b"a", b"aaaaaaaaaaaaaaaaaaaaaaaaaaaaaa", bytes(rs.randint(0, 256, 100000).astype(np.uint8)),
                bytes(rs.randint(0, 4, 300000).astype(np.uint8)), bytes(np.zeros(70000, np.uint8))):
        comp = mio.lzw_encode(raw)
        assert mio.lzw_decode(comp, len(raw)) == raw


@pytest.mark.parametrize("i", range(7))
def test_tiff_written_here_is_read_by_libtiff(tmp_path, i):
    vol = volumes()[i]
    p = str(tmp_path / "v.tif")
    mio.write_tiff_stack(p, vol)
    im = PIL.open(p)
    assert im.n_frames == vol.shape[0]
    for k in range(vol.shape[0]):
        im.seek(k)
        assert im.info.get("compression") == "tiff_lzw"
        assert np.array_equal(np.array(im), vol[k]), k
    assert np.array_equal(mio.read_tiff_stack(p), vol)


@pytest.mark.parametrize("i", range(7))
@pytest.mark.parametrize("compression", ["tiff_lzw", "raw"])
def test_tiff_written_by_libtiff_is_read_here(tmp_path, i, compression):
    vol = volumes()[i]
    p = str(tmp_path / "v.tif")
    frames = [PIL.fromarray(vol[k]) for k in range(vol.shape[0])]
    frames[0].save(p, save_all=True, append_images=frames[1:], compression=compression)
    got = mio.read_tiff_stack(p)
    assert got.dtype == vol.dtype and np.array_equal(got, vol)


def test_instance_tree_and_side_files(tmp_path):
    rs = np.random.RandomState(2)
    prms = [rs.randint(0, 256, (6, 20, 20)).astype(np.uint8) for _ in range(3)]
    dets = rs.rand(3, 7)
    sp = str(tmp_path / "instances" / "4")
    mio.save_prm_instances(sp, prms, dets)
    assert sorted(os.listdir(sp)) == ["0.tif", "1.tif", "2.tif", "dets.npy"]
    d2, p2 = mio.load_prm_instances(sp)
    assert d2.dtype == np.float64 and np.array_equal(d2, dets) and all(np.array_equal(a, b) for a, b in zip(prms, p2))
    seg = np.zeros((4, 16, 16), np.uint16); seg[1:3, 2:9, 2:9] = 2
    mio.save_segmentation(str(tmp_path / "seg"), "img1", seg, np.array([[2, 0.9]], np.float32))
    assert np.array_equal(mio.read_tiff_stack(str(tmp_path / "seg" / "img1.tif")), seg)
    boxes = [[], np.arange(14, dtype=np.float32).reshape(2, 7)]
    mio.save_detections(str(tmp_path / "img1.pkl"), boxes)
    back = pickle.load(open(str(tmp_path / "img1.pkl"), "rb"))
    assert list(back) == ["all_boxes"] and np.array_equal(back["all_boxes"][1], boxes[1])


def test_rle3d_matches_the_reference_module():
    """m3d.io.binary_mask_to_rle / rle_to_binary_mask (csrc/rle3d.c) against the reference's own lib/utils/mask_3d.py - pure NumPy,
    imported from where it lies - on random, empty, full, leading-one and single-voxel masks; plus round trips."""
    import importlib.util
    import os
    from m3d import io as mio
    rs = np.random.RandomState(0)
    cases = [np.zeros((3, 4, 5), np.uint8), np.ones((3, 4, 5), np.uint8), (rs.rand(7, 5, 6) > 0.5).astype(np.uint8),
             (rs.rand(16, 9, 11) > 0.9).astype(np.uint8) * 255, np.ones((1, 1, 1), np.uint8), np.zeros((1, 1, 1), np.uint8)]
    m = np.zeros((4, 3, 2), np.uint8); m[0, 0, 0] = 1; cases.append(m)
    m = np.zeros((4, 3, 2), np.uint8); m[-1, -1, -1] = 7; cases.append(m)
    m = (rs.rand(20, 31, 17) > 0.3).astype(np.uint8); m[:, 10:20] = 0; cases.append(m)
    ref = None
    path = "/root/reference/lib/utils/mask_3d.py"
    if os.path.exists(path):                                  # build container only; the expected values below pin it elsewhere
        spec = importlib.util.spec_from_file_location("ref_mask_3d", path)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    for m in cases:
        r = mio.binary_mask_to_rle(m)
        assert r["size"] == list(m.shape) and sum(r["counts"]) == m.size
        back = mio.rle_to_binary_mask(r)
        assert back.dtype == np.uint8 and np.array_equal(back, (m != 0).astype(np.uint8))
        if ref is not None:
            rr = ref.binary_mask_to_rle(m)
            assert [int(v) for v in rr["counts"]] == r["counts"] and list(rr["size"]) == r["size"]
            assert np.array_equal(ref.rle_to_binary_mask(rr), back)
    # known answers (the reference module's own __main__ example, mask_3d.py:75-79, and the edge cases)
    a = np.array([[[1, 1, 1, 0, 0, 0], [1, 1, 1, 0, 0, 0]], [[1, 1, 1, 1, 1, 0], [1, 1, 1, 0, 0, 0]]], np.uint8)
    assert mio.binary_mask_to_rle(a) == {"counts": [0, 12, 1, 1, 3, 1, 6], "size": [2, 2, 6]}
    assert mio.binary_mask_to_rle(np.zeros((2, 3, 4), np.uint8))["counts"] == [24]
    assert mio.binary_mask_to_rle(np.ones((2, 3, 4), np.uint8))["counts"] == [0, 24]
    with pytest.raises(ValueError):
        mio.binary_mask_to_rle(np.zeros((3, 4), np.uint8))
    with pytest.raises(ValueError):
        mio.binary_mask_to_rle(np.zeros((3, 4, 5), np.float32))
    with pytest.raises(AssertionError):
        mio.rle_to_binary_mask({"counts": [5], "size": [2, 2, 2]})


def test_run_accelerated_lzw_equals_the_textbook_encoder_byte_for_byte():
    """m3d_tiff_lzw_encode jumps through runs of zero bytes along the dictionary chain 0, 00, 000, ...; it must emit exactly what the
    byte-per-probe greedy encoder emits - sparse, dense, all-zero, tiny and table-reset-length inputs."""
    import ctypes as C
    L = mio._lib()
    rs = np.random.RandomState(3)

    def enc(fn, a):
        cap = L.m3d_tiff_lzw_bound(a.size)
        d = np.empty(cap, np.uint8)
        return d[:fn(a.ctypes.data, a.size, d.ctypes.data, cap)].tobytes()
    cases = [np.zeros(n, np.uint8) for n in (0, 1, 2, 5, 40000, 1500000)]
    for n in (1, 2, 3, 100, 40000, 400000):
        for dens in (0.0005, 0.01, 0.1, 0.5, 1.0):
            cases.append(((rs.rand(n) < dens) * rs.randint(0, 256, n)).astype(np.uint8))
            cases.append(((rs.rand(n) < dens) * rs.randint(0, 3, n)).astype(np.uint8))
    for a in cases:
        a = np.ascontiguousarray(a)
        comp = enc(L.m3d_tiff_lzw_encode, a)
        assert comp == enc(L.m3d_tiff_lzw_encode_plain, a), a.size
        assert mio.lzw_decode(comp, a.size) == a.tobytes()


def test_c_file_builder_equals_the_python_framing_and_the_window_form_equals_the_dense_map(tmp_path):
    """m3d_tiff_encode_stack == write_tiff_stack_py (independent framing) for uint8 and uint16 stacks; m3d_tiff_encode_window_stack_u8
    (a peak response map from its non-zero window: what the volume driver hands the writer pool) == the stack of the dense map,
    windows sticking out of the tile on every side, slice padding removed (z_first > 0)."""
    rs = np.random.RandomState(4)
    for v in volumes():
        mio.write_tiff_stack_py(str(tmp_path / "a.tif"), v)
        assert mio.encode_tiff_stack(v).tobytes() == open(str(tmp_path / "a.tif"), "rb").read()
    S, H, W, n = 12, 40, 52, 20
    win = (rs.randint(0, 256, (n, n, n)) * (rs.rand(n, n, n) < 0.3)).astype(np.uint8)
    for oz, oy, ox in ((2, 5, 7), (-9, -11, 40), (0, 0, 0), (8, 35, -15), (30, 0, 0), (-25, 3, 3)):
        dense = np.zeros((S, H, W), np.uint8)
        for z in range(n):
            if 0 <= oz + z < S:
                y0, y1, x0, x1 = max(0, oy), min(H, oy + n), max(0, ox), min(W, ox + n)
                if y1 > y0 and x1 > x0:
                    dense[oz + z, y0:y1, x0:x1] = win[z, y0 - oy:y1 - oy, x0 - ox:x1 - ox]
        for zf, pages in ((0, S), (2, S - 3)):
            got = mio.encode_window_stack_u8(win, (oz, oy, ox), zf, pages, H, W).tobytes()
            assert got == mio.encode_tiff_stack(dense[zf:zf + pages]).tobytes(), (oz, oy, ox, zf)
            (tmp_path / "w.tif").write_bytes(got)
            assert np.array_equal(mio.read_tiff_stack(str(tmp_path / "w.tif")), dense[zf:zf + pages])
