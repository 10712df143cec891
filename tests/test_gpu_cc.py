"""GPU parity: largest 26-connected component / hole filling / 6-connected closing / instance painting
(csrc/cc3d.hip) against the oracle's scipy restatement of tools/binarization_soma.py:96-105 and
tools/binarization_nuclei.py:125-150.  Byte / index work: bit-exact."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m3d():
    import m3d as _m
    assert torch.cuda.is_available()
    return _m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def pack(masks):
    offs = np.concatenate(([0], np.cumsum([m.size for m in masks]))).astype(np.int64)
    dims = np.array([m.shape for m in masks], np.int32)
    flat = np.concatenate([m.ravel() for m in masks]).astype(np.uint8)
    return dev(flat), dev(offs), dev(dims), offs


def snake(shape):
    """One long serpentine component + a few specks: worst case for label propagation."""
    m = np.zeros(shape, np.uint8)
    D, H, W = shape
    for z in range(0, D, 2):
        for y in range(0, H, 2):
            m[z, y, :] = 255
            if y + 1 < H:
                m[z, y + 1, (W - 1) if (y // 2) % 2 == 0 else 0] = 255
        if z + 1 < D:
            last_y = (H - 1) // 2 * 2
            m[z + 1, last_y if (z // 2) % 2 == 0 else 0, 0] = 255
    return m


def random_masks(rs, n):
    out = []
    for i in range(n):
        shape = tuple(int(v) for v in rs.randint(1, 40, 3))
        p = rs.choice([0.05, 0.15, 0.3, 0.6, 0.9])
        out.append(((rs.rand(*shape) < p) * 255).astype(np.uint8))
    return out


def test_largest_cc_both_tie_rules(m3d):
    rs = np.random.RandomState(0)
    masks = random_masks(rs, 40)
    # explicit size ties: two / three equal blobs, first and last in raster order
    t = np.zeros((5, 9, 9), np.uint8); t[0, 0, 0:2] = 255; t[2, 4, 4:6] = 255; t[4, 8, 7:9] = 255
    masks += [t, snake((9, 13, 17)), snake((24, 30, 31)), np.full((3, 4, 5), 255, np.uint8), np.zeros((4, 4, 4), np.uint8),
              np.array([[[255]]], np.uint8)]
    flat, offs, dims, ho = pack(masks)
    for tie_last, ref in ((True, O.largest_cc_soma), (False, O.largest_cc_nuclei)):
        out, status = m3d.cc_largest_batch(flat, offs, dims, invert=False, tie_last=tie_last)
        out, status = out.cpu().numpy(), status.cpu().numpy()
        for r, m in enumerate(masks):
            got = out[ho[r]:ho[r + 1]].reshape(m.shape)
            if not m.any():
                assert status[r] == 1 and not got.any()
                continue
            assert status[r] == 0
            assert np.array_equal(got, ref(m).astype(np.uint8) * 255), (r, tie_last, m.shape)


def test_fill_holes_and_closing(m3d):
    rs = np.random.RandomState(1)
    masks = []
    for i in range(30):
        shape = tuple(int(v) for v in rs.randint(3, 36, 3))
        zz, yy, xx = np.mgrid[0:shape[0], 0:shape[1], 0:shape[2]]
        c = [s / 2 for s in shape]
        d = ((zz - c[0]) / (shape[0] / 2.5)) ** 2 + ((yy - c[1]) / (shape[1] / 2.5)) ** 2 + ((xx - c[2]) / (shape[2] / 2.5)) ** 2
        m = (d < 1) & (rs.rand(*shape) < 0.85)                   # a blob with holes; touches borders sometimes
        if not m.any():
            m[0, 0, 0] = True
        masks.append(m.astype(np.uint8) * 255)
    flat, offs, dims, ho = pack(masks)
    cc, st = m3d.cc_largest_batch(flat, offs, dims, invert=False, tie_last=False)
    filled, _ = m3d.cc_largest_batch(cc, offs, dims, invert=True, tie_last=False)
    closed = m3d.binary_closing6_batch(filled, offs, dims)
    closed = closed.cpu().numpy()
    for r, m in enumerate(masks):
        ref = O.fill_and_close_nuclei(O.largest_cc_nuclei(m))
        assert np.array_equal(closed[ho[r]:ho[r + 1]].reshape(m.shape), ref.astype(np.uint8) * 255), r


def test_paint_first_writer_wins(m3d):
    rs = np.random.RandomState(2)
    D, H, W = 12, 30, 33
    R = 25
    boxes, masks = [], []
    for r in range(R):
        lo = np.array([rs.randint(0, W - 6), rs.randint(0, H - 6), rs.randint(0, D - 3)])
        hi = np.minimum(lo + rs.randint(1, 14, 3), [W - 1, H - 1, D - 1])
        boxes.append([lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]])
        masks.append(((rs.rand(hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1) < 0.6) * 255).astype(np.uint8))
    flat, offs, dims, ho = pack(masks)
    ids = np.arange(1, R + 1, dtype=np.int32)
    ids[7] = -1                                                    # a skipped detection
    vol = m3d.paint_instances(flat, offs, dev(np.array(boxes, np.int32)), dev(ids), (D, H, W)).cpu().numpy()
    ref = np.zeros((D, H, W), np.int32)
    for r, (x1, y1, z1, x2, y2, z2) in enumerate(boxes):
        if ids[r] < 0:
            continue
        v = ref[z1:z2 + 1, y1:y2 + 1, x1:x2 + 1]
        z = v == 0
        v[z] = (masks[r] > 0).astype(np.int32)[z] * ids[r]
    assert np.array_equal(vol, ref)


@pytest.mark.parametrize("mode", ["soma", "nuclei"])
def test_segment_tile_bit_exact(m3d, mode):
    """quantise -> crop/normalise -> Otsu -> largest CC (-> fill -> close) -> paint, device vs oracle loop."""
    from m3d.binarize import segment_tile, det_boxes_int
    rs = np.random.RandomState(5)
    D, H, W = 24, 64, 64
    zz, yy, xx = np.mgrid[0:D, 0:H, 0:W]
    img = (rs.randn(D, H, W) * 12 + 110).clip(0, 65535)
    R = 14
    prms, dets = [], []
    for r in range(R):
        c = np.array([rs.uniform(4, D - 4), rs.uniform(8, H - 8), rs.uniform(8, W - 8)])
        rad = rs.uniform(3, 6)
        d2 = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2
        img += rs.uniform(300, 900) * np.exp(-d2 / (2 * rad * rad))
        p = np.exp(-d2 / (2 * (rad * 0.8) ** 2)).astype(np.float32) * (rs.rand(D, H, W).astype(np.float32) * 0.3 + 0.7)
        p[d2 > (3 * rad) ** 2] = 0
        if r == 3:
            p[:] = 0                                                # an empty PRM: skipped, but keeps its mask id
        prms.append(p / max(p.sum(), 1e-9))
        h = 2.2 * rad
        dets.append([c[2] - h, c[1] - h, c[0] - h, c[2] + h, c[1] + h, c[0] + h, rs.uniform(0.5, 1)])
    img = img.clip(0, 65535).astype(np.uint16)
    prms = np.stack(prms).astype(np.float32)
    dets = np.array(dets, np.float32)
    if mode == "soma":                                              # soma boxes are not clamped by the reference
        dets[:, :3] = np.maximum(dets[:, :3], 0); dets[:, 3] = np.minimum(dets[:, 3], W - 1)
        dets[:, 4] = np.minimum(dets[:, 4], H - 1); dets[:, 5] = np.minimum(dets[:, 5], D - 1)
    labels, painted = segment_tile(torch.from_numpy(img).cuda(), dev(prms), dets, mode, max_gray_range=4096)
    q_ref = np.stack([O.quantize_prm_u8(p) if p.max() > 0 else np.zeros(p.shape, np.uint8) for p in prms])
    seg, painted_ref = O.segment_tile(img, q_ref, det_boxes_int(dets, (D, H, W), mode), mode)
    assert np.array_equal(labels.cpu().numpy().astype(np.uint16), seg)
    assert np.array_equal(painted.cpu().numpy(), painted_ref)
    assert painted_ref.sum() >= R - 3 and not painted_ref[3]


def _synth_volume_with_tiles(dataset, rs):
    """A volume with Gaussian blobs, and per-tile (dets, uint8 PRM) records for every tile that contains a blob -
    the same blob shows up in all overlapping tiles, so cross-tile NMS and first-writer painting both matter."""
    from m3d.binarize import soma_tiles, nuclei_tiles
    if dataset == "soma":
        S, H, W = 128, 256, 256
        grid, tshape, nblob = soma_tiles(), (64, 160, 160), 14
    else:
        S, H, W = 64, 300, 300          # boxes must be >= 32 wide or touch the border to survive binarization_nuclei.py:71-77
        grid, tshape, nblob = nuclei_tiles(H, W), (S, 200, 200), 10
    img = (rs.randn(S, H, W) * 12 + 110)
    blobs = []
    for i in range(nblob):
        if dataset == "soma":
            c = np.array([rs.uniform(6, S - 6), rs.uniform(12, H - 12), rs.uniform(12, W - 12)])
            rad = rs.uniform(3, 6)
        else:
            c = np.array([rs.uniform(24, 40), rs.uniform(22, H - 22), rs.uniform(22, W - 22)])
            rad = rs.uniform(7.5, 9) if i % 4 else rs.uniform(4, 6)
        lo = np.maximum(np.floor(c - 4 * rad).astype(int), 0); hi = np.minimum(np.ceil(c + 4 * rad).astype(int) + 1, [S, H, W])
        zz, yy, xx = np.mgrid[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        d2 = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2
        img[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] += rs.uniform(400, 900) * np.exp(-d2 / (2 * rad * rad))
        blobs.append((c, rad))
    img = img.clip(0, 65535).astype(np.uint16)
    tiles = {}
    for num, ss, hs, ws in grid:
        dets, prms = [], []
        for c, rad in blobs:
            lc = c - np.array([ss, hs, ws])
            h = 2.2 * rad
            if np.any(lc - h < 1) or np.any(lc + h > np.array(tshape) - 2):
                continue
            zz, yy, xx = np.mgrid[0:tshape[0], 0:tshape[1], 0:tshape[2]]
            d2 = (zz - lc[0]) ** 2 + (yy - lc[1]) ** 2 + (xx - lc[2]) ** 2
            p = np.exp(-d2 / (2 * (rad * 0.8) ** 2)) * (rs.rand(*tshape) * 0.3 + 0.7)
            p[d2 > (3 * rad) ** 2] = 0
            prms.append((p / p.max() * 255).astype(np.uint8))
            jit = rs.uniform(-0.4, 0.4, 6)
            dets.append([lc[2] - h + jit[0], lc[1] - h + jit[1], lc[0] - h + jit[2], lc[2] + h + jit[3], lc[1] + h + jit[4],
                         lc[0] + h + jit[5], rs.uniform(0.45, 1.0)])
        if dets:
            tiles[num] = (np.array(dets, np.float64), np.stack(prms))
    return img, tiles


@pytest.mark.parametrize("dataset", ["soma", "nuclei"])
def test_binarize_volume_through_the_file_tree(m3d, dataset, tmp_path):
    """PRM instance tree on disk (LZW TIFF + dets.npy, as infer_simple writes it) -> cross-tile NMS -> per-detection
    Otsu / components -> uint16 label stack + table, against the oracle's restatement of the two scripts."""
    from m3d.binarize import binarize_volume
    from m3d import io as mio
    rs = np.random.RandomState(21)
    img, tiles = _synth_volume_with_tiles(dataset, rs)
    assert len(tiles) >= 4
    root = str(tmp_path / "prm" / "img1")
    for num, (dets, prms) in tiles.items():
        mio.save_prm_instances(root + "/instances/%d" % num, prms, dets)
    loaded = {}
    for num in tiles:
        d, p = mio.load_prm_instances(root + "/instances/%d" % num)
        loaded[num] = (d, np.stack(p))
        assert np.array_equal(loaded[num][1], tiles[num][1])
    seg, table = binarize_volume(img, loaded, dataset, max_gray_range=4096)
    seg_ref, table_ref = O.binarize_volume(img, tiles, dataset)
    assert seg_ref.max() >= 5 and len(table_ref) >= 5
    assert np.array_equal(seg, seg_ref)
    assert table.shape == table_ref.shape and np.array_equal(table, table_ref.astype(np.float64))
    mio.save_segmentation(str(tmp_path / "out"), "img1", seg, table)
    assert np.array_equal(mio.read_tiff_stack(str(tmp_path / "out" / "img1.tif")), seg_ref)


@pytest.mark.parametrize("shape", [(7, 9, 11), (1, 5, 3), (3, 2, 70), (20, 33, 47), (64, 128, 96)])
def test_prefilters_bit_exact_with_scipy(m3d, shape):
    """ndimage.gaussian_filter(uint16, sigma=1) (uint16 after every axis pass, reflect borders also for axes shorter than
    the kernel radius) and ndimage.median_filter(size=3): the reference's own SciPy calls (binarization_nuclei.py:44-45)."""
    from scipy import ndimage
    rs = np.random.RandomState(sum(shape))
    a = rs.randint(0, 65536 if shape[0] % 2 else 4000, shape).astype(np.uint16)
    g = m3d.gaussian_filter_u16(dev(a), 1.0)
    ref_g = ndimage.gaussian_filter(a, sigma=1)
    assert np.array_equal(g.cpu().numpy(), ref_g)
    m = m3d.median_filter3_u16(g)
    assert np.array_equal(m.cpu().numpy(), ndimage.median_filter(ref_g, size=3))
    g2 = m3d.gaussian_filter_u16(dev(a), 2.0)                      # another sigma: radius 8
    assert np.array_equal(g2.cpu().numpy(), ndimage.gaussian_filter(a, sigma=2))


def _many_tile_nuclei_set(rs, S=12, H=1000, W=800, per_tile=270, real_every=4):
    """63 overlapping nuclei tiles (9 x 7 of 200 x 200, binarization_nuclei.py:50-56), 270 detections each = 17 010 rows in the
    cross-tile NMS - above the 16 384 boxes one workgroup can resolve.  Every fourth tile carries ONE real detection (an anisotropic
    blob in the tile centre, the largest box of its neighbourhood, score > 0.4); the other rows are low-score boxes >= 32 wide (they
    pass :71-77, take part in the volume-ordered NMS and are dropped by the 0.4 score cut of :83-85).  All rows of a tile share one
    uint8 map (a stride-0 view), so the set costs 0.5 MB per tile."""
    from m3d.binarize import nuclei_tiles
    grid = nuclei_tiles(H, W)
    img = (rs.randn(S, H, W) * 12 + 110)
    tiles = {}
    zz, yy, xx = np.mgrid[0:S, 0:200, 0:200]
    for num, ss, hs, ws in grid:
        n = per_tile
        c = rs.uniform(20, 180, (n, 2))                         # (x, y) centres inside the tile
        wh = rs.uniform(32.5, 44, (n, 2))
        z1 = rs.uniform(0, 3, n); z2 = rs.uniform(S - 4, S - 1.01, n)
        dets = np.stack([c[:, 0] - wh[:, 0] / 2, c[:, 1] - wh[:, 1] / 2, z1, c[:, 0] + wh[:, 0] / 2, c[:, 1] + wh[:, 1] / 2, z2,
                         np.round(rs.uniform(0.01, 0.4, n), 3)], 1)
        dets[:, [0, 1]] = np.maximum(dets[:, [0, 1]], 0); dets[:, [3, 4]] = np.minimum(dets[:, [3, 4]], 199)
        pmap = np.zeros((S, 200, 200), np.uint8)
        if num % real_every == 0:
            cz, cy, cx = (S - 1) / 2.0, 100.0 + rs.uniform(-3, 3), 100.0 + rs.uniform(-3, 3)
            d2 = ((zz - cz) / 3.0) ** 2 + ((yy - cy) / 9.0) ** 2 + ((xx - cx) / 9.0) ** 2
            blob = np.exp(-d2 / 2)
            img[:, hs:hs + 200, ws:ws + 200] += rs.uniform(500, 900) * blob
            p = blob * (rs.rand(S, 200, 200) * 0.3 + 0.7)
            p[d2 > 9] = 0
            pmap = (p / p.max() * 255).astype(np.uint8)
            dets[0] = [cx - 26, cy - 26, 0.3, cx + 26, cy + 26, S - 1.2, rs.uniform(0.5, 1.0)]     # 53 x 53 x S: the biggest box around
        tiles[num] = (dets.astype(np.float64), np.broadcast_to(pmap, (n, S, 200, 200)))
    return img.clip(0, 65535).astype(np.uint16), tiles, grid


def test_binarize_volume_with_more_detections_than_one_nms_workgroup_resolves(m3d):
    """tools/binarization_nuclei.py:81 runs nms_3d_volume over EVERY tile's detections at once, with no bound.  63 tiles x 270 rows =
    17 010 boxes go through the blocked form of m3d_nms3d (csrc/box_ops.hip: N > 16 384); labels and table against the oracle's
    restatement of the script, bit for bit, and the NMS stage alone against the oracle's NMS on the same rows."""
    from m3d.binarize import binarize_volume
    rs = np.random.RandomState(63)
    img, tiles, grid = _many_tile_nuclei_set(rs)
    assert len(grid) >= 60 and sum(len(t[0]) for t in tiles.values()) > 16384
    seg, table = binarize_volume(img, tiles, "nuclei", max_gray_range=4096)
    seg_ref, table_ref = O.binarize_volume(img, tiles, "nuclei")
    assert seg_ref.max() >= 5 and len(table_ref) >= 5
    assert np.array_equal(seg, seg_ref)
    assert table.shape == table_ref.shape and np.array_equal(table, table_ref.astype(np.float64))
