#!/usr/bin/env python3
"""Differential fuzzing of the HIP ops against the CPU oracle (test infrastructure; not part of the product path):
random shapes / parameters / degenerate inputs per op, bit-exact where the contract says so.
  python tools/fuzz_parity.py [seconds per op] [seed]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, m3d
import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rand_boxes(rs, n, span=100, smax=40, grid=None):
    c = rs.uniform(0, span, (n, 3)); s = rs.uniform(0.5, smax, (n, 3))
    b = np.hstack((c - s / 2, c + s / 2))
    if grid:                                     # coarse grid -> many exact coordinate / IoU ties
        b = np.round(b / grid) * grid
    return b


def f_nms(rs):
    n = int(rs.choice([1, 2, 3, 63, 64, 65, 127, 300, 1000, 2500]))
    if rs.rand() < 0.01:                         # round 6: the blocked form above 16 384 rows (chunk edges at multiples of 1024)
        n = int(rs.choice([16385, 17408, 17409, 20000, 33000]))
    b = rand_boxes(rs, n, span=rs.choice([30, 100, 300]), grid=rs.choice([None, None, 1.0, 4.0]))
    mode = rs.randint(4)
    sc = rs.uniform(0, 1, n) if mode == 0 else np.round(rs.uniform(0, 1, n), 1) if mode == 1 else \
        np.ones(n) if mode == 2 else rs.permutation(n) / n
    if rs.rand() < 0.2:
        b[rs.randint(n)] = b[rs.randint(n)]      # exact duplicate box
    if rs.rand() < 0.2:
        b[rs.randint(n), 3:] = b[rs.randint(n), :3] - 5   # inverted box (negative extents)
    dets = np.hstack((b, sc[:, None])).astype(np.float32)
    thr = float(rs.choice([0.0, 0.15, 0.23, 0.5, 0.99, 1.0]))
    byv = bool(rs.randint(2))
    got = m3d.nms3d(dev(dets), thr, by_volume=byv).cpu().numpy()
    ref = O.nms_3d(dets, thr, by_volume=byv)
    assert np.array_equal(got, ref), ("nms", n, thr, byv, mode)


def f_overlaps(rs):
    n, k = int(rs.randint(1, 700)), int(rs.randint(1, 90))
    a = rand_boxes(rs, n, grid=rs.choice([None, 2.0])).astype(np.float32)
    q = rand_boxes(rs, k, grid=rs.choice([None, 2.0])).astype(np.float32)
    if rs.rand() < 0.3:
        q[0] = a[0]
    assert np.array_equal(m3d.bbox_overlaps3d(dev(a), dev(q)).cpu().numpy(), O.bbox_overlaps_3d(a, q)), ("overlaps", n, k)


def f_transform(rs):
    n, k = int(rs.randint(1, 1200)), int(rs.choice([1, 2]))
    b = rand_boxes(rs, n).astype(np.float32)
    d = (rs.randn(n, 6 * k) * rs.choice([0.1, 1.0, 5.0])).astype(np.float32)
    w = (1., 1., 1., 1., 1., 1.) if rs.rand() < 0.5 else (10., 10., 10., 5., 5., 5.)
    got = m3d.bbox_transform3d(dev(b), dev(d), w).cpu().numpy()
    ref = O.bbox_transform_3d(b, d, w)
    assert np.allclose(got, ref, rtol=3e-7, atol=1e-4) and (got == ref).mean() > 0.95, ("transform", n, k)
    shp = (int(rs.randint(8, 200)), int(rs.randint(8, 300)), int(rs.randint(8, 300)))
    got = m3d.bbox_transform3d(dev(b), dev(d), w, clip_to=shp).cpu().numpy()
    ref = O.clip_tiled_boxes_3d(ref, shp)
    assert np.allclose(got, ref, rtol=3e-7, atol=1e-4), ("clip", n, k)


def f_proposals(rs):
    cfg = O.Cfg() if rs.rand() < 0.5 else O.Cfg.soma()
    A = cfg.anchors.shape[0]
    S, H, W = int(rs.randint(1, 9)), int(rs.randint(1, 20)), int(rs.randint(1, 20))
    sc = rs.uniform(0, 1, (A, S, H, W)).astype(np.float32)
    mode = rs.randint(3)
    if mode == 1:
        sc[rs.uniform(0, 1, sc.shape) > 0.98] = 1.0
    elif mode == 2:
        sc = np.round(sc, 2)
    dl = (rs.randn(6 * A, S, H, W) * rs.choice([0.05, 0.3, 2.0])).astype(np.float32)
    st = cfg.stride
    info = np.array([S * st, H * st, W * st, 1.0])
    pre, post = int(rs.choice([1, 50, 1000, 6000])), int(rs.choice([1, 30, 1000]))
    thr = float(rs.choice([0.15, 0.23, 0.7]))
    r0, p0, k0 = O.generate_proposals_3d(sc, dl, info, cfg.anchors, st, pre, post, thr, 0)
    r1, p1, k1 = m3d.generate_proposals3d(dev(sc), dev(dl), cfg.anchors, float(st), info, pre, post, thr)
    assert np.array_equal(k1.cpu().numpy(), k0), ("proposals idx", A, S, H, W, pre, post, thr, mode)
    assert np.array_equal(p1.cpu().numpy(), p0)
    assert np.allclose(r1.cpu().numpy(), r0, rtol=3e-7, atol=1e-4)
    if min(pre, A * S * H * W) <= m3d.fused_max_boxes():       # the batched fused path (multi-workgroup radix select)
        B = int(rs.randint(1, 4))
        scb = np.stack([sc] + [np.roll(sc, i + 1, axis=1) for i in range(B - 1)])
        dlb = np.stack([dl] * B)
        rb, pb, kb, nb = m3d.generate_proposals3d_batched(dev(scb), dev(dlb), cfg.anchors, float(st), info, pre, post, thr)
        n0 = int(nb[0])
        assert n0 == len(k0) and np.array_equal(kb[0, :n0].cpu().numpy(), k0), ("proposals batched idx", A, S, H, W, pre, post, thr, mode)
        assert np.array_equal(pb[0, :n0].cpu().numpy(), p0.ravel())
        for b in range(1, B):
            rr, pp, kk = O.generate_proposals_3d(scb[b], dl, info, cfg.anchors, st, pre, post, thr, 0)
            n = int(nb[b])
            assert n == len(kk) and np.array_equal(kb[b, :n].cpu().numpy(), kk), ("proposals batched item", b)


def f_roialign(rs):
    B, Cc = int(rs.randint(1, 3)), int(rs.choice([1, 3, 16, 32, 64, 96]))
    S, H, W = int(rs.randint(1, 18)), int(rs.randint(1, 28)), int(rs.randint(1, 28))
    f = rs.randn(B, Cc, S, H, W).astype(np.float32)
    R = int(rs.randint(1, 60))
    scale = float(rs.choice([0.125, 0.25]))
    ext = max(S, H, W) / scale
    c = rs.uniform(-0.2 * ext, 1.2 * ext, (R, 3)); s = rs.uniform(0.5, 0.8 * ext, (R, 3))
    rois = np.hstack((rs.randint(0, B, (R, 1)), c - s / 2, c + s / 2)).astype(np.float32)
    res = int(rs.choice([1, 3, 7]))
    ratio = int(rs.choice([0, 1, 2, 3]))
    got = m3d.roi_align3d_forward(dev(f), dev(rois), res, res, res, scale, ratio, exact=True).cpu().numpy()
    ref = O.roi_align_3d_forward(f, rois, res, res, res, scale, ratio)
    assert np.array_equal(got, ref), ("roialign exact", f.shape, R, res, ratio)
    fast = m3d.roi_align3d_forward(dev(f), dev(rois), res, res, res, scale, ratio).cpu().numpy()
    assert np.abs(fast - ref).max() <= 2e-5 * max(1.0, np.abs(f).max()), ("roialign sep", f.shape, R, res, ratio, np.abs(fast - ref).max())


def f_otsu(rs):
    n = int(rs.randint(1, 12))
    imgs, prms = [], []
    for _ in range(n):
        shp = tuple(int(v) for v in rs.randint(1, 40, 3))
        kind = rs.randint(4)
        zz, yy, xx = np.mgrid[0:shp[0], 0:shp[1], 0:shp[2]]
        r = np.sqrt((zz - shp[0] / 2) ** 2 + (yy - shp[1] / 2) ** 2 + (xx - shp[2] / 2) ** 2)
        img = (rs.uniform(100, 3000) * np.exp(-(r / rs.uniform(2, 12)) ** 2) + rs.uniform(0, 200) + rs.randn(*shp) * rs.uniform(0, 30)).clip(0, 65535)
        prm = (255 * np.exp(-(r / rs.uniform(2, 10)) ** 2) * (rs.rand(*shp) if kind == 1 else 1)).astype(np.uint8)
        if kind == 2:
            prm[:] = rs.randint(0, 3)
        if kind == 3:
            img = np.round(img / 50) * 50
        img = img.astype(np.uint16)
        if prm.max() == 0:
            prm.flat[0] = 1
        a, b = (O.normalize_soma if rs.rand() < 0.5 else O.normalize_nuclei)(img, prm)
        imgs.append(a); prms.append(b)
    offs = np.concatenate(([0], np.cumsum([a.size for a in imgs]))).astype(np.int64)
    G = 8192
    mask, kb, st = m3d.otsu2d_batch(dev(np.concatenate([a.ravel() for a in imgs])), dev(np.concatenate([b.ravel() for b in prms])), dev(offs), G)
    mask, kb, st = mask.cpu().numpy(), kb.cpu().numpy(), st.cpu().numpy()
    for i in range(n):
        try:
            m, k, b = O.otsu_py_2d_fast(imgs[i], prms[i])
        except ValueError:
            assert st[i] != 0, ("otsu status", i)
            continue
        assert st[i] == 0 and (k, b) == tuple(kb[i]) and np.array_equal(mask[offs[i]:offs[i + 1]].reshape(m.shape), m), ("otsu", imgs[i].shape, k, b, kb[i])


def f_cc(rs):
    n = int(rs.randint(1, 10))
    masks = []
    for _ in range(n):
        shp = tuple(int(v) for v in rs.randint(1, 34 if rs.rand() < 0.85 else 90, 3))
        p = rs.choice([0.03, 0.1, 0.2, 0.35, 0.6, 0.95])
        m = (rs.rand(*shp) < p)
        if rs.rand() < 0.3:
            m = np.kron(m[::2, ::2, ::2], np.ones((2, 2, 2), bool))[:shp[0], :shp[1], :shp[2]] if min(shp) > 1 else m
        masks.append(np.ascontiguousarray(m).astype(np.uint8) * 255)
    offs = np.concatenate(([0], np.cumsum([m.size for m in masks]))).astype(np.int64)
    dims = np.array([m.shape for m in masks], np.int32)
    flat = dev(np.concatenate([m.ravel() for m in masks]))
    for tie_last, ref in ((True, O.largest_cc_soma), (False, O.largest_cc_nuclei)):
        out, st = m3d.cc_largest_batch(flat, dev(offs), dev(dims), invert=False, tie_last=tie_last)
        out, st = out.cpu().numpy(), st.cpu().numpy()
        for i, m in enumerate(masks):
            if not m.any():
                assert st[i] == 1; continue
            assert np.array_equal(out[offs[i]:offs[i + 1]].reshape(m.shape), ref(m).astype(np.uint8) * 255), ("cc", m.shape, tie_last)
    cc, st = m3d.cc_largest_batch(flat, dev(offs), dev(dims), invert=False, tie_last=False)
    fl, st2 = m3d.cc_largest_batch(cc, dev(offs), dev(dims), invert=True, tie_last=False)
    cl = m3d.binary_closing6_batch(fl, dev(offs), dev(dims)).cpu().numpy()
    st, st2 = st.cpu().numpy(), st2.cpu().numpy()
    for i, m in enumerate(masks):
        if st[i] != 0 or st2[i] != 0:
            continue
        ref = O.fill_and_close_nuclei(O.largest_cc_nuclei(m))
        assert np.array_equal(cl[offs[i]:offs[i + 1]].reshape(m.shape), ref.astype(np.uint8) * 255), ("fill+close", m.shape)


def f_conv(rs):
    k = int(rs.choice([1, 3, 3, 3, 5]))
    cin = 1 if k == 5 else int(rs.choice([1, 3, 16, 32, 33, 64, 100, 128]))
    cout = int(rs.choice([1, 5, 32, 33, 64, 96, 128])) if k != 5 else int(rs.choice([8, 32, 48]))
    B = int(rs.choice([1, 1, 2]))
    D, H, W = int(rs.randint(1, 12)), int(rs.randint(1, 24)), int(rs.randint(1, 140 if k in (3, 5) else 70))
    x = torch.from_numpy(rs.randn(B, cin, D, H, W).astype(np.float32))
    w = torch.from_numpy((rs.randn(cout, cin, k, k, k) * (2.0 / (cin * k ** 3)) ** 0.5).astype(np.float32))
    ref = torch.nn.functional.conv3d(x.double(), w.double(), None, 1, k // 2)
    y = m3d.PackedConv3d(w.cuda())(x.cuda()).cpu().double()
    den = max(ref.abs().max().item(), 1e-3)
    assert (y - ref).abs().max().item() / den < 1e-5, ("conv fwd", B, cin, cout, D, H, W, k)
    if k != 5:
        gy = torch.from_numpy(rs.randn(B, cout, D, H, W).astype(np.float32))
        refx = torch.nn.grad.conv3d_input(x.shape, w.double(), gy.double(), 1, k // 2)
        gx = m3d.PackedConv3d(w.cuda(), mode=m3d.W_DGRAD)(gy.cuda()).cpu().double()
        assert (gx - refx).abs().max().item() / max(refx.abs().max().item(), 1e-3) < 1e-5, ("conv dgrad", B, cin, cout, D, H, W, k)
    if k == 5 and W >= 32:                         # Winograd F(2,5) stem (+ fused pool)
        sc = torch.from_numpy((rs.rand(cout) + 0.5).astype(np.float32)); sh = torch.from_numpy(rs.randn(cout).astype(np.float32))
        sw = m3d.StemWinoConv3d(w.cuda())
        r2 = torch.relu(ref * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1))
        ys = sw(x.cuda(), scale=sc.cuda(), shift=sh.cuda(), relu=True).cpu().double()
        assert (ys - r2).abs().max().item() / max(r2.abs().max().item(), 1e-3) < 2e-5, ("stem wino", B, cout, D, H, W)
        if D >= 2 and H >= 2:
            yp = sw.pooled(x.cuda(), scale=sc.cuda(), shift=sh.cuda(), relu=True).cpu().double()
            rp = torch.nn.functional.max_pool3d(r2, 2, 2)
            assert (yp - rp).abs().max().item() / max(rp.abs().max().item(), 1e-3) < 2e-5, ("stem wino pool", B, cout, D, H, W)
    two_d = bool(rs.randint(2))
    if k == 3 and W >= (12 if two_d else 24):      # Winograd forward (split-K path below 24 wide; fused pool when it applies)
        sc = torch.from_numpy((rs.rand(cout) + 0.5).astype(np.float32)); sh = torch.from_numpy(rs.randn(cout).astype(np.float32))
        local = two_d and bool(rs.randint(2))          # the exactly-local F(2x2,3x3) family (PRM strips) or the default F(2x4,3x3)
        wc = m3d.WinoConv3d(w.cuda(), two_d=two_d, local=local)
        tol = 2e-5 if (two_d and not local) else 1e-5  # F(4,3) along x: about twice F(2,3)'s error
        r2 = torch.relu(ref * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1))
        yw = wc(x.cuda(), scale=sc.cuda(), shift=sh.cuda(), relu=True).cpu().double()
        assert (yw - r2).abs().max().item() / max(r2.abs().max().item(), 1e-3) < tol, ("conv wino", two_d, local, B, cin, cout, D, H, W)
        if wc.supports_pool(W) and D >= 2 and H >= 2:
            rp = torch.nn.functional.max_pool3d(r2, 2, 2)
            if two_d and rs.rand() < 0.5:              # fused pool + arg-max (PRM forward): the indexed values are the pooled values
                yp, am = wc.pooled(x.cuda(), scale=sc.cuda(), shift=sh.cuda(), relu=True, return_argmax=True)
                yp = yp.cpu().double()
                assert int(am.max()) <= 7
            else:
                yp = wc.pooled(x.cuda(), scale=sc.cuda(), shift=sh.cuda(), relu=True).cpu().double()
            assert (yp - rp).abs().max().item() / max(rp.abs().max().item(), 1e-3) < 2e-5, ("conv wino pool", two_d, B, cin, cout, D, H, W)
    gy = torch.from_numpy(rs.randn(B, cout, D, H, W).astype(np.float32))
    refw = torch.nn.grad.conv3d_weight(x.double(), w.shape, gy.double(), 1, k // 2)
    gw = m3d.conv3d_wgrad(x.cuda(), gy.cuda(), k).cpu().double()
    # a weight gradient is a sum over all voxels: scale the tolerance by the magnitude of the summed terms, not by a
    # result that may cancel to ~0
    terms = torch.nn.grad.conv3d_weight(x.double().abs(), w.shape, gy.double().abs(), 1, k // 2).max().item()
    assert (gw - refw).abs().max().item() / max(terms, 1e-3) < 2e-6, ("conv wgrad", B, cin, cout, D, H, W, k)


def f_linear(rs):
    """Both box-head GEMMs against fp64: the fp32-input MFMA kernel and the bf16x3 split (exact 3-way cut, six MFMAs per product);
    ragged M / N, many K chunks and slices, operand magnitudes spread over 10^+-12 per row (the cut must stay exact)."""
    M = int(rs.choice([1, 2, 31, 64, 65, 127, 128, 129, 255, 256, 257, 300, 700]))
    N = int(rs.choice([64, 65, 100, 128, 192, 1024]))
    K = 32 * int(rs.choice([1, 2, 3, 7, 33, 172, 500]))
    x = rs.randn(M, K).astype(np.float32) * np.float32(10.0) ** rs.randint(-12, 13, (M, 1)).astype(np.float32)
    w = (rs.randn(N, K) / np.sqrt(K)).astype(np.float32) * np.float32(10.0) ** rs.randint(-3, 4, (N, 1)).astype(np.float32)
    b = rs.randn(N).astype(np.float32)
    relu = bool(rs.randint(2))
    xd, wd, bd = dev(x), dev(w), dev(b)
    ref = xd.double() @ wd.double().t() + bd.double()
    if relu:
        ref = torch.relu(ref)
    # per-row scale: rows differ by 24 orders of magnitude
    rows = (xd.double().abs() @ wd.double().abs().t()).max(dim=1, keepdim=True).values + bd.double().abs().max()
    tol = 2e-6 * max(1.0, np.sqrt(K / 1024.0))
    y32 = m3d.linear(xd, wd, bd, relu=relu)
    y3 = m3d.SplitLinear(wd, bd)(xd, relu=relu)
    assert ((y32.double() - ref).abs() / rows).max().item() < tol, ("linear fp32", M, N, K)
    assert ((y3.double() - ref).abs() / rows).max().item() < tol, ("linear bf16x3", M, N, K)
    # round 6: the f16x2 split (two scaled fp16 pieces per operand, one scale per TENSOR): operands of bounded dynamic range - ReLU-like
    # x whose rows differ by 10^+-2, weights whose rows differ by 10^+-1 - exact bound, a 100 x loose bound, and the bound swept inside
    x2 = np.maximum(rs.randn(M, K), 0).astype(np.float32) * np.float32(10.0) ** rs.uniform(-2, 2, (M, 1)).astype(np.float32)
    w2 = (rs.randn(N, K) / np.sqrt(K)).astype(np.float32) * np.float32(10.0) ** rs.uniform(-1, 1, (N, 1)).astype(np.float32)
    xd2, wd2 = dev(x2), dev(w2)
    ref2 = xd2.double() @ wd2.double().t() + bd.double()
    if relu:
        ref2 = torch.relu(ref2)
    rows2 = (xd2.double().abs() @ wd2.double().abs().t()).max(dim=1, keepdim=True).values + bd.double().abs().max()
    lin16 = m3d.ops.SplitLinearF16(wd2, bd)
    xb = m3d.ops.absmax(xd2)
    for bound in (xb, xb * 100.0, None):
        y16 = lin16(xd2, relu=relu, x_bound=bound)
        assert ((y16.double() - ref2).abs() / rows2).max().item() < tol, ("linear f16x2", M, N, K)


def f_mask_paste(rs):
    """segm_results' paste (csrc/mask_paste.hip) against the oracle's scipy.ndimage restatement of skimage's resize: identical masks
    except voxels whose soft value is within 2e-6 of the threshold."""
    from m3d.mask_head import segm_results
    M = int(rs.choice([4, 7, 14, 28]))
    C = int(rs.choice([2, 3]))
    shape = tuple(int(v) for v in rs.randint(6, 48, 3))
    R = int(rs.randint(1, 12))
    masks = np.clip(rs.rand(R, C, M, M, M).astype(np.float32) * rs.choice([0.6, 1.0, 1.6]), 0, 1)
    ctr = rs.uniform(0, max(shape), (R, 3)); size = rs.choice([0.3, 1.0, 3.0, 6.0, 11.0, 20.0, 33.0, 60.0], (R, 3))
    lim = np.array([shape[2] - 1, shape[1] - 1, shape[0] - 1] * 2, np.float64)
    boxes = np.clip(np.hstack([ctr - size / 2, ctr + size / 2]), 0, lim)
    n1 = int(rs.randint(0, R + 1)) if C > 2 else R
    cls_boxes = [np.zeros((0, 7)), np.zeros((n1, 7))] + ([np.zeros((R - n1, 7))] if C > 2 else [])
    spec = bool(rs.randint(2))
    thr = float(rs.choice([0.5, 0.3, 0.7]))
    got = segm_results(cls_boxes, masks, boxes, *shape, num_classes=C, resolution=M, cls_specific=spec, thresh=thr)
    ref = O.segm_results(cls_boxes, masks, boxes, *shape, num_classes=C, resolution=M, cls_specific=spec, thresh=thr)
    rb = O.expand_boxes(boxes, (M + 2.0) / M).astype(np.int32)
    ind = 0
    for j in range(1, C):
        for g, r in zip(got[j], ref[j]):
            if not np.array_equal(g, r):
                b = rb[ind]
                pad = np.zeros((M + 2,) * 3, np.float32); pad[1:-1, 1:-1, 1:-1] = masks[ind, j if spec else 0]
                soft = O.skimage_resize_nd(pad, (max(b[5] - b[2] + 1, 1), max(b[4] - b[1] + 1, 1), max(b[3] - b[0] + 1, 1)))
                zz, yy, xx = np.nonzero(g != r)
                assert (np.abs(soft[zz - b[2], yy - b[1], xx - b[0]] - np.float32(thr)) < 2e-6).all(), ("mask paste", M, shape, ind)
            ind += 1


_prm_cache = {}


def f_prm(rs):
    """Whole PRM tile (forward with hooks, detections, batched cone-cropped back-propagation) on a small random volume
    against the oracle's autograd-free restatement (oracle.prm_tile)."""
    from m3d.model import DetectorM3D
    from m3d.prm import PRMEngine
    soma = bool(rs.randint(2))
    key = (soma, int(rs.randint(3)))
    if key not in _prm_cache:
        P = O.make_params(stride=4 if soma else 8, num_anchors=14 if soma else 35, mlp_dim=32, seed=key[1])
        cfg = O.Cfg.soma(mlp_dim=32) if soma else O.Cfg(mlp_dim=32, score_thresh=0.0)
        _prm_cache[key] = (P, cfg, PRMEngine(DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg), peak_chunk=int(rs.choice([0, 3, 7]))))
    P, cfg, eng = _prm_cache[key]
    st = cfg.stride
    S, H, W = [int(st * rs.randint(1, 4)) for _ in range(3)]
    vol = torch.from_numpy(rs.randn(1, 1, S, H, W).astype(np.float32))
    ref = O.prm_tile(P, cfg, vol)
    out = eng.prm_tile(vol.cuda())
    if ref is None or ref[1] is None:
        assert out is None or out[0] is None if isinstance(out, tuple) else (out is None or out.get("dets") is None), ("prm none", S, H, W)
        return
    crm, peaks, prms, dets = ref[0], ref[1], ref[2], ref[3]
    assert np.array_equal(out["peaks"].cpu().numpy(), np.asarray(peaks)), ("prm peaks", S, H, W, soma)
    assert np.allclose(out["dets"].cpu().numpy(), np.asarray(dets), rtol=1e-4, atol=1e-3)
    pr = np.asarray(prms)
    got = out["prms"].cpu().numpy()
    # a peak whose gradient dies on the way down has an all-zero map; prm / prm.sum() is then NaN in the reference too
    # (peak_response_mapping_3d.py:171): the NaN pattern must be identical
    assert np.array_equal(np.isnan(got), np.isnan(pr)), ("prm NaN pattern", S, H, W, soma)
    scale = np.nanmax(pr) if np.isfinite(pr).any() else 1.0
    assert np.allclose(got, pr, rtol=5e-3, atol=5e-6 * scale, equal_nan=True), ("prm maps", S, H, W, soma)


def f_quant_segment(rs):
    """uint8 maps from the cone-cropped windows == from the dense maps; segment_tile fed either way gives the same labels."""
    from m3d import binarize
    D, H, W = int(rs.randint(4, 30)), int(rs.randint(4, 40)), int(rs.randint(4, 40))
    Wn = int(rs.choice([4, 8, 12, 40]))
    P = int(rs.randint(1, 7))
    win = (rs.rand(P, Wn, Wn, Wn) * (rs.rand(P, Wn, Wn, Wn) > rs.choice([0.0, 0.3, 0.9]))).astype(np.float32)
    if rs.rand() < 0.3:
        win[0] = 0
    org = np.stack([rs.randint(-Wn + 1, D, P), rs.randint(-Wn + 1, H, P), rs.randint(-Wn + 1, W, P)], 1).astype(np.int32)
    if rs.rand() < 0.3:
        org[-1] = -1                                 # may cover the whole tile
    sums = np.maximum(win.reshape(P, -1).sum(1), 1e-6).astype(np.float32)
    w_, s_, o_ = dev(win), dev(sums), dev(org)
    dense = m3d.prm_scatter(w_, s_, o_, (D, H, W))
    assert torch.equal(m3d.prm_quantize_windows_u8(w_, s_, o_, (D, H, W)), m3d.prm_quantize_u8(dense)), ("quantise", D, H, W, Wn)
    img = dev(rs.randint(0, 3000, (D, H, W)).astype(np.uint16))
    c = np.stack([rs.uniform(0, W, P), rs.uniform(0, H, P), rs.uniform(0, D, P)], 1); e = rs.uniform(2, 24, (P, 3))
    dets = np.hstack((c - e / 2, c + e / 2, rs.uniform(0.4, 1, (P, 1))))
    mode = "soma" if rs.rand() < 0.5 else "nuclei"
    l0, p0 = binarize.segment_tile(img, dense, dets, mode=mode)
    l1, p1 = binarize.segment_tile(img, (w_, s_, o_), dets, mode=mode)
    assert torch.equal(l0, l1) and torch.equal(p0, p1), ("segment", D, H, W, Wn, mode)


def f_segment_oracle(rs):
    """binarize.segment_tile (device: quantised windows -> crops -> 2D-Otsu -> components -> painting) against the oracle's loop body of
    tools/binarization_soma.py:65-105 / binarization_nuclei.py:92-150 on the same uint8 maps and integer boxes: labels and painted flags
    identical."""
    from m3d import binarize
    D, H, W = int(rs.randint(6, 24)), int(rs.randint(8, 40)), int(rs.randint(8, 40))
    Wn = int(rs.choice([8, 12, 20]))
    P = int(rs.randint(1, 6))
    win = (rs.rand(P, Wn, Wn, Wn) ** 2 * (rs.rand(P, Wn, Wn, Wn) > rs.choice([0.0, 0.3, 0.8]))).astype(np.float32)
    if rs.rand() < 0.25:
        win[rs.randint(P)] = 0
    org = np.stack([rs.randint(-Wn // 2, D - 2, P), rs.randint(-Wn // 2, H - 2, P), rs.randint(-Wn // 2, W - 2, P)], 1).astype(np.int32)
    sums = np.maximum(win.reshape(P, -1).sum(1), 1e-6).astype(np.float32)
    w_, s_, o_ = dev(win), dev(sums), dev(org)
    shape = (D, H, W)
    q = m3d.prm_quantize_windows_u8(w_, s_, o_, shape).cpu().numpy()
    img_np = (rs.randint(0, 3000, shape) * (rs.rand(*shape) > 0.1)).astype(np.uint16)
    c = np.stack([rs.uniform(0, W, P), rs.uniform(0, H, P), rs.uniform(0, D, P)], 1); e = rs.uniform(2, 20, (P, 3))
    dets = np.hstack((c - e / 2, c + e / 2, rs.uniform(0.4, 1, (P, 1))))
    mode = "soma" if rs.rand() < 0.5 else "nuclei"
    boxes = binarize.det_boxes_int(dets, shape, mode)
    okb = (boxes[:, 3] >= boxes[:, 0]) & (boxes[:, 4] >= boxes[:, 1]) & (boxes[:, 5] >= boxes[:, 2]) & (boxes[:, :3].min(1) >= 0) & \
          (boxes[:, 3] < W) & (boxes[:, 4] < H) & (boxes[:, 5] < D)
    if not okb.all():                                    # the reference's slicing of an inverted / outside box is a different code path: boxes inside only
        return
    l1, p1 = binarize.segment_tile(dev(img_np), (w_, s_, o_), dets, mode=mode)
    seg, painted = O.segment_tile(img_np, q, boxes, mode)
    got = l1.cpu().numpy()
    got = np.where(got < 0, 0, got).astype(np.uint16)
    assert np.array_equal(got, seg), ("segment labels", shape, Wn, mode, int((got != seg).sum()))
    assert np.array_equal(p1.cpu().numpy().astype(bool), painted), ("painted", shape, mode)


def f_zwconv(rs):
    """m3d_conv3d_zw_forward (f16x2 split + Winograd F(2,3) along z on the f16 matrix cores) against float64: conv + scale/shift + ReLU
    [+ MaxPool3d(2,2)], ragged shapes, both column-block widths, batches, loose / exact operand bounds, and a two-layer chain whose second
    layer takes the bound the first one's epilogue left (never a sweep)."""
    from m3d import ops as mops
    cin = int(16 * rs.randint(1, 9)); cout = int(rs.choice([8, 32, 64, 96, 128, 200]))
    D, H, W = int(rs.randint(2, 12)), int(rs.randint(4, 20)), int(rs.randint(12, 70))
    B = int(rs.randint(1, 3))
    mag = float(10.0 ** rs.uniform(-6, 6))                                  # the scale follows the data: any magnitude
    x = (np.maximum(rs.randn(B, cin, D, H, W), 0) * mag).astype(np.float32) if rs.rand() < 0.7 else (rs.randn(B, cin, D, H, W) * mag).astype(np.float32)
    w = (rs.randn(cout, cin, 3, 3, 3) * (2.0 / (cin * 27)) ** 0.5).astype(np.float32)
    sc = (rs.rand(cout) + 0.5).astype(np.float32); sh = (rs.randn(cout) * mag).astype(np.float32)
    xt, wt = torch.from_numpy(x), torch.from_numpy(w)
    conv = mops.ZwConv3d(wt.cuda())
    if not conv.supports((D, H, W)):
        return
    bound = mops.ZwConv3d.bound_of(xt.cuda()) * float(rs.choice([1.0, 1.0, 3.0, 100.0]))
    ref = torch.relu(torch.nn.functional.conv3d(xt.double(), wt.double(), padding=1) * torch.from_numpy(sc).double().view(1, -1, 1, 1, 1)
                     + torch.from_numpy(sh).double().view(1, -1, 1, 1, 1))
    pool = bool(rs.randint(2)) and conv.supports((D, H, W), pool=True)
    got, gm = conv(xt.cuda(), bound, scale=torch.from_numpy(sc).cuda(), shift=torch.from_numpy(sh).cuda(), relu=True, pool=pool)
    want = torch.nn.functional.max_pool3d(ref, 2, 2) if pool else ref
    scale = float(want.abs().max()) or 1.0
    assert got.shape == want.shape and float((got.cpu().double() - want).abs().max()) <= 5e-6 * max(scale, float(ref.abs().max())), \
        ("zw conv error", cin, cout, (B, D, H, W), pool, float((got.cpu().double() - want).abs().max()) / scale)
    assert float(gm.max()) == float(got.abs().max()), ("zw conv bound", cin, cout, (B, D, H, W), pool)
    if cout % 16 == 0 and conv.supports(tuple(got.shape[-3:])) and mops.ZwConv3d.supported(torch.empty(8, cout, 3, 3, 3)):
        w2 = (rs.randn(24, cout, 3, 3, 3) * (2.0 / (cout * 27)) ** 0.5).astype(np.float32)
        c2 = mops.ZwConv3d(torch.from_numpy(w2).cuda())
        if c2.supports(tuple(got.shape[-3:])):
            y2, _ = c2(got, gm)
            r2 = torch.nn.functional.conv3d(want, torch.from_numpy(w2).double(), padding=1)
            assert float((y2.cpu().double() - r2).abs().max()) <= 1e-5 * (float(r2.abs().max()) or 1.0), ("zw conv chain", cin, cout, (B, D, H, W), pool)


def f_x3conv(rs):
    """m3d_conv3d_x3_forward (bf16x3 cut on the bf16 matrix cores) as conv3d(x - min x, relu(W), padding 1) against float64: fp32-level
    error, exact zeros where the inputs under every positive weight are zero; ragged shapes, odd chunk counts, batches."""
    from m3d import ops as mops
    cin = int(16 * rs.randint(1, 9)); cout = int(rs.choice([8, 32, 64, 96, 128, 200]))
    D, H, W = int(rs.randint(1, 10)), int(rs.randint(1, 14)), int(rs.randint(1, 40))
    B = int(rs.randint(1, 3))
    x = (rs.rand(B, cin, D, H, W) * (rs.rand(B, cin, D, H, W) > rs.choice([0.0, 0.5, 0.95]))).astype(np.float32) + 0.5
    if rs.rand() < 0.5:
        x[:, :, : max(1, D // 2)] = 0.5
    w = (rs.randn(cout, cin, 3, 3, 3) * 0.1).astype(np.float32)
    xt, wt = torch.from_numpy(x), torch.from_numpy(w)
    xc, wc = xt.cuda(), wt.cuda()
    off = mops.reduce_min(xc)
    got = mops.X3Conv3d(wc, mops.W_RELU)(xc, in_offset=off).cpu().double()
    ref = torch.nn.functional.conv3d((xt - float(off)).double(), torch.relu(wt).double(), padding=1)
    scale = float(ref.abs().max()) or 1.0
    assert float((got - ref).abs().max()) <= 4e-6 * scale, ("x3 conv error", cin, cout, (D, H, W), float((got - ref).abs().max()) / scale)
    assert torch.equal(got == 0, ref == 0), ("x3 conv zeros", cin, cout, (D, H, W))


ops = [("quantise/segment", f_quant_segment), ("segment_tile vs oracle", f_segment_oracle), ("conv3d bf16x3", f_x3conv), ("conv3d f16x2 F(2,3)z", f_zwconv), ("nms3d", f_nms), ("bbox_overlaps3d", f_overlaps), ("bbox_transform3d", f_transform), ("generate_proposals3d", f_proposals),
       ("roi_align3d", f_roialign), ("otsu2d", f_otsu), ("cc/closing", f_cc), ("conv3d fwd/dgrad/wgrad/winograd", f_conv), ("linear fp32 / bf16x3 / f16x2", f_linear), ("mask paste", f_mask_paste), ("prm tile", f_prm)]
only = os.environ.get("FUZZ_ONLY")
if only:
    ops = [o for o in ops if any(t in o[0] for t in only.split(","))]
bad = 0
for name, fn in ops:
    t0, n, fails = time.time(), 0, 0
    while time.time() - t0 < budget:
        rs = np.random.RandomState(seed0 * 1000003 + n)
        try:
            fn(rs)
        except AssertionError as e:
            fails += 1
            if fails <= 3:
                print("  MISMATCH %s case %d: %s" % (name, n, e)); traceback.print_exc(limit=1)
        n += 1
    print("%-28s %5d cases, %d mismatches" % (name, n, fails), flush=True)
    bad += fails
sys.exit(1 if bad else 0)
