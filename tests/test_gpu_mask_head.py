"""GPU: mask branch (dilated convs, ConvTranspose3d as 1x1x1 conv + voxel shuffle, classifier) against torch fp64 on the CPU.
Reference: lib/modeling/mask_rcnn_heads.py:132-193,20-68; lib/core/test.py:439-476.  Tolerance: fp32 MFMA accumulation vs fp64,
rtol 1e-4 of the layer's largest magnitude."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle as O

pytestmark = pytest.mark.gpu


def _mask_params(cin, dim, nc, convs, seed):
    g = torch.Generator().manual_seed(seed)
    P = {}
    c = cin
    for i in range(convs):
        P["Mask_Head.conv_fcn.%d.weight" % (2 * i)] = torch.randn((dim, c, 3, 3, 3), generator=g) * (2.0 / (27 * c)) ** 0.5
        P["Mask_Head.conv_fcn.%d.bias" % (2 * i)] = torch.randn((dim,), generator=g) * 0.1
        c = dim
    P["Mask_Head.upconv.weight"] = torch.randn((dim, dim, 2, 2, 2), generator=g) * (2.0 / dim) ** 0.5
    P["Mask_Head.upconv.bias"] = torch.randn((dim,), generator=g) * 0.1
    P["Mask_Outs.classify.weight"] = torch.randn((nc, dim, 1, 1, 1), generator=g) * (1.0 / dim) ** 0.5
    P["Mask_Outs.classify.bias"] = torch.randn((nc,), generator=g) * 0.1
    return P


def _ref_from_pooled(P, x, convs, dil):
    x = x.double()
    for i in range(convs):
        x = F.relu(F.conv3d(x, P["Mask_Head.conv_fcn.%d.weight" % (2 * i)].double(), P["Mask_Head.conv_fcn.%d.bias" % (2 * i)].double(),
                            padding=dil, dilation=dil))
    x = F.relu(F.conv_transpose3d(x, P["Mask_Head.upconv.weight"].double(), P["Mask_Head.upconv.bias"].double(), stride=2))
    y = F.conv3d(x, P["Mask_Outs.classify.weight"].double(), P["Mask_Outs.classify.bias"].double())
    return x, torch.sigmoid(y)


@pytest.mark.parametrize("dil,res,convs", [(2, 7, 4), (2, 14, 2), (1, 7, 3), (2, 5, 1)])
def test_dilated_conv_stack_and_upconv(dil, res, convs):
    import m3d
    from m3d.mask_head import MaskHeadM3D
    cin, dim, nc, R = 48, 40, 2, 5
    P = _mask_params(cin, dim, nc, convs, seed=res * 10 + dil)
    head = MaskHeadM3D({k: v.cuda() for k, v in P.items()}, O.Cfg(mlp_dim=64), roi_res=res, dilation=dil, resolution=2 * res)
    feat = torch.randn((1, cin, 8, 9, 10), generator=torch.Generator().manual_seed(3))
    rois = torch.tensor([[0, 3, 2, 1, 60, 50, 40], [0, 0, 0, 0, 79, 71, 63], [0, 10, 10, 10, 14, 13, 12],
                         [0, 30.5, 20.25, 8, 70, 69, 50], [0, 5, 40, 30, 25, 60, 62]], dtype=torch.float32)
    pooled = m3d.roi_align3d_forward(feat.cuda(), rois.cuda(), res, res, res, 1.0 / 8, 0)
    up_ref, prob_ref = _ref_from_pooled(P, pooled.cpu(), convs, dil)
    up = head.head(feat.cuda(), rois.cuda())
    prob = head.outputs(up)
    assert tuple(up.shape) == (R, dim, 2 * res, 2 * res, 2 * res) and tuple(prob.shape) == (R, nc, 2 * res, 2 * res, 2 * res)
    assert np.allclose(up.cpu().numpy(), up_ref.numpy(), rtol=1e-4, atol=1e-4 * float(up_ref.abs().max()))
    assert np.allclose(prob.cpu().numpy(), prob_ref.numpy(), rtol=1e-4, atol=1e-5)
    out = head.mask_net(feat.cuda(), {"mask_rois": rois.numpy()})
    assert torch.equal(out, prob)


def test_dilated_conv_rejects_what_it_does_not_implement():
    import m3d
    from m3d._lib import M3DError
    w = torch.randn((8, 8, 3, 3, 3)).cuda()
    conv = m3d.PackedConv3d(w)
    x = torch.randn((1, 8, 7, 7, 7)).cuda()
    with pytest.raises(M3DError):
        conv(x, dilation=3)
    y = conv(x, dilation=1)
    assert torch.equal(y, conv(x))


def test_im_detect_mask_shapes_and_empty():
    from m3d.mask_head import MaskHeadM3D, im_detect_mask
    cfg = O.Cfg(mlp_dim=64)
    P = _mask_params(32, 16, cfg.num_classes, 2, seed=1)
    head = MaskHeadM3D({k: v.cuda() for k, v in P.items()}, cfg)
    feat = torch.randn((1, 32, 8, 8, 8)).cuda()
    empty = im_detect_mask(head, [1.0], np.zeros((0, 6), np.float32), feat)
    assert empty.shape == (0, 14, 14, 14) and empty.dtype == np.float32            # core/test.py:457-459
    boxes = np.array([[2, 3, 4, 40, 41, 42], [10, 0, 5, 63, 63, 30], [20, 20, 20, 29, 31, 33]], np.float32)
    m = im_detect_mask(head, [1.0], boxes, feat)
    assert m.shape == (3, cfg.num_classes, 14, 14, 14) and m.dtype == np.float32
    assert (m > 0).all() and (m < 1).all()
    one = im_detect_mask(head, [1.0], boxes[:1], feat)                              # squeeze() then reshape (:469-474)
    assert one.shape == (1, cfg.num_classes, 14, 14, 14) and np.allclose(one[0], m[0], rtol=1e-5, atol=1e-6)


def test_driver_mask_net_and_im_detect_mask():
    """model.module.mask_net(blob_conv, {'mask_rois': ...}) and im_detect_mask(model, im_scale, boxes, blob_conv) as the
    reference's drivers call them (core/test.py:165-168,468)."""
    from m3d.drivers import Generalized_RCNN, im_detect_mask
    cfg = O.Cfg(mlp_dim=64)
    P = O.make_params(stride=8, num_anchors=cfg.anchors.shape[0], mlp_dim=64, seed=5)
    plain = Generalized_RCNN(P, cfg)
    with pytest.raises(AttributeError):
        plain.mask_net(None, {})
    body_dim = P["Conv_Body.conv4b.weight"].shape[0] if "Conv_Body.conv4b.weight" in P else 256
    P = dict(P)
    P.update(_mask_params(body_dim, 32, cfg.num_classes, 2, seed=9))
    model = Generalized_RCNN(P, cfg)
    vol = torch.randn((1, 1, 64, 64, 64), generator=torch.Generator().manual_seed(2))
    out = model(data=[vol], im_info=[torch.tensor([[64, 64, 64, 1.0]])])
    boxes = np.array([[4, 4, 4, 40, 44, 48], [0, 10, 20, 63, 50, 60]], np.float32)
    masks = im_detect_mask(model, [1.0], boxes, out["blob_conv"])
    assert masks.shape == (2, cfg.num_classes, 14, 14, 14)
    rois = np.hstack([np.zeros((2, 1), np.float32), boxes])
    direct = model.module.mask_net(out["blob_conv"], {"mask_rois": rois}).cpu().numpy()
    assert np.array_equal(direct, masks)


def _paste_case(rs, R, M, C, shape):
    S, H, W = shape
    masks = rs.rand(R, C, M, M, M).astype(np.float32)
    masks[:, :, M // 4: 3 * M // 4, M // 4: 3 * M // 4, M // 4: 3 * M // 4] += 0.5       # a blob above the threshold
    masks = np.clip(masks, 0, 1)
    ctr = rs.uniform(-4, max(shape) + 4, (R, 3))
    size = rs.choice([0.4, 2.0, 5.0, 9.0, 14.0, 27.0, 30.0, 31.0, 45.0, 90.0], (R, 3))      # down to 1 voxel, up to 3x the block
    boxes = np.hstack([ctr - size / 2, ctr + size / 2])
    lim = np.array([W - 1, H - 1, S - 1] * 2, np.float64)
    boxes = np.clip(boxes, 0, lim)          # detections are clipped to the image (core/test.py:247); only the expansion leaves it
    return masks, boxes


@pytest.mark.parametrize("M,C,shape,R", [(14, 2, (40, 48, 56), 48), (28, 2, (64, 64, 64), 24), (6, 3, (17, 9, 33), 40)])
def test_segm_results_paste_equals_the_scipy_restatement(M, C, shape, R):
    """segm_results (lib/core/test.py:886-945) on the device == the oracle (skimage's n-D resize restated on scipy.ndimage): identical
    uint8 masks except voxels whose resized value lies within 2e-6 of the threshold; boxes that leave the volume after the (M+2)/M expansion,
    1-voxel boxes (sigma 14.5, radius 58 > the block), boxes 3x the block (pure interpolation), class-specific channels."""
    from m3d.mask_head import segm_results
    rs = np.random.RandomState(M * 100 + R)
    masks, boxes = _paste_case(rs, R, M, C, shape)
    n1 = R // 2 if C > 2 else R
    cls_boxes = [np.zeros((0, 7))] + [np.zeros((n1, 7))] + ([np.zeros((R - n1, 7))] if C > 2 else [])
    got = segm_results(cls_boxes, masks, boxes, *shape, num_classes=C, resolution=M, cls_specific=True, thresh=0.5)
    ref = O.segm_results(cls_boxes, masks, boxes, *shape, num_classes=C, resolution=M, cls_specific=True, thresh=0.5)
    assert len(got) == C and [len(g) for g in got] == [len(r) for r in ref]
    rb = O.expand_boxes(boxes, (M + 2.0) / M).astype(np.int32)
    ind, written = 0, 0
    for j in range(1, C):
        for g, r in zip(got[j], ref[j]):
            assert g.shape == tuple(shape) and g.dtype == np.uint8
            if not np.array_equal(g, r):
                # only near-threshold voxels may differ: recompute the soft values for this detection
                b = rb[ind]
                pad = np.zeros((M + 2,) * 3, np.float32); pad[1:-1, 1:-1, 1:-1] = masks[ind, j]
                soft = O.skimage_resize_nd(pad, (max(b[5] - b[2] + 1, 1), max(b[4] - b[1] + 1, 1), max(b[3] - b[0] + 1, 1)))
                zz, yy, xx = np.nonzero(g != r)
                assert (np.abs(soft[zz - b[2], yy - b[1], xx - b[0]] - 0.5) < 2e-6).all()
            written += int(r.sum())
            ind += 1
    assert ind == R and written > 0
    # not class specific: channel 0 for every detection
    got0 = segm_results(cls_boxes, masks, boxes, *shape, num_classes=C, resolution=M, cls_specific=False, thresh=0.5)
    ref0 = O.segm_results(cls_boxes, masks, boxes, *shape, num_classes=C, resolution=M, cls_specific=False, thresh=0.5)
    diff = sum(int((a != b).sum()) for j in range(1, C) for a, b in zip(got0[j], ref0[j]))
    assert diff <= 2
    empty = segm_results([np.zeros((0, 7))] * C, np.zeros((0, C, M, M, M), np.float32), np.zeros((0, 6)), *shape, num_classes=C, resolution=M)
    assert [len(e) for e in empty] == [0] * C


def test_im_detect_all_with_a_mask_checkpoint_returns_rle_segms():
    """MODEL.MASK_ON flow of im_detect_all (lib/core/test.py:123-173): masks of every tile's kept boxes, filtered by the cross-tile
    NMS together with their boxes, pasted over the whole volume and RLE-encoded - one segm per kept detection, each decoding to a
    mask that lies inside its (M+2)/M-expanded box and equals the paste of its own soft mask."""
    from m3d.drivers import Generalized_RCNN, im_detect_all
    from m3d.io import rle_to_binary_mask
    from m3d.synth import synth_volume
    cfg = O.Cfg(mlp_dim=64)
    cfg.in_size, cfg.crop_ovlp, cfg.mask_on, cfg.score_thresh = (32, 64, 64), 16, True, 0.0
    P = dict(O.make_params(stride=8, num_anchors=cfg.anchors.shape[0], mlp_dim=64, seed=5))
    P.update(_mask_params(P["Conv_Body.conv4b.weight"].shape[0], 32, cfg.num_classes, 2, seed=9))
    model = Generalized_RCNN(P, cfg)
    im = synth_volume(3, (32, 80, 80))
    cls_boxes, cls_segms, keyps = im_detect_all(model, im)
    assert keyps is None and len(cls_segms) == cfg.num_classes and cls_segms[0] == []
    n = cls_boxes[1].shape[0]
    assert n > 0 and len(cls_segms[1]) == n
    M = model.mask_head.M
    rb = O.expand_boxes(cls_boxes[1][:, :6], (M + 2.0) / M).astype(np.int32)
    some = 0
    for i, rle in enumerate(cls_segms[1]):
        assert tuple(rle["size"]) == im.shape
        m = rle_to_binary_mask(rle)
        assert m.shape == im.shape and m.dtype == np.uint8
        zz, yy, xx = np.nonzero(m)
        if len(zz):
            some += 1
            assert xx.min() >= rb[i, 0] and xx.max() <= rb[i, 3] and yy.min() >= rb[i, 1] and yy.max() <= rb[i, 4] and zz.min() >= rb[i, 2] and zz.max() <= rb[i, 5]
    assert some > 0
    # the same checkpoint with the branch switched off in the cfg: detections unchanged, no segms (the reference's MASK_ON False path)
    cfg.mask_on = False
    cb2, cs2, _ = im_detect_all(Generalized_RCNN(P, cfg), im)
    assert np.array_equal(cb2[1], cls_boxes[1]) and cs2 == [[] for _ in range(cfg.num_classes)]
