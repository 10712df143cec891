#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE (CPU) through oracle/ref_harness.py.

Run in the build container only (needs /root/reference and oracle/_ref built by oracle/build_ref.sh):
    python tests/golden/gen_golden.py
The fixtures are data (inputs + the reference's outputs); no reference source is stored.  Weights are
not stored: they are re-created from a seed by oracle.make_params (same torch build on the GPU box).
Fixture list follows SURVEY.md 8c (1)-(10); (11) tiling: gen_tiling.py; configs[0] 64^3 run: gen_cfg0.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as H  # noqa: E402
import oracle as O  # noqa: E402

NUC = 'configs/cell_tracking_baseline/e2e_mask_rcnn_N3DH_SIM_dsn_body.yaml'
SOMA = 'configs/soma_starting/e2e_mask_rcnn_soma_dsn_body.yaml'


def save(name, **kw):
    p = os.path.join(HERE, name + ".npz")
    np.savez_compressed(p, **kw)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(p) / 1024))


def rand_boxes(rng, n, lim=(64, 200, 200), smin=2, smax=40):
    c = rng.uniform(0, 1, (n, 3)) * np.array([lim[2], lim[1], lim[0]])
    s = rng.uniform(smin, smax, (n, 3))
    b = np.hstack((c - s / 2, c + s / 2)).astype(np.float32)
    return b


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(4)

    def roi_align_plug(features, rois, AS, AH, AW, scale, ratio):
        out = O.roi_align_3d_forward(features.detach().numpy(), rois.detach().numpy(), AS, AH, AW, scale, ratio)
        return torch.from_numpy(out)

    H.install(roi_align_plug)
    cfg = H.load_cfg(NUC)
    from modeling.generate_anchors import generate_anchors_3d
    import utils.boxes_3d as B
    import core.config as CC

    # (1) anchors
    a_n = generate_anchors_3d(stride=8., sizes=cfg.RPN.SIZES, aspect_ratios=cfg.RPN.ASPECT_RATIOS)
    soma_sizes = (10, 12, 14, 16, 18, 20, 22, 24, 28, 30, 34, 36, 38, 40)
    a_s = generate_anchors_3d(stride=4., sizes=soma_sizes, aspect_ratios=[[1.0, 1.0]])
    save("anchors", nuclei=a_n, soma=a_s)

    # (2) bbox_transform_3d / clip_tiled_boxes_3d
    rng = np.random.RandomState(7)
    boxes = rand_boxes(rng, 64)
    d1 = (rng.randn(64, 6) * 0.5).astype(np.float32)
    d1[3, 3:] = 9.0  # exercises BBOX_XFORM_CLIP
    d2 = (rng.randn(64, 12) * 2.0).astype(np.float32)
    t1 = B.bbox_transform_3d(boxes.astype(np.float64), d1, (1.0,) * 6)
    t2 = B.bbox_transform_3d(boxes, d2, cfg.MODEL.BBOX_REG_WEIGHTS)
    c1 = B.clip_tiled_boxes_3d(t1.copy(), np.array([64., 200., 200.]))
    c2 = B.clip_tiled_boxes_3d(t2.copy(), np.array([64., 200., 200.]))
    save("boxes", boxes=boxes, d1=d1, d2=d2, t1=t1, t2=t2, c1=c1, c2=c2,
         w2=np.array(cfg.MODEL.BBOX_REG_WEIGHTS, np.float64))

    # (3) nms_3d / nms_3d_volume
    out = {}
    ci = 0
    for n in (1, 2, 64, 200, 1000):
        for thr in (0.15, 0.23, 0.5):
            b = rand_boxes(rng, n, lim=(64, 128, 128), smin=4, smax=48)
            if n >= 64:
                b[5] = b[4]                       # duplicate box
                b[6, 3:] = b[6, :3] - 1.0         # zero-volume (x2 = x1 - 1)
                b[7, 3] = b[7, 0] - 5.0           # inverted x extent
                b[9, :3] = b[8, :3] + 2; b[9, 3:] = b[8, 3:] - 2   # fully contained
            s = rng.permutation(n).astype(np.float32) / n + rng.uniform(0, 1e-3)   # distinct scores
            dets = np.hstack((b, s[:, None])).astype(np.float32)
            vol = (b[:, 3] - b[:, 0] + 1) * (b[:, 4] - b[:, 1] + 1) * (b[:, 5] - b[:, 2] + 1)
            assert len(np.unique(s)) == n
            out["dets%d" % ci] = dets
            out["thr%d" % ci] = np.float64(thr)
            out["keep%d" % ci] = np.asarray(B.nms_3d(dets, thr), np.int64)
            if len(np.unique(vol)) == n:
                out["keepvol%d" % ci] = np.asarray(B.nms_3d_volume(dets, thr), np.int64)
            ci += 1
    out["ncases"] = np.int64(ci)
    save("nms", **out)

    # (4) bbox_overlaps_3d 64x48
    from utils.cython_bbox_3d import bbox_overlaps_3d
    bq = rand_boxes(rng, 48, lim=(64, 128, 128))
    bb = rand_boxes(rng, 64, lim=(64, 128, 128))
    bb[0] = bq[0]
    save("overlaps", boxes=bb, query=bq, out=bbox_overlaps_3d(bb, bq))

    # (5) GenerateProposalsOp_3d
    from modeling.generate_proposals_3d import GenerateProposalsOp_3d
    res = {}
    for tag, anchors, stride, thr in (("n", a_n, 8., 0.15), ("s", a_s, 4., 0.23)):
        A = anchors.shape[0]
        S, Hh, W = 4, 6, 6
        nsc = A * S * Hh * W
        sc = (0.01 + 0.98 * (rng.permutation(nsc) + 0.5) / nsc).astype(np.float32).reshape(1, A, S, Hh, W)
        dl = (rng.randn(1, 6 * A, S, Hh, W) * 0.3).astype(np.float32)
        assert len(np.unique(sc)) == sc.size
        im_info = np.array([[S * stride, Hh * stride, W * stride, 1.0]])
        CC.cfg.TEST.RPN_NMS_THRESH = thr
        CC.cfg.TEST.RPN_PRE_NMS_TOP_N = 300
        CC.cfg.TEST.RPN_POST_NMS_TOP_N = 100
        op = GenerateProposalsOp_3d(anchors, 1. / stride).eval()
        rois, probs, keep_idx = op(torch.from_numpy(sc), torch.from_numpy(dl), torch.from_numpy(im_info))
        res.update({tag + "_scores": sc, tag + "_deltas": dl, tag + "_im_info": im_info, tag + "_rois": rois,
                    tag + "_probs": probs, tag + "_keep_idx": np.asarray(keep_idx, np.int64),
                    tag + "_thr": np.float64(thr), tag + "_stride": np.float64(stride)})
    res["pre"] = np.int64(300)
    res["post"] = np.int64(100)
    save("proposals", **res)
    CC.cfg.TEST.RPN_PRE_NMS_TOP_N = 1000
    CC.cfg.TEST.RPN_POST_NMS_TOP_N = 1000
    CC.cfg.TEST.RPN_NMS_THRESH = 0.15

    # (6) box_results_with_nms_and_limit
    from core.test import box_results_with_nms_and_limit
    n = 400
    bx = np.hstack((rand_boxes(rng, n), rand_boxes(rng, n))).astype(np.float32)
    s1 = (rng.permutation(n).astype(np.float32) + 0.5) / n
    scr = np.stack((1 - s1, s1), 1).astype(np.float32)
    kidx = rng.permutation(100000)[:n].astype(np.int64)
    CC.cfg.TEST.DETECTIONS_PER_IM = 300
    sc_o, bx_o, cls_o, keep_o = box_results_with_nms_and_limit(scr, bx, kidx)
    sc_p, bx_p, cls_p, _ = box_results_with_nms_and_limit(scr, bx)
    # NOTE: the DETECTIONS_PER_IM cap cannot be pinned: lib/core/test.py:878 raises TypeError whenever the cap
    # triggers (cls_keep_idx[j] is a python list or a 1-D array indexed with [keep, :]).
    save("box_results", scores=scr, boxes=bx, keep_idx=kidx, o_scores=sc_o, o_boxes=bx_o, o_cls1=cls_o[1],
         o_keep1=np.asarray(keep_o[1], np.int64), p_scores=sc_p, p_boxes=bx_p)

    # (7)(8)(9) small nets: body+RPN forward, PRM one-hot backward, full PRM forward tuple
    from modeling.model_builder import Generalized_RCNN
    from prm.peak_response_mapping_3d import PeakResponseMapping_3d
    for tag, yaml_rel, stride, A in (("n", NUC, 8, 35), ("s", SOMA, 4, 14)):
        H.load_cfg(yaml_rel)
        CC.cfg.FAST_RCNN.MLP_HEAD_DIM = 64
        CC.cfg.PRM_ON = True
        CC.cfg.TEST.SCORE_THRESH = 0.0
        vol = (rng.randn(1, 1, 16, 24, 24) * 1.0 + 0.2).astype(np.float32)
        P = O.make_params(stride=stride, num_anchors=A, mlp_dim=64, seed=11 + stride)
        model = PeakResponseMapping_3d()
        sd = model.state_dict()
        for k_, v_ in P.items():
            assert k_ in sd, k_
            sd[k_] = v_.clone()
        model.load_state_dict(sd)
        model.inference()
        data = torch.from_numpy(vol.copy())
        im_info = torch.from_numpy(np.array([[16., 24., 24., 1.0]]))
        # body + RPN forward (patched forward returns the same values as a plain conv)
        d0 = data.clone()
        feat = model.Conv_Body(d0)
        rp = model.RPN(feat, im_info, None)
        # (8) one-hot backward at 3 peaks (corner, edge, interior) of the class response map
        crm = rp['class_response_maps']
        d1_ = data.clone().requires_grad_()
        crm1 = model.RPN(model.Conv_Body(d1_), im_info, None)['class_response_maps']
        s_, h_, w_ = crm1.shape[-3:]
        peaks = [(0, 0, 0, 0, 0), (0, A // 2, s_ - 1, h_ // 2, 0), (0, A - 1, s_ // 2, h_ // 2, w_ // 2)]
        grads = []
        for pk in peaks:
            g = torch.zeros_like(crm1)
            g[pk] = 1.
            if d1_.grad is not None:
                d1_.grad.zero_()
            crm1.backward(g, retain_graph=True)
            grads.append(d1_.grad.detach().clone().numpy())
        # (9) full forward tuple
        agg, crm_o, vpl, prms, dets = model(data.clone(), [im_info[0]] if False else im_info, 1.0)
        save("prm_small_" + tag, vol=vol, feat=feat.detach().numpy(), crm=crm.detach().numpy(),
             rpn_deltas=rp['rpn_bbox_pred'].detach().numpy(), rois=rp['rpn_rois'], keep_idx=np.asarray(
                 rp['scores_keep_idx'], np.int64), peaks=np.array(peaks, np.int64), grads=np.stack(grads),
             o_crm=crm_o.numpy(), o_peaks=vpl.numpy(), o_prms=prms.numpy().astype(np.float32),
             o_dets=dets.numpy(), seed=np.int64(11 + stride), stride=np.int64(stride), A=np.int64(A))

    # (10) otsu_py_2d_fast on 6 crops
    import otsu as ref_otsu
    oz = {}
    zz, yy, xx = np.mgrid[0:18, 0:20, 0:22]
    for i in range(6):
        r = np.sqrt((zz - 9) ** 2 + (yy - 10) ** 2 + (xx - 11) ** 2)
        img = (600 * np.exp(-(r / (5.0 + i)) ** 2) + 100 + rng.randn(*r.shape) * 15).clip(0, 65535).astype(np.uint16)
        prm = (255 * np.exp(-(r / (4.0 + i)) ** 2) * rng.uniform(0.7, 1.0, r.shape)).astype(np.uint8)
        if i == 4:
            prm[:] = 0
            prm[9, 10, 11] = 200        # nearly flat PRM
        if i < 3 or i == 4:
            a, b = O.normalize_soma(img, prm)
        else:
            a, b = O.normalize_nuclei(img, prm)
        if i == 5:                       # G < 10
            a = (a // 64).astype(np.uint16)
            b = (b // 64).astype(np.uint16)
        m, k, bm = ref_otsu.otsu_py_2d_fast(a, b)
        oz["img%d" % i], oz["prm%d" % i], oz["mask%d" % i] = a, b, m
        oz["kb%d" % i] = np.array([k, bm], np.int64)
    save("otsu", **oz)

    # (11) tiling index lists: tests/golden/gen_tiling.py (executes the reference's own statements, infer_simple.py:180-212 and
    # core/test.py:76-90, from where they lie)


if __name__ == "__main__":
    main()
