#!/usr/bin/env python3
"""Differential fuzzing of one PRM-mode tile end to end - forward, proposals, box head, box results, peak selection - against the oracle's
restatement of PeakResponseMapping_3d.forward (lib/prm/peak_response_mapping_3d.py:85-193): the kept peaks (anchor, s, h, w) must be the
same list in the same order, the detections equal to 1e-4.  Random nets (stride 4 / 8), tile shapes and contents.  Test infrastructure.
  python tools/fuzz_tile.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import oracle as O
from m3d.model import DetectorM3D
from m3d.prm import PRMEngine

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
bad = 0
cond = 0
t0 = time.time()
for ci in range(cases):
    rs = np.random.RandomState(seed0 * 1000 + ci)
    stride = int(rs.choice([4, 8]))
    A = 35 if stride == 8 else 14
    shape = tuple(int(stride * rs.randint(2, 7) + rs.choice([0, 0, 1, 3])) for _ in range(3))
    shape = (min(shape[0], 40), min(shape[1] + 16, 80), min(shape[2] + 16, 80))
    P = O.make_params(stride=stride, num_anchors=A, mlp_dim=32, seed=int(rs.randint(1000)))
    kw = dict(mlp_dim=32, score_thresh=float(rs.choice([0.0, 0.05])), pre_nms_topN=int(rs.choice([50, 300, 1000])), post_nms_topN=int(rs.choice([30, 300, 1000])))
    cfg = O.Cfg(**kw) if stride == 8 else O.Cfg.soma(**kw)
    if stride == 8:
        P = dict(P)
        for k in ("RPN.RPN_cls_score.weight", "RPN.RPN_cls_score.bias"):
            P[k] = P[k] * 0.25
    mode = rs.randint(3)
    vol = rs.rand(1, 1, *shape).astype(np.float32)
    if mode == 1:
        vol *= (rs.rand(1, 1, *shape) > 0.5)
    elif mode == 2:
        vol = np.round(vol * 4) / 4                          # many equal values: ties in scores and arg-max
    vol = torch.from_numpy(vol)
    thr = float(rs.choice([0.0, 0.1, 0.3]))
    eng = PRMEngine(DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg))
    out = eng.prm_tile(vol.cuda(), peak_threshold=thr, dense=False)
    try:
        with torch.no_grad():
            ref = O.prm_tile(P, cfg, vol, peak_threshold=thr, max_peaks=0)
    except RuntimeError as e:                                # no proposal at all: the box head's x.view(0, -1) raises in torch, as it would in the
        ref = (None, None, None, None)                       # reference (fast_rcnn_heads.py:112); the engine returns None for such a tile
        print("   (oracle: %s)" % str(e)[:60])
    if (out is None) != (ref[0] is None):
        ok, what = False, "one side empty"
    elif out is None:
        ok, what = True, "empty"
    else:
        gp, gd = out["peaks"].numpy(), out["dets"].numpy()
        rp, rd = np.asarray(ref[1]), np.asarray(ref[3])
        ok = gp.shape == rp.shape and np.array_equal(gp, rp) and np.allclose(gd, rd, rtol=1e-4, atol=1e-3)
        what = "%d peaks" % len(rp) if ok else "peaks %s vs %s, first diff row %s" % (gp.shape, rp.shape, (np.nonzero((gp != rp).any(1))[0][:1] if gp.shape == rp.shape else "-"))
    if not ok and out is not None and ref[0] is not None:
        # A threshold decision (RPN NMS, score cut, per-class NMS) can flip on the last bits of the forward: the engine's convolutions
        # and the oracle's agree to ~1e-6, not bit for bit.  Such a case is CONDITIONING, not a defect, iff the oracle's own
        # post-processing chain, fed the ENGINE's RPN outputs and feature map, reproduces the engine's peaks and detections - and the two
        # forwards agree to rounding.  (round 6: 1 of 300 cases at seed 29; tools/fuzz_tile_case.py prints such a case in detail.)
        with torch.no_grad():
            feat, prob, deltas, _, _ = eng.forward(vol.cuda())
            S_, H_, W_ = shape
            im_info = np.array([S_, H_, W_, 1.0], np.float64)
            rois, _, keep_idx = O.generate_proposals_3d(prob[0].cpu().numpy(), deltas[0].cpu().numpy(), im_info, cfg.anchors, cfg.stride,
                                                        cfg.pre_nms_topN, cfg.post_nms_topN, cfg.rpn_nms_thresh, cfg.rpn_min_size)
            cls, bbox = O.box_head_forward(P, feat.cpu(), rois, cfg.roi_res, 1.0 / cfg.stride, cfg.sampling_ratio)
            scores = cls.numpy().reshape(-1, cls.shape[-1])
            pred = O.clip_tiled_boxes_3d(O.bbox_transform_3d(rois[:, 1:7], bbox.numpy().reshape(-1, bbox.shape[-1]), cfg.bbox_reg_weights), im_info[:3])
            sc, bx, _, cls_keep = O.box_results_with_nms_and_limit(scores, pred, keep_idx, cfg.num_classes, cfg.score_thresh, cfg.nms, cfg.detections_per_im)
            B_, A_, s_, h_, w_ = prob.shape
            pk2 = [np.array((b0, a0, s0, h0, w0)) for (b0, s0, h0, w0, a0), scv in
                   zip((np.unravel_index(i, (B_, s_, h_, w_, A_)) for i in cls_keep[1]), sc) if scv > thr]
            d2 = np.array([np.append(bx[i], sc[i]) for i in range(len(sc)) if sc[i] > thr]).reshape(-1, 7)
            fwd = float((prob.cpu() - ref[0]).abs().max())
        same = len(pk2) == len(gp) and (len(pk2) == 0 or np.array_equal(np.stack(pk2), gp)) and np.allclose(gd, d2, rtol=1e-4, atol=1e-3)
        if same and fwd <= 2e-5:
            ok = True
            cond += 1
            what += " -> CONDITIONING: the oracle's post-processing on the engine's forward gives the engine's result; forwards differ by %.1e" % fwd
    bad += 0 if ok else 1
    print("case %3d stride %d tile %-14s data %d pre/post %4d/%4d thr %.2f: %s%s" % (ci, stride, shape, mode, cfg.pre_nms_topN, cfg.post_nms_topN, thr, what, "" if ok else "  BAD"), flush=True)
print("fuzz_tile: %d cases, %d differ, %d decided by the forward's last bits (conditioning), %.0f s" % (cases, bad, cond, time.time() - t0))
sys.exit(1 if bad else 0)
