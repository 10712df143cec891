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
    bad += 0 if ok else 1
    print("case %3d stride %d tile %-14s data %d pre/post %4d/%4d thr %.2f: %s%s" % (ci, stride, shape, mode, cfg.pre_nms_topN, cfg.post_nms_topN, thr, what, "" if ok else "  BAD"), flush=True)
print("fuzz_tile: %d cases, %d differ, %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
