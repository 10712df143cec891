#!/usr/bin/env python3
"""Latency of the fused, batched box stages (csrc/box_fused.hip) on the tensors the detector really produces for the bench
volumes, and - with the diagnostic library (`make -C .../csrc stamps`, M3D_LIB_PATH=.../libm3d_stamps.so) - cycles per phase."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import m3d  # noqa: E402
from m3d._lib import lib  # noqa: E402
from m3d.config import Cfg  # noqa: E402
from m3d.model import DetectorM3D  # noqa: E402
from m3d.synth import make_params, synth_volume  # noqa: E402


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = 4
    cfg = Cfg.nuclei(in_size=(128, 128, 128))
    P = make_params(stride=8, num_anchors=35, mlp_dim=cfg.mlp_dim, seed=0)
    det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
    x = torch.stack([m3d.norm1(torch.from_numpy(synth_volume(i, (128, 128, 128))).cuda()) for i in range(B)])[:, None]
    feat = det.conv_body(x)
    prob, deltas = det.rpn(feat)
    info = np.array([128., 128., 128., 1.0])
    f = lambda: m3d.generate_proposals3d_batched(prob, deltas, det.anchors, 8.0, info, 1000, 1000, 0.15)   # noqa: E731
    print("proposals, batch of %d tiles (143 360 anchors each): %.1f us per launch" % (B, timeit(f)))
    f1 = lambda: m3d.generate_proposals3d_batched(prob[:1], deltas[:1], det.anchors, 8.0, info, 1000, 1000, 0.15)   # noqa: E731
    print("proposals, one tile: %.1f us" % timeit(f1))
    L = lib()
    if hasattr(L, "m3d_debug_read_stamps"):
        f(); torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)()
        L.m3d_debug_read_stamps(buf)
        s = list(buf)
        names = ["radix select", "final key compaction", "key sort", "decode", "ordered compaction", "nms prepare (sort)"]
        for i, nm in enumerate(names):
            print("   stage 1  %-22s %8d cycles" % (nm, s[i + 1] - s[i]))
        print("   stage 3  %-22s %8d cycles" % ("nms resolve", s[11] - s[10]))
        print("   stage 3  %-22s %8d cycles" % ("gather", s[12] - s[11]))
        print("   stage 1 start -> stage 3 end: %d cycles (incl. stage 2 and two launch boundaries)" % (s[12] - s[0]))
    r = det.detect_batch(x, info, as_dicts=False)
    offs = torch.from_numpy(r["offsets"]).cuda()
    g = lambda: m3d.box_results3d_batched(r["cls"], r["pred_boxes"], None, offs, 2, cfg.score_thresh, cfg.nms, 300, 1000)   # noqa: E731
    print("box_results, batch of %d tiles (%s rois): %.1f us" % (B, r["num_rois"], timeit(g)))
    h = lambda: m3d.nms3d_batched(r["cls_boxes"][:, 1], r["cls_counts"][:, 1], cfg.nms, pack_cap=300, want_keep=False)   # noqa: E731
    print("cross-tile nms + pack, batch of %d (%s dets in): %.1f us" % (B, r["cls_counts"][:, 1].tolist(), timeit(h)))
    for N in (300, 1000, 2000):
        rs = np.random.RandomState(N)
        c = rs.uniform(0, 128, (N, 3)); e = rs.uniform(8, 40, (N, 3))
        d = torch.from_numpy(np.hstack((c - e / 2, c + e / 2, rs.permutation(N)[:, None] / N)).astype(np.float32)).cuda()
        print("nms3d N=%4d (one launch + count read-back): %.1f us" % (N, timeit(lambda: m3d.nms3d(d, 0.15))))


if __name__ == "__main__":
    main()
