#!/usr/bin/env python3
"""RoI geometry of the soma PRM tile (configs[3]) and RoIAlign timing per size class."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import m3d
from m3d.config import Cfg
from m3d.model import DetectorM3D
from m3d.synth import make_params, synth_volume
from m3d import tiling

cfg = Cfg.soma()
P = make_params(stride=cfg.stride, num_anchors=cfg.num_anchors, mlp_dim=cfg.mlp_dim, seed=0)
det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
S, H, W = cfg.in_size
vol = torch.from_numpy(tiling.norm1(synth_volume(0, (S, H, W)), np.float32).astype(np.float32)).reshape(1, 1, S, H, W).cuda()
feat = det.conv_body(vol)
prob, deltas = det.rpn(feat)
rois, probs, keep = det.proposals(prob, deltas, np.array([S, H, W, 1.0]))
print("feat", tuple(feat.shape), "rois", tuple(rois.shape), "roi_res", cfg.roi_res, "ratio", cfg.sampling_ratio)
r = rois.cpu().numpy()
ext = (r[:, 4:7] - r[:, 1:4]) / cfg.stride
print("extent in cells (x,y,z): mean", ext.mean(0), "p50", np.percentile(ext, 50, 0), "p90", np.percentile(ext, 90, 0), "max", ext.max(0))
big = (ext.max(1) > 28)
print("rois with an axis > 28 cells (bins > 4 cells):", int(big.sum()), "of", len(r))
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for name, sel in (("all", np.ones(len(r), bool)), ("<=28", ~big), (">28", big)):
    if sel.sum():
        rr = torch.from_numpy(r[sel]).cuda()
        t = timeit(lambda: m3d.roi_align3d_forward(feat, rr, cfg.roi_res, cfg.roi_res, cfg.roi_res, 1.0 / cfg.stride, cfg.sampling_ratio))
        print("   %-5s %4d rois: %.3f ms = %.2f us per roi" % (name, sel.sum(), t, t * 1e3 / sel.sum()))
