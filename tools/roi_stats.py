#!/usr/bin/env python3
"""RoI geometry of the bench workload (which RoIAlign tier each RoI takes) + RoIAlign timing on those RoIs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import m3d
from m3d.config import Cfg
from m3d.model import DetectorM3D
from m3d.synth import make_params, synth_volume

B = 4
cfg = Cfg.nuclei(in_size=(128, 128, 128))
if "--stress" in sys.argv:                      # RPN NMS off: RPN_POST_NMS_TOP_N = 1000 proposals per volume (bench.py's stress_rois sub-record)
    cfg.rpn_nms_thresh = 1.0
P = make_params(stride=8, num_anchors=35, mlp_dim=cfg.mlp_dim, seed=0)
det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
x = torch.stack([m3d.norm1(torch.from_numpy(synth_volume(i, (128, 128, 128))).cuda()) for i in range(B)])[:, None]
r = det.detect_batch(x, as_dicts=False)
rois = torch.cat([r["rois"][b, :r["num_rois"][b]] for b in range(B)]).cpu().numpy()
s = rois[:, 1:] * 0.125
ext = []
for q in s:
    e = []
    for a, dim in ((0, 16), (1, 16), (2, 16)):
        lo, hi = q[a], q[a + 3]
        roi = max(hi - lo, 1.0)
        c0 = lo + 0.25 * roi / 7
        c1 = lo + roi - 0.25 * roi / 7
        l0 = int(np.clip(np.floor(max(c0, 0)), 0, dim - 1)); l1 = int(np.clip(np.floor(max(c1, 0)) + 1, 0, dim - 1))
        e.append(l1 - l0 + 1)
    ext.append(e)
ext = np.array(ext)
sub = ext.prod(1)
per = sub + ext[:, 2] * ext[:, 1] * 7 + ext[:, 2] * 49
print("rois", len(rois), "extent mean", ext.mean(0), "sub-volume floats: median %d  p90 %d  max %d" % (np.median(sub), np.percentile(sub, 90), sub.max()))
print("tier small (4*per_ch <= 6144): %d   medium (per_ch <= 6144): %d   huge: %d" % ((4 * per <= 6144).sum(), ((4 * per > 6144) & (per <= 6144)).sum(), (per > 6144).sum()))
feat = r["feat"]
ro = torch.from_numpy(rois).cuda()
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
t = timeit(lambda: m3d.roi_align3d_forward(feat, ro, 7, 7, 7, 0.125, 2))
by = len(rois) * 256 * 343 * 4
print("roi_align3d on these %d rois: %.3f ms  (%.0f MB written, %.2f TB/s = %.1f%% of 8 TB/s)" % (len(rois), t, by / 1e6, by / t / 1e9, by / t / 1e9 / 8 * 100))
fb = m3d.ops.absmax(feat)
tg = timeit(lambda: m3d.roi_align3d_forward(feat, ro, 7, 7, 7, 0.125, 2, feat_absmax=fb))
print("   with the matrix-core form for sub-volumes <= 128 voxels (%d of them <= 64, %d <= 128): %.3f ms  (%.2f TB/s = %.1f%% of 8 TB/s)"
      % ((sub <= 64).sum(), (sub <= 128).sum(), tg, by / tg / 1e9, by / tg / 1e9 / 8 * 100))
ta = timeit(lambda: m3d.ops.absmax(feat))
print("   (absmax of the feature maps: %.3f ms)" % ta)
for name, sel in (("small", 4 * per <= 6144), ("medium", (4 * per > 6144) & (per <= 6144)), ("huge", per > 6144)):
    if sel.sum():
        rr = torch.from_numpy(rois[sel]).cuda()
        t = timeit(lambda: m3d.roi_align3d_forward(feat, rr, 7, 7, 7, 0.125, 2))
        print("   %-6s %4d rois: %.3f ms = %.2f us per roi" % (name, sel.sum(), t, t * 1e3 / sel.sum()))
heavy = ~(4 * per <= 6144)
for name, order in (("heavy first", np.argsort(~heavy, kind="stable")), ("heavy last", np.argsort(heavy, kind="stable")), ("by size desc", np.argsort(-per))):
    rr = torch.from_numpy(rois[order]).cuda()
    t = timeit(lambda: m3d.roi_align3d_forward(feat, rr, 7, 7, 7, 0.125, 2))
    print("   order %-12s: %.3f ms" % (name, t))
z = timeit(lambda: torch.zeros((len(rois), 256, 7, 7, 7), device="cuda"))
print("   torch.zeros of the output alone: %.3f ms" % z)
