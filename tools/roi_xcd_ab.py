#!/usr/bin/env python3
"""RoIAlign3D on the bench's RoIs with the two workgroup -> (RoI, channels) maps (option tune_roi_xcd, tuning build): 0 = one workgroup per
small RoI over all 256 channels (the release library), 1 = XCD-aware: every XCD takes one eighth of every RoI's channels, so that an
XCD's L2 has to hold 2.1 MB of the four volumes' feature maps instead of 16.8 MB.  Prints the time of both; run under
`rocprofv3 --pmc FETCH_SIZE` (and, separately, `--pmc WRITE_SIZE`) with ROI_XCD_PMC=1 for one launch of each map.
usage: python tools/roi_xcd_ab.py [--stress]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import m3d
from m3d import _lib
from m3d.config import Cfg
from m3d.model import DetectorM3D
from m3d.synth import make_params, synth_volume
_lib.tuning().__enter__()

cfg = Cfg.nuclei(in_size=(128, 128, 128))
if "--stress" in sys.argv:
    cfg.rpn_nms_thresh = 1.0
P = make_params(stride=8, num_anchors=35, mlp_dim=64, seed=0)
det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
x = torch.stack([m3d.norm1(torch.from_numpy(synth_volume(i, (128, 128, 128))).cuda()) for i in range(4)])[:, None]
r = det.detect_batch(x, as_dicts=False)
rois = torch.cat([r["rois"][b, :r["num_rois"][b]] for b in range(4)]).contiguous()
feat = r["feat"]
R = int(rois.shape[0])
by = R * 256 * 343 * 4


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
for mode in (0, 1, 0, 1):
    _lib.set_option("tune_roi_xcd", mode)
    out = m3d.roi_align3d_forward(feat, rois, 7, 7, 7, 0.125, 2)
    if ref is None:
        ref = out.clone()
    assert torch.equal(out, ref)                       # the map is a speed option: the same arithmetic per output row
    if os.environ.get("ROI_XCD_PMC"):
        torch.cuda.synchronize()
        continue
    t = timeit(lambda: m3d.roi_align3d_forward(feat, rois, 7, 7, 7, 0.125, 2))
    print("tune_roi_xcd = %d: %d RoIs  %.3f ms  (%.0f MB written, %.2f TB/s = %.1f %% of 8 TB/s)" % (mode, R, t, by / 1e6, by / t / 1e9, by / t / 1e9 / 8 * 100), flush=True)
