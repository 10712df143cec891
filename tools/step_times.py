#!/usr/bin/env python3
"""Per-step wall times of the serial detection loop (host clock after each step's device synchronize): which steps of a short timed
region are slow.  usage: python tools/step_times.py [steps] [probed_steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd")]
import numpy as np, torch, m3d
from m3d.model import DetectorM3D, Probe
from m3d.config import Cfg
from m3d.synth import make_params, synth_volume

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
PROBED = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = Cfg.nuclei()
P = make_params(stride=8, num_anchors=35, mlp_dim=cfg.mlp_dim, seed=0)
det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
raw = torch.from_numpy(np.stack([synth_volume(i, (128, 128, 128)) for i in range(4)])).cuda()
xb = torch.empty((4, 1, 128, 128, 128), device="cuda")
info = np.array([128, 128, 128, 1.0])
cap = cfg.detections_per_im

def step():
    m3d.norm1_batched(raw, f32_arith=True, out=xb)
    r = det.detect_batch(xb, info, as_dicts=False)
    return m3d.nms3d_batched(r["cls_boxes"][:, 1], r["cls_counts"][:, 1], cfg.nms, pack_cap=cap, want_keep=False)["packed"]

for _ in range(13):
    step()
torch.cuda.synchronize()
for rep in range(2):
    probe = Probe()
    t = [time.perf_counter()]
    for k in range(K):
        det.probe = probe if k < PROBED else None
        step()
        t.append(time.perf_counter())           # host time when the step's launches are enqueued (the step's host read has passed)
    torch.cuda.synchronize()
    tend = time.perf_counter()
    d = np.diff(np.array(t)) * 1e3
    print("rep %d: total %.3f ms per step over %d steps; enqueue-to-enqueue per step: %s ; drain after the last enqueue %.3f ms" %
          (rep, (tend - t[0]) / K * 1e3, K, " ".join("%.2f" % v for v in d), (tend - t[-1]) * 1e3))
