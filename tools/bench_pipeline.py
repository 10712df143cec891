"""Serial vs software-pipelined detection steps (batch of 4 x 128^3): begin(k+1) = norm1 + backbone + RPN + proposals on one
stream while finish(k) = RoIAlign + box head + box results + cross-tile NMS runs on another (m3d.model.detect_batch_begin / _finish)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd")]
import m3d  # noqa: E402
from m3d.config import Cfg  # noqa: E402
from m3d.model import DetectorM3D  # noqa: E402
from m3d.synth import make_params, synth_volume  # noqa: E402

VOL, B, K = 128, int(os.environ.get("B", "4")), int(os.environ.get("K", "20"))
cfg = Cfg.nuclei(in_size=(VOL, VOL, VOL))
P = make_params(stride=8, num_anchors=35, mlp_dim=cfg.mlp_dim, seed=0)
det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
raw = torch.from_numpy(np.stack([synth_volume(i, (VOL, VOL, VOL)) for i in range(B)])).cuda()
info = np.array([VOL, VOL, VOL, 1.0])
xb = [torch.empty((B, 1, VOL, VOL, VOL), device="cuda") for _ in range(2)]
cap = cfg.detections_per_im


def pack(r):
    return m3d.nms3d_batched(r["cls_boxes"][:, 1], r["cls_counts"][:, 1], cfg.nms, pack_cap=cap, want_keep=False)["packed"]


def serial():
    m3d.norm1_batched(raw, f32_arith=True, out=xb[0])
    return pack(det.detect_batch(xb[0], info, as_dicts=False))


sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def pipelined(n):
    prev, outs = None, []
    for i in range(n + 1):
        if i < n:
            with torch.cuda.stream(sA):
                m3d.norm1_batched(raw, f32_arith=True, out=xb[i & 1])
                st = det.detect_batch_begin(xb[i & 1], info)
        if prev is not None:
            with torch.cuda.stream(sB):
                outs.append(pack(det.detect_batch_finish(prev, as_dicts=False)))
        prev = st if i < n else None
    return outs


for _ in range(3):
    ref = serial()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    serial()
torch.cuda.synchronize()
ts = (time.perf_counter() - t0) / K * 1e3
pipelined(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
outs = pipelined(K)
torch.cuda.synchronize()
tp = (time.perf_counter() - t0) / K * 1e3
same = all(torch.equal(o, ref) for o in outs)
print("batch %d: serial %.3f ms/step, pipelined %.3f ms/step (%.1f %%), outputs identical: %s" % (B, ts, tp, 100 * (ts / tp - 1), same))
