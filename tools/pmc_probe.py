#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`):
calibration launches with a known byte count in the access widths the conv kernels use (4 B/lane coalesced dword loads:
m3d_reduce_min; 16 B/lane: a torch copy), then three backbone forwards.  tools/pmc_traffic.py turns the two
counter_collection.csv files into profiles/rNN_pmc_traffic.json, which bench.py reports as roofline.traffic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d
from m3d.config import Cfg
from m3d.model import DetectorM3D
from m3d.synth import make_params

torch.manual_seed(0)
N_CAL = 320 * 1024 * 1024                       # 1.25 GiB of fp32: well past the 256 MiB Infinity Cache
x = torch.rand(N_CAL, device="cuda")
y = torch.empty_like(x)
for _ in range(2):
    m3d.reduce_min(x)                            # min_partial_kernel: reads N_CAL*4 bytes, dword loads
    y.copy_(x)                                   # 16 B/lane reads + writes
torch.cuda.synchronize()
del x, y
from m3d.synth import synth_volume
cfg = Cfg.nuclei(in_size=(128, 128, 128))
P = make_params(stride=8, num_anchors=35, mlp_dim=cfg.mlp_dim, seed=0)
det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
# the bench's step: a batch of 4 synthetic 128^3 volumes through the whole detection pipeline (bench.py --workload detect)
x = torch.stack([m3d.norm1(torch.from_numpy(synth_volume(i, (128, 128, 128))).cuda()) for i in range(4)])[:, None].contiguous()
for _ in range(3):
    r = det.detect_batch(x, as_dicts=False)
    m3d.nms3d_batched(r["cls_boxes"][:, 1], r["cls_counts"][:, 1], cfg.nms, pack_cap=300, want_keep=False)
torch.cuda.synchronize()
print("pmc_probe done; calibration bytes", N_CAL * 4, "rois", r["num_rois"])
