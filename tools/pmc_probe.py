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
# the stress step of the bench (`stress_rois`: RPN NMS off, RPN_POST_NMS_TOP_N = 1000 RoIs per volume reach RoIAlign3D and the box head)
keep_thr = cfg.rpn_nms_thresh
cfg.rpn_nms_thresh = 1.0
for _ in range(2):
    r_s = det.detect_batch(x, as_dicts=False)
torch.cuda.synchronize()
print("stress rois", r_s["num_rois"])
cfg.rpn_nms_thresh = keep_thr
# configs[1]: the backbone alone on ONE volume (the sub-record `configs1_backbone`; same kernels, a quarter of the grid)
for _ in range(3):
    det.conv_body(x[:1].contiguous())
torch.cuda.synchronize()
del det, x
# PRM tiles (configs[3] soma 64x160x160 and the nuclei net's 64x200x200): three tiles each
from m3d.prm import PRMEngine
import numpy as np
from m3d import tiling
for cfgp in (Cfg.soma(), Cfg.nuclei(score_thresh=0.0)):
    Pp = make_params(stride=cfgp.stride, num_anchors=cfgp.num_anchors, mlp_dim=cfgp.mlp_dim, seed=0)
    eng = PRMEngine(DetectorM3D({k: v.cuda() for k, v in Pp.items()}, cfgp))
    S, H, W = cfgp.in_size
    vol = torch.from_numpy(tiling.norm1(synth_volume(0, (S, H, W)), np.float32).astype(np.float32)).reshape(1, 1, S, H, W).cuda()
    for _ in range(3):
        out = eng.prm_tile(vol, dense=False)
    torch.cuda.synchronize()
    print("prm tile", cfgp.in_size, "peaks", None if out is None else int(out["peaks"].shape[0]))
    del eng, Pp
    torch.cuda.empty_cache()
print("pmc_probe done; calibration bytes", N_CAL * 4, "rois", r["num_rois"])
