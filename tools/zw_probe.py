#!/usr/bin/env python3
"""Workload for PMC passes over csrc/conv3d_zw.hip alone (tools/zw_pmc.sh): the detection step's f16x2 conv layers, five launches each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch
from m3d import ops
for cin, cout, S, pool in [(32, 64, 64, False), (64, 64, 64, True), (128, 128, 32, True), (256, 256, 16, False)]:
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn((4, cin, S, S, S), generator=g)).cuda()
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * (2.0 / (cin * 27)) ** 0.5).cuda()
    zw = ops.ZwConv3d(w)
    xm = ops.ZwConv3d.bound_of(x)
    for _ in range(5):
        zw(x, xm, relu=True, pool=pool)
    torch.cuda.synchronize()
