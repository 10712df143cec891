#!/usr/bin/env python3
"""Where the 2-D Winograd kernel stops paying: direct MFMA kernel vs Winograd (forced) on shapes whose tile score is near the
threshold WinoConv3d.supports() applies (m3d_conv3d_wino2_score >= threshold)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d
from m3d._lib import lib

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n

shapes = [(1, 128, 128, 16, 40, 40), (1, 64, 128, 16, 40, 40), (1, 128, 128, 16, 16, 16), (1, 256, 256, 8, 25, 25), (1, 64, 64, 32, 80, 80),
          (1, 32, 64, 32, 80, 80), (1, 256, 256, 16, 16, 16), (2, 256, 256, 8, 25, 25), (1, 128, 256, 8, 25, 25), (1, 128, 128, 8, 20, 20)]
for (B, cin, cout, D, H, W) in shapes:
    x = torch.randn(B, cin, D, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    out = torch.empty(B, cout, D, H, W, device="cuda")
    d = m3d.PackedConv3d(w); wn = m3d.WinoConv3d(w, two_d=True)
    sc = lib().m3d_conv3d_wino2_score(B, cin, cout, D, H, W)
    td = timeit(lambda: d(x, relu=True, out=out)); tw = timeit(lambda: wn(x, relu=True, out=out))
    print("B%d %3d->%3d %2dx%2dx%2d  score %.2f  direct %.3f ms  winograd %.3f ms  -> %s" % (B, cin, cout, D, H, W, sc, td, tw, "winograd" if tw < td else "direct"))
