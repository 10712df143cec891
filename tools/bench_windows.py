#!/usr/bin/env python3
"""Conv throughput on batches of PRM windows (batch = peaks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d
cases = [("rpn 3^3", 256, 256, 3, 64), ("4b 5^3", 256, 256, 5, 64), ("4a 7^3", 256, 128, 7, 64), ("3b 16^3", 128, 128, 16, 64),
         ("3a 18^3", 128, 64, 18, 64), ("2b 38^3", 64, 64, 38, 64), ("2a 40^3", 64, 32, 40, 64)]
for name, cin, cout, w, P in cases:
    x = torch.randn(P, cin, w, w, w, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    conv = m3d.PackedConv3d(wt)
    out = torch.empty(P, cout, w, w, w, device="cuda")
    for _ in range(2):
        conv(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5):
        conv(x, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * cin * cout * 27 * w ** 3 * P
    print("%-8s cin %3d cout %3d x%d peaks: %8.3f ms %8.2f GFLOP %7.2f TFLOP/s" % (name, cin, cout, P, ms, fl / 1e9, fl / ms / 1e9))
