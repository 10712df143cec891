#!/usr/bin/env python3
"""Split-K factor sweep of the fc1 GEMM at the RoI counts the bench produces (M3D option tune_fc_slices)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import m3d
from m3d import _lib
_lib.tuning().__enter__()      # option sweeps: the tuning build (libm3d_tune.so) for the whole process
N, K = 1024, 87808
w = torch.randn(N, K, device="cuda") / K ** 0.5
b = torch.randn(N, device="cuda")
for M in (320, 1281, 1400, 2562):
    x = torch.randn(M, K, device="cuda")
    res = []
    for s in (0, 4, 5, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 48):
        _lib.set_option("tune_fc_slices", s if s else -1)
        for _ in range(2):
            m3d.linear(x, w, b, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m3d.linear(x, w, b, relu=True)
        e1.record(); torch.cuda.synchronize()
        res.append("%s:%.3f" % (s if s else "auto", e0.elapsed_time(e1) / 5))
    print("M=%d  ms by slices: %s" % (M, "  ".join(res)), flush=True)
