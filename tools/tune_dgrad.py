#!/usr/bin/env python3
"""Sweep the compiled k=3 tile variants (M3D_TUNE_K3) over the shapes where cout < cin (the dgrad of a widening layer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d

shapes = [(128, 64, 32), (256, 128, 16), (64, 32, 64)]
variants = {32: [None, 1, 2, 5, 30, 31, 33, 34, 35, 36], 16: [None, 10, 11, 12, 13, 20, 21, 23, 24], 64: [None, 0, 1, 3, 4, 6, 7]}
for cin, cout, s in shapes:
    x = torch.randn(1, cin, s, s, s, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    conv = m3d.PackedConv3d(w)
    out = torch.empty(1, cout, s, s, s, device="cuda")
    fl = 2.0 * cin * cout * 27 * s ** 3
    for v in variants[s]:
        if v is None:
            os.environ.pop("M3D_TUNE_K3", None)
        else:
            os.environ["M3D_TUNE_K3"] = str(v)
        try:
            for _ in range(3):
                conv(x, out=out)
        except Exception as e:
            print("cin %d cout %d %d^3 variant %s: %s" % (cin, cout, s, v, e)); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            conv(x, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("cin %3d cout %3d %2d^3 variant %-4s %.3f ms %.1f TFLOP/s" % (cin, cout, s, v, ms, fl / ms / 1e9))
