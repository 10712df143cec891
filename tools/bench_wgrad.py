#!/usr/bin/env python3
"""Per-layer timing of the backward kernels (dgrad via the forward kernel on flipped weights; MFMA split-K wgrad)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = [("conv1a", 1, 32, 5, size), ("conv2a", 32, 64, 3, size // 2), ("conv2b", 64, 64, 3, size // 2),
     ("conv3a", 64, 128, 3, size // 4), ("conv3b", 128, 128, 3, size // 4), ("conv4a", 128, 256, 3, size // 8),
     ("conv4b", 256, 256, 3, size // 8), ("rpn_heads", 256, 245, 1, size // 8)]


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tw = td = tf = 0.0
for name, cin, cout, k, s in L:
    x = torch.randn(1, cin, s, s, s, device="cuda")
    gy = torch.randn(1, cout, s, s, s, device="cuda")
    w = torch.randn(cout, cin, k, k, k, device="cuda") * 0.05
    fl = 2.0 * cin * cout * k ** 3 * s ** 3
    ms_w = timeit(lambda: m3d.conv3d_wgrad(x, gy, k))
    line = "%-10s cin %3d cout %3d k%d %3d^3  wgrad %8.3f ms %7.2f TFLOP/s (%.1f%%)" % (name, cin, cout, k, s, ms_w, fl / ms_w / 1e9, fl / ms_w / 1e9 / 157.3 * 100)
    if k != 5:
        dg = m3d.PackedConv3d(w, mode=m3d.W_DGRAD)
        ms_d = timeit(lambda: dg(gy))
        line += "   dgrad %8.3f ms %7.2f TFLOP/s (%.1f%%)" % (ms_d, fl / ms_d / 1e9, fl / ms_d / 1e9 / 157.3 * 100)
        td += ms_d
    tw += ms_w; tf += fl
    print(line)
print("TOTAL wgrad %.3f ms (%.1f TFLOP/s)   dgrad (without stem) %.3f ms" % (tw, tf / tw / 1e9, td))
