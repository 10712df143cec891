#!/usr/bin/env python3
"""Sweep the compiled k = 3 direct-kernel tile variants (tuning option tune_k3, libm3d_tune.so) over the NORM-convolution shapes of the PRM
tiles (soma 64x160x160, nuclei 64x200x200): which variant the dispatcher should pick where the library's heuristic was tuned on 128^3 maps."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import m3d
from m3d import _lib

SHAPES = [("soma 2a", 32, 64, (32, 80, 80)), ("soma 2b", 64, 64, (32, 80, 80)), ("soma 3a", 64, 128, (16, 40, 40)), ("soma 3b/rpn", 128, 128, (16, 40, 40)),
          ("nuc 2a", 32, 64, (32, 100, 100)), ("nuc 2b", 64, 64, (32, 100, 100)), ("nuc 3a", 64, 128, (16, 50, 50)), ("nuc 3b", 128, 128, (16, 50, 50)),
          ("nuc 4a", 128, 256, (8, 25, 25)), ("nuc 4b/rpn", 256, 256, (8, 25, 25))]
VARIANTS = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16, 20, 21, 22, 23, 24, 30, 31, 33, 34, 35, 36]


def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with _lib.tuning():
    for name, cin, cout, shp in SHAPES:
        x = torch.randn((1, cin) + shp, device="cuda")
        w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
        conv = m3d.PackedConv3d(w, m3d.W_RELU)
        off = x.min().reshape(1)
        gf = 2.0 * cin * cout * 27 * shp[0] * shp[1] * shp[2] / 1e9
        res = []
        for v in VARIANTS:
            _lib.set_option("tune_k3", v)
            try:
                ms = t(lambda: conv(x, in_offset=off))
                res.append((ms, v))
            except Exception:
                pass
        res.sort()
        base = [m for m, v in res if v == -1][0]
        print("%-12s %3d->%3d %-14s library %.3f ms (%5.1f TF)   best: %s" % (name, cin, cout, "x".join(map(str, shp)), base, gf / base,
              "  ".join("v%d %.3f" % (v, m) for m, v in res[:4])), flush=True)
