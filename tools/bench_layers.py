#!/usr/bin/env python3
"""Per-layer conv timing (HIP events) for tuning: python tools/bench_layers.py [size] [reps]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = [("conv1a", 1, 32, 5, size), ("conv2a", 32, 64, 3, size // 2), ("conv2b", 64, 64, 3, size // 2),
     ("conv3a", 64, 128, 3, size // 4), ("conv3b", 128, 128, 3, size // 4), ("conv4a", 128, 256, 3, size // 8),
     ("conv4b", 256, 256, 3, size // 8), ("rpn_conv", 256, 256, 3, size // 8), ("rpn_heads", 256, 245, 1, size // 8)]
tot_t, tot_f, tot_w = 0.0, 0.0, 0.0
only = os.environ.get('LAYERS')
BATCH = int(os.environ.get("BATCH", "1"))
for name, cin, cout, k, s in L:
    if only and name not in only.split(','):
        continue
    x = torch.randn(BATCH, cin, s, s, s, device="cuda")
    w = torch.randn(cout, cin, k, k, k, device="cuda") * 0.05
    conv = m3d.PackedConv3d(w)
    sc = torch.rand(cout, device="cuda"); sh = torch.rand(cout, device="cuda")
    out = torch.empty(BATCH, cout, s, s, s, device="cuda")
    fused = name in ("conv1a", "conv2b") and os.environ.get("FUSED", "1") == "1"   # as the real pipeline runs them
    run = (lambda: conv.pooled(x, scale=sc, shift=sh, relu=True)) if fused else (lambda: conv(x, scale=sc, shift=sh, relu=True, out=out))
    if fused:
        name = name + "+pool"
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * cin * cout * k ** 3 * s ** 3 * BATCH
    tot_t += ms; tot_f += fl
    line = "%-12s cin %3d cout %3d k%d %3d^3  direct %8.3f ms %6.2f TFLOP/s (%.1f%%)" % (name, cin, cout, k, s, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100)
    msw = ms
    mode = int(os.environ.get("M3D_WINO", "2"))
    if k == 3 and mode != 0:                                                 # what the pipeline runs by default
        for two_d in ((False, True) if mode == 2 else (False,)):
            wino = m3d.WinoConv3d(w, two_d=two_d)
            if not wino.supports(s):
                continue
            runw = (lambda: wino.pooled(x, scale=sc, shift=sh, relu=True)) if (fused and wino.supports_pool(s)) else \
                   (lambda: wino(x, scale=sc, shift=sh, relu=True, out=out))
            for _ in range(3):
                runw()
            torch.cuda.synchronize(); e0.record()
            for _ in range(reps):
                runw()
            e1.record(); torch.cuda.synchronize()
            msw = e0.elapsed_time(e1) / reps
            fam4 = two_d and m3d._lib.lib().m3d_conv3d_wino2_family() == 4        # F(2x4,3x3): 1/3 of the multiplies, else F(2x2,3x3): 4/9
            frac = (1.0 / 3.0 if fam4 else 4.0 / 9.0) if two_d else 2.0 / 3.0
            line += "   %s %7.3f ms %6.2f TF alg. (%.0f%%; %.0f%% issued)" % (
                ("F(2x4,3x3)" if fam4 else "F(2x2,3x3)") if two_d else "F(2,3)x", msw, fl / msw / 1e9, fl / msw / 1e9 / 157.3 * 100, fl / msw / 1e9 / 157.3 * 100 * frac)
    if k == 3 and mode != 0 and os.environ.get("M3D_CONV_F16", "1") == "1" and m3d.ZwConv3d.supported(w, (s, s, s)):   # round 6: what the pipeline runs
        zw = m3d.ZwConv3d(w)
        xr = torch.relu(x)
        xm = m3d.ZwConv3d.bound_of(xr)
        om = torch.zeros(32, device="cuda")
        fz = name.startswith(("conv2b", "conv3b")) and zw.supports((s, s, s), pool=True)
        outz = torch.empty((BATCH, cout, s // 2, s // 2, s // 2) if fz else (BATCH, cout, s, s, s), device="cuda")
        runz = lambda: zw(xr, xm, scale=sc, shift=sh, relu=True, pool=fz, out=outz, out_max=om)
        for _ in range(3):
            runz()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps):
            runz()
        e1.record(); torch.cuda.synchronize()
        msw = e0.elapsed_time(e1) / reps
        line += "   f16x2 F(2,3)z%s %7.3f ms %6.2f TF alg. (%.0f TF f16 issued = %.0f%% of 2.5 PF)" % (
            "+pool" if fz else "", msw, fl / msw / 1e9, 2 * fl / msw / 1e9, 2 * fl / msw / 1e9 / 2500 * 100)
    if k == 5 and mode != 0 and m3d.StemWinoConv3d.supports(s):
        sw = m3d.StemWinoConv3d(w)
        runw = (lambda: sw.pooled(x, scale=sc, shift=sh, relu=True)) if fused else (lambda: sw(x, scale=sc, shift=sh, relu=True, out=out))
        for _ in range(3):
            runw()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps):
            runw()
        e1.record(); torch.cuda.synchronize()
        msw = e0.elapsed_time(e1) / reps
        line += "   F(2,5)x %7.3f ms %6.2f TF alg. (%.0f%%; %.0f%% issued)" % (msw, fl / msw / 1e9, fl / msw / 1e9 / 157.3 * 100, fl / msw / 1e9 / 157.3 * 100 * 0.624)
    tot_w += msw
    print(line)
print("TOTAL direct %.3f ms (%.2f TFLOP/s)   as run (f16x2 / Winograd where supported) %.3f ms (%.2f TFLOP/s algorithmic)  %.2f GFLOP" %
      (tot_t, tot_f / tot_t / 1e9, tot_w, tot_f / tot_w / 1e9, tot_f / 1e9))
