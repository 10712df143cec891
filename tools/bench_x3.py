"""Timing of the bf16x3 direct 3^3 conv (csrc/conv3d_x3.hip) at launch sizes of exactly one / half / two rounds of the chip and two channel
counts, and against the fp32-MFMA direct kernel on the PRM norm-conv shapes.  M3D_LIB_PATH selects an ablation build (tools/x3_ablate.sh).
usage: python tools/bench_x3.py"""
import sys; sys.path.insert(0, "/root/repo"); import __graft_entry__  # noqa
import torch
from m3d import ops
for name, cin, cout, shape in [("512 WGs", 64, 64, (32, 64, 64)), ("256 WGs", 64, 64, (16, 64, 64)), ("1024 WGs", 64, 64, (64, 64, 64)), ("512 WGs cin256", 256, 64, (32, 64, 64)), ("512 WGs cin16", 16, 64, (32, 64, 64))]:
    g = torch.Generator().manual_seed(1)
    x = torch.rand((1, cin) + shape, generator=g).cuda()
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.1).cuda()
    off = ops.reduce_min(x)
    conv = ops.X3Conv3d(w, ops.W_RELU)
    for _ in range(3): conv(x, in_offset=off)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): conv(x, in_offset=off)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gf = 2.0 * 27 * cin * cout * shape[0] * shape[1] * shape[2] / 1e9
    taps = cin // 16 * 27
    print("%-16s %.3f ms  %.0f TF alg = %.2f of bf16 peak issued; per tap %.0f ns" % (name, ms, gf / ms, 6 * gf / ms / 2500, ms * 1e6 / taps / max(1, (shape[0] * shape[1] * shape[2] // 256 + 511) // 512)))

print("PRM norm-conv shapes: bf16x3 vs fp32 direct (ms)")
for name, cin, cout, shape in [("soma 2a", 32, 64, (32, 80, 80)), ("soma 2b", 64, 64, (32, 80, 80)), ("soma 3a", 64, 128, (16, 40, 40)), ("soma 3b/rpn", 128, 128, (16, 40, 40)),
                               ("nuc 2a", 32, 64, (32, 100, 100)), ("nuc 2b", 64, 64, (32, 100, 100)), ("nuc 3a", 64, 128, (16, 50, 50)), ("nuc 3b", 128, 128, (16, 50, 50)),
                               ("nuc 4a", 128, 256, (8, 25, 25)), ("nuc 4b/rpn", 256, 256, (8, 25, 25))]:
    g = torch.Generator().manual_seed(1)
    x = torch.rand((1, cin) + shape, generator=g).cuda()
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.1).cuda()
    off = ops.reduce_min(x)
    res = []
    mx = ops.reduce_minmax_multi([x])[1]
    c16 = ops.X3Conv3d(w, ops.W_RELU, f16=True)
    for conv in (ops.X3Conv3d(w, ops.W_RELU), ops.PackedConv3d(w, ops.W_RELU), lambda x_, in_offset: c16(x_, in_offset=in_offset, in_max=mx)):
        for _ in range(3): conv(x, in_offset=off)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): conv(x, in_offset=off)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    gf = 2.0 * 27 * cin * cout * shape[0] * shape[1] * shape[2] / 1e9
    print("%-12s %4d->%4d %-14s bf16x3 %.3f ms (%.0f TF alg)   fp32 %.3f ms (%.0f TF)   f16x2 %.3f ms (%.0f TF)" % (name, cin, cout, shape, res[0], gf / res[0], res[1], gf / res[1], res[2], gf / res[2]))
