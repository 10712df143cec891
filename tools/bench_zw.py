"""csrc/conv3d_zw.hip (f16x2 split + F(2,3) along z) against the fp32 F(2x4,3x3) Winograd kernel on the detection step's conv layers
(batch of 4 x 128^3 volumes).  usage: python tools/bench_zw.py"""
import sys; sys.path.insert(0, "/root/repo"); import __graft_entry__  # noqa
import torch
from m3d import ops

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for name, cin, cout, S, pool in [("conv2a", 32, 64, 64, False), ("conv2b+pool", 64, 64, 64, True), ("conv3a", 64, 128, 32, False),
                                 ("conv3b+pool", 128, 128, 32, True), ("conv4a", 128, 256, 16, False), ("conv4b", 256, 256, 16, False)]:
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn((4, cin, S, S, S), generator=g)).cuda()
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * (2.0 / (cin * 27)) ** 0.5).cuda()
    sc = (torch.rand(cout, generator=g) + 0.5).cuda(); sh = torch.randn(cout, generator=g).cuda()
    wino = ops.WinoConv3d(w, two_d=True)
    zw = ops.ZwConv3d(w)
    xm = ops.ZwConv3d.bound_of(x)
    om = torch.zeros(32, device="cuda")
    if pool:
        ref = wino.pooled(x, scale=sc, shift=sh, relu=True)
        t0 = timed(lambda: wino.pooled(x, scale=sc, shift=sh, relu=True))
    else:
        ref = wino(x, scale=sc, shift=sh, relu=True)
        t0 = timed(lambda: wino(x, scale=sc, shift=sh, relu=True))
    out = torch.empty_like(ref)
    y, _ = zw(x, xm, scale=sc, shift=sh, relu=True, pool=pool, out=out, out_max=om)
    err = float((y - ref).abs().max() / ref.abs().max())
    t1 = timed(lambda: zw(x, xm, scale=sc, shift=sh, relu=True, pool=pool, out=out, out_max=om))
    gf = 2.0 * 27 * cin * cout * 4 * S ** 3 / 1e9
    print("%-12s %3d->%3d @%3d^3  fp32 F(2x4,3x3) %.3f ms (%.0f TF alg)   f16x2 F(2,3)z %.3f ms (%.0f TF alg, %.0f TF f16 issued)   diff %.1e"
          % (name, cin, cout, S, t0, gf / t0, t1, gf / t1, 2 * gf / t1, err), flush=True)
