"""The f16x2 F(2,3)z conv kernel against the fp32 F(2x4,3x3) kernel on the PRM strips' shapes (windows side by side along x; one global
bound here - the engine uses one scale per window), with the cost of the bound sweep.  usage: python tools/bench_zw_strips.py"""
import sys; sys.path.insert(0, "/root/repo"); import __graft_entry__  # noqa
import torch
from m3d import ops
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for name, cin, cout, D, H, W in [("nuc conv2b dgrad", 64, 64, 38, 38, 57 * 40 + 4), ("nuc conv2a dgrad", 64, 32, 40, 40, 57 * 44), ("nuc conv3b", 128, 128, 18, 18, 57 * 20 + 4), ("nuc conv3a", 128, 64, 20, 20, 57 * 24),
                                ("soma conv2b", 64, 64, 18, 18, 121 * 20 + 4), ("soma conv2a", 64, 32, 20, 20, 121 * 24)]:
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn((1, cin, D, H, W), generator=g)).cuda()
    w = torch.relu(torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.05).cuda()
    wino = ops.WinoConv3d(w, two_d=True)
    t0 = timed(lambda: wino(x))
    if ops.ZwConv3d.supported(w):
        zw = ops.ZwConv3d(w)
        xm = ops.ZwConv3d.bound_of(x)
        om = torch.zeros(32, device="cuda")
        out = torch.empty((1, cout, D, H, W), device="cuda")
        t1 = timed(lambda: zw(x, xm, out=out, out_max=om))
        ts = timed(lambda: ops.ZwConv3d.bound_of(x))
        err = float((zw(x, xm)[0] - wino(x)).abs().max() / wino(x).abs().max())
    else:
        t1 = ts = err = float("nan")
    print("%-18s %3d->%3d %dx%dx%d  fp32 F(2x4) %.3f ms   f16x2 F(2,3)z %.3f ms  (bound sweep %.3f ms)  diff %.1e" % (name, cin, cout, D, H, W, t0, t1, ts, err), flush=True)
