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
