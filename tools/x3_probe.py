"""PMC probe of the bf16x3 direct conv (csrc/conv3d_x3.hip): a few launches of the PRM norm-conv shapes.
usage: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -- python3 tools/x3_probe.py"""
import sys; sys.path.insert(0, "/root/repo"); import __graft_entry__  # noqa
import torch
from m3d import ops
for cin, cout, shape in [(64, 64, (32, 80, 80)), (128, 128, (16, 40, 40))]:
    g = torch.Generator().manual_seed(1)
    x = torch.rand((1, cin) + shape, generator=g).cuda()
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.1).cuda()
    off = ops.reduce_min(x)
    conv = ops.X3Conv3d(w, ops.W_RELU)
    for _ in range(4):
        conv(x, in_offset=off)
    torch.cuda.synchronize()
