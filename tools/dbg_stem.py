import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import m3d
from m3d import _lib
g = torch.Generator().manual_seed(4)
x = torch.randn(2, 1, 19, 21, 75, generator=g)
w = torch.randn(40, 1, 5, 5, 5, generator=g) * (2.0 / 125) ** 0.5
sc = torch.rand(40, generator=g) - 0.3
sh = torch.randn(40, generator=g)
for tune in (4, 8):
    _lib.set_option("tune_stem", tune)
    conv = m3d.StemWinoConv3d(w.cuda())
    y = conv(x.cuda(), scale=sc.cuda(), shift=sh.cuda(), relu=True)
    yp = conv.pooled(x.cuda(), scale=sc.cuda(), shift=sh.cuda(), relu=True)
    r = torch.nn.functional.max_pool3d(y, 2, 2)
    d = (yp - r).abs()
    print(tune, "max diff", d.max().item(), "n diff", (d > 0).sum().item(), "of", d.numel())
    idx = (d > 0).nonzero()
    print(idx[:10].tolist()); print(idx[-5:].tolist())
    ch = (d > 0).sum(dim=(0, 2, 3, 4)); print("per channel", ch.tolist())
    print("per z", (d > 0).sum(dim=(0, 1, 3, 4)).tolist()); print("per y", (d > 0).sum(dim=(0, 1, 2, 4)).tolist())
    b_, c_, pz, py, px = idx[0].tolist()
    print("window", y[b_, c_, 2*pz:2*pz+2, 2*py:2*py+2, 2*px:2*px+2].flatten().tolist(), "pooled", yp[b_, c_, pz, py, px].item(), "sc", sc[c_].item(), "sh", sh[c_].item())
    y2 = conv(x.cuda(), scale=None, shift=None, relu=False)
    yp2 = conv.pooled(x.cuda(), scale=None, shift=None, relu=False)
    print("no affine: equal", torch.equal(yp2, torch.nn.functional.max_pool3d(y2, 2, 2)))
    y3 = conv(x.cuda(), scale=sc.cuda().abs(), shift=None, relu=False)
    yp3 = conv.pooled(x.cuda(), scale=sc.cuda().abs(), shift=None, relu=False)
    print("positive scale only: equal", torch.equal(yp3, torch.nn.functional.max_pool3d(y3, 2, 2)))
    y4 = conv(x.cuda(), scale=None, shift=sh.cuda(), relu=False)
    yp4 = conv.pooled(x.cuda(), scale=None, shift=sh.cuda(), relu=False)
    print("shift only: equal", torch.equal(yp4, torch.nn.functional.max_pool3d(y4, 2, 2)))
