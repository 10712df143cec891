#!/usr/bin/env python3
"""Winograd-x conv kernel vs the direct MFMA kernel: error against fp64 on a sub-volume, and timing per tile variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = [("conv2a", 32, 64, size // 2, [0, 1]), ("conv2b", 64, 64, size // 2, [0, 1]),
     ("conv3a", 64, 128, size // 4, [4, 5, 6]), ("conv3b", 128, 128, size // 4, [4, 5, 6]),
     ("conv4a", 128, 256, size // 8, []), ("conv4b", 256, 256, size // 8, [])]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, cin, cout, s, variants in L:
    torch.manual_seed(0)
    x = torch.randn(1, cin, s, s, s, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * (2.0 / (cin * 27)) ** 0.5
    sc = torch.rand(cout, device="cuda") + 0.5; sh = torch.randn(cout, device="cuda")
    direct = m3d.PackedConv3d(w)
    wino = m3d.WinoConv3d(w)
    out = torch.empty(1, cout, s, s, s, device="cuda")
    fl = 2.0 * cin * cout * 27 * s ** 3
    os.environ.pop("M3D_TUNE_WINO", None)
    yd = direct(x, scale=sc, shift=sh, relu=True)
    ref = torch.relu(torch.nn.functional.conv3d(x[:, :, :10].double(), w.double(), None, 1, 1) * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1))[:, :, :8]
    ms_d = timeit(lambda: direct(x, scale=sc, shift=sh, relu=True, out=out))
    print("%-7s direct            %.3f ms %6.1f TF  err %.2e" % (name, ms_d, fl / ms_d / 1e9, (yd[:, :, :8].double() - ref).abs().max().item() / ref.abs().max().item()))
    w2 = m3d.WinoConv3d(w, two_d=True)
    for v2 in [None] + ([0, 1, 2] if s >= 48 else [3, 4, 5] if s >= 24 else []):
        if v2 is None:
            os.environ.pop("M3D_TUNE_WINO2", None)
        else:
            os.environ["M3D_TUNE_WINO2"] = str(v2)
        try:
            yw = w2(x, scale=sc, shift=sh, relu=True)
            err = (yw[:, :, :8].double() - ref).abs().max().item() / ref.abs().max().item()
            ms = timeit(lambda: w2(x, scale=sc, shift=sh, relu=True, out=out))
            print("%-7s wino2D variant %-4s %.3f ms %6.1f TF (algorithmic)  err vs fp64 %.2e" % (name, v2, ms, fl / ms / 1e9, err))
        except Exception as e:
            print("%-7s wino2D variant %-4s failed: %s" % (name, v2, e))
    os.environ.pop("M3D_TUNE_WINO2", None)
    if s >= 48:
        yp = w2.pooled(x, scale=sc, shift=sh, relu=True)
        errp = (yp - torch.nn.functional.max_pool3d(yd, 2, 2)).abs().max().item() / yd.abs().max().item()
        ms = timeit(lambda: w2.pooled(x, scale=sc, shift=sh, relu=True))
        print("%-7s wino2D +pool       %.3f ms %6.1f TF (algorithmic)  err vs direct+pool %.2e" % (name, ms, fl / ms / 1e9, errp))
    for v in ([None] + variants) if s >= 24 else []:
        if v is None:
            os.environ.pop("M3D_TUNE_WINO", None)
        else:
            os.environ["M3D_TUNE_WINO"] = str(v)
        try:
            yw = wino(x, scale=sc, shift=sh, relu=True)
            err = (yw[:, :, :8].double() - ref).abs().max().item() / ref.abs().max().item()
            errd = (yw - yd).abs().max().item() / yd.abs().max().item()
            ms = timeit(lambda: wino(x, scale=sc, shift=sh, relu=True, out=out))
            print("%-7s wino variant %-4s %.3f ms %6.1f TF (algorithmic)  err vs fp64 %.2e  vs direct %.2e" % (name, v, ms, fl / ms / 1e9, err, errd))
        except Exception as e:
            print("%-7s wino variant %-4s failed: %s" % (name, v, e))
