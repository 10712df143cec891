#!/usr/bin/env python3
"""Per-tap timeline of the f16x2 conv kernel (csrc/conv3d_zw.hip) from s_memtime stamps of wave 0 of every workgroup in its second unit.
Needs the diagnostic library (make -C instanceseg-without-voxelwise-labeling_amd/csrc zw_stamps) and a GPU:
    M3D_LIB_PATH=.../csrc/libm3d_zwstamps.so python tools/zw_stamps.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import numpy as np, torch
from m3d import ops, _lib
L = _lib.lib()
if not hasattr(L, "m3d_debug_set_stamp_buffer_zw"):
    sys.exit("not the stamps build: set M3D_LIB_PATH to libm3d_zwstamps.so")
for name, cin, cout, S, pool in [("conv2a", 32, 64, 64, False), ("conv2b+pool", 64, 64, 64, True), ("conv3b+pool", 128, 128, 32, True), ("conv4b", 256, 256, 16, False)]:
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn((4, cin, S, S, S), generator=g)).cuda()
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * (2.0 / (cin * 27)) ** 0.5).cuda()
    zw = ops.ZwConv3d(w)
    xm = ops.ZwConv3d.bound_of(x)
    buf = torch.zeros((256, 256), dtype=torch.int64, device="cuda")
    L.m3d_debug_set_stamp_buffer_zw(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        zw(x, xm, relu=True, pool=pool)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64)
    chunks = cin // 16
    ok = t[:, 0] > 0
    if ok.sum() == 0:
        print(name, "no second unit (one unit per workgroup)"); continue
    t = t[ok]
    taps = t[:, 1:1 + chunks * 9]
    d = np.diff(np.concatenate([t[:, :1], taps, t[:, 200:201], t[:, 202:203]], 1), axis=1)          # unit start -> tap 0 -> ... -> K loop end -> unit end
    med = np.median(d, 0)
    # s_memtime counts at 100 MHz on this part: print in ns
    ns = med * 10.0
    print("%-12s %d workgroups; ns (median): prologue %.0f | taps of chunk 0: %s | chunk means: %s | K loop end -> unit end (epilogue) %.0f | unit %.0f"
          % (name, int(ok.sum()), ns[0], " ".join("%.0f" % v for v in ns[1:10]),
             " ".join("%.0f" % ns[1 + 9 * c:10 + 9 * c].sum() for c in range(chunks)), ns[-1], ns.sum()), flush=True)
    print("             per tap position (mean over chunks): %s" % " ".join("%.0f" % np.mean([ns[1 + 9 * c + k] for c in range(chunks) if 1 + 9 * c + k < len(ns) - 1]) for k in range(9)))
