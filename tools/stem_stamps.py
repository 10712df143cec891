#!/usr/bin/env python3
"""Where a workgroup of the stem 'rows' kernel (conv3d_stem_wino4_kernel) spends its cycles: prologue, then K loop / epilogue of each of
its four rows.  Needs the diagnostic library (make -C instanceseg-without-voxelwise-labeling_amd/csrc w2_stamps) and a GPU:
    M3D_LIB_PATH=.../csrc/libm3d_w2stamps.so python tools/stem_stamps.py
Ideal K-loop cycles per row = 78 MFMAs x 64 x 2 waves per SIMD = 9984 when both workgroups of the CU are in their loops."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d
from m3d._lib import lib

BATCH = int(os.environ.get("BATCH", "4"))
L = lib()
if not hasattr(L, "m3d_debug_set_stamp_buffer_stem"):
    sys.exit("not the stamps build: set M3D_LIB_PATH to libm3d_w2stamps.so")
for pool in (True, False):
    x = torch.randn(BATCH, 1, 128, 128, 128, device="cuda")
    w = torch.randn(32, 1, 5, 5, 5, device="cuda") * 0.05
    sc = torch.rand(32, device="cuda"); sh = torch.rand(32, device="cuda")
    conv = m3d.StemWinoConv3d(w)
    out = torch.empty(BATCH, 32, 128, 128, 128, device="cuda")
    run = (lambda: conv.pooled(x, scale=sc, shift=sh, relu=True)) if pool else (lambda: conv(x, scale=sc, shift=sh, relu=True, out=out))
    buf = torch.zeros(1 << 21, dtype=torch.int64, device="cuda")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.m3d_debug_set_stamp_buffer_stem(ctypes.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    L.m3d_debug_set_stamp_buffer_stem(ctypes.c_void_p(0))
    ms = e0.elapsed_time(e1)
    st = buf.view(-1, 16).cpu()
    st = st[st[:, 0] != 0]
    t = st[:, :12].double()
    seg = [(t[:, i + 1] - t[:, i]).median().item() for i in range(9)]
    tot = (t[:, 11] - t[:, 0])
    nrows = int(os.environ.get("ROWS", "16"))
    clk = (tot / (st[:, 13] - st[:, 12]).double().clamp(min=1) * 0.1).median().item()
    span = (t[:, 11].max() - t[:, 0].min()).item()
    print("conv1a%s batch %d: %d workgroups, kernel %.3f ms (with stamps), clock %.2f GHz" % ("+pool" if pool else "", BATCH, st.shape[0], ms, clk))
    print("   median cycles: prologue %6.0f | rows (K loop / epilogue): %s | total %7.0f" %
          (seg[0], "  ".join("%5.0f / %4.0f" % (seg[1 + 2 * r], seg[2 + 2 * r]) for r in range(4)), tot.median().item()))
    loops = sum(seg[1 + 2 * r] for r in range(4)); epi = sum(seg[2 + 2 * r] for r in range(4))
    print("   first four rows of %d: prologue %.1f %% of the total, K loop %.0f per row (9984 = two waves sharing the SIMD at full rate), epilogue %.0f per row; "
          "MFMA-ideal share of a workgroup's time (2 waves x rows x 4992 / total) = %.1f %%;  sum of workgroup times / (512 slots x span) = %.3f"
          % (nrows, seg[0] / tot.median().item() * 100, loops / 4, epi / 4, 2 * nrows * 4992 / tot.median().item() * 100, tot.sum().item() / (512.0 * span)))
