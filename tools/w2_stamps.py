#!/usr/bin/env python3
"""Where a workgroup of the eta-split Winograd kernel spends its cycles: prologue / K loop / eta exchange / epilogue.

Needs the diagnostic library (make -C instanceseg-without-voxelwise-labeling_amd/csrc w2_stamps) and a GPU:
    M3D_LIB_PATH=.../csrc/libm3d_w2stamps.so python tools/w2_stamps.py [layer ...]
One wave per workgroup stamps s_memtime at kernel entry, after the prologue barrier, after the K loop, after the exchange barrier
and after its last store (+ s_memrealtime at both ends for the clock).  Ideal K-loop cycles per chunk = 96 MFMAs per SIMD x 64 (eta-split kernel: two
waves of ONE workgroup per SIMD; quad kernel, M3D_TUNE_WINO2=399 or the default: a wave shares its SIMD with a wave of another workgroup, so
its 48 MFMAs per chunk take 6144 cycles when both workgroups are in their K loops)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d
from m3d._lib import lib

LAYERS = {"conv2a": (32, 64, 64, False), "conv2b": (64, 64, 64, True), "conv3a": (64, 128, 32, False), "conv3b": (128, 128, 32, True),
          "conv4a": (128, 256, 16, False), "conv4b": (256, 256, 16, False)}
BATCH = int(os.environ.get("BATCH", "4"))
L = lib()
if not hasattr(L, "m3d_debug_set_stamp_buffer"):
    sys.exit("not the stamps build: set M3D_LIB_PATH to libm3d_w2stamps.so")
for name in (sys.argv[1:] or ["conv2b", "conv2a", "conv3b", "conv4b"]):
    cin, cout, s, pool = LAYERS[name]
    x = torch.randn(BATCH, cin, s, s, s, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    sc = torch.rand(cout, device="cuda"); sh = torch.rand(cout, device="cuda")
    conv = m3d.WinoConv3d(w, two_d=True)
    out = torch.empty(BATCH, cout, s, s, s, device="cuda")
    run = (lambda: conv.pooled(x, scale=sc, shift=sh, relu=True)) if pool else (lambda: conv(x, scale=sc, shift=sh, relu=True, out=out))
    buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    for f in ("m3d_debug_set_stamp_buffer", "m3d_debug_set_stamp_buffer_q", "m3d_debug_set_stamp_buffer_24", "m3d_debug_set_stamp_buffer_24w"):   # every 2-D Winograd family
        getattr(L, f)(ctypes.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    for f in ("m3d_debug_set_stamp_buffer", "m3d_debug_set_stamp_buffer_q", "m3d_debug_set_stamp_buffer_24", "m3d_debug_set_stamp_buffer_24w"):
        getattr(L, f)(ctypes.c_void_p(0))
    ms = e0.elapsed_time(e1)
    st = buf.view(-1, 8).cpu()
    st = st[st[:, 0] != 0]
    n = st.shape[0]
    t = st[:, :5].double()
    seg = [(t[:, i + 1] - t[:, i]) for i in range(4)]
    tot = t[:, 4] - t[:, 0]
    clk = (tot / (st[:, 6] - st[:, 5]).double().clamp(min=1) * 0.1).median().item()     # GHz: cycles per 10 ns tick
    fam = L.m3d_conv3d_wino2_family()
    nchunk = cin // 2 if fam == 5 else cin // 4                    # family 5 stages one channel pair per chunk
    IDEAL = 2304 if fam == 5 else 4608 if fam == 4 else 6144        # MFMA cycles per chunk and SIMD: 36 / 72 (F(2x4)) or 96 (F(2x2)) x 64
    med = [x_.median().item() for x_ in seg]
    span = (t[:, 4].max() - t[:, 0].min()).item()
    print("%-7s batch %d: %d workgroups, kernel %.3f ms (with stamps), clock %.2f GHz" % (name, BATCH, n, ms, clk))
    print("   median cycles: prologue %6.0f | K loop %7.0f (%5.0f per chunk, ideal %d -> %.1f %%) | exchange %5.0f | epilogue %5.0f | total %7.0f"
          % (med[0], med[1], med[1] / nchunk, IDEAL, IDEAL * nchunk / med[1] * 100, med[2], med[3], tot.median().item()))
    print("   shares of a workgroup's time: prologue %.1f %%  loop %.1f %%  exchange %.1f %%  epilogue %.1f %%;  MFMA-ideal share of the total %.1f %%"
          % tuple([m_ / tot.median().item() * 100 for m_ in med] + [IDEAL * nchunk / tot.median().item() * 100]))
    print("   sum of workgroup times / (256 CUs x first-to-last span) = %.3f" % (tot.sum().item() / (256.0 * span)))
