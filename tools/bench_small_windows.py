"""3^3 / 5^3 / 7^3 window batches: direct windowed kernel vs the dense GEMM kernel (csrc/prm_small.hip)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd")]
import m3d  # noqa: E402


def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, P, cout_f, cin_f, n, dims in [("nuclei rpn", 67, 256, 256, 3, (8, 25, 25)), ("nuclei 4b", 67, 256, 256, 5, (8, 25, 25)),
                                        ("nuclei 4a", 67, 256, 128, 7, (8, 25, 25)), ("soma rpn", 128, 128, 128, 3, (16, 40, 40)),
                                        ("soma 3b", 128, 128, 128, 5, (16, 40, 40)), ("soma 3a", 128, 128, 64, 7, (16, 40, 40))]:
    w = torch.randn((cout_f, cin_f, 3, 3, 3), device="cuda") * 0.05
    gn = torch.rand((P, cout_f, n, n, n), device="cuda")
    full = torch.randn((cin_f,) + dims, device="cuda")
    off = full.min().reshape(1)
    org = torch.zeros((P, 3), dtype=torch.int32, device="cuda")
    d = m3d.PackedConv3d(w, m3d.W_DGRAD_RELU)
    s = m3d.SmallWindowDgrad(w, f16=False)
    s16 = m3d.SmallWindowDgrad(w)
    gf = 2.0 * P * n ** 3 * cout_f * cin_f * 27 / 1e9
    td = t(lambda: m3d.conv3d_windowed(d, gn, full, off, org))
    ts = t(lambda: s(gn, full, off, org))
    t16 = t(lambda: s16(gn, full, off, org))
    a, b = s16(gn, full, off, org), s(gn, full, off, org)
    print("%-11s P=%3d %3d->%3d %d^3: direct %.3f ms (%5.1f TF)   dense GEMM fp32 %.3f ms (%5.1f TF)   f16x2 %.3f ms (%5.1f TF fp32-equivalent; max diff %.1e of max)"
          % (name, P, cout_f, cin_f, n, td, gf / td, ts, gf / ts, t16, gf / t16, float((a - b).abs().max() / b.abs().max())), flush=True)
