"""A/B of the two box-head GEMM kernels at the fc1 shape: fp32 MFMA (m3d_linear_forward) vs bf16x3 split (m3d_linear_bf16x3_forward).
usage: python tools/bench_fc.py [M ...]"""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd")]
from m3d import ops as m3d  # noqa: E402
from m3d import _lib  # noqa: E402
_lib.tuning().__enter__()      # option sweeps: the tuning build (libm3d_tune.so) for the whole process


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    Ms = [int(v) for v in sys.argv[1:]] or [1200, 1281, 300]
    N, K = 1024, int(os.environ.get("FC_K", "87808"))
    w = (torch.randn(N, K) / np.sqrt(K)).cuda()
    b = torch.randn(N).cuda()
    lin = m3d.SplitLinear(w, b)
    F16 = [m3d.SplitLinearF16(w, b)]
    for M in Ms:
        x = torch.randn(M, K).cuda()
        t32 = timed(lambda: m3d.linear(x, w, b, relu=True))
        ref = torch.relu(x[:64].double() @ w.double().t() + b.double())
        e32 = (m3d.linear(x, w, b, relu=True)[:64].double() - ref).abs().max().item() / ref.abs().max().item()
        fl = 2.0 * M * N * K
        print("M=%5d  fp32 MFMA %.3f ms (%.1f TF, err %.2e)" % (M, t32, fl / t32 / 1e9, e32), flush=True)
        t3 = timed(lambda: lin(x, relu=True, variant="w32"))
        e3 = (lin(x, relu=True, variant="w32")[:64].double() - ref).abs().max().item() / ref.abs().max().item()
        print("         bf16x3 256 x 256 tiles, fp32 W: %.3f ms (%.1f TF fp32-equivalent, %.0f TF bf16 issued, err %.2e)"
              % (t3, fl / t3 / 1e9, 6 * fl / t3 / 1e9, e3), flush=True)
        _lib.set_option("tune_fc_x3_rows", -1)
        t3 = timed(lambda: lin(x, relu=True))
        e3 = (lin(x, relu=True)[:64].double() - ref).abs().max().item() / ref.abs().max().item()
        eall = (lin(x, relu=True)[-64:].double() - torch.relu(x[-64:].double() @ w.double().t() + b.double())).abs().max().item() / ref.abs().max().item()
        print("         bf16x3 as the model calls it (variant by shape): %.3f ms (%.1f TF fp32-equivalent, err %.2e / last rows %.2e)"
              % (t3, fl / t3 / 1e9, e3, eall), flush=True)
        # f16x2 split (round 6): three products per fp32 product; x = relu-like data as the RoIAlign output is, bound from a sweep / given
        _lib.set_option("tune_fc_x3_rows", -1)
        lf = F16[0]
        xb = m3d.absmax(x)
        rms = lambda a_: float((a_.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        for rows in (-1, 128, 256, 512):
            _lib.set_option("tune_fc_x3_rows", rows)
            tf = timed(lambda: lf(x, relu=True, x_bound=xb))
            got = lf(x, relu=True, x_bound=xb)[:64]
            ef = (got.double() - ref).abs().max().item() / ref.abs().max().item()
            print("         f16x2 rows %4s, bound given: %.3f ms (%.1f TF fp32-equivalent, %.0f TF f16 issued, err %.2e; rms err %.2e, fp32 kernel %.2e, bf16x3 %.2e)"
                  % (rows if rows > 0 else "auto", tf, fl / tf / 1e9, 3 * fl / tf / 1e9, ef, rms(got), rms(m3d.linear(x, w, b, relu=True)[:64]),
                     rms(lin(x, relu=True)[:64])), flush=True)
        _lib.set_option("tune_fc_x3_rows", -1)
        tf = timed(lambda: lf(x, relu=True))
        print("         f16x2 auto, bound swept from x inside the call: %.3f ms" % tf, flush=True)
        for rows in (128, 256):
            _lib.set_option("tune_fc_x3_rows", rows)
            t3 = timed(lambda: lin(x, relu=True, variant="packed"))
            e3 = (lin(x, relu=True, variant="packed")[:64].double() - ref).abs().max().item() / ref.abs().max().item()
            print("         bf16x3 rows %4s: %.3f ms (%.1f TF fp32-equivalent, %.0f TF bf16 issued, err %.2e)"
                  % (rows if rows > 0 else "auto", t3, fl / t3 / 1e9, 6 * fl / t3 / 1e9, e3), flush=True)


if __name__ == "__main__":
    main()
