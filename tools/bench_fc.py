#!/usr/bin/env python3
"""fc1 / fc2 of the box head: the split-K fp32 MFMA GEMM (csrc/fc_gemm.hip) vs the library GEMM torch dispatches to."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import m3d  # noqa: E402


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    N = 1024
    for K in (87808, 43904, 1024):
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        for M in (64, 128, 320, 640, 1000, 1280, 1281, 1313, 2560):
            x = torch.randn(M, K, device="cuda")
            t_own = timeit(lambda: m3d.linear(x, w, b, relu=True))
            t_lib = timeit(lambda: torch.relu(torch.nn.functional.linear(x, w, b)))
            fl = 2.0 * M * N * K
            print("K=%6d M=%5d  own %.3f ms %6.1f TF (%.0f%% of 157.3)   library %.3f ms %6.1f TF" %
                  (K, M, t_own, fl / t_own / 1e9, fl / t_own / 1e9 / 157.3 * 100, t_lib, fl / t_lib / 1e9), flush=True)


if __name__ == "__main__":
    main()
