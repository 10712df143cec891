#!/usr/bin/env python3
"""Sweep the compiled k=3 tile variants (M3D_TUNE_K3) over the backbone layers."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
layers32 = "conv2b,conv3a,conv3b"
layers16 = "conv4a,conv4b,rpn_conv"
for v in [0, 1, 5, 30, 31, 33]:
    env = dict(os.environ, M3D_TUNE_K3=str(v), LAYERS=layers32)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools/bench_layers.py"), "128", "10"], env=env, capture_output=True, text=True).stdout
    print("== variant", v); print("\n".join(l for l in out.splitlines() if l.startswith("conv")))
for v in [10, 11, 20, 21, 22, 23]:
    env = dict(os.environ, M3D_TUNE_K3=str(v), LAYERS=layers16)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools/bench_layers.py"), "128", "10"], env=env, capture_output=True, text=True).stdout
    print("== variant", v); print("\n".join(l for l in out.splitlines() if l.startswith("conv")))
