import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import m3d
from m3d import _lib
_lib.tuning().__enter__()      # option sweeps: the tuning build (libm3d_tune.so) for the whole process
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
N, K = 1024, 87808
w = torch.randn(N, K, device="cuda") / K ** 0.5
b = torch.randn(N, device="cuda")
for M in (1280, 1281, 1313):
    x = torch.randn(M, K, device="cuda")
    _lib.set_option("tune_fc_slices", -1); _lib.set_option("tune_fc_slices_tail", -1)
    print("M=%d library plan: %.3f ms" % (M, timeit(lambda: m3d.linear(x, w, b, relu=True))), flush=True)
    for s in (5, 6):
        for st in (4, 5, 6):
            _lib.set_option("tune_fc_slices", s); _lib.set_option("tune_fc_slices_tail", st)
            t = timeit(lambda: m3d.linear(x, w, b, relu=True))
            print("M=%d s=%d st=%d: %.3f ms" % (M, s, st, t), flush=True)
