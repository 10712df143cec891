"""Would the F(2x2,3x3) kernel pay on the PRM window batches if the windows were laid side by side along x (one zero column
between neighbours)?  Times the direct kernel on [P,C,Wn,Wn,Wn] against the Winograd kernel on [1,C,Wn,Wn,P*(Wn+1)+1]."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd")]
from m3d import ops  # noqa: E402


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, P, cin, cout, Wn in [("2b", 67, 64, 64, 38), ("2a", 67, 64, 32, 40), ("3b", 67, 128, 128, 16), ("3a", 67, 128, 64, 18),
                               ("soma 2b", 128, 64, 64, 16), ("soma 2a", 128, 64, 32, 18)]:
    w = torch.randn((cout, cin, 3, 3, 3), device="cuda") * 0.05
    x = torch.randn((P, cin, Wn, Wn, Wn), device="cuda")
    xc = torch.randn((1, cin, Wn, Wn, P * (Wn + 1) + 1), device="cuda")
    d = ops.PackedConv3d(w)
    wn = ops.WinoConv3d(w, two_d=True)
    gf = 2.0 * P * Wn ** 3 * cin * cout * 27 / 1e9
    td = t(lambda: d(x))
    tw = t(lambda: wn(xc))
    print("%-8s P=%d %d->%d %d^3: direct %.3f ms (%.0f TF)   wino2 on x-concatenated %.3f ms (%.0f TF alg)" %
          (name, P, cin, cout, Wn, td, gf / td, tw, gf / tw), flush=True)
