import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import torch, m3d
torch.manual_seed(0)
for (B, cin, cout, D, H, W) in [(1, 32, 64, 6, 9, 64), (1, 8, 32, 2, 4, 64), (1, 64, 128, 6, 11, 32), (1, 8, 32, 4, 8, 64)]:
    x = torch.randn(B, cin, D, H, W)
    w = torch.randn(cout, cin, 3, 3, 3) * (2.0 / (cin * 27)) ** 0.5
    ref = torch.nn.functional.conv3d(x.double(), w.double(), None, 1, 1)
    refp = torch.nn.functional.max_pool3d(ref, 2, 2)
    conv = m3d.WinoConv3d(w.cuda(), two_d=True)
    y = conv(x.cuda()).cpu().double()
    yp = conv.pooled(x.cuda()).cpu().double()
    e = (yp - refp).abs()
    print((B, cin, cout, D, H, W), "plain err %.2e pooled err %.2e" % ((y - ref).abs().max(), e.max()))
    bad = (e > 1e-3)
    print("  bad fraction %.3f; by cout block:" % bad.float().mean().item(), [round(bad[:, c:c + 32].float().mean().item(), 2) for c in range(0, cout, 32)],
          " by z:", [round(bad[:, :, z].float().mean().item(), 2) for z in range(refp.shape[2])], " by y:", [round(bad[:, :, :, yy].float().mean().item(), 2) for yy in range(refp.shape[3])])
    print("  by x (first 16):", [round(bad[..., xx].float().mean().item(), 2) for xx in range(min(16, refp.shape[4]))])
    print("  sample got/ref:", yp[0, 0, 0, 0, :4].tolist(), refp[0, 0, 0, 0, :4].tolist())
