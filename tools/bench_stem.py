"""Times m3d_prm_stem_dgrad_fused alone on synthetic inputs (nuclei tile: 67 peaks, U = 40; soma: 128 peaks, U = 18).
M3D_LIB_PATH selects an ablation build (make -C csrc stem_variants)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd")]
from m3d import ops  # noqa: E402


def run(P, U, D, H, W, reps=5):
    g = torch.Generator(device="cuda").manual_seed(0)
    UD, UH, UW = D // 2, H // 2, W // 2
    gup = torch.rand((P, 32, U, U, U), device="cuda", generator=g)
    org = torch.stack([torch.randint(-6, 1, (P,), device="cuda", generator=g),
                       torch.randint(0, max(1, UH - U), (P,), device="cuda", generator=g),
                       torch.randint(0, max(1, UW - U), (P,), device="cuda", generator=g)], 1).to(torch.int32).contiguous()
    den = torch.rand((32, UD, UH, UW), device="cuda", generator=g) + 0.5
    am = torch.randint(0, 8, (32, UD, UH, UW), device="cuda", generator=g, dtype=torch.uint8)
    scale = torch.rand((32,), device="cuda", generator=g) + 0.5
    wa = ops.prm_stem_mfma_weights(torch.randn((32, 1, 5, 5, 5), device="cuda", generator=g))
    data = torch.rand((D, H, W), device="cuda", generator=g)
    off = torch.zeros((1,), device="cuda")
    for _ in range(2):
        ops.prm_stem_dgrad_fused(gup, org, den, am, scale, wa, data, off)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.prm_stem_dgrad_fused(gup, org, den, am, scale, wa, data, off)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    Wn = 2 * U + 4
    gf = P * Wn ** 3 * 4000 * 2 / 1e9
    print("%s P=%d U=%d: %.3f ms  %.1f TF algorithmic  (%.1f TF issued at 32/25)" %
          (os.path.basename(os.environ.get("M3D_LIB_PATH", "libm3d.so")), P, U, ms, gf / ms, gf / ms * 32 / 25), flush=True)


if __name__ == "__main__":
    run(67, 40, 64, 200, 200)
    run(128, 18, 64, 160, 160)
