"""GPU: batched, cone-cropped PRM back-propagation against the reference's autograd (golden) and the oracle."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def _engine(P, cfg):
    from m3d.model import DetectorM3D
    from m3d.prm import PRMEngine
    return PRMEngine(DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg), peak_chunk=7)


@pytest.mark.parametrize("tag", ["n", "s"])
def test_prm_golden(golden, tag):
    g = golden("prm_small_" + tag)
    stride, A = int(g["stride"]), int(g["A"])
    P = O.make_params(stride=stride, num_anchors=A, mlp_dim=64, seed=int(g["seed"]))
    cfg = O.Cfg(mlp_dim=64, score_thresh=0.0) if stride == 8 else O.Cfg.soma(mlp_dim=64)
    eng = _engine(P, cfg)
    data = torch.from_numpy(g["vol"]).cuda()
    feat, prob, deltas, saved, top = eng.forward(data)
    assert np.allclose(feat.cpu().numpy(), g["feat"], rtol=1e-4, atol=1e-4)
    assert np.allclose(prob.cpu().numpy(), g["crm"], rtol=1e-4, atol=1e-5)
    # (8) one-hot backward at the 3 fixture peaks == the reference's autograd through its hooks (data.grad)
    pk = torch.from_numpy(g["peaks"][:, 1:].astype(np.int32)).cuda()
    win, sums, origins = eng.backward_windows(pk, saved, top, data)
    import m3d
    dense = m3d.prm_scatter(win, torch.ones_like(sums), origins, g["vol"].shape[-3:]).cpu().numpy()
    for i in range(3):
        ref = np.clip(g["grads"][i][0, 0], 0, None)       # kernel stores clamp(min=0) of data.grad
        assert np.allclose(dense[i], ref, rtol=2e-3, atol=2e-6 * ref.max()), i
    # (9) the full forward tuple
    out = eng.prm_tile(data)
    assert np.array_equal(out["peaks"].cpu().numpy(), g["o_peaks"])
    assert np.allclose(out["dets"].cpu().numpy(), g["o_dets"], rtol=1e-4, atol=1e-3)
    assert np.allclose(out["prms"].cpu().numpy(), g["o_prms"], rtol=2e-3, atol=2e-6 * g["o_prms"].max())
    assert np.allclose(out["prms"].sum((1, 2, 3)).cpu().numpy(), 1.0, atol=1e-4)


def test_prm_vs_oracle_border_peaks():
    """Peaks whose cones stick out of the tile on every side (virtual-window handling)."""
    P = O.make_params(stride=8, num_anchors=35, mlp_dim=64, seed=5)
    cfg = O.Cfg(mlp_dim=64, score_thresh=0.0)
    eng = _engine(P, cfg)
    torch.manual_seed(1)
    vol = torch.randn(1, 1, 24, 40, 32)
    data = vol.cuda()
    feat, prob, deltas, saved, top = eng.forward(data)
    f2, p2, d2, osaved = O.prm_forward(P, cfg, vol)
    s_, h_, w_ = p2.shape[-3:]
    peaks = [(0, 0, 0, 0, 0), (0, 34, s_ - 1, h_ - 1, w_ - 1), (0, 7, 1, h_ - 1, 0), (0, 20, s_ // 2, h_ // 2, w_ // 2)]
    pk = torch.tensor([p[1:] for p in peaks], dtype=torch.int32).cuda()
    win, sums, origins = eng.backward_windows(pk, saved, top, data)
    import m3d
    dense = m3d.prm_scatter(win, sums, origins, vol.shape[-3:]).cpu()
    for i, p in enumerate(peaks):
        ref = O.prm_backward(P, osaved, p, p2.shape)[0]
        assert torch.allclose(dense[i], ref, rtol=2e-3, atol=2e-6 * float(ref.max())), i


def test_infer_prm_tiles_quantised_maps_and_tree(tmp_path):
    """m3d.infer.infer_prm (tools/infer_simple.py:176-247): norm1, slice padding, tiling, per-tile PRM, uint8 quantisation on
    device, instance tree on disk - every tile against the oracle run on the same crop."""
    from m3d.infer import infer_prm
    from m3d import io as mio, tiling
    P = O.make_params(stride=8, num_anchors=35, mlp_dim=32, seed=2)
    cfg = O.Cfg(mlp_dim=32, score_thresh=0.0)
    eng = _engine(P, cfg)
    rs = np.random.RandomState(3)
    im = (rs.rand(12, 40, 40) * 900 + 50).astype(np.uint16)          # 12 slices -> padded to the 16-slice patch (pad_s = 2)
    patch, overlap = (16, 24, 24), 8
    res = infer_prm(eng, im, dataset="nuclei", patch=patch, overlap=overlap, out_dir=str(tmp_path / "img"))
    vol = tiling.norm1(im, np.float64)
    vol, pad_s = tiling.pad_slices(vol, patch[0])
    assert pad_s == 2 and len(res) >= 1
    for r in res:
        s, h, w = r["start"]
        crop = torch.from_numpy(vol[s:s + 16, h:h + 24, w:w + 24].astype(np.float32))[None, None]
        ref = O.prm_tile(P, cfg, crop)
        assert ref[1] is not None and np.array_equal(r["peaks"], np.asarray(ref[1]))
        assert np.allclose(r["dets"], np.asarray(ref[3]), rtol=1e-4, atol=1e-3)
        for ch, fm in enumerate(np.asarray(ref[2])):
            q = O.quantize_prm_u8(fm.copy())[pad_s:pad_s + 12]
            d = np.abs(r["prm_u8"][ch].astype(int) - q.astype(int))
            assert r["prm_u8"][ch].shape == (12, 24, 24) and d.max() <= 1 and (d > 0).mean() < 5e-3     # fp32 map +-1e-3 -> +-1 level, rarely
        dets, prms = mio.load_prm_instances(str(tmp_path / "img" / "instances" / str(r["num"])))
        assert np.array_equal(dets, r["dets"]) and all(np.array_equal(a, b) for a, b in zip(prms, r["prm_u8"]))
