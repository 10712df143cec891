"""GPU: the two shipped tile geometries at full size (nuclei 64x200x200 stride-8 net, soma 64x160x160 stride-4 net):
size-independent properties of the whole path (no CPU oracle run at this size)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(cfg_name):
    from m3d.config import Cfg
    from m3d.synth import make_params, synth_volume
    from m3d.model import DetectorM3D
    from m3d.prm import PRMEngine
    from m3d import tiling
    cfg = Cfg.nuclei(mlp_dim=128) if cfg_name == "nuclei" else Cfg.soma(mlp_dim=128)
    P = make_params(stride=cfg.stride, num_anchors=cfg.num_anchors, mlp_dim=128, seed=1)
    det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
    S, H, W = cfg.in_size
    vol = tiling.norm1(synth_volume(3, (S, H, W)), np.float32).astype(np.float32)
    data = torch.from_numpy(vol)[None, None].cuda()
    return cfg, det, PRMEngine(det), data


@pytest.mark.parametrize("name", ["nuclei", "soma"])
def test_full_tile_detect_and_prm(name):
    cfg, det, eng, data = _run(name)
    S, H, W = cfg.in_size
    out = det.detect_tile(data)
    s = cfg.stride
    assert out["feat"].shape == (1, 256 if s == 8 else 128, S // s, H // s, W // s)
    assert torch.isfinite(out["feat"]).all()
    r = out["rois"]
    assert 0 < r.shape[0] <= cfg.post_nms_topN
    assert r[:, 1:].min() >= 0 and r[:, [1, 4]].max() <= W - 1 and r[:, [2, 5]].max() <= H - 1 and r[:, [3, 6]].max() <= S - 1
    p = out["roi_probs"].flatten()
    assert torch.all(p[:-1] >= p[1:])                                   # RPN NMS keeps score order
    assert out["det_scores"].numel() <= cfg.detections_per_im
    # feature map equals the torch (MIOpen) convolution stack within fp32 tolerance
    x = data
    import torch.nn.functional as F
    from m3d.model import dsn_layers
    for cname, bname, pool in dsn_layers(s):
        c, b = "Conv_Body." + cname, "Conv_Body." + bname
        k = det.P[c + ".weight"].shape[-1]
        x = F.conv3d(x.double(), det.P[c + ".weight"].double(), det.P[c + ".bias"].double(), 1, k // 2)
        x = F.batch_norm(x, det.P[b + ".running_mean"].double(), det.P[b + ".running_var"].double(), det.P[b + ".weight"].double(),
                         det.P[b + ".bias"].double(), False, 0.0, 1e-5)
        x = F.relu(x).float()
        if pool:
            x = F.max_pool3d(x, 2, 2)
    err = (out["feat"] - x).abs().max().item() / x.abs().max().item()
    assert err < 1e-4, err
    # PRM on the full tile: every kept peak gets a normalised, non-negative map supported inside its cone
    res = eng.prm_tile(data, dense=False)
    if res is not None and res.get("windows") is not None:
        w, sums, org = res["windows"], res["sums"], res["origins"]
        assert w.shape[0] == res["peaks"].shape[0] == res["dets"].shape[0] <= cfg.detections_per_im
        assert (w >= 0).all() and torch.isfinite(w).all()
        assert w.shape[1] == (84 if s == 8 else 40)
        tot = w.flatten(1).sum(1)
        assert torch.allclose(tot, sums, rtol=1e-3)
        pk = res["peaks"][:, 2:].to(torch.int32) * s                      # the peak's own voxel lies inside its window
        assert ((pk >= org) & (pk < org + w.shape[1])).all()
