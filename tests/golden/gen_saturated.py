#!/usr/bin/env python3
"""tests/golden/prm_saturated.npz: the DEGENERATE peak, pinned by running the reference's own code (CPU, oracle/ref_harness.py).

lib/prm/peak_response_mapping_3d.py:170-171 returns `prm / prm.sum()`.  Where the RPN sigmoid of a kept peak saturates to exactly
1.0f its derivative y (1 - y) is exactly 0, the back-propagated map is all zero and the returned map is 0 / 0 = NaN at every voxel.
tools/infer_simple.py:233-240 then quantises that map ((fm - min) / max * 255 -> uint8) and writes it as a TIFF page.  This script
  1. builds the small nuclei net of gen_golden.py with RPN.RPN_cls_score scaled by SCALE (saturating a part of the kept peaks),
  2. runs the reference's PeakResponseMapping_3d.forward on a [1,1,16,24,24] volume (RoIAlign3D plugged from the oracle, as in
     gen_golden.py: the reference's own is CUDA-only),
  3. EXECUTES the statements of tools/infer_simple.py:233-238 from where they lie on the returned maps,
and stores inputs and outputs: the volume, the scale, the peaks / dets, the float32 maps (NaN included) and the uint8 volumes.
Nothing of the reference's text is stored.   Run in the build container only:   python tests/golden/gen_saturated.py
"""
import os
import sys
import textwrap
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as H  # noqa: E402
import oracle as O  # noqa: E402

NUC = 'configs/cell_tracking_baseline/e2e_mask_rcnn_N3DH_SIM_dsn_body.yaml'
# (seed, scale) candidates found by a search over the ORACLE's RPN forward for nets where exactly ONE anchor's sigmoid is 1.0f and all
# 630 RPN scores are distinct: the reference sorts proposals with NumPy's unstable argsort()[::-1] (generate_proposals_3d.py:135-146), so a
# fixture with tied scores - several saturated anchors tie at 1.0f by definition - would pin NumPy's sort internals, not the algorithm
# (SURVEY 8c caveat i).  The first candidate whose saturated anchor survives to the kept detections is stored.
CANDIDATES = ((79, 1.5), (152, 1.5), (154, 1.75))


def ref_lines(rel, first, last):
    with open(os.path.join(H.REF, rel)) as f:
        lines = f.read().split("\n")
    return textwrap.dedent("\n".join(lines[first - 1:last]))


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(4)

    def roi_align_plug(features, rois, AS, AH, AW, scale, ratio):
        return torch.from_numpy(O.roi_align_3d_forward(features.detach().numpy(), rois.detach().numpy(), AS, AH, AW, scale, ratio))

    H.install(roi_align_plug)
    H.load_cfg(NUC)
    import core.config as CC
    from prm.peak_response_mapping_3d import PeakResponseMapping_3d
    CC.cfg.FAST_RCNN.MLP_HEAD_DIM = 64
    CC.cfg.PRM_ON = True
    CC.cfg.TEST.SCORE_THRESH = 0.0
    rng = np.random.RandomState(3)
    vol = (rng.randn(1, 1, 16, 24, 24) * 1.0 + 0.2).astype(np.float32)
    im_info = torch.from_numpy(np.array([[16., 24., 24., 1.0]]))
    for SEED, SCALE in CANDIDATES:
        P = O.make_params(stride=8, num_anchors=35, mlp_dim=64, seed=SEED)
        P["RPN.RPN_cls_score.weight"] = P["RPN.RPN_cls_score.weight"] * SCALE
        P["RPN.RPN_cls_score.bias"] = P["RPN.RPN_cls_score.bias"] * SCALE
        model = PeakResponseMapping_3d()
        sd = model.state_dict()
        for k_, v_ in P.items():
            assert k_ in sd, k_
            sd[k_] = v_.clone()
        model.load_state_dict(sd)
        model.inference()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            agg, crm, vpl, prms, dets = model(torch.from_numpy(vol.copy()), im_info, 1.0)
        if prms is None:
            continue
        prm = prms.numpy().astype(np.float32)
        nan_maps = np.array([bool(np.isnan(m).all()) for m in prm])
        fin_maps = np.array([bool(np.isfinite(m).all()) for m in prm])
        scores = crm.numpy().ravel()
        print("seed %d scale %.1f: peaks %d, %d all-NaN maps, %d finite maps, %d saturated anchors, scores distinct: %s"
              % (SEED, SCALE, len(prm), int(nan_maps.sum()), int(fin_maps.sum()), int((scores == 1.0).sum()), len(np.unique(scores)) == scores.size))
        if nan_maps.any() and fin_maps.any() and bool((nan_maps | fin_maps).all()) and len(np.unique(scores)) == scores.size:
            break
    else:
        raise SystemExit("no candidate gives both kinds of map")
    pk = vpl.numpy()
    sat = np.array([float(crm[0, a, s, h, w]) == 1.0 for _, a, s, h, w in pk])
    assert np.array_equal(sat, nan_maps), "a map is NaN exactly where the peak's sigmoid is 1.0f"
    # tools/infer_simple.py:233-238, executed from where they lie; the loop body's result is collected after its last statement
    env = {"np": np, "prm": prm.copy(), "u8": []}
    body = ref_lines("tools/infer_simple.py", 233, 238)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exec(body + "\n    u8.append(fm_ch)", env)
    u8 = np.stack(env["u8"])
    assert u8.dtype == np.uint8 and u8.shape == prm.shape
    p = os.path.join(HERE, "prm_saturated.npz")
    # float maps of the NaN peaks and of the first 7 regular ones (the uint8 volumes of all): keeps the fixture small
    idx = np.concatenate((np.nonzero(nan_maps)[0], np.nonzero(~nan_maps)[0][:7])).astype(np.int64)
    np.savez_compressed(p, vol=vol, scale=np.float64(SCALE), seed=np.int64(SEED), o_peaks=pk, o_dets=dets.numpy(),
                        o_prms_idx=idx, o_prms=prm[idx], o_u8=u8, nan_maps=nan_maps)
    print("wrote prm_saturated.npz (%.1f KB); NaN maps quantise to bytes %s" % (os.path.getsize(p) / 1024, np.unique(u8[nan_maps]).tolist()))


if __name__ == "__main__":
    main()
