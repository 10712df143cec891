#!/usr/bin/env python3
"""BASELINE.json configs[0]: one synthetic 1x64x64x64 volume through e2e_mask_rcnn_N3DH_SIM_dsn_body.yaml on the REFERENCE's
own CPU path (oracle/ref_harness.py; TEST.IN_SIZE (64,64,64), MLP_HEAD_DIM 1024 as in the YAML), PRM_ON False
(Generalized_RCNN + lib/core/test.py im_detect_bbox / box_results_with_nms_and_limit) and then True
(PeakResponseMapping_3d.forward).  Writes tests/golden/cfg0_64.npz: inputs are re-created from seeds (m3d.synth), outputs are
stored compactly (full class response map, rois, detections, peaks; feature map and peak response maps as projections).

Run in the build container only:   python tests/golden/gen_cfg0.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import ref_harness as H  # noqa: E402
import oracle as O  # noqa: E402

NUC = 'configs/cell_tracking_baseline/e2e_mask_rcnn_N3DH_SIM_dsn_body.yaml'
SEED_P, SEED_V, SIZE = 0, 0, 64


def main():
    import torch
    from m3d.synth import make_params, synth_volume
    torch.manual_seed(0)
    torch.set_num_threads(8)

    def roi_align_plug(features, rois, AS, AH, AW, scale, ratio):
        return torch.from_numpy(O.roi_align_3d_forward(features.detach().numpy(), rois.detach().numpy(), AS, AH, AW, scale, ratio))

    H.install(roi_align_plug)
    H.load_cfg(NUC)
    import core.config as CC
    from core.test import im_detect_bbox, box_results_with_nms_and_limit
    from modeling.model_builder import Generalized_RCNN
    from prm.peak_response_mapping_3d import PeakResponseMapping_3d
    import utils.blob as blob_utils
    CC.cfg.TEST.IN_SIZE = (SIZE, SIZE, SIZE)
    sys.path.insert(0, os.path.dirname(HERE))
    from conftest import cfg0_params                              # the same weights the tests rebuild
    P = cfg0_params(CC.cfg.FAST_RCNN.MLP_HEAD_DIM, SEED_P)
    raw = synth_volume(SEED_V, (SIZE, SIZE, SIZE))
    # norm1 exactly as the detection branch does it (lib/utils/blob.py:179-184 via prep_im_for_blob)
    im = raw.astype(np.float32, copy=False)
    mask = im > 0
    vol = ((im - np.mean(im[mask])) / np.std(im[mask])).astype(np.float32)

    def load(model):
        sd = model.state_dict()
        for k_, v_ in P.items():
            assert k_ in sd, k_
            sd[k_] = v_.clone()
        model.load_state_dict(sd)
        model.eval()
        return model

    out = {}
    # ---- PRM_ON False: detection mode (model_builder.py:151-240 + core/test.py:194-263,806-883)
    CC.cfg.PRM_ON = False
    net = load(Generalized_RCNN())

    def model(data, im_info):
        # what mynn.DataParallel(minibatch=True, cpu_keywords=['im_info','roidb']) does around the module on one device
        # (lib/nn/parallel/data_parallel.py:78-116): hand over the list elements, gather NumPy results as tensors
        ret = net(data[0], im_info[0])
        return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in ret.items()}
    inputs = {"data": vol[None, None].copy(), "im_info": np.array([[SIZE, SIZE, SIZE, 1.0]])}
    scores, boxes, _, blob_conv = im_detect_bbox(model, inputs, 1.0)
    sc, bx, cls_boxes, _ = box_results_with_nms_and_limit(scores, boxes)
    feat = blob_conv.detach().numpy()
    out.update(d_scores=scores.astype(np.float32), d_pred_boxes=boxes.astype(np.float32), d_cls1=cls_boxes[1].astype(np.float32),
               feat_sum_c=feat.sum(1)[0].astype(np.float32), feat_sample=feat.ravel()[::997].astype(np.float32),
               feat_absmax=np.float32(np.abs(feat).max()))
    # ---- PRM_ON True: PeakResponseMapping_3d.forward (peak_response_mapping_3d.py:85-193)
    CC.cfg.PRM_ON = True
    pm = load(PeakResponseMapping_3d())
    pm.inference()
    data = torch.from_numpy(vol[None, None].copy())
    im_info = torch.from_numpy(np.array([[SIZE, SIZE, SIZE, 1.0]]))
    agg, crm, vpl, prms, dets = pm(data, im_info, 1.0)
    prms = prms.numpy().astype(np.float32)
    out.update(crm=crm.numpy().astype(np.float32), p_peaks=vpl.numpy().astype(np.int64), p_dets=dets.numpy().astype(np.float64),
               p_prm_sum=prms.sum((1, 2, 3)), p_prm_z=prms.sum((2, 3)), p_prm_y=prms.sum((1, 3)), p_prm_x=prms.sum((1, 2)),
               p_prm_max=prms.reshape(len(prms), -1).max(1), p_prm_argmax=prms.reshape(len(prms), -1).argmax(1).astype(np.int64),
               # round 6: per-voxel values - every map's 2048 largest voxels (flat index ascending among equal values) and a strided sample
               p_prm_top_idx=np.stack([np.argsort(-m.ravel(), kind="stable")[:2048] for m in prms]).astype(np.int32),
               p_prm_top_val=np.stack([m.ravel()[np.argsort(-m.ravel(), kind="stable")[:2048]] for m in prms]).astype(np.float32),
               p_prm_stride_val=np.stack([m.ravel()[::257] for m in prms]).astype(np.float32),
               distinct_scores=np.int64(len(np.unique(crm.numpy())) == crm.numel()),
               seed_params=np.int64(SEED_P), seed_volume=np.int64(SEED_V), size=np.int64(SIZE))
    p = os.path.join(HERE, "cfg0_64.npz")
    np.savez_compressed(p, **out)
    print("wrote cfg0_64.npz (%.1f KB): %d rois scored, %d detections, %d peaks" %
          (os.path.getsize(p) / 1024, len(scores), len(cls_boxes[1]), len(vpl)))


if __name__ == "__main__":
    main()
