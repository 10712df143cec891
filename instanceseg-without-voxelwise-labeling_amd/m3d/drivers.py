"""Driver boundary: callables with the reference's own call shape and return shape, on the HIP path.

  Generalized_RCNN          model(data=[T], im_info=[T]) -> return_dict {'blob_conv','rois','cls_score','bbox_pred'}
                            (lib/modeling/model_builder.py:140-240, eval branch :228-238)
  PeakResponseMapping_3d    model(data=[T], im_info=[T], im_scale=[1.0]) ->
                            (None, class_response_maps [1,A,s,h,w], valid_peak_list int64 [P,5], peak_response_maps
                            [P,S,H,W], dets float64 [P,7])   or   (None,)*5
                            (lib/prm/peak_response_mapping_3d.py:85-193; driver tools/infer_simple.py:217-226)
  im_detect_bbox / im_detect_all   lib/core/test.py:194-263 / :54-177 with the reference signatures and NumPy returns

The reference wraps its model in mynn.DataParallel(minibatch=True, cpu_keywords=['im_info','roidb'])
(tools/infer_simple.py:152), which hands each GPU the list ELEMENT; the callables here therefore accept both the
list-wrapped kwargs the drivers build and bare tensors.  Tensors may live on the host (they are what
torch.from_numpy gives the reference driver) — the H2D copy happens here, as DataParallel's scatter does there.
"""
import numpy as np
import torch

from . import ops, tiling
from .model import DetectorM3D
from .prm import PRMEngine
from .mask_head import MaskHeadM3D, im_detect_mask as _im_detect_mask, segm_results as _segm_results


def _first(x):
    return x[0] if isinstance(x, (list, tuple)) else x


def _im_info_row(im_info):
    """im_info: [[S,H,W,scale]] tensor / ndarray (float64 in the reference driver) -> float64 ndarray [4]."""
    v = _first(im_info)
    v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    return np.asarray(v, dtype=np.float64).reshape(-1, 4)[0]


def _to_params(state, device):
    """Accept a checkpoint dict ({'model': state_dict}, infer_simple.py:142-150) or a bare state dict; tensors -> fp32 CUDA.
    Keys keep the reference layout (Conv_Body.conv1a.weight ... Box_Outs.bbox_pred.bias; a DataParallel 'module.' prefix
    is dropped)."""
    if "model" in state and isinstance(state["model"], dict):
        state = state["model"]
    out = {}
    for k, v in state.items():
        if k.startswith("module."):
            k = k[7:]
        if torch.is_tensor(v):
            out[k] = v.detach().to(device=device, dtype=torch.float32).contiguous() if v.is_floating_point() else v.to(device)
    return out


class Generalized_RCNN(object):
    """Detection-mode model with the reference's call shape (model_builder.py:140-240, eval mode only)."""

    def __init__(self, state_dict, cfg, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        params = _to_params(state_dict, self.device)
        self.det = DetectorM3D(params, cfg)
        self.training = False
        # MODEL.MASK_ON checkpoints carry a mask branch (model_builder.py:111-116); both shipped configs have it off
        self.mask_head = MaskHeadM3D(params, cfg) if "Mask_Head.upconv.weight" in params else None

    def mask_net(self, blob_conv, rpn_blob):
        """model_builder.py:327-331."""
        if self.mask_head is None:
            raise AttributeError("this checkpoint has no Mask_Head / Mask_Outs weights (MODEL.MASK_ON False)")
        return self.mask_head.mask_net(blob_conv, rpn_blob)

    # nn.Module look-alikes the reference drivers call on the model
    def eval(self):
        return self

    def cuda(self, *a, **k):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the HIP path is inference-only (SURVEY 2: the training loop is out of scope)")
        return self

    @property
    def module(self):               # model.module.<...> as under DataParallel
        return self

    def _data(self, data):
        x = _first(data)
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x))
        assert x.dim() == 5, "data must be [1,1,S,H,W]"
        return x.to(device=self.device, dtype=torch.float32, non_blocking=True).contiguous()

    def __call__(self, data, im_info, roidb=None, **rpn_kwargs):
        return self.forward(data, im_info, roidb, **rpn_kwargs)

    def forward(self, data, im_info, roidb=None, **rpn_kwargs):
        x = self._data(data)
        info = _im_info_row(im_info)
        out = self.det.detect_tile(x, info)
        R = out["rois"].shape[0]
        dev = x.device
        nc = self.cfg.num_classes
        return {"blob_conv": out["feat"], "rois": out["rois"],                                         # :166,:231
                "cls_score": out["cls"] if "cls" in out else torch.zeros((R, nc), device=dev),          # :233
                "bbox_pred": out["bbox"] if "bbox" in out else torch.zeros((R, 6 * nc), device=dev),    # :234
                "_m3d": out}


def im_detect_mask(model, im_scale, boxes, blob_conv):
    """lib/core/test.py:439-476 with the reference's signature (model = Generalized_RCNN, possibly behind .module)."""
    net = model.module if hasattr(model, "module") else model
    if net.mask_head is None:
        raise AttributeError("this checkpoint has no Mask_Head / Mask_Outs weights (MODEL.MASK_ON False)")
    return _im_detect_mask(net.mask_head, im_scale, boxes, blob_conv)


def segm_results(model, cls_boxes, masks, ref_boxes, im_s, im_h, im_w):
    """lib/core/test.py:886-945; the reference reads cfg.MODEL.NUM_CLASSES / MRCNN.* globals, here they come from the model's cfg."""
    net = model.module if hasattr(model, "module") else model
    if net.mask_head is None:
        raise AttributeError("this checkpoint has no Mask_Head / Mask_Outs weights (MODEL.MASK_ON False)")
    mh = net.mask_head
    return _segm_results(cls_boxes, masks, ref_boxes, im_s, im_h, im_w, num_classes=mh.cfg.num_classes, resolution=mh.M,
                         cls_specific=mh.cls_specific, thresh=getattr(mh.cfg, "mask_thresh_binarize", 0.5), device=net.device)


class PeakResponseMapping_3d(Generalized_RCNN):
    """PRM-mode model with the reference's call and return shape (peak_response_mapping_3d.py:85-193)."""

    def __init__(self, state_dict, cfg, device="cuda", **kargs):
        super().__init__(state_dict, cfg, device)
        if kargs.get("enable_peak_stimulation", False):
            raise NotImplementedError("peak stimulation is disabled by default in the reference (:25) and not on the path")
        self.engine = PRMEngine(self.det)
        self.inferencing = True

    def inference(self):            # :202-206 (patches the convs there; the engine here always runs the PRM rule)
        self.inferencing = True
        return self

    def __call__(self, data, im_info, im_scale=1.0, roidb=None, peak_threshold=0.1, retrieval_cfg=None):
        return self.forward(data, im_info, im_scale, roidb, peak_threshold, retrieval_cfg)

    def forward(self, data, im_info, im_scale=1.0, roidb=None, peak_threshold=0.1, retrieval_cfg=None):
        if retrieval_cfg is not None:
            raise NotImplementedError("instance_seg is `pass` in the reference (:82-83)")
        scale = float(_first(im_scale))
        if scale != 1.0:
            raise NotImplementedError("im_scale is always 1.0 on the reference path (infer_simple.py:219)")
        x = self._data(data)
        assert x.shape[0] == 1, "Currently inference mode (with peak backpropagation) only supports one image at a time."   # :148
        out = self.engine.prm_tile(x, peak_threshold=float(peak_threshold), dense=True)
        if out is None:
            return None, None, None, None, None                                                           # :189-190
        return None, out["crm"], out["peaks"].to(torch.int64).to(out["crm"].device), out["prms"], out["dets"].to(out["crm"].device)                    # :180-185


# ------------------------------------------------------------------------------------------- lib/core/test.py drivers
def im_detect_bbox(model, inputs, im_scale, boxes=None):
    """lib/core/test.py:194-263 (FASTER_RCNN on, BBOX_REG on, class-specific regression): returns
    (scores [R,nc], pred_boxes [R,6nc], im_scale, blob_conv) as NumPy / tensor exactly like the reference."""
    input_shape = np.asarray(inputs["im_info"])[0][:3]
    ret = model(data=[torch.from_numpy(np.ascontiguousarray(inputs["data"]))] if not torch.is_tensor(inputs["data"]) else [inputs["data"]],
                im_info=[torch.from_numpy(np.asarray(inputs["im_info"], dtype=np.float64))])
    c = model.cfg
    rois = ret["rois"]
    R = rois.shape[0]
    if R == 0:
        return (np.zeros((0, c.num_classes), np.float32), np.zeros((0, 6 * c.num_classes), np.float32), im_scale, ret["blob_conv"])
    m = ret["_m3d"]
    pred = m["pred_boxes"] if (im_scale == 1.0 and "pred_boxes" in m) else \
        ops.bbox_transform3d((rois[:, 1:7] / im_scale).contiguous(), ret["bbox_pred"], c.bbox_reg_weights, clip_to=input_shape)
    scores = ret["cls_score"].cpu().numpy().reshape(-1, c.num_classes)                                   # :228-233
    return scores, pred.cpu().numpy(), im_scale, ret["blob_conv"]


def box_results_with_nms_and_limit(model, scores, boxes, scores_keep_idx=None):
    """lib/core/test.py:806-883 on NumPy inputs (what im_detect_bbox returns), NMS on the device.
    Returns (scores [n], boxes [n,6], cls_boxes list of [n_j,7] float32, cls_keep_idx list)."""
    dev = model.device
    s = torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float32)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32)).to(dev)
    k = None if scores_keep_idx is None else torch.from_numpy(np.ascontiguousarray(scores_keep_idx, dtype=np.int64)).to(dev)
    sc, bx, cls_boxes, cls_keep = model.det.box_results_with_nms_and_limit(s, b, k)
    return (sc.cpu().numpy(), bx.cpu().numpy(), [t.cpu().numpy() for t in cls_boxes], [t.cpu().numpy() for t in cls_keep])


def im_detect_all(model, im, box_proposals=None, timers=None):
    """lib/core/test.py:54-177 with TEST.NEED_CROP: norm1 (blob.py:179-184), slice padding, tiles from TEST.IN_SIZE /
    TEST.CROP_OVLP, per-tile im_detect_bbox + box_results_with_nms_and_limit, tile offsets, cross-tile nms_3d per class.
    Returns (cls_boxes_total, cls_segms_total, cls_keyps).  With MODEL.MASK_ON (cfg.mask_on and a checkpoint that carries the
    mask branch): im_detect_mask on every tile's kept boxes (:123-126), masks and boxes filtered by the cross-tile NMS (:161-163),
    segm_results over the whole volume (:166) and binary_mask_to_rle per detection (:170-172); otherwise the segms list stays
    empty, as in the reference.  keyps is None (KEYPOINTS_ON: `pass` in the reference)."""
    c = model.cfg
    net = model.module if hasattr(model, "module") else model
    mask_on = bool(getattr(c, "mask_on", False)) and getattr(net, "mask_head", None) is not None
    patch = tuple(c.in_size)
    im = np.asarray(im)
    vol = tiling.norm1(im, np.float32).astype(np.float32)
    vol, pad_s = tiling.pad_slices(vol, patch[0])                                                          # :79-86
    sidx, hidx, widx = tiling.detect_grid(c, vol.shape)                                                    # :87-90
    cls_total = [np.empty((0, 7), dtype=np.float32) for _ in range(c.num_classes)]
    boxes_total = np.empty((0, 6), dtype=np.float32)                                                       # :97
    masks_total = None
    for ss in sidx:
        for hs in hidx:
            for ws in widx:
                cube = {"data": np.ascontiguousarray(vol[ss:ss + patch[0], hs:hs + patch[1], ws:ws + patch[2]])[None, None],
                        "im_info": np.array(patch + (1.0,), dtype=np.float64)[None, :]}
                scores, boxes, im_scale, blob_conv = im_detect_bbox(model, cube, 1.0)                      # :106
                _, boxes, cls_boxes, _ = box_results_with_nms_and_limit(model, scores, boxes)              # :114
                if mask_on and boxes.shape[0] > 0:                                                         # :123-126
                    masks = im_detect_mask(model, [im_scale], boxes, blob_conv)
                    masks_total = masks if masks_total is None else np.append(masks_total, masks, axis=0)  # :147
                off = np.array([ws, hs, ss - pad_s, ws, hs, ss - pad_s, 0], dtype=np.float32)              # :117-121,140-141
                boxes_total = np.append(boxes_total, boxes.reshape(-1, 6) + off[:6], axis=0)               # :143
                for j in range(1, c.num_classes):
                    cls_total[j] = np.append(cls_total[j], cls_boxes[j] + off, axis=0)                     # :144-145
    for j in range(1, c.num_classes):                                                                      # :150-163
        d = np.ascontiguousarray(cls_total[j], dtype=np.float32)
        if d.shape[0]:
            keep = ops.nms3d(torch.from_numpy(d).to(model.device), c.nms).cpu().numpy()
            cls_total[j] = d[keep, :]
            if mask_on and masks_total is not None and masks_total.shape[0] > 0:
                boxes_total, masks_total = boxes_total[keep, :], masks_total[keep, :]
    cls_segms = [[] for _ in range(c.num_classes)]
    if mask_on and masks_total is not None:                                                                # :164-172
        from .io import binary_mask_to_rle
        cls_segms = segm_results(model, cls_total, masks_total, boxes_total, im.shape[0], im.shape[1], im.shape[2])
        for j in range(1, c.num_classes):
            cls_segms[j] = [binary_mask_to_rle(m) for m in cls_segms[j]]
    return cls_total, cls_segms, None
