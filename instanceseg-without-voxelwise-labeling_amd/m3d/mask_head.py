"""Mask branch of the detector: `mask_net` = Mask_Head (RoIAlign3D -> N x [Conv3d 3^3 dilated + ReLU] -> ConvTranspose3d(2, 2) + ReLU)
followed by Mask_Outs (Conv3d 1^3 -> sigmoid), and `im_detect_mask` around it.

Reference: lib/modeling/mask_rcnn_heads.py:132-193 (head), :20-68 (outputs), lib/modeling/model_builder.py:327-331 (mask_net),
lib/core/test.py:439-476 (im_detect_mask).  Both shipped configs run with MODEL.MASK_ON False, so nothing on the headline path
reaches this; it exists so that a MASK_ON checkpoint has somewhere to go.  `segm_results` (core/test.py:886-945) pastes the soft
M^3 masks into the volume on the device (csrc/mask_paste.hip: skimage.transform.resize restated on scipy.ndimage's arithmetic).

Every convolution runs in the HIP library:
  * the dilated 3x3x3 convs through m3d_conv3d_forward_dilated (direct MFMA kernel, halo = dilation);
  * ConvTranspose3d(kernel 2, stride 2) never overlaps its outputs, so it is a 1x1x1 conv to 8*Cout channels
    (one per output parity (a,b,c)) followed by a voxel shuffle to the 2x grid; bias + ReLU ride in the conv epilogue;
  * the 1x1x1 classifier through m3d_conv3d_forward."""
import numpy as np
import torch

from . import ops


class MaskHeadM3D:
    def __init__(self, params, cfg, resolution=None, roi_res=None, sampling_ratio=None, dilation=None, cls_specific=None):
        """params: CUDA fp32 tensors under the reference's state-dict keys `Mask_Head.conv_fcn.{0,2,..}.{weight,bias}`,
        `Mask_Head.upconv.{weight,bias}`, `Mask_Outs.classify.{weight,bias}`.  The MRCNN.* settings come from cfg
        (Cfg.from_yaml reads them; defaults = lib/core/config.py:751-780) unless given here."""
        self.cfg = cfg
        resolution = getattr(cfg, "mask_resolution", 14) if resolution is None else resolution
        roi_res = getattr(cfg, "mask_roi_res", 7) if roi_res is None else roi_res
        sampling_ratio = getattr(cfg, "mask_sampling_ratio", 0) if sampling_ratio is None else sampling_ratio
        dilation = getattr(cfg, "mask_dilation", 2) if dilation is None else dilation
        cls_specific = getattr(cfg, "mask_cls_specific", True) if cls_specific is None else cls_specific
        self.M, self.roi_res, self.sampling_ratio, self.dilation = int(resolution), int(roi_res), int(sampling_ratio), int(dilation)
        self.cls_specific = bool(cls_specific)
        self.convs = []
        i = 0
        while "Mask_Head.conv_fcn.%d.weight" % (2 * i) in params:                 # mask_rcnn_heads.py:146-153
            w = params["Mask_Head.conv_fcn.%d.weight" % (2 * i)]
            assert tuple(w.shape[2:]) == (3, 3, 3)
            self.convs.append((ops.PackedConv3d(w), params["Mask_Head.conv_fcn.%d.bias" % (2 * i)].contiguous()))
            i += 1
        if not self.convs:
            raise KeyError("no Mask_Head.conv_fcn.* weights in the state dict")
        wu = params["Mask_Head.upconv.weight"]                                      # [Cin, Cout, 2, 2, 2] (:156)
        assert tuple(wu.shape[2:]) == (2, 2, 2)
        cin, cout = int(wu.shape[0]), int(wu.shape[1])
        self.up_cout = cout
        # out[co, 2z+a, 2y+b, 2x+c] = bias[co] + sum_ci in[ci,z,y,x] * W[ci,co,a,b,c]: channel (co,a,b,c) of a 1x1x1 conv
        self.up = ops.PackedConv3d(wu.permute(1, 2, 3, 4, 0).reshape(cout * 8, cin, 1, 1, 1).contiguous())
        self.up_bias = params["Mask_Head.upconv.bias"].repeat_interleave(8).contiguous()
        wc = params["Mask_Outs.classify.weight"]                                    # [n_classes or 1, C, 1, 1, 1] (:31)
        assert tuple(wc.shape[2:]) == (1, 1, 1) and wc.shape[0] == (cfg.num_classes if self.cls_specific else 1)
        self.classify = ops.PackedConv3d(wc)
        self.classify_bias = params["Mask_Outs.classify.bias"].contiguous()
        self._ones = {}

    def _one(self, n, device):
        t = self._ones.get(n)
        if t is None:
            t = self._ones[n] = torch.ones((n,), dtype=torch.float32, device=device)
        return t

    def head(self, feat, mask_rois):
        """Mask_Head.forward (mask_rcnn_heads.py:181-193): [R, C, 2M', 2M', 2M'] with M' = roi_res."""
        c = self.cfg
        x = ops.roi_align3d_forward(feat, mask_rois, self.roi_res, self.roi_res, self.roi_res, 1.0 / c.stride, self.sampling_ratio)
        for conv, bias in self.convs:
            x = conv(x, scale=self._one(conv.cout, x.device), shift=bias, relu=True, dilation=self.dilation)
        y = self.up(x, scale=self._one(self.up.cout, x.device), shift=self.up_bias, relu=True)        # :193 (ReLU is pointwise)
        R, r = y.shape[0], self.roi_res
        y = y.view(R, self.up_cout, 2, 2, 2, r, r, r).permute(0, 1, 5, 2, 6, 3, 7, 4)
        return y.reshape(R, self.up_cout, 2 * r, 2 * r, 2 * r)

    def outputs(self, x):
        """Mask_Outs.forward in eval mode (mask_rcnn_heads.py:62-68, UPSAMPLE_RATIO 1, USE_FC_OUTPUT False)."""
        y = self.classify(x, scale=self._one(self.classify.cout, x.device), shift=self.classify_bias)
        return torch.sigmoid(y)

    def mask_net(self, blob_conv, rpn_blob):
        """model_builder.py:327-331: rpn_blob = {'mask_rois': [R, 7] (batch, x1, y1, z1, x2, y2, z2)} (tensor or ndarray)."""
        rois = rpn_blob["mask_rois"] if isinstance(rpn_blob, dict) else rpn_blob
        if not torch.is_tensor(rois):
            rois = torch.from_numpy(np.ascontiguousarray(rois, dtype=np.float32))
        rois = rois.to(device=blob_conv.device, dtype=torch.float32)
        return self.outputs(self.head(blob_conv, rois))


def im_detect_mask(mask_head, im_scale, boxes, blob_conv):
    """lib/core/test.py:439-476.  boxes: ndarray [R, 6] in image coordinates; returns ndarray [R, K, M, M, M] float32
    (K = NUM_CLASSES when class specific, else 1); [0, M, M, M] for no boxes (:457-459)."""
    M = mask_head.M
    boxes = np.asarray(boxes)
    if boxes.shape[0] == 0:
        return np.zeros((0, M, M, M), np.float32)
    scale = float(im_scale[0] if np.ndim(im_scale) else im_scale)
    rois = np.hstack([np.zeros((boxes.shape[0], 1)), boxes.astype(np.float64) * scale]).astype(np.float32)   # _get_rois_blob (:967-980)
    pred = mask_head.mask_net(blob_conv, {"mask_rois": rois})
    pred = pred.cpu().numpy().squeeze()                                                                       # :469
    K = mask_head.cfg.num_classes if mask_head.cls_specific else 1
    return pred.reshape([-1, K, M, M, M])                                                                     # :471-474


def expand_boxes(boxes, scale):
    """lib/utils/boxes_3d.py:271-292."""
    boxes = np.asarray(boxes, dtype=np.float64)
    half = (boxes[:, 3:6] - boxes[:, 0:3]) * .5 * scale
    ctr = (boxes[:, 3:6] + boxes[:, 0:3]) * .5
    return np.hstack([ctr - half, ctr + half])


def segm_results(cls_boxes, masks, ref_boxes, im_s, im_h, im_w, num_classes=2, resolution=None, cls_specific=True, thresh=0.5,
                 device="cuda"):
    """lib/core/test.py:886-945 with the reference's signature (the cfg values are keyword arguments): cls_boxes = per-class lists
    of [n_j, 7] detections, masks [R, K, M, M, M] from im_detect_mask, ref_boxes [R, 6] (the same class-major order).
    Returns cls_segms: per class a list of [im_s, im_h, im_w] uint8 arrays (host, as the reference)."""
    masks_t = masks if torch.is_tensor(masks) else torch.from_numpy(np.ascontiguousarray(masks, dtype=np.float32))
    masks_t = masks_t.to(device)
    M = int(masks_t.shape[-1]) if resolution is None else int(resolution)
    rb = expand_boxes(ref_boxes, (M + 2.0) / M).astype(np.int32)                              # :896-898
    counts = [0] + [int(cls_boxes[j].shape[0]) for j in range(1, num_classes)]
    assert sum(counts) == masks_t.shape[0], "segm_results: masks and cls_boxes disagree"        # :944
    channel = np.concatenate([np.full((counts[j],), j if cls_specific else 0, np.int32) for j in range(num_classes)]) \
        if masks_t.shape[0] else np.zeros((0,), np.int32)
    pasted = ops.mask_paste3d(masks_t, channel, rb, (im_s, im_h, im_w), thresh).cpu().numpy()
    cls_segms, ind = [[] for _ in range(num_classes)], 0
    for j in range(1, num_classes):
        cls_segms[j] = [pasted[ind + i] for i in range(counts[j])]
        ind += counts[j]
    return cls_segms
