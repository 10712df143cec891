"""Configuration keys the hot path reads, and the anchor generator (product-side host logic).

Reference: lib/core/config.py (keys listed in SURVEY 5), the two shipped YAMLs
(configs/cell_tracking_baseline/e2e_mask_rcnn_N3DH_SIM_dsn_body.yaml, configs/soma_starting/
e2e_mask_rcnn_soma_dsn_body.yaml) and lib/modeling/generate_anchors.py:67-201."""
import numpy as np

BBOX_XFORM_CLIP = float(np.log(1000. / 16.))       # lib/core/config.py:947


def generate_anchors_3d(stride=8, sizes=(12, 20, 30), aspect_ratios=((1., 1.), (1., 0.5))):
    """[A,6] float64 anchors (x1,y1,z1,x2,y2,z2), rows ratio-major / size-minor
    (generate_anchors.py:67-78 -> _generate_anchors_3d :92-101)."""
    stride = float(stride)
    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64).reshape(-1, 2)

    def sctrs(a):                                   # _sctrs_3d :113-121
        w, h, s = a[3] - a[0] + 1, a[4] - a[1] + 1, a[5] - a[2] + 1
        return w, h, s, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1), a[2] + 0.5 * (s - 1)

    def mk(w, h, s, cx, cy, cz):                    # _mkanchors_3d :141-159
        w, h, s = w[:, None], h[:, None], s[:, None]
        return np.hstack((cx - 0.5 * (w - 1), cy - 0.5 * (h - 1), cz - 0.5 * (s - 1),
                          cx + 0.5 * (w - 1), cy + 0.5 * (h - 1), cz + 0.5 * (s - 1)))

    base = np.array([1, 1, 1, stride, stride, stride], dtype=np.float64) - 1        # :96
    w, h, s, cx, cy, cz = sctrs(base)
    size_ratios = (w * h * s) / (ratios[:, 0] * ratios[:, 1])                        # _ratio_enum_3d :173-182
    ws = np.round(size_ratios ** (1. / 3))
    hs = np.round(ws * ratios[:, 0])
    ss = np.round(ws * ratios[:, 1])
    ratio_anchors = mk(ws, hs, ss, cx, cy, cz)
    out = []
    for i in range(ratio_anchors.shape[0]):                                          # _scale_enum_3d :194-201
        w, h, s, cx, cy, cz = sctrs(ratio_anchors[i])
        out.append(mk(w * scales, h * scales, s * scales, cx, cy, cz))
    return np.vstack(out)


class Cfg:
    """Hot-path configuration; defaults = the nuclei YAML merged over lib/core/config.py."""

    def __init__(self, **kw):
        self.stride = 8                                     # RPN.STRIDE
        self.sizes = (10, 27, 33, 38, 42, 46, 50)           # RPN.SIZES
        self.aspect_ratios = [[1.0, 0.5], [0.5, 0.5], [2., 0.5], [0.2, 0.5], [3., 2.]]   # RPN.ASPECT_RATIOS
        self.pre_nms_topN = 1000                            # TEST.RPN_PRE_NMS_TOP_N
        self.post_nms_topN = 1000                           # TEST.RPN_POST_NMS_TOP_N
        self.rpn_nms_thresh = 0.15                          # TEST.RPN_NMS_THRESH
        self.rpn_min_size = 0                               # TEST.RPN_MIN_SIZE (config.py:224)
        self.nms = 0.15                                     # TEST.NMS
        self.score_thresh = 0.05                            # TEST.SCORE_THRESH (config.py:233)
        self.detections_per_im = 300                        # TEST.DETECTIONS_PER_IM
        self.bbox_reg_weights = (10., 10., 10., 5., 5., 5.)  # MODEL.BBOX_REG_WEIGHTS
        self.num_classes = 2                                # MODEL.NUM_CLASSES
        self.roi_res = 7                                    # FAST_RCNN.ROI_XFORM_RESOLUTION
        self.sampling_ratio = 2                             # FAST_RCNN.ROI_XFORM_SAMPLING_RATIO
        self.mlp_dim = 1024                                 # FAST_RCNN.MLP_HEAD_DIM
        self.in_size = (64, 200, 200)                       # TEST.IN_SIZE
        self.crop_ovlp = 100                                # TEST.CROP_OVLP
        self.dataset = "nuclei"
        # mask branch (MODEL.MASK_ON False in both shipped YAMLs); defaults = lib/core/config.py:751-786
        self.mask_on = False                                # MODEL.MASK_ON
        self.mask_resolution = 14                           # MRCNN.RESOLUTION
        self.mask_roi_res = 7                               # MRCNN.ROI_XFORM_RESOLUTION
        self.mask_sampling_ratio = 0                        # MRCNN.ROI_XFORM_SAMPLING_RATIO
        self.mask_dilation = 2                              # MRCNN.DILATION
        self.mask_cls_specific = True                       # MRCNN.CLS_SPECIFIC_MASK
        self.mask_thresh_binarize = 0.5                     # MRCNN.THRESH_BINARIZE
        self.prm_on = True                                  # PRM_ON
        self.pp_method = "norm1"                            # PP_METHOD
        self.__dict__.update(kw)

    @staticmethod
    def nuclei(**kw):
        return Cfg(**kw)

    @staticmethod
    def soma(**kw):
        d = dict(stride=4, sizes=(10, 12, 14, 16, 18, 20, 22, 24, 28, 30, 34, 36, 38, 40), aspect_ratios=[[1.0, 1.0]],
                 rpn_nms_thresh=0.23, nms=0.23, score_thresh=0.0, in_size=(64, 160, 160), crop_ovlp=32, dataset="soma")
        d.update(kw)
        return Cfg(**d)

    # reference YAML key -> attribute (the keys the hot path reads; every other key of the file is accepted and ignored, as
    # lib/core/config.py's merge would accept it)
    YAML_KEYS = {
        ("RPN", "STRIDE"): "stride", ("RPN", "SIZES"): "sizes", ("RPN", "ASPECT_RATIOS"): "aspect_ratios",
        ("TEST", "RPN_PRE_NMS_TOP_N"): "pre_nms_topN", ("TEST", "RPN_POST_NMS_TOP_N"): "post_nms_topN",
        ("TEST", "RPN_NMS_THRESH"): "rpn_nms_thresh", ("TEST", "RPN_MIN_SIZE"): "rpn_min_size", ("TEST", "NMS"): "nms",
        ("TEST", "SCORE_THRESH"): "score_thresh", ("TEST", "DETECTIONS_PER_IM"): "detections_per_im",
        ("TEST", "IN_SIZE"): "in_size", ("TEST", "CROP_OVLP"): "crop_ovlp",
        ("MODEL", "BBOX_REG_WEIGHTS"): "bbox_reg_weights", ("MODEL", "NUM_CLASSES"): "num_classes", ("MODEL", "MASK_ON"): "mask_on",
        ("FAST_RCNN", "ROI_XFORM_RESOLUTION"): "roi_res", ("FAST_RCNN", "ROI_XFORM_SAMPLING_RATIO"): "sampling_ratio",
        ("FAST_RCNN", "MLP_HEAD_DIM"): "mlp_dim",
        ("MRCNN", "RESOLUTION"): "mask_resolution", ("MRCNN", "ROI_XFORM_RESOLUTION"): "mask_roi_res",
        ("MRCNN", "ROI_XFORM_SAMPLING_RATIO"): "mask_sampling_ratio", ("MRCNN", "DILATION"): "mask_dilation",
        ("MRCNN", "CLS_SPECIFIC_MASK"): "mask_cls_specific", ("MRCNN", "THRESH_BINARIZE"): "mask_thresh_binarize",
        ("PRM_ON",): "prm_on", ("PP_METHOD",): "pp_method",
    }

    @staticmethod
    def from_yaml(path_or_text, dataset=None, **kw):
        """The reference's `cfg_from_file(args.cfg_file)` (tools/infer_simple.py:104, lib/core/config.py:1063-1070) for the keys of
        this path: YAML values are merged over the defaults of lib/core/config.py; tuples written as strings ('(64, 200, 200)')
        are decoded with literal_eval as `_merge_a_into_b` / `_decode_cfg_value` do (:1120-1160).  `dataset` ('nuclei' / 'soma':
        the reference's --dataset argument, which selects the tile grid and the binarisation rule) defaults to 'soma' for
        stride-4 files.  Checks the model family: only the generalized_rcnn / DSN.dsn_body / roi_2mlp_head / RoIAlign path exists."""
        import ast
        import os
        import yaml
        text = open(path_or_text).read() if (len(str(path_or_text)) < 4096 and os.path.exists(str(path_or_text))) else str(path_or_text)
        y = yaml.safe_load(text) or {}

        def decode(v):
            if isinstance(v, str):
                try:
                    return ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    return v
            return v
        want = {("MODEL", "CONV_BODY"): "DSN.dsn_body", ("MODEL", "TYPE"): "generalized_rcnn",
                ("FAST_RCNN", "ROI_BOX_HEAD"): "fast_rcnn_heads.roi_2mlp_head", ("FAST_RCNN", "ROI_XFORM_METHOD"): "RoIAlign"}
        for (sec, key), val in want.items():
            got = (y.get(sec) or {}).get(key, val)
            if got != val:
                raise NotImplementedError("%s.%s = %r: only %r is on the accelerated path" % (sec, key, got, val))
        if (y.get("FPN") or {}).get("FPN_ON", False):
            raise NotImplementedError("FPN.FPN_ON: True is not on the accelerated path (both shipped configs have it off)")
        # lib/core/config.py's own defaults for these keys (:31-33,199-250,414-443,634-683), NOT the nuclei values of Cfg():
        # a key a YAML leaves out takes the reference default (the soma file has no CROP_OVLP: 32)
        d = dict(stride=16, sizes=(64, 128, 256, 512), aspect_ratios=(0.5, 1, 2), pre_nms_topN=12000, post_nms_topN=2000,
                 rpn_nms_thresh=0.7, rpn_min_size=0, nms=0.3, score_thresh=0.05, detections_per_im=100, in_size=(64, 240, 240),
                 crop_ovlp=32, bbox_reg_weights=(10., 10., 5., 5.), num_classes=-1, mask_on=False, roi_res=14, sampling_ratio=0,
                 mlp_dim=1024, prm_on=False, pp_method="norm1")
        for keys, attr in Cfg.YAML_KEYS.items():
            node, ok = y, True
            for k in keys:
                if isinstance(node, dict) and k in node:
                    node = node[k]
                else:
                    ok = False
                    break
            if ok:
                v = decode(node)
                d[attr] = tuple(v) if attr in ("sizes", "in_size", "bbox_reg_weights") else v
        if len(d["aspect_ratios"]) and np.ndim(d["aspect_ratios"][0]) == 0:
            raise NotImplementedError("RPN.ASPECT_RATIOS must be [h/w, s/w] pairs (3-D anchors, generate_anchors.py:67-78)")
        d["aspect_ratios"] = [list(map(float, r)) for r in d["aspect_ratios"]]
        if len(d["bbox_reg_weights"]) != 6 or d["num_classes"] < 2:
            raise NotImplementedError("MODEL.BBOX_REG_WEIGHTS needs 6 entries and MODEL.NUM_CLASSES >= 2 on the 3-D path")
        d["dataset"] = dataset or ("soma" if int(d.get("stride", 8)) == 4 else "nuclei")
        d.update(kw)
        return Cfg(**d)

    @property
    def anchors(self):
        return generate_anchors_3d(self.stride, self.sizes, self.aspect_ratios)

    @property
    def num_anchors(self):
        return len(self.sizes) * len(self.aspect_ratios)
