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

    @property
    def anchors(self):
        return generate_anchors_3d(self.stride, self.sizes, self.aspect_ratios)

    @property
    def num_anchors(self):
        return len(self.sizes) * len(self.aspect_ratios)
