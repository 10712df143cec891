"""TEST INFRASTRUCTURE — import harness for the *reference's own Python* (CPU, this container only).

`/root/reference` is PyTorch-0.4 / NumPy-1.x era code that refuses to import unmodified on the
torch-2.10 / NumPy-2.2 image.  This module installs the minimum set of aliases and stub modules
(SURVEY.md §8c recipe) so that `lib/` can be imported *from where it lies* and executed on the CPU to
generate golden vectors (tests/golden/gen_golden.py) and to cross-check `oracle/`.

Nothing here is shipped: only tests/ and the golden generator import it, it reads /root/reference at
run time and is therefore unusable (and unused) on the GPU box.  No reference source is copied.
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("M3D_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
_REFSO = os.path.join(_HERE, "_ref")


def available():
    return os.path.isdir(os.path.join(REF, "lib", "modeling"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install(roi_align_callable=None):
    """Make `import modeling.*, core.*, utils.*, prm.*` resolve to /root/reference/lib.

    roi_align_callable(features, rois, AS, AH, AW, scale, ratio) -> Tensor is plugged into the stub
    that replaces the CUDA-only RoIAlignFunction_3d (the reference has no CPU implementation,
    functions/roi_align_3d.py:31-32).
    """
    global _installed
    import torch

    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    if _installed:
        if roi_align_callable is not None:
            sys.modules["modeling.roi_xfrom.roi_align_3d.functions.roi_align_3d"]._impl[0] = roi_align_callable
        return

    # (1) NumPy-1 aliases
    for n, t in (("int", int), ("float", float), ("bool", bool)):
        if not hasattr(np, n):
            setattr(np, n, t)

    # (2) heavy / absent third-party modules imported at module top but unused on this path
    cv2 = _stub("cv2")
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *_: None)
    sk = _stub("skimage")
    sk.io = _stub("skimage.io")
    sk.transform = _stub("skimage.transform", resize=None)
    sk.exposure = _stub("skimage.exposure")
    _stub("libtiff", TIFF=None)
    pc = _stub("pycocotools")
    pc.mask = _stub("pycocotools.mask")
    pc.coco = _stub("pycocotools.coco", COCO=None)
    pc.cocoeval = _stub("pycocotools.cocoeval", COCOeval=None)
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
            matplotlib.use("Agg")
        except Exception:
            mpl = _stub("matplotlib")
            mpl.pyplot = _stub("matplotlib.pyplot")

    # (3) scipy.misc.imresize is gone
    import scipy.misc
    if not hasattr(scipy.misc, "imresize"):
        scipy.misc.imresize = None

    # (4) torch-0.4 private names
    _stub("torch._six", string_classes=str, int_classes=int)
    import torch.utils.data.dataloader as _dl
    if not hasattr(_dl, "numpy_type_map"):
        _dl.numpy_type_map = {}

    sys.path.insert(0, os.path.join(REF, "lib"))
    sys.path.insert(0, os.path.join(REF, "tools"))

    # (5) native-op python wrappers whose _ext needs torch.utils.ffi
    class _Dead:
        def __init__(self, *a, **k):
            raise NotImplementedError("2D legacy op (SURVEY §8a-15)")

    for pk in ("model", "model.roi_pooling", "model.roi_pooling.functions",
               "model.roi_crop", "model.roi_crop.functions"):
        _stub(pk).__path__ = []
    _stub("model.roi_pooling.functions.roi_pool", RoIPoolFunction=_Dead)
    _stub("model.roi_crop.functions.roi_crop", RoICropFunction=_Dead)

    impl = [roi_align_callable]

    class RoIAlignFunction_3d:
        def __init__(self, AS, AH, AW, scale, ratio):
            self.args = (int(AS), int(AH), int(AW), float(scale), int(ratio))

        def __call__(self, features, rois):
            if impl[0] is None:
                raise NotImplementedError("no RoIAlign3D implementation plugged into the harness")
            return impl[0](features, rois, *self.args)

    import importlib
    importlib.import_module("modeling")
    for pk in ("modeling.roi_xfrom", "modeling.roi_xfrom.roi_align_3d",
               "modeling.roi_xfrom.roi_align_3d.functions"):
        _stub(pk).__path__ = []
    m = _stub("modeling.roi_xfrom.roi_align_3d.functions.roi_align_3d",
              RoIAlignFunction_3d=RoIAlignFunction_3d)
    m._impl = impl

    # (6) 2D / RLE cython modules (unused) + the freshly built 3D ones
    import utils  # reference lib/utils package
    _stub("utils.cython_bbox", bbox_overlaps=None)
    _stub("utils.cython_nms", nms=None, soft_nms=None)
    _stub("utils.cython_mask_3d", binary_mask_to_rle=None, rle_to_binary_mask=None)
    if not os.path.isdir(_REFSO):
        raise RuntimeError("run oracle/build_ref.sh first")
    utils.__path__.append(_REFSO)

    # (8) `.cuda()` is the identity on the CPU harness
    torch.Tensor.cuda = lambda self, *a, **k: self
    _installed = True


def load_cfg(yaml_rel, overrides=()):
    """(7) yaml.safe_load -> AttrDict -> merge (the reference's own loader calls yaml.load(f) w/o Loader)."""
    import yaml
    import core.config as C
    from utils.my_collections import AttrDict

    def to_attr(d):
        a = AttrDict()
        for k, v in d.items():
            a[k] = to_attr(v) if isinstance(v, dict) else v
        return a

    with open(os.path.join(REF, yaml_rel)) as f:
        y = to_attr(yaml.safe_load(f))
    C.merge_cfg_from_cfg(y)
    if overrides:
        C.merge_cfg_from_list(list(overrides))
    C.cfg.MODEL.LOAD_IMAGENET_PRETRAINED_WEIGHTS = False
    C.cfg.RPN.RPN_ON = True  # what assert_and_infer_cfg sets (config.py:1026)
    return C.cfg
