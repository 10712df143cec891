"""Drop-in modules under the reference's own dotted names, backed by libm3d.so.

    import m3d.compat; m3d.compat.install()          # before `import modeling...` / `import utils...`

registers:
  modeling.roi_xfrom.roi_align_3d.functions.roi_align_3d   RoIAlignFunction_3d          (functions/roi_align_3d.py:7-51)
  modeling.roi_xfrom.roi_align_3d.modules.roi_align_3d     RoIAlign_3d / Avg_3d / Max_3d (modules/roi_align_3d.py:6-48)
  utils.cython_nms_3d      nms_3d, nms_3d_volume, soft_nms_3d                            (lib/utils/cython_nms_3d.pyx)
  utils.cython_bbox_3d     bbox_overlaps_3d                                              (lib/utils/cython_bbox_3d.pyx)
  model.roi_pooling.functions.roi_pool / model.roi_crop.functions.roi_crop   import-compat shims (2D legacy,
                           model_builder.py:11-12 imports them; the 3D path never calls them)
  otsu                     otsu_py_2d_fast                                               (tools/otsu.py:199-284)
and install_conv3d() routes qualifying torch.nn.functional.conv3d calls (fp32, CUDA, NCDHW, stride 1,
padding k//2, dilation 1, groups 1, k in {1,3} or the 5^3/Cin=1 stem) to the MFMA kernels, forward and
backward-data, so lib/modeling/DSN.py and lib/prm/peak_backprop_3d.py run unchanged on them.
"""
import importlib
import sys
import types
import weakref

import numpy as np
import torch

from . import ops


# ----------------------------------------------------------------------------- RoIAlign3D
class _RoIAlign3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, rois, AS, AH, AW, scale, ratio):
        ctx.save_for_backward(rois)
        ctx.cfg = (AS, AH, AW, scale, ratio, tuple(features.shape))
        return ops.roi_align3d_forward(features, rois, AS, AH, AW, scale, ratio)

    @staticmethod
    def backward(ctx, grad_output):
        rois, = ctx.saved_tensors
        AS, AH, AW, scale, ratio, fshape = ctx.cfg
        g = ops.roi_align3d_backward(grad_output.contiguous(), rois, fshape, AS, AH, AW, scale, ratio)
        return g, None, None, None, None, None, None


class RoIAlignFunction_3d(object):
    """Same call shape as the reference's legacy instance-style Function:
    RoIAlignFunction_3d(AS, AH, AW, scale, ratio)(features, rois)   (model_builder.py:311-312)."""

    def __init__(self, aligned_slices, aligned_height, aligned_width, spatial_scale, sampling_ratio):
        self.aligned_slices = int(aligned_slices)
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)

    def __call__(self, features, rois):
        if not features.is_cuda:
            raise NotImplementedError          # functions/roi_align_3d.py:31-32
        return _RoIAlign3dFn.apply(features, rois, self.aligned_slices, self.aligned_height, self.aligned_width,
                                   self.spatial_scale, self.sampling_ratio)


class RoIAlign_3d(torch.nn.Module):
    def __init__(self, aligned_slices, aligned_height, aligned_width, spatial_scale, sampling_ratio):
        super().__init__()
        self.aligned_width, self.aligned_height, self.aligned_slices = int(aligned_width), int(aligned_height), int(aligned_slices)
        self.spatial_scale, self.sampling_ratio = float(spatial_scale), int(sampling_ratio)

    def _fn(self, extra):
        return RoIAlignFunction_3d(self.aligned_slices + extra, self.aligned_height + extra, self.aligned_width + extra,
                                   self.spatial_scale, self.sampling_ratio)

    def forward(self, features, rois):
        return self._fn(0)(features, rois)


class RoIAlignAvg_3d(RoIAlign_3d):
    def forward(self, features, rois):       # modules/roi_align_3d.py:30-33
        return torch.nn.functional.avg_pool3d(self._fn(1)(features, rois), kernel_size=2, stride=1)


class RoIAlignMax_3d(RoIAlign_3d):
    def forward(self, features, rois):       # modules/roi_align_3d.py:45-48
        return torch.nn.functional.max_pool3d(self._fn(1)(features, rois), kernel_size=2, stride=1)


# ----------------------------------------------------------------------------- Cython-API box ops (NumPy in/out)
def _check_f32_2d(a, name):
    # Cython buffer typing `np.ndarray[np.float32_t, ndim=2]` rejects anything else with ValueError
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)" % (name, type(a).__name__))
    if a.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % a.ndim)
    if a.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t' but got '%s'" % a.dtype)


def nms_3d(dets, thresh):
    _check_f32_2d(dets, "dets")
    d = torch.from_numpy(np.ascontiguousarray(dets)).cuda()
    return ops.nms3d(d, thresh).cpu().numpy()


def nms_3d_volume(dets, thresh):
    _check_f32_2d(dets, "dets")
    d = torch.from_numpy(np.ascontiguousarray(dets)).cuda()
    return ops.nms3d(d, thresh, by_volume=True).cpu().numpy()


def soft_nms_3d(boxes_in, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    raise NotImplementedError(
        "soft_nms_3d is unreachable in the reference (TEST.SOFT_NMS.ENABLED=False, core/config.py:370; the wrapper "
        "boxes_3d.soft_nms is mis-named vs its caller core/test.py:843 and the kernel overwrites column 4 three times, "
        "cython_nms_3d.pyx:280-282); deliberately not reproduced")


def bbox_overlaps_3d(boxes, query_boxes):
    _check_f32_2d(boxes, "boxes")
    _check_f32_2d(query_boxes, "query_boxes")
    b = torch.from_numpy(np.ascontiguousarray(boxes)).cuda()
    q = torch.from_numpy(np.ascontiguousarray(query_boxes)).cuda()
    return ops.bbox_overlaps3d(b, q).cpu().numpy()


def otsu_py_2d_fast(image, prm, b_range=None):
    """tools/otsu.py:199 signature; uint16 volumes (what both callers pass).  Returns (uint8 mask, k, b)."""
    if b_range is not None:
        raise NotImplementedError("b_range is never passed by the reference callers")
    img = np.ascontiguousarray(image)
    pr = np.ascontiguousarray(prm)
    if img.dtype != np.uint16 or pr.dtype != np.uint16:
        raise TypeError("otsu_py_2d_fast (HIP) takes the uint16 crops the reference callers produce")
    offs = torch.tensor([0, img.size], dtype=torch.int64, device="cuda")
    G = int(img.max()) - int(img.min()) + 1
    mask, kb, status = ops.otsu2d_batch(torch.from_numpy(img.ravel()).cuda(), torch.from_numpy(pr.ravel()).cuda(), offs,
                                        max_gray_range=max(G, 2))
    if int(status[0]) != 0:
        raise UnboundLocalError("local variable 'k_max' referenced before assignment")   # otsu.py:277 behaviour
    return mask.cpu().numpy().reshape(img.shape), int(kb[0, 0]), int(kb[0, 1])


class _Legacy2D(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("2D legacy op: no 3D version exists in the reference (SURVEY 8a-15)")


# ----------------------------------------------------------------------------- F.conv3d interception
_orig_conv3d = None
_pack_cache = {}          # (id(parameter), mode) -> (weakref to the parameter, _version, data_ptr, pack)
STEM_DGRAD = "stem-dgrad"  # pseudo-mode: tap-flipped [C,125] weights of the 5^3 / Cin = 1 stem's backward-data


def _make_pack(weight, mode):
    w = weight.detach()
    return ops.conv3d_stem5_dgrad_weights(w) if mode == STEM_DGRAD else ops.PackedConv3d(w, mode)


def _packed(weight, mode):
    """MFMA-order pack of `weight`.  Only nn.Parameters are cached, and a cached pack is served only to the SAME live
    object at the same `_version` and address: the reference's pr_conv3d passes a fresh `F.relu(self.weight).detach()`
    every call (peak_backprop_3d.py:41) whose address the allocator may later hand to another layer's temporary of the
    same shape - temporaries are therefore re-packed on every call (7-25 us, small next to the conv)."""
    if not isinstance(weight, torch.nn.Parameter):
        return _make_pack(weight, mode)
    key = (id(weight), mode)
    ent = _pack_cache.get(key)
    if ent is not None:
        ref, ver, ptr, pack = ent
        if ref() is weight and ver == weight._version and ptr == weight.data_ptr():
            return pack
    pack = _make_pack(weight, mode)
    _pack_cache[key] = (weakref.ref(weight, lambda _r, k=key: _pack_cache.pop(k, None)), weight._version, weight.data_ptr(), pack)
    return pack


class _Conv3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(weight, x if weight.requires_grad else None)
        ctx.has_bias = bias is not None
        return _packed(weight, ops.W_PLAIN)(x, shift=bias.detach().contiguous() if bias is not None else None)

    @staticmethod
    def backward(ctx, gy):
        weight, x = ctx.saved_tensors
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            if weight.shape[2] == 5:          # conv1a: one input channel -> VALU dgrad kernel (csrc/prm.hip)
                gx = ops.conv3d_stem5_dgrad(gy, _packed(weight, STEM_DGRAD))
            else:
                gx = _packed(weight, ops.W_DGRAD)(gy)
        gw = ops.conv3d_wgrad(x, gy, weight.shape[2]) if ctx.needs_input_grad[1] else None
        gb = ops.conv3d_bias_grad(gy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


def _qualifies(x, weight, stride, padding, dilation, groups):
    def tup(v):
        return (v,) * 3 if isinstance(v, int) else tuple(v)
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 5):
        return False
    k = weight.shape[2]
    if weight.shape[3] != k or weight.shape[4] != k or groups != 1:
        return False
    if tup(stride) != (1, 1, 1) or tup(dilation) != (1, 1, 1) or isinstance(padding, str) or tup(padding) != (k // 2,) * 3:
        return False
    if k in (1, 3):
        return True
    return k == 5 and weight.shape[1] == 1 and weight.shape[0] <= 64


def conv3d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if _qualifies(input, weight, stride, padding, dilation, groups):
        return _Conv3dFn.apply(input.contiguous(), weight, bias)
    return _orig_conv3d(input, weight, bias, stride, padding, dilation, groups)


# ----------------------------------------------------------------------------- F.linear interception (box head: fast_rcnn_heads.py:84-85,114-117,42-45)
_orig_linear = None
_lin_cache = {}           # id(parameter) -> (weakref, _version, data_ptr, bias id / version, SplitLinear)


def _split_linear(weight, bias):
    """bf16x3 pack of an nn.Linear weight, cached like the conv packs: only for live nn.Parameters at the same version / address."""
    if not isinstance(weight, torch.nn.Parameter):
        return ops.SplitLinear(weight.detach(), None if bias is None else bias.detach())
    bkey = None if bias is None else (id(bias), bias._version, bias.data_ptr())
    ent = _lin_cache.get(id(weight))
    if ent is not None:
        ref, ver, ptr, bk, lin = ent
        if ref() is weight and ver == weight._version and ptr == weight.data_ptr() and bk == bkey:
            return lin
    lin = ops.SplitLinear(weight.detach(), None if bias is None else bias.detach())
    _lin_cache[id(weight)] = (weakref.ref(weight, lambda _r, k=id(weight): _lin_cache.pop(k, None)), weight._version, weight.data_ptr(), bkey, lin)
    return lin


class _LinearFn(torch.autograd.Function):
    """x [M,K] . W[N,K]^T + b on libm3d's GEMMs: the bf16x3 split kernels (fp32 accuracy) where K % 32 == 0 and N >= 64 (fc1, fc2),
    else the fp32-input MFMA kernel (cls_score, bbox_pred).  Backward (training; off the inference path): the same kernels on
    transposed operands."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if ops.SplitLinear.supported(weight):
            return _split_linear(weight, bias)(x)
        return ops.linear(x, weight.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.linear(gy, weight.detach().t().contiguous()) if gy.shape[1] % 4 == 0 else gy @ weight.detach()
        if ctx.needs_input_grad[1]:
            gw = ops.linear(gy.t().contiguous(), x.detach().t().contiguous()) if gy.shape[0] % 4 == 0 else gy.t() @ x.detach()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb


def _aligned16(t):
    return t.data_ptr() % 16 == 0 and t.is_contiguous()


def linear(input, weight, bias=None):
    # libm3d's GEMMs read 16-byte quads: K % 4 == 0 and 16-byte aligned, contiguous operands (a view into the middle of a
    # buffer goes to torch's own linear instead of raising M3D_EUNSUPPORTED)
    if (input.is_cuda and input.dtype == torch.float32 and weight.dtype == torch.float32 and input.dim() == 2 and weight.dim() == 2
            and input.shape[1] == weight.shape[1] and input.shape[1] % 4 == 0 and input.shape[0] > 0 and _aligned16(weight)):
        x = input.contiguous()
        if x.data_ptr() % 16 == 0:
            return _LinearFn.apply(x, weight, bias)
    return _orig_linear(input, weight, bias)


def invalidate_packs():
    """Drop every cached weight pack (conv and linear).  The caches are keyed on the parameter's `_version`, which in-place
    updates through `.data` (`p.data.mul_()`, `p.data.copy_()`, old-style optimisers) do NOT bump: call this after such an update,
    or update through the parameter itself (`with torch.no_grad(): p.mul_()`, `load_state_dict`), which does."""
    _pack_cache.clear()
    _lin_cache.clear()


def install_linear():
    global _orig_linear
    if _orig_linear is None:
        _orig_linear = torch.nn.functional.linear
        torch.nn.functional.linear = linear


def uninstall_linear():
    global _orig_linear
    if _orig_linear is not None:
        torch.nn.functional.linear = _orig_linear
        _orig_linear = None


def install_conv3d():
    global _orig_conv3d
    if _orig_conv3d is None:
        _orig_conv3d = torch.nn.functional.conv3d
        torch.nn.functional.conv3d = conv3d


def uninstall_conv3d():
    global _orig_conv3d
    if _orig_conv3d is not None:
        torch.nn.functional.conv3d = _orig_conv3d
        _orig_conv3d = None


# ----------------------------------------------------------------------------- registration
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__m3d__ = True
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _pkg(name):
    """The real package when the reference's lib/ is importable, else an empty stand-in package."""
    if name not in sys.modules:
        try:
            importlib.import_module(name)
        except Exception:
            m = _mod(name)
            m.__path__ = []
    return sys.modules[name]


def install(conv3d=True, linear=True):
    """Register the drop-in modules.  Packages that already exist (the reference's `modeling`, `utils`) are kept;
    only the native-op leaf modules are replaced."""
    for pk in ("modeling", "modeling.roi_xfrom", "modeling.roi_xfrom.roi_align_3d",
               "modeling.roi_xfrom.roi_align_3d.functions", "modeling.roi_xfrom.roi_align_3d.modules",
               "utils", "model", "model.roi_pooling", "model.roi_pooling.functions", "model.roi_crop",
               "model.roi_crop.functions"):
        _pkg(pk)
    _mod("modeling.roi_xfrom.roi_align_3d.functions.roi_align_3d", RoIAlignFunction_3d=RoIAlignFunction_3d)
    _mod("modeling.roi_xfrom.roi_align_3d.modules.roi_align_3d", RoIAlign_3d=RoIAlign_3d, RoIAlignAvg_3d=RoIAlignAvg_3d,
         RoIAlignMax_3d=RoIAlignMax_3d, RoIAlignFunction_3d=RoIAlignFunction_3d)
    _mod("utils.cython_nms_3d", nms_3d=nms_3d, nms_3d_volume=nms_3d_volume, soft_nms_3d=soft_nms_3d)
    _mod("utils.cython_bbox_3d", bbox_overlaps_3d=bbox_overlaps_3d)
    _mod("model.roi_pooling.functions.roi_pool", RoIPoolFunction=_Legacy2D)
    _mod("model.roi_crop.functions.roi_crop", RoICropFunction=_Legacy2D)
    try:                                   # tools/otsu.py also holds otsu_py / otsu_py_2d (binarization_nuclei.py:12)
        importlib.import_module("otsu").otsu_py_2d_fast = otsu_py_2d_fast
    except Exception:
        _mod("otsu", otsu_py_2d_fast=otsu_py_2d_fast)
    if conv3d:
        install_conv3d()
    if linear:
        install_linear()
