"""Tensor-level wrappers over the C ABI (include/m3d.h).  Every function takes/returns torch CUDA tensors and
launches on the current torch stream; none of them synchronises unless the result size is data dependent."""
import ctypes as C
import math
import threading

import numpy as np
import torch

from ._lib import lib, check, M3DError

BBOX_XFORM_CLIP = float(np.log(1000. / 16.))   # lib/core/config.py:947

__all__ = ["compact_rows", "compact_rows2", "box_head_outputs", "roi_align3d_forward", "roi_align3d_backward", "nms3d", "bbox_overlaps3d", "bbox_transform3d",
           "generate_proposals3d", "generate_proposals3d_batched", "box_results3d_batched", "nms3d_batched", "fused_max_boxes", "PackedConv3d", "maxpool3d_2x", "maxpool3d_2x_backward", "reduce_min", "reduce_min_multi", "norm1", "norm1_batched", "linear", "SplitLinear", "linear_roi_fused", "mask_paste3d",
           "otsu2d_batch", "prm_quantize_u8", "prm_quantize_windows_u8", "prm_quantize_windows_compact_u8", "roi_normalize", "conv3d_wgrad", "conv3d_bias_grad", "WinoConv3d", "ZwConv3d", "X3Conv3d", "StemWinoConv3d", "gaussian_filter_u16", "median_filter3_u16", "cc_largest_batch", "binary_closing6_batch", "paint_instances", "paint_instances_into", "paint_finish", "paint_begin", "crop_offsets", "upload_packed", "conv3d_windowed", "prm_seed", "strip_geometry", "prm_select_peaks", "PinnedPool", "upload", "prm_prepare", "prm_stem_dgrad", "prm_stem_prepare_weights", "SmallWindowDgrad", "prm_den_pool", "prm_stem_mfma_weights", "prm_stem_dgrad_fused", "prm_stem_dgrad_fused_supported", "prm_scatter", "conv3d_stem5_dgrad", "conv3d_stem5_dgrad_weights", "M3DError", "BBOX_XFORM_CLIP", "W_PLAIN", "W_RELU", "W_DGRAD", "W_DGRAD_RELU"]

W_PLAIN, W_RELU, W_DGRAD, W_DGRAD_RELU = 0, 1, 2, 3


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise M3DError("m3d ops need CUDA (ROCm) tensors; there is no CPU path")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32c(t):
    return t.contiguous().float() if (t.dtype != torch.float32 or not t.is_contiguous()) else t


# ------------------------------------------------------------------ RoIAlign3D
def roi_align3d_forward(features, rois, AS, AH, AW, spatial_scale, sampling_ratio, exact=False, ordered=True, feat_absmax=None):
    """exact=True: the reference kernel's fp32 operation order (bit-identical to the oracle); default: the
    separable fast form (same samples and weights, different summation order).  ordered=False: RoIs launched in index order (A/B).
    feat_absmax (round 6; a 1-element device tensor >= max |features|, ops.absmax): RoIs with sub-volumes of <= 128 voxels run as one GEMM
    per RoI on the f16 matrix cores (m3d_roi_align3d_forward_ws2); None keeps every RoI on the separable kernels."""
    _need_gpu(features, rois)
    features, rois = _f32c(features), _f32c(rois)
    B, Cc, S, H, W = features.shape
    R = rois.shape[0]
    # functions/roi_align_3d.py:24 zero-fills before the call; every forward kernel here writes every output element
    # (RoIs without a valid sample included), so the 351 MB memset is not repeated
    out = torch.empty((R, Cc, AS, AH, AW), dtype=torch.float32, device=features.device)
    cols = int(rois.shape[1]) if rois.dim() == 2 else 0
    if exact:
        check(lib().m3d_roi_align3d_forward_exact(int(AS), int(AH), int(AW), C.c_float(spatial_scale), int(sampling_ratio), _ptr(features), B, Cc, S, H, W,
                                                  _ptr(rois), R, cols, _ptr(out), _stream()), "roi_align3d_forward_exact")
        return out
    wsb = int(lib().m3d_roi_align3d_workspace_bytes(R)) if ordered else 0          # launch order (heaviest RoI first) + per-RoI set-up records
    ws = torch.empty((max(wsb, 4),), dtype=torch.uint8, device=features.device) if ordered else None
    if feat_absmax is not None and ordered:
        _need_gpu(feat_absmax)
        check(lib().m3d_roi_align3d_forward_ws2(int(AS), int(AH), int(AW), C.c_float(spatial_scale), int(sampling_ratio), _ptr(features), B, Cc, S, H, W,
                                                _ptr(rois), R, cols, _ptr(out), _ptr(ws), C.c_size_t(wsb), _ptr(feat_absmax), _stream()),
              "roi_align3d_forward")
        return out
    check(lib().m3d_roi_align3d_forward_ws(int(AS), int(AH), int(AW), C.c_float(spatial_scale), int(sampling_ratio), _ptr(features), B, Cc, S, H, W,
                                           _ptr(rois), R, cols, _ptr(out), _ptr(ws), C.c_size_t(wsb), _stream()), "roi_align3d_forward")
    return out


def roi_align3d_backward(top_grad, rois, feature_size, AS, AH, AW, spatial_scale, sampling_ratio):
    _need_gpu(top_grad, rois)
    top_grad, rois = _f32c(top_grad), _f32c(rois)
    B, Cc, S, H, W = feature_size
    grad = torch.zeros((B, Cc, S, H, W), dtype=torch.float32, device=top_grad.device)      # roi_align_3d.py:41-42
    check(lib().m3d_roi_align3d_backward(int(AS), int(AH), int(AW), C.c_float(spatial_scale), int(sampling_ratio),
                                         _ptr(top_grad), _ptr(rois), rois.shape[0], int(rois.shape[1]), _ptr(grad),
                                         B, Cc, S, H, W, _stream()), "roi_align3d_backward")
    return grad


# ------------------------------------------------------------------ NMS / IoU / decode
def nms3d(dets, thresh, by_volume=False):
    """dets [N,7] fp32 CUDA -> int64 CUDA tensor of kept input indices (ascending)."""
    _need_gpu(dets)
    dets = _f32c(dets)
    n = dets.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64, device=dets.device)
    if dets.dim() != 2 or dets.shape[1] != 7:
        raise ValueError("dets must be [N,7]")
    if n <= fused_max_boxes():                       # the fused kernels (three launches instead of seven)
        # the count lands in pinned host memory, written by the kernel itself: a stream synchronize instead of a D2H copy
        # (N = 1000: 90 -> 78 us wall; the kernels themselves take 63 us back to back)
        dev = dets.device
        keep = torch.empty((1, n), dtype=torch.int64, device=dev)
        num = _pinned_i32(dev)
        wsb = lib().m3d_nms3d_batched_workspace_bytes(1)
        ws = _workspace(wsb, dev, "nmsb")
        check(lib().m3d_nms3d_batched(_ptr(dets), C.c_size_t(7 * n), None, 0, 1, n, C.c_float(np.float32(thresh)), int(bool(by_volume)),
                                      0, None, _ptr(keep), _ptr(num), _ptr(ws), C.c_size_t(wsb), _stream()), "nms3d_batched")
        torch.cuda.current_stream().synchronize()
        return keep[0, :int(num[0])]
    keep = torch.empty((n,), dtype=torch.int64, device=dets.device)
    num = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    wsb = lib().m3d_nms3d_workspace_bytes(n)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dets.device)
    check(lib().m3d_nms3d(_ptr(dets), n, C.c_float(np.float32(thresh)), int(bool(by_volume)), _ptr(keep), _ptr(num),
                          _ptr(ws), C.c_size_t(wsb), _stream()), "nms3d")
    return keep[:int(num.item())]


def bbox_overlaps3d(boxes, query):
    _need_gpu(boxes, query)
    boxes, query = _f32c(boxes), _f32c(query)
    out = torch.empty((boxes.shape[0], query.shape[0]), dtype=torch.float32, device=boxes.device)
    check(lib().m3d_bbox_overlaps3d(_ptr(boxes), boxes.shape[0], _ptr(query), query.shape[0], _ptr(out), _stream()),
          "bbox_overlaps3d")
    return out


def bbox_transform3d(boxes, deltas, weights=(1.,) * 6, clip_to=None, xform_clip=BBOX_XFORM_CLIP):
    """boxes [N,6], deltas [N,6k] -> [N,6k]; clip_to=(slices,height,width) fuses clip_tiled_boxes_3d."""
    _need_gpu(boxes, deltas)
    boxes, deltas = _f32c(boxes), _f32c(deltas)
    n, k = boxes.shape[0], deltas.shape[1] // 6
    out = torch.empty_like(deltas)
    w = (C.c_double * 6)(*[float(x) for x in weights])
    cs, ch, cw = (0., 0., 0.) if clip_to is None else [float(v) for v in clip_to]
    check(lib().m3d_bbox_transform3d(_ptr(boxes), _ptr(deltas), n, k, w, C.c_double(xform_clip), C.c_double(cs),
                                     C.c_double(ch), C.c_double(cw), _ptr(out), _stream()), "bbox_transform3d")
    return out


def generate_proposals3d(scores, deltas, anchors, feat_stride, im_info, pre_nms_topN, post_nms_topN, nms_thresh,
                         min_size=0.0, batch_index=0, xform_clip=BBOX_XFORM_CLIP):
    """scores [A,S,H,W], deltas [6A,S,H,W] CUDA fp32; anchors [A,6] float64 numpy.
    Returns (rois [R,7] f32, probs [R,1] f32, keep_idx [R] i64) on device."""
    _need_gpu(scores, deltas)
    scores, deltas = _f32c(scores), _f32c(deltas)
    A, S, H, W = scores.shape
    assert deltas.shape == (6 * A, S, H, W)
    total = A * S * H * W
    K = total if (pre_nms_topN <= 0 or pre_nms_topN >= total) else pre_nms_topN
    cap = K
    dev = scores.device
    rois = torch.empty((cap, 7), dtype=torch.float32, device=dev)
    probs = torch.empty((cap,), dtype=torch.float32, device=dev)
    kidx = torch.empty((cap,), dtype=torch.int64, device=dev)
    num = torch.zeros((1,), dtype=torch.int32, device=dev)
    wsb = lib().m3d_generate_proposals3d_workspace_bytes(A, S, H, W, int(pre_nms_topN))
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    anc = np.ascontiguousarray(anchors, dtype=np.float64)
    info = np.ascontiguousarray(im_info, dtype=np.float64)
    check(lib().m3d_generate_proposals3d(_ptr(scores), _ptr(deltas), A, S, H, W, anc.ctypes.data_as(C.c_void_p),
                                         C.c_double(feat_stride), info.ctypes.data_as(C.c_void_p), int(pre_nms_topN),
                                         int(post_nms_topN), C.c_float(np.float32(nms_thresh)), C.c_double(min_size),
                                         C.c_double(xform_clip), int(batch_index), _ptr(rois), _ptr(probs), _ptr(kidx),
                                         _ptr(num), _ptr(ws), C.c_size_t(wsb), _stream()), "generate_proposals3d")
    r = int(num.item())
    return rois[:r], probs[:r].unsqueeze(1), kidx[:r]


# ------------------------------------------------------------------ batched, fused box stages (csrc/box_fused.hip)
def fused_max_boxes():
    return int(lib().m3d_fused_max_boxes())


_ws_cache = {}


_pin_cache = {}


def _pinned_i32(device):
    """One pinned int32 per device: a kernel writes a count there, the caller synchronizes its stream and reads it at once."""
    t = _pin_cache.get(str(device))
    if t is None:
        t = _pin_cache[str(device)] = torch.zeros((1,), dtype=torch.int32).pin_memory()
    return t


def _workspace(nbytes, device, tag):
    """One scratch buffer per (stream, stage): the fused stages are launched back to back on a stream, so re-using the
    buffer is safe, and allocating ~1 MB per tile on every call would cost more than the kernels."""
    key = (tag, torch.cuda.current_stream().cuda_stream, str(device))
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _ws_cache[key] = torch.empty((int(nbytes),), dtype=torch.uint8, device=device)
    return ws


def generate_proposals3d_batched(scores, deltas, anchors, feat_stride, im_info, pre_nms_topN, post_nms_topN, nms_thresh,
                                 min_size=0.0, first_batch_index=0, xform_clip=BBOX_XFORM_CLIP):
    """scores [B,A,S,H,W], deltas [B,6A,S,H,W] -> (rois [B,rows,7], probs [B,rows], keep_idx [B,rows], num int32 [B]), all on
    the device, ONE launch, no host synchronisation; item b's valid rows are [0, num[b])."""
    _need_gpu(scores, deltas)
    scores, deltas = _f32c(scores), _f32c(deltas)
    B, A, S, H, W = scores.shape
    assert deltas.shape == (B, 6 * A, S, H, W)
    total = A * S * H * W
    K = total if (pre_nms_topN <= 0 or pre_nms_topN >= total) else pre_nms_topN
    # post_nms_topN only cuts the list when NMS runs (generate_proposals_3d.py:167-171)
    rows = K if (post_nms_topN <= 0 or nms_thresh <= 0) else min(K, post_nms_topN)
    dev = scores.device
    rois = torch.empty((B, rows, 7), dtype=torch.float32, device=dev)
    probs = torch.empty((B, rows), dtype=torch.float32, device=dev)
    kidx = torch.empty((B, rows), dtype=torch.int64, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    wsb = lib().m3d_generate_proposals3d_batched_workspace_bytes(B, A, S, H, W, int(pre_nms_topN))
    ws = _workspace(wsb, dev, "prop")
    anc = np.ascontiguousarray(anchors, dtype=np.float64)
    info = np.ascontiguousarray(im_info, dtype=np.float64)
    check(lib().m3d_generate_proposals3d_batched(_ptr(scores), _ptr(deltas), B, A, S, H, W, anc.ctypes.data_as(C.c_void_p),
                                                 C.c_double(feat_stride), info.ctypes.data_as(C.c_void_p), int(pre_nms_topN),
                                                 int(post_nms_topN), C.c_float(np.float32(nms_thresh)), C.c_double(min_size),
                                                 C.c_double(xform_clip), int(first_batch_index), rows, _ptr(rois), _ptr(probs),
                                                 _ptr(kidx), _ptr(num), _ptr(ws), C.c_size_t(wsb), _stream()),
          "generate_proposals3d_batched")
    return rois, probs, kidx, num


def compact_rows(src, counts, offsets=None):
    """src [B,rows,...] (contiguous, 4-byte elements or int64), counts int32 [B] on the device -> (packed [B*rows,...] whose first
    sum(counts) rows are item 0's valid rows, item 1's, ..., offsets int32 [B+1] on the device).  No host synchronisation: the caller
    slices `packed[:total]` once it knows the total."""
    _need_gpu(src, counts)
    assert src.is_contiguous() and counts.dtype == torch.int32 and counts.is_contiguous() and src.dim() >= 2
    B, rows = src.shape[0], src.shape[1]
    row_bytes = src[0, 0].numel() * src.element_size()
    out = torch.empty((B * rows,) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
    if offsets is None:
        offsets = torch.empty((B + 1,), dtype=torch.int32, device=src.device)
    check(lib().m3d_compact_rows(_ptr(src), C.c_size_t(rows * row_bytes), C.c_size_t(row_bytes), _ptr(counts), B, rows, _ptr(out),
                                 _ptr(offsets), _stream()), "compact_rows")
    return out, offsets


def compact_rows2(src_a, src_b, counts, host_counts=None):
    """compact_rows of two row sets with the same counts in ONE launch -> (packed_a, packed_b, offsets int32 [B+1]).  host_counts: a
    PINNED int32 tensor [B] the kernel itself fills with the counts - record an event behind this call and wait for that."""
    _need_gpu(src_a, src_b, counts)
    assert src_a.is_contiguous() and src_b.is_contiguous() and counts.dtype == torch.int32 and counts.is_contiguous()
    B, rows = src_a.shape[0], src_a.shape[1]
    assert src_b.shape[0] == B and src_b.shape[1] == rows
    rb_a = src_a[0, 0].numel() * src_a.element_size()
    rb_b = src_b[0, 0].numel() * src_b.element_size()
    out_a = torch.empty((B * rows,) + tuple(src_a.shape[2:]), dtype=src_a.dtype, device=src_a.device)
    out_b = torch.empty((B * rows,) + tuple(src_b.shape[2:]), dtype=src_b.dtype, device=src_b.device)
    offsets = torch.empty((B + 1,), dtype=torch.int32, device=src_a.device)
    if host_counts is not None:
        assert host_counts.is_pinned() and host_counts.dtype == torch.int32 and host_counts.numel() >= B
    check(lib().m3d_compact_rows2(_ptr(src_a), C.c_size_t(rows * rb_a), C.c_size_t(rb_a), _ptr(src_b), C.c_size_t(rows * rb_b), C.c_size_t(rb_b),
                                  _ptr(counts), B, rows, _ptr(out_a), _ptr(out_b), _ptr(offsets),
                                  C.c_void_p(host_counts.data_ptr()) if host_counts is not None else None, _stream()), "compact_rows2")
    return out_a, out_b, offsets


def box_head_outputs(outs, rois, num_classes, weights, clip_to=None, xform_clip=BBOX_XFORM_CLIP):
    """outs [M, nc + 6 nc] (cls_score and bbox_pred as one GEMM), rois [M,7] -> (cls [M,nc] softmax, bbox [M,6nc] raw deltas,
    pred [M,6nc] decoded (+ clipped to clip_to = (slices, height, width))); one launch (m3d_box_head_outputs)."""
    _need_gpu(outs, rois)
    outs, rois = _f32c(outs), _f32c(rois)
    M, nc = int(outs.shape[0]), int(num_classes)
    assert outs.shape[1] == 7 * nc and rois.shape == (M, 7)
    cls = torch.empty((M, nc), dtype=torch.float32, device=outs.device)
    bbox = torch.empty((M, 6 * nc), dtype=torch.float32, device=outs.device)
    pred = torch.empty((M, 6 * nc), dtype=torch.float32, device=outs.device)
    w = (C.c_double * 6)(*[float(x) for x in weights])
    cs, ch, cw = (0., 0., 0.) if clip_to is None else [float(v) for v in clip_to]
    check(lib().m3d_box_head_outputs(_ptr(outs), _ptr(rois), M, nc, w, C.c_double(xform_clip), C.c_double(cs), C.c_double(ch), C.c_double(cw),
                                     _ptr(cls), _ptr(bbox), _ptr(pred), _stream()), "box_head_outputs")
    return cls, bbox, pred


def box_results3d_batched(scores, boxes, keep_idx, offsets, num_classes, score_thresh, nms_thresh, detections_per_im, max_rows):
    """scores [R,nc], boxes [R,6nc], keep_idx int64 [R] or None, offsets int32 [B+1] (device) ->
    (cls_boxes [B,nc,max_rows,7], cls_keep int64 [B,nc,max_rows], counts int32 [B,nc]); ONE launch, no host sync."""
    _need_gpu(scores, boxes, offsets)
    scores, boxes = _f32c(scores), _f32c(boxes)
    assert offsets.dtype == torch.int32 and offsets.is_contiguous()
    B = offsets.numel() - 1
    dev = scores.device
    cls_boxes = torch.empty((B, num_classes, max_rows, 7), dtype=torch.float32, device=dev)
    cls_keep = torch.empty((B, num_classes, max_rows), dtype=torch.int64, device=dev)
    counts = torch.empty((B, num_classes), dtype=torch.int32, device=dev)
    wsb = lib().m3d_box_results3d_batched_workspace_bytes(B)
    ws = _workspace(wsb, dev, "boxres")
    if keep_idx is not None:
        assert keep_idx.dtype == torch.int64 and keep_idx.is_contiguous()
    check(lib().m3d_box_results3d_batched(_ptr(scores), _ptr(boxes), _ptr(keep_idx), _ptr(offsets), B, int(num_classes),
                                          C.c_float(np.float32(score_thresh)), C.c_float(np.float32(nms_thresh)),
                                          int(detections_per_im), int(max_rows), _ptr(cls_boxes), _ptr(cls_keep), _ptr(counts),
                                          _ptr(ws), C.c_size_t(wsb), _stream()), "box_results3d_batched")
    return cls_boxes, cls_keep, counts


def nms3d_batched(dets, counts, thresh, by_volume=False, pack_cap=None, want_keep=True):
    """dets [B,n,7] (dim-0 stride arbitrary, inner dims contiguous), counts int32 view [B] (any stride) or None ->
    dict(keep int64 [B,n], num int32 [B], packed [B,pack_cap+1,7] if pack_cap is not None); ONE launch, no host sync."""
    _need_gpu(dets)
    assert dets.dim() == 3 and dets.shape[2] == 7 and dets.dtype == torch.float32 and dets.stride(2) == 1 and dets.stride(1) == 7
    B, n = dets.shape[0], dets.shape[1]
    dev = dets.device
    keep = torch.empty((B, n), dtype=torch.int64, device=dev) if want_keep else None
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    packed = torch.empty((B, pack_cap + 1, 7), dtype=torch.float32, device=dev) if pack_cap is not None else None
    cstride = 0
    if counts is not None:
        assert counts.dtype == torch.int32 and counts.dim() == 1 and counts.numel() == B
        cstride = counts.stride(0)
    wsb = lib().m3d_nms3d_batched_workspace_bytes(B)
    ws = _workspace(wsb, dev, "nmsb")
    check(lib().m3d_nms3d_batched(_ptr(dets), C.c_size_t(dets.stride(0)), _ptr(counts), int(cstride), B, n,
                                  C.c_float(np.float32(thresh)), int(bool(by_volume)), int(pack_cap or 0), _ptr(packed), _ptr(keep),
                                  _ptr(num), _ptr(ws), C.c_size_t(wsb), _stream()), "nms3d_batched")
    return dict(keep=keep, num=num, packed=packed)


# ------------------------------------------------------------------ conv / pool
class PackedConv3d:
    """A Conv3d weight packed once into MFMA A-fragment order (m3d_conv3d_pack_weights)."""

    def __init__(self, weight, mode=W_PLAIN):
        _need_gpu(weight)
        weight = _f32c(weight)
        self.cout_w, self.cin_w, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
        assert weight.shape[2] == weight.shape[3] == weight.shape[4]
        self.mode = mode
        dgrad = mode in (W_DGRAD, W_DGRAD_RELU)
        self.cin = self.cout_w if dgrad else self.cin_w      # logical conv this pack implements
        self.cout = self.cin_w if dgrad else self.cout_w
        nbytes = lib().m3d_conv3d_packed_weight_bytes(self.cin_w, self.cout_w, self.k, mode)
        self.packed = torch.empty((nbytes // 4,), dtype=torch.float32, device=weight.device)
        check(lib().m3d_conv3d_pack_weights(_ptr(weight), self.cin_w, self.cout_w, self.k, mode, _ptr(self.packed), _stream()),
              "conv3d_pack_weights")

    def supports_pool(self, width, voxels=None):
        """Fused conv+pool exists for the 5^3 stem and for k=3 on 32-wide x blocks; it uses the big 32x4x4 tile, so
        it only pays when that tile still yields >= 512 workgroups (same rule as the C dispatcher)."""
        if self.k == 5 and self.cin == 1 and self.cout <= 32:
            return True
        if self.k != 3 or width < 24:
            return False
        return voxels is None or (voxels // 512) * ((self.cout + 63) // 64) >= 512

    def pooled(self, x, scale=None, shift=None, relu=False, in_offset=None, return_argmax=False):
        """conv + scale/shift + ReLU + MaxPool3d(2,2) in one kernel (m3d_conv3d_forward_pool2)."""
        _need_gpu(x)
        x = _f32c(x)
        B, Cin, D, H, W = x.shape
        if Cin != self.cin:
            raise ValueError("expected %d input channels, got %d" % (self.cin, Cin))
        out = torch.empty((B, self.cout, D // 2, H // 2, W // 2), dtype=torch.float32, device=x.device)
        am = torch.empty(out.shape, dtype=torch.uint8, device=x.device) if return_argmax else None
        check(lib().m3d_conv3d_forward_pool2(_ptr(x), _ptr(self.packed), _ptr(out), _ptr(am), B, Cin, self.cout, D, H, W,
                                             self.k, _ptr(in_offset), _ptr(scale), _ptr(shift), int(bool(relu)), _stream()),
              "conv3d_forward_pool2")
        return (out, am) if return_argmax else out

    def split_sigmoid(self, x, split, shift=None):
        """One conv, two epilogues (m3d_conv3d_forward_split_sigmoid): (sigmoid(conv)[:, :split], conv[:, split:]) as two contiguous
        tensors - the RPN's cls_score + sigmoid and bbox_pred heads (rpn_heads.py:96-98,116)."""
        _need_gpu(x)
        x = _f32c(x)
        B, Cin, D, H, W = x.shape
        if Cin != self.cin:
            raise ValueError("expected %d input channels, got %d" % (self.cin, Cin))
        a = torch.empty((B, split, D, H, W), dtype=torch.float32, device=x.device)
        b = torch.empty((B, self.cout - split, D, H, W), dtype=torch.float32, device=x.device)
        check(lib().m3d_conv3d_forward_split_sigmoid(_ptr(x), _ptr(self.packed), _ptr(a), _ptr(b), B, Cin, self.cout, int(split), D, H, W, self.k,
                                                     _ptr(shift), _stream()), "conv3d_forward_split_sigmoid")
        return a, b

    def __call__(self, x, scale=None, shift=None, relu=False, in_offset=None, mul=None, out=None, dilation=1):
        _need_gpu(x)
        x = _f32c(x)
        B, Cin, D, H, W = x.shape
        if Cin != self.cin:
            raise ValueError("expected %d input channels, got %d" % (self.cin, Cin))
        if out is None:
            out = torch.empty((B, self.cout, D, H, W), dtype=torch.float32, device=x.device)
        for t in (scale, shift):
            if t is not None:
                assert t.is_cuda and t.dtype == torch.float32 and t.numel() == self.cout and t.is_contiguous()
        if dilation != 1:                                     # mask head (mask_rcnn_heads.py:148-151): padding = dilation
            assert in_offset is None and mul is None
            check(lib().m3d_conv3d_forward_dilated(_ptr(x), _ptr(self.packed), _ptr(out), B, Cin, self.cout, D, H, W, self.k,
                                                   int(dilation), _ptr(scale), _ptr(shift), int(bool(relu)), _stream()),
                  "conv3d_forward_dilated")
            return out
        if mul is not None:
            assert mul.shape == out.shape and mul.is_contiguous() and mul.dtype == torch.float32
        check(lib().m3d_conv3d_forward(_ptr(x), _ptr(self.packed), _ptr(out), B, Cin, self.cout, D, H, W, self.k,
                                       _ptr(in_offset), _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(mul), _stream()),
              "conv3d_forward")
        return out


class X3Conv3d:
    """A 3x3x3 'same' conv at fp32 accuracy on the bf16 matrix cores (csrc/conv3d_x3.hip: exact three-way bf16 cut of both operands,
    six products per fp32 product) - the exact direct kernel for the PRM norm convs: out = conv3d(x - in_offset, W or relu(W), padding 1).
    weight [cout, cin, 3, 3, 3]; mode W_PLAIN or W_RELU."""

    @staticmethod
    def supported(weight):
        return weight.dim() == 5 and tuple(weight.shape[2:]) == (3, 3, 3) and bool(lib().m3d_conv3d_x3_supported(weight.shape[1], weight.shape[0]))

    def __init__(self, weight, mode=W_PLAIN, f16=False):
        """f16 (round 6): the f16x2 split - two scaled fp16 pieces per operand, three products per fp32 product (m3d_conv3d_x3f_*); the call then
        needs `in_max` (a device scalar: max x with in_offset, max |x| without)."""
        _need_gpu(weight)
        weight = _f32c(weight)
        assert mode in (W_PLAIN, W_RELU) and X3Conv3d.supported(weight)
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.f16 = bool(f16)
        if self.f16:
            nbytes = lib().m3d_conv3d_x3f_packed_bytes(self.cin, self.cout)
            self.packed = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
            check(lib().m3d_conv3d_x3f_pack(_ptr(weight), self.cin, self.cout, int(mode == W_RELU), _ptr(self.packed), _stream()), "conv3d_x3f_pack")
            return
        nbytes = lib().m3d_conv3d_x3_packed_bytes(self.cin, self.cout)
        self.packed = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
        check(lib().m3d_conv3d_x3_pack(_ptr(weight), self.cin, self.cout, int(mode == W_RELU), _ptr(self.packed), _stream()), "conv3d_x3_pack")

    def workgroups(self, shape):
        """launch size for x of `shape` [B, cin, D, H, W]: the library's own count (64 output channels x 16 x 4 x 4 voxels per
        workgroup, times the K ranges m3d_conv3d_x3_forward_ws cuts a small launch into)"""
        B, _, D, H, W = shape
        return int(lib().m3d_conv3d_x3_launch_units(int(B), self.cin, self.cout, int(D), int(H), int(W)))

    def __call__(self, x, in_offset=None, out=None, in_max=None):
        _need_gpu(x)
        x = _f32c(x)
        B, Cin, D, H, W = x.shape
        if Cin != self.cin:
            raise ValueError("expected %d input channels, got %d" % (self.cin, Cin))
        if self.f16 and in_max is None:
            raise ValueError("X3Conv3d(f16=True) needs in_max (ops.reduce_minmax_multi / ops.absmax)")
        if out is None:
            out = torch.empty((B, self.cout, D, H, W), dtype=torch.float32, device=x.device)
        wsb = lib().m3d_conv3d_x3_workspace_bytes(B, Cin, self.cout, D, H, W)
        ws = None
        if wsb:
            key = (torch.cuda.current_stream().cuda_stream, x.device)           # one scratch buffer per stream (the norm convs run on their own)
            cache = self.__dict__.setdefault("_ws", {})
            ws = cache.get(key)
            if ws is None or ws.numel() < wsb:
                ws = cache[key] = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
        if self.f16:
            _need_gpu(in_max)
            check(lib().m3d_conv3d_x3f_forward_ws(_ptr(x), _ptr(self.packed), _ptr(out), B, Cin, self.cout, D, H, W, _ptr(in_offset), _ptr(in_max),
                                                  _ptr(ws), C.c_size_t(wsb), _stream()), "conv3d_x3f_forward")
            return out
        check(lib().m3d_conv3d_x3_forward_ws(_ptr(x), _ptr(self.packed), _ptr(out), B, Cin, self.cout, D, H, W, _ptr(in_offset), _ptr(ws),
                                             C.c_size_t(wsb), _stream()), "conv3d_x3_forward")
        return out


def maxpool3d_2x(x, return_argmax=False):
    _need_gpu(x)
    x = _f32c(x)
    B, Cc, D, H, W = x.shape
    out = torch.empty((B, Cc, D // 2, H // 2, W // 2), dtype=torch.float32, device=x.device)
    am = torch.empty(out.shape, dtype=torch.uint8, device=x.device) if return_argmax else None
    check(lib().m3d_maxpool3d_2x_forward(_ptr(x), _ptr(out), _ptr(am), B * Cc, D, H, W, _stream()), "maxpool3d_2x_forward")
    return (out, am) if return_argmax else out


def maxpool3d_2x_backward(grad_out, argmax, in_shape):
    _need_gpu(grad_out, argmax)
    grad_out = _f32c(grad_out)
    B, Cc, D, H, W = in_shape
    gin = torch.empty(in_shape, dtype=torch.float32, device=grad_out.device)
    check(lib().m3d_maxpool3d_2x_backward(_ptr(grad_out), _ptr(argmax), _ptr(gin), B * Cc, D, H, W, _stream()),
          "maxpool3d_2x_backward")
    return gin


def reduce_min(x):
    _need_gpu(x)
    x = _f32c(x)
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    wsb = lib().m3d_reduce_min_workspace_bytes()
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
    check(lib().m3d_reduce_min(_ptr(x), C.c_int64(x.numel()), _ptr(out), _ptr(ws), C.c_size_t(wsb), _stream()), "reduce_min")
    return out


REDUCE_MIN_MULTI_MAX = 12      # kMinMulti of csrc/pool_bn.hip: m3d_reduce_min_multi returns M3D_EINVAL beyond it


def reduce_min_multi(xs):
    """Minima of any number of tensors, two launches per group of up to 12 (the entry point's capacity) -> float32 [len(xs)]
    (slice i:i+1 is the device scalar of tensor i)."""
    _need_gpu(*xs)
    xs = [_f32c(x) for x in xs]
    n = len(xs)
    out = torch.empty((max(n, 1),), dtype=torch.float32, device=xs[0].device)
    wsb = lib().m3d_reduce_min_multi_workspace_bytes()
    for g0 in range(0, n, REDUCE_MIN_MULTI_MAX):
        grp = xs[g0:g0 + REDUCE_MIN_MULTI_MAX]
        m = len(grp)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=xs[0].device)      # per group: the groups' launches are in flight together
        ptrs = (C.c_void_p * m)(*[x.data_ptr() for x in grp])
        cnts = (C.c_int64 * m)(*[x.numel() for x in grp])
        check(lib().m3d_reduce_min_multi(ptrs, cnts, m, _ptr(out[g0:g0 + m]), _ptr(ws), C.c_size_t(wsb), _stream()), "reduce_min_multi")
    return out


def reduce_minmax_multi(xs):
    """(minima, maxima) of any number of tensors, two launches per group of up to 12 -> two float32 [len(xs)] device tensors."""
    _need_gpu(*xs)
    xs = [_f32c(x) for x in xs]
    n = len(xs)
    out = torch.empty((2, max(n, 1)), dtype=torch.float32, device=xs[0].device)
    wsb = lib().m3d_reduce_minmax_multi_workspace_bytes()
    for g0 in range(0, n, REDUCE_MIN_MULTI_MAX):
        grp = xs[g0:g0 + REDUCE_MIN_MULTI_MAX]
        m = len(grp)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=xs[0].device)
        ptrs = (C.c_void_p * m)(*[x.data_ptr() for x in grp])
        cnts = (C.c_int64 * m)(*[x.numel() for x in grp])
        check(lib().m3d_reduce_minmax_multi(ptrs, cnts, m, _ptr(out[0, g0:g0 + m]), _ptr(out[1, g0:g0 + m]), _ptr(ws), C.c_size_t(wsb), _stream()),
              "reduce_minmax_multi")
    return out[0], out[1]


def linear(x, weight, bias=None, relu=False, out=None):
    """act(x @ weight.T + bias) on the split-K fp32 MFMA GEMM (m3d_linear_forward): x [M,K], weight [N,K] (nn.Linear layout),
    bias [N] or None -> [M,N].  fast_rcnn_heads.py:84-85,114-115,15-19."""
    _need_gpu(x, weight, bias)
    x, weight = _f32c(x), _f32c(weight)
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError("linear: x is [%d,%d] but weight is %s" % (M, K, tuple(weight.shape)))
    if bias is not None:
        bias = _f32c(bias)
        assert bias.numel() == N
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if M == 0:
        return out
    wsb = lib().m3d_linear_workspace_bytes(M, N, K)
    ws = torch.empty((max(wsb, 16) // 4,), dtype=torch.float32, device=x.device)
    check(lib().m3d_linear_forward(_ptr(x), _ptr(weight), _ptr(bias), _ptr(out), M, N, K, int(bool(relu)), _ptr(ws),
                                   C.c_size_t(wsb), _stream()), "linear_forward")
    return out


class SplitLinear:
    """nn.Linear weights cut once into three bf16 planes for m3d_linear_bf16x3_forward (fp32 accuracy at the bf16 matrix rate:
    x = xh + xm + xl exactly, six bf16 MFMAs per fp32 product).  `supported(weight)`: K a multiple of 32."""

    @staticmethod
    def supported(weight):
        return weight.dim() == 2 and weight.shape[1] % 32 == 0 and weight.shape[0] >= 64

    def __init__(self, weight, bias=None):
        _need_gpu(weight, bias)
        weight = _f32c(weight)
        self.N, self.K = int(weight.shape[0]), int(weight.shape[1])
        nb = lib().m3d_linear_bf16x3_packed_bytes(self.N, self.K)
        if nb == 0:
            raise ValueError("SplitLinear: K = %d is not a multiple of 32" % self.K)
        self.packed = torch.empty((nb,), dtype=torch.uint8, device=weight.device)
        check(lib().m3d_linear_bf16x3_pack(_ptr(weight), self.N, self.K, _ptr(self.packed), _stream()), "linear_bf16x3_pack")
        self.bias = None if bias is None else _f32c(bias)
        self.weight = weight                    # the many-rows kernel (256 x 256 tiles) cuts the fp32 weight itself
        self.tail_rows = 64                     # <= this many rows past a multiple of 256 go to the fp32 kernel's ragged-tile path
        self.big_rows = 2048                    # from this many rows on (N >= 256): m3d_linear_bf16x3_w32_forward (2560 rows: 2.21 vs 2.35 ms)

    def __call__(self, x, relu=False, out=None, variant=None):
        """variant: None = by shape, "packed" (128 / 256 x 128 tiles on the packed planes), "w32" (256 x 256 tiles, fp32 weight)."""
        _need_gpu(x)
        x = _f32c(x)
        M, K = x.shape
        if K != self.K:
            raise ValueError("SplitLinear: x is [%d,%d] but the weight is [%d,%d]" % (M, K, self.N, self.K))
        if out is None:
            out = torch.empty((M, self.N), dtype=torch.float32, device=x.device)
        if M == 0:
            return out
        if variant is None:
            if M <= 32:                                   # a handful of rows: the fp32-input kernel's ragged-tile path (M = 2: 0.09 vs 0.18 ms)
                return linear(x, self.weight, self.bias, relu=relu, out=out)
            variant = "w32" if (M >= self.big_rows and self.N >= 256) else "packed"
            # a few rows past a multiple of the 256-row tile (M = 1281 = 5 x 256 + 1 pads 20 % more rows): those rows go through the
            # fp32-input kernel's ragged-tile path instead (same accuracy class), the rest through full tiles
            rem = M % 256
            if variant == "packed" and M > 512 and 0 < rem <= self.tail_rows:
                self(x[:M - rem], relu=relu, out=out[:M - rem], variant="packed")
                linear(x[M - rem:], self.weight, self.bias, relu=relu, out=out[M - rem:])
                return out
        if variant == "w32":
            wsb = lib().m3d_linear_bf16x3_w32_workspace_bytes(M, self.N, K)
            ws = torch.empty((max(wsb, 16) // 4,), dtype=torch.float32, device=x.device)
            check(lib().m3d_linear_bf16x3_w32_forward(_ptr(x), _ptr(self.weight), _ptr(self.bias), _ptr(out), M, self.N, K, int(bool(relu)),
                                                      _ptr(ws), C.c_size_t(wsb), _stream()), "linear_bf16x3_w32_forward")
            return out
        wsb = lib().m3d_linear_bf16x3_workspace_bytes(M, self.N, K)
        ws = torch.empty((max(wsb, 16) // 4,), dtype=torch.float32, device=x.device)
        check(lib().m3d_linear_bf16x3_forward(_ptr(x), _ptr(self.packed), _ptr(self.bias), _ptr(out), M, self.N, K, int(bool(relu)),
                                              _ptr(ws), C.c_size_t(wsb), _stream()), "linear_bf16x3_forward")
        return out


def absmax(x):
    """max |x| of a float32 CUDA tensor as a 1-element device tensor (m3d_absmax; no host read).  Used as the bound of |RoIAlign
    output| that sets the f16x2 GEMM's operand scale: the RoIAlign output is a convex combination of the feature map's values."""
    _need_gpu(x)
    x = _f32c(x)
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    check(lib().m3d_absmax(_ptr(x), C.c_longlong(x.numel()), _ptr(out), _stream()), "absmax")
    return out


class SplitLinearF16:
    """nn.Linear on the f16 matrix cores at fp32 accuracy with THREE products per fp32 product (csrc/fc_gemm.hip, "f16x2 split": both
    operands scaled by a power of two and cut into two fp16 numbers, 22 bits; the dropped terms are below the rounding of the fp32
    accumulation).  Weights cut once (4 bytes per element).  __call__(x, relu, out, x_bound): x_bound = a 1-element device tensor
    >= max|x| (e.g. ops.absmax of the feature map the RoIAlign read); None: the library sweeps x itself."""

    @staticmethod
    def supported(weight):
        return SplitLinear.supported(weight)

    def __init__(self, weight, bias=None):
        _need_gpu(weight, bias)
        weight = _f32c(weight)
        self.N, self.K = int(weight.shape[0]), int(weight.shape[1])
        nb = lib().m3d_linear_f16x2_packed_bytes(self.N, self.K)
        if nb == 0:
            raise ValueError("SplitLinearF16: K = %d is not a multiple of 32" % self.K)
        self.packed = torch.empty((nb,), dtype=torch.uint8, device=weight.device)
        check(lib().m3d_linear_f16x2_pack(_ptr(weight), self.N, self.K, _ptr(self.packed), _stream()), "linear_f16x2_pack")
        self.bias = None if bias is None else _f32c(bias)
        self.weight = weight

    def __call__(self, x, relu=False, out=None, x_bound=None):
        _need_gpu(x)
        x = _f32c(x)
        M, K = x.shape
        if K != self.K:
            raise ValueError("SplitLinearF16: x is [%d,%d] but the weight is [%d,%d]" % (M, K, self.N, self.K))
        if out is None:
            out = torch.empty((M, self.N), dtype=torch.float32, device=x.device)
        if M == 0:
            return out
        if M <= 32:                                       # a handful of rows: the fp32-input kernel's ragged-tile path
            return linear(x, self.weight, self.bias, relu=relu, out=out)
        if x_bound is not None:
            _need_gpu(x_bound)
            assert x_bound.dtype == torch.float32 and x_bound.numel() == 1
        wsb = lib().m3d_linear_f16x2_workspace_bytes(M, self.N, K)
        ws = torch.empty((wsb // 4,), dtype=torch.float32, device=x.device)
        check(lib().m3d_linear_f16x2_forward(_ptr(x), _ptr(self.packed), _ptr(self.bias), _ptr(out), M, self.N, K, int(bool(relu)),
                                             _ptr(x_bound), _ptr(ws), C.c_size_t(wsb), _stream()), "linear_f16x2_forward")
        return out


def linear_roi_fused(split, features, rois, spatial_scale, relu=False):
    """f-1 A/B: act(RoIAlign3D(features, rois).view(R, -1) @ W.T + b) with the gather inside the GEMM's operand loader
    (m3d_linear_bf16x3_roi_forward; 7^3 bins, sampling grid 2): the [R, C*343] intermediate is never written.  split: SplitLinear of the
    layer.  The product path stays RoIAlign + SplitLinear (two launches): see profiles/r04_f1_ab.json."""
    _need_gpu(features, rois)
    features, rois = _f32c(features), _f32c(rois)
    B, Cc, S, H, W = features.shape
    R = int(rois.shape[0])
    assert split.K == Cc * 343 and rois.shape[1] == 7
    out = torch.empty((R, split.N), dtype=torch.float32, device=features.device)
    if R == 0:
        return out
    tab = torch.empty((R, 42, 4), dtype=torch.int32, device=features.device)
    rb = torch.empty((R,), dtype=torch.int32, device=features.device)
    check(lib().m3d_roi_align3d_tap_tables(_ptr(rois), R, C.c_float(spatial_scale), B, S, H, W, _ptr(tab), _ptr(rb), _stream()), "roi_align3d_tap_tables")
    wsb = lib().m3d_linear_bf16x3_roi_workspace_bytes(R, split.N, split.K)
    ws = torch.empty((max(wsb, 16) // 4,), dtype=torch.float32, device=features.device)
    check(lib().m3d_linear_bf16x3_roi_forward(_ptr(features), B, Cc, S, H, W, _ptr(tab), _ptr(rb), _ptr(split.packed), _ptr(split.bias), _ptr(out),
                                              R, split.N, int(bool(relu)), _ptr(ws), C.c_size_t(wsb), _stream()), "linear_bf16x3_roi_forward")
    return out


def mask_paste3d(masks, channel, boxes_int, shape, thresh=0.5):
    """segm_results' paste (lib/core/test.py:886-945) for all detections at once.  masks [R, C, M, M, M] CUDA fp32; channel [R]
    ints; boxes_int [R, 6] = expand_boxes(ref_boxes, (M+2)/M).astype(int32) (host); shape = (im_s, im_h, im_w).
    Returns uint8 [R, im_s, im_h, im_w] on the device.  The Gaussian taps of skimage's anti-aliasing filter are computed here,
    on the host, exactly as scipy.ndimage does (np.exp, NumPy's pairwise sum), per detection and axis."""
    _need_gpu(masks)
    masks = _f32c(masks)
    R, Cc, M = int(masks.shape[0]), int(masks.shape[1]), int(masks.shape[2])
    S, H, W = (int(v) for v in shape)
    out = torch.empty((R, S, H, W), dtype=torch.uint8, device=masks.device)
    if R == 0:
        return out
    rb = np.ascontiguousarray(boxes_int, dtype=np.int32).reshape(R, 6)
    P = M + 2
    size = np.maximum(rb[:, [5, 4, 3]] - rb[:, [2, 1, 0]] + 1, 1).astype(np.float64)           # (s, h, w)
    sigma = np.maximum(0, (float(P) / size - 1) / 2)                                           # skimage: (in/out - 1) / 2
    radius = np.where(sigma > 1e-15, (4.0 * sigma + 0.5).astype(np.int64), 0).astype(np.int32)  # scipy: int(truncate * sd + 0.5)
    ws_ = int(radius.max()) + 1
    weights = np.zeros((R, 3, ws_), dtype=np.float64)
    for (r, a) in zip(*np.nonzero(radius)):
        rad, sg = int(radius[r, a]), float(sigma[r, a])
        x = np.arange(-rad, rad + 1)
        phi = np.exp(-0.5 / (sg * sg) * x ** 2)
        phi = phi / phi.sum()
        weights[r, a, :rad + 1] = phi[rad:]                                                    # symmetric: [d] = distance d
    dev = masks.device
    d_ch = torch.from_numpy(np.ascontiguousarray(channel, dtype=np.int32)).to(dev)
    d_rb = torch.from_numpy(rb).to(dev)
    d_rad = torch.from_numpy(np.ascontiguousarray(radius)).to(dev)
    d_w = torch.from_numpy(weights).to(dev)
    wsb = lib().m3d_mask_paste3d_workspace_bytes(R, M)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    check(lib().m3d_mask_paste3d(_ptr(masks), R, Cc, M, _ptr(d_ch), _ptr(d_rb), _ptr(d_rad), _ptr(d_w), ws_, C.c_float(thresh),
                                 S, H, W, _ptr(out), _ptr(ws), C.c_size_t(wsb), _stream()), "mask_paste3d")
    return out


def norm1(vol, f32_arith=True, out=None, return_stats=False):
    """mask = vol > 0; (vol - mean(vol[mask])) / std(vol[mask]) on the device -> fp32 tensor of vol's shape.
    vol: uint16 (raw) or float32 CUDA tensor.  f32_arith=True mirrors prep_im_for_blob (blob.py:179-184, float32
    arithmetic); False mirrors infer_simple.py:180-183 (float64, rounded to fp32 once)."""
    _need_gpu(vol)
    vol = vol.contiguous()
    if vol.dtype not in (torch.uint16, torch.float32):
        raise TypeError("norm1 takes uint16 or float32 volumes")
    if out is None:
        out = torch.empty(vol.shape, dtype=torch.float32, device=vol.device)
    wsb = lib().m3d_norm1_workspace_bytes()
    ws = torch.empty((wsb // 8,), dtype=torch.float64, device=vol.device)
    stats = torch.empty((3,), dtype=torch.float64, device=vol.device) if return_stats else None
    check(lib().m3d_norm1(_ptr(vol), 0 if vol.dtype == torch.uint16 else 1, C.c_int64(vol.numel()), int(bool(f32_arith)),
                          _ptr(out), _ptr(stats), _ptr(ws), C.c_size_t(wsb), _stream()), "norm1")
    return (out, stats) if return_stats else out


def norm1_batched(vols, f32_arith=True, out=None):
    """norm1 of every volume of a batch [B, ...] with its own statistics, one launch per pass for the whole batch."""
    _need_gpu(vols)
    vols = vols.contiguous()
    if vols.dtype not in (torch.uint16, torch.float32):
        raise TypeError("norm1 takes uint16 or float32 volumes")
    B = vols.shape[0]
    n = vols[0].numel()
    if out is None:
        out = torch.empty(vols.shape, dtype=torch.float32, device=vols.device)
    assert out.is_contiguous() and out.numel() == vols.numel() and out.dtype == torch.float32
    wsb = lib().m3d_norm1_workspace_bytes() * B
    ws = torch.empty((wsb // 8,), dtype=torch.float64, device=vols.device)
    check(lib().m3d_norm1_batched(_ptr(vols), 0 if vols.dtype == torch.uint16 else 1, B, C.c_int64(n), int(bool(f32_arith)),
                                  _ptr(out), None, _ptr(ws), C.c_size_t(wsb), _stream()), "norm1_batched")
    return out


# ------------------------------------------------------------------ Otsu 2D
def otsu2d_batch(image, prm, offsets, max_gray_range=4096):
    """image, prm: flat uint16 CUDA tensors (crops concatenated); offsets int64 [R+1] CUDA.
    Returns (mask uint8 flat, kb int32 [R,2], status int32 [R])."""
    _need_gpu(image, prm, offsets)
    assert image.dtype == torch.uint16 and prm.dtype == torch.uint16 and offsets.dtype == torch.int64
    R = offsets.numel() - 1
    mask = torch.empty(image.shape, dtype=torch.uint8, device=image.device)
    kb = torch.empty((R, 2), dtype=torch.int32, device=image.device)          # otsu_eval_kernel writes both for every RoI, on every path
    status = torch.empty((R,), dtype=torch.int32, device=image.device)
    wsb = lib().m3d_otsu2d_workspace_bytes(R, int(max_gray_range))
    ws = torch.empty((wsb,), dtype=torch.uint8, device=image.device)
    check(lib().m3d_otsu2d_batch(_ptr(image), _ptr(prm), _ptr(offsets), R, int(max_gray_range), _ptr(mask), _ptr(kb),
                                 _ptr(status), _ptr(ws), C.c_size_t(wsb), _stream()), "otsu2d_batch")
    return mask, kb, status


# ------------------------------------------------------------------ PRM window kernels
def conv3d_windowed(packed, x, full, full_off, origins):
    """Batched same-conv on cropped windows + fused (full[co, origin+pos] - off) multiply (m3d_conv3d_forward_windowed)."""
    _need_gpu(x, full, origins)
    x = _f32c(x)
    B, Cin, D, H, W = x.shape
    assert Cin == packed.cin and full.shape[0] == packed.cout and full.dim() == 4 and full.is_contiguous()
    assert origins.dtype == torch.int32 and origins.shape == (B, 3) and origins.is_contiguous()
    out = torch.empty((B, packed.cout, D, H, W), dtype=torch.float32, device=x.device)
    check(lib().m3d_conv3d_forward_windowed(_ptr(x), _ptr(packed.packed), _ptr(out), B, Cin, packed.cout, D, H, W, packed.k,
                                            _ptr(full), _ptr(full_off), _ptr(origins), full.shape[1], full.shape[2],
                                            full.shape[3], _stream()), "conv3d_forward_windowed")
    return out


class SmallWindowDgrad:
    """Backward-data of a 3^3 conv with relu(W) on batches of 3^3 / 5^3 / 7^3 windows, all peaks in one dense GEMM
    (csrc/prm_small.hip); weight = the forward conv's [cout, cin, 3, 3, 3], packed once."""
    SIZES = (3, 5, 7)

    def __init__(self, weight, f16=True):
        """f16 (round 6, default where the forward conv's cout is a multiple of 16): the f16x2 split on v_mfma_f32_32x32x16_f16 (csrc/
        prm_small_f16.hip: two scaled fp16 pieces per operand, three products, per-peak gradient scales) instead of the fp32 MFMA GEMM."""
        _need_gpu(weight)
        w = _f32c(weight)
        assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3)
        self.cout_fwd, self.cin_fwd = int(w.shape[0]), int(w.shape[1])
        self.f16 = bool(f16) and bool(lib().m3d_prm_small_dgrad_f16_supported(self.cout_fwd, self.cin_fwd))
        if self.f16:
            nbytes = lib().m3d_prm_small_dgrad_f16_packed_bytes(self.cout_fwd, self.cin_fwd)
            self.packed = torch.empty((nbytes // 4,), dtype=torch.float32, device=w.device)
            check(lib().m3d_prm_small_dgrad_f16_pack(_ptr(w), self.cout_fwd, self.cin_fwd, _ptr(self.packed), _stream()), "prm_small_dgrad_f16_pack")
            return
        nbytes = lib().m3d_prm_small_dgrad_packed_bytes(self.cout_fwd, self.cin_fwd)
        self.packed = torch.empty((nbytes // 4,), dtype=torch.float32, device=w.device)
        check(lib().m3d_prm_small_dgrad_pack(_ptr(w), self.cout_fwd, self.cin_fwd, _ptr(self.packed), _stream()), "prm_small_dgrad_pack")

    def __call__(self, gn, full, full_off, origins):
        """gn [P, cout_fwd, n, n, n]; full [cin_fwd, D, H, W]; origins int32 [P,3] -> [P, cin_fwd, n, n, n]."""
        _need_gpu(gn, full, full_off, origins)
        gn = _f32c(gn)
        P, Cc, n = gn.shape[0], gn.shape[1], gn.shape[2]
        assert Cc == self.cout_fwd and n in self.SIZES and full.shape[0] == self.cin_fwd
        out = torch.empty((P, self.cin_fwd, n, n, n), dtype=torch.float32, device=gn.device)
        if self.f16:
            wsb = lib().m3d_prm_small_dgrad_f16_workspace_bytes(P)
            ws = torch.empty((wsb // 4,), dtype=torch.float32, device=gn.device)
            check(lib().m3d_prm_small_dgrad_f16(_ptr(gn), _ptr(self.packed), P, self.cout_fwd, self.cin_fwd, n, _ptr(_f32c(full)), _ptr(full_off),
                                                _ptr(origins), full.shape[1], full.shape[2], full.shape[3], _ptr(out), _ptr(ws), C.c_size_t(wsb),
                                                _stream()), "prm_small_dgrad_f16")
            return out
        check(lib().m3d_prm_small_dgrad(_ptr(gn), _ptr(self.packed), P, self.cout_fwd, self.cin_fwd, n, _ptr(_f32c(full)), _ptr(full_off),
                                        _ptr(origins), full.shape[1], full.shape[2], full.shape[3], _ptr(out), _stream()),
              "prm_small_dgrad")
        return out


def prm_seed(peaks, prob, norm_cls, w_cls, h, h_off, return_origin=False):
    """peaks int32 [P,4] (a,s,h,w); prob/norm_cls [A,S,H,W]; w_cls [A,C]; h [C,S,H,W] -> [P,C,1,1,1] (+ with return_origin the peaks'
    (s,h,w) as int32 [P,3], written by the same launch)."""
    _need_gpu(peaks, prob, norm_cls, w_cls, h, h_off)
    P = peaks.shape[0]
    A, S, H, W = prob.shape
    Cc = h.shape[0]
    out = torch.empty((P, Cc, 1, 1, 1), dtype=torch.float32, device=prob.device)
    org = torch.empty((P, 3), dtype=torch.int32, device=prob.device) if return_origin else None
    check(lib().m3d_prm_seed_ex(_ptr(peaks), P, _ptr(prob), _ptr(norm_cls), _ptr(w_cls), _ptr(h), _ptr(h_off), A, Cc, S, H, W,
                                _ptr(out), _ptr(org), _stream()), "prm_seed")
    return (out, org) if return_origin else out


class _StagingRing:
    """Small host arrays (index lists, box tables, offsets) on their way to the device: a ring of pinned 64 KB slots, each guarded by
    the event of its last copy.  `torch.from_numpy(a).to(device)` from pageable memory is a BLOCKING copy that first waits for
    everything queued on the stream - one idle gap on the GPU per call (tools/prm_timeline.sh showed 20-70 us each)."""
    SLOT = 1 << 16

    def __init__(self, n=32):
        self.pool = torch.empty((n * self.SLOT,), dtype=torch.uint8).pin_memory()       # ONE pinning call (each costs milliseconds)
        self.slots = [self.pool[k * self.SLOT:(k + 1) * self.SLOT] for k in range(n)]
        self.events, self.i, self.n, self.big = [None] * n, 0, n, []
        self.lock = threading.Lock()       # slot choice + fill + copy + event are one critical section: drivers own worker threads

    def put(self, arr, device):
        arr = np.ascontiguousarray(arr)
        nb = arr.nbytes
        dt = torch.from_numpy(np.empty((0,), arr.dtype)).dtype
        out = torch.empty(arr.shape, dtype=dt, device=device)
        if nb == 0:
            return out
        with self.lock:
            return self._put_locked(arr, nb, out)

    def _put_locked(self, arr, nb, out):
        if nb > self.SLOT:                                     # big tables: a one-off pinned buffer
            t = torch.from_numpy(arr).pin_memory()
            out.copy_(t, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            self.big = [(b, e) for b, e in self.big if not e.query()] + [(t, ev)]
            return out
        k = self.i % self.n
        self.i += 1
        if self.events[k] is not None:
            self.events[k].synchronize()                       # n copies ago: long done
        slot = self.slots[k]
        slot.numpy()[:nb] = arr.reshape(-1).view(np.uint8)
        out.view(torch.uint8).reshape(-1).copy_(slot[:nb], non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
        self.events[k] = ev
        return out


_staging = {}


def upload(arr, device="cuda"):
    """Host ndarray -> device tensor of the same dtype/shape through pinned staging, asynchronous on the current stream."""
    key = str(device)
    ring = _staging.get(key)
    if ring is None:
        ring = _staging[key] = _StagingRing()
    return ring.put(arr, device)


def upload_packed(arrays, device="cuda"):
    """Several small host arrays -> device tensors through ONE staged copy (each array at a 16-byte aligned offset of one buffer)."""
    arrays = [np.ascontiguousarray(a) for a in arrays]
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + 15) // 16 * 16
    host = np.zeros((max(total, 16),), np.uint8)
    for a, o in zip(arrays, offs):
        host[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    dev = upload(host, device)
    out = []
    for a, o in zip(arrays, offs):
        dt = torch.from_numpy(np.empty((0,), a.dtype)).dtype
        out.append(dev[o:o + a.nbytes].view(dt).view(a.shape) if a.nbytes else torch.empty(a.shape, dtype=dt, device=device))
    return out


class PinnedPool:
    """Pinned host buffers of one size, handed out per call and returned after the read: any number of tiles may be in flight
    (double-buffered volume drivers) without one overwriting another's mirror.  Pinning is a slow driver call, hence the pool."""

    def __init__(self):
        self.free = {}
        self.lock = threading.Lock()

    def take(self, nbytes):
        with self.lock:
            lst = self.free.setdefault(int(nbytes), [])
            if lst:
                return lst.pop()
        fresh = [torch.empty((int(nbytes),), dtype=torch.uint8).pin_memory() for _ in range(2)]
        with self.lock:
            self.free[int(nbytes)].append(fresh[1])
        return fresh[0]

    def give(self, buf):
        with self.lock:
            self.free.setdefault(int(buf.numel()), []).append(buf)


_peak_pool = PinnedPool()


def prm_select_peaks(dets, keep_idx, count, peak_threshold, A, fmap_shape, cap=None, prob=None):
    """Device-side peak selection (peak_response_mapping_3d.py:124-139,161-163): dets [rows,7] (class 1's kept detections), keep_idx
    int64 [rows], count int32 [1] (all on the device) -> dict(num int32 [1], peaks int32 [cap,4] = (a,s,h,w), dets f32 [cap,7]) on the
    device plus `host`: a pinned mirror the SAME kernel writes (host["num"] int32 [1], host["peaks"], host["dets"] as NumPy views),
    valid once the returned `event` has completed; call `release()` after reading it.  No host synchronisation here.
    prob ([A,S,H,W], the class response map): the mirror also carries host["dead"] int32 [cap] - 1 where the peak's sigmoid derivative is
    exactly 0 (a saturated score): its back-propagated map is all zero and the engine skips it."""
    _need_gpu(dets, keep_idx, count)
    assert dets.dtype == torch.float32 and dets.is_contiguous() and dets.dim() == 2 and dets.shape[1] == 7
    assert keep_idx.dtype == torch.int64 and keep_idx.is_contiguous() and count.dtype == torch.int32
    rows = int(dets.shape[0])
    cap = int(cap or rows)
    S, H, W = (int(v) for v in fmap_shape)
    dev = dets.device
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    peaks = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    out = torch.empty((cap, 7), dtype=torch.float32, device=dev)
    capb = max(cap, fused_max_boxes())                      # ONE buffer size for every tile (a new size would pin new memory)
    buf = _peak_pool.take(64 + capb * 48)
    base = buf.data_ptr()
    if prob is not None:
        _need_gpu(prob)
        assert prob.dtype == torch.float32 and prob.is_contiguous() and tuple(prob.shape[-4:]) == (int(A), S, H, W)
    check(lib().m3d_prm_select_peaks_ex(_ptr(dets), _ptr(keep_idx), _ptr(count), rows, C.c_float(np.float32(peak_threshold)), int(A), S, H, W,
                                        cap, _ptr(num), _ptr(peaks), _ptr(out), C.c_void_p(base), C.c_void_p(base + 64),
                                        C.c_void_p(base + 64 + capb * 16), _ptr(prob), None,
                                        C.c_void_p(base + 64 + capb * 44) if prob is not None else None, _stream()), "prm_select_peaks")
    ev = torch.cuda.Event()
    ev.record()
    hb = buf.numpy()
    host = dict(num=hb[:4].view(np.int32), peaks=hb[64:64 + cap * 16].view(np.int32).reshape(cap, 4),
                dets=hb[64 + capb * 16:64 + capb * 16 + cap * 28].view(np.float32).reshape(cap, 7),
                dead=hb[64 + capb * 44:64 + capb * 44 + cap * 4].view(np.int32) if prob is not None else None)
    return dict(num=num, peaks=peaks, dets=out, host=host, event=ev, release=lambda: _peak_pool.give(buf))


def strip_geometry(n, mode, P):
    """(pitch, lead, L) of the strip layout of P windows n voxels wide (include/m3d.h: m3d_prm_strip_geometry); mode 1 / True: pitch
    n + 1 (exactly-local F(2x2,3x3)), mode 2: quad-aligned for F(2x4,3x3)."""
    pitch, lead = C.c_int32(0), C.c_int32(0)
    L = lib().m3d_prm_strip_geometry(int(n), int(mode), int(P), C.byref(pitch), C.byref(lead))
    return pitch.value, lead.value, int(L)


def prm_prepare(gup, origin_up, pool, border, argmax, xnext, scale, norm, in_strip=False, out_strip=False, up_off=None, dims=None,
                in_slab=False, out_slab=False, peak_max=False):
    """gup [P,C,U,U,U]; xnext [C,UD,UH,UW]; norm [C,D,H,W] -> (window [P,C,Wn,Wn,Wn], origin int32 [P,3]).
    Strip layout (in_strip / out_strip): the P windows side by side along x with one separator column after each,
    [C,n,n,P*(n+1)] - what the Winograd kernel convolves as one wide volume; dims = (P, C, U) is then required.
    up_off (1-element tensor): gup is the bare backward-data of the layer above and its PreHook multiply by (xnext - up_off)
    (peak_backprop_3d.py:16-18) is applied here.
    in_slab / out_slab (strips only): the strip stores the planes of the layer's map - [C, xnext depth, U, L] / [C, norm depth, Wn, L] -
    instead of each window's (m3d.h: depth-clipped strips)."""
    _need_gpu(gup, origin_up, xnext, norm)
    P, Cc, U = dims if dims is not None else (gup.shape[0], gup.shape[1], gup.shape[2])
    in_strip, out_strip = int(in_strip), int(out_strip)                       # 0 batch-major, 1 strip (pitch n + 1), 2 quad-aligned strip
    in_slab, out_slab = bool(in_slab) and bool(in_strip), bool(out_slab) and bool(out_strip)
    assert tuple(gup.shape) == ((Cc, xnext.shape[1] if in_slab else U, U, strip_geometry(U, in_strip, P)[2]) if in_strip else (P, Cc, U, U, U))
    Wn = (2 if pool else 1) * U + 2 * border
    out = torch.empty((Cc, norm.shape[1] if out_slab else Wn, Wn, strip_geometry(Wn, out_strip, P)[2]) if out_strip else (P, Cc, Wn, Wn, Wn),
                      dtype=torch.float32, device=gup.device)
    oo = torch.empty((P, 3), dtype=torch.int32, device=gup.device)
    # peak_max: the launch also leaves the largest |value| of every peak's window on the output tensor (`_m3d_peak_max` [P]): the per-window
    # operand bounds of ZwConv3d.strip
    pm = torch.empty((P, 32), dtype=torch.float32, device=gup.device) if peak_max else None      # (one cache line per peak, value in column 0)
    check(lib().m3d_prm_prepare_ex3(_ptr(gup), _ptr(origin_up), P, Cc, U, int(bool(pool)), int(border), _ptr(argmax), _ptr(xnext),
                                    xnext.shape[1], xnext.shape[2], xnext.shape[3], _ptr(scale), _ptr(norm), norm.shape[1],
                                    norm.shape[2], norm.shape[3], in_strip, int(in_slab), out_strip, int(out_slab), _ptr(up_off), _ptr(out),
                                    _ptr(oo), _ptr(pm) if pm is not None else None, _stream()), "prm_prepare")
    if pm is not None:
        out._m3d_peak_max = pm
    return out, oo


def prm_stem_prepare_weights(weight):
    """weight [C,1,5,5,5] -> [C,125] flipped relu(W) for prm_stem_dgrad."""
    _need_gpu(weight)
    weight = _f32c(weight)
    Cc = weight.shape[0]
    wf = torch.empty((Cc, 125), dtype=torch.float32, device=weight.device)
    check(lib().m3d_prm_stem_prepare_weights(_ptr(weight), Cc, _ptr(wf), _stream()), "prm_stem_prepare_weights")
    return wf


def prm_stem_dgrad(gn, weight, data, data_off, origins):
    """gn [P,C,Wn,Wn,Wn]; weight = prm_stem_prepare_weights(conv1a.weight); data [D,H,W] ->
    (windows [P,Wn,Wn,Wn] clamped, sums [P])."""
    _need_gpu(gn, weight, data, data_off, origins)
    assert weight.dim() == 2 and weight.shape[1] == 125
    P, Cc, Wn = gn.shape[0], gn.shape[1], gn.shape[2]
    out = torch.empty((P, Wn, Wn, Wn), dtype=torch.float32, device=gn.device)
    sums = torch.empty((P,), dtype=torch.float32, device=gn.device)
    check(lib().m3d_prm_stem_dgrad(_ptr(gn), _ptr(weight), _ptr(data), _ptr(data_off), _ptr(origins), P, Cc, Wn, data.shape[0],
                                   data.shape[1], data.shape[2], _ptr(out), _ptr(sums), _stream()), "prm_stem_dgrad")
    return out, sums


def prm_den_pool(argmax, xnext, norm):
    """Peak-independent denominator map of a conv + MaxPool3d(2,2) layer: argmax uint8 / xnext [C,UD,UH,UW] (pool argmax and
    pooled activation), norm [C,D,H,W] (norm conv) -> den [C,UD,UH,UW] = |N| + 1e-10 at the argmax child where xnext > 0 and
    N >= 1e-10, else 0 (peak_backprop_3d.py:30-33 + ReLU / max-unpool routing)."""
    _need_gpu(argmax, xnext, norm)
    assert argmax.dtype == torch.uint8 and argmax.shape == xnext.shape
    Cc, UD, UH, UW = xnext.shape
    den = torch.empty_like(xnext)
    check(lib().m3d_prm_den_pool(_ptr(argmax.contiguous()), _ptr(_f32c(xnext)), _ptr(_f32c(norm)), Cc, UD, UH, UW, norm.shape[1],
                                 norm.shape[2], norm.shape[3], _ptr(den), _stream()), "prm_den_pool")
    return den


def prm_stem_mfma_weights(weight):
    """conv1a.weight [32,1,5,5,5] -> MFMA A-operand pack [80,64] of the tap-flipped relu(W) for prm_stem_dgrad_fused."""
    _need_gpu(weight)
    weight = _f32c(weight)
    assert tuple(weight.shape[1:]) == (1, 5, 5, 5)
    wa = torch.empty((80, 64), dtype=torch.float32, device=weight.device)
    check(lib().m3d_prm_stem_mfma_prepare_weights(_ptr(weight), weight.shape[0], _ptr(wa), _stream()), "prm_stem_mfma_prepare_weights")
    return wa


def prm_stem_dgrad_fused_supported(channels, up_size):
    return bool(lib().m3d_prm_stem_dgrad_fused_supported(int(channels), int(up_size)))


def prm_stem_dgrad_fused(gup, origin_up, den, argmax, scale, wa, data, data_off, strip=False, xnext=None, up_off=None, dims=None, slab=False):
    """Max-unpool + ReLU + BN + PostHook + backward-data of conv1a + PreHook in one MFMA kernel.  gup [P,32,U,U,U] (gradient
    w.r.t. the pooled stem output; strip=True: [32,U,U,P*(U+1)], dims = (P, 32, U)), origin_up int32 [P,3] (pooled
    coordinates), den = prm_den_pool(...), argmax uint8 [32,UD,UH,UW], scale [32] or None, wa = prm_stem_mfma_weights(W),
    data [D,H,W]; xnext + up_off: gup is the bare backward-data of conv2a, multiplied here by (xnext - up_off) ->
    (windows [P,Wn,Wn,Wn] clamped (Wn = 2U + 4), sums [P], origins int32 [P,3])."""
    _need_gpu(gup, origin_up, den, argmax, wa, data, data_off)
    P, Cc, U = dims if dims is not None else (gup.shape[0], gup.shape[1], gup.shape[2])
    strip = int(strip)
    slab = bool(slab) and bool(strip)                                            # gup stores the pooled layer's planes (den.shape[1] of them)
    assert tuple(gup.shape) == ((Cc, den.shape[1] if slab else U, U, strip_geometry(U, strip, P)[2]) if strip else (P, Cc, U, U, U)) and gup.is_contiguous()
    assert (xnext is None) == (up_off is None)
    Wn = 2 * U + 4
    out = torch.empty((P, Wn, Wn, Wn), dtype=torch.float32, device=gup.device)
    sums = torch.empty((P,), dtype=torch.float32, device=gup.device)
    oo = torch.empty((P, 3), dtype=torch.int32, device=gup.device)
    check(lib().m3d_prm_stem_dgrad_fused_ex2(_ptr(gup), strip, int(slab), _ptr(xnext), _ptr(up_off), _ptr(origin_up), P, Cc, U,
                                            _ptr(den), _ptr(argmax), _ptr(scale), den.shape[1], den.shape[2], den.shape[3],
                                            _ptr(wa), _ptr(data), _ptr(data_off), data.shape[0], data.shape[1], data.shape[2],
                                            _ptr(out), _ptr(sums), _ptr(oo), _stream()), "prm_stem_dgrad_fused")
    return out, sums, oo


def conv3d_stem5_dgrad_weights(weight):
    """weight [C,1,5,5,5] -> [C,125] tap-flipped (no ReLU) for conv3d_stem5_dgrad."""
    _need_gpu(weight)
    weight = _f32c(weight)
    assert tuple(weight.shape[1:]) == (1, 5, 5, 5)
    wf = torch.empty((weight.shape[0], 125), dtype=torch.float32, device=weight.device)
    check(lib().m3d_conv3d_stem5_prepare_dgrad_weights(_ptr(weight), weight.shape[0], _ptr(wf), _stream()), "stem5_prepare_dgrad_weights")
    return wf


def conv3d_stem5_dgrad(grad_out, wf):
    """Backward-data of the 5^3 / Cin=1 stem conv: grad_out [B,C,D,H,W], wf = conv3d_stem5_dgrad_weights(W) -> [B,1,D,H,W]."""
    _need_gpu(grad_out, wf)
    grad_out = _f32c(grad_out)
    B, Cc, D, H, W = grad_out.shape
    assert wf.shape == (Cc, 125)
    gin = torch.empty((B, 1, D, H, W), dtype=torch.float32, device=grad_out.device)
    check(lib().m3d_conv3d_stem5_dgrad(_ptr(grad_out), _ptr(wf), _ptr(gin), B, Cc, D, H, W, _stream()), "conv3d_stem5_dgrad")
    return gin


def prm_scatter(windows, sums, origins, shape):
    _need_gpu(windows, sums, origins)
    P, Wn = windows.shape[0], windows.shape[1]
    D, H, W = shape
    dense = torch.empty((P, D, H, W), dtype=torch.float32, device=windows.device)     # the library writes every voxel
    check(lib().m3d_prm_scatter(_ptr(windows), _ptr(sums), _ptr(origins), P, Wn, D, H, W, _ptr(dense), _stream()), "prm_scatter")
    return dense


# ------------------------------------------------------------------ PRM post-processing -> Otsu
def prm_quantize_u8(prms):
    """prms float32 [P, D, H, W] CUDA -> uint8 [P, D, H, W] (infer_simple.py:233-238 per map)."""
    _need_gpu(prms)
    prms = _f32c(prms)
    out = torch.empty(prms.shape, dtype=torch.uint8, device=prms.device)
    P = prms.shape[0]
    check(lib().m3d_prm_quantize_u8(_ptr(prms), P, C.c_int64(prms[0].numel() if P else 1), _ptr(out), _stream()), "prm_quantize_u8")
    return out


def prm_quantize_windows_u8(windows, sums, origins, shape, return_nonempty=False):
    """The uint8 maps of prm_quantize_u8(prm_scatter(windows, sums, origins, shape)) without the dense float maps: windows
    float32 [P,Wn,Wn,Wn] (un-normalised, clamped), sums [P], origins int32 [P,3], shape = (D,H,W) -> uint8 [P,D,H,W]
    (+ bool [P]: the map has a non-zero voxel)."""
    _need_gpu(windows, sums, origins)
    windows = _f32c(windows)
    P, Wn = windows.shape[0], windows.shape[1]
    D, H, W = (int(v) for v in shape)
    out = torch.empty((P, D, H, W), dtype=torch.uint8, device=windows.device)
    ws = torch.empty((max(16 * P, 16),), dtype=torch.uint8, device=windows.device)
    check(lib().m3d_prm_quantize_windows_u8(_ptr(windows), _ptr(_f32c(sums)), _ptr(origins.contiguous()), P, Wn, D, H, W, _ptr(out),
                                            _ptr(ws), C.c_size_t(ws.numel()), _stream()), "prm_quantize_windows_u8")
    if return_nonempty:
        return out, ws[:16 * P].view(torch.int32).view(P, 4)[:, 3] != 0
    return out


def prm_quantize_windows_compact_u8(windows, sums, origins, shape, out=None, return_nonempty=False):
    """The uint8 values of prm_quantize_windows_u8 as windows [P,Wn,Wn,Wn] (0 outside the tile): with `origins`, everything a writer
    needs to rebuild each uint8 map (m3d.io.encode_window_stack_u8)."""
    _need_gpu(windows, sums, origins)
    windows = _f32c(windows)
    P, Wn = windows.shape[0], windows.shape[1]
    D, H, W = (int(v) for v in shape)
    if out is None:
        out = torch.empty((P, Wn, Wn, Wn), dtype=torch.uint8, device=windows.device)
    ws = torch.empty((max(16 * P, 16),), dtype=torch.uint8, device=windows.device)
    check(lib().m3d_prm_quantize_windows_compact_u8(_ptr(windows), _ptr(_f32c(sums)), _ptr(origins.contiguous()), P, Wn, D, H, W, _ptr(out),
                                                    _ptr(ws), C.c_size_t(ws.numel()), _stream()), "prm_quantize_windows_compact_u8")
    if return_nonempty == "stats":                   # the raw per-map words (word 3 != 0: the map has a non-zero voxel), for paint_begin
        return out, ws[:16 * P].view(torch.int32).view(P, 4)
    if return_nonempty:
        return out, ws[:16 * P].view(torch.int32).view(P, 4)[:, 3] != 0
    return out


def roi_normalize(image_u16, prm_u8, boxes, mode, boxes_host=None, map_index=None, win_origins=None, offsets=None):
    """image_u16 [D,H,W] uint16 CUDA; prm_u8 [R,D,H,W] uint8; boxes int32 [R,6] inclusive (x1,y1,z1,x2,y2,z2).
    Returns (img crops uint16 flat, prm crops uint16 flat, offsets int64 [R+1]) - the inputs of otsu2d_batch.
    boxes_host: the same boxes as an ndarray when the caller has them (saves the device read-back that sizes the crops).
    map_index (int32 [R] on the device): RoI r reads prm_u8[map_index[r]] - prm_u8 then holds ALL maps of the tile, not a gathered subset.
    win_origins (int32 [P,3]): prm_u8 is the COMPACT form [P,n,n,n] of prm_quantize_windows_compact_u8 (zero outside each window)."""
    _need_gpu(image_u16, prm_u8, boxes)
    assert image_u16.dtype == torch.uint16 and prm_u8.dtype == torch.uint8 and boxes.dtype == torch.int32
    R = boxes.shape[0]
    D, H, W = image_u16.shape
    win = 0 if win_origins is None else int(prm_u8.shape[1])
    if win_origins is not None:
        assert prm_u8.dim() == 4 and prm_u8.shape[1] == prm_u8.shape[2] == prm_u8.shape[3] and win_origins.dtype == torch.int32 and win_origins.is_contiguous()
    b = (np.asarray(boxes_host) if boxes_host is not None else boxes.cpu().numpy()).astype(np.int64)
    sizes = (b[:, 3] - b[:, 0] + 1) * (b[:, 4] - b[:, 1] + 1) * (b[:, 5] - b[:, 2] + 1)
    assert R == 0 or (sizes.min() > 0 and b[:, :3].min() >= 0 and b[:, 3].max() < W and b[:, 4].max() < H and b[:, 5].max() < D)
    offs_h = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    offs = upload(offs_h, image_u16.device) if offsets is None else offsets     # offsets: the caller uploaded crop_offsets(boxes_host) itself
    total = int(offs_h[-1]) if R else 0
    oi = torch.empty((total,), dtype=torch.uint16, device=image_u16.device)
    op = torch.empty((total,), dtype=torch.uint16, device=image_u16.device)
    if R == 0:
        return oi, op, offs
    ws = torch.empty((24 * R,), dtype=torch.uint8, device=image_u16.device)
    if map_index is not None:
        assert map_index.dtype == torch.int32 and map_index.is_cuda and map_index.numel() == R
    check(lib().m3d_roi_normalize_idx(_ptr(image_u16.contiguous()), _ptr(prm_u8.contiguous()), _ptr(map_index), win, _ptr(win_origins), _ptr(boxes.contiguous()), _ptr(offs),
                                     R, C.c_int64(total), D, H, W, {"soma": 0, "nuclei": 1}[mode], _ptr(oi), _ptr(op), _ptr(ws),
                                     C.c_size_t(ws.numel()), _stream()), "roi_normalize")
    return oi, op, offs


# ------------------------------------------------------------------ Winograd-x 3x3x3 forward
class ZwConv3d(object):
    """3x3x3 forward conv (stride 1, pad 1) + scale/shift + ReLU [+ MaxPool3d(2,2)] on the f16 matrix cores at fp32 accuracy
    (csrc/conv3d_zw.hip: f16x2 split, Winograd F(2,3) along z).  The operand scale comes from a BOUND of the input's largest magnitude
    that travels with the activations: `in_max` is a device array of SLOTS floats (its largest entry is the bound; `bound_of(x)` sweeps a
    tensor), and every call returns the same kind of array for its OUTPUT (filled by the kernel's epilogue), so a chain of layers sweeps
    only its first input.  __call__ -> (out, out_max)."""
    SLOTS = 32

    @staticmethod
    def supported(weight, shape=None, pool=False):
        """weight [cout, cin, 3, 3, 3] with cin % 16 == 0; shape = (D, H, W) of the input (None: any supported map)."""
        if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or int(weight.shape[1]) % 16 != 0:
            return False
        if shape is None:
            return True
        D, H, W = (int(v) for v in shape)
        return bool(lib().m3d_conv3d_zw_supported(int(weight.shape[1]), int(weight.shape[0]), D, H, W, int(bool(pool))))

    def __init__(self, weight):
        _need_gpu(weight)
        w = _f32c(weight)
        assert ZwConv3d.supported(w)
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        nbytes = lib().m3d_conv3d_zw_packed_bytes(self.cin, self.cout)
        self.packed = torch.empty((nbytes,), dtype=torch.uint8, device=w.device)
        check(lib().m3d_conv3d_zw_pack(_ptr(w), self.cin, self.cout, _ptr(self.packed), _stream()), "conv3d_zw_pack")

    def supports(self, shape, pool=False):
        D, H, W = (int(v) for v in shape[-3:])
        return bool(lib().m3d_conv3d_zw_supported(self.cin, self.cout, D, H, W, int(bool(pool))))

    def strip(self, gn, pitch, P, out=None, bounds=None):
        """PRM window strip gn [cin, planes, U, L] (windows side by side along x, cell p = columns [pitch p, pitch (p + 1)): strip_geometry
        mode 2) -> [cout, planes, U, L], one operand scale per window (bounds [P, 32], column 0 = the largest |value| of each window, e.g. what
        prm_prepare(peak_max=True) left on gn; None: swept here, m3d_prm_strip_absmax).  Returns None when the library has no
        configuration for the shape."""
        _need_gpu(gn)
        assert gn.dim() == 4 and gn.shape[0] == self.cin and gn.is_contiguous() and gn.dtype == torch.float32
        cin, D, H, L = (int(v) for v in gn.shape)
        if L < 24 or not self.supports((D, H, L)) or gn.numel() * 4 >= 0x7FFFFF00 or pitch % 4 or L % 4:
            return None
        bounds = self._strip_bounds(gn, pitch, P, bounds)
        if out is None:
            out = torch.empty((self.cout, D, H, L), dtype=torch.float32, device=gn.device)
        check(lib().m3d_conv3d_zw_forward_strip(_ptr(gn), _ptr(self.packed), _ptr(out), cin, self.cout, D, H, L, _ptr(bounds), int(P), int(pitch),
                                                _stream()), "conv3d_zw_forward_strip")
        return out

    def _strip_bounds(self, gn, pitch, P, bounds):
        if bounds is None:                                  # (the producer did not leave them: one sweep of the strip)
            bounds = torch.empty((P, 32), dtype=torch.float32, device=gn.device)
            check(lib().m3d_prm_strip_absmax(_ptr(gn), C.c_longlong(gn.shape[0] * gn.shape[1] * gn.shape[2]), int(gn.shape[3]), int(pitch), int(P),
                                             _ptr(bounds), _stream()), "prm_strip_absmax")
        assert tuple(bounds.shape) == (P, 32) and bounds.dtype == torch.float32 and bounds.is_contiguous()
        return bounds

    def strip_prepare(self, gn, dims, origin, xnext, scale, norm, up_off, in_slab=False, out_slab=False, bounds=None):
        """WinoConv3d.strip_prepare on this kernel (m3d_prm_strip_dgrad_prepare_zw): the strip conv fused with the prepare step of the layer
        below; (strip [cout, planes', U + 2, L(U + 2)], origin - 1), or None where the library has no configuration."""
        _need_gpu(gn, origin, xnext, norm, up_off)
        P, cin, U = (int(v) for v in dims)
        assert cin == self.cin and xnext.shape[0] == self.cout == norm.shape[0] and xnext.shape == norm.shape
        MD, MH, MW = (int(v) for v in norm.shape[1:])
        in_slab, out_slab = bool(in_slab), bool(out_slab)
        pitch, _, L = strip_geometry(U, 2, P)
        assert tuple(gn.shape) == (cin, MD if in_slab else U, U, L) and gn.is_contiguous()
        if L < 24 or pitch % 4 or gn.numel() * 4 >= 0x7FFFFF00 or not self.supports((gn.shape[1], U, L)):
            return None
        bounds = self._strip_bounds(gn, pitch, P, bounds)
        out = torch.empty((self.cout, MD if out_slab else U + 2, U + 2, strip_geometry(U + 2, 2, P)[2]), dtype=torch.float32, device=gn.device)
        oo = torch.empty((P, 3), dtype=torch.int32, device=gn.device)
        rc = lib().m3d_prm_strip_dgrad_prepare_zw(_ptr(gn), _ptr(self.packed), cin, self.cout, P, U, int(in_slab), _ptr(origin), _ptr(xnext),
                                                  _ptr(norm), _ptr(scale), _ptr(up_off), MD, MH, MW, int(out_slab), _ptr(bounds), _ptr(out),
                                                  _ptr(oo), _stream())
        if rc == -4:                                        # M3D_EUNSUPPORTED
            return None
        check(rc, "prm_strip_dgrad_prepare_zw")
        return out, oo

    def units(self, shape):
        """workgroups of a launch on an input [B, cin, D, H, W] (or (D, H, W): one item): (64 channels) x (32 x 4 x 2 or 16 x 8 x 2 voxels)"""
        B = int(shape[0]) if len(shape) == 5 else 1
        D, H, W = (int(v) for v in shape[-3:])
        xb, ty = (32, 4) if W >= 24 else (16, 8)
        return B * ((self.cout + 63) // 64) * ((W + xb - 1) // xb) * ((H + ty - 1) // ty) * ((D + 1) // 2)

    @staticmethod
    def bound_of(x):
        """[SLOTS] device floats, slot 0 = max |x| (one sweep of x)."""
        _need_gpu(x)
        x = _f32c(x)
        out = torch.empty((ZwConv3d.SLOTS,), dtype=torch.float32, device=x.device)
        check(lib().m3d_conv3d_zw_bound_of(_ptr(x), C.c_longlong(x.numel()), _ptr(out), _stream()), "conv3d_zw_bound_of")
        return out

    def __call__(self, x, in_max, scale=None, shift=None, relu=False, pool=False, out=None, out_max=None):
        """out_max: a ZEROED [SLOTS] float tensor to receive the output's bound (None: a fresh one; False: no bound output)."""
        _need_gpu(x, in_max)
        x = _f32c(x)
        B, cin, D, H, W = x.shape
        assert cin == self.cin and in_max.numel() == self.SLOTS and in_max.dtype == torch.float32
        assert x[0].numel() * 4 < 0x7FFFFFFF                  # one batch item is addressed with 32-bit buffer offsets
        oshape = (B, self.cout, D // 2, H // 2, W // 2) if pool else (B, self.cout, D, H, W)
        if out is None:
            out = torch.empty(oshape, dtype=torch.float32, device=x.device)
        assert tuple(out.shape) == oshape and out.is_contiguous()
        if out_max is None:
            out_max = torch.zeros((self.SLOTS,), dtype=torch.float32, device=x.device)
        elif out_max is False:                             # nobody needs the output's bound (the last layer of a chain)
            out_max = None
        check(lib().m3d_conv3d_zw_forward(_ptr(x), _ptr(self.packed), _ptr(out), B, cin, self.cout, D, H, W,
                                          _ptr(scale) if scale is not None else None, _ptr(shift) if shift is not None else None,
                                          int(bool(relu)), int(bool(pool)), _ptr(in_max), _ptr(out_max) if out_max is not None else None,
                                          _stream()), "conv3d_zw_forward")
        return out, out_max


class WinoConv3d(object):
    """3x3x3 forward conv (stride 1, pad 1) through a Winograd MFMA kernel; weights transformed and packed once.
    two_d=False: F(2,3) along x (2/3 of the MFMA work); two_d=True: F(2x2,3x3) on the (y,x) plane (4/9).
    `supports(width)` says whether the kernel has a tile configuration for the map (else use PackedConv3d)."""

    def __init__(self, weight, two_d=False, local=False):
        """local=True (2-D kernels): only the F(2x2,3x3) family, whose outputs depend on nothing outside their own 3 x 3 support even by
        rounding - for data laid out with unrelated neighbours (the PRM strips); the default family uses F(4,3) along x."""
        _need_gpu(weight)
        w = _f32c(weight)
        assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3)
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        self.two_d = bool(two_d)
        self.local = bool(local) and self.two_d
        L = lib()
        self._bytes, self._pack, self._fwd, self._pool = \
            (L.m3d_conv3d_wino2_packed_weight_bytes, L.m3d_conv3d_wino2_pack_weights, L.m3d_conv3d_wino2_forward,
             L.m3d_conv3d_wino2_forward_pool2) if self.two_d else \
            (L.m3d_conv3d_wino_packed_weight_bytes, L.m3d_conv3d_wino_pack_weights, L.m3d_conv3d_wino_forward,
             L.m3d_conv3d_wino_forward_pool2)
        nbytes = self._bytes(self.cin, self.cout)
        self.packed = torch.empty((nbytes // 4,), dtype=torch.float32, device=w.device)
        check(self._pack(_ptr(w), self.cin, self.cout, _ptr(self.packed), _stream()), "wino_pack")

    def supports(self, width, shape=None):
        """maps >= 24 voxels wide; the 2-D kernel also has a 16-wide tile with split-K from 12.  With shape = (B, D, H, W)
        the 2-D kernel additionally asks the library whether its tiles fit the map well enough to beat the direct kernel."""
        if width < (12 if self.two_d else 24):
            return False
        if self.two_d and shape is not None:
            B, D, H, W = (int(v) for v in shape)
            # below ~0.3 of useful tile volume x chip fill the direct kernel wins (tools/bench_wino_threshold.py; with F(2x4,3x3) the
            # 0.39-score layers of the soma net run 0.194 vs 0.253 ms, so the round-2 threshold of 0.5 came down)
            score = lib().m3d_conv3d_wino2_local_score if self.local else lib().m3d_conv3d_wino2_score     # each family asks about its own tiles
            return score(B, self.cin, self.cout, D, H, W) >= 0.3
        return True

    def __call__(self, x, scale=None, shift=None, relu=False, out=None):
        _need_gpu(x)
        x = _f32c(x)
        B, cin, D, H, W = x.shape
        assert cin == self.cin
        if out is None:
            out = torch.empty((B, self.cout, D, H, W), dtype=torch.float32, device=x.device)
        if self.two_d:                                     # the library picks the tile; small / ragged maps may split K
            wsb_fn, fwd_fn = (lib().m3d_conv3d_wino2_local_workspace_bytes, lib().m3d_conv3d_wino2_local_forward_ws) if self.local else \
                             (lib().m3d_conv3d_wino2_workspace_bytes, lib().m3d_conv3d_wino2_forward_ws)
            wsb = wsb_fn(B, cin, self.cout, D, H, W)
            key = (torch.cuda.current_stream().cuda_stream, x.device)           # one scratch buffer per stream: tiles on
            cache = self.__dict__.setdefault("_ws", {})                           # different streams run concurrently
            ws = cache.get(key)
            if ws is None or ws.numel() < wsb:
                ws = cache[key] = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=x.device)
            check(fwd_fn(_ptr(x), _ptr(self.packed), _ptr(out), B, cin, self.cout, D, H, W,
                         _ptr(scale) if scale is not None else None,
                         _ptr(shift) if shift is not None else None, int(bool(relu)),
                         _ptr(ws), C.c_size_t(wsb), _stream()), "conv3d_wino2_forward_ws")
            return out
        check(self._fwd(_ptr(x), _ptr(self.packed), _ptr(out), B, cin, self.cout, D, H, W,
                        _ptr(scale) if scale is not None else None, _ptr(shift) if shift is not None else None,
                        int(bool(relu)), _stream()), "conv3d_wino_forward")
        return out

    def plan(self, shape):
        """(family, tile id, K split) the library would use for an input [B, cin, D, H, W] (m3d_conv3d_wino2_plan; 2-D kernels only)."""
        B, D, H, W = (int(shape[0]),) + tuple(int(v) for v in shape[-3:])
        f, t, k = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().m3d_conv3d_wino2_plan(int(self.local), B, self.cin, self.cout, D, H, W, C.byref(f), C.byref(t), C.byref(k)), "conv3d_wino2_plan")
        return f.value, t.value, k.value

    def strip_prepare(self, gn, dims, origin, xnext, scale, norm, up_off, in_slab=False, out_slab=False):
        """PRM strips: this conv's backward-data on the prepared strip `gn` [cin, planes, U, L(U)] (quad-aligned layout, dims = (P, cin, U))
        FUSED with the prepare step of the layer below (xnext / scale / norm / up_off as ops.prm_prepare with pool = False, border = 1):
        returns (strip [cout, planes', U + 2, L(U + 2)], origin - 1), exactly what prm_prepare(out_strip=2) would make of this conv's
        result - or None when the library runs this shape through another kernel (the caller then takes the two launches)."""
        _need_gpu(gn, origin, xnext, norm, up_off)
        assert self.two_d and not self.local
        P, cin, U = (int(v) for v in dims)
        assert cin == self.cin and xnext.shape[0] == self.cout == norm.shape[0] and xnext.shape == norm.shape
        MD, MH, MW = (int(v) for v in norm.shape[1:])
        in_slab, out_slab = bool(in_slab), bool(out_slab)
        assert tuple(gn.shape) == (cin, MD if in_slab else U, U, strip_geometry(U, 2, P)[2]) and gn.is_contiguous()
        out = torch.empty((self.cout, MD if out_slab else U + 2, U + 2, strip_geometry(U + 2, 2, P)[2]), dtype=torch.float32, device=gn.device)
        oo = torch.empty((P, 3), dtype=torch.int32, device=gn.device)
        rc = lib().m3d_prm_strip_dgrad_prepare(_ptr(gn), _ptr(self.packed), cin, self.cout, P, U, int(in_slab), _ptr(origin), _ptr(xnext), _ptr(norm),
                                               _ptr(scale), _ptr(up_off), MD, MH, MW, int(out_slab), _ptr(out), _ptr(oo), _stream())
        if rc == -4:                                        # M3D_EUNSUPPORTED
            return None
        check(rc, "prm_strip_dgrad_prepare")
        return out, oo

    def supports_pool(self, width):
        return width >= (24 if self.two_d else 48)

    def pooled(self, x, scale=None, shift=None, relu=False, out=None, return_argmax=False):
        """conv + scale/shift + ReLU + MaxPool3d(2,2) in one launch; returns [B,cout,D//2,H//2,W//2] (+ the pool's uint8 argmax,
        2-D kernel only)."""
        _need_gpu(x)
        x = _f32c(x)
        B, cin, D, H, W = x.shape
        assert cin == self.cin
        if out is None:
            out = torch.empty((B, self.cout, D // 2, H // 2, W // 2), dtype=torch.float32, device=x.device)
        if return_argmax:
            assert self.two_d
            am = torch.empty(out.shape, dtype=torch.uint8, device=x.device)
            check(lib().m3d_conv3d_wino2_forward_pool2_argmax(_ptr(x), _ptr(self.packed), _ptr(out), _ptr(am), B, cin, self.cout, D, H, W,
                                                              _ptr(scale) if scale is not None else None,
                                                              _ptr(shift) if shift is not None else None, int(bool(relu)), _stream()),
                  "conv3d_wino2_forward_pool2_argmax")
            return out, am
        check(self._pool(_ptr(x), _ptr(self.packed), _ptr(out), B, cin, self.cout, D, H, W,
                         _ptr(scale) if scale is not None else None,
                         _ptr(shift) if shift is not None else None, int(bool(relu)), _stream()),
              "conv3d_wino_forward_pool2")
        return out


class StemWinoConv3d(object):
    """conv1a (5x5x5, Cin = 1, stride 1, pad 2) through the Winograd F(2,5)-along-x MFMA kernel, optionally fused with
    MaxPool3d(2,2); maps >= 32 voxels wide (else use PackedConv3d)."""

    def __init__(self, weight):
        _need_gpu(weight)
        w = _f32c(weight)
        assert w.dim() == 5 and tuple(w.shape[1:]) == (1, 5, 5, 5)
        self.cout, self.cin = int(w.shape[0]), 1
        nbytes = lib().m3d_conv3d_stem_wino_packed_weight_bytes(self.cout)
        self.packed = torch.empty((nbytes // 4,), dtype=torch.float32, device=w.device)
        check(lib().m3d_conv3d_stem_wino_pack_weights(_ptr(w), self.cout, _ptr(self.packed), _stream()), "stem_wino_pack")

    @staticmethod
    def supports(width):
        return width >= 32

    def _run(self, x, scale, shift, relu, pool, out, bound=False):
        _need_gpu(x)
        x = _f32c(x)
        B, cin, D, H, W = x.shape
        assert cin == 1
        if out is None:
            shp = (B, self.cout, D // 2, H // 2, W // 2) if pool else (B, self.cout, D, H, W)
            out = torch.empty(shp, dtype=torch.float32, device=x.device)
        # bound: True (a fresh zeroed slot array) or a ZEROED [SLOTS] float tensor of the caller's
        om = None if bound is False or bound is None else \
            (torch.zeros((ZwConv3d.SLOTS,), dtype=torch.float32, device=x.device) if bound is True else bound)
        bound = om is not None
        check(lib().m3d_conv3d_stem_wino_forward_bound(_ptr(x), _ptr(self.packed), _ptr(out), B, self.cout, D, H, W,
                                                       _ptr(scale) if scale is not None else None,
                                                       _ptr(shift) if shift is not None else None, int(bool(relu)), int(bool(pool)),
                                                       _ptr(om) if bound else None, _stream()), "conv3d_stem_wino_forward")
        if bound:
            out._m3d_bound = (om, out._version)            # the operand bound of the f16x2 conv that reads `out` (ZwConv3d), valid for this version of `out`
        return out

    def __call__(self, x, scale=None, shift=None, relu=False, out=None, bound=False):
        return self._run(x, scale, shift, relu, False, out, bound)

    def pooled(self, x, scale=None, shift=None, relu=False, bound=False):
        return self._run(x, scale, shift, relu, True, None, bound)


# ------------------------------------------------------------------ conv backward-weights / bias gradient
def conv3d_wgrad(x, grad_out, k):
    """dW [cout,cin,k,k,k] of a stride-1 pad-k/2 conv: x [B,cin,D,H,W], grad_out [B,cout,D,H,W] (fp32 CUDA)."""
    _need_gpu(x, grad_out)
    x, grad_out = _f32c(x), _f32c(grad_out)
    B, cin, D, H, W = x.shape
    cout = grad_out.shape[1]
    assert grad_out.shape == (B, cout, D, H, W)
    dw = torch.empty((cout, cin, k, k, k), dtype=torch.float32, device=x.device)
    wsb = lib().m3d_conv3d_wgrad_workspace_bytes(B, cin, cout, D, H, W, k)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
    check(lib().m3d_conv3d_wgrad(_ptr(x), _ptr(grad_out), _ptr(dw), B, cin, cout, D, H, W, k, _ptr(ws), C.c_size_t(wsb), _stream()),
          "conv3d_wgrad")
    return dw


def conv3d_bias_grad(grad_out):
    _need_gpu(grad_out)
    grad_out = _f32c(grad_out)
    B, cout, D, H, W = grad_out.shape
    db = torch.empty((cout,), dtype=torch.float32, device=grad_out.device)
    check(lib().m3d_conv3d_bias_grad(_ptr(grad_out), _ptr(db), B, cout, D, H, W, _stream()), "conv3d_bias_grad")
    return db


# ------------------------------------------------------------------ volume pre-filters (binarization_nuclei.py:44-45)
def gaussian_filter_u16(vol, sigma=1.0, truncate=4.0):
    """scipy.ndimage.gaussian_filter(vol, sigma) for a uint16 CUDA volume [D,H,W], bit-exact (uint16 after every pass)."""
    _need_gpu(vol)
    assert vol.dtype == torch.uint16 and vol.dim() == 3
    vol = vol.contiguous()
    radius = int(truncate * float(sigma) + 0.5)                      # scipy/ndimage/_filters.py gaussian_filter1d
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)      # _gaussian_kernel1d, order 0
    phi = phi / phi.sum()
    w = torch.from_numpy(np.ascontiguousarray(phi[radius:])).to(vol.device)      # centre first
    out, tmp = torch.empty_like(vol), torch.empty_like(vol)
    D, H, W = vol.shape
    check(lib().m3d_gaussian_filter_u16(_ptr(vol), _ptr(out), _ptr(tmp), D, H, W, _ptr(w), radius, _stream()), "gaussian_filter_u16")
    return out


def median_filter3_u16(vol):
    """scipy.ndimage.median_filter(vol, size=3) for a uint16 CUDA volume [D,H,W]."""
    _need_gpu(vol)
    assert vol.dtype == torch.uint16 and vol.dim() == 3
    vol = vol.contiguous()
    out = torch.empty_like(vol)
    D, H, W = vol.shape
    check(lib().m3d_median_filter3_u16(_ptr(vol), _ptr(out), D, H, W, _stream()), "median_filter3_u16")
    return out


# ------------------------------------------------------------------ connected components / closing / painting
def cc_largest_batch(mask, offsets, dims, invert=False, tie_last=True):
    """mask uint8 flat (crops concatenated), offsets int64 [R+1], dims int32 [R,3] (ez,ey,ex).
    Returns (out uint8 flat {0,255}, status int32 [R])."""
    _need_gpu(mask, offsets, dims)
    assert mask.dtype == torch.uint8 and offsets.dtype == torch.int64 and dims.dtype == torch.int32
    R = dims.shape[0]
    total = int(mask.numel())
    out = torch.empty_like(mask)
    status = torch.empty((R,), dtype=torch.int32, device=mask.device)        # cc_init (bad crop: 2) / cc_write (0 / 1) write every entry
    wsb = lib().m3d_cc_workspace_bytes(C.c_int64(total))
    ws = torch.empty((wsb,), dtype=torch.uint8, device=mask.device)
    check(lib().m3d_cc_largest_batch(_ptr(mask), _ptr(offsets), _ptr(dims.contiguous()), R, C.c_int64(total), int(bool(invert)),
                                     int(bool(tie_last)), _ptr(out), _ptr(status), _ptr(ws), C.c_size_t(wsb), _stream()),
          "cc_largest_batch")
    return out, status


def binary_closing6_batch(mask, offsets, dims):
    _need_gpu(mask, offsets, dims)
    R = dims.shape[0]
    total = int(mask.numel())
    out = torch.empty_like(mask)
    ws = torch.empty((total + 512,), dtype=torch.uint8, device=mask.device)
    check(lib().m3d_binary_closing6_batch(_ptr(mask), _ptr(offsets), _ptr(dims.contiguous()), R, C.c_int64(total), _ptr(out),
                                          _ptr(ws), C.c_size_t(total + 512), _stream()), "binary_closing6_batch")
    return out


def paint_instances_into(vol, mask, offsets, boxes, ids):
    """Accumulating form: vol is an int32 [D,H,W] volume pre-filled with -1 (0xFFFFFFFF); several calls (one per tile)
    may paint into it; finish with `torch.where(vol == -1, 0, vol)`."""
    _need_gpu(vol, mask, offsets, boxes, ids)
    assert vol.dtype == torch.int32 and vol.is_contiguous() and boxes.dtype == torch.int32 and ids.dtype == torch.int32
    D, H, W = vol.shape
    check(lib().m3d_paint_instances(_ptr(mask), _ptr(offsets), _ptr(boxes.contiguous()), _ptr(ids.contiguous()), boxes.shape[0],
                                    D, H, W, _ptr(vol), _stream()), "paint_instances")
    return vol


def paint_finish(vol, max_id, present=None):
    """In place: sentinel -> 0; returns bool [max_id + 1]: id occurs in the volume (index 0 unused).  No host synchronisation.
    present: a zeroed uint8 [max_id + 1] buffer (paint_begin's), else one is made here."""
    _need_gpu(vol)
    assert vol.dtype == torch.int32 and vol.is_contiguous()
    if present is None:
        present = torch.zeros((int(max_id) + 1,), dtype=torch.uint8, device=vol.device)
    assert present.numel() == int(max_id) + 1 and present.dtype == torch.uint8
    check(lib().m3d_paint_finish(_ptr(vol), C.c_int64(vol.numel()), int(max_id), _ptr(present), _stream()), "paint_finish")
    return present.view(torch.bool)


def crop_offsets(boxes_host):
    """int64 [R+1]: start of each box's crop in the concatenated crop buffers (boxes inclusive (x1,y1,z1,x2,y2,z2))."""
    b = np.asarray(boxes_host).astype(np.int64)
    sizes = (b[:, 3] - b[:, 0] + 1) * (b[:, 4] - b[:, 1] + 1) * (b[:, 5] - b[:, 2] + 1)
    return np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)


def paint_begin(shape, num_present, st_otsu, st_cc, map_stats, idx, first_id, device):
    """One launch: label volume int32 [D,H,W] at the 0xFFFFFFFF sentinel, present uint8 [num_present] at 0, ids int32 [R] (idx + first_id,
    -1 where a stage failed or the detection's map is all zero).  Returns (volume, present, ids)."""
    _need_gpu(st_otsu, st_cc, idx)
    R = int(idx.numel())
    assert idx.dtype == torch.int64 and st_otsu.dtype == torch.int32 and st_cc.dtype == torch.int32 and st_otsu.numel() == R == st_cc.numel()
    vol = torch.empty(tuple(shape), dtype=torch.int32, device=device)
    present = torch.empty((int(num_present),), dtype=torch.uint8, device=device)
    ids = torch.empty((R,), dtype=torch.int32, device=device)
    if map_stats is not None:
        assert map_stats.dtype == torch.int32 and map_stats.is_contiguous()
    check(lib().m3d_paint_begin(_ptr(vol), C.c_int64(vol.numel()), _ptr(present), int(num_present), _ptr(st_otsu), _ptr(st_cc), _ptr(map_stats),
                                _ptr(idx), R, int(first_id), _ptr(ids), _stream()), "paint_begin")
    return vol, present, ids


def paint_instances(mask, offsets, boxes, ids, shape):
    """Returns the int32 label volume [D,H,W]: id of the first (lowest-id) instance covering each voxel, 0 elsewhere.
    ids < 0 (0xFFFFFFFF as unsigned) never win, i.e. mark a skipped detection."""
    vol = torch.full(tuple(shape), -1, dtype=torch.int32, device=mask.device)      # 0xFFFFFFFF sentinel
    paint_instances_into(vol, mask, offsets, boxes, ids)
    return torch.where(vol == -1, torch.zeros_like(vol), vol)
