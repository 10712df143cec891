"""Host-side counterpart of the reference's inference graph, running on the HIP kernels.

Mirrors (reference paths): lib/modeling/DSN.py:57-68 (dsn_body), lib/modeling/rpn_heads.py:88-137
(single_scale_rpn_outputs), lib/modeling/generate_proposals_3d.py (via m3d.generate_proposals3d),
lib/modeling/model_builder.py:294-312 (roi_feature_transform, RoIAlign branch),
lib/modeling/fast_rcnn_heads.py:104-117,39-47 (roi_2mlp_head, fast_rcnn_outputs),
lib/core/test.py:194-263,806-883 (im_detect_bbox, box_results_with_nms_and_limit).
The state-dict key layout is the reference's: Conv_Body.conv1a.weight ... Box_Outs.bbox_pred.bias.

Every op, the linear layers included (split-K fp32 MFMA GEMM, csrc/fc_gemm.hip), is a libm3d.so kernel; torch supplies
device memory, streams and a few elementwise helpers (sigmoid, softmax, cat).
"""
import os

import numpy as np
import torch

from . import ops

BN_EPS = 1e-5   # nn.BatchNorm3d default, DSN.py:20


def dsn_layers(stride):
    if stride == 8:
        return [("conv1a", "bn1a", True), ("conv2a", "bn2a", False), ("conv2b", "bn2b", True), ("conv3a", "bn3a", False),
                ("conv3b", "bn3b", True), ("conv4a", "bn4a", False), ("conv4b", "bn4b", False)]
    return [("conv1a", "bn1a", True), ("conv2a", "bn2a", False), ("conv2b", "bn2b", True), ("conv3a", "bn3a", False),
            ("conv3b", "bn3b", False)]


class Probe:
    """Optional HIP-event spans around named launches (bench.py: live duration of the dominant kernels inside the timed
    region, on the stream they are launched on).  `det.probe = Probe()` switches it on; None costs nothing."""

    def __init__(self):
        self.spans = {}

    class _Span:
        def __init__(self, probe, name):
            self.p, self.name = probe, name

        def __enter__(self):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

        def __exit__(self, *a):
            self.e1.record()
            self.p.spans.setdefault(self.name, []).append((self.e0, self.e1))

    def __call__(self, name):
        return Probe._Span(self, name)

    def mean_ms(self):
        """name -> mean duration in ms (call after a device synchronize)."""
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.spans.items()}

    def median_ms(self):
        """name -> median duration in ms: one stalled launch (a context switch on a shared box) moves a mean over 10 steps, not this."""
        out = {}
        for k, v in self.spans.items():
            d = sorted(a.elapsed_time(b) for a, b in v)
            out[k] = d[len(d) // 2] if len(d) % 2 else 0.5 * (d[len(d) // 2 - 1] + d[len(d) // 2])
        return out


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOSPAN = _NoSpan()


class DetectorM3D:
    def __init__(self, params, cfg):
        """params: dict of CUDA fp32 tensors with the reference's state-dict keys; cfg: object with the
        attributes of oracle.Cfg (stride, anchors, pre/post_nms_topN, thresholds, ...)."""
        self.cfg = cfg
        self.P = params
        self.probe = None
        self.anchors = np.ascontiguousarray(cfg.anchors, dtype=np.float64)
        self.body = []
        for cname, bname, pool in dsn_layers(cfg.stride):
            c, b = "Conv_Body." + cname, "Conv_Body." + bname
            conv = ops.PackedConv3d(params[c + ".weight"])
            scale = (params[b + ".weight"] / torch.sqrt(params[b + ".running_var"] + BN_EPS)).contiguous()
            shift = ((params[c + ".bias"] - params[b + ".running_mean"]) * scale + params[b + ".bias"]).contiguous()
            self.body.append((conv, scale, shift, pool))
        # plain-forward 3x3x3 layers also get Winograd weights: F(2x2,3x3) on (y,x) (4/9 of the MFMA work; M3D_WINO=2,
        # default) or F(2,3) along x (2/3; M3D_WINO=1); maps >= 24 voxels wide.  M3D_WINO=0: direct kernels only.
        # The PRM engine keeps using the direct kernels in self.body (its masks test exact zeros).
        self.wino_mode = int(os.environ.get("M3D_WINO", "2"))
        self.use_wino = self.wino_mode != 0
        self.body_wino = [ops.WinoConv3d(params["Conv_Body." + cname + ".weight"], two_d=(self.wino_mode == 2))
                          if (self.use_wino and params["Conv_Body." + cname + ".weight"].shape[-1] == 3) else None
                          for cname, _, _ in dsn_layers(cfg.stride)]
        self.stem_wino = None                                      # conv1a: F(2,5) along x
        w1 = params["Conv_Body.conv1a.weight"]
        if self.use_wino and tuple(w1.shape[1:]) == (1, 5, 5, 5):
            self.stem_wino = ops.StemWinoConv3d(w1)
        # Round 6: the 3^3 layers with cin % 16 == 0 on the f16 matrix cores at fp32 accuracy (csrc/conv3d_zw.hip: f16x2 split + F(2,3)
        # along z; 1.5 x the fp32 F(2x4,3x3) kernel on the step's layers).  The operand bound travels with the activations (tensor
        # attribute `_m3d_bound`, filled by the producing launch's epilogue); M3D_CONV_F16=0 keeps the fp32 Winograd kernels (A/B).
        self.conv_f16 = self.use_wino and os.environ.get("M3D_CONV_F16", "1") == "1"
        self.body_zw = [ops.ZwConv3d(params["Conv_Body." + cname + ".weight"])
                        if (self.conv_f16 and ops.ZwConv3d.supported(params["Conv_Body." + cname + ".weight"])) else None
                        for cname, _, _ in dsn_layers(cfg.stride)]
        self.rpn_conv_zw = ops.ZwConv3d(params["RPN.RPN_conv.weight"]) \
            if (self.conv_f16 and ops.ZwConv3d.supported(params["RPN.RPN_conv.weight"])) else None
        self.rpn_conv = ops.PackedConv3d(params["RPN.RPN_conv.weight"])
        self.rpn_conv_wino = ops.WinoConv3d(params["RPN.RPN_conv.weight"], two_d=(self.wino_mode == 2)) if self.use_wino else None
        self.rpn_conv_bias = params["RPN.RPN_conv.bias"].contiguous()
        self.A = params["RPN.RPN_cls_score.weight"].shape[0]
        # the two 1x1x1 heads share their input: one conv with A + 6A output channels (rpn_heads.py:96-98)
        w = torch.cat([params["RPN.RPN_cls_score.weight"], params["RPN.RPN_bbox_pred.weight"]], 0).contiguous()
        self.rpn_heads = ops.PackedConv3d(w)
        self.rpn_heads_bias = torch.cat([params["RPN.RPN_cls_score.bias"], params["RPN.RPN_bbox_pred.bias"]]).contiguous()
        self.has_head = "Box_Head.fc1.weight" in params
        # RoIAlign3D's matrix-core form for sub-volumes of <= 128 voxels (round 6 A/B, M3D_ROI_GEMM=1).  NOT the product path: correct
        # (tests/test_gpu_ops.py) and slower - 0.49 instead of 0.23 ms at R = 1281: its two launches take 0.16 ms each (1 408 dword stores
        # per RoI from two waves per SIMD) and run AFTER the separable launch, whose time is set by its heavy RoIs either way
        self.roi_gemm = os.environ.get("M3D_ROI_GEMM", "0") == "1"
        if self.has_head:
            self.outs_w = torch.cat([params["Box_Outs.cls_score.weight"], params["Box_Outs.bbox_pred.weight"]], 0).contiguous()
            self.outs_b = torch.cat([params["Box_Outs.cls_score.bias"], params["Box_Outs.bbox_pred.bias"]]).contiguous()
            # fc1 / fc2 on the 16-bit matrix cores at fp32 accuracy (csrc/fc_gemm.hip): weights are cut once here.  Round 6 default: the
            # f16x2 split (two scaled fp16 planes per operand, three products per fp32 product); M3D_FC_SPLIT=bf16x3 keeps the exact 3-way
            # bf16 cut (six products), M3D_FC_SPLIT=0 the fp32-input MFMA kernel (A/B tooling).
            self.fc_split = {}
            mode = os.environ.get("M3D_FC_SPLIT", "f16x2")
            if mode != "0":
                cls = ops.SplitLinear if mode in ("bf16x3", "1") else ops.SplitLinearF16
                for name in ("fc1", "fc2"):
                    w = params["Box_Head.%s.weight" % name]
                    if cls.supported(w):
                        self.fc_split[name] = cls(w, params["Box_Head.%s.bias" % name])

    # ---- lib/modeling/DSN.py:57-68
    def body_layer(self, li, x, bound_slot=None):
        """conv + eval-BN + ReLU (+ MaxPool) of body layer li: Winograd-x kernel where it has a tile configuration,
        otherwise the direct MFMA kernel; the pool is fused into the conv launch when the map is large enough."""
        conv, scale, shift, pool = self.body[li]
        wino = self.body_wino[li]
        width = x.shape[-1]
        small = x[0].numel() * 4 < 0x7FFFFFFF          # the Winograd kernels address one batch item with 32-bit buffer offsets
        if not small:
            wino = None
        if li == 0 and small and self.stem_wino is not None and self.stem_wino.supports(width):
            sb = (bound_slot if bound_slot is not None else True) if self.conv_f16 else False
            return self.stem_wino.pooled(x, scale=scale, shift=shift, relu=True, bound=sb) if pool else \
                self.stem_wino(x, scale=scale, shift=shift, relu=True, bound=sb)
        zw = self.body_zw[li]
        if zw is not None and small and self._zw_ok(zw, x):
            fused = pool and zw.supports(x.shape, pool=True)
            y, ym = zw(x, self._bound(x), scale=scale, shift=shift, relu=True, pool=fused, out_max=bound_slot)
            if pool and not fused:
                y = ops.maxpool3d_2x(y)
            y._m3d_bound = (ym, y._version)                   # (a pooled map's bound is its un-pooled map's)
            return y
        if wino is not None and wino.supports(width, (x.shape[0],) + tuple(x.shape[2:])):
            if pool and wino.supports_pool(width):
                return wino.pooled(x, scale=scale, shift=shift, relu=True)
            x = wino(x, scale=scale, shift=shift, relu=True)
            return ops.maxpool3d_2x(x) if pool else x
        if pool and conv.supports_pool(width, x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4]):
            return conv.pooled(x, scale=scale, shift=shift, relu=True)          # conv+BN+ReLU+MaxPool in one kernel
        x = conv(x, scale=scale, shift=shift, relu=True)
        return ops.maxpool3d_2x(x) if pool else x

    @staticmethod
    def _bound(x):
        """The operand bound of an activation tensor for the f16x2 conv kernels: left on the tensor by the launch that produced it, else
        one sweep of it (the chain's first layer, or a tensor that came from somewhere else)."""
        b = getattr(x, "_m3d_bound", None)
        if b is not None and b[1] == x._version:           # (an in-place write since the producing launch makes the bound stale: sweep again)
            return b[0]
        return ops.ZwConv3d.bound_of(x)

    @staticmethod
    def _zw_ok(zw, x):
        """the f16x2 kernel runs one workgroup per (64 channels, 32 x 4 x 2 voxels): maps that give it less than ~0.8 of a round of the
        chip's 256 CUs stay with the fp32 Winograd kernels (split-K over workgroups)"""
        return zw.supports(x.shape) and zw.units(x.shape) >= 200

    def span(self, name):
        return self.probe(name) if self.probe is not None else _NOSPAN

    # MFMA multiply-adds a kernel family ISSUES per algorithmic multiply-add: Winograd F(2x2,3x3) 16/36, F(2x4,3x3) 24/72, F(2,3) along x 4/6,
    # the stem's F(2,5) along x 78/125 (13 row pairs x 6 xi per output pair against 125 taps per output)
    # "f16x2 F(2,3)z" (csrc/conv3d_zw.hip): 36/54 of the direct products, each cut into three fp16 products - 2 f16 multiply-adds issued
    # per algorithmic one, priced against the f16 peak (dtype "f16" in conv_work's records)
    ISSUED_FRACTION = {"winograd F(2x2,3x3)": 4.0 / 9.0, "winograd F(2x4,3x3)": 1.0 / 3.0, "winograd F(2,3)x": 2.0 / 3.0,
                       "winograd F(2,5)x stem": 78.0 / 125.0, "direct": 1.0, "f16x2 F(2,3)z": 2.0}
    F16_KINDS = ("f16x2 F(2,3)z",)

    def conv_work(self, batch, size):
        """Per probe span of the convolution family (conv1a .. conv4b, rpn): algorithmic FLOPs (2*Cin*Cout*k^3 per output voxel), the
        FLOPs the chosen kernel issues on the matrix cores (Winograd fraction, output channels padded to blocks of 32) and the
        kernel kind, for a batch of `batch` volumes of `size` = (S, H, W) - the same decisions as body_layer() / rpn()."""
        out = {}
        S, H, W = size
        names = dsn_layers(self.cfg.stride)
        two_d = "winograd F(2x4,3x3)" if ops.lib().m3d_conv3d_wino2_family() == 4 else "winograd F(2x2,3x3)"

        def pad32(c):
            return (c + 31) // 32 * 32 / float(c)

        def pad64(c):
            return (c + 63) // 64 * 64 / float(c)
        for li, (cname, _, pool) in enumerate(names):
            w = self.P["Conv_Body." + cname + ".weight"]
            cout, cin, k = int(w.shape[0]), int(w.shape[1]), int(w.shape[-1])
            small = cin * S * H * W * 4 < 0x7FFFFFFF
            kind = "direct"
            if li == 0 and small and self.stem_wino is not None and self.stem_wino.supports(W):
                kind = "winograd F(2,5)x stem"
            elif small and self.body_zw[li] is not None and self.body_zw[li].supports((S, H, W)) and self.body_zw[li].units((batch, cin, S, H, W)) >= 200:
                kind = "f16x2 F(2,3)z"
            elif small and self.body_wino[li] is not None and self.body_wino[li].supports(W, (batch, S, H, W)):
                kind = two_d if self.wino_mode == 2 else "winograd F(2,3)x"
            alg = 2.0 * cin * cout * k ** 3 * S * H * W * batch
            out[cname] = dict(algorithmic_flop=alg, issued_flop=alg * self.ISSUED_FRACTION[kind] * (pad64(cout) if kind in self.F16_KINDS else pad32(cout)),
                              kernel=kind, shape="%d->%d k%d @ %dx%dx%d" % (cin, cout, k, S, H, W), dtype="f16" if kind in self.F16_KINDS else "f32")
            if pool:
                S, H, W = S // 2, H // 2, W // 2
        w = self.P["RPN.RPN_conv.weight"]
        cout, cin = int(w.shape[0]), int(w.shape[1])
        kind = "direct"
        if self.rpn_conv_zw is not None and cin * S * H * W * 4 < 0x7FFFFFFF and self.rpn_conv_zw.supports((S, H, W)) \
                and self.rpn_conv_zw.units((batch, cin, S, H, W)) >= 200:
            kind = "f16x2 F(2,3)z"
        elif self.rpn_conv_wino is not None and self.rpn_conv_wino.supports(W, (batch, S, H, W)) and cin * S * H * W * 4 < 0x7FFFFFFF:
            kind = two_d if self.wino_mode == 2 else "winograd F(2,3)x"
        alg = 2.0 * cin * cout * 27 * S * H * W * batch
        nh = 7 * self.A
        alg_h = 2.0 * cout * nh * S * H * W * batch
        out["rpn"] = dict(algorithmic_flop=alg + alg_h, issued_flop=alg * self.ISSUED_FRACTION[kind] * pad32(cout) + alg_h * pad32(nh),
                          dtype="f16" if kind in self.F16_KINDS else "f32",
                          kernel=kind + " (3^3 conv) + direct (the two 1^3 heads as one conv)",
                          shape="%d->%d k3, %d->%d k1 @ %dx%dx%d" % (cin, cout, cout, nh, S, H, W))
        return out

    def conv_body(self, x, first=0, last=None):
        names = dsn_layers(self.cfg.stride)
        # the f16x2 layers' operand bounds (one zeroed slot array per layer, filled by the producing launch): ONE fill for the whole body
        slots = torch.zeros((len(self.body), ops.ZwConv3d.SLOTS), dtype=torch.float32, device=x.device) if self.conv_f16 else None
        for li in range(first, len(self.body) if last is None else last):
            with self.span(names[li][0]):
                x = self.body_layer(li, x, slots[li] if slots is not None else None)
        return x

    def capture_body(self, x, first=0, last=None):
        """HIP graph of body layers [first, last) on the STATIC input tensor x: returns (replay, out).  `replay()` re-runs
        the captured launches (one graph launch instead of one per kernel); `out` is the static output tensor.
        The caller keeps x alive and refills it in place."""
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm up on a side stream (lazy module loads, attribute calls)
            for _ in range(2):
                self.conv_body(x, first, last)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self.conv_body(x, first, last)
        return g.replay, out

    # ---- lib/modeling/rpn_heads.py:94-116
    def rpn(self, feat):
        rc = self.rpn_conv_wino if (self.rpn_conv_wino is not None and self.rpn_conv_wino.supports(feat.shape[-1], (feat.shape[0],) + tuple(feat.shape[2:]))
                                    and feat[0].numel() * 4 < 0x7FFFFFFF) else self.rpn_conv
        zw = self.rpn_conv_zw
        if zw is not None and feat[0].numel() * 4 < 0x7FFFFFFF and self._zw_ok(zw, feat):
            h, _ = zw(feat, self._bound(feat), shift=self.rpn_conv_bias, relu=True, out_max=False)
        else:
            h = rc(feat, shift=self.rpn_conv_bias, relu=True)
        return self.rpn_outputs(h)

    def rpn_outputs(self, h):
        """The two 1x1x1 heads as one conv + sigmoid (rpn_heads.py:96-98,116): (prob [B,A,s,h,w], deltas [B,6A,s,h,w])."""
        return self.rpn_heads.split_sigmoid(h, self.A, shift=self.rpn_heads_bias)      # one launch: conv + sigmoid + the split

    def _fused_ok(self, prob):
        """The one-workgroup-per-tile kernels (csrc/box_fused.hip) hold at most 2048 candidates per tile."""
        total = prob.shape[-4] * prob.shape[-3] * prob.shape[-2] * prob.shape[-1]
        k = total if (self.cfg.pre_nms_topN <= 0 or self.cfg.pre_nms_topN >= total) else self.cfg.pre_nms_topN
        return k <= ops.fused_max_boxes()

    def proposals(self, prob, deltas, im_info, item=0):
        """One tile: (rois [R,7], probs [R,1], keep_idx [R]) - generate_proposals_3d.py:19-104."""
        c = self.cfg
        if self._fused_ok(prob):
            rois, probs, kidx, num = ops.generate_proposals3d_batched(
                prob[item:item + 1], deltas[item:item + 1], self.anchors, float(c.stride), im_info, c.pre_nms_topN,
                c.post_nms_topN, c.rpn_nms_thresh, c.rpn_min_size, first_batch_index=item)
            r = int(num.item())
            return rois[0, :r], probs[0, :r].unsqueeze(1), kidx[0, :r]
        return ops.generate_proposals3d(prob[item].contiguous(), deltas[item], self.anchors, float(c.stride), im_info,
                                        c.pre_nms_topN, c.post_nms_topN, c.rpn_nms_thresh, c.rpn_min_size, batch_index=item)

    # ---- lib/modeling/fast_rcnn_heads.py:104-117,39-47
    def box_head_outputs(self, feat, rois, clip_to=None):
        """RoIAlign3D -> fc1 -> fc2 -> (cls_score | bbox_pred as one GEMM) -> softmax / deltas / decoded + clipped boxes in one launch:
        (cls [R,nc], bbox [R,6nc], pred_boxes [R,6nc]).  rois [R,7]."""
        c, P = self.cfg, self.P
        with self.span("roi_align3d"):
            # the feature maps' largest magnitude (16 MB sweep, once): the operand scale of the f16x2 kernels - RoIAlign3D's matrix-core form
            # for small sub-volumes and, since every RoIAlign value is a convex combination of feature-map values, fc1's x scale
            fmax = ops.absmax(feat) if self.roi_gemm or any(isinstance(v, ops.SplitLinearF16) for v in self.fc_split.values()) else None
            x = ops.roi_align3d_forward(feat, rois, c.roi_res, c.roi_res, c.roi_res, 1.0 / c.stride, c.sampling_ratio,
                                        feat_absmax=fmax if self.roi_gemm else None)
        x = x.view(x.shape[0], -1)
        for name in ("fc1", "fc2"):                                                                 # :114-115
            with self.span(name):
                if name in self.fc_split:
                    lin = self.fc_split[name]
                    if isinstance(lin, ops.SplitLinearF16):
                        # fc1's operand scale from the feature map (16 MB) instead of the RoIAlign output (440 MB): every RoIAlign value is a
                        # convex combination of feature-map values; fc2's input is small and swept by the call itself
                        x = lin(x, relu=True, x_bound=fmax if name == "fc1" else None)
                    else:
                        x = lin(x, relu=True)
                else:
                    x = ops.linear(x, P["Box_Head.%s.weight" % name], P["Box_Head.%s.bias" % name], relu=True)
        o = ops.linear(x, self.outs_w, self.outs_b)              # cls_score and bbox_pred share their input: one GEMM (:42,45)
        return ops.box_head_outputs(o, rois, c.num_classes, c.bbox_reg_weights, clip_to=clip_to)   # :43-44 (eval) + core/test.py:250-251

    def box_head(self, feat, rois):
        cls, bbox, _ = self.box_head_outputs(feat, rois)
        return cls, bbox

    # ---- lib/core/test.py:806-883 (device-side; SOFT_NMS / BBOX_VOTE off)
    def box_results_with_nms_and_limit(self, scores, boxes, scores_keep_idx=None):
        c = self.cfg
        R = int(scores.shape[0])
        if 0 < R <= ops.fused_max_boxes():                # one launch, one host read of the per-class counts
            off = torch.arange(2, dtype=torch.int32, device=scores.device) * R      # [0, R] built on the device (a host list would be a blocking copy)
            kin = None if scores_keep_idx is None else scores_keep_idx.to(torch.int64).contiguous()
            cb, ck, cnt = ops.box_results3d_batched(scores, boxes, kin, off, c.num_classes, c.score_thresh, c.nms, c.detections_per_im, R)
            n = cnt[0].cpu().tolist()
            cls_boxes = [cb[0, j, :n[j]] for j in range(c.num_classes)]
            cls_keep = [ck[0, j, :n[j]] if scores_keep_idx is not None else torch.zeros((0,), dtype=torch.int64, device=scores.device)
                        for j in range(c.num_classes)]
            im_results = torch.cat([cls_boxes[j] for j in range(1, c.num_classes)], 0)
            return im_results[:, -1], im_results[:, :-1], cls_boxes, cls_keep
        cls_boxes = [torch.zeros((0, 7), device=scores.device) for _ in range(c.num_classes)]
        cls_keep = [torch.zeros((0,), dtype=torch.int64, device=scores.device) for _ in range(c.num_classes)]
        for j in range(1, c.num_classes):
            inds = torch.nonzero(scores[:, j] > c.score_thresh).squeeze(1)           # :836
            dets_j = torch.cat([boxes[inds, j * 6:(j + 1) * 6], scores[inds, j:j + 1]], 1).float().contiguous()
            keep = ops.nms3d(dets_j, c.nms)                                          # :851
            cls_boxes[j] = dets_j[keep]
            if scores_keep_idx is not None:
                cls_keep[j] = scores_keep_idx[inds][keep]
        if c.detections_per_im > 0:                                                  # :869-878 (cap semantics)
            image_scores = torch.cat([cls_boxes[j][:, -1] for j in range(1, c.num_classes)])
            if image_scores.numel() > c.detections_per_im:
                image_thresh = torch.sort(image_scores)[0][-c.detections_per_im]
                for j in range(1, c.num_classes):
                    keep = torch.nonzero(cls_boxes[j][:, -1] >= image_thresh).squeeze(1)
                    cls_boxes[j] = cls_boxes[j][keep]
                    if scores_keep_idx is not None:
                        cls_keep[j] = cls_keep[j][keep]
        im_results = torch.cat([cls_boxes[j] for j in range(1, c.num_classes)], 0)
        return im_results[:, -1], im_results[:, :-1], cls_boxes, cls_keep

    # ---- several tiles of equal size at once.  The reference runs one tile per forward (core/test.py:91-145); tiles are
    # independent, so the convolutions (batch dimension of the kernels), RoIAlign (batch index of each RoI) and the box-head
    # GEMMs (rows = RoIs of all tiles) run once for the whole batch: the 16^3-class layers fill the chip and the 360 MB fc1
    # weight matrix is streamed once per batch instead of once per tile.  Per-tile results equal detect_tile's.
    def detect_batch(self, data, im_info=None, as_dicts=True):
        """data [B,1,S,H,W].  as_dicts=True: list of B per-tile dicts with detect_tile's keys (two host reads in total: the
        proposal counts and the detection counts).  as_dicts=False: one dict of batched device tensors (rois [B,rows,7] +
        num_rois, cls_boxes [B,nc,rows,7] + cls_counts [B,nc], ...) after ONE host read (the proposal counts, which size the
        box-head GEMM) - what bench.py and the sharded driver consume.
        = detect_batch_finish(detect_batch_begin(...)): a driver that has the next batch ready can launch its `begin` (backbone,
        RPN, proposals: no host read) before it finishes this one, on another stream (bench.py does)."""
        return self.detect_batch_finish(self.detect_batch_begin(data, im_info), as_dicts)

    def detect_batch_begin(self, data, im_info=None):
        """Everything up to and including the proposal kernels, launched on the current stream without any host read; the
        proposal counts travel to pinned host memory behind an event.  Returns the state detect_batch_finish takes."""
        c = self.cfg
        S, H, W = data.shape[-3:]
        if im_info is None:
            im_info = np.array([S, H, W, 1.0], np.float64)
        feat = self.conv_body(data)
        with self.span("rpn"):
            prob, deltas = self.rpn(feat)
        st = dict(feat=feat, prob=prob, deltas=deltas, im_info=im_info, fused=self._fused_ok(prob))
        if st["fused"]:
            with self.span("proposals"):
                st["props"] = ops.generate_proposals3d_batched(
                    prob, deltas, self.anchors, float(c.stride), im_info, c.pre_nms_topN, c.post_nms_topN, c.rpn_nms_thresh,
                    c.rpn_min_size)
                num = st["props"][3]
                st["num_host"] = self._pinned_counts(num)
                # ONE launch packs the valid RoIs of all tiles and their score indices for the box head (+ the row offsets) from the
                # device-side counts and writes those counts into the pinned host buffer itself: the host waits for the event behind
                # it - no device-to-host copy, no torch.cat, finish() only slices
                st["rois_packed"], st["kidx_packed"], st["offs_dev"] = ops.compact_rows2(st["props"][0], st["props"][2], num, st["num_host"])
                st["ready"] = torch.cuda.Event()
                st["ready"].record()
                # finish() may run on another stream: it orders its readers of the packed buffers behind this event with a device-side
                # wait as well (the host wait on it covers the same-thread case)
                st["packed_ready"] = st["ready"]
        return st

    def _pinned_counts(self, like):
        """A pinned host buffer for the proposal counts of one batch in flight.  Buffers are pooled: `detect_batch_finish` hands its
        state's buffer back after reading it; a begin() that finds the pool empty (more batches in flight than ever before) pins a
        new one, so an earlier batch's counts are never overwritten however many begin() calls are outstanding."""
        pool = self.__dict__.setdefault("_count_pool", {})
        key = (like.numel(), like.dtype)
        free = pool.setdefault(key, [])
        if not free:                                         # pinning is slow (a driver call): four at once
            free.extend(torch.empty((like.numel(),), dtype=like.dtype).pin_memory() for _ in range(4))
        return free.pop()

    def _release_counts(self, buf):
        self.__dict__.setdefault("_count_pool", {}).setdefault((buf.numel(), buf.dtype), []).append(buf)

    def detect_batch_finish(self, st, as_dicts=True):
        """The rest of detect_batch on the current stream (which may differ from begin's: it waits for begin's event, and the
        tensors begin produced are marked as used here for the caching allocator)."""
        c = self.cfg
        feat, prob, deltas, im_info = st["feat"], st["prob"], st["deltas"], st["im_info"]
        B = feat.shape[0]
        if not st["fused"]:
            return self._detect_batch_unfused(feat, prob, deltas, im_info, as_dicts)
        rois_b, probs_b, kidx_b, num = st["props"]
        st["ready"].synchronize()                                                  # host read 1: sizes the GEMM rows
        counts = st["num_host"].tolist()
        total = sum(counts)
        # the GPU is idle from here until the first launch below: nothing but that launch's arguments is prepared first, the rest of
        # the bookkeeping (stream marks, dictionaries, host-side offsets) follows behind the box head's launches
        head = self.has_head and total > 0
        torch.cuda.current_stream().wait_event(st["packed_ready"])                 # device-side: compact_rows -> RoIAlign / box results
        if head:
            rois, kidx = st["rois_packed"][:total], st["kidx_packed"][:total]       # (were two torch.cat launches behind the host read)
            cls, bbox, pred = self.box_head_outputs(feat, rois, clip_to=im_info[:3])   # one RoIAlign + one GEMM chain for all tiles
        self._release_counts(st.pop("num_host"))                                   # read: back to the pool
        cur = torch.cuda.current_stream()
        for t in (feat, prob, deltas, rois_b, probs_b, kidx_b, st["rois_packed"], st["kidx_packed"], st["offs_dev"]):
            t.record_stream(cur)
        rows = rois_b.shape[1]
        raw = dict(feat=feat, rpn_prob=prob, rpn_deltas=deltas, rois=rois_b, roi_probs=probs_b, keep_idx=kidx_b, num_rois=counts)
        offs = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
        # the row offsets are on the device already (compact_rows in begin()): a pageable `.to(device)` of `offs` where they are used,
        # behind the box head, made the host wait for the whole box head and left the GPU idle for 44 us per step
        offs_dev = st["offs_dev"]
        if head:
            with self.span("box_results"):
                cb, ck, cnt = ops.box_results3d_batched(cls, pred, kidx, offs_dev,
                                                        c.num_classes, c.score_thresh, c.nms, c.detections_per_im, rows)
            raw.update(cls=cls, bbox=bbox, pred_boxes=pred, offsets=offs, cls_boxes=cb, cls_keep=ck, cls_counts=cnt)
        if not as_dicts:
            return raw
        outs = [dict(feat=feat[b:b + 1], rpn_prob=prob[b:b + 1], rpn_deltas=deltas[b:b + 1], rois=rois_b[b, :counts[b]],
                     roi_probs=probs_b[b, :counts[b]].unsqueeze(1), keep_idx=kidx_b[b, :counts[b]]) for b in range(B)]
        if "cls_boxes" in raw:
            n = raw["cls_counts"].cpu().tolist()                                   # host read 2
            for b in range(B):
                if counts[b] == 0:
                    continue
                o0, o1 = int(offs[b]), int(offs[b + 1])
                cls_boxes = [cb[b, j, :n[b][j]] for j in range(c.num_classes)]
                cls_keep = [ck[b, j, :n[b][j]] for j in range(c.num_classes)]
                im = torch.cat([cls_boxes[j] for j in range(1, c.num_classes)], 0)
                outs[b].update(cls=cls[o0:o1], bbox=bbox[o0:o1], pred_boxes=pred[o0:o1], det_scores=im[:, -1], det_boxes=im[:, :-1],
                               cls_boxes=cls_boxes, cls_keep=cls_keep)
        return outs

    def _detect_batch_unfused(self, feat, prob, deltas, im_info, as_dicts):
        """pre_nms_topN beyond the fused kernels' 2048 candidates per tile: per-tile multi-launch proposal / NMS kernels."""
        if not as_dicts:
            raise ops.M3DError("detect_batch(as_dicts=False) needs pre_nms_topN <= %d" % ops.fused_max_boxes())
        c = self.cfg
        B = feat.shape[0]
        props = [self.proposals(prob, deltas, im_info, item=b) for b in range(B)]
        outs = [dict(feat=feat[b:b + 1], rpn_prob=prob[b:b + 1], rpn_deltas=deltas[b:b + 1], rois=props[b][0], roi_probs=props[b][1],
                     keep_idx=props[b][2]) for b in range(B)]
        counts = [int(p[0].shape[0]) for p in props]
        if not self.has_head or sum(counts) == 0:
            return outs
        rois = torch.cat([p[0] for p in props], 0)
        cls, bbox, pred = self.box_head_outputs(feat, rois, clip_to=im_info[:3])
        o = 0
        for b in range(B):
            n = counts[b]
            if n == 0:
                continue
            cb, bb, pb = cls[o:o + n], bbox[o:o + n], pred[o:o + n]
            sc, bx, cls_boxes, cls_keep = self.box_results_with_nms_and_limit(cb, pb, props[b][2])
            outs[b].update(cls=cb, bbox=bb, pred_boxes=pb, det_scores=sc, det_boxes=bx, cls_boxes=cls_boxes, cls_keep=cls_keep)
            o += n
        return outs

    # ---- one tile, detection only: model_builder.py:151-240 + core/test.py:194-263
    def detect_tile(self, data, im_info=None):
        """One tile [1,1,S,H,W] -> dict of every intermediate the parity tests compare (same kernels as detect_batch)."""
        return self.detect_batch(data, im_info)[0]
