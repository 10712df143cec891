"""Peak-response mapping on the HIP kernels: host-side counterpart of PeakResponseMapping_3d.forward
(lib/prm/peak_response_mapping_3d.py:85-193) with the per-conv rule of lib/prm/peak_backprop_3d.py:8-44.

Forward: every conv runs twice, as in the reference's pr_conv3d — the response conv (bias + eval BatchNorm +
ReLU fused in the epilogue) and the norm conv on (x - min(x)) with relu(W).  The RPN_bbox_pred norm conv the
reference also computes (its Conv3d is patched too) is skipped: nothing back-propagates through it.
Backward: ALL kept peaks of the tile at once, as batches of receptive-field windows (csrc/prm.hip).
"""
import numpy as np
import os
import torch

from . import ops
from .model import DetectorM3D, _NOSPAN


class PRMEngine:
    def __init__(self, det: DetectorM3D, peak_chunk=None, window_budget=3 << 30, fused_stem=True, strip_wino=True, strip_min=16, wino_forward=True, small_gemm=True,
                 strip_f24=True, strip_f24_min=16, norm_stream=True, backward_streams=1, backward_split_min=8, slab_strips=True, x3_norm=True,
                 fused_prepare=True, skip_dead_peaks=True, x3_f16=True, strip_zw=None):
        self.det = det
        # skip_dead_peaks: a kept peak whose RPN sigmoid is exactly 1.0f has the derivative (1 - y) y == 0: its seed, every layer of its
        # back-propagation and its map are exactly zero (the reference then returns 0 / 0 = NaN for it, peak_response_mapping_3d.py:170-171).
        # The selection kernel flags such peaks and prm_tile back-propagates only the others; the dead ones get zero windows, sum 0 and the
        # origin the kernels would have written - what the full batch gives for them, bit for bit (the live peaks' windows agree with
        # the full batch's up to summation order, as between any two batch sizes; tests/test_gpu_prm.py).  Trained detectors
        # saturate on their confident detections; rounds 1-3 of the nuclei bench were 67 dead peaks out of 67.
        self.skip_dead_peaks = bool(skip_dead_peaks)
        # strip_zw (round 6, second half): the quad-aligned strips' backward-data convs on the f16 matrix cores (ops.ZwConv3d.strip: f16x2 cut +
        # F(2,3) along z, one operand scale per window) where the conv has >= 64 output channels; such a layer runs conv + prepare as two
        # launches (the fp32 kernel's fused prepare epilogue does not exist there; the prepare launch leaves every peak's largest value for
        # the conv's per-window scales).  Nuclei tile 13.9 -> 12.4 ms, soma 4.67 -> 4.51.  None: environment M3D_PRM_STRIP_ZW (default on;
        # 0 = the fp32 F(2x4,3x3) strips of rounds 4-5 everywhere).
        self.strip_zw = (os.environ.get("M3D_PRM_STRIP_ZW", "1") == "1") if strip_zw is None else bool(strip_zw)
        # fused_prepare: where two consecutive layers both run on the quad-aligned strip with no pooling between them (conv3b -> conv3a,
        # conv2b -> conv2a), the upper conv's backward-data writes the lower layer's PREPARED strip from its epilogue
        # (ops.WinoConv3d.strip_prepare): the bare gradient strip is never stored and re-read, and the prepare launch disappears.
        # False keeps conv + prm_prepare as two launches (what the tests compare the fused path with, bit for bit)
        self.fused_prepare = bool(fused_prepare)
        # norm_stream: prm_tile runs the norm convs on a second HIP stream next to proposals / box head / peak selection (launches of
        # 1-128 workgroups that leave most of the chip idle) instead of queueing them behind those launches on the tile's stream
        self.norm_stream = bool(norm_stream)
        self._side = None
        # slab_strips: where a layer's map has fewer planes than the window (a thin tile: the nuclei net's stride-2 maps are 32 planes
        # deep, its windows there 38 / 40), the strip stores the LAYER's planes instead of each window's - the planes of a cone outside
        # the volume (zero gradient in, never read out) are not stored, convolved or streamed (m3d.h: depth-clipped strips)
        self.slab_strips = bool(slab_strips)
        # x3_norm: the 3^3 norm convs on the bf16 matrix cores at fp32 accuracy (ops.X3Conv3d: exact bf16x3 cut, six products) instead of
        # the fp32-MFMA direct kernel; sums of non-negative products either way, so the exact zeros are the same
        self.x3_norm = bool(x3_norm)
        # x3_f16 (round 6): those norm convs with the f16x2 split (two scaled fp16 pieces per operand, three products instead of bf16x3's
        # six; ops.X3Conv3d(f16=True)): the operand X - min X is scaled by its largest value max X - min X, which the forward's two-launch
        # sweep now delivers beside the minima (ops.reduce_minmax_multi); the exact zeros are the same (tests/test_gpu_ops.py)
        self.x3_f16 = bool(x3_f16)
        # backward_streams = 2: the tile's peaks are back-propagated as two halves on two HIP streams - while one half's element-wise
        # `prepare` pass streams through HBM the other half's window convolution holds the matrix cores, and each launch's last,
        # partly filled round of workgroups is filled from the other chain.  A half's launches are the ones the one-stream engine issues
        # for that half alone; against the all-peaks batch a window can differ in the last bits (the library picks tile and K split from the
        # batch's shape), as it already does between `peak_chunk` settings.
        self.backward_streams = int(backward_streams)
        self.backward_split_min = int(backward_split_min)
        self._bstreams = None
        self.cfg = det.cfg
        self.probe = None                             # optional m3d.model.Probe: HIP-event spans around the phases of prm_tile (bench.py)
        # optional list: backward_windows appends one record per layer it ran - dict(layer, kernel, plan (family, tile, K split of a
        # strip conv; None elsewhere), P, U, strip, slab, t = a copy of the layer's output window batch) - so a test can tell WHICH
        # launch makes two batch sizes differ in the last bits (tests/test_gpu_prm.py; never set on a product path: it copies)
        self.trace = None
        # windows >= strip_min voxels wide run their backward-data through the F(2x2,3x3) kernel on the strip layout (all peaks
        # side by side along x): 4/9 of the MFMA work and tiles that fit; strip_wino=False keeps the direct kernel everywhere
        self.strip_wino = bool(strip_wino)
        # windows >= strip_f24_min wide take the F(2x4,3x3) family (1/3 instead of 4/9 of the MFMA work) on the QUAD-ALIGNED strip layout
        # (ops.strip_geometry mode 2: no output quad of one window reads another window's columns, so F(4,3)'s rounding-level footprint
        # stays inside the peak); the alignment costs columns (38 -> pitch 40 instead of 39, 16 -> 20 instead of 17): at 16 voxels the two
        # nearly cancel (3/4 of the MFMAs x 20/17 of the columns; measured -0.7 % of a tile), below 16 there are no strips at all.
        # strip_f24_min = 17 keeps the exactly-local F(2x2) family on the dense layout for the 16-voxel windows (rounds 3 / 4 first half)
        self.strip_f24 = bool(strip_f24) and self.strip_wino
        self.strip_f24_min = int(strip_f24_min)
        # forward: the RESPONSE convs of the un-pooled layers may take the detection path's Winograd kernels (values differ from the
        # direct kernel by ~1e-6 relative); the NORM convs never do - their exact zeros (sums of non-negative products) gate the
        # PostHook's `N < 1e-10` test, and a Winograd 1e-8 in place of a 0 would be divided by
        self.wino_forward = bool(wino_forward) and det.use_wino
        self.strip_min = int(strip_min)
        self.peak_chunk = peak_chunk or None          # 0 / None: size the batches from window_budget
        self.window_budget = window_budget
        P = det.P
        self.layers = []
        names = [c for c, _, _ in __import__("m3d.model", fromlist=["dsn_layers"]).dsn_layers(det.cfg.stride)]
        for (conv, scale, shift, pool), cname in zip(det.body, names):
            w = P["Conv_Body.%s.weight" % cname]
            self.layers.append(dict(name=cname, conv=conv, scale=scale, shift=shift, pool=pool, k=w.shape[2],
                                    norm_conv=self._norm_conv(w),
                                    dgrad=None if w.shape[2] == 5 else ops.PackedConv3d(w, ops.W_DGRAD_RELU),
                                    dgrad_wino=self._dgrad_wino(w) if (strip_wino and w.shape[2] == 3) else None,
                                    dgrad_wino24=self._dgrad_wino(w, local=False) if (strip_wino and strip_f24 and w.shape[2] == 3) else None,
                                    dgrad_zw=self._dgrad_zw(w) if (self.strip_zw and strip_wino and strip_f24 and w.shape[2] == 3) else None,
                                    dgrad_small=ops.SmallWindowDgrad(w) if (small_gemm and w.shape[2] == 3) else None, weight=w))
        w = P["RPN.RPN_conv.weight"]
        self.rpn = dict(norm_conv=self._norm_conv(w), dgrad=ops.PackedConv3d(w, ops.W_DGRAD_RELU),
                        dgrad_small=ops.SmallWindowDgrad(w) if small_gemm else None)
        self.stem_wf = ops.prm_stem_prepare_weights(P["Conv_Body.conv1a.weight"])
        # stem step on the matrix cores with the un-pool/prepare fused (csrc/prm_stem_mfma.hip); fused_stem=False keeps the
        # two-kernel VALU path (prepare + m3d_prm_stem_dgrad), which the tests compare it with
        self.fused_stem = fused_stem and P["Conv_Body.conv1a.weight"].shape[0] == 32
        self.stem_wa = ops.prm_stem_mfma_weights(P["Conv_Body.conv1a.weight"]) if self.fused_stem else None
        self.w_cls = P["RPN.RPN_cls_score.weight"]
        self.cls_norm_conv = ops.PackedConv3d(self.w_cls, ops.W_RELU)
        self.w_cls2d = self.w_cls.reshape(self.w_cls.shape[0], self.w_cls.shape[1]).contiguous()

    def span(self, name):
        return self.probe(name) if self.probe is not None else _NOSPAN

    def _norm_conv(self, w):
        direct = ops.PackedConv3d(w, ops.W_RELU)
        if not (self.x3_norm and ops.X3Conv3d.supported(w)):
            return direct
        x3 = ops.X3Conv3d(w, ops.W_RELU, f16=self.x3_f16)

        def conv(x, in_offset=None, in_max=None):
            # the 16-bit kernel's 64-channel x 16 x 4 x 4-voxel workgroups need about a round of the chip to pay (8 x 25 x 25 maps of the
            # nuclei net: 112 workgroups, slower than the fp32 kernel's small tiles; 16 x 40 x 40 and up: 1.3-1.7 x faster)
            if x3.workgroups(x.shape) >= 192:
                return x3(x, in_offset=in_offset, in_max=in_max) if x3.f16 else x3(x, in_offset=in_offset)
            return direct(x, in_offset=in_offset)
        return conv

    @staticmethod
    def _dgrad_wino(w, local=True):
        """backward-data of a 'same' 3^3 conv with relu(W) (peak_backprop_3d.py:41-42) as a forward conv: taps flipped, channel
        roles swapped; packed for the F(2x2,3x3) kernel (local: windows of different peaks are neighbours in the dense strip and need
        exact locality) or for the library's default family (F(2x4,3x3): the quad-aligned strip)."""
        wd = torch.relu(w).flip(2, 3, 4).transpose(0, 1).contiguous()
        return ops.WinoConv3d(wd, two_d=True, local=local)

    @staticmethod
    def _dgrad_zw(w):
        """the same backward-data conv packed for the f16x2 F(2,3)z kernel; None where that kernel would idle (fewer than 64 output
        channels = the forward conv's input channels) or has no configuration (its input channels = the forward conv's output channels % 16)"""
        wd = torch.relu(w).flip(2, 3, 4).transpose(0, 1).contiguous()
        return ops.ZwConv3d(wd) if (wd.shape[0] >= 64 and ops.ZwConv3d.supported(wd)) else None

    # ---------------------------------------------------------------- forward (peak_backprop_3d.py:37-44 per conv)
    # pr_conv3d computes two convolutions per layer: the response Y = conv(X, W, b) that feeds the next layer, and the norm conv
    # N = conv(X - min X, relu(W)) that only the backward hooks read.  Nothing of the detection path (RPN, proposals, box head, box
    # results) depends on N, so the forward is split: `forward_response` is the detection-mode forward (+ pool argmax), and
    # `forward_norms` computes N (and min X) for a set of layers later - prm_tile queues them behind the launches whose results the
    # host has to wait for (RoI count, peak count), so those waits never leave the GPU idle.
    def forward_response(self, data):
        det = self.det
        saved = []
        x = data
        for li, L in enumerate(self.layers):
            wnp = det.body_wino[li] if (self.wino_forward and L["pool"] and L["k"] == 3) else None
            if wnp is not None and wnp.two_d and wnp.supports_pool(x.shape[-1]) and x[0].numel() * 4 < 0x7FFFFFFF and \
                    wnp.supports(x.shape[-1], (x.shape[0],) + tuple(x.shape[2:])):
                xn, am = wnp.pooled(x, scale=L["scale"], shift=L["shift"], relu=True, return_argmax=True)     # F(2x4,3x3) + pool + argmax
            elif L["pool"] and L["conv"].supports_pool(x.shape[-1], x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4]):
                xn, am = L["conv"].pooled(x, scale=L["scale"], shift=L["shift"], relu=True, return_argmax=True)
            else:
                wn = det.body_wino[li] if self.wino_forward else None
                zw = det.body_zw[li] if (self.wino_forward and det.conv_f16) else None
                if zw is not None and not L["pool"] and x[0].numel() * 4 < 0x7FFFFFFF and det._zw_ok(zw, x):
                    # response conv on the f16 matrix cores as in detection mode (csrc/conv3d_zw.hip; the input's bound: left by the
                    # launch that produced x, else one sweep)
                    y, ym = zw(x, det._bound(x), scale=L["scale"], shift=L["shift"], relu=True)
                    y._m3d_bound = (ym, y._version)
                elif wn is not None and not L["pool"] and wn.supports(x.shape[-1], (x.shape[0],) + tuple(x.shape[2:])):
                    y = wn(x, scale=L["scale"], shift=L["shift"], relu=True)        # response conv: Winograd as in detection mode
                else:
                    y = L["conv"](x, scale=L["scale"], shift=L["shift"], relu=True)
                if L["pool"]:
                    xn, am = ops.maxpool3d_2x(y, return_argmax=True)
                else:
                    xn, am = y, None
            saved.append(dict(name=L["name"], x=x[0], off=None, n=None, scale=L["scale"], pool=L["pool"], argmax=None if am is None else am[0],
                              xnext=xn[0], k=L["k"], dgrad=L["dgrad"], dgrad_wino=L["dgrad_wino"], dgrad_wino24=L["dgrad_wino24"], dgrad_zw=L["dgrad_zw"], dgrad_small=L["dgrad_small"],
                              weight=L["weight"], norm_conv=L["norm_conv"]))
            x = xn
        feat = x
        wn = det.rpn_conv_wino if self.wino_forward else None
        if wn is not None and wn.supports(feat.shape[-1], (feat.shape[0],) + tuple(feat.shape[2:])):
            h = wn(feat, shift=det.rpn_conv_bias, relu=True)
        else:
            h = det.rpn_conv(feat, shift=det.rpn_conv_bias, relu=True)
        saved.append(dict(name="RPN_conv", x=feat[0], off=None, n=None, scale=None, pool=False, argmax=None, xnext=h[0], k=3,
                          dgrad=self.rpn["dgrad"], dgrad_small=self.rpn["dgrad_small"], norm_conv=self.rpn["norm_conv"]))
        prob, deltas = det.rpn_outputs(h)
        top = dict(h=h[0], off_h=None, n_cls=None, prob=prob[0])
        return feat, prob, deltas, saved, top

    def forward_norms(self, saved, top, layers=None, cls=True, top_first=False):
        """The norm convs (+ input minima, + the stem's denominator map) of saved[i] for i in `layers` (default: all that are still
        missing) and, with cls, of the RPN_cls_score conv.  top_first: in the order the backward consumes them (RPN_cls_score, then the
        layers from the top down), each followed by an event on the current stream (rec["ready"]) that backward_windows waits for - the
        side-stream mode of prm_tile, where the backward of the small top windows runs beside the norm convs of the large bottom layers."""
        idx = [i for i in (range(len(saved)) if layers is None else layers) if saved[i]["n"] is None]
        need_cls = cls and top["n_cls"] is None
        # every `input.min()` of this call (peak_backprop_3d.py:38) in TWO launches, not two per layer
        xs = [saved[i]["x"] for i in idx] + ([top["h"]] if need_cls else [])
        if xs:
            mins, maxs = ops.reduce_minmax_multi(xs)
            for k, i in enumerate(idx):
                saved[i]["off"] = mins[k:k + 1]
                saved[i]["max"] = maxs[k:k + 1]
            if need_cls:
                top["off_h"] = mins[len(idx):len(idx) + 1]

        def cls_norm():
            if need_cls:
                h = top["h"].unsqueeze(0)
                top["n_cls"] = self.cls_norm_conv(h, in_offset=top["off_h"])[0]
                if top_first:
                    top["ready"] = torch.cuda.Event()
                    top["ready"].record()

        if top_first:
            cls_norm()
        for i in (reversed(idx) if top_first else idx):
            rec = saved[i]
            x = rec["x"].unsqueeze(0)
            nc = rec["norm_conv"]
            rec["n"] = (nc(x, in_offset=rec["off"]) if isinstance(nc, ops.PackedConv3d) else nc(x, in_offset=rec["off"], in_max=rec["max"]))[0]
            if rec["k"] == 5 and rec["pool"] and self.fused_stem:
                rec["den"] = ops.prm_den_pool(rec["argmax"], rec["xnext"], rec["n"])      # peak-independent part of the prepare step
            if top_first:
                rec["ready"] = torch.cuda.Event()
                rec["ready"].record()
        if not top_first:
            cls_norm()

    def forward(self, data):
        feat, prob, deltas, saved, top = self.forward_response(data)
        self.forward_norms(saved, top)
        return feat, prob, deltas, saved, top

    # ---------------------------------------------------------------- backward for a batch of peaks
    def backward_windows(self, peaks_ashw, saved, top, data):
        """peaks_ashw: int32 CUDA [P,4] (anchor, s, h, w).  Returns (windows [P,Wn,Wn,Wn] (un-normalised, clamped),
        sums [P], origins int32 [P,3]).

        All peaks go through a layer in ONE launch while the layer's window batch stays under `window_budget`
        bytes (the stride-8/4 layers have 3^3..18^3 windows: per-launch latency, not work, dominates there);
        the big-window tail (38^3..84^3) is processed in peak chunks to bound the working set.

        A window batch travels as a dict: t (tensor), strip (layout, see ops.prm_prepare), P, C, U, and up_off - the input
        offset of the layer that produced it when that layer ran without its PreHook epilogue (the Winograd path), so that
        the consumer multiplies by (X - up_off)."""
        budget = self.window_budget

        def fused(rec, wb):
            return self.fused_stem and rec["k"] == 5 and "den" in rec and ops.prm_stem_dgrad_fused_supported(wb["C"], wb["U"])

        def wino(rec, Wn):
            """0: no strip; 1: dense strip + exactly-local F(2x2); 2: quad-aligned strip + F(2x4)"""
            if not (self.strip_wino and rec["k"] == 3 and rec.get("dgrad_wino") is not None and Wn >= self.strip_min):
                return 0
            return 2 if (self.strip_f24 and rec.get("dgrad_wino24") is not None and Wn >= self.strip_f24_min) else 1

        def run_layer(rec, wb, origin, border, nxt=None):
            dims = (wb["P"], wb["C"], wb["U"])
            if rec.get("ready") is not None:                 # norm conv of this layer on prm_tile's side stream
                torch.cuda.current_stream().wait_event(rec["ready"])
            if wb.get("prepared"):                           # the layer above wrote this layer's prepared strip from its epilogue
                gn, Wn, strip, slab = wb["t"], wb["U"], wb["strip"], wb.get("slab", False)
                return conv_strip(rec, gn, wb["P"], Wn, strip, slab, origin, nxt)
            if fused(rec, wb):                   # un-pool + prepare + stem dgrad + PreHook in one MFMA kernel
                w, s, origin = ops.prm_stem_dgrad_fused(wb["t"], origin, rec["den"], rec["argmax"], rec["scale"], self.stem_wa,
                                                        data[0, 0], rec["off"], strip=wb["strip"],
                                                        xnext=rec["xnext"] if wb["up_off"] is not None else None,
                                                        up_off=wb["up_off"], dims=dims, slab=wb.get("slab", False))
                return (w, s), origin
            Wn = (2 if rec["pool"] else 1) * wb["U"] + 2 * border
            strip = wino(rec, Wn)
            slab = bool(strip) and self.slab_strips and rec["n"].shape[1] < Wn       # the layer's map is thinner than the window
            gn, origin = ops.prm_prepare(wb["t"], origin, rec["pool"], border, rec["argmax"], rec["xnext"], rec["scale"], rec["n"],
                                         in_strip=wb["strip"], out_strip=strip, up_off=wb["up_off"], dims=dims,
                                         in_slab=wb.get("slab", False), out_slab=slab,
                                         peak_max=(strip == 2 and rec.get("dgrad_zw") is not None))
            if rec["k"] == 5:                    # conv1a: 5^3, one input channel -> VALU stem dgrad
                w, s = ops.prm_stem_dgrad(gn, self.stem_wf, data[0, 0], rec["off"], origin)
                return (w, s), origin
            cout = rec["x"].shape[0]
            if strip:                            # Winograd over the whole strip; its PreHook multiply moves to the consumer
                return conv_strip(rec, gn, wb["P"], Wn, strip, slab, origin, nxt)
            small = rec.get("dgrad_small")
            if small is not None and Wn in small.SIZES:      # 3^3 / 5^3 / 7^3: all peaks in one dense GEMM (csrc/prm_small.hip)
                y = small(gn, rec["x"], rec["off"], origin)
            else:
                y = ops.conv3d_windowed(rec["dgrad"], gn, rec["x"], rec["off"], origin)
            return dict(t=y, strip=0, P=wb["P"], C=cout, U=Wn, up_off=None, slab=False), origin

        def conv_strip(rec, gn, P, Wn, strip, slab, origin, nxt):
            """backward-data of `rec`'s conv on its prepared strip; with the next layer (`nxt`) on the quad-aligned strip too and no pool
            between them, that layer's prepare runs in this conv's epilogue"""
            cout = rec["x"].shape[0]
            if strip == 2 and rec.get("dgrad_zw") is not None:
                pb = getattr(gn, "_m3d_peak_max", None)
                if self.fused_prepare and nxt is not None and nxt["k"] == 3 and not nxt["pool"] and wino(nxt, Wn + 2) == 2:
                    if nxt.get("ready") is not None:
                        torch.cuda.current_stream().wait_event(nxt["ready"])
                    nslab = self.slab_strips and nxt["n"].shape[1] < Wn + 2
                    r = rec["dgrad_zw"].strip_prepare(gn, (P, gn.shape[0], Wn), origin, nxt["xnext"], nxt["scale"], nxt["n"], rec["off"],
                                                      in_slab=slab, out_slab=nslab, bounds=pb)
                    if r is not None:
                        return dict(t=r[0], strip=2, P=P, C=cout, U=Wn + 2, up_off=None, slab=nslab, prepared=True,
                                    kernel="strip f16x2 F(2,3)z + fused prepare of " + nxt.get("name", "?"), plan=None), r[1]
                y = rec["dgrad_zw"].strip(gn, ops.strip_geometry(Wn, 2, P)[0], P, bounds=pb)
                if y is not None:
                    return dict(t=y, strip=2, P=P, C=cout, U=Wn, up_off=rec["off"], slab=slab, kernel="strip f16x2 F(2,3)z", plan=None), origin
            if self.fused_prepare and strip == 2 and nxt is not None and nxt["k"] == 3 and not nxt["pool"] and wino(nxt, Wn + 2) == 2:
                if nxt.get("ready") is not None:
                    torch.cuda.current_stream().wait_event(nxt["ready"])
                nslab = self.slab_strips and nxt["n"].shape[1] < Wn + 2
                r = rec["dgrad_wino24"].strip_prepare(gn, (P, gn.shape[0], Wn), origin, nxt["xnext"], nxt["scale"], nxt["n"], rec["off"],
                                                      in_slab=slab, out_slab=nslab)
                if r is not None:
                    return dict(t=r[0], strip=2, P=P, C=cout, U=Wn + 2, up_off=None, slab=nslab, prepared=True,
                                kernel="strip F(2x4) + fused prepare of " + nxt.get("name", "?"),
                                plan=rec["dgrad_wino24"].plan((1,) + tuple(gn.shape[-3:])) if self.trace is not None else None), r[1]
            wc = rec["dgrad_wino24"] if strip == 2 else rec["dgrad_wino"]
            y = wc(gn.unsqueeze(0))[0]
            return dict(t=y, strip=strip, P=P, C=cout, U=Wn, up_off=rec["off"], slab=slab, kernel="strip F(2x4)" if strip == 2 else "strip F(2x2) local",
                        plan=wc.plan((1,) + tuple(gn.shape[-3:])) if self.trace is not None else None), origin

        def take(wb, c0, c1):
            """peaks [c0, c1) of a window batch"""
            if wb["strip"]:
                pitch, lead, _ = ops.strip_geometry(wb["U"], wb["strip"], wb["P"])
                L = ops.strip_geometry(wb["U"], wb["strip"], c1 - c0)[2]      # the sub-strip has the same lead; its tail pad may hold the
                t = wb["t"][..., c0 * pitch:c0 * pitch + L].contiguous()        # next window's first columns (beyond every tile that is read)
            else:
                t = wb["t"][c0:c1].contiguous()
            return dict(wb, t=t, P=c1 - c0)

        def tail(layers, wb, origin):
            """Run the remaining layers (top -> bottom) on the given peak subset."""
            rec = layers[0]
            border = 2 if rec["k"] == 5 else 1
            P, Cc, U = wb["P"], rec["n"].shape[0], wb["U"]
            Wn = U if wb.get("prepared") else (2 if rec["pool"] else 1) * U + 2 * border     # prepared: U is already this conv's window
            cmax = max(Cc, rec["x"].shape[0])
            per_peak = 4 * Wn ** 3 * cmax * 2                            # prepare output + conv output
            if fused(rec, wb):
                per_peak = 4 * Wn ** 3                                    # the un-pooled window is never materialised
            chunk = self.peak_chunk if self.peak_chunk else max(1, min(P, int(budget // per_peak)))
            sm = wino(rec, Wn)
            if sm:                                                        # 32-bit offsets inside one strip
                pitch = ops.strip_geometry(Wn, sm, 1)[0]
                chunk = max(1, min(chunk, (2 ** 31 - 1) // (4 * cmax * Wn * Wn * pitch) - 1))
            if chunk >= P:
                out, o2 = run_layer(rec, wb, origin, border, layers[1] if len(layers) > 1 else None)
                if self.trace is not None:
                    if rec["k"] == 5:
                        self.trace.append(dict(layer=rec.get("name"), kernel="stem dgrad", plan=None, P=P, U=int(out[0].shape[1]), strip=0, slab=False,
                                               t=out[0].clone()))
                    else:
                        self.trace.append(dict(layer=rec.get("name"), kernel=out.get("kernel", "window GEMM / direct windowed conv"), plan=out.get("plan"),
                                               P=out["P"], U=out["U"], strip=out["strip"], slab=bool(out.get("slab")), t=out["t"].clone()))
                return (out, o2) if rec["k"] == 5 else tail(layers[1:], out, o2)
            outs = [tail(layers, take(wb, c0, min(P, c0 + chunk)), origin[c0:c0 + chunk].contiguous()) for c0 in range(0, P, chunk)]
            wins = torch.cat([o[0][0] for o in outs]); sums = torch.cat([o[0][1] for o in outs]); orig = torch.cat([o[1] for o in outs])
            return (wins, sums), orig

        def chain(pk):
            if top.get("ready") is not None:
                torch.cuda.current_stream().wait_event(top["ready"])
            g, origin = ops.prm_seed(pk, top["prob"], top["n_cls"], self.w_cls2d, top["h"], top["off_h"], return_origin=True)
            wb = dict(t=g, strip=0, P=g.shape[0], C=g.shape[1], U=1, up_off=None)
            return tail(list(reversed(saved)), wb, origin)

        pk = peaks_ashw.contiguous()
        P = pk.shape[0]
        try:
            if self.backward_streams >= 2 and P >= self.backward_split_min:
                if self._bstreams is None:
                    self._bstreams = [torch.cuda.Stream() for _ in range(self.backward_streams)]
                main = torch.cuda.current_stream()
                start = torch.cuda.Event()
                start.record()
                ns = self.backward_streams
                cuts = [P * i // ns for i in range(ns + 1)]
                parts = []
                for st, c0, c1 in zip(self._bstreams, cuts[:-1], cuts[1:]):
                    with torch.cuda.stream(st):
                        st.wait_event(start)
                        parts.append(chain(pk[c0:c1].contiguous()))
                        ev = torch.cuda.Event()
                        ev.record()
                    main.wait_event(ev)                                  # the halves' tensors go back to their streams' pools after this
                win = torch.cat([q[0][0] for q in parts]); sums = torch.cat([q[0][1] for q in parts]); origins = torch.cat([q[1] for q in parts])
                parts = None
            else:
                (win, sums), origins = chain(pk)
        finally:
            # `tail` calls itself, so the function object and its closure cell form a reference cycle that also holds `saved` (every
            # forward tensor of the tile) until the cyclic collector runs - by then the next tile has allocated its own: break it here
            tail = run_layer = take = fused = wino = chain = conv_strip = None
        return win, sums, origins

    # ---------------------------------------------------------------- lib/prm/peak_response_mapping_3d.py:85-193
    def prm_tile(self, data, peak_threshold=0.1, dense=True):
        """One tile.  Returns None when the tile yields no RoI or no detection above `peak_threshold` (the reference
        returns (None,)*5 and its driver `continue`s, infer_simple.py:225-226), else a dict: crm [1,A,s,h,w],
        peaks int64 [P,5] (b,a,s,h,w) and dets float64 [P,7] (host tensors: they arrive with the peak count), peaks_dev int32
        [P,4] / dets_dev float32 [P,7] (device), windows/sums/origins (cone-cropped maps) and, with dense=True, prms [P,S,H,W]
        (each map divided by its sum).

        Two host waits per tile - the RoI count that sizes the box head and the peak count that sizes the back-propagation.  The norm
        convs of the forward, which only the backward needs, run on a second stream beside the launches the host waits for (or, with
        norm_stream=False, queued behind them); what is still exposed of the second wait (the host is woken and builds the backward's
        launches: ~0.1 ms) disappears when tiles are fed through `TilePipeline`, which enqueues the next tile's forward in front of it."""
        g = self.prm_tile_phases(data, peak_threshold, dense)
        try:
            while True:
                next(g)
        except StopIteration as e:
            return e.value

    def prm_tile_phases(self, data, peak_threshold=0.1, dense=True):
        """prm_tile as a generator that yields in front of each host wait ("rois": the forward and the proposals are enqueued; "peaks":
        box head and peak selection are enqueued) and returns prm_tile's result; a tile on the per-stage path returns at the first
        `next`.  Whoever drives it may enqueue other work at a yield (TilePipeline: the next tile's forward)."""
        det, c = self.det, self.cfg
        S, H, W = data.shape[-3:]
        im_info = np.array([S, H, W, 1.0], np.float64)
        with self.span("forward_response"):
            feat, prob, deltas, saved, top = self.forward_response(data)
        if not det._fused_ok(prob) or not det.has_head:
            self.forward_norms(saved, top)
            return self._prm_tile_unfused(data, feat, prob, deltas, saved, top, peak_threshold, dense)
        norms_done = None
        if self.norm_stream:
            # the norm convs read what forward_response wrote on this stream and allocate on the side stream; the backward (this stream)
            # waits for them, so the side stream's blocks are never reused before this stream has read them (next tile: same order)
            if self._side is None:
                self._side = torch.cuda.Stream()
            fwd = torch.cuda.Event()
            fwd.record()
            with torch.cuda.stream(self._side):
                self._side.wait_event(fwd)
                with self.span("norm_convs"):                            # events on the side stream: this span overlaps the ones below
                    self.forward_norms(saved, top, top_first=True)
                norms_done = torch.cuda.Event()                          # the backward waits layer by layer (rec["ready"]); a tile that
                norms_done.record()                                      # ends before its backward waits here, before its tensors are freed
        with self.span("proposals"):
            rois_b, probs_b, kidx_b, num = ops.generate_proposals3d_batched(prob, deltas, det.anchors, float(c.stride), im_info, c.pre_nms_topN,
                                                                            c.post_nms_topN, c.rpn_nms_thresh, c.rpn_min_size)
            num_host = det._pinned_counts(num)
            # the RoI count reaches the pinned buffer through the kernel that also builds the row offsets box results needs
            _, _, offs_dev = ops.compact_rows2(rois_b, kidx_b, num, num_host)
            ready = torch.cuda.Event()
            ready.record()
        nl = len(saved)
        late = [i for i in (0, 1) if i < nl - 1]                      # conv1a / conv2a: the last layers the backward reaches
        if norms_done is None:
            with self.span("norm_convs"):
                self.forward_norms(saved, top, layers=[i for i in range(nl) if i not in late], cls=True)
        yield "rois"
        ready.synchronize()                                           # host wait 1 (covered by the norm convs above)
        R = int(num_host[0])
        det._release_counts(num_host)
        if R == 0:
            if norms_done is not None:
                torch.cuda.current_stream().wait_event(norms_done)
            return None                                               # nothing survives -> the reference returns five Nones (:190)
        rois, keep_idx = rois_b[0, :R], kidx_b[0, :R]
        with self.span("box_head"):
            cls, bbox, pred = det.box_head_outputs(feat, rois, clip_to=im_info[:3])                                # :121-122
            cb, ck, cnt = ops.box_results3d_batched(cls, pred, keep_idx, offs_dev, c.num_classes, c.score_thresh, c.nms,
                                                    c.detections_per_im, R)                                          # :124
            A = prob.shape[1]
            sel = ops.prm_select_peaks(cb[0, 1], ck[0, 1], cnt[0, 1:2], peak_threshold, A, prob.shape[-3:],          # :125,136-139,161-163
                                       prob=prob[0] if self.skip_dead_peaks else None)
        if norms_done is None:
            with self.span("norm_convs_late"):
                self.forward_norms(saved, top, layers=late, cls=False)
        yield "peaks"
        sel["event"].synchronize()                                    # host wait 2 (covered by the two norm convs above / the next tile)
        P = int(sel["host"]["num"][0])
        if P == 0:
            sel["release"]()
            if norms_done is not None:
                torch.cuda.current_stream().wait_event(norms_done)
            return None                                # no score above peak_threshold (:161-162 never true, :189-190)
        hp = sel["host"]["peaks"][:P]
        peaks = torch.from_numpy(np.concatenate((np.zeros((P, 1), np.int64), hp.astype(np.int64)), 1))          # (b,a,s,h,w), b = 0
        dets = torch.from_numpy(sel["host"]["dets"][:P].astype(np.float64))                                     # :163
        dead = None if sel["host"]["dead"] is None else (sel["host"]["dead"][:P] != 0)
        if dead is not None and dead.any():
            hp = np.array(hp, copy=True)                                   # the pinned mirror goes back to the pool below
        sel["release"]()
        nlive = P if dead is None else int(P - int(dead.sum()))
        with self.span("backward"):
            if dead is not None and dead.any():
                win, sums, origins = self._backward_live(hp, dead, saved, top, data)
            else:
                win, sums, origins = self.backward_windows(sel["peaks"][:P], saved, top, data)
        if nlive == 0 and norms_done is not None:
            # no peak was back-propagated, so nothing on this stream has waited for the side stream's norm convs (rec["ready"] is only
            # consumed inside backward_windows): wait here, as the R == 0 / P == 0 returns do, before the tile's tensors are freed
            torch.cuda.current_stream().wait_event(norms_done)
        out = dict(crm=prob, peaks=peaks, dets=dets, peaks_dev=sel["peaks"][:P], dets_dev=sel["dets"][:P], windows=win, sums=sums,
                   origins=origins, num_live=nlive)
        if dense:
            out["prms"] = ops.prm_scatter(win, sums, origins, (S, H, W))
        return out

    def _backward_live(self, peaks_host, dead, saved, top, data):
        """backward_windows for the peaks that are not `dead` (see skip_dead_peaks); the dead ones: zero windows, sum 0, and the window
        origin every layer's rule gives (prepare: (pool ? 2 o : o) - border, top to bottom) - what the kernels write for them."""
        P = len(peaks_host)
        o = peaks_host[:, 1:4].astype(np.int64)
        U = 1
        for rec in reversed(saved):
            border = 2 if rec["k"] == 5 else 1
            o = (2 * o if rec["pool"] else o) - border
            U = (2 if rec["pool"] else 1) * U + 2 * border
        live = np.nonzero(~dead)[0]
        dev = data.device
        win = torch.zeros((P, U, U, U), dtype=torch.float32, device=dev)
        sums = torch.zeros((P,), dtype=torch.float32, device=dev)
        origins = ops.upload(o.astype(np.int32), dev)
        if len(live):
            w, s_, og = self.backward_windows(ops.upload(np.ascontiguousarray(peaks_host[live]).astype(np.int32), dev), saved, top, data)
            assert tuple(w.shape[1:]) == (U, U, U)
            idx = ops.upload(live.astype(np.int64), dev)
            win.index_copy_(0, idx, w)
            sums.index_copy_(0, idx, s_)
        return win, sums, origins

    def _prm_tile_unfused(self, data, feat, prob, deltas, saved, top, peak_threshold, dense):
        """pre_nms_topN beyond the fused box kernels' capacity (or a model without a box head): the per-stage path with a host read
        per stage."""
        det, c = self.det, self.cfg
        S, H, W = data.shape[-3:]
        im_info = np.array([S, H, W, 1.0], np.float64)
        rois, probs, keep_idx = det.proposals(prob, deltas, im_info)
        if rois.shape[0] == 0:
            return None
        cls, bbox, pred = det.box_head_outputs(feat, rois, clip_to=im_info[:3])
        sc, bx, _, cls_keep = det.box_results_with_nms_and_limit(cls, pred, keep_idx)
        keep = cls_keep[1]
        A = prob.shape[1]
        s_, h_, w_ = prob.shape[-3:]
        a = keep % A
        pos = keep // A
        peaks = torch.stack([torch.zeros_like(a), a, pos // (h_ * w_), (pos // w_) % h_, pos % w_], 1)
        vidx = torch.nonzero(sc > peak_threshold).squeeze(1)
        if vidx.numel() == 0:
            return None
        peaks_v = peaks.index_select(0, vidx)
        dets32 = torch.cat([bx.index_select(0, vidx), sc.index_select(0, vidx).unsqueeze(1)], 1)
        pk32 = peaks_v[:, 1:].to(torch.int32).contiguous()
        win, sums, origins = self.backward_windows(pk32, saved, top, data)
        out = dict(crm=prob, peaks=peaks_v.cpu(), dets=dets32.double().cpu(), peaks_dev=pk32, dets_dev=dets32, windows=win, sums=sums,
                   origins=origins, num_live=int(pk32.shape[0]))
        if dense:
            out["prms"] = ops.prm_scatter(win, sums, origins, (S, H, W))
        return out



class TilePipeline:
    """(A/B option - measured no faster than one prm_tile call per tile, DESIGN.md "Round 4" 8.)  Tiles through PRMEngine.prm_tile,
    software-pipelined over the two host waits of a tile: tile k+1's forward and proposals are
    enqueued BEFORE the host waits for tile k's peak count, so that wait (and the building of the backward's launches behind it) has a
    millisecond of queued work in front of it, and tile k+1's RoI count has long arrived when the host asks for it.  Stream order:
    forward(k+1) proposals(k+1) | backward(k) [caller's post-processing of k] | box head(k+1) selection(k+1) | forward(k+2) ...
    Results come back in the order the tiles went in; two tiles' forward tensors are alive at a time.

        pipe = TilePipeline(engine, dense=False)
        for key, data in tiles:
            for k, out in pipe.push(key, data): ...      # 0, 1 or 2 finished tiles (out is None for an empty tile)
        for k, out in pipe.flush(): ...
    """

    def __init__(self, engine, peak_threshold=0.1, dense=False):
        self.engine, self.peak_threshold, self.dense = engine, peak_threshold, dense
        self.pending = None                                     # (key, generator suspended in front of the peak-count wait)

    @staticmethod
    def _advance(g):
        try:
            return False, next(g)
        except StopIteration as e:
            return True, e.value

    def _finish(self, g):
        while True:
            done, v = self._advance(g)
            if done:
                return v

    def push(self, key, data):
        g = self.engine.prm_tile_phases(data, self.peak_threshold, self.dense)
        done, v = self._advance(g)                              # forward + proposals enqueued (or: the per-stage path ran to its end)
        out = []
        if self.pending is not None:
            pk, pg = self.pending
            self.pending = None
            out.append((pk, self._finish(pg)))                  # previous tile: peak count -> backward
        if not done:
            done, v = self._advance(g)                          # RoI count (arrived long ago) -> box head + peak selection enqueued
        if done:
            out.append((key, v))
        else:
            self.pending = (key, g)
        return out

    def flush(self):
        if self.pending is None:
            return []
        pk, pg = self.pending
        self.pending = None
        return [(pk, self._finish(pg))]
