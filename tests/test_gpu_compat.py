"""GPU: the drop-in modules (reference dotted names) and the F.conv3d / F.linear interceptions."""
import sys

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def compat():
    import m3d.compat as c
    c.install()
    yield c
    c.uninstall_conv3d()
    c.uninstall_linear()


def test_reference_import_names(compat, golden):
    from utils.cython_nms_3d import nms_3d, nms_3d_volume
    from utils.cython_bbox_3d import bbox_overlaps_3d
    from modeling.roi_xfrom.roi_align_3d.functions.roi_align_3d import RoIAlignFunction_3d
    from model.roi_pooling.functions.roi_pool import RoIPoolFunction      # model_builder.py:11
    from model.roi_crop.functions.roi_crop import RoICropFunction          # model_builder.py:12
    g = golden("nms")
    assert np.array_equal(nms_3d(g["dets8"], np.float32(g["thr8"])), g["keep8"])
    assert nms_3d(g["dets8"], 0.3).dtype == np.int64
    with pytest.raises(ValueError):
        nms_3d(g["dets8"].astype(np.float64), 0.3)          # Cython buffer dtype check
    go = golden("overlaps")
    assert np.array_equal(bbox_overlaps_3d(go["boxes"], go["query"]), go["out"])
    f = torch.randn(1, 4, 6, 6, 6, device="cuda", requires_grad=True)
    rois = torch.tensor([[0, 2, 2, 2, 30, 30, 30.]], device="cuda")
    out = RoIAlignFunction_3d(7, 7, 7, 0.125, 2)(f, rois)
    assert np.allclose(out.detach().cpu().numpy(),
                       O.roi_align_3d_forward(f.detach().cpu().numpy(), rois.cpu().numpy(), 7, 7, 7, 0.125, 2), atol=1e-5)
    out.sum().backward()
    assert f.grad is not None and abs(f.grad.sum().item() - 4 * 343) < 1e-2
    with pytest.raises(NotImplementedError):
        RoIAlignFunction_3d(7, 7, 7, 0.125, 2)(f.detach().cpu(), rois.cpu())   # functions/roi_align_3d.py:31-32
    with pytest.raises(NotImplementedError):
        RoIPoolFunction(7, 7, 0.125)


def test_otsu_dropin(compat, golden):
    from otsu import otsu_py_2d_fast
    g = golden("otsu")
    m, k, b = otsu_py_2d_fast(g["img1"], g["prm1"])
    assert (k, b) == tuple(g["kb1"]) and np.array_equal(m, g["mask1"])


def test_conv3d_interception_runs_unmodified_module_code(compat):
    """nn.Conv3d modules (as lib/modeling/DSN.py builds them) hit the MFMA kernels through F.conv3d;
    the PRM-style pr_conv3d pattern (peak_backprop_3d.py:37-44) back-propagates through the dgrad kernel."""
    import torch.nn as nn
    import torch.nn.functional as F
    torch.manual_seed(0)
    conv = nn.Conv3d(8, 16, 3, 1, 1, bias=True).cuda()
    x = torch.randn(1, 8, 6, 10, 34, device="cuda", requires_grad=True)
    y = conv(x)
    assert F.conv3d is compat.conv3d
    ref = compat._orig_conv3d(x.detach().double(), conv.weight.detach().double(), conv.bias.detach().double(), 1, 1)
    assert (y.detach().double() - ref).abs().max().item() < 1e-5
    wpos = F.relu(conv.weight).detach()
    n = F.conv3d(x - x.min().detach(), wpos, None, 1, 1)
    g = torch.randn_like(n)
    n.backward(g)
    ref_gx = torch.nn.grad.conv3d_input(x.shape, wpos.double(), g.double(), 1, 1)
    assert (x.grad.double() - ref_gx).abs().max().item() / ref_gx.abs().max().item() < 5e-6
    # training-style backward: weight and bias gradients come from the wgrad / bias-grad kernels
    conv.zero_grad()
    x2 = torch.randn(2, 8, 6, 10, 34, device="cuda", requires_grad=True)
    y2 = conv(x2)
    g2 = torch.randn_like(y2)
    y2.backward(g2)
    xr = x2.detach().double().cpu().requires_grad_()
    wr = conv.weight.detach().double().cpu().requires_grad_()
    br = conv.bias.detach().double().cpu().requires_grad_()
    compat._orig_conv3d(xr, wr, br, 1, 1).backward(g2.double().cpu())
    for got, ref in ((x2.grad, xr.grad), (conv.weight.grad, wr.grad), (conv.bias.grad, br.grad)):
        assert (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item() < 1e-5
    # a non-qualifying call (stride 2) falls through to torch's own conv
    z = F.conv3d(x.detach(), conv.weight.detach(), None, 2, 1)
    assert z.shape[-1] == 17


def test_pack_cache_never_serves_another_layers_weights(compat):
    """Two same-shape nn.Conv3d run through the pr_conv3d pattern (fresh relu(W).detach() temporaries, freed between
    calls so the allocator re-uses their address) alternately: every result must match ITS layer's fp64 conv."""
    import torch.nn as nn
    import torch.nn.functional as F
    torch.manual_seed(1)
    convs = [nn.Conv3d(16, 16, 3, 1, 1, bias=True).cuda() for _ in range(2)]
    x = torch.randn(1, 16, 6, 10, 20, device="cuda")
    for it in range(6):
        conv = convs[it % 2]
        wpos = F.relu(conv.weight).detach()                  # same shape, usually the same recycled address
        n = F.conv3d(x, wpos, None, 1, 1)
        ref = compat._orig_conv3d(x.double(), wpos.double(), None, 1, 1)
        assert (n.double() - ref).abs().max().item() < 1e-5, it
        del wpos, n, ref
    # in-place update of a cached Parameter (optimizer step / load_state_dict): _version moves, the pack is rebuilt
    conv = convs[0]
    y0 = conv(x)
    with torch.no_grad():
        conv.weight.mul_(-0.5)
    y1 = conv(x)
    ref = compat._orig_conv3d(x.double(), conv.weight.detach().double(), conv.bias.detach().double(), 1, 1)
    assert (y1.double() - ref).abs().max().item() < 1e-5 and (y1 - y0).abs().max().item() > 1e-2
    # .data re-assignment keeps _version but moves the address
    conv.weight.data = torch.randn_like(conv.weight)
    ref = compat._orig_conv3d(x.double(), conv.weight.detach().double(), conv.bias.detach().double(), 1, 1)
    assert (conv(x).double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()      # N(0,1) weights: |y| ~ 40
    assert all(isinstance(k[0], int) for k in compat._pack_cache)          # only Parameters are cached
    n_before = len(compat._pack_cache)
    del conv, convs, y0, y1
    import gc
    gc.collect()
    assert len(compat._pack_cache) < n_before                                # weakref eviction


def test_stem_conv_with_input_grad_stays_on_the_hip_path(compat):
    """The reference's PRM mode: data.requires_grad_() (peak_response_mapping_3d.py:88) then conv1a (5^3, Cin=1, DSN.py:19)
    through pr_conv3d - forward AND backward-data must come from libm3d (no MIOpen fallback), x.grad vs torch's dgrad."""
    import torch.nn as nn
    import torch.nn.functional as F
    torch.manual_seed(2)
    conv = nn.Conv3d(1, 32, 5, 1, 2, bias=True).cuda()
    x = torch.randn(1, 1, 12, 21, 37, device="cuda", requires_grad=True)
    calls = []
    orig = compat._orig_conv3d
    compat._orig_conv3d = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        y = conv(x)
        wpos = F.relu(conv.weight).detach()
        n = F.conv3d(x - x.min().detach(), wpos, None, 1, 2)
        g = torch.randn_like(n)
        n.backward(g)
    finally:
        compat._orig_conv3d = orig
    assert not calls, "stem conv fell through to torch's conv"
    ref_y = orig(x.detach().double(), conv.weight.detach().double(), conv.bias.detach().double(), 1, 2)
    assert (y.detach().double() - ref_y).abs().max().item() < 1e-5
    ref_gx = torch.nn.grad.conv3d_input(x.shape, wpos.double(), g.double(), 1, 2)
    assert (x.grad.double() - ref_gx).abs().max().item() / ref_gx.abs().max().item() < 5e-6
    # plain autograd through the stem with a Parameter weight: dgrad + wgrad + bias grad
    conv.zero_grad(); x.grad = None
    y = conv(x)
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.detach().double().cpu().requires_grad_()
    wr = conv.weight.detach().double().cpu().requires_grad_()
    br = conv.bias.detach().double().cpu().requires_grad_()
    orig(xr, wr, br, 1, 2).backward(g.double().cpu())
    for got, ref in ((x.grad, xr.grad), (conv.weight.grad, wr.grad), (conv.bias.grad, br.grad)):
        assert (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item() < 1e-5


def test_linear_interception_runs_the_box_head_modules_unmodified(compat):
    """The reference's roi_2mlp_head / fast_rcnn_outputs are plain nn.Linear modules (fast_rcnn_heads.py:84-85,114-117,15-19,42-45):
    with the interception installed their F.linear calls run on libm3d's GEMMs (fc1 / fc2: bf16x3 split; cls / bbox: fp32 MFMA) -
    forward vs fp64, gradients vs torch's own, pack cache follows in-place weight updates and never crosses layers."""
    import torch.nn as nn
    import torch.nn.functional as F
    from m3d import ops
    torch.manual_seed(0)
    fc1, fc2, cls, box = nn.Linear(3424, 256).cuda(), nn.Linear(256, 256).cuda(), nn.Linear(256, 2).cuda(), nn.Linear(256, 12).cuda()
    x = torch.randn(77, 3424, device="cuda")
    calls = {"split": 0, "fp32": 0}
    orig_split, orig_lin = ops.SplitLinear.__call__, ops.linear

    def count_split(self, *a, **k):
        calls["split"] += 1
        return orig_split(self, *a, **k)

    def count_lin(*a, **k):
        calls["fp32"] += 1
        return orig_lin(*a, **k)
    ops.SplitLinear.__call__, ops.linear = count_split, count_lin
    try:
        h = F.relu(fc1(x), inplace=True)                      # the module code as the reference writes it (:114-115)
        h = F.relu(fc2(h), inplace=True)
        s, b = cls(h), box(h)
    finally:
        ops.SplitLinear.__call__, ops.linear = orig_split, orig_lin
    assert calls == {"split": 2, "fp32": 2}                     # fc1, fc2 on the split kernels; the two narrow heads on the fp32 kernel
    with torch.no_grad():
        hd = torch.relu(x.double() @ fc1.weight.double().t() + fc1.bias.double())
        hd = torch.relu(hd @ fc2.weight.double().t() + fc2.bias.double())
        sd, bd = hd @ cls.weight.double().t() + cls.bias.double(), hd @ box.weight.double().t() + box.bias.double()
    assert (s.double() - sd).abs().max().item() <= 1e-5 * sd.abs().max().item() and (b.double() - bd).abs().max().item() <= 1e-5 * bd.abs().max().item()
    # in-place update of the weight (optimizer step): the cached pack must not be served
    with torch.no_grad():
        fc2.weight.mul_(0.5)
    h1 = F.relu(fc1(x))
    got = fc2(h1)
    ref = h1.double() @ fc2.weight.double().t() + fc2.bias.double()
    assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    # autograd through the interception == torch's own linear
    lin = nn.Linear(128, 64).cuda()
    xg = torch.randn(36, 128, device="cuda", requires_grad=True)
    y = lin(xg)
    gy = torch.randn_like(y)
    y.backward(gy)
    compat.uninstall_linear()
    try:
        xr = xg.detach().clone().requires_grad_(True)
        lr = nn.Linear(128, 64).cuda()
        lr.load_state_dict(lin.state_dict())
        yr = lr(xr)
        yr.backward(gy)
    finally:
        compat.install_linear()
    for a, b_ in ((y, yr), (xg.grad, xr.grad), (lin.weight.grad, lr.weight.grad), (lin.bias.grad, lr.bias.grad)):
        assert torch.allclose(a.detach(), b_.detach(), rtol=1e-4, atol=1e-5 * float(b_.detach().abs().max()))
    # what does not qualify falls through to torch (3-D input, fp64)
    assert fc2(torch.randn(2, 3, 256, device="cuda")).shape == (2, 3, 256)
    assert F.linear(torch.randn(4, 8, device="cuda", dtype=torch.float64), torch.randn(5, 8, device="cuda", dtype=torch.float64)).dtype == torch.float64


def test_linear_interception_falls_back_for_unaligned_operands_and_packs_can_be_invalidated(compat):
    """F.linear interception: operands that libm3d's 16-byte loads cannot take (a view that starts 4 bytes into a buffer) go to
    torch's own linear instead of raising; after an in-place update through `.data` (which does not bump `_version`)
    `compat.invalidate_packs()` makes the next call repack."""
    import torch.nn.functional as F
    torch.manual_seed(3)
    lin = torch.nn.Linear(256, 128).cuda()
    x = torch.randn(40, 256, device="cuda")
    ref = x.double() @ lin.weight.double().t() + lin.bias.double()
    y = lin(x)
    assert (y.double() - ref).abs().max().item() < 1e-4
    buf = torch.randn(40 * 256 + 1, device="cuda")
    xu = buf[1:].view(40, 256)                                        # 4 bytes past a 16-byte boundary
    assert xu.data_ptr() % 16 != 0
    yu = F.linear(xu, lin.weight, lin.bias)
    refu = xu.double() @ lin.weight.double().t() + lin.bias.double()
    assert (yu.double() - refu).abs().max().item() < 1e-3             # torch's own GEMM (may use reduced-precision paths)
    lin.weight.data.mul_(2.0)                                         # `_version` unchanged: the cached pack is stale
    compat.invalidate_packs()
    y2 = lin(x)
    ref2 = x.double() @ lin.weight.double().t() + lin.bias.double()
    assert (y2.double() - ref2).abs().max().item() < 2e-4
