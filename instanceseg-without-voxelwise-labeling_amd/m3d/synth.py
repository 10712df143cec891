"""Synthetic inputs for benchmarks and smoke tests: seeded weights with the reference's state-dict key
layout and nuclei-style volumes.  No dataset or checkpoint ships with the reference (README.md:31)."""
import math

import numpy as np
import torch

from .model import dsn_layers


def make_params(stride=8, num_anchors=35, mlp_dim=1024, roi_res=7, num_classes=2, seed=0, head=True):
    """kaiming-normal conv/linear, biases N(0,0.1), BN gamma U[0.5,1.5], beta N(0,0.1), mean N(0,0.1), var U[0.5,1.5]."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    chans = {"conv1a": (1, 32, 5), "conv2a": (32, 64, 3), "conv2b": (64, 64, 3), "conv3a": (64, 128, 3),
             "conv3b": (128, 128, 3), "conv4a": (128, 256, 3), "conv4b": (256, 256, 3)}

    def conv(name, cin, cout, k, scale=1.0):
        P[name + ".weight"] = torch.randn(cout, cin, k, k, k, generator=g) * math.sqrt(2.0 / (cin * k ** 3)) * scale
        P[name + ".bias"] = torch.randn(cout, generator=g) * 0.1

    def lin(name, cin, cout, scale=1.0):
        P[name + ".weight"] = torch.randn(cout, cin, generator=g) * math.sqrt(2.0 / cin) * scale
        P[name + ".bias"] = torch.randn(cout, generator=g) * 0.1

    for cname, bname, _ in dsn_layers(stride):
        cin, cout, k = chans[cname]
        conv("Conv_Body." + cname, cin, cout, k)
        P["Conv_Body.%s.weight" % bname] = torch.rand(cout, generator=g) + 0.5
        P["Conv_Body.%s.bias" % bname] = torch.randn(cout, generator=g) * 0.1
        P["Conv_Body.%s.running_mean" % bname] = torch.randn(cout, generator=g) * 0.1
        P["Conv_Body.%s.running_var" % bname] = torch.rand(cout, generator=g) + 0.5
    dim = 256 if stride == 8 else 128
    conv("RPN.RPN_conv", dim, dim, 3)
    conv("RPN.RPN_cls_score", dim, num_anchors, 1, scale=2.0)
    conv("RPN.RPN_bbox_pred", dim, num_anchors * 6, 1, scale=0.2)
    if head:
        lin("Box_Head.fc1", dim * roi_res ** 3, mlp_dim)
        lin("Box_Head.fc2", mlp_dim, mlp_dim)
        lin("Box_Outs.cls_score", mlp_dim, num_classes)
        lin("Box_Outs.bbox_pred", mlp_dim, 6 * num_classes, scale=0.3)
    return P


def unsaturated_rpn(P, scale=0.25):
    """A copy of `P` with RPN_cls_score's weight and bias scaled.  The random init of the stride-8 net gives RPN class logits of +-40: the
    sigmoid of every top-ranked position is exactly 1.0f, its derivative (1 - y) y exactly 0, and the reference's peak back-propagation then
    returns 0 / 0 maps for every kept peak (lib/prm/peak_response_mapping_3d.py:164-171).  Logits of +-10, as a trained net has, keep the
    proposals' ranking (a monotone function of the same logits, up to the former ties) and give the PRM workloads maps that are not empty."""
    P = dict(P)
    for k in ("RPN.RPN_cls_score.weight", "RPN.RPN_cls_score.bias"):
        P[k] = P[k] * scale
    return P


def synth_volume(i, shape=(128, 128, 128)):
    """N(100,10) background + 40 isotropic Gaussian blobs, clipped to uint16; returns the raw uint16 volume."""
    rng = np.random.RandomState(1234 + i)
    S, H, W = shape
    v = rng.normal(100, 10, shape).astype(np.float32)
    zz, yy, xx = np.mgrid[0:S, 0:H, 0:W].astype(np.float32)
    for _ in range(40):
        c = rng.uniform(0, 1, 3) * np.array(shape)
        s = rng.uniform(4, 8)
        a = rng.uniform(300, 900)
        r = int(4 * s)
        z0, z1 = max(0, int(c[0]) - r), min(S, int(c[0]) + r + 1)
        y0, y1 = max(0, int(c[1]) - r), min(H, int(c[1]) + r + 1)
        x0, x1 = max(0, int(c[2]) - r), min(W, int(c[2]) + r + 1)
        d2 = (zz[z0:z1, y0:y1, x0:x1] - c[0]) ** 2 + (yy[z0:z1, y0:y1, x0:x1] - c[1]) ** 2 + (xx[z0:z1, y0:y1, x0:x1] - c[2]) ** 2
        v[z0:z1, y0:y1, x0:x1] += a * np.exp(-d2 / (2 * s * s))
    return np.clip(v, 0, 65535).astype(np.uint16)
