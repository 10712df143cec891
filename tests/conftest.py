import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs /root/reference (build container only)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def cfg0_params(mlp_dim=1024, seed=0):
    """Weights of the configs[0] fixture (tests/golden/gen_cfg0.py): the bench weights with the RPN class scores de-saturated.
    The synthetic kaiming x2 cls weights push thousands of sigmoid outputs to the same float (0.9999999, 1.0), and NumPy's
    unstable argsort()[::-1] leaves the order of equal scores unspecified (SURVEY 8c caveat i): a bit-exact fixture needs
    distinct scores near the top."""
    from m3d.synth import make_params
    P = make_params(stride=8, num_anchors=35, mlp_dim=mlp_dim, seed=seed)
    P["RPN.RPN_cls_score.weight"] = P["RPN.RPN_cls_score.weight"] * 0.02
    P["RPN.RPN_cls_score.bias"] = P["RPN.RPN_cls_score.bias"] - 1.0
    return P
