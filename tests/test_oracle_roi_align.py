"""CPU: known-answer tests for the RoIAlign3D restatement.  The reference kernel is CUDA-only
(roi_align_kernel_3d.cu) and cannot run here: these properties are what pins the oracle."""
import numpy as np

import oracle as O


def _affine(S, H, W, a=0.5, b=-0.25, c=2.0, d=1.0):
    z, y, x = np.mgrid[0:S, 0:H, 0:W].astype(np.float32)
    return (a * z + b * y + c * x + d).astype(np.float32)


def test_constant_feature():
    f = np.full((1, 3, 6, 7, 8), 2.5, np.float32)
    rois = np.array([[0, 4, 6, 3, 40, 44, 30]], np.float32)
    out = O.roi_align_3d_forward(f, rois, 7, 7, 7, 0.125, 2)
    assert out.shape == (1, 3, 7, 7, 7) and np.allclose(out, 2.5, atol=1e-6)


def test_affine_feature_gives_bin_centres_in_quirk_layout():
    S, H, W = 10, 12, 14
    f = _affine(S, H, W)[None, None]
    x1, y1, z1, x2, y2, z2 = 8., 16., 8., 64., 72., 48.   # *0.125 -> inside [1, dim-2]
    rois = np.array([[0, x1, y1, z1, x2, y2, z2]], np.float32)
    AS = AH = AW = 7
    out = O.roi_align_3d_forward(f, rois, AS, AH, AW, 0.125, 2)[0, 0]
    sc = 0.125
    for i in range(7):          # out[r,c,i,j,k] is bin (y=i, x=j, z=k): memory order (ph,pw,ps)
        for j in range(7):
            for k in range(7):
                zc = z1 * sc + (k + 0.5) * (z2 - z1) * sc / AS
                yc = y1 * sc + (i + 0.5) * (y2 - y1) * sc / AH
                xc = x1 * sc + (j + 0.5) * (x2 - x1) * sc / AW
                assert abs(out[i, j, k] - (0.5 * zc - 0.25 * yc + 2.0 * xc + 1.0)) < 1e-4


def test_roi_outside_is_zero_and_malformed_is_one_wide():
    f = np.random.RandomState(0).rand(1, 2, 6, 6, 6).astype(np.float32)
    out = O.roi_align_3d_forward(f, np.array([[0, 400, 400, 400, 480, 480, 480]], np.float32), 3, 3, 3, 0.125, 2)
    assert np.all(out == 0)
    a = O.roi_align_3d_forward(f, np.array([[0, 16, 16, 16, 8, 8, 8]], np.float32), 2, 2, 2, 0.125, 2)   # x2 < x1
    b = O.roi_align_3d_forward(f, np.array([[0, 16, 16, 16, 24, 24, 24]], np.float32), 2, 2, 2, 0.125, 2)  # 1 wide
    assert np.array_equal(a, b)


def test_adaptive_grid_ratio0_matches_explicit_ratio():
    f = np.random.RandomState(1).rand(1, 2, 8, 8, 8).astype(np.float32)
    rois = np.array([[0, 0, 0, 0, 47.9, 47.9, 47.9]], np.float32)   # roi 5.99 / 3 bins -> ceil = 2
    assert np.array_equal(O.roi_align_3d_forward(f, rois, 3, 3, 3, 0.125, 0),
                          O.roi_align_3d_forward(f, rois, 3, 3, 3, 0.125, 2))


def test_batch_index_and_backward_quirk():
    rs = np.random.RandomState(2)
    f = rs.rand(2, 2, 5, 6, 7).astype(np.float32)
    rois = np.array([[1, 4, 4, 4, 40, 36, 30], [0, 0, 0, 0, 20, 20, 20]], np.float32)
    out = O.roi_align_3d_forward(f, rois, 2, 3, 4, 0.125, 2)
    single = O.roi_align_3d_forward(f[1:2], np.array([[0, 4, 4, 4, 40, 36, 30]], np.float32), 2, 3, 4, 0.125, 2)
    assert np.array_equal(out[0], single[0])
    # backward: total mass conservation (every in-range sample spreads weight 1/count over 8 corners)
    top = rs.rand(2, 2, 2, 3, 4).astype(np.float32)
    g = O.roi_align_3d_backward(top, rois, f.shape, 2, 3, 4, 0.125, 2)
    assert g.shape == f.shape and abs(g.sum() - top.sum()) < 1e-3
    # the backward reads top_diff with a different permutation from the forward (not the true adjoint)
    lin = (O.roi_align_3d_forward(f, rois, 2, 3, 4, 0.125, 2) * top).sum()
    assert abs((g * f).sum() - lin) > 1e-4
