#!/usr/bin/env python3
"""Per-op timings for the non-conv kernels (HIP events), with the roofline each one is bounded by:
RoIAlign3D (HBM write), NMS-3D / proposals (latency: report us vs the CPU figure), IoU, Otsu-2D (HBM/L2), maxpool (HBM)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, m3d
from m3d.config import Cfg

HBM = 8.0e12


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


rs = np.random.RandomState(0)
# ---- RoIAlign3D: R=1000, C=256, 16^3 feature map (config[2] upper bound)
feat = torch.randn(1, 256, 16, 16, 16, device="cuda")
for R in (332, 1000):
    c = rs.uniform(10, 118, (R, 3)); s = rs.uniform(10, 50, (R, 3))
    rois = torch.from_numpy(np.hstack((np.zeros((R, 1)), c - s / 2, c + s / 2)).astype(np.float32)).cuda()
    for name, ex in (("separable", False), ("exact-order", True)):
        t = timeit(lambda: m3d.roi_align3d_forward(feat, rois, 7, 7, 7, 0.125, 2, exact=ex))
        by = R * 256 * 343 * 4
        print("roi_align3d %-11s R=%4d C=256: %8.1f us  %6.1f MB written  %6.2f TB/s  (%.1f%% of 8 TB/s HBM)" %
              (name, R, t * 1e6, by / 1e6, by / t / 1e12, by / t / HBM * 100))
# ---- NMS-3D
import oracle as O
for N in (300, 1000, 4000, 16000):
    c = rs.uniform(0, 128, (N, 3)); s = rs.uniform(8, 40, (N, 3))
    dets = np.hstack((c - s / 2, c + s / 2, rs.permutation(N)[:, None] / N)).astype(np.float32)
    d = torch.from_numpy(dets).cuda()
    t = timeit(lambda: m3d.nms3d(d, 0.15), reps=10)
    t0 = time.perf_counter(); O.nms_3d(dets, 0.15); tc = time.perf_counter() - t0
    print("nms3d N=%5d: %8.1f us (incl. count readback)   CPU oracle %8.1f us   x%.1f" % (N, t * 1e6, tc * 1e6, tc / t))
# ---- IoU matrix
a = torch.rand(4000, 6, device="cuda") * 50; a[:, 3:] += a[:, :3]
q = torch.rand(500, 6, device="cuda") * 50; q[:, 3:] += q[:, :3]
t = timeit(lambda: m3d.bbox_overlaps3d(a, q))
print("bbox_overlaps3d 4000x500: %8.1f us  %.2f G pairs/s" % (t * 1e6, 4000 * 500 / t / 1e9))
# ---- proposals (128^3 volume: A=35, 16^3)
cfg = Cfg.nuclei()
sc = torch.rand(35, 16, 16, 16, device="cuda"); dl = torch.randn(210, 16, 16, 16, device="cuda") * 0.2
t = timeit(lambda: m3d.generate_proposals3d(sc, dl, cfg.anchors, 8., np.array([128., 128., 128., 1.]), 1000, 1000, 0.15), reps=10)
print("generate_proposals3d 143360 anchors -> top1000 -> NMS: %8.1f us" % (t * 1e6))
cfg = Cfg.soma()
sc = torch.rand(14, 16, 40, 40, device="cuda"); dl = torch.randn(84, 16, 40, 40, device="cuda") * 0.2
t = timeit(lambda: m3d.generate_proposals3d(sc, dl, cfg.anchors, 4., np.array([64., 160., 160., 1.]), 1000, 1000, 0.23), reps=10)
print("generate_proposals3d 358400 anchors -> top1000 -> NMS: %8.1f us" % (t * 1e6))
# ---- maxpool
x = torch.randn(1, 64, 64, 64, 64, device="cuda")
t = timeit(lambda: m3d.maxpool3d_2x(x))
print("maxpool3d_2x 64x64^3: %8.1f us  %.2f TB/s of 8 (reads+writes %.0f MB)" % (t * 1e6, x.numel() * 4.5 / t / 1e12, x.numel() * 4.5 / 1e6))
# ---- Otsu: 300 RoIs of ~30^3
imgs, prms, shapes = [], [], []
for i in range(300):
    shp = tuple(rs.randint(20, 40, 3))
    zz, yy, xx = np.mgrid[0:shp[0], 0:shp[1], 0:shp[2]]
    r = np.sqrt((zz - shp[0] / 2) ** 2 + (yy - shp[1] / 2) ** 2 + (xx - shp[2] / 2) ** 2)
    img = (600 * np.exp(-(r / 8) ** 2) + 100 + rs.randn(*shp) * 15).clip(0, 65535).astype(np.uint16)
    prm = (255 * np.exp(-(r / 7) ** 2)).astype(np.uint8)
    a_, b_ = O.normalize_soma(img, prm)
    imgs.append(a_.ravel()); prms.append(b_.ravel()); shapes.append(shp)
offs = torch.from_numpy(np.concatenate(([0], np.cumsum([a_.size for a_ in imgs]))).astype(np.int64)).cuda()
I = torch.from_numpy(np.concatenate(imgs)).cuda(); Pm = torch.from_numpy(np.concatenate(prms)).cuda()
t = timeit(lambda: m3d.otsu2d_batch(I, Pm, offs, 1024), reps=10)
V = I.numel()
t0 = time.perf_counter(); O.otsu_py_2d_fast(imgs[0].reshape(-1, 1, 1), prms[0].reshape(-1, 1, 1)); tc = time.perf_counter() - t0
print("otsu2d_batch 300 RoIs (%.1f Mvoxel): %8.1f us  %.0f RoIs/s  %.2f TB/s algorithmic (2x2B read x2 + 1B written)  [oracle C: %.0f us per RoI]" %
      (V / 1e6, t * 1e6, 300 / t, V * 9 / t / 1e12, tc * 1e6))
# ---- largest CC / hole fill / closing on the Otsu masks of those 300 RoIs
mask, _, _ = m3d.otsu2d_batch(I, Pm, offs, 1024)
dims = torch.from_numpy(np.array(shapes, np.int32)).cuda()
t1 = timeit(lambda: m3d.cc_largest_batch(mask, offs, dims, invert=False, tie_last=True), reps=10)
cc, _ = m3d.cc_largest_batch(mask, offs, dims, invert=False, tie_last=False)
t2 = timeit(lambda: m3d.cc_largest_batch(cc, offs, dims, invert=True, tie_last=False), reps=10)
t3 = timeit(lambda: m3d.binary_closing6_batch(cc, offs, dims), reps=10)
m0 = mask[:int(offs[1])].cpu().numpy().reshape(shapes[0])
t0 = time.perf_counter(); O.largest_cc_soma(m0); tc = time.perf_counter() - t0
print("cc_largest_batch 300 RoIs (%.1f Mvoxel): foreground %8.1f us, complement (hole fill) %8.1f us, closing6 %8.1f us  "
      "[scipy label per RoI: %.0f us]" % (V / 1e6, t1 * 1e6, t2 * 1e6, t3 * 1e6, tc * 1e6))
# ---- volume pre-filters of binarization_nuclei.py:44-45 (uint16, 64 x 1024 x 1024)
from scipy import ndimage
vol16 = torch.from_numpy(rs.randint(0, 4000, (64, 1024, 1024)).astype(np.uint16)).cuda()
tg = timeit(lambda: m3d.gaussian_filter_u16(vol16, 1.0), reps=5)
gv = m3d.gaussian_filter_u16(vol16, 1.0)
tm = timeit(lambda: m3d.median_filter3_u16(gv), reps=5)
small = vol16[:16, :256, :256].cpu().numpy()
t0 = time.perf_counter(); ndimage.median_filter(ndimage.gaussian_filter(small, sigma=1), size=3); tc = (time.perf_counter() - t0) * 64
print("prefilters 64x1024x1024 uint16: gaussian(sigma 1) %8.1f us  median 3^3 %8.1f us   [SciPy on one host core, extrapolated from 1/64 of the volume: %.1f s]" %
      (tg * 1e6, tm * 1e6, tc))
