#!/usr/bin/env python3
"""End-to-end demo on synthetic data with seeded random weights (no checkpoints ship with the reference):
volume -> PRM-mode inference (per-tile instance tree on disk: {ch}.tif LZW + dets.npy, as tools/infer_simple.py
writes it) -> whole-volume binarisation (tools/binarization_{soma,nuclei}.py) -> uint16 label stack + table.

  python tools/run_pipeline.py --dataset soma --out /tmp/m3d_demo [--check]

--check re-runs the binarisation with the CPU oracle on the same tree and compares (slow; test use only)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
import numpy as np, torch
from m3d.config import Cfg
from m3d.model import DetectorM3D
from m3d.prm import PRMEngine
from m3d.synth import make_params, synth_volume, unsaturated_rpn
from m3d.infer import infer_prm
from m3d.binarize import binarize_volume
from m3d import io as mio


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="soma", choices=["soma", "nuclei"])
    ap.add_argument("--out", default="/tmp/m3d_demo")
    ap.add_argument("--max-peaks", type=int, default=0)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    if a.dataset == "soma":
        cfg, shape = Cfg.soma(), (128, 256, 256)
        P = make_params(stride=4, num_anchors=14, seed=0)
    else:
        cfg, shape = Cfg.nuclei(), (64, 300, 300)
        P = unsaturated_rpn(make_params(stride=8, num_anchors=35, seed=0))      # saturated RPN sigmoids would give 0 / 0 maps for every peak
    eng = PRMEngine(DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg))
    vol = synth_volume(0, shape)
    name = "img0"
    t0 = time.perf_counter()
    res = infer_prm(eng, vol, dataset=a.dataset, out_dir=os.path.join(a.out, "prm", name))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tiles = {}
    for r in res:
        d, p = mio.load_prm_instances(os.path.join(a.out, "prm", name, "instances", str(r["num"])))
        tiles[r["num"]] = (d, np.stack(p))
    t2 = time.perf_counter()
    seg, table = binarize_volume(vol, tiles, a.dataset)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    mio.save_segmentation(os.path.join(a.out, "seg"), name, seg, table)
    npk = sum(len(v[0]) for v in tiles.values())
    print("%s volume %s: %d tiles with detections, %d peaks; infer+write %.2f s, read tree %.2f s, binarise %.3f s; %d instances painted"
          % (a.dataset, shape, len(tiles), npk, t1 - t0, t2 - t1, t3 - t2, len(table)))
    if a.check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        seg_ref, table_ref = O.binarize_volume(vol, tiles, a.dataset)
        same = np.array_equal(seg, seg_ref) and np.array_equal(table, np.asarray(table_ref, np.float64))
        print("oracle check:", "identical" if same else "DIFFERENT (%d voxels)" % int((seg != seg_ref).sum()))
        sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
