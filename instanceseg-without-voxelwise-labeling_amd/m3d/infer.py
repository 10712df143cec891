"""Counterparts of the reference's inference drivers on the HIP path.

  im_detect_all   lib/core/test.py:54-177  (detection mode: pad, tile, per-tile detect, offset, cross-tile NMS)
  infer_prm       tools/infer_simple.py:176-247 (PRM mode: per tile `instances/{num}/{ch}.tif` (LZW) + dets.npy, the tree
                  tools/binarization_*.py read; whole-volume binarisation: m3d.binarize.binarize_volume)
"""
import os

import numpy as np
import torch

from . import io as mio
from . import ops, tiling


def im_detect_all(det, im, patch=None, overlap=None, dist=None, tile_batch=4):
    """det: DetectorM3D; im: raw (S,H,W) volume (any dtype).  Returns cls_boxes_total: list (per class) of
    [n,7] float32 arrays (x1,y1,z1,x2,y2,z2,score) in volume coordinates, after the cross-tile nms_3d
    (core/test.py:159).  With `dist` initialised the tiles are sharded round-robin over ranks and their
    detections exchanged by one all_gather (m3d.shard)."""
    from . import shard
    c = det.cfg
    patch = patch or c.in_size                                                # TEST.IN_SIZE
    overlap = c.crop_ovlp if overlap is None else overlap                     # TEST.CROP_OVLP (core/test.py:87): 100 nuclei, 32 soma
    vol = tiling.norm1(np.asarray(im), np.float32).astype(np.float32)          # blob.py:179-184
    vol, pad_s = tiling.pad_slices(vol, patch[0])                             # core/test.py:79-86
    sidx, hidx, widx = tiling.detect_grid(c, vol.shape, patch, overlap)
    tiles = tiling.enumerate_tiles(sidx, hidx, widx)
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    dvol = torch.from_numpy(vol).cuda()
    local = []
    mine = shard.partition(len(tiles), rank, world)
    for j in range(0, len(mine), tile_batch):                                 # tiles are independent: `tile_batch` of them per pass
        grp = [tiles[i] for i in mine[j:j + tile_batch]]
        cubes = torch.stack([dvol[s:s + patch[0], h:h + patch[1], w:w + patch[2]] for _, s, h, w in grp])[:, None].contiguous()
        for (_, s, h, w), out in zip(grp, det.detect_batch(cubes)):
            d = out["cls_boxes"][1] if "cls_boxes" in out else torch.zeros((0, 7), device="cuda")
            off = torch.tensor([w, h, s - pad_s, w, h, s - pad_s, 0], dtype=torch.float32, device=d.device)   # :117-121,140-141
            local.append(d + off)
    allt = shard.all_gather_detections(local, c.detections_per_im, len(tiles), dist, device="cuda")   # ONE collective
    dets = torch.cat(allt, 0) if allt else torch.zeros((0, 7), device="cuda")
    keep = ops.nms3d(dets.contiguous(), c.nms)                                 # :159
    res = [np.zeros((0, 7), np.float32) for _ in range(c.num_classes)]
    res[1] = dets[keep].cpu().numpy()
    return res


def infer_prm(engine, im, dataset=None, patch=None, overlap=None, out_dir=None, peak_threshold=0.1):
    """engine: PRMEngine.  Returns a list of per-tile dicts {num, start, dets (float64 [P,7]), prm_u8 (list of
    uint8 volumes, slice padding removed)} for tiles that produced detections (infer_simple.py:209-247)."""
    c = engine.cfg
    patch = patch or c.in_size
    overlap = c.crop_ovlp if overlap is None else overlap                     # :197
    dataset = dataset or getattr(c, "dataset", "nuclei")                      # args.dataset (:197,201)
    vol = tiling.norm1(np.asarray(im), np.float64)                            # :180-183
    orig_slices = vol.shape[0]
    vol, pad_s = tiling.pad_slices(vol, patch[0])                             # :188-195
    sidx, hidx, widx = tiling.tile_grid(vol.shape, patch, overlap, dataset)   # :196-204
    results = []
    for num, s, h, w in tiling.enumerate_tiles(sidx, hidx, widx):
        crop = vol[s:s + patch[0], h:h + patch[1], w:w + patch[2]].copy().astype(np.float32)   # :217
        out = engine.prm_tile(torch.from_numpy(crop[None, None]).cuda(), peak_threshold=peak_threshold, dense=False)
        if out is None:
            continue                                                          # :225-226
        dets = out["dets"].cpu().numpy()
        # :233-238 on device, straight from the cone-cropped windows (no dense float maps), then 1 byte per voxel D2H
        q = ops.prm_quantize_windows_u8(out["windows"], out["sums"], out["origins"], crop.shape[-3:]).cpu().numpy()
        u8 = [q[ch][pad_s:pad_s + orig_slices] if pad_s else q[ch] for ch in range(q.shape[0])]   # :239-240
        rec = dict(num=num, start=(s, h, w), dets=dets, prm_u8=u8, peaks=out["peaks"].cpu().numpy())
        results.append(rec)
        if out_dir is not None:                                               # :213-216,246-247
            mio.save_prm_instances(os.path.join(out_dir, "instances", str(num)), u8, dets)   # {ch}.tif (LZW) + dets.npy
    return results
