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


def im_detect_all(det, im, patch=None, overlap=None, dist=None, tile_batch=4, device="cuda", nms_fn=None):
    """det: DetectorM3D; im: raw (S,H,W) volume (any dtype).  Returns cls_boxes_total: list (per class) of
    [n,7] float32 arrays (x1,y1,z1,x2,y2,z2,score) in volume coordinates, after the cross-tile nms_3d
    (core/test.py:159).  With `dist` initialised the tiles are sharded round-robin over ranks and their
    detections exchanged by ONE all_gather (m3d.shard) - entered by every rank, also by one that holds no tile.
    `device` / `nms_fn` exist for the multi-process CPU rehearsal of the sharding (tests/test_host_logic.py: gloo, a stub
    detector): the product path is device="cuda" with the library's NMS."""
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
    dvol = torch.from_numpy(vol).to(device)
    local = []
    mine = shard.partition(len(tiles), rank, world)
    for j in range(0, len(mine), tile_batch):                                 # tiles are independent: `tile_batch` of them per pass
        grp = [tiles[i] for i in mine[j:j + tile_batch]]
        cubes = torch.stack([dvol[s:s + patch[0], h:h + patch[1], w:w + patch[2]] for _, s, h, w in grp])[:, None].contiguous()
        for (_, s, h, w), out in zip(grp, det.detect_batch(cubes)):
            d = out["cls_boxes"][1] if "cls_boxes" in out else torch.zeros((0, 7), device=device)
            off = torch.tensor([w, h, s - pad_s, w, h, s - pad_s, 0], dtype=torch.float32, device=d.device)   # :117-121,140-141
            local.append(d + off)
    allt = shard.all_gather_detections(local, c.detections_per_im, len(tiles), dist, device=device)   # ONE collective
    dets = torch.cat(allt, 0) if allt else torch.zeros((0, 7), device=device)
    keep = (nms_fn or ops.nms3d)(dets.contiguous(), c.nms)                     # :159
    res = [np.zeros((0, 7), np.float32) for _ in range(c.num_classes)]
    res[1] = dets[keep].cpu().numpy()
    return res


def _device_volume(im, patch_s):
    """infer_simple.py:180-195 on the device: norm1 in float64 (mask = im > 0; (im - mean) / std, rounded to fp32 once - what
    `.astype(np.float32)` of the float64 crop gives, :217) and the edge-replicating slice pad.  Integer volumes go up as uint16 (half
    the bytes of fp32); anything else is normalised on the host in float64 exactly as the reference does.
    Returns (device fp32 volume [S',H,W], pad_s, original slice count)."""
    im = np.asarray(im)
    slices = int(im.shape[0])
    if im.dtype in (np.uint8, np.uint16):
        raw = ops.upload(np.ascontiguousarray(im.astype(np.uint16, copy=False)).view(np.uint16), "cuda")
        vol = ops.norm1(raw, f32_arith=False)
    else:                                                                     # float32 included: NumPy keeps mean / std / the quotient in the
        # array's own dtype there (:180-183 on a float32 ndarray is float32 pairwise arithmetic), which the host call reproduces as is
        vol = ops.upload(tiling.norm1(im, np.float64).astype(np.float32), "cuda")
    pad_s = 0
    if slices < patch_s:                                                      # :188-195
        pad_s = int((patch_s - slices) / 2)
        pad_e = patch_s - slices - pad_s
        vol = torch.cat([vol[:1].expand(pad_s, -1, -1), vol, vol[-1:].expand(pad_e, -1, -1)], 0).contiguous()
    return vol, pad_s, slices


def infer_prm_serial(engine, im, dataset=None, patch=None, overlap=None, out_dir=None, peak_threshold=0.1, device_norm=True):
    """The straightforward driver: one tile at a time, the dense uint8 maps quantised on the device, copied back and written before the
    next tile starts.  Kept as the statement `infer_prm` (pipelined) is tested against, file for file and byte for byte.
    Returns a list of per-tile dicts {num, start, dets (float64 [P,7]), prm_u8 (list of uint8 volumes, slice padding removed), peaks}
    for tiles that produced detections (infer_simple.py:209-247)."""
    c = engine.cfg
    patch = patch or c.in_size
    overlap = c.crop_ovlp if overlap is None else overlap                     # :197
    dataset = dataset or getattr(c, "dataset", "nuclei")                      # args.dataset (:197,201)
    if device_norm:
        dvol, pad_s, orig_slices = _device_volume(im, patch[0])
        shape = tuple(dvol.shape)
    else:
        vol = tiling.norm1(np.asarray(im), np.float64)                        # :180-183
        orig_slices = vol.shape[0]
        vol, pad_s = tiling.pad_slices(vol, patch[0])                         # :188-195
        shape = vol.shape
    sidx, hidx, widx = tiling.tile_grid(shape, patch, overlap, dataset)       # :196-204
    results = []
    for num, s, h, w in tiling.enumerate_tiles(sidx, hidx, widx):
        if device_norm:
            crop_t = dvol[s:s + patch[0], h:h + patch[1], w:w + patch[2]].contiguous()[None, None]
        else:
            crop_t = torch.from_numpy(vol[s:s + patch[0], h:h + patch[1], w:w + patch[2]].copy().astype(np.float32)[None, None]).cuda()   # :217
        out = engine.prm_tile(crop_t, peak_threshold=peak_threshold, dense=False)
        if out is None:
            continue                                                          # :225-226
        dets = out["dets"].cpu().numpy()
        # :233-238 on device, straight from the cone-cropped windows (no dense float maps), then 1 byte per voxel D2H
        q = ops.prm_quantize_windows_u8(out["windows"], out["sums"], out["origins"], crop_t.shape[-3:]).cpu().numpy()
        u8 = [q[ch][pad_s:pad_s + orig_slices] if pad_s else q[ch] for ch in range(q.shape[0])]   # :239-240
        rec = dict(num=num, start=(s, h, w), dets=dets, prm_u8=u8, peaks=out["peaks"].cpu().numpy())
        results.append(rec)
        if out_dir is not None:                                               # :213-216,246-247
            mio.save_prm_instances(os.path.join(out_dir, "instances", str(num)), u8, dets)   # {ch}.tif (LZW) + dets.npy
    return results


class _WriterPool:
    """Host side of the pipelined driver.  Two `drain` threads wait for a tile's device-to-host copy (an event wait: the GIL is free)
    and hand the tile to ONE foreign call that encodes and writes every `{ch}.tif` on `workers` C threads (m3d_tiff_write_window_stacks_u8;
    the GIL is free there too: a Python-level task per peak - future, buffer, file object - cost the launching thread more than the
    encoding cost the workers).  Pinned staging buffers are pooled by size."""

    def __init__(self, workers=None):
        from concurrent.futures import ThreadPoolExecutor
        if workers is None:
            try:
                workers = len(os.sched_getaffinity(0))
            except Exception:
                workers = os.cpu_count() or 1
            try:
                q, p = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    workers = min(workers, max(1, int(int(q) / int(p))))
            except Exception:
                pass
        self.workers = max(1, min(32, int(workers)))
        self.drain = ThreadPoolExecutor(max_workers=2)
        self.pinned = {}
        self.pending = []
        import threading
        self.lock = threading.Lock()

    def take(self, nbytes):
        size = 1 << max(20, int(nbytes - 1).bit_length())                    # power-of-two size classes: few pinning calls
        with self.lock:
            lst = self.pinned.setdefault(size, [])
            if lst:
                return lst.pop()
        return torch.empty((size,), dtype=torch.uint8).pin_memory()

    def give(self, buf):
        with self.lock:
            self.pinned.setdefault(int(buf.numel()), []).append(buf)

    def submit(self, fn, *a):
        self.pending.append(self.drain.submit(fn, *a))

    def finish(self):
        """Wait for every submitted task; the list is swapped out first, so a failed task is reported ONCE (the first error, after
        all the others have been waited for) and never again by a later, unrelated finish()."""
        pending, self.pending = self.pending, []
        first = None
        for f in pending:
            try:
                f.result()
            except BaseException as e:                                        # noqa: B902 - collected, re-raised below
                if first is None:
                    first = e
        if first is not None:
            raise first

    def close(self):
        self.finish()
        self.drain.shutdown()


_pool = None


def writer_pool():
    global _pool
    if _pool is None:
        _pool = _WriterPool()
    return _pool


def infer_prm(engine, im, dataset=None, patch=None, overlap=None, out_dir=None, peak_threshold=0.1, keep_maps=True, pool=None, tile_pipeline=False):
    """tools/infer_simple.py:176-247 for one volume, pipelined: the volume goes up once and is normalised (float64) and padded on the
    device; per tile the back-propagation's windows are quantised to uint8 ON THE DEVICE AS WINDOWS and copied to pinned host memory on
    a copy stream while the next tile computes; a thread pool rebuilds each peak's pages around its window, LZW-encodes and writes
    `instances/{num}/{ch}.tif` + `dets.npy` off the critical path.  Files are byte-identical to infer_prm_serial's.
    keep_maps=False: the returned records carry no dense `prm_u8` maps (files only; nothing dense ever exists on the host).
    tile_pipeline=True feeds the tiles through m3d.prm.TilePipeline (the next tile's forward in front of a tile's peak-count wait): it
    removes the last ~0.1 ms gap per tile and measured no faster (the forward then no longer runs beside the previous tile's small
    launches) - an A/B option."""
    c = engine.cfg
    patch = tuple(patch or c.in_size)
    overlap = c.crop_ovlp if overlap is None else overlap
    dataset = dataset or getattr(c, "dataset", "nuclei")
    pool = pool or writer_pool()
    dvol, pad_s, orig_slices = _device_volume(im, patch[0])
    sidx, hidx, widx = tiling.tile_grid(tuple(dvol.shape), patch, overlap, dataset)
    z_first, pages = (pad_s, orig_slices) if pad_s else (0, patch[0])         # :239-240
    copy_stream = engine.__dict__.setdefault("_copy_stream", torch.cuda.Stream())
    results = []

    def finish_tile(rec, ev, hbuf, P, wn, origins_h, tile_dir):
        try:
            ev.synchronize()                                                  # the windows have landed
            wins = hbuf.numpy()[:P * wn ** 3].reshape(P, wn, wn, wn)
            org = origins_h.copy()
            if tile_dir is not None:                                          # every `{ch}.tif` of the tile in one foreign call (C worker threads)
                mio.write_window_stacks_u8(tile_dir, wins, org, z_first, pages, patch[1], patch[2], threads=pool.workers)
                np.save(os.path.join(tile_dir, "dets.npy"), np.asarray(rec["dets"]))
            if keep_maps:
                rec["prm_u8"] = [mio.window_to_dense_u8(wins[ch], org[ch], z_first, pages, patch[1], patch[2]) for ch in range(P)]
        finally:
            pool.give(hbuf)                                                   # also when a write fails: the pinned buffer goes back

    def post(key, out):
        """a finished tile: quantise its windows, start their copy, hand the rest to the writer pool"""
        if out is None:
            return                                                            # :225-226
        num, s, h, w = key
        P, wn = int(out["windows"].shape[0]), int(out["windows"].shape[1])
        q = ops.prm_quantize_windows_compact_u8(out["windows"], out["sums"], out["origins"], patch)   # :233-238, as windows
        org_dev = out["origins"]
        done = torch.cuda.Event()
        done.record()
        hbuf = pool.take(P * wn ** 3 + 16 * P)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(done)
            hbuf[:P * wn ** 3].copy_(q.reshape(-1), non_blocking=True)
            horg = hbuf[P * wn ** 3:P * wn ** 3 + 12 * P].view(torch.int32).view(P, 3) if (P * wn ** 3) % 4 == 0 else None
            if horg is not None:
                horg.copy_(org_dev, non_blocking=True)
            landed = torch.cuda.Event()
            landed.record()
        q.record_stream(copy_stream)
        org_dev.record_stream(copy_stream)
        if horg is None:
            origins_h = org_dev.cpu().numpy()
        else:
            origins_h = horg.numpy()
        rec = dict(num=num, start=(s, h, w), dets=out["dets"].numpy() if out["dets"].device.type == "cpu" else out["dets"].cpu().numpy(),
                   peaks=out["peaks"].numpy() if out["peaks"].device.type == "cpu" else out["peaks"].cpu().numpy(),
                   peaks_back_propagated=int(out.get("num_live", P)))
        results.append(rec)
        tile_dir = None
        if out_dir is not None:                                               # :213-216
            tile_dir = os.path.join(out_dir, "instances", str(num))
            os.makedirs(tile_dir, exist_ok=True)
        pool.submit(finish_tile, rec, landed, hbuf, P, wn, origins_h, tile_dir)

    from .prm import TilePipeline
    pipe = TilePipeline(engine, peak_threshold=peak_threshold, dense=False) if tile_pipeline else None
    for num, s, h, w in tiling.enumerate_tiles(sidx, hidx, widx):
        crop = dvol[s:s + patch[0], h:h + patch[1], w:w + patch[2]].contiguous()[None, None]      # :217
        if pipe is None:
            post((num, s, h, w), engine.prm_tile(crop, peak_threshold=peak_threshold, dense=False))
        else:
            for key, out in pipe.push((num, s, h, w), crop):
                post(key, out)
    if pipe is not None:
        for key, out in pipe.flush():
            post(key, out)
    pool.finish()
    return results
