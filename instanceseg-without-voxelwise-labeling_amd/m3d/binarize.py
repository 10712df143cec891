"""Per-detection binarisation on device: uint8 PRM quantisation -> box crop + normalisation -> 2D-Otsu.

Counterpart of the per-detection loop of tools/binarization_soma.py:66-94 and tools/binarization_nuclei.py:92-124
up to (and including) `otsu_py_2d_fast` (`binarize_tile`), and the whole loop body including largest connected
component, hole filling, closing and label painting (binarization_soma.py:96-105, binarization_nuclei.py:125-150)
without a host round trip (`segment_tile`)."""
import numpy as np
import torch

from . import ops


def det_boxes_int(dets, tile_shape, mode):
    """Reference integer crop boxes.  soma: (det[:6] - tile offset).astype(int) (binarization_soma.py:78);
    nuclei additionally clamps to the tile (binarization_nuclei.py:98-104).  dets here are already in tile
    coordinates; astype(int) truncates toward zero."""
    b = np.asarray(dets)[:, :6].astype(np.float64).astype(np.int64)
    S, H, W = tile_shape
    if mode == "nuclei":
        b[:, 0] = np.maximum(0, b[:, 0]); b[:, 1] = np.maximum(0, b[:, 1]); b[:, 2] = np.maximum(0, b[:, 2])
        b[:, 3] = np.minimum(W - 1, b[:, 3]); b[:, 4] = np.minimum(H - 1, b[:, 4]); b[:, 5] = np.minimum(S - 1, b[:, 5])
    return b.astype(np.int32)


def binarize_tile(image_u16, prms, dets, mode="soma", max_gray_range=8192):
    """image_u16: raw uint16 tile [S,H,W] (CUDA); prms: float32 [P,S,H,W] peak response maps (CUDA, as returned by
    PRMEngine.prm_tile(dense=True)); dets: [P,7] in tile coordinates.
    Returns a list of (box int32[6], mask uint8 ndarray {0,255} of the box shape, k, b); detections whose PRM crop
    is empty (binarization_soma.py:74-76) or whose Otsu has no separating line are skipped (mask None)."""
    S, H, W = image_u16.shape
    boxes = det_boxes_int(dets.cpu().numpy() if torch.is_tensor(dets) else dets, (S, H, W), mode)
    ok = (boxes[:, 3] >= boxes[:, 0]) & (boxes[:, 4] >= boxes[:, 1]) & (boxes[:, 5] >= boxes[:, 2]) & (boxes[:, :3].min(1) >= 0) & \
         (boxes[:, 3] < W) & (boxes[:, 4] < H) & (boxes[:, 5] < S)
    q, _, _ = _quantised(prms, (S, H, W))                                   # what the reference reads back from its TIFFs
    idx = np.nonzero(ok)[0]
    out = [(boxes[i], None, 0, 0) for i in range(len(boxes))]
    if len(idx) == 0:
        return out
    bsel = torch.from_numpy(boxes[idx]).cuda()
    oi, op, offs = ops.roi_normalize(image_u16, q[torch.from_numpy(idx).cuda()].contiguous(), bsel, mode)
    mask, kb, status = ops.otsu2d_batch(oi, op, offs, max_gray_range)
    mask, kb, status, offs = mask.cpu().numpy(), kb.cpu().numpy(), status.cpu().numpy(), offs.cpu().numpy()
    for j, i in enumerate(idx):
        if status[j] != 0:
            continue
        x1, y1, z1, x2, y2, z2 = boxes[i]
        m = mask[offs[j]:offs[j + 1]].reshape(z2 - z1 + 1, y2 - y1 + 1, x2 - x1 + 1)
        out[i] = (boxes[i], m, int(kb[j, 0]), int(kb[j, 1]))
    return out


def _tile_instance_masks(image_u16, q, boxes, mode, max_gray_range, nonempty_all=None, win_origins=None, raw_status=False):
    """Device pipeline for the detections of one tile: crop + normalise -> 2D-Otsu -> largest 26-connected component
    (-> hole fill -> 6-closing for nuclei).  image_u16 [S,H,W] CUDA, q uint8 [P,S,H,W] CUDA (quantised PRMs; with win_origins int32
    [P,3]: their compact form [P,n,n,n], the maps being zero outside their windows - nothing dense is built),
    boxes int32 ndarray [P,6] inclusive tile coordinates.  Returns (masks uint8 flat, offsets, boxes CUDA int32 [n,6],
    idx LongTensor [n] (rows of `boxes` that were processed), ok bool [n] (False: the reference skips / fails))."""
    S, H, W = image_u16.shape
    dev = image_u16.device
    okb = (boxes[:, 3] >= boxes[:, 0]) & (boxes[:, 4] >= boxes[:, 1]) & (boxes[:, 5] >= boxes[:, 2]) & (boxes[:, :3].min(1) >= 0) & \
          (boxes[:, 3] < W) & (boxes[:, 4] < H) & (boxes[:, 5] < S)
    idx = np.nonzero(okb)[0]
    if len(idx) == 0:
        return None
    bh = boxes[idx]
    # every index table of the tile in ONE staged copy: the processed rows, their boxes, crop extents (ez, ey, ex), the map of each row, crop offsets
    dims_h = np.stack([bh[:, 5] - bh[:, 2] + 1, bh[:, 4] - bh[:, 1] + 1, bh[:, 3] - bh[:, 0] + 1], 1).astype(np.int32)
    idx_t, bsel, dims, midx, offs_d = ops.upload_packed([idx.astype(np.int64), bh.astype(np.int32), dims_h, idx.astype(np.int32), ops.crop_offsets(bh)], dev)
    # detections with a valid crop are a subset of the tile's peaks: the crop kernels read map idx[r] of the full stack (a gathered copy
    # of the maps was 200 MB per soma tile)
    sub = len(idx) != q.shape[0]
    oi, op, offs = ops.roi_normalize(image_u16, q, bsel, mode, boxes_host=bh, map_index=midx if sub else None, win_origins=win_origins, offsets=offs_d)
    mask, _, st_otsu = ops.otsu2d_batch(oi, op, offs, max_gray_range)
    cc, st_cc = ops.cc_largest_batch(mask, offs, dims, invert=False, tie_last=(mode == "soma"))
    if mode == "nuclei":
        cc, _ = ops.cc_largest_batch(cc, offs, dims, invert=True, tie_last=False)         # fill holes
        cc = ops.binary_closing6_batch(cc, offs, dims)
    if raw_status:                                   # segment_tile: the three conditions meet inside m3d_paint_begin
        return cc, offs, bsel, idx_t, (st_otsu, st_cc)
    nonempty = nonempty_all[idx_t] if nonempty_all is not None else (q[idx_t] if sub else q).reshape(len(idx), -1).amax(1) > 0   # binarization_soma.py:74-76
    return cc, offs, bsel, idx_t, (st_otsu == 0) & (st_cc == 0) & nonempty


def _quantised(prms, shape, compact=False, stats=False):
    """uint8 maps from dense float maps [P,S,H,W], or straight from the back-propagation's cone-cropped windows when `prms` is the
    triple (windows, sums, origins) of PRMEngine.prm_tile(dense=False) - the dense float maps are then never built.  compact (triple
    only): the uint8 WINDOWS [P,n,n,n] instead of dense uint8 maps (the crop kernels read them through the origins).
    Returns (maps, nonempty flags or None, window origins or None)."""
    if isinstance(prms, (tuple, list)):
        if compact:
            q, ne = ops.prm_quantize_windows_compact_u8(prms[0], prms[1], prms[2], shape, return_nonempty="stats" if stats else True)
            return q, ne, prms[2].contiguous()
        q, ne = ops.prm_quantize_windows_u8(prms[0], prms[1], prms[2], shape, return_nonempty=True)
        return q, ne, None
    return ops.prm_quantize_u8(prms), None, None


def segment_tile(image_u16, prms, dets, mode="soma", max_gray_range=8192, first_id=1):
    """The reference's per-detection loop body for one tile, entirely on device.
    image_u16 [S,H,W] uint16 CUDA, prms float32 [P,S,H,W] CUDA (or the (windows, sums, origins) triple), dets [P,7] in tile coordinates, ALREADY in the
    order the reference loops over them (soma: NMS-kept, score-descending, binarization_soma.py:56-61; nuclei: NMS
    order filtered by score > 0.4, binarization_nuclei.py:80-86).  Detection d gets mask id first_id + d whether or not
    it ends up painted (mask_id is incremented before the `continue`, binarization_soma.py:66).
    Returns (labels int32 [S,H,W] CUDA, painted bool [P] CUDA): labels holds, per voxel, the id of the first
    detection whose final mask covers it; painted[d] says whether id d occurs at all (`mask_id in np.unique(seg)`)."""
    S, H, W = image_u16.shape
    P = int((prms[0] if isinstance(prms, (tuple, list)) else prms).shape[0])
    dev = image_u16.device
    if P == 0:
        return torch.zeros((S, H, W), dtype=torch.int32, device=dev), torch.zeros((P,), dtype=torch.bool, device=dev)
    boxes = det_boxes_int(dets.cpu().numpy() if torch.is_tensor(dets) else dets, (S, H, W), mode)
    windows = isinstance(prms, (tuple, list))
    q, nonempty, worg = _quantised(prms, (S, H, W), compact=True, stats=windows)   # windows only: the crops are cut out of them directly
    r = _tile_instance_masks(image_u16, q, boxes, mode, max_gray_range, nonempty, win_origins=worg, raw_status=windows)
    if r is None:
        return torch.zeros((S, H, W), dtype=torch.int32, device=dev), torch.zeros((P,), dtype=torch.bool, device=dev)
    cc, offs, bsel, idx_t, ok = r
    if windows:                                       # sentinel fill + present flags + paint ids (skipped detections: -1) in one launch
        labels, present, ids = ops.paint_begin((S, H, W), first_id + P, ok[0], ok[1], nonempty, idx_t, first_id, dev)
    else:
        ids = (idx_t + first_id).to(torch.int32)
        ids = torch.where(ok, ids, torch.full_like(ids, -1))                               # skipped: never paints
        labels = torch.full((S, H, W), -1, dtype=torch.int32, device=dev)                   # 0xFFFFFFFF sentinel
        present = None
    ops.paint_instances_into(labels, cc, offs, bsel, ids)
    present = ops.paint_finish(labels, first_id + P - 1, present)[first_id:first_id + P]  # (torch.bincount would make the host wait here)
    return labels, present


def segment_tile_on(stream, image_u16, prms, dets, span=None, **kw):
    """segment_tile queued on `stream` behind everything the current stream holds; returns (labels, painted, done event) at once.
    The binarisation stage is a chain of small launches (a workgroup or a few per detection): on its own stream it runs beside the NEXT
    tile's convolutions instead of in front of them.  The inputs are marked as in use by `stream` (record_stream), so the allocator
    does not hand their blocks to the next tile before the stage has read them; the results belong to `stream` - wait for `done`
    (or synchronize) before reading them elsewhere."""
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(stream):
        stream.wait_event(ev)
        if span is not None:                                    # a HIP-event span (m3d.model.Probe) recorded on `stream`
            with span:
                labels, painted = segment_tile(image_u16, prms, dets, **kw)
        else:
            labels, painted = segment_tile(image_u16, prms, dets, **kw)
        done = torch.cuda.Event()
        done.record()
    for t in ([image_u16] + list(prms if isinstance(prms, (tuple, list)) else [prms]) + ([dets] if torch.is_tensor(dets) and dets.is_cuda else [])):
        t.record_stream(stream)
    return labels, painted, done


# ----------------------------------------------------------------------------- whole-volume drivers
def soma_tiles():
    """binarization_soma.py:42-52: (num, ss, hs, ws) of the fixed 3 x 2 x 2 tile grid (64 x 160 x 160 tiles)."""
    return [(s * 4 + h * 2 + w, s * 32, h * 96, w * 96) for s in range(3) for h in range(2) for w in range(2)]


def nuclei_tiles(height, width, patch_size=200, overlap=100):
    """binarization_nuclei.py:50-56."""
    widx = list(range(0, width - patch_size, patch_size - overlap)) + [width - patch_size]
    hidx = list(range(0, height - patch_size, patch_size - overlap)) + [height - patch_size]
    return [(ih * len(widx) + iw, 0, h, w) for ih, h in enumerate(hidx) for iw, w in enumerate(widx)]


def binarize_volume(img, tiles, dataset, nms_thresh=None, max_gray_range=8192):
    """Whole-volume counterpart of tools/binarization_soma.py:34-109 / tools/binarization_nuclei.py:36-154.
    img: the raw uint16 volume [S,H,W] (ndarray); tiles: {num: (dets float64 [n,7] tile coordinates, prm_u8 uint8
    [n,s,h,w])} as written by infer_simple's PRM branch (m3d.io.load_prm_instances / m3d.infer.infer_prm).
    Returns (seg uint16 ndarray [S,H,W], table float64): soma rows [mask_id, score] (:103-104); nuclei rows
    [mask_id, x1, y1, z1, x2, y2, z2, score] (:147-148)."""
    img = np.asarray(img)
    S, H, W = img.shape
    if dataset == "nuclei":                                                   # :44-45, on device, bit-exact with SciPy
        img_dev = ops.median_filter3_u16(ops.gaussian_filter_u16(torch.from_numpy(np.ascontiguousarray(img, np.uint16)).cuda(), 1.0))
        img = img_dev.cpu().numpy()
        grid, tshape = nuclei_tiles(H, W), (S, 200, 200)
        nms_thresh = 0.15 if nms_thresh is None else nms_thresh
    else:
        grid, tshape = soma_tiles(), (64, 160, 160)
        nms_thresh = 0.23 if nms_thresh is None else nms_thresh
    dets = np.empty((0, 7), np.float32 if dataset == "soma" else np.float64)
    inst = []                                                                 # (num, i, ws, hs, ss)
    for num, ss, hs, ws in grid:
        if num not in tiles or len(tiles[num][0]) == 0:
            continue
        off = np.array([ws, hs, ss, ws, hs, ss, 0], np.float32 if dataset == "soma" else np.int64)
        d = np.asarray(tiles[num][0]) + off
        dets = np.concatenate((dets, d), 0)
        if dataset == "soma":
            dets = dets.astype(np.float32)                                    # :52
        inst += [(num, i, ws, hs, ss) for i in range(len(d))]
    inst = np.array(inst, dtype=np.int64).reshape(-1, 5)
    if dataset == "nuclei":                                                   # :71-77 (condition2 really tests `width`)
        c1 = (dets[:, 0] > 10) & (dets[:, 3] < W - 10) & ((dets[:, 3] - dets[:, 0] + 1) < 32)
        c2 = (dets[:, 1] > 10) & (dets[:, 4] < W - 10) & ((dets[:, 4] - dets[:, 1] + 1) < 32)
        keep = ~(c1 | c2)
        dets, inst = dets[keep].astype(np.float32), inst[keep]
    empty_tab = np.zeros((0, 2 if dataset == "soma" else 8), np.float64)
    if len(dets) == 0:
        return np.zeros(img.shape, np.uint16), empty_tab
    keep = ops.nms3d(torch.from_numpy(np.ascontiguousarray(dets)).cuda(), nms_thresh, by_volume=(dataset == "nuclei")).cpu().numpy()
    dets, inst = dets[keep].copy(), inst[keep].copy()
    if dataset == "soma":
        order = np.argsort(dets[:, -1], kind="stable")[::-1]                  # :59 (tie rule: descending index)
    else:
        order = np.nonzero(dets[:, -1] > 0.4)[0]                              # :83-85
    dets, inst = dets[order], inst[order]
    n = len(dets)
    mask_ids = np.arange(1, n + 1)
    img_d = torch.from_numpy(np.ascontiguousarray(img.astype(np.uint16))).cuda()
    vol = torch.full((S, H, W), -1, dtype=torch.int32, device="cuda")
    boxes_glob = np.zeros((n, 6), np.int64)
    for num in np.unique(inst[:, 0]):
        sel = np.nonzero(inst[:, 0] == num)[0]
        _, _, ws, hs, ss = inst[sel[0]]
        off6 = np.array([ws, hs, ss, ws, hs, ss])
        ts, th, tw = tshape
        tile_img = img_d[ss:ss + ts, hs:hs + th, ws:ws + tw].contiguous()
        local = dets[sel, :6].astype(np.float64) - off6                       # soma :78, nuclei :97-98
        boxes = det_boxes_int(np.hstack((local, dets[sel, 6:7])), tuple(tile_img.shape), dataset)
        boxes_glob[sel] = boxes + off6
        q = torch.from_numpy(np.ascontiguousarray(np.asarray(tiles[num][1])[inst[sel, 1]])).cuda()
        r = _tile_instance_masks(tile_img, q, boxes, dataset, max_gray_range)
        if r is None:
            continue
        cc, offs, bsel, idx_t, ok = r
        ids = torch.from_numpy(mask_ids[sel]).to(idx_t.device)[idx_t].to(torch.int32)
        ids = torch.where(ok, ids, torch.full_like(ids, -1))
        gb = (bsel + torch.tensor([ws, hs, ss, ws, hs, ss], dtype=torch.int32, device=bsel.device)).contiguous()
        ops.paint_instances_into(vol, cc, offs, gb, ids)
    present = ops.paint_finish(vol, n)[1:n + 1].cpu().numpy()
    labels = vol
    seg = labels.cpu().numpy().astype(np.uint16)
    if dataset == "soma":
        table = np.array([[mask_ids[d], dets[d, -1]] for d in range(n) if present[d]], np.float64).reshape(-1, 2)
    else:
        table = np.array([[mask_ids[d], *boxes_glob[d], dets[d, -1]] for d in range(n) if present[d]], np.float64).reshape(-1, 8)
    return seg, table
