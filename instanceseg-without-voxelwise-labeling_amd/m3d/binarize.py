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
    q = ops.prm_quantize_u8(prms)                                           # what the reference reads back from its TIFFs
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


def segment_tile(image_u16, prms, dets, mode="soma", max_gray_range=8192, first_id=1):
    """The reference's per-detection loop body for one tile, entirely on device.
    image_u16 [S,H,W] uint16/int32 CUDA, prms float32 [P,S,H,W] CUDA, dets [P,7] in tile coordinates, ALREADY in the
    order the reference loops over them (soma: NMS-kept, score-descending, binarization_soma.py:56-61; nuclei: NMS
    order filtered by score > 0.4, binarization_nuclei.py:80-86).  Detection d gets mask id first_id + d whether or not
    it ends up painted (mask_id is incremented before the `continue`, binarization_soma.py:66).
    Returns (labels int32 [S,H,W] CUDA, painted bool [P] CUDA): labels holds, per voxel, the id of the first
    detection whose final mask covers it; painted[d] says whether id d occurs at all (`mask_id in np.unique(seg)`)."""
    S, H, W = image_u16.shape
    P = int(prms.shape[0])
    dev = image_u16.device
    if P == 0:
        return torch.zeros((S, H, W), dtype=torch.int32, device=dev), torch.zeros((0,), dtype=torch.bool, device=dev)
    boxes = det_boxes_int(dets.cpu().numpy() if torch.is_tensor(dets) else dets, (S, H, W), mode)
    ok = (boxes[:, 3] >= boxes[:, 0]) & (boxes[:, 4] >= boxes[:, 1]) & (boxes[:, 5] >= boxes[:, 2]) & (boxes[:, :3].min(1) >= 0) & \
         (boxes[:, 3] < W) & (boxes[:, 4] < H) & (boxes[:, 5] < S)
    idx = np.nonzero(ok)[0]
    painted = torch.zeros((P,), dtype=torch.bool, device=dev)
    if len(idx) == 0:
        return torch.zeros((S, H, W), dtype=torch.int32, device=dev), painted
    q = ops.prm_quantize_u8(prms)
    idx_t = torch.from_numpy(idx).to(dev)
    bsel = torch.from_numpy(boxes[idx]).to(dev)
    oi, op, offs = ops.roi_normalize(image_u16, q[idx_t].contiguous(), bsel, mode)
    mask, _, st_otsu = ops.otsu2d_batch(oi, op, offs, max_gray_range)
    dims = torch.stack([bsel[:, 5] - bsel[:, 2] + 1, bsel[:, 4] - bsel[:, 1] + 1, bsel[:, 3] - bsel[:, 0] + 1], 1).to(torch.int32)
    cc, st_cc = ops.cc_largest_batch(mask, offs, dims, invert=False, tie_last=(mode == "soma"))
    if mode == "nuclei":
        cc, _ = ops.cc_largest_batch(cc, offs, dims, invert=True, tie_last=False)         # fill holes
        cc = ops.binary_closing6_batch(cc, offs, dims)
    ids = (idx_t + first_id).to(torch.int32)
    nonempty = q[idx_t].reshape(len(idx), -1).amax(1) > 0                                 # binarization_soma.py:74-76
    ids = torch.where((st_otsu == 0) & (st_cc == 0) & nonempty, ids, torch.full_like(ids, -1))   # skipped: never paints
    labels = ops.paint_instances(cc, offs, bsel, ids, (S, H, W))
    present = torch.bincount(labels.reshape(-1), minlength=first_id + P)[first_id:first_id + P] > 0
    return labels, present
