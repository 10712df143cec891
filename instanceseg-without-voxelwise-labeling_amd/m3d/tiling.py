"""Host-side volume pre-processing and tiling of the inference drivers (pure index logic, no device work).

Reference: tools/infer_simple.py:180-212 (PRM branch), lib/core/test.py:76-90,140-145 (detection branch),
lib/utils/blob.py:179-184 (norm1)."""
import numpy as np


def norm1(im, dtype=np.float64):
    """mask = im > 0; (im - mean(im[mask])) / std(im[mask]).  float64 in infer_simple's PRM branch (:180-183),
    float32 in prep_im_for_blob (blob.py:179-184)."""
    if dtype == np.float32:
        im = im.astype(np.float32, copy=False)
    mask = im > 0
    mean_val = np.mean(im[mask])
    std_val = np.std(im[mask])
    return (im - mean_val) / std_val


def pad_slices(im, patch_s):
    """Edge-replicate along the slice axis up to patch_s (infer_simple.py:188-195, core/test.py:79-86).
    Returns (padded volume, pad_s)."""
    slices = im.shape[0]
    if slices < patch_s:
        pad_s = int((patch_s - slices) / 2)
        pad_e = patch_s - slices - pad_s
        im = np.append(np.tile(im[0, :, :], (pad_s, 1, 1)), im, axis=0)
        im = np.append(im, np.tile(im[-1, :, :], (pad_e, 1, 1)), axis=0)
        return im, pad_s
    return im, 0


def tile_starts(dim, patch, overlap):
    """list(range(0, dim - patch, patch - overlap)) + [dim - patch]  (infer_simple.py:198-200).
    Like the reference, assumes dim >= patch (a negative start otherwise, core/test.py:88-90)."""
    return list(range(0, dim - patch, patch - overlap)) + [dim - patch]


def tile_grid(shape, patch, overlap, dataset="nuclei"):
    """Tile start lists (sidx, hidx, widx) for a (padded) volume shape.  'soma' uses the fixed grid of
    infer_simple.py:201-204."""
    if dataset == "soma":
        return [0, 32, 64], [0, 96], [0, 96]
    return (tile_starts(shape[0], patch[0], overlap), tile_starts(shape[1], patch[1], overlap),
            tile_starts(shape[2], patch[2], overlap))


def detect_grid(cfg, shape, patch=None, overlap=None):
    """Detection-mode tile starts for a (padded) volume: TEST.IN_SIZE / TEST.CROP_OVLP from the config
    (core/test.py:77,87-90; CROP_OVLP is 100 in the nuclei YAML, the default 32 under the soma YAML, core/config.py:250)."""
    patch = tuple(patch or cfg.in_size)
    overlap = cfg.crop_ovlp if overlap is None else overlap
    return tile_grid(shape, patch, overlap)


def enumerate_tiles(sidx, hidx, widx):
    """(num, s, h, w) with num = iss*len_w*len_h + ih*len_w + iw  (infer_simple.py:209-212)."""
    out = []
    for iss, s in enumerate(sidx):
        for ih, h in enumerate(hidx):
            for iw, w in enumerate(widx):
                out.append((iss * len(widx) * len(hidx) + ih * len(widx) + iw, s, h, w))
    return out


def quantize_u8(fm):
    """(fm - min) / max * 255 -> uint8, in place order of infer_simple.py:234-238."""
    fm = np.array(fm, copy=True)
    fm -= np.min(fm)
    fm /= np.max(fm)
    fm *= 255.
    return fm.astype(np.uint8)
