"""Data-parallel sharding of independent (volume, tile) work items, one process per GPU.

The reference's only multi-GPU inference story is one subprocess per GPU collated through pickle files
(lib/utils/my_subprocess.py:66-115, lib/core/test_engine.py:139-185).  Tiles never exchange halos
(lib/core/test.py:91-145): the one real exchange on the path is the concatenation of per-tile detections before
the cross-tile NMS (lib/core/test.py:150-160).  Here that is ONE all_gather of fixed-shape padded detections
[items_per_rank, DETECTIONS_PER_IM, 7] + counts (RCCL over xGMI with backend nccl; gloo on CPU for tests)."""
import torch


def partition(num_items, rank, world):
    """Static round-robin: item i belongs to rank i % world."""
    return list(range(rank, num_items, world))


def items_per_rank(num_items, world):
    return (num_items + world - 1) // world


def pack_detections(dets_list, cap, device=None):
    """dets_list: per local item a [n_i,7] tensor -> (padded [L,cap,7] fp32, counts [L] int32)."""
    L = len(dets_list)
    device = device or (dets_list[0].device if L else "cpu")
    padded = torch.zeros((L, cap, 7), dtype=torch.float32, device=device)
    counts = torch.zeros((L,), dtype=torch.int32, device=device)
    for i, d in enumerate(dets_list):
        n = min(int(d.shape[0]), cap)
        if n:
            padded[i, :n] = d[:n].to(torch.float32)
        counts[i] = n
    return padded, counts


def all_gather_detections(padded, counts, num_items, dist=None):
    """One collective for the boxes and one tiny one for the counts.  Returns a list of num_items [n_i,7]
    tensors in GLOBAL item order (item i was produced by rank i % world at local slot i // world)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [padded[i, :int(counts[i])] for i in range(min(num_items, padded.shape[0]))]
    world = dist.get_world_size()
    per = items_per_rank(num_items, world)
    if padded.shape[0] < per:       # ranks with one item fewer pad to the common shape
        pad = per - padded.shape[0]
        padded = torch.cat([padded, padded.new_zeros((pad,) + tuple(padded.shape[1:]))])
        counts = torch.cat([counts, counts.new_zeros((pad,))])
    gp = [torch.empty_like(padded) for _ in range(world)]
    gc = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(gp, padded.contiguous())
    dist.all_gather(gc, counts.contiguous())
    out = []
    for i in range(num_items):
        r, slot = i % world, i // world
        out.append(gp[r][slot, :int(gc[r][slot])])
    return out
