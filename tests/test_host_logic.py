"""CPU: host-side driver logic (tiling, sharding) against the golden index lists and a 2-rank gloo run."""
import os
import socket

import numpy as np
import pytest
import torch


def test_tiling_matches_reference_lists(golden):
    """Product host logic (m3d.tiling) against the reference's own tiling statements exec'd on synthetic volumes
    (tests/golden/gen_tiling.py): padding, tile starts of both drivers and both dataset rules, tile ids, norm1 + padding values."""
    from m3d import tiling
    g = golden("tiling")
    for i in range(int(g["n"])):
        shape, patch, ov, ds = g["shape%d" % i], tuple(int(v) for v in g["patch%d" % i]), int(g["ov%d" % i]), str(g["ds%d" % i])
        im, pad_s = tiling.pad_slices(np.zeros(shape, np.float32), patch[0])
        assert pad_s == int(g["pad%d" % i]) and tuple(im.shape) == tuple(g["pshape%d" % i]) == tuple(g["d_pshape%d" % i])
        s, h, w = tiling.tile_grid(im.shape, patch, ov, ds)                                  # infer_simple.py:196-204
        assert (list(s), list(h), list(w)) == (list(g["s%d" % i]), list(g["h%d" % i]), list(g["w%d" % i]))
        assert [list(t) for t in tiling.enumerate_tiles(s, h, w)] == g["nums%d" % i].tolist()   # :209-212
        ds_, dh, dw = tiling.tile_grid(im.shape, patch, ov)                                  # core/test.py:87-90 (always the nuclei rule)
        assert (list(ds_), list(dh), list(dw)) == (list(g["d_s%d" % i]), list(g["d_h%d" % i]), list(g["d_w%d" % i]))
        if g["seed_im%d" % i].size:
            vol, _ = tiling.pad_slices(tiling.norm1(g["seed_im%d" % i], np.float64), patch[0])   # :180-195
            assert np.array_equal(vol, g["pim%d" % i])
    assert tiling.tile_grid((96, 256, 256), (64, 160, 160), 0, "soma") == ([0, 32, 64], [0, 96], [0, 96])


def test_detection_tile_grid_reads_crop_ovlp_from_the_config():
    """core/test.py:87-90 with TEST.CROP_OVLP: 100 under the nuclei YAML, 32 (core/config.py:250) under the soma YAML."""
    from m3d import tiling
    from m3d.config import Cfg

    def ref(dim, p, ov):
        return list(range(0, dim - p, p - ov)) + [dim - p]
    for cfg, shape in ((Cfg.soma(), (96, 512, 400)), (Cfg.nuclei(), (64, 700, 512))):
        p = cfg.in_size
        want = tuple(ref(d, q, cfg.crop_ovlp) for d, q in zip(shape, p))
        assert tuple(tiling.detect_grid(cfg, shape)) == want
    assert tiling.detect_grid(Cfg.soma(), (96, 512, 400))[1] == [0, 128, 256, 352]       # step 160 - 32, not 160 - 100


def test_norm1_and_quantize():
    from m3d import tiling
    import oracle as O
    rs = np.random.RandomState(0)
    im = rs.randint(0, 900, (5, 6, 7)).astype(np.uint16)
    assert np.array_equal(tiling.norm1(im), O.norm1(im))
    a = rs.rand(4, 5, 6).astype(np.float32)
    assert np.array_equal(tiling.quantize_u8(a), O.quantize_prm_u8(a))
    p, ps = tiling.pad_slices(np.arange(24, dtype=np.float32).reshape(2, 3, 4), 5)
    assert ps == 1 and p.shape == (5, 3, 4) and np.array_equal(p[0], p[1]) and np.array_equal(p[-1], p[-3])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "instanceseg-without-voxelwise-labeling_amd"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from m3d import shard
    n_items = 5
    mine = shard.partition(n_items, rank, world)
    local = [torch.full((i + 1, 7), float(i)) for i in mine]      # item i has i+1 detections, all valued i
    calls = []
    orig = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    out = shard.all_gather_detections(local, 4, n_items, dist)
    assert len(calls) == 1                                        # exactly ONE collective per exchange (counts ride along)
    q.put((rank, [(int(t.shape[0]), float(t[0, 0])) for t in out]))
    dist.destroy_process_group()


def test_shard_all_gather_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
    expect = [(min(i + 1, 4), float(i)) for i in range(5)]      # capped at 4, global item order restored
    assert res[0] == expect and res[1] == expect


def test_product_anchors_match_reference(golden):
    from m3d.config import Cfg, generate_anchors_3d
    g = golden("anchors")
    assert np.array_equal(Cfg.nuclei().anchors, g["nuclei"]) and Cfg.nuclei().anchors.dtype == np.float64
    assert np.array_equal(Cfg.soma().anchors, g["soma"])
    assert Cfg.nuclei().num_anchors == 35 and Cfg.soma().num_anchors == 14
    assert generate_anchors_3d().shape == (6, 6)


def test_bench_launcher_starts_n_ranks_itself_and_does_one_gather():
    """`python bench.py --gpus 2 --backend gloo --dry`: the SAME launcher and exchange code path as the GPU run (bench.py
    spawns the ranks itself when no external launcher set WORLD_SIZE), stub detect step, rank 0 prints the one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry"] is True and d["config"]["volumes_per_step"] == 8 and d["steps"] == 2   # 4 volumes per rank at every N
    # N = 1 path: no process group, same code
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_launcher_stops_the_other_ranks_when_one_dies():
    """A rank that exits (here: rank 1, before its first collective) used to leave rank 0 waiting in the all_gather and the
    launcher waiting for rank 0, for ever.  The launcher now ends the others and returns the failing rank's code."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["M3D_BENCH_TEST_KILL_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 7 and time.time() - t0 < 200
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no bench line from a broken run
    # more ranks than GPUs over RCCL is refused up front, with a message, not discovered by a hang
    env.pop("M3D_BENCH_TEST_KILL_RANK")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0 and "GPU(s)" in (r.stderr + r.stdout)


_NUCLEI_YAML = """
PP_METHOD: 'norm1'
PRM_ON: True
MODEL:
  TYPE: generalized_rcnn
  CONV_BODY: DSN.dsn_body
  MASK_ON: False
  NUM_CLASSES: 2
  BBOX_REG_WEIGHTS: (10., 10., 10., 5., 5., 5.)
FPN:
  FPN_ON: False
RPN:
  SIZES: (10, 27, 33, 38, 42, 46, 50)
  STRIDE: 8
  ASPECT_RATIOS: [[1.0, 0.5], [0.5, 0.5], [2., 0.5], [0.2, 0.5], [3., 2.]]
FAST_RCNN:
  ROI_BOX_HEAD: fast_rcnn_heads.roi_2mlp_head
  ROI_XFORM_METHOD: RoIAlign
  ROI_XFORM_RESOLUTION: 7
  ROI_XFORM_SAMPLING_RATIO: 2
  MLP_HEAD_DIM: 1024
MRCNN:
  RESOLUTION: 14
  DILATION: 1  # default 2
  CLS_SPECIFIC_MASK: False
  ROI_XFORM_SAMPLING_RATIO: 2
TRAIN:
  IN_SIZE: (64, 256, 256)
  SOME_KEY_THIS_PATH_DOES_NOT_READ: 3
TEST:
  NEED_CROP: True
  CROP_OVLP: 100
  NMS: 0.15
  RPN_NMS_THRESH: 0.15
  RPN_PRE_NMS_TOP_N: 1000  # Per FPN level
  RPN_POST_NMS_TOP_N: 1000
  DETECTIONS_PER_IM: 300
  IN_SIZE: (64, 200, 200)
"""


def test_cfg_from_yaml_reads_the_reference_keys():
    """Cfg.from_yaml == cfg_from_file for the keys of this path (lib/core/config.py:1063-1160): tuples as strings are decoded,
    unknown keys ignored, other model families refused; the two shipped YAMLs (read where the reference lies, when it is there)
    give exactly Cfg.nuclei() / Cfg.soma()."""
    from m3d.config import Cfg
    c = Cfg.from_yaml(_NUCLEI_YAML)
    ref = Cfg.nuclei()
    for k in ("stride", "sizes", "aspect_ratios", "pre_nms_topN", "post_nms_topN", "rpn_nms_thresh", "nms", "detections_per_im",
              "bbox_reg_weights", "num_classes", "roi_res", "sampling_ratio", "mlp_dim", "in_size", "crop_ovlp", "dataset", "score_thresh"):
        assert getattr(c, k) == getattr(ref, k), k
    assert np.array_equal(c.anchors, ref.anchors) and c.num_anchors == 35
    assert (c.mask_on, c.mask_dilation, c.mask_cls_specific, c.mask_sampling_ratio, c.mask_resolution) == (False, 1, False, 2, 14)
    soma_like = _NUCLEI_YAML.replace("STRIDE: 8", "STRIDE: 4").replace("  CROP_OVLP: 100\n", "").replace("IN_SIZE: (64, 200, 200)", "IN_SIZE: (64, 160, 160)")
    c4 = Cfg.from_yaml(soma_like)
    assert c4.dataset == "soma" and Cfg.from_yaml(soma_like, dataset="nuclei").dataset == "nuclei"
    assert c4.in_size == (64, 160, 160) and c4.crop_ovlp == 32          # a key the file leaves out: lib/core/config.py's default (:250)
    for bad in ("MODEL:\n  CONV_BODY: ResNet.ResNet50_conv4_body\n", "FPN:\n  FPN_ON: True\n", "FAST_RCNN:\n  ROI_XFORM_METHOD: RoIPoolF\n",
                "RPN:\n  STRIDE: 8\n"):                                   # the last: 2-D defaults left in place (aspect ratios, 4 box weights)
        with pytest.raises(NotImplementedError):
            Cfg.from_yaml(bad)
    base = "/root/reference/configs"
    if os.path.isdir(base):
        n = Cfg.from_yaml(os.path.join(base, "cell_tracking_baseline", "e2e_mask_rcnn_N3DH_SIM_dsn_body.yaml"))
        s_ = Cfg.from_yaml(os.path.join(base, "soma_starting", "e2e_mask_rcnn_soma_dsn_body.yaml"))
        for got, want in ((n, Cfg.nuclei()), (s_, Cfg.soma())):
            for k in ("stride", "sizes", "aspect_ratios", "pre_nms_topN", "post_nms_topN", "rpn_nms_thresh", "nms", "detections_per_im",
                      "bbox_reg_weights", "num_classes", "roi_res", "sampling_ratio", "mlp_dim", "in_size", "crop_ovlp", "dataset"):
                assert getattr(got, k) == getattr(want, k), (want.dataset, k, getattr(got, k), getattr(want, k))


class _StubDetector:
    """Stands in for DetectorM3D in the CPU rehearsal of the sharded driver: detections are a deterministic function of the cube, so
    every partition of the tiles over ranks must end in the same cross-tile result."""

    def __init__(self):
        from m3d.config import Cfg
        self.cfg = Cfg.nuclei(in_size=(8, 16, 16), crop_ovlp=4)

    def detect_batch(self, cubes):
        outs = []
        for c in cubes:
            m = float(c.mean())
            k = 1 + int(abs(m) * 1000) % 3
            rows = [[1.0 + j, 2.0 + j, 1.0, 8.0 + j, 9.0 + j, 5.0, 0.3 + 0.2 * j + (abs(m) % 0.05)] for j in range(k)]
            outs.append({"cls_boxes": [None, torch.tensor(rows, dtype=torch.float32)]})
        return outs


def _detect_all_worker(rank, world, port, q, shape):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (os.path.join(here, "instanceseg-without-voxelwise-labeling_amd"), os.path.join(here, "oracle")):
        sys.path.insert(0, p_)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from m3d import infer, shard
    calls = []
    orig = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    im = (np.random.RandomState(5).rand(*shape) * 500 + 20).astype(np.uint16)
    nms = lambda d, t: torch.from_numpy(O.nms_3d(d.numpy(), t))          # noqa: E731  (the checker's NMS: no GPU in this test)
    res = infer.im_detect_all(_StubDetector(), im, dist=dist, tile_batch=2, device="cpu", nms_fn=nms)
    mine = shard.partition(1, 0, 1)
    q.put((rank, len(calls), res[1].tolist(), len(mine)))
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,ntiles,world", [((8, 53, 24), 10, 4), ((8, 32, 16), 3, 4), ((8, 100, 100), 64, 8)])
def test_im_detect_all_sharded_over_four_ranks_uneven_and_empty_ranks(shape, ntiles, world):
    """im_detect_all(dist=...) end to end on 4 gloo ranks: 10 tiles (3 + 3 + 2 + 2) and 3 tiles (rank 3 holds NONE and must still enter
    the path's one collective) - every rank ends with the single-process result, after exactly one all_gather.  And BASELINE
    configs[4]'s partition: 64 work items over 8 ranks, 8 each (gloo stands in for RCCL, which has never run here: no 8-GPU node)."""
    import sys
    import torch.multiprocessing as mp
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "oracle"))
    import oracle as O
    from m3d import infer, tiling
    det = _StubDetector()
    assert len(tiling.enumerate_tiles(*tiling.detect_grid(det.cfg, shape))) == ntiles
    im = (np.random.RandomState(5).rand(*shape) * 500 + 20).astype(np.uint16)
    nms = lambda d, t: torch.from_numpy(O.nms_3d(d.numpy(), t))          # noqa: E731
    single = infer.im_detect_all(det, im, dist=None, tile_batch=2, device="cpu", nms_fn=nms)[1].tolist()
    assert len(single) >= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_detect_all_worker, args=(r, world, port, q, shape)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in ps:
        p.join(60)
    assert sorted(g[0] for g in got) == list(range(world))
    for rank, ncalls, res, _ in got:
        assert ncalls == 1, (rank, ncalls)                                # one collective per rank, also on the rank without a tile
        assert res == single, rank


def test_bench_dry_run_at_world_8_is_configs4_shape_and_a_compact_line():
    """`bench.py --gpus 8 --vols-per-rank 8 --backend gloo --dry`: BASELINE configs[4]'s partition (64 volumes, 8 per rank) through the
    launcher and the path's one all_gather on 8 gloo ranks; the line is as compact as the N = 1 line and names the exchange."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--vols-per-rank", "8", "--backend", "gloo", "--dry",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert len(last) < 8000
    d = json.loads(last)
    assert d["n_gpus"] == 8 and d["config"]["volumes_per_step"] == 64 and d["config"]["volumes_per_rank"] == 8 and d["scaling"] == "weak"
    assert d["exchange"]["ranks"] == 8 and d["exchange"]["backend"] == "gloo" and d["exchange"]["us"] > 0


def test_writer_pool_reports_a_failed_task_once_and_keeps_working():
    """m3d.infer._WriterPool.finish(): a failed tile write is raised by the finish() that waited for it - after every other task of
    that batch has been waited for - and NOT again by the next volume's finish() (the pool is process-global)."""
    from m3d.infer import _WriterPool
    pool = _WriterPool(workers=2)
    done = []

    def bad():
        raise IOError("disk full")

    def good(tag):
        done.append(tag)
    pool.submit(bad)
    pool.submit(good, "a")
    with pytest.raises(IOError, match="disk full"):
        pool.finish()
    assert done == ["a"] and pool.pending == []            # the good task of the same batch ran and was waited for
    pool.submit(good, "b")
    pool.finish()                                          # no stale exception
    assert done == ["a", "b"] and pool.pending == []
    pool.submit(bad)
    pool.submit(bad)
    with pytest.raises(IOError):
        pool.finish()
    pool.finish()
    pool.close()
