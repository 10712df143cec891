#!/usr/bin/env python3
"""tests/golden/tiling.npz by EXECUTING the reference's own tiling code from where it lies: the statements of
tools/infer_simple.py:180-212 (PRM branch: norm1, slice padding, tile starts for 'nuclei' / 'soma', tile ids) and of
lib/core/test.py:76-90 (detection branch) are read at generation time, dedented and exec'd on synthetic volumes with a stand-in
`cfg` / `args` (the scripts themselves cannot run here: infer_simple.py exits without CUDA, :80-81, and hard-codes its paths,
:85-93).  Nothing of the reference is stored: the fixture holds shapes, index lists and checksums of the padded volume.

Run in the build container only:   python tests/golden/gen_tiling.py
"""
import os
import textwrap
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("M3D_REFERENCE", "/root/reference")
if not hasattr(np, "int"):
    np.int = int                                     # NumPy-1 alias the reference uses (np.int(...))


def ref_lines(rel, first, last):
    with open(os.path.join(REF, rel)) as f:
        lines = f.read().split("\n")
    return textwrap.dedent("\n".join(lines[first - 1:last]))


def ns_cfg(in_size, ovlp):
    return types.SimpleNamespace(PRM_ON=True, TEST=types.SimpleNamespace(IN_SIZE=in_size, CROP_OVLP=ovlp, NEED_CROP=True))


def run_infer_simple(im, in_size, ovlp, dataset):
    """tools/infer_simple.py:180-208 (pre-process ... len_w) + the triple loop header and `num =` of :209-212."""
    env = {"np": np, "im": im.copy(), "cfg": ns_cfg(in_size, ovlp), "args": types.SimpleNamespace(dataset=dataset), "nums": []}
    exec(ref_lines("tools/infer_simple.py", 180, 208), env)
    loop = ref_lines("tools/infer_simple.py", 209, 212)
    exec(loop + "\n" + " " * 12 + "nums.append((num, s, h, w))", env)      # the loop body's first statement, then collect
    return env


def run_core_test(vol, in_size, ovlp):
    """lib/core/test.py:76-90 (detection branch of im_detect_all): padding and tile starts."""
    inputs = {"data": vol[None, None].astype(np.float32), "im_info": np.array([list(vol.shape) + [1.0]])}
    env = {"np": np, "tile": np.tile, "inputs": inputs, "cfg": ns_cfg(in_size, ovlp)}
    exec(ref_lines("lib/core/test.py", 76, 90), env)
    return env


def main():
    rs = np.random.RandomState(5)
    out, i = {}, 0
    cases = [((59, 350, 350), (64, 200, 200), 100, "nuclei"), ((64, 200, 200), (64, 200, 200), 100, "nuclei"),
             ((100, 256, 256), (64, 200, 200), 100, "nuclei"), ((100, 256, 256), (64, 160, 160), 32, "nuclei"),
             ((96, 256, 256), (64, 160, 160), 32, "soma"), ((12, 40, 40), (16, 24, 24), 8, "nuclei")]
    for shape, patch, ov, ds in cases:
        im = (rs.rand(*shape) * 900 + 50).astype(np.uint16)
        im[rs.rand(*shape) < 0.01] = 0
        e = run_infer_simple(im, patch, ov, ds)
        out["shape%d" % i], out["patch%d" % i], out["ov%d" % i] = np.array(shape), np.array(patch), np.int64(ov)
        out["ds%d" % i] = np.array(ds)
        out["seed_im%d" % i] = im if im.size < 30000 else np.zeros((0,), np.uint16)     # small volumes are stored whole
        out["pad%d" % i] = np.int64(e["pad_s"])
        out["s%d" % i], out["h%d" % i], out["w%d" % i] = np.array(e["sidx"]), np.array(e["hidx"]), np.array(e["widx"])
        out["nums%d" % i] = np.array(e["nums"], np.int64)
        pim = e["im"]
        out["pshape%d" % i] = np.array(pim.shape)
        out["pstat%d" % i] = np.array([pim.mean(), pim.std(), pim[0].sum(), pim[-1].sum(), pim[pim.shape[0] // 2].sum()], np.float64)
        if im.size < 30000:
            out["pim%d" % i] = pim.astype(np.float64)
        # detection branch on the float32 norm1 volume (blob.py:179-184 runs before; same padding / tile expressions)
        d = run_core_test(im.astype(np.float32), patch, ov)
        out["d_pad%d" % i] = np.int64(d["pad_s"])
        out["d_s%d" % i], out["d_h%d" % i], out["d_w%d" % i] = np.array(d["sidx"]), np.array(d["hidx"]), np.array(d["widx"])
        out["d_pshape%d" % i] = np.array(d["orig_im"].shape)
        i += 1
    out["n"] = np.int64(i)
    p = os.path.join(HERE, "tiling.npz")
    np.savez_compressed(p, **out)
    print("wrote tiling.npz (%.1f KB), %d cases" % (os.path.getsize(p) / 1024, i))


if __name__ == "__main__":
    main()
