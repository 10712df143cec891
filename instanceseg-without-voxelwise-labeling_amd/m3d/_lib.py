"""ctypes loader for libm3d.so (include/m3d.h).  Fails loudly when the HIP library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M3D_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libm3d.so")   # env override: tuning builds only

M3D_OK = 0
_lib = None

SYMBOLS = [
    "m3d_version", "m3d_error_string", "m3d_last_hip_error", "m3d_set_option", "m3d_get_option", "m3d_tuning_build",
    "m3d_conv3d_stem5_prepare_dgrad_weights", "m3d_conv3d_stem5_dgrad", "m3d_norm1_workspace_bytes", "m3d_norm1", "m3d_norm1_batched",
    "m3d_linear_workspace_bytes", "m3d_linear_forward", "m3d_linear_bf16x3_packed_bytes", "m3d_linear_bf16x3_pack", "m3d_absmax", "m3d_linear_f16x2_packed_bytes", "m3d_linear_f16x2_pack", "m3d_linear_f16x2_workspace_bytes", "m3d_linear_f16x2_forward",
    "m3d_linear_bf16x3_workspace_bytes", "m3d_linear_bf16x3_forward", "m3d_mask_paste3d_workspace_bytes", "m3d_mask_paste3d", "m3d_linear_bf16x3_w32_workspace_bytes", "m3d_linear_bf16x3_w32_forward", "m3d_roi_align3d_tap_tables", "m3d_linear_bf16x3_roi_workspace_bytes", "m3d_linear_bf16x3_roi_forward",
    "m3d_fused_max_boxes", "m3d_compact_rows", "m3d_compact_rows2", "m3d_box_head_outputs", "m3d_conv3d_forward_split_sigmoid", "m3d_generate_proposals3d_batched_workspace_bytes", "m3d_generate_proposals3d_batched",
    "m3d_box_results3d_batched_workspace_bytes", "m3d_box_results3d_batched", "m3d_nms3d_batched_workspace_bytes", "m3d_nms3d_batched",
    "m3d_roi_align3d_forward", "m3d_roi_align3d_forward_exact", "m3d_roi_align3d_backward", "m3d_roi_align3d_workspace_bytes", "m3d_roi_align3d_forward_ws", "m3d_roi_align3d_forward_ws2",
    "m3d_nms3d_workspace_bytes", "m3d_nms3d", "m3d_bbox_overlaps3d", "m3d_bbox_transform3d",
    "m3d_generate_proposals3d_workspace_bytes", "m3d_generate_proposals3d",
    "m3d_conv3d_packed_weight_bytes", "m3d_conv3d_pack_weights", "m3d_conv3d_forward", "m3d_conv3d_forward_dilated", "m3d_conv3d_forward_windowed", "m3d_conv3d_forward_pool2",
    "m3d_prm_seed", "m3d_prm_seed_ex", "m3d_prm_strip_geometry", "m3d_prm_select_peaks", "m3d_prm_select_peaks_ex", "m3d_prm_prepare", "m3d_prm_stem_prepare_weights", "m3d_prm_stem_dgrad", "m3d_prm_scatter", "m3d_prm_den_pool", "m3d_prm_stem_mfma_prepare_weights", "m3d_prm_stem_dgrad_fused_supported", "m3d_prm_stem_dgrad_fused", "m3d_prm_stem_dgrad_fused_ex", "m3d_prm_prepare_ex", "m3d_prm_stem_dgrad_fused_ex2", "m3d_prm_prepare_ex2", "m3d_prm_prepare_ex3", "m3d_prm_strip_dgrad_prepare", "m3d_conv3d_x3_packed_bytes", "m3d_conv3d_x3f_packed_bytes", "m3d_conv3d_x3f_pack", "m3d_conv3d_x3f_forward_ws", "m3d_reduce_minmax_multi", "m3d_reduce_minmax_multi_workspace_bytes", "m3d_conv3d_x3_supported", "m3d_conv3d_x3_pack", "m3d_conv3d_x3_forward", "m3d_conv3d_x3_workspace_bytes", "m3d_conv3d_x3_forward_ws", "m3d_conv3d_x3_launch_units", "m3d_conv3d_zw_supported", "m3d_conv3d_zw_packed_bytes", "m3d_conv3d_zw_pack", "m3d_conv3d_zw_slots", "m3d_conv3d_zw_bound_of", "m3d_conv3d_zw_forward", "m3d_conv3d_zw_forward_strip", "m3d_prm_strip_dgrad_prepare_zw", "m3d_prm_strip_absmax", "m3d_prm_small_dgrad_packed_bytes", "m3d_prm_small_dgrad_pack", "m3d_prm_small_dgrad_f16_supported", "m3d_prm_small_dgrad_f16_packed_bytes", "m3d_prm_small_dgrad_f16_pack", "m3d_prm_small_dgrad_f16_workspace_bytes", "m3d_prm_small_dgrad_f16", "m3d_prm_small_dgrad",
    "m3d_maxpool3d_2x_forward", "m3d_maxpool3d_2x_backward",
    "m3d_reduce_min_workspace_bytes", "m3d_reduce_min", "m3d_reduce_min_multi_workspace_bytes", "m3d_reduce_min_multi",
    "m3d_otsu2d_workspace_bytes", "m3d_otsu2d_batch", "m3d_prm_quantize_u8", "m3d_roi_normalize", "m3d_prm_quantize_windows_u8", "m3d_prm_quantize_windows_compact_u8", "m3d_roi_normalize_ws", "m3d_roi_normalize_idx",
    "m3d_conv3d_wgrad_workspace_bytes", "m3d_conv3d_wgrad", "m3d_conv3d_bias_grad",
    "m3d_conv3d_wino_packed_weight_bytes", "m3d_conv3d_wino_pack_weights", "m3d_conv3d_wino_forward", "m3d_conv3d_wino_forward_pool2",
    "m3d_conv3d_wino2_packed_weight_bytes", "m3d_conv3d_wino2_pack_weights", "m3d_conv3d_wino2_forward", "m3d_conv3d_wino2_forward_pool2", "m3d_conv3d_wino2_forward_pool2_argmax", "m3d_conv3d_wino2_workspace_bytes", "m3d_conv3d_wino2_forward_ws", "m3d_conv3d_wino2_score", "m3d_conv3d_wino2_local_score", "m3d_conv3d_wino2_family", "m3d_conv3d_wino2_plan", "m3d_conv3d_wino2_local_workspace_bytes", "m3d_conv3d_wino2_local_forward_ws",
    "m3d_conv3d_stem_wino_packed_weight_bytes", "m3d_conv3d_stem_wino_pack_weights", "m3d_conv3d_stem_wino_forward", "m3d_conv3d_stem_wino_forward_bound",
    "m3d_gaussian_filter_u16", "m3d_median_filter3_u16",
    "m3d_cc_workspace_bytes", "m3d_cc_largest_batch", "m3d_binary_closing6_batch", "m3d_paint_instances", "m3d_paint_finish", "m3d_paint_begin",
]


class M3DError(RuntimeError):
    pass


_TUNE_ENV = (("M3D_XCD_MAP", "xcd_map"), ("M3D_TUNE_K3", "tune_k3"), ("M3D_TUNE_WINO", "tune_wino"),
             ("M3D_TUNE_WINO2", "tune_wino2"), ("M3D_TUNE_WINO2_XT", "tune_wino2_xt"), ("M3D_TUNE_FC_SLICES", "tune_fc_slices"),
             ("M3D_TUNE_FC_SLICES_TAIL", "tune_fc_slices_tail"), ("M3D_TUNE_FC_X3_ROWS", "tune_fc_x3_rows"), ("M3D_TUNE_STEM", "tune_stem"), ("M3D_TUNE_FC_X_ALIAS", "tune_fc_x_alias"), ("M3D_TUNE_ROI_XCD", "tune_roi_xcd"))
TUNE_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libm3d_tune.so")
_tune = None


def _load(path):
    if not os.path.exists(path):
        raise M3DError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C instanceseg-without-voxelwise-labeling_amd/csrc`. There is no CPU fallback." % path)
    L = C.CDLL(path)
    L.m3d_error_string.restype = C.c_char_p
    L.m3d_last_hip_error.restype = C.c_char_p
    L.m3d_conv3d_wino2_score.restype = C.c_double
    L.m3d_conv3d_wino2_local_score.restype = C.c_double
    L.m3d_prm_strip_geometry.restype = C.c_int64
    L.m3d_conv3d_x3_launch_units.restype = C.c_longlong
    for n in ("m3d_roi_align3d_workspace_bytes", "m3d_nms3d_workspace_bytes", "m3d_generate_proposals3d_workspace_bytes",
              "m3d_conv3d_packed_weight_bytes", "m3d_conv3d_x3_packed_bytes", "m3d_conv3d_x3f_packed_bytes", "m3d_conv3d_zw_packed_bytes", "m3d_reduce_minmax_multi_workspace_bytes", "m3d_conv3d_x3_workspace_bytes", "m3d_reduce_min_workspace_bytes", "m3d_reduce_min_multi_workspace_bytes", "m3d_norm1_workspace_bytes", "m3d_prm_small_dgrad_packed_bytes", "m3d_prm_small_dgrad_f16_packed_bytes", "m3d_prm_small_dgrad_f16_workspace_bytes", "m3d_linear_workspace_bytes", "m3d_linear_bf16x3_packed_bytes", "m3d_linear_bf16x3_workspace_bytes", "m3d_linear_f16x2_packed_bytes", "m3d_linear_f16x2_workspace_bytes", "m3d_linear_bf16x3_w32_workspace_bytes", "m3d_linear_bf16x3_roi_workspace_bytes", "m3d_mask_paste3d_workspace_bytes", "m3d_generate_proposals3d_batched_workspace_bytes",
              "m3d_box_results3d_batched_workspace_bytes", "m3d_nms3d_batched_workspace_bytes", "m3d_otsu2d_workspace_bytes",
              "m3d_cc_workspace_bytes", "m3d_conv3d_wgrad_workspace_bytes", "m3d_conv3d_wino_packed_weight_bytes", "m3d_conv3d_wino2_packed_weight_bytes", "m3d_conv3d_wino2_workspace_bytes", "m3d_conv3d_wino2_local_workspace_bytes", "m3d_conv3d_stem_wino_packed_weight_bytes"):
        getattr(L, n).restype = C.c_size_t
    return L


def lib():
    """The library every m3d op calls: libm3d.so (release: no mutable process-wide state), or - inside `with tuning():`, or for the
    whole process when an M3D_TUNE_* / M3D_XCD_MAP knob of the A/B tools is in the environment - the tuning build libm3d_tune.so.
    The library itself never reads the environment; this loader - the caller - forwards the knobs once."""
    global _lib
    if _lib is None:
        knobs = [(env, name) for env, name in _TUNE_ENV if env in os.environ]
        if knobs and "M3D_LIB_PATH" not in os.environ:
            _lib = _load(TUNE_LIB_PATH)
        else:
            _lib = _load(LIB_PATH)
        if knobs and _lib.m3d_tuning_build():
            for env, name in knobs:
                set_option(name, int(os.environ[env]))
    return _lib


class tuning:
    """`with tuning(): set_option(...)`: m3d ops go to libm3d_tune.so (the build whose option table is mutable) inside the block and
    back to the release library after it; options are restored to their defaults on exit.  For A/B tools and kernel-family tests."""

    def __enter__(self):
        global _lib, _tune
        lib()
        if _tune is None:
            _tune = _lib if _lib.m3d_tuning_build() else _load(TUNE_LIB_PATH)
        self.prev = _lib
        _lib = _tune
        return self

    def __exit__(self, *a):
        global _lib
        for _, name in _TUNE_ENV:
            _tune.m3d_set_option(name.encode(), 1 if name == "xcd_map" else -1)
        _lib = self.prev
        return False


def set_option(name, value):
    rc = lib().m3d_set_option(name.encode(), int(value))
    if rc == -4 and not lib().m3d_tuning_build():
        raise M3DError("set_option(%s): the release library has no mutable options; use `with m3d._lib.tuning():` (libm3d_tune.so)" % name)
    check(rc, "set_option(%s)" % name)


def get_option(name):
    v = C.c_int(0)
    check(lib().m3d_get_option(name.encode(), C.byref(v)), "get_option(%s)" % name)
    return v.value


def check(rc, what=""):
    if rc != M3D_OK:
        L = lib()
        msg = L.m3d_error_string(rc).decode()
        hip = L.m3d_last_hip_error().decode()
        raise M3DError("%s failed: %s (%d) %s" % (what or "m3d call", msg, rc, hip))
