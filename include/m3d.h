/* m3d.h — C ABI of libm3d.so: the MI355X (gfx950) implementation of the 3D detection hot path of
 * MeowMeowLady/InstanceSeg-Without-Voxelwise-Labeling.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer named d_* is a DEVICE pointer (HBM), caller-allocated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is asynchronous on it
 *     and never synchronises the device, so calls may be captured into a hipGraph;
 *   - return value: M3D_OK (0) or a negative M3D_E* code; nothing ever calls exit()
 *     (the reference's launchers print and exit(-1): roi_align_kernel_3d.cu:165-169);
 *   - re-entrant; scratch memory comes from the caller (`d_ws`, sized by the matching *_workspace_bytes());
 *     the library never reads the environment and keeps no mutable process-wide state (the tuning knobs below are compiled out of
 *     libm3d.so; they live in the separate tuning build libm3d_tune.so).
 * Paths in comments are relative to the reference repository.
 */
#ifndef M3D_H_
#define M3D_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3D_OK 0
#define M3D_EINVAL (-1)   /* bad argument (shape, NULL pointer, unsupported size) */
#define M3D_ELAUNCH (-2)  /* hipLaunch / runtime error (hipGetLastError) */
#define M3D_EWORKSPACE (-3) /* workspace too small */
#define M3D_EUNSUPPORTED (-4)

int m3d_version(void);
const char* m3d_error_string(int code);
/* Last HIP runtime error text seen by this thread (for M3D_ELAUNCH). */
const char* m3d_last_hip_error(void);
/* Tuning options (benchmark / A-B tooling).  The release library libm3d.so keeps NO mutable process-wide state: there
 * m3d_set_option returns M3D_EUNSUPPORTED for every known name and m3d_get_option reports the built-in defaults.  The knobs are live only
 * in libm3d_tune.so (same objects, m3d_core built with -DM3D_TUNING; m3d_tuning_build() == 1), which the scripts under tools/ and the kernel-family tests load
 * beside the release library.  Names: "xcd_map" (1: XCD-aware workgroup->tile order, default; 0: plain order), "tune_k3", "tune_wino",
 * "tune_wino2", "tune_wino2_xt" (tile-variant overrides of the conv dispatchers, -1 = library chooses), "tune_fc_slices" /
 * "tune_fc_slices_tail" (split-K factors of m3d_linear_forward), "tune_fc_x3_rows" (128 / 256: tile height of m3d_linear_bf16x3_forward),
 * "tune_fc_x_alias" (> 0: m3d_linear_bf16x3_forward reads row m of x from row m % value - a cache-resident operand, WRONG results:
 * the GEMM's cost with a free operand, tools/f1_ab.py), "tune_stem" (1: the round-2 one-row stem kernel; 4 / 8: the rows kernel with that many planes per workgroup; -1: rows kernel, planes by
 * grid size), "tune_fc_x3_rows" = 512 (the 256 x 256 tiles of m3d_linear_f16x2_forward), "tune_roi_xcd" (1: RoIAlign3D's XCD-aware
 * channel split - every XCD one eighth of every RoI's channels; an A/B that removed the L2 misses and not the time, round 6).
 * Unknown name -> M3D_EINVAL. */
int m3d_set_option(const char* name, int value);
int m3d_get_option(const char* name, int* value);
int m3d_tuning_build(void);

/* ---------------------------------------------------------------------------------------------------------
 * RoIAlign 3D.  Replaces roi_align_forward_cuda_3d / roi_align_backward_cuda_3d
 * (lib/modeling/roi_xfrom/roi_align_3d/src/roi_align_cuda_3d.h:1-5, .c:7-83; kernels
 * src/roi_align_kernel_3d.cu:81-151, 238-338).
 * features [batch,channels,slices,height,width] fp32; rois [num_rois,roi_cols] fp32, roi_cols must be 7
 * (batch,x1,y1,z1,x2,y2,z2) else M3D_EINVAL (the reference returns 0, roi_align_cuda_3d.c:19-22);
 * output [num_rois,channels,AS,AH,AW] as allocated by functions/roi_align_3d.py:24, written in the
 * reference kernel's (n,c,ph,pw,ps) memory order (roi_align_kernel_3d.cu:87-91).  Caller allocates
 * (and, for backward, zero-fills) the outputs exactly like functions/roi_align_3d.py:24,41-42.
 * ------------------------------------------------------------------------------------------------------- */
int m3d_roi_align3d_forward(int aligned_slices, int aligned_height, int aligned_width, float spatial_scale,
                            int sampling_ratio, const float* d_features, int batch, int channels, int slices,
                            int height, int width, const float* d_rois, int num_rois, int roi_cols,
                            float* d_output, void* stream);
/* Same contract, but every output element is computed in the reference kernel's exact fp32 operation order
 * (8 samples x 8 corner products, roi_align_kernel_3d.cu:73-76,128-147): bit-identical to oracle/m3d_oracle.c.
 * m3d_roi_align3d_forward uses the algebraically equal separable form (three 1-D passes through LDS, ~4x fewer
 * operand reads); the two agree to ~1e-6 * max|feature|. */
int m3d_roi_align3d_forward_exact(int aligned_slices, int aligned_height, int aligned_width, float spatial_scale,
                                  int sampling_ratio, const float* d_features, int batch, int channels, int slices,
                                  int height, int width, const float* d_rois, int num_rois, int roi_cols,
                                  float* d_output, void* stream);
/* m3d_roi_align3d_forward with a caller workspace (m3d_roi_align3d_workspace_bytes(num_rois) bytes, contents irrelevant): the launch takes
 * the RoIs in descending order of their work instead of index order, so the few large RoIs do not form its tail, and the per-RoI set-up
 * (sample tables, folds) is computed once and kept in the workspace instead of once per workgroup.  Identical results. */
size_t m3d_roi_align3d_workspace_bytes(int num_rois);
int m3d_roi_align3d_forward_ws(int aligned_slices, int aligned_height, int aligned_width, float spatial_scale,
                               int sampling_ratio, const float* d_features, int batch, int channels, int slices,
                               int height, int width, const float* d_rois, int num_rois, int roi_cols,
                               float* d_output, void* d_workspace, size_t workspace_bytes, void* stream);
/* Round 6: the same with d_feat_absmax, a device pointer to ONE float >= max |d_features| (m3d_absmax), or NULL (then exactly _ws): RoIs
 * whose sub-volume has <= 128 voxels run as ONE GEMM per RoI on the f16 matrix cores - out[c][bin] = sum_k f[c][k] M[k][bin], M the
 * separable interpolation operator of the RoI, both operands scaled and cut into two fp16 numbers, three products - the others through
 * the separable kernels as before.  Within 1e-6 max |f| of m3d_roi_align3d_forward_ws (the fast mode's contract is 1e-5 max |f|;
 * m3d_roi_align3d_forward_exact stays the reference-order, bit-exact form). */
int m3d_roi_align3d_forward_ws2(int aligned_slices, int aligned_height, int aligned_width, float spatial_scale,
                                int sampling_ratio, const float* d_features, int batch, int channels, int slices,
                                int height, int width, const float* d_rois, int num_rois, int roi_cols,
                                float* d_output, void* d_workspace, size_t workspace_bytes, const float* d_feat_absmax, void* stream);
int m3d_roi_align3d_backward(int aligned_slices, int aligned_height, int aligned_width, float spatial_scale,
                             int sampling_ratio, const float* d_top_grad, const float* d_rois, int num_rois,
                             int roi_cols, float* d_bottom_grad, int batch, int channels, int slices, int height,
                             int width, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Greedy 3D NMS.  Replaces utils.cython_nms_3d.nms_3d / nms_3d_volume
 * (lib/utils/cython_nms_3d.pyx:39-96, 102-159; wrappers lib/utils/boxes_3d.py:364-374).
 * d_dets [n,7] fp32 rows (x1,y1,z1,x2,y2,z2,score).  Visiting order: descending score (by_volume=0) or
 * descending volume (by_volume=1), ties in descending input index.  d_keep receives the surviving INPUT
 * indices in ascending order (np.where(suppressed == 0)[0], pyx:96); *d_num_keep their count.
 * fp32 arithmetic, `ovr >= thresh`, bit-exact with the reference.
 * n <= 16384: rank sort, N x N/64 IoU bitmask, one workgroup resolves it from LDS.  16384 < n <= 2^20 (round 6; the cross-tile NMS of
 * a whole volume's detections - tools/binarization_nuclei.py:81, tools/binarization_soma.py:57, lib/core/test.py:159 - has no bound in
 * the reference): the blocked form - chunks of 1024 sorted rows, a chunk's own 1024 x 1024 bitmask resolved by one workgroup against
 * the global removed bitmap, its kept boxes then tested against every later live box by a wide launch; same result, O(kept x alive)
 * IoUs, workspace n x 176 bytes.  n > 2^20: M3D_EUNSUPPORTED (the O(n^2) rank sort alone would take seconds).
 * ------------------------------------------------------------------------------------------------------- */
size_t m3d_nms3d_workspace_bytes(int n);
int m3d_nms3d(const float* d_dets, int n, float thresh, int by_volume, int64_t* d_keep, int32_t* d_num_keep,
              void* d_ws, size_t ws_bytes, void* stream);

/* N x K IoU matrix.  Replaces utils.cython_bbox_3d.bbox_overlaps_3d (lib/utils/cython_bbox_3d.pyx:32-80).
 * d_boxes [n,6], d_query [k,6] fp32 -> d_out [n,k] fp32 (fp32 intersection, fp64 union and divide). */
int m3d_bbox_overlaps3d(const float* d_boxes, int n, const float* d_query, int k, float* d_out, void* stream);

/* Box decode + clip.  Replaces utils.boxes_3d.bbox_transform_3d / clip_tiled_boxes_3d
 * (lib/utils/boxes_3d.py:167-225, 144-163).  d_boxes [n,6] fp32, d_deltas [n,6*classes] fp32,
 * weights[6] host doubles, xform_clip = cfg.BBOX_XFORM_CLIP; d_out [n,6*classes].
 * If clip_slices > 0 the result is also clipped to [0,dim-1] (x:width, y:height, z:slices). */
int m3d_bbox_transform3d(const float* d_boxes, const float* d_deltas, int n, int classes, const double* weights,
                         double xform_clip, double clip_slices, double clip_height, double clip_width,
                         float* d_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * RPN proposal generation for one image, fully on device.  Replaces GenerateProposalsOp_3d.forward /
 * proposals_for_one_image / _filter_boxes_3d (lib/modeling/generate_proposals_3d.py:19-192), which in the
 * reference copies scores and deltas to the host (:58-63).
 * d_scores [A,S,H,W] fp32 (post-sigmoid), d_deltas [6A,S,H,W] fp32, anchors[A*6] host doubles
 * (generate_anchors_3d), im_info = (slices,height,width,scale) host doubles.
 * Outputs (capacity post_nms_topN, or pre_nms_topN when post <= 0): d_rois [R,7] fp32 (batch column =
 * batch_index), d_probs [R] fp32, d_keep_idx [R] int64 = flat index into (S,H,W,A) (:160,174-175),
 * *d_num = R.  Top-N tie rule: descending score, ties ascending flat index.
 * ------------------------------------------------------------------------------------------------------- */
size_t m3d_generate_proposals3d_workspace_bytes(int A, int S, int H, int W, int pre_nms_topN);
int m3d_generate_proposals3d(const float* d_scores, const float* d_deltas, int A, int S, int H, int W,
                             const double* anchors, double feat_stride, const double* im_info,
                             int pre_nms_topN, int post_nms_topN, float nms_thresh, double min_size,
                             double xform_clip, int batch_index, float* d_rois, float* d_probs,
                             int64_t* d_keep_idx, int32_t* d_num, void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * 3D convolution (cross-correlation), NCDHW fp32, stride 1, zero padding k/2, dilation 1, groups 1 — what
 * torch.nn.Conv3d computes for every conv of lib/modeling/DSN.py:19-36 and lib/modeling/rpn_heads.py:54-61.
 * fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM, LDS-staged halo tiles, no im2col buffer.
 *
 * Weights are packed once (m3d_conv3d_pack_weights) into the MFMA A-fragment order; `mode` selects
 *   M3D_W_PLAIN     W                       (forward conv)
 *   M3D_W_RELU      relu(W)                 (PRM norm conv, lib/prm/peak_backprop_3d.py:41-42)
 *   M3D_W_DGRAD     flip + transpose W      (backward-data of a stride-1 same conv)
 *   M3D_W_DGRAD_RELU flip + transpose relu(W) (PRM backward, peak_backprop_3d.py:16-18 via autograd)
 * Forward computes, per output element:  y = conv(x - in_offset, Wp) ; y = y*scale[c] + shift[c] (if given;
 * bias and eval-mode BatchNorm fold into scale/shift) ; y = max(y,0) (if relu) ; y *= mul[..] (if d_mul given:
 * same shape as the output; the PRM PreHook product).  Padding voxels stay 0 after the offset subtraction.
 * ------------------------------------------------------------------------------------------------------- */
enum { M3D_W_PLAIN = 0, M3D_W_RELU = 1, M3D_W_DGRAD = 2, M3D_W_DGRAD_RELU = 3 };
size_t m3d_conv3d_packed_weight_bytes(int cin, int cout, int k, int mode);
int m3d_conv3d_pack_weights(const float* d_weight /*[cout,cin,k,k,k]*/, int cin, int cout, int k, int mode,
                            float* d_packed, void* stream);
int m3d_conv3d_forward(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                       int depth, int height, int width, int k, const float* d_in_offset /*1 float or NULL*/,
                       const float* d_scale, const float* d_shift, int relu, const float* d_mul, void* stream);
/* The two 1x1x1 RPN heads as ONE convolution with both epilogues (lib/modeling/rpn_heads.py:96-98 and the sigmoid of :116): output
 * channels [0, split) -> 1 / (1 + exp(-v)) -> d_out_sigmoid [batch, split, D, H, W]; channels [split, cout) -> d_out_rest
 * [batch, cout - split, D, H, W].  d_packed = m3d_conv3d_pack_weights of the concatenated weights; d_shift = the concatenated biases. */
int m3d_conv3d_forward_split_sigmoid(const float* d_in, const float* d_packed, float* d_out_sigmoid, float* d_out_rest, int batch, int cin,
                                     int cout, int split, int depth, int height, int width, int k, const float* d_shift, void* stream);
/* 3x3x3 "same" convolution with dilation 2 / padding 2 (dilation 1 = m3d_conv3d_forward): the convolutions of the mask head
 * (lib/modeling/mask_rcnn_heads.py:148-151 with MRCNN.DILATION = 2, lib/core/config.py:767).  d_packed: m3d_conv3d_pack_weights. */
int m3d_conv3d_forward_dilated(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout, int depth,
                               int height, int width, int k, int dilation, const float* d_scale, const float* d_shift, int relu,
                               void* stream);

/* 3x3x3 forward convolution with the Winograd F(2,3) transform along x (csrc/conv3d_wino.hip): the same operation as
 * m3d_conv3d_forward for k = 3, plain weights, no input offset / PRM multiply, at 2/3 of the MFMA work.  Results
 * agree with the direct kernel to a few fp32 ulp (tolerance of this path: 1e-4 relative, north_star), not bit for
 * bit, so the PRM norm / backward convolutions never use it.  Own packed-weight layout (36 slots per cout x cin).
 * Returns M3D_EUNSUPPORTED for maps narrower than 24 voxels (callers then use m3d_conv3d_forward). */
size_t m3d_conv3d_wino_packed_weight_bytes(int cin, int cout);
int m3d_conv3d_wino_pack_weights(const float* d_weight /*[cout,cin,3,3,3]*/, int cin, int cout, float* d_packed,
                                 void* stream);
int m3d_conv3d_wino_forward(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                            int depth, int height, int width, const float* d_scale, const float* d_shift, int relu,
                            void* stream);
/* the same fused with MaxPool3d(2,2) (DSN.py:60-61): writes [batch,cout,D/2,H/2,W/2]; width >= 48 */
int m3d_conv3d_wino_forward_pool2(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                  int depth, int height, int width, const float* d_scale, const float* d_shift,
                                  int relu, void* stream);

/* The same with Winograd F(2x2,3x3) on the (y,x) plane (csrc/conv3d_wino2.hip): 4/9 of the MFMA work; own packed layout
 * (48 slots per cout x cin); maps >= 24 wide, fused pool >= 48 wide; same few-ulp agreement with the direct kernel. */
size_t m3d_conv3d_wino2_packed_weight_bytes(int cin, int cout);
int m3d_conv3d_wino2_pack_weights(const float* d_weight /*[cout,cin,3,3,3]*/, int cin, int cout, float* d_packed,
                                  void* stream);
int m3d_conv3d_wino2_forward(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                             int depth, int height, int width, const float* d_scale, const float* d_shift, int relu,
                             void* stream);
/* small maps (12..23 voxels wide, e.g. the 16^3 stage-4 layers): 16x16x2 output tiles with split-K over workgroups;
 * partial results go to the caller's workspace and are summed in a fixed order by a second kernel (deterministic).
 * m3d_conv3d_wino2_forward_ws dispatches to m3d_conv3d_wino2_forward for maps >= 24 wide (workspace unused). */
size_t m3d_conv3d_wino2_workspace_bytes(int batch, int cin, int cout, int depth, int height, int width);
/* useful-work x chip-fill score (0..1) of the tile the library would pick (64x2x4, 32x8x2 or 16x16x2 with split-K);
 * below ~0.5 the direct kernel is the better choice */
double m3d_conv3d_wino2_score(int batch, int cin, int cout, int depth, int height, int width);
/* ... and for the exactly-local F(2x2,3x3) family of m3d_conv3d_wino2_local_forward_ws (other tiles, hence its own score) */
double m3d_conv3d_wino2_local_score(int batch, int cin, int cout, int depth, int height, int width);
/* the 2-D Winograd kernel family in use: 1-3 = F(2x2,3x3) variants (4/9 of the direct convolution's multiplies), 4 = F(2x4,3x3)
 * (F(2,3) along y, F(4,3) along x: 1/3; the default), 5 = F(2x4,3x3) with 64 output channels per 4-wave workgroup (A/B only; layers
 * whose cout is not a multiple of 64 run family 4); option "tune_wino2" / 100 selects one for A/B runs */
int m3d_conv3d_wino2_family(void);
/* What m3d_conv3d_wino2_forward_ws (local = 0) / m3d_conv3d_wino2_local_forward_ws (local = 1) would launch for this shape: kernel
 * family, tile id (32 / 16 / 8; 0: none) and K split (1: one accumulation chain per output over the input channels; s > 1: s partial
 * sums added in a fixed order, i.e. another summation order).  A pure function of the shape - no launch, no device access; replaces
 * nothing in the reference (cuDNN's algorithm choice behind `F.conv3d`, lib/prm/peak_backprop_3d.py:40-42, is equally shape-dependent). */
int m3d_conv3d_wino2_plan(int local, int batch, int cin, int cout, int depth, int height, int width, int* family_out, int* tile_out,
                          int* ksplit_out);
int m3d_conv3d_wino2_forward_ws(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                int depth, int height, int width, const float* d_scale, const float* d_shift, int relu,
                                void* d_ws, size_t ws_bytes, void* stream);
/* The same forward through the F(2x2,3x3) family only, whose outputs depend on nothing outside their own 3 x 3 (y, x) support, not
 * even by rounding (the default family's F(4,3) along x cancels the rest of its 6-wide footprint only to ~1e-7 of those inputs).
 * Used where unrelated data sits right next to a window (the PRM strip layout, m3d_prm_prepare_ex). */
size_t m3d_conv3d_wino2_local_workspace_bytes(int batch, int cin, int cout, int depth, int height, int width);
int m3d_conv3d_wino2_local_forward_ws(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                      int depth, int height, int width, const float* d_scale, const float* d_shift, int relu,
                                      void* d_ws, size_t ws_bytes, void* stream);
int m3d_conv3d_wino2_forward_pool2(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                   int depth, int height, int width, const float* d_scale, const float* d_shift,
                                   int relu, void* stream);
/* the same with the pool's argmax (uint8 [batch, cout, D/2, H/2, W/2], dz*4 + dy*2 + dx of the first maximum: the rule of
 * m3d_maxpool3d_2x_forward) - the response conv of the pooled layers in PRM mode */
int m3d_conv3d_wino2_forward_pool2_argmax(const float* d_in, const float* d_packed, float* d_out, unsigned char* d_argmax, int batch,
                                          int cin, int cout, int depth, int height, int width, const float* d_scale,
                                          const float* d_shift, int relu, void* stream);

/* conv1a (5x5x5, one input channel; DSN.py:19,58) with Winograd F(2,5) along x (csrc/conv3d_stem_wino.hip): 0.62x the MFMA
 * work of the direct stem kernel; own packed layout; pool != 0 fuses MaxPool3d(2,2) and writes [batch,cout,D/2,H/2,W/2].
 * Maps >= 32 voxels wide; agreement with the direct kernel ~1e-6 relative to the tensor maximum. */
size_t m3d_conv3d_stem_wino_packed_weight_bytes(int cout);
int m3d_conv3d_stem_wino_pack_weights(const float* d_weight /*[cout,1,5,5,5]*/, int cout, float* d_packed, void* stream);
int m3d_conv3d_stem_wino_forward(const float* d_in, const float* d_packed, float* d_out, int batch, int cout, int depth,
                                 int height, int width, const float* d_scale, const float* d_shift, int relu, int pool,
                                 void* stream);
/* Round 6: the same launch also leaves the largest |output| in d_out_max (32 floats, zeroed by the caller, atomic maxima; NULL: exactly
 * the call above) - the operand bound m3d_conv3d_zw_forward takes as d_in_max, so that the layer behind the stem needs no sweep. */
int m3d_conv3d_stem_wino_forward_bound(const float* d_in, const float* d_packed, float* d_out, int batch, int cout, int depth,
                                       int height, int width, const float* d_scale, const float* d_shift, int relu, int pool,
                                       float* d_out_max, void* stream);

/* Backward-weights and bias gradient of the same stride-1 "same" convolution (what autograd computes for the
 * F.conv3d calls of lib/prm/peak_backprop_3d.py:40-42 and every nn.Conv3d of lib/modeling/DSN.py:19-36 in training):
 *   dW[co,ci,dz,dy,dx] = sum_{b,z,y,x} gy[b,co,z,y,x] * x[b,ci,z+dz-k/2,...]     db[co] = sum gy[b,co,...]
 * fp32 MFMA implicit GEMM with the voxel index as the reduction dimension; split-K partials go to the workspace and
 * are summed in a fixed order (deterministic).  k = 1, 3 any cin; k = 5 with cin = 1.  d_grad_weight is [cout,cin,k,k,k]. */
size_t m3d_conv3d_wgrad_workspace_bytes(int batch, int cin, int cout, int depth, int height, int width, int k);
int m3d_conv3d_wgrad(const float* d_in, const float* d_grad_out, float* d_grad_weight, int batch, int cin, int cout,
                     int depth, int height, int width, int k, void* d_ws, size_t ws_bytes, void* stream);
int m3d_conv3d_bias_grad(const float* d_grad_out, float* d_grad_bias, int batch, int cout, int depth, int height,
                         int width, void* stream);

/* Conv + scale/shift + ReLU + MaxPool3d(2,2) fused (DSN.py:58,60-61: pool1/pool2 directly follow a conv):
 * writes only the pooled tensor [batch,cout,D/2,H/2,W/2] (8x fewer output bytes, no separate pool pass) and,
 * if d_argmax != NULL, the window index (z*4+y*2+x, first maximum) for the PRM un-pooling.  Supported for the
 * 5^3 stem (cout <= 32) and k = 3 with width >= 24; otherwise M3D_EUNSUPPORTED (use conv + m3d_maxpool3d_2x). */
int m3d_conv3d_forward_pool2(const float* d_in, const float* d_packed, float* d_out_pooled, uint8_t* d_argmax,
                             int batch, int cin, int cout, int depth, int height, int width, int k,
                             const float* d_in_offset, const float* d_scale, const float* d_shift, int relu,
                             void* stream);

/* Same conv on a batch of cropped windows, with the PRM PreHook multiply fused: out[b,co,z,y,x] *=
 * full[co, origin_b + (z,y,x)] - *d_full_offset, and 0 where that position lies outside the full tensor
 * (lib/prm/peak_backprop_3d.py:16-18 restricted to each peak's receptive-field cone).
 * d_full [cout, full_depth, full_height, full_width]; d_origins int32 [batch,3] = (z,y,x) of each window. */
int m3d_conv3d_forward_windowed(const float* d_in, const float* d_packed, float* d_out, int batch, int cin, int cout,
                                int depth, int height, int width, int k, const float* d_full,
                                const float* d_full_offset, const int32_t* d_origins, int full_depth,
                                int full_height, int full_width, void* stream);

/* MaxPool3d(kernel 2, stride 2, floor) forward with argmax (lib/modeling/DSN.py:21,26,32) and its
 * backward (gradient routed to the argmax voxel; first maximum in z,y,x scan order wins, as PyTorch). */
int m3d_maxpool3d_2x_forward(const float* d_in, float* d_out, uint8_t* d_argmax /*may be NULL*/, int batch_channels,
                             int depth, int height, int width, void* stream);
int m3d_maxpool3d_2x_backward(const float* d_grad_out, const uint8_t* d_argmax, float* d_grad_in,
                              int batch_channels, int depth, int height, int width, void* stream);

/* Global minimum of a device fp32 array into d_out[0] (the `input.min()` offset of
 * lib/prm/peak_backprop_3d.py:38); stays on device so the PRM convs never synchronise with the host. */
size_t m3d_reduce_min_workspace_bytes(void);
int m3d_reduce_min(const float* d_in, int64_t n, float* d_out, void* d_ws, size_t ws_bytes, void* stream);
/* The minima of up to 12 arrays in two launches (the PRM forward: one `input.min()` per patched conv): d_ins / counts are HOST arrays of
 * `count` device pointers / element counts, d_out [count] on the device. */
size_t m3d_reduce_min_multi_workspace_bytes(void);
int m3d_reduce_min_multi(const float* const* d_ins, const int64_t* counts, int count, float* d_out, void* d_ws, size_t ws_bytes,
                         void* stream);
/* Round 6: minima AND maxima of the same arrays in the same two launches (d_mins / d_maxs [count]); workspace
 * m3d_reduce_minmax_multi_workspace_bytes().  The f16x2 norm convolutions (m3d_conv3d_x3f_forward_ws) scale X - min X by max X - min X. */
size_t m3d_reduce_minmax_multi_workspace_bytes(void);
int m3d_reduce_minmax_multi(const float* const* d_ins, const int64_t* counts, int count, float* d_mins, float* d_maxs, void* d_ws,
                            size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Batched, fused box post-processing: ONE launch per stage for a whole batch of tiles, one workgroup per tile, no host
 * round trips (csrc/box_fused.hip).  Same results as the per-tile entry points above, item by item.  Every item may
 * hold at most m3d_fused_max_boxes() (2048) boxes, else M3D_EUNSUPPORTED (use the per-tile entry points).
 *   m3d_generate_proposals3d_batched  GenerateProposalsOp_3d.forward (lib/modeling/generate_proposals_3d.py:19-192) for
 *       d_scores [batch,A,S,H,W] / d_deltas [batch,6A,S,H,W]: outputs d_rois [batch,out_rows,7] (column 0 =
 *       first_batch_index + item), d_probs [batch,out_rows], d_keep_idx [batch,out_rows], d_num [batch].
 *   m3d_box_results3d_batched         box_results_with_nms_and_limit (lib/core/test.py:806-883) per item: rows
 *       [d_offsets[b], d_offsets[b+1]) of d_scores [R,nc] / d_boxes [R,6nc] / d_keep_idx [R] (may be NULL) ->
 *       d_cls_boxes [batch,nc,max_rows,7], d_cls_keep [batch,nc,max_rows] (may be NULL), d_counts [batch,nc].
 *       Contract: d_offsets is non-decreasing and no item holds more than max_rows_per_item rows (the outputs and the
 *       scratch are sized for that many); rows of an item beyond max_rows_per_item are ignored.
 *   m3d_nms3d_batched                 nms_3d / nms_3d_volume (lib/utils/cython_nms_3d.pyx:39-159) per item of d_dets
 *       (item b at d_dets + b*item_stride_floats, count d_counts[b*count_stride] or max_boxes if d_counts is NULL):
 *       d_keep [batch,max_boxes] kept indices (ascending), d_num_keep [batch], and/or d_packed [batch,out_cap+1,7] = the
 *       kept rows, zero padded, with the count in element [out_cap][0] - the block m3d.shard all-gathers
 *       (lib/core/test.py:150-160). */
int m3d_fused_max_boxes(void);
/* Rows [0, min(d_counts[b], max_rows)) of every item b of d_src (item b at d_src + b*item_stride_bytes, rows of row_bytes, a multiple
 * of 4) packed back to back in item order into d_dst; d_offsets (may be NULL) receives the batch+1 row offsets.  The batched form of
 * `rois[:num]` per tile (lib/core/test.py:106-114 hands each tile's valid RoIs to the box head): no host round trip. */
int m3d_compact_rows(const void* d_src, size_t item_stride_bytes, size_t row_bytes, const int32_t* d_counts, int batch, int max_rows,
                     void* d_dst, int32_t* d_offsets, void* stream);
/* m3d_compact_rows for TWO row sets that share the counts (the RoIs and their score indices), one launch; h_counts (optional): a
 * device-accessible HOST mirror (pinned memory) of the clamped counts, written by the kernel - the caller waits for one stream event
 * instead of issuing a device-to-host copy. */
int m3d_compact_rows2(const void* d_src_a, size_t item_stride_bytes_a, size_t row_bytes_a, const void* d_src_b, size_t item_stride_bytes_b,
                      size_t row_bytes_b, const int32_t* d_counts, int batch, int max_rows, void* d_dst_a, void* d_dst_b,
                      int32_t* d_offsets, int32_t* h_counts, void* stream);
/* The box head's outputs in one launch (lib/modeling/fast_rcnn_heads.py:42-45, lib/core/test.py:225,250-251): d_outs [num_rois,
 * nc + 6 nc] = the cls_score and bbox_pred linears computed as ONE GEMM; d_rois [num_rois, 7] (batch, x1, y1, z1, x2, y2, z2) ->
 * d_cls [num_rois, nc] = softmax of the scores, d_bbox [num_rois, 6 nc] = the raw deltas, d_pred [num_rois, 6 nc] =
 * bbox_transform_3d(rois[:, 1:7], deltas, weights) clipped to (clip_slices, clip_height, clip_width) when clip_slices > 0. */
int m3d_box_head_outputs(const float* d_outs, const float* d_rois, int num_rois, int num_classes, const double* weights, double xform_clip,
                         double clip_slices, double clip_height, double clip_width, float* d_cls, float* d_bbox, float* d_pred, void* stream);
size_t m3d_generate_proposals3d_batched_workspace_bytes(int batch, int A, int S, int H, int W, int pre_nms_topN);
/* m3d_generate_proposals3d_batched: out_rows = rows of the per-item output blocks, >= min(K, post_nms_topN) with NMS; with nms_thresh <= 0
 * every valid box of the K pre-NMS candidates is a proposal (generate_proposals_3d.py:167-171), so out_rows must be >= K - M3D_EINVAL
 * otherwise, never a silently truncated list. */
int m3d_generate_proposals3d_batched(const float* d_scores, const float* d_deltas, int batch, int A, int S, int H, int W,
                                     const double* anchors, double feat_stride, const double* im_info, int pre_nms_topN,
                                     int post_nms_topN, float nms_thresh, double min_size, double xform_clip,
                                     int first_batch_index, int out_rows, float* d_rois, float* d_probs,
                                     int64_t* d_keep_idx, int32_t* d_num, void* d_ws, size_t ws_bytes, void* stream);
size_t m3d_box_results3d_batched_workspace_bytes(int batch);
int m3d_box_results3d_batched(const float* d_scores, const float* d_boxes, const int64_t* d_keep_idx,
                              const int32_t* d_offsets, int batch, int num_classes, float score_thresh, float nms_thresh,
                              int detections_per_im, int max_rows_per_item, float* d_cls_boxes, int64_t* d_cls_keep,
                              int32_t* d_counts, void* d_ws, size_t ws_bytes, void* stream);
size_t m3d_nms3d_batched_workspace_bytes(int batch);
int m3d_nms3d_batched(const float* d_dets, size_t item_stride_floats, const int32_t* d_counts, int count_stride, int batch,
                      int max_boxes, float thresh, int by_volume, int out_cap, float* d_packed, int64_t* d_keep,
                      int32_t* d_num_keep, void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fully-connected layer: d_out [M,N] = act(d_x [M,K] . d_weight [N,K]^T + d_bias [N]).  Replaces the cuBLAS SGEMMs of
 * nn.Linear at lib/modeling/fast_rcnn_heads.py:84-85,114-115 (fc1: K = C*7^3 = 87 808, N = MLP_HEAD_DIM; fc2) and
 * :15-19,42-45 (cls_score, bbox_pred).  Operands keep PyTorch's layouts (both K-contiguous).  fp32 MFMA split-K GEMM;
 * partial sums live in d_ws (m3d_linear_workspace_bytes) and are reduced in a fixed order (deterministic).
 * K must be a multiple of 4 and both operands 16-byte aligned (else M3D_EUNSUPPORTED); d_bias may be NULL; relu != 0
 * fuses max(.,0). */
size_t m3d_linear_workspace_bytes(int M, int N, int K);
int m3d_linear_forward(const float* d_x, const float* d_weight, const float* d_bias, float* d_out, int M, int N, int K,
                       int relu, void* d_ws, size_t ws_bytes, void* stream);

/* The same layer on the bf16 matrix cores at fp32 accuracy ("bf16x3 split"): every fp32 operand is cut EXACTLY into three bf16
 * numbers (x = xh + xm + xl, 8 significand bits each) and six bf16 MFMAs (the products down to 2^-23 of the full product)
 * accumulate in fp32 what one fp32 MFMA step does - 16 / 6 of the fp32 matrix rate, error vs fp64 as the fp32 kernel's.
 * Finite operands only: an inf or NaN operand makes its output row / column NaN (fp32 would keep inf * finite = inf).
 * The weight is cut once: m3d_linear_bf16x3_pack writes three bf16 planes in tile order (m3d_linear_bf16x3_packed_bytes =
 * 6 bytes per weight, N rounded up to 128); x stays fp32 and is cut inside the kernel.  K must be a multiple of 32
 * (else M3D_EUNSUPPORTED / packed_bytes 0: use m3d_linear_forward).  Same split-K / fixed-order reduction contract. */
size_t m3d_linear_bf16x3_packed_bytes(int N, int K);
int m3d_linear_bf16x3_pack(const float* d_weight, int N, int K, void* d_packed, void* stream);
size_t m3d_linear_bf16x3_workspace_bytes(int M, int N, int K);
int m3d_linear_bf16x3_forward(const float* d_x, const void* d_packed, const float* d_bias, float* d_out, int M, int N, int K,
                              int relu, void* d_ws, size_t ws_bytes, void* stream);
/* The variant for many rows: 256 x 256 tiles, BOTH operands fp32 in HBM (d_weight = the nn.Linear weight itself, [N, K]) and cut on
 * the way to LDS - 16 KB of loads per 128 x 128 x 32 of work instead of 28-40, which is what bounds the bf16x3 GEMM.  Same accuracy
 * and contract as above; its own workspace size. */
size_t m3d_linear_bf16x3_w32_workspace_bytes(int M, int N, int K);
int m3d_linear_bf16x3_w32_forward(const float* d_x, const float* d_weight, const float* d_bias, float* d_out, int M, int N, int K,
                                  int relu, void* d_ws, size_t ws_bytes, void* stream);

/* Round 6 - the same layer with THREE products per fp32 product ("f16x2 split"; replaces the same cuBLAS SGEMMs,
 * lib/modeling/fast_rcnn_heads.py:84-85,114-115): each operand is scaled by a power of two (its largest magnitude just under 2^15) and
 * cut into two fp16 numbers, x s = xh + xl with |x s - xh - xl| <= 2^-22 |x s| (11 + 11 significand bits, round to nearest); three
 * v_mfma_f32_32x32x16_f16 products (hh, hl, lh) accumulate in fp32, the scales are undone exactly in the epilogue.  Dropped:
 * xl.wl and the cut residuals, each <= 2^-22 of the product and of random sign - below what the fp32 accumulation of K products
 * rounds in this kernel and in an SGEMM alike; tests compare both with fp64 on the shipped shape.  Half the matrix-core time of
 * m3d_linear_bf16x3_forward, 4 bytes per packed weight (packed_bytes also holds the weight's largest magnitude).
 * d_x_bound: device pointer to ONE float >= max|x| (m3d_absmax of x or of whatever bounds x: a RoIAlign output is a convex
 * combination of its feature map's values, so the map's largest magnitude does), or NULL: the call sweeps x itself first.
 * A bound smaller than max|x| overflows fp16 (inf / NaN outputs); a bound up to 2^8 too large costs no accuracy.  Finite operands only.
 * Workspace m3d_linear_f16x2_workspace_bytes (always >= 256, 16-byte aligned).  K % 32 == 0 else M3D_EUNSUPPORTED. */
int m3d_absmax(const float* d_x, long long n, float* d_out, void* stream);
size_t m3d_linear_f16x2_packed_bytes(int N, int K);
int m3d_linear_f16x2_pack(const float* d_weight, int N, int K, void* d_packed, void* stream);
size_t m3d_linear_f16x2_workspace_bytes(int M, int N, int K);
int m3d_linear_f16x2_forward(const float* d_x, const void* d_packed, const float* d_bias, float* d_out, int M, int N, int K,
                             int relu, const float* d_x_bound, void* d_ws, size_t ws_bytes, void* stream);

/* f-1 A/B: the fc1 GEMM with the RoIAlign gather in its A-operand loader - the [M, C * 343] RoIAlign output is never
 * written.  m3d_roi_align3d_tap_tables: the RoIs' per-axis sample tables (7^3 bins, sampling grid 2 - the shipped geometry) -> d_tab
 * (16 bytes x 3 axes x 14 samples per RoI) and d_roi_batch [num_rois]; m3d_linear_bf16x3_roi_forward: out[M, N] = act(RoIAlign3D(features,
 * rois) W^T + b) with d_packed = m3d_linear_bf16x3_pack(W) and K = channels * 343.  Same samples and weights as m3d_roi_align3d_forward
 * (another summation order).  Measured against the two-launch path (RoIAlign, then m3d_linear_bf16x3_forward) in profiles/r04_f1_ab.json:
 * the two-launch path is the product path. */
int m3d_roi_align3d_tap_tables(const float* d_rois, int num_rois, float spatial_scale, int batch, int slices, int height, int width,
                               void* d_tab, int32_t* d_batch, void* stream);
size_t m3d_linear_bf16x3_roi_workspace_bytes(int M, int N, int K);
int m3d_linear_bf16x3_roi_forward(const float* d_features, int batch, int channels, int slices, int height, int width, const void* d_tab,
                                  const int32_t* d_roi_batch, const void* d_packed, const float* d_bias, float* d_out, int M, int N, int relu,
                                  void* d_ws, size_t ws_bytes, void* stream);

/* Paste of the mask branch's soft masks into full-volume uint8 masks: segm_results, lib/core/test.py:886-945.
 * d_masks [num_dets, channels, M, M, M] (M = resolution, MRCNN.RESOLUTION); d_channel [num_dets]: the channel of each detection
 * (its class when MRCNN.CLS_SPECIFIC_MASK, else 0); d_boxes [num_dets, 6] = expand_boxes(ref_boxes, (M+2)/M).astype(int32) as
 * x0, y0, z0, x1, y1, z1 (inclusive; may leave the volume).  The (M+2)^3 zero-padded block of a detection is resized to its box's
 * (s, h, w) as skimage.transform.resize(order=1, mode='reflect', anti_aliasing=True) does (Gaussian pre-filter + order-1
 * map_coordinates, 'mirror' boundary; restated on scipy.ndimage's arithmetic, see csrc/mask_paste.hip), thresholded (> thresh) and
 * written into d_out [num_dets, depth, height, width] (uint8, zeroed here).  The Gaussian taps are the caller's: d_radius
 * [num_dets, 3] (z, y, x; 0 = axis not filtered) and d_weights [num_dets, 3, weight_stride] doubles, [d] = weight at distance d. */
size_t m3d_mask_paste3d_workspace_bytes(int num_dets, int resolution);
int m3d_mask_paste3d(const float* d_masks, int num_dets, int channels, int resolution, const int32_t* d_channel,
                     const int32_t* d_boxes, const int32_t* d_radius, const double* d_weights, int weight_stride, float thresh,
                     int depth, int height, int width, unsigned char* d_out, void* d_ws, size_t ws_bytes, void* stream);

/* norm1 pre-processing of a raw volume on the device: mask = im > 0; out = (im - mean(im[mask])) / std(im[mask])
 * (np.std: population).  Replaces the host NumPy code of lib/utils/blob.py:179-184 (float32; f32_arith = 1) and
 * tools/infer_simple.py:180-183 (float64, crops cast to float32 at :217; f32_arith = 0), so the raw uint16 volume is what
 * crosses PCIe.  in_dtype: 0 = uint16, 1 = float32.  Statistics are two-pass fp64 sums in a fixed order (deterministic);
 * d_stats (optional, 3 doubles) receives mean, std, count. */
size_t m3d_norm1_workspace_bytes(void);
int m3d_norm1(const void* d_in, int in_dtype, int64_t n, int f32_arith, float* d_out, double* d_stats, void* d_ws,
              size_t ws_bytes, void* stream);
/* batch volumes of n voxels each, contiguous; each with its own statistics (one launch per pass for the whole batch).
 * d_stats [batch,3] or null; d_ws: batch * m3d_norm1_workspace_bytes() */
int m3d_norm1_batched(const void* d_in, int in_dtype, int batch, int64_t n, int f32_arith, float* d_out, double* d_stats, void* d_ws,
                      size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Peak-response back-propagation on cropped windows (all kept peaks of a tile as one batch).  Replaces the
 * per-detection `class_response_maps.backward(...)` loop of lib/prm/peak_response_mapping_3d.py:157-172 and
 * the hooks of lib/prm/peak_backprop_3d.py:8-34.  Windows are cubes in virtual coordinates (positions
 * outside the tile are zero), d_origin* are int32 [num_peaks,3] = (z,y,x).
 *   m3d_prm_seed      sigmoid' + the 1x1x1 RPN_cls_score conv for one-hot seeds: d_peaks int32 [P,4] =
 *                     (anchor,s,h,w); d_prob/d_norm_cls [A,S,H,W]; d_w_cls [A,C]; d_h [C,S,H,W] (the ReLU'd
 *                     RPN_conv output) and its min; writes d_out [P,C] (a 1^3 window per peak at (s,h,w)).
 *   m3d_prm_prepare   upper window d_gup [P,C,U,U,U] (gradient w.r.t. this layer's post-activation output, at
 *                     d_origin_up) -> this layer's conv-input window d_out [P,C,Wn,Wn,Wn], Wn = (pool?2:1)*U +
 *                     2*border: max-unpool routing by d_argmax (if pool), ReLU mask (d_xnext > 0; pooled values
 *                     if pool), * d_scale[c] (BatchNorm, may be NULL), / (|norm|+1e-10) with norm < 1e-10 -> 0.
 *   m3d_prm_stem_prepare_weights  d_weight [C,1,5,5,5] -> d_wf [C,125] = flipped relu(W) (once per model).
 *   m3d_prm_stem_dgrad backward-data of conv1a (5^3, one output channel) with d_wf, times (data - offset),
 *                     clamp(min=0); d_out [P,Wn^3]; d_sums [P] = per-peak sum (for prm / prm.sum()).
 *   m3d_prm_scatter   dense [P,D,H,W] = window / sum at each peak's origin, 0 / sum elsewhere (every voxel is written: a peak whose
 *                     map sums to 0 - saturated sigmoid - gives NaN everywhere, as the reference's prm / prm.sum() does).
 * ------------------------------------------------------------------------------------------------------- */
int m3d_prm_seed(const int32_t* d_peaks, int num_peaks, const float* d_prob, const float* d_norm_cls,
                 const float* d_w_cls, const float* d_h, const float* d_h_offset, int A, int C, int S, int H, int W,
                 float* d_out, void* stream);
/* the same + d_origin_out int32 [P,3] (or null): every peak's (s, h, w), the origin of its 1^3 window */
int m3d_prm_seed_ex(const int32_t* d_peaks, int num_peaks, const float* d_prob, const float* d_norm_cls,
                    const float* d_w_cls, const float* d_h, const float* d_h_offset, int A, int C, int S, int H, int W,
                    float* d_out, int32_t* d_origin_out, void* stream);
/* Peak selection on the device (lib/prm/peak_response_mapping_3d.py:124-139,161-163): class 1's kept detections d_dets [rows,7]
 * (x1,y1,z1,x2,y2,z2,score; the first *d_count rows are valid) with their flat score indices d_keep_idx into (S,H,W,A)
 * (generate_proposals_3d.py:160) -> the detections whose score > peak_threshold, in order: *d_num (<= cap), d_peaks int32 [cap,4] =
 * (anchor,s,h,w) = unravel_index(idx,(S,H,W,A)) reordered, d_out_dets [cap,7].  h_num / h_peaks / h_out_dets (optional, may be NULL):
 * device-accessible HOST mirrors (pinned memory) the kernel writes as well, so that the caller learns count, peaks and detections
 * from one stream event instead of three read-backs.  One workgroup; no host synchronisation. */
int m3d_prm_select_peaks(const float* d_dets, const int64_t* d_keep_idx, const int32_t* d_count, int rows, float peak_threshold,
                         int A, int S, int H, int W, int cap, int32_t* d_num, int32_t* d_peaks, float* d_out_dets, int32_t* h_num,
                         int32_t* h_peaks, float* h_out_dets, void* stream);
/* the same + d_prob [A,S,H,W] (the class response map, may be null) -> d_dead / h_dead int32 [cap] (each may be null): 1 for a peak whose
 * sigmoid derivative (1 - y) y is exactly 0 - its back-propagated map is all zero (`prm / prm.sum()` = 0 / 0) and need not be computed */
int m3d_prm_select_peaks_ex(const float* d_dets, const int64_t* d_keep_idx, const int32_t* d_count, int rows, float peak_threshold,
                            int A, int S, int H, int W, int cap, int32_t* d_num, int32_t* d_peaks, float* d_out_dets, int32_t* h_num,
                            int32_t* h_peaks, float* h_out_dets, const float* d_prob, int32_t* d_dead, int32_t* h_dead, void* stream);
/* Geometry of the strip layouts [C, n, n, L] of a window batch (all peaks side by side along x; see m3d_prm_prepare_ex): mode 1 =
 * pitch n + 1 for the exactly-local F(2x2,3x3) kernel, mode 2 = quad-aligned windows for the F(2x4,3x3) kernel (no output quad of one
 * window reads another window's columns).  Returns L; window p occupies columns lead + p * pitch .. + n - 1, all others are zero. */
int64_t m3d_prm_strip_geometry(int window, int mode, int num_peaks, int32_t* pitch, int32_t* lead);
int m3d_prm_prepare(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size, int pool,
                    int border, const uint8_t* d_argmax, const float* d_xnext, int up_depth, int up_height,
                    int up_width, const float* d_scale, const float* d_norm, int depth, int height, int width,
                    float* d_out, int32_t* d_origin_out, void* stream);
int m3d_prm_stem_prepare_weights(const float* d_weight, int channels, float* d_wf, void* stream);
int m3d_prm_stem_dgrad(const float* d_gn, const float* d_wf, const float* d_data, const float* d_data_offset,
                       const int32_t* d_origins, int num_peaks, int channels, int win, int depth, int height,
                       int width, float* d_out, float* d_sums, void* stream);
int m3d_prm_scatter(const float* d_windows, const float* d_sums, const int32_t* d_origins, int num_peaks, int win,
                    int depth, int height, int width, float* d_dense, void* stream);

/* Round 2: the stem step of the peak back-propagation on the matrix cores, with the prepare step fused (replaces
 * m3d_prm_prepare(pool = 1, border = 2) + m3d_prm_stem_dgrad for 32 stem channels and even window sizes; same results up to
 * fp32 summation order).
 *   m3d_prm_den_pool   peak-independent denominator map of a conv + MaxPool3d(2,2) layer, once per tile: d_den [C,UD,UH,UW] =
 *                      |N| + 1e-10 at the argmax child where the pooled activation is > 0 and N >= 1e-10, else 0
 *                      (lib/prm/peak_backprop_3d.py:30-33 with the ReLU mask and the max-unpool routing folded in).
 *   m3d_prm_stem_mfma_prepare_weights  conv1a weight [32,1,5,5,5] -> d_wa [80,64]: tap-flipped relu(W) in MFMA A-operand order.
 *   m3d_prm_stem_dgrad_fused_supported 1 when the fused kernel has a configuration for (channels, up_size).
 *   m3d_prm_stem_dgrad_fused  d_gup [P,32,U,U,U] (gradient w.r.t. the pooled stem output), d_origin_up int32 [P,3] (pooled
 *                      coordinates) -> d_out [P,Wn,Wn,Wn] (Wn = 2U + 4) = clamp((data - offset) * dgrad, min 0), d_sums [P],
 *                      d_origins_out int32 [P,3] = 2 * origin_up - 2   (peak_response_mapping_3d.py:169-171). */
int m3d_prm_den_pool(const uint8_t* d_argmax, const float* d_xnext, const float* d_norm, int channels, int up_depth, int up_height,
                     int up_width, int depth, int height, int width, float* d_den, void* stream);
int m3d_prm_stem_mfma_prepare_weights(const float* d_weight, int channels, float* d_wa, void* stream);
int m3d_prm_stem_dgrad_fused_supported(int channels, int up_size);
int m3d_prm_stem_dgrad_fused(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size,
                             const float* d_den, const uint8_t* d_argmax, const float* d_scale, int up_depth, int up_height,
                             int up_width, const float* d_wa, const float* d_data, const float* d_data_offset, int depth, int height,
                             int width, float* d_out, float* d_sums, int32_t* d_origins_out, void* stream);

/* Strip layout for window batches: the P windows of n^3 voxels side by side along x, one separator column after each -
 * [C, n, n, P*(n+1)] instead of [P, C, n, n, n] - so that m3d_conv3d_wino2_forward convolves the whole batch as ONE wide volume
 * (its 64-wide tiles fit 38 / 40-voxel windows badly, a 2 600-voxel strip well).  That kernel has no PreHook epilogue, so the
 * consumer of its output applies the multiply by (X - offset) of peak_backprop_3d.py:16-18:
 *   m3d_prm_prepare_ex           m3d_prm_prepare with in_strip / out_strip (0 batch-major, 1 strip) and d_up_offset (non-null: d_gup
 *                                is a bare backward-data result, multiplied here by (d_xnext - *d_up_offset)); writes the zero separators.
 *   m3d_prm_stem_dgrad_fused_ex  m3d_prm_stem_dgrad_fused with gup_strip and (d_xnext [32,UD,UH,UW], d_up_offset) for the same purpose. */
int m3d_prm_prepare_ex(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size, int pool, int border,
                       const uint8_t* d_argmax, const float* d_xnext, int up_depth, int up_height, int up_width, const float* d_scale,
                       const float* d_norm, int depth, int height, int width, int in_strip, int out_strip, const float* d_up_offset,
                       float* d_out, int32_t* d_origin_out, void* stream);
int m3d_prm_stem_dgrad_fused_ex(const float* d_gup, int gup_strip, const float* d_xnext, const float* d_up_offset,
                                const int32_t* d_origin_up, int num_peaks, int channels, int up_size, const float* d_den,
                                const uint8_t* d_argmax, const float* d_scale, int up_depth, int up_height, int up_width,
                                const float* d_wa, const float* d_data, const float* d_data_offset, int depth, int height, int width,
                                float* d_out, float* d_sums, int32_t* d_origins_out, void* stream);
/* Depth-clipped ("slab") strips, round 4: [C, zn, n, L] with zn = the depth of the LAYER's map - plane k of every window is plane k
 * of the map - instead of the n planes of each window.  Where a tile is thinner than the receptive-field cone (the nuclei tile's
 * stride-2 maps: 32 planes against windows of 38 / 40) the planes of a window outside the volume carry zero gradient in and are never
 * read out (peak_backprop_3d.py:30-33 only ever divides inside the map); a slab strip does not store, convolve or stream them.  y / x
 * geometry, separators and origins are those of the window strips (origins stay the cone's, z possibly negative).
 *   in_slab / out_slab / gup_slab != 0: that strip is a slab strip (d_gup: up_depth planes; d_out: depth planes); strip layouts only. */
int m3d_prm_prepare_ex2(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size, int pool, int border,
                        const uint8_t* d_argmax, const float* d_xnext, int up_depth, int up_height, int up_width, const float* d_scale,
                        const float* d_norm, int depth, int height, int width, int in_strip, int in_slab, int out_strip, int out_slab,
                        const float* d_up_offset, float* d_out, int32_t* d_origin_out, void* stream);
/* ... and d_peak_max [num_peaks x 32] (or NULL: exactly the call above): element 32 p = the largest |value| written for peak p (zeroed and
 * filled by this call; one cache line per peak) - the per-window operand bound m3d_conv3d_zw_forward_strip takes, so that the strip is never swept for it. */
int m3d_prm_prepare_ex3(const float* d_gup, const int32_t* d_origin_up, int num_peaks, int channels, int up_size, int pool,
                        int border, const uint8_t* d_argmax, const float* d_xnext, int up_depth, int up_height, int up_width,
                        const float* d_scale, const float* d_norm, int depth, int height, int width, int in_strip, int in_slab,
                        int out_strip, int out_slab, const float* d_up_offset, float* d_out, int32_t* d_origin_out,
                        float* d_peak_max, void* stream);
int m3d_prm_stem_dgrad_fused_ex2(const float* d_gup, int gup_strip, int gup_slab, const float* d_xnext, const float* d_up_offset,
                                 const int32_t* d_origin_up, int num_peaks, int channels, int up_size, const float* d_den,
                                 const uint8_t* d_argmax, const float* d_scale, int up_depth, int up_height, int up_width,
                                 const float* d_wa, const float* d_data, const float* d_data_offset, int depth, int height, int width,
                                 float* d_out, float* d_sums, int32_t* d_origins_out, void* stream);

/* 3x3x3 "same" convolution at fp32 accuracy on the bf16 matrix cores (csrc/conv3d_x3.hip; round 4): out = conv3d(x - *d_in_offset, W',
 * padding = 1) with W' = W or relu(W) chosen at pack time - the norm convolution of lib/prm/peak_backprop_3d.py:37-44 (pr_conv3d's second
 * F.conv3d: N = conv(X - min X, relu(W))), which has to stay an exact sum of products.  Both fp32 operands are cut into three bf16 pieces
 * (exact), six bf16 MFMA products per fp32 product: the error is the fp32 kernel's, non-negative operands give exact zeros exactly where
 * the fp32 sum has them.  cin must be a multiple of 16 (m3d_conv3d_x3_supported); zero padding applies to the SHIFTED input.
 *   m3d_conv3d_x3_packed_bytes / _pack   weight [cout, cin, 3, 3, 3] fp32 -> packed pieces (once per model; relu_weights: relu(W))
 *   m3d_conv3d_x3_forward                 x [B, cin, D, H, W] -> out [B, cout, D, H, W]; d_in_offset: device scalar or null */
size_t m3d_conv3d_x3_packed_bytes(int cin, int cout);
int m3d_conv3d_x3_supported(int cin, int cout);
int m3d_conv3d_x3_pack(const float* d_weight, int cin, int cout, int relu_weights, void* d_packed, void* stream);
int m3d_conv3d_x3_forward(const float* d_x, const void* d_packed, float* d_out, int batch, int cin, int cout, int depth, int height,
                          int width, const float* d_in_offset, void* stream);
/* The same with a caller workspace (m3d_conv3d_x3_workspace_bytes; 0 = none needed): maps too small to fill the chip are cut along K into
 * up to four ranges of input channels, one workgroup each, whose partial sums a second kernel adds in range order (deterministic). */
size_t m3d_conv3d_x3_workspace_bytes(int batch, int cin, int cout, int depth, int height, int width);
int m3d_conv3d_x3_forward_ws(const float* d_x, const void* d_packed, float* d_out, int batch, int cin, int cout, int depth, int height,
                             int width, const float* d_in_offset, void* d_workspace, size_t workspace_bytes, void* stream);
/* Round 6 - the same convolution with the "f16x2 split" of m3d_linear_f16x2_forward: both operands scaled by a power of two and cut into
 * two fp16 numbers (22 bits), three v_mfma_f32_32x32x16_f16 products per fp32 product (replaces the same F.conv3d call,
 * lib/prm/peak_backprop_3d.py:37-44).  d_in_max: device pointer to max(x) when d_in_offset is given (the operand x - offset is bounded by
 * max - offset), to max |x| when it is not (m3d_reduce_minmax_multi / m3d_absmax).  For the norm convolutions every operand is >= 0 and
 * every PRODUCT TRIPLE hh + hl + lh is >= 0: a result is zero exactly when every x w is, as with the fp32 and bf16x3 kernels.  Pack with
 * m3d_conv3d_x3f_pack (4 bytes per weight + the weight's largest magnitude); workspace / launch units as m3d_conv3d_x3_*. */
size_t m3d_conv3d_x3f_packed_bytes(int cin, int cout);
int m3d_conv3d_x3f_pack(const float* d_weight, int cin, int cout, int relu_weights, void* d_packed, void* stream);
int m3d_conv3d_x3f_forward_ws(const float* d_x, const void* d_packed, float* d_out, int batch, int cin, int cout, int depth, int height,
                              int width, const float* d_in_offset, const float* d_in_max, void* d_workspace, size_t workspace_bytes,
                              void* stream);
/* Workgroups m3d_conv3d_x3_forward_ws launches for this shape when given its workspace (spatial x cout tiles, times the K ranges): the
 * library's own decision, for callers that choose between this kernel and m3d_conv3d_forward by how well a launch fills the chip. */
long long m3d_conv3d_x3_launch_units(int batch, int cin, int cout, int depth, int height, int width);

/* Round 6 - the FORWARD 3x3x3 convolution + eval-BN + ReLU [+ MaxPool3d(2,2)] of lib/modeling/DSN.py:57-68 (and rpn_heads.py:94-96) on the
 * f16 matrix cores at fp32 accuracy: the f16x2 split (three fp16 products per fp32 product) with Winograd F(2,3) along z (36 instead of 54
 * products per output pair): csrc/conv3d_zw.hip.  cin must be a multiple of 16; depth >= 2, height >= 4, width >= 12 (pool: even extents,
 * width >= 24) - m3d_conv3d_zw_supported.  d_in_max: device array of m3d_conv3d_zw_slots() (= 32) non-negative floats whose LARGEST is
 * >= max |d_in| (the operand scale; an input beyond the bound overflows fp16); d_out_max (or NULL): the same kind of array for d_out,
 * updated with atomic maxima - the caller zeroes it before the launch and hands it to the next layer as its d_in_max, so that a chain of
 * layers needs one sweep (m3d_conv3d_zw_bound_of) for its first input only.  relu / d_scale / d_shift as m3d_conv3d_forward; pool = 1:
 * d_out is [batch, cout, depth/2, height/2, width/2].  Differences to the fp32 kernels: ~1e-6 of the largest output. */
int m3d_conv3d_zw_supported(int cin, int cout, int depth, int height, int width, int pool);
size_t m3d_conv3d_zw_packed_bytes(int cin, int cout);
int m3d_conv3d_zw_pack(const float* d_weight, int cin, int cout, void* d_packed, void* stream);
int m3d_conv3d_zw_slots(void);
int m3d_conv3d_zw_bound_of(const float* d_x, long long n, float* d_slots, void* stream);
int m3d_conv3d_zw_forward(const float* d_in, const void* d_packed, float* d_out, int batch, int cin, int cout, int depth, int height,
                          int width, const float* d_scale, const float* d_shift, int relu, int pool, const float* d_in_max,
                          float* d_out_max, void* stream);
/* The same convolution on a PRM window strip [cin, depth, height, width] (peak_backprop_3d.py:8-34 batched over peaks: the windows of all
 * peaks side by side along x, cell p = columns [pitch p, pitch (p + 1)), pitch a multiple of 4 - m3d_prm_strip_geometry mode 2) with ONE
 * OPERAND SCALE PER WINDOW: d_col_bound [num_peaks x 32], element 32 p = the largest |input| of window p (one cache line per window;
 * m3d_prm_strip_absmax: one sweep, rows = cin x depth x height; or the d_peak_max of m3d_prm_prepare_ex3).  The gradients of different peaks differ by orders of magnitude: each keeps its own 22 bits, and a window's
 * outputs do not depend on which other windows share the strip.  F(2,3) along z and the direct (y, x) taps are exactly local: a window's
 * outputs read nothing beyond its own 3^3 supports.  No scale / shift / ReLU / pool; width >= 24. */
int m3d_prm_strip_absmax(const float* d_strip, long long rows, int L, int pitch, int num_peaks, float* d_out, void* stream);
int m3d_conv3d_zw_forward_strip(const float* d_in, const void* d_packed, float* d_out, int cin, int cout, int depth, int height,
                                int width, const float* d_col_bound, int num_peaks, int pitch, void* stream);
/* m3d_conv3d_zw_forward_strip FUSED with the prepare step of the layer below: the contract of m3d_prm_strip_dgrad_prepare (the fp32
 * F(2x4,3x3) kernel's fused form), the same arguments plus d_col_bound (and d_packed from m3d_conv3d_zw_pack).  The strip it writes is the
 * two launches' (m3d_conv3d_zw_forward_strip, then m3d_prm_prepare_ex2 with pool = 0, border = 1) bit for bit; the bare gradient strip is
 * never stored.  M3D_EUNSUPPORTED where the kernel has no configuration (the caller takes the two launches). */
int m3d_prm_strip_dgrad_prepare_zw(const float* d_gn, const void* d_packed, int cin, int cout, int num_peaks, int window, int in_slab,
                                   const int32_t* d_origin, const float* d_xnext, const float* d_norm, const float* d_scale,
                                   const float* d_up_offset, int depth, int height, int width, int out_slab,
                                   const float* d_col_bound, float* d_out, int32_t* d_origin_out, void* stream);

/* Round 5: backward-data of a 3^3 conv on the quad-aligned strip FUSED with the prepare step of the layer below (no pooling between them):
 * d_gn [cin, in_planes, window, L(window)] (m3d_prm_prepare_ex2's out_strip = 2 layout; in_slab: the map's planes), d_packed =
 * m3d_conv3d_wino2_pack_weights of the backward-data weights -> d_out [cout, out_planes, window + 2, L(window + 2)], the strip
 * m3d_prm_prepare_ex2(in_strip = 2, out_strip = 2, pool = 0, border = 1, d_up_offset) would have made of the conv's result, bit for bit;
 * d_origin_out = d_origin - 1.  d_out is zero-filled by the call.  M3D_EUNSUPPORTED: take the two-launch path. */
int m3d_prm_strip_dgrad_prepare(const float* d_gn, const float* d_packed, int cin, int cout, int num_peaks, int window, int in_slab,
                                const int32_t* d_origin, const float* d_xnext, const float* d_norm, const float* d_scale,
                                const float* d_up_offset, int depth, int height, int width, int out_slab, float* d_out,
                                int32_t* d_origin_out, void* stream);

/* Backward-data of a 3x3x3 conv with relu(W) on batches of SMALL windows (win in {3, 5, 7}: the stride-8 / 4 stages of the peak
 * back-propagation), peaks batched densely into the GEMM N dimension; same operation as m3d_conv3d_forward_windowed on
 * dgrad-packed weights: d_gn [P, cout_fwd, win^3] -> d_out [P, cin_fwd, win^3] = (d_full[co][origin + v] - *d_full_offset) *
 * sum relu(W)[ci][co][26 - t] * gn[p, ci][v + t - 1]  (lib/prm/peak_backprop_3d.py:16-18,41-42).  d_weight: the forward conv's
 * [cout_fwd, cin_fwd, 3, 3, 3]; pack once per model. */
size_t m3d_prm_small_dgrad_packed_bytes(int cout_fwd, int cin_fwd);
int m3d_prm_small_dgrad_pack(const float* d_weight, int cout_fwd, int cin_fwd, float* d_packed, void* stream);
int m3d_prm_small_dgrad(const float* d_gn, const float* d_packed, int num_peaks, int cout_fwd, int cin_fwd, int win,
                        const float* d_full, const float* d_full_offset, const int32_t* d_origins, int depth, int height, int width,
                        float* d_out, void* stream);
/* Round 6 - the same operation on the f16 matrix cores ("f16x2 split", see m3d_linear_f16x2_forward): both operands scaled by a power of
 * two and cut into two fp16 numbers (22 bits), three v_mfma_f32_32x32x16_f16 products per fp32 product.  The gradient windows are
 * scaled PER PEAK (every peak's [cout_fwd, win^3] block is swept once for its largest magnitude into the workspace; peaks differ by
 * orders of magnitude), relu(W) per layer at pack time.  A column's result depends on its own peak only: a sub-batch gives the batch's
 * rows bit for bit.  cout_fwd must be a multiple of 16 (m3d_prm_small_dgrad_f16_supported), win in {3, 5, 7}; workspace
 * m3d_prm_small_dgrad_f16_workspace_bytes(num_peaks).  Agreement with m3d_prm_small_dgrad: ~1e-6 of the layer's largest value. */
int m3d_prm_small_dgrad_f16_supported(int cout_fwd, int cin_fwd);
size_t m3d_prm_small_dgrad_f16_packed_bytes(int cout_fwd, int cin_fwd);
int m3d_prm_small_dgrad_f16_pack(const float* d_weight, int cout_fwd, int cin_fwd, void* d_packed, void* stream);
size_t m3d_prm_small_dgrad_f16_workspace_bytes(int num_peaks);
int m3d_prm_small_dgrad_f16(const float* d_gn, const void* d_packed, int num_peaks, int cout_fwd, int cin_fwd, int win,
                            const float* d_full, const float* d_full_offset, const int32_t* d_origins, int depth, int height, int width,
                            float* d_out, void* d_ws, size_t ws_bytes, void* stream);
/* Backward-data of the 5^3 / Cin = 1 stem conv for autograd (what cuDNN dgrad computes for conv1a when the input requires
 * grad: the reference's PRM mode, lib/prm/peak_response_mapping_3d.py:88 + lib/prm/peak_backprop_3d.py:37-44, lib/modeling/DSN.py:19).
 *   m3d_conv3d_stem5_prepare_dgrad_weights  d_weight [C,1,5,5,5] -> d_wf [C,125], taps flipped (no ReLU: the caller passes
 *                                           whatever weight the conv ran with)
 *   m3d_conv3d_stem5_dgrad                  d_grad_out [batch,C,D,H,W] -> d_grad_in [batch,1,D,H,W] */
int m3d_conv3d_stem5_prepare_dgrad_weights(const float* d_weight, int channels, float* d_wf, void* stream);
int m3d_conv3d_stem5_dgrad(const float* d_grad_out, const float* d_wf, float* d_grad_in, int batch, int channels, int depth,
                           int height, int width, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * PRM post-processing feeding the Otsu step.
 *   m3d_prm_quantize_u8  per map: fm -= min; fm /= max; fm *= 255 -> uint8, float32 arithmetic
 *                        (tools/infer_simple.py:233-238; what the reference stores in its PRM TIFFs).
 *   m3d_roi_normalize    per detection: crop the inclusive box (x1,y1,z1,x2,y2,z2: int32 [R,6], tile coords) from
 *                        the uint16 tile and from the RoI's own uint8 PRM map, normalise both to similar ranges and
 *                        write them back to back at d_offsets[r] (int64 [R+1], offsets[r+1]-offsets[r] = crop voxels):
 *                        mode 0 = soma  (tools/binarization_soma.py:81-91), mode 1 = nuclei
 *                        (tools/binarization_nuclei.py:107-121).  float64 arithmetic, np.round / astype semantics.
 *                        The outputs are exactly the (image, prm) inputs of m3d_otsu2d_batch.
 * ------------------------------------------------------------------------------------------------------- */
int m3d_prm_quantize_u8(const float* d_prm, int num_maps, int64_t voxels_per_map, uint8_t* d_out, void* stream);
int m3d_roi_normalize(const uint16_t* d_image, const uint8_t* d_prm_u8, const int32_t* d_boxes,
                      const int64_t* d_offsets, int num_rois, int depth, int height, int width, int mode,
                      uint16_t* d_out_image, uint16_t* d_out_prm, void* stream);

/* Round 2: element-parallel forms of the two functions above, results identical.
 *   m3d_prm_quantize_windows_u8  the uint8 maps straight from the cone-cropped windows the back-propagation returns (d_windows
 *       [P, win^3] un-normalised, d_sums [P], d_origins int32 [P,3]): equals m3d_prm_quantize_u8 of the maps m3d_prm_scatter would
 *       build, without ever materialising them (tools/infer_simple.py:233-238).  d_ws: 16 * num_peaks bytes; on return
 *       int32 word 3 of each 16-byte record is 1 when the peak's uint8 map has a non-zero voxel.
 *   m3d_roi_normalize_ws         m3d_roi_normalize over a (chunks, RoI) grid; total_voxels = d_offsets[num_rois];
 *       d_ws: 24 * num_rois bytes. */
int m3d_prm_quantize_windows_u8(const float* d_windows, const float* d_sums, const int32_t* d_origins, int num_peaks, int win,
                                int depth, int height, int width, uint8_t* d_out, void* d_ws, size_t ws_bytes, void* stream);
/* Round 4: the same uint8 values as WINDOWS [P, win^3] (0 where a window voxel lies outside the tile) - what the whole-volume driver
 * sends to the host writer (m3d_io.h: m3d_tiff_encode_window_stack_u8) instead of the dense maps.  Same d_ws contract. */
int m3d_prm_quantize_windows_compact_u8(const float* d_windows, const float* d_sums, const int32_t* d_origins, int num_peaks, int win,
                                        int depth, int height, int width, uint8_t* d_out_windows, void* d_ws, size_t ws_bytes, void* stream);
int m3d_roi_normalize_ws(const uint16_t* d_image, const uint8_t* d_prm_u8, const int32_t* d_boxes, const int64_t* d_offsets,
                         int num_rois, int64_t total_voxels, int depth, int height, int width, int mode, uint16_t* d_out_image,
                         uint16_t* d_out_prm, void* d_ws, size_t ws_bytes, void* stream);
/* ... with an indirection and an optional compact source: RoI r reads map d_map_index[r] (int32 [num_rois]; NULL = map r) of d_prm_u8 =
 * the dense stack (win = 0) or the uint8 windows [*, win^3] of m3d_prm_quantize_windows_compact_u8 with d_win_origins int32 [*, 3] (a
 * map is zero outside its window): neither a gathered copy of the valid detections' maps nor the dense maps are needed. */
int m3d_roi_normalize_idx(const uint16_t* d_image, const uint8_t* d_prm_u8, const int32_t* d_map_index, int win,
                          const int32_t* d_win_origins, const int32_t* d_boxes, const int64_t* d_offsets, int num_rois,
                          int64_t total_voxels, int depth, int height, int width, int mode, uint16_t* d_out_image, uint16_t* d_out_prm,
                          void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Per-RoI 2D-Otsu binarisation.  Replaces otsu.otsu_py_2d_fast (tools/otsu.py:199-284) for uint16 inputs
 * (the callers normalise to uint16: tools/binarization_soma.py:85-91, binarization_nuclei.py:110-121).
 * A batch of `num_rois` independent crops: crop r occupies voxels [offsets[r], offsets[r+1]) of d_image /
 * d_prm / d_mask.  d_kb receives (k, b_max) per RoI; status[r] != 0 marks "no separating line".
 * ------------------------------------------------------------------------------------------------------- */
size_t m3d_otsu2d_workspace_bytes(int num_rois, int max_gray_range);
int m3d_otsu2d_batch(const uint16_t* d_image, const uint16_t* d_prm, const int64_t* d_offsets, int num_rois,
                     int max_gray_range, uint8_t* d_mask, int32_t* d_kb, int32_t* d_status, void* d_ws,
                     size_t ws_bytes, void* stream);

/* Volume pre-filters of tools/binarization_nuclei.py:44-45, bit-exact with SciPy 1.15 (uint16 volumes):
 *   m3d_gaussian_filter_u16  ndimage.gaussian_filter(img, sigma): separable correlate1d along z, y, x with 'reflect'
 *                            borders, fp64 accumulation in SciPy's order, truncation to uint16 after every pass.
 *                            d_weights: radius+1 doubles (centre first) computed by the host exactly as SciPy does;
 *                            d_tmp: scratch volume of the same size.
 *   m3d_median_filter3_u16   ndimage.median_filter(img, size=3): rank 13 of the 27 reflect-padded neighbours. */
int m3d_gaussian_filter_u16(const uint16_t* d_in, uint16_t* d_out, uint16_t* d_tmp, int depth, int height, int width,
                            const double* d_weights, int radius, void* stream);
int m3d_median_filter3_u16(const uint16_t* d_in, uint16_t* d_out, int depth, int height, int width, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Connected-component post-processing of the Otsu masks, batched per RoI (the step after Otsu: SURVEY 8f-2).
 * Crops are concatenated; crop r = voxels [d_offsets[r], d_offsets[r+1]) with dims d_dims[3r..3r+2] = (ez,ey,ex).
 *   m3d_cc_largest_batch   keep the largest 26-connected component (skimage.measure.label / cc3d default
 *                          connectivity; tools/binarization_soma.py:96-98, binarization_nuclei.py:125-130).
 *                          tie_last = 1: among equal sizes the highest label id (argsort(...)[-1], soma),
 *                          0: the lowest (np.argmax, nuclei).  invert = 1 runs on the complement and writes
 *                          255 everywhere except its largest component = hole filling (nuclei :132-137).
 *                          d_status[r] = 1 when there is no component at all, 2 for inconsistent dims / offsets,
 *                          3 if the label propagation did not converge within its sweep bound (output zeroed).
 *   m3d_binary_closing6_batch  scipy.ndimage.binary_closing defaults: 6-neighbourhood, 1 iteration, border 0
 *                          (binarization_nuclei.py:139).
 *   m3d_paint_instances    write d_ids[r] into the uint32 label volume wherever crop r (box int32 [R,6] inclusive)
 *                          is set and no SMALLER id covers the voxel == "paint only where still 0" in ascending
 *                          id order (binarization_soma.py:99-102).  The volume must be pre-filled with 0xFFFFFFFF.
 * ------------------------------------------------------------------------------------------------------- */
size_t m3d_cc_workspace_bytes(int64_t total_voxels);
int m3d_cc_largest_batch(const uint8_t* d_mask, const int64_t* d_offsets, const int32_t* d_dims, int num_rois,
                         int64_t total_voxels, int invert, int tie_last, uint8_t* d_out, int32_t* d_status,
                         void* d_ws, size_t ws_bytes, void* stream);
int m3d_binary_closing6_batch(const uint8_t* d_mask, const int64_t* d_offsets, const int32_t* d_dims, int num_rois,
                              int64_t total_voxels, uint8_t* d_out, void* d_ws, size_t ws_bytes, void* stream);
int m3d_paint_instances(const uint8_t* d_mask, const int64_t* d_offsets, const int32_t* d_boxes, const int32_t* d_ids,
                        int num_rois, int depth, int height, int width, uint32_t* d_volume, void* stream);
/* After the last m3d_paint_instances: sentinel 0xFFFFFFFF -> 0 and d_present[id] = 1 (uint8 [max_id + 1], caller zero-fills; may be
 * NULL) for every id in [1, max_id] that occurs in the volume (`mask_id in np.unique(seg)`, binarization_soma.py:103). */
int m3d_paint_finish(uint32_t* d_volume, int64_t num_voxels, int max_id, uint8_t* d_present, void* stream);
/* A tile's painting starts with ONE launch: the volume to the 0xFFFFFFFF sentinel, d_present [num_present] to 0 and the paint id of each
 * of the num_rois processed detections: d_ids[r] = d_idx[r] + first_id, or -1 (never paints) where the Otsu / component stage failed
 * (status != 0) or the detection's response map is all zero (d_map_stats: the workspace m3d_prm_quantize_windows*_u8 filled, 4 words
 * per map; null: not tested) - binarization_soma.py:74-76,94-98. */
int m3d_paint_begin(uint32_t* d_volume, int64_t num_voxels, uint8_t* d_present, int num_present, const int32_t* d_status_otsu,
                    const int32_t* d_status_cc, const int32_t* d_map_stats, const int64_t* d_idx, int num_rois, int first_id,
                    int32_t* d_ids, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M3D_H_ */
