/*
 * ver_ops.h -- C ABI of the MI355X (gfx950) kernels for VER's 2D->3D lifting path.
 *
 * Drop-in boundary.  The reference (DefaultRui/VLN-VER) is pure Python; the only native
 * entry points on its hot path are two symbols of the third-party mmcv-full 1.4.0 `_ext`
 * library, loaded at
 *   projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:11-12
 * and called at :118-124 (forward) and :150-160 (backward).  `ver_msda_forward` /
 * `ver_msda_backward` below are exactly those two calls with torch tensors replaced by raw
 * device pointers + sizes.  The remaining entry points replace Python/torch code of the
 * reference that sits on the same path (file:line cited per function) with fused kernels.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - buffers are dense, row-major, in the index order written in the comment;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it
 *     (no allocation, no synchronisation, no global state besides the last-error string);
 *   - return value: 0 on success, negative VER_E* on failure; `ver_last_error()` returns a
 *     thread-local description of the last failure;
 *   - dtype: fp32 everywhere (the reference forces fp32 on this op:
 *     multi_scale_deformable_attn_function.py:93 `custom_fwd(cast_inputs=torch.float32)`),
 *     except `value_dtype` where noted (VER_F32 = 0, VER_BF16 = 1: value stored as bf16).
 *     Arithmetic on VER_BF16 value: fp32 everywhere EXCEPT the forward gather's point
 *     accumulation, which is packed fp16 over the <= 8 points of a (voxel, head, corner) and
 *     fp32 after the corner fold -- range and precision contract at `ver_sca_forward`.
 */
#ifndef VER_OPS_H
#define VER_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VER_ABI_VERSION 29

#define VER_OK            0
#define VER_EINVAL       -1   /* bad argument (null pointer, non-positive size, ...) */
#define VER_EUNSUPPORTED -2   /* shape outside what the fused kernels are built for   */
#define VER_ELAUNCH      -3   /* HIP reported an error at launch                      */

#define VER_F32  0
#define VER_BF16 1

int         ver_abi_version(void);
const char* ver_last_error(void);

/* ---------------------------------------------------------------------------------------
 * mmcv op boundary: ext_module.ms_deform_attn_forward(value, value_spatial_shapes,
 *   value_level_start_index, sampling_locations, attention_weights, im2col_step)
 *   (multi_scale_deformable_attn_function.py:118-124).
 *
 *   value       f32 [B, num_keys, heads, head_dim]
 *   shapes_hw   i64 [levels, 2]   (h, w) per level            (device, like the reference)
 *   level_start i64 [levels]
 *   loc         f32 [B, Nq, heads, levels, points, 2]   (x, y) normalised to [0,1]
 *   attn_w      f32 [B, Nq, heads, levels, points]
 *   out         f32 [B, Nq, heads*head_dim]              (written, need not be zeroed)
 * out[b,q,h,:] = sum_{l,p} attn_w * bilinear(value_l[b,:,h,:], x = loc_x*W-0.5, y = loc_y*H-0.5),
 * zero outside the map.  `im2col_step` of the reference only batches launches and has no
 * numerical effect; it is accepted and ignored.
 */
int ver_msda_forward(const float* value, const int64_t* shapes_hw, const int64_t* level_start,
                     const float* loc, const float* attn_w, float* out,
                     int B, int num_keys, int heads, int head_dim, int levels, int points,
                     int Nq, int im2col_step, void* stream);

/* ext_module.ms_deform_attn_backward(value, shapes, level_start, loc, attn_w, grad_output,
 *   grad_value, grad_sampling_loc, grad_attn_weight, im2col_step)
 *   (multi_scale_deformable_attn_function.py:150-160).  As in the reference the three
 *   gradient buffers are caller-allocated and ZERO-INITIALISED by the caller (:146-148);
 *   grad_value is accumulated with atomics, grad_loc / grad_attn_w are written.
 */
int ver_msda_backward(const float* value, const int64_t* shapes_hw, const int64_t* level_start,
                      const float* loc, const float* attn_w, const float* grad_out,
                      float* grad_value, float* grad_loc, float* grad_attn_w,
                      int B, int num_keys, int heads, int head_dim, int levels, int points,
                      int Nq, int im2col_step, void* stream);

/* 3-D (trilinear) twin used by the detection decoder: the reference's in-tree
 *   voxel_multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations,
 *   attention_weights) (bevformer/modules/voxel_temporal_self_attention.py:275-335), called from
 *   VoxelCustomMSDeformableAttention.forward (bevformer/modules/voxel_decoder.py:312-313).
 *   shapes_dhw i64 [levels,3] = (D,H,W); loc f32 [B,Nq,heads,levels,points,3] = (x,y,z) in [0,1];
 *   flat key index = (z*H + y)*W + x; 5-D grid_sample semantics (pixel = loc*size - 0.5, zeros).
 *   Gradient buffers of the backward are caller-allocated and ZERO-INITIALISED.
 */
int ver_msda3d_forward(const float* value, const int64_t* shapes_dhw, const int64_t* level_start,
                       const float* loc, const float* attn_w, float* out,
                       int B, int num_keys, int heads, int head_dim, int levels, int points,
                       int Nq, void* stream);
int ver_msda3d_backward(const float* value, const int64_t* shapes_dhw, const int64_t* level_start,
                        const float* loc, const float* attn_w, const float* grad_out,
                        float* grad_value, float* grad_loc, float* grad_attn_w,
                        int B, int num_keys, int heads, int head_dim, int levels, int points,
                        int Nq, void* stream);

/* ---------------------------------------------------------------------------------------
 * Hit table: the per-viewpoint visibility structure shared by the three encoder layers.
 *   uv        f32 [B, Ncam, Nq, D, 2]  projected, normalised pixel coords (NOT clamped)
 *   vis       u8  [B, Nq]              bit c set <=> camera c sees voxel n (any anchor)
 *   vis_list  i32 [B, Ncam, Nq]        ascending voxel ids seen by camera c  (= the reference's
 *                                      `indexes[c]`, spatial_cross_attention.py:139-141)
 *   vis_cnt   i32 [B, Ncam]
 *   zero_list i32 [B, Nq]              ascending voxel ids NOT seen by exactly one camera: their
 *                                      output rows are zero-filled before the gather (unseen ->
 *                                      stay zero; seen by several cameras -> atomically summed)
 *   zero_cnt  i32 [B]
 *   fwd_list  i32 [B, Ncam, Nq]        work order of ver_sca_forward for camera c: the voxels ONLY camera c
 *                                      sees from the front (ascending), the voxels it shares with other
 *                                      cameras from the back (fwd_list[Nq-1], fwd_list[Nq-2], ...)
 *   fwd_cnt   i32 [B, Ncam, 2]         {#only-this-camera, #shared}; their sum is vis_cnt
 */

/* VoxelFormerEncoder.get_reference_points('3d') + point_sampling
 *   (bevformer/modules/voxel_encoder.py:54-83, 119-195) for B viewpoints at once, followed
 *   by the list construction above (replaces the per-camera nonzero() host syncs of
 *   spatial_cross_attention.py:139-142).  D = 1 (the reference never produces more anchors).
 *   world2pixel f32 [B, Ncam, 4, 4] row-major; origin f32 [B, 3];
 *   pc_range: HOST pointer to 6 floats (xmin,ymin,zmin,xmax,ymax,zmax);
 *   flat voxel index n = k*H*W + j*W + i.
 */
int ver_project_points(const float* world2pixel, const float* origin, const float* pc_range,
                       int B, int Ncam, int bev_z, int bev_h, int bev_w,
                       float img_w, float img_h,
                       float* uv, uint8_t* vis, int32_t* vis_list, int32_t* vis_cnt,
                       int32_t* zero_list, int32_t* zero_cnt, int32_t* fwd_list, int32_t* fwd_cnt,
                       void* stream);

/* Same lists from a caller-supplied mask in the reference's layout
 *   bev_mask u8/bool [Ncam, B, Nq, D]  (spatial_cross_attention.py:87,139-141,170).
 */
int ver_hits_from_mask(const uint8_t* bev_mask, int B, int Ncam, int Nq, int D,
                       uint8_t* vis, int32_t* vis_list, int32_t* vis_cnt,
                       int32_t* zero_list, int32_t* zero_cnt, int32_t* fwd_list, int32_t* fwd_cnt,
                       void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused multi-view gather = the body of SpatialCrossAttention.forward between the three
 * input projections and output_proj (spatial_cross_attention.py:139-173 together with
 * MSDeformableAttention3D.forward :345-398): per-camera re-batching, softmax over the
 * points, location arithmetic, bilinear sampling, scatter-add over cameras and division by
 * the camera count -- without padded rows or host syncs.  Rows of voxels seen by exactly one
 * camera are plain stores; rows seen by several cameras are zero-filled first and accumulated
 * with fp32 atomic adds, so the result is bitwise reproducible run to run as long as no voxel is
 * seen by more than two cameras (with three or more the last bit depends on the add order).
 *
 *   value   f32|bf16 [B, Ncam, map_h*map_w, heads, head_dim]   value_proj output
 *   offsets f32 [B, Nq, heads, points, 2]     sampling_offsets output, in pixels (one level)
 *   logits  f32 [B, Nq, heads, points]        attention_weights output, PRE-softmax
 *   slots   f32 [B, Nq, heads*head_dim]       (written; every row, unseen voxels get zeros)
 * slots[b,n] = (1/max(1,#cams seeing n)) * sum_{c sees n} sum_p softmax(logits)[p] *
 *              bilinear(value[b,c], uv[b,c,n,p%D] + offsets[n,p]/(map_w,map_h))
 * Supported: one feature level; head_dim in {8,16,32,64,96,128}; points in {4,8}; D | points;
 *   one (camera, head) value tile must fit the 160 KiB LDS: map_h*map_w*head_dim*4 <= 160 KiB
 *   in forward (<= 80 KiB keeps two workgroups per CU), twice that tile in backward.
 * Arithmetic and numerical contract, by value_dtype:
 *   VER_F32   fp32 throughout (<= 2e-5 against the fp32 reference formula).
 *   VER_BF16  with points == 8 and head_dim % 32 == 0 (the vocc.py shape) the staged tile is converted to fp16 in LDS
 *             and the <= 8 products weight * value of one (voxel, head, corner) are accumulated with packed fp16 FMAs;
 *             the corner fold result is converted to fp32 once and the camera sum / division are fp32.
 *             RANGE: the bf16 -> fp16 conversion is exact for |value| in [6.1e-5, 65504]; larger magnitudes SATURATE at
 *             +-65504 (round toward zero: no inf is produced, NaN stays NaN), smaller ones truncate into the fp16
 *             subnormals (absolute error < 6e-8); sample weights below ~6e-8 flush to zero.
 *             PRECISION: ~5e-4 of the partial sums (max |delta| 3.7e-3 at |slots| <= 1.2, 1e-3 relative L2 against
 *             fp32 accumulation of the same bf16 values; the north star's bf16 bound is 1e-2).
 *             VER_SCA_FWD_MATH=0 in the environment (read once per process) selects the exact form: bf16 tile
 *             unpacked per use, fp32 accumulation, <= 2e-5 like VER_F32.  Other shapes always use fp32 arithmetic.
 * flags: 0, or VER_SCA_ROWS_PREZEROED -- the caller has already zero-filled the rows `zero_list` names
 *   (`ver_sca_zero_rows(zero_list, zero_cnt, slots, B, Nq, heads*head_dim, side_stream)`, ordered before this call by
 *   an event): the fill depends on the hit table only, so it can run under the projections that precede the gather
 *   instead of in front of it.
 */
#define VER_SCA_ROWS_PREZEROED 1
/*        VER_SCA_VALUE_HEAD_MAJOR (ver_sca_forward and ver_sca_backward): `value` is laid out
 *   [heads, B, Ncam, map_h*map_w, head_dim] instead of the reference's [B, Ncam, map_h*map_w, heads, head_dim]: a (camera,
 *   head) tile is then ONE contiguous block of HBM (every 1-KB LDS-DMA request covers 8 whole 128-byte lines, none shared
 *   with another head's workgroup).  Only where `ver_sca_head_major_supported(...)` returns 1 (bf16 tiles, 8 points,
 *   head_dim % 32 == 0, 14x14 maps: the vocc.py shape); `grad_value` of ver_sca_backward keeps the REFERENCE layout
 *   either way (the weight gradient of value_proj reads it as a plain [rows, heads*head_dim] matrix).
 */
#define VER_SCA_VALUE_HEAD_MAJOR 2
/*        VER_SCA_GRAD_SLOTS_BF16 (ver_sca_backward): `grad_slots` is bf16 [B, Nq, heads*head_dim] -- what the GEMM behind the
 *   gather hands back under bf16 autocast -- and is used as it is, one bf16 term per element (the fp32 form is split into
 *   bf16 hi + lo); only where ver_sca_backward_grad_dtype(...) == VER_BF16, with grad_value_dtype = VER_BF16.
 */
#define VER_SCA_GRAD_SLOTS_BF16 4
int ver_sca_head_major_supported(int value_dtype, int head_dim, int points, int map_h, int map_w);
int ver_sca_zero_rows(const int32_t* zero_list, const int32_t* zero_cnt, float* slots, int B, int Nq, int row_floats,
                      void* stream);
int ver_sca_forward(const void* value, int value_dtype, const float* offsets, const float* logits,
                    const float* uv, const uint8_t* vis, const int32_t* vis_list,
                    const int32_t* vis_cnt, const int32_t* zero_list, const int32_t* zero_cnt,
                    const int32_t* fwd_list, const int32_t* fwd_cnt,
                    float* slots,
                    int B, int Ncam, int Nq, int D, int heads, int head_dim, int points,
                    int map_h, int map_w, int flags, void* stream);

/* Gradient of ver_sca_forward.
 *   grad_slots   f32 [B, Nq, heads*head_dim]
 *   grad_value   f32 | bf16 [B, Ncam, map_h*map_w, heads, head_dim]  (written in full); grad_value_dtype = VER_F32
 *                always works, VER_BF16 only where ver_sca_backward_grad_dtype() says so (the matrix-core path
 *                rounds its fp32 accumulators once, instead of a separate cast pass over the tensor)
 *   grad_offsets f32 [B, Nq, heads, points, 2]                (written in full)
 *   grad_logits  f32 [B, Nq, heads, points]                   (written in full; softmax bwd fused)
 * None of the outputs needs to be zeroed by the caller.
 */
int ver_sca_backward_grad_dtype(int value_dtype, int head_dim, int points, int map_h, int map_w);
int ver_sca_backward(const void* value, int value_dtype, const float* offsets, const float* logits,
                     const float* uv, const uint8_t* vis, const int32_t* vis_list,
                     const int32_t* vis_cnt, const int32_t* fwd_list, const int32_t* fwd_cnt,
                     const void* grad_slots,
                     void* grad_value, int grad_value_dtype, float* grad_offsets, float* grad_logits,
                     int B, int Ncam, int Nq, int D, int heads, int head_dim, int points,
                     int map_h, int map_w, int flags, void* stream);

/* ---------------------------------------------------------------------------------------
 * Data movement of the even-lattice coarse-to-fine upsample: the reference's three
 * ConvTranspose3d(768,768,(3,5,5),s=(1,2,2),p=(2,4,4),d=(2,2,2),op=(0,1,1))
 * (dense_heads/voxelformer_occupancy_head.py:251-258, applied at :560) are evaluated as
 * im2col + GEMM over the channels-last data lattice.
 *   src  f32|bf16 [B, Z, H, W, C]            channels-last lattice
 *   col  f32|bf16 [B*Z*H*W, ntaps, C]        col[(b,z,y,x), t, :] = src[b, z+dz_t, y+dy_t, x+dx_t, :]
 *                                             (zero outside the lattice)
 *   taps HOST pointer to ntaps*(dz,dy,dx) int32 offsets, ntaps <= 80
 * ver_lattice_col2im is the adjoint (gradient of im2col), in gather form (no atomics).
 */
int ver_lattice_im2col(const void* src, void* col, const int* taps, int ntaps,
                       int B, int Z, int H, int W, int C, int dtype, void* stream);
int ver_lattice_col2im(const void* grad_col, void* grad_src, const int* taps, int ntaps,
                       int B, int Z, int H, int W, int C, int dtype, void* stream);

/* Generalised lattice gather / scatter used by the parity-class upsample layers.  Rows enumerate
 * (b, zr < Zr, y, x); tap t reads the source position (zr + dz_t, y + dy_t, x + dx_t) of a lattice
 * with Zs z-layers (zero outside) into col[r * col_stride + col_offset[t] .. + C) (elements; offsets
 * and stride multiples of 16 bytes; columns between tap blocks are left untouched unless `const_rows` fills them);
 * ntaps <= 64.  Source layouts (`layout`):
 *   0 plain [B,Zs,H,W,C]      1 planar: four planes [B,Zs,H/2,W/2,C], plane 2*pm+pn = positions
 *   (2y'+pm, 2x'+pn) of the combined (H, W) lattice      2 z-split [B,2,H,W,2,C] (Zs = 4): element
 *   (b,z,y,x) at row (b, z&1, y, x), channel block z>>1      3 planar z-split: four planes of layout 2.
 * scatter is the adjoint (gather form, no atomics): grad_src is overwritten in the source's layout.
 */
/* const_rows (may be NULL): [Zr*H*W][const_blocks][const_width] elements, the constant-pattern blocks of one viewpoint's
 * rows; the gather copies them to columns const_offset[0 .. const_blocks) of every viewpoint (whole 16-byte vectors).
 */
int ver_lattice_gather(const void* src, void* col, const int* taps, const long* col_offset, long col_stride,
                       int ntaps, int B, int Zr, int Zs, int H, int W, int C, int layout, int dtype,
                       const void* const_rows, const long* const_offset, int const_blocks, int const_width, void* stream);
int ver_lattice_scatter(const void* grad_col, void* grad_src, const int* taps, const long* col_offset,
                        long col_stride, int ntaps, int B, int Zr, int Zs, int H, int W, int C, int layout, int dtype,
                        void* stream);

/* Layout changes of the coarse-to-fine head (LDS-tiled transposes):
 *   ver_convt_weight_forward : ConvTranspose3d weight f32 [pairs = Ci*Co][3*5*5] (the reference's
 *       parameter layout, head:251-258) -> correlation taps [75][pairs] in `dtype`, tap (a,b,c) =
 *       weight[.., 2-a, 4-b, 4-c];  _backward: gradient of the taps -> f32 gradient of the weight.
 *   ver_lattice_transpose    : even lattice channels-last (one of the four layouts of
 *       ver_lattice_gather) <-> channel-first rows cf[b*cf_stride + ((c*Z + z)*H + y)*W + x]
 *       (to_channel_first != 0: channels_last is read; else it is written).
 */
int ver_convt_weight_forward(const float* weight, void* taps, long pairs, int dtype, void* stream);
int ver_convt_weight_backward(const void* grad_taps, float* grad_weight, long pairs, int dtype, void* stream);
/*   ver_convt_weight_backward_blocks : the same adjoint taken straight from the weight gradients of a lattice layer's class
 *       GEMMs (no reference counterpart: the reference's ConvTranspose3d backward, head:251-258 through autograd): tap t
 *       is the fp32 sum of up to two [ci x co] blocks of `blocks` (row pitch `ld` elements, `dtype`) at element offsets
 *       block_offsets[2t], block_offsets[2t+1] (device int64 [75][2], -1 = none) plus prev_bias[ci] * grad_v[t*co + co']
 *       (both `dtype`, or both NULL) -> grad_weight f32 [ci*co][75], taps flipped as above.  bf16 with even co and ld:
 *       the offsets must be even too (4-byte loads; the layers' offsets row * ld + {0, co} are). */
int ver_convt_weight_backward_blocks(const void* blocks, const long* block_offsets, long ld, const void* prev_bias,
                                     const void* grad_v, float* grad_weight, int ci, int co, int dtype, void* stream);
/*   ver_convt_weight_forward_blocks (ABI 29): the forward twin -- the f32 ConvTranspose3d weight [ci*co][75] written STRAIGHT
 *       into the weight matrices the class GEMMs of a lattice layer read: tap t's [ci x co] block goes to element offsets
 *       block_offsets[2t], block_offsets[2t+1] (-1 = none) of `blocks` (row pitch ld, `dtype`): its "lower half" / "upper
 *       half" slots in the class-stacked [sum K_c, 2 co] matrix.  Replaces ver_convt_weight_forward + a concatenation with the
 *       constant rows + one row gather per parity class (dense_heads/upsample.py).  Even co / ld / offsets.
 *   ver_blocks_vec_forward / _backward (ABI 29): a row vector through every [ci x ncols] block of such a stacked matrix:
 *       vec[b][col] = sum_ci x[ci] * blocks[(block_rows[b] + ci) * ld + col]   (the previous layer's bias seen through every
 *       tap, head:251-258's bias-valued odd positions: v = b_prev^T K[t] for all taps in one pass over the bf16 weights),
 *       grad_x[ci] = sum_b sum_col blocks[(block_rows[b] + ci) * ld + col] * grad_vec[b][col].  x, grad_vec f32.  Both write
 *       PARTIAL results that the caller adds up in order (no atomics, bitwise reproducible): vec is f32 [8][nblocks][ncols]
 *       (8 slices of ci), grad_x f32 [nblocks][ci] (one row per block).  block_rows: device int64 [nblocks]. */
int ver_convt_weight_forward_blocks(const float* weight, const long* block_offsets, long ld, void* blocks, int ci, int co,
                                    int dtype, void* stream);
int ver_blocks_vec_forward(const void* blocks, const long* block_rows, int nblocks, long ld, int ci, int ncols, const float* x,
                           float* vec, int dtype, void* stream);
int ver_blocks_vec_backward(const void* blocks, const long* block_rows, int nblocks, long ld, int ci, int ncols,
                            const float* grad_vec, float* grad_x, int dtype, void* stream);
int ver_lattice_transpose(void* channels_last, void* channel_first, long cf_stride, int B, int Z, int H, int W,
                          int C, int layout, int to_channel_first, int dtype, void* stream);
/*   ver_lattice_rows : ver_lattice_transpose and ver_run_gather / _scatter (below) in ONE pass, bf16: the lattice
 *       (channels-last, one of the four layouts) <-> the gathered operand rows of occ_proj's GEMMs, when the runs of the
 *       raw .view (head:564) tile the flat channel-first lattice periodically: flat = k*quarter + row*period + off, the
 *       segment j with seg_off[j] <= off < seg_off[j] + seg_len[j] (host tables, nseg <= 8, the segments cover the
 *       period in order) is a pattern group whose rows live at rows[seg_base[j] + (b*seg_rows[j] + row)*seg_pitch[j]
 *       + k*seg_len[j] + (off - seg_off[j])] (elements).  to_rows != 0: the lattice is read; else it is written. */
int ver_lattice_rows(void* channels_last, void* rows, long quarter, int period, int nseg, const int* seg_off,
                     const int* seg_len, const long* seg_base, const int* seg_pitch, const int* seg_rows, int B, int Z,
                     int H, int W, int C, int layout, int to_rows, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused LayerNorm(128) + ReLU of the occupancy MLP (`occ_branches`, layers 1-2 and 4-5:
 * dense_heads/voxelformer_occupancy_head.py:241-248, applied at :580 to 504 000 rows per
 * viewpoint): y = relu((x - mean) * rstd * gamma + beta), eps as nn.LayerNorm (1e-5).
 *   x, y, grad_y, grad_x  f32|bf16 [N, 128]   (dtype = VER_F32 | VER_BF16; statistics in fp32)
 *   gamma, beta, grad_gamma, grad_beta f32 [128];  mean, rstd f32 [N] (written by forward)
 * backward writes grad_x in full and the two parameter gradients (zeroed inside).
 */
int ver_ln_relu_forward(const void* x, const float* gamma, const float* beta, void* y,
                        float* mean, float* rstd, long N, int W, float eps, int dtype, void* stream);
int ver_ln_relu_backward(const void* x, const void* grad_y, const float* gamma, const float* beta,
                         const float* mean, const float* rstd, void* grad_x,
                         float* grad_gamma, float* grad_beta, long N, int W, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Sigmoid focal loss of the occupancy term (dense_heads/voxelformer_occupancy_head.py:977-989 ->
 * mmdet 2.14 `FocalLoss(use_sigmoid=True)` / py_sigmoid_focal_loss, configured at
 * projects/configs/verformer/vocc.py:190-195), weight=None:
 *   logits f32|bf16 [N, C] (C % 8 == 0), target int64 [N] in [0, C] (C = background)
 *   forward : partial[b] = sum of the elementwise loss over the elements workgroup b visited,
 *             b < ver_focal_loss_blocks(N, C); the caller adds the partials (and applies
 *             loss_weight / avg_factor).  A target outside [0, C] (F.one_hot raises on it in the
 *             reference) makes the sum NaN and ORs 1 into *bad_labels (device int32, may be NULL;
 *             never cleared by the kernel): a caller that cleans NaNs out of its losses, as the
 *             head does, can still be loud about it without a synchronisation per step
 *   backward: grad[n,c] = scale[0] * d loss[n,c] / d logits[n,c]   (scale: device scalar; grad in
 *             the logits' dtype)
 *   arithmetic: fp32 per element.  f32 logits: log1p / exact division (1e-4 parity with the reference's
 *             fp32 formulas).  bf16 logits: the hardware exp2 / log2 / reciprocal (1 ulp each; exp(-|x|)
 *             below 2^-126 flushes to zero) -- the error stays far below the bf16 rounding of the inputs
 *             and of the gradient that is written back
 */
int ver_focal_loss_blocks(long N, int C);
int ver_focal_loss_forward(const void* logits, const int64_t* target, float* partial, long N, int C,
                           float gamma, float alpha, int dtype, int32_t* bad_labels, void* stream);
/*   forward_grad: the forward pass that ALSO writes the unscaled gradient d loss[n,c] / d logits[n,c] (logits' dtype) to
 *             `grad`, which may be the logits buffer itself -- for steps that need the loss and its gradient, not the logits */
int ver_focal_loss_forward_grad(const void* logits, const int64_t* target, float* partial, void* grad, long N, int C,
                                float gamma, float alpha, int dtype, int32_t* bad_labels, void* stream);
/* ... with the labels as bytes (C <= 254; ABI 28): what a caller that permutes and counts its labels as bytes hands over */
int ver_focal_loss_forward_grad_u8(const void* logits, const uint8_t* target, float* partial, void* grad, long N, int C,
                                   float gamma, float alpha, int dtype, int32_t* bad_labels, void* stream);
int ver_focal_loss_backward(const void* logits, const int64_t* target, const float* scale, void* grad,
                            long N, int C, float gamma, float alpha, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * y = LayerNorm(residual + dropout(a)): how both branches of the reference's VoxelFormerLayer end
 * (voxel_encoder.py:344-464 with operation_order cross_attn - norm - ffn - norm; SpatialCrossAttention.forward
 * returns `dropout(output_proj(slots)) + residual`, spatial_cross_attention.py:173-176; mmcv FFN returns
 * `identity + dropout(layers(x))`; a LayerNorm over C = embed_dims follows), one pass each way.
 *   a [N,C] f32 or bf16 (a_dtype), residual f32 [N,C], gamma / beta f32 [C]; C in {256, 512, 768, 1024}
 *   dropout: element i is kept iff hash(seed[0], i) < 1 - p_drop (seed: device int64; the backward pass recomputes
 *            the decision -- no mask tensor), kept values scaled by 1 / (1 - p_drop); p_drop = 0: no dropout, seed
 *            may be NULL.  (The reference's nn.Dropout draws from torch's generator: same distribution, different
 *            stream; parity vectors are taken in eval mode.)
 *   forward : y f32 [N,C], y_bf16 (optional, NULL to skip: the copy the next Linear reads under bf16 autocast),
 *             mean, rstd f32 [N]
 *   backward: grad_y f32 (+ grad_y_bf16, optional: the gradient that arrived through y_bf16) ->
 *             grad_a (a's dtype), grad_residual f32, grad_gamma / grad_beta f32 [C] (zeroed inside)
 */
int ver_add_ln_forward(const void* a, int a_dtype, const float* residual, const float* gamma, const float* beta,
                       const int64_t* seed, float p_drop, float eps, float* y, void* y_bf16, float* mean,
                       float* rstd, long N, int C, void* stream);
int ver_add_ln_backward(const float* grad_y, const void* grad_y_bf16, const void* a, int a_dtype,
                        const float* residual, const float* gamma, const float* mean, const float* rstd,
                        const int64_t* seed, float p_drop, void* grad_a, float* grad_residual, float* grad_gamma,
                        float* grad_beta, long N, int C, void* stream);
/*   y = dropout(relu(x)): the hidden activation of the layer's FFN (mmcv FFN: Linear - ReLU - Dropout - Linear), the
 *   same hash for the keep decision.  backward: grad_x = y > 0 ? grad_y / (1 - p_drop) : 0 (needs y only).
 *   n elements (a multiple of 4), dtype VER_F32 / VER_BF16; y may alias x.
 */
int ver_relu_dropout_forward(const void* x, void* y, const int64_t* seed, float p_drop, long n, int dtype, void* stream);
int ver_relu_dropout_backward(const void* y, const void* grad_y, void* grad_x, float p_drop, long n, int dtype,
                              void* stream);

/* ---------------------------------------------------------------------------------------
 * Run copies between the channel-first even lattice and the rows of the gathered `occ_proj` operand
 * (the reference's raw `.view(bs, Z, X, Y, C)` + `permute(0,2,3,1,4).flatten(3)` of the upsampled volume,
 * dense_heads/voxelformer_occupancy_head.py:564-570, restricted to the columns that are not constants).
 * A row (sample b, member i of a row group) is `runs` contiguous runs of `run_len` elements of sample b's image
 * starting at run_start[i*runs + k], followed by n_aug single elements image[aug_idx[i*n_aug + j]]:
 *   gather : rows[(b*n_rows + i)*row_elems + ...] <- image[b*image_stride + ...]       (forward operand)
 *   scatter: image[b*image_stride + run_start[..] + o] <- rows[(b*n_rows + i)*row_elems + k*run_len + o]
 *            (backward: every image element is written by exactly one row)
 * dtype VER_F32 / VER_BF16; run_len, run starts, image_stride and row_elems multiples of 8 bytes.
 */
int ver_run_gather(const void* image, long image_stride, const int32_t* run_start, const int32_t* aug_idx,
                   void* rows, int B, int n_rows, int runs, int run_len, int n_aug, int row_elems, int dtype,
                   void* stream);
int ver_run_scatter(const void* rows, void* image, long image_stride, const int32_t* run_start, int B,
                    int n_rows, int runs, int run_len, int row_elems, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused occupancy MLP = the reference's `occ_branches` Sequential
 * (dense_heads/voxelformer_occupancy_head.py:241-248; applied at :580):
 *   Linear(128,128) LayerNorm ReLU Linear(128,128) LayerNorm ReLU Linear(128,16)
 * bf16 operands, fp32 accumulation / LayerNorm statistics (the arithmetic of the reference's
 * layers under bf16 autocast); the hidden activations never leave the registers.
 *   pack    : W1, W2 f32 [128,128], W3 f32 [16,128] (nn.Linear layout [out,in]) -> `image`
 *             (ver_occ_mlp_image_bytes() bytes, 16-byte aligned): MFMA weight fragments
 *   vectors : f32 [ver_occ_mlp_vector_floats()] = b1 gamma1 beta1 b2 gamma2 beta2 (128 each) b3 (16)
 *   forward : x bf16 [N,128] -> logits bf16 [N,16]
 *   first_linear = 0: the first Linear has been folded into the producer of x (two Linears in a row compose:
 *             `occ_proj` :571 feeds `occ_branches[0]` :580 with nothing in between), x is ITS output and the chain
 *             starts at the first LayerNorm; W1 / b1 of image / vectors are ignored, grad_a1 may be NULL,
 *             grad_x = d loss / d x is then the gradient w.r.t. that output, and no dW1 is to be formed.
 *   ver_occ_mlp_forward takes `first_linear` as FLAGS: bit 0 as above, bit 1 = VER_OCC_MLP_CENTERED: the caller has
 *             centred the weights and biases of both hidden Linears over their OUTPUT axis (W <- W - mean_o W,
 *             b <- b - mean b; with first_linear = 0 the centring of Linear 1 sits in the producer of x), so every
 *             LayerNorm input has zero row mean and the kernel skips the mean pass.  LayerNorm is invariant to a
 *             per-row constant: LN(Wx + b) = LN(PWx + Pb), P = I - 11^T/128 -- the same function of the parameters;
 *             the caller maps the gradients of the centred parameters back through P (autograd does).
 */
#define VER_OCC_MLP_CENTERED 2
long ver_occ_mlp_image_bytes(void);
int ver_occ_mlp_vector_floats(void);
int ver_occ_mlp_pack(const float* W1, const float* W2, const float* W3, void* image, void* stream);
int ver_occ_mlp_forward(const void* x, const void* image, const float* vectors, void* logits,
                        long N, int width, int classes, float eps, int first_linear, void* stream);
/*   the same, also writing the reciprocal standard deviation of both LayerNorms per row (ABI 24):
 *     rstd f32 [N, 2]  (may be NULL: plain forward).  ver_occ_mlp_backward_fused_stats reads it back, so that its two
 *     recomputed LayerNorm-forward steps need no statistics (elementwise; VER_OCC_MLP_CENTERED rows only).
 */
int ver_occ_mlp_forward_stats(const void* x, const void* image, const float* vectors, void* logits, float* rstd,
                              long N, int width, int classes, float eps, int first_linear, void* stream);
/*   backward: re-computes the forward from x, then
 *     grad_x  bf16 [N,128]                      d loss / d x
 *     grad_a1, grad_a2 bf16 [N,128]             gradients w.r.t. the outputs of Linear 1 / Linear 2
 *     h1      bf16 [N,128]                      input of Linear 2 (post-ReLU activation)
 *     param_grads f32 [6*128 + 16*128 + 16]     d gamma1, d beta1, d b1, d gamma2, d beta2, d b2, then
 *                                               d W3 [16,128] and d b3 [16] (accumulated in-kernel); zeroed inside
 *   grad_a*, h1 are stored in FRAGMENT feature order: column 32t + 8g + j of a row holds feature
 *   32t + (j < 4 ? 4g + j : 16 + 4g + j - 4); the caller forms dW2 = grad_a2^T h1, dW1 = grad_a1^T x
 *   from them and un-permutes.
 */
int ver_occ_mlp_backward(const void* x, const void* grad_logits, const void* image, const float* vectors,
                         void* grad_x, void* grad_a1, void* grad_a2, void* h1,
                         float* param_grads, long N, int width, int classes, float eps, int first_linear,
                         void* stream);
/*   backward, folded first Linear (first_linear = 0), EVERYTHING accumulated in the kernel (ABI 22): reads x and
 *   grad_logits, writes grad_x; no side tensors.  W2 f32 [128,128], W3 f32 [16,128] are the raw nn.Linear weights
 *   (no image), `vectors` as above (b1 is ignored).
 *     param_grads f32 [6*128 + 16*128 + 16 + 128*128]   d gamma1, d beta1, (unused), d gamma2, d beta2, d b2, then
 *                                               d W3 [16,128], d b3 [16], d W2 [128,128] -- natural order; zeroed inside
 */
/*   flags: 0 or VER_OCC_MLP_CENTERED (the forward ran centred: W2 / b2 passed here are the centred ones, the recomputed
 *   LayerNorm-forward steps skip the mean pass).
 */
/*   grad_scale: NULL, or a DEVICE scalar that multiplies grad_logits as it is read (the gradient of the loss sum that
 *   `ver_focal_loss_forward_grad` left unscaled: the training loss then needs no backward pass over the logits).
 */
int ver_occ_mlp_backward_fused(const void* x, const void* grad_logits, const float* W2, const float* W3,
                               const float* vectors, void* grad_x, float* param_grads, long N, int width,
                               int classes, float eps, const float* grad_scale, int flags, void* stream);
/*   the same with the statistics ver_occ_mlp_forward_stats saved (rstd f32 [N, 2], may be NULL; used with
 *   VER_OCC_MLP_CENTERED only, ignored otherwise) */
int ver_occ_mlp_backward_fused_stats(const void* x, const void* grad_logits, const float* W2, const float* W3,
                                     const float* vectors, const float* rstd, void* grad_x, float* param_grads, long N,
                                     int width, int classes, float eps, const float* grad_scale, int flags, void* stream);

/* ---------------------------------------------------------------------------------------
 * Occupancy post-processing: VoxelFormerOccupancyHead.get_occupancy_prediction, focal-loss branch
 * (dense_heads/voxelformer_occupancy_head.py:1505-1540): sigmoid, the threshold as an extra last
 * ("empty") column, arg-max, sparse (voxel index, class) pairs of the occupied voxels.
 *   logits     f32|bf16 [N, C]  (C % 8 == 0, 16-byte aligned)
 *   block_work i32 [ver_occ_predict_blocks(N)]   scratch
 *   pairs      i64 [N, 2] capacity; rows [0, *count) are written: (row index, class), ascending row index
 *   count      i64 device scalar: number of occupied rows
 * Semantics of torch.argmax: the first of equal maxima wins (so a class probability EQUAL to the threshold is
 * occupied), NaN counts as the maximum.  Sigmoid is evaluated in fp32 (1 / (1 + exp(-x))).
 */
long ver_occ_predict_blocks(long N);
int  ver_occ_predict(const void* logits, int dtype, long N, int C, float threshold, int32_t* block_work,
                     int64_t* pairs, int64_t* count, void* stream);

/* ---------------------------------------------------------------------------------------
 * Weight gradient of the head's GEMM layers with ROWS on the contraction axis (ABI 24):
 *     out[Ka, N] = A[M, Ka]^T G[M, N]
 * A = the operand of the forward GEMM (tap matrix of a lattice layer / gathered occ_proj rows), G = the gradient of
 * its output; replaces the `a.t() @ g` of dense_heads/upsample.py::rows_tn, i.e. the d(weight) of the reference's
 * ConvTranspose3d stack and occ_proj (voxelformer_occupancy_head.py:251-258, :560, :571) as autograd forms it.
 *   a          bf16 [M, lda]  (the first Ka columns are used; a column range of a wider matrix is passed by pointer)
 *   g          bf16 [M, ldg]  (first N columns)
 *   out        bf16 | f32 [Ka, ldo]  (out_dtype VER_BF16 | VER_F32), written
 *   workspace  f32 [splits, Ka, N]: the row axis is split into `splits` chunks (0: ver_wgrad_tn_splits), every
 *              chunk's product stays fp32 until the chunks are added up (no bf16 rounding of partial sums)
 * Requirements: a, g 16-byte aligned, lda % 8 == 0, ldg % 8 == 0, N % 4 == 0, ldo % 4 == 0; any M (M = 0: zeros).
 * flags (experiments; 0 = default): bits 0-2 = prefetch distance in 16-row slabs, bit 3 = two slabs per phase.
 */
int  ver_wgrad_tn_splits(long M, int Ka, int N);
/*   the same choice knowing the widest row pitch ld = max(lda, ldg) in elements (ABI 29): a row chunk has to stay inside the
 *   4-GiB range of a buffer offset, so wide rows need more chunks than the default cap of 45 000 rows assumes */
int  ver_wgrad_tn_splits_ld(long M, int Ka, int N, long ld);
long ver_wgrad_tn_workspace(long M, int Ka, int N, int splits);
int  ver_wgrad_tn(const void* a, long lda, const void* g, long ldg, long M, int Ka, int N, void* out, long ldo,
                  int out_dtype, int splits, int flags, void* workspace, long workspace_bytes, void* stream);
/*   the same product with the IMPLICIT tap matrix of ver_gemm_nn_segments as A (ABI 29): out[Ka, N] = A^T g, A's columns =
 *   the segments in order (taps int [nseg][3]: (dz, dy, dx) = C columns, (-1 - k, 0, 0) = the cw columns of pattern block k of
 *   cst), rows = the cells of the combined (H, W) lattice, M = B 2 H W.  Segment widths must be multiples of 64 (a wave's
 *   64-column LDS-DMA piece lies inside one segment): C % 64 == 0, cw % 64 == 0.  2 H W < 65 536, the source lattice below
 *   2 GiB.  ver_wgrad_tn_segments_splits: the row-chunk count for `splits` = 0 (sizes the workspace f32 [splits][Ka][N]). */
int  ver_wgrad_tn_segments(const void* lattice, int layout, int B, int H, int W, int C, const int* taps, int nseg, const void* cst,
                           int ncst, int cw, const void* g, long ldg, int N, void* out, long ldo, int out_dtype, int splits,
                           void* workspace, long workspace_bytes, void* stream);
int  ver_wgrad_tn_segments_splits(int B, int H, int W, long Ka, int N, long ldg);

/* Forward product of the same layers (ABI 24):  c[M, N] = a[M, K] w[K, N] (+ bias[N]), bf16 in, fp32 accumulation, bf16 out.
 * Replaces the `torch.mm(a_mat[:, c0:c1], w, out=...)` / `addmm` of dense_heads/upsample.py and the `a @ wa.t()` of
 * occ_proj_lattice.py (with wa.t() materialised as [K, N]) -- the reference's ConvTranspose3d / occ_proj forward
 * (voxelformer_occupancy_head.py:560, :571) on the even lattice.
 *   a    bf16 [M, lda] (first K columns; K contiguous)      w  bf16 [K, ldw] (first N columns; N contiguous)
 *   bias f32 [N] or NULL                                     c  bf16 [M, ldc] (first N columns written)
 * Requirements: K % 32 == 0, K >= 64, a / w 16-byte aligned, lda % 8 == 0, ldw % 8 == 0; any M, N.  flags: 0.
 */
int  ver_gemm_nn(const void* a, long lda, const void* w, long ldw, const float* bias, void* c, long ldc, long M, int K,
                 int N, int flags, void* stream);
/*   the same product cut into K slices (ABI 29) for SKINNY operands -- the 450- and 1 800-row tap matrices of a
 *   one-viewpoint step (vocc.py:222 samples_per_gpu = 1) give 12-42 output tiles, a fraction of the chip: every slice's
 *   partial tile stays fp32 in `workspace` (f32 [splits][M][N]) and one pass adds them up, adds the bias and rounds once.
 *   ver_gemm_nn_splits: the slice count the library would pick (1: no workspace needed).  Needs N % 4 == 0, ldc % 4 == 0. */
/*   the same product with an IMPLICIT operand (ABI 29): the tap matrix of a Z = 4 lattice layer is never written -- row r of
 *   the product is cell (b, zl, y, x) of the combined (H, W) lattice (r = ((b 2 + zl) H + y) W + x; M = B 2 H W), its K axis
 *   `ntaps` blocks of C channels, block t = the channel vector of cell (zl + dz, y + dy, x + dx) of the source lattice
 *   (taps int [ntaps][3] = (dz in {0, 2}, dy, dx), HOST memory; zeros outside) -- what ver_lattice_gather would have copied,
 *   read straight from the lattice by the kernel's LDS-DMA.  layout: 0 plain [B,4,H,W,C], 2 z-split, 3 planar z-split (as
 *   ver_lattice_gather).  w bf16 [ntaps*C, ldw].  rowpos f32 [2 H W][N] or NULL: added to row r by its position r % (2 H W)
 *   (the constant-pattern columns of the explicit tap matrix times their weight rows); bias f32 [N] or NULL.
 *   Requirements: C % 32 == 0, ntaps <= 64, the source lattice below 2 GiB. */
int  ver_gemm_nn_taps(const void* lattice, int layout, int B, int H, int W, int C, const int* taps, int ntaps, const void* w,
                      long ldw, const float* rowpos, const float* bias, void* c, long ldc, int N, void* stream);
/*   ... with constant-pattern segments between the tap blocks: a taps entry (-1 - k, 0, 0) stands for the `cw` columns of
 *   block k of cst bf16 [2 H W][ncst][cw], which depend on the row's position r % (2 H W) only (the 0/1 patterns of the
 *   bias-valued odd input positions and the ones column of dense_heads/upsample.py's class layout
 *   [P00 | G1 | P10 | G2 | P11 | G3 | G4 | P01]); w then has sum-of-segment-widths rows, in segment order: the explicit
 *   tap matrix's column ranges and class weight matrices as they are.  cw % 32 == 0. */
int  ver_gemm_nn_segments(const void* lattice, int layout, int B, int H, int W, int C, const int* taps, int ntaps,
                          const void* cst, int ncst, int cw, const void* w, long ldw, const float* rowpos, const float* bias,
                          void* c, long ldc, int N, void* stream);
/*   ... reading its tap blocks from SEVERAL source lattices of one shape, plane_elems elements apart (tap t from plane
 *   tap_plane[t], host int [ntaps]; each plane below 2 GiB): d(input) of a lattice layer as ONE gather-form product over the four
 *   class planes of the output gradient, d_e[cell] = sum over (class, tap, half) of g_class[cell - tap] W^T -- no explicit
 *   d(tap matrix), no ver_lattice_scatter (the reference's ConvTranspose3d backward-data, head:251-258 through autograd). */
int  ver_gemm_nn_planes(const void* lattice, int layout, int B, int H, int W, int C, long plane_elems, int nplanes,
                        const int* tap_plane, const int* taps, int ntaps, const void* cst, int ncst, int cw, const void* w,
                        long ldw, const float* rowpos, const float* bias, void* c, long ldc, int N, void* stream);
int  ver_gemm_nn_splits(long M, int K, int N);
int  ver_gemm_nn_splitk(const void* a, long lda, const void* w, long ldw, const float* bias, void* c, long ldc, long M, int K,
                        int N, int splits, void* workspace, long workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Gradient clipping by the global L2 norm + AdamW, one call per step (ABI 27).  Replaces, for fp32 parameters, the
 * optimizer step the reference configures: mmcv OptimizerHook(grad_clip=dict(max_norm=..., norm_type=2)) in front of
 * torch.optim.AdamW (projects/configs/verformer/vocc.py:268-274) -- torch.nn.utils.clip_grad_norm_ (coefficient
 * min(1, max_norm / (norm + 1e-6))) followed by AdamW (decoupled weight decay, amsgrad off, bias corrections of `step`).
 *   table        device array of 4 * n_tensors pointers: parameters | gradients | exp_avg | exp_avg_sq (all f32, same sizes)
 *   sizes        device long [n_tensors]: elements per tensor
 *   chunk_tensor, chunk_index   device int [n_chunks]: the tensor of every chunk of `chunk_elems` elements (a multiple of 4)
 *                and the chunk's index inside that tensor
 *   partial      device float [n_chunks] scratch; norm_out: device float receiving the gradient norm BEFORE clipping, or NULL
 *   max_norm <= 0: no clipping.  Gradients are read, not rewritten.  step = 1 for the first update. */
int ver_clip_adamw_step(void* const* table, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                        int n_tensors, int n_chunks, int chunk_elems, float* partial, float* norm_out, float max_norm,
                        float lr, float beta1, float beta2, float eps, float weight_decay, long step, void* stream);

/* The same step with PER-TENSOR hyper-parameters and update counts (ABI 29; several parameter groups -- vocc.py:260-267
 * `paramwise_cfg` gives img_backbone its own lr -- and parameters that received their first gradient at different steps;
 * torch.optim.AdamW tracks `step` per parameter), everything the launch reads resident on the device so that a captured
 * hipGraph of the step replays correctly:
 *   hyper   device float [n_tensors][6] = lr, beta1, beta2, eps, weight_decay, (unused) of tensor t
 *   steps   device int   [n_tensors]    = updates applied to tensor t so far; this call adds 1 to every entry and uses the
 *                                         new value in the bias corrections 1 - beta^step
 * The clip norm is global over all tensors, as in clip_grad_norm_; a non-finite norm gives a NaN factor (torch.clamp). */
int ver_clip_adamw_step_tensors(void* const* table, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                                const float* hyper, int* steps, int n_tensors, int n_chunks, int chunk_elems, float* partial,
                                float* norm_out, float max_norm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VER_OPS_H */
