"""ctypes binding of libver_hip.so (C ABI: include/ver_ops.h) + the autograd wrappers.

There is NO fallback: if the library is missing or a tensor is not on the GPU these
functions raise.  PyTorch is used only for device memory, the current HIP stream and
autograd bookkeeping.
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

_PKG = os.path.dirname(os.path.abspath(__file__))
# (VER_HIP_LIB: another build of the same ABI, e.g. the host-ASan build libver_hip_asan.so of tests/test_abi_cpu.py)
LIB_PATH = os.environ.get('VER_HIP_LIB') or os.path.join(_PKG, 'libver_hip.so')
ABI_VERSION = 29
SYMBOLS = ('ver_abi_version', 'ver_last_error', 'ver_sca_backward_grad_dtype', 'ver_msda_forward', 'ver_msda_backward',
           'ver_project_points', 'ver_hits_from_mask', 'ver_sca_zero_rows', 'ver_sca_head_major_supported',
           'ver_sca_forward', 'ver_sca_backward',
           'ver_lattice_im2col', 'ver_lattice_col2im', 'ver_ln_relu_forward', 'ver_ln_relu_backward',
           'ver_msda3d_forward', 'ver_msda3d_backward', 'ver_focal_loss_blocks', 'ver_focal_loss_forward', 'ver_focal_loss_forward_grad', 'ver_focal_loss_forward_grad_u8',
           'ver_focal_loss_backward', 'ver_occ_mlp_image_bytes', 'ver_occ_mlp_vector_floats', 'ver_occ_mlp_pack',
           'ver_occ_mlp_forward', 'ver_occ_mlp_backward', 'ver_occ_mlp_backward_fused', 'ver_lattice_gather', 'ver_lattice_scatter',
           'ver_convt_weight_forward', 'ver_convt_weight_backward', 'ver_convt_weight_backward_blocks', 'ver_convt_weight_forward_blocks', 'ver_blocks_vec_forward', 'ver_blocks_vec_backward', 'ver_lattice_transpose', 'ver_lattice_rows', 'ver_run_gather',
           'ver_run_scatter', 'ver_add_ln_forward', 'ver_add_ln_backward',
           'ver_relu_dropout_forward', 'ver_relu_dropout_backward', 'ver_occ_predict_blocks', 'ver_occ_predict',
           'ver_wgrad_tn_splits', 'ver_wgrad_tn_splits_ld', 'ver_wgrad_tn_segments', 'ver_wgrad_tn_segments_splits', 'ver_wgrad_tn_workspace', 'ver_wgrad_tn', 'ver_occ_mlp_forward_stats',
           'ver_occ_mlp_backward_fused_stats', 'ver_gemm_nn', 'ver_gemm_nn_splits', 'ver_gemm_nn_splitk', 'ver_gemm_nn_taps', 'ver_gemm_nn_segments', 'ver_gemm_nn_planes', 'ver_clip_adamw_step', 'ver_clip_adamw_step_tensors')

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load libver_hip.so once; raise loudly when it is absent or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                '%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name in SYMBOLS:
            if not hasattr(handle, name):
                raise HipLibraryError('%s does not export %s' % (LIB_PATH, name))
        handle.ver_last_error.restype = ctypes.c_char_p
        handle.ver_occ_mlp_image_bytes.restype = ctypes.c_long
        handle.ver_occ_predict_blocks.restype = ctypes.c_long
        handle.ver_wgrad_tn_workspace.restype = ctypes.c_long
        if handle.ver_abi_version() != ABI_VERSION:
            raise HipLibraryError('libver_hip.so ABI %d != expected %d: rebuild'
                                  % (handle.ver_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = lib().ver_last_error()
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


class KernelTimer:
    """Optional HIP-event timing of every C-ABI launch (bench.py's roofline leg).  Events are
    recorded on the stream the kernel is launched on (torch's current stream)."""

    def __init__(self):
        self.records = []          # (name, start_event, end_event, meta)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, meta in self.records:
            d = out.setdefault(name, dict(count=0, ms=0.0, flops=0.0, meta=meta))
            d['count'] += 1
            d['ms'] += e0.elapsed_time(e1)
            if meta and 'flops' in meta:
                d['flops'] += meta['flops']
        return out


KERNEL_TIMER = None


def _launch(name, fn, meta=None):
    """Run ``fn`` (one C-ABI call) and check its return code; time it when a timer is set."""
    timer = KERNEL_TIMER
    if timer is None:
        return _check(fn(), name)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    timer.records.append((name, e0, e1, meta))
    return _check(rc, name)


class timed:
    """``with timed('head_gemm_fwd', flops): torch.mm(...)`` -- the same HIP-event bracket as ``_launch`` for work that is
    not a C-ABI call (the library GEMMs of the head), so that bench.py can put the dense part's achieved TFLOP/s next to
    the gather's GB/s.  Free when no timer is set."""

    def __init__(self, name, flops=0.0):
        self.name, self.flops = name, flops

    def __enter__(self):
        self.timer = KERNEL_TIMER
        if self.timer is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.timer.records.append((self.name, self.e0, e1, dict(flops=self.flops)))
        return False


def _p(t):
    return ctypes.c_void_p(t.data_ptr())       # NULL for empty tensors; the C side returns early


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gpu(t, name, dtype=None):
    if not t.is_cuda:
        raise RuntimeError('%s must be a GPU tensor: the VER ops only exist as HIP kernels' % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    return t.contiguous()


# ------------------------------------------------------------------------------------------
class MultiScaleDeformableAttnFunction_fp32(Function):
    """Same call signature and gradient contract as the reference's wrapper of the mmcv op
    (bevformer/modules/multi_scale_deformable_attn_function.py:90-163)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        value = _gpu(value, 'value', torch.float32)
        loc = _gpu(sampling_locations, 'sampling_locations', torch.float32)
        aw = _gpu(attention_weights, 'attention_weights', torch.float32)
        shapes = _gpu(value_spatial_shapes, 'value_spatial_shapes').to(torch.int64)
        lsi = _gpu(value_level_start_index, 'value_level_start_index').to(torch.int64)
        bs, nk, heads, hd = value.shape
        _, nq, _, nl, npt, _ = loc.shape
        ctx.im2col_step = im2col_step
        out = value.new_empty(bs, nq, heads * hd)
        _check(lib().ver_msda_forward(_p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), _p(out),
                                      bs, nk, heads, hd, nl, npt, nq, int(im2col_step), _stream()),
               'ver_msda_forward')
        ctx.save_for_backward(value, shapes, lsi, loc, aw)
        return out

    @staticmethod
    @once_differentiable
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        bs, nk, heads, hd = value.shape
        _, nq, _, nl, npt, _ = loc.shape
        grad_value = torch.zeros_like(value)
        grad_loc = torch.zeros_like(loc)
        grad_aw = torch.zeros_like(aw)
        go = _gpu(grad_output, 'grad_output').float().contiguous()
        _check(lib().ver_msda_backward(_p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), _p(go),
                                       _p(grad_value), _p(grad_loc), _p(grad_aw), bs, nk, heads, hd,
                                       nl, npt, nq, int(ctx.im2col_step), _stream()),
               'ver_msda_backward')
        return grad_value, None, None, grad_loc, grad_aw, None


# the reference selects the fp32 variant for every dtype (spatial_cross_attention.py:388-391)
MultiScaleDeformableAttnFunction_fp16 = MultiScaleDeformableAttnFunction_fp32


# ------------------------------------------------------------------------------------------
class HitTable:
    """Per-batch visibility structure (layout: include/ver_ops.h, "Hit table")."""

    __slots__ = ('uv', 'vis', 'vis_list', 'vis_cnt', 'zero_list', 'zero_cnt', 'fwd_list', 'fwd_cnt',
                 'B', 'Ncam', 'Nq', 'D')

    def __init__(self, B, Ncam, Nq, D, device):
        self.B, self.Ncam, self.Nq, self.D = B, Ncam, Nq, D
        self.uv = torch.empty(B, Ncam, Nq, D, 2, dtype=torch.float32, device=device)
        self.vis = torch.empty(B, Nq, dtype=torch.uint8, device=device)
        self.vis_list = torch.empty(B, Ncam, Nq, dtype=torch.int32, device=device)
        self.vis_cnt = torch.empty(B, Ncam, dtype=torch.int32, device=device)
        self.zero_list = torch.empty(B, Nq, dtype=torch.int32, device=device)
        self.zero_cnt = torch.empty(B, dtype=torch.int32, device=device)
        self.fwd_list = torch.empty(B, Ncam, Nq, dtype=torch.int32, device=device)
        self.fwd_cnt = torch.empty(B, Ncam, 2, dtype=torch.int32, device=device)

    def mask(self):
        """bool [Ncam, B, Nq, 1] in the reference's bev_mask layout (for inspection/tests)."""
        bits = torch.arange(self.Ncam, device=self.vis.device, dtype=torch.uint8)
        m = (self.vis[None] >> bits[:, None, None]) & 1
        return m.bool().unsqueeze(-1)


def project_points(world2pixel, origin, pc_range, bev_z, bev_h, bev_w, img_w=1280.0, img_h=1024.0):
    """get_reference_points('3d') + point_sampling + list building for B viewpoints
    (voxel_encoder.py:54-83,119-195).  world2pixel f32[B,Ncam,4,4], origin f32[B,3]."""
    w2p = _gpu(world2pixel, 'world2pixel', torch.float32)
    org = _gpu(origin, 'origin', torch.float32)
    B, ncam = w2p.shape[0], w2p.shape[1]
    nq = bev_z * bev_h * bev_w
    hit = HitTable(B, ncam, nq, 1, w2p.device)
    rng = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    _launch('ver_project_points', lambda: lib().ver_project_points(
        _p(w2p), _p(org), rng, B, ncam, bev_z, bev_h, bev_w, ctypes.c_float(img_w), ctypes.c_float(img_h),
        _p(hit.uv), _p(hit.vis), _p(hit.vis_list), _p(hit.vis_cnt), _p(hit.zero_list), _p(hit.zero_cnt),
        _p(hit.fwd_list), _p(hit.fwd_cnt), _stream()))
    return hit


def hits_from_mask(reference_points_cam, bev_mask):
    """Hit table from tensors in the reference's layout: reference_points_cam
    [Ncam,B,Nq,D,2], bev_mask [Ncam,B,Nq,D] (spatial_cross_attention.py:86-87)."""
    ncam, B, nq, D = bev_mask.shape
    mask = _gpu(bev_mask, 'bev_mask').to(torch.uint8).contiguous()
    hit = HitTable(B, ncam, nq, D, mask.device)
    hit.uv.copy_(reference_points_cam.to(torch.float32).permute(1, 0, 2, 3, 4))
    _check(lib().ver_hits_from_mask(_p(mask), B, ncam, nq, D, _p(hit.vis), _p(hit.vis_list),
                                    _p(hit.vis_cnt), _p(hit.zero_list), _p(hit.zero_cnt), _p(hit.fwd_list),
                                    _p(hit.fwd_cnt), _stream()),
           'ver_hits_from_mask')
    return hit


_SIDE_STREAMS = {}
SCA_PREZERO = os.environ.get('VER_SCA_PREZERO', '1') != '0'


def _side_stream(device):
    st = _SIDE_STREAMS.get(device)
    if st is None:
        st = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return st


class PreparedSlots:
    """Output buffer of one ``ver_sca_forward`` call whose zero fill is already under way on a side stream."""
    __slots__ = ('slots', 'done')

    def __init__(self, slots, done):
        self.slots, self.done = slots, done


def sca_prepare_slots(hit, row_floats):
    """Allocate the gather's output and zero-fill the rows ``hit.zero_list`` names on a SIDE stream
    (``ver_sca_zero_rows``).  The fill depends on the hit table only, so a caller that does this before the
    projections feeding the gather (value_proj, sampling_offsets / attention_weights) hides it under them instead
    of paying it in front of the gather.  Returns None when it cannot be overlapped safely (stream capture,
    VER_SCA_PREZERO=0): ``sca_gather`` then fills in line, as before."""
    if not SCA_PREZERO or torch.cuda.is_current_stream_capturing():
        return None
    dev = hit.vis.device
    slots = torch.empty(hit.B, hit.Nq, row_floats, dtype=torch.float32, device=dev)
    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
    # the block may still be in use by kernels queued on the main stream: the fill starts where the main stream is now
    side.wait_stream(main)
    slots.record_stream(side)       # (a buffer dropped before the gather consumed it must outlive the fill)
    with torch.cuda.stream(side):
        _launch('ver_sca_zero_rows', lambda: lib().ver_sca_zero_rows(
            _p(hit.zero_list), _p(hit.zero_cnt), _p(slots), hit.B, hit.Nq, row_floats, _stream()))
        done = torch.cuda.Event()
        done.record(side)
    return PreparedSlots(slots, done)


class SCAGatherFunction(Function):
    """slots = fused multi-view gather (ver_sca_forward / ver_sca_backward).

    ``value`` may be fp32 or bf16 (what ``value_proj`` emits under bf16 autocast).  fp32 tiles: fp32 arithmetic
    throughout, like the reference's fp32-forced op.  bf16 tiles at the vocc.py shape: the points of a (voxel, head,
    corner) are accumulated in packed fp16, everything after the corner fold in fp32 (contract: include/ver_ops.h)."""

    @staticmethod
    def forward(ctx, value, offsets, logits, hit, map_h, map_w, prepared=None, head_major=False, lowp_out=False):
        if value.dtype not in (torch.float32, torch.bfloat16):
            value = value.float()
        value = _gpu(value, 'value')
        offsets = _gpu(offsets, 'offsets').float().contiguous()
        logits = _gpu(logits, 'logits').float().contiguous()
        vdt = 1 if value.dtype == torch.bfloat16 else 0
        if head_major:                      # value [heads, B, Ncam, Nk, hd] (VER_SCA_VALUE_HEAD_MAJOR)
            heads, B, ncam, nk, hd = value.shape
        else:
            B, ncam, nk, heads, hd = value.shape
        points = logits.shape[-1]
        nq = hit.Nq
        assert nk == map_h * map_w and B == hit.B and ncam == hit.Ncam
        assert offsets.shape == (B, nq, heads, points, 2) and logits.shape == (B, nq, heads, points)
        flags = 2 if head_major else 0
        if prepared is not None:
            slots = prepared.slots
            assert slots.shape == (B, nq, heads * hd) and slots.dtype == torch.float32 and slots.is_contiguous()
            # (long done: it ran under the GEMMs.  With a KernelTimer the join is bracketed by events on the launch stream:
            #  what bench.py reports as the part of the zero fill that was NOT hidden)
            with timed('ver_sca_zero_wait'):
                torch.cuda.current_stream(slots.device).wait_event(prepared.done)
            with timed('ver_event_floor'):          # (an empty bracket: what two event records cost by themselves)
                pass
            flags |= 1                                                               # VER_SCA_ROWS_PREZEROED
        else:
            slots = torch.empty(B, nq, heads * hd, dtype=torch.float32, device=value.device)
        _launch('ver_sca_forward', lambda: lib().ver_sca_forward(
            _p(value), vdt, _p(offsets), _p(logits), _p(hit.uv), _p(hit.vis), _p(hit.vis_list),
            _p(hit.vis_cnt), _p(hit.zero_list), _p(hit.zero_cnt), _p(hit.fwd_list), _p(hit.fwd_cnt), _p(slots),
            B, ncam, nq, hit.D, heads, hd,
            points, map_h, map_w, flags, _stream()), meta=dict(prezeroed=bool(flags & 1), head_major=head_major))
        ctx.save_for_backward(value, offsets, logits)
        ctx.hit, ctx.map_hw, ctx.vdt, ctx.head_major = hit, (map_h, map_w), vdt, head_major
        if lowp_out and vdt == 1:
            # the consumer is a bf16 GEMM (output_proj under autocast): hand it the bf16 copy it would make anyway -- the
            # gradient then comes back in bf16 and ver_sca_backward reads it as it is (VER_SCA_GRAD_SLOTS_BF16) instead of
            # a cast kernel writing an fp32 copy for it first
            return slots.to(torch.bfloat16)
        return slots

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_slots):
        value, offsets, logits = ctx.saved_tensors
        hit = ctx.hit
        map_h, map_w = ctx.map_hw
        if ctx.head_major:
            heads, B, ncam, nk, hd = value.shape
        else:
            B, ncam, nk, heads, hd = value.shape
        points = logits.shape[-1]
        # the matrix-core backward rounds d(value) to bf16 itself (no separate cast pass over the tensor)
        gdt = lib().ver_sca_backward_grad_dtype(ctx.vdt, hd, points, map_h, map_w)
        gs = _gpu(grad_slots, 'grad_slots')
        gs_bf16 = gs.dtype == torch.bfloat16 and gdt == 1
        gs = gs.contiguous() if gs_bf16 else gs.float().contiguous()
        # d(value) is always written in the REFERENCE layout [B, Ncam, Nk, heads, hd]; for a head-major value it is handed
        # back as the permuted view of that buffer (same shape as value, no copy)
        g_value = torch.empty((B, ncam, nk, heads, hd), dtype=torch.bfloat16 if gdt == 1 else torch.float32, device=value.device)
        g_off = torch.empty_like(offsets)
        g_log = torch.empty_like(logits)
        _launch('ver_sca_backward', lambda: lib().ver_sca_backward(
            _p(value), ctx.vdt, _p(offsets), _p(logits), _p(hit.uv), _p(hit.vis), _p(hit.vis_list),
            _p(hit.vis_cnt), _p(hit.fwd_list), _p(hit.fwd_cnt), _p(gs), _p(g_value), gdt, _p(g_off), _p(g_log), B, ncam,
            hit.Nq, hit.D, heads,
            hd, points, map_h, map_w, (2 if ctx.head_major else 0) | (4 if gs_bf16 else 0), _stream()),
            meta=dict(grad_slots_bf16=gs_bf16))
        g_value = g_value.to(value.dtype)
        if ctx.head_major:
            g_value = g_value.permute(3, 0, 1, 2, 4)
        return g_value, g_off, g_log, None, None, None, None, None, None


def sca_gather(value, offsets, logits, hit, map_h, map_w, prepared=None, head_major=False, lowp_out=False):
    return SCAGatherFunction.apply(value, offsets, logits, hit, map_h, map_w, prepared, head_major, lowp_out)


def sca_head_major_supported(dtype, head_dim, points, map_h, map_w):
    """True where ``ver_sca_forward`` / ``ver_sca_backward`` read a head-major value tensor (contiguous tiles)."""
    return dtype == torch.bfloat16 and bool(lib().ver_sca_head_major_supported(1, head_dim, points, map_h, map_w))


class HeadMajorLinearFunction(Function):
    """``value_proj`` writing the head-major layout: x bf16 [M, C_in], weight [heads*hd, C_in], bias [heads*hd] ->
    [heads, M, hd]: one plain GEMM per head into its slab of the output (no stride-0 batch operand: those fault inside
    some hipBLASLt solutions when TunableOp tries them, dense_heads/row_linear.py).  The gradient comes
    back as the permuted view of a reference-layout [M, heads*hd] buffer (SCAGatherFunction.backward), so d(weight) is a
    plain split-row GEMM over [M, heads*hd]; d(x) likewise when it is needed."""

    @staticmethod
    def forward(ctx, x, weight, bias, heads):
        m, c_in = x.shape
        w = weight.to(x.dtype)
        hd = w.shape[0] // heads
        out = x.new_empty(heads, m, hd)
        b = bias.to(x.dtype).view(heads, hd)
        w3 = w.view(heads, hd, c_in)
        for h in range(heads):
            torch.addmm(b[h], x, w3[h].t(), out=out[h])
        ctx.save_for_backward(x, w)
        ctx.heads = heads
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        heads = ctx.heads
        m = x.shape[0]
        g2 = g.permute(1, 0, 2).reshape(m, -1)              # [M, heads*hd]: a view when g is the gather's permuted buffer
        from .dense_heads.upsample import rows_tn
        dw = rows_tn(g2, x) if ctx.needs_input_grad[1] else None        # [heads*hd, C_in]
        db = g2.sum(0, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        dx = g2 @ w if ctx.needs_input_grad[0] else None
        return dx, dw, db, None


def head_major_linear(x, weight, bias, heads):
    return HeadMajorLinearFunction.apply(x, weight, bias, heads)


# ------------------------------------------------------------------------------------------
class LatticeIm2colFunction(Function):
    """col = im2col(lattice) for the even-lattice upsample (ver_lattice_im2col / _col2im).
    lattice [B,Z,H,W,C] channels-last, fp32 or bf16 -> [B*Z*H*W, ntaps*C]."""

    @staticmethod
    def forward(ctx, lattice, taps):
        lattice = _gpu(lattice, 'lattice')
        if lattice.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError('lattice must be fp32 or bf16')
        B, Z, H, W, C = lattice.shape
        flat = [int(v) for t in taps for v in t]
        arr = (ctypes.c_int * len(flat))(*flat)
        dt = 1 if lattice.dtype == torch.bfloat16 else 0
        col = torch.empty(B * Z * H * W, len(taps) * C, dtype=lattice.dtype, device=lattice.device)
        _launch('ver_lattice_im2col', lambda: lib().ver_lattice_im2col(
            _p(lattice), _p(col), arr, len(taps), B, Z, H, W, C, dt, _stream()))
        ctx.geom = (B, Z, H, W, C, dt, arr, len(taps))
        return col

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_col):
        B, Z, H, W, C, dt, arr, ntaps = ctx.geom
        want = torch.bfloat16 if dt else torch.float32
        g = _gpu(grad_col, 'grad_col').to(want).contiguous()
        grad = torch.empty(B, Z, H, W, C, dtype=want, device=g.device)
        _launch('ver_lattice_col2im', lambda: lib().ver_lattice_col2im(
            _p(g), _p(grad), arr, ntaps, B, Z, H, W, C, dt, _stream()))
        return grad, None


def lattice_im2col(lattice, taps):
    return LatticeIm2colFunction.apply(lattice, taps)


class ConvTWeightFunction(Function):
    """ConvTranspose3d weight fp32 [Ci,Co,3,5,5] -> correlation taps [75,Ci,Co] (fp32 or bf16),
    ver_convt_weight_forward / _backward."""

    @staticmethod
    def forward(ctx, weight, dtype):
        weight = _gpu(weight, 'weight')
        if weight.dtype != torch.float32 or tuple(weight.shape[2:]) != (3, 5, 5):
            raise TypeError('weight must be fp32 [Ci,Co,3,5,5]')
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError('taps dtype must be fp32 or bf16')
        weight = weight.contiguous()
        ci, co = weight.shape[:2]
        taps = torch.empty(75, ci, co, dtype=dtype, device=weight.device)
        dt = 1 if dtype == torch.bfloat16 else 0
        _launch('ver_convt_weight_forward', lambda: lib().ver_convt_weight_forward(
            _p(weight), _p(taps), ctypes.c_long(ci * co), dt, _stream()))
        ctx.shape, ctx.dt, ctx.dtype = tuple(weight.shape), dt, dtype
        return taps

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_taps):
        g = _gpu(grad_taps, 'grad_taps').to(ctx.dtype).contiguous()
        ci, co = ctx.shape[:2]
        gw = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        _launch('ver_convt_weight_backward', lambda: lib().ver_convt_weight_backward(
            _p(g), _p(gw), ctypes.c_long(ci * co), ctx.dt, _stream()))
        return gw, None


def convt_weight_taps(weight, dtype):
    return ConvTWeightFunction.apply(weight, dtype)


def convt_weight_backward_blocks(blocks, block_offsets, prev_bias, grad_v, ci, co):
    """ver_convt_weight_backward_blocks (no autograd): the fp32 gradient [Ci,Co,3,5,5] of a ConvTranspose3d weight from the
    class-stacked weight gradients of a lattice layer: ``blocks`` [rows, ld] (fp32 or bf16, unit column stride); tap t is
    the sum of the [Ci x Co] blocks at element offsets ``block_offsets[t]`` (int64 [75,2] on the device, -1 = none) plus
    ``prev_bias[ci] * grad_v[t, co]`` (both in blocks' dtype, or both None)."""
    src = _gpu(blocks, 'blocks')
    if src.dim() != 2 or src.stride(1) != 1 or src.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('convt_weight_backward_blocks: blocks must be an fp32 / bf16 matrix with unit column stride')
    off = _gpu(block_offsets, 'block_offsets')
    if off.dtype != torch.int64 or tuple(off.shape) != (75, 2) or not off.is_contiguous():
        raise TypeError('convt_weight_backward_blocks: block_offsets must be a contiguous int64 [75, 2]')
    if (prev_bias is None) != (grad_v is None):
        raise ValueError('convt_weight_backward_blocks: prev_bias and grad_v come together')
    if prev_bias is not None:
        prev_bias = _gpu(prev_bias, 'prev_bias').to(src.dtype).contiguous()
        grad_v = _gpu(grad_v, 'grad_v').to(src.dtype).contiguous()
        if prev_bias.numel() != ci or tuple(grad_v.shape) != (75, co):
            raise ValueError('convt_weight_backward_blocks: prev_bias [Ci] and grad_v [75, Co] expected')
    gw = torch.empty(ci, co, 3, 5, 5, dtype=torch.float32, device=src.device)
    dt = 1 if src.dtype == torch.bfloat16 else 0
    _launch('ver_convt_weight_backward_blocks', lambda: lib().ver_convt_weight_backward_blocks(
        _p(src), _p(off), ctypes.c_long(src.stride(0)), _p(prev_bias) if prev_bias is not None else None,
        _p(grad_v) if grad_v is not None else None, _p(gw), int(ci), int(co), dt, _stream()))
    return gw


def convt_weight_forward_blocks(weight, block_offsets, blocks, ci, co):
    """ver_convt_weight_forward_blocks (no autograd): the fp32 ConvTranspose3d weight [Ci,Co,3,5,5] written into the
    class-stacked weight matrix ``blocks`` [rows, ld] (fp32 or bf16, unit column stride): tap t at element offsets
    ``block_offsets[t]`` (int64 [75,2] on the device, -1 = none).  Rows that hold no tap block are not touched."""
    w = _gpu(weight, 'weight')
    dst = _gpu(blocks, 'blocks')
    if w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != (ci, co, 3, 5, 5):
        raise TypeError('convt_weight_forward_blocks: weight must be a contiguous fp32 [Ci, Co, 3, 5, 5]')
    if dst.dim() != 2 or dst.stride(1) != 1 or dst.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('convt_weight_forward_blocks: blocks must be an fp32 / bf16 matrix with unit column stride')
    off = _gpu(block_offsets, 'block_offsets')
    if off.dtype != torch.int64 or tuple(off.shape) != (75, 2) or not off.is_contiguous():
        raise TypeError('convt_weight_forward_blocks: block_offsets must be a contiguous int64 [75, 2]')
    _launch('ver_convt_weight_forward_blocks', lambda: lib().ver_convt_weight_forward_blocks(
        _p(w), _p(off), ctypes.c_long(dst.stride(0)), _p(dst), int(ci), int(co), 1 if dst.dtype == torch.bfloat16 else 0, _stream()))
    return blocks


def _blocks_vec_args(blocks, block_rows, name):
    src = _gpu(blocks, 'blocks')
    if src.dim() != 2 or src.stride(1) != 1 or src.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('%s: blocks must be an fp32 / bf16 matrix with unit column stride' % name)
    rows = _gpu(block_rows, 'block_rows')
    if rows.dtype != torch.int64 or rows.dim() != 1 or not rows.is_contiguous():
        raise TypeError('%s: block_rows must be a contiguous int64 vector' % name)
    return src, rows


def blocks_vec_forward(blocks, block_rows, ci, x):
    """ver_blocks_vec_forward (no autograd): x fp32 [Ci] through every [Ci x ncols] block of ``blocks`` (first rows
    ``block_rows``, int64 on the device) -> fp32 [nblocks, ncols]."""
    src, rows = _blocks_vec_args(blocks, block_rows, 'blocks_vec_forward')
    x = _gpu(x, 'x').float().contiguous()
    if x.numel() != ci:
        raise ValueError('blocks_vec_forward: x must have Ci elements')
    part = torch.empty(8, rows.numel(), src.shape[1], dtype=torch.float32, device=src.device)      # 8 slices of ci
    _launch('ver_blocks_vec_forward', lambda: lib().ver_blocks_vec_forward(
        _p(src), _p(rows), int(rows.numel()), ctypes.c_long(src.stride(0)), int(ci), int(src.shape[1]), _p(x), _p(part),
        1 if src.dtype == torch.bfloat16 else 0, _stream()))
    return part.sum(0)


def blocks_vec_backward(blocks, block_rows, ci, grad_vec):
    """ver_blocks_vec_backward (no autograd): the adjoint of ``blocks_vec_forward`` in x: fp32 [Ci]."""
    src, rows = _blocks_vec_args(blocks, block_rows, 'blocks_vec_backward')
    gv = _gpu(grad_vec, 'grad_vec').float().contiguous()
    if tuple(gv.shape) != (rows.numel(), src.shape[1]):
        raise ValueError('blocks_vec_backward: grad_vec must be [nblocks, ncols]')
    part = torch.empty(rows.numel(), ci, dtype=torch.float32, device=src.device)                     # one row per block
    _launch('ver_blocks_vec_backward', lambda: lib().ver_blocks_vec_backward(
        _p(src), _p(rows), int(rows.numel()), ctypes.c_long(src.stride(0)), int(ci), int(src.shape[1]), _p(gv), _p(part),
        1 if src.dtype == torch.bfloat16 else 0, _stream()))
    return part.sum(0)


PLAIN, PLANAR, ZSPLIT, PLANAR_ZSPLIT = 0, 1, 2, 3      # lattice layouts of ver_lattice_gather / _transpose


def _lattice_dims(t, layout):
    """(B, Zs, C) of a lattice tensor in the given layout."""
    s = t.shape
    if layout == PLAIN:          # [B,Z,H,W,C]
        return s[0], s[1], s[4]
    if layout == PLANAR:         # [4,B,Z,H/2,W/2,C]
        return s[1], s[2], s[5]
    if layout == ZSPLIT:         # [B,2,H,W,2,C]
        return s[0], 4, s[5]
    return s[1], 4, s[6]         # [4,B,2,H/2,W/2,2,C]


def lattice_transpose(channels_last, channel_first, combined_hw, layout, to_channel_first):
    """ver_lattice_transpose (no autograd): channels_last lattice in one of the four layouts
    <-> channel_first [B, stride] rows holding [C,Z,H,W] at their start."""
    cl, cf = _gpu(channels_last, 'channels_last'), _gpu(channel_first, 'channel_first')
    if not (cl.is_contiguous() and cf.is_contiguous() and cl.dtype == cf.dtype):
        raise ValueError('lattice_transpose: contiguous buffers of one dtype required')
    if cl.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('lattice_transpose: fp32 or bf16')
    H, W = combined_hw
    B, Z, C = _lattice_dims(cl, int(layout))
    dt = 1 if cl.dtype == torch.bfloat16 else 0
    _launch('ver_lattice_transpose', lambda: lib().ver_lattice_transpose(
        _p(cl), _p(cf), ctypes.c_long(cf.shape[1]), B, Z, H, W, C, int(layout), int(to_channel_first), dt,
        _stream()))


def lattice_rows(channels_last, rows, row_map, combined_hw, layout, to_rows):
    """ver_lattice_rows (no autograd): bf16 lattice (channels-last, layout 0-3) <-> ``rows``, ONE flat bf16 buffer that
    holds the operand matrices of all pattern groups; ``row_map``: dict(quarter, period, seg_off, seg_len, seg_base,
    seg_pitch, seg_rows) of dense_heads/occ_proj_lattice.py (element offsets into ``rows``)."""
    cl, buf = _gpu(channels_last, 'channels_last'), _gpu(rows, 'rows')
    if not (cl.is_contiguous() and buf.is_contiguous() and cl.dtype == torch.bfloat16 and buf.dtype == torch.bfloat16):
        raise TypeError('lattice_rows: contiguous bf16 buffers required')
    H, W = combined_hw
    B, Z, C = _lattice_dims(cl, int(layout))
    n = len(row_map['seg_off'])
    need = max(b + B * r * p for b, r, p in zip(row_map['seg_base'], row_map['seg_rows'], row_map['seg_pitch']))
    if buf.numel() < need:
        raise ValueError('lattice_rows: rows buffer of %d elements, %d needed' % (buf.numel(), need))
    ints = lambda k: (ctypes.c_int * n)(*[int(v) for v in row_map[k]])
    base = (ctypes.c_long * n)(*[int(v) for v in row_map['seg_base']])
    _launch('ver_lattice_rows', lambda: lib().ver_lattice_rows(
        _p(cl), _p(buf), ctypes.c_long(int(row_map['quarter'])), int(row_map['period']), n, ints('seg_off'), ints('seg_len'),
        base, ints('seg_pitch'), ints('seg_rows'), B, Z, H, W, C, int(layout), int(to_rows), 1, _stream()))


def run_gather(image, run_start, aug_idx, rows, n_rows, run_len):
    """ver_run_gather (no autograd): image [B, stride] -> rows [B*n_rows, row_elems]; run_start int32 [n_rows, runs],
    aug_idx int32 [n_rows, n_aug] (indices into a sample's image row)."""
    img, out = _gpu(image, 'image'), _gpu(rows, 'rows')
    if not (img.is_contiguous() and out.is_contiguous() and img.dtype == out.dtype):
        raise ValueError('run_gather: contiguous buffers of one dtype required')
    if img.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('run_gather: fp32 or bf16')
    B = img.shape[0]
    runs, n_aug = run_start.shape[1], aug_idx.shape[1]
    _launch('ver_run_gather', lambda: lib().ver_run_gather(
        _p(img), ctypes.c_long(img.shape[1]), _p(run_start), _p(aug_idx), _p(out), B, n_rows, runs, run_len, n_aug,
        out.shape[1], 1 if img.dtype == torch.bfloat16 else 0, _stream()))


def run_scatter(rows, image, run_start, n_rows, run_len):
    """ver_run_scatter (no autograd): rows [B*n_rows, row_elems] -> the runs of image [B, stride]."""
    src, img = _gpu(rows, 'rows'), _gpu(image, 'image')
    if not (img.is_contiguous() and src.is_contiguous() and img.dtype == src.dtype):
        raise ValueError('run_scatter: contiguous buffers of one dtype required')
    if img.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('run_scatter: fp32 or bf16')
    B = img.shape[0]
    _launch('ver_run_scatter', lambda: lib().ver_run_scatter(
        _p(src), _p(img), ctypes.c_long(img.shape[1]), _p(run_start), B, n_rows, run_start.shape[1], run_len,
        src.shape[1], 1 if img.dtype == torch.bfloat16 else 0, _stream()))


def _tap_args(taps, col_offset):
    flat = [int(v) for t in taps for v in t]
    return (ctypes.c_int * len(flat))(*flat), (ctypes.c_long * len(col_offset))(*[int(o) for o in col_offset])


def lattice_gather(src, col, taps, col_offset, combined_hw, layout, row_z=None, const_rows=None, const_offset=None):
    """ver_lattice_gather (no autograd): src lattice in `layout` -> tap blocks of
    col [B*row_z*H*W, stride] at the given column offsets (row_z defaults to the source's z count).
    ``const_rows`` [row_z*H*W, n_blocks, width] + ``const_offset`` (n_blocks column offsets): constant-pattern blocks of
    one viewpoint's rows, written into every viewpoint's rows by the same kernel."""
    src, col = _gpu(src, 'src'), _gpu(col, 'col')
    if not (src.is_contiguous() and col.is_contiguous() and src.dtype == col.dtype):
        raise ValueError('lattice_gather: contiguous src / col of one dtype required')
    H, W = combined_hw
    B, Zs, C = _lattice_dims(src, int(layout))
    Zr = Zs if row_z is None else int(row_z)
    arr, offs = _tap_args(taps, col_offset)
    dt = 1 if src.dtype == torch.bfloat16 else 0
    cptr, coffs, nblk, cw = None, None, 0, 0
    if const_rows is not None:
        const_rows = _gpu(const_rows, 'const_rows')
        if not (const_rows.is_contiguous() and const_rows.dtype == col.dtype and const_rows.shape[0] == Zr * H * W):
            raise ValueError('lattice_gather: const_rows must be a contiguous [row_z*H*W, blocks, width] table of col.dtype')
        nblk, cw = int(const_rows.shape[1]), int(const_rows.shape[2])
        cptr, coffs = _p(const_rows), (ctypes.c_long * nblk)(*[int(o) for o in const_offset])
    _launch('ver_lattice_gather', lambda: lib().ver_lattice_gather(
        _p(src), _p(col), arr, offs, ctypes.c_long(col.shape[1]), len(taps), B, Zr, Zs, H, W, C, int(layout), dt,
        cptr, coffs, nblk, cw, _stream()))
    return col


def lattice_scatter(grad_col, grad_src, taps, col_offset, combined_hw, layout, row_z=None):
    """ver_lattice_scatter (no autograd): the adjoint of ``lattice_gather`` into grad_src (overwritten)."""
    grad_col, grad_src = _gpu(grad_col, 'grad_col'), _gpu(grad_src, 'grad_src')
    if not (grad_src.is_contiguous() and grad_col.is_contiguous() and grad_src.dtype == grad_col.dtype):
        raise ValueError('lattice_scatter: contiguous buffers of one dtype required')
    H, W = combined_hw
    B, Zs, C = _lattice_dims(grad_src, int(layout))
    Zr = Zs if row_z is None else int(row_z)
    arr, offs = _tap_args(taps, col_offset)
    dt = 1 if grad_src.dtype == torch.bfloat16 else 0
    _launch('ver_lattice_scatter', lambda: lib().ver_lattice_scatter(
        _p(grad_col), _p(grad_src), arr, offs, ctypes.c_long(grad_col.shape[1]), len(taps), B, Zr, Zs, H, W, C,
        int(layout), dt, _stream()))
    return grad_src


# ------------------------------------------------------------------------------------------
class LayerNormReluFunction(Function):
    """relu(layer_norm(x)) over rows of 128 channels (ver_ln_relu_forward / _backward);
    x fp32 or bf16 [..., 128], output in x's dtype."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _gpu(x, 'x')
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError('x must be fp32 or bf16')
        w = x.shape[-1]
        n = x.numel() // w
        gamma = _gpu(gamma, 'gamma').float().contiguous()
        beta = _gpu(beta, 'beta').float().contiguous()
        dt = 1 if x.dtype == torch.bfloat16 else 0
        y = torch.empty_like(x)
        mean = torch.empty(n, dtype=torch.float32, device=x.device)
        rstd = torch.empty(n, dtype=torch.float32, device=x.device)
        _launch('ver_ln_relu_forward', lambda: lib().ver_ln_relu_forward(
            _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), ctypes.c_long(n), w,
            ctypes.c_float(eps), dt, _stream()))
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.dt = dt
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        w = x.shape[-1]
        n = x.numel() // w
        gy = _gpu(grad_y, 'grad_y').to(x.dtype).contiguous()
        gx = torch.empty_like(x)
        gg = torch.empty(w, dtype=torch.float32, device=x.device)
        gb = torch.empty(w, dtype=torch.float32, device=x.device)
        _launch('ver_ln_relu_backward', lambda: lib().ver_ln_relu_backward(
            _p(x), _p(gy), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(gx), _p(gg), _p(gb),
            ctypes.c_long(n), w, ctx.dt, _stream()))
        return gx, gg, gb, None


def layer_norm_relu(x, gamma, beta, eps=1e-5):
    return LayerNormReluFunction.apply(x, gamma, beta, eps)


# ------------------------------------------------------------------------------------------
class LabelRangeFlag:
    """Sticky device-side "a label was outside [0, C]" flag of the fused focal loss, with an ASYNCHRONOUS host mirror:
    every fused call ORs into the device int (ver_focal_loss_forward) and queues a copy of it into pinned host memory;
    ``poll()`` reads the newest copy that has already arrived -- no device synchronisation -- and raises once it is
    set, so a bad label is reported one or two calls later, every call after that, whatever the caller does with the
    NaN loss in between (the head cleans NaNs out of its losses, as the reference does).  ``poll(sync=True)`` waits."""

    _per_device = {}

    def __init__(self, device):
        self.dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = None
        self.classes = None

    @classmethod
    def of(cls, device):
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        f = cls._per_device.get(key)
        if f is None:
            f = cls._per_device[key] = cls(device)
        return f

    def mirror(self, classes):
        if torch.cuda.is_current_stream_capturing():
            return
        self.classes = classes
        if self.event is not None and not self.event.query():
            return                                          # the previous copy is still in flight: do not overwrite it
        self.host.copy_(self.dev, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def poll(self, sync=False):
        if torch.cuda.is_current_stream_capturing():
            return                                          # (no event queries / host reads inside a capture)
        if sync:
            # always the blocking read: fused calls under stream capture never queued a mirror copy (mirror() is a
            # no-op there), and a queued mirror may predate the offending call
            self.host.copy_(self.dev)
        elif self.event is None or not self.event.query():
            return
        if int(self.host[0]) != 0:
            raise RuntimeError('FocalLoss: a target label outside [0, %s] reached the fused focal loss on %s (the loss of '
                               'that call was NaN; F.one_hot raises on it in the reference)'
                               % (self.classes, self.dev.device))

    def reset(self):
        self.dev.zero_()
        self.host.zero_()
        self.event = None


class SigmoidFocalLossSumFunction(Function):
    """sum over all elements of mmdet's sigmoid focal loss (ver_focal_loss_forward / _backward):
    logits fp32|bf16 [N, C] with C % 8 == 0, target int64 [N] in [0, C]; returns an fp32 scalar."""

    @staticmethod
    def forward(ctx, logits, target, gamma, alpha):
        logits = _gpu(logits, 'logits')
        if logits.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError('logits must be fp32 or bf16')
        logits = logits.contiguous()
        target = _gpu(target, 'target').to(torch.int64).contiguous()
        n, c = logits.shape
        if target.shape != (n,):
            raise ValueError('target must be [N]')
        dt = 1 if logits.dtype == torch.bfloat16 else 0
        blocks = lib().ver_focal_loss_blocks(ctypes.c_long(n), c)
        partial = torch.zeros(blocks, dtype=torch.float32, device=logits.device)
        flag = LabelRangeFlag.of(logits.device)
        flag.poll()                                          # a bad label of an EARLIER call is reported here
        _launch('ver_focal_loss_forward', lambda: lib().ver_focal_loss_forward(
            _p(logits), _p(target), _p(partial), ctypes.c_long(n), c, ctypes.c_float(gamma),
            ctypes.c_float(alpha), dt, _p(flag.dev), _stream()))
        flag.mirror(c)
        ctx.save_for_backward(logits, target)
        ctx.cfg = (gamma, alpha, dt)
        return partial.sum()

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        logits, target = ctx.saved_tensors
        gamma, alpha, dt = ctx.cfg
        n, c = logits.shape
        scale = _gpu(grad_out, 'grad_out').float().reshape(1).contiguous()
        grad = torch.empty_like(logits)
        _launch('ver_focal_loss_backward', lambda: lib().ver_focal_loss_backward(
            _p(logits), _p(target), _p(scale), _p(grad), ctypes.c_long(n), c, ctypes.c_float(gamma),
            ctypes.c_float(alpha), dt, _stream()))
        return grad, None, None, None


def sigmoid_focal_loss_sum(logits, target, gamma=2.0, alpha=0.25):
    return SigmoidFocalLossSumFunction.apply(logits, target, float(gamma), float(alpha))


# ------------------------------------------------------------------------------------------
def occ_mlp_pack(w1, w2, w3):
    """fp32 nn.Linear weights of ``occ_branches`` -> MFMA fragment image (ver_occ_mlp_pack)."""
    w1, w2, w3 = (_gpu(w, 'weight').detach().float().contiguous() for w in (w1, w2, w3))
    if w1.shape != (128, 128) or w2.shape != (128, 128) or w3.shape != (16, 128):
        raise ValueError('occ_mlp is built for Linear(128,128) x2 + Linear(128,16)')
    image = torch.empty(lib().ver_occ_mlp_image_bytes() // 2, dtype=torch.bfloat16, device=w1.device)
    _launch('ver_occ_mlp_pack', lambda: lib().ver_occ_mlp_pack(_p(w1), _p(w2), _p(w3), _p(image), _stream()))
    return image


def occ_mlp_vectors(b1, g1, be1, b2, g2, be2, b3):
    vec = torch.cat([_gpu(v, 'vector').detach().float().reshape(-1) for v in (b1, g1, be1, b2, g2, be2, b3)])
    if vec.numel() != lib().ver_occ_mlp_vector_floats():
        raise ValueError('occ_mlp vectors: expected 6x128 + 16 floats')
    return vec.contiguous()


# 1 (default): on centred rows the forward kernel saves 1/std of both LayerNorms per row (8 B on 288 B of traffic) and the
# wave-specialised backward kernel reads them back instead of recomputing the statistics; 0: recompute (round-4 form)
_OCC_MLP_SAVE_RSTD = os.environ.get('VER_OCC_MLP_SAVE_RSTD', '1') == '1'


def occ_mlp_forward(x, image, vectors, eps=1e-5, first_linear=True, centered=False, want_rstd=False):
    """x bf16 [..., 128] -> logits bf16 [..., 16] (ver_occ_mlp_forward).  ``first_linear=False``: x is already the
    output of the first Linear (folded into its producer).  ``centered``: the hidden Linears' weights / biases were
    centred over their output axis (VER_OCC_MLP_CENTERED): the LayerNorms skip the mean pass.  ``want_rstd``: returns
    ``(logits, rstd f32 [N, 2])`` -- 1/std of the two LayerNorms per row, for ver_occ_mlp_backward_fused_stats."""
    x = _gpu(x, 'x')
    if x.dtype != torch.bfloat16 or x.shape[-1] != 128:
        raise TypeError('x must be bf16 [..., 128]')
    x = x.contiguous()
    n = x.numel() // 128
    logits = torch.empty(x.shape[:-1] + (16,), dtype=torch.bfloat16, device=x.device)
    rstd = torch.empty(n, 2, dtype=torch.float32, device=x.device) if want_rstd else None
    _launch('ver_occ_mlp_forward', lambda: lib().ver_occ_mlp_forward_stats(
        _p(x), _p(image), _p(vectors), _p(logits), _p(rstd) if rstd is not None else None, ctypes.c_long(n), 128, 16,
        ctypes.c_float(eps), (1 if first_linear else 0) | (2 if centered else 0), _stream()))
    return (logits, rstd) if want_rstd else logits


def _occ_mlp_wants_rstd(folded, centered, n):
    return bool(_OCC_MLP_SAVE_RSTD and folded and centered and _OCC_MLP_BWD_FUSED and occ_mlp_backward_takes_grad_scale()
                and n < (1 << 28))


def _rows_tn(a, b, chunk=8000, with_colsum=True):
    """a^T b in fp32 for tall a [N,P], b [N,Q] (N ~ 1e7, P,Q <= 128) plus the column sums of a:
    the row dimension is split into chunks run as ONE batched GEMM (a plain GEMM would own a
    single output tile and run on one CU)."""
    n = a.shape[0]
    s = n // chunk
    main = s * chunk
    prod = a.new_zeros((a.shape[1], b.shape[1]), dtype=torch.float32)
    colsum = a.new_zeros((a.shape[1],), dtype=torch.float32)
    if s:
        a3 = a[:main].view(s, chunk, -1)
        prod += torch.bmm(a3.transpose(1, 2), b[:main].view(s, chunk, -1)).sum(0, dtype=torch.float32)
        if with_colsum:
            ones = a.new_ones((1, 1, chunk)).expand(s, 1, chunk)
            colsum += torch.bmm(ones, a3).sum((0, 1), dtype=torch.float32)
    if main < n:
        prod += (a[main:].t() @ b[main:]).float()
        if with_colsum:
            colsum += a[main:].sum(0, dtype=torch.float32)
    return prod, colsum


def occ_mlp_backward_takes_grad_scale():
    """True when ``ver_occ_mlp_backward_fused`` runs the wave-specialised kernel, the one that takes ``grad_scale``
    (VER_OCC_MLP_WS, read once by the library too: 0 selects the phase-locked N-split kernel, which rejects it)."""
    return os.environ.get('VER_OCC_MLP_WS', '1') not in ('0',)


_FRAG_ORDER = {}


def _frag_order(device):
    """inverse of the kernels' fragment feature order: inv[feature] = column."""
    key = str(device)
    if key not in _FRAG_ORDER:
        pos = torch.arange(128)
        t, g, j = pos // 32, (pos % 32) // 8, pos % 8
        feat = 32 * t + torch.where(j < 4, 4 * g + j, 16 + 4 * g + j - 4)
        inv = torch.empty(128, dtype=torch.long)
        inv[feat] = pos
        _FRAG_ORDER[key] = inv.to(device)
    return _FRAG_ORDER[key]


_OCC_MLP_BWD_FUSED = os.environ.get('VER_OCC_MLP_BWD_FUSED', '1') == '1'      # (0: row-split kernel + host GEMM for d(W2))


class OccMLPFunction(Function):
    """``occ_branches`` (head:241-248) as one fused kernel each way (ver_occ_mlp_*): x bf16 [N,128]
    -> logits bf16 [N,16].  Nothing but x is kept for the backward pass (the chain is re-computed)."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, eps, centered=False):
        x = _gpu(x, 'x').contiguous()
        # w1 is None: the first Linear was folded into the producer of x (two Linears in a row compose, see
        # VoxelFormerOccupancyHead.occupancy_from_volume); the kernels then run it as the identity and its weight
        # gradient -- a [128, N] x [N, 128] product over all rows -- is not formed here at all
        ctx.folded = w1 is None
        if ctx.folded:
            # (the folded kernels read W2 from W1's image sections: natural k order forward, natural-order output rows
            #  in the dgrad -- the layouts a chain that starts with a LayerNorm on the loaded rows needs)
            w1, b1 = w2, torch.zeros(128, device=x.device)
        image = occ_mlp_pack(w1, w2, w3)
        vec = occ_mlp_vectors(b1, g1, be1, b2, g2, be2, b3)
        ctx.eps, ctx.centered = eps, bool(centered)
        ctx.has_rstd = _occ_mlp_wants_rstd(ctx.folded, centered, x.numel() // 128) and any(ctx.needs_input_grad)
        out = occ_mlp_forward(x, image, vec, eps, first_linear=not ctx.folded, centered=centered, want_rstd=ctx.has_rstd)
        logits, rstd = out if ctx.has_rstd else (out, None)
        ctx.save_for_backward(x, image, vec, w2.detach(), w3.detach(), *((rstd,) if ctx.has_rstd else ()))
        return logits

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_logits):
        x, image, vec, w2, w3 = ctx.saved_tensors[:5]
        rstd = ctx.saved_tensors[5] if ctx.has_rstd else None
        shape = x.shape
        x2 = x.view(-1, 128)
        n = x2.shape[0]
        gl = _gpu(grad_logits, 'grad_logits').to(torch.bfloat16).contiguous().view(n, 16)
        gscale = None
        if ctx.folded and _OCC_MLP_BWD_FUSED:
            # N-split kernel: d(W2) and every other parameter gradient accumulated in the kernel, no side tensors
            gx = torch.empty_like(x2)
            pg = torch.empty(6 * 128 + 16 * 128 + 16 + 128 * 128, dtype=torch.float32, device=x.device)
            _launch('ver_occ_mlp_backward_fused', lambda: lib().ver_occ_mlp_backward_fused_stats(
                _p(x2), _p(gl), _p(w2.float().contiguous()), _p(w3.float().contiguous()), _p(vec),
                _p(rstd) if rstd is not None else None, _p(gx), _p(pg),
                ctypes.c_long(n), 128, 16, ctypes.c_float(ctx.eps), _p(gscale) if gscale is not None else None,
                2 if ctx.centered else 0, _stream()))
            vecs = pg[:768].view(6, 128)
            dw3 = pg[768:768 + 2048].view(16, 128)
            db3 = pg[768 + 2048:768 + 2048 + 16]
            dw2 = pg[768 + 2048 + 16:].view(128, 128)
            return (gx.view(shape), None, None, vecs[0], vecs[1], dw2, vecs[5], vecs[3], vecs[4], dw3, db3, None, None)
        gx, ga2, h1 = (torch.empty_like(x2) for _ in range(3))
        ga1 = None if ctx.folded else torch.empty_like(x2)
        pg = torch.empty(6 * 128 + 16 * 128 + 16, dtype=torch.float32, device=x.device)
        _launch('ver_occ_mlp_backward', lambda: lib().ver_occ_mlp_backward(
            _p(x2), _p(gl), _p(image), _p(vec), _p(gx), _p(ga1) if ga1 is not None else None, _p(ga2), _p(h1), _p(pg),
            ctypes.c_long(n), 128, 16, ctypes.c_float(ctx.eps), 0 if ctx.folded else 1, _stream()))
        inv = _frag_order(x.device)
        vecs = pg[:768].view(6, 128)
        dw3 = pg[768:768 + 2048].view(16, 128)
        db3 = pg[768 + 2048:]
        dw2, _ = _rows_tn(ga2, h1, with_colsum=False)
        if ctx.folded:                               # h1 comes back in natural feature order
            dw2 = dw2.index_select(0, inv)
            return (gx.view(shape), None, None, vecs[0], vecs[1], dw2, vecs[5], vecs[3], vecs[4], dw3, db3, None, None)
        dw2 = dw2.index_select(0, inv).index_select(1, inv)
        dw1, _ = _rows_tn(ga1, x2, with_colsum=False)
        dw1 = dw1.index_select(0, inv)
        return (gx.view(shape), dw1, vecs[2], vecs[0], vecs[1], dw2, vecs[5], vecs[3], vecs[4], dw3, db3, None, None)


class OccMLPFocalLossFunction(Function):
    """sum over all elements of the sigmoid focal loss of ``occ_branches(x)`` against integer targets -- the occupancy term
    of a TRAINING step that needs the loss and its gradient, not the logits (folded first Linear, fused bf16 kernels):
    forward = ``ver_occ_mlp_forward`` + ``ver_focal_loss_forward_grad``, which leaves the UNSCALED gradient of the loss sum
    in the logits buffer; backward = ``ver_occ_mlp_backward_fused`` reading that buffer with the incoming scalar as
    ``grad_scale``.  No backward pass of the focal loss over the [N, 16] tensor, no scaled copy of it."""

    @staticmethod
    def forward(ctx, x, g1, be1, w2, b2, g2, be2, w3, b3, target, eps, gamma, alpha, centered):
        x = _gpu(x, 'x').contiguous()
        image = occ_mlp_pack(w2, w2, w3)
        vec = occ_mlp_vectors(torch.zeros(128, device=x.device), g1, be1, b2, g2, be2, b3)
        ctx.has_rstd = _occ_mlp_wants_rstd(True, centered, x.numel() // 128)
        out = occ_mlp_forward(x, image, vec, eps, first_linear=False, centered=centered, want_rstd=ctx.has_rstd)
        logits, rstd = out if ctx.has_rstd else (out, None)
        l2 = logits.view(-1, 16)
        n = l2.shape[0]
        target = _gpu(target, 'target')
        as_bytes = target.dtype == torch.uint8                # (labels the caller has permuted / counted as bytes)
        target = (target if as_bytes else target.to(torch.int64)).contiguous()
        if target.shape != (n,):
            raise ValueError('target must be [N]')
        blocks = lib().ver_focal_loss_blocks(ctypes.c_long(n), 16)
        partial = torch.zeros(blocks, dtype=torch.float32, device=x.device)
        flag = LabelRangeFlag.of(x.device)
        flag.poll()
        entry = lib().ver_focal_loss_forward_grad_u8 if as_bytes else lib().ver_focal_loss_forward_grad
        _launch('ver_focal_loss_forward_grad', lambda: entry(
            _p(l2), _p(target), _p(partial), _p(l2), ctypes.c_long(n), 16, ctypes.c_float(gamma), ctypes.c_float(alpha),
            1, _p(flag.dev), _stream()))                      # (in place: the logits buffer now holds d loss / d logits)
        flag.mirror(16)
        ctx.save_for_backward(x, vec, w2.detach(), w3.detach(), l2, *((rstd,) if ctx.has_rstd else ()))
        ctx.eps, ctx.centered = eps, bool(centered)
        return partial.sum()

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, vec, w2, w3, gl = ctx.saved_tensors[:5]
        rstd = ctx.saved_tensors[5] if ctx.has_rstd else None
        shape = x.shape
        x2 = x.view(-1, 128)
        n = x2.shape[0]
        gscale = _gpu(grad_out, 'grad_out').float().reshape(1).contiguous()
        gx = torch.empty_like(x2)
        pg = torch.empty(6 * 128 + 16 * 128 + 16 + 128 * 128, dtype=torch.float32, device=x.device)
        _launch('ver_occ_mlp_backward_fused', lambda: lib().ver_occ_mlp_backward_fused_stats(
            _p(x2), _p(gl), _p(w2.float().contiguous()), _p(w3.float().contiguous()), _p(vec),
            _p(rstd) if rstd is not None else None, _p(gx), _p(pg),
            ctypes.c_long(n), 128, 16, ctypes.c_float(ctx.eps), _p(gscale), 2 if ctx.centered else 0, _stream()))
        vecs = pg[:768].view(6, 128)
        dw3 = pg[768:768 + 2048].view(16, 128)
        db3 = pg[768 + 2048:768 + 2048 + 16]
        dw2 = pg[768 + 2048 + 16:].view(128, 128)
        return (gx.view(shape), vecs[0], vecs[1], dw2, vecs[5], vecs[3], vecs[4], dw3, db3, None, None, None, None, None)


def occ_mlp_focal_loss_sum(x, g1, be1, w2, b2, g2, be2, w3, b3, target, eps=1e-5, gamma=2.0, alpha=0.25, centered=False):
    return OccMLPFocalLossFunction.apply(x, g1, be1, w2, b2, g2, be2, w3, b3, target, eps, float(gamma), float(alpha), centered)


def occ_mlp(x, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, eps=1e-5, centered=False):
    """``centered``: the caller passes hidden Linears whose weights / biases are centred over the output axis (and, with
    w1 None, has centred the folded first Linear in the producer of x): the forward LayerNorms skip the mean pass."""
    return OccMLPFunction.apply(x, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, eps, centered)


# ------------------------------------------------------------------------------------------
class AddDropoutLayerNormFunction(Function):
    """y = LayerNorm(residual + dropout(a)) as one pass each way (ver_add_ln_*): the tail of both branches of an
    encoder layer.  Returns (y fp32, y_bf16 or None): the bf16 copy is what the next Linear reads under autocast."""

    @staticmethod
    def forward(ctx, a, residual, gamma, beta, p_drop, eps, want_bf16):
        a = _gpu(a, 'a')
        if a.dtype not in (torch.float32, torch.bfloat16):
            a = a.float()
        a = a.contiguous()
        res = _gpu(residual, 'residual').float().contiguous()
        C = a.shape[-1]
        n = a.numel() // C
        y = torch.empty_like(res)
        y16 = torch.empty(res.shape, dtype=torch.bfloat16, device=res.device) if want_bf16 else None
        mean = torch.empty(n, dtype=torch.float32, device=res.device)
        rstd = torch.empty_like(mean)
        seed = torch.randint(0, 2 ** 62, (1,), device=res.device, dtype=torch.int64) if p_drop > 0 else None
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        _launch('ver_add_ln_forward', lambda: lib().ver_add_ln_forward(
            _p(a), 1 if a.dtype == torch.bfloat16 else 0, _p(res), _p(g), _p(b), _p(seed) if seed is not None else None,
            ctypes.c_float(p_drop), ctypes.c_float(eps), _p(y), _p(y16) if y16 is not None else None, _p(mean), _p(rstd),
            ctypes.c_long(n), C, _stream()))
        ctx.save_for_backward(a, res, g, mean, rstd, seed if seed is not None else torch.empty(0, device=res.device))
        ctx.p_drop, ctx.has_y16 = p_drop, want_bf16
        if want_bf16:
            return y, y16
        return y, None

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y, grad_y16):
        a, res, g, mean, rstd, seed = ctx.saved_tensors
        C = a.shape[-1]
        n = a.numel() // C
        gy = (torch.zeros_like(res) if grad_y is None else _gpu(grad_y, 'grad_y').float().contiguous())
        gy16 = None
        if ctx.has_y16 and grad_y16 is not None:
            gy16 = _gpu(grad_y16, 'grad_y_bf16').to(torch.bfloat16).contiguous()
        d_a = torch.empty_like(a)
        d_res = torch.empty_like(res)
        dg = torch.empty(C, dtype=torch.float32, device=res.device)
        db = torch.empty_like(dg)
        _launch('ver_add_ln_backward', lambda: lib().ver_add_ln_backward(
            _p(gy), _p(gy16) if gy16 is not None else None, _p(a), 1 if a.dtype == torch.bfloat16 else 0, _p(res), _p(g),
            _p(mean), _p(rstd), _p(seed) if ctx.p_drop > 0 else None, ctypes.c_float(ctx.p_drop), _p(d_a), _p(d_res),
            _p(dg), _p(db), ctypes.c_long(n), C, _stream()))
        return d_a, d_res, dg, db, None, None, None


class ReluDropoutFunction(Function):
    """y = dropout(relu(x)) in one pass (ver_relu_dropout_*); the backward pass needs y only."""

    @staticmethod
    def forward(ctx, x, p_drop):
        x = _gpu(x, 'x')
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        y = torch.empty_like(x)
        seed = torch.randint(0, 2 ** 62, (1,), device=x.device, dtype=torch.int64) if p_drop > 0 else None
        _launch('ver_relu_dropout_forward', lambda: lib().ver_relu_dropout_forward(
            _p(x), _p(y), _p(seed) if seed is not None else None, ctypes.c_float(p_drop), ctypes.c_long(x.numel()),
            1 if x.dtype == torch.bfloat16 else 0, _stream()))
        ctx.save_for_backward(y)
        ctx.p_drop = p_drop
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        y, = ctx.saved_tensors
        gy = _gpu(grad_y, 'grad_y').to(y.dtype).contiguous()
        gx = torch.empty_like(y)
        _launch('ver_relu_dropout_backward', lambda: lib().ver_relu_dropout_backward(
            _p(y), _p(gy), _p(gx), ctypes.c_float(ctx.p_drop), ctypes.c_long(y.numel()),
            1 if y.dtype == torch.bfloat16 else 0, _stream()))
        return gx, None


def relu_dropout(x, p_drop=0.0):
    """dropout(relu(x)), x fp32 or bf16 with a multiple of 4 elements."""
    return ReluDropoutFunction.apply(x, float(p_drop))


def add_dropout_layer_norm(a, residual, gamma, beta, p_drop=0.0, eps=1e-5, want_bf16=False):
    """(y fp32, y_bf16 | None) = LayerNorm(residual + dropout(a)); C = a.shape[-1] in {256, 512, 768, 1024}."""
    return AddDropoutLayerNormFunction.apply(a, residual, gamma, beta, float(p_drop), float(eps), bool(want_bf16))


# ------------------------------------------------------------------------------------------
class VoxelMSDeformAttnFunction(Function):
    """3-D (trilinear) deformable sampling of the detection decoder
    (voxel_temporal_self_attention.py:275-335) on ver_msda3d_forward / _backward."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
        value = _gpu(value, 'value').float().contiguous()
        loc = _gpu(sampling_locations, 'sampling_locations').float().contiguous()
        aw = _gpu(attention_weights, 'attention_weights').float().contiguous()
        shapes = _gpu(spatial_shapes, 'spatial_shapes').to(torch.int64).contiguous()
        lsi = _gpu(level_start_index, 'level_start_index').to(torch.int64).contiguous()
        bs, nk, heads, hd = value.shape
        _, nq, _, nl, npt, _ = loc.shape
        out = value.new_empty(bs, nq, heads * hd)
        _launch('ver_msda3d_forward', lambda: lib().ver_msda3d_forward(
            _p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), _p(out), bs, nk, heads, hd, nl, npt, nq, _stream()))
        ctx.save_for_backward(value, shapes, lsi, loc, aw)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        bs, nk, heads, hd = value.shape
        _, nq, _, nl, npt, _ = loc.shape
        gv, gl, ga = torch.zeros_like(value), torch.zeros_like(loc), torch.zeros_like(aw)
        go = _gpu(grad_output, 'grad_output').float().contiguous()
        _launch('ver_msda3d_backward', lambda: lib().ver_msda3d_backward(
            _p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), _p(go), _p(gv), _p(gl), _p(ga), bs, nk, heads,
            hd, nl, npt, nq, _stream()))
        return gv, None, None, gl, ga


def voxel_msda(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    return VoxelMSDeformAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations,
                                           attention_weights)


# ------------------------------------------------------------------------------------------
def occ_predict(logits, threshold=0.25):
    """Sparse occupancy prediction (ver_occ_predict; head:1505-1540): logits fp32|bf16 [N, C] ->
    int64 [K, 2] pairs (row index, class) of the rows whose best class probability reaches ``threshold``,
    in ascending row order.  One device->host read of K, as the reference's ``torch.where``."""
    logits = _gpu(logits, 'logits')
    if logits.dtype not in (torch.float32, torch.bfloat16):
        logits = logits.float()
    logits = logits.contiguous()
    n, c = logits.shape
    dt = 1 if logits.dtype == torch.bfloat16 else 0
    nb = lib().ver_occ_predict_blocks(ctypes.c_long(n))
    work = torch.empty(max(nb, 1), dtype=torch.int32, device=logits.device)
    pairs = torch.empty(max(n, 1), 2, dtype=torch.int64, device=logits.device)
    count = torch.empty(1, dtype=torch.int64, device=logits.device)
    _launch('ver_occ_predict', lambda: lib().ver_occ_predict(
        _p(logits), dt, ctypes.c_long(n), c, ctypes.c_float(threshold), _p(work), _p(pairs), _p(count), _stream()))
    return pairs[:int(count.item())]


# ------------------------------------------------------------------------------------------
def wgrad_tn_supported(a, g):
    """Shapes / strides ``wgrad_tn`` takes: bf16 GPU matrices, unit column stride, 16-byte aligned rows."""
    return (a.is_cuda and g.is_cuda and a.dtype == torch.bfloat16 and g.dtype == torch.bfloat16 and a.dim() == 2
            and g.dim() == 2 and a.shape[0] == g.shape[0] and a.stride(1) == 1 and g.stride(1) == 1
            and a.stride(0) % 8 == 0 and g.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0
            and g.shape[1] % 4 == 0)


def wgrad_tn(a, g, out_dtype=None, splits=0, flags=0, out=None):
    """``a.t() @ g`` for tall bf16 operands with the ROWS on the contraction axis (ver_wgrad_tn): a [M, Ka] (may be a
    column range of a wider row-major matrix), g [M, N] -> [Ka, N] in ``out_dtype`` (default: a's dtype; ``out``: a matrix
    of that dtype to write into, e.g. a row range of a stacked buffer), fp32 accumulation over all rows.  The weight gradient of a lattice layer / of occ_proj (dense_heads/upsample.py::rows_tn)."""
    if not wgrad_tn_supported(a, g):
        raise RuntimeError('wgrad_tn: unsupported operands %s %s / %s %s' % (tuple(a.shape), a.stride(), tuple(g.shape), g.stride()))
    m, ka = a.shape
    n = g.shape[1]
    out_dtype = out_dtype or a.dtype
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise TypeError('wgrad_tn: out_dtype must be bf16 or fp32')
    L = lib()
    if splits <= 0:
        splits = L.ver_wgrad_tn_splits_ld(ctypes.c_long(m), ka, n, ctypes.c_long(max(a.stride(0), g.stride(0))))
    nbytes = L.ver_wgrad_tn_workspace(ctypes.c_long(m), ka, n, splits)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device)
    if out is None:
        out = torch.empty(ka, n, dtype=out_dtype, device=a.device)
    elif not (out.is_cuda and out.shape == (ka, n) and out.dtype == out_dtype and out.stride(1) == 1):
        raise RuntimeError('wgrad_tn: out must be a [Ka, N] GPU matrix of out_dtype with unit column stride')
    _launch('ver_wgrad_tn', lambda: L.ver_wgrad_tn(
        _p(a), ctypes.c_long(a.stride(0)), _p(g), ctypes.c_long(g.stride(0)), ctypes.c_long(m), ka, n, _p(out),
        ctypes.c_long(out.stride(0)), 1 if out_dtype == torch.bfloat16 else 0, int(splits), int(flags), _p(ws), ctypes.c_long(nbytes),
        _stream()), meta=dict(flops=2.0 * m * ka * n))
    return out


def gemm_nn_supported(a, w):
    """Shapes / strides ``gemm_nn`` takes: bf16 GPU matrices, unit column stride, 16-byte aligned rows, K % 32 == 0."""
    return (a.is_cuda and w.is_cuda and a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.dim() == 2
            and w.dim() == 2 and a.shape[1] == w.shape[0] and a.stride(1) == 1 and w.stride(1) == 1
            and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and a.shape[1] % 32 == 0 and a.shape[1] >= 64)


def gemm_nn_splits(m, k, n):
    """K slices ``gemm_nn`` would cut an [m, k] x [k, n] product into (1: one pass, no workspace): > 1 for skinny operands."""
    return int(lib().ver_gemm_nn_splits(ctypes.c_long(m), int(k), int(n)))


def gemm_nn(a, w, bias=None, out=None, splits=None, timer_class='ver_gemm_nn'):
    """``a @ w (+ bias)`` on ver_gemm_nn: a bf16 [M, K] (may be a column range of a wider row-major matrix), w bf16 [K, N]
    row-major, bias fp32 [N] or None -> bf16 [M, N] (``out``: a bf16 matrix with unit column stride to write into).
    ``splits``: K slices (None: the library's choice -- 1 except for skinny operands; fp32 partial tiles, added up once).
    ``timer_class``: the name a KernelTimer files the launch under (bench.py's classes of the head's products)."""
    if not gemm_nn_supported(a, w):
        raise RuntimeError('gemm_nn: unsupported operands %s %s / %s %s' % (tuple(a.shape), a.stride(), tuple(w.shape), w.stride()))
    m, k = a.shape
    n = w.shape[1]
    if out is None:
        out = torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    if out.shape != (m, n) or out.dtype != torch.bfloat16 or out.stride(1) != 1 or not out.is_cuda:
        raise RuntimeError('gemm_nn: out must be a bf16 [M, N] GPU matrix with unit column stride')
    if bias is not None:
        bias = _gpu(bias, 'bias').float().contiguous()
    if splits is None:
        splits = gemm_nn_splits(m, k, n)
    if splits > 1 and (n % 4 or out.stride(0) % 4 or out.data_ptr() % 8):
        splits = 1
    ws = torch.empty(splits * m * n, dtype=torch.float32, device=a.device) if splits > 1 else None
    _launch(timer_class, lambda: lib().ver_gemm_nn_splitk(
        _p(a), ctypes.c_long(a.stride(0)), _p(w), ctypes.c_long(w.stride(0)), _p(bias) if bias is not None else None,
        _p(out), ctypes.c_long(out.stride(0)), ctypes.c_long(m), k, n, int(splits), _p(ws) if ws is not None else None,
        ctypes.c_long(ws.numel() * 4 if ws is not None else 0), _stream()), meta=dict(flops=2.0 * m * k * n))
    return out


def gemm_nn_taps_supported(lattice, layout, w, c):
    """What ``gemm_nn_taps`` takes: a contiguous bf16 lattice below 2 GiB in layout 0 / 2 / 3 with C % 32 == 0, a bf16 weight
    matrix with 16-byte aligned rows."""
    return (lattice.is_cuda and lattice.dtype == torch.bfloat16 and lattice.is_contiguous() and layout in (PLAIN, ZSPLIT, PLANAR_ZSPLIT)
            and lattice.numel() * 2 < 2 ** 31 - 1 and c % 32 == 0 and c >= 64 and w.is_cuda and w.dtype == torch.bfloat16
            and w.dim() == 2 and w.stride(1) == 1 and w.stride(0) % 8 == 0 and w.data_ptr() % 16 == 0)


def gemm_nn_taps(lattice, layout, combined_hw, taps, w, rowpos=None, bias=None, out=None, const_rows=None, planes=None,
                 timer_class='ver_gemm_nn'):
    """ver_gemm_nn_segments: ``tap_matrix(lattice) @ w (+ rowpos by row position) (+ bias)`` without the tap matrix: lattice
    bf16 in layout 0 / 2 / 3 (see ``lattice_gather``), rows = the cells (b, zl, y, x) of the combined (H, W) lattice.  ``taps``:
    the segments of the K axis in order -- (dz in {0, 2}, dy, dx) = the C channels of that neighbouring cell, or ('c', block) =
    the columns of constant-pattern block ``block`` of ``const_rows`` bf16 [2 H W, blocks, width] (what ``lattice_gather``
    copies into every viewpoint's rows).  w bf16 [sum of the segment widths, N] -> bf16 [B * 2 * H * W, N].
    ``planes`` = list with the source plane of every tap: ``lattice`` is then [planes, ...] -- several lattices of one shape
    (the four class planes of a layer's output gradient), tap t reads ``lattice[planes[t]]``."""
    lat, w = _gpu(lattice, 'lattice'), _gpu(w, 'w')
    H, W = combined_hw
    plane_elems = nplanes = 0
    if planes is not None:
        nplanes, plane_elems = int(lat.shape[0]), int(lat[0].numel())
        if len(planes) != len(taps) or not lat.is_contiguous():
            raise ValueError('gemm_nn_taps: one source plane per tap, contiguous planes')
        B, _, C = _lattice_dims(lat[0], int(layout))
        lat_one = lat[0]
    else:
        B, _, C = _lattice_dims(lat, int(layout))
        lat_one = lat
    if not gemm_nn_taps_supported(lat_one, int(layout), w, C):
        raise RuntimeError('gemm_nn_taps: unsupported operands %s layout %d / %s %s' % (tuple(lat.shape), layout, tuple(w.shape), w.stride()))
    m, n = B * 2 * H * W, w.shape[1]
    ncst = cw = 0
    if const_rows is not None:
        const_rows = _gpu(const_rows, 'const_rows')
        if not (const_rows.is_contiguous() and const_rows.dtype == torch.bfloat16 and const_rows.dim() == 3 and const_rows.shape[0] == 2 * H * W):
            raise ValueError('gemm_nn_taps: const_rows must be a contiguous bf16 [2 H W, blocks, width] table')
        ncst, cw = int(const_rows.shape[1]), int(const_rows.shape[2])
    taps = [((-1 - int(t[1]), 0, 0) if t[0] == 'c' else t) for t in taps]
    kdim = sum(cw if t[0] < 0 else C for t in taps)
    if w.shape[0] != kdim:
        raise ValueError('gemm_nn_taps: w has %d rows, the segments span %d columns' % (w.shape[0], kdim))
    if out is None:
        out = torch.empty(m, n, dtype=torch.bfloat16, device=lat.device)
    if out.shape != (m, n) or out.dtype != torch.bfloat16 or out.stride(1) != 1 or not out.is_cuda:
        raise RuntimeError('gemm_nn_taps: out must be a bf16 [M, N] GPU matrix with unit column stride')
    if rowpos is not None:
        rowpos = _gpu(rowpos, 'rowpos').float().contiguous()
        if tuple(rowpos.shape) != (2 * H * W, n):
            raise ValueError('gemm_nn_taps: rowpos must be [2 H W, N]')
    if bias is not None:
        bias = _gpu(bias, 'bias').float().contiguous()
    flat = [int(v) for t in taps for v in t]
    arr = (ctypes.c_int * len(flat))(*flat)
    parr = (ctypes.c_int * len(taps))(*[int(v) for v in planes]) if planes is not None else None
    _launch(timer_class, lambda: lib().ver_gemm_nn_planes(
        _p(lat), int(layout), int(B), int(H), int(W), int(C), ctypes.c_long(plane_elems), nplanes, parr, arr, len(taps),
        _p(const_rows) if const_rows is not None else None,
        ncst, cw, _p(w), ctypes.c_long(w.stride(0)), _p(rowpos) if rowpos is not None else None, _p(bias) if bias is not None else None, _p(out), ctypes.c_long(out.stride(0)),
        int(n), _stream()), meta=dict(flops=2.0 * m * w.shape[0] * n))
    return out


def _segment_args(lat, layout, combined_hw, taps, const_rows, name):
    H, W = combined_hw
    B, _, C = _lattice_dims(lat, int(layout))
    ncst = cw = 0
    if const_rows is not None:
        const_rows = _gpu(const_rows, 'const_rows')
        if not (const_rows.is_contiguous() and const_rows.dtype == torch.bfloat16 and const_rows.dim() == 3 and const_rows.shape[0] == 2 * H * W):
            raise ValueError('%s: const_rows must be a contiguous bf16 [2 H W, blocks, width] table' % name)
        ncst, cw = int(const_rows.shape[1]), int(const_rows.shape[2])
    taps = [((-1 - int(t[1]), 0, 0) if t[0] == 'c' else t) for t in taps]
    kdim = sum(cw if t[0] < 0 else C for t in taps)
    flat = [int(v) for t in taps for v in t]
    return B, H, W, C, const_rows, ncst, cw, kdim, (ctypes.c_int * len(flat))(*flat), len(taps)


def wgrad_tn_segments_supported(lattice, layout, c, const_width, g):
    """What ``wgrad_tn_segments`` takes (segment widths multiples of 64, the lattice below 2 GiB, 2 H W < 65 536 checked by
    the library)."""
    return (lattice.is_cuda and lattice.dtype == torch.bfloat16 and lattice.is_contiguous() and layout in (PLAIN, ZSPLIT, PLANAR_ZSPLIT)
            and lattice.numel() * 2 < 2 ** 31 - 1 and c % 64 == 0 and const_width % 64 == 0 and g.is_cuda and g.dtype == torch.bfloat16
            and g.dim() == 2 and g.stride(1) == 1 and g.stride(0) % 8 == 0 and g.data_ptr() % 16 == 0 and g.shape[1] % 4 == 0)


def wgrad_tn_segments(lattice, layout, combined_hw, taps, g, out_dtype=None, out=None, const_rows=None, splits=0):
    """ver_wgrad_tn_segments: ``tap_matrix(lattice).t() @ g`` without the tap matrix (operands as ``gemm_nn_taps``): the weight
    gradient of a lattice layer's class GEMM, [sum of the segment widths, N] in ``out_dtype`` (``out``: a matrix to write into)."""
    lat, g = _gpu(lattice, 'lattice'), _gpu(g, 'g')
    B, H, W, C, const_rows, ncst, cw, ka, arr, nseg = _segment_args(lat, layout, combined_hw, taps, const_rows, 'wgrad_tn_segments')
    if not wgrad_tn_segments_supported(lat, int(layout), C, cw, g):
        raise RuntimeError('wgrad_tn_segments: unsupported operands %s layout %d / %s %s' % (tuple(lat.shape), layout, tuple(g.shape), g.stride()))
    if g.shape[0] != B * 2 * H * W:
        raise ValueError('wgrad_tn_segments: g has %d rows, %d cells' % (g.shape[0], B * 2 * H * W))
    n = g.shape[1]
    out_dtype = out_dtype or (out.dtype if out is not None else lat.dtype)
    if out is None:
        out = torch.empty(ka, n, dtype=out_dtype, device=lat.device)
    elif not (out.is_cuda and out.shape == (ka, n) and out.dtype == out_dtype and out.stride(1) == 1):
        raise RuntimeError('wgrad_tn_segments: out must be a [Ka, N] GPU matrix of out_dtype with unit column stride')
    L = lib()
    if splits <= 0:
        splits = L.ver_wgrad_tn_segments_splits(int(B), int(H), int(W), ctypes.c_long(ka), int(n), ctypes.c_long(g.stride(0)))
    ws = torch.empty(splits * ka * n, dtype=torch.float32, device=lat.device)
    _launch('ver_wgrad_tn', lambda: L.ver_wgrad_tn_segments(
        _p(lat), int(layout), int(B), int(H), int(W), int(C), arr, nseg, _p(const_rows) if const_rows is not None else None, ncst, cw,
        _p(g), ctypes.c_long(g.stride(0)), int(n), _p(out), ctypes.c_long(out.stride(0)), 1 if out_dtype == torch.bfloat16 else 0,
        int(splits), _p(ws), ctypes.c_long(ws.numel() * 4), _stream()), meta=dict(flops=2.0 * g.shape[0] * ka * n))
    return out
