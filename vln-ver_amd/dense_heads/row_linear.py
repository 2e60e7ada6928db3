"""Linear layers over millions of rows (``occ_branches``, head:241-248: 504 000 voxels x bs rows
of width 128).

Forward and d(input) are ordinary GEMMs.  d(weight) = G^T X is a [out x rows] x [rows x in]
product with out, in <= 128 and rows = 4e6..16e6: a GEMM library sees ONE or two output tiles
and runs it on one or two CUs (measured 4.7 ms per layer for 8 viewpoints = 14 % of the step).
``row_linear`` splits the row dimension into chunks, runs them as one batched GEMM (thousands of
workgroups) and adds the fp32 partials; the bias gradient is a column sum.  Used on CUDA tensors only; CPU
tensors take ``F.linear``.
"""
import torch
import torch.nn.functional as F

_CHUNK = 8000            # 504000 = 63 * 8000


class _RowLinear(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias):
        w = weight.to(x.dtype)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.weight_dtype = weight.dtype
        if bias is None:
            return x @ w.t()
        return torch.addmm(bias.to(x.dtype), x, w.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        n, o = g.shape
        gx = g @ w if ctx.needs_input_grad[0] else None
        gb = None
        if ctx.has_bias:
            # (a column sum, not a batched product with a stride-0 row of ones: a GEMM with M = 1 buys nothing over the
            #  reduction kernel, and stride-0 batch operands fault inside some hipBLASLt solutions when TunableOp tries them)
            gb = g.sum(0, dtype=torch.float32)
        if g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and n >= 4096 and min(o, x.shape[1]) >= 192:
            # rows on the contraction axis: the hand-written kernel of the head's weight gradients (csrc/ver_wgrad.hip;
            # fp32 sums over all rows, row chunks chosen by its cost model); the 128-wide Linears of occ_branches would fill a
            # quarter of its 256 x 256 tile and keep the batched form below
            from ..hipops import wgrad_tn, wgrad_tn_supported
            if wgrad_tn_supported(g, x):
                # (fp32 for an fp32 master weight; a bf16 working copy -- modules/lowp_params.py -- gets the bf16 gradient
                #  autograd expects for it, widened with all the others in one pass)
                wide = torch.float32 if ctx.weight_dtype == torch.float32 else torch.bfloat16
                if wide == torch.bfloat16 and gb is not None:
                    gb = gb.to(torch.bfloat16)
                return gx, wgrad_tn(g, x, out_dtype=wide), gb
        s = n // _CHUNK
        main = s * _CHUNK
        gw = x.new_zeros((o, x.shape[1]), dtype=torch.float32)
        if s:
            g3 = g[:main].view(s, _CHUNK, o)
            gw += torch.bmm(g3.transpose(1, 2), x[:main].view(s, _CHUNK, -1)).sum(0, dtype=torch.float32)
        if main < n:
            gw += (g[main:].t() @ x[main:]).float()
        return gx, gw, gb


def row_linear(x, weight, bias):
    """``F.linear`` for x [..., in] with a split-K weight gradient (see module docstring)."""
    if not x.is_cuda:
        return F.linear(x, weight, bias)
    if torch.is_autocast_enabled('cuda'):
        x = x.to(torch.get_autocast_dtype('cuda'))
    shape = x.shape
    with torch.autocast('cuda', enabled=False):
        y = _RowLinear.apply(x.reshape(-1, shape[-1]), weight, bias)
    return y.view(*shape[:-1], -1)
