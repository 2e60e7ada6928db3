"""bf16 working copies of many small Linear parameters, made -- and their gradients widened -- in ONE pass each.

Under bf16 autocast every ``nn.Linear`` call casts its fp32 weight and bias on the way in (one launch each) and autograd
widens each bf16 gradient on the way out (another launch each).  For the detection half of the vocc.py head (6 decoder
layers + 6 cls / 6 reg branches: ~170 parameter tensors of 768 x 768 and smaller, 100 queries per viewpoint) that is ~340
of the step's launches, each moving a few KB.  ``LowpParams`` casts the whole set with one multi-tensor copy into one flat
bf16 buffer, lends the slices to the modules for the duration of a forward, and widens all the gradients with one
multi-tensor copy when the last of them has arrived.  The values are exactly autocast's (round-to-nearest bf16 of the fp32
master; the bf16 gradient of the GEMM widened to fp32), the master parameters and their ``.grad`` stay fp32.
While lent, a plain ``nn.Linear`` also runs through ``_LowpLinear``, whose weight gradient is the hand-written
rows-on-the-contraction-axis kernel (``ver_wgrad_tn``) instead of the library's 144-tile product.
"""
import contextlib
import functools

import torch
import torch.nn as nn

_ALIGN = 128            # elements: every slice of the flat buffer starts on a 256-byte boundary (GEMM operand alignment)


class _CastAll(torch.autograd.Function):
    """fp32 tensors -> bf16 slices of one flat buffer; backward: every bf16 gradient -> fp32, one multi-tensor copy."""

    @staticmethod
    def forward(ctx, offsets, total, *params):
        flat = torch.empty(total, dtype=torch.bfloat16, device=params[0].device)
        outs = [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offsets, params)]
        torch._foreach_copy_(outs, [p.detach() for p in params])
        ctx.offsets, ctx.total = offsets, total
        ctx.set_materialize_grads(False)          # a parameter the step does not use keeps grad None, as without the copies
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        live = [i for i, g in enumerate(grads) if g is not None]
        out = [None] * len(grads)
        if live:
            flat = torch.empty(ctx.total, dtype=torch.float32, device=grads[live[0]].device)
            wide = [flat[ctx.offsets[i]:ctx.offsets[i] + grads[i].numel()].view(grads[i].shape) for i in live]
            torch._foreach_copy_(wide, [grads[i] for i in live])
            for i, w in zip(live, wide):
                out[i] = w
        return (None, None) + tuple(out)


class _LowpLinear(torch.autograd.Function):
    """``addmm(b, x, w^T)`` on bf16 operands; backward: d(input) by the library, d(weight) = g^T x with the ROWS on the
    contraction axis by ``ver_wgrad_tn`` (split over the rows, fp32 sums: 28 us against the library's 41 for the
    6 400 x 768 x 768 products of a 64-viewpoint step, 70 against 170 for value_proj's 57 600 rows), bias gradient a column
    sum.  All gradients bf16, like the operands."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return x @ w.t() if b is None else torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            from ..hipops import wgrad_tn, wgrad_tn_supported
            if g.shape[0] >= 512 and min(g.shape[1], x.shape[1]) >= 32 and wgrad_tn_supported(g, x):
                gw = wgrad_tn(g, x, out_dtype=torch.bfloat16)
            else:
                gw = g.t() @ x
        gb = g.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


def _lent_linear_forward(mod, x):
    """``nn.Linear.forward`` while the module holds bf16 copies of its parameters (``LowpParams.lent``)."""
    w, b = mod.weight, mod.bias
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != torch.bfloat16:
        x2 = x2.to(torch.bfloat16)                       # (the cast autocast's linear makes)
    with torch.autocast('cuda', enabled=False):
        y = _LowpLinear.apply(x2, w, b)
    return y.view(*x.shape[:-1], y.shape[-1])


class LowpParams:
    """The fp32 ``weight`` / ``bias`` (and ``in_proj_*``) tensors of every ``nn.Linear`` / ``nn.MultiheadAttention`` under
    ``roots``.  ``with lowp.lent():`` replaces them by their bf16 copies inside the owning modules (LayerNorm and embedding
    parameters are left alone: autocast keeps those in fp32) and puts the fp32 parameters back on the way out."""

    def __init__(self, roots):
        self.slots = []                                  # (module, name)
        seen = set()
        for root in roots:
            for m in root.modules():
                if isinstance(m, nn.Linear):
                    names = ('weight', 'bias')
                elif isinstance(m, nn.MultiheadAttention):
                    names = ('in_proj_weight', 'in_proj_bias')
                else:
                    continue
                for n in names:
                    p = m._parameters.get(n)
                    if p is not None and id(p) not in seen and p.dtype == torch.float32:
                        seen.add(id(p))
                        self.slots.append((m, n))
        self.offsets, off = [], 0
        for m, n in self.slots:
            self.offsets.append(off)
            off += -(-m._parameters[n].numel() // _ALIGN) * _ALIGN
        self.total = off

    def applies(self, like):
        return (bool(self.slots) and like.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled('cuda')
                and torch.get_autocast_dtype('cuda') == torch.bfloat16)

    @contextlib.contextmanager
    def lent(self):
        masters = [m._parameters[n] for m, n in self.slots]
        with torch.autocast('cuda', enabled=False):
            copies = _CastAll.apply(tuple(self.offsets), self.total, *masters)
        linears = {id(m): m for m, _ in self.slots if type(m) is nn.Linear}.values()
        try:
            for (m, n), c in zip(self.slots, copies):
                m._parameters[n] = c
            for m in linears:
                m.forward = functools.partial(_lent_linear_forward, m)
            yield
        finally:
            for (m, n), p in zip(self.slots, masters):
                m._parameters[n] = p
            for m in linears:
                m.__dict__.pop('forward', None)
