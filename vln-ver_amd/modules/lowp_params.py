"""bf16 working copies of many small Linear parameters, made -- and their gradients widened -- in ONE pass each.

Under bf16 autocast every ``nn.Linear`` call casts its fp32 weight and bias on the way in (one launch each) and autograd
widens each bf16 gradient on the way out (another launch each).  For the detection half of the vocc.py head (6 decoder
layers + 6 cls / 6 reg branches: ~170 parameter tensors of 768 x 768 and smaller, 100 queries per viewpoint) that is ~340
of the step's launches, each moving a few KB.  ``LowpParams`` casts the whole set with one multi-tensor copy into one flat
bf16 buffer, lends the slices to the modules for the duration of a forward, and widens all the gradients with one
multi-tensor copy when the last of them has arrived.  The values are exactly autocast's (round-to-nearest bf16 of the fp32
master; the bf16 gradient of the GEMM widened to fp32), the master parameters and their ``.grad`` stay fp32.
"""
import contextlib

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
        try:
            for (m, n), c in zip(self.slots, copies):
                m._parameters[n] = c
            yield
        finally:
            for (m, n), p in zip(self.slots, masters):
                m._parameters[n] = p
