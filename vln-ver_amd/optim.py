"""The optimizer step the reference configures for vocc.py -- gradient clipping by the global L2 norm in front of AdamW
(projects/configs/verformer/vocc.py:268-274: ``optimizer=dict(type='AdamW', ...)``, ``optimizer_config=dict(grad_clip=
dict(max_norm=..., norm_type=2))``, run by mmcv's OptimizerHook as ``clip_grad_norm_`` + ``optimizer.step()``) -- as ONE
C-ABI call over all parameters: ``ver_clip_adamw_step`` (csrc/ver_optim.hip; two launches, 32 bytes per parameter).

``ClipAdamW`` keeps torch.optim.AdamW's state layout (``exp_avg`` / ``exp_avg_sq`` / ``step`` per parameter) and arithmetic
(decoupled weight decay, bias corrections, amsgrad off); parameters and gradients are dense fp32 GPU tensors of ONE parameter
group (the clip norm is global).  There is no CPU path here: CPU tensors raise, as every product path of this package does
without its HIP library.
"""
import ctypes

import torch


class ClipAdamW(torch.optim.Optimizer):
    CHUNK = 32768            # elements per workgroup

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError('ClipAdamW: hyper-parameters out of range')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, max_norm=max_norm))
        if len(self.param_groups) != 1:
            raise NotImplementedError('ClipAdamW: one parameter group (the clip norm is taken over all parameters)')
        self._tables = {}

    def _chunk_tables(self, sizes, device):
        key = (tuple(sizes), str(device))
        hit = self._tables.get(key)
        if hit is None:
            tensor, index = [], []
            for t, n in enumerate(sizes):
                c = -(-n // self.CHUNK)
                tensor += [t] * c
                index += list(range(c))
            hit = self._tables[key] = (
                torch.tensor(list(sizes), dtype=torch.int64, device=device), torch.tensor(tensor, dtype=torch.int32, device=device),
                torch.tensor(index, dtype=torch.int32, device=device), torch.empty(len(tensor), dtype=torch.float32, device=device))
            if len(self._tables) > 8:
                self._tables.pop(next(iter(self._tables)))
        return hit

    @torch.no_grad()
    def step(self, closure=None):
        """One clipped AdamW update of every parameter that has a gradient.  Returns the gradient norm before clipping
        (a device scalar, like ``clip_grad_norm_``)."""
        if closure is not None:
            raise NotImplementedError('ClipAdamW: no closure')
        from . import hipops
        group = self.param_groups[0]
        ps, gs = [], []
        for p in group['params']:
            g = p.grad
            if g is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse and p.is_contiguous()):
                raise TypeError('ClipAdamW: dense fp32 GPU parameters and gradients only (got %s %s on %s)' % (p.dtype, g.dtype, p.device))
            ps.append(p)
            gs.append(g if g.is_contiguous() else g.contiguous())
        # parameter / moment pointers and the step count only change when the SET of parameters with a gradient does
        key = tuple(id(p) for p in ps)
        if key != getattr(self, '_static_key', None):
            steps = set()
            for p in ps:
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                steps.add(st['step'])
            if len(steps) > 1:
                raise RuntimeError('ClipAdamW: parameters at different step counts %s: one bias correction per call' % sorted(steps))
            self._static_key = key
            self._static_ptrs = ([p.data_ptr() for p in ps], [self.state[p]['exp_avg'].data_ptr() for p in ps],
                                 [self.state[p]['exp_avg_sq'].data_ptr() for p in ps])
            self._static_states = [self.state[p] for p in ps]
            self._static_sizes = [p.numel() for p in ps]
        step = (self._static_states[0]['step'] + 1) if ps else 0
        for st in self._static_states:
            st['step'] = step
        dev = ps[0].device if ps else None
        norm = torch.zeros((), dtype=torch.float32, device=dev) if ps else torch.zeros(())
        if not ps:
            return norm
        n = len(ps)
        sizes, chunk_tensor, chunk_index, partial = self._chunk_tables(self._static_sizes, dev)
        pp, mp, vp = self._static_ptrs
        table = torch.tensor(pp + [g.data_ptr() for g in gs] + mp + vp, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
        b1, b2 = group['betas']
        L = hipops.lib()
        hipops._launch('ver_clip_adamw_step', lambda: L.ver_clip_adamw_step(
            hipops._p(table), hipops._p(sizes), hipops._p(chunk_tensor), hipops._p(chunk_index), n, int(chunk_tensor.numel()),
            self.CHUNK, hipops._p(partial), hipops._p(norm), ctypes.c_float(float(group['max_norm'] or 0.0)),
            ctypes.c_float(group['lr']), ctypes.c_float(b1), ctypes.c_float(b2), ctypes.c_float(group['eps']),
            ctypes.c_float(group['weight_decay']), ctypes.c_long(int(step)), hipops._stream()))
        # (table and contiguous copies may go out of scope: the caching allocator reuses their blocks only behind this
        #  launch on the same stream)
        return norm
