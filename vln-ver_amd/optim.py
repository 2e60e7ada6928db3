"""The optimizer step the reference configures for vocc.py -- gradient clipping by the global L2 norm in front of AdamW
(projects/configs/verformer/vocc.py:260-274: ``optimizer=dict(type='AdamW', ..., paramwise_cfg=...)``,
``optimizer_config=dict(grad_clip=dict(max_norm=..., norm_type=2))``, run by mmcv's OptimizerHook as ``clip_grad_norm_`` +
``optimizer.step()``) -- as ONE C-ABI call over all parameters: ``ver_clip_adamw_step_tensors`` (csrc/ver_optim.hip; two
launches, 32 bytes per parameter).

``ClipAdamW`` keeps torch.optim.AdamW's state layout (``exp_avg`` / ``exp_avg_sq`` / ``step`` per parameter) and arithmetic
(decoupled weight decay, bias corrections from the PARAMETER's own step count, amsgrad off).  Any number of parameter groups
(mmcv's ``paramwise_cfg`` builds one per parameter: vocc.py gives ``img_backbone`` lr_mult 0.1) with their own lr / betas /
eps / weight decay; the clip norm is global over every parameter that has a gradient, as in ``clip_grad_norm_``, and a
non-finite norm poisons every gradient the way ``clip_grad_norm_`` does.  Parameters and gradients are dense fp32 GPU tensors.
There is no CPU path here: CPU tensors raise, as every product path of this package does without its HIP library.

What a launch reads lives on the device -- a pointer table (parameter, gradient, both moments per tensor), the per-tensor
hyper-parameters and the per-tensor update counts, which the kernel advances itself -- so a hipGraph that contains
``step()`` replays correctly (``vln-ver_amd/graphs.py``).  The host keeps what those buffers SHOULD hold and compares on
every call: a changed gradient / moment / parameter pointer (``zero_grad(set_to_none=True)``, ``load_state_dict``,
``model.to()``), a changed lr (schedulers) or a step count that was set from outside is uploaded again before the launch --
nothing is cached by object identity.
"""
import ctypes

import numpy as np
import torch


class ClipAdamW(torch.optim.Optimizer):
    CHUNK = 32768            # elements per workgroup

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError('ClipAdamW: hyper-parameters out of range')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, max_norm=max_norm))
        self._tables = {}
        self._dev = None             # what the device buffers hold: dict(ptrs, hyper, steps, n) + the buffers themselves
        self._keep = []              # pinned staging buffers a captured graph copies from on every replay

    # ------------------------------------------------------------------ host-side bookkeeping
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

    def _collect(self):
        """-> (parameters with a gradient, their gradients, hyper rows [n][6], max_norm): state created on first sight, ``step``
        normalised to an int (a torch AdamW checkpoint holds one tensor per parameter)."""
        ps, gs, hyper = [], [], []
        max_norm = None
        for group in self.param_groups:
            mn = float(group.get('max_norm') or 0.0)
            if max_norm is None:
                max_norm = mn
            elif mn != max_norm:
                raise ValueError('ClipAdamW: one max_norm for all parameter groups (the clip norm is global)')
            b1, b2 = group['betas']
            if not (group['lr'] >= 0 and 0 <= b1 < 1 and 0 <= b2 < 1 and group['eps'] >= 0 and group['weight_decay'] >= 0):
                raise ValueError('ClipAdamW: hyper-parameters out of range in a parameter group')
            row = (float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']), 0.0)
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse and p.is_contiguous()):
                    raise TypeError('ClipAdamW: dense fp32 GPU parameters and gradients only (got %s %s on %s)' % (p.dtype, g.dtype, p.device))
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif not isinstance(st['step'], int):
                    st['step'] = int(st['step'])
                ps.append(p)
                gs.append(g if g.is_contiguous() else g.contiguous())
                hyper.append(row)
        return ps, gs, hyper, max_norm or 0.0

    def _upload(self, ptrs, hyper, steps, dev, capturing):
        """Bring the device buffers to (ptrs, hyper, steps).  In place whenever the tensor count is unchanged: a captured
        graph keeps reading the same addresses.  While a graph is being captured only the pointer table may move (the copy
        becomes a graph node that re-copies the same values on every replay); hyper-parameters and update counts must
        already be resident -- they change between replays, from outside the graph."""
        d = self._dev
        n = len(steps)
        fresh = d is None or d['n'] != n or d['device'] != dev
        if fresh:
            if capturing:
                raise RuntimeError('ClipAdamW: the first step() for this set of parameters inside a graph capture; run '
                                   '`prepare_capture()` (or one eager step) before capturing')
            d = self._dev = dict(n=n, device=dev, ptrs=None, hyper=None, steps=None,
                                 table=torch.empty(4 * n, dtype=torch.int64, device=dev),
                                 hyper_dev=torch.empty(n, 6, dtype=torch.float32, device=dev),
                                 steps_dev=torch.empty(n, dtype=torch.int32, device=dev))

        def push(dst, array):
            stage = getattr(self, '_stage', None)
            if capturing and stage is not None and stage.numel() == array.size and array.dtype == np.int64:
                src = stage                               # (pinned before the capture: no host allocation inside it)
                src.numpy()[:] = array
                self._stage = None
            else:
                src = torch.from_numpy(array).pin_memory()
            if capturing:
                self._keep.append(src)
            dst.copy_(src.view(dst.shape), non_blocking=True)
        if d['ptrs'] != ptrs:
            push(d['table'], np.asarray(ptrs, dtype=np.int64))
            d['ptrs'] = ptrs
        if d['hyper'] != hyper or d['steps'] != steps:
            if capturing:
                raise RuntimeError('ClipAdamW: hyper-parameters / step counts changed inside a graph capture; call '
                                   '`prepare_capture()` right before capturing and `refresh()` between replays')
            if d['hyper'] != hyper:
                push(d['hyper_dev'], np.asarray(hyper, dtype=np.float32).reshape(n, 6))
                d['hyper'] = hyper
            if d['steps'] != steps:
                push(d['steps_dev'], np.asarray(steps, dtype=np.int32))
                d['steps'] = steps
        return d

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def step(self, closure=None):
        """One clipped AdamW update of every parameter that has a gradient.  Returns the gradient norm before clipping
        (a device scalar, like ``clip_grad_norm_``)."""
        if closure is not None:
            raise NotImplementedError('ClipAdamW: no closure')
        from . import hipops
        ps, gs, hyper, max_norm = self._collect()
        if not ps:
            return torch.zeros(())
        dev = ps[0].device
        capturing = torch.cuda.is_current_stream_capturing()
        n = len(ps)
        states = [self.state[p] for p in ps]
        ptrs = ([p.data_ptr() for p in ps] + [g.data_ptr() for g in gs] + [st['exp_avg'].data_ptr() for st in states]
                + [st['exp_avg_sq'].data_ptr() for st in states])
        steps = [st['step'] for st in states]
        d = self._upload(ptrs, hyper, steps, dev, capturing)
        norm = torch.zeros((), dtype=torch.float32, device=dev)
        sizes, chunk_tensor, chunk_index, partial = self._chunk_tables([p.numel() for p in ps], dev)
        L = hipops.lib()
        hipops._launch('ver_clip_adamw_step', lambda: L.ver_clip_adamw_step_tensors(
            hipops._p(d['table']), hipops._p(sizes), hipops._p(chunk_tensor), hipops._p(chunk_index), hipops._p(d['hyper_dev']),
            hipops._p(d['steps_dev']), n, int(chunk_tensor.numel()), self.CHUNK, hipops._p(partial), hipops._p(norm),
            ctypes.c_float(max_norm), hipops._stream()))
        if capturing:
            # nothing has run: the replays advance the device counts, ``replayed()`` the host's
            self._captured = states
        else:
            for st in states:
                st['step'] += 1
            d['steps'] = [s + 1 for s in steps]       # (the kernel advanced the device counts)
        # (contiguous gradient copies may go out of scope: the caching allocator reuses their blocks only behind this
        #  launch on the same stream)
        return norm

    # ------------------------------------------------------------------ hipGraph support
    @torch.no_grad()
    def prepare_capture(self):
        """Call right before capturing a graph that contains ``step()``, after at least one eager step over the same set of
        parameters (their state, hyper-parameters and update counts are then resident on the device): pins the staging
        buffer the captured ``step()`` fills with the pointer table of the capture's own gradient buffers."""
        n = sum(1 for group in self.param_groups for p in group['params'] if self.state.get(p))
        if not n or self._dev is None or self._dev['n'] != n:
            raise RuntimeError('ClipAdamW.prepare_capture(): run one eager step over the parameters of the step first')
        self._stage = torch.empty(4 * n, dtype=torch.int64).pin_memory()

    def replayed(self, times=1):
        """A graph holding ``step()`` was replayed ``times`` times: advance the host's step counts to what the device holds."""
        states = getattr(self, '_captured', None)
        if not states:
            raise RuntimeError('ClipAdamW.replayed(): no step() has been captured')
        for st in states:
            st['step'] += times
        if self._dev is not None and self._dev['steps'] is not None:
            self._dev['steps'] = [s + times for s in self._dev['steps']]

    @torch.no_grad()
    def refresh(self):
        """Between replays: upload hyper-parameters that changed (a scheduler moved lr) into the buffers the graph reads."""
        d = self._dev
        if d is None:
            return
        hyper = []
        states = getattr(self, '_captured', None) or []
        ids = {id(st) for st in states}
        for group in self.param_groups:
            b1, b2 = group['betas']
            row = (float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']), 0.0)
            hyper += [row for p in group['params'] if id(self.state.get(p)) in ids]
        if len(hyper) == d['n'] and hyper != d['hyper']:
            src = torch.from_numpy(np.asarray(hyper, dtype=np.float32).reshape(d['n'], 6)).pin_memory()
            d['hyper_dev'].copy_(src, non_blocking=True)
            d['hyper'] = hyper

    def load_state_dict(self, state_dict):
        """A torch.optim.AdamW checkpoint loads too (per-parameter step tensors become ints on the next step); its
        parameter groups carry no ``max_norm`` -- the groups REPLACE ours in torch's loader -- so keys a group lost get this
        optimizer's constructor values back."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            for k, v in self.defaults.items():
                group.setdefault(k, v)
        self._dev = None             # (pointers and counts are compared on every step anyway; drop the buffers with the old state)

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault('_tables', {})
        self.__dict__['_dev'] = None
        self.__dict__.setdefault('_keep', [])
