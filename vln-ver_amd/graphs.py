"""hipGraph replay of the head (``GraphedHead``) and of the whole lifting step (``GraphedLiftStep``) for small, launch-bound
batches.

At the reference's own batch point (vocc.py: ``samples_per_gpu=1``) the full multi-task step is bound by the HOST: ~1 500
module calls forward and as many autograd nodes backward, 13-16 ms + 23-28 ms of Python / dispatcher time around ~20 ms of
GPU work (scratch/r03/full_step_host_profile.py).  ``GraphedHead`` records the head's forward and its backward once, for
one fixed batch shape, as two HIP graphs (``torch.cuda.make_graphed_callables``) and replays them: the Hungarian assignment,
the loss terms, gradient clipping and the optimizer stay eager between the two.  Everything the graphs contain is what the
eager path launches -- same kernels, same arithmetic, fresh dropout seeds per replay (they come from the device generator).

Nothing per step may change shape: one ``GraphedHead`` per (batch size, dtype); the cameras and features are copied into
the graphs' static inputs on every call.  Build it BEFORE wrapping the head in DistributedDataParallel (PyTorch's rule
for graphed callables).  ``bench.py`` uses it for the ``config.latency`` records of the full multi-task workload on one
rank (``graphed: true`` in the record); the headline step is always eager.

One rule found the hard way (ROCm 7.2 / PyTorch 2.10): NO loss of an earlier eager step may still be referenced when the
graphs are captured.  A live scalar with a spent autograd graph behind it (``last = step()`` kept around for logging) makes
``hipStreamEndCapture`` of the backward graph crash the process -- a segmentation fault, not an exception
(scratch/r03/graphed_full_step.py reproduces it with LIKE_BENCH=3, and not with DEL_LAST=1).  ``GraphedHead`` therefore
looks for such tensors first and refuses with an error that says what to drop.
"""
import gc

import torch


def _live_losses():
    """0-dim tensors that still carry an autograd graph: what a training loop keeps of its previous steps."""
    found = []
    for obj in gc.get_objects():
        try:
            if isinstance(obj, torch.Tensor) and obj.dim() == 0 and obj.grad_fn is not None:
                found.append(obj)
        except Exception:                                  # objects that do not like being inspected
            continue
    return found


class _HeadForward(torch.nn.Module):
    """Tensor-in / tensor-out view of ``VoxelFormerOccupancyHead.forward`` (graphed callables take and return tensors)."""

    def __init__(self, head, autocast_dtype, occupancy_rows):
        super().__init__()
        self.head = head
        self.autocast_dtype = autocast_dtype
        self.occupancy_rows = occupancy_rows
        self.row_plan = None

    def forward(self, feats, world2pixel, origin):
        with torch.autocast('cuda', dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None,
                            cache_enabled=False):
            outs = self.head(feats, None, world2pixel=world2pixel, origin=origin, occupancy_rows=self.occupancy_rows)
        occ = outs['occupancy_preds']
        if isinstance(occ, tuple):                       # (logits in GEMM row order, plan, bs): the plan is host data
            occ, plan, bs = occ
            self.row_plan = (plan, bs)
        return outs['all_cls_scores'].float(), outs['all_bbox_preds'].float(), occ, outs['bev_embed']


class GraphedHead:
    """``outs = GraphedHead(head, feats, w2p, org)(feats, w2p, org)``: the dict ``head.loss`` takes, produced by graph replay.

    feats [Ncam, B, Nk, C], world2pixel [B, Ncam, 4, 4], origin [B, 3] on the GPU are the SAMPLE inputs that fix the
    shapes; ``autocast_dtype=torch.bfloat16`` (default) runs the head under bf16 autocast as ``bench.py`` does, None in
    fp32.  The head must be in the mode (train / eval) and have the ``requires_grad`` flags it will be used with (checked
    on every call).  Outputs are valid until the next call (they alias the graphs' static buffers)."""

    def __init__(self, head, feats, world2pixel, origin, autocast_dtype=torch.bfloat16, occupancy_rows=True,
                 check_live_losses=True):
        if not feats.is_cuda:
            raise RuntimeError('GraphedHead needs GPU tensors (HIP graphs)')
        if head.only_occ or head.only_det or head.add_layout:
            raise NotImplementedError('GraphedHead covers the default multi-task branch of the head')
        if getattr(head, 'getbev', None) is not None:
            raise NotImplementedError('GraphedHead: a head built with getbev=<store> writes volumes to the host inside '
                                      'forward (device -> host copies and file I/O), which cannot be captured')
        self.head, self.training = head, head.training
        self.grad_flags = [p.requires_grad for p in head.parameters()]
        if check_live_losses:
            gc.collect()
            n = len(_live_losses())
            if n:
                raise RuntimeError(
                    'GraphedHead: %d scalar tensor(s) with an autograd graph are still alive (losses of earlier eager '
                    'steps?).  Drop them (`loss = None`, or keep `loss.detach()` / `float(loss)`) before building the '
                    'graphs: with one alive, ending the capture of the backward graph crashes the process on this '
                    'ROCm / PyTorch build (vln-ver_amd/graphs.py).' % n)
        self.module = _HeadForward(head, autocast_dtype, occupancy_rows)
        self.graphed = torch.cuda.make_graphed_callables(self.module, (feats, world2pixel, origin), allow_unused_input=True)

    def __call__(self, feats, world2pixel, origin):
        """NOTE: the returned tensors ALIAS the graph's static output buffers -- ``bev_embed``, the class scores and the
        boxes are overwritten by the next call; clone what has to survive it."""
        if self.head.training != self.training or [p.requires_grad for p in self.head.parameters()] != self.grad_flags:
            raise RuntimeError('GraphedHead: head.training / requires_grad flags differ from capture time; build a new '
                               'GraphedHead for the new mode')
        cls, box, occ, bev = self.graphed(feats, world2pixel, origin)
        if self.module.row_plan is not None:
            occ = (occ,) + self.module.row_plan
        return dict(bev_embed=bev, all_cls_scores=cls, all_bbox_preds=box, all_layout_preds=None, occupancy_preds=occ,
                    flow_preds=None, enc_cls_scores=None, enc_bbox_preds=None, enc_occupancy_preds=None)


class GraphedLiftStep:
    """One training step of the lifting path -- forward, occupancy loss, backward, gradient clipping + AdamW -- for ONE fixed
    batch shape, captured as a single hipGraph and replayed: the reference's own operating point (vocc.py:222
    ``samples_per_gpu=1``) is bound by the host in the eager step (~450 launches behind ~9 ms of Python / dispatcher time for
    ~4 ms of GPU work).  Same kernels, same arithmetic as the eager step; dropout seeds come from the device generator and
    advance per replay; ``optim.ClipAdamW`` keeps its pointer table, hyper-parameters and per-tensor update counts on the
    device, so the captured update uses the right bias corrections on every replay (``optimizer.replayed()`` keeps the host's
    counts in step, ``optimizer.refresh()`` uploads a learning rate a scheduler has moved).

    ``model(feats, w2p, org, gt) -> scalar loss`` (bench.py's LiftTrainer); ``optimizer``: a ClipAdamW over the model's trainable
    parameters.  Construction runs ``warmup`` REAL eager steps on the sample inputs (allocator, AdamW state, library
    heuristics) before the capture -- they train the model like any other step.  One rank only: the gradient all-reduce of
    DistributedDataParallel is not part of the graph.  ``loss = step(feats, w2p, org, gt)`` copies the inputs into the graph's
    static buffers (unless they ARE those buffers: ``step.inputs``) and returns the graph's loss tensor, overwritten by the
    next call."""

    def __init__(self, model, optimizer, feats, world2pixel, origin, gt, warmup=3, check_live_losses=True):
        if not feats.is_cuda:
            raise RuntimeError('GraphedLiftStep needs GPU tensors (HIP graphs)')
        if not hasattr(optimizer, 'prepare_capture'):
            raise TypeError('GraphedLiftStep needs an optimizer whose step() can be captured (optim.ClipAdamW)')
        self.model, self.optimizer = model, optimizer
        self.inputs = tuple(t.detach().clone() for t in (feats, world2pixel, origin, gt))
        self.training = model.training
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            norm = None
            for _ in range(max(1, warmup)):                 # (at least one: the optimizer state must exist before the capture)
                optimizer.zero_grad(set_to_none=True)
                loss = model(*self.inputs)
                loss.backward()
                norm = optimizer.step()
            self._clean_warmup = norm is not None and bool(torch.isfinite(norm))
            self._unchecked = 3
            loss = None
            optimizer.zero_grad(set_to_none=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if check_live_losses:
            gc.collect()
            n = len(_live_losses())
            if n:
                raise RuntimeError('GraphedLiftStep: %d scalar tensor(s) with an autograd graph are still alive (losses of '
                                   'earlier eager steps?); drop them before building the graph (see GraphedHead)' % n)
        optimizer.prepare_capture()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            loss = model(*self.inputs)
            loss.backward()
            self.grad_norm = optimizer.step()
            self.loss = loss.detach()
        loss = None

    def __call__(self, feats, world2pixel, origin, gt):
        if self.model.training != self.training:
            raise RuntimeError('GraphedLiftStep: model.training differs from capture time')
        for dst, src in zip(self.inputs, (feats, world2pixel, origin, gt)):
            if src is not dst:
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise RuntimeError('GraphedLiftStep: input %s %s, captured with %s %s' % (tuple(src.shape), src.dtype, tuple(dst.shape), dst.dtype))
                dst.copy_(src, non_blocking=True)
        self.optimizer.refresh()
        self.graph.replay()
        self.optimizer.replayed()
        if self._unchecked:
            # the first replays are checked (one synchronisation each): a replay that leaves a non-finite gradient norm behind
            # after a clean capture step is the runtime's memset-node problem (vln-ver_amd/__init__.py), not the model's
            self._unchecked -= 1
            if not bool(torch.isfinite(self.grad_norm)) and self._clean_warmup:
                raise RuntimeError(
                    'GraphedLiftStep: the gradient norm of a replayed step is not finite although the eager warm-up steps were '
                    'clean.  On ROCm 7.2 hipGraph replay needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment BEFORE the '
                    'first GPU call of the process (memset nodes are not ordered against the kernels behind them otherwise); '
                    'importing vln-ver_amd sets it, but only takes effect if HIP was not initialised earlier.')
        return self.loss
