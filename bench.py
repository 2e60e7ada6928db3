#!/usr/bin/env python3
"""Benchmark of the VER 2D->3D lifting path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 from a plain shell: bench.py starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
process -- before anything has touched the GPU -- and relays its output and exit code (the reference's
tools/dist_train.sh:12-14 does the same with torch.distributed.launch).  Under a launcher (WORLD_SIZE in the
environment) it is one of the N ranks: one process per GPU, RCCL gradient all-reduce.

A "step" = one training pass of the hot path over one batch of synthetic viewpoints:
6x14x14x768 ViT features -> VERFormer encoder (projection, hit table, 3x fused multi-view
gather + GEMMs) -> coarse-to-fine occupancy head (even-lattice upsample, occ_proj, occ MLP)
-> sigmoid focal loss on [504000,16] -> backward -> gradient all-reduce (N>1) -> grad-clip ->
AdamW.  Workload = BASELINE.json configs[2] ("vocc.py coarse-to-fine multi-scale volume, bf16
fwd+bwd"), i.e. the config the metric "viewpoints/sec (multi-view->voxel fwd+bwd), vocc.py
config" is quoted on.  Inputs are resident in HBM before the timed region.  One JSON line on
rank 0 (contract in the task statement), with `roofline` (fused gather kernel, HIP-event timed
in the timed region) and `cpu_baseline` (the CPU oracle on this host's cores, N=1 only).
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time
import warnings

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=None,
                    help='viewpoints per GPU per step (default: 192 for vocc_c2f_train, 64 for the other workloads)')
    ap.add_argument('--micro', type=int, default=192,
                    help='viewpoints per head micro-batch (192 = the whole default batch at once: the upsample / occ_proj '
                         'GEMMs run at up to 1.6 PFLOP/s with 345 600 rows against 1.25 with 115 200; 158 GiB of the 288)')
    ap.add_argument('--no-tuned-gemms', action='store_true',
                    help='do not load the recorded hipBLASLt solution table (vln-ver_amd/tuning)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--workload', default='vocc_c2f_train', choices=['vocc_c2f_train', 'c2_single_scale_fwd', 'vocc_full_train'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=30.0,
                    help='budget of the cpu_baseline leg: 1 warm-up + up to 10 timed single-viewpoint passes (at least 5)')
    ap.add_argument('--backend', default='nccl', help='nccl (= RCCL); gloo only to exercise the N>1 path on one GPU')
    ap.add_argument('--config', default=None, help='mmcv-style config file (default: the vocc config shipped with the '
                                                   'package; the reference\'s projects/configs/verformer/vocc.py loads as is)')
    ap.add_argument('--latency-batches', default='1,8',
                    help='viewpoints per GPU and step of the untimed-by-headline config.latency records (SURVEY 8d C4: '
                         'vocc.py runs samples_per_gpu=1); empty string: none')
    ap.add_argument('--latency-steps', type=int, default=10)
    ap.add_argument('--graph-max-batch', type=int, default=8,
                    help='config.latency records up to this many viewpoints per step are hipGraph replays (one rank only; 0: '
                         'always eager): the lifting step as ONE graph of forward + loss + backward + ClipAdamW '
                         '(graphs.GraphedLiftStep), the full multi-task workload with its head forward / backward as two graphs '
                         '(graphs.GraphedHead; capped at 4 viewpoints there)')
    ap.add_argument('--graph-full-train', type=int, default=0,
                    help='config.full_train sub-record (64 viewpoints per step, one rank): 0 (default) = the eager step, whose '
                         'Hungarian assignment is started in forward() and solved on the host under the occupancy head '
                         '(138-139 ms); 1 = the head\'s forward and backward replayed as two hipGraphs (vln-ver_amd/graphs.py; '
                         '144-148 ms: the replay removes launch gaps the 64-viewpoint step does not have, and the assignment '
                         'waits for the whole forward graph)')
    ap.add_argument('--host-fed-steps', type=int, default=3,
                    help='steps of the config.host_fed record: the same step with the features handed over in (pinned) HOST '
                         'memory, as the detector does, the PCIe copy inside the timed region; 0: none')
    ap.add_argument('--sub-records', default='full_train,fp32',
                    help='untimed-by-headline sub-records measured in the same process after the headline (default workload '
                         'only): full_train = BASELINE configs[4] (vocc_full_train, bf16, 64 viewpoints per step), fp32 = the '
                         'default workload in fp32 at 8 viewpoints per step; empty string: none')
    ap.add_argument('--sub-steps', type=int, default=3)
    ap.add_argument('--torch-optimizer', action='store_true',
                    help='clip_grad_norm_ + torch.optim.AdamW(fused=True) instead of optim.ClipAdamW (ver_clip_adamw_step)')
    return ap.parse_args()


def spawn_ranks(args):
    """`--gpus N` without a launcher: become the launcher.  This process has not initialised the GPU (importing torch
    does not) and never will: the ranks are children, their stdout/stderr are ours, their exit code is returned."""
    n_dev = torch.cuda.device_count()           # counts devices without creating a HIP context on this image
    if args.backend == 'nccl' and n_dev < args.gpus:
        raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible (RCCL needs one GPU per rank; '
                         '--backend gloo shares GPUs between ranks)' % (args.gpus, n_dev))
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)                                  # (HSA_ENABLE_IPC_MODE_LEGACY=0 was set at the top of main())
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


class LiftTrainer(torch.nn.Module):
    """vocc.py head restricted to the lifting path (encoder + occupancy branch + its loss).
    The encoder runs on the whole per-GPU batch (one fused-gather launch per layer covers all
    viewpoints); the 120x120x35 head runs in micro-batches to bound activation memory."""

    def __init__(self, head, micro, dtype):
        super().__init__()
        self.head = head
        self.micro = micro
        self.autocast = dtype == 'bf16'

    def forward(self, feats, w2p, org, gt):
        bs = feats.shape[1]
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=self.autocast):
            emb = self.head(feats, None, only_bev=True, world2pixel=w2p, origin=org)   # [bs,Nq,C]
            total = emb.new_zeros((), dtype=torch.float32)
            for s in range(0, bs, self.micro):
                nb = min(self.micro, bs - s)
                # (logits stay in the GEMMs' row order, the targets are permuted to match: same loss, same gradients)
                part = emb if nb == bs else emb[s:s + nb]          # (the whole batch: the tensor itself, with its bf16 side copy)
                total = total + self.head.occupancy_loss_from_volume(part, gt[s:s + nb]) * nb
        return total / bs


class FullTrainer(torch.nn.Module):
    """BASELINE.json configs[4]: the whole vocc.py head (encoder, 6-layer detection decoder, cls/reg
    branches, coarse-to-fine occupancy) with the reference's loss dict (Hungarian targets on the
    host, focal + L1 + occupancy focal)."""

    def __init__(self, head, dtype):
        super().__init__()
        self.head = head
        self.autocast = dtype == 'bf16'

    def forward(self, feats, w2p, org, gt, gt_boxes, gt_labels):
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=self.autocast):
            # (targets_for: the Hungarian cost matrices leave for the host right behind the decoder and are solved there
            #  while the GPU runs the occupancy head; head.loss picks the assignment up)
            outs = self.head(feats, None, world2pixel=w2p, origin=org, occupancy_rows=True, targets_for=(gt_boxes, gt_labels))
        outs = {k: (v.float() if torch.is_tensor(v) and k != 'occupancy_preds' else v) for k, v in outs.items()}
        losses = self.head.loss(gt_boxes, gt_labels, gt, outs)
        return sum(losses.values())


def make_optimizer(params, own=True):
    """-> (optimizer, update): the reference's step (vocc.py:268-274: AdamW lr 1e-4 / weight decay 0.01 behind
    grad_clip max_norm) -- by default ``optim.ClipAdamW`` (clip + AdamW as one C-ABI call, ver_clip_adamw_step), with
    ``--torch-optimizer`` torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW(fused=True): the same arithmetic in four passes."""
    if own:
        opt = importlib.import_module('vln-ver_amd.optim').ClipAdamW(params, lr=1e-4, weight_decay=0.01, max_norm=300.0)

        def update():
            opt.step()
            opt.zero_grad(set_to_none=True)
    else:
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)

        def update():
            torch.nn.utils.clip_grad_norm_(params, 300.0)       # vocc.py:274 grad_clip max_norm
            opt.step()
            opt.zero_grad(set_to_none=True)
    return opt, update


def build_model(args, dev):
    pkg = importlib.import_module('vln-ver_amd')
    syn = importlib.import_module('vln-ver_amd.synthetic')
    config = importlib.import_module('vln-ver_amd.config')
    # the model dict comes from the config FILE (tools/train.py:105-135: Config.fromfile -> build_model), through the
    # package's own loader; the head gets train_cfg.pts the way the detector passes it on
    model_cfg = config.load_model_cfg(args.config)
    if args.workload == 'c2_single_scale_fwd':          # BASELINE configs[1]: the same head on a 50x50x16 single-scale grid
        cfg = config.head_cfg(model_cfg, train=False, bev_z=16, bev_h=50, bev_w=50, refine_occ=False)
        cfg['positional_encoding'].update(row_num_embed=50, col_num_embed=50, z_num_embed=16)
    else:
        cfg = config.head_cfg(model_cfg, train=args.workload == 'vocc_full_train')
    torch.manual_seed(2)
    head = pkg.registry.build_head(cfg)
    head.init_weights()
    if args.workload == 'vocc_full_train':
        # add_layout is off in vocc.py: the layout branches / layout query embedding are built (state-dict parity
        # with the reference) but never run, and DDP without find_unused_parameters stalls on parameters that get
        # no gradient -- freeze them, like everything else the step does not touch
        for k, p in head.named_parameters():
            if k.startswith(('layout_branches.', 'query_layout_embedding.')):
                p.requires_grad_(False)
        # (main() additionally freezes whatever a probing step leaves without a gradient, e.g. the positional
        # encoding the reference computes but never adds in this encoder: SURVEY.md section 0.6)
        n_train = sum(p.numel() for p in head.parameters() if p.requires_grad)
        return pkg, syn, head.to(dev), n_train
    # the lifting path does not touch the detection decoder / branches: freeze them so that
    # DDP reduces (and AdamW updates) exactly the parameters the path trains
    lift_prefixes = ('transformer.encoder.', 'transformer.level_embeds', 'transformer.cams_embeds',
                     'voxel_embedding.', 'up_sample.', 'occ_proj.', 'occ_branches.')
    n_train = 0
    for k, p in head.named_parameters():
        p.requires_grad_(k.startswith(lift_prefixes))
        n_train += p.numel() if p.requires_grad else 0
    return pkg, syn, head.to(dev), n_train


def gather_algorithmic_bytes(hit_counts, B, value_bytes, ncam=6, nk=196, c=768, heads=8, points=8, grad_slots_bytes=4):
    """SURVEY.md 8(d) with `s` = the bytes per element of each operand AS THE KERNEL SEES IT: per viewpoint and
    layer, forward = value once (6*196*768*s_v, s_v = 2 under bf16 autocast: value_proj emits bf16) +
    Sigma_n*(128 + 64 + 768)*4 (offsets, logits and the output row of every visible (camera, voxel) pair are fp32);
    backward = value read (s_v) + d(value) written (s_v too: the matrix-core backward writes bf16 for bf16 tiles,
    fp32 for fp32 tiles) + Sigma_n*(768*s_g + (2*192 + 192)*4), s_g = bytes per element of the slots' gradient rows.
    (Round 1 priced the bf16 value tensor at 4 B/element: its 0.48 was 0.36 by this rule.)"""
    sn = float(hit_counts)
    nval = B * ncam * nk * c
    fwd = nval * value_bytes + sn * (heads * points * 2 + heads * points + c) * 4
    # (round 4: under bf16 autocast the grad rows of the slots arrive in bf16 and are read as such -- priced at their size)
    bwd = nval * (value_bytes + value_bytes) + sn * (c * grad_slots_bytes + (2 * 192 + 192) * 4)
    return fwd, bwd


def source_hash(name='ver_sca.hip'):
    import hashlib
    with open(os.path.join(ROOT, 'vln-ver_amd', 'csrc', name), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def measured_traffic(kernels, B):
    """HBM bytes per launch of the gather kernels from the committed rocprofv3 PMC passes of THIS command
    (profiles/rNN_bench_pmc_fetch_write.csv: `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py ...` and
    the same with WRITE_SIZE, separate runs, scratch/r04/run_bench_profiles.sh).  Units are KiB per dispatch;
    FETCH_SIZE is doubled (on gfx950 it reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM;
    the value tiles arrive as 16-byte-per-lane LDS-DMA), WRITE_SIZE is taken as reported (calibrated on
    k_zero_rows, whose byte count is known: profiles/README.md).  Only used when the profile was taken at the
    same number of viewpoints per launch AND from the kernel source that is running now: the CSV header carries the
    sha256 of csrc/ver_sca.hip at profiling time, a different (or missing) hash returns None."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_pmc_fetch_write.csv')))
    if not files:
        return None, None
    vals, batch, sha = {}, None, None
    for line in open(files[-1]):
        if line.startswith('# viewpoints_per_launch'):
            batch = int(line.split('=')[1])
        if line.startswith('# ver_sca_sha256'):
            sha = line.split('=')[1].strip()
    if batch != B or sha != source_hash():
        return None, None
    for row in csv.reader(l for l in open(files[-1]) if not l.startswith('#')):
        if len(row) == 4 and any(k in row[0] for k in kernels) and row[1] in ('FETCH_SIZE', 'WRITE_SIZE'):
            vals[row[1]] = vals.get(row[1], 0.0) + float(row[2])          # several kernels of one launch add up
    if len(vals) != 2:
        return None, None
    return (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0, os.path.basename(files[-1])


def cpu_baseline(head, syn, seconds, dev=None, dtype='bf16'):
    """The CPU oracle (oracle/ver_oracle.py = pinned restatement of the reference) on this host's cores: the vocc.py
    lifting path fwd+bwd at FULL size, one viewpoint per pass (the reference's samples_per_gpu=1): 1 untimed warm-up
    pass, then timed passes until `seconds` are spent (at least 5, at most 10); value = 1 / median pass time.
    The warm-up pass doubles as a PARITY PROBE of the run that was just timed (`dev` given): the product evaluates the same
    viewpoint, labels and (post-step) parameters on the GPU in the bench's arithmetic -- one forward + backward of the
    training kernels, dropout off -- and `loss_oracle`, `loss_gpu`, `rel_diff` (+ the gradient norms over the path's
    parameters) go on the line; main() exits non-zero when the losses are more than 2e-2 apart."""
    oracle = importlib.import_module('oracle.ver_oracle')
    # all cores of a 256-thread host oversubscribe these small CPU ops (measured 6x slower than 8
    # threads); 16 is near the knee.  `cores` reports what was actually used.
    host_threads = os.cpu_count() or 1
    torch.set_num_threads(min(16, host_threads))
    cpu_model = 'unknown'
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                cpu_model = ln.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    p = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point() and k != 'code_weights')
         for k, v in head.state_dict().items()}
    w2p, org = syn.camera_batch(1, seed=1)
    feats = torch.from_numpy(syn.vit_features(1, seed=0))[0].unsqueeze(1)
    gt = torch.from_numpy(np.random.default_rng(3).integers(0, 17, size=504000))

    probe = {}

    def one_pass(keep=False):
        t = time.perf_counter()
        _, occ = oracle.lifting_forward(p, feats, torch.from_numpy(w2p[0]), torch.from_numpy(org[0]))
        loss = oracle.focal_loss(occ[0], gt, avg_factor=(gt < 16).sum() * 1.0)
        loss.backward()
        dt = time.perf_counter() - t
        if keep:
            probe['loss_oracle'] = float(loss)
            probe['grad_norm_oracle'] = float(torch.sqrt(sum(v.grad.double().pow(2).sum() for k, v in p.items()
                                                             if v.grad is not None and k in trained)))
        for v in p.values():
            v.grad = None
        return dt

    trained = {k for k, v in head.named_parameters() if v.requires_grad}
    if dev is not None:
        # the product on the same viewpoint / labels / parameters (before the oracle's threads are set: GPU work only)
        was_training = head.training
        head.eval()
        for v in head.parameters():
            v.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == 'bf16'):
            emb = head(feats.to(dev), None, only_bev=True, world2pixel=torch.from_numpy(w2p).to(dev), origin=torch.from_numpy(org).to(dev))
            loss_gpu = head.occupancy_loss_from_volume(emb, gt.to(dev)[None])
        loss_gpu.backward()
        torch.cuda.synchronize()
        probe['loss_gpu'] = float(loss_gpu)
        probe['grad_norm_gpu'] = float(torch.sqrt(sum(v.grad.double().pow(2).sum() for k, v in head.named_parameters()
                                                      if v.grad is not None and k in trained)))
        for v in head.parameters():
            v.grad = None
        head.train(was_training)
        del emb, loss_gpu

    t0 = time.perf_counter()
    warm = one_pass(keep=True)
    if dev is not None:
        probe['rel_diff'] = abs(probe['loss_gpu'] - probe['loss_oracle']) / max(abs(probe['loss_oracle']), 1e-30)
        probe['grad_norm_rel_diff'] = abs(probe['grad_norm_gpu'] - probe['grad_norm_oracle']) / max(probe['grad_norm_oracle'], 1e-30)
        probe = {k: float('%.6g' % v) for k, v in probe.items()}
        probe['probe'] = ('viewpoint seed 0 / rig seed 1 / labels seed 3 through the product at 1 viewpoint per step (%s, training '
                          'kernels, dropout off) against the oracle\'s warm-up pass on the same post-step parameters' % dtype)
    else:
        probe = {}
    times = []
    while len(times) < 5 or (len(times) < 10 and time.perf_counter() - t0 + warm < seconds):
        times.append(one_pass())
    med = statistics.median(times)
    return dict(value=1.0 / med, unit='viewpoints/s', cores=torch.get_num_threads(), kind='port',
                host_threads=host_threads, cpu_model=cpu_model, passes=len(times), warmup_passes=1, **probe,
                median_s=round(med, 3), min_s=round(min(times), 3), max_s=round(max(times), 3),
                sample='%d timed single-viewpoint passes after 1 warm-up (median), vocc.py 15x15x4 -> 120x120x35x16 '
                       'lifting path fwd+bwd at full size, fp32, oracle/ver_oracle.py (torch-CPU, %d threads of %d), '
                       '%.1f s in all' % (len(times), torch.get_num_threads(), host_threads, time.perf_counter() - t0))


def sub_record(base, name, dev, rank, world, distributed):
    """One untimed-by-headline sub-record of the default line (`config.full_train`, `config.fp32`): another workload /
    precision built and stepped in the same process after the headline region -- its own head, optimiser and inputs,
    two priming steps + 1 warm-up, then `--sub-steps` timed steps between barriers, max over ranks.  Never part of `value`."""
    a = argparse.Namespace(**vars(base))
    name, _, nb = name.partition(':')           # `name:batch` overrides the viewpoints per step (tests)
    if name == 'full_train':
        a.workload, a.dtype, a.batch = 'vocc_full_train', 'bf16', 64
    elif name == 'fp32':
        a.workload, a.dtype, a.batch, a.micro = 'vocc_c2f_train', 'fp32', 8, 8
    else:
        raise SystemExit('bench.py: unknown sub-record %r' % name)
    if nb:
        a.batch = int(nb)
        a.micro = min(a.micro, a.batch)
    torch.cuda.reset_peak_memory_stats()
    pkg, syn, head, n_train = build_model(a, dev)
    full = a.workload == 'vocc_full_train'
    B = a.batch
    model = (FullTrainer(head, a.dtype) if full else LiftTrainer(head, a.micro, a.dtype)).to(dev).train()
    w2p_np, org_np = syn.camera_batch(B, seed=1 + rank)
    feats = torch.from_numpy(syn.vit_features(B, seed=100 + rank)).to(dev).permute(1, 0, 2, 3).contiguous()
    w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
    gt = torch.from_numpy(np.random.default_rng(7 + rank).integers(0, 17, size=(B, head.voxel_num))).to(dev)
    extra = ()
    if full:
        gts = [syn.detection_gt(seed=40 + rank * 1000 + i, num_gt=3 + i % 5) for i in range(B)]
        extra = ([torch.from_numpy(g[0][:, :7]).to(dev) for g in gts], [torch.from_numpy(g[1]).to(dev) for g in gts])
        # parameters the full head's forward never touches (layout branches aside: build_model froze them) -- one probing step
        model(feats[:, :2], w2p[:2], org[:2], gt[:2], extra[0][:2], extra[1][:2]).backward()
        for prm in model.parameters():
            if prm.requires_grad and prm.grad is None:
                prm.requires_grad_(False)
            prm.grad = None
    net = model
    if distributed:
        net = importlib.import_module('vln-ver_amd.ddp').wrap_ddp(model, device=dev, bf16_gradients=base.backend == 'nccl')
    params = [prm for prm in model.parameters() if prm.requires_grad]
    opt, update = make_optimizer(params, own=not base.torch_optimizer)
    graphed = None
    if full and not distributed and base.graph_full_train:
        # same kernels, same arithmetic, replayed: Hungarian targets, loss terms, clip and AdamW stay eager (graphs.py)
        graphed = importlib.import_module('vln-ver_amd.graphs').GraphedHead(head, feats, w2p, org, autocast_dtype=torch.bfloat16)

    def step():
        if graphed is not None:
            loss = sum(head.loss(extra[0], extra[1], gt, graphed(feats, w2p, org)).values())
        else:
            loss = net(feats, w2p, org, gt, *extra)
        loss.backward()
        update()
        return loss
    for _ in range(3):                          # two priming steps (allocator, AdamW state) + 1 warm-up
        last = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(base.sub_steps):
        last = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if distributed:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    assert torch.isfinite(last).all(), 'non-finite loss in sub-record %s' % name
    ms = float(dt) / base.sub_steps * 1e3
    rec = dict(workload=a.workload, dtype=a.dtype, viewpoints_per_gpu_per_step=B, head_micro_batch=(None if full else a.micro),
               steps=base.sub_steps, warmup=1, graphed=graphed is not None, ms_per_step=round(ms, 3),
               viewpoints_per_s=round(B * world / ms * 1e3, 2),
               trainable_params=sum(prm.numel() for prm in params),
               peak_hbm_gib=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
    del opt, net, model, head, last, graphed, step
    torch.cuda.empty_cache()
    return rec


def main():
    args = parse()
    # dmabuf IPC: the mode this image's host driver supports (task environment notes: without it RCCL / cross-process
    # tensor sharing fails in hipIpcGetMemHandle).  Set before the first GPU call on EVERY path -- self-spawned ranks
    # inherit it, ranks started by an external launcher (the driver's torch.distributed.run) get it here.
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # hipGraph replay (config.latency): the runtime's AQL packet capture loses the order between memset nodes and the kernels
    # behind them on this ROCm (vln-ver_amd/__init__.py); read when the HIP runtime starts, so set before any GPU call
    os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))             # (nothing above this line touches the GPU)
    under_launcher = 'WORLD_SIZE' in os.environ
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        print('bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d'
              % (args.gpus, world, world), file=sys.stderr)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback of the product path)'
    if args.backend == 'nccl' and world > torch.cuda.device_count():
        raise SystemExit('bench.py: %d ranks but %d GPU(s): RCCL needs one GPU per rank' % (world, torch.cuda.device_count()))
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if under_launcher:
        # also with ONE rank (torchrun --nproc-per-node 1): the RCCL communicator, the bf16 compression hook and the
        # bucket views are then built and used on hardware exactly as with N ranks
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import datetime
        wait = datetime.timedelta(minutes=10)       # a peer that never arrives is an error, not a 30-minute hang
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev, timeout=wait)
        else:
            dist.init_process_group(args.backend, timeout=wait)
    distributed = dist.is_available() and dist.is_initialized()
    hip = importlib.import_module('vln-ver_amd.hipops')
    hip.lib()
    tuned = False
    if not args.no_tuned_gemms:
        tuned = importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
    pkg, syn, head, n_train = build_model(args, dev)
    if args.batch is None:
        args.batch = 192 if args.workload == 'vocc_c2f_train' and args.dtype == 'bf16' else 64
    B = args.batch
    train = args.workload in ('vocc_c2f_train', 'vocc_full_train')
    full = args.workload == 'vocc_full_train'
    model = (FullTrainer(head, args.dtype) if full else LiftTrainer(head, args.micro, args.dtype)).to(dev)
    model.train(train)
    ddp = model                                 # train(): dropout ON, as in the reference's step
    probe_unused = train and full               # the full head has parameters its forward never touches

    def wrap(m):
        return importlib.import_module('vln-ver_amd.ddp').wrap_ddp(m, device=dev, bf16_gradients=args.backend == 'nccl')

    if distributed and train and not probe_unused:
        ddp = wrap(model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt, update = make_optimizer(params, own=not args.torch_optimizer) if train else (None, None)

    # synthetic, HBM-resident inputs (different viewpoints per rank)
    w2p_np, org_np = syn.camera_batch(B, seed=1 + rank)
    feats = torch.from_numpy(syn.vit_features(B, seed=100 + rank)).to(dev).permute(1, 0, 2, 3).contiguous()
    w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
    nvox = head.voxel_num if head.refine_occ else head.bev_h * head.bev_w * head.occ_zdim
    gt = torch.from_numpy(np.random.default_rng(7 + rank).integers(0, 17, size=(B, nvox))).to(dev)
    gt_boxes = gt_labels = None
    if full:
        gts = [syn.detection_gt(seed=40 + rank * 1000 + i, num_gt=3 + i % 5) for i in range(B)]
        gt_boxes = [torch.from_numpy(g[0][:, :7]).to(dev) for g in gts]
        gt_labels = [torch.from_numpy(g[1]).to(dev) for g in gts]

    def make_step(nb, graph=False):
        """One step over the first `nb` viewpoints of the resident inputs.  `graph` (full multi-task step on one rank
        only): the head's forward and backward are replayed as two hipGraphs (vln-ver_amd/graphs.py), everything else
        -- Hungarian targets, loss terms, clip, AdamW -- runs as in the eager step."""
        f, w, o, g = feats[:, :nb].contiguous(), w2p[:nb], org[:nb], gt[:nb]
        gb, gl = (gt_boxes[:nb], gt_labels[:nb]) if full else (None, None)
        if graph and not full:
            # the whole lifting step as one hipGraph (its construction runs three eager steps, then captures)
            lift = importlib.import_module('vln-ver_amd.graphs').GraphedLiftStep(model, opt, f, w, o, g)
            f, w, o, g = lift.inputs
            return lambda: lift(f, w, o, g)
        if graph:
            graphed = importlib.import_module('vln-ver_amd.graphs').GraphedHead(
                head, f, w, o, autocast_dtype=torch.bfloat16 if args.dtype == 'bf16' else None)

            def graph_step():
                loss = sum(head.loss(gb, gl, g, graphed(f, w, o)).values())
                loss.backward()
                update()
                return loss
            return graph_step

        def step():
            if train:
                loss = ddp(f, w, o, g, gb, gl) if full else ddp(f, w, o, g)
                loss.backward()
                update()                                            # vocc.py:268-274: grad_clip + AdamW
                return loss
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=args.dtype == 'bf16'):
                emb = head(f, None, only_bev=True, world2pixel=w, origin=o)
                out = [head.occupancy_from_volume(emb[s:s + args.micro]) for s in range(0, nb, args.micro)]
                return out[-1].float().mean()
        return step

    if probe_unused:
        # One un-wrapped probing step: parameters that end it without a gradient are not part of this workload's
        # graph (the reference pays find_unused_parameters=True for them, apis/mmdet_train.py:73); freeze them so
        # that DDP reduces -- and AdamW updates -- exactly what the step trains.
        model(feats[:, :2], w2p[:2], org[:2], gt[:2], gt_boxes[:2], gt_labels[:2]).backward()
        for prm in model.parameters():
            if prm.requires_grad and prm.grad is None:
                prm.requires_grad_(False)
            prm.grad = None
        params = [prm for prm in model.parameters() if prm.requires_grad]
        n_train = sum(prm.numel() for prm in params)
        opt, update = make_optimizer(params, own=not args.torch_optimizer)
        if distributed:
            ddp = wrap(model)
    step = make_step(B)
    # Setup: two untimed priming steps.  The first step allocates ~55 GiB through hipMalloc and creates
    # the AdamW state, so the caching allocator still grows during the second; with them here the W
    # warm-up steps the caller asks for (even W = 0) are not spent on one-time allocator / library work.
    # (The host-side label-range check of the fused focal loss also runs here, once: dense_heads/losses.py.)
    for _ in range(2):
        last = step()
    for _ in range(args.warmup):
        last = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    timer = hip.KernelTimer()
    hip.KERNEL_TIMER = timer
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    hip.KERNEL_TIMER = None
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)
    assert torch.isfinite(last).all(), 'non-finite loss'
    peak_gib = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
    last = None                                 # (no loss of an eager step may be alive when graphs are captured: graphs.py)

    # config.latency: the same step at the reference's own batch points (samples_per_gpu = 1, and 8),
    # measured AFTER the headline region and never part of `value`; every rank runs them (the gradient sum is collective)
    latency = []
    for nb in [int(x) for x in args.latency_batches.split(',') if x.strip()]:
        if not train or nb > B:
            continue
        # the full multi-task step is host bound at these sizes (~1 500 module calls + as many autograd nodes): on one
        # rank its forward / backward are replayed as hipGraphs; `graphed` in the record says which form was timed
        graphed = not distributed and (nb <= min(args.graph_max_batch, 4) if full
                                       else nb <= args.graph_max_batch and not args.torch_optimizer)
        small = make_step(nb, graph=graphed)
        for _ in range(3):
            small()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.latency_steps):
            small()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if distributed:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        ms = float(dt) / args.latency_steps * 1e3
        if graphed and not full:
            # the replayed steps must have TRAINED: finite loss and parameters (a graph that silently produced garbage --
            # vln-ver_amd/__init__.py -- would otherwise only show in the parity probe at the end)
            assert torch.isfinite(small()).all() and all(torch.isfinite(p).all() for p in params), \
                'non-finite loss / parameters after the replayed %d-viewpoint steps' % nb
            opt.zero_grad(set_to_none=True)          # (the gradients live in the graph's pool: the next eager step makes its own)
        latency.append(dict(viewpoints_per_gpu_per_step=nb, steps=args.latency_steps, warmup=3,
                            graphed=graphed,
                            ms_per_step=round(ms, 3), viewpoints_per_s=round(nb * world / ms * 1e3, 2)))

    # config.host_fed: the step as the reference's detector drives it -- the six feature maps of every viewpoint arrive
    # in HOST memory (detectors/voxelformer.py:285-289) -- with the host -> HBM copy on the step's stream inside the timed
    # region.  The PCIe-inclusive rate; measured after the headline region, never part of `value`.
    host_fed = None
    if train and args.host_fed_steps > 0:
        host = torch.empty(feats.shape, dtype=feats.dtype, pin_memory=True)
        host.copy_(feats)

        def fed():
            feats.copy_(host, non_blocking=True)
            return step()
        fed()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        t2 = time.perf_counter()
        for _ in range(args.host_fed_steps):
            fed()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t2], device=dev, dtype=torch.float64)
        if distributed:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        ms = float(dt) / args.host_fed_steps * 1e3
        host_fed = dict(steps=args.host_fed_steps, host_bytes_per_step=host.numel() * host.element_size(),
                        ms_per_step=round(ms, 3), viewpoints_per_s=round(B * world / ms * 1e3, 2))
        del host

    # config.full_train / config.fp32: BASELINE configs[4] and the fp32 form of the headline workload on the driver's record
    subs = {}
    if train and not full and args.dtype == 'bf16':
        for name in [x.strip() for x in args.sub_records.split(',') if x.strip()]:
            subs[name.partition(':')[0]] = sub_record(args, name, dev, rank, world, distributed)

    exit_code = 0
    if rank == 0:
        kt = timer.summary()
        hit = hip.project_points(w2p, org, head.point_cloud_range, head.bev_z, head.bev_h, head.bev_w)
        sigma_n = int(hit.vis_cnt.sum())
        vbytes = 2 if args.dtype == 'bf16' else 4          # value_proj emits bf16 under autocast
        fwd_b, bwd_b = gather_algorithmic_bytes(sigma_n, B, vbytes, grad_slots_bytes=2 if args.dtype == 'bf16' else 4)
        roof, others = None, []
        for name, byts, kernels in (('ver_sca_forward', fwd_b, ('k_sca_fwd', 'k_zero_rows')),
                                    ('ver_sca_backward', bwd_b, ('k_sca_bwd',))):
            if name in kt and kt[name]['count']:
                avg_ms = kt[name]['ms'] / kt[name]['count']
                ach = byts / (avg_ms * 1e-3) / 1e9
                traffic, src = measured_traffic(kernels, B)
                obj = dict(kernel=name, bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                           frac=round(ach / HBM_PEAK_GBS, 4), traffic=int(traffic) if traffic else None,
                           traffic_source=(None if src is None else '%s (committed PMC pass of this command on another box, '
                                           'keyed to the sha256 of csrc/ver_sca.hip; not a measurement of this run)' % src),
                           avg_launch_us=round(avg_ms * 1e3, 2),
                           launches=kt[name]['count'], algorithmic_bytes_per_launch=int(byts),
                           value_bytes_per_element=vbytes, viewpoints_per_launch=B, sigma_n=sigma_n)
                if name == 'ver_sca_forward':
                    # the zero fill of the rows the gather accumulates into (the reference's zeros_like inside the op,
                    # spatial_cross_attention.py:139) runs on a side stream under the projection GEMMs: its duration there,
                    # and what of it the launch stream still had to wait for in front of the gather (events on that stream
                    # around the join) -- `frac_with_zero_fill` prices the unhidden part with the kernel
                    zr, zw = kt.get('ver_sca_zero_rows'), kt.get('ver_sca_zero_wait')
                    if zr and zw and zr['count'] and zw['count']:
                        fl = kt.get('ver_event_floor')
                        floor_ms = fl['ms'] / fl['count'] if fl and fl['count'] else 0.0
                        join_ms = zw['ms'] / zw['count']
                        unhidden_ms = max(0.0, join_ms - floor_ms)
                        obj['zero_fill'] = dict(side_stream_us=round(zr['ms'] / zr['count'] * 1e3, 2), join_us=round(join_ms * 1e3, 2),
                                                empty_bracket_us=round(floor_ms * 1e3, 2), unhidden_us=round(unhidden_ms * 1e3, 2),
                                                launches=zr['count'], bytes=int(hit.zero_cnt.sum()) * 768 * 4,
                                                note='zero fill of the rows the gather accumulates into, on a side stream under the '
                                                     'projection GEMMs; join = events on the launch stream around the wait for it, '
                                                     'unhidden = join - an empty event bracket')
                        obj['frac_with_zero_fill'] = round(byts / ((avg_ms + unhidden_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    roof = obj
                else:
                    others.append(obj)
        # the dense part ("MFMA utilisation for the dense part"): the head's GEMMs -- the three lattice layers
        # and occ_proj (which kernel runs which product: the `implementation` strings below) -- bracketed by HIP events on
        # the launch stream in the timed region; flops = 2 m k n of every GEMM AS EXECUTED (with the constant / pad columns
        # the operands carry, ~3 % over the useful count).  `whole_step` prices the USEFUL multiply-adds of one step
        # (473.9 GFLOP per viewpoint forward, x 3 with the two backward products) on the step time.
        classes = (('ver_gemm_nn', 'forward of the lattice layers: ver_gemm_nn_segments, operands read from the lattice (csrc/ver_gemm.hip)'),
                   ('head_gemm_fwd', 'forward of occ_proj (and of small batches): hipBLASLt N x N'),
                   ('head_gemm_dgrad', 'd(input): lattice layers ver_gemm_nn_planes (one product per layer, operands read from the '
                                       'gradient lattice), occ_proj ver_gemm_nn (832-column groups) / hipBLASLt (768-column groups)'),
                   ('ver_wgrad_tn', 'd(weight): ver_wgrad_tn / ver_wgrad_tn_segments (csrc/ver_wgrad.hip)'))
        tot_f = tot_ms = 0.0
        for name, impl in classes:
            if name in kt and kt[name]['ms'] > 0:
                tf = kt[name]['flops'] / (kt[name]['ms'] * 1e-3) / 1e12
                tot_f += kt[name]['flops']
                tot_ms += kt[name]['ms']
                others.append(dict(kernel=name, implementation=impl, bound='mfma', achieved=round(tf, 1), peak=MFMA_BF16_PEAK_TF,
                                   unit='TFLOP/s', frac=round(tf / MFMA_BF16_PEAK_TF, 4), launches=kt[name]['count'],
                                   ms_per_step=round(kt[name]['ms'] / args.steps, 3),
                                   tflop_per_step=round(kt[name]['flops'] / args.steps / 1e12, 3)))
        if tot_ms > 0 and args.dtype == 'bf16':
            tf = tot_f / (tot_ms * 1e-3) / 1e12
            others.append(dict(kernel='head_gemms', bound='mfma', achieved=round(tf, 1), peak=MFMA_BF16_PEAK_TF, unit='TFLOP/s',
                               frac=round(tf / MFMA_BF16_PEAK_TF, 4), ms_per_step=round(tot_ms / args.steps, 3),
                               tflop_per_step=round(tot_f / args.steps / 1e12, 3),
                               note='lattice layers 1-3 + occ_proj: forward, d(input), d(weight); HIP events, flops as executed'))
        if train and not full and args.dtype == 'bf16' and args.workload == 'vocc_c2f_train' and args.config is None:
            useful = 3 * 473.9e9 * B * args.steps                       # fwd + dgrad + wgrad of every GEMM of the path
            tf = useful / elapsed / 1e12
            others.append(dict(kernel='whole_step', bound='mfma', achieved=round(tf, 1), peak=MFMA_BF16_PEAK_TF, unit='TFLOP/s',
                               frac=round(tf / MFMA_BF16_PEAK_TF, 4),
                               note='useful multiply-adds of the step (473.9 GFLOP per viewpoint forward x 3) / step time'))
        total_vp = B * world * args.steps
        line = {
            'metric': 'viewpoints/sec (multi-view->voxel fwd+bwd), vocc.py config' if train
                      else 'viewpoints/sec (multi-view->voxel fwd), 50x50x16 single-scale',
            'value': round(total_vp / elapsed, 3), 'unit': 'viewpoints/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic (N(0,1) ViT features, 6-camera pinhole rig, random occupancy labels; '
                    'reference-rule random-init weights)',
            'config': {'workload': ('vocc.py full multi-task head (encoder + detection decoder + cls/reg branches + '
                                    'coarse-to-fine occupancy), reference loss dict, fwd+bwd+AdamW') if full else
                                   'vocc.py coarse-to-fine lifting path (15x15x4 -> 120x120x35x16): encoder + '
                                   'occupancy head + focal loss, fwd+bwd+AdamW' if train else
                                   'single-scale 50x50x16 volume, forward only',
                       'config_file': os.path.relpath(args.config or importlib.import_module('vln-ver_amd.config').VOCC, ROOT),
                       'viewpoints_per_gpu_per_step': B, 'global_viewpoints_per_step': B * world,
                       'head_micro_batch': args.micro, 'parallelism': 'dp%d' % world,
                       'gradient_allreduce': (('RCCL, bf16-compressed 200 MB buckets' if args.backend == 'nccl'
                                               else args.backend) if distributed and train else None),
                       'trainable_params': n_train, 'tuned_gemm_table': tuned,
                       'optimizer': (None if not train else 'torch clip_grad_norm_ + AdamW(fused)' if args.torch_optimizer
                                     else 'optim.ClipAdamW (ver_clip_adamw_step)'),
                       'peak_hbm_gib': peak_gib,
                       'arithmetic': ('bf16 autocast GEMMs and lattices; multi-view gather on bf16 value tiles: packed-fp16 point '
                                      'accumulation (<= 8 terms per voxel, head, corner), fp32 after the corner fold; fp32 '
                                      'offsets / softmax / LayerNorm statistics / loss') if args.dtype == 'bf16' else 'fp32',
                       'evaluation': ('same loss and gradients as the reference step, evaluated where the data is: '
                                      'occupancy logits stay in the GEMM row order and the targets are permuted to match; '
                                      'on the bf16 path occ_branches[0] is composed with occ_proj every step '
                                      '(DESIGN.md sections 1, 3.3, 6)') if train else 'forward only',
                       'latency': latency, 'host_fed': host_fed,
                       'full_train': subs.get('full_train'), 'fp32': subs.get('fp32')},
            'roofline': roof, 'roofline_other_kernels': others,
        }
        if world == 1 and not args.no_cpu_baseline and train and not full:
            line['cpu_baseline'] = cpu_baseline(head, syn, args.cpu_seconds, dev, args.dtype)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
        cb = line['cpu_baseline']
        if cb and cb.get('rel_diff') is not None and not cb['rel_diff'] <= 2e-2:
            print('bench.py: PARITY PROBE FAILED: loss on the GPU %.6g, oracle %.6g (rel. diff %.3g > 2e-2)'
                  % (cb['loss_gpu'], cb['loss_oracle'], cb['rel_diff']), file=sys.stderr)
            exit_code = 3
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == '__main__':
    main()
