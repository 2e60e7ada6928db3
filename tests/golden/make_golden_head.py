"""Golden vectors for the head-level path (a10 + "next" rows): builds the REFERENCE's
``VoxelFormerOccupancyHead`` verbatim from the ``model['pts_bbox_head']`` dict of
projects/configs/verformer/vocc.py and runs its forward (and an occupancy-loss backward).

Adds to ``_mmcv_stub`` the mmdet / mmdet3d symbols the head file imports (restated from the
published behaviour of mmdet 2.14.0; not vendored by the reference).
Build-container only; loaded by make_golden.py."""
import copy
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

import _mmcv_stub as stub
import cases

syn = importlib.import_module('vln-ver_amd.synthetic')
T = torch.from_numpy
HERE = os.path.dirname(os.path.abspath(__file__))
REF_CFG = stub.REFERENCE_ROOT + '/projects/configs/verformer/vocc.py'


def _weight_reduce(loss, weight, reduction, avg_factor):
    """mmdet.models.losses.utils.weight_reduce_loss."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return {'mean': loss.mean, 'sum': loss.sum, 'none': lambda: loss}[reduction]()
    assert reduction == 'mean'
    return loss.sum() / avg_factor


class FocalLoss(nn.Module):
    """mmdet 2.14.0 FocalLoss, CPU path (py_sigmoid_focal_loss)."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        import torch.nn.functional as F
        nc = pred.size(1)
        t = F.one_hot(target, num_classes=nc + 1)[:, :nc].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        fw = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw
        if weight is not None and weight.shape != loss.shape:
            weight = weight.view(-1, 1)
        return self.loss_weight * _weight_reduce(loss, weight, reduction_override or self.reduction, avg_factor)


class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        if target.numel() == 0:
            return pred.sum() * 0
        return self.loss_weight * _weight_reduce((pred - target).abs(), weight,
                                                 reduction_override or self.reduction, avg_factor)


class GIoULoss(nn.Module):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight


for _c in (FocalLoss, L1Loss, GIoULoss):
    stub.LOSSES.register_module()(_c)


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class BaseAssigner:
    pass


class _Sampling:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :] if gt_bboxes.numel() else gt_bboxes.view(-1, 4)


class PseudoSampler:
    """mmdet 2.14.0 PseudoSampler.sample."""

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return _Sampling(pos, neg, bboxes, gt_bboxes, assign_result)


class FocalLossCost:
    """mmdet 2.14.0 FocalLossCost."""

    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


class IoUCost:
    def __init__(self, iou_mode='giou', weight=1.):
        self.weight = weight


stub.MATCH_COST.register_module()(FocalLossCost)
stub.MATCH_COST.register_module()(IoUCost)


class BaseBBoxCoder:
    def __init__(self, **kwargs):
        pass


class DETRHead(stub.BaseModule):
    """Attribute contract of mmdet 2.14.0 DETRHead.__init__."""

    def __init__(self, num_classes, in_channels, num_query=100, num_reg_fcs=2, transformer=None,
                 sync_cls_avg_factor=False, positional_encoding=None, loss_cls=None, loss_bbox=None,
                 loss_iou=None, train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.bg_cls_weight = 0
        self.sync_cls_avg_factor = sync_cls_avg_factor
        self.num_query = num_query
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.num_reg_fcs = num_reg_fcs
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.fp16_enabled = False
        self.loss_cls = stub.build_from_cfg(loss_cls, stub.LOSSES)
        self.loss_bbox = stub.build_from_cfg(loss_bbox, stub.LOSSES)
        self.loss_iou = stub.build_from_cfg(loss_iou, stub.LOSSES)
        self.cls_out_channels = num_classes if self.loss_cls.use_sigmoid else num_classes + 1
        if train_cfg:
            self.assigner = stub.build_from_cfg(train_cfg['assigner'], stub.BBOX_ASSIGNERS)
            self.sampler = PseudoSampler()
        self.positional_encoding = stub.build_positional_encoding(positional_encoding)
        self.transformer = stub.build_transformer(transformer)
        self.embed_dims = self.transformer.embed_dims
        assert positional_encoding['num_feats'] * 2 == self.embed_dims
        self._init_layers()


def _inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def install_head_stubs():
    stub.install()
    m = stub._mod
    m('mmdet.core', multi_apply=lambda f, *a, **k: tuple(map(list, zip(*map(f, *a)))),
      reduce_mean=lambda t: t).__path__ = []
    m('mmdet.core.bbox', BaseBBoxCoder=BaseBBoxCoder).__path__ = []
    m('mmdet.core.bbox.builder', BBOX_CODERS=stub.BBOX_CODERS, BBOX_ASSIGNERS=stub.BBOX_ASSIGNERS)
    m('mmdet.core.bbox.assigners', AssignResult=AssignResult, BaseAssigner=BaseAssigner)
    m('mmdet.core.bbox.match_costs',
      build_match_cost=lambda cfg: stub.build_from_cfg(cfg, stub.MATCH_COST)).__path__ = []
    m('mmdet.core.bbox.match_costs.builder', MATCH_COST=stub.MATCH_COST)
    sys.modules['mmdet.models'].HEADS = stub.HEADS
    m('mmdet.models.dense_heads', DETRHead=DETRHead)
    sys.modules['mmdet.models.utils.transformer'].inverse_sigmoid = _inverse_sigmoid
    for name in ('mmdet3d', 'mmdet3d.core', 'mmdet3d.core.bbox', 'mmdet3d.models'):
        m(name).__path__ = []
    m('mmdet3d.core.bbox.coders',
      build_bbox_coder=lambda cfg, **kw: stub.build_from_cfg(cfg, stub.BBOX_CODERS))
    m('mmdet3d.models.builder', build_loss=lambda cfg: stub.build_from_cfg(cfg, stub.LOSSES),
      build_head=lambda cfg: stub.build_from_cfg(cfg, stub.HEADS))
    m('h5py')
    root = stub.REFERENCE_ROOT + '/projects/mmdet3d_plugin'
    stub._pkg('projects.mmdet3d_plugin.core', root + '/core')
    stub._pkg('projects.mmdet3d_plugin.core.bbox', root + '/core/bbox')
    stub._pkg('projects.mmdet3d_plugin.core.bbox.coders', root + '/core/bbox/coders')
    stub._pkg('projects.mmdet3d_plugin.bevformer.dense_heads', root + '/bevformer/dense_heads')
    stub._pkg('projects.mmdet3d_plugin.core.bbox.assigners', root + '/core/bbox/assigners')
    stub._pkg('projects.mmdet3d_plugin.core.bbox.match_costs', root + '/core/bbox/match_costs')
    importlib.import_module('projects.mmdet3d_plugin.core.bbox.assigners.hungarian_assigner_3d')
    importlib.import_module('projects.mmdet3d_plugin.core.bbox.match_costs.match_cost')
    importlib.import_module('projects.mmdet3d_plugin.core.bbox.coders.nms_free_coder')
    importlib.import_module('projects.mmdet3d_plugin.core.bbox.coders.layout_coder')
    stub.ref_modules()
    return importlib.import_module(
        'projects.mmdet3d_plugin.bevformer.dense_heads.voxelformer_occupancy_head')


def reference_head_cfg():
    """exec the reference config (plain Python; `_base_` is just a list) -> model['pts_bbox_head']."""
    ns = {}
    exec(compile(open(REF_CFG).read(), REF_CFG, 'exec'), ns)
    return ns['model']['pts_bbox_head'], ns['model'].get('train_cfg')


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def gen_head(ref):
    from make_golden import _CameraDir, save
    head_mod = install_head_stubs()
    cfg, _ = reference_head_cfg()
    assert _plain(cfg) == _plain(cases.vocc_head_cfg()), 'cases.vocc_head_cfg() drifted from vocc.py'
    arrays = {}
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)

    # ---- C3: vocc.py as written (15x15x4 -> 120x120x35x16), forward + occupancy backward
    c = copy.deepcopy(cfg)
    c.pop('type')
    head = head_mod.VoxelFormerOccupancyHead(**c).eval()
    syn.load_seeded(head, 7)
    sd = head.state_dict()
    arrays['sd_names'] = np.array(list(sd.keys()))
    arrays['sd_shapes'] = np.array([','.join(str(d) for d in v.shape) for v in sd.values()])
    print('  head params: %.3f M, %d state-dict entries'
          % (sum(p.numel() for p in head.parameters()) / 1e6, len(sd)))
    with _CameraDir(w2p, org) as metas:
        for b in range(2):
            head.zero_grad()
            mlvl = T(feats[b]).unsqueeze(1)
            outs = head(mlvl, metas[b])
            occ = outs['occupancy_preds']
            key = 'c3_b%d_' % b
            arrays[key + 'occ'] = occ[0, ::997].detach().numpy()
            arrays[key + 'occ_norm'] = np.float64(occ.detach().double().norm())
            arrays[key + 'occ_mean'] = np.float64(occ.detach().double().mean())
            arrays[key + 'bev'] = outs['bev_embed'][::7, 0].detach().numpy()
            arrays[key + 'cls'] = outs['all_cls_scores'].detach().numpy()
            arrays[key + 'bbox'] = outs['all_bbox_preds'].detach().numpy()
            print('  c3 vp%d occ norm %.3f mean %.4f' % (b, arrays[key + 'occ_norm'],
                                                      arrays[key + 'occ_mean']))
            if b == 0:
                g = T(np.random.default_rng(60).standard_normal((504000, 16)).astype(np.float32))
                (occ[0] * g).sum().backward()
                names, norms = [], []
                for k, p in sorted(head.named_parameters()):
                    if p.grad is not None and p.requires_grad:
                        names.append(k)
                        norms.append(float(p.grad.double().norm()))
                arrays['c3_grad_names'] = np.array(names)
                arrays['c3_grad_norms'] = np.array(norms, dtype=np.float64)
                arrays['c3_grad_voxel_embedding'] = head.voxel_embedding.weight.grad[::9].numpy()
                arrays['c3_grad_up0'] = head.up_sample[0].weight.grad[::37, ::41].numpy()
                arrays['c3_grad_occ_proj'] = head.occ_proj.weight.grad[::53, ::29].numpy()

    # ---- C2: single-scale 50x50x16 (refine_occ=False), forward only
    c = copy.deepcopy(cases.vocc_head_cfg(bev=(16, 50, 50), refine_occ=False))
    c.pop('type')
    head = head_mod.VoxelFormerOccupancyHead(**c).eval()
    syn.load_seeded(head, 8)
    with _CameraDir(w2p, org) as metas, torch.no_grad():
        outs = head(T(feats[0]).unsqueeze(1), metas[0])
        occ = outs['occupancy_preds']
        assert occ.shape == (1, 50 * 50 * 35, 16)
        arrays['c2_b0_occ'] = occ[0, ::173].numpy()
        arrays['c2_b0_occ_norm'] = np.float64(occ.double().norm())
        arrays['c2_b0_bev'] = outs['bev_embed'][::97, 0].numpy()
        print('  c2 vp0 occ norm %.3f' % arrays['c2_b0_occ_norm'])
    save('head_vocc', **arrays)


def gen_loss(ref):
    """next-row 2: the reference head's own ``loss_single`` (head:903-990: Hungarian targets via
    its HungarianAssigner3D, focal / L1 / occupancy losses) on the stored decoder outputs of the
    vocc.py head (viewpoint 0, last decoder layer) and synthetic ground truth."""
    from make_golden import save
    head_mod = install_head_stubs()
    cfg, train_cfg = reference_head_cfg()
    assert _plain(train_cfg['pts']) == _plain(cases.VOCC_TRAIN_CFG), 'cases.VOCC_TRAIN_CFG drifted from vocc.py'
    c = copy.deepcopy(cfg)
    c.pop('type')
    # a light head is enough for the losses: same classes / ranges, tiny transformer is not possible
    # (768 is hard-wired), so build the real one once (no forward needed)
    head = head_mod.VoxelFormerOccupancyHead(train_cfg=train_cfg['pts'], **c)
    g = np.load(os.path.join(HERE, 'head_vocc.npz'))
    cls = T(g['c3_b0_cls'][-1]).clone().requires_grad_(True)            # [1,100,17]
    box = T(g['c3_b0_bbox'][-1]).clone().requires_grad_(True)           # [1,100,10]
    boxes, labels = cases.detection_gt()
    logits, gt_occ = cases.occupancy_loss_inputs()
    occ = T(logits).requires_grad_(True)
    lc, lb, lo, lf = head.loss_single(cls, box, occ, None, [T(boxes)], T(labels), None, T(gt_occ), None)
    (lc + lb + lo).backward()
    res = head.assigner.assign(box[0].detach(), cls[0].detach(), T(boxes), T(labels), None)
    print('  loss_cls %.6f loss_bbox %.6f loss_occ %.6f; matched queries %s'
          % (float(lc), float(lb), float(lo), torch.nonzero(res.gt_inds > 0).squeeze(-1).tolist()))
    save('loss_vocc', loss_cls=np.float64(lc.detach()), loss_bbox=np.float64(lb.detach()),
         loss_occ=np.float64(lo.detach()), gt_inds=res.gt_inds.numpy(), assigned_labels=res.labels.numpy(),
         grad_cls=cls.grad.numpy(), grad_box=box.grad.numpy(), grad_occ=occ.grad.numpy())


def gen_post(ref):
    """next-rows 3/4: the reference's get_occupancy_prediction (head:1505-1540) and SSCMetrics
    (datasets/occupancy_metrics.py, imported directly: pure numpy)."""
    from make_golden import save
    import importlib.util
    head_mod = install_head_stubs()
    logits, gt = cases.occupancy_loss_inputs(seed=33, n=6000)
    obj = head_mod.VoxelFormerOccupancyHead.__new__(head_mod.VoxelFormerOccupancyHead)
    obj.occ_loss_type, obj.occupancy_classes, obj.flow_gt_dimension = 'focal_loss', 16, 2
    res = head_mod.VoxelFormerOccupancyHead.get_occupancy_prediction(
        obj, dict(occupancy_preds=T(logits)[None], flow_preds=None))
    sparse = res['occupancy_preds'].numpy()
    spec = importlib.util.spec_from_file_location(
        'ref_occ_metrics', stub.REFERENCE_ROOT + '/projects/mmdet3d_plugin/datasets/occupancy_metrics.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    m = mod.SSCMetrics(n_classes=17)
    dense = np.full(6000, 16, dtype=np.int64)
    dense[sparse[:, 0]] = sparse[:, 1]
    m.add_batch(dense, gt)
    m.add_batch(dense[::-1].copy(), gt)
    st = m.get_stats()
    print('  sparse voxels %d, iou %.3f miou %.3f' % (len(sparse), st['iou'], st['miou']))
    save('post_vocc', sparse=sparse, hist=m.hist, iou=np.float64(st['iou']), precision=np.float64(st['precision']),
         recall=np.float64(st['recall']), iou_ssc=st['iou_ssc'], miou=np.float64(st['miou']))


class _CudaIsHere:
    """The layout loss of the reference calls ``.cuda()`` on freshly made index tensors (head:786-787, :1175); this
    container has no GPU.  For the duration of the golden run ``Tensor.cuda`` returns the tensor itself -- the
    arithmetic is untouched."""

    def __enter__(self):
        self.old = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t

    def __exit__(self, *a):
        torch.Tensor.cuda = self.old


class _Boxes:
    """Stand-in for an mmdet3d box object: the two attributes head:1181-1186 read."""

    def __init__(self, arr):
        self.tensor = T(arr)
        self.gravity_center = self.tensor[:, :3]


def gen_layout(ref):
    """Room-layout branch (add_layout=True; head:436-532 forward, :760-902 targets, :992-1248 losses, layout_coder.py):
    the reference head built from vocc.py + add_layout + a loss_layout entry, same seeded weights as the C3 case."""
    from make_golden import _CameraDir, save
    head_mod = install_head_stubs()
    cfg, train_cfg = reference_head_cfg()
    c = copy.deepcopy(cfg)
    c.pop('type')
    head = head_mod.VoxelFormerOccupancyHead(train_cfg=train_cfg['pts'], add_layout=True,
                                             loss_layout=dict(cases.LAYOUT_LOSS_CFG), **c).eval()
    code_weights = head.code_weights.detach().clone()
    syn.load_seeded(head, 7)
    head.code_weights.data.copy_(code_weights)          # (the seeded fill also hits this non-trainable vector: keep the default)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    g = np.load(os.path.join(HERE, 'head_vocc.npz'))
    arrays = {}
    with _CameraDir(w2p, org) as metas:
        outs = head(T(feats[0]).unsqueeze(1), metas[0])
    occ = outs['occupancy_preds']
    assert occ.shape == (1, 35 * 15 * 15, 16) and outs['all_layout_preds'].shape == (6, 1, 100, 10)
    # same weights, same decoder: the detection outputs equal the C3 case's
    assert float((outs['all_cls_scores'].detach() - T(g['c3_b0_cls'])).abs().max()) < 1e-5
    arrays['layout_preds'] = outs['all_layout_preds'].detach().numpy()
    arrays['occ'] = occ[0, ::5].detach().numpy()
    arrays['occ_norm'] = np.float64(occ.detach().double().norm())
    boxes, labels = cases.detection_gt()
    lay = cases.layout_gt()
    rng = np.random.default_rng(34)
    nocc = 35 * 15 * 15
    sparse_idx = np.sort(rng.choice(nocc, 2000, replace=False))
    sparse = np.stack([sparse_idx, rng.integers(0, 16, 2000)], 1).astype(np.int64)
    gt_occ = np.full(nocc, 16, dtype=np.int64)
    gt_occ[sparse[:, 0]] = sparse[:, 1]
    arrays['occ_sparse'] = sparse
    # ---- one layer (the last), with gradients
    cls = outs['all_cls_scores'][-1].detach().clone().requires_grad_(True)
    box = outs['all_bbox_preds'][-1].detach().clone().requires_grad_(True)
    lp = outs['all_layout_preds'][-1].detach().clone().requires_grad_(True)
    oc = occ.detach().reshape(-1, 16).clone().requires_grad_(True)
    gtb = torch.cat([T(boxes)[:, :7], torch.zeros(len(boxes), 2)], 1)
    gtl = torch.cat([T(lay), torch.zeros(1, 2)], 1)
    with _CudaIsHere():
        lc, lb, ll, lo, lf = head.loss_single_layout(cls, box, lp, oc, None, [gtb], T(labels), [gtl], None,
                                                     T(gt_occ), None)
        (lc + lb + ll + lo).backward()
        res = head.assigner.assign(lp[0].detach(), None, gtl, torch.zeros(1).long(), None, layout=True)
    arrays.update(loss_cls=np.float64(lc.detach()), loss_bbox=np.float64(lb.detach()), loss_layout=np.float64(ll.detach()),
                  loss_occ=np.float64(lo.detach()), layout_gt_inds=res.gt_inds.numpy(), grad_layout=lp.grad.numpy(),
                  grad_cls=cls.grad.numpy(), grad_box=box.grad.numpy(), grad_occ=oc.grad[::5].numpy())
    print('  layout: loss_cls %.6f bbox %.6f layout %.6f occ %.6f; layout query %s'
          % (float(lc), float(lb), float(ll), float(lo), torch.nonzero(res.gt_inds > 0).squeeze(-1).tolist()))
    # ---- the whole dict through loss_addlayout (bs = 1, box objects, sparse occupancy ground truth).  It sizes the
    # dense target by head.voxel_num (= 504000, the refined grid), which this branch's coarse occupancy output never
    # has: the attribute is set to the output's own size for the call
    head.voxel_num = nocc
    with _CudaIsHere():
        d = head.loss_addlayout(_Boxes(boxes[:, :7]), labels, _Boxes(lay), None, [[sparse]], None,
                                {k: (v.detach() if torch.is_tensor(v) else v) for k, v in outs.items()})
    arrays['dict_keys'] = np.array(sorted(d))
    arrays['dict_vals'] = np.array([float(d[k]) for k in sorted(d)], dtype=np.float64)
    # ---- decoding (layout_coder.py)
    with _CudaIsHere():
        dec = head.layout_coder.decode({'all_layout_preds': outs['all_layout_preds'].detach()})
    arrays['decoded'] = dec[0]['layouts'].numpy()
    print('  layout: %d boxes decoded inside the range' % len(arrays['decoded']))
    save('layout_vocc', **arrays)


def gen_init(ref):
    """a11: the reference's ``init_weights`` chain (head:269-279 -> voxel_transformer.py:99-116 ->
    spatial_cross_attention.py:255-273) on the head built from vocc.py: the deterministic results in full (offset-bias
    ring, zeroed projections, focal-prior biases), the random ones by their statistics, and WHICH entries the chain
    touches at all."""
    from make_golden import save
    sys.path.insert(0, os.path.dirname(HERE))
    from util import init_report                    # (shared with the test that runs it on OUR head)
    head_mod = install_head_stubs()
    cfg, _ = reference_head_cfg()
    c = copy.deepcopy(cfg)
    c.pop('type')
    torch.manual_seed(0)
    head = head_mod.VoxelFormerOccupancyHead(**c)
    rep = init_report(head)
    sd = head.state_dict()
    pre = 'transformer.encoder.layers.0.attentions.0.deformable_attention.'
    arrays = {'rep_' + k: v for k, v in rep.items()}
    arrays['enc_offset_bias'] = sd[pre + 'sampling_offsets.bias'].numpy()
    dec = 'transformer.decoder.layers.0.attentions.1.'
    arrays['dec_offset_bias'] = sd[dec + 'sampling_offsets.bias'].numpy()      # construction-time ring (3-D); the chain skips it
    arrays['cls_prior_bias'] = sd['cls_branches.5.6.bias'].numpy()
    arrays['occ_prior_bias'] = sd['occ_branches.6.bias'].numpy()
    n_changed = int(rep['changed'].sum())
    print('  init: %d of %d float entries re-initialised; cls prior %.5f' % (n_changed, len(rep['names']),
                                                                         float(arrays['cls_prior_bias'][0])))
    save('init_vocc', **arrays)


GENERATORS = {'head': gen_head, 'loss': gen_loss, 'post': gen_post, 'layout': gen_layout, 'init': gen_init}
