"""Throw-away stand-in for the mmcv / mmdet symbols the reference hot path imports.

USED ONLY BY ``make_golden.py`` IN THE BUILD CONTAINER (where /root/reference
exists). Nothing under tests/, bench.py or the package imports this at run time
on the GPU box; the committed ``*.npz`` vectors are what travels.

mmcv-full 1.4.0 / mmdet 2.14.0 are pinned by the reference (docs/install.md:15-41)
but are not vendored under /root/reference, so their behaviour is restated here
from the published semantics of those versions.  The one
non-trivial function, ``multi_scale_deformable_attn_pytorch``, is additionally
cross-checked by make_golden.py against the reference's in-tree 3-D twin
(bevformer/modules/voxel_temporal_self_attention.py:275-335) at depth 1.
"""
import copy
import functools
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = '/root/reference'


# ----------------------------------------------------------------------------- registry
class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **kw):
        return build_from_cfg(cfg, self, kw or None)


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop('type')
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError('%s is not in the %s registry' % (typ, registry.name))
    return cls(**args)


ATTENTION = Registry('attention')
FEEDFORWARD_NETWORK = Registry('feed-forward network')
POSITIONAL_ENCODING = Registry('position encoding')
TRANSFORMER_LAYER = Registry('transformerLayer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
TRANSFORMER = Registry('Transformer')
HEADS = Registry('head')
LOSSES = Registry('loss')
BBOX_CODERS = Registry('bbox coder')
BBOX_ASSIGNERS = Registry('bbox assigner')
MATCH_COST = Registry('match cost')


def build_attention(cfg, default_args=None):
    return build_from_cfg(cfg, ATTENTION, default_args)


def build_feedforward_network(cfg, default_args=None):
    return build_from_cfg(cfg, FEEDFORWARD_NETWORK, default_args)


def build_transformer_layer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER, default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER_SEQUENCE, default_args)


def build_positional_encoding(cfg, default_args=None):
    return build_from_cfg(cfg, POSITIONAL_ENCODING, default_args)


def build_transformer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER, default_args)


# ----------------------------------------------------------------------------- runner bits
class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()
        self._is_init = True


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def _identity_decorator(*dargs, **dkwargs):
    # force_fp32 / auto_fp16 are no-ops unless module.fp16_enabled (never, B.7)
    def deco(fn):
        return fn
    return deco


def deprecated_api_warning(name_dict, cls_name=None):
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            for old, new in name_dict.items():
                if old in kwargs:
                    kwargs[new] = kwargs.pop(old)
            return fn(*args, **kwargs)
        return wrapper
    return deco


# ----------------------------------------------------------------------------- cnn bits
def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    import math
    return float(-math.log((1 - prior_prob) / prior_prob))


def build_norm_layer(cfg, num_features, postfix=''):
    assert cfg['type'] == 'LN'
    return 'ln' + str(postfix), nn.LayerNorm(num_features)


def build_activation_layer(cfg):
    cfg = dict(cfg)
    typ = cfg.pop('type')
    return {'ReLU': nn.ReLU, 'GELU': nn.GELU}[typ](**cfg)


class FFN(BaseModule):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.embed_dims = embed_dims
        layers = []
        in_channels = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(Sequential(nn.Linear(in_channels, feedforward_channels),
                                     build_activation_layer(act_cfg), nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = Sequential(*layers)
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        if identity is None:
            identity = x
        return identity + out


FEEDFORWARD_NETWORK.register_module()(FFN)


class TransformerLayerSequence(BaseModule):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        self.num_layers = num_layers
        self.layers = ModuleList()
        for i in range(num_layers):
            self.layers.append(build_transformer_layer(transformerlayers[i]))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


class MultiheadAttention(BaseModule):
    """mmcv 1.4.0 wrapper over nn.MultiheadAttention."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=dict(type='Dropout', drop_prob=0.), init_cfg=None,
                 batch_first=False, **kwargs):
        super().__init__(init_cfg)
        if 'dropout' in kwargs:
            attn_drop = kwargs['dropout']
            dropout_layer = dict(type='Dropout', drop_prob=kwargs.pop('dropout'))
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Dropout(dropout_layer['drop_prob']) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = [t.transpose(0, 1) for t in (query, key, value)]
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


ATTENTION.register_module()(MultiheadAttention)


class BaseTransformerLayer(BaseModule):
    """mmcv 1.4.0 BaseTransformerLayer; the reference keeps an in-tree copy at
    bevformer/modules/custom_base_transformer_layer.py:38-260 whose construction
    logic this follows (batch_first defaults to False upstream)."""

    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024, num_fcs=2,
                               ffn_drop=0., act_cfg=dict(type='ReLU', inplace=True)),
                 operation_order=None, norm_cfg=dict(type='LN'), init_cfg=None,
                 batch_first=False, **kwargs):
        ffn_cfgs = copy.deepcopy(ffn_cfgs)
        for ori, new in dict(feedforward_channels='feedforward_channels', ffn_dropout='ffn_drop',
                             ffn_num_fcs='num_fcs').items():
            if ori in kwargs:
                ffn_cfgs[new] = kwargs[ori]
        super().__init__(init_cfg)
        self.batch_first = batch_first
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = ModuleList()
        index = 0
        for name in operation_order:
            if name in ('self_attn', 'cross_attn'):
                cfg = copy.deepcopy(attn_cfgs[index])
                cfg.setdefault('batch_first', self.batch_first)
                att = build_attention(cfg)
                att.operation_name = name
                self.attentions.append(att)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = ModuleList()
        num_ffns = operation_order.count('ffn')
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [copy.deepcopy(ffn_cfgs) for _ in range(num_ffns)]
        for i in range(num_ffns):
            ffn_cfgs[i].setdefault('embed_dims', self.embed_dims)
            self.ffns.append(build_feedforward_network(ffn_cfgs[i]))
        self.norms = ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(build_norm_layer(norm_cfg, self.embed_dims)[1])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None] * self.num_attn
        for layer in self.operation_order:
            if layer == 'self_attn':
                query = self.attentions[attn_index](
                    query, query, query, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=query_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=query_key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'norm':
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=key_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'ffn':
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


class DetrTransformerDecoderLayer(BaseTransformerLayer):
    """mmdet 2.14.0: BaseTransformerLayer that asserts the 6-op DETR order (B.9)."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2,
                 **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                         norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6


TRANSFORMER_LAYER.register_module()(BaseTransformerLayer)
TRANSFORMER_LAYER.register_module()(DetrTransformerDecoderLayer)


# ----------------------------------------------------------------------------- the op
def multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations,
                                        attention_weights):
    """Published mmcv CPU path: per level grid_sample(bilinear, zeros,
    align_corners=False) on grid = 2*loc-1, weighted sum over level*point."""
    bs, _, heads, hd = value.shape
    _, nq, _, nl, npt, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in value_spatial_shapes]
    per_level = value.split(sizes, dim=1)
    grid = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(value_spatial_shapes):
        v = per_level[lvl].flatten(2).transpose(1, 2).reshape(bs * heads, hd, int(h), int(w))
        g = grid[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros',
                                     align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(bs * heads, 1, nq, nl * npt)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(bs, heads * hd, nq)
    return out.transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------- install
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


class _ExtStub:
    def __getattr__(self, k):
        def missing(*a, **kw):
            raise RuntimeError('mmcv _ext.%s is CUDA-only and absent here' % k)
        return missing


_INSTALLED = False


def install():
    """Insert the fake modules and path-only ``projects`` packages (idempotent)."""
    global _INSTALLED
    if _INSTALLED:
        return sys.modules['projects.mmdet3d_plugin.bevformer.modules']
    _INSTALLED = True
    sys.dont_write_bytecode = True
    ext_loader = types.SimpleNamespace(load_ext=lambda name, funcs: _ExtStub())

    def digit_version(v):
        return tuple(int(x) for x in str(v).split('+')[0].split('.')[:3] if x.isdigit())

    common = dict(ConfigDict=ConfigDict, build_from_cfg=build_from_cfg,
                  deprecated_api_warning=deprecated_api_warning, Registry=Registry)
    mmcv = _mod('mmcv', **common)
    mmcv.__path__ = []
    _mod('mmcv.utils', ext_loader=ext_loader, TORCH_VERSION=torch.__version__,
         digit_version=digit_version, to_2tuple=lambda x: (x, x), **common)
    cnn = _mod('mmcv.cnn', xavier_init=xavier_init, constant_init=constant_init,
               bias_init_with_prob=bias_init_with_prob, Linear=nn.Linear,
               build_activation_layer=build_activation_layer, build_norm_layer=build_norm_layer)
    cnn.__path__ = []
    bricks = _mod('mmcv.cnn.bricks')
    bricks.__path__ = []
    _mod('mmcv.cnn.bricks.registry', ATTENTION=ATTENTION, FEEDFORWARD_NETWORK=FEEDFORWARD_NETWORK,
         POSITIONAL_ENCODING=POSITIONAL_ENCODING, TRANSFORMER_LAYER=TRANSFORMER_LAYER,
         TRANSFORMER_LAYER_SEQUENCE=TRANSFORMER_LAYER_SEQUENCE)
    _mod('mmcv.cnn.bricks.transformer', build_attention=build_attention,
         build_feedforward_network=build_feedforward_network,
         build_transformer_layer=build_transformer_layer,
         build_transformer_layer_sequence=build_transformer_layer_sequence,
         build_positional_encoding=build_positional_encoding,
         TransformerLayerSequence=TransformerLayerSequence, FFN=FFN,
         BaseTransformerLayer=BaseTransformerLayer, MultiheadAttention=MultiheadAttention,
         POSITIONAL_ENCODING=POSITIONAL_ENCODING, ATTENTION=ATTENTION,
         TRANSFORMER_LAYER=TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE=TRANSFORMER_LAYER_SEQUENCE)
    runner = _mod('mmcv.runner', force_fp32=_identity_decorator, auto_fp16=_identity_decorator,
                  BaseModule=BaseModule)
    runner.__path__ = []
    _mod('mmcv.runner.base_module', BaseModule=BaseModule, ModuleList=ModuleList,
         Sequential=Sequential)
    ops = _mod('mmcv.ops')
    ops.__path__ = []
    _mod('mmcv.ops.multi_scale_deform_attn',
         multi_scale_deformable_attn_pytorch=multi_scale_deformable_attn_pytorch)
    for name in ('mmdet', 'mmdet.models', 'mmdet.models.utils'):
        _mod(name).__path__ = []
    _mod('mmdet.models.utils.builder', TRANSFORMER=TRANSFORMER)
    _mod('mmdet.models.utils.transformer', inverse_sigmoid=None)
    _mod('cv2')
    tv = _mod('torchvision')
    tv.__path__ = []
    _mod('torchvision.transforms').__path__ = []
    _mod('torchvision.transforms.functional', rotate=None)

    root = REFERENCE_ROOT + '/projects'
    _pkg('projects', root)
    _pkg('projects.mmdet3d_plugin', root + '/mmdet3d_plugin')
    _pkg('projects.mmdet3d_plugin.bevformer', root + '/mmdet3d_plugin/bevformer')
    _pkg('projects.mmdet3d_plugin.bevformer.modules', root + '/mmdet3d_plugin/bevformer/modules')
    _pkg('projects.mmdet3d_plugin.models', root + '/mmdet3d_plugin/models')
    _pkg('projects.mmdet3d_plugin.models.utils', root + '/mmdet3d_plugin/models/utils')
    import importlib
    br = importlib.import_module('projects.mmdet3d_plugin.models.utils.bricks')
    sys.modules['projects.mmdet3d_plugin.models.utils'].run_time = br.run_time
    _mod('projects.mmdet3d_plugin.models.utils.visual', save_tensor=lambda *a, **k: None)

    class CustomMSDeformableAttention(BaseModule):   # never-shipped decoder.py symbol
        pass
    _mod('projects.mmdet3d_plugin.bevformer.modules.decoder',
         CustomMSDeformableAttention=CustomMSDeformableAttention)
    return sys.modules['projects.mmdet3d_plugin.bevformer.modules']


def ref_modules():
    """Import the reference hot-path files (file by file; package __init__s never run)."""
    import importlib
    install()
    base = 'projects.mmdet3d_plugin.bevformer.modules.'
    names = ['multi_scale_deformable_attn_function', 'spatial_cross_attention',
             'custom_base_transformer_layer', 'voxel_encoder', 'voxel_positional_embedding',
             'voxel_temporal_self_attention', 'voxel_decoder', 'voxel_transformer']
    return {n: importlib.import_module(base + n) for n in names}
