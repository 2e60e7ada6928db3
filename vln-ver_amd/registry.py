"""mmcv-style registries for the plugin API of the lifting path.

The reference creates every hot-path class with ``build_from_cfg(cfg, REGISTRY)`` from the
``type=`` strings of projects/configs/verformer/vocc.py (registrations:
voxel_encoder.py:30,299; spatial_cross_attention.py:31,179; voxel_transformer.py:24;
voxel_positional_embedding.py:10; voxelformer_occupancy_head.py:31).  When mmcv / mmdet are
importable our classes are registered into THEIR registries (``force=True``), which is what
makes them drop into an unmodified vocc.py; otherwise the minimal registries below carry the
same names and the same ``build_from_cfg`` contract.
"""
import copy
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def _register(self, cls, name=None, force=False):
        key = name or cls.__name__
        if not force and key in self._module_dict:
            raise KeyError('%s is already registered in %s' % (key, self._name))
        self._module_dict[key] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


class ConfigDict(dict):
    """dict with attribute access (the subset of mmcv.ConfigDict the path relies on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def build_from_cfg(cfg, registry, default_args=None):
    """Same contract as mmcv.utils.build_from_cfg: ``cfg['type']`` names a registered class
    (or is the class), remaining keys are constructor kwargs, errors name the class."""
    if not isinstance(cfg, dict):
        raise TypeError('cfg must be a dict, but got %s' % type(cfg))
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError('`cfg` or `default_args` must contain the key "type", but got %s' % cfg)
    args = copy.copy(dict(cfg))
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop('type')
    if isinstance(typ, str):
        cls = registry.get(typ)
        if cls is None:
            raise KeyError('%s is not in the %s registry' % (typ, registry.name))
    elif inspect.isclass(typ):
        cls = typ
    else:
        raise TypeError('type must be a str or valid type, but got %s' % type(typ))
    try:
        return cls(**args)
    except Exception as e:
        raise type(e)('%s: %s' % (cls.__name__, e))


def _try_mmcv():
    try:
        from mmcv.cnn.bricks import registry as r           # noqa: F401
        return r
    except Exception:
        return None


_mm = _try_mmcv()
if _mm is not None:                                             # pragma: no cover (no mmcv here)
    ATTENTION = _mm.ATTENTION
    FEEDFORWARD_NETWORK = _mm.FEEDFORWARD_NETWORK
    POSITIONAL_ENCODING = _mm.POSITIONAL_ENCODING
    TRANSFORMER_LAYER = _mm.TRANSFORMER_LAYER
    TRANSFORMER_LAYER_SEQUENCE = _mm.TRANSFORMER_LAYER_SEQUENCE
    USING_MMCV = True
else:
    ATTENTION = Registry('attention')
    FEEDFORWARD_NETWORK = Registry('feed-forward network')
    POSITIONAL_ENCODING = Registry('position encoding')
    TRANSFORMER_LAYER = Registry('transformerLayer')
    TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
    USING_MMCV = False

try:                                                            # pragma: no cover
    from mmdet.models.utils.builder import TRANSFORMER
    from mmdet.models.builder import HEADS
except Exception:
    TRANSFORMER = Registry('Transformer')
    HEADS = Registry('head')

try:                                                            # pragma: no cover
    from mmdet.models.builder import DETECTORS
except Exception:
    DETECTORS = Registry('detector')


try:                                                            # pragma: no cover
    from mmdet.models.builder import LOSSES
    from mmdet.core.bbox.builder import BBOX_CODERS, BBOX_ASSIGNERS
    from mmdet.core.bbox.match_costs.builder import MATCH_COST
except Exception:
    LOSSES = Registry('loss')
    BBOX_CODERS = Registry('bbox coder')
    BBOX_ASSIGNERS = Registry('bbox assigner')
    MATCH_COST = Registry('match cost')


def build_loss(cfg, default_args=None):
    return build_from_cfg(cfg, LOSSES, default_args)


def build_bbox_coder(cfg, default_args=None):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


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


def build_head(cfg, default_args=None):
    return build_from_cfg(cfg, HEADS, default_args)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d.models.build_model / build_detector: ``cfg['type']`` names a registered detector; ``train_cfg`` /
    ``test_cfg`` may be given here or inside ``cfg`` (not both)."""
    if train_cfg is not None or test_cfg is not None:
        assert cfg.get('train_cfg') is None or train_cfg is None, 'train_cfg specified in both outer field and model field'
        assert cfg.get('test_cfg') is None or test_cfg is None, 'test_cfg specified in both outer field and model field'
    return build_from_cfg(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
