"""Product-side loader of mmcv-style python config files (``projects/configs/verformer/vocc.py``).

The reference reads its config with ``mmcv.Config.fromfile`` (tools/train.py:105-135), then
``build_model(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))``; the
detector hands ``train_cfg.pts`` / ``test_cfg.pts`` to ``pts_bbox_head`` (mmdet3d
``MVXTwoStageDetector.__init__``).  mmcv is not a dependency of this package, so the subset of
that behaviour the lifting path needs is restated here:

* a config is a python file executed in an empty namespace; every public, non-module name is a key;
* ``_base_`` names files (relative to the config) that are loaded first and merged underneath,
  dict by dict; a base that does not exist is skipped and recorded in ``cfg['_missing_bases_']``
  (vocc.py's bases are dataset / runtime files the lifting path does not read -- and the reference's
  own tree ships without them);
* ``_delete_=True`` inside an overriding dict replaces instead of merging.

``load_model_cfg(path)`` returns ``cfg['model']``; ``head_cfg(model)`` is the dict the detector would
build the head from.  ``VOCC`` is the config shipped with this package
(``vln-ver_amd/configs/vocc_lifting.py``), whose ``model['pts_bbox_head']`` / ``train_cfg`` equal the
reference file's (asserted against the file itself in ``tests/test_config_cpu.py`` when the
reference tree is present).
"""
import copy
import os
import types

from .registry import ConfigDict

HERE = os.path.dirname(os.path.abspath(__file__))
VOCC = os.path.join(HERE, 'configs', 'vocc_lifting.py')


def _to_configdict(obj):
    if isinstance(obj, dict):
        return ConfigDict((k, _to_configdict(v)) for k, v in obj.items())
    if isinstance(obj, list):
        return [_to_configdict(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_to_configdict(v) for v in obj)
    return obj


def _merge(base, over):
    """mmcv.Config._merge_a_into_b for dict values: ``over`` wins, dicts merge recursively."""
    out = copy.deepcopy(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict) and '_delete_' in v:
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = copy.deepcopy(v)
    return out


def _exec_file(path):
    with open(path) as f:
        src = f.read()
    ns = {'__file__': path}
    exec(compile(src, path, 'exec'), ns)
    return {k: v for k, v in ns.items()
            if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}


def load_cfg(path):
    """Whole config as a ConfigDict (bases merged underneath when they exist)."""
    path = os.path.abspath(path)
    if not os.path.isfile(path):
        raise FileNotFoundError('config file %s does not exist' % path)
    own = _exec_file(path)
    bases = own.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged, missing = {}, []
    for b in bases:
        bp = os.path.normpath(os.path.join(os.path.dirname(path), b))
        if not os.path.isfile(bp):
            missing.append(b)
            continue
        sub = dict(load_cfg(bp))
        missing += sub.pop('_missing_bases_', [])
        dup = set(merged) & set(sub)
        if dup:
            raise KeyError('duplicate key in base configs: %s' % sorted(dup))
        merged.update(sub)
    cfg = _merge(merged, own)
    cfg['_missing_bases_'] = missing
    return _to_configdict(cfg)


def load_model_cfg(path=None):
    """``cfg.model`` of a config file (default: the vocc config shipped with the package)."""
    cfg = load_cfg(path or VOCC)
    if 'model' not in cfg:
        raise KeyError('%s defines no `model`' % (path or VOCC))
    return cfg['model']


def head_cfg(model, train=True, **overrides):
    """The dict ``pts_bbox_head`` is built from: the config's own entry plus ``train_cfg`` /
    ``test_cfg`` = the ``pts`` part of the model's, as the detector passes them on.  ``train=False``
    leaves the assigner out (inference / lifting-only use).  ``overrides`` replace top-level keys
    (e.g. ``bev_h=50`` for the single-scale parity case)."""
    head = copy.deepcopy(dict(model['pts_bbox_head']))
    tc = model.get('train_cfg')
    if train and tc and tc.get('pts') is not None:
        head['train_cfg'] = copy.deepcopy(tc['pts'])
    tc = model.get('test_cfg')
    if tc and tc.get('pts') is not None:
        head['test_cfg'] = copy.deepcopy(tc['pts'])
    head.update(overrides)
    return _to_configdict(head)


def plain(obj):
    """ConfigDict tree -> plain dicts / lists (for equality checks and JSON)."""
    if isinstance(obj, dict):
        return {k: plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [plain(v) for v in obj]
    return obj
