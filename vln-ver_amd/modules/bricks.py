"""The few mmcv building blocks the lifting path constructs (restated from the published
behaviour of mmcv-full 1.4.0; mmcv is not vendored by the
reference).  Parameter names are the compatibility surface: ``layers.0.0.*``, ``layers.1.*``."""
import copy
import os

import torch
import torch.nn as nn

from ..registry import FEEDFORWARD_NETWORK, USING_MMCV, build_transformer_layer


class BaseModule(nn.Module):
    """nn.Module carrying ``init_cfg`` (mmcv.runner.BaseModule)."""

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


def build_norm_layer(cfg, num_features):
    """-> (name, layer); only LayerNorm is used on this path
    (custom_base_transformer_layer.py:163 with norm_cfg=dict(type='LN'))."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    if typ != 'LN':
        raise KeyError('norm layer %s not available on the lifting path' % typ)
    cfg.pop('requires_grad', None)
    return 'ln', nn.LayerNorm(num_features, **cfg)


def build_activation_layer(cfg):
    cfg = dict(cfg)
    typ = cfg.pop('type')
    table = {'ReLU': nn.ReLU, 'GELU': nn.GELU, 'LeakyReLU': nn.LeakyReLU}
    if typ not in table:
        raise KeyError('activation %s not available' % typ)
    return table[typ](**cfg)


class FFN(BaseModule):
    """Linear -> act -> drop (x num_fcs-1) -> Linear -> drop, plus identity."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2, 'num_fcs should be no less than 2. got %s.' % num_fcs
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        layers = []
        in_channels = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(Sequential(nn.Linear(in_channels, feedforward_channels),
                                     build_activation_layer(act_cfg), nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = Sequential(*layers)
        drop = (dropout_layer or {}).get('drop_prob', 0.) if dropout_layer else 0.
        self.dropout_layer = nn.Dropout(drop) if drop else nn.Identity()
        self.add_identity = add_identity

    def _hidden(self, block, x):
        """One hidden block ``Sequential(Linear, ReLU, Dropout)``: ReLU + Dropout as one HIP pass on the GPU."""
        if (_FUSED_ADD_LN and isinstance(block, Sequential) and len(block) == 3 and isinstance(block[0], nn.Linear)
                and isinstance(block[1], nn.ReLU) and isinstance(block[2], nn.Dropout) and x.is_cuda):
            h = tall_linear(block[0], x)
            if h.numel() % 4 == 0 and h.dtype in (torch.float32, torch.bfloat16):
                from ..hipops import relu_dropout
                return relu_dropout(h, block[2].p if block[2].training else 0.0)
            return block[2](block[1](h))
        return block(x)

    def forward(self, x, identity=None, defer_residual=False):
        """``defer_residual``: return ``PendingResidual(out, identity, p)`` instead of ``identity + dropout(out)`` when
        the caller follows up with a LayerNorm (``residual_layer_norm`` does all three in one pass)."""
        x_in = lowp_view(x)
        last = self.layers[len(self.layers) - 1]
        if (defer_residual and self.add_identity and isinstance(self.dropout_layer, nn.Identity)
                and isinstance(last, nn.Dropout)):
            out = x_in
            for i in range(len(self.layers) - 1):
                lay = self.layers[i]
                out = tall_linear(lay, out) if isinstance(lay, nn.Linear) else self._hidden(lay, out)
            return PendingResidual(out, x if identity is None else identity, last.p if last.training else 0.0)
        out = self.layers(x_in)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


_FUSED_ADD_LN = os.environ.get('VER_FUSED_ADD_LN', '1') == '1'       # (0: plain torch ops, for A/B runs)


class PendingResidual:
    """``residual + dropout(branch)`` not formed yet: what an attention / FFN hands to the LayerNorm that follows it."""

    def __init__(self, branch, residual, p):
        self.branch, self.residual, self.p = branch, residual, float(p)

    def materialize(self):
        return self.residual + torch.nn.functional.dropout(self.branch, self.p, self.p > 0)


def _autocast_bf16():
    return torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16


_TALL_ROWS = 16000      # rows from which the weight gradient of a Linear is split (two chunks of dense_heads/row_linear.py)


def tall_linear(mod, x, weight=None, bias=None):
    """``mod(x)`` for an ``nn.Linear`` over the ~2e5 voxel / token rows of a batched encoder call.  Forward and d(input)
    are the library's GEMMs either way; the weight gradient G^T X is a [768 x rows] x [rows x 768] product that the library
    runs on the few workgroups its 9-18 output tiles give (GEMM ledger, profiles/r04_gemm_ledger.csv: 0.15-0.6 PFLOP/s) --
    ``row_linear`` splits the row axis into one batched GEMM with fp32-summed partials.  ``weight`` / ``bias``: fused
    parameter stacks (the two narrow projections of the query) instead of ``mod``'s own."""
    if weight is None and bias is None and mod is not None and 'forward' in mod.__dict__:
        return mod(x)                   # the module is lent bf16 parameters and its own bf16 path (modules/lowp_params.py)
    w = mod.weight if weight is None else weight
    b = (mod.bias if mod is not None else None) if bias is None else bias
    if x.is_cuda and torch.is_grad_enabled() and w.requires_grad and x.numel() // x.shape[-1] >= _TALL_ROWS:
        from ..dense_heads.row_linear import row_linear
        return row_linear(x, w, b)
    return torch.nn.functional.linear(x, w, b)


def lowp_view(x):
    """The bf16 copy a fused LayerNorm wrote next to its fp32 output (``residual_layer_norm``), if x carries one and
    bf16 autocast is on: what the next Linear would cast x to anyway."""
    x16 = getattr(x, '_ver_lowp', None)
    return x16 if (x16 is not None and _autocast_bf16()) else x


def residual_layer_norm(pending, norm):
    """LayerNorm(residual + dropout(branch)).  On the GPU one fused pass each way (``ver_add_ln_*``) that also writes
    the bf16 copy for the next Linear under bf16 autocast; elsewhere the plain sequence of torch ops."""
    branch, res = pending.branch, pending.residual
    c = branch.shape[-1]
    if (_FUSED_ADD_LN and branch.is_cuda and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None
            and tuple(norm.normalized_shape) == (c,) and c % 256 == 0 and c <= 1024
            and branch.shape == res.shape and res.dtype == torch.float32):
        from ..hipops import add_dropout_layer_norm
        y, y16 = add_dropout_layer_norm(branch, res, norm.weight, norm.bias, pending.p, norm.eps,
                                        want_bf16=_autocast_bf16())
        if y16 is not None:
            y._ver_lowp = y16
        return y
    return norm(pending.materialize())


if not USING_MMCV:
    FEEDFORWARD_NETWORK.register_module(module=FFN)


class TransformerLayerSequence(BaseModule):
    """``num_layers`` deep copies of one layer config (mmcv TransformerLayerSequence)."""

    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = ModuleList()
        for i in range(num_layers):
            self.layers.append(build_transformer_layer(transformerlayers[i]))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


_CONST_TENSORS = {}


def const_tensor(values, device, dtype=torch.long):
    """A small constant tensor (``spatial_shapes``, ``level_start_index`` ...) on ``device``, created once per
    (values, dtype, device): the reference builds these from Python lists on every forward -- a pageable host -> device
    copy each time, which serialises with the stream and is not allowed while a HIP graph is being captured.
    The tensor is created outside any ``inference_mode`` (a constant first built during an eval / export pass must
    still be usable as a saved tensor of a later training step) and the key holds the RESOLVED device ('cuda' and
    'cuda:0' are one entry).  The cache only ever holds the handful of shape constants of the configured model."""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = (repr(values), dtype, str(device))
    t = _CONST_TENSORS.get(key)
    if t is None:
        with torch.inference_mode(False):
            t = _CONST_TENSORS[key] = torch.tensor(values, dtype=dtype, device=device)
        _CONST_VALUES[id(t)] = (t, values)
    return t


_CONST_VALUES = {}


def host_values(t):
    """``t.tolist()`` without the device -> host round trip when ``t`` is one of the cached constants above (the modules
    below the transformer read ``spatial_shapes`` back on the host, as the reference does with ``.tolist()`` / ``int()``:
    one stream synchronisation per attention layer)."""
    hit = _CONST_VALUES.get(id(t))
    if hit is not None and hit[0] is t:
        return hit[1]
    return t.tolist()


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if module is None:
        return
    if getattr(module, 'weight', None) is not None:
        (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(
            module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)
