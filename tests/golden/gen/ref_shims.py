"""Import plumbing for tests/golden/gen/make_golden.py -- BUILD CONTAINER ONLY.

The reference model file imports torchvision / monai names that are not
installed in this image.  None of them carries arithmetic that the golden
vectors depend on beyond three trivial containers (torchvision ``MLP`` =
Linear/GELU/Dropout/Linear/Dropout, ``Permute``, ``StochasticDepth``, which is
the identity in eval mode); those are restated here from their published
semantics (torchvision >= 0.13, SURVEY.md §8c "third-party arithmetic"; StochasticDepth's train mode too, with recorded draws).
Everything else is an inert placeholder so that ``import`` succeeds.

This module is never imported by the product package, the oracle, bench.py
or the -m gpu tests; it cannot run on the GPU box (no /root/reference there).
"""
import os
import sys
import types

import torch
from torch import nn

REF_ROOT = "/root/reference"


class _MLP(nn.Sequential):
    def __init__(self, in_channels, hidden_channels, norm_layer=None, activation_layer=nn.ReLU,
                 inplace=None, bias=True, dropout=0.0):
        kw = {} if inplace is None else {"inplace": inplace}
        layers, d = [], in_channels
        for h in hidden_channels[:-1]:
            layers.append(nn.Linear(d, h, bias=bias))
            if norm_layer is not None:
                layers.append(norm_layer(h))
            layers.append(activation_layer(**kw))
            layers.append(nn.Dropout(dropout, **kw))
            d = h
        layers.append(nn.Linear(d, hidden_channels[-1], bias=bias))
        layers.append(nn.Dropout(dropout, **kw))
        super().__init__(*layers)


class _Permute(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.dims = dims

    def forward(self, x):
        return torch.permute(x, self.dims)


class _StochasticDepth(nn.Module):
    """torchvision.ops.StochasticDepth as published (torchvision >= 0.13, ops/stochastic_depth.py): identity in eval mode or at
    p = 0; in train mode, mode "row": one bernoulli(1 - p) draw per sample, divided by the survival rate, multiplies the whole
    residual branch of that sample.  The draws come from the class-level generator ``rng`` and are RECORDED in ``draws`` (one
    float[B] per call, in call order) so that a golden of the real SwinTransformerBlock in train mode carries the noise it used
    (make_golden.gen_swin_train); without a generator train mode raises, as before."""
    rng = None
    draws = None

    def __init__(self, p, mode):
        super().__init__()
        self.p, self.mode = p, mode

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        cls = type(self)
        if cls.rng is None or self.mode != "row":
            raise RuntimeError("train-mode StochasticDepth needs _StochasticDepth.rng / .draws (gen_swin_train) and mode 'row'")
        survival = 1.0 - self.p
        noise = torch.empty([x.shape[0]] + [1] * (x.ndim - 1), dtype=x.dtype).bernoulli_(survival, generator=cls.rng)
        if survival > 0.0:
            noise.div_(survival)
        cls.draws.append(noise.flatten().clone())
        return x * noise


class _Weights:
    def __init__(self, url=None, transforms=None, meta=None):
        self.url, self.transforms, self.meta = url, transforms, meta


class _WeightsEnumMeta(type):
    pass


class _WeightsEnum:
    """Stand-in for torchvision's WeightsEnum: class attributes holding _Weights
    become members with .meta / .get_state_dict."""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        for k, v in list(vars(cls).items()):
            if isinstance(v, _Weights):
                m = object.__new__(cls)
                m.value, m.meta, m.name, m.url, m.transforms = v, v.meta, k, v.url, v.transforms
                setattr(cls, k, m)

    @classmethod
    def verify(cls, obj):
        return obj

    def get_state_dict(self, progress=True):
        # No network: the ImageNet checkpoint cannot be fetched.  Return a
        # 3-channel-stem Swin-T state dict built by the reference class itself
        # under a fixed seed; make_golden.py overwrites every tensor with the
        # closed-form filler afterwards, so only the key set/shapes matter.
        from builder.models.src import swin_transformer as st
        g = torch.random.get_rng_state()
        torch.manual_seed(0)
        m = st.SwinTransformer(patch_size=[4, 4], embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24],
                               window_size=[7, 7], stochastic_depth_prob=0.2,
                               num_classes=len(self.meta["categories"]))
        torch.random.set_rng_state(g)
        return m.state_dict()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Install the placeholders and put the reference on sys.path."""
    import transformers  # noqa: F401  (must be imported before a spec-less 'torchvision' exists)
    ident = lambda *a, **k: (lambda f: f)
    tv = _mod("torchvision")
    tv.__path__ = []
    _mod("torchvision.ops").__path__ = []
    _mod("torchvision.ops.misc", MLP=_MLP, Permute=_Permute, Conv2dNormActivation=object)
    _mod("torchvision.ops.stochastic_depth", StochasticDepth=_StochasticDepth)
    _mod("torchvision.transforms").__path__ = []
    _mod("torchvision.transforms._presets", ImageClassification=object,
         InterpolationMode=types.SimpleNamespace(BICUBIC=3, BILINEAR=2))
    _mod("torchvision.utils", _log_api_usage_once=lambda *_: None)
    _mod("torchvision.models").__path__ = []
    _mod("torchvision.models._api", Weights=_Weights, WeightsEnum=_WeightsEnum, register_model=ident)
    _mod("torchvision.models._meta", _IMAGENET_CATEGORIES=["c%d" % i for i in range(1000)])
    _mod("torchvision.models._utils", _ovewrite_named_param=lambda kw, k, v: kw.__setitem__(k, v),
         handle_legacy_interface=ident, _ModelURLs=dict)
    _mod("monai").__path__ = []
    _mod("monai.networks").__path__ = []
    _mod("monai.networks.blocks").__path__ = []
    _mod("monai.networks.blocks.patchembedding", PatchEmbeddingBlock=object)
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
