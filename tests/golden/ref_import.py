"""Import harness for the upstream reference (THIS container only).

The reference (``/root/reference``, pure Python/PyTorch) imports ``timm``,
``torchvision``, ``kornia`` and ``cv2`` at module top, none of which is
installed here.  None of those symbols does arithmetic on the VMAE predictor
path, so we install minimal ``sys.modules`` stand-ins and import
the reference modules unmodified.  The reference is only ever *executed* here to
produce the golden vectors under ``tests/golden/`` (see ``make_golden.py``); it
never travels to the GPU box and nothing in the product imports this file.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CWM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "cwm", "models"))


def _module(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs() -> None:
    import torch
    import torch.nn.functional as F

    if "timm" in sys.modules and getattr(sys.modules["timm"], "_cwm_stub", False):
        return

    def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def drop_path(x, drop_prob=0.0, training=False, scale_by_keep=True):
        assert not training or not drop_prob
        return x

    timm = _module("timm", _cwm_stub=True)
    timm.models = _module("timm.models")
    timm.models.registry = _module("timm.models.registry", register_model=lambda f: f)
    timm.models.layers = _module(
        "timm.models.layers", trunc_normal_=trunc_normal_, to_2tuple=to_2tuple, drop_path=drop_path
    )
    timm.data = _module("timm.data")
    timm.data.constants = _module(
        "timm.data.constants",
        IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406),
        IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225),
    )

    class CenterCrop:
        """Exact slicing; the reference only crops even paddings (perturbation.py:264,269)."""

        def __init__(self, size):
            self.size = to_2tuple(size)

        def __call__(self, x):
            h, w = self.size
            H, W = x.shape[-2:]
            top = int(round((H - h) / 2.0))
            left = int(round((W - w) / 2.0))
            return x[..., top : top + h, left : left + w]

    class Resize:
        def __init__(self, size, **kw):
            self.size = to_2tuple(size)

        def __call__(self, x):
            return F.interpolate(x.float(), size=self.size, mode="bilinear", align_corners=False)

    class Compose:
        def __init__(self, ts):
            self.ts = list(ts)

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToPILImage:
        def __call__(self, x):  # pragma: no cover - visualisation only
            raise NotImplementedError

    tv = _module("torchvision")
    tv.transforms = _module(
        "torchvision.transforms", CenterCrop=CenterCrop, Resize=Resize, Compose=Compose, ToPILImage=ToPILImage
    )
    tv.models = _module("torchvision.models", vgg16=None)
    _module("kornia")
    _module("cv2")


def import_reference():
    """Returns a namespace with the reference modules on the predictor path."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never write __pycache__ into the reference tree
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib

    ns = types.SimpleNamespace()
    ns.vmae = importlib.import_module("cwm.models.VideoMAE.vmae")
    ns.vutils = importlib.import_module("cwm.models.VideoMAE.utils")
    ns.conj = importlib.import_module("cwm.models.VideoMAE.conjoined_vmae")
    ns.transformer = importlib.import_module("cwm.models.transformer")
    ns.prediction = importlib.import_module("cwm.models.prediction")
    ns.masking = importlib.import_module("cwm.models.masking")
    ns.perturbation = importlib.import_module("cwm.models.perturbation")
    ns.patches = importlib.import_module("cwm.models.patches")
    ns.utils = importlib.import_module("cwm.models.utils")
    try:
        ns.segmentation = importlib.import_module("cwm.models.segmentation")
    except Exception as e:  # pragma: no cover - optional (needs scipy etc.)
        ns.segmentation = None
        ns.segmentation_error = e
    return ns
