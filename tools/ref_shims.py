"""Import harness for the *reference's own Python* (build container only).

Used by tools/make_golden.py to produce golden vectors.  Nothing from
/root/reference is copied: the reference is imported in place with
  * sys.modules stubs for packages that are not installed here (cv2, IPython,
    visdom, plyfile, vispy, OpenGL, torchvision, torchsample, torch._six),
  * a source patch at import for KPD/src/utils/img.py (``async=True`` is a
    SyntaxError on Python >= 3.7),
  * ``.cuda()`` neutralised (no GPU in this container).
torchvision's ``Resize(size, interpolation=3)`` + ``ToTensor`` are stood in for
by Pillow itself (that is what torchvision calls for a PIL image);
torchsample's ``SpecialCrop(size, 1)`` / ``Pad(size)`` are restated from their
published behaviour (top-left crop; centred zero pad with ceil/floor split) --
flagged "restated third-party" in tests/golden/MANIFEST.json.
"""
from __future__ import annotations

import importlib.abc
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/3_6Dpose_estimator"


class _Permissive(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        def _f(*a, **k):
            return 0
        return _f


def _stub(name, **attrs):
    m = _Permissive(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _ImgPatchLoader(importlib.abc.Loader):
    def __init__(self, path):
        self.path = path

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        src = open(self.path).read().replace("async=True", "non_blocking=True")
        module.__file__ = self.path
        exec(compile(src, self.path, "exec"), module.__dict__)


class _ImgPatchFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname == "KPD.src.utils.img":
            p = os.path.join(REF, "KPD/src/utils/img.py")
            return importlib.util.spec_from_loader(fullname, _ImgPatchLoader(p))
        return None


def install():
    import torch
    from PIL import Image

    if getattr(install, "_done", False):
        return
    install._done = True
    sys.dont_write_bytecode = True
    sys.argv = ["x", "--sp", "--nClasses", "50"]
    sys.path.insert(0, REF)
    os.chdir(REF)

    _stub("cv2", VideoWriter_fourcc=lambda *a: 0)
    _stub("IPython", embed=lambda *a, **k: None)
    _stub("visdom")
    _stub("plyfile", PlyData=object)
    vis = _stub("vispy")
    vis.app = _stub("vispy.app", Canvas=object)
    vis.gloo = _stub("vispy.gloo")
    _stub("OpenGL"); _stub("OpenGL.GL")
    _stub("scipy.misc")
    six = _stub("torch._six")
    six.string_classes = (str,)
    six.int_classes = (int,)

    # torchvision.transforms: Compose / Resize / ToTensor on PIL images
    tv = _stub("torchvision")
    tvt = _stub("torchvision.transforms")
    tv.transforms = tvt

    class Compose:
        def __init__(self, ts):
            self.ts = ts
        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size, interpolation=2):
            self.size, self.interp = size, interpolation
        def __call__(self, img):
            h, w = self.size
            return img.resize((w, h), self.interp)

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img, dtype=np.uint8)
            t = torch.from_numpy(a.transpose(2, 0, 1).copy())
            return t.float().div(255)

    tvt.Compose, tvt.Resize, tvt.ToTensor = Compose, Resize, ToTensor

    # torchsample.transforms: restated third-party behaviour
    ts = _stub("torchsample")
    tst = _stub("torchsample.transforms")
    ts.transforms = tst

    class SpecialCrop:
        def __init__(self, size, crop_type=0):
            self.size, self.crop_type = size, crop_type
        def __call__(self, x):
            assert self.crop_type == 1
            return x[:, 0:int(self.size[0]), 0:int(self.size[1])]

    class Pad:
        def __init__(self, size):
            self.size = size
        def __call__(self, x):
            x = x.numpy()
            shape_diffs = [int(np.ceil((int(i_s) - d_s))) for d_s, i_s in zip(x.shape, self.size)]
            shape_diffs = np.maximum(shape_diffs, 0)
            pad_sizes = [(int(np.ceil(s / 2.)), int(np.floor(s / 2.))) for s in shape_diffs]
            x = np.pad(x, pad_sizes, mode="constant")
            return torch.from_numpy(x)

    tst.SpecialCrop, tst.Pad = SpecialCrop, Pad

    sys.meta_path.insert(0, _ImgPatchFinder())
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
