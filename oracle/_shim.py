"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

In-process import shim that lets the UNMODIFIED reference (/root/reference, PKU-Alignment/align-anything)
be imported in the build container, where several of its import-time dependencies are absent
(deepspeed, peft, diffusers, librosa, torchvision, wandb, tensorboard, ...) and where the installed
transformers (5.x) moved three names the reference imports from `transformers.tokenization_utils`.

Only `oracle/gen_golden.py` uses this, to produce the committed fixtures under tests/golden/.
/root/reference does not exist on the GPU box, so nothing that runs there may import this module.
Recipe: SURVEY.md §8(c).
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = '/root/reference'
_STUBS = {'librosa', 'torchvision', 'deepspeed', 'peft', 'diffusers', 'wandb', 'tensorboard', 'torchaudio',
          'soundfile', 'vllm', 'ray', 'gradio', 'cv2', 'av', 'decord', 'moviepy'}


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split('.')[0] in _STUBS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        return None


_installed = False


def install() -> None:
    """Make `import align_anything...` work in this container (idempotent)."""
    global _installed
    if _installed:
        return
    import transformers  # noqa: F401  -- must be imported (and probed) BEFORE the stubs exist
    from transformers import AutoProcessor, AutoTokenizer, CLIPImageProcessor  # noqa: F401
    import transformers.image_utils  # noqa: F401
    import transformers.tokenization_utils as tu
    import transformers.tokenization_utils_base as tub
    for n in ('BatchEncoding', 'PaddingStrategy', 'TruncationStrategy'):
        if not hasattr(tu, n):
            setattr(tu, n, getattr(tub, n))
    tb = types.ModuleType('torch.utils.tensorboard')

    class SummaryWriter:  # pragma: no cover - dummy
        def __init__(self, *a, **k):
            pass

    tb.SummaryWriter = SummaryWriter
    sys.modules['torch.utils.tensorboard'] = tb
    sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True
