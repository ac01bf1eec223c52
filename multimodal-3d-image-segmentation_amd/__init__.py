"""MI355X-native spectral-operator segmentation engine (drop-in for the reference's
``nets`` package and ``experiments/train_test.py`` loop; hot path in hand-written HIP)."""
from . import _lib, ops  # noqa: F401
from . import nets  # noqa: F401
from . import optim  # noqa: F401

__all__ = ['nets', 'ops', 'optim', '_lib']
