"""Public model API, same names as the reference's nets/__init__.py:11-12."""
from .architectures import VNetDS, NeuralOperatorSeg, HartleyMHASeg
from .hnosegxs import HNOSegXS

__all__ = ['VNetDS', 'NeuralOperatorSeg', 'HartleyMHASeg', 'HNOSegXS']
