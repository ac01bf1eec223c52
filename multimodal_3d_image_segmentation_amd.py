"""Import shim: the package directory is named ``multimodal-3d-image-segmentation_amd`` (not a
valid Python identifier); this module exposes it as ``multimodal_3d_image_segmentation_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'multimodal-3d-image-segmentation_amd')]
__package__ = __name__
__file__ = _os.path.join(__path__[0], '__init__.py')
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, 'exec'))
