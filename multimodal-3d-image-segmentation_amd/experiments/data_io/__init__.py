from .dataset import ImageTransform, apply_transform, flip_axis, transform_matrix_offset_center  # noqa: F401
from .input_data import InputData  # noqa: F401
