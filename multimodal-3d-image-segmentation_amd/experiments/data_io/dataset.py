"""Random affine augmentation on the GPU (reference experiments/data_io/dataset.py:63-244).

``ImageTransform`` keeps the reference's constructor and draws its random numbers from
``numpy.random.default_rng(seed)`` in the same order (augment?, one angle per non-zero rotation range,
one shift per non-zero shift range, zoom, then one coin per flipped axis), so a given seed produces the
same sequence of transforms; the resampling itself is one HIP gather kernel per image
(hno_affine_nearest) on tensors that already live in HBM, instead of SimpleITK in DataLoader workers.
"""
import numpy as np
import torch

from ... import _lib


def transform_matrix_offset_center(matrix, img_size):
    """Conjugates a homogeneous matrix so that it acts about the point size / 2 + 0.5 (reference :192-199)."""
    n = matrix.shape[0]
    to_centre, back = np.eye(n), np.eye(n)
    to_centre[:-1, -1] = np.asarray(img_size, dtype=np.float64) / 2.0 + 0.5
    back[:-1, -1] = -to_centre[:-1, -1]
    return to_centre @ matrix @ back


def _matrix12(transform_matrix, spatial):
    """(x, y[, z]) homogeneous matrix -> the 3 x 4 [M | t] rows hno_affine_nearest takes."""
    size_xyz = tuple(spatial)[::-1]
    full = transform_matrix_offset_center(np.asarray(transform_matrix, dtype=np.float64), size_xyz)
    if full.shape[0] == 3:   # 2-D: identity along z
        m = np.eye(4)
        m[:2, :2], m[:2, 3] = full[:2, :2], full[:2, 2]
        full = m
    return np.ascontiguousarray(full[:3, :4], dtype=np.float64)


def _resample(x, m12, cval, flip_mask):
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise _lib.HnoError('the GPU augmentation needs tensors on the GPU (no CPU fallback)')
    dtype = x.dtype
    xf = x.float().contiguous()
    C = xf.shape[0]
    sp = tuple(xf.shape[1:])
    D, H, W = ((1,) + sp) if len(sp) == 2 else sp
    out = torch.empty_like(xf)
    _lib.check(_lib.lib().hno_affine_nearest(_lib.ptr(xf), _lib.ptr(out), m12.ctypes.data, float(cval), int(flip_mask), C, D, H, W,
                                             _lib.stream_ptr()), 'hno_affine_nearest')
    return out if dtype == torch.float32 else out.to(dtype)


def apply_transform(x, transform_matrix, cval):
    """x (C, H, W) or (C, D, H, W) on the GPU; transform_matrix homogeneous in (x, y[, z]) (reference :202-237)."""
    return _resample(x, _matrix12(transform_matrix, x.shape[1:]), cval, 0)


def flip_axis(x, axis):
    """reference :240-244 (a copy here, not a view)."""
    return torch.flip(x, (axis,))


class ImageTransform:
    """Same arguments and random stream as the reference's ImageTransform (:63-189)."""

    def __init__(self, rotation_range=None, shift_range=None, zoom_range=None, flip=None, cval=0.,
                 augmentation_probability=1.0, seed=None):
        self.rotation_range, self.shift_range, self.zoom_range = rotation_range, shift_range, zoom_range
        self.flip, self.cval, self.augmentation_probability = flip, cval, augmentation_probability
        self.rng = np.random.default_rng(seed)

    # -- host side: the random draw, as a homogeneous matrix (or None) plus the flip bits ------------------
    def draw(self, shape):
        """shape = x.shape (C, ...).  Returns (transform_matrix or None, flip axes as a tuple of array axes)."""
        nd = len(shape) - 1
        if not self.rng.binomial(1, self.augmentation_probability):
            return None, ()
        rad = np.pi / 180
        theta = None
        if self.rotation_range is not None:
            if np.isscalar(self.rotation_range):
                assert nd == 2
                theta = rad * self.rng.uniform(-self.rotation_range, self.rotation_range) if self.rotation_range else 0
            else:
                assert len(self.rotation_range) == 3
                theta = [rad * self.rng.uniform(-r, r) if r else 0 for r in self.rotation_range]
        shift = None
        if self.shift_range is not None:
            assert len(self.shift_range) == nd
            shift = [self.rng.uniform(-s, s) * shape[1 + i] if s else 0 for i, s in enumerate(self.shift_range)]
        zoom = self.rng.uniform(self.zoom_range[0], self.zoom_range[1]) if self.zoom_range is not None else None

        mat = None
        if theta is not None:
            if np.isscalar(theta):
                if theta != 0:
                    c, s = np.cos(theta), np.sin(theta)
                    mat = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
            elif any(t != 0 for t in theta):
                # angles are given per (depth, height, width) axis; the matrix acts on (x, y, z) = (w, h, d), so the
                # depth angle turns about z: R = Rz(depth angle) Ry(height angle) Rx(width angle)
                a, b, g = theta[2], theta[1], theta[0]
                rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
                ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
                rz = np.array([[np.cos(g), -np.sin(g), 0], [np.sin(g), np.cos(g), 0], [0, 0, 1]])
                mat = np.eye(4)
                mat[:3, :3] = rz @ ry @ rx
        if shift is not None and any(s != 0 for s in shift):
            sm = np.eye(nd + 1)
            sm[:-1, -1] = np.asarray(shift[::-1])   # (x, y, z) order
            mat = sm if mat is None else sm @ mat
        if zoom is not None and zoom != 1:
            zm = np.eye(nd + 1)
            zm[:-1, :-1] *= zoom
            mat = zm if mat is None else zm @ mat
        flips = ()
        if self.flip is not None:
            assert len(self.flip) == nd
            flips = tuple(1 + i for i, f in enumerate(self.flip) if f and self.rng.random() < 0.5)
        return mat, flips

    def __call__(self, x, y=None):
        mat, flips = self.draw(tuple(x.shape))
        nd = x.ndim - 1
        if mat is not None or flips:
            mask = 0
            for ax in flips:   # array axis -> (depth, height, width) bit
                mask |= 1 << (ax - 1 + (3 - nd))
            m12 = _matrix12(mat if mat is not None else np.eye(nd + 1), x.shape[1:])
            x = _resample(x, m12, self.cval, mask)
            if y is not None:
                y = _resample(y, m12, self.cval, mask)
        return x if y is None else (x, y)
