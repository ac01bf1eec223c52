"""Discrete Hartley transforms on the HIP kernels (reference: nets/dht.py:16-66).

dhtn(x) = Re F(x) - Im F(x) with 1/N on the forward only; the "inverse" is the same
forward-sign transform left unscaled.  A full (un-truncated) transform is the mode-truncated
kernel with every mode kept, which needs even sizes (2m = N); odd sizes keep N-1 modes and are
therefore only available through TransformCrop / PadInverse (which is all the models use).
"""
import numpy as np
import torch

from .. import ops


def _full_modes(spatial):
    if any(s % 2 for s in spatial):
        raise NotImplementedError('un-truncated dhtn on odd sizes is not provided by the HIP path; '
                                  'use TransformCrop/PadInverse (mode-truncated) instead')
    return tuple(s // 2 for s in spatial)


def _unshuffle(z, spatial):
    """[low | high] block with every mode kept is already natural order (0..N-1)."""
    return z


def dhtn(x, dim, is_inverse=False):
    dims = sorted(d % x.ndim for d in dim)
    nd = len(dims)
    assert dims == list(range(x.ndim - nd, x.ndim)), 'only the innermost dimensions can be transformed'
    lead = x.shape[:x.ndim - nd]
    spatial = tuple(x.shape[x.ndim - nd:])
    x5 = x.reshape((1, int(np.prod(lead)) if lead else 1) + (1,) * (3 - nd) + spatial)
    sp3 = tuple(x5.shape[2:])
    if nd < 3:
        raise NotImplementedError('2-D dhtn is not provided by the HIP path yet')
    modes = _full_modes(sp3)
    scale = 1.0 if is_inverse else 1.0 / float(np.prod(sp3))
    out = ops.DhtCropFn.apply(x5, modes, scale)
    return out.reshape(x.shape)


def dht2(x, is_inverse=False):
    return dhtn(x, dim=(-2, -1), is_inverse=is_inverse)


def dht3(x, is_inverse=False):
    return dhtn(x, dim=(-3, -2, -1), is_inverse=is_inverse)
