"""Discrete Hartley transforms on the HIP kernels (reference: nets/dht.py:16-66).

dhtn(x) = Re F(x) - Im F(x) with 1/N on the forward only; the "inverse" is the same
forward-sign transform left unscaled.  The un-truncated transform is the mode-truncated
kernel pair with every frequency kept (hno_dht3_full): any even or odd sizes, one, two or
three innermost dimensions (fewer dimensions are degenerate leading axes of size 1).
"""
import numpy as np
import torch

from .. import ops

_MAX_BC = 65535   # batch * channel planes per launch (grid.y)


def dhtn(x, dim, is_inverse=False):
    dims = sorted(d % x.ndim for d in dim)
    nd = len(dims)
    assert 1 <= nd <= 3, 'one to three dimensions can be transformed'
    assert dims == list(range(x.ndim - nd, x.ndim)), 'only the innermost dimensions can be transformed'
    lead = int(np.prod(x.shape[:x.ndim - nd])) if x.ndim > nd else 1
    spatial = tuple(x.shape[x.ndim - nd:])
    sp3 = (1,) * (3 - nd) + spatial
    if nd == 1:   # the plane kernels want the two innermost axes non-degenerate: (lead, 1, N) -> rows of one plane
        raise NotImplementedError('1-D dhtn is not provided by the HIP path (the reference only uses dht2 / dht3)')
    scale = 1.0 if is_inverse else 1.0 / float(np.prod(spatial))
    x4 = x.reshape((lead,) + sp3)
    outs = [ops.DhtFullFn.apply(x4[i:i + _MAX_BC].unsqueeze(0), scale).squeeze(0) for i in range(0, lead, _MAX_BC)]
    out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
    return out.reshape(x.shape)


def dht2(x, is_inverse=False):
    return dhtn(x, dim=(-2, -1), is_inverse=is_inverse)


def dht3(x, is_inverse=False):
    return dhtn(x, dim=(-3, -2, -1), is_inverse=is_inverse)
