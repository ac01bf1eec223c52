"""Discrete Hartley transforms on the HIP kernels (reference: nets/dht.py:16-66).

dhtn(x) = Re F(x) - Im F(x) with 1/N on the forward only; the "inverse" is the same
forward-sign transform left unscaled.  The un-truncated transform is the mode-truncated
kernel pair with every frequency kept (hno_dht3_full): any even or odd sizes, two or three
innermost dimensions (two: a degenerate leading axis of size 1).  Round 5: a first axis longer
than the D-axis kernels' 63 points and the 1-D transform run as fp32 matrix-core GEMMs against
cos / sin tables (hno_bmm) -- the reference accepts any size and 1..3 dims (nets/dht.py:16-36).
"""
import numpy as np
import torch

from .. import ops

_MAX_BC = 65535   # batch * channel planes per launch (grid.y)
_MAX_M0 = 31      # the D-axis kernels keep at most 31 frequencies per sign (hno_dht.hip): first axes beyond 63 points take the GEMM route
_tables = {}


def _cos_sin(N, device):
    """cos / sin (N x N) of 2 pi k n / N, computed in float64 (angles reduced modulo N first), cached per (N, device)"""
    key = (int(N), str(device))
    hit = _tables.get(key)
    if hit is None:
        if len(_tables) > 16:
            _tables.clear()
        k = np.arange(N, dtype=np.int64)
        ang = 2.0 * np.pi * ((np.outer(k, k) % N).astype(np.float64) / N)
        hit = _tables[key] = (torch.from_numpy(np.cos(ang)).float().to(device), torch.from_numpy(np.sin(ang)).float().to(device))
    return hit


def _dht1(x2, scale):
    """1-D transform of the rows of x2 (lead, N): H = x (cos + sin), one fp32 matrix-core GEMM (hno_bmm)"""
    c, s = _cos_sin(x2.shape[-1], x2.device)
    return ops.BmmFn.apply(x2.unsqueeze(0), (c + s).unsqueeze(0), False, False, float(scale)).squeeze(0)


def _dht3_long_axis0(x4, scale):
    """3-D transform whose first axis is longer than the D-axis kernels' 63 points: cas(a + b) = cos a cas b + sin a cas(-b), so
    H3[k0, k1, k2] = sum_n0 cos(2 pi k0 n0 / N0) H2[n0, k1, k2] + sin(2 pi k0 n0 / N0) H2[n0, -k1, -k2]
    with H2 the 2-D transform of every (N1, N2) plane (the plane kernels, first axis degenerate) and the first-axis sums two batched
    fp32 matrix-core GEMMs (hno_bmm); the frequency reversal is a flip + roll of H2 (data movement only)."""
    lead, N0, N1, N2 = x4.shape
    step = max(1, _MAX_BC // N0)               # (b * c planes per launch: grid.y)
    parts = [ops.DhtFullFn.apply(x4[i:i + step].reshape(1, -1, 1, N1, N2), 1.0).reshape(-1, N0, N1, N2) for i in range(0, lead, step)]
    h2 = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
    rev = torch.roll(torch.flip(h2, dims=(2, 3)), shifts=(1, 1), dims=(2, 3))
    c, s = _cos_sin(N0, x4.device)
    cb, sb = c.expand(lead, N0, N0).contiguous(), s.expand(lead, N0, N0).contiguous()
    out = ops.AddFn.apply(ops.BmmFn.apply(cb, h2.reshape(lead, N0, N1 * N2), False, False, float(scale)),
                          ops.BmmFn.apply(sb, rev.reshape(lead, N0, N1 * N2), False, False, float(scale)))
    return out.reshape(lead, N0, N1, N2)


def dhtn(x, dim, is_inverse=False):
    dims = sorted(d % x.ndim for d in dim)
    nd = len(dims)
    assert 1 <= nd <= 3, 'one to three dimensions can be transformed'
    assert dims == list(range(x.ndim - nd, x.ndim)), 'only the innermost dimensions can be transformed'
    lead = int(np.prod(x.shape[:x.ndim - nd])) if x.ndim > nd else 1
    spatial = tuple(x.shape[x.ndim - nd:])
    sp3 = (1,) * (3 - nd) + spatial
    scale = 1.0 if is_inverse else 1.0 / float(np.prod(spatial))
    if nd == 1:   # (the plane kernels want the two innermost axes non-degenerate: a 1-D transform is one GEMM with the cas matrix)
        return _dht1(x.reshape(lead, spatial[0]), scale).reshape(x.shape)
    x4 = x.reshape((lead,) + sp3)
    if sp3[0] // 2 > _MAX_M0:
        return _dht3_long_axis0(x4, scale).reshape(x.shape)
    outs = [ops.DhtFullFn.apply(x4[i:i + _MAX_BC].unsqueeze(0), scale).squeeze(0) for i in range(0, lead, _MAX_BC)]
    out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
    return out.reshape(x.shape)


def dht2(x, is_inverse=False):
    return dhtn(x, dim=(-2, -1), is_inverse=is_inverse)


def dht3(x, is_inverse=False):
    return dhtn(x, dim=(-3, -2, -1), is_inverse=is_inverse)
