"""Per-mode ('individual') Hartley weights (reference nets/hartley_operator.py:197-241, 294-317)."""
import torch

from .. import ops


def _reverse_cropped(z):
    """x[N-k] on the cropped grid: flip + roll by one along the three mode axes (index permutation)."""
    dims = (-3, -2, -1)
    return torch.roll(torch.flip(z, dims), (1, 1, 1), dims)


def hartley_mix_individual(z, weight, act=ops.ACT_NONE, full_spatial=None, x_full=None, modes=None):
    """use_transform=False: z is the cropped spectrum, the reversal is taken on the cropped grid
    (reference :294-299, with its documented difference at the highest negative frequency).
    use_transform=True (x_full given): the reversal is taken on the FULL spectrum before cropping
    (reference :198-241), which needs the frequency +m that the [low | high] block does not keep; the
    transform is therefore evaluated with m + 1 modes and both blocks are gathered from it."""
    if x_full is None:
        y = ops.PerModeHartleyFn.apply(z, _reverse_cropped(z).contiguous(), weight)
    else:
        spatial = tuple(x_full.shape[2:])
        # 2m + 1 == N on some axis: +m is only available from the un-truncated transform
        use_full = any(n > 1 and 2 * m + 1 == n for n, m in zip(spatial, modes))
        big, idx_k, idx_r = [], [], []
        for n, m in zip(spatial, modes):
            if n == 1:                                                   # degenerate axis of 2-D data
                big.append(0)
                idx_k.append(torch.zeros(1, dtype=torch.long, device=x_full.device))
                idx_r.append(idx_k[-1])
                continue
            mb = m + 1 if 2 * (m + 1) <= n else m
            big.append(mb)
            ks = list(range(m)) + list(range(-m, 0))                     # kept signed frequencies, [low | high]
            if use_full or 2 * mb == n:
                pos = lambda k, n=n: k % n                               # noqa: E731  natural order
            else:
                pos = lambda k, mb=mb: k if k >= 0 else k + 2 * mb       # noqa: E731  [low | high] block of mb modes
            idx_k.append(torch.tensor([pos(k) for k in ks], device=x_full.device))
            idx_r.append(torch.tensor([pos(-k) for k in ks], device=x_full.device))
        import numpy as np
        if use_full:
            zb = ops.DhtFullFn.apply(x_full, 1.0 / float(np.prod(spatial)))
        else:
            zb = ops.DhtCropFn.apply(x_full, tuple(big), 1.0 / float(np.prod(spatial)))
        zk, zr = zb, zb
        for ax in range(3):
            zk = zk.index_select(2 + ax, idx_k[ax])
            zr = zr.index_select(2 + ax, idx_r[ax])
        y = ops.PerModeHartleyFn.apply(zk.contiguous(), zr.contiguous(), weight)
    return ops.ActFn.apply(y, act) if act != ops.ACT_NONE else y
