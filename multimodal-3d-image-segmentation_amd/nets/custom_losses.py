"""Pearson / Dice losses on the HIP reduction kernels (reference nets/custom_losses.py:17-133).

``y_true`` may be the reference's one-hot fp32 tensor (B,K,...) or -- cheaper -- a uint8 class
map (B,...) / (B,1,...): the kernels take class indices, so the 134 MB one-hot tensor of the
reference loop never needs to exist.
"""
import torch
from torch.nn import Module

from .. import ops


def _labels_u8(y_pred, y_true):
    if y_true.dtype == torch.uint8:
        lab = y_true.reshape((y_true.shape[0],) + tuple(y_pred.shape[2:]))
        return lab.contiguous()
    assert y_true.shape == y_pred.shape, 'y_true must be one-hot with the shape of y_pred, or a uint8 class map'
    return ops.onehot_to_u8(y_true)


def _run(y_pred, y_true, kind, param=0.0):
    assert y_pred.ndim in (3, 4, 5)
    lab = _labels_u8(y_pred, y_true)
    done = ops.precomputed_loss(y_pred, lab, ops.LOSS_KINDS[kind], param)      # the head took the sums already (ops.expected_loss)
    if done is not None:
        return done
    return ops.SegLossFn.apply(y_pred, lab, ops.LOSS_KINDS[kind], param)


def corrcoef(y_pred, y_true):
    """Pearson correlation per (batch, label) (reference :17-41).  Not differentiable on its own;
    use PCCLoss for training."""
    return _run(y_pred, y_true, 'pcc')[1][..., 0]


def dice_coef(y_pred, y_true):
    """Soft Dice per (batch, label) (reference :73-90)."""
    return _run(y_pred, y_true, 'dice')[1][..., 0]


class PCCLoss(Module):
    """mean(1 - (r + 1) / 2) (reference :44-70)."""
    hno_loss_spec = ('pcc', 0.0)

    @staticmethod
    def forward(y_pred, y_true):
        return _run(y_pred, y_true, 'pcc')[0]


class DiceLoss(Module):
    """mean(1 - dice) (reference :93-111)."""
    hno_loss_spec = ('dice', 0.0)

    @staticmethod
    def forward(y_pred, y_true):
        return _run(y_pred, y_true, 'dice')[0]


class ExpDiceLoss(Module):
    """mean((-ln clamp(dice))^exp) (reference :114-133)."""

    def __init__(self, exp=0.3):
        super().__init__()
        self.exp = exp

    @property
    def hno_loss_spec(self):
        return 'expdice', self.exp

    def forward(self, y_pred, y_true):
        return _run(y_pred, y_true, 'expdice', self.exp)[0]
