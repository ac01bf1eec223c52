"""Label helpers of the reference's experiments/utils.py (:74-119) on the HIP kernels."""
import torch

from .. import ops


def to_categorical(y, num_classes=None):
    """(B,1,D,H,W) int/float label tensor -> one-hot fp32 (B,K,D,H,W) (reference utils.py:74-97).
    The training loop does not need the one-hot tensor (the loss kernels take uint8 class maps);
    this exists for API compatibility."""
    assert y.shape[1] == 1, 'Can only handle single label per pixel.'
    if not num_classes:
        num_classes = int(y.max().item()) + 1
    return ops.labels_prepare(y, num_classes, None, want_onehot=True)[1]


def remap_labels(label, mapping):
    """Copy of `label` with every key of `mapping` replaced by its value; keys are matched against the
    ORIGINAL labels (reference utils.py:100-119)."""
    if not isinstance(label, torch.Tensor):
        raise ValueError('Input "label" must be a PyTorch tensor on the GPU for the HIP path.')
    flat = label.reshape(label.shape[0], 1, *label.shape[1:]) if label.ndim == 4 else label
    num = int(max(max(mapping.values()), label.max().item())) + 1
    u8 = ops.labels_prepare(flat, num, mapping)
    return u8.to(label.dtype).reshape(label.shape)


def labels_to_u8(y, num_labels, mapping=None):
    """What the loop actually uses: (B,1,...) labels -> uint8 class map (B,...) with the optional remap fused."""
    return ops.labels_prepare(y, num_labels, mapping)
