"""Label and normalisation helpers of the reference's experiments/utils.py (:25-119) on the HIP kernels."""
import torch

from .. import _lib, ops


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


def normalize_modalities(data, mask_val=None, clip_val=None, return_stats=False):
    """Per-modality masked z-score on the GPU (reference utils.py:25-71; run.py:52-55 binds mask_val = 0).

    data: (C, ...) one sample, or (B, C, ...) a batch -- every (sample, modality) volume is normalised on its
    own, as the reference's per-sample ``x_processing`` does.  Values equal to ``mask_val`` (after clipping)
    are left out of mean / std and come out as 0."""
    if not isinstance(data, torch.Tensor):
        data = torch.as_tensor(data)
    if not data.is_cuda:
        raise _lib.HnoError('normalize_modalities runs on the GPU; move the raw volume there first (no CPU fallback)')
    x = data.float().contiguous()
    C, V = x.shape[0], x[0].numel()
    L = _lib.lib()
    ws = torch.empty(L.hno_zscore_workspace_bytes(C) // 8, device=x.device, dtype=torch.float64)
    out = torch.empty_like(x)
    stats = torch.empty((C, 2), device=x.device, dtype=torch.float32) if return_stats else None
    lo, hi = (float(clip_val[0]), float(clip_val[1])) if clip_val is not None else (0.0, 0.0)
    _lib.check(L.hno_zscore_modalities(_lib.ptr(x), _lib.ptr(out), _lib.ptr(stats), _lib.ptr(ws), C, V,
                                       int(mask_val is not None), float(mask_val if mask_val is not None else 0.0),
                                       int(clip_val is not None), lo, hi, _lib.stream_ptr()), 'hno_zscore_modalities')
    return (out, stats) if return_stats else out


def normalize_batch(batch, mask_val=None, clip_val=None):
    """(B, C, ...) raw batch -> normalised batch in one launch pair (B * C independent volumes)."""
    flat = batch.reshape((batch.shape[0] * batch.shape[1],) + tuple(batch.shape[2:]))
    return normalize_modalities(flat, mask_val, clip_val).reshape(batch.shape)


def save_model_summary(model, input_size, path=None):
    """Layer table of `model` for an input of `input_size` (reference utils.py:122-134, which delegates to
    ``torchinfo.summary(model, input_size=..., device='meta')``; torchinfo is not a dependency here).  Like the reference it
    runs a SHAPE-ONLY forward of a copy of the model on the ``meta`` device -- no memory, no kernels (ops._HnoFunction.meta)
    -- and lists every module with its output shape and parameter count."""
    import copy
    m = copy.deepcopy(model).to('meta')
    rows, hooks = [], []

    def shape_of(out):
        if torch.is_tensor(out):
            return list(out.shape)
        if isinstance(out, (tuple, list)):
            return [shape_of(o) for o in out]
        return None

    shapes = {}
    for name, mod in m.named_modules():
        def hook(mod, inp, out, name=name):
            shapes[name] = shape_of(out)
        hooks.append(mod.register_forward_hook(hook))
    was_training = m.training
    m.eval()
    with torch.no_grad():
        out = m(torch.empty(tuple(input_size), device='meta'))
    m.train(was_training)
    for h in hooks:
        h.remove()
    for name, mod in m.named_modules():      # modules whose kernels are fused into their parent's launch never run on their own
        own = sum(p.numel() for p in mod.parameters(recurse=False))
        rows.append((name or type(m).__name__, type(mod).__name__, shapes.get(name, '(fused into parent)'), own))
    total = sum(p.numel() for p in m.parameters())
    trainable = sum(p.numel() for p in m.parameters() if p.requires_grad)
    w = max([len(r[0]) for r in rows] + [10])
    lines = ['=' * (w + 70), f'{"Layer":<{w}}  {"Type":<28}{"Output shape":<28}{"Param #":>10}', '=' * (w + 70)]
    for name, typ, shp, own in rows:
        lines.append(f'{name:<{w}}  {typ:<28}{str(shp):<28}{own:>10,}')
    lines += ['=' * (w + 70), f'Total params: {total:,}', f'Trainable params: {trainable:,}',
              f'Non-trainable params: {total - trainable:,}', f'Input size: {tuple(input_size)}',
              f'Output size: {tuple(out.shape) if torch.is_tensor(out) else shape_of(out)}', '=' * (w + 70)]
    text = '\n'.join(lines)
    if path is not None:
        with open(path, 'w') as f:
            print(text, file=f)
    return text
