"""Closed-form, seed-free input generators shared by make_golden.py and the tests.

Golden fixtures store only expected OUTPUTS (and weights); the inputs are regenerated
from these formulas on both sides, which keeps the fixtures small.
"""
import numpy as np


def formula_tensor(shape, tag=0, dtype=np.float32):
    """Deterministic pseudo-random-looking values in roughly [-1.5, 1.5] (float64 math)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * 0.7390851332151607 + 1.3 * tag + 0.25) + 0.5 * np.cos(i * 0.01170019 * (tag + 1) + 0.5 * tag)
    v = v + 0.1 * np.sin(i * i * 1e-6 + tag)
    return v.reshape(shape).astype(dtype)


def formula_labels(shape, num_classes, tag=0):
    """Integer labels in [0, num_classes) with spatial structure, shape (B,1,...)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = np.floor((np.sin(i * 0.00931 + tag) * 0.5 + 0.5) * num_classes * 0.999 + 0.3 * np.sin(i * 0.7 + tag))
    return np.clip(v, 0, num_classes - 1).reshape(shape).astype(np.float32)


def sample_indices(n_total, n_samples, tag=0):
    """Fixed, well-spread flat indices for storing a subset of a big output."""
    step = 0.6180339887498949
    return np.unique(np.floor(((np.arange(n_samples) * step + 0.1 * tag) % 1.0) * n_total).astype(np.int64))


# (B, C, spatial, modes) cases shared by make_golden.py (G2) and the parity tests
CROP_CASES = [
    (1, 2, (13, 15, 11), (3, 4, 2)),
    (1, 2, (33, 33, 33), (10, 14, 14)),
    (1, 1, (65, 65, 65), (10, 14, 14)),
    (1, 2, (61, 61, 40), (10, 14, 14)),
    (2, 1, (9, 8, 7), (10, 14, 14)),       # clamped: m -> s // 2
    (1, 3, (16, 12, 20), (8, 6, 10)),      # even sizes, 2m == N on every axis
]


# 2-D (ndim = 4) cases: (B, C, (H, W), modes)
CROP_CASES_2D = [
    (2, 3, (15, 11), (4, 2)),
    (1, 2, (33, 65), (14, 14)),
    (2, 1, (9, 8), (10, 14)),              # clamped
    (1, 3, (16, 20), (8, 10)),             # 2m == N
]


def formula_volume(shape, tag=0, noise=0.25, dtype=np.float32):
    """Smooth multi-channel volume (B,C,D,H,W): a few low-frequency waves per channel plus broadband
    texture -- like z-scored MR volumes, most of the energy sits in the low modes the operators keep."""
    b, c, d, h, w = shape
    z, y, x = np.meshgrid(np.arange(d) / d, np.arange(h) / h, np.arange(w) / w, indexing='ij')
    out = np.empty(shape, dtype=np.float64)
    for bi in range(b):
        for ci in range(c):
            s = 1.0 + bi * c + ci + 0.37 * tag
            vol = np.zeros((d, h, w))
            for j in range(1, 4):
                fz, fy, fx = (j + s) % 3, (2 * j + s) % 4, (j * j + s) % 3
                vol += (1.0 / j) * np.sin(2 * np.pi * (fz * z + fy * y + fx * x) + 0.7 * j * s)
            out[bi, ci] = vol
    out += noise * formula_tensor(shape, tag + 100, np.float64)
    return out.astype(dtype)


# small, well-conditioned HNOSeg-XS variants (golden G6s): name -> (ctor kwargs, input shape)
SMALL_MODELS = {
    'xs_small': (dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=[2, 2, 2, 2], num_modes=(4, 5, 5)),
                 (1, 2, 24, 20, 28)),
    'xs_odd_noskip': (dict(in_channels=1, out_channels=2, filters=8, num_transform_blocks=[1, 2, 1], num_modes=(3, 3, 4),
                           use_unet_skip=False), (2, 1, 18, 22, 26)),
    'xs_clamped': (dict(in_channels=2, out_channels=2, filters=16, num_transform_blocks=[1, 1], num_modes=(10, 14, 14)),
                   (1, 2, 16, 20, 24)),
    # per-mode weights (the unfused NeuralOperatorBlock path) and the add-skip / deep-supervision switches
    'xs_individual': (dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=[1, 2, 1], num_modes=(3, 3, 4),
                           weights_type='individual'), (1, 2, 20, 20, 24)),
    'xs_add_ds': (dict(in_channels=2, out_channels=2, filters=8, num_transform_blocks=[2, 1, 2], num_modes=(3, 4, 4),
                       use_block_concat=False, use_deep_supervision=True), (1, 2, 20, 24, 20)),
}

# HNOXSBlock with the spatial conv branch inside every frequency-domain layer (nets/hnosegxs.py:211,293-294): the
# HNOSegXS constructor never sets it, so the block is pinned on its own (golden G6b).
XSBLOCK_BRANCH = dict(num_convs=2, in_channels=8, out_channels=8, num_modes=(3, 4, 3), shape=(1, 8, 11, 12, 13))


# ---- training-loop trajectory case (golden G8), shared by make_golden.py and the tests
TRAIN_CASE = {
    'model': dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=[1, 1, 1, 1], num_modes=(3, 4, 4)),
    'image_size': (16, 20, 24), 'batch_size': 2, 'num_train': 4, 'num_valid': 2,
    'epochs': 4, 'lr': 5e-3, 'eta_min': 1e-3,
    'mapping': {3: 2},           # labels are drawn from 0..3 and label 3 is merged into 2
}


class _TrainInput:
    """Duck type of the reference's InputData (methods used by training()), deterministic data."""

    def __init__(self):
        c = TRAIN_CASE
        self.batch_size = c['batch_size']
        self._n = (c['num_train'], c['num_valid'])
        self._size = c['image_size']

    def _sample(self, i):
        import torch
        cin = TRAIN_CASE['model']['in_channels']
        x = formula_volume((1, cin) + self._size, 50 + i)[0]
        y = formula_labels((1, 1) + self._size, 4, 60 + i)[0]
        return torch.from_numpy(x), torch.from_numpy(y)

    def _flow(self, first, count):
        import torch
        for i in range(0, count, self.batch_size):
            xs, ys = zip(*[self._sample(first + j) for j in range(i, min(i + self.batch_size, count))])
            yield torch.stack(xs), torch.stack(ys)

    def get_train_flow(self, shuffle=True):     # deterministic order: the trajectory must be reproducible
        outer = self

        class It:
            def __iter__(self_inner):
                return outer._flow(0, outer._n[0])
        return It()

    def get_valid_flow(self):
        outer = self

        class It:
            def __iter__(self_inner):
                return outer._flow(outer._n[0], outer._n[1])
        return It()

    def get_train_num_batches(self):
        return -(-self._n[0] // self.batch_size)

    def get_valid_num_batches(self):
        return -(-self._n[1] // self.batch_size)

    def get_train_image_size(self):
        return self._size


def make_train_input():
    return _TrainInput()


# ---- inference protocol case (golden G14): the reference's testing() on three deterministic samples, batch size 1
TEST_CASE = {
    'model': dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=[1, 1, 1, 1], num_modes=(3, 4, 4)),
    'image_size': (20, 24, 28), 'num_test': 3, 'mapping': {1: 5, 2: 9},
}


class _TestInput:
    """Duck type of the reference's InputData for testing() (train_test.py:359-366): batch size 1, (x, y) pairs."""
    batch_size = 1

    def __init__(self):
        self.data_lists_test = [[f'case_{i}' for i in range(TEST_CASE['num_test'])]]

    def get_test_num_batches(self):
        return TEST_CASE['num_test']

    def get_test_flow(self):
        import torch
        size, cin = TEST_CASE['image_size'], TEST_CASE['model']['in_channels']
        for i in range(TEST_CASE['num_test']):
            x = formula_volume((1, cin) + size, 70 + i)
            y = formula_labels((1, 1) + size, TEST_CASE['model']['out_channels'], 80 + i)
            yield torch.from_numpy(x), torch.from_numpy(y)


def make_test_input():
    return _TestInput()


# tiny NeuralOperatorSeg variants (golden G7): name -> (ctor kwargs, input shape)
NOSEG_MODELS = {
    'hnoseg': (dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=3, num_modes=(4, 5, 5),
                    transform_type='Hartley'), (1, 2, 24, 20, 28)),
    'fnoseg': (dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=3, num_modes=(4, 5, 5),
                    transform_type='Fourier'), (1, 2, 24, 20, 28)),
    'fnoseg_addskip_bias': (dict(in_channels=1, out_channels=2, filters=8, num_transform_blocks=2, num_modes=(3, 3, 4),
                                 transform_type='Fourier', use_block_concat=False, use_bias_conv_branch=True),
                            (2, 1, 18, 22, 26)),
    'fno_individual': (dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=2, num_modes=(3, 4, 4),
                            transform_type='Fourier', weights_type='individual', use_bias_conv_branch=True,
                            use_block_skip=False), (2, 2, 24, 20, 28)),          # the FNO configuration (config_fno.ini)
    'hno_individual': (dict(in_channels=2, out_channels=2, filters=8, num_transform_blocks=2, num_modes=(3, 4, 4),
                            transform_type='Hartley', weights_type='individual'), (1, 2, 24, 20, 28)),
    'hnoseg_noskip_clamped': (dict(in_channels=2, out_channels=2, filters=8, num_transform_blocks=2,
                                   num_modes=(10, 14, 14), transform_type='Hartley', use_block_skip=False),
                              (1, 2, 16, 20, 24)),
    # (modes (3, 4, 4) puts one pre-activation of block 0 at 1e-7, on the SELU kink: the sign of a rounding error then
    # decides between two derivatives and fp32 implementations legitimately disagree by 4e-3 -- not a usable fixture)
    'hnoseg_deep_supervision': (dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=2, num_modes=(4, 4, 5),
                                     transform_type='Hartley', use_deep_supervision=True), (1, 2, 24, 20, 28)),
}


# Hartley MHA operator cases (golden G4): (in_ch, key_dim, heads, modes, patch, n_inputs) on (1, 6, 12, 14, 12)
MHA_CASES = [
    (6, 4, 2, (2, 3, 2), (2, 1, 2), 1),
    (6, 4, 2, (2, 3, 2), None, 1),
    (6, 4, 3, (2, 2, 2), (2, 2, 2), 2),
    (6, 5, 2, (3, 3, 2), None, 3),
]
# (in_channels, key_dim, heads, modes, patch, number of inputs, use_transform) with use_bias=True (golden G12)
MHA_BIAS_CASES = [
    (6, 4, 2, (2, 3, 2), (2, 1, 2), 1, True),
    (6, 4, 3, (2, 2, 2), None, 2, True),
    (6, 5, 2, (3, 3, 2), (2, 2, 2), 3, False),
]
MHASEG_MODEL = (dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=2, num_heads=2, num_modes=(4, 4, 6),
                     patch_size=(2, 2, 2)), (1, 2, 24, 20, 28))


# tiny V-Net-DS variants (golden G7v): name -> (ctor kwargs, input shape)
VNET_MODELS = {
    'vnet_ds': (dict(in_channels=2, out_channels=3, base_num_filters=4, num_blocks=[1, 2, 1], right_leg_indexes=[0, 1, 2]),
                (1, 2, 20, 24, 28)),
    'vnet_noresize_odd': (dict(in_channels=1, out_channels=2, base_num_filters=4, num_blocks=[1, 1], use_resize=False,
                               right_leg_indexes=None), (2, 1, 9, 11, 13)),
    # round 6: the `kernel_size` constructor argument (nets/architectures.py:55-70) -- 5 x 5 x 5 section, down and transposed convolutions
    'vnet_ds_k5': (dict(in_channels=2, out_channels=3, base_num_filters=4, num_blocks=[1, 1], right_leg_indexes=[0, 1], kernel_size=5),
                   (1, 2, 16, 20, 24)),
}


# ---- bf16 autocast goldens (G7b): (class name, ctor kwargs, input shape); channel counts are multiples of 8 (the bf16 path's unit)
BF16_MODELS = {
    'vnet_ds_bf16': ('VNetDS', dict(in_channels=2, out_channels=3, base_num_filters=8, num_blocks=[1, 2, 1],
                                    right_leg_indexes=[0, 1, 2]), (1, 2, 20, 24, 28)),
    'vnet_oneleg_bf16': ('VNetDS', dict(in_channels=4, out_channels=2, base_num_filters=8, num_blocks=[1, 1],
                                        right_leg_indexes=None), (2, 4, 18, 22, 14)),
    'fnoseg_bf16': ('NeuralOperatorSeg', dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=3, num_modes=(4, 5, 5),
                                             transform_type='Fourier'), (1, 2, 24, 20, 28)),
    'hnoseg_bf16': ('NeuralOperatorSeg', dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=3, num_modes=(4, 5, 5),
                                             transform_type='Hartley'), (1, 2, 24, 20, 28)),
    # 24 filters: the channel counts of the BASELINE configurations, for which the pointwise kernels have bf16 matrix-core variants
    'fnoseg24_bf16': ('NeuralOperatorSeg', dict(in_channels=2, out_channels=3, filters=24, num_transform_blocks=3, num_modes=(4, 5, 5),
                                               transform_type='Fourier'), (1, 2, 24, 20, 28)),
    'hnoseg24_bf16': ('NeuralOperatorSeg', dict(in_channels=2, out_channels=3, filters=24, num_transform_blocks=2, num_modes=(4, 5, 5),
                                               transform_type='Hartley'), (1, 2, 24, 20, 28)),
    'xs24_bf16': ('HNOSegXS', dict(in_channels=2, out_channels=3, filters=24, num_transform_blocks=[1, 2, 1, 1], num_modes=(4, 5, 5)),
                  (1, 2, 24, 20, 28)),
}


# ---- input pipeline (G11)
AUG_CASES = {
    'aug3d': (dict(rotation_range=[30, 10, 5], shift_range=[0.2, 0.1, 0.3], zoom_range=[0.8, 1.2], flip=[True, False, True],
                   augmentation_probability=0.8, seed=7), (2, 6, 7, 9)),
    'brats': (dict(rotation_range=[30, 0, 0], shift_range=[0.2, 0.2, 0.2], zoom_range=[0.8, 1.2],
                   augmentation_probability=0.8, seed=11), (1, 8, 10, 12)),
    'aug2d': (dict(rotation_range=25, shift_range=[0.1, 0.2], zoom_range=[0.7, 1.3], flip=[False, True],
                   augmentation_probability=0.9, seed=3), (1, 9, 11)),
}


def raw_modalities():
    """(3, 10, 12, 14) raw-MR-like volume: per-modality offset / scale and a zero background (the mask)."""
    vol = formula_tensor((3, 10, 12, 14), 400).astype(np.float64)
    vol = vol * np.array([300.0, 40.0, 1.0]).reshape(3, 1, 1, 1) + np.array([500.0, 90.0, 0.5]).reshape(3, 1, 1, 1)
    vol[:, :2] = 0
    vol[:, :, :3, :4] = 0
    return vol.astype(np.float32)


# tiny 2-D (ndim = 4) models (golden G13): name -> (class name, ctor kwargs, input shape)
MODELS_2D = {
    'xs2d': ('HNOSegXS', dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=[1, 2, 1, 1], num_modes=(4, 5), ndim=4),
             (1, 2, 24, 28)),
    'xs2d_ds_noskip': ('HNOSegXS', dict(in_channels=1, out_channels=2, filters=8, num_transform_blocks=[1, 1], num_modes=(3, 3),
                                        use_deep_supervision=True, use_unet_skip=False, ndim=4), (2, 1, 17, 20)),
    'hnoseg2d': ('NeuralOperatorSeg', dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=2, num_modes=(4, 5),
                                           transform_type='Hartley', ndim=4), (1, 2, 24, 28)),
    'fnoseg2d': ('NeuralOperatorSeg', dict(in_channels=2, out_channels=2, filters=8, num_transform_blocks=2, num_modes=(4, 5),
                                           transform_type='Fourier', ndim=4), (1, 2, 24, 28)),
    'fno2d_individual': ('NeuralOperatorSeg', dict(in_channels=2, out_channels=2, filters=8, num_transform_blocks=2, num_modes=(3, 4),
                                                   transform_type='Fourier', weights_type='individual', use_bias_conv_branch=True,
                                                   use_block_skip=False, ndim=4), (1, 2, 24, 28)),
    'mhaseg2d': ('HartleyMHASeg', dict(in_channels=2, out_channels=3, filters=8, num_transform_blocks=2, num_heads=2,
                                       num_modes=(4, 6), patch_size=(2, 2), ndim=4), (1, 2, 24, 28)),
    'xs2d_sigmoid': ('HNOSegXS', dict(in_channels=2, out_channels=2, filters=8, num_transform_blocks=[1, 1], num_modes=(3, 4),
                                      output_activation='sigmoid', ndim=4), (1, 2, 20, 24)),      # multi-label style output
    'vnet2d': ('VNetDS', dict(in_channels=2, out_channels=3, base_num_filters=4, num_blocks=[1, 2, 1], right_leg_indexes=[0, 1, 2],
                              ndim=4), (1, 2, 40, 36)),
}
