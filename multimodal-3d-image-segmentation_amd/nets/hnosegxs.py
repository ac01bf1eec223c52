"""HNOSeg-XS on the HIP kernels (reference nets/hnosegxs.py:20-494).

Same constructor kwargs, attributes, module tree and state-dict keys as the reference, so
reference checkpoints load directly.  Data flow of one block (reference :253-279):

    [mapping_conv] -> TransformCrop -> n_XS x (z <- selu(W z + z)) -> PadInverse -> selu
                   -> conv_concat(cat[x, skip])

runs as: hno_pwconv (concat fused) -> hno_dht3_crop -> hno_specmix_shared -> hno_pad_idht3
(SELU fused on store) -> hno_pwconv (concat fused); the torch.cat / zeros / slice traffic of
the reference does not exist here.
"""
from functools import partial
from typing import Union

import os

import numpy as np
import torch
from torch import nn

from .. import ops
from .hartley_operator import HartleyOperator
from .nets_utils import ConvNormAct, init_weights_for_snn, spatial_padcrop, _is_selu


def _tuple_modes(num_modes, ndim):
    if np.isscalar(num_modes):
        return (num_modes,) * (ndim - 2)
    assert len(num_modes) == ndim - 2
    return tuple(num_modes)


class TransformCrop(nn.Module):
    """dhtn + gather of the [low | high] mode block (reference :332-410)."""

    def __init__(self, num_modes, ndim):
        super().__init__()
        assert ndim in (4, 5)
        self.num_modes = _tuple_modes(num_modes, ndim)

    def forward(self, x):
        if x.ndim == 4:   # 2-D (reference _call2d :355-376): a degenerate leading axis through the same kernels
            spatial = tuple(x.shape[2:])
            modes = (0,) + ops.clamp_modes(self.num_modes, spatial)
            return ops.DhtCropFn.apply(x.unsqueeze(2), modes, 1.0 / float(np.prod(spatial))).squeeze(2)
        spatial = tuple(x.shape[2:])
        nm = (0,) + tuple(self.num_modes) if len(self.num_modes) == 2 else self.num_modes   # 2-D model on a (B, C, 1, H, W) view
        modes = ops.clamp_modes(nm, spatial)
        return ops.DhtCropFn.apply(x, modes, 1.0 / float(np.prod(spatial)))


class PadInverse(nn.Module):
    """zero-pad the mode block to the full grid + unscaled transform (reference :413-494).
    `act` fuses the activation that follows in HNOXSBlock (reference :267-268)."""

    def __init__(self, ndim):
        super().__init__()
        assert ndim in (4, 5)

    def forward(self, x, spatial_shape, act=ops.ACT_NONE):
        if x.ndim == 4:   # 2-D (reference _call2d :437-452)
            assert all(s >= 2 * (zs // 2) for s, zs in zip(spatial_shape, x.shape[2:]))
            return ops.PadIdhtFn.apply(x.unsqueeze(2), (1,) + tuple(spatial_shape), 1.0, act).squeeze(2)
        assert all(s >= 2 * (zs // 2) for s, zs in zip(spatial_shape, x.shape[2:]))
        return ops.PadIdhtFn.apply(x, tuple(spatial_shape), 1.0, act)


class NeuralOperatorBlock(nn.Module):
    """One frequency-domain convolution: x <- act(op(x) [+ conv_branch(x)] + x) (reference :282-329)."""

    def __init__(self, in_channels, out_channels, num_modes, weights_type, ndim, activation, device,
                 use_conv_branch=False):
        super().__init__()
        self.op = HartleyOperator(in_channels, out_channels, num_modes, use_bias=False, weights_type=weights_type,
                                  use_transform=False, ndim=ndim, device=device)
        self.conv_branch = None
        if use_conv_branch:
            conv = nn.Conv2d if ndim == 4 else nn.Conv3d
            self.conv_branch = conv(in_channels, out_channels, kernel_size=1, bias=False, device=device)
        self.normalization = None
        if not _is_selu(activation):
            self.normalization = nn.GroupNorm(1, out_channels, device=device)
        self.activation = getattr(nn.functional, activation) if isinstance(activation, str) else activation

    def fusable(self):
        return (self.op.weights_type == 'shared' and self.conv_branch is None and self.normalization is None
                and self.op.in_channels == self.op.out_channels)

    def forward(self, x):
        act = ops.act_id(self.activation)
        if self.fusable():
            return ops.SpecMixFn.apply(x, 1, act, self.op.weight)
        y = self.op(x)
        if self.conv_branch is not None:
            y = ops.AddFn.apply(y, ops.PwConvFn.apply(x, None, self.conv_branch.weight, None, ops.ACT_NONE))
        y = ops.AddFn.apply(y, x)
        if self.normalization is not None:
            from .conv3d import group_norm_act
            return group_norm_act(y, self.normalization, act)
        return ops.ActFn.apply(y, act) if act != ops.ACT_NONE else y


class HNOXSBlock(nn.Module):
    """HNO-XS block with the block skip connection (reference :185-279)."""

    def __init__(self, num_convs, in_channels, out_channels, num_modes, weights_type='shared', ndim=5,
                 activation='selu', device=None, use_conv_branch=False, use_block_concat=True):
        super().__init__()
        cur = in_channels
        self.mapping_conv = None
        if cur != out_channels:
            self.mapping_conv = ConvNormAct(cur, out_channels, use_bias=True, activation=activation, ndim=ndim,
                                            device=device)
            cur = out_channels
        self.transform_crop = TransformCrop(num_modes, ndim)
        self.conv_blocks = nn.ModuleList()
        for _ in range(num_convs):
            self.conv_blocks.append(NeuralOperatorBlock(cur, out_channels, num_modes, weights_type, ndim, activation,
                                                        device, use_conv_branch))
            cur = out_channels
        self.pad_inverse = PadInverse(ndim)
        self.normalization = None
        if not _is_selu(activation):
            self.normalization = nn.GroupNorm(1, cur, device=device)
        self.activation = getattr(nn.functional, activation) if isinstance(activation, str) else activation
        self.conv_concat = None
        if use_block_concat:
            self.conv_concat = ConvNormAct(cur + out_channels, out_channels, use_bias=True, activation=activation,
                                           ndim=ndim, device=device)

    def _fused_ok(self):
        return (len(self.conv_blocks) > 0 and all(b.fusable() for b in self.conv_blocks) and self.normalization is None
                and self.conv_concat is not None and self.conv_concat.normalization is None
                and (self.mapping_conv is None or self.mapping_conv.normalization is None))

    def chain_ok(self, nxt, x):
        """conv_concat of this block and mapping_conv of `nxt` can run as one pass (ops.XSBlockFn with nmap_*): both blocks are the
        fused node, 24 -> 24 channels with a 48 -> 24 mapping, fp32 (not under autocast), same activation"""
        from .. import ops_bf16
        return (x.ndim == 5 and x.is_cuda and self._fused_ok() and nxt._fused_ok() and nxt.mapping_conv is not None
                and self.conv_concat.op.out_channels == 24 and tuple(nxt.mapping_conv.op.weight.shape[:2]) == (24, 48)
                and self.activation is nxt.activation and not ops_bf16.autocast_bf16()
                and os.environ.get('HNO_PW_CHAIN', '1') != '0')

    def forward(self, x, skip=None, passthrough=False, chain_to=None, next_skip=None, premapped=False):
        """`skip` is the U-Net skip tensor the reference concatenates in HNOSegXS.forward (:161-162);
        passing it separately lets the mapping conv read both tensors without a torch.cat.
        passthrough=True additionally returns the block INPUT (an alias) for use as a later block's skip, which
        routes that block's skip gradient through this block's backward (see ops.XSBlockFn).
        chain_to / next_skip (round 4): the next block and its skip tensor: this block returns the next block's MAPPED input
        (conv_concat and the next mapping_conv in one pass); that block is then called with premapped=True."""
        if x.ndim == 5 and self._fused_ok():
            # standard configuration: the whole block is one autograd node (ops.XSBlockFn)
            act = ops.act_id(self.activation)
            mc = self.mapping_conv.op if (self.mapping_conv is not None and not premapped) else None
            assert mc is not None or skip is None
            cc = self.conv_concat.op
            # chain_to: the next HNOXSBlock (its mapping_conv is chained) or the model's conv_out (nn.Conv3d: the logits are returned)
            nm = chain_to.mapping_conv.op if isinstance(chain_to, HNOXSBlock) else chain_to
            return ops.XSBlockFn.apply(x, skip, mc.weight if mc is not None else None, mc.bias if mc is not None else None,
                                       cc.weight, cc.bias, self.transform_crop.num_modes, act, passthrough,
                                       nm.weight if nm is not None else None, nm.bias if nm is not None else None,
                                       next_skip if isinstance(chain_to, HNOXSBlock) else None,
                                       *[b.op.weight for b in self.conv_blocks])
        assert chain_to is None and not premapped
        if passthrough:   # unfused configurations: plain autograd accumulates the two gradients of x
            return self.forward(x, skip), x
        if self.mapping_conv is not None:
            x = self.mapping_conv(x, skip)
        else:
            assert skip is None
        tmp = x
        spatial = tuple(x.shape[2:])
        z = self.transform_crop(x)
        if len(self.conv_blocks) and all(b.fusable() for b in self.conv_blocks):
            # all n_XS layers in one launch sequence: z <- act((W + I) z)
            z = ops.SpecMixFn.apply(z, 1, ops.act_id(self.conv_blocks[0].activation), *[b.op.weight for b in self.conv_blocks])
        else:
            for block in self.conv_blocks:
                z = block(z)
        act = ops.act_id(self.activation)
        if self.normalization is None:
            u = self.pad_inverse(z, spatial, act)
        else:
            from .conv3d import group_norm_act
            u = group_norm_act(self.pad_inverse(z, spatial), self.normalization, act)
        if self.conv_concat is not None:
            return self.conv_concat(u, tmp)
        return ops.AddFn.apply(u, tmp)


class HNOSegXS(nn.Module):
    """HNOSeg-XS (reference :20-182).  See the reference docstring for the arguments; they are
    identical here."""
    # a captured training step of this family MAY run the two halves of its batch as two concurrent passes (experiments.train_test.
    # SampleSplit): whether it does is measured per batch shape when the step is captured (train_test.choose_schedule: both forms are
    # captured and replayed, the faster is kept; by hand in round 4: +3-6 % at the BraTS training shape, -2 ... -10 % at 80^3 ... 160^3)
    hno_sample_split = 'measure'

    def __init__(self, in_channels, out_channels, filters, num_transform_blocks, num_modes, weights_type='shared',
                 use_resize=True, use_deep_supervision=False, use_unet_skip=True, use_block_concat=True,
                 activation='selu', output_activation: Union[str, callable] = 'softmax', ndim=5, device=None):
        super().__init__()
        self.in_channels, self.out_channels, self.filters = in_channels, out_channels, filters
        self.num_transform_blocks = num_transform_blocks
        self.num_modes, self.weights_type = num_modes, weights_type
        self.use_resize, self.use_deep_supervision = use_resize, use_deep_supervision
        self.use_unet_skip, self.use_block_concat = use_unet_skip, use_block_concat
        self.activation, self.output_activation = activation, output_activation
        self.ndim, self.device = ndim, device
        self.conv_in = self.conv1 = self.layers = self.conv_out = None
        assert self.ndim in (4, 5)
        if np.isscalar(self.num_transform_blocks):
            self.num_transform_blocks = [self.num_transform_blocks]
        self.block = partial(HNOXSBlock, num_modes=num_modes, weights_type=weights_type, ndim=ndim,
                             activation=activation, device=device, use_block_concat=use_block_concat)
        self.create_layers()

    def create_layers(self):
        ds_channels, enc_channels = [], {}
        cur, filters = self.in_channels, self.filters
        if self.use_resize:
            self.conv_in = ConvNormAct(cur, filters, kernel_size=2, stride=2, use_bias=True, activation=self.activation,
                                       ndim=self.ndim, device=self.device)
            cur = filters
        self.conv1 = ConvNormAct(cur, filters, use_bias=True, activation=self.activation, ndim=self.ndim,
                                 device=self.device)
        cur = filters
        if self.use_deep_supervision:
            ds_channels.append(cur)
        assert isinstance(self.num_transform_blocks, (list, tuple))
        self.layers = nn.ModuleList()
        nb = len(self.num_transform_blocks)
        for i, n_convs in enumerate(self.num_transform_blocks):
            if self.use_unet_skip and i > nb // 2:  # decoding block: takes the mirrored encoder output too
                cur += enc_channels[nb - 1 - i]
            self.layers.append(self.block(n_convs, cur, filters))
            cur = filters
            if self.use_deep_supervision:
                ds_channels.append(cur)
            if self.use_unet_skip and i < nb // 2:
                enc_channels[i] = cur
        if ds_channels:
            cur = sum(ds_channels)
        conv = nn.Conv2d if self.ndim == 4 else nn.Conv3d
        self.conv_out = conv(cur, self.out_channels, kernel_size=1, bias=False, device=self.device)
        self._softmax, self._out_act = ops.output_act(self.output_activation)
        if isinstance(self.output_activation, str):
            fn = getattr(nn.functional, self.output_activation)
            self.output_activation = partial(fn, dim=1) if self._softmax else fn
        if _is_selu(self.activation):
            self.apply(init_weights_for_snn)

    def forward(self, x):
        if x.ndim == 4:
            # 2-D model (ndim = 4): the same kernels on a (B, C, 1, H, W) view.  A size-1 axis is transformed by a copy, the
            # 2 x 2 / stride 2 / padding 1 stem is the 3-D kernel with its taps at kd = 1, trilinear resampling with depth
            # 1 -> 1 is bilinear, and every other layer is pointwise.
            return self.forward(x.unsqueeze(2)).squeeze(2)
        image_size = tuple(x.shape[2:])
        # channel-padded activations (ops.channel_padded): the stem hands the blocks tensors whose channel stride is rounded up to
        # 128 B when the working grid (65^3 for 128^3 images) is one both plane transforms serve and every block is the fused node
        grid = tuple(v // 2 + 1 for v in image_size) if self.use_resize else image_size
        pad = (x.is_cuda and self.use_resize and self.conv_in.normalization is None and self.conv1.normalization is None
               and all(l._fused_ok() for l in self.layers) and ops.padded_ok(grid, _tuple_modes(self.num_modes, self.ndim)))
        with ops.channel_padded(pad):
            return self._forward5(x, image_size)

    def _forward5(self, x, image_size):
        ds, enc = [], {}
        if self._stem_chain_ok(x):
            # conv_in and conv1 as one pass (ops.StemChainFn): conv_in's output is not written, the backward recomputes it
            x = ops.StemChainFn.apply(x, self.conv_in.op.weight, self.conv_in.op.bias, self.conv1.op.weight, self.conv1.op.bias,
                                      ops.act_id(self.conv_in.activation))
        else:
            if self.use_resize:
                x = self.conv_in(x)
            x = self.conv1(x)
        if self.use_deep_supervision:
            ds.append(x)
        nb = len(self.num_transform_blocks)
        premapped = False
        for i, layer in enumerate(self.layers):
            skip = enc[nb - 1 - i] if (self.use_unet_skip and i > nb // 2) else None
            # the output of block i - 1 (this block's input) is a later block's skip: take the alias from this block
            feeds_skip = self.use_unet_skip and 0 <= i - 1 < nb // 2 and (nb - 1 - (i - 1)) > nb // 2
            # decoder blocks: the next block's mapping_conv over cat[this block's output, its U-Net skip] is chained to this block's
            # conv_concat (one pass over the activations instead of two) when nobody else reads this block's output
            nxt = self.layers[i + 1] if i + 1 < nb else None
            nskip = enc.get(nb - 1 - (i + 1)) if (nxt is not None and self.use_unet_skip and i + 1 > nb // 2) else None
            chain = (nxt is not None and nskip is not None and not self.use_deep_supervision and not (self.use_unet_skip and i < nb // 2)
                     and not (feeds_skip and torch.is_grad_enabled()) and layer.chain_ok(nxt, x))
            if premapped:
                skip = None
            if feeds_skip and torch.is_grad_enabled():
                assert not chain
                x, enc[i - 1] = layer(x, skip, passthrough=True, premapped=premapped)
            elif chain:
                x = layer(x, skip, chain_to=nxt, next_skip=nskip, premapped=premapped)
            elif i == nb - 1 and nb > 1 and self._head_chain_ok(layer, x):
                # the last block hands back the low-resolution logits: its conv_concat and conv_out run as one pass
                logits = layer(x, skip, chain_to=self.conv_out, premapped=premapped)
                return self._head_logits(logits, image_size)
            else:
                x = layer(x, skip, premapped=premapped)
            premapped = chain
            if self.use_deep_supervision:
                ds.append(x)
            if self.use_unet_skip and i < nb // 2:
                enc[i] = x
        return self._head(ds if ds else [x], image_size)

    def _stem_chain_ok(self, x):
        from .. import ops_bf16
        hooked = any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in (self.conv_in, self.conv1))
        return (self.use_resize and not hooked      # (the two modules' own forward() is bypassed: hooks on them keep the layers apart)
                and x.ndim == 5 and x.is_cuda and self.conv_in.normalization is None and self.conv1.normalization is None
                and self.conv_in.activation is self.conv1.activation and not ops_bf16.autocast_bf16()
                and os.environ.get('HNO_STEM_CHAIN', '1') != '0'
                and ops.StemChainFn.supported(x, self.conv_in.op.weight, self.conv1.op.weight))

    def _head_chain_ok(self, layer, x):
        """the last block's conv_concat and conv_out (24 -> 4, no bias) can run as one pass (ops.XSBlockFn with the conv_out weight)"""
        from .. import ops_bf16
        return (x.ndim == 5 and x.is_cuda and layer._fused_ok() and not self.use_deep_supervision and self.conv_out.bias is None
                and layer.conv_concat.op.out_channels == 24 and tuple(self.conv_out.weight.shape[:2]) == (4, 24)
                and not ops_bf16.autocast_bf16() and os.environ.get('HNO_PW_CHAIN', '1') != '0'
                and os.environ.get('HNO_PW_CHAIN_HEAD', '1') != '0')

    def _head(self, feats, image_size):
        """conv_out at LOW resolution (it commutes with the per-channel trilinear interpolation),
        then fused upsample + softmax (reference :171-180)."""
        w = self.conv_out.weight
        if len(feats) == 1:
            logits = ops.PwConvFn.apply(feats[0], None, w, None, ops.ACT_NONE)
        else:  # deep supervision: conv over the channel concat = sum of per-tensor convs
            logits, c0 = None, 0
            for f in feats:
                part = ops.PwConvFn.apply(f, None, w[:, c0:c0 + f.shape[1]].contiguous(), None, ops.ACT_NONE)
                logits = part if logits is None else ops.AddFn.apply(logits, part)
                c0 += f.shape[1]
        return self._head_logits(logits, image_size)

    def _head_logits(self, logits, image_size):
        if self.use_resize:
            y = ops.head_output(logits, image_size, self._softmax, self._out_act)
        else:
            y = ops.head_output(logits, tuple(logits.shape[2:]), self._softmax, self._out_act)
        return spatial_padcrop(y, image_size)
