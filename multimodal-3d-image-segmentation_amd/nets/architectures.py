"""FNO / FNOSeg / HNOSeg (NeuralOperatorSeg), HartleyMHASeg and V-Net-DS on the HIP kernels
(reference nets/architectures.py:26-653).  Same constructors, module tree and state-dict keys."""
from functools import partial
from typing import Union

import numpy as np
import torch
from torch import nn

from .. import ops
from .fourier_operator import FourierOperator
from .hartley_operator import HartleyOperator
from .hartley_mha import HartleyMultiHeadAttention
from .nets_utils import ConvNormAct, ConvTransposeNormAct, init_weights_for_snn, spatial_padcrop, _is_selu

import os
_NO_BLOCK_FUSION = bool(int(os.environ.get('HNO_NO_BLOCK_FUSION', '0')))   # A/B switch for the fused FNOSeg / HNOSeg block


class _TransBlock(nn.Module):
    """x1 = op(x); x2 = conv_branch(x); x = act(norm(x1 + x2)); block skip (reference :511-548)."""

    def __init__(self):
        super().__init__()
        self.op = self.conv_branch = self.normalization = self.activation = None
        self.use_block_skip = None
        self.conv_concat = None

    def _fusable(self, act):
        op = self.op
        return (not _NO_BLOCK_FUSION and isinstance(op, (FourierOperator, HartleyOperator)) and self.normalization is None
                and act != ops.ACT_NONE and self.use_block_skip and self.conv_concat is not None
                and op.weights_type == 'shared' and op.use_transform and not op.use_bias
                and self.conv_concat.normalization is None and ops.act_id(self.conv_concat.activation) == act)

    def _fused_block(self, x, act):
        """The whole block as one autograd node (ops.NOBlockFn) when it has the FNOSeg / HNOSeg shape: shared-weight
        Fourier / Hartley operator with transform and without bias, SELU (no GroupNorm), concat skip."""
        op = self.op
        if x.ndim != 5 or not self._fusable(act):
            return None
        fourier = isinstance(op, FourierOperator)
        op_ws = (op.weight_real, op.weight_imag) if fourier else (op.weight,)
        w2 = getattr(self, '_w2_pre', None)          # composed by the model for all its blocks at once (_TransSeg._forward5)
        if fourier and w2 is not None:
            op_ws = op_ws + (w2,)
        br = self.conv_branch
        cc = self.conv_concat.op
        return ops.NOBlockFn.apply(x, fourier, tuple(op.num_modes), act, None if br is None else br.weight,
                                   None if br is None else br.bias, cc.weight, cc.bias, *op_ws)

    def forward(self, x):
        act = ops.act_id(self.activation)
        fused = self._fused_block(x, act)
        if fused is not None:
            return fused
        fuse_act = act if self.normalization is None else ops.ACT_NONE
        x2 = None
        if self.conv_branch is not None:
            x2 = ops.PwConvFn.apply(x, None, self.conv_branch.weight, self.conv_branch.bias, ops.ACT_NONE)
        assert self.op is not None or x2 is not None
        if self.op is not None:
            # the operator adds the conv branch and applies the activation on store
            y = self.op.forward_fused(x, addend=x2, act=fuse_act)
        else:
            y = ops.ActFn.apply(x2, fuse_act) if fuse_act != ops.ACT_NONE else x2
        if self.normalization is not None:
            from .conv3d import group_norm_act
            y = group_norm_act(y, self.normalization, act)
        if self.use_block_skip:
            if self.conv_concat is not None:
                return self.conv_concat(y, x)
            return ops.AddFn.apply(y, x)
        return y


class NeuralOperatorBlock(_TransBlock):
    """The FNO / HNO block (reference nets/architectures.py:551-608): spectral operator + 1x1x1 conv
    branch -> activation -> block skip (concat conv, add, or none)."""

    def __init__(self, in_channels, out_channels, num_modes, transform_type, weights_type='shared', ndim=5,
                 activation='selu', device=None, use_conv_branch=True, use_bias_conv_branch=False, use_block_skip=True,
                 use_block_concat=True):
        super().__init__()
        assert transform_type in ('Fourier', 'Hartley')
        self.use_block_skip = use_block_skip
        op = FourierOperator if transform_type == 'Fourier' else HartleyOperator
        self.op = op(in_channels, out_channels, num_modes, use_bias=False, weights_type=weights_type, ndim=ndim,
                     device=device)
        if use_conv_branch:
            conv = nn.Conv2d if ndim == 4 else nn.Conv3d
            self.conv_branch = conv(in_channels, out_channels, kernel_size=1, bias=use_bias_conv_branch, device=device)
        if not _is_selu(activation):
            self.normalization = nn.GroupNorm(1, out_channels, device=device)
        self.activation = getattr(nn.functional, activation) if isinstance(activation, str) else activation
        if self.use_block_skip and use_block_concat:
            self.conv_concat = ConvNormAct(in_channels + out_channels, out_channels, use_bias=True, activation=activation,
                                           ndim=ndim, device=device)


class _TransSeg(nn.Module):
    """Common body of NeuralOperatorSeg and HartleyMHASeg (reference nets/architectures.py:255-353)."""

    def __init__(self):
        super().__init__()
        self.in_channels = self.out_channels = self.filters = self.num_transform_blocks = None
        self.use_resize = self.use_deep_supervision = self.activation = self.output_activation = None
        self.ndim = self.device = self.block = None
        self.conv_in = self.conv1 = self.layers = self.conv_out = self.conv_ds = None

    def create_layers(self):
        ds_channels = []
        cur = self.in_channels
        if self.use_resize:
            self.conv_in = ConvNormAct(cur, self.filters, kernel_size=2, stride=2, use_bias=True,
                                       activation=self.activation, ndim=self.ndim, device=self.device)
            cur = self.filters
        self.conv1 = ConvNormAct(cur, self.filters, use_bias=True, activation=self.activation, ndim=self.ndim,
                                 device=self.device)
        cur = self.filters
        if self.use_deep_supervision:
            ds_channels.append(cur)
        self.layers = nn.ModuleList()
        for _ in range(self.num_transform_blocks):
            self.layers.append(self.block(cur, self.filters))
            cur = self.filters
            if self.use_deep_supervision:
                ds_channels.append(cur)
        if ds_channels:
            cur = sum(ds_channels)
            self.conv_ds = ConvNormAct(cur, self.out_channels, use_bias=True, activation=self.activation, ndim=self.ndim,
                                       device=self.device)
            cur = self.out_channels
        conv = nn.Conv2d if self.ndim == 4 else nn.Conv3d
        self.conv_out = conv(cur, self.out_channels, kernel_size=1, bias=False, device=self.device)
        self._softmax, self._out_act = ops.output_act(self.output_activation)
        if isinstance(self.output_activation, str):
            fn = getattr(nn.functional, self.output_activation)
            self.output_activation = partial(fn, dim=1) if self._softmax else fn
        if _is_selu(self.activation):
            self.apply(init_weights_for_snn)

    def forward(self, x):
        if x.ndim == 4:   # 2-D model (ndim = 4): the same kernels on a (B, C, 1, H, W) view (see HNOSegXS.forward)
            return self.forward(x.unsqueeze(2)).squeeze(2)
        image_size = tuple(x.shape[2:])
        # channel-padded activations (ops.channel_padded, see HNOSegXS.forward): every block the fused node, both plane transforms
        # serve the working grid
        grid = tuple(v // 2 + 1 for v in image_size) if self.use_resize else image_size
        blocks = list(self.layers)
        pad = (x.is_cuda and self.use_resize and not self.use_deep_supervision and self.conv_in.normalization is None
               and self.conv1.normalization is None and len(blocks) > 0
               and all(isinstance(l, _TransBlock) and l._fusable(ops.act_id(l.activation)) for l in blocks)
               and len(blocks[0].op.num_modes) == 3 and ops.padded_ok(grid, tuple(blocks[0].op.num_modes)))
        with ops.channel_padded(pad):
            return self._forward5(x, image_size)

    def _forward5(self, x, image_size):
        tensors = []
        if self.use_resize:
            x = self.conv_in(x)
        x = self.conv1(x)
        if self.use_deep_supervision:
            tensors.append(x)
        # the complex weights of all fusable Fourier blocks in their composed real form, ONE launch per forward pass (round 5: each block
        # ran its own hno_cmix_compose: 24 launches of ~4 us in FNOSeg's forward chain)
        fblocks = [l for l in self.layers if isinstance(l, _TransBlock) and isinstance(l.op, FourierOperator) and x.ndim == 5
                   and l._fusable(ops.act_id(l.activation))]
        w2_all = None
        if len(fblocks) > 1 and x.is_cuda:
            with torch.no_grad():
                w2_all = ops.cmix_compose_all([(l.op.weight_real, l.op.weight_imag) for l in fblocks])
        # torch.autocast(bfloat16): conv1's output and every block's input / output are bf16 tensors in the reference
        # (experiments/train_test.py:154-160).  When every block can take and return bf16 activations (ops.noblock_io16_ok) the chain keeps
        # them bf16 IN MEMORY, with one cast at each end; otherwise storage stays fp32 and only the convolutions' arithmetic is bf16.
        io16 = (x.is_cuda and not self.use_deep_supervision and len(fblocks) == len(self.layers) > 0 and ops._autocast_bf16()
                and ops.chan_stride(x) is not None
                and all(l.conv_branch is not None and
                        ops.noblock_io16_ok(tuple(x.shape[2:]), x.shape[1], True, tuple(l.op.num_modes), l.conv_branch.weight.shape,
                                            l.conv_concat.op.weight.shape) for l in fblocks))
        if io16:
            x = ops.CastFn.apply(x, True)
        try:
            if w2_all is not None:
                for l, w2 in zip(fblocks, w2_all.unbind(0)):
                    l._w2_pre = w2
            for layer in self.layers:
                x = layer(x)
                if self.use_deep_supervision:
                    tensors.append(x)
        finally:
            for l in fblocks:
                l._w2_pre = None
        if io16:
            x = ops.CastFn.apply(x, False)
        if tensors:
            from .deep_supervision import conv_over_concat
            x = conv_over_concat(self.conv_ds, tensors)
        # conv_out commutes with the per-channel trilinear interpolation: run it at low resolution
        logits = ops.PwConvFn.apply(x, None, self.conv_out.weight, None, ops.ACT_NONE)
        y = ops.head_output(logits, image_size if self.use_resize else tuple(logits.shape[2:]), self._softmax, self._out_act)
        return spatial_padcrop(y, image_size)


class NeuralOperatorSeg(_TransSeg):
    """FNO / FNOSeg / HNOSeg by arguments (reference nets/architectures.py:356-429); same constructor."""

    def __init__(self, in_channels, out_channels, filters, num_transform_blocks, num_modes, transform_type,
                 weights_type='shared', use_resize=True, use_deep_supervision=False, use_bias_conv_branch=False,
                 use_block_skip=True, use_block_concat=True, activation='selu',
                 output_activation: Union[str, callable] = 'softmax', ndim=5, device=None):
        super().__init__()
        self.in_channels, self.out_channels, self.filters = in_channels, out_channels, filters
        self.num_transform_blocks, self.num_modes = num_transform_blocks, num_modes
        self.transform_type, self.weights_type = transform_type, weights_type
        self.use_resize, self.use_deep_supervision = use_resize, use_deep_supervision
        self.use_bias_conv_branch, self.use_block_skip, self.use_block_concat = use_bias_conv_branch, use_block_skip, use_block_concat
        self.activation, self.output_activation = activation, output_activation
        self.ndim, self.device = ndim, device
        assert self.transform_type in ('Fourier', 'Hartley')
        assert self.ndim in (4, 5)
        self.block = partial(NeuralOperatorBlock, num_modes=num_modes, transform_type=transform_type,
                             weights_type=weights_type, ndim=ndim, activation=activation, device=device,
                             use_bias_conv_branch=use_bias_conv_branch, use_block_skip=use_block_skip,
                             use_block_concat=use_block_concat)
        self.create_layers()


class HartleyMHABlock(_TransBlock):
    """Hartley-MHA block (reference nets/architectures.py:611-635)."""

    def __init__(self, in_channels, key_dim, num_heads, num_modes, patch_size, attention_activation, ndim, activation,
                 device, use_conv_branch=True, use_bias_conv_branch=False, use_block_skip=True, use_block_concat=True):
        super().__init__()
        self.use_block_skip = use_block_skip
        self.op = HartleyMultiHeadAttention(in_channels, key_dim, num_heads, num_modes, patch_size, attention_activation,
                                            ndim=ndim, device=device)
        if use_conv_branch:
            conv = nn.Conv2d if ndim == 4 else nn.Conv3d
            self.conv_branch = conv(in_channels, key_dim, kernel_size=1, bias=use_bias_conv_branch, device=device)
        if not _is_selu(activation):
            self.normalization = nn.GroupNorm(1, key_dim, device=device)
        self.activation = getattr(nn.functional, activation) if isinstance(activation, str) else activation
        if self.use_block_skip and use_block_concat:
            self.conv_concat = ConvNormAct(in_channels + key_dim, key_dim, use_bias=True, activation=activation, ndim=ndim,
                                           device=device)


class HartleyMHASeg(_TransSeg):
    """HartleyMHA architecture (reference nets/architectures.py:432-508); same constructor."""

    def __init__(self, in_channels, out_channels, filters, num_transform_blocks, num_heads, num_modes, patch_size,
                 attention_activation='selu', use_resize=True, use_deep_supervision=True, use_bias_conv_branch=False,
                 use_block_skip=True, use_block_concat=True, activation='selu',
                 output_activation: Union[str, callable] = 'softmax', ndim=5, device=None):
        super().__init__()
        self.in_channels, self.out_channels, self.filters = in_channels, out_channels, filters
        self.num_transform_blocks, self.num_heads = num_transform_blocks, num_heads
        self.num_modes, self.patch_size, self.attention_activation = num_modes, patch_size, attention_activation
        self.use_resize, self.use_deep_supervision = use_resize, use_deep_supervision
        self.use_bias_conv_branch, self.use_block_skip, self.use_block_concat = use_bias_conv_branch, use_block_skip, use_block_concat
        self.activation, self.output_activation = activation, output_activation
        self.ndim, self.device = ndim, device
        assert self.ndim in (4, 5)
        self.block = partial(HartleyMHABlock, num_heads=num_heads, num_modes=num_modes, patch_size=patch_size,
                             attention_activation=attention_activation, ndim=ndim, activation=activation, device=device,
                             use_bias_conv_branch=use_bias_conv_branch, use_block_skip=use_block_skip,
                             use_block_concat=use_block_concat)
        self.create_layers()


def upsampling(tensors):
    """Nearest-neighbour upsampling of a dict of tensors to the size of tensors[0]
    (reference nets/architectures.py:638-653)."""
    ref_size = tuple(tensors[0].shape[2:])
    return [t if tuple(t.shape[2:]) == ref_size else ops.NearestUpFn.apply(t, ref_size) for t in tensors.values()]


class VNetDS(nn.Module):
    """V-Net with deep supervision (reference nets/architectures.py:26-252); same constructor, module tree
    and state-dict keys.  3x3x3 convolutions run as implicit GEMMs on the fp32 matrix cores, every
    GroupNorm(1, C) + ELU is one fused pass, and the right-leg deep supervision applies conv_ds to each
    leg at its own resolution before the nearest-neighbour upsampling (the two commute), so the
    744-channel concatenated tensor of the reference is never built."""

    def __init__(self, in_channels, out_channels, base_num_filters, num_blocks, use_resize=True, right_leg_indexes=None,
                 kernel_size=3, activation='elu', use_snn=False, output_activation='softmax', use_residual=True, ndim=5,
                 device=None):
        super().__init__()
        assert isinstance(num_blocks, (list, tuple))
        self.in_channels, self.out_channels, self.num_blocks = in_channels, out_channels, num_blocks
        self.base_num_filters = base_num_filters
        self.kernel_size = kernel_size
        self.use_resize, self.right_leg_indexes = use_resize, right_leg_indexes
        self.output_activation, self.use_residual, self.ndim = output_activation, use_residual, ndim
        if self.right_leg_indexes is None:
            self.right_leg_indexes = [0]
        assert self.ndim in (4, 5)
        nsec = len(self.num_blocks)
        conv = partial(ConvNormAct, stride=1, use_bias=True, activation=activation, use_snn=use_snn, ndim=ndim, device=device)
        cur = in_channels
        self.conv_in = None
        if self.use_resize:
            self.conv_in = ConvNormAct(cur, base_num_filters, kernel_size=2, stride=2, use_bias=True, activation=activation,
                                       use_snn=use_snn, ndim=ndim, device=device)
            cur = base_num_filters
        enc_ch, leg_ch = {}, {}
        self.encode_layers = nn.ModuleDict()
        for i in range(nsec):
            layers = nn.ModuleList()
            filters = base_num_filters * (2 ** i)
            tmp_in = cur if self.use_residual else None
            for _ in range(self.num_blocks[i]):
                layers.append(conv(cur, filters, kernel_size=kernel_size))
                cur = filters
            if self.use_residual:
                layers.append(conv(tmp_in, filters, kernel_size=1))
                cur = filters
            if i != nsec - 1:
                enc_ch[i] = filters
                layers.append(ConvNormAct(cur, filters, kernel_size=kernel_size, stride=2, use_bias=True, activation=activation,
                                          use_snn=use_snn, ndim=ndim, device=device))
                cur = filters
            elif i in self.right_leg_indexes:
                leg_ch[i] = cur
            self.encode_layers[str(i)] = layers
        self.decode_layers = nn.ModuleDict()
        for i in reversed(range(nsec - 1)):
            layers = nn.ModuleList()
            filters = base_num_filters * (2 ** i)
            layers.append(ConvTransposeNormAct(cur, filters, kernel_size=kernel_size, use_bias=True, activation=activation,
                                               ndim=ndim, device=device))
            cur = filters + enc_ch[i]
            tmp_in = cur if self.use_residual else None
            for _ in range(self.num_blocks[i]):
                layers.append(conv(cur, filters, kernel_size=kernel_size))
                cur = filters
            if self.use_residual:
                layers.append(conv(tmp_in, filters, kernel_size=1))
                cur = filters
            if i in self.right_leg_indexes:
                leg_ch[i] = cur
            self.decode_layers[str(i)] = layers
        self.conv_ds = None
        self._leg_channels = dict(leg_ch)
        if len(leg_ch) == 1:
            cur = leg_ch[0]
        else:
            cur = sum(leg_ch.values())
            self.conv_ds = ConvNormAct(cur, self.out_channels, use_bias=True, activation=activation, use_snn=use_snn,
                                       ndim=ndim, device=device)
            cur = self.out_channels
        convc = nn.Conv2d if ndim == 4 else nn.Conv3d
        self.conv_out = convc(cur, self.out_channels, kernel_size=1, bias=False, device=device)
        self._softmax, self._out_act = ops.output_act(self.output_activation)
        if isinstance(self.output_activation, str):
            fn = getattr(nn.functional, self.output_activation)
            self.output_activation = partial(fn, dim=1) if self._softmax else fn
        self.encode_tensors = None
        self.right_leg = None
        if use_snn and _is_selu(activation):
            self.apply(init_weights_for_snn)

    def forward(self, x):
        from .. import ops_bf16
        # bf16 matrix-core path under autocast when every hidden channel count (base_num_filters 2^k) is a multiple of 8 -- the bf16
        # kernels' fragment width; other widths run the fp32 kernels (still on the GPU: autocast only ever lowers precision)
        # (a 2-D model -- ndim = 4, Conv2d containers -- and kernel sizes other than 3 keep the fp32 kernels under autocast: round 6)
        if ops_bf16.autocast_bf16() and self.base_num_filters % 8 == 0 and self.ndim == 5 and x.ndim == 5 and self.kernel_size == 3:
            return self._forward_bf16(x)
        if x.ndim == 4:   # 2-D model (ndim = 4): the same kernels on a (B, C, 1, H, W) view (see nets/conv3d.py for the 3x3 layers)
            return self.forward(x.unsqueeze(2)).squeeze(2)
        image_size = tuple(x.shape[2:])
        self.encode_tensors, self.right_leg = {}, {}
        if self.use_resize:
            x = self.conv_in(x)
        x = self.decode(self.encode(x))
        logits = ops.PwConvFn.apply(x, None, self.conv_out.weight, None, ops.ACT_NONE)   # commutes with the upsampling
        y = ops.head_output(logits, image_size if self.use_resize else tuple(logits.shape[2:]), self._softmax, self._out_act)
        return spatial_padcrop(y, image_size)

    # ---- bf16 matrix-core path (torch.autocast(bfloat16); reference train_test.py:154-160) -------------------------------
    def _forward_bf16(self, x):
        """The same network on channels-last bf16 activations: every 3x3x3 / 2x2x2 / 1x1x1 convolution and transposed
        convolution is a bf16 MFMA gather GEMM with fp32 accumulation (hno_cb_conv), GroupNorm statistics come out of the
        convolution's epilogue, GroupNorm + ELU and the section's residual sum are one pass, the decoder's concat is fused
        into the two-input convolutions.  The deep-supervision legs leave the bf16 body through their (C -> out_channels)
        1x1x1 convolution; upsampling, the leg sum, conv_ds' GroupNorm, conv_out, trilinear + softmax and the loss stay fp32."""
        from .. import ops_bf16 as ob
        # bf16 GEMM operands of every convolution, packed by one launch (the table is rebuilt if the parameters moved)
        if x.is_meta:
            return self._forward_bf16_body(x, ob)
        ops_ = [m.op for m in self.modules() if hasattr(m, 'op') and isinstance(m.op, (nn.Conv3d, nn.ConvTranspose3d))]
        pk = getattr(self, '_packed', None)
        if pk is None or pk.key != tuple(o.weight.data_ptr() for o in ops_ if o.weight.data_ptr() in pk.entries) or \
                any(o.weight.data_ptr() not in pk.entries for o in ops_ if o.weight.shape[0] % 8 == 0 and o.weight.shape[1] % 8 == 0):
            pk = ob.PackedWeights(ops_)
            object.__setattr__(self, '_packed', pk)
        pk.refresh()
        try:
            return self._forward_bf16_body(x, ob)
        finally:
            ob.PackedWeights.release()

    def _forward_bf16_body(self, x, ob):
        image_size = tuple(x.shape[2:])
        nsec = len(self.num_blocks)
        h = ob.PackInputFn.apply(x, (x.shape[1] + 7) // 8 * 8)
        if self.use_resize:
            h = ob.conv_norm_act(self.conv_in, h)
        enc, legs = {}, {}

        def section(layers, xa, xb, nconv):
            it = iter(layers)
            convs = [next(it) for _ in range(nconv)]
            res = next(it) if self.use_residual else None
            cur_a, cur_b = xa, xb
            for k, layer in enumerate(convs):
                last = k == nconv - 1
                if last and res is not None:
                    cur_a = ob.conv_norm_act(layer, cur_a, cur_b, residual=ob.conv_norm_act(res, xa, xb, residual='defer'))
                else:
                    cur_a = ob.conv_norm_act(layer, cur_a, cur_b)
                cur_b = None
            return cur_a, it
        for i in range(nsec):
            h, it = section(self.encode_layers[str(i)], h, None, self.num_blocks[i])
            if i != nsec - 1:
                enc[i] = h
                h = ob.conv_norm_act(next(it), h)           # strided down-convolution
            elif i in self.right_leg_indexes:
                legs[i] = h
        for i in reversed(range(nsec - 1)):
            layers = self.decode_layers[str(i)]
            up = ob.conv_norm_act(layers[0], h)             # transposed convolution (+ GroupNorm + act on its full output)
            d, hh, w = enc[i].shape[1:4]
            assert all(a >= b for a, b in zip(up.shape[1:4], (d, hh, w)))
            lo = [(a - b) // 2 for a, b in zip(up.shape[1:4], (d, hh, w))]      # spatial_padcrop: the extra element goes high
            if tuple(up.shape[1:4]) != (d, hh, w):
                up = ob.CropHighFn.apply(up, lo, (d, hh, w)) if (not up.is_meta and os.environ.get('HNO_VNET_GLUE', '1') != '0') else up[:, lo[0]:lo[0] + d, lo[1]:lo[1] + hh, lo[2]:lo[2] + w, :].contiguous()
            h, _ = section(list(layers)[1:], up, enc[i], self.num_blocks[i])    # the concat is fused into the convolutions
            if i in self.right_leg_indexes:
                legs[i] = h
        if len(legs) == 1:
            feat = legs[0]
            logits = ob.pointwise_to_f32(feat, self.conv_out.weight, None, self.out_channels)
        else:
            op = self.conv_ds.op
            w = op.weight.reshape(op.weight.shape[0], -1)
            ref_size = tuple(legs[0].shape[1:4])
            acc, c0 = None, 0
            wlegs = ob.LegWeightsFn.apply(w, tuple(t.shape[4] for t in legs.values())) if (not x.is_meta and os.environ.get('HNO_VNET_GLUE', '1') != '0') else None
            for idx, (key, t) in enumerate(legs.items()):   # insertion order = concat order of the reference
                c = t.shape[4]
                if wlegs is not None:
                    part = ob.pointwise_to_f32(t, wlegs[idx], op.bias if idx == 0 else None, w.shape[0], padded=True)
                else:
                    part = ob.pointwise_to_f32(t, w[:, c0:c0 + c], op.bias if idx == 0 else None, w.shape[0])
                if tuple(part.shape[2:]) != ref_size:
                    part = ops.NearestUpFn.apply(part, ref_size)
                acc = part if acc is None else ops.AddFn.apply(acc, part)
                c0 += c
            act = ops.act_id(self.conv_ds.activation)
            if self.conv_ds.normalization is not None:
                from .conv3d import group_norm_act
                acc = group_norm_act(acc, self.conv_ds.normalization, act)
            elif act != ops.ACT_NONE:
                acc = ops.ActFn.apply(acc, act)
            logits = ops.PwConvFn.apply(acc, None, self.conv_out.weight, None, ops.ACT_NONE)
        y = ops.head_output(logits, image_size if self.use_resize else tuple(logits.shape[2:]), self._softmax, self._out_act)
        return spatial_padcrop(y, image_size)

    def _section(self, layers, x, nconv):
        it = iter(layers)
        tmp = x if self.use_residual else None
        for _ in range(nconv):
            x = next(it)(x)
        if tmp is not None:
            x = ops.AddFn.apply(x, next(it)(tmp))
        return x, it

    def encode(self, x):
        nsec = len(self.num_blocks)
        for i in range(nsec):
            x, it = self._section(self.encode_layers[str(i)], x, self.num_blocks[i])
            if i != nsec - 1:
                self.encode_tensors[i] = x
                x = next(it)(x)  # strided down-convolution
            elif i in self.right_leg_indexes:
                self.right_leg[i] = x
        return x

    def decode(self, x):
        nsec = len(self.num_blocks)
        for i in reversed(range(nsec - 1)):
            layers = self.decode_layers[str(i)]
            x = layers[0](x)                                            # transposed convolution
            x = spatial_padcrop(x, tuple(self.encode_tensors[i].shape[2:]))
            x = torch.cat([x.contiguous(), self.encode_tensors[i]], dim=1)
            x, _ = self._section(list(layers)[1:], x, self.num_blocks[i])
            if i in self.right_leg_indexes:
                self.right_leg[i] = x
        if len(self.right_leg) == 1:
            return self.right_leg[0]
        return self._deep_supervision()

    def _deep_supervision(self):
        """conv_ds(cat(upsampling(legs))) = act(norm(sum_legs up(W_leg . leg) + b)): the 1x1x1 conv is applied per
        leg at its own resolution, only out_channels channels are upsampled and summed."""
        op = self.conv_ds.op
        w = op.weight.reshape(op.weight.shape[0], -1)
        ref_size = tuple(self.right_leg[0].shape[2:])
        acc, c0 = None, 0
        for idx, (key, t) in enumerate(self.right_leg.items()):        # insertion order = concat order of the reference
            c = t.shape[1]
            part = ops.PwConvFn.apply(t, None, w[:, c0:c0 + c].contiguous(), op.bias if idx == 0 else None, ops.ACT_NONE)
            if tuple(part.shape[2:]) != ref_size:
                part = ops.NearestUpFn.apply(part, ref_size)
            acc = part if acc is None else ops.AddFn.apply(acc, part)
            c0 += c
        act = ops.act_id(self.conv_ds.activation)
        if self.conv_ds.normalization is not None:
            from .conv3d import group_norm_act
            return group_norm_act(acc, self.conv_ds.normalization, act)
        return ops.ActFn.apply(acc, act) if act != ops.ACT_NONE else acc
