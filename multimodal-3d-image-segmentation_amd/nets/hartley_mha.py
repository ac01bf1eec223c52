"""Hartley multi-head attention on the HIP kernels (reference nets/hartley_mha.py:18-524).

Frequency-domain self-attention: Hartley transform + mode truncation (hno_dht3_crop), per-head
query/key/value projections (pointwise MFMA convs over the mode axis), patch grouping (pure index
permutation), att = act(Q^T K / sqrt(C)) and out = V att^T on the batched fp32 MFMA GEMM (hno_bmm),
output projection, zero pad + unscaled inverse transform (hno_pad_idht3).  SELU, not softmax, is the
default attention activation, so no online normalisation is needed.
"""
import math
import os
from typing import Union

import numpy as np
import torch
from torch.nn import Module, Parameter, init

from .. import ops


def grouping3d(x, patch_size):
    """(B,Z,C,D,H,W) -> (B,Z,C*pd*ph*pw, D/pd, H/ph, W/pw): each new voxel is a patch of the old ones
    (reference :473-498).  Index permutation only."""
    assert len(patch_size) == 3
    pd, ph, pw = patch_size
    b, z, c, d, h, w = x.shape
    assert d % pd == 0 and h % ph == 0 and w % pw == 0
    nd_, nh, nw = d // pd, h // ph, w // pw
    x = x.reshape(b, z, c, nd_, pd, nh, ph, nw, pw).permute(0, 1, 2, 4, 6, 8, 3, 5, 7)
    return x.reshape(b, z, c * pd * ph * pw, nd_, nh, nw)


def ungrouping3d(x, num_channels, patch_size):
    """Inverse of grouping3d (reference :501-524)."""
    pd, ph, pw = patch_size
    b, z, _, nd_, nh, nw = x.shape
    c = num_channels
    x = x.reshape(b, z, c, pd, ph, pw, nd_, nh, nw).permute(0, 1, 2, 6, 3, 7, 4, 8, 5)
    return x.reshape(b, z, c, nd_ * pd, nh * ph, nw * pw)


def grouping2d(x, patch_size):
    """(B,Z,C,H,W) -> (B,Z,C*ph*pw, H/ph, W/pw) (reference :421-445).  Index permutation only."""
    assert len(patch_size) == 2
    return grouping3d(x.unsqueeze(3), (1,) + tuple(patch_size)).squeeze(3)


def ungrouping2d(x, num_channels, patch_size):
    """Inverse of grouping2d (reference :448-470)."""
    assert len(patch_size) == 2
    return ungrouping3d(x.unsqueeze(3), num_channels, (1,) + tuple(patch_size)).squeeze(3)


class HartleyMultiHeadAttention(Module):
    """Same constructor and parameters as the reference (:49-114): weight_query/key (Z,K,Ci),
    weight_value (Z,V,Ci), weight_out (V, V*Z), optional biases."""

    def __init__(self, in_channels, key_dim, num_heads, num_modes, patch_size=None,
                 attention_activation: Union[str, callable] = 'selu', value_dim=None, key_in_channels=None,
                 value_in_channels=None, use_bias=False, use_transform=True, ndim=5, device=None, dtype=None):
        super().__init__()
        fk = {'device': device, 'dtype': dtype}
        self.in_channels, self.key_dim, self.num_heads = in_channels, key_dim, num_heads
        self.num_modes, self.patch_size = num_modes, patch_size
        self.attention_activation = attention_activation
        self.value_dim = value_dim or key_dim
        self.key_in_channels = key_in_channels or in_channels
        self.value_in_channels = value_in_channels or self.key_in_channels
        self.use_bias, self.use_transform = use_bias, use_transform
        if np.isscalar(self.num_modes):
            self.num_modes = (self.num_modes,) * (ndim - 2)
        else:
            assert len(self.num_modes) == ndim - 2
            self.num_modes = tuple(self.num_modes)
        if np.isscalar(self.patch_size):
            self.patch_size = (self.patch_size,) * (ndim - 2)
        if isinstance(self.attention_activation, str):
            self.attention_activation = getattr(torch.nn.functional, self.attention_activation)
        self.weight_query = Parameter(torch.empty((num_heads, key_dim, self.in_channels), **fk))
        self.weight_key = Parameter(torch.empty((num_heads, key_dim, self.key_in_channels), **fk))
        self.weight_value = Parameter(torch.empty((num_heads, self.value_dim, self.value_in_channels), **fk))
        self.weight_out = Parameter(torch.empty((self.value_dim, self.value_dim * num_heads), **fk))
        if use_bias:
            ones = (1,) * (ndim - 2)
            self.bias_query = Parameter(torch.empty((1, num_heads, key_dim) + ones, **fk))
            self.bias_key = Parameter(torch.empty((1, num_heads, key_dim) + ones, **fk))
            self.bias_value = Parameter(torch.empty((1, num_heads, self.value_dim) + ones, **fk))
            self.bias_out = Parameter(torch.empty((1, self.value_dim) + ones, **fk))
        else:
            for n in ('bias_query', 'bias_key', 'bias_value', 'bias_out'):
                self.register_parameter(n, None)
        self.reset_parameters()

    def reset_parameters(self):
        for w, b in ((self.weight_query, self.bias_query), (self.weight_key, self.bias_key),
                     (self.weight_value, self.bias_value), (self.weight_out, self.bias_out)):
            init.kaiming_uniform_(w, a=math.sqrt(5))
            if b is not None:
                init.zeros_(b)

    # ------------------------------------------------------------------------------------
    def forward(self, inputs):
        return self.forward_fused(inputs)

    def forward_fused(self, inputs, addend=None, act=ops.ACT_NONE):
        """act(attention(inputs) + addend); add and activation fused into the inverse transform's store."""
        if isinstance(inputs, (tuple, list)):
            if len(inputs) not in (2, 3):
                raise ValueError('Invalid inputs.')
            srcs = list(inputs)
        else:
            srcs = [inputs]
        if srcs[0].ndim == 4:
            # 2-D (reference freq_conv2d / grouping2d :165-168, :180-184): the same kernels on (B, C, 1, H, W)
            out = self.forward_fused([t.unsqueeze(2) for t in srcs] if len(srcs) > 1 else srcs[0].unsqueeze(2),
                                     None if addend is None else addend.unsqueeze(2), act)
            return out.squeeze(2)
        spatial = tuple(srcs[0].shape[2:])
        modes = self.num_modes
        patch = self.patch_size
        if len(modes) == 2:                     # a 2-D attention on the (B, C, 1, H, W) view: depth has one mode and patch 1
            modes = (0,) + tuple(modes)
            patch = None if patch is None else (1,) + tuple(patch)
        if self.use_transform:
            assert all(s >= 2 * m for s, m in zip(spatial, modes))      # no clamping here (reference :159-163)
            n3 = float(np.prod(spatial))
            specs = [ops.DhtCropFn.apply(t, modes, 1.0 / n3) for t in srcs]
        else:
            specs = srcs
        q_src, k_src, v_src = specs[0], specs[min(1, len(specs) - 1)], specs[-1]
        act_att = ops.act_id(self.attention_activation)
        Z = self.num_heads
        # freq_conv3d 'zoi,bidhw->bzodhw' for all heads at once: the head axis folds into the output channels, and with a
        # single source (self-attention) the three projections are ONE pointwise conv with the stacked weights
        wq, wk, wv = (w.reshape(-1, w.shape[-1]) for w in (self.weight_query, self.weight_key, self.weight_value))
        fsp = tuple(q_src.shape[2:])
        # biases (reference :174-177, :217-218) are per (head, channel) on the projections and per channel on the output,
        # all on the CROPPED spectrum: they ride on the pointwise convs
        bq, bk, bv, bo = ((b.reshape(-1) if b is not None else None)
                          for b in (self.bias_query, self.bias_key, self.bias_value, self.bias_out))
        if q_src is k_src and k_src is v_src:
            b_all = torch.cat([bq, bk, bv]) if self.use_bias else None
            y = ops.PwConvFn.apply(q_src, None, torch.cat([wq, wk, wv], dim=0), b_all, ops.ACT_NONE)
            q, k, v = torch.split(y, [wq.shape[0], wk.shape[0], wv.shape[0]], dim=1)
        else:
            q = ops.PwConvFn.apply(q_src, None, wq, bq, ops.ACT_NONE)
            k = ops.PwConvFn.apply(k_src, None, wk, bk, ops.ACT_NONE)
            v = ops.PwConvFn.apply(v_src, None, wv, bv, ops.ACT_NONE)
        # split + grouping3d + the (B, Z, C', T) copies as one permutation launch each way when the three projections are one tensor
        one_pass = (patch is not None and q_src is k_src and k_src is v_src and (y.is_cuda or y.is_meta)
                    and os.environ.get('HNO_MHA_GROUP', '1') != '0')
        Kq, Kv = wq.shape[0] // Z, wv.shape[0] // Z
        if (one_pass and wk.shape[0] == wq.shape[0] and os.environ.get('HNO_MHA_GROUP', '1') != '2'
                and (y.is_meta or ops.GroupedAttentionFn.supported(Z, Kq, Kv, tuple(patch), act_att))):
            # grouping, attention and ungrouping as one node: the attention's partial sums are added by the ungrouping permutation
            out = ops.GroupedAttentionFn.apply(y, Z, Kq, Kv, tuple(patch), 1.0 / math.sqrt(Kq * int(np.prod(patch))), act_att)
            out = ops.PwConvFn.apply(out, None, self.weight_out, bo, ops.ACT_NONE)
            if not self.use_transform:
                assert addend is None and act == ops.ACT_NONE
                return out
            if addend is None:
                return ops.PadIdhtFn.apply(out, spatial, 1.0, act)
            return ops.PadIdhtAddFn.apply(out, addend, spatial, 1.0, act)
        if one_pass:
            q, k, v = ops.PatchGroupQKVFn.apply(y, Z, wq.shape[0] // Z, wk.shape[0] // Z, wv.shape[0] // Z, tuple(patch))
            freq_shape = tuple(a // b for a, b in zip(fsp, patch))
        else:
            q, k, v = (t.reshape(t.shape[0], Z, t.shape[1] // Z, *fsp) for t in (q, k, v))   # (B, Z, K, d, h, w)
            if patch is not None:
                q, k, v = (grouping3d(t, patch) for t in (q, k, v))
            freq_shape = tuple(q.shape[3:])
            q, k, v = (t.reshape(t.shape[0], Z, t.shape[2], -1).contiguous() for t in (q, k, v))   # (B, Z, C', T)
        alpha = 1.0 / math.sqrt(k.shape[2])
        if q.is_meta or (q.shape[3] == k.shape[3] and ops.hmha_supported(q.shape[2], v.shape[2])):
            # fused: QK^T -> scale -> activation -> .V in one kernel each way, the (T, T) matrix is never written
            out = ops.HartleyAttentionFn.apply(q, k, v, alpha, act_att)                           # (B, Z, C', Tq)
        else:   # > 128 grouped channels per head: batched GEMMs with the attention matrix in memory
            att = ops.BmmFn.apply(q, k, True, False, alpha)                                      # (B, Z, Tq, Tk)
            if act_att != ops.ACT_NONE:
                att = ops.ActFn.apply(att, act_att)
            out = ops.BmmFn.apply(v, att, False, True, 1.0)                                     # (B, Z, C', Tq)
        if one_pass:
            out = ops.PatchUngroupFn.apply(out, Z, self.value_dim, tuple(patch), fsp)
        else:
            out = out.reshape(out.shape[0], Z, out.shape[2], *freq_shape)
            if patch is not None:
                out = ungrouping3d(out, self.value_dim, patch)
            out = out.reshape(out.shape[0], Z * self.value_dim, *out.shape[3:]).contiguous()
        out = ops.PwConvFn.apply(out, None, self.weight_out, bo, ops.ACT_NONE)                 # 'oi,bidhw->bodhw' (+ bias_out)
        if not self.use_transform:
            assert addend is None and act == ops.ACT_NONE
            return out
        if addend is None:
            return ops.PadIdhtFn.apply(out, spatial, 1.0, act)
        return ops.PadIdhtAddFn.apply(out, addend, spatial, 1.0, act)
