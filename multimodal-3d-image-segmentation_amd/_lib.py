"""ctypes binding of libhno.so (the C ABI declared in include/hno.h).

PyTorch tensors are containers only: every call passes raw device pointers, sizes and the
current HIP stream.  There is NO fallback: if the library is missing or a call fails the
caller gets an exception.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HNO_LIB: another build of the SAME library (same-box A/B of two builds, tools/r6/lib_ab.sh); it is loaded and checked like the default
LIB_PATH = os.environ.get('HNO_LIB') or os.path.join(_HERE, 'libhno.so')
_lib = None

c_void_p, c_int, c_float, c_ll, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/hno.h declares
SIGNATURES = {
    'hno_version': (c_int, []),
    'hno_last_error': (ctypes.c_char_p, []),
    'hno_dht3_workspace_bytes': (c_size_t, [c_int] * 7),
    'hno_dht3_crop': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 7 + [c_float, c_void_p]),
    'hno_dht3_full': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_float, c_void_p]),
    'hno_pad_idht3': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 7 + [c_float, c_void_p]),
    'hno_rfft3_crop': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 8 + [c_float, c_int, c_void_p]),
    'hno_irfft3_pad': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 8 + [c_float, c_int, c_void_p]),
    'hno_rfft3_crop_ld': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 8 + [c_float, c_int, c_ll, c_void_p]),
    'hno_irfft3_pad_ld': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 8 + [c_float, c_int, c_ll, c_void_p]),
    'hno_specmix_shared_fwd': (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    'hno_specmix_shared_bwd': (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    'hno_specmix_bwd_workspace_bytes': (c_size_t, [c_int] * 4),
    'hno_spec_mid_supported': (c_int, [c_int] * 6),
    'hno_dht3_planes': (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_ll, c_void_p]),
    'hno_spec_mid_fwd': (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_float, c_void_p]),
    'hno_spec_mid_bwd_workspace_bytes': (c_size_t, [c_int] * 4),
    'hno_spec_mid_bwd': (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 9 + [c_float, c_void_p]),
    'hno_spec_mid_fourier_supported': (c_int, [c_int] * 5),
    'hno_spec_mid_fourier_fwd': (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_float, c_int, c_int, c_void_p]),
    'hno_spec_mid_fourier_bwd_workspace_bytes': (c_size_t, [c_int] * 3),
    'hno_spec_mid_fourier_bwd': (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 6 + [c_float, c_int, c_int, c_void_p]),
    'hno_idht3_planes': (c_int, [c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 7 + [c_float, c_ll, c_void_p]),
    'hno_dht3_planes_b16': (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_ll, c_void_p]),
    'hno_idht3_planes_b16': (c_int, [c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 7 + [c_float, c_ll, c_void_p]),
    'hno_dht3_ld_supported': (c_int, [c_int] * 6),
    'hno_dht3_crop_ld': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 7 + [c_float, c_ll, c_void_p]),
    'hno_pad_idht3_ld': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 7 + [c_float, c_ll, c_void_p]),
    'hno_specmix_layers_fwd': (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    'hno_specmix_layers_bwd': (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    'hno_pwconv_fwd': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_int, c_void_p]),
    'hno_pwconv_bwd_workspace_bytes': (c_size_t, [c_int, c_int]),
    'hno_pwconv_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_int, c_int, c_int, c_void_p]),
    'hno_pwconv_fwd_branch': (c_int, [c_void_p] * 8 + [c_int] * 4 + [c_ll, c_int, c_void_p]),
    'hno_pwconv_fwd_chain_supported': (c_int, [c_int] * 3),
    'hno_pwconv_fwd_chain': (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_ll, c_int, c_int, c_void_p]),
    'hno_pwconv_bwd_chain_workspace_bytes': (c_size_t, [c_int]),
    'hno_pwconv_bwd_chain': (c_int, [c_void_p] * 13 + [c_int, c_int, c_int, c_ll, c_int, c_int, c_int, c_void_p]),
    'hno_pwconv_bwd_branch_workspace_bytes': (c_size_t, [c_int] * 3),
    'hno_pwconv_bwd_branch': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 6 + [c_int, c_int, c_ll, c_int, c_int, c_void_p]),
    'hno_cmix_compose': (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p]),
    'hno_cmix_compose_multi': (c_int, [c_void_p] * 3 + [c_int, c_int, c_int, c_void_p]),
    'hno_cmix_split_grad': (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p]),
    'hno_cmix_split_grad_ex': (c_int, [c_void_p] * 3 + [c_int, c_int, c_int, c_void_p]),
    'hno_conv_k2s2_fwd': (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_ll, c_void_p]),
    'hno_conv_k2s2_bwd': (c_int, [c_void_p] * 8 + [c_int] * 7 + [c_ll, c_void_p]),
    'hno_conv_k2s2_chain_supported': (c_int, [c_int] * 3),
    'hno_conv_k2s2_chain_bwd_workspace_bytes': (c_size_t, [c_int] * 3),
    'hno_conv_k2s2_chain_fwd': (c_int, [c_void_p] * 6 + [c_int] * 9 + [c_ll, c_void_p]),
    'hno_conv_k2s2_chain_bwd': (c_int, [c_void_p] * 8 + [c_int] * 9 + [c_ll, c_void_p]),
    'hno_upsoftmax_fwd': (c_int, [c_void_p] * 2 + [c_int] * 9 + [c_void_p]),
    'hno_up_argmax': (c_int, [c_void_p] * 2 + [c_int] * 8 + [c_void_p]),
    'hno_upsoftmax_bwd_workspace_bytes': (c_size_t, [c_int] * 8),
    'hno_upsoftmax_bwd': (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p]),
    'hno_upsoftmax_fwd_ld': (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_ll, c_void_p]),
    'hno_up_argmax_ld': (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_ll, c_void_p]),
    'hno_upsoftmax_bwd_ld': (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_ll, c_void_p]),
    'hno_conv3d_k3_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'hno_conv3d_k3_fwd_workspace_bytes': (c_size_t, [c_int] * 7),
    'hno_conv3d_k3': (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 13 + [c_void_p]),
    'hno_conv3d_k3_wgrad': (c_int, [c_void_p] * 4 + [c_int] * 12 + [c_void_p]),
    'hno_convk': (c_int, [c_void_p] * 4 + [c_int] * 13 + [c_void_p]),
    'hno_convk_wgrad': (c_int, [c_void_p] * 3 + [c_int] * 13 + [c_void_p]),
    'hno_groupnorm1_fwd': (c_int, [c_void_p] * 6 + [c_int, c_int, c_ll, c_float, c_int, c_void_p]),
    'hno_groupnorm1_bwd': (c_int, [c_void_p] * 10 + [c_int, c_int, c_ll, c_int, c_void_p]),
    'hno_nearest3d': (c_int, [c_void_p] * 2 + [c_int] * 9 + [c_void_p]),
    'hno_channel_sum_workspace_bytes': (c_size_t, [c_int]),
    'hno_channel_sum': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p]),
    'hno_cb_packed_weight_bytes': (c_size_t, [c_int] * 3),
    'hno_cb_pack_weights': (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    'hno_cb_pack_weights_both': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    'hno_cb_pack_table_row': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 4),
    'hno_cb_pack_row_chunks': (c_ll, [c_void_p]),
    'hno_cb_pack_weights_multi': (c_int, [c_void_p, c_int, c_ll, c_void_p]),
    'hno_cb_conv_workspace_bytes': (c_size_t, [c_int] * 7),
    'hno_cb_conv_stats_floats': (c_size_t, [c_int] * 5),
    'hno_cb_conv': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_size_t]
                    + [c_int] * 12 + [ctypes.POINTER(c_int), c_void_p]),
    'hno_cb_conv_split': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t] + [c_int] * 12 + [c_void_p]),
    'hno_cb_wgrad_workspace_bytes': (c_size_t, [c_int] * 3),
    'hno_cb_wgrad': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t] + [c_int] * 11 + [c_void_p]),
    'hno_cb_gn_apply': (c_int, [c_void_p] * 9 + [c_int, c_int, c_ll, c_int, c_int, c_int, c_float, c_void_p]),
    'hno_cb_gn_bwd_workspace_bytes': (c_size_t, [c_int] * 2),
    'hno_cb_gn_bwd': (c_int, [c_void_p] * 10 + [c_int, c_int, c_ll, c_int, c_int, c_void_p]),
    'hno_cb_pack_input': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_ll, c_void_p]),
    'hno_cb_unpack': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_ll, c_void_p]),
    'hno_cb_colsum_workspace_bytes': (c_size_t, [c_int]),
    'hno_cb_colsum': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_ll, c_void_p]),
    'hno_permode_fwd': (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    'hno_permode_bwd': (c_int, [c_void_p] * 9 + [c_int] * 8 + [c_void_p]),
    'hno_bmm': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_float, c_void_p]),
    'hno_hmha_supported': (c_int, [c_int, c_int]),
    'hno_hmha_workspace_bytes': (c_size_t, [c_int] * 4),
    'hno_hmha_fwd': (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 4 + [c_float, c_int, c_void_p]),
    'hno_hmha_bwd': (c_int, [c_void_p] * 8 + [c_size_t] + [c_int] * 4 + [c_float, c_int, c_void_p]),
    'hno_patch_group3': (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_void_p]),
    'hno_hmha_nsplit': (c_int, [c_int, c_int]),
    'hno_hmha_nsplit_bwd': (c_int, [c_int] * 4),
    'hno_hmha_parts_supported': (c_int, [c_int] * 3),
    'hno_hmha_fwd_parts': (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    'hno_hmha_bwd_parts': (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    'hno_patch_group3_sum': (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_void_p]),
    'hno_act_fwd': (c_int, [c_void_p, c_void_p, c_ll, c_int, c_void_p]),
    'hno_act_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p]),
    'hno_bias_act': (c_int, [c_void_p, c_void_p, c_int, c_int, c_ll, c_int, c_void_p]),
    'hno_add': (c_int, [c_void_p, c_void_p, c_void_p, c_ll, c_void_p]),
    'hno_chan_restride': (c_int, [c_void_p, c_void_p, c_ll, c_ll, c_ll, c_ll, c_void_p]),
    'hno_cast_f32_bf16': (c_int, [c_void_p, c_void_p, c_ll, c_void_p]),
    'hno_debug_invpw_fwd_probe': (c_int, [c_void_p] * 8 + [c_int, c_ll, c_float, c_int, c_void_p]),
    'hno_cast_bf16_f32': (c_int, [c_void_p, c_void_p, c_ll, c_void_p]),
    'hno_axpby': (c_int, [c_float, c_void_p, c_float, c_void_p, c_void_p, c_ll, c_void_p]),
    'hno_loss_fwd': (c_int, [c_void_p] * 5 + [c_int, c_int, c_ll, c_int, c_float, c_void_p]),
    'hno_loss_workspace_doubles': (c_size_t, [c_int, c_int, c_ll]),
    'hno_loss_fwd_ws': (c_int, [c_void_p] * 3 + [c_size_t, c_void_p, c_void_p, c_int, c_int, c_ll, c_int, c_float, c_void_p]),
    'hno_loss_bwd': (c_int, [c_void_p] * 5 + [c_int, c_int, c_ll, c_void_p]),
    'hno_uphead_loss_supported': (c_int, [c_int] * 8),
    'hno_uphead_loss_workspace_doubles': (c_size_t, [c_int, c_int]),
    'hno_uphead_loss_fwd': (c_int, [c_void_p] * 4 + [c_size_t, c_void_p, c_void_p] + [c_int] * 8 + [c_ll, c_int, c_float, c_void_p]),
    'hno_upsoftmax_loss_bwd': (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_ll, c_void_p]),
    'hno_labels_prepare': (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 2 + [c_int, c_int, c_ll, c_void_p]),
    'hno_adamax_chunk_rows': (c_int, []),
    'hno_adamax_multi': (c_int, [c_void_p, c_int] + [c_float] * 5 + [c_ll, c_float, c_void_p]),
    'hno_adamax_state_doubles': (c_int, []),
    'hno_adamax_multi_dev': (c_int, [c_void_p, c_int, c_void_p] + [c_float] * 5 + [c_void_p]),
    'hno_pwmulti_supported': (c_int, [c_int, c_int, c_int]),
    'hno_pwmulti_fwd': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_ll, c_void_p]),
    'hno_pwmulti_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_ll]),
    'hno_pwmulti_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_ll, c_ll, c_void_p]),
    'hno_adamax_multi_dev_amp': (c_int, [c_void_p, c_int, c_void_p, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'hno_sum_pairs': (c_int, [c_void_p] * 5 + [c_int, c_void_p]),
    'hno_zscore_workspace_bytes': (c_size_t, [c_int]),
    'hno_zscore_modalities': (c_int, [c_void_p] * 4 + [c_int, c_ll, c_int, c_float, c_int, c_float, c_float, c_void_p]),
    'hno_affine_nearest': (c_int, [c_void_p, c_void_p, c_void_p, c_float] + [c_int] * 5 + [c_void_p]),
    'hno_set_defer_reduce': (c_int, [c_int]),
    'hno_pending_reduces': (c_int, []),
    'hno_discard_reduces': (c_int, []),
    'hno_flush_reduces': (c_int, [c_void_p]),
    'hno_profile_begin': (c_int, [c_int]),
    'hno_profile_end': (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    'hno_profile_kernel_name': (ctypes.c_char_p, [c_int]),
    'hno_debug_stamps': (c_int, [c_void_p, c_int]),
    'hno_set_debug': (c_int, [c_int]),
    'hno_debug_last_plane_family': (c_int, [c_int]),
    'hno_debug_reduce_launches': (c_ll, [c_int]),
    'hno_selftest_gemm': (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
}


class HnoError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libhno.so for gfx950 with hipcc (works without a GPU)."""
    cmd = ['make', '-C', os.path.join(_HERE, 'csrc'), '-j', '8']
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-8000:])
    if res.returncode != 0:
        raise HnoError('building libhno.so failed')
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HnoError(f'{LIB_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                           f'(there is no CPU/PyTorch fallback for the HIP path)')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chan_padded(t):
    """(B, C, D, H, W) fp32 whose (b, c) volumes are contiguous and a multiple of 32 floats apart (ops.chan_stride)"""
    if t.dim() != 5:
        return False
    B, C, D, H, W = t.shape
    st = t.stride()
    return st[1] % 32 == 0 and 0 <= st[1] - D * H * W < 32 and tuple(st[2:]) == (H * W, W, 1) and (B == 1 or st[0] == C * st[1])


def ptr(t):
    """Device pointer of a contiguous (or channel-padded: ops.chan_stride) tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and (t.is_contiguous() or _chan_padded(t)), 'libhno needs contiguous device tensors'
    return ctypes.c_void_p(t.data_ptr())


def check(rc, what):
    if rc != 0:
        msg = lib().hno_last_error().decode(errors='replace')
        if rc == -1:
            raise ValueError(f'{what}: {msg}')
        raise HnoError(f'{what} failed ({rc}): {msg}')


class KernelProfile:
    """Context manager around hno_profile_begin/end: per-kernel HIP-event durations."""

    def __init__(self, max_records=65536):
        self.max_records = max_records
        self.records = []   # (kernel name, milliseconds, algorithmic bytes)

    def __enter__(self):
        check(lib().hno_profile_begin(self.max_records), 'hno_profile_begin')
        return self

    def __exit__(self, *exc):
        ids = (ctypes.c_int * self.max_records)()
        ms = (ctypes.c_float * self.max_records)()
        nbytes = (ctypes.c_double * self.max_records)()
        n = lib().hno_profile_end(ids, ms, nbytes, self.max_records)
        if n < 0:
            check(n, 'hno_profile_end')
        names = {}
        for i in range(n):
            if ids[i] not in names:
                names[ids[i]] = lib().hno_profile_kernel_name(ids[i]).decode()
            self.records.append((names[ids[i]], ms[i], nbytes[i]))
        return False

    def summary(self):
        """name -> (calls, total ms, average ms, total algorithmic bytes)"""
        agg = {}
        for name, t, nb in self.records:
            c, s, b = agg.get(name, (0, 0.0, 0.0))
            agg[name] = (c + 1, s + t, b + nb)
        return {k: (c, s, s / c, b) for k, (c, s, b) in agg.items()}
