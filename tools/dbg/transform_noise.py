"""fp32 round-off of the mode-truncated transforms: HIP kernels vs the reference's op sequence (torch.fft in fp32 on the CPU),
both against the float64 dense formulation."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import multimodal_3d_image_segmentation_amd as pkg
from oracle import hno_oracle as O
from multimodal_3d_image_segmentation_amd.nets.hnosegxs import TransformCrop, PadInverse
torch.manual_seed(0)
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
def rms(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for sp, modes in [((33, 33, 33), (10, 14, 14)), ((65, 65, 65), (10, 14, 14)), ((21, 19, 23), (10, 14, 14)), ((61, 61, 40), (10, 14, 14))]:
    x = torch.randn(1, 4, *sp)
    m = O.clamp_modes(modes, sp)
    z64 = O.dht_crop_dense(x.double(), m)
    z_ref = O.transform_crop(x, m)                      # reference op sequence, fp32 torch.fft on the CPU
    z_hip = TransformCrop(modes, 5)(x.cuda()).cpu()
    y64 = O.pad_idht_dense(z64, sp)
    y_ref = O.pad_inverse(z64.float(), sp)
    y_hip = PadInverse(5)(z64.float().cuda(), sp).cpu()
    print(f'{sp}: crop  rms err ref {rms(z_ref, z64):.2e} hip {rms(z_hip, z64):.2e} | max ref {rel(z_ref, z64):.2e} hip {rel(z_hip, z64):.2e}   '
          f'|| pad_inverse rms ref {rms(y_ref, y64):.2e} hip {rms(y_hip, y64):.2e} | max ref {rel(y_ref, y64):.2e} hip {rel(y_hip, y64):.2e}')
