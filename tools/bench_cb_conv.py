"""Stand-alone timing of hno_cb_conv / hno_cb_wgrad on one V-Net layer shape (bf16 path), with ablation flags."""
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops_bf16 as ob
L = pkg._lib.lib()
SHAPES = {'l0_48_24': (24, 24, 24, (81, 97, 65)), 'l0_24_24': (24, 0, 24, (81, 97, 65)), 'l1_96_48': (48, 48, 48, (41, 49, 33)),
          'l1_48_48': (48, 0, 48, (41, 49, 33)), 'l2_96_96': (96, 0, 96, (21, 25, 17)), 'l2_192_96': (96, 96, 96, (21, 25, 17)),
          'l3_192_192': (192, 0, 192, (11, 13, 9)), 'l4_384_384': (384, 0, 384, (6, 7, 5))}
names = [a for a in sys.argv[1:] if a in SHAPES] or list(SHAPES)
flags = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0]
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n
for name in names:
    Ca, Cb, Cout, sp = SHAPES[name]
    xa = torch.randn((1,) + sp + (Ca,), device='cuda').bfloat16()
    xb = torch.randn((1,) + sp + (Cb,), device='cuda').bfloat16() if Cb else None
    g = torch.randn((1,) + sp + (Cout,), device='cuda').bfloat16()
    W = torch.randn(Cout, Ca + Cb, 3, 3, 3, device='cuda') * 0.05
    wp = ob.pack_weights(W, 0, Ca + Cb, Cout, 3)
    wpd = ob.pack_weights(W, 1, Ca + Cb, Cout, 3)
    gf = 2.0 * sp[0] * sp[1] * sp[2] * 27 * (Ca + Cb) * Cout / 1e9
    for f in flags:
        L.hno_set_debug(f)
        t_f = timeit(lambda: ob.conv_raw(xa, xb, wp, None, Cout, sp, 0, 3, 1, 1, True))
        t_d = timeit(lambda: ob.conv_raw(g, None, wpd, None, Ca + Cb, sp, 1, 3, 1, 1, False))
        t_w = timeit(lambda: ob.wgrad_raw(g, xa, xb, W.shape, False, 3, 1, 1))
        print(f'{name:12s} flags {f:5d}: fwd {t_f * 1e6:7.1f} us {gf / t_f / 1e3:6.1f} TF | dgrad {t_d * 1e6:7.1f} us {gf / t_d / 1e3:6.1f} TF | '
              f'wgrad {t_w * 1e6:7.1f} us {gf / t_w / 1e3:6.1f} TF   ({gf:.2f} GF)')
    L.hno_set_debug(0)
