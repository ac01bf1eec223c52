import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
L = pkg._lib.lib(); P, S = pkg._lib.ptr, pkg._lib.stream_ptr
dev = 'cuda'
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (B, K, d, D) in ((2, 4, (65, 65, 65), (128, 128, 128)), (1, 4, (121, 121, 78), (240, 240, 155)), (1, 3, (9, 11, 13), (16, 20, 24))):
    lr = torch.randn(B, K, *d, device=dev)
    out = {}
    for flag in (16, 0, 1024 << 8, 2048 << 8, 8192 << 8):
        L.hno_set_debug(flag)
        pr = torch.empty(B, K, *D, device=dev); lab = torch.empty(B, *D, dtype=torch.uint8, device=dev)
        t1 = timeit(lambda: L.hno_upsoftmax_fwd(P(lr), P(pr), B, K, *d, *D, 1, S()))
        t2 = timeit(lambda: L.hno_up_argmax(P(lr), P(lab), B, K, *d, *D, S()))
        out[flag] = (pr.clone(), lab.clone())
        print(f'{D} flags {flag} ({"voxel form" if flag == 16 else "segment form grid " + str(flag >> 8)}): upsoftmax {t1:.1f} us, upargmax {t2:.1f} us')
    L.hno_set_debug(0)
    print('   probs identical:', torch.equal(out[0][0], out[16][0]), (out[0][0] - out[16][0]).abs().max().item(), ' labels identical:', torch.equal(out[0][1], out[16][1]))
