"""Does a memory-bound pointwise kernel overlap with an issue-bound plane kernel when both are resident on the same compute units?
Two independent chains (one sample each: the halves of the two-stream schedule) run alone and concurrently on two streams of one graph.
    python tools/r6/corun.py            (environment: HNO_PWF_WAVES, HNO_PWF_GRID_CAP, HNO_PWB_GRID_CAP, HNO_ITEM_FWD_WAVES, HNO_ITEM_INV_WAVES ...)"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from multimodal_3d_image_segmentation_amd._lib import lib, ptr, check, stream_ptr

dev = torch.device('cuda')
torch.manual_seed(0)
B, C, N = 1, 24, 65
modes = (10, 14, 14)
ld = ops._pad_ld(N ** 3)
def act():
    t = ops.act_empty(B, C, (N, N, N), dev, ld)
    t.as_strided((B * C * ld,), (1,)).normal_()
    return t
u, t, x, gy, y = act(), act(), act(), act(), act()
W = torch.randn(24, 48, device=dev) * 0.1
bias = torch.zeros(24, device=dev)
L = lib()
ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N, N, N, *modes) // 4, device=dev)
ws2 = torch.empty_like(ws)
out = act()

def pw_fwd():
    return ops.pwconv_fwd_raw(u, t, W, bias, ops.ACT_SELU)
def pw_bwd():
    return ops.pwconv_bwd_raw(gy, y, u, t, W, ops.ACT_SELU, True, xa_act=ops.ACT_SELU)
def plane_fwd():
    check(L.hno_dht3_planes(ptr(x), ptr(ws), B * C, N, N, N, *modes, ld, stream_ptr()), 'planes')
def plane_inv():
    check(L.hno_idht3_planes(ptr(ws2), None, ops.ACT_SELU, ptr(out), B * C, N, N, N, *modes, 1.0, ld, stream_ptr()), 'iplanes')
def plane_inv_add():
    check(L.hno_idht3_planes(ptr(ws2), ptr(t), ops.ACT_NONE, ptr(out), B * C, N, N, N, *modes, 1.0, ld, stream_ptr()), 'iplanes')

plane_fwd(); ws2.copy_(ws); pw_fwd(); pw_bwd(); plane_inv(); plane_inv_add()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
REP = 12

def graph_of(fa, fb):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            if fa is not None:
                s1.wait_stream(side)
                with torch.cuda.stream(s1):
                    for _ in range(REP):
                        fa()
            if fb is not None:
                s2.wait_stream(side)
                with torch.cuda.stream(s2):
                    for _ in range(REP):
                        fb()
            if fa is not None:
                side.wait_stream(s1)
            if fb is not None:
                side.wait_stream(s2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return g

def time_graph(g):
    for _ in range(3):
        g.replay()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    return best

res = {}
singles = {'pw_fwd': pw_fwd, 'pw_bwd': pw_bwd, 'plane_fwd': plane_fwd, 'plane_inv': plane_inv, 'plane_inv_add': plane_inv_add}
for k, f in singles.items():
    res[k] = round(time_graph(graph_of(f, None)), 2)
for a in ('pw_fwd', 'pw_bwd'):
    for b in ('plane_fwd', 'plane_inv', 'plane_inv_add', a):
        if b == a:
            continue
        res[f'{a}||{b}'] = round(time_graph(graph_of(singles[a], singles[b])), 2)
        res[f'{a}+{b}'] = round(res[a] + res[b], 2)
print(json.dumps({'env': {k: v for k, v in os.environ.items() if k.startswith('HNO_')}, 'us_per_iteration': res}))
