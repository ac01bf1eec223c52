"""weights after k steps: eager Adamax vs device-stepped eager vs CapturedStep with the optimizer inside.  python tools/dbg/opt_modes.py"""
import sys, os, copy, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.experiments.train_test import CapturedStep
from multimodal_3d_image_segmentation_amd.nets import custom_losses
ops = pkg.ops
MODES = tuple(int(v) for v in os.environ.get('MODES', '3,3,3').split(','))
torch.manual_seed(0)
model0 = pkg.nets.HNOSegXS(2, 3, 8, [1, 1, 1, 1], MODES).cuda()
w0 = copy.deepcopy(model0.state_dict())
x = torch.randn(2, 2, 16, 20, 24, device='cuda')
labf = torch.randint(0, 3, (2, 1, 16, 20, 24), device='cuda').float()
loss_fn = custom_losses.PCCLoss()
NSTEP = 6


def run(mode):
    model = pkg.nets.HNOSegXS(2, 3, 8, [1, 1, 1, 1], MODES).cuda()
    model.load_state_dict(w0)
    opt = pkg.optim.Adamax(model.parameters(), lr=5e-3)
    cap = None
    if mode != 'eager':
        assert opt.device_stepped(None)
    if mode == 'captured':
        cap = CapturedStep(model, loss_fn, 3, None, None, optimizer=opt if os.environ.get('OPT_IN', '1') == '1' else None)
    snaps = []
    for i in range(NSTEP):
        l = cap.step(x, labf) if cap is not None else None
        if l is not None and not cap.steps_optimizer:
            opt.step()
        if l is None:
            for p in model.parameters():
                p.grad = None
            lab = ops.labels_prepare(labf, 3)
            with ops.expected_loss(lab, loss_fn):
                y = model(x)
            loss_fn(y, lab).backward()
            opt.step()
        torch.cuda.synchronize()
        snaps.append([p.detach().clone() for p in model.parameters()])
    return snaps, [n for n, _ in model.named_parameters()]


order = (sys.argv[1] if len(sys.argv) > 1 else 'eager,device,captured').split(',')
res = {m: run(m) for m in order}
if len(res) < 3:
    print('ran', order); sys.exit(0)
a, names = res['eager']
b, _ = res['device']
c, _ = res['captured']
for i in range(NSTEP):
    eb = max(float((u - v).abs().max() / v.abs().max()) for u, v in zip(b[i], a[i]))
    ec = [(float((u - v).abs().max() / v.abs().max()), n) for u, v, n in zip(c[i], a[i], names)]
    print('step', i + 1, 'device vs eager %.2e' % eb, ' captured vs eager %.2e (%s)' % max(ec))
