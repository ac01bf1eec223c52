"""training() with and without graph-replayed steps (CapturedStep) for a small V-Net-DS / HartleyMHASeg / FNOSeg: steps by launch form and
the loss trajectories (identical).  (A capture used to die inside hipStreamEndCapture while the loop still held the previous eager step's
loss: DESIGN lesson 22.)  python tools/dbg/graph_train_ab.py [vnet|mha|fno]"""
import os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
from multimodal_3d_image_segmentation_amd.experiments.synthetic import SyntheticInputData
from multimodal_3d_image_segmentation_amd.nets import custom_losses
res = {}
for flag in (False, True):
    torch.manual_seed(1)
    which = sys.argv[1] if len(sys.argv) > 1 else 'vnet'
    model = (pkg.nets.VNetDS(2, 3, 8, [1, 1], right_leg_indexes=[0, 1]) if which == 'vnet' else
             pkg.nets.HartleyMHASeg(2, 3, 8, 2, 2, (3, 3, 3), 2, 'selu') if which == 'mha' else
             pkg.nets.NeuralOperatorSeg(2, 3, 8, 2, (3, 3, 3), 'Fourier'))
    opt = pkg.optim.Adamax(model.parameters(), lr=5e-3)
    data = SyntheticInputData((16, 16, 16), 2, 3, batch_size=2, num_train=6, num_valid=2, seed=3)
    out = tempfile.mkdtemp()
    before = dict(tt.step_stats)
    tt.training(model, data, out, custom_losses.DiceLoss(), opt, None, num_epochs=3, selection_epoch_portion=0.5, checkpoint_epoch=2,
                is_print=False, device='cuda', use_graph=flag)
    tl, vl = tt.get_losses_from_file(os.path.join(out, 'stdout.txt'))
    res[flag] = tl
    print('use_graph', flag, 'replayed', tt.step_stats['replayed'] - before['replayed'], 'eager', tt.step_stats['eager'] - before['eager'], tl)
print('max loss difference', max(abs(a - b) for a, b in zip(res[False], res[True])))
