"""training(use_autocast=True) end to end on device-resident synthetic batches: seconds per training step with the autocast step replayed
from a HIP graph (round 6 default) and eager (HNO_TRAIN_GRAPH_AUTOCAST=0, rounds 3-5).  python tools/r6/train_autocast_ab.py [fnoseg|vnet]"""
import os, sys, time, json, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
from multimodal_3d_image_segmentation_amd.nets import custom_losses
which = sys.argv[1] if len(sys.argv) > 1 else 'fnoseg'
if which == 'fnoseg':
    make = lambda: pkg.nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Fourier')
    size, bs = (128, 128, 128), 2
else:
    make = lambda: pkg.nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4])
    size, bs = (160, 192, 128), 1
NB = int(os.environ.get("LAB_NB", "24"))

class Data:      # device-resident batches (the PCIe copy is not what is compared here); the InputData methods training() uses
    def __init__(self):
        g = torch.Generator(device='cuda').manual_seed(1)
        self.batch_size = bs
        self.x = torch.randn((bs, 4) + size, device='cuda', generator=g)
        self.y = torch.randint(0, 4, (bs, 1) + size, device='cuda', generator=g).float()
        self.num_labels, self.in_channels = 4, 4
    def get_train_flow(self, shuffle=True): return [(self.x, self.y)] * NB
    def get_valid_flow(self): return []
    def get_train_num_batches(self): return NB
    def get_valid_num_batches(self): return 0
    def get_train_image_size(self): return size
res = {}
for mode in ('replayed', 'eager'):
    os.environ['HNO_TRAIN_GRAPH_AUTOCAST'] = '1' if mode == 'replayed' else '0'
    torch.manual_seed(0)
    model = make().cuda()
    opt = pkg.optim.Adamax(model.parameters(), lr=1e-3)
    tt.step_stats.update(replayed=0, eager=0)
    with tempfile.TemporaryDirectory() as d:
        torch.cuda.synchronize(); t0 = time.time()
        tt.training(model, Data(), os.path.join(d, 'o'), custom_losses.PCCLoss(), opt, None, num_epochs=3, selection_epoch_portion=2.0,
                    checkpoint_epoch=100, is_print=False, use_autocast=True, device='cuda')
        torch.cuda.synchronize(); dt = time.time() - t0
        tl, _ = tt.get_losses_from_file(os.path.join(d, 'o', 'stdout.txt'))
    res[mode] = {'seconds_total': round(dt, 3), 'steps': dict(tt.step_stats), 'train_loss': tl}
    del model, opt
    torch.cuda.empty_cache()
# steady state: the last epoch's steps are all replayed; time a fourth-epoch-like burst directly would need internals -- report totals
print(json.dumps({'model': which, 'batches_per_epoch': NB, 'epochs': 3, **res}))
