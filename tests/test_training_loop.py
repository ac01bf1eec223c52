"""Training-loop mirror vs the reference's own `training()` trajectory (golden G8) -- GPU."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from _inputs import TRAIN_CASE, make_train_input

pytestmark = pytest.mark.gpu


def _setup(g):
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    model = pkg.nets.HNOSegXS(**TRAIN_CASE['model'])
    model.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd0::')})
    opt = torch.optim.Adamax(model.parameters(), lr=TRAIN_CASE['lr'])
    data = make_train_input()
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(
        opt, T_0=data.get_train_num_batches() * TRAIN_CASE['epochs'], eta_min=TRAIN_CASE['eta_min'])
    return model, opt, sched, data, custom_losses.PCCLoss()


def test_training_matches_reference_trajectory(tmp_path):
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    g = load_golden('g8_training.npz')
    model, opt, sched, data, loss_fn = _setup(g)
    tt.training(model, data, str(tmp_path), loss_fn, opt, sched, label_mapping=TRAIN_CASE['mapping'],
                num_epochs=TRAIN_CASE['epochs'], selection_epoch_portion=0.5, checkpoint_epoch=2, is_print=False,
                device='cuda')
    tl, vl = tt.get_losses_from_file(os.path.join(tmp_path, 'stdout.txt'))
    assert np.abs(np.array(tl) - g['train_loss']).max() < 2e-5      # loss values are ~0.49
    assert np.abs(np.array(vl) - g['valid_loss']).max() < 2e-5
    assert sorted(os.listdir(os.path.join(tmp_path, 'model'))) == list(g['files'])
    ck = torch.load(os.path.join(tmp_path, 'model', 'checkpoint.pt'), weights_only=False)
    assert sorted(ck.keys()) == list(g['checkpoint_keys'])
    assert ck['epoch'] == int(g['checkpoint_epoch'])
    assert (ck['best_epoch'] is None) == (int(g['best_epoch']) < 0)
    assert abs(opt.param_groups[0]['lr'] - float(g['final_lr'])) < 1e-12
    for k, v in model.state_dict().items():                            # weights after 8 Adamax steps
        assert rel_err(v.cpu().numpy(), g[f'sd1::{k}']) < 2e-3, k


@pytest.mark.parametrize('two_streams', ['0', '1'])
def test_training_graph_replay_equals_eager(tmp_path, monkeypatch, two_streams):
    """training() replays forward + loss + backward of recurring batch shapes from a HIP graph (CapturedStep, round 3): the same
    kernels, so the same trajectory as eager launches -- and the first batch of a shape still runs eagerly."""
    monkeypatch.setenv('HNO_SPLIT_STREAMS', two_streams)      # captured steps as one pass / as two half-batches on two streams (SampleSplit)
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    g = load_golden('g8_training.npz')
    runs = {}
    for tag, flag in (('graph', True), ('eager', False)):
        model, opt, sched, data, loss_fn = _setup(g)
        before = dict(tt.step_stats)
        tt.training(model, data, str(tmp_path / tag), loss_fn, opt, sched, label_mapping=TRAIN_CASE['mapping'],
                    num_epochs=TRAIN_CASE['epochs'], selection_epoch_portion=0.5, checkpoint_epoch=2, is_print=False,
                    device='cuda', use_graph=flag)
        n_rep, n_eag = tt.step_stats['replayed'] - before['replayed'], tt.step_stats['eager'] - before['eager']
        assert ((n_rep > 0 and n_eag == 1) if flag else (n_rep == 0 and n_eag > 1)), (tag, n_rep, n_eag)
        runs[tag] = (tt.get_losses_from_file(os.path.join(tmp_path / tag, 'stdout.txt')), [v.clone() for v in model.state_dict().values()])
    assert np.abs(np.array(runs['graph'][0][0]) - np.array(runs['eager'][0][0])).max() < 2e-6
    assert np.abs(np.array(runs['graph'][0][1]) - np.array(runs['eager'][0][1])).max() < 2e-6
    for a, b in zip(runs['graph'][1], runs['eager'][1]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-4


@pytest.mark.parametrize('two_streams', ['0', '1'])
def test_training_graph_replay_with_ragged_last_batch(tmp_path, monkeypatch, two_streams):
    """Five samples in batches of two: shapes (2, ...) and (1, ...) alternate.  Each shape runs eagerly once, then from its own graph;
    gradients written by one form must never leak into a step of the other (CapturedStep re-installs .grad after every replay)."""
    monkeypatch.setenv('HNO_SPLIT_STREAMS', two_streams)      # captured steps as one pass / as two half-batches on two streams (SampleSplit)
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    from multimodal_3d_image_segmentation_amd.experiments.synthetic import SyntheticInputData
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    runs = {}
    for tag, flag in (('graph', True), ('eager', False)):
        torch.manual_seed(2)
        model = pkg.nets.HNOSegXS(2, 3, 8, [1, 1, 1], (3, 3, 3))
        opt = pkg.optim.Adamax(model.parameters(), lr=5e-3)
        data = SyntheticInputData((16, 16, 16), 2, 3, batch_size=2, num_train=5, num_valid=2, seed=5)
        before = dict(tt.step_stats)
        tt.training(model, data, str(tmp_path / tag), custom_losses.DiceLoss(), opt, None, num_epochs=4, selection_epoch_portion=0.5,
                    checkpoint_epoch=2, is_print=False, device='cuda', use_graph=flag)
        n_rep, n_eag = tt.step_stats['replayed'] - before['replayed'], tt.step_stats['eager'] - before['eager']
        assert (n_rep, n_eag) == ((10, 2) if flag else (0, 12)), (tag, n_rep, n_eag)
        runs[tag] = (tt.get_losses_from_file(os.path.join(tmp_path / tag, 'stdout.txt')), [v.clone() for v in model.state_dict().values()])
    assert np.abs(np.array(runs['graph'][0][0]) - np.array(runs['eager'][0][0])).max() < 2e-6
    for a, b in zip(runs['graph'][1], runs['eager'][1]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-4


def test_training_with_optimizer_and_schedule_inside_the_captured_step(tmp_path, monkeypatch):
    """Round 4: optim.Adamax moves its step counter, the learning rate and the per-batch CosineAnnealingWarmRestarts schedule
    (experiments/run.py:92-103, train_test.py:173-174) onto the device; training() then captures the update behind backward, so a step
    is ONE graph replay.  Same trajectory as the eager optimizer + scheduler.step() (G8 golden incl. the final learning rate), the
    host objects are in sync at every epoch end (checkpoint layout unchanged: it resumes into an eager run and vice versa)."""
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    g = load_golden('g8_training.npz')

    def setup():
        model = pkg.nets.HNOSegXS(**TRAIN_CASE['model'])
        model.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd0::')})
        model = model.cuda()
        opt = pkg.optim.Adamax(model.parameters(), lr=TRAIN_CASE['lr'])
        data = make_train_input()
        sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(
            opt, T_0=data.get_train_num_batches() * TRAIN_CASE['epochs'], eta_min=TRAIN_CASE['eta_min'])
        return model, opt, sched, data, custom_losses.PCCLoss()
    kw = dict(label_mapping=TRAIN_CASE['mapping'], selection_epoch_portion=0.5, checkpoint_epoch=2, is_print=False, device='cuda')
    runs = {}
    for tag, flag in (('captured', '1'), ('eager_opt', '0')):
        monkeypatch.setenv('HNO_TRAIN_GRAPH_OPT', flag)
        model, opt, sched, data, loss_fn = setup()
        before = dict(tt.step_stats)
        tt.training(model, data, str(tmp_path / tag), loss_fn, opt, sched, num_epochs=TRAIN_CASE['epochs'], use_graph=True, **kw)
        assert tt.last_run['device_stepped_optimizer'] == (flag == '1') and not opt.is_device_stepped      # (training() leaves the mode it entered)
        assert tt.step_stats['replayed'] - before['replayed'] > 0
        tl, vl = tt.get_losses_from_file(os.path.join(tmp_path / tag, 'stdout.txt'))
        assert np.abs(np.array(tl) - g['train_loss']).max() < 2e-5 and np.abs(np.array(vl) - g['valid_loss']).max() < 2e-5
        sd = opt.state_dict()
        runs[tag] = (tl, vl, opt.param_groups[0]['lr'], sched.state_dict(), sd['state'][0]['step'], [v.clone() for v in model.state_dict().values()])
        assert abs(opt.param_groups[0]['lr'] - float(g['final_lr'])) < 1e-12
    a, b = runs['captured'], runs['eager_opt']
    assert np.abs(np.array(a[0]) - np.array(b[0])).max() < 2e-6 and np.abs(np.array(a[1]) - np.array(b[1])).max() < 2e-6
    assert abs(a[2] - b[2]) < 1e-15 and float(a[4]) == float(b[4])
    for k in ('T_cur', 'T_i', 'last_epoch'):
        assert a[3][k] == b[3][k], (k, a[3][k], b[3][k])
    for u, v in zip(a[5], b[5]):
        assert rel_err(u.cpu().numpy(), v.cpu().numpy()) < 1e-5
    # resume: two epochs device-stepped, the rest eager (the checkpoint is torch's layout) -> the golden trajectory again
    monkeypatch.setenv('HNO_TRAIN_GRAPH_OPT', '1')
    model, opt, sched, data, loss_fn = setup()
    tt.training(model, data, str(tmp_path / 'resume'), loss_fn, opt, sched, num_epochs=2, use_graph=True, **kw)
    monkeypatch.setenv('HNO_TRAIN_GRAPH_OPT', '0')
    model2, opt2, sched2, data2, _ = setup()
    tt.training(model2, data2, str(tmp_path / 'resume'), loss_fn, opt2, sched2, num_epochs=TRAIN_CASE['epochs'], use_graph=False, **kw)
    tl, _ = tt.get_losses_from_file(os.path.join(tmp_path / 'resume', 'stdout.txt'))
    assert len(tl) == TRAIN_CASE['epochs'] and np.abs(np.array(tl) - g['train_loss']).max() < 5e-5


def test_training_resume_from_checkpoint(tmp_path):
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    g = load_golden('g8_training.npz')
    kw = dict(label_mapping=TRAIN_CASE['mapping'], selection_epoch_portion=0.5, checkpoint_epoch=2, is_print=False,
              device='cuda')
    model, opt, sched, data, loss_fn = _setup(g)
    tt.training(model, data, str(tmp_path), loss_fn, opt, sched, num_epochs=2, **kw)       # stops after a checkpoint
    model2, opt2, sched2, data2, _ = _setup(g)                                              # fresh objects, as in a new run
    tt.training(model2, data2, str(tmp_path), loss_fn, opt2, sched2, num_epochs=TRAIN_CASE['epochs'], **kw)
    tl, vl = tt.get_losses_from_file(os.path.join(tmp_path, 'stdout.txt'))
    assert len(tl) == TRAIN_CASE['epochs']
    assert np.abs(np.array(tl) - g['train_loss']).max() < 5e-5
    with pytest.raises(RuntimeError):                                   # exhausted epochs (train_test.py:85-86)
        tt.training(model2, data2, str(tmp_path), loss_fn, opt2, sched2, num_epochs=TRAIN_CASE['epochs'], **kw)


def test_label_helpers_gpu():
    from multimodal_3d_image_segmentation_amd.experiments import utils
    g = load_golden('g9_misc.npz')
    lab = torch.from_numpy(g['labels']).cuda()
    assert np.array_equal(utils.to_categorical(lab, 5).cpu().numpy(), g['onehot5'])
    assert np.array_equal(utils.to_categorical(lab).cpu().numpy(), g['onehot_auto'])
    mapping = {int(k): int(v) for k, v in zip(g['remap_keys'], g['remap_vals'])}
    assert np.array_equal(utils.remap_labels(lab, mapping).cpu().numpy(), g['remapped'])


@pytest.mark.parametrize('family', ['hnosegxs', 'fnoseg', 'vnetds'])
def test_testing_labels_equal_argmax_of_probabilities(tmp_path, family):
    """testing() (fused upsample + argmax on the GPU, uint8 over PCIe) must give exactly the class map the reference
    protocol gives: argmax over channels of model(x) on the host (experiments/train_test.py:398-408)."""
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    from multimodal_3d_image_segmentation_amd.experiments.synthetic import SyntheticInputData
    torch.manual_seed(3)
    nets = pkg.nets
    if family == 'hnosegxs':
        model, size = nets.HNOSegXS(2, 3, 8, [2, 2], (3, 4, 4)), (16, 20, 24)
    elif family == 'fnoseg':
        model, size = nets.NeuralOperatorSeg(2, 3, 8, 2, (3, 4, 4), 'Fourier'), (16, 20, 24)
    else:
        model, size = nets.VNetDS(2, 3, 4, [1, 1], right_leg_indexes=[0, 1]), (16, 16, 16)
    model = model.cuda()
    data = SyntheticInputData(size, 2, 3, batch_size=1, num_train=0, num_valid=0, num_test=3)
    mapping = {0: 0, 1: 4, 2: 9}
    y_true, y_pred = tt.testing(model, data, str(tmp_path / 'out'), label_mapping=mapping, is_print=False, device='cuda')
    assert len(y_pred) == 3 and os.path.exists(tmp_path / 'out' / 'prediction_time_memory.txt')
    model.eval()
    for i, (x, y) in enumerate(data.get_test_flow()):
        with torch.no_grad():
            probs = model(x.cuda())
        want = probs.argmax(1).cpu().numpy().astype(np.uint8)[0]
        want = np.vectorize(mapping.get)(want).astype(want.dtype)
        assert y_pred[i].shape == size and np.array_equal(y_pred[i], want)
        assert np.array_equal(y_true[i], np.asarray(y, dtype=np.uint8)[0, 0])
        assert np.array_equal(np.load(tmp_path / 'out' / 'images' / f'{i}_pred.npy'), want)


def test_testing_vs_reference_protocol_golden(tmp_path):
    """f1: the class maps of ``testing()`` against what the REFERENCE's own ``testing()`` wrote for the same weights and
    inputs (golden G14: model.eval(), batch 1, host arg max over the probabilities, label remapping; train_test.py:332-426).
    Exact equality on every voxel whose top-2 probability margin exceeds 1e-4 (the fp32 parity tolerance: below it the
    arg max of two correct fp32 evaluations may differ); in total no more than 0.1 % of the voxels may differ."""
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    from _inputs import TEST_CASE, make_test_input
    g = load_golden('g14_testing.npz')
    model = pkg.nets.HNOSegXS(**TEST_CASE['model'])
    model.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd::')})
    y_true, y_pred = tt.testing(model.cuda(), make_test_input(), str(tmp_path / 'out'), label_mapping=TEST_CASE['mapping'],
                                is_print=False, device='cuda')
    assert sorted(os.listdir(tmp_path / 'out')) == ['images', 'prediction_time_memory.txt']
    assert len(y_pred) == TEST_CASE['num_test']
    for i in range(TEST_CASE['num_test']):
        want, margin = g[f'pred_{i}'], g[f'margin_{i}']
        assert y_pred[i].shape == want.shape and y_pred[i].dtype == np.uint8
        assert np.array_equal(y_true[i], g[f'true_{i}'])
        sure = margin > 1e-4
        assert np.array_equal(y_pred[i][sure], want[sure]), i
        assert float((y_pred[i] != want).mean()) < 1e-3
        assert set(np.unique(y_pred[i])) <= {0, 5, 9}                     # remapped labels (mapping {1: 5, 2: 9})


def test_training_with_autocast_and_grad_scaler(tmp_path):
    """training(use_autocast=True) (reference train_test.py:79,154-168): bf16 autocast around forward + loss, GradScaler around
    backward / step, the scaler's state in the checkpoint, resume.  V-Net-DS with 8 base filters: its convolutions run on the bf16
    matrix-core path; the loss must fall and stay close to the fp32 run of the same schedule."""
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    from multimodal_3d_image_segmentation_amd.experiments.synthetic import SyntheticInputData
    from multimodal_3d_image_segmentation_amd.nets import custom_losses

    def make():
        torch.manual_seed(1)
        model = pkg.nets.VNetDS(2, 3, 8, [1, 1], right_leg_indexes=[0, 1])
        opt = pkg.optim.Adamax(model.parameters(), lr=5e-3)
        data = SyntheticInputData((16, 16, 16), 2, 3, batch_size=2, num_train=4, num_valid=2, seed=3,
                                  generator=lambda i: (torch.randn(2, 16, 16, 16, generator=torch.Generator().manual_seed(i)),
                                                       (torch.arange(16 ** 3).reshape(1, 16, 16, 16) % 3).float()))
        return model, opt, data
    losses = {}
    for tag, ac in (('bf16', True), ('f32', False), ('bf16_eager', True)):
        model, opt, data = make()
        out = tmp_path / tag
        # round 6: autocast runs replay forward + loss + scaled backward from a HIP graph too (GradScaler's step / update eager behind
        # the replay); HNO_TRAIN_GRAPH_AUTOCAST=0 keeps them eager as rounds 3-5 did
        if tag == 'bf16_eager':
            os.environ['HNO_TRAIN_GRAPH_AUTOCAST'] = '0'
        tt.step_stats.update(replayed=0, eager=0)
        try:
            tt.training(model, data, str(out), custom_losses.DiceLoss(), opt, None, num_epochs=4, selection_epoch_portion=0.5,
                        checkpoint_epoch=2, is_print=False, use_autocast=ac, device='cuda')
        finally:
            os.environ.pop('HNO_TRAIN_GRAPH_AUTOCAST', None)
        assert (tt.step_stats['replayed'] > 0) == (tag != 'bf16_eager'), dict(tt.step_stats)
        tl, vl = tt.get_losses_from_file(os.path.join(out, 'stdout.txt'))
        losses[tag] = tl
        ck = torch.load(os.path.join(out, 'model', 'checkpoint.pt'), weights_only=False)
        assert ('scaler_state_dict' in ck) == ac
        assert all(np.isfinite(tl)) and tl[-1] < tl[0]
    assert np.abs(np.array(losses['bf16']) - np.array(losses['f32'])).max() < 2e-2
    # the replayed autocast run follows the eager autocast run (same kernels, same scaler trajectory)
    assert np.abs(np.array(losses['bf16']) - np.array(losses['bf16_eager'])).max() < 1e-5, (losses['bf16'], losses['bf16_eager'])
    # resume an autocast run: the scaler state is restored with the rest of the checkpoint
    model, opt, data = make()
    tt.training(model, data, str(tmp_path / 'bf16'), custom_losses.DiceLoss(), opt, None, num_epochs=6, selection_epoch_portion=0.5,
                checkpoint_epoch=2, is_print=False, use_autocast=True, device='cuda')
    tl, _ = tt.get_losses_from_file(os.path.join(tmp_path / 'bf16', 'stdout.txt'))
    assert len(tl) == 6 and tl[:4] == losses['bf16']


@pytest.mark.gpu
@pytest.mark.parametrize('loss_name', ['pcc', 'dice', 'expdice'])
def test_sample_split_gives_the_batch_gradients(loss_name, monkeypatch):
    """Round 4b: experiments.train_test.SampleSplit -- the two halves of a batch as two concurrent passes (model and its storage-aliasing
    twin) on two streams of a captured graph.  Loss and every gradient equal the whole-batch pass (means over (sample, class): the halves
    average exactly); the twin follows in-place weight updates (every replay computes with the current weights); a model whose storage
    moved is detected; outside a capture the split refuses to run."""
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd import ops
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    from multimodal_3d_image_segmentation_amd.nets import custom_losses as CL
    loss_fn = {'pcc': CL.PCCLoss(), 'dice': CL.DiceLoss(), 'expdice': CL.ExpDiceLoss(0.3)}[loss_name]
    torch.manual_seed(3)
    model = pkg.nets.HNOSegXS(4, 4, 24, [1, 1, 1, 1], (4, 6, 6)).cuda()
    x = torch.randn(4, 4, 32, 32, 32, device='cuda')
    lab = torch.randint(0, 4, (4, 32, 32, 32), device='cuda').to(torch.uint8)
    # a model class names itself a CANDIDATE ('measure': CapturedStep / bench.py time both forms when the step is captured and keep the
    # faster, test_captured_step_measures_its_schedule); HNO_SPLIT_STREAMS=1 / 0 force it on / off for every model
    monkeypatch.delenv('HNO_SPLIT_STREAMS', raising=False)
    assert tt.SampleSplit.candidate(model, loss_fn, x) == 'measure'
    other = pkg.nets.NeuralOperatorSeg(4, 4, 24, 2, (4, 6, 6), 'Fourier').cuda()      # a family that is no candidate (measured: no gain)
    assert tt.SampleSplit.candidate(other, loss_fn, x) is False
    monkeypatch.setenv('HNO_SPLIT_STREAMS', '0')
    assert tt.SampleSplit.candidate(model, loss_fn, x) is False
    monkeypatch.setenv('HNO_SPLIT_STREAMS', '1')
    assert tt.SampleSplit.candidate(model, loss_fn, x) == 'force' and tt.SampleSplit.candidate(other, loss_fn, x) == 'force'
    assert tt.SampleSplit.usable(model, loss_fn, x) and not tt.SampleSplit.usable(model, loss_fn, x[:3]) and not tt.SampleSplit.usable(model, torch.nn.MSELoss(), x)
    assert tt.SampleSplit.usable(other, loss_fn, x)
    params = [p for p in model.parameters()]

    def whole():
        for p in params:
            p.grad = None
        with ops.expected_loss(lab, loss_fn):
            y = model(x)
        l = loss_fn(y, lab)
        l.backward()
        out = float(l.detach()), [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        del y, l
        return out
    l0, g0 = whole()
    whole()                                                        # (every kernel attribute / table of the half-batch shapes exists)
    split = tt.SampleSplit(model)
    assert split.aliased()
    with pytest.raises(RuntimeError):
        split.fwd_bwd(x, lab, loss_fn)
    # the half-batch shapes once eagerly through the model (tables and kernel attributes are created at first use, not capturable)
    with torch.no_grad():
        model(x[:2])
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    prev = ops.set_defer_reduce(True)
    try:
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                loss = split.fwd_bwd(x, lab, loss_fn)
    finally:
        ops.set_defer_reduce(prev)
    torch.cuda.current_stream().wait_stream(side)
    grads = [p.grad for p in params]

    def check(lref, gref):
        graph.replay()
        torch.cuda.synchronize()
        assert abs(float(loss) - lref) < 2e-6
        num = sum(float(((a - g) ** 2).sum()) for a, g in zip(grads, gref)) ** 0.5
        den = sum(float((g ** 2).sum()) for g in gref) ** 0.5
        assert num / den < 2e-5, num / den
        for a, g in zip(grads, gref):
            assert rel_err(a.cpu().numpy(), g.cpu().numpy()) < 2e-4
    check(l0, g0)
    check(l0, g0)
    with torch.no_grad():                                          # in-place updates reach the twin: it shares the storage
        for p in params:
            p.mul_(1.01)
    l2, g2 = whole()
    check(l2, g2)
    params[0].data = params[0].data.clone()                        # a re-materialised parameter does not: detected
    assert not split.aliased()


@pytest.mark.gpu
@pytest.mark.parametrize('transform', ['Fourier', 'Hartley'])
def test_neural_operator_seg_backward_batches_every_reduction(transform):
    """FNOSeg / HNOSeg with deferred reductions on (what bench.py and the captured step of training() run): ONE batched slab reduction
    per backward pass -- until round 5 every Fourier block reduced its dW2 on its own (24 launches per cfg3 step) because the real /
    imaginary split read it at once; the split is recorded behind the reduction now and runs as ONE kernel for all blocks
    (hno_cmix_split_grad_ex).  Gradients equal the undeferred pass bit for bit."""
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd import ops, _lib
    from multimodal_3d_image_segmentation_amd.nets import custom_losses as CL
    torch.manual_seed(9)
    model = pkg.nets.NeuralOperatorSeg(4, 4, 24, 3, (10, 14, 14), transform).cuda()
    loss_fn = CL.PCCLoss()
    x = torch.randn(1, 4, 64, 64, 64, device='cuda')
    y = torch.randint(0, 4, (1, 1, 64, 64, 64), device='cuda').float()
    params = list(model.parameters())
    L = _lib.lib()

    def run(defer):
        prev = ops.set_defer_reduce(defer)
        try:
            lab = ops.labels_prepare(y, 4)
            for p in params:
                p.grad = None
            out = model(x)
            l = loss_fn(out, lab)
            s0, m0 = L.hno_debug_reduce_launches(0), L.hno_debug_reduce_launches(1)
            l.backward()
            torch.cuda.synchronize()
            counts = L.hno_debug_reduce_launches(0) - s0, L.hno_debug_reduce_launches(1) - m0
            return float(l.detach()), [p.grad.clone() for p in params], counts
        finally:
            ops.set_defer_reduce(prev)
    l0, g0, c0 = run(False)
    l1, g1, c1 = run(True)
    assert c0[1] == 0 and c0[0] >= 3                       # undeferred: a launch per slab set
    assert c1[1] == 1 and c1[0] <= 1, c1                   # deferred: one batched launch (the stride-2 stem convolution keeps its own)
    assert l0 == l1
    for a, b in zip(g0, g1):
        assert bool((a == b).all())


def test_captured_step_measures_its_schedule(monkeypatch):
    """Round 5: whether a captured step runs its batch as one pass or as two half-batches on two streams is MEASURED when the step is
    captured (train_test.choose_schedule: both forms captured, replayed, the faster kept) -- round 4 carried a hand-measured list of
    shapes in model code.  Whatever wins, the replayed step gives the eager step's loss and gradients, its weight-gradient slabs are
    reduced by batched launches only (round 4's split silently fell back to 17 single launches per pass), and a forced answer
    (HNO_SPLIT_STREAMS) skips the measurement."""
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd import ops, _lib
    from multimodal_3d_image_segmentation_amd.experiments import train_test as tt
    from multimodal_3d_image_segmentation_amd.nets import custom_losses as CL
    monkeypatch.delenv('HNO_SPLIT_STREAMS', raising=False)
    monkeypatch.setenv('HNO_TRAIN_GRAPH_QUIET', '1')
    torch.manual_seed(5)
    model = pkg.nets.HNOSegXS(4, 4, 24, [1, 1, 1, 1], (4, 6, 6)).cuda()
    loss_fn = CL.PCCLoss()
    x = torch.randn(2, 4, 32, 32, 32, device='cuda')
    y = torch.randint(0, 4, (2, 1, 32, 32, 32), device='cuda').float()
    params = [p for p in model.parameters()]

    def eager():
        lab = ops.labels_prepare(y, 4)
        for p in params:
            p.grad = None
        with ops.expected_loss(lab, loss_fn):
            out = model(x)
        l = loss_fn(out, lab)
        l.backward()
        res = float(l.detach()), [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        del out, l
        return res
    l_ref, g_ref = eager()
    with torch.no_grad():            # a validation-style pass in between must not disturb the batched reduction (round-4 bug)
        model(x)
    L = _lib.lib()
    for forced in (None, '1', '0'):
        if forced is None:
            monkeypatch.delenv('HNO_SPLIT_STREAMS', raising=False)
        else:
            monkeypatch.setenv('HNO_SPLIT_STREAMS', forced)
        cap = tt.CapturedStep(model, loss_fn, 4)
        assert cap.step(x, y) is None                      # first sighting of a shape: the caller runs it eagerly
        single0, multi0 = L.hno_debug_reduce_launches(0), L.hno_debug_reduce_launches(1)
        loss = cap.step(x, y)                              # second: measured (or forced), captured, replayed
        assert loss is not None
        torch.cuda.synchronize()
        key = next(iter(cap.entries))
        if forced is None:
            form, ms_one, ms_split = cap.schedule[key]
            assert ms_one > 0 and ms_split > 0 and form == ('two streams' if ms_split < ms_one else 'one pass')
            passes = 1 + 2 + (2 if form == 'two streams' else 1)      # both trial captures + the kept one
        else:
            assert not cap.schedule
            passes = 2 if forced == '1' else 1
        # every backward pass that was enqueued reduced ALL its slab sets in one batched launch
        assert L.hno_debug_reduce_launches(0) == single0
        assert L.hno_debug_reduce_launches(1) - multi0 == passes
        assert abs(float(loss) - l_ref) < 2e-6
        for p, g in zip(params, g_ref):
            assert rel_err(p.grad.cpu().numpy(), g.cpu().numpy()) < 2e-4
        loss2 = cap.step(x, y)
        torch.cuda.synchronize()
        assert abs(float(loss2) - l_ref) < 2e-6
        del cap
        for p in params:
            p.grad = None
