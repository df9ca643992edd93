"""Drop-in trainer API on the MI355X: forward_one_batch against the reference goldens, calculate_metrics on arbitrary tensors,
in-loop validation + checkpoints + early stop through an AudioDataLoader (both trainers), greedy decoding against the
reference golden, the long-utterance configuration against the live oracle, one RCCL execution of the collective path."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests.test_parity_gpu import make, _pass_parity, _rel_errs, _set_oracle_params, RTOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name', ['F0', 'F1'])
def test_forward_one_batch_matches_reference_goldens(name):
    """TransientTrainer.forward_one_batch (transient_trainer.py:25-73): loss, total_cer, total_char of the three training
    batches evaluated at theta0 against what the REAL reference returned (golden `fwd/0/{0,2,4}/{loss,cer}`), the in-place
    scaling of src_percentages (Q6), and loss.backward() against the oracle with the pass's branch decisions replayed."""
    from oracle import refimpl as R
    from oracle import branches
    z, cfg, spec = gu.load(name)
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    oracle = R.build_model(cfg)
    tr, _val = gu.batches_for(cfg, spec, 0, z['data_call_index'])
    trainer = mtl_amd.TransientTrainer()
    for m, (x, lens, y) in enumerate(tr):
        key = 'fwd/0/%d' % (2 * m)
        pct = lens.float() / x.shape[3]
        pct0 = pct.clone()
        model.zero_grad()
        loss, cer, nchar = trainer.forward_one_batch(model, vocab, x.cuda(), y.cuda(), pct, lens, (y != 0).sum(1).to(torch.int32), 0.0, 'ce')
        assert (cer, nchar) == tuple(int(v) for v in z[key + '/cer']), key
        assert abs(float(loss) - float(z[key + '/loss'])) <= RTOL * float(z[key + '/loss'])
        assert torch.equal(pct, pct0 * int(y.shape[1] + 1))                  # reference quirk: caller's tensor scaled in place
        loss.backward()
        gates = branches.gates_from_engine(model.engine)
        pr, gr, _ = oracle(x, lens, y, gates=gates)
        grads = torch.autograd.grad(R.ce_loss(pr, gr), list(oracle.parameters()))
        errs = _rel_errs(model, model.flat_grad, oracle, grads)
        assert max(errs.values()) < RTOL, max(errs.items(), key=lambda kv: kv[1])


def test_forward_one_batch_label_smoothing_matches_reference_formula():
    """--label-smoothing through the compatibility call: loss and gradients vs utils/metrics.py:113-124 restated in torch."""
    import torch.nn.functional as F
    from oracle import refimpl as R
    from oracle import branches
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    oracle = R.build_model(cfg)
    x, lens, y = gu.batches_for(cfg, spec, 0, z['data_call_index'])[0][1]
    eps = 0.1
    model.zero_grad()
    loss, _, _ = mtl_amd.TransientTrainer().forward_one_batch(model, vocab, x.cuda(), y.cuda(), lens.float() / x.shape[3], lens,
                                                            (y != 0).sum(1).to(torch.int32), eps, 'ce')
    loss.backward()
    pred, gold, _ = oracle(x, lens, y, gates=branches.gates_from_engine(model.engine))
    V = pred.size(2)
    p2, g2 = pred.view(-1, V), gold.view(-1)
    mask = g2.ne(0)
    one_hot = torch.zeros_like(p2).scatter(1, (mask.long() * g2).view(-1, 1), 1)
    one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / V
    ref = -(one_hot * F.log_softmax(p2, dim=1)).sum(1).masked_select(mask).sum() / int(mask.sum())
    assert abs(float(loss) - float(ref)) < RTOL * float(ref)
    grads = torch.autograd.grad(ref, list(oracle.parameters()))
    errs = _rel_errs(model, model.flat_grad, oracle, grads)
    assert max(errs.values()) < RTOL, max(errs.items(), key=lambda kv: kv[1])


def test_calculate_metrics_on_arbitrary_tensors():
    """calculate_metrics is a real op (utils/metrics.py:68-126), not a view of the engine's last forward: leaf logits, a SLICE of a
    model output, an explicit non_pad_mask (in-place padding of gold like the reference), smoothing; d(pred) vs torch autograd."""
    import torch.nn.functional as F
    import mtl_amd
    g = torch.Generator().manual_seed(5)
    pred = torch.randn(3, 7, 100, generator=g)
    gold = torch.randint(1, 100, (3, 7), generator=g)
    gold[1, 4:] = 0
    gold[2, 6:] = 0
    for smoothing in (0.0, 0.2):
        pd = pred.clone().cuda().requires_grad_(True)
        loss, ncorrect = mtl_amd.calculate_metrics(pd[:2], gold[:2].cuda(), 0, smoothing=smoothing)      # a slice of "pred"
        (2.5 * loss).backward()
        pc = pred.clone().requires_grad_(True)
        p2, g2 = pc[:2].reshape(-1, 100), gold[:2].reshape(-1)
        if smoothing > 0:
            mask = g2.ne(0)
            one_hot = torch.zeros_like(p2).scatter(1, (mask.long() * g2).view(-1, 1), 1)
            one_hot = one_hot * (1 - smoothing) + (1 - one_hot) * smoothing / 100
            ref = -(one_hot * F.log_softmax(p2, dim=1)).sum(1).masked_select(mask).sum() / int(mask.sum())
        else:
            ref = F.cross_entropy(p2, g2, ignore_index=0, reduction='mean')
        (2.5 * ref).backward()
        assert abs(float(loss) - float(ref)) < 2e-6 * float(ref)
        assert ncorrect == int((p2.max(1)[1].eq(g2) & g2.ne(0)).sum())
        assert float((pd.grad.cpu() - pc.grad).abs().max()) < 2e-6 * float(pc.grad.abs().max())
        assert float(pd.grad[2].abs().max()) == 0.0
    # explicit mask: positions masked out are padded in the caller's gold tensor, exactly like the reference
    gd = gold.clone().cuda()
    mask = gd.ne(0)
    mask[0, 0] = False
    loss, _ = mtl_amd.calculate_metrics(pred.cuda(), gd, 0, non_pad_mask=mask)
    assert int(gd[0, 0]) == 0
    g3 = gold.clone()
    g3[0, 0] = 0
    assert abs(float(loss) - float(F.cross_entropy(pred.view(-1, 100), g3.view(-1), ignore_index=0))) < 2e-6 * float(loss)
    with pytest.raises(RuntimeError):
        mtl_amd.calculate_metrics(pred, gold, 0)                         # CPU tensors: no fallback


class _ListDataset(torch.utils.data.Dataset):
    """(spectrogram (F, T), transcript ids) items, what SpectrogramDataset.__getitem__ yields (utils/data_loader.py:323-340)"""

    def __init__(self, seed, n, V):
        g = torch.Generator().manual_seed(seed)
        self.items = []
        for _ in range(n):
            T = int(torch.randint(24, 72, (1,), generator=g))
            L = int(torch.randint(2, 8, (1,), generator=g))
            self.items.append((torch.randn(161, T, generator=g), torch.randint(4, V, (L,), generator=g).tolist()))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _oracle_validation(oracle, vocab, loader):
    """what the reference's validation loop computes for one loader, through the oracle in eval mode"""
    from oracle import refimpl as R
    import mtl_amd
    from mtl_amd.trainer import cer_counts
    oracle.eval()
    tot_loss, tot_cer, tot_char, nb = 0.0, 0, 0, 0
    with torch.no_grad():
        for src, trg, _pct, src_lengths, _tl in loader:
            pred, gold, hyp = oracle(src, src_lengths, trg)
            tot_loss += float(R.ce_loss(pred, gold))
            c, n = cer_counts(vocab, gold, hyp)
            tot_cer += c
            tot_char += n
            nb += 1
    oracle.train()
    return tot_loss / nb, tot_cer * 100 / tot_char


def test_in_loop_validation_checkpoints_and_early_stop(tmp_path):
    """TransientTrainer.train with evaluate_every=1 over two AudioDataLoaders (utils/data_loader.py:401-440 layout): per-set loss
    and CER equal the oracle's on the same batches, metrics / history like transient_trainer.py:280-331, epoch_N.th every
    save_every, best_model.th on improvement, and the early-stop counter ends the run (meta_lr = 0: nothing improves)."""
    from oracle import refimpl as R
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, dict(spec, meta_lr=0.0), name='valid')
    args.save_folder, args.save_every = str(tmp_path), 2
    model = model.cuda()
    oracle = R.build_model(cfg)
    V = cfg['vocab_size']
    loaders = [mtl_amd.AudioDataLoader(vocab.PAD_ID, dataset=_ListDataset(100 + i, 5, V), batch_size=2) for i in range(2)]
    expect = [_oracle_validation(oracle, vocab, ld) for ld in loaders]
    tasks = [mtl_amd.SyntheticTask(m, 2, 64, 8, V, variable=True) for m in range(3)]
    trainer = mtl_amd.TransientTrainer()
    trainer.train(model, vocab, tasks, loaders, 'ce', 0, 10, args, evaluate_every=1, early_stop='cer,2', is_copy_grad=True)
    hist = trainer.history
    assert len(hist) == 3                                   # best at it 1, count_stop 1 at it 2, 2 at it 3 -> EARLY STOP
    for h in hist:
        for i, (loss_ref, cer_ref) in enumerate(expect):
            assert abs(h['valid_loss'][i] - loss_ref) < RTOL * loss_ref
            assert abs(h['valid_cer'][i] - cer_ref) < 1e-9
        assert abs(h['avg_valid_cer'] - sum(c for _, c in expect) / 2) < 1e-9
    folder = os.path.join(str(tmp_path), 'valid')
    assert sorted(os.listdir(folder)) == ['best_model.th', 'epoch_2.th']
    m2, v2, inner2, outer2, epoch, metrics, a2 = mtl_amd.load_meta_model(os.path.join(folder, 'epoch_2.th'))
    assert epoch == 2 and metrics['valid_cer'] == hist[1]['valid_cer'] and v2.id2label == vocab.id2label
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1.cpu(), p2.cpu())            # meta_lr 0: theta never moved
    assert model.training


def test_joint_trainer_validation_and_checkpoint(tmp_path):
    """JointTrainer.train (joint_trainer.py:306-380): validation every iteration, save_joint_model layout
    ('vocab','args','epoch','model_state_dict','opt','metrics'), load_joint_model restores weights and the Adam state."""
    from oracle import refimpl as R
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, spec, name='joint')
    args.save_folder, args.save_every, args.loss = str(tmp_path), 1, 'ce'
    model = model.cuda()
    V = cfg['vocab_size']
    loaders = [mtl_amd.AudioDataLoader(vocab.PAD_ID, dataset=_ListDataset(7, 4, V), batch_size=4)]
    tasks = [mtl_amd.SyntheticTask(m, 2, 64, 8, V, variable=True) for m in range(3)]
    tr = mtl_amd.JointTrainer()
    tr.train(model, vocab, tasks, loaders, 'ce', 0, 2, args, evaluate_every=1, early_stop='loss,5')
    assert len(tr.history) == 2 and np.isfinite(tr.history[1]['avg_valid_loss'])
    assert tr.history[1]['avg_valid_loss'] != tr.history[0]['avg_valid_loss']                            # theta moved
    folder = os.path.join(str(tmp_path), 'joint')
    assert sorted(os.listdir(folder)) == ['best_model.th', 'epoch_1.th', 'epoch_2.th']
    raw = mtl_amd.functions.load_checkpoint_dict(os.path.join(folder, 'epoch_2.th'))
    assert sorted(raw) == ['args', 'epoch', 'metrics', 'model_state_dict', 'opt', 'vocab']
    m2, v2, opt2, epoch, metrics, a2 = mtl_amd.load_joint_model(os.path.join(folder, 'epoch_2.th'))
    assert epoch == 2
    for (n1, p1), (_, p2) in zip(model.named_parameters(), m2.named_parameters()):
        assert torch.equal(p1.cpu(), p2.cpu())
    st = opt2.state_dict()['state']
    assert len(st) == 68 and int(st[0]['step']) == 2
    flat = mtl_amd.FlatAdam.from_torch(m2, opt2)
    assert torch.equal(flat.m.cpu(), tr.opt.m.cpu()) and torch.equal(flat.v.cpu(), tr.opt.v.cpu())
    # validation loss of the final model against the oracle carrying the same weights
    oracle = R.build_model(cfg)
    _set_oracle_params(oracle, model, model.flat_parameters)
    loss_ref, cer_ref = _oracle_validation(oracle, vocab, loaders[0])
    assert abs(tr.history[1]['valid_loss'][0] - loss_ref) < RTOL * loss_ref and abs(tr.history[1]['valid_cer'][0] - cer_ref) < 1e-9


def test_greedy_decoding_matches_reference_golden():
    """Transformer.evaluate / PassEngine.greedy_decode (K/V-cached, device-resident feedback) against the REAL reference's
    Decoder.greedy_search (tests/golden/G0.npz): the token ids of all 300 steps and the returned strings."""
    gspec, ids, strs, golds = gu.load_greedy()
    z, cfg, spec = gu.load('F0')
    cfg = dict(cfg, tgt_max_len=gspec['tgt_max_len'])
    mtl_amd, args, vocab, model = make(cfg, spec)
    gu.perturb_output_layer(model.decoder.output_linear.weight, gspec)
    model = model.cuda()
    from oracle import refimpl as R
    x, lens, y = R.synth_batch(gspec['seed'], gspec['k'], gspec['T'], gspec['L'], cfg['vocab_size'], True)
    _, hyps, gold_strs = model.evaluate(x.cuda(), lens, y, args, start_token=vocab.SOS_ID)
    assert np.array_equal(model.last_greedy_ids.numpy(), ids)
    assert hyps == strs and gold_strs == golds


@pytest.mark.parametrize('B,lens', [(1, [5000]), (2, [5000, 1300])])
def test_long_utterance_config_against_live_oracle(B, lens):
    """BASELINE.json configs[3] (src-max-len 5000, T' = 1250, dim-input 5120) against the oracle at small batch: labels
    bit-exact, loss, and every gradient tensor within 1e-4 with the branch decisions replayed (second case: a padded row,
    raw-length masks on the pooled axis)."""
    from oracle import refimpl as R
    z, cfg, spec = gu.load('NS')
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    oracle = R.build_model(cfg)
    g = torch.Generator().manual_seed(50 + B)
    x = torch.randn(B, 1, 161, 5000, generator=g)
    y = torch.randint(4, cfg['vocab_size'], (B, 40), generator=g)
    for i, n in enumerate(lens):
        x[i, :, :, n:] = 0
    if B > 1:
        y[1, 25:] = 0
    _pass_parity(model, oracle, (x, torch.tensor(lens, dtype=torch.int32), y), model.flat_parameters, 'T=5000 B=%d' % B,
                 max_flips=120)


def test_long_utterance_full_batch_against_live_oracle():
    """BASELINE.json configs[3] at its FULL batch: the training batch of the T5 record (8 utterances, up to 5000 frames, variable lengths:
    raw-length masks on the 1250-wide pooled axis, zero tails) against the live oracle -- labels bit-exact, loss, and with the device's
    ReLU / max-pool decisions replayed EVERY gradient tensor within 1e-4 (measured: 62 near-ties decided differently, margin 1.9e-6,
    worst tensor 2.5e-5).  The oracle pass takes ~40 s of CPU time and ~10 GB of host memory."""
    from oracle import refimpl as R
    z, cfg, spec = gu.load('T5')
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    oracle = R.build_model(cfg)
    tr, _val = gu.batches_for(cfg, spec, 0, z['data_call_index'])
    assert tuple(tr[0][0].shape) == (8, 1, 161, 5000) and int(tr[0][1].min()) < 1250
    _pass_parity(model, oracle, tr[0], model.flat_parameters, 'T=5000 B=8', max_flips=400)


def _norm_per_utterance(x, lens):
    """utils/data_loader.py:84-94 on a synthetic batch: every utterance to zero mean / unit std over its OWN frames, zero tail beyond"""
    x = x.clone()
    for i, n in enumerate(lens.tolist()):
        v = x[i, :, :, :n]
        x[i, :, :, :n] = (v - v.mean()) / v.std()
        x[i, :, :, n:] = 0
    return x


def test_h2_range_guard_census_is_clean_on_normalised_utterances_of_very_different_lengths():
    """The guard of the two-piece fp16 convolutions (TransientTrainer.h2_check_every): on features normalised per utterance
    (utils/data_loader.py:84-94) with zero tails of very different lengths -- what the real collate hands over -- every h2 operand of
    every pass (activations forward, gradients backward) stays in the full-precision regime: no non-zero element below 16 bits, the
    trainer stays on h2, and the pass meets the oracle at 1e-4 on every tensor."""
    from oracle import refimpl as R
    z, cfg, spec = gu.load('F1')
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    assert model._need_engine().conv_h2
    g = torch.Generator().manual_seed(4242)
    lens = torch.tensor([240, 30, 111, 8], dtype=torch.int32)
    mk = lambda: (_norm_per_utterance(torch.randn(4, 1, 161, 240, generator=g) * 3.0 + 1.5, lens), lens,
                  torch.randint(4, cfg['vocab_size'], (4, 9), generator=g))
    tasks = [mk() for _ in range(2)]
    val = mk()
    as5 = lambda b: (b[0].cuda(), b[1], None, b[2], None)
    inner, outer = mtl_amd.FlatSGD(model, spec['lr']), mtl_amd.FlatAdam(model, spec['meta_lr'])
    model.zero_copy_grad()
    for batch_tasks in (True, False):                    # one stacked pass per phase / a lane per task
        tr = mtl_amd.TransientTrainer()
        tr.h2_check_every, tr.batch_tasks = 1, batch_tasks
        tr.run_iteration(model, vocab, [as5(b) for b in tasks], as5(val), 2, inner, outer, args)
        cen = tr.h2_census
        assert cen is not None and set(cen) == {'conv0 out', 'pool1', 'conv5 out', 'pool2', 'd pool2', 'd conv5 out', 'd pool1', 'd input-linear out'}
        print('census (%s):' % tr.last_schedule, {k: ('%.1e' % v[0], '%.1e' % v[1]) for k, v in cen.items()})
        assert max(v[1] for v in cen.values()) <= 1e-4 < tr.h2_limit and all(e.conv_h2 for e in model.engines)    # (measured: <= 6e-6, gradients)
        assert max(cen[k][0] for k in ('conv0 out', 'pool1', 'conv5 out', 'pool2')) <= 1e-3
    oracle = R.build_model(cfg)
    _pass_parity(model, oracle, tasks[0], model.flat_parameters, 'normalised utterances, lengths 240 / 30 / 111 / 8')


def test_h2_range_guard_fires_on_a_loud_and_a_quiet_sample_and_moves_to_the_exact_split():
    """One scale per tensor: a sample 2^24 quieter than its batch-mate keeps ~15 bits in the two fp16 pieces.  (With the reference's
    biases the convolutions' outputs of a quiet INPUT are bias-dominated -- not quiet; the spread is produced here the way it can arise
    inside the stack: bias-free convolutions.)  The census counts the quiet sample's elements, the guard moves the engines to the exact
    3 x bf16 split, and against the oracle the quiet sample's logits are then ~100 x closer than under h2."""
    from oracle import refimpl as R
    from oracle import branches
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, spec)
    with torch.no_grad():
        for i in (0, 2, 5, 7):
            getattr(model.conv, str(i)).bias.zero_()
    model = model.cuda()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 1, 161, 64, generator=g)
    x[1] *= 2.0 ** -24
    lens = torch.tensor([64, 64], dtype=torch.int32)
    y = torch.randint(4, cfg['vocab_size'], (2, 8), generator=g)
    batch = (x, lens, y)
    as5 = lambda b: (b[0].cuda(), b[1], None, b[2], None)
    oracle = R.build_model(cfg)
    _set_oracle_params(oracle, model, model.flat_parameters)
    p2 = {}
    model.pass_forward(x.cuda(), lens, y)
    p2['h2'] = model.engine.arena['p2'].clone()               # (2, T/4, F/4, 128): the conv stack's output, per sample
    inner, outer = mtl_amd.FlatSGD(model, 0.0), mtl_amd.FlatAdam(model, 0.0)       # (no parameter motion: the same theta afterwards)
    model.zero_copy_grad()
    tr = mtl_amd.TransientTrainer()
    tr.h2_check_every = 1
    assert tr.h2_guard == 'x3' and all(e.conv_h2 for e in model.engines)
    tr.run_iteration(model, vocab, [as5(batch)], as5(batch), 1, inner, outer, args)
    cen = tr.h2_census
    print('census:', {k: ('%.2f' % v[0], '%.2f' % v[1]) for k, v in cen.items()})
    assert cen['conv0 out'][1] > 0.3 and cen['pool2'][1] > 0.3           # the quiet sample's half of the non-zero elements
    assert all(e.conv_mode == 'x3' and not e.conv_h2 for e in model.engines) and not tr._cmdlists
    model.pass_forward(x.cuda(), lens, y)
    p2['x3'] = model.engine.arena['p2'].clone()
    # the exact split is the yardstick (its parity with the oracle: the whole-pass check at the end): under h2 the LOUD sample's conv
    # output agrees with it to fp32 rounding, the quiet sample's -- every element ~2^24 below the tensor's bound -- only to ~2^-14
    d = [float((p2['h2'][i] - p2['x3'][i]).norm() / p2['x3'][i].norm()) for i in range(2)]
    print('conv stack output, h2 against the exact split: loud sample %.2e, quiet sample %.2e' % (d[0], d[1]))
    assert float(p2['x3'][1].norm()) > 0 and d[0] < 2e-6 and d[1] > 20 * d[0]
    tr2 = mtl_amd.TransientTrainer()                         # 'raise' stops instead
    for e in model.engines:
        e.conv_mode, e.conv_x3, e.conv_h2 = 'h2', True, True
    tr2.h2_check_every, tr2.h2_guard = 1, 'raise'
    with pytest.raises(RuntimeError, match='h2 range guard'):
        tr2.run_iteration(model, vocab, [as5(batch)], as5(batch), 1, inner, outer, args)
    # the whole pass on the exact split against the oracle: labels, loss, every gradient tensor (decisions replayed)
    for e in model.engines:
        e.conv_mode, e.conv_x3, e.conv_h2 = 'x3', True, False
    _pass_parity(model, oracle, batch, model.flat_parameters, 'loud / quiet pair on the exact split')


def test_rccl_collective_path_executes():
    """The `nccl` (= RCCL) branch of dist.init_from_env and the flat-G all-reduce on this box's GPU: one rank under torchrun with
    MTL_DIST_ALWAYS=1 (the collective is issued even at world size 1).  Multi-rank arithmetic is covered by the gloo tests; this
    pins that RCCL initialises and runs the collective in this environment, and that it leaves the step unchanged."""
    common = ['--steps', '2', '--warmup', '1', '--tasks', '2', '--k', '2', '--frames', '200', '--labels', '20', '--no-cpu-baseline']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('MTL_DIST_BACKEND', None)
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + common, capture_output=True, text=True,
                         env=env, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    rccl = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                           '127.0.0.1', '--master-port', '29741', os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + common,
                          capture_output=True, text=True, env=dict(env, MTL_DIST_ALWAYS='1'), timeout=300)
    assert rccl.returncode == 0, rccl.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith('{')][-1])
    j2 = json.loads([l for l in rccl.stdout.splitlines() if l.startswith('{')][-1])
    assert j2['config']['collective'] == 'nccl' and j1['config']['collective'] == 'none'
    assert j1['last_step'] == j2['last_step'] and j1['theta_checksum'] == j2['theta_checksum']
    # ... that run issued the all-reduce group by group on a communication stream under the validation backward (the default with a
    # process group: dist.ChunkedAllReduce); ONE collective after the backward and a single local task (the lane path a rank of the
    # 8-GPU configuration takes) must leave the same bits
    whole = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                            '127.0.0.1', '--master-port', '29743', os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + common,
                           capture_output=True, text=True, env=dict(env, MTL_DIST_ALWAYS='1', MTL_CHUNKED_ALLREDUCE='0'), timeout=300)
    assert whole.returncode == 0, whole.stderr[-2000:]
    j3 = json.loads([l for l in whole.stdout.splitlines() if l.startswith('{')][-1])
    assert j3['theta_checksum'] == j2['theta_checksum']
    one_task = common[:]
    one_task[one_task.index('--tasks') + 1] = '1'
    ck = []
    for chunked in ('1', '0'):
        r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                            '127.0.0.1', '--master-port', '29745', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--no-extras'] + one_task,
                           capture_output=True, text=True, env=dict(env, MTL_DIST_ALWAYS='1', MTL_CHUNKED_ALLREDUCE=chunked), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        ck.append(json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])['theta_checksum'])
    assert ck[0] == ck[1], ck


def test_host_one_iteration_ahead_is_bitwise_equal(capsys):
    """TransientTrainer.train enqueues iteration i + 1 before it resolves iteration i (enqueue_iteration / PendingIteration): the
    log lines (loss, CER, learning rate of every iteration) and theta after 5 iterations are identical to the loop that resolves
    every iteration at once (MTL_PIPELINE=0 behaviour), with one task (single lane) and with three (task lanes)."""
    import re
    z, cfg, spec = gu.load('F0')
    for n_tasks in (1, 3):
        out = {}
        for pipe in (False, True):
            mtl_amd, args, vocab, model = make(cfg, spec, name='pipe')
            model = model.cuda()
            tasks = [mtl_amd.SyntheticTask(m, 2, 64, 8, cfg['vocab_size'], variable=True) for m in range(n_tasks)]
            trainer = mtl_amd.TransientTrainer()
            trainer.pipeline = pipe
            capsys.readouterr()
            trainer.train(model, vocab, tasks, [], 'ce', 0, 5, args, evaluate_every=10 ** 9, early_stop='cer,200', is_copy_grad=True)
            lines = [re.sub(r' TOTAL TIME:.*', '', l) for l in capsys.readouterr().out.splitlines() if l.startswith('(Iteration')]
            assert len(lines) == 5 and [int(l.split(')')[0].split()[1]) for l in lines] == [1, 2, 3, 4, 5]
            out[pipe] = (lines, model.flat_parameters.detach().cpu().clone())
        assert out[False][0] == out[True][0]
        assert torch.equal(out[False][1], out[True][1])


@pytest.mark.parametrize('ragged', [False, True])
def test_uploads_on_their_own_stream_are_bitwise_equal(ragged):
    """Batches handed over in (pinned) host memory go up on a separate stream into alternating input sets, one iteration ahead of
    the kernels (TransientTrainer._batched_iteration, transient_trainer.py:182-184,210-212 are inside the reference's span): theta and
    the per-iteration results after 6 pipelined iterations equal those of the in-stream upload (MTL_OVERLAP_UPLOADS=0 behaviour) and
    those of device-resident inputs, bit for bit -- also across the eager / recording / replay transitions of the command list.
    ragged: every batch at a width of its own, new ones every iteration (a batch lands contiguously at the front of its slab of the
    landing set and is spread into the zero-filled stack on the main stream)."""
    z, cfg, spec = gu.load('F0')
    out = {}
    for mode in ('resident', 'in_stream', 'overlapped'):
        mtl_amd, args, vocab, model = make(cfg, spec, name='upl')
        model = model.cuda()
        inner, outer = mtl_amd.FlatSGD(model, spec['lr']), mtl_amd.FlatAdam(model, 1e-3)
        model.zero_copy_grad()
        tr = mtl_amd.TransientTrainer()
        tr.overlap_uploads = mode == 'overlapped'
        tr.ragged_quantum = 16
        place = (lambda x: x.cuda()) if mode == 'resident' else (lambda x: x.pin_memory())
        pending, results = [], []
        for it in range(6):
            widths = [64 - ((5 * it + 7 * m) % 23) for m in range(4)] if ragged else [64] * 4
            b = [mtl_amd.synth_batch(100 * it + m, 2, widths[m], 8, cfg['vocab_size'], variable=True) for m in range(4)]
            as5 = lambda q: (place(q[0]), q[1], None, q[2], None)
            pending.append(tr.enqueue_iteration(model, vocab, [as5(q) for q in b[:3]], as5(b[3]), 3, inner, outer, args))
            while len(pending) > tr.pipeline_depth:
                results.append(pending.pop(0).result())
        results += [p.result() for p in pending]
        torch.cuda.synchronize()
        out[mode] = (results, model.flat_parameters.detach().cpu().clone())
    for mode in ('in_stream', 'overlapped'):
        assert out[mode][0] == out['resident'][0], mode
        assert torch.equal(out[mode][1], out['resident'][1]), mode


def test_variable_length_batches_keep_the_buffer_pool_and_the_command_lists_bounded():
    """Real manifests bring a new (frames, label width) with almost every batch (collate pads to the batch maximum).  The engines'
    buffer pool is LRU-trimmed against a byte budget and the trainer's command-list / read-back caches are bounded: twelve
    meta-iterations over ever-changing shapes stay under the (here tiny) budget plus the working set of two passes, evictions do
    happen, and an iteration repeated after the evictions reproduces its first result bit for bit (re-recorded command lists, no
    stale addresses)."""
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    as5 = lambda b: (b[0].cuda(), b[1], None, b[2], None)
    inner = mtl_amd.FlatSGD(model, spec['lr'])
    model.zero_copy_grad()
    tr = mtl_amd.TransientTrainer()
    tr.pad_lanes = '0'           # (no rounding to repeating widths here: every iteration is to bring shapes of its own)

    def iteration(T, Lw, seed):
        tasks = [as5(mtl_amd.synth_batch(seed + m, 2, T, Lw, cfg['vocab_size'], variable=True)) for m in range(2)]
        val = as5(mtl_amd.synth_batch(seed + 9, 2, T, Lw, cfg['vocab_size'], variable=True))
        tr.meta_iteration(model, vocab, tasks, val, 2, inner, None, args)
        torch.cuda.synchronize()
        return model._G.clone()

    first = iteration(64, 8, 500)
    eng = model.engine
    one_pass = eng._pool_bytes
    for e in model.engines:
        e.pool_budget = one_pass // 2                    # far below one pass: everything older than two passes must go
    peak, epochs = 0, eng.scratch_epoch
    for i in range(12):
        iteration(64 + 4 * (i % 5) + 4, 5 + i % 4, 600 + 10 * i)
        peak = max(peak, max(e._pool_bytes for e in model.engines))
    assert eng.scratch_epoch > epochs, 'nothing was evicted'
    assert peak < 6 * one_pass, (peak, one_pass)         # without trimming: ~13 distinct shapes' worth
    assert len(tr._cmdlists) <= 64 and len(eng._stage) <= 64 + 4
    again = iteration(64, 8, 500)
    assert torch.equal(first, again)


def test_shared_pool_budget_empties_idle_lane_engines_before_the_working_one():
    """The pool budget is shared by all engines of a device, and an engine only gives up buffers during its OWN pass.  A few iterations
    on the lane-per-task schedule leave buffers on every lane engine; when the schedule then changes to the stacked one (engine 0 only)
    and the shared account is over its budget, the idle lanes' buffers must go first -- not engine 0's own shapes on every pass,
    which would bump its scratch epoch (= re-record the command lists) for ever."""
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    as5 = lambda b: (b[0].cuda(), b[1], None, b[2], None)
    inner = mtl_amd.FlatSGD(model, spec['lr'])
    model.zero_copy_grad()
    tr = mtl_amd.TransientTrainer()
    tr.use_cmdlists = False          # (every pass enqueued call by call, as with shapes that never repeat: a replayed list never trims)
    tasks = [as5(mtl_amd.synth_batch(700 + m, 2, 64, 8, cfg['vocab_size'], variable=True)) for m in range(3)]
    val = as5(mtl_amd.synth_batch(709, 2, 64, 8, cfg['vocab_size'], variable=True))
    tr.batch_tasks = False
    for _ in range(2):
        tr.meta_iteration(model, vocab, tasks, val, 3, inner, None, args)
    torch.cuda.synchronize()
    lanes = [e for e in model.engines[1:] if e._pool_bytes > 0]
    assert lanes, 'the lane schedule did not use a second engine'
    eng = model.engines[0]
    acc = eng.account
    own_lane_bytes = eng._pool_bytes                            # engine 0's buffers of the lane schedule: stale from now on
    tr.batch_tasks = True
    for _ in range(2):                                          # the stacked schedule allocates its own shapes on engine 0
        tr.meta_iteration(model, vocab, tasks, val, 3, inner, None, args)
    torch.cuda.synchronize()
    idle_bytes = sum(e._pool_bytes for e in lanes)
    saved = acc['budget']
    try:
        # over budget by engine 0's own stale buffers AND half of what the idle lanes hold: engine 0 cannot get under on its own
        acc['budget'] = acc['bytes'] - own_lane_bytes - idle_bytes // 2
        epochs = []
        for _ in range(3 * max(8, 4 * len(model.engines))):
            tr.meta_iteration(model, vocab, tasks, val, 3, inner, None, args)
            epochs.append(eng.scratch_epoch)
        torch.cuda.synchronize()
        assert acc['bytes'] <= acc['budget']
        assert sum(e._pool_bytes for e in lanes) < idle_bytes   # the idle lanes paid ...
        assert epochs[-1] == epochs[-8]                         # ... and the working engine's buffers (and command lists) are stable
        assert tr.last_schedule == 'batched'
    finally:
        acc['budget'] = saved


class _CollatedTask:
    """The dataset contract of TransientTrainer.train over a list of utterances, batches formed like the reference's loader: k random
    utterances collated to the batch's OWN longest one (utils/data_loader.py:284-297 = mtl_amd.data.collate)."""

    def __init__(self, seed, n, V, pad_id):
        self.items, self.pad_id = _ListDataset(seed, n, V).items, pad_id
        self.g = torch.Generator().manual_seed(seed + 1)

    def _draw(self, k):
        pick = torch.randperm(len(self.items), generator=self.g)[:k].tolist()
        import mtl_amd
        return mtl_amd.data.collate([self.items[i][0] for i in pick], [self.items[i][1] for i in pick], self.pad_id)

    def sample(self, k_train, k_valid, manifest_id):
        return self._draw(k_train), self._draw(k_valid)


def test_train_loop_on_collated_batches_of_different_widths_equals_the_reference_schedule(tmp_path):
    """TransientTrainer.train fed like the reference (every task's batch collated to its own longest utterance: the frame counts
    differ from task to task and from iteration to iteration).  The default schedule stacks the tasks at the widest in one pass per
    phase; the reference's schedule is a pass per task at its own width (batch_ragged off, no widening): the same losses / CER counts
    per iteration (iterations 2-4 run at parameters the earlier outer steps have moved)."""
    z, cfg, spec = gu.load('F0')
    V = cfg['vocab_size']
    out = {}
    for mode in ('stacked', 'own'):
        mtl_amd, args, vocab, model = make(cfg, spec, name='ragged_' + mode)
        args.save_folder, args.k_train, args.k_valid = str(tmp_path), 3, 3
        model = model.cuda()
        tasks = [_CollatedTask(300 + 7 * m, 12, V, vocab.PAD_ID) for m in range(3)]
        trainer = mtl_amd.TransientTrainer()
        if mode == 'own':
            trainer.batch_ragged, trainer.pad_lanes = False, '0'
        else:
            trainer.ragged_quantum = 8
        trainer.train(model, vocab, tasks, [], 'ce', 0, 4, args, evaluate_every=10 ** 9, early_stop='cer,10', is_copy_grad=True)
        torch.cuda.synchronize()
        out[mode] = (list(trainer.loss_trace), model.flat_parameters.detach().clone(), trainer.last_schedule)
    assert out['stacked'][2] == 'batched-ragged' and out['own'][2] == 'lanes'
    assert len(out['own'][0]) == 4
    for (la, ca, na), (lb, cb, nb) in zip(out['stacked'][0], out['own'][0]):
        assert abs(la - lb) <= 5e-6 * abs(lb) and (ca, na) == (cb, nb), ((la, ca, na), (lb, cb, nb))
    d = float((out['stacked'][1] - out['own'][1]).norm() / out['own'][1].norm())
    print('train() on collated batches: theta after 4 steps, stacked vs own widths: %.2e' % d)
    # (Adam's first steps move every element by ~meta_lr x sign(g): elements whose exact gradient is zero -- the key-projection biases --
    # follow the sign of rounding noise, which the two schedules do not share; the losses of iterations 2-4 above, computed at the moved
    # parameters, are what shows that the steps agree)
    assert d < 5e-4
