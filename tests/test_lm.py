"""SURVEY 8(f) f3 -- LM meta-transfer path.  CPU: the oracle's model / dataset restatement against the golden produced by the REAL
reference classes (tests/golden/L0.npz); GPU: the HIP path against the oracle (the meta loop itself is parity-unpinned: the
reference's loop raises on torch >= 2, see oracle/lm_refimpl.py)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import lm_refimpl as LR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_l0():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'L0.npz'))
    return z, json.loads(bytes(z['spec']).decode())


def test_lm_oracle_model_and_dataset_match_the_reference_golden():
    z, spec = load_l0()
    torch.manual_seed(spec['seed'])
    torch.set_num_threads(8)
    m = LR.RNNModel(spec['ntoken'], spec['ninp'], spec['nhid'], spec['nlayers'], 0.0)
    assert [n for n, _ in m.named_parameters()] == [str(s) for s in z['param_names']]
    h = hashlib.sha256()
    for _, p in m.named_parameters():
        h.update(p.detach().numpy().tobytes())
    assert h.hexdigest() == bytes(z['theta0_sha256']).decode()             # same RNG draw order as lm/model/rnn_model.py
    streams = [LR.synth_corpus(s, spec['ntoken'], spec['corpus_len']) for s in spec['corpus_seeds']]
    tasks = [LR.batchify(s, spec['batch_size']) for s in streams]
    for i in range(3):
        assert np.array_equal(tasks[i].numpy(), z['batchified/%d' % i])
        for it in (0, spec['it'], 57):
            got = LR.sample(tasks, i if i < 2 else -1, it, spec['bptt'])
            for k, v in zip(('tr_x', 'tr_y', 'va_x', 'va_y'), got):
                assert np.array_equal(v.numpy(), z['sample/%d/%d/%s' % (i, it, k)]), (i, it, k)
    x, y, _, _ = LR.sample(tasks, 0, spec['it'], spec['bptt'])
    hidden = (torch.from_numpy(z['h0']), torch.from_numpy(z['c0']))
    out, (hn, cn) = m(x, hidden)
    loss = torch.nn.functional.cross_entropy(out.view(-1, spec['ntoken']), y)
    loss.backward()
    # the step-by-step LSTM of the oracle against nn.LSTM's fused CPU kernel inside the reference model: fp32 rounding only
    assert float((out - torch.from_numpy(z['out'])).abs().max()) < 2e-6
    assert abs(float(loss) - float(z['loss'])) < 1e-6
    assert float((hn - torch.from_numpy(z['hn'])).abs().max()) < 1e-6 and float((cn - torch.from_numpy(z['cn'])).abs().max()) < 1e-6
    for n, p in m.named_parameters():
        ref = torch.from_numpy(z['grad/' + n])
        assert float((p.grad - ref).norm()) <= 2e-5 * float(ref.norm()) + 1e-9, n


def test_lm_meta_step_first_order_definition():
    """the documented first-order reading (oracle/lm_refimpl.py docstring) is what meta_step computes: re-derived here from its
    definition with independent code -- weights (1-r)/2, (1-r)/2, r; hidden carried through the train forwards only; clipping of the
    inner and of the outer gradient; plain SGD outer update"""
    torch.manual_seed(3)
    m = LR.RNNModel(60, 16, 16, 2, 0.0)
    tasks = [LR.batchify(LR.synth_corpus(s, 60, 211), 3) for s in (1, 2, 3)]
    it, bptt, lr, fac, clip, ratio = 4, 5, 2.0, 3.0, 0.25, 0.8
    batches = [LR.sample(tasks, i, it, bptt)[:2] for i in range(3)]
    val = LR.sample(tasks, -1, it, bptt)[2:]
    theta0 = [p.detach().clone() for p in m.parameters()]
    hidden = m.init_hidden(3)
    # independent evaluation
    G = [torch.zeros_like(p) for p in theta0]
    hid = hidden
    for i, (x, y) in enumerate(batches):
        out, hid = m(x, hid)
        g = torch.autograd.grad(torch.nn.functional.cross_entropy(out.view(-1, 60), y), list(m.parameters()))
        tot = torch.sqrt(sum((t ** 2).sum() for t in g))
        coef = min(1.0, clip / (float(tot) + 1e-6))
        with torch.no_grad():
            for p, t in zip(m.parameters(), g):
                p.sub_(lr / fac * coef * t)
        hid = tuple(h.detach() for h in hid)
        out, _ = m(val[0], hid)
        gv = torch.autograd.grad(torch.nn.functional.cross_entropy(out.view(-1, 60), val[1]), list(m.parameters()))
        wi = ratio if i == 2 else (1 - ratio) / 2
        with torch.no_grad():
            for a, t in zip(G, gv):
                a.add_(wi * t)
            for p, t0 in zip(m.parameters(), theta0):
                p.copy_(t0)
    tot = torch.sqrt(sum((t ** 2).sum() for t in G))
    coef = min(1.0, clip / (float(tot) + 1e-6))
    expect = [t0 - lr * coef * g for t0, g in zip(theta0, G)]
    G2, hid2, trl, val_l = LR.meta_step(m, hidden, batches, val, lr, fac, clip, ratio)
    for p, e in zip(m.parameters(), expect):
        assert float((p - e).abs().max()) < 1e-6
    assert torch.equal(hid2[0], hid[0]) and len(trl) == 3 and len(val_l) == 3
    assert LR.task_weights(3, 0.8) == [pytest.approx(0.1), pytest.approx(0.1), 0.8]


# ------------------------------------------------------------------------------------------------------------------ GPU
def _to_oracle(oracle, model, flat):
    with torch.no_grad():
        for n, p in oracle.named_parameters():
            p.copy_(model._layout.view(flat, n).cpu())


def _errs(model, flat_g, oracle, grads):
    gn = float(torch.sqrt(sum((t.double() ** 2).sum() for t in grads)))
    return {n: float((model._layout.view(flat_g, n).cpu() - t).norm() / max(float(t.norm()), 1e-4 * gn))
            for (n, _), t in zip(oracle.named_parameters(), grads)}


@pytest.mark.gpu
@pytest.mark.parametrize('dropout', [0.0, 0.3])
def test_lm_pass_matches_oracle(dropout):
    """one forward + backward of the LSTM LM through the HIP library against the oracle at identical parameters and carried
    state: logits 1e-5, loss 1e-6, new hidden state, every gradient tensor 1e-4; with dropout the Philox keep-masks the device
    drew (embedding, between the layers, output) are replayed in the oracle"""
    import mtl_amd
    z, spec = load_l0()
    torch.manual_seed(spec['seed'])
    model = mtl_amd.lm.RNNModel('LSTM', 300, 48, 64, 2, dropout)
    h = hashlib.sha256()
    torch.manual_seed(spec['seed'])
    ref_init = mtl_amd.lm.RNNModel('LSTM', spec['ntoken'], spec['ninp'], spec['nhid'], spec['nlayers'], 0.0)
    for _, p in ref_init.named_parameters():
        h.update(p.detach().numpy().tobytes())
    assert h.hexdigest() == bytes(z['theta0_sha256']).decode()             # product init == reference init, bit for bit
    with pytest.raises(RuntimeError, match='no CPU'):
        model(torch.zeros(3, 2, dtype=torch.int64), model.init_hidden(2))
    model = model.cuda()
    model.train()
    torch.manual_seed(5)
    oracle = LR.RNNModel(300, 48, 64, 2, dropout)
    _to_oracle(oracle, model, model.flat_parameters)
    T, B = 9, 5
    g = torch.Generator().manual_seed(21)
    x = torch.randint(0, 300, (T, B), generator=g)
    x[3] = x[0]                                                            # repeated ids: the deterministic scatter-add chains
    y = torch.randint(0, 300, (T * B,), generator=g)
    h0, c0 = 0.2 * torch.randn(2, B, 64, generator=g), 0.2 * torch.randn(2, B, 64, generator=g)
    eng = model.engine
    out = eng.forward(model.flat_parameters, x.cuda(), y.cuda(), (h0.cuda(), c0.cuda()), dropout)
    grad = torch.zeros_like(model.flat_grad)
    eng.backward(grad, 1.0)
    masks = None
    if dropout > 0:
        sc = 1.0 / (1 - dropout)
        pool = {k[0]: v for k, v in eng.pool.items()}
        masks = {'emb': pool['m_emb'].cpu().float().view(T, B, 48) * sc, 'l0': pool['m_l0'].cpu().float().view(T, B, 64) * sc,
                 'out': pool['m_l1'].cpu().float().view(T, B, 64) * sc}
        assert abs(float(masks['emb'].gt(0).float().mean()) - 0.7) < 0.05
    o_out, (hn, cn) = oracle(x, (h0, c0), masks)
    loss = torch.nn.functional.cross_entropy(o_out.view(-1, 300), y)
    grads = torch.autograd.grad(loss, list(oracle.parameters()))
    assert float((out['logits'].cpu() - o_out.view(-1, 300)).norm() / o_out.norm()) < 1e-5
    assert abs(float(out['loss']) - float(loss)) < 1e-6 * float(loss)
    assert float((out['hidden'][0].cpu() - hn).abs().max()) < 1e-6 and float((out['hidden'][1].cpu() - cn).abs().max()) < 1e-6
    errs = _errs(model, grad, oracle, grads)
    assert max(errs.values()) < 1e-4, max(errs.items(), key=lambda kv: kv[1])
    grad2 = torch.zeros_like(grad)                                          # linear in the loss scale, accumulating, deterministic
    eng.backward(grad2, 0.5)
    eng.backward(grad2, 0.5)
    assert float((grad - grad2).norm() / grad.norm()) < 1e-6


@pytest.mark.gpu
def test_lm_meta_step_matches_oracle_restatement():
    """two iterations of LMMetaTrainer (3 tasks, hidden state carried, clipped inner and outer steps, weights 0.1 / 0.1 / 0.8) against
    oracle.lm_refimpl.meta_step: losses, the meta-gradient of the second iteration and the parameters after both (PARITY
    UNPINNED against the reference loop, which does not run on torch >= 2 -- see the oracle's header)"""
    import argparse
    import mtl_amd
    torch.manual_seed(11)
    model = mtl_amd.lm.RNNModel('LSTM', 200, 32, 32, 2, 0.0).cuda()
    model.train()
    oracle = LR.RNNModel(200, 32, 32, 2, 0.0)
    _to_oracle(oracle, model, model.flat_parameters)
    args = argparse.Namespace(bptt=6, batch_size=4, cuda=False)
    streams = [LR.synth_corpus(s, 200, 331) for s in (1, 2, 3)]
    ds = mtl_amd.lm.LMDataset(streams, args)
    otasks = [LR.batchify(s, 4) for s in streams]
    tr = mtl_amd.lm.LMMetaTrainer(model, lr=2.0, meta_lr_factor=3.0, clip=0.25, ratio=0.8)
    hid = oracle.init_hidden(4)
    for it in (0, 1):
        vb = ds.sample(-1, it)[2:]
        batches = [ds.sample(i, it)[:2] for i in range(3)]
        for i in range(3):                                                  # dataset restatement == oracle's
            for a, b in zip(ds.sample(i, it), LR.sample(otasks, i, it, 6)):
                assert torch.equal(a, b)
        loss, trl = tr.run_iteration([(x.cuda(), y.cuda()) for x, y in batches], (vb[0].cuda(), vb[1].cuda()))
        G_r, hid, trl_r, val_r = LR.meta_step(oracle, hid, batches, vb, 2.0, 3.0, 0.25, 0.8)
        for a, b in zip(trl, trl_r):
            assert abs(a - b) < 1e-5 * b
        assert abs(loss - sum(w * v for w, v in zip(LR.task_weights(3, 0.8), val_r))) < 1e-5 * loss
        errs = _errs(model, tr.G, oracle, G_r)
        assert max(errs.values()) < 1e-4, (it, max(errs.items(), key=lambda kv: kv[1]))
        for n, p in oracle.named_parameters():
            q = model._layout.view(model.flat_parameters, n).cpu()
            assert float((q - p).abs().max()) < 2e-5 * max(float(p.abs().max()), 1e-3), (it, n)
        assert float((tr.hidden[0].cpu() - hid[0]).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('H,B,T,dropout,NL,E', [(128, 5, 9, 0.0, 2, 48), (256, 7, 6, 0.3, 3, 48), (512, 20, 35, 0.2, 2, 48), (384, 32, 4, 0.0, 2, 48),
                                                (512, 20, 1, 0.0, 2, 48), (256, 3, 5, 0.5, 1, 48), (128, 6, 7, 0.3, 2, 128), (256, 4, 5, 0.0, 3, 256),
                                                (512, 20, 35, 0.2, 2, 512)])
def test_lm_persistent_lstm_layer_kernels(H, B, T, dropout, NL, E):
    """csrc/mtl_lstm.hip -- the layer stack as one wavefront launch per direction ('stack': layers one step apart, per-step hand-offs
    within and between layers) and one launch per layer and direction ('layers') -- against the oracle and against the per-step path
    (recurrent product + cell kernel per step) on the same batch, parameters, carried state and dropout masks: logits, loss, new
    hidden state, every gradient tensor; each persistent mode runs twice to show the launch is reproducible bit for bit
    (fixed-order reductions, re-zeroed arrival counters) and that no wait timed out."""
    import mtl_amd
    # E == H: the weight-gradient inputs share one arena and all 2 NL weight gradients are ONE batched product; the last case is the
    # bench configuration's layer shapes with a vocabulary deep enough (K = 5120) for the decoder's dX to take the split-K form
    V = 5120 if E == 512 else 300
    torch.manual_seed(3)
    model = mtl_amd.lm.RNNModel('LSTM', V, E, H, NL, dropout).cuda()
    model.train()
    torch.manual_seed(5)
    oracle = LR.RNNModel(V, E, H, NL, dropout)
    _to_oracle(oracle, model, model.flat_parameters)
    g = torch.Generator().manual_seed(21 + H)
    x = torch.randint(0, V, (T, B), generator=g)
    y = torch.randint(0, V, (T * B,), generator=g)
    h0, c0 = 0.2 * torch.randn(NL, B, H, generator=g), 0.2 * torch.randn(NL, B, H, generator=g)
    eng = model.engine
    assert eng.persistent and eng.stacked and bool(eng.lib.mtl_lstm_layer_supported(B, H)) and bool(eng.lib.mtl_lstm_stack_supported(B, H, NL))
    runs = {}
    for mode in ('stack', 'stack', 'layers', 'layers', 'steps'):
        eng.persistent, eng.stacked = mode != 'steps', mode == 'stack'
        torch.manual_seed(77)                                               # same Philox seed draw -> same keep-masks
        out = eng.forward(model.flat_parameters, x.cuda(), y.cuda(), (h0.cuda(), c0.cuda()), dropout)
        grad = torch.zeros_like(model.flat_grad)
        eng.backward(grad, 1.0)
        torch.cuda.synchronize()
        res = (out['logits'].clone(), float(out['loss']), out['hidden'][0].clone(), out['hidden'][1].clone(), grad)
        assert eng.saved['stacked'] == (mode == 'stack' and NL > 1) and (eng.saved['arena'] is not None) == (E == H and NL > 1)
        if mode in runs:
            for a, b in zip(res, runs[mode]):
                assert (a == b) if isinstance(a, float) else torch.equal(a, b)
        runs[mode] = res
        assert int(eng.sync_ws[1]) == 0, 'a grid-wide wait timed out'
    eng.persistent = eng.stacked = True
    masks = None
    if dropout > 0:
        sc = 1.0 / (1 - dropout)
        pool = {k[0]: v for k, v in eng.pool.items()}
        masks = {'emb': pool['m_emb'].cpu().float().view(T, B, E) * sc}
        for l in range(NL):
            masks['l%d' % l if l < NL - 1 else 'out'] = pool['m_l%d' % l].cpu().float().view(T, B, H) * sc
    o_out, (hn, cn) = oracle(x, (h0, c0), masks)
    loss = torch.nn.functional.cross_entropy(o_out.view(-1, V), y)
    grads = torch.autograd.grad(loss, list(oracle.parameters()))
    ls = runs['steps']
    for mode in ('stack', 'layers'):
        lp = runs[mode]
        assert float((lp[0].cpu() - o_out.view(-1, V)).norm() / o_out.norm()) < 1e-5, mode
        assert abs(lp[1] - float(loss)) < 1e-6 * float(loss) and abs(lp[1] - ls[1]) < 1e-6 * ls[1], mode
        assert float((lp[2].cpu() - hn).abs().max()) < 2e-6 and float((lp[3].cpu() - cn).abs().max()) < 2e-6, mode
        errs = _errs(model, lp[4], oracle, grads)
        assert max(errs.values()) < 1e-4, (mode, max(errs.items(), key=lambda kv: kv[1]))
        assert float((lp[4] - ls[4]).norm() / ls[4].norm()) < 2e-6, mode      # the device paths: same arithmetic, other summation orders
    # the error word is sticky across launches (they re-zero only their counters / flags) and surfaces as an exception
    eng.check_handoff()
    eng.sync_ws[1] = 1
    eng.forward(model.flat_parameters, x.cuda(), y.cuda(), (h0.cuda(), c0.cuda()), dropout)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='timed out'):
        eng.check_handoff()
    eng.check_handoff()
