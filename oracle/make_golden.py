"""ORACLE tooling (test infrastructure): generate tests/golden/*.npz by running the REAL reference.

Runs only in the build container, where /root/reference exists.  It imports the reference's own
Python modules (never copies them), drives `TransientTrainer.train(..., is_copy_grad=True)` on seeded
synthetic batches and stores inputs + expected outputs as data fixtures.  Recipe: SURVEY.md 8(c).

    python oracle/make_golden.py            # F0 (tiny), F1 (small-real), J0 (joint), B0 (beam search), G0 (greedy search)
    python oracle/make_golden.py --ns       # additionally the north-star-size checksum record (~1 min)
"""
import argparse
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden')


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[-1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def bootstrap_reference():
    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
    stub('Levenshtein', distance=_edit_distance)
    stub('stanfordcorenlp', StanfordCoreNLP=object)
    stub('torchaudio')
    stub('transformers', BertModel=object)
    sys.path.insert(0, '/root/reference')
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self      # SURVEY Q5 workaround (args.cuda=True on CPU)
    return torch


def tensor_digest(t, full_below=2048, nsample=256):
    """sum / l2 in fp64, plus the whole tensor when small, else a strided sample."""
    a = t.detach().cpu().numpy().astype(np.float32).reshape(-1)
    d = {'sum': np.float64(a.astype(np.float64).sum()), 'l2': np.float64(np.sqrt((a.astype(np.float64) ** 2).sum())),
         'numel': np.int64(a.size)}
    if a.size <= full_below:
        d['full'] = a.copy()
    else:
        step = max(a.size // nsample, 1)
        d['sample'] = a[::step][:nsample].copy()
        d['step'] = np.int64(step)
    return d


NSAMPLE = {'n': 256}     # strided samples kept per large tensor (the north-star record stores 4096)


def pack(prefix, named, store):
    for name, t in named:
        for k, v in tensor_digest(t, nsample=NSAMPLE['n']).items():
            store['%s/%s/%s' % (prefix, name, k)] = v


FIXTURES = {
    # tiny: exercises Q2 (length 10 of 64 -> masks on the 16-wide pooled axis) and target padding
    'F0': dict(cfg=dict(num_enc_layers=1, num_dec_layers=1, num_heads=8, dim_model=128, dim_key=16, dim_value=16,
                        dim_inner=128, dim_emb=128, src_max_len=500, tgt_max_len=100, r=100, vocab_size=64),
               k=2, T=64, L=8, n_tasks=3, lr=1e-2, meta_lr=1e-3, iters=2, variable=True),
    # small-real: the north-star architecture on short inputs
    'F1': dict(cfg=dict(num_enc_layers=2, num_dec_layers=4, num_heads=8, dim_model=512, dim_key=64, dim_value=64,
                        dim_inner=512, dim_emb=512, src_max_len=500, tgt_max_len=100, r=100, vocab_size=3765),
               k=2, T=64, L=8, n_tasks=3, lr=1e-4, meta_lr=1e-4, iters=1, variable=True),
    # north-star size, checksum-only
    'NS': dict(cfg=dict(num_enc_layers=2, num_dec_layers=4, num_heads=8, dim_model=512, dim_key=64, dim_value=64,
                        dim_inner=512, dim_emb=512, src_max_len=5000, tgt_max_len=2500, r=100, vocab_size=3765),
               k=8, T=1000, L=100, n_tasks=3, lr=1e-4, meta_lr=1e-4, iters=1, variable=False),
    # BASELINE.json configs[3] at its FULL batch (src-max-len 5000, 8 utterances), checksum-only: one task = one training pass at theta0 and
    # one validation pass at theta' (`--t5000`; ~10 GB of autograd state per pass on the CPU, a few minutes)
    'T5': dict(cfg=dict(num_enc_layers=2, num_dec_layers=4, num_heads=8, dim_model=512, dim_key=64, dim_value=64,
                        dim_inner=512, dim_emb=512, src_max_len=5000, tgt_max_len=2500, r=100, vocab_size=3765),
               k=8, T=5000, L=100, n_tasks=1, lr=1e-4, meta_lr=1e-4, iters=1, variable=True),
}


def run_fixture(name, spec, torch):
    from utils.data import Vocab
    from utils.functions import init_transformer_model
    from trainer.asr.transient_trainer import TransientTrainer
    sys.path.insert(0, ROOT)
    from oracle.refimpl import synth_batch

    cfg = spec['cfg']
    vocab = Vocab()
    for i in range(cfg['vocab_size'] - 4):
        ch = chr(0x4e00 + i)
        vocab.add_token(ch)
        vocab.add_label(ch)
    args = argparse.Namespace(
        feat_extractor='vgg_cnn', sample_rate=16000, window_size=.02, feat='spectrogram', dim_input=161,
        num_enc_layers=cfg['num_enc_layers'], num_dec_layers=cfg['num_dec_layers'], num_heads=cfg['num_heads'],
        dim_model=cfg['dim_model'], dim_key=cfg['dim_key'], dim_value=cfg['dim_value'], dim_inner=cfg['dim_inner'],
        dim_emb=cfg['dim_emb'], src_max_len=cfg['src_max_len'], tgt_max_len=cfg['tgt_max_len'], dropout=0.0,
        emb_trg_sharing=False, label_smoothing=0.0, name='golden_' + name, lr=spec['lr'], meta_lr=spec['meta_lr'],
        k_train=spec['k'], k_valid=spec['k'], cuda=True, clip=False, max_norm=400, save_every=10 ** 9,
        save_folder='/tmp/golden_ckpt')
    torch.manual_seed(123456)
    np.random.seed(123456)
    torch.set_num_threads(8)
    model = init_transformer_model(args, vocab, is_factorized=False, r=cfg['r'])

    store = {}
    names = [n for n, _ in model.named_parameters()]
    h = hashlib.sha256()
    for _, p in model.named_parameters():
        h.update(p.detach().numpy().tobytes())
    store['theta0_sha256'] = np.frombuffer(h.hexdigest().encode(), dtype=np.uint8)
    pack('theta0', model.named_parameters(), store)

    n, k, T, L = spec['n_tasks'], spec['k'], spec['T'], spec['L']

    class FakeTask:
        """duck-types SpectrogramDataset.sample (utils/data_loader.py:245-321)"""

        def __init__(self, task):
            self.task, self.calls = task, 0

        def sample(self, k_train, k_valid, manifest_id):
            it = self.calls
            self.calls += 1
            out = []
            for part in (0, 1):
                x, lens, y = synth_batch(1000 * it + 10 * self.task + part, k, T, L, cfg['vocab_size'],
                                         variable=spec['variable'])
                tl = (y != 0).sum(1).to(torch.int32)
                out.append((x, lens, lens.float() / T, y, tl))
            return tuple(out)

    tasks = [FakeTask(m) for m in range(n)]

    fwd_log = []

    def fwd_hook(mod, inp, outp):
        pred, gold, hyp = outp
        import torch.nn.functional as F
        loss = F.cross_entropy(pred.detach().view(-1, pred.size(2)), gold.view(-1), ignore_index=0, reduction='mean')
        fwd_log.append((pred.detach().clone(), gold.clone(), hyp.clone(), float(loss)))
    model.register_forward_hook(fwd_hook)

    G_log, theta_log, cer_log = [], [], []
    orig_from = model.from_copy_grad

    def from_hook():
        G_log.append([g.clone() for g in model.copy_grad])
        orig_from()
    model.from_copy_grad = from_hook

    trainer = TransientTrainer()
    orig_fob = trainer.forward_one_batch

    def fob(*a, **kw):
        loss, cer, nchar = orig_fob(*a, **kw)
        cer_log.append((int(cer), int(nchar)))
        return loss, cer, nchar
    trainer.forward_one_batch = fob

    # one call per iteration so theta can be snapshotted in between (optimizers persist)
    inner = torch.optim.SGD(model.parameters(), lr=args.lr)
    outer = torch.optim.Adam(model.parameters(), lr=args.meta_lr)
    os.makedirs('log', exist_ok=True)
    for it in range(spec['iters']):
        trainer.train(model, vocab, tasks, [], 'ce', it, it + 1, args, inner_opt=inner, outer_opt=outer,
                      evaluate_every=10 ** 9, early_stop='cer,200', is_copy_grad=True)
        theta_log.append([p.detach().clone() for p in model.parameters()])
    # the trainer's prefetch thread runs one sample() ahead per train() call; data seeds are keyed on the
    # per-task call counter, so record which `it` seeds each iteration actually consumed.
    assert len(G_log) == spec['iters'] and len(fwd_log) == 2 * n * spec['iters']

    store['cfg_keys'] = np.array(sorted(cfg.keys()))
    store['cfg_vals'] = np.array([cfg[k_] for k_ in sorted(cfg.keys())], dtype=np.int64)
    store['spec'] = np.array([spec['k'], spec['T'], spec['L'], spec['n_tasks'], spec['iters'], int(spec['variable'])],
                             dtype=np.int64)
    store['lr'] = np.float64(spec['lr'])
    store['meta_lr'] = np.float64(spec['meta_lr'])
    store['param_names'] = np.array(names)
    # data-call index consumed by iteration `it`: trainer.train() prefetches once before the loop and once inside,
    # so each train() call draws 2 samples per task and uses the first.
    store['data_call_index'] = np.array([2 * it for it in range(spec['iters'])], dtype=np.int64)
    for it in range(spec['iters']):
        pack('G/%d' % it, zip(names, G_log[it]), store)
        pack('theta/%d' % (it + 1), zip(names, theta_log[it]), store)
        for j in range(2 * n):
            pred, gold, hyp, loss = fwd_log[it * 2 * n + j]
            key = 'fwd/%d/%d' % (it, j)
            store[key + '/gold'] = gold.numpy().astype(np.int64)
            store[key + '/hyp'] = hyp.numpy().astype(np.int64)
            store[key + '/loss'] = np.float64(loss)
            store[key + '/cer'] = np.array(cer_log[it * 2 * n + j], dtype=np.int64)
            for k_, v in tensor_digest(pred, nsample=NSAMPLE['n']).items():
                store[key + '/pred/' + k_] = v
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **store)
    print(name, 'written:', len(store), 'arrays; losses', [round(f[3], 6) for f in fwd_log])


def run_joint_fixture(torch):
    """BASELINE.json configs[0]: joint_train.py semantics (JointTrainer), enc1/dec1 d128, 3 tasks, k_train=2, T=200, L=20."""
    from utils.data import Vocab
    from utils.functions import init_transformer_model
    from trainer.asr.joint_trainer import JointTrainer
    sys.path.insert(0, ROOT)
    from oracle.refimpl import synth_batch
    cfg = FIXTURES['F0']['cfg']
    spec = dict(k=2, T=200, L=20, n_tasks=3, lr=1e-3, iters=2)
    vocab = Vocab()
    for i in range(cfg['vocab_size'] - 4):
        vocab.add_token(chr(0x4e00 + i))
        vocab.add_label(chr(0x4e00 + i))
    args = argparse.Namespace(
        feat_extractor='vgg_cnn', sample_rate=16000, window_size=.02, feat='spectrogram', dim_input=161,
        num_enc_layers=cfg['num_enc_layers'], num_dec_layers=cfg['num_dec_layers'], num_heads=cfg['num_heads'],
        dim_model=cfg['dim_model'], dim_key=cfg['dim_key'], dim_value=cfg['dim_value'], dim_inner=cfg['dim_inner'],
        dim_emb=cfg['dim_emb'], src_max_len=cfg['src_max_len'], tgt_max_len=cfg['tgt_max_len'], dropout=0.0,
        emb_trg_sharing=False, label_smoothing=0.0, name='golden_J0', lr=spec['lr'], k_train=spec['k'], cuda=False, clip=False,
        max_norm=400, save_every=10 ** 9, save_folder='/tmp/golden_ckpt')
    torch.manual_seed(123456)
    torch.set_num_threads(8)
    model = init_transformer_model(args, vocab, is_factorized=False, r=cfg['r'])
    names = [n for n, _ in model.named_parameters()]

    class FakeTask:
        def __init__(self, task):
            self.task, self.calls = task, 0

        def sample(self, k_train, k_valid, manifest_id):
            it = self.calls
            self.calls += 1
            out = []
            for part in (0, 1):
                x, lens, y = synth_batch(1000 * it + 10 * self.task + part, spec['k'], spec['T'], spec['L'], cfg['vocab_size'], True)
                out.append((x, lens, lens.float() / spec['T'], y, (y != 0).sum(1).to(torch.int32)))
            return tuple(out)

    tasks = [FakeTask(m) for m in range(spec['n_tasks'])]
    grads, thetas, losses = [], [], []
    orig_step = torch.optim.Adam.step

    def spy_step(self_opt, *a, **kw):
        grads.append([p.grad.detach().clone() for p in model.parameters()])
        return orig_step(self_opt, *a, **kw)
    torch.optim.Adam.step = spy_step
    fwd = []
    model.register_forward_hook(lambda m, i, o: fwd.append((o[1].clone(), o[2].clone(),
                                float(torch.nn.functional.cross_entropy(o[0].detach().view(-1, o[0].size(2)), o[1].view(-1), ignore_index=0)))))
    # JointTrainer builds a fresh Adam per train() call, so both iterations run inside ONE call (data calls 0 and 1)
    JointTrainer().train(model, vocab, tasks, [], 'ce', 0, spec['iters'], args, evaluate_every=10 ** 9, early_stop='cer,200')
    torch.optim.Adam.step = orig_step
    assert len(grads) == spec['iters'] and len(fwd) == spec['iters'] * spec['n_tasks']
    store = {'param_names': np.array(names), 'lr': np.float64(spec['lr']),
             'spec': np.array([spec['k'], spec['T'], spec['L'], spec['n_tasks'], spec['iters'], 1], dtype=np.int64),
             'cfg_keys': np.array(sorted(cfg.keys())), 'cfg_vals': np.array([cfg[k_] for k_ in sorted(cfg.keys())], dtype=np.int64),
             'meta_lr': np.float64(spec['lr']), 'data_call_index': np.arange(spec['iters'], dtype=np.int64)}
    for it in range(spec['iters']):
        pack('G/%d' % it, zip(names, grads[it]), store)
        for j in range(spec['n_tasks']):
            gold, hyp, loss = fwd[it * spec['n_tasks'] + j]
            store['fwd/%d/%d/gold' % (it, j)] = gold.numpy().astype(np.int64)
            store['fwd/%d/%d/hyp' % (it, j)] = hyp.numpy().astype(np.int64)
            store['fwd/%d/%d/loss' % (it, j)] = np.float64(loss)
    pack('theta/final', model.named_parameters(), store)
    np.savez_compressed(os.path.join(OUT, 'J0.npz'), **store)
    print('J0 written; losses', [round(f[2], 6) for f in fwd])


def run_beam_fixture(torch):
    """Decoder.beam_search / Transformer.evaluate(beam_search=True) of the reference on the F0 model whose
    vocabulary projection is perturbed (seeded) so that hypotheses differ and some end with a natural EOS.  Stores the n-best
    id sequences and strings of 3 utterances (beam 3, n-best 2).  (The reference's greedy search needs tgt_max_len > 300.)"""
    from utils.data import Vocab
    from utils.functions import init_transformer_model
    sys.path.insert(0, ROOT)
    from oracle.refimpl import synth_batch
    cfg = FIXTURES['F0']['cfg']
    vocab = Vocab()
    for i in range(cfg['vocab_size'] - 4):
        vocab.add_token(chr(0x4e00 + i))
        vocab.add_label(chr(0x4e00 + i))
    args = argparse.Namespace(
        feat_extractor='vgg_cnn', sample_rate=16000, window_size=.02, feat='spectrogram', dim_input=161,
        num_enc_layers=cfg['num_enc_layers'], num_dec_layers=cfg['num_dec_layers'], num_heads=cfg['num_heads'],
        dim_model=cfg['dim_model'], dim_key=cfg['dim_key'], dim_value=cfg['dim_value'], dim_inner=cfg['dim_inner'],
        dim_emb=cfg['dim_emb'], src_max_len=cfg['src_max_len'], tgt_max_len=cfg['tgt_max_len'], dropout=0.0,
        emb_trg_sharing=False, label_smoothing=0.0, name='golden_B0', cuda=False, beam_width=3, beam_nbest=2)
    torch.manual_seed(123456)
    torch.set_num_threads(8)
    model = init_transformer_model(args, vocab, is_factorized=False, r=cfg['r'])
    g = torch.Generator().manual_seed(7)
    W = model.decoder.output_linear.weight
    W.data += 0.5 * torch.randn(W.shape, generator=g)
    W.data[2] = 1.02 * W.data[47]                               # EOS row ~ a token that dominates late positions: natural terminations
    model.eval()
    x, lens, y = synth_batch(500, 3, 64, 8, cfg['vocab_size'], variable=True)
    with torch.no_grad():
        _, strs_beam, strs_gold = model.evaluate(x, lens, y, args, beam_search=True, start_token=vocab.SOS_ID)
        f = model.conv(x)
        sz = f.size()
        enc, _ = model.encoder(f.view(sz[0], sz[1] * sz[2], sz[3]).transpose(1, 2).contiguous(), lens)
        ids, strs = model.decoder.beam_search(enc, args, beam_width=3, nbest=2, start_token=vocab.SOS_ID)
    width = max(len(r) for r in ids)
    arr = np.full((len(ids), width), -1, dtype=np.int64)
    for i, r in enumerate(ids):
        arr[i, :len(r)] = r
    enc_str = lambda lst: np.frombuffer('\n'.join(lst).encode('utf-8'), dtype=np.uint8)
    store = dict(beam_ids=arr, beam_strs=enc_str(strs), eval_beam_strs=enc_str(strs_beam), gold_strs=enc_str(strs_gold),
                 spec=np.frombuffer(json.dumps(dict(seed=500, k=3, T=64, L=8, beam_width=3,
                                                                                         nbest=2, noise_seed=7, noise=0.5,
                                                                                         eos_from=47, eos_gain=1.02)).encode(), dtype=np.uint8))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'B0.npz'), **store)
    print('B0: %d hypotheses, lengths %s, natural EOS in %d' % (len(ids), [len(r) for r in ids],
                                                                  sum(len(r) < 18 for r in ids)))
    for st in strs:
        print('   ', repr(st))


def run_greedy_fixture(torch):
    """Decoder.greedy_search / Transformer.evaluate(beam_search=False) of the REAL reference (300 fixed arg-max
    steps, whole decoder re-run on the growing prefix, modules/decoder.py:131-185) on the F0 model built with tgt_max_len = 320
    (the positional table must cover 301 positions) and the B0 perturbation of the vocabulary projection, so the three
    utterances end with a natural EOS at different steps.  Stores the token ids of all 300 steps (captured at the arg-max of
    output_linear's last position, which is what greedy_search takes) and the returned strings."""
    from utils.data import Vocab
    from utils.functions import init_transformer_model
    sys.path.insert(0, ROOT)
    from oracle.refimpl import synth_batch
    cfg = dict(FIXTURES['F0']['cfg'], tgt_max_len=320)
    vocab = Vocab()
    for i in range(cfg['vocab_size'] - 4):
        vocab.add_token(chr(0x4e00 + i))
        vocab.add_label(chr(0x4e00 + i))
    args = argparse.Namespace(
        feat_extractor='vgg_cnn', sample_rate=16000, window_size=.02, feat='spectrogram', dim_input=161,
        num_enc_layers=cfg['num_enc_layers'], num_dec_layers=cfg['num_dec_layers'], num_heads=cfg['num_heads'],
        dim_model=cfg['dim_model'], dim_key=cfg['dim_key'], dim_value=cfg['dim_value'], dim_inner=cfg['dim_inner'],
        dim_emb=cfg['dim_emb'], src_max_len=cfg['src_max_len'], tgt_max_len=cfg['tgt_max_len'], dropout=0.0,
        emb_trg_sharing=False, label_smoothing=0.0, name='golden_G0', cuda=False)
    torch.manual_seed(123456)
    torch.set_num_threads(8)
    model = init_transformer_model(args, vocab, is_factorized=False, r=cfg['r'])
    spec = dict(seed=321, k=3, T=72, L=6, noise_seed=7, noise=0.5, eos_from=47, eos_gain=1.02, tgt_max_len=320, steps=300)
    g = torch.Generator().manual_seed(spec['noise_seed'])
    W = model.decoder.output_linear.weight
    W.data += spec['noise'] * torch.randn(W.shape, generator=g)
    W.data[2] = spec['eos_gain'] * W.data[spec['eos_from']]
    model.eval()
    x, lens, y = synth_batch(spec['seed'], spec['k'], spec['T'], spec['L'], cfg['vocab_size'], variable=True)
    steps = []
    # greedy_search projects the whole (B, t, d) prefix; the teacher-forced pass before it projects sample by sample (batch 1)
    hook = model.decoder.output_linear.register_forward_hook(
        lambda m, i, o: steps.append(o[:, -1].max(dim=1)[1].clone()) if o.size(0) == spec['k'] else None)
    with torch.no_grad():
        _, strs_hyps, strs_gold = model.evaluate(x, lens, y, args, beam_search=False, start_token=vocab.SOS_ID)
    hook.remove()
    assert len(steps) == 300
    ids = torch.stack(steps).numpy().astype(np.int64)                       # (300, B)
    enc_str = lambda lst: np.frombuffer('\n'.join(lst).encode('utf-8'), dtype=np.uint8)
    store = dict(greedy_ids=ids, greedy_strs=enc_str(strs_hyps), gold_strs=enc_str(strs_gold),
                 spec=np.frombuffer(json.dumps(spec).encode(), dtype=np.uint8))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'G0.npz'), **store)
    print('G0: first EOS at steps', [int((ids[:, b] == 2).nonzero()[0][0]) if (ids[:, b] == 2).any() else -1 for b in range(ids.shape[1])])
    for st in strs_hyps:
        print('   ', repr(st))


def run_lm_fixture(torch):
    """what CAN be pinned of the LM meta-transfer path (its loop raises on torch >= 2, see oracle/lm_refimpl.py):
    the real `lm/model/rnn_model.py:RNNModel` (LSTM, 2 layers) -- initialisation hash, logits / loss / gradients / new hidden
    state of one batch at dropout 0 -- and the real `lm/util/data.py:LMDataset` batchify / sample on a seeded token stream."""
    for name in [m for m in list(sys.modules) if m == 'utils' or m.startswith('utils.') or m == 'models' or m.startswith('models.')]:
        del sys.modules[name]                                        # the ASR tree's `utils` would shadow lm/util? (different names; be safe)
    sys.path.insert(0, '/root/reference/lm')
    from model.rnn_model import RNNModel
    import util.data as ldata
    sys.path.insert(0, ROOT)
    from oracle.lm_refimpl import synth_corpus
    spec = dict(ntoken=300, ninp=48, nhid=48, nlayers=2, bptt=7, batch_size=4, seed=1111, corpus_seeds=[5, 6, 7], corpus_len=403, it=9)
    torch.manual_seed(spec['seed'])
    torch.set_num_threads(8)
    model = RNNModel('LSTM', spec['ntoken'], spec['ninp'], spec['nhid'], spec['nlayers'], 0.0, False)
    model.train()
    names = [n for n, _ in model.named_parameters()]
    h = hashlib.sha256()
    for _, p in model.named_parameters():
        h.update(p.detach().numpy().tobytes())
    args = argparse.Namespace(bptt=spec['bptt'], batch_size=spec['batch_size'], cuda=False)
    streams = [synth_corpus(s_, spec['ntoken'], spec['corpus_len']) for s_ in spec['corpus_seeds']]
    ds = ldata.LMDataset(streams, args)
    store = {'spec': np.frombuffer(json.dumps(spec).encode(), dtype=np.uint8), 'param_names': np.array(names),
             'theta0_sha256': np.frombuffer(h.hexdigest().encode(), dtype=np.uint8)}
    for m in range(3):
        store['batchified/%d' % m] = ds.task_list[m].numpy()
        for it in (0, spec['it'], 57):
            tr_x, tr_y, va_x, va_y = ds.sample(m if m < 2 else -1, it)
            for k_, v in (('tr_x', tr_x), ('tr_y', tr_y), ('va_x', va_x), ('va_y', va_y)):
                store['sample/%d/%d/%s' % (m, it, k_)] = v.numpy()
    x, y, _, _ = ds.sample(0, spec['it'])
    hidden = model.init_hidden(spec['batch_size'])
    g = torch.Generator().manual_seed(3)
    hidden = tuple(0.1 * torch.randn(t.shape, generator=g) for t in hidden)          # a non-trivial carried state
    out, hn = model(x, hidden)
    loss = torch.nn.CrossEntropyLoss()(out.view(-1, spec['ntoken']), y)
    loss.backward()
    store['h0'], store['c0'] = hidden[0].numpy(), hidden[1].numpy()
    store['out'] = out.detach().numpy()
    store['hn'], store['cn'] = hn[0].detach().numpy(), hn[1].detach().numpy()
    store['loss'] = np.float64(float(loss))
    for n_, p in model.named_parameters():
        store['grad/' + n_] = p.grad.numpy()
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'L0.npz'), **store)
    print('L0 written: loss %.6f, %d parameters' % (float(loss), len(names)))


def frontend_waveform(seed=7, n=8037):
    """seeded test waveform of the S0 fixture (two tones, a chirp and noise; 0.5 s + 37 samples at 16 kHz)"""
    rng = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    return (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t * (1 + 0.1 * t)) + 0.05 * rng.randn(n)).astype(np.float32)


def run_frontend_fixture():
    """S0: output of oracle/frontend.py (NOT of the reference: librosa is absent here) on a seeded waveform.  It
    pins the restatement -- and through tests/test_ops_gpu.py the device front-end -- against drift; the row stays "parity unpinned"."""
    from oracle import frontend
    y = frontend_waveform()
    store = {'note': np.array('oracle/frontend.py restating librosa.stft(n_fft=320, hop_length=160, win_length=320, window=scipy.signal.hamming, '
                              'center=True, pad_mode=reflect) -> abs -> log1p -> (x - mean) / std  (utils/data_loader.py:65-96); generated WITHOUT '
                              'librosa: parity unpinned'),
             'seed': np.array([7, 8037]), 'waveform': y,
             'spect_raw': frontend.parse_audio(y, normalize=False).numpy(), 'spect_norm': frontend.parse_audio(y, normalize=True).numpy()}
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'S0.npz'), **store)
    print('S0 written: spectrogram', store['spect_norm'].shape)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--ns', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--t5000', action='store_true', help='only the T = 5000, B = 8 checksum record (tests/golden/T5.npz)')
    a = ap.parse_args()
    if a.t5000:
        a.only = 'T5'
    if a.only == 'S0':
        sys.path.insert(0, ROOT)
        run_frontend_fixture()
        raise SystemExit(0)
    torch = bootstrap_reference()
    todo = [a.only] if a.only else (['F0', 'F1', 'J0', 'B0', 'G0'] + (['NS'] if a.ns else []) + ['L0'])
    for name in todo:
        if name == 'B0':
            run_beam_fixture(torch)
        elif name == 'G0':
            run_greedy_fixture(torch)
        elif name == 'L0':
            run_lm_fixture(torch)
        elif name == 'J0':
            run_joint_fixture(torch)
        else:
            NSAMPLE['n'] = 4096 if name in ('NS', 'T5') else 256
            run_fixture(name, FIXTURES[name], torch)
