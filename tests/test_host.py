"""CPU: host-side logic of the drop-in layer (parameter tree / init order, vocabulary and loaders, decoder I/O prep,
checkpoint format, loud failure without a GPU)."""
import argparse
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from tests import golden_util as gu


def make_args(cfg, **kw):
    base = dict(feat_extractor='vgg_cnn', sample_rate=16000, window_size=.02, feat='spectrogram', dim_input=161, dropout=0.0,
                emb_trg_sharing=False, label_smoothing=0.0, name='t', lr=1e-2, meta_lr=1e-3, k_train=2, k_valid=2, clip=False,
                max_norm=400, save_every=1, save_folder='/tmp/mtl_ckpt_test', cuda=False, is_factorized=False, r=cfg.get('r', 100))
    base.update({k: v for k, v in cfg.items() if k not in ('vocab_size', 'r')})
    base.update(kw)
    return argparse.Namespace(**base)


@pytest.mark.parametrize('name', ['F0', 'F1'])
def test_product_init_matches_reference_bit_for_bit(name):
    import mtl_amd
    z, cfg, spec = gu.load(name)
    torch.manual_seed(123456)
    m = mtl_amd.init_transformer_model(make_args(cfg), mtl_amd.synthetic_vocab(cfg['vocab_size']), r=cfg['r'])
    assert [n for n, _ in m.named_parameters()] == [str(s) for s in z['param_names']]
    h = hashlib.sha256()
    for _, p in m.named_parameters():
        h.update(p.detach().numpy().tobytes())
    assert h.hexdigest() == bytes(z['theta0_sha256']).decode()
    sd = m.state_dict()
    assert 'encoder.positional_encoding.pe' in sd and 'decoder.positional_encoding.pe' in sd and 'conv.7.bias' in sd
    # parameters are views into ONE flat buffer, gradients likewise
    p0 = next(m.parameters())
    assert p0.data_ptr() == m.flat_parameters.data_ptr() and p0.grad.data_ptr() == m.flat_grad.data_ptr()
    assert m.flat_parameters.numel() == sum(p.numel() for p in m.parameters())


def test_no_cpu_fallback():
    import mtl_amd
    z, cfg, spec = gu.load('F0')
    m = mtl_amd.init_transformer_model(make_args(cfg), mtl_amd.synthetic_vocab(64))
    x, lens, y = mtl_amd.synth_batch(0, 2, 64, 8, 64)
    with pytest.raises(RuntimeError, match='no CPU'):
        m(x, lens, y)


def test_decoder_io_matches_oracle():
    from oracle import refimpl as R
    import importlib
    eng = importlib.import_module('mtl_amd.engine')
    y = torch.tensor([[5, 6, 7, 0, 0], [9, 8, 7, 6, 5], [4, 0, 0, 0, 0]])
    a, b = eng.decoder_io(y)
    ra, rb = R.decoder_io(y)
    assert torch.equal(a, ra) and torch.equal(b, rb)
    assert a.tolist()[0] == [1, 5, 6, 7, 2, 2] and b.tolist()[0] == [5, 6, 7, 2, 0, 0]
    # ragged batches (empty rows, pads in the middle of a row -- the reference drops pads wherever they stand, modules/decoder.py:57),
    # with and without a forced width: the vectorised host prep against the oracle's per-row restatement
    g = torch.Generator().manual_seed(0)
    for trial in range(100):
        B, L = int(torch.randint(1, 9, (1,), generator=g)), int(torch.randint(1, 20, (1,), generator=g))
        t = torch.randint(3, 30, (B, L), generator=g)
        for i in range(B):
            n = int(torch.randint(0, L + 1, (1,), generator=g))
            t[i, n:] = 0
            if trial % 3 == 0 and n > 2:
                t[i, 1] = 0
        ra, rb = R.decoder_io(t)
        a, b = eng.decoder_io(t)
        assert torch.equal(a, ra) and torch.equal(b, rb), trial
        a, b = eng.decoder_io(t, width=L + 3)
        assert torch.equal(a[:, :ra.shape[1]], ra) and torch.equal(b[:, :rb.shape[1]], rb) and a.shape[1] == max(L + 3, ra.shape[1])
        assert bool((a[:, ra.shape[1]:] == 2).all()) and bool((b[:, rb.shape[1]:] == 0).all())


def test_vocab_manifest_and_sampling(tmp_path):
    import mtl_amd
    labels = ['_', "'", 'a', 'b', 'c', ' ', '你']
    lp = tmp_path / 'labels.json'
    lp.write_text(json.dumps(labels), encoding='utf-8')
    vocab = mtl_amd.load_vocab(str(lp))
    assert vocab.label2id['<PAD>'] == 0 and vocab.label2id['<SOS>'] == 1 and vocab.label2id['<EOS>'] == 2 and vocab.label2id['<OOV>'] == 3
    assert vocab.label2id['_'] == 4 and vocab.id2label[10] == '你' and len(vocab.label2id) == 11
    rows = []
    for i in range(5):
        t = tmp_path / ('u%d.txt' % i)
        t.write_text('ab c你\n' if i % 2 else 'CAB', encoding='utf8')
        rows.append('%s,%s' % (tmp_path / ('u%d.wav' % i), t))
    mp = tmp_path / 'm.csv'
    mp.write_text('\n'.join(rows) + '\n')
    args = argparse.Namespace(src_max_len=50)
    feats = lambda wav: torch.randn(161, 30 + 7 * int(os.path.basename(wav)[1]))
    ds = mtl_amd.ManifestTaskDataset(vocab, args, [str(mp)], feats, partitions=None)
    np.random.seed(0)
    tr, va = ds.sample(3, 2, 0)
    x, sizes, pct, tgt, tsz = tr
    assert x.shape[0] == 3 and x.shape[1] == 1 and x.shape[2] == 161 and x.shape[3] == int(sizes.max()) <= 50
    assert tgt.dtype == torch.int64 and sizes.dtype == torch.int32 and abs(float(pct.max()) - 1.0) < 1e-6
    assert va[0].shape[0] == 2
    # leading ' ' + lower-casing as the reference's parse_transcript
    assert mtl_amd.data.parse_transcript(vocab, 'AB') == [vocab.label2id['a'], vocab.label2id['b']]
    assert (x[0, 0, :, int(sizes[0]):] == 0).all()
    # need=: a part the caller will not use is drawn (same index stream) but not loaded
    a = mtl_amd.ManifestTaskDataset(vocab, args, [str(mp)], feats, partitions=None, seed=5)
    b = mtl_amd.ManifestTaskDataset(vocab, args, [str(mp)], feats, partitions=None, seed=5)
    for _ in range(3):
        full, part = a.sample(3, 2, 0), b.sample(3, 2, 0, need=(False, True))
        assert part[0] is None and torch.equal(full[1][3], part[1][3]) and torch.equal(full[1][1], part[1][1])


def test_seeded_dataset_item_access_and_audio_loader(tmp_path):
    """ManifestTaskDataset(seed=...) draws from its own RandomState (ranks in lock-step, unaffected by other users of np.random),
    __getitem__ / __len__ follow utils/data_loader.py:323-340, and AudioDataLoader yields the reference's validation batch layout
    (utils/data_loader.py:401-440): sorted by descending length, zero / PAD padded, (inputs, targets, percentages, sizes, sizes)."""
    import mtl_amd
    vocab = mtl_amd.synthetic_vocab(64)
    rows = []
    for i in range(6):
        t = tmp_path / ('u%d.txt' % i)
        t.write_text(''.join(chr(0x4e00 + j) for j in range(1 + i)), encoding='utf8')
        rows.append('%s,%s' % (tmp_path / ('u%d.wav' % i), t))
    mp_ = tmp_path / 'm.csv'
    mp_.write_text('\n'.join(rows) + '\n')
    args = argparse.Namespace(src_max_len=40)
    feats = lambda wav: torch.full((161, 20 + 5 * int(os.path.basename(wav)[1])), float(os.path.basename(wav)[1]))
    a = mtl_amd.ManifestTaskDataset(vocab, args, [str(mp_)], feats, seed=11)
    b = mtl_amd.ManifestTaskDataset(vocab, args, [str(mp_)], feats, seed=11)
    np.random.seed(1)
    ta, _ = a.sample(3, 2, 0)
    np.random.rand(7)                                   # another consumer of the global RNG between the two ranks' draws
    tb, _ = b.sample(3, 2, 0)
    assert all(torch.equal(u, v) for u, v in zip(ta, tb))
    assert len(a) == 6
    spect, trans = a[7]                                 # evaluation indexing: manifest 0, index modulo its length
    assert spect.shape == (161, 25) and trans == [4, 5]
    assert a[5][0].shape == (161, 40)                   # truncated to src_max_len
    loader = mtl_amd.AudioDataLoader(vocab.PAD_ID, dataset=a, batch_size=4)
    batches = list(loader)
    assert len(batches) == 2
    inputs, targets, pct, sizes, tsizes = batches[0]
    assert inputs.shape == (4, 1, 161, 35) and sizes.tolist() == [35, 30, 25, 20] and sizes.dtype == torch.int32
    assert targets.shape == (4, 4) and targets.dtype == torch.int64 and tsizes.tolist() == [4, 3, 2, 1]
    assert targets[3].tolist() == [4, 0, 0, 0] and torch.allclose(pct, sizes.float() / 35)
    assert float(inputs[3, 0, :, 20:].abs().max()) == 0.0 and float(inputs[3, 0, 0, 0]) == 0.0 and float(inputs[0, 0, 0, 0]) == 3.0


def test_joint_checkpoint_roundtrip_and_pickling_is_side_effect_free(tmp_path):
    """save_joint_model / load_joint_model (utils/functions.py:43-71,190-218) and the pickling of Vocab under the reference's
    class path WITHOUT touching process-global state (sys.modules, Vocab.__module__) while another thread may be importing."""
    import sys
    import mtl_amd
    z, cfg, spec = gu.load('F0')
    args = make_args(cfg, save_folder=str(tmp_path), name='jk', loss='ce')
    vocab = mtl_amd.synthetic_vocab(64)
    m = mtl_amd.init_transformer_model(args, vocab)
    opt = torch.optim.Adam(m.parameters(), lr=args.lr)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    before = (sys.modules.get('utils'), sys.modules.get('utils.data'), mtl_amd.Vocab.__module__)
    path = mtl_amd.save_joint_model(m, vocab, 3, opt, {'avg_valid_loss': 2.0}, args, best_model=True)
    assert (sys.modules.get('utils'), sys.modules.get('utils.data'), mtl_amd.Vocab.__module__) == before
    assert path.endswith('jk/best_model.th')
    ck = mtl_amd.functions.load_checkpoint_dict(path)
    assert sorted(ck.keys()) == ['args', 'epoch', 'metrics', 'model_state_dict', 'opt', 'vocab']
    assert isinstance(ck['opt'], torch.optim.Adam) and type(ck['vocab']) is mtl_amd.Vocab
    import zipfile
    with zipfile.ZipFile(path) as zf:
        pkl = zf.read([n for n in zf.namelist() if n.endswith('data.pkl')][0])
    assert b'utils.data' in pkl and b'mtl_amd' not in pkl
    m2, v2, o2, ep, met, a2 = mtl_amd.load_joint_model(path)
    assert ep == 3 and met['avg_valid_loss'] == 2.0 and v2.id2label == vocab.id2label
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    s1, s2 = opt.state_dict()['state'], o2.state_dict()['state']
    assert len(s1) == len(s2) and all(torch.equal(s1[k]['exp_avg'], s2[k]['exp_avg']) for k in s1)


def test_checkpoint_roundtrip_reference_format(tmp_path):
    import mtl_amd
    z, cfg, spec = gu.load('F0')
    args = make_args(cfg, save_folder=str(tmp_path), name='ck')
    vocab = mtl_amd.synthetic_vocab(64)
    m = mtl_amd.init_transformer_model(args, vocab)
    inner = torch.optim.SGD(m.parameters(), lr=args.lr)
    outer = torch.optim.Adam(m.parameters(), lr=args.meta_lr)
    path = mtl_amd.save_meta_model(m, vocab, 7, inner, outer, {'avg_valid_cer': 1.0}, args, best_model=False)
    ck = mtl_amd.functions.load_checkpoint_dict(path)
    assert sorted(ck.keys()) == ['args', 'epoch', 'inner_opt', 'metrics', 'model_state_dict', 'outer_opt', 'vocab']
    # what the reference's loader needs (utils/functions.py:158-188): optimizer OBJECTS with .state_dict(), its own Vocab class path
    assert isinstance(ck['inner_opt'], torch.optim.SGD) and isinstance(ck['outer_opt'], torch.optim.Adam)
    import zipfile
    with zipfile.ZipFile(path) as zf:
        pkl = zf.read([n for n in zf.namelist() if n.endswith('data.pkl')][0])
    assert b'utils.data' in pkl and b'Vocab' in pkl and b'mtl_amd' not in pkl and b'meta-transfer' not in pkl
    assert mtl_amd.Vocab.__module__ != 'utils.data'                     # the alias exists only while dumping
    m2, v2, i2, o2, ep, met, a2 = mtl_amd.load_meta_model(path)
    assert ep == 7 and met['avg_valid_cer'] == 1.0 and len(v2.label2id) == 64
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    assert o2.param_groups[0]['lr'] == args.meta_lr


def test_task_sharding():
    import mtl_amd
    assert mtl_amd.dist.shard_tasks(8, 3, 8) == [3]
    assert mtl_amd.dist.shard_tasks(8, 1, 2) == [1, 3, 5, 7]
    assert sorted(sum((mtl_amd.dist.shard_tasks(3, r, 2) for r in range(2)), [])) == [0, 1, 2]


def test_frontend_oracle_is_the_textbook_stft():
    """oracle/frontend.py (numpy restatement of librosa.stft for the reference's call; parity unpinned, no librosa here) against
    the DFT definition evaluated directly in float64."""
    from oracle import frontend
    from scipy.signal import windows
    rng = np.random.RandomState(1)
    y = rng.randn(1000).astype(np.float32)
    n_fft, hop = 320, 160
    mag = frontend.stft_magnitude(y, n_fft, hop)
    assert mag.shape == (161, 1 + 1000 // 160)
    yp = np.pad(y.astype(np.float64), 160, mode='reflect')
    w = windows.hamming(n_fft)
    assert abs(w[0] - w[-1]) < 1e-12 and abs(w[0] - 0.08) < 1e-12            # symmetric window (callable passed to librosa)
    for t in (0, 3, 6):
        frame = yp[t * hop:t * hop + n_fft] * w
        for f in (0, 1, 57, 160):
            ref = abs(np.sum(frame * np.exp(-2j * np.pi * f * np.arange(n_fft) / n_fft)))
            assert abs(mag[f, t] - ref) < 1e-3 * max(ref, 1.0)
    sp = frontend.parse_audio(y)
    assert abs(float(sp.mean())) < 1e-5 and abs(float(sp.std()) - 1.0) < 1e-5


def test_frontend_oracle_reproduces_its_committed_fixture():
    """tests/golden/S0.npz (oracle/make_golden.py --only S0): the front-end restatement on a seeded waveform.  Generated without
    librosa (absent here), so SURVEY 8(f) f1 stays parity-unpinned; the fixture stops the restatement -- and, in the GPU test, the
    device front-end -- from drifting."""
    from oracle import frontend
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'S0.npz'))
    assert 'parity unpinned' in str(z['note']) and 'librosa.stft' in str(z['note'])
    y = z['waveform']
    assert y.dtype == np.float32 and y.shape == (int(z['seed'][1]),)
    for key, norm in (('spect_raw', False), ('spect_norm', True)):
        got = frontend.parse_audio(y, normalize=norm).numpy()
        assert got.shape == z[key].shape == (161, 1 + y.size // 160)
        assert np.abs(got - z[key]).max() <= 2e-6 * np.abs(z[key]).max()


REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(REF), reason='needs the reference checkout (build container only)')
def test_checkpoints_cross_load_with_the_real_reference(tmp_path):
    """SURVEY 8(f) f4, both directions, in a subprocess that imports the REAL reference: (1) a `.th` written by the reference's
    save_meta_model (after one Adam step, so optimizer state exists) loads here with identical weights, vocabulary and Adam
    moments; (2) a `.th` written here is consumed by the reference's own load path (its classes, its load_state_dict)."""
    import subprocess
    import sys
    import mtl_amd
    z, cfg, spec = gu.load('F0')
    args = make_args(cfg, save_folder=str(tmp_path), name='ours')
    args.is_factorized, args.r, args.cuda = False, 100, False
    vocab = mtl_amd.synthetic_vocab(64)
    m = mtl_amd.init_transformer_model(args, vocab, is_factorized=False, r=100)
    outer = torch.optim.Adam(m.parameters(), lr=args.meta_lr)
    for p in m.parameters():
        p.grad = torch.full_like(p, 0.01)
    outer.step()
    ours = mtl_amd.save_meta_model(m, vocab, 3, torch.optim.SGD(m.parameters(), lr=args.lr), outer, {'avg_valid_cer': 2.0}, args)
    script = r"""
import sys, torch
sys.path.insert(0, %r)
from oracle.make_golden import bootstrap_reference
bootstrap_reference()
import argparse
from utils.data import Vocab
from utils import functions as RF
# (2) our file through the reference's load path (its torch.load call predates weights_only, so the dict is read explicitly)
ck = torch.load(%r, map_location='cpu', weights_only=False)
assert type(ck['vocab']).__module__ == 'utils.data' and type(ck['vocab']) is Vocab
a = ck['args']
model = RF.init_transformer_model(a, ck['vocab'], train=True, is_factorized=a.is_factorized, r=a.r)
model.load_state_dict(ck['model_state_dict'])
io = torch.optim.SGD(model.parameters(), lr=a.lr); oo = torch.optim.Adam(model.parameters(), lr=a.meta_lr)
io.load_state_dict(ck['inner_opt'].state_dict()); oo.load_state_dict(ck['outer_opt'].state_dict())
assert len(oo.state) == len(list(model.parameters())) and ck['epoch'] == 3
# (1) a checkpoint written by the reference itself
for p in model.parameters():
    p.grad = torch.full_like(p, -0.02)
oo.step()
a.save_folder, a.name = %r, 'theirs'
RF.save_meta_model(model, ck['vocab'], 4, io, oo, {'avg_valid_cer': 1.5}, a)
print('REFERENCE_OK')
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ours, str(tmp_path))
    env = dict(os.environ, PYTHONPATH=REF)
    out = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert 'REFERENCE_OK' in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    theirs = os.path.join(str(tmp_path), 'theirs', 'epoch_4.th')
    m2, v2, i2, o2, ep, met, a2 = mtl_amd.load_meta_model(theirs)
    assert ep == 4 and met['avg_valid_cer'] == 1.5 and type(v2) is mtl_amd.Vocab and v2.id2label == vocab.id2label
    raw = mtl_amd.functions.load_checkpoint_dict(theirs)
    for (n, p) in m2.named_parameters():
        assert torch.equal(p, raw['model_state_dict'][n])
    st = o2.state_dict()['state']
    assert len(st) == len(list(m2.parameters())) and float(st[0]['step']) == 2.0
    assert torch.equal(st[0]['exp_avg'], raw['outer_opt'].state_dict()['state'][0]['exp_avg'])


def test_spectrogram_dataset_constructed_as_the_reference_entry_script_does(tmp_path, capsys):
    """meta_transfer_train.py:141-175 builds `SpectrogramDataset(vocab, args, audio_conf, manifest_filepath_list=..., normalize=True,
    augment=args.augment, input_type=args.input_type, is_train=True, partitions=args.train_partition_list)` once per training manifest and
    `BucketingSampler(valid_data, batch_size=args.k_train)` per validation manifest (utils/data_loader.py:171-236,480-500): the same
    calls must work on this package by import swap, with the reference object's attributes, console lines, index stream
    (np.random.choice over the partition's leading fraction, global RNG) and .sample() layout."""
    import mtl_amd
    vocab = mtl_amd.synthetic_vocab(64)
    manifests = []
    for m, n in enumerate((7, 4)):
        rows = []
        for i in range(n):
            t = tmp_path / ('m%d_u%d.txt' % (m, i))
            t.write_text(''.join(chr(0x4e00 + (3 * m + i + j) % 60) for j in range(2 + i)), encoding='utf8')
            rows.append('%s,%s' % (tmp_path / ('m%d_u%d.wav' % (m, i)), t))
        mp_ = tmp_path / ('train%d.csv' % m)
        mp_.write_text('\n'.join(rows) + '\n')
        manifests.append(str(mp_))
    args = argparse.Namespace(src_max_len=40, sample_rate=16000, window_size=.02, window_stride=.01, window='hamming', augment=False,
                              input_type='char', train_manifest_list=manifests, train_partition_list=[0.5, 1.0], k_train=3,
                              noise_dir=None, noise_prob=0.4, noise_min=0.0, noise_max=0.5)
    audio_conf = dict(sample_rate=args.sample_rate, window_size=args.window_size, window_stride=args.window_stride, window=args.window,
                      noise_dir=args.noise_dir, noise_prob=args.noise_prob, noise_levels=(args.noise_min, args.noise_max))
    feats = lambda wav: torch.full((161, 12 + 3 * int(os.path.basename(wav)[4])), float(os.path.basename(wav)[1]))
    train_data_list = []
    for i in range(len(args.train_manifest_list)):      # (the reference builds one dataset object per manifest, each over ALL manifests)
        train_data_list.append(mtl_amd.SpectrogramDataset(vocab, args, audio_conf, manifest_filepath_list=args.train_manifest_list,
                                                          normalize=True, augment=args.augment, input_type=args.input_type, is_train=True,
                                                          partitions=args.train_partition_list, feature_fn=feats))
    out = capsys.readouterr().out
    assert out.count('max_size: 30000') == 2 and out.count('input_type: char') == 2          # :198-207 (several manifests, is_train)
    ds = train_data_list[1]
    assert len(ds) == 30000 and ds.is_train and ds.input_type == 'char' and ds.manifest_filepath_list == manifests
    assert [len(ids) for ids in ds.ids_list] == [7, 4] and ds.part_len == 4                   # (:211-216: the LAST manifest's partition size)
    assert np.allclose(ds.proba[0], [1 / 3] * 3 + [0] * 4) and np.allclose(ds.proba[1], [0.25] * 4)
    # the index stream of .sample(): np.random.choice(len, k_tr + k_val, p=proba, replace=True) on the global RNG (:247-249)
    np.random.seed(123456)
    want = np.random.choice(np.arange(0, 7), 3 + 2, p=ds.proba[0], replace=True)
    np.random.seed(123456)
    (x, sizes, pct, tgt, tsz), va = ds.sample(3, 2, 0)
    assert [int(v) for v in x[:, 0, 0, 0]] == [0, 0, 0] and want.max() <= 2                  # manifest 0, leading half only
    assert [int(s) for s in sizes] == [12 + 3 * int(j) for j in want[:3]] and [int(s) for s in va[1]] == [12 + 3 * int(j) for j in want[3:]]
    assert [int(n) for n in tsz] == [2 + int(j) for j in want[:3]] and x.shape[1:3] == (1, 161) and tgt.dtype == torch.int64
    assert ds.parse_transcript('一丁') == [vocab.label2id['一'], vocab.label2id['丁']]
    # validation side: one manifest, not training -> max_size = its length; BucketingSampler bins / shuffles like the reference
    valid = mtl_amd.SpectrogramDataset(vocab, args, audio_conf, manifest_filepath_list=[manifests[0]], normalize=True,
                                       augment=args.augment, input_type=args.input_type, feature_fn=feats)
    assert len(valid) == 7 and not valid.is_train and valid.part_len == 7
    spect, transcript = valid[9]                                                             # index % len (:335-340)
    assert spect.shape == (161, 12 + 3 * 2) and len(transcript) == 4
    sampler = mtl_amd.BucketingSampler(valid, batch_size=args.k_train)
    assert len(sampler) == 3 and [len(b) for b in sampler.bins] == [3, 3, 1]
    np.random.seed(7)
    got = [list(b) for b in sampler]
    np.random.seed(7)
    ref_bins = [[0, 1, 2], [3, 4, 5], [6]]
    for b in ref_bins:
        np.random.shuffle(b)
    assert got == ref_bins and sorted(sum(got, [])) == list(range(7))
    np.random.seed(8)
    sampler.shuffle(0)
    np.random.seed(8)
    np.random.shuffle(ref_bins)
    assert sampler.bins == ref_bins
    loader = mtl_amd.AudioDataLoader(pad_token_id=vocab.PAD_ID, dataset=valid, batch_sampler=sampler)
    inputs, targets, percentages, input_sizes, target_sizes = next(iter(loader))
    assert inputs.shape[1:3] == (1, 161) and inputs.shape[0] == len(ref_bins[0]) and int(input_sizes[0]) == int(input_sizes.max())
    # outside the accelerated path: rejected loudly, not ignored
    import pytest
    with pytest.raises(NotImplementedError):
        mtl_amd.SpectrogramDataset(vocab, args, audio_conf, manifest_filepath_list=manifests, augment=True, feature_fn=feats)
    with pytest.raises(NotImplementedError):
        mtl_amd.SpectrogramDataset(vocab, args, dict(audio_conf, noise_dir='/noise'), manifest_filepath_list=manifests, feature_fn=feats)


def test_schedule_choice_for_batches_of_different_widths():
    """Host logic of the manifest-fed case (data.py:77 pads every batch to its own longest utterance): rounded widths repeat and never
    shrink a batch; a fixed-shape workload is never widened; tasks of different frame counts are stacked while they fill the stack."""
    import types
    import mtl_amd
    from mtl_amd.engine import round_width
    for T in (4, 41, 64, 65, 600, 995, 1000, 1025, 2000, 4999, 5000):
        for q in (16, 64):
            w = round_width(T, q)
            assert w >= T and w % q == 0 and w - T < max(q, T // 16 + q)
    assert len({round_width(T, 64) for T in range(600, 1001)}) <= 8 and len({round_width(T, 64) for T in range(3000, 5001)}) <= 16
    tr = mtl_amd.TransientTrainer()
    assert tr.pad_lanes == 'auto' and tr.batch_ragged and tr.ragged_quantum == 64
    fixed = [('lane', 0, 1000), ('lane', 1, 1000), ('lane', 'val', 800)]
    assert not tr._widths_vary(fixed) and not tr._widths_vary(fixed)                 # training and validation widths may differ: still fixed
    assert tr._widths_vary([('lane', 0, 1000), ('lane', 1, 990), ('lane', 'val', 800)])      # a slot brought a second width
    assert tr._widths_vary(fixed)                                                    # ... and it stays on
    tr.pad_lanes = '0'
    assert not tr._widths_vary([('lane', 0, 7)])
    eng = types.SimpleNamespace(fused_attn=True)
    model = types.SimpleNamespace(engines=[eng])
    batch = lambda k, T: (torch.zeros(k, 1, 161, T),)
    tr = mtl_amd.TransientTrainer()
    val = batch(2, 50)
    assert tr._can_batch(model, [batch(2, 64), batch(2, 64)], val)
    assert tr._can_batch(model, [batch(2, 64), batch(2, 40)], val)            # fill 0.81
    assert tr._can_batch(model, [batch(2, 64), batch(2, 8), batch(2, 8)], val)           # fill 80 / 192 = 0.42
    assert not tr._can_batch(model, [batch(2, 64), batch(2, 4), batch(2, 4), batch(2, 4)], val)      # fill 76 / 256 = 0.30: lanes
    assert not tr._can_batch(model, [batch(2, 64), batch(3, 64)], val)        # different sample counts
    assert not tr._can_batch(model, [batch(2, 64), batch(2, 3)], val)         # a batch too short for two poolings
    assert not tr._can_batch(model, [batch(2, 64)], val)                      # one task: a lane
    tr.batch_ragged = False
    assert tr._can_batch(model, [batch(2, 64), batch(2, 64)], val) and not tr._can_batch(model, [batch(2, 64), batch(2, 40)], val)


def test_environment_switches_are_the_documented_ones():
    """Every MTL_* variable the package reads is in INTEGRATION.md's table and is flipped by a test; nothing else is read (round 5 had
    58 of them, most left over from experiments that lost -- every one an untested configuration of a loop people run for days)."""
    import glob
    import re
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(ROOT, 'meta-transfer-learning_amd')
    read = set()
    for path in glob.glob(os.path.join(pkg, '*.py')):
        src = open(path).read()
        read |= set(re.findall(r"environ\.get\(\s*'(MTL_[A-Z0-9_]+)'", src)) | set(re.findall(r"environ\[\s*'(MTL_[A-Z0-9_]+)'", src))
    for path in glob.glob(os.path.join(pkg, 'csrc', '*')):
        if path.endswith(('.hip', '.h', '.inc')):
            assert 'getenv' not in open(path).read(), path          # the library reads no environment at all
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    table = set(re.findall(r"^\| `(MTL_[A-Z0-9_]+)` \|", doc, flags=re.M))
    assert read == table, (sorted(read - table), sorted(table - read))
    assert len(read) <= 15
    # where each one is flipped (file: a test that sets it, by the environment or through the attribute it initialises)
    flipped_in = {
        'MTL_CONV': 'tests/test_parity_gpu.py', 'MTL_BATCH_TASKS': 'tests/test_batched_gpu.py', 'MTL_TASK_LANES': 'tests/test_host.py',
        'MTL_PIPELINE_DEPTH': 'tests/test_trainer_gpu.py', 'MTL_POOL_GB': 'tests/test_parity_gpu.py', 'MTL_RAGGED_FILL': 'tests/test_host.py',
        'MTL_RAGGED_QUANTUM': 'tests/test_batched_gpu.py', 'MTL_CHUNKED_ALLREDUCE': 'tests/test_dist_gloo.py', 'MTL_DIST_BACKEND': 'tests/test_parity_gpu.py',
        'MTL_DIST_ALWAYS': 'tests/test_trainer_gpu.py', 'MTL_HOST_THREADS': 'tests/test_hostenv.py', 'MTL_LIB': 'tests/test_abi.py',
        'MTL_TRACE_PHASES': 'tests/test_host.py'}
    attr = {'MTL_CONV': 'conv_mode', 'MTL_BATCH_TASKS': 'batch_tasks', 'MTL_TASK_LANES': 'n_lanes', 'MTL_PIPELINE_DEPTH': 'pipeline',
            'MTL_RAGGED_FILL': 'ragged_fill', 'MTL_RAGGED_QUANTUM': 'ragged_quantum'}
    assert set(flipped_in) == read
    for var, path in flipped_in.items():
        text = open(os.path.join(ROOT, path)).read()
        assert var in text or attr.get(var, '\0') in text, (var, path)


def test_switches_read_at_construction(monkeypatch):
    """the trainer-side switches: environment -> attribute, and MTL_TRACE_PHASES / MTL_RAGGED_FILL change behaviour"""
    import importlib
    import types
    import mtl_amd
    monkeypatch.setenv('MTL_BATCH_TASKS', '0')
    monkeypatch.setenv('MTL_PIPELINE_DEPTH', '0')
    monkeypatch.setenv('MTL_RAGGED_FILL', '0.9')
    monkeypatch.setenv('MTL_RAGGED_QUANTUM', '16')
    tr = mtl_amd.TransientTrainer()
    assert tr.batch_tasks is False and tr.pipeline is False and tr.pipeline_depth == 0 and tr.ragged_fill == 0.9 and tr.ragged_quantum == 16
    assert tr._turns() == 2
    model = types.SimpleNamespace(engines=[types.SimpleNamespace(fused_attn=True)])
    batch = lambda k, T: (torch.zeros(k, 1, 161, T),)
    tr.batch_tasks = True
    assert not tr._can_batch(model, [batch(2, 64), batch(2, 40)], batch(2, 50))       # fill 0.81 < 0.9: lanes
    monkeypatch.delenv('MTL_PIPELINE_DEPTH')
    assert mtl_amd.TransientTrainer().pipeline_depth == 2
    # MTL_TASK_LANES: engines (stream + buffers) a model keeps for the lane schedule
    z, cfg, spec = gu.load('F0')
    from tests.test_parity_gpu import make
    monkeypatch.setenv('MTL_TASK_LANES', '3')
    assert make(cfg, spec)[3].n_lanes == 3
    monkeypatch.delenv('MTL_TASK_LANES')
    assert make(cfg, spec)[3].n_lanes == 8
    from mtl_amd import _trace
    monkeypatch.setenv('MTL_TRACE_PHASES', '1')
    try:
        importlib.reload(_trace)
        assert _trace.ON
        _trace.begin()
        _trace.mark('x')
        _trace.end()
        assert _trace.steps and 'x' in _trace.steps[-1]
    finally:
        monkeypatch.delenv('MTL_TRACE_PHASES')
        importlib.reload(_trace)
    assert not _trace.ON
