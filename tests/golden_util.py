"""Helpers shared by the oracle and GPU parity tests for reading tests/golden/*.npz."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    z = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    cfg = {str(k): int(v) for k, v in zip(z['cfg_keys'], z['cfg_vals'])}
    k, T, L, n_tasks, iters, variable = [int(v) for v in z['spec']]
    spec = dict(k=k, T=T, L=L, n_tasks=n_tasks, iters=iters, variable=bool(variable),
                lr=float(z['lr']), meta_lr=float(z['meta_lr']))
    return z, cfg, spec


def check_digest(z, prefix, name, t, rtol, what='', floor=1e-30):
    """Compare tensor `t` with the stored digest (sum, l2, and full / strided sample).  rtol=0 -> bit-exact."""
    a = t.detach().cpu().numpy().astype(np.float32).reshape(-1)
    base = '%s/%s/' % (prefix, name)
    l2 = float(z[base + 'l2'])
    if base + 'full' in z.files:
        ref, got = z[base + 'full'], a
    else:
        step = int(z[base + 'step'])
        ref = z[base + 'sample']
        got = a[::step][:ref.size]
    if rtol == 0:
        assert np.array_equal(ref, got), '%s %s%s not bit-identical' % (what, prefix, name)
        return 0.0
    # `floor`: tensors whose exact value is 0 (e.g. key-projection bias grads: softmax is shift-invariant)
    # hold only rounding noise; they are compared against a fraction of the global scale instead.
    denom = max(float(np.sqrt((ref.astype(np.float64) ** 2).sum())), floor)
    err = float(np.sqrt(((ref.astype(np.float64) - got.astype(np.float64)) ** 2).sum())) / denom
    assert err <= rtol, '%s %s%s rel err %.3e > %.1e' % (what, prefix, name, err, rtol)
    got_l2 = float(np.sqrt((a.astype(np.float64) ** 2).sum()))
    assert abs(got_l2 - l2) <= rtol * max(l2, floor) * 4 + 1e-30, '%s %s%s l2 %.6e vs %.6e' % (what, prefix, name, got_l2, l2)
    return err


def batches_for(cfg, spec, it, data_call_index):
    """(task_batches, val_batch) consumed by iteration `it` of the golden run (see oracle/make_golden.py)."""
    from oracle.refimpl import synth_batch
    call = int(data_call_index[it])
    n = spec['n_tasks']
    tr = [synth_batch(1000 * call + 10 * m + 0, spec['k'], spec['T'], spec['L'], cfg['vocab_size'], spec['variable'])
          for m in range(n)]
    val = synth_batch(1000 * call + 10 * (n - 1) + 1, spec['k'], spec['T'], spec['L'], cfg['vocab_size'], spec['variable'])
    return tr, val


def global_l2(z, prefix, names):
    return float(np.sqrt(sum(float(z['%s/%s/l2' % (prefix, n)]) ** 2 for n in names)))


def load_beam():
    """B0.npz (reference Decoder.beam_search on the F0 model with a seeded perturbation of the vocabulary projection):
    -> (spec dict, list of id lists, list of strings, list of evaluate() strings)."""
    import json
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'B0.npz'))
    spec = json.loads(bytes(z['spec']).decode())
    ids = [[int(v) for v in row if v >= 0] for row in z['beam_ids']]
    dec = lambda k: bytes(z[k]).decode('utf-8').split('\n')
    return spec, ids, dec('beam_strs'), dec('eval_beam_strs')


def load_greedy():
    """G0.npz (reference Decoder.greedy_search, 300 steps, on the F0 model with tgt_max_len 320 and the B0-style perturbation):
    -> (spec dict, (300, B) int64 ids, list of strings, list of gold strings)."""
    import json
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'G0.npz'))
    spec = json.loads(bytes(z['spec']).decode())
    dec = lambda k: bytes(z[k]).decode('utf-8').split('\n')
    return spec, z['greedy_ids'], dec('greedy_strs'), dec('gold_strs')


def perturb_output_layer(weight, spec):
    """the fixture's deterministic change of decoder.output_linear.weight (in place), as oracle/make_golden.py applies it"""
    import torch
    g = torch.Generator().manual_seed(int(spec['noise_seed']))
    with torch.no_grad():
        weight += float(spec['noise']) * torch.randn(weight.shape, generator=g).to(weight.device)
        weight[2] = float(spec['eos_gain']) * weight[int(spec['eos_from'])]


def label_words(id2label, specials):
    """num_words(yseq) of modules/decoder.py:257-259 for a synthetic vocabulary"""
    def num_words(yseq):
        st = ''.join(id2label[int(c)] for c in yseq)
        for tok in specials:
            st = st.replace(tok, '')
        return len(st.replace('  ', ' ').split())
    return num_words
