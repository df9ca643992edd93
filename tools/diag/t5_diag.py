"""Diagnostics (GPU box): the T = 5000, B = 8 record (tests/golden/T5.npz) against the device path, per tensor, without gates;
then the training pass against the LIVE oracle with the device's branch decisions replayed.  usage: MTL_CONV=h2|x3|f32 python tools/diag/t5_diag.py [--live]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests import golden_util as gu  # noqa: E402
from tests import test_parity_gpu as TP  # noqa: E402

z, cfg, spec = gu.load('T5')
mtl_amd, args, vocab, model = TP.make(cfg, spec)
names = [str(s) for s in z['param_names']]
model = model.cuda()
print('conv mode', model.engine.conv_mode)
if '--live' in sys.argv:
    from oracle import refimpl as R
    torch.set_num_threads(min(16, torch.get_num_threads()))
    oracle = R.build_model(cfg)
    tr, val = gu.batches_for(cfg, spec, 0, z['data_call_index'])
    try:
        TP._pass_parity(model, oracle, tr[0], model.flat_parameters, 'T5 train pass', max_flips=10 ** 6)
    except AssertionError as e:
        print('live oracle with replay: FAILED', str(e)[:400])
    sys.exit(0)
tasks = [mtl_amd.SyntheticTask(m, spec['k'], spec['T'], spec['L'], cfg['vocab_size'], variable=spec['variable']) for m in range(spec['n_tasks'])]
trainer = mtl_amd.TransientTrainer()
cap = {}
orig = trainer.meta_iteration


def spy(*a, **kw):
    cap['reads'] = orig(*a, **kw)
    return cap['reads']


trainer.meta_iteration = spy
trainer.train(model, vocab, tasks, [], 'ce', 0, 1, args, evaluate_every=10 ** 9, early_stop='cer,200', is_copy_grad=True)
for m, (tr, va) in enumerate(cap['reads']):
    for j, rd in ((2 * m, tr), (2 * m + 1, va)):
        key = 'fwd/0/%d' % j
        print(key, 'gold', np.array_equal(rd.gold_host.numpy(), z[key + '/gold']), 'hyp', np.array_equal(rd.hyp.numpy(), z[key + '/hyp']),
              'loss', float(rd.loss[0]), float(z[key + '/loss']))
floor = 1e-4 * gu.global_l2(z, 'G/0', names)
errs = {}
for nm in names:
    try:
        errs[nm] = gu.check_digest(z, 'G/0', nm, model._layout.view(model._G, nm), rtol=1e9, what='T5', floor=floor)
    except AssertionError as e:
        errs[nm] = float('nan')
        print('l2 mismatch', nm, e)
srt = sorted(errs.items(), key=lambda kv: -(kv[1] if kv[1] == kv[1] else 1e9))
print('%d / %d within 1e-4' % (sum(e <= 1e-4 for e in errs.values()), len(errs)))
for nm, e in srt[:25]:
    print('%.3e  %s' % (e, nm))
