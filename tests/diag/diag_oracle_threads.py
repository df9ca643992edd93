"""How reproducible is the CPU oracle itself across thread counts (north-star size, single pass)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import golden_util as gu
from oracle import refimpl as R
z, cfg, spec = gu.load('NS')
x, lens, y = R.synth_batch(0, spec['k'], spec['T'], spec['L'], cfg['vocab_size'], False)
res = {}
for nt in (8, 32, torch.get_num_threads()):
    torch.set_num_threads(nt)
    m = R.build_model(cfg)
    t = time.time()
    pr, gr, hr = m(x, lens, y)
    grads = torch.autograd.grad(R.ce_loss(pr, gr), list(m.parameters()))
    print('threads', nt, 'pass seconds %.2f' % (time.time() - t), flush=True)
    res[nt] = grads
names = [n for n, _ in m.named_parameters()]
keys = list(res)
for a in keys[1:]:
    rows = sorted(((float((u - v).norm() / v.norm().clamp_min(1e-30)), n) for n, u, v in zip(names, res[a], res[keys[0]])
                   if 'key_linear_b.bias' not in n), reverse=True)[:6]
    print(a, 'vs', keys[0], ' | '.join('%.2e %s' % r for r in rows))
print(torch.__config__.parallel_info()[:300])
