"""GPU diagnostic: per-tensor relative error of one pass's gradients (HIP vs CPU oracle)."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import golden_util as gu
from tests.test_parity_gpu import make
from oracle import refimpl as R

name = sys.argv[1] if len(sys.argv) > 1 else 'F0'
variable = int(sys.argv[2]) if len(sys.argv) > 2 else 1
z, cfg, spec = gu.load(name)
mtl_amd, args, vocab, model = make(cfg, spec)
model = model.cuda()
oracle = R.build_model(cfg)
x, lens, y = R.synth_batch(0, spec['k'], spec['T'], spec['L'], cfg['vocab_size'], bool(variable))
print('lens', lens.tolist())
out = model.pass_forward(x.cuda(), lens, y)
g = torch.zeros_like(model.flat_grad)
model.pass_backward(g, 1.0)
pr, gr, hr = oracle(x, lens, y)
loss = R.ce_loss(pr, gr)
grads = torch.autograd.grad(loss, list(oracle.parameters()))
gn = float(torch.sqrt(sum((t.double() ** 2).sum() for t in grads)))
print('pred rel', float((out['pred'].cpu() - pr).norm() / pr.norm()), 'loss', float(out['loss']), float(loss))
rows = []
for (nm, _), t in zip(oracle.named_parameters(), grads):
    h = model._layout.view(g, nm).cpu()
    rows.append((float((h - t).norm() / max(float(t.norm()), 1e-30)), float(t.norm()) / gn, nm))
for e, share, nm in rows:
    if 'key_linear_b.bias' not in nm and (e > 2e-5 or os.environ.get('ALL')):
        print('%.3e  share %.2e  %s' % (e, share, nm))
# intermediate activations of the conv stack
with torch.no_grad():
    acts = [x]
    for layer in oracle.conv:
        acts.append(layer(acts[-1]))
A = model.engine.arena
def cmp(nm, ours, ref):
    ref = ref.permute(0, 3, 2, 1)
    print(nm, 'rel', float((ours.cpu() - ref).norm() / ref.norm()), 'max abs', float((ours.cpu() - ref).abs().max()))
cmp('y1', A['y1'], acts[2]); cmp('p1', A['p1'], acts[5]); cmp('y5', A['y5'], acts[7]); cmp('p2', A['p2'], acts[10])
