"""GPU diagnostic: where does the meta-step error come from? compares g_tr, theta', g_val(theta') stage by stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import golden_util as gu
from tests.test_parity_gpu import make
from oracle import refimpl as R

name = sys.argv[1] if len(sys.argv) > 1 else 'F0'
z, cfg, spec = gu.load(name)
mtl_amd, args, vocab, model = make(cfg, spec)
model = model.cuda()
oracle = R.build_model(cfg)
tr, val = gu.batches_for(cfg, spec, 0, z['data_call_index'])
x, lens, y = tr[0]
params = list(oracle.parameters())
names = [n for n, _ in oracle.named_parameters()]
L = model._layout

def flat_of(ts):
    f = torch.zeros(L.total)
    for nm, t in zip(names, ts):
        L.view(f, nm).copy_(t)
    return f

def table(tag, mine, ref_list, top=6):
    gn = float(torch.sqrt(sum((t.double() ** 2).sum() for t in ref_list)))
    rows = []
    for nm, t in zip(names, ref_list):
        h = L.view(mine, nm).cpu()
        rows.append((float((h - t).norm() / max(float(t.norm()), 1e-4 * gn)), nm))
    print(tag, ' | '.join('%.2e %s' % r for r in sorted(rows, reverse=True)[:top]))

pred, gold, _ = oracle(x, lens, y)
g_tr = torch.autograd.grad(R.ce_loss(pred, gold), params)
theta0 = [p.detach().clone() for p in params]
with torch.no_grad():
    for p, g in zip(params, g_tr):
        p.add_(g, alpha=-spec['lr'])
theta1 = [p.detach().clone() for p in params]
pv, gv, _ = oracle(*val)
g_val = torch.autograd.grad(R.ce_loss(pv, gv), params)

g = torch.zeros_like(model.flat_grad)
model.pass_forward(x.cuda(), lens, y)
model.pass_backward(g, 1.0)
table('g_tr      ', g, g_tr)
inner = mtl_amd.FlatSGD(model, spec['lr'])
t1 = inner.theta_prime_from(model.flat_parameters, g)
table('theta1    ', t1, theta1)
# (a) val grad at MY theta1
g2 = torch.zeros_like(g)
out = model.pass_forward(val[0].cuda(), val[1], val[2], theta=t1)
model.pass_backward(g2, 1.0)
print('val loss', float(out['loss']), float(R.ce_loss(pv, gv)))
table('g_val@mine', g2, g_val)
# (b) val grad at the ORACLE's theta1 uploaded
t1o = flat_of(theta1).cuda()
g3 = torch.zeros_like(g)
model.pass_forward(val[0].cuda(), val[1], val[2], theta=t1o)
model.pass_backward(g3, 1.0)
table('g_val@orcl', g3, g_val)
# (c) val grad at theta0 (no inner step) both sides
with torch.no_grad():
    for p, t0 in zip(params, theta0):
        p.copy_(t0)
pv0, gv0, _ = oracle(*val)
g_val0 = torch.autograd.grad(R.ce_loss(pv0, gv0), params)
g4 = torch.zeros_like(g)
model.pass_forward(val[0].cuda(), val[1], val[2])
model.pass_backward(g4, 1.0)
table('g_val@th0 ', g4, g_val0)

# ---- sensitivity probes
def mygrad(theta):
    gg = torch.zeros_like(g)
    model.pass_forward(val[0].cuda(), val[1], val[2], theta=theta)
    model.pass_backward(gg, 1.0)
    return gg
def relerr(a, b, nm):
    a, b = L.view(a, nm), L.view(b, nm)
    return float((a - b).norm() / b.norm())
ga = mygrad(t1); gb = mygrad(t1)
print('repeat@t1 conv.0.weight', relerr(ga, gb, 'conv.0.weight'))
t1c = t1.clone()
print('clone@t1 conv.0.weight', relerr(mygrad(t1c), ga, 'conv.0.weight'))
for eps in (1e-8, 1e-7, 1e-6):
    tp = t1 * (1 + eps * torch.randn_like(t1))
    gp = mygrad(tp)
    print('perturb %.0e:' % eps, ' '.join('%s %.2e' % (nm, relerr(gp, ga, nm)) for nm in ('conv.0.weight', 'conv.2.weight', 'conv.7.weight', 'encoder.input_linear.weight')))
# same perturbation study on the oracle (CPU)
def oracle_grad(theta_flat):
    with torch.no_grad():
        for nm, p in zip(names, params):
            p.copy_(L.view(theta_flat, nm))
    pv_, gv_, _ = oracle(*val)
    return flat_of(torch.autograd.grad(R.ce_loss(pv_, gv_), params))
base = oracle_grad(t1.cpu())
for eps in (1e-8, 1e-7, 1e-6):
    tp = t1.cpu() * (1 + eps * torch.randn(t1.numel()))
    gp = oracle_grad(tp)
    print('oracle perturb %.0e:' % eps, ' '.join('%s %.2e' % (nm, relerr(gp, base, nm)) for nm in ('conv.0.weight', 'conv.2.weight', 'conv.7.weight', 'encoder.input_linear.weight')))
print('oracle(t1) vs mine(t1): conv.0.weight', relerr(ga.cpu(), base, 'conv.0.weight'))
print('lens of val', val[1].tolist())
