"""GPU diagnostic: host-side enqueue time of one pass vs its GPU time (is the loop launch-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_util as gu
from tests.test_parity_gpu import make
z, cfg, spec = gu.load('NS')
mtl_amd, args, vocab, model = make(cfg, spec)
model = model.cuda()
x, lens, y = mtl_amd.synth_batch(7, 8, 1000, 100, cfg['vocab_size'])
xd = x.cuda()
g = torch.zeros_like(model.flat_grad)
for _ in range(2):
    model.pass_forward(xd, lens, y); model.pass_backward(g, 1.0)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    model.pass_forward(xd, lens, y)
    t1 = time.perf_counter()
    model.pass_backward(g, 1.0)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print('enqueue fwd %.2f ms, bwd %.2f ms, drain %.2f ms, total %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
