"""GPU timing: Transformer.evaluate greedy decoding (B=8, T=1000, 300 steps) vs the oracle's reference-style loop (30 steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import golden_util as gu
from tests.test_parity_gpu import make
from oracle import refimpl as R
z, cfg, spec = gu.load('NS')
mtl_amd, args, vocab, model = make(cfg, spec)
model = model.cuda()
x, lens, y = mtl_amd.synth_batch(3, 8, 1000, 100, cfg['vocab_size'])
xd = x.cuda()
model.evaluate(xd, lens, y, args, start_token=1, max_steps=8)
torch.cuda.synchronize()
t0 = time.perf_counter()
_, hyps, _ = model.evaluate(xd, lens, y, args, start_token=1, max_steps=300)
torch.cuda.synchronize()
t_hip = time.perf_counter() - t0
torch.set_num_threads(32)
oracle = R.build_model(cfg)
t0 = time.perf_counter()
ref = R.greedy_search(oracle, x, lens, 1, 30)
t_cpu30 = time.perf_counter() - t0
print('HIP greedy 300 steps B=8: %.3f s (%.2f ms/step)' % (t_hip, t_hip / 300 * 1e3))
print('oracle (reference-style full recompute, 32 threads) first 30 steps: %.2f s; ids equal on those steps: %s'
      % (t_cpu30, bool(torch.equal(model.last_greedy_ids[:30].t().contiguous(), ref))))
