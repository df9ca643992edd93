"""Runs N meta-steps of the bench workload and nothing else (for rocprofv3 --kernel-trace timelines, tools/timeline.py).
usage: python tools/run_steps.py [--tasks 1] [--steps 6] [--warmup 3] [--serial]"""
import argparse
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--tasks', type=int, default=1)
ap.add_argument('--total-tasks', type=int, default=8)
ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--warmup', type=int, default=3)
ap.add_argument('--frames', type=int, default=1000)
ap.add_argument('--serial', action='store_true')
ap.add_argument('--no-side', action='store_true', help='weight-gradient jobs on the lane stream itself (no side streams)')
a = ap.parse_args()
with contextlib.redirect_stdout(io.StringIO()):
    import mtl_amd
mdist = mtl_amd.dist
mdist.init_from_env()
dev = torch.device('cuda', 0)
args = bench.make_args(8)
vocab = mtl_amd.synthetic_vocab(bench.CFG['vocab_size'])
torch.manual_seed(123456)
with contextlib.redirect_stdout(io.StringIO()):
    model = mtl_amd.init_transformer_model(args, vocab, r=bench.CFG['r']).to(dev)
trainer = mtl_amd.TransientTrainer()
if a.serial:
    model.n_lanes = 1
    trainer.use_cmdlists = False
    for e in model.engines:
        e.use_side_stream = False
if a.no_side:
    for e in model.engines:
        e.use_side_stream = False
inner, outer = mtl_amd.FlatSGD(model, args.lr), mtl_amd.FlatAdam(model, args.meta_lr)
model.zero_copy_grad()
tasks = [bench.ResidentTask(mtl_amd, m, 8, a.frames, 100, bench.CFG['vocab_size'], dev) for m in range(a.tasks)]
dt, last = bench.timed_steps(trainer, model, vocab, tasks, list(range(a.tasks)), a.total_tasks, inner, outer, args, a.steps, a.warmup,
                             mdist, dev)
print('%d task(s): %.3f ms per step' % (a.tasks, dt / a.steps * 1e3))
