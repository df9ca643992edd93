"""What a manifest-fed run sees: every task's batch padded to its OWN longest utterance (data.py collate), so frame counts differ
from task to task and from step to step.  Measures meta-steps/s of the production loop on such batches against the fixed-shape
workload of bench.py.  Usage: python tools/probe/ragged_steps.py [--steps 20] [--lo 600] [--hi 1000] [--mode ragged|fixed|padmax]"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


RaggedTask = bench.RaggedTask


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--tasks', type=int, default=8)
    ap.add_argument('--k', type=int, default=8)
    ap.add_argument('--lo', type=int, default=600)
    ap.add_argument('--hi', type=int, default=1000)
    ap.add_argument('--labels', type=int, default=100)
    ap.add_argument('--mode', default='ragged')
    ap.add_argument('--lanes', action='store_true', help='a lane per task instead of the stacked pass (trainer.batch_ragged = False)')
    ap.add_argument('--own-widths', action='store_true', help='... at every batch\'s own width (trainer.pad_lanes = \'0\'): the reference\'s schedule')
    ap.add_argument('--vary-labels', action='store_true', help='label width of every batch drawn from labels / 2 ... labels')
    ap.add_argument('--hetero', type=float, default=1.0, help='task m draws its utterances from [lo, hi] scaled by hetero + (1 - hetero) m / (n - 1): corpora of different utterance lengths')
    a = ap.parse_args()
    with contextlib.redirect_stdout(io.StringIO()):
        import mtl_amd
    mtl_amd.hostenv.bound_torch_threads()
    dev = torch.device('cuda', 0)
    args = bench.make_args(a.k)
    vocab = mtl_amd.synthetic_vocab(bench.CFG['vocab_size'])
    torch.manual_seed(123456)
    with contextlib.redirect_stdout(io.StringIO()):
        model = mtl_amd.init_transformer_model(args, vocab, r=bench.CFG['r']).to(dev)
    trainer = mtl_amd.TransientTrainer()
    if a.lanes:
        trainer.batch_ragged = False
    if a.own_widths:
        trainer.pad_lanes = '0'
    inner, outer = mtl_amd.FlatSGD(model, args.lr), mtl_amd.FlatAdam(model, args.meta_lr)
    model.zero_copy_grad()
    sc = [a.hetero + (1 - a.hetero) * m / max(a.tasks - 1, 1) for m in range(a.tasks)]
    tasks = [RaggedTask(m, a.k, int(a.lo * sc[m]), int(a.hi * sc[m]), a.labels, bench.CFG['vocab_size'], dev, a.mode) for m in range(a.tasks)]

    for t in tasks:
        t.vary_labels = a.vary_labels

    def batches():
        return [t.batch() for t in tasks], tasks[-1].batch()
    pending, host, frames = [], [], 0
    depth = max(getattr(trainer, 'pipeline_depth', 1), 1)
    for phase, n in (('setup', 4), ('timed', a.steps)):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            local, val = batches()
            if phase == 'timed':
                frames += sum(int(b[0].shape[0] * b[0].shape[3]) for b in local) + len(local) * int(val[0].shape[0] * val[0].shape[3])
            h0 = time.perf_counter()
            pending.append(trainer.enqueue_iteration(model, vocab, local, val, a.tasks, inner, outer, args))
            host.append((time.perf_counter() - h0) * 1e3)
            while len(pending) > depth:
                last = pending.pop(0).result()
        while pending:
            last = pending.pop(0).result()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    eng = model.engines[0]
    print(json.dumps(dict(mode=a.mode, ms_per_step=round(dt / a.steps * 1e3, 2), meta_steps_per_s=round(a.steps / dt, 2),
                          frames_per_step=frames // a.steps, us_per_kiloframe=round(dt / frames * 1e9, 1),
                          host_enqueue_ms=dict(mean=round(sum(host[4:]) / len(host[4:]), 1), max=round(max(host[4:]), 1)),
                          pool_gb=round(sum(e._pool_bytes for e in model.engines) / 2 ** 30, 2), cmdlists=len(getattr(trainer, '_cmdlists', {})),
                          mem_reserved_gb=round(torch.cuda.memory_reserved(dev) / 2 ** 30, 1), host_rss_gb=round(__import__('psutil').Process().memory_info().rss / 2 ** 30, 2),
                          allocator_retries=torch.cuda.memory_stats(dev).get('num_alloc_retries', 0), account_gb=round(eng.account['bytes'] / 2 ** 30, 1), schedule=trainer.last_schedule, loss=float(last[0]))))


if __name__ == '__main__':
    main()
