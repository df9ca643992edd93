"""bench.py -- meta-steps/sec of the `--copy-grad` meta-transfer step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = one meta-iteration of `TransientTrainer` (the reference's timed span, trainer/asr/transient_trainer.py:152-264):
every task does {train forward+backward at theta0, fused inner SGD, validation forward+backward at theta'}, copy_grad
accumulation, ONE all-reduce of the flat meta-gradient (N > 1), Adam, and the loss/CER read-back.

Workload (all N): 8 synthetic meta-tasks, k_train = k_valid = 8 utterances of 1000 frames x 161 bins, 100 labels,
enc2/dec4 d512 h8 r100 V=3765, fp32, dropout 0 -- tasks sharded round-robin over the ranks (8/N per GPU, strong scaling;
SURVEY.md 8(d)).  The README-faithful 3-task/1-GPU configuration (BASELINE.json configs[1]) is timed in
the same run and reported in `configs1_3task`.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (live HIP-event timing of the dominant
kernel on its launch stream) and `cpu_baseline` (the CPU oracle on the host cores, bounded sample, N = 1 only).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CFG = dict(num_enc_layers=2, num_dec_layers=4, num_heads=8, dim_model=512, dim_key=64, dim_value=64, dim_inner=512,
           dim_emb=512, src_max_len=5000, tgt_max_len=2500, r=100, vocab_size=3765)
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, no TF32 on gfx950
PEAK_BF16_MFMA_TFLOPS = 2516.6    # same guide: v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate, dense
# The split-bf16 ("x3") convolution kernels issue SIX bf16 MFMAs per fp32-equivalent multiply-accumulate step, so the
# roof of their ALGORITHMIC (fp32-equivalent) FLOP rate is the dense bf16 peak / 6.
PEAK_X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def conv_arithmetic(engine, name):
    """Which MFMA pipe a conv-stack kernel class runs on (mirrors the choices in PassEngine.forward/backward)."""
    if not engine.conv_x3 or name.startswith('conv0'):
        return 'f32'
    if name == 'conv5_wgrad' and not engine.wgrad_x3_dense:
        return 'f32'
    return 'x3'


class ResidentTask:
    """Synthetic task whose (train, valid) batches already live in HBM (the contract's `.sample` duck-type)."""

    def __init__(self, mtl, task_id, k, T, L, V, device):
        def mk(part):
            x, lens, y = mtl.synth_batch(10 * task_id + part, k, T, L, V)
            return (x.to(device), lens, lens.float() / T, y, (y != 0).sum(1).to(torch.int32))
        self.batches = (mk(0), mk(1))

    def sample(self, k_train, k_valid, manifest_id):
        return self.batches


def make_args(k, lr=1e-4, meta_lr=1e-4):
    return argparse.Namespace(feat_extractor='vgg_cnn', sample_rate=16000, window_size=.02, feat='spectrogram', dim_input=161,
                              dropout=0.0, emb_trg_sharing=False, label_smoothing=0.0, name='bench', lr=lr, meta_lr=meta_lr,
                              k_train=k, k_valid=k, clip=False, max_norm=400, save_every=10 ** 9, save_folder='/tmp/mtl_bench',
                              cuda=True, **{a: b for a, b in CFG.items() if a not in ('vocab_size', 'r')})


def conv_rows(prof):
    rows = []
    for name, (flops, evs) in prof.items():
        times = [s.elapsed_time(e) * 1e-3 for s, e in evs]
        rows.append((sum(times), name, flops, sum(times) / len(times), len(times)))
    rows.sort(reverse=True)
    return rows


def serial_profile(trainer, model, vocab, tasks, my_tasks, n_tasks, inner, outer, args, dev):
    """One extra meta-iteration with task lanes and the side stream switched off, HIP events around every conv launch:
    isolated kernel durations (what rocprofv3 reports for a non-overlapped dispatch)."""
    lanes, model.n_lanes = model.n_lanes, 1
    for e in model.engines:
        e.use_side_stream = False
    prof = {}
    model.engines[0].prof = prof
    val = tasks[-1].sample(0, 0, 0)[1]
    local = [tasks[m].sample(0, 0, m)[0] for m in my_tasks]
    trainer.run_iteration(model, vocab, local, val, n_tasks, inner, outer, args)
    torch.cuda.synchronize(dev)
    model.engines[0].prof = None
    model.n_lanes = lanes
    for e in model.engines:
        e.use_side_stream = True
    return conv_rows(prof)


def timed_steps(trainer, model, vocab, tasks, my_tasks, n_tasks, inner, outer, args, steps, warmup, mdist, dev, profile=False):
    val = tasks[-1].sample(0, 0, 0)[1]
    local = [tasks[m].sample(0, 0, m)[0] for m in my_tasks]

    def one():
        return trainer.run_iteration(model, vocab, local, val, n_tasks, inner, outer, args)
    for _ in range(warmup):
        one()
    mdist.barrier()
    torch.cuda.synchronize(dev)
    if profile:
        prof = {}                                              # HIP events around the conv launches, on their stream
        for e in model.engines:
            e.prof = prof
    t0 = time.perf_counter()
    for _ in range(steps):
        last = one()
    torch.cuda.synchronize(dev)
    mdist.barrier()
    dt = time.perf_counter() - t0
    if mdist.world_size() > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    return dt, last


def cpu_baseline(n_tasks, k, T, L, threads):
    """The CPU oracle (bit-pinned restatement of the reference) on the host cores: ONE task of the same
    workload (train pass + validation pass, each forward+backward) + the Adam step, extrapolated to n_tasks."""
    from oracle import refimpl as R
    torch.set_num_threads(threads)
    model = R.build_model(CFG)
    adam = R.AdamState(list(model.parameters()), 1e-4)
    tr = [R.synth_batch(0, k, T, L, CFG['vocab_size'])]
    val = R.synth_batch(1, k, T, L, CFG['vocab_size'])
    R.meta_gradient(model, tr, val, 1e-4)                      # warm-up (thread pools, oneDNN primitive cache)
    t0 = time.perf_counter()
    G, _, _, _ = R.meta_gradient(model, tr, val, 1e-4)
    t_task = time.perf_counter() - t0
    t0 = time.perf_counter()
    adam.step(list(model.parameters()), G)
    t_outer = time.perf_counter() - t0
    return dict(value=1.0 / (n_tasks * t_task + t_outer), unit='meta-steps/s', cores=threads, kind='port',
                sample='1 of %d tasks timed after 1 warm-up task (2 fwd+bwd passes, %.2f s) + Adam step (%.3f s), x%d tasks'
                       % (n_tasks, t_task, t_outer, n_tasks), seconds_per_task=t_task)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--tasks', type=int, default=8)
    ap.add_argument('--k', type=int, default=8)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--labels', type=int, default=100)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--serial', action='store_true', help='no task lanes / side stream (for rocprofv3 per-kernel durations)')
    a = ap.parse_args()

    with contextlib.redirect_stdout(io.StringIO()):          # the model factory prints; keep stdout to ONE JSON line
        import mtl_amd
    mdist = mtl_amd.dist
    local_rank = mdist.init_from_env()
    world, rank = mdist.world_size(), mdist.rank()
    if world != a.gpus and not (a.gpus == 1 and world == 1):
        raise SystemExit('--gpus %d but WORLD_SIZE %d' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    args = make_args(a.k)
    vocab = mtl_amd.synthetic_vocab(CFG['vocab_size'])
    torch.manual_seed(123456)
    with contextlib.redirect_stdout(io.StringIO()):
        model = mtl_amd.init_transformer_model(args, vocab, r=CFG['r']).to(dev)
    if a.serial:
        model.n_lanes = 1
        for e in model.engines:
            e.use_side_stream = False
    trainer = mtl_amd.TransientTrainer()
    inner, outer = mtl_amd.FlatSGD(model, args.lr), mtl_amd.FlatAdam(model, args.meta_lr)
    model.zero_copy_grad()
    tasks = [ResidentTask(mtl_amd, m, a.k, a.frames, a.labels, CFG['vocab_size'], dev) for m in range(a.tasks)]
    my_tasks = mdist.shard_tasks(a.tasks, rank, world)

    dt, last = timed_steps(trainer, model, vocab, tasks, my_tasks, a.tasks, inner, outer, args, a.steps, a.warmup, mdist, dev,
                           profile=True)
    prof = model.engine.prof
    for e in model.engines:
        e.prof = None

    # every rank runs the serial profiling step (it contains the collective); only rank 0 reports it
    conc = conv_rows(prof)
    rows = serial_profile(trainer, model, vocab, tasks, my_tasks, a.tasks, inner, outer, args, dev)
    out = None
    if rank == 0:
        ms = dt / a.steps * 1e3
        # dominant kernel = the conv class with the largest accumulated time; its duration is taken from a serial profiling
        # step (lanes / side stream off) because HIP events around a launch that shares the GPU with another lane's kernels
        # measure the sharing, not the kernel; the concurrent figures of the timed region are kept next to it
        tot, name, flops, avg, cnt = rows[0]
        conv_time = sum(r[0] for r in rows)
        conv_flops = sum(r[2] * r[4] for r in rows)
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        except Exception:
            pass
        arith = conv_arithmetic(model.engine, name)
        peak = PEAK_X3_TFLOPS if arith == 'x3' else PEAK_F32_MFMA_TFLOPS
        peak_of = lambda n: PEAK_X3_TFLOPS if conv_arithmetic(model.engine, n) == 'x3' else PEAK_F32_MFMA_TFLOPS
        roofline = dict(bound='mfma', kernel=name, achieved=flops / avg / 1e12, peak=peak, unit='TFLOP/s',
                        frac=flops / avg / 1e12 / peak, traffic=pmc.get(name), gflop_per_launch=flops / 1e9,
                        arithmetic=('split-bf16 x3: 6 v_mfma_f32_32x32x16_bf16 per fp32-equivalent step, peak = dense bf16 / 6'
                                    if arith == 'x3' else 'v_mfma_f32_32x32x2_f32'),
                        avg_launch_ms=avg * 1e3, launches_timed=cnt, timing='HIP events, serial profiling step after the timed region',
                        conv_stack=dict(tflops=conv_flops / conv_time / 1e12, ms_per_pass=conv_time / (2 * len(my_tasks)) * 1e3,
                                        per_kernel={r[1]: dict(ms=r[3] * 1e3, tflops=r[2] / r[3] / 1e12,
                                                               frac=r[2] / r[3] / 1e12 / peak_of(r[1])) for r in rows}),
                        timed_region_concurrent={r[1]: dict(ms=r[3] * 1e3, tflops=r[2] / r[3] / 1e12, launches=r[4]) for r in conc})
        out = dict(metric='meta-steps/sec', value=a.steps / dt, unit='meta-steps/s', n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=ms, higher_is_better=True, scaling='strong', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='meta_transfer_train --copy-grad, enc2/dec4 d512 h8 r100 V3765, %d synthetic tasks '
                                        '(%d per GPU), k_train=k_valid=%d, %d frames x 161 bins, %d labels, dropout 0'
                                        % (a.tasks, len(my_tasks), a.k, a.frames, a.labels),
                               tasks=a.tasks, k_train=a.k, src_frames=a.frames, tgt_len=a.labels, parallelism='task-sharded dp%d' % world,
                               collective=mdist.backend_name(),
                               conv_arithmetic=('3x3 convolutions as exact 3-way bf16 splits of fp32 operands, fp32 accumulate '
                                                '(fp32-class error, same test tolerances as the fp32-MFMA kernels)'
                                                if model.engine.conv_x3 else 'fp32 MFMA')),
                   roofline=roofline, last_step=dict(val_loss=last[0] / a.tasks, cer_edits=last[1], chars=last[2]))

    if world == 1:
        # README-faithful configuration (BASELINE.json configs[1]): 3 tasks on one GPU, same run
        k3 = max(a.steps // 2, 3)
        if a.tasks >= 3:
            dt3, _ = timed_steps(trainer, model, vocab, tasks[:3], [0, 1, 2], 3, inner, outer, args, k3, 3, mdist, dev)
            out['configs1_3task'] = dict(value=k3 / dt3, unit='meta-steps/s', ms_per_step=dt3 / k3 * 1e3,
                                         note='README-faithful: 3 tasks on one GPU, dropout 0 (parity setting)')
        # the README trains with --dropout 0.1: same 8-task workload with the Philox dropout active
        model.encoder.dropout_rate = model.decoder.dropout_rate = 0.1
        model.train()
        dtd, _ = timed_steps(trainer, model, vocab, tasks, my_tasks, a.tasks, inner, outer, args, k3, 3, mdist, dev)
        out['dropout_0.1'] = dict(value=k3 / dtd, unit='meta-steps/s', ms_per_step=dtd / k3 * 1e3, tasks=a.tasks)
        model.encoder.dropout_rate = model.decoder.dropout_rate = 0.0
        model.train()
        if not a.no_cpu_baseline:
            threads = a.cpu_threads or min(os.cpu_count() or 8, 32)
            out['cpu_baseline'] = cpu_baseline(a.tasks, a.k, a.frames, a.labels, threads)
    if rank == 0:
        print(json.dumps(out), flush=True)
    mdist.barrier()


if __name__ == '__main__':
    main()
