"""`TransientTrainer`: drop-in for trainer/asr/transient_trainer.py (the `--copy-grad` meta-transfer loop).

Same `train(...)` / `forward_one_batch(...)` signatures, log lines and meta-gradient definition
    G = sum_m [ grad L_tr,m(theta0) + (1/n) grad L_val(theta0 - alpha grad L_tr,m(theta0)) ]      
but the iteration is restructured for the device:
  * theta0 is never mutated inside an iteration (theta' is materialised by one fused kernel into a second flat
    buffer), which removes deepcopy(state_dict) / n x load_state_dict (transient_trainer.py:155-160,237);
  * gradients, copy_grad, Adam moments are single flat fp32 buffers updated by one kernel each;
  * label/loss read-backs are asynchronous copies resolved once per iteration instead of ~1600 `int(x)` device
    syncs per forward (transient_trainer.py:29-35,46);
  * with torch.distributed initialised, tasks are sharded round-robin over ranks and the flat G is summed with ONE
    all-reduce (RCCL over xGMI) before the (replicated, deterministic) Adam step.
"""
import collections
import logging
import os
import threading
import time
from collections import deque

import torch

from . import _lib, _trace, dist as mdist, hostenv
from .engine import CENSUS_NAMES, CENSUS_SLOTS, PAD_ID, round_width
from .functions import post_process, save_joint_model, save_meta_model
from .metrics import calculate_cer, calculate_metrics

check = _lib.check


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) over the model's flat parameter buffer."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.m = torch.zeros_like(model.flat_parameters)
        self.v = torch.zeros_like(model.flat_parameters)
        self.step_count = 0
        self.param_groups = [{'lr': lr, 'betas': betas, 'eps': eps}]

    @classmethod
    def from_torch(cls, model, opt):
        """Import the state of a torch.optim.Adam built over model.parameters() (reference checkpoints pickle those)."""
        g = opt.param_groups[0]
        self = cls(model, g['lr'], tuple(g['betas']), g['eps'])
        lay = model._layout
        for name, p in zip(lay.order, opt.param_groups[0]['params']):
            st = opt.state.get(p, None)
            if st:
                lay.view(self.m, name).copy_(st['exp_avg'])
                lay.view(self.v, name).copy_(st['exp_avg_sq'])
                self.step_count = int(st['step'])
        return self

    def to_torch(self):
        opt = torch.optim.Adam(self.model.parameters(), lr=self.param_groups[0]['lr'], betas=self.betas, eps=self.eps)
        if self.step_count:
            lay = self.model._layout
            for name, p in zip(lay.order, self.model.parameters()):
                opt.state[p] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': lay.view(self.m, name).clone(),
                                'exp_avg_sq': lay.view(self.v, name).clone()}
        return opt

    def zero_grad(self):
        self.model.zero_grad()

    def step(self, grad=None):
        m = self.model
        g = m.flat_grad if grad is None else grad
        self.step_count += 1
        th = m.flat_parameters
        check(_lib.lib().mtl_adam_step(torch.cuda.current_stream(th.device).cuda_stream, th.data_ptr(), g.data_ptr(),
                                       self.m.data_ptr(), self.v.data_ptr(), self.step_count, float(self.param_groups[0]['lr']),
                                       self.betas[0], self.betas[1], self.eps, th.numel()), 'mtl_adam_step')


class FlatSGD:
    """torch.optim.SGD(lr) (no momentum / weight decay) as theta' = theta0 - lr*g into a separate buffer."""

    def __init__(self, model, lr):
        self.model = model
        self.param_groups = [{'lr': lr}]
        self.theta_prime = torch.empty_like(model.flat_parameters)

    @classmethod
    def from_torch(cls, model, opt):
        return cls(model, opt.param_groups[0]['lr'])

    def to_torch(self):
        return torch.optim.SGD(self.model.parameters(), lr=self.param_groups[0]['lr'])

    def zero_grad(self):
        self.model.zero_grad()

    def theta_prime_from(self, theta0, grad, out=None):
        out = self.theta_prime if out is None else out
        check(_lib.lib().mtl_sgd_theta_prime(torch.cuda.current_stream(theta0.device).cuda_stream, theta0.data_ptr(),
                                             grad.data_ptr(), float(self.param_groups[0]['lr']), out.data_ptr(),
                                             theta0.numel()), 'mtl_sgd_theta_prime')
        return out


def clip_flat_grad_(model, grad, max_norm, lane=0):
    """torch.nn.utils.clip_grad_norm_ on the flat buffer: grad *= min(1, max_norm / (||grad|| + 1e-6)), no host sync."""
    model._need_engine()
    eng = model.engines[lane]
    ws = eng.scratch(8192)
    coef = eng.buf('_clip_coef', (1,))
    st = eng.stream
    check(eng.lib.mtl_sumsq(st, grad.data_ptr(), grad.numel(), coef.data_ptr(), ws, 2, float(max_norm)), 'mtl_sumsq')
    check(eng.lib.mtl_scale(st, grad.data_ptr(), 1.0, coef.data_ptr(), grad.numel()), 'mtl_scale')


_PINNED = collections.OrderedDict()
_PINNED_MAX = 512          # variable-length data brings new read-back shapes all the time: least recently used entries go


def _pinned(key, shape, dtype, turns=1):
    """Page-locked host buffer of a read-back call site: pin_memory() costs a host allocation + registration that serialises against
    the device, and the read-backs of one iteration are only consumed after that iteration's device sync, so the memory is re-used.
    `key` ends with the index of the read-back set (`_turn`).  The cache holds RAW bytes per (site, set), sized for the largest shape
    seen so far and grown geometrically for ALL `turns` sets at once: variable-length data brings a new shape with almost every batch
    and must not bring a page-locked allocation with it, and none may happen a few iterations later inside a timed region."""
    shape = tuple(int(v) for v in shape)
    nbytes = torch.empty((), dtype=dtype).element_size()
    for v in shape:
        nbytes *= v
    raw = _PINNED.get(key)
    if raw is None or raw.numel() < nbytes:
        cap = max(nbytes + nbytes // 2, 256)
        for turn in range(turns):
            _PINNED[(key[:-1] + (turn,)) if turns > 1 else key] = torch.empty(cap, dtype=torch.uint8).pin_memory()
        raw = _PINNED[key]
        while len(_PINNED) > _PINNED_MAX:
            _PINNED.popitem(last=False)
    else:
        _PINNED.move_to_end(key)
    return raw[:nbytes].view(dtype).view(shape)


def _label_width(targets, quantum, most):
    """decoder positions of a pass on these padded targets (longest label sequence + 1), rounded up to a multiple of `quantum`: label
    widths change from batch to batch like the frame counts, and a wider decoder is exact (positions beyond a sequence are padding:
    masked as keys, zeroed as rows, outside the loss; their PAD labels leave the CER strings)"""
    own = max(int((t != PAD_ID).sum(1).max()) for t in targets) + 1
    q = max(int(quantum), 1)
    return max(min(-(-own // q) * q, most), own)


class PendingIteration:
    """An enqueued meta-iteration: result() waits for ITS end-of-iteration event (not for the device, which may already be
    running the next iteration), then computes the reference's log quantities from the read-backs."""

    def __init__(self, reads, done, vocab, dev, census=None):
        self.reads, self.done, self.vocab, self.dev = reads, done, vocab, dev
        self.census = census          # (pinned counters, callback) of an iteration that sampled the h2 census
        self._result = None

    def result(self):
        if self._result is None:
            self.done.synchronize()
            total_loss, total_cer, total_char = 0.0, 0, 0
            for tr_read, va_read in self.reads:
                c, n = cer_counts(self.vocab, tr_read.gold_host, tr_read.hyp)    # the reference reports the TRAIN batches' CER
                total_cer += c
                total_char += n
                total_loss += float(va_read.loss[0])
            self._result = mdist.allreduce_scalars([total_loss, total_cer, total_char], self.dev)
            self.reads = None
            if self.census is not None:
                counters, report = self.census
                self.census = None
                report(counters)
        return self._result


class _Readback:
    """Asynchronous D2H of (gold, hyp, loss) of one forward; resolved after the iteration's single sync.  `key` names the
    call site (task index, pass) whose pinned buffers are re-used from iteration to iteration."""

    def __init__(self, out, key, turns=1):
        self.gold_host = out['gold_host']
        self.hyp = _pinned(('hyp',) + tuple(key), out['hyp'].shape, torch.int64, turns)
        self.loss = _pinned(('loss',) + tuple(key), (1,), torch.float32, turns)
        self.hyp.copy_(out['hyp'], non_blocking=True)
        self.loss.copy_(out['loss'], non_blocking=True)


class _TaskRead:
    """One task's share of a task-batched pass's read-backs: views into the pinned whole-pass buffers (valid once the
    iteration's end event has been waited for)."""

    def __init__(self, gold_host, hyp, loss):
        self.gold_host, self.hyp, self.loss = gold_host, hyp, loss


def _strings(vocab, rows):
    return [''.join(vocab.id2label[int(t)] for t in row) for row in rows.tolist()]


def cer_counts(vocab, gold, hyp):
    """total edit distance / reference length over a batch, exactly as transient_trainer.py:29-35,52-64."""
    total_cer, total_char = 0, 0
    for g, h in zip(_strings(vocab, gold), _strings(vocab, hyp)):
        h = post_process(h, vocab.special_token_list).replace(' ', '')
        g = post_process(g, vocab.special_token_list).replace(' ', '')
        total_cer += calculate_cer(h, g)
        total_char += len(g)
    return total_cer, total_char


def sync_replicas_from_rank0(model, opts):
    """Multi-rank start-up: broadcast theta and every optimizer's state (Adam m, v, step count) from rank 0, so that replicas are
    bit-identical whatever each rank's RNG / checkpoint state was.  From then on they stay identical by construction (same G
    after the all-reduce, deterministic Adam kernel); check_replicas() verifies that periodically."""
    if mdist.world_size() <= 1:
        return
    import torch.distributed as td
    td.broadcast(model.flat_parameters, src=0)
    for opt in opts:
        if isinstance(opt, FlatAdam):
            td.broadcast(opt.m, src=0)
            td.broadcast(opt.v, src=0)
            step = torch.tensor([opt.step_count], dtype=torch.int64, device=model.flat_parameters.device)
            td.broadcast(step, src=0)
            opt.step_count = int(step.item())


def check_replicas(model, val_batch, it):
    """Cheap divergence check (two small all-reduces: float checksums of theta and the validation batch, bit-level integer
    checksums of theta): every rank must hold the same theta and must have drawn the same
    shared validation batch (ManifestTaskDataset(seed=...) / the seeding contract in INTEGRATION.md).  Raises on a mismatch
    instead of letting the all-reduced meta-gradient silently mix gradients taken at different parameters."""
    if mdist.world_size() <= 1:
        return
    import torch.distributed as td
    th = model.flat_parameters
    vx = val_batch[0]
    probe = torch.stack([th.double().sum(), th[::997].double().abs().sum(),
                         vx.to(th.device, non_blocking=True).double().sum() + float(val_batch[3].sum())])
    if not bool(torch.isfinite(probe).all()):
        # (MIN / MAX never compare equal on NaN: without this check a numerical blow-up would be reported as a seeding problem)
        raise RuntimeError('iteration %d: non-finite parameters or validation batch on rank %d (checksums %s): the run diverged '
                           'numerically, this is not a replica mismatch' % (it + 1, mdist.rank(), probe.tolist()))
    # float checksums: MAX of [p, -p] gives max and -min with one all-reduce
    t = torch.cat([probe, -probe])
    td.all_reduce(t, op=td.ReduceOp.MAX)
    hi, lo = t[:3], -t[3:]
    # bit-level checksums of theta (a localised or sign-cancelling divergence does not survive these): integer column sums of the
    # raw 32-bit words, plain and position-weighted (wrap-around int64 arithmetic is deterministic)
    bits = th.view(torch.int32)
    n = bits.numel() // 4096 * 4096
    cols = bits[:n].view(-1, 4096).sum(0, dtype=torch.int64)
    tail = bits[n:].sum(dtype=torch.int64)
    chk = torch.stack([cols.sum() + tail, (cols * torch.arange(1, 4097, dtype=torch.int64, device=th.device)).sum() + 7 * tail])
    u = torch.cat([chk, -chk])
    td.all_reduce(u, op=td.ReduceOp.MAX)
    if not torch.equal(lo, hi) or not torch.equal(u[:2], -u[2:]):
        raise RuntimeError('iteration %d: replicas diverged (theta / validation-batch checksums differ across ranks: %s vs %s, '
                           'parameter bit checksums %s vs %s); seed every rank identically (see INTEGRATION.md)'
                           % (it + 1, lo.tolist(), hi.tolist(), (-u[2:]).tolist(), u[:2].tolist()))


def run_validation(forward_one_batch, model, vocab, valid_loader_list, it, args, history, loss_type, save_fn, criteria, stop_val,
                   best, count_stop, rank):
    """The in-loop validation of both trainers (transient_trainer.py:280-360, joint_trainer.py:306-380): eval mode, no autograd,
    `forward_one_batch` over every batch of every loader (AudioDataLoader layout: src, trg, percentages, src_lengths,
    trg_lengths), per-loader loss = mean over batches, CER = edits*100/chars, metrics dict + history, save every
    args.save_every iterations, best-model save and early-stop counter on the chosen criterion.
    save_fn(metrics, best_model) writes a checkpoint (rank 0 only).  -> (stop, best, count_stop)"""
    say = print if rank == 0 else (lambda *a, **k: None)
    say('')
    logging.info('VALID')
    smoothing = float(getattr(args, 'label_smoothing', 0.0) or 0.0)
    dev = model.flat_parameters.device
    model.eval()
    final_losses, final_cers = [], []
    try:
        with torch.no_grad():
            for ind, loader in enumerate(valid_loader_list):
                tot_loss, tot_cer, tot_char, nb = 0.0, 0, 0, 0
                for data in loader:
                    src, trg, pct, src_lengths, trg_lengths = data
                    if getattr(args, 'cuda', True):
                        src, trg = src.to(dev), trg.to(dev)
                    loss, cer, nchar = forward_one_batch(model, vocab, src, trg, pct, src_lengths, trg_lengths, smoothing, loss_type)
                    tot_cer += cer
                    tot_char += nchar
                    tot_loss += loss.item()
                    nb += 1
                final_losses.append(tot_loss / max(nb, 1))
                final_cers.append(tot_cer * 100 / max(tot_char, 1))
                msg = '(Iteration {}) VALID SET {} LOSS:{:.4f} CER:{:.2f}%'.format((it + 1), ind, final_losses[-1], final_cers[-1])
                say(msg)
                logging.info(msg)
    finally:
        model.train()
    if not final_losses:
        return False, best, count_stop
    metrics = {'avg_valid_loss': sum(final_losses) / len(final_losses), 'avg_valid_cer': sum(final_cers) / len(final_cers),
               'valid_loss': final_losses, 'valid_cer': final_cers, 'history': history}
    history.append(metrics)
    msg = '(Iteration {}) AVG VALID LOSS:{:.4f} AVG CER:{:.2f}%'.format((it + 1), metrics['avg_valid_loss'], metrics['avg_valid_cer'])
    say(msg)
    logging.info(msg)
    if rank == 0 and (it + 1) % args.save_every == 0:
        save_fn(metrics, False)
    cur = metrics['avg_valid_cer'] if criteria == 'cer' else metrics['avg_valid_loss']
    say('CRITERIA: CER' if criteria == 'cer' else 'CRITERIA: LOSS')
    if best > cur:
        count_stop, best = 0, cur
        if rank == 0:
            save_fn(metrics, True)
    else:
        count_stop += 1
        say('count_stop:', count_stop)
    if count_stop >= stop_val:
        logging.info('EARLY STOP')
        say('EARLY STOP\n')
        return True, best, count_stop
    return False, best, count_stop


MAX_CONSECUTIVE_FAILURES = 20


def _sample_takes_need(ds):
    """Does ds.sample accept the `need=` keyword (this package's datasets: load only what the rank uses)?  Read from the
    signature: catching TypeError around the call would also swallow errors raised INSIDE sample() -- after the index stream
    has advanced -- and the retry would draw a second time, taking this rank out of lock-step with the others."""
    import inspect
    try:
        params = inspect.signature(ds.sample).parameters
    except (TypeError, ValueError):
        return False
    return 'need' in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())


class TransientTrainer():
    def __init__(self):
        logging.info('Transient Trainer is initialized')
        # Runtime switches read from the environment (INTEGRATION.md "Environment"): MTL_BATCH_TASKS, MTL_RAGGED_FILL,
        # MTL_RAGGED_QUANTUM, MTL_PIPELINE_DEPTH.  Everything else below is a plain attribute (tests and bench.py set them).
        # command lists: the library calls of a task body are recorded once per (lane, shapes, scalars) and then replayed from C with
        # one ctypes call per task (include/mtl_hip.h "command lists"): host cost per pass 4.4 -> ~1 ms
        self.use_cmdlists = True
        self._cmdlists = {}
        # the local tasks of a meta-step as ONE task-batched pass per phase (training passes at theta0, validation passes at the
        # theta' stack) instead of one pass chain per task on concurrent lanes; MTL_BATCH_TASKS=0: the lanes
        self.batch_tasks = os.environ.get('MTL_BATCH_TASKS', '1') != '0'
        # tasks whose batches have different frame counts in one pass, stacked at the widest (False: one lane per task), as long as the
        # tasks' frames fill at least this share of the stack
        self.batch_ragged = True
        self.ragged_fill = float(os.environ.get('MTL_RAGGED_FILL', '0.4'))
        # ... and the stack's width (training and validation) rounded up to a multiple of this many frames: any width at or above the
        # widest task is exact (every task keeps its own border), and widths that repeat keep the buffer pool's allocations -- 11 GB
        # per width at the north-star size, otherwise re-allocated for every new widest utterance -- and the recorded command lists
        self.ragged_quantum = int(os.environ.get('MTL_RAGGED_QUANTUM', '64'))
        # the same rounding for a task that runs on a lane of its own (one task per rank; tasks too unequal to stack): 'auto' = from the
        # moment the lanes have seen two different widths (fixed-shape workloads are never padded), '1' always, '0' never
        self.pad_lanes = 'auto'
        self.label_quantum = 8      # ... and the decoder width to a multiple of this many positions
        self._lane_widths, self._lane_widths_vary = {}, False
        # train() enqueues iteration i + 1 before it resolves (logs) iteration i (enqueue_iteration).  MTL_PIPELINE_DEPTH: how many
        # iterations may be enqueued beyond the one being resolved (nothing the host needs to enqueue an iteration comes from the
        # device): 0 resolves at once, 1 hides the host's per-iteration work, 2 (default) also rides out host stalls of up to one
        # iteration's GPU time (shared hosts: measured enqueue times of 5 - 90 ms for the same iteration).  Read-back buffer sets:
        # depth + 1.
        self.pipeline_depth = max(0, int(os.environ.get('MTL_PIPELINE_DEPTH', '2')))
        self.pipeline = self.pipeline_depth > 0
        # batches handed over in host memory are uploaded on a separate stream, one iteration ahead of the kernels (task-batched passes)
        self.overlap_uploads = True
        self.pin_batches = True             # train(): batches drawn on the host are page-locked by the prefetch thread
        self.replica_check_every = 100      # several ranks: iterations between two bit-level divergence checks of the replicas
        # Guard of the two-piece fp16 ("h2") convolution arithmetic (csrc/mtl_h2.h: ONE power-of-two scale per tensor, so an element more
        # than ~2^17.5 below its tensor's maximum keeps fewer than 22 significand bits).  The reference's features are normalised per
        # utterance (utils/data_loader.py:84-94), which keeps every operand of the path in the full-precision regime -- this checks it
        # instead of assuming it: every `h2_check_every` iterations (the first included) the iteration also counts, for every h2 operand
        # of every pass (activations forward, gradients backward), the non-zero elements that keep fewer than 22 / fewer than 16 bits
        # (mtl_h2_census; ~5 % on that one iteration).  `h2_census` holds the latest fractions; when the share below 16 bits of any
        # operand exceeds `h2_limit`, `h2_guard` acts: 'x3' (default) moves the convolutions to the exact 3 x bf16 split for the rest of
        # the run, 'raise' stops, 'warn' only reports.  0 = never sample.
        self.h2_check_every, self.h2_limit, self.h2_guard = 100, 1e-3, 'x3'
        self.h2_census, self._h2_iter, self._h2_buf, self._census_on = None, 0, None, None
        self._turn = 0

    # ------------------------------------------------------------------ drop-in single-batch API
    def forward_one_batch(self, model, vocab, src, trg, src_percentages, src_lengths, trg_lengths, smoothing, loss_type,
                          verbose=False):
        """-> (loss tensor, total_cer, total_char); `loss.backward()` runs the HIP backward (transient_trainer.py:25-73).
        Same steps as the reference: model forward, calculate_metrics (label smoothing included), CER of the post-processed
        strings; the ~2*B*T per-element `int(x)` device syncs become one D2H copy of gold / hyp."""
        if loss_type != 'ce':
            raise NotImplementedError("only loss_type='ce' is on the accelerated path")
        pred, gold, hyp = model(src, src_lengths, trg, verbose=False)
        sizes = src_percentages.mul_(int(pred.size(1))).int()     # the reference scales the caller's tensor in place
        loss, _ = calculate_metrics(pred, gold, vocab.PAD_ID, input_lengths=sizes, target_lengths=trg_lengths,
                                    smoothing=smoothing, loss_type=loss_type)
        total_cer, total_char = cer_counts(vocab, gold.cpu(), hyp.cpu())
        if verbose:
            print('Total CER', total_cer)
            print('Total char', total_char)
        return loss, total_cer, total_char

    def get_lr(self, optimizer):
        for param_group in optimizer.param_groups:
            return param_group['lr']

    # ------------------------------------------------------------------ one meta-iteration on the device
    def meta_iteration(self, model, vocab, task_batches, val_batch, n_tasks, inner, outer, args, task_ids=None):
        """task_batches: this rank's [(inputs, input_sizes, percentages, targets, target_sizes)], val_batch: same 5-tuple.
        n_tasks is the GLOBAL task count (the 1/n of the validation loss).  Leaves G in model._G; returns the read-backs.
        With several ranks and chunking on, the schedules that hook the validation backward leave model._G ALREADY summed over
        the ranks and set self._G_reduced (reset here on entry); a direct caller must check that flag -- or call
        reduce_meta_gradient(), which does -- before posting a collective of its own.

        Tasks are independent given theta0, so they are dealt round-robin onto the model's task lanes (own stream, buffer
        arena, gradient / theta' / G buffers): one task's many small transformer kernels fill the CUs another task's
        convolutions leave idle.  Lane accumulators are summed in lane order, so G stays run-to-run deterministic."""
        dev = model.flat_parameters.device
        theta0 = model.flat_parameters
        self._G_reduced = False
        smoothing = float(getattr(args, 'label_smoothing', 0.0) or 0.0)
        hooked = any(e.prof is not None or e.forward_hook is not None for e in model.engines)
        use_cmdlists = self.use_cmdlists and not hooked
        self.last_schedule = 'lanes'                         # (diagnostics / tests: which schedule the last iteration took)
        if self._can_batch(model, task_batches, val_batch):
            frames = [int(tb[0].shape[3]) for tb in task_batches]
            self.last_schedule = 'batched' if min(frames) == max(frames) else 'batched-ragged'
            return self._batched_iteration(model, task_batches, val_batch, n_tasks, inner, args, smoothing, use_cmdlists)
        n_lanes = min(model.n_lanes, max(len(task_batches), 1))
        if len(task_batches) > n_lanes:                      # several rounds: equal rounds (8 tasks on 6 lanes measured slower than on 3)
            rounds = -(-len(task_batches) // n_lanes)
            n_lanes = -(-len(task_batches) // rounds)
        bufs = self._lane_buffers(model, n_lanes)
        main = torch.cuda.current_stream(dev)
        vx = val_batch[0].to(dev, non_blocking=True)
        # widths that change from batch to batch (manifest-fed) would be enqueued call by call for ever: rounded up to a quantum they
        # repeat (recorded lists replay, the pool keeps its buffers); the batch keeps its own border and encoder length (prepare(frames))
        varies = self._widths_vary([('lane', i, int(tb[0].shape[3])) for i, tb in enumerate(task_batches)] + [('lane', 'val', int(vx.shape[3]))])
        q = self.ragged_quantum if varies else 0
        most = 4 * model.engines[0].hp.src_max_len

        def widened(x, eng_, name):
            """-> (x or its copy in a zero-filled buffer of the rounded width, own frame count or None)"""
            T_own = int(x.shape[3])
            Tq = max(min(round_width(T_own, q), most), T_own) if q > 1 else T_own
            if Tq == T_own:
                return x, None
            xp = eng_.buf(name, tuple(x.shape[:3]) + (Tq,))
            xp.zero_()
            xp[:, :, :, :T_own].copy_(x, non_blocking=True)
            return xp, T_own
        vx, v_own = widened(vx, model.engines[0], 'lane.x_va')
        ready = torch.cuda.Event()
        ready.record(main)
        reads = [None] * len(task_batches)
        streams = [model.lane_streams[lane] if n_lanes > 1 else main for lane in range(n_lanes)]
        for lane in range(n_lanes):
            with torch.cuda.stream(streams[lane]):
                streams[lane].wait_event(ready)
                bufs[lane][2].zero_()
        for idx, (tx, tsz, _tp, ty, _tl) in enumerate(task_batches):      # enqueue task by task, alternating lanes
            lane = idx % n_lanes
            eng = model.engines[lane]
            with torch.cuda.stream(streams[lane]):
                tx, t_own = widened(tx.to(dev, non_blocking=True), eng, 'lane.x_tr')
                lw = (lambda y_: _label_width([y_], self.label_quantum, eng.hp.tgt_max_len)) if (q > 1 and self.label_quantum > 1) else (lambda y_: None)
                eng.census = self._census_on[lane:lane + 1] if self._census_on is not None else None
                m_tr = eng.prepare(tsz, ty, tx.shape[0], tx.shape[3], slot=0, frames=t_own, width=lw(ty))   # host ints -> static device buffers
                m_va = eng.prepare(val_batch[1], val_batch[3], vx.shape[0], vx.shape[3], slot=1, frames=v_own, width=lw(val_batch[3]))
                slots = self._slots(model, lane, m_tr, m_va)
                key = (lane, tuple(tx.shape), tuple(vx.shape), t_own is not None, v_own is not None, m_tr['Td'], m_va['Td'], n_tasks, bool(args.clip),
                       float(args.max_norm), smoothing, float(inner.param_groups[0]['lr']), theta0.data_ptr(), eng.dropout_p,
                       tuple(b.data_ptr() for b in bufs[lane]), streams[lane].cuda_stream, eng.use_side_stream, eng.census is not None)
                # a rank that holds ONE task (8 tasks on 8 GPUs): G = its g, handed to the all-reduce group by group under the backward
                chunk = None
                if n_lanes == 1 and len(task_batches) == 1:
                    g_, G_ = bufs[lane][0], bufs[lane][2]
                    chunk = self._chunk_hook(model, eng, lambda lo, n, st, g_=g_, G_=G_: check(
                        _lib.lib().mtl_axpy(st, G_.data_ptr() + 4 * lo, g_.data_ptr() + 4 * lo, 1.0, n), 'mtl_axpy'))
                    key = key + ('chunked',) if chunk is not None else key
                body = lambda xa, xb: self._task_body(model, lane, bufs[lane], theta0, xa, m_tr, xb, m_va, n_tasks, inner, args,
                                                      smoothing, slots, chunk)
                if use_cmdlists:
                    self._run_recorded(key, eng, tx, vx, body, on_break=chunk)
                else:
                    body(tx, vx)
                reads[idx] = (_Readback(dict(gold_host=m_tr['gold_host'], hyp=slots['hyp_tr'], loss=slots['loss_tr']), (idx, 0, self._turn), self._turns()),
                              _Readback(dict(gold_host=m_va['gold_host'], hyp=slots['hyp_va'], loss=slots['loss_va']), (idx, 1, self._turn), self._turns()))
        if streams[0] is not main:
            for lane in range(n_lanes):
                done = torch.cuda.Event()
                done.record(streams[lane])
                main.wait_event(done)
        Gm = model._G
        self._chunk_join(dev)
        if n_lanes == 1:
            pass                                              # lane 0 accumulated straight into model._G
        else:
            Gm.copy_(bufs[0][2])
            for lane in range(1, n_lanes):
                model._axpy(Gm, bufs[lane][2], 1.0)
        return reads

    # ------------------------------------------------------------------ task-batched passes
    def _widths_vary(self, sightings):
        """sightings: [(schedule, slot, frames)] of this iteration's batches.  -> whether batches are to be widened to repeating widths:
        pad_lanes '1' always, '0' never, 'auto' from the moment any slot (a task's position, the validation batch) has brought two
        different widths -- a fixed-shape workload is never padded, a manifest-fed one from its second iteration on."""
        if self.pad_lanes != 'auto':
            return self.pad_lanes == '1'
        if not self._lane_widths_vary:
            for sched, slot, frames in sightings:
                if self._lane_widths.setdefault((sched, slot), frames) != frames:
                    self._lane_widths_vary = True
                    break
        return self._lane_widths_vary

    def _can_batch(self, model, task_batches, val_batch):
        """All local tasks in one pass per phase: needs >= 2 tasks with identical batch shapes (samples x frames; label widths may
        differ) and the fused attention kernel."""
        if not self.batch_tasks or len(task_batches) < 2:
            return False
        eng = model.engines[0]
        if not eng.fused_attn:
            return False
        if val_batch[0].dim() != 4 or any(tb[0].dim() != 4 for tb in task_batches):
            return False
        shape = tuple(task_batches[0][0].shape)
        if all(tuple(tb[0].shape) == shape for tb in task_batches):
            return True
        # manifest-fed batches: every task's batch is padded to its OWN longest utterance (data.py:77), so the frame counts differ.
        # They are stacked at the widest (engine.prepare_tasks(frames=...) keeps each task's own image border and encoder length) as
        # long as the padding does not outweigh what one pass for all tasks saves over a lane per task (which, on shapes that never
        # repeat, is enqueued call by call: ~88 ms of host time per 8-task north-star step against ~55 ms of kernels).
        if not self.batch_ragged:
            return False
        if any(tuple(tb[0].shape[:3]) != shape[:3] for tb in task_batches):
            return False
        frames = [int(tb[0].shape[3]) for tb in task_batches]
        return min(frames) >= 4 and sum(frames) >= self.ragged_fill * len(frames) * max(frames)

    def _batched_iteration(self, model, task_batches, val_batch, n_tasks, inner, args, smoothing, use_cmdlists):
        """trainer/asr/transient_trainer.py:178-237 for all local tasks at once.  The tasks of a meta-step are independent given
        theta0, so their training passes run as ONE pass over nt x k_train samples (shared parameters; per-task losses and a
        (nt, P) stack of per-task gradients), the nt inner steps as one kernel into a (nt, P) stack of theta', and their
        validation passes as ONE pass in which task t reads its own theta'_t (task = outermost batch index of every product).
        G = sum_t (g_tr,t + g_val,t / n) in task order.  The ~200 latency-bound transformer launches of a pass run once per phase
        instead of once per task."""
        dev = model.flat_parameters.device
        theta0 = model.flat_parameters
        eng = model.engines[0]
        nt = len(task_batches)
        total = model._layout.total
        B, _, F, T = task_batches[0][0].shape
        frames = [int(tb[0].shape[3]) for tb in task_batches]
        T = max(frames)
        ragged = min(frames) != T
        vx_in = val_batch[0]
        Tv = Tv_own = int(vx_in.shape[3])
        # (uniform across the tasks but changing from step to step -- a loader that pads all tasks alike -- is rounded as well, from the
        # second width on: `_widths_vary`)
        varies = self._widths_vary([('batched', 'train', T), ('batched', 'val', Tv)]) and self.batch_ragged
        if (ragged or varies) and self.ragged_quantum > 1:
            most = 4 * eng.hp.src_max_len                                    # (the positional table bounds the encoder length)
            T = max(min(round_width(T, self.ragged_quantum), most), T)
            Tv = max(min(round_width(Tv, self.ragged_quantum), most), Tv)
        key_b = (id(theta0), nt)
        if getattr(self, '_stack_key', None) != key_b:
            self._stack = (torch.zeros(nt * total, dtype=torch.float32, device=dev), torch.empty(nt * total, dtype=torch.float32, device=dev))
            self._stack_key = key_b
        g, theta1 = self._stack
        # The passes read their inputs from two STATIC buffers (recorded command lists hold their addresses, and addresses derived
        # from them per task).  Inputs that arrive in HOST memory (the reference uploads every batch inside its timed span,
        # transient_trainer.py:182-184,210-212) go up on their own stream into one of two landing sets, so the copy engine moves
        # iteration i + 1's batches under iteration i's kernels (the host enqueues ahead); the main stream waits for the set's `ready`
        # event and moves it into the static buffers with one device copy each (~20 us).  A landing set is rewritten only after that
        # device copy has run (`_xfree`).  Device-resident inputs are copied straight into the static buffers.
        main = torch.cuda.current_stream(dev)
        Xtr = eng.buf('tb.x_tr', (nt * B, 1, F, T))
        Xva = eng.buf('tb.x_va', tuple(vx_in.shape[:3]) + (Tv,))
        _trace.mark('setup')
        on_host = not vx_in.is_cuda or any(not tb[0].is_cuda for tb in task_batches)
        own_tr = min(frames) != T                            # some task is narrower than the stack: borders / lengths of their own
        wide = own_tr or Tv != Tv_own
        nv = vx_in.numel()

        def place(src_tr, src_va):
            """every task's frames at the front of its slab, zeros behind them (the border its own, narrower image ends in);
            src_tr(t) / src_va: device tensors of the batches' own shapes"""
            Xtr.zero_()
            X5 = Xtr.view(nt, B, 1, F, T)
            for t in range(nt):
                X5[t, :, :, :, :frames[t]].copy_(src_tr(t), non_blocking=True)
            if Tv != Tv_own:
                Xva.zero_()
            Xva[:, :, :, :Tv_own].copy_(src_va, non_blocking=True)
        if on_host and self.overlap_uploads:
            xs = self._xset = (getattr(self, '_xset', 0) + 1) % 2
            if getattr(self, '_upload_stream', None) is None:
                self._upload_stream, self._xfree = torch.cuda.Stream(dev), {}
            Ltr = eng.buf('tb.land_tr.%d' % xs, (nt * B, 1, F, T))
            Lva = eng.buf('tb.land_va.%d' % xs, tuple(Xva.shape))
            L2 = Ltr.view(nt, -1)                            # a task's batch lands CONTIGUOUSLY at the front of its slab (its own shape)
            up = self._upload_stream
            with torch.cuda.stream(up):
                free = self._xfree.get((xs, Ltr.data_ptr(), Lva.data_ptr()))
                if free is not None:
                    up.wait_event(free)
                else:
                    up.wait_stream(main)                 # first use of this set: behind whatever produced / last used the memory
                    Ltr.record_stream(up)                # (the pool may hand the block back to torch's allocator one day)
                    Lva.record_stream(up)
                if vx_in.is_cuda or any(tb[0].is_cuda for tb in task_batches):
                    # mixed residency (e.g. a cached device-resident validation batch beside host training batches): a device tensor
                    # may have been produced on the main stream in this very iteration
                    up.wait_stream(main)
                for t, (tx, _tsz, _tp, _ty, _tl) in enumerate(task_batches):
                    L2[t, :tx.numel()].copy_(tx.reshape(-1), non_blocking=True)
                Lva.view(-1)[:nv].copy_(vx_in.reshape(-1), non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(up)
            main.wait_event(ready)
            if wide:
                place(lambda t: L2[t, :B * F * frames[t]].view(B, 1, F, frames[t]), Lva.view(-1)[:nv].view(vx_in.shape))
            else:
                Xtr.copy_(Ltr, non_blocking=True)
                Xva.copy_(Lva, non_blocking=True)
            free = torch.cuda.Event()
            free.record(main)
            self._xfree = {(xs, Ltr.data_ptr(), Lva.data_ptr()): free, **{k_: v_ for k_, v_ in self._xfree.items() if k_[0] != xs}}
        elif wide:
            place(lambda t: task_batches[t][0] if task_batches[t][0].is_cuda else task_batches[t][0].to(dev, non_blocking=True),
                  vx_in if vx_in.is_cuda else vx_in.to(dev, non_blocking=True))
        else:
            for t, (tx, _tsz, _tp, _ty, _tl) in enumerate(task_batches):
                Xtr[t * B:(t + 1) * B].copy_(tx, non_blocking=True)
            Xva.copy_(vx_in, non_blocking=True)
        _trace.mark('input_copies')
        wq = (ragged or varies) and self.ragged_quantum > 1 and self.label_quantum > 1      # label widths repeat like the frame counts
        lw = (lambda ys: _label_width(ys, self.label_quantum, eng.hp.tgt_max_len)) if wq else (lambda ys: None)
        m_tr = eng.prepare_tasks([(tsz, ty) for (_tx, tsz, _tp, ty, _tl) in task_batches], B, T, slot=0, frames=frames if own_tr else None,
                                 width=lw([tb[3] for tb in task_batches]))
        m_va = eng.prepare_tasks([(val_batch[1], val_batch[3])] * nt, vx_in.shape[0], Tv, slot=1, frames=[Tv_own] * nt if Tv != Tv_own else None,
                                 width=lw([val_batch[3]]))
        _trace.mark('prepare_tasks')
        Bv = vx_in.shape[0]
        slots = dict(hyp_tr=eng.buf('slot.hyp_tr', (nt * B, m_tr['Td']), torch.int64), loss_tr=eng.buf('slot.loss_tr', (nt,)),
                     hyp_va=eng.buf('slot.hyp_va', (nt * Bv, m_va['Td']), torch.int64), loss_va=eng.buf('slot.loss_va', (nt,)))
        G = model._G
        lr = float(inner.param_groups[0]['lr'])
        # several ranks: G's three parameter groups are summed over the local tasks and all-reduced one by one, each as soon as the
        # validation backward has produced it (decoder first, conv last) -- on a communication stream, under the rest of the backward
        chunk = self._chunk_hook(model, eng, lambda lo, n, st: check(
            _lib.lib().mtl_sum_tasks_strided(st, G.data_ptr() + 4 * lo, g.data_ptr() + 4 * lo, n, nt, total, 0), 'mtl_sum_tasks_strided'))

        eng.census = self._census_on[:nt] if self._census_on is not None else None

        def body(_xa=None, _xb=None):
            eng.zero_(g)                                                         # inner_opt.zero_grad()   (:198), every task
            eng.forward_device(theta0, Xtr, m_tr, smoothing, hyp_out=slots['hyp_tr'], loss_out=slots['loss_tr'])      # (:188)
            eng.backward(g, 1.0, sG=total)                                       # tr_loss.backward()      (:199)
            if args.clip:
                for t in range(nt):
                    clip_flat_grad_(model, g[t * total:(t + 1) * total], args.max_norm, lane=0)      # (:205-206)
            check(eng.lib.mtl_sgd_theta_prime_tasks(eng.stream, theta0.data_ptr(), g.data_ptr(), lr, theta1.data_ptr(), total, nt),
                  'mtl_sgd_theta_prime_tasks')                                   # inner_opt.step()        (:207)
            eng.forward_device(theta1, Xva, m_va, smoothing, hyp_out=slots['hyp_va'], loss_out=slots['loss_va'], sP=total)   # (:215)
            eng.slice_hook = chunk
            try:
                eng.backward(g, 1.0 / n_tasks, sG=total)                         # (val_loss/n).backward(): g_t += g_val,t/n (Q1)
            finally:
                eng.slice_hook = None
            if chunk is None:
                check(eng.lib.mtl_sum_tasks(eng.stream, G.data_ptr(), g.data_ptr(), total, nt, 0), 'mtl_sum_tasks')    # add_copy_grad() (:229)

        key = ('batched', nt, (B, F, T), own_tr, tuple(Xva.shape), Tv != Tv_own, m_tr['Td'], m_va['Td'], n_tasks, bool(args.clip), float(args.max_norm),
               smoothing, lr, theta0.data_ptr(), eng.dropout_p, g.data_ptr(), theta1.data_ptr(), G.data_ptr(),
               torch.cuda.current_stream(dev).cuda_stream, eng.use_side_stream, chunk is not None, eng.census is not None)
        _trace.mark('keys')
        if use_cmdlists:
            self._run_recorded(key, eng, Xtr, Xva, body, on_break=chunk)
        else:
            body()
        _trace.mark('launches')
        self._chunk_join(dev)
        hyp_tr = _pinned(('tb.hyp', 0, self._turn), slots['hyp_tr'].shape, torch.int64, self._turns())
        hyp_va = _pinned(('tb.hyp', 1, self._turn), slots['hyp_va'].shape, torch.int64, self._turns())
        loss_tr = _pinned(('tb.loss', 0, self._turn), (nt,), torch.float32, self._turns())
        loss_va = _pinned(('tb.loss', 1, self._turn), (nt,), torch.float32, self._turns())
        for dst, src in ((hyp_tr, slots['hyp_tr']), (hyp_va, slots['hyp_va']), (loss_tr, slots['loss_tr']), (loss_va, slots['loss_va'])):
            dst.copy_(src, non_blocking=True)
        _trace.mark('readback_copies')
        return [(_TaskRead(m_tr['gold_hosts'][t], hyp_tr[t * B:(t + 1) * B], loss_tr[t:t + 1]),
                 _TaskRead(m_va['gold_hosts'][t], hyp_va[t * Bv:(t + 1) * Bv], loss_va[t:t + 1])) for t in range(nt)]

    def _chunk_hook(self, model, eng, accumulate_slice):
        """-> callable(tag) for PassEngine.slice_hook (None when the meta-gradient is not all-reduced in chunks).  At the point where
        the validation backward has enqueued the last kernel of a parameter group, the hook -- on the communication stream, behind the
        engine's main AND side stream -- forms that group's slice of G from the local gradients (accumulate_slice(first, count, raw
        stream)) and starts its all-reduce (dist.ChunkedAllReduce); _chunk_join() makes the main stream wait for all three before the
        outer step.  Fixed slice order on every rank: decoder, encoder, conv (the order the backward finishes them in)."""
        if not mdist.chunked_on():
            self._chunks = None
            return None
        dev = model.flat_parameters.device
        if getattr(self, '_comm_stream', None) is None:
            self._comm_stream = torch.cuda.Stream(dev)
        bounds = eng.slice_bounds()
        if sorted(bounds) != ['conv', 'decoder', 'encoder']:
            raise RuntimeError('unexpected parameter groups %s' % sorted(bounds))
        self._chunks = mdist.ChunkedAllReduce()
        G, comm = model._G, self._comm_stream

        def hook(tag):
            lo, hi = bounds[tag]
            comm.wait_stream(torch.cuda.current_stream(dev))
            if eng.side is not None:
                comm.wait_stream(eng.side)
            with torch.cuda.stream(comm):
                accumulate_slice(lo, hi - lo, comm.cuda_stream)
                self._chunks.issue(G[lo:hi])
        return hook

    def _chunk_join(self, dev):
        """the main stream waits for the chunked collectives (and the slice sums in front of them); marks G as already reduced"""
        if getattr(self, '_chunks', None) is None:
            return
        with torch.cuda.stream(self._comm_stream):
            self._chunks.wait()                        # comm stream waits for the collectives' stream
        torch.cuda.current_stream(dev).wait_stream(self._comm_stream)
        self._chunks = None
        self._G_reduced = True

    def _task_body(self, model, lane, bufs, theta0, x_tr, m_tr, x_va, m_va, n_tasks, inner, args, smoothing, slots, chunk=None):
        """Kernels of ONE task on the current stream (eager, or recorded into a command list): train pass at theta0, fused inner
        SGD into theta', validation pass at theta', accumulation into the lane's copy_grad buffer."""
        eng = model.engines[lane]
        g, theta1, G = bufs
        eng.zero_(g)                                                     # inner_opt.zero_grad()   (:198)
        eng.forward_device(theta0, x_tr, m_tr, smoothing, hyp_out=slots['hyp_tr'], loss_out=slots['loss_tr'])   # meta-train forward (:188)
        eng.backward(g, 1.0)                                             # tr_loss.backward()      (:199)
        if args.clip:
            clip_flat_grad_(model, g, args.max_norm, lane=lane)          # (:205-206)
        eng.sgd_theta_prime(theta0, g, inner.param_groups[0]['lr'], theta1)     # inner_opt.step() (:207)
        eng.forward_device(theta1, x_va, m_va, smoothing, hyp_out=slots['hyp_va'], loss_out=slots['loss_va'])   # meta-validation forward (:215)
        eng.slice_hook = chunk                                           # (several ranks: G's groups leave for their all-reduce as they finish)
        try:
            eng.backward(g, 1.0 / n_tasks)                               # (val_loss/n).backward(): g += g_val/n (Q1)
        finally:
            eng.slice_hook = None
        if chunk is None:
            eng.axpy_(G, g, 1.0)                                         # add_copy_grad()         (:229)

    def _turns(self):
        """read-back buffer sets: one per iteration the host may have in flight (pipeline depth) + the one being resolved"""
        return max(getattr(self, 'pipeline_depth', 1), 1) + 1

    def _slots(self, model, lane, m_tr, m_va):
        eng = model.engines[lane]
        return dict(hyp_tr=eng.buf('slot.hyp_tr', (m_tr['B'], m_tr['Td']), torch.int64), loss_tr=eng.buf('slot.loss_tr', (1,)),
                    hyp_va=eng.buf('slot.hyp_va', (m_va['B'], m_va['Td']), torch.int64), loss_va=eng.buf('slot.loss_va', (1,)))

    def _run_recorded(self, key, eng, tx, vx, body, on_break=None):
        """Command-list execution of a task body.  First sighting of a key: plain eager run (it also brings every arena buffer to
        its final size); second sighting: eager run through a Recorder that logs the calls; afterwards: the two input pointers
        are re-pointed to this task's batches and the recorded calls are replayed by mtl_cmdlist_run.
        eng: one engine, or the list of engines of a body that spans several lanes (ONE list holds all their calls, each with its
        own stream); tx = vx = None: the body reads static input buffers, nothing is re-pointed."""
        engs = list(eng) if isinstance(eng, (list, tuple)) else [eng]
        epoch_now = lambda: tuple(e.scratch_epoch for e in engs)
        ent = self._cmdlists.get(key)
        if ent is None:
            while len(self._cmdlists) >= 64:              # least recently used key goes (variable-length data: keys rarely repeat)
                self._cmdlists.pop(next(iter(self._cmdlists)))
            self._cmdlists[key] = 'warm'
            return body(tx, vx)
        self._cmdlists[key] = self._cmdlists.pop(key)     # most recently used: to the end
        if ent != 'warm' and ent['epoch'] != epoch_now():
            ent = 'warm'                                  # an engine's scratch buffer moved since the recording
        if ent == 'warm':
            if tx is not None and tx.data_ptr() == vx.data_ptr():
                return body(tx, vx)                       # the two inputs must be distinguishable by address to be re-pointed
            cl = _lib.CommandList()
            reals, epoch = [e.lib for e in engs], epoch_now()
            recorder = _lib.Recorder(reals[0], cl)
            for e in engs:
                e.lib = recorder
            try:
                body(tx, vx)
            finally:
                for e, real in zip(engs, reals):
                    e.lib = real
            if epoch_now() == epoch:                      # (a buffer that moved during the run would leave stale addresses)
                self._cmdlists[key] = dict(cl=cl.finish(), x_tr=tx.data_ptr() if tx is not None else None,
                                           x_va=vx.data_ptr() if vx is not None else None, epoch=epoch)
            return
        cl = ent['cl']
        if tx is not None and (ent['x_tr'] != tx.data_ptr() or ent['x_va'] != vx.data_ptr()):
            tmp = 8                                       # two-step re-pointing through a dummy value: the batches may swap addresses
            cl.repoint(ent['x_tr'], tmp)
            cl.repoint(ent['x_va'], vx.data_ptr())
            cl.repoint(tmp, tx.data_ptr())
            ent['x_tr'], ent['x_va'] = tx.data_ptr(), vx.data_ptr()
        cl.run(on_break)

    def _lane_buffers(self, model, n_lanes):
        """(grad, theta', G) per lane; lane 0 uses the model's own flat_grad / copy_grad buffers when it is the only lane."""
        key = (id(model.flat_parameters), n_lanes)
        if getattr(self, '_lane_key', None) != key:
            th = model.flat_parameters
            if n_lanes == 1:
                self._lanes = [(model.flat_grad, torch.empty_like(th), model._G)]
            else:
                self._lanes = [(torch.zeros_like(th), torch.empty_like(th), torch.zeros_like(th)) for _ in range(n_lanes)]
            self._lane_key = key
        return self._lanes

    def enqueue_iteration(self, model, vocab, task_batches, val_data, n_tasks, inner_opt, outer_opt, args):
        """Enqueues one meta-iteration (transient_trainer.py:152-264: local tasks, ONE all-reduce of G, Adam) WITHOUT waiting for
        it and returns a PendingIteration whose result() resolves the loss / label read-backs.  Nothing the host needs to enqueue
        iteration i + 1 comes from the device, so train() (and bench.py) enqueue it before resolving iteration i: the host's
        per-iteration work (batch preparation, CER strings, logging: 1.3 - 2.9 ms, all of it GPU idle time when a rank holds a
        single 12 ms task) runs under the next iteration's kernels.  Read-back buffers rotate through pipeline_depth + 1 sets (`_turn`)."""
        dev = model.flat_parameters.device
        t_host = time.perf_counter()
        _trace.begin()
        self._turn = (self._turn + 1) % self._turns()
        outer_opt.zero_grad()
        _trace.mark('zero_grad')
        self._G_reduced = False
        census = self._census_begin(model, len(task_batches))
        try:
            reads = self.meta_iteration(model, vocab, task_batches, val_data, n_tasks, inner_opt, outer_opt, args)
        finally:
            self._census_on = None
            for e in model.engines:
                e.census = None
        _trace.mark('meta_iteration_rest')
        G = model._G
        if not self._G_reduced:                                  # (already summed over the ranks group by group, under the backward: _chunk_hook)
            self.reduce_meta_gradient(model)
        if args.clip:
            clip_flat_grad_(model, G, args.max_norm)             # (:253-254) on the summed meta-gradient
        outer_opt.step(G)                                        # from_copy_grad() + outer_opt.step()  (:248-255)
        _trace.mark('allreduce_adam')
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        _trace.mark('done_event')
        _trace.end()
        pend = None
        if census is not None:
            host = _pinned(('h2census', self._turn), census.shape, torch.int64, self._turns())
            host.copy_(census, non_blocking=True)
            done.record(torch.cuda.current_stream(dev))
            pend = (host, lambda counters, model=model: self._census_report(model, counters))
        self.host_enqueue_s = time.perf_counter() - t_host       # host side of the iteration (diagnostics: bench.py reports it)
        return PendingIteration(reads, done, vocab, dev, census=pend)

    def _census_begin(self, model, n_local):
        """-> the zeroed counter buffer when this iteration samples the h2 census (see __init__), else None.  Row t belongs to local
        task t of a task-batched pass, or to lane t of the lane schedule."""
        self._census_on = None
        engs = getattr(model, 'engines', None)
        if not engs or self.h2_check_every <= 0 or not all(e.conv_h2 for e in engs) or any(e.prof is not None for e in engs):
            return None
        self._h2_iter += 1
        if (self._h2_iter - 1) % self.h2_check_every:
            return None
        rows = max(len(engs), n_local, 1)
        dev = model.flat_parameters.device
        if self._h2_buf is None or self._h2_buf.shape[0] < rows or self._h2_buf.device != dev:
            self._h2_buf = torch.zeros(rows, CENSUS_SLOTS, 4, dtype=torch.int64, device=dev)
        engs[0].zero_(self._h2_buf)
        self._census_on = self._h2_buf
        return self._h2_buf

    def _census_report(self, model, counters):
        """the read-back counters of a census iteration -> self.h2_census {operand: (share below 22 bits, share below 16 bits)} of
        the non-zero elements; acts per `h2_guard` when an operand's share below 16 bits exceeds `h2_limit`"""
        c = counters.numpy().reshape(-1, CENSUS_SLOTS, 4).sum(0)
        if mdist.world_size() > 1:
            # every rank samples on the same iterations: the counters are summed over the ranks (host collective), so that all of them
            # see the same shares and take the same decision
            flat = mdist.allreduce_scalars([float(v) for v in c.reshape(-1)], counters.device)
            c = torch.tensor(flat, dtype=torch.float64).view(CENSUS_SLOTS, 4).numpy()
        self.h2_census = {CENSUS_NAMES[i]: (float(c[i, 1]) / max(int(c[i, 0]), 1), float(c[i, 2]) / max(int(c[i, 0]), 1))
                          for i in sorted(CENSUS_NAMES) if int(c[i, 0]) > 0}
        worst = max(self.h2_census, key=lambda k_: self.h2_census[k_][1], default=None)
        msg = 'H2 CENSUS share of non-zero elements below 22 / 16 significand bits: ' + ', '.join(
            '%s %.2e / %.2e' % (k_, v[0], v[1]) for k_, v in self.h2_census.items())
        logging.info(msg)
        if worst is None or self.h2_census[worst][1] <= self.h2_limit:
            return
        what = ('%.2e of the non-zero elements of "%s" keep fewer than 16 significand bits under one scale per tensor (limit %.1e): '
                'the batch mixes magnitudes more than 2^22 apart (features not normalised per utterance?)' %
                (self.h2_census[worst][1], worst, self.h2_limit))
        if self.h2_guard == 'raise':
            raise RuntimeError('h2 range guard: ' + what)
        if self.h2_guard == 'x3' and all(e.conv_h2 for e in model.engines):
            for e in model.engines:
                e.conv_mode, e.conv_x3, e.conv_h2 = 'x3', True, False
            self._cmdlists.clear()                           # (the recorded lists hold the h2 entry points)
            what += '; the 3x3 convolutions and the input Linear run on the exact 3 x bf16 split from the next enqueued iteration on'
        logging.warning('h2 range guard: ' + what)
        print('WARNING: h2 range guard: ' + what, flush=True)

    def reduce_meta_gradient(self, model):
        """The collective of the path for an iteration whose backward did not hand G's groups over as it went (several differently
        shaped local tasks on lanes, no local task at all, the split-task schedule, hooks off).  Which schedule a rank takes depends
        on ITS data, so the sequence of collectives must not: with chunking on, every rank always posts the same three all-reduces
        (decoder, encoder, conv slices, the order _chunk_hook uses) -- here back to back on the current stream; with
        MTL_CHUNKED_ALLREDUCE=0 every rank posts the single all-reduce of the whole flat buffer."""
        G = model._G
        if mdist.chunked_on():
            bounds = model._layout.group_bounds()
            for tag in mdist.SLICE_ORDER:
                lo, hi = bounds[tag]
                mdist.allreduce_sum_(G[lo:hi])
        else:
            mdist.allreduce_sum_(G)
        self._G_reduced = True

    def run_iteration(self, model, vocab, task_batches, val_data, n_tasks, inner_opt, outer_opt, args):
        """The timed body of one meta-iteration, resolved: -> (sum val loss, CER edits, chars), global."""
        return self.enqueue_iteration(model, vocab, task_batches, val_data, n_tasks, inner_opt, outer_opt, args).result()

    def train(self, model, vocab, train_data_list, valid_loader_list, loss_type, start_it, num_it, args, inner_opt=None,
              outer_opt=None, evaluate_every=1000, window_size=100, last_summary_every=1000, last_metrics=None, early_stop=10,
              cpu_state_dict=False, is_copy_grad=False):
        if loss_type != 'ce':
            raise NotImplementedError("only loss_type='ce' is on the accelerated path")
        if not is_copy_grad:
            raise NotImplementedError('the accelerated path is the --copy-grad loop (is_copy_grad=True)')
        history = []
        best_valid_val = 1000000000
        early_stop_criteria, early_stop_val = early_stop.split(',')[0], int(early_stop.split(',')[1])
        count_stop = 0
        hostenv.bound_torch_threads()        # torch's CPU pool <= the CPUs this container may use (hostenv.py: quota throttling stalls the loop)
        logging.info('name ' + args.name)
        total_time = 0
        logging.info('TRAIN')
        rank, world = mdist.rank(), mdist.world_size()
        if rank == 0:
            print('TRAIN')
        model.train()

        if inner_opt is None:
            inner_opt = FlatSGD(model, args.lr)
        elif not isinstance(inner_opt, FlatSGD):
            inner_opt = FlatSGD.from_torch(model, inner_opt)
        if outer_opt is None:
            outer_opt = FlatAdam(model, args.meta_lr)
        elif not isinstance(outer_opt, FlatAdam):
            outer_opt = FlatAdam.from_torch(model, outer_opt)
        self.inner_opt, self.outer_opt = inner_opt, outer_opt
        model.zero_copy_grad()

        last_sum_loss, last_sum_cer, last_sum_char = deque(maxlen=window_size), deque(maxlen=window_size), deque(maxlen=window_size)
        k_train, k_valid = args.k_train, args.k_valid
        n_tasks = len(train_data_list)
        train_data_buffer = [[] for _ in range(n_tasks)]
        my_tasks = mdist.shard_tasks(n_tasks, rank, world)

        takes_need = [_sample_takes_need(ds) for ds in train_data_list]      # decided ONCE from the signature (never by catching)
        pin_batches = bool(getattr(args, 'cuda', True)) and torch.cuda.is_available() and self.pin_batches

        def fetch_train_batch(buf):
            # every rank DRAWS every task (the index streams stay in lock-step), but only what this rank uses is loaded and
            # featurised: its own tasks' training batches and the last task's validation batch, which all tasks share (:168)
            for manifest_id in range(n_tasks):
                need = (manifest_id in my_tasks, manifest_id == n_tasks - 1)
                ds = train_data_list[manifest_id]
                if takes_need[manifest_id]:
                    item = ds.sample(k_train, k_valid, manifest_id, need=need)
                else:                                                # a duck-typed dataset with the reference's 3-argument sample()
                    item = ds.sample(k_train, k_valid, manifest_id)
                if pin_batches:
                    # the feature tensors this rank will upload are page-locked HERE, on the prefetch thread: a copy from pageable
                    # memory holds the enqueueing thread for its whole duration (transient_trainer.py:182-184,210-212 upload inside
                    # the timed span); torch's caching host allocator re-uses the blocks from iteration to iteration
                    item = tuple(((part[0].pin_memory(),) + tuple(part[1:])) if (want and torch.is_tensor(part[0]) and not part[0].is_cuda
                                                                                 and not part[0].is_pinned()) else part
                                 for part, want in zip(item, need))
                buf[manifest_id].insert(0, item)

        prefetch = threading.Thread(target=fetch_train_batch, args=(train_data_buffer,))
        prefetch.start()
        dev = model.flat_parameters.device
        sync_replicas_from_rank0(model, [outer_opt])
        check_every = int(self.replica_check_every)
        it = start_it
        failures = 0
        pending = deque()                                 # enqueued, not yet resolved: (it, step), oldest first
        clock = [time.time()]
        self.loss_trace = []                              # (validation loss / n, CER edits, characters) of every resolved iteration

        def resolve(it_, step):
            nonlocal total_time
            total_loss, total_cer, total_char = step.result()
            last_sum_cer.append(total_cer)
            last_sum_char.append(total_char)
            last_sum_loss.append(total_loss / n_tasks)
            self.loss_trace.append((total_loss / n_tasks, total_cer, total_char))
            now = time.time()
            diff_time, clock[0] = now - clock[0], now    # iterations overlap: the time between two resolutions is one iteration
            total_time += diff_time
            self.last_iteration_seconds = diff_time
            msg = '(Iteration {}) TRAIN LOSS:{:.4f} CER:{:.2f}% LR:{:.7f} TOTAL TIME:{:.7f}'.format(
                (it_ + 1), total_loss / n_tasks, total_cer * 100 / max(total_char, 1), self.get_lr(outer_opt), total_time)
            if rank == 0:
                print(msg)
            logging.info(msg)
            if (it_ + 1) % last_summary_every == 0:
                msg = '(Summary Iteration {} | MA {}) TRAIN LOSS:{:.4f} CER:{:.2f}%'.format(
                    (it_ + 1), window_size, sum(last_sum_loss) / len(last_sum_loss), sum(last_sum_cer) * 100 / sum(last_sum_char))
                if rank == 0:
                    print(msg, flush=True)
                logging.info(msg)

        while it < num_it:
            # like the reference (:141-376) a failing iteration is reported and skipped (new data is fetched, `it` does not
            # advance); unlike it, MAX_CONSECUTIVE_FAILURES failures in a row re-raise instead of looping forever
            try:
                prefetch.join()
                prefetch = threading.Thread(target=fetch_train_batch, args=(train_data_buffer,))
                prefetch.start()

                _, val_data = train_data_buffer[-1][-1]                  # the LAST task's validation batch, shared by all (:168)
                popped = [train_data_buffer[m].pop() for m in range(n_tasks)]
                task_batches = [popped[m][0] for m in my_tasks]
                if world > 1 and (it == start_it or (check_every > 0 and (it + 1) % check_every == 0)):
                    check_replicas(model, val_data, it)
                step = self.enqueue_iteration(model, vocab, task_batches, val_data, n_tasks, inner_opt, outer_opt, args)
                pending.append((it, step))                # (the enqueued step is on record before anything else can raise)
                drain = not self.pipeline or (it + 1) % evaluate_every == 0 or it + 1 >= num_it
                while pending and (drain or len(pending) > self.pipeline_depth):
                    resolve(*pending.popleft())           # older iterations are logged while the device runs the newer ones

                if (it + 1) % evaluate_every == 0:
                    save_fn = lambda metrics, best_model: save_meta_model(model, vocab, (it + 1), inner_opt, outer_opt, metrics,
                                                                          args, best_model=best_model)
                    stop, best_valid_val, count_stop = run_validation(
                        self.forward_one_batch, model, vocab, valid_loader_list, it, args, history, loss_type, save_fn,
                        early_stop_criteria, early_stop_val, best_valid_val, count_stop, rank)
                    clock[0] = time.time()                # evaluation time is not training time
                    if stop:
                        break
                it += 1
                failures = 0
            except KeyboardInterrupt:
                raise
            except Exception as e:
                failures += 1
                if failures >= MAX_CONSECUTIVE_FAILURES or world > 1:     # ranks must not skip different iterations
                    raise
                print('Error: {}, fetching new data...'.format(e), flush=True)
                logging.info('Error: {}, fetching new data...'.format(e))
                torch.cuda.synchronize(dev)
                # an iteration that was enqueued before the failure is complete now: log it and advance past it BEFORE the retry
                # re-uses its read-back buffers (the failed enqueue has flipped the buffer set)
                while pending:
                    done_it, done_step = pending.popleft()
                    try:
                        resolve(done_it, done_step)
                    except Exception as e2:               # its Adam step has been applied either way
                        logging.info('Error while resolving iteration {}: {}'.format(done_it + 1, e2))
                    it = max(it, done_it + 1)
        while pending:
            resolve(*pending.popleft())
        prefetch.join()
        self.history = history


class JointTrainer():
    """Drop-in for trainer/asr/joint_trainer.py (BASELINE.json configs[0], joint_train.py): per iteration the gradient of
    sum_m L_tr,m / n over all tasks, optional clip, ONE Adam(lr=args.lr) step (joint_trainer.py:182-262, no discriminator).
    Same engine and kernels as the meta loop; tasks shard over ranks the same way."""

    def __init__(self):
        logging.info('Joint Trainer is initialized')

    def get_lr(self, optimizer):
        return optimizer.param_groups[0]['lr']

    def run_iteration(self, model, vocab, task_batches, n_tasks, opt, args):
        dev = model.flat_parameters.device
        g = model.flat_grad
        smoothing = float(getattr(args, 'label_smoothing', 0.0) or 0.0)
        g.zero_()                                                       # opt.zero_grad()
        reads = []
        for (tx, tsz, _tp, ty, _tl) in task_batches:
            out = model.pass_forward(tx.to(dev, non_blocking=True), tsz, ty, smoothing=smoothing)
            reads.append(_Readback(out, ('joint', len(reads), 0)))
            model.pass_backward(g, 1.0 / n_tasks)                       # (tr_loss / n).backward()
        mdist.allreduce_sum_(g)
        if args.clip:
            clip_flat_grad_(model, g, args.max_norm)
        opt.step(g)
        torch.cuda.synchronize(dev)
        total_loss, total_cer, total_char = 0.0, 0, 0
        for rd in reads:
            c, n = cer_counts(vocab, rd.gold_host, rd.hyp)
            total_cer += c
            total_char += n
            total_loss += float(rd.loss[0])
        return mdist.allreduce_scalars([total_loss, total_cer, total_char], dev)

    def train(self, model, vocab, train_data_list, valid_loader_list, loss_type, start_it, num_it, args, evaluate_every=1000,
              window_size=100, last_summary_every=1000, last_metrics=None, early_stop=10, cpu_state_dict=False, is_copy_grad=False,
              opt_name='adam', discriminator=None):
        if loss_type != 'ce' or discriminator is not None or opt_name != 'adam':
            raise NotImplementedError("accelerated joint training: loss_type='ce', opt_name='adam', no discriminator")
        rank, world = mdist.rank(), mdist.world_size()
        if rank == 0:
            print('TRAIN')
        hostenv.bound_torch_threads()
        model.train()
        opt = FlatAdam(model, args.lr)                                  # the reference builds a fresh Adam per train() call
        self.opt = opt
        n_tasks = len(train_data_list)
        my_tasks = mdist.shard_tasks(n_tasks, rank, world)
        buf = [[] for _ in range(n_tasks)]

        takes_need = [_sample_takes_need(ds) for ds in train_data_list]

        def fetch():
            for m in range(n_tasks):                         # all tasks are drawn; only this rank's training batches are loaded
                if takes_need[m]:
                    item = train_data_list[m].sample(args.k_train, 1, m, need=(m in my_tasks, False))
                else:
                    item = train_data_list[m].sample(args.k_train, 1, m)
                buf[m].insert(0, item)
        prefetch = threading.Thread(target=fetch)
        prefetch.start()
        total_time, it = 0, start_it
        self.loss_trace = []
        history = []
        best_valid_val, count_stop, failures = 1000000000, 0, 0
        early_stop_criteria, early_stop_val = early_stop.split(',')[0], int(early_stop.split(',')[1])
        last_sum_loss, last_sum_cer, last_sum_char = deque(maxlen=window_size), deque(maxlen=window_size), deque(maxlen=window_size)
        sync_replicas_from_rank0(model, [opt])
        fob = TransientTrainer.forward_one_batch.__get__(self)             # the two reference trainers share this method body
        while it < num_it:
            try:
                prefetch.join()
                prefetch = threading.Thread(target=fetch)
                prefetch.start()
                start_time = time.time()
                popped = [buf[m].pop() for m in range(n_tasks)]
                total_loss, total_cer, total_char = self.run_iteration(model, vocab, [popped[m][0] for m in my_tasks], n_tasks, opt,
                                                                       args)
                total_time += time.time() - start_time
                self.loss_trace.append(total_loss / n_tasks)
                last_sum_cer.append(total_cer)
                last_sum_char.append(total_char)
                last_sum_loss.append(total_loss)
                msg = '(Iteration {}) TRAIN LOSS:{:.4f} CER:{:.2f}% LR:{:.7f} TOTAL TIME:{:.7f}'.format(
                    (it + 1), total_loss / n_tasks, total_cer * 100 / max(total_char, 1), self.get_lr(opt), total_time)
                if rank == 0:
                    print(msg)
                logging.info(msg)
                if (it + 1) % last_summary_every == 0:
                    msg = '(Summary Iteration {} | MA {}) TRAIN LOSS:{:.4f} CER:{:.2f}%'.format(
                        (it + 1), window_size, sum(last_sum_loss) / len(last_sum_loss), sum(last_sum_cer) * 100 / sum(last_sum_char))
                    if rank == 0:
                        print(msg, flush=True)
                    logging.info(msg)
                if (it + 1) % evaluate_every == 0:                         # joint_trainer.py:306-380
                    save_fn = lambda metrics, best_model: save_joint_model(model, vocab, (it + 1), opt, metrics, args,
                                                                           best_model=best_model)
                    stop, best_valid_val, count_stop = run_validation(
                        fob, model, vocab, valid_loader_list, it, args, history, loss_type, save_fn, early_stop_criteria,
                        early_stop_val, best_valid_val, count_stop, rank)
                    if stop:
                        break
                it += 1
                failures = 0
            except KeyboardInterrupt:
                raise
            except Exception as e:                                          # joint_trainer.py:382-392: report, fetch new data
                failures += 1
                if failures >= MAX_CONSECUTIVE_FAILURES or world > 1:
                    raise
                print('Error: {}, fetching new data...'.format(e), flush=True)
                logging.info('Error: {}, fetching new data...'.format(e))
                torch.cuda.synchronize(model.flat_parameters.device)
        prefetch.join()
        self.history = history
