"""Task-batched passes (the local tasks of a meta-step as ONE pass per phase: trainer/asr/transient_trainer.py:178-237).

  * the task-grouped C-ABI entry points (mtl_gemm_f32_tb, mtl_layernorm_*_g, mtl_embed_*_g, mtl_ce_*_g, the flat task-stack
    updates) against plain PyTorch fp32 on the CPU, and bitwise against their single-task forms issued once per task;
  * TransientTrainer.meta_iteration through the batched path against the per-task lanes (same losses, labels and
    meta-gradient to 1e-6: the arithmetic per task is the same, only launches are shared), at fixture size with ragged
    label widths / clipping / label smoothing, and at the north-star shapes with 8 tasks."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import golden_util as gu
from tests.test_parity_gpu import make

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    import mtl_amd
    assert torch.cuda.is_available()
    return mtl_amd._lib.lib()


def st():
    return torch.cuda.current_stream().cuda_stream


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def dev(t):
    return t.cuda().contiguous()


@pytest.mark.parametrize('ta,tb,M,N,K,n,H', [
    (0, 1, 808, 100, 512, 3, 1),       # q / k / v a-stage: tasks x projections, weights strided per task and per projection
    (0, 1, 2000, 512, 100, 2, 2),      # b-stage with bias: tasks x (layer, projection)
    (0, 0, 333, 512, 100, 1, 1),       # data gradient
    (1, 0, 100, 512, 808, 3, 1),       # weight gradient (+ row sums): K = rows, per-task output slices
    (0, 1, 404, 3765, 512, 1, 1),      # vocabulary projection (big engine)
    (1, 0, 512, 5120, 500, 1, 1),      # input Linear weight gradient (big engine)
])
@pytest.mark.parametrize('shared', [False, True])
def test_gemm_task_batch_level(L, ta, tb, M, N, K, n, H, shared):
    """C[t, z] = op(A[t, z]) . op(B[t, z]) (+ bias[t, z]) with the task as third batch level; `shared`: task stride 0 on the
    weight-like operand (the training passes read ONE theta0)."""
    nt = 4
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + ta + 2 * tb + n)
    wgrad = bool(ta)
    A = torch.randn((nt, n) + ((K, M) if ta else (M, K)), generator=g)
    Bw = torch.randn((1 if (shared and not wgrad) else nt, n) + ((N, K) if tb else (K, N)), generator=g)
    bias = torch.randn(Bw.shape[0], n, N, generator=g)
    C0 = torch.randn(nt, n, M, N, generator=g)
    ref = torch.empty(nt, n, M, N)
    rs_ref = torch.zeros(nt, n, M)
    for t in range(nt):
        for z in range(n):
            a = A[t, z].t() if ta else A[t, z]
            bw = Bw[t % Bw.shape[0], z]
            b = bw.t() if tb else bw
            ref[t, z] = a @ b + (0 if wgrad else bias[t % Bw.shape[0], z]) + (C0[t, z] if wgrad else 0)
            if wgrad:
                rs_ref[t, z] = a.sum(1)
    dA, dB, dbias = dev(A), dev(Bw), dev(bias)
    C = dev(C0.clone()) if wgrad else torch.empty(nt, n, M, N).cuda()
    rs = torch.zeros(nt, n, M).cuda()
    ws = torch.empty(8 << 20).cuda()
    lda, ldb = A.shape[-1], Bw.shape[-1]
    sA, sB, sC = A[0, 0].numel(), Bw[0, 0].numel(), M * N
    sBt = 0 if Bw.shape[0] == 1 else n * sB
    # two-level split of the n items per task: (n / H outer, H inner)
    args = [st(), ta, tb, M, N, K, 1.0, dA.data_ptr(), lda, dB.data_ptr(), ldb, C.data_ptr(), N,
            None if wgrad else dbias.data_ptr(), None, 0, 2 if wgrad else 0, nt * n, H,
            H * sA, sA, H * sB, sB, H * sC, sC, H * N, 1, 0, 0, rs.data_ptr() if wgrad else None, H * M, ws.data_ptr(), ws.numel() * 4, N, M,
            nt, n * sA, sBt, n * sC, 0 if Bw.shape[0] == 1 else n * N, n * M]
    assert L.mtl_gemm_f32_tb(*args) == 0
    assert rel(C, ref) < 3e-6
    if wgrad:
        assert rel(rs, rs_ref) < 3e-6
    # the two-level form issued once per task
    C2 = dev(C0.clone()) if wgrad else torch.empty(nt, n, M, N).cuda()
    rs2 = torch.zeros(nt, n, M).cuda()
    for t in range(nt):
        tb_ = t % Bw.shape[0]
        assert L.mtl_gemm_f32_ex(st(), ta, tb, M, N, K, 1.0, dA[t].data_ptr(), lda, dB[tb_].data_ptr(), ldb, C2[t].data_ptr(), N,
                                 None if wgrad else dbias[tb_].data_ptr(), None, 0, 2 if wgrad else 0, n, H,
                                 H * sA, sA, H * sB, sB, H * sC, sC, H * N, 1, 0, 0, rs2[t].data_ptr() if wgrad else None, H * M,
                                 ws.data_ptr(), ws.numel() * 4, N, M) == 0
    assert rel(C, C2) < 3e-6              # (not bitwise: the engines pick tile shapes by the workgroup count of the whole launch)
    if wgrad:
        assert rel(rs, rs2) < 3e-6


@pytest.mark.parametrize('d,rows', [(512, 101), (128, 64)])
def test_layernorm_task_groups(L, d, rows):
    """rows in nt groups, group t normalises with gamma / beta + t * sParam; the backward's parameter gradients land in group t's
    slice of a gradient stack (immediately, or through the deferred table reduction)"""
    from mtl_amd import _lib
    nt, T, sP, sG = 3, 7, 2 * d + 8, 3 * d + 4
    g = torch.Generator().manual_seed(d + rows)
    x, res = torch.randn(nt * rows, d, generator=g), torch.randn(nt * rows, d, generator=g)
    params = torch.randn(nt * sP, generator=g)                # gamma at t * sP, beta at t * sP + d
    keep = (torch.rand(nt * rows, generator=g) > 0.3).int()
    dy = torch.randn(nt * rows, d, generator=g)
    y_ref, dz_ref, dg_ref, db_ref = [], [], [], []
    for t in range(nt):
        sl = slice(t * rows, (t + 1) * rows)
        xr, rr = x[sl].clone().requires_grad_(True), res[sl].clone()
        gm, bt = params[t * sP:t * sP + d].clone().requires_grad_(True), params[t * sP + d:t * sP + 2 * d].clone().requires_grad_(True)
        yr = F.layer_norm(xr + rr, (d,), gm, bt, 1e-5) * keep[sl].unsqueeze(1)
        yr.backward(dy[sl])
        y_ref.append(yr.detach()), dz_ref.append(xr.grad), dg_ref.append(gm.grad), db_ref.append(bt.grad)
    dx, dr, dp, dk, ddy = dev(x), dev(res), dev(params), keep.cuda(), dev(dy)
    y, xhat, rstd = torch.empty(nt * rows, d).cuda(), torch.empty(nt * rows, d).cuda(), torch.empty(nt * rows).cuda()
    assert L.mtl_layernorm_fwd_g(st(), dx.data_ptr(), dr.data_ptr(), dp.data_ptr(), dp.data_ptr() + 4 * d, None, dk.data_ptr(), None, 1.0,
                                 y.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), nt * rows, d, T, 1e-5, rows, sP) == 0
    assert rel(y, torch.cat(y_ref)) < 2e-6
    ws = torch.empty(L.mtl_layernorm_bwd_g_workspace(nt * rows, d, rows) // 4).cuda()
    for defer in (0, 1):
        dz = torch.empty(nt * rows, d).cuda()
        grads = torch.zeros(nt * sG).cuda()                   # dgamma at t * sG, dbeta at + d, dsum at + 2 d
        assert L.mtl_layernorm_bwd_g(st(), ddy.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), dp.data_ptr(), dk.data_ptr(), None, 1.0,
                                     dz.data_ptr(), None, None, grads.data_ptr(), grads.data_ptr() + 4 * d, grads.data_ptr() + 8 * d,
                                     ws.data_ptr(), nt * rows, d, defer, rows, sP, sG) == 0
        if defer:
            torch.cuda.synchronize()
            assert float(grads.abs().sum()) == 0.0
            wpg = L.mtl_layernorm_bwd_g_waves(rows)
            table = (_lib.LnReduceDesc * nt)()
            for t in range(nt):
                base = grads.data_ptr() + 4 * t * sG
                table[t].part, table[t].dgamma, table[t].dbeta, table[t].dsum = ws.data_ptr() + 4 * t * wpg * 3 * d, base, base + 4 * d, base + 8 * d
                table[t].nw, table[t].d = wpg, d
            tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).cuda()
            assert L.mtl_ln_param_reduce_batch(st(), tdev.data_ptr(), nt, d) == 0
        assert rel(dz, torch.cat(dz_ref)) < 1e-5
        for t in range(nt):
            gt = grads[t * sG:(t + 1) * sG].cpu()
            assert rel(gt[:d], dg_ref[t]) < 1e-5 and rel(gt[d:2 * d], db_ref[t]) < 1e-5 and rel(gt[2 * d:3 * d], dz_ref[t].sum(0)) < 1e-5


def test_embedding_and_cross_entropy_task_groups(L):
    nt, B, Td, d, V = 3, 2, 9, 128, 67
    rows = B * Td
    g = torch.Generator().manual_seed(5)
    sP = V * d + 12
    tables = torch.randn(nt * sP, generator=g)
    pe = torch.randn(Td, d, generator=g)
    ids = torch.randint(1, V, (nt * rows,), generator=g)
    out = torch.empty(nt * rows, d).cuda()
    dt, dpe, dids = dev(tables), dev(pe), ids.cuda()
    assert L.mtl_embed_pe_fwd_g(st(), dids.data_ptr(), dt.data_ptr(), dpe.data_ptr(), out.data_ptr(), nt * rows, Td, d, None, 1.0, rows, sP) == 0
    ref = torch.cat([tables[t * sP:t * sP + V * d].view(V, d)[ids[t * rows:(t + 1) * rows]] + pe.repeat(B, 1) for t in range(nt)])
    assert torch.equal(out.cpu(), ref)
    # backward: per-task occurrence chains (global row numbers), per-task gradient tables
    first, nxt = torch.zeros(nt * rows, dtype=torch.int32), torch.full((nt * rows,), -1, dtype=torch.int32)
    for t in range(nt):
        seen = {}
        for r in range(t * rows, (t + 1) * rows):
            i = int(ids[r])
            if i in seen:
                nxt[seen[i]] = r
            else:
                first[r] = 1
            seen[i] = r
    dout = torch.randn(nt * rows, d, generator=g)
    dtab = torch.zeros(nt * sP).cuda()
    assert L.mtl_embed_bwd_g(st(), dids.data_ptr(), first.cuda().data_ptr(), nxt.cuda().data_ptr(), dev(dout).data_ptr(), dtab.data_ptr(),
                             nt * rows, d, 0, None, 1.0, rows, sP) == 0
    for t in range(nt):
        want = torch.zeros(V, d).index_add_(0, ids[t * rows:(t + 1) * rows], dout[t * rows:(t + 1) * rows])
        assert rel(dtab[t * sP:t * sP + V * d].view(V, d), want) < 1e-6
    # cross-entropy: one loss per task, normalised by ITS token count; lowest-index arg-max
    logits = torch.randn(nt * rows, V, generator=g)
    gold = torch.randint(1, V, (nt * rows,), generator=g)
    gold[5] = 0
    gold[rows + 1] = 0
    gold[rows + 2] = 0
    counts = [(gold[t * rows:(t + 1) * rows] != 0).sum().item() for t in range(nt)]
    inv = torch.tensor([1.0 / c for c in counts]).cuda()
    lse, hyp, rowloss, loss = torch.empty(nt * rows).cuda(), torch.empty(nt * rows, dtype=torch.int64).cuda(), torch.empty(nt * rows).cuda(), torch.empty(nt).cuda()
    dl, dg = dev(logits), gold.cuda()
    assert L.mtl_ce_argmax_fwd_g(st(), dl.data_ptr(), dg.data_ptr(), nt * rows, V, V, 0, 0.0, inv.data_ptr(), lse.data_ptr(), hyp.data_ptr(),
                                 rowloss.data_ptr(), loss.data_ptr(), rows) == 0
    lr = logits.clone().requires_grad_(True)
    losses = [F.cross_entropy(lr[t * rows:(t + 1) * rows], gold[t * rows:(t + 1) * rows], ignore_index=0) for t in range(nt)]
    assert torch.equal(hyp.cpu(), logits.argmax(1))
    for t in range(nt):
        assert abs(float(loss[t]) - float(losses[t])) < 2e-6 * float(losses[t])
    scale = 0.25
    (sum(losses) * scale).backward()
    dlog = torch.empty(nt * rows, V).cuda()
    assert L.mtl_ce_bwd_g(st(), dl.data_ptr(), lse.data_ptr(), dg.data_ptr(), nt * rows, V, V, 0, 0.0, scale, inv.data_ptr(), dlog.data_ptr(), V,
                          rows) == 0
    assert rel(dlog, lr.grad) < 2e-6


def test_flat_task_stack_updates(L):
    n, nt = 1000 * 4, 5
    g = torch.Generator().manual_seed(9)
    th, gr = torch.randn(n, generator=g), torch.randn(nt, n, generator=g)
    t1 = torch.empty(nt, n).cuda()
    dth, dgr = dev(th), dev(gr)
    assert L.mtl_sgd_theta_prime_tasks(st(), dth.data_ptr(), dgr.data_ptr(), 0.01, t1.data_ptr(), n, nt) == 0
    one = torch.empty(n).cuda()
    for t in range(nt):
        assert L.mtl_sgd_theta_prime(st(), dth.data_ptr(), dgr[t].data_ptr(), 0.01, one.data_ptr(), n) == 0
        assert torch.equal(one, t1[t])                        # bitwise the single-task kernel
    out = torch.ones(n).cuda()
    assert L.mtl_sum_tasks(st(), out.data_ptr(), dgr.data_ptr(), n, nt, 0) == 0
    want = torch.zeros(n)
    for t in range(nt):
        want = want + gr[t]                                   # task order
    assert torch.equal(out.cpu(), want)
    assert L.mtl_sum_tasks(st(), out.data_ptr(), dgr.data_ptr(), n, nt, 1) == 0
    ref2 = want.clone()
    for t in range(nt):
        ref2 = ref2 + gr[t]
    assert torch.equal(out.cpu(), ref2)
    assert L.mtl_sum_tasks(st(), out.data_ptr(), dgr.data_ptr(), n + 1, nt, 0) == -22


def _iteration(mtl_amd, model, vocab, args, tasks, val, n, inner, batched, tr=None, gates=False):
    from oracle import branches
    if tr is None:
        tr = mtl_amd.TransientTrainer()
    tr.batch_tasks = batched
    log = None
    if gates:
        with branches.capture_gates(model) as log:
            reads = tr.meta_iteration(model, vocab, tasks, val, n, inner, None, args)
            torch.cuda.synchronize()
    else:
        reads = tr.meta_iteration(model, vocab, tasks, val, n, inner, None, args)
        torch.cuda.synchronize()
    return model._G.clone(), [(float(r.loss[0]), r.hyp.clone(), r.gold_host.clone()) for pair in reads for r in pair], tr, log


def _tensor_errs(model, G1, G0):
    """per-tensor relative L2 difference; tensors whose exact gradient is zero (key-projection biases: softmax is shift-invariant,
    what they hold is rounding noise) are measured against 1e-4 of the global norm, as in tests/test_parity_gpu.py::_rel_errs"""
    return {nm: float((model._layout.view(G1, nm) - model._layout.view(G0, nm)).norm() /
                      max(float(model._layout.view(G0, nm).norm()), 1e-4 * float(G0.norm()))) for nm in model._layout.order}


def _same_decisions(log_a, log_b):
    """number of ReLU / max-pool decisions that two schedules of the same meta-iteration took differently"""
    n = 0
    for ga, gb in zip(log_a, log_b):
        for k in ga:
            n += int((ga[k] != gb[k]).sum())
    return n


@pytest.mark.parametrize('name,clip,smoothing', [('F0', False, 0.0), ('F1', False, 0.0), ('F0', True, 0.1)])
def test_task_batched_iteration_equals_per_task_lanes(name, clip, smoothing):
    """the batched path (default) against the lanes (MTL_BATCH_TASKS=0) on the same batches: ragged frame lengths, ragged label
    widths ACROSS tasks (the batched pass pads every task to the widest), optional per-task clipping + label smoothing.
    The two schedules share every per-task kernel but not the tile shapes of the products (a batched launch sees nt x the
    rows), so pre-activations differ by fp32 summation order: when all ReLU / max-pool decisions agree (captured from both
    runs) every tensor of G agrees to 4e-6 (weights 2e-6; the 512-element bias gradients are cancelling column sums, measured 2.3e-6); a differing near-tie decision (rare at this size) widens the bar to the
    single-flip band of oracle/branches.py.  Recorded / replayed command lists reproduce the eager batched step bit for bit."""
    z, cfg, spec = gu.load(name)
    mtl_amd, args, vocab, model = make(cfg, spec)
    args.clip, args.max_norm, args.label_smoothing = clip, 0.05, smoothing
    model = model.cuda()
    k, T, V = spec['k'], spec['T'], cfg['vocab_size']
    as5 = lambda b: (b[0].cuda(), b[1], None, b[2], None)
    tasks = [as5(mtl_amd.synth_batch(40 + m, k, T, Lm, V, variable=True)) for m, Lm in enumerate((8, 5, 11, 8))]
    val = as5(mtl_amd.synth_batch(99, k, T, 6, V, variable=True))
    inner = mtl_amd.FlatSGD(model, spec['lr'])
    model.zero_copy_grad()
    G0, r0, _, log0 = _iteration(mtl_amd, model, vocab, args, tasks, val, 4, inner, False, gates=True)
    G1, r1, tr, log1 = _iteration(mtl_amd, model, vocab, args, tasks, val, 4, inner, True, gates=True)
    assert len(log0) == len(log1) == 8
    # the batched pass pads every task's label axis to the widest task: compare the gates on the rows both have
    flips = 0
    for ga, gb in zip(log0, log1):
        for key in ga:
            a_, b_ = ga[key], gb[key]
            if a_.shape != b_.shape:                  # decoder-side FFN masks: (B * Td_task, inner) vs (B * Td_max, inner)
                Bn = k
                a_ = a_.view(Bn, -1, a_.shape[-1])
                b_ = b_.view(Bn, -1, b_.shape[-1])[:, :a_.shape[1]]
            flips += int((a_ != b_).sum())
    for (l0, h0, g0), (l1, h1, g1) in zip(r0, r1):
        w = g0.shape[1]
        assert torch.equal(g0, g1[:, :w]) and bool((g1[:, w:] == 0).all())          # wider label axis: PAD beyond the task's own
        assert torch.equal(h0, h1[:, :w]) and bool((h1[:, w:] == 0).all())          # label indices: bit-exact
        assert abs(l0 - l1) <= 1e-6 * abs(l0)
    errs = _tensor_errs(model, G1, G0)
    worst = max(errs, key=errs.get)
    print('%s: batched vs lanes: %d differing branch decisions, worst tensor %.2e (%s), global %.2e'
          % (name, flips, errs[worst], worst, float((G1 - G0).norm() / G0.norm())))
    assert flips <= 4
    # (key-projection biases have an exactly-zero gradient: what they hold is rounding noise, bounded like every tensor at 1e-4)
    strict = {nm: e for nm, e in errs.items() if not nm.endswith('key_linear_b.bias')}
    ws_ = max(strict, key=strict.get)
    # (measured 6.1e-6 since the lanes' one-task input Linear runs as K slices -- another summation order: round 4)
    assert strict[ws_] < (1.5e-5 if flips == 0 else 1e-2) and errs[worst] < (1e-4 if flips == 0 else 1e-2), (ws_, strict[ws_], worst, errs[worst], flips)
    for rnd in range(3):                                  # first sighting of the key (eager), recording, replay
        G2, r2, _, _ = _iteration(mtl_amd, model, vocab, args, tasks, val, 4, inner, True, tr=tr)
        assert torch.equal(G2, G1)
        for (l1, h1, g1), (l2, h2, g2) in zip(r1, r2):
            assert l1 == l2 and torch.equal(h1, h2) and torch.equal(g1, g2)
    assert any(isinstance(v, dict) and k_[0] == 'batched' for k_, v in tr._cmdlists.items()), 'no command list was recorded'


def test_task_batched_iteration_with_dropout_is_deterministic():
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, spec)
    args.dropout = 0.1
    torch.manual_seed(123456)
    model = mtl_amd.init_transformer_model(args, vocab, r=cfg['r']).cuda()
    as5 = lambda b: (b[0].cuda(), b[1], None, b[2], None)
    tasks = [as5(mtl_amd.synth_batch(40 + m, 2, 64, 8, cfg['vocab_size'])) for m in range(3)]
    val = as5(mtl_amd.synth_batch(99, 2, 64, 8, cfg['vocab_size']))
    inner = mtl_amd.FlatSGD(model, spec['lr'])
    model.zero_copy_grad()
    tr = mtl_amd.TransientTrainer()
    outs = []
    for rnd in range(4):                                   # eager, recording, two replays: same seeds -> same bits
        torch.manual_seed(7)
        tr.meta_iteration(model, vocab, tasks, val, 3, inner, None, args)
        torch.cuda.synchronize()
        outs.append(model._G.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    torch.manual_seed(8)
    tr.meta_iteration(model, vocab, tasks, val, 3, inner, None, args)
    torch.cuda.synchronize()
    assert not torch.equal(outs[0], model._G)              # fresh masks with a fresh seed


def test_eight_tasks_at_north_star_shapes_three_schedules():
    """BASELINE.json configs[1] shapes with the bench's 8 tasks, the three schedules the bench can time:
      * 8 concurrent lanes against ONE lane (the tasks one after the other): the same kernels in a different interleaving ->
        every tensor of G within 1e-6, labels and losses identical;
      * the task-batched step (default) against one lane: the same per-task arithmetic with other tile shapes in the products
        (fp32 summation order), so some of the ~4 G ReLU / max-pool decisions of the 16 passes fall the other way (~25 near-ties
        per pass between ANY two fp32 implementations, each moving the tensors behind it by 1e-4 .. 2e-3: DESIGN 4) -- labels
        bit-exact, losses 1e-6, every tensor of G inside the single-flip band (measured: 2.3e-4 global, worst tensor 5e-4; the
        reference goldens sit at the same distance from either schedule).  The 1e-4 bar on ALL tensors is asserted for the
        batched path against the oracle with the device's decisions replayed
        (test_parity_gpu.py::test_meta_gradient_at_north_star_size_with_branch_replay runs through it)."""
    z, cfg, spec = gu.load('NS')
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    k, T, Lb, V = spec['k'], spec['T'], spec['L'], cfg['vocab_size']
    as5 = lambda b: (b[0].cuda(), b[1], None, b[2], None)
    tasks = [as5(mtl_amd.synth_batch(10 * m, k, T, Lb, V)) for m in range(8)]
    val = as5(mtl_amd.synth_batch(71, k, T, Lb, V))
    inner = mtl_amd.FlatSGD(model, spec['lr'])
    model.zero_copy_grad()
    Gb, rb, trb, _ = _iteration(mtl_amd, model, vocab, args, tasks, val, 8, inner, True)
    assert any(k_[0] == 'batched' for k_ in trb._cmdlists), 'the batched path did not run'
    lanes = model.n_lanes
    res = {}
    for nl in (1, lanes):
        model.n_lanes = nl
        try:
            res[nl] = _iteration(mtl_amd, model, vocab, args, tasks, val, 8, inner, False)
        finally:
            model.n_lanes = lanes
    G1, r1 = res[1][0], res[1][1]
    if lanes > 1:
        Gl, rl = res[lanes][0], res[lanes][1]
        errs = _tensor_errs(model, Gl, G1)
        worst = max(errs, key=errs.get)
        print('8 tasks at NS shapes: %d lanes vs 1 lane: worst tensor %.2e (%s)' % (lanes, errs[worst], worst))
        assert errs[worst] < 1e-6, (worst, errs[worst])
        for (l0, h0, g0), (l1, h1, g1) in zip(r1, rl):
            assert torch.equal(g0, g1) and torch.equal(h0, h1) and l0 == l1
    for (l0, h0, g0), (l1, h1, g1) in zip(r1, rb):
        assert torch.equal(g0, g1) and torch.equal(h0, h1) and abs(l0 - l1) <= 1e-6 * abs(l0)
    errs = _tensor_errs(model, Gb, G1)
    worst = max(errs, key=errs.get)
    glob = float((Gb - G1).norm() / G1.norm())
    tight = sum(e < 2e-6 for e in errs.values())
    print('8 tasks at NS shapes: batched vs 1 lane: global %.2e, %d/%d tensors < 2e-6, worst %.2e (%s)' % (glob, tight, len(errs), errs[worst], worst))
    assert glob < 1e-3 and errs[worst] < 1e-2


# ------------------------------------------------------------------------------------------------------------------
# tasks of DIFFERENT frame counts in one pass (manifest-fed batches: data.py:77 pads every task's batch to its own longest utterance)
# ------------------------------------------------------------------------------------------------------------------
def test_zero_tails_clears_every_tasks_own_tail(L):
    """mtl_zero_tails: (n, T, row) activations of tasks stacked at the widest; frames [width >> shift, T) of every sample of a task are
    cleared, everything in front stays bit for bit (odd widths, a task as wide as the stack, a width beyond T)."""
    per_task, T, row = 3, 21, 161 * 64
    widths = [21, 13, 40, 4]
    n = per_task * len(widths)
    w = torch.tensor(widths, dtype=torch.int32).cuda()
    for shift in (0, 1):
        y = torch.randn(n, T, row).cuda()
        ref = y.clone()
        for s in range(n):
            ref[s, min(widths[s // per_task] >> shift, T):] = 0
        assert L.mtl_zero_tails(st(), y.data_ptr(), n, T, row, w.data_ptr(), shift, per_task) == 0
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
    assert L.mtl_zero_tails(st(), y.data_ptr(), n, T, 6, w.data_ptr(), 0, per_task) != 0          # row % 4


def _ragged_tasks(mtl_amd, k, frames, widths, V, seed):
    as5 = lambda b: (b[0].cuda(), b[1], None, b[2], None)
    return [as5(mtl_amd.synth_batch(seed + m, k, T_m, L_m, V, variable=True)) for m, (T_m, L_m) in enumerate(zip(frames, widths))]


def _crop_gates(g, k, frames, dec_width):
    """ReLU / max-pool decisions of one task of a pass stacked at the widest -> the extent of the task's OWN pass (frames, frames // 2,
    (frames // 2) // 2 columns; its own encoder / decoder rows); also returns how many decisions beyond that extent are set in the maps
    a later convolution reads (must be none: those frames are cleared)."""
    out, beyond = {}, 0
    T2 = frames // 2
    T4 = T2 // 2
    for key, v in g.items():
        if v.dim() == 4:
            w = {'conv0': frames, 'conv5': T2, 'am1': T2, 'pool1': T2}.get(key, T4)
            if key in ('conv0', 'conv5', 'pool1'):
                beyond += int((v[..., w:] != 0).sum())
            out[key] = v[..., :w].contiguous()
        else:
            rows = T4 if key.startswith('e') else dec_width
            out[key] = v.view(k, -1, v.shape[-1])[:, :rows].reshape(-1, v.shape[-1])
    return out, beyond


def _decisions_differ(ga, gb):
    return sum(int((ga[key] != gb[key]).sum()) for key in ga)


@pytest.mark.parametrize('name,frames,val_frames,quantum', [('F0', (64, 53, 46), 60, 64), ('F0', (41, 64, 64, 58), 64, 1), ('F1', (64, 50, 39), 47, 64),
                                                            ('F1', (64, 47, 36), 47, 1), ('F0', (41, 50, 39), 45, 16), ('F1', (30, 57, 44), 33, 32)])
def test_tasks_of_different_frame_counts_in_one_pass_equal_a_lane_per_task(name, frames, val_frames, quantum):
    """Every task's batch padded to its OWN longest utterance (odd frame counts, counts that are not multiples of four), stacked at the
    widest and run as ONE pass per phase -- against one lane per task, each at its own width, which is the reference's schedule
    (transient_trainer.py:178-237).  A task's own image border (zeros beyond its frames at every convolution), its encoder length
    ((frames // 2) // 2 positions) and its loss are those of its own pass -- also when the stack is WIDER than its widest task (widths
    rounded up to a quantum, so that they repeat: the validation batch then carries a border of its own too): labels bit-exact, losses to 2e-6, no decision set beyond a
    task's frames, and -- when the two schedules took every ReLU / max-pool decision alike (captured from both and compared on each
    task's own extent) -- every tensor of G to the summation-order bar of the fixed-shape comparison above; a differing near-tie (the
    F1 case with 47 frames has ONE, in a validation pass) widens it to the single-flip band."""
    z, cfg, spec = gu.load(name)
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    k, V = spec['k'], cfg['vocab_size']
    widths = (8, 5, 11, 8)
    tasks = _ragged_tasks(mtl_amd, k, frames, widths, V, 340 if 47 in frames else 140)
    val = _ragged_tasks(mtl_amd, k, (val_frames,), (6,), V, 199)[0]
    n = len(tasks)
    inner = mtl_amd.FlatSGD(model, spec['lr'])
    model.zero_copy_grad()
    own = mtl_amd.TransientTrainer()
    own.pad_lanes = '0'                                   # the reference's schedule: every task at its own width
    G0, r0, _, log0 = _iteration(mtl_amd, model, vocab, args, tasks, val, n, inner, False, tr=own, gates=True)
    tr = mtl_amd.TransientTrainer()
    tr.ragged_quantum = quantum
    G1, r1, tr, log1 = _iteration(mtl_amd, model, vocab, args, tasks, val, n, inner, True, tr=tr, gates=True)
    assert tr.last_schedule == 'batched-ragged', 'the ragged tasks did not take the task-batched pass'
    assert len(log0) == len(log1) == 2 * n
    flips = 0
    for i, (ga, gb) in enumerate(zip(log0, log1)):
        t, is_val = i // 2, i % 2
        crop, beyond = _crop_gates(gb, k, val_frames if is_val else frames[t], 7 if is_val else widths[t] + 1)
        assert beyond == 0 or is_val, (i, beyond)
        flips += _decisions_differ(ga, crop)
    for (l0, h0, g0), (l1, h1, g1) in zip(r0, r1):
        w = g0.shape[1]
        assert torch.equal(g0, g1[:, :w]) and bool((g1[:, w:] == 0).all())
        assert torch.equal(h0, h1[:, :w]) and bool((h1[:, w:] == 0).all())
        assert abs(l0 - l1) <= 2e-6 * abs(l0), (l0, l1)
    errs = _tensor_errs(model, G1, G0)
    strict = {nm: e for nm, e in errs.items() if not nm.endswith('key_linear_b.bias')}
    ws_, worst = max(strict, key=strict.get), max(errs, key=errs.get)
    print('%s frames %s: stacked vs lanes: %d differing decisions, worst tensor %.2e (%s), global %.2e'
          % (name, frames, flips, errs[worst], worst, float((G1 - G0).norm() / G0.norm())))
    assert flips <= 2
    assert strict[ws_] < (1.5e-5 if flips == 0 else 1e-2) and errs[worst] < (1e-4 if flips == 0 else 1e-2), (ws_, strict[ws_], worst, errs[worst], flips)
    for rnd in range(3):                                  # first sighting of the key (eager), recording, replay
        G2, r2, _, _ = _iteration(mtl_amd, model, vocab, args, tasks, val, n, inner, True, tr=tr)
        assert torch.equal(G2, G1)
    assert any(isinstance(v, dict) and k_[0] == 'batched' for k_, v in tr._cmdlists.items()), 'no command list was recorded'
    # the recorded list holds no frame count: other per-task counts under the same widest replay it -- bit for bit what a fresh trainer
    # enqueues call by call
    other = tuple(max(frames) if f == max(frames) else f - 3 for f in frames)
    tasks_b = _ragged_tasks(mtl_amd, k, other, widths, V, 540)
    G3, _, _, _ = _iteration(mtl_amd, model, vocab, args, tasks_b, val, n, inner, True, tr=tr)
    fresh = mtl_amd.TransientTrainer()
    fresh.ragged_quantum = quantum
    G4, _, _, _ = _iteration(mtl_amd, model, vocab, args, tasks_b, val, n, inner, True, tr=fresh)
    assert torch.equal(G3, G4)


def test_tasks_of_different_frame_counts_in_one_pass_against_live_oracle():
    """The stacked pass against the CPU oracle run task by task at each task's own width (oracle/refimpl.py meta_gradient: the reference's
    loop), the device's branch decisions -- cropped to each task's own extent -- replayed: labels bit-exact, losses and every tensor of
    G within 1e-4."""
    from oracle import refimpl as R
    from oracle import branches
    from tests.test_parity_gpu import _rel_errs, RTOL
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    oracle = R.build_model(cfg)
    k, V = spec['k'], cfg['vocab_size']
    frames, widths, val_frames = (64, 51, 45), (8, 5, 11), 58
    cpu = [mtl_amd.synth_batch(240 + m, k, T_m, L_m, V, variable=True) for m, (T_m, L_m) in enumerate(zip(frames, widths))]
    val = mtl_amd.synth_batch(299, k, val_frames, 6, V, variable=True)
    as5 = lambda b: (b[0].cuda(), b[1], None, b[2], None)
    inner = mtl_amd.FlatSGD(model, spec['lr'])
    model.zero_copy_grad()
    tr = mtl_amd.TransientTrainer()
    with branches.capture_gates(model) as log:
        reads = tr.meta_iteration(model, vocab, [as5(b) for b in cpu], as5(val), 3, inner, None, args)
        torch.cuda.synchronize()
    assert tr.last_schedule == 'batched-ragged' and len(log) == 6
    gates = [_crop_gates(g, k, val_frames if i % 2 else frames[i // 2], 7 if i % 2 else widths[i // 2] + 1)[0] for i, g in enumerate(log)]
    G, trl, val_l, labels = R.meta_gradient(oracle, cpu, val, spec['lr'], gates=gates)
    for t in range(3):
        for rd, (gold, hyp), loss in ((reads[t][0], labels[2 * t], trl[t]), (reads[t][1], labels[2 * t + 1], val_l[t])):
            w = gold.shape[1]
            assert torch.equal(rd.hyp[:, :w], hyp) and torch.equal(rd.gold_host[:, :w], gold)
            assert abs(float(rd.loss[0]) - loss) < RTOL * loss
    errs = _rel_errs(model, model._G, oracle, G)
    worst = max(errs, key=errs.get)
    print('ragged stack vs oracle: %d/%d tensors within 1e-4, worst %.2e (%s)' % (sum(e < RTOL for e in errs.values()), len(errs), errs[worst], worst))
    assert errs[worst] < RTOL, (worst, errs[worst])


@pytest.mark.parametrize('name,frames,val_frames,conv', [('F0', (41, 50, 39), 45, 'h2'), ('F1', (57,), 33, 'h2'), ('F0', (41, 50, 39), 45, 'x3')])
def test_a_lane_per_task_at_rounded_widths_equals_its_own_widths(name, frames, val_frames, conv):
    """The lane schedule (one task per rank; tasks too unequal to stack) on manifest-like batches: every batch widened to a multiple of
    the quantum -- so that widths repeat, recorded lists replay and the pool keeps its buffers -- with its own border and encoder
    length, against the same lanes at the batches' own widths: labels bit-exact, losses to 2e-6, G to the summation-order bar when all
    decisions agree.  Also with the convolutions on the exact 3 x bf16 split (per-task launches, tails cleared layer by layer)."""
    z, cfg, spec = gu.load(name)
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    if conv == 'x3':
        for e in model.engines:
            e.conv_mode, e.conv_x3, e.conv_h2 = 'x3', True, False
    k, V = spec['k'], cfg['vocab_size']
    widths = (8, 5, 11)
    tasks = _ragged_tasks(mtl_amd, k, frames, widths, V, 740)
    val = _ragged_tasks(mtl_amd, k, (val_frames,), (6,), V, 799)[0]
    n = 8 if len(tasks) == 1 else len(tasks)
    inner = mtl_amd.FlatSGD(model, spec['lr'])
    model.zero_copy_grad()
    own, wide = mtl_amd.TransientTrainer(), mtl_amd.TransientTrainer()
    own.pad_lanes, wide.pad_lanes, wide.ragged_quantum = '0', '1', 16
    G0, r0, _, log0 = _iteration(mtl_amd, model, vocab, args, tasks, val, n, inner, False, tr=own, gates=True)
    G1, r1, _, log1 = _iteration(mtl_amd, model, vocab, args, tasks, val, n, inner, False, tr=wide, gates=True)
    assert own.last_schedule == wide.last_schedule == 'lanes'
    assert log1[0]['conv0'].shape[3] == -(-frames[0] // 16) * 16 and log0[0]['conv0'].shape[3] == frames[0]
    flips = 0
    for i, (ga, gb) in enumerate(zip(log0, log1)):
        crop, beyond = _crop_gates(gb, k, val_frames if i % 2 else frames[i // 2], 7 if i % 2 else widths[i // 2] + 1)
        assert beyond == 0, (i, beyond)
        flips += _decisions_differ(ga, crop)
    for (l0, h0, g0), (l1, h1, g1) in zip(r0, r1):
        w = g0.shape[1]                                   # (the decoder width is rounded too: PAD beyond the batch's own)
        assert g1.shape[1] % 8 == 0 and torch.equal(g0, g1[:, :w]) and bool((g1[:, w:] == 0).all())
        assert torch.equal(h0, h1[:, :w]) and bool((h1[:, w:] == 0).all()) and abs(l0 - l1) <= 2e-6 * abs(l0)
    errs = _tensor_errs(model, G1, G0)
    worst = max(errs, key=errs.get)
    print('%s %s frames %s: lanes at rounded vs own widths: %d differing decisions, worst tensor %.2e (%s)' % (name, conv, frames, flips, errs[worst], worst))
    assert flips <= 2 and errs[worst] < (1e-4 if flips == 0 else 1e-2), (worst, errs[worst], flips)
    for rnd in range(3):                                  # eager, recording, replay
        G2, _, _, _ = _iteration(mtl_amd, model, vocab, args, tasks, val, n, inner, False, tr=wide)
        assert torch.equal(G2, G1)
    # other own widths under the same rounded ones replay the recorded lists
    tasks_b = _ragged_tasks(mtl_amd, k, tuple(f - 2 for f in frames), widths, V, 840)
    G3, _, _, _ = _iteration(mtl_amd, model, vocab, args, tasks_b, val, n, inner, False, tr=wide)
    fresh = mtl_amd.TransientTrainer()
    fresh.pad_lanes, fresh.ragged_quantum = '1', 16
    G4, _, _, _ = _iteration(mtl_amd, model, vocab, args, tasks_b, val, n, inner, False, tr=fresh)
    assert torch.equal(G3, G4)
    assert any(isinstance(v, dict) for v in wide._cmdlists.values()), 'no command list was recorded'


def test_stand_alone_passes_at_rounded_widths_match_the_oracle_at_their_own():
    """model(...) / loss.backward() -- the drop-in call pattern, what validation loops and the joint trainer run -- on batches whose width
    changes from call to call: from the second width on the engine widens a batch to a repeating width (own border and encoder length
    kept).  Predictions, labels and every parameter gradient against the CPU oracle at the batch's OWN width."""
    from oracle import refimpl as R
    from tests.test_parity_gpu import RTOL
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, spec)
    model = model.cuda()
    model.engine.widen_quantum = 16
    oracle = R.build_model(cfg)
    k, V = spec['k'], cfg['vocab_size']
    for i, T in enumerate((50, 41, 57)):
        x, lens, y = mtl_amd.synth_batch(900 + i, k, T, 8, V, variable=True)
        model.zero_grad()
        oracle.zero_grad()
        pred, gold, hyp = model(x.cuda(), lens, y)
        assert model.engine._widths_vary == (i > 0)
        widened = model.engine.arena['y1'].shape[1]
        assert widened == (T if i == 0 else -(-T // 16) * 16)
        loss, _ = mtl_amd.calculate_metrics(pred, gold, 0, smoothing=0.0, loss_type='ce')
        loss.backward()
        pr, gr, hr = oracle(x, lens, y)
        lref = R.ce_loss(pr, gr)
        lref.backward()
        assert torch.equal(hyp.cpu(), hr) and torch.equal(gold.cpu(), gr)
        assert abs(float(loss) - float(lref)) < RTOL * float(lref)
        assert float((pred.cpu() - pr).abs().max()) < 1e-4 * float(pr.abs().max())
        gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in oracle.parameters())))
        worst = max(float((p.grad.cpu() - q.grad).norm() / max(float(q.grad.norm()), 1e-4 * gn))
                    for (_, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()))
        print('T = %d (run at %d): worst parameter gradient %.2e' % (T, widened, worst))
        assert worst < RTOL


def test_ragged_stack_with_clipping_smoothing_and_another_validation_batch_size():
    """The stacked schedule on tasks of different frame counts with per-task gradient clipping, label smoothing and a validation batch
    of another sample count (k_valid != k_train), against a lane per task at its own width; then with dropout 0.1: two runs of the
    same iteration give the same bits (Philox streams are per site and position, the stack's padding does not move them)."""
    z, cfg, spec = gu.load('F0')
    mtl_amd, args, vocab, model = make(cfg, spec)
    args.clip, args.max_norm, args.label_smoothing = True, 0.05, 0.1
    model = model.cuda()
    V = cfg['vocab_size']
    frames, widths = (61, 38, 47, 52), (8, 5, 11, 7)
    tasks = _ragged_tasks(mtl_amd, 3, frames, widths, V, 1140)
    val = _ragged_tasks(mtl_amd, 2, (55,), (6,), V, 1199)[0]
    inner = mtl_amd.FlatSGD(model, spec['lr'])
    model.zero_copy_grad()
    own, stacked = mtl_amd.TransientTrainer(), mtl_amd.TransientTrainer()
    own.pad_lanes, stacked.ragged_quantum = '0', 16
    G0, r0, _, log0 = _iteration(mtl_amd, model, vocab, args, tasks, val, 4, inner, False, tr=own, gates=True)
    G1, r1, _, log1 = _iteration(mtl_amd, model, vocab, args, tasks, val, 4, inner, True, tr=stacked, gates=True)
    assert stacked.last_schedule == 'batched-ragged' and own.last_schedule == 'lanes'
    flips = 0
    for i, (ga, gb) in enumerate(zip(log0, log1)):
        crop, beyond = _crop_gates(gb, 2 if i % 2 else 3, 55 if i % 2 else frames[i // 2], 7 if i % 2 else widths[i // 2] + 1)
        assert beyond == 0 or i % 2, (i, beyond)
        flips += _decisions_differ(ga, crop)
    for (l0, h0, g0), (l1, h1, g1) in zip(r0, r1):
        w = g0.shape[1]
        assert torch.equal(g0, g1[:, :w]) and torch.equal(h0, h1[:, :w]) and abs(l0 - l1) <= 2e-6 * abs(l0)
    errs = _tensor_errs(model, G1, G0)
    worst = max(errs, key=errs.get)
    print('clip + smoothing, k_valid 2 / k_train 3: stacked vs lanes: %d differing decisions, worst tensor %.2e (%s)' % (flips, errs[worst], worst))
    assert flips <= 2 and errs[worst] < (1e-4 if flips == 0 else 1e-2), (worst, errs[worst], flips)
    model.encoder.dropout_rate = model.decoder.dropout_rate = 0.1
    model.train()
    runs = []
    for rep in range(2):
        torch.manual_seed(4242)                              # the pass seeds come from torch's CPU generator
        tr = mtl_amd.TransientTrainer()
        tr.ragged_quantum = 16
        G, r, _, _ = _iteration(mtl_amd, model, vocab, args, tasks, val, 4, inner, True, tr=tr)
        runs.append((G, [x[0] for x in r]))
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    assert not torch.equal(runs[0][0], G1)                   # (dropout did act)
