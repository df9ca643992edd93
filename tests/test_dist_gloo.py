"""CPU, world_size 2, gloo: sharding tasks over ranks + ONE all-reduce of the flat meta-gradient reproduces the
single-process sequential accumulation (the N>1 path of SURVEY.md 8(e)); compute is the oracle's (test-only)."""
import os

import torch
import torch.multiprocessing as mp

from tests import golden_util as gu


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import mtl_amd
    from oracle import refimpl as R
    mtl_amd.dist.init_from_env(backend='gloo')
    z, cfg, spec = gu.load('F0')
    model = R.build_model(cfg)
    tr, val = gu.batches_for(cfg, spec, 0, z['data_call_index'])
    n = len(tr)
    mine = mtl_amd.dist.shard_tasks(n, rank, world)
    # local part of G with the GLOBAL 1/n, exactly what each GPU rank accumulates
    params = list(model.parameters())
    G = torch.zeros(sum(p.numel() for p in params))
    for m in mine:
        g_m, _, _, _ = R.meta_gradient(model, [tr[m]], val, spec['lr'])
        # meta_gradient used n=1 for its val term; rebuild with the global n
        pred, gold, _ = model(*tr[m])
        g_tr = torch.autograd.grad(R.ce_loss(pred, gold), params)
        g_val_1 = [a - b for a, b in zip(g_m, g_tr)]
        G += R.flat([a + b / n for a, b in zip(g_tr, g_val_1)])
    mtl_amd.dist.allreduce_sum_(G)
    vals = mtl_amd.dist.allreduce_scalars([1.0, float(rank)], torch.device('cpu'))
    assert vals == [2.0, 1.0]
    if rank == 0:
        G_ref, _, _, _ = R.meta_gradient(model, tr, val, spec['lr'])
        ret['err'] = float((G - R.flat(G_ref)).norm() / R.flat(G_ref).norm())
    mtl_amd.dist.barrier()


def test_two_rank_allreduce_equals_sequential():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29611, ret), nprocs=2, join=True)
    assert ret['err'] < 1e-6, ret['err']


def _replica_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import argparse
    import mtl_amd
    from mtl_amd import trainer as T
    mtl_amd.dist.init_from_env(backend='gloo')
    z, cfg, spec = gu.load('F0')
    args = argparse.Namespace(feat_extractor='vgg_cnn', sample_rate=16000, window_size=.02, feat='spectrogram', dim_input=161,
                              dropout=0.0, emb_trg_sharing=False, **{k: v for k, v in cfg.items() if k not in ('vocab_size', 'r')})
    torch.manual_seed(1000 + rank)                                  # ranks start from DIFFERENT weights and Adam states
    model = mtl_amd.init_transformer_model(args, mtl_amd.synthetic_vocab(cfg['vocab_size']), r=cfg['r'])
    adam = mtl_amd.FlatAdam(model, 1e-3)
    adam.m.fill_(float(rank + 1))
    adam.v.fill_(float(rank + 2))
    adam.step_count = 5 + rank
    T.sync_replicas_from_rank0(model, [adam])
    probe = torch.stack([model.flat_parameters.double().sum(), adam.m.double().sum(), adam.v.double().sum()])
    lo, hi = probe.clone(), probe.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    val = mtl_amd.synth_batch(3, 2, 64, 8, cfg['vocab_size'])
    val5 = (val[0], val[1], None, val[2], None)
    T.check_replicas(model, val5, 0)                                # identical replicas, identical validation batch: passes
    raised = [False, False, False]
    try:
        bad = mtl_amd.synth_batch(3 + rank, 2, 64, 8, cfg['vocab_size'])         # rank 1 drew a different validation batch
        T.check_replicas(model, (bad[0], bad[1], None, bad[2], None), 1)
    except RuntimeError:
        raised[0] = True
    if rank == 1:
        model.flat_parameters[123] += 1.0                           # a replica drifted
    try:
        T.check_replicas(model, val5, 2)
    except RuntimeError:
        raised[1] = True
    if rank == 1:
        model.flat_parameters[123] -= 1.0
        a, b = float(model.flat_parameters[5]), float(model.flat_parameters[6])      # two entries swapped: every float checksum is
        model.flat_parameters[5], model.flat_parameters[6] = b, a                    # unchanged, only the bit-level one sees it
    try:
        T.check_replicas(model, val5, 3)
    except RuntimeError as e:
        raised[2] = 'bit checksums' in str(e)
    ret[rank] = (bool(torch.equal(lo, hi)), adam.step_count, float(adam.m[0]), raised)
    mtl_amd.dist.barrier()


def test_replicas_are_synchronised_at_start_and_divergence_is_detected():
    """trainer.sync_replicas_from_rank0 / check_replicas (the multi-rank contract of TransientTrainer.train): broadcast of theta,
    Adam m / v / step from rank 0; a different validation batch, a drifted replica, or two swapped entries (invisible to sums) raise on EVERY rank."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_replica_worker, args=(2, 29613, ret), nprocs=2, join=True)
    for rank in (0, 1):
        same, step, m0, raised = ret[rank]
        assert same and step == 5 and m0 == 1.0 and raised == [True, True, True], (rank, ret[rank])


def _chunk_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import mtl_amd
    mdist = mtl_amd.dist
    mdist.init_from_env(backend='gloo')
    assert mdist.collective_on() and mdist.chunked_on()
    g = torch.Generator().manual_seed(100 + rank)
    total = 4 * 5003
    local = torch.randn(total, generator=g) * torch.logspace(-6, 3, total)           # wide dynamic range: an order change would show
    bounds = {'encoder': (0, 4 * 1601), 'decoder': (4 * 1601, 4 * 4900), 'conv': (4 * 4900, total)}   # flat layout: encoder | decoder | conv
    whole = local.clone()
    mdist.allreduce_sum_(whole)                                                        # ONE collective after the backward
    chunked = local.clone()
    ch = mdist.ChunkedAllReduce()
    for tag in ('decoder', 'encoder', 'conv'):                                         # the order the validation backward finishes them in
        lo, hi = bounds[tag]
        ch.issue(chunked[lo:hi])                                                       # asynchronous: three collectives in flight
    assert len(ch.works) == 3
    ch.wait()
    assert not ch.works
    other = torch.randn(total, generator=torch.Generator().manual_seed(100 + (1 - rank))) * torch.logspace(-6, 3, total)
    ret[rank] = (bool(torch.equal(whole, chunked)), bool(torch.equal(whole, local + other) or torch.equal(whole, other + local)))
    os.environ['MTL_CHUNKED_ALLREDUCE'] = '0'
    assert mdist.collective_on() and not mdist.chunked_on()
    mdist.barrier()


def test_chunked_allreduce_equals_the_single_collective_bitwise():
    """The meta-gradient leaves for its all-reduce in three slices (decoder, encoder, conv: dist.ChunkedAllReduce, issued by
    trainer._chunk_hook under the validation backward).  World 2 over gloo: every element is a + b whichever way it travels, so
    the three asynchronous collectives must reproduce the single one bit for bit."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_chunk_worker, args=(2, 29617, ret), nprocs=2, join=True)
    assert ret[0] == (True, True) and ret[1] == (True, True), dict(ret)


def test_command_list_breaks_hand_control_back_between_segments():
    """CommandList.add_break / run(on_break): the replay of a recorded task body stops where the eager run called the engine's
    slice hook (PassEngine._slice_done) so that the same host code -- the start of a slice's all-reduce -- runs in both (host
    logic only: the segments here are empty, nothing is dispatched)."""
    import mtl_amd
    cl = mtl_amd._lib.CommandList()
    cl.add_break('decoder')
    cl.add_break('encoder')
    cl.add_break('conv')
    cl.finish()
    seen = []
    cl.run(seen.append)
    assert seen == ['decoder', 'encoder', 'conv'] and cl.n == 0
    cl.run()                                   # no callback: breaks are ignored
