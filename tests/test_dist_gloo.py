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
