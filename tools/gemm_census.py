"""Census of the mtl_gemm_f32 calls of one training pass (forward + backward) at the north-star size: every distinct
(transA, transB, M, N, K, batch, H) with its call count, its isolated duration and its share.  GPU only."""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtl_amd  # noqa: E402

dev = torch.device('cuda:0')
args = bench.make_args(8)
vocab = mtl_amd.synthetic_vocab(bench.CFG['vocab_size'])
model = mtl_amd.init_transformer_model(args, vocab, r=bench.CFG['r']).to(dev)
model.train()
eng = model.engine
calls = collections.Counter()
orig = eng.gemm


def spy(ta, tb, M, N, K, *args, **kw):
    calls[(ta, tb, M, N, K, kw.get('batch', 1), kw.get('H', 1), kw.get('kbatch', 1), bool(kw.get('rowsum')))] += 1
    return orig(ta, tb, M, N, K, *args, **kw)


eng.gemm = spy
task = bench.ResidentTask(mtl_amd, 0, 8, 1000, 100, bench.CFG['vocab_size'], dev)
x, lens, pct, y, ylen = task.batches[0]
model.engine  # noqa: B018 (built lazily)
model.pass_forward(x, lens, y)
model.pass_backward()
torch.cuda.synchronize()
eng.gemm = orig
L = eng.lib
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(8 << 20, device=dev)
rows = []
for (ta, tb, M, N, K, batch, H, kb, rs), n in calls.items():
    zb = batch
    A = torch.randn(zb * M * K + 64, device=dev)
    Bm = torch.randn(zb * N * K + 64, device=dev)
    C = torch.empty(zb * M * N + 64, device=dev)
    lda, ldb = (M if ta else K), (K if tb else N)
    f = lambda: L.mtl_gemm_f32(st, ta, tb, M, N, K, 1.0, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N, None, None, 0, 0,
                               batch, 1, M * K, 0, N * K, 0, M * N, 0, 0, ws.data_ptr(), ws.numel() * 4)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    rows.append((n * us, n, us, (ta, tb, M, N, K, batch, H, kb, rs)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total %.2f ms per pass over %d calls' % (tot / 1e3, sum(r[1] for r in rows)))
for t, n, us, key in rows:
    ta, tb, M, N, K, batch, H, kb, rs = key
    print('%5.1f%%  x%-3d %7.1f us  %6.1f TF  ta%d tb%d M%-5d N%-5d K%-5d batch%-3d kbatch%d rowsum%d' % (100 * t / tot, n, us, 2.0 * M * N * K * batch / us / 1e6, ta, tb, M, N, K, batch, kb, rs))
