"""conv0 forward / weight gradient and the bias column sum: 8 launches over one task's samples against one launch over all of them
(the bound of merging these per-task launches of a task-batched pass).  usage: python tools/probe/small_batch_tasks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mtl_amd
from mtl_amd import _lib
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
dev = 'cuda'


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


nt, B, T, F = 8, 8, 1000, 161
x = torch.randn(nt * B, 1, F, T, device=dev)
w = torch.randn(64, 1, 3, 3, device=dev); b = torch.randn(64, device=dev)
y = torch.empty(nt * B, T, F, 64, device=dev)
am = torch.zeros(nt, 2048, device=dev)
f8 = lambda: [L.mtl_conv0_relu_fwd(st(), x[t * B:].data_ptr(), w.data_ptr(), b.data_ptr(), y[t * B:].data_ptr(), B, T, F, am[t].data_ptr()) for t in range(nt)]
f1 = lambda: L.mtl_conv0_relu_fwd(st(), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), nt * B, T, F, am.data_ptr())
print('conv0 forward:  8 launches %.0f us | one launch %.0f us' % (timeit(f8), timeit(f1)))
dy = torch.randn_like(y); dw = torch.zeros(64, 9, device=dev); db = torch.zeros(64, device=dev)
ws = torch.empty(L.mtl_conv0_wgrad_workspace() // 4, device=dev)
g8 = lambda: [L.mtl_conv0_wgrad(st(), x[t * B:].data_ptr(), dy[t * B:].data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, T, F) for t in range(nt)]
g1 = lambda: L.mtl_conv0_wgrad(st(), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nt * B, T, F)
print('conv0 wgrad:    8 launches %.0f us | one launch %.0f us' % (timeit(g8), timeit(g1)))
rows = B * 250 * 40
X = torch.randn(nt * rows, 128, device=dev); out = torch.zeros(128, device=dev)
cw = torch.empty(L.mtl_colsum_workspace(nt * rows, 128) // 4 + 64, device=dev)
c8 = lambda: [L.mtl_colsum_accum(st(), X[t * rows:].data_ptr(), rows, 128, 128, out.data_ptr(), cw.data_ptr(), am[t].data_ptr()) for t in range(nt)]
c1 = lambda: L.mtl_colsum_accum(st(), X.data_ptr(), nt * rows, 128, 128, out.data_ptr(), cw.data_ptr(), am.data_ptr())
print('bias colsum:    8 launches %.0f us | one launch %.0f us' % (timeit(c8), timeit(c1)))
