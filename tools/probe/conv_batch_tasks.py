"""Would ONE convolution launch over the samples of all tasks beat one launch per task?  (The engine issues the 3x3 convolutions per task:
per-task h2 bounds.)  h2 forward kernels of the three layers, 8 launches of B = 8 back to back against one launch of B = 64, same data
volume; per-launch prologue / tail / launch boundary is the difference.   usage: python tools/probe/conv_batch_tasks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mtl_amd
from mtl_amd import _lib
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
dev = 'cuda'


def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def case(name, T_, F_, cin, cout, pooled, nt=8, B=8):
    x = torch.relu(torch.randn(nt * B, T_, F_, cin, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev) * 0.1
    nb = L.mtl_conv3x3_wprep_h2_bytes(cout, cin)
    w2f = torch.empty(nb, dtype=torch.uint8, device=dev); w2d = torch.empty_like(w2f)
    L.mtl_conv3x3_wprep_h2(st(), w.data_ptr(), w2f.data_ptr(), w2d.data_ptr(), cout, cin)
    ax = x.abs().max().reshape(1).repeat(2048)
    slot = torch.zeros(2048, device=dev)
    Tp, Fp = T_ // 2, F_ // 2
    if pooled:
        y = torch.empty(nt * B, Tp, Fp, cout, device=dev); am = torch.empty(nt * B, Tp, Fp, cout, dtype=torch.uint8, device=dev)
        f = lambda xs, ys, ams, b: L.mtl_conv3x3_relu_pool_fwd_h2(st(), xs.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), ys.data_ptr(), ams.data_ptr(), slot.data_ptr(), b, T_, F_, cin, cout)
    else:
        y = torch.empty(nt * B, T_, F_, cout, device=dev); am = y
        f = lambda xs, ys, ams, b: L.mtl_conv3x3_relu_fwd_h2(st(), xs.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), ys.data_ptr(), slot.data_ptr(), b, T_, F_, cin, cout)
    def per_task():
        for t in range(nt):
            assert f(x[t * B:], y[t * B:], am[t * B:], B) == 0
    def one():
        assert f(x, y, am, nt * B) == 0
    dy = torch.randn_like(y)
    ady = dy.abs().max().reshape(1).repeat(2048)
    dx = torch.empty_like(x)
    amp = (lambda t: am[t * B:].data_ptr()) if pooled else (lambda t: None)
    def d_per_task():
        for t in range(nt):
            assert L.mtl_conv3x3_dgrad_h2(st(), dy[t * B:].data_ptr(), ady.data_ptr(), amp(t), w2d.data_ptr(), x[t * B:].data_ptr(), dx[t * B:].data_ptr(), None, B, T_, F_, cin, cout) == 0
    def d_one():
        assert L.mtl_conv3x3_dgrad_h2(st(), dy.data_ptr(), ady.data_ptr(), amp(0), w2d.data_ptr(), x.data_ptr(), dx.data_ptr(), None, nt * B, T_, F_, cin, cout) == 0
    axs = ax.repeat(nt, 1).contiguous(); adys = ady.repeat(nt, 1).contiguous(); slots = torch.zeros(nt, 2048, device=dev)
    def tb():
        if pooled:
            assert L.mtl_conv3x3_relu_pool_fwd_h2_tb(st(), x.data_ptr(), axs.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), am.data_ptr(), slots.data_ptr(), B, T_, F_, cin, cout, nt, 0, 0, 2048, 2048, None, 0) == 0
        else:
            assert L.mtl_conv3x3_relu_fwd_h2_tb(st(), x.data_ptr(), axs.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), slots.data_ptr(), B, T_, F_, cin, cout, nt, 0, 0, 2048, 2048, None, 0) == 0
    def d_tb():
        assert L.mtl_conv3x3_dgrad_h2_tb(st(), dy.data_ptr(), adys.data_ptr(), amp(0), w2d.data_ptr(), x.data_ptr(), dx.data_ptr(), slots.data_ptr(), B, T_, F_, cin, cout, nt, 0, 2048, 2048, None, 0) == 0
    a, b, c, d, e, f2 = timeit(per_task), timeit(one), timeit(d_per_task), timeit(d_one), timeit(tb), timeit(d_tb)
    print('%-6s forward: %d launches of B=%d %.3f ms | one launch of B=%d %.3f ms (%+.1f %%) | %d tasks in one launch %.3f ms    data gradient: %.3f | %.3f ms (%+.1f %%) | %.3f ms'
          % (name, nt, B, a, nt * B, b, 100 * (b - a) / a, nt, e, c, d, 100 * (d - c) / c, f2))


case('conv2', 1000, 161, 64, 64, True)
case('conv5', 500, 80, 64, 128, False)
case('conv7', 500, 80, 128, 128, True)


def wcase(name, T_, F_, cin, cout, pooled, nt=8, B=8):
    """weight gradients: nt launches over B samples against ONE over nt B samples (same work, one dW instead of nt: the bound of what
    merging the tasks of a pass into one launch could save)"""
    x = torch.relu(torch.randn(nt * B, T_, F_, cin, device=dev))
    Tp, Fp = T_ // 2, F_ // 2
    shp = (nt * B, Tp, Fp, cout) if pooled else (nt * B, T_, F_, cout)
    dy = torch.randn(shp, device=dev)
    am = torch.randint(0, 4, shp, dtype=torch.uint8, device=dev)
    ax, ady = x.abs().max().reshape(1).repeat(2048), dy.abs().max().reshape(1).repeat(2048)
    dw = torch.zeros(cout, cin, 3, 3, device=dev); db = torch.zeros(cout, device=dev)
    def run(b, xs, dys, ams):
        need = L.mtl_conv3x3_wgrad_x3_workspace(b, T_, F_, cin, cout, 1 if pooled else 0)
        ws = torch.empty(need // 4 + 64, device=dev)
        return lambda: L.mtl_conv3x3_wgrad_h2(st(), xs.data_ptr(), ax.data_ptr(), dys.data_ptr(), ady.data_ptr(), ams.data_ptr() if pooled else None, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), need, b, T_, F_, cin, cout)
    fns = [run(B, x[t * B:], dy[t * B:], am[t * B:]) for t in range(nt)]
    def per_task():
        for f in fns:
            assert f() == 0
    one = run(nt * B, x, dy, am)
    a, b = timeit(per_task), timeit(lambda: one())
    print('%-6s weight gradient: %d launches of B=%d %.3f ms | one launch of B=%d %.3f ms (%+.1f %%)' % (name, nt, B, a, nt * B, b, 100 * (b - a) / a))


wcase('conv2', 1000, 161, 64, 64, True)
wcase('conv5', 500, 80, 64, 128, False)
wcase('conv7', 500, 80, 128, 128, True)
