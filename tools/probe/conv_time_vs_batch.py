"""Fixed cost of a persistent convolution launch: h2 forward / data-gradient time against the number of samples (a x B + c).
usage: python tools/probe/conv_time_vs_batch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mtl_amd
from mtl_amd import _lib
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
dev = 'cuda'


def timeit(fn, reps=8):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def case(name, T_, F_, cin, cout, pooled):
    BM = 64
    x = torch.relu(torch.randn(BM, T_, F_, cin, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev) * 0.1
    nb = L.mtl_conv3x3_wprep_h2_bytes(cout, cin)
    w2f = torch.empty(nb, dtype=torch.uint8, device=dev); w2d = torch.empty_like(w2f)
    L.mtl_conv3x3_wprep_h2(st(), w.data_ptr(), w2f.data_ptr(), w2d.data_ptr(), cout, cin)
    ax = x.abs().max().reshape(1).repeat(2048); slot = torch.zeros(2048, device=dev)
    Tp, Fp = T_ // 2, F_ // 2
    shp = (BM, Tp, Fp, cout) if pooled else (BM, T_, F_, cout)
    y = torch.empty(shp, device=dev); am = torch.empty(shp, dtype=torch.uint8, device=dev)
    dy = torch.randn(shp, device=dev); ady = dy.abs().max().reshape(1).repeat(2048); dx = torch.empty_like(x)
    rows = []
    for b in (1, 2, 4, 8, 16, 32, 64):
        if pooled:
            f = lambda: L.mtl_conv3x3_relu_pool_fwd_h2(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), am.data_ptr(), slot.data_ptr(), b, T_, F_, cin, cout)
        else:
            f = lambda: L.mtl_conv3x3_relu_fwd_h2(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), slot.data_ptr(), b, T_, F_, cin, cout)
        d = lambda: L.mtl_conv3x3_dgrad_h2(st(), dy.data_ptr(), ady.data_ptr(), am.data_ptr() if pooled else None, w2d.data_ptr(), x.data_ptr(), dx.data_ptr(), None, b, T_, F_, cin, cout)
        rows.append((b, timeit(f), timeit(d)))
    print(name, ' '.join('B=%d: %.0f / %.0f us (%.1f / %.1f per sample)' % (b, tf, td, tf / b, td / b) for b, tf, td in rows))


case('conv2', 1000, 161, 64, 64, True)
case('conv5', 500, 80, 64, 128, False)
case('conv7', 500, 80, 128, 128, True)
