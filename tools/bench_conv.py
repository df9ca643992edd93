"""GPU micro-benchmark of the conv / GEMM entry points at the north-star shapes (HIP events, isolated launches).
usage: python tools/bench_conv.py [path/to/libmtl_hip.so]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mtl_amd
from mtl_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
    _lib._lib = None
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
B, T, F = 8, 1000, 161
dev = 'cuda'

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

ZERO = os.environ.get('MTL_BENCH_ZERO') == '1'      # all-zero operands: same instruction stream, less switching power (DVFS probe)

def conv_case(name, T_, F_, cin, cout, pooled):
    x = torch.relu(torch.randn(B, T_, F_, cin, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    if ZERO:
        x.zero_(); w.zero_()
    bias = torch.randn(cout, device=dev) * 0.1
    wf, wd = torch.empty(9, cin, cout, device=dev), torch.empty(9, cout, cin, device=dev)
    L.mtl_conv3x3_wprep(st(), w.data_ptr(), wf.data_ptr(), wd.data_ptr(), cout, cin)
    flops = 2.0 * B * T_ * F_ * 9 * cin * cout
    Tp, Fp = T_ // 2, F_ // 2
    if pooled:
        y = torch.empty(B, Tp, Fp, cout, device=dev); am = torch.empty(B, Tp, Fp, cout, dtype=torch.uint8, device=dev)
        t = timeit(lambda: L.mtl_conv3x3_relu_pool_fwd(st(), x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), am.data_ptr(), B, T_, F_, cin, cout))
        dy = torch.randn_like(y); amp = am.data_ptr()
        if ZERO: dy.zero_()
    else:
        y = torch.empty(B, T_, F_, cout, device=dev)
        t = timeit(lambda: L.mtl_conv3x3_relu_fwd(st(), x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), B, T_, F_, cin, cout))
        dy = torch.randn_like(y); amp = None
        if ZERO: dy.zero_()
    print('%-8s fwd   %.3f ms %6.1f TF' % (name, t, flops / t / 1e9))
    dx = torch.empty_like(x)
    t = timeit(lambda: L.mtl_conv3x3_dgrad(st(), dy.data_ptr(), amp, wd.data_ptr(), x.data_ptr(), dx.data_ptr(), B, T_, F_, cin, cout))
    print('%-8s dgrad %.3f ms %6.1f TF' % (name, t, flops / t / 1e9))
    w3f = torch.empty(3 * 9 * cin * cout, dtype=torch.bfloat16, device=dev); w3d = torch.empty_like(w3f)
    L.mtl_conv3x3_wprep_x3(st(), w.data_ptr(), w3f.data_ptr(), w3d.data_ptr(), cout, cin)
    if pooled:
        t = timeit(lambda: L.mtl_conv3x3_relu_pool_fwd_x3(st(), x.data_ptr(), w3f.data_ptr(), bias.data_ptr(), y.data_ptr(), am.data_ptr(), B, T_, F_, cin, cout))
    else:
        t = timeit(lambda: L.mtl_conv3x3_relu_fwd_x3(st(), x.data_ptr(), w3f.data_ptr(), bias.data_ptr(), y.data_ptr(), B, T_, F_, cin, cout))
    print('%-8s fwd   x3 %.3f ms %6.1f TF' % (name, t, flops / t / 1e9))
    t = timeit(lambda: L.mtl_conv3x3_dgrad_x3(st(), dy.data_ptr(), amp, w3d.data_ptr(), x.data_ptr(), dx.data_ptr(), B, T_, F_, cin, cout))
    print('%-8s dgrad x3 %.3f ms %6.1f TF' % (name, t, flops / t / 1e9))
    nb = L.mtl_conv3x3_wprep_h2_bytes(cout, cin)
    w2f = torch.empty(nb, dtype=torch.uint8, device=dev); w2d = torch.empty_like(w2f)
    L.mtl_conv3x3_wprep_h2(st(), w.data_ptr(), w2f.data_ptr(), w2d.data_ptr(), cout, cin)
    ax, ady = x.abs().max().reshape(1).repeat(2048), dy.abs().max().reshape(1).repeat(2048)
    slot = torch.zeros(2048, device=dev)
    if pooled:
        t = timeit(lambda: L.mtl_conv3x3_relu_pool_fwd_h2(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), am.data_ptr(), slot.data_ptr(), B, T_, F_, cin, cout))
    else:
        t = timeit(lambda: L.mtl_conv3x3_relu_fwd_h2(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), slot.data_ptr(), B, T_, F_, cin, cout))
    print('%-8s fwd   h2 %.3f ms %6.1f TF' % (name, t, flops / t / 1e9))
    t = timeit(lambda: L.mtl_conv3x3_dgrad_h2(st(), dy.data_ptr(), ady.data_ptr(), amp, w2d.data_ptr(), x.data_ptr(), dx.data_ptr(), None, B, T_, F_, cin, cout))
    print('%-8s dgrad h2 %.3f ms %6.1f TF' % (name, t, flops / t / 1e9))
    need = L.mtl_conv3x3_wgrad_workspace(B, T_, F_, cin, cout, 1 if pooled else 0)
    ws = torch.empty(need // 4 + 64, device=dev); dw = torch.zeros_like(w)
    t = timeit(lambda: L.mtl_conv3x3_wgrad(st(), x.data_ptr(), dy.data_ptr(), amp, dw.data_ptr(), ws.data_ptr(), need, B, T_, F_, cin, cout))
    print('%-8s wgrad %.3f ms %6.1f TF' % (name, t, flops / t / 1e9))
    need = L.mtl_conv3x3_wgrad_x3_workspace(B, T_, F_, cin, cout, 1 if pooled else 0)
    ws = torch.empty(need // 4 + 64, device=dev)
    t = timeit(lambda: L.mtl_conv3x3_wgrad_x3(st(), x.data_ptr(), dy.data_ptr(), amp, dw.data_ptr(), ws.data_ptr(), need, B, T_, F_, cin, cout))
    print('%-8s wgrad x3 %.3f ms %6.1f TF' % (name, t, flops / t / 1e9))
    dbias = torch.zeros(cout, device=dev)
    t = timeit(lambda: L.mtl_conv3x3_wgrad_h2(st(), x.data_ptr(), ax.data_ptr(), dy.data_ptr(), ady.data_ptr(), amp, dw.data_ptr(), dbias.data_ptr(), ws.data_ptr(), need, B, T_, F_, cin, cout))
    print('%-8s wgrad h2 %.3f ms %6.1f TF' % (name, t, flops / t / 1e9))

conv_case('conv2', T, F, 64, 64, True)
conv_case('conv5', T // 2, F // 2, 64, 128, False)
conv_case('conv7', T // 2, F // 2, 128, 128, True)
for (M, N, K, ta, tb) in ((2000, 512, 5120, 0, 1), (2000, 5120, 512, 0, 0), (512, 5120, 2000, 1, 0), (4096, 4096, 4096, 0, 1), (808, 3765, 512, 0, 1)):
    A = torch.randn((K, M) if ta else (M, K), device=dev); Bm = torch.randn((N, K) if tb else (K, N), device=dev); C = torch.empty(M, N, device=dev)
    ws = torch.empty(8 << 20, device=dev)
    t = timeit(lambda: L.mtl_gemm_f32(st(), ta, tb, M, N, K, 1.0, A.data_ptr(), A.shape[1], Bm.data_ptr(), Bm.shape[1], C.data_ptr(), N, None, None, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, ws.data_ptr(), ws.numel() * 4))
    print('gemm %dx%dx%d ta%d tb%d  %.3f ms %6.1f TF' % (M, N, K, ta, tb, t, 2.0 * M * N * K / t / 1e9))
