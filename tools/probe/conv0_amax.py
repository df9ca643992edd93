import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, mtl_amd
L = mtl_amd._lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
B, T, F = 8, 1000, 161
x = torch.randn(B, 1, F, T, device='cuda'); w = torch.randn(64, 1, 3, 3, device='cuda') * 0.3; b = torch.randn(64, device='cuda')
y = torch.empty(B, T, F, 64, device='cuda'); slot = torch.zeros(2048, device='cuda')
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
print('no amax   %.1f us' % timeit(lambda: L.mtl_conv0_relu_fwd(st(), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, T, F, None)))
print('amax warm %.1f us' % timeit(lambda: L.mtl_conv0_relu_fwd(st(), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, T, F, slot.data_ptr())))
def cold():
    slot.zero_()
    L.mtl_conv0_relu_fwd(st(), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, T, F, slot.data_ptr())
print('amax cold %.1f us (incl. fill)' % timeit(cold))
dy = torch.randn(B, T, F, 64, device='cuda'); dw = torch.zeros(64, 1, 3, 3, device='cuda'); db = torch.zeros(64, device='cuda')
ws = torch.empty(L.mtl_conv0_wgrad_workspace() // 4, device='cuda')
print('wgrad     %.1f us' % timeit(lambda: L.mtl_conv0_wgrad(st(), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, T, F)))
