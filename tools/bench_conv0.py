"""conv0 forward / weight-gradient (C_in = 1, 330 MB streams at the north-star size): isolated timings.  GPU only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtl_amd  # noqa: E402

L = mtl_amd._lib.lib()
B, T, F = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 161
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, 1, F, T).cuda()
w, b = (torch.randn(64, 1, 3, 3) * 0.3).cuda(), torch.randn(64).cuda()
y = torch.empty(B, T, F, 64).cuda()
amax = torch.zeros(2048).cuda()
dy = torch.randn(B, T, F, 64).cuda()
wg, bg = torch.zeros(64, 1, 3, 3).cuda(), torch.zeros(64).cuda()
ws = torch.empty(L.mtl_conv0_wgrad_workspace() // 4).cuda()
nbytes = y.numel() * 4


def run(name, f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print('%-12s %7.1f us  %5.2f TB/s' % (name, us, nbytes / us / 1e6))


run('conv0_fwd', lambda: L.mtl_conv0_relu_fwd(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, T, F, amax.data_ptr()))
run('conv0_wgrad', lambda: L.mtl_conv0_wgrad(st, x.data_ptr(), dy.data_ptr(), wg.data_ptr(), bg.data_ptr(), ws.data_ptr(), B, T, F))
