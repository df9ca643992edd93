import os, sys
sys.path.insert(0, '/root/repo')
import torch
import mtl_amd
L = mtl_amd._lib.lib()
dev = torch.device('cuda')
st = lambda: torch.cuda.current_stream().cuda_stream
ws = torch.empty(8 << 20, device=dev)
nt = 8
def timeit(fn, reps=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
M, N = 808, 512
for K in (512, 2048):
    for pad in (0, 4, 16, 32, 64, 96):
        lda = K + pad
        A = torch.randn(nt, M, lda, device=dev); B = torch.randn(N, lda, device=dev); C = torch.empty(nt, M, N, device=dev)
        def x3():
            assert L.mtl_gemm_f32_tb(st(), 0, 1, M, N, K, 1.0, A.data_ptr(), lda, B.data_ptr(), lda, C.data_ptr(), N, None, None, 0, 0, nt, 1, 0, 0, 0, 0, 0, 0,
                                     0, 1, 0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0, nt, M * lda, 0, M * N, 0, 0) == 0
        print('808x512x%-5d x8 lda = K + %-3d: x3 %6.1f us' % (K, pad, timeit(x3)))
