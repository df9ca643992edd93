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
for M, N in ((808, 512), (2000, 512), (808, 128)):
    for K in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        A = torch.randn(nt, M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(nt, M, N, device=dev)
        S = 2048
        aa = A.abs().amax().reshape(1, 1).repeat(nt, S).contiguous(); ab = B.abs().amax().reshape(1).repeat(S).contiguous()
        def x3():
            assert L.mtl_gemm_f32_tb(st(), 0, 1, M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, None, None, 0, 0, nt, 1, 0, 0, 0, 0, 0, 0,
                                     0, 1, 0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0, nt, M * K, 0, M * N, 0, 0) == 0
        def h2():
            assert L.mtl_gemm_h2_tb(st(), 1, M, N, K, A.data_ptr(), K, aa.data_ptr(), S, B.data_ptr(), K, ab.data_ptr(), 0, C.data_ptr(), N, None, None, 0,
                                    nt, M * K, 0, M * N, 0, None, 0) == 0
        print('%dx%dx%-5d x8: x3 %6.1f us   h2 %6.1f us' % (M, N, K, timeit(x3), timeit(h2)))
