import os, sys
sys.path.insert(0, '/root/repo')
import torch, mtl_amd
L = mtl_amd._lib.lib(); dev = torch.device('cuda'); st = torch.cuda.current_stream().cuda_stream
def bench(name, ta, tb, M, N, K, nb, pada, padb, fl=0):
    sa = (K, M + pada) if ta else (M, K + pada); sb = (N, K + padb) if tb else (K, N + padb)
    A = torch.randn(nb, *sa, device=dev); B = torch.randn(nb, *sb, device=dev); C = torch.zeros(nb, M, N, device=dev)
    def run():
        assert L.mtl_gemm_f32_ex(st, ta, tb, M, N, K, 1.0, A.data_ptr(), sa[1], B.data_ptr(), sb[1], C.data_ptr(), N, None, None, 0, fl, nb, 1,
                                 A[0].numel(), 0, B[0].numel(), 0, M * N, 0, 0, 1, 0, 0, None, 0, None, 0, 0, 0) == 0
    for _ in range(5): run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): run()
    b.record(); torch.cuda.synchronize()
    print('%-14s pad A %2d B %2d: %6.1f us' % (name, pada, padb, a.elapsed_time(b) * 1e3 / 50))

for fl in (0, 256, 512, 768, 2):
    for name, ta, tb, M, N, K, nb in (('a-stage enc', 0, 1, 2000, 100, 512, 3), ('b-stage enc', 0, 1, 2000, 512, 100, 3), ('dW ffn enc', 1, 0, 512, 512, 2000, 1), ('dB enc', 1, 0, 512, 100, 2000, 3),
                                      ('ffn1 dec', 0, 1, 808, 512, 512, 1), ('tiny', 0, 1, 64, 64, 64, 1)):
        print('flags', fl, end=' ')
        bench(name, ta, tb, M, N, K, nb, 0, 0, fl)
