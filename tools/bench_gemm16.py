"""GPU micro-benchmark of mtl_gemm_f32_ex on the small products of the north-star pass (HIP events, 50 reps each)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mtl_amd
L = mtl_amd._lib.lib()
dev = torch.device('cuda')
# (name, ta, tb, M, N, K, batch, rowsum)
shapes = [('dB enc  dy^T a', 1, 0, 512, 100, 2000, 3, 1), ('dA enc  da^T x', 1, 0, 100, 512, 2000, 3, 0), ('dW ffn enc', 1, 0, 512, 512, 2000, 1, 1),
          ('dB dec', 1, 0, 512, 100, 808, 3, 1), ('dW ffn dec', 1, 0, 512, 512, 808, 1, 1),
          ('a-stage enc', 0, 1, 2000, 100, 512, 3, 0), ('b-stage enc', 0, 1, 2000, 512, 100, 3, 0), ('ffn1 dec', 0, 1, 808, 512, 512, 1, 0),
          ('dX b-stage dec', 0, 0, 808, 100, 512, 3, 0), ('vocab dX', 0, 0, 808, 512, 3768, 1, 0),
          ('a-stage dec', 0, 1, 808, 100, 512, 1, 0), ('b-stage dec', 0, 1, 808, 512, 100, 1, 0), ('ffn dX dec', 0, 0, 808, 512, 512, 1, 0),
          ('ffn1 enc', 0, 1, 2000, 512, 512, 1, 0), ('da dec', 0, 0, 808, 100, 512, 1, 0), ('dx a-stage dec', 0, 0, 808, 512, 100, 1, 0)]
st = torch.cuda.current_stream().cuda_stream
for name, ta, tb, M, N, K, nb, rsum in shapes:
    A = torch.randn(nb, K, M, device=dev) if ta else torch.randn(nb, M, K, device=dev)
    B = torch.randn(nb, N, K, device=dev) if tb else torch.randn(nb, K, N, device=dev)
    C = torch.zeros(nb, M, N, device=dev)
    rs = torch.zeros(nb, M, device=dev)
    lda, ldb = A.shape[2], B.shape[2]
    def run():
        assert L.mtl_gemm_f32_ex(st, ta, tb, M, N, K, 1.0, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), N, None, None, 0, 2, nb, 1,
                                 A[0].numel(), 0, B[0].numel(), 0, M * N, 0, 0, 1, 0, 0, rs.data_ptr() if rsum else None, M, None, 0, 0, 0) == 0
    for _ in range(5):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    print('%-16s ta%d tb%d %5dx%4dx%5d x%d: %7.1f us  %6.1f TF' % (name, ta, tb, M, N, K, nb, us, 2.0 * M * N * K * nb / us / 1e6))
