"""Tile / K-group choices of the small-product fp32 engine at the 8-task shapes that stay on it (MTL_G16_TILE / MTL_G16_KG are read once per process:
run once per setting).  usage: MTL_G16_TILE=2 python tools/probe/bench_g16_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mtl_amd
L = mtl_amd._lib.lib()
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream
L.mtl_gemm_x3_min_tiles(0)          # keep everything on the fp32 engines
for name, ta, tb, M, N, K, nb in (('a-stage dec x8', 0, 1, 808, 100, 512, 8), ('dX b-stage dec x8', 0, 0, 808, 100, 512, 8), ('a-stage enc x8', 0, 1, 2000, 100, 512, 8),
                                  ('b-stage dec x8', 0, 1, 808, 512, 100, 8), ('a-stage dec x1', 0, 1, 808, 100, 512, 1), ('ffn dec x1', 0, 1, 808, 512, 512, 1),
                                  ('ffn enc x1', 0, 1, 2000, 512, 512, 1), ('b-stage dec x1', 0, 1, 808, 512, 100, 1)):
    A = torch.randn(nb, K, M, device=dev) if ta else torch.randn(nb, M, K, device=dev)
    B = torch.randn(nb, N, K, device=dev) if tb else torch.randn(nb, K, N, device=dev)
    C = torch.zeros(nb, M, N, device=dev)
    lda, ldb = A.shape[2], B.shape[2]
    def run():
        assert L.mtl_gemm_f32_ex(st, ta, tb, M, N, K, 1.0, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), N, None, None, 0, 0, nb, 1,
                                 A[0].numel(), 0, B[0].numel(), 0, M * N, 0, 0, 1, 0, 0, None, M, None, 0, 0, 0) == 0
    for _ in range(5): run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): run()
    b.record(); torch.cuda.synchronize()
    print('%-20s %7.1f us' % (name, a.elapsed_time(b) * 1e3 / 50))
