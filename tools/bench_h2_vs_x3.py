"""GPU micro-benchmark: the tile engine of mtl_gemm_x3.hip on exact bf16 triples (mtl_gemm_f32_tb) vs on fp16 pairs (mtl_gemm_h2_tb), at the
product shapes of an 8-task batched pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for name, tb, M, N, K in (('a-stage dec', 1, 808, 100, 512), ('b-stage dec', 1, 808, 512, 100), ('ffn dec', 1, 808, 512, 512), ('ffn enc', 1, 2000, 512, 512),
                          ('dX a dec', 0, 808, 512, 100), ('dX b dec', 0, 808, 100, 512), ('vocab', 1, 808, 3768, 512), ('vocab dX', 0, 808, 512, 3768)):
    A = torch.randn(nt, M, K, device=dev); B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
    C = torch.empty(nt, M, N, device=dev)
    S = 2048
    aa = A.abs().amax().reshape(1, 1).repeat(nt, S).contiguous(); ab = B.abs().amax().reshape(1).repeat(S).contiguous()
    ldb = K if tb else N
    def x3():
        assert L.mtl_gemm_f32_tb(st(), 0, tb, M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), ldb, C.data_ptr(), N, None, None, 0, 0, nt, 1, 0, 0, 0, 0, 0, 0,
                                 0, 1, 0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0, nt, M * K, 0, M * N, 0, 0) == 0
    def h2():
        assert L.mtl_gemm_h2_tb(st(), tb, M, N, K, A.data_ptr(), K, aa.data_ptr(), S, B.data_ptr(), ldb, ab.data_ptr(), 0, C.data_ptr(), N, None, None, 0,
                                nt, M * K, 0, M * N, 0, None, 0) == 0
    x3(); Cx = C.clone(); h2()
    print('%-12s %dx%dx%d x%d: x3 %6.1f us   h2 %6.1f us   rel diff %.1e' % (name, M, N, K, nt, timeit(x3), timeit(h2), float((C - Cx).norm() / Cx.norm())))
