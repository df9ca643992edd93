"""GPU micro-benchmark of mtl_gemm_nt_h2 against mtl_gemm_f32 on the encoder input-projection shapes (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mtl_amd
L = mtl_amd._lib.lib()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, M, N, K, gate in [('fwd  e0 = p2 . wp^T', 2000, 512, 5120, False), ('dgrad dp2 = de0 . wp', 2000, 5120, 512, True),
                            ('T=5000 fwd', 10000, 512, 5120, False), ('T=5000 dgrad', 10000, 5120, 512, True), ('4096^3', 4096, 4096, 4096, False)]:
    A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
    g = torch.randn(M, N, device='cuda') if gate else None
    aa = A.abs().max().reshape(1).repeat(2048); ab = B.abs().max().reshape(1).repeat(2048)
    need = L.mtl_gemm_nt_h2_workspace(M, N, K); ws = torch.empty(need // 4 + 4, device='cuda')
    t = timeit(lambda: L.mtl_gemm_nt_h2(st, M, N, K, A.data_ptr(), K, aa.data_ptr(), B.data_ptr(), K, ab.data_ptr(), C.data_ptr(), N, None,
                                        g.data_ptr() if gate else None, N, ws.data_ptr(), need))
    ws32 = torch.empty(64 << 20, device='cuda', dtype=torch.uint8)
    t32 = timeit(lambda: L.mtl_gemm_f32(st, 0, 1, M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, None, g.data_ptr() if gate else None, N,
                                        0, 1, 1, 0, 0, 0, 0, 0, 0, 0, ws32.data_ptr(), 64 << 20))
    fl = 2.0 * M * N * K
    print('%-22s %5dx%5dx%5d  h2 %7.1f us %6.1f TF (split %d)   fp32 %7.1f us %6.1f TF' % (name, M, N, K, t, fl / t / 1e6, need // (M * N * 4) or 1, t32, fl / t32 / 1e6))
