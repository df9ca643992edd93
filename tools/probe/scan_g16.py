"""Scan of the small-product engine's tile / K-group choice on the products of a ONE-task pass (what a rank of the 8-GPU configuration
runs: the chain of ~120 dependent products per pass sets its step time).  Needs the probe build:
    MTL_LIB=tools/probe/libmtl_g16probe.so python tools/probe/scan_g16.py
Prints, per shape, the time of the default dispatch and of every forced (tile, K groups) pair, best first."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import mtl_amd  # noqa: E402

L = mtl_amd._lib.lib()
raw = ctypes.CDLL(mtl_amd._lib.LIB_PATH)
force = raw.mtl_probe_g16_force
force.argtypes = [ctypes.c_int, ctypes.c_int]
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream
# (ta, tb, M, N, K, batch, kbatch): main-stream products of a one-task pass (profiles/r6/one_task/gemm_shapes_1task.txt)
shapes = [(0, 1, 808, 100, 512, 1, 1), (0, 0, 808, 100, 512, 1, 1), (0, 0, 808, 512, 512, 1, 1), (0, 1, 808, 512, 512, 1, 1),
          (0, 1, 808, 512, 100, 1, 1), (0, 0, 808, 512, 100, 1, 1), (0, 0, 2000, 512, 512, 1, 1), (0, 1, 2000, 512, 512, 1, 1),
          (0, 0, 808, 100, 512, 3, 1), (0, 1, 808, 100, 512, 3, 1), (0, 1, 808, 512, 100, 3, 1), (0, 0, 808, 512, 100, 1, 3),
          (0, 0, 2000, 512, 100, 1, 4), (0, 0, 2000, 100, 512, 3, 1), (0, 1, 2000, 100, 512, 3, 1), (0, 1, 2000, 512, 100, 3, 1),
          (0, 0, 2000, 100, 512, 8, 1), (0, 1, 2000, 512, 100, 8, 1)]
ws = torch.empty(8 << 20, device=dev)
for ta, tb, M, N, K, nb, kb in shapes:
    z = nb * kb
    A = torch.randn(z, M, K, device=dev)
    B = torch.randn(z, N, K, device=dev) if tb else torch.randn(z, K, N, device=dev)
    C = torch.zeros(nb, M, N, device=dev)
    lda, ldb = K, (K if tb else N)

    def run():
        assert L.mtl_gemm_f32_ex(st, ta, tb, M, N, K, 1.0, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), N, None, None, 0, 0, nb, 1,
                                 kb * M * K, 0, kb * B[0].numel(), 0, M * N, 0, 0, kb, M * K, B[0].numel(), None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0) == 0
    res = []
    for tile, kg in [(0, 0)] + [(t, k) for t in (1, 2, 3) for k in (1, 2, 4)]:
        force(tile, kg)
        for _ in range(5):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            run()
        b.record()
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b) * 1e3 / 200, tile, kg))
    force(0, 0)
    d = res[0][0]
    best = sorted(res[1:])[:3]
    print('ta%d tb%d M%-4d N%-4d K%-4d b%d kb%d  default %6.2f us | best %s' % (
        ta, tb, M, N, K, nb, kb, d, '  '.join('t%d k%d %.2f (%.0f%%)' % (t, k, us, 100 * us / d) for us, t, k in best)), flush=True)
