"""Scan of the small-product engine's tile / K-group choices on the MAIN-STREAM products of the north-star pass: one process per
forced configuration (the knobs MTL_G16_TILE / MTL_G16_KG are read once), default dispatch first.  GPU only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    import mtl_amd
    L = mtl_amd._lib.lib()
    dev = torch.device('cuda')
    shapes = [(0, 1, 808, 100, 512, 1), (0, 1, 808, 512, 100, 1), (0, 1, 808, 100, 512, 3), (0, 1, 808, 512, 100, 3), (0, 1, 808, 512, 512, 1),
              (0, 1, 2000, 100, 512, 3), (0, 1, 2000, 512, 100, 3), (0, 1, 2000, 512, 512, 1), (0, 1, 2000, 100, 512, 1), (0, 1, 2000, 512, 100, 1),
              (0, 0, 808, 100, 512, 1), (0, 0, 808, 512, 100, 1), (0, 0, 808, 100, 512, 3), (0, 0, 808, 512, 512, 1), (0, 0, 2000, 100, 512, 3),
              (0, 0, 2000, 512, 512, 1), (0, 0, 2000, 512, 100, 1)]
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for ta, tb, M, N, K, nb in shapes:
        A = torch.randn(nb, M, K, device=dev)
        B = torch.randn(nb, N, K, device=dev) if tb else torch.randn(nb, K, N, device=dev)
        C = torch.zeros(nb, M, N, device=dev)
        lda, ldb = K, B.shape[2]

        def run():
            assert L.mtl_gemm_f32_ex(st, ta, tb, M, N, K, 1.0, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), N, None, None, 0, 0, nb, 1,
                                     A[0].numel(), 0, B[0].numel(), 0, M * N, 0, 0, 1, 0, 0, None, 0, None, 0, 0, 0) == 0
        for _ in range(5):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            run()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / 100)
    print(' '.join('%6.1f' % v for v in out))
    sys.exit(0)
print('columns: tb1 808x100x512 | 808x512x100 | b3 808x100x512 | b3 808x512x100 | 808x512x512 | b3 2000x100x512 | b3 2000x512x100 | 2000x512x512 | '
      '2000x100x512 | 2000x512x100 || tb0 808x100x512 | 808x512x100 | b3 808x100x512 | 808x512x512 | b3 2000x100x512 | 2000x512x512 | 2000x512x100')
for tile, kg in [(0, 0)] + [(t, k) for t in (1, 2, 3) for k in (1, 2, 4)]:
    env = dict(os.environ)
    if tile:
        env['MTL_G16_TILE'], env['MTL_G16_KG'] = str(tile), str(kg)
    r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
    print('tile %d kg %d: %s' % (tile, kg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]))
