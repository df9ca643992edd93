import os, sys
sys.path.insert(0, '/root/repo')
import torch
import mtl_amd
L = mtl_amd._lib.lib()
dev = torch.device('cuda')
st = lambda: torch.cuda.current_stream().cuda_stream
ws = torch.empty(8 << 20, device=dev)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
L.mtl_gemm_x3_min_tiles(1)
K = 4096
# (a) w workgroups of 128 x 128, ONE shared B; (b) each workgroup its own B (batch items)
for w in ((8, 128) if os.environ.get('MTL_LIB') else (8, 16, 32, 64, 128, 192, 224, 256)):
    A = torch.randn(w, 128, K, device=dev); B1 = torch.randn(128, K, device=dev); Bw = torch.randn(w, 128, K, device=dev); C = torch.empty(w, 128, 128, device=dev)
    def shared():
        assert L.mtl_gemm_f32_tb(st(), 0, 1, 128, 128, K, 1.0, A.data_ptr(), K, B1.data_ptr(), K, C.data_ptr(), 128, None, None, 0, 0, w, 1, 128 * K, 0, 0, 0, 128 * 128, 0,
                                 0, 1, 0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0, 1, 0, 0, 0, 0, 0) == 0
    def own():
        assert L.mtl_gemm_f32_tb(st(), 0, 1, 128, 128, K, 1.0, A.data_ptr(), K, Bw.data_ptr(), K, C.data_ptr(), 128, None, None, 0, 0, w, 1, 128 * K, 0, 128 * K, 0, 128 * 128, 0,
                                 0, 1, 0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0, 1, 0, 0, 0, 0, 0) == 0
    ts, to = timeit(shared), timeit(own)
    print('%3d workgroups x (128 x 128 x %d): shared B %6.1f us (%.2f us / step)   own B %6.1f us (%.2f us / step)' % (w, K, ts, ts / 128, to, to / 128))
