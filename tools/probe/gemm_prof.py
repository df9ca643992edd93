"""Per-wave phase breakdown of the bf16-split tile engine's K step from s_memtime stamps inside the kernel (probe build only:
hipcc -DMTL_X3G_PROF mtl_gemm_x3.hip, linked as tools/probe/libmtl_gprof.so; tools/probe/build_probes.sh).
usage: MTL_LIB=tools/probe/libmtl_gprof.so python tools/probe/gemm_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mtl_amd
from mtl_amd import _lib
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.mtl_x3g_prof_set.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda')
st = lambda: torch.cuda.current_stream().cuda_stream
ws = torch.empty(8 << 20, device=dev)
NWG = 1024
prof = torch.zeros(NWG * 8 * 8, dtype=torch.int64, device=dev)

def case(name, M, N, K, tasks, tb=1):
    A = torch.randn(tasks, M, K, device=dev); B = torch.randn(tasks, N, K, device=dev) if tb else torch.randn(tasks, K, N, device=dev)
    C = torch.empty(tasks, M, N, device=dev)
    def run():
        assert L.mtl_gemm_f32_tb(st(), 0, tb, M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K if tb else N, C.data_ptr(), N, None, None, 0, 0, tasks, 1,
                                 M * K, 0, N * K, 0, M * N, 0, 0, 1, 0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0, 1, 0, 0, 0, 0, 0) == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    prof.zero_()
    raw.mtl_x3g_prof_set(prof.data_ptr())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record()
    torch.cuda.synchronize()
    raw.mtl_x3g_prof_set(None)
    p = prof.view(NWG, 8, 8).cpu().double()
    p = p[p[:, :, 4].sum(dim=1) > 0]
    nsteps = (K + 31) // 32 - 3
    for half, sl in (('waves 0-3', slice(0, 4)), ('waves 4-7', slice(4, 8))):
        m = p[:, sl].mean(dim=(0, 1))
        tot = m[4]
        print('%-22s %s: %.1f us, %d workgroups, main-loop step = %.0f ticks | fetch issue %4.1f %% | reads + MFMA issue %4.1f %% | split + commit %4.1f %% | barrier %4.1f %%'
              % (name, half, a.elapsed_time(b) * 1e3, p.shape[0], tot / max(nsteps, 1), 100 * m[0] / tot, 100 * m[1] / tot, 100 * m[2] / tot, 100 * m[3] / tot))

case('ffn dec 808x512x512 x8', 808, 512, 512, 8)
case('ffn enc 2000x512x512 x8', 2000, 512, 512, 8)
case('808x512x4096 x8', 808, 512, 4096, 8)
case('dX 808x512x512 x8 (NN)', 808, 512, 512, 8, tb=0)
