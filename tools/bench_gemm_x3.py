"""GPU: the bf16-split engine (csrc/mtl_gemm_x3.hip) against the exact-fp32 engines on the products of a task-batched pass
(HIP events, isolated launches) with both results measured against fp64.
usage: python tools/bench_gemm_x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mtl_amd
from mtl_amd import _lib
L = _lib.lib()
dev = 'cuda'
st = lambda: torch.cuda.current_stream().cuda_stream
ws = torch.empty(8 << 20, device=dev)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def case(name, ta, tb, M, N, K, nt, shared_b=False, relu=False, bias=False):
    """nt tasks; op(A) (M x K) per task; op(B) per task or shared; C per task"""
    A = torch.randn(nt, (K if ta else M), (M if ta else K), device=dev)
    Bm = torch.randn(1 if shared_b else nt, (N if tb else K), (K if tb else N), device=dev) * 0.05
    C = torch.zeros(nt, M, N, device=dev)
    bv = torch.randn(N, device=dev) if bias else None
    lda, ldb = A.shape[2], Bm.shape[2]
    def run():
        return L.mtl_gemm_f32_tb(st(), ta, tb, M, N, K, 1.0, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N,
                                 bv.data_ptr() if bias else None, None, 0, 1 if relu else 0, nt, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, None, 0,
                                 ws.data_ptr(), ws.numel() * 4, 0, 0, nt, A[0].numel(), 0 if shared_b else Bm[0].numel(), M * N, 0, 0)
    opA = A.double().transpose(1, 2) if ta else A.double()
    opB = Bm.double().transpose(1, 2) if tb else Bm.double()
    ref = opA @ (opB if not shared_b else opB.expand(nt, -1, -1))
    if bias: ref = ref + bv.double()
    if relu: ref = ref.clamp_min(0)
    out = []
    for mt in (0, 1):
        old = L.mtl_gemm_x3_min_tiles(mt)
        assert run() == 0
        torch.cuda.synchronize()
        err = float((C.double() - ref).norm() / ref.norm())
        t = timeit(run)
        out.append((t, err))
        L.mtl_gemm_x3_min_tiles(old)
    fl = 2.0 * M * N * K * nt
    print('%-28s ta%d tb%d M%-5d N%-5d K%-5d x%d | fp32 %7.1f us %6.1f TF err %.1e | x3 %7.1f us %6.1f TF err %.1e | %.2fx' % (
        name, ta, tb, M, N, K, nt, out[0][0], fl / out[0][0] / 1e6, out[0][1], out[1][0], fl / out[1][0] / 1e6, out[1][1], out[0][0] / out[1][0]))


ONLY = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else None
if ONLY:
    _case = case
    def case(name, *a, **k):
        if name == ONLY: _case(name, *a, **k)
for nt in ((8,) if ONLY else (8, 1)):
    case('ffn fwd (enc)', 0, 1, 2000, 512, 512, nt, shared_b=True, relu=True, bias=True)
    case('ffn fwd (dec)', 0, 1, 808, 512, 512, nt, relu=True, bias=True)
    case('ffn dgrad (enc)', 0, 0, 2000, 512, 512, nt)
    case('ffn wgrad (enc)', 1, 0, 512, 512, 2000, nt)
    case('ffn wgrad (dec)', 1, 0, 512, 512, 808, nt)
    case('lowrank a fwd (enc)', 0, 1, 2000, 100, 512, nt)
    case('lowrank b fwd (enc)', 0, 1, 2000, 512, 100, nt, bias=True)
    case('lowrank a dgrad (enc)', 0, 0, 2000, 512, 100, nt)
    case('lowrank b dgrad (enc)', 0, 0, 2000, 100, 512, nt)
    case('lowrank a wgrad (enc)', 1, 0, 100, 512, 2000, nt)
    case('lowrank b wgrad (enc)', 1, 0, 512, 100, 2000, nt)
    case('vocab fwd', 0, 1, 808, 3768, 512, nt)
    case('vocab dgrad', 0, 0, 808, 512, 3768, nt)
    case('vocab wgrad', 1, 0, 3768, 512, 808, nt)
    case('input linear wgrad', 1, 0, 512, 5120, 2000, nt)
