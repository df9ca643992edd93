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
# the products of one task-batched pass at the north-star shapes (8 tasks; nz = batch items of the launch)
case('ffn fwd (enc)', 0, 1, 2000, 512, 512, 8, shared_b=True, relu=True, bias=True)
case('ffn fwd (dec)', 0, 1, 808, 512, 512, 8, relu=True, bias=True)
case('ffn dgrad (enc)', 0, 0, 2000, 512, 512, 8)
case('ffn dgrad (dec)', 0, 0, 808, 512, 512, 8)
case('ffn wgrad (enc, 2 layers)', 1, 0, 512, 512, 2000, 16)
case('ffn wgrad (dec, 4 layers)', 1, 0, 512, 512, 808, 32)
case('lowrank a fwd (enc qkv)', 0, 1, 2000, 100, 512, 24)
case('lowrank b fwd (enc qkv)', 0, 1, 2000, 512, 100, 24, bias=True)
case('lowrank a fwd (dec)', 0, 1, 808, 100, 512, 8)
case('lowrank b fwd (dec)', 0, 1, 808, 512, 100, 8, bias=True)
case('lowrank b fwd (cross kv)', 0, 1, 2000, 512, 100, 64, bias=True)
case('lowrank a dgrad (dec)', 0, 0, 808, 512, 100, 8)
case('lowrank b dgrad (dec)', 0, 0, 808, 100, 512, 8)
case('lowrank a wgrad (enc)', 1, 0, 100, 512, 2000, 48)
case('lowrank b wgrad (enc)', 1, 0, 512, 100, 2000, 48)
case('lowrank a wgrad (dec)', 1, 0, 100, 512, 808, 96)
case('vocab fwd', 0, 1, 808, 3768, 512, 8)
case('vocab dgrad', 0, 0, 808, 512, 3768, 8)
case('vocab wgrad', 1, 0, 3768, 512, 808, 8)
case('input linear wgrad', 1, 0, 512, 5120, 2000, 8)
if not ONLY:      # one task per GPU (BASELINE.json configs[2]: the rank's own pass)
    case('ffn fwd (enc), 1 task', 0, 1, 2000, 512, 512, 1, relu=True, bias=True)
    case('ffn wgrad (enc), 1 task', 1, 0, 512, 512, 2000, 2)
    case('vocab fwd, 1 task', 0, 1, 808, 3768, 512, 1)
    case('input linear wgrad, 1 task', 1, 0, 512, 5120, 2000, 1)
