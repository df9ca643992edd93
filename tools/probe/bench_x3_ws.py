"""(Needs tools/probe/gemm_x3_wave_specialised.patch applied to csrc/mtl_gemm_x3.hip: the production library has no MTL_GEMM_X3_WS switch.)
x3 engine, wave-specialised form: K-step anatomy by runtime ablation (MTL_GEMM_X3_WSDBG bits: 2 no fetch after the prologue, 4 no
split / commit, 8 no fragment reads / MFMAs) on a few shapes of the batched pass.  usage: python tools/probe/bench_x3_ws.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
import mtl_amd
from mtl_amd import _lib
L = _lib.lib(); dev = 'cuda'
st = lambda: torch.cuda.current_stream().cuda_stream
ws = torch.empty(8 << 20, device=dev)
def t(ta, tb, M, N, K, nt):
    A = torch.randn(nt, (K if ta else M), (M if ta else K), device=dev); B = torch.randn(nt, (N if tb else K), (K if tb else N), device=dev); C = torch.zeros(nt, M, N, device=dev)
    run = lambda: L.mtl_gemm_f32_tb(st(), ta, tb, M, N, K, 1.0, A.data_ptr(), A.shape[2], B.data_ptr(), B.shape[2], C.data_ptr(), N, None, None, 0, 0, nt, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0, nt, A[0].numel(), B[0].numel(), M * N, 0, 0)
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3
print(' '.join('%%7.1f' %% t(*c) for c in [(0,1,808,512,512,8),(0,1,808,512,2048,8),(0,1,2000,512,512,8),(0,1,2000,512,2048,8),(1,0,512,512,2000,16)]))
''' % ROOT
print('%-34s %s' % ('us per launch', '808x512x512x8 808x512x2048x8 2000x512x512x8 2000x512x2048x8 wgrad512x512x2000x16'))
for name, env in (('lock-step kernel (WS=0)', dict(MTL_GEMM_X3_WS='0')), ('wave-specialised', dict(MTL_GEMM_X3_WS='1')),
                  ('  no fetch after prologue', dict(MTL_GEMM_X3_WSDBG='2')), ('  no fetch, no split/commit', dict(MTL_GEMM_X3_WSDBG='6')),
                  ('  no MFMA / fragment reads', dict(MTL_GEMM_X3_WSDBG='8')), ('  bare loop + barrier', dict(MTL_GEMM_X3_WSDBG='14')),
                  ('  fragment reads, no MFMA', dict(MTL_GEMM_X3_WSDBG='16')), ('  MFMA, no fragment reads', dict(MTL_GEMM_X3_WSDBG='32')),
                  ('  reads only, no producer work', dict(MTL_GEMM_X3_WSDBG='22')), ('  MFMA only, no producer work', dict(MTL_GEMM_X3_WSDBG='38')),
                  ('  commit only (no fetch, no consumer)', dict(MTL_GEMM_X3_WSDBG='10'))):
    out = subprocess.run([sys.executable, '-c', CODE], env=dict(os.environ, **env), capture_output=True, text=True)
    print('%-34s %s' % (name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]))
