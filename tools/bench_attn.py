"""GPU micro-benchmark: the fused attention kernels on the three attention shapes of the north-star pass (HIP events, 50 reps)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mtl_amd
L = mtl_amd._lib.lib()
dev = torch.device('cuda')
BT = int(os.environ.get("MTL_BENCH_B", "8"))
shapes = [('enc self', BT, 8, 250, 250, 0), ('dec self', BT, 8, 101, 101, 1), ('dec cross', BT, 8, 101, 250, 0), ('T=5000 enc', 8, 8, 1250, 1250, 0)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
dk = dv = 64
for name, B, H, Tq, Tk, causal in shapes:
    q = torch.randn(B, Tq, H * dk, device=dev)
    k = torch.randn(B, Tk, H * dk, device=dev)
    v = torch.randn(B, Tk, H * dv, device=dev)
    dO = torch.randn(B, Tq, H * dv, device=dev)
    O = torch.empty_like(dO)
    lse = torch.empty(B, H, Tq, device=dev)
    delta = torch.empty(B * H * Tq, device=dev)
    gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    klen = torch.full((B,), Tk, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def fwd():
        assert L.mtl_attn_fwd(st, q.data_ptr(), k.data_ptr(), v.data_ptr(), H * dk, H * dk, H * dv, klen.data_ptr(), causal, 0.125, B, H, Tq,
                              Tk, dk, dv, None, 0, 1.0, O.data_ptr(), H * dv, lse.data_ptr()) == 0
    def bwd():
        assert L.mtl_attn_bwd(st, q.data_ptr(), k.data_ptr(), v.data_ptr(), H * dk, H * dk, H * dv, klen.data_ptr(), causal, 0.125, B, H, Tq,
                              Tk, dk, dv, None, 0, 1.0, O.data_ptr(), dO.data_ptr(), H * dv, lse.data_ptr(), delta.data_ptr(), gq.data_ptr(),
                              gk.data_ptr(), gv.data_ptr(), H * dk, H * dk, H * dv) == 0
    for fn, nm, units in ((fwd, 'fwd', 2), (bwd, 'bwd', 7)):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 50
        fl = units * 2.0 * B * H * Tq * Tk * dk * (0.5 if causal else 1.0)
        print('%-12s %s: %7.1f us  %6.1f TF (algorithmic, %d products)' % (name, nm, us, fl / us / 1e6, units))
