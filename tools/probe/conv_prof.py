"""Stall breakdown of the halo-tiled convolution kernels from s_memtime stamps inside the kernel (probe build only:
hipcc -DMTL_X3_PROF mtl_mfma.hip, linked as tools/probe/libmtl_prof.so).  Unlike data-removing ablations this runs the real
instruction stream on real operands, so the clock / power state is the production one.
usage: MTL_LIB=tools/probe/libmtl_prof.so python tools/probe/conv_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mtl_amd
from mtl_amd import _lib
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.mtl_x3_prof_set.argtypes = [ctypes.c_void_p]
st = lambda: torch.cuda.current_stream().cuda_stream
B, T, F = 8, 1000, 161
dev = 'cuda'
NWG, NW = 512, 16
prof = torch.zeros(NWG * NW * 8, dtype=torch.int64, device=dev)

def report(name, ncons, launch):
    for _ in range(3): launch()
    torch.cuda.synchronize()
    prof.zero_()
    raw.mtl_x3_prof_set(prof.data_ptr())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); launch(); b.record()
    torch.cuda.synchronize()
    raw.mtl_x3_prof_set(None)
    ms = a.elapsed_time(b)
    p = prof.view(NWG, NW, 8).cpu().double()
    live = p[:, :, :].sum(dim=(1, 2)) > 0
    p = p[live]
    nw = int((p.sum(dim=(0, 2)) > 0).sum())
    cons, wgt, halo = p[:, :ncons], p[:, ncons:ncons + 1], p[:, ncons + 1:ncons + 4]
    tot = cons[:, :, 4].mean()
    f = lambda t: '%5.1f %%' % (100 * float(t) / float(tot))
    print('%s: %.3f ms, %d workgroups, %d waves; consumer main loop = %.0f shader-clock ticks' % (name, ms, p.shape[0], nw, tot))
    c = cons.mean(dim=(0, 1))
    print('   consumers: issue(reads+mfma) %s | step barrier %s | epilogue %s | swap barrier %s' % (f(c[0]), f(c[1]), f(c[2]), f(c[3])))
    w = wgt.mean(dim=(0, 1))
    print('   weight wave: dma issue %s | landing wait %s | step barrier %s | swap barrier %s' % (f(w[0]), f(w[1]), f(w[2]), f(w[3])))
    h = halo.mean(dim=(0, 1))
    print('   halo waves: prologue %s | tap barriers %s | commit %s | fetch issue %s | swap close %s' % (f(h[0]), f(h[1]), f(h[2]), f(h[3]), f(h[4])))

def case(name, T_, F_, cin, cout, pooled, ncons_f, ncons_d):
    x = torch.relu(torch.randn(B, T_, F_, cin, device=dev)); w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev) * 0.1
    Tp, Fp = T_ // 2, F_ // 2
    nb = L.mtl_conv3x3_wprep_h2_bytes(cout, cin)
    w2f = torch.empty(nb, dtype=torch.uint8, device=dev); w2d = torch.empty_like(w2f)
    L.mtl_conv3x3_wprep_h2(st(), w.data_ptr(), w2f.data_ptr(), w2d.data_ptr(), cout, cin)
    slot = torch.zeros(2048, device=dev)
    if pooled:
        y = torch.empty(B, Tp, Fp, cout, device=dev); am = torch.empty(B, Tp, Fp, cout, dtype=torch.uint8, device=dev)
        ax = x.abs().max().reshape(1).repeat(2048)
        report(name + ' fwd+pool', ncons_f, lambda: L.mtl_conv3x3_relu_pool_fwd_h2(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), am.data_ptr(), slot.data_ptr(), B, T_, F_, cin, cout))
        amp = am.data_ptr()
    else:
        y = torch.empty(B, T_, F_, cout, device=dev)
        ax = x.abs().max().reshape(1).repeat(2048)
        report(name + ' fwd', ncons_f, lambda: L.mtl_conv3x3_relu_fwd_h2(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), slot.data_ptr(), B, T_, F_, cin, cout))
        amp = None
    dy = torch.randn_like(y); ady = dy.abs().max().reshape(1).repeat(2048); dx = torch.empty_like(x)
    report(name + ' dgrad', ncons_d, lambda: L.mtl_conv3x3_dgrad_h2(st(), dy.data_ptr(), ady.data_ptr(), amp, w2d.data_ptr(), x.data_ptr(), dx.data_ptr(), None, B, T_, F_, cin, cout))

case('conv7', T // 2, F // 2, 128, 128, True, 8, 8)
case('conv5', T // 2, F // 2, 64, 128, False, 8, 8)
case('conv2', T, F, 64, 64, True, 4, 4)
