import sys, os
sys.path.insert(0, os.getcwd())
import torch, mtl_amd
from mtl_amd import _lib
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
dev = 'cuda'
torch.manual_seed(0)
def fwd(x, w2f, bias, cin, cout, pooled, ax):
    B, T_, F_, _ = x.shape
    slot = torch.zeros(2048, device=dev)
    if pooled:
        y = torch.empty(B, T_ // 2, F_ // 2, cout, device=dev); am = torch.empty(B, T_ // 2, F_ // 2, cout, dtype=torch.uint8, device=dev)
        assert L.mtl_conv3x3_relu_pool_fwd_h2(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), am.data_ptr(), slot.data_ptr(), B, T_, F_, cin, cout) == 0
        return y, am, slot
    y = torch.empty(B, T_, F_, cout, device=dev)
    assert L.mtl_conv3x3_relu_fwd_h2(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(), slot.data_ptr(), B, T_, F_, cin, cout) == 0
    return y, None, slot
for (T_, F_, cin, cout, pooled) in ((64, 161, 64, 64, True), (32, 80, 64, 128, False), (32, 80, 128, 128, True), (1000, 161, 64, 64, True)):
    B = 2
    x = torch.relu(torch.randn(B, T_, F_, cin, device=dev)); w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05; bias = torch.randn(cout, device=dev) * 0.1
    nb = L.mtl_conv3x3_wprep_h2_bytes(cout, cin)
    w2f = torch.empty(nb, dtype=torch.uint8, device=dev); w2d = torch.empty_like(w2f)
    L.mtl_conv3x3_wprep_h2(st(), w.data_ptr(), w2f.data_ptr(), w2d.data_ptr(), cout, cin)
    ax = x.abs().max().reshape(1).repeat(2048)
    y1, a1, s1 = fwd(x, w2f, bias, cin, cout, pooled, ax)
    y2, a2, s2 = fwd(x, w2f, bias, cin, cout, pooled, ax)
    ys = [fwd(x[i:i + 1].contiguous(), w2f, bias, cin, cout, pooled, ax) for i in range(B)]
    ysplit = torch.cat([t[0] for t in ys])
    print((T_, F_, cin, cout, pooled), 'repeat equal', torch.equal(y1, y2), 'split equal', torch.equal(y1, ysplit), 'amax equal', float(s1.max()), float(s2.max()), [float(t[2].max()) for t in ys],
          'argmax equal', (a1 is None) or torch.equal(a1, torch.cat([t[1] for t in ys])))
    # data gradient
    dy = torch.randn_like(y1); ady = dy.abs().max().reshape(1).repeat(2048)
    def dg(dy_, x_, am_):
        dx = torch.empty_like(x_)
        assert L.mtl_conv3x3_dgrad_h2(st(), dy_.data_ptr(), ady.data_ptr(), am_.data_ptr() if am_ is not None else None, w2d.data_ptr(), x_.data_ptr(), dx.data_ptr(), None, x_.shape[0], T_, F_, cin, cout) == 0
        return dx
    d1 = dg(dy, x, a1); d2 = dg(dy, x, a1)
    dsp = torch.cat([dg(dy[i:i + 1].contiguous(), x[i:i + 1].contiguous(), None if a1 is None else a1[i:i + 1].contiguous()) for i in range(B)])
    print('    dgrad repeat equal', torch.equal(d1, d2), 'split equal', torch.equal(d1, dsp))
