"""Error of the three convolution arithmetics against an fp64 convolution on the same data (forward, no bias effects)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
import mtl_amd
L = mtl_amd._lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
nhwc = lambda t: t.permute(0, 3, 2, 1).contiguous()
rel = lambda a, b: float((a - b).norm() / b.norm())
for Cin, Cout, B, T, Fq in [(64, 64, 2, 21, 161), (128, 128, 1, 9, 19)]:
    g = torch.Generator().manual_seed(Cin + Cout + T + 2)
    x = torch.relu(torch.randn(B, Cin, Fq, T, generator=g)); x[0, 0, 0, 0] = 40.0
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin)
    b = torch.zeros(Cout)
    y64 = torch.relu(F.conv2d(x.double(), w.double(), padding=1))
    y32 = torch.relu(F.conv2d(x, w, padding=1))
    y32g = torch.relu(F.conv2d(x.cuda(), w.cuda(), padding=1)).cpu()
    dxn, dw, db = nhwc(x).cuda(), w.cuda(), b.cuda()
    y = torch.empty(B, T, Fq, Cout).cuda()
    out = {'torch cpu fp32': rel(y32.double(), y64), 'torch gpu fp32': rel(y32g.double(), y64)}
    wf, wd = torch.empty(9, Cin, Cout).cuda(), torch.empty(9, Cout, Cin).cuda()
    L.mtl_conv3x3_wprep(st(), dw.data_ptr(), wf.data_ptr(), wd.data_ptr(), Cout, Cin)
    L.mtl_conv3x3_relu_fwd(st(), dxn.data_ptr(), wf.data_ptr(), db.data_ptr(), y.data_ptr(), B, T, Fq, Cin, Cout)
    out['fp32 mfma'] = rel(y.permute(0, 3, 2, 1).double().cpu(), y64)
    w3f = torch.empty(3 * 9 * Cin * Cout, dtype=torch.bfloat16).cuda(); w3d = torch.empty_like(w3f)
    L.mtl_conv3x3_wprep_x3(st(), dw.data_ptr(), w3f.data_ptr(), w3d.data_ptr(), Cout, Cin)
    L.mtl_conv3x3_relu_fwd_x3(st(), dxn.data_ptr(), w3f.data_ptr(), db.data_ptr(), y.data_ptr(), B, T, Fq, Cin, Cout)
    out['x3'] = rel(y.permute(0, 3, 2, 1).double().cpu(), y64)
    nb = L.mtl_conv3x3_wprep_h2_bytes(Cout, Cin)
    w2f = torch.empty(nb, dtype=torch.uint8).cuda(); w2d = torch.empty_like(w2f)
    L.mtl_conv3x3_wprep_h2(st(), dw.data_ptr(), w2f.data_ptr(), w2d.data_ptr(), Cout, Cin)
    for name, ax in (('h2', 40.0), ('h2 amax x8', 320.0), ('h2 amax x64', 2560.0)):
        a = torch.tensor([ax] * 2048).cuda()
        L.mtl_conv3x3_relu_fwd_h2(st(), dxn.data_ptr(), a.data_ptr(), w2f.data_ptr(), db.data_ptr(), y.data_ptr(), None, B, T, Fq, Cin, Cout)
        out[name] = rel(y.permute(0, 3, 2, 1).double().cpu(), y64)
    print((Cin, Cout), ' '.join('%s=%.2e' % kv for kv in out.items()))
