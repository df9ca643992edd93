import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, mtl_amd
L = mtl_amd._lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
for (B, T, F, Cin, Cout) in [(2, 21, 161, 64, 64), (8, 500, 80, 64, 128), (1, 9, 19, 128, 128)]:
    x = torch.relu(torch.randn(B, T, F, Cin, device='cuda')); dy = torch.randn(B, T, F, Cout, device='cuda') * 1e-3
    ax = x.abs().max().reshape(1).repeat(2048); ady = dy.abs().max().reshape(1).repeat(2048)
    need = L.mtl_conv3x3_wgrad_x3_workspace(B, T, F, Cin, Cout, 0)
    ws = torch.empty(need // 4 + 16, device='cuda'); dw = torch.zeros(Cout, Cin, 3, 3, device='cuda'); db = torch.zeros(Cout, device='cuda')
    rc = L.mtl_conv3x3_wgrad_h2(st(), x.data_ptr(), ax.data_ptr(), dy.data_ptr(), ady.data_ptr(), None, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), need, B, T, F, Cin, Cout)
    want = dy.double().sum((0, 1, 2))
    print((B, T, F, Cin, Cout), rc, 'rel', float((db.double() - want).norm() / want.norm()), db[:4].tolist(), want[:4].tolist(), 'ratio', float(db.double().norm() / want.norm()))
for (B, T, F, Cin, Cout) in [(2, 21, 161, 64, 64), (8, 500, 80, 128, 128)]:
    Tp, Fp = T // 2, F // 2
    x = torch.relu(torch.randn(B, T, F, Cin, device='cuda')); dp = torch.randn(B, Tp, Fp, Cout, device='cuda') * 1e-3
    am = torch.randint(0, 4, (B, Tp, Fp, Cout), device='cuda', dtype=torch.uint8)
    ax = x.abs().max().reshape(1).repeat(2048); ady = dp.abs().max().reshape(1).repeat(2048)
    need = L.mtl_conv3x3_wgrad_x3_workspace(B, T, F, Cin, Cout, 1)
    ws = torch.empty(need // 4 + 16, device='cuda'); dw = torch.zeros(Cout, Cin, 3, 3, device='cuda'); db = torch.zeros(Cout, device='cuda')
    rc = L.mtl_conv3x3_wgrad_h2(st(), x.data_ptr(), ax.data_ptr(), dp.data_ptr(), ady.data_ptr(), am.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), need, B, T, F, Cin, Cout)
    want = dp.double().sum((0, 1, 2))
    print('pooled', (B, T, F, Cin, Cout), rc, 'rel', float((db.double() - want).norm() / want.norm()), 'ratio', float(db.double().norm() / want.norm()))
