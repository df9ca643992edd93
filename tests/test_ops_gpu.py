"""Per-kernel parity: every C-ABI entry point of libmtl_hip.so against the same op in plain PyTorch fp32 on the CPU.
Floating-point kernels -> tolerance stated per test (relative L2); integer outputs (arg-max, pool indices) bit-exact."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    import mtl_amd
    assert torch.cuda.is_available()
    return mtl_amd._lib.lib()


def st():
    return torch.cuda.current_stream().cuda_stream


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def dev(t):
    return t.cuda().contiguous()


@pytest.mark.parametrize('ta,tb', [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(101, 250, 64), (808, 3765, 512), (2000, 100, 512), (512, 5120, 300), (64, 64, 32)])
def test_gemm_transposes(L, ta, tb, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    ref = (A.t() if ta else A) @ (B.t() if tb else B)
    dA, dB, dbias = dev(A), dev(B), dev(bias)
    C = dev(C0.clone())
    assert L.mtl_gemm_f32(st(), ta, tb, M, N, K, 1.0, dA.data_ptr(), A.shape[1], dB.data_ptr(), B.shape[1], C.data_ptr(), N,
                          None, None, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, None, 0) == 0
    assert rel(C, ref) < 2e-6
    C = dev(C0.clone())           # bias + relu + accumulate
    assert L.mtl_gemm_f32(st(), ta, tb, M, N, K, 0.5, dA.data_ptr(), A.shape[1], dB.data_ptr(), B.shape[1], C.data_ptr(), N,
                          dbias.data_ptr(), None, 0, 3, 1, 1, 0, 0, 0, 0, 0, 0, 0, None, 0) == 0
    assert rel(C, torch.relu(0.5 * ref + bias) + C0) < 2e-6


@pytest.mark.parametrize('ta,tb', [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_gemm_split_k_deterministic(L, ta, tb):
    """few output tiles + long K -> split-K through the workspace; same epilogue semantics, bitwise repeatable"""
    g = torch.Generator().manual_seed(11 + ta * 2 + tb)
    M, N, K = 100, 512, 2000
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias, C0, gate = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    ref = (torch.relu((A.t() if ta else A) @ (B.t() if tb else B) + bias)) * (gate > 0) + C0
    dA, dB, dbias, dgate = dev(A), dev(B), dev(bias), dev(gate)
    ws = torch.empty(4 << 20).cuda()
    outs = []
    for _ in range(2):
        C = dev(C0.clone())
        assert L.mtl_gemm_f32(st(), ta, tb, M, N, K, 1.0, dA.data_ptr(), A.shape[1], dB.data_ptr(), B.shape[1], C.data_ptr(), N,
                              dbias.data_ptr(), dgate.data_ptr(), N, 3, 1, 1, 0, 0, 0, 0, 0, 0, 0, ws.data_ptr(), ws.numel() * 4) == 0
        outs.append(C.cpu())
    assert rel(outs[0], ref) < 3e-6
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('ta,tb', [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_gemm_long_k_big_tile_split_k(L, ta, tb):
    """the 128x128-tile split-K branch of the dispatch (48..96 big tiles, K >= 2048: the 5120-deep input projection and the
    shapes it also captures): all transposes, bias + ReLU + gate + accumulate epilogue, a strided batch with H = 1, bitwise
    repeatability, and a workspace too small for the split (falls back to the plain path, same result)"""
    g = torch.Generator().manual_seed(41 + ta * 2 + tb)
    M, N, K, nb = 2048, 512, 5120, 3
    A = torch.randn((nb, K, M) if ta else (nb, M, K), generator=g)
    B = torch.randn((nb, N, K) if tb else (nb, K, N), generator=g)
    bias, C0, gate = torch.randn(nb, N, generator=g), torch.randn(nb, M, N, generator=g), torch.randn(nb, M, N, generator=g)
    prod = torch.stack([(A[i].t() if ta else A[i]).double() @ (B[i].t() if tb else B[i]).double() for i in range(nb)])
    ref = torch.relu(prod + bias.double().unsqueeze(1)) * (gate > 0) + C0.double()
    dA, dB, dbias, dgate = dev(A), dev(B), dev(bias), dev(gate)
    lda, ldb = A.shape[2], B.shape[2]
    big = torch.empty(48 << 20).cuda()                         # 192 MiB: room for the slabs of all three items
    small = torch.empty(1 << 10).cuda()

    def run(ws, batch):
        C = dev(C0.clone())
        assert L.mtl_gemm_f32(st(), ta, tb, M, N, K, 1.0, dA.data_ptr(), lda, dB.data_ptr(), ldb, C.data_ptr(), N,
                              dbias.data_ptr(), dgate.data_ptr(), N, 3, batch, 1, A[0].numel(), 0, B[0].numel(), 0, M * N, 0, N,
                              ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0) == 0
        return C.cpu()
    one = run(big, 1)
    assert rel(one[0], ref[0]) < 3e-6 and torch.equal(one[1], C0[1])
    assert torch.equal(run(big, 1), one)
    assert rel(run(small, 1)[0], ref[0]) < 3e-6 and rel(run(None, 1)[0], ref[0]) < 3e-6
    allb = run(big, nb)
    assert rel(allb, ref) < 3e-6 and torch.equal(run(big, nb), allb)


def _gemm_ex(L, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, gate=None, ldg=0, flags=0, alpha=1.0, batch=1, H=1,
             sA=(0, 0), sB=(0, 0), sC=(0, 0), sbias=0, kbatch=1, sAk=0, sBk=0, rowsum=None, srow=0, sbias_h=0, srow_h=0):
    return L.mtl_gemm_f32_ex(st(), ta, tb, M, N, K, alpha, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc,
                             bias.data_ptr() if bias is not None else None, gate.data_ptr() if gate is not None else None, ldg, flags,
                             batch, H, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], sbias, kbatch, sAk, sBk,
                             rowsum.data_ptr() if rowsum is not None else None, srow, None, 0, sbias_h, srow_h)


@pytest.mark.parametrize('ta,tb', [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(101, 250, 64), (808, 100, 512), (100, 512, 2000), (33, 31, 7), (512, 100, 808), (5, 3765, 301)])
def test_small_tile_gemm(L, ta, tb, M, N, K):
    """mtl_gemm_f32_ex on the small-tile engine (every shape here has < 192 tiles of 64 x 64): all transposes, tile tails in
    M / N / K, unaligned leading dimensions (dword-load instantiation), bias + ReLU + gate + accumulate, bitwise repeatable."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias, C0, gate = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    prod = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    dA, dB, dbias, dgate = dev(A), dev(B), dev(bias), dev(gate)
    C = dev(C0.clone())
    assert _gemm_ex(L, ta, tb, M, N, K, dA, A.shape[1], dB, B.shape[1], C, N) == 0
    assert rel(C, prod) < 2e-6
    outs = []
    for _ in range(2):
        C = dev(C0.clone())
        assert _gemm_ex(L, ta, tb, M, N, K, dA, A.shape[1], dB, B.shape[1], C, N, bias=dbias, gate=dgate, ldg=N, flags=3, alpha=0.5) == 0
        outs.append(C.cpu())
    assert rel(outs[0], torch.relu(0.5 * prod + bias.double()) * (gate > 0) + C0.double()) < 2e-6
    assert torch.equal(outs[0], outs[1])


def test_gemm_k_batching_and_row_sums(L):
    """the two extensions of mtl_gemm_f32_ex: dx += sum_z da[z] . W_a[z] in ONE launch (parameters at a constant stride, like the
    Q/K/V a-stages in the flat buffer), and the bias gradient colsum(dy) produced by the weight-gradient product dW = dy^T . x,
    also strided-batch (three b-stage gradients + three bias gradients in one call)."""
    g = torch.Generator().manual_seed(77)
    rows, r, d, n = 808, 100, 512, 3
    da = torch.randn(n, rows, r, generator=g)
    theta = torch.randn(n * (r * d + 1000), generator=g)                    # a "flat parameter buffer": W_a[z] every r*d+1000 floats
    sa = r * d + 1000
    Wa = [theta[z * sa: z * sa + r * d].view(r, d) for z in range(n)]
    dx0 = torch.randn(rows, d, generator=g)
    ref = dx0.double() + sum(da[z].double() @ Wa[z].double() for z in range(n))
    dda, dth = dev(da), dev(theta)
    outs = []
    for _ in range(2):
        dx = dev(dx0.clone())
        assert _gemm_ex(L, 0, 0, rows, d, r, dda, r, dth, d, dx, d, flags=2, kbatch=n, sAk=rows * r, sBk=sa) == 0
        outs.append(dx.cpu())
    assert rel(outs[0], ref) < 2e-6 and torch.equal(outs[0], outs[1])
    # dW_b[z] += dy[z]^T a[z]  and  db[z] += colsum(dy[z]),  z = 0..2, outputs strided into a flat gradient buffer
    wd = 512
    dy = torch.randn(n, rows, wd, generator=g)
    a = torch.randn(n, rows, r, generator=g)
    sb = wd * r + wd + 24                                                    # weight, then its bias, then something else
    G0 = torch.randn(n * sb, generator=g)
    ddy, da_, G = dev(dy), dev(a), dev(G0.clone())
    Wview = G[:].data_ptr()
    assert L.mtl_gemm_f32_ex(st(), 1, 0, wd, r, rows, 1.0, ddy.data_ptr(), wd, da_.data_ptr(), r, Wview, r, None, None, 0, 2,
                             n, 1, rows * wd, 0, rows * r, 0, sb, 0, 0, 1, 0, 0, Wview + 4 * wd * r, sb, None, 0, 0, 0) == 0
    Gc = G.cpu()
    for z in range(n):
        w_ref = G0[z * sb: z * sb + wd * r].view(wd, r).double() + dy[z].double().t() @ a[z].double()
        b_ref = G0[z * sb + wd * r: z * sb + wd * r + wd].double() + dy[z].double().sum(0)
        assert rel(Gc[z * sb: z * sb + wd * r].view(wd, r), w_ref) < 2e-6
        assert rel(Gc[z * sb + wd * r: z * sb + wd * r + wd], b_ref) < 3e-6
        assert torch.equal(Gc[z * sb + wd * r + wd: (z + 1) * sb], G0[z * sb + wd * r + wd: (z + 1) * sb])   # untouched
    assert L.mtl_gemm_f32_ex(st(), 0, 0, 8, 8, 8, 1.0, ddy.data_ptr(), 8, da_.data_ptr(), 8, Wview, 8, None, None, 0, 0,
                             1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, Wview, 0, None, 0, 0, 0) == -22      # row sums need transA


def test_gemm_gate_and_batched_heads(L):
    g = torch.Generator().manual_seed(5)
    Bn, H, Tq, Tk, dk = 3, 8, 101, 250, 16
    q = torch.randn(Bn, Tq, H * dk, generator=g)
    k = torch.randn(Bn, Tk, H * dk, generator=g)
    ld = (Tk + 3) // 4 * 4
    S = torch.full((Bn, H, Tq, ld), float('nan')).cuda()
    dq, dk_ = dev(q), dev(k)
    assert L.mtl_gemm_f32(st(), 0, 1, Tq, Tk, dk, 1.0, dq.data_ptr(), H * dk, dk_.data_ptr(), H * dk, S.data_ptr(), ld, None, None,
                          0, 0, Bn * H, H, Tq * H * dk, dk, Tk * H * dk, dk, H * Tq * ld, Tq * ld, 0, None, 0) == 0
    ref = torch.einsum('bqhd,bkhd->bhqk', q.view(Bn, Tq, H, dk), k.view(Bn, Tk, H, dk))
    assert rel(S[..., :Tk], ref) < 2e-6
    M, N, K = 300, 200, 96
    A, Bm, gate = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g), torch.randn(M, N, generator=g)
    C = torch.empty(M, N).cuda()
    dA, dB, dg = dev(A), dev(Bm), dev(gate)
    assert L.mtl_gemm_f32(st(), 0, 0, M, N, K, 1.0, dA.data_ptr(), K, dB.data_ptr(), N, C.data_ptr(), N, None, dg.data_ptr(), N, 0,
                          1, 1, 0, 0, 0, 0, 0, 0, 0, None, 0) == 0
    assert rel(C, (A @ Bm) * (gate > 0)) < 2e-6


@pytest.mark.parametrize('M,N,K,batch', [(8, 512, 512, 1), (8, 100, 512, 3), (8, 512, 100, 3), (8, 3765, 512, 1), (1, 64, 128, 1), (16, 37, 2052, 2),
                                         (3, 16, 4, 1)])
def test_gemm_few_rows_weight_streaming_kernel(L, M, N, K, batch):
    """mtl_gemm_f32_ex with M <= 16 rows, NT (what a decode step issues against every decoder weight: gemm_rows_kernel): against fp64,
    with bias + ReLU, accumulate, a strided batch with a leading dimension larger than N (rows of a K/V cache), bitwise repeatable; and the
    route report says 3."""
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    A = torch.randn(batch, M, K, generator=g)
    W = torch.randn(batch, N, K, generator=g) / np.sqrt(K)
    bias = torch.randn(batch, N, generator=g)
    ldc = N + 12
    C0 = torch.randn(batch, M, ldc, generator=g)
    dA, dW, db = dev(A), dev(W), dev(bias)
    assert L.mtl_gemm_f32_ex_route(M, N, K, batch, 1, 0) == 3
    for flags in (0, 1, 2, 3):                            # RELU = 1, ACCUM = 2
        outs = []
        for _ in range(2):
            C = dev(C0.clone())
            assert L.mtl_gemm_f32_ex(st(), 0, 1, M, N, K, 0.5, dA.data_ptr(), K, dW.data_ptr(), K, C.data_ptr(), ldc, db.data_ptr(), None, 0, flags,
                                     batch, 1, M * K, 0, N * K, 0, M * ldc, 0, N, 1, 0, 0, None, 0, None, 0, 0, 0) == 0
            outs.append(C.cpu())
        assert torch.equal(outs[0], outs[1])
        ref = 0.5 * torch.einsum('zmk,znk->zmn', A.double(), W.double()) + bias.double()[:, None, :]
        if flags & 1:
            ref = ref.clamp_min(0)
        if flags & 2:
            ref = ref + C0[:, :, :N].double()
        assert rel(outs[0][:, :, :N], ref) < 2e-6, flags
        assert torch.equal(outs[0][:, :, N:], C0[:, :, N:])          # the columns beyond N (other rows' cache entries) are untouched


@pytest.mark.parametrize('M,N,K,ta,tb', [(333, 100, 512, 0, 1), (512, 100, 700, 1, 0), (100, 512, 1999, 1, 0), (2000, 512, 100, 0, 1)])
def test_gemm_strided_parameter_batch(L, M, N, K, ta, tb):
    """three products in one call (the Q/K/V projections: operands at a constant stride inside one flat buffer, per-item
    bias, accumulate into strided outputs), with and without split-K; bitwise repeatable and equal to three separate calls"""
    g = torch.Generator().manual_seed(M + N + K)
    pad = 52                                              # other parameters sit between the batched ones
    sa, sb = (K * M if ta else M * K) + pad, (N * K if tb else K * N) + pad
    A, Bm = torch.randn(3 * sa, generator=g), torch.randn(3 * sb, generator=g)
    bias, C0 = torch.randn(3 * (N + 12), generator=g), torch.randn(3 * (M * N + pad), generator=g)
    dA, dB, dbias = dev(A), dev(Bm), dev(bias)
    ws = torch.empty(8 << 20).cuda()
    lda, ldb = (M if ta else K), (K if tb else N)
    outs = []
    for _ in range(2):
        C = dev(C0.clone())
        assert L.mtl_gemm_f32(st(), ta, tb, M, N, K, 1.0, dA.data_ptr(), lda, dB.data_ptr(), ldb, C.data_ptr(), N, dbias.data_ptr(), None,
                              0, 2, 3, 1, sa, 0, sb, 0, M * N + pad, 0, N + 12, ws.data_ptr(), ws.numel() * 4) == 0
        outs.append(C.cpu())
    assert torch.equal(outs[0], outs[1])
    ref, single = C0.clone(), dev(C0.clone())
    for i in range(3):
        a = A[i * sa:i * sa + M * K].view((K, M) if ta else (M, K))
        b = Bm[i * sb:i * sb + N * K].view((N, K) if tb else (K, N))
        o = i * (M * N + pad)
        ref[o:o + M * N] += ((a.t() if ta else a) @ (b.t() if tb else b) + bias[i * (N + 12):i * (N + 12) + N]).reshape(-1)
        assert L.mtl_gemm_f32(st(), ta, tb, M, N, K, 1.0, dA.data_ptr() + 4 * i * sa, lda, dB.data_ptr() + 4 * i * sb, ldb,
                              single.data_ptr() + 4 * o, N, dbias.data_ptr() + 4 * i * (N + 12), None, 0, 2, 1, 1, 0, 0, 0, 0, 0, 0, 0,
                              None, 0) == 0
    assert rel(outs[0], ref) < 3e-6
    assert rel(outs[0], single.cpu()) < 2e-6            # (split-K changes the summation order, not the value class)
    assert torch.equal(outs[0][M * N:M * N + pad], C0[M * N:M * N + pad])       # the gaps are untouched


def nhwc(t):   # reference (B,C,F,T) -> ours (B,T,F,C)
    return t.permute(0, 3, 2, 1).contiguous()


def from_nhwc(t):
    return t.permute(0, 3, 2, 1).contiguous()


@pytest.mark.parametrize('B,T,Fq', [(2, 37, 161), (1, 16, 21)])
def test_conv0(L, B, T, Fq):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 1, Fq, T, generator=g)
    w, b = torch.randn(64, 1, 3, 3, generator=g) * 0.3, torch.randn(64, generator=g)
    ref = torch.relu(F.conv2d(x, w, b, padding=1))
    dx, dw, db = dev(x), dev(w), dev(b)
    y = torch.empty(B, T, Fq, 64).cuda()
    amax = torch.zeros(2048).cuda()                                             # MTL_AMAX_FLOATS: 64 slot heads, 32 floats apart
    assert L.mtl_conv0_relu_fwd(st(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), y.data_ptr(), B, T, Fq, amax.data_ptr()) == 0
    assert rel(from_nhwc(y), ref) < 2e-6
    assert float(amax.max()) == float(y.max())                                        # the scalar a following h2 convolution scales by
    dy = torch.randn(B, 64, Fq, T, generator=g)
    wg = torch.zeros(64, 1, 3, 3).cuda()
    bg = torch.zeros(64).cuda()
    ws = torch.empty(L.mtl_conv0_wgrad_workspace() // 4).cuda()
    d_dy = dev(nhwc(dy))
    assert L.mtl_conv0_wgrad(st(), dx.data_ptr(), d_dy.data_ptr(), wg.data_ptr(), bg.data_ptr(), ws.data_ptr(), B, T, Fq) == 0
    wref = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1)
    assert rel(wg, wref) < 1e-5 and rel(bg, dy.sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize('Cin,Cout,B,T,Fq', [(64, 64, 2, 21, 161), (64, 128, 2, 18, 80), (128, 128, 1, 9, 19)])
def test_conv3x3_all(L, Cin, Cout, B, T, Fq):
    g = torch.Generator().manual_seed(Cin + Cout + T)
    x = torch.relu(torch.randn(B, Cin, Fq, T, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / np.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g) * 0.1
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = torch.relu(F.conv2d(xr, wr, b, padding=1))
    pr, idx = F.max_pool2d(yr, 2, stride=2, return_indices=True)
    dxn, dw, db = dev(nhwc(x)), dev(w), dev(b)
    wf, wd = torch.empty(9, Cin, Cout).cuda(), torch.empty(9, Cout, Cin).cuda()
    assert L.mtl_conv3x3_wprep(st(), dw.data_ptr(), wf.data_ptr(), wd.data_ptr(), Cout, Cin) == 0
    # plain conv + relu
    y = torch.empty(B, T, Fq, Cout).cuda()
    assert L.mtl_conv3x3_relu_fwd(st(), dxn.data_ptr(), wf.data_ptr(), db.data_ptr(), y.data_ptr(), B, T, Fq, Cin, Cout) == 0
    assert rel(from_nhwc(y), yr) < 3e-6
    # fused pool
    Tp, Fp = T // 2, Fq // 2
    p = torch.empty(B, Tp, Fp, Cout).cuda()
    am = torch.empty(B, Tp, Fp, Cout, dtype=torch.uint8).cuda()
    assert L.mtl_conv3x3_relu_pool_fwd(st(), dxn.data_ptr(), wf.data_ptr(), db.data_ptr(), p.data_ptr(), am.data_ptr(), B, T, Fq,
                                       Cin, Cout) == 0
    assert rel(from_nhwc(p), pr) < 3e-6
    # arg-max: torch index = f*T + t inside the (F,T) plane
    amc = from_nhwc(am).long().cpu()
    fgrid = torch.arange(Fp).view(1, 1, Fp, 1) * 2 + (amc >> 1)
    tgrid = torch.arange(Tp).view(1, 1, 1, Tp) * 2 + (amc & 1)
    mism = (fgrid * T + tgrid) != idx
    # ties (e.g. all-zero windows after ReLU) must pick torch's first element; values must agree wherever indices differ
    assert int(mism.sum()) == 0
    # backward of the pooled layer: dgrad (fused un-pool + ReLU gate of the INPUT activation) and wgrad
    dp = torch.randn(pr.shape, generator=g)
    pr.backward(dp)
    dpn = dev(nhwc(dp * (pr.detach() > 0)))           # the engine hands over ReLU-gated pooled grads
    dx = torch.empty(B, T, Fq, Cin).cuda()
    assert L.mtl_conv3x3_dgrad(st(), dpn.data_ptr(), am.data_ptr(), wd.data_ptr(), dxn.data_ptr(), dx.data_ptr(), B, T, Fq, Cin,
                               Cout) == 0
    assert rel(from_nhwc(dx), xr.grad * (x > 0)) < 1e-5
    need = L.mtl_conv3x3_wgrad_workspace(B, T, Fq, Cin, Cout, 1)
    ws = torch.empty(need // 4 + 16).cuda()
    wg = torch.zeros(Cout, Cin, 3, 3).cuda()
    assert L.mtl_conv3x3_wgrad(st(), dxn.data_ptr(), dpn.data_ptr(), am.data_ptr(), wg.data_ptr(), ws.data_ptr(), need, B, T, Fq,
                               Cin, Cout) == 0
    assert rel(wg, wr.grad) < 1e-5
    # split-bf16 weight gradient (halo-tiled, transpose reads): same contract, accumulates into dw like the fp32 one
    need3 = L.mtl_conv3x3_wgrad_x3_workspace(B, T, Fq, Cin, Cout, 1)
    ws3 = torch.empty(need3 // 4 + 16).cuda()
    wg3 = torch.ones(Cout, Cin, 3, 3).cuda()
    assert L.mtl_conv3x3_wgrad_x3(st(), dxn.data_ptr(), dpn.data_ptr(), am.data_ptr(), wg3.data_ptr(), ws3.data_ptr(), need3, B, T,
                                  Fq, Cin, Cout) == 0
    assert rel(wg3 - 1.0, wr.grad) < 1e-5
    assert L.mtl_conv3x3_wgrad_x3(st(), dxn.data_ptr(), dpn.data_ptr(), am.data_ptr(), wg3.data_ptr(), ws3.data_ptr(), need3 - 4, B,
                                  T, Fq, Cin, Cout) != 0          # short workspace is refused
    # dense (un-pooled) backward
    xr2 = x.clone().requires_grad_(True)
    wr2 = w.clone().requires_grad_(True)
    y2 = torch.relu(F.conv2d(xr2, wr2, b, padding=1))
    dy = torch.randn(y2.shape, generator=g)
    y2.backward(dy)
    dyn = dev(nhwc(dy * (y2.detach() > 0)))
    assert L.mtl_conv3x3_dgrad(st(), dyn.data_ptr(), None, wd.data_ptr(), dxn.data_ptr(), dx.data_ptr(), B, T, Fq, Cin, Cout) == 0
    assert rel(from_nhwc(dx), xr2.grad * (x > 0)) < 1e-5
    need = L.mtl_conv3x3_wgrad_workspace(B, T, Fq, Cin, Cout, 0)
    ws = torch.empty(need // 4 + 16).cuda()
    wg.zero_()
    assert L.mtl_conv3x3_wgrad(st(), dxn.data_ptr(), dyn.data_ptr(), None, wg.data_ptr(), ws.data_ptr(), need, B, T, Fq, Cin,
                               Cout) == 0
    assert rel(wg, wr2.grad) < 1e-5
    need3 = L.mtl_conv3x3_wgrad_x3_workspace(B, T, Fq, Cin, Cout, 0)
    ws3 = torch.empty(need3 // 4 + 16).cuda()
    wg3 = torch.zeros(Cout, Cin, 3, 3).cuda()
    for _ in range(2):          # bitwise reproducible (fixed slab assignment and reduction order)
        wg3.zero_()
        assert L.mtl_conv3x3_wgrad_x3(st(), dxn.data_ptr(), dyn.data_ptr(), None, wg3.data_ptr(), ws3.data_ptr(), need3, B, T, Fq,
                                      Cin, Cout) == 0
        torch.cuda.synchronize()
        first = wg3.clone() if _ == 0 else first
    assert torch.equal(first, wg3)
    assert rel(wg3, wr2.grad) < 1e-5


@pytest.mark.parametrize('d', [128, 512])
def test_layernorm(L, d):
    g = torch.Generator().manual_seed(d)
    rows, T = 77, 11
    x, res = torch.randn(rows, d, generator=g), torch.randn(rows, d, generator=g)
    gam, bet, pe = torch.randn(d, generator=g), torch.randn(d, generator=g), torch.randn(T, d, generator=g)
    keep = (torch.rand(rows, generator=g) > 0.3).int()
    xr, rr, gr, br = [t.clone().requires_grad_(True) for t in (x, res, gam, bet)]
    yr = (F.layer_norm(xr + rr, (d,), gr, br, 1e-5) + pe[torch.arange(rows) % T]) * keep.unsqueeze(1)
    dy = torch.randn(rows, d, generator=g)
    yr.backward(dy)
    y, xhat, rstd = torch.empty(rows, d).cuda(), torch.empty(rows, d).cuda(), torch.empty(rows).cuda()
    a = [dev(t) for t in (x, res, gam, bet, pe)]
    kd = keep.cuda()
    assert L.mtl_layernorm_fwd(st(), a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr(),
                               kd.data_ptr(), None, 1.0, y.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), rows, d, T, 1e-5) == 0
    assert rel(y, yr) < 2e-6
    dz, dg, db = torch.empty(rows, d).cuda(), torch.zeros(d).cuda(), torch.zeros(d).cuda()
    ws = torch.empty(L.mtl_layernorm_bwd_workspace(rows, d) // 4).cuda()
    ddy = dev(dy)
    dsum = torch.ones(d).cuda()
    assert L.mtl_layernorm_bwd(st(), ddy.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), a[2].data_ptr(), kd.data_ptr(), None, 1.0,
                               dz.data_ptr(), None, None, dg.data_ptr(), db.data_ptr(), dsum.data_ptr(), ws.data_ptr(), rows, d, 0) == 0
    assert rel(dz, xr.grad) < 1e-5 and rel(dg, gr.grad) < 1e-5 and rel(db, br.grad) < 1e-5
    assert rel(dsum, xr.grad.sum(0) + 1) < 1e-5
    # deferred form: the partials stay in the workspace, one batched launch adds them (two table entries here, one without dsum)
    import ctypes
    from mtl_amd import _lib
    dz_b, dg2, db2, dg3, db3, ds2 = (torch.empty(rows, d).cuda(), torch.zeros(d).cuda(), torch.zeros(d).cuda(), torch.ones(d).cuda(),
                                     torch.zeros(d).cuda(), torch.zeros(d).cuda())
    assert L.mtl_layernorm_bwd(st(), ddy.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), a[2].data_ptr(), kd.data_ptr(), None, 1.0,
                               dz_b.data_ptr(), None, None, dg2.data_ptr(), db2.data_ptr(), ds2.data_ptr(), ws.data_ptr(), rows, d, 1) == 0
    torch.cuda.synchronize()
    assert float(dg2.abs().sum()) == 0.0 and torch.equal(dz_b, dz)
    table = (_lib.LnReduceDesc * 2)()
    nw = L.mtl_layernorm_bwd_workspace(rows, d) // (3 * d * 4)
    table[0].part, table[0].dgamma, table[0].dbeta, table[0].dsum, table[0].nw, table[0].d = ws.data_ptr(), dg2.data_ptr(), db2.data_ptr(), ds2.data_ptr(), nw, d
    table[1].part, table[1].dgamma, table[1].dbeta, table[1].dsum, table[1].nw, table[1].d = ws.data_ptr(), dg3.data_ptr(), db3.data_ptr(), None, nw, d
    tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).cuda()
    assert L.mtl_ln_param_reduce_batch(st(), tdev.data_ptr(), 2, d) == 0
    assert torch.equal(dg2, dg) and torch.equal(db2, db) and torch.equal(ds2 + 1, dsum) and torch.equal(dg3, dg + 1) and torch.equal(db3, db)
    # with dropout on the sub-layer output x (before the residual): forward and both gradient branches
    p = 0.25
    seed = torch.tensor([1234567], dtype=torch.int64).cuda()
    mask = torch.empty(rows, d, dtype=torch.uint8).cuda()
    assert L.mtl_dropout_mask(st(), mask.data_ptr(), rows * d, p, seed.data_ptr(), 7 << 40) == 0
    mf = mask.cpu().float() / (1 - p)
    xr, rr, gr, br = [t.clone().requires_grad_(True) for t in (x, res, gam, bet)]
    yr = (F.layer_norm(xr * mf + rr, (d,), gr, br, 1e-5) + pe[torch.arange(rows) % T]) * keep.unsqueeze(1)
    yr.backward(dy)
    sc = 1.0 / (1 - p)
    assert L.mtl_layernorm_fwd(st(), a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr(),
                               kd.data_ptr(), mask.data_ptr(), sc, y.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), rows, d, T, 1e-5) == 0
    assert rel(y, yr) < 2e-6
    dzm, dz2 = torch.empty(rows, d).cuda(), torch.empty(rows, d).cuda()
    dsum.zero_(); dg.zero_(); db.zero_()
    assert L.mtl_layernorm_bwd(st(), ddy.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), a[2].data_ptr(), kd.data_ptr(), mask.data_ptr(),
                               sc, dz.data_ptr(), dzm.data_ptr(), dz2.data_ptr(), dg.data_ptr(), db.data_ptr(), dsum.data_ptr(), ws.data_ptr(),
                               rows, d, 0) == 0
    assert rel(dz, rr.grad) < 1e-5 and rel(dzm, xr.grad) < 1e-5 and rel(dsum, xr.grad.sum(0)) < 1e-5 and rel(dg, gr.grad) < 1e-5
    assert torch.equal(dz2, dz)                                             # second copy for the residual path


@pytest.mark.parametrize('B,H,Tq,Tk,dk,causal,klens,drop', [
    (3, 8, 101, 250, 64, 0, [250, 31, 1], 0.0),       # encoder-decoder attention: ragged key lengths, one key only
    (2, 8, 101, 101, 64, 1, [101, 40], 0.0),          # decoder self-attention: causal + EOS-keyed lengths
    (2, 8, 250, 250, 64, 0, None, 0.0),               # encoder self-attention, tile tails (250 = 3 x 64 + 58)
    (2, 4, 130, 130, 64, 1, [130, 65], 0.25),         # dropout on the probabilities, causal, three key tiles
    (3, 8, 9, 16, 16, 0, [16, 10, 3], 0.0),           # fixture head size (d_k = 16)
    (2, 8, 70, 70, 16, 1, [70, 9], 0.3),
    (2, 8, 2100, 2100, 64, 1, [2100, 900], 0.0),      # > 1024 workgroups: the backward's two-launch form (query side, then key side)
    (8, 8, 1, 301, 64, 0, [1, 2, 64, 65, 150, 256, 257, 301], 0.0),     # ONE query row per (batch, head): the decode kernel over a K/V cache
    (3, 8, 1, 250, 64, 0, None, 0.0),                 # ... cross-attention of a decoding step: every key
    (2, 8, 1, 40, 16, 0, [40, 7], 0.0),               # ... fixture head size
])
def test_fused_attention_forward_and_backward(L, B, H, Tq, Tk, dk, causal, klens, drop):
    """mtl_attn_fwd / mtl_attn_bwd against ScaledDotProductAttention written out in torch (modules/common_layers.py:317-331:
    bmm, /temperature, masked_fill(-inf), softmax, dropout mask, bmm) and its autograd gradients: output 3e-6, gradients 1e-5,
    bitwise repeatable.  Heads are read by stride from (rows, H*d) matrices, exactly as the engine stores q / k / v."""
    g = torch.Generator().manual_seed(B * 100 + Tq + Tk + dk + causal)
    dv = dk
    q = torch.randn(B, Tq, H * dk, generator=g, requires_grad=True)
    k = torch.randn(B, Tk, H * dk, generator=g, requires_grad=True)
    v = torch.randn(B, Tk, H * dv, generator=g, requires_grad=True)
    dO = torch.randn(B, Tq, H * dv, generator=g)
    temp = float(np.power(dk, 0.5))
    blocked = torch.zeros(B, 1, Tq, Tk, dtype=torch.bool)
    if klens is not None:
        blocked = blocked | (torch.arange(Tk).view(1, 1, 1, Tk) >= torch.tensor(klens).view(B, 1, 1, 1))
    if causal:
        blocked = blocked | torch.triu(torch.ones(Tq, Tk, dtype=torch.bool), diagonal=1).view(1, 1, Tq, Tk)
    ldm = (Tk + 3) // 4 * 4
    keep = (torch.rand(B, H, Tq, ldm, generator=g) >= drop).to(torch.uint8) if drop > 0 else None
    qh = q.view(B, Tq, H, dk).transpose(1, 2)
    kh = k.view(B, Tk, H, dk).transpose(1, 2)
    vh = v.view(B, Tk, H, dv).transpose(1, 2)
    sc = (qh @ kh.transpose(2, 3)) / temp
    pr = torch.softmax(sc.masked_fill(blocked, -np.inf), dim=-1)
    if keep is not None:
        pr = pr * keep[..., :Tk].float() / (1 - drop)
    ref = (pr @ vh).transpose(1, 2).reshape(B, Tq, H * dv)
    ref.backward(dO)
    lse_ref = torch.logsumexp(sc.masked_fill(blocked, -np.inf), dim=-1).detach()

    dq_, dk_, dv_, ddO = dev(q.detach()), dev(k.detach()), dev(v.detach()), dev(dO)
    klen = dev(torch.tensor(klens, dtype=torch.int32)) if klens is not None else None
    dkeep = dev(keep) if keep is not None else None
    outs = []
    for _ in range(2):
        O = torch.full((B, Tq, H * dv), float('nan')).cuda()
        lse = torch.empty(B, H, Tq).cuda()
        assert L.mtl_attn_fwd(st(), dq_.data_ptr(), dk_.data_ptr(), dv_.data_ptr(), H * dk, H * dk, H * dv,
                              klen.data_ptr() if klen is not None else None, causal, 1.0 / temp, B, H, Tq, Tk, dk, dv,
                              dkeep.data_ptr() if dkeep is not None else None, ldm, 1.0 / (1 - drop), O.data_ptr(), H * dv,
                              lse.data_ptr()) == 0
        gq = torch.full((B, Tq, H * dk), float('nan')).cuda()
        gk = torch.full((B, Tk, H * dk), float('nan')).cuda()
        gv = torch.full((B, Tk, H * dv), float('nan')).cuda()
        delta = torch.empty(B * H * Tq).cuda()
        assert L.mtl_attn_bwd(st(), dq_.data_ptr(), dk_.data_ptr(), dv_.data_ptr(), H * dk, H * dk, H * dv,
                              klen.data_ptr() if klen is not None else None, causal, 1.0 / temp, B, H, Tq, Tk, dk, dv,
                              dkeep.data_ptr() if dkeep is not None else None, ldm, 1.0 / (1 - drop), O.data_ptr(), ddO.data_ptr(),
                              H * dv, lse.data_ptr(), delta.data_ptr(), gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), H * dk, H * dk,
                              H * dv) == 0
        outs.append([t.cpu() for t in (O, lse, gq, gk, gv)])
    O, lse, gq, gk, gv = outs[0]
    assert rel(O, ref.detach()) < 3e-6
    assert float((lse - lse_ref).abs().max()) < 1e-5
    assert rel(gq, q.grad) < 1e-5 and rel(gk, k.grad) < 1e-5 and rel(gv, v.grad) < 1e-5
    if klens is not None:      # keys beyond a sample's length never receive gradient: exact zeros, not rounding noise
        for b, n in enumerate(klens):
            assert float(gk[b, n:].abs().max() if n < Tk else 0.0) == 0.0 and float(gv[b, n:].abs().max() if n < Tk else 0.0) == 0.0
    for a, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a, b_)


@pytest.mark.parametrize('causal', [0, 1])
def test_softmax(L, causal):
    g = torch.Generator().manual_seed(causal)
    Bn, H, Tq, Tk = 3, 4, 9, 13 if not causal else 9
    ld = (Tk + 3) // 4 * 4
    s = torch.randn(Bn, H, Tq, Tk, generator=g) * 3
    klen = torch.tensor([Tk, 5, 1], dtype=torch.int32)
    blocked = torch.arange(Tk).view(1, 1, 1, Tk) >= klen.view(Bn, 1, 1, 1)
    if causal:
        blocked = blocked | torch.triu(torch.ones(Tq, Tk, dtype=torch.bool), 1)
    sr = s.clone().requires_grad_(True)
    pr = torch.softmax((sr / 4.0).masked_fill(blocked, -np.inf), -1)
    dp = torch.randn(pr.shape, generator=g)
    pr.backward(dp)
    S = torch.zeros(Bn, H, Tq, ld)
    S[..., :Tk] = s
    S = S.cuda()
    kd = klen.cuda()
    assert L.mtl_softmax_mask_fwd(st(), S.data_ptr(), kd.data_ptr(), causal, 0.25, Bn, H, Tq, Tk, ld, None, 1.0, None) == 0
    assert rel(S[..., :Tk], pr) < 2e-6
    D = torch.zeros(Bn, H, Tq, ld)
    D[..., :Tk] = dp
    D = D.cuda()
    assert L.mtl_softmax_bwd(st(), S.data_ptr(), D.data_ptr(), 0.25, Bn * H * Tq, Tk, ld, None, 1.0) == 0
    assert rel(D[..., :Tk], sr.grad) < 1e-5
    # dropout on the probabilities: dropped copy for P.V, un-dropped P kept for the backward
    p = 0.3
    seed = torch.tensor([99], dtype=torch.int64).cuda()
    mask = torch.empty(Bn, H, Tq, ld, dtype=torch.uint8).cuda()
    assert L.mtl_dropout_mask(st(), mask.data_ptr(), mask.numel(), p, seed.data_ptr(), 3 << 40) == 0
    mf = mask.cpu().float()[..., :Tk] / (1 - p)
    sr = s.clone().requires_grad_(True)
    pd_ref = torch.softmax((sr / 4.0).masked_fill(blocked, -np.inf), -1) * mf
    pd_ref.backward(dp)
    S = torch.zeros(Bn, H, Tq, ld); S[..., :Tk] = s; S = S.cuda()
    Pd = torch.zeros(Bn, H, Tq, ld).cuda()
    assert L.mtl_softmax_mask_fwd(st(), S.data_ptr(), kd.data_ptr(), causal, 0.25, Bn, H, Tq, Tk, ld, mask.data_ptr(), 1 / (1 - p),
                                  Pd.data_ptr()) == 0
    assert rel(Pd[..., :Tk], pd_ref) < 2e-6 and rel(S[..., :Tk], pr) < 2e-6
    D = torch.zeros(Bn, H, Tq, ld); D[..., :Tk] = dp; D = D.cuda()
    assert L.mtl_softmax_bwd(st(), S.data_ptr(), D.data_ptr(), 0.25, Bn * H * Tq, Tk, ld, mask.data_ptr(), 1 / (1 - p)) == 0
    assert rel(D[..., :Tk], sr.grad) < 1e-5


def test_embed_and_ce(L):
    g = torch.Generator().manual_seed(3)
    V, d, B, T = 3765, 128, 3, 7
    table, pe = torch.randn(V, d, generator=g), torch.randn(T, d, generator=g)
    ids = torch.randint(1, V, (B, T), generator=g)
    ids[0, 1] = ids[0, 0]
    out = torch.empty(B * T, d).cuda()
    dt, dpe, dids = dev(table), dev(pe), ids.cuda()
    assert L.mtl_embed_pe_fwd(st(), dids.data_ptr(), dt.data_ptr(), dpe.data_ptr(), out.data_ptr(), B * T, T, d, None, 1.0) == 0
    assert rel(out, (table[ids] + pe.unsqueeze(0)).view(B * T, d)) == 0
    dout = torch.randn(B * T, d, generator=g)
    tg = torch.zeros(V, d).cuda()
    ddo = dev(dout)
    flat = ids.view(-1).tolist()
    last, first, nxt = {}, [], [-1] * len(flat)
    for r, v in enumerate(flat):
        first.append(0 if v in last else 1)
        if v in last:
            nxt[last[v]] = r
        last[v] = r
    dfirst, dnext = torch.tensor(first, dtype=torch.int32).cuda(), torch.tensor(nxt, dtype=torch.int32).cuda()
    assert L.mtl_embed_bwd(st(), dids.data_ptr(), dfirst.data_ptr(), dnext.data_ptr(), ddo.data_ptr(), tg.data_ptr(), B * T, d, 0, None,
                           1.0) == 0
    ref = torch.zeros(V, d).index_add_(0, ids.view(-1), dout)
    assert rel(tg, ref) < 1e-6
    # cross entropy + arg-max (ties -> lowest index; padded rows are all-zero logits)
    rows = 40
    logits = torch.randn(rows, V, generator=g)
    logits[5] = 0
    logits[6, 50] = logits[6, 100] = logits[6, 3000] = 9.0
    gold = torch.randint(4, V, (rows,), generator=g)
    gold[5] = 0
    gold[9] = 0
    lr_ = logits.clone().requires_grad_(True)
    loss_ref = F.cross_entropy(lr_, gold, ignore_index=0, reduction='mean')
    (loss_ref / 3).backward()
    dl, dgold = dev(logits), gold.cuda()
    lse, hyp = torch.empty(rows).cuda(), torch.empty(rows, dtype=torch.int64).cuda()
    rowloss, loss = torch.empty(rows).cuda(), torch.empty(1).cuda()
    nn_ = int((gold != 0).sum())
    assert L.mtl_ce_argmax_fwd(st(), dl.data_ptr(), dgold.data_ptr(), rows, V, V, 0, 0.0, nn_, None, lse.data_ptr(), hyp.data_ptr(),
                               rowloss.data_ptr(), loss.data_ptr()) == 0
    inv = torch.tensor([1.0 / nn_]).cuda()
    loss2 = torch.empty(1).cuda()
    assert L.mtl_ce_argmax_fwd(st(), dl.data_ptr(), dgold.data_ptr(), rows, V, V, 0, 0.0, 0, inv.data_ptr(), lse.data_ptr(),
                               hyp.data_ptr(), rowloss.data_ptr(), loss2.data_ptr()) == 0
    assert abs(float(loss2) - float(loss)) < 1e-6 * float(loss)
    assert abs(float(loss) - float(loss_ref)) < 2e-6 * float(loss_ref)
    assert torch.equal(hyp.cpu(), torch.topk(logits, 1, dim=1)[1].squeeze(1))
    assert int(hyp[5]) == 0 and int(hyp[6]) == 50
    ldd = (V + 3) // 4 * 4
    dlog = torch.empty(rows, ldd).cuda()
    assert L.mtl_ce_bwd(st(), dl.data_ptr(), lse.data_ptr(), dgold.data_ptr(), rows, V, V, 0, 0.0, (1.0 / 3) / nn_, None,
                        dlog.data_ptr(), ldd) == 0
    assert rel(dlog[:, :V], lr_.grad) < 1e-5


@pytest.mark.parametrize('V,ld,rows', [(3765, 3765, 37), (5, 5, 9), (64, 67, 11), (1000, 1000, 13), (4093, 4093, 6), (4094, 4096, 5), (5000, 5003, 7)])
def test_ce_forward_row_widths_and_alignments(L, V, ld, rows):
    """ce_fwd_kernel reads a row once, as 16-byte quads from the aligned address below its first element (rows start at every
    alignment when ld % 4 != 0), up to 4093 logits; longer rows take the two-sweep form.  lse / loss (with label smoothing: the sum
    over the row) against fp64, arg-max bit-exact with lowest-index ties, at every row alignment, with a buffer that ENDS at the
    last logit (nothing beyond it may be read)."""
    g = torch.Generator().manual_seed(V + ld)
    flat = torch.randn(rows * ld - (ld - V), generator=g)                    # the buffer ends with the last valid logit of the last row
    pad = torch.cat([flat, torch.zeros(ld - V)]).view(rows, ld)
    logits = pad[:, :V].clone()
    logits[1] = 0.0                                                          # a zeroed (padded) row: every logit ties
    if V > 4:
        logits[2, V - 1] = logits[2, 3] = logits[2].max() + 1.0              # a tie between the last element and an early one
    flat[:] = torch.cat([logits, torch.zeros(rows, ld - V)], 1).reshape(-1)[:flat.numel()]
    gold = torch.randint(1, V, (rows,), generator=g)
    gold[1] = 0
    dflat, dgold = dev(flat), gold.cuda()
    for smoothing in (0.0, 0.1):
        lse, hyp = torch.empty(rows).cuda(), torch.empty(rows, dtype=torch.int64).cuda()
        rowloss, loss = torch.empty(rows).cuda(), torch.empty(1).cuda()
        nn_ = int((gold != 0).sum())
        assert L.mtl_ce_argmax_fwd(st(), dflat.data_ptr(), dgold.data_ptr(), rows, V, ld, 0, smoothing, nn_, None, lse.data_ptr(),
                                   hyp.data_ptr(), rowloss.data_ptr(), loss.data_ptr()) == 0
        l64 = logits.double()
        lse_ref = torch.logsumexp(l64, 1)
        assert float((lse.cpu().double() - lse_ref).abs().max()) < 2e-6 * float(lse_ref.abs().max())
        hyp_ref = torch.stack([(row == row.max()).nonzero()[0, 0] for row in logits])       # lowest index of the maximum
        assert torch.equal(hyp.cpu(), hyp_ref)
        assert int(hyp[1]) == 0 and (V <= 4 or int(hyp[2]) == 3)
        logp = l64 - lse_ref[:, None]
        nll = -logp.gather(1, gold[:, None]).squeeze(1)
        ref = ((1 - smoothing) * nll + smoothing / V * (-logp.sum(1) - nll)) if smoothing else nll
        ref = torch.where(gold != 0, ref, torch.zeros_like(ref))
        assert float((rowloss.cpu().double() - ref).abs().max()) < 5e-6 * float(ref.abs().max())
        assert abs(float(loss) - float(ref.sum() / nn_)) < 5e-6 * float(ref.sum() / nn_)


def test_flat_updates_colsum_permute(L):
    g = torch.Generator().manual_seed(9)
    n = 100003
    n4 = n // 4 * 4
    th, gr = torch.randn(n4, generator=g), torch.randn(n4, generator=g)
    dth, dgr, t1 = dev(th), dev(gr), torch.empty(n4).cuda()
    assert L.mtl_sgd_theta_prime(st(), dth.data_ptr(), dgr.data_ptr(), 0.01, t1.data_ptr(), n4) == 0
    assert torch.equal(t1.cpu(), th - 0.01 * gr) or rel(t1, th - 0.01 * gr) < 1e-7
    y = dev(th.clone())
    assert L.mtl_axpy(st(), y.data_ptr(), dgr.data_ptr(), 1.0, n4) == 0
    assert torch.equal(y.cpu(), th + gr)
    # Adam, 3 steps, against torch.optim.Adam
    p = th.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-3)
    m, v, w = torch.zeros(n4).cuda(), torch.zeros(n4).cuda(), dev(th.clone())
    for step in range(1, 4):
        gg = torch.randn(n4, generator=g)
        p.grad = gg.clone()
        opt.step()
        dg = dev(gg)
        assert L.mtl_adam_step(st(), w.data_ptr(), dg.data_ptr(), m.data_ptr(), v.data_ptr(), step, 1e-3, 0.9, 0.999, 1e-8, n4) == 0
    assert rel(w, p.detach()) < 1e-6
    # clip coefficient
    ws, coef = torch.empty(2048).cuda(), torch.empty(1).cuda()
    assert L.mtl_sumsq(st(), dgr.data_ptr(), n4, coef.data_ptr(), ws.data_ptr(), 2, 5.0) == 0
    assert abs(float(coef) - min(1.0, 5.0 / (float(gr.norm()) + 1e-6))) < 1e-6
    # column sums
    X = torch.randn(5000, 100, generator=g)
    out = torch.ones(100).cuda()
    ws = torch.empty(L.mtl_colsum_workspace(5000, 100) // 4).cuda()
    dX = dev(X)
    assert L.mtl_colsum_accum(st(), dX.data_ptr(), 5000, 100, 100, out.data_ptr(), ws.data_ptr(), None) == 0
    assert rel(out, X.sum(0) + 1) < 1e-5
    # (c,h) <-> (h,c) permutation of input_linear columns
    rows, C, H = 7, 128, 5
    wt = torch.randn(rows, C * H, generator=g)
    wp = torch.empty(rows, C * H).cuda()
    dwt = dev(wt)
    amx = torch.zeros(2048).cuda()
    assert L.mtl_permute_hc(st(), dwt.data_ptr(), wp.data_ptr(), rows, C, H, 0, amx.data_ptr()) == 0
    assert float(amx.max()) == float(dwt.abs().max())
    assert torch.equal(wp.cpu(), wt.view(rows, C, H).transpose(1, 2).reshape(rows, -1))
    back = torch.zeros(rows, C * H).cuda()
    assert L.mtl_permute_hc(st(), wp.data_ptr(), back.data_ptr(), rows, C, H, 1, None) == 0
    assert torch.equal(back.cpu(), wt)
    # task-batched form (16-byte path: C % 4 == 0 and H % 4 == 0), strided sources / bounds, accumulating inverse
    tasks, rows, C, H = 3, 5, 128, 40
    wt = torch.randn(tasks, rows + 2, C * H, generator=g)              # task stride larger than the weight
    dwt = dev(wt)
    wp = torch.empty(tasks, rows, C * H).cuda()
    amx = torch.zeros(tasks, 4096).cuda()
    assert L.mtl_permute_hc_tb(st(), dwt.data_ptr(), wp.data_ptr(), rows, C, H, 0, amx.data_ptr(), tasks, (rows + 2) * C * H, rows * C * H, 4096) == 0
    for t in range(tasks):
        assert float(amx[t].max()) == float(dwt[t, :rows].abs().max())
        assert torch.equal(wp[t].cpu(), wt[t, :rows].view(rows, C, H).transpose(1, 2).reshape(rows, -1))
    back = torch.ones(tasks, rows + 2, C * H).cuda()
    assert L.mtl_permute_hc_tb(st(), wp.data_ptr(), back.data_ptr(), rows, C, H, 1, None, tasks, rows * C * H, (rows + 2) * C * H, 0) == 0
    assert torch.equal(back[:, :rows].cpu(), wt[:, :rows] + 1) and torch.equal(back[:, rows:].cpu(), torch.ones(tasks, 2, C * H))


def test_dropout_mask_statistics_and_determinism(L):
    n = 1 << 22
    seed = torch.tensor([2 ** 40 + 17], dtype=torch.int64).cuda()
    a, b, c = (torch.empty(n, dtype=torch.uint8).cuda() for _ in range(3))
    for p in (0.1, 0.5):
        assert L.mtl_dropout_mask(st(), a.data_ptr(), n, p, seed.data_ptr(), 1 << 40) == 0
        assert L.mtl_dropout_mask(st(), b.data_ptr(), n, p, seed.data_ptr(), 1 << 40) == 0
        assert L.mtl_dropout_mask(st(), c.data_ptr(), n, p, seed.data_ptr(), 2 << 40) == 0
        keep = float(a.float().mean())
        assert abs(keep - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n) + 1e-4          # 4 sigma
        assert torch.equal(a, b) and not torch.equal(a, c)                        # same (seed, offset) -> same mask
        assert abs(float((a.float() * c.float()).mean()) - (1 - p) ** 2) < 2e-3    # sites are independent
    assert set(a.unique().tolist()) <= {0, 1}


@pytest.mark.parametrize('Cin,Cout,B,T,Fq', [(64, 64, 2, 21, 161), (64, 128, 2, 18, 80), (128, 128, 1, 9, 19)])
def test_conv3x3_split_bf16_matches_fp32_reference(L, Cin, Cout, B, T, Fq):
    """the "x3" kernels (3-way bf16 operand split, six bf16 MFMAs per product) keep fp32-class accuracy: same tolerances as
    the fp32-MFMA kernels, bit-identical pool arg-max decisions on this data"""
    g = torch.Generator().manual_seed(Cin + Cout + T + 1)
    x = torch.relu(torch.randn(B, Cin, Fq, T, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / np.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = torch.relu(F.conv2d(xr, wr, b, padding=1))
    pr, idx = F.max_pool2d(yr, 2, stride=2, return_indices=True)
    dxn, dw, db = dev(nhwc(x)), dev(w), dev(b)
    w3f = torch.empty(3 * 9 * Cin * Cout, dtype=torch.bfloat16).cuda()
    w3d = torch.empty(3 * 9 * Cin * Cout, dtype=torch.bfloat16).cuda()
    assert L.mtl_conv3x3_wprep_x3(st(), dw.data_ptr(), w3f.data_ptr(), w3d.data_ptr(), Cout, Cin) == 0
    # the split is exact; layout [tap][cin/32][cout][cin%32] with the four 8-value chunks of a row stored at chunk ^ ((cout >> 2) & 3)
    pieces = w3f.view(3, 9, Cin // 32, Cout, 4, 8).float().sum(0).cpu()
    unsw = torch.empty_like(pieces)
    for co in range(Cout):
        for c in range(4):
            unsw[:, :, co, c] = pieces[:, :, co, c ^ ((co >> 2) & 3)]
    pieces = unsw.reshape(9, Cin // 32, Cout, 32)
    assert torch.equal(pieces, w.reshape(Cout, Cin // 32, 32, 9).permute(3, 1, 0, 2))
    y = torch.empty(B, T, Fq, Cout).cuda()
    assert L.mtl_conv3x3_relu_fwd_x3(st(), dxn.data_ptr(), w3f.data_ptr(), db.data_ptr(), y.data_ptr(), B, T, Fq, Cin, Cout) == 0
    assert rel(from_nhwc(y), yr) < 3e-6
    Tp, Fp = T // 2, Fq // 2
    p = torch.empty(B, Tp, Fp, Cout).cuda()
    am = torch.empty(B, Tp, Fp, Cout, dtype=torch.uint8).cuda()
    assert L.mtl_conv3x3_relu_pool_fwd_x3(st(), dxn.data_ptr(), w3f.data_ptr(), db.data_ptr(), p.data_ptr(), am.data_ptr(), B, T, Fq,
                                          Cin, Cout) == 0
    assert rel(from_nhwc(p), pr) < 3e-6
    amc = from_nhwc(am).long().cpu()
    fgrid = torch.arange(Fp).view(1, 1, Fp, 1) * 2 + (amc >> 1)
    tgrid = torch.arange(Tp).view(1, 1, 1, Tp) * 2 + (amc & 1)
    assert int(((fgrid * T + tgrid) != idx).sum()) == 0
    dp = torch.randn(pr.shape, generator=g)
    pr.backward(dp)
    dpn = dev(nhwc(dp * (pr.detach() > 0)))
    dx = torch.empty(B, T, Fq, Cin).cuda()
    assert L.mtl_conv3x3_dgrad_x3(st(), dpn.data_ptr(), am.data_ptr(), w3d.data_ptr(), dxn.data_ptr(), dx.data_ptr(), B, T, Fq, Cin,
                                  Cout) == 0
    assert rel(from_nhwc(dx), xr.grad * (x > 0)) < 1e-5
    xr2 = x.clone().requires_grad_(True)
    y2 = torch.relu(F.conv2d(xr2, w, b, padding=1))
    dy = torch.randn(y2.shape, generator=g)
    y2.backward(dy)
    dyn = dev(nhwc(dy * (y2.detach() > 0)))
    assert L.mtl_conv3x3_dgrad_x3(st(), dyn.data_ptr(), None, w3d.data_ptr(), dxn.data_ptr(), dx.data_ptr(), B, T, Fq, Cin, Cout) == 0
    assert rel(from_nhwc(dx), xr2.grad * (x > 0)) < 1e-5


@pytest.mark.parametrize('k', [0, 10, 16, 20, 24, 27])
def test_conv3x3_two_piece_fp16_dynamic_range_inside_one_tensor(L, k):
    """What the ONE power-of-two scale per tensor of the h2 kernels costs a quiet sample that shares a batch with a loud one
    (csrc/mtl_h2.h): sample 1 is sample-0-like data times 2^-k, its outputs are measured against an fp64 convolution.  With the
    tensor's maximum M scaled into [2^14, 2^15) an element x keeps both fp16 pieces' full 22 bits while |x| >= 2^-17.5 M; below
    that the low piece is a subnormal fp16 and the element carries an ABSOLUTE error of at most 2^-39 M -- a relative error of
    2^(k - 38.5) for the quiet sample's elements, which the 576 / 1152-term sums average down.  Asserted: the quiet sample is
    fp32-class (<= 1e-6 normwise) up to a 2^16 spread, <= 2^(k - 37) beyond (measured 4.5e-6 at 2^20, 7.3e-5 at 2^24, 5.8e-4 at 2^27), and the
    loud sample is never disturbed.  The path's
    own tensors stay far inside the first regime: features are normalised per utterance (utils/data_loader.py:84-94; the
    synthetic ones are N(0,1)), and DESIGN.md 4 lists the measured max / rms ratios of every h2 operand of a pass."""
    Cin, Cout, B, T, Fq = 64, 128, 2, 18, 80
    g = torch.Generator().manual_seed(100 + k)
    x = torch.relu(torch.randn(B, Cin, Fq, T, generator=g))
    x[1] *= 2.0 ** -k
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / np.sqrt(9 * Cin))
    b = torch.zeros(Cout)
    y64 = torch.relu(F.conv2d(x.double(), w.double(), None, padding=1))
    dxn, dw, db = dev(nhwc(x)), dev(w), dev(b)
    nb = L.mtl_conv3x3_wprep_h2_bytes(Cout, Cin)
    w2f, w2d = torch.empty(nb, dtype=torch.uint8).cuda(), torch.empty(nb, dtype=torch.uint8).cuda()
    assert L.mtl_conv3x3_wprep_h2(st(), dw.data_ptr(), w2f.data_ptr(), w2d.data_ptr(), Cout, Cin) == 0
    S = 2048
    ax = dxn.abs().max().reshape(1).repeat(S)
    slots = torch.zeros(S).cuda()
    y = torch.empty(B, T, Fq, Cout).cuda()
    assert L.mtl_conv3x3_relu_fwd_h2(st(), dxn.data_ptr(), ax.data_ptr(), w2f.data_ptr(), db.data_ptr(), y.data_ptr(),
                                     slots.data_ptr(), B, T, Fq, Cin, Cout) == 0
    got = from_nhwc(y).double().cpu()
    loud, quiet = rel(got[0], y64[0]), rel(got[1], y64[1])
    bound = max(1e-6, 2.0 ** (k - 37))
    print('spread 2^%d: loud sample %.2e, quiet sample %.2e (bound %.2e)' % (k, loud, quiet, bound))
    assert loud < 1e-6 and quiet < bound, (k, loud, quiet, bound)


def test_weight_preparation_and_bounds_for_several_parameter_sets(L):
    """mtl_conv3x3_wprep_h2_batch_tb (all three layers of the theta' stack in two launches) and mtl_absmax_f32_tb (the bounds of nt
    tensors in one launch) against the per-set calls: bit for bit."""
    g = torch.Generator().manual_seed(77)
    nt, total = 5, 4 * 70001          # (a set must hold its last layer: 110640 + 128 * 128 * 9 = 258096 floats; a shorter one read past the allocation)
    theta = (torch.randn(nt, total, generator=g) * torch.tensor([3.0 ** k for k in range(nt)]).view(-1, 1)).cuda()
    layers = ((64, 64, 0), (128, 64, 36864 + 16), (128, 128, 36864 + 16 + 73728 + 32))        # (Cout, Cin, offset of the weight in theta)
    nbs = [(L.mtl_conv3x3_wprep_h2_bytes(co, ci) + 255) // 256 * 256 for co, ci, _ in layers]
    fn = [torch.zeros(nt, nb, dtype=torch.uint8).cuda() for nb in nbs]; dn = [torch.zeros(nt, nb, dtype=torch.uint8).cuda() for nb in nbs]
    f1 = [torch.zeros(nt, nb, dtype=torch.uint8).cuda() for nb in nbs]; d1 = [torch.zeros(nt, nb, dtype=torch.uint8).cuda() for nb in nbs]
    for t in range(nt):
        spec = []
        for i, (co, ci, off) in enumerate(layers):
            spec += [theta[t, off:].data_ptr(), fn[i][t].data_ptr(), dn[i][t].data_ptr(), co, ci]
        assert L.mtl_conv3x3_wprep_h2_batch(st(), 3, *spec) == 0
    spec = []
    for i, (co, ci, off) in enumerate(layers):
        spec += [theta[0, off:].data_ptr(), f1[i].data_ptr(), d1[i].data_ptr(), co, ci]
    assert L.mtl_conv3x3_wprep_h2_batch_tb(st(), 3, *spec, nt, total, nbs[0], nbs[1], nbs[2]) == 0
    torch.cuda.synchronize()
    for i, (co, ci, _) in enumerate(layers):
        used = L.mtl_conv3x3_wprep_h2_bytes(co, ci) - 12
        assert torch.equal(f1[i][:, :used], fn[i][:, :used]) and torch.equal(d1[i][:, :used], dn[i][:, :used]), i
        assert int(f1[i][:, :used].sum()) != 0
    S = 2048
    an, a1 = torch.zeros(nt, S).cuda(), torch.zeros(nt, S).cuda()
    for t in range(nt):
        assert L.mtl_absmax_f32(st(), theta[t].data_ptr(), total - 1, an[t].data_ptr()) == 0
    assert L.mtl_absmax_f32_tb(st(), theta.data_ptr(), total - 1, a1.data_ptr(), nt, total, S) == 0
    torch.cuda.synchronize()
    assert torch.equal(a1.view(nt, -1, 32)[:, :, 0].max(1)[0], an.view(nt, -1, 32)[:, :, 0].max(1)[0])
    assert torch.equal(a1.view(nt, -1, 32)[:, :, 0].max(1)[0].cpu(), theta[:, :total - 1].abs().max(1)[0].cpu())


@pytest.mark.parametrize('nt,B,T,Fq,shared_x,shared_w', [(3, 2, 37, 161, False, True), (8, 2, 50, 40, True, False), (5, 3, 21, 80, False, False)])
def test_conv0_and_bias_sums_for_several_tasks_in_one_launch(L, nt, B, T, Fq, shared_x, shared_w):
    """mtl_conv0_relu_fwd_tb / mtl_conv0_wgrad_tb / mtl_colsum_accum_tb (task = a grid dimension; the shared validation batch is a zero
    input stride, theta0 a zero weight stride) against the per-task calls: forward, output bounds and column sums bit for bit; the
    weight gradient within fp32 rounding (the tasks share the workspace's partial rows) and deterministic."""
    g = torch.Generator().manual_seed(nt + B + T)
    S = 2048
    nx, nw = (1 if shared_x else nt), (1 if shared_w else nt)
    x = (torch.randn(nx, B, 1, Fq, T, generator=g) * torch.tensor([2.0 ** k for k in range(nx)]).view(-1, 1, 1, 1, 1)).cuda()
    w = (torch.randn(nw, 64, 1, 3, 3, generator=g) * 0.3).cuda()
    b = (torch.randn(nw, 64, generator=g) * 0.1).cuda()
    xi, wi = (lambda t: 0 if shared_x else t), (lambda t: 0 if shared_w else t)
    yn, y1 = torch.zeros(nt * B, T, Fq, 64).cuda(), torch.zeros(nt * B, T, Fq, 64).cuda()
    an, a1 = torch.zeros(nt, S).cuda(), torch.zeros(nt, S).cuda()
    for t in range(nt):
        assert L.mtl_conv0_relu_fwd(st(), x[xi(t)].data_ptr(), w[wi(t)].data_ptr(), b[wi(t)].data_ptr(), yn[t * B:].data_ptr(), B, T, Fq, an[t].data_ptr()) == 0
    assert L.mtl_conv0_relu_fwd_tb(st(), x.data_ptr(), w.data_ptr(), b.data_ptr(), y1.data_ptr(), B, T, Fq, a1.data_ptr(), nt,
                                   0 if shared_x else x[0].numel(), 0 if shared_w else 576, 0 if shared_w else 64, S) == 0
    torch.cuda.synchronize()
    assert torch.equal(y1, yn) and float(yn.abs().max()) > 0
    assert torch.equal(a1.view(nt, -1, 32)[:, :, 0].max(1)[0], an.view(nt, -1, 32)[:, :, 0].max(1)[0])
    dy = torch.randn(nt * B, T, Fq, 64, generator=g).cuda()
    ws = torch.empty(L.mtl_conv0_wgrad_workspace() // 4).cuda()
    dwn, dbn = torch.full((nt, 64, 9), 0.5).cuda(), torch.full((nt, 64), 0.25).cuda()
    for t in range(nt):
        assert L.mtl_conv0_wgrad(st(), x[xi(t)].data_ptr(), dy[t * B:].data_ptr(), dwn[t].data_ptr(), dbn[t].data_ptr(), ws.data_ptr(), B, T, Fq) == 0
    outs = []
    for rep in range(2):
        dw1, db1 = torch.full((nt, 64, 9), 0.5).cuda(), torch.full((nt, 64), 0.25).cuda()
        assert L.mtl_conv0_wgrad_tb(st(), x.data_ptr(), dy.data_ptr(), dw1.data_ptr(), db1.data_ptr(), ws.data_ptr(), B, T, Fq, nt,
                                    0 if shared_x else x[0].numel(), 576, 64) == 0
        torch.cuda.synchronize()
        outs.append((dw1, db1))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for t in range(nt):
        assert rel(outs[0][0][t] - 0.5, dwn[t] - 0.5) < 3e-6 and rel(outs[0][1][t] - 0.25, dbn[t] - 0.25) < 3e-6, t
    # bias column sums of nt row blocks (+ their max|X|)
    rows, cols = B * (T // 4) * (Fq // 4) + 3, 128
    X = torch.randn(nt, rows, cols, generator=g).cuda()
    per = ((L.mtl_colsum_workspace(rows, cols) // 4 + 3) // 4 * 4) * 4
    cws = torch.empty(nt * per // 4 + 16).cuda()
    on, o1 = torch.full((nt, 700), 1.5).cuda(), torch.full((nt, 700), 1.5).cuda()
    mn, m1 = torch.zeros(nt, S).cuda(), torch.zeros(nt, S).cuda()
    for t in range(nt):
        assert L.mtl_colsum_accum(st(), X[t].data_ptr(), rows, cols, cols, on[t].data_ptr(), cws.data_ptr(), mn[t].data_ptr()) == 0
    assert L.mtl_colsum_accum_tb(st(), X.data_ptr(), rows, cols, o1.data_ptr(), cws.data_ptr(), m1.data_ptr(), nt, 700, S) == 0
    torch.cuda.synchronize()
    assert torch.equal(o1, on) and torch.equal(m1, mn) and rel(o1[:, :cols] - 1.5, X.sum(1)) < 2e-5
    assert bool((o1[:, cols:] == 1.5).all())


@pytest.mark.parametrize('Cin,Cout,B,T,Fq,nt', [(64, 64, 2, 37, 161, 3), (64, 128, 3, 34, 80, 4), (128, 128, 2, 50, 40, 5), (64, 64, 8, 96, 161, 8)])
@pytest.mark.parametrize('shared_w', [True, False])
def test_conv3x3_two_piece_fp16_several_tasks_in_one_launch(L, Cin, Cout, B, T, Fq, nt, shared_w):
    """mtl_conv3x3_*_h2_tb: the samples of nt meta-tasks in ONE launch (per-task operand bounds, weights shared -- training passes at
    theta0 -- or per task -- validation passes at the theta' stack --, per-task bias and output bounds) against nt single-task launches:
    forward, fused pool (+ arg-max), both data gradients and every output bound must agree BIT FOR BIT (same tiles, same scales; a
    workgroup's tile sequence crosses task boundaries, also inside the multi-stage pipeline of the halo / weight waves)."""
    g = torch.Generator().manual_seed(Cin + Cout + T + nt)
    S = 2048
    mags = [10.0 ** (t - 1) for t in range(nt)]                                  # the tasks' tensors differ by orders of magnitude: a shared scale would show
    x = torch.cat([torch.relu(torch.randn(B, T, Fq, Cin, generator=g)) * m for m in mags]).cuda()
    nw = 1 if shared_w else nt
    w = torch.randn(nw, Cout, Cin, 3, 3, generator=g) * (1.0 / np.sqrt(9 * Cin))
    w = (w * torch.tensor([3.0 ** k for k in range(nw)]).view(-1, 1, 1, 1, 1)).cuda()
    bias = (torch.randn(nw, Cout, generator=g) * 0.1).cuda()
    nb = (L.mtl_conv3x3_wprep_h2_bytes(Cout, Cin) + 15) // 16 * 16
    w2f, w2d = torch.empty(nw, nb, dtype=torch.uint8).cuda(), torch.empty(nw, nb, dtype=torch.uint8).cuda()
    for k in range(nw):
        assert L.mtl_conv3x3_wprep_h2(st(), w[k].data_ptr(), w2f[k].data_ptr(), w2d[k].data_ptr(), Cout, Cin) == 0
    ax = torch.stack([x[t * B:(t + 1) * B].abs().max().reshape(1).repeat(S) for t in range(nt)]).contiguous()
    sW, sB = (0, 0) if shared_w else (nb, Cout)
    wk = lambda t: 0 if shared_w else t
    Tp, Fp = T // 2, Fq // 2
    for pooled in (False, True):
        shp = (nt * B, Tp, Fp, Cout) if pooled else (nt * B, T, Fq, Cout)
        y1, yn = torch.zeros(shp).cuda(), torch.zeros(shp).cuda()
        am1, amn = torch.zeros(shp, dtype=torch.uint8).cuda(), torch.zeros(shp, dtype=torch.uint8).cuda()
        ay1, ayn = torch.zeros(nt, S).cuda(), torch.zeros(nt, S).cuda()
        for t in range(nt):
            sl = slice(t * B, (t + 1) * B)
            if pooled:
                assert L.mtl_conv3x3_relu_pool_fwd_h2(st(), x[sl].data_ptr(), ax[t].data_ptr(), w2f[wk(t)].data_ptr(), bias[wk(t)].data_ptr(), yn[sl].data_ptr(),
                                                      amn[sl].data_ptr(), ayn[t].data_ptr(), B, T, Fq, Cin, Cout) == 0
            else:
                assert L.mtl_conv3x3_relu_fwd_h2(st(), x[sl].data_ptr(), ax[t].data_ptr(), w2f[wk(t)].data_ptr(), bias[wk(t)].data_ptr(), yn[sl].data_ptr(),
                                                 ayn[t].data_ptr(), B, T, Fq, Cin, Cout) == 0
        if pooled:
            assert L.mtl_conv3x3_relu_pool_fwd_h2_tb(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y1.data_ptr(), am1.data_ptr(),
                                                     ay1.data_ptr(), B, T, Fq, Cin, Cout, nt, sW, sB, S, S, None, 0) == 0
        else:
            assert L.mtl_conv3x3_relu_fwd_h2_tb(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y1.data_ptr(), ay1.data_ptr(),
                                                B, T, Fq, Cin, Cout, nt, sW, sB, S, S, None, 0) == 0
        torch.cuda.synchronize()
        assert torch.equal(y1, yn) and float(yn.abs().max()) > 0, ('forward', pooled)
        assert torch.equal(ay1.view(nt, -1, 32)[:, :, 0].max(1)[0], ayn.view(nt, -1, 32)[:, :, 0].max(1)[0]), ('output bound', pooled)
        if pooled:
            assert torch.equal(am1, amn)
        # data gradient of the same layer (pooled: the gradient lives on the pooled grid + arg-max)
        dy = torch.cat([torch.randn((B,) + shp[1:], generator=g) * m for m in reversed(mags)]).cuda()
        ady = torch.stack([dy[t * B:(t + 1) * B].abs().max().reshape(1).repeat(S) for t in range(nt)]).contiguous()
        dx1, dxn = torch.zeros_like(x), torch.zeros_like(x)
        ad1, adn = torch.zeros(nt, S).cuda(), torch.zeros(nt, S).cuda()
        for t in range(nt):
            sl = slice(t * B, (t + 1) * B)
            assert L.mtl_conv3x3_dgrad_h2(st(), dy[sl].data_ptr(), ady[t].data_ptr(), amn[sl].data_ptr() if pooled else None, w2d[wk(t)].data_ptr(),
                                          x[sl].data_ptr(), dxn[sl].data_ptr(), adn[t].data_ptr(), B, T, Fq, Cin, Cout) == 0
        assert L.mtl_conv3x3_dgrad_h2_tb(st(), dy.data_ptr(), ady.data_ptr(), amn.data_ptr() if pooled else None, w2d.data_ptr(), x.data_ptr(),
                                         dx1.data_ptr(), ad1.data_ptr(), B, T, Fq, Cin, Cout, nt, sW, S, S, None, 0) == 0
        torch.cuda.synchronize()
        assert torch.equal(dx1, dxn) and float(dxn.abs().max()) > 0, ('data gradient', pooled)
        assert torch.equal(ad1.view(nt, -1, 32)[:, :, 0].max(1)[0], adn.view(nt, -1, 32)[:, :, 0].max(1)[0]), ('dx bound', pooled)
        # tasks with frame counts of their own (`widths`): pixel-tile rows wholly beyond a task's frames are left out of the launch --
        # what is computed is bit for bit the full launch, what is left out keeps the buffer's previous content (the caller clears it)
        wd_host = [max(T - 7 * t - (3 if t else 0), 4) for t in range(nt)]
        wd = torch.tensor(wd_host, dtype=torch.int32).cuda()
        yw = torch.full(shp, -7.0).cuda()
        amw = torch.full(shp, 9, dtype=torch.uint8).cuda()
        ayw = torch.zeros(nt, S).cuda()
        if pooled:
            assert L.mtl_conv3x3_relu_pool_fwd_h2_tb(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), yw.data_ptr(), amw.data_ptr(),
                                                     ayw.data_ptr(), B, T, Fq, Cin, Cout, nt, sW, sB, S, S, wd.data_ptr(), 0) == 0
        else:
            assert L.mtl_conv3x3_relu_fwd_h2_tb(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), yw.data_ptr(), ayw.data_ptr(),
                                                B, T, Fq, Cin, Cout, nt, sW, sB, S, S, wd.data_ptr(), 0) == 0
        dxw = torch.full(x.shape, -7.0).cuda()
        assert L.mtl_conv3x3_dgrad_h2_tb(st(), dy.data_ptr(), ady.data_ptr(), amn.data_ptr() if pooled else None, w2d.data_ptr(), x.data_ptr(),
                                         dxw.data_ptr(), None, B, T, Fq, Cin, Cout, nt, sW, S, 0, wd.data_ptr(), 0) == 0
        torch.cuda.synchronize()
        left_out = 0
        for t in range(nt):
            sl = slice(t * B, (t + 1) * B)
            own = wd_host[t] // 2 if pooled else wd_host[t]                      # output rows of the task's own extent
            assert torch.equal(yw[sl, :own], yn[sl, :own]) and torch.equal(dxw[sl, :wd_host[t]], dxn[sl, :wd_host[t]]), (t, pooled)
            if pooled:
                assert torch.equal(amw[sl, :own], amn[sl, :own])
            edge = -(-wd_host[t] // 16) * 16                                     # (tile rows are 8 or 16 pixel rows)
            assert bool((dxw[sl, edge:] == -7.0).all()) and bool((yw[sl, (edge // 2 if pooled else edge):] == -7.0).all()), (t, pooled)
            left_out += int((dxw[sl] == -7.0).sum())
        assert left_out > 0 or T < 24
        # weight + bias gradients of all tasks in one launch (mtl_conv3x3_wgrad_h2_tb): the launch's partial slabs are dealt to the tasks,
        # so a task's pixels are partitioned over fewer slabs than in its own launch -- same arithmetic, another summation tree: equal to
        # the per-task launches within fp32 rounding (and to fp64 like them), accumulating onto the stack, deterministic
        if shared_w:
            need = L.mtl_conv3x3_wgrad_x3_workspace(B, T, Fq, Cin, Cout, 1 if pooled else 0)
            ws = torch.empty(need // 4 + 64).cuda()
            dwn, dbn = torch.full((nt, Cout, Cin, 3, 3), 0.5).cuda(), torch.full((nt, Cout), 0.25).cuda()
            dw1, db1 = dwn.clone(), dbn.clone()
            for t in range(nt):
                sl = slice(t * B, (t + 1) * B)
                assert L.mtl_conv3x3_wgrad_h2(st(), x[sl].data_ptr(), ax[t].data_ptr(), dy[sl].data_ptr(), ady[t].data_ptr(),
                                              amn[sl].data_ptr() if pooled else None, dwn[t].data_ptr(), dbn[t].data_ptr(), ws.data_ptr(), need,
                                              B, T, Fq, Cin, Cout) == 0
            for rep in range(2):
                dw_, db_ = (dw1, db1) if rep == 0 else (torch.full_like(dw1, 0.5), torch.full_like(db1, 0.25))
                assert L.mtl_conv3x3_wgrad_h2_tb(st(), x.data_ptr(), ax.data_ptr(), dy.data_ptr(), ady.data_ptr(), amn.data_ptr() if pooled else None,
                                                 dw_.data_ptr(), db_.data_ptr(), ws.data_ptr(), need, B, T, Fq, Cin, Cout, nt, S, S,
                                                 dw_[0].numel(), Cout) == 0
                torch.cuda.synchronize()
                if rep:
                    assert torch.equal(dw_, dw1) and torch.equal(db_, db1)          # deterministic
            for t in range(nt):
                assert rel(dw1[t] - 0.5, dwn[t] - 0.5) < 3e-6, ('dW', pooled, t, rel(dw1[t] - 0.5, dwn[t] - 0.5))
                assert rel(db1[t] - 0.25, dbn[t] - 0.25) < 3e-6, ('db', pooled, t)


@pytest.mark.parametrize('Cin,Cout,B,T,Fq,nt', [(64, 64, 2, 37, 161, 3), (64, 128, 3, 34, 80, 4), (128, 128, 2, 50, 40, 5)])
@pytest.mark.parametrize('shared_w', [True, False])
def test_conv3x3_exact_split_several_tasks_in_one_launch(L, Cin, Cout, B, T, Fq, nt, shared_w):
    """mtl_conv3x3_*_x3_tb (the exact 3 x bf16 split, MTL_CONV=x3): the samples of nt meta-tasks in ONE launch against nt single-task
    launches -- forward, fused pool (+ arg-max) and both data gradients BIT FOR BIT (weights shared or per task, per-task bias), rows
    beyond a task's own frames left out; weight + bias gradients of all tasks in one launch equal to the per-task launches within fp32
    rounding (other slab partition), accumulating onto the stack, deterministic."""
    g = torch.Generator().manual_seed(Cin + Cout + T + nt + 1)
    x = torch.cat([torch.relu(torch.randn(B, T, Fq, Cin, generator=g)) * 10.0 ** (t - 1) for t in range(nt)]).cuda()
    nw = 1 if shared_w else nt
    w = (torch.randn(nw, Cout, Cin, 3, 3, generator=g) * (1.0 / np.sqrt(9 * Cin))).cuda()
    bias = (torch.randn(nw, Cout, generator=g) * 0.1).cuda()
    w3f = torch.empty(nw, 3, 9, Cin, Cout, dtype=torch.bfloat16).cuda()
    w3d = torch.empty(nw, 3, 9, Cout, Cin, dtype=torch.bfloat16).cuda()
    for k in range(nw):
        assert L.mtl_conv3x3_wprep_x3(st(), w[k].data_ptr(), w3f[k].data_ptr(), w3d[k].data_ptr(), Cout, Cin) == 0
    sW, sB = (0, 0) if shared_w else (w3f[0].numel() * 2, Cout)
    wk = lambda t: 0 if shared_w else t
    Tp, Fp = T // 2, Fq // 2
    for pooled in (False, True):
        shp = (nt * B, Tp, Fp, Cout) if pooled else (nt * B, T, Fq, Cout)
        y1, yn = torch.zeros(shp).cuda(), torch.zeros(shp).cuda()
        am1, amn = torch.zeros(shp, dtype=torch.uint8).cuda(), torch.zeros(shp, dtype=torch.uint8).cuda()
        for t in range(nt):
            sl = slice(t * B, (t + 1) * B)
            if pooled:
                assert L.mtl_conv3x3_relu_pool_fwd_x3(st(), x[sl].data_ptr(), w3f[wk(t)].data_ptr(), bias[wk(t)].data_ptr(), yn[sl].data_ptr(),
                                                      amn[sl].data_ptr(), B, T, Fq, Cin, Cout) == 0
            else:
                assert L.mtl_conv3x3_relu_fwd_x3(st(), x[sl].data_ptr(), w3f[wk(t)].data_ptr(), bias[wk(t)].data_ptr(), yn[sl].data_ptr(),
                                                 B, T, Fq, Cin, Cout) == 0
        wd_host = [max(T - 7 * t - (3 if t else 0), 4) for t in range(nt)]
        wd = torch.tensor(wd_host, dtype=torch.int32).cuda()
        yw, amw = torch.full(shp, -7.0).cuda(), torch.full(shp, 9, dtype=torch.uint8).cuda()
        for widths, yo, ao in ((None, y1, am1), (wd.data_ptr(), yw, amw)):
            if pooled:
                assert L.mtl_conv3x3_relu_pool_fwd_x3_tb(st(), x.data_ptr(), w3f.data_ptr(), bias.data_ptr(), yo.data_ptr(), ao.data_ptr(),
                                                         B, T, Fq, Cin, Cout, nt, sW, sB, widths, 0) == 0
            else:
                assert L.mtl_conv3x3_relu_fwd_x3_tb(st(), x.data_ptr(), w3f.data_ptr(), bias.data_ptr(), yo.data_ptr(), B, T, Fq, Cin, Cout,
                                                    nt, sW, sB, widths, 0) == 0
        torch.cuda.synchronize()
        assert torch.equal(y1, yn) and float(yn.abs().max()) > 0, ('forward', pooled)
        if pooled:
            assert torch.equal(am1, amn)
        dy = torch.cat([torch.randn((B,) + shp[1:], generator=g) * 10.0 ** (1 - t) for t in range(nt)]).cuda()
        dx1, dxn, dxw = torch.zeros_like(x), torch.zeros_like(x), torch.full(x.shape, -7.0).cuda()
        for t in range(nt):
            sl = slice(t * B, (t + 1) * B)
            assert L.mtl_conv3x3_dgrad_x3(st(), dy[sl].data_ptr(), amn[sl].data_ptr() if pooled else None, w3d[wk(t)].data_ptr(), x[sl].data_ptr(),
                                          dxn[sl].data_ptr(), B, T, Fq, Cin, Cout) == 0
        for widths, dxo in ((None, dx1), (wd.data_ptr(), dxw)):
            assert L.mtl_conv3x3_dgrad_x3_tb(st(), dy.data_ptr(), amn.data_ptr() if pooled else None, w3d.data_ptr(), x.data_ptr(), dxo.data_ptr(),
                                             B, T, Fq, Cin, Cout, nt, sW, widths, 0) == 0
        torch.cuda.synchronize()
        assert torch.equal(dx1, dxn) and float(dxn.abs().max()) > 0, ('data gradient', pooled)
        for t in range(nt):
            sl = slice(t * B, (t + 1) * B)
            own = wd_host[t] // 2 if pooled else wd_host[t]
            assert torch.equal(yw[sl, :own], yn[sl, :own]) and torch.equal(dxw[sl, :wd_host[t]], dxn[sl, :wd_host[t]]), (t, pooled)
            edge = -(-wd_host[t] // 16) * 16
            assert bool((dxw[sl, edge:] == -7.0).all()) and bool((yw[sl, (edge // 2 if pooled else edge):] == -7.0).all()), (t, pooled)
        if shared_w:
            need = L.mtl_conv3x3_wgrad_x3_workspace(B, T, Fq, Cin, Cout, 1 if pooled else 0)
            ws = torch.empty(need // 4 + 64).cuda()
            dwn, dw1, db1 = torch.full((nt, Cout, Cin, 3, 3), 0.5).cuda(), torch.full((nt, Cout, Cin, 3, 3), 0.5).cuda(), torch.full((nt, Cout), 0.25).cuda()
            for t in range(nt):
                sl = slice(t * B, (t + 1) * B)
                assert L.mtl_conv3x3_wgrad_x3(st(), x[sl].data_ptr(), dy[sl].data_ptr(), amn[sl].data_ptr() if pooled else None, dwn[t].data_ptr(),
                                              ws.data_ptr(), need, B, T, Fq, Cin, Cout) == 0
            outs = []
            for rep in range(2):
                dw_, db_ = (dw1, db1) if rep == 0 else (torch.full_like(dw1, 0.5), torch.full_like(db1, 0.25))
                assert L.mtl_conv3x3_wgrad_x3_tb(st(), x.data_ptr(), dy.data_ptr(), amn.data_ptr() if pooled else None, dw_.data_ptr(), db_.data_ptr(),
                                                 ws.data_ptr(), need, B, T, Fq, Cin, Cout, nt, dw_[0].numel(), Cout) == 0
                torch.cuda.synchronize()
                if rep:
                    assert torch.equal(dw_, dw1) and torch.equal(db_, db1)
            for t in range(nt):
                sl = slice(t * B, (t + 1) * B)
                assert rel(dw1[t] - 0.5, dwn[t] - 0.5) < 3e-6, ('dW', pooled, t)
                if pooled:      # the bias gradient = sums of the un-pooled dy = sums of dy on the pooled grid
                    ref_b = dy[sl].double().sum((0, 1, 2))
                else:
                    ref_b = dy[sl].double().sum((0, 1, 2))
                assert rel(db1[t] - 0.25, ref_b) < 3e-6, ('db', pooled, t)


@pytest.mark.parametrize('Cin,Cout,B,T,Fq', [(64, 64, 2, 21, 161), (64, 128, 2, 18, 80), (128, 128, 1, 9, 19)])
@pytest.mark.parametrize('mag', [1.0, 3e-7, 4e5])
def test_conv3x3_two_piece_fp16_is_fp32_class(L, Cin, Cout, B, T, Fq, mag):
    """the "h2" kernels (2-way fp16 split of power-of-two-scaled operands, three fp16 MFMAs per product): measured against an
    fp64 convolution their error is that of the exact-fp32 MFMA kernels of this library on the same data (0.9x forward, at most
    1.6x on the sparse pooled gradients; both are dominated by the fp32 accumulation chain, the split itself is good to 2^-22)
    and below 1e-6 normwise, at any operand magnitude and with loose amax bounds: forward, fused pool, both dgrads, both wgrads."""
    g = torch.Generator().manual_seed(Cin + Cout + T + 2)
    x = torch.relu(torch.randn(B, Cin, Fq, T, generator=g)) * mag
    x[0, 0, 0, 0] = 40.0 * mag                                 # an outlier 10x above the bulk: small values keep their bits
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / np.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g) * 0.1 * mag
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = torch.relu(F.conv2d(x64, w64, b.double(), padding=1))
    p64, idx = F.max_pool2d(y64, 2, stride=2, return_indices=True)
    dxn, dw, db = dev(nhwc(x)), dev(w), dev(b)
    wf, wd = torch.empty(9, Cin, Cout).cuda(), torch.empty(9, Cout, Cin).cuda()
    assert L.mtl_conv3x3_wprep(st(), dw.data_ptr(), wf.data_ptr(), wd.data_ptr(), Cout, Cin) == 0
    nb = L.mtl_conv3x3_wprep_h2_bytes(Cout, Cin)
    w2f, w2d = torch.empty(nb, dtype=torch.uint8).cuda(), torch.empty(nb, dtype=torch.uint8).cuda()
    assert L.mtl_conv3x3_wprep_h2(st(), dw.data_ptr(), w2f.data_ptr(), w2d.data_ptr(), Cout, Cin) == 0
    # the two pieces re-assemble every weight to 2^-21 of the largest; the trailer is the power-of-two scale
    nw = 2 * 9 * Cin * Cout * 2
    scale = float(w2f[nw:nw + 4].view(torch.float32))
    assert 2 ** 14 <= float(w.abs().max()) * scale < 2 ** 15 and np.log2(scale) == int(np.log2(scale))
    pieces = (w2f[:nw].view(torch.float16).view(2, 9, Cin // 32, Cout, 4, 8).double().sum(0) / scale).cpu()
    unsw = torch.empty_like(pieces)
    for co in range(Cout):
        for c in range(4):
            unsw[:, :, co, c] = pieces[:, :, co, c ^ ((co >> 2) & 3)]
    wt = w.double().reshape(Cout, Cin // 32, 32, 9).permute(3, 1, 0, 2)
    assert float((unsw.reshape(9, Cin // 32, Cout, 32) - wt).abs().max()) <= 2.0 ** -21 * float(w.abs().max())
    S = 2048                                                   # MTL_AMAX_FLOATS: a bound is 64 slot heads (32 floats apart)
    ax = dxn.abs().max().reshape(1).repeat(S)
    ax[32:] = 0                                                # ... any ONE head carrying the bound is enough
    slots = torch.zeros(4 * S).cuda()
    report = []

    def cmp(what, got_h2, got_f32, want):
        e2, e32 = rel(got_h2.double().cpu(), want.detach()), rel(got_f32.double().cpu(), want.detach())
        report.append('%s: h2 %.2e, fp32 MFMA kernel %.2e' % (what, e2, e32))
        assert e2 < 2.0 * e32 + 1e-30 and e2 < 1e-6, report

    y, yf = torch.empty(B, T, Fq, Cout).cuda(), torch.empty(B, T, Fq, Cout).cuda()
    assert L.mtl_conv3x3_relu_fwd(st(), dxn.data_ptr(), wf.data_ptr(), db.data_ptr(), yf.data_ptr(), B, T, Fq, Cin, Cout) == 0
    assert L.mtl_conv3x3_relu_fwd_h2(st(), dxn.data_ptr(), ax.data_ptr(), w2f.data_ptr(), db.data_ptr(), y.data_ptr(),
                                     slots.data_ptr(), B, T, Fq, Cin, Cout) == 0
    cmp('fwd', from_nhwc(y), from_nhwc(yf), y64)
    assert float(y.max()) <= float(slots[:S].max()) <= 8.0 * float(y.max())     # amax_y: an upper bound (max|acc| + max|bias|) within a few bits
    loose = ax.roll(63 * 32) * 64.0                                                      # a bound 6 bits too high costs nothing
    y2 = torch.empty_like(y)
    assert L.mtl_conv3x3_relu_fwd_h2(st(), dxn.data_ptr(), loose.data_ptr(), w2f.data_ptr(), db.data_ptr(), y2.data_ptr(), None, B, T, Fq,
                                     Cin, Cout) == 0
    cmp('fwd, amax x64', from_nhwc(y2), from_nhwc(yf), y64)
    Tp, Fp = T // 2, Fq // 2
    p, pf = torch.empty(B, Tp, Fp, Cout).cuda(), torch.empty(B, Tp, Fp, Cout).cuda()
    am, amf = torch.empty(B, Tp, Fp, Cout, dtype=torch.uint8).cuda(), torch.empty(B, Tp, Fp, Cout, dtype=torch.uint8).cuda()
    assert L.mtl_conv3x3_relu_pool_fwd(st(), dxn.data_ptr(), wf.data_ptr(), db.data_ptr(), pf.data_ptr(), amf.data_ptr(), B, T, Fq, Cin, Cout) == 0
    assert L.mtl_conv3x3_relu_pool_fwd_h2(st(), dxn.data_ptr(), ax.data_ptr(), w2f.data_ptr(), db.data_ptr(), p.data_ptr(),
                                          am.data_ptr(), slots[S:].data_ptr(), B, T, Fq, Cin, Cout) == 0
    cmp('pool', from_nhwc(p), from_nhwc(pf), p64)
    assert float(p.max()) <= float(slots[S:2 * S].max())
    amc = from_nhwc(am).long().cpu()
    flat = (torch.arange(Fp).view(1, 1, Fp, 1) * 2 + (amc >> 1)) * T + torch.arange(Tp).view(1, 1, 1, Tp) * 2 + (amc & 1)
    mism = flat != idx
    if int(mism.sum()):      # arg-max decisions equal the fp64 ones except where the two candidates tie to fp32 precision
        ydense = y64.detach().reshape(B, Cout, -1)
        a = torch.gather(ydense, 2, flat.reshape(B, Cout, -1))
        bb = torch.gather(ydense, 2, idx.reshape(B, Cout, -1))
        assert float(((a - bb).abs() / (bb.abs() + 1e-30))[mism.reshape(B, Cout, -1)].max()) < 1e-5
    gate = (x > 0).double()
    dx, dxf = torch.empty(B, T, Fq, Cin).cuda(), torch.empty(B, T, Fq, Cin).cuda()
    if int(mism.sum()) == 0:      # pooled backward (needs identical pooling decisions to have ONE truth)
        dp = torch.randn(p64.shape, generator=g) * 1e-3 * mag
        p64.backward(dp.double(), retain_graph=True)
        dpn = dev(nhwc(dp * (p64.detach() > 0).float()))
        adp = dpn.abs().max().reshape(1).repeat(S)
        assert L.mtl_conv3x3_dgrad(st(), dpn.data_ptr(), am.data_ptr(), wd.data_ptr(), dxn.data_ptr(), dxf.data_ptr(), B, T, Fq, Cin, Cout) == 0
        assert L.mtl_conv3x3_dgrad_h2(st(), dpn.data_ptr(), adp.data_ptr(), am.data_ptr(), w2d.data_ptr(), dxn.data_ptr(), dx.data_ptr(),
                                      None, B, T, Fq, Cin, Cout) == 0
        cmp('dgrad (pooled)', from_nhwc(dx), from_nhwc(dxf), x64.grad * gate)
        need, needf = L.mtl_conv3x3_wgrad_x3_workspace(B, T, Fq, Cin, Cout, 1), L.mtl_conv3x3_wgrad_workspace(B, T, Fq, Cin, Cout, 1)
        ws = torch.empty(max(need, needf) // 4 + 16).cuda()
        wg, wgf = torch.zeros(Cout, Cin, 3, 3).cuda(), torch.zeros(Cout, Cin, 3, 3).cuda()
        assert L.mtl_conv3x3_wgrad(st(), dxn.data_ptr(), dpn.data_ptr(), am.data_ptr(), wgf.data_ptr(), ws.data_ptr(), needf, B, T, Fq, Cin, Cout) == 0
        dbp = torch.zeros(Cout).cuda()
        assert L.mtl_conv3x3_wgrad_h2(st(), dxn.data_ptr(), ax.data_ptr(), dpn.data_ptr(), adp.data_ptr(), am.data_ptr(), wg.data_ptr(),
                                      dbp.data_ptr(), ws.data_ptr(), need, B, T, Fq, Cin, Cout) == 0
        cmp('wgrad (pooled)', wg, wgf, w64.grad)
        assert rel(dbp.double().cpu(), dpn.double().sum((0, 1, 2)).cpu()) < 1e-5           # the bias gradient rides along
        x64.grad = None
        w64.grad = None
    # dense backward
    dy = torch.randn(y64.shape, generator=g) * 1e-3 * mag
    y64.backward(dy.double())
    dyn = dev(nhwc(dy * (y64.detach() > 0).float()))
    ady = (dyn.abs().max() * 3.0).reshape(1).repeat(S)          # an upper BOUND is enough
    assert L.mtl_conv3x3_dgrad(st(), dyn.data_ptr(), None, wd.data_ptr(), dxn.data_ptr(), dxf.data_ptr(), B, T, Fq, Cin, Cout) == 0
    assert L.mtl_conv3x3_dgrad_h2(st(), dyn.data_ptr(), ady.data_ptr(), None, w2d.data_ptr(), dxn.data_ptr(), dx.data_ptr(), slots[2 * S:].data_ptr(),
                                  B, T, Fq, Cin, Cout) == 0
    cmp('dgrad', from_nhwc(dx), from_nhwc(dxf), x64.grad * gate)
    assert float(dx.abs().max()) <= float(slots[2 * S:3 * S].max())
    need, needf = L.mtl_conv3x3_wgrad_x3_workspace(B, T, Fq, Cin, Cout, 0), L.mtl_conv3x3_wgrad_workspace(B, T, Fq, Cin, Cout, 0)
    ws = torch.empty(max(need, needf) // 4 + 16).cuda()
    wg, wgf = torch.zeros(Cout, Cin, 3, 3).cuda(), torch.zeros(Cout, Cin, 3, 3).cuda()
    assert L.mtl_conv3x3_wgrad(st(), dxn.data_ptr(), dyn.data_ptr(), None, wgf.data_ptr(), ws.data_ptr(), needf, B, T, Fq, Cin, Cout) == 0
    dbd = torch.zeros(Cout).cuda()
    assert L.mtl_conv3x3_wgrad_h2(st(), dxn.data_ptr(), ax.data_ptr(), dyn.data_ptr(), ady.data_ptr(), None, wg.data_ptr(), dbd.data_ptr(),
                                  ws.data_ptr(), need, B, T, Fq, Cin, Cout) == 0
    cmp('wgrad', wg, wgf, w64.grad)
    assert rel(dbd.double().cpu(), dyn.double().sum((0, 1, 2)).cpu()) < 1e-5
    assert L.mtl_conv3x3_wgrad_h2(st(), dxn.data_ptr(), ax.data_ptr(), dyn.data_ptr(), ady.data_ptr(), None, wg.data_ptr(), dbd.data_ptr(),
                                  ws.data_ptr(), need, B, T, Fq, Cin, Cout) == 0
    assert rel(dbd.double().cpu(), 2 * dyn.double().sum((0, 1, 2)).cpu()) < 1e-5             # db accumulates
    # missing scalars are refused
    assert L.mtl_conv3x3_relu_fwd_h2(st(), dxn.data_ptr(), None, w2f.data_ptr(), db.data_ptr(), y.data_ptr(), None, B, T, Fq, Cin, Cout) != 0
    print(report)


def test_absmax_and_colsum_amax(L):
    g = torch.Generator().manual_seed(5)
    X = torch.randn(3000, 128, generator=g)
    X[1234, 77] = -9.5
    dX = dev(X)
    slot = torch.zeros(4096).cuda()
    assert L.mtl_absmax_f32(st(), dX.data_ptr(), X.numel(), slot.data_ptr()) == 0
    out = torch.zeros(128).cuda()
    ws = torch.empty(L.mtl_colsum_workspace(3000, 128) // 4).cuda()
    assert L.mtl_colsum_accum(st(), dX.data_ptr(), 3000, 128, 128, out.data_ptr(), ws.data_ptr(), slot[2048:].data_ptr()) == 0
    assert float(slot[:2048].max()) == 9.5 and bool((slot[2048::32] == 9.5).all())
    assert rel(out, X.sum(0)) < 1e-5
    # the 16-byte form (contiguous rows of 4 x 2^k columns: the conv bias gradients) at the north-star extent of conv.7.bias, a ragged
    # row count, with and without the max|X| by-product, accumulating onto `out`; and shapes that stay on the scalar form
    for rows, cols, with_amax in ((80000, 128, True), (77777, 64, False), (1003, 256, True), (4999, 16, False), (3000, 100, False), (3000, 192, True)):
        X = torch.randn(rows, cols, generator=g)
        X[rows // 3, cols // 2] = 11.25
        dX = dev(X)
        out = torch.full((cols,), 2.0).cuda()
        am = torch.zeros(2048).cuda()
        ws = torch.empty(L.mtl_colsum_workspace(rows, cols) // 4).cuda()
        assert L.mtl_colsum_accum(st(), dX.data_ptr(), rows, cols, cols, out.data_ptr(), ws.data_ptr(), am.data_ptr() if with_amax else None) == 0
        assert rel(out - 2.0, X.double().sum(0).float()) < 2e-5, (rows, cols)
        if with_amax:
            assert bool((am[::32] == 11.25).all()), (rows, cols)
        out2 = torch.full((cols,), 2.0).cuda()
        assert L.mtl_colsum_accum(st(), dX.data_ptr(), rows, cols, cols, out2.data_ptr(), ws.data_ptr(), None) == 0
        assert torch.equal(out, out2)                                   # deterministic


def test_spectrogram_front_end_matches_oracle(L, tmp_path):
    """wav -> STFT (DFT-as-GEMM) -> log1p|.| -> normalise on the device vs the numpy restatement."""
    import wave
    import mtl_amd
    from oracle import frontend
    rng = np.random.RandomState(0)
    t = np.arange(16000 * 2 + 37) / 16000.0
    y = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t * (1 + 0.1 * t)) + 0.05 * rng.randn(t.size)).astype(np.float32)
    fe = mtl_amd.SpectrogramFrontEnd(16000, 0.02, 0.01, 'hamming', normalize=True)
    got = fe(y).cpu()
    ref = frontend.parse_audio(y)
    assert got.shape == ref.shape == (161, 1 + y.size // 160)
    assert rel(got, ref) < 2e-5                       # fp32 DFT with K = 320 vs float32 FFT
    raw_fe = mtl_amd.SpectrogramFrontEnd(16000, 0.02, 0.01, 'hamming', normalize=False)
    raw = raw_fe(y).cpu()
    assert rel(raw, frontend.parse_audio(y, normalize=False)) < 2e-5
    # 16-bit PCM wav path (utils/audio.py:7-15)
    p = str(tmp_path / 'a.wav')
    with wave.open(p, 'wb') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((np.clip(y, -1, 1) * 32767).astype('<i2').tobytes())
    yw = mtl_amd.load_wav_pcm16(p)
    assert abs(yw - np.clip(y, -1, 1)).max() < 1e-4 and rel(fe(yw).cpu(), frontend.parse_audio(yw)) < 2e-5
    # the committed fixture of the restatement (tests/golden/S0.npz; generated without librosa: the row stays parity-unpinned)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'S0.npz'))
    assert rel(fe(z['waveform']).cpu(), torch.from_numpy(z['spect_norm'])) < 2e-5
    assert rel(raw_fe(z['waveform']).cpu(), torch.from_numpy(z['spect_raw'])) < 2e-5


@pytest.mark.parametrize('transB,M,N,K,tasks,shared,gate,bias', [(1, 300, 512, 640, 3, True, False, True), (0, 300, 640, 512, 3, True, True, False),
                                                                 (1, 2000, 512, 5120, 2, False, False, True), (0, 257, 132, 100, 2, False, True, False),
                                                                 (1, 2000, 512, 5120, 1, False, False, True), (0, 300, 512, 2100, 1, True, False, False)])
def test_gemm_two_piece_fp16_task_batched(L, transB, M, N, K, tasks, shared, gate, bias):
    """mtl_gemm_h2_tb (the tile engine of mtl_gemm_x3.hip with fp16 pairs): per-task operands, bounds and biases by stride, shared or
    per-task weights, both weight orientations, gate, ragged M / N / K; against fp64 no less accurate than 2x the exact-fp32 engine of
    this library, bitwise reproducible; a quiet task beside a loud one keeps its accuracy (the scale is per task).  ONE task with few
    output tiles and K >= 2048 (the input Linear of a rank that holds a single task): the K range is split over the grid into the
    workspace (ragged last slice: 5120 = 4 x 1280, 2100 = 2 x 1056 - 12) and summed in a fixed order -- same bars."""
    g = torch.Generator().manual_seed(M + N + K + tasks)
    A = torch.randn(tasks, M, K, generator=g) * 3e-3
    if tasks > 1:
        A[1] *= 2.0 ** -12                                   # a quiet task: its own bound, its own scale
    nb = 1 if shared else tasks
    Bm = torch.randn(nb, N, K, generator=g) * 0.5 if transB else torch.randn(nb, K, N, generator=g) * 0.5
    bv = torch.randn(tasks, N, generator=g) * 1e-4 if bias else None
    gt = torch.randn(tasks, M, N, generator=g) if gate else None
    opB = Bm.double().transpose(1, 2) if transB else Bm.double()
    want = A.double() @ (opB.expand(tasks, -1, -1) if shared else opB)
    if bias:
        want = want + bv.double().unsqueeze(1)
    if gate:
        want = want * (gt > 0).double()
    dA, dB = dev(A), dev(Bm)
    S = 2048
    aa, ab = torch.zeros(tasks, S).cuda(), torch.zeros(nb, S).cuda()
    for t in range(tasks):
        assert L.mtl_absmax_f32(st(), dA[t].data_ptr(), A[t].numel(), aa[t].data_ptr()) == 0
    for t in range(nb):
        assert L.mtl_absmax_f32(st(), dB[t].data_ptr(), Bm[t].numel(), ab[t].data_ptr()) == 0
    dbv, dgt = (dev(bv) if bias else None), (dev(gt) if gate else None)
    C, C2 = torch.full((tasks, M, N), 7.0).cuda(), torch.empty(tasks, M, N).cuda()
    wsk = torch.empty(8 << 20).cuda()
    args = lambda out: (st(), transB, M, N, K, dA.data_ptr(), K, aa.data_ptr(), S, dB.data_ptr(), Bm.shape[2], ab.data_ptr(), 0 if shared else S,
                        out.data_ptr(), N, dbv.data_ptr() if bias else None, dgt.data_ptr() if gate else None, N, tasks, M * K,
                        0 if shared else Bm[0].numel(), M * N, N if bias else 0, wsk.data_ptr(), wsk.numel() * 4)
    assert L.mtl_gemm_h2_tb(*args(C)) == 0
    assert L.mtl_gemm_h2_tb(*args(C2)) == 0
    assert torch.equal(C, C2)
    old = L.mtl_gemm_x3_min_tiles(0)                          # the exact-fp32 engines as the yardstick
    F32 = torch.empty(tasks, M, N).cuda()
    ws = torch.empty(16 << 20).cuda()
    try:
        assert L.mtl_gemm_f32_tb(st(), 0, transB, M, N, K, 1.0, dA.data_ptr(), K, dB.data_ptr(), Bm.shape[2], F32.data_ptr(), N,
                                 dbv.data_ptr() if bias else None, dgt.data_ptr() if gate else None, N, 0, tasks, 1, 0, 0, 0, 0, 0, 0, 0, 1,
                                 0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0, tasks, M * K, 0 if shared else Bm[0].numel(), M * N,
                                 N if bias else 0, 0) == 0
    finally:
        L.mtl_gemm_x3_min_tiles(old)
    for t in range(tasks):
        e2, e32 = rel(C[t].double().cpu(), want[t]), rel(F32[t].double().cpu(), want[t])
        assert e2 < 2.0 * e32 + 1e-30 and e2 < 1e-6, (t, e2, e32)


@pytest.fixture
def x3_forced(L):
    """every eligible product of mtl_gemm_f32_ex / _tb goes to the bf16-split engine (threshold 1 tile) for the duration of a test"""
    old = L.mtl_gemm_x3_min_tiles(1)
    yield
    L.mtl_gemm_x3_min_tiles(old)


@pytest.mark.parametrize('ta,tb', [(0, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize('M,N,K', [(101, 252, 64), (808, 100, 512), (100, 512, 2000), (33, 36, 7), (512, 100, 808), (5, 3768, 301),
                                   (2000, 512, 100), (257, 129 * 4, 33)])
def test_bf16_split_engine_contract(L, x3_forced, ta, tb, M, N, K):
    """csrc/mtl_gemm_x3.hip with the routing threshold forced to one tile: both tile configurations, the three operand orientations
    of the pass, ragged M / N / K (K tails inside a quad for the K-major forms, partial 16-byte quads at the row end of the MN-major
    ones), bias + ReLU + gate + accumulate + alpha, two batch levels with strided operands, against fp64 at the tolerance of the
    fp32 engines' tests; bitwise repeatable."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    lda = ((M if ta else K) + 3) // 4 * 4                   # 16-byte aligned rows (the engine's eligibility rule), extents may be ragged
    ldb = ((K if tb else N) + 3) // 4 * 4
    nz, H = 4, 2
    A = torch.randn(nz, (K if ta else M), lda, generator=g)
    B = torch.randn(nz, (N if tb else K), ldb, generator=g)
    bias, C0, gate = torch.randn(nz, N, generator=g), torch.randn(nz, M, N, generator=g), torch.randn(nz, M, N, generator=g)
    opA = (A[:, :, :M].transpose(1, 2) if ta else A[:, :, :K]).double()
    opB = (B[:, :, :K].transpose(1, 2) if tb else B[:, :, :N]).double()
    prod = opA @ opB
    dA, dB, dbias, dgate = dev(A), dev(B), dev(bias), dev(gate)
    assert L.mtl_gemm_f32_ex_route(M, N, K, nz, 1, 0) == 2
    sA, sB = A[0].numel(), B[0].numel()
    C = dev(C0.clone())
    assert _gemm_ex(L, ta, tb, M, N, K, dA, lda, dB, ldb, C, N, batch=nz, H=H, sA=(H * sA, sA), sB=(H * sB, sB), sC=(H * M * N, M * N)) == 0
    assert rel(C, prod) < 2e-6
    outs = []
    for _ in range(2):
        C = dev(C0.clone())
        assert _gemm_ex(L, ta, tb, M, N, K, dA, lda, dB, ldb, C, N, bias=dbias, gate=dgate, ldg=N, flags=3, alpha=0.5, batch=nz, H=H,
                        sA=(H * sA, sA), sB=(H * sB, sB), sC=(H * M * N, M * N), sbias=H * N, sbias_h=N) == 0
        outs.append(C.cpu())
    want = torch.relu(0.5 * prod + bias.double().unsqueeze(1)) * (gate > 0) + C0.double()
    assert rel(outs[0], want) < 2e-6
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0)])
def test_bf16_split_engine_split_k(L, ta, tb):
    """few output tiles x very long K (the LM decoder's dX, lm/model/rnn_model.py:56 backward: 700 x 512 x 10000): with a workspace
    the product runs as K slices on the bf16-split engine + a fixed-order sum (bias once, += C when accumulating); without a
    workspace, with a K that has no admissible slicing, or with ReLU it takes the other engines -- same results to fp32 rounding;
    bitwise repeatable."""
    g = torch.Generator().manual_seed(31 + ta * 2 + tb)
    ws = torch.empty(4 << 20).cuda()
    for M, N, K, flags in ((700, 512, 10000, 0), (300, 128, 4096, 2), (700, 512, 4099 * 2, 0), (260, 256, 8192, 1)):
        A = torch.randn((K, M) if ta else (M, K), generator=g)
        B = torch.randn((N, K) if tb else (K, N), generator=g)
        bias, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
        want = (A.t() if ta else A).double() @ (B.t() if tb else B).double() + bias.double()
        if flags & 1:
            want = want.clamp_min(0)
        if flags & 2:
            want = want + C0.double()
        dA, dB, db = dev(A), dev(B), dev(bias)
        outs = []
        for w in (ws, ws, None):
            C = dev(C0.clone())
            assert L.mtl_gemm_f32_ex(st(), ta, tb, M, N, K, 1.0, dA.data_ptr(), A.shape[1], dB.data_ptr(), B.shape[1], C.data_ptr(), N,
                                     db.data_ptr(), None, 0, flags, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, None, 0,
                                     w.data_ptr() if w is not None else None, w.numel() * 4 if w is not None else 0, 0, 0) == 0
            assert rel(C, want) < 3e-6, (M, N, K, flags, w is None)
            outs.append(C)
        assert torch.equal(outs[0], outs[1])


def test_bf16_split_engine_k_batching_row_sums_and_tasks(L, x3_forced):
    """the remaining parts of the contract on the bf16-split engine: C += sum_z A_z . B_z inside one launch with a ragged K
    (the masked main loop), row sums of op(A) as a by-product of transposed-A products (both tile configurations), and the third
    (task) batch level with per-task weights, outputs and biases."""
    test_gemm_k_batching_and_row_sums(L)                   # same assertions as on the fp32 engines, now routed to the split engine
    g = torch.Generator().manual_seed(9)
    for M, N, K, nz in ((512, 100, 808, 3), (100, 256, 301, 2)):            # 256-row and 128-row tiles, ragged K
        dy = torch.randn(nz, K, M, generator=g)
        x = torch.randn(nz, K, N, generator=g)
        W0, b0 = torch.randn(nz, M, N, generator=g), torch.randn(nz, M, generator=g)
        dW, db = dev(W0.clone()), dev(b0.clone())
        assert L.mtl_gemm_f32_ex_route(M, N, K, nz, 1, 1) == 2
        assert _gemm_ex(L, 1, 0, M, N, K, dev(dy), M, dev(x), N, dW, N, flags=2, batch=nz, sA=(K * M, 0), sB=(K * N, 0), sC=(M * N, 0),
                        rowsum=db, srow=M) == 0
        assert rel(dW, W0.double() + dy.double().transpose(1, 2) @ x.double()) < 2e-6
        assert rel(db, b0.double() + dy.double().sum(1)) < 3e-6
    tasks, per, M, N, K = 3, 2, 300, 200, 96
    A, Bm = torch.randn(tasks, per, M, K, generator=g), torch.randn(tasks, N, K, generator=g)
    bias = torch.randn(tasks, N, generator=g)
    C = torch.empty(tasks, per, M, N).cuda()
    ws = torch.empty(1 << 20).cuda()
    assert L.mtl_gemm_f32_tb(st(), 0, 1, M, N, K, 1.0, dev(A).data_ptr(), K, dev(Bm).data_ptr(), K, C.data_ptr(), N, dev(bias).data_ptr(),
                             None, 0, 0, tasks * per, 1, M * K, 0, 0, 0, M * N, 0, 0, 1, 0, 0, None, 0, ws.data_ptr(), ws.numel() * 4, 0, 0,
                             tasks, per * M * K, N * K, per * M * N, N, 0) == 0
    want = A.double() @ Bm.double().transpose(1, 2).unsqueeze(1) + bias.double().view(tasks, 1, 1, N)
    assert rel(C, want) < 2e-6


@pytest.mark.parametrize('Cin,Cout,B,T,Fq', [(64, 64, 2, 21, 161), (128, 128, 1, 30, 37), (64, 64, 1, 8, 16), (128, 128, 2, 17, 80)])
def test_conv3x3_pooled_wgrad_on_the_sparse_matrix_cores_matches_fp64(L, Cin, Cout, B, T, Fq):
    """mtl_conv3x3_wgrad_h2 with an arg-max map runs on v_smfmac_f32_32x32x32_f16 (conv3x3_wgrad_sp_kernel: the un-pooled gradient
    has one non-zero per 2x2 window and channel, stored as {value, 0} + the arg-max code as the 2:4 index).  ARBITRARY arg-max codes
    and pooled gradients (not tied to a forward, so every code / window / edge combination occurs), ragged extents (odd T and F:
    the last row / column has no window; tiles that are cut by the edge), against the fp64 weight gradient of the dense un-pooled
    gradient; the bias gradient rides along; results accumulate into dw / db."""
    g = torch.Generator().manual_seed(Cin + 3 * Cout + T + Fq)
    x = torch.relu(torch.randn(B, Cin, Fq, T, generator=g))
    Tp, Fp = T // 2, Fq // 2
    dp = torch.randn(B, Cout, Fp, Tp, generator=g) * 1e-2
    dp[torch.rand(dp.shape, generator=g) < 0.3] = 0.0                       # ReLU-gated entries
    am = torch.randint(0, 4, (B, Cout, Fp, Tp), generator=g)                # code = (f & 1) << 1 | (t & 1)
    dy = torch.zeros(B, Cout, Fq, T, dtype=torch.float64)
    bi, ci, fi, ti = torch.meshgrid(torch.arange(B), torch.arange(Cout), torch.arange(Fp), torch.arange(Tp), indexing='ij')
    dy[bi, ci, 2 * fi + (am >> 1), 2 * ti + (am & 1)] = dp.double()
    want = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy, padding=1)
    dxn, dpn, amn = dev(nhwc(x)), dev(nhwc(dp)), dev(nhwc(am.to(torch.uint8)))
    S = 2048
    ax, adp = dxn.abs().max().reshape(1).repeat(S), dpn.abs().max().reshape(1).repeat(S)
    need = L.mtl_conv3x3_wgrad_x3_workspace(B, T, Fq, Cin, Cout, 1)
    ws = torch.empty(need // 4 + 16).cuda()
    wg, dbp = torch.zeros(Cout, Cin, 3, 3).cuda(), torch.zeros(Cout).cuda()
    for rep in (1, 2):                                                       # the second call accumulates
        assert L.mtl_conv3x3_wgrad_h2(st(), dxn.data_ptr(), ax.data_ptr(), dpn.data_ptr(), adp.data_ptr(), amn.data_ptr(), wg.data_ptr(),
                                      dbp.data_ptr(), ws.data_ptr(), need, B, T, Fq, Cin, Cout) == 0
        assert rel(wg.double().cpu(), rep * want) < 1e-6, (rep, rel(wg.double().cpu(), rep * want))
        assert rel(dbp.double().cpu(), rep * dp.double().sum((0, 2, 3))) < 1e-5
    # per-tap check (a mis-routed tap or window position would hide in a norm over a random tensor far less than in its own slice)
    for kh in range(3):
        for kw in range(3):
            assert rel(wg[:, :, kh, kw].double().cpu(), 2 * want[:, :, kh, kw]) < 2e-6, (kh, kw)


@pytest.mark.parametrize('M,N,K,tasks', [(512, 5120, 2000, 2), (512, 640, 300, 3), (132, 252, 77, 1)])
def test_gemm_two_piece_fp16_transposed_a(L, M, N, K, tasks):
    """mtl_gemm_h2_tn_tb: C_t = A_t^T B_t on fp16 pairs (the input Linear's weight gradient dW = de0^T p2, modules/encoder.py:72 backward):
    per-task operands and bounds, ragged M / N / K; against fp64 fp32-class, bitwise repeatable; a quiet task keeps its accuracy."""
    g = torch.Generator().manual_seed(M + N + K + tasks)
    A = torch.randn(tasks, K, M, generator=g) * 2e-3
    Bm = torch.relu(torch.randn(tasks, K, N, generator=g))
    if tasks > 1:
        A[1] *= 2.0 ** -10
    want = A.double().transpose(1, 2) @ Bm.double()
    dA, dB = dev(A), dev(Bm)
    S = 2048
    aa, ab = torch.zeros(tasks, S).cuda(), torch.zeros(tasks, S).cuda()
    for t in range(tasks):
        assert L.mtl_absmax_f32(st(), dA[t].data_ptr(), A[t].numel(), aa[t].data_ptr()) == 0
        assert L.mtl_absmax_f32(st(), dB[t].data_ptr(), Bm[t].numel(), ab[t].data_ptr()) == 0
    outs = []
    for _ in range(2):
        C = torch.full((tasks, M, N), float('nan')).cuda()
        assert L.mtl_gemm_h2_tn_tb(st(), M, N, K, dA.data_ptr(), M, aa.data_ptr(), S, dB.data_ptr(), N, ab.data_ptr(), S, C.data_ptr(), N, tasks,
                                   K * M, K * N, M * N) == 0
        outs.append(C.cpu())
    for t in range(tasks):
        assert rel(outs[0][t].double(), want[t]) < 1e-6, (t, rel(outs[0][t].double(), want[t]))
    assert torch.equal(outs[0], outs[1])
