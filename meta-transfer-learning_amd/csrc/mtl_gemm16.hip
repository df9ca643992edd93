// Small-tile fp32 GEMM for gfx950 (MI355X): 32 x 32 output tiles on v_mfma_f32_16x16x4_f32.
//
// The transformer half of the pass is ~200 products per pass whose outputs are only a few hundred KB (rank-100 low-rank
// projections, their weight gradients, decoder-side linears): with the 64 x 64 tiles of the big engine they fill a quarter
// of the 256 CUs, every wave walks the whole K loop alone (6.8 us at K = 512) and the remedy used so far -- split-K into a
// workspace plus a second reduction launch -- doubles the launch count.  Here a workgroup owns a 32 x 32 tile (4 waves, one
// 16 x 16 MFMA tile each, K alternating between two accumulators so consecutive MFMAs are independent): 4x the workgroups,
// a quarter of the K-loop latency per wave, no workspace and no second launch.  Two extensions remove further launches:
//   * K-batching: C = sum_z opA(A_z) . opB(B_z) inside ONE launch (dx of the three Q/K/V low-rank a-stages, which used to be
//     three serialised accumulate launches);
//   * row sums of op(A) over K as a by-product of transposed-A products (the bias gradient colsum(dy) of dW = dy^T x, which
//     used to be two more launches per linear).
// All reductions are fixed-order: bitwise reproducible.
//
// Same contract as mtl_gemm_f32 (include/mtl_hip.h); replaces nn.Linear forward / backward of the small products
// (modules/common_layers.py:130,287-289,303) and the bias-gradient reductions of their autograd backward.
#include "mtl_common.h"
#include "../../include/mtl_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 32, TN = 32, TK = 64, LDT = TK + 4;

struct G16P {
    const float *A, *B;
    float* C;
    const float* bias;
    const float* gate;
    float* rowsum;
    int M, N, K, lda, ldb, ldc, ldg;
    float alpha;
    int flags, H;
    long sAb, sAh, sBb, sBh, sCb, sCh, sBias;
    int kb;
    long sAk, sBk, sRow;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// One 32-row x 64-k operand tile: HBM -> registers -> LDS [row][k] (ld 68).  KMAJ: the source is [row][k] (k contiguous),
// otherwise [k][row] (row contiguous; transposed while committing).  Out-of-range elements read as zero.
template <bool KMAJ, bool VEC>
struct Opnd {
    float4 v[2];
    __device__ __forceinline__ void fetch(const float* src, long ld, int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r, k, dr, dk;              // (r, k) of element 0; the 4 elements step along dr / dk
            if (KMAJ) {
                r = row0 + (tid >> 4) + 16 * i, k = k0 + (tid & 15) * 4, dr = 0, dk = 1;
            } else {
                k = k0 + (tid >> 3) + 32 * i, r = row0 + (tid & 7) * 4, dr = 1, dk = 0;
            }
            const bool ok0 = r < nrows && k < K;
            const float* q = src + (KMAJ ? (long)r * ld + k : (long)k * ld + r);
            const bool ok1 = r + dr < nrows && k + dk < K, ok2 = r + 2 * dr < nrows && k + 2 * dk < K,
                       ok3 = r + 3 * dr < nrows && k + 3 * dk < K;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (VEC) {
                if (ok0) x = *reinterpret_cast<const float4*>(q);
            } else {
                if (ok0) x.x = q[0];
                if (ok1) x.y = q[1];
                if (ok2) x.z = q[2];
                if (ok3) x.w = q[3];
            }
            x.y = ok1 ? x.y : 0.f;
            x.z = ok2 ? x.z : 0.f;
            x.w = ok3 ? x.w : 0.f;
            v[i] = x;
        }
    }
    __device__ __forceinline__ void commit(float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (KMAJ) {
                *reinterpret_cast<float4*>(lds + ((tid >> 4) + 16 * i) * LDT + (tid & 15) * 4) = v[i];
            } else {
                float* d = lds + (tid & 7) * 4 * LDT + (tid >> 3) + 32 * i;
                d[0] = v[i].x;
                d[LDT] = v[i].y;
                d[2 * LDT] = v[i].z;
                d[3 * LDT] = v[i].w;
            }
        }
    }
};

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm16_kernel(G16P p) {
    __shared__ __attribute__((aligned(16))) float As[TM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[TN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int z = blockIdx.z, zb = z / p.H, zh = z - zb * p.H;
    const float* A = p.A + zb * p.sAb + zh * p.sAh;
    const float* B = p.B + zb * p.sBb + zh * p.sBh;
    Opnd<!TA, VEC> ra;           // op(A) is M x K: stored [m][k] unless transposed
    Opnd<TB, VEC> rb;            // op(B) is K x N: stored [n][k] when transposed
    const int nk = (p.K + TK - 1) / TK, iters = nk * p.kb;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = TA && p.rowsum && blockIdx.x == 0;
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    ra.fetch(A, p.lda, m0, p.M, 0, p.K, tid);
    rb.fetch(B, p.ldb, n0, p.N, 0, p.K, tid);
    for (int it = 0; it < iters; ++it) {
        ra.commit(As, tid);
        rb.commit(Bs, tid);
        if (do_rowsum) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                rs[0] += ra.v[i].x;
                rs[1] += ra.v[i].y;
                rs[2] += ra.v[i].z;
                rs[3] += ra.v[i].w;
            }
        }
        __syncthreads();
        if (it + 1 < iters) {                         // next tile's loads fly under this tile's MFMAs
            const int zn = (it + 1) / nk, kt = (it + 1) - zn * nk;
            ra.fetch(A + zn * p.sAk, p.lda, m0, p.M, kt * TK, p.K, tid);
            rb.fetch(B + zn * p.sBk, p.ldb, n0, p.N, kt * TK, p.K, tid);
        }
        const float* pa = As + (16 * wm + l16) * LDT + 2 * g;
        const float* pb = Bs + (16 * wn + l16) * LDT + 2 * g;
        float2 a[2], b[2];
        a[0] = *reinterpret_cast<const float2*>(pa);
        b[0] = *reinterpret_cast<const float2*>(pb);
#pragma unroll
        for (int s = 0; s < TK / 8; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < TK / 8) {
                a[nxt] = *reinterpret_cast<const float2*>(pa + 8 * (s + 1));
                b[nxt] = *reinterpret_cast<const float2*>(pb + 8 * (s + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = mfma4(a[cur].x, b[cur].x, acc0);
            acc1 = mfma4(a[cur].y, b[cur].y, acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    const long co = zb * p.sCb + zh * p.sCh;
    float* C = p.C + co;
    const float* gate = p.gate ? p.gate + co : nullptr;
    const int col = n0 + 16 * wn + l16;
    if (col < p.N) {
        const float bb = p.bias ? p.bias[zb * p.sBias + col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * wm + 4 * g + r;
            if (row >= p.M) continue;
            float x = p.alpha * (acc0[r] + acc1[r]) + bb;
            if (p.flags & MTL_GEMM_RELU) x = fmaxf(x, 0.f);
            if (gate) x = gate[(long)row * p.ldg + col] > 0.f ? x : 0.f;
            float* c = C + (long)row * p.ldc + col;
            if (p.flags & MTL_GEMM_ACCUM) x += *c;
            *c = x;
        }
    }
    if (do_rowsum) {
        // thread (k-lane q = tid >> 3, row group tid & 7) holds the sum over ITS k's of 4 rows: combine the 32 k-lanes through
        // LDS in a fixed order (the tile buffers are free: the loop ended with a barrier)
        float* red = As;
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(tid >> 3) * 33 + (tid & 7) * 4 + j] = rs[j];
        __syncthreads();
        if (tid < TM && m0 + tid < p.M) {
            float t = 0.f;
            for (int q = 0; q < 32; ++q) t += red[q * 33 + tid];
            p.rowsum[zb * p.sRow + m0 + tid] += t;
        }
    }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <bool TA, bool TB>
int launch16(const G16P& p, int batch, hipStream_t s) {
    const bool vec = al16(p.A) && al16(p.B) && (p.lda & 3) == 0 && (p.ldb & 3) == 0 &&
                     ((p.sAb | p.sAh | p.sBb | p.sBh | p.sAk | p.sBk) & 3) == 0;
    dim3 grid((p.N + TN - 1) / TN, (p.M + TM - 1) / TM, batch);
    if (vec)
        hipLaunchKernelGGL((gemm16_kernel<TA, TB, true>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((gemm16_kernel<TA, TB, false>), grid, dim3(256), 0, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

}  // namespace

extern "C" {

int mtl_gemm_f32_ex(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                    const float* B, int ldb, float* C, int ldc, const float* bias, const float* gate, int ldg, int flags,
                    int batch, int H, long sAb, long sAh, long sBb, long sBh, long sCb, long sCh, long sBias, int kbatch, long sAk,
                    long sBk, float* rowsum, long sRowsum, float* workspace, long workspace_bytes) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || H <= 0 || kbatch <= 0 || !A || !B || !C) return MTL_EINVAL;
    if (rowsum && !transA) return MTL_EINVAL;
    const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64) * batch;
    const long tiles32 = (long)((M + 31) / 32) * ((N + 31) / 32) * batch;
    // the big engine (64 x 64 / 128 x 128 tiles, split-K through the workspace) keeps every product that fills the chip with
    // its own tiles, and the few-tile / very-long-K ones (the 5120-deep input projection's weight gradient)
    const bool small = tiles64 < 192 && !(tiles32 < 48 && (long)K * kbatch >= 4096);
    if (kbatch == 1 && !rowsum && !small)
        return mtl_gemm_f32(stream, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, gate, ldg, flags, batch, H, sAb,
                            sAh, sBb, sBh, sCb, sCh, sBias, workspace, workspace_bytes);
    G16P p{A, B, C, bias, gate, rowsum, M, N, K, lda, ldb, ldc, ldg, alpha, flags, H, sAb, sAh, sBb, sBh, sCb, sCh, sBias, kbatch,
           sAk, sBk, sRowsum};
    hipStream_t s = as_stream(stream);
    if (!transA && transB) return launch16<false, true>(p, batch, s);
    if (!transA && !transB) return launch16<false, false>(p, batch, s);
    if (transA && !transB) return launch16<true, false>(p, batch, s);
    return launch16<true, true>(p, batch, s);
}

}  // extern "C"
