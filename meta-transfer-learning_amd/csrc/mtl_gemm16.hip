// Small-product fp32 GEMM for gfx950 (MI355X) on v_mfma_f32_16x16x4_f32: 32..64-row tiles, K groups, K-batching, row sums.
//
// The transformer half of the pass is ~200 products per pass whose outputs are only a few hundred KB (rank-100 low-rank
// projections, their weight gradients, decoder-side linears).  With the fixed 64 x 64 tiles of the big engine they fill a
// quarter of the 256 CUs, every wave walks the whole K loop alone (6.8 us at K = 512) and the remedy used so far -- split-K
// into a workspace plus a second reduction launch -- doubles the launch count.  This engine picks, per product,
//   * the workgroup tile (32 x 32, 64 x 32 or 64 x 64; a wave owns 1, 2 or 4 16 x 16 MFMA tiles): 32 x 32 up to ~1000
//     workgroups (4x the workgroups of the big engine, a quarter of the K-loop latency per wave), larger tiles beyond;
//   * the number of K groups (1, 2 or 4 groups of 4 waves per workgroup, each streaming every KG-th K tile through its own
//     LDS buffers): a long-K weight-gradient product with 64 output tiles keeps 4x the loads in flight per CU and its
//     partial accumulators are combined through LDS in group order -- no workspace, no second launch.
// Two extensions remove further launches:
//   * K-batching: C = sum_z opA(A_z) . opB(B_z) inside ONE launch (dx of the three Q/K/V low-rank a-stages, which used to be
//     three serialised accumulate launches);
//   * row sums of op(A) over K as a by-product of transposed-A products (the bias gradient colsum(dy) of dW = dy^T x, which
//     used to be two more launches per linear).
// All reductions are fixed-order: bitwise reproducible.  Workgroups are numbered XCD-aware: XCD x (= workgroup id % 8, the
// observed dispatch order, used for speed only) owns a contiguous range of the (z, m, n) tile sequence, so tiles that share
// an A row-block are neighbours in one XCD's L2.
//
// Same contract as mtl_gemm_f32 (include/mtl_hip.h); replaces nn.Linear forward / backward of the small products
// (modules/common_layers.py:130,287-289,303) and the bias-gradient reductions of their autograd backward.
#include <cstdlib>
#include <type_traits>

#include "mtl_common.h"
#include "../../include/mtl_hip.h"

#ifdef MTL_G16_PROBE
// probe builds only (tools/probe/scan_g16.py, tools/probe/build_probes.sh): forced tile / K-group choice of the small-product engine
static int g_g16_tile = 0, g_g16_kg = 0;
extern "C" void mtl_probe_g16_force(int tile, int kg) { g_g16_tile = tile, g_g16_kg = kg; }
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TK = 64;
constexpr int LDT = TK + 4;      // K-major LDS tile [row][k]: 8-byte fragment reads of 16 rows x 2 lane groups hit 32 distinct bank pairs

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
    long sBiasH, sRowH;           // strides of the INNER batch index for bias / rowsum
    int Zt;                       // batch items per TASK (third, outermost batch level: z = (zt * Zt/H + zb) * H + zh)
    long sAt, sBt, sCt, sBiasT, sRowT;
    int total;       // workgroups = tiles in N x tiles in M x batch
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// One ROWS x 64-k operand tile: HBM -> registers -> LDS, kept in the orientation it has in memory so that every commit is a
// 16-byte store: KMAJ source [row][k] -> LDS [row][68], otherwise source [k][row] -> LDS [k][ROWS + 8] (the two k rows of a
// lane-group pair land 16 banks apart).  frag() returns the two operand values (k = 8 s + 2 g, + 1) of row `rb + l16` for one
// MFMA pair.  fetch() is BRANCH-FREE: unconditional loads from clamped (always valid) addresses plus a 4-bit validity mask; the
// zero-fill happens in commit(), i.e. after the MFMAs of the current tile (with the bounds checks as branches hipcc waits for
// every load right where it is issued: four serialised round trips per K tile).
template <bool KMAJ, bool VEC, int ROWS>
struct Opnd {
    static constexpr int NV = ROWS / 16;                  // float4 per thread per tile
    static constexpr int LDM = ROWS + 8;
    static constexpr int FLOATS = KMAJ ? ROWS * LDT : TK * LDM;
    static constexpr int TPR = ROWS / 4;                  // MN-major: threads per k row
    float4 v[NV];
    unsigned m[NV];
    __device__ __forceinline__ void fetch(const float* src, long ld, int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, k, dr, dk;              // (r, k) of element 0; the 4 elements step along dr / dk
            if (KMAJ) {
                r = row0 + (tid >> 4) + 16 * i, k = k0 + (tid & 15) * 4, dr = 0, dk = 1;
            } else {
                k = k0 + tid / TPR + (256 / TPR) * i, r = row0 + (tid % TPR) * 4, dr = 1, dk = 0;
            }
            const bool ok0 = r < nrows && k < K, ok1 = r + dr < nrows && k + dk < K, ok2 = r + 2 * dr < nrows && k + 2 * dk < K,
                       ok3 = r + 3 * dr < nrows && k + 3 * dk < K;
            m[i] = (ok0 ? 1u : 0u) | (ok1 ? 2u : 0u) | (ok2 ? 4u : 0u) | (ok3 ? 8u : 0u);
            const float* q = src + (ok0 ? (KMAJ ? (long)r * ld + k : (long)k * ld + r) : 0);
            if (VEC)
                v[i] = *reinterpret_cast<const float4*>(q);      // 16-byte aligned; a partially valid quad stays inside the row's ld
            else
                v[i] = make_float4(q[0], q[ok1 ? 1 : 0], q[ok2 ? 2 : 0], q[ok3 ? 3 : 0]);
        }
    }
    __device__ __forceinline__ float4 masked(int i) const {
        float4 x = v[i];
        x.x = (m[i] & 1u) ? x.x : 0.f;
        x.y = (m[i] & 2u) ? x.y : 0.f;
        x.z = (m[i] & 4u) ? x.z : 0.f;
        x.w = (m[i] & 8u) ? x.w : 0.f;
        return x;
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < NV; ++i) m[i] = 0u;
    }
    __device__ __forceinline__ void commit(float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float4 x = masked(i);
            if (KMAJ)
                *reinterpret_cast<float4*>(lds + ((tid >> 4) + 16 * i) * LDT + (tid & 15) * 4) = x;
            else
                *reinterpret_cast<float4*>(lds + (tid / TPR + (256 / TPR) * i) * LDM + (tid % TPR) * 4) = x;
        }
    }
    static __device__ __forceinline__ float2 frag(const float* lds, int rb, int s, int l16, int g) {
        if (KMAJ) return *reinterpret_cast<const float2*>(lds + (rb + l16) * LDT + 8 * s + 2 * g);
        const float* q = lds + (8 * s + 2 * g) * LDM + rb + l16;
        return make_float2(q[0], q[LDM]);
    }
};

// Workgroup tile (32 WM) x (32 WN): 4 waves as 2 x 2, a wave owns WM x WN MFMA tiles (rows 16 (2 i + wm), cols 16 (2 j + wn)).
// KG "K groups" of 4 waves each share the output tile: group kg takes the K tiles kg, kg + KG, ...
template <bool TA, bool TB, bool VEC, int KG, int WM, int WN>
__global__ __launch_bounds__(256 * KG) void gemm16_kernel(G16P p0) {
    using OA = Opnd<!TA, VEC, 32 * WM>;          // op(A) is M x K: stored [m][k] unless transposed
    using OB = Opnd<TB, VEC, 32 * WN>;           // op(B) is K x N: stored [n][k] when transposed
    constexpr int TM = 32 * WM, TN = 32 * WN;
    __shared__ __attribute__((aligned(16))) float As[KG][OA::FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[KG][OB::FLOATS];
    const int kg = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int per = (p0.total + 7) >> 3;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);       // XCD-aware order (see the header)
    if (t >= p0.total) return;
    const G16P& p = p0;
    const int nx = (p.N + TN - 1) / TN, ny = (p.M + TM - 1) / TM;
    const int z = t / (nx * ny), rem = t - z * (nx * ny);
    const int m0 = (rem / nx) * TM, n0 = (rem % nx) * TN;
    const int zt = z / p.Zt, zz = z - zt * p.Zt;
    const int zb = zz / p.H, zh = zz - zb * p.H;
    const float* A = p.A + zt * p.sAt + zb * p.sAb + zh * p.sAh;
    const float* B = p.B + zt * p.sBt + zb * p.sBb + zh * p.sBh;
    OA ra;
    OB rb;
    const int nk = (p.K + TK - 1) / TK, tiles = nk * p.kb, iters = (tiles + KG - 1) / KG;
    f32x4 acc[WM][WN][2];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[i][j][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = TA && p.rowsum && n0 == 0;
    float rs[OA::NV][4];
#pragma unroll
    for (int i = 0; i < OA::NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rs[i][j] = 0.f;
    auto fetch = [&](int it) {
        const int tile = it * KG + kg;
        if (tile < tiles) {
            const int zn = tile / nk, kt = tile - zn * nk;
            ra.fetch(A + zn * p.sAk, p.lda, m0, p.M, kt * TK, p.K, tid);
            rb.fetch(B + zn * p.sBk, p.ldb, n0, p.N, kt * TK, p.K, tid);
        } else {                                     // this group has run out of K tiles: contribute zeros
            ra.zero();
            rb.zero();
        }
    };
    fetch(0);
    for (int it = 0; it < iters; ++it) {
        ra.commit(As[kg], tid);
        rb.commit(Bs[kg], tid);
        if (do_rowsum) {
#pragma unroll
            for (int i = 0; i < OA::NV; ++i) {
                const float4 x = ra.masked(i);
                rs[i][0] += x.x;
                rs[i][1] += x.y;
                rs[i][2] += x.z;
                rs[i][3] += x.w;
            }
        }
        __syncthreads();
        if (it + 1 < iters) fetch(it + 1);           // next tile's loads fly under this tile's MFMAs
        float2 a[2][WM], b[2][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[0][i] = OA::frag(As[kg], 16 * (2 * i + wm), 0, l16, g);
#pragma unroll
        for (int j = 0; j < WN; ++j) b[0][j] = OB::frag(Bs[kg], 16 * (2 * j + wn), 0, l16, g);
#pragma unroll
        for (int s = 0; s < TK / 8; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < TK / 8) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a[nxt][i] = OA::frag(As[kg], 16 * (2 * i + wm), s + 1, l16, g);
#pragma unroll
                for (int j = 0; j < WN; ++j) b[nxt][j] = OB::frag(Bs[kg], 16 * (2 * j + wn), s + 1, l16, g);
            }
            __builtin_amdgcn_sched_barrier(0);       // next step's LDS reads are issued BEFORE this step's MFMAs
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j][0] = mfma4(a[cur][i].x, b[cur][j].x, acc[i][j][0]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j][1] = mfma4(a[cur][i].y, b[cur][j].y, acc[i][j][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    float* red = &As[0][0];                          // the tile buffers are free: the loop ended with a barrier
    constexpr int NACC = WM * WN * 4;
    float out[WM][WN][4];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[i][j][r] = acc[i][j][0][r] + acc[i][j][1][r];
    if (KG > 1) {
        // [group][register][thread]: conflict-free; group 0 adds the others in fixed order
        static_assert((KG - 1) * NACC * 256 <= KG * OA::FLOATS, "reduction scratch exceeds the A tile buffers");
        if (kg > 0) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((kg - 1) * NACC + (i * WN + j) * 4 + r) * 256 + tid] = out[i][j][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int q = 1; q < KG; ++q)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) out[i][j][r] += red[((q - 1) * NACC + (i * WN + j) * 4 + r) * 256 + tid];
        }
        __syncthreads();
    }
    if (kg == 0) {
        const long co = zt * p.sCt + zb * p.sCb + zh * p.sCh;
        float* C = p.C + co;
        const float* gate = p.gate ? p.gate + co : nullptr;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + 16 * (2 * j + wn) + l16;
            if (col >= p.N) continue;
            const float bb = p.bias ? p.bias[zt * p.sBiasT + zb * p.sBias + zh * p.sBiasH + col] : 0.f;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                // the accumulate / gate operands of the four rows are requested together (clamped rows), then consumed
                float cold[4], gt[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = min(m0 + 16 * (2 * i + wm) + 4 * g + r, p.M - 1);
                    cold[r] = (p.flags & MTL_GEMM_ACCUM) ? C[(long)row * p.ldc + col] : 0.f;
                    gt[r] = gate ? gate[(long)row * p.ldg + col] : 1.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + 16 * (2 * i + wm) + 4 * g + r;
                    if (row >= p.M) continue;
                    float x = p.alpha * out[i][j][r] + bb;
                    if (p.flags & MTL_GEMM_RELU) x = fmaxf(x, 0.f);
                    x = gt[r] > 0.f ? x : 0.f;
                    C[(long)row * p.ldc + col] = x + cold[r];
                }
            }
        }
    }
    if (do_rowsum) {
        // thread (group kg, k-lane tid / TPR, row quad tid % TPR) holds the sum over ITS k's of 4 rows: combine the
        // KG x (256 / TPR) partial rows through LDS in a fixed order
        constexpr int KL = 256 / OA::TPR, LDR = TM + 1;
        static_assert(KG * KL * LDR <= KG * OA::FLOATS, "row-sum scratch exceeds the A tile buffers");
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < OA::NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[j] += rs[i][j];
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(kg * KL + tid / OA::TPR) * LDR + (tid % OA::TPR) * 4 + j] = part[j];
        __syncthreads();
        if (kg == 0 && tid < TM && m0 + tid < p.M) {
            float tsum = 0.f;
            for (int q = 0; q < KG * KL; ++q) tsum += red[q * LDR + tid];
            p.rowsum[zt * p.sRowT + zb * p.sRow + zh * p.sRowH + m0 + tid] += tsum;
        }
    }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// ---- few-row products (incremental decoding: one new position of B hypotheses against every decoder weight, M = B rows) ----
// C[M <= 16, N] (=|+=) alpha A[M, K] . W[N, K]^T (+ bias, ReLU): a weight-streaming product -- every weight is read once and multiplied
// with all M rows.  The tile engine above pays a 32 x 32 MFMA tile and its K-group combine for it (~10 us); here a workgroup owns 8
// output columns: a wave streams 2 weight rows with 32 lanes x 16 bytes each (512 contiguous bytes per row and step: K = 512 in four
// steps, all requested up front), A sits in LDS, the 32 partial sums of a column meet by shuffles.  Exact fp32 FMAs in a fixed order.
constexpr int RW_COLS = 8, RW_KC = 1024;       // columns per workgroup (two per wave, 32 lanes each); K chunk staged in LDS (16 rows x 1024 floats = 64 KiB)
struct RowsP {
    const float *A, *W;
    float* C;
    const float* bias;
    int M, N, K, lda, ldw, ldc;
    float alpha;
    int flags;
    long sA, sW, sC, sBias;
};
template <int MB>
__global__ __launch_bounds__(256) void gemm_rows_kernel(RowsP p) {
    extern __shared__ __attribute__((aligned(16))) float rows_lds[];              // [MB][min(K, RW_KC)] (row stride kc)
    constexpr int NW = RW_KC / 128;                                                // weight quads of a lane per chunk: k = 4 j + 128 i
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, col = lane >> 5, j = lane & 31;
    const int z = blockIdx.y;
    const float* A = p.A + z * p.sA;
    const float* W = p.W + z * p.sW;
    const int n = blockIdx.x * RW_COLS + wv * 2 + col;
    const float* wrow = W + (long)min(n, p.N - 1) * p.ldw;
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += RW_KC) {
        const int kc = min(RW_KC, p.K - k0);                                      // (a multiple of 4)
        // the chunk's weights of this lane are requested FIRST: they fly while the rows are staged (a step of a decode is a chain of
        // such launches: what counts is the latency of one workgroup, not the bandwidth of the launch)
        float4 w4[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int k = 4 * j + 128 * i;
            w4[i] = k < kc ? *reinterpret_cast<const float4*>(wrow + k0 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (k0) __syncthreads();
        for (int e = tid * 4; e < MB * kc; e += 1024) {
            const int m = e / kc, k = e - m * kc;
            *reinterpret_cast<float4*>(rows_lds + e) =
                m < p.M ? *reinterpret_cast<const float4*>(A + (long)m * p.lda + k0 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int k = 4 * j + 128 * i;
            if (k < kc) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const float4 a4 = *reinterpret_cast<const float4*>(rows_lds + m * kc + k);
                    acc[m] = fmaf(a4.w, w4[i].w, fmaf(a4.z, w4[i].z, fmaf(a4.y, w4[i].y, fmaf(a4.x, w4[i].x, acc[m]))));
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) acc[m] += __shfl_xor(acc[m], o, 64);     // the 32 lanes of a column: fixed tree
    }
    if (n < p.N) {
        const float bb = p.bias ? p.bias[z * p.sBias + n] : 0.f;
        float* C = p.C + z * p.sC;
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m == j && m < p.M) {                                              // lane j of a column writes row j
                float x = p.alpha * acc[m] + bb;
                if (p.flags & MTL_GEMM_RELU) x = fmaxf(x, 0.f);
                float* c = C + (long)m * p.ldc + n;
                *c = (p.flags & MTL_GEMM_ACCUM) ? *c + x : x;
            }
    }
}

static bool rows_ok(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* gate, int H,
                    long sAb, long sBb, int kbatch, const float* rowsum, int tasks) {
    return !transA && transB && M <= 16 && N >= 1 && kbatch == 1 && !rowsum && !gate && H == 1 && tasks == 1 && (K & 3) == 0 && (lda & 3) == 0 &&
           (ldb & 3) == 0 && ((sAb | sBb) & 3) == 0 && al16(A) && al16(B);
}

static int launch_rows(hipStream_t s, const RowsP& p, int batch) {
    const int kc = p.K < RW_KC ? p.K : RW_KC;
    const dim3 grid((p.N + RW_COLS - 1) / RW_COLS, batch);
    if (p.M <= 8) {
        static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              8 * RW_KC * 4) == hipSuccess ? 0 : MTL_ELAUNCH;
        if (attr) return attr;
        hipLaunchKernelGGL(gemm_rows_kernel<8>, grid, dim3(256), 8 * kc * 4, s, p);
    } else {
        static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              16 * RW_KC * 4) == hipSuccess ? 0 : MTL_ELAUNCH;
        if (attr) return attr;
        hipLaunchKernelGGL(gemm_rows_kernel<16>, grid, dim3(256), 16 * kc * 4, s, p);
    }
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

template <bool TA, bool TB, bool VEC, int KG, int WM, int WN>
int launch_cfg(const G16P& p, int batch, hipStream_t s) {
    G16P q = p;
    q.total = ((p.N + 32 * WN - 1) / (32 * WN)) * ((p.M + 32 * WM - 1) / (32 * WM)) * batch;
    dim3 grid(((q.total + 7) / 8) * 8);
    hipLaunchKernelGGL((gemm16_kernel<TA, TB, VEC, KG, WM, WN>), grid, dim3(256 * KG), 0, s, q);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

template <bool TA, bool TB, int WM, int WN>
int launch_kg(const G16P& p, int batch, hipStream_t s, int kg) {
    if (kg == 4) return launch_cfg<TA, TB, true, 4, WM, WN>(p, batch, s);
    if (kg == 2) return launch_cfg<TA, TB, true, 2, WM, WN>(p, batch, s);
    return launch_cfg<TA, TB, true, 1, WM, WN>(p, batch, s);
}

template <bool TA, bool TB>
int launch16(const G16P& p, int batch, hipStream_t s) {
    const bool vec = al16(p.A) && al16(p.B) && (p.lda & 3) == 0 && (p.ldb & 3) == 0 &&
                     ((p.sAb | p.sAh | p.sBb | p.sBh | p.sAk | p.sBk | p.sAt | p.sBt) & 3) == 0;
    if (!vec) return launch_cfg<TA, TB, false, 1, 1, 1>(p, batch, s);     // dword loads: unaligned operands (rare)
    auto wgs = [&](int wm, int wn) { return (long)((p.M + 32 * wm - 1) / (32 * wm)) * ((p.N + 32 * wn - 1) / (32 * wn)) * batch; };
    // grow the tile while the grid still holds about one chip-full of workgroups (256 CUs)
    // measured (tools/bench_gemm16.py): 32 x 32 tiles win or tie up to ~1000 workgroups; beyond that the larger tiles' operand
    // re-use pays (vocabulary-projection dX: 64 -> 57 us with 64 x 32)
    // (round 6 scan, tools/probe/scan_g16.py: 2000 x 512 x 512 and its K-batched rank-100 sibling -- 1008 tiles of 32 x 32 -- run 13 %
    // faster on 64 x 32 tiles with two K groups; everything up to ~770 tiles stays on 32 x 32)
    int tile = wgs(1, 1) <= 768 ? 1 : (wgs(2, 1) <= 1024 ? 2 : 3);
#ifdef MTL_G16_PROBE
    if (g_g16_tile) tile = g_g16_tile;
#endif
    const long n_wg = tile == 3 ? wgs(2, 2) : (tile == 2 ? wgs(2, 1) : wgs(1, 1));
    // K groups: only while the chip is not already full of workgroups, and each group keeps >= 2 K tiles
    const long ktiles = (long)((p.K + TK - 1) / TK) * p.kb;
    int kg = 1;
    if (n_wg <= 320 && ktiles >= 8) kg = 4;
    else if (n_wg <= 640 && ktiles >= 4) kg = 2;
#ifdef MTL_G16_PROBE
    if (g_g16_kg) kg = g_g16_kg;
#endif
    if (tile == 3) return launch_kg<TA, TB, 2, 2>(p, batch, s, kg == 4 ? 2 : kg);    // 64 x 64 x 4 groups would exceed 160 KiB of LDS
    if (tile == 2) return launch_kg<TA, TB, 2, 1>(p, batch, s, kg);
    return launch_kg<TA, TB, 1, 1>(p, batch, s, kg);
}

}  // namespace

static bool route_small(int M, int N, int K, int batch, int kbatch, bool rowsum) {
    const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64) * batch;
    const long tiles32 = (long)((M + 31) / 32) * ((N + 31) / 32) * batch;
    // the big engine (64 x 64 / 128 x 128 tiles of v_mfma_f32_32x32x2_f32, split-K through the workspace) keeps every product
    // that fills the chip with its own tiles, and the few-tile / very-long-K ones (the 5120-deep input projection)
    // ... and long-K products with few tiles (the LM decoder's dX: 700 x 512 x 10000 on 88 tiles): split-K over the chip instead of
    // K groups inside 88 workgroups (176 -> 60 us)
    const bool small = tiles64 <= 320 && !(tiles32 < 48 && (long)K * kbatch >= 4096) && !(tiles64 <= 128 && (long)K * kbatch >= 4096);
    return kbatch > 1 || rowsum || small;
}

extern "C" {

/* 1: this product runs on the small-tile engine (gemm16_kernel<...>), 0: it is forwarded to mtl_gemm_f32 (gemm_kernel<...>) */
int mtl_gemm_f32_ex_route(int M, int N, int K, int batch, int kbatch, int has_rowsum) {
    // (3: the few-row kernel, for an NT call with aligned operands, no gate and one batch level -- what the decode session issues)
    if (M <= 16 && kbatch == 1 && !has_rowsum && (K & 3) == 0) return 3;
    if (mtl_gemm_x3_eligible(M, N, batch)) return 2;
    // (split-K form of the same engine: assumes an NN / NT / TN call with 16-byte aligned operands and a workspace that holds the slices)
    if (batch == 1 && kbatch == 1 && !has_rowsum && mtl_gemm_x3_splitk_slices(0, 0, M, N, K, 0, 1L << 40)) return 2;
    return route_small(M, N, K, batch, kbatch, has_rowsum != 0) ? 1 : 0;
}

int mtl_gemm_f32_tb(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                    const float* B, int ldb, float* C, int ldc, const float* bias, const float* gate, int ldg, int flags,
                    int batch, int H, long sAb, long sAh, long sBb, long sBh, long sCb, long sCh, long sBias, int kbatch, long sAk,
                    long sBk, float* rowsum, long sRowsum, float* workspace, long workspace_bytes, long sBiasH, long sRowsumH,
                    int tasks, long sAt, long sBt, long sCt, long sBiasT, long sRowsumT) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || H <= 0 || kbatch <= 0 || tasks <= 0 || !A || !B || !C) return MTL_EINVAL;
    if (rowsum && !transA) return MTL_EINVAL;
    if (batch % tasks != 0 || (batch / tasks) % H != 0) return MTL_EINVAL;
    if (rows_ok(transA, transB, M, N, K, A, lda, B, ldb, gate, H, sAb, sBb, kbatch, rowsum, tasks)) {
        const RowsP rp{A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, flags, sAb, sBb, sCb, sBias};
        return launch_rows(as_stream(stream), rp, batch);
    }
    {   // large products: the bf16-split engine (mtl_gemm_x3.hip)
        const int rc = mtl_gemm_x3_route(stream, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, gate, ldg, flags, batch, H,
                                         sAb, sAh, sBb, sBh, sCb, sCh, sBias, kbatch, sAk, sBk, rowsum, sRowsum, sBiasH, sRowsumH, tasks,
                                         sAt, sBt, sCt, sBiasT, sRowsumT);
        if (rc != 0) return rc < 0 ? rc : MTL_OK;
    }
    if (batch == 1 && kbatch == 1 && H == 1 && !rowsum && !gate) {      // few tiles x very long K: split-K on the bf16-split engine
        const int rc = mtl_gemm_x3_splitk(stream, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, flags, workspace, workspace_bytes);
        if (rc != 0) return rc < 0 ? rc : MTL_OK;
    }
    if (!route_small(M, N, K, batch, kbatch, rowsum != nullptr))
        return mtl_gemm_f32_3l(stream, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, gate, ldg, flags, batch, H, sAb,
                               sAh, sBb, sBh, sCb, sCh, sBias, sBiasH, workspace, workspace_bytes, tasks, sAt, sBt, sCt, sBiasT);
    G16P p{A, B, C, bias, gate, rowsum, M, N, K, lda, ldb, ldc, ldg, alpha, flags, H, sAb, sAh, sBb, sBh, sCb, sCh, sBias, kbatch,
           sAk, sBk, sRowsum, sBiasH, sRowsumH, batch / tasks, sAt, sBt, sCt, sBiasT, sRowsumT, 0};
    hipStream_t s = as_stream(stream);
    if (!transA && transB) return launch16<false, true>(p, batch, s);
    if (!transA && !transB) return launch16<false, false>(p, batch, s);
    if (transA && !transB) return launch16<true, false>(p, batch, s);
    return launch16<true, true>(p, batch, s);
}

int mtl_gemm_f32_ex(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                    const float* B, int ldb, float* C, int ldc, const float* bias, const float* gate, int ldg, int flags,
                    int batch, int H, long sAb, long sAh, long sBb, long sBh, long sCb, long sCh, long sBias, int kbatch, long sAk,
                    long sBk, float* rowsum, long sRowsum, float* workspace, long workspace_bytes, long sBiasH, long sRowsumH) {
    return mtl_gemm_f32_tb(stream, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, gate, ldg, flags, batch, H, sAb, sAh,
                           sBb, sBh, sCb, sCh, sBias, kbatch, sAk, sBk, rowsum, sRowsum, workspace, workspace_bytes, sBiasH, sRowsumH,
                           1, 0, 0, 0, 0, 0);
}

}  // extern "C"
