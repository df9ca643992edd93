// Fused low-rank pair  y = (x . A^T) . B^T (+ bias)  for gfx950 (MI355X), exact fp32 on v_mfma_f32_16x16x4_f32.
//
// FactorizedMultiHeadAttention projects with rank-r pairs: q = W_b(W_a x) (modules/common_layers.py:287-289) and the output
// projection (:303), r = 100.  As two GEMM launches the M x r intermediate makes a round trip through HBM and each launch pays
// its fixed cost (~10 us of a 12-16 us small-product launch are neither loads nor MFMAs).  Here one workgroup owns
// 16 or 32 rows of x: stage 1 streams x and A through LDS in 64-deep K chunks and leaves the (rows x r) intermediate in LDS
// (zero-padded to 128 columns), stage 2 multiplies it with 128-row chunks of B.  The intermediate is still written once (the
// weight gradient dW_b = dy^T t needs it), never re-read by this kernel.
//   * z = 0..n-1 pairs per launch (the Q / K / V projections of a block: parameters at a constant stride in the flat buffer);
//   * SUM mode: y (+)= sum_z (x_z . A_z^T) . B_z^T -- the backward data path dx = sum_z (dy_z . W_b,z) . W_a,z of the same three
//     projections in ONE launch, given transposed copies of the weights (mtl_transpose_batch, once per parameter version).
// Fragment conventions and LDS layouts as in mtl_attn.hip / mtl_gemm16.hip (k = 8 s + 2 g + {0, 1}; rows padded to 4 mod 8 floats).
#include "mtl_common.h"
#include "../../include/mtl_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 64, LDK = CK + 4;      // K chunk of stage 1, LDS row stride
constexpr int RP = 128;                   // r padded (intermediate columns / A rows in LDS)
constexpr int LDT = RP + 4;               // intermediate tile row stride
constexpr int NC = 128;                   // output-column chunk of stage 2
constexpr int LDB = 108;                  // B chunk row stride: >= 8 ceil(r / 8) for r <= 104, = 4 (mod 8)
constexpr int MAXZ = 3;

struct LrP {
    const float* x;
    long sx;
    int ldx;
    const float* A;
    long sA;
    const float* B;
    long sB;
    const float* bias;
    long sbias;
    float* t;
    long st;
    float* y;
    long sy;
    int ldy;
    int M, Kin, r, N, n, accum;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float4 sel4(float4 v, bool ok) { return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f); }

template <int RT, bool SUM>
__global__ __launch_bounds__(256) void lowrank_pair_kernel(LrP p) {
    constexpr int BM = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                          // stage 1: [BM][LDK]
    float* As = smem + BM * LDK;               // stage 1: [RP][LDK]
    float* Bs = smem;                          // stage 2: [NC][LDB]   (aliases the stage-1 buffers)
    constexpr int R0 = (BM * LDK + RP * LDK) > (NC * LDB) ? (BM * LDK + RP * LDK) : (NC * LDB);
    float* Ts = smem + R0;                     // [n or 1][BM][LDT]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int z0 = SUM ? 0 : blockIdx.y, nz = SUM ? p.n : 1;
    const int ctiles = (p.r + 15) >> 4;                       // 16-column tiles of the intermediate that hold data
    const int kp = (p.r + 7) >> 3;                            // MFMA pairs over the rank in stage 2
    const int r4 = p.r >> 2;

    // ------------------------------------------------------------ stage 1: T_z = x_z . A_z^T
    for (int zi = 0; zi < nz; ++zi) {
        const int z = z0 + zi;
        const float* x = p.x + z * p.sx;
        const float* A = p.A + z * p.sA;
        f32x4 acc[RT][2];
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 rx[RT], ra[8];
        unsigned okx = 0u, oka = 0u;                 // validity bits (bool arrays ended up in scratch)
        const int lr = tid >> 4, c4 = (tid & 15) * 4;
        auto fetch = [&](int k0) {
            const bool kok = k0 + c4 < p.Kin;
            okx = oka = 0u;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int row = m0 + lr + 16 * i;
                const bool ok = row < p.M && kok;
                okx |= (ok ? 1u : 0u) << i;
                rx[i] = *reinterpret_cast<const float4*>(x + (ok ? (long)row * p.ldx + k0 + c4 : 0));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = lr + 16 * i;
                const bool ok = row < p.r && kok;
                oka |= (ok ? 1u : 0u) << i;
                ra[i] = *reinterpret_cast<const float4*>(A + (ok ? (long)row * p.Kin + k0 + c4 : 0));
            }
        };
        const int nk = (p.Kin + CK - 1) / CK;
        fetch(0);
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int i = 0; i < RT; ++i) *reinterpret_cast<float4*>(Xs + (lr + 16 * i) * LDK + c4) = sel4(rx[i], (okx >> i) & 1u);
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(As + (lr + 16 * i) * LDK + c4) = sel4(ra[i], (oka >> i) & 1u);
            __syncthreads();
            if (kt + 1 < nk) fetch((kt + 1) * CK);
            const float* pa = Xs + l16 * LDK + 2 * g;
            const float* pb = As + (32 * w + l16) * LDK + 2 * g;
            const bool c1 = 2 * w + 1 < ctiles;                // the wave's second 16-column tile holds data
            if (2 * w < ctiles) {
                float2 a[2][RT], b[2][2];
#pragma unroll
                for (int i = 0; i < RT; ++i) a[0][i] = *reinterpret_cast<const float2*>(pa + 16 * i * LDK);
                b[0][0] = *reinterpret_cast<const float2*>(pb);
                b[0][1] = *reinterpret_cast<const float2*>(pb + 16 * LDK);
#pragma unroll
                for (int s = 0; s < CK / 8; ++s) {
                    const int cur = s & 1, nxt = cur ^ 1;
                    if (s + 1 < CK / 8) {
#pragma unroll
                        for (int i = 0; i < RT; ++i) a[nxt][i] = *reinterpret_cast<const float2*>(pa + 16 * i * LDK + 8 * (s + 1));
                        b[nxt][0] = *reinterpret_cast<const float2*>(pb + 8 * (s + 1));
                        b[nxt][1] = *reinterpret_cast<const float2*>(pb + 16 * LDK + 8 * (s + 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        acc[i][0] = mfma4(a[cur][i].x, b[cur][0].x, acc[i][0]);
                        if (c1) acc[i][1] = mfma4(a[cur][i].x, b[cur][1].x, acc[i][1]);
                    }
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        acc[i][0] = mfma4(a[cur][i].y, b[cur][0].y, acc[i][0]);
                        if (c1) acc[i][1] = mfma4(a[cur][i].y, b[cur][1].y, acc[i][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
        // intermediate: C layout -> LDS tile (A-layout reads in stage 2) and, once, to HBM for the weight gradient
        float* Tz = Ts + zi * BM * LDT;
        float* tg = p.t ? p.t + z * p.st : nullptr;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int col = 32 * w + 16 * cc + l16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 16 * i + 4 * g + q;
                    const float v = acc[i][cc][q];                        // exact zeros beyond column r (zero rows of A)
                    Tz[row * LDT + col] = v;
                    if (tg && col < p.r && m0 + row < p.M) tg[(long)(m0 + row) * p.r + col] = v;
                }
            }
    }
    __syncthreads();
    // zero padding of the B chunk beyond column r (the chunk rows are rewritten below, the padding stays)
    const int padw = LDB - p.r;
    for (int e = tid; e < NC * padw; e += 256) Bs[(e / padw) * LDB + p.r + e % padw] = 0.f;

    // ------------------------------------------------------------ stage 2: y_z = T_z . B_z^T (+ bias)   [SUM: y = sum_z ...]
    constexpr int NB = 13;                                     // float4 per thread per B chunk: 128 rows x (r / 4 <= 26) / 256
    float4 rb[NB];
    unsigned okb = 0u;
    auto fetchb = [&](const float* Bz, int n0) {
        okb = 0u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / r4, c = idx - row * r4;
            const bool ok = row < NC && n0 + row < p.N;
            okb |= (ok ? 1u : 0u) << i;
            rb[i] = *reinterpret_cast<const float4*>(Bz + (ok ? (long)(n0 + row) * p.r + 4 * c : 0));
        }
    };
    auto commitb = [&]() {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / r4, c = idx - row * r4;
            if (row < NC) *reinterpret_cast<float4*>(Bs + row * LDB + 4 * c) = sel4(rb[i], (okb >> i) & 1u);
        }
    };
    const int nchunks = (p.N + NC - 1) / NC;
    const int steps = nchunks * nz;
    fetchb(p.B + z0 * p.sB, 0);
    f32x4 acc2[RT][2];
    for (int st = 0; st < steps; ++st) {
        const int nc = st / nz, zi = st - nc * nz;
        if (zi == 0) {
#pragma unroll
            for (int i = 0; i < RT; ++i) acc2[i][0] = acc2[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        commitb();
        __syncthreads();
        if (st + 1 < steps) {
            const int nc1 = (st + 1) / nz, z1 = (st + 1) - nc1 * nz;
            fetchb(p.B + (z0 + z1) * p.sB, nc1 * NC);
        }
        {
            const float* pa = Ts + zi * BM * LDT + l16 * LDT + 2 * g;
            const float* pb = Bs + (32 * w + l16) * LDB + 2 * g;
            float2 a[2][RT], b[2][2];
#pragma unroll
            for (int i = 0; i < RT; ++i) a[0][i] = *reinterpret_cast<const float2*>(pa + 16 * i * LDT);
            b[0][0] = *reinterpret_cast<const float2*>(pb);
            b[0][1] = *reinterpret_cast<const float2*>(pb + 16 * LDB);
#pragma unroll
            for (int s = 0; s < LDB / 8; ++s) {                 // kp <= 13 pairs: fully unrolled (register arrays); steps >= kp only read
                const int cur = s & 1, nxt = cur ^ 1;
                if (s + 1 < LDB / 8) {
#pragma unroll
                    for (int i = 0; i < RT; ++i) a[nxt][i] = *reinterpret_cast<const float2*>(pa + 16 * i * LDT + 8 * (s + 1));
                    b[nxt][0] = *reinterpret_cast<const float2*>(pb + 8 * (s + 1));
                    b[nxt][1] = *reinterpret_cast<const float2*>(pb + 16 * LDB + 8 * (s + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s < kp) {
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        acc2[i][0] = mfma4(a[cur][i].x, b[cur][0].x, acc2[i][0]);
                        acc2[i][1] = mfma4(a[cur][i].x, b[cur][1].x, acc2[i][1]);
                    }
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        acc2[i][0] = mfma4(a[cur][i].y, b[cur][0].y, acc2[i][0]);
                        acc2[i][1] = mfma4(a[cur][i].y, b[cur][1].y, acc2[i][1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (zi == nz - 1) {
            const int z = SUM ? 0 : z0;
            float* y = p.y + z * p.sy;
            const float* bias = p.bias ? p.bias + z * p.sbias : nullptr;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int col = nc * NC + 32 * w + 16 * cc + l16;
                if (col >= p.N) continue;
                const float bb = bias ? bias[col] : 0.f;
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    float old[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = min(m0 + 16 * i + 4 * g + q, p.M - 1);
                        old[q] = p.accum ? y[(long)row * p.ldy + col] : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = m0 + 16 * i + 4 * g + q;
                        if (row < p.M) y[(long)row * p.ldy + col] = acc2[i][cc][q] + bb + old[q];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// dst_i (C x R) = src_i (R x C)^T for a table of matrices: the transposed weight copies the backward pair product reads
__global__ __launch_bounds__(256) void transpose_batch_kernel(const mtl_transpose_desc* table, int n) {
    __shared__ float tile[32][33];
    const mtl_transpose_desc d = table[blockIdx.z];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tiles_c = (d.cols + 31) / 32, tiles_r = (d.rows + 31) / 32;
    for (int tI = blockIdx.x; tI < tiles_c * tiles_r; tI += gridDim.x) {
        const int r0 = (tI / tiles_c) * 32, c0 = (tI % tiles_c) * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty + 8 * j, c = c0 + tx;
            tile[ty + 8 * j][tx] = (r < d.rows && c < d.cols) ? d.src[(long)r * d.cols + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < d.rows && c < d.cols) d.dst[(long)c * d.rows + r] = tile[tx][ty + 8 * j];
        }
        __syncthreads();
    }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int RT, bool SUM>
int launch_pair(const LrP& p, hipStream_t s) {
    constexpr int BM = 16 * RT;
    constexpr int R0 = (BM * LDK + RP * LDK) > (NC * LDB) ? (BM * LDK + RP * LDK) : (NC * LDB);
    const int bytes = (R0 + (SUM ? MAXZ : 1) * BM * LDT) * 4;
    static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(lowrank_pair_kernel<RT, SUM>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : MTL_ELAUNCH;
    if (attr) return attr;
    dim3 grid((p.M + BM - 1) / BM, SUM ? 1 : p.n);
    hipLaunchKernelGGL((lowrank_pair_kernel<RT, SUM>), grid, dim3(256), bytes, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

}  // namespace

extern "C" {

int mtl_lowrank_supported(int Kin, int r, int N) { return r >= 4 && r <= 104 && (r & 3) == 0 && (Kin & 3) == 0 && Kin > 0 && N > 0; }

int mtl_lowrank_pair(void* stream, const float* x, long sx, int ldx, const float* A, long sA, const float* B, long sB,
                     const float* bias, long sbias, float* t, long st, float* y, long sy, int ldy, int M, int Kin, int r, int N,
                     int n, int sum_over_z, int accum) {
    if (!x || !A || !B || !y || M <= 0 || n <= 0 || !mtl_lowrank_supported(Kin, r, N)) return MTL_EINVAL;
    if (sum_over_z && n > MAXZ) return MTL_EINVAL;
    if (!al16(x) || !al16(A) || !al16(B) || (ldx & 3) || ((sx | sA | sB) & 3)) return MTL_EINVAL;
    LrP p{x, sx, ldx, A, sA, B, sB, bias, sbias, t, st, y, sy, ldy, M, Kin, r, N, n, accum};
    hipStream_t s = as_stream(stream);
    const bool big = M > 1024;                         // 32-row tiles halve the weight re-reads once there are enough row tiles
    if (sum_over_z) return big ? launch_pair<2, true>(p, s) : launch_pair<1, true>(p, s);
    return big ? launch_pair<2, false>(p, s) : launch_pair<1, false>(p, s);
}

int mtl_transpose_batch(void* stream, const mtl_transpose_desc* table_dev, int n) {
    if (!table_dev || n <= 0) return MTL_EINVAL;
    const int gx = n >= 64 ? 16 : (n >= 8 ? 128 : 1024);        // workgroups per matrix (each strides over the 32 x 32 tiles)
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(gx, 1, n), dim3(256), 0, as_stream(stream), table_dev, n);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

}  // extern "C"
