// Large "NT" products  C[M,N] = A[M,K] . B[N,K]^T (+ bias) (gated)  on two-piece fp16 operands (mtl_h2.h) for gfx950.
//
// The VGG output projection of the encoder (models/asr/transformer.py:136-140: Linear(5120 -> 512) on the flattened feature map)
// and its data gradient are the two products of the pass that are big enough to be compute-bound: 10.5 GFLOP each, 190 us on the
// exact-fp32 MFMA engine (55 TF).  Here both operands are split on the way from HBM to LDS (3 v_mfma_f32_32x32x16_f16 per 16-deep
// step instead of 8 v_mfma_f32_32x32x2_f32), like the h2 convolutions; the caller provides the max|.| bounds (MTL_AMAX_SLOTS floats
// each).  Workgroup = 4 waves (2 x 2), tile 128 x 128 x 32, each wave 64 x 64 (four 32 x 32 accumulators); LDS holds [stage][operand]
// [piece][128 rows][32 k] fp16 with the chunk swizzle of mtl_h2.h (64 KB, two stages: one barrier per K step, next tile prefetched
// into registers under the MFMAs).  Products with few output tiles split K over workgroups: partial tiles to a workspace, summed in
// a fixed order by a second kernel that also applies the epilogue (deterministic).
#include "mtl_common.h"
#include "mtl_h2.h"
#include "../../include/mtl_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int PLANE = 128 * 64;                 // one fp16 piece of one operand tile: 128 rows x 64 bytes
constexpr int STAGE = 4 * PLANE;                // A: h, l | B: h, l
constexpr int SMEM = 2 * STAGE;

struct H2P {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* gate;
    const float* amax_a;
    const float* amax_b;
    float* part;
    int M, N, K, lda, ldb, ldc, ldg;
    int tiles_n, ksplit, steps;                  // steps = K steps (of 32) per split
};

__global__ __launch_bounds__(NT) void gemm_nt_h2_kernel(H2P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    int id = blockIdx.x;
    const int split = id % p.ksplit;
    id /= p.ksplit;
    const int m0 = (id / p.tiles_n) * BM, n0 = (id % p.tiles_n) * BN;
    const int nk_all = p.K / BK;
    const int kt0 = split * p.steps, nk = min(p.steps, nk_all - kt0);
    const float sa = pow2_scale(amax_read(p.amax_a)), sb = pow2_scale(amax_read(p.amax_b));

    // loader: rows (tid >> 3) + 32 i, k = (tid & 7) * 4 .. + 3 (one 128-byte line per 8 threads)
    const int lr = tid >> 3, k4 = (tid & 7) * 4;
    const float* ga[4];
    const float* gb[4];
    int dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lr + 32 * i;
        ga[i] = p.A + (long)min(m0 + r, p.M - 1) * p.lda + (long)kt0 * BK + k4;          // rows past the edge: clamped, masked at the store
        gb[i] = p.B + (long)min(n0 + r, p.N - 1) * p.ldb + (long)kt0 * BK + k4;
        dst[i] = r * 64 + (((k4 >> 3) ^ ((r >> 2) & 3)) << 4) + (k4 & 7) * 2;
    }
    // two register sets: the tile of step s lives in set s & 1 from the start of step s - 2 (two steps of flight time) until it is
    // split into LDS stage (s & 1) during step s - 1
    float4 ra0[4], rb0[4], ra1[4], rb1[4];
    auto fetch = [&](int kt, float4 (&ra)[4], float4 (&rb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const float4*>(ga[i] + (long)kt * BK);
            rb[i] = *reinterpret_cast<const float4*>(gb[i] + (long)kt * BK);
        }
    };
    auto commit = [&](int stage, const float4 (&ra)[4], const float4 (&rb)[4]) {
        unsigned char* s = sm + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 h, l;
            split2x2(ra[i].x * sa, ra[i].y * sa, h.x, l.x);
            split2x2(ra[i].z * sa, ra[i].w * sa, h.y, l.y);
            *reinterpret_cast<uint2*>(s + dst[i]) = h;
            *reinterpret_cast<uint2*>(s + PLANE + dst[i]) = l;
            split2x2(rb[i].x * sb, rb[i].y * sb, h.x, l.x);
            split2x2(rb[i].z * sb, rb[i].w * sb, h.y, l.y);
            *reinterpret_cast<uint2*>(s + 2 * PLANE + dst[i]) = h;
            *reinterpret_cast<uint2*>(s + 3 * PLANE + dst[i]) = l;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    // fragment addresses: row (wm|wn) * 64 + 32 i + l31, chunk (2 st + hi) ^ ((row >> 2) & 3)
    const int arow = (wm * 64 + l31) * 64, brow = 2 * PLANE + (wn * 64 + l31) * 64;
    int csw[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) csw[st] = ((st * 2 + hi) ^ ((l31 >> 2) & 3)) << 4;

    auto compute = [&](int stage) {
        const unsigned char* s = sm + stage * STAGE;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            uint4 a[2][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    a[i][pc] = *reinterpret_cast<const uint4*>(s + pc * PLANE + arow + i * 32 * 64 + csw[st]);
                    b[i][pc] = *reinterpret_cast<const uint4*>(s + pc * PLANE + brow + i * 32 * 64 + csw[st]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = h2_mfma(a[i], b[j], acc[i][j]);
        }
    };
    // step kt: issue the loads of step kt + 2 (their set was emptied during step kt - 1), multiply stage kt & 1, then split the
    // tile of step kt + 1 into the other stage (last read in step kt - 1: every wave is past that barrier).  The scheduling
    // barriers keep the split -- and the wait for its loads -- BEHIND the MFMAs instead of hoisted above them.
    auto step = [&](int kt, float4 (&rac)[4], float4 (&rbc)[4], const float4 (&ran)[4], const float4 (&rbn)[4]) {
        if (kt + 2 < nk) fetch(kt + 2, rac, rbc);
        __builtin_amdgcn_sched_barrier(0);
        compute(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) commit((kt + 1) & 1, ran, rbn);
        __syncthreads();
    };
    if (nk > 0) fetch(0, ra0, rb0);
    if (nk > 1) fetch(1, ra1, rb1);
    if (nk > 0) commit(0, ra0, rb0);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, ra0, rb0, ra1, rb1);
        if (kt + 1 < nk) step(kt + 1, ra1, rb1, ra0, rb0);
    }

    const float inv = 1.f / (sa * sb);
    const bool direct = p.ksplit == 1;
    float* out = direct ? p.C : p.part + (long)split * p.M * p.N;
    const int ldo = direct ? p.ldc : p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = min(n0 + wn * 64 + j * 32 + l31, p.N - 1);
            const bool cok = n0 + wn * 64 + j * 32 + l31 < p.N;
            const float bb = (direct && p.bias) ? p.bias[col] : 0.f;
            float gt[16];                                   // the 16 gate values of this accumulator in flight together
            if (direct && p.gate) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = min(m0 + wm * 64 + i * 32 + 8 * (v >> 2) + 4 * hi + (v & 3), p.M - 1);
                    gt[v] = p.gate[(long)row * p.ldg + col];
                }
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + wm * 64 + i * 32 + 8 * (v >> 2) + 4 * hi + (v & 3);
                float x = acc[i][j][v] * inv + bb;
                if (direct && p.gate) x = gt[v] > 0.f ? x : 0.f;
                if (cok && row < p.M) out[(long)row * ldo + col] = x;
            }
        }
}

// C = epilogue(sum_s part[s]) in split order
__global__ __launch_bounds__(256) void gemm_h2_reduce_kernel(H2P p) {
    const long total4 = (long)p.M * p.N / 4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long)gridDim.x * 256) {
        const long o = e * 4;
        const int row = (int)(o / p.N), col = (int)(o - (long)row * p.N);
        float4 s = *reinterpret_cast<const float4*>(p.part + o);
        for (int k = 1; k < p.ksplit; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(p.part + (long)k * p.M * p.N + o);
            s.x += t.x;
            s.y += t.y;
            s.z += t.z;
            s.w += t.w;
        }
        if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
            s.x += b.x;
            s.y += b.y;
            s.z += b.z;
            s.w += b.w;
        }
        if (p.gate) {
            const float4 g = *reinterpret_cast<const float4*>(p.gate + (long)row * p.ldg + col);
            s.x = g.x > 0.f ? s.x : 0.f;
            s.y = g.y > 0.f ? s.y : 0.f;
            s.z = g.z > 0.f ? s.z : 0.f;
            s.w = g.w > 0.f ? s.w : 0.f;
        }
        *reinterpret_cast<float4*>(p.C + (long)row * p.ldc + col) = s;
    }
}

int plan_ksplit(int M, int N, int K) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), nk = K / BK;
    int ks = 1;
    while (tiles * ks < 256 && ks * 2 <= nk / 4 && ks < 16) ks *= 2;       // fill the 256 CUs, keep >= 4 steps per split
    return ks;
}

}  // namespace

extern "C" {

int mtl_gemm_nt_h2_supported(int M, int N, int K) { return M > 0 && N > 0 && K >= BK && K % BK == 0 && N % 4 == 0; }

long mtl_gemm_nt_h2_workspace(int M, int N, int K) {
    if (!mtl_gemm_nt_h2_supported(M, N, K)) return 0;
    const int ks = plan_ksplit(M, N, K);
    return ks > 1 ? (long)ks * M * N * 4 : 0;
}

int mtl_gemm_nt_h2(void* stream, int M, int N, int K, const float* A, int lda, const float* amax_a, const float* B, int ldb,
                   const float* amax_b, float* C, int ldc, const float* bias, const float* gate, int ldg, float* workspace,
                   long workspace_bytes) {
    if (!A || !B || !C || !amax_a || !amax_b || !mtl_gemm_nt_h2_supported(M, N, K)) return MTL_EINVAL;
    if ((lda | ldb | ldc) & 3 || (gate && (ldg & 3)) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) |
                                                          reinterpret_cast<uintptr_t>(C)) & 15))
        return MTL_EINVAL;
    H2P p{A, B, C, bias, gate, amax_a, amax_b, workspace, M, N, K, lda, ldb, ldc, ldg, 0, 0, 0};
    p.tiles_n = (N + BN - 1) / BN;
    p.ksplit = plan_ksplit(M, N, K);
    const int nk = K / BK;
    p.steps = (nk + p.ksplit - 1) / p.ksplit;
    if (p.ksplit > 1 && (!workspace || workspace_bytes < (long)p.ksplit * M * N * 4)) return MTL_EINVAL;
    static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          SMEM) == hipSuccess ? 0 : MTL_ELAUNCH;
    if (attr) return attr;
    hipStream_t s = as_stream(stream);
    const int tiles = ((M + BM - 1) / BM) * p.tiles_n;
    hipLaunchKernelGGL(gemm_nt_h2_kernel, dim3(tiles * p.ksplit), dim3(NT), SMEM, s, p);
    MTL_CHECK_LAUNCH();
    if (p.ksplit > 1) {
        hipLaunchKernelGGL(gemm_h2_reduce_kernel, dim3(grid_for((long)M * N / 4, 256, 2048)), dim3(256), 0, s, p);
        MTL_CHECK_LAUNCH();
    }
    return MTL_OK;
}

}  // extern "C"
