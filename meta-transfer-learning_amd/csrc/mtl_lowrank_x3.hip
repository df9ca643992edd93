// Fused rank-r projection pair on the bf16-split ("x3") matrix instructions of gfx950 (MI355X):
//     T = X . op(W1)   (M x R, R <= 128: the rank-100 bottleneck),     Y (+)= sum_z T_z . op(W2_z) (+ bias)   (M x N, N % 512 == 0)
// in ONE launch, T staying in LDS between the two products (and leaving for HBM once, as the side output the weight gradients need).
//
// Replaces the two nn.Linear of every low-rank projection of FactorizedMultiHeadAttention (modules/common_layers.py:276-306:
// query / key / value `_linear_a` -> `_linear_b`, `output_linear_a` -> `output_linear_b`) and the two data-gradient products of their
// autograd backward (d -> d . W_b -> . W_a, summed over the projections that share an input).  As two launches on the tile engine of
// mtl_gemm_x3.hip these are its worst shapes (PMC / per-shape timings of round 4: 7.6 of the 55 ms of an 8-task step at 30-40 TF):
// N = 100 gives the first product ONE 128-wide column of tiles (56 workgroups on 256 CUs at 8 tasks), K = 100 gives the second one
// four K steps between a prologue and an epilogue of the same length, and the M x 100 intermediate makes a round trip through HBM.
//
// Workgroup = 8 waves (2 x 4), 64 rows of X -> 64 rows of Y.
//   stage 1: the K loop of the tile engine (two LDS stages of [operand][piece][row][32 k] bf16 with the 16-byte chunk swizzle, loads of
//            tile k + 2 in flight, split of tile k + 1 beside the MFMAs of tile k); a wave owns a 32 x 32 block of the 64 x 128 (R padded)
//            result.
//   hand-over: accumulators -> fp32 scratch (C layout: conflict-free dword stores) -> per thread two 8-value row chunks: exact
//            3-way bf16 split into the A-operand image of stage 2 ([k16 step][piece][row][16 k]) and, optionally, the fp32 side output.
//   stage 2: a wave owns 32 rows x 128 columns (four accumulators) of the 64 x 512 output; op(W2) streams through two 48 KB buffers
//            one 16-deep K step at a time (ceil(R / 16) steps), split on the way like every other operand.
// Every fp32 operand element is split EXACTLY into three bf16 pieces (no scale, no range caveat), a block product is six
// v_mfma_f32_32x32x16_bf16 (a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0, smallest first), accumulation in fp32 -- the arithmetic of
// mtl_gemm_x3.hip; T itself is rounded to fp32 between the products exactly as the two-launch form rounds it.  Fixed-order: bitwise
// reproducible.
#include <cstdlib>
#include <type_traits>

#include "mtl_common.h"
#include "../../include/mtl_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int LR_NT = 512, LR_BM = 64, LR_RP = 128, LR_BK = 32, LR_BN = 512;
constexpr int LR_PA = LR_BM * 64, LR_PB = LR_RP * 64;            // one bf16 piece of a stage-1 operand tile (64-byte rows)
constexpr int LR_STAGE = 3 * (LR_PA + LR_PB);                   // 36864
constexpr int LR_SCR_LD = LR_RP + 4;                            // fp32 scratch row (floats)
constexpr int LR_PB2 = LR_BN * 32;                              // one piece of a stage-2 op(W2) step: 512 rows x 16 k
constexpr int LR_BUF2 = 3 * LR_PB2;                             // 49152
constexpr int LR_REGION_A = 2 * LR_BUF2;                        // 98304: stage-1 stages / scratch / stage-2 buffers
constexpr int LR_TSTEP = 3 * LR_BM * 32;                        // T image of one k16 step: [piece][row][16 k]
constexpr int LR_SMEM = LR_REGION_A + (LR_RP / 16) * LR_TSTEP;  // + 49152

struct LrP {
    const float *X, *W1, *W2;
    float *T, *Y;
    const float* bias;
    int M, Kin, R, N, ldx, ldw1, ldw2, ldt, ldy, flags;
    int H, Zt, kb, total;
    long sXb, sXh, sXt, sXk;
    long sW1b, sW1h, sW1t, sW1k;
    long sW2b, sW2h, sW2t, sW2k;
    long sTb, sTh, sTt, sTk;
    long sYb, sYh, sYt;
    long sBb, sBh, sBt;
};

__device__ __forceinline__ unsigned pack_hi(float x0, float x1) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
// x = h + m + l EXACTLY (truncating split, see mtl_gemm_x3.hip)
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x0) & 0xffff0000u);
    const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x1) & 0xffff0000u);
    const float r0 = x0 - h0, r1 = x1 - h1;
    const float m0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r0) & 0xffff0000u);
    const float m1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    const float q0 = r0 - m0, q1 = r1 - m1;
    h = pack_hi(x0, x1);
    m = pack_hi(r0, r1);
    l = pack_hi(q0, q1);
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ f32x16 mfma6(const uint4 (&a)[3], const uint4 (&b)[3], f32x16 cc) {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, a[0]), a1 = __builtin_bit_cast(bf16x8, a[1]), a2 = __builtin_bit_cast(bf16x8, a[2]);
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, b[0]), b1 = __builtin_bit_cast(bf16x8, b[1]), b2 = __builtin_bit_cast(bf16x8, b[2]);
    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, cc, 0, 0, 0);
    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, cc, 0, 0, 0);
    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, cc, 0, 0, 0);
    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, cc, 0, 0, 0);
    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, cc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, cc, 0, 0, 0);
}

// TB: the weights are stored [out][in] (nn.Linear: the forward pair, contraction index contiguous); !TB: [in][out] as seen from the
// product (the backward pair multiplies by the un-transposed weights: output index contiguous, 4 x 4 blocks transposed in registers)
template <bool TB>
__global__ __launch_bounds__(LR_NT) void lowrank_pair_x3_kernel(LrP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* const Tl = sm + LR_REGION_A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, hi = lane >> 5;
    const int per = (p.total + 7) >> 3;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);       // XCD-aware order: an XCD owns a contiguous range of row tiles
    if (t >= p.total) return;
    const int ny = (p.M + LR_BM - 1) / LR_BM, nx = p.N / LR_BN;
    const int z = t / (ny * nx), rem = t - z * (ny * nx);
    const int m0 = (rem / nx) * LR_BM, n0 = (rem % nx) * LR_BN;
    const int zt = z / p.Zt, zz = z - zt * p.Zt;
    const int zb = zz / p.H, zh = zz - zb * p.H;
    const int nk1 = (p.Kin + LR_BK - 1) / LR_BK, nks = (p.R + 15) >> 4;
    const int kq = tid & 7;

    f32x16 acc2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc2[j][v] = 0.f;

#pragma unroll 1
    for (int zi = 0; zi < p.kb; ++zi) {
        const float* X = p.X + zt * p.sXt + zb * p.sXb + zh * p.sXh + zi * p.sXk;
        const float* W1 = p.W1 + zt * p.sW1t + zb * p.sW1b + zh * p.sW1h + zi * p.sW1k;
        const float* W2 = p.W2 + zt * p.sW2t + zb * p.sW2b + zh * p.sW2h + zi * p.sW2k;
        // ------------------------------------------------------------------ stage 1: T = X . op(W1)
        const float* xrow = X + (long)min(m0 + (tid >> 3), p.M - 1) * p.ldx;
        const float* w1row[2];
        if (TB) {
#pragma unroll
            for (int i = 0; i < 2; ++i) w1row[i] = W1 + (long)min((tid >> 3) + 64 * i, p.R - 1) * p.ldw1;
        } else {
            w1row[0] = W1 + min(((tid >> 3) & 31) * 4, (p.R - 4) & ~3);      // (a quad beyond the rank is clamped: its rows are never used)
            w1row[1] = nullptr;
        }
        float4 ra[2], rb[2][2];
        unsigned kma[2], kmb[2];
        auto fetch1 = [&](int kt, int s) {
            const int k = kt * LR_BK + kq * 4;
            kma[s] = (k < p.Kin ? 1u : 0u) | (k + 1 < p.Kin ? 2u : 0u) | (k + 2 < p.Kin ? 4u : 0u) | (k + 3 < p.Kin ? 8u : 0u);
            ra[s] = *reinterpret_cast<const float4*>(xrow + (kma[s] ? k : 0));
            if (TB) {
                kmb[s] = kma[s];
#pragma unroll
                for (int i = 0; i < 2; ++i) rb[s][i] = *reinterpret_cast<const float4*>(w1row[i] + (kma[s] ? k : 0));
            } else {
                kmb[s] = 0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int kk = kt * LR_BK + kq * 4 + 2 * (tid >> 8) + j;
                    kmb[s] |= (kk < p.Kin ? 1u : 0u) << j;
                    rb[s][j] = *reinterpret_cast<const float4*>(w1row[0] + (long)min(kk, p.Kin - 1) * p.ldw1);
                }
            }
        };
        auto put = [&](unsigned char* dst, int plane, float x0, float x1, float x2, float x3) {      // one k quad of a row: three 8-byte stores
            unsigned h0, m0_, l0, h1, m1_, l1;
            split3(x0, x1, h0, m0_, l0);
            split3(x2, x3, h1, m1_, l1);
            *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(dst + plane) = make_uint2(m0_, m1_);
            *reinterpret_cast<uint2*>(dst + 2 * plane) = make_uint2(l0, l1);
        };
        auto commit1 = [&](int s, unsigned char* stage) {
            {
                const int row = tid >> 3;
                const float4 x = ra[s];
                const unsigned km = kma[s];
                put(stage + row * 64 + (((kq >> 1) ^ ((row >> 2) & 3)) << 4) + (kq & 1) * 8, LR_PA, (km & 1u) ? x.x : 0.f, (km & 2u) ? x.y : 0.f,
                    (km & 4u) ? x.z : 0.f, (km & 8u) ? x.w : 0.f);
            }
            unsigned char* sb = stage + 3 * LR_PA;
            if (TB) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = (tid >> 3) + 64 * i;
                    const float4 x = rb[s][i];
                    const unsigned km = kmb[s];
                    put(sb + row * 64 + (((kq >> 1) ^ ((row >> 2) & 3)) << 4) + (kq & 1) * 8, LR_PB, (km & 1u) ? x.x : 0.f, (km & 2u) ? x.y : 0.f,
                        (km & 4u) ? x.z : 0.f, (km & 8u) ? x.w : 0.f);
                }
            } else {
                const int mq = (tid >> 3) & 31, kh = tid >> 8;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x0 = (kmb[s] & 1u) ? comp(rb[s][0], i) : 0.f, x1 = (kmb[s] & 2u) ? comp(rb[s][1], i) : 0.f;
                    unsigned h, m, l;
                    split3(x0, x1, h, m, l);
                    unsigned char* dst = sb + (mq * 4 + i) * 64 + (((kq >> 1) ^ (mq & 3)) << 4) + (kq & 1) * 8 + kh * 4;
                    *reinterpret_cast<unsigned*>(dst) = h;
                    *reinterpret_cast<unsigned*>(dst + LR_PB) = m;
                    *reinterpret_cast<unsigned*>(dst + 2 * LR_PB) = l;
                }
            }
        };
        f32x16 acc1;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc1[v] = 0.f;
        const int arow = (wm * 32 + l31) * 64, brow = 3 * LR_PA + (wn * 32 + l31) * 64;
        int csw[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) csw[st] = ((st * 2 + hi) ^ ((l31 >> 2) & 3)) << 4;
        auto compute1 = [&](const unsigned char* stage) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                uint4 a[3], b[3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    a[pc] = *reinterpret_cast<const uint4*>(stage + pc * LR_PA + arow + csw[st]);
                    b[pc] = *reinterpret_cast<const uint4*>(stage + pc * LR_PB + brow + csw[st]);
                }
                acc1 = mfma6(a, b, acc1);
            }
        };
        fetch1(0, 0);
        if (nk1 > 1) fetch1(1, 1);
        commit1(0, sm);
        lds_barrier();
        // (register sets are addressed with compile-time indices: a run-time `kt & 1` would put them into scratch memory)
        auto step1 = [&](int kt, auto cur) {
            constexpr int C = decltype(cur)::value;
            if (kt + 2 < nk1) fetch1(kt + 2, C);
            __builtin_amdgcn_sched_barrier(0);           // the loads go out first
            compute1(sm + C * LR_STAGE);
            if (kt + 1 < nk1) commit1(C ^ 1, sm + (C ^ 1) * LR_STAGE);
            lds_barrier();
        };
#pragma unroll 1
        for (int kt = 0; kt < nk1; kt += 2) {
            step1(kt, std::integral_constant<int, 0>{});
            if (kt + 1 < nk1) step1(kt + 1, std::integral_constant<int, 1>{});
        }
        // ------------------------------------------------------------------ hand-over: T -> fp32 scratch -> A image of stage 2 (+ side output)
        float* scr = reinterpret_cast<float*>(sm);       // (the stage buffers are free: the loop ended with a barrier)
#pragma unroll
        for (int v = 0; v < 16; ++v) scr[(wm * 32 + 8 * (v >> 2) + 4 * hi + (v & 3)) * LR_SCR_LD + wn * 32 + l31] = acc1[v];
        __syncthreads();
        float* Tz = p.T ? p.T + zt * p.sTt + zb * p.sTb + zh * p.sTh + zi * p.sTk : nullptr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + LR_NT * i, row = c >> 4, cc = c & 15, col = cc * 8;
            float4 x = *reinterpret_cast<const float4*>(scr + row * LR_SCR_LD + col), y = *reinterpret_cast<const float4*>(scr + row * LR_SCR_LD + col + 4);
            if (col >= p.R) x = make_float4(0.f, 0.f, 0.f, 0.f);               // (R % 4 == 0: a quad is inside or outside)
            if (col + 4 >= p.R) y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Tz && m0 + row < p.M) {
                if (col < p.R) *reinterpret_cast<float4*>(Tz + (long)(m0 + row) * p.ldt + col) = x;
                if (col + 4 < p.R) *reinterpret_cast<float4*>(Tz + (long)(m0 + row) * p.ldt + col + 4) = y;
            }
            if ((cc >> 1) < nks) {
                uint4 h, m, l;
                split3(x.x, x.y, h.x, m.x, l.x);
                split3(x.z, x.w, h.y, m.y, l.y);
                split3(y.x, y.y, h.z, m.z, l.z);
                split3(y.z, y.w, h.w, m.w, l.w);
                unsigned char* dst = Tl + (cc >> 1) * LR_TSTEP + row * 32 + (cc & 1) * 16;
                *reinterpret_cast<uint4*>(dst) = h;
                *reinterpret_cast<uint4*>(dst + LR_BM * 32) = m;
                *reinterpret_cast<uint4*>(dst + 2 * LR_BM * 32) = l;
            }
        }
        __syncthreads();                                 // the scratch is consumed (stage 2 overwrites it), the T image is visible
        // ------------------------------------------------------------------ stage 2: acc2 += T . op(W2)
        float4 r2[4];
        unsigned km2 = 0;
        const int kq2 = tid & 3;
        auto fetch2 = [&](int ks) {
            const int k = ks * 16 + kq2 * 4;
            if (TB) {            // W2 [n][k]: rows (tid >> 2) + 128 i, this thread's k quad
                km2 = (k < p.R ? 1u : 0u) | (k + 1 < p.R ? 2u : 0u) | (k + 2 < p.R ? 4u : 0u) | (k + 3 < p.R ? 8u : 0u);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    r2[i] = *reinterpret_cast<const float4*>(W2 + (long)(n0 + (tid >> 2) + 128 * i) * p.ldw2 + (km2 ? k : 0));
            } else {             // W2 [k][n]: k rows 4 kq2 .. + 3 of the step, columns 4 (tid >> 2) .. + 3
                km2 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    km2 |= (k + j < p.R ? 1u : 0u) << j;
                    r2[j] = *reinterpret_cast<const float4*>(W2 + (long)min(k + j, p.R - 1) * p.ldw2 + n0 + (tid >> 2) * 4);
                }
            }
        };
        auto commit2 = [&](unsigned char* buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x0, x1, x2, x3;
                int row;
                if (TB) {
                    row = (tid >> 2) + 128 * i;
                    x0 = r2[i].x, x1 = r2[i].y, x2 = r2[i].z, x3 = r2[i].w;
                } else {
                    row = (tid >> 2) * 4 + i;
                    x0 = comp(r2[0], i), x1 = comp(r2[1], i), x2 = comp(r2[2], i), x3 = comp(r2[3], i);
                }
                put(buf + row * 32 + kq2 * 8, LR_PB2, (km2 & 1u) ? x0 : 0.f, (km2 & 2u) ? x1 : 0.f, (km2 & 4u) ? x2 : 0.f, (km2 & 8u) ? x3 : 0.f);
            }
        };
        auto compute2 = [&](int ks, const unsigned char* buf) {
            uint4 a[3], b[4][3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                a[pc] = *reinterpret_cast<const uint4*>(Tl + ks * LR_TSTEP + pc * LR_BM * 32 + (wm * 32 + l31) * 32 + hi * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j][pc] = *reinterpret_cast<const uint4*>(buf + pc * LR_PB2 + (wn * 128 + j * 32 + l31) * 32 + hi * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[j] = mfma6(a, b[j], acc2[j]);
        };
        fetch2(0);
        commit2(sm);
        if (nks > 1) fetch2(1);
        lds_barrier();
#pragma unroll 1
        for (int ks = 0; ks < nks; ++ks) {
            compute2(ks, sm + (ks & 1) * LR_BUF2);
            if (ks + 1 < nks) {
                commit2(sm + ((ks + 1) & 1) * LR_BUF2);
                if (ks + 2 < nks) fetch2(ks + 2);
            }
            lds_barrier();
        }
    }
    // ---------------------------------------------------------------------- epilogue: Y (+)= acc2 (+ bias)
    float* Y = p.Y + zt * p.sYt + zb * p.sYb + zh * p.sYh;
    const float* bias = p.bias ? p.bias + zt * p.sBt + zb * p.sBb + zh * p.sBh : nullptr;
    const bool accum = p.flags & MTL_GEMM_ACCUM;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 128 + j * 32 + l31;
        const float bb = bias ? bias[col] : 0.f;
        float x[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) x[v] = acc2[j][v] + bb;
        if (accum) {
            float cold[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) cold[v] = Y[(long)min(m0 + wm * 32 + 8 * (v >> 2) + 4 * hi + (v & 3), p.M - 1) * p.ldy + col];
#pragma unroll
            for (int v = 0; v < 16; ++v) x[v] += cold[v];
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + wm * 32 + 8 * (v >> 2) + 4 * hi + (v & 3);
            if (row < p.M) Y[(long)row * p.ldy + col] = x[v];
        }
    }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <bool TB>
int launch_lr(LrP p, hipStream_t s) {
    static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(lowrank_pair_x3_kernel<TB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          LR_SMEM) == hipSuccess ? 0 : MTL_ELAUNCH;
    if (attr) return attr;
    dim3 grid(((p.total + 7) / 8) * 8);
    hipLaunchKernelGGL(lowrank_pair_x3_kernel<TB>, grid, dim3(LR_NT), LR_SMEM, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

}  // namespace

extern "C" {

/* see include/mtl_hip.h */
int mtl_lowrank_pair_supported(int Kin, int R, int N) { return Kin >= 32 && Kin % 4 == 0 && R >= 16 && R <= LR_RP && R % 4 == 0 && N >= LR_BN && N % LR_BN == 0; }

int mtl_lowrank_pair_f32(void* stream, int weights_out_in, int M, int Kin, int R, int N, const float* X, int ldx, const float* W1, int ldw1,
                         const float* W2, int ldw2, float* T, int ldt, float* Y, int ldy, const float* bias, int flags, int batch, int H,
                         long sXb, long sXh, long sW1b, long sW1h, long sW2b, long sW2h, long sTb, long sTh, long sYb, long sYh, long sBiasB,
                         long sBiasH, int kbatch, long sXk, long sW1k, long sW2k, long sTk, int tasks, long sXt, long sW1t, long sW2t, long sTt,
                         long sYt, long sBiasT) {
    if (M <= 0 || batch <= 0 || H <= 0 || kbatch <= 0 || tasks <= 0 || !X || !W1 || !W2 || !Y) return MTL_EINVAL;
    if (!mtl_lowrank_pair_supported(Kin, R, N) || batch % tasks != 0 || (batch / tasks) % H != 0 || (flags & ~MTL_GEMM_ACCUM)) return MTL_EINVAL;
    if (!al16(X) || !al16(W1) || !al16(W2) || (T && !al16(T)) || (ldx & 3) || (ldw1 & 3) || (ldw2 & 3) || (ldt & 3)) return MTL_EINVAL;
    if ((sXb | sXh | sXk | sXt | sW1b | sW1h | sW1k | sW1t | sW2b | sW2h | sW2k | sW2t | sTb | sTh | sTk | sTt) & 3) return MTL_EINVAL;
    LrP p{};
    p.X = X, p.W1 = W1, p.W2 = W2, p.T = T, p.Y = Y, p.bias = bias;
    p.M = M, p.Kin = Kin, p.R = R, p.N = N, p.ldx = ldx, p.ldw1 = ldw1, p.ldw2 = ldw2, p.ldt = ldt, p.ldy = ldy, p.flags = flags;
    p.H = H, p.Zt = batch / tasks, p.kb = kbatch;
    p.total = ((M + LR_BM - 1) / LR_BM) * (N / LR_BN) * batch;
    p.sXb = sXb, p.sXh = sXh, p.sXt = sXt, p.sXk = sXk;
    p.sW1b = sW1b, p.sW1h = sW1h, p.sW1t = sW1t, p.sW1k = sW1k;
    p.sW2b = sW2b, p.sW2h = sW2h, p.sW2t = sW2t, p.sW2k = sW2k;
    p.sTb = sTb, p.sTh = sTh, p.sTt = sTt, p.sTk = sTk;
    p.sYb = sYb, p.sYh = sYh, p.sYt = sYt;
    p.sBb = sBiasB, p.sBh = sBiasH, p.sBt = sBiasT;
    return weights_out_in ? launch_lr<true>(p, as_stream(stream)) : launch_lr<false>(p, as_stream(stream));
}

}  // extern "C"
