// fp32 GEMM on the bf16 matrix pipe of gfx950 (MI355X): every fp32 operand element is split EXACTLY into three bf16 pieces
// (8 + 8 + 8 significand bits, bf16 has fp32's exponent range: no scaling, no range caveat) on its way from HBM to LDS, and a
// 32 x 32 x 16 block product is six v_mfma_f32_32x32x16_bf16 (a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0; the dropped terms are
// < 2^-22 of |a||b|) accumulated in fp32 -- fp32-class results (measured against fp64: as accurate as the fp32 MFMA engines) at
// 16 / 6 = 2.67 x the rate of v_mfma_f32_32x32x2_f32 (roof 2517 / 6 = 419 TF against 157).
//
// This engine takes the large products of the transformer half of a (task-batched) pass -- the FFN, the rank-100 projection
// stages, the vocabulary projection, their data gradients and their weight gradients (modules/common_layers.py:130,287-289,303,
// modules/decoder.py:109 and their autograd backward) -- with the full contract of mtl_gemm_f32_tb: three batch levels
// (task, outer, inner), K-batching (C = sum_z op(A_z) op(B_z)), row sums of op(A) (bias gradients), bias / ReLU / gate /
// accumulate epilogue.  mtl_gemm_f32_tb routes a product here when its output tiles fill a good part of the chip and its
// operands are 16-byte aligned; everything else stays on the exact-fp32 engines (mtl_gemm16.hip, mtl_mfma.hip).
//
// Workgroup = 8 waves (4 x 2), tile 256 x 128 x 32, a wave owns 64 x 64 (four 32 x 32 accumulators), two waves per SIMD
// (128 x 128 tiles with 32 x 64 per wave for products whose 256-row tiles would leave CUs idle).  LDS
// holds TWO stages of [operand][piece][rows][32 k] bf16 (2 x 72 KB) with the 16-byte chunk swizzle of mtl_h2.h (fragment reads
// are conflict-free ds_read_b128).  Per K step: the global loads of tile kt + 2 are issued into the register set that tile kt left,
// the MFMAs of tile kt read stage kt & 1, and -- in the same basic block, so that the VALU work sits between the MFMAs instead of
// in front of them -- tile kt + 1 (loaded a whole step ago) is split into the other stage; one barrier per step.  Operands keep
// their HBM orientation: a K-major source (k contiguous) is split quad by quad; an MN-major one (rows contiguous: transposed A,
// non-transposed B) is loaded as 4-row blocks and transposed in registers, so both land in the same [row][k] image.
// Workgroups are numbered XCD-aware (XCD x = workgroup id % 8 owns a contiguous range of the (z, m, n) tile sequence).
// All reductions are fixed-order: bitwise reproducible.
#include <cstdlib>
#include <type_traits>

#include "mtl_common.h"
#include "mtl_h2.h"
#include "../../include/mtl_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef MTL_X3G_DBG
#define MTL_X3G_DBG 0      // ablation builds only (tools/probe): 1 no MFMAs, 2 no fragment reads either, 4 operands fetched once, 8 no split / LDS commit, 16 B operand fetched / split once
#endif
constexpr int X3G_DBG = MTL_X3G_DBG;
constexpr int BN = 128, BK = 32, NT = 512;               // BM (256 or 128) is a template parameter of the kernel
constexpr int PLANE_B = BN * 64;                         // one bf16 piece of one operand tile: rows x 64 bytes

struct X3P {
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
    long sBiasH, sRowH;
    int Zt;
    long sAt, sBt, sCt, sBiasT, sRowT;
    int total;
    // two-piece fp16 form (NP = 2): bounds of max|A|, max|B| (MTL_AMAX_SLOTS slot heads each, mtl_h2.h) and their task strides in floats
    const float *amax_a, *amax_b;
    long sAmaxA, sAmaxB;
    int Ksplit;            // split-K launches: the product's full K; item zb (outer batch index) covers k in [zb K, min((zb + 1) K, Ksplit)) -- 0: off
    int pingpong;          // the two waves of a SIMD run a K step's halves in opposite order (0: lock-step, A/B measurements with a probe build)
#ifdef MTL_X3G_PROF
    unsigned long long* prof;   // probe builds only (tools/probe/gemm_prof.py): [workgroup][wave][8] accumulated s_memtime intervals
#endif
};
#ifdef MTL_X3G_PROF
static unsigned long long* g_x3g_prof = nullptr;
extern "C" void mtl_x3g_prof_set(void* buf) { g_x3g_prof = (unsigned long long*)buf; }
#define XG_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define XG_ACC(slot, a, b) prof_acc[slot] += (b) - (a)
#else
#define XG_T(v)
#define XG_ACC(slot, a, b)
#endif

// x0, x1 -> three dwords of packed bf16 pairs, x = h + m + l EXACTLY: h and m are truncations (top 8 significand bits of x and of
// the exact residual x - h), which leaves at most 8 significant bits for l.  Truncation keeps the dependency chain at and -> sub ->
// and -> sub (a round-to-nearest split is cvt -> shift -> sub twice over) and the three packs (v_perm_b32) hang off it sideways.
__device__ __forceinline__ unsigned pack_hi(float x0, float x1) {      // {bf16 bits of x0 (low half), of x1 (high half)}, truncating
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
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

// Workgroup barrier for the K loop: LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. for the global loads of the tile
// two steps ahead that were issued at the top of the step -- their flight time would be exposed at every barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// NP pieces of a pair: 3 = exact bf16 triple (scale unused), 2 = fp16 pair of the scaled values (mtl_h2.h)
template <int NP>
__device__ __forceinline__ void pieces(float x0, float x1, float s, unsigned (&pc)[NP]) {
    if constexpr (NP == 3) split3(x0, x1, pc[0], pc[1], pc[2]);
    else split2x2(x0 * s, x1 * s, pc[0], pc[1]);
}

__device__ __forceinline__ float sel(unsigned m, unsigned bit, float x) { return (m & bit) ? x : 0.f; }
__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// One ROWS x 32-k operand tile on its way HBM -> registers -> LDS (512 threads).  The loop-invariant part of every address (the
// thread's rows, clamped into range) is set up once (init); fetch() adds the K position, is BRANCH-FREE (clamped addresses) and
// leaves NV float4 in registers:
//   KMAJ   (source [row][k]): thread (lr = tid >> 3, kq = tid & 7) holds k = 4 kq .. + 3 of rows lr + 64 i, i < ROWS / 64;
//   MN-major (source [k][row]), 256 rows: thread (mq = tid >> 3, kq) holds the 4 x 4 block rows 4 mq .. + 3, k = 4 kq .. + 3;
//   MN-major, 128 rows: thread (kh = tid >> 8, mq = (tid >> 3) & 31, kq) holds rows 4 mq .. + 3 of k = 4 kq + 2 kh, + 1.
// Rows beyond the operand's extent are NOT zeroed: row m of op(A) only reaches row m of C and row n of op(B) only column n, and
// those are never stored (nor is their row sum), so whatever the clamped address delivers is harmless.  Only k >= K must
// contribute zeros: km (a 4-bit mask per quad, K-major) / a validity bit per k row (MN-major), applied by commit<FULL = false>.
template <bool KMAJ, int ROWS, int NP>
struct Opnd {
    static constexpr int NV = ROWS / 64;          // float4 per thread and tile (4 or 2)
    static constexpr int PLANE = ROWS * 64;
    const float* base[KMAJ ? NV : 1];
    long ld;
    __device__ __forceinline__ void init(const float* src, long ld_, int row0, int nrows, int tid) {
        ld = ld_;
        if (KMAJ) {
#pragma unroll
            for (int i = 0; i < NV; ++i) base[i] = src + (long)min(row0 + (tid >> 3) + 64 * i, nrows - 1) * ld_;
        } else {
            // a quad that straddles the last row stays inside the row's ld (16-byte aligned rows, ld % 4 == 0); a quad beyond is clamped
            const int r = row0 + (ROWS == 256 ? (tid >> 3) : ((tid >> 3) & 31)) * 4;
            base[0] = src + (r < nrows ? r : 0);
        }
    }
    struct Regs {
        float4 v[NV];
        unsigned km;            // K-major: bit e = element e of the quads is inside K; MN-major: bit j = k row j is
    };
    // zoff: element offset of the K-batch item; k0: first k of the tile
    __device__ __forceinline__ void fetch(Regs& r, long zoff, int k0, int K, int tid) const {
        const int kq = tid & 7;
        if (KMAJ) {
            const int k = k0 + kq * 4;
            r.km = (k < K ? 1u : 0u) | (k + 1 < K ? 2u : 0u) | (k + 2 < K ? 4u : 0u) | (k + 3 < K ? 8u : 0u);
            const long off = zoff + (r.km ? k : 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) r.v[i] = *reinterpret_cast<const float4*>(base[i] + off);
        } else {
            r.km = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int k = k0 + kq * 4 + (ROWS == 256 ? j : 2 * (tid >> 8) + j);
                r.km |= (k < K ? 1u : 0u) << j;
                r.v[j] = *reinterpret_cast<const float4*>(base[0] + zoff + (long)min(k, K - 1) * ld);
            }
        }
    }
    // FULL: the tile lies inside K (wave-uniform), no masks.  rs: row sums of the thread's 4 rows (MN-major A only)
    template <bool FULL, bool RS>
    __device__ __forceinline__ void commit(const Regs& r, unsigned char* lds, int tid, float (&rs)[4], float sc) const {
        const int kq = tid & 7;
        if (KMAJ) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int row = (tid >> 3) + 64 * i;
                float4 x = r.v[i];
                if (!FULL) x = make_float4(sel(r.km, 1u, x.x), sel(r.km, 2u, x.y), sel(r.km, 4u, x.z), sel(r.km, 8u, x.w));
                unsigned p0[NP], p1[NP];
                pieces<NP>(x.x, x.y, sc, p0);
                pieces<NP>(x.z, x.w, sc, p1);
                unsigned char* dst = lds + row * 64 + (((kq >> 1) ^ ((row >> 2) & 3)) << 4) + (kq & 1) * 8;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(dst + q * PLANE) = make_uint2(p0[q], p1[q]);
            }
        } else if (ROWS == 256) {
            const int mq = tid >> 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 x = make_float4(comp(r.v[0], i), comp(r.v[1], i), comp(r.v[2], i), comp(r.v[3], i));
                if (!FULL) x = make_float4(sel(r.km, 1u, x.x), sel(r.km, 2u, x.y), sel(r.km, 4u, x.z), sel(r.km, 8u, x.w));
                if (RS) rs[i] += (x.x + x.y) + (x.z + x.w);
                unsigned p0[NP], p1[NP];
                pieces<NP>(x.x, x.y, sc, p0);
                pieces<NP>(x.z, x.w, sc, p1);
                unsigned char* dst = lds + (mq * 4 + i) * 64 + (((kq >> 1) ^ (mq & 3)) << 4) + (kq & 1) * 8;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(dst + q * PLANE) = make_uint2(p0[q], p1[q]);
            }
        } else {
            const int mq = (tid >> 3) & 31, kh = tid >> 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x0 = comp(r.v[0], i), x1 = comp(r.v[1], i);
                if (!FULL) {
                    x0 = sel(r.km, 1u, x0);
                    x1 = sel(r.km, 2u, x1);
                }
                if (RS) rs[i] += x0 + x1;
                unsigned p0[NP];
                pieces<NP>(x0, x1, sc, p0);
                unsigned char* dst = lds + (mq * 4 + i) * 64 + (((kq >> 1) ^ (mq & 3)) << 4) + (kq & 1) * 8 + kh * 4;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<unsigned*>(dst + q * PLANE) = p0[q];
            }
        }
    }
};

// BM = 256: a wave owns 64 x 64 (TM = 2 row blocks); BM = 128 (products with too few 256-row tiles to fill the chip): 32 x 64
// NP = 3: exact bf16 triples, six MFMAs per block product; NP = 2: fp16 pairs of power-of-two scaled operands, three MFMAs (mtl_h2.h)
template <bool TA, bool TB, bool RS, int BM, int NP>      // RS: row sums of op(A) ride along (TA only)
__global__ __launch_bounds__(NT) void gemm_x3_kernel(X3P p) {
    constexpr int PLANE_A = BM * 64, STAGE = NP * (PLANE_A + PLANE_B), WTM = BM / 4, TM = WTM / 32;
    using OA = Opnd<!TA, BM, NP>;          // op(A) is M x K: stored [m][k] unless transposed
    using OB = Opnd<TB, BN, NP>;           // op(B) is K x N: stored [n][k] when transposed
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int per = (p.total + 7) >> 3;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);       // XCD-aware order (see the header)
    if (t >= p.total) return;
    const int nx = (p.N + BN - 1) / BN, ny = (p.M + BM - 1) / BM;
    const int z = t / (nx * ny), rem = t - z * (nx * ny);
    const int m0 = (rem / nx) * BM, n0 = (rem % nx) * BN;
    const int zt = z / p.Zt, zz = z - zt * p.Zt;
    const int zb = zz / p.H, zh = zz - zb * p.H;
    const int Kz = p.Ksplit ? min(p.K, p.Ksplit - zb * p.K) : p.K;      // this item's reduction length (the last K slice may be shorter)
    float sa = 1.f, sb = 1.f;                     // NP = 2: the power-of-two scales of this task's operands (full waves read the bounds)
    if constexpr (NP == 2) {
        sa = pow2_scale(amax_read(p.amax_a + zt * p.sAmaxA));
        sb = pow2_scale(amax_read(p.amax_b + zt * p.sAmaxB));
    }
    OA la;
    OB lb;
    la.init(p.A + zt * p.sAt + zb * p.sAb + zh * p.sAh, p.lda, m0, p.M, tid);
    lb.init(p.B + zt * p.sBt + zb * p.sBb + zh * p.sBh, p.ldb, n0, p.N, tid);
    typename OA::Regs ra0, ra1;
    typename OB::Regs rb0, rb1;
    const int nk = (Kz + BK - 1) / BK, tiles = nk * p.kb;
    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    const bool do_rowsum = RS && n0 == 0;     // (every workgroup sums -- 16 adds per tile --, the first column of workgroups stores)
    const bool kfull = Kz % BK == 0;              // no ragged last K tile
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int tile, typename OA::Regs& ra, typename OB::Regs& rb) {
        if ((X3G_DBG & 4) && tile > 1) return;
        const int zn = tile / nk, kt = tile - zn * nk;
        la.fetch(ra, zn * p.sAk, kt * BK, Kz, tid);
        if (!((X3G_DBG & 16) && tile > 1)) lb.fetch(rb, zn * p.sBk, kt * BK, Kz, tid);
    };
    auto commit = [&](auto full_tag, const typename OA::Regs& ra, const typename OB::Regs& rb, unsigned char* stage) {
        constexpr bool FULL = decltype(full_tag)::value;
        if ((X3G_DBG & 8) && stage != sm) return;
        la.template commit<FULL, RS>(ra, stage, tid, rs, sa);
        if (!((X3G_DBG & 16) && stage != sm)) lb.template commit<FULL, false>(rb, stage + NP * PLANE_A, tid, rs, sb);
    };
    // fragment addresses: row (wm | wn) * 64 + 32 i + l31, chunk (2 st + hi) ^ ((row >> 2) & 3)
    const int arow = (wm * WTM + l31) * 64, brow = NP * PLANE_A + (wn * 64 + l31) * 64;
    int csw[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) csw[st] = ((st * 2 + hi) ^ ((l31 >> 2) & 3)) << 4;
    auto compute = [&](const unsigned char* stage) {
        if (X3G_DBG & 2) return;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            uint4 a[TM][NP], b[2][NP];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i][pc] = *reinterpret_cast<const uint4*>(stage + pc * PLANE_A + arow + i * 32 * 64 + csw[st]);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j][pc] = *reinterpret_cast<const uint4*>(stage + pc * PLANE_B + brow + j * 32 * 64 + csw[st]);
            }
            // the cross terms, smallest first, each over all accumulators (dependent MFMAs are 2 TM issues apart)
            if (X3G_DBG & 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j][0] += __builtin_bit_cast(float, a[i][0].x ^ b[j][NP - 1].w);
                continue;
            }
            constexpr int NTERM = NP == 3 ? 6 : 3;
            constexpr int PA[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
            for (int tm = 0; tm < NTERM; ++tm)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (NP == 3)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][PA[tm]]),
                                                                                __builtin_bit_cast(bf16x8, b[j][PB[tm]]), acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][PA[tm]]),
                                                                               __builtin_bit_cast(f16x8, b[j][PB[tm]]), acc[i][j], 0, 0, 0);
                    }
        }
    };
    // step kt: tile kt + 2 -> the register set tile kt left; multiply stage kt & 1; split tile kt + 1 into the other stage.
    // The main loop's steps are ONE basic block each (no condition between the MFMAs and the split: hipcc interleaves them); it
    // runs unmasked -- it never meets the (possibly ragged) last K tile, except with K-batching, where only K % 32 == 0 qualifies.
    // The last steps go through the general form.
    // PING-PONG (round 4): the two waves of a SIMD (wave w and w + 4) run the two halves of a step in OPPOSITE order -- waves 0-3
    // multiply stage kt and then split tile kt + 1, waves 4-7 split first and multiply second -- so that on every SIMD one wave's MFMAs
    // run beside the other wave's VALU split and LDS stores.  In lock-step (all eight waves: multiply, then split) the phases of a step
    // simply add up: compile-time ablation builds of this kernel measured barrier + loop 0.19, MFMAs 0.50, fragment reads 0.07, global
    // fetch 0.18, split + commit 0.25 = 1.19 of the 1.24 us per K step (profiles/r4/gemm_experiments.txt).  Both halves only touch what
    // the step's single barrier already separates (stage kt is read, stage kt + 1 is written), so no second barrier is needed.
    // (128-row form only: with 64 x 64 per wave the second code path does not fit the 256-register budget -- 98-173 spilled registers)
    const bool pong = BM == 128 && __builtin_amdgcn_readfirstlane(tid >> 8) != 0 && p.pingpong;
#ifdef MTL_X3G_PROF
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    auto step_main = [&](auto full_tag, int kt, typename OA::Regs& rac, typename OB::Regs& rbc, const typename OA::Regs& ran,
                         const typename OB::Regs& rbn) {
        XG_T(t0);
        fetch(kt + 2, rac, rbc);
        __builtin_amdgcn_sched_barrier(0);           // the loads go out FIRST (hipcc sinks them behind the MFMAs otherwise: a step of flight time lost)
        XG_T(t1);
        XG_ACC(0, t0, t1);                           // operand fetch issued
        if (pong) {
            commit(full_tag, ran, rbn, sm + ((kt + 1) & 1) * STAGE);
            __builtin_amdgcn_sched_barrier(0);
            XG_T(t2);
            XG_ACC(2, t1, t2);                       // split + LDS commit (includes the wait for the operands)
            compute(sm + (kt & 1) * STAGE);
            XG_T(t3);
            XG_ACC(1, t2, t3);                       // fragment reads + matrix instructions issued
        } else {
            compute(sm + (kt & 1) * STAGE);
            __builtin_amdgcn_sched_barrier(0);
            XG_T(t2);
            XG_ACC(1, t1, t2);
            commit(full_tag, ran, rbn, sm + ((kt + 1) & 1) * STAGE);
            XG_T(t3);
            XG_ACC(2, t2, t3);
        }
        XG_T(t4);
        lds_barrier();
        XG_T(t5);
        XG_ACC(3, t4, t5);                           // barrier (drain + wait for the slowest wave)
        XG_ACC(4, t0, t5);                           // whole step
    };
    auto step_tail = [&](int kt, typename OA::Regs& rac, typename OB::Regs& rbc, const typename OA::Regs& ran, const typename OB::Regs& rbn) {
        if (kt + 2 < tiles) fetch(kt + 2, rac, rbc);
        compute(sm + (kt & 1) * STAGE);
        if (kt + 1 < tiles) commit(std::false_type{}, ran, rbn, sm + ((kt + 1) & 1) * STAGE);
        lds_barrier();
    };
    fetch(0, ra0, rb0);
    if (tiles > 1) fetch(1, ra1, rb1);
    commit(std::false_type{}, ra0, rb0, sm);
    lds_barrier();
    int kt = 0;
    if (p.kb == 1 || kfull) {
#pragma unroll 1
        for (; kt + 3 < tiles; kt += 2) {
            step_main(std::true_type{}, kt, ra0, rb0, ra1, rb1);
            step_main(std::true_type{}, kt + 1, ra1, rb1, ra0, rb0);
        }
    } else {
#pragma unroll 1
        for (; kt + 3 < tiles; kt += 2) {
            step_main(std::false_type{}, kt, ra0, rb0, ra1, rb1);
            step_main(std::false_type{}, kt + 1, ra1, rb1, ra0, rb0);
        }
    }
#pragma unroll 1
    for (; kt < tiles; kt += 2) {
        step_tail(kt, ra0, rb0, ra1, rb1);
        if (kt + 1 < tiles) step_tail(kt + 1, ra1, rb1, ra0, rb0);
    }

#ifdef MTL_X3G_PROF
    if (p.prof && lane == 0) {
        for (int k_ = 0; k_ < 8; ++k_) p.prof[((long)blockIdx.x * 8 + wave) * 8 + k_] = prof_acc[k_];
    }
#endif
    // epilogue: straight-line per accumulator -- the optional operands (gate, old C) are fetched by wave-uniform branches, all 16
    // of an accumulator in flight together (clamped addresses), the stores are predicated (no load -> wait -> store chains)
    const long co = zt * p.sCt + zb * p.sCb + zh * p.sCh;
    float* C = p.C + co;
    const float* gate = p.gate ? p.gate + co : nullptr;
    const bool accum = p.flags & MTL_GEMM_ACCUM;
    const float lo = (p.flags & MTL_GEMM_RELU) ? 0.f : -__builtin_inff();
    const float alpha = NP == 2 ? p.alpha / (sa * sb) : p.alpha;      // (powers of two: exact)
    const int rbase = m0 + wm * WTM + 4 * hi;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int colr = n0 + wn * 64 + j * 32 + l31;
            const bool cok = colr < p.N;
            const int col = cok ? colr : p.N - 1;
            const float bb = p.bias ? p.bias[zt * p.sBiasT + zb * p.sBias + zh * p.sBiasH + col] : 0.f;
            float x[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) x[v] = fmaxf(alpha * acc[i][j][v] + bb, lo);
            if (gate) {
                float gt[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) gt[v] = gate[(long)min(rbase + i * 32 + 8 * (v >> 2) + (v & 3), p.M - 1) * p.ldg + col];
#pragma unroll
                for (int v = 0; v < 16; ++v) x[v] = gt[v] > 0.f ? x[v] : 0.f;
            }
            if (accum) {
                float cold[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) cold[v] = C[(long)min(rbase + i * 32 + 8 * (v >> 2) + (v & 3), p.M - 1) * p.ldc + col];
#pragma unroll
                for (int v = 0; v < 16; ++v) x[v] += cold[v];
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = rbase + i * 32 + 8 * (v >> 2) + (v & 3);
                if (cok && row < p.M) C[(long)row * p.ldc + col] = x[v];
            }
        }
    if (RS) {
        // a thread holds the sums over ITS k's of rows 4 mq .. + 3 (BM = 256: 8 k lanes per row; 128: 8 x 2): combined in a fixed order
        float* red = reinterpret_cast<float*>(sm);   // the tile buffers are free: the loop ended with a barrier
        constexpr int NPART = BM == 256 ? 8 : 16;
        const int kq = tid & 7, mq = BM == 256 ? (tid >> 3) : ((tid >> 3) & 31), part = BM == 256 ? kq : (tid >> 8) * 8 + kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) red[part * (BM + 1) + mq * 4 + i] = rs[i];
        __syncthreads();
        if (do_rowsum && tid < BM && m0 + tid < p.M) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < NPART; ++q) s += red[q * (BM + 1) + tid];
            p.rowsum[zt * p.sRowT + zb * p.sRow + zh * p.sRowH + m0 + tid] += s;
        }
    }
}

template <bool TA, bool TB, bool RS, int BM, int NP = 3>
int launch_x3(X3P p, hipStream_t s) {
    constexpr int SMEM = 2 * NP * (BM + BN) * 64;
    static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_kernel<TA, TB, RS, BM, NP>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) == hipSuccess ? 0 : MTL_ELAUNCH;
    if (attr) return attr;
    p.pingpong = 1;
#ifdef MTL_X3G_PROF
    p.prof = g_x3g_prof;
#endif
    p.total = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.total;       // (p.total arrives as the number of batch items)
    dim3 grid(((p.total + 7) / 8) * 8);
    hipLaunchKernelGGL((gemm_x3_kernel<TA, TB, RS, BM, NP>), grid, dim3(NT), SMEM, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

template <bool TA, bool TB, bool RS>
int launch_x3_bm(const X3P& p, hipStream_t s, bool big) {
    return big ? launch_x3<TA, TB, RS, 256>(p, s) : launch_x3<TA, TB, RS, 128>(p, s);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

static int g_min_tiles = -1;
static int min_tiles_now() {
    if (g_min_tiles < 0) g_min_tiles = 128;       // (mtl_gemm_x3_min_tiles() changes it: bench.py's exact-fp32 leg sets 0 = engine off)
    return g_min_tiles;
}

static long x3_tiles(int M, int N, int batch, int bm) { return (long)((M + bm - 1) / bm) * ((N + BN - 1) / BN) * batch; }

// would a 16-byte aligned, not doubly transposed product go to the bf16-split engine?  (grid counted in 128 x 128 tiles)
int mtl_gemm_x3_eligible(int M, int N, int batch) {
    const int mt = min_tiles_now();
    const long tiles = x3_tiles(M, N, batch, 128);
    return mt > 0 && tiles >= mt && tiles <= (1L << 30);
}

extern "C" int mtl_gemm_x3_min_tiles(int set) {
    const int old = min_tiles_now();
    if (set >= 0) g_min_tiles = set;
    return old;
}

// 1: the product was issued on the bf16-split engine; 0: not eligible (the caller falls back); < 0: launch error.
int mtl_gemm_x3_route(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                      int ldb, float* C, int ldc, const float* bias, const float* gate, int ldg, int flags, int batch, int H, long sAb,
                      long sAh, long sBb, long sBh, long sCb, long sCh, long sBias, int kbatch, long sAk, long sBk, float* rowsum,
                      long sRowsum, long sBiasH, long sRowsumH, int tasks, long sAt, long sBt, long sCt, long sBiasT, long sRowsumT) {
    if ((transA && transB) || !mtl_gemm_x3_eligible(M, N, batch)) return 0;
    if (!al16(A) || !al16(B) || (lda & 3) || (ldb & 3) || ((sAb | sAh | sBb | sBh | sAk | sBk | sAt | sBt) & 3)) return 0;
    X3P p{A, B, C, bias, gate, rowsum, M, N, K, lda, ldb, ldc, ldg, alpha, flags, H, sAb, sAh, sBb, sBh, sCb, sCh, sBias, kbatch,
          sAk, sBk, sRowsum, sBiasH, sRowsumH, batch / tasks, sAt, sBt, sCt, sBiasT, sRowsumT, batch, nullptr, nullptr, 0, 0};
    // 256-row tiles (less operand traffic and split work per MFMA) once they give every CU a workgroup; 128-row tiles otherwise
    constexpr int big_from = 224;
    const bool big = x3_tiles(M, N, batch, 256) >= big_from;
    hipStream_t s = as_stream(stream);
    int rc;
    if (!transA && transB) rc = launch_x3_bm<false, true, false>(p, s, big);
    else if (!transA && !transB) rc = launch_x3_bm<false, false, false>(p, s, big);
    else if (rowsum) rc = launch_x3_bm<true, false, true>(p, s, big);
    else rc = launch_x3_bm<true, false, false>(p, s, big);
    return rc == MTL_OK ? 1 : rc;
}

// Few output tiles x very long K (the LM decoder's dX: 700 x 512 x 10000, 24 tiles): the K range is split over the grid -- one
// batched launch writes a partial per K slice into the workspace, a second kernel sums the slices in fixed order (+ bias, + C when
// accumulating).  1: done, 0: not applicable (the caller takes its other engines), < 0: launch error.
__global__ __launch_bounds__(256) void x3_splitk_sum_kernel(const float* __restrict__ ws, float* __restrict__ C, const float* __restrict__ bias,
                                                            int M, int N, int ldc, int S, int accum) {
    const long n4 = N / 4, total = (long)M * n4, slice = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / n4, c = (i - m * n4) * 4;
        float4 a = *reinterpret_cast<const float4*>(ws + m * N + c);
        for (int q = 1; q < S; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(ws + q * slice + m * N + c);
            a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
        }
        float4* dst = reinterpret_cast<float4*>(C + m * ldc + c);
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + c);
            a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        }
        if (accum) {
            const float4 o = *dst;
            a.x += o.x, a.y += o.y, a.z += o.z, a.w += o.w;
        }
        *dst = a;
    }
}

// K slices of the split-K form: -> number of slices S (0: not applicable) and the slice length Kc (a multiple of 32: slice starts are
// tile- and 16-byte aligned; the last slice is shorter when S Kc > K -- X3P::Ksplit).  Enough slices to give every CU a workgroup,
// each at least `mindepth` deep, all partials inside the workspace.
static int splitk_plan(long tiles, int M, int N, int K, long ws_bytes, int* Kc_out) {
    constexpr int mindepth = 256;
    int S = (int)((256 + tiles - 1) / tiles);
    if (S > K / mindepth) S = K / mindepth;
    while (S >= 2 && (long)S * M * N * 4 > ws_bytes) --S;
    if (S < 2) return 0;
    const int Kc = ((K + S - 1) / S + 31) / 32 * 32;
    S = (K + Kc - 1) / Kc;
    if (S < 2) return 0;
    *Kc_out = Kc;
    return S;
}

// number of K slices the split-K form would use (0: not applicable)
int mtl_gemm_x3_splitk_slices(int transA, int transB, int M, int N, int K, int flags, long ws_bytes) {
    const int mt = min_tiles_now();
    const long tiles = x3_tiles(M, N, 1, 128);
    // (round 4: from 2048 deep -- the one-task vocabulary-projection data gradient, 808 x 512 x 3765 on 28 tiles: 72 -> 30 us)
    constexpr long longk = 2048;
    if (mt <= 0 || (transA && transB) || tiles >= mt || K < longk || (N & 3) || (flags & ~MTL_GEMM_ACCUM)) return 0;
    int Kc;
    return splitk_plan(tiles, M, N, K, ws_bytes, &Kc);
}

int mtl_gemm_x3_splitk(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                       int ldb, float* C, int ldc, const float* bias, int flags, float* ws, long ws_bytes) {
    if (!ws || (ldc & 3) || !al16(C) || (bias && !al16(bias)) || !al16(A) || !al16(B) || (lda & 3) || (ldb & 3)) return 0;
    if (!mtl_gemm_x3_splitk_slices(transA, transB, M, N, K, flags, ws_bytes)) return 0;
    int Kc = 0;
    const int S = splitk_plan(x3_tiles(M, N, 1, 128), M, N, K, ws_bytes, &Kc);
    const long sA = transA ? (long)Kc * lda : Kc, sB = transB ? Kc : (long)Kc * ldb;
    X3P p{A, B, ws, nullptr, nullptr, nullptr, M, N, Kc, lda, ldb, N, 0, alpha, 0, 1, sA, 0, sB, 0, (long)M * N, 0, 0, 1,
          0, 0, 0, 0, 0, S, 0, 0, 0, 0, 0, S, nullptr, nullptr, 0, 0, K};
    hipStream_t s = as_stream(stream);
    const bool big = x3_tiles(M, N, S, 256) >= 224;
    int rc;
    if (!transA && transB) rc = launch_x3_bm<false, true, false>(p, s, big);
    else if (!transA && !transB) rc = launch_x3_bm<false, false, false>(p, s, big);
    else rc = launch_x3_bm<true, false, false>(p, s, big);
    if (rc != MTL_OK) return rc;
    const long total = (long)M * (N / 4);
    hipLaunchKernelGGL(x3_splitk_sum_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, s, ws, C, bias, M, N, ldc, S,
                       (flags & MTL_GEMM_ACCUM) ? 1 : 0);
    MTL_CHECK_LAUNCH();
    return 1;
}

extern "C" int mtl_gemm_h2_tb(void* stream, int transB, int M, int N, int K, const float* A, int lda, const float* amax_a, long sAmaxA,
                              const float* B, int ldb, const float* amax_b, long sAmaxB, float* C, int ldc, const float* bias,
                              const float* gate, int ldg, int tasks, long sAt, long sBt, long sCt, long sBiasT, float* workspace,
                              long workspace_bytes) {
    if (M <= 0 || N <= 0 || K <= 0 || tasks <= 0 || !A || !B || !C || !amax_a || !amax_b) return MTL_EINVAL;
    if (!al16(A) || !al16(B) || (lda & 3) || (ldb & 3) || ((sAt | sBt) & 3)) return MTL_EINVAL;
    hipStream_t s = as_stream(stream);
    // ONE task, few output tiles, long K (the encoder's input Linear of a rank that holds a single task: 2000 x 512 x 5120 = 64 tiles
    // on 256 CUs, 115 us): K slices over the grid into the workspace + the fixed-order sum (bias there)
    const long tiles = x3_tiles(M, N, 1, 128);
    if (tasks == 1 && !gate && workspace && tiles < 128 && K >= 2048 && !(N & 3) && !(ldc & 3) && al16(C) && (!bias || al16(bias))) {
        int Kc = 0;
        const int S = splitk_plan(tiles, M, N, K, workspace_bytes, &Kc);
        if (S >= 2) {
            X3P p{A, B, workspace, nullptr, nullptr, nullptr, M, N, Kc, lda, ldb, N, 0, 1.f, 0, 1, (long)Kc, 0, transB ? (long)Kc : (long)Kc * ldb, 0,
                  (long)M * N, 0, 0, 1, 0, 0, 0, 0, 0, S, 0, 0, 0, 0, 0, S, amax_a, amax_b, 0, 0, K};
            const bool big = x3_tiles(M, N, S, 256) >= 224;
            int rc;
            if (transB) rc = big ? launch_x3<false, true, false, 256, 2>(p, s) : launch_x3<false, true, false, 128, 2>(p, s);
            else rc = big ? launch_x3<false, false, false, 256, 2>(p, s) : launch_x3<false, false, false, 128, 2>(p, s);
            if (rc != MTL_OK) return rc;
            const long total = (long)M * (N / 4);
            hipLaunchKernelGGL(x3_splitk_sum_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, s, workspace, C,
                               bias, M, N, ldc, S, 0);
            MTL_CHECK_LAUNCH();
            return MTL_OK;
        }
    }
    X3P p{A, B, C, bias, gate, nullptr, M, N, K, lda, ldb, ldc, ldg, 1.f, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1,
          0, 0, 0, 0, 0, 1, sAt, sBt, sCt, sBiasT, 0, tasks, amax_a, amax_b, sAmaxA, sAmaxB};
    const bool big = x3_tiles(M, N, tasks, 256) >= 128;
    if (transB) return big ? launch_x3<false, true, false, 256, 2>(p, s) : launch_x3<false, true, false, 128, 2>(p, s);
    return big ? launch_x3<false, false, false, 256, 2>(p, s) : launch_x3<false, false, false, 128, 2>(p, s);
}

/* see include/mtl_hip.h */
extern "C" int mtl_gemm_h2_tn_tb(void* stream, int M, int N, int K, const float* A, int lda, const float* amax_a, long sAmaxA, const float* B,
                                 int ldb, const float* amax_b, long sAmaxB, float* C, int ldc, int tasks, long sAt, long sBt, long sCt) {
    if (M <= 0 || N <= 0 || K <= 0 || tasks <= 0 || !A || !B || !C || !amax_a || !amax_b) return MTL_EINVAL;
    if (!al16(A) || !al16(B) || (lda & 3) || (ldb & 3) || ((sAt | sBt) & 3)) return MTL_EINVAL;
    X3P p{A, B, C, nullptr, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, 1.f, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1,
          0, 0, 0, 0, 0, 1, sAt, sBt, sCt, 0, 0, tasks, amax_a, amax_b, sAmaxA, sAmaxB};
    const bool big = x3_tiles(M, N, tasks, 256) >= 224;
    hipStream_t s = as_stream(stream);
    return big ? launch_x3<true, false, false, 256, 2>(p, s) : launch_x3<true, false, false, 128, 2>(p, s);
}
