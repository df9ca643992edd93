// Fused scaled-dot-product attention for gfx950 (MI355X) on the matrix cores: fp32-class arithmetic (exact bf16 triples / fp32).
//
// Replaces ScaledDotProductAttention.forward (modules/common_layers.py:317-331: bmm -> /temperature -> masked_fill(-inf) ->
// softmax -> dropout -> bmm) and its autograd backward, including the head split / merge copies of
// FactorizedMultiHeadAttention.forward (common_layers.py:291-293,301: heads are addressed by stride) and the h-fold
// `mask.repeat` (:296: masks are derived in-kernel from klen[] / the causal flag).
//
// The (B, h, Tq, Tk) score tensor never exists in HBM (it was 16 MB per layer at T = 1000 and 400 MB per layer at T = 5000,
// written and re-read forward and backward): one workgroup owns 64 query rows of one (batch, head), streams 64-key tiles of
// K and V through LDS and keeps an online softmax (running max / sum) per row; the backward recomputes the probabilities from
// the saved log-sum-exp.  All reductions are fixed-order -> bitwise reproducible.
//
// Arithmetic.  Head size 64 (the path's): the bf16-split ("x3") form of csrc/mtl_gemm_x3.hip -- every fp32 operand element is
// split EXACTLY into three bf16 pieces (bf16 has fp32's exponent range: no scale, no bound to deliver) and a 16 x 16 x 32 block
// product is six v_mfma_f32_16x16x32_bf16 (a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0, smallest first; the dropped terms are
// < 2^-22 |a||b|), accumulated in fp32: fp32-class results at 2.7x the rate of the fp32 matrix instructions (round 4; rounds 2-3
// ran everything below on v_mfma_f32_16x16x4_f32).  K / V / Q / dO tiles are split on their way HBM -> registers -> LDS (three
// bf16 planes, 144-byte rows: conflict-free for both fragment forms), the probability tile is split in registers on its way from
// the wave-private fp32 patch to the A operand, softmax statistics stay fp32 wave-level reductions (DPP).  Head size 16 (the
// small test fixture) keeps the exact fp32 instructions: v_mfma_f32_16x16x4_f32, 32-cycle issue.
// One wave owns 16 rows; the 16 x 64 score tile is four independent accumulators, so back-to-back MFMAs never wait on the
// dependent latency.  Operand fragments (both instruction families):
//   A (rows x k):  lane l supplies A[l & 15][k(l >> 4)],   B (k x cols): lane l supplies B[k(l >> 4)][l & 15],
//   C/D: lane l, register r holds C[4 (l >> 4) + r][l & 15].
// The contraction index may be permuted freely as long as A and B use the same permutation: fp32: an MFMA pair (x, y) of step s
// uses k = 8 s + 2 (l >> 4) + {0, 1} (one 8-byte LDS read fetches both operands of the pair); bf16: step s uses k = 32 s + 8 (l >> 4)
// + {0..7} (one 16-byte read per piece, or two transposing ds_read_b64_tr_b16 when the contraction runs over the tile's ROWS).
// fp32 LDS tiles are stored [row][D + 4]: the 8-byte fragment reads of 16 rows x 2 lane groups then hit 32 distinct bank pairs.
// A probability tile leaves the MFMA in C layout and is needed as an A operand by the next product: it takes one round trip
// through a wave-private LDS patch (ds_write_b32 in C layout, reads in A layout), no workgroup barrier involved.
#include <type_traits>

#include "mtl_common.h"
#include "../../include/mtl_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AT_ROWS = 64;   // rows (queries, or keys in the key/value-gradient kernel) per workgroup: 16 per wave
constexpr int AT_TILE = 64;   // streamed tile (keys, or queries in the key/value-gradient kernel)
constexpr int AT_LDP = AT_TILE + 4;

struct AttnP {
    const float *q, *k, *v;
    int ldq, ldk, ldv;
    const int* klen;
    int causal;
    float scale;
    int B, H, Tq, Tk;
    const uint8_t* pmask;
    int ldm;
    float pscale;
    float* O;
    int ldo;
    float* lse;
    const float* Oc;      // backward: forward output
    const float* dO;
    float* delta;
    float *dq, *dk, *dv;
    int lddq, lddk, lddv;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 zero_acc() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}
// max / sum over the 16 lanes that hold one C-layout row: four DPP moves (xor 1, xor 2 inside the quad, half-row mirror, row
// mirror) -- plain VALU; the ds_bpermute form of __shfl_xor cost one LDS-crossbar round trip per step (32 serialised per tile)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));     // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));     // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));    // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));    // row_mirror
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
// LDS writes of this wave become visible to its own later LDS reads (DS ops of one wave execute in order; this only stops the
// compiler from moving the reads above the writes)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ================================================================== exact-fp32 engine (head size 16)
template <int D>
struct EngF32 {
    static constexpr int TILE_BYTES = AT_TILE * (D + 4) * 4;
    // A 64-row x D tile of a head-strided matrix: HBM -> registers (16-byte loads, rows past `nrows` read as zero) -> LDS [row][D+4]
    struct Tile {
        static constexpr int LD = D + 4;
        static constexpr int VPR = D / 4;          // float4 per row
        static constexpr int RPP = 256 / VPR;      // rows per pass of the 256 threads
        static constexpr int NV = 64 / RPP;
        float4 v[NV];
        __device__ __forceinline__ void fetch(const float* base, int ld, int row0, int nrows, int tid) {
            const int c4 = (tid % VPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int row = row0 + tid / VPR + i * RPP;
                const bool ok = row < nrows;
                const float4 x = *reinterpret_cast<const float4*>(base + (long)(ok ? row : 0) * ld + c4);
                v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __device__ __forceinline__ void commit(unsigned char* lds_, int tid) const {
            float* lds = reinterpret_cast<float*>(lds_);
            const int c4 = (tid % VPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(lds + (tid / VPR + i * RPP) * LD + c4) = v[i];
        }
    };
    struct Frag {
        float2 f[D / 8];
    };
    // this lane's D / 4 elements of a 16 x D operand row (the k positions of its fragments), fp32
    static __device__ __forceinline__ void load_raw(float (&r)[D / 4], const float* rowptr, int g) {
#pragma unroll
        for (int s = 0; s < D / 8; ++s) {
            const float2 x = *reinterpret_cast<const float2*>(rowptr + 8 * s + 2 * g);
            r[2 * s] = x.x, r[2 * s + 1] = x.y;
        }
    }
    static __device__ __forceinline__ void to_frag(Frag& fr, const float (&r)[D / 4]) {
#pragma unroll
        for (int s = 0; s < D / 8; ++s) fr.f[s] = make_float2(r[2 * s], r[2 * s + 1]);
    }
    // acc[t] (16 rows x 16 cols, t = 0..3) += frag (16 x D, registers, A layout) . tile[16 t + j][.]^T   (tile rows as columns)
    // The B fragments of step s+1 are requested before the MFMAs of step s issue (pinned with sched_barrier), so an MFMA group
    // never waits on its own LDS read.
    static __device__ __forceinline__ void mm_rows(f32x4 (&acc)[4], const Frag& fr, const unsigned char* tile_, int l16, int g) {
        const float* src = reinterpret_cast<const float*>(tile_) + l16 * (D + 4) + 2 * g;
        float2 b[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[0][t] = *reinterpret_cast<const float2*>(src + 16 * t * (D + 4));
#pragma unroll
        for (int s = 0; s < D / 8; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < D / 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t) b[nxt][t] = *reinterpret_cast<const float2*>(src + 16 * t * (D + 4) + 8 * (s + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma4(fr.f[s].x, b[cur][t].x, acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma4(fr.f[s].y, b[cur][t].y, acc[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // acc[n] (16 x 16, n = 0..D/16-1) += P (16 x 64, wave-private LDS patch, A layout reads) . tile (64 x D)
    static __device__ __forceinline__ void mm_patch(f32x4 (&acc)[D / 16], const float* patch, const unsigned char* tile_, int l16, int g) {
        constexpr int N = D / 16;
        const float* pa = patch + l16 * AT_LDP + 2 * g;
        const float* pb = reinterpret_cast<const float*>(tile_) + 2 * g * (D + 4) + l16;
        float2 a[2];
        float b0[2][N], b1[2][N];
        a[0] = *reinterpret_cast<const float2*>(pa);
#pragma unroll
        for (int n = 0; n < N; ++n) {
            b0[0][n] = pb[16 * n];
            b1[0][n] = pb[(D + 4) + 16 * n];
        }
#pragma unroll
        for (int s = 0; s < AT_TILE / 8; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < AT_TILE / 8) {
                a[nxt] = *reinterpret_cast<const float2*>(pa + 8 * (s + 1));
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    b0[nxt][n] = pb[8 * (s + 1) * (D + 4) + 16 * n];
                    b1[nxt][n] = pb[(8 * (s + 1) + 1) * (D + 4) + 16 * n];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = mfma4(a[cur].x, b0[cur][n], acc[n]);
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = mfma4(a[cur].y, b1[cur][n], acc[n]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// ================================================================== bf16-split ("x3") engine (head size 64)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// exact three-way split of two fp32 values into packed bf16 pairs: x = h + m + l, each piece the bf16 ROUNDING of what the
// previous ones left (the residuals are exact in fp32); low half = first value
__device__ __forceinline__ void split3x2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{q0, q1}, bf16x2));
}
struct X3Frag8 {       // 8 k-values of one operand row as three bf16 pieces
    uint4 p[3];
};
__device__ __forceinline__ X3Frag8 split8(const float (&x)[8]) {
    X3Frag8 f;
    split3x2(x[0], x[1], f.p[0].x, f.p[1].x, f.p[2].x);
    split3x2(x[2], x[3], f.p[0].y, f.p[1].y, f.p[2].y);
    split3x2(x[4], x[5], f.p[0].z, f.p[1].z, f.p[2].z);
    split3x2(x[6], x[7], f.p[0].w, f.p[1].w, f.p[2].w);
    return f;
}
// cc += (a0 + a1 + a2)(b0 + b1 + b2) without the terms below 2^-22: six 16 x 16 x 32 bf16 MFMAs, smallest first
__device__ __forceinline__ f32x4 x3_mfma(const X3Frag8& a, const X3Frag8& b, f32x4 cc) {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, a.p[0]), a1 = __builtin_bit_cast(bf16x8, a.p[1]), a2 = __builtin_bit_cast(bf16x8, a.p[2]);
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, b.p[0]), b1 = __builtin_bit_cast(bf16x8, b.p[1]), b2 = __builtin_bit_cast(bf16x8, b.p[2]);
    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, cc, 0, 0, 0);
    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, cc, 0, 0, 0);
    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, cc, 0, 0, 0);
    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, cc, 0, 0, 0);
    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, cc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, cc, 0, 0, 0);
}

struct EngX3 {
    static constexpr int D = 64;
    static constexpr int RS = 2 * D + 16;                 // bytes per tile row of a plane: 64 bf16 + 16 (see the header: both read forms conflict-free)
    static constexpr int PLANE = AT_TILE * RS;
    static constexpr int TILE_BYTES = 3 * PLANE;
    struct Tile {
        static constexpr int VPR = D / 4, RPP = 256 / VPR, NV = 64 / RPP;
        float4 v[NV];
        __device__ __forceinline__ void fetch(const float* base, int ld, int row0, int nrows, int tid) {
            const int c4 = (tid % VPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int row = row0 + tid / VPR + i * RPP;
                const bool ok = row < nrows;
                const float4 x = *reinterpret_cast<const float4*>(base + (long)(ok ? row : 0) * ld + c4);
                v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __device__ __forceinline__ void commit(unsigned char* lds, int tid) const {      // the split happens here: once per element and tile
            const int c4 = (tid % VPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                uint2 h, m, l;
                split3x2(v[i].x, v[i].y, h.x, m.x, l.x);
                split3x2(v[i].z, v[i].w, h.y, m.y, l.y);
                unsigned char* dst = lds + (tid / VPR + i * RPP) * RS + c4 * 2;
                *reinterpret_cast<uint2*>(dst) = h;
                *reinterpret_cast<uint2*>(dst + PLANE) = m;
                *reinterpret_cast<uint2*>(dst + 2 * PLANE) = l;
            }
        }
    };
    struct Frag {
        X3Frag8 f[D / 32];
    };
    static __device__ __forceinline__ void load_raw(float (&r)[D / 4], const float* rowptr, int g) {     // k = 32 s + 8 g + {0..7}
#pragma unroll
        for (int s = 0; s < D / 32; ++s) {
            const float4 x = *reinterpret_cast<const float4*>(rowptr + 32 * s + 8 * g), y = *reinterpret_cast<const float4*>(rowptr + 32 * s + 8 * g + 4);
            r[8 * s] = x.x, r[8 * s + 1] = x.y, r[8 * s + 2] = x.z, r[8 * s + 3] = x.w;
            r[8 * s + 4] = y.x, r[8 * s + 5] = y.y, r[8 * s + 6] = y.z, r[8 * s + 7] = y.w;
        }
    }
    static __device__ __forceinline__ void to_frag(Frag& fr, const float (&r)[D / 4]) {
#pragma unroll
        for (int s = 0; s < D / 32; ++s) {
            const float x[8] = {r[8 * s], r[8 * s + 1], r[8 * s + 2], r[8 * s + 3], r[8 * s + 4], r[8 * s + 5], r[8 * s + 6], r[8 * s + 7]};
            fr.f[s] = split8(x);
        }
    }
    static __device__ __forceinline__ X3Frag8 read_rows(const unsigned char* q) {       // one 16-byte chunk per piece
        X3Frag8 b;
        b.p[0] = *reinterpret_cast<const uint4*>(q);
        b.p[1] = *reinterpret_cast<const uint4*>(q + PLANE);
        b.p[2] = *reinterpret_cast<const uint4*>(q + 2 * PLANE);
        return b;
    }
    // acc[t] += frag . tile[16 t + j][.]^T: the contraction runs along the tile rows' own 64 values -> B fragment = one 16-byte chunk
    // of row 16 t + l16 per piece; the chunk of step (s, t) + 1 is requested before the six MFMAs of (s, t)
    static __device__ __forceinline__ void mm_rows(f32x4 (&acc)[4], const Frag& fr, const unsigned char* tile, int l16, int g) {
        const unsigned char* src = tile + l16 * RS + g * 16;
        X3Frag8 b[2];
        b[0] = read_rows(src);
#pragma unroll
        for (int i = 0; i < (D / 32) * 4; ++i) {
            const int s = i >> 2, t = i & 3;
            if (i + 1 < (D / 32) * 4) b[(i + 1) & 1] = read_rows(src + ((i + 1) & 3) * 16 * RS + ((i + 1) >> 2) * 64);
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = x3_mfma(fr.f[s], b[i & 1], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static __device__ __forceinline__ X3Frag8 read_cols(const unsigned char* q) {       // 8 tile ROWS of this lane's column: two transposing reads per piece
        X3Frag8 b;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + pc * PLANE));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + pc * PLANE + 4 * RS));
            const uint2 a = __builtin_bit_cast(uint2, lo), c = __builtin_bit_cast(uint2, hi);
            b.p[pc] = make_uint4(a.x, a.y, c.x, c.y);
        }
        return b;
    }
    // acc[n] += P (16 x 64 fp32 patch: read in A layout, split in registers) . tile (64 x D): the contraction runs over the tile's
    // ROWS -> ds_read_b64_tr_b16: lane i of a 16-lane group addresses row 8 g + (i >> 2) (+ 4), columns 16 n + 4 (i & 3) and
    // receives column 16 n + i of four rows
    static __device__ __forceinline__ void mm_patch(f32x4 (&acc)[D / 16], const float* patch, const unsigned char* tile, int l16, int g) {
        const float* pa = patch + l16 * AT_LDP + 8 * g;
        const unsigned char* src = tile + (8 * g + (l16 >> 2)) * RS + (l16 & 3) * 8;
#pragma unroll
        for (int s = 0; s < AT_TILE / 32; ++s) {
            const float4 x = *reinterpret_cast<const float4*>(pa + 32 * s), y = *reinterpret_cast<const float4*>(pa + 32 * s + 4);
            const float xs[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
            const X3Frag8 a = split8(xs);
            X3Frag8 b[2];
            b[0] = read_cols(src + 32 * s * RS);
#pragma unroll
            for (int n = 0; n < D / 16; ++n) {
                if (n + 1 < D / 16) b[(n + 1) & 1] = read_cols(src + 32 * s * RS + (n + 1) * 32);
                __builtin_amdgcn_sched_barrier(0);
                acc[n] = x3_mfma(a, b[n & 1], acc[n]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
};

template <int D>
using Eng = typename std::conditional<D == 64, EngX3, EngF32<D>>::type;

// ------------------------------------------------------------------ forward
template <int DK, int DV>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP p) {
    using EK = Eng<DK>;
    using EV = Eng<DV>;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[EK::TILE_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[EV::TILE_BYTES];
    __shared__ __attribute__((aligned(16))) float Ps[4 * 16 * AT_LDP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int q0 = blockIdx.x * AT_ROWS;
    const int klim = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    int kmax = klim;
    if (p.causal) kmax = min(kmax, min(q0 + AT_ROWS, p.Tq));
    const int nkt = (kmax + AT_TILE - 1) / AT_TILE;
    const float* kbase = p.k + (long)b * p.Tk * p.ldk + h * DK;
    const float* vbase = p.v + (long)b * p.Tk * p.ldv + h * DV;
    typename EK::Frag qf;
    {
        float raw[DK / 4];
        EK::load_raw(raw, p.q + ((long)b * p.Tq + min(q0 + 16 * w + l16, p.Tq - 1)) * p.ldq + h * DK, g);
        EK::to_frag(qf, raw);
    }
    f32x4 o[DV / 16];
#pragma unroll
    for (int n = 0; n < DV / 16; ++n) o[n] = zero_acc();
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, lsum[4] = {0.f, 0.f, 0.f, 0.f};     // running max (log2 domain), sum
    const float c2 = p.scale * 1.4426950408889634f;          // log2(e) / temperature: exp(x) = v_exp_f32(x log2 e)
    float* Pw = Ps + w * 16 * AT_LDP;
    typename EK::Tile rk;
    typename EV::Tile rv;
    rk.fetch(kbase, p.ldk, 0, p.Tk, tid);
    rv.fetch(vbase, p.ldv, 0, p.Tk, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        rk.commit(Ks, tid);
        rv.commit(Vs, tid);
        __syncthreads();
        if (kt + 1 < nkt) {                      // next tile's loads fly under this tile's MFMAs
            rk.fetch(kbase, p.ldk, (kt + 1) * AT_TILE, p.Tk, tid);
            rv.fetch(vbase, p.ldv, (kt + 1) * AT_TILE, p.Tk, tid);
        }
        unsigned keepbits = 0xffffu;                 // dropout keep bits of this lane's 4 x 4 scores, fetched under the MFMAs
        if (p.pmask) {
            keepbits = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = q0 + 16 * w + 4 * g + r;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int key = kt * AT_TILE + 16 * t + l16;
                    const bool in = key < p.Tk && qi < p.Tq;
                    const unsigned kb = in ? p.pmask[((long)bh * p.Tq + qi) * p.ldm + key] : 0u;
                    keepbits |= (kb ? 1u : 0u) << (4 * r + t);
                }
            }
        }
        f32x4 sc[4] = {zero_acc(), zero_acc(), zero_acc(), zero_acc()};
        EK::mm_rows(sc, qf, Ks, l16, g);
        // scores in the log2 domain: x2 = (q.k) * (log2 e / temperature); a tile whose 64 keys are visible to all 16 rows of this wave
        // (no length / causal boundary inside) skips the per-element masking -- most tiles; the decision is wave-uniform
        float mx[4], alpha[4], rs[4];
        const bool full = kt * AT_TILE + AT_TILE <= klim && (!p.causal || kt * AT_TILE + AT_TILE - 1 <= q0 + 16 * w);
        if (full) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int t = 0; t < 4; ++t) sc[t][r] *= c2;
                mx[r] = fmaxf(fmaxf(sc[0][r], sc[1][r]), fmaxf(sc[2][r], sc[3][r]));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = q0 + 16 * w + 4 * g + r;
                mx[r] = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int key = kt * AT_TILE + 16 * t + l16;
                    const bool valid = key < klim && (!p.causal || key <= qi);
                    const float x = valid ? sc[t][r] * c2 : -INFINITY;            // (q.k)/temperature, then masked_fill(-inf)
                    sc[t][r] = x;
                    mx[r] = fmaxf(mx[r], x);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = row16_max(mx[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mnew = fmaxf(m[r], mx[r]);
            const float msafe = mnew == -INFINITY ? 0.f : mnew;
            alpha[r] = __builtin_amdgcn_exp2f(m[r] - msafe);      // m = -inf (first tile) -> 0
            m[r] = mnew;
            rs[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float e = __builtin_amdgcn_exp2f(sc[t][r] - msafe);
                sc[t][r] = e;
                rs[r] += e;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[r] = row16_sum(rs[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            lsum[r] = lsum[r] * alpha[r] + rs[r];
#pragma unroll
            for (int n = 0; n < DV / 16; ++n) o[n][r] *= alpha[r];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float e = sc[t][r];
                // dropout on the probabilities (common_layers.py:328); the row sum stays un-dropped
                if (p.pmask) e = (keepbits >> (4 * r + t)) & 1u ? e * p.pscale : 0.f;
                Pw[(4 * g + r) * AT_LDP + 16 * t + l16] = e;
            }
        }
        wave_lds_sync();
        EV::mm_patch(o, Pw, Vs, l16, g);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + 16 * w + 4 * g + r;
        if (qi >= p.Tq) continue;
        const float inv = 1.f / lsum[r];
        float* orow = p.O + ((long)b * p.Tq + qi) * p.ldo + h * DV;
#pragma unroll
        for (int n = 0; n < DV / 16; ++n) orow[16 * n + l16] = o[n][r] * inv;
        if (l16 == 0) p.lse[(long)bh * p.Tq + qi] = m[r] * 0.6931471805599453f + logf(lsum[r]);    // natural log-sum-exp
    }
}

// ------------------------------------------------------------------ backward, query side: dQ (and delta = rowsum(dO * O))
// dS = P * (dP - delta) * scale with P = exp(S * scale - lse) recomputed, dP = (dO . V^T) [* mask * pscale];  dQ = dS . K
template <int DK, int DV>
__device__ __forceinline__ void attn_bwd_q_role(const AttnP& p, int qblk, unsigned char* Ks, unsigned char* Vs, float* Ps) {
    using EK = Eng<DK>;
    using EV = Eng<DV>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * AT_ROWS;
    const int klim = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    int kmax = klim;
    if (p.causal) kmax = min(kmax, min(q0 + AT_ROWS, p.Tq));
    const int nkt = (kmax + AT_TILE - 1) / AT_TILE;
    const float* kbase = p.k + (long)b * p.Tk * p.ldk + h * DK;
    const float* vbase = p.v + (long)b * p.Tk * p.ldv + h * DV;
    const int qrow = min(q0 + 16 * w + l16, p.Tq - 1);
    typename EK::Frag qf;
    typename EV::Frag dof;
    float dl_row;
    {
        float raw[DK / 4], rdo[DV / 4], ro[DV / 4];
        EK::load_raw(raw, p.q + ((long)b * p.Tq + qrow) * p.ldq + h * DK, g);
        EK::to_frag(qf, raw);
        EV::load_raw(rdo, p.dO + ((long)b * p.Tq + qrow) * p.ldo + h * DV, g);
        EV::load_raw(ro, p.Oc + ((long)b * p.Tq + qrow) * p.ldo + h * DV, g);
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < DV / 4; ++s) part += rdo[s] * ro[s];
        EV::to_frag(dof, rdo);
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        dl_row = part;                             // delta of row l16, in every lane group
        if (g == 0 && q0 + 16 * w + l16 < p.Tq) p.delta[(long)bh * p.Tq + q0 + 16 * w + l16] = part;
    }
    float dl[4], ls[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dl[r] = __shfl(dl_row, 4 * g + r, 64);
        ls[r] = p.lse[(long)bh * p.Tq + min(q0 + 16 * w + 4 * g + r, p.Tq - 1)] * 1.4426950408889634f;     // log2 domain
    }
    const float c2 = p.scale * 1.4426950408889634f;
    f32x4 dq[DK / 16];
#pragma unroll
    for (int n = 0; n < DK / 16; ++n) dq[n] = zero_acc();
    float* Pw = Ps + w * 16 * AT_LDP;
    typename EK::Tile rk;
    typename EV::Tile rv;
    rk.fetch(kbase, p.ldk, 0, p.Tk, tid);
    rv.fetch(vbase, p.ldv, 0, p.Tk, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        rk.commit(Ks, tid);
        rv.commit(Vs, tid);
        __syncthreads();
        if (kt + 1 < nkt) {
            rk.fetch(kbase, p.ldk, (kt + 1) * AT_TILE, p.Tk, tid);
            rv.fetch(vbase, p.ldv, (kt + 1) * AT_TILE, p.Tk, tid);
        }
        unsigned keepbits = 0xffffu;
        if (p.pmask) {
            keepbits = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = q0 + 16 * w + 4 * g + r;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int key = kt * AT_TILE + 16 * t + l16;
                    const bool in = key < p.Tk && qi < p.Tq;
                    const unsigned kb = in ? p.pmask[((long)bh * p.Tq + qi) * p.ldm + key] : 0u;
                    keepbits |= (kb ? 1u : 0u) << (4 * r + t);
                }
            }
        }
        f32x4 sc[4] = {zero_acc(), zero_acc(), zero_acc(), zero_acc()};
        f32x4 dp[4] = {zero_acc(), zero_acc(), zero_acc(), zero_acc()};
        EK::mm_rows(sc, qf, Ks, l16, g);
        EV::mm_rows(dp, dof, Vs, l16, g);
        // (wave-uniform: no length / causal boundary inside this tile for the wave's 16 rows -> no per-element masking)
        const bool full = kt * AT_TILE + AT_TILE <= klim && (!p.causal || kt * AT_TILE + AT_TILE - 1 <= q0 + 16 * w);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + 16 * w + 4 * g + r;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int key = kt * AT_TILE + 16 * t + l16;
                const bool valid = full || (key < klim && (!p.causal || key <= qi));
                const float pr = valid ? __builtin_amdgcn_exp2f(sc[t][r] * c2 - ls[r]) : 0.f;
                float d = dp[t][r];
                if (p.pmask) d = (keepbits >> (4 * r + t)) & 1u ? d * p.pscale : 0.f;
                Pw[(4 * g + r) * AT_LDP + 16 * t + l16] = pr * (d - dl[r]) * p.scale;
            }
        }
        wave_lds_sync();
        EK::mm_patch(dq, Pw, Ks, l16, g);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + 16 * w + 4 * g + r;
        if (qi >= p.Tq) continue;
        float* drow = p.dq + ((long)b * p.Tq + qi) * p.lddq + h * DK;
#pragma unroll
        for (int n = 0; n < DK / 16; ++n) drow[16 * n + l16] = dq[n][r];
    }
}

// sum over the VPR consecutive lanes that hold one tile row (16 for D = 64: a DPP row; 4 for D = 16: a quad)
template <int VPR>
__device__ __forceinline__ float tile_row_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    if (VPR == 16) {
        v += dpp_mov<0x141>(v);
        v += dpp_mov<0x140>(v);
    }
    return v;
}

// ------------------------------------------------------------------ backward, key side: dK = dS^T . Q, dV = Pd^T . dO
// one workgroup owns 64 keys (16 per wave) and streams 64-query tiles of Q and dO; everything is computed transposed
// (keys are the MFMA rows), so the per-query statistics lse / delta are per-COLUMN values here.  delta = rowsum(dO * O) of a
// query tile is computed HERE from the dO tile on its way to LDS and an O tile fetched beside it (one dot product of two float4
// and four DPP adds per thread and row group): the key side does not depend on the query side's output, so both sides are ONE
// launch (attn_bwd_kernel) whose workgroups run concurrently instead of two dependent launches of 128-256 workgroups each.
// OWN_DELTA = false: delta is read from p.delta (written by a query-side launch that ran BEFORE this one): long sequences, where
// the chip is full either way and the extra O tile only costs (T = 5000: 932 vs 1020 us).
template <int DK, int DV, bool OWN_DELTA>
__device__ __forceinline__ void attn_bwd_kv_role(const AttnP& p, int kblk, unsigned char* Qs, unsigned char* Ds, float* Ps, float* lse_s, float* dl_s) {
    using EK = Eng<DK>;
    using EV = Eng<DV>;
    using TV = typename EV::Tile;
    static_assert(TV::VPR == 16 || TV::VPR == 4, "tile_row_sum covers 16- and 4-lane rows");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int k0 = kblk * AT_ROWS;
    const int klim = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    f32x4 dk[DK / 16], dv[DV / 16];
#pragma unroll
    for (int n = 0; n < DK / 16; ++n) dk[n] = zero_acc();
#pragma unroll
    for (int n = 0; n < DV / 16; ++n) dv[n] = zero_acc();
    const int nqt = (p.Tq + AT_TILE - 1) / AT_TILE;
    const float c2 = p.scale * 1.4426950408889634f;
    const int qt0 = p.causal ? k0 / AT_TILE : 0;           // queries before the first key of this block never see it
    if (k0 < klim) {                                       // (keys at or beyond klen[b] are masked for every query: zero gradient)
        const int krow = min(k0 + 16 * w + l16, p.Tk - 1);
        typename EK::Frag kf;
        typename EV::Frag vf;
        {
            float rk_[DK / 4], rv_[DV / 4];
            EK::load_raw(rk_, p.k + ((long)b * p.Tk + krow) * p.ldk + h * DK, g);
            EK::to_frag(kf, rk_);
            EV::load_raw(rv_, p.v + ((long)b * p.Tk + krow) * p.ldv + h * DV, g);
            EV::to_frag(vf, rv_);
        }
        const float* qbase = p.q + (long)b * p.Tq * p.ldq + h * DK;
        const float* dobase = p.dO + (long)b * p.Tq * p.ldo + h * DV;
        const float* obase = p.Oc + (long)b * p.Tq * p.ldo + h * DV;
        float* Pw = Ps + w * 16 * AT_LDP;
        typename EK::Tile rq;
        TV rd, ro;
        float r_lse = 0.f, r_dl = 0.f;
        auto fetch_stats = [&](int qt) {
            if (tid < AT_TILE) {
                const int qi = min(qt * AT_TILE + tid, p.Tq - 1);
                r_lse = p.lse[(long)bh * p.Tq + qi];
                if (!OWN_DELTA) r_dl = p.delta[(long)bh * p.Tq + qi];
            }
        };
        if (qt0 < nqt) {
            rq.fetch(qbase, p.ldq, qt0 * AT_TILE, p.Tq, tid);
            rd.fetch(dobase, p.ldo, qt0 * AT_TILE, p.Tq, tid);
            if (OWN_DELTA) ro.fetch(obase, p.ldo, qt0 * AT_TILE, p.Tq, tid);
            fetch_stats(qt0);
        }
        for (int qt = qt0; qt < nqt; ++qt) {
            rq.commit(Qs, tid);
            rd.commit(Ds, tid);
            if (tid < AT_TILE) {
                lse_s[tid] = r_lse * 1.4426950408889634f;      // log2 domain
                if (!OWN_DELTA) dl_s[tid] = r_dl;
            }
            if (OWN_DELTA) {
#pragma unroll
                for (int i = 0; i < TV::NV; ++i) {   // delta of tile row tid / VPR + i RPP (rows past Tq were fetched as zeros)
                    float part = rd.v[i].x * ro.v[i].x + rd.v[i].y * ro.v[i].y + rd.v[i].z * ro.v[i].z + rd.v[i].w * ro.v[i].w;
                    part = tile_row_sum<TV::VPR>(part);
                    if (tid % TV::VPR == 0) dl_s[tid / TV::VPR + i * TV::RPP] = part;
                }
            }
            __syncthreads();
            if (qt + 1 < nqt) {
                rq.fetch(qbase, p.ldq, (qt + 1) * AT_TILE, p.Tq, tid);
                rd.fetch(dobase, p.ldo, (qt + 1) * AT_TILE, p.Tq, tid);
                if (OWN_DELTA) ro.fetch(obase, p.ldo, (qt + 1) * AT_TILE, p.Tq, tid);
                fetch_stats(qt + 1);
            }
            unsigned keepbits = 0xffffu;
            if (p.pmask) {
                keepbits = 0u;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int qi = qt * AT_TILE + 16 * t + l16;
                    const int key4 = k0 + 16 * w + 4 * g;            // 4 consecutive keys of one query row: one 4-byte load
                    unsigned word = 0u;
                    if (qi < p.Tq && key4 < p.Tk) word = *reinterpret_cast<const unsigned*>(p.pmask + ((long)bh * p.Tq + qi) * p.ldm + key4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) keepbits |= (((word >> (8 * r)) & 0xffu) ? 1u : 0u) << (4 * r + t);
                }
            }
            f32x4 st[4] = {zero_acc(), zero_acc(), zero_acc(), zero_acc()};
            f32x4 dpt[4] = {zero_acc(), zero_acc(), zero_acc(), zero_acc()};
            EK::mm_rows(st, kf, Qs, l16, g);               // S^T  [key][query]
            EV::mm_rows(dpt, vf, Ds, l16, g);              // dPd^T = V . dO^T
            float ds[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + 16 * w + 4 * g + r;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int qi = qt * AT_TILE + 16 * t + l16;
                    const bool valid = key < klim && qi < p.Tq && (!p.causal || key <= qi);
                    const float pr = valid ? __builtin_amdgcn_exp2f(st[t][r] * c2 - lse_s[16 * t + l16]) : 0.f;
                    float pd = pr, d = dpt[t][r];
                    if (p.pmask) {
                        const bool keep = (keepbits >> (4 * r + t)) & 1u;
                        pd = keep ? pr * p.pscale : 0.f;
                        d = keep ? d * p.pscale : 0.f;
                    }
                    ds[t][r] = pr * (d - dl_s[16 * t + l16]) * p.scale;
                    Pw[(4 * g + r) * AT_LDP + 16 * t + l16] = pd;
                }
            }
            wave_lds_sync();
            EV::mm_patch(dv, Pw, Ds, l16, g);              // dV += Pd^T . dO
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) Pw[(4 * g + r) * AT_LDP + 16 * t + l16] = ds[t][r];
            wave_lds_sync();
            EK::mm_patch(dk, Pw, Qs, l16, g);              // dK += dS^T . Q
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * w + 4 * g + r;
        if (key >= p.Tk) continue;
        float* kr = p.dk + ((long)b * p.Tk + key) * p.lddk + h * DK;
        float* vr = p.dv + ((long)b * p.Tk + key) * p.lddv + h * DV;
#pragma unroll
        for (int n = 0; n < DK / 16; ++n) kr[16 * n + l16] = dk[n][r];
#pragma unroll
        for (int n = 0; n < DV / 16; ++n) vr[16 * n + l16] = dv[n][r];
    }
}

// ROLES 3: both sides in one grid -- blockIdx.x < nkb -> key side of key block blockIdx.x (the longer role goes first), else query
// side; ROLES 1 / 2: the query side / the key side (delta from p.delta) alone, for the two-launch form of long sequences
template <int DK, int DV, int ROLES>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnP p, int nkb) {
    __shared__ __attribute__((aligned(16))) unsigned char As[Eng<DK>::TILE_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[Eng<DV>::TILE_BYTES];
    __shared__ __attribute__((aligned(16))) float Ps[4 * 16 * AT_LDP];
    __shared__ float lse_s[AT_TILE], dl_s[AT_TILE];
    if (ROLES == 3) {
        if ((int)blockIdx.x < nkb)
            attn_bwd_kv_role<DK, DV, true>(p, blockIdx.x, As, Bs, Ps, lse_s, dl_s);
        else
            attn_bwd_q_role<DK, DV>(p, blockIdx.x - nkb, As, Bs, Ps);
    } else if (ROLES == 1) {
        attn_bwd_q_role<DK, DV>(p, blockIdx.x, As, Bs, Ps);
    } else {
        attn_bwd_kv_role<DK, DV, false>(p, blockIdx.x, As, Bs, Ps, lse_s, dl_s);
    }
}

template <int DK, int DV>
void launch_attn_bwd(const AttnP& p, int nqb, int nkb, hipStream_t s) {
    const int bh = p.B * p.H;
    if ((long)(nqb + nkb) * bh <= 1024) {      // the two sides together are at most a few workgroups per CU: one launch
        hipLaunchKernelGGL((attn_bwd_kernel<DK, DV, 3>), dim3(nqb + nkb, bh), dim3(256), 0, s, p, nkb);
    } else {
        hipLaunchKernelGGL((attn_bwd_kernel<DK, DV, 1>), dim3(nqb, bh), dim3(256), 0, s, p, 0);
        hipLaunchKernelGGL((attn_bwd_kernel<DK, DV, 2>), dim3(nkb, bh), dim3(256), 0, s, p, nkb);
    }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// ------------------------------------------------------------------ one query row per (batch, head): incremental decoding
// A decoding step attends ONE new position of every hypothesis over the K / V cache (modules/decoder.py:131-185 re-runs the whole
// decoder on the growing prefix; PassEngine's decode session keeps a cache): the 64-row flash tile above spends 63 of its rows on
// nothing and 12.9 us per launch.  Here a workgroup owns one (batch, head): a thread scores one key (D fused multiply-adds on its own
// cache row), the softmax statistics are two block reductions, and the weighted sum of the value rows is taken by D / 4 lanes per row
// (16-byte loads, 256 / (D / 4) key groups, combined through LDS in a fixed order).  Exact fp32; the forward kernel's contract (klen
// masks keys, lse = log-sum-exp of the scaled scores); no dropout, no causal flag (a single query sees its whole prefix through klen).
constexpr int AD_MAXK = 4096;         // keys per (batch, head) the probabilities of which fit the workgroup's LDS
template <int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnP p) {
    constexpr int LPR = D / 4, G = 256 / LPR;                       // lanes per value row, key groups
    __shared__ float prob[AD_MAXK];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float part[G][D];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int klim = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    const float* kbase = p.k + (long)b * p.Tk * p.ldk + h * D;
    const float* vbase = p.v + (long)b * p.Tk * p.ldv + h * D;
    float4 q[LPR];
#pragma unroll
    for (int i = 0; i < LPR; ++i) q[i] = *reinterpret_cast<const float4*>(p.q + (long)b * p.ldq + h * D + 4 * i);
    float mx = -INFINITY;
    for (int j = tid; j < klim; j += 256) {
        const float4* kr = reinterpret_cast<const float4*>(kbase + (long)j * p.ldk);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LPR; ++i) {
            const float4 kv = kr[i];
            s = fmaf(q[i].w, kv.w, fmaf(q[i].z, kv.z, fmaf(q[i].y, kv.y, fmaf(q[i].x, kv.x, s))));
        }
        s *= p.scale;
        prob[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < klim; j += 256) {
        const float e = expf(prob[j] - mx);
        prob[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + w] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const int l4 = tid % LPR, g = tid / LPR;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int j = g; j < klim; j += G) {
        const float pj = prob[j];
        const float4 vv = *reinterpret_cast<const float4*>(vbase + (long)j * p.ldv + 4 * l4);
        acc.x = fmaf(pj, vv.x, acc.x), acc.y = fmaf(pj, vv.y, acc.y), acc.z = fmaf(pj, vv.z, acc.z), acc.w = fmaf(pj, vv.w, acc.w);
    }
    *reinterpret_cast<float4*>(&part[g][4 * l4]) = acc;
    __syncthreads();
    if (tid < D) {
        float o = 0.f;
        for (int gg = 0; gg < G; ++gg) o += part[gg][tid];
        p.O[(long)b * p.ldo + h * D + tid] = klim > 0 ? o / sum : 0.f;
    }
    if (tid == 0) p.lse[bh] = klim > 0 ? mx + logf(sum) : -INFINITY;
}

bool attn_args_ok(const AttnP& p, int dk, int dv) {
    if (!p.q || !p.k || !p.v || p.B <= 0 || p.H <= 0 || p.Tq <= 0 || p.Tk <= 0) return false;
    if (!((dk == 64 && dv == 64) || (dk == 16 && dv == 16))) return false;
    if (!al16(p.q) || !al16(p.k) || !al16(p.v) || (p.ldq & 3) || (p.ldk & 3) || (p.ldv & 3)) return false;
    if (p.ldq < p.H * dk || p.ldk < p.H * dk || p.ldv < p.H * dv) return false;
    if (p.pmask && p.ldm < p.Tk) return false;
    return true;
}

}  // namespace

extern "C" {

int mtl_attn_supported(int dk, int dv) { return (dk == 64 && dv == 64) || (dk == 16 && dv == 16); }

int mtl_attn_fwd(void* stream, const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int* klen,
                 int causal, float scale, int B, int H, int Tq, int Tk, int dk, int dv, const unsigned char* pmask, int ldm,
                 float pscale, float* O, int ldo, float* lse) {
    AttnP p{};
    p.q = q, p.k = k, p.v = v, p.ldq = ldq, p.ldk = ldk, p.ldv = ldv, p.klen = klen, p.causal = causal, p.scale = scale;
    p.B = B, p.H = H, p.Tq = Tq, p.Tk = Tk, p.pmask = pmask, p.ldm = ldm, p.pscale = pscale, p.O = O, p.ldo = ldo, p.lse = lse;
    if (!attn_args_ok(p, dk, dv) || !O || !lse || ldo < H * dv) return MTL_EINVAL;
    if (Tq == 1 && !pmask && Tk <= AD_MAXK && (!causal || Tk == 1)) {      // one query row per (batch, head): the decode kernel
        if (dk == 64)
            hipLaunchKernelGGL(attn_decode_kernel<64>, dim3(B * H), dim3(256), 0, as_stream(stream), p);
        else
            hipLaunchKernelGGL(attn_decode_kernel<16>, dim3(B * H), dim3(256), 0, as_stream(stream), p);
        MTL_CHECK_LAUNCH();
        return MTL_OK;
    }
    dim3 grid((Tq + AT_ROWS - 1) / AT_ROWS, B * H);
    if (dk == 64)
        hipLaunchKernelGGL((attn_fwd_kernel<64, 64>), grid, dim3(256), 0, as_stream(stream), p);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<16, 16>), grid, dim3(256), 0, as_stream(stream), p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_attn_bwd(void* stream, const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int* klen,
                 int causal, float scale, int B, int H, int Tq, int Tk, int dk, int dv, const unsigned char* pmask, int ldm,
                 float pscale, const float* O, const float* dO, int ldo, const float* lse, float* delta, float* dq, float* dk_,
                 float* dv_, int lddq, int lddk, int lddv) {
    AttnP p{};
    p.q = q, p.k = k, p.v = v, p.ldq = ldq, p.ldk = ldk, p.ldv = ldv, p.klen = klen, p.causal = causal, p.scale = scale;
    p.B = B, p.H = H, p.Tq = Tq, p.Tk = Tk, p.pmask = pmask, p.ldm = ldm, p.pscale = pscale, p.Oc = O, p.dO = dO, p.ldo = ldo;
    p.lse = const_cast<float*>(lse), p.delta = delta, p.dq = dq, p.dk = dk_, p.dv = dv_, p.lddq = lddq, p.lddk = lddk, p.lddv = lddv;
    if (!attn_args_ok(p, dk, dv) || !O || !dO || !lse || !delta || !dq || !dk_ || !dv_) return MTL_EINVAL;
    if (!al16(dO) || (ldo & 3) || ldo < H * dv || lddq < H * dk || lddk < H * dk || lddv < H * dv) return MTL_EINVAL;
    if (!al16(O)) return MTL_EINVAL;
    const int nqb = (Tq + AT_ROWS - 1) / AT_ROWS, nkb = (Tk + AT_ROWS - 1) / AT_ROWS;
    hipStream_t s = as_stream(stream);
    if (dk == 64)
        launch_attn_bwd<64, 64>(p, nqb, nkb, s);
    else
        launch_attn_bwd<16, 16>(p, nqb, nkb, s);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

}  // extern "C"
