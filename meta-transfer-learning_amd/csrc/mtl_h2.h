// Two-piece fp16 ("h2") arithmetic shared by the convolution and GEMM kernels (gfx950).
// x s = h + l (+ at most 2^-23 |x s|) with two fp16 pieces (2 x 11 significand bits), s a power of two that puts the LARGEST
// magnitude of the tensor into [2^14, 2^15) -- fp16 has 5 exponent bits, so the caller passes an upper bound of max|x| (device
// floats: `amax`, MTL_AMAX_SLOTS of them).  An element keeps the full 22 bits of both pieces while |x| >= 2^-17.5 of that maximum
// (the low piece, <= 2^-11 |x s|, is a NORMAL fp16 down to 2^-14); below, the low piece is a subnormal fp16 (spacing 2^-24) and
// the element carries an absolute error of at most 2^-39 of the maximum -- a relative error of 2^(k - 38.5) for an element 2^k
// below the maximum (tests/test_ops_gpu.py::test_conv3x3_two_piece_fp16_dynamic_range_inside_one_tensor measures a quiet
// sample beside a loud one).  The tensors of the path are normalised per utterance and stay in the first regime.
// A product is h h' + h l' + l h' (the dropped l l' is < 2^-22 relative): three
// v_mfma_f32_32x32x16_f16 instead of six bf16 ones (x3) or eight fp32 ones, fp32 accumulation, un-scaled exactly (powers of two)
// in the epilogue.  Measured against fp64 the kernels built on this are as accurate as the exact-fp32 MFMA kernels (both are
// dominated by the fp32 accumulation chain; tests/test_ops_gpu.py).
#pragma once
#include "mtl_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float pow2_scale(float amax) {   // 2^k with amax 2^k in [2^14, 2^15); 1 for amax = 0 / denormal
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);     // amax in [2^(e-127), 2^(e-126))
    int k = 141 - e;
    k = e == 0 ? 0 : min(max(k, -60), 60);
    return __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
}

__device__ __forceinline__ void split2x2(float x0, float x1, unsigned& h, unsigned& l) {
    const f16x2 hh = __builtin_convertvector(f32x2{x0, x1}, f16x2);
    const f32x2 r = f32x2{x0, x1} - __builtin_convertvector(hh, f32x2);            // exact
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// acc += (a_h + a_l)(b_h + b_l) without the l l' term; operands are 8 k-values per lane (MFMA 32x32x16 layout), [0] = h, [1] = l
__device__ __forceinline__ f32x16 h2_mfma(const uint4 (&a)[2], const uint4 (&b)[2], f32x16 cc) {
    const f16x8 a0 = __builtin_bit_cast(f16x8, a[0]), a1 = __builtin_bit_cast(f16x8, a[1]);
    const f16x8 b0 = __builtin_bit_cast(f16x8, b[0]), b1 = __builtin_bit_cast(f16x8, b[1]);
    cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, cc, 0, 0, 0);           // smallest terms first
    cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, cc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, cc, 0, 0, 0);
}

// LDS / prepared-weight rows hold 32 k-values (64 bytes) as four 16-byte chunks; chunk c of row r sits at c ^ ((r >> 2) & 3):
// a ds_read_b128 of one chunk from 16 rows {0-3, 12-15, 20-27} (what a half-wave of an MFMA fragment read touches) is conflict-free
__host__ __device__ inline int x3_swz(int k, int row) { return (((k >> 3) ^ ((row >> 2) & 3)) << 3) | (k & 7); }
