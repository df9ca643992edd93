// Measurement aid, NOT part of the product library (its own libmtl_probe.so, loaded only by bench.py and tools/): the rate at which
// the chip sustains back-to-back v_mfma_f32_32x32x16_f16 on all CUs for a given OPERAND PATTERN.  On MI355X that rate is set by the
// package power limit, not by the instruction stream: with pseudo-random fp16 operands the matrix pipes run at ~0.6 of the nominal
// 2.5 PFLOP/s, with zeros at ~0.8 (profiles/r4/conv_experiments.txt).  bench.py reports the convolution kernels against this
// measured ceiling next to the nominal one.
//   mode 0: one workgroup of 8 waves per CU, each step = 24 matrix instructions per wave on register operands
//   mode 1: the fragment traffic of conv3x3_x3h_kernel<128, 2, ...> added: 16 ds_read_b128 per wave and step, one s_barrier per step
//   mode 4: the step of mode 1 on v_mfma_f32_16x16x32_f16 (4 x 4 tiles of 16 x 16, one 16-byte fragment per lane and tile row / column: same flops, same LDS bytes)
//   mode 2 / 3: modes 0 / 1 with the instructions that share an operand register issued back to back (an experiment on switching power)
//   fill 0 zeros | 1 pseudo-random fp16 in [-2, 2) | 2 the same with half of one operand zero (activations after a ReLU)
#include <hip/hip_runtime.h>

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ABYTES = 51840, BBYTES = 3 * 16384;

template <bool LDSREADS, int ORDER = 0>
__global__ __launch_bounds__(512) void probe_kernel(int steps, int fill, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    for (int i = threadIdx.x; i < (ABYTES + BBYTES) / 2; i += 512) {
        unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        unsigned short v = fill == 0 ? 0 : (unsigned short)((h & 0x8000u) | 0x3000u | (h & 0x0fffu));
        if (fill == 2 && i < ABYTES / 2 && ((i >> 2) * 2246822519u >> 31)) v = 0;
        reinterpret_cast<unsigned short*>(sm)[i] = v;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const unsigned char* aB = sm + (l31 + (wave & 3) * 32) * 80 + hi * 16;
    const unsigned char* bB = sm + ABYTES + ((wave >> 2) * 64 + l31) * 64 + ((hi ^ ((l31 >> 2) & 3)) * 16);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    uint4 a[2][2], b[2][2];
    auto reads = [&](int s_, int st) {
        const unsigned char* bS = bB + (s_ % 3) * 16384;
        const int toff = (s_ % 9) * 80;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) a[i][pc] = *reinterpret_cast<const uint4*>(aB + pc * 25920 + i * 128 * 80 + toff + st * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) b[j][pc] = *reinterpret_cast<const uint4*>(bS + pc * 8192 + j * 2048 + st * 32);
    };
    reads(0, 0);
    for (int s_ = 0; s_ < steps; ++s_) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (LDSREADS && ORDER != 2) reads(s_, st);
            if (ORDER == 2) {      // the same 64 x 64 x 32 wave-tile product on v_mfma_f32_16x16x32_f16: 4 x 4 tiles of 16 x 16, one ds_read_b128 per fragment
                if (st == 1) continue;                                 // (one k = 32 step per loop iteration)
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                uint4 af[4][2], bf[4][2];
                const int l15 = lane & 15, q = lane >> 4;
                const unsigned char* bS2 = sm + ABYTES + (s_ % 3) * 16384 + ((wave >> 2) * 64 + l15) * 64 + ((q ^ ((l15 >> 2) & 3)) * 16);
                const unsigned char* aS2 = sm + (l15 + (wave & 3) * 64) * 80 + q * 16 + (s_ % 9) * 80;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        af[i][pc] = *reinterpret_cast<const uint4*>(aS2 + pc * 25920 + i * 16 * 80);
                        bf[i][pc] = *reinterpret_cast<const uint4*>(bS2 + pc * 8192 + i * 16 * 64);
                    }
                f32x4* c4 = reinterpret_cast<f32x4*>(acc);             // 16 accumulators of 4 registers = the same 64 registers
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 c = c4[i * 4 + j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i][1]), __builtin_bit_cast(f16x8, bf[j][0]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, bf[j][1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, bf[j][0]), c, 0, 0, 0);
                        c4[i * 4 + j] = c;
                    }
            } else if (ORDER == 1) {      // operand-sharing order: the four instructions that take a[i][0] back to back, then the two with a[i][1]
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f16x8 a0 = __builtin_bit_cast(f16x8, a[i][0]), a1 = __builtin_bit_cast(f16x8, a[i][1]);
                    acc[i * 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, __builtin_bit_cast(f16x8, b[0][0]), acc[i * 2], 0, 0, 0);
                    acc[i * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, __builtin_bit_cast(f16x8, b[1][0]), acc[i * 2 + 1], 0, 0, 0);
                    acc[i * 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(f16x8, b[0][1]), acc[i * 2], 0, 0, 0);
                    acc[i * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(f16x8, b[1][1]), acc[i * 2 + 1], 0, 0, 0);
                    acc[i * 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(f16x8, b[0][0]), acc[i * 2], 0, 0, 0);
                    acc[i * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(f16x8, b[1][0]), acc[i * 2 + 1], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16& c = acc[i * 2 + j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][1]), __builtin_bit_cast(f16x8, b[j][0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][0]), c, 0, 0, 0);
                }
        }
        if (LDSREADS) __builtin_amdgcn_s_barrier();
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) sink[0] = s;
}
}  // namespace

extern "C" {
/* Enqueues `steps` steps of 8 waves x 24 v_mfma_f32_32x32x16_f16 per workgroup on `grid` workgroups (one per CU at grid = CU count).
 * FLOPs of the launch = grid * steps * 8 * 24 * 32768.  sink: 4 bytes of device memory.  Returns 0 or a hipError_t. */
int mtl_probe_mfma_f16(void* stream, int grid, int steps, int mode, int fill, float* sink) {
    const int smem = ABYTES + BBYTES;
    static int once = (hipFuncSetAttribute((const void*)probe_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) |
                       hipFuncSetAttribute((const void*)probe_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) |
                       hipFuncSetAttribute((const void*)probe_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) |
                       hipFuncSetAttribute((const void*)probe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) |
                       hipFuncSetAttribute((const void*)probe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    if (once) return once;
    if (mode == 4) hipLaunchKernelGGL((probe_kernel<true, 2>), dim3(grid), dim3(512), smem, (hipStream_t)stream, steps, fill, sink);
    else if (mode == 3) hipLaunchKernelGGL((probe_kernel<true, 1>), dim3(grid), dim3(512), smem, (hipStream_t)stream, steps, fill, sink);
    else if (mode == 2) hipLaunchKernelGGL((probe_kernel<false, 1>), dim3(grid), dim3(512), smem, (hipStream_t)stream, steps, fill, sink);
    else if (mode) hipLaunchKernelGGL(probe_kernel<true>, dim3(grid), dim3(512), smem, (hipStream_t)stream, steps, fill, sink);
    else hipLaunchKernelGGL(probe_kernel<false>, dim3(grid), dim3(512), smem, (hipStream_t)stream, steps, fill, sink);
    return (int)hipGetLastError();
}
}
