// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the meta-transfer hot path.
// wave = 64 lanes everywhere; nothing here is portable to 32-wide hardware on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MTL_OK 0
#define MTL_EINVAL (-22)
#define MTL_ELAUNCH (-5)

#define MTL_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return MTL_ELAUNCH;          \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// big-engine GEMM with bias strides for both batch levels and a third, outermost batch level (`tasks` items of batch / tasks
// each: the tasks of a meta-step in one launch) (mtl_mfma.hip; what mtl_gemm_f32_tb forwards to)
int mtl_gemm_f32_3l(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                    int ldb, float* C, int ldc, const float* bias, const float* gate, int ldg, int flags, int batch, int H, long sAb,
                    long sAh, long sBb, long sBh, long sCb, long sCh, long sBias, long sBiasH, float* workspace, long workspace_bytes,
                    int tasks, long sAt, long sBt, long sCt, long sBiasT);

// bf16-split engine (mtl_gemm_x3.hip): 1 = issued there, 0 = not eligible (stay on the fp32 engines), < 0 = error
int mtl_gemm_x3_route(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                      int ldb, float* C, int ldc, const float* bias, const float* gate, int ldg, int flags, int batch, int H, long sAb,
                      long sAh, long sBb, long sBh, long sCb, long sCh, long sBias, int kbatch, long sAk, long sBk, float* rowsum,
                      long sRowsum, long sBiasH, long sRowsumH, int tasks, long sAt, long sBt, long sCt, long sBiasT, long sRowsumT);

int mtl_gemm_x3_eligible(int M, int N, int batch);
int mtl_gemm_x3_splitk_slices(int transA, int transB, int M, int N, int K, int flags, long ws_bytes);
int mtl_gemm_x3_splitk(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                       int ldb, float* C, int ldc, const float* bias, int flags, float* ws, long ws_bytes);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#ifndef MTL_AMAX_SLOTS
#define MTL_AMAX_SLOTS 64
#define MTL_AMAX_STRIDE 32
#define MTL_AMAX_FLOATS (MTL_AMAX_SLOTS * MTL_AMAX_STRIDE)
#endif
// A bound max|tensor| is kept in MTL_AMAX_SLOTS floats, one 128-byte line apart (include/mtl_hip.h: L2 serialises atomics per
// LINE, 8 k raises of one line cost 40 us): readers take the maximum of the slot heads (full waves only) ...
__device__ __forceinline__ float amax_read(const float* a) { return wave_max(a[(threadIdx.x & (MTL_AMAX_SLOTS - 1)) * MTL_AMAX_STRIDE]); }
// ... writers raise the slot of their workgroup.  Same-address atomics (and coherent loads) serialise at ~9 ns each: only a wave
// that would RAISE its slot issues one, judged by a plain CACHED load (a stale smaller value only costs a redundant atomic).
__device__ __forceinline__ void amax_raise(float* a, float mx) {
    mx = wave_max(mx);
    float* slot = a + (blockIdx.x & (MTL_AMAX_SLOTS - 1)) * MTL_AMAX_STRIDE;
    if ((threadIdx.x & 63) == 0 && mx > *slot) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(mx));
}

static inline int grid_for(long n, int per_block, int cap = 4096) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
