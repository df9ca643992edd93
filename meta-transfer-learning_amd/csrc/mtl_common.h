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

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int grid_for(long n, int per_block, int cap = 4096) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
