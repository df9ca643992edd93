// LDS-DMA (global_load_lds_dwordx4) vs register-path (global_load_dwordx4 + ds_write_b128) fill rate of a 16 KiB LDS slot per CU
// from an L2-resident buffer (every workgroup streams the same 576 KiB, like the weight tiles of conv7), with 1 / 2 / 4 loader waves.
// hipcc --offload-arch=gfx950 -O3 tools/probe/ldsdma_rate.hip -o tools/probe/ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int TILE = 16384, NTILE = 36, RING = 3;

template <int LOADERS, bool DMA>
__global__ __launch_bounds__(256) void fill_kernel(const unsigned char* w, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int PER = 16 / LOADERS;          // 1 KiB instructions per loader wave and tile
    float acc = 0.f;
    if (wave < LOADERS) {
        uint4 r[2][PER];
        for (int it = 0; it < iters; ++it) {
            const unsigned char* src = w + (long)(it % NTILE) * TILE + wave * PER * 1024 + lane * 16;
            unsigned char* dst = sm + (it % RING) * TILE + wave * PER * 1024;
            if (DMA) {
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");      // previous tile landed
            } else {
#pragma unroll
                for (int i = 0; i < PER; ++i) r[it & 1][i] = *reinterpret_cast<const uint4*>(src + i * 1024);
                if (it) {
                    unsigned char* pd = sm + ((it - 1) % RING) * TILE + wave * PER * 1024 + lane * 16;
#pragma unroll
                    for (int i = 0; i < PER; ++i) *reinterpret_cast<uint4*>(pd + i * 1024) = r[(it - 1) & 1][i];
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    acc = *reinterpret_cast<float*>(sm + threadIdx.x * 4);
    if (acc == 12345.678f) sink[0] = acc;
}

template <int LOADERS, bool DMA>
void run(const unsigned char* w, float* sink, const char* name) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)fill_kernel<LOADERS, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, RING * TILE);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {8, 256}) {
        fill_kernel<LOADERS, DMA><<<grid, 256, RING * TILE>>>(w, iters, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        fill_kernel<LOADERS, DMA><<<grid, 256, RING * TILE>>>(w, iters, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("%-28s grid %3d: %.3f us per 16 KiB tile = %.1f GB/s per CU, %.2f TB/s chip\n", name, grid, ms * 1e3 / iters,
               TILE / (ms * 1e-3 / iters) / 1e9, grid * (double)TILE / (ms * 1e-3 / iters) / 1e12);
    }
}

int main() {
    unsigned char* w; float* sink;
    hipMalloc(&w, NTILE * TILE); hipMemset(w, 1, NTILE * TILE); hipMalloc(&sink, 4);
    run<1, true>(w, sink, "LDS-DMA, 1 loader wave");
    run<2, true>(w, sink, "LDS-DMA, 2 loader waves");
    run<4, true>(w, sink, "LDS-DMA, 4 loader waves");
    run<1, false>(w, sink, "registers, 1 loader wave");
    run<2, false>(w, sink, "registers, 2 loader waves");
    run<4, false>(w, sink, "registers, 4 loader waves");
    return 0;
}
