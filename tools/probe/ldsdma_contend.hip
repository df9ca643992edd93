// What slows the LDS-DMA weight stream inside the convolution kernel?  One loader wave per CU fills 16 KiB slots (16 x 1 KiB
// global_load_lds_dwordx4, two tiles in flight) from an L2-resident buffer while 8 consumer waves run, per "step", R ds_read_b128
// and M v_mfma_f32_32x32x16_f16 each (the convolution: R = 16, M = 24).  Free-running (no barriers): the loader's tile period is
// measured on its own.   hipcc --offload-arch=gfx950 -O3 tools/probe/ldsdma_contend.hip -o tools/probe/ldsdma_contend
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int TILE = 16384, NTILE = 36, RING = 3;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int M, int LOADER, int REG = 0>
__global__ __launch_bounds__(768) void k(const unsigned char* w, int iters, float* sink, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 8) {
        if (wave - 8 >= LOADER) return;
        constexpr int PER = LOADER ? 16 / LOADER : 16;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (REG == 2) {  // plain global loads into registers, no LDS at all (two tiles in flight)
            uint4 ra[PER], rb[PER];
            unsigned x = 0;
            const unsigned char* base = w + lane * 16 + (wave - 8) * PER * 1024;
#pragma unroll
            for (int i = 0; i < PER; ++i) ra[i] = *reinterpret_cast<const uint4*>(base + i * 1024);
            for (int it = 0; it < iters; it += 2) {
                const unsigned char* s1 = base + (long)((it + 1) % NTILE) * TILE;
#pragma unroll
                for (int i = 0; i < PER; ++i) rb[i] = *reinterpret_cast<const uint4*>(s1 + i * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < PER; ++i) x ^= ra[i].x;
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* s2 = base + (long)((it + 2) % NTILE) * TILE;
#pragma unroll
                for (int i = 0; i < PER; ++i) ra[i] = *reinterpret_cast<const uint4*>(s2 + i * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < PER; ++i) x ^= rb[i].x;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (x == 0x12345) sink[1] = 1.f;
        } else if (REG) {      // register path: the tile of step it + 1 is in flight while the tile of step it is written to LDS
            uint4 ra[PER], rb[PER];
            auto ld = [&](uint4* r, int it) {
                const unsigned char* src = w + (long)(it % NTILE) * TILE + lane * 16 + (wave - 8) * PER * 1024;
#pragma unroll
                for (int i = 0; i < PER; ++i) r[i] = *reinterpret_cast<const uint4*>(src + i * 1024);
            };
            auto stw = [&](const uint4* r, int it) {
                unsigned char* dst = sm + (it % RING) * TILE + (wave - 8) * PER * 1024 + lane * 16;
#pragma unroll
                for (int i = 0; i < PER; ++i) *reinterpret_cast<uint4*>(dst + i * 1024) = r[i];
            };
            ld(ra, 0);
            for (int it = 0; it < iters; it += 2) {
                ld(rb, it + 1);
                __builtin_amdgcn_sched_barrier(0);
                stw(ra, it);
                __builtin_amdgcn_sched_barrier(0);
                ld(ra, it + 2);
                __builtin_amdgcn_sched_barrier(0);
                stw(rb, it + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        for (int it = 0; it < iters; ++it) {
            const unsigned char* src = w + (long)(it % NTILE) * TILE + lane * 16 + (wave - 8) * PER * 1024;
            unsigned char* dst = sm + (it % RING) * TILE + (wave - 8) * PER * 1024;
#pragma unroll
            for (int i = 0; i < PER; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && wave == 8) ticks[blockIdx.x * 2] = t1 - t0;
        return;
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        uint4 f[R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < R; ++r) f[r] = *reinterpret_cast<const uint4*>(sm + ((it + r) % RING) * TILE + ((wave * 64 + lane + r * 7) & 1023) * 16);
        if (M < 0) {
#pragma unroll
            for (int m = 0; m < -M * 8; ++m) acc[m & 3][m & 15] = __builtin_fmaf(acc[m & 3][m & 15], 1.0001f, 0.5f);
        }
        if (M == 99) {          // half duty: 12 matrix instructions, then sleep about as long
#pragma unroll
            for (int m = 0; m < 12; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)(_Float16)1.f, (f16x8)(_Float16)1.f, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_s_sleep(12);
        }
#pragma unroll
        for (int m = 0; m < (M == 99 ? 0 : M); ++m) {
            f16x8 a, b;
            if (R > 0) {
                a = __builtin_bit_cast(f16x8, f[m % R]);
                b = __builtin_bit_cast(f16x8, f[(m + 1) % R]);
            } else {
                a = (f16x8)(_Float16)1.f; b = a;
            }
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
        }
        if (M == 0 && R > 0) {
            float s = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) s += __builtin_bit_cast(float, f[r].x);
            acc[0][0] += s;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && wave == 0) { ticks[blockIdx.x * 2 + 1] = t1 - t0; ticks[1024 + blockIdx.x] = t0; ticks[2048 + blockIdx.x] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 4) | (__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 20) ? 0 : 0); }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) sink[0] = s;
}

template <int R, int M, int LOADER, int REG = 0>
void run(const unsigned char* w, float* sink, unsigned long long* ticks, const char* name) {
    const int iters = 1000, grid = 256;
    (void)hipFuncSetAttribute((const void*)k<R, M, LOADER, REG>, hipFuncAttributeMaxDynamicSharedMemorySize, RING * TILE);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<R, M, LOADER, REG><<<grid, 768, RING * TILE>>>(w, iters, sink, ticks);
    (void)hipDeviceSynchronize();
    (void)hipMemset(ticks, 0, grid * 16);
    (void)hipEventRecord(a);
    k<R, M, LOADER, REG><<<grid, 768, RING * TILE>>>(w, iters, sink, ticks);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h[512];
    (void)hipMemcpy(h, ticks, grid * 16, hipMemcpyDeviceToHost);
    double tl = 0, tc = 0;
    for (int i = 0; i < grid; ++i) { tl += h[2 * i]; tc += h[2 * i + 1]; }
    if (R == 0 && M == 24 && LOADER == 0) {
        unsigned long long st[256], hw[256];
        (void)hipMemcpy(st, ticks + 1024, 256 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hw, ticks + 2048, 256 * 8, hipMemcpyDeviceToHost);
        unsigned long long mn = ~0ull;
        for (int i = 0; i < 256; ++i) mn = st[i] < mn ? st[i] : mn;
        int late = 0;
        for (int i = 0; i < 256; ++i) late += (st[i] - mn) > 1000;
        printf("   workgroups starting more than 10 us after the first: %d of 256; start offsets (x10ns) of wg 0..15:", late);
        for (int i = 0; i < 16; ++i) printf(" %llu", st[i] - mn);
        printf("\n");
    }
    printf("%-44s kernel %.3f us/step | loader %.0f x10ns/tile | consumer %.0f x10ns/step\n", name, ms * 1e3 / iters, tl / grid / iters, tc / grid / iters);
}

int main() {
    unsigned char* w; float* sink; unsigned long long* ticks;
    (void)hipMalloc(&w, NTILE * TILE); (void)hipMemset(w, 0x3c, NTILE * TILE); (void)hipMalloc(&sink, 8); (void)hipMalloc(&ticks, 8 * 4096);
    run<0, 0, 1>(w, sink, ticks, "loader alone");
    run<16, 24, 0>(w, sink, ticks, "consumers alone (16 reads + 24 MFMA)");
    run<0, 24, 0>(w, sink, ticks, "consumers alone (24 MFMA, no reads)");
    run<16, 24, 1>(w, sink, ticks, "1 loader + consumers (16 reads + 24 MFMA)");
    run<16, 24, 2>(w, sink, ticks, "2 loaders + consumers (16 reads + 24 MFMA)");
    run<16, 24, 4>(w, sink, ticks, "4 loaders + consumers (16 reads + 24 MFMA)");
    run<0, 24, 1>(w, sink, ticks, "1 loader + consumers (24 MFMA, no reads)");
    run<0, 24, 4>(w, sink, ticks, "4 loaders + consumers (24 MFMA, no reads)");
    run<16, 0, 1>(w, sink, ticks, "1 loader + consumers (16 reads, no MFMA)");
    run<0, -24, 1>(w, sink, ticks, "1 loader + consumers (192 v_fma, no MFMA)");
    run<0, 99, 1>(w, sink, ticks, "1 loader + consumers (12 MFMA + sleep)");
    run<0, 0, 1, 2>(w, sink, ticks, "plain-load loader alone");
    run<0, 24, 1, 2>(w, sink, ticks, "1 plain-load loader + consumers (24 M)");
    run<0, 24, 4, 2>(w, sink, ticks, "4 plain-load loaders + consumers (24 M)");
    run<16, 24, 4, 2>(w, sink, ticks, "4 plain-load loaders + consumers (16 r + 24 M)");
    return 0;
}
