// Probe of ds_read_b64_tr_b16 semantics on gfx950: every lane supplies the address of 4 consecutive 16-bit values
// (value = lane * 4 + element), and we print what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* dst) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[256];
    for (int i = threadIdx.x; i < 256; i += 64) sm[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    auto v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + lane * 4));
    for (int j = 0; j < 4; ++j) dst[lane * 4 + j] = v[j];
}
// second probe: per-lane addresses with a free row stride (rows of 40 values, group g at column 16 g)
__global__ void k2(unsigned short* dst) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, x = lane & 15;
    auto v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + (x >> 2) * 40 + (g & 1) * 16 + (g >> 1) * 320 + (x & 3) * 4));
    for (int j = 0; j < 4; ++j) dst[lane * 4 + j] = v[j];
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256];
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        for (int j = 0; j < 4; ++j) {
            const int src_lane = g * 16 + 4 * j + (i >> 2), want = src_lane * 4 + (i & 3);
            if (h[l * 4 + j] != want) ++bad;
        }
    }
    for (int l = 0; l < 20; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    printf("model result(lane i, elem j) = data[lane 4j + (i>>2)][i&3] per 16-lane group: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        for (int j = 0; j < 4; ++j) {
            const int want = j * 40 + (g & 1) * 16 + (g >> 1) * 320 + i;      // row j, column i of the group's [4][16] block
            if (h[l * 4 + j] != want) ++bad;
        }
    }
    printf("strided rows (40 values), result(lane i, elem j) = block[row j][col i]: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    return 0;
}
