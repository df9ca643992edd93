// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950 (hip_ext.h says the flag is not supported on GFX9)?
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/anyorder.hip -o tools/probe/anyorder ; run on the GPU box
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void spin(long cycles, int* out) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
    }
    if (out) out[0] = 1;
}

static double run(hipStream_t s, int flags, int n) {
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, (i ? flags : 0), 20000L, (int*)nullptr);
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    run(s, 0, 2);
    for (int rep = 0; rep < 3; ++rep) {
        const double a = run(s, 0, 8), b = run(s, hipExtAnyOrderLaunch, 8);
        printf("8 x 200 us spin kernels (100 MHz wall clock), one stream: in order %.0f us, any-order flag %.0f us\n", a, b);
    }
    return 0;
}
