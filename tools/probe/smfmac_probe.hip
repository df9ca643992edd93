// Empirical register layout of v_smfmac_f32_32x32x32_f16 on gfx950 (2:4 structured-sparse A, dense B).
//   hipcc --offload-arch=gfx950 -O2 tools/probe/smfmac_probe.hip -o tools/probe/smfmac_probe && tools/probe/smfmac_probe
// Prints, for every compressed A element (lane, e) : its output row m, the lane / bit field of the index register that steers
// it, and -- for every 2-bit index value -- which B element (lane, e) of output column 0 it multiplies.  B elements carry unique
// integer codes (lane * 16 + e, exact in fp16), so ONE instruction per (A element, index value) reveals the selected B element.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Job {
    int la, ea;        // one-hot A element (la < 0: A all ones)
    int lb, eb;        // one-hot B element (lb < 0: B coded: value = 1 + lane * 16 + e)
    unsigned idx_all;  // index register of every lane ...
    int lx;            // ... except lane lx, which gets idx_x
    unsigned idx_x;
    int abid;
};

template <int ABID>
__device__ f32x16 run(h8 a, h16 b, unsigned idx) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    return __builtin_amdgcn_smfmac_f32_32x32x32_f16(a, b, acc, (int)idx, 0, ABID);
}

__global__ void probe(const Job* jobs, float* out) {   // one wave per job; out[job][32 rows][32 cols]
    const Job j = jobs[blockIdx.x];
    const int l = threadIdx.x;
    h8 a;
    h16 b;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)((j.la < 0) ? 1.f : ((j.la == l && j.ea == e) ? 1.f : 0.f));
    for (int e = 0; e < 16; ++e)
        b[e] = (_Float16)((j.lb == -1) ? (float)(1 + l * 16 + e) : (j.lb == -2 ? 1.f : ((j.lb == l && j.eb == e) ? 1.f : 0.f)));
    const unsigned idx = (l == j.lx) ? j.idx_x : j.idx_all;
    f32x16 acc = j.abid == 0 ? run<0>(a, b, idx) : (j.abid == 1 ? run<1>(a, b, idx) : (j.abid == 2 ? run<2>(a, b, idx) : run<3>(a, b, idx)));
    float* o = out + (long)blockIdx.x * 1024;
    const int l31 = l & 31, hi = l >> 5;
    for (int v = 0; v < 16; ++v) o[(8 * (v >> 2) + 4 * hi + (v & 3)) * 32 + l31] = acc[v];   // standard 32x32 C layout: row, col = l31
}

int main() {
    std::vector<Job> jobs;
    auto add = [&](int la, int ea, int lb, int eb, unsigned all, int lx, unsigned ix, int abid) {
        jobs.push_back(Job{la, ea, lb, eb, all, lx, ix, abid});
        return (int)jobs.size() - 1;
    };
    const unsigned REP[4] = {0x00000000u, 0x55555555u, 0xAAAAAAAAu, 0xFFFFFFFFu};
    // T1/T3: A one-hot, B coded, all index fields = v
    int t3[64][8][4];
    for (int la = 0; la < 64; ++la)
        for (int ea = 0; ea < 8; ++ea)
            for (int v = 0; v < 4; ++v) t3[la][ea][v] = add(la, ea, -1, 0, REP[v], -1, 0, 0);
    // T2: B one-hot, A all ones, all fields = v
    int t2[64][16][4];
    for (int lb = 0; lb < 64; ++lb)
        for (int eb = 0; eb < 16; ++eb)
            for (int v = 0; v < 4; ++v) t2[lb][eb][v] = add(-1, 0, lb, eb, REP[v], -1, 0, 0);
    // T4: which lane's register / which bit field steers A element (la, ea): all lanes 0, lane lx = 3 << (2 j)
    //     (only for la in {0, 5, 37}: the lane dependence; lx in {la, la ^ 32, (la + 1) & 63})
    const int LAS[3] = {0, 5, 37};
    int t4[3][8][3][16];
    for (int a = 0; a < 3; ++a)
        for (int ea = 0; ea < 8; ++ea)
            for (int w = 0; w < 3; ++w)
                for (int jb = 0; jb < 16; ++jb) {
                    const int la = LAS[a], lx = w == 0 ? la : (w == 1 ? (la ^ 32) : ((la + 1) & 63));
                    t4[a][ea][w][jb] = add(la, ea, -1, 0, 0u, lx, 3u << (2 * jb), 0);
                }
    // T5: abid = 1 with the fields in the upper 16 bits (lane la's own register)
    int t5[3][8][16];
    for (int a = 0; a < 3; ++a)
        for (int ea = 0; ea < 8; ++ea)
            for (int jb = 0; jb < 16; ++jb) t5[a][ea][jb] = add(LAS[a], ea, -1, 0, 0u, LAS[a], 3u << (2 * jb), 1);

    Job* dj;
    float* dout;
    const size_t n = jobs.size();
    if (hipMalloc(&dj, n * sizeof(Job)) != hipSuccess || hipMalloc(&dout, n * 1024 * sizeof(float)) != hipSuccess) return 2;
    hipMemcpy(dj, jobs.data(), n * sizeof(Job), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3((unsigned)n), dim3(64), 0, 0, dj, dout);
    if (hipDeviceSynchronize() != hipSuccess) {
        fprintf(stderr, "kernel failed\n");
        return 3;
    }
    std::vector<float> out(n * 1024);
    hipMemcpy(out.data(), dout, n * 1024 * sizeof(float), hipMemcpyDeviceToHost);
    auto at = [&](int job, int m, int c) { return out[(size_t)job * 1024 + m * 32 + c]; };
    // the single non-zero row of a one-hot-A job and its value in column c
    auto row_of = [&](int job, int c, float& val) {
        int m = -1;
        val = 0.f;
        for (int r = 0; r < 32; ++r)
            if (at(job, r, c) != 0.f) {
                if (m >= 0) m = -2;
                else m = r, val = at(job, r, c);
            }
        return m;
    };
    printf("# T3: A element (lane, e) -> row m ; per index value v the B element (lane:e) of columns 0 and 1 it multiplies\n");
    for (int la = 0; la < 64; ++la)
        for (int ea = 0; ea < 8; ++ea) {
            printf("A %2d %d :", la, ea);
            for (int v = 0; v < 4; ++v) {
                float x0, x1;
                const int m0 = row_of(t3[la][ea][v], 0, x0), m1 = row_of(t3[la][ea][v], 1, x1);
                const int c0 = (int)x0 - 1, c1 = (int)x1 - 1;
                printf("  v%d m%d/%d B0=%d:%d B1=%d:%d", v, m0, m1, c0 >= 0 ? c0 / 16 : -1, c0 >= 0 ? c0 % 16 : -1, c1 >= 0 ? c1 / 16 : -1,
                       c1 >= 0 ? c1 % 16 : -1);
            }
            printf("\n");
        }
    printf("# T2: B element (lane, e) -> column n ; per index value v (all fields) the value seen in row 0 of that column (A all ones)\n");
    for (int lb = 0; lb < 64; ++lb)
        for (int eb = 0; eb < 16; ++eb) {
            printf("B %2d %2d :", lb, eb);
            for (int v = 0; v < 4; ++v) {
                int col = -1;
                float val = 0.f;
                for (int c = 0; c < 32; ++c)
                    if (at(t2[lb][eb][v], 0, c) != 0.f) col = c, val = at(t2[lb][eb][v], 0, c);
                printf("  v%d n%d x%g", v, col, val);
            }
            printf("\n");
        }
    printf("# T4: A element (lane, e), all index registers 0 except lane lx = 3 << 2j: selected B element of column 0 per j (baseline = v0 of T3)\n");
    for (int a = 0; a < 3; ++a)
        for (int ea = 0; ea < 8; ++ea)
            for (int w = 0; w < 3; ++w) {
                printf("X la%2d e%d lx=%s :", LAS[a], ea, w == 0 ? "la" : (w == 1 ? "la^32" : "la+1"));
                for (int jb = 0; jb < 16; ++jb) {
                    float x0;
                    row_of(t4[a][ea][w][jb], 0, x0);
                    const int c0 = (int)x0 - 1;
                    printf(" %d:%d", c0 >= 0 ? c0 / 16 : -1, c0 >= 0 ? c0 % 16 : -1);
                }
                printf("\n");
            }
    printf("# T5: same as T4 (lx = la) with ABID = 1\n");
    for (int a = 0; a < 3; ++a)
        for (int ea = 0; ea < 8; ++ea) {
            printf("Y la%2d e%d :", LAS[a], ea);
            for (int jb = 0; jb < 16; ++jb) {
                float x0;
                row_of(t5[a][ea][jb], 0, x0);
                const int c0 = (int)x0 - 1;
                printf(" %d:%d", c0 >= 0 ? c0 / 16 : -1, c0 >= 0 ? c0 % 16 : -1);
            }
            printf("\n");
        }
    return 0;
}
