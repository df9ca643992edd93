// Persistent LSTM layer kernels for gfx950 (MI355X): all T time steps of one layer in ONE launch per direction.
//
// The LM meta loop (lm/main_meta_transfer.py:277-411 on lm/model/rnn_model.py:12-70: 2-layer nn.LSTM, bptt 35, batch 20) is a chain
// of T dependent steps per layer; as separate launches every step is a 42 MFLOP recurrent product + a cell kernel (280 launches
// per pass at ~12 us each).  Here workgroup w owns the hidden units j in [8 w, 8 w + 8) for the whole sequence:
//   * its 32 rows of W_hh (gate-major: rows g H + j) stay in REGISTERS across the steps -- thread (r = tid & 31, kc = tid >> 5)
//     holds W_hh[row r][kc H/8 .. + H/8) (<= 64 registers);
//   * per step the previous hidden state (B x H, <= 64 KB) is staged in LDS and every thread multiplies its row chunk with the
//     B batch rows (operand reads are LDS broadcasts), the 8 chunk partials of an output are summed in a fixed order, the
//     threads (b, unit) apply the cell (same formulas as lstm_cell_fwd_kernel, mtl_elem.hip) and store h_t, c_t, the gate
//     activations and the (dropped) copy for the next layer;
//   * a grid-wide hand-off per step: plain stores -> workgroup barrier -> one lane: agent-scope release + arrival on a monotonic
//     counter; consumers: one lane polls (relaxed, agent scope), agent-scope acquire, workgroup barrier, plain loads.  Every spin
//     is bounded (an error word is set instead of hanging the device); the grid is H / 8 <= 64 workgroups, far below the 256 CUs,
//     so all of them are resident.
// The backward runs the same ownership in reverse time: workgroup w produces dh_rec[:, its units] = dG_{t+1} . W_hh[:, its units]
// (the column slice of W_hh in registers; dG_{t+1}, B x 4H, staged through LDS a quarter at a time), applies the cell backward
// for its units (dc carried in registers) and stores its 4 x 8 columns of dG_t.  The weight gradients remain three products over
// all T steps (lm.py).  Reductions are fixed-order: bitwise reproducible.
#include "mtl_common.h"
#include "../../include/mtl_hip.h"

namespace {

constexpr int LU = 8;                        // hidden units per workgroup
constexpr unsigned SPIN_LIMIT = 1u << 22;    // polls before a wait gives up (~seconds): sets the error word

struct LstmP {
    const float *gx, *whh, *bhh;
    float *hall, *call, *acts, *xout;
    const uint8_t* mask;
    float mscale;
    int T, B, H;
    unsigned* sync;                           // [0] arrivals (zeroed by the launcher), [1] error word
    // backward
    const float* dx_up;
    float* dG;
};

__device__ __forceinline__ void grid_wait(unsigned* sync, unsigned target) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) {
                __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __forceinline__ void grid_arrive(unsigned* sync) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (restated where the compiler cannot drop it: guide, G16 pitfall)
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int KC>      // H = 8 KC
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void lstm_layer_fwd_kernel(LstmP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = p.B, H = 8 * KC, T = p.T;
    float* hs = sm;                     // [B][H]
    float* red = sm + B * H;            // [8 chunks][B][32 rows]
    const int tid = threadIdx.x, r = tid & 31, kc = tid >> 5;
    const int j0 = blockIdx.x * LU;
    const unsigned nwg = gridDim.x;
    float w[KC];
    {
        const float* wrow = p.whh + (long)((r >> 3) * H + j0 + (r & 7)) * H + kc * KC;
#pragma unroll
        for (int i = 0; i < KC; i += 4) {
            const float4 v = *reinterpret_cast<const float4*>(wrow + i);
            w[i] = v.x, w[i + 1] = v.y, w[i + 2] = v.z, w[i + 3] = v.w;
        }
    }
    const bool cell = tid < B * LU;
    const int cb = tid / LU, cu = tid % LU, cj = j0 + cu;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    float c_prev = 0.f;
    if (cell) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[g] = p.bhh[g * H + cj];
        c_prev = p.call[(long)cb * H + cj];
    }
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (t > 0) grid_wait(p.sync, (unsigned)t * nwg);        // every workgroup has published h_t
        const float4* src = reinterpret_cast<const float4*>(p.hall + (long)t * B * H);
        for (int i = tid; i < B * H / 4; i += 256) reinterpret_cast<float4*>(hs)[i] = src[i];
        __syncthreads();
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            const float* hb = hs + b * H + kc * KC;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int i = 0; i < KC; i += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(hb + i);
                a0 = fmaf(w[i], hv.x, a0);
                a1 = fmaf(w[i + 1], hv.y, a1);
                a0 = fmaf(w[i + 2], hv.z, a0);
                a1 = fmaf(w[i + 3], hv.w, a1);
            }
            red[(kc * B + b) * 32 + r] = a0 + a1;
        }
        __syncthreads();
        if (cell) {
            const long row = (long)t * B + cb;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) s += red[(q * B + cb) * 32 + g * 8 + cu];
                pre[g] = p.gx[row * 4 * H + g * H + cj] + (s + bias[g]);
            }
            const float ai = 1.f / (1.f + expf(-pre[0])), af = 1.f / (1.f + expf(-pre[1])), ag = tanhf(pre[2]),
                        ao = 1.f / (1.f + expf(-pre[3]));
            const float cn = af * c_prev + ai * ag;
            const float hn = ao * tanhf(cn);
            float* ac = p.acts + row * 4 * H + cj;
            ac[0] = ai, ac[H] = af, ac[2 * H] = ag, ac[3 * H] = ao;
            const long e = row * H + cj;
            p.call[e + (long)B * H] = cn;              // call[t + 1]
            p.hall[e + (long)B * H] = hn;              // hall[t + 1]
            if (p.xout) p.xout[e] = p.mask ? (p.mask[e] ? hn * p.mscale : 0.f) : hn;
            c_prev = cn;
        }
        if (t + 1 < T) grid_arrive(p.sync);
    }
}

template <int KC>      // H = 8 KC; a thread holds W_hh[rows of its chunk][unit u] of all four quarters
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void lstm_layer_bwd_kernel(LstmP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = p.B, H = 8 * KC, T = p.T;
    constexpr int QC = KC / 4;          // rows per thread and quarter
    float* gs = sm;                     // [B][H]: one quarter of dG_{t+1}
    float* red = sm + B * H;            // [32 chunks][B][8 units]
    const int tid = threadIdx.x, u = tid & 7, rc = tid >> 3;
    const int j0 = blockIdx.x * LU;
    const unsigned nwg = gridDim.x;
    float w[KC];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < QC; ++i) w[q * QC + i] = p.whh[(long)(q * H + rc * QC + i) * H + j0 + u];
    const bool cell = tid < B * LU;
    const int cb = tid / LU, cu = tid % LU, cj = j0 + cu;
    float dc_next = 0.f;
#pragma unroll 1
    for (int t = T - 1; t >= 0; --t) {
        float dh_rec = 0.f;
        if (t < T - 1) {
            grid_wait(p.sync, (unsigned)(T - 1 - t) * nwg);     // every workgroup has stored its columns of dG_{t+1}
            float part[32];
#pragma unroll
            for (int b = 0; b < 32; ++b) part[b] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                        // (unrolled: w[] must be indexed by constants to stay in registers)
                __syncthreads();                                 // (the previous quarter / the previous step's reduction is consumed)
                for (int i = tid; i < B * H / 4; i += 256) {
                    const int b = i / (H / 4), c4 = i - b * (H / 4);
                    reinterpret_cast<float4*>(gs)[i] =
                        *reinterpret_cast<const float4*>(p.dG + ((long)(t + 1) * B + b) * 4 * H + q * H + c4 * 4);
                }
                __syncthreads();
#pragma unroll
                for (int b = 0; b < 32; ++b) {
                    if (b < B) {
                        const float* gb = gs + b * H + rc * QC;
                        float a = part[b];
#pragma unroll
                        for (int i = 0; i < QC; i += 4) {
                            const float4 gv = *reinterpret_cast<const float4*>(gb + i);
                            a = fmaf(w[q * QC + i], gv.x, a);
                            a = fmaf(w[q * QC + i + 1], gv.y, a);
                            a = fmaf(w[q * QC + i + 2], gv.z, a);
                            a = fmaf(w[q * QC + i + 3], gv.w, a);
                        }
                        part[b] = a;
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < 32; ++b)
                if (b < B) red[(rc * B + b) * 8 + u] = part[b];
            __syncthreads();
            if (cell) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 32; ++q) s += red[(q * B + cb) * 8 + cu];
                dh_rec = s;
            }
        }
        if (cell) {
            const long row = (long)t * B + cb;
            const long e = row * H + cj;
            float dh = p.dx_up ? (p.mask ? (p.mask[e] ? p.dx_up[e] * p.mscale : 0.f) : p.dx_up[e]) : 0.f;
            dh += dh_rec;
            const float* ac = p.acts + row * 4 * H + cj;
            const float ai = ac[0], af = ac[H], ag = ac[2 * H], ao = ac[3 * H];
            const float cc = p.call[e + (long)B * H], cprev = p.call[e];
            const float tc = tanhf(cc);
            const float dc = dc_next + dh * ao * (1.f - tc * tc);
            float* dg = p.dG + row * 4 * H + cj;
            dg[0] = dc * ag * ai * (1.f - ai);
            dg[H] = dc * cprev * af * (1.f - af);
            dg[2 * H] = dc * ai * (1.f - ag * ag);
            dg[3 * H] = dh * tc * ao * (1.f - ao);
            dc_next = dc * af;
        }
        if (t > 0) grid_arrive(p.sync);
    }
}

template <int KC>
int launch_fwd(const LstmP& p, hipStream_t s) {
    const int smem = (p.B * 8 * KC + 8 * p.B * 32) * 4;
    static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_layer_fwd_kernel<KC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (32 * 8 * KC + 8 * 32 * 32) * 4) == hipSuccess ? 0 : MTL_ELAUNCH;
    if (attr) return attr;
    hipLaunchKernelGGL(lstm_layer_fwd_kernel<KC>, dim3(p.H / LU), dim3(256), smem, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}
template <int KC>
int launch_bwd(const LstmP& p, hipStream_t s) {
    const int smem = (p.B * 8 * KC + 32 * p.B * 8) * 4;
    static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_layer_bwd_kernel<KC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (32 * 8 * KC + 32 * 32 * 8) * 4) == hipSuccess ? 0 : MTL_ELAUNCH;
    if (attr) return attr;
    hipLaunchKernelGGL(lstm_layer_bwd_kernel<KC>, dim3(p.H / LU), dim3(256), smem, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

}  // namespace

extern "C" {

int mtl_lstm_layer_supported(int B, int H) { return B >= 1 && B <= 32 && (H == 128 || H == 256 || H == 384 || H == 512); }

long mtl_lstm_layer_workspace(void) { return 256; }

int mtl_lstm_layer_fwd(void* stream, const float* gx, const float* w_hh, const float* b_hh, float* hall, float* call, float* acts,
                       float* xout, const unsigned char* mask, float mscale, int T, int B, int H, void* workspace) {
    if (!gx || !w_hh || !b_hh || !hall || !call || !acts || !workspace || T <= 0 || !mtl_lstm_layer_supported(B, H)) return MTL_EINVAL;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(workspace, 0, 8, s) != hipSuccess) return MTL_ELAUNCH;
    LstmP p{gx, w_hh, b_hh, hall, call, acts, xout, mask, mscale, T, B, H, reinterpret_cast<unsigned*>(workspace), nullptr, nullptr};
    switch (H / 8) {
        case 16: return launch_fwd<16>(p, s);
        case 32: return launch_fwd<32>(p, s);
        case 48: return launch_fwd<48>(p, s);
        default: return launch_fwd<64>(p, s);
    }
}

int mtl_lstm_layer_bwd(void* stream, const float* dx_up, const unsigned char* mask, float mscale, const float* w_hh, const float* acts,
                       const float* call, float* dG, int T, int B, int H, void* workspace) {
    if (!w_hh || !acts || !call || !dG || !workspace || T <= 0 || !mtl_lstm_layer_supported(B, H)) return MTL_EINVAL;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(workspace, 0, 8, s) != hipSuccess) return MTL_ELAUNCH;
    LstmP p{nullptr, w_hh, nullptr, nullptr, const_cast<float*>(call), const_cast<float*>(acts), nullptr, mask, mscale, T, B, H,
            reinterpret_cast<unsigned*>(workspace), dx_up, dG};
    switch (H / 8) {
        case 16: return launch_bwd<16>(p, s);
        case 32: return launch_bwd<32>(p, s);
        case 48: return launch_bwd<48>(p, s);
        default: return launch_bwd<64>(p, s);
    }
}

}  // extern "C"
