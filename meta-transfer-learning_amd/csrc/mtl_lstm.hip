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
//   * a grid-wide hand-off per step: write-through (sc1) stores of the shared payload -> vmcnt(0) -> workgroup barrier -> one lane
//     arrives on a monotonic counter; consumers: one lane polls (relaxed, agent scope), workgroup barrier, sc1 loads.  Every spin
//     is bounded (an error word is set instead of hanging the device); the grid is H / 8 <= 64 workgroups, far below the 256 CUs,
//     so all of them are resident.
// The backward runs the same ownership in reverse time (see lstm_layer_bwd_kernel).  The weight gradients remain three products
// over all T steps (lm.py).  Reductions are fixed-order: bitwise reproducible.
#include "mtl_common.h"
#include "../../include/mtl_hip.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int LU = 8;                        // hidden units per workgroup
constexpr unsigned SPIN_LIMIT = 1u << 22;    // polls before a wait gives up (~seconds): sets the error word
constexpr int HDR = 1024;                    // workspace header in 4-byte words: [0] arrival counter (single-layer kernels), [1] error word,
                                             // [64 + 64 g ..): the per-workgroup step flags of group g of the stack kernels
constexpr int MAXL = MTL_LSTM_MAX_LAYERS;
constexpr long PBUF = 64L * 32 * 512;        // largest [nwg][B][H] partial buffer, floats

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

// Hand-off without cache maintenance (guide, Guideline 16 form R1): the payload that crosses workgroups (h_t, dG_t) is stored
// write-through (relaxed agent-scope atomic stores = `global_store ... sc1`) and read with `sc1` loads (relaxed agent-scope atomic
// loads: served below the reader's L1), so neither an L2 write-back (release fence, 2-6 us with fresh dirty lines) nor an L1
// invalidate (acquire fence, 1.7 us) is paid per step; the flag follows the payload behind `s_waitcnt vmcnt(0)`.
__device__ __forceinline__ void grid_wait(unsigned* sync, unsigned target) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > SPIN_LIMIT) {
                __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void grid_arrive(unsigned* sync) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// B x H floats written by other workgroups -> LDS (same linear layout), 256 threads.  16-byte `sc1` loads issued back to back with ONE
// wait: the relaxed agent-scope atomic loads of HIP (`__hip_atomic_load`) are kept in program order by the compiler -- a wait after every
// load -- which made this copy a chain of 20 memory round trips (5.0 us of a 13.6 us step, measured with s_memrealtime stamps;
// 1.0 us in this form).
__device__ __forceinline__ void stage_shared(float* lds, const float* src, int count, int tid) {
    if (count % 1024 == 0 && count / 1024 <= 16) {
        const int n = count / 1024;
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < n) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(src + 4 * (tid + 256 * k)) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < n) reinterpret_cast<float4*>(lds)[tid + 256 * k] = v[k];
    } else {
        for (int i = tid; i < count / 2; i += 256) {
            float2 v;
            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src + 2 * i) : "memory");
            reinterpret_cast<float2*>(lds)[i] = v;
        }
    }
}
// four `sc1` dword loads in flight, one wait (the gate pre-activations of one cell are H floats apart)
__device__ __forceinline__ void load_shared_x4(const float* q, long stride, float (&v)[4]) {
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[0]) : "v"(q) : "memory");
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[1]) : "v"(q + stride) : "memory");
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[2]) : "v"(q + 2 * stride) : "memory");
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[3]) : "v"(q + 3 * stride) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void store_shared(float* q, float v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int KC>      // H = 8 KC
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void lstm_layer_fwd_kernel(LstmP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = p.B, H = 8 * KC, T = p.T;
    float* hs = sm;                     // [B][H]
    float* red = sm + B * H;            // [16 chunks][B][32 rows]
    // thread (rp = tid & 15, kc = tid >> 4): rows 2 rp, 2 rp + 1 of the 32 gate rows x the k chunk [kc H/16, + H/16): every LDS read of
    // the state feeds two rows (half the LDS traffic of one row x H/8 per thread: the product is LDS-bound)
    constexpr int KH = KC / 2;
    const int tid = threadIdx.x, rp = tid & 15, kc = tid >> 4;
    const int j0 = blockIdx.x * LU;
    const unsigned nwg = gridDim.x;
    float w[2][KH];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 2 * rp + q;
        const float* wrow = p.whh + (long)((r >> 3) * H + j0 + (r & 7)) * H + kc * KH;
#pragma unroll
        for (int i = 0; i < KH; i += 4) {
            const float4 v = *reinterpret_cast<const float4*>(wrow + i);
            w[q][i] = v.x, w[q][i + 1] = v.y, w[q][i + 2] = v.z, w[q][i + 3] = v.w;
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
        float gxv[4] = {0.f, 0.f, 0.f, 0.f};                    // this step's input contributions: requested before the wait
        if (cell) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gxv[g] = p.gx[((long)t * B + cb) * 4 * H + g * H + cj];
        }
        if (t > 0) grid_wait(p.sync, (unsigned)t * nwg);        // every workgroup has published h_t
        const float* src = p.hall + (long)t * B * H;
        stage_shared(hs, src, B * H, tid);
        __syncthreads();
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            const float* hb = hs + b * H + kc * KH;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int i = 0; i < KH; i += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(hb + i);
                a0 = fmaf(w[0][i], hv.x, a0);
                a1 = fmaf(w[1][i], hv.x, a1);
                a0 = fmaf(w[0][i + 1], hv.y, a0);
                a1 = fmaf(w[1][i + 1], hv.y, a1);
                a0 = fmaf(w[0][i + 2], hv.z, a0);
                a1 = fmaf(w[1][i + 2], hv.z, a1);
                a0 = fmaf(w[0][i + 3], hv.w, a0);
                a1 = fmaf(w[1][i + 3], hv.w, a1);
            }
            *reinterpret_cast<float2*>(red + (kc * B + b) * 32 + 2 * rp) = make_float2(a0, a1);
        }
        __syncthreads();
        if (cell) {
            const long row = (long)t * B + cb;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) s += red[(q * B + cb) * 32 + g * 8 + cu];
                pre[g] = gxv[g] + (s + bias[g]);
            }
            const float ai = 1.f / (1.f + expf(-pre[0])), af = 1.f / (1.f + expf(-pre[1])), ag = tanhf(pre[2]),
                        ao = 1.f / (1.f + expf(-pre[3]));
            const float cn = af * c_prev + ai * ag;
            const float hn = ao * tanhf(cn);
            float* ac = p.acts + row * 4 * H + cj;
            ac[0] = ai, ac[H] = af, ac[2 * H] = ag, ac[3 * H] = ao;
            const long e = row * H + cj;
            p.call[e + (long)B * H] = cn;              // call[t + 1]
            store_shared(p.hall + e + (long)B * H, hn);        // hall[t + 1]: read by every workgroup in the next step
            if (p.xout) p.xout[e] = p.mask ? (p.mask[e] ? hn * p.mscale : 0.f) : hn;
            c_prev = cn;
        }
        if (t + 1 < T) grid_arrive(p.sync);
    }
}

// Backward, same ownership as the forward (workgroup w: the 32 gate rows of its 8 units, in registers): the recurrent gradient
// dh_rec_t = dG_{t+1} . W_hh is split by ROWS -- workgroup w multiplies the 32 columns of dG_{t+1} it has just produced itself
// (in LDS) with its 32 rows of W_hh, for all H outputs (thread: 1-2 output units, 32 x 1-2 weights in registers, the dG operand is
// an LDS broadcast) and publishes that partial (B x H, write-through); after the hand-off every workgroup sums the partials of
// its own 8 units over the workgroups in a fixed order.  Per step a workgroup writes B H floats and reads B x 8 x (H / 8) of them
// (the column-split alternative re-reads all of dG_{t+1}, B x 4H, in every workgroup: measured 24 us per step).  Two partial
// buffers alternate with the step parity (a workgroup is at most one hand-off ahead of the slowest).
template <int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void lstm_layer_bwd_kernel(LstmP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = p.B, H = 8 * KC, T = p.T;
    constexpr int JT = KC > 32 ? 2 : 1;           // output units per thread in the partial product
    float* dgs = sm;                              // [B][32]: this workgroup's columns of dG_t (local row = gate * 8 + unit)
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * LU;
    const unsigned nwg = gridDim.x;
    const int jt = tid * JT;                      // first output unit of this thread
    const bool active = jt < H;
    float w[32][JT];
    if (active) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const float* q = p.whh + (long)((r >> 3) * H + j0 + (r & 7)) * H + jt;
#pragma unroll
            for (int c = 0; c < JT; ++c) w[r][c] = q[c];
        }
    }
    float* P = reinterpret_cast<float*>(p.sync) + HDR;        // two buffers of [nwg][B][H] behind the header
    const long pbuf = (long)nwg * B * H;
    const bool cell = tid < B * LU;
    const int cb = tid / LU, cu = tid % LU, cj = j0 + cu;
    float dc_next = 0.f;
#pragma unroll 1
    for (int t = T - 1; t >= 0; --t) {
        float dh_rec = 0.f;
        // this step's own operands do not depend on the other workgroups: requested before the wait
        float dh_up = 0.f, ai = 0.f, af = 0.f, ag = 0.f, ao = 0.f, cc = 0.f, cprev = 0.f;
        if (cell) {
            const long row = (long)t * B + cb;
            const long e = row * H + cj;
            dh_up = p.dx_up ? (p.mask ? (p.mask[e] ? p.dx_up[e] * p.mscale : 0.f) : p.dx_up[e]) : 0.f;
            const float* ac = p.acts + row * 4 * H + cj;
            ai = ac[0], af = ac[H], ag = ac[2 * H], ao = ac[3 * H];
            cc = p.call[e + (long)B * H], cprev = p.call[e];
        }
        if (t < T - 1) {
            grid_wait(p.sync, (unsigned)(T - 1 - t) * nwg);     // every workgroup has published its partial of step t + 1
            if (cell) {
                const float* q = P + ((t + 1) & 1) * pbuf + (long)cb * H + cj;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll 16
                for (unsigned wq = 0; wq < nwg; wq += 2) {      // (32 loads in flight)      // fixed order (nwg = H / 8 is even)
                    s0 += __hip_atomic_load(q + (long)wq * B * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s1 += __hip_atomic_load(q + (long)(wq + 1) * B * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                dh_rec = s0 + s1;
            }
        }
        if (cell) {
            const long row = (long)t * B + cb;
            const float dh = dh_up + dh_rec;
            const float tc = tanhf(cc);
            const float dc = dc_next + dh * ao * (1.f - tc * tc);
            const float di = dc * ag * ai * (1.f - ai), df = dc * cprev * af * (1.f - af), dg_ = dc * ai * (1.f - ag * ag),
                        do_ = dh * tc * ao * (1.f - ao);
            float* dg = p.dG + row * 4 * H + cj;
            dg[0] = di, dg[H] = df, dg[2 * H] = dg_, dg[3 * H] = do_;
            float* l = dgs + cb * 32 + cu;
            l[0] = di, l[8] = df, l[16] = dg_, l[24] = do_;
            dc_next = dc * af;
        }
        if (t > 0) {
            __syncthreads();
            if (active) {
                float* out = P + (t & 1) * pbuf + (long)blockIdx.x * B * H + jt;
#pragma unroll 1
                for (int b = 0; b < B; ++b) {
                    const float* gb = dgs + b * 32;
                    float a[JT];
#pragma unroll
                    for (int c = 0; c < JT; ++c) a[c] = 0.f;
#pragma unroll
                    for (int r = 0; r < 32; r += 4) {
                        const float4 gv = *reinterpret_cast<const float4*>(gb + r);
#pragma unroll
                        for (int c = 0; c < JT; ++c) {
                            a[c] = fmaf(gv.x, w[r][c], a[c]);
                            a[c] = fmaf(gv.y, w[r + 1][c], a[c]);
                            a[c] = fmaf(gv.z, w[r + 2][c], a[c]);
                            a[c] = fmaf(gv.w, w[r + 3][c], a[c]);
                        }
                    }
                    if (JT == 2) {
                        const unsigned long long u = (unsigned long long)__builtin_bit_cast(unsigned, a[0]) |
                                                     ((unsigned long long)__builtin_bit_cast(unsigned, a[JT - 1]) << 32);
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(out + (long)b * H), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        store_shared(out + (long)b * H, a[0]);
                    }
                }
            }
            grid_arrive(p.sync);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Layer stack as ONE wavefront launch per direction: grid = NL x (H / 8) workgroups, workgroup (l, w) owns units [8 w, 8 w + 8) of
// layer l.  Layer l step t needs layer l step t - 1 (its own counter) and layer l - 1 step t (the lower layer's counter), so the
// layers run one step apart instead of one after the other: ~T + NL - 1 hand-offs per pass instead of NL T.
//   forward:  the input contributions W_ih x_t + b_ih of a layer >= 1 are formed per step by a SECOND group of H / 8 workgroups (same
//             ownership, W_ih rows in registers) that depends on the layer below only and therefore runs ahead of its layer: the
//             layer's own step stays as short as layer 0's (inside the layer's workgroups the product cost 8 us of a 20 us step).
//             Layer 0 takes its input contributions from one product over all T steps (gx[0]), as before.
//   backward: layer l >= 1 also multiplies its columns of dG_t with its rows of W_ih: a second row-split partial (the gradient into
//             the lower layer's output) -- computed after the arrival that publishes the recurrent partial, again in the shadow of
//             the hand-off, and acknowledged by the next arrival.  The upper layer is not gated by the lower one and may run ahead,
//             so these partials are buffered for ALL T steps ([NL-1][T][nwg][B][H], p.p2).
struct StackP {
    float* gx[MAXL];
    const float *wih[MAXL], *bih[MAXL], *whh[MAXL], *bhh[MAXL];
    float *hall[MAXL], *call[MAXL], *acts[MAXL], *xout[MAXL], *dG[MAXL];
    const uint8_t* mask[MAXL];
    const float* dx_up;
    float* p2;
    float mscale;
    int T, B, H, NL;
    unsigned* sync;
};

// Hand-off of the stack kernels: every workgroup owns ONE flag word (the number of steps it has published) instead of sharing an
// arrival counter -- 64 read-modify-writes on one address serialise in L2, 64 write-through stores do not; the waiting workgroup's
// first wave reads all flags of a group with one load per poll.
__device__ __forceinline__ void flags_wait(const unsigned* flags, unsigned n, unsigned target, unsigned* err) {
    if (threadIdx.x < 64) {
        unsigned spins = 0;
        for (;;) {
            const unsigned v = threadIdx.x < n ? __hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
            if (__all(v >= target)) break;
            if (++spins > SPIN_LIMIT) {
                if (threadIdx.x == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void flags_arrive(unsigned* flag, unsigned steps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void lstm_stack_fwd_kernel(StackP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = p.B, H = 8 * KC, T = p.T, NL = p.NL;
    float* hs = sm;                     // [B][H]: the operand of this group's product (previous state / the lower layer's output)
    float* red = sm + B * H;            // [16 chunks][B][32 rows]
    constexpr int KH = KC / 2;
    const int tid = threadIdx.x, rp = tid & 15, kc = tid >> 4;
    const unsigned nwg = H / LU;
    // groups 0 .. NL-1: the layers; groups NL .. 2 NL - 2: the input products of layers 1 .. NL-1 (they only depend on the layer
    // below, so they run ahead of their layer instead of lengthening its step)
    const int grp = blockIdx.x / nwg, j0 = (blockIdx.x % nwg) * LU;
    const bool xgroup = grp >= NL;
    const int l = xgroup ? grp - NL + 1 : grp;
    const int wg = blockIdx.x % nwg;
    unsigned* own = p.sync + 64 + 64 * l;                    // step flags of layer l
    unsigned* xin = p.sync + 64 + 64 * (MAXL + l);           // step flags of the input-product group of layer l
    unsigned* low = p.sync + 64 + 64 * (l > 0 ? l - 1 : 0);
    unsigned* err = p.sync + 1;
    f32x2 w[KH];      // {row 2 rp, row 2 rp + 1} at column kc KH + i: one packed FMA (v_pk_fma_f32) per column, the state value broadcast
    {
        const float* wsrc = xgroup ? p.wih[l] : p.whh[l];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = 2 * rp + q;
            const float* wrow = wsrc + (long)((r >> 3) * H + j0 + (r & 7)) * H + kc * KH;
#pragma unroll
            for (int i = 0; i < KH; i += 4) {
                const float4 v = *reinterpret_cast<const float4*>(wrow + i);
                w[i][q] = v.x, w[i + 1][q] = v.y, w[i + 2][q] = v.z, w[i + 3][q] = v.w;
            }
        }
    }
    const bool cell = tid < B * LU;
    const int cb = tid / LU, cu = tid % LU, cj = j0 + cu;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    float c_prev = 0.f;
    float* hall = p.hall[l];
    float* call = p.call[l];
    float* acts = p.acts[l];
    float* xout = p.xout[l];
    float* gx = p.gx[l];
    const uint8_t* mask = p.mask[l];
    if (cell) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[g] = (xgroup ? p.bih[l] : p.bhh[l])[g * H + cj];
        if (!xgroup) c_prev = call[(long)cb * H + cj];
    }
    // the B x H operand at `src` (written by other workgroups: sc1 loads) against this thread's two row chunks; the 16 chunk
    // partials of an output are summed in a fixed order by the cell threads (after the barrier)
    auto product = [&](const float* src) {
        stage_shared(hs, src, B * H, tid);
        __syncthreads();
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            const float* hb = hs + b * H + kc * KH;
            f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};       // two independent chains (one wave per SIMD: nothing else hides the FMA latency)
#pragma unroll
            for (int i = 0; i < KH; i += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(hb + i);
                a0 = __builtin_elementwise_fma(w[i], f32x2{hv.x, hv.x}, a0);
                a1 = __builtin_elementwise_fma(w[i + 1], f32x2{hv.y, hv.y}, a1);
                a0 = __builtin_elementwise_fma(w[i + 2], f32x2{hv.z, hv.z}, a0);
                a1 = __builtin_elementwise_fma(w[i + 3], f32x2{hv.w, hv.w}, a1);
            }
            *reinterpret_cast<float2*>(red + (kc * B + b) * 32 + 2 * rp) = make_float2(a0.x + a1.x, a0.y + a1.y);
        }
        __syncthreads();
    };
    if (xgroup) {
        const float* xlow = p.xout[l - 1];
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            flags_wait(low, nwg, (unsigned)(t + 1), err);      // the layer below has published its output of step t
            product(xlow + (long)t * B * H);
            if (cell) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) s += red[(q * B + cb) * 32 + g * 8 + cu];
                    store_shared(gx + ((long)t * B + cb) * 4 * H + g * H + cj, s + bias[g]);
                }
            }
            flags_arrive(xin + wg, (unsigned)(t + 1));      // (its barrier also orders the reads of `red` before the next step's writes)
        }
        return;
    }
    // input contributions W_ih x_t + b_ih of this thread's cell.  Layer 0: plain loads issued before the step's wait.  Layers >= 1:
    // written by the input-product group during this launch -- the values of step t + 1 are fetched (after a wait on that group,
    // which runs ahead) at the END of step t, in the shadow of the step's own hand-off, so a step has ONE wait on its critical path
    float gxn[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch_gx = [&](int t) {
        if (l > 0) flags_wait(xin, nwg, (unsigned)(t + 1), err);
        if (cell) {
            const float* q = gx + ((long)t * B + cb) * 4 * H + cj;
            if (l > 0) {
                load_shared_x4(q, H, gxn);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) gxn[g] = q[g * H];
            }
        }
    };
    if (l > 0) fetch_gx(0);
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (l == 0) fetch_gx(t);
        float gxv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) gxv[g] = gxn[g];
        if (t > 0) flags_wait(own, nwg, (unsigned)t, err);     // every workgroup of this layer has published h_t
        product(hall + (long)t * B * H);
        if (cell) {
            const long row = (long)t * B + cb;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) s += red[(q * B + cb) * 32 + g * 8 + cu];
                pre[g] = gxv[g] + (s + bias[g]);
            }
            const float ai = 1.f / (1.f + expf(-pre[0])), af = 1.f / (1.f + expf(-pre[1])), ag = tanhf(pre[2]),
                        ao = 1.f / (1.f + expf(-pre[3]));
            const float cn = af * c_prev + ai * ag;
            const float hn = ao * tanhf(cn);
            float* ac = acts + row * 4 * H + cj;
            ac[0] = ai, ac[H] = af, ac[2 * H] = ag, ac[3 * H] = ao;
            const long e = row * H + cj;
            call[e + (long)B * H] = cn;
            store_shared(hall + e + (long)B * H, hn);
            store_shared(xout + e, mask ? (mask[e] ? hn * p.mscale : 0.f) : hn);      // read by the group above
            c_prev = cn;
        }
        flags_arrive(own + wg, (unsigned)(t + 1));      // also after the last step: the group above waits for it
        if (l > 0 && t + 1 < T) fetch_gx(t + 1);
    }
}

template <int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_stack_bwd_kernel(StackP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = p.B, H = 8 * KC, T = p.T, NL = p.NL;
    constexpr int JT = KC > 32 ? 2 : 1;
    float* dgs = sm;                              // [B][32]
    const int tid = threadIdx.x;
    const unsigned nwg = H / LU;
    const int l = blockIdx.x / nwg, wg = blockIdx.x % nwg, j0 = wg * LU;
    unsigned* own = p.sync + 64 + 64 * l;
    unsigned* up = p.sync + 64 + 64 * (l + 1 < NL ? l + 1 : l);
    unsigned* err = p.sync + 1;
    const int jt = tid * JT;
    const bool active = jt < H;
    float w[32][JT], wi[32][JT];
    if (active) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const long ro = (long)((r >> 3) * H + j0 + (r & 7)) * H + jt;
#pragma unroll
            for (int c = 0; c < JT; ++c) {
                w[r][c] = p.whh[l][ro + c];
                wi[r][c] = l > 0 ? p.wih[l][ro + c] : 0.f;
            }
        }
    }
    const long pbuf = (long)nwg * B * H;
    float* P1 = reinterpret_cast<float*>(p.sync) + HDR + (long)l * 2 * PBUF;       // this layer's recurrent partials, two parities
    const float* P2r = p.p2 + (long)l * T * pbuf;                                      // partials from the layer above (l < NL - 1)
    float* P2w = l > 0 ? p.p2 + (long)(l - 1) * T * pbuf + (long)wg * B * H : nullptr; // partials for the layer below
    const bool cell = tid < B * LU;
    const int cb = tid / LU, cu = tid % LU, cj = j0 + cu;
    const float* acts = p.acts[l];
    const float* call = p.call[l];
    float* dG = p.dG[l];
    const uint8_t* mask = p.mask[l];
    const bool top = l == NL - 1;
    float dc_next = 0.f;
    // partial product of this workgroup's 32 columns of dG_t (in dgs) with its 32 rows of a weight matrix, all H outputs
    auto partial = [&](const float (&ww)[32][JT], float* out) {
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            const float* gb = dgs + b * 32;
            if constexpr (JT == 2) {
                // two output units per thread: one packed FMA (v_pk_fma_f32) per weight pair, the dG operand broadcast to both halves;
                // two accumulator pairs keep the dependent chains at half the length
                f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 32; r += 4) {
                    const float4 gv = *reinterpret_cast<const float4*>(gb + r);
                    a0 = __builtin_elementwise_fma(f32x2{ww[r][0], ww[r][1]}, f32x2{gv.x, gv.x}, a0);
                    a1 = __builtin_elementwise_fma(f32x2{ww[r + 1][0], ww[r + 1][1]}, f32x2{gv.y, gv.y}, a1);
                    a0 = __builtin_elementwise_fma(f32x2{ww[r + 2][0], ww[r + 2][1]}, f32x2{gv.z, gv.z}, a0);
                    a1 = __builtin_elementwise_fma(f32x2{ww[r + 3][0], ww[r + 3][1]}, f32x2{gv.w, gv.w}, a1);
                }
                const float lo = a0.x + a1.x, hi = a0.y + a1.y;
                const unsigned long long u = (unsigned long long)__builtin_bit_cast(unsigned, lo) | ((unsigned long long)__builtin_bit_cast(unsigned, hi) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(out + (long)b * H), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int r = 0; r < 32; r += 4) {
                    const float4 gv = *reinterpret_cast<const float4*>(gb + r);
                    a0 = fmaf(gv.x, ww[r][0], a0);
                    a1 = fmaf(gv.y, ww[r + 1][0], a1);
                    a0 = fmaf(gv.z, ww[r + 2][0], a0);
                    a1 = fmaf(gv.w, ww[r + 3][0], a1);
                }
                store_shared(out + (long)b * H, a0 + a1);
            }
        }
    };
    auto gather = [&](const float* q) {            // fixed-order sum of the nwg partials of this thread's (b, unit)
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 32
        for (unsigned wq = 0; wq < nwg; wq += 2) {
            s0 += __hip_atomic_load(q + (long)wq * B * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s1 += __hip_atomic_load(q + (long)(wq + 1) * B * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return s0 + s1;
    };
    // gradient into this layer's output from the layer above: the sum of its input-gradient partials of step t.  The layer above
    // acknowledges the partial of step t with the arrival of step t - 1 (the final one for t <= 1).
    float dh_up_next = 0.f;
    auto gather_up = [&](int t) {
        const int k = T - t + 1 < T ? T - t + 1 : T;
        flags_wait(up, nwg, (unsigned)k, err);
        if (cell) dh_up_next = gather(P2r + (long)t * pbuf + (long)cb * H + cj);
    };
    if (!top) gather_up(T - 1);
#pragma unroll 1
    for (int t = T - 1; t >= 0; --t) {
        float dh_rec = 0.f, dh_up = 0.f, ai = 0.f, af = 0.f, ag = 0.f, ao = 0.f, cc = 0.f, cprev = 0.f, keep = 1.f;
        if (cell) {
            const long row = (long)t * B + cb;
            const long e = row * H + cj;
            keep = mask ? (mask[e] ? p.mscale : 0.f) : 1.f;
            if (top) dh_up = p.dx_up[e];
            const float* ac = acts + row * 4 * H + cj;
            ai = ac[0], af = ac[H], ag = ac[2 * H], ao = ac[3 * H];
            cc = call[e + (long)B * H], cprev = call[e];
        }
        if (!top) dh_up = dh_up_next;        // gathered at the end of the previous iteration (below), off this step's critical path
        dh_up *= keep;
        if (t < T - 1) {
            flags_wait(own, nwg, (unsigned)(T - 1 - t), err);
            if (cell) dh_rec = gather(P1 + ((t + 1) & 1) * pbuf + (long)cb * H + cj);
        }
        if (cell) {
            const long row = (long)t * B + cb;
            const float dh = dh_up + dh_rec;
            const float tc = tanhf(cc);
            const float dc = dc_next + dh * ao * (1.f - tc * tc);
            const float di = dc * ag * ai * (1.f - ai), df = dc * cprev * af * (1.f - af), dg_ = dc * ai * (1.f - ag * ag),
                        do_ = dh * tc * ao * (1.f - ao);
            float* dg = dG + row * 4 * H + cj;
            dg[0] = di, dg[H] = df, dg[2 * H] = dg_, dg[3 * H] = do_;
            float* q = dgs + cb * 32 + cu;
            q[0] = di, q[8] = df, q[16] = dg_, q[24] = do_;
            dc_next = dc * af;
        }
        __syncthreads();
        if (t > 0) {
            if (active) partial(w, P1 + (t & 1) * pbuf + (long)wg * B * H + jt);
            flags_arrive(own + wg, (unsigned)(T - t));
        }
        if (l > 0 && active) partial(wi, P2w + (long)t * pbuf + jt);     // in the shadow of the hand-off; acknowledged by the next arrival
        if (!top && t > 0) gather_up(t - 1);                             // likewise: the next step's input gradient
    }
    if (l > 0) flags_arrive(own + wg, (unsigned)T);
}

template <int KC>
int launch_fwd(const LstmP& p, hipStream_t s) {
    const int smem = (p.B * 8 * KC + 16 * p.B * 32) * 4;
    static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_layer_fwd_kernel<KC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (32 * 8 * KC + 16 * 32 * 32) * 4) == hipSuccess ? 0 : MTL_ELAUNCH;
    if (attr) return attr;
    hipLaunchKernelGGL(lstm_layer_fwd_kernel<KC>, dim3(p.H / LU), dim3(256), smem, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}
template <int KC>
int launch_bwd(const LstmP& p, hipStream_t s) {
    const int smem = p.B * 32 * 4;
    hipLaunchKernelGGL(lstm_layer_bwd_kernel<KC>, dim3(p.H / LU), dim3(256), smem, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

template <int KC>
int launch_stack_fwd(const StackP& p, hipStream_t s) {
    const int smem = (p.B * 8 * KC + 16 * p.B * 32) * 4;
    static int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_stack_fwd_kernel<KC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (32 * 8 * KC + 16 * 32 * 32) * 4) == hipSuccess ? 0 : MTL_ELAUNCH;
    if (attr) return attr;
    hipLaunchKernelGGL(lstm_stack_fwd_kernel<KC>, dim3((2 * p.NL - 1) * (p.H / LU)), dim3(256), smem, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}
template <int KC>
int launch_stack_bwd(const StackP& p, hipStream_t s) {
    hipLaunchKernelGGL(lstm_stack_bwd_kernel<KC>, dim3(p.NL * (p.H / LU)), dim3(256), p.B * 32 * 4, s, p);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

bool fill_stack(StackP& p, const mtl_lstm_stack* d, int NL, bool bwd) {
    for (int l = 0; l < NL; ++l) {
        p.wih[l] = d->w_ih[l], p.bih[l] = d->b_ih[l], p.whh[l] = d->w_hh[l], p.bhh[l] = d->b_hh[l];
        p.hall[l] = d->hall[l], p.call[l] = d->call[l], p.acts[l] = d->acts[l], p.xout[l] = d->xout[l], p.dG[l] = d->dG[l];
        p.mask[l] = d->mask[l], p.gx[l] = d->gx[l];
        if (!p.whh[l] || !p.call[l] || !p.acts[l] || (l > 0 && !p.wih[l])) return false;
        if (bwd ? !p.dG[l] : (!p.bhh[l] || !p.hall[l] || !p.xout[l] || !p.gx[l] || (l > 0 && !p.bih[l]))) return false;
    }
    return true;
}

}  // namespace

extern "C" {

// Every workgroup of a persistent launch must be resident at once (the grid-wide hand-offs spin on each other): the bound is the
// device's own CU count (hipDeviceAttributeMultiprocessorCount: a partitioned / shared device reports its share), minus a margin
// of one eighth for whatever else holds CUs.
static int lstm_resident_limit() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;                          // no device visible (shape queries on a build host): the MI355X's own count
    return n - n / 8;
}
int mtl_lstm_layer_supported(int B, int H) {
    return B >= 1 && B <= 32 && (H == 128 || H == 256 || H == 384 || H == 512) && H / LU <= lstm_resident_limit();
}

long mtl_lstm_layer_workspace(void) { return HDR * 4 + MAXL * 2 * PBUF * 4; }      // header + two partial buffers per layer (backward)

int mtl_lstm_stack_supported(int B, int H, int NL) {      // every workgroup must be resident: (2 NL - 1) H / 8 of the device's CUs
    return mtl_lstm_layer_supported(B, H) && NL >= 1 && NL <= MAXL && (2 * NL - 1) * (H / LU) <= lstm_resident_limit();
}

long mtl_lstm_stack_scratch(int T, int B, int H, int NL) { return NL > 1 ? (long)(NL - 1) * T * (H / LU) * B * H * 4 : 0; }

int mtl_lstm_stack_fwd(void* stream, const mtl_lstm_stack* layers, float mscale, int T, int B, int H, int NL, void* workspace) {
    if (!layers || !workspace || T <= 0 || !mtl_lstm_stack_supported(B, H, NL)) return MTL_EINVAL;
    StackP p{};
    if (!fill_stack(p, layers, NL, false)) return MTL_EINVAL;
    p.mscale = mscale, p.T = T, p.B = B, p.H = H, p.NL = NL, p.sync = reinterpret_cast<unsigned*>(workspace);
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(static_cast<char*>(workspace) + 256, 0, (HDR - 64) * 4, s) != hipSuccess) return MTL_ELAUNCH;      // the step flags; the error word [1] is sticky
    switch (H / 8) {
        case 16: return launch_stack_fwd<16>(p, s);
        case 32: return launch_stack_fwd<32>(p, s);
        case 48: return launch_stack_fwd<48>(p, s);
        default: return launch_stack_fwd<64>(p, s);
    }
}

int mtl_lstm_stack_bwd(void* stream, const mtl_lstm_stack* layers, const float* dx_up, float mscale, float* scratch, int T, int B, int H,
                       int NL, void* workspace) {
    if (!layers || !dx_up || !workspace || T <= 0 || !mtl_lstm_stack_supported(B, H, NL) || (NL > 1 && !scratch)) return MTL_EINVAL;
    StackP p{};
    if (!fill_stack(p, layers, NL, true)) return MTL_EINVAL;
    p.dx_up = dx_up, p.p2 = scratch, p.mscale = mscale, p.T = T, p.B = B, p.H = H, p.NL = NL, p.sync = reinterpret_cast<unsigned*>(workspace);
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(static_cast<char*>(workspace) + 256, 0, (HDR - 64) * 4, s) != hipSuccess) return MTL_ELAUNCH;      // the step flags; the error word [1] is sticky
    switch (H / 8) {
        case 16: return launch_stack_bwd<16>(p, s);
        case 32: return launch_stack_bwd<32>(p, s);
        case 48: return launch_stack_bwd<48>(p, s);
        default: return launch_stack_bwd<64>(p, s);
    }
}

int mtl_lstm_layer_fwd(void* stream, const float* gx, const float* w_hh, const float* b_hh, float* hall, float* call, float* acts,
                       float* xout, const unsigned char* mask, float mscale, int T, int B, int H, void* workspace) {
    if (!gx || !w_hh || !b_hh || !hall || !call || !acts || !workspace || T <= 0 || !mtl_lstm_layer_supported(B, H)) return MTL_EINVAL;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(workspace, 0, 4, s) != hipSuccess) return MTL_ELAUNCH;      // the arrival counter; the error word [1] is sticky
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
    if (hipMemsetAsync(workspace, 0, 4, s) != hipSuccess) return MTL_ELAUNCH;      // the arrival counter; the error word [1] is sticky
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
