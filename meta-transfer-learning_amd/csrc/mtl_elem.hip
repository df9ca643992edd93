// HBM-bound kernels of the meta-transfer hot path for gfx950 (MI355X): wave64 shuffle reductions,
// 16-byte-per-lane coalesced accesses, deterministic two-stage reductions (no float atomics).
//
// Reference ops replaced (file:line in /root/reference):
//   nn.LayerNorm + residual + mask multiply   modules/common_layers.py:131,304; modules/encoder.py:72,101,104;
//                                             modules/decoder.py:314,318,321
//   masked softmax                            modules/common_layers.py:322-327
//   nn.Embedding + positional encoding        modules/decoder.py:96
//   F.cross_entropy + topk(1)                 utils/metrics.py:126; models/asr/transformer.py:146-147
//   Conv2d(1->64)+ReLU                        models/asr/transformer.py:48-49
//   SGD / copy_grad / Adam                    trainer/asr/transient_trainer.py:207,229,255; models/asr/transformer.py:205-240
#include <algorithm>

#include "mtl_common.h"
#include "mtl_h2.h"
#include "../../include/mtl_hip.h"

namespace {

// ------------------------------------------------------------------ flat-vector updates (one fp32 buffer for theta / G / m / v)
__global__ void sgd_theta_prime_kernel(const float4* __restrict__ t0, const float4* __restrict__ g, float4* __restrict__ t1,
                                       float alpha, long n4, const float* t0s, const float* gs, float* t1s, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 a = t0[i], b = g[i];
        t1[i] = make_float4(a.x - alpha * b.x, a.y - alpha * b.y, a.z - alpha * b.z, a.w - alpha * b.w);
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) t1s[i] = t0s[i] - alpha * gs[i];
}

// theta1[t][i] = theta0[i] - alpha * g[t][i] for the `tasks` gradient rows of a stack (theta0 is read once per element)
__global__ void sgd_theta_prime_tasks_kernel(const float4* __restrict__ t0, const float4* __restrict__ g, float4* __restrict__ t1,
                                             float alpha, long n4, int tasks) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 a = t0[i];
        for (int t = 0; t < tasks; ++t) {
            const float4 b = g[t * n4 + i];
            t1[t * n4 + i] = make_float4(a.x - alpha * b.x, a.y - alpha * b.y, a.z - alpha * b.z, a.w - alpha * b.w);
        }
    }
}
// out[i] (+)= sum_t x[t][i], t ascending (fixed order)
__global__ void sum_tasks_kernel(float4* __restrict__ out, const float4* __restrict__ x, long n4, int tasks, long ts4, int accumulate) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = accumulate ? out[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < tasks; ++t) {
            const float4 b = x[t * ts4 + i];
            a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        }
        out[i] = a;
    }
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
    const long n4 = n / 4, stride = (long)gridDim.x * blockDim.x;
    float4* y4 = reinterpret_cast<float4*>(y);
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 u = y4[i];
        const float4 v = x4[i];
        u.x += a * v.x;
        u.y += a * v.y;
        u.z += a * v.z;
        u.w += a * v.w;
        y4[i] = u;
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) y[i] += a * x[i];
}

__global__ void scale_kernel(float* __restrict__ y, float a, const float* a_dev, long n) {
    if (a_dev) a = *a_dev;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] *= a;
}

// torch.optim.Adam (defaults, no amsgrad / weight decay): denom = sqrt(v)/sqrt(bc2) + eps ; theta -= lr/bc1 * m/denom
__global__ void adam_kernel(float* __restrict__ th, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float lr, float b1, float b2, float eps, float bc1, float sqrt_bc2, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    const float step = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = m[i] * b1 + (1.f - b1) * gi;         // m.lerp_(g, 1-b1) == m + (g-m)*(1-b1); see note in DESIGN.md
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        th[i] -= step * (mi / denom);
    }
}

// sum of squares, stage 1: one partial per block; stage 2: single block, fixed order
__global__ void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void sum_final_kernel(const float* __restrict__ part, int np, float* __restrict__ out, int mode, float arg,
                                 const float* __restrict__ inv_dev) {
    // mode 0: out = sum ; mode 1: out = sum/arg (or sum * *inv_dev) ; mode 2: clip coefficient min(1, arg/(sqrt(sum)+1e-6))
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = sh[0] + sh[1] + sh[2] + sh[3];
        if (mode == 1) t = inv_dev ? t * (*inv_dev) : t / arg;
        if (mode == 2) t = fminf(1.f, arg / (sqrtf(t) + 1e-6f));
        *out = t;
    }
}

// out[g] = (sum of part[g * per .. (g + 1) * per)) * inv_dev[g]: the per-task losses of a task-batched pass, one block per task
__global__ void sum_groups_kernel(const float* __restrict__ part, int per, float* __restrict__ out, const float* __restrict__ inv_dev) {
    __shared__ float sh[4];
    const float* q = part + (long)blockIdx.x * per;
    float s = 0.f;
    for (int i = threadIdx.x; i < per; i += blockDim.x) s += q[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (sh[0] + sh[1] + sh[2] + sh[3]) * inv_dev[blockIdx.x];
}

// ------------------------------------------------------------------ dropout keep-masks (Philox4x32-10, counter based)
// keep[i] = 1 with probability 1-p.  The 64-bit seed is read from DEVICE memory so that a captured hipGraph draws fresh
// masks on every replay; `offset` separates the dropout sites of one pass.  (reference sites: modules/decoder.py:96,
// modules/common_layers.py:130,303,328 -- active because the meta loop runs in model.train())
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c.z;
        c = make_uint4((unsigned)(p1 >> 32) ^ c.y ^ k.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k.y, (unsigned)p0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
__global__ void dropout_mask_kernel(uint8_t* __restrict__ keep, long n, unsigned thresh, const long* __restrict__ seed_dev,
                                    unsigned long long offset) {
    const unsigned long long seed = (unsigned long long)*seed_dev;
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
    const long n4 = (n + 3) / 4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
        const unsigned long long ctr = offset + (unsigned long long)q;
        const uint4 r = philox4x32_10(make_uint4((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u), key);
        const long i = q * 4;
        if (i < n) keep[i] = r.x >= thresh;
        if (i + 1 < n) keep[i + 1] = r.y >= thresh;
        if (i + 2 < n) keep[i + 2] = r.z >= thresh;
        if (i + 3 < n) keep[i + 3] = r.w >= thresh;
    }
}

// ------------------------------------------------------------------ LayerNorm (+residual, +positional table, *row keep)
// one wave per row; d is a multiple of 64 up to 1024 (NPL = d/64 values per lane).  For d % 256 == 0 a lane owns NPL/4 quads of four
// consecutive columns (quad j of lane l = columns 4 (64 j + l) ..): every row access is a 16-byte load / store (these launches
// are latency-bound -- 8 dword loads per operand and lane cost more issue slots and address registers than 2 dwordx4); otherwise
// lane-strided single columns.
template <int NPL>
struct LnRow {
    static constexpr bool V4 = NPL % 4 == 0;
    static __device__ __forceinline__ int col(int i, int lane) { return V4 ? ((i >> 2) * 64 + lane) * 4 + (i & 3) : i * 64 + lane; }
    static __device__ __forceinline__ void load(const float* __restrict__ p, int lane, float (&v)[NPL]) {
        if (V4) {
#pragma unroll
            for (int j = 0; j < NPL / 4; ++j) {
                const float4 t = *reinterpret_cast<const float4*>(p + (j * 64 + lane) * 4);
                v[4 * j] = t.x, v[4 * j + 1] = t.y, v[4 * j + 2] = t.z, v[4 * j + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPL; ++i) v[i] = p[i * 64 + lane];
        }
    }
    static __device__ __forceinline__ void load_mask(const uint8_t* __restrict__ p, int lane, bool (&m)[NPL]) {
        if (V4) {
#pragma unroll
            for (int j = 0; j < NPL / 4; ++j) {
                const unsigned t = *reinterpret_cast<const unsigned*>(p + (j * 64 + lane) * 4);
                m[4 * j] = (t & 0xffu) != 0, m[4 * j + 1] = (t & 0xff00u) != 0, m[4 * j + 2] = (t & 0xff0000u) != 0, m[4 * j + 3] = (t >> 24) != 0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPL; ++i) m[i] = p[i * 64 + lane] != 0;
        }
    }
    static __device__ __forceinline__ void store(float* __restrict__ p, int lane, const float (&v)[NPL]) {
        if (V4) {
#pragma unroll
            for (int j = 0; j < NPL / 4; ++j)
                *reinterpret_cast<float4*>(p + (j * 64 + lane) * 4) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < NPL; ++i) p[i * 64 + lane] = v[i];
        }
    }
};

template <int NPL>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ pe, const int* __restrict__ keep,
                                                            const uint8_t* __restrict__ xmask, float xscale,
                                                            float* __restrict__ y, float* __restrict__ xhat,
                                                            float* __restrict__ rstd, int rows, int T, float eps, int rpg, long sParam) {
    constexpr int D = NPL * 64;
    using IO = LnRow<NPL>;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long po = (long)(row / rpg) * sParam;         // parameters of this row's group (task)
    gamma += po;
    beta += po;
    float v[NPL], rv[NPL], gm[NPL], bt[NPL], pv[NPL];
    bool mk[NPL];
    // every load of the row is issued before the first reduction
    IO::load(x + (long)row * D, lane, v);
    if (res) IO::load(res + (long)row * D, lane, rv);
    if (xmask) IO::load_mask(xmask + (long)row * D, lane, mk);
    IO::load(gamma, lane, gm);
    IO::load(beta, lane, bt);
    if (pe) IO::load(pe + (long)(row % T) * D, lane, pv);
    const float kp = keep ? (float)keep[row] : 1.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        float t = v[i];
        if (xmask) t = mk[i] ? t * xscale : 0.f;      // dropout on the sub-layer output, before the residual
        if (res) t += rv[i];
        v[i] = t;
        s += t;
    }
    const float mean = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const float dlt = v[i] - mean;
        q += dlt * dlt;
    }
    const float rs = 1.f / sqrtf(wave_sum(q) * (1.f / D) + eps);
    float h[NPL], o[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        h[i] = (v[i] - mean) * rs;
        float t = h[i] * gm[i] + bt[i];
        if (pe) t += pv[i];
        o[i] = t * kp;
    }
    IO::store(xhat + (long)row * D, lane, h);
    IO::store(y + (long)row * D, lane, o);
    if (lane == 0) rstd[row] = rs;
}

// dz = rstd * (dxh - mean(dxh) - xhat*mean(dxh*xhat)), dxh = dy*keep*gamma ; per-wave partial dgamma/dbeta rows
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const int* __restrict__ keep, const uint8_t* __restrict__ xmask,
                                                            float xscale, float* __restrict__ dz, float* __restrict__ dzm,
                                                            float* __restrict__ dz2, float* __restrict__ part, int rows,
                                                            int rows_per_wave, int rpg, int wpg, long sParam) {
    constexpr int D = NPL * 64;
    using IO = LnRow<NPL>;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int grp = gw / wpg;                           // a wave's rows belong to ONE group (task): wpg waves per group
    gamma += (long)min(grp, (rows - 1) / rpg) * sParam;      // (padding waves beyond the last group own no rows)
    float ag[NPL], ab[NPL], az[NPL], gm[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        ag[i] = 0.f;
        ab[i] = 0.f;
        az[i] = 0.f;
    }
    IO::load(gamma, lane, gm);
    // RB rows at a time with every load of the group issued before the first reduction: the wave pays the HBM latency once per
    // group instead of once per row (it was latency-bound: 22 us for 12 MB)
    constexpr int RB = 4;
    const int r0 = min(grp * rpg + (gw - grp * wpg) * rows_per_wave, rows);
    const int rend = min(min(r0 + rows_per_wave, (grp + 1) * rpg), rows);
    for (int rb = r0; rb < rend; rb += RB) {
        float dv[RB][NPL], hv[RB][NPL], kp[RB], rs[RB];
        bool mk[RB][NPL];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int row = min(rb + q, rows - 1);
            kp[q] = (rb + q < rend) ? (keep ? (float)keep[row] : 1.f) : 0.f;
            rs[q] = rstd[row];
            IO::load(dy + (long)row * D, lane, dv[q]);
            IO::load(xhat + (long)row * D, lane, hv[q]);
            if (xmask) IO::load_mask(xmask + (long)row * D, lane, mk[q]);
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            if (rb + q >= rend) break;
            const int row = rb + q;
            float g[NPL];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const float d = dv[q][i] * kp[q];
                ag[i] += d * hv[q][i];
                ab[i] += d;
                g[i] = d * gm[i];
                s1 += g[i];
                s2 += g[i] * hv[q][i];
            }
            s1 = wave_sum(s1) * (1.f / D);
            s2 = wave_sum(s2) * (1.f / D);
            float o[NPL], om[NPL];
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                o[i] = rs[q] * (g[i] - s1 - hv[q][i] * s2);
                om[i] = o[i];
                if (xmask) om[i] = mk[q][i] ? o[i] * xscale : 0.f;      // gradient of the dropped sub-layer branch (residual branch gets dz)
                az[i] += om[i];
            }
            IO::store(dz + (long)row * D, lane, o);
            if (dz2) IO::store(dz2 + (long)row * D, lane, o);          // second copy: the residual path starts from dz
            if (xmask) IO::store(dzm + (long)row * D, lane, om);
        }
    }
    IO::store(part + ((long)gw * 3) * D, lane, ag);
    IO::store(part + ((long)gw * 3 + 1) * D, lane, ab);
    IO::store(part + ((long)gw * 3 + 2) * D, lane, az);
}
// out[which][c] += sum_w part[w][which][c]  for which = gamma, beta, colsum(dz); one block per (which, 64 columns),
// 4 waves stride the partial rows and combine through LDS in a fixed order
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ part, int nw, int D,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dsum) {
    __shared__ float sh[16][64];
    const int which = blockIdx.y;
    float* out = which == 0 ? dgamma : (which == 1 ? dbeta : dsum);
    if (!out) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f;
    if (c < D) {
        int w = wv;
        for (; w + 16 < nw; w += 32) {
            a0 += part[((long)w * 3 + which) * D + c];
            a1 += part[((long)(w + 16) * 3 + which) * D + c];
        }
        if (w < nw) a0 += part[((long)w * 3 + which) * D + c];
    }
    sh[wv][lane] = a0 + a1;
    __syncthreads();
    if (wv == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][lane];
        out[c] += t;
    }
}

// the same reduction for a whole table of LayerNorm instances (grid.z): ONE launch at the end of a backward pass instead of one
// 5 us launch behind each of its 17 LayerNorm backward kernels
__global__ __launch_bounds__(1024) void ln_param_reduce_batch_kernel(const mtl_ln_reduce_desc* __restrict__ table) {
    __shared__ float sh[16][64];
    const mtl_ln_reduce_desc t = table[blockIdx.z];
    const int which = blockIdx.y, D = t.d, nw = t.nw;
    float* out = which == 0 ? t.dgamma : (which == 1 ? t.dbeta : t.dsum);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    if (!out || blockIdx.x * 64 >= D) return;
    float a0 = 0.f, a1 = 0.f;
    if (c < D) {
        int w = wv;
        for (; w + 16 < nw; w += 32) {
            a0 += t.part[((long)w * 3 + which) * D + c];
            a1 += t.part[((long)(w + 16) * 3 + which) * D + c];
        }
        if (w < nw) a0 += t.part[((long)w * 3 + which) * D + c];
    }
    sh[wv][lane] = a0 + a1;
    __syncthreads();
    if (wv == 0 && c < D) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sh[k][lane];
        out[c] += s;
    }
}

// ------------------------------------------------------------------ masked softmax over keys, one wave per (b,h,q) row
// P = softmax(S*scale) with keys k >= klen[b] (and k > q when causal) filled with -inf.   In place.
__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* __restrict__ S, const int* __restrict__ klen, int causal,
                                                          float scale, int H, int Tq, int Tk, int ld, long rows,
                                                          const uint8_t* __restrict__ pmask, float pscale,
                                                          float* __restrict__ Pd) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int q = (int)(row % Tq);
    const int b = (int)(row / ((long)Tq * H));
    int lim = klen ? min(klen[b], Tk) : Tk;
    if (causal) lim = min(lim, q + 1);
    float* s = S + row * ld;
    float mx = -INFINITY;
    for (int k = lane; k < lim; k += 64) mx = fmaxf(mx, s[k] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < lim; k += 64) sum += expf(s[k] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int k = lane; k < Tk; k += 64) {
        const float p = k < lim ? expf(s[k] * scale - mx) * inv : 0.f;
        s[k] = p;
        if (pmask) Pd[row * ld + k] = pmask[row * ld + k] ? p * pscale : 0.f;     // dropped copy feeds P.V; P itself feeds the bwd
    }
}
// dS = P * (dP - sum_k dP*P) * scale, in place on dP
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, float scale,
                                                          int Tk, int ld, long rows, const uint8_t* __restrict__ pmask,
                                                          float pscale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + row * ld;
    float* d = dP + row * ld;
    const uint8_t* m = pmask ? pmask + row * ld : nullptr;
    float dot = 0.f;
    for (int k = lane; k < Tk; k += 64) {
        if (m) d[k] = m[k] ? d[k] * pscale : 0.f;           // gradient through the dropout on the probabilities
        dot += p[k] * d[k];
    }
    dot = wave_sum(dot);
    for (int k = lane; k < Tk; k += 64) d[k] = p[k] * (d[k] - dot) * scale;
}

// ------------------------------------------------------------------ embedding + positional encoding
__global__ void embed_pe_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ table, const float* __restrict__ pe,
                                    float* __restrict__ out, int rows, int T, int d, const uint8_t* __restrict__ mask, float mscale,
                                    int rpg, long sParam) {
    const long total = (long)rows * d;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / d), c = (int)(e - (long)r * d);
        float v = table[(r / rpg) * sParam + ids[r] * d + c] + pe[(long)(r % T) * d + c];
        if (mask) v = mask[e] ? v * mscale : 0.f;
        out[e] = v;
    }
}
// Scatter-add with duplicate ids, deterministic, ONE launch: `next[r]` (host-built) links row r to the next row with the
// same id (-1 = last) and `first[r]` marks chain heads; the thread of (head row, column) walks its chain in row order.
__global__ void embed_bwd_kernel(const long* __restrict__ ids, const int* __restrict__ first, const int* __restrict__ next,
                                 const float* __restrict__ dout, float* __restrict__ dtable, int rows, int d, long pad_id,
                                 const uint8_t* __restrict__ mask, float mscale, int rpg, long sGrad) {
    const long total = (long)rows * d;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / d), c = (int)(e - (long)r * d);
        const long id = ids[r];
        if (id == pad_id || !first[r]) continue;
        float acc = 0.f;
        for (int q = r; q >= 0; q = next[q]) {
            const float v = dout[(long)q * d + c];
            acc += mask ? (mask[(long)q * d + c] ? v * mscale : 0.f) : v;
        }
        dtable[(r / rpg) * sGrad + id * d + c] += acc;
    }
}

// ------------------------------------------------------------------ cross-entropy (+arg-max), one workgroup (4 waves) per row
// lse = max + log(sum exp(x-max)); rowloss = gold!=pad ? lse - x[gold] : 0 ; hyp = lowest index of the max
// (one WAVE per row walked V = 3765 logits in 59 dependent steps per pass over the row: 30 us for 808 rows)
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const long* __restrict__ gold, int rows,
                                                     int V, int ld, long pad_id, float smoothing, float* __restrict__ lse,
                                                     long* __restrict__ hyp, float* __restrict__ rowloss) {
    __shared__ float smx[4], ssum[4], ssx[4];
    __shared__ int sarg[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = blockIdx.x;
    const float* x = logits + (long)row * ld;
    // The row is read ONCE, as 16-byte quads from the 16-byte boundary at or below its first element (rows of V = 3765 floats start at
    // every alignment), and kept in registers (up to 4 quads per thread: V <= 4093) for the second sweep; quads that straddle the
    // row's end are read element by element (nothing beyond the caller's buffer is touched; elements before the row belong to the
    // previous row of the same buffer).  Longer rows take the two-sweep scalar form.
    const int mis = (int)((reinterpret_cast<uintptr_t>(x) >> 2) & 3);
    const int nq = (V + mis + 3) >> 2;
    const bool fast = nq <= 4 * 256 && !(reinterpret_cast<uintptr_t>(logits) & 3);
    float4 q[4];
    float mx = -INFINITY;
    int arg = 0x7fffffff;
    if (fast) {
        const float* xa = x - mis;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = tid + 256 * i, e0 = 4 * v - mis;       // first element (row index) of quad v
            float4 t = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            if (v < nq) {
                if (e0 >= 0 && e0 + 3 < V) {
                    t = *reinterpret_cast<const float4*>(xa + 4 * v);
                } else {
                    if (e0 >= 0 && e0 < V) t.x = x[e0];
                    if (e0 + 1 >= 0 && e0 + 1 < V) t.y = x[e0 + 1];
                    if (e0 + 2 >= 0 && e0 + 2 < V) t.z = x[e0 + 2];
                    if (e0 + 3 >= 0 && e0 + 3 < V) t.w = x[e0 + 3];
                }
            }
            q[i] = t;
            if (t.x > mx) { mx = t.x; arg = e0; }                // ascending index within the thread: ties keep the lowest
            if (t.y > mx) { mx = t.y; arg = e0 + 1; }
            if (t.z > mx) { mx = t.z; arg = e0 + 2; }
            if (t.w > mx) { mx = t.w; arg = e0 + 3; }
        }
    } else {
        for (int j = tid; j < V; j += 256) {
            const float v = x[j];
            if (v > mx) {
                mx = v;
                arg = j;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (om > mx || (om == mx && oa < arg)) {
            mx = om;
            arg = oa;
        }
    }
    if (lane == 0) {
        smx[wv] = mx;
        sarg[wv] = arg;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float om = smx[w];
        const int oa = sarg[w];
        if (om > mx || (om == mx && oa < arg)) {
            mx = om;
            arg = oa;
        }
    }
    float s = 0.f, sx = 0.f;
    if (fast) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                             // (elements outside the row hold -inf: exp -> 0, excluded from sx)
            const float4 t = q[i];
            s += (expf(t.x - mx) + expf(t.y - mx)) + (expf(t.z - mx) + expf(t.w - mx));
            sx += ((t.x > -INFINITY ? t.x : 0.f) + (t.y > -INFINITY ? t.y : 0.f)) + ((t.z > -INFINITY ? t.z : 0.f) + (t.w > -INFINITY ? t.w : 0.f));
        }
    } else {
        for (int j = tid; j < V; j += 256) {
            const float v = x[j];
            s += expf(v - mx);
            sx += v;
        }
    }
    s = wave_sum(s);
    sx = wave_sum(sx);
    if (lane == 0) {
        ssum[wv] = s;
        ssx[wv] = sx;
    }
    __syncthreads();
    if (tid == 0) {
        s = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        sx = (ssx[0] + ssx[1]) + (ssx[2] + ssx[3]);
        const float l = mx + logf(s);
        lse[row] = l;
        hyp[row] = arg;
        const long g = gold[row];
        float loss = 0.f;
        if (g != pad_id) {
            // label smoothing (utils/metrics.py:113-124): target = (1-eps) one-hot + eps/V elsewhere
            const float nll_gold = l - x[g];
            if (smoothing > 0.f) {
                const float nll_all = l * V - sx;  // sum_j -logp_j
                loss = (1.f - smoothing) * nll_gold + smoothing / V * (nll_all - nll_gold);
            } else {
                loss = nll_gold;
            }
        }
        rowloss[row] = loss;
    }
}
// dlogits = gscale * (softmax - target) on non-pad rows, 0 on pad rows.  gscale = upstream_grad / n_nonpad
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                     const long* __restrict__ gold, int rows, int V, int ld, long pad_id,
                                                     float smoothing, float gscale, const float* gscale_dev,
                                                     float* __restrict__ dlogits, int ldd, int rpg) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (gscale_dev) gscale *= gscale_dev[row / rpg];
    const float* x = logits + (long)row * ld;
    float* d = dlogits + (long)row * ldd;
    const long g = gold[row];
    if (g == pad_id) {
        for (int j = lane; j < V; j += 64) d[j] = 0.f;
        return;
    }
    const float l = lse[row];
    const float off = smoothing > 0.f ? smoothing / V : 0.f;
    const float on = smoothing > 0.f ? 1.f - smoothing : 1.f;
    // the reference's smoothed target (utils/metrics.py:113-118) sums to 1 - eps/V, not 1: d/dx_j = tsum * p_j - t_j
    const float tsum = on + (V - 1) * off;
    for (int j = lane; j < V; j += 64) {
        const float p = expf(x[j] - l);
        d[j] = gscale * (tsum * p - (j == g ? on : off));
    }
}

// ------------------------------------------------------------------ column sums (bias gradients), deterministic two-stage
// stage 1: block (64 columns, row chunk): 4 waves stride the rows of the chunk, lanes = columns (256-B coalesced rows),
// combined through LDS -> part[chunk][c];  stage 2: out[c] += sum_chunk part[chunk][c] in fixed order.
// AMAX: the pass over X also yields max|X| (what a 2-piece fp16 convolution needs of its input): per-block maxima behind the
// partial sums, reduced by the final kernel -- no second read of X.
template <bool AMAX>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, long rows, int cols, long ld,
                                                             long rows_per_block, float* __restrict__ part, float* __restrict__ pmax) {
    __shared__ float sh[4][64];
    __shared__ float shm[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float s0 = 0.f, s1 = 0.f, mx = 0.f;
    if (c < cols) {
        long r = r0 + wv;
        for (; r + 4 < r1; r += 8) {
            const float a = X[r * ld + c], b = X[(r + 4) * ld + c];
            s0 += a;
            s1 += b;
            if (AMAX) mx = fmaxf(mx, fmaxf(fabsf(a), fabsf(b)));
        }
        if (r < r1) {
            const float a = X[r * ld + c];
            s0 += a;
            if (AMAX) mx = fmaxf(mx, fabsf(a));
        }
    }
    sh[wv][lane] = s0 + s1;
    if (AMAX) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) shm[wv] = mx;
    }
    __syncthreads();
    if (wv == 0 && c < cols) part[(long)blockIdx.y * cols + c] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
    if (AMAX && threadIdx.x == 0) pmax[(long)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
}
// The same stage 1 for contiguous rows of 4 C4 columns (C4 a power of two <= 64: the conv bias gradients, 64 / 128 channels): a
// thread owns one 16-byte column quad and every (256 / C4)-th row of the chunk, four rows in flight per thread (the scalar form
// above keeps two 4-byte loads in flight: 1.9 TB/s on the 41 MB conv7 bias sum).  Fixed assignment and order: deterministic.
template <bool AMAX>
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const float4* __restrict__ X, long rows, int C4, long rows_per_block,
                                                                 float* __restrict__ part, float* __restrict__ pmax, long sWs = 0) {
    __shared__ float4 sh[256];
    __shared__ float shm[4];
    if (blockIdx.z) {                                    // task blockIdx.z: its rows behind the previous tasks', its own workspace region
        X += (long)blockIdx.z * rows * C4;
        part += blockIdx.z * sWs;
        pmax += blockIdx.z * sWs;
    }
    const int tid = threadIdx.x, q = tid & (C4 - 1), ph = tid / C4, nph = 256 / C4;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    float mx = 0.f;
    long r = r0 + ph;
    for (; r + 3L * nph < r1; r += 4L * nph) {
        const float4 v0 = X[r * C4 + q], v1 = X[(r + nph) * C4 + q], v2 = X[(r + 2L * nph) * C4 + q], v3 = X[(r + 3L * nph) * C4 + q];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        if (AMAX) {
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))),
                                 fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)))));
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w))),
                                 fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)))));
        }
    }
    for (; r < r1; r += nph) {
        const float4 v0 = X[r * C4 + q];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        if (AMAX) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))));
    }
    sh[tid] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    if (AMAX) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((tid & 63) == 0) shm[tid >> 6] = mx;
    }
    __syncthreads();
    if (tid < C4) {
        float4 t = sh[tid];
        for (int p = 1; p < nph; ++p) {
            const float4 u = sh[p * C4 + tid];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        reinterpret_cast<float4*>(part + (long)blockIdx.y * (4 * C4))[tid] = t;
    }
    if (AMAX && tid == 0) pmax[blockIdx.y] = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
}
// 16 waves per column block: the tall conv-bias sums leave ~1000 partial rows, which 4 waves walked in 126 us
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, int nblk, int cols, float* __restrict__ out,
                                                            const float* __restrict__ pmax, int npmax, float* __restrict__ amax,
                                                            long sWs = 0, long sOut = 0, long sAmax = 0) {
    __shared__ float sh[16][64];
    if (blockIdx.y) {                                    // task blockIdx.y
        part += blockIdx.y * sWs;
        pmax += blockIdx.y * sWs;
        out += blockIdx.y * sOut;
        if (amax) amax += blockIdx.y * sAmax;
    }
    if (amax && blockIdx.x == 0) {             // block 0 also reduces the per-block maxima and WRITES the result (no atomics, no reset)
        __shared__ float shm[16];
        float mx = 0.f;
        for (int i = threadIdx.x; i < npmax; i += 1024) mx = fmaxf(mx, pmax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x < MTL_AMAX_SLOTS) {
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, shm[i]);
            amax[threadIdx.x * MTL_AMAX_STRIDE] = mx;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int b = wv;
        for (; b + 48 < nblk; b += 64) {
            s0 += part[(long)b * cols + c];
            s1 += part[(long)(b + 16) * cols + c];
            s2 += part[(long)(b + 32) * cols + c];
            s3 += part[(long)(b + 48) * cols + c];
        }
        for (; b < nblk; b += 16) s0 += part[(long)b * cols + c];
    }
    sh[wv][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wv == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += sh[w][lane];
        out[c] += t;
    }
}

// ------------------------------------------------------------------ first conv layer (C_in = 1): direct
// x is the reference's (B,1,F,T) tensor (T contiguous); y is (B,T,F,64) channels-last.
// The layer is 2.97 G multiply-adds on the vector ALU next to a 330 MB stream: at one v_fmac_f32 per multiply-add the ALU time
// alone (75 us) exceeds the stream's (52 us at 6.3 TB/s), so the arithmetic runs on v_pk_fma_f32 (two channels per lane and
// instruction: 38 us) and everything that is not arithmetic is kept off the per-pixel path:
//   * a workgroup owns C0_NB tiles of C0_TB = 4 frames x all F bins of one utterance; the (F + 2) x 6 input patch of a tile
//     (5 MB tensor, L2-resident) is staged once in LDS as 8-float rows, zero-padded at the borders -- no bounds logic later;
//   * a thread = (4-channel group cg, bin slot): it fetches rows f-1, f, f+1 of the patch with six ds_read_b128 and produces the
//     4 frames x 4 channels of its bin from registers (144 multiply-adds = 72 v_pk_fma_f32), 27 instructions per pixel and lane
//     instead of ~250 (per-pixel 64-bit index divisions, nine predicated scalar loads);
//   * a wave's four bin slots are consecutive bins: every store instruction of a wave writes 1 KiB contiguous.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int C0_TB = 4, C0_NB = 2, C0_ROW = 8;

// acc += x * w on both halves of w, x = the low (hi = 0) or high (hi = 1) half of the register pair xp: the operand-select bits of
// v_pk_fma_f32 broadcast either half of src0 (hipcc only finds the low-half form and copies the high halves into new pairs)
__device__ __forceinline__ f32x2 pk_fma_bcast(int hi, f32x2 xp, f32x2 w, f32x2 acc) {      // (hi folds after unrolling)
    if (hi)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(xp), "v"(w));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(xp), "v"(w));
    return acc;
}

// the 6 patch values of one row as three register pairs
struct C0Row {
    f32x2 p[3];
    __device__ __forceinline__ void load(const float* r) {
        const float4 lo = *reinterpret_cast<const float4*>(r);
        const float2 hi = *reinterpret_cast<const float2*>(r + 4);
        p[0] = f32x2{lo.x, lo.y}, p[1] = f32x2{lo.z, lo.w}, p[2] = f32x2{hi.x, hi.y};
    }
};

// LDS patch of tile g (utterance b, frames t0 .. t0 + 3): xs[(f + 1) * 8 + c] = x[b][f][t0 - 1 + c], c < 6, zero outside the tensor
__device__ __forceinline__ void conv0_stage(const float* __restrict__ x, float* __restrict__ xs, int g, int ntiles, int ntb, int T,
                                            int F) {
    const int c = threadIdx.x & 7;
    const bool live = g < ntiles && c < 6;
    const int b = live ? g / ntb : 0;
    const int t = live ? (g - b * ntb) * C0_TB - 1 + c : -1;
    const float* xb = x + (long)b * F * T + t;
    for (int fr = threadIdx.x >> 3; fr < F + 2; fr += 32) {
        const int f = fr - 1;
        xs[fr * C0_ROW + c] = (live && (unsigned)t < (unsigned)T && (unsigned)f < (unsigned)F) ? xb[(long)f * T] : 0.f;
    }
}

// blockIdx.y = task of a task-batched pass (round 5): its B samples of x at + task sX floats (0: every task reads the same batch), of y
// at + task B T F 64, its weights / bias / bound at + task sW / sBias / sAmax floats
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B, int T,
                                                        int F, float* __restrict__ amax_y, long sX = 0, long sW = 0, long sBias = 0,
                                                        long sAmax = 0) {
    extern __shared__ __attribute__((aligned(16))) float c0_lds[];          // [C0_NB][F + 2][8]
    const int cg = threadIdx.x & 15, sl = threadIdx.x >> 4;  // channels 4cg..4cg+3, bin slot
    if (blockIdx.y) {
        x += blockIdx.y * sX;
        y += (long)blockIdx.y * B * T * F * 64;
        w += blockIdx.y * sW;
        bias += blockIdx.y * sBias;
        if (amax_y) amax_y += blockIdx.y * sAmax;
    }
    // this workgroup's slot of the bound, read NOW (a stale value only costs a redundant atomic): read at the end, the round trip
    // sat on the tail of every short-lived workgroup (+24 us per launch)
    float* slot = amax_y ? amax_y + (blockIdx.x & (MTL_AMAX_SLOTS - 1)) * MTL_AMAX_STRIDE : nullptr;
    const float seen = slot ? *slot : 0.f;
    f32x2 w01[9], w23[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        w01[k] = f32x2{w[(cg * 4 + 0) * 9 + k], w[(cg * 4 + 1) * 9 + k]};
        w23[k] = f32x2{w[(cg * 4 + 2) * 9 + k], w[(cg * 4 + 3) * 9 + k]};
    }
    const f32x2 b01 = {bias[cg * 4], bias[cg * 4 + 1]}, b23 = {bias[cg * 4 + 2], bias[cg * 4 + 3]};
    const int ntb = (T + C0_TB - 1) / C0_TB, rows = F + 2;
    const int ntiles = B * ntb;
    f32x2 mx = {0.f, 0.f};
    for (int g0 = blockIdx.x * C0_NB; g0 < ntiles; g0 += gridDim.x * C0_NB) {
        __syncthreads();                                     // the previous patches have been consumed
#pragma unroll
        for (int nb = 0; nb < C0_NB; ++nb) conv0_stage(x, c0_lds + nb * rows * C0_ROW, g0 + nb, ntiles, ntb, T, F);
        __syncthreads();
        const int bA = g0 / ntb, tA = (g0 - bA * ntb) * C0_TB;             // the two tiles' (utterance, first frame): uniform
        const int bB = (g0 + 1) / ntb, tB = (g0 + 1 - bB * ntb) * C0_TB;
        const int s_end = (g0 + 1 < ntiles ? 2 : 1) * F;
        for (int s = sl; s < s_end; s += 16) {
            const int nb = s >= F ? 1 : 0, f = s - nb * F;   // (C0_NB == 2)
            const int b = nb ? bB : bA, t0 = nb ? tB : tA;
            const float* r = c0_lds + (nb * rows + f) * C0_ROW;           // patch row of bin f - 1
            C0Row xr[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) xr[kh].load(r + kh * C0_ROW);
            float* yp = y + (((long)b * T + t0) * F + f) * 64 + cg * 4;
#pragma unroll
            for (int tt = 0; tt < C0_TB; ++tt) {
                f32x2 a01 = b01, a23 = b23;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        a01 = pk_fma_bcast((tt + kw) & 1, xr[kh].p[(tt + kw) >> 1], w01[kh * 3 + kw], a01);
                        a23 = pk_fma_bcast((tt + kw) & 1, xr[kh].p[(tt + kw) >> 1], w23[kh * 3 + kw], a23);
                    }
                const f32x2 z = {0.f, 0.f};
                a01 = __builtin_elementwise_max(a01, z);
                a23 = __builtin_elementwise_max(a23, z);
                if (t0 + tt < T) {
                    mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(a01, a23));
                    const f32x4_t out = {a01.x, a01.y, a23.x, a23.y};
                    __builtin_nontemporal_store(out, reinterpret_cast<f32x4_t*>(yp + (long)tt * F * 64));   // streamed: read next by another kernel
                }
            }
        }
    }
    if (amax_y) {       // max of the outputs (>= 0 after the ReLU), one candidate per workgroup; the caller zeroes the slots
        __shared__ float shm[4];
        float m1 = wave_max(fmaxf(mx.x, mx.y));
        if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m1;
        __syncthreads();
        const float cand = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
        if (threadIdx.x == 0 && cand > seen) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(cand));
    }
}
// dw0[c][tap] = sum_pix x[pix+tap]*dy[pix][c], db0[c] = sum_pix dy[pix][c]: per-block partials [blk][64][10].  Same tiling as the
// forward kernel; a thread keeps the 4 channels x (9 taps + bias) sums of its bin slot as v_pk_fma_f32 pairs.
// blockIdx.y = task (see conv0_fwd_kernel): x at + task sX, dy at + task B T F 64, its partial rows behind the previous tasks'
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ part, int B, int T, int F, long sX = 0) {
    extern __shared__ __attribute__((aligned(16))) float c0_lds[];          // [C0_NB][F + 2][8], re-used for the final [16][64][10]
    const int cg = threadIdx.x & 15, sl = threadIdx.x >> 4;
    if (blockIdx.y) {
        x += blockIdx.y * sX;
        dy += (long)blockIdx.y * B * T * F * 64;
        part += (long)blockIdx.y * gridDim.x * 640;
    }
    f32x2 a01[10], a23[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) a01[k] = a23[k] = f32x2{0.f, 0.f};
    const int ntb = (T + C0_TB - 1) / C0_TB, rows = F + 2;
    const int ntiles = B * ntb;
    for (int g0 = blockIdx.x * C0_NB; g0 < ntiles; g0 += gridDim.x * C0_NB) {
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < C0_NB; ++nb) conv0_stage(x, c0_lds + nb * rows * C0_ROW, g0 + nb, ntiles, ntb, T, F);
        __syncthreads();
        const int bA = g0 / ntb, tA = (g0 - bA * ntb) * C0_TB;             // the two tiles' (utterance, first frame): uniform
        const int bB = (g0 + 1) / ntb, tB = (g0 + 1 - bB * ntb) * C0_TB;
        const int s_end = (g0 + 1 < ntiles ? 2 : 1) * F;
        for (int s = sl; s < s_end; s += 16) {
            const int nb = s >= F ? 1 : 0, f = s - nb * F;   // (C0_NB == 2)
            const int b = nb ? bB : bA, t0 = nb ? tB : tA;
            const float* dp = dy + (((long)b * T + t0) * F + f) * 64 + cg * 4;
            f32x4_t d[C0_TB];
#pragma unroll
            for (int tt = 0; tt < C0_TB; ++tt)               // the four 16-byte loads of the stream are issued together
                d[tt] = (t0 + tt < T) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(dp + (long)tt * F * 64))
                                      : f32x4_t{0.f, 0.f, 0.f, 0.f};
            const float* r = c0_lds + (nb * rows + f) * C0_ROW;
            C0Row xr[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) xr[kh].load(r + kh * C0_ROW);
#pragma unroll
            for (int tt = 0; tt < C0_TB; ++tt) {
                const f32x2 d01 = {d[tt].x, d[tt].y}, d23 = {d[tt].z, d[tt].w};
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        a01[kh * 3 + kw] = pk_fma_bcast((tt + kw) & 1, xr[kh].p[(tt + kw) >> 1], d01, a01[kh * 3 + kw]);
                        a23[kh * 3 + kw] = pk_fma_bcast((tt + kw) & 1, xr[kh].p[(tt + kw) >> 1], d23, a23[kh * 3 + kw]);
                    }
                a01[9] += d01;
                a23[9] += d23;
            }
        }
    }
    __syncthreads();
    float* sh = c0_lds;                                      // [16 slots][64 channels][10]
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        sh[(sl * 64 + cg * 4 + 0) * 10 + k] = a01[k].x;
        sh[(sl * 64 + cg * 4 + 1) * 10 + k] = a01[k].y;
        sh[(sl * 64 + cg * 4 + 2) * 10 + k] = a23[k].x;
        sh[(sl * 64 + cg * 4 + 3) * 10 + k] = a23[k].y;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 640; e += 256) {
        float t = 0.f;
        for (int p = 0; p < 16; ++p) t += sh[p * 640 + e];
        part[(long)blockIdx.x * 640 + e] = t;
    }
}
// one wave per output element: lanes stride the per-block partials, fixed-order shuffle tree
__global__ __launch_bounds__(256) void conv0_wgrad_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dw,
                                                                float* __restrict__ db, long sDw = 0, long sDb = 0) {
    if (blockIdx.y) {
        part += (long)blockIdx.y * nblk * 640;
        dw += blockIdx.y * sDw;
        db += blockIdx.y * sDb;
    }
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= 640) return;
    float s = 0.f;
    for (int b = lane; b < nblk; b += 64) s += part[(long)b * 640 + e];
    s = wave_sum(s);
    if (lane == 0) {
        const int c = e / 10, k = e % 10;
        if (k < 9)
            dw[c * 9 + k] += s;
        else
            db[c] += s;
    }
}

// ------------------------------------------------------------------ input_linear weight permutation
// The conv stack emits features as (T', H=F/4, C) per frame; the reference flattens them as c*H + h.
// wp[o][h*C + c] = w[o][c*H + h]   (and the inverse, accumulating, for the gradient)
// One workgroup per row: the (C, Hh) <-> (Hh, C) transpose goes through LDS so that both the read and the write of HBM are
// coalesced (the element-wise form below read 4-byte words 4 Hh bytes apart: 28 us for 10 MB).
// Task-batched (round 4): workgroup (task, row); the per-task launches of the validation pass (8 x 10 MB) were launch-bound.
template <bool VEC>
__global__ __launch_bounds__(256) void permute_hc_lds_kernel(const float* __restrict__ w, float* __restrict__ wp, int rows, int C, int Hh,
                                                             int inverse_accum, float* __restrict__ amax, long sSrc, long sDst, long sAmax) {
    extern __shared__ float tile[];
    const int n = C * Hh;
    const int task = blockIdx.x / rows, row = blockIdx.x - task * rows;
    const float* src = w + task * sSrc + (long)row * n;
    float* dst = wp + task * sDst + (long)row * n;
    if (amax) amax += task * sAmax;
    float mx = 0.f;
    if (VEC) {        // C % 4 == 0 and Hh % 4 == 0: 16-byte HBM accesses on both sides; the tile is kept in (c, h) order with rows of Hh + 1
        const int P = Hh + 1;
        if (inverse_accum) {      // src (h, c) order
            for (int e = threadIdx.x * 4; e < n; e += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(src + e);
                const int h = e / C, c = e - h * C;
                tile[c * P + h] = v.x, tile[(c + 1) * P + h] = v.y, tile[(c + 2) * P + h] = v.z, tile[(c + 3) * P + h] = v.w;
            }
            __syncthreads();
            for (int e = threadIdx.x * 4; e < n; e += 1024) {
                const int c = e / Hh, h = e - c * Hh;
                float4 o = *reinterpret_cast<const float4*>(dst + e);
                o.x += tile[c * P + h], o.y += tile[c * P + h + 1], o.z += tile[c * P + h + 2], o.w += tile[c * P + h + 3];
                *reinterpret_cast<float4*>(dst + e) = o;
            }
        } else {                  // src (c, h) order
            for (int e = threadIdx.x * 4; e < n; e += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(src + e);
                const int c = e / Hh, h = e - c * Hh;
                tile[c * P + h] = v.x, tile[c * P + h + 1] = v.y, tile[c * P + h + 2] = v.z, tile[c * P + h + 3] = v.w;
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
            if (amax) amax_raise(amax, mx);
            __syncthreads();
            for (int e = threadIdx.x * 4; e < n; e += 1024) {
                const int h = e / C, c = e - h * C;
                *reinterpret_cast<float4*>(dst + e) = make_float4(tile[c * P + h], tile[(c + 1) * P + h], tile[(c + 2) * P + h], tile[(c + 3) * P + h]);
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < n; e += 256) {
        const float v = src[e];
        tile[e] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    if (amax) amax_raise(amax, mx);        // max|w| rides along (the bound of the h2 GEMM that multiplies with the permuted weight)
    __syncthreads();
    if (inverse_accum) {      // src is in (h, c) order; dst (reference order, (c, h)) accumulates
        for (int e = threadIdx.x; e < n; e += 256) {
            const int c = e / Hh, h = e - c * Hh;
            dst[e] += tile[h * C + c];
        }
    } else {                  // src is in reference (c, h) order; dst[h][c]
        for (int e = threadIdx.x; e < n; e += 256) {
            const int h = e / C, c = e - h * C;
            dst[e] = tile[c * Hh + h];
        }
    }
}
__global__ void permute_hc_kernel(const float* __restrict__ w, float* __restrict__ wp, int rows, int C, int Hh, int inverse_accum) {
    const long total = (long)rows * C * Hh;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int h = (int)((e / C) % Hh);
        const long o = e / ((long)C * Hh);
        const long ref = o * C * Hh + (long)c * Hh + h;
        if (inverse_accum)
            wp[ref] += w[e];  // w = gradient in (h,c) order, wp = reference-order gradient
        else
            wp[e] = w[ref];
    }
}

// ------------------------------------------------------------------ spectrogram front-end
// reim is the DFT-as-GEMM output (T frames x [F real | F imag], ld); out[f*T + t] = log1p(|X[t][f]|) (the reference's
// (freq, time) layout, utils/data_loader.py:80-88) plus per-block partial (sum, sum of squares) for the normalisation.
__global__ __launch_bounds__(256) void spect_logmag_kernel(const float* __restrict__ reim, int ld, int T, int F,
                                                           float* __restrict__ out, double* __restrict__ part) {
    __shared__ double sh[2][4];
    double s = 0.0, q = 0.0;
    const long total = (long)T * F;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int t = (int)(e / F), f = (int)(e - (long)t * F);
        const float re = reim[(long)t * ld + f], im = reim[(long)t * ld + F + f];
        const float v = log1pf(sqrtf(re * re + im * im));
        out[(long)f * T + t] = v;
        s += v;
        q += (double)v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = s;
        sh[1][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}
// x = (x - mean) / std, std unbiased (torch.Tensor.std), statistics combined in fp64 from the fixed-order partials
__global__ void spect_normalize_kernel(float* __restrict__ x, long n, const double* __restrict__ part, int nb) {
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nb; ++b) {
        s += part[2 * b];
        q += part[2 * b + 1];
    }
    const double mean = s / (double)n;
    const double var = (q - (double)n * mean * mean) / (double)(n - 1);
    const float m = (float)mean, inv = (float)(1.0 / sqrt(var));
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) x[e] = (x[e] - m) * inv;
}

// ------------------------------------------------------------------ LSTM cell (lm/model/rnn_model.py:20 nn.LSTM), one time step
// gates = gx + gh (both B x 4H, biases already added by the two GEMMs), torch order [i | f | g | o]:
//   i, f, o = sigmoid, g = tanh;  c = f * c_prev + i * g;  h = o * tanh(c)   (+ optional dropped copy of h for the next layer)
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gh,
                                                            const float* __restrict__ c_prev, float* __restrict__ acts,
                                                            float* __restrict__ c, float* __restrict__ h, float* __restrict__ h_drop,
                                                            const uint8_t* __restrict__ mask, float mscale, int B, int H) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * H) return;
    const int b = e / H, j = e - b * H;
    const long base = (long)b * 4 * H + j;
    const float pi = gx[base] + gh[base], pf = gx[base + H] + gh[base + H], pg = gx[base + 2 * H] + gh[base + 2 * H],
                po = gx[base + 3 * H] + gh[base + 3 * H];
    const float ai = 1.f / (1.f + expf(-pi)), af = 1.f / (1.f + expf(-pf)), ag = tanhf(pg), ao = 1.f / (1.f + expf(-po));
    const float cn = af * c_prev[e] + ai * ag;
    const float hn = ao * tanhf(cn);
    acts[base] = ai;
    acts[base + H] = af;
    acts[base + 2 * H] = ag;
    acts[base + 3 * H] = ao;
    c[e] = cn;
    h[e] = hn;
    if (h_drop) h_drop[e] = mask ? (mask[e] ? hn * mscale : 0.f) : hn;
}
// dh_total = dh_up [* mask * mscale] + dh_rec ; do = dh_total * tanh(c) ; dc = dc_next + dh_total * o * (1 - tanh(c)^2)
// di = dc * g ; df = dc * c_prev ; dg = dc * i ; dc_prev = dc * f ; pre-activation gradients through sigmoid' / tanh'
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ dh_up, const uint8_t* __restrict__ mask,
                                                            float mscale, const float* __restrict__ dh_rec,
                                                            const float* __restrict__ dc_next, const float* __restrict__ acts,
                                                            const float* __restrict__ c, const float* __restrict__ c_prev,
                                                            float* __restrict__ dgates, float* __restrict__ dc_prev, int B, int H) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * H) return;
    const int b = e / H, j = e - b * H;
    const long base = (long)b * 4 * H + j;
    float dh = dh_up ? (mask ? (mask[e] ? dh_up[e] * mscale : 0.f) : dh_up[e]) : 0.f;
    if (dh_rec) dh += dh_rec[e];
    const float ai = acts[base], af = acts[base + H], ag = acts[base + 2 * H], ao = acts[base + 3 * H];
    const float tc = tanhf(c[e]);
    const float dc = (dc_next ? dc_next[e] : 0.f) + dh * ao * (1.f - tc * tc);
    dgates[base] = dc * ag * ai * (1.f - ai);
    dgates[base + H] = dc * c_prev[e] * af * (1.f - af);
    dgates[base + 2 * H] = dc * ai * (1.f - ag * ag);
    dgates[base + 3 * H] = dh * tc * ao * (1.f - ao);
    dc_prev[e] = dc * af;
}

}  // namespace

extern "C" {

int mtl_lstm_cell_fwd(void* stream, const float* gx, const float* gh, const float* c_prev, float* acts, float* c, float* h,
                      float* h_drop, const unsigned char* mask, float mscale, int B, int H) {
    if (!gx || !gh || !c_prev || !acts || !c || !h || B <= 0 || H <= 0) return MTL_EINVAL;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, as_stream(stream), gx, gh, c_prev, acts, c, h,
                       h_drop, mask, mscale, B, H);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_lstm_cell_bwd(void* stream, const float* dh_up, const unsigned char* mask, float mscale, const float* dh_rec,
                      const float* dc_next, const float* acts, const float* c, const float* c_prev, float* dgates, float* dc_prev,
                      int B, int H) {
    if (!acts || !c || !c_prev || !dgates || !dc_prev || B <= 0 || H <= 0) return MTL_EINVAL;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, as_stream(stream), dh_up, mask, mscale, dh_rec,
                       dc_next, acts, c, c_prev, dgates, dc_prev, B, H);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_spect_logmag(void* stream, const float* reim, int ld, int T, int F, float* out, double* partials, int normalize) {
    if (!reim || !out || !partials || T <= 0 || F <= 0 || ld < 2 * F) return MTL_EINVAL;
    const int nb = grid_for((long)T * F, 256, 128);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(spect_logmag_kernel, dim3(nb), dim3(256), 0, s, reim, ld, T, F, out, partials);
    if (normalize)
        hipLaunchKernelGGL(spect_normalize_kernel, dim3(grid_for((long)T * F, 256, 256)), dim3(256), 0, s, out, (long)T * F, partials, nb);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}


int mtl_sgd_theta_prime(void* stream, const float* theta0, const float* g, float alpha, float* theta1, long n) {
    if (!theta0 || !g || !theta1 || n <= 0) return MTL_EINVAL;
    if ((reinterpret_cast<uintptr_t>(theta0) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(theta1)) & 15)
        return MTL_EINVAL;
    const long n4 = n / 4;
    hipLaunchKernelGGL(sgd_theta_prime_kernel, dim3(grid_for(n4, 256, 2048)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(theta0), reinterpret_cast<const float4*>(g),
                       reinterpret_cast<float4*>(theta1), alpha, n4, theta0, g, theta1, n);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_sgd_theta_prime_tasks(void* stream, const float* theta0, const float* g, float alpha, float* theta1, long n, int tasks) {
    if (!theta0 || !g || !theta1 || n <= 0 || tasks <= 0 || (n & 3)) return MTL_EINVAL;
    if ((reinterpret_cast<uintptr_t>(theta0) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(theta1)) & 15)
        return MTL_EINVAL;
    hipLaunchKernelGGL(sgd_theta_prime_tasks_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(theta0), reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(theta1),
                       alpha, n / 4, tasks);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_sum_tasks_strided(void* stream, float* out, const float* x, long n, int tasks, long task_stride, int accumulate) {
    if (!out || !x || n <= 0 || tasks <= 0 || (n & 3) || (task_stride & 3) || task_stride < n) return MTL_EINVAL;
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x)) & 15) return MTL_EINVAL;
    hipLaunchKernelGGL(sum_tasks_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float4*>(out), reinterpret_cast<const float4*>(x), n / 4, tasks, task_stride / 4, accumulate);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_sum_tasks(void* stream, float* out, const float* x, long n, int tasks, int accumulate) {
    return mtl_sum_tasks_strided(stream, out, x, n, tasks, n, accumulate);
}

int mtl_axpy(void* stream, float* y, const float* x, float a, long n) {
    if (!y || !x || n <= 0) return MTL_EINVAL;
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x)) & 15) return MTL_EINVAL;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, as_stream(stream), y, x, a, n);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_dropout_mask(void* stream, unsigned char* keep, long n, float p, const long* seed_dev, unsigned long long offset) {
    if (!keep || !seed_dev || n <= 0 || !(p >= 0.f && p < 1.f)) return MTL_EINVAL;
    const unsigned thresh = (unsigned)((double)p * 4294967296.0);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((n + 3) / 4, 256, 2048)), dim3(256), 0, as_stream(stream), keep, n, thresh,
                       seed_dev, offset);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_copy_f32(void* stream, float* dst, const float* src, long n) {
    if (!dst || !src || n <= 0) return MTL_EINVAL;
    return hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, as_stream(stream)) == hipSuccess ? MTL_OK : MTL_ELAUNCH;
}

int mtl_scale(void* stream, float* y, float a, const float* a_dev, long n) {
    if (!y || n <= 0) return MTL_EINVAL;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, as_stream(stream), y, a, a_dev, n);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_adam_step(void* stream, float* theta, const float* G, float* m, float* v, int step, float lr, float beta1,
                  float beta2, float eps, long n) {
    if (!theta || !G || !m || !v || n <= 0 || step < 1) return MTL_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, as_stream(stream), theta, G, m, v, lr, beta1,
                       beta2, eps, (float)bc1, (float)sqrt(bc2), n);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

// out (device scalar) = mode 0: sum x^2 ; mode 2: clip coefficient min(1, arg/(||x||+1e-6)) (torch clip_grad_norm_)
int mtl_sumsq(void* stream, const float* x, long n, float* out, float* workspace, int mode, float arg) {
    if (!x || !out || !workspace || n <= 0) return MTL_EINVAL;
    const int nb = grid_for(n, 256 * 8, 1024);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), x, n, workspace);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), workspace, nb, out, mode, arg, (const float*)nullptr);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_layernorm_fwd_g(void* stream, const float* x, const float* residual, const float* gamma, const float* beta,
                        const float* pe, const int* keep, const unsigned char* xmask, float xscale, float* y, float* xhat,
                        float* rstd, int rows, int d, int T, float eps, int rows_per_group, long sParam) {
    if (!x || !gamma || !beta || !y || !xhat || !rstd || rows <= 0 || rows_per_group <= 0) return MTL_EINVAL;
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t s = as_stream(stream);
#define LN_FWD(N)                                                                                                                  \
    hipLaunchKernelGGL(layernorm_fwd_kernel<N>, grid, block, 0, s, x, residual, gamma, beta, pe, keep, xmask, xscale, y, xhat, rstd, \
                       rows, T > 0 ? T : 1, eps, rows_per_group, sParam)
    switch (d) {
        case 64: LN_FWD(1); break;
        case 128: LN_FWD(2); break;
        case 256: LN_FWD(4); break;
        case 512: LN_FWD(8); break;
        case 1024: LN_FWD(16); break;
        default: return MTL_EINVAL;
    }
#undef LN_FWD
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_layernorm_fwd(void* stream, const float* x, const float* residual, const float* gamma, const float* beta,
                      const float* pe, const int* keep, const unsigned char* xmask, float xscale, float* y, float* xhat,
                      float* rstd, int rows, int d, int T, float eps) {
    return mtl_layernorm_fwd_g(stream, x, residual, gamma, beta, pe, keep, xmask, xscale, y, xhat, rstd, rows, d, T, eps,
                               rows > 0 ? rows : 1, 0);
}

long mtl_layernorm_bwd_workspace(int rows, int d) {
    const int waves = ((rows + 3) / 4 + 3) / 4 * 4;
    return (long)waves * 3 * d * 4;
}

static const int kLnRowsPerWave = 4;
int mtl_layernorm_bwd_g_waves(int rows_per_group) { return (rows_per_group + kLnRowsPerWave - 1) / kLnRowsPerWave; }
long mtl_layernorm_bwd_g_workspace(int rows, int d, int rows_per_group) {
    if (rows <= 0 || rows_per_group <= 0) return 0;
    const int groups = (rows + rows_per_group - 1) / rows_per_group;
    const int waves = (groups * mtl_layernorm_bwd_g_waves(rows_per_group) + 3) / 4 * 4;
    return (long)waves * 3 * d * 4;
}

int mtl_layernorm_bwd_g(void* stream, const float* dy, const float* xhat, const float* rstd, const float* gamma,
                        const int* keep, const unsigned char* xmask, float xscale, float* dz, float* dzm, float* dz2, float* dgamma,
                        float* dbeta, float* dsum, float* workspace, int rows, int d, int defer_reduce, int rows_per_group,
                        long sParam, long sGrad) {
    if (!dy || !xhat || !rstd || !gamma || !dz || !dgamma || !dbeta || !workspace || rows <= 0 || (xmask && !dzm) ||
        rows_per_group <= 0)
        return MTL_EINVAL;
    const int rpw = kLnRowsPerWave;
    const int groups = (rows + rows_per_group - 1) / rows_per_group;
    const int wpg = mtl_layernorm_bwd_g_waves(rows_per_group);
    const int waves = (groups * wpg + 3) / 4 * 4;
    dim3 grid(waves / 4), block(256);
    hipStream_t s = as_stream(stream);
#define LN_BWD(N)                                                                                                                 \
    hipLaunchKernelGGL(layernorm_bwd_kernel<N>, grid, block, 0, s, dy, xhat, rstd, gamma, keep, xmask, xscale, dz, dzm, dz2, workspace, \
                       rows, rpw, rows_per_group, wpg, sParam)
    switch (d) {
        case 64: LN_BWD(1); break;
        case 128: LN_BWD(2); break;
        case 256: LN_BWD(4); break;
        case 512: LN_BWD(8); break;
        case 1024: LN_BWD(16); break;
        default: return MTL_EINVAL;
    }
#undef LN_BWD
    if (!defer_reduce) {
        if (groups == 1) {
            hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((d + 63) / 64, 3), dim3(1024), 0, s, workspace, waves, d, dgamma, dbeta, dsum);
        } else {
            for (int g = 0; g < groups; ++g)
                hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((d + 63) / 64, 3), dim3(1024), 0, s, workspace + (long)g * wpg * 3 * d, wpg,
                                   d, dgamma + g * sGrad, dbeta + g * sGrad, dsum ? dsum + g * sGrad : nullptr);
        }
    }
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_layernorm_bwd(void* stream, const float* dy, const float* xhat, const float* rstd, const float* gamma,
                      const int* keep, const unsigned char* xmask, float xscale, float* dz, float* dzm, float* dz2, float* dgamma,
                      float* dbeta, float* dsum, float* workspace, int rows, int d, int defer_reduce) {
    return mtl_layernorm_bwd_g(stream, dy, xhat, rstd, gamma, keep, xmask, xscale, dz, dzm, dz2, dgamma, dbeta, dsum, workspace, rows, d,
                               defer_reduce, rows > 0 ? rows : 1, 0, 0);
}

int mtl_ln_param_reduce_batch(void* stream, const mtl_ln_reduce_desc* table_dev, int n, int dmax) {
    if (!table_dev || n <= 0 || dmax <= 0) return MTL_EINVAL;
    hipLaunchKernelGGL(ln_param_reduce_batch_kernel, dim3((dmax + 63) / 64, 3, n), dim3(1024), 0, as_stream(stream), table_dev);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_softmax_mask_fwd(void* stream, float* S, const int* klen, int causal, float scale, int B, int H, int Tq, int Tk,
                         int ld, const unsigned char* pmask, float pscale, float* P_dropped) {
    if (!S || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || ld < Tk || (pmask && !P_dropped)) return MTL_EINVAL;
    const long rows = (long)B * H * Tq;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), S, klen, causal,
                       scale, H, Tq, Tk, ld, rows, pmask, pscale, P_dropped);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_softmax_bwd(void* stream, const float* P, float* dP, float scale, long rows, int Tk, int ld, const unsigned char* pmask,
                    float pscale) {
    if (!P || !dP || rows <= 0 || Tk <= 0 || ld < Tk) return MTL_EINVAL;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), P, dP, scale, Tk,
                       ld, rows, pmask, pscale);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_embed_pe_fwd_g(void* stream, const long* ids, const float* table, const float* pe, float* out, int rows, int T, int d,
                       const unsigned char* mask, float mscale, int rows_per_group, long sParam) {
    if (!ids || !table || !pe || !out || rows <= 0 || T <= 0 || rows_per_group <= 0) return MTL_EINVAL;
    hipLaunchKernelGGL(embed_pe_fwd_kernel, dim3(grid_for((long)rows * d, 256, 2048)), dim3(256), 0, as_stream(stream), ids,
                       table, pe, out, rows, T, d, mask, mscale, rows_per_group, sParam);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_embed_pe_fwd(void* stream, const long* ids, const float* table, const float* pe, float* out, int rows, int T, int d,
                     const unsigned char* mask, float mscale) {
    return mtl_embed_pe_fwd_g(stream, ids, table, pe, out, rows, T, d, mask, mscale, rows > 0 ? rows : 1, 0);
}

int mtl_embed_bwd_g(void* stream, const long* ids, const int* first, const int* next, const float* dout, float* dtable, int rows,
                    int d, long pad_id, const unsigned char* mask, float mscale, int rows_per_group, long sGrad) {
    if (!ids || !first || !next || !dout || !dtable || rows <= 0 || rows_per_group <= 0) return MTL_EINVAL;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for((long)rows * d, 256, 2048)), dim3(256), 0, as_stream(stream), ids, first,
                       next, dout, dtable, rows, d, pad_id, mask, mscale, rows_per_group, sGrad);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_embed_bwd(void* stream, const long* ids, const int* first, const int* next, const float* dout, float* dtable, int rows,
                  int d, long pad_id, const unsigned char* mask, float mscale) {
    return mtl_embed_bwd_g(stream, ids, first, next, dout, dtable, rows, d, pad_id, mask, mscale, rows > 0 ? rows : 1, 0);
}

int mtl_ce_argmax_fwd(void* stream, const float* logits, const long* gold, int rows, int V, int ld, long pad_id,
                      float smoothing, int n_nonpad, const float* inv_count_dev, float* lse, long* hyp, float* rowloss,
                      float* loss_out) {
    if (!logits || !gold || !lse || !hyp || !rowloss || !loss_out || rows <= 0 || V <= 0 || (n_nonpad <= 0 && !inv_count_dev))
        return MTL_EINVAL;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(rows), dim3(256), 0, s, logits, gold, rows, V, ld, pad_id, smoothing, lse,
                       hyp, rowloss);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, s, rowloss, rows, loss_out, 1, (float)n_nonpad, inv_count_dev);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_ce_argmax_fwd_g(void* stream, const float* logits, const long* gold, int rows, int V, int ld, long pad_id, float smoothing,
                        const float* inv_count_dev, float* lse, long* hyp, float* rowloss, float* loss_out, int rows_per_group) {
    if (!logits || !gold || !lse || !hyp || !rowloss || !loss_out || rows <= 0 || V <= 0 || !inv_count_dev || rows_per_group <= 0 ||
        rows % rows_per_group)
        return MTL_EINVAL;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(rows), dim3(256), 0, s, logits, gold, rows, V, ld, pad_id, smoothing, lse,
                       hyp, rowloss);
    hipLaunchKernelGGL(sum_groups_kernel, dim3(rows / rows_per_group), dim3(256), 0, s, rowloss, rows_per_group, loss_out, inv_count_dev);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_ce_bwd_g(void* stream, const float* logits, const float* lse, const long* gold, int rows, int V, int ld, long pad_id,
                 float smoothing, float gscale, const float* gscale_dev, float* dlogits, int ldd, int rows_per_group) {
    if (!logits || !lse || !gold || !dlogits || rows <= 0 || rows_per_group <= 0) return MTL_EINVAL;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), logits, lse, gold, rows, V, ld,
                       pad_id, smoothing, gscale, gscale_dev, dlogits, ldd, rows_per_group);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_ce_bwd(void* stream, const float* logits, const float* lse, const long* gold, int rows, int V, int ld, long pad_id,
               float smoothing, float gscale, const float* gscale_dev, float* dlogits, int ldd) {
    return mtl_ce_bwd_g(stream, logits, lse, gold, rows, V, ld, pad_id, smoothing, gscale, gscale_dev, dlogits, ldd, rows > 0 ? rows : 1);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ amax, long sX = 0,
                                                     long sAmax = 0) {
    if (blockIdx.y) {                           // tensor blockIdx.y of a strided batch (the tasks of a task-batched pass)
        x += blockIdx.y * sX;
        amax += blockIdx.y * sAmax;
    }
    float mx = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) mx = fmaxf(mx, fabsf(x[(n4 << 2) + threadIdx.x]));
    __shared__ float shm[4];                    // one candidate per workgroup
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x < 64) amax_raise(amax, fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3])));
}

static long colsum_chunks(long rows, int cols) {
    const long colblocks = (cols + 63) / 64;
    long nblk = (rows + 31) / 32;                 // >= 32 rows per block
    const long want = (1024 + colblocks - 1) / colblocks;
    if (nblk > want) nblk = want;
    if (nblk < 1) nblk = 1;
    return nblk;
}
int mtl_absmax_f32(void* stream, const float* x, long n, float* amax) {
    if (!x || !amax || n <= 0) return MTL_EINVAL;
    if (reinterpret_cast<uintptr_t>(x) & 15) return MTL_EINVAL;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n / 4 + 1, 256, 1024)), dim3(256), 0, as_stream(stream), x, n, amax);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_absmax_f32_tb(void* stream, const float* x, long n, float* amax, int tasks, long sX, long sAmax) {
    if (!x || !amax || n <= 0 || tasks < 1 || tasks > 65535 || (sX & 3)) return MTL_EINVAL;
    if (reinterpret_cast<uintptr_t>(x) & 15) return MTL_EINVAL;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n / 4 + 1, 256, 1024), tasks), dim3(256), 0, as_stream(stream), x, n, amax, sX, sAmax);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

// Census of an h2 operand against its bound (csrc/mtl_h2.h): with s = pow2_scale(bound) an element keeps both fp16 pieces' 22 bits
// while |x| s >= 2^-3 (the low piece is a normal fp16); below, it keeps 22 - (its distance below that line) bits.  Counted per tensor:
// [0] non-zero elements, [1] those with |x| s < 2^-3 (fewer than 22 bits), [2] those with |x| s < 2^-9 (fewer than 16 bits).
// Integer atomics: exact and order-independent (one per workgroup and counter).
__global__ __launch_bounds__(256) void h2_census_kernel(const float* __restrict__ x, long n, const float* __restrict__ amax,
                                                        unsigned long long* __restrict__ counts, long sX, long sAmax, long sCounts) {
    x += blockIdx.y * sX;
    amax += blockIdx.y * sAmax;
    counts += blockIdx.y * sCounts;
    const float s = pow2_scale(amax_read(amax));
    const float lim22 = 0.125f / s, lim16 = 0.001953125f / s;          // exact: powers of two
    unsigned nz = 0, b22 = 0, b16 = 0;
    auto see = [&](float v) {
        const float a = fabsf(v);
        nz += a > 0.f;
        b22 += a > 0.f && a < lim22;
        b16 += a > 0.f && a < lim16;
    };
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        see(v.x), see(v.y), see(v.z), see(v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) see(x[(n4 << 2) + threadIdx.x]);
    __shared__ unsigned shm[3][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nz += __shfl_xor(nz, o, 64);
        b22 += __shfl_xor(b22, o, 64);
        b16 += __shfl_xor(b16, o, 64);
    }
    if ((threadIdx.x & 63) == 0) shm[0][threadIdx.x >> 6] = nz, shm[1][threadIdx.x >> 6] = b22, shm[2][threadIdx.x >> 6] = b16;
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned long long t = (unsigned long long)shm[threadIdx.x][0] + shm[threadIdx.x][1] + shm[threadIdx.x][2] + shm[threadIdx.x][3];
        if (t) atomicAdd(counts + threadIdx.x, t);
    }
}

int mtl_h2_census(void* stream, const float* x, long n, const float* amax, unsigned long long* counts, int tasks, long sX, long sAmax,
                  long sCounts) {
    if (!x || !amax || !counts || n <= 0 || tasks < 1 || tasks > 65535 || (sX & 3)) return MTL_EINVAL;
    if (reinterpret_cast<uintptr_t>(x) & 15) return MTL_EINVAL;
    hipLaunchKernelGGL(h2_census_kernel, dim3(grid_for(n / 4 + 1, 1024, 2048), tasks), dim3(256), 0, as_stream(stream), x, n, amax, counts, sX,
                       sAmax, sCounts);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

// Tasks of different widths in one batch padded to the widest: y is (n, T, row) floats, sample s belongs to task s / per_task, and its
// frames [widths[task] >> shift, T) are cleared -- the next 3x3 convolution then sees the zero border the task's own (narrower) image
// ends in, and the ReLU gates of the backward (act > 0) keep every gradient out of those frames.
__global__ void __launch_bounds__(256) zero_tails_kernel(float* __restrict__ y, long T, long row, const int* __restrict__ widths, int shift,
                                                         int per_task) {
    const long s = blockIdx.y;
    long t0 = widths[s / per_task] >> shift;
    t0 = t0 < T ? t0 : T;
    const long n4 = (T - t0) * row / 4;
    float4* p = reinterpret_cast<float4*>(y + (s * T + t0) * row);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int mtl_zero_tails(void* stream, float* y, int n, int T, int row, const int* widths, int shift, int per_task) {
    if (!y || !widths || n < 1 || n > 65535 || T < 1 || row < 4 || (row & 3) || shift < 0 || shift > 8 || per_task < 1) return MTL_EINVAL;
    if (reinterpret_cast<uintptr_t>(y) & 15) return MTL_EINVAL;
    hipLaunchKernelGGL(zero_tails_kernel, dim3(32, n), dim3(256), 0, as_stream(stream), y, (long)T, (long)row, widths, shift, per_task);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

long mtl_colsum_workspace(long rows, int cols) {
    const long chunks = colsum_chunks(rows, cols);
    return chunks * cols * 4 + chunks * ((cols + 63) / 64) * 4;       // partial sums + per-block maxima
}

int mtl_colsum_accum(void* stream, const float* X, long rows, int cols, long ld, float* out, float* workspace, float* amax) {
    if (!X || !out || !workspace || rows <= 0 || cols <= 0) return MTL_EINVAL;
    long nblk = colsum_chunks(rows, cols);
    const long rpb = (rows + nblk - 1) / nblk;
    nblk = (rows + rpb - 1) / rpb;
    hipStream_t s = as_stream(stream);
    const int cb = (cols + 63) / 64;
    float* pmax = workspace + colsum_chunks(rows, cols) * cols;
    const int c4 = cols / 4;
    const bool vec = ld == cols && cols % 4 == 0 && c4 <= 64 && (c4 & (c4 - 1)) == 0 && !(reinterpret_cast<uintptr_t>(X) & 15) &&
                     !(reinterpret_cast<uintptr_t>(workspace) & 15);
    int npmax = (int)(nblk * cb);
    if (vec) {
        npmax = (int)nblk;
        if (amax)
            hipLaunchKernelGGL(colsum_partial_vec_kernel<true>, dim3(1, (unsigned)nblk), dim3(256), 0, s, reinterpret_cast<const float4*>(X), rows, c4, rpb, workspace, pmax);
        else
            hipLaunchKernelGGL(colsum_partial_vec_kernel<false>, dim3(1, (unsigned)nblk), dim3(256), 0, s, reinterpret_cast<const float4*>(X), rows, c4, rpb, workspace, pmax);
    } else if (amax)
        hipLaunchKernelGGL(colsum_partial_kernel<true>, dim3(cb, (unsigned)nblk), dim3(256), 0, s, X, rows, cols, ld, rpb, workspace, pmax);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<false>, dim3(cb, (unsigned)nblk), dim3(256), 0, s, X, rows, cols, ld, rpb, workspace, pmax);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cb), dim3(1024), 0, s, workspace, (int)nblk, cols, out, pmax, npmax, amax);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

/* column sums of `tasks` row blocks (rows x cols each, contiguous, ld = cols) in one launch pair: task k accumulates onto out + k sOut and
 * writes its max|X| to amax + k sAmax; workspace: tasks x mtl_colsum_workspace(rows, cols) bytes.  Needs the 16-byte form (cols = 4 x 2^j <= 256) */
int mtl_colsum_accum_tb(void* stream, const float* X, long rows, int cols, float* out, float* workspace, float* amax, int tasks, long sOut,
                        long sAmax) {
    if (!X || !out || !workspace || rows <= 0 || cols <= 0 || tasks < 1 || tasks > 65535) return MTL_EINVAL;
    const int c4 = cols / 4;
    if (cols % 4 || c4 > 64 || (c4 & (c4 - 1)) || (reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return MTL_EINVAL;
    long nblk = colsum_chunks(rows, cols);
    const long rpb = (rows + nblk - 1) / nblk;
    nblk = (rows + rpb - 1) / rpb;
    hipStream_t s = as_stream(stream);
    const int cb = (cols + 63) / 64;
    const long sWs = (mtl_colsum_workspace(rows, cols) / 4 + 3) / 4 * 4;          // floats per task, 16-byte aligned
    float* pmax = workspace + colsum_chunks(rows, cols) * cols;
    if (amax)
        hipLaunchKernelGGL(colsum_partial_vec_kernel<true>, dim3(1, (unsigned)nblk, tasks), dim3(256), 0, s, reinterpret_cast<const float4*>(X), rows, c4, rpb, workspace, pmax, sWs);
    else
        hipLaunchKernelGGL(colsum_partial_vec_kernel<false>, dim3(1, (unsigned)nblk, tasks), dim3(256), 0, s, reinterpret_cast<const float4*>(X), rows, c4, rpb, workspace, pmax, sWs);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cb, tasks), dim3(1024), 0, s, workspace, (int)nblk, cols, out, pmax, (int)nblk, amax, sWs, sOut, sAmax);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_conv0_relu_fwd(void* stream, const float* x, const float* w, const float* bias, float* y, int B, int T, int F, float* amax_y) {
    if (!x || !w || !bias || !y) return MTL_EINVAL;
    const long lds = (long)C0_NB * (F + 2) * C0_ROW * 4;
    if (B <= 0 || T <= 0 || F <= 0 || lds > 64 * 1024) return MTL_EINVAL;
    const long tiles = (long)B * ((T + C0_TB - 1) / C0_TB);
    hipLaunchKernelGGL(conv0_fwd_kernel, dim3((unsigned)std::min<long>((tiles + C0_NB - 1) / C0_NB, 4096)), dim3(256), (size_t)lds,
                       as_stream(stream), x, w, bias, y, B, T, F, amax_y);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_conv0_relu_fwd_tb(void* stream, const float* x, const float* w, const float* bias, float* y, int B, int T, int F, float* amax_y,
                          int tasks, long sX, long sW, long sBias, long sAmax) {
    if (!x || !w || !bias || !y || tasks < 1 || tasks > 65535) return MTL_EINVAL;
    const long lds = (long)C0_NB * (F + 2) * C0_ROW * 4;
    if (B <= 0 || T <= 0 || F <= 0 || lds > 64 * 1024) return MTL_EINVAL;
    const long tiles = (long)B * ((T + C0_TB - 1) / C0_TB);
    hipLaunchKernelGGL(conv0_fwd_kernel, dim3((unsigned)std::min<long>((tiles + C0_NB - 1) / C0_NB, 4096), tasks), dim3(256), (size_t)lds,
                       as_stream(stream), x, w, bias, y, B, T, F, amax_y, sX, sW, sBias, sAmax);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

long mtl_conv0_wgrad_workspace(void) { return 1024L * 640 * 4; }

int mtl_conv0_wgrad_tb(void* stream, const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int T, int F, int tasks,
                       long sX, long sDw, long sDb) {
    if (!x || !dy || !dw || !db || !workspace || tasks < 1 || tasks > 256) return MTL_EINVAL;
    const long lds = std::max<long>((long)C0_NB * (F + 2) * C0_ROW * 4, 16L * 640 * 4);
    if (B <= 0 || T <= 0 || F <= 0 || lds > 64 * 1024) return MTL_EINVAL;
    const long tiles = (long)B * ((T + C0_TB - 1) / C0_TB);
    const int nb = (int)std::min<long>((tiles + C0_NB - 1) / C0_NB, 1024 / tasks);      // the tasks share the 1024 partial rows of the workspace
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(conv0_wgrad_kernel, dim3(nb, tasks), dim3(256), (size_t)lds, s, x, dy, workspace, B, T, F, sX);
    hipLaunchKernelGGL(conv0_wgrad_final_kernel, dim3(160, tasks), dim3(256), 0, s, workspace, nb, dw, db, sDw, sDb);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_conv0_wgrad(void* stream, const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int T, int F) {
    if (!x || !dy || !dw || !db || !workspace) return MTL_EINVAL;
    const long lds = std::max<long>((long)C0_NB * (F + 2) * C0_ROW * 4, 16L * 640 * 4);
    if (B <= 0 || T <= 0 || F <= 0 || lds > 64 * 1024) return MTL_EINVAL;
    const long tiles = (long)B * ((T + C0_TB - 1) / C0_TB);
    const int nb = (int)std::min<long>((tiles + C0_NB - 1) / C0_NB, 1024);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(conv0_wgrad_kernel, dim3(nb), dim3(256), (size_t)lds, s, x, dy, workspace, B, T, F);
    hipLaunchKernelGGL(conv0_wgrad_final_kernel, dim3(160), dim3(256), 0, s, workspace, nb, dw, db);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_permute_hc_tb(void* stream, const float* src, float* dst, int rows, int C, int Hh, int inverse_accum, float* amax, int tasks,
                      long sSrc, long sDst, long sAmax) {
    if (!src || !dst || rows <= 0 || C <= 0 || Hh <= 0 || tasks <= 0) return MTL_EINVAL;
    const bool vec = C % 4 == 0 && Hh % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && ((sSrc | sDst) & 3) == 0;
    const long lds = (long)C * (Hh + (vec ? 1 : 0)) * 4;
    if (lds > 64 * 1024) {
        for (int t = 0; t < tasks; ++t) {
            const int rc = mtl_permute_hc(stream, src + t * sSrc, dst + t * sDst, rows, C, Hh, inverse_accum, amax ? amax + t * sAmax : nullptr);
            if (rc != MTL_OK) return rc;
        }
        return MTL_OK;
    }
    if (vec) hipLaunchKernelGGL(permute_hc_lds_kernel<true>, dim3(rows * tasks), dim3(256), lds, as_stream(stream), src, dst, rows, C, Hh,
                                inverse_accum, amax, sSrc, sDst, sAmax);
    else hipLaunchKernelGGL(permute_hc_lds_kernel<false>, dim3(rows * tasks), dim3(256), lds, as_stream(stream), src, dst, rows, C, Hh,
                            inverse_accum, amax, sSrc, sDst, sAmax);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

int mtl_permute_hc(void* stream, const float* src, float* dst, int rows, int C, int Hh, int inverse_accum, float* amax) {
    if (!src || !dst) return MTL_EINVAL;
    if ((long)C * (Hh + 1) * 4 <= 64 * 1024) return mtl_permute_hc_tb(stream, src, dst, rows, C, Hh, inverse_accum, amax, 1, 0, 0, 0);
    else {
        if (amax) hipLaunchKernelGGL(absmax_kernel, dim3(grid_for((long)rows * C * Hh / 4 + 1, 256, 1024)), dim3(256), 0, as_stream(stream), src,
                                     (long)rows * C * Hh, amax);
        hipLaunchKernelGGL(permute_hc_kernel, dim3(grid_for((long)rows * C * Hh, 256, 4096)), dim3(256), 0, as_stream(stream), src,
                           dst, rows, C, Hh, inverse_accum);
    }
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

// host-side Levenshtein distance on code points (replaces python-Levenshtein in utils/metrics.py:38-44)
int mtl_levenshtein_u32(const unsigned int* a, int na, const unsigned int* b, int nb) {
    if (na < 0 || nb < 0) return MTL_EINVAL;
    if (na == 0) return nb;
    if (nb == 0) return na;
    int stackbuf[1024];
    int* prev = nb + 1 <= 1024 ? stackbuf : new int[nb + 1];
    for (int j = 0; j <= nb; ++j) prev[j] = j;
    for (int i = 1; i <= na; ++i) {
        int diag = prev[0];
        prev[0] = i;
        for (int j = 1; j <= nb; ++j) {
            const int sub = diag + (a[i - 1] != b[j - 1]);
            diag = prev[j];
            int best = prev[j] + 1;
            if (prev[j - 1] + 1 < best) best = prev[j - 1] + 1;
            if (sub < best) best = sub;
            prev[j] = best;
        }
    }
    const int d = prev[nb];
    if (prev != stackbuf) delete[] prev;
    return d;
}


}  // extern "C"
