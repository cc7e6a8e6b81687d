// pie_fused.hip -- row A2-head, one pass per direction: the PIENet attention pooling around its w_1 GEMM.
//
// Reference: src/networks/models/pie_model.py:28-40 (MultiHeadSelfAttention.forward, n_head = 1: attn = softmax_P(w_2 .
//            tanh(w_1 x)), masked positions -> -inf, output = attn^T x), image_encoder.py:54-57 (the mean pooling of the
//            same feature map that feeds fc).
//
// pie.hip holds the first version of this row: scores / pool / three backward kernels, fp32 operands only (under bf16
// autocast the caller converted X and H to fp32 first: 0.4 GB of extra traffic per step at ResNet-101, batch 256).
// Here X [N,P,Cd] and H = w_1 X [N,P,dh] are read ONCE per direction in the dtype the trunk / the GEMM produced
// (fp32 or bf16; the arithmetic is fp32 either way) and dX / dH are written once in that dtype:
//   forward : one 1024-thread workgroup per sample (x column slices when N is small): scores from H (tanh on the
//             exp/rcp units), softmax over P in LDS, attention + mean pooling of X.     bytes = N P (Cd + dh) e
//   backward: one workgroup per sample: pass over X (da_p = <d_pooled, X_p>, dX written in the same pass), softmax
//             backward in LDS, pass over H (dH written, dw_2 partial per sample).        bytes = 2 N P (Cd + dh) e
// At N = 256 there is one workgroup per CU and every thread keeps 6-12 16-byte loads in flight (stream2).
// Shapes outside cfl_pie_fused_supported (dh % 4 != 0: the GRU text head with dh = 150) stay on pie.hip.
#include "common.h"
#include "colmap.h"

namespace {

typedef unsigned short u16;

template <typename T> struct PV;
template <> struct PV<float> {
    static constexpr int V = 4;
    typedef f32x4 Raw;
    template <bool NT> static __device__ __forceinline__ Raw ldraw(const float* p) {
        if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        return *reinterpret_cast<const f32x4*>(p);
    }
    static __device__ __forceinline__ void unpack(const Raw& x, float (&v)[4]) { v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3]; }
    static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
        f32x4 x = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = x;
    }
};
template <> struct PV<u16> {
    static constexpr int V = 8;
    typedef U4 Raw;
    template <bool NT> static __device__ __forceinline__ Raw ldraw(const u16* p) {
        if (NT) return ld_nt(reinterpret_cast<const U4*>(p));
        return *reinterpret_cast<const U4*>(p);
    }
    static __device__ __forceinline__ void unpack(const Raw& x, float (&v)[8]) { unpack8(x, v); }
    static __device__ __forceinline__ void st(u16* p, const float (&v)[8]) { *reinterpret_cast<U4*>(p) = pack8(v); }
};

// Streaming loop with two register sets of U 16-byte loads per thread: set B is requested before set A is consumed
// and vice versa, so a thread always has U..2U loads in flight (one workgroup per CU: nothing else hides the HBM
// latency).  addr(i) -> pointer of item i, use(i, raw) consumes it; items past cnt re-read the last one (unconditional
// loads keep the compiler's vmcnt accounting exact) and are not consumed.  NT: nontemporal loads.  Measured at ResNet-101 / batch 256
// (us, forward / backward): bf16 operands (77 MB, the bench regime) default 24-27 / 34, nt in both 27 / 41, nt in the
// backward only 26 / 39 -- the forward leaves X and H in the Infinity Cache for the backward and nt loads forfeit that;
// fp32 operands (154 MB) default 50 / 57, nt forward 36 / 59.  So: nt only for the fp32 forward.
template <typename T, int U, bool NT, typename AddrFn, typename UseFn>
__device__ __forceinline__ void stream2(int cnt, AddrFn addr, UseFn use) {
    if (cnt <= 0) return;
    typename PV<T>::Raw a[U], b[U];
    const int last = cnt - 1;
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = PV<T>::template ldraw<NT>(addr(u < last ? u : last));
    for (int base = 0; base < cnt; base += 2 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = base + U + u; b[u] = PV<T>::template ldraw<NT>(addr(i < last ? i : last)); }
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u < cnt) use(base + u, a[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = base + 2 * U + u; a[u] = PV<T>::template ldraw<NT>(addr(i < last ? i : last)); }
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + U + u < cnt) use(base + U + u, b[u]);
    }
}
constexpr int PIE_U = 6;

// V consecutive floats from LDS / global fp32 memory (16-byte aligned)
template <int V>
__device__ __forceinline__ void ldf(const float* p, float (&v)[V]) {
#pragma unroll
    for (int i = 0; i < V; i += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p + i);
        v[i] = x[0]; v[i + 1] = x[1]; v[i + 2] = x[2]; v[i + 3] = x[3];
    }
}
template <int V>
__device__ __forceinline__ void stf(float* p, const float (&v)[V]) {
#pragma unroll
    for (int i = 0; i < V; i += 4) {
        f32x4 x = {v[i], v[i + 1], v[i + 2], v[i + 3]};
        *reinterpret_cast<f32x4*>(p + i) = x;
    }
}

// tanh x = 1 - 2 / (1 + e^{2x}) on the transcendental unit (v_exp_f32 + v_rcp_f32, ~1 ulp each: absolute error
// <= 3e-7; saturates to +-1 through e = inf / 0).  tanhf() is ~30 VALU instructions per element and made the
// scores pass VALU-bound (12.8 M tanh at ResNet-101 batch 256).
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
}

__device__ __forceinline__ float block_sum_1024(float v, float* red) {      // red[16]; every thread gets the sum
    v = wave_sum(v);
    __syncthreads();                                                         // red may still be read by a previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max_1024(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s = fmaxf(s, red[i]);
    return s;
}

// Column reduction over the G row groups of the workgroup: part[g][QX*V] -> out[t*V ..] for t < QX (fixed order).
template <int V>
__device__ __forceinline__ void group_reduce(const float (&acc)[V], bool active, int g, int q, int G, int QX, float scale,
                                             float* part, float* out) {
    if (active) stf<V>(part + ((long long)g * QX + q) * V, acc);
    __syncthreads();
    if ((int)threadIdx.x < QX) {
        float r[V];
#pragma unroll
        for (int e = 0; e < V; ++e) r[e] = 0.f;
        for (int gg = 0; gg < G; ++gg) {
            float v[V];
            ldf<V>(part + ((long long)gg * QX + threadIdx.x) * V, v);
#pragma unroll
            for (int e = 0; e < V; ++e) r[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) r[e] *= scale;
        stf<V>(out + (long long)threadIdx.x * V, r);
    }
    __syncthreads();
}

constexpr int PIE_PART = 8192;          // floats of LDS reduction scratch
constexpr int PIE_DH_MAX = 4096;       // w_2 staged in LDS by the forward
constexpr int PIE_SEGP = 4096;          // (row, 64-lane segment) partial dots of the backward

// ---- forward ------------------------------------------------------------------------------------------------------
// grid (N, S): workgroup (n, s) recomputes the P scores of sample n (H[n] is re-read from L2 by the S slices) and pools
// the columns [s Cd/S, (s+1) Cd/S).
template <typename T>
__global__ __launch_bounds__(1024) void cfl_pie_fwd_fused_kernel(const T* __restrict__ X, const T* __restrict__ H,
                                                                 const float* __restrict__ w2, const unsigned char* __restrict__ mask,
                                                                 int P, int Cd, int dh, int S, float* __restrict__ attn,
                                                                 float* __restrict__ pooled, float* __restrict__ xmean) {
    constexpr int V = PV<T>::V;
    constexpr bool FWD_NT = sizeof(T) == 4;              // see stream2
    __shared__ float sa[1024];
    __shared__ __attribute__((aligned(16))) float part[PIE_PART];
    __shared__ float red[16];
    __shared__ __attribute__((aligned(16))) float sw2[PIE_DH_MAX];
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;

    // w_2 lives in LDS: a global load inside the consuming code would put a vmcnt(0) -- a wait for EVERY prefetched
    // load of the wave -- in front of each task
    for (int c = t * 4; c < dh; c += 4096) {
        float v4[4];
        ldf<4>(w2 + c, v4);
        stf<4>(sw2 + c, v4);
    }
    __syncthreads();

    // 1. scores: a task = 64 lanes x V columns of one row; a wave streams its tasks through two register sets
    constexpr int SEGW = 64 * V;
    const int nseg = (dh + SEGW - 1) / SEGW, ntask = P * nseg;
    const T* h = H + (long long)n * P * dh;
    stream2<T, PIE_U, FWD_NT>((ntask - w + 15) / 16,
        [&](int i) {
            const int k = w + 16 * i, p = k / nseg;
            int j = (k - p * nseg) * SEGW + lane * V;
            j = j < dh ? j : dh - V;
            return h + (long long)p * dh + j;
        },
        [&](int i, const typename PV<T>::Raw& raw) {
            const int k = w + 16 * i, p = k / nseg;
            const int j = (k - p * nseg) * SEGW + lane * V;
            float s = 0.f;
            if (j < dh) {
                float v[V], wv[V];
                PV<T>::unpack(raw, v);
                ldf<V>(sw2 + j, wv);
#pragma unroll
                for (int e = 0; e < V; ++e) s = fmaf(wv[e], tanh_fast(v[e]), s);
            }
            s = wave_sum(s);
            if (lane == 0) part[k] = s;
        });
    __syncthreads();

    // 2. softmax over the P positions (thread p)
    float sc = -INFINITY;
    if (t < P) {
        float s = 0.f;
        for (int sg = 0; sg < nseg; ++sg) s += part[t * nseg + sg];
        sc = (mask && mask[(long long)n * P + t]) ? -INFINITY : s;
    }
    const float mx = block_max_1024(sc, red);
    const float ex = t < P ? expf(sc - mx) : 0.f;
    const float inv = 1.f / block_sum_1024(ex, red);
    if (t < P) {
        const float a = ex * inv;
        sa[t] = a;
        if (blockIdx.y == 0) attn[(long long)n * P + t] = a;
    }
    __syncthreads();

    // 3. attention pooling + mean pooling of this workgroup's column slice
    const int CS = Cd / S, c_base = blockIdx.y * CS, QX = CS / V;
    const T* x = X + (long long)n * P * Cd + c_base;
    float* po = pooled + (long long)n * Cd + c_base;
    float* mo = xmean ? xmean + (long long)n * Cd + c_base : nullptr;
    const float invP = 1.f / (float)P;
    const int G = QX >= 1024 ? 1 : (1024 / QX < P ? 1024 / QX : P);
    const int QE = QX >= 1024 ? 1024 : QX;                              // threads per row group
    const int g = t / QE, q0 = t - g * QE;
    const bool active = g < G;
    for (int qb = 0; qb < QX; qb += 1024) {                            // one trip unless a row has > 1024 vectors
        const int q = qb + q0;
        float acc[V], mean[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { acc[e] = 0.f; mean[e] = 0.f; }
        const bool act = active && q < QX;
        stream2<T, PIE_U, FWD_NT>(act ? (P - g + G - 1) / G : 0,
            [&](int i) { return x + (long long)(g + i * G) * Cd + (long long)q * V; },
            [&](int i, const typename PV<T>::Raw& raw) {
                float v[V];
                PV<T>::unpack(raw, v);
                const float a = sa[g + i * G];
#pragma unroll
                for (int e = 0; e < V; ++e) { acc[e] = fmaf(a, v[e], acc[e]); mean[e] += v[e]; }
            });
        if (G == 1) {
            if (act) {
                stf<V>(po + (long long)q * V, acc);
                if (mo) {
#pragma unroll
                    for (int e = 0; e < V; ++e) mean[e] *= invP;
                    stf<V>(mo + (long long)q * V, mean);
                }
            }
        } else {                                                       // G > 1 implies QX < 1024: a single trip
            group_reduce<V>(acc, act, g, q, G, QX, 1.f, part, po);
            if (mo) group_reduce<V>(mean, act, g, q, G, QX, invP, part, mo);
        }
    }
}

// ---- backward -----------------------------------------------------------------------------------------------------
// dX[n,p,:] = attn_p d_pooled_n + d_xmean_n / P ;  da_p = <d_pooled_n, X_np> ;  ds_p = attn_p (da_p - sum_q attn_q da_q)
// dH[n,p,j] = ds_p w2_j (1 - tanh^2 H_npj) ;  partial[n][j] = sum_p ds_p tanh H_npj   (dw_2 = sum_n partial)
// dynamic LDS (floats): sa[1024] sds[1024] red[16] segp[PIE_SEGP] gp[Cd] gm[Cd] part[PIE_PART]
template <typename T>
__global__ __launch_bounds__(1024) void cfl_pie_bwd_fused_kernel(const T* __restrict__ X, const T* __restrict__ H,
                                                                 const float* __restrict__ w2, const float* __restrict__ attn,
                                                                 const float* __restrict__ dpooled, const float* __restrict__ dxmean,
                                                                 int P, int Cd, int dh, T* __restrict__ dX, T* __restrict__ dH,
                                                                 float* __restrict__ partial) {
    constexpr int V = PV<T>::V;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sa = lds;
    float* sds = sa + 1024;
    float* red = sds + 1024;
    float* segp = red + 16;
    float* gp = segp + PIE_SEGP;
    float* gm = gp + Cd;
    float* part = gm + Cd;
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float invP = 1.f / (float)P;

    for (int c = t * 4; c < Cd; c += 4096) {
        float a[4], b[4] = {0.f, 0.f, 0.f, 0.f};
        ldf<4>(dpooled + (long long)n * Cd + c, a);
        if (dxmean) {
            ldf<4>(dxmean + (long long)n * Cd + c, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] *= invP;
        }
        stf<4>(gp + c, a);
        stf<4>(gm + c, b);
    }
    if (t < P) sa[t] = attn[(long long)n * P + t];
    __syncthreads();

    // A. one pass over X: partial dots per (row, segment) task, dX written from the staged gradients
    constexpr int SEGW = 64 * V;
    const int nsegx = (Cd + SEGW - 1) / SEGW, ntask = P * nsegx;
    const T* x = X + (long long)n * P * Cd;
    T* dx = dX + (long long)n * P * Cd;
    stream2<T, PIE_U, false>((ntask - w + 15) / 16,
        [&](int i) {
            const int k = w + 16 * i, p = k / nsegx;
            int j = (k - p * nsegx) * SEGW + lane * V;
            j = j < Cd ? j : Cd - V;
            return x + (long long)p * Cd + j;
        },
        [&](int i, const typename PV<T>::Raw& raw) {
            const int k = w + 16 * i, p = k / nsegx;
            const int j = (k - p * nsegx) * SEGW + lane * V;
            float s = 0.f;
            if (j < Cd) {
                float v[V], gv[V], mv[V], o[V];
                PV<T>::unpack(raw, v);
                ldf<V>(gp + j, gv);
                ldf<V>(gm + j, mv);
                const float a = sa[p];
#pragma unroll
                for (int e = 0; e < V; ++e) { s = fmaf(v[e], gv[e], s); o[e] = fmaf(a, gv[e], mv[e]); }
                PV<T>::st(dx + (long long)p * Cd + j, o);
            }
            s = wave_sum(s);
            if (lane == 0) segp[k] = s;
        });
    __syncthreads();

    // B. softmax backward
    float da = 0.f, a = 0.f;
    if (t < P) {
        for (int sg = 0; sg < nsegx; ++sg) da += segp[t * nsegx + sg];
        a = sa[t];
    }
    const float dot = block_sum_1024(a * da, red);
    if (t < P) sds[t] = a * (da - dot);                    // masked positions: attn = 0 => ds = 0
    __syncthreads();

    // C. one pass over H
    const int QH = dh / V;
    const T* h = H + (long long)n * P * dh;
    T* dhp = dH + (long long)n * P * dh;
    float* po = partial + (long long)n * dh;
    const int G = QH >= 1024 ? 1 : (1024 / QH < P ? 1024 / QH : P);
    const int QE = QH >= 1024 ? 1024 : QH;
    const int g = t / QE, q0 = t - g * QE;
    const bool active = g < G;
    for (int qb = 0; qb < QH; qb += 1024) {
        const int q = qb + q0;
        const bool act = active && q < QH;
        float acc[V], wv[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { acc[e] = 0.f; wv[e] = 0.f; }
        if (act) ldf<V>(w2 + (long long)q * V, wv);
        stream2<T, PIE_U, false>(act ? (P - g + G - 1) / G : 0,
            [&](int i) { return h + (long long)(g + i * G) * dh + (long long)q * V; },
            [&](int i, const typename PV<T>::Raw& raw) {
                float v[V], o[V];
                PV<T>::unpack(raw, v);
                const int p = g + i * G;
                const float d = sds[p];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float th = tanh_fast(v[e]);
                    o[e] = d * wv[e] * (1.f - th * th);
                    acc[e] = fmaf(d, th, acc[e]);
                }
                PV<T>::st(dhp + (long long)p * dh + (long long)q * V, o);
            });
        if (G == 1) {
            if (act) stf<V>(po + (long long)q * V, acc);
        } else {
            group_reduce<V>(acc, act, g, q, G, QH, 1.f, part, po);
        }
    }
}

// dw2[j] = sum_n partial[n][j]: 256 columns per workgroup (64 lanes x 4), 16 row groups, fixed order.
__global__ __launch_bounds__(1024) void cfl_pie_dw2_reduce_kernel(const float* __restrict__ partial, int N, int dh, float* __restrict__ dw2) {
    __shared__ __attribute__((aligned(16))) float sm[16][256];
    const int t = threadIdx.x, cq = t & 63, g = t >> 6;
    const int col = blockIdx.x * 256 + cq * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < dh) {
#pragma unroll 8
        for (int nn = g; nn < N; nn += 16) acc += *reinterpret_cast<const f32x4*>(partial + (long long)nn * dh + col);
    }
    *reinterpret_cast<f32x4*>(&sm[g][cq * 4]) = acc;
    __syncthreads();
    if (t < 256 && blockIdx.x * 256 + t < dh) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += sm[i][t];
        dw2[blockIdx.x * 256 + t] = s;
    }
}

int fused_slices(int N, int Cd, int V) {
    int S = 1;
    while (N * S < 256 && S < 8 && Cd % (2 * S * V) == 0 && Cd / (2 * S * V) >= 64) S *= 2;
    return S;
}

size_t bwd_lds_bytes(int Cd) { return (size_t)(1024 + 1024 + 16 + PIE_SEGP + 2 * Cd + PIE_PART) * sizeof(float); }

}  // namespace

extern "C" {

// 1 when the single-pass kernels take [N,P,Cd] / [N,P,dh] operands of this element type (bf16 != 0: bfloat16, else fp32)
int cfl_pie_fused_supported(int N, int P, int Cd, int dh, int bf16) {
    const int V = bf16 ? 8 : 4;
    if (N <= 0 || P <= 0 || P > 1024 || Cd < V || dh < V || Cd % V != 0 || dh % V != 0 || Cd > 8192 || dh > PIE_DH_MAX) return 0;
    const int segw = 64 * V;
    if ((long long)P * cfl_cdiv(dh, segw) > PIE_PART || (long long)P * cfl_cdiv(Cd, segw) > PIE_SEGP) return 0;
    return 1;
}

int cfl_pie_head_fwd(const void* X, const void* H, int bf16, const float* w2, const unsigned char* mask, int N, int P, int Cd,
                     int dh, float* attn, float* pooled, float* xmean, void* stream_) {
    if (!X || !H || !w2 || !attn || !pooled || N <= 0 || P <= 0 || Cd <= 0 || dh <= 0) return CFL_EINVAL;
    if (!cfl_pie_fused_supported(N, P, Cd, dh, bf16)) return CFL_ELIMIT;
    if (((uintptr_t)X | (uintptr_t)H | (uintptr_t)w2 | (uintptr_t)pooled | (uintptr_t)xmean) & 15) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const int S = fused_slices(N, Cd, bf16 ? 8 : 4);
    if (bf16)
        CFL_LAUNCH(K_PIE_FWD_FUSED, cfl_pie_fwd_fused_kernel<u16>, dim3(N, S), dim3(1024), 0, stream, (const u16*)X, (const u16*)H,
                   w2, mask, P, Cd, dh, S, attn, pooled, xmean);
    else
        CFL_LAUNCH(K_PIE_FWD_FUSED, cfl_pie_fwd_fused_kernel<float>, dim3(N, S), dim3(1024), 0, stream, (const float*)X,
                   (const float*)H, w2, mask, P, Cd, dh, S, attn, pooled, xmean);
    return 0;
}

// ws: cfl_pie_ws_bytes(N, P, Cd, dh) (the per-sample dw_2 partials [N, dh])
int cfl_pie_head_bwd(const void* X, const void* H, int bf16, const float* w2, const float* attn, const float* d_pooled,
                     const float* d_xmean, int N, int P, int Cd, int dh, void* dX, void* dH, float* dw2, void* ws, void* stream_) {
    if (!X || !H || !w2 || !attn || !d_pooled || !dX || !dH || !dw2 || !ws || N <= 0 || P <= 0 || Cd <= 0 || dh <= 0) return CFL_EINVAL;
    if (!cfl_pie_fused_supported(N, P, Cd, dh, bf16)) return CFL_ELIMIT;
    if (((uintptr_t)X | (uintptr_t)H | (uintptr_t)w2 | (uintptr_t)d_pooled | (uintptr_t)d_xmean | (uintptr_t)dX | (uintptr_t)dH |
         (uintptr_t)ws) & 15)
        return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    float* partial = (float*)ws;
    const size_t lds = bwd_lds_bytes(Cd);
    if (bf16) {
        CFL_SET_LDS(cfl_pie_bwd_fused_kernel<u16>, bwd_lds_bytes(8192));
        CFL_LAUNCH(K_PIE_BWD_FUSED, cfl_pie_bwd_fused_kernel<u16>, dim3(N), dim3(1024), lds, stream, (const u16*)X, (const u16*)H, w2,
                   attn, d_pooled, d_xmean, P, Cd, dh, (u16*)dX, (u16*)dH, partial);
    } else {
        CFL_SET_LDS(cfl_pie_bwd_fused_kernel<float>, bwd_lds_bytes(8192));
        CFL_LAUNCH(K_PIE_BWD_FUSED, cfl_pie_bwd_fused_kernel<float>, dim3(N), dim3(1024), lds, stream, (const float*)X,
                   (const float*)H, w2, attn, d_pooled, d_xmean, P, Cd, dh, (float*)dX, (float*)dH, partial);
    }
    CFL_LAUNCH(K_PIE_BWD_DW2, cfl_pie_dw2_reduce_kernel, dim3(cfl_cdiv(dh, 256)), dim3(1024), 0, stream, partial, N, dh, dw2);
    return 0;
}

}  // extern "C"
