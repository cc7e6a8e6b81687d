// gru.hip -- row A2c, the text towers' recurrence (src/networks/language_model.py:93-107, src/networks/models/caption_encoder.py:87-101).
//
// The reference runs a packed bidirectional GRU over every caption and then keeps ONE vector per caption: the output at the last
// valid step, `gather(padded, 1, lengths - 1)`.  At that position the forward direction holds its final state (len steps of the
// recurrence) and the backward direction -- which walks from the last word to the first -- holds its FIRST step: one GRU cell on the
// last word from a zero state.  Nothing else of the [B, L, 2H] output is read (the PIE head pools the word embeddings, not the
// states).  The library's GRU computes all of it, both directions, as ~50 small launches per time step (its element-wise tensor
// ops were 52 000 launches = 5 % of the clients' kernel time and most of the text client's host time), and the packed form needs
// the lengths on the host.
//
// Here:  xp = words W_ih^T + b_ih for all (b, t) is ONE library GEMM (the caller's); the recurrence
//     g   = h W_hh^T + b_hh
//     r   = sigmoid(xp_r + g_r),  z = sigmoid(xp_z + g_z),  n = tanh(xp_n + r * g_n),  h' = (1 - z) n + z h        (torch.nn.GRU)
// runs as ONE launch per direction of autograd: batch rows are independent, so a workgroup owns R = 3 rows for all their time
// steps and never talks to another.  3H threads; thread j keeps row j of W_hh (H floats) in registers for the whole launch and
// forms g[., j] for the workgroup's rows from the state in LDS (broadcast reads); then thread (r, i) = (j / H, j % H) does the
// gate arithmetic of one state element.  Lengths are read on the device: no packing, no host copy of the lengths, rows simply
// stop at their own length.  fp32 FMA chains in k order, expf / tanhf of the device library: the reference's client precision.
// The backward launch walks t downwards with W_hh^T columns in registers, writes the pre-activation gradients for every (b, t)
// (zeros beyond a row's length) and leaves the four weight / input gradients to library GEMMs over those buffers.
// The backward direction's single cell is an element-wise kernel pair (cfl_gru_cell0_*).
// This is latency work (T dependent steps of a 3 x H x 3H product): ~2 us per step, ~40 workgroups.
#include "common.h"
#include <cstdlib>

namespace {

template <int H, int R>
__global__ __launch_bounds__(3 * H) void cfl_gru_fwd_kernel(const float* __restrict__ xp, const float* __restrict__ w_hh,
                                                            const float* __restrict__ b_hh, const int* __restrict__ lens,
                                                            float* __restrict__ out, float* __restrict__ hs,
                                                            float* __restrict__ gates, int B, int T) {
    static_assert(R == 3, "one state element per thread: R * H == 3 * H");
    __shared__ __attribute__((aligned(16))) float h[R][H];
    __shared__ float g[R][3 * H];
    __shared__ int slen[R];
    const int j = threadIdx.x, b0 = blockIdx.x * R;
    const int r = j / H, i = j % H, b = b0 + r;
    float w[H];
    {
        const f32x4* wr = reinterpret_cast<const f32x4*>(w_hh + (size_t)j * H);
#pragma unroll
        for (int k = 0; k < H / 4; ++k) {
            const f32x4 v = wr[k];
            w[4 * k] = v[0]; w[4 * k + 1] = v[1]; w[4 * k + 2] = v[2]; w[4 * k + 3] = v[3];
        }
    }
    const float bj = b_hh[j];
    if (j < R) slen[j] = b0 + j < B ? min(max(lens[b0 + j], 0), T) : 0;
    h[r][i] = 0.f;
    __syncthreads();
    const int len = slen[r];
    int tmax = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) tmax = max(tmax, slen[q]);
    const bool live = b < B;
    // hs is time-major [T + 1, B, H]: hs[t] = state BEFORE step t (hs[0] = 0), so hs[:T] is the contiguous "previous state"
    // operand of the weight-gradient GEMM
    if (hs && live) hs[(size_t)b * H + i] = 0.f;
    const float* xrow = xp + (size_t)(live ? b : 0) * T * 3 * H;
    float hv = 0.f;
    float xr = 0.f, xz = 0.f, xn = 0.f;
    if (len > 0) { xr = xrow[i]; xz = xrow[H + i]; xn = xrow[2 * H + i]; }
    for (int t = 0; t < tmax; ++t) {
        float acc[R];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = bj;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(&h[q][k]);
                acc[q] = fmaf(w[k], s[0], acc[q]);
                acc[q] = fmaf(w[k + 1], s[1], acc[q]);
                acc[q] = fmaf(w[k + 2], s[2], acc[q]);
                acc[q] = fmaf(w[k + 3], s[3], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) g[q][j] = acc[q];
        __syncthreads();
        const bool act = t < len;
        float nxr = 0.f, nxz = 0.f, nxn = 0.f;
        if (t + 1 < len) {                                           // next step's input projections: in flight over the gate math
            const float* xq = xrow + (size_t)(t + 1) * 3 * H;
            nxr = xq[i]; nxz = xq[H + i]; nxn = xq[2 * H + i];
        }
        if (act) {
            const float gr = g[r][i], gz = g[r][H + i], gn = g[r][2 * H + i];
            const float rr = sigmoidf(xr + gr), zz = sigmoidf(xz + gz);
            const float nn = tanhf(xn + rr * gn);
            hv = (1.f - zz) * nn + zz * hv;
            h[r][i] = hv;
            if (gates) {
                float* gp = gates + ((size_t)b * T + t) * 4 * H;
                gp[i] = rr; gp[H + i] = zz; gp[2 * H + i] = nn; gp[3 * H + i] = gn;
            }
        }
        if (hs && live) hs[((size_t)(t + 1) * B + b) * H + i] = hv;
        xr = nxr; xz = nxz; xn = nxn;
        __syncthreads();
    }
    if (live) {
        out[(size_t)b * H + i] = hv;
        if (hs)
            for (int t = tmax; t < T; ++t) hs[((size_t)(t + 1) * B + b) * H + i] = hv;     // finite beyond the length (times a zero gradient)
    }
}

template <int H, int R>
__global__ __launch_bounds__(3 * H) void cfl_gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ w_hh,
                                                            const int* __restrict__ lens, const float* __restrict__ hs,
                                                            const float* __restrict__ gates, float* __restrict__ dxp,
                                                            float* __restrict__ dg, int B, int T) {
    static_assert(R == 3, "one state element per thread");
    __shared__ __attribute__((aligned(16))) float sg[R][3 * H];       // gradients of the hidden-side pre-activations of this step
    __shared__ float part[R][3][H];
    __shared__ int slen[R];
    const int j = threadIdx.x, b0 = blockIdx.x * R;
    const int r = j / H, i = j % H, b = b0 + r;                        // as item: row r, state element i; as product thread: gate block r
    float w[H];                                                        // w[k] = W_hh[r * H + k][i]: column i of gate block r
#pragma unroll
    for (int k = 0; k < H; ++k) w[k] = w_hh[((size_t)r * H + k) * H + i];
    if (j < R) slen[j] = b0 + j < B ? min(max(lens[b0 + j], 0), T) : 0;
    __syncthreads();
    const int len = slen[r];
    int tmax = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) tmax = max(tmax, slen[q]);
    const bool live = b < B;
    float dh = live ? dout[(size_t)b * H + i] : 0.f;
    const float* grow = gates + (size_t)(live ? b : 0) * T * 4 * H;
    float* dxrow = dxp + (size_t)(live ? b : 0) * T * 3 * H;           // batch-major [B, T, 3H] (meets the batch-major word embeddings)
    float rr = 0.f, zz = 0.f, nn = 0.f, gn = 0.f, hp = 0.f;
    auto fetch = [&](int t) {
        const float* gp = grow + (size_t)t * 4 * H;
        rr = gp[i]; zz = gp[H + i]; nn = gp[2 * H + i]; gn = gp[3 * H + i];
        hp = hs[((size_t)t * B + b) * H + i];
    };
    if (tmax > 0 && tmax - 1 < len) fetch(tmax - 1);
    for (int t = tmax - 1; t >= 0; --t) {
        const bool act = t < len;
        float keep = dh;
        if (act) {
            const float dn = dh * (1.f - zz), dz = dh * (hp - nn);
            const float dan = dn * (1.f - nn * nn);
            const float daz = dz * zz * (1.f - zz);
            const float dar = dan * gn * rr * (1.f - rr);
            const float dgn = dan * rr;
            keep = dh * zz;
            sg[r][i] = dar; sg[r][H + i] = daz; sg[r][2 * H + i] = dgn;
            float* dx = dxrow + (size_t)t * 3 * H;
            dx[i] = dar; dx[H + i] = daz; dx[2 * H + i] = dan;
            float* dq = dg + ((size_t)t * B + b) * 3 * H;              // time-major [T, B, 3H] (meets hs[:T])
            dq[i] = dar; dq[H + i] = daz; dq[2 * H + i] = dgn;
        } else {
            sg[r][i] = 0.f; sg[r][H + i] = 0.f; sg[r][2 * H + i] = 0.f;
        }
        __syncthreads();
        if (t > 0 && t - 1 < len) fetch(t - 1);                        // in flight over the product
        float acc[R];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = 0.f;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(&sg[q][r * H + k]);
                acc[q] = fmaf(w[k], s[0], acc[q]);
                acc[q] = fmaf(w[k + 1], s[1], acc[q]);
                acc[q] = fmaf(w[k + 2], s[2], acc[q]);
                acc[q] = fmaf(w[k + 3], s[3], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) part[q][r][i] = acc[q];
        __syncthreads();
        dh = keep + (part[r][0][i] + part[r][1][i] + part[r][2][i]);
    }
    if (live) {
        for (int t = len; t < T; ++t) {
            float* dx = dxrow + (size_t)t * 3 * H;
            dx[i] = 0.f; dx[H + i] = 0.f; dx[2 * H + i] = 0.f;
            float* dq = dg + ((size_t)t * B + b) * 3 * H;
            dq[i] = 0.f; dq[H + i] = 0.f; dq[2 * H + i] = 0.f;
        }
    }
}

// ---- any width (H % 4 == 0, H <= 512): W_hh streamed from L2 every step ----------------------------------------------------------
// Beyond H = 128 a row of W_hh no longer fits a thread's registers (H = 256: 786 KB per matrix, more than a CU's register file).
// Same block structure (a workgroup owns R = 2 rows for all their steps), but the matrix passes L2 -> CU once per step and
// workgroup (3H * H * 4 bytes: ~5 us at H = 256 at the ~64 B/clk a CU pulls), which bounds the step.  To keep that stream busy
// the CONTRACTION is split over the 8 waves: wave w owns k in [w K/8, (w + 1) K/8), lane l the output columns 4l .. 4l + 3 (+ 256 c)
// -- one 16-byte load per lane, 1 KB contiguous per wave and k (forward: W_hh^T [H, 3H]; backward: W_hh [3H, H] itself) -- with a
// two-stage register ring so that 6 KB per wave are in flight while the previous stage is multiplied; the 8 partial sums meet
// in LDS and are added in wave order (deterministic).  First version of these kernels (contraction not split, 4-byte loads,
// no ring): 70 us per step at H = 256.
constexpr int GS_T = 512, GS_W = 8, GS_R = 2;

template <int C>                                                    // 16-byte column groups per lane: 3H <= 1024 C
__global__ __launch_bounds__(GS_T) void cfl_gru_fwd_stream_kernel(const float* __restrict__ xp, const float* __restrict__ w_hh_t,
                                                                  const float* __restrict__ b_hh, const int* __restrict__ lens,
                                                                  float* __restrict__ out, float* __restrict__ hs,
                                                                  float* __restrict__ gates, int B, int T, int H) {
    constexpr int KS = C <= 3 ? 2 : 1;                                // k per ring stage: 6 loads of 16 bytes per lane and stage
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = 3 * H;
    float* h = sm;                                                    // [R][H]
    float* part = sm + GS_R * H;                                      // [W][R][N]
    __shared__ int slen[GS_R];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b0 = blockIdx.x * GS_R;
    if (tid < GS_R) slen[tid] = b0 + tid < B ? min(max(lens[b0 + tid], 0), T) : 0;
    for (int e = tid; e < GS_R * H; e += GS_T) {
        h[e] = 0.f;
        const int b = b0 + e / H;
        if (hs && b < B) hs[(size_t)b * H + e % H] = 0.f;
    }
    __syncthreads();
    int tmax = 0;
#pragma unroll
    for (int q = 0; q < GS_R; ++q) tmax = max(tmax, slen[q]);
    const int kper = (H + GS_W - 1) / GS_W, k_lo = min(w * kper, H), k_hi = min(k_lo + kper, H);
    const int nst = (k_hi - k_lo + KS - 1) / KS;
    bool colok[C];
#pragma unroll
    for (int c = 0; c < C; ++c) colok[c] = 4 * lane + 256 * c < N;
    for (int t = 0; t < tmax; ++t) {
        f32x4 acc[GS_R][C];
#pragma unroll
        for (int q = 0; q < GS_R; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 ring0[KS][C], ring1[KS][C];                                // (two named stages: the slot must be a compile-time index)
        auto issue = [&](int st, f32x4 (&ring)[KS][C]) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int k = min(k_lo + st * KS + kk, H - 1);
#pragma unroll
                for (int c = 0; c < C; ++c)
                    ring[kk][c] = colok[c] ? *reinterpret_cast<const f32x4*>(w_hh_t + (size_t)k * N + 4 * lane + 256 * c)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto mult = [&](int st, const f32x4 (&ring)[KS][C]) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int k = k_lo + st * KS + kk;
                if (k < k_hi) {
#pragma unroll
                    for (int q = 0; q < GS_R; ++q) {
                        const float hv = h[q * H + k];
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[q][c] += ring[kk][c] * hv;
                    }
                }
            }
        };
        if (nst > 0) issue(0, ring0);
        for (int st = 0; st < nst; st += 2) {
            if (st + 1 < nst) issue(st + 1, ring1);
            mult(st, ring0);
            if (st + 2 < nst) issue(st + 2, ring0);
            if (st + 1 < nst) mult(st + 1, ring1);
        }
#pragma unroll
        for (int q = 0; q < GS_R; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (colok[c]) *reinterpret_cast<f32x4*>(part + ((size_t)w * GS_R + q) * N + 4 * lane + 256 * c) = acc[q][c];
        __syncthreads();
        for (int e = tid; e < GS_R * H; e += GS_T) {
            const int q = e / H, i = e % H, b = b0 + q;
            if (b >= B) continue;
            float hv = h[e];
            if (t < slen[q]) {
                float gr = b_hh[i], gz = b_hh[H + i], gn = b_hh[2 * H + i];
#pragma unroll
                for (int ww = 0; ww < GS_W; ++ww) {
                    const float* pp = part + ((size_t)ww * GS_R + q) * N;
                    gr += pp[i]; gz += pp[H + i]; gn += pp[2 * H + i];
                }
                const float* xq = xp + ((size_t)b * T + t) * N;
                const float rr = sigmoidf(xq[i] + gr), zz = sigmoidf(xq[H + i] + gz);
                const float nn = tanhf(xq[2 * H + i] + rr * gn);
                hv = (1.f - zz) * nn + zz * hv;
                h[e] = hv;
                if (gates) {
                    float* gp = gates + ((size_t)b * T + t) * 4 * H;
                    gp[i] = rr; gp[H + i] = zz; gp[2 * H + i] = nn; gp[3 * H + i] = gn;
                }
            }
            if (hs) hs[((size_t)(t + 1) * B + b) * H + i] = hv;
        }
        __syncthreads();
    }
    for (int e = tid; e < GS_R * H; e += GS_T) {
        const int i = e % H, b = b0 + e / H;
        if (b >= B) continue;
        out[(size_t)b * H + i] = h[e];
        if (hs)
            for (int t = tmax; t < T; ++t) hs[((size_t)(t + 1) * B + b) * H + i] = h[e];
    }
}

// backward: the contraction runs over the 3H hidden-side pre-activation gradients, the outputs are the H state-gradient elements:
// lane l owns elements 4l .. 4l + 3 (+ 256 c) of W_hh's rows, wave w the rows j in its eighth of [0, 3H).
template <int C>                                                    // H <= 256 C
__global__ __launch_bounds__(GS_T) void cfl_gru_bwd_stream_kernel(const float* __restrict__ dout, const float* __restrict__ w_hh,
                                                                  const int* __restrict__ lens, const float* __restrict__ hs,
                                                                  const float* __restrict__ gates, float* __restrict__ dxp,
                                                                  float* __restrict__ dg, int B, int T, int H) {
    constexpr int KS = C == 1 ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = 3 * H;
    float* sg = sm;                                                   // [R][3H] hidden-side pre-activation gradients of this step
    float* dh = sm + GS_R * N;                                        // [R][H]  gradient of the state after step t
    float* keep = dh + GS_R * H;                                      // [R][H]  dh * z
    float* part = keep + GS_R * H;                                    // [W][R][H]
    __shared__ int slen[GS_R];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b0 = blockIdx.x * GS_R;
    if (tid < GS_R) slen[tid] = b0 + tid < B ? min(max(lens[b0 + tid], 0), T) : 0;
    for (int e = tid; e < GS_R * H; e += GS_T) {
        const int b = b0 + e / H;
        dh[e] = b < B ? dout[(size_t)b * H + e % H] : 0.f;
    }
    __syncthreads();
    int tmax = 0;
#pragma unroll
    for (int q = 0; q < GS_R; ++q) tmax = max(tmax, slen[q]);
    const int jper = (N + GS_W - 1) / GS_W, j_lo = min(w * jper, N), j_hi = min(j_lo + jper, N);
    const int nst = (j_hi - j_lo + KS - 1) / KS;
    bool colok[C];
#pragma unroll
    for (int c = 0; c < C; ++c) colok[c] = 4 * lane + 256 * c < H;
    for (int t = tmax - 1; t >= 0; --t) {
        for (int e = tid; e < GS_R * H; e += GS_T) {
            const int q = e / H, i = e % H, b = b0 + q;
            float dar = 0.f, daz = 0.f, dgn = 0.f, kp = dh[e];
            if (b < B && t < slen[q]) {
                const float* gp = gates + ((size_t)b * T + t) * 4 * H;
                const float rr = gp[i], zz = gp[H + i], nn = gp[2 * H + i], gn = gp[3 * H + i];
                const float hp = hs[((size_t)t * B + b) * H + i], d = dh[e];
                const float dan = d * (1.f - zz) * (1.f - nn * nn);
                daz = d * (hp - nn) * zz * (1.f - zz);
                dar = dan * gn * rr * (1.f - rr);
                dgn = dan * rr;
                kp = d * zz;
                float* dx = dxp + ((size_t)b * T + t) * N;
                dx[i] = dar; dx[H + i] = daz; dx[2 * H + i] = dan;
                float* dq = dg + ((size_t)t * B + b) * N;
                dq[i] = dar; dq[H + i] = daz; dq[2 * H + i] = dgn;
            }
            sg[q * N + i] = dar; sg[q * N + H + i] = daz; sg[q * N + 2 * H + i] = dgn;
            keep[e] = kp;
        }
        __syncthreads();
        f32x4 acc[GS_R][C];
#pragma unroll
        for (int q = 0; q < GS_R; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 ring0[KS][C], ring1[KS][C];
        auto issue = [&](int st, f32x4 (&ring)[KS][C]) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int j = min(j_lo + st * KS + kk, N - 1);
#pragma unroll
                for (int c = 0; c < C; ++c)
                    ring[kk][c] = colok[c] ? *reinterpret_cast<const f32x4*>(w_hh + (size_t)j * H + 4 * lane + 256 * c)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto mult = [&](int st, const f32x4 (&ring)[KS][C]) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int j = j_lo + st * KS + kk;
                if (j < j_hi) {
#pragma unroll
                    for (int q = 0; q < GS_R; ++q) {
                        const float sv = sg[q * N + j];
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[q][c] += ring[kk][c] * sv;
                    }
                }
            }
        };
        if (nst > 0) issue(0, ring0);
        for (int st = 0; st < nst; st += 2) {
            if (st + 1 < nst) issue(st + 1, ring1);
            mult(st, ring0);
            if (st + 2 < nst) issue(st + 2, ring0);
            if (st + 1 < nst) mult(st + 1, ring1);
        }
#pragma unroll
        for (int q = 0; q < GS_R; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (colok[c]) *reinterpret_cast<f32x4*>(part + ((size_t)w * GS_R + q) * H + 4 * lane + 256 * c) = acc[q][c];
        __syncthreads();
        for (int e = tid; e < GS_R * H; e += GS_T) {
            const int q = e / H;
            if (t < slen[q]) {
                float a = 0.f;
#pragma unroll
                for (int ww = 0; ww < GS_W; ++ww) a += part[((size_t)ww * GS_R + q) * H + e % H];
                dh[e] = keep[e] + a;
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < GS_R * H; e += GS_T) {
        const int q = e / H, i = e % H, b = b0 + q;
        if (b >= B) continue;
        for (int t = slen[q]; t < T; ++t) {
            float* dx = dxp + ((size_t)b * T + t) * N;
            dx[i] = 0.f; dx[H + i] = 0.f; dx[2 * H + i] = 0.f;
            float* dq = dg + ((size_t)t * B + b) * N;
            dq[i] = 0.f; dq[H + i] = 0.f; dq[2 * H + i] = 0.f;
        }
    }
}

// One GRU cell from a zero state (the backward direction's output at the last valid position): gx = x W_ih^T + b_ih [B, 3H].
__global__ __launch_bounds__(256) void cfl_gru_cell0_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ b_hh,
                                                                float* __restrict__ out, float* __restrict__ saved, int B, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, i = idx % H;
    const float* x = gx + (size_t)b * 3 * H;
    const float rr = sigmoidf(x[i] + b_hh[i]), zz = sigmoidf(x[H + i] + b_hh[H + i]);
    const float nn = tanhf(x[2 * H + i] + rr * b_hh[2 * H + i]);
    out[idx] = (1.f - zz) * nn;
    if (saved) {
        float* s = saved + (size_t)b * 3 * H;
        s[i] = rr; s[H + i] = zz; s[2 * H + i] = nn;
    }
}

// dgx [B, 3H] = gradient of gx (and of b_ih once summed over b); dgh [B, 3H] = gradient of the hidden-side bias b_hh per row.
__global__ __launch_bounds__(256) void cfl_gru_cell0_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ saved,
                                                                const float* __restrict__ b_hh, float* __restrict__ dgx,
                                                                float* __restrict__ dgh, int B, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, i = idx % H;
    const float* s = saved + (size_t)b * 3 * H;
    const float rr = s[i], zz = s[H + i], nn = s[2 * H + i], dh = dout[idx];
    const float dan = dh * (1.f - zz) * (1.f - nn * nn);
    const float daz = -dh * nn * zz * (1.f - zz);
    const float dar = dan * b_hh[2 * H + i] * rr * (1.f - rr);
    float* x = dgx + (size_t)b * 3 * H;
    float* q = dgh + (size_t)b * 3 * H;
    x[i] = dar; x[H + i] = daz; x[2 * H + i] = dan;
    q[i] = dar; q[H + i] = daz; q[2 * H + i] = dan * rr;
}

constexpr int GRU_R = 3;

}  // namespace

extern "C" {

static bool gru_in_registers(int H) {
    static const bool force_stream = [] { const char* e = getenv("CFL_GRU_STREAM"); return e && e[0] == '1'; }();   // (A/B switch)
    return !force_stream && (H == 32 || H == 64 || H == 128);
}
int cfl_gru_supported(int H) { return gru_in_registers(H) || (H > 0 && H % 4 == 0 && H <= 512); }
int cfl_gru_streams_weights(int H) { return cfl_gru_supported(H) && !gru_in_registers(H); }

int cfl_gru_fwd(const float* xp, const float* w_hh, const float* w_hh_t, const float* b_hh, const int* lens, float* out, float* hs,
                float* gates, int B, int T, int H, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!cfl_gru_supported(H)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (!gru_in_registers(H)) {
        if (!w_hh_t) return (int)hipErrorInvalidValue;
        const size_t lds = ((size_t)GS_R * H + (size_t)GS_W * GS_R * 3 * H) * sizeof(float);        // H = 512: 102 KB
        const dim3 grid(cfl_cdiv(B, GS_R));
        if (3 * H <= 768) {
            CFL_SET_LDS(cfl_gru_fwd_stream_kernel<3>, 100 * 1024);
            CFL_LAUNCH(K_GRU_FWD, cfl_gru_fwd_stream_kernel<3>, grid, dim3(GS_T), lds, st, xp, w_hh_t, b_hh, lens, out, hs, gates, B, T, H);
        } else {
            CFL_SET_LDS(cfl_gru_fwd_stream_kernel<6>, 104 * 1024);
            CFL_LAUNCH(K_GRU_FWD, cfl_gru_fwd_stream_kernel<6>, grid, dim3(GS_T), lds, st, xp, w_hh_t, b_hh, lens, out, hs, gates, B, T, H);
        }
        return 0;
    }
    const dim3 grid(cfl_cdiv(B, GRU_R));
    if (H == 128)
        CFL_LAUNCH(K_GRU_FWD, (cfl_gru_fwd_kernel<128, GRU_R>), grid, dim3(384), 0, st, xp, w_hh, b_hh, lens, out, hs, gates, B, T);
    else if (H == 64)
        CFL_LAUNCH(K_GRU_FWD, (cfl_gru_fwd_kernel<64, GRU_R>), grid, dim3(192), 0, st, xp, w_hh, b_hh, lens, out, hs, gates, B, T);
    else
        CFL_LAUNCH(K_GRU_FWD, (cfl_gru_fwd_kernel<32, GRU_R>), grid, dim3(96), 0, st, xp, w_hh, b_hh, lens, out, hs, gates, B, T);
    return 0;
}

int cfl_gru_bwd(const float* dout, const float* w_hh, const int* lens, const float* hs, const float* gates, float* dxp, float* dg,
                int B, int T, int H, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!cfl_gru_supported(H)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (!gru_in_registers(H)) {
        const size_t lds = ((size_t)GS_R * 3 * H + 2 * (size_t)GS_R * H + (size_t)GS_W * GS_R * H) * sizeof(float);   // H = 512: 53 KB
        const dim3 grid(cfl_cdiv(B, GS_R));
        if (H <= 256)
            CFL_LAUNCH(K_GRU_BWD, cfl_gru_bwd_stream_kernel<1>, grid, dim3(GS_T), lds, st, dout, w_hh, lens, hs, gates, dxp, dg, B, T, H);
        else
            CFL_LAUNCH(K_GRU_BWD, cfl_gru_bwd_stream_kernel<2>, grid, dim3(GS_T), lds, st, dout, w_hh, lens, hs, gates, dxp, dg, B, T, H);
        return 0;
    }
    const dim3 grid(cfl_cdiv(B, GRU_R));
    if (H == 128)
        CFL_LAUNCH(K_GRU_BWD, (cfl_gru_bwd_kernel<128, GRU_R>), grid, dim3(384), 0, st, dout, w_hh, lens, hs, gates, dxp, dg, B, T);
    else if (H == 64)
        CFL_LAUNCH(K_GRU_BWD, (cfl_gru_bwd_kernel<64, GRU_R>), grid, dim3(192), 0, st, dout, w_hh, lens, hs, gates, dxp, dg, B, T);
    else
        CFL_LAUNCH(K_GRU_BWD, (cfl_gru_bwd_kernel<32, GRU_R>), grid, dim3(96), 0, st, dout, w_hh, lens, hs, gates, dxp, dg, B, T);
    return 0;
}

int cfl_gru_cell0_fwd(const float* gx, const float* b_hh, float* out, float* saved, int B, int H, void* stream) {
    if (B <= 0 || H <= 0) return 0;
    CFL_LAUNCH(K_GRU_CELL0, cfl_gru_cell0_fwd_kernel, dim3(cfl_cdiv(B * H, 256)), dim3(256), 0, (hipStream_t)stream, gx, b_hh, out,
               saved, B, H);
    return 0;
}

int cfl_gru_cell0_bwd(const float* dout, const float* saved, const float* b_hh, float* dgx, float* dgh, int B, int H, void* stream) {
    if (B <= 0 || H <= 0) return 0;
    CFL_LAUNCH(K_GRU_CELL0, cfl_gru_cell0_bwd_kernel, dim3(cfl_cdiv(B * H, 256)), dim3(256), 0, (hipStream_t)stream, dout, saved,
               b_hh, dgx, dgh, B, H);
    return 0;
}

}  // extern "C"
