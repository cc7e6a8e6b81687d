// pie.hip -- row A2-head: the PCME attention-pooling head (PIENet) around its two library GEMMs.
//
// Reference: src/networks/models/pie_model.py:28-40 (MultiHeadSelfAttention, n_head = 1),
//            :61-67 (PIENet.forward), src/networks/models/image_encoder.py:54-71 (avgpool + glue),
//            src/utils/tensor_utils.py:25-27 (l2_normalize).
//
// All kernels here are HBM-bound streaming kernels: every element of X [N,P,Cd] and H [N,P,dh]
// is read once per pass with 16-byte lane-contiguous loads; reductions are wavefront shuffles.
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// scores[row] = w2 . tanh(H[row, :]); one wave per (n, p) row.
__global__ __launch_bounds__(256) void cfl_pie_scores_kernel(const float* __restrict__ H, const float* __restrict__ w2,
                                                             const unsigned char* __restrict__ mask, long long rows, int dh,
                                                             int vec, float* scores) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* h = H + row * dh;
    float s = 0.f;
    if (vec) {
        for (int j = lane * 4; j < dh; j += 256) {
            const f32x4 v = ld4(h + j), w = ld4(w2 + j);
            s = fmaf(w[0], tanhf(v[0]), s); s = fmaf(w[1], tanhf(v[1]), s);
            s = fmaf(w[2], tanhf(v[2]), s); s = fmaf(w[3], tanhf(v[3]), s);
        }
    } else {
        for (int j = lane; j < dh; j += 64) s = fmaf(w2[j], tanhf(h[j]), s);
    }
    s = wave_sum(s);
    if (lane == 0) scores[row] = (mask && mask[row]) ? -INFINITY : s;
}

// softmax over P (per sample) + attention pooling + mean pooling.  grid (N, ceil(Cd / 1024)).
__global__ __launch_bounds__(256) void cfl_pie_pool_kernel(const float* __restrict__ X, const float* __restrict__ scores,
                                                           int P, int Cd, int vec, float* attn, float* pooled, float* xmean) {
    __shared__ float sa[1024];
    __shared__ float red[4];
    const int n = blockIdx.x, t = threadIdx.x;
    float mx = -INFINITY;
    for (int p = t; p < P; p += 256) { const float s = scores[(long long)n * P + p]; sa[p] = s; mx = fmaxf(mx, s); }
    mx = wave_max(mx);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int p = t; p < P; p += 256) { const float e = expf(sa[p] - mx); sa[p] = e; sum += e; }
    sum = block_sum_256(sum, red);
    const float inv = 1.f / sum;
    for (int p = t; p < P; p += 256) {
        const float a = sa[p] * inv;
        sa[p] = a;
        if (blockIdx.y == 0) attn[(long long)n * P + p] = a;
    }
    __syncthreads();
    const float* x = X + (long long)n * P * Cd;
    const float invP = 1.f / (float)P;
    if (vec) {
        const int c = blockIdx.y * 1024 + t * 4;
        if (c >= Cd) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, mean = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 7
        for (int p = 0; p < P; ++p) {
            const f32x4 v = ld4(x + (long long)p * Cd + c);
            const float a = sa[p];
            acc[0] = fmaf(a, v[0], acc[0]); acc[1] = fmaf(a, v[1], acc[1]);
            acc[2] = fmaf(a, v[2], acc[2]); acc[3] = fmaf(a, v[3], acc[3]);
            mean += v;
        }
        st4(pooled + (long long)n * Cd + c, acc);
        if (xmean) st4(xmean + (long long)n * Cd + c, mean * invP);
    } else {
        for (int c = blockIdx.y * 1024 + t; c < min(Cd, (int)(blockIdx.y + 1) * 1024); c += 256) {
            float acc = 0.f, mean = 0.f;
            for (int p = 0; p < P; ++p) { const float v = x[(long long)p * Cd + c]; acc = fmaf(sa[p], v, acc); mean += v; }
            pooled[(long long)n * Cd + c] = acc;
            if (xmean) xmean[(long long)n * Cd + c] = mean * invP;
        }
    }
}

// backward of the softmax: da_p = <d_pooled_n, X_np>; ds_p = attn_p (da_p - sum_q attn_q da_q).
// One 1024-thread block (16 waves, one wave per position p at a time) per sample n.
__global__ __launch_bounds__(1024) void cfl_pie_bwd_ds_kernel(const float* __restrict__ X, const float* __restrict__ attn,
                                                              const float* __restrict__ dpooled, int P, int Cd, int vec, float* ds) {
    __shared__ float sda[1024];
    __shared__ float red[16];
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float* x = X + (long long)n * P * Cd;
    const float* g = dpooled + (long long)n * Cd;
    for (int p = w; p < P; p += 16) {
        float s = 0.f;
        if (vec) {
            for (int c = lane * 4; c < Cd; c += 256) {
                const f32x4 v = ld4(x + (long long)p * Cd + c), d = ld4(g + c);
                s = fmaf(v[0], d[0], s); s = fmaf(v[1], d[1], s); s = fmaf(v[2], d[2], s); s = fmaf(v[3], d[3], s);
            }
        } else {
            for (int c = lane; c < Cd; c += 64) s = fmaf(x[(long long)p * Cd + c], g[c], s);
        }
        s = wave_sum(s);
        if (lane == 0) sda[p] = s;
    }
    __syncthreads();
    float dot = 0.f;
    for (int p = t; p < P; p += 1024) dot = fmaf(attn[(long long)n * P + p], sda[p], dot);
    dot = wave_sum(dot);
    if (lane == 0) red[w] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) dot += red[i];
    for (int p = t; p < P; p += 1024) ds[(long long)n * P + p] = attn[(long long)n * P + p] * (sda[p] - dot);
}

// dX[n,p,c] = attn[n,p] * d_pooled[n,c] + d_xmean[n,c] / P.   grid (N*P, ceil(Cd/1024))
__global__ __launch_bounds__(256) void cfl_pie_bwd_dx_kernel(const float* __restrict__ attn, const float* __restrict__ dpooled,
                                                             const float* __restrict__ dxmean, int P, int Cd, int vec, float* dX) {
    const long long row = blockIdx.x;
    const int n = (int)(row / P);
    const float a = attn[row];
    const float invP = 1.f / (float)P;
    if (vec) {
        const int c = blockIdx.y * 1024 + threadIdx.x * 4;
        if (c >= Cd) return;
        f32x4 v = ld4(dpooled + (long long)n * Cd + c) * a;
        if (dxmean) v += ld4(dxmean + (long long)n * Cd + c) * invP;
        st4(dX + row * Cd + c, v);
    } else {
        for (int c = blockIdx.y * 1024 + threadIdx.x; c < min(Cd, (int)(blockIdx.y + 1) * 1024); c += 256) {
            float v = a * dpooled[(long long)n * Cd + c];
            if (dxmean) v += dxmean[(long long)n * Cd + c] * invP;
            dX[row * Cd + c] = v;
        }
    }
}

// dH[row,j] = ds[row] w2[j] (1 - tanh^2 H[row,j]);  partial[chunk][j] = sum_{rows in chunk} ds[row] tanh(H[row,j])
// grid (row chunks of PIE_RC rows, ceil(dh/1024)); a thread owns 4 consecutive columns (16-byte accesses).
constexpr int PIE_RC = 32;
__global__ __launch_bounds__(256) void cfl_pie_bwd_dh_kernel(const float* __restrict__ H, const float* __restrict__ w2,
                                                             const float* __restrict__ ds, long long rows, int dh, int vec,
                                                             float* dH, float* partial) {
    const long long r0 = (long long)blockIdx.x * PIE_RC;
    const long long r1 = min(rows, r0 + PIE_RC);
    if (vec) {
        const int j = blockIdx.y * 1024 + threadIdx.x * 4;
        if (j >= dh) return;
        const f32x4 w = ld4(w2 + j);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (long long r = r0; r < r1; ++r) {
            const f32x4 h = ld4(H + r * dh + j);
            const float d = ds[r];                      // masked rows have attn = 0 => ds = 0
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = tanhf(h[k]);
                o[k] = d * w[k] * (1.f - t * t);
                acc[k] = fmaf(d, t, acc[k]);
            }
            st4(dH + r * dh + j, o);
        }
        st4(partial + (long long)blockIdx.x * dh + j, acc);
    } else {
        for (int j = blockIdx.y * 1024 + threadIdx.x; j < min(dh, (int)(blockIdx.y + 1) * 1024); j += 256) {
            const float w = w2[j];
            float acc = 0.f;
            for (long long r = r0; r < r1; ++r) {
                const float t = tanhf(H[r * dh + j]);
                const float d = ds[r];
                dH[r * dh + j] = d * w * (1.f - t * t);
                acc = fmaf(d, t, acc);
            }
            partial[(long long)blockIdx.x * dh + j] = acc;
        }
    }
}
// dw2[j] = sum over chunks of partial[c][j]; block = 64 columns x 4 chunk groups
__global__ __launch_bounds__(256) void cfl_pie_bwd_dw2_kernel(const float* __restrict__ partial, int nchunks, int dh, float* dw2) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (j < dh)
        for (int c = w; c < nchunks; c += 4) s += partial[(long long)c * dh + j];
    sm[w][lane] = s;
    __syncthreads();
    if (w == 0 && j < dh) dw2[j] = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
}

// epilogue fwd: one wave per row.
__global__ __launch_bounds__(256) void cfl_pie_epi_fwd_kernel(const float* __restrict__ out, const float* __restrict__ res_pre,
                                                              const float* __restrict__ lw, const float* __restrict__ lb,
                                                              int N, int D, float eps, int flags, float* y, float* o, float* r,
                                                              float* stats) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const long long base = (long long)n * D;
    float sum = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float rv = sigmoidf(res_pre[base + k]);
        if (r) r[base + k] = rv;
        sum += out[base + k] + rv;
    }
    const float mean = wave_sum(sum) / (float)D;
    float var = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float z = out[base + k] + sigmoidf(res_pre[base + k]) - mean;
        var = fmaf(z, z, var);
    }
    const float rstd = 1.f / sqrtf(wave_sum(var) / (float)D + eps);
    float nn = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float z = out[base + k] + sigmoidf(res_pre[base + k]);
        const float ov = (z - mean) * rstd * lw[k] + lb[k];
        if (o) o[base + k] = ov;
        nn = fmaf(ov, ov, nn);
    }
    const float inv = (flags & CFL_EPI_NO_L2NORM) ? 1.f : 1.f / fmaxf(sqrtf(wave_sum(nn)), 1e-12f);
    for (int k = lane; k < D; k += 64) {
        const float z = out[base + k] + sigmoidf(res_pre[base + k]);
        y[base + k] = ((z - mean) * rstd * lw[k] + lb[k]) * inv;
    }
    if (lane == 0) { stats[n * 4 + 0] = mean; stats[n * 4 + 1] = rstd; stats[n * 4 + 2] = inv; stats[n * 4 + 3] = 0.f; }
}

// epilogue bwd (row part): one wave per row.  go [N,D] (ws) receives the total gradient w.r.t. o.
__global__ __launch_bounds__(256) void cfl_pie_epi_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ do_,
                                                              const float* __restrict__ dres, const float* __restrict__ out,
                                                              const float* __restrict__ r, const float* __restrict__ lw,
                                                              const float* __restrict__ lb, const float* __restrict__ stats,
                                                              int N, int D, int flags, float* d_out, float* d_res_pre, float* go) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const long long base = (long long)n * D;
    const float mean = stats[n * 4 + 0], rstd = stats[n * 4 + 1], inv = stats[n * 4 + 2];
    const bool l2 = !(flags & CFL_EPI_NO_L2NORM);
    // <dy, y> with y = o * inv
    float dyy = 0.f;
    if (l2) {
        for (int k = lane; k < D; k += 64) {
            const float xh = (out[base + k] + r[base + k] - mean) * rstd;
            const float yv = (xh * lw[k] + lb[k]) * inv;
            dyy = fmaf(dy[base + k], yv, dyy);
        }
        dyy = wave_sum(dyy);
    }
    float s1 = 0.f, s2 = 0.f;      // sum(dxhat), sum(dxhat * xhat)
    for (int k = lane; k < D; k += 64) {
        const float xh = (out[base + k] + r[base + k] - mean) * rstd;
        float g = dy[base + k];
        if (l2) { const float yv = (xh * lw[k] + lb[k]) * inv; g = inv * (g - yv * dyy); }
        if (do_) g += do_[base + k];
        go[base + k] = g;
        const float dxh = g * lw[k];
        s1 += dxh; s2 = fmaf(dxh, xh, s2);
    }
    s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
    for (int k = lane; k < D; k += 64) {
        const float rv = r[base + k];
        const float xh = (out[base + k] + rv - mean) * rstd;
        const float dz = rstd * (go[base + k] * lw[k] - s1 - xh * s2);
        d_out[base + k] = dz;
        const float dr = dz + (dres ? dres[base + k] : 0.f);
        d_res_pre[base + k] = dr * rv * (1.f - rv);
    }
}
// d_ln_w[j] = sum_n go[n,j] xhat[n,j], d_ln_b[j] = sum_n go[n,j].  grid ceil(D/64); lanes = columns, 16 waves = row phases
// (the rows of a wave are a serial chain of dependent-latency loads: 16 waves instead of 4 quarter that chain).
__global__ __launch_bounds__(1024) void cfl_pie_epi_bwd_ln_kernel(const float* __restrict__ go, const float* __restrict__ out,
                                                                  const float* __restrict__ r, const float* __restrict__ stats,
                                                                  int N, int D, float* d_lw, float* d_lb) {
    __shared__ float sw[16][64], sb[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    float aw = 0.f, ab = 0.f;
    if (j < D) {
#pragma unroll 4
        for (int n = w; n < N; n += 16) {
            const long long e = (long long)n * D + j;
            const float xh = (out[e] + r[e] - stats[n * 4 + 0]) * stats[n * 4 + 1];
            const float g = go[e];
            aw = fmaf(g, xh, aw); ab += g;
        }
    }
    sw[w][lane] = aw; sb[w][lane] = ab;
    __syncthreads();
    if (w == 0 && j < D) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a += sw[i][lane]; b += sb[i][lane]; }
        d_lw[j] = a;
        d_lb[j] = b;
    }
}

__global__ __launch_bounds__(256) void cfl_l2norm_fwd_kernel(const float* __restrict__ x, int N, int D, float* y, float* inv_norm) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const long long base = (long long)n * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s = fmaf(x[base + k], x[base + k], s);
    const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int k = lane; k < D; k += 64) y[base + k] = x[base + k] * inv;
    if (lane == 0 && inv_norm) inv_norm[n] = inv;
}
__global__ __launch_bounds__(256) void cfl_l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ inv_norm, int N, int D, float* dx) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const long long base = (long long)n * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s = fmaf(dy[base + k], y[base + k], s);
    s = wave_sum(s);
    const float inv = inv_norm[n];
    for (int k = lane; k < D; k += 64) dx[base + k] = inv * (dy[base + k] - y[base + k] * s);
}

}  // namespace

extern "C" {

size_t cfl_pie_ws_bytes(int N, int P, int Cd, int dh) {
    if (N <= 0 || P <= 0) return 256;
    const size_t rows = (size_t)N * P;
    const size_t nch = (rows + PIE_RC - 1) / PIE_RC;
    const size_t npart = nch > (size_t)N ? nch : (size_t)N;     // pie.hip: partials per row chunk; pie_fused.hip: per sample
    size_t a = rows + npart * (size_t)(dh > 0 ? dh : 1);      // pool: scores/ds + dw2 partials
    size_t b = (size_t)N * (size_t)(Cd > 0 ? Cd : 1);         // epilogue: go [N, D] (call with Cd = D)
    return cfl_align256((a > b ? a : b) * sizeof(float));
}

int cfl_pie_pool_fwd(const float* X, const float* H, const float* w2, const unsigned char* mask,
                     int N, int P, int Cd, int dh, float* attn, float* pooled, float* xmean, void* ws,
                     void* stream_) {
    if (!X || !H || !w2 || !attn || !pooled || !ws || N <= 0 || P <= 0 || Cd <= 0 || dh <= 0) return CFL_EINVAL;
    if (P > 1024) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const long long rows = (long long)N * P;
    float* scores = (float*)ws;
    const int vh = cfl_vec_ok(H, dh) && cfl_vec_ok(w2, 4);
    CFL_LAUNCH(K_PIE_SCORES, cfl_pie_scores_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream,
               H, w2, mask, rows, dh, vh, scores);
    const int vx = cfl_vec_ok(X, Cd) && cfl_vec_ok(pooled, Cd) && (!xmean || cfl_vec_ok(xmean, Cd));
    CFL_LAUNCH(K_PIE_POOL, cfl_pie_pool_kernel, dim3(N, cfl_cdiv(Cd, 1024)), dim3(256), 0, stream,
               X, scores, P, Cd, vx, attn, pooled, xmean);
    return 0;
}

int cfl_pie_pool_bwd(const float* X, const float* H, const float* w2, const unsigned char* mask,
                     const float* attn, const float* d_pooled, const float* d_xmean,
                     int N, int P, int Cd, int dh, float* dX, float* dH, float* dw2, void* ws, void* stream_) {
    (void)mask;
    if (!X || !H || !w2 || !attn || !d_pooled || !dX || !dH || !dw2 || !ws || N <= 0 || P <= 0 || Cd <= 0 || dh <= 0)
        return CFL_EINVAL;
    if (P > 1024) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const long long rows = (long long)N * P;
    const int nch = (int)((rows + PIE_RC - 1) / PIE_RC);
    float* ds = (float*)ws;
    float* partial = ds + rows;
    const int vx = cfl_vec_ok(X, Cd) && cfl_vec_ok(d_pooled, Cd) && cfl_vec_ok(dX, Cd) && (!d_xmean || cfl_vec_ok(d_xmean, Cd));
    CFL_LAUNCH(K_PIE_BWD_DS, cfl_pie_bwd_ds_kernel, dim3(N), dim3(1024), 0, stream, X, attn, d_pooled, P, Cd, vx, ds);
    CFL_LAUNCH(K_PIE_BWD_DX, cfl_pie_bwd_dx_kernel, dim3((unsigned)rows, cfl_cdiv(Cd, 1024)), dim3(256), 0, stream,
               attn, d_pooled, d_xmean, P, Cd, vx, dX);
    const int vh = cfl_vec_ok(H, dh) && cfl_vec_ok(w2, 4) && cfl_vec_ok(dH, dh);
    CFL_LAUNCH(K_PIE_BWD_DH, cfl_pie_bwd_dh_kernel, dim3(nch, cfl_cdiv(dh, 1024)), dim3(256), 0, stream,
               H, w2, ds, rows, dh, vh, dH, partial);
    CFL_LAUNCH(K_PIE_BWD_DW2, cfl_pie_bwd_dw2_kernel, dim3(cfl_cdiv(dh, 64)), dim3(256), 0, stream, partial, nch, dh, dw2);
    return 0;
}

int cfl_pie_epilogue_fwd(const float* out, const float* res_pre, const float* ln_w, const float* ln_b,
                         int N, int D, float ln_eps, int flags, float* y, float* o, float* r,
                         float* stats, void* stream_) {
    if (!out || !res_pre || !ln_w || !ln_b || !y || !stats || N <= 0 || D <= 0) return CFL_EINVAL;
    if (D > 4096) return CFL_ELIMIT;
    CFL_LAUNCH(K_PIE_EPI_FWD, cfl_pie_epi_fwd_kernel, dim3(cfl_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream_,
               out, res_pre, ln_w, ln_b, N, D, ln_eps, flags, y, o, r, stats);
    return 0;
}

int cfl_pie_epilogue_bwd(const float* dy, const float* do_, const float* dres, const float* out,
                         const float* r, const float* ln_w, const float* ln_b, const float* stats,
                         int N, int D, int flags, float* d_out, float* d_res_pre, float* d_ln_w,
                         float* d_ln_b, void* ws, void* stream_) {
    if (!dy || !out || !r || !ln_w || !ln_b || !stats || !d_out || !d_res_pre || !d_ln_w || !d_ln_b || !ws || N <= 0 || D <= 0)
        return CFL_EINVAL;
    if (D > 4096) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    float* go = (float*)ws;
    CFL_LAUNCH(K_PIE_EPI_BWD, cfl_pie_epi_bwd_kernel, dim3(cfl_cdiv(N, 4)), dim3(256), 0, stream,
               dy, do_, dres, out, r, ln_w, ln_b, stats, N, D, flags, d_out, d_res_pre, go);
    CFL_LAUNCH(K_PIE_EPI_BWD_LN, cfl_pie_epi_bwd_ln_kernel, dim3(cfl_cdiv(D, 64)), dim3(1024), 0, stream,
               go, out, r, stats, N, D, d_ln_w, d_ln_b);
    return 0;
}

int cfl_l2norm_fwd(const float* x, int N, int D, float* y, float* inv_norm, void* stream_) {
    if (!x || !y || N <= 0 || D <= 0) return CFL_EINVAL;
    CFL_LAUNCH(K_L2NORM_FWD, cfl_l2norm_fwd_kernel, dim3(cfl_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream_, x, N, D, y, inv_norm);
    return 0;
}
int cfl_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, int N, int D, float* dx, void* stream_) {
    if (!dy || !y || !inv_norm || !dx || N <= 0 || D <= 0) return CFL_EINVAL;
    CFL_LAUNCH(K_L2NORM_BWD, cfl_l2norm_bwd_kernel, dim3(cfl_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream_, dy, y, inv_norm, N, D, dx);
    return 0;
}

}  // extern "C"
