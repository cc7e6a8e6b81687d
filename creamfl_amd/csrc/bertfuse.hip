// bertfuse.hip -- the element-wise glue of the BERT text tower (row A2 of SURVEY section 8: the reference builds
// `BertModel.from_pretrained('bert-base-uncased')` at src/networks/models/pcme.py:36-38 and reads its [CLS] state at
// :44).  The layer arithmetic is the third-party `transformers` BertLayer (not vendored in the reference; restated
// from its published definition):
//     attn.output : x1 = LayerNorm(dropout(dense(ctx)) + x)                      (BertSelfOutput)
//     intermediate: h  = gelu(dense(x1))                                          (BertIntermediate, exact erf GELU)
//     output      : x2 = LayerNorm(dropout(dense(h)) + x1)                        (BertOutput)
// The GEMMs stay on hipBLASLt.  Everything between them is HBM streaming over tiny tensors (6144 x 768 bf16 =
// 9.4 MB at the bench shape) where the eager path is launch-bound: per layer ~25 kernels (bias-grad reductions,
// dropout, add, bf16<->fp32 casts around LayerNorm, LayerNorm, GELU) -- 5 ms of the 65 ms step in the rocprofv3
// trace.  Here:
//   daln  : bias + dropout + residual add + LayerNorm in one pass; the backward also produces the column sums
//           (dgamma, dbeta, dbias) and adds the TWO incoming gradients of its output (linear branch + residual
//           branch), so the autograd add kernels disappear too.
//   bgelu : bias + GELU; the backward produces dbias.
// Dropout masks are never stored: keep(e) is a counter hash of (seed, element index), recomputed in the backward.
#include "common.h"
#include "colmap.h"

namespace {

struct __attribute__((aligned(8))) U2 { u32 x, y; };            // 4 bf16

__device__ __forceinline__ void unpack4(const U2& u, float (&f)[4]) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ U2 pack4(const float (&f)[4]) {
    U2 u;
    u.x = bf16_rne(f[0]) | (bf16_rne(f[1]) << 16);
    u.y = bf16_rne(f[2]) | (bf16_rne(f[3]) << 16);
    return u;
}
__device__ __forceinline__ float bf16_round(float f) { return __uint_as_float(bf16_rne(f) << 16); }

__device__ __forceinline__ u32 mix32(u32 seed, u32 i) {
    u32 h = (i * 0x9E3779B1u) ^ seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// Steps replayed from a HIP graph: the host's per-call seed is a constant of the graph, so the masks would repeat every step.  With a
// step tick on the device (cfl_set_dropout_tick; the caller increments it on the stream once per step) the effective seed of a launch is
// seed + tick * odd constant: forward and backward of one step read the same tick, successive replays different ones.
__device__ __forceinline__ u32 tick_seed(u32 seed, const u32* __restrict__ tick) {
    return tick ? seed + *tick * 0x85EBCA6Bu : seed;
}
// keep flags of the 4 consecutive elements starting at element index e (e % 4 == 0): 16 random bits per element
__device__ __forceinline__ void keep4(u32 seed, long long e, u32 thr16, bool (&k)[4]) {
    const u32 i = (u32)(e >> 1);
    const u32 s2 = seed + (u32)(e >> 33) * 0x632BE5ABu;
    const u32 h0 = mix32(s2, i), h1 = mix32(s2, i + 1);
    k[0] = (h0 & 0xffffu) >= thr16; k[1] = (h0 >> 16) >= thr16;
    k[2] = (h1 & 0xffffu) >= thr16; k[3] = (h1 >> 16) >= thr16;
}
template <int N>
__device__ __forceinline__ void load_bias(const void* bias, int bias_bf16, int col, float (&b)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (!bias) b[k] = 0.f;
        else if (bias_bf16) b[k] = __uint_as_float((u32)reinterpret_cast<const unsigned short*>(bias)[col + k] << 16);
        else b[k] = reinterpret_cast<const float*>(bias)[col + k];
    }
}
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float u) {
    return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
}

// ---- dropout + add + LayerNorm ----------------------------------------------------------------------------------
// One wave per row; lane l owns columns j*256 + 4l .. +3 (8-byte accesses, 512 contiguous bytes per wave instruction).
template <int NJ>
__global__ __launch_bounds__(256) void cfl_daln_fwd_kernel(const U2* __restrict__ g, const void* __restrict__ bias, int bias_bf16,
                                                           const U2* __restrict__ res, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int T, int H, float eps, u32 thr16,
                                                           float scale, u32 seed, U2* __restrict__ z, U2* __restrict__ s_out,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           const u32* __restrict__ tick) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= T) return;
    seed = tick_seed(seed, tick);
    const long long base = (long long)row * H;
    float s[NJ][4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = j * 256 + lane * 4;
        if (col < H) {
            float gv[4], rv[4], b[4];
            unpack4(g[(base + col) >> 2], gv);
            unpack4(res[(base + col) >> 2], rv);
            load_bias<4>(bias, bias_bf16, col, b);
            bool k[4];
            keep4(seed, base + col, thr16, k);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = k[t] ? (gv[t] + b[t]) * scale : 0.f;
                s[j][t] = bf16_round(v + rv[t]);          // the eager path holds dropout(dense) + residual in bf16
                sum += s[j][t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) s[j][t] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = j * 256 + lane * 4;
        if (col < H) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { const float d = s[j][t] - mean; sq = fmaf(d, d, sq); }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = j * 256 + lane * 4;
        if (col < H) {
            float o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = fmaf((s[j][t] - mean) * rstd, gamma[col + t], beta[col + t]);
            z[(base + col) >> 2] = pack4(o);
            if (s_out) s_out[(base + col) >> 2] = pack4(s[j]);
        }
    }
    if (lane == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// Backward: dz = dz_a + dz_b;  gk = dz*gamma;  ds = rstd*(gk - mean(gk) - xhat*mean(gk*xhat));  dy = keep ? ds*scale : 0.
// Column sums (dgamma = sum dz*xhat, dbeta = sum dz, dbias = sum dy) -> partials[block][3][H].
template <int NJ>
__global__ __launch_bounds__(256) void cfl_daln_bwd_kernel(const U2* __restrict__ s, const U2* __restrict__ dz_a,
                                                           const U2* __restrict__ dz_b, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, int T,
                                                           int H, u32 thr16, float scale, u32 seed, int rows_per_block,
                                                           U2* __restrict__ ds_out, U2* __restrict__ dy_out, float* __restrict__ partials,
                                                           const u32* __restrict__ tick, const U2* __restrict__ ds_add) {
    __shared__ float red[4][NJ * 256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    seed = tick_seed(seed, tick);
    float dgam[NJ][4], dbet[NJ][4], dbia[NJ][4], gam[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = j * 256 + lane * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            dgam[j][t] = 0.f; dbet[j][t] = 0.f; dbia[j][t] = 0.f;
            gam[j][t] = (col + t < H) ? gamma[col + t] : 0.f;
        }
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(T, r0 + rows_per_block);
    for (int row = r0 + wave; row < r1; row += 4) {
        const long long base = (long long)row * H;
        const float mu = mean[row], rs = rstd[row];
        float xh[NJ][4], dz[NJ][4];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = j * 256 + lane * 4;
            if (col < H) {
                float sv[4], da[4];
                unpack4(s[(base + col) >> 2], sv);
                unpack4(dz_a[(base + col) >> 2], da);
                if (dz_b) {
                    float db[4];
                    unpack4(dz_b[(base + col) >> 2], db);
#pragma unroll
                    for (int t = 0; t < 4; ++t) da[t] += db[t];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    xh[j][t] = (sv[t] - mu) * rs;
                    dz[j][t] = da[t];
                    const float gk = da[t] * gam[j][t];
                    c1 += gk;
                    c2 = fmaf(gk, xh[j][t], c2);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) { xh[j][t] = 0.f; dz[j][t] = 0.f; }
            }
        }
        c1 = wave_sum(c1) / (float)H;
        c2 = wave_sum(c2) / (float)H;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = j * 256 + lane * 4;
            if (col < H) {
                float d[4], dy[4], da[4] = {0.f, 0.f, 0.f, 0.f};
                bool k[4];
                keep4(seed, base + col, thr16, k);
                if (ds_add) unpack4(ds_add[(base + col) >> 2], da);      // pre-LN: the gradient that reaches s directly (the residual stream)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    d[t] = rs * (dz[j][t] * gam[j][t] - c1 - xh[j][t] * c2) + da[t];
                    dy[t] = k[t] ? d[t] * scale : 0.f;
                    dgam[j][t] = fmaf(dz[j][t], xh[j][t], dgam[j][t]);
                    dbet[j][t] += dz[j][t];
                    dbia[j][t] += dy[t];
                }
                ds_out[(base + col) >> 2] = pack4(d);
                if (dy_out) dy_out[(base + col) >> 2] = pack4(dy);
            }
        }
    }
    // cross-wave column reduction, one quantity at a time
    float* out = partials + (long long)blockIdx.x * 3 * H;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                red[wave][j * 256 + lane * 4 + t] = q == 0 ? dgam[j][t] : (q == 1 ? dbet[j][t] : dbia[j][t]);
        __syncthreads();
        for (int c = threadIdx.x; c < NJ * 256; c += 256)
            if (c < H) out[q * H + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// out[c] = sum_b partials[b][c]  (c in [0, n)): 16 partial groups x 16 columns per block, fixed order
__global__ __launch_bounds__(256) void cfl_colsum_final_kernel(const float* __restrict__ partials, int nblk, int n,
                                                               float* __restrict__ out0, int n0, void* __restrict__ out1,
                                                               int out1_bf16) {
    __shared__ float sm[16][16];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), grp = threadIdx.x >> 4;
    float a = 0.f;
    if (c < n)
        for (int i = grp; i < nblk; i += 16) a += partials[(long long)i * n + c];
    sm[grp][threadIdx.x & 15] = a;
    __syncthreads();
    if (grp == 0 && c < n) {
        a = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) a += sm[g2][threadIdx.x & 15];
        if (c < n0) { if (out0) out0[c] = a; }                       // fp32 outputs (dgamma | dbeta)
        else if (out1) {                                             // the bias gradient, in the parameter's dtype
            if (out1_bf16) reinterpret_cast<unsigned short*>(out1)[c - n0] = (unsigned short)bf16_rne(a);
            else reinterpret_cast<float*>(out1)[c - n0] = a;
        }
    }
}

// ---- bias + GELU ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfl_bias_gelu_fwd_kernel(const U4* __restrict__ g, const void* __restrict__ bias,
                                                                int bias_bf16, long long nchunks, int I, U4* __restrict__ h) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long long)gridDim.x * 256) {
        const int col = (int)((i * 8) % I);
        float v[8], b[8];
        unpack8(g[i], v);
        load_bias<8>(bias, bias_bf16, col, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = gelu_f(v[k] + b[k]);
        h[i] = pack8(v);
    }
}

__global__ __launch_bounds__(256) void cfl_bias_gelu_bwd_kernel(const U4* __restrict__ g, const void* __restrict__ bias,
                                                                int bias_bf16, const U4* __restrict__ dh, long long R, int C,
                                                                int rows_per_block, U4* __restrict__ du, float* __restrict__ pcol) {
    __shared__ float lds[2048];
    const Map m = make_map(C);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m.active) {
        float b[8];
        load_bias<8>(bias, bias_bf16, m.c0, b);
        const long long rb = (long long)blockIdx.x * rows_per_block;
        const long long re = min(R, rb + rows_per_block);
        const long long stride = (long long)m.rpp * (C >> 3);
        long long off = (rb + m.rsub) * (C >> 3) + (m.c0 >> 3);
#pragma unroll 2
        for (long long r = rb + m.rsub; r < re; r += m.rpp, off += stride) {
            float u[8], d[8];
            unpack8(g[off], u);
            unpack8(dh[off], d);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                d[k] *= gelu_grad(u[k] + b[k]);
                acc[k] += d[k];
            }
            du[off] = pack8(d);
        }
    }
    // column reduction across the row phases of the block -> pcol[blockIdx.x][c]
    const int w = m.tprb * 8;
    if (m.rsub < m.rpp) {
        const int col = (threadIdx.x % m.tprb) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) lds[m.rsub * w + col + k] = acc[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < w; c += 256) {
        float sa = 0.f;
        for (int r = 0; r < m.rpp; ++r) sa += lds[r * w + c];
        const int cg = blockIdx.y * 2048 + c;
        if (cg < C) pcol[(long long)blockIdx.x * C + cg] = sa;
    }
}

// test helper: the keep mask of n elements (n % 4 == 0)
__global__ __launch_bounds__(256) void cfl_dropout_mask_kernel(u32 seed, u32 thr16, long long n4, unsigned char* keep,
                                                               const u32* __restrict__ tick) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    seed = tick_seed(seed, tick);
    bool k[4];
    keep4(seed, i * 4, thr16, k);
#pragma unroll
    for (int t = 0; t < 4; ++t) keep[i * 4 + t] = k[t] ? 1 : 0;
}

inline u32 thr_of(float p) {
    if (!(p > 0.f)) return 0;
    const float t = p * 65536.f + 0.5f;
    return t >= 65535.f ? 65535u : (u32)t;
}
inline float scale_of(u32 thr16) { return 65536.f / (float)(65536u - thr16); }
inline int daln_rows_per_block(int T) {
    int rpb = cfl_cdiv(T, 256);                       // ~one block per CU; few partials
    rpb = cfl_cdiv(rpb, 4) * 4;
    return rpb < 4 ? 4 : rpb;
}
struct GPlan { int rows_per_block, nblk, gy; };
inline GPlan gelu_plan(long long R, int C) {
    GPlan p;
    const int tpr = C >> 3, tprb = tpr < 256 ? tpr : 256, rpp = 256 / tprb;
    p.gy = cfl_cdiv(tpr, 256);
    long long want = 1024 / p.gy;
    long long rpb = (R + want - 1) / want;
    const long long unit = (long long)rpp * 2;
    rpb = ((rpb + unit - 1) / unit) * unit;
    if (rpb < unit) rpb = unit;
    p.rows_per_block = (int)rpb;
    p.nblk = (int)((R + rpb - 1) / rpb);
    return p;
}

}  // namespace

static const unsigned* g_dropout_tick = nullptr;     // cfl_set_dropout_tick: device word added (x odd constant) to every dropout seed

extern "C" {

size_t cfl_daln_ws_bytes(int T, int H) {
    if (T <= 0 || H <= 0) return 256;
    return cfl_align256((size_t)cfl_cdiv(T, daln_rows_per_block(T)) * 3 * H * sizeof(float));
}

int cfl_daln_fwd(const void* g, const void* bias, int bias_bf16, const void* residual, const float* gamma, const float* beta,
                 int T, int H, float eps, float p, unsigned seed, void* z, void* s, float* mean, float* rstd, void* stream_) {
    if (!g || !residual || !gamma || !beta || !z || T <= 0 || H <= 0 || (mean && !rstd)) return CFL_EINVAL;
    if (H % 4 != 0 || H > 2048 || p < 0.f || p >= 1.f) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const u32 thr = thr_of(p);
    const float sc = scale_of(thr);
    const dim3 grid(cfl_cdiv(T, 4));
#define DALN_FWD(NJ) CFL_LAUNCH(K_BERT_DALN, (cfl_daln_fwd_kernel<NJ>), grid, dim3(256), 0, stream, (const U2*)g, bias, bias_bf16, \
                                (const U2*)residual, gamma, beta, T, H, eps, thr, sc, seed, (U2*)z, (U2*)s, mean, rstd, g_dropout_tick)
    const int nj = cfl_cdiv(H, 256);
    if (nj <= 1) DALN_FWD(1); else if (nj <= 2) DALN_FWD(2); else if (nj <= 3) DALN_FWD(3); else if (nj <= 4) DALN_FWD(4); else DALN_FWD(8);
#undef DALN_FWD
    return 0;
}

int cfl_daln_bwd(const void* s, const void* dz_a, const void* dz_b, const float* gamma, const float* mean, const float* rstd,
                 int T, int H, float p, unsigned seed, void* ds, void* dy, float* dgamma_dbeta, void* dbias, int dbias_bf16,
                 void* ws, void* stream_) {
    if (!s || !dz_a || !gamma || !mean || !rstd || !ds || !dgamma_dbeta || !ws || T <= 0 || H <= 0) return CFL_EINVAL;
    if (H % 4 != 0 || H > 2048 || p < 0.f || p >= 1.f) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const u32 thr = thr_of(p);
    const float sc = scale_of(thr);
    const int rpb = daln_rows_per_block(T), nblk = cfl_cdiv(T, rpb);
    float* partials = (float*)ws;
#define DALN_BWD(NJ) CFL_LAUNCH(K_BERT_DALN, (cfl_daln_bwd_kernel<NJ>), dim3(nblk), dim3(256), 0, stream, (const U2*)s, (const U2*)dz_a, \
                                (const U2*)dz_b, gamma, mean, rstd, T, H, thr, sc, seed, rpb, (U2*)ds, (U2*)dy, partials, g_dropout_tick, \
                                (const U2*)nullptr)
    const int nj = cfl_cdiv(H, 256);
    if (nj <= 1) DALN_BWD(1); else if (nj <= 2) DALN_BWD(2); else if (nj <= 3) DALN_BWD(3); else if (nj <= 4) DALN_BWD(4); else DALN_BWD(8);
#undef DALN_BWD
    CFL_LAUNCH(K_BERT_DALN, cfl_colsum_final_kernel, dim3(cfl_cdiv(3 * H, 16)), dim3(256), 0, stream, partials, nblk, 3 * H,
               dgamma_dbeta, 2 * H, dbias, dbias_bf16);
    return 0;
}

// Pre-LN blocks (the ViT trunk of configs[4]): s = g + bias + residual is the NEXT residual and z = LayerNorm(s) feeds the next GEMM, so
// two different tensors leave the forward (cfl_daln_fwd with p = 0 writes both) and the gradient of s is
//     ds = LayerNorm'(dz) + ds_direct        (ds_direct: what reaches s over the residual stream; NULL = none)
// = d/d g = d/d residual; dbias = its column sums.
int cfl_preln_bwd(const void* s, const void* dz, const void* ds_direct, const float* gamma, const float* mean, const float* rstd, int T,
                  int H, void* ds, float* dgamma_dbeta, void* dbias, int dbias_bf16, void* ws, void* stream_) {
    if (!s || !dz || !gamma || !mean || !rstd || !ds || !dgamma_dbeta || !ws || T <= 0 || H <= 0) return CFL_EINVAL;
    if (H % 4 != 0 || H > 2048) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const u32 thr = thr_of(0.f);
    const float sc = scale_of(thr);
    const int rpb = daln_rows_per_block(T), nblk = cfl_cdiv(T, rpb);
    float* partials = (float*)ws;
#define PRELN_BWD(NJ) CFL_LAUNCH(K_BERT_DALN, (cfl_daln_bwd_kernel<NJ>), dim3(nblk), dim3(256), 0, stream, (const U2*)s, (const U2*)dz, \
                                 (const U2*)nullptr, gamma, mean, rstd, T, H, thr, sc, 0u, rpb, (U2*)ds, (U2*)nullptr, partials,         \
                                 (const u32*)nullptr, (const U2*)ds_direct)
    const int nj = cfl_cdiv(H, 256);
    if (nj <= 1) PRELN_BWD(1); else if (nj <= 2) PRELN_BWD(2); else if (nj <= 3) PRELN_BWD(3); else if (nj <= 4) PRELN_BWD(4); else PRELN_BWD(8);
#undef PRELN_BWD
    CFL_LAUNCH(K_BERT_DALN, cfl_colsum_final_kernel, dim3(cfl_cdiv(3 * H, 16)), dim3(256), 0, stream, partials, nblk, 3 * H,
               dgamma_dbeta, 2 * H, dbias, dbias_bf16);
    return 0;
}

size_t cfl_bias_gelu_ws_bytes(long long T, int I) {
    if (T <= 0 || I <= 0) return 256;
    return cfl_align256((size_t)gelu_plan(T, I).nblk * I * sizeof(float));
}

int cfl_bias_gelu_fwd(const void* g, const void* bias, int bias_bf16, long long T, int I, void* h, void* stream_) {
    if (!g || !h || T <= 0 || I <= 0) return CFL_EINVAL;
    if (I % 8 != 0) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const long long nchunks = T * (I / 8);
    const long long blocks = (nchunks + 255) / 256;
    CFL_LAUNCH(K_BERT_GELU, cfl_bias_gelu_fwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
               (const U4*)g, bias, bias_bf16, nchunks, I, (U4*)h);
    return 0;
}

int cfl_bias_gelu_bwd(const void* g, const void* bias, int bias_bf16, const void* dh, long long T, int I, void* du, void* dbias,
                      int dbias_bf16, void* ws, void* stream_) {
    if (!g || !dh || !du || !ws || T <= 0 || I <= 0) return CFL_EINVAL;
    if (I % 8 != 0 || ((I >> 3) < 256 && 256 % (I >> 3) != 0)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const GPlan p = gelu_plan(T, I);
    float* pcol = (float*)ws;
    CFL_LAUNCH(K_BERT_GELU, cfl_bias_gelu_bwd_kernel, dim3(p.nblk, p.gy), dim3(256), 0, stream, (const U4*)g, bias, bias_bf16,
               (const U4*)dh, T, I, p.rows_per_block, (U4*)du, pcol);
    if (dbias)
        CFL_LAUNCH(K_BERT_GELU, cfl_colsum_final_kernel, dim3(cfl_cdiv(I, 16)), dim3(256), 0, stream, pcol, p.nblk, I,
                   (float*)nullptr, 0, dbias, dbias_bf16);
    return 0;
}

int cfl_set_dropout_tick(const unsigned* tick_dev) {
    g_dropout_tick = tick_dev;
    return 0;
}

int cfl_dropout_mask(unsigned seed, float p, long long n, unsigned char* keep, void* stream_) {
    if (!keep || n <= 0 || n % 4 != 0 || p < 0.f || p >= 1.f) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    CFL_LAUNCH(K_BERT_DALN, cfl_dropout_mask_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, seed, thr_of(p),
               n / 4, keep, g_dropout_tick);
    return 0;
}

}  // extern "C"
