// bnorm.hip -- fused training-mode BatchNorm (+ residual add) (+ ReLU) for NHWC bf16 activations.
//
// Where it sits: the ResNet trunk of the image tower (row A2 / A2c of SURVEY section 8; reference
// src/networks/models/image_encoder.py:27,55 -> torchvision Bottleneck: bn -> relu, bn3 -> += identity -> relu;
// src/networks/resnet_client.py:60-98).  Convolutions stay on MIOpen; the normalisation, the residual add and the
// activation between them are pure HBM streaming and are fused here: the rocprofv3 trace of the bench step showed
// 31 % of GPU time in MIOpen's BatchNorm kernels plus 12 % in separate bf16 add / clamp kernels.
//
// Layout: a channels_last activation [N, C, H, W] is the row-major matrix [R = N*H*W, C] in bf16.  A thread owns 8
// consecutive channels (one 16-byte access) for all its rows, so per-channel scale/shift live in registers and every
// wave access is lane-contiguous.  Statistics are accumulated in fp32.
//   fwd : stats  (read x)                      -> per-block partial sum / sum-of-squares [blocks, C]
//         final  mean, invstd, running stats
//         apply  y = relu?( (x-mean)*invstd*gamma + beta (+ residual) )     (read x (+res), write y)
//   bwd : reduce (read dy, x, y)               -> partial sum(dy'), sum(dy' * xhat),  dy' = dy * (y > 0) if relu
//         (without a residual the mask is recomputed from x and y is not read; dy may arrive as two tensors)
//         final  dgamma, dbeta
//         apply  dx = gamma*invstd*(dy' - mean(dy') - xhat*mean(dy'*xhat));  dres = dy'   (write dx (+ dres))
#include "common.h"
#include "colmap.h"

namespace {

template <class G>
__global__ __launch_bounds__(256) void cfl_bn_stats_kernel(const G* __restrict__ x, long long R, int C, int rows_per_block,
                                                           float* psum, float* psq) {
    __shared__ float lds[4096];
    const Map m = make_map(C);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long rb = (long long)blockIdx.x * rows_per_block;
    const long long re = min(R, rb + rows_per_block);
    if (m.active) {
        const long long stride = (long long)m.rpp * (C >> 3);
        const G* p = x + (rb + m.rsub) * (C >> 3) + (m.c0 >> 3);
        long long r = rb + m.rsub;
        for (; r + 3 * m.rpp < re; r += 4 * m.rpp, p += 4 * stride) {
            const G u0 = p[0], u1 = p[stride], u2 = p[2 * stride], u3 = p[3 * stride];
            float f[8];
            unpack8(u0, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
            unpack8(u1, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
            unpack8(u2, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
            unpack8(u3, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
        }
        for (; r < re; r += m.rpp, p += stride) {
            float f[8];
            unpack8(p[0], f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
        }
    }
    block_col_reduce(m, s, q, C, psum, psq, lds);
}

// mean / invstd (+ running statistics) from the partials
constexpr int BN_FG = 64;              // partial groups of the two final kernels (1024 threads)
__global__ __launch_bounds__(1024) void cfl_bn_final_kernel(const float* __restrict__ psum, const float* __restrict__ psq,
                                                           int nblk, int C, long long R, float eps, float momentum,
                                                           float* mean, float* invstd, float* rmean, float* rvar) {
    __shared__ float sa[BN_FG][16], sb[BN_FG][16];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), grp = threadIdx.x >> 4;
    float a, b;
    reduce_partials<BN_FG>(psum, psq, nblk, C, c, grp, a, b, sa, sb);
    if (grp == 0 && c < C) {
        const float mu = a / (float)R;
        const float var = fmaxf(b / (float)R - mu * mu, 0.f);
        mean[c] = mu;
        invstd[c] = rsqrtf(var + eps);
        if (rmean) {
            const float unb = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
        }
    }
}

template <class G, bool RES, bool RELU>
__global__ __launch_bounds__(256) void cfl_bn_apply_kernel(const G* __restrict__ x, const G* __restrict__ res,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           long long R, int C, int rows_per_block, G* y,
                                                           unsigned char* __restrict__ relu_mask) {
    const Map m = make_map(C);
    if (!m.active) return;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float g = gamma[m.c0 + k] * invstd[m.c0 + k];
        sc[k] = g;
        sh[k] = beta[m.c0 + k] - mean[m.c0 + k] * g;
    }
    const long long rb = (long long)blockIdx.x * rows_per_block;
    const long long re = min(R, rb + rows_per_block);
    const long long stride = (long long)m.rpp * (C >> 3);
    long long off = (rb + m.rsub) * (C >> 3) + (m.c0 >> 3);
#pragma unroll 4
    for (long long r = rb + m.rsub; r < re; r += m.rpp, off += stride) {
        float f[8], g[8];
        unpack8(x[off], f);
        if (RES) unpack8(res[off], g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = fmaf(f[k], sc[k], sh[k]);
            if (RES) v += g[k];
            if (RELU) v = fmaxf(v, 0.f);
            f[k] = v;
        }
        const G o = pack8g<G>(f);
        y[off] = o;
        if (RELU && relu_mask) {                           // bit k = (y_k > 0) of the STORED bf16 value: the backward reads
            relu_mask[off] = (unsigned char)relu_bits(o);
        }
    }
}

// ---- stem tail: BatchNorm + ReLU + 3x3 / stride 2 / pad 1 max pooling in one pass, and its backward without the pooled
// gradient ever being scattered to memory (ResNet.bn1 / relu / maxpool inside image_encoder.py:27-36).  Unfused, the stem
// BatchNorm writes its [N,112,112,64] output (411 MB at batch 256) only for the pool to read it once, and the pool's backward
// writes a gradient of the same size that the two BatchNorm backward passes read once each: 2 GB of traffic per step.
//   forward : y_pool = maxpool(bf16(relu(x * sc + sh)))  -- the taps are normalised on the fly (rounded to bf16 exactly as the
//             stored activation would have been, so values, arg-max taps and ties are bit-identical to the unfused path)
//   backward: the BatchNorm reduce / apply passes take dy(n,h,w,:) from pool_grad8 (gather over the <= 4 windows that cover
//             the pixel, rounded to bf16 like the tensor the pool backward would have written) instead of loading it.
struct __attribute__((aligned(8))) B8 { unsigned int lo, hi; };          // 8 tap indices (same record as pool.hip)

template <class G>
__global__ __launch_bounds__(256) void cfl_bn_pool_fwd_kernel(const G* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int N, int H, int W, int C8, int Ho,
                                                              int Wo, G* __restrict__ y, B8* __restrict__ idx) {
    const long long total = (long long)N * Ho * Wo * C8;
    const long long i = (long long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C8);
    long long p = i / C8;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float g = gamma[c * 8 + k] * invstd[c * 8 + k];
        sc[k] = g;
        sh[k] = beta[c * 8 + k] - mean[c * 8 + k] * g;
    }
    float best[8];
    unsigned int tap[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; tap[k] = 0; }
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
        const int ih = 2 * oh - 1 + dh;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const int iw = 2 * ow - 1 + dw;
            if (iw < 0 || iw >= W) continue;
            float v[8];
            unpack8(x[(((long long)n * H + ih) * W + iw) * C8 + c], v);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(fmaf(v[k], sc[k], sh[k]), 0.f);
            unpack8(pack8g<G>(v), v);                                    // the value the unfused path stores and re-reads (bf16 rounding)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; tap[k] = dh * 3 + dw; }
        }
    }
    y[i] = pack8g<G>(best);
    B8 t;
    t.lo = tap[0] | (tap[1] << 8) | (tap[2] << 16) | (tap[3] << 24);
    t.hi = tap[4] | (tap[5] << 8) | (tap[6] << 16) | (tap[7] << 24);
    idx[i] = t;
}

// gradient of the pooling w.r.t. input pixel (n, h, w), channels 8 c .. 8 c + 7 (what cfl_maxpool_bwd_kernel writes there)
template <class G>
__device__ __forceinline__ void pool_grad8(const G* __restrict__ g, const B8* __restrict__ idx, int n, int h, int w, int c, int C8,
                                           int Ho, int Wo, float (&d)[8]) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int oh0 = h >> 1, oh1 = (h + 1) >> 1, ow0 = w >> 1, ow1 = (w + 1) >> 1;
    B8 tp[4];
    G gv[4];
    unsigned int want[4];
    bool ok[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int oh = a ? oh1 : oh0, ow = b ? ow1 : ow0;
            ok[a * 2 + b] = oh < Ho && ow < Wo && (a == 0 || oh1 != oh0) && (b == 0 || ow1 != ow0);
            const int ohc = oh < Ho ? oh : Ho - 1, owc = ow < Wo ? ow : Wo - 1;
            want[a * 2 + b] = (unsigned int)(h - (2 * oh - 1)) * 3 + (unsigned int)(w - (2 * ow - 1));
            const long long o = (((long long)n * Ho + ohc) * Wo + owc) * C8 + c;
            tp[a * 2 + b] = idx[o];
            gv[a * 2 + b] = g[o];
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!ok[q]) continue;
        float gg[8];
        unpack8(gv[q], gg);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned int tk = ((k < 4 ? tp[q].lo : tp[q].hi) >> (8 * (k & 3))) & 0xffu;
            if (tk == want[q]) acc[k] += gg[k];
        }
    }
    unpack8(pack8g<G>(acc), d);
}

// XMASK: the ReLU mask is recomputed from x (y is not read) -- only without a residual, where y = relu(x*sc + sh) with
// exactly the forward's sc/sh, so (x*sc + sh > 0) == (y > 0).  dy2 (may be NULL) is a second upstream gradient that is
// added on the fly: the block output feeds the next convolution AND the next residual add, and autograd would
// otherwise spend a separate read-read-write kernel on summing the two gradients.
template <class G, bool RELU, bool XMASK, bool POOL = false>
__global__ __launch_bounds__(256) void cfl_bn_bwd_reduce_kernel(const G* __restrict__ dy, const G* __restrict__ dy2,
                                                                const G* __restrict__ x, const G* __restrict__ y,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                long long R, int C, int rows_per_block, float* pdb, float* pdg,
                                                                const unsigned char* __restrict__ relu_mask,
                                                                const B8* __restrict__ pidx = nullptr, int PH = 0, int PW = 0) {
    __shared__ float lds[4096];
    const Map m = make_map(C);
    float db[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m.active) {
        float mu[8], is[8], sc[8], sh[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mu[k] = mean[m.c0 + k]; is[k] = invstd[m.c0 + k];
            if (XMASK) { sc[k] = gamma[m.c0 + k] * is[k]; sh[k] = beta[m.c0 + k] - mu[k] * sc[k]; }
        }
        const long long rb = (long long)blockIdx.x * rows_per_block;
        const long long re = min(R, rb + rows_per_block);
        const long long stride = (long long)m.rpp * (C >> 3);
        long long off = (rb + m.rsub) * (C >> 3) + (m.c0 >> 3);
#pragma unroll 2
        for (long long r = rb + m.rsub; r < re; r += m.rpp, off += stride) {
            float d[8], f[8], o[8];
            if constexpr (POOL) {                                        // dy = the pooling's gradient at this pixel, never stored
                const int pw = (int)(r % PW);
                const long long t_ = r / PW;
                pool_grad8(dy, pidx, (int)(t_ / PH), (int)(t_ % PH), pw, m.c0 >> 3, C >> 3, (PH - 1) / 2 + 1, (PW - 1) / 2 + 1, d);
            } else {
                unpack8(dy[off], d);
            }
            unpack8(x[off], f);
            unsigned mb = 0;
            if (RELU && !XMASK) {
                if (relu_mask) mb = relu_mask[off];
                else unpack8(y[off], o);
            }
            if (dy2) {
                float d2[8];
                unpack8(dy2[off], d2);
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] += d2[k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool on = !RELU || (XMASK ? fmaf(f[k], sc[k], sh[k]) > 0.f : (relu_mask ? ((mb >> k) & 1u) != 0 : o[k] > 0.f));
                const float dd = on ? d[k] : 0.f;
                db[k] += dd;
                dg[k] = fmaf(dd, (f[k] - mu[k]) * is[k], dg[k]);
            }
        }
    }
    block_col_reduce(m, db, dg, C, pdb, pdg, lds);
}

__global__ __launch_bounds__(1024) void cfl_bn_bwd_final_kernel(const float* __restrict__ pdb, const float* __restrict__ pdg,
                                                               int nblk, int C, float* dbeta, float* dgamma) {
    __shared__ float sa[BN_FG][16], sb[BN_FG][16];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), grp = threadIdx.x >> 4;
    float a, b;
    reduce_partials<BN_FG>(pdb, pdg, nblk, C, c, grp, a, b, sa, sb);
    if (grp == 0 && c < C) { dbeta[c] = a; dgamma[c] = b; }
}

template <class G, bool RES, bool RELU, bool XMASK, bool POOL = false>
__global__ __launch_bounds__(256) void cfl_bn_bwd_apply_kernel(const G* __restrict__ dy, const G* __restrict__ dy2,
                                                               const G* __restrict__ x, const G* __restrict__ y,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                               long long R, int C, int rows_per_block, G* dx, G* dres,
                                                               const unsigned char* __restrict__ relu_mask,
                                                               const B8* __restrict__ pidx = nullptr, int PH = 0, int PW = 0) {
    const Map m = make_map(C);
    if (!m.active) return;
    float mu[8], is[8], a[8], b[8], c[8], sh[8];
    const float invR = 1.f / (float)R;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[m.c0 + k]; is[k] = invstd[m.c0 + k];
        a[k] = gamma[m.c0 + k] * is[k];                 // dx = a * (dy' - b - xhat * c); also the forward's scale
        b[k] = dbeta[m.c0 + k] * invR;
        c[k] = dgamma[m.c0 + k] * invR;
        if (XMASK) sh[k] = beta[m.c0 + k] - mu[k] * a[k];
    }
    const long long rb = (long long)blockIdx.x * rows_per_block;
    const long long re = min(R, rb + rows_per_block);
    const long long stride = (long long)m.rpp * (C >> 3);
    long long off = (rb + m.rsub) * (C >> 3) + (m.c0 >> 3);
#pragma unroll 2
    for (long long r = rb + m.rsub; r < re; r += m.rpp, off += stride) {
        float d[8], f[8], o[8];
        if constexpr (POOL) {
            const int pw = (int)(r % PW);
            const long long t_ = r / PW;
            pool_grad8(dy, pidx, (int)(t_ / PH), (int)(t_ % PH), pw, m.c0 >> 3, C >> 3, (PH - 1) / 2 + 1, (PW - 1) / 2 + 1, d);
        } else {
            unpack8(ld_nt(dy + off), d);
        }
        unpack8(ld_nt(x + off), f);
        unsigned mb = 0;
        if (RELU && !XMASK) {
            if (relu_mask) mb = relu_mask[off];
            else unpack8(ld_nt(y + off), o);
        }
        if (dy2) {
            float d2[8];
            unpack8(ld_nt(dy2 + off), d2);
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] += d2[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool on = !RELU || (XMASK ? fmaf(f[k], a[k], sh[k]) > 0.f : (relu_mask ? ((mb >> k) & 1u) != 0 : o[k] > 0.f));
            const float dd = on ? d[k] : 0.f;
            d[k] = dd;
            f[k] = a[k] * (dd - b[k] - (f[k] - mu[k]) * is[k] * c[k]);
        }
        dx[off] = pack8g<G>(f);
        if (RES) dres[off] = pack8g<G>(d);
    }
}


// ---- channel-sliced block map (round 5) ------------------------------------------------------------------------------------
// A workgroup owns a 64-CHANNEL SLICE (128-byte row segments: 8 threads x 16 bytes) of a ROW RANGE -- in the statistics /
// reduce pass AND in the apply pass behind it.  The apply pass then needs only the few partial rows of its own slice
// (768 * 64 / C of them: 48 at C = 1024, 192 at C = 256), sums them in its prologue in a fixed order (every workgroup of the
// slice redundantly, bit-identically) and the `final` launch between the two passes disappears: 2 x 104 launches of
// 5 - 10 us per server step sat serially between the BatchNorm passes (step 41.7 vs 42.9 ms with all of them skipped).
// Streams as fast as whole rows (docs/history/tools/hip/slice_stream_probe.hip: 1 - 10 % faster at the trunk shapes).  Narrow tensors
// (C < 256: 384 / 768 partial rows per slice) keep the whole-row map above with its `final` kernels.
constexpr int SL_RL = 32;                                        // row lanes of a sliced workgroup (256 threads / 8)

// per-(part, slice) column sums of two per-thread 8-vectors over the 32 row lanes -> pa / pb [part][C]
__device__ __forceinline__ void slice_col_reduce(const float (&a)[8], const float (&b)[8], int C, float* pa, float* pb, float* lds) {
    const int cseg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    float* la = lds;                                             // [32][64]
    float* lb = lds + SL_RL * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) { la[rl * 64 + cseg * 8 + k] = a[k]; lb[rl * 64 + cseg * 8 + k] = b[k]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const float* l = threadIdx.x < 64 ? la : lb;
        const int c = threadIdx.x & 63;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int r = 0; r < SL_RL; r += 4) { s0 += l[r * 64 + c]; s1 += l[(r + 1) * 64 + c]; s2 += l[(r + 2) * 64 + c]; s3 += l[(r + 3) * 64 + c]; }
        (threadIdx.x < 64 ? pa : pb)[(long long)blockIdx.x * C + blockIdx.y * 64 + c] = (s0 + s1) + (s2 + s3);
    }
}

// totals of this workgroup's 64 channels over the `nparts` partial rows: four interleaved chains per channel, fixed order
__device__ __forceinline__ void slice_totals(const float* __restrict__ pa, const float* __restrict__ pb, int nparts, int C,
                                             float* lds, float (&ta)[8], float (&tb)[8]) {
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float* qa = pa + blockIdx.y * 64 + cl;
    const float* qb = pb + blockIdx.y * 64 + cl;
    float a = 0.f, b = 0.f;
    int q = g;
    for (; q + 12 < nparts; q += 16) {
        const float a0 = qa[(long long)q * C], a1 = qa[(long long)(q + 4) * C], a2 = qa[(long long)(q + 8) * C], a3 = qa[(long long)(q + 12) * C];
        const float b0 = qb[(long long)q * C], b1 = qb[(long long)(q + 4) * C], b2 = qb[(long long)(q + 8) * C], b3 = qb[(long long)(q + 12) * C];
        a += (a0 + a1) + (a2 + a3); b += (b0 + b1) + (b2 + b3);
    }
    for (; q < nparts; q += 4) { a += qa[(long long)q * C]; b += qb[(long long)q * C]; }
    lds[g * 64 + cl] = a; lds[256 + g * 64 + cl] = b;
    __syncthreads();
    const int c = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ta[k] = (lds[c + k] + lds[64 + c + k]) + (lds[128 + c + k] + lds[192 + c + k]);
        tb[k] = (lds[256 + c + k] + lds[320 + c + k]) + (lds[384 + c + k] + lds[448 + c + k]);
    }
}

template <class G>
__global__ __launch_bounds__(256) void cfl_bn_stats_sl_kernel(const G* __restrict__ x, long long R, int C, int rows_per_block,
                                                              float* psum, float* psq) {
    __shared__ float lds[2 * SL_RL * 64];
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long rb = (long long)blockIdx.x * rows_per_block;
    const long long re = min(R, rb + rows_per_block);
    const long long stride = (long long)SL_RL * (C >> 3);
    long long r = rb + (threadIdx.x >> 3);
    const G* p = x + r * (C >> 3) + blockIdx.y * 8 + (threadIdx.x & 7);
    for (; r + 3 * SL_RL < re; r += 4 * SL_RL, p += 4 * stride) {
        const G u0 = p[0], u1 = p[stride], u2 = p[2 * stride], u3 = p[3 * stride];
        float f[8];
        unpack8(u0, f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
        unpack8(u1, f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
        unpack8(u2, f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
        unpack8(u3, f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
    }
    for (; r < re; r += SL_RL, p += stride) {
        float f[8];
        unpack8(p[0], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] = fmaf(f[k], f[k], q[k]); }
    }
    slice_col_reduce(s, q, C, psum, psq, lds);
}

// apply pass of the forward on the sliced map; its prologue is the old `final` kernel for this slice (workgroup 0 of a slice
// also stores mean / invstd for the backward and updates the running statistics)
template <class G, bool RES, bool RELU>
__global__ __launch_bounds__(256) void cfl_bn_apply_sl_kernel(const G* __restrict__ x, const G* __restrict__ res,
                                                              const float* __restrict__ psum, const float* __restrict__ psq, int nparts,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              long long R, int C, int rows_per_block, float eps, float momentum, G* y,
                                                              unsigned char* __restrict__ relu_mask, float* mean, float* invstd,
                                                              float* rmean, float* rvar) {
    __shared__ float lds[512];
    float ta[8], tb[8], sc[8], sh[8];
    slice_totals(psum, psq, nparts, C, lds, ta, tb);
    const int c0 = blockIdx.y * 64 + (threadIdx.x & 7) * 8;
    const bool writer = blockIdx.x == 0 && (threadIdx.x >> 3) == 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float mu = ta[k] / (float)R;
        const float var = fmaxf(tb[k] / (float)R - mu * mu, 0.f);
        const float is = rsqrtf(var + eps);
        const float g = gamma[c0 + k] * is;
        sc[k] = g;
        sh[k] = beta[c0 + k] - mu * g;
        if (writer) {
            mean[c0 + k] = mu;
            invstd[c0 + k] = is;
            if (rmean) {
                const float unb = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
                rmean[c0 + k] = (1.f - momentum) * rmean[c0 + k] + momentum * mu;
                rvar[c0 + k] = (1.f - momentum) * rvar[c0 + k] + momentum * unb;
            }
        }
    }
    const long long rb = (long long)blockIdx.x * rows_per_block;
    const long long re = min(R, rb + rows_per_block);
    const long long stride = (long long)SL_RL * (C >> 3);
    long long off = (rb + (threadIdx.x >> 3)) * (C >> 3) + blockIdx.y * 8 + (threadIdx.x & 7);
#pragma unroll 4
    for (long long r = rb + (threadIdx.x >> 3); r < re; r += SL_RL, off += stride) {
        float f[8], g[8];
        unpack8(x[off], f);
        if (RES) unpack8(res[off], g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = fmaf(f[k], sc[k], sh[k]);
            if (RES) v += g[k];
            if (RELU) v = fmaxf(v, 0.f);
            f[k] = v;
        }
        const G o = pack8g<G>(f);
        y[off] = o;
        if (RELU && relu_mask) {
            relu_mask[off] = (unsigned char)relu_bits(o);
        }
    }
}

template <class G, bool RELU, bool XMASK>
__global__ __launch_bounds__(256) void cfl_bn_bwd_reduce_sl_kernel(const G* __restrict__ dy, const G* __restrict__ dy2,
                                                                   const G* __restrict__ x, const G* __restrict__ y,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   long long R, int C, int rows_per_block, float* pdb, float* pdg,
                                                                   const unsigned char* __restrict__ relu_mask) {
    __shared__ float lds[2 * SL_RL * 64];
    float db[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float mu[8], is[8], sc[8], sh[8];
    const int c0 = blockIdx.y * 64 + (threadIdx.x & 7) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k]; is[k] = invstd[c0 + k];
        if (XMASK) { sc[k] = gamma[c0 + k] * is[k]; sh[k] = beta[c0 + k] - mu[k] * sc[k]; }
    }
    const long long rb = (long long)blockIdx.x * rows_per_block;
    const long long re = min(R, rb + rows_per_block);
    const long long stride = (long long)SL_RL * (C >> 3);
    long long off = (rb + (threadIdx.x >> 3)) * (C >> 3) + blockIdx.y * 8 + (threadIdx.x & 7);
#pragma unroll 2
    for (long long r = rb + (threadIdx.x >> 3); r < re; r += SL_RL, off += stride) {
        float d[8], f[8], o[8];
        unpack8(dy[off], d);
        unpack8(x[off], f);
        unsigned mb = 0;
        if (RELU && !XMASK) {
            if (relu_mask) mb = relu_mask[off];
            else unpack8(y[off], o);
        }
        if (dy2) {
            float d2[8];
            unpack8(dy2[off], d2);
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] += d2[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool on = !RELU || (XMASK ? fmaf(f[k], sc[k], sh[k]) > 0.f : (relu_mask ? ((mb >> k) & 1u) != 0 : o[k] > 0.f));
            const float dd = on ? d[k] : 0.f;
            db[k] += dd;
            dg[k] = fmaf(dd, (f[k] - mu[k]) * is[k], dg[k]);
        }
    }
    slice_col_reduce(db, dg, C, pdb, pdg, lds);
}

template <class G, bool RES, bool RELU, bool XMASK>
__global__ __launch_bounds__(256) void cfl_bn_bwd_apply_sl_kernel(const G* __restrict__ dy, const G* __restrict__ dy2,
                                                                  const G* __restrict__ x, const G* __restrict__ y,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ pdb, const float* __restrict__ pdg, int nparts,
                                                                  float* dbeta, float* dgamma, long long R, int C, int rows_per_block,
                                                                  G* dx, G* dres, const unsigned char* __restrict__ relu_mask) {
    __shared__ float lds[512];
    float tdb[8], tdg[8];
    slice_totals(pdb, pdg, nparts, C, lds, tdb, tdg);
    const int c0 = blockIdx.y * 64 + (threadIdx.x & 7) * 8;
    const bool writer = blockIdx.x == 0 && (threadIdx.x >> 3) == 0;
    float mu[8], is[8], a[8], b[8], c[8], sh[8];
    const float invR = 1.f / (float)R;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k]; is[k] = invstd[c0 + k];
        a[k] = gamma[c0 + k] * is[k];
        b[k] = tdb[k] * invR;
        c[k] = tdg[k] * invR;
        if (XMASK) sh[k] = beta[c0 + k] - mu[k] * a[k];
        if (writer) { dbeta[c0 + k] = tdb[k]; dgamma[c0 + k] = tdg[k]; }
    }
    const long long rb = (long long)blockIdx.x * rows_per_block;
    const long long re = min(R, rb + rows_per_block);
    const long long stride = (long long)SL_RL * (C >> 3);
    long long off = (rb + (threadIdx.x >> 3)) * (C >> 3) + blockIdx.y * 8 + (threadIdx.x & 7);
#pragma unroll 2
    for (long long r = rb + (threadIdx.x >> 3); r < re; r += SL_RL, off += stride) {
        float d[8], f[8], o[8];
        unpack8(ld_nt(dy + off), d);
        unpack8(ld_nt(x + off), f);
        unsigned mb = 0;
        if (RELU && !XMASK) {
            if (relu_mask) mb = relu_mask[off];
            else unpack8(ld_nt(y + off), o);
        }
        if (dy2) {
            float d2[8];
            unpack8(ld_nt(dy2 + off), d2);
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] += d2[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool on = !RELU || (XMASK ? fmaf(f[k], a[k], sh[k]) > 0.f : (relu_mask ? ((mb >> k) & 1u) != 0 : o[k] > 0.f));
            const float dd = on ? d[k] : 0.f;
            d[k] = dd;
            f[k] = a[k] * (dd - b[k] - (f[k] - mu[k]) * is[k] * c[k]);
        }
        dx[off] = pack8g<G>(f);
        if (RES) dres[off] = pack8g<G>(d);
    }
}

struct SPlan { int rows_per_block, parts, slices; };
static int& bn_sliced_on() {
    static int on = getenv("CFL_BN_NO_SLICE") ? 0 : 1;
    return on;
}
// the sliced map takes C = 256 ... 2048 in whole 64-channel slices (<= 192 partial rows per slice)
static bool bn_use_sliced(long long R, int C) { return bn_sliced_on() && C >= 256 && C % 64 == 0 && C <= 2048 && R >= SL_RL; }
static SPlan bn_splan(long long R, int C) {
    SPlan p;
    p.slices = C / 64;
    long long want = 768 / p.slices;                      // ~3 workgroups per CU in all
    if (want < 1) want = 1;
    long long rpb = (R + want - 1) / want;
    rpb = ((rpb + SL_RL - 1) / SL_RL) * SL_RL;
    p.rows_per_block = (int)rpb;
    p.parts = (int)((R + rpb - 1) / rpb);
    return p;
}


// ---- conv3 weight gradient INSIDE the BatchNorm backward-apply pass that produces its dY (round 5) --------------------------
// A bottleneck's bn3 backward (pre-joined gradient: no mask, no second gradient; ops.JOIN) writes dY3 = the upstream gradient of
// conv3, a 1 x 1 convolution whose weight gradient dW3[c, k] = sum_r dY3[r, c] A2[r, k] the library computes on a side stream from
// dY3 read back from memory (CK batched GEMM with fp32 atomics + memset + cast: 63 us alone, ~114 us inside the step at 14 x 14,
// 1024 x 256 -- 4 of its 5 operand units are dY3).  Here the apply pass keeps the dY3 tile it has just computed: a workgroup
// (512 threads) owns 128 channels of a row range; per 64-row stage it streams its dy / x segments through registers, writes
// dY3 to memory AND -- bf16, row-major, 16-byte pieces XOR-swizzled by (row & 3) << 2 -- into an LDS tile, the stage's A2 rows
// [64 x 256] arrive by LDS-DMA (same swizzle, applied on the source side), and both MFMA operands are read TRANSPOSED out of LDS
// (ds_read_b64_tr_b16: 8 consecutive rows of one column per lane = the operand layout of v_mfma_f32_32x32x16_bf16).  8 waves =
// 2 channel halves x 4 column quarters, 64 x 64 accumulators per wave; one barrier per stage (the MFMAs of stage s run in the
// same phase as the arithmetic and LDS writes of stage s + 1, the DMA and register loads of stage s + 2 are issued behind the
// barrier).  Split-K partials [parts][C][256] fp32 (one per workgroup row range: 32 MB at 32 parts), a fixed-order reduce that
// casts: deterministic, unlike the library's atomics.  Stand-alone at R = 50 176, C = 1024 (tools/hip/bnwg_probe.hip): apply
// 79 us -> apply + weight gradient 95 us + 3.6 us reduce.  P = 256 only (layer3 of ResNet-50 / -101: 23 of 33 blocks).
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* wg_lds_vptr;
typedef const __attribute__((address_space(1))) void* wg_glb_vptr;
typedef unsigned short u16;

__device__ __attribute__((aligned(256))) unsigned char g_bn_zero_page[256];

constexpr int WG_CH = 128;                 // channels per workgroup
constexpr int WG_KS = 64;                  // rows per stage
constexpr int WG_PN = 256;                 // columns of the convolution's input (planes)
constexpr int WG_TILE_A = WG_KS * 256;     // 16 KB: [64 rows][128 channels] bf16
constexpr int WG_TILE_B = 2 * WG_KS * 256; // 32 KB: two [64][128] sub-tiles

__device__ __forceinline__ wg_bf16x8 wg_frag_tr(const char* p) {
    const wg_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)(p));
    const wg_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)(p + 4 * 256));
    union { wg_s16x4 h[2]; wg_bf16x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}

// rows m0 .. m0 + 63 of A [R, 256] into two [64][128] sub-tiles; rows >= mend come from a page of zeros.  32 wave instructions
// of 4 rows x 256 bytes, wave w of 8 issues 4.
__device__ __forceinline__ void wg_stage_b(const u16* __restrict__ A, long long m0, long long mend, char* dst, int w, int lane) {
    const int r4 = lane >> 4, pp = lane & 15;
    const int lp = pp ^ (r4 << 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ins = w + 8 * j;
        const int sub = ins >> 4, rg = ins & 15;
        const long long m = m0 + rg * 4 + r4;
        const void* p = m < mend ? (const void*)(A + m * WG_PN + sub * 128 + lp * 8) : (const void*)g_bn_zero_page;
        __builtin_amdgcn_global_load_lds((wg_glb_vptr)p, (wg_lds_vptr)(dst + sub * (WG_KS * 256) + rg * 1024), 16, 0, 0);
    }
}

// grid = (parts, C / 128), 512 threads, dynamic LDS = 2 WG_TILE_A + 2 WG_TILE_B + 4 KB.  pdb / pdg: the nparts_r partial rows
// the reduce pass (64-channel slices) left for every channel.
__global__ __launch_bounds__(512, 1) void cfl_bn_bwd_apply_wgrad_kernel(const U4* __restrict__ dy, const U4* __restrict__ x,
                                                                       const u16* __restrict__ A, const float* __restrict__ mean,
                                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                       const float* __restrict__ pdb, const float* __restrict__ pdg,
                                                                       int nparts_r, float* dbeta, float* dgamma, long long R, int C,
                                                                       int rows_per_block, U4* __restrict__ dx, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    char* ldsA = wlds;
    char* ldsB = wlds + 2 * WG_TILE_A;
    float* tot = reinterpret_cast<float*>(wlds + 2 * WG_TILE_A + 2 * WG_TILE_B);      // [2][4][128]
    const int seg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w & 1, wn = w >> 1;
    // totals of this workgroup's 128 channels over the reduce pass's partial rows: four interleaved chains per channel (the
    // order of slice_totals, so dbeta / dgamma are bit-identical to the plain sliced pass)
    {
        const int cl = threadIdx.x & 127, g = threadIdx.x >> 7;
        const float* qa = pdb + blockIdx.y * WG_CH + cl;
        const float* qb = pdg + blockIdx.y * WG_CH + cl;
        float ta = 0.f, tb = 0.f;
        int q = g;
        for (; q + 12 < nparts_r; q += 16) {
            const float a0 = qa[(long long)q * C], a1 = qa[(long long)(q + 4) * C], a2 = qa[(long long)(q + 8) * C], a3 = qa[(long long)(q + 12) * C];
            const float b0 = qb[(long long)q * C], b1 = qb[(long long)(q + 4) * C], b2 = qb[(long long)(q + 8) * C], b3 = qb[(long long)(q + 12) * C];
            ta += (a0 + a1) + (a2 + a3); tb += (b0 + b1) + (b2 + b3);
        }
        for (; q < nparts_r; q += 4) { ta += qa[(long long)q * C]; tb += qb[(long long)q * C]; }
        tot[g * 128 + cl] = ta; tot[512 + g * 128 + cl] = tb;
    }
    __syncthreads();
    const int c0 = blockIdx.y * WG_CH + seg * 8;
    const bool writer = blockIdx.x == 0 && rl == 0;
    float a[8], b[8], cc[8], mu[8], is[8];
    const float invR = 1.f / (float)R;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = seg * 8 + k;
        const float tdb = (tot[c] + tot[128 + c]) + (tot[256 + c] + tot[384 + c]);
        const float tdg = (tot[512 + c] + tot[640 + c]) + (tot[768 + c] + tot[896 + c]);
        mu[k] = mean[c0 + k]; is[k] = invstd[c0 + k];
        a[k] = gamma[c0 + k] * is[k];
        b[k] = tdb * invR;
        cc[k] = tdg * invR;
        if (writer) { dbeta[c0 + k] = tdb; dgamma[c0 + k] = tdg; }
    }
    const long long rb = (long long)blockIdx.x * rows_per_block, re = min(R, rb + rows_per_block);
    const int nst = (int)((re - rb + WG_KS - 1) / WG_KS);
    const int g = lane >> 4, p = lane & 15;
    int offA[2], offB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ca = wm * 64 + 32 * t + 16 * (g & 1) + 4 * (p & 3);
        const int cb = wn * 64 + 32 * t + 16 * (g & 1) + 4 * (p & 3);
        const int row = 8 * (g >> 1) + (p >> 2);
        offA[t] = row * 256 + ((((ca >> 3) ^ ((p >> 2) << 2)) & 15) << 4) + (ca & 7) * 2;
        const int cs = cb & 127;
        offB[t] = (cb >> 7) * (WG_KS * 256) + row * 256 + ((((cs >> 3) ^ ((p >> 2) << 2)) & 15) << 4) + (cs & 7) * 2;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const long long rowU4 = C >> 3;
    const long long gbase = blockIdx.y * 16 + seg;
    U4 rd[2], rx[2];
    auto load_regs = [&](int st) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r = rb + (long long)st * WG_KS + rl + 32 * h;
            const long long o = (r < re ? r : re - 1) * rowU4 + gbase;
            rd[h] = dy[o];
            rx[h] = x[o];
        }
    };
    auto compute_store = [&](int st, char* tile) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rr = rl + 32 * h;
            const long long r = rb + (long long)st * WG_KS + rr;
            float d[8], f[8];
            unpack8(rd[h], d);
            unpack8(rx[h], f);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = a[k] * (d[k] - b[k] - (f[k] - mu[k]) * is[k] * cc[k]);
            U4 o = pack8(f);
            if (r < re) dx[r * rowU4 + gbase] = o;
            else { o.x = 0; o.y = 0; o.z = 0; o.w = 0; }
            *reinterpret_cast<U4*>(tile + rr * 256 + ((seg ^ ((rr & 3) << 2)) << 4)) = o;
        }
    };
    wg_stage_b(A, rb, re, ldsB, w, lane);
    load_regs(0);
    compute_store(0, ldsA);
    if (nst > 1) { wg_stage_b(A, rb + WG_KS, re, ldsB + WG_TILE_B, w, lane); load_regs(1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const char* ta = ldsA + (st & 1) * WG_TILE_A;
        const char* tb = ldsB + (st & 1) * WG_TILE_B;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wg_bf16x8 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = wg_frag_tr(ta + offA[t] + ks * 16 * 256);
                fb[t] = wg_frag_tr(tb + offB[t] + ks * 16 * 256);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
        }
        if (st + 1 < nst) compute_store(st + 1, ldsA + ((st + 1) & 1) * WG_TILE_A);
        // everything issued one iteration ago has landed (the DMA of stage st + 1 included); the two dY stores just issued may
        // stay out.  The last two iterations drain everything: a ragged last stage issues fewer stores than the count assumes.
        if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 2 < nst) {
            wg_stage_b(A, rb + (long long)(st + 2) * WG_KS, re, ldsB + (st & 1) * WG_TILE_B, w, lane);
            load_regs(st + 2);
        }
    }
    float* out = part + ((long long)blockIdx.x * C + blockIdx.y * WG_CH) * WG_PN;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int j = (wn * 2 + n) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (wm * 2 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(long long)i * WG_PN + j] = acc[m][n][r];
            }
        }
}

// dW = sum over the row parts (fixed order), bf16
__global__ __launch_bounds__(256) void cfl_bn_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, long long n, u32* __restrict__ out) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= nparts; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (long long)(k + u) * n + i);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nparts; ++k) s += *reinterpret_cast<const f32x4*>(part + (long long)k * n + i);
    out[i / 2] = bf16_rne(s[0]) | (bf16_rne(s[1]) << 16);
    out[i / 2 + 1] = bf16_rne(s[2]) | (bf16_rne(s[3]) << 16);
}

struct WPlan { int rows_per_block, parts; };
static WPlan bn_wplan(long long R, int C) {
    WPlan p;
    long long want = 256 / (C / WG_CH);                   // one workgroup per CU (96 KB of LDS)
    if (want < 1) want = 1;
    long long rpb = (R + want - 1) / want;
    rpb = ((rpb + WG_KS - 1) / WG_KS) * WG_KS;
    p.rows_per_block = (int)rpb;
    p.parts = (int)((R + rpb - 1) / rpb);
    return p;
}

struct Plan { int rows_per_block, nblk, gy; };
static Plan bn_plan(long long R, int C) {
    Plan p;
    const int tpr = C >> 3;
    const int tprb = tpr < 256 ? tpr : 256;
    const int rpp = 256 / tprb;
    p.gy = cfl_cdiv(tpr, 256);
    long long want = 768 / p.gy;                          // ~3 blocks per CU; few partials keep the final kernels short
    long long rpb = (R + want - 1) / want;
    const long long unit = (long long)rpp * 4;            // keep the 4x unrolled loop busy
    rpb = ((rpb + unit - 1) / unit) * unit;
    if (rpb < unit) rpb = unit;
    p.rows_per_block = (int)rpb;
    p.nblk = (int)((R + rpb - 1) / rpb);
    return p;
}

template <class G>
static int bn_fwd_t(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
               float* running_var, long long R, int C, float eps, float momentum, int relu, void* y,
               float* save_mean, float* save_invstd, unsigned char* relu_mask, void* ws, void* stream_) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !ws || R <= 0 || C <= 0) return CFL_EINVAL;
    if (C % 8 != 0 || ((C >> 3) < 256 && 256 % (C >> 3) != 0)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    if (bn_use_sliced(R, C)) {
        // two launches: statistics and apply on the same 64-channel slices; the apply pass sums its slice's partials itself
        const SPlan sp = bn_splan(R, C);
        float* qsum = (float*)ws;
        float* qsq = qsum + (size_t)sp.parts * C;
        const dim3 sg(sp.parts, sp.slices);
        const G* xs = (const G*)x; const G* rs = (const G*)residual; G* ys = (G*)y;
        CFL_LAUNCH(K_BN_STATS, (cfl_bn_stats_sl_kernel<G>), sg, dim3(256), 0, stream, xs, R, C, sp.rows_per_block, qsum, qsq);
#define BN_APPLY_SL(RES_, RELU_) CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_sl_kernel<G, RES_, RELU_>), sg, dim3(256), 0, stream, xs, rs, qsum, qsq, \
                                            sp.parts, gamma, beta, R, C, sp.rows_per_block, eps, momentum, ys, relu_mask, save_mean,      \
                                            save_invstd, running_mean, running_var)
        if (residual && relu) BN_APPLY_SL(true, true);
        else if (residual) BN_APPLY_SL(true, false);
        else if (relu) BN_APPLY_SL(false, true);
        else BN_APPLY_SL(false, false);
#undef BN_APPLY_SL
        return 0;
    }
    const Plan p = bn_plan(R, C);
    float* psum = (float*)ws;
    float* psq = psum + (size_t)p.nblk * C;
    const dim3 grid(p.nblk, p.gy);
    CFL_LAUNCH(K_BN_STATS, (cfl_bn_stats_kernel<G>), grid, dim3(256), 0, stream, (const G*)x, R, C, p.rows_per_block, psum, psq);
    CFL_LAUNCH(K_BN_FINAL, cfl_bn_final_kernel, dim3(cfl_cdiv(C, 16)), dim3(16 * BN_FG), 0, stream, psum, psq, p.nblk, C, R, eps,
               momentum, save_mean, save_invstd, running_mean, running_var);
    const G* xr = (const G*)x; const G* rr = (const G*)residual; G* yy = (G*)y;
    if (residual && relu)
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<G, true, true>), grid, dim3(256), 0, stream, xr, rr, save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    else if (residual)
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<G, true, false>), grid, dim3(256), 0, stream, xr, rr, save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    else if (relu)
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<G, false, true>), grid, dim3(256), 0, stream, xr, rr, save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    else
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<G, false, false>), grid, dim3(256), 0, stream, xr, rr, save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    return 0;
}

template <class G>
static int bn_apply_t(const void* x, const void* residual, const float* mean, const float* invstd, const float* gamma,
                 const float* beta, long long R, int C, int relu, void* y, void* stream_) {
    unsigned char* relu_mask = nullptr;
    if (!x || !mean || !invstd || !gamma || !beta || !y || R <= 0 || C <= 0) return CFL_EINVAL;
    if (C % 8 != 0 || ((C >> 3) < 256 && 256 % (C >> 3) != 0)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const Plan p = bn_plan(R, C);
    const dim3 grid(p.nblk, p.gy);
    const G* xr = (const G*)x; const G* rr = (const G*)residual; G* yy = (G*)y;
    if (residual && relu)
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<G, true, true>), grid, dim3(256), 0, stream, xr, rr, mean, invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    else if (residual)
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<G, true, false>), grid, dim3(256), 0, stream, xr, rr, mean, invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    else if (relu)
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<G, false, true>), grid, dim3(256), 0, stream, xr, rr, mean, invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    else
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<G, false, false>), grid, dim3(256), 0, stream, xr, rr, mean, invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    return 0;
}

template <class G>
static int bn_bwd_t(const void* dy, const void* dy2, const void* x, const void* y, const unsigned char* relu_mask, const float* gamma,
               const float* beta, const float* save_mean, const float* save_invstd, long long R, int C, int relu, int has_residual,
               void* dx, void* dres, float* dgamma, float* dbeta, void* ws, void* stream_) {
    if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !ws || R <= 0 || C <= 0) return CFL_EINVAL;
    const bool xmask = relu && !y && !relu_mask;         // ReLU mask recomputed from x: needs beta, and no residual
    if ((xmask && (has_residual || !beta)) || (has_residual && !dres)) return CFL_EINVAL;
    if (C % 8 != 0 || ((C >> 3) < 256 && 256 % (C >> 3) != 0)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const G *d = (const G*)dy, *d2 = (const G*)dy2, *xx = (const G*)x, *yy = (const G*)y;
    if (bn_use_sliced(R, C)) {
        const SPlan sp = bn_splan(R, C);
        float* qdb = (float*)ws;
        float* qdg = qdb + (size_t)sp.parts * C;
        const dim3 sg(sp.parts, sp.slices);
#define BN_REDUCE_SL(RELU_, XM_) CFL_LAUNCH(K_BN_BWD_REDUCE, (cfl_bn_bwd_reduce_sl_kernel<G, RELU_, XM_>), sg, dim3(256), 0, stream, d, d2, xx, yy, \
                                            save_mean, save_invstd, gamma, beta, R, C, sp.rows_per_block, qdb, qdg, relu_mask)
        if (xmask) BN_REDUCE_SL(true, true); else if (relu) BN_REDUCE_SL(true, false); else BN_REDUCE_SL(false, false);
#undef BN_REDUCE_SL
        G *sx = (G*)dx, *sr = (G*)dres;
#define BN_APPLY_SL(RES_, RELU_, XM_) CFL_LAUNCH(K_BN_BWD_APPLY, (cfl_bn_bwd_apply_sl_kernel<G, RES_, RELU_, XM_>), sg, dim3(256), 0, stream, d, d2, \
                                                 xx, yy, save_mean, save_invstd, gamma, beta, qdb, qdg, sp.parts, dbeta, dgamma, R, C,       \
                                                 sp.rows_per_block, sx, sr, relu_mask)
        if (xmask) BN_APPLY_SL(false, true, true);
        else if (has_residual && relu) BN_APPLY_SL(true, true, false);
        else if (has_residual) BN_APPLY_SL(true, false, false);
        else if (relu) BN_APPLY_SL(false, true, false);
        else BN_APPLY_SL(false, false, false);
#undef BN_APPLY_SL
        return 0;
    }
    const Plan p = bn_plan(R, C);
    float* pdb = (float*)ws;
    float* pdg = pdb + (size_t)p.nblk * C;
    const dim3 grid(p.nblk, p.gy);
#define BN_REDUCE(RELU_, XM_) CFL_LAUNCH(K_BN_BWD_REDUCE, (cfl_bn_bwd_reduce_kernel<G, RELU_, XM_>), grid, dim3(256), 0, stream, d, d2, xx, yy, \
                                         save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, pdb, pdg, relu_mask, (const B8*)nullptr, 0, 0)
    if (xmask) BN_REDUCE(true, true); else if (relu) BN_REDUCE(true, false); else BN_REDUCE(false, false);
#undef BN_REDUCE
    CFL_LAUNCH(K_BN_BWD_FINAL, cfl_bn_bwd_final_kernel, dim3(cfl_cdiv(C, 16)), dim3(16 * BN_FG), 0, stream, pdb, pdg, p.nblk, C, dbeta, dgamma);
    G *ox = (G*)dx, *orr = (G*)dres;
#define BN_APPLY(RES_, RELU_, XM_) CFL_LAUNCH(K_BN_BWD_APPLY, (cfl_bn_bwd_apply_kernel<G, RES_, RELU_, XM_>), grid, dim3(256), 0, stream, d, d2, \
                                              xx, yy, save_mean, save_invstd, gamma, beta, dbeta, dgamma, R, C, p.rows_per_block, ox, orr, relu_mask, (const B8*)nullptr, 0, 0)
    if (xmask) BN_APPLY(false, true, true);
    else if (has_residual && relu) BN_APPLY(true, true, false);
    else if (has_residual) BN_APPLY(true, false, false);
    else if (relu) BN_APPLY(false, true, false);
    else BN_APPLY(false, false, false);
#undef BN_APPLY
    return 0;
}
template <class G>
static int bn_pool_fwd_t(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int N, int H,
                         int W, int C, float eps, float momentum, void* y_pool, void* idx, float* save_mean, float* save_invstd, void* ws,
                         void* stream_) {
    if (!x || !gamma || !beta || !y_pool || !idx || !save_mean || !save_invstd || !ws || N <= 0 || H <= 0 || W <= 0 || C <= 0) return CFL_EINVAL;
    if (C % 8 != 0 || ((C >> 3) < 256 && 256 % (C >> 3) != 0)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const long long R = (long long)N * H * W;
    const Plan p = bn_plan(R, C);
    float* psum = (float*)ws;
    float* psq = psum + (size_t)p.nblk * C;
    CFL_LAUNCH(K_BN_STATS, (cfl_bn_stats_kernel<G>), dim3(p.nblk, p.gy), dim3(256), 0, stream, (const G*)x, R, C, p.rows_per_block, psum, psq);
    CFL_LAUNCH(K_BN_FINAL, cfl_bn_final_kernel, dim3(cfl_cdiv(C, 16)), dim3(16 * BN_FG), 0, stream, psum, psq, p.nblk, C, R, eps,
               momentum, save_mean, save_invstd, running_mean, running_var);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 8);
    CFL_LAUNCH(K_BN_POOL_FWD, (cfl_bn_pool_fwd_kernel<G>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const G*)x,
               save_mean, save_invstd, gamma, beta, N, H, W, C / 8, Ho, Wo, (G*)y_pool, (B8*)idx);
    return 0;
}

template <class G>
static int bn_pool_bwd_t(const void* g_pool, const void* idx, const void* x, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, int N, int H, int W, int C, void* dx, float* dgamma, float* dbeta, void* ws,
                         void* stream_) {
    if (!g_pool || !idx || !x || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !ws || N <= 0 || H <= 0 ||
        W <= 0 || C <= 0)
        return CFL_EINVAL;
    if (C % 8 != 0 || ((C >> 3) < 256 && 256 % (C >> 3) != 0)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const long long R = (long long)N * H * W;
    const Plan p = bn_plan(R, C);
    float* pdb = (float*)ws;
    float* pdg = pdb + (size_t)p.nblk * C;
    const dim3 grid(p.nblk, p.gy);
    const G* none = nullptr;
    CFL_LAUNCH(K_BN_POOL_BWD_REDUCE, (cfl_bn_bwd_reduce_kernel<G, true, true, true>), grid, dim3(256), 0, stream, (const G*)g_pool, none,
               (const G*)x, none, save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, pdb, pdg, (const unsigned char*)nullptr,
               (const B8*)idx, H, W);
    CFL_LAUNCH(K_BN_BWD_FINAL, cfl_bn_bwd_final_kernel, dim3(cfl_cdiv(C, 16)), dim3(16 * BN_FG), 0, stream, pdb, pdg, p.nblk, C, dbeta, dgamma);
    CFL_LAUNCH(K_BN_POOL_BWD_APPLY, (cfl_bn_bwd_apply_kernel<G, false, true, true, true>), grid, dim3(256), 0, stream, (const G*)g_pool, none,
               (const G*)x, none, save_mean, save_invstd, gamma, beta, dbeta, dgamma, R, C, p.rows_per_block, (G*)dx, (G*)nullptr,
               (const unsigned char*)nullptr, (const B8*)idx, H, W);
    return 0;
}
}  // namespace

extern "C" {

size_t cfl_bn_ws_bytes(long long R, int C) {
    if (R <= 0 || C <= 0) return 256;
    const Plan p = bn_plan(R, C);
    size_t n = (size_t)p.nblk;
    if (C >= 256 && C % 64 == 0 && (size_t)bn_splan(R, C).parts > n) n = (size_t)bn_splan(R, C).parts;
    return cfl_align256((size_t)2 * n * C * sizeof(float));
}

// measurement / fallback switch of the channel-sliced map: on = 0 / 1 sets it, a negative value only queries; returns the
// previous setting (also CFL_BN_NO_SLICE=1 in the environment)
int cfl_bn_sliced(int on) {
    const int old = bn_sliced_on();
    if (on >= 0) bn_sliced_on() = on ? 1 : 0;
    return old;
}

int cfl_bn_fwd(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
               float* running_var, long long R, int C, float eps, float momentum, int relu, void* y,
               float* save_mean, float* save_invstd, unsigned char* relu_mask, void* ws, void* stream_) {
    return bn_fwd_t<U4>(x, residual, gamma, beta, running_mean, running_var, R, C, eps, momentum, relu, y, save_mean, save_invstd, relu_mask, ws, stream_);
}

// the same for fp32 activations (channels_last [R, C] floats): the clients' fp32 encoders
int cfl_bn_fwd_f32(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
               float* running_var, long long R, int C, float eps, float momentum, int relu, void* y,
               float* save_mean, float* save_invstd, unsigned char* relu_mask, void* ws, void* stream_) {
    return bn_fwd_t<F8>(x, residual, gamma, beta, running_mean, running_var, R, C, eps, momentum, relu, y, save_mean, save_invstd, relu_mask, ws, stream_);
}

// cfl_bn_fwd without its statistics pass: psum / psq [nblk, C] were written by the producer of x (the statistics epilogue of
// cfl_gemm_bf16_nt_stats: per-column sum / sum of squares of the stored bf16 values, one partial row per wave).
int cfl_bn_fwd_pre(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, long long R, int C, float eps, float momentum, int relu, void* y,
                   float* save_mean, float* save_invstd, unsigned char* relu_mask, const float* psum, const float* psq, int nblk,
                   void* stream_) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !psum || !psq || nblk <= 0 || R <= 0 || C <= 0) return CFL_EINVAL;
    if (C % 8 != 0 || ((C >> 3) < 256 && 256 % (C >> 3) != 0)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const Plan p = bn_plan(R, C);
    const dim3 grid(p.nblk, p.gy);
    CFL_LAUNCH(K_BN_FINAL, cfl_bn_final_kernel, dim3(cfl_cdiv(C, 16)), dim3(16 * BN_FG), 0, stream, psum, psq, nblk, C, R, eps,
               momentum, save_mean, save_invstd, running_mean, running_var);
    const U4* xr = (const U4*)x; const U4* rr = (const U4*)residual; U4* yy = (U4*)y;
    if (residual && relu)
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<U4, true, true>), grid, dim3(256), 0, stream, xr, rr, save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    else if (residual)
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<U4, true, false>), grid, dim3(256), 0, stream, xr, rr, save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    else if (relu)
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<U4, false, true>), grid, dim3(256), 0, stream, xr, rr, save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    else
        CFL_LAUNCH(K_BN_APPLY, (cfl_bn_apply_kernel<U4, false, false>), grid, dim3(256), 0, stream, xr, rr, save_mean, save_invstd, gamma, beta, R, C, p.rows_per_block, yy, relu_mask);
    return 0;
}

int cfl_bn_apply(const void* x, const void* residual, const float* mean, const float* invstd, const float* gamma,
                 const float* beta, long long R, int C, int relu, void* y, void* stream_) {
    return bn_apply_t<U4>(x, residual, mean, invstd, gamma, beta, R, C, relu, y, stream_);
}

// the same for fp32 activations (channels_last [R, C] floats): the clients' fp32 encoders
int cfl_bn_apply_f32(const void* x, const void* residual, const float* mean, const float* invstd, const float* gamma,
                 const float* beta, long long R, int C, int relu, void* y, void* stream_) {
    return bn_apply_t<F8>(x, residual, mean, invstd, gamma, beta, R, C, relu, y, stream_);
}

int cfl_bn_bwd(const void* dy, const void* dy2, const void* x, const void* y, const unsigned char* relu_mask, const float* gamma,
               const float* beta, const float* save_mean, const float* save_invstd, long long R, int C, int relu, int has_residual,
               void* dx, void* dres, float* dgamma, float* dbeta, void* ws, void* stream_) {
    return bn_bwd_t<U4>(dy, dy2, x, y, relu_mask, gamma, beta, save_mean, save_invstd, R, C, relu, has_residual, dx, dres, dgamma, dbeta, ws, stream_);
}

// the same for fp32 activations (channels_last [R, C] floats): the clients' fp32 encoders
int cfl_bn_bwd_f32(const void* dy, const void* dy2, const void* x, const void* y, const unsigned char* relu_mask, const float* gamma,
               const float* beta, const float* save_mean, const float* save_invstd, long long R, int C, int relu, int has_residual,
               void* dx, void* dres, float* dgamma, float* dbeta, void* ws, void* stream_) {
    return bn_bwd_t<F8>(dy, dy2, x, y, relu_mask, gamma, beta, save_mean, save_invstd, R, C, relu, has_residual, dx, dres, dgamma, dbeta, ws, stream_);
}

// cfl_bn_bwd for a pre-joined gradient (no ReLU mask, no residual, one upstream gradient) whose apply pass also produces the
// weight gradient of the 1 x 1 convolution that made x:  dw[C, P] (bf16) = dx^T a_in, a_in [R, P] bf16 = that convolution's input.
int cfl_bn_bwd_wgrad_supported(long long R, int C, int P) {
    return (P == WG_PN && C % WG_CH == 0 && C >= 256 && C <= 2048 && R >= 4 * WG_KS && bn_use_sliced(R, C)) ? 1 : 0;
}

size_t cfl_bn_bwd_wgrad_ws_bytes(long long R, int C, int P) {
    if (!cfl_bn_bwd_wgrad_supported(R, C, P)) return 256;
    const size_t red = cfl_align256((size_t)2 * bn_splan(R, C).parts * C * sizeof(float));
    return red + cfl_align256((size_t)bn_wplan(R, C).parts * C * P * sizeof(float));
}

int cfl_bn_bwd_wgrad(const void* dy, const void* x, const void* a_in, int P, const float* gamma, const float* save_mean,
                     const float* save_invstd, long long R, int C, void* dx, float* dgamma, float* dbeta, void* dw, void* ws,
                     void* stream_) {
    if (!dy || !x || !a_in || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !dw || !ws || R <= 0 || C <= 0) return CFL_EINVAL;
    if (!cfl_bn_bwd_wgrad_supported(R, C, P) || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)a_in | (uintptr_t)dx | (uintptr_t)dw) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const SPlan sp = bn_splan(R, C);
    const WPlan wp = bn_wplan(R, C);
    float* qdb = (float*)ws;
    float* qdg = qdb + (size_t)sp.parts * C;
    float* part = (float*)((char*)ws + cfl_align256((size_t)2 * sp.parts * C * sizeof(float)));
    const U4 *d = (const U4*)dy, *xx = (const U4*)x, *none = nullptr;
    CFL_LAUNCH(K_BN_BWD_REDUCE, (cfl_bn_bwd_reduce_sl_kernel<U4, false, false>), dim3(sp.parts, sp.slices), dim3(256), 0, stream, d, none, xx, none,
               save_mean, save_invstd, gamma, (const float*)nullptr, R, C, sp.rows_per_block, qdb, qdg, (const unsigned char*)nullptr);
    constexpr size_t LDS = (size_t)2 * WG_TILE_A + 2 * WG_TILE_B + 4096;
    CFL_SET_LDS(cfl_bn_bwd_apply_wgrad_kernel, LDS);
    CFL_LAUNCH(K_BN_BWD_APPLY_WG, cfl_bn_bwd_apply_wgrad_kernel, dim3(wp.parts, C / WG_CH), dim3(512), LDS, stream, d, xx, (const u16*)a_in,
               save_mean, save_invstd, gamma, (const float*)qdb, (const float*)qdg, sp.parts, dbeta, dgamma, R, C, wp.rows_per_block,
               (U4*)dx, part);
    const long long n = (long long)C * P;
    CFL_LAUNCH(K_BN_WGRAD_REDUCE, cfl_bn_wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)part,
               wp.parts, n, (u32*)dw);
    return 0;
}

// Stem tail in one pass per direction (see cfl_bn_pool_fwd_kernel).  x [N,H,W,C] bf16 (C % 8 == 0, C <= 2048);
// y_pool [N,Ho,Wo,C] bf16, idx [N*Ho*Wo*C] bytes (Ho = (H-1)/2+1); ws: cfl_bn_ws_bytes(N*H*W, C).  The _f32 forms take fp32
// activations (the clients' encoders, src/networks/resnet_client.py:25-29,64).
int cfl_bn_pool_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int N, int H,
                    int W, int C, float eps, float momentum, void* y_pool, void* idx, float* save_mean, float* save_invstd, void* ws,
                    void* stream_) {
    return bn_pool_fwd_t<U4>(x, gamma, beta, running_mean, running_var, N, H, W, C, eps, momentum, y_pool, idx, save_mean, save_invstd, ws, stream_);
}
int cfl_bn_pool_fwd_f32(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int N, int H,
                        int W, int C, float eps, float momentum, void* y_pool, void* idx, float* save_mean, float* save_invstd, void* ws,
                        void* stream_) {
    return bn_pool_fwd_t<F8>(x, gamma, beta, running_mean, running_var, N, H, W, C, eps, momentum, y_pool, idx, save_mean, save_invstd, ws, stream_);
}

// g_pool: gradient w.r.t. y_pool [N,Ho,Wo,C]; dx [N,H,W,C] = gradient w.r.t. x (both in the activation's type); ws as above.
int cfl_bn_pool_bwd(const void* g_pool, const void* idx, const void* x, const float* gamma, const float* beta, const float* save_mean,
                    const float* save_invstd, int N, int H, int W, int C, void* dx, float* dgamma, float* dbeta, void* ws, void* stream_) {
    return bn_pool_bwd_t<U4>(g_pool, idx, x, gamma, beta, save_mean, save_invstd, N, H, W, C, dx, dgamma, dbeta, ws, stream_);
}
int cfl_bn_pool_bwd_f32(const void* g_pool, const void* idx, const void* x, const float* gamma, const float* beta, const float* save_mean,
                        const float* save_invstd, int N, int H, int W, int C, void* dx, float* dgamma, float* dbeta, void* ws,
                        void* stream_) {
    return bn_pool_bwd_t<F8>(g_pool, idx, x, gamma, beta, save_mean, save_invstd, N, H, W, C, dx, dgamma, dbeta, ws, stream_);
}

}  // extern "C"
