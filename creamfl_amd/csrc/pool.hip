// pool.hip -- 3x3 / stride 2 / pad 1 max pooling on NHWC bf16 activations: the ResNet stem pool
// (torchvision ResNet.maxpool inside the trunk built at src/networks/models/image_encoder.py:27-36 and
// src/networks/resnet_client.py:19).  torch's NHWC kernels take 257 us forward / 614 us backward on the bench's
// [256, 112, 112, 64] stem output (0.85-1.9 TB/s); this is pure streaming: 565 MB each way.
//   fwd: thread = 8 channels of one output pixel; the 9 taps are 16-byte loads; the arg-max tap (0..8, first maximum
//        in row-major window order like torch; NaN wins) is kept as one byte per element for the backward.
//   bwd: gather form, thread = 8 channels of one INPUT pixel: the <= 4 windows that cover it are checked against
//        their stored tap -- no atomics, every dx element written exactly once.
#include "common.h"
#include "colmap.h"

namespace {

struct __attribute__((aligned(8))) B8 { unsigned int lo, hi; };          // 8 tap indices

__global__ __launch_bounds__(256) void cfl_maxpool_fwd_kernel(const U4* __restrict__ x, int N, int H, int W, int C8, int Ho, int Wo,
                                                              U4* __restrict__ y, B8* __restrict__ idx) {
    const long long total = (long long)N * Ho * Wo * C8;
    // XCD-contiguous block order: vertically neighbouring output rows share an input row; dispatched round-robin they sit on
    // different XCDs (private L2s) and the shared row is fetched twice (round 1 PMC: 776 MB fetched for 565 MB algorithmic)
    for (long long i = (long long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C8);
        long long p = i / C8;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
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
                for (int k = 0; k < 8; ++k)
                    if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; tap[k] = dh * 3 + dw; }
            }
        }
        y[i] = pack8(best);
        B8 t;
        t.lo = tap[0] | (tap[1] << 8) | (tap[2] << 16) | (tap[3] << 24);
        t.hi = tap[4] | (tap[5] << 8) | (tap[6] << 16) | (tap[7] << 24);
        idx[i] = t;
    }
}

__global__ __launch_bounds__(256) void cfl_maxpool_bwd_kernel(const U4* __restrict__ dy, const B8* __restrict__ idx, int N, int H, int W,
                                                              int C8, int Ho, int Wo, U4* __restrict__ dx) {
    const long long total = (long long)N * H * W * C8;
    // XCD-contiguous block order (see the forward): the <= 4 windows of an input pixel are re-read from the SAME L2
    for (long long i = (long long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C8);
        long long p = i / C8;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // windows oh with 2*oh - 1 <= h <= 2*oh + 1: at most 2 x 2.  All four (tap, gradient) pairs are requested up front
        // (clamped addresses, masked afterwards): loads under the validity branches would serialize on their latency.
        const int oh0 = h >> 1, oh1 = (h + 1) >> 1, ow0 = w >> 1, ow1 = (w + 1) >> 1;
        B8 tp[4];
        U4 gv[4];
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
                gv[a * 2 + b] = dy[o];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;
            float g[8];
            unpack8(gv[q], g);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned int tk = ((k < 4 ? tp[q].lo : tp[q].hi) >> (8 * (k & 3))) & 0xffu;
                if (tk == want[q]) acc[k] += g[k];
            }
        }
        dx[i] = pack8(acc);
    }
}

// ---- stem space-to-depth ------------------------------------------------------------------------------------------------
// The ResNet stem (torchvision ResNet.conv1: 7x7 / stride 2 / pad 3 on 3 channels) is the worst-shaped convolution of the trunk
// for an implicit-GEMM kernel (3 input channels: 6-byte pixels, K = 147).  With 7 = 2 * 4 - 1 it is EXACTLY a 4x4 / stride-1
// convolution of the space-to-depth image: y[oh, ow] = sum_{a,b in -2..1} sum_{p,q in 0..1} w[2a+p+3, 2b+q+3] x[2(oh+a)+p, 2(ow+b)+q]
// (the tap 2a+p+3 = -1 does not exist: zero weight).  This kernel writes that image in one pass:
//   out[n, i, j, (2p + q) * 3 + c] = x[n, 2 (i - 2) + p, 2 (j - 2) + q, c]   for 2 <= i, j < Ho + 2, zero elsewhere (the
//   convolution's padding: two rows before, one after) and in the 4 padding channels;  out [N, Ho + 3, Wo + 3, 16] bf16.
// One thread per output pixel: two 12-byte (bf16 images) or 24-byte (fp32 images, rounded to bf16 here) reads -- pixels
// 2j', 2j'+1 of rows 2i', 2i'+1 --, one 32-byte write.
template <bool F32>
__global__ __launch_bounds__(256) void cfl_stem_s2d_kernel(const void* __restrict__ x_, int N, int H, int W, U4* __restrict__ out) {
    const int Ho = H >> 1, Wo = W >> 1, Hp = Ho + 3, Wp = Wo + 3;
    const long long total = (long long)N * Hp * Wp;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % Wp);
    const long long t = i / Wp;
    const int r = (int)(t % Hp), n = (int)(t / Hp);
    U4 lo = {0u, 0u, 0u, 0u}, hi = {0u, 0u, 0u, 0u};
    if (r >= 2 && r < Ho + 2 && j >= 2 && j < Wo + 2) {
        // a row contributes 6 consecutive elements (pixels 2j', 2j'+1) starting at the EVEN element index
        // ((n H + row) W + 2 j') 3; the next row is W * 3 elements further (W even)
        const long long e0 = (((long long)n * H + 2 * (r - 2)) * W + 2 * (j - 2)) * 3;
        unsigned int d[6];                                    // 12 bf16: row p = 0 then row p = 1
        if (F32) {                                            // fp32 images (the autocast regime): rounded here, no separate cast pass
            typedef float f2 __attribute__((ext_vector_type(2)));
            const float* x = reinterpret_cast<const float*>(x_);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f2* s = reinterpret_cast<const f2*>(x + e0 + (long long)p * W * 3);
#pragma unroll
                for (int k = 0; k < 3; ++k) { const f2 v = s[k]; d[p * 3 + k] = bf16_rne(v[0]) | (bf16_rne(v[1]) << 16); }
            }
        } else {
            const unsigned int* x = reinterpret_cast<const unsigned int*>(x_);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int k = 0; k < 3; ++k) d[p * 3 + k] = x[((e0 + (long long)p * W * 3) >> 1) + k];
        }
        lo.x = d[0]; lo.y = d[1]; lo.z = d[2];                // channels 0..5  = (p=0, q=0..1, c)
        lo.w = d[3]; hi.x = d[4]; hi.y = d[5];                // channels 6..11 = (p=1, q=0..1, c)
    }
    out[i * 2] = lo;
    out[i * 2 + 1] = hi;
}

}  // namespace

extern "C" {

int cfl_stem_s2d(const void* x, int x_f32, int N, int H, int W, void* out, void* stream_) {
    if (!x || !out || N <= 0 || H <= 0 || W <= 0) return CFL_EINVAL;
    if ((H & 1) || (W & 1) || (((uintptr_t)x) & 7) || (((uintptr_t)out) & 15)) return CFL_ELIMIT;
    const long long total = (long long)N * ((H >> 1) + 3) * ((W >> 1) + 3);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (x_f32)
        CFL_LAUNCH(K_MAXPOOL, cfl_stem_s2d_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream_, x, N, H, W, (U4*)out);
    else
        CFL_LAUNCH(K_MAXPOOL, cfl_stem_s2d_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream_, x, N, H, W, (U4*)out);
    return 0;
}

int cfl_maxpool3s2_fwd(const void* x, int N, int H, int W, int C, void* y, void* idx, void* stream_) {
    if (!x || !y || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return CFL_EINVAL;
    if (C % 8 != 0) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 8);
    const long long blocks = (total + 255) / 256;
    CFL_LAUNCH(K_MAXPOOL, cfl_maxpool_fwd_kernel, dim3((unsigned)(blocks < (1LL << 30) ? blocks : (1LL << 30))), dim3(256), 0, stream,
               (const U4*)x, N, H, W, C / 8, Ho, Wo, (U4*)y, (B8*)idx);
    return 0;
}

int cfl_maxpool3s2_bwd(const void* dy, const void* idx, int N, int H, int W, int C, void* dx, void* stream_) {
    if (!dy || !idx || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return CFL_EINVAL;
    if (C % 8 != 0) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * H * W * (C / 8);
    const long long blocks = (total + 255) / 256;
    // one pass per thread (a capped grid with a 1.5-trip grid-stride loop leaves a third of the machine idle in the second trip)
    CFL_LAUNCH(K_MAXPOOL, cfl_maxpool_bwd_kernel, dim3((unsigned)(blocks < (1LL << 30) ? blocks : (1LL << 30))), dim3(256), 0, stream,
               (const U4*)dy, (const B8*)idx, N, H, W, C / 8, Ho, Wo, (U4*)dx);
    return 0;
}

}  // extern "C"
