// wgrad3x3_x3.hip -- weight gradient of the 3 x 3 / stride 1 / padding 1 convolutions of the CLIENTS' fp32 ResNet-18 at fp32-class
// accuracy on the bf16 matrix pipe (the "3 x bf16 split" of tile_x3.h / conv3x3_x3.hip).
//
//   dW[co][kh][kw][ci] = sum over (n, h, w) of dY[n, h, w, co] * X[n, h + kh - 1, w + kw - 1, ci]          (zero padding)
//
// Where it sits: BasicBlock.conv1 / conv2 of src/networks/resnet_client.py:33-66 (the reference runs them in fp32 on cuDNN).  After
// conv3x3_x3.hip took the forward and the data gradient, the library's fp32 weight-gradient kernels (253-301 us each at batch 128,
// 13 per image-client step = a third of the step) were what was left of the BasicBlocks on MIOpen.
//
// The arrangement is wgrad3x3.hip's (a TN GEMM [Co] x [9 Ci] reduced over all positions, one image row per K group, the nine taps
// sharing three w-shifted X fragments and the dY fragments of rows r - 1, r, r + 1, operands read TRANSPOSED out of LDS by
// ds_read_b64_tr_b16, the zero padding of a row supplied by zero columns of the padded row, split-K with the tiles of one range on
// one XCD, fp32 partials + a fixed-order reduce: deterministic) with two differences that the fp32 operands force:
//   * an fp32 element cannot go to LDS by DMA and be split on the way, so rows are staged through registers: a thread loads 16-byte
//     pieces of rows ahead at the top of a row (two register sets: two rows in flight), and two rows later splits them (x = hi + lo,
//     two bf16) and writes the hi and lo planes of that row's LDS slot -- plain HIP, the compiler's counted vmcnt waits are the pipeline;
//   * each (dY row, X shift) pair is three MFMAs into one accumulator: hi.hi + lo.hi + hi.lo (dropped term: 2^-16 relative).
// A workgroup is 4 waves = 64 dY channels x 64 X channels x 9 taps (144 accumulator registers per lane), 27 MFMAs per fragment set
// of 12 (the bf16 kernel: 9 per 6), so the matrix pipe sees three times the work per staged byte.
// Work is split over ROW ranges of the [N H, W, C] matrices (not image ranges: the batch of a client is 128, and a 56 x 56 x 64
// layer is ONE tile -- image ranges would leave half the chip idle): a workgroup walks global rows g0 .. g1 - 1, keeps rows
// g - 1 .. g + 2 in a four-slot LDS ring and knows per row (h = g mod H, uniform) whether the row above / below belongs to the image
// (first / last row of an image: those dY fragments are replaced by zeros).
#include <type_traits>
#include "common.h"
#include "tile_x3.h"

namespace {

typedef __bf16 xw_bf16x8 __attribute__((ext_vector_type(8)));
typedef short xw_s16x4 __attribute__((ext_vector_type(4)));

template <int H_, int W_>
struct XWCfg {
    static constexpr int H = H_, W = W_;
    static constexpr int NCH = (W + 1 + 15) / 16;            // K steps per row; at least one zero column
    static constexpr int WP = 16 * NCH;
    static constexpr int PL = WP * 128;                      // bytes of one plane of a row: WP positions x 64 channels bf16
    static constexpr int SLOT = 4 * PL;                      // X hi | X lo | dY hi | dY lo
    static constexpr int NS = 4;
    static constexpr int LDS = NS * SLOT;
    static constexpr int NL = WP / 16;                       // 16-byte pieces a thread loads per operand and row (256 threads)
    static_assert(W < WP && LDS <= 160 * 1024, "padding / LDS");
};

__device__ __forceinline__ xw_bf16x8 xw_tr(const char* p0, const char* p1) {
    const xw_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) xw_s16x4*)(p0));
    const xw_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) xw_s16x4*)(p1));
    union { xw_s16x4 h[2]; xw_bf16x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}

struct XWFrags { xw_bf16x8 yh[3], yl[3], xh[3], xl[3]; };

// grid = tiles x splits (splits a multiple of 8), 256 threads, C::LDS bytes of dynamic LDS.  part: [splits][Co][9][Ci] fp32.
// TR = N H rows in all, rps rows per split.
template <class C, int OCC>
__global__ __launch_bounds__(256, OCC) void cfl_conv3x3_x3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int TR,
                                                                         int Ci, int Co, int rps, float* __restrict__ part) {
    constexpr int H = C::H, W = C::W, NCH = C::NCH, WP = C::WP, NL = C::NL;
    extern __shared__ __attribute__((aligned(16))) char xwlds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w & 1, wn = w >> 1;                       // dY channel block, X channel block (32 each)
    const int ntci = Ci >> 6, ntile = (Co >> 6) * ntci;
    const int b = blockIdx.x, xcd = b & 7, jb = b >> 3;
    const int tile = jb % ntile, split = xcd + 8 * (jb / ntile);
    const int co0 = (tile / ntci) * 64, ci0 = (tile % ntci) * 64;
    const int g0 = split * rps;
    int g1 = g0 + rps;
    g1 = g1 > TR ? TR : g1;

    // transposing-read offsets (wgrad3x3.hip): lane = 16 g + p reads K row 8 (g >> 1) + (p >> 2) [+ 4], channels 16 (g & 1) +
    // 4 (p & 3) .. + 3 of the wave's 32-channel block; 128-byte position rows, 16-byte piece ^ ((pos >> 1) & 1) << 2
    const int gq = lane >> 4, pq = lane & 15;
    const int krow = 8 * (gq >> 1) + (pq >> 2);
    const int cx = wn * 32 + 16 * (gq & 1) + 4 * (pq & 3), cy = wm * 32 + 16 * (gq & 1) + 4 * (pq & 3);
    auto p_off = [&](int pos, int ch) { return pos * 128 + ((((ch >> 3) ^ (((pos >> 1) & 1) << 2)) & 7) << 4) + (ch & 7) * 2; };
    int offY[2], offX[3][2], offXfirst[2], offXlast[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = krow + 4 * j;
        offY[j] = p_off(k, cy);
#pragma unroll
        for (int s = 0; s < 3; ++s) offX[s][j] = p_off(k + s - 1, cx);
        offXfirst[j] = offX[0][j] + (k == 0 ? WP * 128 : 0);         // position -1 -> WP - 1 (a zero column)
        offXlast[j] = offX[2][j] - (k == 15 ? WP * 128 : 0);         // position WP -> 0 (meets dY's zero column)
    }

    // staging: piece i of a thread = position (i * 256 + t) >> 4, channels 4 ((i * 256 + t) & 15) .. + 3 of the tile's 64
    int st_off[NL];
    bool st_ok[NL];
    const float* px[NL];
    const float* py[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = i * 256 + t, pos = idx >> 4, c4 = idx & 15;
        st_ok[i] = pos < W;
        st_off[i] = pos * 128 + ((((c4 >> 1) ^ (((pos >> 1) & 1) << 2)) & 7) << 4) + (c4 & 1) * 8;
        const long long p = (long long)(g0 - 1) * W + (st_ok[i] ? pos : 0);
        px[i] = x + p * Ci + ci0 + 4 * c4;
        py[i] = dy + p * Co + co0 + 4 * c4;
    }
    const long long stepX = (long long)W * Ci, stepY = (long long)W * Co;
    // TWO register sets alternate: the loads of rows g + 2 and g + 3 are both in flight while row g is computed (a row's compute is
    // 0.4-1.4 us, an HBM round trip under load ~2 us: with one set the kernel ran at one row per round trip, 100-110 us at every shape)
    struct Regs { f32x4 x[NL], y[NL]; };
    Regs RA, RB;
    int gl = g0 - 1;                                          // the row the pointers stand on
    auto load_row = [&](Regs& R) {                            // rows are loaded in order: g0 - 1, g0, ...
        const bool in = gl >= 0 && gl < TR;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const bool ok = in && st_ok[i];
            R.x[i] = ok ? *reinterpret_cast<const f32x4*>(px[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
            R.y[i] = ok ? *reinterpret_cast<const f32x4*>(py[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
            px[i] += stepX;
            py[i] += stepY;
        }
        ++gl;
    };
    auto store_row = [&](const Regs& R, int g) {              // the registers hold row g
        char* s = xwlds + (g & 3) * C::SLOT;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            x3::bf16x4 xh, xl, yh, yl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 a, bb;
                x3::split1(R.x[i][e], a, bb);
                xh[e] = a; xl[e] = bb;
                x3::split1(R.y[i][e], a, bb);
                yh[e] = a; yl[e] = bb;
            }
            *reinterpret_cast<x3::bf16x4*>(s + st_off[i]) = xh;
            *reinterpret_cast<x3::bf16x4*>(s + C::PL + st_off[i]) = xl;
            *reinterpret_cast<x3::bf16x4*>(s + 2 * C::PL + st_off[i]) = yh;
            *reinterpret_cast<x3::bf16x4*>(s + 3 * C::PL + st_off[i]) = yl;
        }
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    // unit (g, c): the X fragments of row g and the dY fragments of rows g - 1 .. g + 1, chunk c, both planes.  ONE row body: on the
    // first / last row of an image the fragments of the row above / below (another image's row, or nothing) are replaced by zeros
    // -- a select on a uniform condition, 16 v_cndmask per 27 MFMAs; three specialised bodies behind a branch made the compiler move
    // all 144 accumulators between the two register files on every row.
    auto load_unit = [&](int g, int c, bool top, bool bot, XWFrags& f) {
        const char* xs = xwlds + (g & 3) * C::SLOT + c * (16 * 128);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int o0 = (s == 0 && c == 0) ? offXfirst[0] : ((s == 2 && c == NCH - 1) ? offXlast[0] : offX[s][0]);
            const int o1 = (s == 0 && c == 0) ? offXfirst[1] : ((s == 2 && c == NCH - 1) ? offXlast[1] : offX[s][1]);
            f.xh[s] = xw_tr(xs + o0, xs + o1);
            f.xl[s] = xw_tr(xs + C::PL + o0, xs + C::PL + o1);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const char* ys = xwlds + ((g - 1 + d) & 3) * C::SLOT + 2 * C::PL + c * (16 * 128);
            f.yh[d] = xw_tr(ys + offY[0], ys + offY[1]);
            f.yl[d] = xw_tr(ys + C::PL + offY[0], ys + C::PL + offY[1]);
        }
        const xw_s16x4 z4 = {0, 0, 0, 0};
        union { xw_s16x4 h[2]; xw_bf16x8 v; } z;
        z.h[0] = z4; z.h[1] = z4;
        f.yh[0] = top ? z.v : f.yh[0];
        f.yl[0] = top ? z.v : f.yl[0];
        f.yh[2] = bot ? z.v : f.yh[2];
        f.yl[2] = bot ? z.v : f.yl[2];
    };
    auto row = [&](int g, bool top, bool bot) {
        XWFrags F[2];
        load_unit(g, 0, top, bot, F[0]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) load_unit(g, c + 1, top, bot, F[(c + 1) & 1]);
            const XWFrags& f = F[c & 1];
            // tap kh pairs X row h with the dY row h - kh + 1: d (row g - 1 + d) belongs to kh = 2 - d
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    f32x16 a = acc[2 - d][s];
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.yh[d], f.xh[s], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.yl[d], f.xh[s], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.yh[d], f.xl[s], a, 0, 0, 0);
                    acc[2 - d][s] = a;
                }
        }
    };

    if (g0 < g1) {
        load_row(RA); store_row(RA, g0 - 1);
        load_row(RA); store_row(RA, g0);
        load_row(RA); store_row(RA, g0 + 1);
        load_row(RA);                                         // rows g0 + 2, g0 + 3 wait in registers
        load_row(RB);
        __syncthreads();
        int h = g0 % H;
        for (int g = g0; g < g1; g += 2) {
            store_row(RA, g + 2);                             // slot of row g - 2: last read during row g - 1
            load_row(RA);                                     // row g + 4
            __builtin_amdgcn_sched_barrier(0);
            row(g, h == 0, h == H - 1);
            h = h + 1 == H ? 0 : h + 1;
            __syncthreads();
            if (g + 1 < g1) {
                store_row(RB, g + 3);
                load_row(RB);                                 // row g + 5
                __builtin_amdgcn_sched_barrier(0);
                row(g + 1, h == 0, h == H - 1);
                h = h + 1 == H ? 0 : h + 1;
                __syncthreads();
            }
        }
    }

    // C layout of the 32 x 32 MFMA: lane -> column (ci) lane & 31, rows (co) (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* out = part + ((long long)split * Co + co0 + wm * 32) * 9 * Ci + ci0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[((long long)i * 9 + a * 3 + c) * Ci] = acc[a][c][r];
            }
}

// dw[e] = sum over splits of part[s][e] in a FIXED order (wgrad3x3.hip's reduce without the cast): a thread sums every 4th split
// for 4 elements, the four partial sums are combined through LDS as (0 + 1) + (2 + 3).
__global__ __launch_bounds__(256) void cfl_conv3x3_x3_wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, long long n,
                                                                          float* __restrict__ dw) {
    __shared__ f32x4 red[4][64];
    const int eg = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long long e = ((long long)blockIdx.x * 64 + eg) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (e < n) {
        int k = sl;
        for (; k + 12 < nsplit; k += 16) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(part + (long long)k * n + e);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(part + (long long)(k + 4) * n + e);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(part + (long long)(k + 8) * n + e);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(part + (long long)(k + 12) * n + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += (v0[j] + v1[j]) + (v2[j] + v3[j]);
        }
        for (; k < nsplit; k += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + (long long)k * n + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += v[j];
        }
    }
    red[sl][eg] = s;
    __syncthreads();
    if (sl != 0 || e >= n) return;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (red[0][eg][j] + red[1][eg][j]) + (red[2][eg][j] + red[3][eg][j]);
    *reinterpret_cast<f32x4*>(dw + e) = o;
}

inline int& xw_splits_override() {
    static int v = getenv("CFL_X3WGRAD_SPLITS") ? atoi(getenv("CFL_X3WGRAD_SPLITS")) : 0;
    return v;
}

inline bool xw_ok(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || Ci <= 0 || Co <= 0 || Ci % 64 != 0 || Co % 64 != 0 || H != W) return false;
    if ((long long)N * H * W >= (1ll << 31)) return false;
    return H == 7 || H == 14 || H == 28 || H == 56;
}

// splits: a multiple of 8 (one row range per XCD and tile group); two workgroups per CU where two fit (maps up to 28 x 28: 64 KB of
// LDS, 256 registers), one at 56 x 56 (128 KB); no more than row pairs.  Measured at batch 128 (profiles/r6_x3wgrad_probe.jsonl):
// 28 x 28 x 128: 64 / 128 / 256 ranges = 100 + 6 / 89 + 12 / 93 + 22 us (kernel + reduce); 14 x 14 x 256: 16 / 64 = 109 + 7 / 89 + 22.
inline int xw_nsplit(int N, int H, int Ci, int Co) {
    const int ntile = (Co / 64) * (Ci / 64);
    int ns = xw_splits_override() > 0 ? xw_splits_override() : (H <= 28 ? 512 : 256) / ntile;
    ns = (ns / 8) * 8;
    if (ns < 8) ns = 8;
    const int TR = N * H;
    while (ns > 8 && (ns - 8) * 2 >= TR) ns -= 8;
    return ns;
}

template <class C, int OCC>
int xw_launch(const float* dy, const float* x, int N, int Ci, int Co, int ns, float* part, hipStream_t stream) {
    const int ntile = (Co / 64) * (Ci / 64);
    const int TR = N * C::H;
    CFL_SET_LDS((cfl_conv3x3_x3_wgrad_kernel<C, OCC>), C::LDS);
    CFL_LAUNCH(K_CONV3_X3_WGRAD, (cfl_conv3x3_x3_wgrad_kernel<C, OCC>), dim3(ntile * ns), dim3(256), C::LDS, stream, dy, x, TR, Ci, Co,
               cfl_cdiv(TR, ns), part);
    return 0;
}

}  // namespace

extern "C" int cfl_conv3x3_x3_wgrad_supported(int N, int H, int W, int Ci, int Co) { return xw_ok(N, H, W, Ci, Co) ? 1 : 0; }

extern "C" int cfl_conv3x3_x3_wgrad_splits(int splits) {
    const int old = xw_splits_override();
    if (splits >= 0) xw_splits_override() = splits;
    return old;
}

extern "C" size_t cfl_conv3x3_x3_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co) {
    if (!xw_ok(N, H, W, Ci, Co)) return 0;
    return cfl_align256((size_t)xw_nsplit(N, H, Ci, Co) * Co * 9 * Ci * sizeof(float));
}

extern "C" int cfl_conv3x3_x3_wgrad(const float* dy, const float* x, int N, int H, int W, int Ci, int Co, float* dw, void* ws,
                                    void* stream_) {
    if (!dy || !x || !dw || !ws) return CFL_EINVAL;
    if (!xw_ok(N, H, W, Ci, Co) || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw | (uintptr_t)ws) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const int ns = xw_nsplit(N, H, Ci, Co);
    float* part = (float*)ws;
    int rc;
    if (H == 7) rc = xw_launch<XWCfg<7, 7>, 2>(dy, x, N, Ci, Co, ns, part, stream);
    else if (H == 14) rc = xw_launch<XWCfg<14, 14>, 2>(dy, x, N, Ci, Co, ns, part, stream);
    else if (H == 28) rc = xw_launch<XWCfg<28, 28>, 2>(dy, x, N, Ci, Co, ns, part, stream);
    else rc = xw_launch<XWCfg<56, 56>, 1>(dy, x, N, Ci, Co, ns, part, stream);
    if (rc) return rc;
    const long long n = (long long)Co * 9 * Ci;
    CFL_LAUNCH(K_CONV3_X3_WGRAD_REDUCE, cfl_conv3x3_x3_wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, stream,
               (const float*)part, ns, n, dw);
    return 0;
}
