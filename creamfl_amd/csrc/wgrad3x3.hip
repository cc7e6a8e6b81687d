// wgrad3x3.hip -- weight gradient of the 3 x 3 / stride 1 / padding 1 convolutions of the ResNet trunk (bf16, channels_last).
//
//   dW[co][kh][kw][ci] = sum over (n, h, w) of dY[n, h, w, co] * X[n, h + kh - 1, w + kw - 1, ci]          (zero padding)
//
// torchvision Bottleneck.conv2 inside src/networks/models/image_encoder.py:27-36; the reference leaves it to cuDNN.  Round 6's
// accounting (profiles/r6_step_bytes.json, r6_bound_wgrad.json) is why this kernel exists: with every weight gradient replaced by
// nothing the bench step takes 37.4 ms instead of 43.9 -- the side stream's library kernels cost the step 6.5 ms although their own
// time is hidden -- and the library's k x k kernels (MIOpen igemm_wrw, 42 launches, 7.6 ms of kernel time per step) move ~20 GB per
// step for 2.3 GB of operands at ~15 % of the MFMA rate.  What bounds the backward pass is the SUM of what both streams ask of HBM and
// of the matrix pipes (docs/history/DESIGN_r1-r4.md section 6.1), so the lever is a kernel that asks for less of both.
//
// The problem is a TN GEMM [Co] x [9 Ci] with the reduction over all positions of the batch -- the slow axis of both channels_last
// operands -- and 59 GFLOP per bottleneck against 50-200 MB of operands: MFMA-bound.  Design:
//   * v_mfma_f32_32x32x16_bf16; an image row is padded to WP = 16 / 32 / 64 positions (w = W .. WP - 1 are zeros, written by the
//     staging from a page of zeros) = 1 / 2 / 4 K steps ("chunks").  Both operands are read TRANSPOSED out of LDS
//     (ds_read_b64_tr_b16: a lane ends up with 8 consecutive positions of one channel = the MFMA operand layout): no VALU touches an
//     operand.
//   * the nine taps share their operands: the three w-shifted X fragments of a row meet the dY fragments of rows r + 1, r, r - 1
//     (taps kh = 0, 1, 2).  A shifted fragment is the same transposing read with per-lane addresses moved by one position MODULO the
//     padded row: position -1 reads position WP - 1 (a zero column), position WP reads position 0 (finite, and multiplied by dY's
//     zero column) -- all reads stay inside the row's slot and the zero padding of the convolution comes from the data itself.
//     Rows -1 and H do not exist: their MFMAs are not issued.
//   * a workgroup owns a TCO (128: 8 waves, or 64: 4 waves) x 64 x 9 tile of dW for a range of images: 144 accumulator registers per
//     lane.  Rows stream through two LDS rings (X, dY) by LDS-DMA with a source-side XOR swizzle (conflict-free transposing reads),
//     D rows ahead; ONE barrier per row, counted vmcnt; the fragments a row needs are read while the MFMAs of the row before run.
//   * the instruction diet matters as much as the bytes: a wave hides ~5 other instructions behind an MFMA.  Version 2 of this file
//     (runtime row index, a switch for the vmcnt immediate, select-to-zero at image borders) spent 7 scalar + 4 vector instructions
//     per MFMA and ran 90 us at layer3's shape where version 1 (everything static) ran 68 (profiles/r6_sq_wgrad3_v2.json).  Now:
//     one-chunk rows (14 x 14, 7 x 7) unroll the whole image (row index static, X rows kept in registers as a rolling window: a
//     row is read from LDS once); wider rows run a row loop whose body exists in three variants (first / middle / last row of an
//     image); waits are immediates.
//   * the LDS-DMA is issued from inline assembly on purpose: the compiler orders every LDS read behind every LDS-DMA it knows of
//     (s_waitcnt vmcnt(0) before the first ds_read that follows a __builtin_amdgcn_global_load_lds -- it cannot tell the ring's
//     slots apart), which would serialise each row's memory latency with its MFMAs.  What it does not see it does not wait for; the
//     counted waits are the synchronisation.  M0 (the LDS base of a transfer) is saved and restored around the instruction.
//   * split-K over image ranges with the tiles of one range on ONE XCD (block b runs on XCD b % 8): an operand row comes from HBM
//     once and from that XCD's L2 for the other tiles.  fp32 partials in the weight's own [Co][3][3][Ci] order + a fixed-order
//     reduce that casts: deterministic (the library accumulates with atomics).
#include <type_traits>
#include "common.h"

namespace {

typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef short w3_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __attribute__((aligned(256))) unsigned char g_w3_zero_page[256];

template <int H_, int W_, int TCO_>
struct W3Cfg {
    static constexpr int H = H_, W = W_, TCO = TCO_;
    static constexpr int NCH = (W + 1 + 15) / 16;            // K steps per row; at least one zero column
    static constexpr int WP = 16 * NCH;
    static constexpr int NW = TCO / 16;                      // waves: (TCO / 32) channel blocks of dY x 2 of X
    static constexpr int XPITCH = 128;                       // bytes per position: 64 channels
    static constexpr int YPITCH = TCO * 2;
    static constexpr int XS = WP * XPITCH, YS = WP * YPITCH; // bytes per row slot
    static constexpr int TX = XS / 1024, TY = YS / 1024;     // 1 KB wave instructions per row
    static constexpr int T = TX + TY;
    static constexpr int IPW = (T + NW - 1) / NW;            // instructions a staging wave issues per row ...
    static constexpr int NIW = T / IPW;                      // ... and how many waves stage (the same count each: one vmcnt immediate)
    // ring: rows q - 1 .. q + 2 are read during row q, the staging of row q writes row q + D  =>  NS >= D + 2
    static constexpr int NS = (XS + YS) <= 6144 ? 16 : ((XS + YS) <= 12288 ? 12 : 8);
    static constexpr int D = NS - 2 > 10 ? 10 : NS - 2;
    static constexpr int FLY = (D - 3) * IPW;                // staging instructions a wave may have in flight behind a row's barrier
    static constexpr int LDS = NS * (XS + YS);
    static_assert(W < WP && LDS <= 160 * 1024 && D >= 4 && T % IPW == 0 && NIW <= NW && FLY <= 60, "padding / LDS / staging");
};

__device__ __forceinline__ w3_bf16x8 w3_tr(const char* p0, const char* p1) {
    const w3_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3_s16x4*)(p0));
    const w3_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3_s16x4*)(p1));
    union { w3_s16x4 h[2]; w3_bf16x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}

template <int N>
__device__ __forceinline__ void w3_land() {                  // this wave's staging but the N youngest instructions, then everybody's
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(N) : "memory");
}

struct W3Frags { w3_bf16x8 b[3], a[3]; };

// grid = tiles x splits (splits a multiple of 8), 64 * NW threads, C::LDS bytes of dynamic LDS.  part: [splits][Co][9][Ci] fp32.
// DBG (measurements only, results are wrong): 1 no staging, 2 no MFMAs, 3 no fragment reads (and no MFMAs)
template <class C, int DBG>
__global__ __launch_bounds__(C::NW * 64, 1) void cfl_conv3x3_wgrad_kernel(const u16* __restrict__ dy, const u16* __restrict__ x, int N,
                                                                           int Ci, int Co, int ips, float* __restrict__ part) {
    constexpr int H = C::H, W = C::W, NCH = C::NCH, WP = C::WP, NS = C::NS, D = C::D;
    extern __shared__ __attribute__((aligned(16))) char w3lds[];
    char* const ldsX = w3lds;
    char* const ldsY = w3lds + NS * C::XS;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w % (C::TCO / 32), wn = w / (C::TCO / 32);
    const int ntci = Ci >> 6, ntile = (Co / C::TCO) * ntci;
    const int b = blockIdx.x, xcd = b & 7, jb = b >> 3;
    const int tile = jb % ntile, split = xcd + 8 * (jb / ntile);
    const int co0 = (tile / ntci) * C::TCO, ci0 = (tile % ntci) * 64;
    const int img0 = split * ips;
    int nimg = N - img0;
    nimg = nimg < 0 ? 0 : (nimg > ips ? ips : nimg);
    const int R = nimg * H;

    // per-lane offsets of the transposing reads relative to a chunk's first position (lane = 16 g + p: K row 8 (g >> 1) + (p >> 2)
    // [+ 4 for the second read], channels 16 (g & 1) + 4 (p & 3) .. + 3 of the wave's 32-channel block).  The swizzles depend on
    // position bits below 16 only, so chunk c just adds 16 c positions -- except where a shifted position leaves the row: shift -1
    // in the first chunk (position -1 -> WP - 1) and shift +1 in the last one (position WP -> 0).
    const int g = lane >> 4, p = lane & 15;
    const int krow = 8 * (g >> 1) + (p >> 2);
    const int cx = wn * 32 + 16 * (g & 1) + 4 * (p & 3), cy = wm * 32 + 16 * (g & 1) + 4 * (p & 3);
    auto x_off = [&](int t) {                                 // t in [-1, 16]: 128-byte rows, 16-byte piece ^ ((t >> 1) & 1) << 2
        return t * C::XPITCH + ((((cx >> 3) ^ (((t >> 1) & 1) << 2)) & 7) << 4) + (cx & 7) * 2;
    };
    auto y_off = [&](int t) {                                 // t in [0, 15]
        if (C::YPITCH == 256) return t * 256 + ((((cy >> 3) ^ ((t & 3) << 2)) & 15) << 4) + (cy & 7) * 2;
        return t * 128 + ((((cy >> 3) ^ (((t >> 1) & 1) << 2)) & 7) << 4) + (cy & 7) * 2;
    };
    int offY[2], offX[3][2], offXfirst[2], offXlast[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = krow + 4 * j;
        offY[j] = y_off(k);
#pragma unroll
        for (int s = 0; s < 3; ++s) offX[s][j] = x_off(k + s - 1);
        offXfirst[j] = offX[0][j] + (k == 0 ? WP * C::XPITCH : 0);
        offXlast[j] = offX[2][j] - (k == 15 ? WP * C::XPITCH : 0);
    }

    // staging.  The rows of a workgroup's image range are CONSECUTIVE rows of the [N H, W, C] matrices, so a staging lane walks one
    // pointer per instruction it owns: source of row q = src + q * W * C (lanes of the zero columns w >= W read a page of zeros).
    // Instruction t of a row (t < TX: X positions 8 t .. 8 t + 7; else dY) belongs to wave t % NIW; waves >= NIW stage nothing.
    const unsigned ldsXa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsX;
    const unsigned ldsYa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsY;
    const u16* st_src[C::IPW];
    bool st_data[C::IPW];
    const bool stager = w < C::NIW;
#pragma unroll
    for (int i = 0; i < C::IPW; ++i) {
        const int t = (stager ? w : 0) + i * C::NIW;
        if (t < C::TX) {
            const int pos = 8 * t + (lane >> 3), pc = lane & 7;
            const int lp = pc ^ (((pos >> 1) & 1) << 2);
            st_data[i] = pos < W;
            st_src[i] = x + ((long long)img0 * H * W + pos) * Ci + ci0 + lp * 8;
        } else if (C::YPITCH == 256) {
            const int pos = 4 * (t - C::TX) + (lane >> 4), pc = lane & 15;
            const int lp = pc ^ ((pos & 3) << 2);
            st_data[i] = pos < W;
            st_src[i] = dy + ((long long)img0 * H * W + pos) * Co + co0 + lp * 8;
        } else {
            const int pos = 8 * (t - C::TX) + (lane >> 3), pc = lane & 7;
            const int lp = pc ^ (((pos >> 1) & 1) << 2);
            st_data[i] = pos < W;
            st_src[i] = dy + ((long long)img0 * H * W + pos) * Co + co0 + lp * 8;
        }
    }
    const long long stepX = (long long)W * Ci, stepY = (long long)W * Co;
    const void* zero_src = (const void*)g_w3_zero_page;
    asm volatile("" : "+v"(zero_src));                   // (kept in registers: the compiler would re-load the symbol's address per row)
    auto dma16 = [&](const void* src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto issue_row = [&](int q) {                        // called once per q, in increasing order
        if (DBG == 1 || !stager) return;
        const int slot = q % NS;
#pragma unroll
        for (int i = 0; i < C::IPW; ++i) {
            const int t = w + i * C::NIW;
            const bool isx = t < C::TX;
            const unsigned dst = isx ? ldsXa + slot * C::XS + t * 1024 : ldsYa + slot * C::YS + (t - C::TX) * 1024;
            if (q < R) dma16(st_data[i] ? (const void*)st_src[i] : zero_src, dst);
            st_src[i] += isx ? stepX : stepY;
        }
    };
    // rows <= q + 3 have landed for everybody (this wave has issued rows <= q + D, IPW instructions each, in order)
    auto land = [&](int q) {
        if (q + D < R) w3_land<C::FLY>();
        else w3_land<0>();
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    auto mma = [&](f32x16& d, const w3_bf16x8& a, const w3_bf16x8& bb) {
        if (DBG < 2) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, d, 0, 0, 0);
    };
    auto read_x = [&](int q, int c, w3_bf16x8 (&f)[3]) {      // the three shifts of X row q, chunk c
        if (DBG == 3) return;
        const char* xs = ldsX + (q % NS) * C::XS + c * (16 * C::XPITCH);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s == 0 && c == 0) f[s] = w3_tr(xs + offXfirst[0], xs + offXfirst[1]);
            else if (s == 2 && c == NCH - 1) f[s] = w3_tr(xs + offXlast[0], xs + offXlast[1]);
            else f[s] = w3_tr(xs + offX[s][0], xs + offX[s][1]);
        }
    };
    auto read_y = [&](int q, int c, w3_bf16x8& f) {           // dY row q, chunk c
        if (DBG == 3) return;
        const char* ys = ldsY + ((q + NS) % NS) * C::YS + c * (16 * C::YPITCH);
        f = w3_tr(ys + offY[0], ys + offY[1]);
    };

#pragma unroll
    for (int q = 0; q < D; ++q) issue_row(q);

    if constexpr (NCH == 1) {
        // ---- one K step per row: the image is unrolled, the row index is static ------------------------------------------------
        // entering row h (stream row q): A = dY row q, B0 = X row q, Bp = X row q + 1 [, Bm = X row q - 1] are in registers (read
        // during the row before), rows <= q + 2 have landed.
        if (R > 0) {
            if (D - 1 < R) w3_land<(D - 3) * C::IPW>(); else w3_land<0>();
            w3_bf16x8 A, Bm[3], B0[3], Bp[3];
            read_y(0, 0, A);
            read_x(0, 0, B0);
            read_x(1, 0, Bp);
            int q = 0;
            for (int im = 0; im < nimg; ++im) {
#pragma unroll
                for (int h = 0; h < H; ++h, ++q) {
                    issue_row(q + D);                     // into the slot of row q + D - NS <= q - 2: dead
                    w3_bf16x8 nA, nB0[3], nBp[3];
                    read_y(q + 1, 0, nA);                 // (behind the last row: slots nobody uses)
                    if (h + 2 < H) read_x(q + 2, 0, nBp);
                    if (h == H - 1) { read_x(q + 1, 0, nB0); read_x(q + 2, 0, nBp); }
                    if (h > 0) {
#pragma unroll
                        for (int s = 0; s < 3; ++s) mma(acc[0][s], A, Bm[s]);
                    }
#pragma unroll
                    for (int s = 0; s < 3; ++s) mma(acc[1][s], A, B0[s]);
                    if (h + 1 < H) {
#pragma unroll
                        for (int s = 0; s < 3; ++s) mma(acc[2][s], A, Bp[s]);
                    }
                    A = nA;
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        if (h == H - 1) { B0[s] = nB0[s]; Bp[s] = nBp[s]; }
                        else { Bm[s] = B0[s]; B0[s] = Bp[s]; if (h + 2 < H) Bp[s] = nBp[s]; }
                    }
                    if (q + 1 < R) land(q);
                }
            }
        }
    } else {
        // ---- several K steps per row: a row loop; unit (q, c) = the X fragments of row q and the dY fragments of rows q - 1, q,
        // q + 1, chunk c.  Two fragment sets used alternately: unit u computes on F[u & 1] while unit u + 1 is read into the other
        // (NCH is even, so a row starts on F[0]).  The body exists three times -- first / middle / last row of an image -- so that the
        // MFMAs of the rows that do not exist are not issued and nothing inside a row is conditional.
        static_assert(NCH == 1 || (NCH & 1) == 0, "row bodies assume an even number of units per row");
        if (R > 0) {
            if (D - 1 < R) w3_land<(D - 3) * C::IPW>(); else w3_land<0>();
            W3Frags F[2];
            auto load_unit = [&](int q, int c, W3Frags& f) {
                read_x(q, c, f.b);
#pragma unroll
                for (int d = 0; d < 3; ++d) read_y(q - 1 + d, c, f.a[d]);
            };
            load_unit(0, 0, F[0]);
            auto row = [&](auto kind, int q) {            // kind: 0 first row of an image, 1 middle, 2 last
                constexpr int K = decltype(kind)::value;
                issue_row(q + D);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + 1 < NCH) load_unit(q, c + 1, F[(c + 1) & 1]);
                    else load_unit(q + 1, 0, F[0]);       // rows q .. q + 2: landed since the last barrier
                    const W3Frags& f = F[c & 1];
                    // tap kh pairs X row h with the dY row h - kh + 1: a[d] (row h - 1 + d) belongs to kh = 2 - d
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        if ((K == 0 && d == 0) || (K == 2 && d == 2)) continue;
#pragma unroll
                        for (int s = 0; s < 3; ++s) mma(acc[2 - d][s], f.a[d], f.b[s]);
                    }
                }
                if (q + 1 < R) land(q);
            };
            int q = 0;
            for (int im = 0; im < nimg; ++im) {
                row(std::integral_constant<int, 0>(), q); ++q;
                for (int h = 1; h < H - 1; ++h, ++q) row(std::integral_constant<int, 1>(), q);
                row(std::integral_constant<int, 2>(), q); ++q;
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

// dw[e] (bf16) = sum over splits of part[s][e] in a FIXED order: a thread sums every 4th split for 4 elements (four loads in
// flight), the four partial sums are combined through LDS as (0 + 1) + (2 + 3).  64 element groups x 4 split lanes per block.
__global__ __launch_bounds__(256) void cfl_conv3x3_wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, long long n,
                                                                       u16* __restrict__ dw) {
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
    f32x4 t;
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = (red[0][eg][j] + red[1][eg][j]) + (red[2][eg][j] + red[3][eg][j]);
    unsigned o[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        unsigned lo = __float_as_uint(t[2 * k]), hi = __float_as_uint(t[2 * k + 1]);
        lo += 0x7fffu + ((lo >> 16) & 1u);
        hi += 0x7fffu + ((hi >> 16) & 1u);
        o[k] = (lo >> 16) | (hi & 0xffff0000u);
    }
    *reinterpret_cast<uint2*>(dw + e) = make_uint2(o[0], o[1]);
}

inline int& w3_splits_override() {
    static int v = getenv("CFL_WGRAD3_SPLITS") ? atoi(getenv("CFL_WGRAD3_SPLITS")) : 0;
    return v;
}
inline int& w3_dbg() {
    static int v = 0;
    return v;
}

// 0: shape not taken, else the channel tile of dY (64 / 128)
inline int w3_tco(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || Ci <= 0 || Co <= 0 || Ci % 64 != 0 || H != W) return 0;
    if (H == 56) return Co == 64 ? 64 : 0;             // (a 128-channel tile of 64-position rows does not fit the LDS rings)
    if (H != 7 && H != 14 && H != 28) return 0;
    return Co % 128 == 0 ? 128 : 0;
}

inline int w3_nsplit(int N, int Ci, int Co, int tco) {
    const int ntile = (Co / tco) * (Ci / 64);
    int ns = w3_splits_override() > 0 ? w3_splits_override() : 256 / ntile;
    ns = (ns / 8) * 8;
    if (ns < 8) ns = 8;
    while (ns > 8 && ns - 8 >= N) ns -= 8;             // no more image ranges than images (rounded up to the 8 XCDs)
    return ns;
}

template <class C, int DBG = 0>
int w3_launch(const u16* dy, const u16* x, int N, int Ci, int Co, int ns, float* part, hipStream_t stream) {
    const int ntile = (Co / C::TCO) * (Ci / 64);
    CFL_SET_LDS((cfl_conv3x3_wgrad_kernel<C, DBG>), C::LDS);
    CFL_LAUNCH(K_CONV3_WGRAD, (cfl_conv3x3_wgrad_kernel<C, DBG>), dim3(ntile * ns), dim3(C::NW * 64), C::LDS, stream, dy, x, N, Ci, Co,
               cfl_cdiv(N, ns), part);
    return 0;
}

}  // namespace

extern "C" int cfl_conv3x3_wgrad_supported(int N, int H, int W, int Ci, int Co) { return w3_tco(N, H, W, Ci, Co) ? 1 : 0; }

extern "C" int cfl_conv3x3_wgrad_splits(int splits) {
    const int old = w3_splits_override();
    if (splits >= 0) w3_splits_override() = splits;
    return old;
}

extern "C" int cfl_conv3x3_wgrad_debug(int mode) {
    const int old = w3_dbg();
    if (mode >= 0) w3_dbg() = mode;
    return old;
}

extern "C" size_t cfl_conv3x3_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co) {
    const int tco = w3_tco(N, H, W, Ci, Co);
    if (!tco) return 0;
    return cfl_align256((size_t)w3_nsplit(N, Ci, Co, tco) * Co * 9 * Ci * sizeof(float));
}

extern "C" int cfl_conv3x3_wgrad(const void* dy, const void* x, int N, int H, int W, int Ci, int Co, void* dw, void* ws, void* stream_) {
    if (!dy || !x || !dw || !ws) return CFL_EINVAL;
    const int tco = w3_tco(N, H, W, Ci, Co);
    if (!tco || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw | (uintptr_t)ws) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const int ns = w3_nsplit(N, Ci, Co, tco);
    float* part = (float*)ws;
    const u16 *d = (const u16*)dy, *xx = (const u16*)x;
    int rc;
    if (H == 7) rc = w3_launch<W3Cfg<7, 7, 128>>(d, xx, N, Ci, Co, ns, part, stream);
    else if (H == 14) {
        switch (w3_dbg()) {                              // (measurement variants of the layer3 shape only)
            case 1: rc = w3_launch<W3Cfg<14, 14, 128>, 1>(d, xx, N, Ci, Co, ns, part, stream); break;
            case 2: rc = w3_launch<W3Cfg<14, 14, 128>, 2>(d, xx, N, Ci, Co, ns, part, stream); break;
            case 3: rc = w3_launch<W3Cfg<14, 14, 128>, 3>(d, xx, N, Ci, Co, ns, part, stream); break;
            default: rc = w3_launch<W3Cfg<14, 14, 128>>(d, xx, N, Ci, Co, ns, part, stream); break;
        }
    } else if (H == 28) rc = w3_launch<W3Cfg<28, 28, 128>>(d, xx, N, Ci, Co, ns, part, stream);
    else rc = w3_launch<W3Cfg<56, 56, 64>>(d, xx, N, Ci, Co, ns, part, stream);
    if (rc) return rc;
    const long long n = (long long)Co * 9 * Ci;
    CFL_LAUNCH(K_CONV3_WGRAD_REDUCE, cfl_conv3x3_wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, stream,
               (const float*)part, ns, n, (u16*)dw);
    return 0;
}
