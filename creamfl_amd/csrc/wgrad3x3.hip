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
//   * the nine taps share their operands.  A unit of work is (X row r, chunk c): its three w-shifted X fragments meet the dY
//     fragments of rows r + 1, r, r - 1 (taps kh = 0, 1, 2) -- 6 fragment reads for 9 MFMAs.  A shifted fragment is the same
//     transposing read with per-lane addresses moved by one position MODULO the padded row: position -1 reads position WP - 1 (a zero
//     column), position WP reads position 0 (finite, and multiplied by dY's zero column) -- all reads stay inside the row's slot and
//     the zero padding of the convolution comes from the data itself.  Rows -1 and H do not exist: their MFMAs are skipped.
//   * a workgroup owns a TCO (128: 8 waves, or 64: 4 waves) x 64 x 9 tile of dW for a range of images: 144 accumulator registers per
//     lane.  Rows stream through two LDS rings (X, dY) by LDS-DMA with a source-side XOR swizzle (conflict-free transposing reads),
//     D rows ahead.  The fragments of unit u + 1 are read while the MFMAs of unit u run (also across the row's barrier), so the matrix
//     pipe starts right behind a barrier; ONE barrier per row, counted vmcnt.
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
    static constexpr int IPW = (T + NW - 1) / NW;            // most instructions a wave issues per row
    // ring: rows q - 1 .. q + 2 are live during step q, the staging of step q writes row q + D  =>  NS >= D + 2
    static constexpr int NS = (XS + YS) <= 6144 ? 16 : ((XS + YS) <= 12288 ? 12 : 8);
    static constexpr int D = NS - 2 > 10 ? 10 : NS - 2;
    static constexpr int LDS = NS * (XS + YS);
    static_assert(W < WP && LDS <= 160 * 1024 && D >= 4, "row padding / LDS budget / staging distance");
};

__device__ __forceinline__ w3_bf16x8 w3_tr(const char* p0, const char* p1) {
    const w3_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3_s16x4*)(p0));
    const w3_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3_s16x4*)(p1));
    union { w3_s16x4 h[2]; w3_bf16x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}

struct W3Frags { w3_bf16x8 b[3], a[3]; };

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate); a count without a case waits for everything
#define W3_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void w3_wait_vm(int n) {
    switch (n) {
        W3_VM_CASE(1) W3_VM_CASE(2) W3_VM_CASE(3) W3_VM_CASE(4) W3_VM_CASE(5) W3_VM_CASE(6) W3_VM_CASE(7) W3_VM_CASE(8)
        W3_VM_CASE(9) W3_VM_CASE(10) W3_VM_CASE(12) W3_VM_CASE(14) W3_VM_CASE(16) W3_VM_CASE(18) W3_VM_CASE(20) W3_VM_CASE(21)
        W3_VM_CASE(24) W3_VM_CASE(28)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// grid = tiles x splits (splits a multiple of 8), 64 * NW threads, C::LDS bytes of dynamic LDS.  part: [splits][Co][9][Ci] fp32.
template <class C>
__global__ __launch_bounds__(C::NW * 64, 1) void cfl_conv3x3_wgrad_kernel(const u16* __restrict__ dy, const u16* __restrict__ x, int N,
                                                                           int Ci, int Co, int ips, float* __restrict__ part) {
    constexpr int H = C::H, W = C::W, NCH = C::NCH, WP = C::WP, NW = C::NW, NS = C::NS, D = C::D;
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
    // Instruction t of a row (t < TX: X positions 8 t .. 8 t + 7; else dY) belongs to wave t % NW.
    const unsigned ldsXa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsX;
    const unsigned ldsYa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsY;
    const u16* st_src[C::IPW];
    bool st_data[C::IPW];
    int my_ipw = 0;
#pragma unroll
    for (int i = 0; i < C::IPW; ++i) {
        const int t = w + i * NW;
        st_src[i] = x;
        st_data[i] = false;
        if (t < C::T) {
            ++my_ipw;
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
#pragma unroll
        for (int i = 0; i < C::IPW; ++i) {
            const int t = w + i * NW;
            if (t < C::T) {
                const bool isx = t < C::TX;
                const unsigned dst = isx ? ldsXa + (q % NS) * C::XS + t * 1024 : ldsYa + (q % NS) * C::YS + (t - C::TX) * 1024;
                if (q < R) dma16(st_data[i] ? (const void*)st_src[i] : zero_src, dst);
                st_src[i] += isx ? stepX : stepY;
            }
        }
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    // unit (q, c): the X fragments of row q (three shifts) and the dY fragments of rows q - 1, q, q + 1, chunk c.  All six are read
    // unconditionally: at the first / last row of an image the neighbour slot holds another image's row (or nothing yet) and the
    // fragment is simply not used -- a conditional read would merge with an undefined value and pull the wait for THIS prefetch in
    // front of the current unit's MFMAs.
    auto load_unit = [&](int q, int c, W3Frags& f) {
        const char* xs = ldsX + (q % NS) * C::XS + c * (16 * C::XPITCH);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s == 0 && c == 0) f.b[s] = w3_tr(xs + offXfirst[0], xs + offXfirst[1]);
            else if (s == 2 && c == NCH - 1) f.b[s] = w3_tr(xs + offXlast[0], xs + offXlast[1]);
            else f.b[s] = w3_tr(xs + offX[s][0], xs + offX[s][1]);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const char* ys = ldsY + ((q + NS - 1 + d) % NS) * C::YS + c * (16 * C::YPITCH);
            f.a[d] = w3_tr(ys + offY[0], ys + offY[1]);
        }
    };
    // tap kh pairs X row h with the dY row h - kh + 1: a[d] (row h - 1 + d) belongs to kh = 2 - d.  A dY row outside the image is
    // replaced by zeros instead of skipping its MFMAs (2 of 3 H fragment rows): straight-line code lets the compiler count the
    // outstanding LDS reads exactly -- behind a branch it waits for ALL of them, i.e. for the prefetch of the next unit.
    auto compute_unit = [&](int h, const W3Frags& f) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const bool in = (unsigned)(h - 1 + d) < (unsigned)H;
            union { w3_bf16x8 v; unsigned u[4]; } a;
            a.v = f.a[d];
#pragma unroll
            for (int k = 0; k < 4; ++k) a.u[k] = in ? a.u[k] : 0u;
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[2 - d][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, f.b[s], acc[2 - d][s], 0, 0, 0);
        }
    };
    // rows <= `upto` have landed for everybody: this wave has issued rows <= `issued`, my_ipw instructions each, in order
    auto land = [&](int upto, int issued) {
        const int last = issued < R - 1 ? issued : R - 1;
        const int fly = last - upto;
        w3_wait_vm(fly > 0 ? fly * my_ipw : 0);
        asm volatile("s_barrier" ::: "memory");          // (bare: every LDS read of the finished step was waited for by its MFMAs)
    };

#pragma unroll
    for (int q = 0; q < D; ++q) issue_row(q);
    if (R > 0) {
        land(2, D - 1);
        // two fragment sets, used alternately (no register copies): unit (q, c) computes on F[(P0 + c) & 1] while the next unit's
        // fragments arrive in the other set; P0 = parity of the row's first unit (rows alternate when a row is ONE unit)
        W3Frags F[2];
        load_unit(0, 0, F[0]);
        auto row = [&](auto p0, int q, int h) {
            constexpr int P0 = decltype(p0)::value;
            issue_row(q + D);                            // into the slot of row q + D - NS <= q - 2: dead
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                // (unconditional: behind the last row this reads a slot nobody uses -- see load_unit)
                if (c + 1 < NCH) load_unit(q, c + 1, F[(P0 + c + 1) & 1]);
                else load_unit(q + 1, 0, F[(P0 + c + 1) & 1]);                     // rows q .. q + 2: landed since the last barrier
                compute_unit(h, F[(P0 + c) & 1]);
            }
            if (q + 1 < R) land(q + 3, q + D);
        };
        int h = 0, q = 0;
        for (; q + 1 < R; q += 2) {
            row(std::integral_constant<int, 0>(), q, h);
            h = h + 1 == H ? 0 : h + 1;
            row(std::integral_constant<int, NCH & 1>(), q + 1, h);
            h = h + 1 == H ? 0 : h + 1;
        }
        if (q < R) row(std::integral_constant<int, 0>(), q, h);
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

// dw[e] (bf16) = sum over splits, in split order, of part[s][e]
__global__ __launch_bounds__(256) void cfl_conv3x3_wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, long long n,
                                                                       u16* __restrict__ dw) {
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(part + e);
    for (int k = 1; k < nsplit; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(part + (long long)k * n + e);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    unsigned o[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        unsigned lo = __float_as_uint(s[2 * k]), hi = __float_as_uint(s[2 * k + 1]);
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

template <class C>
int w3_launch(const u16* dy, const u16* x, int N, int Ci, int Co, int ns, float* part, hipStream_t stream) {
    const int ntile = (Co / C::TCO) * (Ci / 64);
    CFL_SET_LDS((cfl_conv3x3_wgrad_kernel<C>), C::LDS);
    CFL_LAUNCH(K_CONV3_WGRAD, (cfl_conv3x3_wgrad_kernel<C>), dim3(ntile * ns), dim3(C::NW * 64), C::LDS, stream, dy, x, N, Ci, Co,
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
    else if (H == 14) rc = w3_launch<W3Cfg<14, 14, 128>>(d, xx, N, Ci, Co, ns, part, stream);
    else if (H == 28) rc = w3_launch<W3Cfg<28, 28, 128>>(d, xx, N, Ci, Co, ns, part, stream);
    else rc = w3_launch<W3Cfg<56, 56, 64>>(d, xx, N, Ci, Co, ns, part, stream);
    if (rc) return rc;
    const long long n = (long long)Co * 9 * Ci;
    CFL_LAUNCH(K_CONV3_WGRAD_REDUCE, cfl_conv3x3_wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream,
               (const float*)part, ns, n, (u16*)dw);
    return 0;
}
