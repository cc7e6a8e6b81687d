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
// The problem is a TN GEMM [Co] x [9 Ci] with the reduction over the 50 176 positions of the batch -- the slow axis of both
// channels_last operands -- and 59 GFLOP per layer3 block against 51 MB of operands: MFMA-bound.  Design:
//   * v_mfma_f32_32x32x16_bf16; a K step is ONE image row padded to 16 positions (w = W .. 15 are zeros, written by the staging from
//     a page of zeros).  Both operands are read TRANSPOSED out of LDS (ds_read_b64_tr_b16: a lane ends up with 8 consecutive
//     positions of one channel = the MFMA operand layout), so no VALU touches an operand.
//   * the nine taps share their operands: the dY fragment of row h meets the X fragments of rows h - 1, h, h + 1 (kept in registers
//     as a rolling window: a row is read from LDS once) in three w-shifts each.  A shifted fragment is the same transposing read
//     with per-lane addresses moved by one position MODULO the 16-position row: position -1 reads position 15 (a zero column),
//     position 16 reads position 0 (finite, and multiplied by dY's zero column) -- all reads stay inside the row's slot and the zero
//     padding of the convolution comes from the data itself.  Rows -1 and H are not read at all (their MFMAs are skipped).
//   * a workgroup (8 waves = 4 channel blocks of dY x 2 of X) owns a 128 (co) x 64 (ci) x 9 tile of dW for a range of images:
//     144 accumulator registers per lane.  Rows stream through two 16-slot LDS rings (X: 2 KB, dY: 4 KB per row) by LDS-DMA
//     (source-side XOR swizzle: conflict-free transposing reads), 8 rows ahead, one barrier per row, counted vmcnt.
//   * split-K over image ranges with the tiles of one range on ONE XCD (block b runs on XCD b % 8): an operand row comes from HBM
//     once and from that XCD's L2 for the other tiles.  fp32 partials in the weight's own [Co][3][3][Ci] order + a fixed-order
//     reduce that casts: deterministic (the library accumulates with atomics).
#include "common.h"

namespace {

typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef short w3_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __attribute__((aligned(256))) unsigned char g_w3_zero_page[256];

constexpr int W3_SLOTS = 16;              // rows per ring
constexpr int W3_D = 8;                   // rows the staging runs ahead
constexpr int W3_XS = 16 * 128;           // bytes per X slot: 16 positions x 64 channels
constexpr int W3_YS = 16 * 256;           // bytes per dY slot: 16 positions x 128 channels
constexpr int W3_LDS = W3_SLOTS * (W3_XS + W3_YS);

__device__ __forceinline__ w3_bf16x8 w3_tr(const char* p0, const char* p1) {
    const w3_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3_s16x4*)(p0));
    const w3_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3_s16x4*)(p1));
    union { w3_s16x4 h[2]; w3_bf16x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}

// grid = tiles x splits (splits a multiple of 8), 512 threads, W3_LDS bytes of dynamic LDS.
// part: [splits][Co][9][Ci] fp32.
template <int H, int W>
__global__ __launch_bounds__(512, 1) void cfl_conv3x3_wgrad_kernel(const u16* __restrict__ dy, const u16* __restrict__ x, int N, int Ci,
                                                                   int Co, int ips, float* __restrict__ part) {
    static_assert(W >= 1 && W <= 15, "one 16-position K step per image row, at least one zero column");
    extern __shared__ __attribute__((aligned(16))) char w3lds[];
    char* const ldsX = w3lds;
    char* const ldsY = w3lds + W3_SLOTS * W3_XS;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w & 3, wn = w >> 2;
    const int ntci = Ci >> 6, ntile = (Co >> 7) * ntci;
    const int b = blockIdx.x, xcd = b & 7, jb = b >> 3;
    const int tile = jb % ntile, split = xcd + 8 * (jb / ntile);
    const int co0 = (tile / ntci) * 128, ci0 = (tile % ntci) * 64;
    const int img0 = split * ips;
    int nimg = N - img0;
    nimg = nimg < 0 ? 0 : (nimg > ips ? ips : nimg);
    const int R = nimg * H;

    // per-lane offsets of the transposing reads inside a slot (lane = 16 g + p: K row 8 (g >> 1) + (p >> 2) [+ 4 for the second
    // read], channels 16 (g & 1) + 4 (p & 3) .. + 3 of the wave's 32-channel block)
    const int g = lane >> 4, p = lane & 15;
    int offY;
    int offX[3][2];
    {
        const int row = 8 * (g >> 1) + (p >> 2);
        const int cy = wm * 32 + 16 * (g & 1) + 4 * (p & 3);
        offY = row * 256 + ((((cy >> 3) ^ ((row & 3) << 2)) & 15) << 4) + (cy & 7) * 2;
        const int cx = wn * 32 + 16 * (g & 1) + 4 * (p & 3);
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pos = (row + 4 * j + s - 1) & 15;
                offX[s][j] = pos * 128 + ((((cx >> 3) ^ (((pos >> 1) & 1) << 2)) & 7) << 4) + (cx & 7) * 2;
            }
    }

    // one image row of both operands into slot q % 16: 2 + 4 wave instructions of 1 KB, waves 0 .. 5 issue one each.
    // The LDS-DMA is issued from inline assembly ON PURPOSE: the compiler orders every LDS read behind every LDS-DMA it knows of
    // (s_waitcnt vmcnt(0) before the first ds_read that follows a __builtin_amdgcn_global_load_lds -- it cannot tell the ring's slots
    // apart), which would serialise each row's memory latency with its MFMAs.  What it does not see it does not wait for; the
    // counted waits below are the synchronisation.  M0 (the LDS base of the transfer) is saved and restored around the instruction.
    const unsigned ldsXa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsX;
    const unsigned ldsYa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsY;
    auto dma16 = [&](const void* src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    // The rows of a workgroup's image range are CONSECUTIVE rows of the [N H, W, C] matrices, so a staging lane walks one pointer:
    // its source for row q is lane_src + q * W * C (the lanes of the zero columns w >= W read a page of zeros instead).
    const u16* lane_src = nullptr;
    long long row_step = 0;
    bool lane_data = false;
    unsigned lane_dst = 0;                               // LDS base of this wave's instruction inside a slot
    int slot_bytes = 0;
    if (w < 2) {
        const int pos = 8 * w + (lane >> 3), pc = lane & 7;
        const int lp = pc ^ (((pos >> 1) & 1) << 2);
        lane_data = pos < W;
        lane_src = x + ((long long)img0 * H * W + pos) * Ci + ci0 + lp * 8;
        row_step = (long long)W * Ci;
        lane_dst = ldsXa + w * 1024;
        slot_bytes = W3_XS;
    } else if (w < 6) {
        const int j = w - 2;
        const int pos = 4 * j + (lane >> 4), pc = lane & 15;
        const int lp = pc ^ ((pos & 3) << 2);
        lane_data = pos < W;
        lane_src = dy + ((long long)img0 * H * W + pos) * Co + co0 + lp * 8;
        row_step = (long long)W * Co;
        lane_dst = ldsYa + j * 1024;
        slot_bytes = W3_YS;
    }
    const void* zero_src = (const void*)g_w3_zero_page;
    asm volatile("" : "+v"(zero_src));                   // (kept in registers: the compiler would re-load the symbol's address per row)
    auto issue_row = [&](int q) {                        // called once per q, in increasing order
        if (w < 6) {
            if (q < R) dma16(lane_data ? (const void*)lane_src : zero_src, lane_dst + (q & 15) * slot_bytes);
            lane_src += row_step;
        }
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

#pragma unroll
    for (int q = 0; q < W3_D; ++q) issue_row(q);

    int q = 0;
    for (int im = 0; im < nimg; ++im) {
        w3_bf16x8 Bm[3], B0[3], Bp[3];
#pragma unroll
        for (int h = 0; h < H; ++h, ++q) {
            // rows <= q + 1 have landed: a staging wave has one instruction per row in flight, rows up to q + 7 are issued
            // (a bare s_barrier: every LDS read of the previous step was waited for by the MFMAs that consumed it)
            if (q + W3_D < R) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            issue_row(q + W3_D);               // into the slot of row q - 8: dead since step q - 8
            const char* ys = ldsY + (q & 15) * W3_YS;
            const w3_bf16x8 A = w3_tr(ys + offY, ys + offY + 4 * 256);
            if (h == 0) {
                const char* xs = ldsX + (q & 15) * W3_XS;
#pragma unroll
                for (int s = 0; s < 3; ++s) B0[s] = w3_tr(xs + offX[s][0], xs + offX[s][1]);
            }
            if (h + 1 < H) {
                const char* xs = ldsX + ((q + 1) & 15) * W3_XS;
#pragma unroll
                for (int s = 0; s < 3; ++s) Bp[s] = w3_tr(xs + offX[s][0], xs + offX[s][1]);
            }
            if (h > 0) {
#pragma unroll
                for (int s = 0; s < 3; ++s) acc[0][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bm[s], acc[0][s], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[1][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B0[s], acc[1][s], 0, 0, 0);
            if (h + 1 < H) {
#pragma unroll
                for (int s = 0; s < 3; ++s) acc[2][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bp[s], acc[2][s], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) { Bm[s] = B0[s]; B0[s] = Bp[s]; }
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

inline bool w3_shape_ok(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || Ci <= 0 || Co <= 0 || Ci % 64 != 0 || Co % 128 != 0) return false;
    return (H == 14 && W == 14) || (H == 7 && W == 7);
}

inline int w3_nsplit(int N, int Ci, int Co) {
    const int ntile = (Co / 128) * (Ci / 64);
    int ns = w3_splits_override() > 0 ? w3_splits_override() : 256 / ntile;
    ns = (ns / 8) * 8;
    if (ns < 8) ns = 8;
    while (ns > 8 && ns - 8 >= N) ns -= 8;             // no more image ranges than images (rounded up to the 8 XCDs)
    return ns;
}

}  // namespace

extern "C" int cfl_conv3x3_wgrad_supported(int N, int H, int W, int Ci, int Co) { return w3_shape_ok(N, H, W, Ci, Co) ? 1 : 0; }

extern "C" int cfl_conv3x3_wgrad_splits(int splits) {
    const int old = w3_splits_override();
    if (splits >= 0) w3_splits_override() = splits;
    return old;
}

extern "C" size_t cfl_conv3x3_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co) {
    if (!w3_shape_ok(N, H, W, Ci, Co)) return 0;
    return cfl_align256((size_t)w3_nsplit(N, Ci, Co) * Co * 9 * Ci * sizeof(float));
}

extern "C" int cfl_conv3x3_wgrad(const void* dy, const void* x, int N, int H, int W, int Ci, int Co, void* dw, void* ws, void* stream_) {
    if (!dy || !x || !dw || !ws) return CFL_EINVAL;
    if (!w3_shape_ok(N, H, W, Ci, Co) || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw | (uintptr_t)ws) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const int ns = w3_nsplit(N, Ci, Co);
    const int ips = cfl_cdiv(N, ns);
    const int ntile = (Co / 128) * (Ci / 64);
    float* part = (float*)ws;
    if (H == 14) {
        CFL_SET_LDS((cfl_conv3x3_wgrad_kernel<14, 14>), W3_LDS);
        CFL_LAUNCH(K_CONV3_WGRAD, (cfl_conv3x3_wgrad_kernel<14, 14>), dim3(ntile * ns), dim3(512), W3_LDS, stream, (const u16*)dy,
                   (const u16*)x, N, Ci, Co, ips, part);
    } else {
        CFL_SET_LDS((cfl_conv3x3_wgrad_kernel<7, 7>), W3_LDS);
        CFL_LAUNCH(K_CONV3_WGRAD, (cfl_conv3x3_wgrad_kernel<7, 7>), dim3(ntile * ns), dim3(512), W3_LDS, stream, (const u16*)dy,
                   (const u16*)x, N, Ci, Co, ips, part);
    }
    const long long n = (long long)Co * 9 * Ci;
    CFL_LAUNCH(K_CONV3_WGRAD_REDUCE, cfl_conv3x3_wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream,
               (const float*)part, ns, n, (u16*)dw);
    return 0;
}
