// wgrad1x1.hip -- weight gradient of the 1 x 1 / stride 1 convolutions of the ResNet trunk (bf16, channels_last).
//
//   dW[co][ci] = sum over the M = N H W positions of dY[m, co] * X[m, ci]
//
// torchvision Bottleneck.conv1 / conv3 / downsample[0] inside src/networks/models/image_encoder.py:27-36; the reference leaves them to
// cuDNN.  A TN GEMM with a tiny output (16 K .. 1 M elements) and the reduction over the slow axis of both operands: HBM-bound (26 GFLOP
// against 128 MB at layer3's shape).  The library's kernel (CK batched GEMM with fp32 atomics + a zero fill + a cast: 66 launches,
// 6.6 ms of side-stream kernel time and ~21 GB per step, profiles/r6_step_bytes.json) re-reads its operands; three earlier hand-written
// attempts tied it alone and lost inside the step (docs/history/DESIGN_r1-r4.md): 128 x 128 tiles re-read the operands 2 x / 8 x through
// the LDS-DMA path, and -- found in round 6 -- the compiler had put an s_waitcnt vmcnt(0) in front of the first LDS read behind every
// LDS-DMA it knew of, so their "rings" never had more than one stage in flight.  This kernel is built on what csrc/wgrad3x3.hip
// established:
//   * both operands staged ROW-MAJOR by LDS-DMA issued from inline assembly (invisible to the compiler's wait insertion), 16-row stages
//     in a ring 6-10 stages deep, counted vmcnt + one bare barrier per stage;
//   * MFMA fragments read TRANSPOSED out of LDS (ds_read_b64_tr_b16, source-side XOR swizzle), the fragments of stage q + 1 read while
//     the MFMAs of stage q run;
//   * LARGE tiles -- up to 256 x 256 of dW per workgroup (128 accumulator registers per lane, 8 waves) -- so that an operand passes
//     the LDS-DMA path once or twice instead of 2 x / 8 x, and FEW workgroups (~128: the kernel lives on a side stream and is bound
//     by bytes, not by CUs), which keeps the split-K partials at a fraction of the operand bytes;
//   * split-K over row ranges with the tiles of one range on ONE XCD; fp32 partials + a fixed-order reduce that casts:
//     deterministic, unlike the library's atomics.
#include <type_traits>
#include "common.h"

namespace {

typedef __bf16 w1_bf16x8 __attribute__((ext_vector_type(8)));
typedef short w1_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __attribute__((aligned(256))) unsigned char g_w1_zero_page[256];

template <int TA_, int TB_, int WA_, int WB_>
struct W1Cfg {
    static constexpr int TA = TA_, TB = TB_, WA = WA_, WB = WB_;
    static constexpr int NW = WA * WB;
    static constexpr int MA = TA / WA / 32, MB = TB / WB / 32;      // 32 x 32 MFMA tiles per wave
    static constexpr int KS = 16;                                   // rows per stage = one MFMA K step
    static constexpr int APITCH = TA * 2, BPITCH = TB * 2;          // bytes per staged row
    static constexpr int AS = KS * APITCH, BS = KS * BPITCH;
    static constexpr int TAI = AS / 1024, TBI = BS / 1024, T = TAI + TBI;
    static constexpr int IPW = (T + NW - 1) / NW;
    static constexpr int NS0 = (128 * 1024) / (AS + BS);
    static constexpr int NS = NS0 > 12 ? 12 : NS0;
    static constexpr int D = NS - 2;
    static constexpr int LDS = NS * (AS + BS);
    static_assert(MA >= 1 && MB >= 1 && MA * MB * 16 <= 128 && D >= 4 && LDS <= 160 * 1024, "tile / ring");
    static_assert(TA % (32 * WA) == 0 && TB % (32 * WB) == 0 && AS % 1024 == 0 && BS % 1024 == 0, "tile shape");
};

__device__ __forceinline__ w1_bf16x8 w1_tr(const char* p0, const char* p1) {
    const w1_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w1_s16x4*)(p0));
    const w1_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w1_s16x4*)(p1));
    union { w1_s16x4 h[2]; w1_bf16x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}

#define W1_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void w1_wait_vm(int n) {
    switch (n) {
        W1_VM_CASE(1) W1_VM_CASE(2) W1_VM_CASE(3) W1_VM_CASE(4) W1_VM_CASE(5) W1_VM_CASE(6) W1_VM_CASE(7) W1_VM_CASE(8)
        W1_VM_CASE(9) W1_VM_CASE(10) W1_VM_CASE(12) W1_VM_CASE(14) W1_VM_CASE(16) W1_VM_CASE(18) W1_VM_CASE(20)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// 16-byte piece of a staged row: XOR swizzle that spreads the four rows of a transposing half-wave read over the 64 banks
template <int PITCH>
__device__ __forceinline__ int w1_swz(int row) { return PITCH == 128 ? (((row >> 1) & 1) << 2) : ((row & 3) << 2); }

// grid = tiles x splits (splits a multiple of 8), 64 NW threads, C::LDS bytes of dynamic LDS.  part: [splits][Co][Ci] fp32.
template <class C>
__global__ __launch_bounds__(C::NW * 64, 1) void cfl_conv1x1_wgrad_kernel(const u16* __restrict__ dy, const u16* __restrict__ x, long long M,
                                                                           int Ci, int Co, long long rows_per_split, float* __restrict__ part) {
    constexpr int NW = C::NW, NS = C::NS, D = C::D, MA = C::MA, MB = C::MB;
    extern __shared__ __attribute__((aligned(16))) char w1lds[];
    char* const ldsA = w1lds;
    char* const ldsB = w1lds + NS * C::AS;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wa = w % C::WA, wb = w / C::WA;
    const int ntb = Ci / C::TB, ntile = (Co / C::TA) * ntb;
    const int b = blockIdx.x, xcd = b & 7, jb = b >> 3;
    const int tile = jb % ntile, split = xcd + 8 * (jb / ntile);
    const int co0 = (tile / ntb) * C::TA, ci0 = (tile % ntb) * C::TB;
    const long long m0 = (long long)split * rows_per_split;
    long long m1 = m0 + rows_per_split;
    m1 = m1 > M ? M : m1;
    const int R = m1 > m0 ? (int)((m1 - m0 + C::KS - 1) / C::KS) : 0;           // stages of this workgroup

    // per-lane offsets of the transposing reads inside a stage (lane = 16 g + p: K row 8 (g >> 1) + (p >> 2) [+ 4 for the second
    // read], channels 16 (g & 1) + 4 (p & 3) .. + 3 of a 32-channel block); one pair per 32-channel block of the wave's tile
    const int g = lane >> 4, p = lane & 15;
    const int krow = 8 * (g >> 1) + (p >> 2);
    int offA[MA][2], offB[MB][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = krow + 4 * j;
#pragma unroll
        for (int i = 0; i < MA; ++i) {
            const int c = wa * (C::TA / C::WA) + 32 * i + 16 * (g & 1) + 4 * (p & 3);
            offA[i][j] = row * C::APITCH + (((c >> 3) ^ w1_swz<C::APITCH>(row)) << 4) + (c & 7) * 2;
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int c = wb * (C::TB / C::WB) + 32 * i + 16 * (g & 1) + 4 * (p & 3);
            offB[i][j] = row * C::BPITCH + (((c >> 3) ^ w1_swz<C::BPITCH>(row)) << 4) + (c & 7) * 2;
        }
    }

    // staging: instruction t of a stage (t < TAI: dY rows, else X rows; 1 KB = 1024 / pitch rows) belongs to wave t % NW.  A lane
    // walks one pointer per instruction it owns (+ 16 rows per stage); rows >= m1 come from a page of zeros.
    const unsigned ldsAa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsA;
    const unsigned ldsBa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsB;
    const u16* st_src[C::IPW];
    long long st_row[C::IPW];
    int my_ipw = 0;
#pragma unroll
    for (int i = 0; i < C::IPW; ++i) {
        const int t = w + i * NW;
        st_src[i] = x;
        st_row[i] = M;
        if (t < C::T) {
            ++my_ipw;
            if (t < C::TAI) {
                constexpr int PPR = C::APITCH / 16;                 // 16-byte pieces per row
                const int row = t * (64 / PPR) + lane / PPR, pc = lane % PPR;
                const int lp = pc ^ w1_swz<C::APITCH>(row);
                st_row[i] = m0 + row;
                st_src[i] = dy + (m0 + row) * Co + co0 + lp * 8;
            } else {
                constexpr int PPR = C::BPITCH / 16;
                const int row = (t - C::TAI) * (64 / PPR) + lane / PPR, pc = lane % PPR;
                const int lp = pc ^ w1_swz<C::BPITCH>(row);
                st_row[i] = m0 + row;
                st_src[i] = x + (m0 + row) * Ci + ci0 + lp * 8;
            }
        }
    }
    const long long stepA = (long long)C::KS * Co, stepB = (long long)C::KS * Ci;
    const void* zero_src = (const void*)g_w1_zero_page;
    asm volatile("" : "+v"(zero_src));
    auto dma16 = [&](const void* src, unsigned dst) {     // (inline assembly on purpose: see the header and csrc/wgrad3x3.hip)
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto issue_stage = [&](int q) {                       // called once per q, in increasing order
#pragma unroll
        for (int i = 0; i < C::IPW; ++i) {
            const int t = w + i * NW;
            if (t < C::T) {
                const bool isa = t < C::TAI;
                const unsigned dst = isa ? ldsAa + (q % NS) * C::AS + t * 1024 : ldsBa + (q % NS) * C::BS + (t - C::TAI) * 1024;
                if (q < R) dma16(st_row[i] < m1 ? (const void*)st_src[i] : zero_src, dst);
                st_src[i] += isa ? stepA : stepB;
                st_row[i] += C::KS;
            }
        }
    };

    f32x16 acc[MA][MB];
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Frags { w1_bf16x8 a[MA], b[MB]; };
    auto load_unit = [&](int q, Frags& f) {               // unconditional (behind the last stage: a slot nobody uses)
        const char* as = ldsA + (q % NS) * C::AS;
        const char* bs = ldsB + (q % NS) * C::BS;
#pragma unroll
        for (int i = 0; i < MA; ++i) f.a[i] = w1_tr(as + offA[i][0], as + offA[i][1]);
#pragma unroll
        for (int i = 0; i < MB; ++i) f.b[i] = w1_tr(bs + offB[i][0], bs + offB[i][1]);
    };
    auto compute_unit = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < MA; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
    };
    // stages <= `upto` have landed for everybody: this wave has issued stages <= `issued`, my_ipw instructions each, in order
    auto land = [&](int upto, int issued) {
        const int last = issued < R - 1 ? issued : R - 1;
        const int fly = last - upto;
        w1_wait_vm(fly > 0 ? fly * my_ipw : 0);
        asm volatile("s_barrier" ::: "memory");
    };

#pragma unroll
    for (int q = 0; q < D; ++q) issue_stage(q);
    if (R > 0) {
        land(1, D - 1);
        Frags F[2];
        load_unit(0, F[0]);
        auto stage = [&](auto par, int q) {
            constexpr int P = decltype(par)::value;
            issue_stage(q + D);                           // into the slot of stage q + D - NS = q - 2: dead
            load_unit(q + 1, F[P ^ 1]);                   // stage q + 1: landed since the last barrier
            compute_unit(F[P]);
            if (q + 1 < R) land(q + 2, q + D);
        };
        int q = 0;
        for (; q + 1 < R; q += 2) {
            stage(std::integral_constant<int, 0>(), q);
            stage(std::integral_constant<int, 1>(), q + 1);
        }
        if (q < R) stage(std::integral_constant<int, 0>(), q);
    }

    // C layout of the 32 x 32 MFMA: lane -> column (ci) lane & 31, rows (co) (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            float* out = part + ((long long)split * Co + co0 + wa * (C::TA / C::WA) + 32 * i) * Ci + ci0 + wb * (C::TB / C::WB) + 32 * j +
                         (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(long long)row * Ci] = acc[i][j][r];
            }
        }
}

// dw[e] (bf16) = sum over splits, in split order, of part[s][e]
__global__ __launch_bounds__(256) void cfl_conv1x1_wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, long long n,
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

inline int& w1_wgs_target() {
    static int v = getenv("CFL_WGRAD1_WGS") ? atoi(getenv("CFL_WGRAD1_WGS")) : 128;
    return v;
}

// tile configuration of a shape: 0 = not taken; else 1 .. 6
inline int w1_cfg(long long M, int Ci, int Co) {
    if (M < 16 || Ci <= 0 || Co <= 0) return 0;
    if (Co % 256 == 0 && Ci % 256 == 0) return 1;       // <256, 256>
    if (Co % 128 == 0 && Ci % 256 == 0) return 2;       // <128, 256>
    if (Co % 256 == 0 && Ci % 128 == 0) return 3;       // <256, 128>
    if (Co == 64 && Ci % 256 == 0) return 4;            // < 64, 256>
    if (Co % 256 == 0 && Ci == 64) return 5;            // <256,  64>
    if (Co == 64 && Ci == 64) return 6;                 // < 64,  64>
    return 0;
}

inline void w1_tile(int cfg, int& ta, int& tb) {
    static const int TA[7] = {0, 256, 128, 256, 64, 256, 64}, TB[7] = {0, 256, 256, 128, 256, 64, 64};
    ta = TA[cfg]; tb = TB[cfg];
}

inline int w1_nsplit(long long M, int Ci, int Co, int cfg) {
    int ta, tb;
    w1_tile(cfg, ta, tb);
    const int ntile = (Co / ta) * (Ci / tb);
    int ns = w1_wgs_target() / ntile;
    ns = (ns / 8) * 8;
    if (ns < 8) ns = 8;
    const long long stages = (M + 15) / 16;
    while (ns > 8 && (long long)(ns - 8) * 8 >= stages) ns -= 8;       // at least 8 stages per range
    return ns;
}

template <class C>
int w1_launch(const u16* dy, const u16* x, long long M, int Ci, int Co, int ns, float* part, hipStream_t stream) {
    const int ntile = (Co / C::TA) * (Ci / C::TB);
    long long rps = (M + ns - 1) / ns;
    rps = ((rps + C::KS - 1) / C::KS) * C::KS;
    CFL_SET_LDS((cfl_conv1x1_wgrad_kernel<C>), C::LDS);
    CFL_LAUNCH(K_CONV1_WGRAD, (cfl_conv1x1_wgrad_kernel<C>), dim3(ntile * ns), dim3(C::NW * 64), C::LDS, stream, dy, x, M, Ci, Co, rps, part);
    return 0;
}

}  // namespace

extern "C" int cfl_conv1x1_wgrad_supported(long long M, int Ci, int Co) { return w1_cfg(M, Ci, Co) ? 1 : 0; }

extern "C" int cfl_conv1x1_wgrad_workgroups(int wgs) {
    const int old = w1_wgs_target();
    if (wgs > 0) w1_wgs_target() = wgs;
    return old;
}

extern "C" size_t cfl_conv1x1_wgrad_ws_bytes(long long M, int Ci, int Co) {
    const int cfg = w1_cfg(M, Ci, Co);
    if (!cfg) return 0;
    return cfl_align256((size_t)w1_nsplit(M, Ci, Co, cfg) * Co * Ci * sizeof(float));
}

extern "C" int cfl_conv1x1_wgrad(const void* dy, const void* x, long long M, int Ci, int Co, void* dw, void* ws, void* stream_) {
    if (!dy || !x || !dw || !ws) return CFL_EINVAL;
    const int cfg = w1_cfg(M, Ci, Co);
    if (!cfg || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw | (uintptr_t)ws) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const int ns = w1_nsplit(M, Ci, Co, cfg);
    float* part = (float*)ws;
    const u16 *d = (const u16*)dy, *xx = (const u16*)x;
    int rc;
    switch (cfg) {
        case 1: rc = w1_launch<W1Cfg<256, 256, 4, 2>>(d, xx, M, Ci, Co, ns, part, stream); break;
        case 2: rc = w1_launch<W1Cfg<128, 256, 2, 4>>(d, xx, M, Ci, Co, ns, part, stream); break;
        case 3: rc = w1_launch<W1Cfg<256, 128, 4, 2>>(d, xx, M, Ci, Co, ns, part, stream); break;
        case 4: rc = w1_launch<W1Cfg<64, 256, 2, 4>>(d, xx, M, Ci, Co, ns, part, stream); break;
        case 5: rc = w1_launch<W1Cfg<256, 64, 4, 2>>(d, xx, M, Ci, Co, ns, part, stream); break;
        default: rc = w1_launch<W1Cfg<64, 64, 2, 2>>(d, xx, M, Ci, Co, ns, part, stream); break;
    }
    if (rc) return rc;
    const long long n = (long long)Co * Ci;
    CFL_LAUNCH(K_CONV1_WGRAD_REDUCE, cfl_conv1x1_wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream,
               (const float*)part, ns, n, (u16*)dw);
    return 0;
}
