// conv3x3_x3.hip -- 3 x 3 / stride 1 / padding 1 convolution of fp32 channels_last tensors at fp32-class accuracy on the bf16 matrix
// pipe (the "3 x bf16 split" of tile_x3.h): forward, and -- the same kernel on the rotated, transposed weight -- data gradient.
//
//   y[n, h, w, co] = sum over (kh, kw, ci) of x[n, h + kh - 1, w + kw - 1, ci] * W[co, kh, kw, ci]                (zero padding)
//
// Where it sits: the BasicBlock convolutions of the CLIENTS' ResNet-18 (src/networks/resnet_client.py:33-66,102-201), which the
// reference runs in fp32 (ClientTrainer.py has no mixed precision) on cuDNN.  On MI355X the library's fp32 implicit-GEMM / Winograd
// kernels are 68 % of the three client kinds' kernel time and ~85 % of an image client's contrast step (profiles/
// r5_client_step_kernel_stats.csv) at 45-100 TFLOP/s: the fp32 matrix rate of the chip is 157 TFLOP/s, the bf16 rate 2 500.  Every
// fp32 operand element is split x = hi + lo (two bf16) while it is staged and each product runs as three bf16 MFMAs (hi.hi + lo.hi +
// hi.lo; the dropped term is 2^-16 relative): 16 mantissa bits per operand, outputs within 4.6e-6 of scale of fp64 at K = 9 x 128 ..
// 9 x 512 (the library's fp32 kernels: 3e-7 .. 8e-7; TF32, which cuDNN uses for fp32 convolutions by default on the A100-class GPUs
// the reference ran on, carries 10 bits), at a roof of 833 TFLOP/s of fp32-equivalent work.
//
// Implicit GEMM, M = N H W output positions, N = Co, K = 9 Ci walked as (tap, 32-channel chunk): a 128 x 128 (or 256 x 128) output
// tile per workgroup, 4 waves as 2 x 2, the machinery of tile_x3.h (128-byte LDS rows [32 hi | 32 lo], XOR-swizzled, register
// staging with the split on the VALU, double buffer, one barrier per K step).  The A operand of a K step is the input shifted by
// the tap: row r of the tile reads position r + (kh - 1) W + (kw - 1) of the same image, or zeros where that leaves the image --
// the nine validity bits of a thread's rows are computed once, the shift is one pointer offset per tap.  B is the weight as it lies
// in memory ([Co][3][3][Ci] = a K-contiguous [Co, 9 Ci] matrix).
// The data gradient dX = conv(dY, W') with W'[ci][kh][kw][co] = W[co][2 - kh][2 - kw][ci] is the same call on a weight prepared by
// cfl_conv3x3_x3_rot_weight.  The weight gradient stays on the library (fp32): an fp32-accurate TN form is not built.
#include "common.h"
#include "tile_x3.h"

namespace {

template <int TM>
struct ARegs { f32x4 r[2 * TM]; };

// rows of this thread in a 64 TM-row stage: rl = p * 32 + (t >> 3), p < 2 TM; k quad kq = t & 7
template <int TM, int TN>
__global__ __launch_bounds__(256) void cfl_conv3x3_x3_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                              int N, int H, int W, int Ci, int Co) {
    using C = x3::Cfg<TM, TN>;
    constexpr int BM = C::BM, BN = C::BN, NP = 2 * TM;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char* lds = reinterpret_cast<char*>(lds_f);
    const long long M = (long long)N * H * W;
    const int ntc = Co / BN, ntr = (int)((M + BM - 1) / BM);
    int ti, tj;
    tile_swizzle(xcd_remap(blockIdx.x, gridDim.x), ntr, ntc, ti, tj);
    const long long row0 = (long long)ti * BM;
    const int col0 = tj * BN;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    const int kq = t & 7;

    // this thread's A rows: source pointer of the centre tap and the nine validity bits
    const float* abase[NP];
    unsigned avalid[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const long long r = row0 + p * 32 + (t >> 3);
        unsigned v = 0;
        long long rc = r < M ? r : M - 1;
        if (r < M) {
            const int hw = (int)(r % ((long long)H * W));
            const int h = hw / W, ww = hw % W;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if ((unsigned)(h + kh - 1) < (unsigned)H && (unsigned)(ww + kw - 1) < (unsigned)W) v |= 1u << (kh * 3 + kw);
        }
        avalid[p] = v;
        abase[p] = x + rc * Ci + 4 * kq;
    }
    const Opnd Bo{w, (long long)9 * Ci, Co, 9 * Ci, 1};
    const int nchunk = Ci / 32, nk = 9 * nchunk;

    ARegs<TM> ra;
    x3::StageRegs<true, BN> rb;
    auto load_a = [&](int s) {
        const int tap = s / nchunk, c0 = (s - tap * nchunk) * 32;
        const long long off = ((long long)(tap / 3 - 1) * W + (tap % 3 - 1)) * Ci + c0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bool ok = (avalid[p] >> tap) & 1u;
            // (clamped address for the rows that read nothing: the centre tap of the row itself is always inside the tensor)
            const f32x4 v = *reinterpret_cast<const f32x4*>(abase[p] + (ok ? off : (long long)c0));
            ra.r[p] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_a = [&](char* st) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int rl = p * 32 + (t >> 3);
            x3::bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 hh, ll;
                x3::split1(ra.r[p][e], hh, ll);
                hi[e] = hh; lo[e] = ll;
            }
            *reinterpret_cast<x3::bf16x4*>(st + x3::soff(rl, kq >> 1) + (kq & 1) * 8) = hi;
            *reinterpret_cast<x3::bf16x4*>(st + x3::soff(rl, 4 + (kq >> 1)) + (kq & 1) * 8) = lo;
        }
    };
    auto bk = [&](int s) { const int tap = s / nchunk; return tap * Ci + (s - tap * nchunk) * 32; };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    load_a(0);
    x3::stage_kc<BN, 0>(Bo, col0, bk(0), rb.r, lds, XfIdentity());
    store_a(lds);
    x3::stage_kc<BN, 1>(Bo, col0, bk(0), rb.r, lds + BM * 128, XfIdentity());
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
        const char* sa = lds + (s & 1) * C::STAGE_BYTES;
        char* da = lds + ((s + 1) & 1) * C::STAGE_BYTES;
        const bool more = s + 1 < nk;
        if (more) {
            load_a(s + 1);
            x3::stage_kc<BN, 0>(Bo, col0, bk(s + 1), rb.r, da, XfIdentity());
        }
        x3::compute<TM, TN>(sa, sa + BM * 128, acc, lane, wr, wc);
        if (more) {
            store_a(da);
            x3::stage_kc<BN, 1>(Bo, col0, bk(s + 1), rb.r, da + BM * 128, XfIdentity());
        }
        __syncthreads();
    }
    // C / D layout: a lane holds one column (co) of 16 rows per 32 x 32 tile: 32 lanes = 128 contiguous bytes of one output row
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int j = col0 + acc_col<TN>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long i = row0 + acc_row<TM>(wr, m, r, lane);
                if (i < M) y[i * Co + j] = acc[m][n][r];
            }
        }
}

// ---- version 2: the input slab of a 32-channel chunk is staged ONCE for the nine taps ---------------------------------------------
// The kernel above re-reads (and re-splits) its A tile for every tap: 9 x 16 KB + 9 x 16 KB of weights per chunk and 128 x 128 tile, ~920
// MB through L2 at the layer2 shape -- it runs at the L2 -> CU rate (172 us), not at the matrix pipe's (floor 35 us).  Here K runs
// (chunk, tap): per chunk the positions row0 - (W + 1) .. row0 + BM + W of 32 channels go into ONE LDS slab ([32 hi | 32 lo] rows,
// the swizzle of tile_x3.h), and a tap is the same fragment read moved by (kh - 1) W + (kw - 1) rows; fragments whose shifted position
// leaves the image are zeroed in registers (nine validity bits per lane and 32-row block).  Only the weights stream per tap.  A's
// traffic falls 4.5 x, the split of A runs once instead of nine times.  W <= 63 (slab = BM + 128 rows).
template <int TM, int TN>
__global__ __launch_bounds__(256) void cfl_conv3x3_x3s_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                               int N, int H, int W, int Ci, int Co) {
    constexpr int BM = 64 * TM, BN = 64 * TN, SR = BM + 128, NJ = SR / 32;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char* slab = reinterpret_cast<char*>(lds_f);
    const long long M = (long long)N * H * W;
    const int ntc = Co / BN, ntr = (int)((M + BM - 1) / BM);
    int ti, tj;
    tile_swizzle(xcd_remap(blockIdx.x, gridDim.x), ntr, ntc, ti, tj);
    const long long row0 = (long long)ti * BM;
    const int col0 = tj * BN;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    const int kq = t & 7, i32 = lane & 31, hh = lane >> 5;
    const int halo = W + 1;
    const int nj = (BM + 2 * halo + 31) / 32;                 // 32-row groups of the slab actually used (<= NJ)
    char* const bst = slab + nj * (32 * 128);                 // two B stages of BN rows behind it

    unsigned avalid[TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const long long r = row0 + (wr * TM + m) * 32 + i32;
        unsigned v = 0;
        if (r < M) {
            const int hw = (int)(r % ((long long)H * W));
            const int h = hw / W, ww = hw % W;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if ((unsigned)(h + kh - 1) < (unsigned)H && (unsigned)(ww + kw - 1) < (unsigned)W) v |= 1u << (kh * 3 + kw);
        }
        avalid[m] = v;
    }
    const Opnd Bo{w, (long long)9 * Ci, Co, 9 * Ci, 1};
    const int nchunk = Ci / 32, nk = 9 * nchunk;

    f32x4 sreg[NJ];
    x3::StageRegs<true, BN> rb;
    auto load_slab = [&](int c) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j < nj) {
                const long long P = row0 - halo + j * 32 + (t >> 3);
                const bool ok = P >= 0 && P < M;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (ok ? P : 0) * Ci + c * 32 + 4 * kq);
                sreg[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j < nj) {
                const int rl = j * 32 + (t >> 3);
                x3::bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 a, b;
                    x3::split1(sreg[j][e], a, b);
                    hi[e] = a; lo[e] = b;
                }
                *reinterpret_cast<x3::bf16x4*>(slab + x3::soff(rl, kq >> 1) + (kq & 1) * 8) = hi;
                *reinterpret_cast<x3::bf16x4*>(slab + x3::soff(rl, 4 + (kq >> 1)) + (kq & 1) * 8) = lo;
            }
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    auto compute_tap = [&](int tap, const char* sb) {
        const int shift = halo + (tap / 3 - 1) * W + (tap % 3 - 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            x3::bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                const int r = (wr * TM + m) * 32 + i32 + shift;
                f32x4 vh = *reinterpret_cast<const f32x4*>(slab + x3::soff(r, 2 * kk + hh));
                f32x4 vl = *reinterpret_cast<const f32x4*>(slab + x3::soff(r, 4 + 2 * kk + hh));
                if (!((avalid[m] >> tap) & 1u)) { vh = f32x4{0.f, 0.f, 0.f, 0.f}; vl = vh; }
                ah[m] = __builtin_bit_cast(x3::bf16x8, vh);
                al[m] = __builtin_bit_cast(x3::bf16x8, vl);
            }
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int r = (wc * TN + n) * 32 + i32;
                bh[n] = *reinterpret_cast<const x3::bf16x8*>(sb + x3::soff(r, 2 * kk + hh));
                bl[n] = *reinterpret_cast<const x3::bf16x8*>(sb + x3::soff(r, 4 + 2 * kk + hh));
            }
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
        }
    };
    auto bk = [&](int s) { const int c = s / 9; return (s - 9 * c) * Ci + c * 32; };      // step s = chunk c, tap s - 9 c

    load_slab(0);
    x3::stage_kc<BN, 0>(Bo, col0, bk(0), rb.r, bst, XfIdentity());
    store_slab();
    x3::stage_kc<BN, 1>(Bo, col0, bk(0), rb.r, bst, XfIdentity());
    __syncthreads();
    int s = 0;
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++s) {
            const bool more = s + 1 < nk;
            const bool next_slab = tap == 8 && c + 1 < nchunk;
            if (more) x3::stage_kc<BN, 0>(Bo, col0, bk(s + 1), rb.r, bst, XfIdentity());
            if (tap == 7 && c + 1 < nchunk) load_slab(c + 1);
            compute_tap(tap, bst + (s & 1) * (BN * 128));
            if (more) x3::stage_kc<BN, 1>(Bo, col0, bk(s + 1), rb.r, bst + ((s + 1) & 1) * (BN * 128), XfIdentity());
            if (next_slab) {
                __syncthreads();                              // every wave is done with this chunk's slab
                store_slab();
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int j = col0 + acc_col<TN>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long i = row0 + acc_row<TM>(wr, m, r, lane);
                if (i < M) y[i * Co + j] = acc[m][n][r];
            }
        }
}

// ---- version 3: nothing but reads and MFMAs inside a tap -------------------------------------------------------------------------
// Version 2 runs at ~0.3 of the matrix pipe although its bytes are few: a tap's 24 MFMAs (768 pipe cycles) sit among ~350 other
// instructions of the same wave -- 77 v_cndmask (zeroing the fragments whose shifted position leaves the image, and the weight
// staging's bounds), 50 VALU for splitting the weight tile again in every workgroup and tap, shift / swizzle address arithmetic, 27
// waits.  Here:
//   * the WEIGHT is split ONCE per call by cfl_conv3x3_wimage_kernel into an image that is byte for byte the LDS stage
//     ([tap][chunk][co] rows of [32 hi | 32 lo], 16-byte pieces XOR-swizzled): staging a tap's tile is 16-byte loads and 16-byte
//     LDS stores, no VALU;
//   * the slab rows are 144 bytes (128 + 16 padding: conflict-free ds_read_b128 without a swizzle), so a fragment address is
//     row * 144 + constant, and a lane's NINE row addresses (one per tap) are computed once per workgroup -- a tap whose shifted
//     position leaves the image points at a row of zeros instead (no select in the loop);
//   * the nine taps of a chunk are unrolled: per tap 16 ds_read_b128, 2-4 global loads + LDS stores for the next tap's weights,
//     one barrier.
// DBG (measurements only, results are wrong): 1 no weight staging in the loop, 2 no per-tap barrier, 3 no MFMAs, 4 no fragment reads,
// 6 no slab staging in the loop
// S = 2 (round 6): the same kernel for stride 2 (padding 1, even H and W: the three down-sampling convolutions of ResNet-18).  Output
// position p = (n Ho + ho) Wo + wo reads around input position c(p) = 2 (p div Wo) W + 2 (p mod Wo), which grows monotonically with p
// (by 2 inside a row, by W + 2 at a row or image wrap), so a tile of BM output positions still reads ONE contiguous range of input
// positions, c(row0) - (W + 1) ... c(row0 + BM - 1) + (W + 1), at most 4 BM + 3 W + 3 rows of the slab (W <= 63); a lane's nine row addresses
// (the only place the mapping lives) are computed once, the loop is unchanged.  H, W here are the INPUT extents.
template <int TM, int TN, int DBG = 0, int S = 1>
__global__ __launch_bounds__(256) void cfl_conv3x3_x3p_kernel(const float* __restrict__ x, const char* __restrict__ wimg, float* __restrict__ y,
                                                               int N, int H, int W, int Ci, int Co) {
    constexpr int BM = 64 * TM, BN = 64 * TN, NJ = (S == 2 ? 4 * BM + 256 : BM + 128) / 32, NB = BN / 32, PITCH = 144;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char* slab = reinterpret_cast<char*>(lds_f);
    const int Ho = S == 2 ? H / 2 : H, Wo = S == 2 ? W / 2 : W;
    const long long M = (long long)N * Ho * Wo;                  // output positions
    const long long Min = (long long)N * H * W;                  // input positions
    const int ntc = Co / BN, ntr = (int)((M + BM - 1) / BM);
    int ti, tj;
    tile_swizzle(xcd_remap(blockIdx.x, gridDim.x), ntr, ntc, ti, tj);
    const long long row0 = (long long)ti * BM;
    const int col0 = tj * BN;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    const int kq = t & 7, i32 = lane & 31, hh = lane >> 5;
    const int halo = W + 1;
    // centre input position of an output position (S = 1: itself)
    auto centre = [&](long long p) -> long long { return S == 2 ? 2 * (p / Wo) * W + 2 * (p % Wo) : p; };
    const long long last = (row0 + BM - 1 < M ? row0 + BM - 1 : M - 1);
    const long long cbase = centre(row0) - halo;              // input position of slab row 0
    const int nj = S == 2 ? (int)((centre(last) + halo - cbase + 1 + 31) / 32) : (BM + 2 * halo + 31) / 32;   // 32-row groups used (<= NJ)
    const int zrow = nj * 32;                                 // a row of zeros behind the slab
    char* const bst = slab + (nj * 32 + 1) * PITCH;           // two weight stages of BN rows behind it (16-byte aligned: 144 = 9 x 16)

    // a lane's fragment row per tap (byte offset into the slab, lane half included)
    int aoff[TM][9];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int rl = (wr * TM + m) * 32 + i32;
        const long long r = row0 + rl;
        int h = -4, ww = -4;                                  // (rows behind the tensor: every tap invalid)
        int srow = rl + halo;                                 // slab row of the centre tap
        if (r < M) {
            const int hw = (int)(r % ((long long)Ho * Wo));
            h = S * (hw / Wo); ww = S * (hw % Wo);                // the centre's input coordinates
            if (S == 2) srow = (int)(centre(r) - cbase);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            const bool ok = (unsigned)(h + kh - 1) < (unsigned)H && (unsigned)(ww + kw - 1) < (unsigned)W;
            aoff[m][tap] = (ok ? srow + (kh - 1) * W + (kw - 1) : zrow) * PITCH + hh * 16;
        }
    }
    int boff[TN][4];                                          // weight fragments: hi kk = 0, 1, lo kk = 0, 1
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int rb = (wc * TN + n) * 32 + i32;
#pragma unroll
        for (int j = 0; j < 4; ++j) boff[n][j] = x3::soff(rb, (j >> 1) * 4 + 2 * (j & 1) + hh);
    }
    const int nchunk = Ci / 32;
    const long long wstep = (long long)Co * 128;              // bytes between the tiles of consecutive (tap, chunk) steps
    const char* wsrc = wimg + (long long)col0 * 128 + t * 16;

    f32x4 sreg[NJ], breg[NB];
    auto load_slab = [&](int c) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j < nj) {
                const long long P = cbase + j * 32 + (t >> 3);
                const bool ok = P >= 0 && P < Min;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (ok ? P : 0) * Ci + c * 32 + 4 * kq);
                sreg[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j < nj) {
                const int rl = j * 32 + (t >> 3);
                x3::bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 a, b;
                    x3::split1(sreg[j][e], a, b);
                    hi[e] = a; lo[e] = b;
                }
                *reinterpret_cast<x3::bf16x4*>(slab + rl * PITCH + kq * 8) = hi;
                *reinterpret_cast<x3::bf16x4*>(slab + rl * PITCH + 64 + kq * 8) = lo;
            }
        }
    };
    auto load_b = [&](int c, int tap) {                       // step (c, tap): tile of BN x 128 bytes, contiguous in the image
        const char* src = wsrc + ((long long)tap * nchunk + c) * wstep;
#pragma unroll
        for (int i = 0; i < NB; ++i) breg[i] = *reinterpret_cast<const f32x4*>(src + i * 4096);
    };
    auto store_b = [&](char* dst) {
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(dst + t * 16 + i * 4096) = breg[i];
    };
    // DBG = 5 (a real form: results are right): the weight tile of the next step goes to LDS by DMA (global_load_lds_dwordx4: the image
    // is the stage byte for byte, wave w moves kilobytes w, w + 4, ...) -- no staging registers, no ds_write; issued behind the step's
    // fragment reads (the compiler orders every LDS read behind every LDS-DMA it knows of), landed at the step's closing barrier
    auto dma_b = [&](int c, int tap, char* dst) {
        const char* src = wimg + (long long)col0 * 128 + ((long long)tap * nchunk + c) * wstep + lane * 16;
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((glb_vptr)(src + (wid + 4 * i) * 1024), (lds_vptr)(dst + (wid + 4 * i) * 1024), 16, 0, 0);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    if (t < 9) *reinterpret_cast<f32x4*>(slab + zrow * PITCH + t * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    load_slab(0);
    load_b(0, 0);
    store_slab();
    store_b(bst);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const bool next_chunk = c + 1 < nchunk;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int par = (c + tap) & 1;                    // 9 taps per chunk: the stage parity alternates across chunks too
            const char* sb = bst + par * (BN * 128);
            if (DBG != 1 && DBG != 5 && DBG != 7) {
                if (tap < 8) load_b(c, tap + 1);
                else if (next_chunk) load_b(c + 1, 0);
            }
            if (DBG != 6 && tap == 1 && next_chunk) load_slab(c + 1);      // (first touch of these bytes: an HBM round trip, ~7 taps away)
            __builtin_amdgcn_sched_barrier(0);                // (the loads stay in front of the tap's MFMAs: their latency hides there)
            if (DBG == 5 || DBG == 7) {
                x3::bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int m = 0; m < TM; ++m) {
                        ah[kk][m] = *reinterpret_cast<const x3::bf16x8*>(slab + aoff[m][tap] + kk * 32);
                        al[kk][m] = *reinterpret_cast<const x3::bf16x8*>(slab + aoff[m][tap] + 64 + kk * 32);
                    }
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        bh[kk][n] = *reinterpret_cast<const x3::bf16x8*>(sb + boff[n][kk]);
                        bl[kk][n] = *reinterpret_cast<const x3::bf16x8*>(sb + boff[n][2 + kk]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (DBG == 5) {                                   // (DBG = 5: the order of the round's earlier sessions, kept for the A/B)
                    if (tap < 8) dma_b(c, tap + 1, bst + (par ^ 1) * (BN * 128));
                    else if (next_chunk) dma_b(c + 1, 0, bst + (par ^ 1) * (BN * 128));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if (kk == 1 && DBG == 7) {
                        // the next tap's weight tile is requested BEHIND the first half's MFMAs (round 6): with the LDS-DMA between the
                        // reads and the first MFMA the compiler waited for lgkmcnt(0) -- all sixteen fragment reads -- where the first
                        // twelve MFMAs need eight of them (an LDS-DMA in flight makes its wait-count bookkeeping give up partial waits)
                        __builtin_amdgcn_sched_barrier(0);
                        if (tap < 8) dma_b(c, tap + 1, bst + (par ^ 1) * (BN * 128));
                        else if (next_chunk) dma_b(c + 1, 0, bst + (par ^ 1) * (BN * 128));
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk][m], bh[kk][n], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kk][m], bh[kk][n], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk][m], bl[kk][n], acc[m][n], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                x3::bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
                if (DBG == 4) {
#pragma unroll
                    for (int m = 0; m < TM; ++m) { ah[m] = __builtin_bit_cast(x3::bf16x8, sreg[0]); al[m] = ah[m]; }
#pragma unroll
                    for (int n = 0; n < TN; ++n) { bh[n] = __builtin_bit_cast(x3::bf16x8, breg[0]); bl[n] = bh[n]; }
                } else {
#pragma unroll
                    for (int m = 0; m < TM; ++m) {
                        ah[m] = *reinterpret_cast<const x3::bf16x8*>(slab + aoff[m][tap] + kk * 32);
                        al[m] = *reinterpret_cast<const x3::bf16x8*>(slab + aoff[m][tap] + 64 + kk * 32);
                    }
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        bh[n] = *reinterpret_cast<const x3::bf16x8*>(sb + boff[n][kk]);
                        bl[n] = *reinterpret_cast<const x3::bf16x8*>(sb + boff[n][2 + kk]);
                    }
                }
                if (DBG == 3) {
#pragma unroll
                    for (int m = 0; m < TM; ++m) asm volatile("" :: "v"(ah[m]), "v"(al[m]));
#pragma unroll
                    for (int n = 0; n < TN; ++n) asm volatile("" :: "v"(bh[n]), "v"(bl[n]));
                } else {
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (DBG != 1 && DBG != 5 && DBG != 7 && (tap < 8 || next_chunk)) store_b(bst + (par ^ 1) * (BN * 128));
            if (tap == 8 && next_chunk) {
                __syncthreads();                              // every wave is done with this chunk's slab
                if (DBG != 6) store_slab();
            }
            if (DBG != 2 || tap == 8) __syncthreads();
        }
    }
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int j = col0 + acc_col<TN>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long i = row0 + acc_row<TM>(wr, m, r, lane);
                if (i < M) y[i * Co + j] = acc[m][n][r];
            }
        }
}

// ---- version 4 ("q"): the two memory streams on separate waves, weight tiles two taps ahead -------------------------------------------
// What version 3's switch-off forms say (profiles/r6_x3conv_decomposition.jsonl): its memory work and its MFMAs overlap poorly.  A wave's
// vector-memory operations complete IN ORDER: a wave that has the next chunk's slab loads (first touch of those bytes: an HBM round trip)
// in its queue cannot see its next weight tile (an L2 hit) land before them, and a weight tile issued one tap ahead has one tap (~770 matrix
// cycles) for an L2 round trip under a 4-5 TB/s L2 -> LDS load.  Here the streams belong to different waves and are deeper:
//   * waves 0, 1 issue ALL weight-tile DMA (8 / 4 kilobyte pieces each per tap), TWO taps ahead, into a ring of three stages (nine taps per
//     chunk: the stage of a tap is tap mod 3, static); their only wait is a counted vmcnt that leaves the youngest tile in flight.  The DMA
//     is issued from inline assembly (the compiler would put s_waitcnt vmcnt(0) in front of every LDS read that follows a DMA it knows of);
//   * waves 2, 3 load the next chunk's slab (registers) at the chunk's first tap and split / store it behind its last one: eight taps for
//     the HBM round trip, and nobody waits for a weight tile behind them.
template <int TM, int TN>
__global__ __launch_bounds__(256) void cfl_conv3x3_x3q_kernel(const float* __restrict__ x, const char* __restrict__ wimg, float* __restrict__ y,
                                                               int N, int H, int W, int Ci, int Co) {
    constexpr int BM = 64 * TM, BN = 64 * TN, NJ2 = (BM + 128) / 16, NKB = BN / 16, PITCH = 144;   // NKB: kilobyte pieces of a tile per loader wave
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char* slab = reinterpret_cast<char*>(lds_f);
    const long long M = (long long)N * H * W;
    const int ntc = Co / BN, ntr = (int)((M + BM - 1) / BM);
    int ti, tj;
    tile_swizzle(xcd_remap(blockIdx.x, gridDim.x), ntr, ntc, ti, tj);
    const long long row0 = (long long)ti * BM;
    const int col0 = tj * BN;
    const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6), wr = wid >> 1, wc = wid & 1;
    const int kq = t & 7, i32 = lane & 31, hh = lane >> 5;
    const int halo = W + 1;
    const int nj2 = (BM + 2 * halo + 15) / 16;                // 16-row groups of the slab actually used (<= NJ2)
    const int zrow = nj2 * 16;                                // a row of zeros behind the slab
    char* const bst = slab + (zrow + 1) * PITCH;              // three weight stages of BN rows behind it
    const bool dma_wave = wid < 2;                            // wave-uniform roles

    int aoff[TM][9];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int rl = (wr * TM + m) * 32 + i32;
        const long long r = row0 + rl;
        int h = -4, ww = -4;
        if (r < M) {
            const int hw = (int)(r % ((long long)H * W));
            h = hw / W; ww = hw % W;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            const bool ok = (unsigned)(h + kh - 1) < (unsigned)H && (unsigned)(ww + kw - 1) < (unsigned)W;
            aoff[m][tap] = (ok ? rl + halo + (kh - 1) * W + (kw - 1) : zrow) * PITCH + hh * 16;
        }
    }
    int boff[TN][4];
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int rb = (wc * TN + n) * 32 + i32;
#pragma unroll
        for (int j = 0; j < 4; ++j) boff[n][j] = x3::soff(rb, (j >> 1) * 4 + 2 * (j & 1) + hh);
    }
    const int nchunk = Ci / 32, nk = 9 * nchunk;
    const long long wstep = (long long)Co * 128;
    const char* wsrc = wimg + (long long)col0 * 128 + lane * 16;
    const unsigned bst_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)bst;

    // slab staging by the 128 threads of waves 2, 3: rows j * 16 + (t2 >> 3), channels 4 kq .. + 3
    const int t2 = t & 127;
    f32x4 sreg[NJ2];
    auto load_slab = [&](int c) {
#pragma unroll
        for (int j = 0; j < NJ2; ++j) {
            if (j < nj2) {
                const long long P = row0 - halo + j * 16 + (t2 >> 3);
                const bool ok = P >= 0 && P < M;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (ok ? P : 0) * Ci + c * 32 + 4 * kq);
                sreg[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int j = 0; j < NJ2; ++j) {
            if (j < nj2) {
                const int rl = j * 16 + (t2 >> 3);
                x3::bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 a, b;
                    x3::split1(sreg[j][e], a, b);
                    hi[e] = a; lo[e] = b;
                }
                *reinterpret_cast<x3::bf16x4*>(slab + rl * PITCH + kq * 8) = hi;
                *reinterpret_cast<x3::bf16x4*>(slab + rl * PITCH + 64 + kq * 8) = lo;
            }
        }
    };
    // weight tile of step s (chunk s / 9, tap s % 9) into stage s % 3: loader wave w moves kilobytes w, w + 2, ...
    auto dma_b = [&](int s) {
        const int c = s / 9, tap = s - 9 * c;
        const char* src = wsrc + ((long long)tap * nchunk + c) * wstep + wid * 1024;
        const unsigned dst = bst_a + (unsigned)((s % 3) * (BN * 128) + wid * 1024);
#pragma unroll
        for (int i = 0; i < NKB; ++i) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src + i * 2048), "s"(dst + i * 2048) : "memory");
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    if (t < 9) *reinterpret_cast<f32x4*>(slab + zrow * PITCH + t * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (dma_wave) {
        dma_b(0);
        if (nk > 1) dma_b(1);
    } else {
        load_slab(0);
        store_slab();
    }
    if (dma_wave) {
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NKB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int s = 0;
    for (int c = 0; c < nchunk; ++c) {
        const bool next_chunk = c + 1 < nchunk;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++s) {
            const char* sb = bst + (tap % 3) * (BN * 128);
            const bool more = s + 2 < nk;                     // (uniform)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                x3::bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int m = 0; m < TM; ++m) {
                    ah[m] = *reinterpret_cast<const x3::bf16x8*>(slab + aoff[m][tap] + kk * 32);
                    al[m] = *reinterpret_cast<const x3::bf16x8*>(slab + aoff[m][tap] + 64 + kk * 32);
                }
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    bh[n] = *reinterpret_cast<const x3::bf16x8*>(sb + boff[n][kk]);
                    bl[n] = *reinterpret_cast<const x3::bf16x8*>(sb + boff[n][2 + kk]);
                }
                if (kk == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (dma_wave) {
                        if (more) dma_b(s + 2);               // stage (tap + 2) % 3 = (tap - 1) % 3: last read during the tap before
                    } else if (tap == 0 && next_chunk) {
                        load_slab(c + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tap == 8 && next_chunk) {
                __syncthreads();                              // every wave is done with this chunk's slab
                if (!dma_wave) store_slab();
            }
            if (dma_wave) {                                   // the tile of step s + 1 has landed; the one of step s + 2 may still fly
                if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NKB) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int j = col0 + acc_col<TN>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long i = row0 + acc_row<TM>(wr, m, r, lane);
                if (i < M) y[i * Co + j] = acc[m][n][r];
            }
        }
}

// The weight image of version 3: img[((tap * (Ci / 32) + c) * Co + co) * 128 bytes] = the 128-byte LDS row of output channel co for
// the 32 input channels of chunk c at that tap: [32 hi | 32 lo] bf16 with the 16-byte pieces XOR-swizzled as x3::soff does for row
// co (tiles start at multiples of 64, so the tile-local row and co agree in the bits the swizzle reads).  One thread per 4 channels.
__global__ __launch_bounds__(256) void cfl_conv3x3_wimage_kernel(const float* __restrict__ w, int Ci, int Co, char* __restrict__ img) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // over [Co][9][Ci / 4]
    const int q4 = Ci / 4;
    if (i >= (long long)Co * 9 * q4) return;
    const int co = (int)(i / (9 * q4)), rem = (int)(i % (9 * q4)), tap = rem / q4, ci = (rem % q4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(w + ((long long)co * 9 + tap) * Ci + ci);
    x3::bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 a, b;
        x3::split1(v[e], a, b);
        hi[e] = a; lo[e] = b;
    }
    const int c = ci >> 5, k = ci & 31;                       // chunk, channel inside it; piece k >> 3 (hi) / 4 + (k >> 3) (lo)
    char* row = img + (((long long)tap * (Ci >> 5) + c) * Co + co) * 128;
    const int sw = (co >> 1) & 7;
    *reinterpret_cast<x3::bf16x4*>(row + (((k >> 3) ^ sw) << 4) + (k & 7) * 2) = hi;
    *reinterpret_cast<x3::bf16x4*>(row + (((4 + (k >> 3)) ^ sw) << 4) + (k & 7) * 2) = lo;
}

// The image of the ROTATED, TRANSPOSED weight (the data gradient's operand) straight from the weight: rows = input channels ci, K =
// (tap', co): img[((tap' * (Co / 32) + c) * Ci + ci) * 128 bytes] = [32 hi | 32 lo] of w[32 c .. 32 c + 31][8 - tap'][ci], swizzled by ci.
// A block transposes one (32 co x 32 ci) tile of one tap through LDS: coalesced reads along ci, 128-byte rows written whole.
__global__ __launch_bounds__(256) void cfl_conv3x3_wimage_rot_kernel(const float* __restrict__ w, int Ci, int Co, char* __restrict__ img) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;                              // source tap; the image's tap is 8 - tap
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) tile[r][tx] = w[((long long)(co0 + r) * 9 + tap) * Ci + ci0 + tx];
    __syncthreads();
    const int cil = threadIdx.x >> 3, q = threadIdx.x & 7;   // row (input channel) of the tile, quad of output channels
    x3::bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 a, b;
        x3::split1(tile[4 * q + e][cil], a, b);
        hi[e] = a; lo[e] = b;
    }
    const int ci = ci0 + cil;
    char* row = img + (((long long)(8 - tap) * (Co >> 5) + (co0 >> 5)) * Ci + ci) * 128;
    const int sw = (ci >> 1) & 7;
    *reinterpret_cast<x3::bf16x4*>(row + ((((q >> 1)) ^ sw) << 4) + (q & 1) * 8) = hi;
    *reinterpret_cast<x3::bf16x4*>(row + (((4 + (q >> 1)) ^ sw) << 4) + (q & 1) * 8) = lo;
}

// wr[ci][kh][kw][co] = w[co][2 - kh][2 - kw][ci]   (both [out][3][3][in] in memory: the channels_last weight layout)
__global__ __launch_bounds__(256) void cfl_conv3x3_rot_kernel(const float* __restrict__ w, int Ci, int Co, float* __restrict__ wr) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;                              // source tap
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Co && ci < Ci) ? w[((long long)co * 9 + tap) * Ci + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Ci && co < Co) wr[((long long)ci * 9 + (8 - tap)) * Co + co] = tile[tx][r];
    }
}

inline bool x3conv_ok(int N, int H, int W, int Ci, int Co) {
    return N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && Ci % 32 == 0 && Co % 64 == 0 && (long long)N * H * W < (1ll << 31);
}

}  // namespace

extern "C" int cfl_conv3x3_x3_supported(int N, int H, int W, int Ci, int Co) { return x3conv_ok(N, H, W, Ci, Co) ? 1 : 0; }

extern "C" int cfl_conv3x3_x3_fwd(const float* x, const float* w, int N, int H, int W, int Ci, int Co, float* y, int variant,
                                  void* stream_) {
    if (!x || !w || !y) return CFL_EINVAL;
    if (!x3conv_ok(N, H, W, Ci, Co) || (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const long long M = (long long)N * H * W;
    // tile: 128 output channels where they divide, 256 positions when that still fills the chip twice
    const bool bn128 = Co % 128 == 0;
    if (variant == 0) {
        if (W <= 63) variant = (M / 256) * (Co / 64) >= 256 ? 141 : 121;        // slab kernels (measured: profiles/r6_x3conv_probe.jsonl)
        else variant = bn128 ? (M / 256 * (Co / 128) >= 512 ? 42 : 22) : (M / 256 * (Co / 64) >= 512 ? 41 : 21);
    }
#define CFL_X3CONV(TM_, TN_)                                                                                                   \
    do {                                                                                                                       \
        using C = x3::Cfg<TM_, TN_>;                                                                                           \
        const int grid = (int)((M + C::BM - 1) / C::BM) * (Co / C::BN);                                                        \
        CFL_SET_LDS((cfl_conv3x3_x3_kernel<TM_, TN_>), C::LDS_BYTES);                                                          \
        CFL_LAUNCH(K_CONV3_X3, (cfl_conv3x3_x3_kernel<TM_, TN_>), dim3(grid), dim3(256), C::LDS_BYTES, stream, x, w, y, N, H, W, Ci, Co); \
    } while (0)
#define CFL_X3CONVS(TM_, TN_)                                                                                                  \
    do {                                                                                                                       \
        constexpr int BM_ = 64 * TM_, BN_ = 64 * TN_;                                                                          \
        if (W > 63) return CFL_ELIMIT;                                                                                         \
        const int LDS_ = ((BM_ + 2 * (W + 1) + 31) / 32) * (32 * 128) + 2 * BN_ * 128;                                         \
        const int grid = (int)((M + BM_ - 1) / BM_) * (Co / BN_);                                                              \
        CFL_SET_LDS((cfl_conv3x3_x3s_kernel<TM_, TN_>), (BM_ + 128) * 128 + 2 * BN_ * 128);                                    \
        CFL_LAUNCH(K_CONV3_X3, (cfl_conv3x3_x3s_kernel<TM_, TN_>), dim3(grid), dim3(256), LDS_, stream, x, w, y, N, H, W, Ci, Co); \
    } while (0)
    switch (variant) {
        case 122: if (!bn128) return CFL_ELIMIT; CFL_X3CONVS(2, 2); break;
        case 121: CFL_X3CONVS(2, 1); break;
        case 142: if (!bn128) return CFL_ELIMIT; CFL_X3CONVS(4, 2); break;
        case 141: CFL_X3CONVS(4, 1); break;
        case 42: if (!bn128) return CFL_ELIMIT; CFL_X3CONV(4, 2); break;
        case 22: if (!bn128) return CFL_ELIMIT; CFL_X3CONV(2, 2); break;
        case 41: CFL_X3CONV(4, 1); break;
        case 21: CFL_X3CONV(2, 1); break;
        default: return CFL_EINVAL;
    }
#undef CFL_X3CONV
#undef CFL_X3CONVS
    return 0;
}

extern "C" size_t cfl_conv3x3_x3_wimage_bytes(int Ci, int Co) {
    return (Ci > 0 && Co > 0 && Ci % 32 == 0) ? cfl_align256((size_t)9 * Ci * Co * 4) : 0;
}

extern "C" int cfl_conv3x3_x3_wimage(const float* w, int Ci, int Co, void* img, void* stream_) {
    if (!w || !img || Ci <= 0 || Co <= 0) return CFL_EINVAL;
    if (Ci % 32 != 0 || Co % 64 != 0 || (((uintptr_t)w | (uintptr_t)img) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const long long n = (long long)Co * 9 * (Ci / 4);
    CFL_LAUNCH(K_CONV3_X3_WIMAGE, cfl_conv3x3_wimage_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, Ci, Co, (char*)img);
    return 0;
}

extern "C" int cfl_conv3x3_x3_wimage_rot(const float* w, int Ci, int Co, void* img, void* stream_) {
    if (!w || !img || Ci <= 0 || Co <= 0) return CFL_EINVAL;
    if (Co % 32 != 0 || Ci % 64 != 0 || (((uintptr_t)w | (uintptr_t)img) & 15)) return CFL_ELIMIT;     // (roles swapped: K = 9 Co)
    hipStream_t stream = (hipStream_t)stream_;
    CFL_LAUNCH(K_CONV3_X3_WIMAGE, cfl_conv3x3_wimage_rot_kernel, dim3(Ci / 32, Co / 32, 9), dim3(256), 0, stream, w, Ci, Co, (char*)img);
    return 0;
}

// variant: 0 = chosen here; 2MN = TM = M, TN = N (222, 242, 221, 241, 212, 211)
extern "C" int cfl_conv3x3_x3_fwd_img(const float* x, const void* wimg, int N, int H, int W, int Ci, int Co, float* y, int variant,
                                      void* stream_) {
    if (!x || !wimg || !y) return CFL_EINVAL;
    if (!x3conv_ok(N, H, W, Ci, Co) || W > 63 || (((uintptr_t)x | (uintptr_t)wimg | (uintptr_t)y) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const long long M = (long long)N * H * W;
    const bool bn128 = Co % 128 == 0;
    if (variant == 0) {
        // measured at the four BasicBlock shapes of ResNet-18, batch 128 (profiles/r6_x3conv_probe_v3.jsonl): 128 positions x 128
        // channels where the channel count allows, x 64 otherwise; 64- and 256-position tiles lose 5-30 %
        // ... and the weight tiles by LDS-DMA (5xx): 3-5 % at three of the four shapes (profiles/r6_x3conv_decomposition.jsonl)
        // ... and requested behind the first half's MFMAs (8xx: the compiler then waits for the fragments it needs instead of
        // lgkmcnt(0)): 2-4 % at three shapes (profiles/r6_x3conv_late_dma.jsonl); 64-channel tiles where 128-channel ones leave CUs idle
        // (7 x 7 x 512 at batch 128: 196 tiles; 821 89.8 us, 522 99.5, 822 115).  CFL_X3_DMA_EARLY=1: the selection before (A/B).
        static const bool early = getenv("CFL_X3_DMA_EARLY") != nullptr;
        const long long t128 = ((M + 127) / 128) * (Co / 128);
        if (early) variant = bn128 ? 522 : 521;
        else variant = (bn128 && t128 >= 256) ? 822 : 821;
    }
#define CFL_X3CONVP(TM_, TN_, DBG_)                                                                                              \
    do {                                                                                                                       \
        constexpr int BM_ = 64 * TM_, BN_ = 64 * TN_;                                                                          \
        const int LDS_ = (((BM_ + 2 * (W + 1) + 31) / 32) * 32 + 1) * 144 + 2 * BN_ * 128;                                     \
        const int grid = (int)((M + BM_ - 1) / BM_) * (Co / BN_);                                                              \
        CFL_SET_LDS((cfl_conv3x3_x3p_kernel<TM_, TN_, DBG_>), (BM_ + 128 + 1) * 144 + 2 * BN_ * 128);                          \
        CFL_LAUNCH(K_CONV3_X3, (cfl_conv3x3_x3p_kernel<TM_, TN_, DBG_>), dim3(grid), dim3(256), LDS_, stream, x, (const char*)wimg, y, N, \
                   H, W, Ci, Co);                                                                                              \
    } while (0)
#define CFL_X3CONVQ(TM_, TN_)                                                                                                  \
    do {                                                                                                                       \
        constexpr int BM_ = 64 * TM_, BN_ = 64 * TN_;                                                                          \
        const int LDS_ = (((BM_ + 2 * (W + 1) + 15) / 16) * 16 + 1) * 144 + 3 * BN_ * 128;                                     \
        const int grid = (int)((M + BM_ - 1) / BM_) * (Co / BN_);                                                              \
        CFL_SET_LDS((cfl_conv3x3_x3q_kernel<TM_, TN_>), (BM_ + 128 + 1) * 144 + 3 * BN_ * 128);                                \
        CFL_LAUNCH(K_CONV3_X3, (cfl_conv3x3_x3q_kernel<TM_, TN_>), dim3(grid), dim3(256), LDS_, stream, x, (const char*)wimg, y, N, H, W, \
                   Ci, Co);                                                                                                    \
    } while (0)
    switch (variant) {
        case 222: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(2, 2, 0); break;
        case 242: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(4, 2, 0); break;
        case 212: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(1, 2, 0); break;
        case 221: CFL_X3CONVP(2, 1, 0); break;
        case 241: CFL_X3CONVP(4, 1, 0); break;
        case 211: CFL_X3CONVP(1, 1, 0); break;
        case 722: if (!bn128) return CFL_ELIMIT; CFL_X3CONVQ(2, 2); break;          // version 4: loader roles per wave, tiles two taps ahead
        case 721: CFL_X3CONVQ(2, 1); break;
        case 742: if (!bn128) return CFL_ELIMIT; CFL_X3CONVQ(4, 2); break;
        case 1222: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(2, 2, 1); break;      // measurement forms of 222 (wrong results)
        case 2222: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(2, 2, 2); break;
        case 3222: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(2, 2, 3); break;
        case 4222: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(2, 2, 4); break;
        case 6222: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(2, 2, 6); break;
        case 822: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(2, 2, 7); break;       // ... requested behind the first half's MFMAs
        case 842: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(4, 2, 7); break;
        case 821: CFL_X3CONVP(2, 1, 7); break;
        case 522: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(2, 2, 5); break;       // weight tiles by LDS-DMA
        case 542: if (!bn128) return CFL_ELIMIT; CFL_X3CONVP(4, 2, 5); break;
        case 521: CFL_X3CONVP(2, 1, 5); break;
        default: return CFL_EINVAL;
    }
#undef CFL_X3CONVP
#undef CFL_X3CONVQ
    return 0;
}

// stride 2 / padding 1 (even H, W <= 62): y [N, H / 2, W / 2, Co]; the weight image is the stride-1 one (cfl_conv3x3_x3_wimage)
extern "C" int cfl_conv3x3_x3_fwd_img_s2(const float* x, const void* wimg, int N, int H, int W, int Ci, int Co, float* y, void* stream_) {
    if (!x || !wimg || !y) return CFL_EINVAL;
    if (!x3conv_ok(N, H, W, Ci, Co) || W > 63 || (H & 1) || (W & 1) || (((uintptr_t)x | (uintptr_t)wimg | (uintptr_t)y) & 15))
        return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const long long M = (long long)N * (H / 2) * (W / 2);
#define CFL_X3CONVS2(TN_)                                                                                                      \
    do {                                                                                                                       \
        constexpr int BM_ = 128, BN_ = 64 * TN_, LDS_ = (4 * BM_ + 256 + 1) * 144 + 2 * BN_ * 128;                             \
        const int grid = (int)((M + BM_ - 1) / BM_) * (Co / BN_);                                                              \
        CFL_SET_LDS((cfl_conv3x3_x3p_kernel<2, TN_, 7, 2>), LDS_);                                                              \
        CFL_LAUNCH(K_CONV3_X3, (cfl_conv3x3_x3p_kernel<2, TN_, 7, 2>), dim3(grid), dim3(256), LDS_, stream, x, (const char*)wimg, y, N, H, \
                   W, Ci, Co);                                                                                                 \
    } while (0)
    if (Co % 128 == 0) CFL_X3CONVS2(2);
    else CFL_X3CONVS2(1);
#undef CFL_X3CONVS2
    return 0;
}

extern "C" int cfl_conv3x3_x3_rot_weight(const float* w, int Ci, int Co, float* w_rot, void* stream_) {
    if (!w || !w_rot || Ci <= 0 || Co <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    CFL_LAUNCH(K_TRANSPOSE, cfl_conv3x3_rot_kernel, dim3(cfl_cdiv(Ci, 32), cfl_cdiv(Co, 32), 9), dim3(256), 0, stream, w, Ci, Co, w_rot);
    return 0;
}
