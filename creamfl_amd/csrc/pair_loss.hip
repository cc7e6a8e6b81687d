// pair_loss.hip -- row A1: PCME soft-contrastive loss over all N^2 image/caption pairs.
//
// Reference: src/criterions/probemb.py:7-86,150-256 (MCSoftContrastiveLoss).  The reference
// gathers N^2 x D operands and evaluates (a-b)^2 element-wise; here the pair-similarity matrix
// S = I T^T is ONE fp32 MFMA GEMM (exact fp32 FMA chains) and d^2 = |I_i|^2 + |T_j|^2 - 2 S_ij.
// The N diagonal (positive) pairs dominate the loss and suffer the cancellation of the GEMM
// form when matched pairs are close, so their distances are recomputed exactly as
// sum (I_ik - T_ik)^2 by cfl_pair_prep_kernel.
//
// HBM layout: I, T [N, D] row-major fp32; coef [2, N, N] row-major fp32 (the coefficient matrix AND its
// transpose: with the transposed features It, Tt [D, N] kept in ws, both backward GEMMs  coef @ T  and
// coef^T @ I  read K-contiguous operands and take the direct-to-LDS path); ws (floats):
//   ni[N] nt[N] dd[N] rowsum[N] colsum[N] rowpart[NT*N] colpart[NT*N] part[NT*NT*4] It[D*N] Tt[D*N]
// with NT = ceil(N / 64) (sized for the smallest tile).
//
// IMAGE MODE (round 4; N >= 2048, N % 32 == 0, D % 32 == 0, 3 x bf16-split precision): every GEMM operand lives in HBM as a
// PRE-SPLIT image (tile_x3.h: a 128-byte block = [32 x bf16 hi | 32 x bf16 lo] of 32 consecutive k, byte for byte the fp32
// size) -- the features and their transposes written once by cfl_pair_images_kernel (It / Tt region of ws, plus 2 N D floats
// for the untransposed images), the coefficient matrix and its transpose written in split form by the FORWARD EPILOGUE.  All
// three GEMMs then stage by LDS-DMA (global_load_lds) with no conversion VALU and no ds_write; the backward GEMMs (K = N:
// 128 stages at N = 4096, one 128 x 128 tile per CU) run through a 4-stage LDS ring with counted vmcnt waits (three stages
// in flight) instead of a double buffer whose every barrier waits for a fresh round trip to L2 / HBM.
#include "common.h"
#include "tile_x3.h"

namespace {

struct PairWs {
    float *ni, *nt, *dd, *rowsum, *colsum, *rowpart, *colpart, *part, *it, *tt, *img_i, *img_t, *bpart;
};
static PairWs pair_ws(void* ws, int N, int D) {
    const int NT = cfl_cdiv(N, 64);
    float* p = (float*)ws;
    PairWs w;
    w.ni = p; p += N; w.nt = p; p += N; w.dd = p; p += N;
    w.rowsum = p; p += N; w.colsum = p; p += N;
    w.rowpart = p; p += (size_t)NT * N; w.colpart = p; p += (size_t)NT * N;
    p = (float*)(((uintptr_t)p + 15) & ~(uintptr_t)15);      // part is read as float4
    w.part = p; p += (size_t)4 * NT * NT;
    p = (float*)cfl_align256((size_t)(uintptr_t)p);          // 16-byte alignment for vector access
    w.it = p; p += (size_t)D * N; w.tt = p; p += (size_t)D * N;
    w.img_i = p; p += (size_t)N * D; w.img_t = p; p += (size_t)N * D;   // image mode only (N % 32 == 0, D % 32 == 0)
    w.bpart = p;                                           // big-tile backward only: [2][KSPLIT][N][D] split-K partials
    return w;
}

// It[k][i] = I[i][k], Tt likewise (blockIdx.z selects); 32x32 LDS tiles, coalesced both ways
__global__ __launch_bounds__(256) void cfl_pair_transpose_kernel(const float* __restrict__ I, const float* __restrict__ T,
                                                                 int N, int D, float* It, float* Tt) {
    __shared__ float tile[32][33];
    const float* src = blockIdx.z ? T : I;
    float* dst = blockIdx.z ? Tt : It;
    const int i0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // 32 x 8
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (i0 + r < N && k0 + tx < D) ? src[(long long)(i0 + r) * D + k0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < D && i0 + tx < N) dst[(long long)(k0 + r) * N + i0 + tx] = tile[tx][r];
}

// one wave per row: |I_i|^2, |T_i|^2 and the exact diagonal squared distance.
__global__ __launch_bounds__(256) void cfl_pair_prep_kernel(const float* __restrict__ I, const float* __restrict__ T,
                                                            int N, int D, float* ni, float* nt, float* dd) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* a = I + (long long)row * D;
    const float* b = T + (long long)row * D;
    float sa = 0.f, sb = 0.f, sd = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float x = a[k], y = b[k], e = x - y;
        sa = fmaf(x, x, sa); sb = fmaf(y, y, sb); sd = fmaf(e, e, sd);
    }
    sa = wave_sum(sa); sb = wave_sum(sb); sd = wave_sum(sd);
    if (lane == 0) { ni[row] = sa; nt[row] = sb; dd[row] = sd; }
}

// IMAGE MODE.  One pass over I and T (blockIdx.z selects) writes the four operand images of the three GEMMs: the feature image
// [N rows][D k] for S = I T^T and the TRANSPOSED feature image [D rows][N k] (B operand of the backward GEMMs: K runs over the
// pair index).  32 x 32 element tiles through LDS; a thread splits 4 consecutive k of one row (8 bytes hi, 8 bytes lo).
__global__ __launch_bounds__(256) void cfl_pair_images_kernel(const float* __restrict__ I, const float* __restrict__ T, int N, int D,
                                                              float* img_i, float* img_t, float* img_it, float* img_tt) {
    __shared__ float tile[32][33];
    const float* src = blockIdx.z ? T : I;
    char* img = reinterpret_cast<char*>(blockIdx.z ? img_t : img_i);
    char* img_tr = reinterpret_cast<char*>(blockIdx.z ? img_tt : img_it);
    const int i0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = src[(long long)(i0 + r) * D + k0 + tx];
    __syncthreads();
    const int r = threadIdx.x >> 3, q = threadIdx.x & 7;            // 32 rows x 8 quads
    x3::bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { __bf16 h, l; x3::split1(tile[r][4 * q + e], h, l); hi[e] = h; lo[e] = l; }
    char* blk = img + ((long long)(i0 + r) * D + k0) * 4;         // row i0 + r, k block k0 / 32
    *reinterpret_cast<x3::bf16x4*>(blk + q * 8) = hi;
    *reinterpret_cast<x3::bf16x4*>(blk + 64 + q * 8) = lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { __bf16 h, l; x3::split1(tile[4 * q + e][r], h, l); hi[e] = h; lo[e] = l; }
    blk = img_tr + ((long long)(k0 + r) * N + i0) * 4;            // row k0 + r of the transpose, k block i0 / 32
    *reinterpret_cast<x3::bf16x4*>(blk + q * 8) = hi;
    *reinterpret_cast<x3::bf16x4*>(blk + 64 + q * 8) = lo;
}

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void cfl_pair_fwd_kernel(Opnd A, Opnd B, int N, const float* __restrict__ a_dev, const float* __restrict__ b_dev, float eps,
                                                           const float* __restrict__ ni, const float* __restrict__ nt,
                                                           const float* __restrict__ dd, float* coef,
                                                           float* rowpart, float* colpart, float* part, int x3mode) {
    using C = TileCfg<TM, TN, true, true>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntc = (N + C::BN - 1) / C::BN, ntr = (N + C::BM - 1) / C::BM;
    int ti, tj;
    tile_swizzle(xcd_remap(blockIdx.x, gridDim.x), ntr, ntc, ti, tj);
    const int tile = ti * ntc + tj;
    const int row0 = ti * C::BM, col0 = tj * C::BN;
    const float a = a_dev[0], b = b_dev[0];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    constexpr int CLD = C::BN + 1;
    float* cs = lds;                                          // [BM][BN+1] coefficient tile (after the K loop)
    float pos = 0.f, neg = 0.f, da = 0.f, db = 0.f;
    f32x16 acc[TM][TN];
    if (x3mode == 2)         // A, B are pre-split images (cfl_pair_images_kernel): LDS-DMA staging, bf16 x 3 compute
        tile_gemm_seq_glds_with<TM, TN>(A, B, 1, [&](int) { return TileDesc{row0, col0, 0, A.kdim}; }, lds,
                                        [&](int, const f32x16 (&a)[TM][TN]) {
#pragma unroll
                                            for (int m = 0; m < TM; ++m)
#pragma unroll
                                                for (int n = 0; n < TN; ++n) acc[m][n] = a[m][n];
                                        },
                                        [](const float* sa, const float* sb, f32x16 (&c)[TM][TN], int ln, int r_, int c_) {
                                            x3::compute<TM, TN>(reinterpret_cast<const char*>(sa), reinterpret_cast<const char*>(sb), c,
                                                                ln, r_, c_);
                                        });
    else if (x3mode) x3::tile_gemm<TM, TN, true, true>(A, B, row0, col0, 0, A.kdim, lds, acc, XfIdentity());
    else if (glds_ok(A, B)) tile_gemm_glds<TM, TN>(A, B, row0, col0, 0, A.kdim, lds, acc);
    else tile_gemm<TM, TN, true, true>(A, B, row0, col0, 0, A.kdim, lds, acc, XfIdentity());
    // Round 6: a tile that lies inside the matrix and off its diagonal (all but N / BM of the (N / BM)^2 tiles) takes a lean form of
    // the per-pair arithmetic -- no bounds or diagonal selects, m = -1 folded, d and 1 / d from ONE v_rsq (d^2 + eps >= eps > 0: no
    // denormal scaling), log(1 + e) on v_log directly (1 + e in [1, 2]): ~30 vector instructions per pair where the general form
    // compiles to ~100 (13 compares and 12 selects per pair); at N = 4096 the epilogue was ~half of the kernel's SIMD time.
    const bool lean = row0 + C::BM <= N && col0 + C::BN <= N && (row0 + C::BM <= col0 || col0 + C::BN <= row0);
    if (lean) {
        const float a2 = -2.f * a, b2 = 2.f * b;
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            float nim[16];                                        // (one block of 16 pairs at a time: left to itself the compiler hoists
#pragma unroll                                                    //  every load of the 64 pairs and spills past 256 registers)
            for (int r = 0; r < 16; ++r) nim[r] = ni[row0 + acc_row<TM>(wr, m, r, lane)];
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int lc = acc_col<TN>(wc, n, lane);
                const float ntj = nt[col0 + lc];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = acc_row<TM>(wr, m, r, lane);
                    const float t = fmaxf(nim[r] + ntj - 2.f * acc[m][n][r], 0.f) + eps;
                    const float rs = __builtin_amdgcn_rsqf(t);
                    const float d = t * rs;
                    const float x = fmaf(a2, d, b2);                  // -2 m s with m = -1, s = b - a d
                    const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.4426950408889634f);
                    float lg = __builtin_amdgcn_logf(1.f + e) * 0.6931471805599453f;
                    asm volatile("" : "+v"(lg));                      // (both forms evaluated, ONE select: as a ternary the compiler
                    const float l1p = e < 1e-2f ? e * fmaf(e, fmaf(e, 0.33333334f, -0.5f), 1.f) : lg;   // branches per pair and spills)
                    const float nll = fmaxf(x, 0.f) + l1p;
                    const float g = -4.f * (x >= 0.f ? 1.f : e) * __builtin_amdgcn_rcpf(1.f + e);
                    neg += nll;
                    da = fmaf(g, d, da);
                    db -= g;
                    cs[lr * CLD + lc] = a * g * rs;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int lc = acc_col<TN>(wc, n, lane);
            const int j = col0 + lc;
            const float ntj = j < N ? nt[j] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = acc_row<TM>(wr, m, r, lane);
                const int i = row0 + lr;
                float c = 0.f;
                if (i < N && j < N) {
                    const bool diag = (i == j);
                    const float d2 = diag ? dd[i] : fmaxf(ni[i] + ntj - 2.f * acc[m][n][r], 0.f);
                    const float d = sqrtf(d2 + eps);
                    const float s = fmaf(-a, d, b);
                    const float mm = diag ? 1.f : -1.f;
                    const float x = -2.f * mm * s;
                    // softplus(x) and sigmoid(x) from ONE exponential e = exp(-|x|):
                    //   softplus = max(x,0) + log1p(e),  sigmoid = (x >= 0 ? 1 : e) / (1 + e)
                    const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.4426950408889634f);
                    const float l1p = e < 1e-2f ? e * fmaf(e, fmaf(e, 0.33333334f, -0.5f), 1.f) : __logf(1.f + e);
                    const float nll = fmaxf(x, 0.f) + l1p;
                    const float sg = (x >= 0.f ? 1.f : e) * __builtin_amdgcn_rcpf(1.f + e);
                    const float g = 4.f * mm * sg;              // dL/dd / a   (both directions)
                    c = a * g * __builtin_amdgcn_rcpf(d);
                    if (diag) pos += nll; else neg += nll;
                    da = fmaf(g, d, da);
                    db -= g;
                }
                cs[lr * CLD + lc] = c;
            }
        }
    __shared__ float red[4][4];
    pos = wave_sum(pos); neg = wave_sum(neg); da = wave_sum(da); db = wave_sum(db);
    if (lane == 0) { red[wid][0] = pos; red[wid][1] = neg; red[wid][2] = da; red[wid][3] = db; }
    __syncthreads();                      // also: the coefficient tile cs is complete
    if (threadIdx.x < 4) {
        const int e = threadIdx.x;
        part[(size_t)tile * 4 + e] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    }
    const int t = threadIdx.x;
    if (t < C::BM) {
        float s = 0.f;
        for (int j = 0; j < C::BN; ++j) s += cs[t * CLD + j];
        if (row0 + t < N) rowpart[(size_t)tj * N + row0 + t] = s;
    } else if (t < C::BM + C::BN) {
        const int cj = t - C::BM;
        float s = 0.f;
        for (int i = 0; i < C::BM; ++i) s += cs[i * CLD + cj];
        if (col0 + cj < N) colpart[(size_t)ti * N + col0 + cj] = s;
    }
    if (coef && x3mode == 2) {
        // the coefficient tile and its transpose as split IMAGES (the A operands of the two backward GEMMs): per quad of 4
        // consecutive k one 8-byte hi store and one 8-byte lo store; 8 threads fill one 128-byte block (N % 32 == 0 here)
        char* ci = reinterpret_cast<char*>(coef);
        char* ct = reinterpret_cast<char*>(coef + (long long)N * N);
        for (int e = t; e < C::BM * C::BN / 4; e += 256) {
            const int lr = e / (C::BN / 4), q = e % (C::BN / 4);            // row lr, columns 4 q .. 4 q + 3
            if (row0 + lr < N && col0 + 4 * q < N) {
                x3::bf16x4 hi, lo;
#pragma unroll
                for (int u = 0; u < 4; ++u) { __bf16 h, l; x3::split1(cs[lr * CLD + 4 * q + u], h, l); hi[u] = h; lo[u] = l; }
                char* blk = ci + ((long long)(row0 + lr) * N + col0 + ((4 * q) & ~31)) * 4;
                *reinterpret_cast<x3::bf16x4*>(blk + ((4 * q) & 31) * 2) = hi;
                *reinterpret_cast<x3::bf16x4*>(blk + 64 + ((4 * q) & 31) * 2) = lo;
            }
        }
        for (int e = t; e < C::BM * C::BN / 4; e += 256) {
            const int lc = e / (C::BM / 4), q = e % (C::BM / 4);            // transposed: row = column lc, k = rows 4 q .. 4 q + 3
            if (col0 + lc < N && row0 + 4 * q < N) {
                x3::bf16x4 hi, lo;
#pragma unroll
                for (int u = 0; u < 4; ++u) { __bf16 h, l; x3::split1(cs[(4 * q + u) * CLD + lc], h, l); hi[u] = h; lo[u] = l; }
                char* blk = ct + ((long long)(col0 + lc) * N + row0 + ((4 * q) & ~31)) * 4;
                *reinterpret_cast<x3::bf16x4*>(blk + ((4 * q) & 31) * 2) = hi;
                *reinterpret_cast<x3::bf16x4*>(blk + 64 + ((4 * q) & 31) * 2) = lo;
            }
        }
    } else if (coef) {
        float* coef_t = coef + (long long)N * N;
        for (int e = t; e < C::BM * C::BN; e += 256) {
            const int lr = e / C::BN, lc = e % C::BN;
            if (row0 + lr < N && col0 + lc < N) coef[(long long)(row0 + lr) * N + col0 + lc] = cs[lr * CLD + lc];
        }
        for (int e = t; e < C::BM * C::BN; e += 256) {         // transposed copy: rows of the tile are the fast axis
            const int lc = e / C::BM, lr = e % C::BM;
            if (row0 + lr < N && col0 + lc < N) coef_t[(long long)(col0 + lc) * N + row0 + lr] = cs[lr * CLD + lc];
        }
    }
}

// deterministic reductions: out8 from per-tile partials; rowsum/colsum from per-tile-column partials
__global__ __launch_bounds__(256) void cfl_pair_final_kernel(const float* part, int ntiles, const float* rowpart,
                                                             const float* colpart, int ntr, int ntc, int N,
                                                             float* rowsum, float* colsum, float* out8) {
    __shared__ float red[4];
    if (blockIdx.x == 0) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = threadIdx.x; t < ntiles; t += 256) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(part + (size_t)t * 4);
            for (int e = 0; e < 4; ++e) v[e] += q[e];
        }
        for (int e = 0; e < 4; ++e) v[e] = block_sum_256(v[e], red);
        if (threadIdx.x == 0) {
            out8[0] = 2.f * (v[0] + v[1]); out8[1] = v[0]; out8[2] = v[1]; out8[3] = v[2]; out8[4] = v[3];
            out8[5] = 0.f; out8[6] = 0.f; out8[7] = 0.f;
        }
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) {
        // 8 independent loads in flight per trip (one load per trip was a chain of ntc + ntr L2 round trips: 19 us at N = 4096);
        // the summation order stays fixed (t ascending) => deterministic
        auto colsum_of = [&](const float* src, int n) {
            float s = 0.f;
            int t = 0;
            for (; t + 8 <= n; t += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(t + u) * N + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; t < n; ++t) s += src[(size_t)t * N + i];
            return s;
        };
        rowsum[i] = colsum_of(rowpart, ntc);
        colsum[i] = colsum_of(colpart, ntr);
    }
}

// z = 0: dI = gout * (I * rowsum - coef   @ T)   A = coef   [N, N],  B = Tt [D, N]   (both K-contiguous)
// z = 1: dT = gout * (T * colsum - coef^T @ I)   A = coef^T [N, N],  B = It [D, N]
template <int TM, int TN>
__global__ __launch_bounds__(256) void cfl_pair_bwd_kernel(const float* __restrict__ I, const float* __restrict__ T,
                                                           const float* __restrict__ coef, const float* __restrict__ It,
                                                           const float* __restrict__ Tt, int N, int D, int vecN,
                                                           const float* __restrict__ rowsum, const float* __restrict__ colsum,
                                                           const float* __restrict__ gout, float* dI, float* dT, int x3mode) {
    using C = TileCfg<TM, TN, true, true>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntc = (D + C::BN - 1) / C::BN, ntr = (N + C::BM - 1) / C::BM;
    int ti, tj;
    tile_swizzle(xcd_remap(blockIdx.x, gridDim.x), ntr, ntc, ti, tj);
    const int row0 = ti * C::BM, col0 = tj * C::BN;
    const bool second = blockIdx.z != 0;
    const float* X = second ? T : I;        // the tensor whose gradient this block produces
    Opnd Ao{second ? coef + (long long)N * N : coef, N, N, N, vecN};
    Opnd Bo{second ? It : Tt, N, D, N, vecN};
    f32x16 acc[TM][TN];
    if (x3mode) x3::tile_gemm<TM, TN, true, true>(Ao, Bo, row0, col0, 0, N, lds, acc, XfIdentity());
    else if (glds_ok(Ao, Bo)) tile_gemm_glds<TM, TN>(Ao, Bo, row0, col0, 0, N, lds, acc);
    else tile_gemm<TM, TN, true, true>(Ao, Bo, row0, col0, 0, N, lds, acc, XfIdentity());
    const float* sums = second ? colsum : rowsum;
    float* out = second ? dT : dI;
    const float g = gout[0];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int j = col0 + acc_col<TN>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = row0 + acc_row<TM>(wr, m, r, lane);
                if (i < N && j < D) {
                    const long long o = (long long)i * D + j;
                    out[o] = g * (sums[i] * X[o] - acc[m][n][r]);
                }
            }
        }
}

// IMAGE MODE backward: the same two GEMMs on pre-split operand images (coefficient image / its transpose written by the
// forward epilogue, transposed feature images by cfl_pair_images_kernel), 128 x 128 tiles, K = N.  At N = 4096, D = 512 the
// launch is exactly one tile per CU (256 workgroups, one wave per SIMD) with 128 K stages each: nothing else on the CU hides a
// stage's round trip to L2 / HBM, so the stages run through a ring of NS LDS buffers with NS - 1 of them in flight -- the wait
// before a stage's MFMA block is a COUNTED s_waitcnt (8 LDS-DMA instructions per wave and stage; the younger stages stay
// outstanding) followed by a bare s_barrier; __syncthreads() would drain vmcnt to 0 and serialise memory latency with compute.
template <int NS>
__global__ __launch_bounds__(256) void cfl_pair_bwd_img_kernel(const float* __restrict__ I, const float* __restrict__ T,
                                                               const float* __restrict__ coef, const float* __restrict__ img_it,
                                                               const float* __restrict__ img_tt, int N, int D,
                                                               const float* __restrict__ rowsum, const float* __restrict__ colsum,
                                                               const float* __restrict__ gout, float* dI, float* dT) {
    constexpr int TM = 2, TN = 2, BM = 128, BN = 128, STAGE = (BM + BN) * 32;      // floats per stage (32 KB)
    static_assert(NS == 4, "the counted waits below are written for three stages in flight");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntc = (D + BN - 1) / BN, ntr = (N + BM - 1) / BM;
    int ti, tj;
    tile_swizzle(xcd_remap(blockIdx.x, gridDim.x), ntr, ntc, ti, tj);
    const int row0 = ti * BM, col0 = tj * BN;
    const bool second = blockIdx.z != 0;
    const float* X = second ? T : I;
    const Opnd Ao{second ? coef + (long long)N * N : coef, N, N, N, 1};
    const Opnd Bo{second ? img_it : img_tt, N, D, N, 1};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int nk = N / 32;
    auto issue = [&](int kt) {
        float* st = lds + (kt % NS) * STAGE;
        glds_stage<BM>(Ao, row0, kt * 32, st);
        glds_stage<BN>(Bo, col0, kt * 32, st + BM * 32);
    };
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt must have landed (this wave's part: vmcnt; everybody's: the barrier); stages kt + 1, kt + 2 may still fly.
        // The barrier also says that every wave is done reading the buffer stage kt + 3 is about to overwrite (stage kt - 1).
        const int rem = nk - 1 - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
        const char* sa = reinterpret_cast<const char*>(lds + (kt % NS) * STAGE);
        x3::compute<TM, TN>(sa, sa + BM * 128, acc, lane, wr, wc);
    }
    const float* sums = second ? colsum : rowsum;
    float* out = second ? dT : dI;
    const float g = gout[0];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int j = col0 + acc_col<TN>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = row0 + acc_row<TM>(wr, m, r, lane);
                if (i < N && j < D) {
                    const long long o = (long long)i * D + j;
                    out[o] = g * (sums[i] * X[o] - acc[m][n][r]);
                }
            }
        }
}

// IMAGE MODE backward, 256 x 256 tiles + split-K (round 6).  The kernel above is bound by the L2 -> LDS path, not by HBM or the matrix
// pipe: 256 workgroups x 128 stages x 32 KB = 1 GB per launch through it in 137 us = 7.6 TB/s of the 8-9 that path sustains, for
// 41 us of MFMA work.  A 256 x 256 tile moves half the bytes per flop (64 KB per stage for four times the MFMAs): N D / 256^2 tiles
// per GEMM are too few to fill the chip, so the contraction (K = N) is split KS ways -- (N / 256) (D / 256) x KS x 2 workgroups, one
// per CU, one wave per SIMD with 256 accumulator registers -- and the fp32 partials [2][KS][N][D] are summed in fixed order by
// cfl_pair_bwd_reduce_kernel, which also applies the epilogue  g (sums . X - acc).  Two 64 KB stages (the 4-stage ring does not fit
// and a stage is 3072 MFMA cycles long: one stage ahead covers its load).
__global__ __launch_bounds__(256, 1) void cfl_pair_bwd_img_big_kernel(const float* __restrict__ coef, const float* __restrict__ img_it,
                                                                     const float* __restrict__ img_tt, int N, int D, int ksplit,
                                                                     float* __restrict__ part) {
    constexpr int TM = 4, TN = 4, BM = 256, BN = 256, STAGE = (BM + BN) * 32;      // floats per stage (64 KB)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntc = D / BN, ntr = N / BM;
    int ti, tj;
    tile_swizzle(xcd_remap(blockIdx.x, gridDim.x), ntr, ntc, ti, tj);
    const int row0 = ti * BM, col0 = tj * BN;
    const int ks = blockIdx.y;
    const bool second = blockIdx.z != 0;
    const Opnd Ao{second ? coef + (long long)N * N : coef, N, N, N, 1};
    const Opnd Bo{second ? img_it : img_tt, N, D, N, 1};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int nk = N / 32 / ksplit, kt0 = ks * nk;
    auto issue = [&](int kt) {
        float* st = lds + (kt & 1) * STAGE;
        glds_stage<BM>(Ao, row0, (kt0 + kt) * 32, st);
        glds_stage<BN>(Bo, col0, (kt0 + kt) * 32, st + BM * 32);
    };
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) issue(kt + 1);
        const char* sa = reinterpret_cast<const char*>(lds + (kt & 1) * STAGE);
        x3::compute<TM, TN>(sa, sa + BM * 128, acc, lane, wr, wc);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float* out = part + ((long long)(second ? 1 : 0) * ksplit + ks) * N * D;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int j = col0 + acc_col<TN>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = row0 + acc_row<TM>(wr, m, r, lane);
                out[(long long)i * D + j] = acc[m][n][r];
            }
        }
}

// dI / dT = g (sums . X - sum over the KS split partials, ks ascending): one thread per 16 bytes of [2][N][D]
__global__ __launch_bounds__(256) void cfl_pair_bwd_reduce_kernel(const float* __restrict__ I, const float* __restrict__ T,
                                                                  const float* __restrict__ part, int N, int D, int ksplit,
                                                                  const float* __restrict__ rowsum, const float* __restrict__ colsum,
                                                                  const float* __restrict__ gout, float* __restrict__ dI,
                                                                  float* __restrict__ dT) {
    const long long nd4 = (long long)N * D / 4;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= 2 * nd4) return;
    const int z = e >= nd4;
    const long long o = (e - (z ? nd4 : 0)) * 4;
    const int i = (int)(o / D);
    const float* X = z ? T : I;
    const float* sums = z ? colsum : rowsum;
    float* out = z ? dT : dI;
    const float* p = part + (long long)z * ksplit * N * D + o;
    f32x4 s = *reinterpret_cast<const f32x4*>(p);
    for (int k = 1; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(p + (long long)k * N * D);
    const f32x4 x = *reinterpret_cast<const f32x4*>(X + o);
    const float g = gout[0], sm = sums[i];
    *reinterpret_cast<f32x4*>(out + o) = (x * sm - s) * g;
}

// split factor of the big-tile backward: 0 = not taken (shape, or switched off for an A/B: CFL_PAIR_BWD_BIG=0)
static inline int pair_bwd_big_ksplit(int N, int D) {
    static const bool off = [] { const char* e = getenv("CFL_PAIR_BWD_BIG"); return e && e[0] == '0'; }();
    if (off || N % 256 != 0 || D % 256 != 0) return 0;
    const long long tiles = (long long)(N / 256) * (D / 256) * 2;
    int ks = 1;
    while (ks < 8 && tiles * ks < 224 && (N / 32) % (2 * ks) == 0) ks *= 2;
    return tiles * ks >= 128 ? ks : 0;                       // (too few workgroups even split 8 ways: the 128 x 128 kernel)
}

// image mode: 3 x bf16-split precision, shapes the split images tile without padding, and enough 128 x 128 tiles to fill the
// chip (below N = 2048 the smaller tiles of the register-staged path give more workgroups)
static inline bool pair_image_mode(int N, int D) {
    return !cfl_get_exact_gemm() && N >= 2048 && N % 32 == 0 && D % 32 == 0;
}

}  // namespace

extern "C" {

size_t cfl_pair_loss_ws_bytes(int N, int D) {
    (void)D;
    if (N <= 0) return 256;
    const size_t NT = (size_t)cfl_cdiv(N, 64);
    // 2 N D: It, Tt (image mode: their split images); + 2 N D: the untransposed feature images of image mode
    // + 2 KS N D: the split-K partials of the big-tile backward (image mode, N % 256 == 0, D % 256 == 0; KS <= 8)
    const size_t bp = (N >= 2048 && N % 256 == 0 && D % 256 == 0) ? (size_t)16 * N * D : 0;
    return cfl_align256((5 * (size_t)N + 2 * NT * N + 4 * NT * NT + 4 * (size_t)N * (D > 0 ? D : 1) + bp) * sizeof(float)) + 512;
}

int cfl_pair_loss_fwd(const float* I, const float* T, int N, int D, const float* a_dev, const float* b_dev, float eps,
                      float* out8, float* coef, void* ws, void* stream_) {
    if (!I || !T || !a_dev || !b_dev || !out8 || !ws || N <= 0 || D <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    PairWs w = pair_ws(ws, N, D);
    const bool img = pair_image_mode(N, D);
    if (img)        // the four operand images in one pass (the transposed ones are only read by the backward, but cost nothing extra)
        CFL_LAUNCH(K_PAIR_PREP, cfl_pair_images_kernel, dim3(D / 32, N / 32, 2), dim3(256), 0, stream, I, T, N, D, w.img_i, w.img_t,
                   w.it, w.tt);
    else if (coef)
        CFL_LAUNCH(K_PAIR_PREP, cfl_pair_transpose_kernel, dim3(cfl_cdiv(D, 32), cfl_cdiv(N, 32), 2), dim3(256), 0, stream,
                   I, T, N, D, w.it, w.tt);
    CFL_LAUNCH(K_PAIR_PREP, cfl_pair_prep_kernel, dim3(cfl_cdiv(N, 4)), dim3(256), 0, stream, I, T, N, D, w.ni, w.nt, w.dd);
    Opnd A{img ? w.img_i : I, D, N, D, img ? 1 : cfl_opnd_vec(I, D, D)};
    Opnd B{img ? w.img_t : T, D, N, D, img ? 1 : cfl_opnd_vec(T, D, D)};
    // S = I T^T on the 3 x bf16-split MFMA (positives stay exact: dd); 2 = from pre-split images, coefficients written as images
    const int x3mode = cfl_get_exact_gemm() ? 0 : (img ? 2 : 1);
    // 128x128 tiles once they fill the chip, 64x64 tiles below that (latency-bound regime)
    const bool big = (long long)cfl_cdiv(N, 128) * cfl_cdiv(N, 128) >= 256;
    int ntr, ntc;
    if (big) {
        using C = TileCfg<2, 2, true, true>;
        ntr = cfl_cdiv(N, C::BM); ntc = cfl_cdiv(N, C::BN);
        CFL_SET_LDS((cfl_pair_fwd_kernel<2, 2>), C::LDS_BYTES);
        CFL_LAUNCH(K_PAIR_FWD, (cfl_pair_fwd_kernel<2, 2>), dim3(ntr * ntc), dim3(256), C::LDS_BYTES, stream,
                   A, B, N, a_dev, b_dev, eps, w.ni, w.nt, w.dd, coef, w.rowpart, w.colpart, w.part, x3mode);
    } else {
        using C = TileCfg<1, 1, true, true>;
        ntr = cfl_cdiv(N, C::BM); ntc = cfl_cdiv(N, C::BN);
        CFL_LAUNCH(K_PAIR_FWD, (cfl_pair_fwd_kernel<1, 1>), dim3(ntr * ntc), dim3(256), C::LDS_BYTES, stream,
                   A, B, N, a_dev, b_dev, eps, w.ni, w.nt, w.dd, coef, w.rowpart, w.colpart, w.part, x3mode);
    }
    CFL_LAUNCH(K_PAIR_FINAL, cfl_pair_final_kernel, dim3(cfl_cdiv(N, 256)), dim3(256), 0, stream,
               w.part, ntr * ntc, w.rowpart, w.colpart, ntr, ntc, N, w.rowsum, w.colsum, out8);
    return 0;
}

int cfl_pair_loss_bwd(const float* I, const float* T, const float* coef, int N, int D,
                      const float* gout_dev, float* dI, float* dT, void* ws, void* stream_) {
    if (!I || !T || !coef || !gout_dev || !dI || !dT || !ws || N <= 0 || D <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    PairWs w = pair_ws(ws, N, D);
    if (pair_image_mode(N, D)) {          // coef / It / Tt hold split images (cfl_pair_loss_fwd took the same branch)
        if (const int ks = pair_bwd_big_ksplit(N, D)) {
            constexpr int LDSB2 = 2 * 512 * 128;
            CFL_SET_LDS(cfl_pair_bwd_img_big_kernel, LDSB2);
            CFL_LAUNCH(K_PAIR_BWD, cfl_pair_bwd_img_big_kernel, dim3((N / 256) * (D / 256), ks, 2), dim3(256), LDSB2, stream, coef, w.it, w.tt,
                       N, D, ks, w.bpart);
            const long long n4 = (long long)N * D / 2;
            CFL_LAUNCH(K_PAIR_BWD_REDUCE, cfl_pair_bwd_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, I, T, w.bpart, N, D, ks,
                       w.rowsum, w.colsum, gout_dev, dI, dT);
            return 0;
        }
        constexpr int NS = 4, LDSB = NS * 256 * 128;
        CFL_SET_LDS((cfl_pair_bwd_img_kernel<NS>), LDSB);
        CFL_LAUNCH(K_PAIR_BWD, (cfl_pair_bwd_img_kernel<NS>), dim3(cfl_cdiv(N, 128) * cfl_cdiv(D, 128), 1, 2), dim3(256), LDSB, stream,
                   I, T, coef, w.it, w.tt, N, D, w.rowsum, w.colsum, gout_dev, dI, dT);
        return 0;
    }
    const int vecN = (cfl_opnd_vec(coef, N, N) && cfl_vec_ok(w.it, N) && cfl_vec_ok(w.tt, N)) ? 1 : 0;
    const int x3mode = cfl_get_exact_gemm() ? 0 : 1;
    // tile choice by workgroup count (two GEMMs share the launch, grid.z = 2): 128x128 when that alone gives >= 2
    // workgroups per CU, else 128x64, else 64x64 (latency-bound small batches)
    const long long t128 = (long long)cfl_cdiv(N, 128) * cfl_cdiv(D, 128) * 2;
    const long long t12864 = (long long)cfl_cdiv(N, 128) * cfl_cdiv(D, 64) * 2;
    if (t128 >= 512) {
        using C = TileCfg<2, 2, true, true>;
        CFL_SET_LDS((cfl_pair_bwd_kernel<2, 2>), C::LDS_BYTES);
        CFL_LAUNCH(K_PAIR_BWD, (cfl_pair_bwd_kernel<2, 2>), dim3(cfl_cdiv(N, C::BM) * cfl_cdiv(D, C::BN), 1, 2), dim3(256),
                   C::LDS_BYTES, stream, I, T, coef, w.it, w.tt, N, D, vecN, w.rowsum, w.colsum, gout_dev, dI, dT, x3mode);
    } else if (t12864 >= 256) {
        using C = TileCfg<2, 1, true, true>;
        CFL_LAUNCH(K_PAIR_BWD, (cfl_pair_bwd_kernel<2, 1>), dim3(cfl_cdiv(N, C::BM) * cfl_cdiv(D, C::BN), 1, 2), dim3(256),
                   C::LDS_BYTES, stream, I, T, coef, w.it, w.tt, N, D, vecN, w.rowsum, w.colsum, gout_dev, dI, dT, x3mode);
    } else {
        using C = TileCfg<1, 1, true, true>;
        CFL_LAUNCH(K_PAIR_BWD, (cfl_pair_bwd_kernel<1, 1>), dim3(cfl_cdiv(N, C::BM) * cfl_cdiv(D, C::BN), 1, 2), dim3(256),
                   C::LDS_BYTES, stream, I, T, coef, w.it, w.tt, N, D, vecN, w.rowsum, w.colsum, gout_dev, dI, dT, x3mode);
    }
    return 0;
}

}  // extern "C"
