// common.h -- shared device/host helpers for libcreamfl_hip.so (gfx950 only).
//
// The central piece is tile_gemm(): an LDS-staged, register-prefetched fp32 tile GEMM on
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 64 FLOP/clk/SIMD).  A workgroup is 4 waves
// arranged 2x2; each wave owns TM x TN MFMA tiles of 32x32, so the workgroup tile is
// (64*TM) x (64*TN).  K advances in steps of 32 through a double-buffered LDS stage with ONE
// barrier per step; the next step's global loads are issued before the MFMA block and
// written to the other LDS buffer after it.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "creamfl_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CFL_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

// ---- kernel registry (ids index cfl_kernel_name / cfl_prof_query) --------------------------
enum CflKernel {
    K_PAIR_PREP = 0, K_PAIR_FWD, K_PAIR_FINAL, K_PAIR_BWD,
    K_BANK_FWD, K_LSE_FINAL, K_BANK_LOSS, K_BANK_BWD, K_BANK_BWD_REDUCE,
    K_INTRA, K_CONW_COMBINE,
    K_PIE_SCORES, K_PIE_POOL, K_PIE_BWD_DS, K_PIE_BWD_DX, K_PIE_BWD_DH, K_PIE_BWD_DW2,
    K_PIE_EPI_FWD, K_PIE_EPI_BWD, K_PIE_EPI_BWD_LN, K_L2NORM_FWD, K_L2NORM_BWD,
    K_RANK_POSMAX, K_RANK_COUNT,
    K_GRADNORM, K_ADAMP_PASS1, K_ADAMP_DECIDE, K_ADAMP_PASS3,
    K_BN_STATS, K_BN_FINAL, K_BN_APPLY, K_BN_BWD_REDUCE, K_BN_BWD_FINAL, K_BN_BWD_APPLY,
    K_GEMM_PROBE, K_KD_MSE, K_SUP_GLUE, K_GEMM_BF16, K_BERT_DALN, K_BERT_GELU, K_ATTN_SMALL, K_MAXPOOL, K_TRANSPOSE,
    K_PIE_FWD_FUSED, K_PIE_BWD_FUSED,
    K_BN_POOL_FWD, K_BN_POOL_BWD_REDUCE, K_BN_POOL_BWD_APPLY,
    K_BANK_IMAGE, K_BANK_STREAM,
    K_BN_BWD_APPLY_WG, K_BN_WGRAD_REDUCE,
    K_GRU_FWD, K_GRU_BWD, K_GRU_CELL0,
    K_CONV3_WGRAD, K_CONV3_WGRAD_REDUCE, K_CONV1_WGRAD, K_CONV1_WGRAD_REDUCE,
    K_CONV3_X3, K_CONV3_X3_WGRAD, K_CONV3_X3_WGRAD_REDUCE, K_CONV3_X3_WIMAGE,
    K_PAIR_BWD_REDUCE,
    K_NUM
};

// Per-kernel profiler (runtime.hip).  A profiled launch goes through hipExtLaunchKernelGGL with a start and a stop event: the
// events take the dispatch's own begin / end timestamps, so no marker packets are queued around the kernel (hipEventRecord
// before and after cost the stream a few microseconds per launch -- with ~100 profiled launches per step the bench's timed
// region was 0.8 ms longer than the same steps unprofiled) and the durations are the kernel's, without a floor.
bool cfl_prof_begin(int id, hipEvent_t* e0, hipEvent_t* e1);
void cfl_prof_end(int id, hipEvent_t e0, hipEvent_t e1);

#define CFL_LAUNCH(id, kern, grid, block, shmem, stream, ...)                                                  \
    do {                                                                                                       \
        hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                                             \
        if (cfl_prof_begin((id), &pe0_, &pe1_)) {                                                              \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, pe0_, pe1_, 0, __VA_ARGS__);               \
            cfl_prof_end((id), pe0_, pe1_);                                                                    \
        } else {                                                                                               \
            hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);                                 \
        }                                                                                                      \
        CFL_CHECK(hipGetLastError());                                                                          \
    } while (0)

// Dynamic LDS above 64 KiB must be opted into per kernel (once).
#define CFL_SET_LDS(kern, bytes)                                                                   \
    do {                                                                                           \
        static bool done_ = false;                                                                 \
        if (!done_) {                                                                              \
            CFL_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                     \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            done_ = true;                                                                          \
        }                                                                                          \
    } while (0)

static inline int cfl_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t cfl_align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- device helpers ---------------------------------------------------------------------------
// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, each with a private
// 4 MiB L2).  This bijective remap gives every XCD a contiguous range of logical ids so that
// neighbouring tiles (which share operand panels) hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, i = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// Logical (XCD-contiguous) tile id -> (tile row, tile column) in super-tiles of 64 tiles (h x w, w = 8 when it
// divides the tile-column count): the ~64 workgroups resident on one XCD at a time then touch h + w operand panels
// per K-step instead of 2 + 32 (row-major order), i.e. far fewer L2 misses.  Falls back to row-major when the grid
// does not divide.  Speed only.
__device__ __forceinline__ void tile_swizzle(int v, int ntr, int ntc, int& ti, int& tj) {
    const int sw = (ntc % 8 == 0) ? 8 : (ntc % 4 == 0) ? 4 : (ntc % 2 == 0) ? 2 : 1;
    const int sh = 64 / sw;
    if (ntr % sh != 0) { ti = v / ntc; tj = v % ntc; return; }
    const int g = v >> 6, w = v & 63;
    const int gpr = ntc / sw;                       // super-tiles per row of super-tiles
    ti = (g / gpr) * sh + w / sw;
    tj = (g % gpr) * sw + w % sw;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Block-wide sum for 256-thread blocks; `red` is >= 4 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float softplusf(float x) {      // log(1 + e^x), stable
    return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) {
    return x >= 0.f ? 1.f / (1.f + expf(-x)) : expf(x) / (1.f + expf(x));
}
// merge two (max, sum-of-exp) pairs
__device__ __forceinline__ void lse_merge(float& m, float& l, float m2, float l2) {
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; l = 0.f; return; }
    l = l * expf(m - mn) + l2 * expf(m2 - mn);
    m = mn;
}

// ---- tile GEMM ----------------------------------------------------------------------------------
// Operand view. KCONTIG: element(r, k) = p[r*ld + k]   (K is the fast axis in memory)
//               else   : element(r, k) = p[k*ld + r]   (r is the fast axis in memory)
struct Opnd {
    const float* p;
    long long ld;
    int rows;      // extent of the r axis (rows of the output for A, columns for B)
    int kdim;      // extent of the k axis
    int vec;       // 1 if 16-byte vector access is legal (p 16B-aligned and ld % 4 == 0)
};

struct XfIdentity {
    __device__ __forceinline__ float operator()(float v, int /*r*/, int /*k*/) const { return v; }
};

template <int TM, int TN, bool A_KC, bool B_KC>
struct TileCfg {
    static constexpr int BM = 64 * TM, BN = 64 * TN, KC = 32;
    static constexpr int A_LD = A_KC ? (KC + 4) : BM;
    static constexpr int B_LD = B_KC ? (KC + 4) : BN;
    static constexpr int A_ELEMS = (A_KC ? BM : KC) * A_LD;
    static constexpr int B_ELEMS = (B_KC ? BN : KC) * B_LD;
    static constexpr int STAGE = A_ELEMS + B_ELEMS;          // floats per LDS stage
    static constexpr int LDS_FLOATS = 2 * STAGE;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

// global -> registers for one operand tile of R rows x 32 k (R/32 float4 per thread), in two phases so that the
// loads stay in flight across the MFMA block:
//   PHASE 0 (issue): branch-free loads from clamped in-tensor addresses -- nothing consumes the values yet;
//   PHASE 1 (finish, called after the MFMA block, right before the LDS write): staging transform `xf` and
//            select-to-zero of out-of-range elements.
// Fast path needs o.vec (base 16-byte aligned, ld % 4 == 0, contiguous extent % 4 == 0 => a float4 is entirely in or
// out of the tensor).  Slow path (odd shapes): per-element bounds checks, everything done in PHASE 0.
template <bool KCONTIG, int R, int PHASE, class Xf>
__device__ __forceinline__ void g2r(const Opnd& o, int r0, int k0, f32x4 (&reg)[R / 32], Xf xf) {
    const int t = threadIdx.x;
    if (KCONTIG) {
        const int kq = (t & 7) * 4;
        const int k = k0 + kq;
        if (o.vec) {
            const bool kok = k < o.kdim;
            const int kc = kok ? k : 0;
#pragma unroll
            for (int p = 0; p < R / 32; ++p) {
                const int r = r0 + p * 32 + (t >> 3);
                const int rc = r < o.rows ? r : o.rows - 1;
                if (PHASE == 0) {
                    reg[p] = *reinterpret_cast<const f32x4*>(o.p + (long long)rc * o.ld + kc);
                } else {
                    const bool ok = kok && (r < o.rows);
                    f32x4 v = reg[p];
                    v[0] = ok ? xf(v[0], rc, kc) : 0.f; v[1] = ok ? xf(v[1], rc, kc + 1) : 0.f;
                    v[2] = ok ? xf(v[2], rc, kc + 2) : 0.f; v[3] = ok ? xf(v[3], rc, kc + 3) : 0.f;
                    reg[p] = v;
                }
            }
        } else if (PHASE == 0) {
#pragma unroll
            for (int p = 0; p < R / 32; ++p) {
                const int r = r0 + p * 32 + (t >> 3);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < o.rows) {
                    const float* src = o.p + (long long)r * o.ld + k;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < o.kdim) v[e] = xf(src[e], r, k + e);
                }
                reg[p] = v;
            }
        }
    } else {
        constexpr int LPR = R / 4;               // lanes per k-row
        constexpr int KPP = 256 / LPR;           // k-rows per pass
        const int rq = (t % LPR) * 4;
        const int r = r0 + rq;
        if (o.vec) {
            const bool rok = r < o.rows;
            const int rc = rok ? r : 0;
#pragma unroll
            for (int p = 0; p < R / 32; ++p) {
                const int k = k0 + p * KPP + t / LPR;
                const int kc = k < o.kdim ? k : o.kdim - 1;
                if (PHASE == 0) {
                    reg[p] = *reinterpret_cast<const f32x4*>(o.p + (long long)kc * o.ld + rc);
                } else {
                    const bool ok = rok && (k < o.kdim);
                    f32x4 v = reg[p];
                    v[0] = ok ? xf(v[0], rc, kc) : 0.f; v[1] = ok ? xf(v[1], rc + 1, kc) : 0.f;
                    v[2] = ok ? xf(v[2], rc + 2, kc) : 0.f; v[3] = ok ? xf(v[3], rc + 3, kc) : 0.f;
                    reg[p] = v;
                }
            }
        } else if (PHASE == 0) {
#pragma unroll
            for (int p = 0; p < R / 32; ++p) {
                const int k = k0 + p * KPP + t / LPR;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < o.kdim) {
                    const float* src = o.p + (long long)k * o.ld + r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (r + e < o.rows) v[e] = xf(src[e], r + e, k);
                }
                reg[p] = v;
            }
        }
    }
}

// registers -> LDS stage
template <bool KCONTIG, int R, int LD>
__device__ __forceinline__ void r2s(float* s, const f32x4 (&reg)[R / 32]) {
    const int t = threadIdx.x;
    if (KCONTIG) {
#pragma unroll
        for (int p = 0; p < R / 32; ++p)
            *reinterpret_cast<f32x4*>(&s[(p * 32 + (t >> 3)) * LD + (t & 7) * 4]) = reg[p];
    } else {
        constexpr int LPR = R / 4;
        constexpr int KPP = 256 / LPR;
#pragma unroll
        for (int p = 0; p < R / 32; ++p)
            *reinterpret_cast<f32x4*>(&s[(p * KPP + t / LPR) * LD + (t % LPR) * 4]) = reg[p];
    }
}

// MFMA fragment for k-chunk kk (8 wide): lane (i = lane&31, h = lane>>5) receives the four k values
// kk*8 + 4h + {0,1,2,3}; MFMA t of the chunk consumes element t (A and B use the same map, so
// the k-permutation cancels in the contraction).
template <bool KCONTIG, int LD>
__device__ __forceinline__ f32x4 frag(const float* s, int tile_r0, int kk, int lane) {
    if (KCONTIG) {
        return *reinterpret_cast<const f32x4*>(&s[(tile_r0 + (lane & 31)) * LD + kk * 8 + 4 * (lane >> 5)]);
    } else {
        const float* b = &s[(kk * 8 + 4 * (lane >> 5)) * LD + tile_r0 + (lane & 31)];
        f32x4 v = {b[0], b[LD], b[2 * LD], b[3 * LD]};
        return v;
    }
}

// One K-step of MFMAs from an LDS stage: acc[m][n] += A_tile(32 k) * B_tile(32 k)^T  (64 MFMAs for TM = TN = 2).
// Fragments are double-buffered in registers: the ds_reads of k-chunk kk+1 are issued before the 16 MFMAs of
// chunk kk, so LDS latency hides behind the matrix pipe; the MFMA block runs at raised wave priority so that the
// two waves sharing a SIMD take turns on the pipe instead of advancing in lock-step into the same stall.
template <int TM, int TN, bool A_KC, bool B_KC>
__device__ __forceinline__ void tile_compute(const float* sa, const float* sb, f32x16 (&acc)[TM][TN], int lane, int wr, int wc) {
    using C = TileCfg<TM, TN, A_KC, B_KC>;
    f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) fa[0][m] = frag<A_KC, C::A_LD>(sa, (wr * TM + m) * 32, 0, lane);
#pragma unroll
    for (int n = 0; n < TN; ++n) fb[0][n] = frag<B_KC, C::B_LD>(sb, (wc * TN + n) * 32, 0, lane);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
#pragma unroll
            for (int m = 0; m < TM; ++m) fa[(kk + 1) & 1][m] = frag<A_KC, C::A_LD>(sa, (wr * TM + m) * 32, kk + 1, lane);
#pragma unroll
            for (int n = 0; n < TN; ++n) fb[(kk + 1) & 1][n] = frag<B_KC, C::B_LD>(sb, (wc * TN + n) * 32, kk + 1, lane);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][m][t], fb[kk & 1][n][t], acc[m][n], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
}

struct TileDesc { int row0, col0, kbeg, kend; };

// A SEQUENCE of output tiles through one software pipeline.  tile_fn(i) -> TileDesc for i in [0, ntiles);
// epi_fn(i, acc) consumes the finished accumulators of tile i.  The global loads of K-step s+1 -- which may
// already belong to tile i+1 -- are issued before the MFMA block of step s and committed to the other LDS
// buffer after it, so a new tile starts computing right after the previous tile's epilogue (no per-tile
// load-latency bubble).  All 256 threads must call; `lds` needs TileCfg::LDS_FLOATS floats, 16-byte aligned.
// epi_fn must not touch `lds` (the next tile's first stage is already resident); the LDS is free after return.
template <int TM, int TN, bool A_KC, bool B_KC, class XfA, class TileFn, class EpiFn>
__device__ __forceinline__ void tile_gemm_seq(const Opnd& A, const Opnd& B, int ntiles, TileFn tile_fn, float* lds,
                                              XfA xfa, EpiFn epi_fn) {
    using C = TileCfg<TM, TN, A_KC, B_KC>;
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    if (ntiles <= 0) return;
    f32x4 ra[C::BM / 32], rb[C::BN / 32];
    TileDesc cur = tile_fn(0);
    g2r<A_KC, C::BM, 0>(A, cur.row0, cur.kbeg, ra, xfa);
    g2r<B_KC, C::BN, 0>(B, cur.col0, cur.kbeg, rb, XfIdentity());
    g2r<A_KC, C::BM, 1>(A, cur.row0, cur.kbeg, ra, xfa);
    g2r<B_KC, C::BN, 1>(B, cur.col0, cur.kbeg, rb, XfIdentity());
    r2s<A_KC, C::BM, C::A_LD>(lds, ra);
    r2s<B_KC, C::BN, C::B_LD>(lds + C::A_ELEMS, rb);
    __syncthreads();
    int buf = 0;
    f32x16 acc[TM][TN];
    for (int i = 0; i < ntiles; ++i) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        const int nk = (cur.kend - cur.kbeg + C::KC - 1) / C::KC;
        const bool has_next_tile = (i + 1 < ntiles);
        TileDesc nxt = cur;
        if (has_next_tile) nxt = tile_fn(i + 1);
        for (int kt = 0; kt < nk; ++kt) {
            const float* sa = lds + buf * C::STAGE;
            const bool in_tile = (kt + 1 < nk);
            const bool more = in_tile || has_next_tile;
            const int r0 = in_tile ? cur.row0 : nxt.row0;
            const int c0 = in_tile ? cur.col0 : nxt.col0;
            const int k0 = in_tile ? cur.kbeg + (kt + 1) * C::KC : nxt.kbeg;
            if (more) {
                g2r<A_KC, C::BM, 0>(A, r0, k0, ra, xfa);
                g2r<B_KC, C::BN, 0>(B, c0, k0, rb, XfIdentity());
            }
            tile_compute<TM, TN, A_KC, B_KC>(sa, sa + C::A_ELEMS, acc, lane, wr, wc);
            if (more) {
                g2r<A_KC, C::BM, 1>(A, r0, k0, ra, xfa);
                g2r<B_KC, C::BN, 1>(B, c0, k0, rb, XfIdentity());
                float* da = lds + (buf ^ 1) * C::STAGE;
                r2s<A_KC, C::BM, C::A_LD>(da, ra);
                r2s<B_KC, C::BN, C::B_LD>(da + C::A_ELEMS, rb);
            }
            __syncthreads();
            buf ^= 1;
        }
        epi_fn(i, acc);
        cur = nxt;
    }
    __syncthreads();
}

// Single tile: acc[m][n] = A[row0.., k] * B[col0.., k] over k in [kbeg, kend).  (kend - kbeg) must be a multiple
// of 32 unless kend == kdim.  On return the LDS stage is free again.
template <int TM, int TN, bool A_KC, bool B_KC, class XfA>
__device__ __forceinline__ void tile_gemm(const Opnd& A, const Opnd& B, int row0, int col0,
                                          int kbeg, int kend, float* lds,
                                          f32x16 (&acc)[TM][TN], XfA xfa) {
    using C = TileCfg<TM, TN, A_KC, B_KC>;
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wr = wid >> 1, wc = wid & 1;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    f32x4 ra[C::BM / 32], rb[C::BN / 32];
    const int nk = (kend - kbeg + C::KC - 1) / C::KC;
    if (nk <= 0) return;
    g2r<A_KC, C::BM, 0>(A, row0, kbeg, ra, xfa);
    g2r<B_KC, C::BN, 0>(B, col0, kbeg, rb, XfIdentity());
    g2r<A_KC, C::BM, 1>(A, row0, kbeg, ra, xfa);
    g2r<B_KC, C::BN, 1>(B, col0, kbeg, rb, XfIdentity());
    r2s<A_KC, C::BM, C::A_LD>(lds, ra);
    r2s<B_KC, C::BN, C::B_LD>(lds + C::A_ELEMS, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const float* sa = lds + (kt & 1) * C::STAGE;
        const bool more = (kt + 1 < nk);
        if (more) {
            g2r<A_KC, C::BM, 0>(A, row0, kbeg + (kt + 1) * C::KC, ra, xfa);
            g2r<B_KC, C::BN, 0>(B, col0, kbeg + (kt + 1) * C::KC, rb, XfIdentity());
        }
        tile_compute<TM, TN, A_KC, B_KC>(sa, sa + C::A_ELEMS, acc, lane, wr, wc);
        if (more) {
            g2r<A_KC, C::BM, 1>(A, row0, kbeg + (kt + 1) * C::KC, ra, xfa);
            g2r<B_KC, C::BN, 1>(B, col0, kbeg + (kt + 1) * C::KC, rb, XfIdentity());
            float* da = lds + ((kt + 1) & 1) * C::STAGE;
            r2s<A_KC, C::BM, C::A_LD>(da, ra);
            r2s<B_KC, C::BN, C::B_LD>(da + C::A_ELEMS, rb);
        }
        __syncthreads();
    }
}

// ---- direct-to-LDS variant (both operands K-contiguous, K % 32 == 0) ---------------------------------------------
// `global_load_lds_dwordx4` copies 64 x 16 B per wave instruction straight into LDS (no VGPR staging, no ds_write).
// The LDS image of a wave instruction is lane-linear (base + lane*16), so the stage is unpadded [rows][32 floats]
// and the bank-conflict fix is an XOR swizzle applied on the SOURCE side: slot c of row r holds k-chunk
// c ^ ((r>>1)&7); fragment reads apply the same involution.  With a 128-byte row, (r&1, (r>>1)&7) is distinct for
// the 16 rows of every ds_read_b128 lane group => conflict-free.  Out-of-range rows are clamped (duplicated) and must
// be masked by the epilogue (every fwd epilogue here masks by index).
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

template <int R>
__device__ __forceinline__ void glds_stage(const Opnd& o, int r0, int k0, float* stage) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PER_WAVE = R / 32;                    // 8-row groups per wave
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int grp = w * PER_WAVE + j;
        const int r = grp * 8 + (lane >> 3);
        const int q = (lane & 7) ^ ((r >> 1) & 7);
        int rg = r0 + r;
        rg = rg < o.rows ? rg : o.rows - 1;
        const float* src = o.p + (long long)rg * o.ld + k0 + q * 4;
        __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(stage + grp * 256), 16, 0, 0);
    }
}
__device__ __forceinline__ f32x4 frag_swz(const float* s, int tile_r0, int kk, int lane) {
    const int r = tile_r0 + (lane & 31);
    const int slot = (kk * 2 + (lane >> 5)) ^ ((r >> 1) & 7);
    return *reinterpret_cast<const f32x4*>(&s[r * 32 + slot * 4]);
}
template <int TM, int TN>
__device__ __forceinline__ void tile_compute_swz(const float* sa, const float* sb, f32x16 (&acc)[TM][TN], int lane, int wr, int wc) {
    f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) fa[0][m] = frag_swz(sa, (wr * TM + m) * 32, 0, lane);
#pragma unroll
    for (int n = 0; n < TN; ++n) fb[0][n] = frag_swz(sb, (wc * TN + n) * 32, 0, lane);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
#pragma unroll
            for (int m = 0; m < TM; ++m) fa[(kk + 1) & 1][m] = frag_swz(sa, (wr * TM + m) * 32, kk + 1, lane);
#pragma unroll
            for (int n = 0; n < TN; ++n) fb[(kk + 1) & 1][n] = frag_swz(sb, (wc * TN + n) * 32, kk + 1, lane);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][m][t], fb[kk & 1][n][t], acc[m][n], 0, 0, 0);
    }
}
// true when the direct-to-LDS path applies to both operands
__device__ __forceinline__ bool glds_ok(const Opnd& A, const Opnd& B) {
    return A.vec && B.vec && (A.kdim % 32) == 0;
}
// Sequence of tiles, direct-to-LDS staging.  Same contract as tile_gemm_seq (KC/KC only, (kend-kbeg) % 32 == 0).
// ComputeFn(sa, sb, acc, lane, wr, wc) consumes one stage: tile_compute_swz (fp32 operands) or x3::compute on PRE-SPLIT
// operand images (tile_x3.h: a 128-byte row block = [32 x bf16 hi | 32 x bf16 lo], the same bytes as 32 floats).
template <int TM, int TN, class TileFn, class EpiFn, class ComputeFn>
__device__ __forceinline__ void tile_gemm_seq_glds_with(const Opnd& A, const Opnd& B, int ntiles, TileFn tile_fn, float* lds,
                                                        EpiFn epi_fn, ComputeFn compute_fn) {
    constexpr int BM = 64 * TM, BN = 64 * TN, STAGE = (BM + BN) * 32;
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    if (ntiles <= 0) return;
    TileDesc cur = tile_fn(0);
    glds_stage<BM>(A, cur.row0, cur.kbeg, lds);
    glds_stage<BN>(B, cur.col0, cur.kbeg, lds + BM * 32);
    __syncthreads();
    int buf = 0;
    f32x16 acc[TM][TN];
    for (int i = 0; i < ntiles; ++i) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        const int nk = (cur.kend - cur.kbeg) / 32;
        const bool has_next_tile = (i + 1 < ntiles);
        TileDesc nxt = cur;
        if (has_next_tile) nxt = tile_fn(i + 1);
        for (int kt = 0; kt < nk; ++kt) {
            const float* sa = lds + buf * STAGE;
            const bool in_tile = (kt + 1 < nk);
            if (in_tile || has_next_tile) {
                float* da = lds + (buf ^ 1) * STAGE;
                glds_stage<BM>(A, in_tile ? cur.row0 : nxt.row0, in_tile ? cur.kbeg + (kt + 1) * 32 : nxt.kbeg, da);
                glds_stage<BN>(B, in_tile ? cur.col0 : nxt.col0, in_tile ? cur.kbeg + (kt + 1) * 32 : nxt.kbeg, da + BM * 32);
            }
            compute_fn(sa, sa + BM * 32, acc, lane, wr, wc);
            __syncthreads();                 // waits for the LDS-DMA of the next stage (vmcnt) and for all readers
            buf ^= 1;
        }
        epi_fn(i, acc);
        cur = nxt;
    }
    __syncthreads();
}
template <int TM, int TN, class TileFn, class EpiFn>
__device__ __forceinline__ void tile_gemm_seq_glds(const Opnd& A, const Opnd& B, int ntiles, TileFn tile_fn, float* lds,
                                                   EpiFn epi_fn) {
    tile_gemm_seq_glds_with<TM, TN>(A, B, ntiles, tile_fn, lds, epi_fn,
                                    [](const float* sa, const float* sb, f32x16 (&acc)[TM][TN], int lane, int wr, int wc) {
                                        tile_compute_swz<TM, TN>(sa, sb, acc, lane, wr, wc);
                                    });
}

// Single tile, direct-to-LDS staging: acc = A[row0.., k] * B[col0.., k]^T over k in [kbeg, kend), (kend-kbeg) % 32 == 0.
template <int TM, int TN>
__device__ __forceinline__ void tile_gemm_glds(const Opnd& A, const Opnd& B, int row0, int col0, int kbeg, int kend,
                                               float* lds, f32x16 (&acc)[TM][TN]) {
    constexpr int BM = 64 * TM, BN = 64 * TN, STAGE = (BM + BN) * 32;
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wr = wid >> 1, wc = wid & 1;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int nk = (kend - kbeg) / 32;
    if (nk <= 0) return;
    glds_stage<BM>(A, row0, kbeg, lds);
    glds_stage<BN>(B, col0, kbeg, lds + BM * 32);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const float* sa = lds + (kt & 1) * STAGE;
        if (kt + 1 < nk) {
            float* da = lds + ((kt + 1) & 1) * STAGE;
            glds_stage<BM>(A, row0, kbeg + (kt + 1) * 32, da);
            glds_stage<BN>(B, col0, kbeg + (kt + 1) * 32, da + BM * 32);
        }
        tile_compute_swz<TM, TN>(sa, sa + BM * 32, acc, lane, wr, wc);
        __syncthreads();
    }
}

// Output coordinates of accumulator element (m, n, r) held by `lane` of wave (wr, wc), relative
// to the workgroup tile origin (32x32 C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
template <int TM>
__device__ __forceinline__ int acc_row(int wr, int m, int r, int lane) {
    return (wr * TM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
template <int TN>
__device__ __forceinline__ int acc_col(int wc, int n, int lane) {
    return (wc * TN + n) * 32 + (lane & 31);
}

static inline int cfl_vec_ok(const void* p, long long ld) {
    return ((((uintptr_t)p) & 15) == 0 && (ld % 4) == 0) ? 1 : 0;
}
// Opnd::vec additionally needs the extent along the contiguous axis to be a multiple of 4 (a float4 is then
// entirely inside or entirely outside the tensor): kdim for K-contiguous operands, rows for K-strided ones.
static inline int cfl_opnd_vec(const void* p, long long ld, int contiguous_extent) {
    return (cfl_vec_ok(p, ld) && (contiguous_extent % 4) == 0) ? 1 : 0;
}
