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
    K_NUM
};

struct CflProfScope {
    int id; hipStream_t s; hipEvent_t e0, e1; bool on;
    CflProfScope(int id, hipStream_t s);
    ~CflProfScope();
};

#define CFL_LAUNCH(id, kern, grid, block, shmem, stream, ...)                       \
    do {                                                                            \
        { CflProfScope ps_((id), (stream));                                         \
          hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__); }      \
        CFL_CHECK(hipGetLastError());                                               \
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

// global -> registers for one operand tile of R rows x 32 k (R/32 float4 per thread)
template <bool KCONTIG, int R, class Xf>
__device__ __forceinline__ void g2r(const Opnd& o, int r0, int k0, f32x4 (&reg)[R / 32], Xf xf) {
    const int t = threadIdx.x;
    if (KCONTIG) {
        const int kq = (t & 7) * 4;
        const int k = k0 + kq;
#pragma unroll
        for (int p = 0; p < R / 32; ++p) {
            const int r = r0 + p * 32 + (t >> 3);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < o.rows) {
                const float* src = o.p + (long long)r * o.ld + k;
                if (o.vec && k + 3 < o.kdim) {
                    v = *reinterpret_cast<const f32x4*>(src);
                    v[0] = xf(v[0], r, k); v[1] = xf(v[1], r, k + 1);
                    v[2] = xf(v[2], r, k + 2); v[3] = xf(v[3], r, k + 3);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < o.kdim) v[e] = xf(src[e], r, k + e);
                }
            }
            reg[p] = v;
        }
    } else {
        constexpr int LPR = R / 4;               // lanes per k-row
        constexpr int KPP = 256 / LPR;           // k-rows per pass
        const int rq = (t % LPR) * 4;
        const int r = r0 + rq;
#pragma unroll
        for (int p = 0; p < R / 32; ++p) {
            const int k = k0 + p * KPP + t / LPR;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < o.kdim) {
                const float* src = o.p + (long long)k * o.ld + r;
                if (o.vec && r + 3 < o.rows) {
                    v = *reinterpret_cast<const f32x4*>(src);
                    v[0] = xf(v[0], r, k); v[1] = xf(v[1], r + 1, k);
                    v[2] = xf(v[2], r + 2, k); v[3] = xf(v[3], r + 3, k);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (r + e < o.rows) v[e] = xf(src[e], r + e, k);
                }
            }
            reg[p] = v;
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

// acc[m][n] (+)= A[row0.., k] * B[col0.., k] over k in [kbeg, kend).  All 256 threads must call.
// `lds` needs TileCfg::LDS_FLOATS floats, 16-byte aligned.  On return the LDS stage is free again
// (a trailing barrier has been executed).
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
    g2r<A_KC, C::BM>(A, row0, kbeg, ra, xfa);
    g2r<B_KC, C::BN>(B, col0, kbeg, rb, XfIdentity());
    r2s<A_KC, C::BM, C::A_LD>(lds, ra);
    r2s<B_KC, C::BN, C::B_LD>(lds + C::A_ELEMS, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const float* sa = lds + (kt & 1) * C::STAGE;
        const float* sb = sa + C::A_ELEMS;
        const bool more = (kt + 1 < nk);
        if (more) {
            g2r<A_KC, C::BM>(A, row0, kbeg + (kt + 1) * C::KC, ra, xfa);
            g2r<B_KC, C::BN>(B, col0, kbeg + (kt + 1) * C::KC, rb, XfIdentity());
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) fa[m] = frag<A_KC, C::A_LD>(sa, (wr * TM + m) * 32, kk, lane);
#pragma unroll
            for (int n = 0; n < TN; ++n) fb[n] = frag<B_KC, C::B_LD>(sb, (wc * TN + n) * 32, kk, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][t], fb[n][t], acc[m][n], 0, 0, 0);
        }
        if (more) {
            float* da = lds + ((kt + 1) & 1) * C::STAGE;
            r2s<A_KC, C::BM, C::A_LD>(da, ra);
            r2s<B_KC, C::BN, C::B_LD>(da + C::A_ELEMS, rb);
        }
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
