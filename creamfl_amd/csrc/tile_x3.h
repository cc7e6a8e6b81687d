// tile_x3.h -- the tile GEMM of common.h on the bf16 matrix pipe at fp32-class accuracy ("3 x bf16 split").
//
// Every fp32 operand element is split x = hi + lo (two bf16: 16 mantissa bits) WHILE IT IS STAGED into LDS, and each
// product runs as three v_mfma_f32_32x32x16_bf16 (hi.hi + lo.hi + hi.lo, fp32 accumulation; the dropped lo.lo term is
// 2^-16 relative).  A 32-deep K step of a 128 x 128 tile costs 24 MFMAs x 32 cycles = 768 cycles instead of the 64
// v_mfma_f32_32x32x2_f32 x 64 cycles = 4096 of the exact-fp32 path: dot products of unit-norm rows come out to ~1e-6
// absolute, far inside the 1e-4 parity budget of the north star, at 1/5.3 of the matrix-pipe time.
//
// The LDS stage has the geometry of the direct-to-LDS path of common.h -- rows of 128 bytes, 16-byte slots XOR-swizzled
// by ((row >> 1) & 7) -- but a row holds [32 x bf16 hi | 32 x bf16 lo] instead of 32 floats: slot 2 kk + h is the hi
// fragment of MFMA kk for lane half h, slot 4 + 2 kk + h its lo fragment; every ds_read_b128 lane group is conflict-free
// for the same reason as there.  Staging goes through registers (the split needs the VALU): loads are issued before the
// MFMA block of the current stage and converted + written to the other buffer after it (one barrier per stage).
//   K-contiguous operands: a thread owns 4 consecutive k of one row per pass (two 8-byte stores, hi and lo region);
//   K-strided operands (element(r, k) = p[k * ld + r]): a thread owns an 8 (k) x 4 (r) block -- eight row-coalesced float4
//   loads -- transposes it in registers and writes, per r, the 8 k values as one 16-byte hi and one 16-byte lo fragment.
#pragma once
#include "common.h"

namespace x3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int soff(int r, int slot) { return r * 128 + ((slot ^ ((r >> 1) & 7)) << 4); }

__device__ __forceinline__ void split1(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// ---- K-contiguous operand: R rows x 32 k per stage -------------------------------------------------------------------------
template <int R, int PHASE, class Xf>
__device__ __forceinline__ void stage_kc(const Opnd& o, int r0, int k0, f32x4 (&reg)[R / 32], char* st, Xf xf) {
    const int t = threadIdx.x;
    const int kq = t & 7, k = k0 + 4 * kq;
#pragma unroll
    for (int p = 0; p < R / 32; ++p) {
        const int rl = p * 32 + (t >> 3);
        const int r = r0 + rl;
        if (PHASE == 0) {
            if (o.vec) {
                const int rc = r < o.rows ? r : o.rows - 1;
                const int kc = k < o.kdim ? k : 0;
                reg[p] = *reinterpret_cast<const f32x4*>(o.p + (long long)rc * o.ld + kc);
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < o.rows) {
                    const float* src = o.p + (long long)r * o.ld + k;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < o.kdim) v[e] = src[e];
                }
                reg[p] = v;
            }
        } else {
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = r < o.rows && k + e < o.kdim;
                const float x = ok ? xf(reg[p][e], r, k + e) : 0.f;
                __bf16 h, l;
                split1(x, h, l);
                hi[e] = h; lo[e] = l;
            }
            *reinterpret_cast<bf16x4*>(st + soff(rl, kq >> 1) + (kq & 1) * 8) = hi;
            *reinterpret_cast<bf16x4*>(st + soff(rl, 4 + (kq >> 1)) + (kq & 1) * 8) = lo;
        }
    }
}

// ---- K-strided operand: element(r, k) = p[k * ld + r]; R rows x 32 k per stage; thread = (r quad, k octet) -----------------
template <int R, int PHASE, class Xf>
__device__ __forceinline__ void stage_ks(const Opnd& o, int r0, int k0, f32x4 (&reg)[8], char* st, Xf xf) {
    constexpr int NQ = R / 4;                         // r quads per stage; NQ * 4 work items (R <= 256 -> <= 256 items)
    const int t = threadIdx.x;
    if (t >= NQ * 4) return;
    const int rq = t % NQ, ko = t / NQ;
    const int r = r0 + 4 * rq, kb = k0 + 8 * ko;
    if (PHASE == 0) {
        if (o.vec) {
            const int rc = r < o.rows ? r : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = kb + j;
                const int kc = k < o.kdim ? k : o.kdim - 1;
                reg[j] = *reinterpret_cast<const f32x4*>(o.p + (long long)kc * o.ld + rc);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                const int k = kb + j;
                if (k < o.kdim) {
                    const float* src = o.p + (long long)k * o.ld + r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (r + e < o.rows) v[e] = src[e];
                }
                reg[j] = v;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = r + c < o.rows && kb + j < o.kdim;
                const float x = ok ? xf(reg[j][c], r + c, kb + j) : 0.f;
                __bf16 h, l;
                split1(x, h, l);
                hi[j] = h; lo[j] = l;
            }
            *reinterpret_cast<bf16x8*>(st + soff(4 * rq + c, ko)) = hi;
            *reinterpret_cast<bf16x8*>(st + soff(4 * rq + c, 4 + ko)) = lo;
        }
    }
}

template <bool KC, int R>
struct StageRegs { f32x4 r[KC ? R / 32 : 8]; };

template <bool KC, int R, int PHASE, class Xf>
__device__ __forceinline__ void stage(const Opnd& o, int r0, int k0, StageRegs<KC, R>& regs, char* st, Xf xf) {
    if constexpr (KC) stage_kc<R, PHASE>(o, r0, k0, regs.r, st, xf);
    else stage_ks<R, PHASE>(o, r0, k0, regs.r, st, xf);
}

// ---- one 32-deep K step from an LDS stage: acc[m][n] += A_tile * B_tile^T (3 MFMAs per product, term-major) -----------------
template <int TM, int TN>
__device__ __forceinline__ void compute(const char* sa, const char* sb, f32x16 (&acc)[TM][TN], int lane, int wr, int wc) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int r = (wr * TM + m) * 32 + i;
            ah[m] = *reinterpret_cast<const bf16x8*>(sa + soff(r, 2 * kk + h));
            al[m] = *reinterpret_cast<const bf16x8*>(sa + soff(r, 4 + 2 * kk + h));
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int r = (wc * TN + n) * 32 + i;
            bh[n] = *reinterpret_cast<const bf16x8*>(sb + soff(r, 2 * kk + h));
            bl[n] = *reinterpret_cast<const bf16x8*>(sb + soff(r, 4 + 2 * kk + h));
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
}

template <int TM, int TN>
struct Cfg {
    static constexpr int BM = 64 * TM, BN = 64 * TN;
    static constexpr int STAGE_BYTES = (BM + BN) * 128;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;        // <= TileCfg<TM, TN, *, *>::LDS_BYTES of common.h
};

// A SEQUENCE of output tiles through one software pipeline.  DIST = prefetch distance in stages: 1 = the loads of stage
// s + 1 are issued before the MFMA block of stage s and converted + written after it (one register set); 2 = two register
// sets alternate, loads of stage s + 2 issued before the MFMA block of stage s and committed after that of stage s + 1 (a
// bf16 stage is ~770 matrix-pipe cycles, shorter than an L2 round trip).  Measured: 2 helps the plain GEMM (4096^3: 200 ->
// 210-232 TFLOP/s); routed through this generic cursor pipeline the kernels with register-hungry epilogues or K-strided
// operands got slower (pair-loss forward 166 -> 241 us, bank backward 136 -> 233 us), so those use the plain loops below and
// only the GEMM probe uses this one.  tile_fn(i) -> TileDesc for i in [0, ntiles); epi_fn(i, acc) consumes the finished accumulators
// of tile i (must not touch `lds`).  Stages = every (tile, 32-deep K step) in order; the stage stream runs across tile
// boundaries.  All 256 threads must call; the LDS (Cfg::LDS_BYTES) is free on return.
struct Cursor { int i, kt, nk; TileDesc d; };

template <int TM, int TN, bool A_KC, bool B_KC, int DIST, class XfA, class TileFn, class EpiFn>
__device__ __forceinline__ void gemm_pipeline(const Opnd& A, const Opnd& B, int ntiles, TileFn tile_fn, float* lds_f, XfA xfa,
                                              EpiFn epi_fn) {
    using C = Cfg<TM, TN>;
    char* lds = reinterpret_cast<char*>(lds_f);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    if (ntiles <= 0) return;
    auto open_tile = [&](int i) {
        Cursor c;
        c.i = i; c.kt = 0;
        if (i < ntiles) { c.d = tile_fn(i); c.nk = (c.d.kend - c.d.kbeg + 31) / 32; if (c.nk < 1) c.nk = 1; }
        else { c.d = TileDesc{0, 0, 0, 0}; c.nk = 1; }
        return c;
    };
    auto next = [&](const Cursor& c) {
        if (c.kt + 1 < c.nk) { Cursor n = c; n.kt = c.kt + 1; return n; }
        return open_tile(c.i + 1);
    };
    auto valid = [&](const Cursor& c) { return c.i < ntiles; };
    StageRegs<A_KC, C::BM> ra0, ra1;
    StageRegs<B_KC, C::BN> rb0, rb1;
    auto issue = [&](const Cursor& c, StageRegs<A_KC, C::BM>& ra, StageRegs<B_KC, C::BN>& rb) {
        const int k0 = c.d.kbeg + c.kt * 32;
        stage<A_KC, C::BM, 0>(A, c.d.row0, k0, ra, lds, xfa);
        stage<B_KC, C::BN, 0>(B, c.d.col0, k0, rb, lds, XfIdentity());
    };
    auto commit = [&](const Cursor& c, StageRegs<A_KC, C::BM>& ra, StageRegs<B_KC, C::BN>& rb, char* dst) {
        const int k0 = c.d.kbeg + c.kt * 32;
        stage<A_KC, C::BM, 1>(A, c.d.row0, k0, ra, dst, xfa);
        stage<B_KC, C::BN, 1>(B, c.d.col0, k0, rb, dst + C::BM * 128, XfIdentity());
    };
    Cursor cur = open_tile(0), c1 = next(cur), c2 = next(c1);
    issue(cur, ra0, rb0);
    commit(cur, ra0, rb0, lds);
    if (DIST == 2 && valid(c1)) issue(c1, ra1, rb1);
    __syncthreads();
    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    int buf = 0;
    // one pipeline step: stage `cur` is in LDS buffer `buf`.  DIST 2: `r*_in` holds stage c1 (in flight since the previous
    // step), `r*_out` is free for stage c2.  DIST 1: one register set, stage c1 is issued and committed within the step.
    auto step = [&](StageRegs<A_KC, C::BM>& ra_in, StageRegs<B_KC, C::BN>& rb_in, StageRegs<A_KC, C::BM>& ra_out,
                    StageRegs<B_KC, C::BN>& rb_out) {
        if (DIST == 2) { if (valid(c2)) issue(c2, ra_out, rb_out); }
        else if (valid(c1)) issue(c1, ra_in, rb_in);
        const char* sa = lds + buf * C::STAGE_BYTES;
        compute<TM, TN>(sa, sa + C::BM * 128, acc, lane, wr, wc);
        if (valid(c1)) commit(c1, ra_in, rb_in, lds + (buf ^ 1) * C::STAGE_BYTES);
        if (cur.kt == cur.nk - 1) {
            epi_fn(cur.i, acc);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        }
        __syncthreads();
        buf ^= 1;
        cur = c1; c1 = c2; c2 = next(c2);
    };
    if (DIST == 2) {
        while (valid(cur)) {
            step(ra1, rb1, ra0, rb0);
            if (!valid(cur)) break;
            step(ra0, rb0, ra1, rb1);
        }
    } else {
        while (valid(cur)) step(ra0, rb0, ra0, rb0);
    }
}

// Single tile through the two-register-set pipeline (used by the GEMM probe).
template <int TM, int TN, bool A_KC, bool B_KC, int DIST, class XfA>
__device__ __forceinline__ void tile_gemm_pipelined(const Opnd& A, const Opnd& B, int row0, int col0, int kbeg, int kend,
                                                    float* lds_f, f32x16 (&acc)[TM][TN], XfA xfa) {
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    if (kend <= kbeg) return;
    gemm_pipeline<TM, TN, A_KC, B_KC, DIST>(A, B, 1, [&](int) { return TileDesc{row0, col0, kbeg, kend}; }, lds_f, xfa,
                                            [&](int, const f32x16 (&a)[TM][TN]) {
#pragma unroll
                                                for (int m = 0; m < TM; ++m)
#pragma unroll
                                                    for (int n = 0; n < TN; ++n) acc[m][n] = a[m][n];
                                            });
}

// Single tile: acc = A[row0.., k] * B[col0.., k]^T over k in [kbeg, kend).  Same contract as tile_gemm of common.h
// ((kend - kbeg) % 32 == 0 unless kend == kdim; all 256 threads call; the LDS stage is free on return).  Plain loop,
// prefetch distance 1: the form the kernels with register-hungry epilogues / K-strided operands run fastest with.
template <int TM, int TN, bool A_KC, bool B_KC, class XfA>
__device__ __forceinline__ void tile_gemm(const Opnd& A, const Opnd& B, int row0, int col0, int kbeg, int kend, float* lds_f,
                                          f32x16 (&acc)[TM][TN], XfA xfa) {
    using C = Cfg<TM, TN>;
    char* lds = reinterpret_cast<char*>(lds_f);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int nk = (kend - kbeg + 31) / 32;
    if (nk <= 0) return;
    StageRegs<A_KC, C::BM> ra;
    StageRegs<B_KC, C::BN> rb;
    stage<A_KC, C::BM, 0>(A, row0, kbeg, ra, lds, xfa);
    stage<B_KC, C::BN, 0>(B, col0, kbeg, rb, lds, XfIdentity());
    stage<A_KC, C::BM, 1>(A, row0, kbeg, ra, lds, xfa);
    stage<B_KC, C::BN, 1>(B, col0, kbeg, rb, lds + C::BM * 128, XfIdentity());
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const char* sa = lds + (kt & 1) * C::STAGE_BYTES;
        char* da = lds + ((kt + 1) & 1) * C::STAGE_BYTES;
        const bool more = kt + 1 < nk;
        const int k1 = kbeg + (kt + 1) * 32;
        if (more) {
            stage<A_KC, C::BM, 0>(A, row0, k1, ra, da, xfa);
            stage<B_KC, C::BN, 0>(B, col0, k1, rb, da, XfIdentity());
        }
        compute<TM, TN>(sa, sa + C::BM * 128, acc, lane, wr, wc);
        if (more) {
            stage<A_KC, C::BM, 1>(A, row0, k1, ra, da, xfa);
            stage<B_KC, C::BN, 1>(B, col0, k1, rb, da + C::BM * 128, XfIdentity());
        }
        __syncthreads();
    }
}

// A SEQUENCE of output tiles through one software pipeline (both operands K-contiguous); contract of tile_gemm_seq.
template <int TM, int TN, class TileFn, class EpiFn>
__device__ __forceinline__ void tile_gemm_seq(const Opnd& A, const Opnd& B, int ntiles, TileFn tile_fn, float* lds_f, EpiFn epi_fn) {
    using C = Cfg<TM, TN>;
    char* lds = reinterpret_cast<char*>(lds_f);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    if (ntiles <= 0) return;
    StageRegs<true, C::BM> ra;
    StageRegs<true, C::BN> rb;
    TileDesc cur = tile_fn(0);
    stage<true, C::BM, 0>(A, cur.row0, cur.kbeg, ra, lds, XfIdentity());
    stage<true, C::BN, 0>(B, cur.col0, cur.kbeg, rb, lds, XfIdentity());
    stage<true, C::BM, 1>(A, cur.row0, cur.kbeg, ra, lds, XfIdentity());
    stage<true, C::BN, 1>(B, cur.col0, cur.kbeg, rb, lds + C::BM * 128, XfIdentity());
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
        const int nk = (cur.kend - cur.kbeg + 31) / 32;
        const bool has_next = i + 1 < ntiles;
        TileDesc nxt = cur;
        if (has_next) nxt = tile_fn(i + 1);
        for (int kt = 0; kt < nk; ++kt) {
            const char* sa = lds + buf * C::STAGE_BYTES;
            char* da = lds + (buf ^ 1) * C::STAGE_BYTES;
            const bool in_tile = kt + 1 < nk;
            const bool more = in_tile || has_next;
            const int r0 = in_tile ? cur.row0 : nxt.row0, c0 = in_tile ? cur.col0 : nxt.col0;
            const int k0 = in_tile ? cur.kbeg + (kt + 1) * 32 : nxt.kbeg;
            if (more) {
                stage<true, C::BM, 0>(A, r0, k0, ra, da, XfIdentity());
                stage<true, C::BN, 0>(B, c0, k0, rb, da, XfIdentity());
            }
            compute<TM, TN>(sa, sa + C::BM * 128, acc, lane, wr, wc);
            if (more) {
                stage<true, C::BM, 1>(A, r0, k0, ra, da, XfIdentity());
                stage<true, C::BN, 1>(B, c0, k0, rb, da + C::BM * 128, XfIdentity());
            }
            __syncthreads();
            buf ^= 1;
        }
        epi_fn(i, acc);
        cur = nxt;
    }
    __syncthreads();
}

// ---- pre-split operand images ------------------------------------------------------------------------------------------------
// image[r][kb] (128 bytes) = { bf16 hi of src[r][32 kb .. 32 kb + 31], then their bf16 lo }, kb < Kp / 32, Kp = K rounded up to
// 32 (zero padded).  Byte-for-byte the size of the fp32 matrix with K padded: an Opnd {image as float*, ld = Kp, rows, kdim =
// Kp, vec = 1} drives the unchanged direct-to-LDS staging of common.h, and x3::compute reads the stage.
static inline int image_kp(int K) { return (K + 31) / 32 * 32; }

static __global__ __launch_bounds__(256) void split_image_kernel(const float* __restrict__ src, long long ld, int rows, int K, int Kp,
                                                          float* __restrict__ image) {
    const long long quads_per_row = Kp / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads_per_row * rows) return;
    const int r = (int)(i / quads_per_row), k = (int)(i % quads_per_row) * 4;
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = (k + e < K) ? src[(long long)r * ld + k + e] : 0.f;
        __bf16 h, l;
        split1(x, h, l);
        hi[e] = h; lo[e] = l;
    }
    char* blk = reinterpret_cast<char*>(image) + ((long long)r * Kp + (k & ~31)) * 4;
    *reinterpret_cast<bf16x4*>(blk + (k & 31) * 2) = hi;
    *reinterpret_cast<bf16x4*>(blk + 64 + (k & 31) * 2) = lo;
}

}  // namespace x3

// Process-wide precision switch of the dense kernels that exist in both forms (csrc/runtime.hip): 0 = 3 x bf16 split
// (default), 1 = exact fp32 MFMA.
extern "C" int cfl_get_exact_gemm(void);
