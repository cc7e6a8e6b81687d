// bank_attn.hip -- rows A3 + A4 in ONE pass over the frozen global bank (D <= 256).
//
// Reference: src/algorithms/ClientTrainer.py:388,398-419 (inter CE over the 50 000-row bank + intra / MOON term and
//            their combination), src/algorithms/MMClientTrainer.py:173-206.
//
// The inter-modal term is  loss = mean_b [ LSE_m(f_b.G_m / tau) - f_b.G_idx[b] / tau ]  with gradient
//   dF_b = (1 / (tau B)) (softmax_b . G - G_idx[b]),
// i.e. exactly an attention forward with Q = F, K = V = G: the log-sum-exp AND the gradient come out of one stream
// over G.  Round 1 ran an exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate: compute-bound at 28 %
// of that peak), wrote the [M, B] logits, and streamed G a second time for the backward GEMM.  Here:
//   * every fp32 operand is split x = hi + lo (two bf16, 16 mantissa bits) while it is staged into LDS, and each
//     product runs as 3 v_mfma_f32_32x32x16_bf16 (hi.hi + lo.hi + hi.lo; the dropped lo.lo term is 2^-16 relative):
//     logits to ~1e-6 absolute on unit-norm features, 5.3x fewer matrix-pipe cycles than the fp32 MFMA;
//   * a workgroup = 4 waves = 128 feature rows (32 per wave, held as MFMA B-fragments in registers for the whole
//     kernel); it walks its share of 32-row bank chunks; per chunk and wave: S^T[32 g, 32 f] = G F^T (swapped operands:
//     a lane owns ONE feature row, so running max / sum are lane-local), online soft-max with deferred rescaling,
//     O^T[D, 32 f] += G^T P^T.  P^T goes from the S accumulators straight into the B-operand registers of the second
//     MFMA (the contraction index is permuted consistently on both operands: no cross-lane traffic);
//   * the chunk is staged through registers (fp32 -> bf16 hi/lo) into TWO LDS images: [g][d] for the logits and a
//     transposed [d][g] one (in the permuted g order) for G^T, both XOR-swizzled so that every ds_read_b128 /
//     ds_write_b64 lane group is bank-conflict free (checked exhaustively offline); double-buffered, one barrier per chunk,
//     the next chunk's global loads in flight during the MFMA block;
//   * partial (max, sum, O) of the S splits are merged by a combine kernel that also emits the gradient, and ONE epilogue
//     kernel does the exact-fp32 positive dot, the intra / MOON term (A4), the means and the loss combination.
// HBM traffic: G once (M D 4 bytes) + the split partials; no [B, M] tensor exists.
//
// Workspace layout (floats): part_m[RG][S][128] part_l[RG][S][128] part_o[RG][S][DP][128] rowbuf[2][Bp]; `sync` is a
// caller-owned int that must be 0 before the first call and is left 0 (last-block election of the epilogue kernel).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BR = 128;        // feature rows per workgroup
constexpr int GC = 32;         // bank rows per chunk
constexpr float RESCALE_THR = 5.f;   // log2 units: probabilities are kept <= 2^5 relative to the running max

struct AttnPlan { int DT, DP, RG, S, Bp; };
static AttnPlan attn_plan(int B, int M, int D) {
    AttnPlan p;
    p.DT = D <= 64 ? 2 : (D <= 128 ? 4 : 8);
    p.DP = 32 * p.DT;
    p.RG = cfl_cdiv(B, BR);
    p.Bp = p.RG * BR;
    const int nch = cfl_cdiv(M, GC);
    int s = cfl_cdiv(256, p.RG);          // one workgroup per CU (the kernel owns the whole register file)
    if (s > nch) s = nch;
    if (s < 1) s = 1;
    p.S = s;
    return p;
}

__device__ __forceinline__ void split4(const f32x4 v, bool ok, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = ok ? v[e] : 0.f;
        const __bf16 h = (__bf16)x;
        hi[e] = h;
        lo[e] = (__bf16)(x - (float)h);
    }
}

// ---- LDS images (byte offsets) ----------------------------------------------------------------------------------------
// row image: [plane][g 0..31][DP bf16], 16-byte slot s of row g stored at slot s ^ swz_row(g)
template <int DP>
__device__ __forceinline__ int swz_row(int g) { return (DP / 8 >= 16) ? (g & 15) : ((g >> 1) & 7); }
template <int DP>
__device__ __forceinline__ int row_off(int plane, int g, int slot) {
    return plane * (GC * DP * 2) + g * (DP * 2) + ((slot ^ swz_row<DP>(g)) << 4);
}
// transposed image: [d >> 2][(d & 3) ^ cx][slot 0..7][8 bf16]; slot = 4 * plane + 2 * t + h holds, for MFMA t and lane half h,
// the 8 bank rows g = 16 t + 4 h + (j & 3) + 8 (j >> 2), j = 0..7 -- the rows whose probabilities that lane half already owns
__device__ __forceinline__ int t_off(int d, int slot) {
    const int dq = d >> 2, c = d & 3;
    const int cx = (dq >> 2) & 1;
    const int sig = ((dq & 7) ^ (((c >> 1) | (((dq >> 3) & 1) << 1)) << 1)) & 7;
    return dq * 512 + ((c ^ cx) << 7) + ((slot ^ sig) << 4);
}

template <int DT, bool GRAD>
struct Smem {
    static constexpr int DP = 32 * DT;
    static constexpr int ROW_BYTES = 2 * GC * DP * 2;
    static constexpr int T_BYTES = GRAD ? DP * 128 : 0;
    static constexpr int BUF = ROW_BYTES + T_BYTES;
    static constexpr int TOTAL = 2 * BUF;
};

// global -> registers: thread (q, rsub) fetches the 8 (g) x 4 (d) block of chunk rows 8 rsub .. + 7, columns 4 q .. + 3
template <int DP>
__device__ __forceinline__ void chunk_load(const float* __restrict__ G, int M, int D, int g0, f32x4 (&r)[8]) {
    constexpr int QPR = DP / 4;
    const int q = threadIdx.x % QPR, rsub = threadIdx.x / QPR;
    if (rsub >= 4) return;
    int d0 = 4 * q;
    d0 = d0 < D ? d0 : D - 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int g = g0 + 8 * rsub + i;
        g = g < M ? g : M - 1;
        r[i] = *reinterpret_cast<const f32x4*>(G + (long long)g * D + d0);
    }
}

template <int DP, bool GRAD>
__device__ __forceinline__ void chunk_store(char* buf, int M, int D, int g0, const f32x4 (&r)[8]) {
    constexpr int QPR = DP / 4;
    const int q = threadIdx.x % QPR, rsub = threadIdx.x / QPR;
    if (rsub >= 4) return;
    const bool colok = 4 * q < D;
    bf16x4 hi[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split4(r[i], colok && (g0 + 8 * rsub + i < M), hi[i], lo[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gl = 8 * rsub + i;
        *reinterpret_cast<bf16x4*>(buf + row_off<DP>(0, gl, q >> 1) + (q & 1) * 8) = hi[i];
        *reinterpret_cast<bf16x4*>(buf + row_off<DP>(1, gl, q >> 1) + (q & 1) * 8) = lo[i];
    }
    if (GRAD) {
        char* tb = buf + 2 * GC * DP * 2;
        const int t = rsub >> 1, piece = 8 * (rsub & 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = 4 * q + c;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vh[e] = hi[4 * h + e][c]; vl[e] = lo[4 * h + e][c]; }
                *reinterpret_cast<bf16x4*>(tb + t_off(d, 2 * t + h) + piece) = vh;
                *reinterpret_cast<bf16x4*>(tb + t_off(d, 4 + 2 * t + h) + piece) = vl;
            }
        }
    }
}

// grid (S, RG).  F [B, D], G [M, D] fp32 row-major, D % 4 == 0, D <= DP.
template <int DT, bool GRAD>
__global__ __launch_bounds__(256, 1) void cfl_bank_attn_kernel(const float* __restrict__ F, const float* __restrict__ G, int B, int M,
                                                            int D, float sc2, float* __restrict__ part_m,
                                                            float* __restrict__ part_l, float* __restrict__ part_o) {
    constexpr int DP = 32 * DT, KS = DP / 16;
    using SM = Smem<DT, GRAD>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int S = gridDim.x, x = blockIdx.x, rg = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 31, h = lane >> 5;
    const int nch = (M + GC - 1) / GC;
    const int ntiles = x < nch ? (nch - x + S - 1) / S : 0;

    // this wave's 32 feature rows as B-operand fragments: lane (f, h) holds k = 16 kk + 8 h + j
    bf16x8 fh[KS], fl[KS];
    {
        const int fr = rg * BR + 32 * w + fi;
        const float* fp = F + (long long)(fr < B ? fr : B - 1) * D;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int k0 = 16 * kk + 8 * h;
            const bool ok0 = fr < B && k0 < D, ok1 = fr < B && k0 + 4 < D;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(fp + (k0 < D ? k0 : 0));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(fp + (k0 + 4 < D ? k0 + 4 : 0));
            bf16x4 h0, l0, h1, l1;
            split4(v0, ok0, h0, l0);
            split4(v1, ok1, h1, l1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { fh[kk][e] = h0[e]; fh[kk][4 + e] = h1[e]; fl[kk][e] = l0[e]; fl[kk][4 + e] = l1[e]; }
        }
    }

    f32x16 O[GRAD ? DT : 1];
#pragma unroll
    for (int i = 0; i < (GRAD ? DT : 1); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[i][r] = 0.f;
    float run_m = -INFINITY, run_l = 0.f;

    f32x4 stage[8];
    if (ntiles > 0) {
        chunk_load<DP>(G, M, D, x * GC, stage);
        chunk_store<DP, GRAD>(lds, M, D, x * GC, stage);
    }
    __syncthreads();
    int buf = 0;
    for (int i = 0; i < ntiles; ++i) {
        const int g0 = (x + i * S) * GC;
        const bool more = i + 1 < ntiles;
        const int g0n = (x + (i + 1) * S) * GC;
        if (more) chunk_load<DP>(G, M, D, g0n, stage);
        const char* rb = lds + buf * SM::BUF;
        // ---- logits S^T[g, f] = sum_k G[g, k] F[f, k]: hi.hi on one accumulator, the two cross terms on another
        f32x16 sa, sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(rb + row_off<DP>(0, fi, 2 * kk + h));
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(rb + row_off<DP>(1, fi, 2 * kk + h));
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, fh[kk], sa, 0, 0, 0);
            sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, fh[kk], sb, 0, 0, 0);
            sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, fl[kk], sb, 0, 0, 0);
        }
        // ---- online log-sum-exp in the base-2 domain; element r of this lane is bank row g0 + (r&3) + 8 (r>>2) + 4 h
        // (everything in place in `sa`: the kernel lives at the edge of the register file)
        const bool full = g0 + GC <= M;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sa[r] = (sa[r] + sb[r]) * sc2;
            if (!full && g0 + (r & 3) + 8 * (r >> 2) + 4 * h >= M) sa[r] = -INFINITY;
            mx = fmaxf(mx, sa[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));               // the other half-wave holds the other 16 bank rows of f
        // Deferred rescaling: keep the old reference while no probability would exceed 2^THR.  The decision is taken
        // per wave (uniform branch); both half-waves of a feature row always agree on its reference.
        if (__any(mx > run_m + RESCALE_THR)) {
            const float mn = fmaxf(run_m, mx);
            const float alpha = __builtin_amdgcn_exp2f(run_m - mn);        // 0 for the first chunk (run_m = -inf)
            run_l *= alpha;
            if (GRAD) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
            }
            run_m = mn;
        }
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = __builtin_amdgcn_exp2f(sa[r] - run_m); ls += sa[r]; }
        run_l += ls;
        if (GRAD) {
            // ---- O^T[d, f] += sum_g G[g, d] P[f, g]: B operand of MFMA t = accumulator elements 8 t .. 8 t + 7 as they lie
            bf16x8 ph[2], pl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const __bf16 hh = (__bf16)sa[8 * t + j];
                    ph[t][j] = hh;
                    pl[t][j] = (__bf16)(sa[8 * t + j] - (float)hh);
                }
            const char* tb = rb + SM::ROW_BYTES;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int d = 32 * dt + fi;
                    const bf16x8 gh = *reinterpret_cast<const bf16x8*>(tb + t_off(d, 2 * t + h));
                    const bf16x8 gl = *reinterpret_cast<const bf16x8*>(tb + t_off(d, 4 + 2 * t + h));
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, ph[t], O[dt], 0, 0, 0);
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, ph[t], O[dt], 0, 0, 0);
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, pl[t], O[dt], 0, 0, 0);
                }
        }
        if (more) chunk_store<DP, GRAD>(lds + (buf ^ 1) * SM::BUF, M, D, g0n, stage);
        __syncthreads();
        buf ^= 1;
    }
    // ---- split partials: (max, sum) per feature row and O^T as [d][128 rows] (128-byte contiguous runs per store)
    run_l += __shfl_xor(run_l, 32, 64);
    const size_t slab = (size_t)rg * S + x;
    if (h == 0) {
        part_m[slab * BR + 32 * w + fi] = run_m;
        part_l[slab * BR + 32 * w + fi] = run_l;
    }
    if (GRAD) {
        float* po = part_o + slab * DP * BR + 32 * w + fi;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * h;
                po[(size_t)d * BR] = O[dt][r];
            }
    }
}

// Merge the S split partials of one row group: block = one output column d (blockIdx.x) x 128 rows; 32 float4 lanes x 8
// split groups, fixed summation order => deterministic.  Emits lse (block d == 0) and the UNIT gradient of the inter term
//   dF[f][d] = (inv_tau / Bdiv) (O[f][d] / L[f] - G[idx[f]][d]).
__global__ __launch_bounds__(256) void cfl_bank_attn_combine_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                                                 const float* __restrict__ part_o, int S, int DP,
                                                                 const float* __restrict__ G, const long long* __restrict__ idx,
                                                                 int B, int M, int D, float coef, float* __restrict__ lse2,
                                                                 float* __restrict__ dF) {
    __shared__ f32x4 red[8][32];
    __shared__ f32x4 red2[8][32];
    const int d = blockIdx.x, rg = blockIdx.y;
    const int f4 = (threadIdx.x & 31) * 4, xg = threadIdx.x >> 5;
    const float* pm = part_m + (size_t)rg * S * BR + f4;
    const float* pl = part_l + (size_t)rg * S * BR + f4;
    f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int x = xg; x < S; x += 8) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(pm + (size_t)x * BR);
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], m[e]);
    }
    red[xg][threadIdx.x & 31] = mx;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const f32x4 o = red[g][threadIdx.x & 31];
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], o[e]);
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, L = {0.f, 0.f, 0.f, 0.f};
    const bool want_o = (dF != nullptr) && d < D;
    const float* po = part_o + ((size_t)rg * S * DP + d) * BR + f4;
#pragma unroll 4
    for (int x = xg; x < S; x += 8) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(pm + (size_t)x * BR);
        const f32x4 l = *reinterpret_cast<const f32x4*>(pl + (size_t)x * BR);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (want_o) o = *reinterpret_cast<const f32x4*>(po + (size_t)x * DP * BR);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float wgt = __builtin_amdgcn_exp2f(m[e] - mx[e]);
            L[e] = fmaf(wgt, l[e], L[e]);
            acc[e] = fmaf(wgt, o[e], acc[e]);
        }
    }
    red[xg][threadIdx.x & 31] = acc;
    red2[xg][threadIdx.x & 31] = L;
    __syncthreads();
    if (xg == 0) {
        acc = red[0][threadIdx.x]; L = red2[0][threadIdx.x];
#pragma unroll
        for (int g = 1; g < 8; ++g) { acc += red[g][threadIdx.x]; L += red2[g][threadIdx.x]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = rg * BR + f4 + e;
            if (f >= B) continue;
            if (d == 0) lse2[f] = (mx[e] + log2f(L[e])) * 0.6931471805599453f;
            if (want_o) {
                const long long tgt = idx[f];
                const float gpos = (tgt >= 0 && tgt < M) ? G[tgt * D + d] : 0.f;
                dF[(long long)f * D + d] = coef * (acc[e] / L[e] - gpos);
            }
        }
    }
}

// Epilogue, one wave per feature row: exact-fp32 positive dot of the inter term, the intra / MOON term (A4) with its unit
// gradient, per-row losses; the LAST block to finish (agent-scope release / acquire, cdna guide G16) reduces the rows in
// fixed order and writes out[0..4] = {loss, loss_inter, loss_moon, coef_inter, coef_moon}: the combined loss of
// ClientTrainer.py:416-419 and the factors the backward applies to the two unit gradients.
//   mode bit 0: inter term present, bit 1: intra term present, bit 2: --loss_scale
__global__ __launch_bounds__(256) void cfl_contrast_epilogue_kernel(const float* __restrict__ F, const float* __restrict__ Go,
                                                                 const float* __restrict__ Gs, const float* __restrict__ Fo,
                                                                 const long long* __restrict__ idx, int B, int M, int D, int Bdiv,
                                                                 float inv_tau, float weight, int mode, const float* __restrict__ lse,
                                                                 float* __restrict__ pos_out, float* __restrict__ rowbuf, int Bp,
                                                                 float* __restrict__ dF_moon, float* __restrict__ out5,
                                                                 int* __restrict__ sync) {
    __shared__ float red[4];
    __shared__ int is_last;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b < B) {
        const float* f = F + (long long)b * D;
        const long long tgt = idx[b];
        const bool ok = tgt >= 0 && tgt < M;
        if (mode & 1) {
            float dot = 0.f;
            if (ok) {
                const float* g = Go + tgt * D;
                for (int k = lane; k < D; k += 64) dot = fmaf(f[k], g[k], dot);
            }
            dot = wave_sum(dot) * inv_tau;
            if (lane == 0) { if (pos_out) pos_out[b] = dot; rowbuf[b] = lse[b] - dot; }
        }
        if (mode & 2) {
            const float* g = Gs + (ok ? tgt : 0) * D;
            const float* o = Fo + (long long)b * D;
            float pos = 0.f, neg = 0.f;
            for (int k = lane; k < D; k += 64) { pos = fmaf(f[k], ok ? g[k] : 0.f, pos); neg = fmaf(f[k], o[k], neg); }
            pos = wave_sum(pos); neg = wave_sum(neg);
            const float z = (neg - pos) * inv_tau;
            if (lane == 0) rowbuf[Bp + b] = softplusf(z);
            if (dF_moon) {
                const float c = sigmoidf(z) * inv_tau / (float)Bdiv;
                for (int k = lane; k < D; k += 64) dF_moon[(long long)b * D + k] = c * (o[k] - (ok ? g[k] : 0.f));
            }
        }
    }
    // ---- last-block election
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == (int)gridDim.x - 1);
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!is_last) return;
    float si = 0.f, sm = 0.f;
    for (int r = threadIdx.x; r < B; r += 256) {
        if (mode & 1) si += __hip_atomic_load(rowbuf + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mode & 2) sm += __hip_atomic_load(rowbuf + Bp + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    si = block_sum_256(si, red);
    sm = block_sum_256(sm, red);
    if (threadIdx.x == 0) {
        const float li = si / (float)B, lm = sm / (float)Bdiv;
        float loss, ci = 0.f, cm = 0.f;
        if ((mode & 3) == 3) {
            if (mode & 4) {          // (loss_moon + loss_inter / (loss_inter / loss_moon).detach()) * w
                const float r = li / lm;
                loss = (lm + li / r) * weight; ci = weight / r; cm = weight;
            } else {
                loss = (lm + li) * weight; ci = weight; cm = weight;
            }
        } else if (mode & 1) { loss = li; ci = 1.f; }
        else { loss = lm; cm = 1.f; }
        out5[0] = loss; out5[1] = li; out5[2] = lm; out5[3] = ci; out5[4] = cm;
        __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// dF = gout * (coef[3] dF_inter + coef[4] dF_moon)
__global__ __launch_bounds__(256) void cfl_contrast_bwd_kernel(const float* __restrict__ dFi, const float* __restrict__ dFm,
                                                            const float* __restrict__ out5, const float* __restrict__ gout,
                                                            long long n, float* __restrict__ dF) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    if (dFi) v = fmaf(out5[3], dFi[i], v);
    if (dFm) v = fmaf(out5[4], dFm[i], v);
    dF[i] = v * gout[0];
}

struct AttnWs { float *part_m, *part_l, *part_o, *rowbuf; };
static AttnWs attn_ws(void* ws, const AttnPlan& p) {
    AttnWs w;
    float* q = (float*)ws;
    w.part_m = q; q += (size_t)p.RG * p.S * BR;
    w.part_l = q; q += (size_t)p.RG * p.S * BR;
    w.rowbuf = q; q += (size_t)2 * p.Bp;
    w.part_o = q;
    return w;
}

template <int DT, bool GRAD>
static int launch_attn(const float* F, const float* G, int B, int M, int D, float sc2, const AttnPlan& p, const AttnWs& w,
                       hipStream_t stream) {
    using SM = Smem<DT, GRAD>;
    CFL_SET_LDS((cfl_bank_attn_kernel<DT, GRAD>), SM::TOTAL);
    CFL_LAUNCH(K_BANK_FWD, (cfl_bank_attn_kernel<DT, GRAD>), dim3(p.S, p.RG), dim3(256), SM::TOTAL, stream, F, G, B, M, D, sc2,
               w.part_m, w.part_l, w.part_o);
    return 0;
}

}  // namespace

extern "C" {

int cfl_bank_attn_supported(int B, int M, int D) {
    return (B > 0 && M > 0 && D >= 4 && D <= 256 && D % 4 == 0) ? 1 : 0;
}

size_t cfl_bank_attn_ws_bytes(int B, int M, int D, int want_grad) {
    if (!cfl_bank_attn_supported(B, M, D)) return 256;
    const AttnPlan p = attn_plan(B, M, D);
    size_t n = (size_t)2 * p.RG * p.S * BR + (size_t)2 * p.Bp;
    if (want_grad) n += (size_t)p.RG * p.S * p.DP * BR;
    return cfl_align256(n * sizeof(float));
}

int cfl_client_contrast_fwd(const float* F, const float* G_other, const float* G_same, const long long* idx, const float* F_old,
                            int B, int M, int D, int B_div, float inv_tau, float weight, int mode, int want_grad,
                            float* out5, float* lse, float* pos, float* dF_inter, float* dF_moon, void* ws, int* sync,
                            void* stream_) {
    if (!F || !idx || !out5 || !ws || !sync || B <= 0 || M <= 0 || D <= 0 || !(inv_tau > 0.f) || !(mode & 3)) return CFL_EINVAL;
    if ((mode & 1) && (!G_other || !lse)) return CFL_EINVAL;
    if ((mode & 2) && (!G_same || !F_old || B_div <= 0)) return CFL_EINVAL;
    if (want_grad && (((mode & 1) && !dF_inter) || ((mode & 2) && !dF_moon))) return CFL_EINVAL;
    if (!cfl_bank_attn_supported(B, M, D)) return CFL_ELIMIT;
    if ((((uintptr_t)F | (uintptr_t)G_other) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const AttnPlan p = attn_plan(B, M, D);
    const AttnWs w = attn_ws(ws, p);
    if (mode & 1) {
        const float sc2 = inv_tau * 1.4426950408889634f;
        int rc;
        if (want_grad) {
            rc = p.DT == 2 ? launch_attn<2, true>(F, G_other, B, M, D, sc2, p, w, stream)
               : p.DT == 4 ? launch_attn<4, true>(F, G_other, B, M, D, sc2, p, w, stream)
                           : launch_attn<8, true>(F, G_other, B, M, D, sc2, p, w, stream);
        } else {
            rc = p.DT == 2 ? launch_attn<2, false>(F, G_other, B, M, D, sc2, p, w, stream)
               : p.DT == 4 ? launch_attn<4, false>(F, G_other, B, M, D, sc2, p, w, stream)
                           : launch_attn<8, false>(F, G_other, B, M, D, sc2, p, w, stream);
        }
        if (rc) return rc;
        CFL_LAUNCH(K_BANK_BWD_REDUCE, cfl_bank_attn_combine_kernel, dim3(want_grad ? p.DP : 1, p.RG), dim3(256), 0, stream,
                   w.part_m, w.part_l, w.part_o, p.S, p.DP, G_other, idx, B, M, D, inv_tau / (float)B, lse,
                   want_grad ? dF_inter : (float*)nullptr);
    }
    CFL_LAUNCH(K_LSE_FINAL, cfl_contrast_epilogue_kernel, dim3(cfl_cdiv(B, 4)), dim3(256), 0, stream, F, G_other, G_same, F_old, idx,
               B, M, D, (mode & 2) ? B_div : B, inv_tau, weight, mode, lse, pos, w.rowbuf, p.Bp,
               want_grad ? dF_moon : (float*)nullptr, out5, sync);
    return 0;
}

int cfl_client_contrast_bwd(const float* dF_inter, const float* dF_moon, const float* out5, const float* gout_dev, int B, int D,
                            float* dF, void* stream_) {
    if ((!dF_inter && !dF_moon) || !out5 || !gout_dev || !dF || B <= 0 || D <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const long long n = (long long)B * D;
    CFL_LAUNCH(K_BANK_BWD, cfl_contrast_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dF_inter, dF_moon, out5,
               gout_dev, n, dF);
    return 0;
}

}  // extern "C"
