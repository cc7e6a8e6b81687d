// bank_attn.hip -- rows A3 + A4 in ONE pass over the frozen global bank: the finish / backward launches, the C entry points,
// and (included below) the bank pass itself, csrc/bank_gsplit.h.
//
// Reference: src/algorithms/ClientTrainer.py:388,398-419 (inter CE over the 50 000-row bank + intra / MOON term and
//            their combination), src/algorithms/MMClientTrainer.py:173-206.
//
// The inter-modal term is  loss = mean_b [ LSE_m(f_b.G_m / tau) - f_b.G_idx[b] / tau ]  with gradient
//   dF_b = (1 / (tau B)) (softmax_b . G - G_idx[b]),
// i.e. exactly an attention forward with Q = F, K = V = G: the log-sum-exp AND the gradient come out of one stream
// over G (no [B, M] tensor, no second pass).  Every fp32 operand is split x = hi + lo (two bf16) and each product runs as
// 3 bf16 MFMAs (hi.hi + lo.hi + hi.lo; the dropped lo.lo term is 2^-16 relative): logits to ~1e-6 absolute on unit-norm
// features.  The bank pass (bank_gsplit.h, round 3) streams a PRE-SPLIT image of the bank; ONE finish kernel merges the
// (max, sum, O) partials of the S bank splits into lse + the unit gradient, does the exact-fp32 positive dot and the intra /
// MOON term (A4) per row, and -- last block to finish -- the means and the loss combination.
// History: round 2's pass over the fp32 bank (128-row groups, D <= 256, conversion while staging: cfl_bank_attn_kernel /
// cfl_client_contrast_fwd) was kept through round 3 as an A/B reference and removed in round 4 (docs/history/DESIGN_r1-r4.md section 4.3 has its
// measurements); the exact-fp32 two-pass kernels of csrc/bank.hip remain as the one reference path.
//
// Workspace layout (floats): part_m[RG][S][128] part_l[RG][S][128] rowbuf[2][Bp] part_o[Bp][S][DP]; `sync` is a
// caller-owned int that must be 0 before the first call and is left 0 (last-block election of the epilogue kernel).
#include <stdlib.h>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BR = 128;        // feature rows per workgroup
constexpr float RESCALE_THR = 5.f;   // log2 units: probabilities are kept <= 2^5 relative to the running max

struct AttnPlan { int DT, DP, RG, S, Bp; };

__device__ __forceinline__ void split4(const f32x4 v, bool ok, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = ok ? v[e] : 0.f;
        const __bf16 h = (__bf16)x;
        hi[e] = h;
        lo[e] = (__bf16)(x - (float)h);
    }
}

// v_mfma_f32_16x16x32_bf16: A lane (i = l & 15, kg = l >> 4) holds k = 8 kg + j; B likewise with n = l & 15;
// C / D: 4 registers, column n = l & 15, row 4 (l >> 4) + r.
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// 16-byte write-through store (sc1): the split partials are consumed by the NEXT kernel, possibly on another XCD; written
// through they leave the L2 as they are produced instead of as a dirty-line flush at the kernel boundary (cdna guide:
// "publish-large", 8.2 -> 3.0 us per 64 KB per workgroup).
__device__ __forceinline__ void store_wt_x4(float* p, const f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Finish kernel = split merge + per-row terms + losses, ONE launch.  Block = (feature row f, quarter of the columns):
//   * mode bit 0: merge the S split partials of row f (16 float4 lanes x 16 split groups, every load of a thread in flight at
//     once; measured at B = 128, S = 256: 128 blocks x 1 KB granules 35.7 us, 512 x 256 B 18.3 us, 1024 x 128 B 26.6 us;
//     fixed summation order => deterministic) -> lse[f] and the UNIT gradient of the inter term
//       dF_inter[f][d] = (inv_tau / B) (O[f][d] / L[f] - G_other[idx[f]][d]);
//   * the quarter-0 block of a row also does the row's exact-fp32 dots: the positive of the inter term (wave 0) and the intra /
//     MOON term (A4) with its unit gradient (wave 1), and stores the row's loss terms;
//   * the LAST block to finish (write-through stores + agent-scope ticket, cdna guide G16) reduces the rows in fixed order and writes
//     out5 = {loss, loss_inter, loss_moon, coef_inter, coef_moon}: the combined loss of ClientTrainer.py:416-419 and the
//     factors the backward applies to the two unit gradients.   mode bit 1: intra term present, bit 2: --loss_scale,
//     bit 3: add the li / lm out5 already holds (the other modality of a multi-modal client) before combining
//     bit 4 (round 6, "direct"): without --loss_scale the two coefficients are known on the host (w, w | 1 | 1), so when the
//     caller declares the loss the ROOT of the backward pass (upstream gradient = the 1 of loss.backward()) the FINAL gradient
//       dF[f][d] = ci (inv_tau / B) (O / L - G_other[idx]) + cm sigmoid(z) (inv_tau / Bdiv) (F_old - G_same[idx])
//     is written here, once, and the backward launch (2 reads + 1 write of [B, D] and ~5 us) does not exist.  Every column block
//     of a row then forms the row's intra dots itself (3 KB of L2-resident rows, 4 x redundant) instead of waiting for block 0.
template <int NJ>      // a thread merges the splits xg + 16 j, j < NJ (S <= 16 NJ)
__global__ __launch_bounds__(256) void cfl_contrast_finish_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                                               const float* __restrict__ part_o, int S, int DP,
                                                               const float* __restrict__ F, const float* __restrict__ Go,
                                                               const float* __restrict__ Gs, const float* __restrict__ Fo,
                                                               const long long* __restrict__ idx, int B, int M, int D, int Bdiv,
                                                               float inv_tau, float weight, int mode, float* __restrict__ lse2,
                                                               float* __restrict__ pos_out, float* __restrict__ rowbuf, int Bp,
                                                               float* __restrict__ dF, float* __restrict__ dF_moon,
                                                               float* __restrict__ out5, int* __restrict__ sync) {
    __shared__ float redm[4];
    __shared__ f32x4 red[16][16];
    __shared__ float redl[16];
    __shared__ float row_lse, row_pos, row_c;
    __shared__ int is_last;
    const int ncb = DP / 64;                        // blocks per row: 64 columns (16 lanes x 4) each
    const int fl = blockIdx.x / ncb, quarter = blockIdx.x % ncb, rg = blockIdx.y;
    const int f = rg * BR + fl;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const bool live = f < B;
    // Row terms (wave 0 of column block 0: the exact positive dot of the inter term; wave 1: the intra / MOON term): their operands
    // -- the feature row, the bank row(s) of the row's public-set index, the old feature row -- and the merge's own loads share
    // one stretch of memory latency: issue order = index, split partials (independent of the index), then everything the index
    // addresses (the row operands; the 16 bytes of G_other[idx] / G_same[idx] / F_old a gradient-writing thread needs).  They used
    // to run after the merge's two barriers as loops of one load per trip: ~half of this kernel's time at D = 256.
    constexpr int NKP = 12;                                  // D <= 768 = 12 x 64 lanes
    float pf[NKP], pg[NKP], po[NKP];
    const bool direct = (mode & 16) != 0;
    const bool merge = live && (mode & 1);
    const bool rowwork = live && quarter == 0;
    const bool w_inter = rowwork && (mode & 1) && wv == 0;
    const bool w_intra = live && (mode & 2) && wv == 1 && (quarter == 0 || (direct && (mode & 1)));
    const float coef = (mode & 3) == 3 ? weight : 1.f;       // direct: both coefficients, known on the host (no --loss_scale)
    const int c4 = (t & 15) * 4, xg = t >> 4;
    const int d4 = quarter * 64 + c4;
    const bool want_o = dF != nullptr;
    const bool gwriter = merge && want_o && xg == 0 && d4 < D;                // the 16 threads that store the row's gradient piece
    const long long tgt = (w_inter || w_intra || gwriter) ? idx[f] : -1;
    const bool ok = tgt >= 0 && tgt < M;
    // 16 float4 lanes (one quarter of the columns) x 16 split groups.  ONE memory round trip: a thread first issues every load it
    // will ever need (<= 16 splits: max, sum and its 16 bytes of O), then the block agrees on the reference.  The partials of one
    // feature row are contiguous ([row][split][DP]): the four blocks of a row sweep one region.
    float pmv[NJ], plv[NJ];
    f32x4 ov[NJ];
    if (merge) {
        const float* pm = part_m + (size_t)rg * S * BR + fl;
        const float* pl = part_l + (size_t)rg * S * BR + fl;
        const float* pob = part_o + (((size_t)rg * BR + fl) * S) * DP + d4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int x = xg + 16 * j;
            const int xc = x < S ? x : S - 1;                      // unconditional loads (clamped), masked below
            pmv[j] = pm[(size_t)xc * BR];
            plv[j] = pl[(size_t)xc * BR];
            ov[j] = want_o ? *reinterpret_cast<const f32x4*>(pob + (size_t)xc * DP) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (w_inter || w_intra) {
        const float* fr = F + (long long)f * D;
        const float* g = (w_inter ? Go : Gs) + (ok ? tgt : 0) * D;
        const float* o = w_intra ? Fo + (long long)f * D : fr;
#pragma unroll
        for (int j = 0; j < NKP; ++j) {
            const int k = lane + 64 * j;
            const int kc = k < D ? k : 0;                        // unconditional (clamped) loads, masked here
            pf[j] = k < D ? fr[kc] : 0.f;
            pg[j] = (k < D && ok) ? g[kc] : 0.f;
            po[j] = (k < D && w_intra) ? o[kc] : 0.f;
        }
    }
    f32x4 gpos = {0.f, 0.f, 0.f, 0.f}, gsame = gpos, fold = gpos;
    if (gwriter) {
        if (ok) gpos = *reinterpret_cast<const f32x4*>(Go + tgt * D + d4);
        if (direct && (mode & 2)) {
            if (ok) gsame = *reinterpret_cast<const f32x4*>(Gs + tgt * D + d4);
            fold = *reinterpret_cast<const f32x4*>(Fo + (long long)f * D + d4);
        }
    }
    // the intra / MOON term of the row (A4): in a direct block ahead of the merge's barriers, its coefficient goes through LDS
    if (w_intra) {
        float pos = 0.f, neg = 0.f;
#pragma unroll
        for (int j = 0; j < NKP; ++j) { pos = fmaf(pf[j], pg[j], pos); neg = fmaf(pf[j], po[j], neg); }
        pos = wave_sum(pos); neg = wave_sum(neg);
        const float z = (neg - pos) * inv_tau;
        const float c = sigmoidf(z) * inv_tau / (float)Bdiv;
        if (lane == 0) {
            row_c = c;
            if (quarter == 0) {
                __hip_atomic_store(rowbuf + Bp + f, softplusf(z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (dF_moon && quarter == 0) {                        // (absent in a direct call with an inter term: the merge below adds it)
            const float cc = direct ? c * coef : c;
#pragma unroll
            for (int j = 0; j < NKP; ++j) {
                const int k = lane + 64 * j;
                if (k < D) dF_moon[(long long)f * D + k] = cc * (po[j] - pg[j]);
            }
        }
    }
    if (merge) {
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { if (xg + 16 * j >= S) pmv[j] = -INFINITY; mx = fmaxf(mx, pmv[j]); }
        mx = wave_max(mx);
        if (lane == 0) redm[wv] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float L = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float wgt = __builtin_amdgcn_exp2f(pmv[j] - mx);          // 0 for the masked splits (pmv = -inf)
            L = fmaf(wgt, plv[j], L);
            acc += ov[j] * wgt;
        }
        red[xg][t & 15] = acc;
        if ((t & 15) == 0) redl[xg] = L;
        __syncthreads();
        if (xg == 0) {
            L = redl[0];
            acc = red[0][t];
#pragma unroll
            for (int g = 1; g < 16; ++g) { acc += red[g][t]; L += redl[g]; }
            if (quarter == 0 && t == 0) { row_lse = (mx + log2f(L)) * 0.6931471805599453f; lse2[f] = row_lse; }
            if (gwriter) {
                const float inv = 1.f / L;
                f32x4 v = (acc * inv - gpos) * (inv_tau / (float)B);
                if (direct) {
                    v *= coef;
                    if (mode & 2) v += (fold - gsame) * (row_c * coef);
                }
                *reinterpret_cast<f32x4*>(dF + (long long)f * D + d4) = v;
            }
        }
    }
    if (rowwork) {
        if (w_inter) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NKP; ++j) dot = fmaf(pf[j], pg[j], dot);
            dot = wave_sum(dot) * inv_tau;
            if (lane == 0) { row_pos = dot; if (pos_out) pos_out[f] = dot; }
        }
        __syncthreads();
        if ((mode & 1) && t == 0) {
            __hip_atomic_store(rowbuf + f, row_lse - row_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    // ---- last-block election.  The only data handed to the last block are the per-row loss terms, published above with
    // write-through (sc1) atomic stores that were drained (vmcnt(0)) before this barrier and read back below with sc1
    // atomic loads: the "sc1 stores and loads on both sides" form of the cdna guide (G16) -- no L2 write-back / invalidate
    // fences, which cost ~7 us here with the freshly written gradient dirty in every L2.
    // Only the blocks that publish row terms (column block 0 of every row) take a ticket: one counter serialises its
    // arrivals at ~12 ns each (cdna guide, "fanin"), and with every (row, column block) voting that was 512 tickets = 6 us
    // at D = 256 and 2048 = 25 us at D = 512 / B = 256 -- most of this kernel's time.
    if (quarter != 0) return;
    __syncthreads();
    if (t == 0) {
        const int tk = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (tk == (int)((gridDim.x / ncb) * gridDim.y) - 1);
    }
    __syncthreads();
    if (!is_last) return;
    float si = 0.f, sm = 0.f;
    for (int r = t; r < B; r += 256) {
        if (mode & 1) si += __hip_atomic_load(rowbuf + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mode & 2) sm += __hip_atomic_load(rowbuf + Bp + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    si = block_sum_256(si, redm);
    sm = block_sum_256(sm, redm);
    if (t == 0) {
        float li = si / (float)B, lm = sm / (float)Bdiv;
        // mode bit 3 (multi-modal client, MMClientTrainer.py:184-206): out5 still holds the OTHER modality's terms, written by
        // the previous launch on this stream -- loss_inter = loss_1_inter + loss_2_inter, loss_intra = the CE over the stacked
        // [2B, 2] logits = the two modalities' softplus sums over Bdiv = 2B; the combination below then runs on the totals and
        // ONE coefficient pair serves the backward of both modalities.
        if (mode & 8) { li += out5[1]; lm += out5[2]; }
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
        out5[0] = loss; out5[1] = li; out5[2] = lm; out5[3] = ci; out5[4] = cm; out5[5] = loss;
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

#include "bank_gsplit.h"

template <int DT, int NGG, int NWAVE, int NDS, int LA, bool GRAD>
static int launch_stream(const float* F, const void* img, int B, int M, int D, float sc2, const gs::GsPlan& p, const AttnWs& w,
                         hipStream_t stream) {
    using SM = gs::StSmem<DT, NGG, NWAVE, NDS>;
    CFL_SET_LDS((gs::cfl_bank_stream_kernel<DT, NGG, NWAVE, NDS, LA, GRAD>), SM::TOTAL);
    CFL_LAUNCH(K_BANK_STREAM, (gs::cfl_bank_stream_kernel<DT, NGG, NWAVE, NDS, LA, GRAD>), dim3(p.S * p.RG), dim3(64 * NWAVE), SM::TOTAL, stream,
               F, (const char*)img, B, M, D, sc2, p.S, p.RG, w.part_m, w.part_l, w.part_o);
    return 0;
}

// con_w (row A5) on the bank pass: out[i] = <V[row0 + i], G[row0 + i]> - log sum_j exp <V[row0 + i], G[j]>   (MMFL.py:304-307).
// The stream kernel leaves per split (max, sum) of 2^(s log2 e - max); one thread per row merges the S splits in fixed order and
// takes the positive as an exact fp32 dot product.
__global__ __launch_bounds__(256) void cfl_conw_finish_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l, int S,
                                                              const float* __restrict__ V, const float* __restrict__ G, int rows, int D,
                                                              float* __restrict__ out) {
    __shared__ float lse[64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int f0 = blockIdx.x * 64;                                        // 64 consecutive rows per block
    if (threadIdx.x < 64) {                                               // lane = row: the partials of a split are contiguous over rows
        const int f = f0 + lane;
        float v = 0.f;
        if (f < rows) {
            const size_t o = (size_t)(f >> 7) * S * BR + (f & (BR - 1));
            float mm = -INFINITY;
            for (int x = 0; x < S; ++x) mm = fmaxf(mm, part_m[o + (size_t)x * BR]);
            float L = 0.f;
            for (int x = 0; x < S; ++x) L = fmaf(__builtin_amdgcn_exp2f(part_m[o + (size_t)x * BR] - mm), part_l[o + (size_t)x * BR], L);
            v = (mm + __builtin_amdgcn_logf(L)) * 0.6931471805599453f;
        }
        lse[lane] = v;
    }
    __syncthreads();
    for (int i = wv; i < 64; i += 4) {                                    // one wave per row: the positive as an exact fp32 dot product
        const int f = f0 + i;
        if (f >= rows) break;
        const float* v = V + (size_t)f * D;
        const float* g = G + (size_t)f * D;
        float dot = 0.f;
        for (int d = 4 * lane; d < D; d += 256) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(v + d), b = *reinterpret_cast<const f32x4*>(g + d);
            dot += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) dot += __shfl_xor(dot, sft, 64);
        if (lane == 0) out[f] = dot - lse[i];
    }
}

// the finish launch both bank passes share; S = number of splits of the partials
static int launch_finish(const AttnWs& w, int S, int DP, int RGF, int Bp, const float* F, const float* G_other, const float* G_same,
                         const float* F_old, const long long* idx, int B, int M, int D, int B_div, float inv_tau, float weight,
                         int mode, int want_grad, float* out5, float* lse, float* pos, float* dF_inter, float* dF_moon, int* sync,
                         hipStream_t stream) {
    float* dfi = (want_grad && (mode & 1)) ? dF_inter : (float*)nullptr;
    float* dfm = (want_grad && (mode & 2)) ? dF_moon : (float*)nullptr;
    if (want_grad == 2) {                  // direct: ONE final gradient, in dF_inter's place (the merge adds the intra part itself)
        mode |= 16;
        dfm = (mode & 1) ? (float*)nullptr : dF_inter;
    }
    const int bd = (mode & 2) ? B_div : B;
#define CFL_FINISH(NJ)                                                                                                              \
    CFL_LAUNCH(K_LSE_FINAL, cfl_contrast_finish_kernel<NJ>, dim3((DP / 64) * BR, RGF), dim3(256), 0, stream, w.part_m, w.part_l, w.part_o, S, DP, F, \
               G_other, G_same, F_old, idx, B, M, D, bd, inv_tau, weight, mode, lse, pos, w.rowbuf, Bp, dfi, dfm, out5, sync)
    if (S <= 16) { CFL_FINISH(1); }
    else if (S <= 32) { CFL_FINISH(2); }
    else if (S <= 64) { CFL_FINISH(4); }
    else if (S <= 128) { CFL_FINISH(8); }
    else { CFL_FINISH(16); }
#undef CFL_FINISH
    return 0;
}

}  // namespace

extern "C" {

// ---- round 3: the same step on a pre-split bank image (bank_gsplit.h) ---------------------------------------------------
size_t cfl_bank_image_bytes(int M, int D) {
    if (M <= 0 || D <= 0) return 0;
    return cfl_align256(gs::img_bytes(M, D));
}

int cfl_bank_image_build(const float* G, int M, int D, void* image, void* stream_) {
    if (!G || !image || M <= 0 || D <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const int DP = gs::img_dp(D);
    const long long n = (long long)cfl_cdiv(M, gs::SG) * gs::SG * (DP / 8);
    CFL_LAUNCH(K_BANK_IMAGE, gs::cfl_bank_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, G, M, D, DP, (char*)image);
    return 0;
}

int cfl_bank_gsplit_supported(int B, int M, int D) {
    return (B > 0 && M > 0 && D >= 4 && D <= 768 && D % 4 == 0) ? 1 : 0;
}

size_t cfl_bank_gsplit_ws_bytes(int B, int M, int D, int want_grad) {
    if (!cfl_bank_gsplit_supported(B, M, D)) return 256;
    const gs::GsPlan p = gs::gs_plan(B, M, D);
    size_t n = (size_t)2 * p.RGF * p.S * BR + (size_t)2 * p.Bp;
    if (want_grad) n += (size_t)p.Bp * p.S * p.DP;
    return cfl_align256(n * sizeof(float));
}

int cfl_client_contrast_img_fwd(const float* F, const void* image_other, const float* G_other, const float* G_same,
                                const long long* idx, const float* F_old, int B, int M, int D, int B_div, float inv_tau, float weight,
                                int mode, int want_grad, float* out5, float* lse, float* pos, float* dF_inter, float* dF_moon,
                                void* ws, int* sync, void* stream_) {
    if (!F || !idx || !out5 || !ws || !sync || B <= 0 || M <= 0 || D <= 0 || !(inv_tau > 0.f) || !(mode & 3)) return CFL_EINVAL;
    if ((mode & 1) && (!G_other || !image_other || !lse)) return CFL_EINVAL;
    if ((mode & 2) && (!G_same || !F_old || B_div <= 0)) return CFL_EINVAL;
    if (want_grad < 0 || want_grad > 2) return CFL_EINVAL;
    if (want_grad == 1 && (((mode & 1) && !dF_inter) || ((mode & 2) && !dF_moon))) return CFL_EINVAL;
    if (want_grad == 2 && (!dF_inter || (mode & 4))) return CFL_EINVAL;      // direct: host-known coefficients only (no --loss_scale)
    if (!cfl_bank_gsplit_supported(B, M, D)) return CFL_ELIMIT;
    if ((((uintptr_t)F | (uintptr_t)G_other | (uintptr_t)image_other) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const gs::GsPlan p = gs::gs_plan(B, M, D);
    AttnPlan q;
    q.DT = p.DT; q.DP = p.DP; q.RG = p.RGF; q.S = p.S; q.Bp = p.Bp;
    const AttnWs w = attn_ws(ws, q);
    if (mode & 1) {
        const float sc2 = inv_tau * 1.4426950408889634f;
        int rc;
        // <DT, slot streams, waves, column split, staging (0 = LDS-DMA)>: D <= 256: 2 feature groups x 4 slot streams;
        // D = 512 / 768: 4 column-split pairs on one stream; always 8 waves, two per SIMD
        // A/B (profiles/r6_a3_dma_ab.jsonl): CFL_BANK_DMA_ASM=1 issues the slots' LDS-DMA from inline assembly, which takes the compiler's
        // s_waitcnt vmcnt(0) out of the middle of the iteration (it sits in front of the first transposing read of the gradient block with
        // the builtin).  A tie on three leases (D = 256: 32.8-33.4 vs 33.5 us; D = 512: 97.0 vs 98.8; D = 768: 74.8 vs 73.8): that wait is
        // not what the waves wait for.  The builtin form stays the default.
        static const bool dma_builtin = getenv("CFL_BANK_DMA_ASM") == nullptr;
#define CFL_STREAM(DT_, NGG_, NDS_)                                                                                       \
    (dma_builtin ? (want_grad ? launch_stream<DT_, NGG_, 8, NDS_, 0, true>(F, image_other, B, M, D, sc2, p, w, stream)   \
                              : launch_stream<DT_, NGG_, 8, NDS_, 0, false>(F, image_other, B, M, D, sc2, p, w, stream)) \
                 : (want_grad ? launch_stream<DT_, NGG_, 8, NDS_, -1, true>(F, image_other, B, M, D, sc2, p, w, stream)  \
                              : launch_stream<DT_, NGG_, 8, NDS_, -1, false>(F, image_other, B, M, D, sc2, p, w, stream)))
        // (measured and dropped, round 6: D > 256 with ONE wave per SIMD holding its 16 rows' whole F and O -- no column-split pair, no
        // logit exchange, one barrier per slot: D = 512 117 us vs 97, D = 768 spills, 222 vs 72: profiles/r6_a3_1wave_ab.jsonl)
        rc = p.DT == 24 ? CFL_STREAM(24, 1, 2) : p.DT == 16 ? CFL_STREAM(16, 1, 2) : p.DT == 8 ? CFL_STREAM(8, 4, 1) : CFL_STREAM(4, 4, 1);
#undef CFL_STREAM
        if (rc) return rc;
    }
    return launch_finish(w, p.S, p.DP, p.RGF, p.Bp, F, G_other, G_same, F_old, idx, B, M, D, B_div, inv_tau, weight, mode, want_grad,
                         out5, lse, pos, dF_inter, dF_moon, sync, stream);
}

// ---- round 4: con_w log-probabilities on the bank pass (wide-batch forward of bank_gsplit.h) -------------------------------
int cfl_conw_img_supported(int rows, int M, int D) {
    return (rows >= 512 && M > 0 && D >= 4 && D <= 768 && D % 4 == 0) ? 1 : 0;
}

size_t cfl_conw_img_ws_bytes(int rows, int M, int D) {
    if (!cfl_conw_img_supported(rows, M, D)) return 256;
    const gs::GsPlan p = gs::gs_plan(rows, M, D, 1);
    return cfl_align256((size_t)2 * p.RGF * p.S * BR * sizeof(float));
}

int cfl_conw_logprob_img(const float* V, const void* image, const float* G, int M, int D, int row0, int rows, float* out, void* ws,
                         void* stream_) {
    if (!V || !image || !G || !out || !ws || M <= 0 || D <= 0 || row0 < 0 || rows <= 0 || row0 + rows > M) return CFL_EINVAL;
    if (!cfl_conw_img_supported(rows, M, D)) return CFL_ELIMIT;
    if ((((uintptr_t)V | (uintptr_t)G | (uintptr_t)image) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const gs::GsPlan p = gs::gs_plan(rows, M, D, 1);
    AttnWs w;
    float* q = (float*)ws;
    w.part_m = q; q += (size_t)p.RGF * p.S * BR;
    w.part_l = q;
    w.rowbuf = nullptr; w.part_o = nullptr;
    const float* F = V + (size_t)row0 * D;
    const float sc2 = 1.4426950408889634f;
    // two step buffers of two 16-row slots each; D <= 256: 8 waves x 32 rows, two workgroups per CU; D <= 512: 4 waves, one per CU
#define CFL_WIDE32(DT_, NW_, ...)                                                                                                  \
    do {                                                                                                                         \
        CFL_SET_LDS((gs::cfl_bank_wide32_kernel<DT_, NW_, __VA_ARGS__>), 4 * 64 * 32 * DT_);                                      \
        CFL_LAUNCH(K_BANK_STREAM, (gs::cfl_bank_wide32_kernel<DT_, NW_, __VA_ARGS__>), dim3(p.S * p.RG), dim3(64 * NW_), 4 * 64 * 32 * DT_, stream, F, \
                   (const char*)image, rows, M, D, sc2, p.S, p.RG, w.part_m, w.part_l);                                          \
    } while (0)
    static const char* rbv = getenv("CFL_CONW_WIDE_RB");                // measurement knob: fragment-burst length of the 4-wave form
    if (p.DT == 24) {                                                     // D = 768: 16-row steps, three 48 KB slot buffers
        CFL_SET_LDS((gs::cfl_bank_wide16_kernel<24>), 3 * 64 * 32 * 24);
        CFL_LAUNCH(K_BANK_STREAM, (gs::cfl_bank_wide16_kernel<24>), dim3(p.S * p.RG), dim3(256), 3 * 64 * 32 * 24, stream, F, (const char*)image,
                   rows, M, D, sc2, p.S, p.RG, w.part_m, w.part_l);
    } else if (p.DT == 16) {
        // measured at M = 50 000 (profiles/r6_a5_conw_lines.jsonl): bursts of 2 / 4 / 8 contraction steps 7.26 / 6.59 / 7.00 ms; with
        // the issue order pinned (sched_group_barrier) 's' -1.3 %, 't' -2.6 % against the plain burst of 4 on the same lease
        if (rbv && rbv[0] == 'a') CFL_WIDE32(16, 4, 4, 2, true);         // LDS-DMA from inline assembly
        else if (rbv && rbv[0] == '8') CFL_WIDE32(16, 4, 8);
        else if (rbv && rbv[0] == '2') CFL_WIDE32(16, 4, 2);
        else if (rbv && rbv[0] == 's') CFL_WIDE32(16, 4, 4, 1);
        else if (rbv && rbv[0] == '4') CFL_WIDE32(16, 4, 4);
        else CFL_WIDE32(16, 4, 4, 2);
    } else if (p.DT == 8) {
        // M = 50 000, D = 256, one lease (profiles/r6_a5_wide32_branch_ab.jsonl): with the per-burst liveness branch of rounds 4-5 3.25-3.28
        // ms ('b'), without it 3.22-3.23 ('n'), without it and the issue order pinned per burst (eight reads, twelve MFMAs) 3.17-3.18
        // ... and with the LDS-DMA issued from inline assembly (the compiler's waits for the fragment reads become partial): 3.16-3.17 ms
        // against 3.21-3.24 on one lease (profiles/r6_a5_asm_dma_ab.jsonl; D = 512: a tie, the builtin stays there).  'd': builtin DMA.
        if (rbv && rbv[0] == 'b') CFL_WIDE32(8, 8, 4, 3);
        else if (rbv && rbv[0] == 'n') CFL_WIDE32(8, 8, 4, 0);
        else if (rbv && rbv[0] == 'd') CFL_WIDE32(8, 8, 4, 1);
        else CFL_WIDE32(8, 8, 4, 1, true);
    } else CFL_WIDE32(4, 8, 4);
#undef CFL_WIDE32
    CFL_LAUNCH(K_LSE_FINAL, cfl_conw_finish_kernel, dim3(cfl_cdiv(rows, 64)), dim3(256), 0, stream, w.part_m, w.part_l, p.S, F,
               G + (size_t)row0 * D, rows, D, out);
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
