// bank.hip -- rows A3 (inter-modal contrast vs. the frozen global bank), A4 (intra-modal
// contrast) and A5 (con_w log-prob + combine).
//
// Reference: src/algorithms/ClientTrainer.py:369-510, src/algorithms/MMClientTrainer.py:150-324,
//            src/algorithms/MMFL.py:298-335.
//
// cfl_bank_fwd_kernel streams the [M, D] bank G ONCE against a block of feature rows F and
// never materialises the [B, M] softmax: the MFMA runs "swapped" (A = G rows, B = F rows) so
// that every lane of the 32x32 accumulator owns ONE feature row f and 16 bank rows g per tile;
// the running (max, sum-exp) of the online log-sum-exp is then lane-local and needs a single
// cross-lane exchange (lane <-> lane+32) at the very end.  The same kernel with F = V[row0:]
// and inv_tau = 1 is the con_w log-prob (A5), tiled over both axes with an XCD-aware block map.
//
// HBM layout: F [B, D], G [M, D] row-major fp32; logits_t [M, B] (scaled logits, TRANSPOSED so
// that the stores of a wave are 128-byte contiguous); ws: part_m[S, Bp] part_l[S, Bp] (S = number
// of bank splits, Bp = B rounded up to 128) followed by the backward's split-K slabs [KS, B, D].
#include "common.h"
#include "tile_x3.h"

namespace {

struct BankPlan { int TN, BN, S, Bp; };
// Each workgroup loops over bank chunks gc = x, x + S, ... of 128 rows.
static BankPlan bank_plan(int B, int M) {
    BankPlan p;
    p.TN = (B <= 64) ? 1 : 2;
    p.BN = 64 * p.TN;
    p.Bp = cfl_cdiv(B, 128) * 128;
    const int fch = cfl_cdiv(B, p.BN);
    const int nch = cfl_cdiv(M, 128);
    int s = cfl_cdiv(512, fch);          // ~2 workgroups per CU overall
    if (fch >= 8) s = 8;                 // con_w regime: 8 splits x 8 feature chunks share one XCD's L2
    if (s > nch) s = nch;
    if (s < 1) s = 1;
    p.S = s;
    return p;
}

// (Round 4 measured 256 x 128 tiles -- a wave's 128 x 64 share reads 24 KB of LDS fragments per 48 MFMAs instead of 16 KB per
// 24 -- through a 3-stage LDS ring with counted vmcnt waits, one workgroup per CU: 4.79 ms vs 4.28 ms for this kernel at
// M = 50 000, D = 256.  With one wave per SIMD the soft-max epilogue (128 exp2 per lane and tile) runs behind the MFMAs instead
// of under the second workgroup's; the variant was removed again.)
template <int TN>
__global__ __launch_bounds__(256, 2) void cfl_bank_fwd_kernel(Opnd G, Opnd F, int B, int M, int Bp, float inv_tau,
                                                           float* logits_t, float* part_m, float* part_l, int x3mode) {
    constexpr int TM = 2;
    using C = TileCfg<TM, TN, true, true>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = gridDim.x;
    // logical id: XCD-contiguous so that the blocks resident on one XCD are (all splits) x (8 consecutive
    // feature chunks): their working set (8 F tiles + the current 8 G tiles) stays in the 4 MiB L2.
    const int nwg = gridDim.x * gridDim.y;
    const int v = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, nwg);
    const int sx = v % S, fy = v / S;
    const int col0 = fy * C::BN;
    const int nch = (M + C::BM - 1) / C::BM;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;

    float run_m[TN], run_l[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n) { run_m[n] = -INFINITY; run_l[n] = 0.f; }

    const int ntiles = sx < nch ? (nch - sx + S - 1) / S : 0;
    auto tile_fn = [&](int i) { return TileDesc{(sx + i * S) * C::BM, col0, 0, G.kdim}; };
    // online log-sum-exp in the base-2 domain: y = logit * log2(e); run_m holds max y, run_l = sum 2^(y - run_m)
    // (one v_exp_f32 per element instead of a full expf); converted back to natural units after the loop.
    const float sc2 = inv_tau * 1.4426950408889634f;
    auto epi_fn = [&](int i, const f32x16 (&acc)[TM][TN]) {
        const int row0 = (sx + i * S) * C::BM;
        const bool full = (row0 + C::BM <= M);                 // uniform: only the last chunk needs row masks
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int f = col0 + acc_col<TN>(wc, n, lane);
            float tmax = -INFINITY;
            if (full) {
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, acc[m][n][r]);
            } else {
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (row0 + acc_row<TM>(wr, m, r, lane) < M) tmax = fmaxf(tmax, acc[m][n][r]);
            }
            if (logits_t && f < B) {
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int g = row0 + acc_row<TM>(wr, m, r, lane);
                        if (full || g < M) logits_t[(long long)g * B + f] = acc[m][n][r] * inv_tau;
                    }
            }
            if (tmax > -INFINITY) {
                // inv_tau > 0, so the max of the scaled logits is the scaled max
                const float mn = fmaxf(run_m[n], tmax * sc2);
                float s = 0.f;
                if (full) {
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(fmaf(acc[m][n][r], sc2, -mn));
                } else {
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (row0 + acc_row<TM>(wr, m, r, lane) < M) s += __builtin_amdgcn_exp2f(fmaf(acc[m][n][r], sc2, -mn));
                }
                run_l[n] = run_l[n] * __builtin_amdgcn_exp2f(run_m[n] - mn) + s;
                run_m[n] = mn;
            }
        }
    };
    if (x3mode == 2)     // G, F are pre-split [hi | lo] images (launch_bank_fwd): direct-to-LDS staging, bf16 x 3 compute
        tile_gemm_seq_glds_with<TM, TN>(G, F, ntiles, tile_fn, lds, epi_fn,
                                        [](const float* sa, const float* sb, f32x16 (&acc)[TM][TN], int lane, int wr, int wc) {
                                            x3::compute<TM, TN>(reinterpret_cast<const char*>(sa), reinterpret_cast<const char*>(sb), acc,
                                                                lane, wr, wc);
                                        });
    else if (x3mode) x3::tile_gemm_seq<TM, TN>(G, F, ntiles, tile_fn, lds, epi_fn);      // split while staging (registers)
    else if (glds_ok(G, F)) tile_gemm_seq_glds<TM, TN>(G, F, ntiles, tile_fn, lds, epi_fn);
    else tile_gemm_seq<TM, TN, true, true>(G, F, ntiles, tile_fn, lds, XfIdentity(), epi_fn);
    // back to natural-log units, then combine the two half-waves (different bank rows of the same feature row) ...
#pragma unroll
    for (int n = 0; n < TN; ++n) run_m[n] *= 0.6931471805599453f;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const float om = __shfl_xor(run_m[n], 32, 64), ol = __shfl_xor(run_l[n], 32, 64);
        lse_merge(run_m[n], run_l[n], om, ol);
    }
    // ... and the two waves stacked along the bank axis (wr = 0, 1) through LDS
    float* sm = lds;                       // [2][BN] max, then [2][BN] sum
    float* sl = lds + 2 * C::BN;
    if (lane < 32) {
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int c = acc_col<TN>(wc, n, lane);
            sm[wr * C::BN + c] = run_m[n];
            sl[wr * C::BN + c] = run_l[n];
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < C::BN && col0 + t < Bp) {
        float m = sm[t], l = sl[t];
        lse_merge(m, l, sm[C::BN + t], sl[C::BN + t]);
        part_m[(size_t)sx * Bp + col0 + t] = m;
        part_l[(size_t)sx * Bp + col0 + t] = l;
    }
}

// one wave per feature row: merge the split partials -> lse; exact fp32 dot with the target bank
// row -> pos.  mode 0 (bank): target = idx[b];  mode 1 (con_w): target = row0 + b, out = pos - lse.
__global__ __launch_bounds__(256) void cfl_lse_final_kernel(const float* __restrict__ F, const float* __restrict__ G,
                                                            const long long* __restrict__ idx, int B, int M, int D,
                                                            int S, int Bp, float inv_tau, int mode, int row0,
                                                            const float* part_m, const float* part_l,
                                                            float* lse, float* pos, float* out_l) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    float m = -INFINITY, l = 0.f;
    for (int s = lane; s < S; s += 64) lse_merge(m, l, part_m[(size_t)s * Bp + b], part_l[(size_t)s * Bp + b]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64), ol = __shfl_xor(l, o, 64);
        lse_merge(m, l, om, ol);
    }
    const float v_lse = m + logf(l);
    long long tgt = mode == 0 ? idx[b] : (long long)row0 + b;
    float dot = 0.f;
    if (tgt >= 0 && tgt < M) {
        const float* f = F + (long long)b * D;
        const float* g = G + tgt * D;
        for (int k = lane; k < D; k += 64) dot = fmaf(f[k], g[k], dot);
        dot = wave_sum(dot) * inv_tau;
    }
    if (lane == 0) {
        if (lse) lse[b] = v_lse;
        if (pos) pos[b] = dot;
        if (out_l) out_l[b] = dot - v_lse;
    }
}

// loss = mean_b (lse - pos), one block, fixed summation order
__global__ __launch_bounds__(256) void cfl_bank_loss_kernel(const float* lse, const float* pos, int B, float* loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += lse[b] - pos[b];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) loss[0] = s / (float)B;
}

// staging transform of the backward: softmax probability from the stored scaled logit
struct XfSoftmax {
    const float* lse;
    __device__ __forceinline__ float operator()(float v, int f, int /*g*/) const { return expf(v - lse[f]); }
};

// split-K GEMM  slab[x][f][d] = sum_{g in split x} p[f][g] G[g][d],  p = exp(logit - lse[f])
// A[i = f][k = g] = logits_t[g*B + f] (K strided),  B[k = g][j = d] = G[g*D + d] (K strided)
template <int TM, int TN>
__global__ __launch_bounds__(256) void cfl_bank_bwd_kernel(Opnd P, Opnd G, const float* __restrict__ lse, int B, int M, int D,
                                                           int kper, float* slab, int x3mode) {
    using C = TileCfg<TM, TN, false, false>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int row0 = blockIdx.z * C::BM, col0 = blockIdx.y * C::BN;
    const int kbeg = blockIdx.x * kper;
    const int kend = min(M, kbeg + kper);
    f32x16 acc[TM][TN];
    if (x3mode) x3::tile_gemm<TM, TN, false, false>(P, G, row0, col0, kbeg, kend, lds, acc, XfSoftmax{lse});
    else tile_gemm<TM, TN, false, false>(P, G, row0, col0, kbeg, kend, lds, acc, XfSoftmax{lse});
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    float* out = slab + (size_t)blockIdx.x * B * D;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int j = col0 + acc_col<TN>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = row0 + acc_row<TM>(wr, m, r, lane);
                if (i < B && j < D) out[(long long)i * D + j] = acc[m][n][r];
            }
        }
}

// dF[f][d] = coef * (sum_x slab[x][f][d] - G[idx[f]][d]),  coef = inv_tau / B * gout.
// block = 64 consecutive elements x 4 slab groups (wave w sums slabs x = w, w+4, ...), fixed order => deterministic.
__global__ __launch_bounds__(256) void cfl_bank_bwd_reduce_kernel(const float* __restrict__ slab, int KS,
                                                                  const float* __restrict__ G, const long long* __restrict__ idx,
                                                                  int B, int M, int D, float inv_tau,
                                                                  const float* __restrict__ gout, float* dF) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long tot = (long long)B * D;
    const long long e = (long long)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (e < tot)
        for (int x = w; x < KS; x += 4) s += slab[(size_t)x * tot + e];
    sm[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < tot) {
        s = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
        const int f = (int)(e / D), d = (int)(e % D);
        const long long tgt = idx[f];
        const float gpos = (tgt >= 0 && tgt < M) ? G[tgt * D + d] : 0.f;
        dF[e] = (inv_tau / (float)B) * gout[0] * (s - gpos);
    }
}

// A4: one wave per row
__global__ __launch_bounds__(256) void cfl_intra_kernel(const float* __restrict__ F, const float* __restrict__ Gs,
                                                        const long long* __restrict__ idx, const float* __restrict__ Fo,
                                                        int B, int D, int M, int Bdiv, float inv_tau, float* rowloss, float* dF) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* f = F + (long long)b * D;
    const long long tgt = idx[b];
    const bool ok = tgt >= 0 && tgt < M;             // an out-of-range index contributes a zero positive, never an address
    const float* g = Gs + (ok ? tgt : 0) * D;
    const float* o = Fo + (long long)b * D;
    float pos = 0.f, neg = 0.f;
    for (int k = lane; k < D; k += 64) { pos = fmaf(f[k], ok ? g[k] : 0.f, pos); neg = fmaf(f[k], o[k], neg); }
    pos = wave_sum(pos); neg = wave_sum(neg);
    const float z = (neg - pos) * inv_tau;
    if (lane == 0) rowloss[b] = softplusf(z);
    if (dF) {
        const float c = sigmoidf(z) * inv_tau / (float)Bdiv;
        for (int k = lane; k < D; k += 64) dF[(long long)b * D + k] = c * (o[k] - (ok ? g[k] : 0.f));
    }
}
__global__ __launch_bounds__(256) void cfl_intra_loss_kernel(const float* rowloss, int B, int Bdiv, float* loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += rowloss[b];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) loss[0] = s / (float)Bdiv;
}

// KD distillation term: rowloss[b] = sum_d (out[b,d] - agg[idx[b],d])^2 and, optionally,
// dout[b,:] = scale * (out[b,:] - agg[idx[b],:]).  One wave per row; the gather never materialises the target.
__global__ __launch_bounds__(256) void cfl_kd_mse_kernel(const float* __restrict__ out, const float* __restrict__ agg,
                                                         const long long* __restrict__ idx, int B, int D, int M,
                                                         float scale, float* rowloss, float* dout) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const long long t = idx[b];
    const float* o = out + (long long)b * D;
    const float* a = agg + (t >= 0 && t < M ? t : 0) * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float e = o[k] - a[k];
        s = fmaf(e, e, s);
        if (dout) dout[(long long)b * D + k] = scale * e;
    }
    s = wave_sum(s);
    if (lane == 0) rowloss[b] = s;
}
__global__ __launch_bounds__(256) void cfl_kd_mse_final_kernel(const float* rowloss, int B, float scale, float* loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += rowloss[b];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) loss[0] = s * scale;
}

struct PtrPack { const float* p[64]; };

// A5 combine: W = softmax_c L[c, n]; out[n,:] = sum_c W[c,n] V_c[n,:].  One wave per row n.
__global__ __launch_bounds__(256) void cfl_conw_combine_kernel(PtrPack V, const float* __restrict__ L, int Cn, int M, int D,
                                                               float* out, float* W) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= M) return;
    float mx = -INFINITY;
    for (int c = 0; c < Cn; ++c) mx = fmaxf(mx, L[(size_t)c * M + n]);
    float den = 0.f;
    for (int c = 0; c < Cn; ++c) den += expf(L[(size_t)c * M + n] - mx);
    for (int k = lane; k < D; k += 64) {
        float s = 0.f;
        for (int c = 0; c < Cn; ++c) {
            const float w = expf(L[(size_t)c * M + n] - mx) / den;
            s = fmaf(w, V.p[c][(long long)n * D + k], s);
        }
        out[(long long)n * D + k] = s;
    }
    if (W && lane < Cn) W[(size_t)lane * M + n] = expf(L[(size_t)lane * M + n] - mx) / den;
}

static int bwd_ksplits(int B, int M, int D, int BM, int BN, int* kper) {
    const int tiles = cfl_cdiv(B, BM) * cfl_cdiv(D, BN);
    int ks = cfl_cdiv(512, tiles);
    const int kmax = cfl_cdiv(M, 32);
    if (ks > kmax) ks = kmax;
    if (ks > 128) ks = 128;
    if (ks < 1) ks = 1;
    int per = cfl_cdiv(cfl_cdiv(M, ks), 32) * 32;
    *kper = per;
    return cfl_cdiv(M, per);
}

static size_t bank_slab_floats(int B, int M, int D) {          // split-K slabs of the two-pass backward (largest tile variant)
    int kper;
    int ks = bwd_ksplits(B, M, D, 128, 128, &kper);
    const int ks2 = bwd_ksplits(B, M, D, 64, 64, &kper);
    const int ks3 = bwd_ksplits(B, M, D, 128, 64, &kper);
    ks = ks > ks2 ? ks : ks2;
    return (size_t)(ks > ks3 ? ks : ks3) * B * D;
}

struct BankWs { float *part_m, *part_l, *img_g, *img_f, *slab; };
// part_m | part_l | backward slabs [slab_floats] | pre-split image of the bank [M][Kp] | of the feature rows [B][Kp]
static BankWs bank_ws(void* ws, const BankPlan& pl, int B, int M, int D, bool with_slab) {
    const size_t slab_floats = with_slab ? bank_slab_floats(B, M, D) : 0;
    BankWs w;
    float* p = (float*)ws;
    const size_t kp = (size_t)x3::image_kp(D);
    w.part_m = p; p += (size_t)pl.S * pl.Bp;
    w.part_l = p; p += (size_t)pl.S * pl.Bp;
    w.slab = p; p += (slab_floats + 63) / 64 * 64;
    w.img_g = p; p += (size_t)M * kp;
    w.img_f = p;
    return w;
}
static int launch_bank_fwd(const float* F, const float* G, int B, int M, int D, float inv_tau,
                           float* logits_t, const BankPlan& pl, const BankWs& w, hipStream_t stream) {
    Opnd Go{G, D, M, D, cfl_opnd_vec(G, D, D)};
    Opnd Fo{F, D, B, D, cfl_opnd_vec(F, D, D)};
    const dim3 grid(pl.S, cfl_cdiv(B, pl.BN));
    // Pre-split images pay when the bank is re-read by many row groups (con_w: rows = M); a single client batch streams the
    // bank once, where the extra pass over it (and its footprint in the Infinity Cache) costs more than it saves.
    const int x3mode = cfl_get_exact_gemm() ? 0 : (B >= 1024 ? 2 : 1);
    if (x3mode == 2) {
        // split both operands once (a streaming pass, ~25 us per 51 MB) so that the GEMM itself stages with LDS-DMA and
        // spends no VALU / ds_write on conversion: the bf16 x 3 core is 5x shorter per stage than the fp32 MFMA and would
        // otherwise wait for the register staging
        const int kp = x3::image_kp(D);
        const long long qg = (long long)M * (kp / 4), qf = (long long)B * (kp / 4);
        CFL_LAUNCH(K_PAIR_PREP, x3::split_image_kernel, dim3((unsigned)((qg + 255) / 256)), dim3(256), 0, stream, G, (long long)D, M, D, kp,
                   w.img_g);
        CFL_LAUNCH(K_PAIR_PREP, x3::split_image_kernel, dim3((unsigned)((qf + 255) / 256)), dim3(256), 0, stream, F, (long long)D, B, D, kp,
                   w.img_f);
        Go = Opnd{w.img_g, kp, M, kp, 1};
        Fo = Opnd{w.img_f, kp, B, kp, 1};
    }
    if (pl.TN == 1) {
        using C = TileCfg<2, 1, true, true>;
        CFL_LAUNCH(K_BANK_FWD, (cfl_bank_fwd_kernel<1>), grid, dim3(256), C::LDS_BYTES, stream,
                   Go, Fo, B, M, pl.Bp, inv_tau, logits_t, w.part_m, w.part_l, x3mode);
    } else {
        using C = TileCfg<2, 2, true, true>;
        CFL_SET_LDS((cfl_bank_fwd_kernel<2>), C::LDS_BYTES);
        CFL_LAUNCH(K_BANK_FWD, (cfl_bank_fwd_kernel<2>), grid, dim3(256), C::LDS_BYTES, stream,
                   Go, Fo, B, M, pl.Bp, inv_tau, logits_t, w.part_m, w.part_l, x3mode);
    }
    return 0;
}

}  // namespace

extern "C" {

size_t cfl_bank_ws_bytes(int B, int M, int D) {
    if (B <= 0 || M <= 0 || D <= 0) return 256;
    const BankPlan pl = bank_plan(B, M);
    const size_t slab = (bank_slab_floats(B, M, D) + 63) / 64 * 64;
    const size_t img = B >= 1024 ? ((size_t)M + B) * x3::image_kp(D) : 0;       // pre-split images: con_w-sized problems only
    return cfl_align256((2 * (size_t)pl.S * pl.Bp + slab + img) * sizeof(float));
}

int cfl_bank_lse_fwd(const float* F, const float* G, const long long* idx, int B, int M, int D,
                     float inv_tau, float* lse, float* pos, float* loss, float* logits_t,
                     void* ws, void* stream_) {
    if (!F || !G || !idx || !lse || !pos || !ws || B <= 0 || M <= 0 || D <= 0 || !(inv_tau > 0.f)) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const BankPlan pl = bank_plan(B, M);
    BankWs w = bank_ws(ws, pl, B, M, D, true);
    int rc = launch_bank_fwd(F, G, B, M, D, inv_tau, logits_t, pl, w, stream);
    if (rc) return rc;
    CFL_LAUNCH(K_LSE_FINAL, cfl_lse_final_kernel, dim3(cfl_cdiv(B, 4)), dim3(256), 0, stream,
               F, G, idx, B, M, D, pl.S, pl.Bp, inv_tau, 0, 0, w.part_m, w.part_l, lse, pos, (float*)nullptr);
    if (loss) CFL_LAUNCH(K_BANK_LOSS, cfl_bank_loss_kernel, dim3(1), dim3(256), 0, stream, lse, pos, B, loss);
    return 0;
}

int cfl_bank_lse_bwd(const float* logits_t, const float* G, const long long* idx, const float* lse,
                     int B, int M, int D, float inv_tau, const float* gout_dev, float* dF,
                     void* ws, void* stream_) {
    if (!logits_t || !G || !idx || !lse || !gout_dev || !dF || !ws || B <= 0 || M <= 0 || D <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    BankWs w = bank_ws(ws, bank_plan(B, M), B, M, D, true);
    Opnd P{logits_t, B, B, M, cfl_opnd_vec(logits_t, B, B)};
    Opnd Go{G, D, D, M, cfl_opnd_vec(G, D, D)};
    int kper = 0, ks = 0;
    const int x3mode = cfl_get_exact_gemm() ? 0 : 1;
    if (B > 64 && D > 64) {
        // 128-row feature tiles; 64-wide D tiles when 128-wide ones would leave CUs with a single workgroup
        const bool narrow = cfl_cdiv(B, 128) * cfl_cdiv(D, 128) * 128 < 512;
        if (narrow) {
            using C = TileCfg<2, 1, false, false>;
            ks = bwd_ksplits(B, M, D, C::BM, C::BN, &kper);
            CFL_LAUNCH(K_BANK_BWD, (cfl_bank_bwd_kernel<2, 1>), dim3(ks, cfl_cdiv(D, C::BN), cfl_cdiv(B, C::BM)), dim3(256),
                       C::LDS_BYTES, stream, P, Go, lse, B, M, D, kper, w.slab, x3mode);
        } else {
            using C = TileCfg<2, 2, false, false>;
            ks = bwd_ksplits(B, M, D, C::BM, C::BN, &kper);
            CFL_LAUNCH(K_BANK_BWD, (cfl_bank_bwd_kernel<2, 2>), dim3(ks, cfl_cdiv(D, C::BN), cfl_cdiv(B, C::BM)), dim3(256),
                       C::LDS_BYTES, stream, P, Go, lse, B, M, D, kper, w.slab, x3mode);
        }
    } else {
        using C = TileCfg<1, 1, false, false>;
        ks = bwd_ksplits(B, M, D, C::BM, C::BN, &kper);
        CFL_LAUNCH(K_BANK_BWD, (cfl_bank_bwd_kernel<1, 1>), dim3(ks, cfl_cdiv(D, C::BN), cfl_cdiv(B, C::BM)), dim3(256),
                   C::LDS_BYTES, stream, P, Go, lse, B, M, D, kper, w.slab, x3mode);
    }
    const long long tot = (long long)B * D;
    CFL_LAUNCH(K_BANK_BWD_REDUCE, cfl_bank_bwd_reduce_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(256), 0, stream,
               w.slab, ks, G, idx, B, M, D, inv_tau, gout_dev, dF);
    return 0;
}

int cfl_intra_fwd(const float* F, const float* Gsame, const long long* idx, const float* Fold,
                  int B, int D, int M, int B_div, float inv_tau, float* loss, float* dF_unit, void* ws,
                  void* stream_) {
    if (!F || !Gsame || !idx || !Fold || !loss || !ws || B <= 0 || D <= 0 || M <= 0 || B_div <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    float* rowloss = (float*)ws;
    CFL_LAUNCH(K_INTRA, cfl_intra_kernel, dim3(cfl_cdiv(B, 4)), dim3(256), 0, stream,
               F, Gsame, idx, Fold, B, D, M, B_div, inv_tau, rowloss, dF_unit);
    CFL_LAUNCH(K_INTRA, cfl_intra_loss_kernel, dim3(1), dim3(256), 0, stream, rowloss, B, B_div, loss);
    return 0;
}

int cfl_kd_mse(const float* out, const float* agg, const long long* idx, int B, int D, int M, float weight,
               float* loss, float* dout_unit, void* ws, void* stream_) {
    if (!out || !agg || !idx || !loss || !ws || B <= 0 || D <= 0 || M <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    float* rowloss = (float*)ws;                       // cfl_intra_ws_bytes(B)
    const float inv = weight / ((float)B * (float)D);  // nn.MSELoss(): mean over all B*D elements
    CFL_LAUNCH(K_KD_MSE, cfl_kd_mse_kernel, dim3(cfl_cdiv(B, 4)), dim3(256), 0, stream, out, agg, idx, B, D, M, 2.f * inv,
               rowloss, dout_unit);
    CFL_LAUNCH(K_KD_MSE, cfl_kd_mse_final_kernel, dim3(1), dim3(256), 0, stream, rowloss, B, inv, loss);
    return 0;
}

size_t cfl_intra_ws_bytes(int B) { return cfl_align256((size_t)(B > 0 ? B : 1) * sizeof(float)); }

size_t cfl_conw_ws_bytes(int rows, int M, int D) {
    if (rows <= 0 || M <= 0 || D <= 0) return 256;
    const BankPlan pl = bank_plan(rows, M);
    const size_t img = rows >= 1024 ? ((size_t)M + rows) * x3::image_kp(D) : 0;
    return cfl_align256((2 * (size_t)pl.S * pl.Bp + img) * sizeof(float));
}

int cfl_conw_logprob(const float* V, const float* G, int M, int D, int row0, int rows,
                     float* out_l, void* ws, void* stream_) {
    if (!V || !G || !out_l || !ws || M <= 0 || D <= 0 || rows <= 0 || row0 < 0 || row0 + rows > M) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const BankPlan pl = bank_plan(rows, M);
    BankWs w = bank_ws(ws, pl, rows, M, D, false);
    const float* F = V + (long long)row0 * D;
    int rc = launch_bank_fwd(F, G, rows, M, D, 1.0f, nullptr, pl, w, stream);
    if (rc) return rc;
    CFL_LAUNCH(K_LSE_FINAL, cfl_lse_final_kernel, dim3(cfl_cdiv(rows, 4)), dim3(256), 0, stream,
               F, G, (const long long*)nullptr, rows, M, D, pl.S, pl.Bp, 1.0f, 1, row0, w.part_m, w.part_l,
               (float*)nullptr, (float*)nullptr, out_l);
    return 0;
}

int cfl_conw_combine(const float* const* Vptrs_host, const float* L, int Cn, int M, int D,
                     float* out, float* W_out, void* stream_) {
    if (!Vptrs_host || !L || !out || Cn <= 0 || M <= 0 || D <= 0) return CFL_EINVAL;
    if (Cn > 64) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    PtrPack pk;
    for (int c = 0; c < 64; ++c) pk.p[c] = c < Cn ? Vptrs_host[c] : nullptr;
    CFL_LAUNCH(K_CONW_COMBINE, cfl_conw_combine_kernel, dim3(cfl_cdiv(M, 4)), dim3(256), 0, stream,
               pk, L, Cn, M, D, out, W_out);
    return 0;
}

}  // extern "C"
