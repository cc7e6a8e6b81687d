// rank.hip -- row A6: rank of the best positive for every retrieval query, without a sort.
//
// Reference: src/algorithms/eval_coco.py:37-51 (ParallelMatMulModule: fp64 mm of 7x replicated
// vectors, 7x7 fold, sort) and :296-317 (position of every positive in the sorted list, min).
// Equivalent (no ties): rank_q = #{g : sim(q,g) > max_{label_g == label_q} sim(q,g)}; the 49x
// replication is a positive constant factor and cannot change an ordering.  Similarities are
// accumulated in fp64 from the fp32-valued features, as the reference's float64 buffers are.
//
// Two passes of the SAME tile routine (bit-identical similarities in both):
//   pass 0: posmax[q] = max over positives (atomicMax on an order-preserving u64 key)
//   pass 1: ranks[q] += #{g in tile : sim > posmax[q]}          (integer atomics: deterministic)
#include "common.h"

namespace {

__device__ __forceinline__ unsigned long long key_of(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double val_of(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

constexpr int RT = 128;      // tile rows (queries) and columns (gallery)
constexpr int RK = 16;       // k-step
constexpr int RLD = RT + 2;  // LDS row stride in doubles

typedef double f64x4 __attribute__((ext_vector_type(4)));

// One workgroup = one 128 x 128 tile of the similarity matrix on the fp64 matrix pipe: 4 waves as 2 x 2, a wave owns
// 64 x 64 = 4 x 4 tiles of v_mfma_f64_16x16x4_f64 (A: lane = 16 k + row, B: lane = 16 k + column, D: column = lane & 15,
// row = (lane >> 4) + 4 r -- NOT the f32 map).  The fp32 features are widened on the way into LDS ([k][row] images); the next K step is
// fetched into registers while the current one is multiplied (64 MFMAs = ~4096 cycles per wave and step).
// The round-1 kernel did this tile with v_fma_f64 (34 TFLOP/s = 43 % of the fp64 rate: half its issue slots were LDS
// reads) and ran the full product twice; pass 0 now skips every tile without a label match.
template <int PASS>
__global__ __launch_bounds__(256, 2) void cfl_rank_kernel(const float* __restrict__ Q, const float* __restrict__ G,
                                                       const long long* __restrict__ qlab, const long long* __restrict__ glab,
                                                       int Nq, int Ng, int D, unsigned long long* posmax, int* ranks) {
    extern __shared__ __attribute__((aligned(16))) double rank_lds[];       // [2 buffers][q | g][RK][RLD] + labels
    typedef double (*Img)[RLD];
    long long* sql = reinterpret_cast<long long*>(rank_lds + 4 * RK * RLD);
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    const int q0 = blockIdx.y * RT, g0 = blockIdx.x * RT;
    if (PASS == 0) {                                      // positives only: a tile without a label match has nothing to add
        if (t < RT) sql[t] = q0 + t < Nq ? qlab[q0 + t] : 0;
        __syncthreads();
        const int g = g0 + (t & (RT - 1));
        bool hit = false;
        if (g < Ng) {
            const long long gl = glab[g];
            const int qb = (t >> 7) * (RT / 2);
            for (int i = 0; i < RT / 2; ++i) hit |= (q0 + qb + i < Nq) && (sql[qb + i] == gl);
        }
        if (!__syncthreads_or(hit)) return;
    }
    f64x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
    // loader: thread -> rows lr and lr + 64 of both operands, k offsets lk .. lk + 3
    const int lr = t >> 2, lk = (t & 3) * 4;
    float fq[2][4], fg[2][4];
    const bool vec = (D & 3) == 0 && ((((uintptr_t)Q) | ((uintptr_t)G)) & 15) == 0;      // rows are 16-byte aligned
    auto fetch = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = q0 + lr + 64 * h, g = g0 + lr + 64 * h;
            if (vec) {
                const int k = k0 + lk;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 vq = (q < Nq && k < D) ? *reinterpret_cast<const f32x4*>(Q + (long long)q * D + k) : z;
                const f32x4 vg = (g < Ng && k < D) ? *reinterpret_cast<const f32x4*>(G + (long long)g * D + k) : z;
#pragma unroll
                for (int e = 0; e < 4; ++e) { fq[h][e] = vq[e]; fg[h][e] = vg[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lk + e;
                    fq[h][e] = (q < Nq && k < D) ? Q[(long long)q * D + k] : 0.f;
                    fg[h][e] = (g < Ng && k < D) ? G[(long long)g * D + k] : 0.f;
                }
            }
        }
    };
    auto stage = [&](int buf) {                            // registers -> LDS image `buf`, widened to fp64
        Img qs = reinterpret_cast<Img>(rank_lds + (2 * buf) * RK * RLD), gs = reinterpret_cast<Img>(rank_lds + (2 * buf + 1) * RK * RLD);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qs[lk + e][lr + 64 * h] = (double)fq[h][e];
                gs[lk + e][lr + 64 * h] = (double)fg[h][e];
            }
    };
    const int ar = wr * 64 + (lane & 15), bc = wc * 64 + (lane & 15), kq = lane >> 4;
    // One barrier per K step: while step k is multiplied out of image `cur`, step k+1 (fetched during step k-1) is
    // written to the other image and step k+2 is requested -- all in one basic block, so the conversions, LDS writes and
    // global loads issue in the shadow of the 64 MFMAs.
    fetch(0);
    stage(0);
    if (RK < D) fetch(RK);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < D; k0 += RK) {
        Img qs = reinterpret_cast<Img>(rank_lds + (2 * cur) * RK * RLD), gs = reinterpret_cast<Img>(rank_lds + (2 * cur + 1) * RK * RLD);
        if (k0 + RK < D) stage(cur ^ 1);
        if (k0 + 2 * RK < D) fetch(k0 + 2 * RK);
#pragma unroll
        for (int kg = 0; kg < RK / 4; ++kg) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = qs[kg * 4 + kq][ar + 16 * i]; b[i] = gs[kg * 4 + kq][bc + 16 * i]; }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }
    // element (m, n, r) of this lane: query q0 + wr 64 + 16 m + (lane >> 4) + 4 r, gallery g0 + wc 64 + 16 n + (lane & 15)
    const int gbase = g0 + wc * 64 + (lane & 15);
    long long gl[4];
    if (PASS == 0) {
#pragma unroll
        for (int n = 0; n < 4; ++n) gl[n] = gbase + 16 * n < Ng ? glab[gbase + 16 * n] : 0;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = q0 + wr * 64 + 16 * m + (lane >> 4) + 4 * r;        // uniform over the 16 lanes of a group
            if (PASS == 0) {
                unsigned long long key = 0ull;
                if (q < Nq) {
                    const long long ql = qlab[q];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (gbase + 16 * n < Ng && gl[n] == ql) {
                            const unsigned long long kk = key_of(acc[m][n][r]);
                            key = kk > key ? kk : key;
                        }
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    const unsigned long long other = __shfl_xor(key, o);
                    key = other > key ? other : key;
                }
                if ((lane & 15) == 0 && key != 0ull) atomicMax(&posmax[q], key);
            } else {
                int cnt = 0;
                if (q < Nq) {
                    const unsigned long long pk = posmax[q];
                    if (pk == 0ull) {                 // no positive in the gallery: rank = Ng
#pragma unroll
                        for (int n = 0; n < 4; ++n) cnt += (gbase + 16 * n < Ng) ? 1 : 0;
                    } else {
                        const double thr = val_of(pk);
#pragma unroll
                        for (int n = 0; n < 4; ++n) cnt += (gbase + 16 * n < Ng && acc[m][n][r] > thr) ? 1 : 0;
                    }
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) cnt += __shfl_xor(cnt, o);
                if ((lane & 15) == 0 && cnt) atomicAdd(&ranks[q], cnt);
            }
        }
}

__global__ void cfl_rank_init_kernel(unsigned long long* posmax, int* ranks, int Nq) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < Nq) { posmax[q] = 0ull; ranks[q] = 0; }
}

}  // namespace

extern "C" {

size_t cfl_rank_ws_bytes(int Nq, int Ng, int D) {
    (void)Ng; (void)D;
    return cfl_align256((size_t)(Nq > 0 ? Nq : 1) * sizeof(unsigned long long));
}

int cfl_rank_count(const float* Q, const float* G, const long long* qlab, const long long* glab,
                   int Nq, int Ng, int D, int* ranks, void* ws, void* stream_) {
    if (!Q || !G || !qlab || !glab || !ranks || !ws || Nq <= 0 || Ng <= 0 || D <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    unsigned long long* posmax = (unsigned long long*)ws;
    CFL_LAUNCH(K_RANK_POSMAX, cfl_rank_init_kernel, dim3(cfl_cdiv(Nq, 256)), dim3(256), 0, stream, posmax, ranks, Nq);
    const dim3 grid(cfl_cdiv(Ng, RT), cfl_cdiv(Nq, RT));
    constexpr size_t LDS = (size_t)4 * RK * RLD * sizeof(double) + RT * sizeof(long long);
    CFL_SET_LDS((cfl_rank_kernel<0>), LDS);
    CFL_SET_LDS((cfl_rank_kernel<1>), LDS);
    CFL_LAUNCH(K_RANK_POSMAX, (cfl_rank_kernel<0>), grid, dim3(256), LDS, stream, Q, G, qlab, glab, Nq, Ng, D, posmax, ranks);
    CFL_LAUNCH(K_RANK_COUNT, (cfl_rank_kernel<1>), grid, dim3(256), LDS, stream, Q, G, qlab, glab, Nq, Ng, D, posmax, ranks);
    return 0;
}

}  // extern "C"
