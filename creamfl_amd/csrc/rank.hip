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

constexpr int RT = 64;      // tile rows (queries) and columns (gallery)
constexpr int RK = 16;      // k-step
constexpr int RLD = RT + 2; // LDS row stride in doubles

template <int PASS>
__global__ __launch_bounds__(256) void cfl_rank_kernel(const float* __restrict__ Q, const float* __restrict__ G,
                                                       const long long* __restrict__ qlab, const long long* __restrict__ glab,
                                                       int Nq, int Ng, int D, unsigned long long* posmax, int* ranks) {
    __shared__ double qs[RK][RLD];
    __shared__ double gs[RK][RLD];
    const int t = threadIdx.x;
    const int q0 = blockIdx.y * RT, g0 = blockIdx.x * RT;
    const int ty = t >> 4, tx = t & 15;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    const int lr = t >> 2, lk = (t & 3) * 4;        // loader: row lr, k offset lk..lk+3
    for (int k0 = 0; k0 < D; k0 += RK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + lk + e;
            const int q = q0 + lr, g = g0 + lr;
            qs[lk + e][lr] = (q < Nq && k < D) ? (double)Q[(long long)q * D + k] : 0.0;
            gs[lk + e][lr] = (g < Ng && k < D) ? (double)G[(long long)g * D + k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < RK; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = qs[kk][ty * 4 + i]; b[i] = gs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q0 + ty * 4 + i;
        if (q >= Nq) continue;
        if (PASS == 0) {
            const long long ql = qlab[q];
            bool any = false;
            double best = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = g0 + tx * 4 + j;
                if (g < Ng && glab[g] == ql) { best = any ? fmax(best, acc[i][j]) : acc[i][j]; any = true; }
            }
            if (any) atomicMax(&posmax[q], key_of(best));
        } else {
            const unsigned long long pk = posmax[q];
            int cnt = 0;
            if (pk == 0ull) {                 // no positive in the gallery: rank = Ng
                cnt = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) cnt += (g0 + tx * 4 + j < Ng) ? 1 : 0;
            } else {
                const double thr = val_of(pk);
#pragma unroll
                for (int j = 0; j < 4; ++j) cnt += (g0 + tx * 4 + j < Ng && acc[i][j] > thr) ? 1 : 0;
            }
            if (cnt) atomicAdd(&ranks[q], cnt);
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
    CFL_LAUNCH(K_RANK_POSMAX, (cfl_rank_kernel<0>), grid, dim3(256), 0, stream, Q, G, qlab, glab, Nq, Ng, D, posmax, ranks);
    CFL_LAUNCH(K_RANK_COUNT, (cfl_rank_kernel<1>), grid, dim3(256), 0, stream, Q, G, qlab, glab, Nq, Ng, D, posmax, ranks);
    return 0;
}

}  // extern "C"
