// gemm_probe.hip -- C = A * B^T through tile_gemm (both operands K-contiguous): the calibration point for the
// fp32-MFMA building block (tools/kernel_bench.py --cases gemm) and a direct parity check of tile_gemm itself.
#include "common.h"
#include "tile_x3.h"

namespace {
template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void cfl_gemm_nt_kernel(Opnd A, Opnd B, int M, int N, float* Cout, int x3mode) {
    using C = TileCfg<TM, TN, true, true>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntc = (N + C::BN - 1) / C::BN, ntr = (M + C::BM - 1) / C::BM;
    int ti, tj;
    tile_swizzle(xcd_remap(blockIdx.x, gridDim.x), ntr, ntc, ti, tj);
    const int row0 = ti * C::BM, col0 = tj * C::BN;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    f32x16 acc[TM][TN];
    if (x3mode) x3::tile_gemm_pipelined<TM, TN, true, true, 2>(A, B, row0, col0, 0, A.kdim, lds, acc, XfIdentity());
    else if (glds_ok(A, B)) tile_gemm_glds<TM, TN>(A, B, row0, col0, 0, A.kdim, lds, acc);
    else tile_gemm<TM, TN, true, true>(A, B, row0, col0, 0, A.kdim, lds, acc, XfIdentity());
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int j = col0 + acc_col<TN>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = row0 + acc_row<TM>(wr, m, r, lane);
                if (i < M && j < N) Cout[(long long)i * N + j] = acc[m][n][r];
            }
        }
}
}  // namespace

extern "C" int cfl_gemm_nt(const float* A, const float* B, int M, int N, int K, float* C, void* stream_) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    Opnd Ao{A, K, M, K, cfl_opnd_vec(A, K, K)};
    Opnd Bo{B, K, N, K, cfl_opnd_vec(B, K, K)};
    using Cf = TileCfg<2, 2, true, true>;
    CFL_SET_LDS((cfl_gemm_nt_kernel<2, 2>), Cf::LDS_BYTES);
    CFL_LAUNCH(K_GEMM_PROBE, (cfl_gemm_nt_kernel<2, 2>), dim3(cfl_cdiv(M, Cf::BM) * cfl_cdiv(N, Cf::BN)), dim3(256), Cf::LDS_BYTES,
               stream, Ao, Bo, M, N, C, cfl_get_exact_gemm() ? 0 : 1);
    return 0;
}

// ---- ablation probes of the K-loop (tools/kernel_bench.py --cases ablate): same grid and tile work as cfl_gemm_nt,
// MODE 0: MFMA only (fragments read once)   1: + LDS fragment reads every step   2: + barrier every step
// MODE 3: + register->LDS stage writes       4: + global loads (== the real loop, without the C store)
namespace {
template <int MODE>
__global__ __launch_bounds__(256, 2) void cfl_gemm_ablate_kernel(Opnd A, Opnd B, int nk, float* sink) {
    constexpr int TM = 2, TN = 2;
    using C = TileCfg<TM, TN, true, true>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    for (int i = threadIdx.x; i < C::LDS_FLOATS; i += 256) lds[i] = (float)((i * 7) % 13) * 0.01f;
    __syncthreads();
    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    f32x4 ra[C::BM / 32], rb[C::BN / 32];
    const int row0 = (blockIdx.x % 32) * 128, col0 = (blockIdx.x / 32 % 32) * 128;
    if (MODE >= 3) { g2r<true, C::BM, 0>(A, row0, 0, ra, XfIdentity()); g2r<true, C::BN, 0>(B, col0, 0, rb, XfIdentity()); }
    if (MODE == 0) {
        f32x4 fa[TM], fb[TN];
        for (int m = 0; m < TM; ++m) fa[m] = frag<true, C::A_LD>(lds, (wr * TM + m) * 32, 0, lane);
        for (int n = 0; n < TN; ++n) fb[n] = frag<true, C::B_LD>(lds + C::A_ELEMS, (wc * TN + n) * 32, 0, lane);
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < TN; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][t], fb[n][t], acc[m][n], 0, 0, 0);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            const float* sa = lds + (kt & 1) * C::STAGE;
            if (MODE >= 4) {
                g2r<true, C::BM, 0>(A, row0, ((kt + 1) * 32) % A.kdim, ra, XfIdentity());
                g2r<true, C::BN, 0>(B, col0, ((kt + 1) * 32) % A.kdim, rb, XfIdentity());
            }
            tile_compute<TM, TN, true, true>(sa, sa + C::A_ELEMS, acc, lane, wr, wc);
            if (MODE >= 3) {
                float* da = lds + ((kt + 1) & 1) * C::STAGE;
                r2s<true, C::BM, C::A_LD>(da, ra);
                r2s<true, C::BN, C::B_LD>(da + C::A_ELEMS, rb);
            }
            if (MODE >= 2) __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    if (s == 123.456f) sink[threadIdx.x] = s;
}
}  // namespace

extern "C" int cfl_gemm_ablate(const float* A, const float* B, int M, int K, int mode, int nk, float* sink, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Opnd Ao{A, K, M, K, 1};
    Opnd Bo{B, K, M, K, 1};
    using Cf = TileCfg<2, 2, true, true>;
    const dim3 grid(1024);
#define ABL(MD) case MD: CFL_SET_LDS((cfl_gemm_ablate_kernel<MD>), Cf::LDS_BYTES); \
        CFL_LAUNCH(K_GEMM_PROBE, (cfl_gemm_ablate_kernel<MD>), grid, dim3(256), Cf::LDS_BYTES, stream, Ao, Bo, nk, sink); break;
    switch (mode) { ABL(0) ABL(1) ABL(2) ABL(3) ABL(4) default: return CFL_EINVAL; }
#undef ABL
    return 0;
}
