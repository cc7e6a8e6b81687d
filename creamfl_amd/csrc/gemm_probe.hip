// gemm_probe.hip -- C = A * B^T through tile_gemm (both operands K-contiguous): the calibration point for the
// fp32-MFMA building block (tools/kernel_bench.py --cases gemm) and a direct parity check of tile_gemm itself.
#include "common.h"

namespace {
template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void cfl_gemm_nt_kernel(Opnd A, Opnd B, int M, int N, float* Cout) {
    using C = TileCfg<TM, TN, true, true>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntc = (N + C::BN - 1) / C::BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (tile / ntc) * C::BM, col0 = (tile % ntc) * C::BN;
    f32x16 acc[TM][TN];
    tile_gemm<TM, TN, true, true>(A, B, row0, col0, 0, A.kdim, lds, acc, XfIdentity());
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
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
               stream, Ao, Bo, M, N, C);
    return 0;
}
