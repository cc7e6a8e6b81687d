// gemm_bf16.hip -- bf16 MFMA "NT" GEMM probe:  C[M,N] = A[M,K] * B[N,K]^T  (bf16 in/out, fp32 accumulation).
//
// Why it exists: a 1x1 convolution on a channels_last activation is exactly this GEMM on the [M = N*H*W, C] view
// (torchvision Bottleneck conv1 / conv3 inside src/networks/models/image_encoder.py:27-36), and the trunk
// convolutions are ~45 % of the bench step.  tools/conv_probe.py shows MIOpen at the HBM roofline only in layer1 and
// at 2-3 TB/s / 550-700 TFLOP/s in layer3, so this kernel was written to see whether the shared tile machine beats
// it.  Measured (tools/kernel_bench.py --cases gemm16, MI355X): it ties MIOpen (e.g. 14x14 256->1024: 49 vs 46 us,
// 1024->256: 46 vs 39 us, 56x56 64->64: 35 vs 34 us) and does not beat it, so THE PRODUCT PATH KEEPS MIOpen for
// convolutions; the entry point stays as the calibration point for bf16 MFMA work.
// Why it stalls: with 128x128 workgroup tiles the LDS pipe (fragment reads 128 KB + direct-to-LDS writes 64 KB per
// CU per K step vs 1024 MFMA cycles) caps the matrix pipe at ~1/3; 256x256 tiles lift that cap but leave one
// workgroup per CU and 1-16 K steps per tile, where prologue / epilogue latency dominates.
//
// Tile machine: v_mfma_f32_32x32x16_bf16, 4 waves as 2x2, wave tile TM x TN of 32x32, K step 64 bf16 (= one 128-byte
// row), direct-to-LDS staging (global_load_lds, source-side XOR swizzle -- the byte image is exactly the one the
// fp32 kernels use: common.h glds_stage / frag_swz), double-buffered, one barrier per K step, persistent workgroups,
// LDS-transposed epilogue with 16-byte stores.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}

// one 64-deep K step: 4 MFMA sub-steps of 16, fragments double-buffered in registers
template <int TM, int TN>
__device__ __forceinline__ void tile_compute_bf16(const float* sa, const float* sb, f32x16 (&acc)[TM][TN], int lane, int wr, int wc) {
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
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[kk & 1][m]),
                                                                    __builtin_bit_cast(bf16x8, fb[kk & 1][n]), acc[m][n], 0, 0, 0);
    }
}

// C[M, N] (bf16) = A[M, K] * B[N, K]^T.  Operands are addressed as 128-byte rows of "32 floats" (= 64 bf16).
// Persistent workgroups: the K steps of consecutive tiles form one software pipeline (the first stage of tile t+1 is
// in flight while tile t computes and stores), because with K = 64..1024 a tile is only 1..16 K steps long and the
// load latency / store drain of a tile-per-workgroup launch would dominate.
template <int TM, int TN, int OCC>
__global__ __launch_bounds__(256, OCC) void cfl_gemm_bf16_nt_kernel(Opnd A, Opnd B, int M, int N, u16* __restrict__ C, long long ldc,
                                                                    int ntiles) {
    constexpr int BM = 64 * TM, BN = 64 * TN, STAGE = (BM + BN) * 32;
    constexpr int PITCH = 64 * TN + 16;                   // bytes per band row (+16: de-phase the banks)
    constexpr int CPR = 4 * TN;                           // 16-byte chunks per band row
    constexpr int RPI = 64 / CPR;                         // rows per read instruction
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntc = (N + BN - 1) / BN;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    // the epilogue's transpose band lives behind the two stage buffers: no barrier between epilogue and next stage
    char* band = reinterpret_cast<char*>(lds + 2 * STAGE) + wid * (16 * PITCH);
    const int vb = xcd_remap(blockIdx.x, gridDim.x);       // tiles in flight at one time are XCD-contiguous
    const int nk = A.kdim / 32;
    const int my = (ntiles - vb + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my <= 0) return;
    int row0 = (vb / ntc) * BM, col0 = (vb % ntc) * BN;
    glds_stage<BM>(A, row0, 0, lds);
    glds_stage<BN>(B, col0, 0, lds + BM * 32);
    __syncthreads();
    int buf = 0;
    f32x16 acc[TM][TN];
    for (int i = 0; i < my; ++i) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        const bool has_next = (i + 1 < my);
        const int vn = vb + (i + 1) * (int)gridDim.x;
        const int nrow0 = (vn / ntc) * BM, ncol0 = (vn % ntc) * BN;
        for (int kt = 0; kt < nk; ++kt) {
            const float* sa = lds + buf * STAGE;
            const bool in_tile = (kt + 1 < nk);
            if (in_tile || has_next) {
                float* da = lds + (buf ^ 1) * STAGE;
                glds_stage<BM>(A, in_tile ? row0 : nrow0, in_tile ? (kt + 1) * 32 : 0, da);
                glds_stage<BN>(B, in_tile ? col0 : ncol0, in_tile ? (kt + 1) * 32 : 0, da + BM * 32);
            }
            tile_compute_bf16<TM, TN>(sa, sa + BM * 32, acc, lane, wr, wc);
            __syncthreads();
            buf ^= 1;
        }
        // Epilogue: each wave transposes its accumulators through its own LDS band, 16 rows at a time, so that the
        // global stores are 16 bytes per lane and whole 64*TN-byte row segments (the C/D layout holds one column per
        // lane: storing it directly would be 2-byte scattered writes).
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);           // 0..15
                        *reinterpret_cast<u16*>(band + rr * PITCH + (n * 32 + (lane & 31)) * 2) = f2bf(acc[m][n][h * 8 + r]);
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int gr0 = row0 + (wr * TM + m) * 32 + h * 16, gc0 = col0 + wc * TN * 32;
#pragma unroll
                for (int p = 0; p < 16 / RPI; ++p) {
                    const int rr = p * RPI + lane / CPR, c = lane % CPR;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(band + rr * PITCH + c * 16);
                    const int gi = gr0 + rr, gj = gc0 + c * 8;
                    if (gi < M && gj < N) *reinterpret_cast<f32x4*>(C + (long long)gi * ldc + gj) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        row0 = nrow0; col0 = ncol0;
    }
}

template <int TM, int TN, int OCC>
int launch_nt(const Opnd& A, const Opnd& B, int M, int N, u16* C, long long ldc, hipStream_t stream) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr size_t LDS = (size_t)2 * (BM + BN) * 32 * sizeof(float) + (size_t)4 * 16 * (64 * TN + 16);
    const int ntiles = cfl_cdiv(M, BM) * cfl_cdiv(N, BN);
    const int grid = ntiles < 256 * OCC ? ntiles : 256 * OCC;
    CFL_SET_LDS((cfl_gemm_bf16_nt_kernel<TM, TN, OCC>), LDS);
    CFL_LAUNCH(K_GEMM_BF16, (cfl_gemm_bf16_nt_kernel<TM, TN, OCC>), dim3(grid), dim3(256), LDS, stream, A, B, M, N, C, ldc, ntiles);
    return 0;
}

}  // namespace

extern "C" int cfl_gemm_bf16_nt(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc,
                                int M, int N, int K, int variant, void* stream_) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return CFL_EINVAL;
    if (K % 64 != 0 || N % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 ||
        (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    Opnd Ao{(const float*)A, lda / 2, M, K / 2, 1};
    Opnd Bo{(const float*)B, ldb / 2, N, K / 2, 1};
    u16* Cc = (u16*)C;
    if (variant == 0) variant = N >= 128 ? 22 : 21;
    switch (variant) {
        case 44: return launch_nt<4, 4, 1>(Ao, Bo, M, N, Cc, ldc, stream);
        case 42: return launch_nt<4, 2, 1>(Ao, Bo, M, N, Cc, ldc, stream);
        case 22: return launch_nt<2, 2, 2>(Ao, Bo, M, N, Cc, ldc, stream);
        case 21: return launch_nt<2, 1, 2>(Ao, Bo, M, N, Cc, ldc, stream);
        case 41: return launch_nt<4, 1, 2>(Ao, Bo, M, N, Cc, ldc, stream);
        default: return CFL_EINVAL;
    }
}
