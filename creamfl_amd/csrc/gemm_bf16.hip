// gemm_bf16.hip -- bf16 MFMA GEMMs for the 1x1 convolutions of the ResNet trunk (bf16 in/out, fp32 accumulation).
//
// A 1x1 convolution on a channels_last activation is a GEMM on the [M = N*H*W, C] view (torchvision Bottleneck conv1 /
// conv3 inside src/networks/models/image_encoder.py:27-36), and the trunk convolutions are ~45 % of the bench step.
// docs/history/tools/conv_probe.py / wgrad_probe.py show MIOpen at the HBM roofline only in layer1 and at 2-3 TB/s /
// 550-700 TFLOP/s in layer3.  Measured (tools/kernel_bench.py --cases gemm16 / wgrad16, MI355X):
//   NT  C[M,N] = A[M,K] B[N,K]^T : ties MIOpen's FORWARD (14x14 256->1024: 49 vs 46 us; 56x56 64->64: 35 vs 34 us) but
//       beats its BACKWARD-DATA kernels on every ResNet-101 shape (14x14 1024->256: 78 -> 48 us; 56x56 256->64:
//       184 -> 110 us)  => the product routes the data gradient of the 1x1 convolutions here (ops.conv1x1).
//   (A TN kernel for the weight gradient C[N1,N2] = A[M,N1]^T B[M,N2] -- transposing register loader, split-K -- measured on
//   par with MIOpen alone and +0.8 ms inside the step (its 32 MB of split-K partials); removed in round 4, docs/history/DESIGN_r1-r4.md section 7.)
// Why the forward stalls: with 128x128 workgroup tiles the LDS pipe (fragment reads 128 KB + direct-to-LDS writes 64 KB
// per CU per K step vs 1024 MFMA cycles) caps the matrix pipe at ~1/3; 256x256 tiles lift that cap but leave one
// workgroup per CU and 1-16 K steps per tile, where prologue / epilogue latency dominates.  A 3-stage LDS ring with the
// loads two K steps ahead (counted vmcnt + raw s_barrier, one workgroup per CU) was tried and is SLOWER (63-68 us vs
// 48-50 us at 14x14 256<->1024): two resident workgroups per CU matter more than prefetch depth here.
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
// JOIN (round 3): the epilogue adds a second bf16 matrix and applies a 1-bit-per-element mask before the store,
//   C = (A B^T + addp) . mask        (mask byte i covers elements 8 i .. 8 i + 7 of the dense [M, N] matrix, bit k = element k)
// -- the gradient join of a residual block: the data gradient of the block's first 1x1 convolution (this GEMM) + the
// gradient that arrives over the skip connection, masked by the ReLU of the BatchNorm that produced the block input.  The
// BatchNorm backward of that layer then reads ONE pre-masked gradient instead of two gradients + the mask in both of its
// passes, and hands the same tensor on as its residual gradient (bnorm.hip writes nothing for it): -4 bytes per element.
__device__ __forceinline__ f32x4 join8(const f32x4 v, const f32x4 b, unsigned mb) {
    union { f32x4 q; unsigned u[4]; } a, c, o;
    a.q = v; c.q = b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(a.u[i] << 16) + __uint_as_float(c.u[i] << 16);
        const float hi = __uint_as_float(a.u[i] & 0xffff0000u) + __uint_as_float(c.u[i] & 0xffff0000u);
        const unsigned l = ((mb >> (2 * i)) & 1u) ? f2bf(lo) : 0u;
        const unsigned h = ((mb >> (2 * i + 1)) & 1u) ? f2bf(hi) : 0u;
        o.u[i] = l | (h << 16);
    }
    return o.q;
}

template <int TM, int TN, int OCC, bool JOIN>
__global__ __launch_bounds__(256, OCC) void cfl_gemm_bf16_nt_kernel(Opnd A, Opnd B, int M, int N, u16* __restrict__ C, long long ldc,
                                                                    int ntiles, const u16* __restrict__ addp,
                                                                    const unsigned char* __restrict__ maskp) {
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
                        const u16 hv = f2bf(acc[m][n][h * 8 + r]);
                        *reinterpret_cast<u16*>(band + rr * PITCH + (n * 32 + (lane & 31)) * 2) = hv;
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int gr0 = row0 + (wr * TM + m) * 32 + h * 16, gc0 = col0 + wc * TN * 32;
#pragma unroll
                for (int p = 0; p < 16 / RPI; ++p) {
                    const int rr = p * RPI + lane / CPR, c = lane % CPR;
                    f32x4 v = *reinterpret_cast<const f32x4*>(band + rr * PITCH + c * 16);
                    const int gi = gr0 + rr, gj = gc0 + c * 8;
                    if (gi < M && gj < N) {
                        const long long e = (long long)gi * ldc + gj;
                        if (JOIN) v = join8(v, *reinterpret_cast<const f32x4*>(addp + e), maskp[e >> 3]);
                        *reinterpret_cast<f32x4*>(C + e) = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        row0 = nrow0; col0 = ncol0;
    }
}

template <int TM, int TN, int OCC, bool JOIN = false>
int launch_nt(const Opnd& A, const Opnd& B, int M, int N, u16* C, long long ldc, hipStream_t stream, const u16* addp = nullptr,
              const unsigned char* maskp = nullptr) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr size_t LDS = (size_t)2 * (BM + BN) * 32 * sizeof(float) + (size_t)4 * 16 * (64 * TN + 16);
    const int ntc = cfl_cdiv(N, BN);
    const int ntiles = cfl_cdiv(M, BM) * ntc;
    int grid = ntiles < 256 * OCC ? ntiles : 256 * OCC;
    CFL_SET_LDS((cfl_gemm_bf16_nt_kernel<TM, TN, OCC, JOIN>), LDS);
    CFL_LAUNCH(K_GEMM_BF16, (cfl_gemm_bf16_nt_kernel<TM, TN, OCC, JOIN>), dim3(grid), dim3(256), LDS, stream, A, B, M, N, C, ldc, ntiles,
               addp, maskp);
    return 0;
}

// ---- NT with the B tile RESIDENT in LDS (round 3) ---------------------------------------------------------------------------
// The data gradients of the 1x1 convolutions are HBM-bound GEMMs (K, N <= 2048: below the ridge point), and the tile kernel
// above reaches 2-3 TB/s on them: a workgroup has ONE K step (32 KB) in flight between two barriers, i.e. <= 64 KB per CU,
// which at the loaded HBM latency is worth 3-4 TB/s at best.  This kernel is built around bytes in flight instead:
//   * a workgroup (8 waves, one per CU) owns ONE column tile of B (the weight: BN = 64 / 128 columns x all of K <= 1024, 64-128
//     KB) for its whole life -- staged once by LDS-DMA into the swizzled image frag reads want, then only read: the main loop
//     has NO barrier;
//   * every wave streams its own 32-row tiles of A straight from global memory into registers in MFMA operand layout (a lane
//     holds 64 contiguous bytes of its row per 64-deep K step; the k order inside a K step is permuted identically on the B side)
//     through a ring of R = 4 / 8 K steps: 16-32 KB in flight per wave, 128-256 KB per CU, waves drift freely;
//   * the MFMA takes B as its first operand, so a lane holds 4 consecutive columns of one C row per accumulator quad: the wave
//     transposes through its private LDS band with 8-byte writes / 16-byte reads and stores whole row segments;
//   * JOIN operands (skip gradient + ReLU mask bits of the tile) are requested when the tile starts, not in the epilogue.
// Block b runs on XCD b % 8: the column tiles of one row range are consecutive blocks OF ONE XCD, so A comes from HBM once per
// row range and from that XCD's L2 for the other column tiles.
// STATS (forward of a 1x1 convolution that a training-mode BatchNorm follows): per-column sum and sum of squares of the STORED
// bf16 outputs, accumulated by every wave over its own tiles (a lane reads back the same 8 columns of every tile: 16 running
// sums), reduced over the wave at the end and written as ONE partial row per wave, pstat[0][prow][N] / pstat[1][prow][N] with
// prow = 8 * row range + wave -- the layout cfl_bn_final_kernel reduces (nblk = 2048 / column tiles): the BatchNorm needs no
// statistics pass of its own.
template <int NK, int TN, int R, bool JOIN, bool STATS = false>
__global__ __launch_bounds__(512, 1) void cfl_gemm_bf16_nt_bres_kernel(const u16* __restrict__ A, long long lda, const u16* __restrict__ B,
                                                                       long long ldb, int M, int N, u16* __restrict__ C, long long ldc,
                                                                       const u16* __restrict__ addp,
                                                                       const unsigned char* __restrict__ maskp, int ntc,
                                                                       float* __restrict__ pstat = nullptr) {
    constexpr int BN = 32 * TN;
    constexpr int NB = NK >= 16 ? 1 : TN;               // 32-column tiles per epilogue pass (LDS budget)
    constexpr int PITCH = NB * 64 + 16;                 // bytes per band row
    constexpr int CPR = NB * 4;                         // 16-byte pieces per band row
    constexpr int RPI = 64 / CPR;                       // rows per read instruction
    constexpr int NRI = 32 / RPI;                       // read instructions per pass
    constexpr int U0 = NK > R ? NK : R;                 // units per trip of the main loop (slot and K step of a unit are static)
    constexpr int U = (JOIN && U0 < 2 * NK) ? 2 * NK : U0;   // JOIN: an even number of tiles per trip (static operand buffer parity)
    constexpr int NJ = JOIN ? TN * 2 : 1;
    static_assert(!STATS || NB == TN, "statistics epilogue: one band pass per tile");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* const sb = reinterpret_cast<char*>(lds);      // B image: NK stages of [BN rows][128 bytes], 16-byte slots XOR-swizzled
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const band = sb + NK * BN * 128 + wid * (32 * PITCH);
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int ct = jb % ntc, grp = jb / ntc, gpx = 32 / ntc;
    const int col0 = ct * BN;
    {
        constexpr int NG = NK * BN / 8;                 // 8-row groups (1 KB each)
        for (int g = wid; g < NG; g += 8) {
            const int s = g / (BN / 8), rr = (g % (BN / 8)) * 8 + (lane >> 3);
            const int q = (lane & 7) ^ ((rr >> 1) & 7);
            int rg = col0 + rr;
            rg = rg < N ? rg : N - 1;
            const u16* src = B + (long long)rg * ldb + s * 64 + q * 8;
            __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(sb + g * 1024), 16, 0, 0);
        }
    }
    __syncthreads();
    const int T32 = (M + 31) >> 5;
    const int NR = 8 * gpx, rid = xcd * gpx + grp;
    const int t_lo = (int)((long long)rid * T32 / NR), t_hi = (int)((long long)(rid + 1) * T32 / NR);
    const int n_my = (t_hi - t_lo - wid + 7) >> 3;     // 32-row tiles of this wave: t_lo + wid + 8 i
    float ssum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ssq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n_my <= 0) {
        if (STATS && lane < CPR) {                       // a wave without tiles still owns a (zero) partial row
            const long long nblk = 8ll * NR;
            float* ps = pstat + ((long long)rid * 8 + wid) * N + col0 + lane * 8;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(ps) = z; *reinterpret_cast<f32x4*>(ps + 4) = z;
            *reinterpret_cast<f32x4*>(ps + nblk * N) = z; *reinterpret_cast<f32x4*>(ps + nblk * N + 4) = z;
        }
        return;
    }
    const int r = lane & 31, h = lane >> 5;
    const int total = n_my * NK;

    f32x4 ar[R][4];
    auto load_unit = [&](f32x4 (&dst)[4], int tt, int s) {
        tt = tt < n_my ? tt : n_my - 1;
        int row = (t_lo + wid + 8 * tt) * 32 + r;
        row = row < M ? row : M - 1;
        const u16* p = A + (long long)row * lda + s * 64 + h * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const f32x4*>(p + 8 * j);
    };
#pragma unroll
    for (int i = 0; i < R; ++i) load_unit(ar[i], i / NK, i % NK);

    f32x16 acc[TN];
    // JOIN operands (skip gradient, mask bytes) of a tile in the read-back layout of the epilogue, double-buffered: requested one
    // whole tile ahead, so that the epilogue never waits for them
    f32x4 jadd[2][NJ];
    unsigned jm[2][NJ];
    auto load_join = [&](f32x4 (&ja)[NJ], unsigned (&jb)[NJ], int tt) {
        const int tc = tt < n_my ? tt : n_my - 1;
        const int row0 = (t_lo + wid + 8 * tc) * 32;
#pragma unroll
        for (int bp = 0; bp < TN / NB; ++bp)
#pragma unroll
            for (int q = 0; q < NRI; ++q) {
                int gi = row0 + q * RPI + lane / CPR;
                gi = gi < M ? gi : M - 1;
                const long long e = (long long)gi * ldc + col0 + bp * NB * 32 + (lane % CPR) * 8;
                ja[bp * NRI + q] = *reinterpret_cast<const f32x4*>(addp + e);
                jb[bp * NRI + q] = maskp[e >> 3];
            }
    };
    if (JOIN) load_join(jadd[0], jm[0], 0);
    // B fragment of 32-column tile n, K step s, sub-step kk: lane (c = lane & 31, h) reads 16-byte piece 4 h + kk of row n 32 + c
    const int bsw = (r >> 1) & 7;
    const char* const brow = sb + r * 128;
    for (int base = 0; base < total; base += U) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int slot = i % R, s = i % NK;
            const int tt = base / NK + i / NK;
            if (s == 0) {
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
                if (JOIN) load_join(jadd[((i / NK) + 1) & 1], jm[((i / NK) + 1) & 1], tt + 1);
            }
            f32x4 fb[2][TN];
#pragma unroll
            for (int n = 0; n < TN; ++n)
                fb[0][n] = *reinterpret_cast<const f32x4*>(brow + s * (BN * 128) + n * 4096 + (((4 * h + 0) ^ bsw) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk < 3) {
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        fb[(kk + 1) & 1][n] =
                            *reinterpret_cast<const f32x4*>(brow + s * (BN * 128) + n * 4096 + (((4 * h + kk + 1) ^ bsw) << 4));
                }
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[kk & 1][n]),
                                                                     __builtin_bit_cast(bf16x8, ar[slot][kk]), acc[n], 0, 0, 0);
            }
            load_unit(ar[slot], base / NK + (i + R) / NK, (i + R) % NK);
            if (s == NK - 1 && tt < n_my) {
                const int row0 = (t_lo + wid + 8 * tt) * 32;
#pragma unroll
                for (int bp = 0; bp < TN / NB; ++bp) {
#pragma unroll
                    for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x16& a = acc[bp * NB + nn];
                            uint2 o;
                            o.x = (unsigned)f2bf(a[4 * g + 0]) | ((unsigned)f2bf(a[4 * g + 1]) << 16);
                            o.y = (unsigned)f2bf(a[4 * g + 2]) | ((unsigned)f2bf(a[4 * g + 3]) << 16);
                            *reinterpret_cast<uint2*>(band + r * PITCH + (nn * 32 + 8 * g + 4 * h) * 2) = o;
                        }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int q = 0; q < NRI; ++q) {
                        const int rr = q * RPI + lane / CPR, c = lane % CPR;
                        f32x4 v = *reinterpret_cast<const f32x4*>(band + rr * PITCH + c * 16);
                        const int gi = row0 + rr;
                        if (gi < M) {
                            const long long e = (long long)gi * ldc + col0 + bp * NB * 32 + c * 8;
                            if (JOIN) v = join8(v, jadd[(i / NK) & 1][bp * NRI + q], jm[(i / NK) & 1][bp * NRI + q]);
                            *reinterpret_cast<f32x4*>(C + e) = v;
                            if (STATS) {
                                union { f32x4 qv; unsigned u[4]; } w;
                                w.qv = v;
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const float lo = __uint_as_float(w.u[k] << 16), hi = __uint_as_float(w.u[k] & 0xffff0000u);
                                    ssum[2 * k] += lo; ssq[2 * k] = fmaf(lo, lo, ssq[2 * k]);
                                    ssum[2 * k + 1] += hi; ssq[2 * k + 1] = fmaf(hi, hi, ssq[2 * k + 1]);
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
        }
    }
    if (STATS) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) {
                ssum[k] += __shfl_xor(ssum[k], o, 64);
                ssq[k] += __shfl_xor(ssq[k], o, 64);
            }
        if (lane < CPR) {
            const long long nblk = 8ll * NR;
            float* ps = pstat + ((long long)rid * 8 + wid) * N + col0 + lane * 8;
            const f32x4 a0 = {ssum[0], ssum[1], ssum[2], ssum[3]}, a1 = {ssum[4], ssum[5], ssum[6], ssum[7]};
            const f32x4 b0 = {ssq[0], ssq[1], ssq[2], ssq[3]}, b1 = {ssq[4], ssq[5], ssq[6], ssq[7]};
            *reinterpret_cast<f32x4*>(ps) = a0; *reinterpret_cast<f32x4*>(ps + 4) = a1;
            *reinterpret_cast<f32x4*>(ps + nblk * N) = b0; *reinterpret_cast<f32x4*>(ps + nblk * N + 4) = b1;
        }
    }
}

template <int NK, int TN, int R>
int launch_bres_stats(const u16* A, long long lda, const u16* B, long long ldb, int M, int N, u16* C, float* pstat, hipStream_t stream) {
    constexpr int BN = 32 * TN;
    constexpr size_t LDS = (size_t)NK * BN * 128 + (size_t)8 * 32 * (TN * 64 + 16);
    CFL_SET_LDS((cfl_gemm_bf16_nt_bres_kernel<NK, TN, R, false, true>), LDS);
    CFL_LAUNCH(K_GEMM_BF16, (cfl_gemm_bf16_nt_bres_kernel<NK, TN, R, false, true>), dim3(256), dim3(512), LDS, stream, A, lda, B, ldb, M, N, C,
               (long long)N, (const u16*)nullptr, (const unsigned char*)nullptr, N / BN, pstat);
    return 0;
}

// partial rows the statistics epilogue writes per column (0: shape not taken)
inline int bres_stats_nblk(int M, int N, int K) {
    if (M <= 0 || N % 128 != 0 || (K != 64 && K != 128 && K != 256)) return 0;
    const int ntc = N / 128;
    if (ntc > 32 || (32 % ntc) != 0) return 0;
    return 2048 / ntc;
}

template <int NK, int TN, int R, bool JOIN>
int launch_bres(const u16* A, long long lda, const u16* B, long long ldb, int M, int N, u16* C, long long ldc, const u16* addp,
                const unsigned char* maskp, hipStream_t stream) {
    constexpr int BN = 32 * TN, NB = NK >= 16 ? 1 : TN;
    constexpr size_t LDS = (size_t)NK * BN * 128 + (size_t)8 * 32 * (NB * 64 + 16);
    CFL_SET_LDS((cfl_gemm_bf16_nt_bres_kernel<NK, TN, R, JOIN>), LDS);
    CFL_LAUNCH(K_GEMM_BF16, (cfl_gemm_bf16_nt_bres_kernel<NK, TN, R, JOIN>), dim3(256), dim3(512), LDS, stream, A, lda, B, ldb, M, N, C, ldc,
               addp, maskp, N / BN, (float*)nullptr);
    return 0;
}

// 0 when the B-resident kernel does not take the shape (the tile kernel does), else 1 after the launch
template <bool JOIN>
int try_bres(const u16* A, long long lda, const u16* B, long long ldb, int M, int N, int K, u16* C, long long ldc, const u16* addp,
             const unsigned char* maskp, hipStream_t stream) {
    static const bool off = getenv("CFL_GEMM_NO_BRES") != nullptr;
    if (off || N % 64 != 0) return 0;
    const bool wide = (N % 128 == 0) && K <= 256;
    const int ntc = N / (wide ? 128 : 64);
    if (ntc > 32 || (32 % ntc) != 0) return 0;
    // ring depth: 8 K steps where the registers allow it (256 per wave at 8 waves per CU), else 4
#define CFL_BRES(NK_, TN_)                                                                                                  \
    launch_bres<NK_, TN_, (JOIN && NK_ == 2 && TN_ == 4) ? 2 : (JOIN || (NK_ <= 2 && TN_ == 4) || (NK_ == 4 && TN_ == 2)) ? 4 : 8, JOIN>( \
        A, lda, B, ldb, M, N, C, ldc, addp, maskp, stream)
    int rc;
    switch (K) {
        case 64: rc = wide ? CFL_BRES(1, 4) : CFL_BRES(1, 2); break;
        case 128: rc = wide ? CFL_BRES(2, 4) : CFL_BRES(2, 2); break;
        case 256: rc = wide ? CFL_BRES(4, 4) : CFL_BRES(4, 2); break;
        case 512: rc = CFL_BRES(8, 2); break;
        case 1024: rc = CFL_BRES(16, 2); break;
        default: return 0;
    }
    return rc < 0 ? rc : 1;
#undef CFL_BRES
}

inline int& bres_min_m() {
    static int v = getenv("CFL_GEMM_BRES_MIN_M") ? atoi(getenv("CFL_GEMM_BRES_MIN_M")) : 32768;
    return v;
}

// dst[C][R] = src[R][C]^T for a small bf16 matrix (the 1x1 convolution weight [Co][Ci] -> [Ci][Co] that the data
// gradient needs K-contiguous): 64x64 tiles through LDS, 2-byte elements, coalesced on both sides.
__global__ __launch_bounds__(256) void cfl_transpose_bf16_kernel(const u16* __restrict__ src, int R, int C, u16* __restrict__ dst) {
    __shared__ u16 tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < R && c < C) ? src[(long long)r * C + c] : (u16)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (r < R && c < C) dst[(long long)c * R + r] = tile[tx][ty * 16 + i];
    }
}

// All weight transposes of a backward pass in ONE launch (66 1x1 and 30 3x3 weights at ResNet-101: as separate kernels they are
// a hundred small launches on the critical path of the backward).  meta[t] = {src, dst, R, C, first tile, tiles per row band,
// source row stride, destination row stride}: dst[c * ldd + r] = src[r * lds + c].  A dense [R, C] matrix has lds = C,
// ldd = R; one tap of a k x k weight stored [Co][k][k][Ci] (channels_last) is the matrix [Co, Ci] with lds = k k Ci, and its
// transpose goes to tap (k-1-kh, k-1-kw) of the [Ci][k][k][Co] image with ldd = k k Co -- the rotated, transposed weight
// that turns the FORWARD convolution kernel into the data gradient (ops._ConvSplitFn).  A workgroup finds its record by
// binary search on tile0 and transposes one 64x64 tile.
struct TrMeta { const u16* src; u16* dst; int R, C, tile0, tiles_c, lds, ldd; };

__global__ __launch_bounds__(256) void cfl_transpose_multi_kernel(const TrMeta* __restrict__ meta, int ntensors) {
    __shared__ u16 tile[64][66];
    int lo = 0, hi = ntensors - 1;
    while (lo < hi) {                                    // last record with tile0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= meta[mid].tile0) lo = mid; else hi = mid - 1;
    }
    const TrMeta m = meta[lo];
    const int local = blockIdx.x - m.tile0;
    const int r0 = (local / m.tiles_c) * 64, c0 = (local % m.tiles_c) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < m.R && c < m.C) ? m.src[(long long)r * m.lds + c] : (u16)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (r < m.R && c < m.C) m.dst[(long long)c * m.ldd + r] = tile[tx][ty * 16 + i];
    }
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
    if (variant == 90) {     // the B-resident kernel or a refusal (probes / tests; measured: no gain without the join operands)
        const int rc = try_bres<false>((const u16*)A, lda, (const u16*)B, ldb, M, N, K, Cc, ldc, nullptr, nullptr, stream);
        return rc < 0 ? rc : (rc == 0 ? CFL_ELIMIT : 0);
    }
    if (variant == 0) variant = N >= 128 ? 22 : 21;
    switch (variant) {
        case 44: return launch_nt<4, 4, 1>(Ao, Bo, M, N, Cc, ldc, stream);
        case 24: return launch_nt<2, 4, 1>(Ao, Bo, M, N, Cc, ldc, stream);      // 128 x 256: A read once when N <= 256
        case 42: return launch_nt<4, 2, 1>(Ao, Bo, M, N, Cc, ldc, stream);
        case 22: return launch_nt<2, 2, 2>(Ao, Bo, M, N, Cc, ldc, stream);
        case 21: return launch_nt<2, 1, 2>(Ao, Bo, M, N, Cc, ldc, stream);
        case 41: return launch_nt<4, 1, 2>(Ao, Bo, M, N, Cc, ldc, stream);
        default: return CFL_EINVAL;
    }
}

extern "C" int cfl_gemm_bf16_nt_stats_nblk(int M, int N, int K) { return bres_stats_nblk(M, N, K); }

extern "C" int cfl_gemm_bf16_nt_stats(const void* A, long long lda, const void* B, long long ldb, void* C, int M, int N, int K,
                                      float* pstat, void* stream_) {
    if (!A || !B || !C || !pstat || M <= 0 || N <= 0 || K <= 0) return CFL_EINVAL;
    if (bres_stats_nblk(M, N, K) == 0 || lda % 8 != 0 || ldb % 8 != 0 || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)pstat) & 15))
        return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const u16 *a = (const u16*)A, *b = (const u16*)B;
    switch (K) {
        case 64: return launch_bres_stats<1, 4, 4>(a, lda, b, ldb, M, N, (u16*)C, pstat, stream);
        case 128: return launch_bres_stats<2, 4, 2>(a, lda, b, ldb, M, N, (u16*)C, pstat, stream);      // deeper rings spill
        default: return launch_bres_stats<4, 4, 4>(a, lda, b, ldb, M, N, (u16*)C, pstat, stream);
    }
}

extern "C" int cfl_gemm_bf16_bres_min_m(int min_m) {
    const int old = bres_min_m();
    if (min_m >= 0) bres_min_m() = min_m;
    return old;
}

extern "C" int cfl_gemm_bf16_nt_join(const void* A, long long lda, const void* B, long long ldb, void* C, const void* add,
                                     const unsigned char* mask, int M, int N, int K, void* stream_) {
    if (!A || !B || !C || !add || !mask || M <= 0 || N <= 0 || K <= 0) return CFL_EINVAL;
    if (K % 64 != 0 || N % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0 ||
        (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)add) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    Opnd Ao{(const float*)A, lda / 2, M, K / 2, 1};
    Opnd Bo{(const float*)B, ldb / 2, N, K / 2, 1};
    if (M >= bres_min_m() && K <= 256) {     // below: too few 32-row tiles per wave (8 waves x 256 workgroups) for the streaming kernel
        const int rc = try_bres<true>((const u16*)A, lda, (const u16*)B, ldb, M, N, K, (u16*)C, N, (const u16*)add, mask, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (N >= 128) return launch_nt<2, 2, 2, true>(Ao, Bo, M, N, (u16*)C, N, stream, (const u16*)add, mask);
    return launch_nt<2, 1, 2, true>(Ao, Bo, M, N, (u16*)C, N, stream, (const u16*)add, mask);
}

extern "C" int cfl_transpose_bf16(const void* src, int R, int C, void* dst, void* stream_) {
    if (!src || !dst || R <= 0 || C <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    CFL_LAUNCH(K_TRANSPOSE, cfl_transpose_bf16_kernel, dim3(cfl_cdiv(C, 64), cfl_cdiv(R, 64)), dim3(256), 0, stream, (const u16*)src, R,
               C, (u16*)dst);
    return 0;
}

// meta: device array of ntensors records {src ptr, dst ptr, int R, C, tile0, tiles_c, lds, ldd} (40 bytes each, tile0
// ascending, tiles of record t = ceil(R/64) * tiles_c with tiles_c = ceil(C/64)); total_tiles = sum of tiles.
extern "C" int cfl_transpose_bf16_multi(const void* meta, int ntensors, int total_tiles, void* stream_) {
    if (!meta || ntensors <= 0 || total_tiles <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    CFL_LAUNCH(K_TRANSPOSE, cfl_transpose_multi_kernel, dim3(total_tiles), dim3(256), 0, stream, (const TrMeta*)meta, ntensors);
    return 0;
}
