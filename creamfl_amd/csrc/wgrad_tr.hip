// wgrad_tr.hip -- the weight gradient of a 1x1 convolution, dW[Co, Ci] = dY^T X, as a bf16 MFMA GEMM whose reduction runs along
// the SLOW axis of both operands (C[N1, N2] = A[M, N1]^T B[M, N2], M = N*H*W up to 802 816).
//
// Reference: the torchvision Bottleneck 1x1 convolutions inside src/networks/models/image_encoder.py:27-36 (backward of
// conv1 / conv3); the reference delegates to cuDNN, this build's default is MIOpen (a CK batched GEMM with fp32 atomics + a
// cast: 63 us alone / ~125 us inside the step at 14x14, 1024 x 256).
//
// Round 2's TN kernel transposed in REGISTERS (eight 16-byte loads + 32 v_perm per thread and K step, ds_write of the
// transposed image): 195 VALU instructions per wave and K step, matrix pipe 17 % busy, on par with the library.  Here the
// operands are never touched by the VALU:
//   * a stage = 64 rows of A (x 128 columns) and 64 rows of B, copied ROW-MAJOR, as they lie in memory, by LDS-DMA
//     (global_load_lds_dwordx4: one wave instruction = 4 rows x 256 bytes); rows past M and columns past N1 / N2 come from a
//     256-byte page of zeros, so ragged edges need no masking in the loop;
//   * the MFMA fragments (8 consecutive m for one column per lane) are read TRANSPOSED out of that image with
//     ds_read_b64_tr_b16: the 16 lanes of a group address a [4 rows x 16 columns] block and each receives one column of it
//     (probed on gfx950: out[j] = E[4 j + (i >> 2)][i & 3]); two reads fill one operand of v_mfma_f32_32x32x16_bf16.  The 16-byte
//     pieces of a row are XOR-swizzled by ((row & 3) << 2) -- applied on the SOURCE side of the DMA -- so that the four rows a
//     transposing read touches fall into four different 64-byte bank windows (conflict-free);
//   * 128 x 128 tiles, 8 waves = 2 x 2 output quadrants x 2 K halves of every stage (two waves per SIMD), a ring of 4 LDS
//     stages with three in flight (counted vmcnt waits), one workgroup per CU; split-K over M with fp32 partial
//     tiles and a fixed-order reduction that also casts (deterministic -- the library's atomics are not); blocks are ordered
//     split-major on an XCD so that the tiles sharing an M range share its slabs in that XCD's L2.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];      // zero-initialised device memory

constexpr int BM = 128, BN = 128, KS = 64;                  // tile of C, rows of m per stage
constexpr int OPND = KS * 256;                              // bytes of one operand tile in a stage (64 rows x 128 bf16)
constexpr int STAGE = 2 * OPND;                             // 32 KB
constexpr int NS = 4;                                       // LDS stages in the ring (three in flight)

__device__ __forceinline__ unsigned f2bf_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// One operand tile of a stage: rows m0 .. m0 + 63 of src[M, ld], columns c0 .. c0 + 127, into `dst` (16 KB, row-major with
// swizzled 16-byte pieces).  16 wave instructions; wave w of the 8 issues instructions w and w + 8.
__device__ __forceinline__ void stage_tile(const u16* __restrict__ src, long long ld, long long M, int Ctot, long long m0, int c0,
                                           char* dst, int w, int lane) {
    const int r4 = lane >> 4, pp = lane & 15;               // row within the instruction's 4 rows, physical piece
    const int lp = pp ^ (r4 << 2);                          // logical piece stored there (row & 3 == r4: instructions start at row % 4 == 0)
    const int col = c0 + lp * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ins = w + 8 * j;
        const long long m = m0 + ins * 4 + r4;
        const void* p = (m < M && col < Ctot) ? (const void*)(src + m * ld + col) : (const void*)g_zero_page;
        __builtin_amdgcn_global_load_lds((glb_vptr)p, (lds_vptr)(dst + ins * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 frag_tr(const char* p) {   // 8 consecutive m of this lane's column: rows +0..3 and +4..7
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * 256));
    union { s16x4 h[2]; bf16x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}

// grid: ntiles * nsplit workgroups; virtual id v (XCD-contiguous): split = v / ntiles, tile = v % ntiles.
// 8 waves = 4 output quadrants (64 x 64 each) x 2 K halves: waves 0-3 take the first 32 rows of every 64-row stage, waves 4-7
// the other 32 -- two waves per SIMD (one's LDS reads and MFMA dependencies hide behind the other's MFMAs; with four waves a
// stage took ~2200 cycles for 512 cycles of MFMA: every K step waited for its own transposing reads) without a second
// workgroup's LDS or a second set of split-K partials; the two halves meet through LDS at the end.
__global__ __launch_bounds__(512, 1) void cfl_wgrad_tr_kernel(const u16* __restrict__ A, long long lda, const u16* __restrict__ B,
                                                              long long ldb, long long M, int N1, int N2, int stages_per_split,
                                                              float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nt2 = (N2 + BN - 1) / BN, ntiles = ((N1 + BM - 1) / BM) * nt2;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int split = v / ntiles, tile = v % ntiles;
    const int row0 = (tile / nt2) * BM, col0 = (tile % nt2) * BN;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wq = w & 3, kh = w >> 2, wr = wq >> 1, wc = wq & 1;
    const long long nst = (M + KS - 1) / KS;
    const long long sbeg = (long long)split * stages_per_split;
    long long send = sbeg + stages_per_split;
    if (send > nst) send = nst;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    // fragment addresses inside an operand tile: lane (g = lane >> 4, p = lane & 15) reads rows 8 (g >> 1) + (p >> 2) [+ 4],
    // columns wave_col0 + 32 t + 16 (g & 1) + 4 (p & 3) .. + 3; piece = col >> 3, swizzled by (row & 3) = p >> 2
    const int g = lane >> 4, p = lane & 15;
    int offA[2], offB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ca = wr * 64 + 32 * t + 16 * (g & 1) + 4 * (p & 3);
        const int cb = wc * 64 + 32 * t + 16 * (g & 1) + 4 * (p & 3);
        const int row = 32 * kh + 8 * (g >> 1) + (p >> 2);
        offA[t] = row * 256 + ((((ca >> 3) ^ ((p >> 2) << 2)) & 15) << 4) + (ca & 7) * 2;
        offB[t] = OPND + row * 256 + ((((cb >> 3) ^ ((p >> 2) << 2)) & 15) << 4) + (cb & 7) * 2;
    }
    // Ring of NS stages, NS - 1 in flight: the wait in front of a stage is a counted s_waitcnt (4 LDS-DMA instructions per wave
    // and stage; the two younger stages stay outstanding) + a bare s_barrier, which also frees the buffer the next DMA
    // overwrites.  One workgroup per CU (128 KB of LDS): the ring covers the round trip to HBM.
    auto issue = [&](long long st) {
        char* dst = lds + ((st - sbeg) & (NS - 1)) * STAGE;
        stage_tile(A, lda, M, N1, st * KS, row0, dst, w, lane);
        stage_tile(B, ldb, M, N2, st * KS, col0, dst + OPND, w, lane);
    };
    for (long long st = sbeg; st < sbeg + NS - 1 && st < send; ++st) issue(st);
    for (long long st = sbeg; st < send; ++st) {
        const long long rem = send - 1 - st;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (st + NS - 1 < send) issue(st + NS - 1);
        const char* cur = lds + ((st - sbeg) & (NS - 1)) * STAGE;
        // all 16 transposing reads of this wave's two K steps first, then the 8 MFMAs: the LDS latency is paid once per stage
        bf16x8 a[2][2], b[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[ks][t] = frag_tr(cur + offA[t] + ks * 16 * 256);
                b[ks][t] = frag_tr(cur + offB[t] + ks * 16 * 256);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][m], b[ks][n], acc[m][n], 0, 0, 0);
    }
    // the two K halves of a quadrant meet through LDS (the ring is free after a barrier): 4 x 16 KB of fp32
    __syncthreads();
    float* xch = reinterpret_cast<float*>(lds) + wq * 4096;
    if (kh == 1) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((m * 2 + n) * 16 + r) * 64 + lane] = acc[m][n][r];
    }
    __syncthreads();
    if (kh == 1) return;
    float* out = part + (long long)split * N1 * N2;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int j = col0 + acc_col<2>(wc, n, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = row0 + acc_row<2>(wr, m, r, lane);
                if (i < N1 && j < N2) out[(long long)i * N2 + j] = acc[m][n][r] + xch[((m * 2 + n) * 16 + r) * 64 + lane];
            }
        }
}

// C = sum over splits (fixed order), written in the weight's dtype.  4 elements per thread, 8 independent 16-byte loads in flight.
__global__ __launch_bounds__(256) void cfl_wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, long long n, void* out,
                                                               int out_bf16) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (long long)(k + u) * n + i);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(part + (long long)k * n + i);
    if (out_bf16) {
        reinterpret_cast<unsigned*>(out)[i / 2] = f2bf_rne(s[0]) | (f2bf_rne(s[1]) << 16);
        reinterpret_cast<unsigned*>(out)[i / 2 + 1] = f2bf_rne(s[2]) | (f2bf_rne(s[3]) << 16);
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + i) = s;
    }
}

struct WgPlan { int ntiles, nsplit, sps; };
inline WgPlan wg_plan(long long M, int N1, int N2) {
    WgPlan p;
    p.ntiles = cfl_cdiv(N1, BM) * cfl_cdiv(N2, BN);
    const long long nst = (M + KS - 1) / KS;
    long long want = 256 / p.ntiles;                               // one workgroup per CU (its stage ring holds 128 KB of LDS)
    const long long cap = (16ll << 20) / ((long long)N1 * N2 * 4);   // partial tiles: at most 16 MB written + read back
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (want > nst) want = nst;
    p.sps = (int)((nst + want - 1) / want);
    p.nsplit = (int)((nst + p.sps - 1) / p.sps);
    return p;
}

}  // namespace

extern "C" size_t cfl_gemm_bf16_tn_ws_bytes(long long M, int N1, int N2) {
    if (M <= 0 || N1 <= 0 || N2 <= 0) return 256;
    return cfl_align256((size_t)wg_plan(M, N1, N2).nsplit * N1 * N2 * sizeof(float));
}

extern "C" int cfl_gemm_bf16_tn(const void* A, long long lda, const void* B, long long ldb, void* C, int c_bf16, long long M, int N1,
                                int N2, void* ws, void* stream_) {
    if (!A || !B || !C || !ws || M <= 0 || N1 <= 0 || N2 <= 0) return CFL_EINVAL;
    if (N1 % 8 != 0 || N2 % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0 || (N1 * (long long)N2) % 4 != 0 ||
        (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15))
        return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    const WgPlan p = wg_plan(M, N1, N2);
    constexpr size_t LDS = (size_t)NS * STAGE;
    CFL_SET_LDS(cfl_wgrad_tr_kernel, LDS);
    CFL_LAUNCH(K_WGRAD, cfl_wgrad_tr_kernel, dim3(p.ntiles * p.nsplit), dim3(512), LDS, stream, (const u16*)A, lda, (const u16*)B,
               ldb, M, N1, N2, p.sps, (float*)ws);
    const long long n = (long long)N1 * N2;
    CFL_LAUNCH(K_WGRAD_REDUCE, cfl_wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)ws,
               p.nsplit, n, C, c_bf16);
    return 0;
}
