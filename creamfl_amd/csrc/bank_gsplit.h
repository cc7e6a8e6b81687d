// bank_gsplit.h -- rows A3 + A4, round 3: the bank pass on a PRE-SPLIT bank image, 16-row bank slots streamed through LDS.
// Included by bank_attn.hip (inside its anonymous namespace, after the finish kernel it shares).
//
// Reference: src/algorithms/ClientTrainer.py:388,398-419, src/algorithms/MMClientTrainer.py:173-206.
//
// Why a second kernel.  Round 2 (cfl_bank_attn_kernel in bank_attn.hip) gives every workgroup 128 feature rows, so a client
// batch of 128 is ONE row group and the bank must be cut into 256 splits to fill the chip: 256 x [128, D] fp32 partial
// gradients = 33.5 MB written and read back for a 51 MB bank (PMC: 2.4x the algorithmic traffic, the finish kernel a quarter
// of the time); it keeps a transposed LDS copy of every chunk for the gradient GEMM, which is what stops it at D = 256; and
// the fp32 -> bf16 hi/lo conversion of every bank row is repeated by every step although the bank is frozen for the whole
// round (ClientTrainer.py:369-372: the global features are fixed while the client trains).  Here
//   * the bank is converted ONCE per round into an image of 16-row slots (cfl_bank_image_build): per slot
//     [plane hi|lo][16 rows][DP bf16], 16-byte pieces XOR-swizzled -- byte for byte the LDS image the kernel computes on,
//     and byte for byte the size of the fp32 bank.  A step streams it with plain 16-byte loads: no conversion VALU, and the
//     transposed copy is gone (the gradient GEMM reads its operand with ds_read_b64_tr_b16), so D goes to 768;
//   * 32 feature rows per workgroup => 4 row groups for a batch of 128 => 64 bank splits: 64 x [128, D] partials = 8.4 MB
//     (was 33.5).  The row groups of a split are neighbours on one XCD (block ids 8 apart), so HBM sees the image once and
//     the others hit that XCD's L2 (PMC: 51.2 MB fetched for a 51.2 MB image with 4 row groups).
// Measured and dropped on the way (B = 128, M = 50 000, D = 256; docs/history/DESIGN_r1-r4.md section 4.3): 32 rows per WAVE (F and O = D/2
// registers each, 4 waves, one per SIMD, private slots, 32x32x16 MFMA for the gradient with the probabilities moved between
// the two accumulator layouts by v_permlane16_swap) -- correct, but 46 us: with 256 registers gone there is room for half a
// slot of prefetch, and every slot waited twice for a load issued half an iteration earlier (3.3 us per slot, 0.7 of MFMA).
#pragma once

namespace gs {

constexpr int FB = 32;        // feature rows per workgroup
constexpr int SG = 16;        // bank rows per image slot
constexpr int NW = 4;         // waves per workgroup (one per SIMD)

__host__ __device__ inline int img_dp(int D) { return D <= 128 ? 128 : (D <= 256 ? 256 : (D <= 512 ? 512 : 768)); }
__host__ __device__ inline size_t img_slot_bytes(int DP) { return (size_t)64 * DP; }     // 2 planes x 16 rows x DP x 2 B
__host__ __device__ inline size_t img_bytes(int M, int D) { return (size_t)((M + SG - 1) / SG) * img_slot_bytes(img_dp(D)); }

// 16-byte piece s of row g lives at piece s ^ swz(g): bits [3:2] = g & 3 (the four rows of a transposing read fall into
// four different 64-byte bank windows), bits [1:0] = L[g >> 2], L = {0, 2, 3, 1} (the 16 lanes of every ds_read_b128
// service group -- 8 rows at piece s, 8 at piece s ^ 1 -- hit 16 different pieces).  docs/history/tools/lds_swizzle_check.py enumerates
// every access pattern of both kernels against the LDS service groups: 0 conflict cycles.
__host__ __device__ inline int img_swz(int g) { return ((g & 3) << 2) | ((0x78 >> (2 * ((g >> 2) & 3))) & 3); }
__host__ __device__ inline int img_off(int DP, int plane, int g, int piece) {
    return plane * (SG * DP * 2) + g * (DP * 2) + ((piece ^ img_swz(g)) << 4);
}

// ---- image build: one thread = one 16-byte piece (8 columns) of one bank row, both planes ------------------------------
__global__ __launch_bounds__(256) void cfl_bank_image_kernel(const float* __restrict__ G, int M, int D, int DP,
                                                          char* __restrict__ img) {
    const int SL = DP / 8;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nslots = (M + SG - 1) / SG;
    if (e >= nslots * SG * SL) return;
    const int piece = (int)(e % SL);
    const int g = (int)((e / SL) % SG);
    const long long c = e / ((long long)SL * SG);
    const long long row = c * SG + g;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = 8 * piece + j;
        v[j] = (row < M && d < D) ? G[row * D + d] : 0.f;
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        hi[j] = h;
        lo[j] = (__bf16)(v[j] - (float)h);
    }
    char* slot = img + c * (long long)img_slot_bytes(DP);
    *reinterpret_cast<bf16x8*>(slot + img_off(DP, 0, g, piece)) = hi;
    *reinterpret_cast<bf16x8*>(slot + img_off(DP, 1, g, piece)) = lo;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// 8 bank rows (4 q .. 4 q + 3 for q = 0, 1 relative to r0) of one column per lane, by two transposing reads: lane p of a
// 16-lane group addresses row (p >> 2), columns 4 (p & 3) .. + 3 of the group's [4 rows x 16 columns] block and receives
// column p of it (probed on gfx950: out[j] = E[4 j + (i >> 2)][i & 3], docs/history/tools/hip/tr_probe.hip).
__device__ __forceinline__ bf16x8 tr_read8(const char* p0, const char* p1) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
    union { s16x4 h[2]; bf16x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}

// two bf16 (lo half = first) packed into one register
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    union { __bf16 h[2]; unsigned u; } x;
    x.h[0] = (__bf16)a; x.h[1] = (__bf16)b;
    return x.u;
}
__device__ __forceinline__ float bf16_lo_f(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi_f(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// max over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the lanes that share a feature row of a 16x16 accumulator) on the VALU:
// v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane rows between two registers (same value in both: afterwards
// one holds the even rows twice, the other the odd rows twice).  __shfl_xor compiles to ds_bpermute_b32 -- an LDS-queue
// round trip (lgkmcnt) in the middle of the serial soft-max chain, twice per slot.
__device__ __forceinline__ float row_max4(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned w = __float_as_uint(m);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// ---- the streaming kernel: 16 feature rows per wave, NGG bank-slot streams per workgroup ----------------------------------
// The arrangement:
//   * a wave owns 16 feature rows (F and O = D/4 registers each) -- at D <= 256 two waves per SIMD, eight per workgroup;
//   * the workgroup's waves form NFG feature groups x NGG slot streams: stream gg walks bank slots c0 + gg, c0 + gg + NGG, ...
//     through its own double-buffered LDS slot pair, shared by the NFG waves of the stream (each stages 1/NFG of a slot);
//     32 rows per workgroup (NFG = 2, NGG = 4) at D <= 256,
//     and D = 512 / 768 run as NFG = 4, NGG = 1 (one wave per SIMD, 64 rows per workgroup);
//   * the freed registers hold LA = 2 staging sets: the loads of slot i + 2 are issued at the top of iteration i and written
//     to LDS at the end of iteration i + 1 -- a two-iteration window (~1.7 us) for every byte, two waves per SIMD to fill
//     whatever stall is left, one barrier per iteration;
//   * gradient GEMM on v_mfma_f32_16x16x32_bf16 with the two bf16 PLANES folded into the contraction index so that a 16-row
//     slot still fills k = 32:  k-slot (kg, j < 4) = (row 4 kg + j, hi plane), (kg, j >= 4) = (row 4 kg + j - 4, lo plane) --
//     exactly the four rows whose probabilities lane group kg already holds in its S accumulator (no exchange at all):
//       O^T += [Gh | Gl]^T [ph | ph]   (hi.hi + lo.hi in ONE MFMA)      and      O^T += [Gh | Gl]^T [pl | 0]   (hi.lo),
//     4 MFMAs per 32 bank rows and column tile instead of 3, the A fragment (one transposing read per plane) shared by both;
//   * the NGG streams of a feature group merge their (max, sum, O) through LDS at the end.
//   * beyond D = 256 the per-wave state is cut by NDS = 2: the two waves of a PAIR share 16 feature rows, each takes half of
//     the contraction steps of the logits (its half of F: D/8 + D/8 registers) and half of the gradient columns (D/8
//     registers of O); the partial logits -- 4 floats per lane -- cross through LDS (one extra barrier per slot), so no MFMA
//     is issued twice and every wave reads only its half of the slot.  D = 512 and 768: 4 pairs = 8 waves, two per SIMD,
//     64 rows per workgroup (at D = 768 F + O = 192 registers per wave; with the slots arriving by LDS-DMA and read bursts
//     of 4 the kernel fits 249 registers without spilling.  Measured on the way at D = 768, B = 128, M = 50 000: F 192 + O
//     192 in ONE wave: 166 registers spilled into the loop, 442 us; 2 pairs with one wave per SIMD: 105 us; 4 pairs: 73 us).
template <int DT, int NGG, int NWAVE, int NDS>
struct StSmem {
    static constexpr int DP = 32 * DT;
    static constexpr int SLOT = 64 * DP;
    static constexpr int STREAMS = NGG * 2 * SLOT;
    static constexpr int ROWB = DP * 4 + 16;                              // merge image: [16 rows][DP floats + 16 B pad] per wave
    static constexpr int MERGE = NGG > 1 ? NWAVE * 16 * ROWB : 0;
    static constexpr int BODY = STREAMS > MERGE ? STREAMS : MERGE;
    static constexpr int ML = NWAVE * 16 * 2 * 4;                         // (max, sum) per wave and row | logit exchange (NDS > 1)
    static constexpr int XS = NDS > 1 ? NWAVE * 1024 : 0;
    static constexpr int TOTAL = BODY + (ML > XS ? ML : XS);
};

template <int DT, int NGG, int NWAVE, int NDS, int LA, bool GRAD>
__global__ __launch_bounds__(64 * NWAVE, NWAVE / 4) void cfl_bank_stream_kernel(const float* __restrict__ F, const char* __restrict__ img,
                                                                              int B, int M, int D, float sc2, int S, int RG,
                                                                              float* __restrict__ part_m, float* __restrict__ part_l,
                                                                              float* __restrict__ part_o) {
    static_assert(NDS == 1 || NGG == 1, "column-split pairs run on a single slot stream");
    constexpr int DP = 32 * DT, NSW = NWAVE / NGG, NFG = NSW / NDS, FR = 16 * NFG;
    constexpr int KSW = DT / NDS, NDTW = 2 * DT / NDS;                   // contraction steps / gradient column tiles per wave
    static_assert(NDS == 1 || (KSW % 4 == 0 && NDTW % 8 == 0), "a wave's share must start on a 256-byte window");
    constexpr int STRIDE = NSW * 1024, NPT = (64 * DP) / STRIDE;         // 16-byte loads per thread and slot
    constexpr bool TIGHT = (DT == 24 && NWAVE == 8);                     // 2 waves per SIMD at D = 768: shorter bursts
    constexpr int RB = TIGHT ? 4 : (KSW % 8 == 0 ? 8 : (KSW % 6 == 0 ? 6 : 4));   // contraction steps per read burst of the logits block
    constexpr int GW = (NDTW >= 8 && LA <= 1 && !TIGHT) ? 8 : 4;                   // column tiles per read burst of the gradient block
    static_assert(KSW % RB == 0 && NDTW % GW == 0, "bursts must tile the wave's share");
    using SM = StSmem<DT, NGG, NWAVE, NDS>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gg = w % NGG, sw = w / NGG;                                // stream, wave of the stream
    const int dh = sw % NDS, fg = sw / NDS;                              // column half, feature group
    const int f16 = lane & 15, kg = lane >> 4;
    const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3;
    const int rg = j8 % RG, x = (j8 / RG) * 8 + xcd;
    const int nslot = (M + SG - 1) / SG;
    const int base = nslot / S, extra = nslot % S;
    const int c0 = x * base + (x < extra ? x : extra);
    const int nmine = base + (x < extra ? 1 : 0);                        // slots of this split
    const int niter = (nmine + NGG - 1) / NGG;
    const bool wave_live = rg * FR + 16 * fg < B;
    char* sbuf = lds + gg * 2 * SM::SLOT;                                // this stream's two slot buffers
    const int toff = (sw * 64 + lane) * 16;                              // this thread's 16 bytes of every STRIDE-byte stripe
    const char* sbase = img + toff;
    auto slot_src = [&](int it) { return sbase + (size_t)(c0 + gg + it * NGG) * SM::SLOT; };
    auto has_slot = [&](int it) { return gg + it * NGG < nmine; };

    // LA = 0: the slots travel by LDS-DMA (global_load_lds_dwordx4: 64 x 16 B per wave instruction straight into the LDS
    // buffer, lane-linear -- the image is stored in LDS order for exactly this): no staging registers, no ds_write (the
    // 64 KB per iteration a workgroup pushes through ds_write_b128 occupy the LDS write path for ~800 cycles of the ~3800 an
    // iteration takes), issued at the top of an iteration and complete at its closing barrier.
    // LA = -1 (round 6): the same LDS-DMA issued from INLINE ASSEMBLY.  The compiler orders every LDS read it cannot tell apart from
    // the transfer's target behind every LDS-DMA it knows of: with the builtin it put s_waitcnt vmcnt(0) in front of the first
    // TRANSPOSING read of the gradient block (and, at D > 256, in front of the logit-exchange barrier) -- the next slot had to land in
    // the MIDDLE of the iteration that issued it, not at its closing barrier.  What the compiler does not see it does not wait for;
    // the wait is the explicit vmcnt(0) in front of the closing barrier.  M0 (the LDS base of a transfer) is saved and restored.
    constexpr bool DMA = (LA <= 0), ASMDMA = (LA < 0);
    constexpr int NSET = DMA ? 1 : LA;
    u32x4v stage[NSET][DMA ? 1 : NPT];
    const unsigned sbuf_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sbuf;
    auto dma_slot = [&](int it, int b) {
        const char* src = slot_src(it);
        if (ASMDMA) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(sbuf_a + b * SM::SLOT + sw * 1024);
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src + j * STRIDE), "s"(dst + j * STRIDE) : "memory");
            }
            return;
        }
        char* dst = sbuf + b * SM::SLOT + sw * 1024;          // this wave's 1 KB of every STRIDE-byte stripe (lane l: + 16 l)
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            __builtin_amdgcn_global_load_lds((glb_vptr)(src + j * STRIDE), (lds_vptr)(dst + j * STRIDE), 16, 0, 0);
    };
    // prologue: slot 0 straight into buffer 0; with LA = 2 the loads of slot 1 are left in flight in set 1
    if (DMA) {
        if (has_slot(0)) dma_slot(0, 0);
    } else {
        if (has_slot(0)) {
#pragma unroll
            for (int j = 0; j < NPT; ++j) stage[0][j] = *reinterpret_cast<const u32x4v*>(slot_src(0) + j * STRIDE);
        }
        if (LA == 2 && has_slot(1)) {
#pragma unroll
            for (int j = 0; j < NPT; ++j) stage[NSET - 1][j] = *reinterpret_cast<const u32x4v*>(slot_src(1) + j * STRIDE);
        }
    }
    // this wave's share of its 16 feature rows: contraction steps dh KSW .. + KSW - 1
    bf16x8 fh[KSW], fl[KSW];
    {
        const int fr = rg * FR + 16 * fg + f16;
        const float* fp = F + (long long)(fr < B ? fr : B - 1) * D;
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            const int k0 = 32 * (dh * KSW + ks) + 8 * kg;
            const bool ok0 = fr < B && k0 < D, ok1 = fr < B && k0 + 4 < D;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(fp + (k0 < D ? k0 : 0));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(fp + (k0 + 4 < D ? k0 + 4 : 0));
            bf16x4 h0, l0, h1, l1;
            split4(v0, ok0, h0, l0);
            split4(v1, ok1, h1, l1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { fh[ks][e] = h0[e]; fh[ks][4 + e] = h1[e]; fl[ks][e] = l0[e]; fl[ks][4 + e] = l1[e]; }
        }
    }
    f32x4 O[GRAD ? NDTW : 1];
#pragma unroll
    for (int i = 0; i < (GRAD ? NDTW : 1); ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float run_m = -INFINITY, run_l = 0.f;
    if (!DMA && has_slot(0)) {
#pragma unroll
        for (int j = 0; j < NPT; ++j) *reinterpret_cast<u32x4v*>(sbuf + j * STRIDE + toff) = stage[0][j];
    }
    if (ASMDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // piece addresses: a few per-lane bases + compile-time multiples of 256 (the swizzle permutes within 256-byte windows);
    // a column-split wave starts dh KSW x 64 / dh NDTW x 32 bytes into the row (whole windows)
    int la[4];                                                   // logits operand: row g = f16, piece 4 ks + kg
#pragma unroll
    for (int j = 0; j < 4; ++j) la[j] = f16 * (DP * 2) + dh * (KSW * 64) + (((4 * j + kg) ^ img_swz(f16)) << 4);
    int ta[8];                                                   // gradient operand: rows 4 kg + (p >> 2), piece 2 dt + ((p & 3) >> 1)
    {
        const int tr_row = 4 * kg + (f16 >> 2);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            ta[j] = tr_row * (DP * 2) + dh * (NDTW * 32) + 8 * (f16 & 1) + (((2 * j + ((f16 & 3) >> 1)) ^ img_swz(tr_row)) << 4);
    }
    f32x4* xs = reinterpret_cast<f32x4*>(lds + SM::BODY);        // NDS > 1: [wave][lane] partial logits

    // one iteration: `ld` receives the loads of slot it + LA, `wr` (holding slot it + 1) goes to the other buffer at the end
    auto iteration = [&](int it, u32x4v (&ld)[DMA ? 1 : NPT], u32x4v (&wr)[DMA ? 1 : NPT]) {
        if (DMA) {
            if (has_slot(it + 1)) dma_slot(it + 1, (it + 1) & 1);
        } else if (has_slot(it + LA)) {
#pragma unroll
            for (int j = 0; j < NPT; ++j) ld[j] = *reinterpret_cast<const u32x4v*>(slot_src(it + LA) + j * STRIDE);
        }
        const char* sb = sbuf + (it & 1) * SM::SLOT;
        const bool work = wave_live && has_slot(it);
        f32x4 sa = {0.f, 0.f, 0.f, 0.f};
        if (work) {
            f32x4 sbb = sa, sc = sa;
            // bank fragments in bursts of RB contraction steps: all reads of a burst are issued before its first MFMA, so the
            // LDS latency is paid once per burst (one read at a time -- what the compiler does when registers are short --
            // leaves the matrix pipe 30 % busy: every MFMA pair waits for its own ds_read)
#pragma unroll
            for (int k0 = 0; k0 < KSW; k0 += RB) {
                bf16x8 ah[RB], al[RB];
#pragma unroll
                for (int e = 0; e < RB; ++e) {
                    const int off = la[(k0 + e) & 3] + ((k0 + e) >> 2) * 256;
                    ah[e] = *reinterpret_cast<const bf16x8*>(sb + off);
                    al[e] = *reinterpret_cast<const bf16x8*>(sb + SG * DP * 2 + off);
                }
#pragma unroll
                for (int e = 0; e < RB; ++e) {
                    sa = MFMA16(ah[e], fh[k0 + e], sa);
                    sbb = MFMA16(al[e], fh[k0 + e], sbb);
                    sc = MFMA16(ah[e], fl[k0 + e], sc);
                }
            }
            sa += sbb + sc;
        }
        if (NDS > 1) {                                           // the pair's partial logits meet (all waves take the barrier)
            xs[w * 64 + lane] = sa;
            // a bare barrier behind this wave's own LDS write (round 6): __syncthreads() is a fence the compiler drains vmcnt to 0
            // for, i.e. the LDS-DMA of the NEXT slot, issued at the top of this iteration, had to land here -- in the middle of the
            // iteration -- instead of at its closing barrier; nobody reads that buffer before the closing barrier
            if (ASMDMA) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else __syncthreads();
#pragma unroll
            for (int o = 1; o < NDS; ++o) sa += xs[(w - dh + (dh + o) % NDS) * 64 + lane];
        }
        if (work) {
            const int g0 = (c0 + gg + it * NGG) * SG + 4 * kg;
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = sa[r] * sc2;
                if (g0 + r >= M) y = -INFINITY;
                sa[r] = y;
                mx = fmaxf(mx, y);
            }
            mx = row_max4(mx);
            if (__any(mx > run_m + RESCALE_THR)) {
                const float mn = fmaxf(run_m, mx);
                const float alpha = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(run_m - mn);
                run_l *= alpha;
                if (GRAD) {
#pragma unroll
                    for (int dt = 0; dt < NDTW; ++dt) O[dt] *= alpha;
                }
                run_m = mn;
            }
            float pv[4], ls = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[r] = run_m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sa[r] - run_m);
                ls += pv[r];
            }
            run_l += ls;
            if (GRAD) {
                union { unsigned u[4]; bf16x8 v; } b1, b2;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned ph = pack_bf16(pv[2 * h], pv[2 * h + 1]);
                    b1.u[h] = ph; b1.u[2 + h] = ph;                                                      // [ph | ph]
                    b2.u[h] = pack_bf16(pv[2 * h] - bf16_lo_f(ph), pv[2 * h + 1] - bf16_hi_f(ph));       // [pl | 0]
                    b2.u[2 + h] = 0u;
                }
                const char* sl1 = sb + SG * DP * 2;
#pragma unroll
                for (int d4 = 0; d4 < NDTW; d4 += GW) {
                    bf16x8 ga[GW];
#pragma unroll
                    for (int e = 0; e < GW; ++e) {
                        const int o = ta[(d4 + e) & 7] + ((d4 + e) >> 3) * 256;
                        ga[e] = tr_read8(sb + o, sl1 + o);           // [4 rows hi | the same 4 rows lo] of column 16 dt + f16
                    }
#pragma unroll
                    for (int e = 0; e < GW; ++e) O[d4 + e] = MFMA16(ga[e], b1.v, O[d4 + e]);
#pragma unroll
                    for (int e = 0; e < GW; ++e) O[d4 + e] = MFMA16(ga[e], b2.v, O[d4 + e]);
                }
            }
        }
        if (!DMA && has_slot(it + 1)) {
            char* nb = sbuf + ((it + 1) & 1) * SM::SLOT;
#pragma unroll
            for (int j = 0; j < NPT; ++j) *reinterpret_cast<u32x4v*>(nb + j * STRIDE + toff) = wr[j];
        }
        if (ASMDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // also the landing point of this iteration's LDS-DMA (vmcnt)
    };
    for (int it = 0; it < niter; it += NSET) {
        iteration(it, stage[0], stage[NSET - 1]);
        if (NSET == 2 && it + 1 < niter) iteration(it + 1, stage[NSET - 1], stage[0]);
    }

    run_l += __shfl_xor(run_l, 16, 64);
    run_l += __shfl_xor(run_l, 32, 64);
    const int f = rg * FR + 16 * fg + f16;
    if (NGG == 1) {
        if (!wave_live) return;
        if (kg == 0 && dh == 0) {
            const size_t o = (size_t)(f >> 7) * S * BR + (size_t)x * BR + (f & (BR - 1));
            part_m[o] = run_m;
            part_l[o] = run_l;
        }
        if (GRAD) {
            float* po = part_o + ((size_t)f * S + x) * DP + dh * (NDTW * 16) + 4 * kg;
#pragma unroll
            for (int dt = 0; dt < NDTW; ++dt) store_wt_x4(po + 16 * dt, O[dt]);
        }
        return;
    }
    // ---- merge of the NGG streams of every feature group through LDS (the slot buffers are free: last barrier above)
    float* ml = reinterpret_cast<float*>(lds + SM::BODY);
    if (kg == 0) { ml[(w * 16 + f16) * 2] = run_m; ml[(w * 16 + f16) * 2 + 1] = run_l; }
    if (GRAD) {
        char* ro = lds + w * 16 * SM::ROWB + f16 * SM::ROWB + 16 * kg;
#pragma unroll
        for (int dt = 0; dt < NDTW; ++dt) *reinterpret_cast<f32x4*>(ro + 64 * dt) = O[dt];
    }
    __syncthreads();
    // waves of a feature group: fg NGG + 0 .. NGG - 1 in wave order  (w = (fg) * NGG + gg when NDS = 1)
    const int t = threadIdx.x;
    auto wave_of = [&](int g2, int v) { return g2 * NGG + v; };      // gg = w % NGG, fg = w / NGG
    if (t < FR) {                                               // row t of the workgroup: feature group t / 16, row t % 16
        const int g2 = t >> 4, r = t & 15;
        float mm = -INFINITY;
#pragma unroll
        for (int v = 0; v < NGG; ++v) mm = fmaxf(mm, ml[(wave_of(g2, v) * 16 + r) * 2]);
        float L = 0.f;
#pragma unroll
        for (int v = 0; v < NGG; ++v) {
            const float mv = ml[(wave_of(g2, v) * 16 + r) * 2];
            L = fmaf(mm == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mv - mm), ml[(wave_of(g2, v) * 16 + r) * 2 + 1], L);
        }
        const int fo = rg * FR + t;
        if (fo < B) {
            const size_t o = (size_t)(fo >> 7) * S * BR + (size_t)x * BR + (fo & (BR - 1));
            part_m[o] = mm;
            part_l[o] = L;
        }
    }
    if (GRAD) {
        for (int e = t; e < FR * (DP / 4); e += 64 * NWAVE) {
            const int row = e / (DP / 4), d4 = e % (DP / 4);          // consecutive lanes = consecutive 16-byte pieces of one row
            const int g2 = row >> 4, r = row & 15;
            float mm = -INFINITY;
#pragma unroll
            for (int v = 0; v < NGG; ++v) mm = fmaxf(mm, ml[(wave_of(g2, v) * 16 + r) * 2]);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < NGG; ++v) {
                const int wv = wave_of(g2, v);
                const float wgt = mm == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ml[(wv * 16 + r) * 2] - mm);
                acc += *reinterpret_cast<const f32x4*>(lds + (wv * 16 + r) * SM::ROWB + 16 * d4) * wgt;
            }
            const int fo = rg * FR + row;
            if (fo < B) store_wt_x4(part_o + ((size_t)fo * S + x) * DP + 4 * d4, acc);
        }
    }
}

// ---- the wide-batch forward on v_mfma_f32_32x32x16_bf16 (row A5, con_w; D <= 256) ------------------------------------------
// con_w is the A3 problem with B = M = 50 000 feature rows and no gradient: one side of the product can live in REGISTERS.  The 8
// waves of a workgroup hold 256 rows of V (32 per wave, pre-scaled and pre-split: D/2 registers) and stream the bank image through
// ONE slot stream, so the image passes through a CU once per 256 rows: 196 x 51 MB = 10 GB through the L2 -> LDS path where the 128 x
// 128 tile GEMM of bank.hip moves 39 GB (its bound: ~4.3 ms at the 8-9 TB/s that path sustains).  A wave multiplies 32 bank rows
// (two 16-row slots) by its 32 feature rows with 32 x 32 x 16 MFMAs (the stream kernel's 16 x 16 x 32 arrangement with two row
// groups per wave was built first: same time, twice the matrix instructions, two cross-lane swaps per 16 rows): 32 cycles and ~5
// free issue slots beside each, and the
// accumulator layout (lane = feature row, 16 registers = 16 of the 32 bank rows, the other 16 in lane ^ 32) makes the row
// maximum 15 in-lane v_max + ONE v_permlane32_swap instead of two swaps per 16 rows.  Same image (the A fragment of lane l is
// piece 2 ks + (l >> 5) of bank row l & 15 of slot (l >> 4) & 1: conflict-free under the image's swizzle,
// docs/history/tools/lds_swizzle_check.py), same splits, same partial layout; two 32-row buffers, LDS-DMA one step ahead, the soft-max of a
// step deferred into the next step's first fragment-read latency.
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

//
// D = 512 (round 6): the pre-split rows of V are D/2 = 256 registers per wave -- the whole budget of a wave at two waves per SIMD.
// The same kernel runs there with NW = 4 waves (ONE per SIMD, up to 512 registers: V 256 + accumulators 48 + two fragment bursts),
// 128 rows of V per workgroup, two 64 KB step buffers = one workgroup per CU: the LDS-read : MFMA ratio of a step is the one of the
// 8-wave D = 256 form (every wave reads the whole step, 32 x 3 MFMAs of 32 cycles per 64 KB), the image passes a CU once per 128
// rows (391 x 102 MB = 40 GB L2 -> LDS per client where the tile GEMM moves 78 GB), and what hides the fragment-read latency is
// the next burst's reads issued ahead of this burst's MFMAs inside the one wave (the registers are there) instead of a second
// wave.  D = 768 does not fit either way (384 registers of V, 2 x 96 KB of step buffers): it stays on the tile GEMM of bank.hip.
template <int DT, int NW, int RB = 4, int SCHED = 0, bool ADMA = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void cfl_bank_wide32_kernel(const float* __restrict__ F, const char* __restrict__ img, int B,
                                                                        int M, int D, float sc2, int S, int RG,
                                                                        float* __restrict__ part_m, float* __restrict__ part_l) {
    constexpr int DP = 32 * DT, SLOT = 64 * DP, STEP = 2 * SLOT, KS = DP / 16, FR = 32 * NW;
    constexpr int STRIPE = NW * 1024;                                     // bytes one DMA instruction of every wave covers
    constexpr int NPT = STEP / STRIPE;                                    // 16-byte DMA pieces per thread and step
    static_assert(KS % RB == 0 && STEP % STRIPE == 0 && SLOT % STRIPE == 0, "bursts tile the contraction, stripes tile a slot");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, kh = lane >> 5;
    const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3;
    const int rg = j8 % RG, x = (j8 / RG) * 8 + xcd;
    const int nslot = (M + SG - 1) / SG;
    const int base = nslot / S, extra = nslot % S;
    const int c0 = x * base + (x < extra ? x : extra);
    const int nmine = base + (x < extra ? 1 : 0);                        // 16-row slots of this split
    const int nstep = (nmine + 1) >> 1;
    const int limit = min(M, (c0 + nmine) * SG);                         // first bank row that is not this split's
    const bool wave_live = rg * FR + 32 * w < B;
    const char* sbase = img + (size_t)c0 * SLOT + (w * 64 + lane) * 16;
    // ADMA (round 6): the LDS-DMA issued from inline assembly.  While a transfer the compiler KNOWS of is in flight its wait-count
    // bookkeeping gives up partial LDS waits: every wait for a fragment read becomes lgkmcnt(0), i.e. also for the reads issued last.
    // What it does not see it does not account for; the explicit vmcnt(0) in front of the step's barrier is the synchronisation.
    const unsigned lds_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto dma_step = [&](int it) {
        const char* src = sbase + (size_t)it * STEP;
        char* dst = lds + (it & 1) * STEP + w * 1024;
        const unsigned dst_a = __builtin_amdgcn_readfirstlane(lds_a + (it & 1) * STEP + w * 1024);
        const int valid = (2 * it + 1 < nmine) ? STEP : SLOT;
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (j * STRIPE < valid) {
                if (ADMA) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src + j * STRIPE), "s"(dst_a + j * STRIPE) : "memory");
                } else {
                    __builtin_amdgcn_global_load_lds((glb_vptr)(src + j * STRIPE), (lds_vptr)(dst + j * STRIPE), 16, 0, 0);
                }
            }
    };
    if (nstep > 0) dma_step(0);
    bf16x8 fh[KS], fl[KS];                                       // B operand: feature row r32, k = 16 ks + 8 kh .. + 7, pre-scaled
    {
        const int fr = rg * FR + 32 * w + r32;
        const float* fp = F + (long long)(fr < B ? fr : B - 1) * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 16 * ks + 8 * kh;
            const bool ok0 = fr < B && k0 < D, ok1 = fr < B && k0 + 4 < D;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(fp + (k0 < D ? k0 : 0));
            f32x4 v1 = *reinterpret_cast<const f32x4*>(fp + (k0 + 4 < D ? k0 + 4 : 0));
            v0 *= sc2; v1 *= sc2;
            bf16x4 h0, l0, h1, l1;
            split4(v0, ok0, h0, l0);
            split4(v1, ok1, h1, l1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { fh[ks][e] = h0[e]; fh[ks][4 + e] = h1[e]; fl[ks][e] = l0[e]; fl[ks][4 + e] = l1[e]; }
        }
    }
    int la[8];                                                   // A operand: bank row r32 of the step, piece 2 ks + kh
    {
        const int g = r32 & 15;
#pragma unroll
        for (int j = 0; j < 8; ++j) la[j] = (r32 >> 4) * SLOT + g * (DP * 2) + (((2 * j + kh) ^ img_swz(g)) << 4);
    }
    float run_m = -INFINITY, run_l = 0.f;
    auto soft = [&](f32x16& v, int row0) {                       // v[r]: bank row row0 + 8 (r >> 2) + 4 kh + (r & 3), feature row r32
        if (__builtin_amdgcn_readfirstlane(row0) + 32 > limit) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (row0 + 8 * (r >> 2) + 4 * kh + (r & 3) >= limit) v[r] = -INFINITY;
        }
        float mx = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, v[r]);
        {
            const unsigned u = __float_as_uint(mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        if (__any(mx > run_m + RESCALE_THR)) {
            const float mn = fmaxf(run_m, mx);
            run_l *= run_m == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(run_m - mn);
            run_m = mn;
        }
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            a0 += __builtin_amdgcn_exp2f(v[r] - run_m);
            a1 += __builtin_amdgcn_exp2f(v[r + 1] - run_m);
        }
        run_l += a0 + a1;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 pend;
    int pend_row0 = 0;
    bool pend_ok = false;
    for (int it = 0; it < nstep; ++it) {
        if (it + 1 < nstep) dma_step(it + 1);
        const char* sb = lds + (it & 1) * STEP;
        f32x16 sa, sbb, sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sbb[r] = 0.f; sc[r] = 0.f; }
        if (NW == 8) {
#pragma unroll
            for (int k0 = 0; k0 < KS; k0 += RB) {
                bf16x8 ah[RB], al[RB];
#pragma unroll
                for (int e = 0; e < RB; ++e) {
                    const int off = la[(k0 + e) & 7] + ((k0 + e) >> 3) * 256;
                    ah[e] = *reinterpret_cast<const bf16x8*>(sb + off);
                    al[e] = *reinterpret_cast<const bf16x8*>(sb + SG * DP * 2 + off);
                }
                if (k0 == 0 && pend_ok) soft(pend, pend_row0);   // the previous step's soft-max, inside this burst's LDS latency
                // (no branch on wave_live, round 6: behind a branch per burst the compiler issued the fragment reads of bursts 1 .. 3 one
                // pair at a time, each waited for at lgkmcnt(0) right in front of its MFMA triple; a dead wave -- last row group only --
                // multiplies its clamped rows and stores nothing)
                if (SCHED == 3 ? wave_live : true) {             // (SCHED = 3: the branch of rounds 4-5, kept for the A/B)
#pragma unroll
                    for (int e = 0; e < RB; ++e) {
                        sa = MFMA32(ah[e], fh[k0 + e], sa);
                        sbb = MFMA32(al[e], fh[k0 + e], sbb);
                        sc = MFMA32(ah[e], fl[k0 + e], sc);
                    }
                }
                if (SCHED == 1) {                                // pinned: the burst's eight reads, then its twelve MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * RB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * RB, 0);
                }
            }
        } else {
            // one wave per SIMD: nobody else covers a burst's LDS latency, so burst k + 1 is read (second register set) before the
            // MFMAs of burst k are issued; no branch on wave_live (a dead wave multiplies its clamped rows and stores nothing):
            // a branch per burst makes the compiler wait for every read pair right behind its issue
            bf16x8 ah[2][RB], al[2][RB];
            auto rd = [&](int k0, bf16x8 (&h)[RB], bf16x8 (&l)[RB]) {
#pragma unroll
                for (int e = 0; e < RB; ++e) {
                    const int off = la[(k0 + e) & 7] + ((k0 + e) >> 3) * 256;
                    h[e] = *reinterpret_cast<const bf16x8*>(sb + off);
                    l[e] = *reinterpret_cast<const bf16x8*>(sb + SG * DP * 2 + off);
                }
            };
            rd(0, ah[0], al[0]);
#pragma unroll
            for (int k0 = 0; k0 < KS; k0 += RB) {
                const int cur = (k0 / RB) & 1;
                if (k0 + RB < KS) rd(k0 + RB, ah[cur ^ 1], al[cur ^ 1]);
                if (k0 == 0 && pend_ok) soft(pend, pend_row0);
#pragma unroll
                for (int e = 0; e < RB; ++e) {
                    sa = MFMA32(ah[cur][e], fh[k0 + e], sa);
                    sbb = MFMA32(al[cur][e], fh[k0 + e], sbb);
                    sc = MFMA32(ah[cur][e], fl[k0 + e], sc);
                }
                if (SCHED && k0 + RB < KS) {
                    // issue order of the burst: the two fragment reads of contraction step e of the NEXT burst between the MFMA
                    // triples of this one (left alone the compiler issues all eight reads behind the last MFMA and, two bursts
                    // on, waits for lgkmcnt(0) with them fresh in flight)
#pragma unroll
                    for (int e = 0; e < RB; ++e) {
                        __builtin_amdgcn_sched_group_barrier(0x008, SCHED == 2 ? 1 : 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        if (SCHED == 2) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    }
                }
            }
        }
        pend = sa + (sbb + sc);
        pend_row0 = (c0 + 2 * it) * SG;
        pend_ok = wave_live;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (pend_ok) soft(pend, pend_row0);
    if (!wave_live) return;
    run_l += __shfl_xor(run_l, 32, 64);
    const int f = rg * FR + 32 * w + r32;
    if (kh == 0 && f < B) {
        const size_t o = (size_t)(f >> 7) * S * BR + (size_t)x * BR + (f & (BR - 1));
        part_m[o] = run_m;
        part_l[o] = run_l;
    }
}

// (Measured and dropped, round 6: D <= 256 with 64 rows of V per wave -- two 32-row B tiles per A fragment read, four waves, one per
// SIMD, half the LDS bytes per MFMA of the 8-wave form, per-lane running log-sum-exp, pinned issue order: 3.47-3.50 ms against 3.30-3.31
// ms at M = 50 000, D = 256 on the same lease (profiles/r6_a5_wide64_ab.jsonl).  The LDS read path is not what holds the 8-wave form at
// 0.47 of the roof.)

// ---- D = 768 (round 6): 16-row steps on v_mfma_f32_16x16x32_bf16 ----------------------------------------------------------------
// 384 registers of pre-split V per wave (32 rows) and 2 x 96 KB of step buffers rule out the 32 x 32 form above.  Here a step is ONE
// 16-row slot (48 KB): three slot buffers (144 KB), the LDS-DMA two steps ahead with a counted vmcnt wait; a wave's 32 rows of V are
// two 16-row B tiles of the 16 x 16 x 32 MFMA (the stream kernel's logits arrangement: A = the slot, a lane owns one row of V and
// four of the slot's bank rows per tile), so every fragment read serves two MFMA triples and the LDS-read : MFMA ratio is the
// 32 x 32 form's.  The running (max, sum) is kept PER LANE over the bank rows the lane sees (4 of every 16) -- partial log-sum-exps
// over disjoint row sets merge associatively -- so the loop has no cross-lane operation at all; the four lanes of a row merge once,
// at the end.  Four waves, one per SIMD, ~480 registers, one workgroup per CU, 128 rows of V per workgroup.
#define MFMA16W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
template <int DT>
__global__ __launch_bounds__(256, 1) void cfl_bank_wide16_kernel(const float* __restrict__ F, const char* __restrict__ img, int B, int M,
                                                               int D, float sc2, int S, int RG, float* __restrict__ part_m,
                                                               float* __restrict__ part_l) {
    constexpr int DP = 32 * DT, SLOT = 64 * DP, KS = DT, RB = 2, NW = 4, FR = 32 * NW;
    constexpr int STRIPE = NW * 1024, NPT = SLOT / STRIPE;                // 16-byte DMA pieces per thread and slot
    static_assert(KS % (2 * RB) == 0 && SLOT % STRIPE == 0 && NPT <= 31, "bursts tile the contraction, stripes tile a slot");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f16 = lane & 15, kg = lane >> 4;
    const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3;
    const int rg = j8 % RG, x = (j8 / RG) * 8 + xcd;
    const int nslot = (M + SG - 1) / SG;
    const int base = nslot / S, extra = nslot % S;
    const int c0 = x * base + (x < extra ? x : extra);
    const int nstep = base + (x < extra ? 1 : 0);                        // 16-row slots of this split = steps
    const int limit = min(M, (c0 + nstep) * SG);
    const bool wave_live = rg * FR + 32 * w < B;
    const char* sbase = img + (size_t)c0 * SLOT + (w * 64 + lane) * 16;
    auto dma_step = [&](int it, int buf) {
        const char* src = sbase + (size_t)it * SLOT;
        char* dst = lds + buf * SLOT + w * 1024;
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            __builtin_amdgcn_global_load_lds((glb_vptr)(src + j * STRIPE), (lds_vptr)(dst + j * STRIPE), 16, 0, 0);
    };
    if (nstep > 0) dma_step(0, 0);
    if (nstep > 1) dma_step(1, 1);
    bf16x8 fh[2][KS], fl[2][KS];                                 // B operands: row 16 tt + f16 of the wave's 32, k = 32 ks + 8 kg .. + 7
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int fr = rg * FR + 32 * w + 16 * tt + f16;
        const float* fp = F + (long long)(fr < B ? fr : B - 1) * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 32 * ks + 8 * kg;
            const bool ok0 = fr < B && k0 < D, ok1 = fr < B && k0 + 4 < D;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(fp + (k0 < D ? k0 : 0));
            f32x4 v1 = *reinterpret_cast<const f32x4*>(fp + (k0 + 4 < D ? k0 + 4 : 0));
            v0 *= sc2; v1 *= sc2;
            bf16x4 h0, l0, h1, l1;
            split4(v0, ok0, h0, l0);
            split4(v1, ok1, h1, l1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { fh[tt][ks][e] = h0[e]; fh[tt][ks][4 + e] = h1[e]; fl[tt][ks][e] = l0[e]; fl[tt][ks][4 + e] = l1[e]; }
        }
    }
    int la[4];                                                   // A operand: bank row f16 of the slot, piece 4 ks + kg
#pragma unroll
    for (int j = 0; j < 4; ++j) la[j] = f16 * (DP * 2) + (((4 * j + kg) ^ img_swz(f16)) << 4);
    float run_m[2] = {-INFINITY, -INFINITY}, run_l[2] = {0.f, 0.f};
    auto soft = [&](const f32x4 (&v)[2], int row0) {             // v[tt][r]: bank row row0 + 4 kg + r, row 16 tt + f16 of V
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = row0 + 4 * kg + r < limit ? v[tt][r] : -INFINITY;
            const float mn = fmaxf(fmaxf(run_m[tt], fmaxf(y[0], y[1])), fmaxf(y[2], y[3]));
            const float ref = mn == -INFINITY ? 0.f : mn;
            float a = run_l[tt] * __builtin_amdgcn_exp2f(run_m[tt] - ref);
            a += __builtin_amdgcn_exp2f(y[0] - ref) + __builtin_amdgcn_exp2f(y[1] - ref);
            a += __builtin_amdgcn_exp2f(y[2] - ref) + __builtin_amdgcn_exp2f(y[3] - ref);
            run_l[tt] = a;
            run_m[tt] = mn;
        }
    };
    if (nstep > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    f32x4 pend[2];
    int pend_row0 = 0;
    bool pend_ok = false;
    int buf = 0;                                                 // step it lives in buffer it % 3
    for (int it = 0; it < nstep; ++it) {
        const int nb2 = buf == 0 ? 2 : buf - 1;                  // (it + 2) % 3: the buffer step it - 1 was read from
        if (it + 2 < nstep) dma_step(it + 2, nb2);
        const char* sb = lds + buf * SLOT;
        f32x4 sa[2], sbb[2], sc[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { sa[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; sbb[tt] = sa[tt]; sc[tt] = sa[tt]; }
        bf16x8 ah[2][RB], al[2][RB];
        auto rd = [&](int k0, bf16x8 (&h)[RB], bf16x8 (&l)[RB]) {
#pragma unroll
            for (int e = 0; e < RB; ++e) {
                const int off = la[(k0 + e) & 3] + ((k0 + e) >> 2) * 256;
                h[e] = *reinterpret_cast<const bf16x8*>(sb + off);
                l[e] = *reinterpret_cast<const bf16x8*>(sb + SG * DP * 2 + off);
            }
        };
        rd(0, ah[0], al[0]);
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += RB) {
            const int cur = (k0 / RB) & 1;
            if (k0 + RB < KS) rd(k0 + RB, ah[cur ^ 1], al[cur ^ 1]);
            if (k0 == 0 && pend_ok) soft(pend, pend_row0);       // the previous step's soft-max, inside the first bursts' LDS latency
#pragma unroll
            for (int e = 0; e < RB; ++e)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    sa[tt] = MFMA16W(ah[cur][e], fh[tt][k0 + e], sa[tt]);
                    sbb[tt] = MFMA16W(al[cur][e], fh[tt][k0 + e], sbb[tt]);
                    sc[tt] = MFMA16W(ah[cur][e], fl[tt][k0 + e], sc[tt]);
                }
            if (k0 + RB < KS) {
#pragma unroll
                for (int e = 0; e < RB; ++e) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) pend[tt] = sa[tt] + (sbb[tt] + sc[tt]);
        pend_row0 = (c0 + it) * SG;
        pend_ok = wave_live;
        // step it + 1 must have landed (issued one iteration ago); step it + 2's pieces may stay in flight
        // (a bare s_barrier: __syncthreads() is a fence the compiler drains vmcnt to 0 for, which would put step it + 2's round trip
        // back on the critical path; every ds_read of this step was consumed by an MFMA above)
        if (it + 2 < nstep) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (pend_ok) soft(pend, pend_row0);
    if (!wave_live) return;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float om = __shfl_xor(run_m[tt], off, 64), ol = __shfl_xor(run_l[tt], off, 64);
            const float mn = fmaxf(run_m[tt], om);
            const float ref = mn == -INFINITY ? 0.f : mn;
            run_l[tt] = run_l[tt] * __builtin_amdgcn_exp2f(run_m[tt] - ref) + ol * __builtin_amdgcn_exp2f(om - ref);
            run_m[tt] = mn;
        }
        const int f = rg * FR + 32 * w + 16 * tt + f16;
        if (kg == 0 && f < B) {
            const size_t o = (size_t)(f >> 7) * S * BR + (size_t)x * BR + (f & (BR - 1));
            part_m[o] = run_m[tt];
            part_l[o] = run_l[tt];
        }
    }
}

struct GsPlan { int DT, DP, RG, S, RGF, Bp, wide, big; };
// big: the wide-batch forward (no gradient state): 8 waves x NRG x 16 rows per workgroup on one slot stream
static int gs_big_rows(int D) { return D > 256 ? 128 : 256; }       // cfl_bank_wide32_kernel: 8 waves x 32 rows, 4 waves beyond D = 256
static GsPlan gs_plan(int B, int M, int D, int big = 0) {
    GsPlan p;
    p.wide = D > 256;
    p.big = big;
    p.DP = D <= 128 ? 128 : (D <= 256 ? 256 : (D <= 512 ? 512 : 768));
    p.DT = p.DP / 32;
    p.RG = cfl_cdiv(B, big ? gs_big_rows(D) : (p.wide ? 64 : FB));   // rows per workgroup: 32 at D <= 256, 64 beyond (4 column-split pairs)
    p.RGF = cfl_cdiv(B, BR);
    p.Bp = p.RGF * BR;
    int s = (256 / p.RG) & ~7;             // ~one workgroup per CU; a multiple of 8 (block id -> XCD mapping), <= 256 (finish)
    if (s < 8) s = 8;
    if (big) {                             // many row groups, one workgroup per CU: enough splits that the last round of workgroups
        while (s < 32 && p.RG * s < 256 * 24) s += 8;     // is a small share (M = 50 000: 8 splits 3.58 ms, 16: 3.42, 32: 3.36, 64: 3.40)
    }
    const int nslot = cfl_cdiv(M, SG);
    while (s > 8 && s > nslot) s -= 8;
    p.S = s;
    return p;
}

}  // namespace gs
