// attn_small.hip -- self-attention of the BERT text tower for short sequences (L <= 32 tokens, head dim 64).
//
// Where it sits: BertSelfAttention inside the `BertModel` the reference builds at src/networks/models/pcme.py:36-38
// (third-party `transformers`; softmax(Q K^T / sqrt(d) + key-padding mask) V per head).  Captions are ~12-30 tokens,
// so one (batch, head) problem is a single 32x32 score tile: the generic flash kernels torch dispatches to spend
// 24 us forward / 103 us backward per layer at the bench shape (256 x 12 heads x 24 tokens), almost all of it
// latency.  Here one wavefront owns one (batch, head): Q, K, V (and dO) are staged once in LDS (4 KB each), every
// product is a handful of v_mfma_f32_32x32x16_bf16, the softmax never leaves registers, and nothing but the inputs
// is saved for the backward (it recomputes the 32x32 probabilities).
//
// Layout trick (same as the online-LSE kernels): the score tile is produced TRANSPOSED, S^T = K Q^T, so that a lane
// owns one query and its keys sit in the lane's own accumulator registers (+ the partner lane 32 away): row max / sum
// are in-register.  The accumulator registers r = 8kk .. 8kk+7 of a lane are, in the MFMA's own K-index convention,
// eight distinct keys -- they are fed straight back as the A operand of P V (or dS K), with the B operand gathered
// from LDS in the same key order.  For dV / dK the products need lanes = keys, so the (cheap) score tile is simply
// recomputed un-transposed; only the per-query statistics (max, 1/sum, delta) cross lanes, through 3 x 32 floats of LDS.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32;

constexpr int TP = 72;                      // LDS tile row pitch in bf16 elements (64 + 8: 144 B, 16-byte aligned rows)
constexpr int TILE = 32 * TP;               // one [32][64] bf16 tile

__device__ __forceinline__ u32 bf_rne(float f) {
    u32 u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf_to_f(u16 v) { return __uint_as_float((u32)v << 16); }

// global [L rows][64] bf16 (row stride ld elements) -> LDS tile, rows >= L zero-filled.  One wave.
__device__ __forceinline__ void stage_tile(const u16* __restrict__ src, long long ld, int L, u16* tile, int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = it * 64 + lane, row = c >> 3, cc = c & 7;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < L) v = *reinterpret_cast<const f32x4*>(src + (long long)row * ld + cc * 8);
        *reinterpret_cast<f32x4*>(tile + row * TP + cc * 8) = v;
    }
}
// X[row = lane&31][16kk + 8(lane>>5) .. +7]: the A operand "rows of X", or the B operand "X^T"
__device__ __forceinline__ bf16x8 frag_rows(const u16* tile, int kk, int lane) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(tile + (lane & 31) * TP + kk * 16 + (lane >> 5) * 8));
}
// B operand X[k][col] for col = 32dt + (lane&31), k running over the accumulator-register order of k-step kk:
// row(t) = 16kk + 4(lane>>5) + (t&3) + 8(t>>2)
__device__ __forceinline__ bf16x8 frag_gather(const u16* tile, int kk, int dt, int lane) {
    const u16* p = tile + (kk * 16 + (lane >> 5) * 4) * TP + dt * 32 + (lane & 31);
    u32 w[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r0 = (2 * t) & 3, r1 = (2 * t + 1) & 3, g = (2 * t) >> 2;
        w[t] = (u32)p[(r0 + 8 * g) * TP] | ((u32)p[(r1 + 8 * g) * TP] << 16);
    }
    f32x4 v;
    v[0] = __uint_as_float(w[0]); v[1] = __uint_as_float(w[1]); v[2] = __uint_as_float(w[2]); v[3] = __uint_as_float(w[3]);
    return __builtin_bit_cast(bf16x8, v);
}
// accumulator registers 8kk .. 8kk+7 as an A operand (bf16)
__device__ __forceinline__ bf16x8 frag_acc(const float (&a)[16], int kk) {
    f32x4 v;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = __uint_as_float(bf_rne(a[8 * kk + 2 * t]) | (bf_rne(a[8 * kk + 2 * t + 1]) << 16));
    return __builtin_bit_cast(bf16x8, v);
}
// row index of accumulator register r for this lane (C/D layout of the 32x32 MFMA)
__device__ __forceinline__ int acc_r(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// acc = X rows (A) * Y rows^T (B) over d = 64: acc[row of X][row of Y]
__device__ __forceinline__ void mma_rows(const u16* X, const u16* Y, f32x16& acc, int lane) {
    zero16(acc);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(X, kk, lane), frag_rows(Y, kk, lane), acc, 0, 0, 0);
}
// out[row][d] (two 32-wide d tiles) = sum_k W[row][k] * X[k][d], W given in accumulator-register form
__device__ __forceinline__ void mma_apply(const float (&w)[16], const u16* X, f32x16 (&out)[2], int lane) {
    const bf16x8 a0 = frag_acc(w, 0), a1 = frag_acc(w, 1);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        zero16(out[dt]);
        out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, frag_gather(X, 0, dt, lane), out[dt], 0, 0, 0);
        out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, frag_gather(X, 1, dt, lane), out[dt], 0, 0, 0);
    }
}
// store out[row = acc_r][d = 32dt + lane&31] for rows < L
__device__ __forceinline__ void store_rows(const f32x16 (&out)[2], u16* __restrict__ dst, long long ld, int L, int lane) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_r(r, lane);
            if (row < L) dst[(long long)row * ld + dt * 32 + (lane & 31)] = (u16)bf_rne(out[dt][r]);
        }
}

struct AttnArgs {
    const u16 *q, *k, *v;            // [B][L][...]: element (b, t, h*64 + d) at b*bs + t*ld + h*64 + d
    long long ld, bs;
    const unsigned char* mask;       // [B][L] key-padding mask (1 = attend) or NULL
    u16* o;                          // forward output / backward: unused
    long long ldo, bso;
    const u16* dout;                 // backward: gradient of o (same ldo / bso)
    u16 *dq, *dk, *dv;               // backward outputs, addressed like q/k/v with ldg / bsg
    long long ldg, bsg;
    int B, L, heads;
    float scale;
    const int* cu;                   // packed ("varlen") form: sequence b owns rows cu[b] .. cu[b + 1] - 1 of [T][...] tensors (bs unused,
                                     // no mask: every row of a sequence is a token); NULL = the padded [B][L] form
};

// first row, row count and valid-key bits of one (batch) problem
struct SeqRows { long long row0; int len; unsigned bits; };

__device__ __forceinline__ u32 key_bits(const AttnArgs& a, int b);
__device__ __forceinline__ SeqRows seq_rows(const AttnArgs& a, int b) {
    SeqRows s;
    if (a.cu) {
        const int r0 = a.cu[b], r1 = a.cu[b + 1];
        s.row0 = r0;
        s.len = min(r1 - r0, 32);
        s.bits = s.len >= 32 ? 0xffffffffu : ((1u << s.len) - 1u);
    } else {
        s.row0 = -1;
        s.len = a.L;
        s.bits = key_bits(a, b);
    }
    return s;
}

// valid-key bits of one batch row (bit j set = key j takes part)
__device__ __forceinline__ u32 key_bits(const AttnArgs& a, int b) {
    u32 bits = a.L >= 32 ? 0xffffffffu : ((1u << a.L) - 1u);
    if (a.mask) {
        u32 m = 0;
        for (int j = 0; j < a.L; ++j) m |= (a.mask[(long long)b * a.L + j] ? 1u : 0u) << j;
        bits &= m;
    }
    return bits;
}

// transposed scores: lane = query i (= lane&31), st[r] = score of key acc_r(r); returns probabilities in st, row max / 1/sum
__device__ __forceinline__ void softmax_t(float (&st)[16], u32 bits, float scale, int lane, float& m, float& inv_l) {
    m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool ok = (bits >> acc_r(r, lane)) & 1u;
        st[r] = ok ? st[r] * scale : -INFINITY;
        m = fmaxf(m, st[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        st[r] = (m == -INFINITY) ? 0.f : __expf(st[r] - m);
        l += st[r];
    }
    l += __shfl_xor(l, 32, 64);
    inv_l = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] *= inv_l;
}

__global__ __launch_bounds__(256) void cfl_attn_small_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) u16 lds[4][3 * TILE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= a.B * a.heads) return;
    const int b = bh / a.heads, h = bh % a.heads;
    u16 *Q = lds[wave], *K = Q + TILE, *V = K + TILE;
    const SeqRows sr = seq_rows(a, b);
    const long long off = (a.cu ? sr.row0 * a.ld : (long long)b * a.bs) + h * 64;
    stage_tile(a.q + off, a.ld, sr.len, Q, lane);
    stage_tile(a.k + off, a.ld, sr.len, K, lane);
    stage_tile(a.v + off, a.ld, sr.len, V, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const u32 bits = sr.bits;
    f32x16 acc;
    mma_rows(K, Q, acc, lane);                       // S^T[key][query]
    float p[16], m, inv_l;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = acc[r];
    softmax_t(p, bits, a.scale, lane, m, inv_l);
    f32x16 out[2];
    mma_apply(p, V, out, lane);                      // O[query][d]
    store_rows(out, a.o + (a.cu ? sr.row0 * a.ldo : (long long)b * a.bso) + h * 64, a.ldo, sr.len, lane);
}

// two waves per workgroup: 4 staged tiles per wave (Q, K, V, dO) = 36 KB of LDS per workgroup
__global__ __launch_bounds__(128) void cfl_attn_small_bwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) u16 lds[2][4 * TILE];
    __shared__ float stats[2][3][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 2 + wave;
    if (bh >= a.B * a.heads) return;
    const int b = bh / a.heads, h = bh % a.heads;
    u16 *Q = lds[wave], *K = Q + TILE, *V = K + TILE, *G = V + TILE;
    const SeqRows sr = seq_rows(a, b);
    const int L = sr.len;
    const long long off = (a.cu ? sr.row0 * a.ld : (long long)b * a.bs) + h * 64;
    stage_tile(a.q + off, a.ld, L, Q, lane);
    stage_tile(a.k + off, a.ld, L, K, lane);
    stage_tile(a.v + off, a.ld, L, V, lane);
    stage_tile(a.dout + (a.cu ? sr.row0 * a.ldo : (long long)b * a.bso) + h * 64, a.ldo, L, G, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const u32 bits = sr.bits;
    const long long goff = (a.cu ? sr.row0 * a.ldg : (long long)b * a.bsg) + h * 64;
    f32x16 acc, out[2];
    float p[16], w[16], m, inv_l;
    // ---- lanes = queries: P^T, dP^T -> delta, dS^T -> dQ
    mma_rows(K, Q, acc, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = acc[r];
    softmax_t(p, bits, a.scale, lane, m, inv_l);
    mma_rows(V, G, acc, lane);                       // dP^T[key][query] = V[key] . dO[query]
    float delta = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) delta = fmaf(p[r], acc[r], delta);
    delta += __shfl_xor(delta, 32, 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = p[r] * (acc[r] - delta) * a.scale;
    mma_apply(w, K, out, lane);                      // dQ[query][d] = sum_key dS[query][key] K[key][d]
    store_rows(out, a.dq + goff, a.ldg, L, lane);
    if (lane < 32) { stats[wave][0][lane] = m; stats[wave][1][lane] = inv_l; stats[wave][2][lane] = delta; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- lanes = keys: P, dP recomputed un-transposed (registers run over queries) -> dV, dK
    const bool key_ok = (bits >> (lane & 31)) & 1u;
    mma_rows(Q, K, acc, lane);                       // S[query][key]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = acc_r(r, lane);
        const float mi = stats[wave][0][i];
        p[r] = (key_ok && mi != -INFINITY) ? __expf(acc[r] * a.scale - mi) * stats[wave][1][i] : 0.f;
    }
    mma_rows(G, V, acc, lane);                       // dP[query][key]
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = p[r] * (acc[r] - stats[wave][2][acc_r(r, lane)]) * a.scale;
    // A operand rows must be keys: the accumulators hold [query regs][key lane], i.e. already "row = lane's key"
    mma_apply(p, G, out, lane);                      // dV[key][d] = sum_query P[query][key] dO[query][d]
    store_rows(out, a.dv + goff, a.ldg, L, lane);
    mma_apply(w, Q, out, lane);                      // dK[key][d] = sum_query dS[query][key] Q[query][d]
    store_rows(out, a.dk + goff, a.ldg, L, lane);
}

}  // namespace

extern "C" {

int cfl_attn_small_fwd(const void* q, const void* k, const void* v, long long ld, long long bs, const unsigned char* mask,
                       int B, int L, int heads, int head_dim, void* o, long long ldo, long long bso, void* stream_) {
    if (!q || !k || !v || !o || B <= 0 || L <= 0 || heads <= 0) return CFL_EINVAL;
    if (L > 32 || head_dim != 64 || ld % 8 != 0 || bs % 8 != 0 ||
        (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    AttnArgs a{};
    a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.ld = ld; a.bs = bs; a.mask = mask;
    a.o = (u16*)o; a.ldo = ldo; a.bso = bso; a.B = B; a.L = L; a.heads = heads; a.scale = 0.125f;
    CFL_LAUNCH(K_ATTN_SMALL, cfl_attn_small_fwd_kernel, dim3(cfl_cdiv(B * heads, 4)), dim3(256), 0, stream, a);
    return 0;
}

int cfl_attn_small_bwd(const void* q, const void* k, const void* v, long long ld, long long bs, const unsigned char* mask,
                       int B, int L, int heads, int head_dim, const void* dout, long long ldo, long long bso, void* dq, void* dk,
                       void* dv, long long ldg, long long bsg, void* stream_) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || B <= 0 || L <= 0 || heads <= 0) return CFL_EINVAL;
    if (L > 32 || head_dim != 64 || ld % 8 != 0 || bs % 8 != 0 || ldo % 8 != 0 || bso % 8 != 0 ||
        (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    AttnArgs a{};
    a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.ld = ld; a.bs = bs; a.mask = mask;
    a.dout = (const u16*)dout; a.ldo = ldo; a.bso = bso; a.dq = (u16*)dq; a.dk = (u16*)dk; a.dv = (u16*)dv; a.ldg = ldg; a.bsg = bsg;
    a.B = B; a.L = L; a.heads = heads; a.scale = 0.125f;
    CFL_LAUNCH(K_ATTN_SMALL, cfl_attn_small_bwd_kernel, dim3(cfl_cdiv(B * heads, 2)), dim3(128), 0, stream, a);
    return 0;
}

// ---- packed ("varlen") form, round 6: q / k / v / o / gradients are [T][...] with T = the batch's token count; sequence b owns rows
// cu[b] .. cu[b + 1] - 1 (cu: B + 1 ints on the device, cu[0] = 0, every sequence <= 32 tokens).  Same kernels, same arithmetic per
// sequence as the padded form with its key-padding mask -- the padded rows (a third of a COCO batch) are simply not there.
int cfl_attn_small_fwd_varlen(const void* q, const void* k, const void* v, long long ld, const int* cu_seqlens, int B, int heads,
                              int head_dim, void* o, long long ldo, void* stream_) {
    if (!q || !k || !v || !o || !cu_seqlens || B <= 0 || heads <= 0) return CFL_EINVAL;
    if (head_dim != 64 || ld % 8 != 0 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15)) return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    AttnArgs a{};
    a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.ld = ld; a.cu = cu_seqlens;
    a.o = (u16*)o; a.ldo = ldo; a.B = B; a.L = 32; a.heads = heads; a.scale = 0.125f;
    CFL_LAUNCH(K_ATTN_SMALL, cfl_attn_small_fwd_kernel, dim3(cfl_cdiv(B * heads, 4)), dim3(256), 0, stream, a);
    return 0;
}

int cfl_attn_small_bwd_varlen(const void* q, const void* k, const void* v, long long ld, const int* cu_seqlens, int B, int heads,
                              int head_dim, const void* dout, long long ldo, void* dq, void* dk, void* dv, long long ldg,
                              void* stream_) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || !cu_seqlens || B <= 0 || heads <= 0) return CFL_EINVAL;
    if (head_dim != 64 || ld % 8 != 0 || ldo % 8 != 0 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout) & 15))
        return CFL_ELIMIT;
    hipStream_t stream = (hipStream_t)stream_;
    AttnArgs a{};
    a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.ld = ld; a.cu = cu_seqlens;
    a.dout = (const u16*)dout; a.ldo = ldo; a.dq = (u16*)dq; a.dk = (u16*)dk; a.dv = (u16*)dv; a.ldg = ldg;
    a.B = B; a.L = 32; a.heads = heads; a.scale = 0.125f;
    CFL_LAUNCH(K_ATTN_SMALL, cfl_attn_small_bwd_kernel, dim3(cfl_cdiv(B * heads, 2)), dim3(128), 0, stream, a);
    return 0;
}

}  // extern "C"
