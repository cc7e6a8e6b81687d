// colmap.h -- helpers shared by the HBM-streaming kernels that treat a bf16 activation as a row-major matrix
// [R rows, C columns] (csrc/bnorm.hip: NHWC activations, csrc/bertfuse.hip: token-major transformer activations):
// a thread owns 8 consecutive columns (one 16-byte access) for all its rows, column reductions go through per-block
// partials [blocks, C] and a small second kernel (deterministic: fixed summation order).
#pragma once
#include "common.h"

namespace {

typedef unsigned int u32;
struct __attribute__((aligned(16))) U4 { u32 x, y, z, w; };

__device__ __forceinline__ void unpack8(const U4& u, float (&f)[8]) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ u32 bf16_rne(float f) {
    u32 u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ U4 pack8(const float (&f)[8]) {
    U4 u;
    u.x = bf16_rne(f[0]) | (bf16_rne(f[1]) << 16); u.y = bf16_rne(f[2]) | (bf16_rne(f[3]) << 16);
    u.z = bf16_rne(f[4]) | (bf16_rne(f[5]) << 16); u.w = bf16_rne(f[6]) | (bf16_rne(f[7]) << 16);
    return u;
}

// fp32 activations (round 5: the clients' fp32 channels_last encoders): the same 8-channel group per thread, 32 bytes wide
struct __attribute__((aligned(16))) F8 { float v[8]; };
__device__ __forceinline__ void unpack8(const F8& u, float (&f)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = u.v[k];
}
template <class G> __device__ __forceinline__ G pack8g(const float (&f)[8]);
template <> __device__ __forceinline__ U4 pack8g<U4>(const float (&f)[8]) { return pack8(f); }
template <> __device__ __forceinline__ F8 pack8g<F8>(const float (&f)[8]) {
    F8 u;
#pragma unroll
    for (int k = 0; k < 8; ++k) u.v[k] = f[k];
    return u;
}
// bit k = (stored value k > 0): what the 1-bit ReLU mask keeps of an output group
__device__ __forceinline__ unsigned relu_bits(const U4& o) {
    unsigned b = 0;
    b |= (o.x & 0xffffu) ? 1u : 0u;  b |= (o.x >> 16) ? 2u : 0u;
    b |= (o.y & 0xffffu) ? 4u : 0u;  b |= (o.y >> 16) ? 8u : 0u;
    b |= (o.z & 0xffffu) ? 16u : 0u; b |= (o.z >> 16) ? 32u : 0u;
    b |= (o.w & 0xffffu) ? 64u : 0u; b |= (o.w >> 16) ? 128u : 0u;
    return b;
}
__device__ __forceinline__ unsigned relu_bits(const F8& o) {
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) b |= (o.v[k] != 0.f) ? (1u << k) : 0u;
    return b;
}

// last-use streams (activations that are dead after this kernel) bypass the caches
__device__ __forceinline__ U4 ld_nt(const U4* p) {
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
    U4 u; u.x = v[0]; u.y = v[1]; u.z = v[2]; u.w = v[3];
    return u;
}

__device__ __forceinline__ F8 ld_nt(const F8* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    const v4f b = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p) + 1);
    F8 u;
    u.v[0] = a[0]; u.v[1] = a[1]; u.v[2] = a[2]; u.v[3] = a[3]; u.v[4] = b[0]; u.v[5] = b[1]; u.v[6] = b[2]; u.v[7] = b[3];
    return u;
}

// thread -> (channel group, row phase) map shared by every kernel
struct Map {
    int tprb;     // threads per row inside this block (<= 256)
    int rpp;      // rows per pass
    int c0;       // first of this thread's 8 channels
    int rsub;     // row phase
    bool active;
};
__device__ __forceinline__ Map make_map(int C) {
    Map m;
    const int tpr = C >> 3;
    m.tprb = tpr < 256 ? tpr : 256;
    m.rpp = 256 / m.tprb;
    const int t = threadIdx.x;
    m.rsub = t / m.tprb;
    m.c0 = (blockIdx.y * 256 + (t % m.tprb)) * 8;
    m.active = (m.rsub < m.rpp) && (m.c0 < C);
    return m;
}

// column reduction of two per-thread 8-vectors across the row phases of the block -> partial[blockIdx.x][c]
__device__ __forceinline__ void block_col_reduce(const Map& m, const float (&a)[8], const float (&b)[8], int C,
                                                 float* pa, float* pb, float* lds) {
    float* la = lds;                 // [rpp][tprb*8]
    float* lb = lds + 2048;
    const int w = m.tprb * 8;
    if (m.rsub < m.rpp) {
        const int col = (threadIdx.x % m.tprb) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) { la[m.rsub * w + col + k] = a[k]; lb[m.rsub * w + col + k] = b[k]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < w; c += 256) {
        float sa = 0.f, sb = 0.f;
        for (int r = 0; r < m.rpp; ++r) { sa += la[r * w + c]; sb += lb[r * w + c]; }
        const int cg = blockIdx.y * 2048 + c;
        if (cg < C) { pa[(long long)blockIdx.x * C + cg] = sa; pb[(long long)blockIdx.x * C + cg] = sb; }
    }
}

// sum the per-block partials of two [nblk, C] arrays for 16 channels: 16 G threads = 16 channels x G partial groups.
// The partials of a group are a chain of dependent-latency loads (L2 hits, ~0.6 us per round of 8): with G = 64 a thread
// has nblk / 64 of them (12 at the 768 blocks of the BN kernels: 3 rounds instead of 12 with G = 16).  Fixed order.
template <int G>
__device__ __forceinline__ void reduce_partials(const float* __restrict__ pa, const float* __restrict__ pb, int nblk, int C,
                                                int c, int grp, float& a, float& b, float (*sa)[16], float (*sb)[16]) {
    a = 0.f; b = 0.f;
    if (c < C) {
        int i = grp;
        for (; i + 3 * G < nblk; i += 4 * G) {
            const float a0 = pa[(long long)i * C + c], a1 = pa[(long long)(i + G) * C + c];
            const float a2 = pa[(long long)(i + 2 * G) * C + c], a3 = pa[(long long)(i + 3 * G) * C + c];
            const float b0 = pb[(long long)i * C + c], b1 = pb[(long long)(i + G) * C + c];
            const float b2 = pb[(long long)(i + 2 * G) * C + c], b3 = pb[(long long)(i + 3 * G) * C + c];
            a += (a0 + a1) + (a2 + a3); b += (b0 + b1) + (b2 + b3);
        }
        for (; i < nblk; i += G) { a += pa[(long long)i * C + c]; b += pb[(long long)i * C + c]; }
    }
    sa[grp][threadIdx.x & 15] = a; sb[grp][threadIdx.x & 15] = b;
    __syncthreads();
    if (grp == 0) {
        a = 0.f; b = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) { a += sa[g][threadIdx.x & 15]; b += sb[g][threadIdx.x & 15]; }
    }
}


}  // namespace
