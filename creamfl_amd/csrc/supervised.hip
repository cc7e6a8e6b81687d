// supervised.hip -- SURVEY 8f-4: the client's supervised step glue (src/algorithms/ClientTrainer.py:344-361).
//
//   fvec' = fvec - inter_distance * one_hot(labels)                       (:344-347)
//   loss  = CE(fvec', labels)                                             (:349, losses.create('softmax'), mean)
//   G     = class_weight @ class_weight^T ; center = CE(G, arange(C))     (:350)
//   total = loss + center_weight * center                                 (:351, center_weight = 0.5)
//   prec@1, prec@k of fvec' against labels                                (:352-357 -> accuracy :114-129)
//
// ~15 small torch launches (scatter one-hot on the CPU + H2D, sub, log_softmax, nll, mm, topk, eq, ...) per
// batch become three launches forward and one backward.  Everything is a row problem: one wave per row of fvec
// (B rows) or of G (C rows); the rows are independent and tiny (C <= a few hundred), so this is latency work,
// not bandwidth work -- the point is launch count and no host round trip.
#include "common.h"

namespace {

// ws layout (floats): rowval[B+C] | lse[B+C] | G[C*C] | hits[2] (ints)
struct SupWs {
    float* rowval;
    float* lse;
    float* G;
    int* hits;
};
__host__ __device__ inline SupWs sup_ws(void* ws, int B, int C) {
    SupWs w;
    float* p = (float*)ws;
    w.rowval = p;
    w.lse = p + (B + C);
    w.G = p + 2 * (size_t)(B + C);
    w.hits = (int*)(w.G + (size_t)C * C);
    return w;
}

// G = W W^T, one wave per entry (C^2 independent dots of length Dw: C <= a few hundred, so this is ~10^4 waves);
// thread 0 of the grid also clears the precision counters for the row kernel that follows on the stream.
__global__ __launch_bounds__(256) void cfl_sup_gram_kernel(const float* __restrict__ W, int C, int Dw, SupWs ws) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) ws.hits[threadIdx.x] = 0;
    const int i = blockIdx.x, j = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= C) return;
    const float* wi = W + (long long)i * Dw;
    const float* wj = W + (long long)j * Dw;
    float s = 0.f;
    for (int k = lane; k < Dw; k += 64) s = fmaf(wi[k], wj[k], s);
    s = wave_sum(s);
    if (lane == 0) ws.G[(long long)i * C + j] = s;
}

// rows [0,B): CE of the margin-shifted logits + rank of the true class; rows [B,B+C): CE of row i of G vs label i.
__global__ __launch_bounds__(256) void cfl_sup_fwd_kernel(const float* __restrict__ F, const long long* __restrict__ labels,
                                                          int B, int C, float margin, int topk, SupWs ws) {
    extern __shared__ float lds[];                    // 4 waves x C floats
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= B + C) return;
    float* v = lds + (size_t)wave * C;
    int y;
    if (row < B) {
        const long long yl = labels[row];
        y = (yl >= 0 && yl < C) ? (int)yl : -1;
        const float* f = F + (long long)row * C;
        for (int j = lane; j < C; j += 64) v[j] = f[j] - (j == y ? margin : 0.f);
    } else {
        y = row - B;
        const float* gi = ws.G + (long long)y * C;
        for (int j = lane; j < C; j += 64) v[j] = gi[j];
    }
    // the row lives in this wave's LDS slice: order its writes before the cross-lane reads below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float m = -INFINITY;
    for (int j = lane; j < C; j += 64) m = fmaxf(m, v[j]);
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < C; j += 64) l += expf(v[j] - m);
    l = wave_sum(l);
    const float lse = m + logf(l);
    if (y < 0) {                                       // label outside [0,C): fail loudly, like torch's device assert
        if (lane == 0) { ws.rowval[row] = NAN; ws.lse[row] = lse; }
        return;
    }
    const float vy = v[y];
    if (lane == 0) { ws.rowval[row] = lse - vy; ws.lse[row] = lse; }
    if (row < B) {
        // position of the true class in a descending sort (ties: lower class index first)
        int ahead = 0;
        for (int j = lane; j < C; j += 64) ahead += (v[j] > vy || (v[j] == vy && j < y)) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ahead += __shfl_xor(ahead, o, 64);
        if (lane == 0) {
            if (ahead < 1) atomicAdd(&ws.hits[0], 1);
            if (ahead < topk) atomicAdd(&ws.hits[1], 1);
        }
    }
}

// out[0..5) = total, ce, center, prec@1 (%), prec@k (%)
__global__ __launch_bounds__(256) void cfl_sup_final_kernel(SupWs ws, int B, int C, float center_weight, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += ws.rowval[b];
    s = block_sum_256(s, red);
    float c = 0.f;
    for (int i = threadIdx.x; i < C; i += 256) c += ws.rowval[B + i];
    c = block_sum_256(c, red);
    if (threadIdx.x == 0) {
        const float ce = s / (float)B, center = c / (float)C;
        out[0] = ce + center_weight * center;
        out[1] = ce;
        out[2] = center;
        out[3] = 100.f * (float)ws.hits[0] / (float)B;
        out[4] = 100.f * (float)ws.hits[1] / (float)B;
    }
}

// rows [0,B): dF = g/B (softmax(fvec') - onehot);   rows [B, B + C*nk): 256 columns of
// dW_i = g cw/C sum_j (P_ij + P_ji - 2 d_ij) W_j   (G is symmetric, so P_ji = exp(G_ij - lse_j)).
__global__ __launch_bounds__(256) void cfl_sup_bwd_kernel(const float* __restrict__ F, const long long* __restrict__ labels,
                                                          const float* __restrict__ W, int B, int C, int Dw, int nk,
                                                          float margin, float center_weight, const float* __restrict__ g,
                                                          SupWs ws, float* __restrict__ dF, float* __restrict__ dW) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= B + C * nk) return;
    const float gs = g[0];
    if (row < B) {
        if (!dF) return;
        const long long yl = labels[row];
        const int y = (yl >= 0 && yl < C) ? (int)yl : -1;
        const float lse = ws.lse[row], sc = gs / (float)B;
        const float* f = F + (long long)row * C;
        for (int j = lane; j < C; j += 64) {
            const float vj = f[j] - (j == y ? margin : 0.f);
            dF[(long long)row * C + j] = sc * (expf(vj - lse) - (j == y ? 1.f : 0.f));
        }
    } else {
        if (!dW) return;
        const int i = (row - B) / nk, k0 = ((row - B) % nk) * 256;
        const float sc = gs * center_weight / (float)C;
        const float lse_i = ws.lse[B + i];
        const float* gi = ws.G + (long long)i * C;
        const float* lse = ws.lse + B;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // lane j%64 evaluates the coefficient of class j once; it is broadcast when W_j is accumulated
        for (int j0 = 0; j0 < C; j0 += 64) {
            const int jl = j0 + lane;
            float c = 0.f;
            if (jl < C) {
                const float gij = gi[jl];
                c = sc * (expf(gij - lse_i) + expf(gij - lse[jl]) - (jl == i ? 2.f : 0.f));
            }
            const int jn = min(64, C - j0);
            for (int t = 0; t < jn; ++t) {
                const float ct = __shfl(c, t, 64);
                const float* wj = W + (long long)(j0 + t) * Dw + k0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = u * 64 + lane;
                    if (k0 + k < Dw) acc[u] = fmaf(ct, wj[k], acc[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * 64 + lane;
            if (k < Dw) dW[(long long)i * Dw + k] = acc[u];
        }
    }
}

}  // namespace

extern "C" {

size_t cfl_sup_ws_bytes(int B, int C) {
    if (B <= 0 || C <= 0) return 256;
    return cfl_align256(((size_t)2 * (B + C) + (size_t)C * C) * sizeof(float) + 2 * sizeof(int));
}

int cfl_sup_glue_fwd(const float* fvec, const long long* labels, const float* class_weight, int B, int C, int Dw,
                     float inter_distance, int topk, float center_weight, float* out5, void* ws_, void* stream_) {
    if (!fvec || !labels || !class_weight || !out5 || !ws_ || B <= 0 || C <= 0 || Dw <= 0 || topk < 1) return CFL_EINVAL;
    if ((size_t)C * 4 * sizeof(float) > 64 * 1024) return CFL_EINVAL;          // C <= 4096 classes
    hipStream_t stream = (hipStream_t)stream_;
    SupWs ws = sup_ws(ws_, B, C);
    CFL_LAUNCH(K_SUP_GLUE, cfl_sup_gram_kernel, dim3(C, cfl_cdiv(C, 4)), dim3(256), 0, stream, class_weight, C, Dw, ws);
    CFL_LAUNCH(K_SUP_GLUE, cfl_sup_fwd_kernel, dim3(cfl_cdiv(B + C, 4)), dim3(256), (size_t)C * 4 * sizeof(float), stream,
               fvec, labels, B, C, inter_distance, topk, ws);
    CFL_LAUNCH(K_SUP_GLUE, cfl_sup_final_kernel, dim3(1), dim3(256), 0, stream, ws, B, C, center_weight, out5);
    return 0;
}

int cfl_sup_glue_bwd(const float* fvec, const long long* labels, const float* class_weight, int B, int C, int Dw,
                     float inter_distance, float center_weight, const float* gout, const void* ws_, float* dfvec,
                     float* dclass_weight, void* stream_) {
    if (!fvec || !labels || !class_weight || !gout || !ws_ || B <= 0 || C <= 0 || Dw <= 0) return CFL_EINVAL;
    if (!dfvec && !dclass_weight) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    SupWs ws = sup_ws(const_cast<void*>(ws_), B, C);
    const int nk = cfl_cdiv(Dw, 256);
    CFL_LAUNCH(K_SUP_GLUE, cfl_sup_bwd_kernel, dim3(cfl_cdiv(B + C * nk, 4)), dim3(256), 0, stream, fvec, labels, class_weight,
               B, C, Dw, nk, inter_distance, center_weight, gout, ws, dfvec, dclass_weight);
    return 0;
}

}  // extern "C"
