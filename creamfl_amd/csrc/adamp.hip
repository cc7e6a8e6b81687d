// adamp.hip -- SURVEY section 8(f) item 2: fused multi-tensor gradient-clip + AdamP step.
//
// Reference call sites: src/algorithms/retrieval_trainer.py:211-214 (clip_grad_norm_ then
// optimizer.step), src/algorithms/optimizers.py:24 (adamp.AdamP 0.3.0, third-party, not vendored: the
// algorithm is restated from Heo et al., ICLR 2021 -- PARITY UNPINNED, checked against the torch
// restatement in creamfl_amd/algorithms/optimizers.py).
//
// The whole optimizer state (155 M parameters for ResNet-101 + BERT-base) is streamed twice:
//   pass 1  m,v update, perturb = m_hat / (sqrt(v_hat) + eps); per-row sums <g,p> <g,g> <p,p> <p,perturb>
//           (a "row" = one output channel = one contiguous block of `inner` elements); 1-D tensors are
//           finished here.                                             24 B / element
//   decide  per tensor: channel-wise then layer-wise scale-invariance test, projection coefficient per row
//   pass 3  p <- p (1 - lr wd r) - step (perturb - p coef_row)          16 B / element
// All of it is HBM-bound; one wavefront owns a row, lanes read 16 B each, reductions are shuffles.
#include "common.h"

namespace {

struct Hyper {
    float beta1, beta2, eps, lr, wd, delta, wd_ratio, bc1, bc2, max_norm;
    int nesterov;
    const int* gstep;       // cfl_adamp_step_counted: the optimizer's step counter on the device (a replayed HIP graph cannot carry a
                            // host count as a launch argument); CflTensorMeta.step is then every tensor's OFFSET from it
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// streamed-once data (gradients, moments): non-temporal so that 3.7 GB of optimizer state does not evict the
// activations / weights the next forward needs from L2 / Infinity Cache
__device__ __forceinline__ f32x4 ld4_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ void st4_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }

// gradient access: fp32 or bf16 storage
typedef unsigned int u32;
struct __attribute__((aligned(8))) U2 { u32 x, y; };
__device__ __forceinline__ f32x4 ldg4(const void* g, long long e, bool bf) {
    if (bf) {
        typedef u32 v2u __attribute__((ext_vector_type(2)));
        const v2u u = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(reinterpret_cast<const unsigned short*>(g) + e));
        f32x4 v = {__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u),
                   __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xffff0000u)};
        return v;
    }
    return ld4_nt(reinterpret_cast<const float*>(g) + e);
}
__device__ __forceinline__ float ldg1(const void* g, long long e, bool bf) {
    if (bf) return __uint_as_float(((u32) reinterpret_cast<const unsigned short*>(g)[e]) << 16);
    return reinterpret_cast<const float*>(g)[e];
}
__device__ __forceinline__ u32 bf16_rne(float f) {
    u32 u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void st_shadow4(void* p16, long long e, const f32x4& v) {
    U2 u;
    u.x = bf16_rne(v[0]) | (bf16_rne(v[1]) << 16);
    u.y = bf16_rne(v[2]) | (bf16_rne(v[3]) << 16);
    *reinterpret_cast<U2*>(reinterpret_cast<unsigned short*>(p16) + e) = u;
}
__device__ __forceinline__ void st_shadow1(void* p16, long long e, float v) {
    reinterpret_cast<unsigned short*>(p16)[e] = (unsigned short)bf16_rne(v);
}

// ---- gradient norm over the tensors flagged CFL_OPT_CLIP ---------------------------------------------
__global__ __launch_bounds__(256) void cfl_gradnorm_partial_kernel(const CflTensorMeta* __restrict__ meta,
                                                                   const int* __restrict__ items, float* partial) {
    __shared__ float red[4];
    const int* it = items + (size_t)blockIdx.x * 3;
    const CflTensorMeta tm = meta[it[0]];
    float s = 0.f;
    if (tm.flags & CFL_OPT_CLIP) {
        long long e0, e1;
        if (tm.flags & CFL_OPT_MATRIX) { e0 = (long long)it[1] * tm.inner; e1 = e0 + (long long)it[2] * tm.inner; }
        else { e0 = it[1]; e1 = e0 + it[2]; }
        const bool bf = tm.flags & CFL_OPT_GRAD_BF16;
        if (((e0 | e1) & 3) == 0) {
            for (long long e = e0 + threadIdx.x * 4; e < e1; e += 1024) {
                const f32x4 v = bf ? ldg4(tm.g, e, true) : ld4(reinterpret_cast<const float*>(tm.g) + e);
                s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
            }
        } else {
            for (long long e = e0 + threadIdx.x; e < e1; e += 256) { const float x = ldg1(tm.g, e, bf); s = fmaf(x, x, s); }
        }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void cfl_gradnorm_final_kernel(const float* partial, int n, float max_norm, float* out2) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(s);
        out2[0] = nrm;
        out2[1] = fminf(1.f, max_norm / (nrm + 1e-6f));     // torch.nn.utils.clip_grad_norm_
    }
}

// bias corrections of ONE tensor: its own step count when the host filled it in (steps differ inside the group), else the
// launch-wide one; with a device counter, counter + the tensor's offset
__device__ __forceinline__ void bias_corr(const CflTensorMeta& tm, const Hyper& h, float& bc1, float& bc2) {
    if (h.gstep) {
        const int s = max(1, *h.gstep + tm.step);
        bc1 = 1.f - powf(h.beta1, (float)s);
        bc2 = 1.f - powf(h.beta2, (float)s);
    } else if (tm.step > 0) {
        bc1 = 1.f - powf(h.beta1, (float)tm.step);
        bc2 = 1.f - powf(h.beta2, (float)tm.step);
    } else {
        bc1 = h.bc1; bc2 = h.bc2;
    }
}

// ---- pass 1 -------------------------------------------------------------------------------------------
__device__ __forceinline__ float adam_elem(float g, float& m, float& v, const Hyper& h, float rs_bc2) {
    m = h.beta1 * m + (1.f - h.beta1) * g;
    v = h.beta2 * v + (1.f - h.beta2) * g * g;
    const float denom = sqrtf(v) * rs_bc2 + h.eps;
    return (h.nesterov ? (h.beta1 * m + (1.f - h.beta1) * g) : m) / denom;
}

__global__ __launch_bounds__(256) void cfl_adamp_pass1_kernel(const CflTensorMeta* __restrict__ meta,
                                                              const int* __restrict__ items, Hyper h,
                                                              const float* __restrict__ clip, float* rowstats) {
    const int* it = items + (size_t)blockIdx.x * 3;
    const CflTensorMeta tm = meta[it[0]];
    float* p = (float*)tm.p;
    const void* g = tm.g;
    const bool bf = tm.flags & CFL_OPT_GRAD_BF16;
    float* m = (float*)tm.m;
    float* v = (float*)tm.v;
    const float cc = (clip && (tm.flags & CFL_OPT_CLIP)) ? clip[1] : 1.f;
    float bc1, bc2;
    bias_corr(tm, h, bc1, bc2);
    const float rs_bc2 = 1.f / sqrtf(bc2);
    const float step = h.lr / bc1;
    if (!(tm.flags & CFL_OPT_MATRIX)) {
        const long long e1 = (long long)it[1] + it[2];
        const float decay = 1.f - h.lr * h.wd;
        for (long long e = (long long)it[1] + threadIdx.x; e < e1; e += 256) {
            float mm = m[e], vv = v[e];
            const float pert = adam_elem(ldg1(g, e, bf) * cc, mm, vv, h, rs_bc2);
            m[e] = mm; v[e] = vv;
            const float pn = p[e] * decay - step * pert;
            p[e] = pn;
            if (tm.p16) st_shadow1(tm.p16, e, pn);
        }
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long inner = tm.inner;
    const bool vec = (inner & 3) == 0;
    for (int r = it[1] + w; r < it[1] + it[2]; r += 4) {
        const long long base = (long long)r * inner;
        float gp = 0.f, gg = 0.f, pp = 0.f, pq = 0.f;
        if (vec) {
            for (long long e = lane * 4; e < inner; e += 256) {
                const f32x4 pv = ld4(p + base + e);
                f32x4 gv = ldg4(g, base + e, bf) * cc, mv = ld4_nt(m + base + e), vv = ld4_nt(v + base + e);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float mm = mv[k], v2 = vv[k];
                    const float pert = adam_elem(gv[k], mm, v2, h, rs_bc2);
                    mv[k] = mm; vv[k] = v2;
                    gp = fmaf(gv[k], pv[k], gp); gg = fmaf(gv[k], gv[k], gg);
                    pp = fmaf(pv[k], pv[k], pp); pq = fmaf(pv[k], pert, pq);
                }
                st4_nt(m + base + e, mv); st4_nt(v + base + e, vv);
            }
        } else {
            for (long long e = lane; e < inner; e += 64) {
                const float pv = p[base + e], gv = ldg1(g, base + e, bf) * cc;
                float mm = m[base + e], v2 = v[base + e];
                const float pert = adam_elem(gv, mm, v2, h, rs_bc2);
                m[base + e] = mm; v[base + e] = v2;
                gp = fmaf(gv, pv, gp); gg = fmaf(gv, gv, gg); pp = fmaf(pv, pv, pp); pq = fmaf(pv, pert, pq);
            }
        }
        gp = wave_sum(gp); gg = wave_sum(gg); pp = wave_sum(pp); pq = wave_sum(pq);
        if (lane == 0) {
            float* rs = rowstats + (tm.row_base + r) * 4;
            rs[0] = gp; rs[1] = gg; rs[2] = pp; rs[3] = pq;
        }
    }
}

// ---- decide: one block per matrix tensor ------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfl_adamp_decide_kernel(const CflTensorMeta* __restrict__ meta,
                                                               const int* __restrict__ matrix_ids, Hyper h,
                                                               float* rowstats, float* tstats) {
    __shared__ float red[4];
    __shared__ float smax[4];
    const int t = matrix_ids[blockIdx.x];
    const CflTensorMeta tm = meta[t];
    float* rs = rowstats + tm.row_base * 4;
    float cmax = 0.f, GP = 0.f, GG = 0.f, PP = 0.f, PQ = 0.f;
    for (int r = threadIdx.x; r < tm.n0; r += 256) {
        const float gp = rs[r * 4 + 0], gg = rs[r * 4 + 1], pp = rs[r * 4 + 2], pq = rs[r * 4 + 3];
        const float c = fabsf(gp) / (fmaxf(sqrtf(gg), h.eps) * fmaxf(sqrtf(pp), h.eps));
        cmax = fmaxf(cmax, c);
        GP += gp; GG += gg; PP += pp; PQ += pq;
    }
    cmax = wave_max(cmax);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = cmax;
    __syncthreads();
    cmax = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    GP = block_sum_256(GP, red); GG = block_sum_256(GG, red); PP = block_sum_256(PP, red); PQ = block_sum_256(PQ, red);
    const bool hit_c = cmax < h.delta / sqrtf((float)tm.inner);
    const float cos_l = fabsf(GP) / (fmaxf(sqrtf(GG), h.eps) * fmaxf(sqrtf(PP), h.eps));
    const bool hit_l = !hit_c && (cos_l < h.delta / sqrtf((float)tm.numel));
    const float nl = sqrtf(PP) + h.eps;
    const float coef_l = PQ / (nl * nl);
    __syncthreads();
    for (int r = threadIdx.x; r < tm.n0; r += 256) {
        float coef = 0.f;
        if (hit_c) { const float nr = sqrtf(rs[r * 4 + 2]) + h.eps; coef = rs[r * 4 + 3] / (nr * nr); }
        else if (hit_l) coef = coef_l;
        rs[r * 4 + 0] = coef;
    }
    if (threadIdx.x == 0) tstats[t] = (hit_c || hit_l) ? h.wd_ratio : 1.f;
}

// ---- pass 3: matrix tensors only ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfl_adamp_pass3_kernel(const CflTensorMeta* __restrict__ meta,
                                                              const int* __restrict__ items, Hyper h,
                                                              const float* __restrict__ clip,
                                                              const float* __restrict__ rowstats,
                                                              const float* __restrict__ tstats) {
    const int* it = items + (size_t)blockIdx.x * 3;
    const CflTensorMeta tm = meta[it[0]];
    if (!(tm.flags & CFL_OPT_MATRIX)) return;
    float* p = (float*)tm.p;
    const void* g = tm.g;
    const bool bf = tm.flags & CFL_OPT_GRAD_BF16;
    const float* m = (const float*)tm.m;
    const float* v = (const float*)tm.v;
    const float cc = (clip && (tm.flags & CFL_OPT_CLIP)) ? clip[1] : 1.f;
    float bc1, bc2;
    bias_corr(tm, h, bc1, bc2);
    const float rs_bc2 = 1.f / sqrtf(bc2);
    const float step = h.lr / bc1;
    const float decay = 1.f - h.lr * h.wd * tstats[it[0]];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long inner = tm.inner;
    const bool vec = (inner & 3) == 0;
    for (int r = it[1] + w; r < it[1] + it[2]; r += 4) {
        const long long base = (long long)r * inner;
        const float coef = rowstats[(tm.row_base + r) * 4];
        if (vec) {
            for (long long e = lane * 4; e < inner; e += 256) {
                f32x4 pv = ld4(p + base + e);
                const f32x4 mv = ld4_nt(m + base + e), vv = ld4_nt(v + base + e);
                f32x4 gv = {0.f, 0.f, 0.f, 0.f};
                if (h.nesterov) gv = ldg4(g, base + e, bf) * cc;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float num = h.nesterov ? (h.beta1 * mv[k] + (1.f - h.beta1) * gv[k]) : mv[k];
                    const float pert = num / (sqrtf(vv[k]) * rs_bc2 + h.eps);
                    pv[k] = pv[k] * decay - step * (pert - pv[k] * coef);
                }
                st4(p + base + e, pv);
                if (tm.p16) st_shadow4(tm.p16, base + e, pv);
            }
        } else {
            for (long long e = lane; e < inner; e += 64) {
                const float pv = p[base + e];
                const float num = h.nesterov ? (h.beta1 * m[base + e] + (1.f - h.beta1) * ldg1(g, base + e, bf) * cc) : m[base + e];
                const float pert = num / (sqrtf(v[base + e]) * rs_bc2 + h.eps);
                const float pn = pv * decay - step * (pert - pv * coef);
                p[base + e] = pn;
                if (tm.p16) st_shadow1(tm.p16, base + e, pn);
            }
        }
    }
}

}  // namespace

static int adamp_launch(const CflTensorMeta* meta_dev, int n_tensors, const int* items_dev, int n_items,
                        const int* matrix_ids_dev, int n_matrix, float* rowstats_ws, float* tstats_ws,
                        float lr, float beta1, float beta2, float eps, float weight_decay, float delta,
                        float wd_ratio, int nesterov, int step, const int* gstep_dev, const float* clip_dev, void* stream_) {
    if (!meta_dev || !items_dev || n_tensors <= 0 || n_items <= 0 || step <= 0) return CFL_EINVAL;
    if (n_matrix > 0 && (!matrix_ids_dev || !rowstats_ws || !tstats_ws)) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    Hyper h;
    h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.lr = lr; h.wd = weight_decay; h.delta = delta;
    h.wd_ratio = wd_ratio; h.nesterov = nesterov; h.max_norm = 0.f; h.gstep = gstep_dev;
    h.bc1 = 1.f - powf(beta1, (float)step);
    h.bc2 = 1.f - powf(beta2, (float)step);
    CFL_LAUNCH(K_ADAMP_PASS1, cfl_adamp_pass1_kernel, dim3(n_items), dim3(256), 0, stream, meta_dev, items_dev, h, clip_dev,
               rowstats_ws);
    if (n_matrix > 0) {
        CFL_LAUNCH(K_ADAMP_DECIDE, cfl_adamp_decide_kernel, dim3(n_matrix), dim3(256), 0, stream, meta_dev, matrix_ids_dev, h,
                   rowstats_ws, tstats_ws);
        CFL_LAUNCH(K_ADAMP_PASS3, cfl_adamp_pass3_kernel, dim3(n_items), dim3(256), 0, stream, meta_dev, items_dev, h, clip_dev,
                   rowstats_ws, tstats_ws);
    }
    return 0;
}


extern "C" {

int cfl_grad_clip_coef(const CflTensorMeta* meta_dev, const int* items_dev, int n_items, float max_norm,
                       float* partial_ws, float* out2, void* stream_) {
    if (!meta_dev || !items_dev || !partial_ws || !out2 || n_items <= 0) return CFL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    CFL_LAUNCH(K_GRADNORM, cfl_gradnorm_partial_kernel, dim3(n_items), dim3(256), 0, stream, meta_dev, items_dev, partial_ws);
    CFL_LAUNCH(K_GRADNORM, cfl_gradnorm_final_kernel, dim3(1), dim3(256), 0, stream, partial_ws, n_items, max_norm, out2);
    return 0;
}

int cfl_adamp_step(const CflTensorMeta* meta_dev, int n_tensors, const int* items_dev, int n_items,
                   const int* matrix_ids_dev, int n_matrix, float* rowstats_ws, float* tstats_ws,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float delta,
                   float wd_ratio, int nesterov, int step, const float* clip_dev, void* stream_) {
    return adamp_launch(meta_dev, n_tensors, items_dev, n_items, matrix_ids_dev, n_matrix, rowstats_ws, tstats_ws, lr, beta1, beta2,
                        eps, weight_decay, delta, wd_ratio, nesterov, step, nullptr, clip_dev, stream_);
}

int cfl_adamp_step_counted(const CflTensorMeta* meta_dev, int n_tensors, const int* items_dev, int n_items,
                           const int* matrix_ids_dev, int n_matrix, float* rowstats_ws, float* tstats_ws,
                           float lr, float beta1, float beta2, float eps, float weight_decay, float delta,
                           float wd_ratio, int nesterov, const int* gstep_dev, const float* clip_dev, void* stream_) {
    if (!gstep_dev) return CFL_EINVAL;
    return adamp_launch(meta_dev, n_tensors, items_dev, n_items, matrix_ids_dev, n_matrix, rowstats_ws, tstats_ws, lr, beta1, beta2,
                        eps, weight_decay, delta, wd_ratio, nesterov, 1, gstep_dev, clip_dev, stream_);
}

}  // extern "C"
