// runtime.hip -- library identity + per-kernel HIP-event profiler behind the C ABI.
#include <mutex>
#include <vector>
#include <stdlib.h>
#include "common.h"

static const char* const g_kernel_names[K_NUM] = {
    "cfl_pair_prep_kernel", "cfl_pair_fwd_kernel", "cfl_pair_final_kernel", "cfl_pair_bwd_kernel",
    "cfl_bank_fwd_kernel", "cfl_lse_final_kernel", "cfl_bank_loss_kernel", "cfl_bank_bwd_kernel",
    "cfl_bank_bwd_reduce_kernel", "cfl_intra_kernel", "cfl_conw_combine_kernel",
    "cfl_pie_scores_kernel", "cfl_pie_pool_kernel", "cfl_pie_bwd_ds_kernel", "cfl_pie_bwd_dx_kernel",
    "cfl_pie_bwd_dh_kernel", "cfl_pie_bwd_dw2_kernel", "cfl_pie_epi_fwd_kernel",
    "cfl_pie_epi_bwd_kernel", "cfl_pie_epi_bwd_ln_kernel", "cfl_l2norm_fwd_kernel",
    "cfl_l2norm_bwd_kernel", "cfl_rank_posmax_kernel", "cfl_rank_count_kernel",
    "cfl_gradnorm_kernel", "cfl_adamp_pass1_kernel", "cfl_adamp_decide_kernel", "cfl_adamp_pass3_kernel",
    "cfl_bn_stats_kernel", "cfl_bn_final_kernel", "cfl_bn_apply_kernel", "cfl_bn_bwd_reduce_kernel",
    "cfl_bn_bwd_final_kernel", "cfl_bn_bwd_apply_kernel", "cfl_gemm_nt_kernel",
    "cfl_kd_mse_kernel", "cfl_sup_glue_kernel", "cfl_gemm_bf16_kernel", "cfl_bert_daln_kernel", "cfl_bert_gelu_kernel", "cfl_attn_small_kernel", "cfl_maxpool_kernel", "cfl_transpose_bf16_kernel",
    "cfl_pie_fwd_fused_kernel", "cfl_pie_bwd_fused_kernel",
    "cfl_bn_pool_fwd_kernel", "cfl_bn_pool_bwd_reduce_kernel", "cfl_bn_pool_bwd_apply_kernel",
    "cfl_bank_image_kernel", "cfl_bank_stream_kernel",
    "cfl_bn_bwd_apply_wgrad_kernel", "cfl_bn_wgrad_reduce_kernel",
    "cfl_gru_fwd_kernel", "cfl_gru_bwd_kernel", "cfl_gru_cell0_kernel",
    "cfl_conv3x3_wgrad_kernel", "cfl_conv3x3_wgrad_reduce_kernel", "cfl_conv1x1_wgrad_kernel", "cfl_conv1x1_wgrad_reduce_kernel",
    "cfl_conv3x3_x3_kernel", "cfl_conv3x3_x3_wgrad_kernel", "cfl_conv3x3_x3_wgrad_reduce_kernel", "cfl_conv3x3_wimage_kernel",
    "cfl_pair_bwd_reduce_kernel",
};

namespace {
struct Pending { int id; hipEvent_t e0, e1; };
std::mutex g_mu;
bool g_on = false;
int g_only = -1;              // profile only this kernel id (-1 = every kernel)
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_pool;
long long g_launches[K_NUM];
double g_ms[K_NUM];

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void drain_locked() {
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            g_launches[p.id] += 1;
            g_ms[p.id] += (double)ms;
        }
        g_pool.push_back(p.e0);
        g_pool.push_back(p.e1);
    }
    g_pending.clear();
}
}  // namespace

bool cfl_prof_begin(int id, hipEvent_t* e0, hipEvent_t* e1) {
    if (!g_on || (g_only >= 0 && g_only != id)) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    *e0 = get_event();
    *e1 = get_event();
    if (*e0 && *e1) return true;
    // an incomplete pair is not used: hand the event that WAS obtained back to the pool (it used to leak)
    if (*e0) g_pool.push_back(*e0);
    if (*e1) g_pool.push_back(*e1);
    *e0 = *e1 = nullptr;
    return false;
}
void cfl_prof_end(int id, hipEvent_t e0, hipEvent_t e1) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_pending.push_back({id, e0, e1});
}

static int g_exact_gemm = -1;        // -1: not read yet

extern "C" {
// Dense kernels that exist in two forms (pair loss, two-pass bank kernels, con_w log-prob, the GEMM probe):
// 0 = 3 x bf16-split MFMA (csrc/tile_x3.h, default), 1 = exact fp32 MFMA (csrc/common.h).  Initialised from the
// environment variable CFL_GEMM_EXACT on first use.
int cfl_get_exact_gemm(void) {
    if (g_exact_gemm < 0) {
        const char* e = getenv("CFL_GEMM_EXACT");
        g_exact_gemm = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return g_exact_gemm;
}
int cfl_set_exact_gemm(int exact) { g_exact_gemm = exact ? 1 : 0; return 0; }
int cfl_version(void) { return 100; }
const char* cfl_arch(void) { return "gfx950"; }
int cfl_num_kernels(void) { return K_NUM; }
const char* cfl_kernel_name(int id) { return (id >= 0 && id < K_NUM) ? g_kernel_names[id] : ""; }

int cfl_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return 0;
}
int cfl_prof_select(int kernel_id) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_only = (kernel_id >= 0 && kernel_id < K_NUM) ? kernel_id : -1;
    return 0;
}
int cfl_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    for (int i = 0; i < K_NUM; ++i) { g_launches[i] = 0; g_ms[i] = 0.0; }
    return 0;
}
int cfl_prof_query(int id, long long* launches, double* total_ms) {
    if (id < 0 || id >= K_NUM) return CFL_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    if (launches) *launches = g_launches[id];
    if (total_ms) *total_ms = g_ms[id];
    return 0;
}
}
