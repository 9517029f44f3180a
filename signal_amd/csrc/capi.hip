// extern "C" surface of libsignal_hip.so (declared in include/signal_hip.h).
#include <stdarg.h>

#include "../../include/signal_hip.h"
#include "sig_common.h"
#include "sig_kernels.h"

static thread_local char g_err[512] = "";

void sig_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static_assert(SIG_GEMM_F32 == SIG_EPI_F32 && SIG_GEMM_BF16 == SIG_EPI_BF16 && SIG_GEMM_BIAS_F32 == SIG_EPI_BIAS_F32 &&
                  SIG_GEMM_BIAS_BF16 == SIG_EPI_BIAS_BF16 && SIG_GEMM_BIAS_RES_F32 == SIG_EPI_BIAS_RES_F32 &&
                  SIG_GEMM_BIAS_GELU_BF16 == SIG_EPI_BIAS_GELU_BF16 && SIG_GEMM_DGELU_BF16 == SIG_EPI_DGELU_BF16 &&
                  SIG_GEMM_BIAS_GELUERF_BF16 == SIG_EPI_BIAS_GELUERF_BF16 && SIG_GEMM_DGELUERF_BF16 == SIG_EPI_DGELUERF_BF16 &&
                  SIG_GEMM_RES_F32 == SIG_EPI_RES_F32,
              "public and internal epilogue ids must agree");

extern "C" {

const char* sig_last_error(void) { return g_err; }
int sig_version(void) { return SIG_ABI_VERSION; }
int sig_prof_begin(int epilogue, int N, int K, int max_launches) { return sig_prof_begin_impl(epilogue, N, K, max_launches); }
int sig_tune_gemm_tile(int tile) { return sig_tune_gemm_tile_impl(tile); }
int sig_tune_nt_persist(int on) { return sig_tune_nt_persist_impl(on); }
int sig_tune_ln_defer(int on) { return sig_tune_ln_defer_impl(on); }
int sig_tune_tn_overwrite(int on) { return sig_tune_tn_overwrite_impl(on); }
int sig_tune_attn_bwd_waves(int waves) { return sig_tune_attn_bwd_waves_impl(waves); }
int sig_tune_attn_fwd_waves(int waves) { return sig_tune_attn_fwd_waves_impl(waves); }
int sig_ln_flush(void* stream) { return sig_ln_flush_impl((hipStream_t)stream); }
int sig_tune_reserved_cus(int n) { return sig_tune_reserved_cus_impl(n); }
int sig_tune_tn_path(int path) { return sig_tune_tn_path_impl(path); }
int sig_debug_tn_plan(int tiles, int ks, int grid, int cs_units, int* out8) { return sig_debug_tn_plan_impl(tiles, ks, grid, cs_units, out8); }
int sig_prof_end(double* total_ms, int* launches, double* flops) { return sig_prof_end_impl(total_ms, launches, flops); }

int sig_gemm_nt(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, int M, int N, int K, int epilogue, void* out,
                int ldo, const float* bias, const float* res, int ldr, void* aux, int ldaux, int dtype, void* stream) {
    SigGemmNT p;
    p.A = A; p.Bt = Bt; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K;
    p.out = out; p.ldo = ldo; p.bias = bias; p.res = res; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux; p.band = 0; p.colsum = nullptr; p.dt = dtype;
    return sig_launch_gemm_nt(p, epilogue, (hipStream_t)stream);
}

int sig_gemm_tn(const uint16_t* P, int ldp, const uint16_t* Q, int ldq, int Mr, int I, int J, float* out, int ldo,
                int split, int dtype, void* stream) {
    SigGemmTN p;
    p.P = P; p.Q = Q; p.ldp = ldp; p.ldq = ldq; p.Mr = Mr; p.I = I; p.J = J; p.out = out; p.ldo = ldo;
    p.split = split; p.m_chunk = 0; p.ws = nullptr; p.dt = dtype;
    return sig_launch_gemm_tn(p, (hipStream_t)stream);
}

int sig_gemm_tn_grouped(const SigTnJobDesc* jobs, int n, int Mr, int dtype, void* stream) {
    SIG_CHECK_ARG(jobs && n >= 1 && n <= SIG_TN_MAX_JOBS, "gemm_tn_grouped: 1..%d jobs", SIG_TN_MAX_JOBS);
    SigTnJob j[SIG_TN_MAX_JOBS];
    for (int k = 0; k < n; ++k) {
        j[k].P = jobs[k].P; j[k].Q = jobs[k].Q; j[k].out = jobs[k].out;
        j[k].ldp = jobs[k].ldp; j[k].ldq = jobs[k].ldq; j[k].ldo = jobs[k].ldo; j[k].I = jobs[k].I; j[k].J = jobs[k].J;
        j[k].colsum = jobs[k].colsum;
        j[k].cs_rows = 0;          // public contract: P's pad rows are zero, the column sums run over all Mr rows
    }
    return sig_launch_gemm_tn_grouped(j, n, Mr, dtype, (hipStream_t)stream);
}

int sig_comm_unique_id(void* id128) { return sig_comm_unique_id_impl(id128); }
int sig_comm_init(SigComm** comm, int rank, int world, const void* id128) { return sig_comm_init_impl(comm, rank, world, id128); }
int sig_comm_allreduce_async(SigComm* comm, float* buf, size_t count, void* compute_stream) {
    return sig_comm_allreduce_async_impl(comm, buf, count, (hipStream_t)compute_stream);
}
int sig_comm_wait(SigComm* comm, void* stream) { return sig_comm_wait_impl(comm, (hipStream_t)stream); }
int sig_comm_destroy(SigComm* comm) { return sig_comm_destroy_impl(comm); }

int sig_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y_bf16, float* y_f32, float* mean,
                      float* rstd, int M, int D, float eps, int dtype, void* stream) {
    return sig_launch_layernorm_fwd(x, gamma, beta, y_bf16, y_f32, mean, rstd, M, D, eps, dtype, (hipStream_t)stream);
}

int sig_layernorm_bwd(const void* dy, int dy_is_bf16, const float* x, const float* gamma, const float* mean,
                      const float* rstd, const float* dres, float* dx_f32, uint16_t* dx_bf16, float* dgamma, float* dbeta,
                      int M, int D, int dtype, void* stream) {
    return sig_launch_layernorm_bwd(dy, dy_is_bf16, x, gamma, mean, rstd, dres, dx_f32, dx_bf16, dgamma, dbeta, M, D, dtype,
                                    (hipStream_t)stream);
}

int sig_attn_fwd(const uint16_t* qkv, uint16_t* out, float* lse, int S, int L, int H, int dtype, void* stream) {
    return sig_launch_attn_fwd(qkv, out, lse, S, L, H, dtype, (hipStream_t)stream);
}

int sig_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, uint16_t* dqkv, int S,
                 int L, int H, int dtype, void* stream) {
    return sig_launch_attn_bwd(qkv, out, dout, lse, dqkv, S, L, H, dtype, (hipStream_t)stream);
}

int sig_cast_bf16(const float* src, uint16_t* dst, size_t n, int dtype, void* stream) {
    return sig_launch_cast_bf16(src, dst, n, dtype, (hipStream_t)stream);
}
int sig_transpose_cast_bf16(const float* src, uint16_t* dst, int rows, int cols, int dtype, void* stream) {
    return sig_launch_transpose_cast_bf16(src, dst, rows, cols, dtype, (hipStream_t)stream);
}
int sig_transpose16_multi(const int64_t* table, const int* tile_start, int n, int total_tiles, void* stream) {
    return sig_launch_transpose16_multi((const long long*)table, tile_start, n, total_tiles, (hipStream_t)stream);
}
int sig_transpose_cast_multi(const int64_t* table, const int* tile_start, int n, int total_tiles, int dtype, void* stream) {
    return sig_launch_transpose_cast_multi((const long long*)table, tile_start, n, total_tiles, dtype, (hipStream_t)stream);
}
int sig_colsum_bf16(const uint16_t* a, int lda, int M, int N, float* out, int dtype, void* stream) {
    return sig_launch_colsum_bf16(a, lda, M, N, out, dtype, (hipStream_t)stream);
}
int sig_colsum_f32(const float* a, int lda, int M, int N, float* out, void* stream) {
    return sig_launch_colsum_f32(a, lda, M, N, out, (hipStream_t)stream);
}

int sig_im2col(const float* img, uint16_t* out, int nimg, int H, int W, int P, int dtype, void* stream) {
    return sig_launch_im2col(img, out, nimg, H, W, P, dtype, (hipStream_t)stream);
}
int sig_embed_assemble(const float* tok, const float* class_embedding, const float* positional_embedding,
                       const float* cv_embed, const int64_t* cam_label, float sie_coe, const float* ln_w,
                       const float* ln_b, float* x, float* pre_ln, float* mean, float* rstd, int S, int B, int L, int D,
                       float eps, void* stream) {
    return sig_launch_embed_assemble(tok, class_embedding, positional_embedding, cv_embed, cam_label, sie_coe, ln_w, ln_b,
                                     x, pre_ln, mean, rstd, S, B, L, D, eps, (hipStream_t)stream);
}
int sig_embed_assemble_bwd(const float* d_pre_ln, float* dtok_f32, uint16_t* dtok_bf16, float* d_class_embedding,
                  float* d_positional_embedding, float* d_cv_embed, const int64_t* cam_label, float sie_coe, int S, int B,
                  int L, int D, int dtype, void* stream) {
    return sig_launch_embed_bwd(d_pre_ln, dtok_f32, dtok_bf16, d_class_embedding, d_positional_embedding, d_cv_embed,
                                cam_label, sie_coe, S, B, L, D, dtype, (hipStream_t)stream);
}

int sig_adam_step(float* p, const float* g, float* m, float* v, uint16_t* p16, int dtype, const int* seg_end, const float* seg_lr,
                  const float* seg_wd, int nseg, float beta1, float beta2, float eps, int step, float grad_scale,
                  const float* scale_state, size_t n, void* stream) {
    return sig_launch_adam(p, g, m, v, p16, seg_end, seg_lr, seg_wd, nseg, beta1, beta2, eps, step, grad_scale, scale_state, dtype, n,
                           (hipStream_t)stream);
}
int sig_zero_ranges(float* base, const int64_t* table, const int* chunk_start, int n, int total_chunks, void* stream) {
    return sig_launch_zero_ranges(base, (const long long*)table, chunk_start, n, total_chunks, (hipStream_t)stream);
}
int sig_grad_check(const float* g, size_t n, float* scale_state, void* stream) {
    return sig_launch_grad_check(g, n, scale_state, (hipStream_t)stream);
}
int sig_loss_scale_update(float* scale_state, float growth_factor, float backoff_factor, int growth_interval, void* stream) {
    return sig_launch_loss_scale_update(scale_state, growth_factor, backoff_factor, growth_interval, (hipStream_t)stream);
}

int sig_bnneck_fwd(const float* x, const float* bn_w, const float* bn_b, float* running_mean, float* running_var, float momentum,
                   const float* cls_w, int B, int F, int C, float* y, float* mean, float* rstd, float* logits, void* stream) {
    return sig_launch_bnneck_fwd(x, bn_w, bn_b, running_mean, running_var, momentum, cls_w, B, F, C, y, mean, rstd, logits,
                                 (hipStream_t)stream);
}
int sig_bnneck_bwd(const float* x, const float* y, const float* bn_w, const float* mean, const float* rstd, const float* cls_w,
                   const float* dlogits, int B, int F, int C, float* dy_scratch, float* dx, float* dbn_w, float* dbn_b, float* dcls_w,
                   void* stream) {
    return sig_launch_bnneck_bwd(x, y, bn_w, mean, rstd, cls_w, dlogits, B, F, C, dy_scratch, dx, dbn_w, dbn_b, dcls_w,
                                 (hipStream_t)stream);
}
int sig_reid_loss(const float* logits, const float* feat, const int64_t* target, int B, int F, int C, float label_smooth_eps,
                  float id_weight, float triplet_weight, float margin, const float* upstream, float* loss, float* dlogits,
                  float* gram, int* pidx, int* nidx, float* coef, float* dfeat, void* stream) {
    return sig_launch_reid_loss(logits, feat, target, B, F, C, label_smooth_eps, id_weight, triplet_weight, margin, upstream, loss,
                                dlogits, gram, pidx, nidx, coef, dfeat, (hipStream_t)stream);
}

}  // extern "C"
