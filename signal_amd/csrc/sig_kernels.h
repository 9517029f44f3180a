// Internal launcher interface between the kernel translation units and capi.hip.
#pragma once
#include "sig_common.h"

// ---- GEMM ---------------------------------------------------------------------------------------
enum SigEpilogue {
    SIG_EPI_F32 = 0,            // out f32 = acc
    SIG_EPI_BF16 = 1,           // out bf16 = acc
    SIG_EPI_BIAS_F32 = 2,       // out f32 = acc + bias
    SIG_EPI_BIAS_BF16 = 3,      // out bf16 = acc + bias
    SIG_EPI_BIAS_RES_F32 = 4,   // out f32 = acc + bias + res   (res may alias out)
    SIG_EPI_BIAS_GELU_BF16 = 5, // out 16-bit = QuickGELU(acc + bias), aux 16-bit = QuickGELU'(acc + bias) (if aux)
    SIG_EPI_DGELU_BF16 = 6,     // out 16-bit = acc * aux
    SIG_EPI_BIAS_GELUERF_BF16 = 7, // aux bf16 = acc + bias (if aux), out bf16 = GELU_erf(acc + bias)
    SIG_EPI_DGELUERF_BF16 = 8,  // out bf16 = acc * GELU_erf'(aux)
    SIG_EPI_RES_F32 = 9,        // out f32 = acc + res
};

struct SigGemmNT {
    const bf16_t* A;   // [Mpad, lda]  rows >= M readable (buffers are padded to 128 rows)
    const bf16_t* Bt;  // [N, ldb]
    int lda, ldb;
    int M, N, K;
    void* out;         // f32 or bf16 [M, ldo]
    int ldo;
    const float* bias; // [N]
    const float* res;  // [M, ldr] f32
    int ldr;
    void* aux;         // bf16 [M, ldaux]
    int ldaux;
    int band;          // column tiles per L2-resident weight band (filled by the launcher)
    float* colsum;     // optional [N]: += column sums of the (f32, pre-rounding) output over the valid rows
    int dt;            // SIG_DT_BF16 / SIG_DT_F16: type of A, Bt, aux and of 16-bit outputs
};
int sig_launch_gemm_nt(const SigGemmNT& p, int epi, hipStream_t st);
// persistent 192x256 kernel with the store tail of tile n under the main loop of tile n+1 (gemm_nt_persist.hip)
bool sig_nt192p_eligible(const SigGemmNT& p, int epi, int cus);
int sig_launch_nt192p(const SigGemmNT& p, int epi, int cus, hipStream_t st);
int sig_tune_nt_persist_impl(int on);
#ifndef SIG_PROF_TN256
#define SIG_PROF_TN256 100   // sig_prof_begin class: gemm_tn256_kernel launches (N = I, K = J; 0 = any shape)
#endif
int sig_prof_begin_impl(int epi, int N, int K, int max_launches);
int sig_prof_end_impl(double* total_ms, int* launches, double* flops);
int sig_tune_gemm_tile_impl(int tile);
int sig_tune_reserved_cus_impl(int n);
int sig_tune_tn_path_impl(int path);
int sig_tn_path();

struct SigGemmTN {
    const bf16_t* P;  // [Mr, ldp], columns I
    const bf16_t* Q;  // [Mr, ldq], columns J
    int ldp, ldq;
    int Mr, I, J;     // Mr multiple of 64; pad rows must be zero in P or Q
    float* out;       // [I, ldo] f32, accumulated with atomics
    int ldo;
    int split;        // 0 = choose
    int m_chunk;      // filled by the launcher
    float* ws;        // filled by the launcher: per-block partial tiles (256x256 kernel), nullptr = atomics into out
    int dt;           // SIG_DT_BF16 / SIG_DT_F16: type of P and Q
};
int sig_launch_gemm_tn(const SigGemmTN& p, hipStream_t st);
// several weight gradients that share the row count, as ONE stream-K launch + one reduce (gemm_tn_grouped.hip)
#define SIG_TN_MAX_JOBS 4
struct SigTnJob {
    const bf16_t* P;  // dY [Mr, ldp], columns I
    const bf16_t* Q;  // X  [Mr, ldq], columns J
    float* out;       // dW [I, ldo] f32, +=
    int ldp, ldq, ldo, I, J;
    float* colsum;    // optional [I]: += column sums of P over the rows (the bias gradient that goes with dW), else nullptr
    int cs_rows;      // rows the column sums run over (0 = all Mr): the VALID rows when P's pad rows may hold stale data -- the GEMM
                      // itself is protected by Q's zero pad rows, a column sum of P is not
};
int sig_launch_gemm_tn_grouped(const SigTnJob* jobs, int njobs, int Mr, int dt, hipStream_t st);
int sig_debug_tn_plan_impl(int tiles, int ks, int grid, int cs_units, int* out);
int sig_tune_tn_overwrite_impl(int on);                       // grouped weight gradients overwrite dW instead of adding to it
int sig_free_cus();                                            // 256 minus the CUs reserved for RCCL (sig_tune_reserved_cus)
#ifndef SIG_PROF_TN_GROUP
#define SIG_PROF_TN_GROUP 101   // sig_prof_begin class: gemm_tn_group_kernel launches (a block's four weight gradients)
#endif
bool sig_prof_tn_start(hipStream_t st, int cls, int I, int J);   // bench.py's roofline leg
void sig_prof_tn_stop(hipStream_t st, double flops);
// library-owned scratch per (device, stream, slot): nullptr when it cannot be had (callers then fall back to atomics)
float* sig_stream_scratch(hipStream_t st, size_t bytes, int slot);

// ---- row-wise kernels (rowops.hip) ----------------------------------------------------------------
int sig_launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, bf16_t* y_bf16, float* y_f32,
                             float* mean, float* rstd, int M, int D, float eps, int dt, hipStream_t st);
int sig_launch_layernorm_bwd(const void* dy, int dy_is_bf16, const float* x, const float* gamma, const float* mean,
                             const float* rstd, const float* dres, float* dx_f32, bf16_t* dx_bf16, float* dgamma,
                             float* dbeta, int M, int D, int dt, hipStream_t st, float* dx_colsum = nullptr);
int sig_tune_attn_fwd_waves_impl(int waves);   // 9 (default) or 3 waves per block of the attention forward at L in (128, 144]
int sig_tune_attn_bwd_waves_impl(int waves);   // 8 (default) or 4 waves per block of the L = 129 attention backward (attention.hip)
int sig_tune_ln_defer_impl(int on);           // chained column reduce of consecutive LayerNorm backwards (rowops.hip)
int sig_ln_flush_impl(hipStream_t st);
int sig_launch_cast_bf16(const float* src, bf16_t* dst, size_t n, int dt, hipStream_t st);
int sig_launch_transpose_cast_bf16(const float* src, bf16_t* dst, int rows, int cols, int dt, hipStream_t st);
int sig_launch_transpose_cast_multi(const long long* table, const int* tile_start, int n, int total_tiles, int dt, hipStream_t st);
int sig_launch_transpose16_multi(const long long* table, const int* tile_start, int n, int total_tiles, hipStream_t st);
int sig_launch_colsum_bf16(const bf16_t* a, int lda, int M, int N, float* out, int dt, hipStream_t st);
int sig_launch_colsum_f32(const float* a, int lda, int M, int N, float* out, hipStream_t st);
int sig_launch_im2col(const float* img, bf16_t* out, int nimg, int H, int W, int P, int dt, hipStream_t st);
int sig_launch_embed_assemble(const float* tok, const float* cls_emb, const float* pos, const float* cv_embed,
                              const int64_t* cam, float sie_coe, const float* g, const float* b, float* x,
                              float* pre_ln, float* mean, float* rstd, int S, int B, int L, int D, float eps,
                              hipStream_t st);
int sig_launch_embed_bwd(const float* dx_pre, float* dtok_f32, bf16_t* dtok_bf16, float* dcls, float* dpos,
                         float* dcv, const int64_t* cam, float sie_coe, int S, int B, int L, int D, int dt, hipStream_t st);

// ---- attention (attention.hip) ---------------------------------------------------------------------
int sig_launch_attn_fwd(const bf16_t* qkv, bf16_t* out, float* lse, int S, int L, int H, int dt, hipStream_t st);
// ---- SIM (sim.hip) ------------------------------------------------------------------------------------
int sig_launch_sim_select(const float* tokens, int B, int L, const float* Wq, const float* bq, const float* Wk,
                          const float* bk, int topk, int max_keep, float* qprime, float* cconst, float* intra, float* inter,
                          float* mask_f, unsigned char* mask_u8, hipStream_t st);
int sig_launch_sim_gather(const float* tokens, const float* mask_f, int B, int L, bf16_t* sel, bf16_t* cls_b, float* cls_f,
                          int dt, hipStream_t st);
int sig_launch_sim_gather_bwd(const bf16_t* dsel, const float* dcls, const float* mask_f, int B, int L, float* dtokens,
                              int dt, hipStream_t st);
int sig_launch_xattn_fwd(const float* q, const bf16_t* kv, int B, int NK, bf16_t* out, float* probs, int dt, hipStream_t st);
int sig_launch_xattn_bwd(const float* q, const bf16_t* kv, const float* probs, const float* dout, int B, int NK, float* dq,
                         bf16_t* dkv, int dt, hipStream_t st);

int sig_launch_attn_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv,
                        int S, int L, int H, int dt, hipStream_t st);

// ---- GAM / LAM (align.hip) ------------------------------------------------------------------------------
int sig_launch_gam_fwd(const float* tokens, int B, int L, const float* temp, float* fh, float* nrm, float* lv, float* la,
                       float* vec, float* coef, float* loss, hipStream_t st);
int sig_launch_gam_bwd(const float* fh, const float* nrm, const float* coef, const float* dloss, int B, int L, float* dtokens,
                       float* dtemp, hipStream_t st);
int sig_launch_lam_gather(const float* tokens, int B, int L, bf16_t* xb, size_t xstride, int dt, hipStream_t st);
int sig_launch_lam_scatter_add(const float* src, int m, int B, int L, float* dtokens, hipStream_t st);
// per-modality parameter / gradient pointers of the DAS tails (DAS_r, DAS_n, DAS_t); passed by value to the kernels
struct SigLamTailPtrs { const float *wd[3], *bd[3], *w4[3]; float *dwd[3], *dbd[3], *dw4[3]; };
// all three modalities in one launch (grid (B, 3)); a_stride = elements between the modalities' [Rp, 512] activation blocks
int sig_launch_lam_tail_fwd(const float* tokens, int B, int L, int h, int w, const bf16_t* a1, size_t a_stride, const SigLamTailPtrs& tp,
                            float* a2pre, float* offs, float* samp, int dt, hipStream_t st);
int sig_launch_lam_loss(const float* samp, size_t n, float* loss, hipStream_t st);
int sig_launch_lam_tail_bwd(const float* tokens, int B, int L, int h, int w, const bf16_t* a1, const bf16_t* a1pre, size_t a_stride,
                            const SigLamTailPtrs& tp, const float* a2pre, const float* offs, const float* samp_all, size_t nsamp,
                            const float* dloss, bf16_t* da1pre, float* dtokens, int dt, hipStream_t st, float* partials = nullptr);

// ---- optimizer (optim.hip) ------------------------------------------------------------------------------
int sig_launch_adam(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, const int* seg_end, const float* seg_lr,
                    const float* seg_wd, int nseg, float b1, float b2, float eps, int step, float gscale, const float* scale_state, int dt,
                    size_t n, hipStream_t st);
// fp16 loss scaling (engine/processor.py:119,259-261): state = [scale, 1/scale, found_inf, growth_tracker, applied_steps]
int sig_launch_grad_check(const float* g, size_t n, float* state, hipStream_t st);
int sig_launch_zero_ranges(float* base, const long long* table, const int* chunk_start, int n, int total_chunks, hipStream_t st);
int sig_launch_loss_scale_update(float* state, float growth, float backoff, int interval, hipStream_t st);

// ---- ReID head (reid.hip) ----------------------------------------------------------------------------------
int sig_launch_bnneck_fwd(const float* x, const float* bn_w, const float* bn_b, float* run_mean, float* run_var, float momentum,
                          const float* cls_w, int B, int F, int C, float* y, float* mean, float* rstd, float* logits, hipStream_t st);
int sig_launch_bnneck_bwd(const float* x, const float* y, const float* bn_w, const float* mean, const float* rstd, const float* cls_w,
                          const float* dlogits, int B, int F, int C, float* dy_scratch, float* dx, float* dbn_w, float* dbn_b, float* dcls_w,
                          hipStream_t st);
int sig_launch_reid_loss(const float* logits, const float* feat, const int64_t* target, int B, int F, int C, float eps, float w_id,
                         float w_tri, float margin, const float* upstream, float* loss, float* dlogits, float* gram, int* pidx, int* nidx,
                         float* coef, float* dfeat, hipStream_t st);

// ---- collective (comm.hip) -------------------------------------------------------------------------------------
struct SigComm;
int sig_comm_unique_id_impl(void* id128);
int sig_comm_init_impl(SigComm** out, int rank, int world, const void* id128);
int sig_comm_allreduce_async_impl(SigComm* c, float* buf, size_t count, hipStream_t compute_stream);
int sig_comm_wait_impl(SigComm* c, hipStream_t stream);
int sig_comm_destroy_impl(SigComm* c);
