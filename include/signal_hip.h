/* signal_hip.h -- C ABI of libsignal_hip.so, the MI355X (gfx950) implementation of the Signal hot path.
 *
 * The reference (maxingan2412/Signal) has no native layer: every stage below is an ATen call reached from
 * Python.  Each entry point therefore cites the reference Python it replaces (paths relative to the
 * reference root); INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer the caller owns (PyTorch caching allocator); the library borrows
 *     it for the duration of the call, keeps no reference and never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream); calls are re-entrant;
 *   - bf16 tensors are raw uint16_t bit patterns; "rows padded" means the allocation holds
 *     ceil(rows/128)*128 rows and the pad rows are zero (the library never writes them);
 *   - return 0 on success; non-zero = argument (1) or launch (2) error, message via sig_last_error().
 */
#ifndef SIGNAL_HIP_H
#define SIGNAL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIG_ABI_VERSION 1

const char* sig_last_error(void);
int sig_version(void);

/* ---- epilogues of sig_gemm_nt ---------------------------------------------------------------- */
enum {
    SIG_GEMM_F32 = 0,           /* out f32  = acc                                   */
    SIG_GEMM_BF16 = 1,          /* out bf16 = acc                                   */
    SIG_GEMM_BIAS_F32 = 2,      /* out f32  = acc + bias                            */
    SIG_GEMM_BIAS_BF16 = 3,     /* out bf16 = acc + bias                            */
    SIG_GEMM_BIAS_RES_F32 = 4,  /* out f32  = acc + bias + res  (res may alias out) */
    SIG_GEMM_BIAS_GELU_BF16 = 5,/* aux bf16 = acc + bias (if aux), out bf16 = QuickGELU(acc + bias) */
    SIG_GEMM_DGELU_BF16 = 6     /* out bf16 = acc * QuickGELU'(aux)                 */
};

/* out[M,N] = A[M,K] * Bt[N,K]^T (+ epilogue).  Every dense layer of the path: nn.Linear / MHA in-proj /
 * out-proj / `x @ proj` (modeling/clip/model.py:174-178,225,487; AddModule/useA.py:123-124,351-358) and
 * their input gradients (Bt = the transposed weight).  A rows padded; N % 128 == 0, K % 64 == 0. */
int sig_gemm_nt(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, int M, int N, int K, int epilogue,
                void* out, int ldo, const float* bias, const float* res, int ldr, void* aux, int ldaux,
                void* stream);

/* out[I,J] += P[Mr,I]^T * Q[Mr,J]  (f32, atomically accumulated): weight gradients dW = dY^T X that autograd
 * computes for the same layers.  Mr % 64 == 0 with zero pad rows; I, J % 128 == 0; split = 0 lets the
 * library choose the number of row chunks. */
int sig_gemm_tn(const uint16_t* P, int ldp, const uint16_t* Q, int ldq, int Mr, int I, int J, float* out,
                int ldo, int split, void* stream);

/* LayerNorm (fp32 statistics, eps as given): modeling/clip/model.py:154-160, AddModule/useA.py:414-423.
 * y_bf16 / y_f32 / mean / rstd may be NULL when not wanted. */
int sig_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y_bf16, float* y_f32,
                      float* mean, float* rstd, int M, int D, float eps, void* stream);
/* dx = dres + LN'(dy); dgamma/dbeta are ACCUMULATED (atomics) and may both be NULL. */
int sig_layernorm_bwd(const void* dy, int dy_is_bf16, const float* x, const float* gamma, const float* mean,
                      const float* rstd, const float* dres, float* dx_f32, uint16_t* dx_bf16, float* dgamma,
                      float* dbeta, int M, int D, void* stream);

/* Self-attention of nn.MultiheadAttention as called at modeling/clip/model.py:223-225 (no mask, no dropout):
 * qkv bf16 [S*L, 3*H*64] packed (q|k|v, head h = columns 64h..64h+63 of each) -> out bf16 [S*L, H*64],
 * lse f32 [S,H,L] (log-sum-exp of the scaled scores, kept for backward).  L <= 144. */
int sig_attn_fwd(const uint16_t* qkv, uint16_t* out, float* lse, int S, int L, int H, void* stream);
int sig_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse,
                 uint16_t* dqkv, int S, int L, int H, void* stream);

/* Packing helpers: f32 -> bf16 (optionally transposed), column sums (bias gradients, accumulated). */
int sig_cast_bf16(const float* src, uint16_t* dst, size_t n, void* stream);
int sig_transpose_cast_bf16(const float* src, uint16_t* dst, int rows, int cols, void* stream);
int sig_colsum_bf16(const uint16_t* a, int lda, int M, int N, float* out, void* stream);
int sig_colsum_f32(const float* a, int lda, int M, int N, float* out, void* stream);

/* Patch embedding front end (modeling/clip/model.py:448-459; modeling/meta_arch.py:101-103).
 * sig_im2col: img f32 [nimg,3,H,W] -> bf16 [nimg*(H/P)*(W/P), 3*P*P] (column = c*P*P + dy*P + dx), the A
 * operand of the conv1-as-GEMM.  sig_embed_assemble: prepend class_embedding, add sie_coe*cv_embed[cam[b]] to
 * the CLS row, add positional_embedding, apply ln_pre.  Sequences are ordered s = modality*B + b. */
int sig_im2col(const float* img, uint16_t* out, int nimg, int H, int W, int P, void* stream);
int sig_embed_assemble(const float* tok, const float* class_embedding, const float* positional_embedding,
                       const float* cv_embed, const int64_t* cam_label, float sie_coe, const float* ln_w,
                       const float* ln_b, float* x, float* pre_ln, float* mean, float* rstd, int S, int B, int L,
                       int D, float eps, void* stream);
int sig_embed_bwd(const float* d_pre_ln, float* dtok_f32, uint16_t* dtok_bf16, float* d_class_embedding,
                  float* d_positional_embedding, float* d_cv_embed, const int64_t* cam_label, float sie_coe,
                  int S, int B, int L, int D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGNAL_HIP_H */
