// ReID head of the train step: BNNeck in train mode + bias-free classifier
// (modeling/make_model.py:77-81,194-195,213-219), label-smoothed cross entropy (layers/softmax_loss.py:23-34) and the
// batch-hard soft-margin / margin triplet loss on the pre-BN features (layers/triplet_loss.py:16-135), forward and
// backward.  Shapes are tiny (B <= 128 rows, F = 512/1536 features, C ~ 171 classes): fp32 throughout, one thread or
// one wavefront per output, no tensor cores.
#include "sig_common.h"
#include "sig_kernels.h"

// ---- BatchNorm1d, training mode: batch statistics (biased var for the normalisation, unbiased for running_var) ----
// Block = 64 columns x 4 row lanes (rows ty, ty + 4, ...): a column's B <= 128 values stay in registers over the three
// passes and the row lanes are combined through LDS in a fixed order.  (One thread per column walking the rows three times
// was 6 blocks of serialized loads: 32 us for 400 KB.)
#define BN_MAXR 32     // rows per row lane: B <= 128
__device__ __forceinline__ float bn_sum4(float v, float (*red)[64], int tx, int ty) {
    __syncthreads();
    red[ty][tx] = v;
    __syncthreads();
    return (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ run_mean, float* __restrict__ run_var, float momentum, float eps,
                                                     int B, int F, float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const bool ok = c < F;
    float v[BN_MAXR];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < BN_MAXR; ++i) {
        const int r = ty + 4 * i;
        v[i] = (ok && r < B) ? x[(size_t)r * F + c] : 0.f;
        s += v[i];
    }
    const float mu = bn_sum4(s, red, tx, ty) / B;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < BN_MAXR; ++i)
        if (ty + 4 * i < B) { const float d = v[i] - mu; q += d * d; }
    q = bn_sum4(q, red, tx, ty);
    if (!ok) return;
    const float var = q / B, rs = rsqrtf(var + eps);
    if (ty == 0) {
        mean[c] = mu;
        rstd[c] = rs;
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * (B > 1 ? q / (B - 1) : var);
        }
    }
    const float g = w[c], bb = b[c];
#pragma unroll
    for (int i = 0; i < BN_MAXR; ++i) {
        const int r = ty + 4 * i;
        if (r < B) y[(size_t)r * F + c] = (v[i] - mu) * rs * g + bb;
    }
}
// dx += g*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)) ; dw += sum dy*xhat ; db += sum dy (db may be NULL: frozen bias)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, int B, int F,
                                                     float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const bool ok = c < F;
    const float mu = ok ? mean[c] : 0.f, rs = ok ? rstd[c] : 0.f;
    float xh[BN_MAXR], d[BN_MAXR];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < BN_MAXR; ++i) {
        const int r = ty + 4 * i;
        const bool live = ok && r < B;
        d[i] = live ? dy[(size_t)r * F + c] : 0.f;
        xh[i] = live ? (x[(size_t)r * F + c] - mu) * rs : 0.f;
        s1 += d[i];
        s2 += d[i] * xh[i];
    }
    s1 = bn_sum4(s1, red, tx, ty);
    s2 = bn_sum4(s2, red, tx, ty);
    if (!ok) return;
    if (ty == 0) {                   // the only writer of this column: a plain accumulate, no atomics
        if (dw) dw[c] += s2;
        if (db) db[c] += s1;
    }
    const float g = w[c], m1 = s1 / B, m2 = s2 / B;
#pragma unroll
    for (int i = 0; i < BN_MAXR; ++i) {
        const int r = ty + 4 * i;
        if (r < B) dx[(size_t)r * F + c] += g * rs * (d[i] - m1 - xh[i] * m2);   // ACCUMULATES (triplet grad is already there)
    }
}

// ---- tiny f32 contractions ------------------------------------------------------------------------------------
// out[i][j] = sum_k A[i][k] * Bm[j][k]      (wave per output, K % 4 == 0)
__global__ __launch_bounds__(256) void dot_nt_kernel(const float* __restrict__ A, const float* __restrict__ Bm, int I, int J, int K,
                                                     float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= I * J) return;
    const int i = o / J, j = o - i * J;
    const float* a = A + (size_t)i * K;
    const float* b = Bm + (size_t)j * K;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 u = *(const float4*)(a + k), v = *(const float4*)(b + k);
        s += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    }
    s = wave_sum(s);
    if (lane == 0) out[o] = s;
}
// out[i][j] (+)= sum_m P[m][i] * Q[m][j]   (thread per output, M small)      e.g. dW[C,F] = dlogits^T y
__global__ __launch_bounds__(256) void dot_tn_kernel(const float* __restrict__ P, const float* __restrict__ Q, int M, int I, int J,
                                                     float* __restrict__ out, int accumulate) {
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= (size_t)I * J) return;
    const int i = (int)(o / J), j = (int)(o - (size_t)i * J);
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += P[(size_t)m * I + i] * Q[(size_t)m * J + j];
    out[o] = accumulate ? out[o] + s : s;
}
// out[i][j] = sum_k A[i][k] * Bm[k][j]      (thread per output)                e.g. dy[B,F] = dlogits W
__global__ __launch_bounds__(256) void dot_nn_kernel(const float* __restrict__ A, const float* __restrict__ Bm, int I, int K, int J,
                                                     float* __restrict__ out) {
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= (size_t)I * J) return;
    const int i = (int)(o / J), j = (int)(o - (size_t)i * J);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += A[(size_t)i * K + k] * Bm[(size_t)k * J + j];
    out[o] = s;
}

// ---- label-smoothed CE: loss = sum_b sum_k -t_bk logp_bk / B, t = (1-eps) onehot + eps/C ; dlogits = scale*(p - t)/B ----
__global__ __launch_bounds__(256) void ce_ls_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int B, int C, float eps,
                                                    const float* __restrict__ upstream, float weight, float* __restrict__ row_loss,
                                                    float* __restrict__ dlogits) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* z = logits + (size_t)b * C;
    float mx = -INFINITY;
    for (int k = tid; k < C; k += 256) mx = fmaxf(mx, z[k]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int k = tid; k < C; k += 256) s += expf(z[k] - mx);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float lse = mx + logf(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    const int t = (int)target[b];
    float part = 0.f;
    const float up = upstream ? upstream[0] : 1.f;
    for (int k = tid; k < C; k += 256) {
        const float lp = z[k] - lse;
        const float tk = (k == t ? 1.f - eps : 0.f) + eps / C;
        part += -tk * lp;
        if (dlogits) dlogits[(size_t)b * C + k] = weight * up * (expf(lp) - tk) / B;
    }
    part = wave_sum(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    // per-row term; triplet_mine_kernel adds the rows up in a fixed order (atomics across the B blocks reordered the sum)
    if (tid == 0) row_loss[2 * b] = weight * (red[0] + red[1] + red[2] + red[3]) / B;   // (coef[2a]: the slot triplet_mine rewrites last)
}

// ---- batch-hard triplet: gram[B,B] = x x^T given; one workgroup ------------------------------------------------
// dist = sqrt(clamp(|a|^2+|b|^2-2ab, 1e-12)); d_ap = max over same label (self included), d_an = min over other labels;
// loss = mean softplus(d_ap - d_an)  (margin < 0)   or   mean relu(d_ap - d_an + margin).
// Writes per anchor: pidx, nidx and the coefficients ga = dL/d d_ap, gn = dL/d d_an (already times weight*upstream/B).
__global__ __launch_bounds__(128) void triplet_mine_kernel(const float* __restrict__ gram, const int64_t* __restrict__ labels, int B, float margin,
                                                           const float* __restrict__ upstream, float weight, float* __restrict__ loss,
                                                           int* __restrict__ pidx, int* __restrict__ nidx, float* __restrict__ coef) {
    __shared__ float red[2];
    const int a = threadIdx.x;
    float li = 0.f;
    const float ce_row = a < B ? coef[2 * a] : 0.f;     // ce_ls_kernel's per-row ID-loss term (read before coef is rewritten)
    if (a < B) {
        const float naa = gram[(size_t)a * B + a];
        const int64_t la = labels[a];
        float dap = -INFINITY, dan = INFINITY;
        int ip = a, in_ = -1;
        for (int j = 0; j < B; ++j) {
            const float d = sqrtf(fmaxf(naa + gram[(size_t)j * B + j] - 2.f * gram[(size_t)a * B + j], 1e-12f));
            if (labels[j] == la) { if (d > dap) { dap = d; ip = j; } }
            else if (d < dan) { dan = d; in_ = j; }
        }
        const float up = upstream ? upstream[0] : 1.f;
        float g;   // dL_a / d(d_ap - d_an)
        if (in_ < 0) { li = 0.f; g = 0.f; in_ = a; dan = dap; }       // no negative in the batch: contributes nothing
        else if (margin < 0.f) {
            const float t = dap - dan;
            li = t > 20.f ? t : log1pf(expf(t));
            g = 1.f / (1.f + expf(-t));
        } else {
            const float t = dap - dan + margin;
            li = t > 0.f ? t : 0.f;
            g = t > 0.f ? 1.f : 0.f;
        }
        pidx[a] = ip; nidx[a] = in_;
        coef[2 * a] = weight * up * g / B / fmaxf(dap, 1e-6f);        // divided by the distance: d dist/dx = (x_a - x_j)/dist
        coef[2 * a + 1] = -weight * up * g / B / fmaxf(dan, 1e-6f);
        if (dap * dap <= 1e-12f) coef[2 * a] = 0.f;                   // clamp active (self is the hardest positive): zero gradient
    }
    li = wave_sum(weight * li / B + ce_row);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = li;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = red[0] + red[1];       // ID + triplet: one store, the same summation tree every run
}
// anchor a contributes  dx[a] += ca (x_a - x_p) + cn (x_a - x_n) ; dx[p] -= ca (x_a - x_p) ; dx[n] -= cn (x_a - x_n).
// Owner computes: block r gathers every anchor's contribution to ROW r in anchor order -- several anchors share a hardest
// positive / negative, and scattering with atomics made this gradient (which enters the CLS token gradient and with it the
// whole backbone backward) depend on the order the blocks happened to run in.
__global__ __launch_bounds__(256) void triplet_bwd_kernel(const float* __restrict__ x, const int* __restrict__ pidx, const int* __restrict__ nidx,
                                                          const float* __restrict__ coef, int B, int F, float* __restrict__ dx) {
    __shared__ int sp[128], sn[128];
    __shared__ float sca[128], scn[128];
    const int r = blockIdx.x;
    for (int a = threadIdx.x; a < B; a += 256) { sp[a] = pidx[a]; sn[a] = nidx[a]; sca[a] = coef[2 * a]; scn[a] = coef[2 * a + 1]; }
    __syncthreads();
    constexpr int CPT = 8;           // columns per thread (F <= 2048); the loads of one anchor's rows are all in flight together
    float acc[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) acc[i] = 0.f;
    for (int a = 0; a < B; ++a) {    // anchor order = summation order (uniform branch: r is per block)
        const int p = sp[a], n = sn[a];
        if (a != r && p != r && n != r) continue;
        const float ca = sca[a], cn = scn[a];
        float xa[CPT], xp[CPT], xn[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = threadIdx.x + 256 * i;
            const bool ok = c < F;
            xa[i] = ok ? x[(size_t)a * F + c] : 0.f;
            xp[i] = ok ? x[(size_t)p * F + c] : 0.f;
            xn[i] = ok ? x[(size_t)n * F + c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const float gp = ca * (xa[i] - xp[i]), gn = cn * (xa[i] - xn[i]);
            if (a == r) acc[i] += gp + gn;
            if (p == r) acc[i] -= gp;
            if (n == r) acc[i] -= gn;
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < F) dx[(size_t)r * F + c] += acc[i];
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
int sig_launch_bnneck_fwd(const float* x, const float* bn_w, const float* bn_b, float* run_mean, float* run_var, float momentum,
                          const float* cls_w, int B, int F, int C, float* y, float* mean, float* rstd, float* logits, hipStream_t st) {
    SIG_CHECK_ARG(x && bn_w && bn_b && cls_w && y && mean && rstd && logits, "bnneck_fwd: null pointer");
    SIG_CHECK_ARG(B > 0 && B <= 128 && F > 0 && (F & 3) == 0 && C > 0, "bnneck_fwd: bad shape B=%d (<= 128) F=%d C=%d", B, F, C);
    hipLaunchKernelGGL(bn_fwd_kernel, dim3(sig_ceil_div(F, 64)), dim3(256), 0, st, x, bn_w, bn_b, run_mean, run_var, momentum, 1e-5f, B, F, y, mean, rstd);
    SIG_CHECK_LAUNCH("bn_fwd");
    hipLaunchKernelGGL(dot_nt_kernel, dim3(sig_ceil_div(B * C, 4)), dim3(256), 0, st, y, cls_w, B, C, F, logits);
    SIG_CHECK_LAUNCH("classifier");
    return 0;
}
int sig_launch_bnneck_bwd(const float* x, const float* y, const float* bn_w, const float* mean, const float* rstd, const float* cls_w,
                          const float* dlogits, int B, int F, int C, float* dy_scratch, float* dx, float* dbn_w, float* dbn_b, float* dcls_w,
                          hipStream_t st) {
    SIG_CHECK_ARG(x && y && bn_w && mean && rstd && cls_w && dlogits && dy_scratch && dx && dcls_w, "bnneck_bwd: null pointer");
    SIG_CHECK_ARG(B > 0 && B <= 128, "bnneck_bwd: batch %d must be in 1..128", B);
    hipLaunchKernelGGL(dot_tn_kernel, dim3(sig_ceil_div(C * F, 256)), dim3(256), 0, st, dlogits, y, B, C, F, dcls_w, 1);
    SIG_CHECK_LAUNCH("classifier_wgrad");
    hipLaunchKernelGGL(dot_nn_kernel, dim3(sig_ceil_div(B * F, 256)), dim3(256), 0, st, dlogits, cls_w, B, C, F, dy_scratch);
    SIG_CHECK_LAUNCH("classifier_dgrad");
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(sig_ceil_div(F, 64)), dim3(256), 0, st, x, dy_scratch, bn_w, mean, rstd, B, F, dx, dbn_w, dbn_b);
    SIG_CHECK_LAUNCH("bn_bwd");
    return 0;
}
int sig_launch_reid_loss(const float* logits, const float* feat, const int64_t* target, int B, int F, int C, float eps, float w_id,
                         float w_tri, float margin, const float* upstream, float* loss, float* dlogits, float* gram, int* pidx, int* nidx,
                         float* coef, float* dfeat, hipStream_t st) {
    SIG_CHECK_ARG(logits && feat && target && loss && gram && pidx && nidx && coef, "reid_loss: null pointer");
    SIG_CHECK_ARG(B > 1 && B <= 128 && (F & 3) == 0 && F <= 2048, "reid_loss: batch %d must be in 2..128, features %d <= 2048", B, F);
    // ce_ls parks its per-row loss terms in coef[2a]; triplet_mine (always launched) reads them, then rewrites coef
    hipLaunchKernelGGL(ce_ls_kernel, dim3(B), dim3(256), 0, st, logits, target, B, C, eps, upstream, w_id, coef, dlogits);
    SIG_CHECK_LAUNCH("ce_ls");
    hipLaunchKernelGGL(dot_nt_kernel, dim3(sig_ceil_div(B * B, 4)), dim3(256), 0, st, feat, feat, B, B, F, gram);
    SIG_CHECK_LAUNCH("gram");
    hipLaunchKernelGGL(triplet_mine_kernel, dim3(1), dim3(128), 0, st, gram, target, B, margin, upstream, w_tri, loss, pidx, nidx, coef);
    SIG_CHECK_LAUNCH("triplet_mine");
    if (dfeat) {
        hipLaunchKernelGGL(triplet_bwd_kernel, dim3(B), dim3(256), 0, st, feat, pidx, nidx, coef, B, F, dfeat);
        SIG_CHECK_LAUNCH("triplet_bwd");
    }
    return 0;
}
