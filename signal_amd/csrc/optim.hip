// Multi-tensor Adam over the flat parameter buffer (solver/make_optimizer.py builds ONE param group PER
// parameter, i.e. ~200 groups of torch.optim.Adam: hundreds of tiny launches per step).  One launch here:
// every parameter is a 64-element-aligned segment of the flat f32 buffers with its own lr / weight decay
// (torch.optim.Adam semantics: L2 decay added to the gradient, bias-corrected moments, eps outside the sqrt),
// and the bf16 GEMM operand mirror is refreshed in the same pass.
#include "sig_common.h"
#include "sig_kernels.h"

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ p_bf16,
                                                   const int* __restrict__ seg_end, const float* __restrict__ seg_lr,
                                                   const float* __restrict__ seg_wd, int nseg, float b1, float b2, float eps,
                                                   float bc1, float bc2s, float gscale, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        int lo = 0, hi = nseg - 1;            // first segment whose end is > i
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((size_t)seg_end[mid] > i) hi = mid; else lo = mid + 1;
        }
        const float lr = seg_lr[lo], wd = seg_wd[lo];
        float4 pv = *(float4*)(p + i), gv = *(const float4*)(g + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
        float* pp = (float*)&pv; float* gp = (float*)&gv; float* mp = (float*)&mv; float* vp = (float*)&vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gp[e] * gscale + wd * pp[e];
            mp[e] = b1 * mp[e] + (1.0f - b1) * gr;
            vp[e] = b2 * vp[e] + (1.0f - b2) * gr * gr;
            pp[e] -= (lr / bc1) * (mp[e] / (sqrtf(vp[e]) / bc2s + eps));   // torch.optim.Adam's exact form
        }
        *(float4*)(p + i) = pv; *(float4*)(m + i) = mv; *(float4*)(v + i) = vv;
        if (p_bf16) *(uint2*)(p_bf16 + i) = make_uint2(pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3]));
    }
}

int sig_launch_adam(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, const int* seg_end, const float* seg_lr,
                    const float* seg_wd, int nseg, float b1, float b2, float eps, int step, float gscale, size_t n, hipStream_t st) {
    SIG_CHECK_ARG(p && g && m && v && seg_end && seg_lr && seg_wd && nseg > 0 && step > 0, "adam: bad arguments");
    SIG_CHECK_ARG((n & 3) == 0, "adam: flat length must be a multiple of 4");
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = sqrtf(1.0f - powf(b2, (float)step));
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, p_bf16, seg_end, seg_lr, seg_wd, nseg, b1, b2,
                       eps, bc1, bc2, gscale, n);
    SIG_CHECK_LAUNCH("adam");
    return 0;
}
