// Multi-tensor Adam over the flat parameter buffer (solver/make_optimizer.py builds ONE param group PER
// parameter, i.e. ~200 groups of torch.optim.Adam: hundreds of tiny launches per step).  One launch here:
// every parameter is a 64-element-aligned segment of the flat f32 buffers with its own lr / weight decay
// (torch.optim.Adam semantics: L2 decay added to the gradient, bias-corrected moments, eps outside the sqrt),
// and the 16-bit GEMM operand mirror (bf16 or f16) is refreshed in the same pass.
//
// fp16 mode adds the reference's GradScaler (engine/processor.py:119,259-261) without a host round trip: the loss
// scale lives in a 5-float device record  state = [scale, 1/scale, found_inf, growth_tracker, applied_steps];
//   grad_check_kernel        found_inf = 1 if any gradient element is inf / NaN        (GradScaler.unscale_'s check)
//   adam_kernel              reads the record: gradients are multiplied by 1/scale, the whole update is skipped when
//                            found_inf is set (GradScaler.step), bias correction uses applied_steps + 1
//   loss_scale_update_kernel scale *= backoff on overflow, *= growth after `interval` clean steps (GradScaler.update)
#include "sig_common.h"
#include "sig_kernels.h"

enum { LS_SCALE = 0, LS_INV = 1, LS_FOUND = 2, LS_TRACK = 3, LS_STEPS = 4 };

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ p16,
                                                   const int* __restrict__ seg_end, const float* __restrict__ seg_lr,
                                                   const float* __restrict__ seg_wd, int nseg, float b1, float b2, float eps,
                                                   float bc1, float bc2s, float gscale, const float* __restrict__ state, int dt,
                                                   size_t n) {
    if (state) {                                   // uniform: every thread reads the same record
        if (state[LS_FOUND] != 0.f) return;        // overflow somewhere: leave p, m, v and the operand mirror alone
        gscale *= state[LS_INV];
        const float t = state[LS_STEPS] + 1.0f;
        bc1 = 1.0f - powf(b1, t);
        bc2s = sqrtf(1.0f - powf(b2, t));
    }
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
        if (p16) *(uint2*)(p16 + i) = make_uint2(pack2_16(pp[0], pp[1], dt), pack2_16(pp[2], pp[3], dt));
    }
}

int sig_launch_adam(float* p, const float* g, float* m, float* v, bf16_t* p16, const int* seg_end, const float* seg_lr,
                    const float* seg_wd, int nseg, float b1, float b2, float eps, int step, float gscale, const float* scale_state,
                    int dt, size_t n, hipStream_t st) {
    SIG_CHECK_DT(dt, "adam");
    SIG_CHECK_ARG(p && g && m && v && seg_end && seg_lr && seg_wd && nseg > 0 && step > 0, "adam: bad arguments");
    SIG_CHECK_ARG((n & 3) == 0, "adam: flat length must be a multiple of 4");
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = sqrtf(1.0f - powf(b2, (float)step));
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, p16, seg_end, seg_lr, seg_wd, nseg, b1, b2,
                       eps, bc1, bc2, gscale, scale_state, dt, n);
    SIG_CHECK_LAUNCH("adam");
    return 0;
}

// found_inf |= any(!isfinite(g)).  HBM-bound read of the gradient buffer (364 MB, ~60 us); a non-finite f32 has all
// exponent bits set, so the test is one integer compare per element.
__global__ __launch_bounds__(256) void grad_check_kernel(const float* __restrict__ g, size_t n, float* __restrict__ state) {
    bool bad = false;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        const uint4 u = *(const uint4*)(g + i);
        bad |= ((u.x & 0x7f800000u) == 0x7f800000u) | ((u.y & 0x7f800000u) == 0x7f800000u) | ((u.z & 0x7f800000u) == 0x7f800000u) |
               ((u.w & 0x7f800000u) == 0x7f800000u);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) state[LS_FOUND] = 1.0f;   // benign race: every writer stores 1
}
int sig_launch_grad_check(const float* g, size_t n, float* state, hipStream_t st) {
    SIG_CHECK_ARG(g && state && n > 0 && (n & 3) == 0, "grad_check: bad arguments (length must be a multiple of 4)");
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(grad_check_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, n, state);
    SIG_CHECK_LAUNCH("grad_check");
    return 0;
}

// Zero a list of ranges of one buffer in ONE launch: table[k] = {offset, length} in floats, chunk_start[k] = first block of range k
// (4096 floats per block), chunk_start[n] = grid.  The training engine zeroes the gradients that ACCUMULATE during a step this way
// and leaves out the transformer blocks' weight gradients, which its backward overwrites (sig_tune_tn_overwrite).
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const long long* __restrict__ table,
                                                          const int* __restrict__ chunk_start, int n) {
    const int b = blockIdx.x;
    int k = 0;
    while (k + 1 < n && chunk_start[k + 1] <= b) ++k;      // (uniform; a few dozen ranges)
    const long long off = table[2 * k], len = table[2 * k + 1], c0 = (long long)(b - chunk_start[k]) * 4096;
#pragma unroll 4
    for (int i = threadIdx.x; i < 4096; i += 256)
        if (c0 + i < len) base[off + c0 + i] = 0.f;
}
int sig_launch_zero_ranges(float* base, const long long* table, const int* chunk_start, int n, int total_chunks, hipStream_t st) {
    SIG_CHECK_ARG(base && table && chunk_start && n > 0 && total_chunks > 0, "zero_ranges: bad arguments");
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(total_chunks), dim3(256), 0, st, base, table, chunk_start, n);
    SIG_CHECK_LAUNCH("zero_ranges");
    return 0;
}

__global__ void loss_scale_update_kernel(float* __restrict__ state, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float scale = state[LS_SCALE], track = state[LS_TRACK];
    if (state[LS_FOUND] != 0.f) {
        scale *= backoff;
        track = 0.f;
    } else {
        state[LS_STEPS] += 1.0f;             // an optimizer step was applied
        track += 1.0f;
        if (track >= (float)interval) {
            scale *= growth;
            track = 0.f;
        }
    }
    state[LS_SCALE] = scale;
    state[LS_INV] = 1.0f / scale;
    state[LS_FOUND] = 0.f;
    state[LS_TRACK] = track;
}
int sig_launch_loss_scale_update(float* state, float growth, float backoff, int interval, hipStream_t st) {
    SIG_CHECK_ARG(state && growth >= 1.0f && backoff > 0.f && backoff <= 1.0f && interval > 0, "loss_scale_update: bad arguments");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, st, state, growth, backoff, interval);
    SIG_CHECK_LAUNCH("loss_scale_update");
    return 0;
}
