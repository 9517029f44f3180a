// Micro-probe (GPU box): does an in-flight LDS-DMA (global_load_lds) hold up `s_waitcnt lgkmcnt(0)` on gfx950?
// Prints, per probe, cycles from issue to the wait returning.  Build: hipcc --offload-arch=gfx950 -O2 lgkm_dma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
__global__ void probe(const float* src, unsigned long long* out, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t0, t1, t2, t3, t4;
    unsigned long long a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    float x;
    for (int it = 0; it < 17; ++it) {
        const float* g = src + (size_t)blockIdx.x * stride + (size_t)it * 8192 + threadIdx.x * 4;
        STAMP(t0);
        __builtin_amdgcn_global_load_lds((const void*)g, (__attribute__((address_space(3))) void*)smem, 16, 0, 0);
        asm volatile("ds_read_b32 %0, %1 offset:8192\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"((int)threadIdx.x * 4) : "memory");
        STAMP(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(t2);
        asm volatile("ds_read_b32 %0, %1 offset:8192\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"((int)threadIdx.x * 4) : "memory");
        STAMP(t3);
        __builtin_amdgcn_global_load_lds((const void*)(g + 4096), (__attribute__((address_space(3))) void*)smem, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(t4);
        if (it) { a1 += t1 - t0; a2 += t2 - t1; a3 += t3 - t2; a4 += t4 - t3; }
    }
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = a1 / 16; out[blockIdx.x * 4 + 1] = a2 / 16;
        out[blockIdx.x * 4 + 2] = a3 / 16; out[blockIdx.x * 4 + 3] = a4 / 16;
    }
    if (x == 12345.f) out[0] = 0;
}
int main() {
    const int nb = 64; const size_t stride = 1 << 20;
    float *src, *flush; unsigned long long* out;
    hipMalloc(&src, nb * stride * sizeof(float)); hipMemset(src, 0, nb * stride * sizeof(float));
    hipMalloc(&flush, (size_t)1 << 30); hipMemset(flush, 1, (size_t)1 << 30);   // push src out of L2 / infinity cache
    hipMalloc(&out, nb * 4 * 8);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 16384, 0, src, out, stride);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 4);
    hipMemcpy(h.data(), out, nb * 4 * 8, hipMemcpyDeviceToHost);
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < nb; ++b) for (int i = 0; i < 4; ++i) s[i] += h[b * 4 + i];
    printf("cold DMA + ds_read: lgkmcnt(0) returns after %.0f cycles; vmcnt(0) only after another %.0f\n", s[0] / nb, s[1] / nb);
    printf("ds_read alone + lgkmcnt(0): %.0f cycles; cold DMA + vmcnt(0): %.0f cycles\n", s[2] / nb, s[3] / nb);
    return 0;
}
