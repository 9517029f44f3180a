// Micro-probe (GPU box): sustained MFMA rate of the whole chip from registers only (no LDS, no memory), for the two bf16 shapes,
// one and two waves per SIMD, short and long launches -- what clock the chip holds under nothing but matrix work, i.e. the
// ceiling the GEMM kernels' "fraction of the 2.5 PF peak" is really measured against.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512) void spin(float* out, int iters, unsigned long long* clk) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    unsigned long long t0 = __builtin_readcyclecounter();
    float r = 0.f;
    if (SHAPE == 16) {
        f32x4 c[8];
        for (int j = 0; j < 8; ++j) c[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) r += c[j][0] + c[j][3];
    } else {
        f32x16 c[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) r += c[j][0] + c[j][15];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (r == 123.456f) out[0] = r;
}

template <int SHAPE>
static void run(int threads, int iters, float* out, unsigned long long* clk, int cus) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<SHAPE>, dim3(cus), dim3(threads), 0, 0, out, 1000, clk);      // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<SHAPE>, dim3(cus), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per wave and iteration: 8 x 16x16x32 = 8 x 16384 flop, or 4 x 32x32x16 = 4 x 32768 flop
    const double flop = (double)cus * (threads / 64) * iters * 131072.0;
    unsigned long long c0; hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
    printf("mfma %dx%d  %d waves/SIMD  %8d iters: %8.3f ms  %7.1f TFLOP/s   (s_memtime ticks of block 0: %llu = %.1f MHz tick rate)\n", SHAPE, SHAPE,
           threads / 256, iters, ms, flop / ms / 1e9, c0, c0 / ms / 1e3);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out; unsigned long long* clk;
    hipMalloc(&out, 64); hipMalloc(&clk, 8 * cus);
    printf("%s, %d CUs, clockRate %d kHz\n", p.name, cus, p.clockRate);
    for (int threads : {256, 512})
        for (int iters : {20000, 200000, 2000000}) {
            run<16>(threads, iters, out, clk, cus);
            run<32>(threads, iters, out, clk, cus);
        }
    return 0;
}
