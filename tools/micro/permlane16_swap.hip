#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%d:%u,%u ", i, h[i], h[64 + i]); printf("\n");
}
