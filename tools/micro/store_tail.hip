// Micro-probe (GPU box): how fast can ONE CU drain a GEMM epilogue's stores, by the shape of a store instruction?
// Every wave of an 8-wave block issues 16 dwordx4 stores (a 256x256 16-bit tile = 128 KB per block), for several
// lane -> address maps of the same 1 KB per instruction, alone on the chip and with all 256 CUs busy:
//   A  16 rows x 64 B   (the register epilogue of gemm_bf16.hip: 4 lanes per row, row stride = ldo)
//   B   8 rows x 128 B  (whole lines)
//   C   2 rows x 512 B
//   D   1 row  x 1 KB   (contiguous)
//   E  64 rows x 16 B   (row per lane)
//   F   4 rows x 128 B  with dwordx2 (8 B per lane, lane-consecutive; 32 stores per wave for the same 128 KB)
//   G   8 rows x 128 B  like B but the 8 lanes of a line are NOT consecutive (lanes fr, fr+8 of every 16: the DPP exchange)
// Prints cycles per tile (issue only / issue + drain) and B/clk/CU.  Build: hipcc --offload-arch=gfx950 -O2 store_tail.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
constexpr int LDO = 2304;                     // elements (16-bit) per output row
template <int PAT>
__global__ __launch_bounds__(512) void probe(unsigned short* out, unsigned long long* stamps, int tiles_n, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long t0, t1, t2, acc_issue = 0, acc_all = 0;
    uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    for (int r = 0; r < reps; ++r) {
        // tile of this block and repetition: 256 rows x 256 columns somewhere in the [M, LDO] matrix
        const int t = blockIdx.x + r * gridDim.x;
        const int tm = t / tiles_n, tc = t - tm * tiles_n;
        char* base = (char*)(out + ((size_t)tm * 256 + (wave >> 2) * 128) * LDO + tc * 256 + (wave & 3) * 64);   // wave tile 128 x 64
        __syncthreads();
        STAMP(t0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {         // 16 KB per wave = 128 rows x 128 B
            size_t off;
            if (PAT == 0) { const int i = s >> 1, jp = s & 1; off = ((size_t)(i * 16 + (lane & 15)) * LDO) * 2 + jp * 64 + (lane >> 4) * 16; }
            if (PAT == 1) { off = ((size_t)(s * 8 + (lane >> 3)) * LDO) * 2 + (lane & 7) * 16; }
            if (PAT == 2) { off = ((size_t)((wave & 3) * 32 + s * 2 + (lane >> 5)) * LDO) * 2 + (lane & 31) * 16 - (size_t)(wave & 3) * 128; }   // rows of its own per wave, the tile's 512-B rows
            if (PAT == 3) { off = ((size_t)((wave & 3) * 16 + s) * LDO) * 2 + lane * 16 - (size_t)(wave & 3) * 128; }
            if (PAT == 4) { const int i = s >> 3, c = s & 7; off = ((size_t)(i * 64 + lane) * LDO) * 2 + c * 16; }
            if (PAT == 6) { off = ((size_t)(s * 8 + (lane & 7)) * LDO) * 2 + ((lane >> 3) & 1) * 64 + (lane >> 4) * 16; }
            if (PAT != 5) *(uint4*)(base + off) = v;
        }
        if (PAT == 5) {
#pragma unroll
            for (int s = 0; s < 32; ++s)
                *(uint2*)(base + ((size_t)(s * 4 + (lane >> 4)) * LDO) * 2 + (lane & 15) * 8) = make_uint2(v.x, v.y);
        }
        STAMP(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        STAMP(t2);
        if (r) { acc_issue += t1 - t0; acc_all += t2 - t0; }
    }
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = acc_issue / (reps - 1); stamps[blockIdx.x * 2 + 1] = acc_all / (reps - 1); }
}
template <int PAT>
static void run(const char* name, unsigned short* out, unsigned long long* st, int nb, int reps) {
    hipLaunchKernelGGL(probe<PAT>, dim3(nb), dim3(512), 0, 0, out, st, LDO / 256, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 2);
    hipMemcpy(h.data(), st, nb * 16, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int i = 0; i < nb; ++i) { a.push_back((double)h[2 * i]); b.push_back((double)h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-22s blocks %3d: issue %7.0f cycles, issue + drain %7.0f cycles per 128-KB tile (median) = %5.1f B/clk/CU\n", name, nb,
           a[nb / 2], b[nb / 2], 131072.0 / b[nb / 2]);
}
int main() {
    unsigned short* out; unsigned long long* st;
    const size_t rows = 256 * 130;                        // 9 column tiles x 130 row tiles = 1170 tiles >= 256 * reps
    hipMalloc(&out, rows * LDO * 2 + (1 << 20)); hipMemset(out, 0, rows * LDO * 2 + (1 << 20));
    out += 512;                                           // room for the negative column offsets of C / D
    hipMalloc(&st, 256 * 16);
    for (int nb : {8, 256}) {
        const int reps = nb == 8 ? 9 : 4;
        run<0>("A 16 rows x 64 B", out, st, nb, reps);
        run<1>("B 8 rows x 128 B", out, st, nb, reps);
        run<2>("C 2 rows x 512 B", out, st, nb, reps);
        run<3>("D 1 KB contiguous", out, st, nb, reps);
        run<4>("E 64 rows x 16 B", out, st, nb, reps);
        run<5>("F 4 rows x 128 B x2", out, st, nb, reps);
        run<6>("G 8 rows x 128 B split", out, st, nb, reps);
    }
    return 0;
}
