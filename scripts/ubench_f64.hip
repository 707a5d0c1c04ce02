// fp64 MFMA / VALU peak microbenchmark for gfx950 (dev tool, not shipped in libspx).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_f64.hip -o /tmp/ubench_f64 && /tmp/ubench_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

template <int NACC, int MODE>  // MODE 0: mfma only, 1: valu fma only, 2: both interleaved in one wave
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed)
{
    extern __shared__ double lds_pad[];  // only to pin the number of resident blocks per CU
    if (iters < 0) lds_pad[threadIdx.x] = seed;
    d4 acc[NACC];
    double v[8];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){seed, seed, seed, seed};
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
    double av[4], bv[4];   // MODE 4: distinct A/B operand registers in the 4x4 pattern of a GEMM wave tile
    for (int i = 0; i < 4; ++i) { av[i] = a + i * 1e-7; bv[i] = b - i * 1e-7; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 3) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else if (MODE == 4) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av[i & 3]), "v"(bv[(i >> 2) & 3]));
            else if (MODE != 1) acc[i] = MFMA(a, b, acc[i]);
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fma(v[q], a, b);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int MODE>
void run(const char* name, int blocks_per_cu, double* d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu * 4;   // 4 rounds of fully resident blocks
    size_t lds = (size_t)(160 * 1024) / blocks_per_cu - 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NACC, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(256), lds, 0, d, 10, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(256), lds, 0, d, iters, 1e-3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)grid * 4;
    double mf = (MODE != 1) ? waves * iters * NACC * 2048.0 : 0;            // 16*16*4*2 per MFMA
    double vf = (MODE == 1 || MODE == 2) ? waves * iters * NACC * 8 * 64 * 2.0 : 0;      // 8 FMA x 64 lanes
    printf("%-28s blocks/CU=%d  %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF\n", name, blocks_per_cu,
           ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
}

int main()
{
    double* d; hipMalloc(&d, 256 * 8 * 4 * 256 * sizeof(double));
    for (int b = 1; b <= 4; b *= 2) {
        run<4, 0>("mfma only, 4 acc", b, d, 5000);
        run<8, 0>("mfma only, 8 acc", b, d, 2500);
        run<16, 0>("mfma only, 16 acc", b, d, 1250);
        run<8, 3>("mfma VGPR acc (asm), 8 acc", b, d, 2500);
        run<16, 3>("mfma VGPR acc (asm), 16 acc", b, d, 1250);
        run<16, 4>("mfma VGPR acc, 4x4 distinct A/B", b, d, 1250);
        run<8, 1>("valu fma only", b, d, 2500);
        run<8, 2>("mfma + 8 fma interleaved", b, d, 2500);
    }
    return 0;
}
