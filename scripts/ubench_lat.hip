// Dev micro-benchmark: what ONE wave (alone on its SIMD) pays per instruction on gfx950 -- dependent and independent fp64
// VALU operations, v_rsq_f64, v_mfma_f64_16x16x4 alone and with fp64 VALU work beside it, v_readlane round trips.
// The numbers behind the cost model of factor16_mfma (csrc/chol_kernels.hip).
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_lat.hip -o scripts/ubench_lat && scripts/ubench_lat
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 512
template <int T>
__global__ void k(double* out, long long* cyc, double a, double b)
{
    double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    d4 c0 = {x0, x1, x2, x3}, c1 = c0, c2 = c0, c3 = c0;
    int s0 = 0;
    long long t0 = clock64();
    for (int i = 0; i < REP; ++i) {
        if (T == 0) {   // 8 dependent fma
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
        }
        if (T == 1) {   // 8 independent fma
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        }
        if (T == 2) {   // 8 dependent rsq
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_rsq_f64 %0, %0" : "+v"(x0));
        }
        if (T == 3) {   // 8 independent rsq
            asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
        if (T == 4) {   // 8 dependent mfma (same accumulator)
#pragma unroll
            for (int u = 0; u < 8; ++u) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, c0, 0, 0, 0);
        }
        if (T == 5) {   // 8 mfma over 4 accumulators
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, c3, 0, 0, 0);
            }
        }
        if (T == 6) {   // per mfma: 8 dependent fma that do not touch it (does VALU fp64 run beside the matrix pipe?)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, c0, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 8; ++v) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
            }
        }
        if (T == 7) {   // per mfma: 8 dependent 32-bit VALU ops beside it
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, c0, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 8; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s0) : "v"(s0));
            }
        }
        if (T == 8) {   // 8 x (mfma, then a VALU op that reads its result, feeding the next mfma's operand): the pivot loop's shape
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, c0, 0, 0, 0);
                x1 = c0[0] * a;
            }
        }
        if (T == 9) {   // 8 x (VALU fma -> readlane -> VALU use): the scalar round trip
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                int lo = __builtin_amdgcn_readlane(__double2loint(x0), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x0), 5);
                x0 = __hiloint2double(hi, lo);
            }
        }
        if (T == 10) {  // 8 dependent 32-bit VALU
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s0) : "v"(s0));
        }
        if (T == 11) {  // 8 dependent v_cndmask pairs (a 64-bit select)
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s0) : "v"(s0) : "vcc");
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0[0] + c1[1] + c2[2] + c3[3] + s0;
}
template <int T> void run(const char* name, double* d, long long* c, int per)
{
    hipLaunchKernelGGL(k<T>, dim3(1), dim3(64), 0, 0, d, c, 1.0000001, 1e-9);
    hipLaunchKernelGGL(k<T>, dim3(1), dim3(64), 0, 0, d, c, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%-78s %7.1f cycles per group of 8 = %6.1f each\n", name, (double)cy / REP, (double)cy / REP / per);
}
int main()
{
    double* d; long long* c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
    run<0>("v_fma_f64, dependent", d, c, 8);
    run<1>("v_fma_f64, independent", d, c, 8);
    run<2>("v_rsq_f64, dependent", d, c, 8);
    run<3>("v_rsq_f64, independent", d, c, 8);
    run<4>("v_mfma_f64_16x16x4, same accumulator", d, c, 8);
    run<5>("v_mfma_f64_16x16x4, four accumulators", d, c, 8);
    run<6>("mfma + 8 dependent v_fma_f64 that do not touch it (per mfma)", d, c, 8);
    run<7>("mfma + 8 dependent v_add_u32 (per mfma)", d, c, 8);
    run<8>("mfma -> v_mul_f64 of its result -> operand of the next mfma (per mfma)", d, c, 8);
    run<9>("v_fma_f64 -> 2 v_readlane_b32 -> next v_fma_f64 (per round trip)", d, c, 8);
    run<10>("v_add_u32, dependent", d, c, 8);
    run<11>("v_cndmask_b32, dependent", d, c, 8);
    return 0;
}
