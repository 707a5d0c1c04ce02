// Throughput of the fp64 VALU instructions the Matern epilogue uses (dev tool).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_valu.hip -o scripts/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
#define CHAINS 8
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int n)
{
    double x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = a + threadIdx.x * 1e-3 + c;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
            if (OP == 3) asm volatile("v_rsq_f64 %0, %0" : "+v"(x[c]));
            if (OP == 4) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[c]) : "v"(n));
            if (OP == 5) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[c]));
            if (OP == 6) { int t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(x[c])); asm volatile("" :: "v"(t)); }
            if (OP == 7) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
            if (OP == 8) { unsigned lo = __double2loint(x[c]); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(n)); x[c] = __hiloint2double(__double2hiint(x[c]), lo); }
            if (OP == 9) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(x[c]), "v"(b) : "vcc");
            if (OP == 10) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[c]));
        }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITERS * CHAINS;          // wave-instructions
    const double per_simd = winstr / 1024.0;
    printf("%-16s %8.3f ms  %6.2f cycles/wave-instr/SIMD @2.4GHz  (%.1f T lane-ops/s)\n", name, ms,
           ms * 1e-3 * 2.4e9 / per_simd, winstr * 64 / (ms * 1e-3) / 1e12);
}
int main()
{
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("v_fma_f64", d); run<1>("v_mul_f64", d); run<2>("v_add_f64", d); run<3>("v_rsq_f64", d);
    run<4>("v_ldexp_f64", d); run<5>("v_rndne_f64", d); run<6>("v_cvt_i32_f64", d); run<7>("v_max_f64", d);
    run<8>("v_cndmask_b32", d); run<9>("v_cmp_lt_f64", d); run<10>("v_rcp_f64", d);
    return 0;
}
