// Dev micro-benchmark: does a wave's pace depend on what its neighbours on the OTHER SIMDs of the CU do?  Wave 0 runs the
// pivot loop's instruction mix (dependent fp64 VALU, one fp64 MFMA per 8); waves 1-3 of the same workgroup do nothing /
// poll LDS / issue fp64 MFMAs / issue fp64 VALU.  (Question behind it: can factor16_mfma hand its inverse to a helper wave?)
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_neigh.hip -o scripts/ubench_neigh && scripts/ubench_neigh
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 2048
template <int M>
__global__ void k(double* out, long long* cyc, double a, double b)
{
    __shared__ volatile int flag;
    __shared__ double box[64];
    const int wave = threadIdx.x >> 6;
    double x0 = a + threadIdx.x * 1e-9;
    d4 c0 = {x0, x0, x0, x0};
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    if (wave == 0) {
        long long t0 = clock64();
        for (int i = 0; i < REP; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, c0, 0, 0, 0);
            if (M == 5) { box[threadIdx.x] = x0; }      // the chain wave also posts a value per round
        }
        long long t1 = clock64();
        if (threadIdx.x == 0) { cyc[0] = t1 - t0; flag = 1; }
    } else {
        if (M == 1 || M == 5) { while (flag == 0) { x0 += box[threadIdx.x & 63]; } }                       // LDS polling (+ a read of the box)
        if (M == 2) { while (flag == 0) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, c0, 0, 0, 0); } }   // matrix pipe busy
        if (M == 3) { while (flag == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b)); } }
        if (M == 4) { while (flag == 0) { __builtin_amdgcn_s_sleep(4); } }                                  // polling with s_sleep
    }
    out[threadIdx.x] = x0 + c0[0];
}
template <int M> void run(const char* name, double* d, long long* c)
{
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(256), 0, 0, d, c, 1.0000001, 1e-9);
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(256), 0, 0, d, c, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%-64s %7.1f cycles per round (8 dependent v_fma_f64 + 1 mfma)\n", name, (double)cy / REP);
}
int main()
{
    double* d; long long* c; hipMalloc(&d, 256 * 8); hipMalloc(&c, 8);
    run<0>("waves 1-3 idle (left the kernel)", d, c);
    run<1>("waves 1-3 poll an LDS word and read an LDS value", d, c);
    run<4>("waves 1-3 poll an LDS word with s_sleep 4", d, c);
    run<2>("waves 1-3 issue v_mfma_f64_16x16x4 back to back", d, c);
    run<3>("waves 1-3 issue dependent v_fma_f64", d, c);
    run<5>("wave 0 also stores a value to LDS per round; waves 1-3 poll", d, c);
    return 0;
}
