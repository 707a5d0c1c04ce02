// Dev micro-benchmark: phase times inside diag_block (csrc/chol_kernels.hip), one workgroup.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSPX_DIAG_STAMPS -Ispearmint_amd/csrc scripts/ubench_diag.hip -o scripts/ubench_diag
#include "../spearmint_amd/csrc/chol_kernels.hip"
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256, 2) void bench(const double* A, double* Lout, double* Dk, int* info, long long* stamps)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* S = smem; double* XT = S + NB * LDP; double* T16 = XT + NB * LDP;
    long long t0 = clock64();
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) S[(idx >> 6) * LDP + (idx & 63)] = A[idx];
    __syncthreads();
    diag_block(S, XT, T16, info, 0, Lout, 64, Dk);
    __syncthreads();
    if (threadIdx.x == 0) { stamps[0] = t0; for (int i = 0; i < 19; ++i) stamps[1 + i] = g_stamp[i]; stamps[20] = clock64(); }
}
int main()
{
    std::vector<double> A(64 * 64), G(64 * 100);
    unsigned s = 777;
    for (auto& v : G) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) { double v = (i == j) ? 0.05 : 0; for (int k = 0; k < 100; ++k) v += G[i * 100 + k] * G[j * 100 + k]; A[i * 64 + j] = v; }
    double *dA, *dL, *dD; int* dI; long long* dS;
    hipMalloc(&dA, 32768); hipMalloc(&dL, 32768); hipMalloc(&dD, 32768); hipMalloc(&dI, 4); hipMalloc(&dS, 21 * 8);
    hipMemcpy(dA, A.data(), 32768, hipMemcpyHostToDevice); hipMemset(dI, 0, 4);
    size_t lds = (2 * NB * LDP + DIAG_T16_DOUBLES) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void*>(bench), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(bench, dim3(1), dim3(256), lds, 0, dA, dL, dD, dI, dS);
    hipDeviceSynchronize();
    long long st[21]; hipMemcpy(st, dS, sizeof st, hipMemcpyDeviceToHost);
    std::vector<double> L(4096), D(4096); hipMemcpy(L.data(), dL, 32768, hipMemcpyDeviceToHost); hipMemcpy(D.data(), dD, 32768, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) {
        double s1 = 0, s2 = 0;
        for (int k = 0; k < 64; ++k) { s1 += L[i * 64 + k] * L[j * 64 + k]; s2 += D[i * 64 + k] * L[k * 64 + j]; }
        e1 = fmax(e1, fabs(s1 - A[i * 64 + j])); e2 = fmax(e2, fabs(s2 - (i == j)));
    }
    printf("|LL^T - A| %.2e  |Dinv L - I| %.2e\n", e1, e2);
    printf("fill S %lld\n", st[1] - st[0]);
    for (int b = 0; b < 4; ++b)
        printf("round %d: factor16 (+inverse rows) %lld  barrier %lld  (b) sub-panel %lld  (c) trailing %lld\n", b,
               st[2 + 4 * b] - (b ? st[1 + 4 * b] : st[1]), st[3 + 4 * b] - st[2 + 4 * b], st[4 + 4 * b] - st[3 + 4 * b], st[5 + 4 * b] - st[4 + 4 * b]);
    printf("inverse last row %lld  write-out %lld  total %lld cycles\n", st[18] - st[17], st[19] - st[18], st[20] - st[0]);
    return 0;
}
