// Dev micro-benchmark: cycles per pivot of the in-wave 16x16 Cholesky (+ inverse) built from rank-1
// v_mfma_f64_16x16x4 updates (csrc/chol_kernels.hip: factor16_mfma), in several variants.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_factor16.hip -o scripts/ubench_factor16 && scripts/ubench_factor16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double readlane_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// V: 0 = factor + inverse (production), 1 = factor only, 2 = like 0 without the d > 0 select before rsq,
//    3 = like 2 with mask multipliers instead of selects, 4 = factor only, rsq without refinement (chain floor probe)
template <int V>
__device__ __forceinline__ void factor16(d4& C, d4& X, d4& U, int lane, int& bad)
{
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { X[r] = (q + 4 * r == c) ? 1.0 : 0.0; U[r] = 0.0; }
    double mk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mk[k] = (q == k) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int kq = j & 3, rj = j >> 2;
        double d = readlane_f64(C[rj], j + 16 * kq);
        if (V == 0 || V == 1) {
            if (!(d > 0.0)) { if (!bad) bad = j + 1; d = 1.0; }
        }
        const double y0 = __builtin_amdgcn_rsq(d);
        double rinv;
        if (V == 4) rinv = y0;
        else {
            const double e0 = fma(-d * y0, y0, 1.0);
            rinv = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
        }
        if (V >= 2) { if (!(rinv == rinv) && !bad) bad = j + 1; }   // off the chain: NaN for d <= 0 / NaN
        const bool grp = (q == kq);
        double b, lcol;
        if (V == 3) { const double rm = rinv * mk[kq]; b = C[rj] * rm; lcol = b; }
        else { lcol = C[rj] * rinv; b = grp ? lcol : 0.0; }
        C = MFMA_F64(-b, b, C);
        if (V == 0 || V == 2 || V == 3) {
            const double xs = X[rj] * rinv;
            const double bX = (V == 3) ? xs * mk[kq] : (grp ? xs : 0.0);
            const double aX = (grp && c > j) ? -lcol : 0.0;
            X = MFMA_F64(aX, bX, X);
            X[rj] = grp ? xs : X[rj];
        }
        double sd = d * rinv;
        sd = fma(fma(-sd, sd, d), 0.5 * rinv, sd);
        const double keep = (c == j) ? sd : ((c > j) ? lcol : 0.0);
        U[rj] = grp ? keep : U[rj];
    }
}

template <int V>
__global__ void bench(const double* S, double* out, long long* cyc, int reps)
{
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    d4 C0;
    for (int r = 0; r < 4; ++r) C0[r] = S[(q + 4 * r) * 16 + c];
    d4 C, X, U;
    int bad = 0;
    double acc = 0.0;
    long long t0 = clock64();
    for (int it = 0; it < reps; ++it) {
        C = C0;
        C[0] += acc * 1e-300;          // serialise the repetitions
        factor16<V>(C, X, U, lane, bad);
        acc += U[3] + X[3];
    }
    long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    for (int r = 0; r < 4; ++r) {
        out[((q + 4 * r) * 16 + c) * 2 + 0] = U[r];
        out[((q + 4 * r) * 16 + c) * 2 + 1] = X[r];
    }
    if (lane == 0) out[512] = acc + bad;
}

template <int V>
void run(const double* dS, double* dOut, long long* dCyc, const double* hS, const char* name)
{
    const int reps = 200;
    hipLaunchKernelGGL(bench<V>, dim3(1), dim3(64), 0, 0, dS, dOut, dCyc, reps);
    hipLaunchKernelGGL(bench<V>, dim3(1), dim3(64), 0, 0, dS, dOut, dCyc, reps);
    hipDeviceSynchronize();
    long long cyc; double out[513];
    hipMemcpy(&cyc, dCyc, 8, hipMemcpyDeviceToHost);
    hipMemcpy(out, dOut, sizeof out, hipMemcpyDeviceToHost);
    // check U^T U == S
    double err = 0.0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += out[(k * 16 + i) * 2] * out[(k * 16 + j) * 2];
        err = fmax(err, fabs(s - hS[i * 16 + j]));
    }
    printf("%-44s %7.1f cycles / 16x16 block  = %5.1f / pivot   |U^T U - S| = %.2e\n", name, (double)cyc / reps,
           (double)cyc / reps / 16, err);
}

int main()
{
    double hS[256], A[16 * 40];
    unsigned s = 12345;
    for (int i = 0; i < 16 * 40; ++i) { s = s * 1664525u + 1013904223u; A[i] = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double v = (i == j) ? 0.1 : 0.0;
        for (int k = 0; k < 40; ++k) v += A[i * 40 + k] * A[j * 40 + k];
        hS[i * 16 + j] = v;
    }
    double *dS, *dOut; long long* dCyc;
    hipMalloc(&dS, sizeof hS); hipMalloc(&dOut, 513 * 8); hipMalloc(&dCyc, 8);
    hipMemcpy(dS, hS, sizeof hS, hipMemcpyHostToDevice);
    run<0>(dS, dOut, dCyc, hS, "V0 factor + inverse (production)");
    run<1>(dS, dOut, dCyc, hS, "V1 factor only");
    run<2>(dS, dOut, dCyc, hS, "V2 factor + inverse, no select before rsq");
    run<3>(dS, dOut, dCyc, hS, "V3 = V2 with mask multipliers");
    run<4>(dS, dOut, dCyc, hS, "V4 factor only, raw rsq (chain floor probe)");
    return 0;
}
