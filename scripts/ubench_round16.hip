// Dev micro-benchmark: one wavefront factoring a 16-column tile column with rank-1 v_mfma_f64_16x16x4 updates --
// the diagonal tile D, NT tiles below it carried in the same pivot chain, and the inverse X of the diagonal tile
// -- in several code shapes, cycles per pivot (csrc/chol_kernels.hip: factor64_round / factor16_mfma).
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_round16.hip -o scripts/ubench_round16 && scripts/ubench_round16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ double readlane_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// MODE 0: everything of pivot j inside iteration j, compiler-scheduled (X row scaled in place: the round-2 form when NT = 0)
// MODE 1: like 0, but finished rows go to separate registers (Xo): no VALU write to an MFMA accumulator
// MODE 2: like 1 + sched_barrier between pivots
// MODE 3: software-pipelined by one pivot (other tiles' MFMAs of pivot j-1 inside pivot j's chain), one sched_barrier per pivot
// MODE 4: like 3 with the MFMAs pinned one per chain link (the form GPU call 1 measured)
// MODE 5: like 1, no inverse at all (X dropped): D + NT tiles
template <int NT, int MODE>
__device__ __forceinline__ void round16(d4& D, d4 (&T)[3], d4& Xout, d4& U, d4 (&P)[3], int lane, int& bad)
{
    const int c = lane & 15, q = lane >> 4;
    d4 X, Xo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        X[r] = (q + 4 * r == c) ? 1.0 : 0.0; Xo[r] = 0.0; U[r] = 0.0;
#pragma unroll
        for (int t = 0; t < 3; ++t) P[t][r] = 0.0;
    }
    if (MODE <= 2 || MODE == 5) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kq = j & 3, rj = j >> 2;
            double d = readlane_f64(D[rj], j + 16 * kq);
            if (!(d > 0.0)) { if (!bad) bad = j + 1; d = 1.0; }
            const double y0 = __builtin_amdgcn_rsq(d);
            const double e0 = fma(-d * y0, y0, 1.0);
            const double rinv = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
            const bool grp = (q == kq);
            const double lcol = D[rj] * rinv;
            const double bD = grp ? lcol : 0.0;
            const double nbD = -bD;
            D = MFMA_F64(nbD, bD, D);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double lt = T[t][rj] * rinv;
                const double bT = grp ? lt : 0.0;
                T[t] = MFMA_F64(nbD, bT, T[t]);
                P[t][rj] = grp ? lt : P[t][rj];
            }
            if (MODE != 5) {
                const double xs = X[rj] * rinv;
                const double bX = grp ? xs : 0.0;
                const double aX = (grp && c > j) ? -lcol : 0.0;
                X = MFMA_F64(aX, bX, X);
                if (MODE == 0) X[rj] = grp ? xs : X[rj];
                else Xo[rj] = grp ? xs : Xo[rj];
            }
            double sd = d * rinv;
            sd = fma(fma(-sd, sd, d), 0.5 * rinv, sd);
            const double keep = (c == j) ? sd : ((c > j) ? lcol : 0.0);
            U[rj] = grp ? keep : U[rj];
            if (MODE == 2) SB();
        }
        Xout = (MODE == 0) ? X : Xo;
        return;
    }
    double nbD_p = 0.0, aX_p = 0.0, bX_p = 0.0, bT_p[3] = {0.0, 0.0, 0.0};
#define SLOT(s)                                                                       \
    if (j > 0 && (s) <= NT) {                                                          \
        if ((s) < NT) T[(s) < NT ? (s) : 0] = MFMA_F64(nbD_p, bT_p[(s) < NT ? (s) : 0], T[(s) < NT ? (s) : 0]); \
        else X = MFMA_F64(aX_p, bX_p, X);                                              \
    }
#pragma unroll
    for (int j = 0; j <= 16; ++j) {
        const int kq = j & 3, rj = (j >> 2) & 3;
        const bool grp = (q == kq);
        double d = 1.0, y0 = 1.0;
        if (j < 16) {
            d = readlane_f64(D[rj], j + 16 * kq);
            if (!(d > 0.0)) { if (!bad) bad = j + 1; d = 1.0; }
            y0 = __builtin_amdgcn_rsq(d);
        }
        if (MODE == 4) SB();
        SLOT(0)
        const double e0 = fma(-d * y0, y0, 1.0);
        if (MODE == 4) SB();
        SLOT(1)
        const double ye = y0 * e0, pe = fma(0.375, e0, 0.5);
        if (MODE == 4) SB();
        SLOT(2)
        const double rinv = fma(ye, pe, y0);
        if (MODE == 4) SB();
        SLOT(3)
        if (j < 16) {
            const double lcol = D[rj] * rinv;
            const double bD = grp ? lcol : 0.0;
            const double nbD = -bD;
            if (MODE == 4) SB();
            D = MFMA_F64(nbD, bD, D);
            if (MODE == 4) SB();
            nbD_p = nbD;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double lt = T[t][rj] * rinv;
                bT_p[t] = grp ? lt : 0.0;
                P[t][rj] = grp ? lt : P[t][rj];
            }
            const double xs = X[rj] * rinv;
            bX_p = grp ? xs : 0.0;
            aX_p = (grp && c > j) ? -lcol : 0.0;
            Xo[rj] = grp ? xs : Xo[rj];
            double sd = d * rinv;
            sd = fma(fma(-sd, sd, d), 0.5 * rinv, sd);
            const double keep = (c == j) ? sd : ((c > j) ? lcol : 0.0);
            U[rj] = grp ? keep : U[rj];
        }
        SB();
    }
#undef SLOT
    Xout = Xo;
}

template <int NT, int MODE>
__global__ void bench(const double* S, double* out, long long* cyc, int reps)
{
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    d4 D0, T0[3];
    for (int r = 0; r < 4; ++r) {
        D0[r] = S[(q + 4 * r) * 16 + c];
        for (int t = 0; t < 3; ++t) T0[t][r] = S[256 * (t + 1) + (q + 4 * r) * 16 + c];
    }
    d4 D, T[3], X, U, P[3];
    int bad = 0;
    double acc = 0.0;
    long long t0 = clock64();
    for (int it = 0; it < reps; ++it) {
        D = D0;
        for (int t = 0; t < 3; ++t) T[t] = T0[t];
        D[0] += acc * 1e-300;          // serialise the repetitions
        round16<NT, MODE>(D, T, X, U, P, lane, bad);
        acc += U[3] + X[3] + P[0][3] + P[1][3] + P[2][3];
    }
    long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    for (int r = 0; r < 4; ++r) {
        out[((q + 4 * r) * 16 + c) * 3 + 0] = U[r];
        out[((q + 4 * r) * 16 + c) * 3 + 1] = X[r];
        out[((q + 4 * r) * 16 + c) * 3 + 2] = P[0][r];
    }
    if (lane == 0) out[768] = acc + bad;
}

static double hS[1024];
template <int NT, int MODE>
void run(const double* dS, double* dOut, long long* dCyc)
{
    const int reps = 200;
    hipLaunchKernelGGL((bench<NT, MODE>), dim3(1), dim3(64), 0, 0, dS, dOut, dCyc, reps);
    hipLaunchKernelGGL((bench<NT, MODE>), dim3(1), dim3(64), 0, 0, dS, dOut, dCyc, reps);
    hipDeviceSynchronize();
    long long cyc; double out[769];
    hipMemcpy(&cyc, dCyc, 8, hipMemcpyDeviceToHost);
    hipMemcpy(out, dOut, sizeof out, hipMemcpyDeviceToHost);
    // U[n][c] = L[c][n]: check U^T U == S and X L == I, and (NT > 0) P0[n][c] = Lbelow[c][n]: Lbelow L^T == S1^T ...
    double err = 0.0, errx = 0.0, errp = 0.0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0.0, x = 0.0, p = 0.0;
        for (int k = 0; k < 16; ++k) {
            s += out[(k * 16 + i) * 3] * out[(k * 16 + j) * 3];
            x += out[(i * 16 + k) * 3 + 1] * out[(j * 16 + k) * 3];     // X[i][k] L[k][j], L[k][j] = U[j][k]
            p += out[(k * 16 + i) * 3 + 2] * out[(k * 16 + j) * 3];     // sum_k Lb[i][k] L[j][k]
        }
        err = fmax(err, fabs(s - hS[i * 16 + j]));
        errx = fmax(errx, fabs(x - (i == j)));
        errp = fmax(errp, fabs(p - hS[256 + j * 16 + i]));              // tile below held transposed: T[n'][i]
    }
    printf("NT=%d MODE=%d  %7.1f cycles / round = %5.1f / pivot   |U^T U - S| %.1e  |X L - I| %.1e  |Lb L^T - S1| %.1e\n", NT, MODE,
           (double)cyc / reps, (double)cyc / reps / 16, err, MODE == 5 ? 0.0 : errx, NT > 0 ? errp : 0.0);
}

int main()
{
    static double A[64 * 40];
    unsigned s = 12345;
    for (int i = 0; i < 64 * 40; ++i) { s = s * 1664525u + 1013904223u; A[i] = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
    // S = A A^T + 0.1 I (64 x 64); tile 0 = the diagonal tile, tiles 1..3 = (S(ti, 0))^T in accumulator layout [n'][i]
    for (int t = 0; t < 4; ++t)
        for (int n = 0; n < 16; ++n) for (int i = 0; i < 16; ++i) {
            const int row = 16 * t + i, col = n;
            double v = (row == col) ? 0.1 : 0.0;
            for (int k = 0; k < 40; ++k) v += A[row * 40 + k] * A[col * 40 + k];
            hS[256 * t + n * 16 + i] = v;      // [n'][i] = S[16 t + i][n']
        }
    double *dS, *dOut; long long* dCyc;
    hipMalloc(&dS, sizeof hS); hipMalloc(&dOut, 769 * 8); hipMalloc(&dCyc, 8);
    hipMemcpy(dS, hS, sizeof hS, hipMemcpyHostToDevice);
    run<0, 0>(dS, dOut, dCyc); run<0, 1>(dS, dOut, dCyc); run<0, 2>(dS, dOut, dCyc); run<0, 3>(dS, dOut, dCyc);
    run<0, 4>(dS, dOut, dCyc); run<0, 5>(dS, dOut, dCyc);
    run<1, 0>(dS, dOut, dCyc); run<1, 1>(dS, dOut, dCyc); run<1, 2>(dS, dOut, dCyc); run<1, 3>(dS, dOut, dCyc);
    run<1, 4>(dS, dOut, dCyc); run<1, 5>(dS, dOut, dCyc);
    run<3, 1>(dS, dOut, dCyc); run<3, 2>(dS, dOut, dCyc); run<3, 3>(dS, dOut, dCyc); run<3, 4>(dS, dOut, dCyc); run<3, 5>(dS, dOut, dCyc);
    return 0;
}
