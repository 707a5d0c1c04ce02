// EI and its gradient at a single point, for every resident hyper-parameter draw -- the
// objective of the reference's local refinement:
//   GPEIOptChooser.py:391-440  grad_optimize_ei (no-pending branch), summed over draws by
//   grad_optimize_ei_over_hypers (:360-388), minimised by L-BFGS-B (:39-43, :285-289).
//
// Per draw h, with k = amp2 * Matern52(ls; comp, x)  (N),  W = L^-1:
//   t = W k            (beta, :413)            z = W^T t = K^-1 k        (:436)
//   func_m = k.alpha + mean, func_v = amp2(1+1e-6) - |t|^2, EI as in predict_kernels.hip
//   G[j][d] = dk/dr2 (r_j) * 2 (comp_jd/ls_d - x_d/ls_d) / ls_d        (gp.py:129-132, :56-85)
//   grad[d] = 0.5 amp2 ( (alpha . G[:,d]) (-Phi) + (-2 z . G[:,d]) (0.5 phi / s) )   (:427-437,
//             including the reference's factor one half)
// Kernels: k_point_cov (k and dk/dr2), two triangular matrix-vector products with the
// resident WT (reusing k_gamma / k_alpha of chol_kernels.hip), k_point_finish (reductions, EI,
// gradient).  A point costs ~2 x 16 MB of W reads per draw: bandwidth-bound, sub-millisecond.
#include "common.h"

#define SQRT5 2.23606797749978969641

// k[h][j] = amp2 * matern(r_j), dkdr2[h][j] = -(5/6) exp(-sqrt5 r)(1 + sqrt5 r); pad rows -> 0
__global__ __launch_bounds__(256) void k_point_cov(
    const double* __restrict__ Xs /*[nh][Np][Dp]*/, const double* __restrict__ s1 /*[nh][Np]*/,
    const double* __restrict__ hyp /*[nh][3+D]*/, const double* __restrict__ htab,
    const double* __restrict__ x /*[D]*/, double* __restrict__ kvec, double* __restrict__ dkdr2,
    int N, int Np, int D, int Dp)
{
#pragma clang fp contract(off)
    const int h = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Np) return;
    const double* ls = hyp + (size_t)h * (3 + D) + 3;
    const double amp2 = htab[h * SPX_HT + 2];
    const double* xj = Xs + ((size_t)h * Np + j) * Dp;
    double kv = 0.0, dv = 0.0;
    if (j < N) {
        // same expanded form as gp.dist2: -( (xx1 . 2 xx2 - |xx1|^2) - |xx2|^2 ), clamped at 0
        double g = 0.0, s2 = 0.0;
        for (int d = 0; d < D; ++d) {
            const double xc = x[d] / ls[d];
            g = g + xj[d] * (2.0 * xc);
            s2 = s2 + xc * xc;
        }
        const double t = (g - s1[(size_t)h * Np + j]) - s2;
        const double nt = -t;
        double r2 = (nt < 0.0) ? 0.0 : nt;
        r2 = fabs(r2);
        const double r = sqrt(r2);
        const double e = exp(-SQRT5 * r);
        kv = amp2 * (((1.0 + SQRT5 * r) + (5.0 / 3.0) * r2) * e);
        dv = -(5.0 / 6.0) * e * (1.0 + SQRT5 * r);
    }
    kvec[(size_t)h * Np + j] = kv;
    dkdr2[(size_t)h * Np + j] = dv;
}

void launch_point_cov(hipStream_t s, const double* Xs, const double* s1, const double* hyp,
                      const double* htab, const double* x, double* kvec, double* dkdr2, int N, int Np,
                      int D, int Dp, int nh)
{
    hipLaunchKernelGGL(k_point_cov, dim3((Np + 255) / 256, nh), dim3(256), 0, s, Xs, s1, hyp, htab, x,
                       kvec, dkdr2, N, Np, D, Dp);
}

__device__ __forceinline__ double ndtr_r(double a)
{
#pragma clang fp contract(off)
    const double xx = a * 0.70710678118654752440;
    const double z = fabs(xx);
    if (z < 0.70710678118654752440) return 0.5 + 0.5 * erf(xx);
    double y = 0.5 * erfc(z);
    if (xx > 0) y = 1.0 - y;
    return y;
}

// one workgroup per draw: out[h][0] = EI_h(x), out[h][1 + d] = d(-EI_h)/dx_d in the reference's scaling
__global__ __launch_bounds__(256) void k_point_finish(
    const double* __restrict__ Xs, const double* __restrict__ hyp, const double* __restrict__ htab,
    const double* __restrict__ alpha, const double* __restrict__ kvec, const double* __restrict__ dkdr2,
    const double* __restrict__ tvec, const double* __restrict__ zvec, const double* __restrict__ x,
    double best, double* __restrict__ out, int N, int Np, int D, int Dp,
    // EI per second (GPEIperSecChooser.py:349-434): rows H..2H-1 of the tables hold the log-duration
    // GP; kt / dkt are its k and dk/dr2 at x.  Null -> plain EI.
    int H, const double* __restrict__ kt, const double* __restrict__ dkt)
{
    extern __shared__ double red[];   // [256] scratch for the block reductions
    __shared__ double sh_cdf, sh_w;
    const int h = blockIdx.x;
    const int tid = threadIdx.x;
    const double* ah = alpha + (size_t)h * Np;
    const double* kh = kvec + (size_t)h * Np;
    const double* th = tvec + (size_t)h * Np;
    const double* zh = zvec + (size_t)h * Np;
    const double* dh = dkdr2 + (size_t)h * Np;
    const double* ls = hyp + (size_t)h * (3 + D) + 3;
    const double mean = htab[h * SPX_HT + 0], amp2 = htab[h * SPX_HT + 2], prior_v = htab[h * SPX_HT + 3];

    double ka = 0.0, tt = 0.0;
    for (int j = tid; j < N; j += 256) {
        ka += kh[j] * ah[j];
        tt += th[j] * th[j];
    }
    auto block_sum = [&](double v) -> double {
        __syncthreads();
        red[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        return red[0];
    };
    ka = block_sum(ka);
    tt = block_sum(tt);
    if (tid == 0) {
#pragma clang fp contract(off)
        const double func_m = ka + mean;
        const double func_v = prior_v - tt;
        const double func_s = sqrt(func_v);
        const double u = (best - func_m) / func_s;
        const double cdf = ndtr_r(u);
        const double pdf = exp(-(u * u) / 2.0) / 2.50662827463100050242;
        out[(size_t)h * (1 + D)] = func_s * (u * cdf + pdf);
        sh_cdf = cdf;
        sh_w = 0.5 * pdf / func_s;
    }
    __syncthreads();
    const double g_m = -sh_cdf, g_s2 = sh_w;
    // time model of this draw (row h + H)
    const int ht = h + H;
    double time_m = 1.0;
    const double* lst = nullptr;
    const double* aht = nullptr;
    const double* kht = nullptr;
    const double* dht = nullptr;
    double amp2t = 0.0;
    if (kt) {
        lst = hyp + (size_t)ht * (3 + D) + 3;
        aht = alpha + (size_t)ht * Np;
        kht = kt + (size_t)h * Np;
        dht = dkt + (size_t)h * Np;
        amp2t = htab[ht * SPX_HT + 2];
        double kat = 0.0;
        for (int j = tid; j < N; j += 256) kat += kht[j] * aht[j];
        kat = block_sum(kat);
        time_m = exp(kat + htab[ht * SPX_HT + 0]);
    }
    const double ei = out[(size_t)h * (1 + D)];   // written by thread 0 above; visible after the barriers
    for (int d = 0; d < D; ++d) {
        const double xc = x[d] / ls[d];
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int j = tid; j < N; j += 256) {
            const double gj = dh[j] * (2.0 * (Xs[((size_t)h * Np + j) * Dp + d] - xc) * (1.0 / ls[d]));
            a1 += ah[j] * gj;
            a2 += zh[j] * gj;
            if (kt) {
                const double xct = x[d] / lst[d];
                const double gt = dht[j] * (2.0 * (Xs[((size_t)ht * Np + j) * Dp + d] - xct) * (1.0 / lst[d]));
                a3 += aht[j] * gt;
            }
        }
        a1 = block_sum(a1);
        a2 = block_sum(a2);
        if (kt) a3 = block_sum(a3);
        if (tid == 0) {
            double gd = 0.5 * amp2 * (a1 * g_m + (-2.0 * a2) * g_s2);
            if (kt) {
                const double gtd = 0.5 * amp2t * a3 * time_m;
                gd = (time_m * gd - ei * gtd) / (time_m * time_m);
            }
            out[(size_t)h * (1 + D) + 1 + d] = gd;
        }
    }
    __syncthreads();
    if (kt && tid == 0) out[(size_t)h * (1 + D)] = ei / time_m;
}

void launch_point_finish(hipStream_t s, const double* Xs, const double* hyp, const double* htab,
                         const double* alpha, const double* kvec, const double* dkdr2,
                         const double* tvec, const double* zvec, const double* x, double best,
                         double* out, int N, int Np, int D, int Dp, int nh, const double* kt, const double* dkt)
{
    hipLaunchKernelGGL(k_point_finish, dim3(nh), dim3(256), 256 * sizeof(double), s, Xs, hyp, htab, alpha,
                       kvec, dkdr2, tvec, zvec, x, best, out, N, Np, D, Dp, nh, kt, dkt);
}
