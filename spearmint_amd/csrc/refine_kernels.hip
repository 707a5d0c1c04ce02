// EI and its gradient at a BATCH of points, for every resident hyper-parameter draw -- the
// objective of the reference's local refinement:
//   GPEIOptChooser.py:391-440  grad_optimize_ei (no-pending branch), :441-525 (pending branch,
//   EI averaged over the fantasies), summed over draws by grad_optimize_ei_over_hypers
//   (:360-388), minimised by L-BFGS-B (:39-43, :285-289); per second:
//   GPEIperSecChooser.py:349-434.
// The reference evaluates ONE point per call from `grid_subset` (20) serial / forked L-BFGS
// runs; here all points that are waiting for an evaluation go through one call, so W = L^-1
// (33.5 MB per draw at N = 2048) is streamed once per group of PB points, not once per point.
//
// Per draw h and point x, with k = amp2 * Matern52(ls; comp, x)  (N),  W = L^-1:
//   t = W k            (beta, :413)            z = W^T t = K^-1 k        (:436)
//   func_m = k.alpha + mean, func_v = amp2(1+1e-6) - |t|^2, EI as in predict_kernels.hip
//   G[j][d] = dk/dr2 (r_j) * 2 (comp_jd/ls_d - x_d/ls_d) / ls_d        (gp.py:129-132, :56-85)
//   grad[d] = 0.5 amp2 ( (alpha . G[:,d]) (-Phi) + (-2 z . G[:,d]) (0.5 phi / s) )   (:427-437,
//             including the reference's factor one half)
// With S fantasies (observations = [comp; pend]): func_m[s] = t . Gamma_s + mean with
// Gamma_s = W (fant_s - mean) (equal to k . alpha_s), EI_s against bests[s], and
//   grad[d] = 0.5 amp2 ( sum_j G[j][d] u[j] + (-2 z . G[:,d]) mean_s(0.5 phi_s / s) ),
//   u[j] = mean_s( -Phi_s alpha_s[j] ),  alpha_s = W^T Gamma_s                    (:505-523)
// Every point's numbers are computed by its own threads in a fixed order, so a result does not
// depend on which other points share the call.
#include "common.h"

#define SQRT5 2.23606797749978969641
#define SQRT3 1.73205080756887719318
#define PB SPX_REFINE_PB   // right-hand sides that share one pass over W

// k[h][p][j] = amp2 * corr(r_j), dkdr2[h][p][j] = d corr / d r^2 (gp.py:102-105, :115-118, :129-132):
//   Matern-5/2  -(5/6) exp(-sqrt5 r)(1 + sqrt5 r);  Matern-3/2  -1.5 exp(-sqrt3 r);  ARDSE  -0.5 exp(-r^2/2)
// pad rows -> 0
__global__ __launch_bounds__(256) void k_point_cov(
    const double* __restrict__ Xs /*[nh][Np][Dp]*/, const double* __restrict__ s1 /*[nh][Np]*/,
    const double* __restrict__ hyp /*[nh][3+D]*/, const double* __restrict__ htab,
    const double* __restrict__ x /*[P][D]*/, double* __restrict__ kvec, double* __restrict__ dkdr2,
    int N, int Np, int D, int Dp, int P, int kind)
{
#pragma clang fp contract(off)
    const int h = blockIdx.y, p = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Np) return;
    const double* ls = hyp + (size_t)h * (3 + D) + 3;
    const double* xp = x + (size_t)p * D;
    const double amp2 = htab[h * SPX_HT + 2];
    const double* xj = Xs + ((size_t)h * Np + j) * Dp;
    double kv = 0.0, dv = 0.0;
    if (j < N) {
        // same expanded form as gp.dist2: -( (xx1 . 2 xx2 - |xx1|^2) - |xx2|^2 ), clamped at 0
        double g = 0.0, s2 = 0.0;
        for (int d = 0; d < D; ++d) {
            const double xc = xp[d] / ls[d];
            g = g + xj[d] * (2.0 * xc);
            s2 = s2 + xc * xc;
        }
        const double t = (g - s1[(size_t)h * Np + j]) - s2;
        const double nt = -t;
        double r2 = (nt < 0.0) ? 0.0 : nt;
        r2 = fabs(r2);
        const double r = sqrt(r2);
        if (kind == SPX_COV_MATERN52) {
            const double e = exp(-SQRT5 * r);
            kv = amp2 * (((1.0 + SQRT5 * r) + (5.0 / 3.0) * r2) * e);
            dv = -(5.0 / 6.0) * e * (1.0 + SQRT5 * r);
        } else if (kind == SPX_COV_MATERN32) {
            const double e = exp(-SQRT3 * r);
            kv = amp2 * ((1.0 + SQRT3 * r) * e);
            dv = -1.5 * e;
        } else {
            const double e = exp(-0.5 * r2);
            kv = amp2 * e;
            dv = -0.5 * e;
        }
    }
    const size_t o = ((size_t)h * P + p) * Np + j;
    kvec[o] = kv;
    dkdr2[o] = dv;
}

void launch_point_cov(hipStream_t s, const double* Xs, const double* s1, const double* hyp,
                      const double* htab, const double* x, double* kvec, double* dkdr2, int N, int Np,
                      int D, int Dp, int nh, int P, int kind)
{
    hipLaunchKernelGGL(k_point_cov, dim3((Np + 255) / 256, nh, P), dim3(256), 0, s, Xs, s1, hyp, htab, x,
                       kvec, dkdr2, N, Np, D, Dp, P, kind);
}

// out[h][p][i] = sum_{j <= i} WT_h[j][i] rhs[h][p][j]   (t = W k), PB right-hand sides per pass over W.
// A workgroup owns 64 rows i; its four wavefronts split the j range of those rows four ways (wave w takes
// j = w, w + 4, ...: a quarter of the serial depth each -- the loop is latency-bound, one dependent chain per
// row), and the four partial sums of a row are added in the fixed order w = 0..3.  Each partial sum runs over
// its j in increasing order, so a point's result does not depend on what else is in the batch.
__global__ __launch_bounds__(256) void k_trimv_multi(const double* __restrict__ WT,
                                                     const double* __restrict__ rhs,
                                                     double* __restrict__ out, int Np, int P)
{
    // one array for both uses: r[q][t] = sm[q * 256 + t] during the loop, part[w][q][lane] = sm[(w * PB + q) * 64 + lane] after it
    __shared__ double sm[PB * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y, p0 = blockIdx.z * PB;
    const int np = min(PB, P - p0);
    const int i = blockIdx.x * 64 + lane;
    const double* Wh = WT + (size_t)h * Np * Np;
    const double* rh = rhs + ((size_t)h * P + p0) * Np;
    double acc[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) acc[q] = 0.0;
    const int jmax = blockIdx.x * 64 + 63;          // largest row of this workgroup
    for (int jb = 0; jb <= jmax; jb += 256) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PB; ++q)
            sm[q * 256 + threadIdx.x] = (q < np && jb + threadIdx.x < Np) ? rh[(size_t)q * Np + jb + threadIdx.x] : 0.0;
        __syncthreads();
        const int jn = (i < Np) ? min(256, i - jb + 1) : 0;     // this row needs j = jb .. jb + jn - 1
        // eight loads of W in flight per wave (the loop is one dependent chain per row: with the compiler's own
        // unrolling it was a memory round trip per four rows); the sums still run over j in increasing order
        const double* wp = Wh + (size_t)jb * Np + i;
        int t = wave;
        for (; t + 28 < jn; t += 32) {
            double w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = wp[(size_t)(t + 4 * u) * Np];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < PB; ++q) acc[q] = fma(w[u], sm[q * 256 + t + 4 * u], acc[q]);
        }
        for (; t < jn; t += 4) {
            const double w = wp[(size_t)t * Np];
#pragma unroll
            for (int q = 0; q < PB; ++q) acc[q] = fma(w, sm[q * 256 + t], acc[q]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PB; ++q) sm[(wave * PB + q) * 64 + lane] = acc[q];
    __syncthreads();
    if (wave == 0 && i < Np)
        for (int q = 0; q < np; ++q)
            out[((size_t)h * P + p0 + q) * Np + i] =
                ((sm[(0 * PB + q) * 64 + lane] + sm[(1 * PB + q) * 64 + lane]) + sm[(2 * PB + q) * 64 + lane]) + sm[(3 * PB + q) * 64 + lane];
}

void launch_trimv_multi(hipStream_t s, const double* WT, const double* rhs, double* out, int Np, int nh, int P)
{
    hipLaunchKernelGGL(k_trimv_multi, dim3((Np + 63) / 64, nh, (P + PB - 1) / PB), dim3(256), 0, s, WT, rhs,
                       out, Np, P);
}

// out[h][p][j] = sum_{i >= j} WT_h[j][i] rhs[h][p][i]   (z = W^T t): one wavefront per row j
__global__ __launch_bounds__(256) void k_trimvT_multi(const double* __restrict__ WT,
                                                      const double* __restrict__ rhs,
                                                      double* __restrict__ out, int Np, int P)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y, p0 = blockIdx.z * PB;
    const int np = min(PB, P - p0);
    const int j = blockIdx.x * 4 + wave;
    const double* row = WT + ((size_t)h * Np + j) * Np;
    const double* rh = rhs + ((size_t)h * P + p0) * Np;
    double acc[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) acc[q] = 0.0;
    // four 64-element segments of the row per trip (their loads in flight together: the row is a chain of memory round trips
    // otherwise); each lane still adds its elements in increasing i, one fma each: the same sums
    int i = (j & ~63) + lane;
    for (; i + 192 < Np; i += 256) {
        double w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = (i + 64 * u >= j) ? row[i + 64 * u] : 0.0;
#pragma unroll
        for (int q = 0; q < PB; ++q)
            if (q < np) {
                double rv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rv[u] = rh[(size_t)q * Np + i + 64 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + 64 * u >= j) acc[q] = fma(w[u], rv[u], acc[q]);
            }
    }
    for (; i < Np; i += 64)
        if (i >= j) {
            const double w = row[i];
#pragma unroll
            for (int q = 0; q < PB; ++q)
                if (q < np) acc[q] = fma(w, rh[(size_t)q * Np + i], acc[q]);
        }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_xor(acc[q], off);
    }
    if (lane == 0)
        for (int q = 0; q < np; ++q) out[((size_t)h * P + p0 + q) * Np + j] = acc[q];
}

void launch_trimvT_multi(hipStream_t s, const double* WT, const double* rhs, double* out, int Np, int nh, int P)
{
    hipLaunchKernelGGL(k_trimvT_multi, dim3(Np / 4, nh, (P + PB - 1) / PB), dim3(256), 0, s, WT, rhs, out, Np, P);
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

// sum over the 256 threads of a workgroup; every thread gets the result
__device__ __forceinline__ double block_sum(double v, double* red /*[4]*/)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

#define GD 8   // input dimensions per gradient pass

// one workgroup per (draw, point): out[h][p][0] = EI_h(x_p) (mean over fantasies if S > 0),
// out[h][p][1 + d] = d(-EI_h)/dx_d in the reference's scaling
__global__ __launch_bounds__(256) void k_point_finish(
    const double* __restrict__ Xs, const double* __restrict__ hyp, const double* __restrict__ htab,
    const double* __restrict__ alpha, const double* __restrict__ kvec, const double* __restrict__ dkdr2,
    const double* __restrict__ tvec, const double* __restrict__ zvec, const double* __restrict__ x,
    double best, double* __restrict__ out, int N, int Np, int D, int Dp, int P,
    // EI per second (GPEIperSecChooser.py:349-434): rows H..2H-1 of the tables hold the log-duration
    // GP; kt / dkt are its k and dk/dr2 at the points.  Null -> plain EI.
    int H, const double* __restrict__ kt, const double* __restrict__ dkt,
    // pending-experiment fantasies (GPEIOptChooser.py:441-525): S right-hand sides per draw
    int S, const double* __restrict__ gammaS /*[H][S][Np]*/, const double* __restrict__ alphaS /*[H][S][Np]*/,
    const double* __restrict__ bests /*[H][S]*/, double* __restrict__ uvec /*[H][P][Np] scratch*/)
{
    extern __shared__ double dyn[];        // S > 0: [S] -Phi_s / S
    __shared__ double red[4];
    __shared__ double redv[4][3 * GD];
    __shared__ double sh_ei, sh_gm, sh_gs2;
    const int h = blockIdx.x, p = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t vo = ((size_t)h * P + p) * Np;
    const double* ah = alpha + (size_t)h * Np;
    const double* kh = kvec + vo;
    const double* th = tvec + vo;
    const double* zh = zvec + vo;
    const double* dh = dkdr2 + vo;
    const double* xp = x + (size_t)p * D;
    const double* ls = hyp + (size_t)h * (3 + D) + 3;
    const double mean = htab[h * SPX_HT + 0], amp2 = htab[h * SPX_HT + 2], prior_v = htab[h * SPX_HT + 3];
    double* o = out + ((size_t)h * P + p) * (1 + D);

    double ka = 0.0, tt = 0.0;
    for (int j = tid; j < N; j += 256) {
        ka += kh[j] * ah[j];
        tt += th[j] * th[j];
    }
    tt = block_sum(tt, red);
    if (S == 0) {
        ka = block_sum(ka, red);
        if (tid == 0) {
#pragma clang fp contract(off)
            const double func_m = ka + mean;
            const double func_v = prior_v - tt;
            const double func_s = sqrt(func_v);
            const double u = (best - func_m) / func_s;
            const double cdf = ndtr_r(u);
            const double pdf = exp(-(u * u) / 2.0) / 2.50662827463100050242;
            sh_ei = func_s * (u * cdf + pdf);
            sh_gm = -cdf;
            sh_gs2 = 0.5 * pdf / func_s;
        }
        __syncthreads();
    } else {
        // every fantasy: func_m[s] = t . Gamma_s + mean; one wavefront per s
        double* gmS = dyn;                 // [S]  -Phi_s
        double* eiS = dyn + S;             // [S]
        double* gsS = dyn + 2 * S;         // [S]  0.5 phi_s / func_s
        const double func_v = prior_v - tt;
        const double func_s = sqrt(func_v);
        for (int sidx = wave; sidx < S; sidx += 4) {
            const double* gs = gammaS + ((size_t)h * S + sidx) * Np;
            double m = 0.0;
            for (int j = lane; j < N; j += 64) m += th[j] * gs[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off);
            if (lane == 0) {
#pragma clang fp contract(off)
                const double func_m = m + mean;
                const double u = (bests[(size_t)h * S + sidx] - func_m) / func_s;
                const double cdf = ndtr_r(u);
                const double pdf = exp(-(u * u) / 2.0) / 2.50662827463100050242;
                eiS[sidx] = func_s * (u * cdf + pdf);
                gmS[sidx] = -cdf;
                gsS[sidx] = 0.5 * pdf / func_s;
            }
        }
        __syncthreads();
        if (tid == 0) {
            double e = 0.0, g2 = 0.0;
            for (int sidx = 0; sidx < S; ++sidx) { e += eiS[sidx]; g2 += gsS[sidx]; }
            sh_ei = e / (double)S;
            sh_gs2 = g2 / (double)S;
            sh_gm = 1.0;
        }
        // u[j] = mean_s( -Phi_s alpha_s[j] )
        double* uh = uvec + vo;
        for (int j = tid; j < Np; j += 256) {
            double a = 0.0;
            if (j < N)
                for (int sidx = 0; sidx < S; ++sidx) a += gmS[sidx] * alphaS[((size_t)h * S + sidx) * Np + j];
            uh[j] = a / (double)S;
        }
        __syncthreads();
        ah = uh;     // the mean-gradient weights of the fantasy branch (g_m folded in)
    }
    const double g_m = sh_gm, g_s2 = sh_gs2, ei = sh_ei;

    // time model of this draw (row h + H)
    const int ht = h + H;
    double time_m = 1.0, amp2t = 0.0;
    const double *lst = nullptr, *aht = nullptr, *dht = nullptr;
    if (kt) {
        lst = hyp + (size_t)ht * (3 + D) + 3;
        aht = alpha + (size_t)ht * Np;
        dht = dkt + vo;
        amp2t = htab[ht * SPX_HT + 2];
        const double* kht = kt + vo;
        double kat = 0.0;
        for (int j = tid; j < N; j += 256) kat += kht[j] * aht[j];
        kat = block_sum(kat, red);
        time_m = exp(kat + htab[ht * SPX_HT + 0]);
    }
    for (int d0 = 0; d0 < D; d0 += GD) {
        const int nd = min(GD, D - d0);
        double a1[GD], a2[GD], a3[GD], xc[GD], il[GD], xct[GD], ilt[GD];
#pragma unroll
        for (int q = 0; q < GD; ++q) {
            a1[q] = a2[q] = a3[q] = 0.0;
            const int d = d0 + (q < nd ? q : 0);
            xc[q] = xp[d] / ls[d];
            il[q] = 1.0 / ls[d];
            xct[q] = kt ? xp[d] / lst[d] : 0.0;
            ilt[q] = kt ? 1.0 / lst[d] : 0.0;
        }
        for (int j = tid; j < N; j += 256) {
            const double* xr = Xs + ((size_t)h * Np + j) * Dp + d0;
            const double dj = dh[j], aj = ah[j], zj = zh[j];
#pragma unroll
            for (int q = 0; q < GD; ++q)
                if (q < nd) {
                    const double gj = dj * (2.0 * (xr[q] - xc[q]) * il[q]);
                    a1[q] += aj * gj;
                    a2[q] += zj * gj;
                }
            if (kt) {
                const double* xt = Xs + ((size_t)ht * Np + j) * Dp + d0;
                const double djt = dht[j], ajt = aht[j];
#pragma unroll
                for (int q = 0; q < GD; ++q)
                    if (q < nd) a3[q] += ajt * (djt * (2.0 * (xt[q] - xct[q]) * ilt[q]));
            }
        }
#pragma unroll
        for (int q = 0; q < GD; ++q) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                a1[q] += __shfl_xor(a1[q], off);
                a2[q] += __shfl_xor(a2[q], off);
                a3[q] += __shfl_xor(a3[q], off);
            }
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < GD; ++q) {
                redv[wave][q] = a1[q];
                redv[wave][GD + q] = a2[q];
                redv[wave][2 * GD + q] = a3[q];
            }
        }
        __syncthreads();
        if (tid < nd) {
            const int q = tid;
            const double s1v = ((redv[0][q] + redv[1][q]) + redv[2][q]) + redv[3][q];
            const double s2v = ((redv[0][GD + q] + redv[1][GD + q]) + redv[2][GD + q]) + redv[3][GD + q];
            double gd = 0.5 * amp2 * (s1v * g_m + (-2.0 * s2v) * g_s2);
            if (kt) {
                const double s3v = ((redv[0][2 * GD + q] + redv[1][2 * GD + q]) + redv[2][2 * GD + q]) + redv[3][2 * GD + q];
                const double gtd = 0.5 * amp2t * s3v * time_m;
                gd = (time_m * gd - ei * gtd) / (time_m * time_m);
            }
            o[1 + d0 + q] = gd;
        }
    }
    if (tid == 0) o[0] = kt ? ei / time_m : ei;
}

void launch_point_finish(hipStream_t s, const double* Xs, const double* hyp, const double* htab,
                         const double* alpha, const double* kvec, const double* dkdr2,
                         const double* tvec, const double* zvec, const double* x, double best,
                         double* out, int N, int Np, int D, int Dp, int nh, int P, const double* kt,
                         const double* dkt, int S, const double* gammaS, const double* alphaS,
                         const double* bests, double* uvec)
{
    const size_t lds = (size_t)3 * S * sizeof(double);   // up to 96 KB at S = 4096: above the 64 KB default limit
    if (lds > 48 * 1024)
        SPX_LDS_ATTR(k_point_finish, lds);
    hipLaunchKernelGGL(k_point_finish, dim3(nh, P), dim3(256), lds, s, Xs, hyp,
                       htab, alpha, kvec, dkdr2, tvec, zvec, x, best, out, N, Np, D, Dp, P, nh, kt, dkt, S,
                       gammaS, alphaS, bests, uvec);
}
