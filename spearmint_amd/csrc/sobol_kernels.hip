// Sobol candidate grid on the device.
//
// Replaces the pure-Python loop  ExperimentGrid._hypercube_grid -> i4_sobol_generate
//   (spearmint/spearmint/ExperimentGrid.py:192-196, sobol_lib.py:125-157 and i4_sobol :158-13787;
//    spearmint-lite/ExperimentGrid.py:186-191, :238-243, sobol_lib.py:124-431).
// The reference advances a global running vector one Gray-code step per call; the point it
// returns for seed s is  x_s[d] = (XOR over set bits b of s ^ (s >> 1) of V[d][b]) * 2^-30, so
// a point is computable directly, and the next seed's point differs from it by one direction
// integer (the Gray-code step the reference itself takes: lastq ^= v[:, lo0(seed)]).
// One thread owns one dimension d of a run of SOBOL_RUN consecutive points: the first by the
// direct XOR over the set bits of its Gray code, the rest by the one-XOR recurrence.  Lanes are
// laid out dimension-fastest, so one store instruction writes whole rows of the row-major
// grid[n][dim] (the transposed layout ExperimentGrid keeps): the kernel is a pure 8 B/element
// HBM write stream; the 30-column direction table of the `dim` rows in use sits in LDS (120 B per
// dimension).  Integer XORs and one exact scaling by a power of two: bit-identical to the
// reference.
#include "common.h"

#define SOBOL_NCOL 30
#define SOBOL_LDS_DIMS 512   // direction rows staged in LDS (60 KB); beyond that they are read through L2

#define SOBOL_RUN 16       // consecutive points per thread

template <bool LDS>
__global__ __launch_bounds__(256) void k_sobol_grid(const uint32_t* __restrict__ dirs, int dim, int64_t n,
                                                    int64_t skip, double* __restrict__ out)
{
    extern __shared__ uint32_t Vs[];
    if (LDS) {
        for (int t = threadIdx.x; t < dim * SOBOL_NCOL; t += 256) Vs[t] = dirs[t];
        __syncthreads();
    }
    const uint32_t* V = LDS ? Vs : dirs;
    const int64_t nruns = (n + SOBOL_RUN - 1) / SOBOL_RUN;
    const int64_t total = nruns * dim;                     // (run, dimension) pairs, dimension fastest
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t T = (int64_t)blockIdx.x * 256 + threadIdx.x; T < total; T += stride) {
        const int64_t q = T / dim;
        const int d = (int)(T - q * dim);
        const uint32_t* Vd = V + d * SOBOL_NCOL;
        const int64_t i0 = q * SOBOL_RUN;
        // seed of point i: skip + j - 2 with j = i + 1 (sobol_lib.py:153-156); i4_sobol maps a
        // negative seed to 0; seeds < 2^30 (checked by the caller)
        int64_t s = skip - 1 + i0;
        if (s < 0) s = 0;
        uint32_t x = 0;
        for (uint32_t g = (uint32_t)(s ^ (s >> 1)); g; g &= g - 1) x ^= Vd[__builtin_ctz(g)];
        double* o = out + i0 * dim + d;
        const int nr = (int)((n - i0 < SOBOL_RUN) ? (n - i0) : SOBOL_RUN);
#pragma unroll
        for (int r = 0; r < SOBOL_RUN; ++r) {
            if (r < nr) o[(int64_t)r * dim] = (double)x * 9.31322574615478515625e-10;   // recipd = 2^-30
            // step to the next point: one Gray-code flip, unless the seed is still clamped at 0
            const int64_t sn = skip + i0 + r;               // unclamped seed of point i0 + r + 1
            // (the step past seed 2^30 - 1 would need column 30: never stored, index clamped)
            if (sn > 0) x ^= Vd[min(__builtin_ctz(~(uint32_t)(sn - 1)), SOBOL_NCOL - 1)];
        }
    }
}

void launch_sobol_grid(hipStream_t s, const uint32_t* dirs, int dim, int64_t n, int64_t skip, double* out)
{
    const int64_t total = ((n + SOBOL_RUN - 1) / SOBOL_RUN) * dim;   // threads' worth of work
    // two runs per thread amortise the table fill; at least one block
    int64_t blocks = (total + 256 * 2 - 1) / (256 * 2);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    if (dim <= SOBOL_LDS_DIMS) {
        const size_t lds = (size_t)dim * SOBOL_NCOL * sizeof(uint32_t);
        hipLaunchKernelGGL(k_sobol_grid<true>, dim3((unsigned)blocks), dim3(256), lds, s, dirs, dim, n, skip, out);
    } else {
        hipLaunchKernelGGL(k_sobol_grid<false>, dim3((unsigned)blocks), dim3(256), 0, s, dirs, dim, n, skip, out);
    }
}
