// Sobol candidate grid on the device.
//
// Replaces the pure-Python loop  ExperimentGrid._hypercube_grid -> i4_sobol_generate
//   (spearmint/spearmint/ExperimentGrid.py:192-196, sobol_lib.py:125-157 and i4_sobol :158-13787;
//    spearmint-lite/ExperimentGrid.py:186-191, :238-243, sobol_lib.py:124-431).
// The reference advances a global running vector one Gray-code step per call; the point it
// returns for seed s is  x_s[d] = (XOR over set bits b of s ^ (s >> 1) of V[d][b]) * 2^-30, so
// every (point, dimension) element is independent.  One thread per element of the row-major
// grid[n][dim] (the transposed layout ExperimentGrid keeps), consecutive threads on consecutive
// addresses: the kernel is a pure 8 B/element HBM write stream; the 30-column direction table of
// the `dim` rows in use sits in LDS (120 B per dimension).  Integer XORs and one exact scaling by
// a power of two: bit-identical to the reference.
#include "common.h"

#define SOBOL_NCOL 30
#define SOBOL_LDS_DIMS 512   // direction rows staged in LDS (60 KB); beyond that they are read through L2

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
    const int64_t total = n * dim;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    // (point, dimension) of e, advanced incrementally: one 64-bit division per thread, not per element
    int64_t i = e / dim;
    int d = (int)(e - i * dim);
    const int64_t di = stride / dim;
    const int dd = (int)(stride - di * dim);
    for (; e < total; e += stride) {
        int64_t s = skip - 1 + i;                       // seed = skip + j - 2, j = i + 1 (sobol_lib.py:153-156)
        if (s < 0) s = 0;                               // i4_sobol: "if seed < 0: seed = 0"
        uint32_t g = (uint32_t)(s ^ (s >> 1));          // s < 2^30 (checked by the caller)
        uint32_t x = 0;
        const uint32_t* Vd = V + d * SOBOL_NCOL;
        while (g) {
            x ^= Vd[__builtin_ctz(g)];
            g &= g - 1;
        }
        out[e] = (double)x * 9.31322574615478515625e-10;   // recipd = 2^-30
        i += di;
        d += dd;
        if (d >= dim) { d -= dim; ++i; }
    }
}

void launch_sobol_grid(hipStream_t s, const uint32_t* dirs, int dim, int64_t n, int64_t skip, double* out)
{
    const int64_t total = n * dim;
    // ~16 elements per thread amortise the table fill; at least one block
    int64_t blocks = (total + 256 * 16 - 1) / (256 * 16);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    if (dim <= SOBOL_LDS_DIMS) {
        const size_t lds = (size_t)dim * SOBOL_NCOL * sizeof(uint32_t);
        hipLaunchKernelGGL(k_sobol_grid<true>, dim3((unsigned)blocks), dim3(256), lds, s, dirs, dim, n, skip, out);
    } else {
        hipLaunchKernelGGL(k_sobol_grid<false>, dim3((unsigned)blocks), dim3(256), 0, s, dirs, dim, n, skip, out);
    }
}
