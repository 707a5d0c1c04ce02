// numpy's pairwise summation (loops_utils.h.src: < 8 sequential, <= 128 eight accumulators, else halves rounded to
// multiples of 8), restated without recursion.  Shared by the device kernels (predict_kernels.hip) and by a host-compiled
// test (tests/test_host_logic.py builds this header with g++ and compares with np.sum bit for bit).
#pragma once
#include <stdint.h>
#ifndef SPX_HD
#define SPX_HD __device__ __forceinline__
#endif

// One leaf of numpy's pairwise sum (n <= 128): < 8 sequential from -0.0, else eight accumulators and a tail.
SPX_HD double np_pairwise_leaf(const double* a, int64_t stride, int n)
{
#pragma clang fp contract(off)
    if (n < 8) {
        double res = -0.0;
        for (int i = 0; i < n; ++i) res += a[i * stride];
        return res;
    }
    double r0 = a[0], r1 = a[stride], r2 = a[2 * stride], r3 = a[3 * stride];
    double r4 = a[4 * stride], r5 = a[5 * stride], r6 = a[6 * stride], r7 = a[7 * stride];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        const double* q = a + (int64_t)i * stride;
        r0 += q[0];          r1 += q[stride];     r2 += q[2 * stride]; r3 += q[3 * stride];
        r4 += q[4 * stride]; r5 += q[5 * stride]; r6 += q[6 * stride]; r7 += q[7 * stride];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i * stride];
    return res;
}

// The halving above 128 elements, WITHOUT recursion: a recursive __device__ function gives every kernel that calls it a
// private segment (48 B per lane), and the runtime then keeps per-queue scratch alive behind every stream that ever ran
// one -- device memory that spx_destroy cannot give back.  The tree (split at n/2 rounded down to a multiple of 8,
// left + right) is walked depth first with two small stacks that are shift registers (constant indices only, so they
// live in registers): pending segments {start, length} or the marker "add the two top values", and the values.
// n is uniform over the launch, so the walk is too.  Depth: n <= 128 << PW_DEPTH.
#define PW_DEPTH 12
#define NP_PAIRWISE_MAX_N (128 << PW_DEPTH)   // 524 288: the API refuses more draws / fantasies than this (spx_set_hypers)
SPX_HD double np_pairwise(const double* a, int64_t stride, int n)
{
#pragma clang fp contract(off)
    if (n <= 128) return np_pairwise_leaf(a, stride, n);
    int ws[2 * PW_DEPTH + 2], wn[2 * PW_DEPTH + 2];   // wn < 0: the add marker
    double vs[PW_DEPTH + 2];
#pragma unroll
    for (int q = 0; q < 2 * PW_DEPTH + 2; ++q) { ws[q] = 0; wn[q] = 0; }
#pragma unroll
    for (int q = 0; q < PW_DEPTH + 2; ++q) vs[q] = 0.0;
    ws[0] = 0; wn[0] = n;
    int depth = 1;
    while (depth > 0) {
        const int st = ws[0], len = wn[0];
#pragma unroll
        for (int q = 0; q < 2 * PW_DEPTH + 1; ++q) { ws[q] = ws[q + 1]; wn[q] = wn[q + 1]; }   // pop
        --depth;
        if (len < 0) {                     // add the two most recent values
            const double s = vs[1] + vs[0];
#pragma unroll
            for (int q = 1; q < PW_DEPTH + 1; ++q) vs[q] = vs[q + 1];
            vs[0] = s;
        } else if (len <= 128) {           // leaf
            const double s = np_pairwise_leaf(a + (int64_t)st * stride, stride, len);
#pragma unroll
            for (int q = PW_DEPTH + 1; q > 0; --q) vs[q] = vs[q - 1];
            vs[0] = s;
        } else {                           // split: left, right, add
            int n2 = len / 2;
            n2 -= n2 % 8;
#pragma unroll
            for (int q = 2 * PW_DEPTH + 1; q > 2; --q) { ws[q] = ws[q - 3]; wn[q] = wn[q - 3]; }
            ws[0] = st;      wn[0] = n2;
            ws[1] = st + n2; wn[1] = len - n2;
            ws[2] = 0;       wn[2] = -1;
            depth += 3;
        }
    }
    return vs[0];
}

