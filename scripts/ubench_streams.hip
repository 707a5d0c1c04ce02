// Dev micro-benchmark: cost of dependent kernel boundaries on one stream, across two streams
// (event record + stream wait per hop), with already-satisfied cross-stream waits, and as a hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void tiny(double* p, int spin) { double v = p[0]; for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9; if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    double* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const int n = 96;
    std::vector<hipEvent_t> ev(2 * n);
    for (size_t i = 0; i < ev.size(); ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    for (int spin : {0, 2000}) {
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, a, d, spin);
            CK(hipStreamSynchronize(a));
            double t1 = now();
            if (rep) printf("spin %4d  one stream, %d dependent launches:            %.2f us per kernel\n", spin, n, (t1 - t0) / n * 1e6);
        }
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            for (int i = 0; i < n; ++i) {
                hipStream_t s = (i & 1) ? b : a, o = (i & 1) ? a : b;
                if (i) CK(hipStreamWaitEvent(s, ev[i - 1], 0));
                hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, spin);
                CK(hipEventRecord(ev[i], s));
                (void)o;
            }
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            double t1 = now();
            if (rep) printf("spin %4d  two streams ping-pong (record + wait per hop):   %.2f us per kernel\n", spin, (t1 - t0) / n * 1e6);
        }
        for (int rep = 0; rep < 2; ++rep) {
            // chain on a; every kernel also waits an event of stream b recorded two hops earlier (already satisfied)
            double t0 = now();
            for (int i = 0; i < n; ++i) {
                if (i >= 2) CK(hipStreamWaitEvent(a, ev[n + i - 2], 0));
                hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, a, d + 1, spin);
                CK(hipEventRecord(ev[i], a));
                CK(hipStreamWaitEvent(b, ev[i], 0));
                hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, b, d + 2, spin);
                hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, b, d + 3, spin);
                CK(hipEventRecord(ev[n + i], b));
            }
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            double t1 = now();
            if (rep) printf("spin %4d  lookahead pattern (1 kernel on A, 2 on B per step): %.2f us per step\n", spin, (t1 - t0) / n * 1e6);
        }
        // the same lookahead pattern captured into a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(a, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; ++i) {
            if (i >= 2) CK(hipStreamWaitEvent(a, ev[n + i - 2], 0));
            hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, a, d + 1, spin);
            CK(hipEventRecord(ev[i], a));
            CK(hipStreamWaitEvent(b, ev[i], 0));
            hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, b, d + 2, spin);
            hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, b, d + 3, spin);
            CK(hipEventRecord(ev[n + i], b));
        }
        CK(hipStreamWaitEvent(a, ev[2 * n - 1], 0));
        CK(hipStreamEndCapture(a, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipGraphLaunch(ge, a)); CK(hipStreamSynchronize(a));
            double t1 = now();
            if (rep) printf("spin %4d  lookahead pattern as a hipGraph:                   %.2f us per step\n", spin, (t1 - t0) / n * 1e6);
        }
        // single-stream chain as a graph
        hipGraph_t g2; hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(a, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, a, d, spin);
        CK(hipStreamEndCapture(a, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipGraphLaunch(ge2, a)); CK(hipStreamSynchronize(a));
            double t1 = now();
            if (rep) printf("spin %4d  one-stream chain as a hipGraph:                    %.2f us per kernel\n", spin, (t1 - t0) / n * 1e6);
        }
    }
    return 0;
}
