// Developer microbenchmark (GPU box): launch-to-launch time of a chain of dependent tiny kernels on one
// stream -- plain launches vs one hipGraph replay of the same chain (stream capture).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/graph_gap.hip -o tools/tmp/graph_gap && tools/tmp/graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void bump(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int chain = 64;
    for (int n : {256, 65536, 4 << 20}) {
        double* p;
        CK(hipMalloc(&p, sizeof(double) * n));
        CK(hipMemset(p, 0, sizeof(double) * n));
        hipStream_t s;
        CK(hipStreamCreate(&s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const dim3 grid((n + 255) / 256), block(256);
        auto run_stream = [&]() { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(bump, grid, block, 0, s, p, n); };
        run_stream(); CK(hipStreamSynchronize(s));
        float best_s = 1e9f, best_g = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            CK(hipEventRecord(e0, s)); run_stream(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_s) best_s = ms;
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        run_stream();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 10; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_g) best_g = ms;
        }
        printf("n=%8d  chain of %d dependent launches: stream %.2f us per launch, graph replay %.2f us per launch\n",
               n, chain, 1e3 * best_s / chain, 1e3 * best_g / chain);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(p)); CK(hipStreamDestroy(s));
    }
    return 0;
}
