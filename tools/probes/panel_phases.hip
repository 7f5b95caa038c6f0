// developer probe: per-phase cycles of potrf_panel_kernel (s_memtime stamps, workgroup 0)
#define APGP_PANEL_TIMING 1
#include "../approxposterior_amd/csrc/potrf.hip"
#include "../approxposterior_amd/csrc/gram.hip"
#include "../approxposterior_amd/csrc/linalg.hip"
#include <vector>
int main() {
    const long long n = 4096;
    std::vector<double> h(n * n, 0.0), y(n, 1.0);
    for (long long i = 0; i < n; ++i) for (long long j = 0; j <= i; ++j) h[i * n + j] = (i == j) ? 4.0 + 0.001 * i : 0.5 / (1.0 + (i - j));
    double *A, *yd, *z; int* info;
    (void)hipMalloc(&A, n * n * 8); (void)hipMalloc(&yd, n * 8); (void)hipMalloc(&z, n * 8); (void)hipMalloc(&info, 4);
    (void)hipMemcpy(yd, y.data(), n * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice);
        int rc = apgp_potrf(A, n, n, yd, 0.0, z, info, nullptr);
        (void)hipDeviceSynchronize();
        unsigned long long s[16];
        (void)hipMemcpyFromSymbol(s, HIP_SYMBOL(apgp_panel_stamps), sizeof(s));
        printf("rc %d | panel step at column 2048, workgroup 0, cycles after the factorising wavefront's start: block loaded %llu | potf2 done %llu | "
               "write-back + z-solve done %llu | (solving wavefront) panel solve done %llu | rows stored + rhs updated %llu\n", rc,
               s[1] - s[0], s[2] - s[0], s[3] - s[0], s[4] - s[0], s[5] - s[0]);
        unsigned long long q[16];
        (void)hipMemcpyFromSymbol(q, HIP_SYMBOL(apgp_step_stamps), sizeof(q));
        printf("   fused step at column 2048 (10 ns ticks after the workgroup's start; workgroup 0 | workgroup 1): tile updated %lld | %lld ; diagonal tile updated %lld | %lld ; "
               "factor starts %lld | %lld ; factor done %lld | %lld ; wg1 rows solved %lld ; wg1 started %lld after wg0\n",
               (long long)(q[1] - q[0]), (long long)(q[7] - q[6]), (long long)(q[2] - q[0]), (long long)(q[8] - q[6]), (long long)(q[3] - q[0]), (long long)(q[9] - q[6]),
               (long long)(q[4] - q[0]), (long long)(q[10] - q[6]), (long long)(q[11] - q[6]), (long long)(q[6] - q[0]));
        printf("   first column group: diagonal block broadcast %llu | its factor %llu | row solve %llu | published %llu | columns to the right updated %llu\n",
               s[6] - s[1], s[7] - s[6], s[8] - s[7], s[9] - s[8], s[10] - s[9]);
    }
    return 0;
}
