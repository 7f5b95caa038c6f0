// developer probe: apgp_potrf against a host Cholesky (residual of L and of the carried forward solve)
#include "../approxposterior_amd/csrc/potrf.hip"
#include "../approxposterior_amd/csrc/gram.hip"
#include "../approxposterior_amd/csrc/linalg.hip"
#include <vector>
#include <cmath>
int main() {
    for (long long n : {1LL, 2LL, 31LL, 33LL, 63LL, 64LL, 65LL, 100LL, 127LL, 128LL, 129LL, 130LL, 191LL, 192LL, 193LL, 200LL, 257LL, 1024LL, 1100LL}) {
        std::vector<double> h(n * n, 0.0), y(n, 1.0), L(n * n, 0.0);
        for (long long i = 0; i < n; ++i) for (long long j = 0; j <= i; ++j) h[i * n + j] = (i == j) ? 4.0 + 0.001 * i : 0.5 / (1.0 + (i - j));
        // host reference
        std::vector<double> R(h);
        for (long long j = 0; j < n; ++j) {
            double d = R[j * n + j];
            for (long long k = 0; k < j; ++k) d -= R[j * n + k] * R[j * n + k];
            d = sqrt(d); R[j * n + j] = d;
            for (long long i = j + 1; i < n; ++i) {
                double s = R[i * n + j];
                for (long long k = 0; k < j; ++k) s -= R[i * n + k] * R[j * n + k];
                R[i * n + j] = s / d;
            }
        }
        double *A, *yd, *z; int* info;
        (void)hipMalloc(&A, n * n * 8); (void)hipMalloc(&yd, n * 8); (void)hipMalloc(&z, n * 8); (void)hipMalloc(&info, 4);
        (void)hipMemcpy(yd, y.data(), n * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice);
        int rc = apgp_potrf(A, n, n, yd, 0.0, z, info, nullptr);
        (void)hipDeviceSynchronize();
        int hi; (void)hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(L.data(), A, n * n * 8, hipMemcpyDeviceToHost);
        double emax = 0; long long ei = -1, ej = -1;
        for (long long i = 0; i < n; ++i) for (long long j = 0; j <= i; ++j) { double e = fabs(L[i * n + j] - R[i * n + j]); if (!(e <= emax)) { emax = e; ei = i; ej = j; } }
        printf("n %5lld rc %d info %d max |L - Lref| %.3e at (%lld, %lld)\n", n, rc, hi, emax, ei, ej);
        (void)hipFree(A); (void)hipFree(yd); (void)hipFree(z); (void)hipFree(info);
    }
    return 0;
}
