// Probe: direction of DPP row_ror as used by apgp_mma16 (apgp_common.h).  Computes one 16x16x4
// product with the helper and reports which column mapping reproduces the host product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../approxposterior_amd/csrc/apgp_common.h"
__global__ void k(const double* A, const double* B, double* out) {
    const int lane = threadIdx.x;
    const double af = A[(lane & 15) * 4 + (lane >> 4)];   // A[row][k]
    const double bf = B[(lane >> 4) * 16 + (lane & 15)];  // B[k][col]
    double acc[4] = {0, 0, 0, 0};
    apgp_mma16(af, apgp_brot(bf), acc);
    for (int r = 0; r < 4; ++r) out[r * 64 + lane] = acc[r];
}
int main() {
    double hA[64], hB[64], hC[256], ho[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + 0.37 * i; hB[i] = 2 - 0.11 * i * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int q = 0; q < 4; ++q) s += hA[i * 4 + q] * hB[q * 16 + j]; hC[i * 16 + j] = s; }
    double *dA, *dB, *dO;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dO, 2048);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dO);
    hipMemcpy(ho, dO, 2048, hipMemcpyDeviceToHost);
    for (int sgn = -1; sgn <= 1; sgn += 2) {
        int bad = 0;
        for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) {
            const int b = (l >> 2) & 3, row = 4 * b + (l >> 4), col = 4 * ((b + sgn * r + 4) & 3) + (l & 3);
            if (fabs(ho[r * 64 + l] - hC[row * 16 + col]) > 1e-9 * fabs(hC[row * 16 + col]) + 1e-12) ++bad;
        }
        printf("column group (b %c r) & 3 : %d mismatches\n", sgn > 0 ? '+' : '-', bad);
    }
    return 0;
}
