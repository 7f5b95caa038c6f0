// Probe: is v_mfma_f64_4x4x4_4b_f64 a chain of four IEEE FMAs in ascending k -- d = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c)))) --
// bit for bit?  (If so, a rank-4 update done on the matrix cores carries the bits of the VALU chain
// row[j] = fma(-x[k], l[j][k], row[j]), k = 0..3, of panel_trailing in potrf.hip.)  Random operands with heavy
// cancellation, 4096 trials x 64 lanes; also tries the descending chain, c added last, and an unfused sum.
// Operand layout of the 4-block instruction (mma16.h): A lane = row + 16 k within block ... here every lane is
// checked through the known output mapping: block b = (lane >> 2) & 3?  -- not needed: we brute-force by giving every
// block the same 4 x 4 operands, so out(lane) depends on (i, j) = the lane's position inside its block only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
__global__ void k(const double* A, const double* B, const double* C, double* out, int trials) {
    const int lane = threadIdx.x;
    for (int t = 0; t < trials; ++t) {
        // A: 4 x 4 (i, k); B: 4 x 4 (k, j); per block identical.  Fragment convention of the 16x16x4-shaped use in mma16.h:
        // A lane l holds A[row = l & 15 ... ] -- for the 4-block form: lane l -> block (l >> 2) & 3 ?  We avoid relying on it:
        // every lane loads A[(l & 3)][l >> 4] and B[l >> 4][(l & 3)] (i or j = l & 3, k = l >> 4), identical for the 4 blocks.
        const double af = A[t * 16 + (lane & 3) * 4 + (lane >> 4)];
        const double bf = B[t * 16 + (lane >> 4) * 4 + (lane & 3)];
        const double cf = C[t * 64 + lane];
        out[t * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bf, cf, 0, 0, 0);
    }
}
int main() {
    const int T = 4096;
    double *hA = new double[T * 16], *hB = new double[T * 16], *hC = new double[T * 64], *ho = new double[T * 64];
    srand(7);
    auto rnd = [] { return (rand() / (double)RAND_MAX - 0.5) * exp2((rand() % 9) - 4); };
    for (int i = 0; i < T * 16; ++i) { hA[i] = rnd(); hB[i] = rnd(); }
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < 64; ++l) {
            // c close to minus the product sum for half the trials: cancellation exposes the rounding order
            hC[t * 64 + l] = (t & 1) ? rnd() : 0.0;
        }
    double *dA, *dB, *dC, *dO;
    hipMalloc(&dA, T * 16 * 8); hipMalloc(&dB, T * 16 * 8); hipMalloc(&dC, T * 64 * 8); hipMalloc(&dO, T * 64 * 8);
    // first pass with c = 0 / random to learn the lane -> (i, j) mapping from an unfused host product
    hipMemcpy(dA, hA, T * 16 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, hB, T * 16 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, T * 64 * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dO, T);
    hipMemcpy(ho, dO, T * 64 * 8, hipMemcpyDeviceToHost);
    // mapping: for lane l find (i, j) minimising the error on trial 0
    int mi[64], mj[64];
    for (int l = 0; l < 64; ++l) {
        double best = 1e300;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double e = 0;
            for (int t = 0; t < 8; ++t) {
                double s = hC[t * 64 + l];
                for (int q = 0; q < 4; ++q) s += hA[t * 16 + i * 4 + q] * hB[t * 16 + q * 4 + j];
                e += fabs(s - ho[t * 64 + l]);
            }
            if (e < best) { best = e; mi[l] = i; mj[l] = j; }
        }
    }
    printf("lane -> (i, j): "); for (int l = 0; l < 64; l += 5) printf("%d:(%d,%d) ", l, mi[l], mj[l]); printf("\n");
    // second pass: c = -(rounded product sum) * (1 + tiny) for cancellation
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < 64; ++l) {
            double s = 0;
            for (int q = 0; q < 4; ++q) s += hA[t * 16 + mi[l] * 4 + q] * hB[t * 16 + q * 4 + mj[l]];
            if (!(t & 1)) hC[t * 64 + l] = -s * (1.0 + ((t >> 1) % 5) * 1e-13);
        }
    hipMemcpy(dC, hC, T * 64 * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dO, T);
    hipMemcpy(ho, dO, T * 64 * 8, hipMemcpyDeviceToHost);
    long bad_asc = 0, bad_desc = 0, bad_clast = 0, bad_unfused = 0, bad_pair = 0, n = 0;
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < 64; ++l) {
            const double* a = hA + t * 16 + mi[l] * 4;
            double b[4]; for (int q = 0; q < 4; ++q) b[q] = hB[t * 16 + q * 4 + mj[l]];
            const double c = hC[t * 64 + l], got = ho[t * 64 + l];
            double asc = c; for (int q = 0; q < 4; ++q) asc = fma(a[q], b[q], asc);
            double desc = c; for (int q = 3; q >= 0; --q) desc = fma(a[q], b[q], desc);
            double cl = 0; for (int q = 0; q < 4; ++q) cl = fma(a[q], b[q], cl); cl += c;
            double un = c; for (int q = 0; q < 4; ++q) un += a[q] * b[q];
            double pr = fma(a[1], b[1], fma(a[0], b[0], c)) ; double pr2 = fma(a[3], b[3], a[2] * b[2]); pr += pr2;
            ++n;
            if (memcmp(&asc, &got, 8)) ++bad_asc;
            if (memcmp(&desc, &got, 8)) ++bad_desc;
            if (memcmp(&cl, &got, 8)) ++bad_clast;
            if (memcmp(&un, &got, 8)) ++bad_unfused;
            if (memcmp(&pr, &got, 8)) ++bad_pair;
        }
    printf("trials x lanes = %ld\nmismatches: ascending FMA chain %ld | descending %ld | c last %ld | unfused %ld | pairwise %ld\n",
           n, bad_asc, bad_desc, bad_clast, bad_unfused, bad_pair);
    return 0;
}
