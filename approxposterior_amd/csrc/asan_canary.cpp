// Only in the sanitizer build (make asan): two deliberate faults, so that tools/asan_cabi.py can show that the
// pairing "clang instrumentation + gcc's libasan / libubsan" really reports before it trusts a clean run.
#include <cstdint>
#include <cstdlib>

extern "C" int apgp_asan_canary(int what) {
    if (what == 1) {                       // heap-buffer-overflow: one double past a 4-double allocation
        volatile double* p = (volatile double*)std::malloc(4 * sizeof(double));
        volatile int idx = 4;
        p[idx] = 1.0;
        const int r = (int)p[0];
        std::free((void*)p);
        return r;
    }
    if (what == 2) {                       // signed overflow (UBSan)
        volatile int32_t big = INT32_MAX;
        volatile int32_t one = 1;
        return big + one;
    }
    return 0;
}
