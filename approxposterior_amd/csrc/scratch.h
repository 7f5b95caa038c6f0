// Stream-ordered device scratch kept per (device, stream, slot) and grown on demand (hipMallocAsync /
// hipFreeAsync): calls on one stream are serialised, so they can share a buffer; calls on different
// streams cannot.  Slots: 0 = Cholesky (factored diagonal blocks), 1 = blocked triangular solve
// (working right-hand side).  One definition for the whole library (inline function, static locals).
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <tuple>

inline double* apgp_stream_scratch(int slot, hipStream_t s, size_t doubles) {
    struct Scr { double* p; size_t doubles; };
    static std::mutex mu;
    static std::map<std::tuple<int, hipStream_t, int>, Scr> tab;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Scr& e = tab[std::make_tuple(dev, s, slot)];
    if (e.doubles < doubles) {
        if (e.p) (void)hipFreeAsync(e.p, s);
        e.p = nullptr; e.doubles = 0;
        const size_t want = doubles + doubles / 2;
        double* p = nullptr;
        if (hipMallocAsync((void**)&p, want * sizeof(double), s) != hipSuccess) return nullptr;
        e.p = p; e.doubles = want;
    }
    return e.p;
}
