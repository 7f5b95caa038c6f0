// Stream-ordered device scratch kept per (device, stream, slot) and grown on demand (hipMallocAsync /
// hipFreeAsync): calls on one stream are serialised, so they can share a buffer; calls on different
// streams cannot.  Slots: 0 = Cholesky (factored diagonal blocks), 1 = blocked triangular solve
// (working right-hand side), 2 = persistent Cholesky (flags, granule streams), 3 = persistent triangular solve (ticket
// counter, error word, granules), 4 = multi-workgroup ensemble sampler (log-probability granules).  One definition for the whole
// library (inline functions, static locals).
//   * The device is the STREAM's (hipStreamGetDevice), not the caller's current device.
//   * Entries live until apgp_release_scratch(stream) (include/apgp.h) or process exit: call it before
//     destroying a stream -- a recycled stream handle would otherwise inherit the dead stream's entry.
//   * Growing allocates (hipMallocAsync): the entry points that use scratch -- apgp_potrf / apgp_nll_eval*
//     for n > 64, apgp_trsv for n >= 256 -- must not be called during stream capture.
//   * apgp_stream_lock(stream): one mutex per (device, stream).  The launches of one factorisation /
//     blocked solve share the stream's scratch and are enqueued as a unit under it; calls on different
//     streams or devices do not serialise each other (round 2 had one process-wide mutex).
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>

// 64 bytes of pinned, device-mapped host memory per (device, stream): a kernel's last lane writes a small
// result record and a sequence word straight into it (apgp_nll_eval at n <= 64: no D2H copy, no stream
// synchronisation -- the host polls the word)
struct ApgpMailbox {
    volatile double* host = nullptr; double* dev = nullptr; long long seq = 0;
    // a larger pinned, device-mapped staging area (grown on demand) for calls that hand over host buffers
    // (apgp_predict_mean_host): the kernel reads / writes it in place -- no H2D / D2H copies
    double* io_host = nullptr; double* io_dev = nullptr; size_t io_doubles = 0;
};

struct ApgpScratchTable {
    struct Scr { double* p = nullptr; size_t doubles = 0; unsigned long long calls = 0; };
    std::mutex mu;
    std::map<std::tuple<int, hipStream_t, int>, Scr> tab;
    std::map<std::pair<int, hipStream_t>, ApgpMailbox> mail;
    std::map<std::pair<int, hipStream_t>, std::unique_ptr<std::mutex>> locks;
};
inline ApgpScratchTable& apgp_scratch_table() {
    static ApgpScratchTable t;
    return t;
}
inline int apgp_stream_device(hipStream_t s) {
    int dev = 0;
    if (s != nullptr && hipStreamGetDevice(s, &dev) == hipSuccess) return dev;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}

// *fresh (optional): the buffer was (re)allocated by this call -- its contents are undefined.
// *calls (optional): a counter that lives with the entry (the persistent Cholesky's call-unique flag values).
inline double* apgp_stream_scratch_ex(int slot, hipStream_t s, size_t doubles, bool* fresh, unsigned long long** calls) {
    ApgpScratchTable& t = apgp_scratch_table();
    const int dev = apgp_stream_device(s);
    if (dev < 0) return nullptr;
    std::lock_guard<std::mutex> lock(t.mu);
    ApgpScratchTable::Scr& e = t.tab[std::make_tuple(dev, s, slot)];
    if (fresh) *fresh = false;
    if (e.doubles < doubles) {
        if (e.p) (void)hipFreeAsync(e.p, s);
        e.p = nullptr; e.doubles = 0;
        const size_t want = doubles + doubles / 2;
        double* p = nullptr;
        if (hipMallocAsync((void**)&p, want * sizeof(double), s) != hipSuccess) return nullptr;
        e.p = p; e.doubles = want;
        if (fresh) *fresh = true;
    }
    if (calls) *calls = &e.calls;     // (std::map nodes are stable; the caller holds apgp_stream_lock(s))
    return e.p;
}
inline double* apgp_stream_scratch(int slot, hipStream_t s, size_t doubles) {
    return apgp_stream_scratch_ex(slot, s, doubles, nullptr, nullptr);
}

inline std::mutex& apgp_stream_lock(hipStream_t s) {
    ApgpScratchTable& t = apgp_scratch_table();
    const int dev = apgp_stream_device(s);
    std::lock_guard<std::mutex> lock(t.mu);
    std::unique_ptr<std::mutex>& m = t.locks[std::make_pair(dev, s)];
    if (!m) m.reset(new std::mutex());
    return *m;             // (entries are never erased: the reference stays valid)
}

// the stream's mailbox (allocated on first use; NULL host pointer if pinned memory is unavailable)
inline ApgpMailbox* apgp_stream_mailbox(hipStream_t s) {
    ApgpScratchTable& t = apgp_scratch_table();
    const int dev = apgp_stream_device(s);
    std::lock_guard<std::mutex> lock(t.mu);
    ApgpMailbox& m = t.mail[std::make_pair(dev, s)];
    if (!m.host) {
        void* h = nullptr;
        void* d = nullptr;
        // (Portable: the mailbox is keyed by the STREAM's device, which need not be the caller's current one)
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) return &m;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return &m; }
        for (int i = 0; i < 8; ++i) ((volatile double*)h)[i] = 0.0;
        m.host = (volatile double*)h;
        m.dev = (double*)d;
        m.seq = 0;
    }
    return &m;             // (std::map nodes are stable: the pointer stays valid)
}

// pinned staging area of at least `doubles` doubles for the stream (host pointer; *dev = device alias); NULL if
// pinned memory is unavailable.  Caller holds apgp_stream_lock(s) and has no kernel in flight that uses it.
inline double* apgp_stream_pinned_io(hipStream_t s, size_t doubles, double** dev) {
    ApgpMailbox* m = apgp_stream_mailbox(s);
    if (!m || !m->host) return nullptr;
    if (m->io_doubles < doubles) {
        if (m->io_host) { (void)hipStreamSynchronize(s); (void)hipHostFree(m->io_host); }
        m->io_host = nullptr; m->io_dev = nullptr; m->io_doubles = 0;
        const size_t want = doubles + doubles / 2 + 512;
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, want * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) return nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return nullptr; }
        m->io_host = (double*)h; m->io_dev = (double*)d; m->io_doubles = want;
    }
    *dev = m->io_dev;
    return m->io_host;
}

// frees (stream-ordered) every scratch buffer of `s`; returns the number of buffers released.  Must not run
// concurrently with library calls on `s` (a caller polling the stream's mailbox would lose it under its feet):
// synchronise the stream and stop using it first, as before destroying it.
inline int apgp_stream_scratch_release(hipStream_t s) {
    ApgpScratchTable& t = apgp_scratch_table();
    const int dev = apgp_stream_device(s);
    std::lock_guard<std::mutex> lock(t.mu);
    int n = 0;
    {
        auto mit = t.mail.find(std::make_pair(dev, s));
        if (mit != t.mail.end()) {
            if (mit->second.host) { (void)hipHostFree((void*)mit->second.host); ++n; }
            if (mit->second.io_host) { (void)hipHostFree(mit->second.io_host); ++n; }
            t.mail.erase(mit);
        }
    }
    for (auto it = t.tab.begin(); it != t.tab.end();) {
        if (std::get<0>(it->first) == dev && std::get<1>(it->first) == s) {
            if (it->second.p) { (void)hipFreeAsync(it->second.p, s); ++n; }
            it = t.tab.erase(it);
        } else {
            ++it;
        }
    }
    return n;
}
