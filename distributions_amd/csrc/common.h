// Shared host-side plumbing of libdistributions_hip: error reporting, device
// buffers, one-time table upload.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/distributions_hip.h"
#include "special.h"

namespace dist {

struct Error : std::runtime_error {
    explicit Error(const std::string & what) : std::runtime_error(what) {}
};

#define DIST_REQUIRE(cond, msg)                                              \
    do {                                                                     \
        if (!(cond)) throw ::dist::Error(std::string("ERROR ") + (msg));     \
    } while (0)

#define HIP_CHECK(expr)                                                      \
    do {                                                                     \
        hipError_t err_ = (expr);                                            \
        if (err_ != hipSuccess)                                              \
            throw ::dist::Error(std::string("HIP error: ") +                 \
                                hipGetErrorString(err_) + " at " #expr);     \
    } while (0)

void set_last_error(const std::string & what);

// Runs `body`, converting exceptions into the C ABI's status + message.
template <class F>
int guarded(F && body) {
    try {
        body();
        return 0;
    } catch (const std::exception & e) {
        set_last_error(e.what());
        return 1;
    } catch (...) {
        set_last_error("unknown error");
        return 1;
    }
}

// Uploads the special-function tables to the current device (once per device)
// and fails loudly when there is no usable GPU: there is no CPU fallback.
void ensure_device_ready();
hipStream_t stream();

inline size_t grow_capacity(size_t need) {
    size_t c = 64;
    while (c < need) c *= 2;
    return c;
}

// hipMemcpyAsync from PAGEABLE host memory returns when the copy has been
// done, and the copy waits for whatever the stream still holds: a 100-byte
// upload behind a 260 us kernel parks the host for 260 us, and the launches it
// would have queued meanwhile start late (the host-normalised paths upload a
// handful of small tables per batch: 29 + 6 + 4 + 16 us of idle device per C3
// sub-sweep, profiles/r5_timeline_c3.txt).  Small uploads therefore go through
// a ring of pinned memory: the bytes are copied into the next free stretch
// and the asynchronous copy reads from there.  A stretch is written again one
// lap later; each half of the ring carries an event recorded when the writer
// left it, waited for before the writer enters it again (long complete by
// then), so no copy ever reads bytes that were overwritten.
const void * staged_for_upload(const void * host, size_t bytes);

template <class T>
struct DeviceBuf {
    T * p = nullptr;
    size_t cap = 0;
    DeviceBuf() = default;
    DeviceBuf(const DeviceBuf &) = delete;
    DeviceBuf & operator=(const DeviceBuf &) = delete;
    DeviceBuf(DeviceBuf && o) noexcept : p(o.p), cap(o.cap) {
        o.p = nullptr;
        o.cap = 0;
    }
    ~DeviceBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // grows to at least n elements, keeping the first `keep` elements
    void reserve(size_t n, size_t keep) {
        if (n <= cap) return;
        static const bool trace_alloc = getenv("DIST_TRACE_ALLOC");
        if (trace_alloc)
            fprintf(stderr, "[dist] device buffer grows %zu -> %zu elements of %zu B\n",
                    cap, n, sizeof(T));
        T * q = nullptr;
        HIP_CHECK(hipMalloc(&q, n * sizeof(T)));
        HIP_CHECK(hipMemsetAsync(q, 0, n * sizeof(T), stream()));
        if (keep > cap) keep = cap;   // no more than the old buffer holds
        if (p && keep)
            HIP_CHECK(hipMemcpyAsync(q, p, keep * sizeof(T),
                                     hipMemcpyDeviceToDevice, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        if (p) (void)hipFree(p);
        p = q;
        cap = n;
    }
    void upload(const T * host, size_t n) {
        if (n > cap) reserve(grow_capacity(n), 0);   // headroom: no realloc per call
        if (n)
            HIP_CHECK(hipMemcpyAsync(p, staged_for_upload(host, n * sizeof(T)),
                                     n * sizeof(T), hipMemcpyHostToDevice,
                                     stream()));
    }
    void download(T * host, size_t n) const {
        if (n)
            HIP_CHECK(hipMemcpyAsync(host, p, n * sizeof(T),
                                     hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
    }
};

}  // namespace dist
