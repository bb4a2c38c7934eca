// common.h -- shared host-side plumbing of libhip_tagsearch.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/hip_tagsearch.h"

namespace hipts {

std::string& last_error_ref();
int set_error(int status, const char* fmt, ...);

// Makes `device` current; fails with HIPTS_ERR_NO_DEVICE when there is no such gfx950 device.
int use_device(int device);

#define HIPTS_HIP(expr)                                                                        \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::hipts::set_error(HIPTS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                                      hipGetErrorString(_e), __FILE__, __LINE__);              \
    } while (0)

#define HIPTS_TRY(expr)                                                                        \
    do {                                                                                       \
        int _s = (expr);                                                                       \
        if (_s != HIPTS_OK) return _s;                                                         \
    } while (0)

#define HIPTS_REQUIRE(cond, ...)                                                               \
    do {                                                                                       \
        if (!(cond)) return ::hipts::set_error(HIPTS_ERR_INVALID, __VA_ARGS__);                \
    } while (0)

#define HIPTS_LAUNCH_CHECK()                                                                   \
    do {                                                                                       \
        hipError_t _e = hipGetLastError();                                                     \
        if (_e != hipSuccess)                                                                  \
            return ::hipts::set_error(HIPTS_ERR_HIP, "kernel launch failed: %s (%s:%d)",       \
                                      hipGetErrorString(_e), __FILE__, __LINE__);              \
    } while (0)

// Device allocation owned by a handle.
// Pinned host staging memory (grow-only): async copies to / from it neither block the caller nor need a bounce buffer.
struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
    int reserve(size_t n) {
        if (n <= bytes) return HIPTS_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        n = (n + 4095) / 4096 * 4096;
        // coherent (fine-grained) on purpose: hipts_search's one-query path lets its last kernel store results and a completion
        // sequence number here while the host spins on it -- with a non-coherent allocation (HIP_HOST_COHERENT=0 makes the default
        // one) the stores would become visible only when the kernel ends and every query would burn its whole spin budget
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocCoherent | hipHostMallocMapped);
        if (e != hipSuccess) {
            p = nullptr;
            return set_error(HIPTS_ERR_OOM, "hipHostMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return HIPTS_OK;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
        o.p = nullptr;
        o.bytes = 0;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            return set_error(HIPTS_ERR_OOM, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return HIPTS_OK;
    }
    // grow-only
    int reserve(size_t n) { return n <= bytes ? HIPTS_OK : alloc(n); }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

inline int upload(void* dst, const void* src, size_t bytes, hipStream_t s = nullptr) {
    if (bytes == 0) return HIPTS_OK;
    HIPTS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    HIPTS_HIP(hipStreamSynchronize(s));
    return HIPTS_OK;
}

// Once-per-device initialisation (function attributes are a property of the device's copy of a kernel): usage
//     static PerDevice once;  std::lock_guard<std::mutex> lk(once.mu);  if (!once.done(dev)) { ...; once.mark(dev); }
struct PerDevice {
    std::mutex mu;
    uint64_t mask = 0;
    bool done(int dev) const { return dev >= 0 && dev < 64 && ((mask >> dev) & 1); }
    void mark(int dev) { if (dev >= 0 && dev < 64) mask |= (uint64_t)1 << dev; }
};
// Compute units of the CURRENT device (cached per device); 256 if the runtime does not say.
int current_device_cus(int* dev_out = nullptr);

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace hipts
