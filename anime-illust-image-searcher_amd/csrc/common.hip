// common.hip -- error state, device selection, misc ABI entry points.
#include "common.h"

namespace hipts {

std::string& last_error_ref() {
    static thread_local std::string err;
    return err;
}

int set_error(int status, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return status;
}

int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return set_error(HIPTS_ERR_NO_DEVICE,
                         "no HIP device available (%s): libhip_tagsearch has no CPU fallback",
                         e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n)
        return set_error(HIPTS_ERR_NO_DEVICE, "device %d out of range (have %d)", device, n);
    hipDeviceProp_t prop;
    HIPTS_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(HIPTS_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                         prop.gcnArchName);
    HIPTS_HIP(hipSetDevice(device));
    return HIPTS_OK;
}

}  // namespace hipts

extern "C" {

int hipts_abi_version(void) { return HIPTS_ABI_VERSION; }

int hipts_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return HIPTS_ERR_INVALID;
    const std::string& e = hipts::last_error_ref();
    size_t m = e.size() < n - 1 ? e.size() : n - 1;
    memcpy(buf, e.data(), m);
    buf[m] = 0;
    return HIPTS_OK;
}

int hipts_device_count(int* count) {
    if (!count) return HIPTS_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return HIPTS_OK;
}

}  // extern "C"
