// common.hip -- error state, device selection, misc ABI entry points.
#include "common.h"

#include <mutex>
#include <vector>

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
    // The checks (device count, index, architecture) are made once per device and process; afterwards a call costs one
    // hipSetDevice -- this sits on the path of every entry point, the single-query path included.
    static std::mutex mu;
    static std::vector<char> checked;         // 1 = validated gfx950 device
    {
        std::lock_guard<std::mutex> lock(mu);
        if (device >= 0 && device < (int)checked.size() && checked[device]) {
            HIPTS_HIP(hipSetDevice(device));
            return HIPTS_OK;
        }
    }
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
    {
        std::lock_guard<std::mutex> lock(mu);
        if ((int)checked.size() < n) checked.resize(n, 0);
        checked[device] = 1;
    }
    return HIPTS_OK;
}

int current_device_cus(int* dev_out) {
    static std::mutex mu;
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev_out) *dev_out = dev;
    std::lock_guard<std::mutex> lock(mu);
    const int slot = (dev >= 0 && dev < 64) ? dev : 0;
    if (!cus[slot]) {
        hipDeviceProp_t prop;
        cus[slot] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[slot];
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

int hipts_sizeof_config(int kind, size_t* bytes) {
    if (!bytes) return HIPTS_ERR_INVALID;
    switch (kind) {
        case 0: *bytes = sizeof(hipts_vit_config_t); return HIPTS_OK;
        case 1: *bytes = sizeof(hipts_eva_config_t); return HIPTS_OK;
        case 2: *bytes = sizeof(hipts_ccip_config_t); return HIPTS_OK;
        default: return hipts::set_error(HIPTS_ERR_INVALID, "hipts_sizeof_config: kind %d (0 vit, 1 eva, 2 ccip)", kind);
    }
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
