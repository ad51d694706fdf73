// mcr_host.h — host-side utilities shared by the translation units of libmcr_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

namespace mcr {

void set_error(const char* fmt, ...);          // thread-local message for mcr_last_error()
int hip_fail(hipError_t e, const char* what);  // records the HIP error text, returns MCR_ERR_HIP
int use_device(int device);                    // hipSetDevice with range / no-device checks

// RAII bag of device allocations for the *_host convenience entry points.
class DeviceArena {
public:
    DeviceArena() = default;
    DeviceArena(const DeviceArena&) = delete;
    DeviceArena& operator=(const DeviceArena&) = delete;
    ~DeviceArena() {
        for (void* p : ptrs_) (void)hipFree(p);
    }
    hipError_t alloc(void** out, size_t bytes) {
        hipError_t e = hipMalloc(out, bytes ? bytes : 1);
        if (e == hipSuccess) ptrs_.push_back(*out);
        return e;
    }

private:
    std::vector<void*> ptrs_;
};

}  // namespace mcr
