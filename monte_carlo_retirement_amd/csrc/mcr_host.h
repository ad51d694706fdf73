// mcr_host.h — host-side utilities shared by the translation units of libmcr_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/mcr.h"

namespace mcr {

void set_error(const char* fmt, ...);          // thread-local message for mcr_last_error()
int hip_fail(hipError_t e, const char* what);  // records the HIP error text, returns MCR_ERR_HIP
int use_device(int device);                    // hipSetDevice with range / no-device checks

// Every ABI entry point works on `device` and then puts the calling thread's current device back: a
// caller that holds GPU 3 (one process per GPU under torch.distributed) keeps GPU 3 after calling a
// helper with device 0.
class DeviceScope {
public:
    explicit DeviceScope(int device) : prev_(-1), rc(MCR_OK) {
        if (hipGetDevice(&prev_) != hipSuccess) { (void)hipGetLastError(); prev_ = -1; }
        rc = use_device(device);
        if (rc != MCR_OK || prev_ == device) prev_ = -1;   // nothing to restore
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
    ~DeviceScope() {
        if (prev_ >= 0) (void)hipSetDevice(prev_);
    }

private:
    int prev_;

public:
    int rc;
};
#define MCR_ENTER_DEVICE(device)         \
    ::mcr::DeviceScope mcr_scope_(device); \
    if (mcr_scope_.rc != MCR_OK) return mcr_scope_.rc

// RAII bag of device allocations for the rarely used *_host convenience entry points (helpers, shock rows).
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

// Per (host thread, device) context of the host-buffer entry points (mcr_run_batch_host*): a private
// NON-BLOCKING stream — concurrent callers (server executor threads) do not serialise on the null stream
// or on a device-wide synchronise — and one cached scratch block that calls carve their device buffers
// from.  Blocks up to kHostCtxKeepBytes stay cached between calls; larger ones are released when the call
// ends.  Contexts live for the life of the process (destroying HIP objects from thread-exit hooks is not safe).
constexpr size_t kHostCtxKeepBytes = (size_t)256 << 20;
struct HostCtx {
    int device;
    hipStream_t stream;
    void* block;
    size_t capacity;
};
HostCtx* host_ctx(int device);                       // nullptr + error set on failure
hipError_t host_ctx_reserve(HostCtx* c, size_t bytes);  // grows the block (contents are not preserved)
void host_ctx_release_large(HostCtx* c);             // frees the block if it is above the keep limit

}  // namespace mcr
