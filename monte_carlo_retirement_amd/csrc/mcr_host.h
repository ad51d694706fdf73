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

// Context of one host-buffer call (mcr_run_batch_host*): a private NON-BLOCKING stream — concurrent callers (server
// executor threads) do not serialise on the null stream or on a device-wide synchronise — and one scratch block that the
// call carves its device buffers from.  Contexts live in a PROCESS-WIDE pool keyed by device (mutex-guarded): a call
// leases one for its duration (HostCtxLease) and hands it back, so short-lived caller threads and the per-call shard
// workers of the multi-device entry reuse the same few contexts instead of leaving one behind each (round 2 kept them in
// thread-local storage: one stream + up to 256 MiB of HBM leaked per thread that ever called).  At most
// kHostCtxIdlePerDevice idle contexts per device stay cached, each with a block of at most kHostCtxKeepBytes (larger
// blocks are freed when the call ends); mcr_release_cached() frees the idle ones on demand.  Concurrency is bounded by
// the callers: N simultaneous calls use N contexts.
constexpr size_t kHostCtxKeepBytes = (size_t)256 << 20;
constexpr int kHostCtxIdlePerDevice = 4;
struct HostCtx {
    int device;
    hipStream_t stream;
    void* block;
    size_t capacity;
};
HostCtx* host_ctx_acquire(int device);               // nullptr + error set on failure; the caller owns it until release
void host_ctx_release(HostCtx* c);                   // back to the pool (block above the keep limit freed; surplus contexts destroyed)
hipError_t host_ctx_reserve(HostCtx* c, size_t bytes);  // grows the block (contents are not preserved)
class HostCtxLease {
public:
    explicit HostCtxLease(int device) : ctx(host_ctx_acquire(device)) {}
    HostCtxLease(const HostCtxLease&) = delete;
    HostCtxLease& operator=(const HostCtxLease&) = delete;
    ~HostCtxLease() { if (ctx) host_ctx_release(ctx); }
    HostCtx* ctx;
};

// Fork/join streams for calls that spread independent pieces of work over several HIP streams and join them back onto
// the caller's stream: kForkStreams non-blocking side streams with one "done" event each, and a fork event.  Leased from
// a process-wide pool keyed by device (mcr_hip.hip) for the duration of the call's ENQUEUE; at most 2 idle sets per
// device stay cached (mcr_release_cached frees them).
constexpr int kForkStreams = 8;
struct StreamFork {
    int device;
    hipStream_t side[kForkStreams];
    hipEvent_t done[kForkStreams];
    hipEvent_t fork;
};
StreamFork* stream_fork_acquire(int device);         // nullptr on failure (out of resources)
void stream_fork_release(StreamFork* f);
class StreamForkLease {
public:
    explicit StreamForkLease(int device) : f(stream_fork_acquire(device)) {}
    StreamForkLease(const StreamForkLease&) = delete;
    StreamForkLease& operator=(const StreamForkLease&) = delete;
    ~StreamForkLease() { if (f) stream_fork_release(f); }
    StreamFork* f;
};

}  // namespace mcr
