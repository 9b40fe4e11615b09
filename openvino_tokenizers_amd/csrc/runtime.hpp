// runtime.hpp -- host-side plumbing shared by the op entry points: error reporting, device
// buffers, per-call workspaces, kernel-launch timing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ovtk_amd.h"
#include "device_common.hpp"

namespace ovtk {

int set_error(int code, const std::string& msg);
const char* last_error();

#define OVTK_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return ::ovtk::set_error(OVTK_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// Grow-only device buffer.
class DevBuf {
public:
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int ensure(size_t bytes);  // 0 or OVTK_E_HIP
    int upload(const void* host, size_t bytes, hipStream_t s = nullptr);
    void release();
    template <typename T> T* as() const { return static_cast<T*>(p_); }
    size_t size() const { return cap_; }
private:
    void* p_ = nullptr;
    size_t cap_ = 0;
};

// Kernel timing for bench.py: hipEvents on the launch stream, resolved after the call's sync.
class Profiler {
public:
    static Profiler& get();
    bool enabled() const { return on_; }
    void enable(bool on) { on_ = on; }
    void reset();
    bool lookup(const std::string& name, double* ms, int64_t* n);
    std::string dump();
    // per call
    struct Mark { const char* name; hipEvent_t a, b; };
    void begin(const char* name, hipStream_t s, std::vector<Mark>& marks);
    void end(hipStream_t s, std::vector<Mark>& marks);
    void resolve(std::vector<Mark>& marks);
private:
    bool on_ = false;
    std::mutex mu_;
    std::map<std::string, std::pair<double, int64_t>> acc_;
    std::vector<hipEvent_t> pool_;
    hipEvent_t take();
};

// The launches of one call: profiler marks, plus "kernels of this workspace may still be running on `stream`" -- set by
// every launch, cleared where the host has waited for them; a workspace that goes back to the pool while it is set (an
// error return between launches) is synchronised first, so the next call never shares buffers with running kernels.
struct LaunchLog : std::vector<Profiler::Mark> {
    hipStream_t stream = nullptr;
    bool in_flight = false;
    void settled() { in_flight = false; }
};
inline void note_launch(LaunchLog& log, hipStream_t s) { log.stream = s; log.in_flight = true; }
inline void note_launch(std::vector<Profiler::Mark>&, hipStream_t) {}

// Launch a kernel, optionally bracketed by profiler events.
#define OVTK_LAUNCH(marks, name, kernel, grid, block, stream, ...)                   \
    do {                                                                              \
        ::ovtk::note_launch(marks, stream);                                           \
        ::ovtk::Profiler& pf_ = ::ovtk::Profiler::get();                              \
        if (pf_.enabled()) pf_.begin(name, stream, marks);                            \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__);  \
        if (pf_.enabled()) pf_.end(stream, marks);                                    \
    } while (0)

// Scratch memory of one in-flight run() call.  Handles are immutable; every call checks a
// workspace out of the per-device pool, so concurrent calls never share mutable state.
struct Workspace {
    DevBuf row_stage, row_cnt, row_used, stage, deferred, exact, scratch, wave_off, tiles, status;
    DevBuf in_rb, in_re, in_begins, in_ends, in_chars, in_skips;  // staging for OVTK_MEM_HOST calls
    DevBuf out_a, out_b, out_c, out_d, out_e;
    DevBuf gen[8];  // op-specific inputs / temporaries (api_ops.cpp)
    RunStatus* host_status = nullptr;  // pinned
    // encode_small_kernel leaves the device status block zeroed: true for the call of lease number `clean_after_lease` + 1
    // of this workspace if nothing else has used it (any other op leases it, counts, and does its own memset)
    const void* clean_status = nullptr;
    size_t clean_bytes = 0;
    uint64_t lease_count = 0, clean_after_lease = ~0ull;
    // the large path's two status blocks (RowsRun::launch): compact_kernel zeroes the one its call does not use
    const void* zeroed_status = nullptr;   // the block the last large call's compact_kernel zeroed ...
    size_t zeroed_bytes = 0;               // ... that many bytes of it ...
    uint64_t zeroed_after_lease = ~0ull;   // ... valid for the lease right behind that call's
    int status_half = 0;
    hipEvent_t done = nullptr;         // end of the call in flight (RowsRun)
    LaunchLog marks;
    ~Workspace();
};

class WorkspacePool {
public:
    static WorkspacePool& get(int device);
    std::unique_ptr<Workspace> acquire();
    void release(std::unique_ptr<Workspace> w);
private:
    std::mutex mu_;
    std::vector<std::unique_ptr<Workspace>> free_;
};

struct WorkspaceLease {
    explicit WorkspaceLease(int device) : pool(WorkspacePool::get(device)), ws(pool.acquire()) {}
    ~WorkspaceLease() { pool.release(std::move(ws)); }
    WorkspacePool& pool;
    std::unique_ptr<Workspace> ws;
    Workspace* operator->() { return ws.get(); }
};

int device_cu_count(int device);

}  // namespace ovtk
