// runtime.cpp -- see runtime.hpp.
#include "runtime.hpp"

#include <cstdio>
#include <cstring>

namespace ovtk {

namespace {
thread_local std::string g_err;
}

int set_error(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
const char* last_error() { return g_err.c_str(); }

int DevBuf::ensure(size_t bytes) {
    if (bytes <= cap_ && p_) return OVTK_OK;
    release();
    // Headroom of 1/8 (large buffers): consecutive batches of a serving loop differ by a fraction of a percent in size,
    // and every regrowth is a hipFree + hipMalloc -- two device-wide synchronisations in the middle of the pipeline.
    size_t want = bytes < 256 ? 256 : bytes;
    if (want >= (size_t(1) << 16)) want += want / 8;
    OVTK_HIP(hipMalloc(&p_, want));
    cap_ = want;
    return OVTK_OK;
}
int DevBuf::upload(const void* host, size_t bytes, hipStream_t s) {
    if (int rc = ensure(bytes)) return rc;
    if (bytes) OVTK_HIP(hipMemcpyAsync(p_, host, bytes, hipMemcpyHostToDevice, s));
    return OVTK_OK;
}
void DevBuf::release() {
    if (p_) (void)hipFree(p_);
    p_ = nullptr;
    cap_ = 0;
}

Profiler& Profiler::get() {
    static Profiler p;
    return p;
}
void Profiler::reset() {
    std::lock_guard<std::mutex> lk(mu_);
    acc_.clear();
}
bool Profiler::lookup(const std::string& name, double* ms, int64_t* n) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = acc_.find(name);
    if (it == acc_.end()) return false;
    *ms = it->second.first;
    *n = it->second.second;
    return true;
}
std::string Profiler::dump() {
    std::lock_guard<std::mutex> lk(mu_);
    std::string out;
    char line[256];
    for (const auto& kv : acc_) {
        std::snprintf(line, sizeof line, "%s %.6f %lld\n", kv.first.c_str(), kv.second.first, (long long)kv.second.second);
        out += line;
    }
    return out;
}
hipEvent_t Profiler::take() {
    std::lock_guard<std::mutex> lk(mu_);
    if (!pool_.empty()) {
        hipEvent_t e = pool_.back();
        pool_.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void Profiler::begin(const char* name, hipStream_t s, std::vector<Mark>& marks) {
    Mark m{name, take(), take()};
    (void)hipEventRecord(m.a, s);
    marks.push_back(m);
}
void Profiler::end(hipStream_t s, std::vector<Mark>& marks) { (void)hipEventRecord(marks.back().b, s); }
void Profiler::resolve(std::vector<Mark>& marks) {
    for (auto& m : marks) {
        float ms = 0.f;
        (void)hipEventSynchronize(m.b);
        (void)hipEventElapsedTime(&ms, m.a, m.b);
        std::lock_guard<std::mutex> lk(mu_);
        auto& a = acc_[m.name];
        a.first += ms;
        a.second += 1;
        pool_.push_back(m.a);
        pool_.push_back(m.b);
    }
    marks.clear();
}

Workspace::~Workspace() {
    if (done) (void)hipEventDestroy(done);
    if (host_status) (void)hipHostFree(host_status);
}

WorkspacePool& WorkspacePool::get(int device) {
    static std::mutex mu;
    static std::map<int, std::unique_ptr<WorkspacePool>> pools;
    std::lock_guard<std::mutex> lk(mu);
    auto& p = pools[device];
    if (!p) p = std::make_unique<WorkspacePool>();
    return *p;
}
std::unique_ptr<Workspace> WorkspacePool::acquire() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (!free_.empty()) {
            auto w = std::move(free_.back());
            free_.pop_back();
            ++w->lease_count;
            return w;
        }
    }
    auto w = std::make_unique<Workspace>();
    void* p = nullptr;
    if (hipHostMalloc(&p, sizeof(RunStatus), hipHostMallocDefault) == hipSuccess) w->host_status = static_cast<RunStatus*>(p);
    return w;
}
void WorkspacePool::release(std::unique_ptr<Workspace> w) {
    if (!w) return;
    if (w->marks.in_flight) {  // an error return between launches: the kernels already enqueued still use these buffers
        (void)hipStreamSynchronize(w->marks.stream);
        w->marks.settled();
    }
    std::lock_guard<std::mutex> lk(mu_);
    free_.push_back(std::move(w));
}

int device_cu_count(int device) {
    static std::mutex mu;
    static std::map<int, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(device);
    if (it != cache.end()) return it->second;
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    cache[device] = n;
    return n;
}

}  // namespace ovtk
