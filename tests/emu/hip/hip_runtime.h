// tests/emu/hip/hip_runtime.h -- a tiny SIMT emulator standing in for <hip/hip_runtime.h>.
//
// TEST INFRASTRUCTURE ONLY.  The product's kernel sources are compiled unchanged with g++ and
// `-I tests/emu` so that this file shadows the real HIP header; the result (libovtk_emu.so) lets
// the CPU-only test tier (-m "not gpu") exercise the *same* kernel logic that runs on gfx950,
// and lets ASan/UBSan see it (GPU sanitizers are not available on this pool).  It is never
// loaded by the product package: openvino_tokenizers_amd loads libovtk_amd.so (HIP) or fails.
//
// Model: one block at a time; every thread of the block is a ucontext fiber; wave collectives
// (__ballot, __shfl*, wave barrier) and __syncthreads() are rendezvous points.  A collective that
// not every lane of the wave reaches deadlocks -> the scheduler aborts with a message, which is
// exactly the discipline the HIP kernels need (collectives only in wave-uniform control flow).
// Lanes are NOT run in lock-step between rendezvous points, so code that relies on implicit
// wave-synchronous LDS hand-offs without ovtk::wave_sync() fails here (as it may on hardware
// once the compiler reorders it).
#pragma once

#include <ucontext.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <vector>

#define OVTK_SIMT_EMULATOR 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int4 { int x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
};
#define hipHostMallocDefault 0

// Marks the emulator build for the Python side: "device" pointers are host pointers here, so CPU tensors may be handed
// to the device-only entry points (the HIP build does not export this symbol).
extern "C" __attribute__((weak, visibility("default"))) int ovtk_emulator_build() { return 1; }

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK = 256 * 1024;

struct Rendezvous {
    int arrived = 0;
    uint64_t gen = 0;
    uint64_t vals[2][1024];
    uint64_t result[2];
};

struct Fiber {
    ucontext_t ctx;
    uint3_emu tid;
    bool done = false;
    char* stack = nullptr;
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<Rendezvous> wave_rdv;
    Rendezvous block_rdv;
    uint3_emu bid, bdim, gdim;
    ucontext_t sched;
    int cur = -1;
    uint64_t progress = 0;
    std::function<void()> body;
};

inline Block*& g_block() {
    static Block* b = nullptr;
    return b;
}
inline Block& blk() { return *g_block(); }
inline Fiber& cur_fiber() { return blk().fibers[blk().cur]; }
inline void yield() { swapcontext(&cur_fiber().ctx, &blk().sched); }

inline const uint3_emu& thread_idx() { return cur_fiber().tid; }
inline const uint3_emu& block_idx() { return blk().bid; }
inline const uint3_emu& block_dim() { return blk().bdim; }
inline const uint3_emu& grid_dim() { return blk().gdim; }

// Deposit `v`, wait for all n participants, return the generation parity that holds the values.
inline int rendezvous(Rendezvous& r, int idx, int n, uint64_t v) {
    const int par = int(r.gen & 1);
    r.vals[par][idx] = v;
    ++blk().progress;
    if (++r.arrived == n) {
        r.arrived = 0;
        ++r.gen;
    } else {
        const uint64_t g = r.gen;
        while (r.gen == g) yield();
    }
    return par;
}
inline int lane() { return int(thread_idx().x) & (WAVE - 1); }
inline Rendezvous& my_wave() { return blk().wave_rdv[thread_idx().x / WAVE]; }

inline void wave_barrier() { rendezvous(my_wave(), lane(), WAVE, 0); }
// s_barrier waits on the SURVIVING waves only (a wave that has ended does not hold the others: GCN3 ISA, S_BARRIER): the
// rendezvous counts the fibers that have not finished, and a fiber that finishes while others wait may be the one they waited for.
inline int live_fibers() {
    int n = 0;
    for (const Fiber& f : blk().fibers) n += f.done ? 0 : 1;
    return n;
}
inline void block_barrier() { rendezvous(blk().block_rdv, int(thread_idx().x), live_fibers(), 0); }
inline unsigned long long ballot(int pred) {
    Rendezvous& r = my_wave();
    const int par = rendezvous(r, lane(), WAVE, pred ? 1 : 0);
    unsigned long long m = 0;
    for (int i = 0; i < WAVE; ++i) m |= (unsigned long long)(r.vals[par][i] & 1) << i;
    return m;
}
template <typename T>
inline T shfl(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl of <= 64-bit types");
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    Rendezvous& r = my_wave();
    const int par = rendezvous(r, lane(), WAVE, raw);
    uint64_t got = r.vals[par][src & (WAVE - 1)];
    T out;
    std::memcpy(&out, &got, sizeof(T));
    return out;
}

inline void fiber_entry() {
    blk().body();
    cur_fiber().done = true;
    ++blk().progress;
    {   // the fibers waiting at a block barrier were waiting for the survivors: this one is no longer among them
        Rendezvous& r = blk().block_rdv;
        if (r.arrived > 0 && r.arrived == live_fibers()) {
            r.arrived = 0;
            ++r.gen;
        }
    }
    swapcontext(&cur_fiber().ctx, &blk().sched);
}

inline std::vector<char*>& stack_pool() {
    static std::vector<char*> s;
    return s;
}

template <typename K, typename... A>
void launch(K kernel, dim3 grid, dim3 block, A... args) {
    if (block.x % WAVE != 0 || block.y != 1 || block.z != 1) {
        std::fprintf(stderr, "emu: block must be 1-D and a multiple of 64 threads\n");
        std::abort();
    }
    std::vector<char*>& stacks = stack_pool();
    while (stacks.size() < block.x) stacks.push_back(static_cast<char*>(std::malloc(STACK)));
    Block b;
    g_block() = &b;
    b.bdim = {block.x, 1, 1};
    b.gdim = {grid.x, grid.y, grid.z};
    b.body = [&]() { kernel(args...); };
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                b.bid = {bx, by, bz};
                b.fibers.assign(block.x, Fiber{});
                b.wave_rdv.assign(block.x / WAVE, Rendezvous{});
                b.block_rdv = Rendezvous{};
                for (unsigned t = 0; t < block.x; ++t) {
                    Fiber& f = b.fibers[t];
                    f.tid = {t, 0, 0};
                    f.stack = stacks[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
                }
                unsigned live = block.x;
                while (live) {
                    const uint64_t before = b.progress;
                    live = 0;
                    for (unsigned t = 0; t < block.x; ++t) {
                        if (b.fibers[t].done) continue;
                        b.cur = int(t);
                        swapcontext(&b.sched, &b.fibers[t].ctx);
                        if (!b.fibers[t].done) ++live;
                    }
                    if (live && b.progress == before) {
                        std::fprintf(stderr, "emu: deadlock -- a collective was not reached by every lane "
                                             "(block %u, %u fibers stuck; the first: thread", bx, live);
                        for (unsigned t = 0, shown = 0; t < block.x && shown < 4; ++t)
                            if (!b.fibers[t].done) {
                                std::fprintf(stderr, " %u", t);
                                ++shown;
                            }
                        std::fprintf(stderr, ")\n  in %s\n", __PRETTY_FUNCTION__);
                        std::abort();
                    }
                }
            }
    g_block() = nullptr;
}

}  // namespace emu

#define threadIdx (::emu::thread_idx())
#define blockIdx (::emu::block_idx())
#define blockDim (::emu::block_dim())
#define gridDim (::emu::grid_dim())

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::emu::launch(kernel, dim3(grid), dim3(block), ##__VA_ARGS__)

// ---- device-side functions used by the kernels
static inline void __syncthreads() { ::emu::block_barrier(); }
static inline unsigned long long __ballot(int pred) { return ::emu::ballot(pred); }
static inline int __any(int pred) { return ::emu::ballot(pred) != 0; }
static inline int __all(int pred) { return ::emu::ballot(pred) == ~0ull; }
template <typename T> static inline T __shfl(T v, int src, int = 64) { return ::emu::shfl(v, src); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int = 64) {
    int l = ::emu::lane();
    T got = ::emu::shfl(v, l - int(d) < 0 ? l : l - int(d));
    return got;
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int = 64) {
    int l = ::emu::lane();
    return ::emu::shfl(v, l + int(d) > 63 ? l : l + int(d));
}
template <typename T> static inline T __shfl_xor(T v, int m, int = 64) { return ::emu::shfl(v, ::emu::lane() ^ m); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline unsigned long long __brevll(unsigned long long x) {
    unsigned long long r = 0;
    for (int i = 0; i < 64; ++i) r |= ((x >> i) & 1ull) << (63 - i);
    return r;
}
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline unsigned __lane_id() { return unsigned(::emu::lane()); }
#define __builtin_amdgcn_wave_barrier() ::emu::wave_barrier()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
template <typename T> static inline T emu_readfirstlane(T v) { return ::emu::shfl(v, 0); }
#define __builtin_amdgcn_readlane(v, src) ::emu::shfl(v, src)
// DPP (gfx9 semantics) for the controls the kernels use: row_shl:n 0x100+n, row_shr:n 0x110+n, row_bcast15 0x142,
// row_bcast31 0x143, wave_shl:1 0x130, wave_shr:1 0x138.  A lane whose row is not in row_mask / whose bank is not in bank_mask keeps `old`; a lane with
// no valid source gets 0 when bound_ctrl is set, `old` otherwise.  Collective: every lane of the wave must call it.
static inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int l = ::emu::lane(), row = l >> 4, in_row = l & 15;
    int from = -1;
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl - 0x100; if (in_row + n <= 15) from = l + n; }
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; if (in_row - n >= 0) from = l - n; }
    else if (ctrl == 0x130) { if (l + 1 < 64) from = l + 1; }   // wave_shl:1
    else if (ctrl == 0x138) { if (l >= 1) from = l - 1; }        // wave_shr:1
    else if (ctrl == 0x142) { if (row >= 1) from = row * 16 - 1; }
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }
    else { std::fprintf(stderr, "emu: unsupported DPP control 0x%x\n", ctrl); std::abort(); }
    const int got = ::emu::shfl(src, from < 0 ? l : from);  // the rendezvous
    const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> (in_row >> 2)) & 1);
    if (!enabled) return old;
    if (from < 0) return bound_ctrl ? 0 : old;
    return got;
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) \
    emu_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl)
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)

template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <typename T> static inline T __hip_atomic_load(T* p, int, int) { return *p; }
template <typename T> static inline void __hip_atomic_store(T* p, T v, int, int) { *p = v; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
static inline void __threadfence() {}

// ---- host-side runtime
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->name, "SIMT emulator (CPU, tests only)");
    std::strcpy(p->gcnArchName, "emu");
    p->multiProcessorCount = 2;
    return hipSuccess;
}
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <typename T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc(reinterpret_cast<void**>(p), n, f); }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {  // nothing is pinned here: always staged
    a->type = hipMemoryTypeUnregistered; a->device = 0; a->devicePointer = nullptr; a->hostPointer = const_cast<void*>(p);
    return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
