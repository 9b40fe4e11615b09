// How long ONE random table lookup takes a wave (round 4: merge_kernel's and wordpiece_deferred_kernel's store lookups answer after
// ~10 us at the median while the span kernel's memo probes seem to cost 1-2): every lane of every wave reads one random 128-byte
// line (eight 16-byte loads, as store_lookup does) of a table of S MB, W waves in all, once; wall_clock64 around the loads.
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/random_probe tools/random_probe.hip && tools/build/random_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void probe(const uint4* table, uint32_t line_mask, uint32_t seed, int rounds, unsigned long long* ticks, uint32_t* sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t h = (gid + seed) * 0x9E3779B1u;
    uint32_t acc = 0;
    unsigned long long total = 0;
    for (int r = 0; r < rounds; ++r) {
        h ^= h >> 15;
        h *= 0x85EBCA77u;
        h ^= h >> 13;
        const uint4* e = table + size_t((h ^ acc) & line_mask) * 8;   // (acc: the next round depends on this one's data)
        const unsigned long long t0 = wall_clock64();
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = e[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].w;
        asm volatile("" : "+v"(acc));
        total += wall_clock64() - t0;
        acc &= 0;   // (keeps the dependence, not the value)
    }
    if ((threadIdx.x & 63) == 0) ticks[gid >> 6] = total;
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t max_bytes = size_t(512) << 20;
    uint4* table = nullptr;
    CHECK(hipMalloc(&table, max_bytes));
    CHECK(hipMemset(table, 0, max_bytes));
    unsigned long long* ticks = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&ticks, sizeof(unsigned long long) * 65536));
    CHECK(hipMalloc(&sink, 4));
    printf("table MB, waves, rounds: per-lookup latency of a wave in us (10 ns ticks): p50 p90 max\n");
    for (size_t mb : {4, 32, 64, 256, 512}) {
        for (int waves : {256, 1024, 4096, 16384}) {
            for (int rounds : {1, 4}) {
                const uint32_t line_mask = uint32_t((mb << 20) / 128 - 1);
                for (int rep = 0; rep < 3; ++rep) {
                    hipLaunchKernelGGL(probe, dim3(waves / 4), dim3(256), 0, 0, table, line_mask, 12345u + rep * 7919u, rounds, ticks, sink);
                    CHECK(hipDeviceSynchronize());
                }
                std::vector<unsigned long long> h(waves);
                CHECK(hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                printf("%4zu MB %6d waves %d rounds: %6.2f %6.2f %6.2f\n", mb, waves, rounds, h[waves / 2] / 100.0 / rounds, h[waves * 9 / 10] / 100.0 / rounds,
                       h[waves - 1] / 100.0 / rounds);
            }
        }
    }
    return 0;
}
