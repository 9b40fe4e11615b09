// Debug tool (nothing in the product depends on it): are 16-byte global loads at byte addresses correct on gfx950, and what do
// they cost next to aligned ones?  Every lane reads 32 consecutive bytes (two dwordx4) at base + skew + 32 * lane, the way
// lookup_span_kernel reads a block of text; the checksum is compared with a byte-wise host sum; then the rate of v_dot4_u32_u8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

struct __attribute__((packed, aligned(1))) U4 { uint32_t x, y, z, w; };

__global__ __launch_bounds__(256) void read32(const uint8_t* base, size_t n_lanes, int skew, unsigned long long* sum, int reps) {
    unsigned long long acc = 0;
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    for (int r = 0; r < reps; ++r)
        for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_lanes; i += stride) {
            const U4* p = reinterpret_cast<const U4*>(base + skew + 32 * i);
            const U4 a = p[0], b = p[1];
            acc += a.x + 3ull * a.y + 5ull * a.z + 7ull * a.w + 11ull * b.x + 13ull * b.y + 17ull * b.z + 19ull * b.w;
        }
    atomicAdd(sum, acc);
}

__global__ __launch_bounds__(256) void dot_rate(uint32_t* out, int iters) {
    uint32_t a = threadIdx.x * 0x01010101u, b = 0x08040201u, c = 0, d = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            c = __builtin_amdgcn_udot4(a, b, c, false);
            d = __builtin_amdgcn_udot4(c, b, d, false);
            a = __builtin_amdgcn_udot4(d, b, a, false);
            c = __builtin_amdgcn_udot4(a, d, c, false);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ c ^ d;
}

int main() {
    const size_t bytes = size_t(256) << 20, lanes = (bytes - 64) / 32;
    std::vector<uint8_t> h(bytes);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = uint8_t(s >> 24); }
    uint8_t* d; unsigned long long* dsum;
    hipMalloc(&d, bytes); hipMalloc(&dsum, 8);
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int skew = 0; skew < 4; ++skew) {
        unsigned long long want = 0;
        for (size_t i = 0; i < lanes; ++i) {
            uint32_t w[8];
            memcpy(w, h.data() + skew + 32 * i, 32);
            want += w[0] + 3ull * w[1] + 5ull * w[2] + 7ull * w[3] + 11ull * w[4] + 13ull * w[5] + 17ull * w[6] + 19ull * w[7];
        }
        hipMemset(dsum, 0, 8);
        read32<<<2048, 256>>>(d, lanes, skew, dsum, 1);
        unsigned long long got = 0;
        hipMemcpy(&got, dsum, 8, hipMemcpyDeviceToHost);
        hipEventRecord(e0);
        read32<<<2048, 256>>>(d, lanes, skew, dsum, 4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("skew %d: %s  %.3f ms for 4 x 256 MiB = %.0f GB/s\n", skew, got == want ? "correct" : "WRONG", ms, 4.0 * bytes / ms / 1e6);
    }
    uint32_t* out; hipMalloc(&out, 2048 * 256 * 4);
    dot_rate<<<2048, 256>>>(out, 10);
    hipEventRecord(e0);
    dot_rate<<<2048, 256>>>(out, 2000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("v_dot4_u32_u8 (dependent pairs, 8 waves per SIMD): %.2f cycles per instruction\n", ms * 1e-3 * clk * 1e3 / (2000.0 * 64 * 8));
    return 0;
}
