// FETCH_SIZE / WRITE_SIZE calibration on the access patterns the encode kernels use (MI355X_MICROARCH.md "HBM": on gfx950
// FETCH_SIZE reports half the bytes of a 16-B-per-lane streaming read; "other access widths and WRITE_SIZE are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Every kernel touches a KNOWN number of bytes; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/build/fetch_calib      (and once more with --pmc WRITE_SIZE)
// and compare the counters with the byte counts this program prints (tools/summarize_calib.py).
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// 16 bytes per lane, consecutive lanes consecutive: the guide's reference pattern
__global__ void calib_stream16(const uint4* p, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// 4 bytes per lane per load: a wave-instruction covers 256 consecutive bytes (stage_window: a row's text as dwords)
__global__ void calib_stream4(const uint32_t* p, size_t n4, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += size_t(gridDim.x) * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// rows of ~512 bytes at byte offsets that are not multiples of 4 (aligned dwords around each row, as stage_window reads)
__global__ void calib_rows(const uint8_t* text, const uint32_t* row_off, int n_rows, uint32_t* sink) {
    uint32_t acc = 0;
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < n_rows; r += n_waves) {
        const uint32_t b = row_off[r], e = row_off[r + 1];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(text + (b & ~3u));
        const int nd = int(((e + 3) & ~3u) - (b & ~3u)) / 4;
        for (int k = lane; k < nd; k += 64) acc ^= w[k];
    }
    if (acc == 0x12345678u) *sink = acc;
}
// random 32-byte entries (two 16-byte loads) of a table: the memo probe
__global__ void calib_probe32(const uint4* table, uint32_t mask, size_t n_probes, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_probes; i += size_t(gridDim.x) * blockDim.x) {
        uint32_t h = uint32_t(i) * 0x9E3779B1u;
        h ^= h >> 15;
        h *= 0x85EBCA77u;
        const uint32_t slot = (h >> 7) & mask;
        const uint4 a = table[2 * size_t(slot)], b = table[2 * size_t(slot) + 1];
        acc ^= a.x ^ b.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// 4 bytes per lane, consecutive: the staging / ids writes
__global__ void calib_write4(uint32_t* p, size_t n4) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += size_t(gridDim.x) * blockDim.x) p[i] = uint32_t(i);
}
// 2 bytes per lane, consecutive: u16 ids
__global__ void calib_write2(uint16_t* p, size_t n2) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n2; i += size_t(gridDim.x) * blockDim.x) p[i] = uint16_t(i);
}
// 16 bytes per lane
__global__ void calib_write16(uint4* p, size_t n16) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x)
        p[i] = uint4{uint32_t(i), 1u, 2u, 3u};
}
// sparse 4-byte writes: one dword in every 64-byte block (the row arrays / deferred records written here and there)
__global__ void calib_write_sparse(uint32_t* p, size_t n_blocks) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_blocks; i += size_t(gridDim.x) * blockDim.x) p[i * 16] = uint32_t(i);
}

int main() {
    const size_t N = size_t(512) << 20;   // 512 MiB: twice the Infinity Cache
    uint8_t* buf = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&buf, N));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, N));
    const int grid = 256 * 8, block = 256;
    // rows: ~512 bytes each, odd offsets
    const int n_rows = 500000;
    std::vector<uint32_t> off(n_rows + 1);
    uint32_t at = 3;
    for (int r = 0; r <= n_rows; ++r) { off[r] = at; at += 461 + uint32_t((r * 37) % 103); }
    uint32_t* d_off = nullptr;
    CHECK(hipMalloc(&d_off, off.size() * 4));
    CHECK(hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_stream16, dim3(grid), dim3(block), 0, 0, reinterpret_cast<const uint4*>(buf), N / 16, sink);
    printf("calib_stream16 read_bytes %zu\n", N);
    hipLaunchKernelGGL(calib_stream4, dim3(grid), dim3(block), 0, 0, reinterpret_cast<const uint32_t*>(buf), N / 4, sink);
    printf("calib_stream4 read_bytes %zu\n", N);
    hipLaunchKernelGGL(calib_rows, dim3(grid), dim3(block), 0, 0, buf, d_off, n_rows, sink);
    printf("calib_rows read_bytes %zu\n", size_t(off[n_rows] - off[0]));
    for (uint32_t mb : {4u, 16u, 64u, 512u}) {   // table sizes: an XCD's L2, 4 x it, beyond all L2s, beyond the Infinity Cache
        const uint32_t slots = (mb << 20) / 32;
        const size_t probes = size_t(16) << 20;
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(calib_probe32, dim3(grid), dim3(block), 0, 0, reinterpret_cast<const uint4*>(buf), slots - 1, probes, sink);
        printf("calib_probe32 table_MiB %u probes %zu probe_bytes %zu\n", mb, probes, probes * 32);
    }
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_write4, dim3(grid), dim3(block), 0, 0, reinterpret_cast<uint32_t*>(buf), N / 4);
    printf("calib_write4 write_bytes %zu\n", N);
    hipLaunchKernelGGL(calib_write2, dim3(grid), dim3(block), 0, 0, reinterpret_cast<uint16_t*>(buf), N / 2);
    printf("calib_write2 write_bytes %zu\n", N);
    hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(block), 0, 0, reinterpret_cast<uint4*>(buf), N / 16);
    printf("calib_write16 write_bytes %zu\n", N);
    hipLaunchKernelGGL(calib_write_sparse, dim3(grid), dim3(block), 0, 0, reinterpret_cast<uint32_t*>(buf), N / 64);
    printf("calib_write_sparse write_bytes %zu (one dword per 64-byte block over %zu bytes)\n", N / 16, N);
    CHECK(hipDeviceSynchronize());
    return 0;
}
