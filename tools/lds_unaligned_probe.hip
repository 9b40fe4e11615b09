// Do gfx950 LDS reads at byte addresses work (ds_read_b128 / ds_read_b32 behind packed, align-1 types)?  Run on the box:
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/lds_unaligned_probe tools/lds_unaligned_probe.hip && tools/build/lds_unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
struct __attribute__((packed, aligned(1))) U16B { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U4B { uint32_t x; };
__global__ void k(uint32_t* out, int stride, int base) {
    __shared__ uint32_t words[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
        const uint32_t b = 4u * i;
        words[i] = ((b & 0xFF)) | (((b + 1) & 0xFF) << 8) | (((b + 2) & 0xFF) << 16) | (((b + 3) & 0xFF) << 24);
    }
    __syncthreads();
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(words);
    const int off = base + int(threadIdx.x) * stride;
    const U16B v = *reinterpret_cast<const U16B*>(bytes + off);
    const U4B u = *reinterpret_cast<const U4B*>(bytes + off + 16);
    out[threadIdx.x * 5 + 0] = v.x; out[threadIdx.x * 5 + 1] = v.y; out[threadIdx.x * 5 + 2] = v.z; out[threadIdx.x * 5 + 3] = v.w;
    out[threadIdx.x * 5 + 4] = u.x;
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 5 * 4);
    int bad = 0;
    for (int stride : {7, 13, 16, 1, 5}) for (int base : {0, 1, 2, 3}) {
        k<<<1, 64>>>(d, stride, base);
        uint32_t h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int t = 0; t < 64; ++t) for (int q = 0; q < 5; ++q) {
            const int o = base + t * stride + 4 * q;
            uint32_t e = 0; for (int b = 0; b < 4; ++b) e |= uint32_t((o + b) & 0xFF) << (8 * b);
            if (h[t * 5 + q] != e) { if (bad < 5) printf("mismatch stride %d base %d lane %d q %d: %08x != %08x\n", stride, base, t, q, h[t*5+q], e); ++bad; }
        }
    }
    printf("unaligned LDS reads: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return bad != 0;
}
