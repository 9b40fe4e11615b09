// Issue cost of a few integer VALU instructions on gfx950 (debug tool): 4 independent chains per lane, 8 waves per SIMD,
// cycles per wave-instruction = elapsed * clock / (instructions per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7;
    unsigned long long p = a, q = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) { a += b; b += c; c += d; d += a; }
            if (OP == 1) { a *= b | 1; b *= c | 1; c *= d | 1; d *= a | 1; }   // v_or + v_mul_lo_u32
            if (OP == 2) { a = __umul24(a, b); b = __umul24(b, c); c = __umul24(c, d); d = __umul24(d, a); }
            if (OP == 3) { p <<= (q & 7); q += p; p ^= q >> (p & 7); q ^= p; }  // 64-bit shifts
            if (OP == 4) { a = __umulhi(a, b | 1); b = __umulhi(b, c | 1) + 3; c = __umulhi(c, d | 1) + 5; d = __umulhi(d, a | 1) + 7; }
            if (OP == 5) { a |= b; b |= c; c |= d; d |= a; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ uint32_t(p) ^ uint32_t(q);
}
template <int OP>
void run(const char* name, int per_iter) {
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    rate_kernel<OP><<<blocks, 256>>>(out, 10, 1);
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, 256>>>(out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double insts = double(iters) * 16 * per_iter * 8;  // per SIMD
    printf("%-28s %.3f ms, %.2f cycles per listed op at %d MHz\n", name, ms, ms * 1e-3 * clk * 1e3 / insts, clk / 1000);
    hipFree(out);
}
int main() {
    run<0>("v_add_u32", 4);
    run<5>("v_or_b32", 4);
    run<1>("v_or + v_mul_lo_u32 (pair)", 4);
    run<2>("v_mul_u32_u24", 4);
    run<4>("v_or + v_mul_hi_u32 (+add)", 4);
    run<3>("64-bit shift mix (4 stmts)", 4);
    return 0;
}
