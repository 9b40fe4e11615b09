// Issue cost of instruction mixes on gfx950 (debug tool; nothing in the product depends on it).
// Every kernel runs `iters` rounds of one hand-written block of instructions; W waves per SIMD (blocks per CU = W, 256 threads).
// Reported: shader cycles per round and wave (elapsed * clock / (iters * W)) -- the number that says whether two kinds of
// instructions issued by the same wave (or by neighbouring waves) add up or overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7;
    uint32_t addr = (threadIdx.x * 37 + seed) & 1023;
    uint64_t s0 = seed, s1 = seed * 3 + 1;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0)  // 64 independent-ish VALU adds (4 chains)
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 1)  // 64 SALU 64-bit ops (2 chains)
            asm volatile(REP16("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n s_lshl_b64 %0, %0, 1\n s_xor_b64 %1, %1, %0\n")
                         : "+s"(s0), "+s"(s1) : : "scc");
        if (OP == 2)  // 64 VALU + 64 SALU interleaved one by one
            asm volatile(REP16("v_add_u32 %0, %0, %1\n s_and_b64 %4, %4, %5\n v_add_u32 %1, %1, %2\n s_or_b64 %5, %5, %4\n"
                               "v_add_u32 %2, %2, %3\n s_lshl_b64 %4, %4, 1\n v_add_u32 %3, %3, %0\n s_xor_b64 %5, %5, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1) : : "scc");
        if (OP == 3)  // 64 v_cmp into SGPR pairs (the "ballot for free" form)
            asm volatile(REP16("v_cmp_lt_u32 %4, %0, %1\n v_cmp_eq_u32 %5, %1, %2\n v_cmp_lt_u32 %4, %2, %3\n v_cmp_eq_u32 %5, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1));
        if (OP == 4)  // 64 ds_read_u8 at divergent addresses inside a 256-byte table, one wait per 16
            asm volatile(REP4(REP16("ds_read_u8 %1, %0\n") "s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xff, %1\n")
                         : "+v"(addr), "+v"(b));
        if (OP == 5)  // 64 v_perm_b32
            asm volatile(REP16("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 6)  // 64 v_mul_u32_u24
            asm volatile(REP16("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 7)  // 64 dependent VALU adds (one chain)
            asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (OP == 8)  // 64 SDWA compares with a byte select into vcc
            asm volatile(REP16("v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:BYTE_0 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD\n"
                               "v_cmp_eq_u32_sdwa vcc, %2, %3 src0_sel:BYTE_2 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %3, %0 src0_sel:BYTE_3 src1_sel:DWORD\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        if (OP == 9)  // 64 v_lshlrev_b64
            asm volatile(REP16("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 7, %1\n")
                         : "+v"(s0), "+v"(s1));
        if (OP == 10)  // 64 DPP moves
            asm volatile(REP16("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 11)  // 64 ds_read_b32 at divergent addresses over 4 KB, one wait per 16
            asm volatile(REP4(REP16("ds_read_b32 %1, %0\n") "s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xffc, %1\n")
                         : "+v"(addr), "+v"(b));
        if (OP == 12)  // v_readlane / s ops / v_writelane mix: 32 readlane + 32 salu
            asm volatile(REP16("v_readlane_b32 s20, %0, 3\n s_add_u32 s21, s20, 1\n v_readlane_b32 s22, %1, 5\n s_add_u32 s23, s22, 1\n")
                         : "+v"(a), "+v"(b) : : "s20", "s21", "s22", "s23", "scc");
        if (OP == 13)  // 64 v_bfe_u32
            asm volatile(REP16("v_bfe_u32 %0, %1, 8, 8\n v_bfe_u32 %1, %2, 16, 8\n v_bfe_u32 %2, %3, 8, 8\n v_bfe_u32 %3, %0, 16, 8\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 14)  // 64 v_mbcnt pairs (32 lo + 32 hi)
            asm volatile(REP16("v_mbcnt_lo_u32_b32 %0, %4, 0\n v_mbcnt_hi_u32_b32 %0, %5, %0\n v_mbcnt_lo_u32_b32 %1, %4, 0\n v_mbcnt_hi_u32_b32 %1, %5, %1\n")
                         : "+v"(a), "+v"(b) : "s"(uint32_t(s0)), "s"(uint32_t(s1)));

        if (OP == 20) asm volatile(REP16("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 21) asm volatile(REP16("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 22) asm volatile(REP16("v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %1, 5, %2\n v_lshrrev_b32 %2, 3, %3\n v_lshrrev_b32 %3, 1, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 23) asm volatile(REP16("v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 24) asm volatile(REP16("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        if (OP == 25) asm volatile(REP16("v_alignbit_b32 %0, %0, %1, 8\n v_alignbit_b32 %1, %1, %2, 8\n v_alignbit_b32 %2, %2, %3, 24\n v_alignbit_b32 %3, %3, %0, 8\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 26) asm volatile(REP16("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 27) asm volatile(REP16("v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 28) asm volatile(REP16("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x36\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x36\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x36\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x36\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 29) asm volatile(REP16("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 30) asm volatile(REP16("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 31) asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_eq_u32 vcc, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        if (OP == 32) asm volatile(REP16("v_add_u32 %0, 0x1f1f1f1f, %1\n v_add_u32 %1, 0x50505050, %2\n v_add_u32 %2, 0x77777777, %3\n v_add_u32 %3, 0x20202020, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 33) asm volatile(REP16("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 34) asm volatile(REP16("v_dot4_u32_u8 %0, %1, %2, %0\n v_dot4_u32_u8 %1, %2, %3, %1\n v_dot4_u32_u8 %2, %3, %0, %2\n v_dot4_u32_u8 %3, %0, %1, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 35) asm volatile(REP16("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %3, %0, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (OP == 36) asm volatile(REP16("v_ffbl_b32 %0, %1\n v_bcnt_u32_b32 %1, %2, %1\n v_ffbl_b32 %2, %3\n v_bcnt_u32_b32 %3, %0, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ addr ^ uint32_t(s0) ^ uint32_t(s1);
}

template <int OP>
void run(const char* name, int waves_per_simd) {
    uint32_t* out;
    const int blocks = 256 * waves_per_simd;
    hipMalloc(&out, size_t(blocks) * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    probe<OP><<<blocks, 256>>>(out, 10, 1);
    hipEventRecord(e0);
    probe<OP><<<blocks, 256>>>(out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("%-44s W=%d  %.3f ms  %.1f cycles per round and wave (SIMD time / rounds / W)\n", name, waves_per_simd, ms,
           ms * 1e-3 * clk * 1e3 / (double(iters) * waves_per_simd));
    hipFree(out);
}

int main() {
    for (int w : {4, 8}) {
        run<0>("64 v_add_u32 (4 chains)", w);
        run<7>("64 v_add_u32 (1 chain)", w);
        run<1>("64 SALU b64", w);
        run<2>("64 v_add + 64 SALU interleaved", w);
        run<3>("64 v_cmp -> sgpr", w);
        run<8>("64 v_cmp sdwa byte -> vcc", w);
        run<4>("64 ds_read_u8 divergent 256B", w);
        run<11>("64 ds_read_b32 divergent 4KB", w);
        run<5>("64 v_perm_b32", w);
        run<6>("64 v_mul_u32_u24", w);
        run<9>("64 v_lshlrev_b64", w);
        run<10>("64 v_mov dpp", w);
        run<12>("32 v_readlane + 32 s_add", w);
        run<13>("64 v_bfe_u32", w);
        run<14>("64 v_mbcnt", w);
        run<20>("64 v_and_b32", w);
        run<21>("64 v_xor_b32", w);
        run<22>("64 v_lshl/lshr_b32 imm", w);
        run<23>("64 v_sub_u32", w);
        run<24>("64 v_cndmask_b32", w);
        run<25>("64 v_alignbit_b32", w);
        run<26>("64 v_and_or_b32", w);
        run<27>("64 v_or3_b32", w);
        run<28>("64 v_bitop3_b32", w);
        run<29>("64 v_mad_u32_u24", w);
        run<30>("64 v_add3_u32", w);
        run<31>("64 v_cmp -> vcc", w);
        run<32>("64 v_add_u32 literal", w);
        run<33>("64 v_pk_add_u16", w);
        run<34>("64 v_dot4_u32_u8", w);
        run<35>("64 v_add_u32_dpp", w);
        run<36>("64 v_ffbl/v_bcnt", w);
    }
    return 0;
}
