// cu_hog: a kernel that just sits on CU slots for a given time -- stands in for RCCL's persistent all-gather kernel when
// probing, on one GPU, how the encode kernels behave while a collective occupies part of the machine (bench.py --hog).
// Debug tool; not part of libovtk_amd.so.
#include <hip/hip_runtime.h>

__global__ void hog_kernel(long long ticks) {
    extern __shared__ int lds[];  // the launch's dynamic LDS is what keeps other blocks off the CU
    if (ticks < 0) lds[threadIdx.x] = 1;
    const long long t0 = wall_clock64();  // 100 MHz constant clock
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int cu_hog(int blocks, int threads, int lds_bytes, double micros, void* stream) {
    if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(threads), lds_bytes, static_cast<hipStream_t>(stream), (long long)(micros * 100.0));
    return int(hipGetLastError());
}
