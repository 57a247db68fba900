// Diagnostic only (tests/diag/lds_poison.py): fills the whole LDS of every CU with a bit pattern, so that a kernel of the
// library that reads LDS it never wrote computes with that pattern (0xFFFFFFFF = NaN) instead of with leftovers of the
// workgroup that ran there before.  Not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void lds_fill_kernel(uint32_t pattern, uint32_t* sink) {
    extern __shared__ uint32_t lds[];
    volatile uint32_t* v = lds;
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) v[i] = pattern;
    __syncthreads();
    for (int k = 0; k < 64; ++k) __builtin_amdgcn_s_sleep(32);      // keep the CU occupied so the other workgroups spread out
    if (v[threadIdx.x] != pattern) sink[0] = 1;
}

extern "C" int lds_fill(uint32_t pattern, void* sink, void* stream) {
    static bool set = false;
    if (!set) {
        if (hipFuncSetAttribute((const void*)lds_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
        set = true;
    }
    hipLaunchKernelGGL(lds_fill_kernel, dim3(512), dim3(1024), 160 * 1024, (hipStream_t)stream, pattern, (uint32_t*)sink);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
