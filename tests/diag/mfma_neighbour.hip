// Diagnostic only (tests/diag/torch_victim.py): a kernel that is nothing but a loop of matrix-core instructions, launched
// over and over by a NEIGHBOUR process while another process checks its own (vendor) kernels for reproducibility.
// variant: 0 v_mfma_f32_32x32x16_bf16, 4 accumulators round robin     1 the same, ONE accumulator (dependent chain)
//          2 v_mfma_f32_16x16x32_bf16, 4 accumulators                  3 v_mfma_f32_32x32x2_f32, 4 accumulators
//          4 v_mfma_f32_32x32x16_f16, 4 accumulators                   5 variant 0 with a workgroup barrier every 12 instructions
//          6 variant 3 (fp32) with the barrier                          7 variant 4 (f16) with the barrier
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V>
__global__ __launch_bounds__(256, 2) void mfma_loop_kernel(float* sink, int iters, float seed) {
    f32x16 c[4];
    f32x4 d[4];
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
        for (int r = 0; r < 4; ++r) d[i][r] = 0.f;
    }
    const float s = seed + threadIdx.x * 1e-3f;
    bf16x8 a, b;
    f16x8 ah, bh;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(s + q); b[q] = (__bf16)(1.f - s * q); ah[q] = (_Float16)(s + q); bh[q] = (_Float16)(1.f - s * q); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int i = (V == 1) ? 0 : (u & 3);
            if (V == 0 || V == 1 || V == 5) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
            if (V == 2) d[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d[i], 0, 0, 0);
            if (V == 3 || V == 6) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(s, 1.f - s, c[i], 0, 0, 0);
            if (V == 4 || V == 7) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[i], 0, 0, 0);
        }
        if (V >= 5) __syncthreads();
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) t += c[i][r];
        for (int r = 0; r < 4; ++r) t += d[i][r];
    }
    if (t == 12345.678f) sink[0] = t;
}

extern "C" int mfma_loop(int variant, int blocks, int iters, void* sink, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 0: hipLaunchKernelGGL(mfma_loop_kernel<0>, dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 0.5f); break;
        case 1: hipLaunchKernelGGL(mfma_loop_kernel<1>, dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 0.5f); break;
        case 2: hipLaunchKernelGGL(mfma_loop_kernel<2>, dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 0.5f); break;
        case 3: hipLaunchKernelGGL(mfma_loop_kernel<3>, dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 0.5f); break;
        case 4: hipLaunchKernelGGL(mfma_loop_kernel<4>, dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 0.5f); break;
        case 6: hipLaunchKernelGGL(mfma_loop_kernel<6>, dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 0.5f); break;
        case 7: hipLaunchKernelGGL(mfma_loop_kernel<7>, dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 0.5f); break;
        default: hipLaunchKernelGGL(mfma_loop_kernel<5>, dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 0.5f); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
