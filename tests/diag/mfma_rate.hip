// Diagnostic (standalone): what the bf16 matrix pipe sustains on this chip, one 512-thread workgroup per CU, and at what shader
// clock -- the yardstick for the ring kernels (conv_ring.h, gemm_x3_ring.h).  Per variant: time, TFLOP/s, shader clock measured
// inside the kernel (s_memtime cycles / s_memrealtime 100 MHz ticks), MFMA pipe share = issued MFMA cycles / elapsed cycles.
//   0  8 accumulators round robin (no dependent neighbours), nothing else
//   1  the x3 pattern: five dependent MFMAs on one accumulator, then one on another (gemm_x3_ring quarter), 4 tile pairs
//   2  variant 1 + one workgroup barrier per 24 MFMAs
//   3  variant 1 + 15 ds_read_b128 per 24 MFMAs (operands really come from LDS, read one quarter ahead)
//   4  variant 3 + barrier per 24 MFMAs
//   5  variant 4 + s_setprio 1 around the MFMAs
//   6  variant 0 at 256 threads x 2 workgroups per CU
// hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int V>
__global__ __launch_bounds__(512, 2) void rate_kernel(float* sink, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[147456];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (V >= 3) for (int i = tid; i < 147456 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(sm)[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    f32x16 acc[4], cor[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; cor[i][r] = 0.f; }
    bf16x8 A[2][3], B[2][3];
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) for (int q = 0; q < 8; ++q) { A[i][p][q] = (__bf16)(1.f + 0.01f * (lane + q + p)); B[i][p][q] = (__bf16)(0.5f - 0.01f * (lane + i)); }
    const int r32 = lane & 31, h = lane >> 5;
    const int la = (wave >> 1) * 4096 + r32 * 32 + ((h ^ ((r32 >> 3) & 1)) << 4);
    auto ld3 = [&](bf16x8 (&d)[3], int t, int base) {
        const unsigned char* s = sm + (t & 3) * 36864 + base + la;
#pragma unroll
        for (int p = 0; p < 3; ++p) d[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(s + p * 4096));
    };
    auto quarter = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& m, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], m, 0, 0, 0);
    };
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0 || V == 6) {
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                if (u & 4) cor[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[u & 1][u % 3], B[(u >> 1) & 1][u % 3], cor[u & 3], 0, 0, 0);
                else acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[u & 1][u % 3], B[(u >> 1) & 1][u % 3], acc[u & 3], 0, 0, 0);
            }
        } else {
            if (V >= 3) { ld3(B[1], it, 12288); __builtin_amdgcn_sched_barrier(0); }
            if (V == 5) __builtin_amdgcn_s_setprio(1);
            quarter(A[0], B[0], acc[0], cor[0]);
            if (V == 5) __builtin_amdgcn_s_setprio(0);
            if (V >= 3) { __builtin_amdgcn_sched_barrier(0); ld3(A[1], it, 1024); __builtin_amdgcn_sched_barrier(0); }
            if (V == 5) __builtin_amdgcn_s_setprio(1);
            quarter(A[0], B[1], acc[1], cor[1]);
            if (V == 5) __builtin_amdgcn_s_setprio(0);
            if (V >= 3) { __builtin_amdgcn_sched_barrier(0); ld3(B[0], it, 24576); __builtin_amdgcn_sched_barrier(0); }
            if (V == 5) __builtin_amdgcn_s_setprio(1);
            quarter(A[1], B[1], acc[3], cor[3]);
            if (V == 5) __builtin_amdgcn_s_setprio(0);
            if (V >= 3) { __builtin_amdgcn_sched_barrier(0); ld3(A[0], it + 1, 0); ld3(B[1], it + 1, 12288 + 2048); __builtin_amdgcn_sched_barrier(0); }
            if (V == 5) __builtin_amdgcn_s_setprio(1);
            quarter(A[1], B[0], acc[2], cor[2]);
            if (V == 5) __builtin_amdgcn_s_setprio(0);
            if (V >= 3) __builtin_amdgcn_sched_barrier(0);
            if (V == 2 || V >= 4) __builtin_amdgcn_s_barrier();
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r] + cor[i][r];
    if (t == 12345.678f) sink[0] = t;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int V> static void run(const char* name, int threads, int blocks, int iters, float* sink, unsigned long long* clk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<V>, dim3(blocks), dim3(threads), 0, 0, sink, clk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = 2.0 * 32 * 32 * 16 * 24.0 * iters * (threads / 64) * blocks;
        const double mhz = 100.0 * (double)h[0] / (double)h[1];
        const double pipe = 24.0 * iters * (threads / 64 / 4) * 32.0 / (double)h[0];      // MFMA cycles issued per SIMD / elapsed cycles
        if (rep == 2) printf("%-70s %7.3f ms  %7.1f TF  clock %4.0f MHz  pipe share %.2f   [raw: s_memtime %llu, s_memrealtime %llu -> %.1f / %.1f MHz against the event time]\n",
                             name, ms, flops / ms / 1e9, mhz, pipe, h[0], h[1], h[0] / ms / 1e3, h[1] / ms / 1e3);
    }
}

int main() {
    float* sink; unsigned long long* clk;
    hipMalloc(&sink, 64); hipMalloc(&clk, 64);
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int it = 20000;
    run<0>("0 independent accumulators, nothing else", 512, cus, it, sink, clk);
    run<1>("1 x3 pattern (5 dependent + 1)", 512, cus, it, sink, clk);
    run<2>("2 x3 pattern + barrier per 24", 512, cus, it, sink, clk);
    run<3>("3 x3 pattern + 15 ds_read_b128 per 24", 512, cus, it, sink, clk);
    run<4>("4 x3 pattern + reads + barrier", 512, cus, it, sink, clk);
    run<5>("5 x3 pattern + reads + barrier + setprio", 512, cus, it, sink, clk);
    run<6>("6 independent accumulators, 256 threads x 2 workgroups per CU", 256, 2 * cus, it, sink, clk);
    run<0>("0 again", 512, cus, it, sink, clk);
    return 0;
}
