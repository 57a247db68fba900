// Shared helpers for the gfx950 kernels of libgdn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/gdn_hip.h"

#define GDN_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int gdn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GDN_OK : GDN_ERR_LAUNCH;
}

// Plan overrides ride in gdn_conv_geom.hints (gdn_hip.h: GDN_HINT_PLAN_BATCH / GDN_HINT_PLAN_CUS / GDN_HINT_FFT_NP*): the library
// reads no environment variable -- a plan is a function of the geometry a call carries, also inside a captured graph.
int gdn_num_cus();      // conv_igemm.hip: CU count of the current device
// the batch size the plans are made for: every plan that depends on it -- the frequency-domain tile size, the direct kernels'
// tile configuration and split-K factor -- is chosen as if the batch were this, so that a batch-1 run can take the plans of a
// batch-n run and an image's result be compared BITWISE across batch sizes
// (tests/test_hip_robustness.py::test_legacy_inference_b64_graph_matches_single_image)
static inline int gdn_plan_batch(const gdn_conv_geom* g) { const int v = (g->hints >> 8) & 0xff; return v ? v : g->B; }
// the CU count the persistent kernels are planned for (tests: small shapes reach the multi-round / tail-split paths)
static inline int gdn_plan_cus(const gdn_conv_geom* g) { const int v = (g->hints >> 16) & 0xff; return v ? v * 8 : gdn_num_cus(); }

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- wave / block reductions (wave = 64 lanes on CDNA) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for a 256-thread block; result valid in thread 0. `sh` >= 4 doubles.
__device__ __forceinline__ double block_sum_d256(double v, double* sh) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- bf16 storage helpers (configs[2]: bf16 tensors, fp32 arithmetic) ----
__device__ __forceinline__ unsigned short f32_to_bf16_h(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);      // round to nearest even (NaN inputs do not occur on this path)
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_h_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// 4 consecutive elements at element index idx of a float (bf16 == 0) or bfloat16 (bf16 != 0) array
__device__ __forceinline__ f32x4 ld4_any(const void* p, size_t idx, int bf16) {
    if (!bf16) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + idx);
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + idx);
    f32x4 v = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
               __uint_as_float(u.y & 0xffff0000u)};
    return v;
}
__device__ __forceinline__ void st4_any(void* p, size_t idx, f32x4 v, int bf16) {
    if (!bf16) { *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + idx) = v; return; }
    uint2 u;
    u.x = (unsigned)f32_to_bf16_h(v[0]) | ((unsigned)f32_to_bf16_h(v[1]) << 16);
    u.y = (unsigned)f32_to_bf16_h(v[2]) | ((unsigned)f32_to_bf16_h(v[3]) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p) + idx) = u;
}
// 8 consecutive elements (two f32x4 halves): one 16-byte access for bf16, two for fp32
__device__ __forceinline__ void ld8_any(const void* p, size_t idx, int bf16, f32x4& lo, f32x4& hi) {
    if (!bf16) {
        lo = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + idx);
        hi = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + idx + 4);
        return;
    }
    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + idx);
    lo[0] = __uint_as_float(u.x << 16); lo[1] = __uint_as_float(u.x & 0xffff0000u);
    lo[2] = __uint_as_float(u.y << 16); lo[3] = __uint_as_float(u.y & 0xffff0000u);
    hi[0] = __uint_as_float(u.z << 16); hi[1] = __uint_as_float(u.z & 0xffff0000u);
    hi[2] = __uint_as_float(u.w << 16); hi[3] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void st8_any(void* p, size_t idx, f32x4 lo, f32x4 hi, int bf16) {
    if (!bf16) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + idx) = lo;
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + idx + 4) = hi;
        return;
    }
    uint4 u;
    u.x = (unsigned)f32_to_bf16_h(lo[0]) | ((unsigned)f32_to_bf16_h(lo[1]) << 16);
    u.y = (unsigned)f32_to_bf16_h(lo[2]) | ((unsigned)f32_to_bf16_h(lo[3]) << 16);
    u.z = (unsigned)f32_to_bf16_h(hi[0]) | ((unsigned)f32_to_bf16_h(hi[1]) << 16);
    u.w = (unsigned)f32_to_bf16_h(hi[2]) | ((unsigned)f32_to_bf16_h(hi[3]) << 16);
    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p) + idx) = u;
}

__device__ __forceinline__ float ld1_any(const void* p, size_t idx, int bf16) {
    return bf16 ? bf16_h_to_f32(reinterpret_cast<const unsigned short*>(p)[idx]) : reinterpret_cast<const float*>(p)[idx];
}
__device__ __forceinline__ void st1_any(void* p, size_t idx, float v, int bf16) {
    if (bf16) reinterpret_cast<unsigned short*>(p)[idx] = f32_to_bf16_h(v);
    else reinterpret_cast<float*>(p)[idx] = v;
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}
