// Diagnostic only (tests/diag/victim_classes.py): small kernels of distinct instruction mixes, each deterministic, run on one
// stream while a second stream runs the barrier-paced bf16 matrix loop -- which kind of kernel is disturbed?
//   0 scalar fp32 FMA chain (registers only)          1 packed fp32 FMA chain (float2 math -> v_pk_fma_f32)
//   2 LDS transpose with workgroup barriers           3 LDS round trip private to a wave (no barrier)
//   4 private-memory (scratch) array, dynamic index   5 global copy with a multiply
//   6 LDS transpose + barriers + packed math (an FFT-like butterfly stage)
//   8 table look-ups, wave-uniform index (scalar cache)   9 table look-ups, per-lane index (vector L1 hits)
//   10 radix-2 butterflies with twiddles on float2 registers (v_pk_add / v_pk_mul with swaps), no LDS, no tables
//   7 a 192-register working set (statically indexed array kept live across a long chain): ~200 VGPRs per lane
//   3 was removed (its wave-private LDS exchange raced with itself)
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ const float kTab[256] = {
#define T4(a) (a) * 0.01f + 0.5f, (a) * 0.01f + 0.51f, (a) * 0.01f + 0.52f, (a) * 0.01f + 0.53f
#define T16(a) T4(a), T4(a + 1), T4(a + 2), T4(a + 3)
#define T64(a) T16(a), T16(a + 4), T16(a + 8), T16(a + 12)
    T64(0), T64(16), T64(32), T64(48)};

template <int V>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ float2 lds[64 * 33];
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    float x = in[gid], y = in[gid ^ 1];
    if (V == 0) {
        for (int i = 0; i < iters; ++i) { x = fmaf(x, 0.999f, y * 1e-3f); y = fmaf(y, 1.001f, -x * 1e-3f); }
    } else if (V == 1) {
        float2 a = make_float2(x, y), b = make_float2(y, x);
        for (int i = 0; i < iters; ++i) {
            a = make_float2(fmaf(a.x, 0.999f, b.x * 1e-3f), fmaf(a.y, 0.999f, b.y * 1e-3f));
            b = make_float2(fmaf(b.x, 1.001f, -a.x * 1e-3f), fmaf(b.y, 1.001f, -a.y * 1e-3f));
        }
        x = a.x + b.y; y = a.y - b.x;
    } else if (V == 2 || V == 6) {
        const int r = tid >> 3, c = tid & 7;          // 32 x 8 tile of float2, transposed through LDS every iteration
        float2 v = make_float2(x, y);
        for (int i = 0; i < iters; ++i) {
            lds[r * 9 + c] = v;
            __syncthreads();
            const float2 w = lds[((tid & 31)) * 9 + (tid >> 5)];
            __syncthreads();
            if (V == 6) {                            // butterfly with a twiddle: packed multiply-adds
                const float cs = 0.92387953f, sn = 0.38268343f;
                v = make_float2(fmaf(w.x, cs, -w.y * sn) + v.x * 0.5f, fmaf(w.x, sn, w.y * cs) - v.y * 0.5f);
                v = make_float2(v.x * 0.7f, v.y * 0.7f);
            } else {
                v = make_float2(w.y * 0.9999f, w.x * 1.0001f);
            }
        }
        x = v.x; y = v.y;
    } else if (V == 7) {
        float r[192];
#pragma unroll
        for (int q = 0; q < 192; ++q) r[q] = x + q * 1e-3f * y;
        for (int i = 0; i < iters / 8; ++i) {
#pragma unroll
            for (int q = 0; q < 192; ++q) r[q] = fmaf(r[q], 0.9995f, r[(q + 7) % 192] * 5e-4f);
        }
        x = 0.f;
#pragma unroll
        for (int q = 0; q < 192; ++q) x += r[q];
    } else if (V == 10) {                 // complex rotations on eight independent float2 values: v_pk_mul_f32 / v_pk_add_f32 with lane swaps,
        float2 v[8];                       // the instruction mix of a register FFT (no LDS, no tables)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = make_float2(x + q * 0.125f, y - q * 0.0625f);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < 8; q += 2) {                    // radix-2 butterfly + twiddle
                const float2 a = v[q], b = v[q + 1];
                const float2 s = make_float2(a.x + b.x, a.y + b.y), d = make_float2(a.x - b.x, a.y - b.y);
                const float cs = 0.92387953f, sn = -0.38268343f;
                v[q] = make_float2(s.x * 0.5f, s.y * 0.5f);
                v[q + 1] = make_float2((d.x * cs - d.y * sn) * 0.5f, (d.x * sn + d.y * cs) * 0.5f);
            }
            const float2 t = v[0]; v[0] = v[3]; v[3] = v[6]; v[6] = v[5]; v[5] = t;
        }
        x = 0.f; y = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { x += v[q].x; y += v[q].y; }
    } else if (V == 8) {                  // table look-ups with a wave-uniform index: scalar loads through the scalar cache
        for (int i = 0; i < iters; ++i) x = fmaf(x, kTab[(blockIdx.x + i * 7) & 255], y * 1e-3f) * 0.7f;
    } else if (V == 9) {                  // table look-ups with a per-lane index: vector loads that hit the L1 cache
        int j = tid;
        for (int i = 0; i < iters; ++i) { x = fmaf(x, kTab[j & 255], y * 1e-3f) * 0.7f; j = j * 5 + 1 + i; }
    } else if (V == 4) {
        float arr[48];
        for (int q = 0; q < 48; ++q) arr[q] = x + q * y;
        for (int i = 0; i < iters; ++i) {
            const int j = (int)(fabsf(arr[i % 48]) * 7.f) % 48;    // data-dependent index: the array lives in scratch
            arr[j] = arr[j] * 0.999f + arr[(j + 5) % 48] * 1e-3f;
        }
        x = 0.f;
        for (int q = 0; q < 48; ++q) x += arr[q];
    } else {
        for (int i = 0; i < iters; ++i) { x = in[(gid + i * 1024) & 0xfffff] * 1.0001f + x * 0.5f; }
    }
    out[gid] = x + y;
}

extern "C" int victim(int variant, const void* in, void* out, int blocks, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const float* i = (const float*)in; float* o = (float*)out;
    switch (variant) {
        case 0: hipLaunchKernelGGL(victim_kernel<0>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
        case 1: hipLaunchKernelGGL(victim_kernel<1>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
        case 2: hipLaunchKernelGGL(victim_kernel<2>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
        case 3: hipLaunchKernelGGL(victim_kernel<7>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
        case 4: hipLaunchKernelGGL(victim_kernel<4>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
        case 10: hipLaunchKernelGGL(victim_kernel<10>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
        case 8: hipLaunchKernelGGL(victim_kernel<8>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
        case 9: hipLaunchKernelGGL(victim_kernel<9>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
        case 5: hipLaunchKernelGGL(victim_kernel<5>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
        default: hipLaunchKernelGGL(victim_kernel<6>, dim3(blocks), dim3(256), 0, st, i, o, iters); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
