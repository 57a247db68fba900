// KITTI training-time augmentation on the device, bit-exact with the reference's host pipeline
// (datasets_list.py:82-101 -> transform_list.py RandomHorizontalFlip :158-166, RandomScaleCrop :185-199,
// ArrayToTensor :100-118, Normalize :84-92).  RandomScaleCrop's scipy.misc.imresize is bytescale (min-max
// stretch of non-uint8 data to 0..255) + Pillow's two-pass bilinear resampler: double-precision triangle
// weights normalised to 1, 22-bit fixed point, horizontal pass rounded to uint8, then the vertical pass.
// For the up-scaling RandomScaleCrop draws (1.0-1.15x) every output sample has <= 3 taps per axis, so one
// thread rebuilds the <= 3 horizontally filtered bytes it needs and filters them vertically: 9 source reads
// per output pixel, no intermediate image.  HBM-bound byte work: 1 B/channel read (L2-resident re-reads),
// 4 B/channel written.
#include "common.h"

namespace {

#define AUG_PREC 22

struct AxisK { int x0, n; int k[3]; };

// Pillow precompute_coeffs + normalize_coeffs_8bpc for one output coordinate (bilinear filter, support 1).
// Contraction is off: the weights must round exactly as the host's separate multiplies and adds do.
__device__ AxisK axis_coeffs(int in_size, int out_size, int xx) {
#pragma clang fp contract(off)
    AxisK a;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > 3) xmax = 3;          // up-scaling only (host checks scaled >= in)
    double w[3] = {0.0, 0.0, 0.0}, ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        w[x] = t < 1.0 ? 1.0 - t : 0.0;
        ww += w[x];
    }
    for (int x = 0; x < 3; ++x) {
        double v = w[x];
        if (x < xmax && ww != 0.0) v = v / ww;
        a.k[x] = (int)(0.5 + v * (double)(1 << AUG_PREC));
    }
    a.x0 = xmin; a.n = xmax;
    return a;
}

__device__ __forceinline__ int clip8(long long v) {
    const long long s = v >> AUG_PREC;
    return (int)(s < 0 ? 0 : (s > 255 ? 255 : s));
}

// per-image min / max of a float source (bytescale's cmin, cmax)
__global__ __launch_bounds__(256) void aug_minmax_kernel(const float* __restrict__ src, int64_t per_image, float* mm) {
    __shared__ float smin[4], smax[4];
    const float* p = src + (size_t)blockIdx.x * per_image;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = threadIdx.x; i < per_image; i += 256) { const float v = p[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[blockIdx.x * 2 + 0] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        mm[blockIdx.x * 2 + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

__device__ __forceinline__ int src_byte(const void* src, int f32, size_t idx, float cmin, float bscale) {
#pragma clang fp contract(off)
    if (!f32) return reinterpret_cast<const unsigned char*>(src)[idx];
    float b = (reinterpret_cast<const float*>(src)[idx] - cmin) * bscale;     // scipy bytescale, float32 arithmetic
    b = fminf(fmaxf(b, 0.f), 255.f) + 0.5f;
    return (int)(unsigned char)b;
}

__device__ __forceinline__ float normalize01(float v) {
    return __fdiv_rn(__fdiv_rn(v, 255.0f) - 0.5f, 0.5f);        // ArrayToTensor /255, Normalize (t - 0.5) / 0.5
}

__global__ __launch_bounds__(256) void aug_kernel(const void* __restrict__ src, int f32, int B, int H, int W, int C,
                                                  const int* __restrict__ params, int train,
                                                  const float* __restrict__ mm, float* __restrict__ dst) {
    const int64_t total = (int64_t)B * H * W;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % W);
        const int oy = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
        const size_t img = (size_t)b * H * W * C;
        float* out = dst + (size_t)b * C * H * W + (size_t)oy * W + ox;
        if (!train) {
            for (int c = 0; c < C; ++c) {
                const size_t idx = img + ((size_t)oy * W + ox) * C + c;
                const float v = f32 ? reinterpret_cast<const float*>(src)[idx]
                                    : (float)reinterpret_cast<const unsigned char*>(src)[idx];
                out[(size_t)c * H * W] = normalize01(v);
            }
            continue;
        }
        const int* pr = params + b * 5;
        const int flip = pr[0], sh = pr[1], sw = pr[2], offy = pr[3], offx = pr[4];
        float cmin = 0.f, bscale = 1.f;
        if (f32) {
            cmin = mm[b * 2];
            float cs = mm[b * 2 + 1] - cmin;
            if (cs == 0.f) cs = 1.f;
            bscale = (float)(255.0 / (double)cs);
        }
        const AxisK ky = axis_coeffs(H, sh, oy + offy);
        const AxisK kx = axis_coeffs(W, sw, ox + offx);
        for (int c = 0; c < C; ++c) {
            long long v = 1ll << (AUG_PREC - 1);
            for (int r = 0; r < ky.n; ++r) {
                const int sy = ky.x0 + r;
                int hb;
                if (sw != W) {                      // horizontal pass (only when the width changes, as Pillow does)
                    long long hsum = 1ll << (AUG_PREC - 1);
                    for (int j = 0; j < kx.n; ++j) {
                        int sx = kx.x0 + j;
                        if (flip) sx = W - 1 - sx;
                        hsum += (long long)src_byte(src, f32, img + ((size_t)sy * W + sx) * C + c, cmin, bscale) * kx.k[j];
                    }
                    hb = clip8(hsum);
                } else {
                    int sx = ox + offx;
                    if (flip) sx = W - 1 - sx;
                    hb = src_byte(src, f32, img + ((size_t)sy * W + sx) * C + c, cmin, bscale);
                }
                if (sh == H) { v = (long long)hb << AUG_PREC; break; }
                v += (long long)hb * ky.k[r];
            }
            const int ob = sh == H ? (int)(v >> AUG_PREC) : clip8(v);
            out[(size_t)c * H * W] = normalize01((float)ob);
        }
    }
}

}  // namespace

extern "C" size_t gdn_kitti_augment_workspace_bytes(int32_t B) { return (size_t)(B > 0 ? B : 0) * 2 * sizeof(float) + 16; }

extern "C" int gdn_kitti_augment(const void* src, int32_t src_is_f32, int32_t B, int32_t H, int32_t W, int32_t C,
                                 const int32_t* params, int32_t train, float* dst, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4) return GDN_ERR_BAD_ARG;
    if (train && !params) return GDN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* mm = (float*)workspace;
    if (train && src_is_f32) {
        if (!workspace || workspace_bytes < gdn_kitti_augment_workspace_bytes(B)) return GDN_ERR_WORKSPACE;
        hipLaunchKernelGGL(aug_minmax_kernel, dim3(B), dim3(256), 0, st, (const float*)src, (int64_t)H * W * C, mm);
    }
    const int64_t total = (int64_t)B * H * W;
    const int blocks = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(aug_kernel, dim3(blocks), dim3(256), 0, st, src, src_is_f32, B, H, W, C, (const int*)params, train,
                       (const float*)mm, dst);
    return gdn_launch_status();
}
