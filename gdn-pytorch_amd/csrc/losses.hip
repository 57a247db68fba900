// Per-pixel training losses of the GDN hot path, forward value + gradient in
// one pass each, for gfx950.  HBM/latency bound: [B,1,H,W] tensors, wave (64
// lane) shuffles + fixed-order block partials -> bitwise reproducible, and no
// host synchronisation (the reference's boolean-mask indexing costs four D2H
// syncs per step).
#include "common.h"

namespace {

#define LOSS_MAXBLK 1024
#define LOSS_HDR 64

struct LossWs {
    unsigned* maxbits;
    double* part;      // [LOSS_MAXBLK]
    double* part2;     // [LOSS_MAXBLK]
    float* fa;         // [npix]
    float* fb;         // [npix]
};

inline LossWs carve(void* ws, int64_t npix) {
    char* b = (char*)ws;
    LossWs w;
    w.maxbits = (unsigned*)b;
    w.part = (double*)(b + LOSS_HDR);
    w.part2 = w.part + LOSS_MAXBLK;
    w.fa = (float*)(w.part2 + LOSS_MAXBLK);
    w.fb = w.fa + npix;
    return w;
}

inline int loss_blocks(int64_t n) {
    int64_t b = cdiv64(n, 256);
    if (b < 1) b = 1;
    return (int)(b < LOSS_MAXBLK ? b : LOSS_MAXBLK);
}

__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// out = scale * sum(part[0..n)) (+ previous *out when accumulate);  total (nullable) = out + *plus + *plus2: the sum of a
// step's loss terms (trainer.py:456, :757) without a separate add kernel
__global__ __launch_bounds__(256) void finalize_sum_kernel(const double* __restrict__ part, int n, double scale,
                                                           int accumulate, float* out, const float* plus = nullptr,
                                                           const float* plus2 = nullptr, float* total = nullptr) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum_d256(s, sh);
    if (threadIdx.x == 0) {
        const double r = s * scale;
        const float o = accumulate ? (float)((double)*out + r) : (float)r;
        *out = o;
        if (total) *total = (plus ? o + *plus : o) + (plus2 ? *plus2 : 0.f);
    }
}

// ------------------------------------------------------------------ BerHu
__global__ void zero_u32_kernel(unsigned* p) { *p = 0u; }

__global__ __launch_bounds__(256) void absdiff_max_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          int64_t n, unsigned* maxbits) {
    __shared__ float sh[4];
    float m = 0.f;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(a[i] - b[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        atomicMax(maxbits, __float_as_uint(m));   // non-negative floats order like their bit patterns
    }
}

__global__ __launch_bounds__(256) void berhu_kernel(const float* __restrict__ out, const float* __restrict__ gt,
                                                    const float* __restrict__ sparse, int Cs, int B, int H, int W,
                                                    int y1, int y2, int x1, int x2, const unsigned* maxbits,
                                                    double* part, float* __restrict__ dout) {
    __shared__ double sh[4];
    const float c = 0.2f * __uint_as_float(*maxbits);
    const int64_t HW = (int64_t)H * W, n = (int64_t)B * HW;
    const float gsc = 3.0f / (float)n;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = out[i] - gt[i], a = fabsf(d);
        float rho, g;
        if (a > c) { rho = (d * d + c * c) / (2.f * c); g = d / c; }
        else { rho = a; g = sgn(d); }
        float w = 1.f;
        if (sparse) {
            const int64_t b = i / HW, hw = i - b * HW;
            const int y = (int)(hw / W), x = (int)(hw - (int64_t)y * W);
            const bool crop = y >= y1 && y < y2 && x >= x1 && x < x2;
            w = crop ? (sparse[b * Cs * HW + hw] > -1.f ? 1.f : 0.3f) : 0.1f;
        }
        acc += (double)(w * rho);
        if (dout) dout[i] += gsc * w * g;
    }
    acc = block_sum_d256(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// ------------------------------------------------------------------ Sobel L1
// cross-correlation with zero padding, utils.py:107-123
__device__ __forceinline__ float px(const float* __restrict__ im, int y, int x, int H, int W) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? im[(int64_t)y * W + x] : 0.f;
}

__global__ __launch_bounds__(256) void sobel_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        int B, int H, int W, double* part, float* __restrict__ sy,
                                                        float* __restrict__ sx) {
    __shared__ double sh[4];
    const int64_t HW = (int64_t)H * W, n = (int64_t)B * HW;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / HW, hw = i - b * HW;
        const int y = (int)(hw / W), x = (int)(hw - (int64_t)y * W);
        const float* P = pred + b * HW;
        const float* G = gt + b * HW;
        float gyp, gxp, gyg, gxg;
        {
            const float a = px(P, y - 1, x - 1, H, W), bb = px(P, y - 1, x, H, W), cc = px(P, y - 1, x + 1, H, W);
            const float d = px(P, y, x - 1, H, W), f = px(P, y, x + 1, H, W);
            const float g = px(P, y + 1, x - 1, H, W), h = px(P, y + 1, x, H, W), k = px(P, y + 1, x + 1, H, W);
            gxp = (a - cc) + 2.f * (d - f) + (g - k);
            gyp = (a + 2.f * bb + cc) - (g + 2.f * h + k);
        }
        {
            const float a = px(G, y - 1, x - 1, H, W), bb = px(G, y - 1, x, H, W), cc = px(G, y - 1, x + 1, H, W);
            const float d = px(G, y, x - 1, H, W), f = px(G, y, x + 1, H, W);
            const float g = px(G, y + 1, x - 1, H, W), h = px(G, y + 1, x, H, W), k = px(G, y + 1, x + 1, H, W);
            gxg = (a - cc) + 2.f * (d - f) + (g - k);
            gyg = (a + 2.f * bb + cc) - (g + 2.f * h + k);
        }
        const float dy = gyp - gyg, dx = gxp - gxg;
        acc += (double)fabsf(dy) + (double)fabsf(dx);
        sy[i] = sgn(dy);
        sx[i] = sgn(dx);
    }
    acc = block_sum_d256(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// adjoint stencil: pred[y][x] feeds out[y - i + 1][x - j + 1] with weight f[i][j]
__global__ __launch_bounds__(256) void sobel_bwd_kernel(const float* __restrict__ sy, const float* __restrict__ sx,
                                                        int B, int H, int W, float scale, float* __restrict__ dpred) {
    const int64_t HW = (int64_t)H * W, n = (int64_t)B * HW;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / HW, hw = i - b * HW;
        const int y = (int)(hw / W), x = (int)(hw - (int64_t)y * W);
        const float* SY = sy + b * HW;
        const float* SX = sx + b * HW;
        // fy = [[1,2,1],[0,0,0],[-1,-2,-1]]; fx = [[1,0,-1],[2,0,-2],[1,0,-1]]
        float g = 0.f;
        g += 1.f * px(SY, y + 1, x + 1, H, W) + 2.f * px(SY, y + 1, x, H, W) + 1.f * px(SY, y + 1, x - 1, H, W);
        g -= 1.f * px(SY, y - 1, x + 1, H, W) + 2.f * px(SY, y - 1, x, H, W) + 1.f * px(SY, y - 1, x - 1, H, W);
        g += 1.f * px(SX, y + 1, x + 1, H, W) + 2.f * px(SX, y, x + 1, H, W) + 1.f * px(SX, y - 1, x + 1, H, W);
        g -= 1.f * px(SX, y + 1, x - 1, H, W) + 2.f * px(SX, y, x - 1, H, W) + 1.f * px(SX, y - 1, x - 1, H, W);
        dpred[i] += scale * g;
    }
}

// ------------------------------------------------------------------ smoothness
__global__ __launch_bounds__(256) void smooth_kernel(const float* __restrict__ D, const float* __restrict__ I, int Ci,
                                                     int B, int H, int W, double* part, float* __restrict__ dD) {
    __shared__ double sh[4];
    const int64_t HW = (int64_t)H * W, n = (int64_t)B * HW;
    const float gsc = 0.1f / (float)n;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / HW, hw = i - b * HW;
        const int y = (int)(hw / W), x = (int)(hw - (int64_t)y * W);
        const float* d = D + b * HW;
        const float* im = I + b * Ci * HW;
        // weights exp(-mean_c |grad I|) at (y,x), (y,x-1) [x-dir] and (y,x), (y-1,x) [y-dir]
        float sx0 = 0.f, sxm = 0.f, sy0 = 0.f, sym = 0.f;
        for (int c = 0; c < Ci; ++c) {
            const float* ic = im + c * HW;
            const float v = ic[hw];
            if (x < W - 1) sx0 += fabsf(v - ic[hw + 1]);
            if (x >= 1) sxm += fabsf(ic[hw - 1] - v);
            if (y < H - 1) sy0 += fabsf(v - ic[hw + W]);
            if (y >= 1) sym += fabsf(ic[hw - W] - v);
        }
        const float inv = 1.f / (float)Ci;
        const float wx0 = expf(-sx0 * inv), wxm = expf(-sxm * inv);
        const float wy0 = expf(-sy0 * inv), wym = expf(-sym * inv);
        const float v = d[hw];
        const float gx0 = x < W - 1 ? v - d[hw + 1] : 0.f;
        const float gy0 = y < H - 1 ? v - d[hw + W] : 0.f;
        acc += (double)(fabsf(gx0 * wx0) + fabsf(gy0 * wy0));
        if (dD) {
            float g = sgn(gx0) * wx0 + sgn(gy0) * wy0;
            if (x >= 1) g -= sgn(d[hw - 1] - v) * wxm;
            if (y >= 1) g -= sgn(d[hw - W] - v) * wym;
            dD[i] += gsc * g;
        }
    }
    acc = block_sum_d256(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// ------------------------------------------------------------------ MSE
__global__ __launch_bounds__(256) void sqdiff_kernel(const void* __restrict__ a, const void* __restrict__ b,
                                                     int64_t n, double* part, int dt) {
    __shared__ double sh[4];
    double acc = 0.0;
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 d = ld4_any(a, i * 4, dt & 1) - ld4_any(b, i * 4, dt & 2);
        acc += (double)(d[0] * d[0] + d[1] * d[1]) + (double)(d[2] * d[2] + d[3] * d[3]);
    }
    for (int64_t i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = ld1_any(a, i, dt & 1) - ld1_any(b, i, dt & 2);
        acc += (double)(d * d);
    }
    acc = block_sum_d256(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// da = (*gscale) * coef * (a - b): gradient of coef/2 * sum((a-b)^2) scaled by a device-side upstream scalar
__global__ __launch_bounds__(256) void sqdiff_grad_kernel(const void* __restrict__ a, const void* __restrict__ b,
                                                          int64_t n, float coef, const float* __restrict__ gscale,
                                                          void* __restrict__ da, int dt) {
    const float k = coef * (gscale ? *gscale : 1.f);
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        st4_any(da, i * 4, k * (ld4_any(a, i * 4, dt & 1) - ld4_any(b, i * 4, dt & 2)), dt & 4);
    for (int64_t i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        st1_any(da, i, k * (ld1_any(a, i, dt & 1) - ld1_any(b, i, dt & 2)), dt & 4);
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" size_t gdn_loss_workspace_bytes(int64_t npix) {
    return LOSS_HDR + 2 * LOSS_MAXBLK * sizeof(double) + 2 * (size_t)npix * sizeof(float);
}

// max |a - b| over n elements -> *max_out (device float).  A non-negative float's bit pattern orders like the float,
// so the atomicMax on bits IS the float max; exposed so a data-parallel run can all-reduce(MAX) the 4 bytes.
extern "C" int gdn_absdiff_max(const float* a, const float* b, int64_t n, float* max_out, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!a || !b || !max_out || n <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(1), 0, ST(stream), (unsigned*)max_out);
    hipLaunchKernelGGL(absdiff_max_kernel, dim3(loss_blocks(n)), dim3(256), 0, ST(stream), a, b, n, (unsigned*)max_out);
    return gdn_launch_status();
}

extern "C" int gdn_berhu_masked(const float* out, const float* gt, const float* sparse, int32_t Cs, int32_t B, int32_t H,
                                int32_t W, const int32_t box[4], const float* ext_max, float* loss, float* dout,
                                void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!out || !gt || !loss || B <= 0 || H <= 0 || W <= 0) return GDN_ERR_BAD_ARG;
    const int64_t n = (int64_t)B * H * W;
    if (!workspace || workspace_bytes < gdn_loss_workspace_bytes(n)) return GDN_ERR_WORKSPACE;
    LossWs w = carve(workspace, n);
    const int nb = loss_blocks(n);
    int y1 = 0, y2 = H, x1 = 0, x2 = W;
    if (box) { y1 = box[0]; y2 = box[1]; x1 = box[2]; x2 = box[3]; }
    if (!ext_max) {
        hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(1), 0, ST(stream), w.maxbits);
        hipLaunchKernelGGL(absdiff_max_kernel, dim3(nb), dim3(256), 0, ST(stream), out, gt, n, w.maxbits);
    }
    hipLaunchKernelGGL(berhu_kernel, dim3(nb), dim3(256), 0, ST(stream), out, gt, sparse, Cs, B, H, W, y1, y2, x1, x2,
                       ext_max ? (const unsigned*)ext_max : (const unsigned*)w.maxbits, w.part, dout);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)w.part, nb,
                       3.0 / (double)n, 0, loss);
    return gdn_launch_status();
}

extern "C" int gdn_sobel_l1(const float* pred, const float* gt, int32_t B, int32_t H, int32_t W, float weight,
                            float* loss, float* dpred, const float* plus, float* total, void* workspace,
                            size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!pred || !gt || !loss || B <= 0 || H <= 0 || W <= 0) return GDN_ERR_BAD_ARG;
    const int64_t n = (int64_t)B * H * W;
    if (!workspace || workspace_bytes < gdn_loss_workspace_bytes(n)) return GDN_ERR_WORKSPACE;
    LossWs w = carve(workspace, n);
    const int nb = loss_blocks(n);
    hipLaunchKernelGGL(sobel_fwd_kernel, dim3(nb), dim3(256), 0, ST(stream), pred, gt, B, H, W, w.part, w.fa, w.fb);
    if (dpred)
        hipLaunchKernelGGL(sobel_bwd_kernel, dim3(nb), dim3(256), 0, ST(stream), (const float*)w.fa, (const float*)w.fb,
                           B, H, W, weight / (float)n, dpred);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)w.part, nb,
                       (double)weight / (double)n, 0, loss, plus, (const float*)nullptr, total);
    return gdn_launch_status();
}

extern "C" int gdn_smoothness(const float* depth, const float* img, int32_t Ci, int32_t B, int32_t H, int32_t W,
                              float* loss, float* ddepth, const float* plus, const float* plus2, float* total,
                              void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!depth || !img || !loss || Ci <= 0 || B <= 0 || H <= 0 || W <= 0) return GDN_ERR_BAD_ARG;
    const int64_t n = (int64_t)B * H * W;
    if (!workspace || workspace_bytes < gdn_loss_workspace_bytes(n)) return GDN_ERR_WORKSPACE;
    LossWs w = carve(workspace, n);
    const int nb = loss_blocks(n);
    hipLaunchKernelGGL(smooth_kernel, dim3(nb), dim3(256), 0, ST(stream), depth, img, Ci, B, H, W, w.part, ddepth);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)w.part, nb,
                       0.1 / (double)n, 0, loss, plus, plus2, total);
    return gdn_launch_status();
}

extern "C" int gdn_mse(const void* a, const void* b, int64_t n, float weight, int32_t accumulate, float* loss,
                       void* workspace, size_t workspace_bytes, int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!a || !b || !loss || n <= 0) return GDN_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gdn_loss_workspace_bytes(0)) return GDN_ERR_WORKSPACE;
    LossWs w = carve(workspace, 0);
    const int nb = loss_blocks(n / 4 + 1);
    hipLaunchKernelGGL(sqdiff_kernel, dim3(nb), dim3(256), 0, ST(stream), a, b, n, w.part, dtypes);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)w.part, nb,
                       (double)weight / (double)n, accumulate, loss);
    return gdn_launch_status();
}

extern "C" int gdn_mse_grad(const void* a, const void* b, int64_t n, float weight, const float* gscale, void* da,
                            int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!a || !b || !da || n <= 0) return GDN_ERR_BAD_ARG;
    const int nb = loss_blocks(n / 4 + 1);
    hipLaunchKernelGGL(sqdiff_grad_kernel, dim3(nb), dim3(256), 0, ST(stream), a, b, n,
                       (float)(2.0 * (double)weight / (double)n), gscale, da, dtypes);
    return gdn_launch_status();
}
