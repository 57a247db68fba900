// KITTI depth metrics (compute_errors, calculate_error.py:10-103) as one
// workgroup per image: min/max normalisation, Godard crop + validity mask,
// lower-median scaling by an 8-bit radix select over the float bit patterns
// (all values are positive, so bit order == value order), clamp, and the eight
// reductions in fp64.  No host round trips, no sorts, no atomics on global memory.
#include "common.h"

namespace {

#define MT 1024   // threads per image

__device__ __forceinline__ double block_sum_d1024(double v, double* sh) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < MT / 64; ++i) s += sh[i];
    return s;
}

struct PixVals { float g, p, s; bool valid; };

__device__ __forceinline__ PixVals load_px(const float* gs, const float* gt, const float* pr, int i, int W, float gmin,
                                          float gmax, float pmin, float pmax, int y1, int y2, int x1, int x2) {
    PixVals v;
    v.p = ((pr[i] - pmin) / (pmax - pmin)) * 80.f;
    v.g = ((gt[i] - gmin) / (gmax - gmin)) * 80.f;
    v.s = ((gs[i] + 1.0f) / 2.0f) * 80.f;
    const int y = i / W, x = i - y * W;
    v.valid = (v.s < 80.f) && (v.g < 80.f) && (v.s > 1.f) && (v.g > 1.f) && y >= y1 && y < y2 && x >= x1 && x < x2;
    return v;
}

__global__ __launch_bounds__(MT) void depth_metrics_kernel(const float* __restrict__ gt_sparse,
                                                           const float* __restrict__ gt,
                                                           const float* __restrict__ pred, int H, int W, int y1,
                                                           int y2, int x1, int x2, double* __restrict__ per_image) {
    __shared__ double shd[MT / 64];
    __shared__ float shf[4][MT / 64];
    __shared__ int hist[256];
    __shared__ unsigned sel_prefix;
    __shared__ int sel_k;
    const int b = blockIdx.x, n = H * W, tid = threadIdx.x;
    const float* gs = gt_sparse + (size_t)b * n;
    const float* g = gt + (size_t)b * n;
    const float* p = pred + (size_t)b * n;

    // ---- min / max over the whole image (calculate_error.py:38-39) ----
    float gmin = INFINITY, gmax = -INFINITY, pmin = INFINITY, pmax = -INFINITY;
    for (int i = tid; i < n; i += MT) {
        const float a = g[i], c = p[i];
        gmin = fminf(gmin, a); gmax = fmaxf(gmax, a); pmin = fminf(pmin, c); pmax = fmaxf(pmax, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gmin = fminf(gmin, __shfl_xor(gmin, o, 64)); gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
        pmin = fminf(pmin, __shfl_xor(pmin, o, 64)); pmax = fmaxf(pmax, __shfl_xor(pmax, o, 64));
    }
    if ((tid & 63) == 0) { shf[0][tid >> 6] = gmin; shf[1][tid >> 6] = gmax; shf[2][tid >> 6] = pmin; shf[3][tid >> 6] = pmax; }
    __syncthreads();
    for (int i = 0; i < MT / 64; ++i) {
        gmin = fminf(gmin, shf[0][i]); gmax = fmaxf(gmax, shf[1][i]);
        pmin = fminf(pmin, shf[2][i]); pmax = fmaxf(pmax, shf[3][i]);
    }

    // ---- count of valid pixels ----
    int cnt = 0;
    for (int i = tid; i < n; i += MT) cnt += load_px(gs, g, p, i, W, gmin, gmax, pmin, pmax, y1, y2, x1, x2).valid ? 1 : 0;
    const int nvalid = (int)(block_sum_d1024((double)cnt, shd) + 0.5);

    // ---- lower medians (torch.median) of valid gt and valid pred ----
    float med[2] = {0.f, 0.f};
    for (int which = 0; which < 2; ++which) {
        unsigned prefix = 0u, mask = 0u;
        int kk = nvalid > 0 ? (nvalid - 1) / 2 : 0;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            __syncthreads();
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += MT) {
                const PixVals v = load_px(gs, g, p, i, W, gmin, gmax, pmin, pmax, y1, y2, x1, x2);
                if (!v.valid) continue;
                const unsigned bits = __float_as_uint(which == 0 ? v.g : v.p);
                if ((bits & mask) == prefix) atomicAdd(&hist[(bits >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, sel = 255;
                for (int q = 0; q < 256; ++q) {
                    if (cum + hist[q] > kk) { sel = q; break; }
                    cum += hist[q];
                }
                sel_k = kk - cum;
                sel_prefix = prefix | ((unsigned)sel << shift);
            }
            __syncthreads();
            kk = sel_k;
            prefix = sel_prefix;
            mask |= 0xFFu << shift;
        }
        med[which] = __uint_as_float(prefix);
    }

    // ---- the eight reductions over valid pixels ----
    double s_abs = 0, s_rel = 0, s_sq = 0, c1 = 0, c2 = 0, c3 = 0, s_d2 = 0, s_log = 0;
    for (int i = tid; i < n; i += MT) {
        const PixVals v = load_px(gs, g, p, i, W, gmin, gmax, pmin, pmax, y1, y2, x1, x2);
        if (!v.valid) continue;
        float vp = v.p * med[0] / med[1];
        vp = fminf(fmaxf(vp, 1.f), 80.f);
        const float vg = v.g;
        const float th = fmaxf(vg / vp, vp / vg);
        const float d = vg - vp;
        s_abs += (double)fabsf(d);
        s_rel += (double)(fabsf(d) / vg);
        s_sq += (double)((d * d) / vg);
        c1 += th < 1.25f ? 1.0 : 0.0;
        c2 += th < 1.5625f ? 1.0 : 0.0;
        c3 += th < 1.953125f ? 1.0 : 0.0;
        s_d2 += (double)(d * d);
        const float lg = logf(vg) - logf(vp);
        s_log += (double)(lg * lg);
    }
    s_abs = block_sum_d1024(s_abs, shd); s_rel = block_sum_d1024(s_rel, shd); s_sq = block_sum_d1024(s_sq, shd);
    c1 = block_sum_d1024(c1, shd); c2 = block_sum_d1024(c2, shd); c3 = block_sum_d1024(c3, shd);
    s_d2 = block_sum_d1024(s_d2, shd); s_log = block_sum_d1024(s_log, shd);
    if (tid == 0) {
        const double nv = (double)nvalid;
        double* o = per_image + (size_t)b * 8;
        o[0] = s_abs / nv; o[1] = s_rel / nv; o[2] = s_sq / nv;
        o[3] = c1 / nv; o[4] = c2 / nv; o[5] = c3 / nv;
        o[6] = sqrt(s_d2 / nv); o[7] = sqrt(s_log / nv);
    }
}

__global__ void metrics_mean_kernel(const double* __restrict__ per_image, int B, float* __restrict__ errors) {
    const int m = threadIdx.x;
    if (m >= 8) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += per_image[(size_t)b * 8 + m];
    errors[m] = (float)(s / (double)B);
}

}  // namespace

extern "C" size_t gdn_depth_metrics_workspace_bytes(int32_t B, int32_t H, int32_t W) {
    (void)H; (void)W;
    return (size_t)(B > 0 ? B : 0) * 8 * sizeof(double);
}

extern "C" int gdn_depth_metrics(const float* gt_sparse, const float* gt, const float* pred, int32_t B, int32_t H,
                                 int32_t W, int32_t crop, float* errors, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!gt_sparse || !gt || !pred || !errors || B <= 0 || H <= 0 || W <= 0) return GDN_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gdn_depth_metrics_workspace_bytes(B, H, W)) return GDN_ERR_WORKSPACE;
    int y1 = 0, y2 = H, x1 = 0, x2 = W;
    if (crop) {  // Godard crop, calculate_error.py:28-29
        y1 = (int)(0.3324324 * H); y2 = (int)(0.91351351 * H);
        x1 = (int)(0.0359477 * W); x2 = (int)(0.96405229 * W);
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(depth_metrics_kernel, dim3(B), dim3(MT), 0, st, gt_sparse, gt, pred, H, W, y1, y2, x1, x2,
                       (double*)workspace);
    hipLaunchKernelGGL(metrics_mean_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, B, errors);
    return gdn_launch_status();
}
