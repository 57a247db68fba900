// x2 bilinear upsampling (F.interpolate(scale_factor=2, mode='bilinear'), AE_model_unet.py:336-359 / :200-230) folded into
// its neighbours, so the upsampled tensor -- 4x the size of its source -- is never written or read:
//   forward   the consumer convolution's patch loader (fft2d_fwd_kernel, wino_input_kernel) interpolates the four
//             low-resolution neighbours of every element it gathers (border rule of the convolution applied first, on
//             the upsampled coordinates, exactly as ReflectionPad2d / zero padding would see the materialised tensor);
//   backward  the kernel that folds a reflection layer's padded-domain data gradient back onto the image
//             (reflect_fold_up2x_kernel below, replacing fft_reflect_fold_kernel / wino_reflect_fold_kernel) also applies
//             the adjoint of the interpolation in gather form: one pass, deterministic, writes the LOW-resolution gradient.
// The arithmetic is the stand-alone kernels' (pointwise.hip: upsample2x_fwd_kernel / upsample2x_bwd_kernel), which stay
// as the path for consumers without such a loader (direct / bf16 kernels).
#pragma once
#include "common.h"

namespace {

// source coordinate of output index o along one axis (extent `in` -> 2 * in): neighbours i0 <= i1 and the weight of i1
__device__ __forceinline__ void up_src(int o, int in, int align, float& l1, int& i0, int& i1) {
    float src;
    if (align) {
        const float sc = in > 1 ? (float)(in - 1) / (float)(2 * in - 1) : 0.f;
        src = sc * o;
    } else {
        src = 0.5f * (o + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - i0;
}

// value of the upsampled image at a column whose neighbours are (x0, x1, lx), rows r0 / r1 (pointers to the two source rows
// at this thread's channel), row weight ly: same association as upsample2x_fwd_kernel
__device__ __forceinline__ float up2x_at(const float* __restrict__ r0, const float* __restrict__ r1, int ld, float ly,
                                         int x0, int x1, float lx) {
    const float v00 = r0[(size_t)x0 * ld], v01 = r0[(size_t)x1 * ld], v10 = r1[(size_t)x0 * ld], v11 = r1[(size_t)x1 * ld];
    return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

// dx[b][iy][ix] (extent H/2 x W/2) = sum over the upsampled pixels (oy, ox) whose stencil touches (iy, ix) of
// weight * F(oy, ox), F = the padded-domain gradient dxp [B][H+2p][W+2p][C] summed over the padded coordinates that reflect
// onto (oy, ox)  (+ addsrc, low resolution).  H, W: the upsampled extent = the convolution's input extent.
// bf16: all three tensors hold bfloat16 (the bf16 layers' data gradient, round 5: gdn_conv_dgrad dx_up2x); sums in fp32.
__global__ __launch_bounds__(256) void reflect_fold_up2x_kernel(const void* __restrict__ dxp, void* __restrict__ dx, int ldx,
                                                                const void* __restrict__ addsrc, int ld_add,
                                                                int B, int H, int W, int C, int p, int align, int bf16) {
    const int Hp = H + 2 * p, Wp = W + 2 * p, c4n = C / 4, Hl = H / 2, Wl = W / 2;
    const int64_t total = (int64_t)B * Hl * Wl * c4n;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int ix = (int)(t % Wl); t /= Wl;
        const int iy = (int)(t % Hl), b = (int)(t / Hl);
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        for (int jy = 0; jy < 6; ++jy) {
            const int oy = 2 * iy - 2 + jy;
            if (oy < 0 || oy >= H) continue;
            float l; int a0, a1;
            up_src(oy, Hl, align, l, a0, a1);
            const float wy = (a0 == iy ? 1.f - l : 0.f) + (a1 == iy ? l : 0.f);
            if (wy == 0.f) continue;
            int qy[3], ny = 0;
            qy[ny++] = oy + p;
            if (oy >= 1 && oy <= p) qy[ny++] = p - oy;
            if (oy <= H - 2 && oy >= H - 1 - p) qy[ny++] = 2 * (H - 1) - oy + p;
            for (int jx = 0; jx < 6; ++jx) {
                const int ox = 2 * ix - 2 + jx;
                if (ox < 0 || ox >= W) continue;
                up_src(ox, Wl, align, l, a0, a1);
                const float wx = (a0 == ix ? 1.f - l : 0.f) + (a1 == ix ? l : 0.f);
                if (wx == 0.f) continue;
                int qx[3], nx = 0;
                qx[nx++] = ox + p;
                if (ox >= 1 && ox <= p) qx[nx++] = p - ox;
                if (ox <= W - 2 && ox >= W - 1 - p) qx[nx++] = 2 * (W - 1) - ox + p;
                f32x4 f4 = {0.f, 0.f, 0.f, 0.f};
                for (int a = 0; a < ny; ++a)
                    for (int e = 0; e < nx; ++e)
                        f4 += ld4_any(dxp, ((size_t)(b * Hp + qy[a]) * Wp + qx[e]) * C + c4 * 4, bf16);
                s4 += (wy * wx) * f4;
            }
        }
        const size_t op = (size_t)(b * Hl + iy) * Wl + ix;
        if (addsrc) s4 += ld4_any(addsrc, op * ld_add + c4 * 4, bf16);
        st4_any(dx, op * ldx + c4 * 4, s4, bf16);
    }
}

// The same with 8 channels per lane (16-byte accesses on bf16 tensors) and the interpolation weights of the 6 candidate rows /
// columns computed once per lane instead of once per candidate pair.  C % 8 == 0, pitches multiples of 8.
__global__ __launch_bounds__(256) void reflect_fold_up2x8_kernel(const void* __restrict__ dxp, void* __restrict__ dx, int ldx,
                                                                 const void* __restrict__ addsrc, int ld_add,
                                                                 int B, int H, int W, int C, int p, int align, int bf16) {
    const int Hp = H + 2 * p, Wp = W + 2 * p, c8n = C / 8, Hl = H / 2, Wl = W / 2;
    const int64_t total = (int64_t)B * Hl * Wl * c8n;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        int64_t t = i / c8n;
        const int ix = (int)(t % Wl); t /= Wl;
        const int iy = (int)(t % Hl), b = (int)(t / Hl);
        float wy[6], wx[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float l; int a0, a1;
            const int oy = 2 * iy - 2 + j, ox = 2 * ix - 2 + j;
            wy[j] = wx[j] = 0.f;
            if (oy >= 0 && oy < H) { up_src(oy, Hl, align, l, a0, a1); wy[j] = (a0 == iy ? 1.f - l : 0.f) + (a1 == iy ? l : 0.f); }
            if (ox >= 0 && ox < W) { up_src(ox, Wl, align, l, a0, a1); wx[j] = (a0 == ix ? 1.f - l : 0.f) + (a1 == ix ? l : 0.f); }
        }
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jy = 0; jy < 6; ++jy) {
            if (wy[jy] == 0.f) continue;
            const int oy = 2 * iy - 2 + jy;
            int qy[3], ny = 0;
            qy[ny++] = oy + p;
            if (oy >= 1 && oy <= p) qy[ny++] = p - oy;
            if (oy <= H - 2 && oy >= H - 1 - p) qy[ny++] = 2 * (H - 1) - oy + p;
#pragma unroll
            for (int jx = 0; jx < 6; ++jx) {
                if (wx[jx] == 0.f) continue;
                const int ox = 2 * ix - 2 + jx;
                int qx[3], nx = 0;
                qx[nx++] = ox + p;
                if (ox >= 1 && ox <= p) qx[nx++] = p - ox;
                if (ox <= W - 2 && ox >= W - 1 - p) qx[nx++] = 2 * (W - 1) - ox + p;
                f32x4 f0 = {0.f, 0.f, 0.f, 0.f}, f1 = {0.f, 0.f, 0.f, 0.f};
                for (int a = 0; a < ny; ++a)
                    for (int e = 0; e < nx; ++e) {
                        f32x4 g0, g1;
                        ld8_any(dxp, ((size_t)(b * Hp + qy[a]) * Wp + qx[e]) * C + c8 * 8, bf16, g0, g1);
                        f0 += g0; f1 += g1;
                    }
                s0 += (wy[jy] * wx[jx]) * f0;
                s1 += (wy[jy] * wx[jx]) * f1;
            }
        }
        const size_t op = (size_t)(b * Hl + iy) * Wl + ix;
        if (addsrc) {
            f32x4 g0, g1;
            ld8_any(addsrc, op * ld_add + c8 * 8, bf16, g0, g1);
            s0 += g0; s1 += g1;
        }
        st8_any(dx, op * ldx + c8 * 8, s0, s1, bf16);
    }
}

}  // namespace
