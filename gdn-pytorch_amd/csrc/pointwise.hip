// HBM-bound glue kernels of the GDN hot path for gfx950: BatchNorm finalize /
// apply / backward, x2 bilinear up-sampling, layout transforms, fused Adam.
// All are 16-byte-per-lane streaming kernels (NHWC, channel count % 4 == 0).
#include "common.h"
#include "up2x.h"

namespace {

// log2 of v if v is a power of two, else -1: the channel-quad counts of these networks (16..128) all are, which turns
// the per-element 64-bit division of the flat index into a shift (the division alone cost more than the memory traffic).
inline int pow2_shift(int v) {
    for (int s = 0; s < 31; ++s)
        if ((1 << s) == v) return s;
    return -1;
}
#define SPLIT_PIX_C(i, cq, sh, pix, c)                                       \
    const int64_t pix = (sh) >= 0 ? ((i) >> (sh)) : (i) / (cq);              \
    const int c = (int)((sh) >= 0 ? ((i) & ((cq) - 1)) : ((i) - pix * (cq))) * 4

inline int stream_blocks(int64_t n_items, int threads = 256, int cap = 2048) {
    int64_t b = cdiv64(n_items, threads);
    if (b < 1) b = 1;
    return (int)(b < cap ? b : cap);
}

// ------------------------------------------------------------------ BN forward
__global__ __launch_bounds__(256) void bn_finalize_train_kernel(
    const float* __restrict__ stats, int64_t slots, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
    float* scale, float* shift, float* mean_out, float* invstd_out, long long* num_batches_tracked) {
    __shared__ double sh[8];
    const int c = blockIdx.x;
    if (num_batches_tracked && c == 0 && threadIdx.x == 0) *num_batches_tracked += 1;   // nn.BatchNorm2d bookkeeping
    double s1 = 0.0, s2 = 0.0;
    for (int64_t s = threadIdx.x; s < slots; s += 256) {
        s1 += (double)stats[(s * 2 + 0) * C + c];
        s2 += (double)stats[(s * 2 + 1) * C + c];
    }
    s1 = block_sum_d256(s1, sh);
    s2 = block_sum_d256(s2, sh + 4);
    if (threadIdx.x == 0) {
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float sc = (float)((double)g * invstd);
        scale[c] = sc;
        shift[c] = (float)((double)b - mean * (double)g * invstd);
        if (mean_out) mean_out[c] = (float)mean;
        if (invstd_out) invstd_out[c] = (float)invstd;
        if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        if (running_var) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float inv = 1.0f / sqrtf(rv[c] + eps);
    scale[c] = g * inv;
    shift[c] = b - rm[c] * g * inv;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ y, int ldy,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const void* __restrict__ res, int ld_res,
                                                       void* __restrict__ out, int ld_out, int64_t npix, int C,
                                                       int relu, int dt, int sh) {
    const int cq = C >> 2;
    const int64_t total = npix * cq;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        SPLIT_PIX_C(i, cq, sh, pix, c);
        const f32x4 v = ld4_any(y, pix * ldy + c, dt & 1);
        const f32x4 s = *reinterpret_cast<const f32x4*>(scale + c);
        const f32x4 t = *reinterpret_cast<const f32x4*>(shift + c);
        f32x4 o = v * s + t;
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        if (res) o += ld4_any(res, pix * ld_res + c, dt & 2);
        st4_any(out, pix * ld_out + c, o, dt & 4);
    }
}

// bf16 tensors: 8 channels per lane (16-byte accesses); same arithmetic
__global__ __launch_bounds__(256) void bn_apply8_kernel(const void* __restrict__ y, int ldy,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const void* __restrict__ res, int ld_res,
                                                        void* __restrict__ out, int ld_out, int64_t npix, int C,
                                                        int relu, int dt, int sh) {
    const int co = C >> 3;
    const int64_t total = npix * co;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = sh >= 0 ? (i >> sh) : i / co;
        const int c = (int)(sh >= 0 ? (i & (co - 1)) : (i - pix * co)) * 8;
        f32x4 v0, v1;
        ld8_any(y, pix * ldy + c, dt & 1, v0, v1);
        f32x4 o0 = v0 * *reinterpret_cast<const f32x4*>(scale + c) + *reinterpret_cast<const f32x4*>(shift + c);
        f32x4 o1 = v1 * *reinterpret_cast<const f32x4*>(scale + c + 4) + *reinterpret_cast<const f32x4*>(shift + c + 4);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = fmaxf(o0[e], 0.f); o1[e] = fmaxf(o1[e], 0.f); }
        }
        if (res) {
            f32x4 r0, r1;
            ld8_any(res, pix * ld_res + c, dt & 2, r0, r1);
            o0 += r0; o1 += r1;
        }
        st8_any(out, pix * ld_out + c, o0, o1, dt & 4);
    }
}

// ----------------------------------------------------------------- BN backward
#define BNB_MAXBLK 1024
// pass 1: partial[blk][2][C] = sum over the block's pixels of dz, dz*xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const void* __restrict__ dout, int ld_dout, const void* __restrict__ y, int ldy,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ partial, int64_t npix, int C, int relu, int dt) {
    __shared__ float sh[256 * 8];
    const int cq = C >> 2;
    const int CQ = cq < 256 ? cq : 256;      // channel-quad lanes
    const int PY = 256 / CQ;                 // pixel lanes
    const int tx = threadIdx.x % CQ, ty = threadIdx.x / CQ;
    const int64_t per = cdiv64(npix, gridDim.x);
    const int64_t p0 = blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    for (int q = tx; q < cq; q += CQ) {
        const int c = q * 4;
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
        if (ty < PY) {
            const f32x4 s = *reinterpret_cast<const f32x4*>(scale + c);
            const f32x4 t = *reinterpret_cast<const f32x4*>(shift + c);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
            const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
            for (int64_t pix = p0 + ty; pix < p1; pix += PY) {
                const f32x4 d = ld4_any(dout, pix * ld_dout + c, dt & 1);
                const f32x4 v = ld4_any(y, pix * ldy + c, dt & 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float dz = d[e];
                    if (relu && !(v[e] * s[e] + t[e] > 0.f)) dz = 0.f;
                    a1[e] += dz;
                    a2[e] += dz * ((v[e] - mu[e]) * is[e]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) { sh[threadIdx.x * 8 + e] = a1[e]; sh[threadIdx.x * 8 + 4 + e] = a2[e]; }
        __syncthreads();
        if (ty == 0) {
            f32x4 r1 = {0.f, 0.f, 0.f, 0.f}, r2 = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < PY; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { r1[e] += sh[(j * CQ + tx) * 8 + e]; r2[e] += sh[(j * CQ + tx) * 8 + 4 + e]; }
            }
            *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 0) * C + c) = r1;
            *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 1) * C + c) = r2;
        }
    }
}

// pass 2: per-channel totals -> dgamma, dbeta, and the two means used by pass 3
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                              double count, float* dgamma, float* dbeta,
                                                              float* k1, float* k2) {
    __shared__ double sh[8];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        s1 += (double)partial[((size_t)b * 2 + 0) * C + c];
        s2 += (double)partial[((size_t)b * 2 + 1) * C + c];
    }
    s1 = block_sum_d256(s1, sh);
    s2 = block_sum_d256(s2, sh + 4);
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
        k1[c] = (float)(s1 / count);
        k2[c] = (float)(s2 / count);
    }
}

// pass 3: dy = gamma*invstd * (dz - mean(dz) - xhat*mean(dz*xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const void* __restrict__ dout, int ld_dout, const void* __restrict__ y, int ldy,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ k1, const float* __restrict__ k2,
    void* __restrict__ dy, int ld_dy, int64_t npix, int C, int relu, int dt, int sh) {
    const int cq = C >> 2;
    const int64_t total = npix * cq;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        SPLIT_PIX_C(i, cq, sh, pix, c);
        const f32x4 d = ld4_any(dout, pix * ld_dout + c, dt & 1);
        const f32x4 v = ld4_any(y, pix * ldy + c, dt & 2);
        const f32x4 s = *reinterpret_cast<const f32x4*>(scale + c);   // gamma*invstd
        const f32x4 t = *reinterpret_cast<const f32x4*>(shift + c);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 a = *reinterpret_cast<const f32x4*>(k1 + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(k2 + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float dz = d[e];
            if (relu && !(v[e] * s[e] + t[e] > 0.f)) dz = 0.f;
            const float xh = (v[e] - mu[e]) * is[e];
            o[e] = s[e] * (dz - a[e] - xh * b[e]);
        }
        st4_any(dy, pix * ld_dy + c, o, dt & 4);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply8_kernel(
    const void* __restrict__ dout, int ld_dout, const void* __restrict__ y, int ldy,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ k1, const float* __restrict__ k2,
    void* __restrict__ dy, int ld_dy, int64_t npix, int C, int relu, int dt, int sh) {
    const int co = C >> 3;
    const int64_t total = npix * co;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = sh >= 0 ? (i >> sh) : i / co;
        const int c = (int)(sh >= 0 ? (i & (co - 1)) : (i - pix * co)) * 8;
        f32x4 d[2], v[2], o[2];
        ld8_any(dout, pix * ld_dout + c, dt & 1, d[0], d[1]);
        ld8_any(y, pix * ldy + c, dt & 2, v[0], v[1]);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 s = *reinterpret_cast<const f32x4*>(scale + c + 4 * hh);
            const f32x4 t = *reinterpret_cast<const f32x4*>(shift + c + 4 * hh);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c + 4 * hh);
            const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c + 4 * hh);
            const f32x4 a = *reinterpret_cast<const f32x4*>(k1 + c + 4 * hh);
            const f32x4 b = *reinterpret_cast<const f32x4*>(k2 + c + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float dz = d[hh][e];
                if (relu && !(v[hh][e] * s[e] + t[e] > 0.f)) dz = 0.f;
                const float xh = (v[hh][e] - mu[e]) * is[e];
                o[hh][e] = s[e] * (dz - a[e] - xh * b[e]);
            }
        }
        st8_any(dy, pix * ld_dy + c, o[0], o[1], dt & 4);
    }
}

// ------------------------------------------------------------------- upsample
// grid.y = output row (b, oy), grid.x covers the Wo * C/4 quads of that row: no per-thread division
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                             int B, int H, int W, int C, int align, int dt, int sh) {
    const int cq = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const int row = blockIdx.y, b = row / Ho, oy = row - b * Ho;
    float ly; int y0, y1;
    up_src(oy, H, align, ly, y0, y1);
    const size_t xb = (size_t)b * H * W * C;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < Wo * cq; j += gridDim.x * 256) {
        const int ox = sh >= 0 ? j >> sh : j / cq;
        const int c = (sh >= 0 ? j & (cq - 1) : j - ox * cq) * 4;
        float lx; int x0, x1;
        up_src(ox, W, align, lx, x0, x1);
        const f32x4 v00 = ld4_any(x, xb + ((size_t)y0 * W + x0) * C + c, dt & 1);
        const f32x4 v01 = ld4_any(x, xb + ((size_t)y0 * W + x1) * C + c, dt & 1);
        const f32x4 v10 = ld4_any(x, xb + ((size_t)y1 * W + x0) * C + c, dt & 1);
        const f32x4 v11 = ld4_any(x, xb + ((size_t)y1 * W + x1) * C + c, dt & 1);
        const f32x4 o = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        st4_any(y, (((size_t)b * Ho + oy) * Wo + ox) * C + c, o, dt & 2);
    }
}

// Round 5 (row N1, bf16 forward half): out = [relu](y * scale + shift) (+ residual) and its x2 bilinear upsampling in ONE pass
// -- a ResidualBlock's output x + bn2(conv2(.)) followed by F.interpolate (AE_model_unet.py:55-57 -> :336-359).  Reads the raw
// convolution output and the residual, writes the block output (low resolution; `low` may be null) and the upsampled tensor; the
// block output is not read back.  Same arithmetic and the same rounding points as bn_apply8_kernel followed by
// upsample2x_fwd_kernel: the four neighbours are rounded to the block output's storage type before they are interpolated, so the
// two forms agree bit for bit.  grid.y = output row (b, oy); the thread at odd (oy, ox) also stores the low-resolution value.
__global__ __launch_bounds__(256) void bn_apply_up2x_kernel(const void* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const void* __restrict__ res,
                                                            void* __restrict__ low, void* __restrict__ up, int B, int H, int W,
                                                            int C, int relu, int align, int dt, int sh) {
    const int cq = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const int row = blockIdx.y, b = row / Ho, oy = row - b * Ho;
    float ly; int y0, y1;
    up_src(oy, H, align, ly, y0, y1);
    const size_t xb = (size_t)b * H * W * C;
    auto act = [&](int yy, int xx, int c, const f32x4& s, const f32x4& t) {
        const size_t at = xb + ((size_t)yy * W + xx) * C + c;
        f32x4 o = ld4_any(y, at, dt & 1) * s + t;
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        if (res) o += ld4_any(res, at, dt & 2);
        if (dt & 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = bf16_h_to_f32(f32_to_bf16_h(o[e]));
        }
        return o;
    };
    for (int j = blockIdx.x * 256 + threadIdx.x; j < Wo * cq; j += gridDim.x * 256) {
        const int ox = sh >= 0 ? j >> sh : j / cq;
        const int c = (sh >= 0 ? j & (cq - 1) : j - ox * cq) * 4;
        float lx; int x0, x1;
        up_src(ox, W, align, lx, x0, x1);
        const f32x4 s = *reinterpret_cast<const f32x4*>(scale + c), t = *reinterpret_cast<const f32x4*>(shift + c);
        const f32x4 v00 = act(y0, x0, c, s, t), v01 = act(y0, x1, c, s, t), v10 = act(y1, x0, c, s, t), v11 = act(y1, x1, c, s, t);
        const f32x4 o = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        st4_any(up, (((size_t)b * Ho + oy) * Wo + ox) * C + c, o, dt & 8);
        if (low && (oy & 1) && (ox & 1)) {
            const int iy = oy >> 1, ix = ox >> 1;      // (floor(src) of an odd output index is its own source pixel, up to rounding)
            const f32x4 a = (y0 == iy && x0 == ix) ? v00 : act(iy, ix, c, s, t);
            st4_any(low, xb + ((size_t)iy * W + ix) * C + c, a, dt & 4);
        }
    }
}

// The same in tiles (C % 8 == 0, C <= 1024): a workgroup owns TXL low-resolution columns x all channels (8 per lane: 16-byte
// accesses on bf16 tensors) and walks RC low-resolution rows, keeping the last three rows of the block output -- already rounded
// to its storage type -- in LDS with one halo column on either side: every element of y / residual is read once per workgroup
// that needs it (halo rows and columns are the only re-reads), the BatchNorm arithmetic runs once per element, and each lane
// stores the 2 x 2 output pixels of its low-resolution pixel as whole 16-byte pieces of full pixel lines.  Same values as the
// kernel above, bit for bit.
__global__ __launch_bounds__(256) void bn_apply_up2x8_kernel(const void* __restrict__ y, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const void* __restrict__ res,
                                                             void* __restrict__ low, void* __restrict__ up, int B, int H, int W,
                                                             int C, int relu, int align, int dt, int TXL, int RC) {
    extern __shared__ float a_s[];                             // [3 rows][TXL + 2 columns][C]
    const int CG = C >> 3, tid = threadIdx.x;
    const int cg = tid % CG, cl = tid / CG;                    // channel group, local column (>= TXL: only halo duty)
    const int xb0 = blockIdx.x * TXL, b = blockIdx.z, r0 = blockIdx.y * RC, r1 = min(H, r0 + RC);
    const int ix = xb0 + cl, c = cg * 8;
    const bool mine = cl < TXL && ix < W;
    const int rowf = (TXL + 2) * C;                            // floats per LDS row
    const size_t xb = (size_t)b * H * W * C;
    auto act = [&](int yy, int xx, int cc, f32x4& o0, f32x4& o1) {
        const size_t at = xb + ((size_t)yy * W + xx) * C + cc;
        f32x4 v0, v1;
        ld8_any(y, at, dt & 1, v0, v1);
        o0 = v0 * *reinterpret_cast<const f32x4*>(scale + cc) + *reinterpret_cast<const f32x4*>(shift + cc);
        o1 = v1 * *reinterpret_cast<const f32x4*>(scale + cc + 4) + *reinterpret_cast<const f32x4*>(shift + cc + 4);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = fmaxf(o0[e], 0.f); o1[e] = fmaxf(o1[e], 0.f); }
        }
        if (res) {
            f32x4 q0, q1;
            ld8_any(res, at, dt & 2, q0, q1);
            o0 += q0; o1 += q1;
        }
        if (dt & 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = bf16_h_to_f32(f32_to_bf16_h(o0[e])); o1[e] = bf16_h_to_f32(f32_to_bf16_h(o1[e])); }
        }
    };
    auto load_row = [&](int iy) {                              // block output row iy, columns xb0 - 1 .. xb0 + TXL -> slot iy % 3
        if (iy < 0 || iy >= H) return;                         // (never referenced: up_src clamps)
        float* dst = a_s + (iy % 3) * rowf;
        f32x4 o0, o1;
        if (mine) {
            act(iy, ix, c, o0, o1);
            *reinterpret_cast<f32x4*>(dst + (cl + 1) * C + c) = o0;
            *reinterpret_cast<f32x4*>(dst + (cl + 1) * C + c + 4) = o1;
            if (low && iy >= r0 && iy < r1) st8_any(low, xb + ((size_t)iy * W + ix) * C + c, o0, o1, dt & 4);
        }
        if (tid < 2 * CG) {                                    // the two halo columns
            const int side = tid / CG, hx = side ? xb0 + TXL : xb0 - 1;
            if (hx >= 0 && hx < W) {
                act(iy, hx, c, o0, o1);
                *reinterpret_cast<f32x4*>(dst + (side ? TXL + 1 : 0) * C + c) = o0;
                *reinterpret_cast<f32x4*>(dst + (side ? TXL + 1 : 0) * C + c + 4) = o1;
            }
        }
    };
    const int Ho = 2 * H, Wo = 2 * W;
    load_row(r0 - 1);
    load_row(r0);
    for (int iy = r0; iy < r1; ++iy) {
        load_row(iy + 1);
        __syncthreads();
        if (mine) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int oy = 2 * iy + dy;
                float ly; int y0, y1;
                up_src(oy, H, align, ly, y0, y1);
                const float* ra = a_s + (y0 % 3) * rowf + c;
                const float* rb = a_s + (y1 % 3) * rowf + c;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int ox = 2 * ix + dx;
                    float lx; int x0, x1;
                    up_src(ox, W, align, lx, x0, x1);
                    const int j0 = (x0 - xb0 + 1) * C, j1 = (x1 - xb0 + 1) * C;
                    f32x4 o[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const f32x4 v00 = *reinterpret_cast<const f32x4*>(ra + j0 + 4 * hh), v01 = *reinterpret_cast<const f32x4*>(ra + j1 + 4 * hh);
                        const f32x4 v10 = *reinterpret_cast<const f32x4*>(rb + j0 + 4 * hh), v11 = *reinterpret_cast<const f32x4*>(rb + j1 + 4 * hh);
                        o[hh] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
                    }
                    st8_any(up, (((size_t)b * Ho + oy) * Wo + ox) * C + c, o[0], o[1], dt & 8);
                }
            }
        }
        __syncthreads();                                       // (the next row overwrites the slot of row iy - 1)
    }
}

// Gather form of the adjoint (deterministic, no atomics): each input pixel sums
// the <= 6x6 output pixels whose stencil touches it.
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const void* __restrict__ dy, void* __restrict__ dx,
                                                             int B, int H, int W, int C, int align, int dt, int sh) {
    const int cq = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const int row = blockIdx.y, b = row / H, iy = row - b * H;
    float wy[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int oy = 2 * iy - 2 + j;
        float l; int a0, a1;
        wy[j] = 0.f;
        if (oy >= 0 && oy < Ho) { up_src(oy, H, align, l, a0, a1); if (a0 == iy) wy[j] += 1.f - l; if (a1 == iy) wy[j] += l; }
    }
    for (int q = blockIdx.x * 256 + threadIdx.x; q < W * cq; q += gridDim.x * 256) {
        const int ix = sh >= 0 ? q >> sh : q / cq;
        const int c = (sh >= 0 ? q & (cq - 1) : q - ix * cq) * 4;
        float wx[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ox = 2 * ix - 2 + j;
            float l; int a0, a1;
            wx[j] = 0.f;
            if (ox >= 0 && ox < Wo) { up_src(ox, W, align, l, a0, a1); if (a0 == ix) wx[j] += 1.f - l; if (a1 == ix) wx[j] += l; }
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jy = 0; jy < 6; ++jy) {
            if (wy[jy] == 0.f) continue;
            const int oy = 2 * iy - 2 + jy;
#pragma unroll
            for (int jx = 0; jx < 6; ++jx) {
                if (wx[jx] == 0.f) continue;
                const int ox = 2 * ix - 2 + jx;
                s += (wy[jy] * wx[jx]) * ld4_any(dy, (((size_t)b * Ho + oy) * Wo + ox) * C + c, dt & 1);
            }
        }
        st4_any(dx, (((size_t)b * H + iy) * W + ix) * C + c, s, dt & 2);
    }
}

// --------------------------------------------------------------------- layouts
__global__ void nchw_to_nhwc_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int C, int HW, int dt) {
    const int64_t total = (int64_t)B * HW * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t t = i / C;
        const int hw = (int)(t % HW);
        const int b = (int)(t / HW);
        st1_any(y, i, ld1_any(x, ((size_t)b * C + c) * HW + hw, dt & 1), dt & 2);
    }
}
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int C, int HW, int dt) {
    const int64_t total = (int64_t)B * HW * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HW);
        const int64_t t = i / HW;
        const int c = (int)(t % C);
        const int b = (int)(t / C);
        st1_any(y, i, ld1_any(x, ((size_t)b * HW + hw) * C + c, dt & 1), dt & 2);
    }
}

// [ntaps][R][C] -> [ntaps][C][R], 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_taps_kernel(const void* __restrict__ w, void* __restrict__ wt,
                                                             int R, int C, int dt) {
    __shared__ float tile[32][33];
    const size_t tb = (size_t)blockIdx.z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = ld1_any(w, tb + (size_t)(r0 + j) * C + c0 + tx, dt & 1);
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) st1_any(wt, tb + (size_t)(c0 + j) * R + r0 + tx, tile[tx][j], dt & 2);
}

// torch [A][Bc][T] <-> tap-major [T][Cout][Cin]
__global__ void weight_tapmajor_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin,
                                       int T, int a_is_cout, int to_tap) {
    const int64_t total = (int64_t)T * Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const int64_t r = i / Cin;
        const int co = (int)(r % Cout);
        const int t = (int)(r / Cout);
        const size_t ti = a_is_cout ? ((size_t)co * Cin + ci) * T + t : ((size_t)ci * Cout + co) * T + t;
        if (to_tap) dst[i] = src[ti];
        else dst[ti] = src[i];
    }
}

// ------------------------------------------------------------------ elementwise
__global__ void add_kernel(const void* __restrict__ a, const void* __restrict__ b, void* __restrict__ o, int64_t n, int dt) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        st4_any(o, i * 4, ld4_any(a, i * 4, dt & 1) + ld4_any(b, i * 4, dt & 2), dt & 4);
    for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        st1_any(o, i, ld1_any(a, i, dt & 1) + ld1_any(b, i, dt & 2), dt & 4);
}
// out[p][c] = a[p][c] (+ b[p][c]); each tensor with its own pixel pitch (channel slices of wider tensors)
__global__ __launch_bounds__(256) void add_pitched_kernel(const void* __restrict__ a, int lda, const void* __restrict__ b, int ldb,
                                                          void* __restrict__ o, int ldo, int64_t npix, int C, int dt, int sh) {
    const int cq = C >> 2;
    const int64_t total = npix * cq;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        SPLIT_PIX_C(i, cq, sh, pix, c);
        f32x4 v = ld4_any(a, pix * lda + c, dt & 1);
        if (b) v += ld4_any(b, pix * ldb + c, dt & 2);
        st4_any(o, pix * ldo + c, v, dt & 4);
    }
}
// out[i] = x[i] * (*s): the chain rule through a scalar loss whose upstream gradient lives on the device
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ o, int64_t n) {
    const float k = *s;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = x[i] * k;
}
__global__ void cast_kernel(const void* __restrict__ a, void* __restrict__ o, int64_t n, int dt) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        st4_any(o, i * 4, ld4_any(a, i * 4, dt & 1), dt & 2);
    for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        st1_any(o, i, ld1_any(a, i, dt & 1), dt & 2);
}
__global__ void tanh_bwd_kernel(const float* __restrict__ d, const float* __restrict__ o, float* __restrict__ r, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        r[i] = d[i] * (1.f - o[i] * o[i]);
}
__global__ void fill_kernel(float* __restrict__ p, float v, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------ Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                                   float gscale) {
    // torch.optim.Adam: g += wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    // p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
    const float step = lr / bc1;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] * gscale + wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi) / bc2s + eps);
    }
}

// Capturable Adam: the step counter and the running powers beta^t live in device memory, so a captured training step
// advances them on every replay.  state = { double beta1^t, double beta2^t, int32 t, float bc1, float bc2s }.
struct AdamDevState { double p1, p2; int step; float bc1, bc2s; };
__global__ void adam_prep_kernel(AdamDevState* st, const float* __restrict__ hyper) {
    st->step += 1;
    st->p1 *= (double)hyper[1];
    st->p2 *= (double)hyper[2];
    st->bc1 = (float)(1.0 - st->p1);
    st->bc2s = (float)sqrt(1.0 - st->p2);
}
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                       const float* __restrict__ hyper, const AdamDevState* st) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], gscale = hyper[5];
    const float step = lr / st->bc1, bc2s = st->bc2s;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] * gscale + wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi) / bc2s + eps);
    }
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int gdn_bn_finalize_train(const float* stats, int64_t slots, int32_t C, int64_t count, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, float momentum,
                                     float eps, float* scale, float* shift, float* mean, float* invstd,
                                     int64_t* num_batches_tracked, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!stats || slots <= 0 || C <= 0 || count <= 0 || !scale || !shift) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(C), dim3(256), 0, ST(stream), stats, slots, C, (double)count,
                       gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd,
                       (long long*)num_batches_tracked);
    return gdn_launch_status();
}

extern "C" int gdn_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int32_t C, float* scale, float* shift,
                                  void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!running_mean || !running_var || C <= 0 || !scale || !shift) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, ST(stream), gamma, beta, running_mean,
                       running_var, eps, C, scale, shift);
    return gdn_launch_status();
}

extern "C" int gdn_bn_apply(const void* y, int32_t ldy, const float* scale, const float* shift, const void* residual,
                            int32_t ld_res, void* out, int32_t ld_out, int64_t npix, int32_t C, int32_t relu,
                            int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!y || !scale || !shift || !out || npix <= 0 || C <= 0) return GDN_ERR_BAD_ARG;
    if ((C % 4) || (ldy % 4) || (ld_out % 4) || (residual && (ld_res % 4))) return GDN_ERR_UNSUPPORTED;
    if (dtypes && (C % 8) == 0 && (ldy % 8) == 0 && (ld_out % 8) == 0 && (!residual || (ld_res % 8) == 0)) {
        hipLaunchKernelGGL(bn_apply8_kernel, dim3(stream_blocks(npix * (C / 8))), dim3(256), 0, ST(stream), y, ldy, scale,
                           shift, residual, ld_res, out, ld_out, npix, C, relu, dtypes, pow2_shift(C / 8));
        return gdn_launch_status();
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_blocks(npix * (C / 4))), dim3(256), 0, ST(stream), y, ldy, scale,
                       shift, residual, ld_res, out, ld_out, npix, C, relu, dtypes, pow2_shift(C / 4));
    return gdn_launch_status();
}

static int bnb_blocks(int64_t npix) {
    int64_t b = cdiv64(npix, 64);
    if (b < 1) b = 1;
    return (int)(b < BNB_MAXBLK ? b : BNB_MAXBLK);
}

extern "C" size_t gdn_bn_bwd_workspace_bytes(int64_t npix, int32_t C) {
    return ((size_t)bnb_blocks(npix) * 2 * C + 2 * (size_t)C) * sizeof(float);
}

extern "C" int gdn_bn_bwd(const void* dout, int32_t ld_dout, const void* y, int32_t ldy, const float* gamma,
                          const float* scale, const float* shift, const float* mean, const float* invstd, void* dy,
                          int32_t ld_dy, float* dgamma, float* dbeta, int64_t npix, int32_t C, int32_t relu,
                          const float* ext_partial, int64_t ext_slots,
                          void* workspace, size_t workspace_bytes, int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    (void)gamma;
    if (!dout || !y || !scale || !shift || !mean || !invstd || !dy || npix <= 0 || C <= 0) return GDN_ERR_BAD_ARG;
    if (ext_partial && (ext_slots <= 0 || ext_slots > 0x7fffffff)) return GDN_ERR_BAD_ARG;
    if ((C % 4) || (ldy % 4) || (ld_dout % 4) || (ld_dy % 4)) return GDN_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < gdn_bn_bwd_workspace_bytes(npix, C)) return GDN_ERR_WORKSPACE;
    const int nblk = bnb_blocks(npix);
    float* partial = (float*)workspace;
    float* k1 = partial + (size_t)nblk * 2 * C;
    float* k2 = k1 + C;
    if (ext_partial) {
        // pass 1 already happened in the epilogue of the kernel that produced dout (gdn_winoconv_bwd / gdn_fftconv_bwd)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, ST(stream), ext_partial, (int)ext_slots, C,
                           (double)npix, dgamma, dbeta, k1, k2);
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, ST(stream), dout, ld_dout, y, ldy, scale, shift,
                           mean, invstd, partial, npix, C, relu, dtypes);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, ST(stream), (const float*)partial, nblk, C,
                           (double)npix, dgamma, dbeta, k1, k2);
    }
    if (dtypes && (C % 8) == 0 && (ldy % 8) == 0 && (ld_dout % 8) == 0 && (ld_dy % 8) == 0)
        hipLaunchKernelGGL(bn_bwd_apply8_kernel, dim3(stream_blocks(npix * (C / 8))), dim3(256), 0, ST(stream), dout,
                           ld_dout, y, ldy, scale, shift, mean, invstd, (const float*)k1, (const float*)k2, dy, ld_dy,
                           npix, C, relu, dtypes, pow2_shift(C / 8));
    else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_blocks(npix * (C / 4))), dim3(256), 0, ST(stream), dout,
                       ld_dout, y, ldy, scale, shift, mean, invstd, (const float*)k1, (const float*)k2, dy, ld_dy, npix,
                       C, relu, dtypes, pow2_shift(C / 4));
    return gdn_launch_status();
}

// Passes 1 + 2 only: dgamma, dbeta and kk = {k1 = mean(dz), k2 = mean(dz*xhat)}[C] for a consumer that applies pass 3 itself while
// loading (gdn_fftconv_bwd's dyb_* arguments).  ext_partial as in gdn_bn_bwd.
extern "C" int gdn_bn_bwd_coeffs(const void* dout, int32_t ld_dout, const void* y, int32_t ldy, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                 float* kk, int64_t npix, int32_t C, int32_t relu, const float* ext_partial,
                                 int64_t ext_slots, void* workspace, size_t workspace_bytes, int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!scale || !shift || !mean || !invstd || !kk || npix <= 0 || C <= 0) return GDN_ERR_BAD_ARG;
    if (ext_partial && (ext_slots <= 0 || ext_slots > 0x7fffffff)) return GDN_ERR_BAD_ARG;
    if (ext_partial) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, ST(stream), ext_partial, (int)ext_slots, C,
                           (double)npix, dgamma, dbeta, kk, kk + C);
        return gdn_launch_status();
    }
    if (!dout || !y) return GDN_ERR_BAD_ARG;
    if ((C % 4) || (ldy % 4) || (ld_dout % 4)) return GDN_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < gdn_bn_bwd_workspace_bytes(npix, C)) return GDN_ERR_WORKSPACE;
    const int nblk = bnb_blocks(npix);
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, ST(stream), dout, ld_dout, y, ldy, scale, shift,
                       mean, invstd, partial, npix, C, relu, dtypes);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, ST(stream), (const float*)partial, nblk, C,
                       (double)npix, dgamma, dbeta, kk, kk + C);
    return gdn_launch_status();
}

// Backward of out = [relu](y*scale + shift) with FIXED per-channel coefficients (eval-mode BatchNorm of a frozen
// network): dy = scale * dout * [z > 0].  No reductions, no parameter gradients.
__global__ __launch_bounds__(256) void bn_eval_bwd_kernel(const void* __restrict__ dout, int ld_dout,
                                                          const void* __restrict__ y, int ldy,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, void* __restrict__ dy,
                                                          int ld_dy, int64_t npix, int C, int relu, int dt, int sh) {
    const int cq = C >> 2;
    const int64_t total = npix * cq;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        SPLIT_PIX_C(i, cq, sh, pix, c);
        const f32x4 d = ld4_any(dout, pix * ld_dout + c, dt & 1);
        const f32x4 s = *reinterpret_cast<const f32x4*>(scale + c);
        f32x4 o = d * s;
        if (relu) {
            const f32x4 v = ld4_any(y, pix * ldy + c, dt & 2);
            const f32x4 t = *reinterpret_cast<const f32x4*>(shift + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // relu == 2: y already IS the activated output (conv + BN + ReLU fused in the conv epilogue)
                const bool on = relu == 2 ? v[e] > 0.f : v[e] * s[e] + t[e] > 0.f;
                if (!on) o[e] = 0.f;
            }
        }
        st4_any(dy, pix * ld_dy + c, o, dt & 4);
    }
}

extern "C" int gdn_bn_eval_bwd(const void* dout, int32_t ld_dout, const void* y, int32_t ldy, const float* scale,
                               const float* shift, void* dy, int32_t ld_dy, int64_t npix, int32_t C, int32_t relu,
                               int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!dout || !y || !scale || !shift || !dy || npix <= 0 || C <= 0) return GDN_ERR_BAD_ARG;
    if ((C % 4) || (ldy % 4) || (ld_dout % 4) || (ld_dy % 4)) return GDN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_eval_bwd_kernel, dim3(stream_blocks(npix * (C / 4))), dim3(256), 0, ST(stream), dout, ld_dout,
                       y, ldy, scale, shift, dy, ld_dy, npix, C, relu, dtypes, pow2_shift(C / 4));
    return gdn_launch_status();
}

extern "C" int gdn_upsample2x_fwd(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                                  int32_t align_corners, int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return GDN_ERR_BAD_ARG;
    if (C % 4) return GDN_ERR_UNSUPPORTED;
    if ((int64_t)B * 2 * H > 65535) return GDN_ERR_UNSUPPORTED;      // grid.y limit
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(cdiv(2 * W * (C / 4), 256), B * 2 * H), dim3(256), 0,
                       ST(stream), x, y, B, H, W, C, align_corners, dtypes, pow2_shift(C / 4));
    return gdn_launch_status();
}

extern "C" int gdn_bn_apply_up2x(const void* y, const float* scale, const float* shift, const void* residual, void* low,
                                 void* up, int32_t B, int32_t H, int32_t W, int32_t C, int32_t relu, int32_t align_corners,
                                 int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!y || !scale || !shift || !up || B <= 0 || H <= 0 || W <= 0 || C <= 0) return GDN_ERR_BAD_ARG;
    if (C % 4) return GDN_ERR_UNSUPPORTED;
    if ((int64_t)B * 2 * H > 65535) return GDN_ERR_UNSUPPORTED;      // grid.y limit
    if ((C % 8) == 0 && C <= 1024 && B <= 65535) {
        // tiles: TXL low-resolution columns x all channels per workgroup, RC rows per workgroup (more where the tensor is large
        // enough to still give every CU several workgroups: a workgroup re-reads one halo row above and below its range)
        const int TXL = 256 / (C / 8), nbx = cdiv(W, TXL);
        int RC = 16;
        while (RC > 2 && (int64_t)B * nbx * cdiv(H, RC) < 2048) RC >>= 1;
        const size_t lds = (size_t)3 * (TXL + 2) * C * sizeof(float);
        hipLaunchKernelGGL(bn_apply_up2x8_kernel, dim3(nbx, cdiv(H, RC), B), dim3(256), lds, ST(stream), y, scale, shift, residual,
                           low, up, B, H, W, C, relu, align_corners, dtypes, TXL, RC);
        return gdn_launch_status();
    }
    hipLaunchKernelGGL(bn_apply_up2x_kernel, dim3(cdiv(2 * W * (C / 4), 256), B * 2 * H), dim3(256), 0, ST(stream), y, scale,
                       shift, residual, low, up, B, H, W, C, relu, align_corners, dtypes, pow2_shift(C / 4));
    return gdn_launch_status();
}

extern "C" int gdn_upsample2x_bwd(const void* dy, void* dx, int32_t B, int32_t H, int32_t W, int32_t C,
                                  int32_t align_corners, int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return GDN_ERR_BAD_ARG;
    if (C % 4) return GDN_ERR_UNSUPPORTED;
    if ((int64_t)B * H > 65535) return GDN_ERR_UNSUPPORTED;          // grid.y limit
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(cdiv(W * (C / 4), 256), B * H), dim3(256), 0,
                       ST(stream), dy, dx, B, H, W, C, align_corners, dtypes, pow2_shift(C / 4));
    return gdn_launch_status();
}

extern "C" int gdn_nchw_to_nhwc(const void* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtypes,
                                void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(stream_blocks((int64_t)B * C * H * W)), dim3(256), 0, ST(stream), x, y,
                       B, C, H * W, dtypes);
    return gdn_launch_status();
}
extern "C" int gdn_nhwc_to_nchw(const void* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtypes,
                                void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(stream_blocks((int64_t)B * C * H * W)), dim3(256), 0, ST(stream), x, y,
                       B, C, H * W, dtypes);
    return gdn_launch_status();
}

extern "C" int gdn_transpose_taps(const void* w, void* wt, int32_t ntaps, int32_t R, int32_t C, int32_t dtypes,
                                  void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!w || !wt || ntaps <= 0 || R <= 0 || C <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(transpose_taps_kernel, dim3(cdiv(C, 32), cdiv(R, 32), ntaps), dim3(256), 0, ST(stream), w, wt, R,
                       C, dtypes);
    return gdn_launch_status();
}

extern "C" int gdn_weight_to_tapmajor(const float* w_torch, float* w_tap, int32_t Cout, int32_t Cin, int32_t ntaps,
                                      int32_t a_is_cout, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!w_torch || !w_tap || Cout <= 0 || Cin <= 0 || ntaps <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(weight_tapmajor_kernel, dim3(stream_blocks((int64_t)Cout * Cin * ntaps)), dim3(256), 0,
                       ST(stream), w_torch, w_tap, Cout, Cin, ntaps, a_is_cout, 1);
    return gdn_launch_status();
}
extern "C" int gdn_weight_from_tapmajor(const float* w_tap, float* w_torch, int32_t Cout, int32_t Cin, int32_t ntaps,
                                        int32_t a_is_cout, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!w_torch || !w_tap || Cout <= 0 || Cin <= 0 || ntaps <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(weight_tapmajor_kernel, dim3(stream_blocks((int64_t)Cout * Cin * ntaps)), dim3(256), 0,
                       ST(stream), w_tap, w_torch, Cout, Cin, ntaps, a_is_cout, 0);
    return gdn_launch_status();
}

extern "C" int gdn_add(const void* a, const void* b, void* out, int64_t n, int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!a || !b || !out || n <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(add_kernel, dim3(stream_blocks(n / 4 + 1)), dim3(256), 0, ST(stream), a, b, out, n, dtypes);
    return gdn_launch_status();
}
extern "C" int gdn_add_pitched(const void* a, int32_t lda, const void* b, int32_t ldb, void* out, int32_t ld_out,
                               int64_t npix, int32_t C, int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!a || !out || npix <= 0 || C <= 0) return GDN_ERR_BAD_ARG;
    if ((C % 4) || (lda % 4) || (ld_out % 4) || (b && (ldb % 4))) return GDN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(add_pitched_kernel, dim3(stream_blocks(npix * (C / 4))), dim3(256), 0, ST(stream), a, lda, b, ldb, out,
                       ld_out, npix, C, dtypes, pow2_shift(C / 4));
    return gdn_launch_status();
}

extern "C" int gdn_scale_dev(const float* x, const float* s, float* out, int64_t n, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!x || !s || !out || n <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(scale_dev_kernel, dim3(stream_blocks(n)), dim3(256), 0, ST(stream), x, s, out, n);
    return gdn_launch_status();
}

extern "C" int gdn_cast(const void* src, void* dst, int64_t n, int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!src || !dst || n <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(cast_kernel, dim3(stream_blocks(n / 4 + 1, 256, 4096)), dim3(256), 0, ST(stream), src, dst, n, dtypes);
    return gdn_launch_status();
}
extern "C" int gdn_tanh_bwd(const float* dout, const float* out, float* dpre, int64_t n, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!dout || !out || !dpre || n <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, ST(stream), dout, out, dpre, n);
    return gdn_launch_status();
}
extern "C" int gdn_fill(float* p, float value, int64_t n, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!p || n <= 0) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(fill_kernel, dim3(stream_blocks(n)), dim3(256), 0, ST(stream), p, value, n);
    return gdn_launch_status();
}

extern "C" int gdn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int32_t step, float grad_scale, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!p || !g || !m || !v || n <= 0 || step < 1) return GDN_ERR_BAD_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(stream_blocks(n, 256, 4096)), dim3(256), 0, ST(stream), p, g, m, v, n, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    return gdn_launch_status();
}

extern "C" int gdn_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, void* state,
                                 void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!p || !g || !m || !v || n <= 0 || !hyper || !state) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, ST(stream), (AdamDevState*)state, hyper);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(stream_blocks(n, 256, 4096)), dim3(256), 0, ST(stream), p, g, m, v, n, hyper,
                       (const AdamDevState*)state);
    return gdn_launch_status();
}

// ---- shader-clock probe (measurement aid; include/gdn_hip.h) -----------------------------------------------------------------
// One wave: reads the shader cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime), sleeps until the flag
// word is set by clock_probe_set_kernel on the measured stream (or max_ticks have passed), reads both again.
// buf must be HOST-PINNED memory mapped into the device (hipHostMalloc / a pinned torch tensor): the setter and the watcher
// usually run on different XCDs, whose L2s are not coherent with each other for ordinary device memory inside a kernel -- an
// agent-scope load kept hitting the watcher's own stale line (the r06 full-suite run: the watcher ran into its tick limit) and
// polling with L2 invalidates would disturb the kernels being measured.  Host memory is uncached: every poll reads the flag.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ buf, unsigned long long max_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (__hip_atomic_load(buf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0ull && r1 - r0 < max_ticks) {
        __builtin_amdgcn_s_sleep(64);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    r1 = __builtin_amdgcn_s_memrealtime();
    buf[1] = c1 - c0;
    buf[2] = r1 - r0;
    buf[3] = __hip_atomic_load(buf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // 1: ended by the flag, 0: by the tick limit
}
__global__ void clock_probe_set_kernel(unsigned long long* buf, unsigned long long v) {
    __hip_atomic_store(buf, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v == 0ull) { buf[1] = 0ull; buf[2] = 0ull; buf[3] = 0ull; }
}

extern "C" int gdn_clock_probe_arm(uint64_t* buf, void* stream) {
    (void)hipGetLastError();
    if (!buf) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(clock_probe_set_kernel, dim3(1), dim3(1), 0, ST(stream), (unsigned long long*)buf, 0ull);
    return gdn_launch_status();
}
extern "C" int gdn_clock_probe_watch(uint64_t* buf, uint64_t max_ticks, void* side_stream) {
    (void)hipGetLastError();
    if (!buf || max_ticks == 0 || max_ticks > 1000000000ull) return GDN_ERR_BAD_ARG;      // at most 10 s
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, ST(side_stream), (unsigned long long*)buf,
                       (unsigned long long)max_ticks);
    return gdn_launch_status();
}
extern "C" int gdn_clock_probe_stop(uint64_t* buf, void* stream) {
    (void)hipGetLastError();
    if (!buf) return GDN_ERR_BAD_ARG;
    hipLaunchKernelGGL(clock_probe_set_kernel, dim3(1), dim3(1), 0, ST(stream), (unsigned long long*)buf, 1ull);
    return gdn_launch_status();
}

extern "C" int gdn_version(void) { return 222; }

extern "C" const char* gdn_strerror(int status) {
    switch (status) {
        case GDN_OK: return "ok";
        case GDN_ERR_BAD_ARG: return "bad argument";
        case GDN_ERR_UNSUPPORTED: return "unsupported shape/alignment";
        case GDN_ERR_WORKSPACE: return "workspace missing or too small";
        case GDN_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

extern "C" int gdn_device_info(char* name, int name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GDN_ERR_LAUNCH;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return GDN_ERR_LAUNCH;
    if (name && name_len > 0) {
        int i = 0;
        for (; i < name_len - 1 && prop.gcnArchName[i]; ++i) name[i] = prop.gcnArchName[i];
        name[i] = 0;
    }
    return prop.multiProcessorCount;
}
