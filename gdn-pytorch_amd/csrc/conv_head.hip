// 1-channel output head: Conv2d(64->1, k9, p4) (AE_model_unet.py:300,130) and
// ConvTranspose2d(64->1, k9, s1, p4) (:521) + tanh, for gfx950.
//
// With one output channel the MFMA has nothing to amortise (an N=1 GEMM wastes 31/32 of a 32x32 tile), so this is a
// register-blocked VALU kernel on gfx950's packed fp32 pipe: a workgroup owns a 32x64 pixel tile, stages the
// (32+8)x(64+8) input patch in LDS four channels at a time (one float4 per pixel, column-swizzled so that the 16-byte
// reads of neighbouring threads are contiguous), and each thread produces a 2x4 pixel block: every patch row it reads (12
// float4) feeds the filter row of BOTH output rows that touch it (10 row reads for 2x4x81x4 MACs: 15 LDS reads per pixel and
// channel group instead of 27), and the four channels of a tap are two v_pk_fma_f32 on an (even, odd) accumulator pair
// instead of four v_fma.  Weights are wave-uniform (scalar loads).  HBM traffic: the input once (272 MB at B=20) + 4 B
// per pixel out.  Mixed-precision path: x may hold bf16 (weights and the depth map stay fp32).
#include "common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

#define HD_TH 32
#define HD_TW 64
#define HD_PX 4
#define HD_PY 2

template <int K>
__global__ __launch_bounds__(256) void conv_head_kernel(const void* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        float* __restrict__ y, int B, int H, int W, int C, int pad,
                                                        int flip, int act, int tiles_x, int tiles_y, int x_bf16) {
    constexpr int PH = HD_TH + K - 1, PW = HD_TW + K - 1;
    constexpr int PWQ = (PW + 3) / 4;                 // swizzled row: [r = col%4][q = col/4]
    __shared__ f32x4 patch[PH * PWQ * 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    int bid = blockIdx.x;
    const int tix = bid % tiles_x; bid /= tiles_x;
    const int tiy = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y0 = tiy * HD_TH, x0 = tix * HD_TW;
    const size_t xb = (size_t)b * H * W * ldx;

    v2f acc[HD_PY][HD_PX];
#pragma unroll
    for (int r = 0; r < HD_PY; ++r)
#pragma unroll
        for (int p = 0; p < HD_PX; ++p) acc[r][p] = v2f{0.f, 0.f};
    for (int c0 = 0; c0 < C; c0 += 4) {
        __syncthreads();
        for (int idx = tid; idx < PH * PW; idx += 256) {
            const int py = idx / PW, px = idx - py * PW;
            const int iy = y0 - pad + py, ix = x0 - pad + px;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = ld4_any(x, xb + ((size_t)iy * W + ix) * ldx + c0, x_bf16);
            patch[(py * 4 + (px & 3)) * PWQ + (px >> 2)] = v;
        }
        __syncthreads();
#pragma unroll
        for (int iy = 0; iy < K + HD_PY - 1; ++iy) {            // patch row 2 ty + iy feeds output row r with ky = iy - r
            f32x4 xv[HD_PX + K - 1];
            const int rowb = (ty * HD_PY + iy) * 4;
#pragma unroll
            for (int j = 0; j < HD_PX + K - 1; ++j) xv[j] = patch[(rowb + (j & 3)) * PWQ + tx + (j >> 2)];
#pragma unroll
            for (int r = 0; r < HD_PY; ++r) {
                const int ky = iy - r;
                if (ky < 0 || ky >= K) continue;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int tap = flip ? (K - 1 - ky) * K + (K - 1 - kx) : ky * K + kx;
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)tap * C + c0);   // wave-uniform
                    const v2f w01 = v2f{wv[0], wv[1]}, w23 = v2f{wv[2], wv[3]};
#pragma unroll
                    for (int p = 0; p < HD_PX; ++p) {
                        const f32x4 a = xv[p + kx];
                        acc[r][p] += v2f{a[0], a[1]} * w01;
                        acc[r][p] += v2f{a[2], a[3]} * w23;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < HD_PY; ++r) {
        const int oy = y0 + ty * HD_PY + r;
        if (oy >= H) continue;
#pragma unroll
        for (int p = 0; p < HD_PX; ++p) {
            const int ox = x0 + tx * HD_PX + p;
            if (ox < W) {
                float v = acc[r][p].x + acc[r][p].y;
                if (act == GDN_ACT_TANH) v = tanhf(v);
                y[((size_t)b * H + oy) * W + ox] = v;
            }
        }
    }
}

}  // namespace

// Returns GDN_ERR_UNSUPPORTED when the geometry is not a 1-channel stride-1 head this kernel covers.
int gdn_conv_head_fwd(const gdn_conv_geom* g, const void* x, int32_t ldx, const float* w, float* y, int32_t ldy,
                      int32_t act, int32_t x_bf16, void* stream) {
    if (g->Cout != 1 || g->stride != 1 || g->k != 9 || g->pad_mode != 0 || (g->Cin % 4) || (ldx % 4) || ldy != 1)
        return GDN_ERR_UNSUPPORTED;
    // Conv2d: out[o] = sum x[o - p + k] w[k];  ConvTranspose2d (s=1): out[o] = sum x[o + p - k] w[k]
    //  == a correlation with the flipped kernel and padding k-1-p.
    const int pad = g->transposed ? g->k - 1 - g->pad : g->pad;
    const int Ho = g->H + 2 * pad - g->k + 1, Wo = g->W + 2 * pad - g->k + 1;
    if (Ho != g->H || Wo != g->W) return GDN_ERR_UNSUPPORTED;      // "same" heads only
    const int tiles_x = cdiv(g->W, HD_TW), tiles_y = cdiv(g->H, HD_TH);
    hipLaunchKernelGGL((conv_head_kernel<9>), dim3(tiles_x * tiles_y * g->B), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, w, y, g->B, g->H, g->W, g->Cin, pad, g->transposed ? 1 : 0, act, tiles_x, tiles_y, x_bf16);
    return gdn_launch_status();
}
