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
#include "gemm_x3.h"

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

// ---- bf16 activations: the head on the matrix pipe (round 4) ----
// An N = 1 GEMM wastes an MFMA tile -- but the bf16 pipe is 16x the packed-fp32 VALU rate, and the waste disappears when the 81
// TAPS are the GEMM's rows: for one input row, D[tap][q] = sum_c w[tap][c] x[row][q][c] is a [96 x 64] x [64 x 32] product (81 taps
// padded to 3 row tiles; q = the 24 patch pixels of a 16-column output strip), and out[oy][ox] = sum_{ky,kx} D_{oy+ky-4}[ky*9+kx][ox+kx]
// is a shifted gather of it.  A workgroup streams down one strip: per input row three waves run 4 k-steps x 3 weight terms (the
// fp32 weights are split into three bf16 terms once, in registers: bf16 x bf16 products are exact in the fp32 accumulator, so this
// is the fp32-weight product of the VALU kernel, not a rounded one), D goes to LDS, and 144 threads -- (output row mod 9, column) --
// add the nine values of their filter row: thread (s, ox) owns the output rows oy = s (mod 9), which take exactly one filter row
// from every input row, and writes a row out when its ninth has arrived.  x is read once (+ the 8-column halo).
#define HM_SW 16
typedef __bf16 hm_bf16x8 __attribute__((ext_vector_type(8)));
typedef float hm_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned hm_bf16_rn(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// XF32 (round 4): fp32 activations -- the loader wave splits every value into three exact bf16 terms (x3_split2, gemm_x3.h) and
// writes three images; the tap waves run the six products of the bf16 x 3 scheme (w1 x1 into one accumulator, the five correction
// products into a second one, as gemm_x3.hip does): the fp32 head on the matrix pipe, 440 -> ~250 us at B = 20.
template <bool XF32>
__global__ __launch_bounds__(256) void conv_head_mfma_kernel(const void* __restrict__ xv, int ldx, const float* __restrict__ w,
                                                             float* __restrict__ y, int B, int H, int W, int pad, int flip, int act,
                                                             int strips, int segs) {
    const unsigned short* __restrict__ x = reinterpret_cast<const unsigned short*>(xv);
    const float* __restrict__ xf = reinterpret_cast<const float*>(xv);
    constexpr int NX = XF32 ? 3 : 1;                                  // bf16 images of an input row
    __shared__ float Dl[2][96][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int strip = bid % strips; bid /= strips;
    const int seg = bid % segs, b = bid / segs;
    const int x0 = strip * HM_SW;
    const int rows_seg = (H + segs - 1) / segs;
    const int oy_lo = seg * rows_seg, oy_hi = min(H, oy_lo + rows_seg);          // output rows of this workgroup
    const int r = lane & 31, h = lane >> 5;
    // weights: tap tile `wave`, three bf16 terms per value (round to nearest), 4 k-steps of 16 channels
    hm_bf16x8 wa[3][4];
    {
        const int tap = wave * 32 + r;
        const bool ok = wave < 3 && tap < 81;
        const int ky = tap / 9, kx = tap - ky * 9;
        const int widx = flip ? (8 - ky) * 9 + (8 - kx) : tap;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            unsigned short t[3][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = ok ? w[(size_t)widx * 64 + 16 * s + 8 * h + j] : 0.f;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const unsigned hb = hm_bf16_rn(v);
                    t[p][j] = (unsigned short)hb;
                    v -= __uint_as_float(hb << 16);
                }
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                uint4 u = {(unsigned)t[p][0] | ((unsigned)t[p][1] << 16), (unsigned)t[p][2] | ((unsigned)t[p][3] << 16),
                           (unsigned)t[p][4] | ((unsigned)t[p][5] << 16), (unsigned)t[p][6] | ((unsigned)t[p][7] << 16)};
                wa[p][s] = __builtin_bit_cast(hm_bf16x8, u);
            }
        }
    }
    // x rows: wave 3 (no tap tile of its own) is the loader -- whole 128-byte lines (8 lanes per pixel, 24 patch pixels = 3 loads
    // per lane and row), four rows ahead in registers, written to a two-slot LDS image one row ahead of its use; the three MFMA
    // waves read their B fragments from there.  (Fragment-shaped loads straight into the MFMA waves' registers -- 32 lines per
    // instruction, three times over -- kept the texture addresser busy 8x as long: 191 us instead of 130 at B = 20; the VALU kernel above: 440.)
    // image: [pixel q][8 chunks of 16 B], chunk XOR-permuted by (q >> 1) & 7 (conflict-free ds_read_b128 over 32 pixels)
    __shared__ __attribute__((aligned(16))) unsigned char Xl[NX][2][32 * 128];
    constexpr int PF = 4;
    constexpr int NL = XF32 ? 6 : 3;                                 // 16-byte loads per lane and row (fp32: two 32-channel halves per pixel)
    const int lq = lane >> 3, lc = lane & 7;                         // loader: pixel within a piece of 8, logical chunk
    uint4 ring[PF][NL];
    auto load_row = [&](int iy, uint4 (&f)[NL]) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int q = e * 8 + lq, col = x0 - pad + q;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)col < (unsigned)W;
            const size_t px = ((size_t)b * H * W + (size_t)(ok ? iy : 0) * W + (ok ? col : 0)) * ldx;
            if constexpr (XF32) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const uint4 v = *reinterpret_cast<const uint4*>(xf + px + 32 * hh + 4 * lc);
                    f[2 * e + hh] = ok ? v : uint4{0u, 0u, 0u, 0u};
                }
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(x + px + 8 * lc);
                f[e] = ok ? v : uint4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto store_row = [&](int slot, const uint4 (&f)[NL]) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int q = e * 8 + lq;
            if constexpr (XF32) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    // channels 32 hh + 4 lc .. + 3: half of the 16-byte chunk lc / 2 + 4 hh of each term's image
                    const uint4 v = f[2 * e + hh];
                    unsigned p0[3], p1[3];
                    x3_split2(__uint_as_float(v.x), __uint_as_float(v.y), p0[0], p0[1], p0[2]);
                    x3_split2(__uint_as_float(v.z), __uint_as_float(v.w), p1[0], p1[1], p1[2]);
                    const int chunk = (lc >> 1) + 4 * hh;
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        *reinterpret_cast<uint2*>(&Xl[p][slot][q * 128 + ((chunk ^ ((q >> 1) & 7)) << 4) + (lc & 1) * 8]) = uint2{p0[p], p1[p]};
                }
            } else {
                *reinterpret_cast<uint4*>(&Xl[0][slot][q * 128 + ((lc ^ ((q >> 1) & 7)) << 4)]) = f[e];
            }
        }
    };
    const int s9 = tid / HM_SW, ox = tid % HM_SW;                    // gather role (tid < 144)
    float accv = 0.f;
    const int iy_lo = oy_lo - pad, iy_hi = oy_hi - 1 + (8 - pad);     // input rows this segment needs (may lie outside the image)
    if (wave == 3) {
#pragma unroll
        for (int j = 0; j < PF; ++j) load_row(iy_lo + j, ring[j]);
        store_row(0, ring[0]);
        load_row(iy_lo + PF, ring[0]);
    }
#pragma unroll
    for (int p = 0; p < NX; ++p)
        if (tid < 128) *reinterpret_cast<uint4*>(&Xl[p][tid >> 6][(24 + ((tid >> 3) & 7)) * 128 + ((tid & 7) << 4)]) = uint4{0u, 0u, 0u, 0u};   // patch pixels 24..31: zeros
    __syncthreads();
    const unsigned bq = (unsigned)r * 128u, bkey = (unsigned)((r >> 1) & 7);
    for (int iy0 = iy_lo; iy0 <= iy_hi; iy0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int iy = iy0 + j;
            if (iy > iy_hi) break;
            const int buf = j & 1;
            const bool in_img = (unsigned)iy < (unsigned)H;
            if (wave < 3 && in_img) {
                hm_f32x16 acc, cor;
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[e] = 0.f; cor[e] = 0.f; }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const unsigned off = bq + ((((unsigned)(2 * s2 + h)) ^ bkey) << 4);
                    const hm_bf16x8 b0 = __builtin_bit_cast(hm_bf16x8, *reinterpret_cast<const uint4*>(&Xl[0][buf][off]));
                    if constexpr (XF32) {
                        const hm_bf16x8 b1 = __builtin_bit_cast(hm_bf16x8, *reinterpret_cast<const uint4*>(&Xl[1][buf][off]));
                        const hm_bf16x8 b2 = __builtin_bit_cast(hm_bf16x8, *reinterpret_cast<const uint4*>(&Xl[NX - 1][buf][off]));
                        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[2][s2], b0, cor, 0, 0, 0);
                        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][s2], b1, cor, 0, 0, 0);
                        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][s2], b2, cor, 0, 0, 0);
                        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][s2], b0, cor, 0, 0, 0);
                        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][s2], b1, cor, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][s2], b0, acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int p = 0; p < 3; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[p][s2], b0, acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) Dl[buf][wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h][r] = XF32 ? acc[e] + cor[e] : acc[e];
            }
            if (wave == 3) {
                // ring[(j + 1) % PF] holds row iy + 1 (ring[j] was re-used for row iy + PF one iteration ago, or in the prologue)
                store_row(buf ^ 1, ring[(j + 1) % PF]);
                load_row(iy + 1 + PF, ring[(j + 1) % PF]);
            }
            __syncthreads();
            if (tid < 9 * HM_SW) {
                const int base = iy - (8 - pad);                          // the oldest output row this input row still feeds
                const int oy = base + (((s9 - base) % 9) + 9) % 9;        // this thread's row among base .. base + 8
                const int ky = iy - oy + pad;                             // 0 .. 8
                const bool mine = oy >= oy_lo && oy < oy_hi;
                if (mine && in_img) {
                    float a = 0.f;
#pragma unroll
                    for (int kx = 0; kx < 9; ++kx) a += Dl[buf][ky * 9 + kx][ox + kx];
                    accv += a;
                }
                if (oy == base) {                                         // last filter row of output row oy: done
                    if (mine && x0 + ox < W) {
                        float v = accv;
                        if (act == GDN_ACT_TANH) v = tanhf(v);
                        y[((size_t)b * H + oy) * W + x0 + ox] = v;
                    }
                    accv = 0.f;
                }
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
    if (g->Cin == 64 && (ldx % (x_bf16 ? 8 : 4)) == 0) {
        // bf16 activations: the matrix-pipe kernel; every image is cut into row segments until the grid gives a CU ~4 workgroups
        const int strips = cdiv(g->W, HM_SW);
        int segs = 1;
        while (segs < 8 && (int64_t)strips * g->B * segs < 1024 && g->H / (segs * 2) >= 16) segs *= 2;
        if (!x_bf16)
            hipLaunchKernelGGL(conv_head_mfma_kernel<true>, dim3(strips * segs * g->B), dim3(256), 0, (hipStream_t)stream, x, ldx, w, y, g->B,
                               g->H, g->W, pad, g->transposed ? 1 : 0, act, strips, segs);
        else
        hipLaunchKernelGGL(conv_head_mfma_kernel<false>, dim3(strips * segs * g->B), dim3(256), 0, (hipStream_t)stream,
                           x, ldx, w, y, g->B, g->H, g->W, pad, g->transposed ? 1 : 0, act, strips, segs);
        return gdn_launch_status();
    }
    const int tiles_x = cdiv(g->W, HD_TW), tiles_y = cdiv(g->H, HD_TH);
    hipLaunchKernelGGL((conv_head_kernel<9>), dim3(tiles_x * tiles_y * g->B), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, w, y, g->B, g->H, g->W, g->Cin, pad, g->transposed ? 1 : 0, act, tiles_x, tiles_y, x_bf16);
    return gdn_launch_status();
}
