// The 1 <-> 64 channel 9x9 layers of the depth networks on the matrix cores: G's first convolution (ConvBlock(1, 64, k9),
// AE_model_unet.py:496), the data gradient of the 64 -> 1 heads (upconv4, :362 / :521) and the weight gradients of both.
// With one channel on one side the implicit-GEMM kernels degenerate: their k index runs over (tap, channel) with a scalar
// gather per element (392 us per launch at B = 20) and the thin weight-gradient kernel reduces 1 M pixels per tap row
// (413 us), for 11 GFLOP each -- about 0.1 ms of MFMA time.  Here the single-channel image is staged as a PATCH in LDS and
// the GEMM operand is built by the LDS read itself: for v_mfma_f32_32x32x2_f32 a lane supplies ONE float, A[row][k], so
//   forward  (M = 32 consecutive pixels of a row, K = taps):   A[m][tap] = patch[(y + ky) * PW + x + m + kx]
//   wgrad    (M = taps, K = pixels):                            A[tap][p] = patch[(py + ky) * PW + px + kx]
// are plain ds_read_b32 with a per-lane base and a compile-time offset; taps are laid out [ky][kx padded to 10] so the two
// k values of one MFMA (kx, kx + 1) are adjacent in the patch.  No im2col, no gather instructions, no per-element bounds
// logic in the loop (the patch loader applies zero / reflection padding once).
#include "common.h"

#define C1_K 9
#define C1_KP 10             // kx padded to even: the pad tap has zero weight (forward) / is dropped (wgrad)
#define C1_N 64              // the wide side: 64 channels
#define C1_TH 8
#define C1_TW 32
#define C1_PH (C1_TH + C1_K - 1)     // 16
#define C1_PW 41                     // 32 + 8 columns, odd pitch

namespace {

struct C1Geom {
    int B, H, W;
    int pad, reflect, flip;
};

// pixel (y, x) of an image whose pixels are `cs` floats apart (cs = 1: a single-channel image; 3: one channel of an RGB one)
__device__ __forceinline__ float c1_fetch(const float* __restrict__ img, int H, int W, int y, int x, int reflect, int pad, int cs = 1) {
    if (reflect) {
        if (y < -pad || y >= H + pad || x < -pad || x >= W + pad) return 0.f;      // only feeds outputs beyond the image
        y = y < 0 ? -y : (y >= H ? 2 * H - 2 - y : y);
        x = x < 0 ? -x : (x >= W ? 2 * W - 2 - x : x);
        return img[((size_t)y * W + x) * cs];
    }
    return (y >= 0 && y < H && x >= 0 && x < W) ? img[((size_t)y * W + x) * cs] : 0.f;
}

// y[p][n] = sum_{tap, c} x1[p + tap - pad][c] * w[tap][n][c]  (+ epilogue).  A workgroup stages the 81 x 64 (x CIN) weights once
// ([c][ky][kx pad 10][64], 23 KB per input channel) and walks 8 x 32 pixel tiles (persistent grid); wave w owns rows 2w, 2w+1 of a
// tile: 2 pixel tiles x 2 channel tiles of 32 x 32.  stats slot = tile.  CIN = 1: the depth networks' first layer and the heads'
// data gradient; CIN = 3 (round 4): R's first layer Conv2d(3, 64, 9) after ReflectionPad2d(4) (AE_model_unet.py:273), whose
// implicit-GEMM form gathers 243 scalars per output (942 us at B = 20 against 0.23 ms of fp32 MFMA time).
template <int CIN>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ w, void* __restrict__ y,
                                                          int ldy, const void* __restrict__ addsrc, int ld_add, int dtypes,
                                                          float* __restrict__ stats, const float* __restrict__ ep_scale,
                                                          const float* __restrict__ ep_shift, int act, C1Geom g, int tiles_x,
                                                          int tiles_y) {
    constexpr int PSZ = C1_PH * C1_PW, WSZ = C1_K * C1_KP * C1_N;
    __shared__ __attribute__((aligned(16))) float patch[CIN * PSZ];
    __shared__ float wl[CIN * WSZ];
    __shared__ float red[4 * 2 * C1_N];
    __shared__ __attribute__((aligned(16))) float scr1[CIN == 1 ? 4 * 256 : 4];        // epilogue scratch (CIN = 3: the patch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    for (int i = tid; i < C1_K * C1_K * C1_N * CIN; i += 256) {     // source order [tap][n][c]: coalesced, no pad column
        const int c = i % CIN, i1 = i / CIN, n = i1 & (C1_N - 1), tap = i1 >> 6, ky = tap / C1_K, kx = tap - ky * C1_K;
        const int src = g.flip ? (C1_K * C1_K - 1 - tap) : tap;
        wl[c * WSZ + (ky * C1_KP + kx) * C1_N + n] = w[((size_t)src * C1_N + n) * CIN + c];
    }
    for (int i = tid; i < C1_K * C1_N * CIN; i += 256) {            // pad tap kx = 9
        const int c = i / (C1_K * C1_N), r = i - c * (C1_K * C1_N);
        wl[c * WSZ + ((r >> 6) * C1_KP + C1_K) * C1_N + (r & (C1_N - 1))] = 0.f;
    }
    const float* pa = patch + (2 * wave) * C1_PW + l32 + h;        // + (mt + ky) * PW + 2j
    const float* pb = wl + h * C1_N + l32;                         // + (ky * KP + 2j) * N + nt * 32
    const int ntiles = g.B * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int x0 = tx * C1_TW, y0 = ty * C1_TH;
        const float* img = x1 + (size_t)b * g.H * g.W * CIN;
        __syncthreads();                                           // previous tile's patch / red fully consumed
        for (int i = tid; i < CIN * PSZ; i += 256) {                // (column 40 only meets the zero-weight pad tap: keep it finite)
            const int c = i / PSZ, i1 = i - c * PSZ, py = i1 / C1_PW, px = i1 - py * C1_PW;
            patch[i] = px < C1_TW + C1_K - 1 ? c1_fetch(img + c, g.H, g.W, y0 + py - g.pad, x0 + px - g.pad, g.reflect, g.pad, CIN) : 0.f;
        }
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int ky = 0; ky < C1_K; ++ky)
#pragma unroll
                for (int j = 0; j < C1_KP / 2; ++j) {
                    const float a0 = pa[c * PSZ + ky * C1_PW + 2 * j], a1 = pa[c * PSZ + (ky + 1) * C1_PW + 2 * j];
                    const float b0 = pb[c * WSZ + (ky * C1_KP + 2 * j) * C1_N], b1 = pb[c * WSZ + (ky * C1_KP + 2 * j) * C1_N + 32];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
        // epilogue: acc[mt][nt][r] = pixel (row y0 + 2*wave + mt, column x0 + (r&3) + 8*(r>>2) + 4*h), channel nt*32 + l32.
        // The tile leaves through a 1 KB per-wave LDS transposition, 8 pixels x 32 channels at a time: 16-byte stores (8 lanes =
        // the 128 bytes of one pixel's 32 channels) instead of 4-byte stores in the MFMA layout -- the kernel was store-issue-bound
        // (272 MB in 216 us at B = 20).  CIN = 3 has no LDS to spare (two workgroups per CU): its scratch is the patch.
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        if (CIN != 1) __syncthreads();                             // every wave is done reading the patch
        float* scr = (CIN == 1 ? scr1 : patch) + wave * 256;
        const int rrow = lane >> 3, rc4 = (lane & 7) * 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int oy = y0 + 2 * wave + mt;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = nt * 32 + l32;
                const float es = ep_scale ? ep_scale[n] : 1.f, et = ep_shift ? ep_shift[n] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * q + e, ox = x0 + 8 * q + e + 4 * h;
                        float val = acc[mt][nt][r];
                        if (oy < g.H && ox < g.W) { s1[nt] += val; s2[nt] += val * val; }
                        if (ep_scale) val = val * es + et;
                        if (act & GDN_ACT_RELU) val = fmaxf(val, 0.f);
                        scr[(e + 4 * h) * 32 + l32] = val;
                    }
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&scr[rrow * 32 + rc4]);
                    const int ox = x0 + 8 * q + rrow;
                    if (oy < g.H && ox < g.W) {
                        const size_t px = (size_t)(b * g.H + oy) * g.W + ox;
                        f32x4 o = v;
                        if (addsrc) o += ld4_any(addsrc, px * ld_add + nt * 32 + rc4, dtypes & 2);
                        st4_any(y, px * ldy + nt * 32 + rc4, o, dtypes & 1);
                    }
                }
            }
        }
        if (stats) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                s1[nt] += __shfl_xor(s1[nt], 32, 64);
                s2[nt] += __shfl_xor(s2[nt], 32, 64);
                if (h == 0) { red[(wave * 2 + 0) * C1_N + nt * 32 + l32] = s1[nt]; red[(wave * 2 + 1) * C1_N + nt * 32 + l32] = s2[nt]; }
            }
            __syncthreads();
            if (tid < 2 * C1_N) {
                const int which = tid / C1_N, n = tid % C1_N;
                const float v = ((red[(0 * 2 + which) * C1_N + n] + red[(1 * 2 + which) * C1_N + n]) + red[(2 * 2 + which) * C1_N + n]) +
                                red[(3 * 2 + which) * C1_N + n];
                stats[((size_t)t * 2 + which) * C1_N + n] = v;
            }
        }
    }
}

// Weight gradient: D[tap][n] = sum_p gw[p][n] * x1[p + tap - pad].  Persistent workgroups walk 4 x 32 pixel tiles; wave w reduces
// row w of a tile into six 32 x 32 accumulators (96 tap rows [ky][kx pad 10] x 64 channels); the four waves meet in LDS at
// the end and the workgroup writes one partial [96][64] (summed in fixed order by conv_c1_wgrad_reduce_kernel).
#define C1W_TH 4
#define C1W_PH (C1W_TH + C1_K - 1)      // 12
__global__ __launch_bounds__(256, 2) void conv_c1_wgrad_kernel(const float* __restrict__ x1, const void* __restrict__ gw, int ldg, int gw_bf16,
                                                            float* __restrict__ part, C1Geom g, int tiles_x, int tiles_y) {
    __shared__ float patch[C1W_PH * C1_PW];
    __shared__ __attribute__((aligned(16))) float gt[C1W_TH * C1_TW * C1_N];       // [pixel][64]: 32 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    f32x16 acc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    int tapoff[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int tau = mt * 32 + l32;                       // tap row of this lane: (ky, kx) = (tau / 10, tau % 10); rows >= 90 are padding
        tapoff[mt] = tau < C1_K * C1_KP ? (tau / C1_KP) * C1_PW + tau % C1_KP : 0;
    }
    const int ntiles = g.B * tiles_y * tiles_x;
    // software pipeline: the next tile's 128 x 64 gw values (8 x 16 B per thread) and patch values travel in registers while
    // the MFMAs of the current tile run from LDS
    f32x4 rg[8];
    float rp[2];
    auto fetch = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int x0 = tx * C1_TW, y0 = ty * C1W_TH;
        const float* img = x1 + (size_t)b * g.H * g.W;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = tid + e * 256, py = i / C1_PW, px = i - py * C1_PW;
            rp[e] = (i < C1W_PH * C1_PW && px < C1_TW + C1_K - 1)
                        ? c1_fetch(img, g.H, g.W, y0 + py - g.pad, x0 + px - g.pad, g.reflect, g.pad) : 0.f;
        }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int q = ps * 16 + (tid >> 4), c4 = (tid & 15) * 4;
            const int oy = y0 + q / C1_TW, ox = x0 + q % C1_TW;
            rg[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (oy < g.H && ox < g.W) rg[ps] = ld4_any(gw, ((size_t)(b * g.H + oy) * g.W + ox) * ldg + c4, gw_bf16);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                                     // previous tile fully consumed
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (tid + e * 256 < C1W_PH * C1_PW) patch[tid + e * 256] = rp[e];
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) *reinterpret_cast<f32x4*>(&gt[(ps * 16 + (tid >> 4)) * C1_N + (tid & 15) * 4]) = rg[ps];
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
        const float* pa = patch + wave * C1_PW + h;           // + tapoff + px
        const float* pb = gt + (wave * C1_TW + h) * C1_N + l32;   // + px * N + nt * 32
#pragma unroll
        for (int s = 0; s < C1_TW / 2; ++s) {
            const float b0 = pb[(2 * s) * C1_N], b1 = pb[(2 * s) * C1_N + 32];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const float a = pa[tapoff[mt] + 2 * s];
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[mt][1], 0, 0, 0);
            }
        }
    }
    // reduce the four waves through LDS (reusing the gw tile), fixed order, and write this workgroup's partial
    __syncthreads();
    float* redb = gt;                                         // one [96][64] image (6144 of the tile's 8192 floats), wave by wave
    for (int wv = 1; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        redb[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * C1_N + nt * 32 + l32] = acc[mt][nt][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[mt][nt][r] += redb[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * C1_N + nt * 32 + l32];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* dst = part + (size_t)blockIdx.x * 96 * C1_N;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * C1_N + nt * 32 + l32] = acc[mt][nt][r];
    }
}

// dw[tap][n] = sum over workgroups of part[wg][ky * 10 + kx][n]  (fixed order; pad taps dropped; taps flipped when asked).
// One block per tap: thread (n, grp) sums the workgroups grp, grp + 4, ... with eight independent accumulators, the four
// groups meet in LDS.
__global__ __launch_bounds__(256) void conv_c1_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nwg, int flip) {
    __shared__ float sh[4 * C1_N];
    const int tap = blockIdx.x, ky = tap / C1_K, kx = tap % C1_K;
    const int n = threadIdx.x & (C1_N - 1), grp = threadIdx.x >> 6;
    const float* src = part + (size_t)(ky * C1_KP + kx) * C1_N + n;
    const size_t ws = (size_t)96 * C1_N;
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    int wg = grp;
    for (; wg + 28 < nwg; wg += 32) {
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += src[(size_t)(wg + 4 * e) * ws];
    }
    for (; wg < nwg; wg += 4) a[0] += src[(size_t)wg * ws];
    sh[grp * C1_N + n] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (grp == 0) {
        const int to = flip ? (C1_K * C1_K - 1 - tap) : tap;
        dw[(size_t)to * C1_N + n] = (sh[n] + sh[C1_N + n]) + (sh[2 * C1_N + n] + sh[3 * C1_N + n]);
    }
}

bool c1_ok(int32_t B, int32_t H, int32_t W, int32_t N, int32_t k, int32_t pad, int32_t reflect) {
    if (B <= 0 || H <= 0 || W <= 0 || N != C1_N || k != C1_K || pad != C1_K / 2) return false;
    if (reflect && (pad >= H || pad >= W)) return false;
    return true;
}
constexpr int C1_WGRAD_WGS = 1024;
constexpr int C1_FWD_WGS = 1024;      // persistent forward grid: 4 workgroups per CU, the weights staged once each

}  // namespace

extern "C" int64_t gdn_conv_c1_stats_slots(int32_t B, int32_t H, int32_t W) {
    return (int64_t)B * cdiv(H, C1_TH) * cdiv(W, C1_TW);
}

extern "C" int gdn_conv_c1_fwd(const float* x1, int32_t Cin, int32_t B, int32_t H, int32_t W, int32_t N, int32_t k, int32_t pad, int32_t reflect,
                               int32_t flip, const float* w, void* y, int32_t ldy, const void* addsrc, int32_t ld_add,
                               float* stats, const float* ep_scale, const float* ep_shift, int32_t act, int32_t dtypes,
                               void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!c1_ok(B, H, W, N, k, pad, reflect) || (Cin != 1 && Cin != 3)) return GDN_ERR_UNSUPPORTED;
    if (!x1 || !w || !y || (!ep_scale) != (!ep_shift) || (act & GDN_ACT_TANH) || (dtypes & ~3)) return GDN_ERR_BAD_ARG;
    if ((ldy % 4) || (addsrc && (ld_add % 4))) return GDN_ERR_UNSUPPORTED;          // 16-byte (fp32) / 8-byte (bf16) accesses
    const C1Geom g = {B, H, W, pad, reflect ? 1 : 0, flip ? 1 : 0};
    const int tiles_x = cdiv(W, C1_TW), tiles_y = cdiv(H, C1_TH), ntiles = B * tiles_x * tiles_y;
    if (Cin == 1)
        hipLaunchKernelGGL(conv_c1_fwd_kernel<1>, dim3(ntiles < C1_FWD_WGS ? ntiles : C1_FWD_WGS), dim3(256), 0, (hipStream_t)stream, x1, w,
                           y, ldy, addsrc, ld_add, (int)dtypes, stats, ep_scale, ep_shift, act, g, tiles_x, tiles_y);
    else          // 77 KB of LDS per workgroup: two per CU
        hipLaunchKernelGGL(conv_c1_fwd_kernel<3>, dim3(ntiles < C1_FWD_WGS / 2 ? ntiles : C1_FWD_WGS / 2), dim3(256), 0, (hipStream_t)stream,
                           x1, w, y, ldy, addsrc, ld_add, (int)dtypes, stats, ep_scale, ep_shift, act, g, tiles_x, tiles_y);
    return gdn_launch_status();
}

extern "C" size_t gdn_conv_c1_wgrad_workspace_bytes(void) { return (size_t)C1_WGRAD_WGS * 96 * C1_N * sizeof(float); }

extern "C" int gdn_conv_c1_wgrad(const float* x1, const void* gw, int32_t ldg, int32_t gw_bf16, int32_t B, int32_t H, int32_t W, int32_t N,
                                 int32_t k, int32_t pad, int32_t reflect, int32_t flip, float* dw, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!c1_ok(B, H, W, N, k, pad, reflect)) return GDN_ERR_UNSUPPORTED;
    if (!x1 || !gw || !dw || (ldg % 4)) return GDN_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gdn_conv_c1_wgrad_workspace_bytes()) return GDN_ERR_WORKSPACE;
    const C1Geom g = {B, H, W, pad, reflect ? 1 : 0, 0};
    const int tiles_x = cdiv(W, C1_TW), tiles_y = cdiv(H, C1W_TH);
    const int ntiles = B * tiles_x * tiles_y;
    const int nwg = ntiles < C1_WGRAD_WGS ? ntiles : C1_WGRAD_WGS;
    hipLaunchKernelGGL(conv_c1_wgrad_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, x1, gw, ldg, gw_bf16 ? 1 : 0, (float*)workspace,
                       g, tiles_x,
                       tiles_y);
    hipLaunchKernelGGL(conv_c1_wgrad_reduce_kernel, dim3(C1_K * C1_K), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, dw, nwg, flip ? 1 : 0);
    return gdn_launch_status();
}
